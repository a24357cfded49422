// rtu_multi.hip — several GPUs behind ONE handle of the C-ABI (include/rtu_render.h, "Several GPUs"): what the reference's
// SpawnRenderThreads does with hardware_concurrency() CPU workers (main.cpp:29-64) — fan the frame out, wait, hand back
// one image — for the GPUs of a node. Built only on the single-GPU entry points of the same header, so a caller who
// flattens the reference's own scene graph (INTEGRATION.md option B) gets every GPU without re-implementing the fan-out.
//
// Sharding: interleaved 8-row bands, band b -> context b mod G, the scene replicated. One host thread per
// context renders its shard (the frame-record capacity rounds and, for sampled frames, the per-batch synchronisations of
// one GPU must not serialise the others); the float4 shards are then COLLECTED CONCURRENTLY:
//   * G distinct GPUs and RCCL (librccl.so through dlopen: a single-GPU host needs none of it): one grouped
//     ncclSend / ncclRecv gather into the root GPU's buffer — every GPU writes its shard to the root over its own xGMI
//     link, no ring — queued on the contexts' streams, then one copy of the gathered shards to pinned host memory;
//   * otherwise (several contexts on one GPU, no RCCL, or RCCL reporting an error): one asynchronous copy per context
//     into the pinned staging buffer, all of them in flight together.
// The staging buffer is shard-major; the bands are then copied to their rows of the caller's frame, a context at a time
// as its transfer completes, and handed to the progress callback (RenderImage::IncrementNumRenderPixel, scene.h:585-588).
#include "rtu_render.h"

#include <dlfcn.h>

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <set>
#include <string>
#include <thread>
#include <vector>

struct RtuMultiContext {
    std::vector<int>         devices;
    std::vector<RtuContext*> ctx;
    std::string              error;
    int                      gather_kind = 0;
    volatile int             cancel_word = 0;   // what the contexts poll (rtu_set_cancel_flag) when the caller gave no flag of its own
};

namespace {

// RCCL through dlopen: the library does not depend on it.
struct Rccl {
    typedef void* comm_t;
    int (*CommInitAll)(comm_t*, int, const int*) = nullptr;
    int (*CommDestroy)(comm_t) = nullptr;
    int (*CommAbort)(comm_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*Send)(const void*, size_t, int, int, comm_t, void*) = nullptr;
    int (*Recv)(void*, size_t, int, int, comm_t, void*) = nullptr;
    void* lib = nullptr;
    bool load() {
        if (lib) return true;
        for (const char* name : {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
            lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (lib) break;
        }
        if (!lib) return false;
        CommInitAll = (decltype(CommInitAll))dlsym(lib, "ncclCommInitAll");
        CommDestroy = (decltype(CommDestroy))dlsym(lib, "ncclCommDestroy");
        CommAbort = (decltype(CommAbort))dlsym(lib, "ncclCommAbort");
        GroupStart = (decltype(GroupStart))dlsym(lib, "ncclGroupStart");
        GroupEnd = (decltype(GroupEnd))dlsym(lib, "ncclGroupEnd");
        Send = (decltype(Send))dlsym(lib, "ncclSend");
        Recv = (decltype(Recv))dlsym(lib, "ncclRecv");
        return CommInitAll && CommDestroy && GroupStart && GroupEnd && Send && Recv;
    }
};
const int kNcclFloat = 7;  // ncclFloat32 (nccl.h: ncclDataType_t)

struct Shard {
    RtuFrameDesc frame{};
    void*        d_rgbz = nullptr;
    size_t       floats = 0;   // rows * W * 4
    size_t       offset = 0;   // of this shard in the staging buffer, in floats
    int          rc = RTU_OK;
    std::string  error;
};

int fail(RtuMultiContext* m, int code, const std::string& what) {
    if (m) m->error = what;
    return code;
}

bool cancelled(const RtuProgress* p) { return p && p->cancel && *p->cancel; }

}  // namespace

extern "C" {

RtuMultiContext* rtu_create_context_multi(const int* device_ids, int n_devices, int* err_out) {
    if (err_out) *err_out = RTU_OK;
    if (!device_ids || n_devices < 1 || n_devices > 64) {
        if (err_out) *err_out = RTU_ERR_ARG;
        return nullptr;
    }
    RtuMultiContext* m = new RtuMultiContext;
    m->devices.assign(device_ids, device_ids + n_devices);
    for (int g = 0; g < n_devices; g++) {
        int err = 0;
        RtuContext* c = rtu_create_context(device_ids[g], &err);
        if (!c) {
            if (err_out) *err_out = err;
            for (RtuContext* p : m->ctx) rtu_destroy_context(p);
            delete m;
            return nullptr;
        }
        m->ctx.push_back(c);
    }
    return m;
}

void rtu_destroy_context_multi(RtuMultiContext* m) {
    if (!m) return;
    for (RtuContext* c : m->ctx) rtu_destroy_context(c);
    delete m;
}

int rtu_multi_size(const RtuMultiContext* m) { return m ? (int)m->ctx.size() : 0; }

RtuContext* rtu_multi_context(RtuMultiContext* m, int i) { return (m && i >= 0 && i < (int)m->ctx.size()) ? m->ctx[(size_t)i] : nullptr; }

const char* rtu_multi_last_error(const RtuMultiContext* m) { return m ? m->error.c_str() : "NULL multi-context"; }

int rtu_multi_gather_kind(const RtuMultiContext* m) { return m ? m->gather_kind : 0; }

int rtu_multi_upload_scene(RtuMultiContext* m, const RtuSceneDesc* scene) {
    if (!m) return RTU_ERR_ARG;
    // the scene is replicated: every context validates, builds its acceleration structures and uploads — one thread per GPU
    const int G = (int)m->ctx.size();
    std::vector<int> rcs((size_t)G, RTU_OK);
    std::vector<std::thread> th;
    for (int g = 0; g < G; g++) th.emplace_back([&, g] { rcs[(size_t)g] = rtu_upload_scene(m->ctx[(size_t)g], scene); });
    for (std::thread& t : th) t.join();
    for (int g = 0; g < G; g++)
        if (rcs[(size_t)g] != RTU_OK) return fail(m, rcs[(size_t)g], std::string("GPU ") + std::to_string(m->devices[(size_t)g]) + ": " + rtu_last_error(m->ctx[(size_t)g]));
    return RTU_OK;
}

int rtu_multi_render_frame(RtuMultiContext* m, const RtuFrameDesc* frame, float* h_rgbz, const RtuProgress* progress) {
    if (!m || !frame) return RTU_ERR_ARG;
    if (!h_rgbz && !(progress && progress->rows_done)) return fail(m, RTU_ERR_ARG, "neither a frame buffer nor a rows_done callback: nowhere to put the image");
    const int G = (int)m->ctx.size();
    const int W = frame->width, H = frame->height;
    if (W <= 0 || H <= 0) return fail(m, RTU_ERR_ARG, "bad resolution");
    m->gather_kind = 0;
    std::vector<Shard> sh((size_t)G);
    size_t total = 0;
    int rc = RTU_OK;
    const volatile int* flag = (progress && progress->cancel) ? progress->cancel : &m->cancel_word;
    for (int g = 0; g < G; g++) {
        Shard& s = sh[(size_t)g];
        s.frame = *frame;
        s.frame.shard_rank = g;
        s.frame.shard_count = G;
        s.floats = (size_t)rtu_shard_rows(&s.frame) * (size_t)W * 4;
        s.offset = total;
        total += s.floats;
        rtu_set_cancel_flag(m->ctx[(size_t)g], flag);
    }
    // ---- render: one host thread per context; each settles its own frame-record capacities
    {
        std::vector<std::thread> th;
        for (int g = 0; g < G; g++) {
            if (sh[(size_t)g].floats == 0) continue;
            th.emplace_back([&, g] {
                Shard& s = sh[(size_t)g];
                RtuContext* c = m->ctx[(size_t)g];
                s.d_rgbz = rtu_device_alloc(c, s.floats * sizeof(float));
                if (!s.d_rgbz) { s.rc = RTU_ERR_HIP; s.error = "device allocation failed"; return; }
                s.rc = rtu_render_frame_device(c, &s.frame, s.d_rgbz, rtu_context_stream(c));
                // more Shade() frames than provisioned: the context has grown its buffers, render the shard again
                for (int round = 0; s.rc == RTU_OK && (s.rc = rtu_frame_status(c)) == RTU_ERR_CAPACITY && round < 16; round++) {
                    if (*flag) { s.rc = RTU_ERR_CANCELLED; break; }
                    s.rc = rtu_render_frame_device(c, &s.frame, s.d_rgbz, rtu_context_stream(c));
                }
                if (s.rc != RTU_OK) s.error = rtu_last_error(c);
            });
        }
        for (std::thread& t : th) t.join();
    }
    for (int g = 0; g < G && rc == RTU_OK; g++)
        if (sh[(size_t)g].rc != RTU_OK) rc = fail(m, sh[(size_t)g].rc, std::string("GPU ") + std::to_string(m->devices[(size_t)g]) + ": " + sh[(size_t)g].error);
    if (rc == RTU_OK && (*flag || cancelled(progress))) rc = fail(m, RTU_ERR_CANCELLED, "cancelled");

    // ---- collect
    float* host = nullptr;
    if (rc == RTU_OK && total) {
        host = (float*)rtu_host_alloc_pinned(total * sizeof(float));
        if (!host) rc = fail(m, RTU_ERR_HIP, "pinned host allocation failed");
    }
    bool gathered = false;
    if (rc == RTU_OK && total) {
        std::set<int> distinct(m->devices.begin(), m->devices.end());
        // RTU_FORCE_RCCL (tests on a one-GPU box): take the RCCL path with a single context too — a communicator of one, an empty group
        const bool want_rccl = (G > 1 || getenv("RTU_FORCE_RCCL")) && (int)distinct.size() == G && !getenv("RTU_NO_RCCL");
        Rccl nccl;
        if (want_rccl && nccl.load()) {
            std::vector<Rccl::comm_t> comms((size_t)G, nullptr);
            if (nccl.CommInitAll(comms.data(), G, m->devices.data()) == 0) {
                void* d_root = rtu_device_alloc(m->ctx[0], total * sizeof(float));
                bool ok = d_root != nullptr;
                if (ok && nccl.GroupStart() == 0) {
                    // one grouped send / receive: every GPU's shard lands at its offset of the root's buffer. A group that has been
                    // opened is ALWAYS closed, whatever a call inside it returned: the first error is kept and decides afterwards.
                    int first_err = 0;
                    for (int g = 1; g < G && first_err == 0; g++) {
                        if (sh[(size_t)g].floats == 0) continue;
                        first_err = nccl.Recv((float*)d_root + sh[(size_t)g].offset, sh[(size_t)g].floats, kNcclFloat, g, comms[0], rtu_context_stream(m->ctx[0]));
                        if (first_err == 0)
                            first_err = nccl.Send(sh[(size_t)g].d_rgbz, sh[(size_t)g].floats, kNcclFloat, 0, comms[(size_t)g], rtu_context_stream(m->ctx[(size_t)g]));
                    }
                    const int end_err = nccl.GroupEnd();
                    ok = first_err == 0 && end_err == 0;
                } else {
                    ok = false;
                }
                // the root's own shard is copied from its own buffer; the others from the gathered one, all on the root's stream
                if (ok) ok = rtu_copy_to_host_async(m->ctx[0], host, sh[0].d_rgbz, sh[0].floats * sizeof(float), rtu_context_stream(m->ctx[0])) == RTU_OK;
                if (ok && total > sh[0].floats)
                    ok = rtu_copy_to_host_async(m->ctx[0], host + sh[0].floats, (float*)d_root + sh[0].floats, (total - sh[0].floats) * sizeof(float),
                                                rtu_context_stream(m->ctx[0])) == RTU_OK;
                // whatever was queued is waited for before anything else touches the streams or the communicators go away
                for (int g = 0; g < G; g++) ok = (rtu_context_sync(m->ctx[(size_t)g]) == RTU_OK) && ok;
                for (Rccl::comm_t c : comms)
                    if (c) { if (ok || !nccl.CommAbort) nccl.CommDestroy(c); else nccl.CommAbort(c); }
                if (d_root) rtu_device_free(m->ctx[0], d_root);
                gathered = ok;
                if (gathered) m->gather_kind = 3;
            }
        }
        if (!gathered) {
            // asynchronous copies into the pinned staging buffer: every context's transfer is queued before any is awaited
            for (int g = 0; g < G && rc == RTU_OK; g++) {
                if (sh[(size_t)g].floats == 0) continue;
                rc = rtu_copy_to_host_async(m->ctx[(size_t)g], host + sh[(size_t)g].offset, sh[(size_t)g].d_rgbz, sh[(size_t)g].floats * sizeof(float),
                                            rtu_context_stream(m->ctx[(size_t)g]));
                if (rc != RTU_OK) fail(m, rc, rtu_last_error(m->ctx[(size_t)g]));
            }
            if (rc == RTU_OK) m->gather_kind = G > 1 ? 2 : 1;
        }
    }
    // ---- hand the bands over, a context at a time as its transfer completes
    for (int g = 0; g < G && rc == RTU_OK; g++) {
        if (cancelled(progress) || *flag) { rc = fail(m, RTU_ERR_CANCELLED, "cancelled"); break; }
        if (!gathered) {
            rc = rtu_context_sync(m->ctx[(size_t)g]);
            if (rc != RTU_OK) { fail(m, rc, rtu_last_error(m->ctx[(size_t)g])); break; }
        }
        const Shard& s = sh[(size_t)g];
        const int rows = rtu_shard_rows(&s.frame);
        for (int lr = 0; lr < rows; lr += RTU_BAND_ROWS) {
            const int n = rows - lr < RTU_BAND_ROWS ? rows - lr : RTU_BAND_ROWS;
            const int row0 = rtu_shard_global_row(&s.frame, lr);
            const float* src = host + s.offset + (size_t)lr * (size_t)W * 4;
            if (h_rgbz) memcpy(h_rgbz + (size_t)row0 * (size_t)W * 4, src, (size_t)n * (size_t)W * 4 * sizeof(float));
            if (progress && progress->rows_done) progress->rows_done(progress->user, h_rgbz ? h_rgbz + (size_t)row0 * (size_t)W * 4 : src, row0, n);
        }
    }
    if (rc != RTU_OK)  // nothing may still be writing into the staging buffer when it is freed
        for (int g = 0; g < G; g++) (void)rtu_context_sync(m->ctx[(size_t)g]);
    rtu_host_free_pinned(host);
    for (int g = 0; g < G; g++) {
        if (sh[(size_t)g].d_rgbz) rtu_device_free(m->ctx[(size_t)g], sh[(size_t)g].d_rgbz);
        rtu_set_cancel_flag(m->ctx[(size_t)g], nullptr);
    }
    return rc;
}

}  // extern "C"
