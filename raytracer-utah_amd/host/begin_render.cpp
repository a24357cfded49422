// begin_render.cpp — the BeginRender()/StopRender() drop-in (main.cpp:29-72,
// viewport.cpp:36-37: "renderer must run in a separate thread").
//
// The reference detaches one coordinator thread that spawns hardware_concurrency() CPU workers, busy-waits, then
// writes Result.png and ZBuffer.png (main.cpp:29-64). Here the coordinator is ONE host thread that drives the C-ABI of
// rtu_render.h: one context per GPU, the frame sharded by interleaved 8-row bands (band b -> GPU b mod G), every
// shard rendered on its context's own stream, and the float4 shards COLLECTED CONCURRENTLY:
//   * with RCCL (librccl.so, loaded at run time) and G distinct GPUs: one grouped ncclSend / ncclRecv gather into the
//     root GPU's buffer — every GPU writes its shard to the root over its own xGMI link, no ring — queued on the
//     contexts' streams right behind the kernels, then one copy of the whole frame to the host;
//   * otherwise (no RCCL, or several contexts on one GPU): an asynchronous copy per context into one pinned host
//     buffer, all of them in flight together.
// The host then applies the reference's gamma / Color24 / z-image post-pass (image.cpp) band by band and writes the
// PNGs. No busy spin.
#include "host_internal.h"
#include "rtu_render.h"

#include <dlfcn.h>

#include <atomic>
#include <cstdlib>
#include <set>
#include <string>
#include <thread>
#include <vector>

struct RtuRenderJob {
    std::thread       thread;
    std::atomic<bool> cancel{false};
    std::atomic<int>  result{1};  // 1 = running, 0 = ok, <0 = error
    std::atomic<int>  gather_kind{0};  // how the shards were collected: 1 one context, 2 asynchronous host copies, 3 RCCL
    std::string       error;
};

namespace {

// RCCL through dlopen: librtu_host.so does not depend on it (a single-GPU host needs none of it).
struct Rccl {
    typedef void* comm_t;
    int (*CommInitAll)(comm_t*, int, const int*) = nullptr;
    int (*CommDestroy)(comm_t) = nullptr;
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
        GroupStart = (decltype(GroupStart))dlsym(lib, "ncclGroupStart");
        GroupEnd = (decltype(GroupEnd))dlsym(lib, "ncclGroupEnd");
        Send = (decltype(Send))dlsym(lib, "ncclSend");
        Recv = (decltype(Recv))dlsym(lib, "ncclRecv");
        return CommInitAll && CommDestroy && GroupStart && GroupEnd && Send && Recv;
    }
};
const int kNcclFloat = 7;  // ncclFloat32 (nccl.h: ncclDataType_t)

struct Shard {
    RtuContext*  ctx = nullptr;
    RtuFrameDesc frame{};
    void*        d_rgbz = nullptr;
    size_t       floats = 0;   // rows * W * 4
    size_t       offset = 0;   // of this shard in the gathered buffer, in floats
};

void run_job(RtuRenderJob* job, const RtuSceneDesc* desc, RtuImage* img, std::vector<int> devices, int samples, int gather_bounces,
             std::string result_png, std::string zbuffer_png) {
    const int W = rtu_image_width(img), H = rtu_image_height(img);
    const int G = (int)devices.size();
    std::vector<Shard> sh(G);
    int rc = RTU_OK;
    auto fail_with = [&](int code, const std::string& what) { rc = code; job->error = what; };

    for (int g = 0; g < G && rc == RTU_OK; g++) {
        int err = 0;
        sh[g].ctx = rtu_create_context(devices[g], &err);
        if (!sh[g].ctx) { fail_with(err, rtu_error_string(err)); break; }
        rc = rtu_upload_scene(sh[g].ctx, desc);
        if (rc != RTU_OK) job->error = rtu_last_error(sh[g].ctx);
    }
    // launch every GPU's shard on its own stream (asynchronous), then settle capacities, then collect all shards together
    size_t total = 0;
    for (int g = 0; g < G && rc == RTU_OK; g++) {
        Shard& s = sh[g];
        rc = rtu_frame_setup(&desc->camera, W, H, &s.frame);
        if (rc != RTU_OK) break;
        s.frame.shard_rank = g;
        s.frame.shard_count = G;
        s.frame.samples = samples;
        s.frame.gather_bounces = gather_bounces;
        s.floats = (size_t)rtu_shard_rows(&s.frame) * W * 4;
        s.offset = total;
        total += s.floats;
        if (s.floats == 0) continue;
        s.d_rgbz = rtu_device_alloc(s.ctx, s.floats * sizeof(float));
        if (!s.d_rgbz) { fail_with(RTU_ERR_HIP, "device allocation failed"); break; }
        rc = rtu_render_frame_device(s.ctx, &s.frame, s.d_rgbz, rtu_context_stream(s.ctx));
        if (rc != RTU_OK) job->error = rtu_last_error(s.ctx);
    }
    for (int g = 0; g < G && rc == RTU_OK; g++) {
        Shard& s = sh[g];
        if (job->cancel.load()) { fail_with(RTU_ERR_ARG, "cancelled"); break; }
        if (s.floats == 0) continue;
        // more Shade() frames than provisioned: the context has grown its buffers, render the shard again
        for (int round = 0; (rc = rtu_frame_status(s.ctx)) == RTU_ERR_CAPACITY && round < 16; round++) {
            rc = rtu_render_frame_device(s.ctx, &s.frame, s.d_rgbz, rtu_context_stream(s.ctx));
            if (rc != RTU_OK) break;
        }
        if (rc != RTU_OK) job->error = rtu_last_error(s.ctx);
    }

    float* host = nullptr;
    if (rc == RTU_OK && total) {
        host = (float*)rtu_host_alloc_pinned(total * sizeof(float));
        if (!host) fail_with(RTU_ERR_HIP, "pinned host allocation failed");
    }
    if (rc == RTU_OK && total) {
        std::set<int> distinct(devices.begin(), devices.end());
        Rccl nccl;
        std::vector<Rccl::comm_t> comms(G, nullptr);
        void* d_root = nullptr;
        bool gathered = false;
        // RTU_FORCE_RCCL (tests on a one-GPU box): take the RCCL path with a single context too — communicator of one, an empty group
        const bool want_rccl = (G > 1 || getenv("RTU_FORCE_RCCL")) && (int)distinct.size() == G && !getenv("RTU_NO_RCCL");
        if (want_rccl && nccl.load() && nccl.CommInitAll(comms.data(), G, devices.data()) == 0) {
            // one grouped send / receive: every GPU's shard lands at its offset of the root's buffer
            d_root = rtu_device_alloc(sh[0].ctx, total * sizeof(float));
            bool ok = d_root != nullptr && nccl.GroupStart() == 0;
            for (int g = 1; g < G && ok; g++) {
                if (sh[g].floats == 0) continue;
                ok = nccl.Recv((float*)d_root + sh[g].offset, sh[g].floats, kNcclFloat, g, comms[0], rtu_context_stream(sh[0].ctx)) == 0 &&
                     nccl.Send(sh[g].d_rgbz, sh[g].floats, kNcclFloat, 0, comms[g], rtu_context_stream(sh[g].ctx)) == 0;
            }
            ok = ok && nccl.GroupEnd() == 0;
            // the root's own shard is copied from its own buffer; the others from the gathered one, all on the root's stream
            if (ok) ok = rtu_copy_to_host_async(sh[0].ctx, host, sh[0].d_rgbz, sh[0].floats * sizeof(float), rtu_context_stream(sh[0].ctx)) == RTU_OK;
            if (ok && total > sh[0].floats)
                ok = rtu_copy_to_host_async(sh[0].ctx, host + sh[0].floats, (float*)d_root + sh[0].floats, (total - sh[0].floats) * sizeof(float),
                                            rtu_context_stream(sh[0].ctx)) == RTU_OK;
            for (int g = 0; g < G && ok; g++) ok = rtu_context_sync(sh[g].ctx) == RTU_OK;
            gathered = ok;
            if (gathered) job->gather_kind.store(3);
            for (Rccl::comm_t c : comms) if (c) nccl.CommDestroy(c);
            if (d_root) rtu_device_free(sh[0].ctx, d_root);
        }
        if (!gathered) {
            // asynchronous copies into the pinned frame buffer: every context's transfer is queued before any is awaited
            for (int g = 0; g < G && rc == RTU_OK; g++) {
                if (sh[g].floats == 0) continue;
                rc = rtu_copy_to_host_async(sh[g].ctx, host + sh[g].offset, sh[g].d_rgbz, sh[g].floats * sizeof(float), rtu_context_stream(sh[g].ctx));
                if (rc != RTU_OK) job->error = rtu_last_error(sh[g].ctx);
            }
            for (int g = 0; g < G && rc == RTU_OK; g++) {
                rc = rtu_context_sync(sh[g].ctx);
                if (rc != RTU_OK) job->error = rtu_last_error(sh[g].ctx);
            }
            if (rc == RTU_OK) job->gather_kind.store(G > 1 ? 2 : 1);
        }
    }
    // de-interleave band by band (gamma, Color24, z; bumps the rendered-pixel counter per band)
    for (int g = 0; g < G && rc == RTU_OK; g++) {
        if (job->cancel.load()) { fail_with(RTU_ERR_ARG, "cancelled"); break; }
        const int rows = rtu_shard_rows(&sh[g].frame);
        for (int lr = 0; lr < rows; lr += RTU_BAND_ROWS) {
            const int n = rows - lr < RTU_BAND_ROWS ? rows - lr : RTU_BAND_ROWS;
            rtu_image_from_rgbz(img, host + sh[g].offset + (size_t)lr * W * 4, rtu_shard_global_row(&sh[g].frame, lr), n);
        }
    }
    rtu_host_free_pinned(host);
    for (Shard& s : sh) {
        if (s.d_rgbz) rtu_device_free(s.ctx, s.d_rgbz);
        rtu_destroy_context(s.ctx);
    }
    if (rc == RTU_OK) {
        // main.cpp:59-61
        if (!result_png.empty() && rtu_image_save_png(img, result_png.c_str()) != 0) { rc = RTU_ERR_ARG; job->error = "cannot write " + result_png; }
        rtu_image_compute_zimg(img);
        if (rc == RTU_OK && !zbuffer_png.empty() && rtu_image_save_zpng(img, zbuffer_png.c_str()) != 0) { rc = RTU_ERR_ARG; job->error = "cannot write " + zbuffer_png; }
    }
    job->result.store(rc);
}

}  // namespace

extern "C" {

static RtuRenderJob* begin(const RtuScene* scene, RtuImage* img, const int* device_ids, int n_devices, int samples, int gather_bounces,
                           const char* result_png, const char* zbuffer_png) {
    if (!scene || !img || !device_ids || n_devices < 1 || samples < 0) {
        rtu::set_error("rtu_begin_render: bad arguments");
        return nullptr;
    }
    RtuRenderJob* job = new RtuRenderJob;
    std::vector<int> devs(device_ids, device_ids + n_devices);
    job->thread = std::thread(run_job, job, rtu_scene_desc(scene), img, devs, samples, gather_bounces, std::string(result_png ? result_png : ""),
                              std::string(zbuffer_png ? zbuffer_png : ""));
    return job;  // returns immediately, as BeginRender() must
}

RtuRenderJob* rtu_begin_render_sampled(const RtuScene* scene, RtuImage* img, const int* device_ids, int n_devices, int samples,
                                       const char* result_png, const char* zbuffer_png) {
    return begin(scene, img, device_ids, n_devices, samples, 0, result_png, zbuffer_png);
}

RtuRenderJob* rtu_begin_render_paths(const RtuScene* scene, RtuImage* img, const int* device_ids, int n_devices, int samples,
                                     const char* result_png, const char* zbuffer_png) {
    return begin(scene, img, device_ids, n_devices, samples, 4, result_png, zbuffer_png);
}

RtuRenderJob* rtu_begin_render(const RtuScene* scene, RtuImage* img, const int* device_ids, int n_devices,
                               const char* result_png, const char* zbuffer_png) {
    return rtu_begin_render_sampled(scene, img, device_ids, n_devices, 0, result_png, zbuffer_png);
}

void rtu_stop_render(RtuRenderJob* job) {
    if (job) job->cancel.store(true);
}

int rtu_render_wait(RtuRenderJob* job) {
    if (!job) return RTU_ERR_ARG;
    if (job->thread.joinable()) job->thread.join();
    if (job->result.load() != RTU_OK) rtu::set_error(job->error);
    return job->result.load();
}

int rtu_render_gather_kind(RtuRenderJob* job) {
    if (!job) return 0;
    if (job->thread.joinable()) job->thread.join();
    return job->gather_kind.load();
}

void rtu_render_job_free(RtuRenderJob* job) {
    if (!job) return;
    if (job->thread.joinable()) job->thread.join();
    delete job;
}

}  // extern "C"
