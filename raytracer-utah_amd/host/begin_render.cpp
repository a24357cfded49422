// begin_render.cpp — the BeginRender()/StopRender() drop-in (main.cpp:29-72,
// viewport.cpp:36-37: "renderer must run in a separate thread").
//
// The reference detaches one coordinator thread that spawns
// hardware_concurrency() CPU workers, busy-waits, then writes Result.png and
// ZBuffer.png (main.cpp:29-64). Here the coordinator is ONE host thread that
// drives the C-ABI of rtu_render.h: one context per GPU, the frame sharded by
// interleaved 8-row bands, each shard copied back and passed through the
// reference's gamma / Color24 / z-image post-pass; no busy spin.
#include "host_internal.h"
#include "rtu_render.h"

#include <atomic>
#include <string>
#include <thread>
#include <vector>

struct RtuRenderJob {
    std::thread       thread;
    std::atomic<bool> cancel{false};
    std::atomic<int>  result{1};  // 1 = running, 0 = ok, <0 = error
    std::string       error;
};

namespace {

void run_job(RtuRenderJob* job, const RtuSceneDesc* desc, RtuImage* img, std::vector<int> devices, int samples, int gather_bounces,
             std::string result_png, std::string zbuffer_png) {
    const int W = rtu_image_width(img), H = rtu_image_height(img);
    const int G = (int)devices.size();
    std::vector<RtuContext*> ctxs(G, nullptr);
    int rc = RTU_OK;
    auto cleanup = [&]() {
        for (RtuContext* c : ctxs) rtu_destroy_context(c);
    };
    for (int g = 0; g < G && rc == RTU_OK; g++) {
        int err = 0;
        ctxs[g] = rtu_create_context(devices[g], &err);
        if (!ctxs[g]) { rc = err; job->error = rtu_error_string(err); break; }
        rc = rtu_upload_scene(ctxs[g], desc);
        if (rc != RTU_OK) job->error = rtu_last_error(ctxs[g]);
    }
    std::vector<RtuFrameDesc> frames(G);
    std::vector<void*> dbuf(G, nullptr);
    // launch every GPU's shard first (asynchronous), then collect
    for (int g = 0; g < G && rc == RTU_OK; g++) {
        rc = rtu_frame_setup(&desc->camera, W, H, &frames[g]);
        if (rc != RTU_OK) break;
        frames[g].shard_rank = g;
        frames[g].shard_count = G;
        frames[g].samples = samples;
        frames[g].gather_bounces = gather_bounces;
        size_t bytes = (size_t)rtu_shard_rows(&frames[g]) * W * 4 * sizeof(float);
        if (bytes == 0) continue;
        dbuf[g] = rtu_device_alloc(ctxs[g], bytes);
        if (!dbuf[g]) { rc = RTU_ERR_HIP; job->error = "device allocation failed"; break; }
        rc = rtu_render_frame_device(ctxs[g], &frames[g], dbuf[g], nullptr);
        if (rc != RTU_OK) job->error = rtu_last_error(ctxs[g]);
    }
    std::vector<float> shard;
    for (int g = 0; g < G && rc == RTU_OK; g++) {
        if (job->cancel.load()) { rc = RTU_ERR_ARG; job->error = "cancelled"; break; }
        int rows = rtu_shard_rows(&frames[g]);
        if (rows == 0) continue;
        // more Shade() frames than provisioned: the context has grown its buffers, render the shard again
        for (int round = 0; (rc = rtu_frame_status(ctxs[g])) == RTU_ERR_CAPACITY && round < 16; round++) {
            rc = rtu_render_frame_device(ctxs[g], &frames[g], dbuf[g], nullptr);
            if (rc != RTU_OK) break;
        }
        if (rc != RTU_OK) { job->error = rtu_last_error(ctxs[g]); break; }
        shard.resize((size_t)rows * W * 4);
        rc = rtu_copy_to_host(ctxs[g], shard.data(), dbuf[g], shard.size() * sizeof(float));
        if (rc != RTU_OK) { job->error = rtu_last_error(ctxs[g]); break; }
        // de-interleave band by band; bumps the rendered-pixel counter per band
        for (int lr = 0; lr < rows; lr += RTU_BAND_ROWS) {
            int n = rows - lr < RTU_BAND_ROWS ? rows - lr : RTU_BAND_ROWS;
            rtu_image_from_rgbz(img, shard.data() + (size_t)lr * W * 4, rtu_shard_global_row(&frames[g], lr), n);
        }
    }
    for (int g = 0; g < G; g++)
        if (dbuf[g]) rtu_device_free(ctxs[g], dbuf[g]);
    cleanup();
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

void rtu_render_job_free(RtuRenderJob* job) {
    if (!job) return;
    if (job->thread.joinable()) job->thread.join();
    delete job;
}

}  // extern "C"
