// begin_render.cpp — the BeginRender()/StopRender() drop-in (main.cpp:29-72,
// viewport.cpp:36-37: "renderer must run in a separate thread").
//
// The reference detaches one coordinator thread that spawns hardware_concurrency() CPU workers, busy-waits, then
// writes Result.png and ZBuffer.png (main.cpp:29-64). Here the coordinator is ONE host thread that hands the frame to
// the C-ABI's multi-GPU entry (rtu_render.h: rtu_create_context_multi / rtu_multi_render_frame — one context per GPU,
// interleaved 8-row bands, the shards gathered over xGMI with RCCL or by concurrent copies; csrc/rtu_multi.hip), gets the
// bands back as they arrive, applies the reference's gamma / Color24 / z post-pass to each (image.cpp: the rendered-pixel
// counter of the RenderImage mirror advances band by band, scene.h:585-588) and writes the PNGs. No busy spin.
// StopRender() raises the cancel word the library polls (between the sample batches of recipes S / P, between capacity
// rounds, between the shards as they are handed over).
#include "host_internal.h"
#include "rtu_render.h"

#include <atomic>
#include <string>
#include <thread>
#include <vector>

struct RtuRenderJob {
    std::thread       thread;
    volatile int      cancel = 0;      // polled by the library (RtuProgress::cancel)
    std::atomic<int>  result{1};       // 1 = running, 0 = ok, <0 = error
    std::atomic<int>  gather_kind{0};  // how the shards were collected: 1 one context, 2 asynchronous host copies, 3 RCCL
    std::string       error;
};

namespace {

void rows_to_image(void* user, const float* rows, int row0, int nrows) {
    rtu_image_from_rgbz(static_cast<RtuImage*>(user), rows, row0, nrows);  // gamma, Color24, z; bumps the rendered-pixel counter
}

void run_job(RtuRenderJob* job, const RtuSceneDesc* desc, RtuImage* img, std::vector<int> devices, int samples, int gather_bounces,
             std::string result_png, std::string zbuffer_png) {
    const int W = rtu_image_width(img), H = rtu_image_height(img);
    int rc = RTU_OK;
    RtuMultiContext* m = rtu_create_context_multi(devices.data(), (int)devices.size(), &rc);
    if (!m) {
        job->error = rtu_error_string(rc);
        job->result.store(rc != RTU_OK ? rc : RTU_ERR_HIP);
        return;
    }
    RtuFrameDesc frame;
    rc = rtu_multi_upload_scene(m, desc);
    if (rc == RTU_OK) rc = rtu_frame_setup(&desc->camera, W, H, &frame);
    if (rc == RTU_OK) {
        frame.samples = samples;
        frame.gather_bounces = gather_bounces;
        RtuProgress progress;
        progress.cancel = &job->cancel;
        progress.rows_done = rows_to_image;
        progress.user = img;
        rc = rtu_multi_render_frame(m, &frame, nullptr, &progress);
    }
    if (rc != RTU_OK) job->error = rtu_multi_last_error(m);
    job->gather_kind.store(rtu_multi_gather_kind(m));
    rtu_destroy_context_multi(m);
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
    if (job) job->cancel = 1;
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
