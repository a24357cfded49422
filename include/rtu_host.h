/*
 * rtu_host.h — C entry points of librtu_host.so, the host side of the drop-in
 * boundary (pure C++/g++, no GPU code). It is the input and output side of the
 * render path:
 *
 *   scene files  --rtu_scene_load_xml-->  RtuScene (owned, flattened)
 *   RtuScene     --rtu_scene_desc------>  RtuSceneDesc  --> rtu_upload_scene (rtu_render.h)
 *   float4 rgbz  --rtu_image_*---------->  Color24 / z-image / Result.png / ZBuffer.png
 *
 * Reference interfaces replaced (paths relative to the reference tree):
 *   rtu_scene_load_xml      <- int LoadScene(const char*)        ExternalLibrary/xmlload.cpp:64-131
 *                              TriObj::Load                      ExternalLibrary/objects.h:52-60
 *   rtu_image_from_rgbz     <- gamma + Color24 store             RenderFunctions.cpp:152-160
 *   rtu_image_compute_zimg  <- RenderImage::ComputeZBufferImage  ExternalLibrary/scene.h:590-612
 *   rtu_image_save_png      <- RenderImage::SaveImage/SaveZImage ExternalLibrary/scene.h:633-654
 *   rtu_begin_render / rtu_stop_render / rtu_render_wait
 *                           <- BeginRender()/StopRender()        main.cpp:66-72, viewport.cpp:36-37
 */
#ifndef RTU_HOST_H_INCLUDED
#define RTU_HOST_H_INCLUDED

#include "rtu_scene.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- owned scenes ------------------------------------------------------- */
typedef struct RtuScene RtuScene; /* opaque; owns every array its desc points to */

/* Load a scene XML (+ the OBJ files it names). Every occurrence of the prefix
 * `remap_from` at the start of a file name inside the XML is replaced by
 * `remap_to` (the reference's scenes carry the author's absolute paths);
 * either may be NULL. Returns NULL on failure (message via rtu_host_last_error),
 * mirroring LoadScene()'s 0 return. */
RtuScene* rtu_scene_load_xml(const char* xml_path, const char* remap_from, const char* remap_to);

/* Deep-copy an existing description (e.g. one assembled by a caller). */
RtuScene* rtu_scene_clone(const RtuSceneDesc* desc);

/* Blob round trip (little endian, position independent) — the format of the
 * fixtures in tests/golden/. */
RtuScene* rtu_scene_load_blob(const void* blob, size_t size);
RtuScene* rtu_scene_load_blob_file(const char* path);
void*     rtu_scene_to_blob(const RtuSceneDesc* desc, size_t* size_out); /* malloc'ed */
int       rtu_scene_save_blob_file(const RtuSceneDesc* desc, const char* path);
void      rtu_blob_free(void* blob);

const RtuSceneDesc* rtu_scene_desc(const RtuScene* scene);
/* Override the render resolution after loading (set camera.imgWidth /
 * imgHeight after LoadScene, then re-Init the RenderImage). */
void      rtu_scene_set_resolution(RtuScene* scene, int width, int height);
void      rtu_scene_free(RtuScene* scene);

const char* rtu_host_last_error(void);

/* ---- output side: RenderImage mirror ------------------------------------ */
typedef struct RtuImage RtuImage; /* Color24 img[], float zbuffer[], uchar zimg[] */

RtuImage* rtu_image_create(int width, int height);            /* RenderImage::Init, scene.h:552-572 */
void      rtu_image_free(RtuImage* img);
int       rtu_image_width(const RtuImage* img);
int       rtu_image_height(const RtuImage* img);
uint8_t*  rtu_image_pixels(RtuImage* img);                    /* W*H*3, GetPixels() */
float*    rtu_image_zbuffer(RtuImage* img);                   /* W*H,   GetZBuffer() */
uint8_t*  rtu_image_zimage(RtuImage* img);                    /* W*H or NULL before compute */
int       rtu_image_num_rendered(const RtuImage* img);        /* GetNumRenderedPixels() */
int       rtu_image_is_done(const RtuImage* img);             /* IsRenderDone() */

/* Fill rows [row0,row0+nrows) from linear float4 {r,g,b,z}: gamma
 * pow(double(c),1/2.2) -> float, Color24 truncation, z copy; bumps the rendered
 * pixel counter by nrows*W. */
void      rtu_image_from_rgbz(RtuImage* img, const float* rgbz, int row0, int nrows);
void      rtu_image_compute_zimg(RtuImage* img);
int       rtu_image_save_png(const RtuImage* img, const char* path);   /* 8-bit RGB */
int       rtu_image_save_zpng(const RtuImage* img, const char* path);  /* 8-bit grey */
/* Generic 8-bit PNG writer (comp = 1 or 3). */
int       rtu_write_png(const char* path, const uint8_t* data, int width, int height, int comp);

/* ---- BeginRender()/StopRender() drop-in ---------------------------------- */
typedef struct RtuRenderJob RtuRenderJob;

/* Start rendering `scene` into `img` on the given GPUs (device ordinals) and
 * return immediately; a single host thread drives the C-ABI in rtu_render.h,
 * then writes result_png / zbuffer_png (either may be NULL to skip), exactly
 * the sequence of main.cpp:29-64. */
RtuRenderJob* rtu_begin_render(const RtuScene* scene, RtuImage* img,
                               const int* device_ids, int n_devices,
                               const char* result_png, const char* zbuffer_png);
/* The same with `samples` per pixel (RtuFrameDesc.samples in rtu_render.h): 0 is rtu_begin_render; S >= 1
 * renders recipe S, which scenes with soft shadows, glossy bounces or depth of field need (the reference's
 * Render() hard-codes 1024 samples, RenderFunctions.cpp:27). */
RtuRenderJob* rtu_begin_render_sampled(const RtuScene* scene, RtuImage* img,
                                       const int* device_ids, int n_devices, int samples,
                                       const char* result_png, const char* zbuffer_png);
/* HEAD's path-traced mode (config 5): the same plus the 4-bounce Monte-Carlo gather (RtuFrameDesc.gather_bounces = 4). */
RtuRenderJob* rtu_begin_render_paths(const RtuScene* scene, RtuImage* img,
                                     const int* device_ids, int n_devices, int samples,
                                     const char* result_png, const char* zbuffer_png);
void      rtu_stop_render(RtuRenderJob* job);      /* cooperative cancel between bands */
int       rtu_render_wait(RtuRenderJob* job);      /* join; 0 or negative error code */
/* After the job (joins): how the shards reached the host — 1 one context; 2 several contexts, asynchronous copies into one
 * pinned buffer, all in flight together; 3 RCCL (grouped ncclSend / ncclRecv to the root GPU, then one copy). */
int       rtu_render_gather_kind(RtuRenderJob* job);
void      rtu_render_job_free(RtuRenderJob* job);

#ifdef __cplusplus
}
#endif
#endif /* RTU_HOST_H_INCLUDED */
