/*
 * rtu_render.h — the C-ABI of librtu_hip.so: the MI355X (gfx950) render path.
 *
 * This is the drop-in boundary behind the reference's BeginRender()
 * (main.cpp:66-68 -> SpawnRenderThreads main.cpp:29-64 -> Render
 * RenderFunctions.cpp:55-176). The host keeps the reference's scene graph /
 * RenderImage surface (see rtu_host.h) and calls, per frame:
 *
 *   rtu_create_context      once per GPU                    [no reference counterpart]
 *   rtu_upload_scene        <- the globals LoadScene fills  xmlload.cpp:64-131, main.cpp:17-27
 *   rtu_frame_setup         <- CalculateImageOrigin /       RenderFunctions.cpp:243-269
 *                              CalculateCurrentPoint (hoisted: origin,u,v)
 *   rtu_render_frame[_device] <- the per-pixel loop         RenderFunctions.cpp:62-174 with
 *                              PixelIterator::GetPixelLocation (PixelIterator.h:25-38),
 *                              Trace/ShadowTrace (RenderFunctions.cpp:181-240),
 *                              Object::IntersectRay (objFunctions.cpp:15-522),
 *                              MtlBlinn::Shade (mtlFunctions.cpp:120-298),
 *                              Light::Illuminate (lightFunctions.cpp:27-84, lights.h:32,48)
 *   rtu_destroy_context
 *
 * Output is linear float4 {r,g,b,z} per pixel (z = hInfo.z of the primary hit,
 * RTU_BIGFLOAT on a miss); gamma / Color24 / z-image stay on the host
 * (rtu_host.h) as in RenderFunctions.cpp:155-159 and scene.h:590-612.
 *
 * Plain pointers and sizes only; never throws; returns 0 or a negative RTU_ERR_*.
 * A context is single-caller. The caller owns host buffers; the library owns
 * its device buffers (scene, counters, recursion arena, default framebuffer).
 */
#ifndef RTU_RENDER_H_INCLUDED
#define RTU_RENDER_H_INCLUDED

#include "rtu_scene.h"

#ifdef __cplusplus
extern "C" {
#endif

#define RTU_OK               0
#define RTU_ERR_ARG         (-1)  /* NULL / out-of-range argument */
#define RTU_ERR_HIP         (-2)  /* a HIP runtime call failed (see rtu_last_error) */
#define RTU_ERR_UNSUPPORTED (-3)  /* scene exceeds a device-path limit */
#define RTU_ERR_STOCHASTIC  (-4)  /* soft shadows / glossy bounces / depth of field in a frame with samples == 0 */
#define RTU_ERR_NO_SCENE    (-5)  /* render before rtu_upload_scene */
#define RTU_ERR_NO_DEVICE   (-6)  /* no such GPU */
#define RTU_ERR_CAPACITY    (-7)  /* more Shade() frames than provisioned (see rtu_frame_status) */
#define RTU_ERR_CANCELLED   (-8)  /* the caller's cancel flag was raised (rtu_set_cancel_flag, RtuProgress): StopRender(), main.cpp:70-72 */

#define RTU_BAND_ROWS 8  /* image rows per band; one wavefront renders an 8x8 pixel tile */

typedef struct RtuContext RtuContext;

/* One frame. Bands of RTU_BAND_ROWS rows are dealt round-robin to shards:
 * band b belongs to shard (b % shard_count); a context renders only its shard
 * and writes it COMPACTLY (local row lr -> global row rtu_shard_global_row). */
typedef struct RtuFrameDesc {
    int32_t width, height;
    int32_t shard_rank, shard_count;  /* 0,1 for a single GPU */
    int32_t max_bounce;               /* 5, RenderFunctions.cpp:134 */
    int32_t collect_stats;            /* 1: fill the ray / traversal counters RtuStats (the counting variant: the reference's own
                                         tree, no culling — slower); 2: the FAST variant as it is timed, counting per kernel what it
                                         touches (RtuTouched) */
    int32_t coop_threshold;           /* tuning: a deferred-ray list shorter than this is traced by the
                                         cooperative (8 lanes per ray) kernels; 0 = default */
    int32_t samples;                  /* 0: recipe W, one ray through every pixel centre (scenes with stochastic
                                         features are refused, RTU_ERR_STOCHASTIC). S >= 1: recipe S, the sample loop
                                         of Render() (RenderFunctions.cpp:73-152) with S samples per pixel: Halton
                                         pixel offsets, depth of field, soft shadows, glossy bounces, on the sample
                                         streams described below; rgb = mean of the samples, z = mean hInfo.z of the
                                         samples that hit */
    float   cam_pos[3];               /* camera.pos */
    float   origin[3];                /* CalculateImageOrigin(camera.focaldist) */
    float   u[3], v[3];               /* per-pixel steps of CalculateCurrentPoint */
    float   lens_up[3];               /* camera.up (as given) and normalize(dir x up): the lens disk of */
    float   lens_right[3];            /* RenderFunctions.cpp:93 */
    float   dof;                      /* camera.dof */
    int32_t gather_bounces;           /* 0, or 4 with samples >= 1: recipe P (config 5) — recipe S plus the Monte-Carlo gather
                                         of Render() (RenderFunctions.cpp:129-135: MonteCarlo with monteCarloBounces = 4 and one
                                         cosine-weighted hemisphere sample per bounce, :549-590, :320-337). Keys: the hit of the
                                         gather ray has child_key(key, 3), the Shade() tree lit by MonteCarlo()'s AmbientLight
                                         child_key(key, 4); purposes 0x40000, 0x40001: sampleX, samplePhi */
} RtuFrameDesc;

/* Sample streams of recipe S. The reference draws from rand() (shared by its threads, seeded with the
 * time: RenderFunctions.cpp:60), which nothing can reproduce. Here every rand() call becomes
 *     rand31(key, purpose) = mix32(key ^ mix32(purpose * 0x9e3779b9 + 0x85ebca6b)) >> 1     in [0, RAND_MAX]
 *     mix32(x): x ^= x >> 16; x *= 0x7feb352d; x ^= x >> 15; x *= 0x846ca68b; x ^= x >> 16
 * inside the reference's own float expressions. key belongs to one Shade() call: the primary hit of sample
 * i of pixel p (p = x + width * y in the whole image, whatever the sharding) has
 *     sample_key(p, i) = mix32(mix32(p + 0x68bc21eb) ^ (i * 0x9e3779b9 + 1)),
 * the Shade() of the hit of its secondary ray `slot` (0 refracted or totally reflected, 1 Fresnel
 * reflection, 2 mirror reflection) has child_key(key, slot) = mix32(key + (slot + 1) * 0x632be5ab).
 * purpose: 0, 1 lens sample (sampleX, sampleTheta); 16 + 2 l, 17 + 2 l light l's disk sample (sampleR,
 * sampleTheta); 0x10000 / 0x20000 / 0x30000 + 3 a + {0,1,2}: attempt a of SampleSphere for the first /
 * second refraction normal and for the reflection normal. sin, cos (and, recipe P, acos) of the sampled
 * angles are evaluated in binary64 with IEEE operations only and rounded to float (portable_sincos /
 * portable_acos in raytracer-utah_amd/csrc/rtu_intersect.h state the sequences), within one ulp of libm's. */

/* Ray and traversal counters of one frame (all shards of one context). Same
 * fields as RtuOracleStats so CPU and GPU can be compared exactly. */
typedef struct RtuStats {
    uint64_t primary_rays, primary_hits;
    uint64_t secondary_rays;  /* root-level Trace calls issued by Shade */
    uint64_t shadow_rays;     /* root-level ShadowTrace calls issued by Shadow */
    uint64_t node_tests;      /* ray x object-node intersection calls */
    uint64_t mesh_entries;    /* rays that passed a mesh's bounding box */
    uint64_t inner_visits, leaf_visits, leaf_elems;
    uint64_t tri_tests, tri_accepts;
} RtuStats;

int         rtu_device_count(void);
const char* rtu_error_string(int err);

/* What the machine says about GPU `device_id` (hipGetDeviceProperties): bench.py derives the HBM peak of its roofline from the
 * memory clock and bus width reported here ("confirm on the machine, do not hard-code") and falls back to the
 * constant of the microarchitecture guide when the figures are implausible. Needs no context. */
typedef struct RtuDeviceInfo {
    int32_t  compute_units, clock_khz, memory_clock_khz, memory_bus_bits;
    uint64_t l2_bytes, hbm_bytes;
    char     name[64], arch[64];
} RtuDeviceInfo;
int         rtu_device_info(int device_id, RtuDeviceInfo* out);

RtuContext* rtu_create_context(int device_id, int* err_out);
void        rtu_destroy_context(RtuContext* ctx);
const char* rtu_last_error(const RtuContext* ctx);

/* Validate, flatten into the device layout and copy to HBM. May be called again
 * to replace the scene. */
int  rtu_upload_scene(RtuContext* ctx, const RtuSceneDesc* scene);

/* The validation rtu_upload_scene runs first, on its own: every index the kernels follow (node parents, mesh,
 * material and texture ids, vertex / normal / texture-vertex indices, BVH children, leaf ranges, element ids) is
 * range-checked so that a malformed scene is an error code, never an out-of-bounds access on the GPU. Pure host
 * code, needs no GPU and no context; the message goes to err_buf (may be NULL). */
int  rtu_validate_scene(const RtuSceneDesc* scene, char* err_buf, size_t err_len);

/* Fill width/height, cam_pos/origin/u/v from the camera (fp64 tan chain of
 * RenderFunctions.cpp:247 evaluated on the host, once per frame), single shard,
 * max_bounce 5, no stats. Pure host arithmetic; needs no GPU. */
int  rtu_frame_setup(const RtuCamera* camera, int width, int height, RtuFrameDesc* frame_out);

/* Shard geometry helpers (pure arithmetic). */
int  rtu_shard_rows(const RtuFrameDesc* frame);                    /* rows this shard renders */
int  rtu_shard_max_rows(int height, int shard_count);             /* max over ranks (gather padding) */
int  rtu_shard_global_row(const RtuFrameDesc* frame, int local_row);

/* Render this context's shard into DEVICE memory d_rgbz (rtu_shard_rows * width
 * float4, 16-byte aligned), asynchronously on hip_stream (a hipStream_t passed
 * as void*; NULL = the device's default stream, as in HIP itself). Inputs are already resident in
 * HBM; nothing is copied.
 * ONE STREAM PER CONTEXT between two rtu_frame_status calls: the context's frame records, append counters, camera table,
 * coverage masks and tile-occupancy words are shared by every launch sequence queued on it, and only stream order keeps
 * one sequence's kernels from another's. A caller that wants two sequences in flight on two streams uses two contexts
 * (bench.py and rtu_multi_render_frame do); switching streams after rtu_frame_status (which waits for the device) is fine. */
int  rtu_render_frame_device(RtuContext* ctx, const RtuFrameDesc* frame, void* d_rgbz, void* hip_stream);

/* Frames in flight: n_frames (<= RTU_MAX_FRAMES_IN_FLIGHT, and at most 2^26 pixels together) frames of recipe W of the uploaded scene — the same resolution, shard and
 * options, each with its own camera (cam_pos / origin / u / v) — rendered by ONE launch sequence into
 * d_rgbz = n_frames consecutive shard images (frame i at float4 offset i * rtu_shard_rows * width).
 * One 1080p frame at one sample per pixel is too little work to fill 256 CUs: sixteen in
 * flight render at 2.4 times the rays per second. Every frame is the image rtu_render_frame_device
 * gives for it, bit for bit. Asynchronous; rtu_frame_status afterwards as for a single frame. */
#define RTU_MAX_FRAMES_IN_FLIGHT 128
int  rtu_render_frames_device(RtuContext* ctx, const RtuFrameDesc* frames, int n_frames, void* d_rgbz, void* hip_stream);

/* The content of the reference's RenderImage from a rendered float4 image, on the device: d_z[i] = z,
 * d_rgb8[3 i ..] = Color24(pow(c, 1/2.2)) (RenderFunctions.cpp:152-160; binary64 pow, cyColor.h:226 clamp) —
 * 7 bytes per pixel instead of 16, which is what a multi-GPU gather should move. Asynchronous on hip_stream.
 * (The host path, rtu_image_from_rgbz, does the same with glibc's pow; the two agree except where the device
 * library's pow rounds a value sitting on a byte boundary the other way: within the +-1 level bar.) */
int  rtu_pack_image_device(RtuContext* ctx, const void* d_rgbz, size_t n_pixels, void* d_z, void* d_rgb8, void* hip_stream);

/* The two OUTPUT images of a batch of rendered frames, 4 bytes per pixel {r, g, b of Color24, z-image byte}: what Result.png and
 * ZBuffer.png are written from (main.cpp:59-61), and what a multi-GPU gather has to move when the frame rate makes the link to
 * the root the bottleneck (10 000 frames per second x 14.5 MB of float z + Color24 exceed one xGMI link). The z-image needs the
 * FRAME-wide zmin / zmax (ComputeZBufferImage, scene.h:590-612):
 *   rtu_minmax_z_device   d_minmax[2 i], [2 i + 1] (int64 each) = order-preserving keys of zmin and of zmax of frame i over this
 *                         shard's pixels (frame i at float4 offset i * pixels_per_frame), encoded so that the element-wise MINIMUM of
 *                         the arrays of all shards is the frame's (one all-reduce MIN of 2 n_frames int64; a single GPU skips it);
 *   rtu_pack_output_device quantises with those: byte = int((zmax - z) / (zmax - zmin) * 255) clamped, 0 for a miss, in binary32 with
 *                         a correctly rounded division, exactly as the reference; colours as rtu_pack_image_device.
 * Both asynchronous on hip_stream. */
int  rtu_minmax_z_device(RtuContext* ctx, const void* d_rgbz, size_t pixels_per_frame, int n_frames, void* d_minmax, void* hip_stream);
int  rtu_pack_output_device(RtuContext* ctx, const void* d_rgbz, size_t pixels_per_frame, int n_frames, const void* d_minmax, void* d_out4,
                            void* hip_stream);

/* Render into the context's own framebuffer and copy the shard to host memory
 * h_rgbz (rtu_shard_rows * width * 4 floats). Synchronous. stats may be NULL. */
int  rtu_render_frame(RtuContext* ctx, const RtuFrameDesc* frame, float* h_rgbz, RtuStats* stats);

/* After rtu_render_frame_device: wait for the device and report whether the frame is
 * complete. The Shade() recursion is evaluated level by level in pre-sized frame
 * arrays (one frame per pixel per level to begin with); RTU_ERR_CAPACITY means a level
 * overflowed: the arrays are re-provisioned from the counts the frame reported — render the
 * frame again (at most one round per recursion level). rtu_render_frame does the check and
 * the re-render itself. The report is STICKY: it covers every launch sequence queued on this context since the
 * previous rtu_frame_status (or synchronous render), not only the last one — a caller that pipelines several
 * rtu_render_frame(s)_device calls with different cameras and checks once learns that SOME frame of them is
 * incomplete and renders them again. */
int  rtu_frame_status(RtuContext* ctx);

/* Cancellation (StopRender(), main.cpp:70-72): a word the caller may set non-zero at any time; the context reads it between the
 * launch sequences of a sampled frame (recipes S / P: one sequence per batch of samples — a 64-sample 1080p frame is hundreds
 * of milliseconds) and returns RTU_ERR_CANCELLED from the render call. NULL: none. A single launch sequence (a frame of
 * recipe W, a batch of frames in flight) is a fraction of a millisecond and is never interrupted. */
int  rtu_set_cancel_flag(RtuContext* ctx, const volatile int* flag);

/* ---- Several GPUs behind one handle (raytracer-utah_amd/csrc/rtu_multi.hip) ----------------------------------------------
 * What SpawnRenderThreads does with CPU workers (main.cpp:29-64): fan one frame out, wait, hand back one image. The frame is
 * sharded by interleaved RTU_BAND_ROWS-row bands (band b -> the b mod n-th device), the scene replicated on every GPU; the
 * float4 shards are gathered to the first GPU with one grouped RCCL send / receive over xGMI (n distinct GPUs, librccl.so
 * found) or by concurrent asynchronous copies into pinned host memory (several contexts on one GPU, or no RCCL), and land,
 * de-interleaved, in h_rgbz = height * width float4 in image order. device_ids may repeat (n contexts on one GPU: how the
 * one-GPU test box exercises this path). The RCCL branch with n > 1 distinct GPUs has not run on hardware.
 *
 * progress (may be NULL): `cancel` as rtu_set_cancel_flag (polled between sample batches on every GPU, between capacity
 * rounds and between the shards as they are handed over); rows_done(user, rows, row0, nrows) is called once per band as
 * the shards arrive, from the calling thread, with `rows` = nrows * width float4 of image rows [row0, row0 + nrows) — what
 * RenderImage::IncrementNumRenderPixel (scene.h:585-588) counts and the viewport polls (viewport.cpp:390-410). h_rgbz may be
 * NULL when rows_done consumes the rows. frame->shard_rank / shard_count are ignored. Synchronous; 0 or a negative RTU_ERR_*
 * (rtu_multi_last_error). */
typedef struct RtuMultiContext RtuMultiContext;
typedef struct RtuProgress {
    const volatile int* cancel;
    void (*rows_done)(void* user, const float* rows, int row0, int nrows);
    void* user;
} RtuProgress;
RtuMultiContext* rtu_create_context_multi(const int* device_ids, int n_devices, int* err_out);
void        rtu_destroy_context_multi(RtuMultiContext* m);
int         rtu_multi_size(const RtuMultiContext* m);
RtuContext* rtu_multi_context(RtuMultiContext* m, int i);            /* the i-th GPU's context (diagnostics, rtu_debug_*) */
const char* rtu_multi_last_error(const RtuMultiContext* m);
int         rtu_multi_upload_scene(RtuMultiContext* m, const RtuSceneDesc* scene);
int         rtu_multi_render_frame(RtuMultiContext* m, const RtuFrameDesc* frame, float* h_rgbz, const RtuProgress* progress);
/* How the shards of the last rtu_multi_render_frame reached the host: 1 one context; 2 several contexts, asynchronous copies into
 * one pinned buffer, all in flight together; 3 RCCL (grouped ncclSend / ncclRecv to the first GPU, then one copy). */
int         rtu_multi_gather_kind(const RtuMultiContext* m);

/* Diagnostic: render one frame with every wavefront stamping the GPU's constant clock on entry
 * and exit, and return, per kernel launch that ran, its slot (0 primary, 1/2 primary stage 2
 * cooperative/wide, 3+4L.. trace, stage 2 cooperative, stage 2 wide, consume of level L,
 * 27+L combine of level L) and the first-entry / last-exit times in microseconds from the first
 * stamp. Unlike a profiler trace this neither serialises nor pads the launches. Returns the
 * number of entries (<= max_entries) or a negative RTU_ERR_*. Synchronous. */
int  rtu_render_timeline(RtuContext* ctx, const RtuFrameDesc* frame, void* d_rgbz, int max_entries, int* slot_out,
                         double* start_us_out, double* end_us_out);

/* Diagnostic, after rtu_render_timeline: the exit times (microseconds after the kernel's first entry) of
 * the wavefronts of the launch in timeline slot `slot` — one value per wavefront slot (index modulo
 * 8192; a later wavefront overwrites an earlier one). Shows whether a launch is a plateau or a tail.
 * Returns the number of values written. */
int  rtu_timeline_exits(RtuContext* ctx, int slot, int max_values, double* exit_us_out);

/* Test hook for the tail kernel. Recursion levels that were almost empty in the previous frame of a
 * scene are not launched kernel by kernel in the next one: one kernel evaluates every frame of the
 * first such level, subtree and all, with one wavefront per frame. Which level that is
 * comes from the counts of the previous launch of the same shape and is only a hint — any value renders the same image; if the
 * level turns out to hold thousands of frames (the view changed) the kernel refuses it on the device, the frame is reported
 * incomplete like a capacity overflow (rtu_frame_status) and rendered again level by level.
 * This sets the cut level for the NEXT launch only (taken as it is, never refused): 1..5, or 6 for "no tail". */
int  rtu_debug_tail_from(RtuContext* ctx, int level);

/* Test hook: switch the node-level bounds of the fast variant off (0) or on (1, the default after an upload) until the next
 * upload: the world-space box per scene node and, for primary rays, its screen rectangle per camera, by which a ray skips
 * nodes it cannot touch before their transformation and exact test. Results must not change. */
int  rtu_debug_node_bounds(RtuContext* ctx, int on);

/* Experiment switches for performance work (which part of a kernel costs what): bits are defined next to their use in
 * render_impl.h; bits 0..7 render WRONG images: never set them in production paths or tests of results. Five bits only switch an
 * optimisation off and leave every result bit alone (tests compare the images with and without): 256 = no tile occupancy
 * (k_primary tests every tile against the node rectangles and coverage masks itself), 512 = no stage-2 grid hints (both
 * stage-2 kernels of every tracing phase are launched at full size), 64 = the occluder lists of shadow rays only say "empty cell or
 * not" (every listed ray walks the BVH), 2048 = no Shade() call is settled without a frame record (every hit becomes a frame),
 * 8192 = no side mode (stage 2 of the primary phase in the launch stream, before the recursion levels, instead of beside them),
 * 16384 = both stage-2 kernels of every phase are launched whatever the last launch's list lengths said (outside side mode).
 * 131072 (a wrong image; frames with collect_stats == 2 only) = stage 2 of the primary phase writes the work of each ray's BVH walk — two
 * units per inner step, one per triangle test — over the pixel's red channel (tools/scratch/walk_units.py). */
int  rtu_debug_flags(RtuContext* ctx, uint32_t bits);
/* A hint, never needed for correctness: how many launch sequences the caller keeps in flight on this GPU at once, over all of its
 * contexts together (bench.py alternates its batches over two contexts: 2). A context then sizes the grid of its long-running
 * primary kernel for its share of the machine instead of all of it (measured, two sequences in flight: 59.4 -> 62.5 Grays/s).
 * Default 1. Any value renders the same images. */
int  rtu_set_sequences_in_flight(RtuContext* ctx, int n);

/* Test hook: let the walks of the fast trees use at most `entries` stack entries from the next frame on
 * (until the next upload), so that tests can exercise the overflow path — a ray whose walk would
 * need more is finished on the reference's tree — on any scene. Results must not change. */
int  rtu_debug_walk_stack_limit(RtuContext* ctx, uint32_t entries);

/* Diagnostic: the acceleration structures built at upload for mesh `mesh`: out5 = {faces, depth of the
 * binned-SAH tree, deepest stack a walk of its 4-wide form can build (walks are given
 * min(that, RTU_MAX_BVH_STACK) entries; a ray that needs more finishes on the reference's tree),
 * 4-wide nodes, 8-wide nodes}. */
int  rtu_mesh_info(const RtuContext* ctx, uint32_t mesh, uint32_t* out5);

/* Diagnostic: the occluder lists of shadow rays built at upload (rtu_device.h DevLightMask), one per (non-ambient light < 4,
 * masked mesh node) pair that got one, in build order: out5 = {light slot, masked mesh slot, grid size G, entries of all
 * cells together, entries of the longest cell}. RTU_ERR_ARG past the last one. */
int  rtu_light_list_info(const RtuContext* ctx, uint32_t index, uint32_t* out5);

/* Test hook, pure host code (no GPU, no context): the occluder list rtu_upload_scene would build for the light_slot-th non-ambient
 * light and the cover_slot-th mesh node of the scene — frame, grid and, per cell, its entries as FACES of the node's mesh with the
 * depth in front of which an origin cannot see them (ascending per cell) — so that a CPU test can check ray by ray that every
 * triangle a shadow ray hits is listed in the cell of the ray's origin. cell = int((v - v0) * sv) * G + int((u - u0) * su) with
 * (u, v) = ((p - L) . X, (p - L) . Y) [/ ((p - L) . Z) for a point light]. usable == 0: no list from there (arrays NULL). */
typedef struct RtuLightListDump {
    int32_t  usable, node, light;
    uint32_t G, point, n_entries;
    float    X[3], Y[3], Z[3], L[3], u0, v0, su, sv;
    uint32_t* cell_off;     /* [G * G + 1] */
    uint32_t* entry_face;   /* [n_entries] */
    float*    entry_zmin;   /* [n_entries] */
} RtuLightListDump;
int  rtu_debug_light_list(const RtuSceneDesc* scene, uint32_t light_slot, uint32_t cover_slot, RtuLightListDump* out);
void rtu_debug_light_list_free(RtuLightListDump* dump);

/* Diagnostic: Shade() frames per recursion level (6 values) and rays deferred to stage 2 per phase
 * (7 values: primary, then levels 0..5) of the most recent frame (fast variant). Synchronises. */
int  rtu_frame_counts(RtuContext* ctx, uint32_t* frames_out, uint32_t* deferred_out);

/* Counters of the last frame rendered with collect_stats=1 (synchronises). */
int  rtu_get_stats(RtuContext* ctx, RtuStats* stats);

/* Touched-bytes mode (collect_stats == 2): what the kernels of the FAST variant — the ones bench.py times — read and write,
 * per kernel launch of the most recent launch sequence. A slot is a kernel's place in the sequence (rtu_kernel_slot_name:
 * "k_primary", "k_primary2c", "k_primary2", "k_trace(L0)", "k_trace2c(L0)", "k_trace2(L0)", "k_consume(L0)", ..., "k_combine(L0)", ...;
 * the tail kernel reports in the k_trace slot of its cut level). The images of this mode are those of the fast variant, bit for bit.
 * ALGORITHMIC bytes of a launch (rtu_touched_bytes; cache-agnostic, every access counted where it is made):
 *   24 bound_tests (a node's world-space box) + 48 node_tests (itm + pos) + 24 mesh_box_tests (bounding box) + 84 xform_levels (tm + pos + itm of FromNodeCoords)
 *   + 112 inner4 (7 float4 of a 4-wide node) + 256 inner8 (8 x 32 B child records) + 64 inner_ref (a sibling pair of the reference's
 *   tree: exact-tie / stack-overflow fallback) + 64 tri_tests (triangle record) + 100 winners (element, face and normal indices, three
 *   vertices, three normals; 148 with texture vertices) + record_bytes (frame records, lists, shadow results, pixels: counted at
 *   every load / store). SURVEY 8d's per-unit figures with the record sizes of THIS layout in place of the reference's.
 * Wave-uniform data — scene nodes and their bounds, mesh headers, screen rectangles — is read through the constant address space
 * by scalar loads, ONCE PER WAVEFRONT whatever the number of lanes that need it: bound_tests, node_tests and mesh_box_tests count
 * wavefronts, not lanes (the reference reads them once per ray; a GPU lane does not). Everything else is per lane. */
#define RTU_KERNEL_SLOTS 40
typedef struct RtuTouched {
    uint64_t rays;            /* Trace / ShadowTrace walks started by this launch */
    uint64_t node_tests, mesh_box_tests, inner4, inner8, inner_ref, tri_tests, winners, xform_levels;
    uint64_t record_bytes;
    uint64_t bound_tests;     /* node-level bounds tested (24 B each: the node's world-space box) */
    uint64_t inline_shadow_rays; /* of `rays`: shadow rays of childless Shade() calls fired by the lane that found the hit (no frame record) */
} RtuTouched;
int         rtu_get_touched(RtuContext* ctx, RtuTouched* per_slot, int n_slots);   /* synchronises; returns the slots written */
/* A sampled frame (recipes S / P) is many launch sequences — one per batch of samples, ten per batch for recipe P — and its counters are
 * the sums over all of them: per_slot[i] = the number of launches of slot i's kernel that went into the table since it was zeroed
 * (bytes per launch = rtu_touched_bytes / launches). Slot 33 is recipe P's k_gi_roots, slot 34 the k_tail launch of side mode (the few
 * frames stage 2 of the primary phase makes, evaluated on the helper stream beside the recursion levels). */
int         rtu_get_touched_launches(RtuContext* ctx, uint32_t* per_slot, int n_slots);
unsigned long long rtu_touched_bytes(const RtuTouched* t, int textured);
const char* rtu_kernel_slot_name(int slot);

/* Measurement helper for bench.py: bracket every launch of the kernel in `slot` with HIP events on the launch stream, from now on
 * (slot < 0: stop; what was measured stays until it is read). rtu_probe_read synchronises, returns the summed duration and the number of launches measured since the last
 * read (at most 64 are kept) and starts over. Recipe W launch sequences only. */
int  rtu_probe_kernel(RtuContext* ctx, int slot);
int  rtu_probe_read(RtuContext* ctx, float* total_ms_out, int* launches_out);

/* Measurement helper for bench.py: launch the render kernel `iters` times
 * back-to-back on `hip_stream`, bracketed by HIP events recorded on that same
 * stream, and return the AVERAGE kernel duration in milliseconds. */
int  rtu_time_render(RtuContext* ctx, const RtuFrameDesc* frame, void* d_rgbz, void* hip_stream,
                     int iters, float* avg_ms_out);

/* Self-test: the kernels replace binary32 divisions by a per-ray constant with an exact
 * binary64-reciprocal form (rtu_intersect.h); this runs n_pairs pseudo-random operand
 * pairs through both forms on the GPU and returns how many quotients differ in any bit
 * (must be 0). */
int  rtu_selftest_division(RtuContext* ctx, unsigned long long n_pairs, unsigned long long seed,
                           unsigned long long* mismatches_out);

/* Self-test of the sphere / plane intersection routines: the device path evaluates the
 * reference's expressions in a cheaper order (the bounding-box test last, and only when its
 * outcome is not already implied); this runs both orders on n_rays random and adversarial rays
 * (grazing, far away, axis-parallel, origin on the surface) and counts results that differ in
 * any bit. Expected: 0. */
int  rtu_selftest_primitives(RtuContext* ctx, unsigned long long n_rays, unsigned long long seed,
                             unsigned long long* mismatches_out);

/* Device memory helpers so a C/C++ host needs no HIP headers. */
/* The context's own stream (a hipStream_t as void*) and device: a multi-GPU host (host/begin_render.cpp) renders every shard on
 * its context's stream and queues the collection — RCCL send / receive or an asynchronous copy into pinned host memory — behind
 * it on the same stream, so that the shards of all GPUs travel at the same time. */
void* rtu_context_stream(RtuContext* ctx);
int   rtu_context_device(const RtuContext* ctx);
int   rtu_context_sync(RtuContext* ctx);                       /* wait for the context's stream */
void* rtu_host_alloc_pinned(size_t bytes);                     /* page-locked host memory (asynchronous copies need it) */
void  rtu_host_free_pinned(void* p);
int   rtu_copy_to_host_async(RtuContext* ctx, void* h_dst, const void* d_src, size_t bytes, void* hip_stream);
void* rtu_device_alloc(RtuContext* ctx, size_t bytes);
void  rtu_device_free(RtuContext* ctx, void* d_ptr);
int   rtu_copy_to_host(RtuContext* ctx, void* h_dst, const void* d_src, size_t bytes);

#ifdef __cplusplus
}
#endif
#endif /* RTU_RENDER_H_INCLUDED */
