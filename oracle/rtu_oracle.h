/*
 * rtu_oracle.h — C interface of the CPU restatement (oracle/rtu_oracle.cpp).
 * TEST INFRASTRUCTURE: see the header of rtu_oracle.cpp. Loaded only by tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg.
 */
#ifndef RTU_ORACLE_H_INCLUDED
#define RTU_ORACLE_H_INCLUDED

#include "rtu_scene.h"

#ifdef __cplusplus
extern "C" {
#endif

#define RTU_ORACLE_ERR_ARG         (-1)
#define RTU_ORACLE_ERR_STOCHASTIC  (-2) /* soft shadows / glossy / dof: reference is non-deterministic */
#define RTU_ORACLE_ERR_UNSUPPORTED (-3) /* outside the restated scope */

/* Same fields, same meaning as RtuStats in rtu_render.h, so CPU and GPU ray and
 * traversal counters can be compared exactly. */
typedef struct RtuOracleStats {
    uint64_t primary_rays, primary_hits;
    uint64_t secondary_rays;  /* root-level Trace calls issued by Shade */
    uint64_t shadow_rays;     /* root-level ShadowTrace calls issued by Shadow */
    uint64_t node_tests;      /* ray x object-node intersection calls */
    uint64_t mesh_entries;    /* rays that passed a mesh's bounding box */
    uint64_t inner_visits, leaf_visits, leaf_elems;
    uint64_t tri_tests, tri_accepts;
} RtuOracleStats;

/* Recipe W over rows [row0,row0+nrows): rgbz_out holds nrows*width float4
 * {linear r,g,b, z}; z = hInfo.z of the primary hit (RTU_BIGFLOAT on a miss). */
int  rtu_oracle_render_rows(const RtuSceneDesc* scene, int width, int height, int row0, int nrows,
                            float* rgbz_out, RtuOracleStats* stats, int threads);
int  rtu_oracle_render(const RtuSceneDesc* scene, int width, int height, float* rgbz_out,
                       RtuOracleStats* stats, int threads);
/* The whole frame with the work distribution chosen: 0 = chunks of four rows per fetch (what the other entry points do),
 * 1 = the reference's PixelIterator (PixelIterator.h:25-38): one shared atomic counter, one pixel per fetch. Same image. */
int  rtu_oracle_render_scheduled(const RtuSceneDesc* scene, int width, int height, float* rgbz_out,
                                 RtuOracleStats* stats, int threads, int per_pixel_schedule);
/* Recipe S (row f1): spp samples per pixel as in the sample loop of Render() (RenderFunctions.cpp:73-152:
 * Halton pixel offsets, depth of field, soft shadows, glossy bounces), direct lighting only; rgb = mean
 * of the samples, z = mean hInfo.z of the samples that hit. stream: where the integers that replace
 * rand() come from (see rtu_oracle.cpp "Sample streams"); trig: sinf/cosf of libm (as the reference
 * calls them) or the portable evaluation the device uses. */
#define RTU_ORACLE_STREAM_KEYED      0
#define RTU_ORACLE_STREAM_SEQUENTIAL 1
#define RTU_ORACLE_TRIG_PORTABLE     0
#define RTU_ORACLE_TRIG_LIBM         1
int  rtu_oracle_render_samples(const RtuSceneDesc* scene, int width, int height, int row0, int nrows, int spp,
                               int stream, int trig, float* rgbz_out, RtuOracleStats* stats, int threads);
/* Recipe P (config 5): recipe S plus the Monte-Carlo gather of Render() (RenderFunctions.cpp:129-135: MonteCarlo
 * with 4 bounces and 1 sample, :549-590; cosine-weighted hemisphere sampling, :320-337). */
int  rtu_oracle_render_paths(const RtuSceneDesc* scene, int width, int height, int row0, int nrows, int spp,
                             int stream, int trig, float* rgbz_out, RtuOracleStats* stats, int threads);
/* Test hook: 1 = test every triangle of a mesh whatever its boxes say (NOT the reference's algorithm; see rtu_oracle.cpp). */
void rtu_oracle_debug_all_triangles(int on);
void rtu_oracle_portable_sincos(const float* t, int n, float* sin_out, float* cos_out);
void rtu_oracle_portable_acos(const float* x, int n, float* out);
uint32_t rtu_oracle_rand31(uint32_t key, uint32_t idx);
uint32_t rtu_oracle_sample_key(uint32_t pixel, uint32_t sample);
uint32_t rtu_oracle_child_key(uint32_t key, uint32_t slot);
/* pos, origin, u, v of the image plane (RenderFunctions.cpp:243-269). */
int  rtu_oracle_camera_frame(const RtuCamera* cam, int width, int height, float out12[12]);
/* gamma + Color24 + z-image; any output pointer may be NULL. */
void rtu_oracle_postprocess(const float* rgbz, int width, int height, unsigned char* rgb_out,
                            float* z_out, unsigned char* zimg_out);

#ifdef __cplusplus
}
#endif
#endif
