// rtu_device.h — HBM layout of an uploaded scene and the kernel launch interface.
//
// Layout choices (see DESIGN.md "Data layout in HBM"):
//  * scene-graph nodes, materials, lights: tiny, wave-uniform -> read through the
//    scalar cache (s_load), one struct per item, as on the host (rtu_scene.h);
//  * BVH nodes: 32 B, sibling pairs 64-B aligned -> an inner visit is one
//    64-byte line (4 x dwordx4);
//  * triangles: gathered ONCE at upload into leaf order, 64 B per triangle: the vertex
//    A, the normalised face normal the reference recomputes per test
//    (objFunctions.cpp:263), the 2-D projected edges and 1/TriABCArea — everything
//    of IntersectTriangle that does not depend on the ray, evaluated once with the
//    reference's own float operations; a test is four aligned 16-byte loads;
//  * per-vertex normals / texcoords stay indexed (touched only on an accepted hit).
#ifndef RTU_DEVICE_H_INCLUDED
#define RTU_DEVICE_H_INCLUDED

#include "rtu_render.h"
#include "rtu_vec.h"

// Wave-uniform scene data (nodes, lights, materials, mesh headers) is read through
// the CONSTANT address space so the compiler emits scalar loads (s_load) into SGPRs
// instead of 64 identical vector loads.
#define RTU_CONST __attribute__((address_space(4)))
template <class T> __device__ __forceinline__ const RTU_CONST T* as_const(const T* p) {
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wold-style-cast"
    return (const RTU_CONST T*)(p);
#pragma clang diagnostic pop
}

// Per-lane gathers from the BVH / triangle arrays go through the GLOBAL address space with a wave-uniform base in scalar
// registers and a 32-bit byte offset per lane. The pointers come out of a mesh header in memory, so the compiler cannot know
// where they point and emits FLAT loads for plain dereferences: a 64-bit address per lane and load, one counter shared with
// the LDS stack (every pop waits for the loads in flight and vice versa) — measured in the ISA of the round-2 walk.
#define RTU_GLOBAL __attribute__((address_space(1)))
typedef float rtu_v4f __attribute__((ext_vector_type(4)));
struct GBase {
    const RTU_GLOBAL char* p;
};
template <class T> __device__ __forceinline__ GBase global_base(const T* ptr) {  // ptr must be wave-uniform
    const unsigned long long b = (unsigned long long)ptr;
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)b), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(b >> 32));
    GBase g;
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wold-style-cast"
    g.p = (const RTU_GLOBAL char*)(((unsigned long long)hi << 32) | lo);
#pragma clang diagnostic pop
    return g;
}
__device__ __forceinline__ float4 gload4(GBase g, uint32_t byte_off) {
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wold-style-cast"
    const rtu_v4f v = *(const RTU_GLOBAL rtu_v4f*)(g.p + byte_off);
#pragma clang diagnostic pop
    return make_float4(v.x, v.y, v.z, v.w);
}
typedef uint32_t rtu_v2u __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint2 gload2u(GBase g, uint32_t byte_off) {
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wold-style-cast"
    const rtu_v2u v = *(const RTU_GLOBAL rtu_v2u*)(g.p + byte_off);
#pragma clang diagnostic pop
    return make_uint2(v.x, v.y);
}
__device__ __forceinline__ uint32_t gload1(GBase g, uint32_t byte_off) {
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wold-style-cast"
    return *(const RTU_GLOBAL uint32_t*)(g.p + byte_off);
#pragma clang diagnostic pop
}

// One BVH over a mesh's triangles with its leaf-ordered triangle records.
struct DevTree {
    const float4*   bvh;        // 2 float4 per node: {bmin.xyz, index} {bmax.xyz, count}; root = node 1; breadth-first
                                // (nullptr for the fast tree: only its collapsed forms bvh4 / bvh8 are uploaded)
    const float4*   tri;        // 4 float4 per element slot (leaf order), see TriRec in rtu_intersect.h
    const uint32_t* elements;   // element slot -> face id
};

// Two trees per mesh:
//   ref  — the reference's own tree (cy::BVH mean split, cyBVH.h:122-328): defines the ORDER in
//          which triangles are tested, which decides the winner when two triangles give exactly
//          the same t (strict `t < hInfo.z`, objFunctions.cpp:270). The counting variant walks it.
//   fast — a binned-SAH tree over the same triangles (fewer steps per ray). The closest hit is
//          a minimum over the accepted triangles and therefore order-independent UNLESS two
//          accepted triangles tie exactly; the fast walk detects that (rtu_intersect.h) and
//          redoes the ray on `ref`, so the result is always the reference's.
struct DevMesh {
    DevTree ref, fast;
    const uint32_t* f;          // face -> 3 vertex ids
    const float*    v;
    const uint32_t* fn;
    const float*    vn;
    const uint32_t* ft;         // face -> 3 texture-vertex ids, or nullptr (then hits carry uvw = 0)
    const float*    vt;
    float    bmin[3], bmax[3];
    uint32_t n_bvh_nodes, n_elements;
    uint32_t any_empty_box;     // some BVH box has min > max (cannot come from triangles)
    const float4* bvh8;         // the fast tree collapsed to 8 children per node for the cooperative walk: node i =
                                // float4[16 i ..]: child c = {bmin.xyz, ref} {bmax.xyz, -}; ref = index | count << 28 like the
                                // binary nodes (count == 0: node8 index), RTU_REF8_EMPTY for an unused slot; breadth-first
    const float4* bvh4;         // ... and to 4 children per node for the one-lane-per-ray walk: node i = float4[8 i ..]:
                                // {min.x[4]} {min.y[4]} {min.z[4]} {max.x[4]} {max.y[4]} {max.z[4]} {ref[4]} {-}
    uint32_t lds_nodes;         // node8 [0, lds_nodes) are staged in LDS by the cooperative kernels (the top of the tree)
    uint32_t lds_off;           // their offset in the block's LDS node area, in float4
    float    scale;             // largest |coordinate| of the mesh's bounding box (cull margin, rtu_intersect.h)
};

#define RTU_MAX_PCOVER 8      // plane nodes that get a coverage mask for primary rays
struct DevNode {                // one scene-graph node (wave-uniform data)
    float   tm[9], itm[9], pos[3];
    int32_t parent, obj_type, mesh_id, material_id, depth;
    int32_t chain[RTU_MAX_NODE_DEPTH];  // chain[d] = ancestor at depth d (chain[depth] == self)
    // NODE-LEVEL BOUND (fast variant only; SURVEY row f4 — the reference has Node::ComputeChildBoundBox, scene.h:475-489, and
    // never uses it): the world-space box of the object's own bounding box (unit cube / unit square / mesh box) taken
    // through the node's chain of transformations, a little inflated (rtu_capi.hip: world_bounds). A ray that misses it by
    // the margin of rtu_intersect.h "Culling" cannot pass the reference's own box test of this object, so the node is
    // skipped before its transformation and exact test — which still decide every result bit of the rays that remain.
    float   wmin[3], wmax[3];
};

// OCCLUDER LISTS of one (non-ambient light, mesh node) pair, made at upload (rtu_capi.hip: build_light_lists): the mesh seen from the
// light — a point light looks at it through a pinhole at its position, a direct light along its direction — on a G x G grid of
// cells, and for every cell the triangles whose projection (widened by the cull margin, a quarter of a cell of slack) touches it.
// A shadow ray's direction from the light is fixed by its ORIGIN, so one lookup with the origin names every triangle of the
// mesh the ray can possibly touch: an empty cell skips the mesh, a short list replaces the BVH walk (rtu_intersect.h:
// mesh_shadow_cells — the reference's own triangle test on each entry; Shadow() only asks whether there is a hit).
#define RTU_LMASK_LIGHTS 4u
#define RTU_LGRID_MAX 1024u
struct DevLightMask {
    float X[3], Y[3], Z[3], L[3];   // the light's frame; L: its position (point light) or 0 (direct light: orthographic along Z)
    float u0, v0, su, sv;           // cell = ((u - u0) * su, (v - v0) * sv)
    uint32_t usable, point, G, pad;
    const uint32_t* cell_off;       // [G * G + 1] offsets into cell_tri (cell = y * G + x)
    const uint32_t* cell_tri;       // per entry {triangle slot (leaf order of the mesh's `fast` tree: the index of its 64-byte record), zmin (float bits):
                                    // an origin at a smaller depth along Z cannot see the triangle}; a cell's entries in ascending zmin
};

struct DevTexture {              // RtuTexture with the image in device memory
    int32_t type, width, height, pad;
    const uint8_t* rgb;
    float color1[3], color2[3];
};

struct DevScene {
    const DevNode*     nodes;
    const RtuMaterial* materials;
    const RtuLight*    lights;
    const DevMesh*     meshes;
    // textures; textured == 0: none of the following is read and hits carry no uvw
    const DevTexture*  textures;
    const RtuTexMap*   mat_maps;    // nullptr or 4 per material (RTU_MAP_*)
    RtuTexMap bg_map, env_map;      // present only with a real texture
    RtuEnvColor bg, env;
    int32_t  img_w, img_h;          // camera.imgWidth / imgHeight: the background is sampled at (x/imgWidth, y/imgHeight)
    uint32_t textured, pad_tex;
    uint32_t n_nodes, n_lights;
    uint32_t walk_stack_limit;  // test hook (rtu_debug_walk_stack_limit): stack entries the walks of the fast trees may use
    float    wscale;            // largest |coordinate| of any node-level bound: the scale of the cull margin in world space
    unsigned long long obj_mask;  // bit k: node k (< 64) carries an object
    uint32_t nol_ok, n_cover;   // n_cover: mesh nodes with a coverage mask (the first RTU_MAX_COVER of them)
    int32_t  cover_node[8];     // their node indices
    const DevLightMask* lmask;  // [min(non-ambient lights, RTU_LMASK_LIGHTS)][n_cover] occluder lists of shadow rays, or nullptr
    const float4* cover_box[8]; // per masked mesh node: the WORLD-space box of every triangle, 2 float4 {lo, -} {hi, -} (computed at upload in
    uint32_t cover_nf[8];       //   binary64, rounded outwards), and the triangle count   // every non-ambient light's intensity is finite and below 1e15 (make_info: lights behind the surface)
    // PLANE nodes get a coverage mask too (slots n_cover .. n_cover + n_pcover - 1 of KernelArgs::cover): a Plane is the unit square of
    // its node, and the screen rectangle of a square seen at an angle is mostly air (k_plane_cover projects the square itself)
    uint32_t n_pcover, pad_pc;
    int32_t  pcover_node[RTU_MAX_PCOVER];
    float    pcover_quad[RTU_MAX_PCOVER][4][3];  // the square's corners (-1,-1) (1,-1) (1,1) (-1,1) through the node's chain of transformations (binary64, rounded)
    uint32_t dbg;               // experiment switches (rtu_debug_flags), as KernelArgs::dbg
    uint32_t node_bounds;       // 0: node-level bounds off (test hook rtu_debug_node_bounds; results must not change)
    float    background[3];     // background.Sample(...) for an untextured / NULL-map background
    float    environment[3];    // environment.SampleEnvironment(...) likewise
};

// ---- wavefront state ------------------------------------------------------------
// MtlBlinn::Shade recurses (depth <= 5, branching <= 3) and combines child results
// non-linearly. On the GPU the recursion tree is evaluated LEVEL BY LEVEL: every
// Shade() invocation ("frame") of recursion level L of every pixel lives in level
// L's frame arrays; all rays of all frames of a level (shadow rays, the refracted /
// TIR ray, the Fresnel ray, the mirror ray) are traced in parallel by one launch,
// children become frames of level L+1, and results are combined bottom-up in the
// reference's exact term order.
#define RTU_MAX_LEVELS        (RTU_MAX_BOUNCE + 1)
#define RTU_MAX_BATCH 16          // samples of recipe S rendered by one launch sequence
#define RTU_MAX_FRAME_BATCH 128   // frames of recipe W rendered by one launch sequence (== RTU_MAX_FRAMES_IN_FLIGHT)
#define RTU_MAX_SHADOW_LIGHTS 13  // non-ambient lights (a ray id keeps 4 bits for lights + 3 secondary slots); more => RTU_ERR_UNSUPPORTED

// frame info word (fa.w)
#define RTU_FI_MTL_MASK   0x3FFFFu        // bits 0-17 material id
#define RTU_FI_BOUNCE_SH  18              // bits 18-20 bounceCount of this Shade() call
#define RTU_FI_FRONT      (1u << 21)      // hInfo.front
#define RTU_FI_SH         (1u << 22)      // light loop runs (front face, mtlFunctions.cpp:125)
#define RTU_FI_MAIN       (1u << 23)      // refraction property exists and bounce > 0 (:158-160)
#define RTU_FI_TIR        (1u << 24)      // sinTheta2 > 1 (:205): the main ray is the TIR reflection
#define RTU_FI_C          (1u << 25)      // reflection property exists and bounce > 0 (:273)
#define RTU_FI_NOL_SH     27              // bits 27-31: non-ambient light j < 5 is behind the surface (N.L clearly negative): its term of the
#define RTU_FI_NOL_LIGHTS 5u              //   light loop is exactly +-0 whatever Shadow() says, so the fast variant fires no shadow ray (make_info)
#define RTU_FI_AMB        (1u << 26)      // recipe P: this Shade() tree receives MonteCarlo()'s one AmbientLight (LevelBuffers::famb) instead of the scene's lights

// child status codes in fchild
#define RTU_CH_NONE   (-1)   // slot not active
#define RTU_CH_MISS   (-2)   // ray missed
#define RTU_CH_WHITE  (-3)   // hit a node without material: Shade() == (1,1,1)

struct LevelBuffers {
    float4* fa;      // {p.xyz, info}
    float4* fb;      // {N.xyz, w}: w = level 0: pixel index in the launch's [batch entry][pixel of the shard] space; deeper
                     //   levels: the key of the frame's sample streams (recipe S / P) or its batch entry (frames in flight).
                     //   Recipe P: the key at every level
    float4* fc;      // {ray dir.xyz, w}: w = hInfo.z of the hit this frame shades (read at level 0 for the z output);
                     //   recipe P: the chain the frame belongs to, at every level
    float4* fres;    // {Shade() result rgb, -}; holds the direct term until combined
    int4*   fchild;  // {main child, Fresnel child, mirror child, pending}
    float*  fsh;     // [cap * nsl] Shadow() of every non-ambient light
    float4* fslot;   // [cap * 3 * 2] closest hit of the three secondary rays: {p.xyz,z} {N.xyz,packed}
    float4* fuv;     // textured scenes only: [cap] {hInfo.uvw, -} of the frame; [cap * 3] of the secondary hits
    float4* fsuv;
    uint32_t* lmain; // [cap] per shard: the frames (index within the shard) that fire a refracted / TIR (and Fresnel) ray
    uint32_t* lrefl; // [cap] per shard: the frames that fire a mirror ray — k_trace visits these lists for the secondary slots
    uint32_t* fpend; // [cap] per shard: the frames (index within the shard) that wait for children — what k_combine visits
    float4* famb;    // recipe P only: [cap] intensity of the AmbientLight of a RTU_FI_AMB frame
    uint32_t cap_s;  // capacity of ONE shard; frame id = shard * cap_s + index within the shard
    uint32_t pad;
};

// Frames are appended with one atomic per wavefront. A single counter word saturates at
// ~88 atomics/us on MI355X (32 400 tiles => ~370 us, measured), so every level's arrays
// are split into RTU_SHARDS independent regions, each with its own counter; a wavefront
// appends to the shard of its own index, which keeps the regions balanced.
// The cooperative kernels (8 lanes per ray): k_primary2c is one 1024-thread workgroup per CU (128 rays), k_trace2c two 512-thread
// workgroups per CU (64 rays each: measured, one frame alone: 47 / 28 / 30 instead of 55 / 35 / 35 us for levels 0 / 1 / 2 — a
// launch lasts as long as its slowest workgroup, and halves finish sooner —, while k_primary2c's longer list prefers the full one:
// 58 against 75 us). LDS node area of both: 64 KB, the top 256 nodes of the meshes' 8-wide trees in breadth-first order (the teapot's
// whole tree is 230), beside the per-ray traversal stacks (32 / 16 KB): two of the smaller workgroups fit a CU's 160 KB.
#define RTU_COOP_THREADS 1024
#define RTU_COOP_GROUPS  (RTU_COOP_THREADS / 8)
#define RTU_COOP2_THREADS 512
#define RTU_COOP2_GROUPS (RTU_COOP2_THREADS / 8)
#define RTU_STACK8       64   // stack entries per ray of the cooperative walk (8-wide tree: up to 7 pushes per step;
                              // a ray that would need more is finished on the reference's tree, like an exact tie)
#define RTU_REF8_EMPTY   0x0FFFFFFFu
#define RTU_LDS_NODE_F4  4096

#define RTU_MAX_COVER 8       // mesh nodes that get a coverage mask for primary rays
#define RTU_SHARDS 64
#define RTU_TAIL_LEARN   256u   // the host hands levels to k_tail when the cut level held at most this many frames last time
#define RTU_TAIL_DECLINE 4096u  // ... and k_tail refuses a cut level with more than this many (the hint was for another view)
#define RTU_TL_KERNELS 40   // timeline slots: 3 primary + 4 per level + 6 combine (render_impl.h)
#define RTU_TOUCH_FIELDS 12  // Counters::t_* (rtu_intersect.h), RtuTouched (rtu_render.h)
#define RTU_TOUCH_STRIDE 16  // u64 per timeline slot in the counter table of the touched-bytes mode
#define RTU_TL_ENDS 8192u   // exit-stamp slots per kernel (wavefront index modulo; a later wavefront overwrites an earlier one)
#define RTU_TL_STRIDE (64u + RTU_TL_ENDS)

// Device-scope atomics execute at the memory side (the eight XCD L2s are not coherent), a round trip of microseconds, and
// adds to ONE 128-byte line are served one after the other: with the 64 shard counters of a list packed into two lines the
// appends of k_trace(L0) cost 250 of its 373 us (measured: rtu_debug_flags bit 0). So every counter has a line of its own:
// counter of shard s = word [s * RTU_CSTRIDE].
#define RTU_CSTRIDE 32u
struct FrameCounters {
    uint32_t n_frames[RTU_MAX_LEVELS][RTU_SHARDS * RTU_CSTRIDE];
    uint32_t n_defer[RTU_MAX_LEVELS + 1][RTU_SHARDS * RTU_CSTRIDE];  // phase 0 = primary rays, phase 1+L = rays of level L
    uint32_t n_pending[RTU_MAX_LEVELS][RTU_SHARDS * RTU_CSTRIDE];    // frames waiting for children (fpend)
    uint32_t n_lmain[RTU_MAX_LEVELS][RTU_SHARDS * RTU_CSTRIDE];      // entries of lmain / lrefl
    uint32_t n_lrefl[RTU_MAX_LEVELS][RTU_SHARDS * RTU_CSTRIDE];
    uint32_t occ_tiles[RTU_SHARDS * RTU_CSTRIDE];      // 8x8 tiles with anything in them (k_tile_occ), per shard of counters: what the host sizes k_primary's grid by
    uint32_t stage2_frames[RTU_SHARDS * RTU_CSTRIDE];  // level-0 frames appended by stage 2 of the primary phase, per shard like the lists (what the host decides side mode by; zeroed per launch)
    uint32_t overflow;   // a level ran out of capacity: the frame must be re-rendered with more (sticky: rtu_frame_status)
    uint32_t tail_declined;  // k_tail found more frames at its cut level than it takes (RTU_TAIL_DECLINE): it evaluated nothing, the frame
                             // is incomplete and must be rendered again without the tail (sticky, reported like an overflow)
    uint32_t pad[30];
};

struct BatchCam {
    float pos[3], origin[3], u[3], v[3];  // RtuFrameDesc cam_pos / origin / u / v
};

struct KernelArgs {
    DevScene     scene;
    RtuFrameDesc frame;
    float4*      out;               // shard rows * width
    LevelBuffers lv[RTU_MAX_LEVELS];
    LevelBuffers lv_side[RTU_MAX_LEVELS];   // side mode: the level arrays of the frames stage 2 of the primary phase makes (HOST ONLY: copied into lv for its launches)
    unsigned long long* tl;         // GPU-clock timeline stamps (rtu_render_timeline) or nullptr
    FrameCounters* fcnt;
    uint32_t*    defer_list;        // [RTU_SHARDS * defer_cap_s] ray ids waiting for the narrow stage-2 kernel (phases 1..: the levels' rays)
    uint32_t     defer_cap_s;
    // THE PRIMARY PHASE'S OWN LIST AND COUNTERS. Stage 2 of the primary rays (k_primary2 / k_primary2c: a third of the launch sequence,
    // instruction-bound) depends on k_primary alone, and the recursion levels (twenty small latency-bound kernels) depend on the frames
    // k_primary made — not on stage 2's, which on most scenes are a handful (a mesh whose material neither reflects nor refracts is
    // shaded by the lane that found the hit; what remains are hits whose shadow rays could not be settled without a walk). So stage 2
    // runs on the context's helper stream BESIDE the levels ("side mode", chosen by the host per launch): its deferred pixels come from
    // a list of their own (defer_list0, counted in fcnt0->n_defer[0]), the frames it does make go into a small separate set of level
    // arrays and counters (the kernel arguments it is launched with: lv = side arrays, fcnt = fcnt0), and ONE k_tail launch behind it
    // evaluates those frames, subtree by subtree. The host chooses side mode only for a launch shape whose stage 2 made at most a few
    // hundred frames last time (FrameCounters::stage2_frames) — a mirror teapot, or a glass sphere seen through the mesh's bounding
    // box, make thousands, and a k_tail launch is the wrong tool for those; should the view have changed since: more than the side
    // arrays hold or than k_tail takes is refused on the device, reported like an overflow, and the shape goes without side mode.
    // Without side mode fcnt0 == fcnt.
    FrameCounters* fcnt0;
    uint32_t*    defer_list0;       // [RTU_SHARDS * defer_cap0_s] pixels waiting for stage 2 of the primary phase
    uint32_t     defer_cap0_s, side;  // side: 1 in side mode
    uint32_t     pgrid, pad_pg;       // workgroups of k_primary at most (HOST ONLY: launch_all)
    void*        aux_stream;        // HOST ONLY (launch_all): the helper stream and two events of side mode
    void*        aux_ev0;
    void*        aux_ev1;
    uint32_t     dbg;               // experiment switches (rtu_debug_flags); 0 in production
    // COVERAGE MASKS of primary rays (recipe W): per camera of the launch and mesh node, one bit per 8x8-pixel tile of the image:
    // can a primary ray through a pixel of the tile touch ANY triangle of the mesh? (k_mesh_cover: every triangle's world box, widened by
    // the cull margin, projected like the node bound, two pixels of slack.) cover[(entry * n_cover + slot) * (1 + cover_words)]: word 0 = 1 if
    // the mask is unusable (a triangle at or behind the camera plane, or one that covers thousands of tiles), then the bits.
    uint32_t*    cover;
    uint32_t     cover_words, tiles_xf;   // words of one mask; 8x8 tiles per row of the WHOLE image
    uint32_t     cover_faces, pad_cf;     // the largest triangle count among the masked meshes (grid of k_mesh_cover)
    int4*        node_rects;        // recipe W: [batch entry][node] {x0, y0, x1, y1}: the pixels (global x, y; x0 <= x < x1) whose primary ray can
                                    // touch the node's bound from that entry's camera (k_node_rects); nullptr: not in use
    // TILE OCCUPANCY (recipe W, fast variant): per camera of the launch one bit per 8x8 tile OF THE SHARD (tile = band * tiles_x + tx, the
    // index k_primary strides over): 0 = no valid pixel of the tile lies inside any object node's screen rectangle — or, for a masked mesh
    // node, in a tile its coverage mask marks —: k_primary writes the background without looking at cameras, rectangles or masks
    // (k_tile_occ, from node_rects and cover). occ[entry * occ_words + (tile >> 5)]; nullptr: not in use
    const uint32_t* occ;
    uint32_t     occ_words, pad_occ;
    unsigned long long* counters;   // 11 x u64 (RtuStats order); touched-bytes mode: [RTU_TL_KERNELS][RTU_TOUCH_STRIDE]; or nullptr
    uint32_t     tiles_x;           // ceil(width / 8)
    uint32_t     nsl;               // number of non-ambient lights
    int32_t      shadow_light[RTU_MAX_SHADOW_LIGHTS];  // their indices in lights[]
    float        nol_light[RTU_FI_NOL_LIGHTS][4];      // the first non-ambient lights {vec.xyz, 1 = direct}: make_info tests N.L without a dependent load
    uint32_t     n_meshes;
    int32_t      tail_from;         // recursion levels >= this are evaluated by k_tail (RTU_MAX_LEVELS: none)
    // HOST ONLY (launch_all): rays deferred in every tracing phase by the last launch of this shape, + 1 (0: unknown; phase 0 =
    // primary rays, 1 + L = rays of level L). A phase has two stage-2 kernels of which one finds work (narrow_geom); an empty
    // launch of 32768 workgroups costs 9 us, of 512 x 1024 threads 2.4 us (GPU-clock timeline of a single frame). The one that is
    // expected to be idle gets a smaller grid: the one-lane-per-ray kernel one that would still get through twice the last
    // list, the cooperative kernel a token one when the list was beyond twice the threshold. Any grid renders the same image.
    uint32_t     list_n[8];
    uint32_t*    host_launches;     // HOST ONLY (launch_all), touched-bytes mode: launches per timeline slot since the counters were zeroed, or nullptr
    // recipe S (frame.samples >= 1): one launch sequence per sample
    // `batch` consecutive samples at once, as [sample][pixel of the shard] (longer ray lists fill the chip better)
    uint32_t     sampling;          // 0: recipe W
    uint32_t     sample_index;      // first sample of the batch
    uint32_t     batch, batch_pixels, tiles_per_image;
    float        pix_off_x[RTU_MAX_BATCH], pix_off_y[RTU_MAX_BATCH];  // currentOffset + Halton(index, 4 | 5), RenderFunctions.cpp:80-85,96
    // recipe P (config 5): the Monte-Carlo gather. A chain = one sample of one pixel (index as `pix` above);
    // gi_h[(depth * 4 + j) * gi_total + chain], depth 0 (primary hit) .. 4: {p, z} {N, hit | front << 1 | (mtl + 1) << 2}
    // {incoming ray dir, key} {uvw, -}; gi_res[kind * gi_total + chain]: Shade() of the ambient-light tree (0) and of
    // the scene-light tree (1) at the depth just shaded
    float4*      gi_h;
    float4*      gi_res;
    uint32_t     gi_depth, gi_total;
    // a batch of FRAMES of recipe W (rtu_render_frames_device): the same index space, one camera per frame
    uint32_t     frame_batch;       // 0: no
    const BatchCam* cam;            // [batch] in device memory (copied there on the launch stream, ahead of the kernels)
};

// Enqueue one frame (primary pass, then per level: trace, consume; then combine
// bottom-up) on `stream`. Returns hipError_t as int.
// mode: RTU_LAUNCH_ALL; recipe P: RTU_LAUNCH_CHAIN (trace the chain's ray of depth gi_depth, no shading),
// RTU_LAUNCH_SHADE (the two Shade() trees of every chain hit of depth gi_depth)
#define RTU_LAUNCH_ALL   0
#define RTU_LAUNCH_CHAIN 1
#define RTU_LAUNCH_SHADE 2
#define RTU_GI_BOUNCES   4   // monteCarloBounces, RenderFunctions.cpp:31
// stats: 0 the fast variant, 1 the reference-counting variant (RtuStats), 2 the fast variant in touched-bytes mode (RtuTouched;
// recipe W only). probe (may be NULL): bracket the launch in timeline slot `slot` with the two HIP events.
struct LaunchProbe {
    int   slot;
    void* ev0;
    void* ev1;
    int*  recorded;   // set to 1 when the sequence did launch that slot's kernel (a chain launch of recipe P has no k_gi_roots, ...)
};
int rtu_launch_frame(const KernelArgs& args, uint32_t n_tiles, uint32_t bvh_stack_needed, int stats, hipStream_t stream, int mode = RTU_LAUNCH_ALL,
                     const LaunchProbe* probe = nullptr);
int rtu_launch_gi_final(const KernelArgs& args, hipStream_t stream);

// recipe S: add one sample's image to the accumulators / write the mean
int rtu_launch_accumulate(const float4* samples, uint32_t batch, float4* acc, uint32_t* hits, uint32_t pixels, bool first, hipStream_t stream);
int rtu_launch_resolve(const float4* acc, const uint32_t* hits, float4* out, uint32_t pixels, uint32_t samples, hipStream_t stream);

// gamma + Color24 + z of a float4 image: the content of the reference's RenderImage
int rtu_launch_pack_image(const float4* rgbz, unsigned long long pixels, float* z_out, unsigned char* rgb_out, hipStream_t stream);

// the two output images of a batch of frames, 4 bytes per pixel (Color24 + z-image byte), and the per-frame zmin / zmax keys they need
int rtu_launch_minmax_z(const float4* rgbz, uint32_t pixels_per_frame, uint32_t frames, long long* minmax, hipStream_t stream);
int rtu_launch_pack_output(const float4* rgbz, uint32_t pixels_per_frame, uint32_t frames, const long long* minmax, unsigned char* out, hipStream_t stream);

int rtu_launch_selftest_prims(unsigned long long n_rays, unsigned long long seed, unsigned long long* d_mismatches, hipStream_t stream);
int rtu_launch_selftest_fdiv(unsigned long long n_pairs, unsigned long long seed, unsigned long long* d_mismatches, hipStream_t stream);

#endif
