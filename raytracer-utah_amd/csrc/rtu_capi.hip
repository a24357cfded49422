// rtu_capi.hip — host side of the C-ABI in include/rtu_render.h: context, scene
// validation + upload into the HBM layout of rtu_device.h, frame set-up and
// kernel launch. Compiled by hipcc with -ffp-contract=off so the few host-side
// float computations (triangle normals, camera frame) round exactly like the
// reference's CPU code.
#include "rtu_device.h"

#include <algorithm>
#include <array>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <string>
#include <vector>

struct RtuContext {
    int         device = 0;
    hipStream_t stream = nullptr;
    hipStream_t aux_stream = nullptr;        // side mode (rtu_device.h KernelArgs::fcnt0): stage 2 of the primary phase runs here, beside the levels
    hipEvent_t  aux_ev0 = nullptr, aux_ev1 = nullptr;
    hipEvent_t  ev0 = nullptr, ev1 = nullptr;
    std::string error;

    // scene
    bool     has_scene = false;
    std::vector<void*> scene_allocs;
    DevScene dscene{};
    uint32_t bvh_stack_needed = 1;

    // per-frame resources (grown on demand, reused)
    std::vector<void*> level_allocs;
    LevelBuffers lv[RTU_MAX_LEVELS] = {};
    uint32_t level_cap0 = 0;        // pixels the level buffers were sized for
    uint32_t level_nsl = 0;
    uint32_t want_cap_s[RTU_MAX_LEVELS] = {};  // per-shard capacity wanted for levels >= 1 (grown from the counts of an overflowed frame)
    uint32_t want_defer_s = 0;
    FrameCounters* fcnt = nullptr;
    uint32_t* defer_list = nullptr;
    uint32_t  defer_cap_s = 0;
    FrameCounters* fcnt_side = nullptr;      // side mode: counters of the primary phase's defer list and of the side level arrays
    int sequences_in_flight = 1;             // rtu_set_sequences_in_flight: launch sequences the caller keeps in flight on this GPU (all its contexts together)
    uint32_t* defer_list0 = nullptr;         // the primary phase's own defer list
    uint32_t  defer_cap0_s = 0;
    LevelBuffers lv_side[RTU_MAX_LEVELS] = {};
    std::map<uint64_t, bool> side_off;       // launch shapes whose stage 2 made more frames than the side arrays take: no side mode for them
    std::map<uint64_t, uint32_t> occ_hints;   // ... and how many 8x8 tiles had anything in them (k_tile_occ): the grid of k_primary
    std::map<uint64_t, uint64_t> side_frames; // ... and how many level-0 frames stage 2 of the primary phase made in the last launch of a shape
    bool     last_side = false;
    bool     mesh_hits_childless = false;    // no mesh node's material reflects or refracts: a mesh hit's Shade() call is settled by the lane that found it
    bool     any_recursive_material = true;
    bool     textured = false;
    bool     scene_stochastic = false;   // soft shadows / glossy bounces / depth of field: recipe S only
    std::string stochastic_what;
    float4*   acc = nullptr;             // recipe S accumulators: rgb sum + z sum, hit count
    uint32_t* acc_hits = nullptr;
    size_t    acc_pixels = 0;
    float4*   sample_buf = nullptr;      // the images of one batch of samples, [sample][pixel]
    size_t    sample_buf_pixels = 0;
    float4*   gi_h = nullptr;            // recipe P: chain records [5 depths][4][chains], results [2][chains]
    float4*   gi_res = nullptr;
    size_t    gi_chains = 0;
    bool      want_gi = false;           // level buffers carry famb
    // k_tail: the recursion level from which the previous frame of this scene was almost empty (a hint —
    // any value renders the same image); last_tail_from: what the most recent frame was launched with
    int      tail_hint = 0, last_tail_from = RTU_MAX_LEVELS;   // tail_hint: set by rtu_debug_tail_from for the next launch (0: none)
    std::map<uint64_t, int> tail_hints;   // per launch shape (tiles of the launch, feature set): learned cut level
    std::map<uint64_t, std::array<uint32_t, 8>> list_hints;  // ... and the rays deferred in every phase, + 1 (KernelArgs::list_n)
    uint64_t last_tail_key = 0;
    bool     last_stats = false;
    uint32_t n_meshes = 0;
    uint32_t nsl = 0;
    int32_t  shadow_light[RTU_MAX_SHADOW_LIGHTS] = {};
    float    nol_light[RTU_FI_NOL_LIGHTS][4] = {};
    float4* fb = nullptr;
    size_t  fb_bytes = 0;
    unsigned long long* counters = nullptr;  // 11 x u64 (RtuStats), or the touched-bytes table [RTU_TL_KERNELS][RTU_TOUCH_STRIDE]
    uint32_t slot_launches[RTU_TL_KERNELS] = {};  // touched-bytes mode: launches per slot that went into the table (rtu_get_touched_launches)
    // probe: HIP events around the launches of one timeline slot (rtu_probe_kernel)
    static const int kProbePairs = 64;
    int probe_slot = -1, probe_used = 0;
    hipEvent_t probe_ev[2 * kProbePairs] = {};
    struct MeshInfo { uint32_t faces, sah_depth, stack4, nodes4, nodes8; };
    std::vector<MeshInfo> mesh_info;
    struct LightListInfo { uint32_t light, cover, G, entries, longest; };  // the occluder lists built at upload (rtu_light_list_info)
    std::vector<LightListInfo> light_list_info;
    // cameras of a batch of frames: written into the next slot of a ring of pinned host slots, copied to d_cams on the launch
    // stream ahead of the kernels (stream order protects d_cams; an event per slot protects the slot from being rewritten
    // while its copy is still pending)
    static const int kCamSlots = 16;
    BatchCam* d_cams = nullptr;
    BatchCam* h_cams = nullptr;
    hipEvent_t cam_ev[kCamSlots] = {};
    int cam_slot = 0;
    uint32_t dbg = 0;
    const volatile int* cancel = nullptr;    // rtu_set_cancel_flag: polled between the launch sequences of a sampled frame
    uint32_t* cover = nullptr;               // coverage masks of primary rays (KernelArgs::cover), grown on demand
    size_t    cover_cap = 0;                 // in words
    uint32_t  cover_faces = 0;
    uint32_t* occ = nullptr;                 // tile occupancy of primary rays (KernelArgs::occ), grown on demand
    size_t    occ_cap = 0;                   // in words
    int4* node_rects = nullptr;              // [RTU_MAX_FRAME_BATCH][n_nodes] screen rectangles of the node-level bounds (k_node_rects); owned by the scene
    unsigned long long* tl = nullptr;        // timeline stamps, RTU_TL_KERNELS x RTU_TL_STRIDE (rtu_render_timeline)
    bool stamp_next = false;
};

namespace {

const size_t kCounterBytes = (size_t)RTU_TL_KERNELS * RTU_TOUCH_STRIDE * sizeof(unsigned long long);

int fail(RtuContext* ctx, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (ctx) ctx->error = buf;
    return code;
}

#define RTU_HIP(ctx, call)                                                                        \
    do {                                                                                          \
        hipError_t e_ = (call);                                                                   \
        if (e_ != hipSuccess) return fail(ctx, RTU_ERR_HIP, "%s: %s", #call, hipGetErrorString(e_)); \
    } while (0)

void free_scene(RtuContext* ctx) {
    for (void* p : ctx->scene_allocs) (void)hipFree(p);
    ctx->scene_allocs.clear();
    ctx->has_scene = false;
}

template <class T>
int upload(RtuContext* ctx, const T* src, size_t count, const T** dst) {
    *dst = nullptr;
    size_t bytes = sizeof(T) * count;
    if (bytes == 0) bytes = 16;  // keep pointers valid for empty arrays
    void* d = nullptr;
    RTU_HIP(ctx, hipMalloc(&d, bytes));
    ctx->scene_allocs.push_back(d);
    if (count) RTU_HIP(ctx, hipMemcpy(d, src, sizeof(T) * count, hipMemcpyHostToDevice));
    *dst = static_cast<const T*>(d);
    return RTU_OK;
}


// ---- binned-SAH BVH over a mesh's triangles (the `fast` tree of DevMesh) -------------------
// Breadth-first node numbering with adjacent sibling pairs, root = node 1, node 0 unused —
// the layout the kernels expect. Leaves hold <= 4 triangles. Box bounds are the exact
// min/max of the vertex coordinates (no arithmetic), like cy::BVH's.
struct SahTree {
    std::vector<RtuBvhNode> nodes;
    std::vector<uint32_t>   elements;
    uint32_t depth = 0;
};

void build_sah(const RtuMesh& m, SahTree& out) {
    const uint32_t nf = m.nf;
    const uint32_t max_leaf = 4;  // leaves of the one-lane-per-ray walk; the cooperative walk merges subtrees of <= 8 (build_wide8)
    std::vector<float> bmin(3 * (size_t)nf), bmax(3 * (size_t)nf), cen(3 * (size_t)nf);
    for (uint32_t i = 0; i < nf; i++) {
        const uint32_t* fv = m.f + 3 * i;
        for (int k = 0; k < 3; k++) {
            float a = m.v[3 * fv[0] + k], b = m.v[3 * fv[1] + k], c = m.v[3 * fv[2] + k];
            float lo = a < b ? (a < c ? a : c) : (b < c ? b : c);
            float hi = a > b ? (a > c ? a : c) : (b > c ? b : c);
            bmin[3 * i + k] = lo; bmax[3 * i + k] = hi; cen[3 * i + k] = 0.5f * (lo + hi);
        }
    }
    std::vector<uint32_t> idx(nf);
    for (uint32_t i = 0; i < nf; i++) idx[i] = i;
    struct Job { uint32_t begin, end, id, level; };
    out.nodes.assign(2, RtuBvhNode{});
    out.elements.clear();
    std::vector<Job> queue;
    queue.push_back({0, nf, 1, 1});
    auto area = [](const float* lo, const float* hi) {
        float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
        return dx * dy + dy * dz + dz * dx;
    };
    for (size_t qi = 0; qi < queue.size(); qi++) {
        const Job j = queue[qi];
        float lo[3] = {1e30f, 1e30f, 1e30f}, hi[3] = {-1e30f, -1e30f, -1e30f}, clo[3] = {1e30f, 1e30f, 1e30f}, chi[3] = {-1e30f, -1e30f, -1e30f};
        for (uint32_t t = j.begin; t < j.end; t++)
            for (int k = 0; k < 3; k++) {
                uint32_t f = idx[t];
                if (bmin[3 * f + k] < lo[k]) lo[k] = bmin[3 * f + k];
                if (bmax[3 * f + k] > hi[k]) hi[k] = bmax[3 * f + k];
                if (cen[3 * f + k] < clo[k]) clo[k] = cen[3 * f + k];
                if (cen[3 * f + k] > chi[k]) chi[k] = cen[3 * f + k];
            }
        if (out.nodes.size() <= j.id) out.nodes.resize(j.id + 1, RtuBvhNode{});
        RtuBvhNode n{};
        for (int k = 0; k < 3; k++) { n.bmin[k] = lo[k]; n.bmax[k] = hi[k]; }
        if (j.level > out.depth) out.depth = j.level;
        const uint32_t count = j.end - j.begin;
        if (count <= max_leaf) {
            n.index = (uint32_t)out.elements.size();
            n.count = count;
            for (uint32_t t = j.begin; t < j.end; t++) out.elements.push_back(idx[t]);
            out.nodes[j.id] = n;
            continue;
        }
        // best binned split over the three axes
        const int NB = 16;
        int bestAxis = -1, bestBin = 0;
        float bestCost = 3.0e38f;
        for (int ax = 0; ax < 3; ax++) {
            float ext = chi[ax] - clo[ax];
            if (!(ext > 0)) continue;
            uint32_t cnt[NB] = {};
            float blo[NB][3], bhi[NB][3];
            for (int b = 0; b < NB; b++) for (int k = 0; k < 3; k++) { blo[b][k] = 1e30f; bhi[b][k] = -1e30f; }
            for (uint32_t t = j.begin; t < j.end; t++) {
                uint32_t f = idx[t];
                int b = (int)((cen[3 * f + ax] - clo[ax]) / ext * NB);
                if (b >= NB) b = NB - 1;
                if (b < 0) b = 0;
                cnt[b]++;
                for (int k = 0; k < 3; k++) {
                    if (bmin[3 * f + k] < blo[b][k]) blo[b][k] = bmin[3 * f + k];
                    if (bmax[3 * f + k] > bhi[b][k]) bhi[b][k] = bmax[3 * f + k];
                }
            }
            // sweep: left = bins [0,s], right = (s,NB)
            float rArea[NB];
            uint32_t rCnt[NB];
            {
                float rl[3] = {1e30f, 1e30f, 1e30f}, rh[3] = {-1e30f, -1e30f, -1e30f};
                uint32_t rc = 0;
                for (int b = NB - 1; b > 0; b--) {
                    for (int k = 0; k < 3; k++) { if (blo[b][k] < rl[k]) rl[k] = blo[b][k]; if (bhi[b][k] > rh[k]) rh[k] = bhi[b][k]; }
                    rc += cnt[b];
                    rArea[b] = rc ? area(rl, rh) : 0.0f;
                    rCnt[b] = rc;
                }
            }
            float ll[3] = {1e30f, 1e30f, 1e30f}, lh[3] = {-1e30f, -1e30f, -1e30f};
            uint32_t lc = 0;
            for (int sI = 0; sI < NB - 1; sI++) {
                for (int k = 0; k < 3; k++) { if (blo[sI][k] < ll[k]) ll[k] = blo[sI][k]; if (bhi[sI][k] > lh[k]) lh[k] = bhi[sI][k]; }
                lc += cnt[sI];
                if (lc == 0 || rCnt[sI + 1] == 0) continue;
                float cost = area(ll, lh) * (float)lc + rArea[sI + 1] * (float)rCnt[sI + 1];
                if (cost < bestCost) { bestCost = cost; bestAxis = ax; bestBin = sI; }
            }
        }
        uint32_t mid;
        if (j.level > 28) bestAxis = -1;  // a degenerate distribution must not cost unbounded depth: halve from here on
        if (bestAxis < 0) {
            mid = j.begin + count / 2;  // all centroids coincide: split the list in half
        } else {
            float ext = chi[bestAxis] - clo[bestAxis];
            uint32_t i = j.begin, e = j.end;
            while (i < e) {
                uint32_t f = idx[i];
                int b = (int)((cen[3 * f + bestAxis] - clo[bestAxis]) / ext * NB);
                if (b >= NB) b = NB - 1;
                if (b < 0) b = 0;
                if (b <= bestBin) i++;
                else { e--; uint32_t tmp = idx[i]; idx[i] = idx[e]; idx[e] = tmp; }
            }
            mid = i;
            if (mid == j.begin || mid == j.end) mid = j.begin + count / 2;
        }
        const uint32_t child = (uint32_t)out.nodes.size() + (out.nodes.size() & 1u);  // even id: 64-byte aligned pair
        out.nodes.resize(child + 2, RtuBvhNode{});
        n.index = child;
        n.count = 0;
        out.nodes[j.id] = n;
        queue.push_back({j.begin, mid, child, j.level + 1});
        queue.push_back({mid, j.end, child + 1, j.level + 1});
    }
}

// Leaf-order the elements depth-first, so that every subtree owns a contiguous range of element
// slots (the 8-wide tree of the cooperative walk turns whole subtrees of <= 8 triangles into
// leaves). first/total: per binary node, the subtree's slot range.
void dfs_order(SahTree& t, std::vector<uint32_t>& first, std::vector<uint32_t>& total) {
    std::vector<uint32_t> elems;
    elems.reserve(t.elements.size());
    first.assign(t.nodes.size(), 0);
    total.assign(t.nodes.size(), 0);
    std::vector<std::pair<uint32_t, int>> stack;  // (node, state)
    stack.push_back({1u, 0});
    while (!stack.empty()) {
        auto [id, st] = stack.back();
        RtuBvhNode& n = t.nodes[id];
        if (n.count != 0) {
            first[id] = (uint32_t)elems.size();
            total[id] = n.count;
            for (uint32_t i = 0; i < n.count; i++) elems.push_back(t.elements[n.index + i]);
            n.index = first[id];
            stack.pop_back();
        } else if (st == 0) {
            first[id] = (uint32_t)elems.size();
            stack.back().second = 1;
            stack.push_back({n.index, 0});
        } else if (st == 1) {
            stack.back().second = 2;
            stack.push_back({n.index + 1, 0});
        } else {
            total[id] = total[n.index] + total[n.index + 1];
            stack.pop_back();
        }
    }
    t.elements.swap(elems);
}

// The binary SAH tree collapsed to eight children per node (for the cooperative walk, one child
// per lane): starting from a node's two children, the inner child with the largest surface area
// is replaced by ITS two children until eight are reached or only leaves remain. Breadth-first,
// so that the top of the tree is the prefix staged into LDS. Same leaves, same boxes.
void build_wide8(const SahTree& t, const std::vector<uint32_t>& first, const std::vector<uint32_t>& total, std::vector<float4>& out) {
    struct Job { uint32_t bin, id; };
    auto is_leaf = [&](uint32_t b) { return total[b] <= 8u; };  // a subtree of <= 8 triangles is one leaf round for eight lanes
    out.assign(16, make_float4(0, 0, 0, 0));
    std::vector<Job> queue;
    queue.push_back({1u, 0u});
    auto area = [&](uint32_t b) {
        const RtuBvhNode& n = t.nodes[b];
        float dx = n.bmax[0] - n.bmin[0], dy = n.bmax[1] - n.bmin[1], dz = n.bmax[2] - n.bmin[2];
        return dx * dy + dy * dz + dz * dx;
    };
    for (size_t qi = 0; qi < queue.size(); qi++) {
        const Job j = queue[qi];
        std::vector<uint32_t> set;
        if (is_leaf(j.bin)) {
            set.push_back(j.bin);  // a mesh of <= 8 triangles: the root is a leaf
        } else {
            set.push_back(t.nodes[j.bin].index);
            set.push_back(t.nodes[j.bin].index + 1);
            while (set.size() < 8) {
                int best = -1;
                float bestA = -1.0f;
                for (size_t i = 0; i < set.size(); i++)
                    if (!is_leaf(set[i]) && area(set[i]) > bestA) { bestA = area(set[i]); best = (int)i; }
                if (best < 0) break;
                const uint32_t b = set[(size_t)best];
                set[(size_t)best] = t.nodes[b].index;
                set.push_back(t.nodes[b].index + 1);
            }
        }
        for (uint32_t c = 0; c < 8; c++) {
            float4 lo = make_float4(INFINITY, INFINITY, INFINITY, 0), hi = make_float4(-INFINITY, -INFINITY, -INFINITY, 0);
            uint32_t ref = RTU_REF8_EMPTY;
            if (c < set.size()) {
                const RtuBvhNode& n = t.nodes[set[c]];
                lo = make_float4(n.bmin[0], n.bmin[1], n.bmin[2], 0);
                hi = make_float4(n.bmax[0], n.bmax[1], n.bmax[2], 0);
                if (is_leaf(set[c])) {
                    ref = first[set[c]] | (total[set[c]] << 28);
                } else {
                    ref = (uint32_t)(out.size() / 16);
                    out.resize(out.size() + 16, make_float4(0, 0, 0, 0));
                    queue.push_back({set[c], ref});
                }
            }
            memcpy(&lo.w, &ref, 4);
            out[(size_t)j.id * 16 + 2 * c] = lo;
            out[(size_t)j.id * 16 + 2 * c + 1] = hi;
        }
    }
}

// ... and to FOUR children per node for the one-lane-per-ray walk (half the dependent fetches of the
// binary tree at the same instruction count). A node is 8 float4, structure of arrays so that a lane
// picks the near / far plane arrays by the sign of its ray: {min.x[4]} {min.y[4]} {min.z[4]}
// {max.x[4]} {max.y[4]} {max.z[4]} {ref[4]} {-}. stack_need: the deepest stack a walk can build
// (a step pushes all hit children but the nearest).
void build_wide4(const SahTree& t, std::vector<float4>& out, uint32_t& stack_need) {
    struct Job { uint32_t bin, id, pushed; };
    out.assign(8, make_float4(0, 0, 0, 0));
    std::vector<Job> queue;
    queue.push_back({1u, 0u, 0u});
    stack_need = 1;
    auto area = [&](uint32_t b) {
        const RtuBvhNode& n = t.nodes[b];
        float dx = n.bmax[0] - n.bmin[0], dy = n.bmax[1] - n.bmin[1], dz = n.bmax[2] - n.bmin[2];
        return dx * dy + dy * dz + dz * dx;
    };
    for (size_t qi = 0; qi < queue.size(); qi++) {
        const Job j = queue[qi];
        std::vector<uint32_t> set;
        if (t.nodes[j.bin].count != 0) {
            set.push_back(j.bin);
        } else {
            set.push_back(t.nodes[j.bin].index);
            set.push_back(t.nodes[j.bin].index + 1);
            while (set.size() < 4) {
                int best = -1;
                float bestA = -1.0f;
                for (size_t i = 0; i < set.size(); i++)
                    if (t.nodes[set[i]].count == 0 && area(set[i]) > bestA) { bestA = area(set[i]); best = (int)i; }
                if (best < 0) break;
                const uint32_t b = set[(size_t)best];
                set[(size_t)best] = t.nodes[b].index;
                set.push_back(t.nodes[b].index + 1);
            }
        }
        const uint32_t pushed = j.pushed + (uint32_t)set.size() - 1u;
        if (pushed + 1u > stack_need) stack_need = pushed + 1u;
        float v[7][4];
        for (uint32_t c = 0; c < 4; c++) {
            for (int k = 0; k < 3; k++) { v[k][c] = INFINITY; v[3 + k][c] = -INFINITY; }
            uint32_t ref = RTU_REF8_EMPTY;
            if (c < set.size()) {
                const RtuBvhNode& n = t.nodes[set[c]];
                for (int k = 0; k < 3; k++) { v[k][c] = n.bmin[k]; v[3 + k][c] = n.bmax[k]; }
                if (n.count != 0) {
                    ref = n.index | (n.count << 28);
                } else {
                    ref = (uint32_t)(out.size() / 8);
                    out.resize(out.size() + 8, make_float4(0, 0, 0, 0));
                    queue.push_back({set[c], ref, pushed});
                }
            }
            memcpy(&v[6][c], &ref, 4);
        }
        for (int r = 0; r < 7; r++) out[(size_t)j.id * 8 + r] = make_float4(v[r][0], v[r][1], v[r][2], v[r][3]);
    }
}

// 64-byte triangle records (TriRec, rtu_intersect.h) in the order of `elements`: the
// ray-independent part of TriObj::IntersectTriangle (objFunctions.cpp:259-300) evaluated with
// the same float ops.
void build_tri_records(const RtuMesh& m, const uint32_t* elements, uint32_t n, std::vector<float4>& tri) {
    tri.resize((size_t)n * 4);
    for (uint32_t e = 0; e < n; e++) {
        const uint32_t* fv = m.f + 3 * elements[e];
        f3 A = ld3(m.v + 3 * fv[0]), B = ld3(m.v + 3 * fv[1]), C = ld3(m.v + 3 * fv[2]);
        f3 N = norm3(cross3(B - A, C - A));                                        // :263
        float anx = fabsf(N.x), any = fabsf(N.y), anz = fabsf(N.z);
        float maxNormalAxis = smax(smax(anx, any), anz);                           // :274
        uint32_t axis = (maxNormalAxis == anx) ? 0u : (maxNormalAxis == any) ? 1u : 2u;  // :278-296
        float ax = axis == 0 ? A.y : A.x, ay = axis == 2 ? A.y : A.z;
        float bx = axis == 0 ? B.y : B.x, by = axis == 2 ? B.y : B.z;
        float cx = axis == 0 ? C.y : C.x, cy = axis == 2 ? C.y : C.z;
        float e1x = cx - ax, e1y = cy - ay, e2x = bx - ax, e2y = by - ay;
        float TriABCArea = (float)((double)((-e1y) * e2x + e1x * e2y) / 2.0);      // :298, Point2::Cross
        double rcp = 1.0 / (double)TriABCArea;
        uint64_t bits;
        memcpy(&bits, &rcp, 8);
        uint32_t lo = (uint32_t)bits, hi = (uint32_t)(bits >> 32);
        float flo, fhi, faxis;
        memcpy(&flo, &lo, 4); memcpy(&fhi, &hi, 4); memcpy(&faxis, &axis, 4);
        tri[4 * e + 0] = make_float4(A.x, A.y, A.z, N.x);
        tri[4 * e + 1] = make_float4(N.y, N.z, ax, ay);
        tri[4 * e + 2] = make_float4(e1x, e1y, e2x, e2y);
        tri[4 * e + 3] = make_float4(flo, fhi, faxis, 0.0f);
    }
}

// NODE-LEVEL BOUNDS (DevNode::wmin / wmax; the argument is in rtu_intersect.h, trace): the object's own bounding box —
// the unit cube of a sphere, the unit square of a plane (objects.h:25,37), the mesh's box — taken corner by corner through
// the node's chain of transformations (p -> tm p + pos, scene.h:508-512) in binary64, then widened by
//   * 1e-5 of the largest coordinate (the chain itself is binary32 on the device and in the reference), and
//   * for a sphere, what the cancellation in b*b - 4ac can move a grazing root (objFunctions.cpp:25-27): the discriminant
//     carries an error of ~4 ulp of b*b, i.e. the ray may "touch" a sphere it passes at up to ~2.4e-7 * (D/R)^2 radii, and a
//     grazing root is off by up to ~5e-4 * D in t; with D bounded by the scene's diameter S both stay below
//     4e-6 * S^2 / R + 1e-3 * S ... the latter only matters when R < 1e-3 * S, where the former is already larger. R is the
//     smallest half-extent of the sphere's world box.
// Returns the largest |coordinate| of all bounds (the scale of the per-ray margin).
float world_bounds(const RtuSceneDesc* s, std::vector<DevNode>& nodes) {
    const uint32_t n = s->n_nodes;
    std::vector<std::array<double, 6>> box(n);
    std::vector<bool> has(n, false);
    double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
    for (uint32_t i = 0; i < n; i++) {
        const RtuNode& nd = s->nodes[i];
        double l[3], h[3];
        if (nd.obj_type == RTU_OBJ_SPHERE) { l[0] = l[1] = l[2] = -1; h[0] = h[1] = h[2] = 1; }
        else if (nd.obj_type == RTU_OBJ_PLANE) { l[0] = l[1] = -1; h[0] = h[1] = 1; l[2] = h[2] = 0; }
        else if (nd.obj_type == RTU_OBJ_TRIMESH) {
            const RtuMesh& m = s->meshes[nd.mesh_id];
            for (int k = 0; k < 3; k++) { l[k] = m.bound_min[k]; h[k] = m.bound_max[k]; }
        } else continue;
        double wl[3] = {1e300, 1e300, 1e300}, wh[3] = {-1e300, -1e300, -1e300};
        for (int c = 0; c < 8; c++) {
            double p[3] = {(c & 1) ? h[0] : l[0], (c & 2) ? h[1] : l[1], (c & 4) ? h[2] : l[2]};
            for (int j = (int)i; j >= 0; j = s->nodes[j].parent) {
                const RtuNode& a = s->nodes[j];
                const double q[3] = {p[0] * a.tm[0] + p[1] * a.tm[3] + p[2] * a.tm[6] + a.pos[0],
                                     p[0] * a.tm[1] + p[1] * a.tm[4] + p[2] * a.tm[7] + a.pos[1],
                                     p[0] * a.tm[2] + p[1] * a.tm[5] + p[2] * a.tm[8] + a.pos[2]};
                p[0] = q[0]; p[1] = q[1]; p[2] = q[2];
            }
            for (int k = 0; k < 3; k++) { wl[k] = std::min(wl[k], p[k]); wh[k] = std::max(wh[k], p[k]); }
        }
        for (int k = 0; k < 3; k++) { box[i][k] = wl[k]; box[i][3 + k] = wh[k]; lo[k] = std::min(lo[k], wl[k]); hi[k] = std::max(hi[k], wh[k]); }
        has[i] = true;
    }
    double S = 0;
    for (int k = 0; k < 3; k++) if (hi[k] >= lo[k]) S += (hi[k] - lo[k]) * (hi[k] - lo[k]);
    S = std::sqrt(S);
    double scale = 0;
    for (uint32_t i = 0; i < n; i++) {
        DevNode& d = nodes[i];
        for (int k = 0; k < 3; k++) { d.wmin[k] = -INFINITY; d.wmax[k] = INFINITY; }
        if (!has[i]) continue;
        double m = 0;
        for (int k = 0; k < 6; k++) m = std::max(m, std::fabs(box[i][k]));
        double widen = 1e-5 * m;
        if (s->nodes[i].obj_type == RTU_OBJ_SPHERE) {
            double R = 1e300;
            for (int k = 0; k < 3; k++) R = std::min(R, 0.5 * (box[i][3 + k] - box[i][k]));
            widen += R > 0 ? 4e-6 * S * S / R : INFINITY;
            if (R < 1e-3 * S) widen += 1e-3 * S;
        }
        bool finite = std::isfinite(widen);
        for (int k = 0; k < 3 && finite; k++) finite = std::isfinite(box[i][k]) && std::isfinite(box[i][3 + k]);
        if (!finite) continue;  // NaN / infinite transformation: no bound (everything passes an infinite box)
        for (int k = 0; k < 3; k++) {
            d.wmin[k] = std::nextafter((float)(box[i][k] - widen), -INFINITY);
            d.wmax[k] = std::nextafter((float)(box[i][3 + k] + widen), INFINITY);
            scale = std::max(scale, std::max(std::fabs((double)d.wmin[k]), std::fabs((double)d.wmax[k])));
        }
    }
    return (float)scale;
}

// OCCLUDER LISTS OF SHADOW RAYS (DevLightMask): for each of the first RTU_LMASK_LIGHTS non-ambient lights and each masked mesh
// node, the mesh as the light sees it — through a pinhole at a point light, looking at the centre of the mesh; along the
// direction of a direct light — on a G x G grid, and per cell the triangles that a shadow ray whose ORIGIN projects into the
// cell can possibly touch. A triangle is entered into every cell that its projection, grown by a slack S, overlaps (exact
// triangle / square overlap: the separating-axis test on the square's sides and the triangle's edges), where S covers
//   * the cull margin: the ray's line, its rounded direction and the binary32 transformation chain stay within
//     wid = 1e-4 * scene scale + 1e-5 * |coordinates| of the ideal segment origin -> light (shadow rays start on the scene's
//     surfaces: |origin| <= the scene's scale); a displacement of wid on every axis moves a projection by at most
//     wid * sqrt(3) * (1 + |u|) / (depth - wid * sqrt(3)) (pinhole; wid * sqrt(3) orthographic);
//   * the device's binary32 evaluation of the cell coordinates (err: a few 1e-6 of the operands, bounded below per list;
//     a list whose bound exceeds a quarter of a cell is not used);
//   * a quarter of a cell on top.
// A list is unusable when some corner of a triangle's widened box is not in front of the pinhole (the light is inside or too
// close to the mesh's hull) or the mesh has no extent from there. G grows with the triangle count (a triangle spans a few
// cells) up to RTU_LGRID_MAX and is halved while the lists would hold more than 32 M entries.
struct CoverMesh {
    std::vector<float4> boxes;      // per face: world-space box {lo} {hi}, rounded outwards
    std::vector<double> verts;      // per face: 3 world-space vertices (9 doubles)
    std::vector<uint32_t> slot_of;  // face -> slot in the leaf order of the mesh's fast tree
};
// One list: pure host arithmetic (no GPU) — the frame, the grid and the cells' entries {slot, zmin bits} of light `l` looking at the mesh
// `cm`. false: no usable list from there. (rtu_debug_light_list hands the result to the CPU tests, which check it ray by ray.)
struct HostLightList {
    DevLightMask m;                 // frame, offsets, scale, G, point (the device pointers stay null here)
    std::vector<uint32_t> off, ent;
};
bool compute_light_list(const RtuLight& l, const CoverMesh& cm, float wscale, HostLightList& out) {
    DevLightMask& m = out.m;
    memset(&m, 0, sizeof m);
    std::vector<uint32_t>& off = out.off;
    std::vector<uint32_t>& ent = out.ent;
    const double r3 = 1.7320508075688772;
    const bool point = l.type == RTU_LIGHT_POINT;
    {
    const std::vector<float4>& boxes = cm.boxes;
            const size_t nf = boxes.size() / 2;
            double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
            for (size_t f = 0; f < nf; f++) {
                const float4 a = boxes[2 * f], b = boxes[2 * f + 1];
                const double al[3] = {a.x, a.y, a.z}, bh[3] = {b.x, b.y, b.z};
                for (int k = 0; k < 3; k++) { lo[k] = std::min(lo[k], al[k]); hi[k] = std::max(hi[k], bh[k]); }
            }
            double L[3] = {0, 0, 0}, Z[3];
            if (point) {
                for (int k = 0; k < 3; k++) { L[k] = l.vec[k]; Z[k] = 0.5 * (lo[k] + hi[k]) - L[k]; }
            } else {
                for (int k = 0; k < 3; k++) Z[k] = l.vec[k];
            }
            const double zl = std::sqrt(Z[0] * Z[0] + Z[1] * Z[1] + Z[2] * Z[2]);
            if (!(zl > 0) || !std::isfinite(zl)) return false;  // unusable
            for (int k = 0; k < 3; k++) Z[k] /= zl;
            int ax = std::fabs(Z[0]) <= std::fabs(Z[1]) ? (std::fabs(Z[0]) <= std::fabs(Z[2]) ? 0 : 2) : (std::fabs(Z[1]) <= std::fabs(Z[2]) ? 1 : 2);
            double A[3] = {0, 0, 0};
            A[ax] = 1;
            double X[3] = {Z[1] * A[2] - Z[2] * A[1], Z[2] * A[0] - Z[0] * A[2], Z[0] * A[1] - Z[1] * A[0]};
            const double xl = std::sqrt(X[0] * X[0] + X[1] * X[1] + X[2] * X[2]);
            for (int k = 0; k < 3; k++) X[k] /= xl;
            const double Y[3] = {Z[1] * X[2] - Z[2] * X[1], Z[2] * X[0] - Z[0] * X[2], Z[0] * X[1] - Z[1] * X[0]};
            // per triangle: the rectangle of its widened box in the light's (u, v) — the extent of the grid, and the check that
            // everything the list stands for lies in front of the pinhole
            double U0 = 1e300, U1 = -1e300, V0 = 1e300, V1 = -1e300, ratio = 1.0;
            bool ok = true;
            std::vector<double> wid_of(nf);
            for (size_t f = 0; f < nf && ok; f++) {
                const float4 a = boxes[2 * f], b = boxes[2 * f + 1];
                const double al[3] = {a.x, a.y, a.z}, bh[3] = {b.x, b.y, b.z};
                double big = 0;
                for (int k = 0; k < 3; k++) big = std::max(big, std::max(std::fabs(al[k]), std::fabs(bh[k])));
                const double wid = 1e-4 * (double)wscale + 1e-5 * big;
                wid_of[f] = wid;
                for (int cn = 0; cn < 8; cn++) {
                    const double q[3] = {((cn & 1) ? bh[0] + wid : al[0] - wid) - L[0], ((cn & 2) ? bh[1] + wid : al[1] - wid) - L[1],
                                         ((cn & 4) ? bh[2] + wid : al[2] - wid) - L[2]};
                    double u = q[0] * X[0] + q[1] * X[1] + q[2] * X[2], v = q[0] * Y[0] + q[1] * Y[1] + q[2] * Y[2];
                    if (point) {
                        const double depth = q[0] * Z[0] + q[1] * Z[1] + q[2] * Z[2];
                        const double len = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2]);
                        if (!(depth > 1e-3 * len)) { ok = false; break; }  // (within 89.94 degrees of the axis: tan stays below 1000)
                        ratio = std::max(ratio, len / depth);
                        u /= depth; v /= depth;
                    }
                    U0 = std::min(U0, u); U1 = std::max(U1, u); V0 = std::min(V0, v); V1 = std::max(V1, v);
                }
            }
            if (!ok || !(U1 > U0) || !(V1 > V0)) return false;
            const double mag = std::max(std::max(std::fabs(U0), std::fabs(U1)), std::max(std::fabs(V0), std::fabs(V1)));
            if (!std::isfinite(mag) || (U1 - U0) < 1e-4 * mag || (V1 - V0) < 1e-4 * mag) return false;  // no extent a float lookup could resolve
            // grid size: a triangle of an evenly tessellated surface spans ~ G / sqrt(nf / 2) cells; aim at six of them
            uint32_t G = 64;
            static const double kSpan = [] { const char* e = getenv("RTU_LGRID_SPAN"); return e ? atof(e) : 6.0; }();  // tuning knob (any value renders the same image)
            while (G < RTU_LGRID_MAX && (double)G < kSpan * std::sqrt((double)nf * 0.5)) G *= 2;
            for (;; G /= 2) {
                if (G < 16u) { ok = false; break; }
                // the grid spans the extent plus two cells on every side
                const double du = (U1 - U0) / ((double)G - 4), dv = (V1 - V0) / ((double)G - 4);
                const double gu0 = U0 - 2 * du, gv0 = V0 - 2 * dv;
                // the device's binary32 cell coordinate: (dot(p - L, X) [/ depth] - u0) * su — every operand good to a few ulp
                const double coord = 16e-7 * ratio * (1.0 + mag);
                const double err_cells = std::max(coord / du, coord / dv) + 4e-7 * (double)G;
                if (!(err_cells < 0.25)) continue;  // (a coarser grid has larger cells)
                const double S0 = 0.25 + err_cells;
                off.assign((size_t)G * G + 1, 0u);
                size_t total = 0;
                bool too_many = false;
                for (int pass = 0; pass < 2 && !too_many; pass++) {
                    if (pass == 1) {
                        uint32_t run = 0;
                        for (size_t i = 0; i < (size_t)G * G; i++) { const uint32_t n = off[i]; off[i] = run; run += n; }
                        off[(size_t)G * G] = run;
                        ent.assign(2 * total, 0u);
                    }
                    for (size_t f = 0; f < nf; f++) {
                        const double* w = cm.verts.data() + 9 * f;
                        double pu[3], pv[3], dmin = 1e300, umax = 0, vmax = 0;
                        for (int k = 0; k < 3; k++) {
                            const double q[3] = {w[3 * k] - L[0], w[3 * k + 1] - L[1], w[3 * k + 2] - L[2]};
                            double u = q[0] * X[0] + q[1] * X[1] + q[2] * X[2], v = q[0] * Y[0] + q[1] * Y[1] + q[2] * Y[2];
                            if (point) {
                                const double depth = q[0] * Z[0] + q[1] * Z[1] + q[2] * Z[2];
                                dmin = std::min(dmin, depth);
                                u /= depth; v /= depth;
                            }
                            pu[k] = (u - gu0) / du; pv[k] = (v - gv0) / dv;  // in cells
                            umax = std::max(umax, std::fabs(u)); vmax = std::max(vmax, std::fabs(v));
                        }
                        const double wd = wid_of[f] * r3;
                        double Su, Sv;
                        if (point) { Su = wd * (1.0 + umax) / (dmin - wd) / du; Sv = wd * (1.0 + vmax) / (dmin - wd) / dv; }  // (dmin > wd: the box corners passed above)
                        else { Su = wd / du; Sv = wd / dv; }
                        const double S = S0 + std::max(Su, Sv);
                        const double bu0 = std::min(pu[0], std::min(pu[1], pu[2])) - S, bu1 = std::max(pu[0], std::max(pu[1], pu[2])) + S;
                        const double bv0 = std::min(pv[0], std::min(pv[1], pv[2])) - S, bv1 = std::max(pv[0], std::max(pv[1], pv[2])) + S;
                        int x0 = (int)std::floor(bu0), x1 = (int)std::floor(bu1), y0 = (int)std::floor(bv0), y1 = (int)std::floor(bv1);
                        x0 = std::max(x0, 0); y0 = std::max(y0, 0);
                        x1 = std::min(x1, (int)G - 1); y1 = std::min(y1, (int)G - 1);
                        // the triangle's edges as separating lines: a cell (a square of half-width 0.5 + S about its centre) lies
                        // beyond edge i when n_i . (centre - v_i) > (|n_i.x| + |n_i.y|) (0.5 + S), n_i the outward normal
                        const double area2 = (pu[1] - pu[0]) * (pv[2] - pv[0]) - (pu[2] - pu[0]) * (pv[1] - pv[0]);
                        const bool edges = std::fabs(area2) > 1e-9;  // (an edge-on triangle has no inside: its bounding box is all there is)
                        double nx[3], ny[3], nd[3];
                        for (int i = 0; i < 3; i++) {
                            const int k = (i + 1) % 3;
                            const double ex = pu[k] - pu[i], ey = pv[k] - pv[i];
                            const double sg = area2 > 0 ? 1.0 : -1.0;
                            nx[i] = sg * ey; ny[i] = -sg * ex;  // outward for a counter-clockwise triangle (area2 > 0)
                            nd[i] = (std::fabs(nx[i]) + std::fabs(ny[i])) * (0.5 + S);
                        }
                        const uint32_t slot = cm.slot_of[f];
                        // the depth (along Z) in front of which an origin cannot see this triangle at all — every point of it, the cull
                        // margin and the rounding of the device's own depth included, lies beyond (entries are sorted by it)
                        double zmin = 1e300;
                        for (int k = 0; k < 3; k++) zmin = std::min(zmin, (w[3 * k] - L[0]) * Z[0] + (w[3 * k + 1] - L[1]) * Z[1] + (w[3 * k + 2] - L[2]) * Z[2]);
                        zmin -= wd + 8e-6 * (r3 * (double)wscale + std::fabs(L[0]) + std::fabs(L[1]) + std::fabs(L[2]));
                        float zf = (float)zmin;
                        if ((double)zf > zmin) zf = std::nextafter(zf, -INFINITY);
                        uint32_t zbits;
                        memcpy(&zbits, &zf, 4);
                        for (int y = y0; y <= y1; y++)
                            for (int x = x0; x <= x1; x++) {
                                bool out = false;
                                if (edges)
                                    for (int i = 0; i < 3 && !out; i++)
                                        out = nx[i] * ((double)x + 0.5 - pu[i]) + ny[i] * ((double)y + 0.5 - pv[i]) > nd[i];
                                if (out) continue;
                                const size_t cell = (size_t)y * G + (size_t)x;
                                if (pass == 0) { off[cell]++; total++; }
                                else { const uint32_t at = off[cell]++; ent[2 * (size_t)at] = slot; ent[2 * (size_t)at + 1] = zbits; }
                            }
                        if (pass == 0 && total > ((size_t)32 << 20)) { too_many = true; break; }
                    }
                }
                if (too_many) continue;
                // pass 1 advanced every offset to the end of its cell: shift back
                for (size_t i = (size_t)G * G; i > 0; i--) off[i] = off[i - 1];
                off[0] = 0;
                {   // nearest to the light first: a walk of the list ends at the first entry that lies beyond the ray's origin
                    std::vector<std::pair<float, uint32_t>> tmp;
                    for (size_t cell = 0; cell < (size_t)G * G; cell++) {
                        const uint32_t b = off[cell], e = off[cell + 1];
                        if (e - b < 2u) continue;
                        tmp.clear();
                        for (uint32_t i = b; i < e; i++) { float z; memcpy(&z, &ent[2 * (size_t)i + 1], 4); tmp.push_back({z, ent[2 * (size_t)i]}); }
                        std::stable_sort(tmp.begin(), tmp.end(), [](const std::pair<float, uint32_t>& a, const std::pair<float, uint32_t>& b2) { return a.first < b2.first; });
                        for (uint32_t i = b; i < e; i++) { memcpy(&ent[2 * (size_t)i + 1], &tmp[i - b].first, 4); ent[2 * (size_t)i] = tmp[i - b].second; }
                    }
                }
                for (int k = 0; k < 3; k++) { m.X[k] = (float)X[k]; m.Y[k] = (float)Y[k]; m.Z[k] = (float)Z[k]; m.L[k] = (float)L[k]; }
                m.u0 = (float)gu0; m.v0 = (float)gv0; m.su = (float)(1.0 / du); m.sv = (float)(1.0 / dv);
                m.point = point ? 1u : 0u;
                m.G = G;
                break;
            }
            if (!ok) return false;
    }
    m.usable = 1u;
    return true;
}

// The triangles of mesh node `node` in world space: every vertex through the chain p -> tm p + pos in binary64; per face its three
// vertices and their box rounded outwards. fast_elements: slot of the mesh's fast tree -> face.
void make_cover_mesh(const RtuSceneDesc* s, uint32_t node, const std::vector<uint32_t>& fast_elements, CoverMesh& cm) {
    const RtuMesh& m = s->meshes[s->nodes[node].mesh_id];
    cm.boxes.assign((size_t)m.nf * 2, make_float4(0, 0, 0, 0));
    cm.verts.assign((size_t)m.nf * 9, 0.0);
    for (uint32_t f = 0; f < m.nf; f++) {
        double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
        for (int v = 0; v < 3; v++) {
            const float* lp = m.v + 3 * (size_t)m.f[3 * (size_t)f + v];
            double p[3] = {lp[0], lp[1], lp[2]};
            for (int j = (int)node; j >= 0; j = s->nodes[j].parent) {
                const RtuNode& t = s->nodes[j];
                const double q[3] = {p[0] * t.tm[0] + p[1] * t.tm[3] + p[2] * t.tm[6] + t.pos[0], p[0] * t.tm[1] + p[1] * t.tm[4] + p[2] * t.tm[7] + t.pos[1],
                                     p[0] * t.tm[2] + p[1] * t.tm[5] + p[2] * t.tm[8] + t.pos[2]};
                p[0] = q[0]; p[1] = q[1]; p[2] = q[2];
            }
            for (int k = 0; k < 3; k++) { lo[k] = std::min(lo[k], p[k]); hi[k] = std::max(hi[k], p[k]); cm.verts[9 * (size_t)f + 3 * v + k] = p[k]; }
        }
        cm.boxes[2 * (size_t)f] = make_float4(std::nextafter((float)lo[0], -INFINITY), std::nextafter((float)lo[1], -INFINITY), std::nextafter((float)lo[2], -INFINITY), 0.0f);
        cm.boxes[2 * (size_t)f + 1] = make_float4(std::nextafter((float)hi[0], INFINITY), std::nextafter((float)hi[1], INFINITY), std::nextafter((float)hi[2], INFINITY), 0.0f);
    }
    cm.slot_of.assign(m.nf, 0u);
    for (uint32_t sl = 0; sl < (uint32_t)fast_elements.size(); sl++) cm.slot_of[fast_elements[sl]] = sl;
}

int build_light_lists(RtuContext* ctx, const RtuSceneDesc* s, const std::vector<CoverMesh>& cover, float wscale, DevScene& ds) {
    ds.lmask = nullptr;
    std::vector<uint32_t> lights;
    for (uint32_t i = 0; i < s->n_lights && lights.size() < RTU_LMASK_LIGHTS; i++)
        if (s->lights[i].type != RTU_LIGHT_AMBIENT) lights.push_back(i);
    const uint32_t nc = (uint32_t)cover.size();
    if (lights.empty() || nc == 0) return RTU_OK;
    std::vector<DevLightMask> masks(lights.size() * nc);
    memset(masks.data(), 0, masks.size() * sizeof(DevLightMask));
    for (size_t j = 0; j < lights.size(); j++)
        for (uint32_t c = 0; c < nc; c++) {
            HostLightList hl;
            if (!compute_light_list(s->lights[lights[j]], cover[c], wscale, hl)) continue;
            DevLightMask& m = masks[j * nc + c];
            m = hl.m;
            m.usable = 0u;
            int rc;
            if ((rc = upload(ctx, hl.off.data(), hl.off.size(), &m.cell_off)) != RTU_OK) return rc;
            if ((rc = upload(ctx, hl.ent.data(), hl.ent.size(), &m.cell_tri)) != RTU_OK) return rc;
            m.usable = 1u;
            uint32_t longest = 0;
            for (size_t i = 0; i < (size_t)m.G * m.G; i++) longest = std::max(longest, hl.off[i + 1] - hl.off[i]);
            ctx->light_list_info.push_back({(uint32_t)j, c, m.G, (uint32_t)(hl.ent.size() / 2), longest});
        }
    return upload(ctx, masks.data(), masks.size(), &ds.lmask);
}

// Reject anything the kernel's indexing does not expect, so that a malformed
// scene is an error code and never an out-of-bounds access on the GPU.
int validate(RtuContext* ctx, const RtuSceneDesc* s) {
    if (!s || !s->nodes || s->n_nodes == 0) return fail(ctx, RTU_ERR_ARG, "scene has no nodes");
    if (s->n_materials && !s->materials) return fail(ctx, RTU_ERR_ARG, "materials is NULL");
    if (s->n_lights && !s->lights) return fail(ctx, RTU_ERR_ARG, "lights is NULL");
    if (s->n_meshes && !s->meshes) return fail(ctx, RTU_ERR_ARG, "meshes is NULL");
    for (uint32_t i = 0; i < s->n_textures; i++) {
        const RtuTexture& t = s->textures[i];
        if (t.type != RTU_TEX_FILE && t.type != RTU_TEX_CHECKER) return fail(ctx, RTU_ERR_ARG, "texture %u: unknown type", i);
        if (t.type == RTU_TEX_FILE && (t.width < 0 || t.height < 0 || ((size_t)t.width * t.height > 0 && !t.rgb)))
            return fail(ctx, RTU_ERR_ARG, "texture %u: bad image", i);
    }
    auto map_ok = [&](const RtuTexMap& m) { return !m.present || m.texture < (int32_t)s->n_textures; };
    if (!map_ok(s->background_map) || !map_ok(s->environment_map)) return fail(ctx, RTU_ERR_ARG, "background/environment map: bad texture index");
    if (s->material_maps)
        for (uint32_t i = 0; i < s->n_materials * 4; i++)
            if (!map_ok(s->material_maps[i])) return fail(ctx, RTU_ERR_ARG, "material map %u: bad texture index", i);
    if (((s->background.has_map && !s->background.map_is_null) && !s->background_map.present) ||
        ((s->environment.has_map && !s->environment.map_is_null) && !s->environment_map.present))
        return fail(ctx, RTU_ERR_ARG, "textured background/environment without its texture map");
    for (uint32_t i = 0; i < s->n_lights; i++) {
        const RtuLight& l = s->lights[i];
        if (l.type < RTU_LIGHT_AMBIENT || l.type > RTU_LIGHT_POINT) return fail(ctx, RTU_ERR_ARG, "light %u: bad type", i);
    }
    uint32_t n_shadow = 0;
    for (uint32_t i = 0; i < s->n_lights; i++)
        if (s->lights[i].type != RTU_LIGHT_AMBIENT) n_shadow++;
    if (n_shadow > RTU_MAX_SHADOW_LIGHTS) return fail(ctx, RTU_ERR_UNSUPPORTED, "more than %d non-ambient lights", RTU_MAX_SHADOW_LIGHTS);
    if (s->n_materials > RTU_FI_MTL_MASK) return fail(ctx, RTU_ERR_UNSUPPORTED, "too many materials");
    for (uint32_t i = 0; i < s->n_nodes; i++) {
        const RtuNode& n = s->nodes[i];
        if (i == 0 ? n.parent != -1 : (n.parent < 0 || (uint32_t)n.parent >= i))
            return fail(ctx, RTU_ERR_ARG, "node %u: parent %d breaks pre-order", i, n.parent);
        int depth = i == 0 ? 0 : s->nodes[n.parent].depth + 1;
        if (n.depth != depth) return fail(ctx, RTU_ERR_ARG, "node %u: depth %d != %d", i, n.depth, depth);
        if (depth >= RTU_MAX_NODE_DEPTH) return fail(ctx, RTU_ERR_UNSUPPORTED, "node %u deeper than %d", i, RTU_MAX_NODE_DEPTH - 1);
        if (n.obj_type < RTU_OBJ_NONE || n.obj_type > RTU_OBJ_TRIMESH) return fail(ctx, RTU_ERR_ARG, "node %u: bad type", i);
        if (n.obj_type == RTU_OBJ_TRIMESH && (n.mesh_id < 0 || (uint32_t)n.mesh_id >= s->n_meshes))
            return fail(ctx, RTU_ERR_ARG, "node %u: bad mesh id", i);
        if (n.material_id >= (int)s->n_materials) return fail(ctx, RTU_ERR_ARG, "node %u: bad material id", i);
    }
    for (uint32_t mi = 0; mi < s->n_meshes; mi++) {
        const RtuMesh& m = s->meshes[mi];
        if (!m.v || !m.f || !m.vn || !m.fn || !m.bvh || !m.elements)
            return fail(ctx, RTU_ERR_ARG, "mesh %u: missing array (normals are required, objects.h:56)", mi);
        if (m.n_bvh_nodes < 2 || m.n_elements != m.nf || m.nf == 0) return fail(ctx, RTU_ERR_ARG, "mesh %u: empty", mi);
        if (m.n_bvh_nodes >= (1u << 28) || m.n_elements >= (1u << 28)) return fail(ctx, RTU_ERR_UNSUPPORTED, "mesh %u: more than 2^28 nodes/elements", mi);
        // the fast walk addresses its triangle records (64 B) and 4-wide nodes (128 B, fewer than triangles) with 32-bit byte offsets
        if (m.nf >= (1u << 26) || m.n_elements >= (1u << 26)) return fail(ctx, RTU_ERR_UNSUPPORTED, "mesh %u: more than 2^26 triangles", mi);
        if (m.bvh_depth > RTU_MAX_BVH_STACK) return fail(ctx, RTU_ERR_UNSUPPORTED, "mesh %u: BVH depth %u > %d", mi, m.bvh_depth, RTU_MAX_BVH_STACK);
        for (uint32_t i = 0; i < m.nf * 3; i++) {
            if (m.f[i] >= m.nv) return fail(ctx, RTU_ERR_ARG, "mesh %u: vertex index out of range", mi);
            if (m.fn[i] >= m.nvn) return fail(ctx, RTU_ERR_ARG, "mesh %u: normal index out of range", mi);
        }
        // texture coordinates are optional, but half a set or an index past nvt would be read on every accepted hit
        if ((m.vt != nullptr) != (m.ft != nullptr) || ((m.vt || m.ft) && m.nvt == 0) || (m.nvt != 0 && !m.vt))
            return fail(ctx, RTU_ERR_ARG, "mesh %u: texture vertices and texture faces must come together (nvt %u)", mi, m.nvt);
        if (m.ft)
            for (uint32_t i = 0; i < m.nf * 3; i++)
                if (m.ft[i] >= m.nvt) return fail(ctx, RTU_ERR_ARG, "mesh %u: texture-vertex index out of range", mi);
        for (uint32_t i = 0; i < m.n_elements; i++)
            if (m.elements[i] >= m.nf) return fail(ctx, RTU_ERR_ARG, "mesh %u: element out of range", mi);
        // every node reachable from the root must be well formed; children have larger
        // ids than their parent (cyBVH.h:242-251), which also rules out cycles
        std::vector<std::pair<uint32_t, uint32_t>> st;  // node, level
        st.push_back({1u, 1u});
        uint32_t depth = 0;
        while (!st.empty()) {
            auto [id, lvl] = st.back();
            st.pop_back();
            if (lvl > depth) depth = lvl;
            const RtuBvhNode& n = m.bvh[id];
            if (n.count == 0) {
                if (n.index <= id || n.index + 1 >= m.n_bvh_nodes) return fail(ctx, RTU_ERR_ARG, "mesh %u: bad child index at node %u", mi, id);
                st.push_back({n.index, lvl + 1});
                st.push_back({n.index + 1, lvl + 1});
            } else {
                if (n.count > 8 || n.index + n.count > m.n_elements) return fail(ctx, RTU_ERR_ARG, "mesh %u: bad leaf at node %u", mi, id);
            }
        }
        if (depth > m.bvh_depth) return fail(ctx, RTU_ERR_ARG, "mesh %u: bvh_depth %u understates the tree (%u)", mi, m.bvh_depth, depth);
    }
    return RTU_OK;
}

// background.Sample / environment.SampleEnvironment for the supported cases
// (scene.h:421-431): untextured -> colour; TextureMap(NULL) -> colour * black.
void env_value(const RtuEnvColor& e, float out[3]) {
    for (int k = 0; k < 3; k++) out[k] = e.has_map ? e.color[k] * 0.0f : e.color[k];
}

int shard_bands(int height, int rank, int count) {
    int nb = (height + RTU_BAND_ROWS - 1) / RTU_BAND_ROWS;
    if (rank >= nb) return 0;
    return (nb - rank + count - 1) / count;
}

int check_frame(RtuContext* ctx, const RtuFrameDesc* f) {
    if (!f) return fail(ctx, RTU_ERR_ARG, "frame is NULL");
    if (f->width <= 0 || f->height <= 0 || f->width > 65536 || f->height > 65536) return fail(ctx, RTU_ERR_ARG, "bad resolution");
    if (f->shard_count < 1 || f->shard_rank < 0 || f->shard_rank >= f->shard_count) return fail(ctx, RTU_ERR_ARG, "bad shard");
    if (f->max_bounce < 0 || f->max_bounce > RTU_MAX_BOUNCE) return fail(ctx, RTU_ERR_ARG, "max_bounce out of range");
    if (f->samples < 0 || f->samples > 65536) return fail(ctx, RTU_ERR_ARG, "samples out of range");
    if (f->gather_bounces != 0 && (f->gather_bounces != RTU_GI_BOUNCES || f->samples < 1))
        return fail(ctx, RTU_ERR_ARG, "gather_bounces is 0 or %d (recipe P, with samples >= 1)", RTU_GI_BOUNCES);
    if (f->collect_stats < 0 || f->collect_stats > 2) return fail(ctx, RTU_ERR_ARG, "collect_stats is 0, 1 or 2");
    if (f->samples == 0 && ctx->has_scene && (ctx->scene_stochastic || f->dof != 0))
        return fail(ctx, RTU_ERR_STOCHASTIC, "the scene has %s: render it with frame.samples >= 1 (recipe S)",
                    ctx->scene_stochastic ? ctx->stochastic_what.c_str() : "depth of field");
    return RTU_OK;
}

void free_levels(RtuContext* ctx) {
    for (void* p : ctx->level_allocs) (void)hipFree(p);
    ctx->level_allocs.clear();
    memset(ctx->lv, 0, sizeof ctx->lv);
    memset(ctx->lv_side, 0, sizeof ctx->lv_side);
    ctx->defer_list0 = nullptr;
    ctx->defer_cap0_s = 0;
    ctx->level_cap0 = 0;
}

template <class T>
int alloc_level(RtuContext* ctx, T** dst, size_t count) {
    void* d = nullptr;
    RTU_HIP(ctx, hipMalloc(&d, sizeof(T) * (count ? count : 1)));
    ctx->level_allocs.push_back(d);
    *dst = static_cast<T*>(d);
    return RTU_OK;
}

// Frame arrays of every recursion level (rtu_device.h). Level 0 holds at most one frame per
// pixel; a deeper level starts with the same capacity — a QUARTER of it in launches of more than 16 M pixels (batches of frames) —
// and is grown to what an overflowed frame reported (check_overflow): a frame can hold up to 3^L frames per pixel at level L in
// theory, a tenth of a frame per pixel in the reference's scenes. (Round 3: every level as large as level 0 was 80 GB per context
// with 32 frames of 1920 x 1080 in flight — a fourth context on one GPU ran out of memory; now 30 GB.)
int ensure_levels(RtuContext* ctx, uint32_t pixels, uint32_t n_tiles, bool gi = false) {
    if (gi) ctx->want_gi = true;
    // one shard of level 0 receives the frames of every RTU_SHARDS-th 8x8 tile of the launch (ragged right /
    // bottom tiles included), so level 0 cannot overflow
    size_t tiles = n_tiles;
    size_t cap_s0 = ((tiles + RTU_SHARDS - 1) / RTU_SHARDS) * 64;
    size_t want[RTU_MAX_LEVELS];
    size_t maxcap = cap_s0;
    bool fits = ctx->level_nsl == ctx->nsl && (!ctx->textured || ctx->lv[0].fuv) && (!ctx->want_gi || ctx->lv[0].famb);
    const size_t cap_deep = cap_s0 * RTU_SHARDS > ((size_t)16 << 20) ? ((cap_s0 / 4 + 63) / 64) * 64 : cap_s0;
    for (int L = 0; L < RTU_MAX_LEVELS; L++) {
        want[L] = L == 0 ? cap_s0 : std::max<size_t>(cap_deep, ctx->want_cap_s[L]);
        if (want[L] > maxcap) maxcap = want[L];
        fits = fits && ctx->lv[L].cap_s >= want[L];
    }
    // defer list: at most every ray of the largest phase (all slots of the largest level)
    size_t dcap_s = maxcap * (ctx->nsl + 3);
    if (ctx->want_defer_s > dcap_s) dcap_s = ctx->want_defer_s;
    if (fits && ctx->defer_cap_s >= dcap_s && ctx->defer_list0 && ctx->defer_cap0_s >= cap_s0) return RTU_OK;
    // GROW ONLY: a caller that alternates launch shapes (a batch of 24 frames, then the last 8 of its run: the smaller one wants MORE at
    // the deep levels — it is below the quarter rule's 16 M pixels — and less at level 0) must not make the arrays swing between the two
    // (found as a 2.4 s stall per timed region: every launch freed and allocated 20 GB). Every capacity is the larger of what it was and
    // what is wanted now.
    size_t cap_s0_alloc = cap_s0 > ctx->defer_cap0_s ? cap_s0 : ctx->defer_cap0_s;
    for (int L = 0; L < RTU_MAX_LEVELS; L++) {
        if (ctx->lv[L].cap_s > want[L]) want[L] = ctx->lv[L].cap_s;
        if (want[L] > maxcap) maxcap = want[L];
    }
    if (maxcap * (ctx->nsl + 3) > dcap_s) dcap_s = maxcap * (ctx->nsl + 3);
    if (ctx->defer_cap_s > dcap_s) dcap_s = ctx->defer_cap_s;
    free_levels(ctx);
    int rc;
    size_t total = 0;
    for (int L = 0; L < RTU_MAX_LEVELS; L++) total += want[L] * RTU_SHARDS * (size_t)(16 * 11 + 4 * (ctx->nsl ? ctx->nsl : 1) + 12);
    if (total > ((size_t)160 << 30)) return fail(ctx, RTU_ERR_CAPACITY, "the recursion of this frame needs %zu GB of frame records", total >> 30);
    for (int L = 0; L < RTU_MAX_LEVELS; L++) {
        LevelBuffers& lv = ctx->lv[L];
        size_t cap_s = want[L];
        size_t cap = cap_s * RTU_SHARDS;
        if (cap > 0x0FFFFFF0u) return fail(ctx, RTU_ERR_CAPACITY, "more than 2^28 frames in one recursion level");
        if ((rc = alloc_level(ctx, &lv.fa, cap)) != RTU_OK) return rc;
        if ((rc = alloc_level(ctx, &lv.fb, cap)) != RTU_OK) return rc;
        if ((rc = alloc_level(ctx, &lv.fc, cap)) != RTU_OK) return rc;
        if ((rc = alloc_level(ctx, &lv.fres, cap)) != RTU_OK) return rc;
        if ((rc = alloc_level(ctx, &lv.fchild, cap)) != RTU_OK) return rc;
        if ((rc = alloc_level(ctx, &lv.fsh, cap * (ctx->nsl ? ctx->nsl : 1))) != RTU_OK) return rc;
        if ((rc = alloc_level(ctx, &lv.fslot, cap * 6)) != RTU_OK) return rc;
        if ((rc = alloc_level(ctx, &lv.fpend, cap)) != RTU_OK) return rc;
        if (ctx->want_gi && (rc = alloc_level(ctx, &lv.famb, cap)) != RTU_OK) return rc;
        if (ctx->textured) {
            if ((rc = alloc_level(ctx, &lv.fuv, cap)) != RTU_OK) return rc;
            if ((rc = alloc_level(ctx, &lv.fsuv, cap * 3)) != RTU_OK) return rc;
        }
        if ((rc = alloc_level(ctx, &lv.lmain, cap)) != RTU_OK) return rc;
        if ((rc = alloc_level(ctx, &lv.lrefl, cap)) != RTU_OK) return rc;
        lv.cap_s = (uint32_t)cap_s;
    }
    if ((rc = alloc_level(ctx, &ctx->defer_list, dcap_s * RTU_SHARDS)) != RTU_OK) return rc;
    ctx->defer_cap_s = (uint32_t)dcap_s;
    // the primary phase's own defer list (at most every pixel of the launch) and the side set of level arrays (rtu_device.h
    // KernelArgs::fcnt0): small — a k_tail launch refuses more than RTU_TAIL_DECLINE frames anyway
    if ((rc = alloc_level(ctx, &ctx->defer_list0, cap_s0_alloc * RTU_SHARDS)) != RTU_OK) return rc;
    ctx->defer_cap0_s = (uint32_t)cap_s0_alloc;
    for (int L = 0; L < RTU_MAX_LEVELS; L++) {
        LevelBuffers& lv = ctx->lv_side[L];
        const size_t cap_s = 256, cap = cap_s * RTU_SHARDS;
        if ((rc = alloc_level(ctx, &lv.fa, cap)) != RTU_OK) return rc;
        if ((rc = alloc_level(ctx, &lv.fb, cap)) != RTU_OK) return rc;
        if ((rc = alloc_level(ctx, &lv.fc, cap)) != RTU_OK) return rc;
        if ((rc = alloc_level(ctx, &lv.fres, cap)) != RTU_OK) return rc;
        if ((rc = alloc_level(ctx, &lv.fchild, cap)) != RTU_OK) return rc;
        if ((rc = alloc_level(ctx, &lv.fsh, cap * (ctx->nsl ? ctx->nsl : 1))) != RTU_OK) return rc;
        if ((rc = alloc_level(ctx, &lv.fslot, cap * 6)) != RTU_OK) return rc;
        if ((rc = alloc_level(ctx, &lv.fpend, cap)) != RTU_OK) return rc;
        if (ctx->textured) {
            if ((rc = alloc_level(ctx, &lv.fuv, cap)) != RTU_OK) return rc;
            if ((rc = alloc_level(ctx, &lv.fsuv, cap * 3)) != RTU_OK) return rc;
        }
        if ((rc = alloc_level(ctx, &lv.lmain, cap)) != RTU_OK) return rc;
        if ((rc = alloc_level(ctx, &lv.lrefl, cap)) != RTU_OK) return rc;
        lv.cap_s = (uint32_t)cap_s;
    }
    ctx->level_cap0 = pixels;
    ctx->level_nsl = ctx->nsl;
    return RTU_OK;
}

// Halton (scene.h:130-139)
float halton(int index, int base) {
    float r = 0;
    float f = 1.0f / (float)base;
    for (int i = index; i > 0; i /= base) {
        r += f * (float)(i % base);
        f /= (float)base;
    }
    return r;
}

// One launch sequence: the whole frame of recipe W, or samples [sample_index, sample_index + batch) of
// recipe S into d_out as [sample][pixel of the shard].
// frames_batch: `batch` frames of recipe W with their own cameras (frame == &frames_batch[0]).
// gi_mode RTU_LAUNCH_CHAIN / RTU_LAUNCH_SHADE: one step of recipe P at chain depth gi_depth (render_sampled).
int launch(RtuContext* ctx, const RtuFrameDesc* frame, float4* d_out, hipStream_t stream, bool zero_counters, int sample_index = 0, int batch = 1,
           const RtuFrameDesc* frames_batch = nullptr, int gi_mode = RTU_LAUNCH_ALL, int gi_depth = 0) {
    uint32_t tiles_x = (uint32_t)((frame->width + 7) / 8);
    uint32_t bands = (uint32_t)shard_bands(frame->height, frame->shard_rank, frame->shard_count);
    uint32_t n_tiles = tiles_x * bands * (uint32_t)batch;
    uint32_t pixels = (uint32_t)rtu_shard_rows(frame) * (uint32_t)frame->width;
    const bool gi = gi_mode != RTU_LAUNCH_ALL;
    // recipe P: every chain hit is the root of two Shade() trees
    int rc = ensure_levels(ctx, pixels * (uint32_t)batch, gi ? 2u * (n_tiles + RTU_SHARDS) : n_tiles, gi);
    if (rc != RTU_OK) return rc;
    if (gi) {
        const size_t chains = (size_t)pixels * (size_t)batch;
        if (chains > ctx->gi_chains) {
            if (ctx->gi_h) (void)hipFree(ctx->gi_h);
            if (ctx->gi_res) (void)hipFree(ctx->gi_res);
            ctx->gi_h = ctx->gi_res = nullptr;
            ctx->gi_chains = 0;
            RTU_HIP(ctx, hipMalloc((void**)&ctx->gi_h, chains * (RTU_GI_BOUNCES + 1) * 4 * sizeof(float4)));
            RTU_HIP(ctx, hipMalloc((void**)&ctx->gi_res, chains * 2 * sizeof(float4)));
            ctx->gi_chains = chains;
        }
    }
    const int stats = frame->collect_stats;  // 0 fast, 1 reference counting, 2 touched bytes of the fast variant
    if (stats && zero_counters) {
        RTU_HIP(ctx, hipMemsetAsync(ctx->counters, 0, kCounterBytes, stream));
        memset(ctx->slot_launches, 0, sizeof ctx->slot_launches);
    }
    // the append counters start at zero; `overflow` is STICKY — launches only ever set it, check_overflow reads and clears
    // it — so that a frame that ran out of capacity is reported even when later launch sequences were queued behind it
    RTU_HIP(ctx, hipMemsetAsync(ctx->fcnt, 0, offsetof(FrameCounters, overflow), stream));
    KernelArgs a;
    memset(&a, 0, sizeof a);
    a.scene = ctx->dscene;
    a.frame = *frame;
    if (!ctx->any_recursive_material) a.frame.max_bounce = 0;  // no reflection/refraction anywhere: Shade() never recurses
    a.out = d_out;
    memcpy(a.lv, ctx->lv, sizeof a.lv);
    a.fcnt = ctx->fcnt;
    a.tl = ctx->stamp_next ? ctx->tl : nullptr;
    a.defer_list = ctx->defer_list;
    a.defer_cap_s = ctx->defer_cap_s;
    a.defer_list0 = ctx->defer_list0;
    a.defer_cap0_s = ctx->defer_cap0_s;
    a.fcnt0 = ctx->fcnt;
    a.dbg = ctx->dbg;
    a.scene.dbg = ctx->dbg;
    a.counters = stats ? ctx->counters : nullptr;
    a.host_launches = stats == 2 ? ctx->slot_launches : nullptr;
    a.node_rects = (stats != 1 && frame->samples == 0 && ctx->dscene.node_bounds) ? ctx->node_rects : nullptr;
    if (a.node_rects && (ctx->dscene.n_cover + ctx->dscene.n_pcover) && !gi && ((size_t)((frame->width + 7) / 8) * (size_t)((frame->height + 7) / 8) + 31u) / 32u <= 12288u) {  // (the mask has to fit k_mesh_cover's LDS copy)
        a.tiles_xf = (uint32_t)((frame->width + 7) / 8);
        a.cover_words = (a.tiles_xf * (uint32_t)((frame->height + 7) / 8) + 31u) / 32u;
        a.cover_faces = ctx->cover_faces;
        const size_t need = (size_t)batch * (ctx->dscene.n_cover + ctx->dscene.n_pcover) * (1u + a.cover_words);
        if (need > ctx->cover_cap) {
            RTU_HIP(ctx, hipStreamSynchronize(stream));  // (first launch at this size only) nothing may still read the old masks
            if (ctx->cover) (void)hipFree(ctx->cover);
            ctx->cover = nullptr;
            ctx->cover_cap = 0;
            RTU_HIP(ctx, hipMalloc((void**)&ctx->cover, need * sizeof(uint32_t)));
            ctx->cover_cap = need;
        }
        RTU_HIP(ctx, hipMemsetAsync(ctx->cover, 0, need * sizeof(uint32_t), stream));
        a.cover = ctx->cover;
    }
    if (a.node_rects && !gi && ctx->dscene.n_nodes <= 64u && stats != 1 && !(ctx->dbg & 256u)) {  // tile occupancy: every word is written by k_tile_occ on this launch
        a.occ_words = ((tiles_x * bands + 63u) / 64u) * 2u;
        const size_t need = (size_t)batch * a.occ_words;
        if (need > ctx->occ_cap) {
            RTU_HIP(ctx, hipStreamSynchronize(stream));  // (first launch at this size only) nothing may still read the old words
            if (ctx->occ) (void)hipFree(ctx->occ);
            ctx->occ = nullptr;
            ctx->occ_cap = 0;
            RTU_HIP(ctx, hipMalloc((void**)&ctx->occ, need * sizeof(uint32_t)));
            ctx->occ_cap = need;
        }
        a.occ = a.occ_words ? ctx->occ : nullptr;  // (a shard without rows has no tiles: nothing is launched at all)
    }
    a.tiles_x = tiles_x;
    a.tiles_per_image = tiles_x * bands;
    a.nsl = ctx->nsl;
    a.n_meshes = ctx->n_meshes;
    // the cut level for k_tail: what a launch of the same shape showed last time (a hint: any value renders the same image, a
    // wrong one is refused on the device and reported like an overflow); rtu_debug_tail_from overrides it once
    const uint64_t tail_key = ((uint64_t)n_tiles << 8) | (uint64_t)((frame->samples ? 2 : 0) | (frames_batch ? 4 : 0) | (gi ? 8 : 0));
    int hint = RTU_MAX_LEVELS;
    bool forced = false;
    if (ctx->tail_hint != 0) { hint = ctx->tail_hint; ctx->tail_hint = 0; forced = true; }
    else if (ctx->tail_hints.count(tail_key)) hint = ctx->tail_hints[tail_key];
    a.tail_from = stats == 1 ? RTU_MAX_LEVELS : hint;
    ctx->last_tail_key = tail_key;
    // (not for recipe P: its chain and shading launches share one shape and have lists of very different lengths)
    if (stats == 0 && !gi && !(ctx->dbg & 512u) && ctx->list_hints.count(tail_key)) memcpy(a.list_n, ctx->list_hints[tail_key].data(), sizeof a.list_n);
    if (forced) a.dbg |= 128u;  // a cut level set by the test hook is taken as it is (k_tail does not refuse it)
    if (frame->samples >= 1) {
        const float pixelIncrement = (float)(1.0 / frame->samples);  // RenderFunctions.cpp:68
        a.sampling = 1;
        a.sample_index = (uint32_t)sample_index;
        a.batch = (uint32_t)batch;
        a.batch_pixels = pixels;
        a.tiles_per_image = tiles_x * bands;
        for (int b = 0; b < batch; b++) {
            const int index = sample_index + b;
            const float currentOffset = (float)index * pixelIncrement;  // :80
            a.pix_off_x[b] = currentOffset + halton(index, 4);          // :84, :96
            a.pix_off_y[b] = currentOffset + halton(index, 5);          // :85, :96
        }
    }
    if (frames_batch) {
        a.frame_batch = 1;
        a.batch = (uint32_t)batch;
        a.batch_pixels = pixels;
        a.tiles_per_image = tiles_x * bands;
        if (!ctx->d_cams) {
            RTU_HIP(ctx, hipMalloc((void**)&ctx->d_cams, sizeof(BatchCam) * RTU_MAX_FRAME_BATCH));
            RTU_HIP(ctx, hipHostMalloc((void**)&ctx->h_cams, sizeof(BatchCam) * RTU_MAX_FRAME_BATCH * RtuContext::kCamSlots, hipHostMallocDefault));
            for (hipEvent_t& e : ctx->cam_ev) RTU_HIP(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
        }
        const int slot = ctx->cam_slot;
        ctx->cam_slot = (slot + 1) % RtuContext::kCamSlots;
        RTU_HIP(ctx, hipEventSynchronize(ctx->cam_ev[slot]));  // the copy that last read this slot (16 launches ago) is done; a fresh event is "done"
        BatchCam* hc = ctx->h_cams + (size_t)slot * RTU_MAX_FRAME_BATCH;
        for (int b = 0; b < batch; b++) {
            memcpy(hc[b].pos, frames_batch[b].cam_pos, sizeof hc[b].pos);
            memcpy(hc[b].origin, frames_batch[b].origin, sizeof hc[b].origin);
            memcpy(hc[b].u, frames_batch[b].u, sizeof hc[b].u);
            memcpy(hc[b].v, frames_batch[b].v, sizeof hc[b].v);
        }
        RTU_HIP(ctx, hipMemcpyAsync(ctx->d_cams, hc, sizeof(BatchCam) * (size_t)batch, hipMemcpyHostToDevice, stream));
        RTU_HIP(ctx, hipEventRecord(ctx->cam_ev[slot], stream));
        a.cam = ctx->d_cams;
    }
    if (gi) {
        a.gi_h = ctx->gi_h;
        a.gi_res = ctx->gi_res;
        a.gi_depth = (uint32_t)gi_depth;
        a.gi_total = pixels * (uint32_t)batch;
    }
    // side mode (rtu_device.h KernelArgs::fcnt0): recipe W's fast variant on a scene with meshes, unless this launch shape has shown that
    // its stage 2 makes more frames than a k_tail launch takes (rtu_debug_flags 8192: never — results must not change)
    // Only where it pays and cannot surprise: every mesh node's material is childless (stage 2's frames are then the rare hits whose
    // shadow rays could not be settled inline — a mirror teapot would send every hit through the k_tail launch), and the last launch of
    // this shape deferred enough primary rays for the one-lane-per-ray stage 2 (a first launch, or a short list: the old order).
    ctx->last_side = false;
    const uint32_t thr0 = (uint32_t)(frame->coop_threshold > 0 ? frame->coop_threshold : 70000);
    if (stats != 1 && !gi && frame->samples == 0 && ctx->n_meshes > 0 && ctx->mesh_hits_childless && !ctx->stamp_next && !(ctx->dbg & (8192u | 2048u | 64u)) &&
        ctx->dscene.node_bounds && ctx->dscene.lmask &&  // (the occluder lists are what settles a mesh hit's shadow rays inline)
        !ctx->side_off.count(tail_key) && ctx->list_hints.count(tail_key) && ctx->list_hints[tail_key][0] - 1u > thr0 &&
        ctx->side_frames.count(tail_key) && ctx->side_frames[tail_key] <= 256u) {
        if (!ctx->aux_stream) {
            // created at first use: the helper stream of side mode, lowest priority. (A context that never uses side mode creates no stream
            // for it: HIP deals its hardware queues out in creation order — four by default, GPU_MAX_HW_QUEUES —, and a stream too many makes
            // two streams that are meant to overlap share a queue: two contexts alternating, 56.8 -> 47.5 Grays/s, measured.)
            int prio_least = 0, prio_greatest = 0;
            RTU_HIP(ctx, hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest));
            RTU_HIP(ctx, hipStreamCreateWithPriority(&ctx->aux_stream, hipStreamNonBlocking, prio_least));
            RTU_HIP(ctx, hipEventCreateWithFlags(&ctx->aux_ev0, hipEventDisableTiming));
            RTU_HIP(ctx, hipEventCreateWithFlags(&ctx->aux_ev1, hipEventDisableTiming));
        }
        a.side = 1;
        a.fcnt0 = ctx->fcnt_side;
        memcpy(a.lv_side, ctx->lv_side, sizeof a.lv_side);
        a.aux_stream = ctx->aux_stream;
        a.aux_ev0 = ctx->aux_ev0;
        a.aux_ev1 = ctx->aux_ev1;
        RTU_HIP(ctx, hipMemsetAsync(ctx->fcnt_side, 0, offsetof(FrameCounters, overflow), stream));  // (ahead of k_primary, which fills its defer counters)
        ctx->last_side = true;
    }
    // k_primary's grid (render_impl.h launch_all): few, long-lived workgroups when the last launch of this shape found most tiles empty
    a.pgrid = 32768u;
    if (a.occ && !(ctx->dbg & 512u) && ctx->occ_hints.count(tail_key) && (uint64_t)ctx->occ_hints[tail_key] * 3u < (uint64_t)n_tiles)
        // (alone: four rounds of resident wavefronts balance themselves; beside another sequence — rtu_set_sequences_in_flight —: ONE resident
        // set, which the other sequence's kernels fill in around. Two sequences in flight: 2048 / 1536 / 1024 / 768 workgroups: 59.4 / 61.4 /
        // 62.5 / 61.4 Grays/s; three: 60 - 61 whatever the grid.)
        a.pgrid = ctx->sequences_in_flight >= 2 ? 1024u : 4096u;
    ctx->last_tail_from = a.tail_from;
    ctx->last_stats = stats == 1;
    memcpy(a.shadow_light, ctx->shadow_light, sizeof a.shadow_light);
    memcpy(a.nol_light, ctx->nol_light, sizeof a.nol_light);
    int probe_recorded = 0;
    LaunchProbe probe{-1, nullptr, nullptr, &probe_recorded};
    const bool probing = ctx->probe_slot >= 0 && ctx->probe_used < RtuContext::kProbePairs;
    if (probing) {
        probe.slot = ctx->probe_slot;
        probe.ev0 = ctx->probe_ev[2 * ctx->probe_used];
        probe.ev1 = ctx->probe_ev[2 * ctx->probe_used + 1];
    }
    if (gi_mode == RTU_LAUNCH_SHADE && gi_depth == 0) {
        hipError_t e0 = (hipError_t)rtu_launch_frame(a, n_tiles, ctx->bvh_stack_needed, stats, stream, gi_mode, probing ? &probe : nullptr);
        if (probing && probe_recorded) ctx->probe_used++;
        if (e0 != hipSuccess) return fail(ctx, RTU_ERR_HIP, "kernel launch: %s", hipGetErrorString(e0));
        e0 = (hipError_t)rtu_launch_gi_final(a, stream);  // harmless if this step has to be repeated: it only reads the results
        if (e0 != hipSuccess) return fail(ctx, RTU_ERR_HIP, "kernel launch: %s", hipGetErrorString(e0));
        return RTU_OK;
    }
    hipError_t e = (hipError_t)rtu_launch_frame(a, n_tiles, ctx->bvh_stack_needed, stats, stream, gi_mode, probing ? &probe : nullptr);
    if (probing && probe_recorded) ctx->probe_used++;
    if (e != hipSuccess) return fail(ctx, RTU_ERR_HIP, "kernel launch: %s", hipGetErrorString(e));
    return RTU_OK;
}

// After the stream has drained: did any recursion level run out of frame capacity?
// Frames per level of the frame just finished -> where k_tail may take over in the next one.
uint64_t stage2_total(const FrameCounters& h) {
    uint64_t n = 0;
    for (int s = 0; s < RTU_SHARDS; s++) n += h.stage2_frames[s * RTU_CSTRIDE];
    return n;
}

void learn_tail(RtuContext* ctx, const FrameCounters& h, const FrameCounters* side) {
    static const uint32_t kTailEnv = [] { const char* e = getenv("RTU_TAIL_LEARN"); return e ? (uint32_t)strtoul(e, nullptr, 10) : 0u; }();  // tuning knob
    const uint32_t kTailMax = kTailEnv ? kTailEnv : RTU_TAIL_LEARN;  // frames of the cut level, one wavefront each: measured, a few thousand subtrees evaluated
                                               // wavefront by wavefront are slower than their levels kernel by kernel
    uint32_t frames[RTU_MAX_LEVELS];
    for (int L = 0; L < RTU_MAX_LEVELS; L++) {
        frames[L] = 0;
        for (int s = 0; s < RTU_SHARDS; s++) frames[L] += h.n_frames[L][(s) * RTU_CSTRIDE];
    }
    const int used = ctx->last_tail_from;  // levels > used were not materialised: their counts are unknown (zero)
    const int top = used < RTU_MAX_LEVELS ? used : RTU_MAX_LEVELS - 1;
    int hint = RTU_MAX_LEVELS;
    for (int L = 1; L <= top; L++)
        if (frames[L] <= kTailMax) { hint = L; break; }
    // the launch had a tail and its cut level was not small after all: deeper counts are unknown, learn them from a launch without
    ctx->tail_hints[ctx->last_tail_key] = hint;
    // the rays deferred in every phase (+ 1): what the next launch of this shape sizes its idle stage-2 kernels by
    std::array<uint32_t, 8> lh{};
    for (int p = 0; p <= RTU_MAX_LEVELS && p < 8; p++) {
        uint64_t n = 0;
        const FrameCounters& from = (p == 0 && side) ? *side : h;  // (side mode: the primary phase counts in the side counters)
        for (int s = 0; s < RTU_SHARDS; s++) n += from.n_defer[p][(s) * RTU_CSTRIDE];
        lh[p] = (uint32_t)(n < 0xFFFFFFF0ull ? n : 0xFFFFFFF0ull) + 1u;
    }
    ctx->list_hints[ctx->last_tail_key] = lh;
}

// The append counters keep counting past the capacity, so an overflowed frame tells how much its
// first overflowing level really needs (deeper levels may need another round: their parents were
// dropped). Wanted capacities grow to the reported counts (+25 %, at least x2 for the level below).
int check_overflow(RtuContext* ctx, bool* overflow) {
    std::unique_ptr<FrameCounters> hp(new FrameCounters);  // a quarter of a megabyte: not on the stack
    FrameCounters& h = *hp;
    RTU_HIP(ctx, hipMemcpy(&h, ctx->fcnt, sizeof h, hipMemcpyDeviceToHost));
    // side mode: stage 2 of the primary phase made more frames than its k_tail launch takes (or than the side arrays hold): the
    // frames of this report are incomplete, and this launch shape goes without side mode from now on
    std::unique_ptr<FrameCounters> sp;
    bool side_failed = false;
    {
        uint32_t flags[2] = {0, 0};
        RTU_HIP(ctx, hipMemcpy(flags, &ctx->fcnt_side->overflow, sizeof flags, hipMemcpyDeviceToHost));
        if (flags[0] || flags[1]) {
            side_failed = true;
            ctx->side_off[ctx->last_tail_key] = true;
            RTU_HIP(ctx, hipMemset(&ctx->fcnt_side->overflow, 0, 2 * sizeof(uint32_t)));
        }
        if (ctx->last_side) {
            sp.reset(new FrameCounters);
            RTU_HIP(ctx, hipMemcpy(sp.get(), ctx->fcnt_side, sizeof(FrameCounters), hipMemcpyDeviceToHost));
            ctx->side_frames[ctx->last_tail_key] = stage2_total(*sp);
            static const bool kSideVerbose = getenv("RTU_SIDE_VERBOSE") != nullptr;  // diagnostics: what side mode's stage 2 made, per status check
            if (kSideVerbose) {
                unsigned long long f[RTU_MAX_LEVELS] = {}, d0 = 0;
                for (int L = 0; L < RTU_MAX_LEVELS; L++)
                    for (int s = 0; s < RTU_SHARDS; s++) f[L] += sp->n_frames[L][s * RTU_CSTRIDE];
                for (int s = 0; s < RTU_SHARDS; s++) d0 += sp->n_defer[0][s * RTU_CSTRIDE];
                fprintf(stderr, "[side] deferred pixels %llu, side frames per level %llu %llu %llu %llu %llu %llu, failed %d\n", d0, f[0], f[1], f[2], f[3], f[4], f[5], (int)side_failed);
            }
        }
    }
    {
        uint64_t occ = 0;
        for (int s = 0; s < RTU_SHARDS; s++) occ += h.occ_tiles[s * RTU_CSTRIDE];
        if (!ctx->last_stats) ctx->occ_hints[ctx->last_tail_key] = (uint32_t)(occ < 0xFFFFFFFFull ? occ : 0xFFFFFFFFull);
    }
    if (!ctx->last_side && !ctx->last_stats) ctx->side_frames[ctx->last_tail_key] = stage2_total(h);  // (the fast variant without side mode: counted in the main counters)
    *overflow = h.overflow != 0 || h.tail_declined != 0 || side_failed;
    if (!*overflow) {
        learn_tail(ctx, h, sp.get());
        return RTU_OK;
    }
    RTU_HIP(ctx, hipMemset(&ctx->fcnt->overflow, 0, 2 * sizeof(uint32_t)));  // reported: the next status starts clean (overflow, tail_declined)
    if (h.tail_declined) ctx->tail_hints[ctx->last_tail_key] = RTU_MAX_LEVELS;  // this shape is rendered level by level from now on
    if (!h.overflow) return RTU_OK;  // nothing ran out of capacity: render again, that is all
    bool grew = false;
    for (int L = 1; L < RTU_MAX_LEVELS; L++) {
        uint32_t need = 0;
        for (int s = 0; s < RTU_SHARDS; s++) need = h.n_frames[L][(s) * RTU_CSTRIDE] > need ? h.n_frames[L][(s) * RTU_CSTRIDE] : need;
        if (need > ctx->lv[L].cap_s) {
            size_t w = ((size_t)need + need / 4 + 63) / 64 * 64;
            if (w > ctx->want_cap_s[L]) ctx->want_cap_s[L] = (uint32_t)w;
            // its children were not all created: start the next level at least as large
            if (L + 1 < RTU_MAX_LEVELS && ctx->want_cap_s[L + 1] < w && ctx->lv[L + 1].cap_s < w) ctx->want_cap_s[L + 1] = (uint32_t)w;
            grew = true;
        }
    }
    uint32_t dneed = 0;
    for (int p = 0; p <= RTU_MAX_LEVELS; p++)
        for (int s = 0; s < RTU_SHARDS; s++) dneed = h.n_defer[p][(s) * RTU_CSTRIDE] > dneed ? h.n_defer[p][(s) * RTU_CSTRIDE] : dneed;
    if (dneed > ctx->defer_cap_s) {
        ctx->want_defer_s = dneed + dneed / 4;
        grew = true;
    }
    if (!grew) {  // the overflow was an EARLIER launch sequence's (the counts are the last one's): grow every level
        for (int L = 1; L < RTU_MAX_LEVELS; L++) ctx->want_cap_s[L] = ctx->lv[L].cap_s * 2;
    }
    return RTU_OK;
}

// Recipe S: one launch sequence per batch of samples (as many as fit 2^25 pixels, at most RTU_MAX_BATCH),
// each checked for frame-capacity overflow before its images are added to the accumulators in sample
// order; the mean goes to d_out. Synchronises per batch.
int render_sampled(RtuContext* ctx, const RtuFrameDesc* frame, float4* d_out, hipStream_t stream, bool zero_counters) {
    const size_t pixels = (size_t)rtu_shard_rows(frame) * (size_t)frame->width;
    if (pixels == 0) return RTU_OK;
    const bool gi = frame->gather_bounces != 0;
    static const int kGiLog2 = [] { const char* e = getenv("RTU_GI_BATCH_LOG2"); return e ? atoi(e) : 25; }();  // tuning knob (23 / 24 / 25: 138.2 / 131.9 / 130.6 ms for config 5 at 64 spp)
    int batch = (int)(((size_t)1 << (gi ? kGiLog2 : 25)) / pixels);  // recipe P keeps 22 float4 per chain and two roots per chain hit (a larger batch buys nothing: measured)
    if (batch > RTU_MAX_BATCH) batch = RTU_MAX_BATCH;
    if (batch > frame->samples) batch = frame->samples;
    if (batch < 1) batch = 1;
    if (pixels > ctx->acc_pixels) {
        if (ctx->acc) (void)hipFree(ctx->acc);
        if (ctx->acc_hits) (void)hipFree(ctx->acc_hits);
        ctx->acc = nullptr;
        ctx->acc_hits = nullptr;
        ctx->acc_pixels = 0;
        RTU_HIP(ctx, hipMalloc((void**)&ctx->acc, pixels * sizeof(float4)));
        RTU_HIP(ctx, hipMalloc((void**)&ctx->acc_hits, pixels * sizeof(uint32_t)));
        ctx->acc_pixels = pixels;
    }
    if (pixels * (size_t)batch > ctx->sample_buf_pixels) {
        if (ctx->sample_buf) (void)hipFree(ctx->sample_buf);
        ctx->sample_buf = nullptr;
        ctx->sample_buf_pixels = 0;
        RTU_HIP(ctx, hipMalloc((void**)&ctx->sample_buf, pixels * (size_t)batch * sizeof(float4)));
        ctx->sample_buf_pixels = pixels * (size_t)batch;
    }
    int rounds = 0;
    for (int i = 0; i < frame->samples; i += batch) {
        if (ctx->cancel && *ctx->cancel) return fail(ctx, RTU_ERR_CANCELLED, "cancelled after %d of %d samples", i, frame->samples);  // StopRender(), main.cpp:70-72
        int nb = frame->samples - i < batch ? frame->samples - i : batch;
        while (!gi) {
            nb = frame->samples - i < batch ? frame->samples - i : batch;  // (i may have been reset below)
            int rc = launch(ctx, frame, ctx->sample_buf, stream, zero_counters && i == 0, i, nb);
            if (rc != RTU_OK) return rc;
            RTU_HIP(ctx, hipStreamSynchronize(stream));
            bool overflow = false;
            if ((rc = check_overflow(ctx, &overflow)) != RTU_OK) return rc;
            if (!overflow) break;
            if (++rounds > 4 * RTU_MAX_LEVELS) return fail(ctx, RTU_ERR_CAPACITY, "recursion frames still exceed the capacity after %d rounds", rounds);
            if (frame->collect_stats) { i = 0; zero_counters = true; }  // the counters of the dropped pass are in the totals: start again
        }
        if (gi) {
            // recipe P: the chain of gather rays first (depth 0 = the primary ray), then the Shade() trees from the
            // deepest hit up — each depth's AmbientLight needs the results of the depth below (k_gi_roots)
            for (int k = 0; k <= RTU_GI_BOUNCES; k++) {
                int rc = launch(ctx, frame, ctx->sample_buf, stream, zero_counters && i == 0 && k == 0, i, nb, nullptr, RTU_LAUNCH_CHAIN, k);
                if (rc != RTU_OK) return rc;
            }
            // the five shading steps are queued back to back; ONE host synchronisation per batch reads the (sticky) overflow
            // report of all of them. A batch that ran out of frame records is shaded again from the deepest depth with the
            // grown capacities: its chain records do not depend on them.
            for (;;) {
                for (int k = RTU_GI_BOUNCES; k >= 0; k--) {
                    int rc = launch(ctx, frame, ctx->sample_buf, stream, false, i, nb, nullptr, RTU_LAUNCH_SHADE, k);
                    if (rc != RTU_OK) return rc;
                }
                RTU_HIP(ctx, hipStreamSynchronize(stream));
                bool overflow = false;
                int rc = check_overflow(ctx, &overflow);
                if (rc != RTU_OK) return rc;
                if (!overflow) break;
                if (frame->collect_stats) return fail(ctx, RTU_ERR_CAPACITY, "recipe P with counters: frame records ran out; render once without counters first");
                if (++rounds > 8 * RTU_MAX_LEVELS) return fail(ctx, RTU_ERR_CAPACITY, "recursion frames still exceed the capacity after %d rounds", rounds);
            }
        }
        hipError_t e = (hipError_t)rtu_launch_accumulate(ctx->sample_buf, (uint32_t)nb, ctx->acc, ctx->acc_hits, (uint32_t)pixels, i == 0, stream);
        if (e != hipSuccess) return fail(ctx, RTU_ERR_HIP, "kernel launch: %s", hipGetErrorString(e));
    }
    hipError_t e = (hipError_t)rtu_launch_resolve(ctx->acc, ctx->acc_hits, d_out, (uint32_t)pixels, (uint32_t)frame->samples, stream);
    if (e != hipSuccess) return fail(ctx, RTU_ERR_HIP, "kernel launch: %s", hipGetErrorString(e));
    return RTU_OK;
}

}  // namespace

extern "C" {

int rtu_device_info(int device_id, RtuDeviceInfo* out) {
    if (!out) return RTU_ERR_ARG;
    memset(out, 0, sizeof *out);
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, device_id) != hipSuccess) return RTU_ERR_HIP;
    out->compute_units = p.multiProcessorCount;
    out->clock_khz = p.clockRate;
    out->memory_clock_khz = p.memoryClockRate;
    out->memory_bus_bits = p.memoryBusWidth;
    out->l2_bytes = (unsigned long long)p.l2CacheSize;
    out->hbm_bytes = (unsigned long long)p.totalGlobalMem;
    snprintf(out->name, sizeof out->name, "%s", p.name);
    snprintf(out->arch, sizeof out->arch, "%s", p.gcnArchName);
    return RTU_OK;
}

int rtu_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

const char* rtu_error_string(int err) {
    switch (err) {
        case RTU_OK: return "ok";
        case RTU_ERR_ARG: return "invalid argument";
        case RTU_ERR_HIP: return "HIP runtime error";
        case RTU_ERR_UNSUPPORTED: return "scene outside the device path's limits";
        case RTU_ERR_STOCHASTIC: return "scene uses a stochastic feature";
        case RTU_ERR_NO_SCENE: return "no scene uploaded";
        case RTU_ERR_NO_DEVICE: return "no such GPU";
        case RTU_ERR_CAPACITY: return "recursion frame capacity exceeded";
        case RTU_ERR_CANCELLED: return "cancelled";
    }
    return "unknown error";
}

RtuContext* rtu_create_context(int device_id, int* err_out) {
    int n = rtu_device_count();
    if (device_id < 0 || device_id >= n) {
        if (err_out) *err_out = RTU_ERR_NO_DEVICE;
        return nullptr;
    }
    RtuContext* ctx = new RtuContext;
    ctx->device = device_id;
    bool ok = hipSetDevice(device_id) == hipSuccess &&
              hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) == hipSuccess &&
              hipMalloc((void**)&ctx->fcnt_side, sizeof(FrameCounters)) == hipSuccess && hipMemset(ctx->fcnt_side, 0, sizeof(FrameCounters)) == hipSuccess &&
              hipEventCreate(&ctx->ev0) == hipSuccess && hipEventCreate(&ctx->ev1) == hipSuccess &&
              hipMalloc((void**)&ctx->counters, kCounterBytes) == hipSuccess &&
              hipMalloc((void**)&ctx->fcnt, sizeof(FrameCounters)) == hipSuccess &&
              hipMemset(ctx->fcnt, 0, sizeof(FrameCounters)) == hipSuccess &&
              hipMemset(ctx->counters, 0, kCounterBytes) == hipSuccess;
    if (!ok) {
        if (err_out) *err_out = RTU_ERR_HIP;
        rtu_destroy_context(ctx);
        return nullptr;
    }
    if (err_out) *err_out = RTU_OK;
    return ctx;
}

void rtu_destroy_context(RtuContext* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    if (ctx->aux_stream) { (void)hipStreamSynchronize(ctx->aux_stream); (void)hipStreamDestroy(ctx->aux_stream); }
    if (ctx->aux_ev0) (void)hipEventDestroy(ctx->aux_ev0);
    if (ctx->aux_ev1) (void)hipEventDestroy(ctx->aux_ev1);
    if (ctx->fcnt_side) (void)hipFree(ctx->fcnt_side);
    free_scene(ctx);
    free_levels(ctx);
    if (ctx->fcnt) (void)hipFree(ctx->fcnt);
    if (ctx->tl) (void)hipFree(ctx->tl);
    if (ctx->cover) (void)hipFree(ctx->cover);
    if (ctx->occ) (void)hipFree(ctx->occ);
    if (ctx->d_cams) (void)hipFree(ctx->d_cams);
    if (ctx->h_cams) (void)hipHostFree(ctx->h_cams);
    for (hipEvent_t e : ctx->cam_ev)
        if (e) (void)hipEventDestroy(e);
    if (ctx->fb) (void)hipFree(ctx->fb);
    if (ctx->acc) (void)hipFree(ctx->acc);
    if (ctx->acc_hits) (void)hipFree(ctx->acc_hits);
    if (ctx->sample_buf) (void)hipFree(ctx->sample_buf);
    if (ctx->gi_h) (void)hipFree(ctx->gi_h);
    if (ctx->gi_res) (void)hipFree(ctx->gi_res);
    if (ctx->counters) (void)hipFree(ctx->counters);
    if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
    if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
    for (hipEvent_t e : ctx->probe_ev)
        if (e) (void)hipEventDestroy(e);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

const char* rtu_last_error(const RtuContext* ctx) { return ctx ? ctx->error.c_str() : "context is NULL"; }

int rtu_validate_scene(const RtuSceneDesc* s, char* err_buf, size_t err_len) {
    RtuContext tmp;  // plain host state: nothing here touches a GPU
    const int rc = validate(&tmp, s);
    if (err_buf && err_len) {
        snprintf(err_buf, err_len, "%s", tmp.error.c_str());
    }
    return rc;
}

int rtu_upload_scene(RtuContext* ctx, const RtuSceneDesc* s) {
    if (!ctx) return RTU_ERR_ARG;
    int rc = validate(ctx, s);
    if (rc != RTU_OK) return rc;
    RTU_HIP(ctx, hipSetDevice(ctx->device));
    RTU_HIP(ctx, hipStreamSynchronize(ctx->stream));
    free_scene(ctx);

    // scene-graph nodes with their ancestor chains
    std::vector<DevNode> nodes(s->n_nodes);
    for (uint32_t i = 0; i < s->n_nodes; i++) {
        const RtuNode& n = s->nodes[i];
        DevNode& d = nodes[i];
        memset(&d, 0, sizeof d);
        memcpy(d.tm, n.tm, sizeof d.tm);
        memcpy(d.itm, n.itm, sizeof d.itm);
        memcpy(d.pos, n.pos, sizeof d.pos);
        d.parent = n.parent;
        d.obj_type = n.obj_type;
        d.mesh_id = n.mesh_id;
        d.material_id = n.material_id;
        d.depth = n.depth;
        int j = (int)i;
        for (int dd = n.depth; dd >= 0; dd--) {
            d.chain[dd] = j;
            j = s->nodes[j].parent;
        }
    }

    const float wscale = world_bounds(s, nodes);

    // meshes
    std::vector<RtuContext::MeshInfo> mesh_info;
    std::vector<DevMesh> meshes(s->n_meshes);
    std::vector<uint32_t> fast_nodes(s->n_meshes, 0);
    std::vector<std::vector<uint32_t>> fast_elements(s->n_meshes);  // per mesh: slot of the fast tree's leaf order -> face
    uint32_t stack_needed = 1;
    for (uint32_t mi = 0; mi < s->n_meshes; mi++) {
        const RtuMesh& m = s->meshes[mi];
        DevMesh& d = meshes[mi];
        memset(&d, 0, sizeof d);
        std::vector<float4> tri;
        build_tri_records(m, m.elements, m.n_elements, tri);
        for (uint32_t i = 1; i < m.n_bvh_nodes; i++) {
            const RtuBvhNode& bn = m.bvh[i];
            if (bn.bmin[0] > bn.bmax[0] || bn.bmin[1] > bn.bmax[1] || bn.bmin[2] > bn.bmax[2]) d.any_empty_box = 1;
        }
        static_assert(sizeof(RtuBvhNode) == 2 * sizeof(float4), "BVH node is two float4");
        // `ref` tree: the reference's, renumbered breadth-first (sibling pairs stay adjacent, the
        // root stays node 1): same tree, same traversal order.
        std::vector<RtuBvhNode> bfs(m.n_bvh_nodes);
        memset(bfs.data(), 0, bfs.size() * sizeof(RtuBvhNode));
        {
            std::vector<std::pair<uint32_t, uint32_t>> queue;  // (old id, new id)
            queue.push_back({1u, 1u});
            uint32_t next_free = 2;
            for (size_t qi = 0; qi < queue.size(); qi++) {
                auto [oldId, newId] = queue[qi];
                RtuBvhNode nn = m.bvh[oldId];
                if (nn.count == 0) {
                    queue.push_back({nn.index, next_free});
                    queue.push_back({nn.index + 1, next_free + 1});
                    nn.index = next_free;
                    next_free += 2;
                }
                bfs[newId] = nn;
            }
        }
        if ((rc = upload(ctx, reinterpret_cast<const float4*>(bfs.data()), (size_t)m.n_bvh_nodes * 2, &d.ref.bvh)) != RTU_OK) return rc;
        if ((rc = upload(ctx, tri.data(), tri.size(), &d.ref.tri)) != RTU_OK) return rc;
        if ((rc = upload(ctx, m.elements, (size_t)m.n_elements, &d.ref.elements)) != RTU_OK) return rc;
        // `fast` tree: binned SAH over the same triangles
        SahTree sah;
        build_sah(m, sah);
        if (sah.depth > RTU_MAX_BVH_STACK) return fail(ctx, RTU_ERR_UNSUPPORTED, "mesh %u: SAH tree depth %u > %d", mi, sah.depth, RTU_MAX_BVH_STACK);
        std::vector<uint32_t> sub_first, sub_total;
        dfs_order(sah, sub_first, sub_total);
        build_tri_records(m, sah.elements.data(), (uint32_t)sah.elements.size(), tri);
        d.fast.bvh = nullptr;  // the kernels walk the collapsed forms (bvh4 / bvh8) of this tree
        if ((rc = upload(ctx, tri.data(), tri.size(), &d.fast.tri)) != RTU_OK) return rc;
        if ((rc = upload(ctx, sah.elements.data(), sah.elements.size(), &d.fast.elements)) != RTU_OK) return rc;
        fast_elements[mi] = sah.elements;
        std::vector<float4> wide8;
        build_wide8(sah, sub_first, sub_total, wide8);
        if ((rc = upload(ctx, wide8.data(), wide8.size(), &d.bvh8)) != RTU_OK) return rc;
        fast_nodes[mi] = (uint32_t)(wide8.size() / 16);
        std::vector<float4> wide4;
        uint32_t need4 = 1;
        build_wide4(sah, wide4, need4);
        if ((rc = upload(ctx, wide4.data(), wide4.size(), &d.bvh4)) != RTU_OK) return rc;
        mesh_info.push_back({m.nf, sah.depth, need4, (uint32_t)(wide4.size() / 8), (uint32_t)(wide8.size() / 16)});
        if (need4 > RTU_MAX_BVH_STACK) need4 = RTU_MAX_BVH_STACK;  // a walk that needs more finishes on the reference's tree
        if (need4 > stack_needed) stack_needed = need4;
        if (sah.depth > stack_needed) stack_needed = sah.depth;
        if ((rc = upload(ctx, m.f, (size_t)m.nf * 3, &d.f)) != RTU_OK) return rc;
        if ((rc = upload(ctx, m.v, (size_t)m.nv * 3, &d.v)) != RTU_OK) return rc;
        if ((rc = upload(ctx, m.fn, (size_t)m.nf * 3, &d.fn)) != RTU_OK) return rc;
        if ((rc = upload(ctx, m.vn, (size_t)m.nvn * 3, &d.vn)) != RTU_OK) return rc;
        d.vt = nullptr;
        d.ft = nullptr;
        if (s->n_textures > 0 && m.vt && m.ft && m.nvt) {  // texture coordinates only travel with textured scenes
            if ((rc = upload(ctx, m.vt, (size_t)m.nvt * 3, &d.vt)) != RTU_OK) return rc;
            if ((rc = upload(ctx, m.ft, (size_t)m.nf * 3, &d.ft)) != RTU_OK) return rc;
        }
        d.scale = 0.0f;
        for (int k = 0; k < 3; k++) d.scale = fmaxf(d.scale, fmaxf(fabsf(m.bound_min[k]), fabsf(m.bound_max[k])));
        memcpy(d.bmin, m.bound_min, sizeof d.bmin);
        memcpy(d.bmax, m.bound_max, sizeof d.bmax);
        d.n_bvh_nodes = m.n_bvh_nodes;
        d.n_elements = m.n_elements;
        if (m.bvh_depth > stack_needed) stack_needed = m.bvh_depth;
    }

    // LDS node area of the cooperative kernels, handed out in mesh order
    {
        uint32_t budget = (uint32_t)RTU_LDS_NODE_F4 / 16;  // node8
        uint32_t used = 0;
        for (uint32_t mi = 0; mi < s->n_meshes; mi++) {
            uint32_t take = fast_nodes[mi];
            if (take > budget - used) take = budget - used;
            meshes[mi].lds_nodes = take;
            meshes[mi].lds_off = used * 16;
            used += take;
        }
    }

    DevScene ds;
    memset(&ds, 0, sizeof ds);
    if ((rc = upload(ctx, nodes.data(), nodes.size(), &ds.nodes)) != RTU_OK) return rc;
    if ((rc = upload(ctx, s->materials, (size_t)s->n_materials, &ds.materials)) != RTU_OK) return rc;
    if ((rc = upload(ctx, s->lights, (size_t)s->n_lights, &ds.lights)) != RTU_OK) return rc;
    if ((rc = upload(ctx, meshes.data(), meshes.size(), &ds.meshes)) != RTU_OK) return rc;
    // textures
    ds.textured = (s->n_textures > 0) ? 1u : 0u;
    if (ds.textured) {
        std::vector<DevTexture> texs(s->n_textures);
        for (uint32_t i = 0; i < s->n_textures; i++) {
            const RtuTexture& t = s->textures[i];
            DevTexture& o = texs[i];
            memset(&o, 0, sizeof o);
            o.type = t.type; o.width = t.width; o.height = t.height;
            memcpy(o.color1, t.color1, sizeof o.color1);
            memcpy(o.color2, t.color2, sizeof o.color2);
            if (t.type == RTU_TEX_FILE && (size_t)t.width * t.height > 0)
                if ((rc = upload(ctx, t.rgb, (size_t)t.width * t.height * 3, &o.rgb)) != RTU_OK) return rc;
        }
        if ((rc = upload(ctx, texs.data(), texs.size(), &ds.textures)) != RTU_OK) return rc;
        if (s->material_maps)
            if ((rc = upload(ctx, s->material_maps, (size_t)s->n_materials * 4, &ds.mat_maps)) != RTU_OK) return rc;
        ds.bg_map = s->background_map;
        ds.env_map = s->environment_map;
    }
    ds.bg = s->background;
    ds.env = s->environment;
    ds.img_w = s->camera.img_width;
    ds.img_h = s->camera.img_height;
    ctx->textured = ds.textured != 0;
    ds.n_nodes = s->n_nodes;
    ds.walk_stack_limit = 0xFFFFu;
    ds.wscale = wscale;
    ds.nol_ok = 1;
    for (uint32_t i = 0; i < s->n_lights; i++)
        for (int k = 0; k < 3; k++)
            if (!(std::fabs(s->lights[i].intensity[k]) < 1e15f)) ds.nol_ok = 0;
    ds.n_cover = 0;
    ctx->cover_faces = 0;
    std::vector<CoverMesh> cover_host;  // per masked mesh node: the world-space boxes and vertices of its triangles
    for (uint32_t i = 0; i < s->n_nodes && i < 64u; i++) {
        if (s->nodes[i].obj_type == RTU_OBJ_TRIMESH && ds.n_cover < RTU_MAX_COVER) {
            ds.cover_node[ds.n_cover++] = (int32_t)i;
            const RtuMesh& m = s->meshes[s->nodes[i].mesh_id];
            if (m.nf > ctx->cover_faces) ctx->cover_faces = m.nf;
            CoverMesh cm;
            make_cover_mesh(s, i, fast_elements[s->nodes[i].mesh_id], cm);
            if ((rc = upload(ctx, cm.boxes.data(), cm.boxes.size(), &ds.cover_box[ds.n_cover - 1])) != RTU_OK) return rc;
            ds.cover_nf[ds.n_cover - 1] = m.nf;
            cover_host.push_back(std::move(cm));
        }
    }
    // plane nodes with a coverage mask: the corners of the node's unit square in world space (k_plane_cover)
    ds.n_pcover = 0;
    for (uint32_t i = 0; i < s->n_nodes && i < 64u; i++) {
        if (s->nodes[i].obj_type != RTU_OBJ_PLANE || ds.n_pcover >= RTU_MAX_PCOVER) continue;
        static const double sq[4][2] = {{-1, -1}, {1, -1}, {1, 1}, {-1, 1}};
        float quad[4][3];
        bool finite = true;
        for (int c = 0; c < 4; c++) {
            double p[3] = {sq[c][0], sq[c][1], 0.0};
            for (int j = (int)i; j >= 0; j = s->nodes[j].parent) {
                const RtuNode& t = s->nodes[j];
                const double q[3] = {p[0] * t.tm[0] + p[1] * t.tm[3] + p[2] * t.tm[6] + t.pos[0], p[0] * t.tm[1] + p[1] * t.tm[4] + p[2] * t.tm[7] + t.pos[1],
                                     p[0] * t.tm[2] + p[1] * t.tm[5] + p[2] * t.tm[8] + t.pos[2]};
                p[0] = q[0]; p[1] = q[1]; p[2] = q[2];
            }
            for (int k = 0; k < 3; k++) { quad[c][k] = (float)p[k]; finite = finite && std::isfinite(quad[c][k]); }
        }
        if (!finite) continue;  // NaN / infinite transformation: no mask (the node's rectangle is the whole image as well)
        ds.pcover_node[ds.n_pcover] = (int32_t)i;
        memcpy(ds.pcover_quad[ds.n_pcover], quad, sizeof quad);
        ds.n_pcover++;
    }
    ctx->light_list_info.clear();
    if ((rc = build_light_lists(ctx, s, cover_host, wscale, ds)) != RTU_OK) return rc;
    ds.obj_mask = 0;
    for (uint32_t i = 0; i < s->n_nodes && i < 64u; i++)
        if (s->nodes[i].obj_type != RTU_OBJ_NONE) ds.obj_mask |= 1ull << i;
    ds.node_bounds = 1;
    {   // screen rectangles of the node-level bounds, one set per frame in flight (written by k_node_rects on every launch)
        void* d = nullptr;
        RTU_HIP(ctx, hipMalloc(&d, sizeof(int4) * (size_t)RTU_MAX_FRAME_BATCH * s->n_nodes));
        ctx->scene_allocs.push_back(d);
        ctx->node_rects = static_cast<int4*>(d);
    }
    ds.n_lights = s->n_lights;
    env_value(s->background, ds.background);
    env_value(s->environment, ds.environment);
    ctx->dscene = ds;
    ctx->nsl = 0;
    for (uint32_t i = 0; i < s->n_lights; i++)
        if (s->lights[i].type != RTU_LIGHT_AMBIENT) {
            if (ctx->nsl < RTU_FI_NOL_LIGHTS) {
                for (int k = 0; k < 3; k++) ctx->nol_light[ctx->nsl][k] = s->lights[i].vec[k];
                ctx->nol_light[ctx->nsl][3] = s->lights[i].type == RTU_LIGHT_DIRECT ? 1.0f : 0.0f;
            }
            ctx->shadow_light[ctx->nsl++] = (int32_t)i;
        }
    memset(ctx->want_cap_s, 0, sizeof ctx->want_cap_s);
    ctx->want_defer_s = 0;
    ctx->tail_hint = 0;
    ctx->tail_hints.clear();
    ctx->list_hints.clear();
    ctx->n_meshes = s->n_meshes;
    ctx->mesh_info = mesh_info;
    ctx->mesh_hits_childless = true;
    for (uint32_t i = 0; i < s->n_nodes; i++) {
        const RtuNode& nd = s->nodes[i];
        if (nd.obj_type != RTU_OBJ_TRIMESH || nd.material_id < 0) continue;
        const RtuMaterial& mm = s->materials[nd.material_id];
        for (int k = 0; k < 3; k++)
            if (mm.reflection[k] != 0 || mm.refraction[k] != 0) ctx->mesh_hits_childless = false;
    }
    ctx->side_off.clear();
    ctx->side_frames.clear();
    ctx->occ_hints.clear();
    ctx->any_recursive_material = false;
    for (uint32_t i = 0; i < s->n_materials; i++) {
        const RtuMaterial& mm = s->materials[i];
        for (int k = 0; k < 3; k++)
            if (mm.reflection[k] != 0 || mm.refraction[k] != 0) ctx->any_recursive_material = true;
    }
    ctx->scene_stochastic = false;
    ctx->stochastic_what.clear();
    if (s->camera.dof != 0) { ctx->scene_stochastic = true; ctx->stochastic_what = "depth of field"; }
    for (uint32_t i = 0; i < s->n_lights && !ctx->scene_stochastic; i++)
        if (s->lights[i].type == RTU_LIGHT_POINT && s->lights[i].size > 0) { ctx->scene_stochastic = true; ctx->stochastic_what = "a soft shadow"; }
    for (uint32_t i = 0; i < s->n_materials && !ctx->scene_stochastic; i++)
        if (s->materials[i].reflection_glossiness > 0 || s->materials[i].refraction_glossiness > 0) { ctx->scene_stochastic = true; ctx->stochastic_what = "a glossy bounce"; }
    ctx->bvh_stack_needed = stack_needed;
    ctx->has_scene = true;
    return RTU_OK;
}

// CalculateImageOrigin + the u,v of CalculateCurrentPoint (RenderFunctions.cpp:243-269)
int rtu_frame_setup(const RtuCamera* cam, int width, int height, RtuFrameDesc* out) {
    if (!cam || !out || width <= 0 || height <= 0) return RTU_ERR_ARG;
    memset(out, 0, sizeof *out);
    out->width = width;
    out->height = height;
    out->shard_rank = 0;
    out->shard_count = 1;
    out->max_bounce = RTU_MAX_BOUNCE;
    out->collect_stats = 0;
    f3 pos = ld3(cam->pos), dir = ld3(cam->dir), up = ld3(cam->up);
    float distanceToImg = cam->focaldist;
    float actualHeight = (float)(tan((cam->fov / 2) * M_PI / 180.0) * 2 * distanceToImg);  // :247
    float actualWidth = ((float)width / (float)height) * actualHeight;                      // :248
    f3 dirN = norm3(dir), upN = norm3(up);
    f3 topCenterPoint = (pos + dirN * distanceToImg) + upN * (actualHeight / 2);            // :250
    f3 right = norm3(cross3(dirN, upN));
    f3 origin = topCenterPoint - right * (actualWidth / 2);                                 // :252
    f3 u = right * (actualWidth / (float)width);                                            // :263
    f3 v = (upN * -1.0f) * (actualHeight / (float)height);                                  // :264
    out->cam_pos[0] = pos.x; out->cam_pos[1] = pos.y; out->cam_pos[2] = pos.z;
    out->origin[0] = origin.x; out->origin[1] = origin.y; out->origin[2] = origin.z;
    out->u[0] = u.x; out->u[1] = u.y; out->u[2] = u.z;
    out->v[0] = v.x; out->v[1] = v.y; out->v[2] = v.z;
    // the lens disk of RenderFunctions.cpp:93: camera.up as given, normalize(dir x up)
    out->lens_up[0] = up.x; out->lens_up[1] = up.y; out->lens_up[2] = up.z;
    out->lens_right[0] = right.x; out->lens_right[1] = right.y; out->lens_right[2] = right.z;
    out->dof = cam->dof;
    return RTU_OK;
}

int rtu_shard_rows(const RtuFrameDesc* f) {
    if (!f || f->shard_count < 1 || f->shard_rank < 0 || f->shard_rank >= f->shard_count || f->height <= 0) return 0;
    int bands = shard_bands(f->height, f->shard_rank, f->shard_count);
    if (bands == 0) return 0;
    int last_global_band = (bands - 1) * f->shard_count + f->shard_rank;
    int rows = bands * RTU_BAND_ROWS;
    int over = (last_global_band + 1) * RTU_BAND_ROWS - f->height;
    if (over > 0) rows -= over;
    return rows;
}

int rtu_shard_max_rows(int height, int shard_count) {
    if (height <= 0 || shard_count < 1) return 0;
    return shard_bands(height, 0, shard_count) * RTU_BAND_ROWS;
}

int rtu_shard_global_row(const RtuFrameDesc* f, int local_row) {
    if (!f || local_row < 0) return -1;
    int lb = local_row / RTU_BAND_ROWS;
    return (lb * f->shard_count + f->shard_rank) * RTU_BAND_ROWS + local_row % RTU_BAND_ROWS;
}

int rtu_render_frame_device(RtuContext* ctx, const RtuFrameDesc* frame, void* d_rgbz, void* hip_stream) {
    if (!ctx) return RTU_ERR_ARG;
    int rc = check_frame(ctx, frame);
    if (rc != RTU_OK) return rc;
    if (!ctx->has_scene) return fail(ctx, RTU_ERR_NO_SCENE, "no scene uploaded");
    if (!d_rgbz && rtu_shard_rows(frame) > 0) return fail(ctx, RTU_ERR_ARG, "d_rgbz is NULL");
    RTU_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t st = (hipStream_t)hip_stream;  // NULL is the device's default (null) stream
    if (frame->samples >= 1) return render_sampled(ctx, frame, (float4*)d_rgbz, st, true);
    return launch(ctx, frame, (float4*)d_rgbz, st, true);
}

int rtu_render_frames_device(RtuContext* ctx, const RtuFrameDesc* frames, int n_frames, void* d_rgbz, void* hip_stream) {
    if (!ctx) return RTU_ERR_ARG;
    if (!frames || n_frames < 1 || n_frames > RTU_MAX_FRAMES_IN_FLIGHT) return fail(ctx, RTU_ERR_ARG, "1..%d frames per call", RTU_MAX_FRAMES_IN_FLIGHT);
    for (int i = 0; i < n_frames; i++) {
        int rc = check_frame(ctx, &frames[i]);
        if (rc != RTU_OK) return rc;
        const RtuFrameDesc &f = frames[i], &g = frames[0];
        if (f.samples != 0) return fail(ctx, RTU_ERR_ARG, "frames in flight are frames of recipe W (samples == 0)");
        if (f.width != g.width || f.height != g.height || f.shard_rank != g.shard_rank || f.shard_count != g.shard_count ||
            f.max_bounce != g.max_bounce || f.collect_stats != g.collect_stats || f.coop_threshold != g.coop_threshold)
            return fail(ctx, RTU_ERR_ARG, "frames in flight differ in more than their cameras");
    }
    if (!ctx->has_scene) return fail(ctx, RTU_ERR_NO_SCENE, "no scene uploaded");
    const size_t pixels = (size_t)rtu_shard_rows(&frames[0]) * (size_t)frames[0].width;
    if (pixels == 0) return RTU_OK;
    if (!d_rgbz) return fail(ctx, RTU_ERR_ARG, "d_rgbz is NULL");
    if (pixels * (size_t)n_frames > ((size_t)1 << 26)) return fail(ctx, RTU_ERR_ARG, "more than 2^26 pixels in flight");
    RTU_HIP(ctx, hipSetDevice(ctx->device));
    static_assert(RTU_MAX_FRAMES_IN_FLIGHT == RTU_MAX_FRAME_BATCH, "batch size");
    return launch(ctx, &frames[0], (float4*)d_rgbz, (hipStream_t)hip_stream, true, 0, n_frames, frames);
}

int rtu_pack_image_device(RtuContext* ctx, const void* d_rgbz, size_t n_pixels, void* d_z, void* d_rgb8, void* hip_stream) {
    if (!ctx) return RTU_ERR_ARG;
    if (n_pixels == 0) return RTU_OK;
    if (!d_rgbz || !d_z || !d_rgb8) return fail(ctx, RTU_ERR_ARG, "NULL buffer");
    RTU_HIP(ctx, hipSetDevice(ctx->device));
    hipError_t e = (hipError_t)rtu_launch_pack_image((const float4*)d_rgbz, (unsigned long long)n_pixels, (float*)d_z, (unsigned char*)d_rgb8, (hipStream_t)hip_stream);
    if (e != hipSuccess) return fail(ctx, RTU_ERR_HIP, "kernel launch: %s", hipGetErrorString(e));
    return RTU_OK;
}

int rtu_minmax_z_device(RtuContext* ctx, const void* d_rgbz, size_t pixels_per_frame, int n_frames, void* d_minmax, void* hip_stream) {
    if (!ctx || n_frames < 0 || pixels_per_frame > 0xFFFFFFFFull) return RTU_ERR_ARG;
    if (n_frames == 0) return RTU_OK;
    if (!d_minmax || (pixels_per_frame != 0 && !d_rgbz)) return fail(ctx, RTU_ERR_ARG, "NULL buffer");
    RTU_HIP(ctx, hipSetDevice(ctx->device));
    if (pixels_per_frame == 0) {
        // a shard without rows (fewer 8-row bands than ranks) still takes part in the all-reduce MIN of the keys: it must contribute
        // the "nothing yet" keys, not whatever the buffer held (zeros would win every MIN and turn the z-image of EVERY rank black)
        RTU_HIP(ctx, hipMemsetAsync(d_minmax, 0x7F, sizeof(long long) * 2 * (size_t)n_frames, (hipStream_t)hip_stream));
        return RTU_OK;
    }
    hipError_t e = (hipError_t)rtu_launch_minmax_z((const float4*)d_rgbz, (uint32_t)pixels_per_frame, (uint32_t)n_frames, (long long*)d_minmax, (hipStream_t)hip_stream);
    if (e != hipSuccess) return fail(ctx, RTU_ERR_HIP, "kernel launch: %s", hipGetErrorString(e));
    return RTU_OK;
}

int rtu_pack_output_device(RtuContext* ctx, const void* d_rgbz, size_t pixels_per_frame, int n_frames, const void* d_minmax, void* d_out4, void* hip_stream) {
    if (!ctx || n_frames < 0 || pixels_per_frame > 0xFFFFFFFFull) return RTU_ERR_ARG;
    if (pixels_per_frame == 0 || n_frames == 0) return RTU_OK;
    if (!d_rgbz || !d_minmax || !d_out4) return fail(ctx, RTU_ERR_ARG, "NULL buffer");
    RTU_HIP(ctx, hipSetDevice(ctx->device));
    hipError_t e = (hipError_t)rtu_launch_pack_output((const float4*)d_rgbz, (uint32_t)pixels_per_frame, (uint32_t)n_frames, (const long long*)d_minmax,
                                                      (unsigned char*)d_out4, (hipStream_t)hip_stream);
    if (e != hipSuccess) return fail(ctx, RTU_ERR_HIP, "kernel launch: %s", hipGetErrorString(e));
    return RTU_OK;
}

int rtu_get_stats(RtuContext* ctx, RtuStats* stats) {
    if (!ctx || !stats) return RTU_ERR_ARG;
    RTU_HIP(ctx, hipSetDevice(ctx->device));
    RTU_HIP(ctx, hipDeviceSynchronize());
    static_assert(sizeof(RtuStats) == 11 * sizeof(unsigned long long), "RtuStats layout");
    RTU_HIP(ctx, hipMemcpy(stats, ctx->counters, sizeof(RtuStats), hipMemcpyDeviceToHost));
    std::unique_ptr<FrameCounters> hp(new FrameCounters);  // a quarter of a megabyte: not on the stack
    FrameCounters& h = *hp;
    RTU_HIP(ctx, hipMemcpy(&h, ctx->fcnt, sizeof h, hipMemcpyDeviceToHost));
    if (!h.overflow) learn_tail(ctx, h, nullptr);  // (after a counting render: no side mode)
    return RTU_OK;
}

int rtu_get_touched(RtuContext* ctx, RtuTouched* per_slot, int n_slots) {
    if (!ctx || !per_slot || n_slots < 1) return RTU_ERR_ARG;
    RTU_HIP(ctx, hipSetDevice(ctx->device));
    RTU_HIP(ctx, hipDeviceSynchronize());
    std::vector<unsigned long long> h((size_t)RTU_TL_KERNELS * RTU_TOUCH_STRIDE);
    RTU_HIP(ctx, hipMemcpy(h.data(), ctx->counters, kCounterBytes, hipMemcpyDeviceToHost));
    static_assert(sizeof(RtuTouched) == RTU_TOUCH_FIELDS * sizeof(unsigned long long), "RtuTouched layout");
    static_assert(RTU_KERNEL_SLOTS == RTU_TL_KERNELS, "slot count");
    const int n = n_slots < RTU_TL_KERNELS ? n_slots : RTU_TL_KERNELS;
    for (int k = 0; k < n; k++) memcpy(&per_slot[k], &h[(size_t)k * RTU_TOUCH_STRIDE], sizeof(RtuTouched));
    return n;
}

int rtu_get_touched_launches(RtuContext* ctx, uint32_t* per_slot, int n_slots) {
    if (!ctx || !per_slot || n_slots < 0) return RTU_ERR_ARG;
    const int n = n_slots < RTU_TL_KERNELS ? n_slots : RTU_TL_KERNELS;
    for (int i = 0; i < n; i++) per_slot[i] = ctx->slot_launches[i];
    return n;
}

unsigned long long rtu_touched_bytes(const RtuTouched* t, int textured) {
    if (!t) return 0;
    return 24ull * t->bound_tests + 48ull * t->node_tests + 24ull * t->mesh_box_tests + 84ull * t->xform_levels + 112ull * t->inner4 + 256ull * t->inner8 +
           64ull * t->inner_ref + 64ull * t->tri_tests + (textured ? 148ull : 100ull) * t->winners + t->record_bytes;
}

const char* rtu_kernel_slot_name(int slot) {
    static const char* const level_kernels[4] = {"k_trace", "k_trace2c", "k_trace2", "k_consume"};
    static char buf[RTU_TL_KERNELS][24];
    if (slot < 0 || slot >= RTU_TL_KERNELS) return "";
    if (slot == 0) return "k_primary";
    if (slot == 1) return "k_primary2c";
    if (slot == 2) return "k_primary2";
    if (slot < 3 + 4 * RTU_MAX_LEVELS) snprintf(buf[slot], sizeof buf[slot], "%s(L%d)", level_kernels[(slot - 3) % 4], (slot - 3) / 4);
    else if (slot < 3 + 5 * RTU_MAX_LEVELS) snprintf(buf[slot], sizeof buf[slot], "k_combine(L%d)", slot - (3 + 4 * RTU_MAX_LEVELS));
    else if (slot == 3 + 5 * RTU_MAX_LEVELS) return "k_gi_roots";
    else if (slot == 4 + 5 * RTU_MAX_LEVELS) return "k_tail(side)";
    else return "";
    return buf[slot];
}

int rtu_probe_kernel(RtuContext* ctx, int slot) {
    if (!ctx || slot >= RTU_TL_KERNELS) return RTU_ERR_ARG;
    RTU_HIP(ctx, hipSetDevice(ctx->device));
    if (slot >= 0 && !ctx->probe_ev[0])
        for (hipEvent_t& e : ctx->probe_ev) RTU_HIP(ctx, hipEventCreate(&e));
    ctx->probe_slot = slot < 0 ? -1 : slot;
    if (slot >= 0) ctx->probe_used = 0;  // stopping keeps what was measured until it is read
    return RTU_OK;
}

int rtu_probe_read(RtuContext* ctx, float* total_ms_out, int* launches_out) {
    if (!ctx || !total_ms_out || !launches_out) return RTU_ERR_ARG;
    RTU_HIP(ctx, hipSetDevice(ctx->device));
    RTU_HIP(ctx, hipDeviceSynchronize());
    float total = 0;
    for (int i = 0; i < ctx->probe_used; i++) {
        float ms = 0;
        RTU_HIP(ctx, hipEventElapsedTime(&ms, ctx->probe_ev[2 * i], ctx->probe_ev[2 * i + 1]));
        total += ms;
    }
    *total_ms_out = total;
    *launches_out = ctx->probe_used;
    ctx->probe_used = 0;
    return RTU_OK;
}

int rtu_render_frame(RtuContext* ctx, const RtuFrameDesc* frame, float* h_rgbz, RtuStats* stats) {
    if (!ctx) return RTU_ERR_ARG;
    int rc = check_frame(ctx, frame);
    if (rc != RTU_OK) return rc;
    if (!ctx->has_scene) return fail(ctx, RTU_ERR_NO_SCENE, "no scene uploaded");
    if (!h_rgbz) return fail(ctx, RTU_ERR_ARG, "h_rgbz is NULL");
    RTU_HIP(ctx, hipSetDevice(ctx->device));
    size_t rows = (size_t)rtu_shard_rows(frame);
    size_t bytes = rows * (size_t)frame->width * sizeof(float4);
    if (bytes > ctx->fb_bytes) {
        if (ctx->fb) (void)hipFree(ctx->fb);
        ctx->fb = nullptr;
        ctx->fb_bytes = 0;
        RTU_HIP(ctx, hipMalloc((void**)&ctx->fb, bytes));
        ctx->fb_bytes = bytes;
    }
    RtuFrameDesc f = *frame;
    if (stats) f.collect_stats = 1;
    for (int attempt = 0;; attempt++) {
        if (f.samples >= 1) {  // recipe S settles its capacities pass by pass
            if ((rc = render_sampled(ctx, &f, ctx->fb, ctx->stream, true)) != RTU_OK) return rc;
            RTU_HIP(ctx, hipStreamSynchronize(ctx->stream));
            break;
        }
        rc = launch(ctx, &f, ctx->fb, ctx->stream, true);
        if (rc != RTU_OK) return rc;
        RTU_HIP(ctx, hipStreamSynchronize(ctx->stream));
        bool overflow = false;
        if ((rc = check_overflow(ctx, &overflow)) != RTU_OK) return rc;
        if (!overflow) break;
        // more frames than provisioned: check_overflow has raised the wanted capacities; render again
        // (every round settles at least one more recursion level)
        if (attempt >= 2 * RTU_MAX_LEVELS) return fail(ctx, RTU_ERR_CAPACITY, "recursion frames still exceed the capacity after %d rounds", attempt);
    }
    if (bytes) RTU_HIP(ctx, hipMemcpy(h_rgbz, ctx->fb, bytes, hipMemcpyDeviceToHost));
    if (stats) return rtu_get_stats(ctx, stats);
    return RTU_OK;
}

int rtu_frame_status(RtuContext* ctx) {
    if (!ctx) return RTU_ERR_ARG;
    RTU_HIP(ctx, hipSetDevice(ctx->device));
    RTU_HIP(ctx, hipDeviceSynchronize());
    bool overflow = false;
    int rc = check_overflow(ctx, &overflow);
    if (rc != RTU_OK) return rc;
    if (overflow) {
        return fail(ctx, RTU_ERR_CAPACITY, "recursion frames exceeded the provisioned capacity (or the tail kernel refused its cut level); render the frame again");
    }
    return RTU_OK;
}

int rtu_time_render(RtuContext* ctx, const RtuFrameDesc* frame, void* d_rgbz, void* hip_stream, int iters, float* avg_ms_out) {
    if (!ctx || !avg_ms_out || iters < 1) return RTU_ERR_ARG;
    int rc = check_frame(ctx, frame);
    if (rc != RTU_OK) return rc;
    if (!ctx->has_scene) return fail(ctx, RTU_ERR_NO_SCENE, "no scene uploaded");
    if (!d_rgbz) return fail(ctx, RTU_ERR_ARG, "d_rgbz is NULL");
    RTU_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t st = (hipStream_t)hip_stream;
    RTU_HIP(ctx, hipEventRecord(ctx->ev0, st));
    for (int i = 0; i < iters; i++) {
        rc = frame->samples >= 1 ? render_sampled(ctx, frame, (float4*)d_rgbz, st, i == 0) : launch(ctx, frame, (float4*)d_rgbz, st, i == 0);
        if (rc != RTU_OK) return rc;
    }
    RTU_HIP(ctx, hipEventRecord(ctx->ev1, st));
    RTU_HIP(ctx, hipEventSynchronize(ctx->ev1));
    {
        bool overflow = false;
        if ((rc = check_overflow(ctx, &overflow)) != RTU_OK) return rc;
        if (overflow) return fail(ctx, RTU_ERR_CAPACITY, "recursion frames exceeded the provisioned capacity; time the frame again");
    }
    float ms = 0;
    RTU_HIP(ctx, hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
    *avg_ms_out = ms / (float)iters;
    return RTU_OK;
}

int rtu_render_timeline(RtuContext* ctx, const RtuFrameDesc* frame, void* d_rgbz, int max_entries, int* slot_out, double* start_us_out,
                        double* end_us_out) {
    if (!ctx || !slot_out || !start_us_out || !end_us_out || max_entries < 1) return RTU_ERR_ARG;
    int rc = check_frame(ctx, frame);
    if (rc != RTU_OK) return rc;
    if (!ctx->has_scene) return fail(ctx, RTU_ERR_NO_SCENE, "no scene uploaded");
    if (!d_rgbz) return fail(ctx, RTU_ERR_ARG, "d_rgbz is NULL");
    if (frame->samples != 0) return fail(ctx, RTU_ERR_ARG, "the timeline is of one launch sequence of recipe W (samples == 0)");
    RTU_HIP(ctx, hipSetDevice(ctx->device));
    const size_t n = (size_t)RTU_TL_KERNELS * RTU_TL_STRIDE;
    if (!ctx->tl) RTU_HIP(ctx, hipMalloc((void**)&ctx->tl, n * sizeof(unsigned long long)));
    RTU_HIP(ctx, hipMemsetAsync(ctx->tl, 0, n * sizeof(unsigned long long), ctx->stream));
    ctx->stamp_next = true;
    rc = launch(ctx, frame, (float4*)d_rgbz, ctx->stream, true);
    ctx->stamp_next = false;
    if (rc != RTU_OK) return rc;
    RTU_HIP(ctx, hipStreamSynchronize(ctx->stream));
    {
        bool overflow = false;
        if ((rc = check_overflow(ctx, &overflow)) != RTU_OK) return rc;  // also learns where the tail kernel may take over
        if (overflow) return fail(ctx, RTU_ERR_CAPACITY, "recursion frames exceeded the provisioned capacity; render the frame again");
    }
    std::vector<unsigned long long> h(n);
    RTU_HIP(ctx, hipMemcpy(h.data(), ctx->tl, n * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    int khz = 0;
    RTU_HIP(ctx, hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, ctx->device));
    if (khz <= 0) khz = 100000;
    unsigned long long lo[RTU_TL_KERNELS], hi[RTU_TL_KERNELS], t0 = ~0ull;
    for (int k = 0; k < RTU_TL_KERNELS; k++) {
        lo[k] = ~0ull; hi[k] = 0;
        const unsigned long long* row = h.data() + (size_t)k * RTU_TL_STRIDE;
        for (uint32_t j = 0; j < 64; j++) if (row[j] && row[j] < lo[k]) lo[k] = row[j];
        for (uint32_t j = 64; j < RTU_TL_STRIDE; j++) if (row[j] > hi[k]) hi[k] = row[j];
        if (hi[k] && lo[k] < t0) t0 = lo[k];
    }
    int cnt = 0;
    for (int k = 0; k < RTU_TL_KERNELS && cnt < max_entries; k++) {
        if (hi[k] == 0 || lo[k] == ~0ull) continue;  // not launched
        slot_out[cnt] = k;
        start_us_out[cnt] = (double)(lo[k] - t0) * 1e3 / (double)khz;
        end_us_out[cnt] = (double)(hi[k] - t0) * 1e3 / (double)khz;
        cnt++;
    }
    return cnt;
}

int rtu_timeline_exits(RtuContext* ctx, int slot, int max_values, double* exit_us_out) {
    if (!ctx || !exit_us_out || slot < 0 || slot >= RTU_TL_KERNELS || max_values < 1) return RTU_ERR_ARG;
    if (!ctx->tl) return fail(ctx, RTU_ERR_ARG, "no timeline recorded yet (rtu_render_timeline)");
    RTU_HIP(ctx, hipSetDevice(ctx->device));
    std::vector<unsigned long long> h(RTU_TL_STRIDE);
    RTU_HIP(ctx, hipMemcpy(h.data(), ctx->tl + (size_t)slot * RTU_TL_STRIDE, RTU_TL_STRIDE * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    int khz = 0;
    RTU_HIP(ctx, hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, ctx->device));
    if (khz <= 0) khz = 100000;
    unsigned long long t0 = ~0ull;
    for (uint32_t j = 0; j < 64; j++) if (h[j] && h[j] < t0) t0 = h[j];
    if (t0 == ~0ull) return 0;
    int n = 0;
    for (uint32_t j = 64; j < RTU_TL_STRIDE && n < max_values; j++)
        if (h[j]) exit_us_out[n++] = (double)(h[j] - t0) * 1e3 / (double)khz;
    return n;
}

int rtu_set_cancel_flag(RtuContext* ctx, const volatile int* flag) {
    if (!ctx) return RTU_ERR_ARG;
    ctx->cancel = flag;
    return RTU_OK;
}

int rtu_debug_tail_from(RtuContext* ctx, int level) {
    if (!ctx || level < 1 || level > RTU_MAX_LEVELS) return RTU_ERR_ARG;
    ctx->tail_hint = level;
    return RTU_OK;
}

int rtu_set_sequences_in_flight(RtuContext* ctx, int n) {
    if (!ctx || n < 1) return RTU_ERR_ARG;
    ctx->sequences_in_flight = n;
    return RTU_OK;
}

int rtu_debug_flags(RtuContext* ctx, uint32_t bits) {
    if (!ctx) return RTU_ERR_ARG;
    ctx->dbg = bits;
    return RTU_OK;
}

int rtu_debug_node_bounds(RtuContext* ctx, int on) {
    if (!ctx) return RTU_ERR_ARG;
    ctx->dscene.node_bounds = on ? 1u : 0u;
    return RTU_OK;
}

int rtu_debug_walk_stack_limit(RtuContext* ctx, uint32_t entries) {
    if (!ctx || entries < 1) return RTU_ERR_ARG;
    ctx->dscene.walk_stack_limit = entries;
    return RTU_OK;
}

int rtu_mesh_info(const RtuContext* ctx, uint32_t mesh, uint32_t* out5) {
    if (!ctx || !out5 || mesh >= ctx->mesh_info.size()) return RTU_ERR_ARG;
    const RtuContext::MeshInfo& i = ctx->mesh_info[mesh];
    out5[0] = i.faces; out5[1] = i.sah_depth; out5[2] = i.stack4; out5[3] = i.nodes4; out5[4] = i.nodes8;
    return RTU_OK;
}

int rtu_debug_light_list(const RtuSceneDesc* s, uint32_t light_slot, uint32_t cover_slot, RtuLightListDump* out) {
    if (!out) return RTU_ERR_ARG;
    memset(out, 0, sizeof *out);
    RtuContext tmp;  // plain host state: nothing here touches a GPU
    int rc = validate(&tmp, s);
    if (rc != RTU_OK) return rc;
    int li = -1, node = -1;
    uint32_t seen = 0;
    for (uint32_t i = 0; i < s->n_lights; i++)
        if (s->lights[i].type != RTU_LIGHT_AMBIENT && seen++ == light_slot) { li = (int)i; break; }
    seen = 0;
    for (uint32_t i = 0; i < s->n_nodes && i < 64u; i++)
        if (s->nodes[i].obj_type == RTU_OBJ_TRIMESH && seen++ == cover_slot) { node = (int)i; break; }
    if (li < 0 || node < 0) return RTU_ERR_ARG;
    std::vector<DevNode> nodes(s->n_nodes);
    const float wscale = world_bounds(s, nodes);
    SahTree sah;
    build_sah(s->meshes[s->nodes[node].mesh_id], sah);
    CoverMesh cm;
    make_cover_mesh(s, (uint32_t)node, sah.elements, cm);
    HostLightList hl;
    out->node = node;
    out->light = li;
    if (!compute_light_list(s->lights[li], cm, wscale, hl)) return RTU_OK;  // usable == 0
    out->usable = 1;
    out->G = hl.m.G;
    out->point = hl.m.point;
    memcpy(out->X, hl.m.X, sizeof out->X); memcpy(out->Y, hl.m.Y, sizeof out->Y); memcpy(out->Z, hl.m.Z, sizeof out->Z); memcpy(out->L, hl.m.L, sizeof out->L);
    out->u0 = hl.m.u0; out->v0 = hl.m.v0; out->su = hl.m.su; out->sv = hl.m.sv;
    out->n_entries = (uint32_t)(hl.ent.size() / 2);
    out->cell_off = (uint32_t*)malloc(hl.off.size() * sizeof(uint32_t));
    out->entry_face = (uint32_t*)malloc((hl.ent.size() / 2 + 1) * sizeof(uint32_t));
    out->entry_zmin = (float*)malloc((hl.ent.size() / 2 + 1) * sizeof(float));
    if (!out->cell_off || !out->entry_face || !out->entry_zmin) { rtu_debug_light_list_free(out); return RTU_ERR_ARG; }
    memcpy(out->cell_off, hl.off.data(), hl.off.size() * sizeof(uint32_t));
    for (size_t e = 0; e < hl.ent.size() / 2; e++) {
        out->entry_face[e] = sah.elements[hl.ent[2 * e]];  // slot of the fast tree -> face of the mesh
        memcpy(&out->entry_zmin[e], &hl.ent[2 * e + 1], 4);
    }
    return RTU_OK;
}

void rtu_debug_light_list_free(RtuLightListDump* d) {
    if (!d) return;
    free(d->cell_off); free(d->entry_face); free(d->entry_zmin);
    d->cell_off = nullptr; d->entry_face = nullptr; d->entry_zmin = nullptr;
}

int rtu_light_list_info(const RtuContext* ctx, uint32_t index, uint32_t* out5) {
    if (!ctx || !out5) return RTU_ERR_ARG;
    if (index >= ctx->light_list_info.size()) return RTU_ERR_ARG;
    const RtuContext::LightListInfo& i = ctx->light_list_info[index];
    out5[0] = i.light; out5[1] = i.cover; out5[2] = i.G; out5[3] = i.entries; out5[4] = i.longest;
    return RTU_OK;
}

int rtu_frame_counts(RtuContext* ctx, uint32_t* frames_out, uint32_t* deferred_out) {
    if (!ctx || !frames_out || !deferred_out) return RTU_ERR_ARG;
    RTU_HIP(ctx, hipSetDevice(ctx->device));
    RTU_HIP(ctx, hipDeviceSynchronize());
    std::unique_ptr<FrameCounters> hp(new FrameCounters);  // a quarter of a megabyte: not on the stack
    FrameCounters& h = *hp;
    RTU_HIP(ctx, hipMemcpy(&h, ctx->fcnt, sizeof h, hipMemcpyDeviceToHost));
    for (int L = 0; L < RTU_MAX_LEVELS; L++) {
        frames_out[L] = 0;
        for (int s = 0; s < RTU_SHARDS; s++) frames_out[L] += h.n_frames[L][(s) * RTU_CSTRIDE];
    }
    for (int p = 0; p <= RTU_MAX_LEVELS; p++) {
        deferred_out[p] = 0;
        for (int s = 0; s < RTU_SHARDS; s++) deferred_out[p] += h.n_defer[p][(s) * RTU_CSTRIDE];
    }
    if (ctx->last_side) {  // side mode: the primary phase's defer list and the frames its stage 2 made are counted apart (KernelArgs::fcnt0)
        RTU_HIP(ctx, hipMemcpy(&h, ctx->fcnt_side, sizeof h, hipMemcpyDeviceToHost));
        for (int L = 0; L < RTU_MAX_LEVELS; L++)
            for (int s = 0; s < RTU_SHARDS; s++) frames_out[L] += h.n_frames[L][(s) * RTU_CSTRIDE];
        for (int s = 0; s < RTU_SHARDS; s++) deferred_out[0] += h.n_defer[0][(s) * RTU_CSTRIDE];
    }
    return RTU_OK;
}

int rtu_selftest_division(RtuContext* ctx, unsigned long long n_pairs, unsigned long long seed, unsigned long long* mismatches_out) {
    if (!ctx || !mismatches_out) return RTU_ERR_ARG;
    RTU_HIP(ctx, hipSetDevice(ctx->device));
    RTU_HIP(ctx, hipMemsetAsync(ctx->counters, 0, sizeof(unsigned long long), ctx->stream));
    hipError_t e = (hipError_t)rtu_launch_selftest_fdiv(n_pairs, seed, ctx->counters, ctx->stream);
    if (e != hipSuccess) return fail(ctx, RTU_ERR_HIP, "selftest launch: %s", hipGetErrorString(e));
    RTU_HIP(ctx, hipStreamSynchronize(ctx->stream));
    RTU_HIP(ctx, hipMemcpy(mismatches_out, ctx->counters, sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return RTU_OK;
}

int rtu_selftest_primitives(RtuContext* ctx, unsigned long long n_rays, unsigned long long seed, unsigned long long* mismatches_out) {
    if (!ctx || !mismatches_out) return RTU_ERR_ARG;
    RTU_HIP(ctx, hipSetDevice(ctx->device));
    RTU_HIP(ctx, hipMemsetAsync(ctx->counters, 0, sizeof(unsigned long long), ctx->stream));
    hipError_t e = (hipError_t)rtu_launch_selftest_prims(n_rays, seed, ctx->counters, ctx->stream);
    if (e != hipSuccess) return fail(ctx, RTU_ERR_HIP, "selftest launch: %s", hipGetErrorString(e));
    RTU_HIP(ctx, hipStreamSynchronize(ctx->stream));
    RTU_HIP(ctx, hipMemcpy(mismatches_out, ctx->counters, sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return RTU_OK;
}

void* rtu_device_alloc(RtuContext* ctx, size_t bytes) {
    if (!ctx || bytes == 0) return nullptr;
    if (hipSetDevice(ctx->device) != hipSuccess) return nullptr;
    void* p = nullptr;
    if (hipMalloc(&p, bytes) != hipSuccess) return nullptr;
    return p;
}

void rtu_device_free(RtuContext* ctx, void* d_ptr) {
    if (!ctx || !d_ptr) return;
    (void)hipSetDevice(ctx->device);
    (void)hipFree(d_ptr);
}

void* rtu_context_stream(RtuContext* ctx) { return ctx ? (void*)ctx->stream : nullptr; }

int rtu_context_device(const RtuContext* ctx) { return ctx ? ctx->device : -1; }

int rtu_context_sync(RtuContext* ctx) {
    if (!ctx) return RTU_ERR_ARG;
    RTU_HIP(ctx, hipSetDevice(ctx->device));
    RTU_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return RTU_OK;
}

void* rtu_host_alloc_pinned(size_t bytes) {
    void* p = nullptr;
    if (bytes == 0 || hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) return nullptr;
    return p;
}

void rtu_host_free_pinned(void* p) {
    if (p) (void)hipHostFree(p);
}

int rtu_copy_to_host_async(RtuContext* ctx, void* h_dst, const void* d_src, size_t bytes, void* hip_stream) {
    if (!ctx || !h_dst || !d_src) return RTU_ERR_ARG;
    RTU_HIP(ctx, hipSetDevice(ctx->device));
    RTU_HIP(ctx, hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, (hipStream_t)hip_stream));
    return RTU_OK;
}

int rtu_copy_to_host(RtuContext* ctx, void* h_dst, const void* d_src, size_t bytes) {
    if (!ctx || !h_dst || !d_src) return RTU_ERR_ARG;
    RTU_HIP(ctx, hipSetDevice(ctx->device));
    RTU_HIP(ctx, hipDeviceSynchronize());
    RTU_HIP(ctx, hipMemcpy(h_dst, d_src, bytes, hipMemcpyDeviceToHost));
    return RTU_OK;
}

}  // extern "C"
