// render_kernel.hip — the per-pixel render hot path as one hand-written gfx950 kernel.
//
// One 64-lane wavefront per workgroup renders one 8x8 pixel tile; each lane owns
// a pixel for the whole recursion (primary ray -> closest hit through the scene
// node tree and the mesh BVH -> Blinn shading with shadow rays -> reflection /
// refraction bounces). No MFMA: there is no dense contraction on this path.
//
// What replaces what (reference file:line):
//   tile/lane -> (x,y)            PixelIterator::GetPixelLocation   PixelIterator.h:25-38
//   primary ray                   CalculateCurrentPoint + Ray        RenderFunctions.cpp:96-97,258-268
//   trace<false>/trace<true>      Trace / ShadowTrace                RenderFunctions.cpp:181-240
//   to_node / from_node           Node::ToNodeCoords/FromNodeCoords  scene.h:501-512
//   box_slabs                     Box::IntersectRay, BVHBoxIntersection  objFunctions.cpp:143-254,408-522
//   sphere_hit / plane_hit        Sphere/Plane::IntersectRay         objFunctions.cpp:15-140
//   mesh_hit / tri_hit            TriObj::IntersectRay/IntersectTriangle  objFunctions.cpp:257-406
//   direct_light / illuminate     MtlBlinn::Shade :125-155, Illuminate/Shadow  lightFunctions.cpp:27-84
//   the stage machine in render   MtlBlinn::Shade :158-292 (recursion made explicit)
//
// Bit parity: compiled -ffp-contract=off with IEEE divide/sqrt; every expression
// keeps the reference's order and its float->double promotions.
//
// Recursion: Shade() calls itself up to depth 5 with a branching factor of up to
// 3 and combines child results non-linearly, so the recursion is emulated
// exactly with an explicit frame stack (17 floats per level) instead of a
// throughput-weighted ray queue: the parent's partial sum, the pending term and
// the hit are saved, the child frame runs, and the parent resumes at the stage it
// left. Saved frames live in an HBM arena (one coalesced column per lane); the
// BVH traversal stack, which is touched on every node visit, lives in LDS.
#include "rtu_device.h"

namespace {

struct Ray {
    f3 p, dir;
};

struct Hit {  // HitInfo without uvw/duvw (no textures on this path) — scene.h:150-163
    float z;
    f3    p, N;
    int   node;
    bool  front;
};

struct Counters {
    unsigned prim, prim_hit, sec, shd, node, mesh, inner, leafv, leafe, tri, acc;
};

#define RTU_CNT(field) do { if (STATS) cnt.field++; } while (0)

// ---------------------------------------------------------------------------
// Node::ToNodeCoords (scene.h:501-507): p' = itm*(p-pos); d' = itm*((p+d)-pos) - p'
__device__ __forceinline__ Ray to_node(const DevNode& n, const Ray& r) {
    f3 pos = ld3(n.pos);
    Ray o;
    o.p = mat_mul(n.itm, r.p - pos);
    o.dir = mat_mul(n.itm, (r.p + r.dir) - pos) - o.p;
    return o;
}
// Node::FromNodeCoords (scene.h:508-512)
__device__ __forceinline__ void from_node(const DevNode& n, Hit& h) {
    h.p = mat_mul(n.tm, h.p) + ld3(n.pos);
    h.N = norm3(mat_tmul(n.itm, h.N));
}

// ---------------------------------------------------------------------------
// Slab interval of Box::IntersectRay / BVHBoxIntersection (objFunctions.cpp:143-254,
// 408-522). The reference has four branches keyed on the first exactly-zero
// direction component; each branch evaluates the same per-axis quotients and
// only differs in which axes enter max/min, so the quotients are computed
// unconditionally (IEEE: a division by zero cannot trap) and selected.
__device__ __forceinline__ void box_slabs(const Ray& r, f3 bmin, f3 bmax, float& tEntry, float& tExit) {
    float tx0 = (bmin.x - r.p.x) / r.dir.x;
    float tx1 = (bmax.x - r.p.x) / r.dir.x;
    float ty0 = (bmin.y - r.p.y) / r.dir.y;
    float ty1 = (bmax.y - r.p.y) / r.dir.y;
    float tz0 = (bmin.z - r.p.z) / r.dir.z;
    float tz1 = (bmax.z - r.p.z) / r.dir.z;
    if (tx0 > tx1) { float t = tx1; tx1 = tx0; tx0 = t; }
    if (ty0 > ty1) { float t = ty1; ty1 = ty0; ty0 = t; }
    if (tz0 > tz1) { float t = tz1; tz1 = tz0; tz0 = t; }
    if (r.dir.x == 0) {
        tEntry = smax(tz0, ty0);
        tExit = smin(tz1, ty1);
    } else if (r.dir.y == 0) {
        tEntry = smax(tz0, tx0);
        tExit = smin(tz1, tx1);
    } else if (r.dir.z == 0) {
        tEntry = smax(ty0, tx0);
        tExit = smin(ty1, tx1);
    } else {
        tEntry = smax(smax(tx0, ty0), tz0);
        tExit = smin(smin(tx1, ty1), tz1);
    }
}
__device__ __forceinline__ bool box_empty(f3 bmin, f3 bmax) {  // Box::IsEmpty, scene.h:85
    return bmin.x > bmax.x || bmin.y > bmax.y || bmin.z > bmax.z;
}
// Box::IntersectRay(r, t_max) (objFunctions.cpp:143-254)
__device__ __forceinline__ bool box_hit(const Ray& r, f3 bmin, f3 bmax, float t_max) {
    if (box_empty(bmin, bmax)) return false;
    float tEntry, tExit;
    box_slabs(r, bmin, bmax, tEntry, tExit);
    return tEntry <= tExit && tEntry < t_max;
}
// BVHBoxIntersection (objFunctions.cpp:408-522): tEntry + 0.01 in fp64 (:517), or t_max
__device__ __forceinline__ float bvh_box(const Ray& r, f3 bmin, f3 bmax) {
    if (box_empty(bmin, bmax)) return -RTU_BIGFLOAT;
    float tEntry, tExit;
    box_slabs(r, bmin, bmax, tEntry, tExit);
    if (tEntry <= tExit && tEntry < RTU_BIGFLOAT) return (float)((double)tEntry + 0.01);
    return RTU_BIGFLOAT;
}

// ---------------------------------------------------------------------------
// Sphere::IntersectRay (objFunctions.cpp:15-104), including the stale-z
// fall-through of the n<m branch. uvw is not produced (no
// textures on this path).
__device__ __forceinline__ bool sphere_hit(const Ray& ray, Hit& h) {
    if (!box_hit(ray, mk3(-1, -1, -1), mk3(1, 1, 1), RTU_BIGFLOAT)) return false;
    float a = dot3(ray.dir, ray.dir);
    float b = 2 * dot3(ray.p - mk3(0, 0, 0), ray.dir);
    float c = dot3(ray.p, ray.p) - 1;
    float sqrtCheck = b * b - 4 * a * c;
    float sq = sqrtf(sqrtCheck);
    float m = (-b + sq) / (2 * a);
    float n = (-b - sq) / (2 * a);
    bool ret = false;
    if (m == n && m < h.z && (double)m >= 0.001) {
        h.z = m;
        h.front = true;
        ret = true;
    } else if (m < n && m < h.z && (((double)m >= 0.001) | ((double)n >= 0.001))) {
        if ((double)m <= 0.001 && (double)n > 0.001 && n < h.z) {
            h.z = n;
            h.front = false;
        } else if ((double)m > 0.001) {
            h.z = m;
            h.front = true;
        }
        ret = true;
    } else if (n < m && n < h.z && (((double)m >= 0.001) | ((double)n >= 0.001))) {
        if ((double)n <= 0.001 && (double)m > 0.001 && m < h.z) {
            h.z = m;
            h.front = false;
        } else if ((double)n > 0.001) {
            h.z = n;
            h.front = true;
        }
        ret = true;
    }
    if (ret) {
        f3 temp = ray.p + ray.dir * h.z;  // h.z may be stale: reproduced on purpose
        f3 nn = norm3(temp);
        h.N = h.front ? nn : -nn;
        h.p = temp;
    }
    return ret;
}

// Plane::IntersectRay (objFunctions.cpp:107-140)
__device__ __forceinline__ bool plane_hit(const Ray& ray, Hit& h) {
    if (!box_hit(ray, mk3(-1, -1, 0), mk3(1, 1, 0), RTU_BIGFLOAT)) return false;
    if (ray.dir.z != 0) {
        float t = (-ray.p.z) / (ray.dir.z);
        if ((double)t > 0.001 && t < h.z) {
            f3 q = ray.p + ray.dir * t;
            if (q.x > -1 && q.x < 1 && q.y > -1 && q.y < 1) {
                h.front = ray.p.z > 0;
                h.N = mk3(0, 0, h.front ? 1.0f : -1.0f);
                h.z = t;
                h.p = mk3(q.x, q.y, 0);
                return true;
            }
        }
    }
    return false;
}

// Point2::Cross (cyPoint.h:247-249)
__device__ __forceinline__ float cross2(float ax, float ay, float bx, float by) { return (-ay) * bx + ax * by; }

// cyTriMesh::Interpolate (cyTriMesh.h:191)
__device__ __forceinline__ f3 interp(const float* arr, const uint32_t* face, f3 bc) {
    return (ld3(arr + 3 * face[0]) * bc.x + ld3(arr + 3 * face[1]) * bc.y) + ld3(arr + 3 * face[2]) * bc.z;
}

// TriObj::IntersectTriangle (objFunctions.cpp:257-328) on a pre-gathered triangle
// record {A,N.x | B,N.y | C,N.z}.
template <bool STATS>
__device__ __forceinline__ bool tri_hit(const DevMesh& mesh, uint32_t slot, const Ray& ray, Hit& h, Counters& cnt) {
    RTU_CNT(tri);
    float4 r0 = mesh.tri[3 * slot + 0];
    float4 r1 = mesh.tri[3 * slot + 1];
    float4 r2 = mesh.tri[3 * slot + 2];
    f3 A = mk3(r0.x, r0.y, r0.z), B = mk3(r1.x, r1.y, r1.z), C = mk3(r2.x, r2.y, r2.z);
    f3 N = mk3(r0.w, r1.w, r2.w);
    float dn = dot3(ray.dir, N);
    if (dn != 0) {
        float t = dot3(A - ray.p, N) / dn;
        if ((double)t > 0.00001 && t < h.z) {
            f3 q = ray.p + ray.dir * t;
            float anx = fabsf(N.x), any = fabsf(N.y), anz = fabsf(N.z);
            float maxNormalAxis = smax(smax(anx, any), anz);
            float ax, ay, bx, by, cx, cy, qx, qy;
            if (maxNormalAxis == anx) {
                ax = A.y; ay = A.z; bx = B.y; by = B.z; cx = C.y; cy = C.z; qx = q.y; qy = q.z;
            } else if (maxNormalAxis == any) {
                ax = A.x; ay = A.z; bx = B.x; by = B.z; cx = C.x; cy = C.z; qx = q.x; qy = q.z;
            } else {
                ax = A.x; ay = A.y; bx = B.x; by = B.y; cx = C.x; cy = C.y; qx = q.x; qy = q.y;
            }
            // "/2.0" is evaluated in fp64 in the reference (:298-300); halving is exact in
            // binary32 as well except when the result is subnormal, so keep the fp64 form.
            float TriABCArea = (float)((double)cross2(cx - ax, cy - ay, bx - ax, by - ay) / 2.0);
            float TriAPCArea = (float)((double)cross2(cx - ax, cy - ay, qx - ax, qy - ay) / 2.0);
            float TriABPArea = (float)((double)cross2(qx - ax, qy - ay, bx - ax, by - ay) / 2.0);
            float BC1 = TriAPCArea / TriABCArea;
            float BC2 = TriABPArea / TriABCArea;
            float BC3 = (float)(1.0 - (double)BC1 - (double)BC2);  // :304
            if (BC1 > 0 && BC2 > 0 && BC3 > 0 && BC1 < 1 && BC2 < 1 && BC3 < 1) {
                RTU_CNT(acc);
                f3 bc = mk3(BC3, BC1, BC2);
                uint32_t face = mesh.elements[slot];
                h.front = dn < 0;
                h.N = norm3(interp(mesh.vn, mesh.fn + 3 * face, bc));
                h.z = t;
                h.p = interp(mesh.v, mesh.f + 3 * face, bc);
                return true;
            }
        }
    }
    return false;
}

// TriObj::IntersectRay (objFunctions.cpp:333-406). The reference pushes both
// children (far first) and pops; popping the near child right after pushing it is
// the same as continuing with it, so only the far child goes to the LDS stack.
template <int STACK, bool STATS>
__device__ __forceinline__ bool mesh_hit(const DevMesh& mesh, const Ray& ray, Hit& h, uint32_t* stk, Counters& cnt) {
    if (!box_hit(ray, ld3(mesh.bmin), ld3(mesh.bmax), RTU_BIGFLOAT)) return false;
    RTU_CNT(mesh);
    bool hitResult = false;
    int sp = 0;
    uint32_t cur = 1;  // GetRootNodeID, cyBVH.h:76
    for (;;) {
        float4 n0 = mesh.bvh[2 * cur + 0];
        float4 n1 = mesh.bvh[2 * cur + 1];
        uint32_t index = __float_as_uint(n0.w), count = __float_as_uint(n1.w);
        bool pop = true;
        if (count == 0) {
            RTU_CNT(inner);
            uint32_t c1 = index, c2 = index + 1;
            float4 a0 = mesh.bvh[2 * c1 + 0], a1 = mesh.bvh[2 * c1 + 1];
            float4 b0 = mesh.bvh[2 * c2 + 0], b1 = mesh.bvh[2 * c2 + 1];
            float t1 = bvh_box(ray, mk3(a0.x, a0.y, a0.z), mk3(a1.x, a1.y, a1.z));
            float t2 = bvh_box(ray, mk3(b0.x, b0.y, b0.z), mk3(b1.x, b1.y, b1.z));
            bool v1 = t1 != RTU_BIGFLOAT, v2 = t2 != RTU_BIGFLOAT;
            // :361-389: (t1 <= t2) push c2 then c1; else push c1 then c2
            bool firstIsC1 = t1 <= t2;
            uint32_t nearC = firstIsC1 ? c1 : c2, farC = firstIsC1 ? c2 : c1;
            bool nearV = firstIsC1 ? v1 : v2, farV = firstIsC1 ? v2 : v1;
            if (nearV) {
                if (farV) {
                    if (sp < STACK) stk[sp * 64] = farC;
                    sp++;
                }
                cur = nearC;
                pop = false;
            } else if (farV) {
                cur = farC;
                pop = false;
            }
        } else {
            RTU_CNT(leafv);
            if (STATS) cnt.leafe += count;
            for (uint32_t i = 0; i < count; i++)  // :394-396
                hitResult |= tri_hit<STATS>(mesh, index + i, ray, h, cnt);
        }
        if (pop) {
            if (sp == 0) break;
            sp--;
            cur = stk[sp * 64];
        }
    }
    return hitResult;
}

// ---------------------------------------------------------------------------
// Trace / ShadowTrace (RenderFunctions.cpp:181-240), recursion over the node tree
// flattened to a pre-order loop. Only h.z (and h.front in the sphere quirk) feeds
// later intersection tests, so applying FromNodeCoords for the hit node and all of
// its ancestors immediately is equivalent to the reference applying them as the
// recursion unwinds.
//
// ONE instantiation serves both kinds of ray: `shadow` is a per-lane flag, so
// lanes casting shadow rays and lanes casting reflection / refraction rays walk
// the scene together (better SIMD occupancy, a quarter of the code size of four
// specialised copies — the kernel has to stay inside the instruction cache).
template <int STACK, bool STATS>
__device__ __forceinline__ bool trace(const DevScene& s, const Ray& wr, bool shadow, Hit& h, uint32_t* stk, Counters& cnt) {
    bool any = false;
    Ray r0 = to_node(s.nodes[0], wr);  // ray inside the root node
    Ray rp = r0;                       // ray inside node `rp_node` (cached parent space)
    int rp_node = 0;
    for (uint32_t k = 0; k < s.n_nodes; k++) {
        const DevNode& n = s.nodes[k];
        if (n.obj_type == RTU_OBJ_NONE) continue;
        if (shadow && any) continue;  // ShadowTrace returns at the first occluder (:223-225)
        int parent = n.parent;
        Ray pr;
        if (parent < 0) {
            pr = wr;
        } else {
            if (parent != rp_node) {
                const DevNode& pn = s.nodes[parent];
                Ray t = r0;
                for (int d = 1; d <= pn.depth; d++) t = to_node(s.nodes[pn.chain[d]], t);
                rp = t;
                rp_node = parent;
            }
            pr = rp;
        }
        Ray lr = to_node(n, pr);
        RTU_CNT(node);
        bool hit;
        if (n.obj_type == RTU_OBJ_SPHERE) hit = sphere_hit(lr, h);
        else if (n.obj_type == RTU_OBJ_PLANE) hit = plane_hit(lr, h);
        else hit = mesh_hit<STACK, STATS>(s.meshes[n.mesh_id], lr, h, stk, cnt);
        if (hit) {
            any = true;
            if (!shadow) {
                h.node = (int)k;
                from_node(n, h);
                for (int j = parent; j >= 0; j = s.nodes[j].parent) from_node(s.nodes[j], h);
            }
        }
    }
    return any;
}

// sampledNormal of mtlFunctions.cpp:162-165 / :275-277 with SampleSphere(...,0) == (0,0,0)
__device__ __forceinline__ f3 sampled_normal(f3 p, f3 N) {
    f3 sampleOrigin = p + N;
    return norm3((sampleOrigin + mk3(0, 0, 0)) - p);
}
__device__ __forceinline__ f3 reflect_dir(f3 dir, f3 sn) {  // :207, :239, :280
    float k = 2 * dot3(dir, sn);
    return norm3(dir - sn * k);
}

// Snell / Fresnel terms of mtlFunctions.cpp:168-203,236-237. Recomputed from the
// frame whenever a stage resumes (pure ALU) instead of being saved.
struct Refr {
    f3    sn;          // sampled normal
    float cosTheta1;   // after clamping
    float sinTheta2, cosTheta2;
    float n1, n2;
    f3    SVector;
};
__device__ __forceinline__ Refr refraction_terms(f3 dir, f3 p, f3 N, bool front, float ior) {
    Refr r;
    r.sn = sampled_normal(p, N);
    float cosTheta1 = dot3(r.sn, -dir);
    float sinTheta1 = (float)sqrt(1 - (double)cosTheta1 * (double)cosTheta1);  // :169 (pow(x,2) is exact in fp64)
    if (sinTheta1 > 1) sinTheta1 = 1.0f;
    if (sinTheta1 < -1) sinTheta1 = -1.0f;
    if (cosTheta1 > 1) cosTheta1 = 1.0f;
    if (cosTheta1 < -1) cosTheta1 = -1.0f;
    r.cosTheta1 = cosTheta1;
    r.n1 = ior;
    r.n2 = 1.0f;
    if (front) { r.n1 = 1.0f; r.n2 = ior; }
    r.sinTheta2 = (r.n1 / r.n2) * sinTheta1;
    r.cosTheta2 = sqrtf(1 - r.sinTheta2 * r.sinTheta2);  // :197
    if (r.cosTheta2 > 1) r.cosTheta2 = 1.0f;
    r.SVector = norm3(cross3(r.sn, norm3(cross3(r.sn, -dir))));  // :203
    return r;
}
__device__ __forceinline__ float schlick(const Refr& r) {  // :236-237
    float q = (r.n1 - r.n2) / (r.n1 + r.n2);
    float R0 = (float)((double)q * (double)q);
    double x = 1.0 - (double)r.cosTheta1;
    double x5 = x * x * x * x * x;  // pow(x,5); affects colour only (tolerance +-1/255)
    return (float)((double)R0 + (1.0 - (double)R0) * x5);
}
__device__ __forceinline__ f3 absorb(float z, f3 absorption) {  // :213-215, :259-261
    return mk3(expf((-z) * absorption.x), expf((-z) * absorption.y), expf((-z) * absorption.z));
}

// Stage of a Shade() frame = where the lane resumes.
enum Stage {
    ST_PRIMARY = 0,     // fire the primary ray
    ST_LIGHT,           // direct lighting loop over lights, mtlFunctions.cpp:125-155
    ST_REFR_START,      // :160
    ST_TIR_RET,         // child = TIR-reflected hit, :217-221
    ST_REFR_B_RET,      // child = refracted hit ("refractionResult"), :254
    ST_REFR_A_RET,      // child = Fresnel-reflected hit ("frenselResult"), :247
    ST_REFL_START,      // :273
    ST_REFL_RET,        // child = mirror-reflected hit, :286
    ST_DONE
};
// What the ray in flight is for (decides how its result is consumed).
enum Pending { P_NONE = 0, P_PRIMARY, P_SHADOW, P_TIR, P_REFR_B, P_REFR_A, P_REFL };

struct Frame {
    f3    dir, p, N;       // incoming ray direction, hit point, hit normal (world)
    f3    result;          // partial sum of Shade()
    f3    term1;           // pending refraction term absV*refr*refractionResult*(1-S)
    float aux;             // z of the refracted hit (for absorption)
    int   mtl;             // material id
    int   bounce;
    int   stage;
    bool  front;
    bool  aux_front;       // front flag of the refracted hit
};

__device__ __forceinline__ void frame_store(float* arena, uint32_t n_threads, uint32_t tid, int level, const Frame& f) {
    float* b = arena + (size_t)level * RTU_FRAME_FIELDS * n_threads + tid;
    const int packed = (f.mtl << 10) | (f.bounce << 6) | (f.stage << 2) | (f.front ? 2 : 0) | (f.aux_front ? 1 : 0);
    const float v[RTU_FRAME_FIELDS] = {f.dir.x, f.dir.y, f.dir.z, f.p.x, f.p.y, f.p.z, f.N.x, f.N.y, f.N.z,
                                       f.result.x, f.result.y, f.result.z, f.term1.x, f.term1.y, f.term1.z, f.aux,
                                       __int_as_float(packed)};
#pragma unroll
    for (int i = 0; i < RTU_FRAME_FIELDS; i++) b[(size_t)i * n_threads] = v[i];
}
__device__ __forceinline__ void frame_load(const float* arena, uint32_t n_threads, uint32_t tid, int level, Frame& f) {
    const float* b = arena + (size_t)level * RTU_FRAME_FIELDS * n_threads + tid;
    float v[RTU_FRAME_FIELDS];
#pragma unroll
    for (int i = 0; i < RTU_FRAME_FIELDS; i++) v[i] = b[(size_t)i * n_threads];
    f.dir = mk3(v[0], v[1], v[2]);
    f.p = mk3(v[3], v[4], v[5]);
    f.N = mk3(v[6], v[7], v[8]);
    f.result = mk3(v[9], v[10], v[11]);
    f.term1 = mk3(v[12], v[13], v[14]);
    f.aux = v[15];
    int packed = __float_as_int(v[16]);
    f.mtl = packed >> 10;
    f.bounce = (packed >> 6) & 15;
    f.stage = (packed >> 2) & 15;
    f.front = (packed & 2) != 0;
    f.aux_front = (packed & 1) != 0;
}

template <int STACK, bool STATS>
__global__ void __launch_bounds__(64) render_kernel(KernelArgs a) {
    __shared__ uint32_t s_stack[STACK * 64];
    const DevScene& s = a.scene;
    const uint32_t lane = threadIdx.x;
    const uint32_t tile = blockIdx.x;
    const uint32_t band_local = tile / a.tiles_x;
    const uint32_t tx = tile - band_local * a.tiles_x;
    const int x = (int)(tx * 8 + (lane & 7));
    const int ly = (int)(band_local * RTU_BAND_ROWS + (lane >> 3));                                              // row inside the shard
    const int y = (int)((band_local * a.frame.shard_count + a.frame.shard_rank) * RTU_BAND_ROWS + (lane >> 3));  // global row
    uint32_t* stk = s_stack + lane;
    const uint32_t tid = blockIdx.x * 64 + lane;
    Counters cnt = {};
    const bool valid = x < a.frame.width && y < a.frame.height;
    const f3 cam_pos = ld3(a.frame.cam_pos);
    const f3 env = ld3(s.environment);

    // ---- per-lane state of the explicit Shade() recursion -------------------------
    Frame F;
    F.dir = F.p = F.N = F.result = F.term1 = mk3(0, 0, 0);
    F.aux = 0; F.mtl = 0; F.bounce = 0; F.front = true; F.aux_front = true;
    F.stage = ST_PRIMARY;
    int level = 0;
    uint32_t li = 0;             // next light of the ST_LIGHT loop
    f3 ret = mk3(0, 0, 0);       // value returned by the child frame that just finished
    f3 color = mk3(0, 0, 0);     // final pixel colour
    float zprim = RTU_BIGFLOAT;  // hInfo.z of the primary ray
    f3 lightK = mk3(0, 0, 0);    // diffuse + specular*pow(N.H, gloss) of the light whose shadow ray is in flight
    float lightNDotL = 0;
    bool fin = !valid;

    while (!fin) {
        // ======== phase A: run the lane's frame until it needs a ray (or finishes) ========
        int pend = P_NONE;
        Ray nr;
        nr.p = F.p; nr.dir = mk3(0, 0, 0);
        float tmax = RTU_BIGFLOAT;
        while (pend == P_NONE && !fin) {
            if (F.stage == ST_PRIMARY) {
                // RenderFunctions.cpp:258-268 (pixel centre), :97
                f3 cp = (ld3(a.frame.origin) + ld3(a.frame.u) * ((float)x + 0.5f)) + ld3(a.frame.v) * ((float)y + 0.5f);
                nr.p = cam_pos;
                nr.dir = norm3(cp - cam_pos);
                RTU_CNT(prim);
                pend = P_PRIMARY;
            } else if (F.stage == ST_LIGHT) {
                if (!F.front || li >= s.n_lights) {  // :125 only front faces are lit
                    F.stage = ST_REFR_START;
                    continue;
                }
                const RtuLight& l = s.lights[li];
                const RtuMaterial& m = s.materials[F.mtl];
                f3 intensity = ld3(l.intensity);
                f3 diffuse = ld3(m.diffuse);
                if (l.type == RTU_LIGHT_AMBIENT) {
                    F.result = F.result + diffuse * intensity;  // :132
                    li++;
                    continue;
                }
                f3 viewDirection = norm3(cam_pos - F.p);  // :137
                f3 lvec = ld3(l.vec);
                bool isDirect = l.type == RTU_LIGHT_DIRECT;
                f3 ldir = isDirect ? lvec : norm3(F.p - lvec);          // Direction(), lights.h:49,83
                f3 lightDirection = norm3(-ldir);                      // :138
                f3 halfVector = norm3(viewDirection + lightDirection);  // :139
                float NDotL = dot3(F.N, lightDirection);
                float NDotH = dot3(F.N, halfVector);
                if (NDotL < 0.0f) NDotL = 0.0f;
                if (NDotH < 0.0f) NDotH = 0.0f;
                lightNDotL = NDotL;
                lightK = diffuse + ld3(m.specular) * powf(NDotH, m.glossiness);  // :152
                nr.p = F.p;
                if (isDirect) {
                    nr.dir = -lvec;  // lights.h:48
                    tmax = RTU_BIGFLOAT;
                } else {
                    nr.dir = norm3(lvec - F.p);  // lightFunctions.cpp:76
                    tmax = len3(lvec - F.p);     // :78
                }
                RTU_CNT(shd);
                pend = P_SHADOW;
            } else if (F.stage == ST_REFR_START) {
                const RtuMaterial& m = s.materials[F.mtl];
                if (F.bounce > 0 && not_black(ld3(m.refraction))) {
                    Refr r = refraction_terms(F.dir, F.p, F.N, F.front, m.ior);
                    if (r.sinTheta2 > 1) {  // total internal reflection, :205
                        nr.dir = reflect_dir(F.dir, r.sn);
                        pend = P_TIR;
                    } else {
                        nr.dir = norm3((-r.sn) * r.cosTheta2 + r.SVector * r.sinTheta2);  // :229
                        pend = P_REFR_B;
                    }
                    nr.p = F.p;
                    RTU_CNT(sec);
                } else {
                    F.stage = (F.bounce > 0) ? ST_REFL_START : ST_DONE;
                }
            } else if (F.stage == ST_TIR_RET) {
                const RtuMaterial& m = s.materials[F.mtl];
                f3 absorptionV = absorb(RTU_BIGFLOAT, ld3(m.absorption));  // z of a fresh HitInfo, :210-215
                F.result = F.result + absorptionV * ret;                    // :219-221
                F.stage = ST_REFL_START;
            } else if (F.stage == ST_REFR_B_RET) {
                const RtuMaterial& m = s.materials[F.mtl];
                Refr r = refraction_terms(F.dir, F.p, F.N, F.front, m.ior);
                float S = schlick(r);
                f3 absorptionV = mk3(1, 1, 1);
                if (!F.aux_front) absorptionV = absorb(F.aux, ld3(m.absorption));  // :258-262
                F.term1 = ((absorptionV * ld3(m.refraction)) * ret) * (float)(1.0 - (double)S);
                nr.p = F.p;
                nr.dir = reflect_dir(F.dir, r.sn);  // Fresnel reflection ray, :239
                RTU_CNT(sec);
                pend = P_REFR_A;
            } else if (F.stage == ST_REFR_A_RET) {
                const RtuMaterial& m = s.materials[F.mtl];
                Refr r = refraction_terms(F.dir, F.p, F.N, F.front, m.ior);
                float S = schlick(r);
                f3 frenselResult = ld3(m.refraction) * ret;           // :247
                F.result = F.result + (F.term1 + frenselResult * S);  // :264
                F.stage = ST_REFL_START;
            } else if (F.stage == ST_REFL_START) {
                const RtuMaterial& m = s.materials[F.mtl];
                if (not_black(ld3(m.reflection))) {  // :273 (bounce > 0 is implied by reaching this stage)
                    f3 sn = sampled_normal(F.p, F.N);
                    nr.p = F.p;
                    nr.dir = reflect_dir(F.dir, sn);  // :280
                    RTU_CNT(sec);
                    pend = P_REFL;
                } else {
                    F.stage = ST_DONE;
                }
            } else if (F.stage == ST_REFL_RET) {
                const RtuMaterial& m = s.materials[F.mtl];
                F.result = F.result + ld3(m.reflection) * ret;  // :286
                F.stage = ST_DONE;
            } else {  // ST_DONE: return to the caller frame
                ret = F.result;
                if (level == 0) {
                    color = ret;
                    fin = true;
                } else {
                    level--;
                    frame_load(a.arena, a.n_threads, tid, level, F);
                }
            }
        }
        if (fin) break;

        // ======== phase B: the one ray-scene intersection site of the kernel ========
        Hit h;
        h.z = tmax; h.front = true; h.node = -1; h.p = mk3(0, 0, 0); h.N = mk3(0, 0, 0);
        const bool is_shadow = pend == P_SHADOW;
        const bool hit = trace<STACK, STATS>(s, nr, is_shadow, h, stk, cnt);

        // ======== phase C: consume the result ========
        if (is_shadow) {
            // GenLight::Shadow (lightFunctions.cpp:27-37) + Illuminate (lights.h:48, lightFunctions.cpp:75-83)
            float sh = (hit && h.z > 0.0f) ? 0.0f : 1.0f;
            const RtuLight& l = s.lights[li];
            f3 intensity = ld3(l.intensity);
            f3 illum;
            if (l.type == RTU_LIGHT_DIRECT) {
                illum = intensity * sh;
            } else {
                f3 d = ld3(l.vec) - F.p;
                illum = (intensity * (0.0f + sh)) * (1 / dot3(d, d));  // :83
            }
            F.result = F.result + (illum * lightNDotL) * lightK;  // mtlFunctions.cpp:152
            li++;
        } else if (pend == P_PRIMARY) {
            zprim = h.z;
            if (!hit) {
                color = ld3(s.background);  // RenderFunctions.cpp:145
                fin = true;
            } else {
                RTU_CNT(prim_hit);
                int mid = s.nodes[h.node].material_id;
                if (mid < 0) {
                    color = mk3(1, 1, 1);  // null material => white
                    fin = true;
                } else {
                    F.dir = nr.dir; F.p = h.p; F.N = h.N; F.front = h.front;
                    F.mtl = mid; F.bounce = a.frame.max_bounce;
                    F.result = mk3(0, 0, 0);
                    F.stage = ST_LIGHT;
                    li = 0;
                }
            }
        } else if (hit) {
            // a secondary ray hit: Shade() of the hit node becomes the active frame
            int ret_stage = pend == P_TIR ? ST_TIR_RET : pend == P_REFR_B ? ST_REFR_B_RET : pend == P_REFR_A ? ST_REFR_A_RET : ST_REFL_RET;
            if (pend == P_REFR_B) { F.aux = h.z; F.aux_front = h.front; }
            F.stage = ret_stage;
            int cmid = s.nodes[h.node].material_id;
            if (cmid < 0) {
                ret = mk3(1, 1, 1);  // null material (the reference would crash here)
            } else {
                frame_store(a.arena, a.n_threads, tid, level, F);
                level++;
                int cb = F.bounce - 1;
                F.dir = nr.dir; F.p = h.p; F.N = h.N; F.front = h.front;
                F.mtl = cmid; F.bounce = cb;
                F.result = mk3(0, 0, 0); F.term1 = mk3(0, 0, 0); F.aux = 0; F.aux_front = true;
                F.stage = ST_LIGHT;
                li = 0;
            }
        } else {
            // a secondary ray missed
            if (pend == P_TIR) {
                F.stage = ST_REFL_START;  // :217 has no else branch
            } else if (pend == P_REFR_B) {
                F.result = F.result + env;  // :267
                F.stage = ST_REFL_START;
            } else if (pend == P_REFR_A) {
                const RtuMaterial& m = s.materials[F.mtl];
                Refr r = refraction_terms(F.dir, F.p, F.N, F.front, m.ior);
                float S = schlick(r);
                F.result = F.result + (F.term1 + env * S);  // :250, :264
                F.stage = ST_REFL_START;
            } else {
                const RtuMaterial& m = s.materials[F.mtl];
                F.result = F.result + env * ld3(m.reflection);  // :289
                F.stage = ST_DONE;
            }
        }
    }

    if (valid) a.out[(size_t)ly * a.frame.width + x] = make_float4(color.x, color.y, color.z, zprim);

    if (STATS) {
        // wave reduction, then one atomic per counter per wave
        unsigned vals[11] = {cnt.prim, cnt.prim_hit, cnt.sec, cnt.shd, cnt.node, cnt.mesh,
                             cnt.inner, cnt.leafv, cnt.leafe, cnt.tri, cnt.acc};
#pragma unroll
        for (int i = 0; i < 11; i++) {
            unsigned v = vals[i];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
            if (lane == 0 && v) atomicAdd(&a.counters[i], (unsigned long long)v);
        }
    }
}

}  // namespace

int rtu_launch_render(const KernelArgs& args, uint32_t n_blocks, uint32_t bvh_stack_needed, bool stats, hipStream_t stream) {
    if (n_blocks == 0) return (int)hipSuccess;
    dim3 grid(n_blocks), block(64);
#define RTU_LAUNCH(S)                                                                               \
    do {                                                                                            \
        if (stats) hipLaunchKernelGGL((render_kernel<S, true>), grid, block, 0, stream, args);      \
        else hipLaunchKernelGGL((render_kernel<S, false>), grid, block, 0, stream, args);           \
    } while (0)
    if (bvh_stack_needed <= 16) RTU_LAUNCH(16);
    else if (bvh_stack_needed <= 32) RTU_LAUNCH(32);
    else RTU_LAUNCH(RTU_MAX_BVH_STACK);
#undef RTU_LAUNCH
    return (int)hipGetLastError();
}
