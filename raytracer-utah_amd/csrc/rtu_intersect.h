// rtu_intersect.h — device-side restatement of the reference's ray/scene arithmetic:
// everything a single ray needs (node transforms, box / sphere / plane / triangle
// tests, the BVH walk, the scene-graph walk) plus the Snell / Fresnel terms of
// MtlBlinn::Shade. Shared by every kernel in render_kernel.hip.
//
// What replaces what (reference file:line):
//   to_node / from_node           Node::ToNodeCoords/FromNodeCoords  scene.h:501-512
//   box_slabs / box_hit           Box::IntersectRay, BVHBoxIntersection  objFunctions.cpp:143-254,408-522
//   sphere_hit / plane_hit        Sphere/Plane::IntersectRay         objFunctions.cpp:15-140
//   mesh_hit / tri_hit            TriObj::IntersectRay/IntersectTriangle  objFunctions.cpp:257-406
//   trace                         Trace / ShadowTrace                RenderFunctions.cpp:181-240
//   refraction_terms / schlick    MtlBlinn::Shade                    mtlFunctions.cpp:160-203,236-237
//
// Bit parity: compiled -ffp-contract=off with IEEE divide/sqrt; every expression
// keeps the reference's order and its float->double promotions.
#ifndef RTU_INTERSECT_H_INCLUDED
#define RTU_INTERSECT_H_INCLUDED

#include "rtu_device.h"

namespace {

struct Ray {
    f3 p, dir;
};

struct Hit {  // HitInfo (scene.h:150-163); uvw is maintained for textured scenes only
    float z;
    f3    p, N, uvw;
    int   node;
    bool  front;
};

__device__ __forceinline__ void fresh_hit(Hit& h, float tmax) {  // HitInfo::Init, scene.h:162
    h.z = tmax; h.front = true; h.node = -1; h.p = mk3(0, 0, 0); h.N = mk3(0, 0, 0);
}

struct Counters {
    unsigned prim, prim_hit, sec, shd, node, mesh, inner, leafv, leafe, tri, acc;
    // touched-bytes mode of the FAST variant (collect_stats == 2, RtuTouched in rtu_render.h): what the timed
    // kernels themselves read and write — their own trees, their own culling, their own two stages
    unsigned t_rays, t_node, t_meshbox, t_inner4, t_inner8, t_innerref, t_tri, t_win, t_xform, t_bytes, t_bounds;
    unsigned t_inline;  // of t_rays: shadow rays of childless Shade() calls settled by the lane that found the hit (render_impl.h shadows_inline)
};

#define RTU_CNT(field) do { if (STATS) cnt.field++; } while (0)
// FC: the instantiation counts what it touches; `fc_lane`: this lane does the counting (one lane of the eight that share
// a ray in the cooperative kernels)
#define RTU_TOUCH(field, n) do { if (FC && fc_lane) cnt.field += (n); } while (0)
// ... for WAVE-UNIFORM data (scene nodes, their bounds, mesh headers, screen rectangles: read through the constant address space by
// scalar loads, once per wavefront whatever the number of lanes that need them): counted once per wavefront, by its first active lane
#define RTU_TOUCH_WAVE(field, n) do { if (FC && (threadIdx.x & 63u) == (uint32_t)__ffsll((long long)__ballot(1)) - 1u) cnt.field += (n); } while (0)

// ---------------------------------------------------------------------------
// Node::ToNodeCoords (scene.h:501-507): p' = itm*(p-pos); d' = itm*((p+d)-pos) - p'
template <class NodeT>
__device__ __forceinline__ Ray to_node(const NodeT& n, const Ray& r) {
    f3 pos = ld3(n.pos);
    Ray o;
    o.p = mat_mul(n.itm, r.p - pos);
    o.dir = mat_mul(n.itm, (r.p + r.dir) - pos) - o.p;
    return o;
}
// Node::FromNodeCoords (scene.h:508-512)
template <class NodeT>
__device__ __forceinline__ void from_node(const NodeT& n, Hit& h) {
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
// ---- exact division by a per-ray constant ------------------------------------------
// The slab test divides by the same three direction components at every BVH node.
// IEEE binary32 division is ~10 VALU instructions on CDNA4; instead the reciprocal is
// taken ONCE per ray in binary64 and each quotient is
//        q = (float)((double)n * rd),   rd = 1.0 / (double)d
// which is bit-identical to the binary32 quotient n / d for EVERY input:
//  * rd and the product carry a relative error <= 2^-52 in total;
//  * the exact quotient of two binary32 numbers is either representable or at least
//    2^-49 (relative) away from every rounding boundary of binary32 (a midpoint m has
//    a 25-bit significand M; n - d*m = (N*2^s - D*M) * 2^k is a non-zero integer multiple
//    of 2^k while |d*m| < 2^(k+49)), so the binary64 value rounds to the same float;
//  * zeros, infinities and NaN propagate identically (1/±0 = ±inf, 0*inf = NaN, ...).
// tests/test_gpu_parity.py::test_exact_division checks it against `/` on the GPU.
struct RayRcp {
    double rx, ry, rz;
};
__device__ __forceinline__ float fdiv(float n, double rd) { return (float)((double)n * rd); }
__device__ __forceinline__ RayRcp ray_rcp(const Ray& r) {
    RayRcp c;
    c.rx = 1.0 / (double)r.dir.x;
    c.ry = 1.0 / (double)r.dir.y;
    c.rz = 1.0 / (double)r.dir.z;
    return c;
}
// ---- culling (fast variant only) -----------------------------------------------------------
// A BVH child is skipped when, on SOME axis k, the box is geometrically separated from the
// part of the ray that can still produce an accepted hit by more than delta = 1e-4 * scale
// (scale = largest coordinate magnitude of the mesh and the ray origin):
//   near side: the ray reaches the box's k-slab only after t = h.z      (entry_k - h.z) * |dir_k| > delta
//   behind   : the box's k-slab ends before the ray origin              (0 - exit_k)    * |dir_k| > delta
// Why this cannot change a result: the reference accepts a triangle only with 0.00001 < t < h.z
// and its hit point q = p + dir*t projected inside the triangle on two axes (objFunctions.cpp:270-306).
// With the separation on a projected axis, q lies outside the box (hence outside the triangle) by
// delta >> the rounding of q (~2e-7*scale). With the separation on the axis of the dominant
// normal component (|N_k| >= 0.577), the plane residual |dn*t - N.(A-p)| at that t would have to
// exceed 0.577*delta, while for the COMPUTED t it is bounded by the rounding of the two dot
// products and the division, ~1.5e-6*scale. delta leaves a 40x reserve. (An axis with
// dir_k == 0 never culls: lim_k = inf.) The reference itself walks such boxes (its box test has no
// `tExit >= 0` and no comparison with h.z) and rejects their triangles.
struct CullLim {
    float lx, ly, lz;  // delta / |dir_k|
};
__device__ __forceinline__ CullLim cull_limits(const Ray& r, float scale) {
    const float delta = 1e-4f * scale;
    CullLim c;
    c.lx = delta / fabsf(r.dir.x);
    c.ly = delta / fabsf(r.dir.y);
    c.lz = delta / fabsf(r.dir.z);
    return c;
}
// General case of the slab test (no exactly-zero direction component), objFunctions.cpp:223-245.
// `skip` (CULL only): the box cannot contain an acceptable triangle, see above.
template <bool CULL>
__device__ __forceinline__ void box_slabs_rcp(const Ray& r, const RayRcp& c, f3 bmin, f3 bmax, float hz, const CullLim& L,
                                              float& tEntry, float& tExit, bool& skip) {
    float tx0 = fdiv(bmin.x - r.p.x, c.rx), tx1 = fdiv(bmax.x - r.p.x, c.rx);
    float ty0 = fdiv(bmin.y - r.p.y, c.ry), ty1 = fdiv(bmax.y - r.p.y, c.ry);
    float tz0 = fdiv(bmin.z - r.p.z, c.rz), tz1 = fdiv(bmax.z - r.p.z, c.rz);
    bool sx = tx0 > tx1, sy = ty0 > ty1, sz = tz0 > tz1;
    float ax0 = sx ? tx1 : tx0, ax1 = sx ? tx0 : tx1;
    float ay0 = sy ? ty1 : ty0, ay1 = sy ? ty0 : ty1;
    float az0 = sz ? tz1 : tz0, az1 = sz ? tz0 : tz1;
    tEntry = smax(smax(ax0, ay0), az0);
    tExit = smin(smin(ax1, ay1), az1);
    skip = false;
    if (CULL)
        skip = (ax0 - L.lx > hz) || (ay0 - L.ly > hz) || (az0 - L.lz > hz) || (ax1 < -L.lx) || (ay1 < -L.ly) || (az1 < -L.lz);
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

// ---------------------------------------------------------------------------
// Sphere::IntersectRay (objFunctions.cpp:15-104), including the stale-z
// fall-through of the n<m branch. uvw is not produced (no
// textures on this path).
//
// LITERAL = the reference's order: bounding-box test first (:17), then the quadratic.
// The fast form evaluates the same expressions in a cheaper ORDER (all of them are pure):
//  * a negative (or NaN) discriminant makes m and n NaN and every branch of :29-:99 false
//    whatever the box test says -> return before its six divisions;
//  * the unit sphere lies inside its bounding box, so a ray that really crosses the sphere over a
//    parameter interval [n, m] crosses every slab over a superset of it, and the float slab
//    bounds are off by <= 2 ulp each: the box test can fail only if the chord is shorter than
//    that rounding — or if the float discriminant is itself rounding noise (b*b and 4ac cancel:
//    absolute error ~4 ulp of b*b). With sqrt(disc) > 4e-3*|b| the discriminant exceeds its own
//    error 100 times, the true chord exists and is >= 2e-3 of the distance to it, four orders of
//    magnitude above the slab rounding: the box test is known to pass and is skipped. Otherwise
//    (grazing, far away, NaN/inf) it is evaluated as in the reference.
// rtu_selftest_primitives compares both forms bit for bit on random and grazing rays.
// The reference compares floats with the double literals 0.001 and 0.00001 (float -> double
// promotion). Neither is a float, so every float is strictly on one side and the
// comparison can be done in binary32 against the neighbouring float: with c = 0.001f (the
// smallest float above 0.001) x >= 0.001 and x > 0.001 both mean x >= c, and x <= 0.001 means
// x < c; with e = 0.00001f (the largest float below 0.00001) x > 0.00001 means x > e. NaN is false
// on both sides. A binary64 compare costs a conversion and a quarter-rate instruction.
// (tests/test_host.py checks the neighbours; rtu_selftest_primitives runs both forms.)
template <bool LITERAL> __device__ __forceinline__ bool ge_001(float x) { return LITERAL ? (double)x >= 0.001 : x >= 0.001f; }
template <bool LITERAL> __device__ __forceinline__ bool gt_001(float x) { return LITERAL ? (double)x > 0.001 : x >= 0.001f; }
template <bool LITERAL> __device__ __forceinline__ bool le_001(float x) { return LITERAL ? (double)x <= 0.001 : x < 0.001f; }
__device__ __forceinline__ bool gt_00001(float x) { return x > 0.00001f; }

template <bool LITERAL>
__device__ __forceinline__ bool sphere_hit_t(const Ray& ray, Hit& h, bool tex = false) {
    if (LITERAL && !box_hit(ray, mk3(-1, -1, -1), mk3(1, 1, 1), RTU_BIGFLOAT)) return false;
    float a = dot3(ray.dir, ray.dir);
    float b = 2 * dot3(ray.p - mk3(0, 0, 0), ray.dir);
    float c = dot3(ray.p, ray.p) - 1;
    float sqrtCheck = b * b - 4 * a * c;
    if (!LITERAL) {
        // Branch-light form for a 64-wide wavefront (measured: half of the stage-1 kernels' vector
        // instructions were moves, selects and compares around per-lane branches). The decision
        // tree of :28-:99 is symmetric in (m, n): with lo / hi the smaller / larger root it reads
        //   m == n         : hit at m            if m < h.z and m >= 0.001
        //   m != n, ordered: "hit" (ret)         if lo < h.z and (lo >= 0.001 or hi >= 0.001)
        //                    h.z = hi (back face) if lo <= 0.001 < hi < h.z; h.z = lo (front) if lo > 0.001;
        //                    else h.z and h.front stay as they were — the stale fall-through
        // evaluated here as predicates; one wave-uniform early exit, one rare box test.
        bool cand = sqrtCheck >= 0;
        if (!__any(cand)) return false;
        const float sq = sqrtf(sqrtCheck);
        const float m = (-b + sq) / (2 * a);
        const float n = (-b - sq) / (2 * a);
        const bool needBox = cand && !(sq > 4e-3f * fabsf(b));
        if (__any(needBox)) {
            const bool bh = box_hit(ray, mk3(-1, -1, -1), mk3(1, 1, 1), RTU_BIGFLOAT);
            cand = cand && (!needBox || bh);
        }
        const bool lt = m < n, ordered = lt || n < m;
        const float lo = lt ? m : n, hi = lt ? n : m;
        const bool retEq = m == n && m < h.z && ge_001<false>(m);
        const bool condB = ordered && lo < h.z && (ge_001<false>(lo) || ge_001<false>(hi));
        const bool takeHi = condB && le_001<false>(lo) && gt_001<false>(hi) && hi < h.z;
        const bool takeLo = condB && !takeHi && gt_001<false>(lo);
        const bool ret = cand && (retEq || condB);
        if (ret) {
            if (retEq) { h.z = m; h.front = true; }
            else if (takeHi) { h.z = hi; h.front = false; }
            else if (takeLo) { h.z = lo; h.front = true; }
            f3 temp = ray.p + ray.dir * h.z;  // h.z may be stale: reproduced on purpose
            f3 nn = norm3(temp);
            h.N = h.front ? nn : -nn;
            h.p = temp;
            if (tex) {  // :38-41: atan2f / asinf in binary32, the rest in binary64
                const float u = (float)(0.5 - (double)atan2f(h.N.x, h.N.y) / (2 * 3.14159265358979323846));
                const float v = (float)(0.5 + (double)asinf(h.N.z) / 3.14159265358979323846);
                h.uvw = mk3(u, v, 0);
            }
        }
        return ret;
    }
    float sq = sqrtf(sqrtCheck);
    float m = (-b + sq) / (2 * a);
    float n = (-b - sq) / (2 * a);
    bool ret = false;
    if (m == n && m < h.z && ge_001<LITERAL>(m)) {
        h.z = m;
        h.front = true;
        ret = true;
    } else if (m < n && m < h.z && (ge_001<LITERAL>(m) || ge_001<LITERAL>(n))) {
        if (le_001<LITERAL>(m) && gt_001<LITERAL>(n) && n < h.z) {
            h.z = n;
            h.front = false;
        } else if (gt_001<LITERAL>(m)) {
            h.z = m;
            h.front = true;
        }
        ret = true;
    } else if (n < m && n < h.z && (ge_001<LITERAL>(m) || ge_001<LITERAL>(n))) {
        if (le_001<LITERAL>(n) && gt_001<LITERAL>(m) && m < h.z) {
            h.z = m;
            h.front = false;
        } else if (gt_001<LITERAL>(n)) {
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
__device__ __forceinline__ bool sphere_hit(const Ray& ray, Hit& h, bool tex = false) { return sphere_hit_t<false>(ray, h, tex); }

// Plane::IntersectRay (objFunctions.cpp:107-140). Fast form: the bounding-box test (:109) is
// evaluated last and only when it can matter. The box is the unit square itself (z from 0 to 0),
// its z slab is [t, t] with the very t of :111 (0 - p.z == -p.z exactly), so the test asks
// whether t lies inside the x and y slabs, i.e. whether q = p + dir*t lies inside the square —
// which :114 has just established in floating point. The two can disagree only within the
// rounding of q and of the slab bounds, <= 4 ulp * (1 + |p|): the box test is evaluated when q is
// within 1e-5 * (1 + |p|) of an edge (40x reserve), otherwise it is known to pass.
template <bool LITERAL>
__device__ __forceinline__ bool plane_hit_t(const Ray& ray, Hit& h, bool tex = false) {
    if (LITERAL && !box_hit(ray, mk3(-1, -1, 0), mk3(1, 1, 0), RTU_BIGFLOAT)) return false;
    if (ray.dir.z != 0) {
        float t = (-ray.p.z) / (ray.dir.z);
        if (gt_001<LITERAL>(t) && t < h.z) {
            f3 q = ray.p + ray.dir * t;
            if (q.x > -1 && q.x < 1 && q.y > -1 && q.y < 1) {
                if (!LITERAL) {
                    const bool edge = !(fabsf(q.x) < 1.0f - 1e-5f * (1.0f + fabsf(ray.p.x))) || !(fabsf(q.y) < 1.0f - 1e-5f * (1.0f + fabsf(ray.p.y)));
                    if (edge && !box_hit(ray, mk3(-1, -1, 0), mk3(1, 1, 0), RTU_BIGFLOAT)) return false;
                }
                h.front = ray.p.z > 0;
                h.N = mk3(0, 0, h.front ? 1.0f : -1.0f);
                h.z = t;
                h.p = mk3(q.x, q.y, 0);
                if (tex) h.uvw = mk3((q.x + 1) / 2, (q.y + 1) / 2, 0);  // :131
                return true;
            }
        }
    }
    return false;
}
__device__ __forceinline__ bool plane_hit(const Ray& ray, Hit& h, bool tex = false) { return plane_hit_t<false>(ray, h, tex); }

// Point2::Cross (cyPoint.h:247-249)
__device__ __forceinline__ float cross2(float ax, float ay, float bx, float by) { return (-ay) * bx + ax * by; }

// cyTriMesh::Interpolate (cyTriMesh.h:191)
__device__ __forceinline__ f3 interp(const float* arr, const uint32_t* face, f3 bc) {
    return (ld3(arr + 3 * face[0]) * bc.x + ld3(arr + 3 * face[1]) * bc.y) + ld3(arr + 3 * face[2]) * bc.z;
}

// TriObj::IntersectTriangle (objFunctions.cpp:257-328) on a 64-byte triangle record that
// holds everything of the test that does not depend on the ray (computed once at
// upload with the reference's own float operations, rtu_capi.hip):
//   r0 = {A.xyz, N.x}  r1 = {N.y, N.z, A2.x, A2.y}  r2 = {C2-A2, B2-A2}  r3 = {1/TriABCArea (fp64), axis, -}
// N = normalised face normal (:263); axis = dominant axis of N (:276-296); *2 = vertex
// projected on the other two axes; TriABCArea as :298. "/2.0" in fp64 (:298-300) equals
// "* 0.5f": both are the correctly rounded half. BC1/BC2 divide by the per-triangle
// constant TriABCArea -> exact reciprocal form (see fdiv).
struct TriRec {
    float4 r0, r1, r2, r3;
};
__device__ __forceinline__ TriRec load_tri(const float4* tri, uint32_t slot) {
    TriRec t;
    const float4* p = tri + 4 * (size_t)slot;
    t.r0 = p[0]; t.r1 = p[1]; t.r2 = p[2]; t.r3 = p[3];
    return t;
}
// The accepted triangle with the smallest t wins (strict `t < hInfo.z`, earlier triangle on a
// tie). Its interpolated point and normal (:321-324) depend only on (face, barycentrics), so
// they are evaluated ONCE, for the final winner, after the walk (three dependent global
// loads per accepted candidate otherwise; shadow rays never need them).
struct TriWin {
    uint32_t slot;
    f3 bc;
};
// Returns 1 when the triangle is accepted (t < h.z), 0 when not. TIE (fast tree only): returns 2
// when the triangle passes every geometric test with t EXACTLY EQUAL to h.z — which triangle
// wins then depends on the reference's test order, so the caller falls back to the `ref` tree.
template <bool STATS, bool TIE>
__device__ __forceinline__ int tri_hit(const TriRec& T, uint32_t slot, const Ray& ray, Hit& h, TriWin& win, Counters& cnt) {
    RTU_CNT(tri);
    const f3 A = mk3(T.r0.x, T.r0.y, T.r0.z);
    const f3 N = mk3(T.r0.w, T.r1.x, T.r1.y);
    const float dn = dot3(ray.dir, N);
    if (dn != 0) {
        const float t = dot3(A - ray.p, N) / dn;
        if (gt_00001(t) && (TIE ? t <= h.z : t < h.z)) {  // :270, `t > 0.00001`
            const f3 q = ray.p + ray.dir * t;
            const uint32_t axis = __float_as_uint(T.r3.z);
            const float qx = axis == 0 ? q.y : q.x;
            const float qy = axis == 2 ? q.y : q.z;
            const float ax = T.r1.z, ay = T.r1.w;
            const float e1x = T.r2.x, e1y = T.r2.y, e2x = T.r2.z, e2y = T.r2.w;  // C2-A2, B2-A2
            const float TriAPCArea = cross2(e1x, e1y, qx - ax, qy - ay) * 0.5f;
            const float TriABPArea = cross2(qx - ax, qy - ay, e2x, e2y) * 0.5f;
            const double rcpABC = __hiloint2double(__float_as_int(T.r3.y), __float_as_int(T.r3.x));
            const float BC1 = fdiv(TriAPCArea, rcpABC);
            const float BC2 = fdiv(TriABPArea, rcpABC);
            const float BC3 = (float)(1.0 - (double)BC1 - (double)BC2);  // :304
            if (BC1 > 0 && BC2 > 0 && BC3 > 0 && BC1 < 1 && BC2 < 1 && BC3 < 1) {
                if (TIE && t == h.z) return 2;
                RTU_CNT(acc);
                win.slot = slot;
                win.bc = mk3(BC3, BC1, BC2);
                h.front = dn < 0;
                h.z = t;
                return 1;
            }
        }
    }
    return 0;
}

// TriObj::IntersectRay (objFunctions.cpp:333-406) — the hot loop on mesh scenes.
//
// Same visiting ORDER as the reference (near child first by tEntry+0.01, ties to the
// first child; leaves tested in element order), restructured for a 64-wide wavefront:
//  * the reference pushes both children (far first) and pops; popping the near child
//    right after pushing it equals continuing with it, so only the far child is
//    pushed, and it is pushed as {index,count} so a pop needs no node fetch: every
//    inner step is ONE 64-byte fetch (the sibling pair);
//  * "while-while": all lanes descend through inner nodes together and test
//    triangles together, instead of mixing both bodies in every iteration.
// CULL (the fast variant; the counting variant walks exactly the reference's set of
// nodes so its counters equal the CPU oracle's):
//  * a shadow ray only asks "is there an occluder" (GenLight::Shadow,
//    lightFunctions.cpp:27-37: any hit has z > 0) -> leave at the first accepted triangle;
//  * a child box is skipped when its entry distance is beyond the current h.z by a
//    margin that covers the rounding of the slab and triangle arithmetic (see
//    DESIGN.md "Culling margin"): every triangle in it would fail `t < hInfo.z`
//    (objFunctions.cpp:270), so skipping it cannot change any output bit.
template <int STACK, bool STATS, bool CULL, bool TIE, bool FC = false>
__device__ __forceinline__ bool mesh_walk(const RTU_CONST DevMesh& mesh, const float4* bvh, const float4* tris, const uint32_t* elements,
                                          const Ray& ray, bool shadow, Hit& h, uint32_t* stk, Counters& cnt, const uint32_t stride, bool& tie,
                                          const bool fc_lane = true) {
    // The reference's slab test has special cases for an exactly-zero direction component
    // (objFunctions.cpp:154-216). If ANY lane of the wavefront has one, the whole wavefront
    // takes the literal four-branch form; otherwise the branch-free reciprocal form.
    const bool zeroDir = ray.dir.x == 0 || ray.dir.y == 0 || ray.dir.z == 0;
    const bool slowSlabs = __any(zeroDir) != 0;
    const RayRcp rc = ray_rcp(ray);
    const bool emptyBoxes = mesh.any_empty_box != 0;  // never for a BVH built from triangles; uniform
    CullLim lim;
    lim.lx = lim.ly = lim.lz = 0.0f;
    if (CULL) lim = cull_limits(ray, mesh.scale > fmaxf(fabsf(ray.p.x), fmaxf(fabsf(ray.p.y), fabsf(ray.p.z))) ? mesh.scale
                                    : fmaxf(fabsf(ray.p.x), fmaxf(fabsf(ray.p.y), fabsf(ray.p.z))));
    bool hitResult = false;
    TriWin win;
    win.slot = 0;
    win.bc = mk3(0, 0, 0);
    int sp = 0;
    float4 r0 = bvh[2], r1 = bvh[3];  // root = node 1 (cyBVH.h:76)
    uint32_t index = __float_as_uint(r0.w), count = __float_as_uint(r1.w);
    bool alive = true;
    while (alive) {
        while (alive && count == 0) {  // inner nodes
            RTU_CNT(inner);
            RTU_TOUCH(t_innerref, 1);
            const float4* pair = bvh + 2 * index;  // children index, index+1: one 64-byte line
            float4 a0 = pair[0], a1 = pair[1], b0 = pair[2], b1 = pair[3];
            float e1, x1, e2, x2;
            f3 amin = mk3(a0.x, a0.y, a0.z), amax = mk3(a1.x, a1.y, a1.z);
            f3 bmin = mk3(b0.x, b0.y, b0.z), bmax = mk3(b1.x, b1.y, b1.z);
            bool skip1 = false, skip2 = false;
            if (slowSlabs) {  // literal form, no culling
                box_slabs(ray, amin, amax, e1, x1);
                box_slabs(ray, bmin, bmax, e2, x2);
            } else {
                box_slabs_rcp<CULL>(ray, rc, amin, amax, h.z, lim, e1, x1, skip1);
                box_slabs_rcp<CULL>(ray, rc, bmin, bmax, h.z, lim, e2, x2, skip2);
            }
            // BVHBoxIntersection (:408-522): -t_max for an empty box, tEntry + 0.01 (fp64) on a hit, else t_max
            float t1 = (e1 <= x1 && e1 < RTU_BIGFLOAT) ? (float)((double)e1 + 0.01) : RTU_BIGFLOAT;
            float t2 = (e2 <= x2 && e2 < RTU_BIGFLOAT) ? (float)((double)e2 + 0.01) : RTU_BIGFLOAT;
            if (emptyBoxes) {
                if (box_empty(amin, amax)) t1 = -RTU_BIGFLOAT;
                if (box_empty(bmin, bmax)) t2 = -RTU_BIGFLOAT;
            }
            bool v1 = t1 != RTU_BIGFLOAT, v2 = t2 != RTU_BIGFLOAT;
            if (CULL) {
                v1 = v1 && !skip1;
                v2 = v2 && !skip2;
            }
            // :361-389: (t1 <= t2) push c2 then c1; else push c1 then c2
            bool firstIsC1 = t1 <= t2;
            uint32_t p1 = __float_as_uint(a0.w) | (__float_as_uint(a1.w) << 28);
            uint32_t p2 = __float_as_uint(b0.w) | (__float_as_uint(b1.w) << 28);
            uint32_t nearP = firstIsC1 ? p1 : p2, farP = firstIsC1 ? p2 : p1;
            bool nearV = firstIsC1 ? v1 : v2, farV = firstIsC1 ? v2 : v1;
            uint32_t next;
            if (nearV) {
                if (farV) {
                    if (sp < STACK) stk[sp * stride] = farP;
                    sp++;
                }
                next = nearP;
            } else if (farV) {
                next = farP;
            } else if (sp > 0) {
                sp--;
                next = stk[sp * stride];
            } else {
                alive = false;
                next = 1u << 28;  // leave the inner loop
            }
            index = next & 0x0FFFFFFFu;
            count = next >> 28;
        }
        if (alive) {  // leaf: :394-396; the next record is fetched while the current one is tested
            RTU_CNT(leafv);
            if (STATS) cnt.leafe += count;
            RTU_TOUCH(t_tri, count);
            TriRec cur = load_tri(tris, index);
            for (uint32_t i = 0; i < count; i++) {
                TriRec nxt = cur;
                if (i + 1 < count) nxt = load_tri(tris, index + i + 1);
                const int code = tri_hit<STATS, TIE>(cur, index + i, ray, h, win, cnt);
                if (TIE && code == 2 && hitResult && !shadow) tie = true;  // equal t with the current best of THIS mesh
                hitResult |= code == 1;
                cur = nxt;
            }
            if (TIE && tie) alive = false;
            if (CULL && shadow && hitResult) {
                alive = false;
            } else if (sp > 0) {
                sp--;
                uint32_t next = stk[sp * stride];
                index = next & 0x0FFFFFFFu;
                count = next >> 28;
            } else {
                alive = false;
            }
        }
    }
    if (hitResult && !shadow && !(TIE && tie)) {
        // hInfo.N / hInfo.p of the winning triangle (:322, :324)
        RTU_TOUCH(t_win, 1);
        const uint32_t face = elements[win.slot];
        h.N = norm3(interp(mesh.vn, mesh.fn + 3 * face, win.bc));
        h.p = interp(mesh.v, mesh.f + 3 * face, win.bc);
        h.uvw = mesh.vt ? interp(mesh.vt, mesh.ft + 3 * face, win.bc) : mk3(0, 0, 0);  // GetTexCoord, objFunctions.cpp:320
    }
    return hitResult;
}

// ---- the fast tree's box test ---------------------------------------------------------------
// The SAH tree is ours, not the reference's: its box tests decide only WHICH triangles get the
// reference's exact triangle test, never a result bit. So they need not reproduce the reference's
// slab arithmetic — they only have to be CONSERVATIVE: never reject a box that holds a triangle
// the reference would accept. Each box is inflated by delta = 1e-4*scale on every axis (the
// culling margin derived above; the rounding of the test itself, ~2.4e-7*scale, is 0.24 % of it)
// and tested with one fma per slab plane:
//     t_near_k = near_k * r_k + (-p_k*r_k - L_k)      t_far_k = far_k * r_k + (-p_k*r_k + L_k)
// with r_k = 1/dir_k, L_k = delta*|r_k|, near/far chosen by the sign of r_k. A box is visited iff
// max_k t_near_k <= min_k t_far_k (the inflated box is hit), <= h.z (not beyond the best hit) and
// min_k t_far_k >= 0 (not behind the origin) — the same three conditions as box_slabs_rcp<CULL>,
// an inner step is ~55 instructions instead of ~130. A zero direction component is replaced by
// +-2^-100: the slab then constrains nothing when the origin is inside it (by more than delta)
// and rejects everything when it is outside, which is what delta-conservative means there.
struct FastRay {
    f3 r, cn, cf;
    bool px, py, pz;  // r_k >= 0: the near plane is bmin_k
};
__device__ __forceinline__ FastRay fast_ray(const Ray& ray, float meshScale) {
    const float tiny = 0x1p-100f;
    const float pm = fmaxf(fabsf(ray.p.x), fmaxf(fabsf(ray.p.y), fabsf(ray.p.z)));
    const float delta = 1e-4f * (meshScale > pm ? meshScale : pm);
    const float dx = fabsf(ray.dir.x) < tiny ? copysignf(tiny, ray.dir.x) : ray.dir.x;
    const float dy = fabsf(ray.dir.y) < tiny ? copysignf(tiny, ray.dir.y) : ray.dir.y;
    const float dz = fabsf(ray.dir.z) < tiny ? copysignf(tiny, ray.dir.z) : ray.dir.z;
    FastRay f;
    // (v_rcp_f32 is within 1 ulp: five orders of magnitude inside the inflation — three IEEE divisions per mesh entry saved)
    f.r = mk3(__builtin_amdgcn_rcpf(dx), __builtin_amdgcn_rcpf(dy), __builtin_amdgcn_rcpf(dz));
    const f3 pr = mk3(ray.p.x * f.r.x, ray.p.y * f.r.y, ray.p.z * f.r.z);
    const f3 L = mk3(delta * fabsf(f.r.x), delta * fabsf(f.r.y), delta * fabsf(f.r.z));
    f.cn = mk3(-pr.x - L.x, -pr.y - L.y, -pr.z - L.z);
    f.cf = mk3(-pr.x + L.x, -pr.y + L.y, -pr.z + L.z);
    f.px = f.r.x >= 0; f.py = f.r.y >= 0; f.pz = f.r.z >= 0;
    return f;
}
__device__ __forceinline__ bool fast_box(const FastRay& f, float4 lo, float4 hi, float hz, float& tn) {
    const float nx = f.px ? lo.x : hi.x, fx = f.px ? hi.x : lo.x;
    const float ny = f.py ? lo.y : hi.y, fy = f.py ? hi.y : lo.y;
    const float nz = f.pz ? lo.z : hi.z, fz = f.pz ? hi.z : lo.z;
    tn = fmaxf(fmaxf(fmaf(nx, f.r.x, f.cn.x), fmaf(ny, f.r.y, f.cn.y)), fmaf(nz, f.r.z, f.cn.z));
    const float tf = fminf(fminf(fmaf(fx, f.r.x, f.cf.x), fmaf(fy, f.r.y, f.cf.y)), fmaf(fz, f.r.z, f.cf.z));
    return tn <= tf && tn <= hz && tf >= 0.0f;
}

// fast_ray for the node-level bounds: the reciprocals need not be correctly rounded (v_rcp_f32 is within 1 ulp, five
// orders of magnitude inside the margin), and the scale of the margin is the scene's.
__device__ __forceinline__ FastRay fast_ray_world(const Ray& ray, float sceneScale) {
    const float tiny = 0x1p-100f;
    const float pm = fmaxf(fabsf(ray.p.x), fmaxf(fabsf(ray.p.y), fabsf(ray.p.z)));
    const float delta = 1e-4f * (sceneScale > pm ? sceneScale : pm);
    const float dx = fabsf(ray.dir.x) < tiny ? copysignf(tiny, ray.dir.x) : ray.dir.x;
    const float dy = fabsf(ray.dir.y) < tiny ? copysignf(tiny, ray.dir.y) : ray.dir.y;
    const float dz = fabsf(ray.dir.z) < tiny ? copysignf(tiny, ray.dir.z) : ray.dir.z;
    FastRay f;
    f.r = mk3(__builtin_amdgcn_rcpf(dx), __builtin_amdgcn_rcpf(dy), __builtin_amdgcn_rcpf(dz));
    const f3 pr = mk3(ray.p.x * f.r.x, ray.p.y * f.r.y, ray.p.z * f.r.z);
    const f3 L = mk3(delta * fabsf(f.r.x), delta * fabsf(f.r.y), delta * fabsf(f.r.z));
    f.cn = mk3(-pr.x - L.x, -pr.y - L.y, -pr.z - L.z);
    f.cf = mk3(-pr.x + L.x, -pr.y + L.y, -pr.z + L.z);
    f.px = f.r.x >= 0; f.py = f.r.y >= 0; f.pz = f.r.z >= 0;
    return f;
}

// THE ONE WAY THE TREES COULD DISAGREE, closed. The fast tree's boxes are conservative, so its walk tests every triangle the
// reference's walk tests — and possibly one more: a triangle the reference never reaches because ITS OWN box arithmetic
// (BVHBoxIntersection, objFunctions.cpp:408-522, on the triangle's leaf or an ancestor, or :337 on the mesh's box) rejects a
// ray that clips the box within rounding, although the triangle test would accept the hit. Every box the reference tests on
// the way to a triangle CONTAINS the triangle's own bounding box B, and its slab test passes iff entry_i <= exit_j for every
// pair of axes i != j (the same axis always passes: both bounds come from one monotone division). Those intervals only
// widen from B to its ancestors. So: if the ray passes B with entry_i + m_i <= exit_j - m_j for all i != j, where
// m_k = 1e-5 * max(scale, |origin|) / |dir_k| is 50 x the rounding of a slab bound (~2e-7 * (|bound| + |origin|) / |dir_k|),
// the reference has certainly reached the triangle. If not, the ray is ambiguous and is re-walked on the reference's tree
// like an exact tie. Checked for the triangle a walk is about to report (the closest hit, or a shadow ray's occluder).
__device__ __forceinline__ bool reaches_like_the_reference(const float* v, const uint32_t* fv, const Ray& ray, float meshScale) {
    const f3 A = ld3(v + 3 * fv[0]), B = ld3(v + 3 * fv[1]), C = ld3(v + 3 * fv[2]);
    const f3 lo = mk3(fminf(A.x, fminf(B.x, C.x)), fminf(A.y, fminf(B.y, C.y)), fminf(A.z, fminf(B.z, C.z)));
    const f3 hi = mk3(fmaxf(A.x, fmaxf(B.x, C.x)), fmaxf(A.y, fmaxf(B.y, C.y)), fmaxf(A.z, fmaxf(B.z, C.z)));
    const float pm = fmaxf(fabsf(ray.p.x), fmaxf(fabsf(ray.p.y), fabsf(ray.p.z)));
    const float delta = 1e-5f * (meshScale > pm ? meshScale : pm);
    const float tiny = 0x1p-100f;
    const float dx = fabsf(ray.dir.x) < tiny ? copysignf(tiny, ray.dir.x) : ray.dir.x;
    const float dy = fabsf(ray.dir.y) < tiny ? copysignf(tiny, ray.dir.y) : ray.dir.y;
    const float dz = fabsf(ray.dir.z) < tiny ? copysignf(tiny, ray.dir.z) : ray.dir.z;
    const float rx = __builtin_amdgcn_rcpf(dx), ry = __builtin_amdgcn_rcpf(dy), rz = __builtin_amdgcn_rcpf(dz);
    const float mx = delta * fabsf(rx), my = delta * fabsf(ry), mz = delta * fabsf(rz);
    const float x0 = (lo.x - ray.p.x) * rx, x1 = (hi.x - ray.p.x) * rx;
    const float y0 = (lo.y - ray.p.y) * ry, y1 = (hi.y - ray.p.y) * ry;
    const float z0 = (lo.z - ray.p.z) * rz, z1 = (hi.z - ray.p.z) * rz;
    const float ex = fminf(x0, x1) + mx, fx = fmaxf(x0, x1) - mx;  // entry pushed later, exit pulled earlier
    const float ey = fminf(y0, y1) + my, fy = fmaxf(y0, y1) - my;
    const float ez = fminf(z0, z1) + mz, fz = fmaxf(z0, z1) - mz;
    return ex <= fminf(fy, fz) && ey <= fminf(fx, fz) && ez <= fminf(fx, fy);  // NaN anywhere: false -> the reference's tree decides
}

// The fast variant's walk of the SAH tree: near child first (by the inflated entry distance —
// the order only affects speed: an exact tie between two accepted triangles, the one case where
// the order would show, is detected by tri_hit<TIE> and resolved on the reference's tree).
__device__ __forceinline__ TriRec load_tri(GBase tri, uint32_t slot) {
    TriRec t;
    const uint32_t o = slot << 6;
    t.r0 = gload4(tri, o); t.r1 = gload4(tri, o + 16u); t.r2 = gload4(tri, o + 32u); t.r3 = gload4(tri, o + 48u);
    return t;
}
template <int STACK, bool FC = false, class MeshT>
__device__ __forceinline__ bool mesh_walk_fast(const MeshT& mesh, const Ray& ray, bool shadow, Hit& h, uint32_t* stk, Counters& cnt,
                                               const uint32_t stride, bool& tie, const uint32_t stackLimit) {
    const bool fc_lane = true;
    const int slim = (int)(stackLimit < (uint32_t)STACK ? stackLimit : (uint32_t)STACK);
    // The tree collapsed to four children per node (DevMesh::bvh4): half the dependent fetches. Bases in scalar registers,
    // 32-bit byte offsets per lane, global loads (rtu_device.h GBase).
    const GBase bvh4 = global_base(mesh.bvh4);
    const GBase tris = global_base(mesh.fast.tri);
    const FastRay fr = fast_ray(ray, mesh.scale);
    // near / far plane arrays of a node, by the sign of the ray (see build_wide4): byte offsets inside the 128-byte node
    const uint32_t onx = fr.px ? 0u : 48u, ofx = 48u - onx, ony = fr.py ? 16u : 64u, ofy = 80u - ony, onz = fr.pz ? 32u : 80u, ofz = 112u - onz;
    bool hitResult = false;
    TriWin win;
    win.slot = 0;
    win.bc = mk3(0, 0, 0);
    int sp = 0;
    uint32_t index = 0, count = 0;  // root = node4 0
    bool alive = true;
    while (alive) {
        while (alive && count == 0) {  // inner nodes
            RTU_TOUCH(t_inner4, 1);
            const uint32_t nb = index << 7;
            const float4 nx = gload4(bvh4, nb + onx), ny = gload4(bvh4, nb + ony), nz = gload4(bvh4, nb + onz);
            const float4 fx = gload4(bvh4, nb + ofx), fy = gload4(bvh4, nb + ofy), fz = gload4(bvh4, nb + ofz);
            float4 rf = gload4(bvh4, nb + 96u);
            // (the child references are needed only where a child is hit: left alone the compiler sinks their load below the box
            // tests — a SECOND dependent round trip to memory in every step of the walk. Issued with the planes instead.)
            asm volatile("" : "+v"(rf.x), "+v"(rf.y), "+v"(rf.z), "+v"(rf.w));
            const float inf = __builtin_inff();
#define RTU_CHILD(c)                                                                                                       \
            const float tn##c = fmaxf(fmaxf(fmaf(nx.c, fr.r.x, fr.cn.x), fmaf(ny.c, fr.r.y, fr.cn.y)), fmaf(nz.c, fr.r.z, fr.cn.z)); \
            const float tf##c = fminf(fminf(fmaf(fx.c, fr.r.x, fr.cf.x), fmaf(fy.c, fr.r.y, fr.cf.y)), fmaf(fz.c, fr.r.z, fr.cf.z)); \
            const float k##c = (tn##c <= tf##c && tn##c <= h.z && tf##c >= 0.0f) ? tn##c : inf;
            RTU_CHILD(x) RTU_CHILD(y) RTU_CHILD(z) RTU_CHILD(w)
#undef RTU_CHILD
            // the nearest hit child is next; the other hit children go on the stack
            const uint32_t rx = __float_as_uint(rf.x), ry = __float_as_uint(rf.y), rz = __float_as_uint(rf.z), rw = __float_as_uint(rf.w);
            const bool a01 = kx <= ky, a23 = kz <= kw;
            const float k01 = a01 ? kx : ky, k23 = a23 ? kz : kw;
            const uint32_t r01 = a01 ? rx : ry, r23 = a23 ? rz : rw;
            const bool a = k01 <= k23;
            const float kbest = a ? k01 : k23;
            const uint32_t rbest = a ? r01 : r23;
            uint32_t next;
            if (kbest < inf) {
                const int n = (int)(kx < inf) + (int)(ky < inf) + (int)(kz < inf) + (int)(kw < inf);
                if (sp + n - 1 > slim) {  // would not fit: finish on the reference's tree
                    tie = true;
                    alive = false;
                    next = 1u << 28;
                } else {
                    if (kx < inf && rx != rbest) { stk[sp * stride] = rx; sp++; }
                    if (ky < inf && ry != rbest) { stk[sp * stride] = ry; sp++; }
                    if (kz < inf && rz != rbest) { stk[sp * stride] = rz; sp++; }
                    if (kw < inf && rw != rbest) { stk[sp * stride] = rw; sp++; }
                    next = rbest;
                }
            } else if (sp > 0) {
                sp--;
                next = stk[sp * stride];
            } else {
                alive = false;
                next = 1u << 28;  // leave the inner loop
            }
            index = next & 0x0FFFFFFFu;
            count = next >> 28;
        }
        if (alive) {
            // leaf: its records are fetched one test ahead, in two buffers taking turns (a single "current / next" pair costs 26
            // register moves per triangle: measured in the ISA of the round-2 walk)
            RTU_TOUCH(t_tri, count);
            TriRec A = load_tri(tris, index), B = A;
            for (uint32_t i = 0;; i += 2u) {
                const bool hasB = i + 1u < count;
                if (hasB) B = load_tri(tris, index + i + 1u);
                int code = tri_hit<false, true>(A, index + i, ray, h, win, cnt);
                if (code == 2 && hitResult && !shadow) tie = true;  // equal t with the current best of THIS mesh
                hitResult |= code == 1;
                if (!hasB) break;
                const bool hasA = i + 2u < count;
                if (hasA) A = load_tri(tris, index + i + 2u);
                code = tri_hit<false, true>(B, index + i + 1u, ray, h, win, cnt);
                if (code == 2 && hitResult && !shadow) tie = true;
                hitResult |= code == 1;
                if (!hasA) break;
            }
            if (tie) alive = false;
            if (shadow && hitResult) {
                alive = false;
            } else if (sp > 0) {
                sp--;
                const uint32_t next = stk[sp * stride];
                index = next & 0x0FFFFFFFu;
                count = next >> 28;
            } else {
                alive = false;
            }
        }
    }
    if (hitResult && !tie && !reaches_like_the_reference(mesh.v, mesh.f + 3 * mesh.fast.elements[win.slot], ray, mesh.scale)) tie = true;
    if (hitResult && !shadow && !tie) {
        RTU_TOUCH(t_win, 1);
        const uint32_t face = mesh.fast.elements[win.slot];
        h.N = norm3(interp(mesh.vn, mesh.fn + 3 * face, win.bc));
        h.p = interp(mesh.v, mesh.f + 3 * face, win.bc);
        h.uvw = mesh.vt ? interp(mesh.vt, mesh.ft + 3 * face, win.bc) : mk3(0, 0, 0);  // GetTexCoord, objFunctions.cpp:320
    }
    return hitResult;
}

// The counting variant walks the reference's tree; the fast variant walks the SAH tree and
// falls back to the reference's on an exact tie (see DevMesh).
template <int STACK, bool STATS, bool CULL, bool FC = false>
__device__ __forceinline__ bool mesh_hit(const RTU_CONST DevMesh& mesh, const Ray& ray, bool shadow, Hit& h, uint32_t* stk, Counters& cnt,
                                         const uint32_t stackLimit, const uint32_t stride = 64) {
    if (!box_hit(ray, ld3(mesh.bmin), ld3(mesh.bmax), RTU_BIGFLOAT)) return false;
    RTU_CNT(mesh);
    bool tie = false;
    if (!CULL) return mesh_walk<STACK, STATS, false, false>(mesh, mesh.ref.bvh, mesh.ref.tri, mesh.ref.elements, ray, shadow, h, stk, cnt, stride, tie);
    const Hit h0 = h;
    bool r = mesh_walk_fast<STACK, FC>(mesh, ray, shadow, h, stk, cnt, stride, tie, stackLimit);
    if (tie) {  // rare: two accepted triangles with bitwise-equal t — the reference's test order decides
        h = h0;
        bool t2 = false;
        r = mesh_walk<STACK, STATS, CULL, false, FC>(mesh, mesh.ref.bvh, mesh.ref.tri, mesh.ref.elements, ray, shadow, h, stk, cnt, stride, t2);
    }
    return r;
}

// ---------------------------------------------------------------------------
// COOPERATIVE BVH WALK: eight lanes per ray, on the fast tree collapsed to EIGHT children per
// node (DevMesh::bvh8).
//
// Measured (profiles/r01_*): the steps of one ray are sequential and each costs a dependent
// fetch plus ~50-100 instructions at one instruction per ~4.5 cycles; a ray with 100+ steps
// holds its wavefront — and, being the slowest, the whole phase — while the chip idles. The
// steps of ONE ray cannot overlap, but eight lanes can do more per step:
//   inner step: lane c tests child c of the node (fast_box: six fma), the hit children are
//               ranked by entry distance across the group (seven ds_swizzle exchanges) and pushed
//               on the ray's LDS stack in one go, nearest on top — a third of the depth of the
//               binary tree, the same instruction count per step;
//   leaf round: up to 8 triangles -> one triangle per lane, then a min-t reduction with
//               the lower element index winning a tie (= the reference's strict `t < z`
//               applied in element order, objFunctions.cpp:270,394-396).
// All eight lanes hold the same ray and keep identical copies of the walk state (h.z,
// winner, stack pointer), so every lane takes the same branches; values move between lanes
// with ds_swizzle / ds_bpermute. The visiting order only affects speed (see mesh_walk_fast);
// the triangle arithmetic is the reference's.
template <int M>
__device__ __forceinline__ int grp_xor(int v) {  // value of lane ^ M
    return __builtin_amdgcn_ds_swizzle(v, 0x1F | (M << 10));
}

// The caller has established that the ray passes the mesh's bounding box.
template <int STACK, bool CULL, bool FC = false, class MeshT>
__device__ __forceinline__ bool mesh_hit_coop(const MeshT& mesh, const Ray& ray, bool shadow, Hit& h, uint32_t* stk, Counters& cnt,
                                              const uint32_t stride, const float4* lds_nodes, const uint32_t stackLimit) {
    const bool fc_lane = (threadIdx.x & 7u) == 0;  // eight lanes share the ray: one of them counts
    const uint32_t slim = stackLimit < (uint32_t)RTU_STACK8 ? stackLimit : (uint32_t)RTU_STACK8;
    const Hit h0 = h;
    bool tie = false;
    const GBase bvh8 = global_base(mesh.bvh8);
    const GBase tris = global_base(mesh.fast.tri);
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t sub = lane & 7u;
    const uint32_t ldsN = lds_nodes ? mesh.lds_nodes : 0u, ldsOff = mesh.lds_off;  // lds_nodes == nullptr: nothing staged
    const FastRay fr = fast_ray(ray, mesh.scale);
    bool hitResult = false;
    TriWin win;
    win.slot = 0;
    win.bc = mk3(0, 0, 0);
    uint32_t sp = 0;
    uint32_t index = 0, count = 0;  // root = node8 0
    bool alive = true;
    while (alive) {
        while (alive && count == 0) {  // inner node: lane `sub` tests child `sub`
            RTU_TOUCH(t_inner8, 1);
            float4 c0, c1;
            if (index < ldsN) {  // group-uniform: the top of the tree is staged in LDS
                const float4* nd = lds_nodes + ldsOff + ((size_t)index * 8u + sub) * 2u;
                c0 = nd[0]; c1 = nd[1];
            } else {
                const uint32_t o = (index * 8u + sub) << 5;
                c0 = gload4(bvh8, o); c1 = gload4(bvh8, o + 16u);
            }
            float tn;
            const uint32_t ref = __float_as_uint(c0.w);
            const bool valid = fast_box(fr, c0, c1, h.z, tn) && ref != RTU_REF8_EMPTY;
            const uint32_t m8 = (uint32_t)(__ballot(valid) >> (lane & 56u)) & 0xFFu;
            uint32_t next;
            if (m8 == 0) {
                if (sp > 0) {
                    sp--;
                    next = stk[sp * stride];
                } else {
                    alive = false;
                    next = 1u << 28;
                }
            } else {
                const uint32_t n = (uint32_t)__popc(m8);
                if (sp + n > slim) {  // would not fit (group-uniform): finish on the reference's tree
                    tie = true;
                    alive = false;
                    next = 1u << 28;
                } else {
                    // sortable key: entry distance, the child index in the low bits makes it unique
                    const uint32_t tb = __float_as_uint(tn);
                    const int key = valid ? (int)(((tb ^ ((tb >> 31) ? 0xFFFFFFFFu : 0x80000000u)) & ~7u) | sub) ^ (int)0x80000000 : 0x7FFFFFFF;
                    uint32_t rank = 0;
                    rank += grp_xor<1>(key) < key; rank += grp_xor<2>(key) < key; rank += grp_xor<3>(key) < key;
                    rank += grp_xor<4>(key) < key; rank += grp_xor<5>(key) < key; rank += grp_xor<6>(key) < key;
                    rank += grp_xor<7>(key) < key;
                    if (valid) stk[(sp + n - 1u - rank) * stride] = ref;  // nearest on top
                    sp += n - 1u;
                    next = stk[sp * stride];
                }
            }
            index = next & 0x0FFFFFFFu;
            count = next >> 28;
        }
        if (alive) {  // leaf: lane `sub` tests element index+sub
            float myT = RTU_BIGFLOAT;
            bool myFront = true;
            TriWin myWin;
            myWin.slot = index + sub;
            myWin.bc = mk3(0, 0, 0);
            if (sub < count) {
                if (FC) cnt.t_tri++;  // every lane tests its own triangle
                const TriRec T = load_tri(tris, index + sub);
                Hit hl = h;  // test against the best BEFORE this leaf; the reduction below applies the order
                const int code = tri_hit<false, true>(T, index + sub, ray, hl, myWin, cnt);
                if (code == 1) { myT = hl.z; myFront = hl.front; }
                if (code == 2 && hitResult && !shadow) tie = true;
            }
            // min t over the group, lower sub wins a tie: key = (t, sub)
            int kt = __float_as_int(myT), ks = (int)sub;  // accepted t is positive: integer order == float order
            {
                const int big = __float_as_int(RTU_BIGFLOAT);
                int ot = grp_xor<1>(kt), os = grp_xor<1>(ks);
                if (ot == kt && kt != big && !shadow) tie = true;  // two triangles of this leaf accepted with equal t
                if (ot < kt || (ot == kt && os < ks)) { kt = ot; ks = os; }
                ot = grp_xor<2>(kt); os = grp_xor<2>(ks);
                if (ot == kt && kt != big && os != ks && !shadow) tie = true;
                if (ot < kt || (ot == kt && os < ks)) { kt = ot; ks = os; }
                ot = grp_xor<4>(kt); os = grp_xor<4>(ks);
                if (ot == kt && kt != big && os != ks && !shadow) tie = true;
                if (ot < kt || (ot == kt && os < ks)) { kt = ot; ks = os; }
            }
            // any lane of the group saw a tie -> the whole group falls back (keeps the lanes in lockstep)
            {
                int tf = tie ? 1 : 0;
                tf |= grp_xor<1>(tf); tf |= grp_xor<2>(tf); tf |= grp_xor<4>(tf);
                tie = tf != 0;
            }
            if (tie) alive = false;
            const float bestT = __int_as_float(kt);
            if (bestT < h.z) {  // somebody accepted a triangle (accepted t < h.z, misses carry BIGFLOAT >= h.z)
                const int src = (int)(((lane & ~7u) + (uint32_t)ks) << 2);
                h.z = bestT;
                h.front = __builtin_amdgcn_ds_bpermute(src, myFront ? 1 : 0) != 0;
                win.slot = index + (uint32_t)ks;
                win.bc.x = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(myWin.bc.x)));
                win.bc.y = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(myWin.bc.y)));
                win.bc.z = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(myWin.bc.z)));
                hitResult = true;
            }
            if (CULL && shadow && hitResult) {
                alive = false;
            } else if (sp > 0) {
                sp--;
                uint32_t next = stk[sp * stride];
                index = next & 0x0FFFFFFFu;
                count = next >> 28;
            } else {
                alive = false;
            }
        }
    }
    if (hitResult && !tie && !reaches_like_the_reference(mesh.v, mesh.f + 3 * mesh.fast.elements[win.slot], ray, mesh.scale)) tie = true;  // group-uniform
    if (tie) {  // redo this ray on the reference's tree, one lane per ray (all eight lanes identically)
        h = h0;
        bool t2 = false;
        return mesh_walk<STACK, false, CULL, false, FC>(mesh, mesh.ref.bvh, mesh.ref.tri, mesh.ref.elements, ray, shadow, h, stk, cnt, stride, t2, fc_lane);
    }
    if (hitResult && !shadow) {
        RTU_TOUCH(t_win, 1);
        const uint32_t face = mesh.fast.elements[win.slot];
        h.N = norm3(interp(mesh.vn, mesh.fn + 3 * face, win.bc));
        h.p = interp(mesh.v, mesh.f + 3 * face, win.bc);
        h.uvw = mesh.vt ? interp(mesh.vt, mesh.ft + 3 * face, win.bc) : mk3(0, 0, 0);  // GetTexCoord, objFunctions.cpp:320
    }
    return hitResult;
}

// ---------------------------------------------------------------------------
// A HARD SHADOW RAY AGAINST A MESH WITHOUT A BVH WALK (DevLightMask, rtu_capi.hip build_light_lists). The ray runs from its
// origin towards the light, so the cell of the light's grid that the ORIGIN projects into lists every triangle of the mesh the
// ray can touch; entries [e0, e1) of the list get the reference's own triangle test (tri_hit, `t < t_max` as in a fresh
// HitInfo of ShadowTrace, RenderFunctions.cpp:213-240). GenLight::Shadow (lightFunctions.cpp:27-37) asks only WHETHER the
// mesh is hit, and TriObj::IntersectRay (objFunctions.cpp:333-406) says yes exactly when (a) the ray passes the mesh's own
// box (:335-337), and (b) its walk reaches a triangle that accepts the ray. A triangle that is not in the list cannot accept
// it (the list is conservative by the cull margin, like the boxes of the fast tree); so
//   no entry accepts                      -> no occluder in this mesh, whatever the boxes say           (returns 0)
//   an entry accepts, (a) fails           -> no occluder: the reference never looks at the triangles    (returns 0)
//   an entry accepts, (a) holds, the ray passes the triangle's own bounding box with the reserve of
//   reaches_like_the_reference            -> the reference's walk gets to that triangle: an occluder     (returns 1)
//   otherwise                             -> ambiguous within rounding: the caller walks the reference's tree (returns 2)
// (when the reference reaches the triangle it either accepts it or has already accepted another one: a hit both ways).
// UNI: the light (hence the list) is the same for every lane of the wavefront — stage 1 of a tracing phase, where a wavefront
// fires the shadow rays of ONE light: the list's base address stays in scalar registers. Stage 2 and the tail kernel mix the
// lights of their rays: a pointer per lane.
template <bool UNI> struct EntList;
template <> struct EntList<true> {
    GBase b;
    __device__ __forceinline__ explicit EntList(const uint32_t* p) : b(global_base(p)) {}
    __device__ __forceinline__ uint32_t at(uint32_t i) const { return gload1(b, i << 2); }
    __device__ __forceinline__ uint2 at2(uint32_t i) const { return gload2u(b, i << 3); }
};
template <> struct EntList<false> {
    const uint32_t* p;
    __device__ __forceinline__ explicit EntList(const uint32_t* q) : p(q) {}
    __device__ __forceinline__ uint32_t at(uint32_t i) const { return p[i]; }
    __device__ __forceinline__ uint2 at2(uint32_t i) const { return reinterpret_cast<const uint2*>(p)[i]; }
};
// `depth`: the origin's depth along the light's axis (the device's own evaluation; the entries' zmin carry the margin for its
// rounding). A cell's entries are sorted by zmin, the depth in front of which an origin cannot see the triangle: the walk ends
// at the first entry beyond the origin — for a ray that leaves the lit side of a surface, right after the triangles around its
// own origin. Entry i + 1's record and entry i + 2 are in flight while entry i is tested (two record buffers taking turns).
// STEP: 1, or 8 in the cooperative kernels, where the eight lanes of a ray share the list (lane `sub` takes entries sub, sub + 8,
// ...) and agree on the answer afterwards (any certain occluder: 1; else any ambiguous one: 2).
template <bool FC, bool UNI, uint32_t STEP, class MeshT>
__device__ __forceinline__ int mesh_shadow_cells(const uint32_t* cell_tri, uint32_t e0, const uint32_t e1, const float depth, const MeshT& mesh,
                                                 const Ray& lr, float tmax, Counters& cnt, const bool fc_lane) {
    const EntList<UNI> ents(cell_tri);
    const GBase tris = global_base(mesh.fast.tri);
    Hit h;
    fresh_hit(h, tmax);
    TriWin win;
    win.slot = 0;
    win.bc = mk3(0, 0, 0);
    uint32_t found = ~0u;
    if (STEP > 1u) e0 += threadIdx.x & (STEP - 1u);
    // (NaN depth: no entry is "beyond", all are tested)
    bool hasA = e0 < e1, hasB = false;
    uint2 ea = make_uint2(0u, 0u), eb = ea;
    TriRec A = {}, B = {};
    if (hasA) {
        ea = ents.at2(e0);
        eb = ea;
        hasA = !(__uint_as_float(ea.y) > depth);
    }
    if (hasA) {
        A = load_tri(tris, ea.x);
        if (e0 + STEP < e1) eb = ents.at2(e0 + STEP);
        hasB = e0 + STEP < e1 && !(__uint_as_float(eb.y) > depth);
    }
    while (hasA) {
        uint2 ec = eb;
        if (hasB) {
            B = load_tri(tris, eb.x);
            if (e0 + 2u * STEP < e1) ec = ents.at2(e0 + 2u * STEP);
        }
        if (FC) { cnt.t_tri++; cnt.t_bytes += 8u; }  // (every lane tests its own entries)
        if (tri_hit<false, false>(A, ea.x, lr, h, win, cnt) == 1) { found = ea.x; break; }
        if (!hasB) break;
        hasA = e0 + 2u * STEP < e1 && !(__uint_as_float(ec.y) > depth);
        uint2 ed = ec;
        if (hasA) {
            A = load_tri(tris, ec.x);
            if (e0 + 3u * STEP < e1) ed = ents.at2(e0 + 3u * STEP);
        }
        if (FC) { cnt.t_tri++; cnt.t_bytes += 8u; }
        if (tri_hit<false, false>(B, eb.x, lr, h, win, cnt) == 1) { found = eb.x; break; }
        ea = ec;
        eb = ed;
        hasB = hasA && e0 + 3u * STEP < e1 && !(__uint_as_float(ed.y) > depth);
        e0 += 2u * STEP;
    }
    int code = 0;
    if (found != ~0u && box_hit(lr, ld3(mesh.bmin), ld3(mesh.bmax), RTU_BIGFLOAT))
        code = reaches_like_the_reference(mesh.v, mesh.f + 3 * mesh.fast.elements[found], lr, mesh.scale) ? 1 : 2;
    if (STEP > 1u) {  // the group's answer (all lanes of a group are here: they share the ray and the list)
        const uint32_t sh = threadIdx.x & 63u & ~(STEP - 1u);
        const uint32_t c1 = (uint32_t)(__ballot(code == 1) >> sh) & ((1u << STEP) - 1u), c2 = (uint32_t)(__ballot(code == 2) >> sh) & ((1u << STEP) - 1u);
        code = c1 ? 1 : c2 ? 2 : 0;
    }
    return code;
}

// ---------------------------------------------------------------------------
// Trace / ShadowTrace (RenderFunctions.cpp:181-240), recursion over the node tree
// flattened to a pre-order loop. Only h.z (and h.front in the sphere quirk) feeds
// later intersection tests, so FromNodeCoords is applied once, after the loop, to the final
// hit and its ancestors — equivalent to the reference applying it as the recursion unwinds.
//
// ONE instantiation serves both kinds of ray: `shadow` is a per-lane flag, so
// lanes casting shadow rays and lanes casting reflection / refraction rays walk
// the scene together (better SIMD occupancy, a quarter of the code size of four
// specialised copies — the kernel has to stay inside the instruction cache).
//
// DEFER (stage 1 of a two-stage phase, see render_kernel.hip): the moment the ray passes
// the bounding box of a mesh node the walk is abandoned and `deferred` is set; the caller
// queues the ray for the narrow-wavefront stage-2 kernel, which walks the whole scene
// again with DEFER=false. Rays that never touch a mesh complete in stage 1.
//
// NODE-LEVEL BOUNDS (CULL only, i.e. the fast variant; DevNode::wmin / wmax). Every hit the reference can report lies, within
// rounding, inside the object's own bounding box: Box::IntersectRay on that box is the first thing Sphere / Plane /
// TriObj::IntersectRay do (objFunctions.cpp:17, :109, :335), its slab arithmetic is good to a few ulp of the ray's
// coordinates, and where its exactly-zero-direction branches ignore an axis (:154-216) the accepted hit point itself is on
// the sphere, inside the square or inside a triangle. So a ray whose line, between its origin and the best hit so far, stays
// clear of the box by delta = 1e-4 * max(scene scale, |origin|) on some axis — the geometric form of "Culling" above, on
// the world-space box of the node — cannot produce a result there, and the node is skipped before its transformation
// and exact test. (The margin is 100 x the slab rounding; the bound of a sphere is widened further at upload by what the
// cancellation in its discriminant can move a grazing root, rtu_capi.hip world_bounds.) `skip` marks nodes the caller
// has already excluded in the same sense: for primary rays, pixels outside the node's screen rectangle (k_node_rects).
// INL (with DEFER): the occluder lists are walked HERE, in stage 1 (the inline shading of childless Shade() calls, render_impl.h
// shadows_inline): `deferred` then means "this ray cannot be settled without a BVH walk".
template <int STACK, bool STATS, bool CULL, bool DEFER, bool COOP = false, bool TEX = false, bool FC = false, bool ULS = DEFER, bool INL = false>
__device__ __forceinline__ bool trace(const DevScene& s, const Ray& wr, bool shadow, Hit& h, uint32_t* stk, Counters& cnt, bool& deferred,
                                      const uint32_t stride = 64, const float4* lds_nodes = nullptr, const unsigned long long skip = 0,
                                      const bool rays_bounded = false, const int lslot = -1) {
    const bool fc_lane = !COOP || (threadIdx.x & 7u) == 0;
    RTU_TOUCH(t_rays, 1);
    const bool bounds = CULL && s.node_bounds != 0 && !rays_bounded;  // wave-uniform
    FastRay wf = {};
    if (bounds) wf = fast_ray_world(wr, s.wscale);
    const RTU_CONST DevNode* nodes = as_const(s.nodes);
    const RTU_CONST DevMesh* meshes = as_const(s.meshes);
    bool any = false;
    int best = -1;
    f3 lp = mk3(0, 0, 0), lN = mk3(0, 0, 0);
    deferred = false;
    Ray r0 = to_node(nodes[0], wr);  // ray inside the root node
    Ray rp = r0;                     // ray inside node `rp_node` (cached parent space)
    int rp_node = 0;
    for (uint32_t k = 0; k < s.n_nodes; k++) {
        const RTU_CONST DevNode& n = nodes[k];
        if (n.obj_type == RTU_OBJ_NONE) continue;
        if (shadow && any) continue;  // ShadowTrace returns at the first occluder (:223-225)
        if (DEFER && deferred) continue;
        if (CULL && k < 64u && ((skip >> k) & 1ull)) continue;
        // a hard shadow ray towards light `lslot` (lslot < RTU_LMASK_LIGHTS) and a mesh node with an occluder list for that light: the
        // cell of the ray's ORIGIN names the triangles the ray can touch — none: the mesh is skipped; else they are tested below
        uint32_t le0 = 0, le1 = 0;  // le1 > le0: the entries
        const uint32_t* ltri = nullptr;
        float ldepth = 0.0f;
        int lc = -1;
        if (CULL && lslot >= 0 && n.obj_type == RTU_OBJ_TRIMESH && s.lmask && s.node_bounds) {
            int c = -1;
            for (uint32_t i = 0; i < s.n_cover; i++) c = s.cover_node[i] == (int)k ? (int)i : c;
            if (c >= 0) {
                // (ULS: the light is wave-uniform, the list's frame arrives by scalar loads; else by one load per lane)
                const DevLightMask& mg = s.lmask[(uint32_t)lslot * s.n_cover + (uint32_t)c];
                const RTU_CONST DevLightMask& mc = as_const(s.lmask)[(uint32_t)(ULS ? __builtin_amdgcn_readfirstlane(lslot) : 0) * s.n_cover + (uint32_t)c];
                DevLightMask m;
                if (ULS) {
                    for (int q = 0; q < 3; q++) { m.X[q] = mc.X[q]; m.Y[q] = mc.Y[q]; m.Z[q] = mc.Z[q]; m.L[q] = mc.L[q]; }
                    m.u0 = mc.u0; m.v0 = mc.v0; m.su = mc.su; m.sv = mc.sv; m.usable = mc.usable; m.point = mc.point; m.G = mc.G;
                    m.cell_off = mc.cell_off; m.cell_tri = mc.cell_tri;
                } else {
                    m = mg;
                }
                if (m.usable) {
                    const f3 v = wr.p - ld3(m.L);
                    float u = dot3(v, ld3(m.X)), w = dot3(v, ld3(m.Y));
                    bool covered = true, listed = false;
                    const float depth = dot3(v, ld3(m.Z));
                    ldepth = depth;
                    if (m.point) {
                        const float rd = __builtin_amdgcn_rcpf(depth);
                        u *= rd; w *= rd;
                        if (!(depth > 0.0f)) covered = false;  // the origin is on the far side of the light: the mesh is not between them
                    }
                    const float tu = (u - m.u0) * m.su, tw = (w - m.v0) * m.sv;
                    const float G = (float)m.G;
                    if (covered) {
                        if (!(tu >= 0.0f && tw >= 0.0f && tu < G && tw < G)) covered = tu != tu || tw != tw;  // outside the mesh's extent (NaN: no answer)
                        else {
                            const uint32_t cell = (uint32_t)tw * m.G + (uint32_t)tu;
                            const EntList<ULS> offs(m.cell_off);
                            le0 = offs.at(cell);
                            le1 = offs.at(cell + 1u);
                            ltri = m.cell_tri;
                            covered = le1 > le0;
                            listed = true;
                        }
                    }
                    RTU_TOUCH_WAVE(t_bytes, 64u);  // the list's frame (scalar loads)
                    RTU_TOUCH(t_bytes, 8u);        // ... and the lane's cell of it
                    if (!covered) continue;
                    if (listed) lc = c;
                }
            }
        }
        if (bounds) {
            float tn;
            RTU_TOUCH_WAVE(t_bounds, 1);
            if (!fast_box(wf, make_float4(n.wmin[0], n.wmin[1], n.wmin[2], 0.0f), make_float4(n.wmax[0], n.wmax[1], n.wmax[2], 0.0f), h.z, tn)) continue;
        }
        int parent = n.parent;
        Ray pr;
        if (parent < 0) {
            pr = wr;
        } else {
            if (parent != rp_node) {
                const RTU_CONST DevNode& pn = nodes[parent];
                Ray t = r0;
                for (int d = 1; d <= pn.depth; d++) t = to_node(nodes[pn.chain[d]], t);
                rp = t;
                rp_node = parent;
            }
            pr = rp;
        }
        Ray lr = to_node(n, pr);
        RTU_CNT(node);
        RTU_TOUCH_WAVE(t_node, 1);
        if (n.obj_type == RTU_OBJ_TRIMESH) RTU_TOUCH_WAVE(t_meshbox, 1);
        bool hit;
        if (n.obj_type == RTU_OBJ_SPHERE) hit = sphere_hit(lr, h, TEX);
        else if (n.obj_type == RTU_OBJ_PLANE) hit = plane_hit(lr, h, TEX);
        else if (lc >= 0 && !(s.dbg & 64u)) {
            // (rtu_debug_flags bit 6: the occluder lists only say "empty cell or not", every listed ray walks the tree)
            const RTU_CONST DevMesh& mesh = meshes[n.mesh_id];
            // stage 1 only looks the cell up: a ray with entries to test joins the defer list, where stage 2 finds it among rays
            // that all have a list to walk (inline, a fifth of a wavefront's rays walked theirs while the others waited, and the
            // kernel's registers cost it two of its five wavefronts per SIMD: measured, no faster than this and 10 % slower alone)
            const int code = (DEFER && !INL) ? 2 : mesh_shadow_cells<FC, ULS, COOP ? 8u : 1u>(ltri, le0, le1, ldepth, mesh, lr, h.z, cnt, fc_lane);
            hit = code == 1;
            if (code == 2) {  // within rounding of a bounding box: the reference's own walk decides
                if (DEFER) deferred = true;
                else if (COOP) hit = box_hit(lr, ld3(mesh.bmin), ld3(mesh.bmax), RTU_BIGFLOAT) &&
                                     mesh_hit_coop<STACK, CULL, FC>(mesh, lr, shadow, h, stk, cnt, stride, lds_nodes, s.walk_stack_limit);
                else hit = mesh_hit<STACK, STATS, CULL, FC>(mesh, lr, shadow, h, stk, cnt, s.walk_stack_limit);
            }
        } else if (DEFER) {
            const RTU_CONST DevMesh& mesh = meshes[n.mesh_id];
            if (box_hit(lr, ld3(mesh.bmin), ld3(mesh.bmax), RTU_BIGFLOAT)) deferred = true;
            hit = false;
        } else if (COOP) {
            const RTU_CONST DevMesh& mesh = meshes[n.mesh_id];
            hit = box_hit(lr, ld3(mesh.bmin), ld3(mesh.bmax), RTU_BIGFLOAT) &&  // TriObj::IntersectRay's own box test (:335)
                  mesh_hit_coop<STACK, CULL, FC>(mesh, lr, shadow, h, stk, cnt, stride, lds_nodes, s.walk_stack_limit);
        } else {
            hit = (s.dbg & 32u) ? false : mesh_hit<STACK, STATS, CULL, FC>(meshes[n.mesh_id], lr, shadow, h, stk, cnt, s.walk_stack_limit);
        }
        if (hit) {
            any = true;
            best = (int)k;  // h.p / h.N are in node k's space
            lp = h.p;
            lN = h.N;
        }
    }
    // FromNodeCoords for the hit node and its ancestors (scene.h:508-512, applied by the reference as
    // the recursion unwinds), once, for the final hit: only h.z (and h.front) feed later tests.
    if (best >= 0 && !shadow) {
        h.node = best;
        h.p = lp;
        h.N = lN;
        const DevNode* gn = s.nodes;  // per-lane node index: ordinary loads
        for (int j = best; j >= 0; j = gn[j].parent) {
            RTU_TOUCH(t_xform, 1);
            from_node(gn[j], h);
        }
    }
    return any;
}

// ---------------------------------------------------------------------------
// Textures: Texture::TileClamp (scene.h:354-365), TextureFile::Sample bilinear with
// tiling (texture.cpp:95-121), TextureChecker::Sample (:125-133), TextureMap::Sample (scene.h:382),
// TexturedColor::Sample / SampleEnvironment (:421-431). Point sampling: the reference's Shade()
// calls Sample(hInfo.uvw) without derivatives. Same float operations as the reference; only
// atan2f / asinf (sphere uv, environment direction) are the device library's.
__device__ __forceinline__ f3 tile_clamp(f3 uvw) {
    f3 u = mk3(uvw.x - (float)(int)uvw.x, uvw.y - (float)(int)uvw.y, uvw.z - (float)(int)uvw.z);
    if (u.x < 0) u.x += 1;
    if (u.y < 0) u.y += 1;
    if (u.z < 0) u.z += 1;
    return u;
}
__device__ __forceinline__ f3 c24(const uint8_t* p) { return mk3((float)p[0] / 255.0f, (float)p[1] / 255.0f, (float)p[2] / 255.0f); }
__device__ __forceinline__ f3 texture_sample(const DevTexture& t, f3 uvw) {
    const f3 u = tile_clamp(uvw);
    if (t.type == RTU_TEX_CHECKER) {
        const f3 c1 = ld3(t.color1), c2 = ld3(t.color2);
        if (u.x <= 0.5f) return u.y <= 0.5f ? c1 : c2;
        return u.y <= 0.5f ? c2 : c1;
    }
    const int width = t.width, height = t.height;
    if (width + height == 0) return mk3(0, 0, 0);
    const float x = (float)width * u.x, y = (float)height * u.y;
    int ix = (int)x, iy = (int)y;
    const float fx = x - (float)ix, fy = y - (float)iy;
    if (ix < 0) ix -= (ix / width - 1) * width;
    if (ix >= width) ix -= (ix / width) * width;
    int ixp = ix + 1;
    if (ixp >= width) ixp -= width;
    if (iy < 0) iy -= (iy / height - 1) * height;
    if (iy >= height) iy -= (iy / height) * height;
    int iyp = iy + 1;
    if (iyp >= height) iyp -= height;
    const uint8_t* d = t.rgb;
    return ((c24(d + 3 * ((size_t)iy * width + ix)) * ((1 - fx) * (1 - fy)) + c24(d + 3 * ((size_t)iy * width + ixp)) * (fx * (1 - fy))) +
            c24(d + 3 * ((size_t)iyp * width + ix)) * ((1 - fx) * fy)) +
           c24(d + 3 * ((size_t)iyp * width + ixp)) * (fx * fy);
}
__device__ __forceinline__ f3 map_sample(const DevScene& s, const RtuTexMap& m, f3 uvw) {
    if (m.texture < 0) return mk3(0, 0, 0);
    return texture_sample(s.textures[m.texture], mat_mul(m.itm, uvw - ld3(m.pos)));  // TransformTo, scene.h:235
}
// TexturedColor::Sample of material colour k (RTU_MAP_*) of material mtl
template <bool TEX>
__device__ __forceinline__ f3 mtl_color(const DevScene& s, int mtl, int k, f3 color, f3 uvw) {
    if (!TEX || !s.mat_maps) return color;
    const RtuTexMap& m = s.mat_maps[4 * mtl + k];
    return m.present ? color * map_sample(s, m, uvw) : color;
}
__device__ __forceinline__ f3 env_color_sample(const DevScene& s, const RtuEnvColor& e, const RtuTexMap& m, f3 uvw) {
    const f3 c = ld3(e.color);
    if (!e.has_map) return c;
    if (e.map_is_null || !m.present) return c * mk3(0, 0, 0);
    return c * map_sample(s, m, uvw);
}
// background.Sample(Point3(x/imgWidth, y/imgHeight, 0)), RenderFunctions.cpp:145
template <bool TEX>
__device__ __forceinline__ f3 background_sample(const DevScene& s, int x, int y) {
    if (!TEX) return ld3(s.background);
    return env_color_sample(s, s.bg, s.bg_map, mk3((float)x / (float)s.img_w, (float)y / (float)s.img_h, 0));
}
// environment.SampleEnvironment(dir), scene.h:425-431
__device__ __forceinline__ f3 env_sample(const DevScene& s, f3 dir) {
    if (!s.env.has_map) return ld3(s.environment);
    const float z = asinf(-dir.z) / 3.14159265358979323846f + 0.5f;
    const float den = (float)((double)fabsf(dir.x) + (double)fabsf(dir.y));  // fabs() of a float promotes to double
    const float x = dir.x / den, y = dir.y / den;
    const f3 uvw = mk3(0.5f, 0.5f, 0.0f) + (mk3(0.5f, 0.5f, 0) * x + mk3(-0.5f, 0.5f, 0) * y) * z;
    return env_color_sample(s, s.env, s.env_map, uvw);
}

// ---- sample streams of recipe S (include/rtu_render.h states the contract) -----------------
__device__ __forceinline__ uint32_t mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}
__device__ __forceinline__ uint32_t rand31(uint32_t key, uint32_t purpose) { return mix32(key ^ mix32(purpose * 0x9e3779b9U + 0x85ebca6bU)) >> 1; }
__device__ __forceinline__ uint32_t sample_key(uint32_t pixel, uint32_t sample) { return mix32(mix32(pixel + 0x68bc21ebU) ^ (sample * 0x9e3779b9U + 1U)); }
__device__ __forceinline__ uint32_t child_key(uint32_t key, uint32_t slot) { return mix32(key + (slot + 1U) * 0x632be5abU); }
#define RTU_DRAW_LENS  0u
#define RTU_DRAW_LIGHT 16u
#define RTU_DRAW_REFR1 0x10000u
#define RTU_DRAW_REFR2 0x20000u
#define RTU_DRAW_REFL  0x30000u
#define RTU_DRAW_GATHER 0x40000u   // recipe P: the two numbers of SampleHemiSphereCosine
#define RTU_SLOT_GATHER 3u         // child_key slot of the hit of the gather ray
#define RTU_SLOT_AMBIENT_TREE 4u   // child_key slot of the Shade() tree lit by MonteCarlo()'s AmbientLight
#define RTU_RAND_MAX_F 2147483648.0f                  // static_cast<float>(RAND_MAX)
#define RTU_THETA_DIV  ((float)(2147483647 / (2 * 3.14159265358979323846)))  // static_cast<float>(RAND_MAX/(2 * M_PI))
// The key of the Shade() call a frame stands for, and whether the frame is sampled at all.
struct Smp {
    bool on;
    uint32_t key;
};

// sin and cos of sampleTheta in [0, 2 pi]: binary64, IEEE operations only, the sequence of the oracle's
// portable_sincos (quadrant reduction with a two-part pi/2, the fdlibm kernel polynomials).
__device__ __forceinline__ void portable_sincos(float t, float& sn, float& cs) {
    const double x = (double)t;
    const double kd = floor(x * 6.36619772367581382433e-01 + 0.5);
    const int k = (int)kd;
    const double y = (x - kd * 1.57079632673412561417e+00) - kd * 6.07710050650619224932e-11;
    const double y2 = y * y;
    const double ps = -1.66666666666666324348e-01 + y2 * (8.33333333332248946124e-03 + y2 * (-1.98412698298579493134e-04 +
                      y2 * (2.75573137070700676789e-06 + y2 * (-2.50507602534068634195e-08 + y2 * 1.58969099521155010221e-10))));
    const double pc = 4.16666666666666019037e-02 + y2 * (-1.38888888888741095749e-03 + y2 * (2.48015872894767294178e-05 +
                      y2 * (-2.75573143513906633035e-07 + y2 * (2.08757232129817482790e-09 + y2 * -1.13596475577881948265e-11))));
    const double s = y + (y * y2) * ps;
    const double c = 1.0 - (0.5 * y2 - (y2 * y2) * pc);
    const double so = (k & 1) ? c : s, co = (k & 1) ? s : c;
    sn = (float)((k & 2) ? -so : so);
    cs = (float)((((k + 1) & 2) != 0) ? -co : co);
}

// acos of a float in [-1, 1] in binary64 with IEEE operations only (fdlibm's e_acos rational approximation),
// rounded to float: the sequence of the oracle's portable_acos.
__device__ __forceinline__ double acos_poly(double z) {
    const double p = z * (1.66666666666666657415e-01 + z * (-3.25565818622400915405e-01 + z * (2.01212532134862925881e-01 +
                     z * (-4.00555345006794114027e-02 + z * (7.91534994289814532176e-04 + z * 3.47933107596021167570e-05)))));
    const double q = 1.0 + z * (-2.40339491173441421878e+00 + z * (2.02094576023350569471e+00 + z * (-6.88283971605453293030e-01 + z * 7.70381505559019352791e-02)));
    return p / q;
}
__device__ __forceinline__ float portable_acos(float xf) {
    const double pio2_hi = 1.57079632679489655800e+00, pio2_lo = 6.12323399573676603587e-17, pi = 3.14159265358979311600e+00;
    const double x = (double)xf;
    const double ax = fabs(x);
    double r;
    if (ax >= 1.0) r = x > 0 ? 0.0 : pi;
    else if (ax < 0.5) r = pio2_hi - (x - (pio2_lo - x * acos_poly(x * x)));
    else if (x < 0) {
        const double z = (1.0 + x) * 0.5, s = sqrt(z);
        r = pi - 2.0 * (s + (acos_poly(z) * s - pio2_lo));
    } else {
        const double z = (1.0 - x) * 0.5, s = sqrt(z);
        r = 2.0 * (s + acos_poly(z) * s);
    }
    return (float)r;
}
// SampleHemiSphereCosine(origin, normal, 1.0), RenderFunctions.cpp:320-337
__device__ __forceinline__ f3 sample_hemisphere_cosine(f3 normal, uint32_t key) {
    const float sampleX = (float)rand31(key, RTU_DRAW_GATHER) / RTU_RAND_MAX_F;        // :324
    const float samplePhi = (float)rand31(key, RTU_DRAW_GATHER + 1u) / RTU_THETA_DIV;  // :325
    const float sampleTheta = (float)(0.5 * (double)portable_acos(1 - 2 * sampleX));  // :326
    const f3 v1 = norm3(cross3(normal, mk3(sampleX, sampleX, sampleX)));               // :329
    const f3 v2 = norm3(cross3(v1, normal));                                           // :330
    float st, ct, sp, cp;
    portable_sincos(sampleTheta, st, ct);
    portable_sincos(samplePhi, sp, cp);
    return (normal * (1.0f * ct) + v1 * ((1.0f * st) * cp)) + v2 * ((1.0f * st) * sp);  // :332-334
}

// SampleSphere (RenderFunctions.cpp:282-301): a point of the cube [-radius, radius]^3, drawn again
// while it lies outside the sphere (at most 64 attempts).
__device__ __forceinline__ f3 sample_sphere(float radius, uint32_t key, uint32_t base) {
    f3 offset = mk3(0, 0, 0);
    const float div = RTU_RAND_MAX_F / (radius * 2);
    for (uint32_t attempt = 0; attempt < 64u; attempt++) {
        const float rand1 = -radius + (float)rand31(key, base + 3u * attempt) / div;       // :291
        const float rand2 = -radius + (float)rand31(key, base + 3u * attempt + 1u) / div;  // :292
        const float rand3 = -radius + (float)rand31(key, base + 3u * attempt + 2u) / div;  // :293
        offset = mk3(rand1, rand2, rand3);
        if (!(len3(offset) > radius)) break;  // :297
    }
    return offset;
}
// sampledNormal of mtlFunctions.cpp:162-165 / :225-227 / :275-277; SampleSphere(..., 0) == (0,0,0)
__device__ __forceinline__ f3 sampled_normal(f3 p, f3 N, Smp smp, float glossiness, uint32_t base) {
    f3 sampleOrigin = p + N;
    f3 sampledOffset = mk3(0, 0, 0);
    if (smp.on && glossiness > 0) sampledOffset = sample_sphere(glossiness, smp.key, base);
    return norm3((sampleOrigin + sampledOffset) - p);
}
__device__ __forceinline__ f3 reflect_dir(f3 dir, f3 sn) {  // :207, :239, :280
    float k = 2 * dot3(dir, sn);
    return norm3(dir - sn * k);
}

// Snell / Fresnel terms of mtlFunctions.cpp:168-203,236-237. Recomputed from the
// frame whenever a stage resumes (pure ALU) instead of being saved.
struct Refr {
    f3    sn;          // sampled normal (:162-165)
    f3    sn2;         // the second sample (:225-227), which the refracted and the Fresnel ray use
    float cosTheta1;   // after clamping
    float sinTheta2, cosTheta2;
    float n1, n2;
    f3    SVector;
};
__device__ __forceinline__ Refr refraction_terms(f3 dir, f3 p, f3 N, bool front, float ior, Smp smp, float glossiness) {
    Refr r;
    r.sn = sampled_normal(p, N, smp, glossiness, RTU_DRAW_REFR1);
    r.sn2 = r.sn;
    if (smp.on && glossiness > 0) r.sn2 = sampled_normal(p, N, smp, glossiness, RTU_DRAW_REFR2);
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

}  // namespace

#endif
