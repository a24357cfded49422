// rtu_oracle.cpp — CPU restatement of the reference's per-pixel render path.
//
// *** TEST INFRASTRUCTURE. NOT PART OF THE PRODUCT PATH. ***
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
// this library, and only as the checker / the timed CPU baseline. The product
// (librtu_hip.so + librtu_host.so) never links, loads or calls it.
//
// Parity status: PINNED. In the authoring container this restatement is checked
// bit-for-bit (float z AND linear float RGB of every pixel) against the
// reference's own functions compiled from /root/reference (oracle/_ref/ref_render,
// built by oracle/ref_harness/Makefile) on all five BASELINE configs, and against
// the committed goldens under tests/golden/ (see tests/test_oracle.py).
//
// Every function cites the reference file:line it restates. Arithmetic rules
// (SURVEY.md Appendix B): IEEE binary32, no FMA contraction (-ffp-contract=off),
// the reference's evaluation order, and its float->double promotions ("fp64
// islands") reproduced exactly. Build: g++ -O2 -ffp-contract=off (never
// -march=native / -Ofast).
//
// Scope: deterministic "recipe W" (SURVEY.md §8c): one ray through every pixel
// centre, Trace + Shade(...,5); scenes with stochastic features (soft shadows, glossy
// bounces, depth of field) are rejected there with RTU_ORACLE_ERR_STOCHASTIC.
// "Recipe S" (row f1, rtu_oracle_render_samples): the sample loop of Render() with those
// features, rand() replaced by counter-based sample streams (see "Sample streams" below).
// Its pin: with the sequential stream and libm's sinf/cosf it equals, bit for bit, the
// reference built with rand() wrapped to the same stream (oracle/ref_harness --spp).
#include "rtu_oracle.h"

#include <atomic>
#include <cmath>
#include <cstring>
#include <thread>
#include <vector>

namespace {

// ---------------------------------------------------------------------------
// cyPoint.h Point3<float> (ExternalLibrary/cyPoint.h:259-349)
struct V3 {
    float x, y, z;
};
inline V3 mk(float x, float y, float z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
inline V3 operator+(V3 a, V3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }          // :307
inline V3 operator-(V3 a, V3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }          // :308
inline V3 operator-(V3 a) { return mk(-a.x, -a.y, -a.z); }                               // :304
inline V3 operator*(V3 a, float s) { return mk(a.x * s, a.y * s, a.z * s); }             // :313
inline V3 operator/(V3 a, float s) { return mk(a.x / s, a.y / s, a.z / s); }             // :314
inline float dot(V3 a, V3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }             // :348 via Sum :296
inline V3 cross(V3 a, V3 b) {                                                            // :346
    return mk(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
inline float length_sq(V3 a) { return dot(a, a); }                                       // :292
inline float length(V3 a) { return sqrtf(length_sq(a)); }                                // :293, cyCore.h:161
inline V3 normalized(V3 a) { return a / length(a); }                                     // :295
inline V3 ld3(const float* p) { return mk(p[0], p[1], p[2]); }

// cyColor.h Color (ExternalLibrary/cyColor.h:52-122)
struct C3 {
    float r, g, b;
};
inline C3 mkc(float r, float g, float b) { C3 c; c.r = r; c.g = g; c.b = b; return c; }
inline C3 operator+(C3 a, C3 b) { return mkc(a.r + b.r, a.g + b.g, a.b + b.b); }
inline C3 operator*(C3 a, C3 b) { return mkc(a.r * b.r, a.g * b.g, a.b * b.b); }
inline C3 operator*(C3 a, float n) { return mkc(a.r * n, a.g * n, a.b * n); }
inline C3& operator+=(C3& a, C3 b) { a.r += b.r; a.g += b.g; a.b += b.b; return a; }
inline bool not_black(C3 c) { return c.r != 0 || c.g != 0 || c.b != 0; }                 // cyColor.h:115 (!=)
inline C3 ldc(const float* p) { return mkc(p[0], p[1], p[2]); }

struct Ray {  // scene.h:59-69
    V3 p, dir;
};

struct Hit {  // scene.h:150-163
    float z;
    V3 p, N, uvw;
    int node;
    bool front;
};
inline Hit new_hit() {  // HitInfo::Init, scene.h:162
    Hit h;
    h.z = RTU_BIGFLOAT;
    h.p = mk(0, 0, 0);  // uninitialised in the reference; never read before written
    h.N = mk(0, 0, 0);
    h.uvw = mk(0.5f, 0.5f, 0.5f);
    h.node = -1;
    h.front = true;
    return h;
}

// std::max / std::min exactly as libstdc++ defines them (NaN behaviour matters).
inline float smax(float a, float b) { return (a < b) ? b : a; }
inline float smin(float a, float b) { return (b < a) ? b : a; }

struct Ctx {
    const RtuSceneDesc* s;
    RtuOracleStats st;  // per-thread counters
    // sample streams ("next" row f1): where the integers that replace rand() come from
    bool sequential;     // true: the n-th rand() call of a pixel sample (pins the restatement against the reference)
    bool libm_trig;      // true: cosf/sinf of libm as the reference calls them; false: the portable sincos the device uses
    uint32_t seq_key, seq_counter;
    uint64_t gather_rays;
};

// ---------------------------------------------------------------------------
// Sample streams. The reference draws from rand() (shared by its threads, seeded with time(NULL):
// RenderFunctions.cpp:60) — nothing to reproduce bit for bit. Restated: every rand() call becomes
// draw(), an integer in [0, RAND_MAX] from a counter-based hash; the float expressions around it are
// the reference's. Two ways to index the stream:
//   sequential  rand31(sample_key(pixel, sample), n) for the n-th call while that pixel sample is
//               evaluated — the reference's own call order, so a reference build whose rand() is
//               wrapped (oracle/ref_harness, -Wl,--wrap=rand) must give the same image bit for bit;
//   keyed       rand31(key of the Shade() call, purpose) — independent of evaluation order, which is
//               what a level-synchronous device evaluation needs. Keys: the primary Shade() of a
//               sample has sample_key; a child Shade() has child_key(parent, slot).
inline uint32_t mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}
inline uint32_t rand31(uint32_t key, uint32_t idx) { return mix32(key ^ mix32(idx * 0x9e3779b9U + 0x85ebca6bU)) >> 1; }
inline uint32_t sample_key(uint32_t pixel, uint32_t sample) { return mix32(mix32(pixel + 0x68bc21ebU) ^ (sample * 0x9e3779b9U + 1U)); }
inline uint32_t child_key(uint32_t key, uint32_t slot) { return mix32(key + (slot + 1U) * 0x632be5abU); }
enum { DRAW_DOF = 0, DRAW_LIGHT = 16, DRAW_REFR1 = 0x10000, DRAW_REFR2 = 0x20000, DRAW_REFL = 0x30000 };
enum { SLOT_MAIN = 0, SLOT_A = 1, SLOT_C = 2 };  // refracted (or TIR) ray, Fresnel ray, reflection ray
inline float draw(Ctx& cx, uint32_t key, uint32_t idx) {  // static_cast<float>(rand())
    return (float)(cx.sequential ? rand31(cx.seq_key, cx.seq_counter++) : rand31(key, idx));
}
const float RAND_MAX_F = (float)2147483647;                          // static_cast<float>(RAND_MAX)
const float THETA_DIV = (float)(2147483647 / (2 * M_PI));            // static_cast<float>(RAND_MAX/(2 * M_PI))

// sin and cos of a float angle in [0, 2*pi], evaluated in binary64 with IEEE operations only (the same
// sequence on the device), rounded to float: within 1 ulp of libm's sinf/cosf (tests/test_oracle.py).
inline void portable_sincos(float t, float* sn, float* cs) {
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
    double so, co;
    switch (k & 3) {
        case 0: so = s; co = c; break;
        case 1: so = c; co = -s; break;
        case 2: so = -s; co = -c; break;
        default: so = -c; co = s; break;
    }
    *sn = (float)so;
    *cs = (float)co;
}
inline void sincos_of(const Ctx& cx, float t, float* sn, float* cs) {
    if (cx.libm_trig) { *sn = sinf(t); *cs = cosf(t); }  // sin(float) / cos(float): the float overloads
    else portable_sincos(t, sn, cs);
}

// ---------------------------------------------------------------------------
// cyMatrix.h Matrix3 * Point3 (cyMatrix.h:543-547), column-major data[9].
inline V3 mat_mul(const float* m, V3 p) {
    return mk((p.x * m[0] + p.y * m[3]) + p.z * m[6],
              (p.x * m[1] + p.y * m[4]) + p.z * m[7],
              (p.x * m[2] + p.y * m[5]) + p.z * m[8]);
}
// Transformation::TransposeMult (scene.h:253-260): column dot products.
inline V3 mat_tmul(const float* m, V3 d) {
    return mk(dot(mk(m[0], m[1], m[2]), d), dot(mk(m[3], m[4], m[5]), d), dot(mk(m[6], m[7], m[8]), d));
}
// Node::ToNodeCoords (scene.h:501-507)
inline Ray to_node(const RtuNode& n, const Ray& ray) {
    V3 pos = ld3(n.pos);
    Ray r;
    r.p = mat_mul(n.itm, ray.p - pos);                          // TransformTo, scene.h:235
    r.dir = mat_mul(n.itm, (ray.p + ray.dir) - pos) - r.p;
    return r;
}
// Node::FromNodeCoords (scene.h:508-512)
inline void from_node(const RtuNode& n, Hit& h) {
    h.p = mat_mul(n.tm, h.p) + ld3(n.pos);                      // TransformFrom, scene.h:236
    h.N = normalized(mat_tmul(n.itm, h.N));                     // VectorTransformFrom, scene.h:242
}

// ---------------------------------------------------------------------------
// Box::IntersectRay (objFunctions.cpp:143-254). Returns the slab interval.
inline bool box_slabs(const Ray& r, V3 allMin, V3 allMax, float& tEntry, float& tExit) {
    if (allMin.x > allMax.x || allMin.y > allMax.y || allMin.z > allMax.z) return false;  // IsEmpty, scene.h:85
    if (r.dir.x == 0) {
        float ty0 = (allMin.y - r.p.y) / r.dir.y;
        float ty1 = (allMax.y - r.p.y) / r.dir.y;
        float tz0 = (allMin.z - r.p.z) / r.dir.z;
        float tz1 = (allMax.z - r.p.z) / r.dir.z;
        if (ty0 > ty1) { float t = ty1; ty1 = ty0; ty0 = t; }
        if (tz0 > tz1) { float t = tz1; tz1 = tz0; tz0 = t; }
        tEntry = smax(tz0, ty0);
        tExit = smin(tz1, ty1);
    } else if (r.dir.y == 0) {
        float tx0 = (allMin.x - r.p.x) / r.dir.x;
        float tx1 = (allMax.x - r.p.x) / r.dir.x;
        float tz0 = (allMin.z - r.p.z) / r.dir.z;
        float tz1 = (allMax.z - r.p.z) / r.dir.z;
        if (tx0 > tx1) { float t = tx1; tx1 = tx0; tx0 = t; }
        if (tz0 > tz1) { float t = tz1; tz1 = tz0; tz0 = t; }
        tEntry = smax(tz0, tx0);
        tExit = smin(tz1, tx1);
    } else if (r.dir.z == 0) {
        float tx0 = (allMin.x - r.p.x) / r.dir.x;
        float tx1 = (allMax.x - r.p.x) / r.dir.x;
        float ty0 = (allMin.y - r.p.y) / r.dir.y;
        float ty1 = (allMax.y - r.p.y) / r.dir.y;
        if (tx0 > tx1) { float t = tx1; tx1 = tx0; tx0 = t; }
        if (ty0 > ty1) { float t = ty1; ty1 = ty0; ty0 = t; }
        tEntry = smax(ty0, tx0);
        tExit = smin(ty1, tx1);
    } else {
        float tx0 = (allMin.x - r.p.x) / r.dir.x;
        float tx1 = (allMax.x - r.p.x) / r.dir.x;
        float ty0 = (allMin.y - r.p.y) / r.dir.y;
        float ty1 = (allMax.y - r.p.y) / r.dir.y;
        float tz0 = (allMin.z - r.p.z) / r.dir.z;
        float tz1 = (allMax.z - r.p.z) / r.dir.z;
        if (tx0 > tx1) { float t = tx1; tx1 = tx0; tx0 = t; }
        if (ty0 > ty1) { float t = ty1; ty1 = ty0; ty0 = t; }
        if (tz0 > tz1) { float t = tz1; tz1 = tz0; tz0 = t; }
        tEntry = smax(smax(tx0, ty0), tz0);
        tExit = smin(smin(tx1, ty1), tz1);
    }
    return true;
}
inline bool box_hit(const Ray& r, V3 bmin, V3 bmax, float t_max) {  // objFunctions.cpp:248-253
    float tEntry, tExit;
    if (!box_slabs(r, bmin, bmax, tEntry, tExit)) return false;
    return tEntry <= tExit && tEntry < t_max;
}
// BVHBoxIntersection (objFunctions.cpp:408-522): tEntry + 0.01 (fp64 add, :517) or t_max.
inline float bvh_box(const Ray& r, const RtuBvhNode& n, float t_max) {
    float tEntry, tExit;
    if (!box_slabs(r, ld3(n.bmin), ld3(n.bmax), tEntry, tExit)) return -t_max;  // :415-417
    if (tEntry <= tExit && tEntry < t_max) return (float)((double)tEntry + 0.01);
    return t_max;
}

// ---------------------------------------------------------------------------
// Sphere::IntersectRay (objFunctions.cpp:15-104)
inline void sphere_uv(Hit& h) {  // :38-41 (atan2f/asinf float, the rest fp64)
    float u = (float)(0.5 - (double)atan2f(h.N.x, h.N.y) / (2 * M_PI));
    float v = (float)(0.5 + (double)asinf(h.N.z) / M_PI);
    h.uvw = mk(u, v, 0);
}
bool sphere_hit(const Ray& ray, Hit& h) {
    if (!box_hit(ray, mk(-1, -1, -1), mk(1, 1, 1), RTU_BIGFLOAT)) return false;  // :17, objects.h:25
    float a = dot(ray.dir, ray.dir);
    float b = 2 * dot(ray.p - mk(0, 0, 0), ray.dir);
    float c = dot(ray.p, ray.p) - 1;
    float sqrtCheck = b * b - 4 * a * c;
    float m = (-b + sqrtf(sqrtCheck)) / (2 * a);
    float n = (-b - sqrtf(sqrtCheck)) / (2 * a);
    if (m == n && m < h.z && (double)m >= 0.001) {  // :29
        h.z = m;
        h.front = true;
        V3 temp = ray.p + ray.dir * h.z;
        h.N = normalized(temp);
        h.p = temp;
        sphere_uv(h);
        return true;
    } else if (m < n && m < h.z && (((double)m >= 0.001) | ((double)n >= 0.001))) {  // :45
        if ((double)m <= 0.001 && (double)n > 0.001 && n < h.z) {
            h.z = n;
            h.front = false;
        } else if ((double)m > 0.001) {
            h.z = m;
            h.front = true;
        }
        V3 temp = ray.p + ray.dir * h.z;  // stale h.z when neither branch fired (Appendix C-1)
        h.N = h.front ? normalized(temp) : -normalized(temp);
        h.p = temp;
        sphere_uv(h);
        return true;
    } else if (n < m && n < h.z && (((double)m >= 0.001) | ((double)n >= 0.001))) {  // :73
        if ((double)n <= 0.001 && (double)m > 0.001 && m < h.z) {
            h.z = m;
            h.front = false;
        } else if ((double)n > 0.001) {
            h.z = n;
            h.front = true;
        }
        V3 temp = ray.p + ray.dir * h.z;
        h.N = h.front ? normalized(temp) : -normalized(temp);
        h.p = temp;
        sphere_uv(h);
        return true;
    }
    return false;
}

// Plane::IntersectRay (objFunctions.cpp:107-140)
bool plane_hit(const Ray& ray, Hit& h) {
    if (!box_hit(ray, mk(-1, -1, 0), mk(1, 1, 0), RTU_BIGFLOAT)) return false;  // :109, objects.h:37
    if (ray.dir.z != 0) {
        float t = (-ray.p.z) / (ray.dir.z);
        if ((double)t > 0.001 && t < h.z) {
            V3 q = ray.p + ray.dir * t;
            if (q.x > -1 && q.x < 1 && q.y > -1 && q.y < 1) {
                if (ray.p.z > 0) {
                    h.front = true;
                    h.N = mk(0, 0, 1);
                } else {
                    h.front = false;
                    h.N = mk(0, 0, -1);
                }
                q = mk(q.x, q.y, 0);
                h.z = t;
                h.p = q;
                h.uvw = mk((q.x + 1) / 2, (q.y + 1) / 2, 0);
                return true;
            }
        }
    }
    return false;
}

// cyTriMesh::Interpolate (cyTriMesh.h:191)
inline V3 interp(const float* arr, const uint32_t* face, V3 bc) {
    return (ld3(arr + 3 * face[0]) * bc.x + ld3(arr + 3 * face[1]) * bc.y) + ld3(arr + 3 * face[2]) * bc.z;
}
// Point2::Cross (cyPoint.h:247, 249): (-a.y)*b.x + a.x*b.y
inline float cross2(float ax, float ay, float bx, float by) { return (-ay) * bx + ax * by; }

// TriObj::IntersectTriangle (objFunctions.cpp:257-328)
bool tri_hit(Ctx& cx, const RtuMesh& mesh, const Ray& ray, Hit& h, uint32_t faceID) {
    cx.st.tri_tests++;
    const uint32_t* fv = mesh.f + 3 * faceID;
    V3 A = ld3(mesh.v + 3 * fv[0]);
    V3 B = ld3(mesh.v + 3 * fv[1]);
    V3 C = ld3(mesh.v + 3 * fv[2]);
    V3 N = normalized(cross(B - A, C - A));
    if (dot(ray.dir, N) != 0) {
        float t = dot(A - ray.p, N) / dot(ray.dir, N);
        if ((double)t > 0.00001 && t < h.z) {  // :270
            V3 q = ray.p + ray.dir * t;
            float maxNormalAxis = smax(smax(fabsf(N.x), fabsf(N.y)), fabsf(N.z));
            float ax = 0, ay = 0, bx = 0, by = 0, cx2 = 0, cy2 = 0, qx = 0, qy = 0;  // Point2() leaves garbage; all three tests below cover every non-NaN N
            if (maxNormalAxis == fabsf(N.x)) {
                ax = A.y; ay = A.z; bx = B.y; by = B.z; cx2 = C.y; cy2 = C.z; qx = q.y; qy = q.z;
            } else if (maxNormalAxis == fabsf(N.y)) {
                ax = A.x; ay = A.z; bx = B.x; by = B.z; cx2 = C.x; cy2 = C.z; qx = q.x; qy = q.z;
            } else if (maxNormalAxis == fabsf(N.z)) {
                ax = A.x; ay = A.y; bx = B.x; by = B.y; cx2 = C.x; cy2 = C.y; qx = q.x; qy = q.y;
            }
            // :298-300, "/2.0" in fp64 (exact)
            float TriABCArea = (float)((double)cross2(cx2 - ax, cy2 - ay, bx - ax, by - ay) / 2.0);
            float TriAPCArea = (float)((double)cross2(cx2 - ax, cy2 - ay, qx - ax, qy - ay) / 2.0);
            float TriABPArea = (float)((double)cross2(qx - ax, qy - ay, bx - ax, by - ay) / 2.0);
            float BC1 = TriAPCArea / TriABCArea;
            float BC2 = TriABPArea / TriABCArea;
            float BC3 = (float)(1.0 - (double)BC1 - (double)BC2);  // :304
            if (BC1 > 0 && BC2 > 0 && BC3 > 0 && BC1 < 1 && BC2 < 1 && BC3 < 1) {
                V3 bc = mk(BC3, BC1, BC2);
                h.front = dot(ray.dir, N) < 0;
                if (mesh.vt && mesh.ft) h.uvw = interp(mesh.vt, mesh.ft + 3 * faceID, bc);
                else h.uvw = mk(0, 0, 0);  // the reference would dereference a NULL vt array here
                h.N = normalized(interp(mesh.vn, mesh.fn + 3 * faceID, bc));
                h.z = t;
                h.p = interp(mesh.v, fv, bc);
                cx.st.tri_accepts++;
                return true;
            }
        }
    }
    return false;
}

// Test hook (rtu_oracle_debug_all_triangles): every triangle of a mesh is tested, in element order, whatever the boxes say.
// NOT the reference's algorithm — it exists to find rays for which the reference's own box arithmetic hides a triangle its
// triangle test would accept (a ray clipping a box corner within rounding), so that tests can aim at them.
static bool g_all_triangles = false;

// TriObj::IntersectRay (objFunctions.cpp:333-406)
bool mesh_hit(Ctx& cx, const RtuMesh& mesh, const Ray& ray, Hit& h) {
    bool hitResult = false;
    if (g_all_triangles) {
        for (uint32_t i = 0; i < mesh.n_elements; i++) hitResult |= tri_hit(cx, mesh, ray, h, mesh.elements[i]);
        return hitResult;
    }
    if (!box_hit(ray, ld3(mesh.bound_min), ld3(mesh.bound_max), RTU_BIGFLOAT)) return false;  // :337
    cx.st.mesh_entries++;
    static const int STACK_MAX = 256;  // reference: 100, overflow is UB there
    unsigned int stack[STACK_MAX];
    int stackTop = 0;
    stack[0] = 1;  // GetRootNodeID, cyBVH.h:76
    while (stackTop >= 0) {
        unsigned int cur = stack[stackTop];
        stackTop--;
        const RtuBvhNode& node = mesh.bvh[cur];
        if (node.count == 0) {
            cx.st.inner_visits++;
            unsigned int c1 = node.index, c2 = node.index + 1;
            float t1 = bvh_box(ray, mesh.bvh[c1], RTU_BIGFLOAT);
            float t2 = bvh_box(ray, mesh.bvh[c2], RTU_BIGFLOAT);
            if (t1 <= t2) {  // :361
                if (t2 != RTU_BIGFLOAT) stack[++stackTop] = c2;
                if (t1 != RTU_BIGFLOAT) stack[++stackTop] = c1;
            } else if (t1 > t2) {  // :376
                if (t1 != RTU_BIGFLOAT) stack[++stackTop] = c1;
                if (t2 != RTU_BIGFLOAT) stack[++stackTop] = c2;
            }
            if (stackTop >= STACK_MAX - 2) return hitResult;  // cannot happen for depth <= 100 trees
        } else {
            cx.st.leaf_visits++;
            cx.st.leaf_elems += node.count;
            for (uint32_t i = 0; i < node.count; i++)  // :394-396
                hitResult |= tri_hit(cx, mesh, ray, h, mesh.elements[node.index + i]);
        }
    }
    return hitResult;
}

inline bool object_hit(Ctx& cx, const RtuNode& n, const Ray& local, Hit& h) {
    cx.st.node_tests++;
    switch (n.obj_type) {
        case RTU_OBJ_SPHERE: return sphere_hit(local, h);
        case RTU_OBJ_PLANE: return plane_hit(local, h);
        case RTU_OBJ_TRIMESH: return mesh_hit(cx, cx.s->meshes[n.mesh_id], local, h);
    }
    return false;
}

// Trace (RenderFunctions.cpp:181-213). Children of node k are the nodes whose
// parent == k, in pre-order: first child k+1, next sibling = subtree_end.
bool trace(Ctx& cx, const Ray& r, int k, Hit& h) {
    const RtuNode& n = cx.s->nodes[k];
    bool currentNodeIsHit = false;
    if (n.obj_type != RTU_OBJ_NONE) {
        currentNodeIsHit = object_hit(cx, n, to_node(n, r), h);
        if (currentNodeIsHit) {
            h.node = k;
            from_node(n, h);
        }
    }
    for (int c = k + 1; c < n.subtree_end; c = cx.s->nodes[c].subtree_end) {
        bool childIsHit = trace(cx, to_node(n, r), c, h);
        if (childIsHit) from_node(n, h);
        currentNodeIsHit = currentNodeIsHit | childIsHit;
    }
    return currentNodeIsHit;
}

// ShadowTrace (RenderFunctions.cpp:216-240)
bool shadow_trace(Ctx& cx, const Ray& r, int k, Hit& h) {
    const RtuNode& n = cx.s->nodes[k];
    if (n.obj_type != RTU_OBJ_NONE) {
        if (object_hit(cx, n, to_node(n, r), h)) return true;
    }
    for (int c = k + 1; c < n.subtree_end; c = cx.s->nodes[c].subtree_end) {
        if (shadow_trace(cx, to_node(n, r), c, h)) return true;
    }
    return false;
}

// GenLight::Shadow (lightFunctions.cpp:27-37)
float shadow(Ctx& cx, const Ray& ray, float t_max) {
    cx.st.shadow_rays++;
    Hit h = new_hit();
    h.z = t_max;
    if (shadow_trace(cx, ray, 0, h)) {
        if (h.z > 0.0) return 0.0f;
    }
    return 1.0f;
}

// Light::Illuminate: AmbientLight (lights.h:32), DirectLight (lights.h:48),
// PointLight (lightFunctions.cpp:39-84; size > 0: one shadow ray towards a random point of the
// light's disk, :43-65).
C3 illuminate(Ctx& cx, const RtuLight& l, V3 p, uint32_t key, uint32_t light_index) {
    C3 intensity = ldc(l.intensity);
    if (l.type == RTU_LIGHT_AMBIENT) return intensity;
    if (l.type == RTU_LIGHT_DIRECT) {
        Ray sr; sr.p = p; sr.dir = -ld3(l.vec);
        return intensity * shadow(cx, sr, RTU_BIGFLOAT);
    }
    V3 position = ld3(l.vec);
    float shadowIntensity = 0.0f;
    if (l.size > 0) {
        float sampleR = draw(cx, key, DRAW_LIGHT + 2 * light_index) / (RAND_MAX_F / l.size);   // :47
        float sampleTheta = draw(cx, key, DRAW_LIGHT + 2 * light_index + 1) / THETA_DIV;        // :48
        float sn, cs;
        sincos_of(cx, sampleTheta, &sn, &cs);
        float offsetX = sampleR * cs;  // :49
        float offsetY = sampleR * sn;  // :50
        V3 samplePlaneNormal = normalized(position - p);                       // :52
        V3 v1 = normalized(cross(samplePlaneNormal, mk(0, 0, 1)));             // :55
        V3 v2 = normalized(cross(v1, samplePlaneNormal));                      // :56
        V3 currentSamplePos = (position + v1 * offsetX) + v2 * offsetY;        // :58
        Ray sr; sr.p = p; sr.dir = normalized(currentSamplePos - p);           // :60
        shadowIntensity += shadow(cx, sr, length(p - currentSamplePos));       // :62
    } else {
        Ray sr; sr.p = p; sr.dir = normalized(position - p);       // :76
        shadowIntensity += shadow(cx, sr, length(position - p));   // :78
    }
    float result = shadowIntensity;
    return (intensity * result) * (1 / length_sq(position - p));  // :83
}
// Light::Direction (lights.h:33,49,83)
inline V3 light_direction(const RtuLight& l, V3 p) {
    if (l.type == RTU_LIGHT_DIRECT) return ld3(l.vec);
    if (l.type == RTU_LIGHT_POINT) return normalized(p - ld3(l.vec));
    return mk(0, 0, 0);
}

// ---------------------------------------------------------------------------
// Textures ("next" row f2): Texture::TileClamp (scene.h:354-365), TextureFile::Sample
// (texture.cpp:95-121), TextureChecker::Sample (:125-133), TextureMap::Sample (scene.h:382),
// TexturedColor::Sample / SampleEnvironment (scene.h:421-431). Point sampling only: the
// reference's Shade() calls Sample(hInfo.uvw) without derivatives (mtlFunctions.cpp:132-289).
inline V3 tile_clamp(V3 uvw) {
    V3 u;
    u.x = uvw.x - (int)uvw.x;
    u.y = uvw.y - (int)uvw.y;
    u.z = uvw.z - (int)uvw.z;
    if (u.x < 0) u.x += 1;
    if (u.y < 0) u.y += 1;
    if (u.z < 0) u.z += 1;
    return u;
}
inline C3 c24(const uint8_t* p) { return mkc(p[0] / 255.0f, p[1] / 255.0f, p[2] / 255.0f); }  // Color24::ToColor, cyColor.h:214
inline C3 texture_sample(const RtuTexture& t, V3 uvw) {
    V3 u = tile_clamp(uvw);
    if (t.type == RTU_TEX_CHECKER) {
        if (u.x <= 0.5f) return u.y <= 0.5f ? ldc(t.color1) : ldc(t.color2);
        return u.y <= 0.5f ? ldc(t.color2) : ldc(t.color1);
    }
    const int width = t.width, height = t.height;
    if (width + height == 0) return mkc(0, 0, 0);
    float x = width * u.x;
    float y = height * u.y;
    int ix = (int)x;
    int iy = (int)y;
    float fx = x - ix;
    float fy = y - iy;
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
inline C3 map_sample(const RtuSceneDesc& s, const RtuTexMap& m, V3 uvw) {  // TextureMap::Sample
    if (m.texture < 0) return mkc(0, 0, 0);
    return texture_sample(s.textures[m.texture], mat_mul(m.itm, uvw - ld3(m.pos)));  // TransformTo, scene.h:235
}
// TexturedColor::Sample of material colour k (RTU_MAP_*)
inline C3 mtl_color(const RtuSceneDesc& s, int mtl_id, int k, const float* color, V3 uvw) {
    C3 c = ldc(color);
    if (!s.material_maps) return c;
    const RtuTexMap& m = s.material_maps[4 * mtl_id + k];
    return m.present ? c * map_sample(s, m, uvw) : c;
}
inline C3 env_color_sample(const RtuSceneDesc& s, const RtuEnvColor& e, const RtuTexMap& m, V3 uvw) {
    C3 c = ldc(e.color);
    if (!e.has_map) return c;
    if (e.map_is_null || !m.present) return c * mkc(0, 0, 0);  // TextureMap(NULL) samples black
    return c * map_sample(s, m, uvw);
}
// background.Sample(Point3(x/W, y/H, 0)), RenderFunctions.cpp:145
inline C3 background_sample(const RtuSceneDesc& s, int x, int y) {
    return env_color_sample(s, s.background, s.background_map, mk((float)x / s.camera.img_width, (float)y / s.camera.img_height, 0));
}
// environment.SampleEnvironment(dir), scene.h:425-431
inline C3 env_sample(const RtuSceneDesc& s, V3 dir) {
    if (!s.environment.has_map) return ldc(s.environment.color);
    float z = asinf(-dir.z) / float(M_PI) + 0.5f;
    float x = dir.x / (float)(fabs(dir.x) + fabs(dir.y));
    float y = dir.y / (float)(fabs(dir.x) + fabs(dir.y));
    V3 uvw = mk(0.5f, 0.5f, 0.0f) + (mk(0.5f, 0.5f, 0) * x + mk(-0.5f, 0.5f, 0) * y) * z;
    return env_color_sample(s, s.environment, s.environment_map, uvw);
}

// The LightList a Shade() call receives: the scene's lights, or the one AmbientLight that MonteCarlo()
// builds from its indirect estimate (RenderFunctions.cpp:585-590).
struct LightSet {
    const RtuLight* lights;
    uint32_t n;
};
C3 shade(Ctx& cx, int mtl_id, const Ray& ray, const Hit& hInfo, int bounceCount, uint32_t key, const LightSet& ls);

inline C3 shade_node(Ctx& cx, const Hit& h, const Ray& ray, int bounce, uint32_t key, const LightSet& ls) {
    int mid = cx.s->nodes[h.node].material_id;
    if (mid < 0) return mkc(1, 1, 1);  // null material => white (reference would crash)
    return shade(cx, mid, ray, h, bounce, key, ls);
}

// SampleSphere (RenderFunctions.cpp:282-301): a point of the cube [-radius, radius]^3, drawn again
// while it lies outside the sphere. radius == 0 gives (0,0,0) (x / inf == 0) but still draws three
// numbers, which matters to the sequential stream only. At most 64 attempts (p < 1e-20).
inline V3 sample_sphere(Ctx& cx, float radius, uint32_t key, uint32_t base) {
    V3 offset = mk(0, 0, 0);
    if (!(radius > 0)) {
        if (cx.sequential) cx.seq_counter += 3;
        return offset;
    }
    for (uint32_t attempt = 0; attempt < 64; attempt++) {
        float rand1 = -radius + draw(cx, key, base + 3 * attempt) / (RAND_MAX_F / (radius * 2));      // :291
        float rand2 = -radius + draw(cx, key, base + 3 * attempt + 1) / (RAND_MAX_F / (radius * 2));  // :292
        float rand3 = -radius + draw(cx, key, base + 3 * attempt + 2) / (RAND_MAX_F / (radius * 2));  // :293
        offset = mk(rand1, rand2, rand3);
        if (!(length(offset) > radius)) break;  // :297
    }
    return offset;
}
// sampledNormal of mtlFunctions.cpp:162-165, 225-227, 275-277
inline V3 sampled_normal(Ctx& cx, const Hit& h, float glossiness, uint32_t key, uint32_t base) {
    V3 sampleOrigin = h.p + h.N;
    V3 sampledOffset = sample_sphere(cx, glossiness, key, base);
    return normalized((sampleOrigin + sampledOffset) - h.p);
}
inline V3 reflect_dir(V3 dir, V3 sn) {  // :207, :239, :280
    float k = 2 * dot(dir, sn);
    return normalized(dir - sn * k);
}

// MtlBlinn::Shade (mtlFunctions.cpp:120-298)
C3 shade(Ctx& cx, int mtl_id, const Ray& ray, const Hit& hInfo, int bounceCount, uint32_t key, const LightSet& ls) {
    const RtuMaterial& m = cx.s->materials[mtl_id];
    const RtuSceneDesc& s = *cx.s;
    C3 result = mkc(0, 0, 0);
    C3 diffuse = mtl_color(s, mtl_id, RTU_MAP_DIFFUSE, m.diffuse, hInfo.uvw), specular = mtl_color(s, mtl_id, RTU_MAP_SPECULAR, m.specular, hInfo.uvw);
    if (hInfo.front) {  // :125
        for (uint32_t i = 0; i < ls.n; i++) {
            const RtuLight& l = ls.lights[i];
            if (l.type == RTU_LIGHT_AMBIENT) {
                result += diffuse * illuminate(cx, l, hInfo.p, key, i);  // :132
            } else {
                V3 viewDirection = normalized(ld3(s.camera.pos) - hInfo.p);          // :137
                V3 lightDirection = normalized(-light_direction(l, hInfo.p));       // :138
                V3 halfVector = normalized(viewDirection + lightDirection);         // :139
                float NDotL = dot(hInfo.N, lightDirection);
                float NDotH = dot(hInfo.N, halfVector);
                if (NDotL < 0.0) NDotL = 0.0;
                if (NDotH < 0.0) NDotH = 0.0;
                result += (illuminate(cx, l, hInfo.p, key, i) * NDotL) * (diffuse + specular * powf(NDotH, m.glossiness));  // :152
            }
        }
    }
    if (bounceCount > 0) {
        C3 refraction = mtl_color(s, mtl_id, RTU_MAP_REFRACTION, m.refraction, hInfo.uvw);
        if (not_black(refraction)) {  // :160
            V3 sampledNormal = sampled_normal(cx, hInfo, m.refraction_glossiness, key, DRAW_REFR1);  // :162-165
            float cosTheta1 = dot(sampledNormal, -ray.dir);
            float sinTheta1 = (float)sqrt(1 - (double)cosTheta1 * (double)cosTheta1);  // :169, pow(x,2) exact
            if (sinTheta1 > 1) sinTheta1 = 1.0;
            if (sinTheta1 < -1) sinTheta1 = -1.0;
            if (cosTheta1 > 1) cosTheta1 = 1.0;
            if (cosTheta1 < -1) cosTheta1 = -1.0;
            float n1 = m.ior;
            float n2 = 1.0;
            if (hInfo.front) {
                n1 = 1.0;
                n2 = m.ior;
            }
            float sinTheta2 = (n1 / n2) * sinTheta1;
            float cosTheta2 = sqrtf(1 - sinTheta2 * sinTheta2);  // :197
            if (cosTheta2 > 1) cosTheta2 = 1.0;
            V3 SVector = normalized(cross(sampledNormal, normalized(cross(sampledNormal, -ray.dir))));  // :203
            C3 absorption = ldc(m.absorption);
            if (sinTheta2 > 1) {  // total internal reflection, :205
                Ray reflected; reflected.p = hInfo.p; reflected.dir = reflect_dir(ray.dir, sampledNormal);
                Hit rh = new_hit();
                C3 absorptionV = mkc(expf((-rh.z) * absorption.r), expf((-rh.z) * absorption.g), expf((-rh.z) * absorption.b));  // :213, z==BIGFLOAT
                cx.st.secondary_rays++;
                if (trace(cx, reflected, 0, rh)) {
                    C3 TIRResult = absorptionV * shade_node(cx, rh, reflected, bounceCount - 1, child_key(key, SLOT_MAIN), ls);
                    result += TIRResult;
                }
            } else {
                V3 sn2 = sampled_normal(cx, hInfo, m.refraction_glossiness, key, DRAW_REFR2);  // :225-227: a second sample, it shadows the first
                Ray refracted; refracted.p = hInfo.p;
                refracted.dir = normalized((-sn2) * cosTheta2 + SVector * sinTheta2);  // :229
                Hit fh = new_hit();
                cx.st.secondary_rays++;
                if (trace(cx, refracted, 0, fh)) {
                    float q = (n1 - n2) / (n1 + n2);
                    float R0 = (float)((double)q * (double)q);  // :236, pow(x,2) exact
                    float ShlicksApprox = (float)((double)R0 + (1.0 - (double)R0) * pow(1.0 - (double)cosTheta1, 5));  // :237
                    Ray reflected; reflected.p = hInfo.p; reflected.dir = reflect_dir(ray.dir, sn2);
                    Hit rh = new_hit();
                    C3 frenselResult;
                    cx.st.secondary_rays++;
                    if (trace(cx, reflected, 0, rh)) frenselResult = refraction * shade_node(cx, rh, reflected, bounceCount - 1, child_key(key, SLOT_A), ls);  // :247
                    else frenselResult = env_sample(s, reflected.dir);                                                             // :250
                    C3 refractionResult = shade_node(cx, fh, refracted, bounceCount - 1, child_key(key, SLOT_MAIN), ls);  // :254
                    C3 absorptionV = mkc(1, 1, 1);
                    if (!fh.front)
                        absorptionV = mkc(expf((-fh.z) * absorption.r), expf((-fh.z) * absorption.g), expf((-fh.z) * absorption.b));  // :259
                    result += ((absorptionV * refraction) * refractionResult) * (float)(1.0 - (double)ShlicksApprox) +
                              frenselResult * ShlicksApprox;  // :264
                } else {
                    result += env_sample(s, refracted.dir);  // :267
                }
            }
        }
        C3 reflection = mtl_color(s, mtl_id, RTU_MAP_REFLECTION, m.reflection, hInfo.uvw);
        if (not_black(reflection)) {  // :273
            V3 sampledNormal = sampled_normal(cx, hInfo, m.reflection_glossiness, key, DRAW_REFL);  // :275-277
            Ray reflected; reflected.p = hInfo.p; reflected.dir = reflect_dir(ray.dir, sampledNormal);
            Hit rh = new_hit();
            cx.st.secondary_rays++;
            if (trace(cx, reflected, 0, rh)) result += reflection * shade_node(cx, rh, reflected, bounceCount - 1, child_key(key, SLOT_C), ls);  // :286
            else result += env_sample(s, reflected.dir) * ldc(m.reflection);  // :289: reflection.GetColor(), not the sampled colour
        }
    }
    return result;
}

// ---------------------------------------------------------------------------
// Recipe P (config 5): the Monte-Carlo gather of HEAD's Render() — MonteCarlo (RenderFunctions.cpp:549-590,
// monteCarloBounces = 4, monteCarloSampleSize = 1) and SampleHemiSphereCosine (:320-337).
#define RTU_GI_BOUNCES 4
enum { SLOT_GATHER = 3, SLOT_AMBIENT_TREE = 4, DRAW_GATHER = 0x40000 };

// acos of a float in [-1, 1], evaluated in binary64 with IEEE operations only (fdlibm's e_acos rational
// approximation) and rounded to float — the device's stand-in for libm's acosf, as portable_sincos is for sinf/cosf.
inline double acos_poly(double z) {
    const double p = z * (1.66666666666666657415e-01 + z * (-3.25565818622400915405e-01 + z * (2.01212532134862925881e-01 +
                     z * (-4.00555345006794114027e-02 + z * (7.91534994289814532176e-04 + z * 3.47933107596021167570e-05)))));
    const double q = 1.0 + z * (-2.40339491173441421878e+00 + z * (2.02094576023350569471e+00 + z * (-6.88283971605453293030e-01 + z * 7.70381505559019352791e-02)));
    return p / q;
}
inline float portable_acos(float xf) {
    const double pio2_hi = 1.57079632679489655800e+00, pio2_lo = 6.12323399573676603587e-17, pi = 3.14159265358979311600e+00;
    const double x = (double)xf;
    const double ax = fabs(x);
    double r;
    if (ax >= 1.0) r = x > 0 ? 0.0 : pi;                     // |x| == 1 (|x| > 1 cannot come from 1 - 2u)
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

inline RtuLight ambient_light(C3 c) {  // AmbientLight::SetIntensity (:586-588)
    RtuLight a;
    memset(&a, 0, sizeof a);
    a.type = RTU_LIGHT_AMBIENT;
    a.intensity[0] = c.r; a.intensity[1] = c.g; a.intensity[2] = c.b;
    return a;
}

// SampleHemiSphereCosine(origin, normal, 1.0), :320-337
inline V3 sample_hemisphere_cosine(Ctx& cx, V3 normal, uint32_t key) {
    const float radius = 1.0f;
    float sampleX = draw(cx, key, DRAW_GATHER) / RAND_MAX_F;          // :324
    float samplePhi = draw(cx, key, DRAW_GATHER + 1) / THETA_DIV;     // :325
    float y = 1 - 2 * sampleX;
    float sampleTheta = (float)(0.5 * (double)(cx.libm_trig ? acosf(y) : portable_acos(y)));  // :326
    V3 v1 = normalized(cross(normal, mk(sampleX, sampleX, sampleX)));  // :329: Point3(sampleX)
    V3 v2 = normalized(cross(v1, normal));                             // :330
    float st, ct, sp, cp;
    sincos_of(cx, sampleTheta, &st, &ct);
    sincos_of(cx, samplePhi, &sp, &cp);
    return (normal * (radius * ct) + v1 * ((radius * st) * cp)) + v2 * ((radius * st) * sp);  // :332-334
}

// MonteCarlo(list, hInfo, x, y, bounces, 1): returns the intensity of the AmbientLight it appends (:549-590).
// key: the key of the Shade() calls at hInfo (its direct tree); gather draws use it too.
C3 monte_carlo(Ctx& cx, const Hit& hInfo, int bounces, uint32_t key, const LightSet& scene_lights) {
    C3 c = mkc(0, 0, 0);
    if (bounces > 0) {
        Hit h = new_hit();
        V3 sampleOffset = sample_hemisphere_cosine(cx, hInfo.N, key);
        Ray sampleRay; sampleRay.p = hInfo.p; sampleRay.dir = normalized(sampleOffset);  // :562
        cx.gather_rays++;  // not among the ray counters: the reference harness cannot count these Trace calls either (same translation unit)
        if (trace(cx, sampleRay, 0, h)) {  // :565
            const uint32_t hkey = child_key(key, SLOT_GATHER);
            C3 indirect = monte_carlo(cx, h, bounces - 1, hkey, scene_lights);  // :568
            RtuLight amb = ambient_light(indirect);
            LightSet mc = {&amb, 1};
            c += shade_node(cx, h, sampleRay, RTU_MAX_BOUNCE, child_key(hkey, SLOT_AMBIENT_TREE), mc);  // :569
            c += shade_node(cx, h, sampleRay, RTU_MAX_BOUNCE, hkey, scene_lights);                        // :570
        } else {
            c += env_sample(*cx.s, sampleRay.dir);  // :575
        }
        // :581: c /= (float)monteCarloSampleSize, a division by 1
    } else {
        c = mkc(0.1f, 0.1f, 0.1f);  // :584
    }
    return c;
}

// CalculateImageOrigin + CalculateCurrentPoint (RenderFunctions.cpp:243-269),
// hoisted out of the per-pixel loop: origin, u, v.
struct CamFrame {
    V3 pos, origin, u, v;
};
CamFrame camera_frame(const RtuCamera& c, int W, int H) {
    V3 pos = ld3(c.pos), dir = ld3(c.dir), up = ld3(c.up);
    float distanceToImg = c.focaldist;
    float actualHeight = (float)(tan((c.fov / 2) * M_PI / 180.0) * 2 * distanceToImg);  // :247 (fp64 chain)
    float actualWidth = ((float)W / (float)H) * actualHeight;                            // :248
    V3 topCenterPoint = (pos + normalized(dir) * distanceToImg) + normalized(up) * (actualHeight / 2);  // :250
    V3 right = normalized(cross(normalized(dir), normalized(up)));
    CamFrame f;
    f.pos = pos;
    f.origin = topCenterPoint - right * (actualWidth / 2);  // :252
    f.u = right * (actualWidth / (float)W);                 // :263
    f.v = (normalized(up) * -1.0f) * (actualHeight / (float)H);  // :264: (-1*(up_n)) * (h/H)
    return f;
}

void add_stats(RtuOracleStats& a, const RtuOracleStats& b) {
    a.primary_rays += b.primary_rays; a.primary_hits += b.primary_hits;
    a.secondary_rays += b.secondary_rays; a.shadow_rays += b.shadow_rays;
    a.node_tests += b.node_tests; a.mesh_entries += b.mesh_entries;
    a.inner_visits += b.inner_visits; a.leaf_visits += b.leaf_visits; a.leaf_elems += b.leaf_elems;
    a.tri_tests += b.tri_tests; a.tri_accepts += b.tri_accepts;
}

// Halton (scene.h:130-139)
inline float halton(int index, int base) {
    float r = 0;
    float f = 1.0f / (float)base;
    for (int i = index; i > 0; i /= base) {
        r += f * (i % base);
        f /= (float)base;
    }
    return r;
}

struct Sampling {
    int spp;          // 0: recipe W (one ray through the pixel centre); S >= 1: recipe S, the sample loop of Render()
    bool sequential, libm_trig;
    bool gi;          // recipe P: recipe S plus the Monte-Carlo gather of Render() (:129-134)
    bool per_pixel;   // work distribution: false = chunks of rows; true = the reference's PixelIterator (PixelIterator.h:25-38):
                      // one shared atomic counter, ONE PIXEL per fetch, x = i % W, y = i / W
};

void render_rows(const RtuSceneDesc* s, const CamFrame& cf, int W, int H, std::atomic<int>* next_row, int y_begin,
                 int y_end, int chunk, float* rgbz, RtuOracleStats* out, Sampling sm) {
    Ctx cx;
    cx.s = s;
    cx.sequential = sm.sequential;
    cx.libm_trig = sm.libm_trig;
    cx.seq_key = cx.seq_counter = 0;
    cx.gather_rays = 0;
    memset(&cx.st, 0, sizeof cx.st);
    const V3 up = ld3(s->camera.up);
    const V3 right = normalized(cross(normalized(ld3(s->camera.dir)), normalized(up)));
    const LightSet scene_lights = {s->lights, s->n_lights};
    for (;;) {
        int y0, y1, x0 = 0, x1 = W;
        if (sm.per_pixel) {  // next_row counts pixels from y_begin * W
            const int i = next_row->fetch_add(1);
            if (i >= y_end * W) break;
            y0 = i / W; y1 = y0 + 1;
            x0 = i % W; x1 = x0 + 1;
        } else {
            y0 = next_row->fetch_add(chunk);
            if (y0 >= y_end) break;
            y1 = y0 + chunk < y_end ? y0 + chunk : y_end;
        }
        for (int y = y0; y < y1; y++) {
            for (int x = x0; x < x1; x++) {
                float* o = rgbz + 4 * ((size_t)(y - y_begin) * W + x);
                if (sm.spp == 0) {
                    // recipe W: CalculateCurrentPoint(x,y,0.5f,0.5f,org), RenderFunctions.cpp:258-268
                    V3 cp = (cf.origin + cf.u * ((float)x + 0.5f)) + cf.v * ((float)y + 0.5f);
                    Ray ray; ray.p = cf.pos; ray.dir = normalized(cp - cf.pos);  // :97
                    Hit h = new_hit();
                    cx.st.primary_rays++;
                    bool hit = trace(cx, ray, 0, h);  // :103
                    C3 c;
                    if (hit) {
                        cx.st.primary_hits++;
                        c = shade_node(cx, h, ray, RTU_MAX_BOUNCE, 0, scene_lights);  // :134-135
                    } else {
                        c = background_sample(*s, x, y);  // RenderFunctions.cpp:145
                    }
                    o[0] = c.r; o[1] = c.g; o[2] = c.b; o[3] = h.z;
                    continue;
                }
                // recipe S: the sample loop of Render() (RenderFunctions.cpp:73-151) with spp in place of
                // maxSampleSize, direct lighting only, every sample traced and shaded in turn.
                const float pixelIncrement = (float)(1.0 / sm.spp);  // :68
                C3 pixelValuesSum = mkc(0, 0, 0);
                float zSum = 0.0f;
                int numOfHits = 0;
                for (int index = 0; index < sm.spp; index++) {
                    const uint32_t key = sample_key((uint32_t)(x + W * y), (uint32_t)index);
                    cx.seq_key = key;
                    cx.seq_counter = 0;
                    float currentOffset = index * pixelIncrement;  // :80
                    float offsetX = halton(index, 4);              // :84
                    float offsetY = halton(index, 5);              // :85
                    float sampleX = draw(cx, key, DRAW_DOF) / RAND_MAX_F;          // :88
                    float sampleTheta = draw(cx, key, DRAW_DOF + 1) / THETA_DIV;   // :89
                    float sn, cs;
                    sincos_of(cx, sampleTheta, &sn, &cs);
                    float rad = sqrtf((sampleX * s->camera.dof) * s->camera.dof);
                    float camOffsetX = rad * cs;  // :90
                    float camOffsetY = rad * sn;  // :91
                    V3 sampledPosition = (cf.pos + up * camOffsetY) + right * camOffsetX;  // :93
                    V3 cp = (cf.origin + cf.u * ((float)x + (currentOffset + offsetX))) + cf.v * ((float)y + (currentOffset + offsetY));  // :96
                    Ray ray; ray.p = sampledPosition; ray.dir = normalized(cp - sampledPosition);  // :97
                    Hit h = new_hit();
                    cx.st.primary_rays++;
                    C3 c;
                    if (trace(cx, ray, 0, h)) {  // :103
                        cx.st.primary_hits++;
                        zSum += h.z;  // :109
                        numOfHits++;
                        if (sm.gi && s->nodes[h.node].material_id >= 0) {  // recipe P: :129-135
                            C3 indirect = monte_carlo(cx, h, RTU_GI_BOUNCES, key, scene_lights);
                            RtuLight amb = ambient_light(indirect);
                            LightSet mc = {&amb, 1};
                            c = shade_node(cx, h, ray, RTU_MAX_BOUNCE, child_key(key, SLOT_AMBIENT_TREE), mc);  // :134
                            c += shade_node(cx, h, ray, RTU_MAX_BOUNCE, key, scene_lights);                     // :135
                        } else {
                            c = shade_node(cx, h, ray, RTU_MAX_BOUNCE, key, scene_lights);  // :135
                        }
                    } else {
                        c = background_sample(*s, x, y);  // :145
                    }
                    pixelValuesSum += c;  // :148
                }
                // :152 (Color /= float divides) and the z of the commented-out :115
                o[0] = pixelValuesSum.r / (float)sm.spp; o[1] = pixelValuesSum.g / (float)sm.spp; o[2] = pixelValuesSum.b / (float)sm.spp;
                o[3] = numOfHits ? zSum / (float)numOfHits : RTU_BIGFLOAT;
            }
        }
    }
    (void)H;
    *out = cx.st;
}

int check_scene(const RtuSceneDesc* s, bool stochastic_ok) {
    if (!s || !s->nodes || s->n_nodes == 0) return RTU_ORACLE_ERR_ARG;
    if (s->camera.dof != 0 && !stochastic_ok) return RTU_ORACLE_ERR_STOCHASTIC;
    for (uint32_t i = 0; i < s->n_textures; i++)
        if (s->textures[i].type == RTU_TEX_FILE && s->textures[i].width * s->textures[i].height > 0 && !s->textures[i].rgb) return RTU_ORACLE_ERR_ARG;
    for (uint32_t i = 0; i < s->n_lights; i++)
        if (s->lights[i].type == RTU_LIGHT_POINT && s->lights[i].size > 0 && !stochastic_ok) return RTU_ORACLE_ERR_STOCHASTIC;
    for (uint32_t i = 0; i < s->n_materials; i++)
        if ((s->materials[i].reflection_glossiness > 0 || s->materials[i].refraction_glossiness > 0) && !stochastic_ok) return RTU_ORACLE_ERR_STOCHASTIC;
    for (uint32_t i = 0; i < s->n_nodes; i++) {
        const RtuNode& n = s->nodes[i];
        if (n.obj_type == RTU_OBJ_TRIMESH && (n.mesh_id < 0 || (uint32_t)n.mesh_id >= s->n_meshes)) return RTU_ORACLE_ERR_ARG;
        if (n.material_id >= (int)s->n_materials) return RTU_ORACLE_ERR_ARG;
    }
    return 0;
}

}  // namespace

extern "C" {

static int render_impl(const RtuSceneDesc* scene, int width, int height, int row0, int nrows, float* rgbz_out,
                       RtuOracleStats* stats, int threads, Sampling sm) {
    int err = check_scene(scene, sm.spp >= 1);
    if (err) return err;
    if (width <= 0 || height <= 0 || row0 < 0 || nrows < 0 || row0 + nrows > height || !rgbz_out || sm.spp < 0) return RTU_ORACLE_ERR_ARG;
    if (threads < 1) threads = 1;
    CamFrame cf = camera_frame(scene->camera, width, height);
    std::atomic<int> next(sm.per_pixel ? row0 * width : row0);
    std::vector<RtuOracleStats> st(threads);
    // rows per fetch: four, fewer when there are so many threads that four-row chunks would leave each of them one or two
    // (256 threads on 1080 rows: single rows, about four per thread, so that the expensive rows spread out)
    int chunk = nrows / (threads * 4);
    chunk = chunk < 1 ? 1 : chunk > 4 ? 4 : chunk;
    if (threads == 1 && !sm.per_pixel) {
        render_rows(scene, cf, width, height, &next, row0, row0 + nrows, nrows > 0 ? nrows : 1, rgbz_out, &st[0], sm);
    } else {
        std::vector<std::thread> th;
        for (int t = 0; t < threads; t++)
            th.emplace_back(render_rows, scene, cf, width, height, &next, row0, row0 + nrows, chunk, rgbz_out, &st[t], sm);
        for (auto& t : th) t.join();
    }
    if (stats) {
        memset(stats, 0, sizeof *stats);
        for (auto& s : st) add_stats(*stats, s);
    }
    return 0;
}

int rtu_oracle_render_rows(const RtuSceneDesc* scene, int width, int height, int row0, int nrows, float* rgbz_out,
                           RtuOracleStats* stats, int threads) {
    Sampling sm = {0, false, false, false, false};
    return render_impl(scene, width, height, row0, nrows, rgbz_out, stats, threads, sm);
}

int rtu_oracle_render_samples(const RtuSceneDesc* scene, int width, int height, int row0, int nrows, int spp, int stream,
                              int trig, float* rgbz_out, RtuOracleStats* stats, int threads) {
    if (spp < 1 || (stream != RTU_ORACLE_STREAM_KEYED && stream != RTU_ORACLE_STREAM_SEQUENTIAL) ||
        (trig != RTU_ORACLE_TRIG_PORTABLE && trig != RTU_ORACLE_TRIG_LIBM))
        return RTU_ORACLE_ERR_ARG;
    Sampling sm = {spp, stream == RTU_ORACLE_STREAM_SEQUENTIAL, trig == RTU_ORACLE_TRIG_LIBM, false, false};
    return render_impl(scene, width, height, row0, nrows, rgbz_out, stats, threads, sm);
}

int rtu_oracle_render_paths(const RtuSceneDesc* scene, int width, int height, int row0, int nrows, int spp, int stream,
                            int trig, float* rgbz_out, RtuOracleStats* stats, int threads) {
    if (spp < 1 || (stream != RTU_ORACLE_STREAM_KEYED && stream != RTU_ORACLE_STREAM_SEQUENTIAL) ||
        (trig != RTU_ORACLE_TRIG_PORTABLE && trig != RTU_ORACLE_TRIG_LIBM))
        return RTU_ORACLE_ERR_ARG;
    Sampling sm = {spp, stream == RTU_ORACLE_STREAM_SEQUENTIAL, trig == RTU_ORACLE_TRIG_LIBM, true, false};
    return render_impl(scene, width, height, row0, nrows, rgbz_out, stats, threads, sm);
}

void rtu_oracle_portable_acos(const float* x, int n, float* out) {
    for (int i = 0; i < n; i++) out[i] = portable_acos(x[i]);
}

// sin, cos of the portable evaluation, for the test that bounds it against libm.
void rtu_oracle_portable_sincos(const float* t, int n, float* sin_out, float* cos_out) {
    for (int i = 0; i < n; i++) portable_sincos(t[i], sin_out + i, cos_out + i);
}

// The integers of the sample streams, for the tests that pin the device's generator.
uint32_t rtu_oracle_rand31(uint32_t key, uint32_t idx) { return rand31(key, idx); }
uint32_t rtu_oracle_sample_key(uint32_t pixel, uint32_t sample) { return sample_key(pixel, sample); }
uint32_t rtu_oracle_child_key(uint32_t key, uint32_t slot) { return child_key(key, slot); }

void rtu_oracle_debug_all_triangles(int on) { g_all_triangles = on != 0; }

int rtu_oracle_render_scheduled(const RtuSceneDesc* scene, int width, int height, float* rgbz_out, RtuOracleStats* stats, int threads,
                                int per_pixel_schedule) {
    Sampling sm = {0, false, false, false, per_pixel_schedule != 0};
    return render_impl(scene, width, height, 0, height, rgbz_out, stats, threads, sm);
}

int rtu_oracle_render(const RtuSceneDesc* scene, int width, int height, float* rgbz_out, RtuOracleStats* stats, int threads) {
    return rtu_oracle_render_rows(scene, width, height, 0, height, rgbz_out, stats, threads);
}

// Camera set-up alone, for checking the product's rtu_frame_setup (a3).
int rtu_oracle_camera_frame(const RtuCamera* cam, int width, int height, float out12[12]) {
    if (!cam || !out12 || width <= 0 || height <= 0) return RTU_ORACLE_ERR_ARG;
    CamFrame f = camera_frame(*cam, width, height);
    const V3* v[4] = {&f.pos, &f.origin, &f.u, &f.v};
    for (int i = 0; i < 4; i++) { out12[3 * i] = v[i]->x; out12[3 * i + 1] = v[i]->y; out12[3 * i + 2] = v[i]->z; }
    return 0;
}

// Gamma + Color24 (RenderFunctions.cpp:155-159, cyColor.h:226,245-246) and
// RenderImage::ComputeZBufferImage (scene.h:590-612).
static unsigned char float_to_byte(float r) {
    float v = r * 255;
    int i;
    // x86 cvttss2si semantics of int(float) for NaN / out-of-range: INT_MIN
    if (!(v > -2147483904.0f && v < 2147483648.0f)) i = (int)0x80000000;
    else i = (int)v;
    return (unsigned char)(i < 0 ? 0 : (i > 255 ? 255 : i));
}

void rtu_oracle_postprocess(const float* rgbz, int width, int height, unsigned char* rgb_out, float* z_out,
                            unsigned char* zimg_out) {
    size_t n = (size_t)width * height;
    for (size_t i = 0; i < n; i++) {
        const float* p = rgbz + 4 * i;
        if (rgb_out) {
            for (int k = 0; k < 3; k++) {
                float g = (float)pow((double)p[k], 1 / 2.2);
                rgb_out[3 * i + k] = float_to_byte(g);
            }
        }
        if (z_out) z_out[i] = p[3];
    }
    if (zimg_out) {
        float zmin = RTU_BIGFLOAT, zmax = 0;
        for (size_t i = 0; i < n; i++) {
            float z = rgbz[4 * i + 3];
            if (z == RTU_BIGFLOAT) continue;
            if (zmin > z) zmin = z;
            if (zmax < z) zmax = z;
        }
        for (size_t i = 0; i < n; i++) {
            float z = rgbz[4 * i + 3];
            if (z == RTU_BIGFLOAT) zimg_out[i] = 0;
            else {
                float f = (zmax - z) / (zmax - zmin);
                float v = f * 255;
                int c;
                if (!(v > -2147483904.0f && v < 2147483648.0f)) c = (int)0x80000000;
                else c = (int)v;
                if (c < 0) c = 0;
                if (c > 255) c = 255;
                zimg_out[i] = (unsigned char)c;
            }
        }
    }
}

}  // extern "C"
