// rtu_vec.h — float3 / colour arithmetic of the render path, host + device.
//
// Every operation keeps the reference's evaluation order and rounding points
// (cyPoint.h:259-349, cyMatrix.h:543-547, cyColor.h:95-112); the translation
// units that include this header are compiled with -ffp-contract=off and IEEE
// divide/sqrt, so host and device produce the same bits.
#ifndef RTU_VEC_H_INCLUDED
#define RTU_VEC_H_INCLUDED

#include <hip/hip_runtime.h>
#include <math.h>

#define RTU_HD __host__ __device__ __forceinline__

struct f3 {
    float x, y, z;
};

RTU_HD f3 mk3(float x, float y, float z) { f3 r; r.x = x; r.y = y; r.z = z; return r; }
template <class P> RTU_HD f3 ld3(P p) { return mk3(p[0], p[1], p[2]); }  // any address space
RTU_HD f3 operator+(f3 a, f3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
RTU_HD f3 operator-(f3 a, f3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
RTU_HD f3 operator-(f3 a) { return mk3(-a.x, -a.y, -a.z); }
RTU_HD f3 operator*(f3 a, float s) { return mk3(a.x * s, a.y * s, a.z * s); }
RTU_HD f3 operator*(f3 a, f3 b) { return mk3(a.x * b.x, a.y * b.y, a.z * b.z); }  // Color * Color
RTU_HD f3 operator/(f3 a, float s) { return mk3(a.x / s, a.y / s, a.z / s); }
// Point3::Dot = (x*x' + y*y') + z*z'  (cyPoint.h:296,348)
RTU_HD float dot3(f3 a, f3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
RTU_HD f3 cross3(f3 a, f3 b) {  // cyPoint.h:346
    return mk3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
RTU_HD float len3(f3 a) { return sqrtf(dot3(a, a)); }      // cyPoint.h:293 -> sqrtf (cyCore.h:161)
RTU_HD f3 norm3(f3 a) { return a / len3(a); }              // cyPoint.h:295: three IEEE divides
RTU_HD bool not_black(f3 c) { return c.x != 0 || c.y != 0 || c.z != 0; }  // Color::operator!=, cyColor.h:115

// Matrix3 * Point3, column-major (cyMatrix.h:543-547)
template <class P> RTU_HD f3 mat_mul(P m, f3 p) {
    return mk3((p.x * m[0] + p.y * m[3]) + p.z * m[6],
               (p.x * m[1] + p.y * m[4]) + p.z * m[7],
               (p.x * m[2] + p.y * m[5]) + p.z * m[8]);
}
// Transformation::TransposeMult (scene.h:253-260)
template <class P> RTU_HD f3 mat_tmul(P m, f3 d) {
    return mk3(dot3(mk3(m[0], m[1], m[2]), d), dot3(mk3(m[3], m[4], m[5]), d), dot3(mk3(m[6], m[7], m[8]), d));
}

// std::max / std::min as libstdc++ defines them; NaN operands must behave the
// same as in objFunctions.cpp:164-245.
RTU_HD float smax(float a, float b) { return (a < b) ? b : a; }
RTU_HD float smin(float a, float b) { return (b < a) ? b : a; }

#endif
