// render_kernel.hip — dispatch of a launch sequence to the feature set's translation unit
// (render_feat*.hip, all instantiating render_impl.h) plus the small kernels that are not part
// of a launch sequence: recipe S / P accumulation, the RenderImage packer, the self-tests.
#include "rtu_intersect.h"

namespace {
// Self-test of the exact-division identity used by the slab and barycentric tests
// (rtu_intersect.h, fdiv): pseudo-random bit patterns (all exponents, subnormals, zeros,
// infinities, NaNs) plus same-exponent pairs; counts quotients whose bits differ from `/`.
// ------------------------------------------------------------------------------------
__global__ void k_selftest_fdiv(unsigned long long n_pairs, unsigned long long seed, unsigned long long* mismatches) {
    unsigned long long bad = 0;
    for (unsigned long long i = blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x; i < n_pairs;
         i += (unsigned long long)gridDim.x * blockDim.x) {
        unsigned long long x = (i + seed) * 0x9E3779B97F4A7C15ull;
        x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 27; x *= 0x94D049BB133111EBull; x ^= x >> 31;
        unsigned int nb = (unsigned int)x, db = (unsigned int)(x >> 32);
        unsigned int mode = (unsigned int)(i & 7u);
        if (mode == 1) db = (db & 0x007FFFFFu) | (nb & 0xFF800000u);          // same exponent: quotient near 1
        if (mode == 2) nb = (nb & 0x807FFFFFu) | 0x3F800000u;                   // n in [1,2)
        if (mode == 3) { nb &= 0x80FFFFFFu; db = (db & 0x80FFFFFFu) | 0x7E000000u; }  // tiny / huge -> subnormal quotients
        if (mode == 4) db &= 0x807FFFFFu;                                       // subnormal / zero divisor
        float n = __uint_as_float(nb), d = __uint_as_float(db);
        float q1 = n / d;
        float q2 = fdiv(n, 1.0 / (double)d);
        bool same = __float_as_uint(q1) == __float_as_uint(q2) || (q1 != q1 && q2 != q2);
        if (!same) bad++;
    }
    if (bad) atomicAdd(mismatches, bad);
}

// Self-test of the re-ordered sphere / plane tests (rtu_intersect.h): the fast form against the
// literal reference order, bit for bit (return value, z, front, p, N), on rays built to stress
// the places where they could differ — grazing the sphere / the square's edge, far away (cancelling
// discriminant), axis-parallel, starting on the surface, with every incoming h.z regime.
__global__ void k_selftest_prims(unsigned long long n_rays, unsigned long long seed, unsigned long long* mismatches) {
    unsigned long long bad = 0;
    for (unsigned long long i = blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x; i < n_rays;
         i += (unsigned long long)gridDim.x * blockDim.x) {
        unsigned long long x = (i + seed) * 0x9E3779B97F4A7C15ull;
        auto rnd = [&]() {  // uniform in [0,1)
            x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 27; x *= 0x94D049BB133111EBull; x ^= x >> 31;
            return (float)(x >> 40) * (1.0f / 16777216.0f);
        };
        const unsigned mode = (unsigned)(i % 12u);
        const float scale = mode < 4 ? 3.0f : mode < 6 ? 40.0f : mode < 8 ? 3000.0f : 2.0f;
        Ray r;
        r.p = mk3((rnd() * 2 - 1) * scale, (rnd() * 2 - 1) * scale, (rnd() * 2 - 1) * scale);
        // aim at a point on / near the unit sphere or the unit square, then perturb
        f3 tgt = mk3(rnd() * 2 - 1, rnd() * 2 - 1, rnd() * 2 - 1);
        const bool plane = (i & 1u) != 0;
        if (plane) {
            tgt.z = 0;
            if (mode % 3u == 0) tgt.x = (rnd() < 0.5f ? -1.0f : 1.0f) * (1.0f + (rnd() * 2 - 1) * 1e-5f);  // the square's edge
            if (mode % 3u == 1) tgt.y = (rnd() < 0.5f ? -1.0f : 1.0f) * (1.0f + (rnd() * 2 - 1) * 1e-6f);
        } else {
            tgt = norm3(tgt);
            if (mode % 3u == 0) {  // a ray tangent to the sphere at tgt (+- a hair)
                f3 tang = norm3(cross3(tgt, mk3(rnd() + 0.1f, rnd() - 0.5f, rnd() - 0.5f)));
                r.p = tgt * (1.0f + (rnd() * 2 - 1) * 1e-5f) - tang * (rnd() * scale);
            }
            if (mode == 9) r.p = tgt;                       // origin on the surface
            if (mode == 10) r.p = tgt * (rnd() * 0.999f);   // origin inside
        }
        r.dir = tgt - r.p;
        if (mode == 11) { r.dir.x = 0; if (rnd() < 0.5f) r.dir.y = 0; }  // axis-parallel
        if (rnd() < 0.5f) r.dir = norm3(r.dir);
        if (rnd() < 0.1f) r.dir = r.dir * (rnd() * 50.0f);
        const float zsel = rnd();
        Hit h0;
        fresh_hit(h0, zsel < 0.4f ? RTU_BIGFLOAT : zsel < 0.7f ? rnd() * 2.0f * scale : rnd() * 0.01f);
        h0.front = rnd() < 0.5f;
        Hit a = h0, b = h0;
        bool ra, rb;
        if (plane) { ra = plane_hit_t<true>(r, a); rb = plane_hit_t<false>(r, b); }
        else { ra = sphere_hit_t<true>(r, a); rb = sphere_hit_t<false>(r, b); }
        auto same = [](float u, float v) { return __float_as_uint(u) == __float_as_uint(v) || (u != u && v != v); };
        bool ok = ra == rb && same(a.z, b.z) && a.front == b.front;
        if (ra && rb) ok = ok && same(a.p.x, b.p.x) && same(a.p.y, b.p.y) && same(a.p.z, b.p.z) && same(a.N.x, b.N.x) && same(a.N.y, b.N.y) && same(a.N.z, b.N.z);
        if (!ok) bad++;
    }
    if (bad) atomicAdd(mismatches, bad);
}


// ---- recipe P ----
// the pixel of a chain: Shade(h_0, AmbientLight) + Shade(h_0, lights) (:134-135), z of the primary hit
__global__ void __launch_bounds__(256) k_gi_final(KernelArgs a) {
    const uint32_t chain = blockIdx.x * 256u + threadIdx.x;
    if (chain >= a.gi_total) return;
    const uint32_t pk = __float_as_uint(a.gi_h[(size_t)a.gi_total + chain].w);
    if (!(pk & 1u)) return;  // missed (background) or a node without material (white): written by the chain's first launch
    const float4 ra = a.gi_res[chain], rd = a.gi_res[(size_t)a.gi_total + chain];
    a.out[chain] = make_float4(ra.x + rd.x, ra.y + rd.y, ra.z + rd.z, a.gi_h[chain].w);
}

}  // namespace

// ---- recipe S: the sums of RenderFunctions.cpp:109-110,148 in sample order, and :152 -----------
namespace {
__global__ void __launch_bounds__(256) k_accumulate(const float4* samples, uint32_t batch, float4* acc, uint32_t* hits, uint32_t pixels, int first) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= pixels) return;
    float4 s = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    uint32_t n = 0;
    if (!first) { s = acc[i]; n = hits[i]; }
    for (uint32_t b = 0; b < batch; b++) {  // the samples of the batch in their order
        const float4 v = samples[(size_t)b * pixels + i];
        s.x += v.x; s.y += v.y; s.z += v.z;                // pixelValuesSum += currentResult
        if (v.w != RTU_BIGFLOAT) { s.w += v.w; n++; }      // zSum += z; numOfHits++
    }
    acc[i] = s;
    hits[i] = n;
}
__global__ void __launch_bounds__(256) k_resolve(const float4* acc, const uint32_t* hits, float4* out, uint32_t pixels, uint32_t samples) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= pixels) return;
    const float4 s = acc[i];
    const uint32_t n = hits[i];
    const float S = (float)samples;
    out[i] = make_float4(s.x / S, s.y / S, s.z / S, n ? s.w / (float)n : RTU_BIGFLOAT);
}
}  // namespace

// ---- RenderImage content on the device: Color24 pixels + float z (RenderFunctions.cpp:152-160, cyColor.h:226,245) ----
namespace {
__device__ __forceinline__ uint8_t float_to_byte(float r) {  // Color24(Color): Clamp(int(c * 255)), cvttss2si semantics
    const float v = r * 255;
    int i;
    if (!(v > -2147483904.0f && v < 2147483648.0f)) i = (int)0x80000000;
    else i = (int)v;
    return (uint8_t)(i < 0 ? 0 : (i > 255 ? 255 : i));
}
__global__ void __launch_bounds__(256) k_pack_image(const float4* rgbz, unsigned long long pixels, float* z_out, uint8_t* rgb_out) {
    const unsigned long long i = (unsigned long long)blockIdx.x * 256u + threadIdx.x;
    if (i >= pixels) return;
    const float4 v = rgbz[i];
    z_out[i] = v.w;
    rgb_out[3 * i + 0] = float_to_byte((float)pow((double)v.x, 1 / 2.2));  // pow(c, 1/2.2) in binary64, as the reference
    rgb_out[3 * i + 1] = float_to_byte((float)pow((double)v.y, 1 / 2.2));
    rgb_out[3 * i + 2] = float_to_byte((float)pow((double)v.z, 1 / 2.2));
}
}  // namespace

// ---- the two OUTPUT images of a frame, 4 bytes per pixel: Color24 + the z-image byte of ComputeZBufferImage (scene.h:590-612) ----
// The z-image needs the frame-wide zmin / zmax over the pixels that hit (z != BIGFLOAT). k_minmax_z reduces them per frame of a
// batch into order-preserving integer keys (min as key, max as ~key, so that ONE element-wise MIN over the shards of all GPUs
// combines both); k_pack_output then quantises with the reference's float expression (correctly rounded division, truncation).
namespace {
__device__ __forceinline__ uint32_t z_key(float z) {  // monotone: a < b  <=>  z_key(a) < z_key(b), all finite floats
    const uint32_t b = __float_as_uint(z);
    return (b >> 31) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float z_unkey(uint32_t k) { return __uint_as_float((k >> 31) ? (k & 0x7FFFFFFFu) : ~k); }

__global__ void __launch_bounds__(256) k_minmax_z(const float4* rgbz, uint32_t pixels_per_frame, long long* minmax) {
    const uint32_t frame = blockIdx.y;
    uint32_t kmin = 0xFFFFFFFFu, kmax = 0u;  // nothing hit yet: zmin = BIGFLOAT, zmax = 0 are applied when the keys are read
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < pixels_per_frame; i += gridDim.x * 256u) {
        const float z = rgbz[(size_t)frame * pixels_per_frame + i].w;
        if (z == RTU_BIGFLOAT || z != z) continue;  // a miss; a NaN never wins a comparison in the reference's loop either
        const uint32_t k = z_key(z);
        kmin = k < kmin ? k : kmin;
        kmax = k > kmax ? k : kmax;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const uint32_t a = (uint32_t)__shfl_xor((int)kmin, off), b = (uint32_t)__shfl_xor((int)kmax, off);
        kmin = a < kmin ? a : kmin;
        kmax = b > kmax ? b : kmax;
    }
    if ((threadIdx.x & 63u) == 0) {
        if (kmin != 0xFFFFFFFFu) atomicMin((unsigned long long*)&minmax[2 * frame], (unsigned long long)kmin);
        if (kmax != 0u) atomicMin((unsigned long long*)&minmax[2 * frame + 1], (unsigned long long)(~kmax));
    }
}

__global__ void __launch_bounds__(256) k_pack_output(const float4* rgbz, uint32_t pixels_per_frame, const long long* minmax, uchar4* out) {
    const uint32_t frame = blockIdx.y;
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= pixels_per_frame) return;
    // scene.h:596-601: float zmin = BIGFLOAT, zmax = 0, then min / max over the pixels that hit
    const unsigned long long kmin = (unsigned long long)minmax[2 * frame], nkmax = (unsigned long long)minmax[2 * frame + 1];
    float zmin = kmin > 0xFFFFFFFFull ? RTU_BIGFLOAT : z_unkey((uint32_t)kmin);
    float zmax = nkmax > 0xFFFFFFFFull ? 0.0f : z_unkey(~(uint32_t)nkmax);
    if (!(RTU_BIGFLOAT > zmin)) zmin = RTU_BIGFLOAT;  // `if (zmin > z) zmin = z` starting from BIGFLOAT: never above it
    if (!(0.0f < zmax)) zmax = 0.0f;                  // `if (zmax < z) zmax = z` starting from 0: never below it
    const float4 v = rgbz[(size_t)frame * pixels_per_frame + i];
    uchar4 o;
    o.x = float_to_byte((float)pow((double)v.x, 1 / 2.2));
    o.y = float_to_byte((float)pow((double)v.y, 1 / 2.2));
    o.z = float_to_byte((float)pow((double)v.z, 1 / 2.2));
    o.w = v.w == RTU_BIGFLOAT ? (uint8_t)0 : float_to_byte((zmax - v.w) / (zmax - zmin));  // :603-609 (float_to_byte = int(f * 255), clamped)
    out[(size_t)frame * pixels_per_frame + i] = o;
}
}  // namespace

int rtu_launch_minmax_z(const float4* rgbz, uint32_t pixels_per_frame, uint32_t frames, long long* minmax, hipStream_t stream) {
    if (pixels_per_frame == 0 || frames == 0) return (int)hipSuccess;
    hipError_t e = hipMemsetAsync(minmax, 0x7F, sizeof(long long) * 2 * frames, stream);  // "nothing yet": any key is below 0x7F7F...
    if (e != hipSuccess) return (int)e;
    uint32_t bx = (pixels_per_frame + 255u) / 256u;
    if (bx > 1024u) bx = 1024u;
    hipLaunchKernelGGL(k_minmax_z, dim3(bx, frames), dim3(256), 0, stream, rgbz, pixels_per_frame, minmax);
    return (int)hipGetLastError();
}

int rtu_launch_pack_output(const float4* rgbz, uint32_t pixels_per_frame, uint32_t frames, const long long* minmax, unsigned char* out, hipStream_t stream) {
    if (pixels_per_frame == 0 || frames == 0) return (int)hipSuccess;
    hipLaunchKernelGGL(k_pack_output, dim3((pixels_per_frame + 255u) / 256u, frames), dim3(256), 0, stream, rgbz, pixels_per_frame, minmax, (uchar4*)out);
    return (int)hipGetLastError();
}

int rtu_launch_pack_image(const float4* rgbz, unsigned long long pixels, float* z_out, unsigned char* rgb_out, hipStream_t stream) {
    if (pixels == 0) return (int)hipSuccess;
    hipLaunchKernelGGL(k_pack_image, dim3((unsigned)((pixels + 255u) / 256u)), dim3(256), 0, stream, rgbz, pixels, z_out, (uint8_t*)rgb_out);
    return (int)hipGetLastError();
}

int rtu_launch_accumulate(const float4* samples, uint32_t batch, float4* acc, uint32_t* hits, uint32_t pixels, bool first, hipStream_t stream) {
    hipLaunchKernelGGL(k_accumulate, dim3((pixels + 255u) / 256u), dim3(256), 0, stream, samples, batch, acc, hits, pixels, first ? 1 : 0);
    return (int)hipGetLastError();
}
int rtu_launch_resolve(const float4* acc, const uint32_t* hits, float4* out, uint32_t pixels, uint32_t samples, hipStream_t stream) {
    hipLaunchKernelGGL(k_resolve, dim3((pixels + 255u) / 256u), dim3(256), 0, stream, acc, hits, out, pixels, samples);
    return (int)hipGetLastError();
}

int rtu_launch_selftest_fdiv(unsigned long long n_pairs, unsigned long long seed, unsigned long long* d_mismatches, hipStream_t stream) {
    hipLaunchKernelGGL(k_selftest_fdiv, dim3(4096), dim3(256), 0, stream, n_pairs, seed, d_mismatches);
    return (int)hipGetLastError();
}

int rtu_launch_selftest_prims(unsigned long long n_rays, unsigned long long seed, unsigned long long* d_mismatches, hipStream_t stream) {
    hipLaunchKernelGGL(k_selftest_prims, dim3(4096), dim3(256), 0, stream, n_rays, seed, d_mismatches);
    return (int)hipGetLastError();
}


int rtu_launch_feat0(const KernelArgs& args, uint32_t n_tiles, uint32_t bvh_stack_needed, bool stats, hipStream_t stream, int mode, const LaunchProbe* probe);
int rtu_launch_feat1(const KernelArgs& args, uint32_t n_tiles, uint32_t bvh_stack_needed, bool stats, hipStream_t stream, int mode, const LaunchProbe* probe);
int rtu_launch_feat2(const KernelArgs& args, uint32_t n_tiles, uint32_t bvh_stack_needed, bool stats, hipStream_t stream, int mode, const LaunchProbe* probe);
int rtu_launch_feat3(const KernelArgs& args, uint32_t n_tiles, uint32_t bvh_stack_needed, bool stats, hipStream_t stream, int mode, const LaunchProbe* probe);
int rtu_launch_feat4(const KernelArgs& args, uint32_t n_tiles, uint32_t bvh_stack_needed, bool stats, hipStream_t stream, int mode, const LaunchProbe* probe);
int rtu_launch_feat5(const KernelArgs& args, uint32_t n_tiles, uint32_t bvh_stack_needed, bool stats, hipStream_t stream, int mode, const LaunchProbe* probe);
int rtu_launch_feat10(const KernelArgs& args, uint32_t n_tiles, uint32_t bvh_stack_needed, bool stats, hipStream_t stream, int mode, const LaunchProbe* probe);
int rtu_launch_feat11(const KernelArgs& args, uint32_t n_tiles, uint32_t bvh_stack_needed, bool stats, hipStream_t stream, int mode, const LaunchProbe* probe);
int rtu_launch_feat16(const KernelArgs& args, uint32_t n_tiles, uint32_t bvh_stack_needed, bool stats, hipStream_t stream, int mode, const LaunchProbe* probe);
int rtu_launch_feat17(const KernelArgs& args, uint32_t n_tiles, uint32_t bvh_stack_needed, bool stats, hipStream_t stream, int mode, const LaunchProbe* probe);
int rtu_launch_feat18(const KernelArgs& args, uint32_t n_tiles, uint32_t bvh_stack_needed, bool stats, hipStream_t stream, int mode, const LaunchProbe* probe);
int rtu_launch_feat19(const KernelArgs& args, uint32_t n_tiles, uint32_t bvh_stack_needed, bool stats, hipStream_t stream, int mode, const LaunchProbe* probe);
int rtu_launch_feat26(const KernelArgs& args, uint32_t n_tiles, uint32_t bvh_stack_needed, bool stats, hipStream_t stream, int mode, const LaunchProbe* probe);
int rtu_launch_feat27(const KernelArgs& args, uint32_t n_tiles, uint32_t bvh_stack_needed, bool stats, hipStream_t stream, int mode, const LaunchProbe* probe);
int rtu_launch_feat20(const KernelArgs& args, uint32_t n_tiles, uint32_t bvh_stack_needed, bool stats, hipStream_t stream, int mode, const LaunchProbe* probe);
int rtu_launch_feat21(const KernelArgs& args, uint32_t n_tiles, uint32_t bvh_stack_needed, bool stats, hipStream_t stream, int mode, const LaunchProbe* probe);

int rtu_launch_frame(const KernelArgs& args, uint32_t n_tiles, uint32_t bvh_stack_needed, int stats, hipStream_t stream, int mode, const LaunchProbe* probe) {
    // textured scenes, sampled frames and batches of frames run their own instantiations: the others carry no
    // uvw, sample nothing and draw nothing.
    const bool ref = stats == 1;
    if (mode != RTU_LAUNCH_ALL) {
        if (stats == 2) return args.scene.textured ? rtu_launch_feat27(args, n_tiles, bvh_stack_needed, false, stream, mode, probe)
                                                   : rtu_launch_feat26(args, n_tiles, bvh_stack_needed, false, stream, mode, probe);
        if (args.scene.textured) return rtu_launch_feat11(args, n_tiles, bvh_stack_needed, ref, stream, mode, probe);
        return rtu_launch_feat10(args, n_tiles, bvh_stack_needed, ref, stream, mode, probe);
    }
    const int feat = (args.scene.textured ? 1 : 0) | (args.sampling ? 2 : (args.frame_batch ? 4 : 0));
    if (stats == 2) {  // touched-bytes mode: one frame, frames in flight, or a batch of samples
        switch (feat) {
            case 0: return rtu_launch_feat16(args, n_tiles, bvh_stack_needed, false, stream, mode, probe);
            case 1: return rtu_launch_feat17(args, n_tiles, bvh_stack_needed, false, stream, mode, probe);
            case 2: return rtu_launch_feat18(args, n_tiles, bvh_stack_needed, false, stream, mode, probe);
            case 3: return rtu_launch_feat19(args, n_tiles, bvh_stack_needed, false, stream, mode, probe);
            case 4: return rtu_launch_feat20(args, n_tiles, bvh_stack_needed, false, stream, mode, probe);
            case 5: return rtu_launch_feat21(args, n_tiles, bvh_stack_needed, false, stream, mode, probe);
            default: return (int)hipErrorInvalidValue;
        }
    }
    switch (feat) {
        case 0: return rtu_launch_feat0(args, n_tiles, bvh_stack_needed, ref, stream, mode, probe);
        case 1: return rtu_launch_feat1(args, n_tiles, bvh_stack_needed, ref, stream, mode, probe);
        case 2: return rtu_launch_feat2(args, n_tiles, bvh_stack_needed, ref, stream, mode, probe);
        case 3: return rtu_launch_feat3(args, n_tiles, bvh_stack_needed, ref, stream, mode, probe);
        case 4: return rtu_launch_feat4(args, n_tiles, bvh_stack_needed, ref, stream, mode, probe);
        default: return rtu_launch_feat5(args, n_tiles, bvh_stack_needed, ref, stream, mode, probe);
    }
}

int rtu_launch_gi_final(const KernelArgs& args, hipStream_t stream) {
    hipLaunchKernelGGL(k_gi_final, dim3((args.gi_total + 255u) / 256u), dim3(256), 0, stream, args);
    return (int)hipGetLastError();
}
