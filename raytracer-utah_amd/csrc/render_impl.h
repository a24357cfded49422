// render_impl.h — the per-pixel render hot path as a LEVEL-SYNCHRONOUS WAVEFRONT
// of hand-written gfx950 kernels. No MFMA: there is no dense contraction here.
//
// Why not one thread = one pixel for the whole recursion (round-1 v1 of this file):
// rocprof showed the GPU ~95 % idle — a pixel on the glass sphere fires ~20 rays
// ONE AFTER ANOTHER (depth-first Shade recursion), a wave is as slow as its slowest
// lane, and a handful of such waves lasted as long as the whole launch
// (profiles/r01_v1_*). The reference's recursion tree is therefore evaluated level
// by level; everything that is independent runs in parallel across the chip:
//
//   k_primary      one 8x8 pixel tile per wavefront: primary ray -> closest hit; a miss
//                  writes the pixel, a hit appends a level-0 "frame" (= one Shade() call)
//   per level L = 0 .. max_bounce:
//     k_trace(L)   one lane per (frame, ray slot): the shadow ray of every non-ambient
//                  light, the refracted/TIR ray, the Fresnel ray, the mirror ray — slot-major,
//                  so a wavefront traces 64 rays of the same kind for 64 neighbouring frames
//     k_consume(L) one lane per frame: direct lighting in the reference's light order; hits
//                  of secondary rays become frames of level L+1; a frame without children is final
//   per level L = max_bounce-1 .. 0:
//     k_combine(L) frames that wait for children combine them in the reference's exact term
//                  order (mtlFunctions.cpp:205-291); level 0 writes the pixel
//
// What replaces what (reference file:line):
//   tile/lane -> (x,y)     PixelIterator::GetPixelLocation          PixelIterator.h:25-38
//   k_primary              Render(): ray set-up + Trace             RenderFunctions.cpp:96-103,258-268
//   k_trace shadow slots   Light::Illuminate -> GenLight::Shadow    lightFunctions.cpp:27-84, lights.h:48
//   k_trace secondary      MtlBlinn::Shade ray generation           mtlFunctions.cpp:160-229,239,273-283
//   k_consume direct term  MtlBlinn::Shade light loop               mtlFunctions.cpp:125-155
//   finalize()             MtlBlinn::Shade combination              mtlFunctions.cpp:205-291
// The ray/scene arithmetic itself is in rtu_intersect.h.
//
// Exactness: a frame's arithmetic is the same sequence of float operations as the
// recursive code; only the ORDER IN WHICH INDEPENDENT FRAMES ARE EVALUATED changes.
// In the fast variant the Fresnel ray is traced speculatively alongside the
// refracted ray (its result is used only if the refracted ray hit, :234-251); the
// counting variant (collect_stats) traces it in a second pass so that its ray and
// traversal counters equal the CPU oracle's.
#ifndef RTU_RENDER_IMPL_H_INCLUDED
#define RTU_RENDER_IMPL_H_INCLUDED
#include "rtu_intersect.h"

namespace {

// Feature mask of a kernel instantiation (template parameter TEX): bit 0 the scene is textured (uvw
// carried, maps sampled), bit 1 the frame is sampled (recipe S: sample streams, soft shadows, glossy
// bounces, lens). Recipe W on an untextured scene compiles to exactly the code it had before either existed.
#define TEXD ((TEX & 1) != 0)
#define SMPD ((TEX & 2) != 0)
// bit 2: the launch renders a batch of FRAMES of recipe W (rtu_render_frames_device), each with its own
// camera; its pixel index space is [frame in batch][pixel of the shard], like a batch of samples
#define BATD ((TEX & 4) != 0)
// bit 3: recipe P (with bit 1): the launches of the Monte-Carlo gather — chain tracing and the two Shade()
// trees per chain hit, one of them lit by MonteCarlo()'s AmbientLight
#define GID ((TEX & 8) != 0)
// bit 4: touched-bytes mode (collect_stats == 2): the FAST variant's kernels — same trees, same culling, same two stages,
// same images — counting per kernel launch what they read and write (Counters::t_*, RtuTouched in rtu_render.h). The
// roofline of bench.py is computed from these counters, i.e. from the work the timed kernels themselves do.
#define CNTD ((TEX & 16) != 0)
// occupancy hints per kernel family (amdgpu_waves_per_eu), see DESIGN.md 5 for what was measured
#ifndef RTU_OCC_PRIMARY
// k_primary: six wavefronts per SIMD (<= 85 VGPRs) — the tile loop left alone takes 97 and fits five: 335 us per 16 frames against 304;
// eight (64 VGPRs, spills): 350. k_trace / k_consume at six and the stage-2 walks at four (from five / three): slower or no change.
// Round 3: the kernel also settles the childless Shade() calls of its hits (shadows_inline + direct_light: ground and wall pixels
// never become frames); six wavefronts then spill 85 registers (674 us per 32 frames), four spill two (528 us), left alone it takes
// 140 VGPRs and fits three.
#define RTU_OCC_PRIMARY __attribute__((amdgpu_waves_per_eu(4, 4)))
#endif
#ifndef RTU_OCC_PRIMARY_S
// k_primary_sampled (recipes S / P): five wavefronts per SIMD, no register spilled (six, as for recipe W: 13 spilled)
#define RTU_OCC_PRIMARY_S __attribute__((amdgpu_waves_per_eu(5, 5)))
#endif
#ifndef RTU_OCC_TRACE
// k_trace (fast variant): five wavefronts per SIMD (<= 96 VGPRs; left alone it takes 97 since the occluder-list lookups and fits four)
#define RTU_OCC_TRACE __attribute__((amdgpu_waves_per_eu(5, 5)))
#endif
#ifndef RTU_OCC_WALK
// the one-lane-per-ray stage-2 walks: three wavefronts per SIMD (<= 168 VGPRs; k_primary2 left alone takes 173 and fits two)
#define RTU_OCC_WALK __attribute__((amdgpu_waves_per_eu(3, 3)))
#endif
#ifndef RTU_OCC_CONSUME
#define RTU_OCC_CONSUME
#endif
#define RTU_BYTES(n) do { if (CNTD) cnt.t_bytes += (n); } while (0)
// childless Shade() calls settled by the lane that found the hit (primary_pixel), per kernel: stage 1 / the one-lane-per-ray stage 2
#ifndef RTU_INLINE_PRIMARY
#define RTU_INLINE_PRIMARY true
#endif
#ifndef RTU_INLINE_PRIMARY2
#define RTU_INLINE_PRIMARY2 true
#endif

enum { SLOT_MAIN = 0, SLOT_A = 1, SLOT_C = 2 };
// k_trace slot selection bits
enum { SEL_SHADOW = 1, SEL_MAIN = 2, SEL_A = 4, SEL_C = 8, SEL_A_NEEDS_B = 16 };

__device__ __forceinline__ uint32_t lane_id() { return threadIdx.x & 63u; }

// GPU-clock timeline (rtu_render_timeline): when a.tl is set the first 64 workgroups store the
// constant 100 MHz clock on entry and every wavefront stores it on exit (plain stores into
// per-workgroup slots, no atomics); the host takes min / max per kernel. Unlike a profiler's
// trace this does not serialise or pad the launches. kid: RTU_TL_* slot of the launch.
struct Stamp {
    unsigned long long* p;
    __device__ __forceinline__ Stamp(const KernelArgs& a, int kid) : p(nullptr) {
        if (a.tl) {
            p = a.tl + (size_t)kid * RTU_TL_STRIDE;
            if (blockIdx.x < 64u && threadIdx.x == 0) p[blockIdx.x] = (unsigned long long)wall_clock64();
        }
    }
    __device__ __forceinline__ ~Stamp() {
        if (p && lane_id() == 0) p[64u + ((blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) & (RTU_TL_ENDS - 1u))] = (unsigned long long)wall_clock64();
    }
};
enum { RTU_TL_PRIMARY = 0, RTU_TL_PRIMARY2C = 1, RTU_TL_PRIMARY2 = 2, RTU_TL_LEVEL0 = 3 /* +4L: trace, trace2c, trace2, consume */,
       RTU_TL_COMBINE0 = 3 + 4 * RTU_MAX_LEVELS, RTU_TL_GI_ROOTS = RTU_TL_COMBINE0 + RTU_MAX_LEVELS /* recipe P: k_gi_roots */ };
#define RTU_TL_SIDE_TAIL (RTU_TL_GI_ROOTS + 1)  // side mode: the k_tail launch behind stage 2 of the primary phase
static_assert(RTU_TL_SIDE_TAIL < RTU_TL_KERNELS, "timeline slots");

// Wave-aggregated append: every lane with `want` gets a unique index into the level's
// frame arrays; one atomic per wavefront. Must be reached by all 64 lanes.
__device__ __forceinline__ uint32_t wave_append(uint32_t* counter, bool want) {
    unsigned long long mask = __ballot(want);
    uint32_t base = 0;
    uint32_t leader = 0;
    if (mask != 0) {
        leader = (uint32_t)__ffsll((long long)mask) - 1u;
        if (lane_id() == leader) base = atomicAdd(counter, (uint32_t)__popcll(mask));
    }
    base = __shfl(base, (int)leader);
    unsigned long long below = mask & ((1ull << lane_id()) - 1ull);
    return base + (uint32_t)__popcll(below);
}

// Largest shard population of level L (wave-uniform): lane i reads shard i's counter.
__device__ __forceinline__ uint32_t level_max_count(const KernelArgs& a, int L) {
    uint32_t v = a.fcnt->n_frames[L][(lane_id() % RTU_SHARDS) * RTU_CSTRIDE];
    const uint32_t cap = a.lv[L].cap_s;
    if (v > cap) v = cap;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        uint32_t o = (uint32_t)__shfl_xor((int)v, off);
        v = o > v ? o : v;
    }
    return v;
}
__device__ __forceinline__ uint32_t shard_count(const KernelArgs& a, int L, uint32_t shard) {
    uint32_t v = a.fcnt->n_frames[L][(shard) * RTU_CSTRIDE];
    const uint32_t cap = a.lv[L].cap_s;
    return v > cap ? cap : v;
}

// The 64 shard populations of a list, one per lane (lane i: shard i), loaded ONCE per wavefront; a chunk reads its shard's with
// v_readlane. (Every chunk of every kernel used to load its counter from memory first: one more dependent round trip in front
// of the list entry, the frame record, the material — the chunk loops of the streaming kernels wait for memory half their time.)
__device__ __forceinline__ uint32_t shard_counts(const uint32_t* counters, uint32_t cap) {
    uint32_t v = counters[(lane_id() % RTU_SHARDS) * RTU_CSTRIDE];
    return v > cap ? cap : v;
}
__device__ __forceinline__ uint32_t count_of(uint32_t counts, uint32_t shard) {  // shard: wave-uniform
    return (uint32_t)__builtin_amdgcn_readlane((int)counts, (int)shard);
}
__device__ __forceinline__ uint32_t wave_max(uint32_t v) {  // the largest shard population (wave-uniform result)
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const uint32_t o = (uint32_t)__shfl_xor((int)v, off);
        v = o > v ? o : v;
    }
    return v;
}

template <bool STATS>
__device__ __forceinline__ void flush_counters(const KernelArgs& a, const Counters& cnt) {
    if (!STATS) return;
    unsigned vals[11] = {cnt.prim, cnt.prim_hit, cnt.sec, cnt.shd, cnt.node, cnt.mesh,
                         cnt.inner, cnt.leafv, cnt.leafe, cnt.tri, cnt.acc};
#pragma unroll
    for (int i = 0; i < 11; i++) {
        unsigned v = vals[i];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
        if (lane_id() == 0 && v) atomicAdd(&a.counters[i], (unsigned long long)v);
    }
}
// touched-bytes mode: row `kid` (the launch's timeline slot) of the counter table. Must be reached by all 64 lanes.
template <int TEX>
__device__ __forceinline__ void flush_touched(const KernelArgs& a, const Counters& cnt, int kid) {
    if (!CNTD) return;
    unsigned vals[RTU_TOUCH_FIELDS] = {cnt.t_rays, cnt.t_node, cnt.t_meshbox, cnt.t_inner4, cnt.t_inner8, cnt.t_innerref, cnt.t_tri, cnt.t_win,
                                       cnt.t_xform, cnt.t_bytes, cnt.t_bounds, cnt.t_inline};
#pragma unroll
    for (int i = 0; i < RTU_TOUCH_FIELDS; i++) {
        unsigned v = vals[i];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
        if (lane_id() == 0 && v) atomicAdd(&a.counters[(size_t)kid * RTU_TOUCH_STRIDE + i], (unsigned long long)v);
    }
}

// Which rays will this Shade() call fire? Decided once, when the frame is created.
template <int TEX>
__device__ __forceinline__ uint32_t make_info(const KernelArgs& a, int mtl, int bounce, bool front, f3 dir, f3 p, f3 N, f3 uvw, Smp smp, bool amb = false) {
    const DevScene& s = a.scene;
    const RTU_CONST RtuMaterial& m = as_const(s.materials)[mtl];
    uint32_t info = (uint32_t)mtl | ((uint32_t)bounce << RTU_FI_BOUNCE_SH) | (front ? RTU_FI_FRONT : 0u);
    if (amb) info |= RTU_FI_AMB;                                         // the light list is one AmbientLight: no shadow rays
    else if (front && s.n_lights > 0) info |= RTU_FI_SH;                 // mtlFunctions.cpp:125
    // LIGHTS BEHIND THE SURFACE. The light loop clamps N.L at zero (:141-148) and multiplies the light's whole term by it
    // (:152): with N.L <= 0 the term is Illuminate() * (+-0) * (...) = +-0 whether Shadow() returned 0 or 1 — as long as
    // Illuminate() is finite either way, which it is for a finite intensity below 1e15 (nol_ok) at more than 1e-10 from a
    // point light. So no shadow ray is needed there (the fast variant fires none, frame_ray; the counting variant does, and
    // must — and does — render the same bits). The test is on the unnormalised cosine with a margin of 1e-3: the
    // reference's own N.L, computed through two normalisations (:138, lights.h:49,83), differs from the true cosine by
    // ~1e-6, so "clearly negative" here implies "negative" there.
    if ((info & RTU_FI_SH) && s.nol_ok) {
        const uint32_t nl = a.nsl < RTU_FI_NOL_LIGHTS ? a.nsl : RTU_FI_NOL_LIGHTS;
        const float nn = dot3(N, N);
        for (uint32_t j = 0; j < nl; j++) {
            const f3 lvec = ld3(a.nol_light[j]);
            const f3 tl = a.nol_light[j][3] != 0.0f ? -lvec : lvec - p;  // towards the light
            const float c = dot3(N, tl), tt = dot3(tl, tl);
            if (c < 0.0f && c * c > 1e-6f * (tt * nn) && tt > 1e-20f && tt < 1e30f) info |= 1u << (RTU_FI_NOL_SH + j);
        }
    }
    if (bounce > 0) {                                                   // :158
        if (not_black(mtl_color<TEXD>(s, mtl, RTU_MAP_REFRACTION, ld3(m.refraction), uvw))) {  // :160
            info |= RTU_FI_MAIN;
            Refr r = refraction_terms(dir, p, N, front, m.ior, smp, smp.on ? m.refraction_glossiness : 0.0f);
            if (r.sinTheta2 > 1) info |= RTU_FI_TIR;                    // :205
        }
        if (not_black(mtl_color<TEXD>(s, mtl, RTU_MAP_REFLECTION, ld3(m.reflection), uvw))) info |= RTU_FI_C;  // :273
    }
    return info;
}

// Direction of secondary ray `slot` of a frame (mtlFunctions.cpp:207, :229, :239, :280).
__device__ __forceinline__ f3 secondary_dir(int slot, uint32_t info, f3 dir, f3 p, f3 N, const RTU_CONST RtuMaterial& m, Smp smp) {
    if (slot == SLOT_C) return reflect_dir(dir, sampled_normal(p, N, smp, smp.on ? m.reflection_glossiness : 0.0f, RTU_DRAW_REFL));
    Refr t = refraction_terms(dir, p, N, (info & RTU_FI_FRONT) != 0, m.ior, smp, smp.on ? m.refraction_glossiness : 0.0f);
    if (info & RTU_FI_TIR) return reflect_dir(dir, t.sn);    // :207, the first sample
    if (slot == SLOT_A) return reflect_dir(dir, t.sn2);      // :239, the second sample shadows the first
    return norm3((-t.sn2) * t.cosTheta2 + t.SVector * t.sinTheta2);  // :229
}

// Append a level-0 frame (a Shade() call at a primary or, recipe P, a chain hit) and its entries in the two
// slot lists: three appends issued back to back, one wait for the wavefront instead of three. Wave-uniform
// call; `want` says whether the lane has a frame. Returns the frame index or ~0u.
template <int TEX>
__device__ __forceinline__ uint32_t append_root(const KernelArgs& a, bool want, uint32_t shard, uint32_t info, f3 p, f3 N, uint32_t fbw, f3 dir,
                                                float fcw, f3 uvw, Counters& cnt, const bool stage2 = false) {
    const LevelBuffers& lv = a.lv[0];
    const unsigned long long below = (1ull << (threadIdx.x & 63u)) - 1ull;
    const bool wm = want && (info & RTU_FI_MAIN), wc = want && (info & RTU_FI_C);
    const unsigned long long mf = __ballot(want), mm = __ballot(wm), mc = __ballot(wc);
    uint32_t bf = 0, bm = 0, bc = 0;
    if (mf) {
        const uint32_t leader = (uint32_t)__ffsll((long long)mf) - 1u;
        if ((threadIdx.x & 63u) == leader) {
            if (stage2) atomicAdd(&a.fcnt->stage2_frames[(shard) * RTU_CSTRIDE], (uint32_t)__popcll(mf));  // (what the host decides side mode by: FrameCounters)
            bf = atomicAdd(&a.fcnt->n_frames[0][(shard) * RTU_CSTRIDE], (uint32_t)__popcll(mf));
            if (mm) bm = atomicAdd(&a.fcnt->n_lmain[0][(shard) * RTU_CSTRIDE], (uint32_t)__popcll(mm));
            if (mc) bc = atomicAdd(&a.fcnt->n_lrefl[0][(shard) * RTU_CSTRIDE], (uint32_t)__popcll(mc));
        }
        bf = (uint32_t)__shfl((int)bf, (int)leader);
        bm = (uint32_t)__shfl((int)bm, (int)leader);
        bc = (uint32_t)__shfl((int)bc, (int)leader);
    }
    uint32_t idx = ~0u;
    if (want) {
        const uint32_t fl = bf + (uint32_t)__popcll(mf & below);
        if (fl < lv.cap_s) {  // level 0 is sized for every root of the launch (ensure_levels): always true
            idx = fl + shard * lv.cap_s;
            RTU_BYTES(48u + (TEXD ? 16u : 0u) + (wm ? 4u : 0u) + (wc ? 4u : 0u));
            if (TEXD) lv.fuv[idx] = make_float4(uvw.x, uvw.y, uvw.z, 0.0f);
            lv.fa[idx] = make_float4(p.x, p.y, p.z, __uint_as_float(info));
            lv.fb[idx] = make_float4(N.x, N.y, N.z, __uint_as_float(fbw));
            lv.fc[idx] = make_float4(dir.x, dir.y, dir.z, fcw);
            // (the list counters also count the lanes of wavefronts that found the shard full — possible in the small side arrays of side mode —:
            // an entry is written only inside the shard's region; the overflow is reported and the frame rendered again)
            const uint32_t im = bm + (uint32_t)__popcll(mm & below), ic = bc + (uint32_t)__popcll(mc & below);
            if (wm && im < lv.cap_s) lv.lmain[(size_t)shard * lv.cap_s + im] = fl;
            if (wc && ic < lv.cap_s) lv.lrefl[(size_t)shard * lv.cap_s + ic] = fl;
            if ((wm && im >= lv.cap_s) || (wc && ic >= lv.cap_s)) a.fcnt->overflow = 1;
        } else {
            a.fcnt->overflow = 1;
        }
    }
    return idx;
}

// A sampled launch renders a.batch consecutive samples of the frame at once (longer ray lists fill the chip
// better): its pixel index space is [sample in batch][pixel of the shard]. pix -> global x, y and the
// sample's place in the batch.
template <int TEX>
__device__ __forceinline__ void pixel_of(const KernelArgs& a, uint32_t pix, int& x, int& y, uint32_t& sidx) {
    sidx = 0;
    uint32_t lp = pix;
    if (SMPD || BATD) {
        sidx = pix / a.batch_pixels;
        lp = pix - sidx * a.batch_pixels;
    }
    const uint32_t ly = lp / (uint32_t)a.frame.width;
    x = (int)(lp - ly * (uint32_t)a.frame.width);
    y = (int)(((ly / RTU_BAND_ROWS) * a.frame.shard_count + a.frame.shard_rank) * RTU_BAND_ROWS + ly % RTU_BAND_ROWS);
}

// The key of the sample streams of the Shade() call a frame stands for (recipe S): level 0 frames
// carry their pixel in fb.w, deeper frames their key.
template <int TEX>
__device__ __forceinline__ Smp frame_smp(const KernelArgs& a, int L, float fbw) {
    Smp smp;
    smp.on = SMPD;
    smp.key = 0;
    if (smp.on) {
        const uint32_t w = __float_as_uint(fbw);
        if (L == 0 && !GID) {
            int x, y;
            uint32_t sidx;
            pixel_of<TEX>(a, w, x, y, sidx);
            smp.key = sample_key((uint32_t)x + (uint32_t)a.frame.width * (uint32_t)y, a.sample_index + sidx);
        } else {
            smp.key = w;
        }
    }
    return smp;
}

// ------------------------------------------------------------------------------------
// TWO-STAGE PHASES. A ray that never enters a mesh's bounding box costs a few hundred
// instructions; one that does walks a BVH (tens to hundreds of dependent steps) and a
// wavefront is as slow as its slowest lane. So every tracing phase runs twice:
//   stage 1 (wide): 64 rays per wavefront; the scene-graph walk is abandoned the moment a
//           ray passes a mesh's bounding box and the ray id is appended to a defer list;
//   stage 2 (narrow): the deferred rays, compacted, walk the whole scene including the
//           BVH with only R rays per wavefront (R = 8..64 chosen from the list length), so
//           a wavefront waits for the slowest of R rays instead of 64 and all of its lanes
//           are inside the BVH loop together.
// The counting variant (STATS) does everything in stage 1 so that every node test is
// counted exactly once.

// Append one ray id to the defer list of phase `ph` (sharded like the frame arrays).
__device__ __forceinline__ void defer_push(const KernelArgs& a, int ph, uint32_t shard, bool want, uint32_t id) {
    if (ph == 0) {  // the primary phase has a list and counters of its own (KernelArgs::fcnt0)
        uint32_t idx = wave_append(&a.fcnt0->n_defer[0][(shard) * RTU_CSTRIDE], want);
        if (want) {
            if (idx < a.defer_cap0_s) a.defer_list0[(size_t)shard * a.defer_cap0_s + idx] = id;
            else a.fcnt0->overflow = 1;
        }
        return;
    }
    uint32_t idx = wave_append(&a.fcnt->n_defer[ph][(shard) * RTU_CSTRIDE], want);
    if (want) {
        if (idx < a.defer_cap_s) a.defer_list[(size_t)shard * a.defer_cap_s + idx] = id;
        else a.fcnt->overflow = 1;
    }
}

// Geometry of a stage-2 launch: rays per wavefront from the (largest shard of the) list.
struct NarrowGeom {
    uint32_t R, kmax, nmax, sum;
    uint32_t counts;  // lane i: the population of shard i (count_of)
};
__device__ __forceinline__ NarrowGeom narrow_geom(const KernelArgs& a, int ph) {
    uint32_t v = (ph == 0 ? a.fcnt0 : a.fcnt)->n_defer[ph][(lane_id() % RTU_SHARDS) * RTU_CSTRIDE];
    const uint32_t capd = ph == 0 ? a.defer_cap0_s : a.defer_cap_s;
    if (v > capd) v = capd;
    const uint32_t mine = v;
    uint32_t sum = v;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        uint32_t o = (uint32_t)__shfl_xor((int)v, off);
        v = o > v ? o : v;
        sum += (uint32_t)__shfl_xor((int)sum, off);
    }
    NarrowGeom g;
    g.nmax = v;
    g.sum = sum;
    g.counts = mine;
    // enough wavefronts to fill 256 CUs several times over before widening them
    // few rays: latency matters, eight lanes per ray; many rays: throughput matters, one lane per ray
    g.R = sum > (uint32_t)(a.frame.coop_threshold > 0 ? a.frame.coop_threshold : 70000) ? 64u : 8u;  // (round 2: the lists are shorter — masks — and so is the break-even: 40 / 70 / 100 / 120 k measured)
    g.kmax = (v + g.R - 1u) / g.R;
    return g;
}

template <int TEX> __device__ __forceinline__ bool shadows_inline(const KernelArgs& a, uint32_t info, f3 p, uint32_t& lit_mask, Counters& cnt);
template <bool STATS, int TEX, class ShadowFn>
__device__ __forceinline__ f3 direct_light(const KernelArgs& a, uint32_t info, f3 p, f3 N, f3 uvw, f3 cam_pos, f3 direct, ShadowFn sh_of);

// ---- the primary ray of one pixel -------------------------------------------------------
template <int STACK, bool STATS, bool DEFER, bool COOP, int TEX, bool INLINE_SHADE = false>
__device__ __forceinline__ void primary_pixel(const KernelArgs& a, bool valid, int x, int y, uint32_t sidx, uint32_t pix, uint32_t shard,
                                              uint32_t* stk, Counters& cnt, bool& deferred, bool leader = true,
                                              const uint32_t stride = 64, const float4* lds_nodes = nullptr) {
    const DevScene& s = a.scene;
    bool want = false;
    unsigned walk_units = 0;  // touched-bytes mode: the work of this ray's BVH walk (two per inner step of the 4-wide tree, one per triangle test)
    Hit h;
    fresh_hit(h, RTU_BIGFLOAT);
    Ray ray;
    f3 cam_origin = ld3(a.frame.origin), cam_u = ld3(a.frame.u), cam_v = ld3(a.frame.v);
    ray.p = ld3(a.frame.cam_pos);
    if (BATD) {  // every frame of the batch has its own camera
        const RTU_CONST BatchCam& c = as_const(a.cam)[sidx];  // scalar loads where sidx is wave-uniform (stage 1)
        ray.p = ld3(c.pos);
        cam_origin = ld3(c.origin);
        cam_u = ld3(c.u);
        cam_v = ld3(c.v);
    }
    ray.dir = mk3(0, 0, 0);
    int mid = -1;
    deferred = false;
    Smp smp;
    smp.on = SMPD;
    smp.key = 0;
    // node-level bounds of a primary ray (recipe W): the pixel against every node's screen rectangle for this camera
    // (k_node_rects). A wavefront whose pixels lie outside every rectangle has nothing to trace: background, done.
    unsigned long long skip = 0;
    // (stage 1 only: a deferred pixel has passed its rectangles and masks already; stage 2, where every lane has its own camera
    // entry and the lookups would be a dozen per-lane loads, bounds its ray in world space like any other ray — trace())
    const bool rects = !STATS && !SMPD && DEFER && a.node_rects != nullptr;
    if (rects) {
        const RTU_CONST int* rc = as_const(reinterpret_cast<const int*>(a.node_rects)) + 4u * (size_t)sidx * s.n_nodes;  // scalar loads where sidx is wave-uniform
        const uint32_t nn = s.n_nodes < 64u ? s.n_nodes : 64u;
        for (uint32_t k = 0; k < nn; k++) {
            const int x0 = rc[4u * k], y0 = rc[4u * k + 1u], x1 = rc[4u * k + 2u], y1 = rc[4u * k + 3u];
            if (x < x0 || x >= x1 || y < y0 || y >= y1) skip |= 1ull << k;
        }
        if (a.cover) {  // inside a mesh's rectangle: does any of its triangles reach this 8x8 tile? (k_mesh_cover)
            const uint32_t tile = (uint32_t)(y >> 3) * a.tiles_xf + (uint32_t)(x >> 3);
            const uint32_t ncov = s.n_cover + s.n_pcover;  // masked mesh nodes, then masked plane nodes
            for (uint32_t c = 0; c < ncov; c++) {
                const uint32_t k = (uint32_t)(c < s.n_cover ? s.cover_node[c] : s.pcover_node[c - s.n_cover]);
                const bool inside = valid && k < 64u && !((skip >> k) & 1ull);  // inside the mesh's rectangle
                if (!__any(inside)) continue;                                    // (most wavefronts: no load at all)
                const RTU_CONST uint32_t* m = as_const(a.cover) + ((size_t)sidx * ncov + c) * (1u + a.cover_words);
                if (inside && m[0] == 0u && !((m[1u + (tile >> 5)] >> (tile & 31u)) & 1u)) skip |= 1ull << k;
            }
            if (CNTD && leader && valid) cnt.t_bytes += 4u * ncov;  // (the lane's word of each mask it looked at, at most)
        }
        if (CNTD && (threadIdx.x & 63u) == (uint32_t)__ffsll((long long)__ballot(1)) - 1u) cnt.t_bytes += 16u * nn;  // wave-uniform: once per wavefront
        if (s.n_nodes <= 64u && __all(!valid || (skip & s.obj_mask) == s.obj_mask)) {
            if (valid && leader) {
                RTU_CNT(prim);
                const f3 bg = background_sample<TEXD>(s, x, y);  // :145
                a.out[pix] = make_float4(bg.x, bg.y, bg.z, RTU_BIGFLOAT);
                RTU_BYTES(16u);
                if (CNTD) cnt.t_rays++;  // the ray exists; it touches nothing
            }
            return;
        }
    }
    if (GID && a.gi_depth > 0) {
        // recipe P, chain depth k > 0: the gather ray from the hit of depth k - 1 (RenderFunctions.cpp:556-565)
        const size_t hb = (size_t)(a.gi_depth - 1u) * 4u * a.gi_total + pix;
        float4 hA = make_float4(0, 0, 0, 0), hB = hA, hC = hA;
        if (valid) { hA = a.gi_h[hb]; hB = a.gi_h[hb + a.gi_total]; hC = a.gi_h[hb + 2u * (size_t)a.gi_total]; }
        if (valid && leader) RTU_BYTES(48u);  // the chain's record of the depth above
        const size_t ho = (size_t)a.gi_depth * 4u * a.gi_total + pix;
        if (valid && !(__float_as_uint(hB.w) & 1u)) {  // the chain ended above: no hit at this depth either
            if (leader) a.gi_h[ho + a.gi_total] = make_float4(0.0f, 0.0f, 0.0f, __uint_as_float(0u));
            if (leader) RTU_BYTES(16u);
            valid = false;
        }
        if (valid) {
            const uint32_t pkey = __float_as_uint(hC.w);
            const f3 sampleOffset = sample_hemisphere_cosine(mk3(hB.x, hB.y, hB.z), pkey);
            ray.p = mk3(hA.x, hA.y, hA.z);
            ray.dir = norm3(sampleOffset);  // :562
            smp.key = child_key(pkey, RTU_SLOT_GATHER);
            bool hit = trace<STACK, STATS, !STATS, DEFER, COOP, TEXD, CNTD>(s, ray, false, h, stk, cnt, deferred, stride, lds_nodes);
            if (!deferred && leader) {
                const int hmid = hit ? as_const(s.nodes)[h.node].material_id : -1;
                a.gi_h[ho] = make_float4(h.p.x, h.p.y, h.p.z, h.z);
                a.gi_h[ho + a.gi_total] = make_float4(h.N.x, h.N.y, h.N.z, __uint_as_float((hit ? 1u : 0u) | (h.front ? 2u : 0u) | ((uint32_t)(hmid + 1) << 2)));
                a.gi_h[ho + 2u * (size_t)a.gi_total] = make_float4(ray.dir.x, ray.dir.y, ray.dir.z, __uint_as_float(smp.key));
                if (TEXD) a.gi_h[ho + 3u * (size_t)a.gi_total] = make_float4(h.uvw.x, h.uvw.y, h.uvw.z, 0.0f);
                RTU_BYTES(TEXD ? 64u : 48u);  // this depth's record
            }
        }
        return;
    }
    if (valid) {
        float ox = 0.5f, oy = 0.5f;  // recipe W: the pixel centre
        if (smp.on) {
            // recipe S: RenderFunctions.cpp:80-97 — Halton offsets, a point of the lens disk
            smp.key = sample_key((uint32_t)x + (uint32_t)a.frame.width * (uint32_t)y, a.sample_index + sidx);
            ox = a.pix_off_x[sidx];
            oy = a.pix_off_y[sidx];
            const float sampleX = (float)rand31(smp.key, RTU_DRAW_LENS) / RTU_RAND_MAX_F;          // :88
            const float sampleTheta = (float)rand31(smp.key, RTU_DRAW_LENS + 1u) / RTU_THETA_DIV;  // :89
            float sn, cs;
            portable_sincos(sampleTheta, sn, cs);
            const float rad = sqrtf((sampleX * a.frame.dof) * a.frame.dof);
            const float camOffsetX = rad * cs, camOffsetY = rad * sn;                               // :90-91
            ray.p = (ray.p + ld3(a.frame.lens_up) * camOffsetY) + ld3(a.frame.lens_right) * camOffsetX;  // :93
        }
        // RenderFunctions.cpp:258-268, :97
        f3 cp = (cam_origin + cam_u * ((float)x + ox)) + cam_v * ((float)y + oy);
        ray.dir = norm3(cp - ray.p);
        RTU_CNT(prim);
        bool hit = false;
        const unsigned units0 = CNTD ? cnt.t_inner4 * 2u + cnt.t_tri : 0u;
        if (!(a.dbg & 4u)) hit = trace<STACK, STATS, !STATS, DEFER, COOP, TEXD, CNTD>(s, ray, false, h, stk, cnt, deferred, stride, lds_nodes, skip, rects);
        if (CNTD) walk_units = cnt.t_inner4 * 2u + cnt.t_tri - units0;
        if (!deferred && leader) {
            if (!hit) {
                f3 bg = background_sample<TEXD>(s, x, y);  // :145
                if (!(a.dbg & 16u)) a.out[pix] = make_float4(bg.x, bg.y, bg.z, h.z);
                RTU_BYTES(16u);
            } else {
                RTU_CNT(prim_hit);
                mid = as_const(s.nodes)[h.node].material_id;
                if (mid < 0) { a.out[pix] = make_float4(1.0f, 1.0f, 1.0f, h.z); RTU_BYTES(16u); }  // null material => white
                else want = true;
            }
            if (GID) {  // recipe P: the chain's depth-0 record instead of a frame; a null material stays white
                a.gi_h[pix] = make_float4(h.p.x, h.p.y, h.p.z, h.z);
                a.gi_h[pix + a.gi_total] = make_float4(h.N.x, h.N.y, h.N.z, __uint_as_float((want ? 1u : 0u) | (h.front ? 2u : 0u) | ((uint32_t)(mid + 1) << 2)));
                a.gi_h[pix + 2u * (size_t)a.gi_total] = make_float4(ray.dir.x, ray.dir.y, ray.dir.z, __uint_as_float(smp.key));
                if (TEXD) a.gi_h[pix + 3u * (size_t)a.gi_total] = make_float4(h.uvw.x, h.uvw.y, h.uvw.z, 0.0f);
                RTU_BYTES(TEXD ? 64u : 48u);  // the chain's depth-0 record
                want = false;
            }
        }
    }
    if (GID) return;
    uint32_t info = 0;
    if (want) info = make_info<TEX>(a, mid, a.frame.max_bounce, h.front, ray.dir, h.p, h.N, h.uvw, smp);
    // A childless Shade() call is settled by the lane that found the hit (shadows_inline): no frame, the pixel is final.
    // (rtu_debug_flags 2048 switches this off: results must not change)
    if (!STATS && INLINE_SHADE && !(a.dbg & 2048u) && __any(want && !(info & (RTU_FI_MAIN | RTU_FI_C)))) {
        const bool tryI = want && !(info & (RTU_FI_MAIN | RTU_FI_C));
        uint32_t lit = ~0u;
        bool ok = false;
        if (tryI) ok = shadows_inline<TEX>(a, info, h.p, lit, cnt);
        if (tryI && ok) {
            f3 direct = mk3(0, 0, 0);
            if (info & RTU_FI_SH) {
                const f3 cam_pos = BATD ? ld3(a.cam[sidx].pos) : ld3(a.frame.cam_pos);  // :137: camera.pos, whatever the lens sample
                direct = direct_light<false, TEX>(a, info, h.p, h.N, h.uvw, cam_pos, direct, [&](uint32_t j) { return ((lit >> j) & 1u) ? 1.0f : 0.0f; });
            }
            a.out[pix] = make_float4(direct.x, direct.y, direct.z, h.z);
            RTU_BYTES(16u);
            want = false;
        }
    }
    if (!(a.dbg & 8u)) append_root<TEX>(a, want, shard, info, h.p, h.N, pix, ray.dir, h.z, h.uvw, cnt, !DEFER && !STATS);
    // (diagnostics, rtu_debug_flags 131072 in touched-bytes mode: the one-lane-per-ray stage 2 writes the walk's units over the pixel's red
    // channel — the distribution of walk lengths, ray by ray: tools/scratch/walk_units.py. A WRONG image, like the other experiment bits.)
    if (CNTD && !DEFER && !COOP && (a.dbg & 131072u) && valid) a.out[pix].x = (float)walk_units;
}

// stage 1: one 8x8 pixel tile per wavefront, four wavefronts per workgroup
template <int STACK, bool STATS, int TEX>
__device__ __forceinline__ void primary_stage1(const KernelArgs& a, uint32_t n_tiles) {
    const Stamp stamp(a, RTU_TL_PRIMARY);
    __shared__ uint32_t s_stack_all[STATS ? 4 * STACK * 64 : 4];
    uint32_t* stk = s_stack_all + (STATS ? (threadIdx.x >> 6) * (STACK * 64) + (threadIdx.x & 63u) : 0);
    const uint32_t lane = threadIdx.x & 63u;
    Counters cnt = {};
    // tiles are strided over the grid (a wavefront renders several: the launch has fewer, longer-lived wavefronts)
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));  // wave-uniform: the tile arithmetic stays in scalar registers
    for (uint32_t btile = blockIdx.x * 4u + wave; btile < n_tiles; btile += gridDim.x * 4u) {  // whole wavefronts
        // (two scalar divisions per tile; replacing them by additions and carries from a per-wavefront decomposition of the stride was
        // measured: no faster — six more scalar registers in a kernel that already spills them)
        const uint32_t sidx = (SMPD || BATD) ? btile / a.tiles_per_image : 0u;  // batched launches: n_tiles = batch x tiles of the image
        const uint32_t tile = btile - sidx * a.tiles_per_image;
        const uint32_t band_local = tile / a.tiles_x;
        const uint32_t tx = tile - band_local * a.tiles_x;
        const int x = (int)(tx * 8 + (lane & 7));
        const int ly = (int)(band_local * RTU_BAND_ROWS + (lane >> 3));                                              // row inside the shard
        const int y = (int)((band_local * a.frame.shard_count + a.frame.shard_rank) * RTU_BAND_ROWS + (lane >> 3));  // global row
        const bool valid = x < a.frame.width && y < a.frame.height;
        const uint32_t pix = ((SMPD || BATD) ? sidx * a.batch_pixels : 0u) + (uint32_t)ly * (uint32_t)a.frame.width + (uint32_t)x;
        if (!STATS && !SMPD && !GID && a.occ != nullptr) {
            // TILE OCCUPANCY (k_tile_occ): no pixel of this tile is inside any node's screen rectangle (or marked tile of a mesh's
            // coverage mask) for this camera — most of a frame, typically: the background, and on to the next tile without a look
            // at the camera, the rectangles or the masks (the same pixels primary_pixel's own test would send here)
            const uint32_t w = as_const(a.occ)[(size_t)sidx * a.occ_words + (tile >> 5)];  // wave-uniform: a scalar load
            if (!((w >> (tile & 31u)) & 1u)) {
                if (valid) {
                    const f3 bg = background_sample<TEXD>(a.scene, x, y);  // :145
                    a.out[pix] = make_float4(bg.x, bg.y, bg.z, RTU_BIGFLOAT);
                    RTU_BYTES(16u);
                    if (CNTD) cnt.t_rays++;  // the ray exists; it touches nothing
                }
                if (CNTD && lane == 0) cnt.t_bytes += 4u;  // the occupancy word, once per wavefront
                continue;
            }
        }
        const uint32_t shard = btile % RTU_SHARDS;
        bool deferred;
        primary_pixel<STACK, STATS, !STATS, false, TEX, RTU_INLINE_PRIMARY>(a, valid, x, y, sidx, pix, shard, stk, cnt, deferred);
        if (!STATS && !(a.dbg & 8u)) defer_push(a, 0, shard, deferred, pix);
        if (deferred) RTU_BYTES(4u);
    }
    flush_counters<STATS>(a, cnt);
    flush_touched<TEX>(a, cnt, RTU_TL_PRIMARY);
}
// (three kernels around one body: the occupancy hint is recipe W's fast variant's — mostly background tiles and stores; the sampled
// and path-traced variants carry the lens / hemisphere sampling in binary64 and spill 13 registers under it, the counting variant
// keeps a traversal stack per lane)
template <int STACK, bool STATS, int TEX>
__global__ void __launch_bounds__(256) RTU_OCC_PRIMARY k_primary(KernelArgs a, uint32_t n_tiles) { primary_stage1<STACK, STATS, TEX>(a, n_tiles); }
template <int STACK, int TEX>
__global__ void __launch_bounds__(256) RTU_OCC_PRIMARY_S k_primary_sampled(KernelArgs a, uint32_t n_tiles) { primary_stage1<STACK, false, TEX>(a, n_tiles); }
template <int STACK, int TEX>
__global__ void __launch_bounds__(256) k_primary_counting(KernelArgs a, uint32_t n_tiles) { primary_stage1<STACK, true, TEX>(a, n_tiles); }

// Stage the top of every mesh's 8-wide tree (BFS order) into the workgroup's LDS node area.
__device__ __forceinline__ void stage_nodes(const KernelArgs& a, float4* lds_nodes) {
    const RTU_CONST DevMesh* meshes = as_const(a.scene.meshes);
    for (uint32_t m = 0; m < a.n_meshes; m++) {
        const float4* src = meshes[m].bvh8;
        const uint32_t n = meshes[m].lds_nodes * 16u, off = meshes[m].lds_off;
        for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) lds_nodes[off + i] = src[i];
    }
    __syncthreads();
}

// stage 2 of the primary phase, long lists: one lane per deferred pixel, 64 per wavefront
template <int STACK, int TEX>
__global__ void __launch_bounds__(64) RTU_OCC_WALK k_primary2(KernelArgs a, int alone) {
    const Stamp stamp(a, RTU_TL_PRIMARY2);
    __shared__ uint32_t s_stack[STACK * 64];
    const uint32_t lane = threadIdx.x;
    const NarrowGeom g = narrow_geom(a, 0);
    if (g.R == 8u && !alone) return;  // short list: k_primary2c takes it (alone: it was not launched — launch_all)
    const uint32_t kmax = (g.nmax + 63u) / 64u;
    const uint32_t chunks = kmax * RTU_SHARDS;
    Counters cnt = {};
    const uint32_t counts = g.counts;
    for (uint32_t c = blockIdx.x; c < chunks; c += gridDim.x) {
        const uint32_t shard = c % RTU_SHARDS, k = c / RTU_SHARDS;
        const uint32_t ns = count_of(counts, shard);
        const uint32_t e = k * 64u + lane;
        const bool valid = e < ns;
        uint32_t pix = 0;
        if (valid) pix = a.defer_list0[(size_t)shard * a.defer_cap0_s + e];
        if (valid) RTU_BYTES(4u);
        int x, y;
        uint32_t sidx;
        pixel_of<TEX>(a, pix, x, y, sidx);
        bool deferred;
        primary_pixel<STACK, false, false, false, TEX, RTU_INLINE_PRIMARY2>(a, valid, x, y, sidx, pix, shard, s_stack + lane, cnt, deferred);
    }
    flush_touched<TEX>(a, cnt, RTU_TL_PRIMARY2);
}

// stage 2 of the primary phase, short lists: COOPERATIVE — eight lanes per pixel
// (mesh_hit_coop), 128 pixels per 1024-thread workgroup, the top of the BVH in LDS.
template <int STACK, int TEX>
__global__ void __launch_bounds__(RTU_COOP_THREADS) k_primary2c(KernelArgs a, int alone) {
    const Stamp stamp(a, RTU_TL_PRIMARY2C);
    __shared__ float4 s_nodes[RTU_LDS_NODE_F4];
    __shared__ uint32_t s_stack[RTU_COOP_GROUPS * RTU_STACK8];
    const NarrowGeom g = narrow_geom(a, 0);
    if (g.R != 8u && !alone) return;  // long list: k_primary2 takes it (alone: it was not launched — launch_all)
    const uint32_t grp = threadIdx.x >> 3;
    const bool leader = (threadIdx.x & 7u) == 0;
    // A very short list is pure latency: one wavefront per SIMD (32 rays per workgroup) so that
    // the walks do not share VALU issue slots; longer lists use all 16 wavefronts.
    const uint32_t groups = g.sum < 12288u ? 32u : (uint32_t)RTU_COOP_GROUPS;
    const uint32_t kmax = (g.nmax + groups - 1u) / groups;
    const uint32_t chunks = kmax * RTU_SHARDS;
    if (blockIdx.x >= chunks) return;  // nothing for this workgroup: do not stage the tree
    stage_nodes(a, s_nodes);
    if (grp >= groups) return;
    Counters cnt = {};
    for (uint32_t c = blockIdx.x; c < chunks; c += gridDim.x) {
        const uint32_t shard = c % RTU_SHARDS, k = c / RTU_SHARDS;
        const uint32_t ns = count_of(g.counts, shard);
        const uint32_t e = k * groups + grp;
        const bool valid = e < ns;
        uint32_t pix = 0;
        if (valid) pix = a.defer_list0[(size_t)shard * a.defer_cap0_s + e];
        if (valid && leader) RTU_BYTES(4u);
        int x, y;
        uint32_t sidx;
        pixel_of<TEX>(a, pix, x, y, sidx);
        bool deferred;
        primary_pixel<RTU_STACK8, false, false, true, TEX>(a, valid, x, y, sidx, pix, shard, s_stack + grp, cnt, deferred, leader, RTU_COOP_GROUPS, s_nodes);
    }
    flush_touched<TEX>(a, cnt, RTU_TL_PRIMARY2C);
}

// ---- one ray of one frame ------------------------------------------------------------------
// slot < nsl: shadow ray of non-ambient light `slot`; else secondary ray slot - nsl.
// Returns true if the ray was deferred (DEFER only).
template <int STACK, bool STATS, bool DEFER, bool COOP, int TEX>
__device__ __forceinline__ bool frame_ray(const KernelArgs& a, int L, int sel, uint32_t slot, uint32_t f, uint32_t* stk, Counters& cnt,
                                          bool leader = true, const uint32_t stride = 64, const float4* lds_nodes = nullptr) {
    const DevScene& s = a.scene;
    const LevelBuffers& lv = a.lv[L];
    const float4 fa = lv.fa[f];
    if (leader) RTU_BYTES(16u);
    const uint32_t info = __float_as_uint(fa.w);
    const f3 p = mk3(fa.x, fa.y, fa.z);
    Ray r;
    r.p = p;
    float tmax = RTU_BIGFLOAT;
    const bool is_shadow = slot < a.nsl;
    int lslot = -1;  // shadow rays aimed at the light itself (not at a sample of its disk): which of the scene's shadow masks applies
    const int sslot = (int)slot - (int)a.nsl;
    if (is_shadow) {
        // ---- shadow ray (lightFunctions.cpp:27-37, 43-65, 75-78; lights.h:48)
        if (!(sel & SEL_SHADOW) || !(info & RTU_FI_SH)) return false;
        if (!STATS && slot < RTU_FI_NOL_LIGHTS && ((info >> (RTU_FI_NOL_SH + slot)) & 1u)) return false;  // behind the surface: its term is +-0 (make_info)
        const int li = a.shadow_light[slot];
        const RTU_CONST RtuLight& l = as_const(s.lights)[li];
        f3 lvec = ld3(l.vec);
        if (slot < RTU_LMASK_LIGHTS) lslot = (int)slot;  // (a hard shadow ray: the light's occluder lists apply, rtu_intersect.h trace)
        if (l.type == RTU_LIGHT_DIRECT) {
            r.dir = -lvec;
        } else if (SMPD && l.size > 0) {
            lslot = -1;
            // soft shadow: one ray towards a random point of the light's disk
            const Smp smp = frame_smp<TEX>(a, L, lv.fb[f].w);
            const float sampleR = (float)rand31(smp.key, RTU_DRAW_LIGHT + 2u * (uint32_t)li) / (RTU_RAND_MAX_F / l.size);  // :47
            const float sampleTheta = (float)rand31(smp.key, RTU_DRAW_LIGHT + 2u * (uint32_t)li + 1u) / RTU_THETA_DIV;     // :48
            float sn, cs;
            portable_sincos(sampleTheta, sn, cs);
            const float offsetX = sampleR * cs, offsetY = sampleR * sn;         // :49-50
            const f3 samplePlaneNormal = norm3(lvec - p);                      // :52
            const f3 v1 = norm3(cross3(samplePlaneNormal, mk3(0, 0, 1)));      // :55
            const f3 v2 = norm3(cross3(v1, samplePlaneNormal));                // :56
            const f3 currentSamplePos = (lvec + v1 * offsetX) + v2 * offsetY;  // :58
            r.dir = norm3(currentSamplePos - p);                               // :60
            tmax = len3(p - currentSamplePos);                                 // :62
        } else {
            r.dir = norm3(lvec - p);
            tmax = len3(lvec - p);
        }
        RTU_CNT(shd);
    } else {
        // ---- secondary rays of MtlBlinn::Shade (mtlFunctions.cpp:160-229, 239, 273-283)
        if (sslot == SLOT_MAIN && (!(sel & SEL_MAIN) || !(info & RTU_FI_MAIN))) return false;
        if (sslot == SLOT_A && (!(sel & SEL_A) || !(info & RTU_FI_MAIN) || (info & RTU_FI_TIR))) return false;
        if (sslot == SLOT_C && (!(sel & SEL_C) || !(info & RTU_FI_C))) return false;
        if (sslot == SLOT_A && (sel & SEL_A_NEEDS_B)) {
            // counting variant: the Fresnel ray exists only if the refracted ray hit (:234)
            const float4 b1 = lv.fslot[((size_t)f * 3 + SLOT_MAIN) * 2 + 1];
            if (!(__float_as_uint(b1.w) & 1u)) return false;
        }
        const float4 fb = lv.fb[f], fc = lv.fc[f];
        if (leader) RTU_BYTES(32u);
        const f3 N = mk3(fb.x, fb.y, fb.z), dir = mk3(fc.x, fc.y, fc.z);
        r.dir = secondary_dir(sslot, info, dir, p, N, as_const(s.materials)[info & RTU_FI_MTL_MASK], frame_smp<TEX>(a, L, fb.w));
        RTU_CNT(sec);
    }
    Hit h;
    fresh_hit(h, tmax);
    bool deferred;
    const bool hit = trace<STACK, STATS, !STATS, DEFER, COOP, TEXD, CNTD>(s, r, is_shadow, h, stk, cnt, deferred, stride, lds_nodes, 0ull, false, lslot);
    if (DEFER && deferred) return true;
    if (!leader) return false;
    RTU_BYTES(is_shadow ? 4u : (TEXD ? 48u : 32u));
    if (is_shadow) {
        lv.fsh[(size_t)f * a.nsl + slot] = (hit && h.z > 0.0f) ? 0.0f : 1.0f;
    } else {
        int hmid = hit ? as_const(s.nodes)[h.node].material_id : -1;
        // packed: bit0 hit, bit1 front, bits 2.. material id + 1 (0 = node without material)
        uint32_t packed = (hit ? 1u : 0u) | (h.front ? 2u : 0u) | ((uint32_t)(hmid + 1) << 2);
        float4* slotp = lv.fslot + ((size_t)f * 3 + sslot) * 2;
        slotp[0] = make_float4(h.p.x, h.p.y, h.p.z, h.z);
        slotp[1] = make_float4(h.N.x, h.N.y, h.N.z, __uint_as_float(packed));
        if (TEXD) lv.fsuv[(size_t)f * 3 + sslot] = make_float4(h.uvw.x, h.uvw.y, h.uvw.z, 0.0f);
    }
    return false;
}

// stage 1: one lane per (frame, ray slot), slot-major in chunks of 64 frames
template <int STACK, bool STATS, int TEX>
__device__ __forceinline__ void trace_stage1(const KernelArgs& a, int L, int sel, int ph) {
    const Stamp stamp(a, RTU_TL_LEVEL0 + 4 * L + ((sel & SEL_A_NEEDS_B) ? 1 : 0));
    __shared__ uint32_t s_stack[STATS ? STACK * 64 : 1];
    const LevelBuffers& lv = a.lv[L];
    const uint32_t lane = threadIdx.x;
    // shadow slots: every frame of the level; secondary slots: the frames listed for them (list_frame)
    const uint32_t cntF = shard_counts(a.fcnt->n_frames[L], lv.cap_s), cntM = shard_counts(a.fcnt->n_lmain[L], lv.cap_s),
                   cntC = shard_counts(a.fcnt->n_lrefl[L], lv.cap_s);  // (each list's 64 counters: one load, for the geometry and for the chunks)
    const uint32_t chA = ((wave_max(cntF) + 63u) / 64u) * RTU_SHARDS;  // 64-frame chunks, all shards
    const uint32_t chM = ((wave_max(cntM) + 63u) / 64u) * RTU_SHARDS, chC = ((wave_max(cntC) + 63u) / 64u) * RTU_SHARDS;
    const uint32_t total = a.nsl * chA + 2u * chM + chC;
    Counters cnt = {};
    // (fetching the next chunk's record ahead, as k_consume does, was measured here: 245 -> 273 us — eight more registers cost the
    // kernel its fifth wavefront per SIMD)
    for (uint32_t c = blockIdx.x; c < total; c += gridDim.x) {
        uint32_t slot, cc;
        const uint32_t* list = nullptr;
        uint32_t counts = cntF;
        if (c < a.nsl * chA) {
            slot = c / chA;
            cc = c - slot * chA;
        } else if (c - a.nsl * chA < 2u * chM) {
            const uint32_t c2 = c - a.nsl * chA;
            slot = a.nsl + (c2 >= chM ? (uint32_t)SLOT_A : (uint32_t)SLOT_MAIN);
            cc = c2 >= chM ? c2 - chM : c2;
            list = lv.lmain;
            counts = cntM;
        } else {
            slot = a.nsl + (uint32_t)SLOT_C;
            cc = c - a.nsl * chA - 2u * chM;
            list = lv.lrefl;
            counts = cntC;
        }
        const uint32_t shard = cc % RTU_SHARDS, k = cc / RTU_SHARDS;
        const uint32_t e = k * 64u + lane;
        const uint32_t ns = count_of(counts, shard);
        const bool active = e < ns;
        const uint32_t fl = (active && list) ? list[(size_t)shard * lv.cap_s + e] : e;
        if (active && list) RTU_BYTES(4u);
        const uint32_t f = shard * lv.cap_s + fl;
        bool deferred = false;
        if (active && !(a.dbg & 2u)) deferred = frame_ray<STACK, STATS, !STATS, false, TEX>(a, L, sel, slot, f, s_stack + (STATS ? lane : 0), cnt);
        if (!STATS && !(a.dbg & 1u)) defer_push(a, ph, shard, deferred, (slot << 28) | f);
        if (deferred) RTU_BYTES(4u);
    }
    flush_counters<STATS>(a, cnt);
    flush_touched<TEX>(a, cnt, RTU_TL_LEVEL0 + 4 * L);
}
// (two kernels around one body: the occupancy hint is the fast variant's — the counting variant keeps a traversal stack per lane)
template <int STACK, bool STATS, int TEX>
__global__ void __launch_bounds__(64) RTU_OCC_TRACE k_trace(KernelArgs a, int L, int sel, int ph) { trace_stage1<STACK, STATS, TEX>(a, L, sel, ph); }
template <int STACK, int TEX>
__global__ void __launch_bounds__(64) k_trace_counting(KernelArgs a, int L, int sel, int ph) { trace_stage1<STACK, true, TEX>(a, L, sel, ph); }

// stage 2, long lists: one lane per deferred ray
template <int STACK, int TEX>
__global__ void __launch_bounds__(64) RTU_OCC_WALK k_trace2(KernelArgs a, int L, int sel, int ph, int alone) {
    const Stamp stamp(a, RTU_TL_LEVEL0 + 4 * L + 2);
    __shared__ uint32_t s_stack[STACK * 64];
    const uint32_t lane = threadIdx.x;
    const NarrowGeom g = narrow_geom(a, ph);
    if (g.R == 8u && !alone) return;  // short list: k_trace2c takes it (alone: it was not launched — launch_all)
    const uint32_t kmax = (g.nmax + 63u) / 64u;
    const uint32_t chunks = kmax * RTU_SHARDS;
    Counters cnt = {};
    const uint32_t counts = g.counts;
    for (uint32_t c = blockIdx.x; c < chunks; c += gridDim.x) {
        const uint32_t shard = c % RTU_SHARDS, k = c / RTU_SHARDS;
        const uint32_t ns = count_of(counts, shard);
        const uint32_t e = k * 64u + lane;
        if (e >= ns) continue;
        const uint32_t id = a.defer_list[(size_t)shard * a.defer_cap_s + e];
        RTU_BYTES(4u);
        frame_ray<STACK, false, false, false, TEX>(a, L, sel, id >> 28, id & 0x0FFFFFFFu, s_stack + lane, cnt);
    }
    flush_touched<TEX>(a, cnt, RTU_TL_LEVEL0 + 4 * L + 2);
}

// stage 2, short lists: cooperative, eight lanes per ray (see k_primary2c)
template <int STACK, int TEX>
__global__ void __launch_bounds__(RTU_COOP2_THREADS) k_trace2c(KernelArgs a, int L, int sel, int ph, int alone) {
    const Stamp stamp(a, RTU_TL_LEVEL0 + 4 * L + 1);
    __shared__ float4 s_nodes[RTU_LDS_NODE_F4];
    __shared__ uint32_t s_stack[RTU_COOP2_GROUPS * RTU_STACK8];
    const NarrowGeom g = narrow_geom(a, ph);
    if (g.R != 8u && !alone) return;  // long list: k_trace2 takes it (alone: it was not launched — launch_all)
    const uint32_t grp = threadIdx.x >> 3;
    const bool leader = (threadIdx.x & 7u) == 0;
    const uint32_t groups = g.sum < 12288u ? 32u : (uint32_t)RTU_COOP2_GROUPS;  // see k_primary2c
    const uint32_t kmax = (g.nmax + groups - 1u) / groups;
    const uint32_t chunks = kmax * RTU_SHARDS;
    if (blockIdx.x >= chunks) return;
    stage_nodes(a, s_nodes);
    if (grp >= groups) return;
    Counters cnt = {};
    for (uint32_t c = blockIdx.x; c < chunks; c += gridDim.x) {
        const uint32_t shard = c % RTU_SHARDS, k = c / RTU_SHARDS;
        const uint32_t ns = count_of(g.counts, shard);
        const uint32_t e = k * groups + grp;
        if (e >= ns) continue;
        const uint32_t id = a.defer_list[(size_t)shard * a.defer_cap_s + e];
        if (leader) RTU_BYTES(4u);
        frame_ray<RTU_STACK8, false, false, true, TEX>(a, L, sel, id >> 28, id & 0x0FFFFFFFu, s_stack + grp, cnt, leader, RTU_COOP2_GROUPS, s_nodes);
    }
    flush_touched<TEX>(a, cnt, RTU_TL_LEVEL0 + 4 * L + 1);
}

// ------------------------------------------------------------------------------------
// MtlBlinn::Shade combination (mtlFunctions.cpp:205-291) once every child result is
// known. st* >= 0 or RTU_CH_WHITE: that ray hit and ret* holds Shade() of the hit.
template <int TEX>
__device__ __forceinline__ f3 finalize(const DevScene& s, const RTU_CONST RtuMaterial& m, uint32_t info, f3 direct, f3 dir, f3 p,
                                       f3 N, int stMain, int stA, int stC, f3 retMain, f3 retA, f3 retC, float bz, bool bfront, f3 uvw,
                                       Smp smp) {
    const bool front = (info & RTU_FI_FRONT) != 0;
    const int mtl = (int)(info & RTU_FI_MTL_MASK);
    // environment.SampleEnvironment(direction of the ray that missed); a constant without an environment map
    auto env_at = [&](int slot) {
        return (TEXD && s.env.has_map) ? env_sample(s, secondary_dir(slot, info, dir, p, N, m, smp)) : ld3(s.environment);
    };
    f3 result = direct;
    if (info & RTU_FI_MAIN) {
        const f3 refraction = mtl_color<TEXD>(s, mtl, RTU_MAP_REFRACTION, ld3(m.refraction), uvw), absorption = ld3(m.absorption);
        const bool mainHit = stMain >= 0 || stMain == RTU_CH_WHITE;
        if (info & RTU_FI_TIR) {
            if (mainHit) result = result + absorb(RTU_BIGFLOAT, absorption) * retMain;  // :210-221 (z of a fresh HitInfo)
        } else if (mainHit) {
            Refr r = refraction_terms(dir, p, N, front, m.ior, smp, smp.on ? m.refraction_glossiness : 0.0f);
            float S = schlick(r);                                                        // :236-237
            f3 absorptionV = mk3(1, 1, 1);
            if (!bfront) absorptionV = absorb(bz, absorption);                           // :258-262
            f3 term1 = ((absorptionV * refraction) * retMain) * (float)(1.0 - (double)S);
            const bool aHit = stA >= 0 || stA == RTU_CH_WHITE;
            f3 frenselResult = aHit ? refraction * retA : env_at(SLOT_A);  // :247 / :250
            result = result + (term1 + frenselResult * S);                               // :264
        } else {
            result = result + env_at(SLOT_MAIN);  // :267
        }
    }
    if (info & RTU_FI_C) {
        const bool cHit = stC >= 0 || stC == RTU_CH_WHITE;
        if (cHit) result = result + mtl_color<TEXD>(s, mtl, RTU_MAP_REFLECTION, ld3(m.reflection), uvw) * retC;  // :286
        else result = result + env_at(SLOT_C) * ld3(m.reflection);  // :289: GetColor()
    }
    return result;
}

// The light loop of MtlBlinn::Shade (mtlFunctions.cpp:125-155) of one Shade() call with RTU_FI_SH, in light-list order, added to
// `direct` (what the call has accumulated so far: zero, or the AmbientLight term of recipe P). sh_of(j): Shadow() of non-ambient
// light j (asked only where a shadow ray was fired). Shared by k_consume / k_tail (results from the level's shadow array) and by
// the inline shading of childless Shade() calls (results in registers).
template <bool STATS, int TEX, class ShadowFn>
__device__ __forceinline__ f3 direct_light(const KernelArgs& a, uint32_t info, f3 p, f3 N, f3 uvw, f3 cam_pos, f3 direct, ShadowFn sh_of) {
    const DevScene& s = a.scene;
    const RTU_CONST RtuMaterial& m = as_const(s.materials)[info & RTU_FI_MTL_MASK];
    {
        const int mtl = (int)(info & RTU_FI_MTL_MASK);
        const f3 diffuse = mtl_color<TEXD>(s, mtl, RTU_MAP_DIFFUSE, ld3(m.diffuse), uvw);
        const f3 specular = mtl_color<TEXD>(s, mtl, RTU_MAP_SPECULAR, ld3(m.specular), uvw);
        uint32_t j = 0;  // index among the non-ambient lights
        const f3 viewDirection = norm3(cam_pos - p);  // :137 (the same value for every light)
        // A MATERIAL WITHOUT A SPECULAR COLOUR (walls, floors: `specular value="0"`) does not need its highlight: the light's term
        // is (Illuminate * N.L) * (diffuse + specular * pow(N.H, glossiness)) (:152), and with specular = +-0 in every channel the
        // second factor is `diffuse` bit for bit PROVIDED pow() is finite — +-0 * finite = +-0, and x + +-0 = x unless x is -0
        // (excluded below) —: no half vector (a square root, three divisions), no N.H, no powf (~150 instructions on this
        // device). pow(N.H, g) is finite when N.H is a number in [0, 1 + 1e-6] and 0 <= g <= 1e6. N.H = N . normalize(V + L) is a
        // NaN exactly when N is (then N.L is too and the term is NaN whatever the second factor) or when the half vector is:
        // V + L zero or NaN (the light straight behind the surface point as seen from the camera; the camera or a light AT the
        // point) — the loop below takes the literal path then (`hh > 0` fails). The counting variant always takes the literal
        // path: every test that compares the two variants checks this identity, NaN pixels included.
        const bool noHighlight = !STATS && specular.x == 0.0f && specular.y == 0.0f && specular.z == 0.0f &&
                                 m.glossiness >= 0.0f && m.glossiness <= 1e6f && __float_as_uint(diffuse.x) != 0x80000000u &&
                                 __float_as_uint(diffuse.y) != 0x80000000u && __float_as_uint(diffuse.z) != 0x80000000u;
        for (uint32_t i = 0; i < s.n_lights; i++) {
            const RTU_CONST RtuLight& l = as_const(s.lights)[i];
            const f3 intensity = ld3(l.intensity);
            if (l.type == RTU_LIGHT_AMBIENT) {
                direct = direct + diffuse * intensity;  // :132
                continue;
            }
            const f3 lvec = ld3(l.vec);
            const bool isDirect = l.type == RTU_LIGHT_DIRECT;
            const f3 ldir = isDirect ? lvec : norm3(p - lvec);           // Direction(), lights.h:49,83
            const f3 lightDirection = norm3(-ldir);                       // :138
            const f3 hsum = viewDirection + lightDirection;
            float NDotL = dot3(N, lightDirection);
            if (NDotL < 0.0f) NDotL = 0.0f;
            f3 second = diffuse;  // diffuse + specular * pow(N.H, glossiness) of a material without a specular colour
            if (!(noHighlight && dot3(hsum, hsum) > 0.0f)) {
                const f3 halfVector = norm3(hsum);  // :139
                float NDotH = dot3(N, halfVector);
                if (NDotH < 0.0f) NDotH = 0.0f;
                second = diffuse + specular * powf(NDotH, m.glossiness);
            }
            const bool behind = !STATS && j < RTU_FI_NOL_LIGHTS && ((info >> (RTU_FI_NOL_SH + j)) & 1u);  // no shadow ray was fired (make_info)
            const float sh = behind ? 1.0f : sh_of(j);
            j++;
            f3 illum;
            if (isDirect) {
                illum = intensity * sh;  // lights.h:48
            } else {
                const f3 d = lvec - p;
                illum = (intensity * (0.0f + sh)) * (1 / dot3(d, d));  // lightFunctions.cpp:78-83
            }
            direct = direct + (illum * NDotL) * second;  // :152
        }
    }
    return direct;
}

// CHILDLESS Shade() CALLS WITHOUT A FRAME. A Shade() call that fires no secondary ray (no refraction / reflection property, or
// bounce 0) is its light loop and nothing else (mtlFunctions.cpp:125-155, then `return result`): Shadow() of every non-ambient
// light, then direct_light. With the occluder lists (rtu_intersect.h mesh_shadow_cells) a hard shadow ray needs no BVH walk, so
// the lane that found the hit can settle the whole call itself: no frame record written and read back three times, no k_trace /
// k_consume round trip. shadows_inline fires the shadow rays of one call; it returns false for a lane whose ray cannot be
// settled without a walk (a mesh without a list for that light, a list entry within rounding of a bounding box, a soft shadow):
// that lane's call becomes a frame as before. Same rays, same arithmetic, same order within the call.
template <int TEX>
__device__ __forceinline__ bool shadows_inline(const KernelArgs& a, uint32_t info, f3 p, uint32_t& lit_mask, Counters& cnt) {
    const DevScene& s = a.scene;
    lit_mask = ~0u;  // bit j: Shadow() of non-ambient light j returned 1
    bool ok = true;
    if (!(info & RTU_FI_SH)) return true;
    for (uint32_t slot = 0; slot < a.nsl; slot++) {
        if (slot < RTU_FI_NOL_LIGHTS && ((info >> (RTU_FI_NOL_SH + slot)) & 1u)) continue;  // behind the surface: its term is +-0 (make_info)
        const RTU_CONST RtuLight& l = as_const(s.lights)[a.shadow_light[slot]];
        const f3 lvec = ld3(l.vec);
        Ray r;
        r.p = p;
        float tmax = RTU_BIGFLOAT;
        if (l.type == RTU_LIGHT_DIRECT) {
            r.dir = -lvec;
        } else {
            if (SMPD && l.size > 0) ok = false;  // a soft shadow: the ray aims at a sample of the light's disk (frame_ray)
            r.dir = norm3(lvec - p);
            tmax = len3(lvec - p);
        }
        Hit h;
        fresh_hit(h, tmax);
        bool deferred = false;
        if (CNTD) cnt.t_inline++;
        const bool hit = trace<1, false, true, true, false, TEXD, CNTD, true, true>(s, r, true, h, nullptr, cnt, deferred, 64, nullptr, 0ull, false,
                                                                                      slot < RTU_LMASK_LIGHTS ? (int)slot : -1);
        if (deferred) ok = false;
        if (hit && h.z > 0.0f) lit_mask &= ~(1u << slot);
    }
    return ok;
}

// One Shade() frame after its rays are traced (the body of k_consume; also used by k_tail): direct
// lighting, children, lists. Wave-uniform: all 64 lanes call it, `active` says whether the lane has
// a frame. shard: the frame's own shard; cshard: where its children go. st_out: the children.
struct FrameRec {  // a frame's three records and the Shadow() results of its first two lights, fetched ahead of time (k_consume:
    float4 fa, fb, fc;  // the next chunk's while this one is evaluated)
    float  sh0, sh1;
};
template <bool STATS, int TEX>
__device__ __forceinline__ void consume_frame(const KernelArgs& a, int L, uint32_t lane, bool active, uint32_t shard, uint32_t cshard, uint32_t fl,
                                              uint32_t f, int st_out[3], Counters& cnt, const FrameRec* pre = nullptr) {
    const DevScene& s = a.scene;
    const LevelBuffers& lv = a.lv[L];
    const bool haveNext = L + 1 < RTU_MAX_LEVELS;
    const int Ln = haveNext ? L + 1 : L;
    const LevelBuffers& nx = a.lv[Ln];
    f3 cam_pos = ld3(a.frame.cam_pos);
    st_out[0] = st_out[1] = st_out[2] = RTU_CH_NONE;
    float4 fa = make_float4(0, 0, 0, 0), fb = fa, fc = fa;
    if (active) {
        if (pre) { fa = pre->fa; fb = pre->fb; fc = pre->fc; }
        else { fa = lv.fa[f]; fb = lv.fb[f]; fc = lv.fc[f]; }
        RTU_BYTES(TEXD ? 64u : 48u);
    }
    const uint32_t info = __float_as_uint(fa.w);
    const f3 p = mk3(fa.x, fa.y, fa.z), N = mk3(fb.x, fb.y, fb.z), dir = mk3(fc.x, fc.y, fc.z);
    const RTU_CONST RtuMaterial& m = as_const(s.materials)[info & RTU_FI_MTL_MASK];
    const int bounce = (int)((info >> RTU_FI_BOUNCE_SH) & 7u);
    const Smp smp = frame_smp<TEX>(a, L, fb.w);
    uint32_t entry = 0;  // batched frames: which frame of the batch this Shade() belongs to (its camera, :137)
    if (BATD) {
        entry = __float_as_uint(fb.w);
        if (L == 0) entry /= a.batch_pixels;
        if (entry >= a.batch) entry = 0;  // inactive lanes
        cam_pos = ld3(a.cam[entry].pos);
    }
    f3 uvw = mk3(0, 0, 0);  // hInfo.uvw, textured scenes only
    if (TEXD && active) {
        const float4 t = lv.fuv[f];
        uvw = mk3(t.x, t.y, t.z);
    }

    // ---- direct lighting, mtlFunctions.cpp:125-155, in light-list order ----
    f3 direct = mk3(0, 0, 0);
    f3 ambI = mk3(0, 0, 0);
    if (GID && active && (info & RTU_FI_AMB)) {
        // the light list is MonteCarlo()'s one AmbientLight: result += diffuse * intensity on front faces (:125-132)
        const float4 t = lv.famb[f];
        ambI = mk3(t.x, t.y, t.z);
        if (info & RTU_FI_FRONT) direct = direct + mtl_color<TEXD>(s, (int)(info & RTU_FI_MTL_MASK), RTU_MAP_DIFFUSE, ld3(m.diffuse), uvw) * ambI;
    }
    if (active && (info & RTU_FI_SH)) {
        RTU_BYTES(4u * a.nsl);
        direct = direct_light<STATS, TEX>(a, info, p, N, uvw, cam_pos, direct, [&](uint32_t j) {
            return (pre && j < 2u) ? (j == 0u ? pre->sh0 : pre->sh1) : lv.fsh[(size_t)f * a.nsl + j];
        });
    }

    // ---- secondary-ray hits become frames of the next level ----
    // Every append below is one atomic per wavefront whose result the wavefront has to wait for,
    // so they are batched: ONE append for the child frames of all three slots, then the appends to
    // the slot lists and the pending list together (three round trips to L2 instead of ten).
    int st[3] = {RTU_CH_NONE, RTU_CH_NONE, RTU_CH_NONE};
    float bz = 0.0f;
    bool bfront = true;
    float4 s0[3], s1[3];
    uint32_t packed[3];
    bool spawn[3], slotAct[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        bool slotActive = false;
        if (active) {
            if (k == SLOT_MAIN) slotActive = (info & RTU_FI_MAIN) != 0;
            else if (k == SLOT_A) slotActive = (info & RTU_FI_MAIN) && !(info & RTU_FI_TIR) && (packed[SLOT_MAIN] & 1u);  // :234: the refracted ray hit
            else slotActive = (info & RTU_FI_C) != 0;
        }
        s0[k] = make_float4(0, 0, 0, 0);
        s1[k] = s0[k];
        packed[k] = 0;
        if (slotActive) {
            RTU_BYTES(32u);
            const float4* slotp = lv.fslot + ((size_t)f * 3 + k) * 2;
            s0[k] = slotp[0];
            s1[k] = slotp[1];
            packed[k] = __float_as_uint(s1[k].w);
        }
        slotAct[k] = slotActive;
        const bool hit = slotActive && (packed[k] & 1u);
        spawn[k] = hit && (int)(packed[k] >> 2) - 1 >= 0 && haveNext;
        if (k == SLOT_MAIN && hit) { bz = s0[k].w; bfront = (packed[k] & 2u) != 0; }
    }
    // child frame indices: slot 0's children of the wavefront, then slot 1's, then slot 2's
    uint32_t cfl[3];
    {
        const unsigned long long m0 = __ballot(spawn[0]), m1 = __ballot(spawn[1]), m2 = __ballot(spawn[2]);
        const uint32_t n0 = (uint32_t)__popcll(m0), n1 = (uint32_t)__popcll(m1), n2 = (uint32_t)__popcll(m2);
        uint32_t base = 0;
        if (n0 + n1 + n2) {
            if (lane == 0) base = atomicAdd(&a.fcnt->n_frames[Ln][(cshard) * RTU_CSTRIDE], n0 + n1 + n2);
            base = (uint32_t)__shfl((int)base, 0);
        }
        const unsigned long long below = (1ull << lane) - 1ull;
        cfl[0] = base + (uint32_t)__popcll(m0 & below);
        cfl[1] = base + n0 + (uint32_t)__popcll(m1 & below);
        cfl[2] = base + n0 + n1 + (uint32_t)__popcll(m2 & below);
    }
    bool wantMain[3] = {false, false, false}, wantRefl[3] = {false, false, false};
#pragma unroll
    for (int k = 0; k < 3; k++) {
        if (!slotAct[k]) continue;
        const bool hit = (packed[k] & 1u) != 0;
        const int cmid = (int)(packed[k] >> 2) - 1;
        if (!hit) st[k] = RTU_CH_MISS;
        else if (cmid < 0) st[k] = RTU_CH_WHITE;
        else if (spawn[k] && cfl[k] < nx.cap_s) {
            const uint32_t idx = cfl[k] + cshard * nx.cap_s;
            RTU_BYTES(TEXD ? 80u : 48u);  // the child's record (textured: + its uvw, read from fsuv and written to fuv)
            // the child Shade(): ray direction, hit point and normal of the secondary ray
            const f3 cdir = secondary_dir(k, info, dir, p, N, m, smp);
            Smp csmp;
            csmp.on = smp.on;
            csmp.key = smp.on ? child_key(smp.key, (uint32_t)k) : 0u;
            const f3 cp = mk3(s0[k].x, s0[k].y, s0[k].z), cN = mk3(s1[k].x, s1[k].y, s1[k].z);
            f3 cuvw = mk3(0, 0, 0);
            if (TEXD) {
                const float4 t = lv.fsuv[(size_t)f * 3 + k];
                cuvw = mk3(t.x, t.y, t.z);
                nx.fuv[idx] = t;
            }
            const uint32_t cinfo = make_info<TEX>(a, cmid, bounce - 1, (packed[k] & 2u) != 0, cdir, cp, cN, cuvw, csmp, GID && (info & RTU_FI_AMB));
            nx.fa[idx] = make_float4(cp.x, cp.y, cp.z, __uint_as_float(cinfo));
            nx.fb[idx] = make_float4(cN.x, cN.y, cN.z, __uint_as_float(BATD ? entry : csmp.key));
            nx.fc[idx] = make_float4(cdir.x, cdir.y, cdir.z, GID ? fc.w : s0[k].w);  // recipe P: the chain id travels down
            if (GID && (info & RTU_FI_AMB)) nx.famb[idx] = make_float4(ambI.x, ambI.y, ambI.z, 0.0f);
            st[k] = (int)idx;
            wantMain[k] = (cinfo & RTU_FI_MAIN) != 0;
            wantRefl[k] = (cinfo & RTU_FI_C) != 0;
        } else {
            a.fcnt->overflow = 1;  // out of frame capacity: the host re-renders with more
            st[k] = RTU_CH_MISS;
        }
    }
    const bool pending = active && (st[0] >= 0 || st[1] >= 0 || st[2] >= 0);
    st_out[0] = st[0]; st_out[1] = st[1]; st_out[2] = st[2];
    {  // the new frames join the slot lists of their level — the lists of the frames that fire a refracted / mirror ray, so that k_trace visits the secondary slots only where there is a ray —; this frame the pending list of its own
        const unsigned long long below = (1ull << lane) - 1ull;
        const unsigned long long a0 = __ballot(wantMain[0]), a1 = __ballot(wantMain[1]), a2 = __ballot(wantMain[2]);
        const unsigned long long c0 = __ballot(wantRefl[0]), c1 = __ballot(wantRefl[1]), c2 = __ballot(wantRefl[2]);
        const unsigned long long pm = __ballot(pending);
        const uint32_t na = (uint32_t)(__popcll(a0) + __popcll(a1) + __popcll(a2)), nc = (uint32_t)(__popcll(c0) + __popcll(c1) + __popcll(c2));
        uint32_t ba = 0, bc = 0, bp = 0;
        if (lane == 0) {  // independent atomics: issued back to back, one wait
            if (na) ba = atomicAdd(&a.fcnt->n_lmain[Ln][(cshard) * RTU_CSTRIDE], na);
            if (nc) bc = atomicAdd(&a.fcnt->n_lrefl[Ln][(cshard) * RTU_CSTRIDE], nc);
            if (pm) bp = atomicAdd(&a.fcnt->n_pending[L][(shard) * RTU_CSTRIDE], (uint32_t)__popcll(pm));
        }
        ba = (uint32_t)__shfl((int)ba, 0);
        bc = (uint32_t)__shfl((int)bc, 0);
        bp = (uint32_t)__shfl((int)bp, 0);
        const unsigned long long am[3] = {a0, a1, a2}, cm[3] = {c0, c1, c2};
        uint32_t oa = ba, oc = bc;
#pragma unroll
        for (int k = 0; k < 3; k++) {
            if (wantMain[k]) nx.lmain[(size_t)cshard * nx.cap_s + oa + (uint32_t)__popcll(am[k] & below)] = cfl[k];
            if (wantRefl[k]) nx.lrefl[(size_t)cshard * nx.cap_s + oc + (uint32_t)__popcll(cm[k] & below)] = cfl[k];
            RTU_BYTES((wantMain[k] ? 4u : 0u) + (wantRefl[k] ? 4u : 0u));
            oa += (uint32_t)__popcll(am[k]);
            oc += (uint32_t)__popcll(cm[k]);
        }
        if (pending) lv.fpend[(size_t)shard * lv.cap_s + bp + (uint32_t)__popcll(pm & below)] = fl;
    }
    if (!active) return;
    RTU_BYTES((pending ? 4u : 0u) + ((info & (RTU_FI_MAIN | RTU_FI_C)) ? 16u : 0u) + 16u);  // pending list, fchild, the result (pixel or fres)
    if (info & (RTU_FI_MAIN | RTU_FI_C)) lv.fchild[f] = make_int4(st[0], st[1], st[2], pending ? 1 : 0);
    if (!pending) {
        const f3 one = mk3(1, 1, 1);
        const f3 r = finalize<TEX>(s, m, info, direct, dir, p, N, st[0], st[1], st[2], one, one, one, bz, bfront, uvw, smp);
        if (L == 0) {
            if (GID) a.gi_res[((info & RTU_FI_AMB) ? 0u : a.gi_total) + (size_t)__float_as_uint(fc.w)] = make_float4(r.x, r.y, r.z, 0.0f);
            else a.out[__float_as_uint(fb.w)] = make_float4(r.x, r.y, r.z, fc.w);
        } else lv.fres[f] = make_float4(r.x, r.y, r.z, 0.0f);
    } else {
        lv.fres[f] = make_float4(direct.x, direct.y, direct.z, 0.0f);  // the direct term waits for the children
    }
}

template <bool STATS, int TEX>
__global__ void __launch_bounds__(64) RTU_OCC_CONSUME k_consume(KernelArgs a, int L) {
    const Stamp stamp(a, RTU_TL_LEVEL0 + 4 * L + 3);
    const LevelBuffers& lv = a.lv[L];
    const uint32_t lane = threadIdx.x;
    const uint32_t counts = shard_counts(a.fcnt->n_frames[L], lv.cap_s);
    const uint32_t kmax = (wave_max(counts) + 63u) / 64u;
    const uint32_t chunks = kmax * RTU_SHARDS;
    Counters cnt = {};
    // the records of the wavefront's NEXT chunk are fetched before this one is evaluated: the kernel waits for memory half of its
    // time (the records miss L2: every frame is read once), and a chunk's evaluation covers the next one's first round trip
    FrameRec nxt;
    nxt.fa = nxt.fb = nxt.fc = make_float4(0, 0, 0, 0);
    nxt.sh0 = nxt.sh1 = 1.0f;
    auto fetch = [&](uint32_t c) {
        if (c < chunks) {
            const uint32_t shard = c % RTU_SHARDS, fl = (c / RTU_SHARDS) * 64u + lane;
            if (fl < count_of(counts, shard)) {
                const uint32_t f = shard * lv.cap_s + fl;
                nxt.fa = lv.fa[f]; nxt.fb = lv.fb[f]; nxt.fc = lv.fc[f];
                if (a.nsl > 0u) nxt.sh0 = lv.fsh[(size_t)f * a.nsl];  // (of a frame without RTU_FI_SH: written by nobody, read by nobody)
                if (a.nsl > 1u) nxt.sh1 = lv.fsh[(size_t)f * a.nsl + 1u];
            }
        }
    };
    fetch(blockIdx.x);
    for (uint32_t c = blockIdx.x; c < chunks; c += gridDim.x) {
        const uint32_t shard = c % RTU_SHARDS, k = c / RTU_SHARDS;
        const uint32_t fl = k * 64u + lane;
        const bool active = fl < count_of(counts, shard);
        const FrameRec cur = nxt;
        fetch(c + gridDim.x);
        int st[3];
        consume_frame<STATS, TEX>(a, L, lane, active, shard, shard, fl, shard * lv.cap_s + fl, st, cnt, &cur);
    }
    flush_touched<TEX>(a, cnt, RTU_TL_LEVEL0 + 4 * L + 3);
}

// Frames that waited for children: combine bottom-up.
// One frame that waited for its children (the body of k_combine; also used by k_tail).
template <int TEX>
__device__ __forceinline__ void combine_frame(const KernelArgs& a, int L, uint32_t f, Counters& cnt) {
    const DevScene& s = a.scene;
    const LevelBuffers& lv = a.lv[L];
    const LevelBuffers& nx = a.lv[L + 1 < RTU_MAX_LEVELS ? L + 1 : L];
    const float4 fa = lv.fa[f];
    const uint32_t info = __float_as_uint(fa.w);
    const int4 ch = lv.fchild[f];
    const float4 fb = lv.fb[f], fc = lv.fc[f], fr = lv.fres[f];
    const f3 p = mk3(fa.x, fa.y, fa.z), N = mk3(fb.x, fb.y, fb.z), dir = mk3(fc.x, fc.y, fc.z);
    const RTU_CONST RtuMaterial& m = as_const(s.materials)[info & RTU_FI_MTL_MASK];
    const int st[3] = {ch.x, ch.y, ch.z};
    f3 ret[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        ret[k] = mk3(1, 1, 1);
        if (st[k] >= 0) {
            RTU_BYTES(16u);
            const float4 r = nx.fres[st[k]];
            ret[k] = mk3(r.x, r.y, r.z);
        }
    }
    RTU_BYTES((TEXD ? 96u : 80u) + 16u);  // fa, fchild, fb, fc, fres (+ fuv); the result
    float bz = 0.0f;
    bool bfront = true;
    if ((info & RTU_FI_MAIN) && !(info & RTU_FI_TIR)) {
        RTU_BYTES(32u);
        const float4* slotp = lv.fslot + ((size_t)f * 3 + SLOT_MAIN) * 2;
        bz = slotp[0].w;
        bfront = (__float_as_uint(slotp[1].w) & 2u) != 0;
    }
    f3 uvw = mk3(0, 0, 0);
    if (TEXD) {
        const float4 t = lv.fuv[f];
        uvw = mk3(t.x, t.y, t.z);
    }
    const f3 r = finalize<TEX>(s, m, info, mk3(fr.x, fr.y, fr.z), dir, p, N, st[0], st[1], st[2], ret[0], ret[1], ret[2], bz, bfront, uvw,
                               frame_smp<TEX>(a, L, fb.w));
    if (L == 0) {
        if (GID) a.gi_res[((info & RTU_FI_AMB) ? 0u : a.gi_total) + (size_t)__float_as_uint(fc.w)] = make_float4(r.x, r.y, r.z, 0.0f);
        else a.out[__float_as_uint(fb.w)] = make_float4(r.x, r.y, r.z, fc.w);
    } else lv.fres[f] = make_float4(r.x, r.y, r.z, 0.0f);
}

template <int TEX>
__global__ void __launch_bounds__(64) k_combine(KernelArgs a, int L) {
    const Stamp stamp(a, RTU_TL_COMBINE0 + L);
    const LevelBuffers& lv = a.lv[L];
    // only the frames k_consume listed as waiting for children
    const uint32_t counts = shard_counts(a.fcnt->n_pending[L], lv.cap_s);
    const uint32_t chunks = ((wave_max(counts) + 63u) / 64u) * RTU_SHARDS;
    Counters cnt = {};
    for (uint32_t c = blockIdx.x; c < chunks; c += gridDim.x) {
        const uint32_t shard = c % RTU_SHARDS, k = c / RTU_SHARDS;
        const uint32_t e = k * 64u + threadIdx.x;
        if (e >= count_of(counts, shard)) continue;
        const uint32_t f = shard * lv.cap_s + lv.fpend[(size_t)shard * lv.cap_s + e];
        RTU_BYTES(4u);
        combine_frame<TEX>(a, L, f, cnt);
    }
    flush_touched<TEX>(a, cnt, RTU_TL_COMBINE0 + L);
}

// THE TAIL. Deep recursion levels are often almost empty (two frames per level from level 3 on in
// the headline scene) yet each costs four dependent launches (~4 us apiece, whatever their size)
// plus a k_combine. When the previous frame showed that level Ls and below are small, the host
// launches none of their kernels but this one: ONE WAVEFRONT PER FRAME of level Ls evaluates that
// frame's whole Shade() subtree (levels Ls..max) — the same phases, the same device functions on
// the same global frame arrays, sequenced inside the wavefront: rays eight at a time (one 8-lane
// group each, cooperative BVH walk), consume with one lane per frame, children collected in LDS,
// then the combines bottom-up. Correct for any number of frames (the count is only a hint for
// the host's choice); the regular k_combine of levels < Ls follow.
#define RTU_TAIL_CAP 256  // frames of one level in one subtree: at most 3^4 = 81 for a cut at level 1, 243 in theory
template <int TEX>
__global__ void __launch_bounds__(64) k_tail(KernelArgs a, int Ls, int slot) {
    const Stamp stamp(a, slot);
    __shared__ uint32_t s_stack[8 * RTU_STACK8];
    __shared__ uint32_t s_cur[RTU_MAX_LEVELS][RTU_TAIL_CAP];  // my frames of each level
    __shared__ uint32_t s_n[RTU_MAX_LEVELS];
    __shared__ uint8_t s_pend[RTU_MAX_LEVELS][RTU_TAIL_CAP];  // the frame waits for children (fchild is only written for frames with secondary rays)
    const uint32_t lane = threadIdx.x, grp = lane >> 3;
    const bool leader = (lane & 7u) == 0;
    const int levels = a.frame.max_bounce + 1;
    const uint32_t nslots = a.nsl + 3u;
    const int sel = (int)(SEL_SHADOW | SEL_MAIN | SEL_A | SEL_C);
    const uint32_t roots = level_max_count(a, Ls) * RTU_SHARDS;  // (index within shard, shard) slots
    {   // the cut level is the host's guess from an earlier frame. A wavefront per subtree is right for a few hundred of them and
        // a hundred times too slow for a million: refuse those, the host renders the frame again level by level.
        uint32_t total = shard_count(a, Ls, lane % RTU_SHARDS);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) total += (uint32_t)__shfl_xor((int)total, off);
        if (total > RTU_TAIL_DECLINE && !(a.dbg & 128u)) {
            if (blockIdx.x == 0 && lane == 0) a.fcnt->tail_declined = 1;
            return;
        }
    }
    Counters cnt = {};
    for (uint32_t c = blockIdx.x; c < roots; c += gridDim.x) {
        const uint32_t shard = c % RTU_SHARDS, rfl = c / RTU_SHARDS;
        if (rfl >= shard_count(a, Ls, shard)) continue;  // wave-uniform
        uint32_t n = 1;
        if (lane == 0) {
            s_cur[Ls][0] = shard * a.lv[Ls].cap_s + rfl;
            s_n[Ls] = 1;
        }
        int last = Ls;
        for (int L = Ls; L < levels && n > 0; L++) {
            last = L;
            __threadfence();  // s_cur and the frame records of this level are written
            // ---- the rays of my frames: eight at a time, one 8-lane group each
            const uint32_t total = n * nslots;
            for (uint32_t base = 0; base < total; base += 8u) {
                const uint32_t item = base + grp;
                if (item < total) {
                    const uint32_t i = item / nslots, slot = item - i * nslots;
                    frame_ray<RTU_STACK8, false, false, true, TEX>(a, L, sel, slot, s_cur[L][i], s_stack + grp, cnt, leader, 8u, nullptr);
                }
            }
            __threadfence();  // shadow results and secondary hits are visible to the consuming lanes
            // ---- consume: one lane per frame, 64 at a time; the children become my frames of the next level
            uint32_t nn = 0;
            for (uint32_t b0 = 0; b0 < n; b0 += 64u) {
                const uint32_t i = b0 + lane;
                const bool active = i < n;
                const uint32_t f = active ? s_cur[L][i] : 0u;
                int st[3];
                consume_frame<false, TEX>(a, L, lane, active, shard, shard, f - shard * a.lv[L].cap_s, f, st, cnt);
                if (active) s_pend[L][i] = (st[0] >= 0 || st[1] >= 0 || st[2] >= 0) ? 1 : 0;
                if (L + 1 < levels) {
#pragma unroll
                    for (int k = 0; k < 3; k++) {
                        const bool has = active && st[k] >= 0;
                        const unsigned long long m = __ballot(has);
                        const uint32_t at = nn + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
                        if (has) {
                            if (at < (uint32_t)RTU_TAIL_CAP) s_cur[L + 1][at] = (uint32_t)st[k];
                            else a.fcnt->overflow = 1;  // cannot happen below 3^5 frames; reported like any overflow
                        }
                        nn += (uint32_t)__popcll(m);
                    }
                }
            }
            if (nn > (uint32_t)RTU_TAIL_CAP) nn = RTU_TAIL_CAP;
            if (lane == 0 && L + 1 < levels) s_n[L + 1] = nn;
            n = nn;
        }
        // ---- combine bottom-up what waited for children (levels Ls .. last-1)
        for (int L = last - 1; L >= Ls; L--) {
            __threadfence();
            const uint32_t nL = s_n[L];
            for (uint32_t i = lane; i < nL; i += 64u)
                if (s_pend[L][i]) combine_frame<TEX>(a, L, s_cur[L][i], cnt);
        }
        __threadfence();
    }
    flush_touched<TEX>(a, cnt, slot);
}

// ---- recipe P: MonteCarlo() of RenderFunctions.cpp:549-590 unrolled over the chain ---------------------
// Depth k of a chain is shaded after depth k + 1: the AmbientLight MonteCarlo(h_k, 4 - k) appends has the
// intensity c = Shade(h_k+1, its own AmbientLight) + Shade(h_k+1, lights) (:568-570), the environment along
// the gather ray if it missed (:575), or 0.1 at the last bounce (:584). This kernel computes c for every
// chain that has a hit of depth a.gi_depth and appends the two Shade() trees of that hit as level-0 frames.
// inline_ok: the fast variant (the counting variant materialises every Shade() call as a frame: its counters are the reference's).
#ifndef RTU_OCC_GI_ROOTS
// k_gi_roots settles the childless Shade() calls of recipe P itself (shadow rays, light loop): left alone it takes 140 VGPRs and fits three
// wavefronts per SIMD (2578 us per launch of 33 M chains); four: 2191 us; five (41 registers spilled): 3116 us
#define RTU_OCC_GI_ROOTS __attribute__((amdgpu_waves_per_eu(4, 4)))
#endif
template <int TEX>
__global__ void __launch_bounds__(64) RTU_OCC_GI_ROOTS k_gi_roots(KernelArgs a, int inline_ok) {
    const DevScene& s = a.scene;
    const uint32_t lane = threadIdx.x;
    const uint32_t k = a.gi_depth;
    const uint32_t chunks = (a.gi_total + 63u) / 64u;
    Counters cnt = {};
    for (uint32_t c = blockIdx.x; c < chunks; c += gridDim.x) {
        const uint32_t chain = c * 64u + lane;
        const uint32_t shard = c % RTU_SHARDS;
        const size_t hb = (size_t)k * 4u * a.gi_total + chain;
        float4 hA = make_float4(0, 0, 0, 0), hB = hA, hC = hA, hD = hA;
        bool want = false;
        if (chain < a.gi_total) {
            hB = a.gi_h[hb + a.gi_total];
            want = (__float_as_uint(hB.w) & 1u) != 0;
            RTU_BYTES(16u);
        }
        if (want) {
            hA = a.gi_h[hb];
            hC = a.gi_h[hb + 2u * (size_t)a.gi_total];
            if (TEXD) hD = a.gi_h[hb + 3u * (size_t)a.gi_total];
            RTU_BYTES(TEXD ? 48u : 32u);
        }
        const uint32_t pk = __float_as_uint(hB.w);
        const int mid = (int)(pk >> 2) - 1;
        f3 amb = mk3(0.1f, 0.1f, 0.1f);  // :584
        if (want && k < (uint32_t)RTU_GI_BOUNCES) {
            const size_t hn = (size_t)(k + 1u) * 4u * a.gi_total + chain;
            const uint32_t npk = __float_as_uint(a.gi_h[hn + a.gi_total].w);
            RTU_BYTES(16u + ((npk & 1u) ? ((int)(npk >> 2) - 1 < 0 ? 0u : 32u) : 16u));  // the depth below: its hit flag, then its two results or its ray
            if (npk & 1u) {
                if ((int)(npk >> 2) - 1 < 0) {
                    amb = (mk3(0, 0, 0) + mk3(1, 1, 1)) + mk3(1, 1, 1);  // a node without material shades white, twice
                } else {
                    const float4 ra = a.gi_res[chain], rd = a.gi_res[(size_t)a.gi_total + chain];
                    amb = (mk3(0, 0, 0) + mk3(ra.x, ra.y, ra.z)) + mk3(rd.x, rd.y, rd.z);  // :569-570
                }
            } else {
                const float4 nd = a.gi_h[hn + 2u * (size_t)a.gi_total];
                amb = mk3(0, 0, 0) + ((TEXD && s.env.has_map) ? env_sample(s, mk3(nd.x, nd.y, nd.z)) : ld3(s.environment));  // :575
            }
        }
        if (want && mid < 0) {  // no material at this hit: both trees are white, nothing to trace
            a.gi_res[chain] = make_float4(1, 1, 1, 0);
            a.gi_res[(size_t)a.gi_total + chain] = make_float4(1, 1, 1, 0);
            RTU_BYTES(32u);
            want = false;
        }
        const f3 p = mk3(hA.x, hA.y, hA.z), N = mk3(hB.x, hB.y, hB.z), dir = mk3(hC.x, hC.y, hC.z), uvw = mk3(hD.x, hD.y, hD.z);
        const bool front = (pk & 2u) != 0;
        Smp sd, sa;
        sd.on = sa.on = true;
        sd.key = __float_as_uint(hC.w);                        // the tree lit by the scene's lights (:570, :135)
        sa.key = child_key(sd.key, RTU_SLOT_AMBIENT_TREE);     // the tree lit by the AmbientLight (:569, :134)
        uint32_t ia = 0, id = 0;
        if (want) {
            ia = make_info<TEX>(a, mid, a.frame.max_bounce, front, dir, p, N, uvw, sa, true);
            id = make_info<TEX>(a, mid, a.frame.max_bounce, front, dir, p, N, uvw, sd, false);
        }
        // A childless Shade() call (no refraction / reflection property: every material of config 5's scene) needs no frame:
        //  * the tree lit by MonteCarlo()'s AmbientLight is `diffuse * intensity` on a front face (mtlFunctions.cpp:125-132) — what
        //    consume_frame computes for an RTU_FI_AMB frame without children;
        //  * the tree lit by the scene's lights is its light loop, settled here when every shadow ray can be (shadows_inline).
        bool wantA = want, wantD = want;
        if (inline_ok && want && !(ia & (RTU_FI_MAIN | RTU_FI_C))) {
            f3 direct = mk3(0, 0, 0);
            if (ia & RTU_FI_FRONT) direct = direct + mtl_color<TEXD>(s, mid, RTU_MAP_DIFFUSE, ld3(as_const(s.materials)[mid].diffuse), uvw) * amb;
            a.gi_res[chain] = make_float4(direct.x, direct.y, direct.z, 0.0f);
            RTU_BYTES(16u);
            wantA = false;
        }
        if (inline_ok && __any(want && !(id & (RTU_FI_MAIN | RTU_FI_C)))) {
            const bool tryD = want && !(id & (RTU_FI_MAIN | RTU_FI_C));
            uint32_t lit = ~0u;
            bool ok = false;
            if (tryD) ok = shadows_inline<TEX>(a, id, p, lit, cnt);
            if (tryD && ok) {
                f3 direct = mk3(0, 0, 0);
                if (id & RTU_FI_SH) direct = direct_light<false, TEX>(a, id, p, N, uvw, ld3(a.frame.cam_pos), direct, [&](uint32_t j) { return ((lit >> j) & 1u) ? 1.0f : 0.0f; });
                a.gi_res[(size_t)a.gi_total + chain] = make_float4(direct.x, direct.y, direct.z, 0.0f);
                RTU_BYTES(16u);
                wantD = false;
            }
        }
        const uint32_t fa_idx = append_root<TEX>(a, wantA, shard, ia, p, N, sa.key, dir, __uint_as_float(chain), uvw, cnt);
        if (wantA && fa_idx != ~0u) { a.lv[0].famb[fa_idx] = make_float4(amb.x, amb.y, amb.z, 0.0f); RTU_BYTES(16u); }
        append_root<TEX>(a, wantD, shard, id, p, N, sd.key, dir, __uint_as_float(chain), uvw, cnt);
    }
    flush_touched<TEX>(a, cnt, RTU_TL_GI_ROOTS);
}

// NODE-LEVEL BOUNDS of primary rays (recipe W): per camera of the launch and per node, the rectangle of pixels whose
// primary ray can come within the cull margin of the node's world-space bound — the eight corners of the bound, widened by
// delta = 1e-4 * max(scene scale, |camera|), projected through the camera in binary64, two pixels of slack on every side.
// A corner at or behind the camera plane makes the rectangle the whole image. One lane per (camera, node).
__device__ __forceinline__ void node_rects_body(const KernelArgs& a, uint32_t entries) {
    const uint32_t n_nodes = a.scene.n_nodes;
    for (uint32_t i = threadIdx.x; i < entries * n_nodes; i += blockDim.x) {
        const uint32_t e = i / n_nodes, k = i - e * n_nodes;
        const DevNode& n = a.scene.nodes[k];
        int4 r = make_int4(0, 0, a.frame.width, a.frame.height);
        if (n.obj_type != RTU_OBJ_NONE && a.scene.node_bounds) {
            const float* cp = a.frame_batch ? a.cam[e].pos : a.frame.cam_pos;
            const float* co = a.frame_batch ? a.cam[e].origin : a.frame.origin;
            const float* cu = a.frame_batch ? a.cam[e].u : a.frame.u;
            const float* cv = a.frame_batch ? a.cam[e].v : a.frame.v;
            const double P[3] = {cp[0], cp[1], cp[2]};
            const double U[3] = {cu[0], cu[1], cu[2]}, V[3] = {cv[0], cv[1], cv[2]};
            const double O[3] = {co[0] - P[0], co[1] - P[1], co[2] - P[2]};  // image-plane origin seen from the camera
            // w = c * (O + a U + b V): Cramer's rule on [U V O]
            const double det = U[0] * (V[1] * O[2] - V[2] * O[1]) - V[0] * (U[1] * O[2] - U[2] * O[1]) + O[0] * (U[1] * V[2] - U[2] * V[1]);
            const double pm = fmax(fabs(P[0]), fmax(fabs(P[1]), fabs(P[2])));
            const double delta = 1e-4 * fmax((double)a.scene.wscale, pm);
            double amin = 1e300, amax = -1e300, bmin = 1e300, bmax = -1e300;
            bool all_front = det != 0.0;
            for (int c = 0; c < 8 && all_front; c++) {
                const double w[3] = {((c & 1) ? (double)n.wmax[0] + delta : (double)n.wmin[0] - delta) - P[0],
                                     ((c & 2) ? (double)n.wmax[1] + delta : (double)n.wmin[1] - delta) - P[1],
                                     ((c & 4) ? (double)n.wmax[2] + delta : (double)n.wmin[2] - delta) - P[2]};
                const double da = w[0] * (V[1] * O[2] - V[2] * O[1]) - V[0] * (w[1] * O[2] - w[2] * O[1]) + O[0] * (w[1] * V[2] - w[2] * V[1]);
                const double db = U[0] * (w[1] * O[2] - w[2] * O[1]) - w[0] * (U[1] * O[2] - U[2] * O[1]) + O[0] * (U[1] * w[2] - U[2] * w[1]);
                const double dc = U[0] * (V[1] * w[2] - V[2] * w[1]) - V[0] * (U[1] * w[2] - U[2] * w[1]) + w[0] * (U[1] * V[2] - U[2] * V[1]);
                const double cc = dc / det;  // depth along the view axis in units of the image-plane distance
                if (!(cc > 1e-3)) { all_front = false; break; }
                const double pa = da / dc, pb = db / dc;
                amin = fmin(amin, pa); amax = fmax(amax, pa);
                bmin = fmin(bmin, pb); bmax = fmax(bmax, pb);
            }
            if (all_front && amin <= amax && bmin <= bmax) {
                // pixel x lies on the ray through a = x + 0.5
                const double W = a.frame.width, H = a.frame.height;
                const double x0 = floor(fmin(fmax(amin - 2.5, -1.0), W + 1.0)), x1 = ceil(fmin(fmax(amax + 1.5, -1.0), W + 1.0));
                const double y0 = floor(fmin(fmax(bmin - 2.5, -1.0), H + 1.0)), y1 = ceil(fmin(fmax(bmax + 1.5, -1.0), H + 1.0));
                r = make_int4((int)fmax(x0, 0.0), (int)fmax(y0, 0.0), (int)fmin(x1, W), (int)fmin(y1, H));
            }
        }
        a.node_rects[i] = r;
    }
}

// COVERAGE MASKS (KernelArgs::cover): one thread per triangle of a mesh node, per camera of the launch. The triangle's own bounding
// box in world space (its vertices through the node's chain of transformations, computed once at upload), widened by the cull margin
// for this camera, is projected like the node bound in k_node_rects; every 8x8 tile its pixel rectangle (two pixels of slack) touches
// gets its bit. A primary ray that hits the triangle has its hit point in that box, so its pixel is in a marked tile.
__device__ __forceinline__ void mesh_cover_body(const KernelArgs& a, uint32_t entries, uint32_t by) {
    const uint32_t e = by / a.scene.n_cover, c = by % a.scene.n_cover;
    if (e >= entries || blockIdx.x * 256u >= a.scene.cover_nf[c]) return;  // (the grid is as wide as the widest role of k_prelude)
    uint32_t* m = a.cover + ((size_t)e * (a.scene.n_cover + a.scene.n_pcover) + c) * (1u + a.cover_words);
    // the 256 triangles of a workgroup mark a copy of the mask in LDS; its non-zero words are then added to the mask in memory
    // (device-scope atomics are served one after the other per cache line: 6320 triangles x 6 tiles straight into 32 lines took 45 us)
    extern __shared__ uint32_t s_mask[];
    for (uint32_t i = threadIdx.x; i < a.cover_words; i += 256u) s_mask[i] = 0u;
    __syncthreads();
    const uint32_t face = blockIdx.x * 256u + threadIdx.x;
    bool unusable = false;
    if (face < a.scene.cover_nf[c]) {
    // binary32 throughout: its rounding (1e-6 of the coordinates, a few thousandths of a pixel) is far inside the two pixels of slack
    const float* cp = a.frame_batch ? a.cam[e].pos : a.frame.cam_pos;
    const float* co = a.frame_batch ? a.cam[e].origin : a.frame.origin;
    const float* cu = a.frame_batch ? a.cam[e].u : a.frame.u;
    const float* cv = a.frame_batch ? a.cam[e].v : a.frame.v;
    const f3 P = ld3(cp), U = ld3(cu), V = ld3(cv), O = ld3(co) - P;
    // w = c (O + a U + b V): Cramer's rule on [U V O]; rows of the adjugate so that a corner costs three dot products
    const f3 VxO = cross3(V, O), OxU = cross3(O, U), UxV = cross3(U, V);
    const float det = dot3(U, VxO);
    const float4 blo = a.scene.cover_box[c][2u * face], bhi = a.scene.cover_box[c][2u * face + 1u];  // the triangle's world-space box (upload)
    const f3 lo = mk3(blo.x, blo.y, blo.z), hi = mk3(bhi.x, bhi.y, bhi.z);
    const float pm = fmaxf(fabsf(P.x), fmaxf(fabsf(P.y), fabsf(P.z)));
    const float big = fmaxf(fmaxf(fabsf(lo.x), fabsf(hi.x)), fmaxf(fmaxf(fabsf(lo.y), fabsf(hi.y)), fmaxf(fabsf(lo.z), fabsf(hi.z))));
    const float wid = 1e-4f * fmaxf(a.scene.wscale, pm) + 1e-5f * big;  // the cull margin for this camera + the rounding of the chain
    float amin = 3e38f, amax = -3e38f, bmin = 3e38f, bmax = -3e38f;
    bool ok = det != 0.0f && wid == wid;
    const float rdet = 1.0f / det;
    for (int cn = 0; cn < 8 && ok; cn++) {
        const f3 w = mk3(((cn & 1) ? hi.x + wid : lo.x - wid) - P.x, ((cn & 2) ? hi.y + wid : lo.y - wid) - P.y, ((cn & 4) ? hi.z + wid : lo.z - wid) - P.z);
        const float da = dot3(w, VxO), db = dot3(w, OxU), dc = dot3(w, UxV);
        if (!(dc * rdet > 1e-3f)) { ok = false; break; }  // at or behind the camera plane
        const float rc = 1.0f / dc;
        const float pa = da * rc, pb = db * rc;
        amin = fminf(amin, pa); amax = fmaxf(amax, pa);
        bmin = fminf(bmin, pb); bmax = fmaxf(bmax, pb);
    }
    const float W = (float)a.frame.width, H = (float)a.frame.height;
    if (!ok || amin != amin || bmin != bmin || amax != amax || bmax != bmax) {
        unusable = true;  // at or behind the camera plane (or NaN): this mask decides nothing
    } else if (amax + 1.5f >= 0.0f && bmax + 1.5f >= 0.0f && amin - 2.5f <= W && bmin - 2.5f <= H) {  // (else: off the image)
        const int x0 = (int)fmaxf(floorf(amin - 2.5f), 0.0f), x1 = (int)fminf(ceilf(amax + 1.5f), W);  // pixel x lies on the ray through a = x + 0.5
        const int y0 = (int)fmaxf(floorf(bmin - 2.5f), 0.0f), y1 = (int)fminf(ceilf(bmax + 1.5f), H);
        if (x1 > x0 && y1 > y0) {
            const int tx0 = x0 >> 3, tx1 = (x1 - 1) >> 3, ty0 = y0 >> 3, ty1 = (y1 - 1) >> 3;
            if ((long long)(tx1 - tx0 + 1) * (ty1 - ty0 + 1) > 4096) unusable = true;  // a triangle as large as the image: no mask for this mesh
            else
                for (int ty = ty0; ty <= ty1; ty++)
                    for (int tx = tx0; tx <= tx1; tx++) {
                        const uint32_t tile = (uint32_t)ty * a.tiles_xf + (uint32_t)tx;
                        atomicOr(&s_mask[tile >> 5], 1u << (tile & 31u));
                    }
        }
    }
    }
    if (unusable) m[0] = 1u;
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < a.cover_words; i += 256u) {
        const uint32_t v = s_mask[i];
        if (v) atomicOr(&m[1u + i], v);
    }
}

// COVERAGE MASK OF A PLANE node (slots n_cover.. of KernelArgs::cover): one thread per 8x8 tile of the whole image, per camera of
// the launch and masked plane. A Plane is the unit square of its node (objFunctions.cpp:107-140: the hit must lie strictly
// inside (-1,1)^2 of the node's z = 0); the screen RECTANGLE of a square seen at an angle is mostly air (the headline's ground
// is a diamond: half of its rectangle). So the square itself is projected: its four world-space corners (upload, binary64), each
// pushed outwards along both of its edges by the cull margin for this camera (delta = 1e-4 max(scene scale, |camera|): a hundred
// times the rounding of the reference's own ray / square arithmetic, as for the node bounds) divided by the sine of the corner's
// angle, through the camera in binary64 like the node bound's corners; a tile — widened by two pixels on every side — is marked
// unless a separating line is found: the quadrilateral's bounding box, or one of its four edges with the whole tile beyond it.
// A corner at or behind the camera plane, a sliver of a corner (sine < 0.05) or a NaN makes the mask unusable (word 0 = 1).
__device__ __forceinline__ void plane_cover_body(const KernelArgs& a, uint32_t entries, uint32_t by) {
    const DevScene& s = a.scene;
    const uint32_t e = by / s.n_pcover, c = by % s.n_pcover;
    if (e >= entries || blockIdx.x * 256u >= a.cover_words * 32u) return;
    uint32_t* m = a.cover + ((size_t)e * (s.n_cover + s.n_pcover) + s.n_cover + c) * (1u + a.cover_words);
    __shared__ double s_v[4][2];
    __shared__ int s_bad;
    if (threadIdx.x == 0) s_bad = 0;
    __syncthreads();
    if (threadIdx.x < 4u) {
        const uint32_t i = threadIdx.x;
        const float* cp = a.frame_batch ? a.cam[e].pos : a.frame.cam_pos;
        const float* co = a.frame_batch ? a.cam[e].origin : a.frame.origin;
        const float* cu = a.frame_batch ? a.cam[e].u : a.frame.u;
        const float* cv = a.frame_batch ? a.cam[e].v : a.frame.v;
        const double P[3] = {cp[0], cp[1], cp[2]};
        const double U[3] = {cu[0], cu[1], cu[2]}, V[3] = {cv[0], cv[1], cv[2]};
        const double O[3] = {co[0] - P[0], co[1] - P[1], co[2] - P[2]};
        const double det = U[0] * (V[1] * O[2] - V[2] * O[1]) - V[0] * (U[1] * O[2] - U[2] * O[1]) + O[0] * (U[1] * V[2] - U[2] * V[1]);
        const double pm = fmax(fabs(P[0]), fmax(fabs(P[1]), fabs(P[2])));
        const double delta = 1e-4 * fmax((double)s.wscale, pm);
        const float (*q)[3] = s.pcover_quad[c];
        const uint32_t ip = (i + 3u) & 3u, in = (i + 1u) & 3u;
        double w[3], ua[3], ub[3], la = 0, lb = 0, big = 0;
        for (int k = 0; k < 3; k++) {
            ua[k] = (double)q[i][k] - (double)q[ip][k];  // away from the previous corner
            ub[k] = (double)q[i][k] - (double)q[in][k];  // away from the next corner
            la += ua[k] * ua[k];
            lb += ub[k] * ub[k];
            big = fmax(big, fabs((double)q[i][k]));
        }
        la = sqrt(la);
        lb = sqrt(lb);
        double dotab = 0;
        for (int k = 0; k < 3; k++) { ua[k] /= la; ub[k] /= lb; dotab += ua[k] * ub[k]; }
        const double sine = sqrt(fmax(0.0, 1.0 - dotab * dotab));
        bool ok = det != 0.0 && la > 0.0 && lb > 0.0 && sine >= 0.05;
        const double push = (delta + 1e-5 * big) / fmax(sine, 0.05);
        for (int k = 0; k < 3; k++) w[k] = (double)q[i][k] + (ua[k] + ub[k]) * push - P[k];
        const double da = w[0] * (V[1] * O[2] - V[2] * O[1]) - V[0] * (w[1] * O[2] - w[2] * O[1]) + O[0] * (w[1] * V[2] - w[2] * V[1]);
        const double db = U[0] * (w[1] * O[2] - w[2] * O[1]) - w[0] * (U[1] * O[2] - U[2] * O[1]) + O[0] * (U[1] * w[2] - U[2] * w[1]);
        const double dc = U[0] * (V[1] * w[2] - V[2] * w[1]) - V[0] * (U[1] * w[2] - U[2] * w[1]) + w[0] * (U[1] * V[2] - U[2] * V[1]);
        const double cc = dc / det;  // depth along the view axis in units of the image-plane distance
        if (!(cc > 1e-3)) ok = false;
        const double pa = da / dc, pb = db / dc;  // pixel x lies on the ray through a = x + 0.5
        if (!(pa == pa) || !(pb == pb) || fabs(pa) > 1e9 || fabs(pb) > 1e9) ok = false;
        s_v[i][0] = pa;
        s_v[i][1] = pb;
        if (!ok) s_bad = 1;
    }
    __syncthreads();
    if (s_bad) {
        if (blockIdx.x == 0 && threadIdx.x == 0) m[0] = 1u;  // this mask decides nothing
        return;
    }
    const uint32_t tile = blockIdx.x * 256u + threadIdx.x;
    const uint32_t tiles_y = (uint32_t)((a.frame.height + 7) / 8);
    bool mark = false;
    if (tile < a.tiles_xf * tiles_y) {
        const uint32_t ty = tile / a.tiles_xf, tx = tile - ty * a.tiles_xf;
        // the tile's pixels x0 .. x0 + 7 see the rays a in [x0 + 0.5, x0 + 7.5]; two pixels of slack on every side
        const double rx0 = (double)(tx * 8u) + 0.5 - 2.0, rx1 = (double)(tx * 8u) + 7.5 + 2.0;
        const double ry0 = (double)(ty * 8u) + 0.5 - 2.0, ry1 = (double)(ty * 8u) + 7.5 + 2.0;
        double amin = s_v[0][0], amax = amin, bmin = s_v[0][1], bmax = bmin, area2 = 0;
        for (int i = 0; i < 4; i++) {
            amin = fmin(amin, s_v[i][0]); amax = fmax(amax, s_v[i][0]);
            bmin = fmin(bmin, s_v[i][1]); bmax = fmax(bmax, s_v[i][1]);
            const int j = (i + 1) & 3;
            area2 += s_v[i][0] * s_v[j][1] - s_v[j][0] * s_v[i][1];
        }
        mark = !(amax < rx0 || amin > rx1 || bmax < ry0 || bmin > ry1);
        if (mark && fabs(area2) > 1e-6) {  // (an edge-on square has no inside: its bounding box is all there is)
            const double sgn = area2 > 0 ? 1.0 : -1.0;
            for (int i = 0; i < 4 && mark; i++) {
                const int j = (i + 1) & 3;
                const double ex = s_v[j][0] - s_v[i][0], ey = s_v[j][1] - s_v[i][1];
                // inside is where sgn * cross(edge, q - v_i) >= 0: the tile is beyond this edge when all four of its corners are outside
                const double c00 = sgn * (ex * (ry0 - s_v[i][1]) - ey * (rx0 - s_v[i][0])), c10 = sgn * (ex * (ry0 - s_v[i][1]) - ey * (rx1 - s_v[i][0]));
                const double c01 = sgn * (ex * (ry1 - s_v[i][1]) - ey * (rx0 - s_v[i][0])), c11 = sgn * (ex * (ry1 - s_v[i][1]) - ey * (rx1 - s_v[i][0]));
                if (c00 < 0 && c10 < 0 && c01 < 0 && c11 < 0) mark = false;
            }
        }
    }
    const unsigned long long bits = __ballot(mark);
    if ((threadIdx.x & 63u) == 0u) {  // a wavefront's 64 tiles are two whole words of the mask
        const uint32_t w0 = (blockIdx.x * 256u + threadIdx.x) >> 5;
        if (w0 < a.cover_words) m[1u + w0] = (uint32_t)bits;
        if (w0 + 1u < a.cover_words) m[2u + w0] = (uint32_t)(bits >> 32);
    }
}

// THE PRELUDE of a recipe-W launch sequence in ONE launch (three launches of a few microseconds each were 6 % of a single frame):
// blockIdx.y selects the role — the coverage mask of (camera, masked mesh) from the triangle boxes, of (camera, masked plane)
// from the square's corners, or, in the last row, the screen rectangles of the node-level bounds (one block). The roles do not
// depend on each other; k_tile_occ, which needs all three, is the next launch.
__global__ void __launch_bounds__(256) k_prelude(KernelArgs a, uint32_t entries) {
    const uint32_t nm = a.cover ? entries * a.scene.n_cover : 0u, np = a.cover ? entries * a.scene.n_pcover : 0u;
    if (blockIdx.y < nm) mesh_cover_body(a, entries, blockIdx.y);
    else if (blockIdx.y < nm + np) plane_cover_body(a, entries, blockIdx.y - nm);
    else if (blockIdx.x == 0u) node_rects_body(a, entries);
}

// TILE OCCUPANCY (KernelArgs::occ): one thread per 8x8 tile of the shard, per camera of the launch: is any valid pixel of the tile
// inside the screen rectangle of an object node (k_node_rects) and, where that node is a mesh with a usable coverage mask
// (k_mesh_cover), in a tile the mask marks? Exactly the pixels primary_pixel's own rectangle / mask test would keep: a tile with
// a zero bit is one whose wavefront would write the background for every pixel. Needs n_nodes <= 64 (the host checks).
__global__ void __launch_bounds__(64) k_tile_occ(KernelArgs a, uint32_t entries) {
    const uint32_t e = blockIdx.y;
    if (e >= entries) return;
    const DevScene& s = a.scene;
    const uint32_t tiles = a.tiles_per_image;
    const uint32_t tile = blockIdx.x * 64u + threadIdx.x;
    bool occ = false;
    if (tile < tiles) {
        const uint32_t band_local = tile / a.tiles_x, tx = tile - band_local * a.tiles_x;
        const int x0 = (int)(tx * 8u), y0 = (int)((band_local * (uint32_t)a.frame.shard_count + (uint32_t)a.frame.shard_rank) * RTU_BAND_ROWS);
        const int x1 = x0 + 8 < a.frame.width ? x0 + 8 : a.frame.width, y1 = y0 + 8 < a.frame.height ? y0 + 8 : a.frame.height;
        const uint32_t gtile = (uint32_t)(y0 >> 3) * a.tiles_xf + tx;
        for (uint32_t k = 0; k < s.n_nodes && k < 64u; k++) {
            if (!((s.obj_mask >> k) & 1ull)) continue;
            const int4 r = a.node_rects[(size_t)e * s.n_nodes + k];
            if (!(x0 < r.z && x1 > r.x && y0 < r.w && y1 > r.y)) continue;  // no pixel of the tile inside the rectangle
            bool masked_out = false;
            if (a.cover)
                for (uint32_t c = 0; c < s.n_cover + s.n_pcover; c++)
                    if ((uint32_t)(c < s.n_cover ? s.cover_node[c] : s.pcover_node[c - s.n_cover]) == k) {
                        const uint32_t* m = a.cover + ((size_t)e * (s.n_cover + s.n_pcover) + c) * (1u + a.cover_words);
                        if (m[0] == 0u && !((m[1u + (gtile >> 5)] >> (gtile & 31u)) & 1u)) masked_out = true;
                    }
            if (!masked_out) occ = true;
        }
    }
    const unsigned long long bits = __ballot(occ);
    if (threadIdx.x == 0) {
        uint32_t* o = const_cast<uint32_t*>(a.occ) + (size_t)e * a.occ_words + 2u * blockIdx.x;
        o[0] = (uint32_t)bits;
        o[1] = (uint32_t)(bits >> 32);
        if (bits) atomicAdd(&a.fcnt->occ_tiles[((blockIdx.x + 7u * blockIdx.y) % RTU_SHARDS) * RTU_CSTRIDE], (uint32_t)__popcll(bits));  // (for the host: launch_all's grid of k_primary)
    }
}

// One kernel of the sequence; `slot` is its timeline / counter-table slot. With a probe on that slot the launch is
// bracketed by HIP events on the launch stream (bench.py: the dominant kernel's duration inside the timed region).
#define RTU_LAUNCH_ON(strm_, kslot_, kernel, grid, blk, ...)                                  \
    do {                                                                                     \
        const bool probed_ = probe && probe->slot == (kslot_);                               \
        if (a.host_launches) a.host_launches[(kslot_)]++;                                    \
        if (probed_) (void)hipEventRecord((hipEvent_t)probe->ev0, (strm_));                  \
        hipLaunchKernelGGL(kernel, grid, blk, 0, (strm_), __VA_ARGS__);                      \
        if (probed_) { (void)hipEventRecord((hipEvent_t)probe->ev1, (strm_)); if (probe->recorded) *probe->recorded = 1; } \
    } while (0)
#define RTU_LAUNCH(kslot_, kernel, grid, blk, ...) RTU_LAUNCH_ON(stream, kslot_, kernel, grid, blk, __VA_ARGS__)

template <int STACK, int TEX>
int launch_all(const KernelArgs& a, uint32_t n_tiles, bool stats, hipStream_t stream, int mode = RTU_LAUNCH_ALL, const LaunchProbe* probe = nullptr) {
    const int levels = a.frame.max_bounce + 1;
    const dim3 block(64);
    // persistent grids (64-frame chunks are strided over them); an empty launch costs ~1 us per 4096
    // workgroups, so the deeper, usually sparse levels get smaller grids
    // (measured with 16 frames in flight: 32768 instead of 8192 workgroups for the one-lane-per-ray stage 2 and 4096
    // instead of 2048 for k_consume balance the chunks better, -8 %; a single frame is unchanged)
    // (round 2, counters on lines of their own: 32768 workgroups for k_trace(L0) 135 -> 125 us, 16384 for k_consume(L0) 134 -> 109 us;
    // 16384 or 65536 instead of 32768 for the one-lane-per-ray stage 2: +3 % / +1 %)
    const dim3 gridT(32768), gridN(32768), gridS(8192), gridF0(16384), gridF(4096), gridC(1024), gridCoop(512);
    // (KernelArgs::list_n) the stage-2 kernel of a phase that is not expected to find work gets a grid that costs less to launch
    // and still gets through the list should it be the one after all: cooperative up to the threshold (narrow_geom)
    const uint32_t thr = (uint32_t)(a.frame.coop_threshold > 0 ? a.frame.coop_threshold : 70000);
    auto grid_lane = [&](int ph, dim3 full) {  // one lane per ray
        const uint32_t n1 = ph < 8 ? a.list_n[ph] : 0u;
        if (n1 == 0u || n1 - 1u > thr) return full;  // unknown, or this kernel's list
        uint32_t g = ((n1 - 1u + 63u) / 64u) * 2u;
        g = g < 2048u ? 2048u : g;  // (two wavefronts per SIMD: a list ten times the last one is still walked at two thirds of the full rate)
        return g < full.x ? dim3(g) : full;
    };
    auto grid_coop = [&](int ph) {
        const uint32_t n1 = ph < 8 ? a.list_n[ph] : 0u;
        return (n1 != 0u && n1 - 1u > 2u * (uint64_t)thr) ? dim3(16) : gridCoop;
    };
    auto grid_coop2 = [&](int ph) {  // (k_trace2c: workgroups of half the size, twice as many)
        const uint32_t n1 = ph < 8 ? a.list_n[ph] : 0u;
        return (n1 != 0u && n1 - 1u > 2u * (uint64_t)thr) ? dim3(32) : dim3(2u * gridCoop.x);
    };
    // The cooperative kernel of a phase whose list was beyond the threshold last time is not launched at all, and the one-lane-per-ray
    // kernel takes the list whatever its length turns out to be (`alone`; any choice renders the same image). Idle, that kernel costs
    // nothing in an EMPTY machine (launches overlap in the command processor: measured) — but its 1024-thread workgroups want a CU's
    // whole LDS and sixteen wavefront slots, and beside another stream's kernels (side mode's stage 2 of the primary phase, another
    // context's sequence) no CU is ever empty: kernel traces show the idle launch waiting 144 - 1500 us for one, the launch sequence
    // stalled behind it. (rtu_debug_flags 16384: launch both outside side mode, as before.)
    auto no_coop = [&](int ph) {
        const uint32_t n1 = ph < 8 ? a.list_n[ph] : 0u;
        return (a.side || !(a.dbg & 16384u)) && n1 != 0u && n1 - 1u > (uint64_t)thr;
    };
    // ... and the other way round: a phase whose list was below seven eighths of the threshold last time launches the cooperative kernel ONLY,
    // which then takes the list whatever its length (an idle launch of the one-lane-per-ray kernel is 1.3 us of kernel and a microsecond of
    // gap: four of them are 3 % of a single frame).
    auto no_lane = [&](int ph) {
        const uint32_t n1 = ph < 8 ? a.list_n[ph] : 0u;
        return !(a.dbg & 16384u) && n1 != 0u && (uint64_t)(n1 - 1u) <= (uint64_t)thr - thr / 8u;
    };
    if (n_tiles == 0) return (int)hipSuccess;
    // k_primary: one tile per wavefront for the counting variant; the fast variant strides its tiles over at most 32768 workgroups
    const uint32_t blocksP = (n_tiles + 3) / 4;
    // (the fast variant's grid: a wavefront renders several tiles. 32768 workgroups where most tiles have work — fine grain balances them —,
    // 4096 where most are background: a wavefront's prologue and its scalar set-up are then a tenth of what it does. KernelArgs::pgrid,
    // from the occupied tiles the last launch of this shape counted: 363 -> 315 us per 20 frames of the headline, measured)
    const uint32_t pcap = a.pgrid ? a.pgrid : 32768u;
    const dim3 gridP(blocksP), gridPF(blocksP < pcap ? blocksP : pcap);
    if (CNTD) stats = false;  // the touched-bytes instantiations are the fast variant's (the reference-counting kernels are not built for them)
    if (mode == RTU_LAUNCH_SHADE) {
        RTU_LAUNCH(RTU_TL_GI_ROOTS, (k_gi_roots<TEX>), gridN, block, a, (stats || (a.dbg & 2048u)) ? 0 : 1);  // (rtu_debug_flags 2048: no inline shading — results must not change)
    } else if (stats) {
        if constexpr (!CNTD) hipLaunchKernelGGL((k_primary_counting<STACK, TEX>), gridP, dim3(256), 0, stream, a, n_tiles);
    } else {
        if (!SMPD && a.node_rects) {
            const uint32_t entries = BATD ? a.batch : 1u;
            const uint32_t rows = (a.cover ? entries * (a.scene.n_cover + a.scene.n_pcover) : 0u) + 1u;
            uint32_t gx = 1u;
            if (a.cover && a.scene.n_cover) gx = (a.cover_faces + 255u) / 256u > gx ? (a.cover_faces + 255u) / 256u : gx;
            if (a.cover && a.scene.n_pcover) gx = (a.cover_words * 32u + 255u) / 256u > gx ? (a.cover_words * 32u + 255u) / 256u : gx;
            hipLaunchKernelGGL(k_prelude, dim3(gx, rows), dim3(256), a.cover ? a.cover_words * sizeof(uint32_t) : 0, stream, a, entries);
        }
        if (!SMPD && !GID && a.occ) hipLaunchKernelGGL(k_tile_occ, dim3(a.occ_words / 2u, (BATD ? a.batch : 1u)), dim3(64), 0, stream, a, (BATD ? a.batch : 1u));
        if constexpr (SMPD || GID) RTU_LAUNCH(RTU_TL_PRIMARY, (k_primary_sampled<STACK, TEX>), gridPF, dim3(256), a, n_tiles);
        else RTU_LAUNCH(RTU_TL_PRIMARY, (k_primary<STACK, false, TEX>), gridPF, dim3(256), a, n_tiles);
        if (a.n_meshes && a.side && mode == RTU_LAUNCH_ALL) {
            // SIDE MODE (KernelArgs::fcnt0): stage 2 of the primary phase on the helper stream, beside the recursion levels. Its kernels
            // get the side set of level arrays and counters; the few frames they make are evaluated by one k_tail launch behind them.
            KernelArgs a2 = a;
            for (int L = 0; L < RTU_MAX_LEVELS; L++) a2.lv[L] = a.lv_side[L];
            a2.fcnt = a.fcnt0;
            const hipStream_t aux = (hipStream_t)a.aux_stream;
            (void)hipEventRecord((hipEvent_t)a.aux_ev0, stream);      // k_primary is queued: its defer list and counters are what stage 2 reads
            (void)hipStreamWaitEvent(aux, (hipEvent_t)a.aux_ev0, 0);
            {
                const KernelArgs& a = a2;  // (the launch macro takes its arguments from `a`)
                // (side mode is only chosen when the last launch's list was the one-lane-per-ray kernel's: no cooperative launch)
                RTU_LAUNCH_ON(aux, RTU_TL_PRIMARY2, (k_primary2<STACK, TEX>), grid_lane(0, gridN), block, a, 1);
                RTU_LAUNCH_ON(aux, RTU_TL_SIDE_TAIL, (k_tail<TEX>), dim3(1024), block, a, 0, (int)RTU_TL_SIDE_TAIL);
            }
            (void)hipEventRecord((hipEvent_t)a.aux_ev1, aux);
        } else if (a.n_meshes) {  // without meshes nothing is ever deferred
            const bool nc = no_coop(0), nl = no_lane(0);
            if (!nc) RTU_LAUNCH(RTU_TL_PRIMARY2C, (k_primary2c<STACK, TEX>), grid_coop(0), dim3(RTU_COOP_THREADS), a, nl ? 1 : 0);
            if (!nl) RTU_LAUNCH(RTU_TL_PRIMARY2, (k_primary2<STACK, TEX>), grid_lane(0, gridN), block, a, nc ? 1 : 0);
        }
    }
    if (mode == RTU_LAUNCH_CHAIN) return (int)hipGetLastError();
    // levels >= tail_from are evaluated by k_tail (fast variant only; the host passes tail_from >= 1, or 6 for none)
    const int regular = (!stats && a.tail_from >= 1 && a.tail_from < levels) ? a.tail_from : levels;
    for (int L = 0; L < regular; L++) {
        const int ph = 1 + L;  // defer list of this level's tracing phase
        if (stats) {
            if constexpr (!CNTD) {
                hipLaunchKernelGGL((k_trace_counting<STACK, TEX>), gridT, block, 0, stream, a, L, (int)(SEL_SHADOW | SEL_MAIN | SEL_C), ph);
                if (L + 1 < levels) hipLaunchKernelGGL((k_trace_counting<STACK, TEX>), gridT, block, 0, stream, a, L, (int)(SEL_A | SEL_A_NEEDS_B), ph);
                hipLaunchKernelGGL((k_consume<true, TEX>), gridF, block, 0, stream, a, L);
            }
        } else {
            const int sel = (int)(SEL_SHADOW | SEL_MAIN | SEL_A | SEL_C);
            RTU_LAUNCH(RTU_TL_LEVEL0 + 4 * L, (k_trace<STACK, false, TEX>), L == 0 ? gridT : gridS, block, a, L, sel, ph);
            if (a.n_meshes) {
                const bool nc = no_coop(ph), nl = no_lane(ph);
                if (!nc) RTU_LAUNCH(RTU_TL_LEVEL0 + 4 * L + 1, (k_trace2c<STACK, TEX>), grid_coop2(ph), dim3(RTU_COOP2_THREADS), a, L, sel, ph, nl ? 1 : 0);
                if (!nl) RTU_LAUNCH(RTU_TL_LEVEL0 + 4 * L + 2, (k_trace2<STACK, TEX>), grid_lane(ph, L == 0 ? gridN : gridS), block, a, L, sel, ph, nc ? 1 : 0);
            }
            RTU_LAUNCH(RTU_TL_LEVEL0 + 4 * L + 3, (k_consume<false, TEX>), L == 0 ? gridF0 : gridF, block, a, L);
        }
    }
    if (regular < levels) RTU_LAUNCH(RTU_TL_LEVEL0 + 4 * regular, (k_tail<TEX>), dim3(8192), block, a, regular, RTU_TL_LEVEL0 + 4 * regular);
    for (int L = regular - 2 + (regular < levels ? 1 : 0); L >= 0; L--) RTU_LAUNCH(RTU_TL_COMBINE0 + L, (k_combine<TEX>), gridC, block, a, L);
    if (!stats && a.n_meshes && a.side && mode == RTU_LAUNCH_ALL) (void)hipStreamWaitEvent(stream, (hipEvent_t)a.aux_ev1, 0);  // the frame is complete when stage 2's pixels are
    return (int)hipGetLastError();
}

}  // namespace

#endif
