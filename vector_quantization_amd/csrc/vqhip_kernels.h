// Device kernels of libvqhip (gfx950 / CDNA4 only — wave64, MFMA, LDS-DMA).
//
// Arithmetic contract: the result of every index-producing entry point is the argmin of the
// fp32 definition evaluated with k-ordered fma chains — exactly what v_mfma_f32_32x32x2_f32 and the
// scalar fmaf loops below compute, and what oracle/vq_oracle.c restates on the CPU.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "vqhip_layout.h"

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned long long u64;

// internal metric word: low bits = VQHIP_METRIC_L2 / _COS / VQ_METRIC_DOT (1 - x.e on operands used as given: the
// row/column-swapped NearestAnchor pass), bit 8 = the L2 finishing adds the CODE norm first: (c + |code|^2) + |row|^2
#define VQ_METRIC_DOT 2
#define VQ_METRIC_SWAP 0x100
#define VQ_METRIC_BF16 0x4          // with COS / DOT: bf16-autocast semantics (VQHIP_METRIC_COS_BF16 = COS | BF16)
#define VQ_IS_BF16(m) (((m) & VQ_METRIC_BF16) != 0)
#define VQ_IS_L2(m) (((m) & 3) == VQHIP_METRIC_L2)
#define VQ_IS_COS(m) (((m) & 3) == VQHIP_METRIC_COS)
#define VQ_SWAPPED(m) (((m) & VQ_METRIC_SWAP) != 0)

#define VQ_F16_MIN_NORMAL 6.103515625e-05f
#define VQ_U 5.9604644775390625e-08f /* 2^-24 */

// ------------------------------------------------------------------------------------------------
// small helpers
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float bf16_to_f32(uint16_t b) { return __uint_as_float(((uint32_t)b) << 16); }

template <int DT>
__device__ __forceinline__ float load_elem(const void *p, int64_t i) {
    if (DT == 0) return ((const float *)p)[i];
    return bf16_to_f32(((const uint16_t *)p)[i]);
}

// 8 consecutive elements starting at element offset i (i % 8 == 0, rows 16/32-byte aligned)
template <int DT>
__device__ __forceinline__ void load8(const void *p, int64_t i, float (&v)[8]) {
    if (DT == 0) {
        const float4 *q = (const float4 *)((const float *)p + i);
        float4 a = q[0], b = q[1];
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    } else {
        uint4 a = *(const uint4 *)((const uint16_t *)p + i);
        v[0] = __uint_as_float(a.x << 16); v[1] = __uint_as_float(a.x & 0xFFFF0000u);
        v[2] = __uint_as_float(a.y << 16); v[3] = __uint_as_float(a.y & 0xFFFF0000u);
        v[4] = __uint_as_float(a.z << 16); v[5] = __uint_as_float(a.z & 0xFFFF0000u);
        v[6] = __uint_as_float(a.w << 16); v[7] = __uint_as_float(a.w & 0xFFFF0000u);
    }
}


// raw 8-element vector loads for the proposal-pass prologue (kept as integers until all are in flight)
template <int DT> struct RawVec;
template <> struct RawVec<0> {
    struct type { float4 a, b; };
    static __device__ __forceinline__ type load(const void *p, int64_t i) {
        const float4 *q = (const float4 *)((const float *)p + i);
        type t; t.a = q[0]; t.b = q[1]; return t;
    }
    static __device__ __forceinline__ void unpack(const type &t, float (&v)[8]) {
        v[0] = t.a.x; v[1] = t.a.y; v[2] = t.a.z; v[3] = t.a.w; v[4] = t.b.x; v[5] = t.b.y; v[6] = t.b.z; v[7] = t.b.w;
    }
};
template <> struct RawVec<1> {
    typedef uint4 type;
    static __device__ __forceinline__ type load(const void *p, int64_t i) { return *(const uint4 *)((const uint16_t *)p + i); }
    static __device__ __forceinline__ void unpack(const type &a, float (&v)[8]) {
        v[0] = __uint_as_float(a.x << 16); v[1] = __uint_as_float(a.x & 0xFFFF0000u);
        v[2] = __uint_as_float(a.y << 16); v[3] = __uint_as_float(a.y & 0xFFFF0000u);
        v[4] = __uint_as_float(a.z << 16); v[5] = __uint_as_float(a.z & 0xFFFF0000u);
        v[6] = __uint_as_float(a.w << 16); v[7] = __uint_as_float(a.w & 0xFFFF0000u);
    }
};

// fp32 -> nearest bf16 (ties to even), returned as fp32: torch's rounding (NaN / Inf pass through)
__device__ __forceinline__ float bf16_rne(float v) {
    uint32_t b = __float_as_uint(v);
    if ((b & 0x7F800000u) == 0x7F800000u) return v;
    b += 0x7FFFu + ((b >> 16) & 1u);
    return __uint_as_float(b & 0xFFFF0000u);
}
// 1 - similarity; bf16-autocast semantics round the similarity and the difference to bf16 (include/vqhip.h)
__device__ __forceinline__ float cos_distance(float c, int metric) {
    if (((metric) & 0x4) != 0) return bf16_rne(1.0f - bf16_rne(c));
    return 1.0f - c;
}

// The barrier that hands an LDS-DMA tile to the other waves.  __syncthreads() is a workgroup-scope release fence + s_barrier, and a
// release fence owes nothing to outstanding LOADS — which is what global_load_lds is to the compiler: hipcc emits s_waitcnt vmcnt(0)
// in front of such a barrier only where something else on the path needs it (exact_stream_kernel had it at one of its two loop
// barriers, not at the other: right alone on the GPU, where every piece lands long before the barrier, wrong on 1-2 % of the rows
// with eight processes sharing it — round 6, tools/isa_dma_barriers.py lists the barriers of every kernel that fills LDS this way).
// The drain is therefore written out wherever a barrier publishes DMA'd data.
__device__ __forceinline__ void vq_dma_barrier() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
}

// single-instruction max (hipcc otherwise wraps fmaxf on MFMA results in canonicalising v_max pairs)
__device__ __forceinline__ float vmax(float a, float b) {
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// v_max3_f32 / v_max_f32 on values an MFMA produced, issued between MFMAs.  hipcc pads no hazard for an instruction
// INSIDE an asm statement (MI355X guide §5.7 item 2), so the statements below are made hazard-free by construction:
//  * inputs (an MFMA's D needs 12 wait states before a non-MFMA reader): `after` is a fake input — named, never read by
//    the instruction; pass the result of an MFMA issued at least two MFMAs later than the producers: the dependence the
//    compiler sees keeps the statement behind that MFMA, and two back-to-back 16-cycle MFMAs cover the wait states;
//  * output (a VALU write to a register an in-flight MFMA still reads as A/B/C operand corrupts that MFMA — found by
//    the randomised campaign: the allocator had given a temporary the register of the A fragment just issued): the
//    destination is a "+v" variable the caller keeps alive across the whole MFMA loop, so its register is never shared
//    with an MFMA operand (that turned out NOT to be the cause of the mismatches; kept because it costs nothing).
// What the campaign did pin down (tools/fuzz_vs_exact.py, D = 64/128: 1.3 % of the trials wrong): one fake input is not
// enough — hipcc reorders the MFMAs of a chunk, so the single MFMA the statement was tied to could be the one issued
// right behind the producers.  The statement is therefore tied to ALL TT MFMAs of the chunk (tile_max8 below).
// Group records (coarse_kernel, GROUPS): the wave-uniform skip test stays in front of the 4-instruction group update from
// this many token tiles per wave on (measured: with 2 tiles the straight-line form is faster, with 4 the test pays)
#ifndef VQ_REPLAY_BATCH
#define VQ_REPLAY_BATCH 4      // (8 without aux reads: spills inside the replay loop, 123 instead of 92 us at configs[2])
#endif
#ifndef VQ_PIPEH_PF
#define VQ_PIPEH_PF 3
#endif
#ifndef VQ_GROUP_BRANCH_MIN_TT
#define VQ_GROUP_BRANCH_MIN_TT 4
#endif
// Maximum of the 8 accumulator elements p[0..7] of one (token tile, code tile) into `dst` (sc1/sc2: scratch): four asm
// instructions ordered behind ALL the fake inputs after[] — the results of the TT MFMAs of the CURRENT tile, i.e. at least two
// MFMAs after the producers (see above).  (Measured alternatives, profiles/r02_tilemax_variants_c3.txt: compiler-visible
// v_med3 maxima, 7 instructions, hazards padded by hipcc; the same statement behind s_nop 11 / s_nop 3: 1-6 % slower.)
template <int TT>
__device__ __forceinline__ void tile_max8(float &dst, float &sc1, float &sc2, const f32x4 &lo, const f32x4 &hi,
                                          const float (&after)[TT]) {
    const float a0 = after[0], a1 = after[TT > 1 ? 1 : 0], a2 = after[TT > 2 ? 2 : 0], a3 = after[TT > 3 ? 3 : 0];
    asm("v_max3_f32 %0, %3, %4, %5\n\t"
        "v_max3_f32 %1, %6, %7, %8\n\t"
        "v_max_f32 %2, %9, %10\n\t"
        "v_max3_f32 %0, %0, %1, %2"
        : "+v"(dst), "+v"(sc1), "+v"(sc2)
        : "v"(lo[0]), "v"(lo[1]), "v"(lo[2]), "v"(lo[3]), "v"(hi[0]), "v"(hi[1]), "v"(hi[2]), "v"(hi[3]),
          "v"(a0), "v"(a1), "v"(a2), "v"(a3));
}
__device__ __forceinline__ float vmax3(float a, float b, float c) {
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
// max over the four lanes l, l^16, l^32, l^48 (the lanes that share a token in the 16x16 MFMA output):
// v_permlane16_swap / v_permlane32_swap exchange 16-lane rows / 32-lane halves of two registers in place; fed the same
// value twice, {result 0, result 1} = {own, partner} in some order on every lane
__device__ __forceinline__ float quad_rows_max(float v) {
    auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = vmax(__uint_as_float(a[0]), __uint_as_float(a[1]));
    auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return vmax(__uint_as_float(b[0]), __uint_as_float(b[1]));
}

// fp32 -> fp16 (RNE) with subnormal results flushed to zero, so the MFMA never sees an fp16 subnormal
__device__ __forceinline__ _Float16 to_f16_ftz(float v) {
    _Float16 q = (_Float16)v;
    float b = (float)q;
    if (fabsf(b) < VQ_F16_MIN_NORMAL) q = (_Float16)0.0f;   // NaN compares false and stays NaN
    return q;
}

__device__ __forceinline__ float wave_sum_tree(float p) {   // halving tree 32,16,...,1 (oracle order)
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) p = p + __shfl_xor(p, off, 64);
    return p;
}
__device__ __forceinline__ float wave_max(float p) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) p = fmaxf(p, __shfl_xor(p, off, 64));
    return p;
}

// power-of-two scale that maps max|e| into [2^13, 2^14)
__device__ __forceinline__ float cb_scale(const VqCbStats *st) {
    float m = __uint_as_float(st->maxabs_bits);
    if (!(m > 0.0f) || st->nonfinite) return 1.0f;
    int ex;
    (void)frexpf(m, &ex);            // m = f * 2^ex, f in [0.5,1)
    int sh = 14 - ex;
    sh = sh > 100 ? 100 : (sh < -100 ? -100 : sh);
    return ldexpf(1.0f, sh);
}


// atomicMax on a hot word: read first (L2 hit), issue the atomic only when it would raise the value —
// same-address atomics serialise at ~11 ns each, and after the first few waves the filter drops them all
__device__ __forceinline__ void atomic_max_filtered(uint32_t *p, uint32_t v) {
    if (v > __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(p, v);
}

// C/D register -> row of the 32x32 MFMA tile (v_mfma_f32_32x32x2_f32, MI355X guide §3): row = (r&3) + 8*(r>>2) + 4*(lane>>5)
__device__ __forceinline__ int mfma_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// The statistics header as the margin consumes it: header words + the maxima of the VQ_CB_SLOTS slots cb_image_kernel raised
// (lanes 0..15 load one slot each, a 16-lane shuffle tree folds them).  Wave-level: every lane of the wave must call it.
// `folded` of a published header: a checksum of the four words it vouches for (never 0).  A reader that caught the header in the
// middle of a publication by ANOTHER wave — of the same launch, or of a launch on another stream that shares a frozen codebook
// image — sees a flag that does not match the words it loaded, and folds the partials itself as if nothing had been published:
// no acquire fence (an L1 / K-cache invalidate in front of every consumer wave's first instruction) is needed for that.
__device__ __forceinline__ uint32_t cb_folded_mark(uint32_t r2, uint32_t eh2, uint32_t e2, uint32_t nonfinite) {
    uint32_t h = r2 * 0x9E3779B1u;
    h = (h ^ (h >> 15)) + eh2 * 0x85EBCA77u;
    h = (h ^ (h >> 13)) + e2 * 0xC2B2AE3Du;
    h = (h ^ (h >> 16)) + nonfinite * 0x27D4EB2Fu;
    return h | 0x80000000u;
}
__device__ __forceinline__ VqCbStats cb_stats_view(const VqCbStats *st) {
    VqCbStats v = *st;
    const int lane = threadIdx.x & 63;
    const uint32_t *slot = (const uint32_t *)((const char *)st + 256 + (lane & (VQ_CB_SLOTS - 1)) * 128);
    uint32_t r = slot[0], h = slot[1], bad = slot[2];
#pragma unroll
    for (int off = VQ_CB_SLOTS / 2; off >= 1; off >>= 1) {
        const uint32_t r2 = __shfl_xor(r, off, 64), h2 = __shfl_xor(h, off, 64), b2 = __shfl_xor(bad, off, 64);
        r = r > r2 ? r : r2; h = h > h2 ? h : h2; bad |= b2;
    }
    // cosine, already folded into the header by an earlier launch (cb_stats_publish) — and seen whole (see cb_folded_mark)
    if (v.folded != 0u && v.folded == cb_folded_mark(v.r2max_bits, v.eh2max_bits, v.e2max_bits, v.nonfinite)) return v;
    if (v.part2_n != 0u) {      // cosine image made in one launch: per-tile partials instead of the slots (wave-uniform)
        const f32x4 *part = (const f32x4 *)((const char *)st + v.part2_off);
        float pr = 0.0f, ph = 0.0f, pb = 0.0f, pe = 0.0f;
        // 512 partials (K = 16 384) per pass: the lane's eight loads are all in flight before the first is used (a plain
        // loop waited for each in turn: eight serialized L2 round trips in front of every consumer wave's first instruction)
        for (uint32_t base = 0; base < v.part2_n; base += 512u) {
            f32x4 q[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const uint32_t i = base + (uint32_t)lane + 64u * (uint32_t)u;
                q[u] = part[i < v.part2_n ? i : 0u];            // (a repeated entry changes no maximum)
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) { pr = fmaxf(pr, q[u][0]); ph = fmaxf(ph, q[u][1]); pb = fmaxf(pb, q[u][2]); pe = fmaxf(pe, q[u][3]); }
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            pr = fmaxf(pr, __shfl_xor(pr, off, 64)); ph = fmaxf(ph, __shfl_xor(ph, off, 64));
            pb = fmaxf(pb, __shfl_xor(pb, off, 64)); pe = fmaxf(pe, __shfl_xor(pe, off, 64));
        }
        v.r2max_bits = __float_as_uint(pr); v.eh2max_bits = __float_as_uint(ph); v.e2max_bits = __float_as_uint(pe);
        v.nonfinite |= (pb > 0.0f || !(pr == pr) || !(ph == ph) || !(pe == pe)) ? 1u : 0u;
        return v;
    }
    v.r2max_bits = r; v.eh2max_bits = h; v.nonfinite |= bad;
    return v;
}

// Cosine images (cb_cos_body) keep their maxima as per-tile partials; folding them costs every consumer WAVE 4-8 loads and a
// 64-lane reduction (2-4 us in front of the latency-bound decision and re-rank kernels).  The proposal kernel of an argmin —
// the first launch behind the image — has ONE wave do it for everybody that comes later: fold, write the header, set `folded`
// (plain stores: visible to the launches that follow; waves of the same launch still fold for themselves).  Idempotent: a
// second argmin on the same image writes the same values.
__device__ __forceinline__ void cb_stats_publish(const VqCbStats *st) {
    if (blockIdx.x != 0 || (threadIdx.x >> 6) != 0) return;
    if (st->part2_n == 0u || st->folded != 0u) return;
    const VqCbStats v = cb_stats_view(st);
    if ((threadIdx.x & 63) == 0) {
        VqCbStats *w = const_cast<VqCbStats *>(st);
        w->r2max_bits = v.r2max_bits; w->eh2max_bits = v.eh2max_bits; w->e2max_bits = v.e2max_bits; w->nonfinite = v.nonfinite;
        // the four words have reached L2 before the flag is stored (no agent-scope release: its L2 write-back put 3-4 us on this
        // workgroup, i.e. on the kernel).  Launches that follow on the stream see everything; a reader that overlaps this
        // publication validates the flag against the words it loaded (cb_folded_mark) and otherwise folds for itself
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        w->folded = cb_folded_mark(v.r2max_bits, v.eh2max_bits, v.e2max_bits, v.nonfinite);
    }
}

// ---- the units (one translation unit: vqhip.hip includes this header) ----
#include "vqhip_prepare_kernels.h"
#include "vqhip_proposal_kernels.h"
#include "vqhip_refine_kernels.h"
#include "vqhip_proposal32_kernels.h"
#include "vqhip_exact_kernels.h"
#include "vqhip_stream_kernels.h"
#include "vqhip_update_kernels.h"
#include "vqhip_sort_kernels.h"
#include "vqhip_aux_kernels.h"
#include "vqhip_exchange_kernels.h"
#include "vqhip_step_kernels.h"
