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
#ifndef VQ_FILTER_MAX_IMPL
#define VQ_FILTER_MAX_IMPL 3
#endif
// Group records (coarse_kernel, GROUPS): the wave-uniform skip test stays in front of the 4-instruction group update from
// this many token tiles per wave on (measured: with 2 tiles the straight-line form is faster, with 4 the test pays)
#ifndef VQ_REPLAY_BATCH
#define VQ_REPLAY_BATCH 4      // (8 without aux reads: spills inside the replay loop, 123 instead of 92 us at configs[2])
#endif
#ifndef VQ_GROUP_BRANCH_MIN_TT
#define VQ_GROUP_BRANCH_MIN_TT 4
#endif
// Maximum of the 8 accumulator elements p[0..7] of one (token tile, code tile) into `dst` (sc1/sc2: scratch).
// after[]: results of TT MFMAs of the CURRENT tile (fake inputs, see above).  Variants kept for the A/B record:
//   2: compiler-visible v_med3 (max(a,b) = med3(a,b,+inf)), 7 instructions, hazards handled by hipcc
//   3: 4 asm instructions ordered behind ALL the fake inputs (>= 2 MFMAs after the producers)
//   4: as 3 in one statement that opens with s_nop 11 (the full 12 wait states, wherever it is placed)
//   5: as 3 with s_nop 3 in front
template <int TT>
__device__ __forceinline__ void tile_max8(float &dst, float &sc1, float &sc2, const f32x4 &lo, const f32x4 &hi,
                                          const float (&after)[TT]) {
#if VQ_FILTER_MAX_IMPL == 2
    float m = __builtin_amdgcn_fmed3f(lo[0], lo[1], INFINITY);
    m = __builtin_amdgcn_fmed3f(m, lo[2], INFINITY); m = __builtin_amdgcn_fmed3f(m, lo[3], INFINITY);
    m = __builtin_amdgcn_fmed3f(m, hi[0], INFINITY); m = __builtin_amdgcn_fmed3f(m, hi[1], INFINITY);
    m = __builtin_amdgcn_fmed3f(m, hi[2], INFINITY); dst = __builtin_amdgcn_fmed3f(m, hi[3], INFINITY);
    (void)sc1; (void)sc2; (void)after;
#else
    const float a0 = after[0], a1 = after[TT > 1 ? 1 : 0], a2 = after[TT > 2 ? 2 : 0], a3 = after[TT > 3 ? 3 : 0];
#if VQ_FILTER_MAX_IMPL == 4
#define VQ_TM_HEAD "s_nop 11\n\t"
#elif VQ_FILTER_MAX_IMPL == 5
#define VQ_TM_HEAD "s_nop 3\n\t"
#else
#define VQ_TM_HEAD ""
#endif
    asm(VQ_TM_HEAD
        "v_max3_f32 %0, %3, %4, %5\n\t"
        "v_max3_f32 %1, %6, %7, %8\n\t"
        "v_max_f32 %2, %9, %10\n\t"
        "v_max3_f32 %0, %0, %1, %2"
        : "+v"(dst), "+v"(sc1), "+v"(sc2)
        : "v"(lo[0]), "v"(lo[1]), "v"(lo[2]), "v"(lo[3]), "v"(hi[0]), "v"(hi[1]), "v"(hi[2]), "v"(hi[3]),
          "v"(a0), "v"(a1), "v"(a2), "v"(a3));
#undef VQ_TM_HEAD
#endif
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
    v.r2max_bits = r; v.eh2max_bits = h; v.nonfinite |= bad;
    return v;
}

// ------------------------------------------------------------------------------------------------
// row kernels: oracle-order |v|^2 and F.normalize
// ------------------------------------------------------------------------------------------------
template <int DT>
__global__ void row_sqnorm_kernel(const void *v, int64_t R, int D, float *out) {
    int64_t r = (int64_t)blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6);
    int lane = threadIdx.x & 63;
    if (r >= R) return;
    float p = 0.0f;
    for (int d = lane; d < D; d += 64) { float a = load_elem<DT>(v, r * D + d); p = fmaf(a, a, p); }
    p = wave_sum_tree(p);
    if (lane == 0) out[r] = p;
}

template <int DT>
__global__ void normalize_rows_kernel(const void *v, int64_t R, int D, float eps, float *out) {
    int64_t r = (int64_t)blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6);
    int lane = threadIdx.x & 63;
    if (r >= R) return;
    float p = 0.0f;
    for (int d = lane; d < D; d += 64) { float a = load_elem<DT>(v, r * D + d); p = fmaf(a, a, p); }
    p = wave_sum_tree(p);
    float nrm = sqrtf(p);
    float den = (nrm < eps) ? eps : nrm;
    for (int d = lane; d < D; d += 64) out[r * D + d] = load_elem<DT>(v, r * D + d) / den;
}

// D <= 32 (the LlamaGen tokenizer normalises 8-dim latents, VQ-KD 32-dim ones): a whole wave per row leaves 7/8 of the lanes
// idle and launches one wave per token (81 us for 524 288 x 8 where the data is 25 MB).  L lanes per row, 64 / L rows per
// wave; the halving tree runs inside the L-lane group — the very additions of the full-wave tree, whose upper levels only add
// the zeros of the idle lanes — so the results are bit-identical to the kernels above.
template <int DT, int L, bool NORMALIZE>
__global__ void row_small_kernel(const void *v, int64_t R, int D, float eps, float *out) {
    const int lane = threadIdx.x & 63;
    const int64_t r = ((int64_t)blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6)) * (64 / L) + lane / L;
    const int d = lane % L;
    const bool live = r < R && d < D;
    const float a = live ? load_elem<DT>(v, r * D + d) : 0.0f;
    float p = fmaf(a, a, 0.0f);
#pragma unroll
    for (int off = L / 2; off >= 1; off >>= 1) p = p + __shfl_xor(p, off, 64);
    if constexpr (NORMALIZE) {
        const float nrm = sqrtf(p);
        const float den = (nrm < eps) ? eps : nrm;
        if (live) out[r * D + d] = a / den;
    } else {
        if (r < R && d == 0) out[r] = p;
    }
}

// ------------------------------------------------------------------------------------------------
// codebook preparation
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float block_max4(float v, float *red) {    // max over the 4 waves of a 256-thread block
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

// pass 1 (one wave per 4 codes): |e_k|^2 in oracle order, optional normalisation into e_exact, max|e|, flags.
// The four rows of a wave are loaded together and reduced with interleaved shuffle trees; maxima are reduced per
// block and written as one partial per block (same-line atomics from ~1000 concurrent blocks cost ~25 us).
__device__ __forceinline__ void cb_stats_body(int64_t blk, const float *e, int64_t K, int D, int metric, char *cb, const VqCbLayout &L) {
    __shared__ float red[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    VqCbStats *st = (VqCbStats *)(cb + L.off_stats);
    float *en = (float *)(cb + L.off_en);
    float *ex = (float *)(cb + L.off_eexact);
    // the maxima slots cb_image_kernel (the next launch) raises start from zero
    if (blk == 0 && threadIdx.x < VQ_CB_SLOTS) {
        uint32_t *slot = (uint32_t *)(cb + L.off_stats + 256 + threadIdx.x * 128);
        slot[0] = 0u; slot[1] = 0u; slot[2] = 0u;
    }
    const int64_t k0 = (blk * 4 + wave) * 4;
    float p[4] = {0, 0, 0, 0}, amax = 0.0f;
    bool bad = false;
    for (int d = lane; d < D; d += 64) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float a = (k0 + c < K) ? e[(k0 + c) * D + d] : 0.0f;
            p[c] = fmaf(a, a, p[c]); amax = fmaxf(amax, fabsf(a)); bad |= !isfinite(a);
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1)
#pragma unroll
        for (int c = 0; c < 4; ++c) p[c] = p[c] + __shfl_xor(p[c], off, 64);
    float m_e2 = 0.0f, m_en = 0.0f;
    if (VQ_IS_COS(metric)) {
        amax = 0.0f;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            if (k0 + c >= K) continue;
            float nrm = sqrtf(p[c]);
            float den = (nrm < 1e-12f) ? 1e-12f : nrm;
            float q2 = 0.0f;
            for (int d = lane; d < D; d += 64) {
                float a = e[(k0 + c) * D + d] / den;
                if (VQ_IS_BF16(metric)) a = bf16_rne(a);            // bf16-autocast: the einsum sees bf16(normalize(e))
                ex[(k0 + c) * D + d] = a;
                amax = fmaxf(amax, fabsf(a)); bad |= !isfinite(a); q2 = fmaf(a, a, q2);
            }
            q2 = wave_sum_tree(q2);
            bad |= !isfinite(q2);
            m_e2 = fmaxf(m_e2, q2);
            if (lane == 0) en[k0 + c] = 0.0f;
        }
    } else {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            if (k0 + c >= K) continue;
            if (lane == 0) en[k0 + c] = VQ_IS_L2(metric) ? p[c] : 0.0f;     // DOT: operands are used as given, no bias
            bad |= !isfinite(p[c]);
            m_e2 = fmaxf(m_e2, p[c]);
            if (VQ_IS_L2(metric)) m_en = fmaxf(m_en, p[c]);
        }
    }
    amax = wave_max(amax);
    bad = __any(bad);
    amax = block_max4(amax, red); m_e2 = block_max4(m_e2, red); m_en = block_max4(m_en, red);
    float badf = block_max4(bad ? 1.0f : 0.0f, red);
    // per-block partial result; reduced by every block of cb_image_kernel (no hot-word atomics, no memset)
    if (threadIdx.x == 0) ((f32x4 *)(cb + L.off_part1))[blk] = f32x4{amax, m_e2, m_en, badf};
    (void)st;
}
__global__ __launch_bounds__(256) void cb_stats_kernel(const float *e, int64_t K, int D, int metric, char *cb, VqCbLayout L) {
    cb_stats_body(blockIdx.x, e, K, D, metric, cb, L);
}

// pass 2 (one 256-thread block per tile of 32 codes): the MFMA-fragment-major fp16 image, the aux chunk, and the fp16
// residual / image norms with the final scale.
// chunk (tile T, k-step s of 32 dims, half c) holds, for lane l, code T*32 + 16c + (l&15), dims 32s + 8(l>>4) .. +8 —
// exactly the A operand of v_mfma_f32_16x16x32_f16 — so a linear global_load_lds copy gives a conflict-free LDS image.
__global__ __launch_bounds__(256) void cb_image_kernel(const float *e, int64_t K, int D, int metric, char *cb, VqCbLayout L) {
    __shared__ float red[2][8][32];
    __shared__ float red4[4];
    const int64_t tile = blockIdx.x;
    const int64_t stage = tile / L.tps;
    const int ti = (int)(tile % L.tps);
    const int r = threadIdx.x & 31, g = threadIdx.x >> 5;
    VqCbStats *st = (VqCbStats *)(cb + L.off_stats);
    const float *src = (VQ_IS_COS(metric)) ? (const float *)(cb + L.off_eexact) : e;
    const float *en = (const float *)(cb + L.off_en);
    // every block reduces the statistics partials (16 KiB, L2-resident) to the global maxima -> the common scale
    VqCbStats g_st;
    {
        const f32x4 *part = (const f32x4 *)(cb + L.off_part1);
        float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;
        for (int64_t i = threadIdx.x; i < L.nblk1; i += 256) {
            f32x4 v = part[i];
            a0 = fmaxf(a0, v[0]); a1 = fmaxf(a1, v[1]); a2 = fmaxf(a2, v[2]); a3 = fmaxf(a3, v[3]);
        }
        a0 = wave_max(a0); a1 = wave_max(a1); a2 = wave_max(a2); a3 = wave_max(a3);
        a0 = block_max4(a0, red4); a1 = block_max4(a1, red4); a2 = block_max4(a2, red4); a3 = block_max4(a3, red4);
        g_st.maxabs_bits = __float_as_uint(a0); g_st.e2max_bits = __float_as_uint(a1);
        g_st.enmax_bits = (VQ_IS_L2(metric)) ? __float_as_uint(a2) : 0u;
        g_st.nonfinite = a3 > 0.0f ? 1u : 0u;
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            st->maxabs_bits = g_st.maxabs_bits; st->e2max_bits = g_st.e2max_bits; st->enmax_bits = g_st.enmax_bits;
            st->nonfinite = g_st.nonfinite; st->metric = metric; st->finalized = 1u;
            st->r2max_bits = 0u; st->eh2max_bits = 0u;       // (the image's own maxima live in the slots: cb_stats_view)
        }
    }
    const float se = cb_scale(&g_st), inv = 1.0f / se;
    const int64_t k = tile * VQ_TILE_CODES + r;
    char *stage_base = cb + L.off_frag + stage * L.stage_bytes;
    float r2 = 0.0f, h2 = 0.0f;
    // pieces of 8 dims: k-step of 32 dims s32 = piece/4, quarter q4 = piece%4; the tile's two 16-code halves go
    // to chunks (s32, 0) and (s32, 1); within a chunk lane = q4*16 + (code & 15)
    for (int piece = g; piece < L.nstep * 2; piece += 8) {
        const int s = piece >> 2, q4 = piece & 3;
        const int d0 = 32 * s + 8 * q4;
        half8 o;
        if (k < K && d0 < D) {
            float v[8];
            if (d0 + 8 <= D && (D % 4) == 0) { load8<0>(src, k * D + d0, v); }
            else {
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = (d0 + j < D) ? src[k * D + d0 + j] : 0.0f;
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                _Float16 q = to_f16_ftz(v[j] * se);
                float back = (float)q * inv, res = v[j] - back;
                r2 = fmaf(res, res, r2); h2 = fmaf(back, back, h2);
                o[j] = q;
            }
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = (_Float16)0.0f;
        }
        *(half8 *)(stage_base + (int64_t)(ti * L.nstep + 2 * s + (r >> 4)) * VQ_CHUNK_BYTES + (q4 * 16 + (r & 15)) * 16) = o;
    }
    // aux chunk slice of this tile: -se*|e_k|^2/2 for its 32 codes (padded codes: a large FINITE negative score;
    // -inf with the register index or-ed into its mantissa would be a signalling NaN and poison v_max_f32)
    if (g == 0) {
        float v = (k < K) ? (-0.5f * en[k]) * se : -3.0e38f;
        *(float *)(stage_base + (int64_t)L.tps * L.nstep * VQ_CHUNK_BYTES + (ti * 32 + r) * 4) = v;
    }
    red[0][g][r] = r2; red[1][g][r] = h2;
    __syncthreads();
    float a = 0.0f, b = 0.0f;
    if (g == 0) {
#pragma unroll
        for (int i = 0; i < 8; ++i) { a += red[0][i][r]; b += red[1][i][r]; }
    }
    bool bad = !isfinite(a) || !isfinite(b);
    a = wave_max(a); b = wave_max(b);       // waves 1..3 contribute zeros
    a = block_max4(a, red4); b = block_max4(b, red4);
    float badf = block_max4(__any(bad) ? 1.0f : 0.0f, red4);
    // max fp16 residual / image norm / non-finite flag of this tile go into one of VQ_CB_SLOTS slots (zeroed by cb_stats_kernel,
    // the launch before) with fire-and-forget atomics — a 128-byte line per slot, K/512 atomics per word — and every consumer
    // wave folds the 16 slots itself (cb_stats_view): the image is complete when its launch is, no consumer kernel has to run
    // a fold first, so the token side can be prepared in the same launch as the codebook statistics (pre_kernel).
    // (Tried first: an arrival ticket with the last workgroup folding per-block partials — its agent-scope release writes the
    // XCD's dirty L2 lines, i.e. the image, back: 8.5 -> 19 us at K = 16 384, D = 256; and read-then-atomic on three header
    // words — two dependent device-scope round trips at the end of every workgroup: 15 us.)
    if (threadIdx.x == 0) {
        uint32_t *slot = (uint32_t *)(cb + L.off_stats + 256 + (blockIdx.x % VQ_CB_SLOTS) * 128);
        atomicMax(&slot[0], __float_as_uint(a));
        atomicMax(&slot[1], __float_as_uint(b));
        if (badf > 0.0f) atomicMax(&slot[2], 1u);
    }
}

// ------------------------------------------------------------------------------------------------
// token preparation: fp16 (flush-to-zero) fragment-major image of x, |xh|^2 and |x - xh|^2 per row
// ------------------------------------------------------------------------------------------------
// One 256-thread block per 32 tokens.  Image chunk (tile of 16 tokens, k-step s of 32 dims) holds for lane l the dims
// 32s + 8(l>>4) .. +8 of token tile*16 + (l&15): the B operand of v_mfma_f32_16x16x32_f16.
// XNORM (cosine through vqhip_encode): the rows are first normalised exactly as normalize_rows_kernel does — the
// oracle-order |x|^2 this kernel computes anyway is that kernel's sum — written to `xq` as fp32, and everything else
// (image, |xh|^2, residual, |x|^2) is taken from the normalised rows: one launch less, one pass over x less.
// NCHW (vqhip_encode_map: the latents arrive as the feature map [B, D, HW] the encoder / connector produced, the
// reference's 'b c h w -> (b h w) c' of models/base.py:124 is folded into this kernel): every 64-dim x 32-token tile is read
// with the tokens along the lanes (coalesced 64/128-byte segments per channel), turned through LDS, and from there on the
// kernel is the token-major one; the rows it has in registers anyway are also written out token-major (`xrows`, in the
// input's dtype; cosine: additionally the normalised fp32 rows `xq`) for the exact re-rank, the gather and the backward.
template <int DT, bool XNORM = false, bool NCHW = false>
__device__ __forceinline__ void x_prep_body(int64_t blk, const void *__restrict__ x, int64_t N, int D, int nstep,
                                            char *__restrict__ ximg, float *__restrict__ xh2,
                                            float *__restrict__ rho2, float *__restrict__ xn,
                                            int *__restrict__ counters, int *__restrict__ arrive, int narrive,
                                            float *__restrict__ xq, float eps, int xround = 0,
                                            int32_t *__restrict__ hist_zero = nullptr, int64_t hist_len = 0,
                                            int64_t nblocks = 1, int64_t hw = 0, void *__restrict__ xrows = nullptr) {
    __shared__ float red[2][8][32];
    // vqhip_encode(VQHIP_ENCODE_ZERO_HIST): the code-hit histogram the later kernels of this call add into starts from zero
    if (hist_zero != nullptr)
        for (int64_t i = blk * 256 + threadIdx.x; i < hist_len; i += nblocks * 256) hist_zero[i] = 0;
    __shared__ float part[64][32];   // the 64 interleaved partial sums of |x|^2 (oracle order), per token
    __shared__ float den_s[32];
    __shared__ float tile[NCHW ? 64 : 1][33];     // NCHW: 64 dims x 32 tokens of the map, turned here
    if (blk == 0 && threadIdx.x < 8) counters[threadIdx.x] = 0;   // housekeeping for the later kernels of this call (stream-ordered)
    // arrival counters of the proposal kernel's token blocks (at most one per 128 tokens: 4 blocks of this kernel)
    if (arrive != nullptr && threadIdx.x == 0 && blk < narrive) arrive[blk] = 0;
    const int r = threadIdx.x & 31, g = threadIdx.x >> 5;
    const int64_t t = blk * 32 + r;
    const bool tvalid = t < N;
    const int64_t trow = tvalid ? t : (N - 1);
    const int ns32 = nstep >> 1;
    // NCHW: element (token trow, dim d) lives at map_base + d * hw
    const int64_t map_base = NCHW ? ((trow / hw) * (int64_t)D * hw + (trow % hw)) : 0;
    // the 64 dims [64 it, 64 it + 64) of this block's 32 tokens -> tile (uniform: every thread of the block calls it)
    // fast form of the tile load: the block's 32 tokens are 32 consecutive positions of ONE image (hw % 32 == 0, which also
    // keeps every 8-token group 16/32-byte aligned): thread (channel c = tid >> 2, group tg = tid & 3) loads 8 consecutive
    // tokens of its channel with one (bf16) or two (fp32) 16-byte loads — a wave-instruction covers 16 channels x 64/128 B
    const int niter_stage = (ns32 * 4 + 7) / 8;
    const bool vec_tile = NCHW && (hw % 32) == 0 && blk * 32 + 32 <= N;
    const int64_t tile_base = NCHW ? (((blk * 32) / (hw > 0 ? hw : 1)) * (int64_t)D * hw + ((blk * 32) % (hw > 0 ? hw : 1))) : 0;
    // (the tile after the one being consumed is already on its way: its loads are issued right behind the barrier that
    //  publishes the current tile, so the map's latency hides behind the fp16 conversion work of the current one)
    typename RawVec<DT>::type ahead;
    int ahead_it = -1;
    auto stage = [&](int it) {
        if constexpr (NCHW) {
            __syncthreads();                                  // the previous tile has been consumed
            if (vec_tile) {
                const int c = threadIdx.x >> 2, tg = threadIdx.x & 3;
                if (ahead_it != it && 64 * it + c < D) ahead = RawVec<DT>::load(x, tile_base + (int64_t)(64 * it + c) * hw + 8 * tg);
                float v[8];
                if (64 * it + c < D) RawVec<DT>::unpack(ahead, v);
                else {
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = 0.0f;
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) tile[c][8 * tg + j] = v[j];
                __syncthreads();
                ahead_it = it + 1;
                if (ahead_it < niter_stage && 64 * ahead_it + c < D)
                    ahead = RawVec<DT>::load(x, tile_base + (int64_t)(64 * ahead_it + c) * hw + 8 * tg);
                return;
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int dl = g + 8 * j, d = 64 * it + dl;   // a wave-instruction: 2 channels x 32 consecutive tokens
                    tile[dl][r] = (tvalid && d < D) ? load_elem<DT>(x, map_base + (int64_t)d * hw) : 0.0f;
                }
            }
            __syncthreads();
        }
    };
    // 8 consecutive dims of this thread's token for `piece` (dims 8*piece ..): from memory, or from the staged tile
    auto fetch = [&](int piece, float (&v)[8]) {
        if constexpr (NCHW) {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = tile[(8 * piece + j) & 63][r];
        } else {
            load8<DT>(x, trow * D + 32 * (piece >> 2) + 8 * (piece & 3), v);
        }
    };
    const int npieces = ns32 * 4, niter = (npieces + 7) / 8;
    float s_h = 0.0f, s_r = 0.0f;
    // thread g sees exactly the dims with d mod 64 in [8g, 8g+8), in increasing d: partial j = 8g + jj of the oracle's
    // |x|^2 (64 interleaved fma chains, then the halving tree)
    float pn[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) pn[j] = 0.0f;
    float den = 1.0f;
    constexpr int KEEP = 4;              // pieces a thread keeps in registers between the two passes (D <= 256)
    float kept[KEEP][8];
    const bool keep = XNORM && ns32 * 4 <= KEEP * 8;
    if constexpr (XNORM) {
#pragma unroll
        for (int i = 0; i < KEEP; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) kept[i][j] = 0.0f;
#pragma unroll
        for (int i = 0; i < KEEP; ++i) {
            const int piece = g + 8 * i;
            const int d0 = 32 * (piece >> 2) + 8 * (piece & 3);
            if (i < niter) stage(i);
            if (piece < npieces && tvalid && d0 < D) {
                fetch(piece, kept[i]);
#pragma unroll
                for (int j = 0; j < 8; ++j) pn[j] = fmaf(kept[i][j], kept[i][j], pn[j]);
            }
        }
        for (int it = KEEP; it < niter; ++it) {
            const int piece = g + 8 * it;
            const int d0 = 32 * (piece >> 2) + 8 * (piece & 3);
            stage(it);
            if (piece < npieces && tvalid && d0 < D) {
                float v[8];
                fetch(piece, v);
#pragma unroll
                for (int j = 0; j < 8; ++j) pn[j] = fmaf(v[j], v[j], pn[j]);
            }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) part[8 * g + j][r] = pn[j];
        __syncthreads();
        if (g == 0) {                    // halving tree 32, 16, ..., 1 over the partials (normalize_rows_kernel's order)
            float q[32];
#pragma unroll
            for (int j = 0; j < 32; ++j) q[j] = part[j][r] + part[j + 32][r];
#pragma unroll
            for (int off = 16; off >= 1; off >>= 1)
#pragma unroll
                for (int j = 0; j < 16; ++j)
                    if (j < off) q[j] = q[j] + q[j + off];
            const float nrm = sqrtf(q[0]);
            den_s[r] = (nrm < eps) ? eps : nrm;
        }
        __syncthreads();
        den = den_s[r];
#pragma unroll
        for (int j = 0; j < 8; ++j) pn[j] = 0.0f;
    }
    for (int it = 0; it < niter; ++it) {
        const int piece = g + 8 * it;
        const int s = piece >> 2, q4 = piece & 3;
        const int d0 = 32 * s + 8 * q4;
        if (!(XNORM && keep)) stage(it);               // (kept in registers: the map is read once)
        if (piece >= npieces) continue;
        half8 f;
        if (tvalid && d0 < D) {
            float v[8];
            bool have = false;
            if constexpr (XNORM) {
                if (keep) {                        // second pass over registers instead of memory
#pragma unroll
                    for (int i = 0; i < KEEP; ++i)
                        if (piece == g + 8 * i) {
#pragma unroll
                            for (int j = 0; j < 8; ++j) v[j] = kept[i][j];
                            have = true;
                        }
                }
            }
            if (!have) fetch(piece, v);
            if constexpr (NCHW) {                  // the token-major rows as given, in the input's own dtype (exact: a copy)
                if (DT == 0) {
                    float *o = (float *)xrows + trow * D + d0;
                    *(f32x4 *)o = f32x4{v[0], v[1], v[2], v[3]};
                    *(f32x4 *)(o + 4) = f32x4{v[4], v[5], v[6], v[7]};
                } else {
                    uint4 o;
                    o.x = (__float_as_uint(v[0]) >> 16) | (__float_as_uint(v[1]) & 0xFFFF0000u);
                    o.y = (__float_as_uint(v[2]) >> 16) | (__float_as_uint(v[3]) & 0xFFFF0000u);
                    o.z = (__float_as_uint(v[4]) >> 16) | (__float_as_uint(v[5]) & 0xFFFF0000u);
                    o.w = (__float_as_uint(v[6]) >> 16) | (__float_as_uint(v[7]) & 0xFFFF0000u);
                    *(uint4 *)((uint16_t *)xrows + trow * D + d0) = o;
                }
            }
            if constexpr (XNORM) {
#pragma unroll
                for (int j = 0; j < 8; ++j) { v[j] = v[j] / den; if (xround) v[j] = bf16_rne(v[j]); }
                *(f32x4 *)(xq + trow * D + d0) = f32x4{v[0], v[1], v[2], v[3]};
                *(f32x4 *)(xq + trow * D + d0 + 4) = f32x4{v[4], v[5], v[6], v[7]};
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                _Float16 q = to_f16_ftz(v[j]);
                float b = (float)q, res = v[j] - b;
                s_h = fmaf(b, b, s_h); s_r = fmaf(res, res, s_r);
                pn[j] = fmaf(v[j], v[j], pn[j]);
                f[j] = q;
            }
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) f[j] = (_Float16)0.0f;
        }
        *(half8 *)(ximg + ((blk * 2 + (r >> 4)) * ns32 + s) * (int64_t)VQ_CHUNK_BYTES + (q4 * 16 + (r & 15)) * 16) = f;
    }
    red[0][g][r] = s_h; red[1][g][r] = s_r;
#pragma unroll
    for (int j = 0; j < 8; ++j) part[8 * g + j][r] = pn[j];
    __syncthreads();
    if (g == 0 && tvalid) {
        float a = 0.0f, b = 0.0f;
#pragma unroll
        for (int i = 0; i < 8; ++i) { a += red[0][i][r]; b += red[1][i][r]; }
        xh2[t] = a; rho2[t] = b;
    }
    if (g == 1 && tvalid) {          // halving tree 32, 16, ..., 1 over the partials
        float q[32];
#pragma unroll
        for (int j = 0; j < 32; ++j) q[j] = part[j][r] + part[j + 32][r];
#pragma unroll
        for (int off = 16; off >= 1; off >>= 1)
#pragma unroll
            for (int j = 0; j < 16; ++j)
                if (j < off) q[j] = q[j] + q[j + off];
        xn[t] = q[0];
    }
}
template <int DT>
__global__ __launch_bounds__(256) void x_prep_kernel(const void *__restrict__ x, int64_t N, int D, int nstep,
                                                     char *__restrict__ ximg, float *__restrict__ xh2,
                                                     float *__restrict__ rho2, float *__restrict__ xn,
                                                     int *__restrict__ counters, char *cb, VqCbLayout L,
                                                     int *__restrict__ arrive = nullptr, int narrive = 0) {
    x_prep_body<DT, false>(blockIdx.x, x, N, D, nstep, ximg, xh2, rho2, xn, counters, arrive, narrive, nullptr, 0.0f);
}
// vqhip_encode / vqhip_col_argmin: the codebook statistics and the token side in ONE launch (they are independent;
// the image kernel that follows needs the former, the proposal kernel both)
template <int DT, bool XNORM, bool NCHW = false>
__global__ __launch_bounds__(256) void pre_kernel(const float *e, int64_t K, int metric, char *cb, VqCbLayout L, int nblk_stats,
                                                  const void *__restrict__ x, int64_t N, int D, int nstep,
                                                  char *__restrict__ ximg, float *__restrict__ xh2,
                                                  float *__restrict__ rho2, float *__restrict__ xn,
                                                  int *__restrict__ counters, int *__restrict__ arrive, int narrive,
                                                  float *__restrict__ xq, float eps, int32_t *__restrict__ hist_zero,
                                                  int64_t hw = 0, void *__restrict__ xrows = nullptr) {
    if ((int)blockIdx.x < nblk_stats) cb_stats_body(blockIdx.x, e, K, D, metric, cb, L);
    else x_prep_body<DT, XNORM, NCHW>((int64_t)blockIdx.x - nblk_stats, x, N, D, nstep, ximg, xh2, rho2, xn, counters, arrive, narrive, xq, eps,
                                      VQ_IS_BF16(metric) ? 1 : 0, hist_zero, K, (int64_t)gridDim.x - nblk_stats, hw, xrows);
}

// ------------------------------------------------------------------------------------------------
// fp16 MFMA proposal pass
// ------------------------------------------------------------------------------------------------
struct Top2 { float v1, v2, v3; uint32_t c1, c2; };
__device__ __forceinline__ float row_margin(const VqCbStats *st, int Dp, int metric, float X2, float R2, float *bf16_part = nullptr);

// where the decision stage writes (one struct: the proposal kernel carries it as a single argument)
struct VqDecideOut {
    int64_t *idx; int32_t *hist;
    int *rescan_list, *multi_list, *exact_list, *counters;
    u64 *keys; float *thr_out; int *rescan_cnt;
    int *arrive;            // one arrival counter per token block of the proposal kernel (zeroed by x_prep_kernel)
    const int *n_dev;       // nullable DEVICE row count: only rows [0, min(N, *n_dev)) are live (vqhip_col_argmin_rows:
                            // the launch is sized for a capacity, the actual number of listed codes stays on the device)
};
template <bool AGENT>
__device__ __forceinline__ void decide_rows(int64_t n, bool oob, const VqCbStats *st, int Dp, int metric, int nslices,
                                            const float *rec, const float *xh2, const float *rho2, int64_t Np,
                                            const VqDecideOut &o, int *wcount, int *wbase);

__device__ __forceinline__ void top_insert(Top2 &t, float v, uint32_t c) {
    if (v > t.v1) { t.v3 = fmaxf(t.v3, t.v2); t.v2 = t.v1; t.c2 = t.c1; t.v1 = v; t.c1 = c; }
    else if (v > t.v2) { t.v3 = fmaxf(t.v3, t.v2); t.v2 = v; t.c2 = c; }
    else t.v3 = fmaxf(t.v3, v);
}

__device__ __forceinline__ void top_merge_lane(Top2 &t, int xor_mask) {   // fold the partner lane's record into t
    Top2 o;
    o.v1 = __shfl_xor(t.v1, xor_mask, 64); o.v2 = __shfl_xor(t.v2, xor_mask, 64); o.v3 = __shfl_xor(t.v3, xor_mask, 64);
    o.c1 = __shfl_xor(t.c1, xor_mask, 64); o.c2 = __shfl_xor(t.c2, xor_mask, 64);
    if (o.c1 != 0xFFFFFFFFu) top_insert(t, o.v1, o.c1);
    if (o.c2 != 0xFFFFFFFFu) top_insert(t, o.v2, o.c2);
    t.v3 = fmaxf(t.v3, o.v3);
}

// code row inside a 32-code tile for accumulator element e = 4*c + reg of lane l (v_mfma_f32_16x16x32: row = 4(l>>4)+reg)
__device__ __forceinline__ int tile_row16(int e, int lane) { return 16 * (e >> 2) + 4 * (lane >> 4) + (e & 3); }

// One workgroup = WAVES waves x TT token tiles of 16 tokens held in registers as MFMA B fragments for the whole kernel;
// it streams one slice of the codebook image through an LDS ring of NBUF stages (global_load_lds; four stages filled two
// ahead up to D = 256, a double buffer above) and
// keeps, per lane and token, the best score with its tile / register and the runner-up value.
// Scores are a_k = se*(xh . eh_k) - se*|e_k|^2/2 (the accumulator is initialised with the aux value).
// MFMA shape 16x16x32 (the chip holds a higher clock on it than on 32x32x16: +8..11 % measured on this kernel).
// The epilogue of tile t-1 (3 VALU per element) is spread over the MFMAs of tile t (two accumulator sets ping-pong).
//
// FILTER (small D, where 3 VALU per score against D/8 MFMA cycles per score make the kernel VALU-issue-bound): the 8
// elements a lane holds per (token tile, code tile) first go through a 4-instruction maximum (v_max3) and ONE compare
// against the lane's threshold; the per-element update runs only if some lane of the wave reaches its threshold
// (wave-uniform branch).  The threshold of a token is (best score any of its four lanes has seen) - (the row's margin
// m, the very number the decision kernel uses), refreshed once per stage.  A skipped score s satisfies
// s < best_so_far - m <= final best - m = the decision threshold, so it is strictly outside the candidate set the
// margin defines and needs neither identification nor a bound in the record; every score within the margin of the
// running best still goes through the exact per-element update.  Rows without a usable margin never skip.
//
// NOAUX (cosine / dot product: no |e|^2 term, the aux chunk is all zeros except for the padding codes of the very last
// stage): the accumulators start from the inline constant 0 and the two 16-byte aux reads per code tile — half of the
// LDS read traffic at D = 32 — are issued only in `pad_stage` (-1: the codebook fills its last stage).
//
// GROUPS (with FILTER; D <= 32, where the per-element update of the tiles that fail the skip test was the larger half of
// the VALU work): see "group record" in the loop and "group records -> code records" after it.
template <int NSTEP, int TT, int WAVES, int TPS, int NBUF = 2, bool FILTER = false, bool STREAMK = false, bool NOAUX = false,
          bool GROUPS = false>
__global__ __launch_bounds__(WAVES * 64, (FILTER && NSTEP <= 2) ? 4 : WAVES / 4) void coarse_kernel(
    const char *__restrict__ ximg, int64_t N, const char *__restrict__ frag, int64_t nstages, int nslices,
    float *__restrict__ rec, int64_t Np, const VqCbStats *__restrict__ cbst, const float *__restrict__ xh2,
    const float *__restrict__ rho2, int Dp, int metric, VqDecideOut dec, int pad_stage, int tpb) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    static_assert(NSTEP % 2 == 0, "16x16x32 layout: 32-dim k-steps");
    static_assert(!GROUPS || FILTER, "group records are a form of the filtered epilogue");
    constexpr bool GBRANCH = TT >= VQ_GROUP_BRANCH_MIN_TT;
    constexpr bool PIPE = (TPS % 2) == 0;                // epilogue of tile t-1 in the MFMA shadow of tile t (ping-pong by parity)
    constexpr int NS32 = NSTEP / 2;                      // k-steps of 32 dims
    constexpr int NCH = TPS * NSTEP + VQ_AUX_CHUNKS(TPS);   // chunks per stage (2 per k-step and tile, + aux)
    constexpr int STAGE_BYTES = NCH * VQ_CHUNK_BYTES;
    constexpr int BM = WAVES * TT * 16;
    constexpr int NE = 8;                                // accumulator elements per lane, token tile and code tile
#ifndef VQ_SCALAR_WAVE
#define VQ_SCALAR_WAVE 1
#endif
    // the wave index as a SCALAR: the compiler cannot tell that threadIdx.x >> 6 is wave-uniform, and everything indexed by it
    // (the LDS-DMA request loop above all) otherwise runs as a divergent loop.  Measured (profiles/r02_scalar_wave.txt):
    // +1.2 % at D = 256 (3.129 -> 3.093 ms at 524 288 tokens), +0.5-1 % at D >= 128, but -1.5..2.5 % at D = 32
    // (the kernel is at its SGPR limit there): D <= 32 keeps the vector form
    const int lane = threadIdx.x & 63;
    const int wave = (VQ_SCALAR_WAVE && NSTEP > 2) ? __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) : (int)(threadIdx.x >> 6);
    const int64_t ntt = (N + 31) / 32 * 2;               // 16-token tiles in the fp16 token image (of the launch's capacity)
    if (!STREAMK && dec.n_dev != nullptr) {              // device-side row count: token blocks past it have nothing to do
        const int64_t nd = *dec.n_dev;
        N = nd < N ? nd : N;
        if ((int64_t)(blockIdx.x / nslices) * tpb * 16 >= N) return;
    }
    // Work assignment.  STREAMK false: workgroup = (token block tb, codebook slice sl of nslices), one segment.
    // STREAMK true (small D): the (token block x stage) space, block-major, is cut into gridDim.x equal ranges — every
    // CU gets the same share whatever the number of token blocks — and a workgroup walks its range as one or two
    // segments (tail of one block, head of the next).  A block is then covered by at most `nslices` consecutive
    // workgroups; the piece index within the block is the record slot, unused slots are filled with "nothing here".
    // tpb: 16-token tiles per workgroup, <= WAVES*TT (the host picks it so that the workgroups fill whole rounds of the
    // chip: launch_coarse).  Waves past it only help filling the ring; tiles past it belong to the next workgroup.
    const bool wave_active = wave * TT < tpb;
    const int64_t U = ((N + BM - 1) / BM) * nstages;
    int64_t u_next = STREAMK ? ((int64_t)blockIdx.x * U) / gridDim.x : 0;
    const int64_t u_end = STREAMK ? ((int64_t)(blockIdx.x + 1) * U) / gridDim.x : 1;
  for (bool once = true; STREAMK ? (u_next < u_end) : once; once = false) {
    int sl;
    int64_t tb, st0, st1;
    if constexpr (!STREAMK) {
        sl = blockIdx.x % nslices; tb = blockIdx.x / nslices;
        st0 = (nstages * sl) / nslices; st1 = (nstages * (sl + 1)) / nslices;
    } else {
        tb = u_next / nstages; st0 = u_next % nstages;
        st1 = (st0 + (u_end - u_next) < nstages) ? st0 + (u_end - u_next) : nstages;
        int64_t g = blockIdx.x;                          // first workgroup of this block: largest g with start(g) <= tb*nstages
        while (g > 0 && (g * U) / gridDim.x > tb * nstages) --g;
        sl = (int)(blockIdx.x - g);
        u_next += st1 - st0;
    }

    // ---- prologue: this wave's token fragments straight from the fragment-major fp16 image ----
    half8 xf[TT][NS32];
#pragma unroll
    for (int t = 0; t < TT; ++t) {
        int64_t tt = tb * tpb + wave * TT + t;
        tt = tt < ntt ? tt : ntt - 1;                    // out-of-range tiles read a valid tile and are never written
        const char *src = ximg + tt * (int64_t)(NS32 * VQ_CHUNK_BYTES) + lane * 16;
#pragma unroll
        for (int s = 0; s < NS32; ++s) xf[t][s] = *(const half8 *)(src + s * VQ_CHUNK_BYTES);
    }

    float b1[TT], b2[TT], th[TT], mg[TT];
    uint32_t t1[TT];
#pragma unroll
    for (int t = 0; t < TT; ++t) { b1[t] = -INFINITY; b2[t] = -INFINITY; th[t] = -INFINITY; mg[t] = INFINITY; t1[t] = 0; }
    static_assert(!FILTER || PIPE, "the filtered epilogue is written for the ping-pong form");
    float sc0 = 0.0f, sc1 = 0.0f, sc2 = 0.0f;     // destinations of the asm maxima: live across the whole loop (see vmax3_into)
    if constexpr (FILTER) {
        const VqCbStats stv = cb_stats_view(cbst);
#pragma unroll
        for (int t = 0; t < TT; ++t) {
            int64_t tokn = (tb * tpb + wave * TT + t) * 16 + (lane & 15);
            tokn = tokn < N ? tokn : N - 1;
            const float m = row_margin(&stv, Dp, metric, xh2[tokn], rho2[tokn]);
            mg[t] = (m > 0.0f) ? m : INFINITY;                 // no usable bound: threshold -inf, nothing is skipped
        }
    }

    auto issue_stage = [&](int64_t st, int buf) {
        const char *src = frag + st * (int64_t)STAGE_BYTES;
        char *dstb = lds + buf * STAGE_BYTES;
        for (int c = wave; c < NCH; c += WAVES)
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void *)(src + c * VQ_CHUNK_BYTES + lane * 16),
                (__attribute__((address_space(3))) void *)(dstb + c * VQ_CHUNK_BYTES), 16, 0, 0);
    };

    // NBUF == 4: ring of four stages filled two ahead, and the second half of the waves (the SIMD partners of the
    // first half) runs one stage behind.  Measured at D = 256 against the double-buffered form with stages twice the
    // size: ring and look-ahead -6 %, the lag another -2 % (lagging the odd waves instead: -1 % less; three ahead
    // without lag: +9 % slower) — MI355X guide, 'Two waves per SIMD', item 9
    constexpr int AHEAD = NBUF >= 3 ? 2 : 1;                 // NBUF == 3 (large D): two ahead, no lag
    const int lag = (NBUF >= 4 && wave >= WAVES / 2) ? 1 : 0;
    if (st0 < st1) issue_stage(st0, 0);
    if (AHEAD >= 2 && st0 + 1 < st1) issue_stage(st0 + 1, 1);
    __syncthreads();   // drains the LDS-DMA (vmcnt(0)) and makes it visible to every wave

    f32x4 accA[2][TT], accB[2][TT];
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int t = 0; t < TT; ++t)
#pragma unroll
            for (int q = 0; q < 4; ++q)   // "previous tile" of the very first tile: never wins (group records: never even registers)
                accB[c][t][q] = GROUPS ? -INFINITY : -3.0e38f;

#ifdef VQ_STAMPS     // diagnostic build only (MI355X guide, In-kernel stamps): shares of the stage loop, printed by a few waves
#define VQ_STAMP(t) do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
    unsigned long long ts_loop0, ts_a, ts_b, ts_c, acc_issue = 0, acc_body = 0, acc_bar = 0;
    VQ_STAMP(ts_loop0);
#endif
    for (int64_t it = st0; it < st1 + (NBUF >= 4 ? 1 : 0); ++it) {
#ifdef VQ_STAMPS
        VQ_STAMP(ts_a);
#endif
#ifndef VQ_EXP_NODMA           // (timing-only experiment builds, results garbage: profiles/r02_mfma32x32_d32.txt)
        if (it + AHEAD < st1) issue_stage(it + AHEAD, (int)((it + AHEAD - st0) % NBUF));
#endif
#ifdef VQ_STAMPS
        VQ_STAMP(ts_b); acc_issue += ts_b - ts_a;
#endif
        const int64_t st = it - lag;
        if (st < st0 || st >= st1 || !wave_active) { __syncthreads(); continue; }
        const int buf = (int)((st - st0) % NBUF);
        const char *base = lds + buf * STAGE_BYTES;
        const char *aux = base + TPS * NSTEP * VQ_CHUNK_BYTES;
      // the tiles of one stage; WITH_AUX false: accumulators start from the constant 0 (no aux read)
      auto run_stage = [&](auto with_aux_tag) __attribute__((always_inline)) {
        constexpr bool WITH_AUX = decltype(with_aux_tag)::value;
#pragma unroll
        for (int ti = 0; ti < TPS; ++ti) {
            f32x4 (&cur)[2][TT] = (PIPE && (ti & 1)) ? accB : accA;
            f32x4 (&prv)[2][TT] = (PIPE && (ti & 1)) ? accA : accB;
            // accumulator init = -se*|e|^2/2 of this lane's code rows 16c + 4(l>>4) + {0..3}
            if constexpr (WITH_AUX) {
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    f32x4 a4 = *(const f32x4 *)(aux + (ti * 32 + 16 * c + 4 * (lane >> 4)) * 4);
#pragma unroll
                    for (int t = 0; t < TT; ++t) cur[c][t] = a4;
                }
            }
            uint32_t old[TT];
#pragma unroll
            for (int t = 0; t < TT; ++t) old[t] = __float_as_uint(b1[t]);
            // A fragments PF chunks ahead of the MFMAs that consume them (ring of PF+1 register sets);
            // chunk ch = 2*s32 + c feeds the TT MFMAs of code half c at k-step s32
            constexpr int PF = NSTEP <= 32 ? 1 : (NSTEP <= 48 ? 2 : 4);   // deeper where a chunk feeds fewer MFMAs
            half8 af[PF + 1];
#pragma unroll
            for (int i = 0; i < PF; ++i)
                if (i < NSTEP) af[i] = *(const half8 *)(base + (ti * NSTEP + i) * VQ_CHUNK_BYTES + lane * 16);
#pragma unroll
            for (int ch = 0; ch < NSTEP; ++ch) {
                if (ch + PF < NSTEP)
                    af[(ch + PF) % (PF + 1)] = *(const half8 *)(base + (ti * NSTEP + ch + PF) * VQ_CHUNK_BYTES + lane * 16);
#pragma unroll
                for (int t = 0; t < TT; ++t) {
                    if constexpr (!WITH_AUX) {
                        if (ch < 2) {                      // first k-step of this code half: C = inline constant 0
                            cur[ch & 1][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[ch % (PF + 1)], xf[t][ch >> 1], f32x4{0.0f, 0.0f, 0.0f, 0.0f}, 0, 0, 0);
                            continue;
                        }
                    }
                    cur[ch & 1][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[ch % (PF + 1)], xf[t][ch >> 1], cur[ch & 1][t], 0, 0, 0);
                }
                if constexpr (FILTER) {
                    // token tile t of the previous code tile: maximum of its 8 elements, one compare, wave-uniform skip
                    const uint32_t tgp = (uint32_t)(st * TPS + ti) - 1u;
#pragma unroll
                    for (int i = 0; i < (TT + NSTEP - 1) / NSTEP; ++i) {
                        constexpr int EVERY = (NSTEP / TT) > 0 ? NSTEP / TT : 1;       // TT < NSTEP: one token tile every EVERY chunks
                        const int t = (TT >= NSTEP) ? ch * (TT / NSTEP) + i : ((ch % EVERY == 0) ? ch / EVERY : -1);
                        if (t >= 0 && t < TT) {
                            // (ordered behind all TT MFMAs of this chunk: >= TT MFMAs after the previous tile's last one)
                            float after[TT];
#pragma unroll
                            for (int u = 0; u < TT; ++u) after[u] = cur[ch & 1][u][0];
                            tile_max8<TT>(sc0, sc1, sc2, prv[0][t], prv[1][t], after);
                            if constexpr (GROUPS) {
                                // group record: the lane keeps the best GROUP maximum (its 8 codes of one code tile),
                                // the tile it came from and the best maximum of any other group; which of the 8 codes
                                // it was is found after the stream by replaying that one tile (below)
                                if (!GBRANCH || __any(!(sc0 < th[t]))) {
                                    const float nb = vmax(b1[t], sc0);
                                    b2[t] = __builtin_amdgcn_fmed3f(b1[t], b2[t], sc0);
                                    t1[t] = (__float_as_uint(nb) != __float_as_uint(b1[t])) ? tgp : t1[t];
                                    b1[t] = nb;
                                }
                            } else if (__any(!(sc0 < th[t]))) {
                                const uint32_t was = __float_as_uint(b1[t]);
#pragma unroll
                                for (int e = 0; e < NE; ++e) {
                                    float v = __uint_as_float((__float_as_uint(prv[e >> 2][t][e & 3]) & 0xFFFFFFF0u) | (uint32_t)e);
                                    b2[t] = __builtin_amdgcn_fmed3f(b1[t], b2[t], v);
                                    b1[t] = vmax(b1[t], v);
                                }
                                t1[t] = (__float_as_uint(b1[t]) != was) ? tgp : t1[t];
                            }
                        }
                    }
                }
                // retire NE*TT/NSTEP accumulator elements of the previous tile per chunk step
                constexpr int TOTAL = NE * TT;
#pragma unroll
                for (int i = 0; PIPE && !FILTER && i < (TOTAL + NSTEP - 1) / NSTEP; ++i) {
                    constexpr int EVERY = (NSTEP / TOTAL) > 0 ? NSTEP / TOTAL : 1;   // TOTAL < NSTEP: one element every EVERY chunks
                    const int id = (TOTAL >= NSTEP) ? ch * (TOTAL / NSTEP) + i : ((ch % EVERY == 0) ? ch / EVERY : -1);
                    if (id >= 0 && id < TOTAL) {
                        const int t = id / NE, e = id % NE;
                        float v = __uint_as_float((__float_as_uint(prv[e >> 2][t][e & 3]) & 0xFFFFFFF0u) | (uint32_t)e);
                        b2[t] = __builtin_amdgcn_fmed3f(b1[t], b2[t], v);
                        b1[t] = vmax(b1[t], v);
                    }
                }
            }
            if constexpr (!PIPE) {   // large D: one tile per stage, epilogue in place (the SIMD's other wave covers it)
#pragma unroll
                for (int t = 0; t < TT; ++t)
#pragma unroll
                    for (int e = 0; e < NE; ++e) {
                        float v = __uint_as_float((__float_as_uint(cur[e >> 2][t][e & 3]) & 0xFFFFFFF0u) | (uint32_t)e);
                        b2[t] = __builtin_amdgcn_fmed3f(b1[t], b2[t], v);
                        b1[t] = vmax(b1[t], v);
                    }
            }
            if constexpr (!FILTER) {
                const uint32_t tgp = (uint32_t)(st * TPS + ti) - (PIPE ? 1u : 0u);  // tile the retired elements belong to
#pragma unroll
                for (int t = 0; t < TT; ++t)
                    t1[t] = (__float_as_uint(b1[t]) != old[t]) ? tgp : t1[t];
            }
        }
      };   // run_stage
#ifndef VQ_EXP_NOBODY
        if constexpr (NOAUX) {
            if (st == (int64_t)pad_stage) run_stage(std::true_type{}); else run_stage(std::false_type{});
        } else {
            run_stage(std::true_type{});
        }
#endif
        if constexpr (FILTER && (!GROUPS || GBRANCH)) {   // refresh the skip thresholds: best score among the token's four lanes, less the margin
#pragma unroll
            for (int t = 0; t < TT; ++t) th[t] = quad_rows_max(b1[t]) - mg[t];
        }
#ifdef VQ_STAMPS
        VQ_STAMP(ts_c); acc_body += ts_c - ts_b;
        __syncthreads();
        VQ_STAMP(ts_a); acc_bar += ts_a - ts_c;
        if (it + 1 >= st1 + (NBUF >= 4 ? 1 : 0) && lane == 0 && (blockIdx.x % 97) == 0 && (wave == 0 || wave == 5))
            printf("stamps block %d wave %d: stages %lld  loop %llu  issue %llu  body %llu  barrier %llu (cycles)\n", (int)blockIdx.x, wave,
                   (long long)(st1 - st0), ts_a - ts_loop0, acc_issue, acc_body, acc_bar);
        continue;
#endif
        // next stage landed (vmcnt(0)) and everybody is done reading this one.  (A barrier that keeps the pieces of the
        // stage requested in this iteration in flight — s_waitcnt vmcnt(pieces) instead of 0 — was measured: 1-3 %
        // slower at D <= 128 and 2x slower at D = 256, profiles/r02_ring_partial_wait.txt; the full drain stays.)
        __syncthreads();
    }
    // drain: epilogue of the last tile (odd parity: TPS is even, so it sits in accB)
    if (PIPE && st1 > st0) {
#pragma unroll
        for (int t = 0; t < TT; ++t) {
            const uint32_t old = __float_as_uint(b1[t]);
            if constexpr (GROUPS) {
                float g = accB[0][t][0];
#pragma unroll
                for (int e = 1; e < NE; ++e) g = __builtin_amdgcn_fmed3f(g, accB[e >> 2][t][e & 3], INFINITY);   // max, NaN-transparent like v_max
                b2[t] = __builtin_amdgcn_fmed3f(b1[t], b2[t], g);
                b1[t] = vmax(b1[t], g);
            } else {
#pragma unroll
                for (int e = 0; e < NE; ++e) {
                    float v = __uint_as_float((__float_as_uint(accB[e >> 2][t][e & 3]) & 0xFFFFFFF0u) | (uint32_t)e);
                    b2[t] = __builtin_amdgcn_fmed3f(b1[t], b2[t], v);
                    b1[t] = vmax(b1[t], v);
                }
            }
            t1[t] = (__float_as_uint(b1[t]) != old) ? (uint32_t)(st1 * TPS - 1) : t1[t];
        }
    }

    // ---- group records -> code records.  A lane whose best group can matter (its maximum is within the row's margin of
    // the best any of the token's four lanes holds) replays that one code tile — same fragments, same MFMA sequence per
    // accumulator, hence the very scores of the stream — and runs the per-element update (index bits, runner-up) on its 8
    // elements.  Everything else the lane has seen stays a value bound: raised to the largest value the index-bit form
    // of the same score can take (|low 4 mantissa bits| of slack, on the safe side for either sign).
    bool ident[TT];
#pragma unroll
    for (int t = 0; t < TT; ++t) ident[t] = true;
    if constexpr (GROUPS) {
        auto bound_up = [](float v) {
            const uint32_t b = __float_as_uint(v);
            return __uint_as_float((b & 0x80000000u) ? (b & 0xFFFFFFF0u) : (b | 0xFu));
        };
        constexpr int RB = NSTEP <= 2 ? VQ_REPLAY_BATCH : (NSTEP <= 4 ? 2 : 1);   // tiles replayed per round trip (registers)
#pragma unroll
        for (int t = 0; t < TT; ++t) {
            const float top = quad_rows_max(b1[t]);
            // (rows past N are image padding: never written, not replayed; the tile range check turns anything unexpected
            // — a score stream of NaNs, say — into "unidentified", which the decision stage answers with a second pass)
            const int64_t tokn = (tb * tpb + wave * TT + t) * 16 + (lane & 15);
            const bool need = (mg[t] < INFINITY) && (b1[t] > -INFINITY) && !(b1[t] < top - mg[t]) && tokn < N && wave * TT + t < tpb &&
                              t1[t] >= (uint32_t)(st0 * TPS) && t1[t] < (uint32_t)(st1 * TPS);
            float e1 = -INFINITY, e2 = -INFINITY;
            u64 todo = __ballot(need);
            while (todo) {
                uint32_t T[RB];
                u64 rest = todo;
#pragma unroll
                for (int i = 0; i < RB; ++i) {
                    const int l = rest ? (__ffsll((unsigned long long)rest) - 1) : (__ffsll((unsigned long long)todo) - 1);
                    T[i] = (uint32_t)__builtin_amdgcn_readlane((int)t1[t], l);
                    rest &= rest - 1;
                }
                half8 a[RB][NSTEP];
                f32x4 acc[RB][2];
#pragma unroll
                for (int i = 0; i < RB; ++i) {
                    const int64_t rst = T[i] / TPS;
                    const int rti = (int)(T[i] % TPS);
                    const char *sb = frag + rst * (int64_t)STAGE_BYTES;
#pragma unroll
                    for (int ch = 0; ch < NSTEP; ++ch)
                        a[i][ch] = *(const half8 *)(sb + (rti * NSTEP + ch) * VQ_CHUNK_BYTES + lane * 16);
#pragma unroll
                    for (int c = 0; c < 2; ++c) {
                        if (NOAUX && rst != (int64_t)pad_stage) acc[i][c] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
                        else acc[i][c] = *(const f32x4 *)(sb + TPS * NSTEP * VQ_CHUNK_BYTES + (rti * 32 + 16 * c + 4 * (lane >> 4)) * 4);
                    }
                }
                u64 done = 0;
#pragma unroll
                for (int i = 0; i < RB; ++i) {
#pragma unroll
                    for (int ch = 0; ch < NSTEP; ++ch)
                        acc[i][ch & 1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i][ch], xf[t][ch >> 1], acc[i][ch & 1], 0, 0, 0);
                    float w1 = -INFINITY, w2 = -INFINITY;
#pragma unroll
                    for (int e = 0; e < NE; ++e) {
                        float v = __uint_as_float((__float_as_uint(acc[i][e >> 2][e & 3]) & 0xFFFFFFF0u) | (uint32_t)e);
                        w2 = __builtin_amdgcn_fmed3f(w1, w2, v);
                        w1 = vmax(w1, v);
                    }
                    const bool mine = need && t1[t] == T[i];
                    e1 = mine ? w1 : e1; e2 = mine ? w2 : e2;
                    done |= __ballot(mine);
                }
                todo &= ~done;
            }
            const float other = bound_up(b2[t]);
            if (need) { b1[t] = e1; b2[t] = fmaxf(other, e2); }
            else { b2[t] = fmaxf(other, bound_up(b1[t])); }
            ident[t] = need;
        }
    }

    // ---- merge the four lanes that share a token; lanes 0..15 write one record per (token, slice) ----
#pragma unroll
    for (int t = 0; t < TT; ++t) {
        Top2 r; r.v1 = r.v2 = r.v3 = -INFINITY; r.c1 = r.c2 = 0xFFFFFFFFu;
        {
            uint32_t bits = __float_as_uint(b1[t]);
            uint32_t code = t1[t] * 32u + (uint32_t)tile_row16((int)(bits & 7u), lane);
            if (ident[t] && b1[t] > -INFINITY) top_insert(r, b1[t], code);
            r.v3 = fmaxf(r.v3, b2[t]);
        }
        top_merge_lane(r, 16);
        top_merge_lane(r, 32);
        const int64_t tokn = (tb * tpb + wave * TT + t) * 16 + (lane & 15);
        if (lane < 16 && tokn < N && wave * TT + t < tpb) {
            float *rp = rec + (int64_t)sl * VQ_REC_FIELDS * Np + tokn;
            rp[0] = r.v1; rp[Np] = __uint_as_float(r.c1); rp[2 * Np] = r.v2;
            rp[3 * Np] = __uint_as_float(r.c2); rp[4 * Np] = r.v3;
        }
    }

    if constexpr (STREAMK) {
        if (sl == 0) {        // this workgroup opens the block: mark the record slots no piece will write
            int64_t gl = blockIdx.x;                     // last workgroup of the block: largest g with start(g) <= last unit
            while (gl + 1 < (int64_t)gridDim.x && ((gl + 1) * U) / gridDim.x <= tb * nstages + nstages - 1) ++gl;
            const int pieces = (int)(gl - blockIdx.x) + 1;
#pragma unroll
            for (int t = 0; t < TT; ++t) {
                const int64_t tokn = (tb * tpb + wave * TT + t) * 16 + (lane & 15);
                if (lane < 16 && tokn < N)
                    for (int p = pieces; p < nslices; ++p) {
                        float *rp = rec + (int64_t)p * VQ_REC_FIELDS * Np + tokn;
                        rp[0] = -INFINITY; rp[Np] = __uint_as_float(0xFFFFFFFFu); rp[2 * Np] = -INFINITY;
                        rp[3 * Np] = __uint_as_float(0xFFFFFFFFu); rp[4 * Np] = -INFINITY;
                    }
            }
        }
        __syncthreads();      // the next segment refills the stage ring
        continue;
    }

    // ---- decision stage, by the workgroup that completes a token block (dec.idx == nullptr: left to refine_decide_kernel)
    // Arrival counter per token block (MI355X guide, Guideline 16): every wave drains its record stores, the workgroup
    // meets, one lane releases at agent scope and takes a ticket; the workgroup that draws the last ticket of the block
    // (nslices of them) acquires and merges the records of all slices — one launch less on the critical path.
    if (dec.idx != nullptr) {
        int *flags = (int *)lds;                         // the stage ring is free now (first barrier below)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) {
            int last = 1;
            if (nslices > 1) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                last = (atomicAdd(&dec.arrive[tb], 1) == nslices - 1) ? 1 : 0;
                if (last) {
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
            }
            flags[0] = last;
        }
        __syncthreads();
        const bool last = flags[0] != 0;
        __syncthreads();                                 // everybody has read the flag before the LDS words are reused
        if (last) {
            int *wcount = (int *)lds, *wbase = wcount + 3 * 16;
            int64_t n = tb * (int64_t)(tpb * 16) + threadIdx.x;      // tpb*16 <= BM <= WAVES*64 threads: one token per thread
            const bool oob = (int)threadIdx.x >= tpb * 16 || n >= N;
            if (n >= N) n = N - 1;
            decide_rows<true>(n, oob, cbst, Dp, metric, nslices, rec, xh2, rho2, Np, dec, wcount, wbase);
        }
    }
  }   // segments
}

// ------------------------------------------------------------------------------------------------
// exact scalar evaluation (refine)
// ------------------------------------------------------------------------------------------------
template <int DT>
__device__ float sqnorm_thread(const void *x, int64_t off, int D) {   // oracle order, one thread
    float p[64];
#pragma unroll
    for (int j = 0; j < 64; ++j) p[j] = 0.0f;
    for (int base = 0; base < D; base += 64) {
#pragma unroll
        for (int j = 0; j < 64; ++j)
            if (base + j < D) { float a = load_elem<DT>(x, off + base + j); p[j] = fmaf(a, a, p[j]); }
    }
#pragma unroll
    for (int off2 = 32; off2 >= 1; off2 >>= 1)
#pragma unroll
        for (int j = 0; j < 32; ++j)
            if (j < off2) p[j] = p[j] + p[j + off2];
    return p[0];
}

template <int DT>
__device__ float oracle_distance(const void *x, int64_t xoff, const float *erow, int D, int metric, float xn, float en) {
    float c = 0.0f;
    if (VQ_IS_L2(metric)) {
        for (int d = 0; d < D; ++d) c = fmaf(-2.0f * load_elem<DT>(x, xoff + d), erow[d], c);
        float t = (c + xn) + en;
        t = (t < 0.0f) ? 0.0f : t;
        return sqrtf(t);
    }
    for (int d = 0; d < D; ++d) c = fmaf(load_elem<DT>(x, xoff + d), erow[d], c);
    return cos_distance(c, metric);
}

// torch.argmin order on (distance, index): NaN first, then smaller distance, then smaller index
__device__ __forceinline__ u64 dist_key(float d, uint32_t k) {
    if (isnan(d)) return (u64)k;
    if (d == 0.0f) d = 0.0f;                 // -0 and +0 tie (lowest index wins), as in torch.argmin
    uint32_t b = __float_as_uint(d);
    // distances are >= 0 for L2; COS distances may be slightly negative: make the map monotone for both signs
    b = (b & 0x80000000u) ? ~b : (b | 0x80000000u);
    return ((u64)b + 1ull) << 32 | (u64)k;      // b+1 <= 2^32 : fits in the upper 33 bits
}

// Rigorous per-row margin (in scaled score units) between the proposal score and the fp32 definition.
// Returns a negative value when the bound cannot be formed (non-finite data): the row is then flagged.
// bf16_part (optional): receives the share of the returned margin that is the worst-case width of a bf16 tie bucket
// (VQ_METRIC_BF16), so that the decision stage, which knows the row's best score, can put the actual width in its place.
__device__ __forceinline__ float row_margin(const VqCbStats *st, int Dp, int metric, float X2, float R2, float *bf16_part) {
    if (bf16_part) *bf16_part = 0.0f;
    if (st->nonfinite != 0 || !isfinite(X2) || !isfinite(R2)) return -1.0f;
    const float infl = 1.0f + 1e-5f;
    float se = cb_scale(st);
    float Xh = sqrtf(X2) * infl, rho = sqrtf(R2) * infl, Xn = Xh + rho;
    float Emax = sqrtf(__uint_as_float(st->e2max_bits)) * infl;
    float Rmax = sqrtf(__uint_as_float(st->r2max_bits)) * infl;
    float Ehmax = sqrtf(__uint_as_float(st->eh2max_bits)) * infl;
    float ENmax = __uint_as_float(st->enmax_bits);
    float Df = (float)Dp;
    float m;
    if (VQ_IS_L2(metric)) {
        // S: rounding slop of the fp32 definition itself (squared-distance units): the D-term fma chain, the two
        // additions and the sqrt tie window.  B: |proposal score - real score| <= fp16 residuals (Cauchy-Schwarz)
        // + fp32 MFMA accumulation + the 4 low mantissa bits that carry the register index.
        float mag = Xn * Xn + ENmax + 2.0f * Xn * Emax;
        if (!(mag < 1e30f)) return -1.0f;
        float S = 2.0f * (1.01f * Df * VQ_U * 2.0f * Xn * Emax + 2.1f * VQ_U * mag) + 4.0f * VQ_U * mag;
        float B = rho * Emax + Xh * Rmax + (4.0f * Df + 32.0f) * VQ_U * (Xh * Ehmax + 0.5f * ENmax);
        m = 2.0f * B + 0.5f * S;
    } else {
        float B = rho * Emax + Xh * Rmax + (4.0f * Df + 32.0f) * VQ_U * (Xh * Ehmax);
        m = 2.0f * B + 2.0f * (Df + 4.0f) * VQ_U * Xn * Emax + 8.0f * VQ_U;
        // bf16-autocast semantics: every similarity s that rounds to the best one's bf16 distance ties with it (lowest index
        // wins), and s_best - s <= ulp_bf16(s) + ulp_bf16(1 - s) <= 2^-7 (|s| + |1 - s|) <= 3 * 2^-7 for |s| <= 1 (+ rounding slop)
        if (VQ_IS_BF16(metric)) {
            const float wworst = 3.0f * 0.0078125f * 1.01f * fmaxf(1.0f, Xn * Emax);
            m += wworst;
            if (bf16_part) *bf16_part = wworst * se * infl;
        }
    }
    m = m * se * infl + 1e-37f;
    if (!isfinite(m)) { if (bf16_part) *bf16_part = 0.0f; return -1.0f; }
    return m;
}

#define VQ_RESCAN_CAP 32     // candidate slots per rescanned row
#define VQ_RESCAN_LOCAL 8    // ... of which one (row block, slice) item of the second pass may contribute

// thread per token: merge the slice records under the margin.  Outcomes:
//   one candidate                         -> idx written here
//   several identified candidates         -> multi_list   (exact re-rank of those candidates)
//   an unidentified candidate may exist   -> rescan_list  (second proposal pass that emits every score >= thr)
//   no usable bound (non-finite data)     -> exact_list   (whole-codebook fp32 pass)
// counters: [0] rescan rows, [1] multi rows, [2] exact rows
// AGENT: `rec` is read with agent-scope loads (they bypass this CU's L1) — inside the proposal kernel the records of the
// other slices were written by other workgroups moments ago; the stand-alone kernel reads them with plain loads.
// NSL > 0: compile-time slice count, every record load is issued before the first one is used (the stage is a latency
// chain: at 16 slices the run-time loop took 17 us instead of 6 at N = 3072).  wcount / wbase: 3 x 16 ints of LDS each.
template <bool AGENT>
__device__ __forceinline__ float rec_load(const float *p) {
    if constexpr (AGENT) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else return *p;
}

// VQ_METRIC_BF16: the margin carries the WORST-CASE width of a bf16 tie bucket (3 * 2^-7: row_margin).  Knowing the row's best
// score s (similarity = score / scale, up to the proposal error the rest of the margin covers), every s' that ties with it
// satisfies s - s' <= ulp(bf16(s)) + ulp(bf16(1 - s)) <= 2^-7 (|s| + |1 - s|) (1 + 2^-6): three times narrower for s in [0, 1].
__device__ __forceinline__ float bf16_tight_margin(float m, float bf16_worst, float gbest, const VqCbStats *st) {
    if (!(bf16_worst > 0.0f) || !(m > 0.0f) || !isfinite(gbest)) return m;
    const float se = cb_scale(st);
    const float s = gbest / se, err = (m - bf16_worst) / se;            // similarity as proposed, and how far off it can be
    // |s| + |1 - s| = 1 for s in [0, 1]; outside, twice the excursion more (err: the proposal's own uncertainty)
    const float spread = 1.0f + 2.0f * (fmaxf(err, 0.0f) + fmaxf(-s, 0.0f) + fmaxf(s - 1.0f, 0.0f));
    const float width = 0.0078125f * 1.02f * spread * se * 1.0001f;
    return width < bf16_worst ? m - bf16_worst + width : m;
}

template <int NSL, bool AGENT>
__device__ __forceinline__ void decide_rows_impl(int64_t n, bool oob, const VqCbStats *st, int Dp, int metric, int nslices,
                                                 const float *rec, const float *xh2, const float *rho2, int64_t Np,
                                                 const VqDecideOut &o, int *wcount, int *wbase) {
    const VqCbStats stv = cb_stats_view(st);
    float bf16_worst = 0.0f;
    float m = row_margin(&stv, Dp, metric, xh2[n], rho2[n], &bf16_worst);
    bool invalid = !(m > 0.0f);
    float gbest = -INFINITY;
    int nc = 0;
    bool unidentified = false;
    uint32_t best = 0xFFFFFFFFu;
    float thr;
    if constexpr (NSL > 0) {
        float v1[NSL], v2[NSL], v3[NSL], c1[NSL];
#pragma unroll
        for (int s = 0; s < NSL; ++s) {
            const float *rp = rec + (int64_t)s * VQ_REC_FIELDS * Np + n;
            v1[s] = rec_load<AGENT>(rp); c1[s] = rec_load<AGENT>(rp + Np);
            v2[s] = rec_load<AGENT>(rp + 2 * Np); v3[s] = rec_load<AGENT>(rp + 4 * Np);
        }
#pragma unroll
        for (int s = 0; s < NSL; ++s) gbest = fmaxf(gbest, v1[s]);
        if (!(gbest > -INFINITY) || !isfinite(gbest)) invalid = true;
        m = bf16_tight_margin(m, bf16_worst, gbest, &stv);
        thr = gbest - m;           // m > 0, so thr <= gbest and the best record always qualifies
#pragma unroll
        for (int s = 0; s < NSL; ++s) {
            if (v3[s] >= thr) unidentified = true;
            if (v1[s] >= thr) { ++nc; best = __float_as_uint(c1[s]); }
            if (v2[s] >= thr) ++nc;
        }
    } else {
        for (int s = 0; s < nslices; ++s) gbest = fmaxf(gbest, rec_load<AGENT>(rec + (int64_t)s * VQ_REC_FIELDS * Np + n));
        if (!(gbest > -INFINITY) || !isfinite(gbest)) invalid = true;
        m = bf16_tight_margin(m, bf16_worst, gbest, &stv);
        thr = gbest - m;
        for (int s = 0; s < nslices; ++s) {
            const float *rp = rec + (int64_t)s * VQ_REC_FIELDS * Np + n;
            const float v1 = rec_load<AGENT>(rp), v2 = rec_load<AGENT>(rp + 2 * Np), v3 = rec_load<AGENT>(rp + 4 * Np);
            if (v3 >= thr) unidentified = true;
            if (v1 >= thr) { ++nc; best = __float_as_uint(rec_load<AGENT>(rp + Np)); }
            if (v2 >= thr) ++nc;
        }
    }
    // block-aggregated list appends: one atomic per workgroup and list (the three counters are hot words:
    // ~8-11 ns per same-address atomic, so per-wave appends from 1024 waves cost ~15 us)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool to_exact = !oob && (invalid || nc == 0);
    const bool to_rescan = !oob && !to_exact && unidentified;
    const bool to_multi = !oob && !to_exact && !to_rescan && nc > 1;
    const u64 mk_e = __ballot(to_exact), mk_r = __ballot(to_rescan), mk_m = __ballot(to_multi);
    if (lane == 0) { wcount[0 * 16 + wave] = __popcll(mk_r); wcount[1 * 16 + wave] = __popcll(mk_m); wcount[2 * 16 + wave] = __popcll(mk_e); }
    __syncthreads();
    if (threadIdx.x < 3) {
        int tot = 0;
        const int nw = blockDim.x >> 6;
        for (int i = 0; i < nw; ++i) { wbase[threadIdx.x * 16 + i] = tot; tot += wcount[threadIdx.x * 16 + i]; }
        const int base = tot ? atomicAdd(&o.counters[threadIdx.x], tot) : 0;
        for (int i = 0; i < nw; ++i) wbase[threadIdx.x * 16 + i] += base;
    }
    __syncthreads();
    const u64 below = (1ull << lane) - 1ull;
    if (to_exact) { o.exact_list[wbase[2 * 16 + wave] + __popcll(mk_e & below)] = (int)n; o.keys[n] = ~0ull; }
    if (to_rescan) { int pos = wbase[0 * 16 + wave] + __popcll(mk_r & below); o.rescan_list[pos] = (int)n; o.rescan_cnt[pos] = 0; o.thr_out[n] = thr; }
    if (to_multi) o.multi_list[wbase[1 * 16 + wave] + __popcll(mk_m & below)] = (int)n;
    if (oob) return;
    if (!to_exact && !to_rescan && !to_multi) {
        o.idx[n] = (int64_t)best;
        if (o.hist) atomicAdd(&o.hist[best], 1);
    }
}

template <bool AGENT>
__device__ __forceinline__ void decide_rows(int64_t n, bool oob, const VqCbStats *st, int Dp, int metric, int nslices,
                                            const float *rec, const float *xh2, const float *rho2, int64_t Np,
                                            const VqDecideOut &o, int *wcount, int *wbase) {
    switch (nslices) {      // every thread of the workgroup takes the same case (the list appends contain barriers)
#define VQ_DECIDE_CASE(NSL) case NSL: decide_rows_impl<NSL, AGENT>(n, oob, st, Dp, metric, nslices, rec, xh2, rho2, Np, o, wcount, wbase); break;
        VQ_DECIDE_CASE(1) VQ_DECIDE_CASE(2) VQ_DECIDE_CASE(4) VQ_DECIDE_CASE(8) VQ_DECIDE_CASE(16)
#undef VQ_DECIDE_CASE
        default: decide_rows_impl<0, AGENT>(n, oob, st, Dp, metric, nslices, rec, xh2, rho2, Np, o, wcount, wbase); break;
    }
}

// stand-alone form (one thread per token, 1024-thread workgroups): used when the proposal kernel does not decide itself
__global__ void refine_decide_kernel(const char *cb, VqCbLayout L, int64_t N, int metric, int nslices, const float *rec,
                                     const float *xh2, const float *rho2, int64_t Np, VqDecideOut o) {
    __shared__ int wcount[3 * 16];
    __shared__ int wbase[3 * 16];
    if (o.n_dev != nullptr) {            // device-side row count (uniform: taken before any barrier)
        const int64_t nd = *o.n_dev;
        N = nd < N ? nd : N;
        if ((int64_t)blockIdx.x * blockDim.x >= N) return;
    }
    int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool oob = n >= N;
    if (oob) n = N - 1;                  // out-of-range threads compute on a valid row and take part in the barriers
    decide_rows<false>(n, oob, (const VqCbStats *)(cb + L.off_stats), L.Dp, metric, nslices, rec, xh2, rho2, Np, o, wcount, wbase);
}

// Second proposal pass over the rows of rescan_list only: same fp16 MFMA scores as coarse_kernel (bitwise: same
// operands, same instruction sequence per accumulator), but every score >= the row's threshold is appended to the row's
// candidate list.  Same machinery as coarse_kernel — the rows' fragments (from the packed image) stay in registers,
// codebook stages arrive by LDS-DMA through the same ring and are shared by the 8 waves — as a persistent grid over
// (block of WAVES*TT*16 queued rows, slice of stages) items, the slice count chosen on the device from the queue
// length so that every workgroup gets an item.
template <int NSTEP, int TT, int WAVES, int TPS, int NBUF = 2>
__global__ __launch_bounds__(WAVES * 64) void rescan_kernel(const char *__restrict__ ximg, const char *__restrict__ frag,
                                                            int64_t nstages, const int *__restrict__ rescan_list,
                                                            const int *__restrict__ counters, const float *__restrict__ thr,
                                                            int *__restrict__ rescan_cnt, int *__restrict__ cand_list) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    constexpr int NS32 = NSTEP / 2;
    constexpr int NCH = TPS * NSTEP + VQ_AUX_CHUNKS(TPS);
    constexpr int STAGE_BYTES = NCH * VQ_CHUNK_BYTES;
    constexpr int BM = WAVES * TT * 16;
    constexpr int PF = NSTEP <= 32 ? 1 : (NSTEP <= 48 ? 2 : 4);
    // hits are collected per row in LDS (LDS atomics) and appended to the global lists once per item, one global
    // atomic per (row, item): a returning global atomic inside the MFMA loop stalls its wave for a memory round trip
    int *lcnt = (int *)(lds + NBUF * STAGE_BYTES);                    // [BM]
    uint32_t *lcand = (uint32_t *)(lds + NBUF * STAGE_BYTES) + BM;     // [BM][VQ_RESCAN_LOCAL]
    constexpr int AHEAD = NBUF >= 4 ? 2 : 1;                          // ring of four stages, filled two ahead
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nrows = counters[0];
    if (nrows <= 0) return;
    const int64_t ntb = (nrows + BM - 1) / BM;
    int64_t ns = 1;
    while (ntb * ns < (int64_t)gridDim.x && ns * 2 <= nstages) ns <<= 1;

    auto issue_stage = [&](int64_t st, int buf) {
        const char *src = frag + st * (int64_t)STAGE_BYTES;
        char *dstb = lds + buf * STAGE_BYTES;
        for (int c = wave; c < NCH; c += WAVES)
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void *)(src + c * VQ_CHUNK_BYTES + lane * 16),
                (__attribute__((address_space(3))) void *)(dstb + c * VQ_CHUNK_BYTES), 16, 0, 0);
    };

    for (int64_t item = blockIdx.x; item < ntb * ns; item += gridDim.x) {
        const int64_t sl = item % ns, tb = item / ns;
        const int64_t st0 = (nstages * sl) / ns, st1 = (nstages * (sl + 1)) / ns;
        issue_stage(st0, 0);
        if (AHEAD >= 2 && st0 + 1 < st1) issue_stage(st0 + 1, 1);
        for (int i = threadIdx.x; i < BM; i += WAVES * 64) lcnt[i] = 0;
        half8 xf[TT][NS32];
        float mythr[TT];
        int slot[TT];
#pragma unroll
        for (int t = 0; t < TT; ++t) {
            int64_t tt = tb * (BM / 16) + wave * TT + t;
            slot[t] = (int)(tt * 16 + (lane & 15));
            const bool valid = slot[t] < nrows;
            // this lane's queued row (padding slots repeat the last queued row and never emit) and its B fragments,
            // gathered straight from the token image: 16-byte piece (lane>>4, row&15) of chunk (row>>4, s).  All
            // TT*NS32 loads of a lane are independent and in flight together, once per item.
            const int64_t tk = rescan_list[valid ? slot[t] : nrows - 1];
            mythr[t] = valid ? thr[tk] : INFINITY;
            const char *src = ximg + (tk >> 4) * (int64_t)(NS32 * VQ_CHUNK_BYTES) + ((lane >> 4) * 16 + (int)(tk & 15)) * 16;
#pragma unroll
            for (int s = 0; s < NS32; ++s) xf[t][s] = *(const half8 *)(src + s * VQ_CHUNK_BYTES);
        }
        __syncthreads();   // stage st0 landed
        for (int64_t st = st0; st < st1; ++st) {
            const int buf = (int)((st - st0) % NBUF);
            if (st + AHEAD < st1) issue_stage(st + AHEAD, (int)((st + AHEAD - st0) % NBUF));
            const char *base = lds + buf * STAGE_BYTES;
            const char *aux = base + TPS * NSTEP * VQ_CHUNK_BYTES;
#pragma unroll
            for (int ti = 0; ti < TPS; ++ti) {
                f32x4 acc[2][TT];
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    f32x4 a4 = *(const f32x4 *)(aux + (ti * 32 + 16 * c + 4 * (lane >> 4)) * 4);
#pragma unroll
                    for (int t = 0; t < TT; ++t) acc[c][t] = a4;
                }
                half8 af[PF + 1];
#pragma unroll
                for (int i = 0; i < PF; ++i)
                    if (i < NSTEP) af[i] = *(const half8 *)(base + (ti * NSTEP + i) * VQ_CHUNK_BYTES + lane * 16);
#pragma unroll
                for (int ch = 0; ch < NSTEP; ++ch) {
                    if (ch + PF < NSTEP)
                        af[(ch + PF) % (PF + 1)] = *(const half8 *)(base + (ti * NSTEP + ch + PF) * VQ_CHUNK_BYTES + lane * 16);
#pragma unroll
                    for (int t = 0; t < TT; ++t)
                        acc[ch & 1][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[ch % (PF + 1)], xf[t][ch >> 1], acc[ch & 1][t], 0, 0, 0);
                }
                uint32_t hits = 0;      // bit 8t + e
#pragma unroll
                for (int t = 0; t < TT; ++t) {
                    // the lane's 8 scores of this (token tile, code tile): one maximum and one compare first — a queued row
                    // has a handful of scores above its threshold in the whole codebook, so nearly every tile ends here
                    // (med3(a, b, +inf) = max(a, b), visible to the compiler: MFMA-result hazards are its to pad)
                    float m = __builtin_amdgcn_fmed3f(acc[0][t][0], acc[0][t][1], INFINITY);
                    m = __builtin_amdgcn_fmed3f(m, acc[0][t][2], INFINITY); m = __builtin_amdgcn_fmed3f(m, acc[0][t][3], INFINITY);
                    m = __builtin_amdgcn_fmed3f(m, acc[1][t][0], INFINITY); m = __builtin_amdgcn_fmed3f(m, acc[1][t][1], INFINITY);
                    m = __builtin_amdgcn_fmed3f(m, acc[1][t][2], INFINITY); m = __builtin_amdgcn_fmed3f(m, acc[1][t][3], INFINITY);
                    if (__any(m >= mythr[t] || m != m)) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) hits |= (acc[e >> 2][t][e & 3] >= mythr[t]) ? (1u << (8 * t + e)) : 0u;
                    }
                }
                while (hits) {
                    const int b = __ffs((int)hits) - 1;
                    hits &= hits - 1;
                    const int t = b >> 3, e = b & 7;
                    const uint32_t code = (uint32_t)((st * TPS + ti) * 32 + tile_row16(e, lane));
                    const int row = (wave * TT + t) * 16 + (lane & 15);
                    const int pos = atomicAdd(&lcnt[row], 1);
                    if (pos < VQ_RESCAN_LOCAL) lcand[row * VQ_RESCAN_LOCAL + pos] = code;
                }
            }
            __syncthreads();   // next stage landed and everybody is done reading this one
        }
        // flush: thread r owns local row r
        for (int r = threadIdx.x; r < BM; r += WAVES * 64) {
            const int c = lcnt[r];
            const int64_t gs = tb * BM + r;
            if (c > 0 && gs < nrows) {
                // a local list that overflowed lost candidates: push the row's count past the cap (fp32 pass)
                const int base = atomicAdd(&rescan_cnt[gs], c > VQ_RESCAN_LOCAL ? VQ_RESCAN_CAP + 1 : c);
                const int m = c > VQ_RESCAN_LOCAL ? VQ_RESCAN_LOCAL : c;
                for (int i = 0; i < m; ++i)
                    if (base + i < VQ_RESCAN_CAP) cand_list[gs * VQ_RESCAN_CAP + base + i] = (int)lcand[r * VQ_RESCAN_LOCAL + i];
            }
        }
        __syncthreads();       // lists are re-zeroed by the next item
    }
}

// Exact fp32 evaluation of the candidates of the queued rows.  A wave owns P = max(16, S) (row, candidate slot) pairs:
// the S slots of a row sit in S neighbouring lanes (S a power of two, 4..32), lane p < P runs the oracle's fma chain
// (d order) of pair p, and all 64 lanes move the operands: per 32-dim segment the wave fetches the 128-byte piece of
// every pair's code row (8 lanes x 16 bytes per piece, whole cache lines) and of its latent rows one segment ahead
// into registers, parks them in a small wave-private padded LDS tile, and the chain lanes read their rows from it.
// Few pairs per wave means many waves: the operand latency is hidden by occupancy rather than by deep per-lane
// prefetch.  The S lanes of a row finally agree on the smallest (distance, code) key.
// SRC 0: rows of multi_list, slots = the 2*nslices (value, code) fields of the proposal records within the margin.
// SRC 1: rows of rescan_list, slots = the first VQ_RESCAN_CAP emitted candidates; longer lists go to the fp32 pass.
#define VQ_RR_STRIDE 36      // floats per LDS tile row: 32 dims + 4 pad (conflict-free b128 reads of 16 rows)
template <int DT, int SRC>
__device__ __forceinline__ void rerank_rows(float *te, float *tx, int64_t gwave, int64_t nwaves,
                                            const void *__restrict__ x, const float *__restrict__ e_exact,
                                            const char *__restrict__ cb, const VqCbLayout &L, int D, int metric,
                                            int nslices, int S, const float *__restrict__ rec,
                                            const float *__restrict__ xh2, const float *__restrict__ rho2,
                                            const float *__restrict__ xnorm, int64_t Np,
                                            int64_t *__restrict__ idx, int32_t *__restrict__ hist,
                                            const int *__restrict__ row_list, int *__restrict__ counters,
                                            const int *__restrict__ rescan_cnt,
                                            const int *__restrict__ cand_list, int *__restrict__ exact_list,
                                            u64 *__restrict__ keys) {
    constexpr int XL = DT == 0 ? 8 : 4;                       // lanes per 32-dim latent row piece (16 bytes each)
    const int lane = threadIdx.x & 63;
    const VqCbStats stv = cb_stats_view((const VqCbStats *)(cb + L.off_stats));
    const VqCbStats *st = &stv;
    const float *en = (const float *)(cb + L.off_en);
    const int nrows = counters[SRC == 0 ? 1 : 0];
    const int P = S < 16 ? 16 : S;                            // pairs per wave (16 or 32)
    const int ne = P >> 3;                                    // load instructions per code segment
    const int rpw = P / S;                                    // rows per wave (<= 4)
    const bool chain = lane < P;
    const int j = lane & (S - 1);                             // this lane's slot
    const float sx = (VQ_IS_L2(metric)) ? -2.0f : 1.0f;
    const int nseg = (D + 31) >> 5;
    const float *myx = tx + (lane / S) * VQ_RR_STRIDE, *mye = te + (lane & 31) * VQ_RR_STRIDE;
    for (int64_t base = gwave * rpw; base < nrows; base += nwaves * rpw) {
        const int64_t item = base + lane / S;
        const bool rvalid = chain && item < nrows;
        const int64_t n = rvalid ? row_list[item] : 0;
        bool cand = false;
        uint32_t code = 0;
        if (SRC == 0) {
            float v = -INFINITY;
            uint32_t cd = 0xFFFFFFFFu;
            if (rvalid && j < 2 * nslices) {
                const float *rp = rec + (int64_t)(j >> 1) * VQ_REC_FIELDS * Np + n;
                v = rp[(2 * (j & 1)) * Np];
                cd = __float_as_uint(rp[(2 * (j & 1) + 1) * Np]);
            }
            float gbest = (j & 1) ? -INFINITY : v;            // best first-field value over the row's slices
            for (int off = 1; off < S; off <<= 1) gbest = fmaxf(gbest, __shfl_xor(gbest, off, 64));
            const float m = rvalid ? row_margin(st, L.Dp, metric, xh2[n], rho2[n]) : 0.0f;
            cand = rvalid && (j < 2 * nslices) && (v >= gbest - m) && cd != 0xFFFFFFFFu;
            if (cand) code = cd;
        } else {
            const int cnt = rvalid ? rescan_cnt[item] : 0;
            if (rvalid && (cnt > VQ_RESCAN_CAP || cnt <= 0)) {
                if (j == 0) {
                    int pos = atomicAdd(&counters[2], 1);
                    exact_list[pos] = (int)n;
                    keys[n] = ~0ull;
                }
            } else if (rvalid && j < cnt) {
                cand = true;
                code = (uint32_t)cand_list[item * VQ_RESCAN_CAP + j];
            }
        }
        // who loads what (every lane takes part in the shuffles): instruction i of a code segment covers pairs
        // 8i + (lane>>3), 16-byte piece lane&7; the latent segment is one instruction: wave row lane/XL, piece lane%XL
        const float *ep0, *ep1, *ep2, *ep3;
        bool ok0, ok1, ok2, ok3;
        {
            const int q = lane >> 3, pc = 4 * (lane & 7);
            const uint32_t c0 = __shfl(code, q, 64), c1 = __shfl(code, 8 + q, 64), c2 = __shfl(code, 16 + q, 64),
                           c3 = __shfl(code, 24 + q, 64);
            const int k0 = __shfl((int)cand, q, 64), k1 = __shfl((int)cand, 8 + q, 64), k2 = __shfl((int)cand, 16 + q, 64),
                      k3 = __shfl((int)cand, 24 + q, 64);
            ep0 = e_exact + (int64_t)c0 * D + pc; ep1 = e_exact + (int64_t)c1 * D + pc;
            ep2 = e_exact + (int64_t)c2 * D + pc; ep3 = e_exact + (int64_t)c3 * D + pc;
            ok0 = k0 != 0; ok1 = k1 != 0 && ne > 1; ok2 = k2 != 0 && ne > 2; ok3 = k3 != 0 && ne > 2;
        }
        const int xr_row = lane / XL;
        const int xsrc = xr_row * S;
        const int64_t xn_row = __shfl(n, xsrc < 64 ? xsrc : 0, 64);
        const int xrv = __shfl((int)rvalid, xsrc < 64 ? xsrc : 0, 64);
        const bool xok = xr_row < rpw && xrv != 0;
        const int64_t xoff = xn_row * D + (DT == 0 ? 4 : 8) * (lane % XL);
        const int epiece = 4 * (lane & 7), xpiece = (DT == 0 ? 4 : 8) * (lane % XL);

        // register ring PD segments deep: at step g the wave fetches segment g+PD, runs the chains over segment g (in
        // the tile) and parks segment g+1; LDS operations of one wave execute in order, so one tile is enough
        constexpr int PD = 2;
        float4 r[PD][4];
        float4 rxf[PD];                                        // latent piece: fp32 (DT 0) or 8 bf16 (DT 1)
        uint4 rxb[PD];
#pragma unroll
        for (int u = 0; u < PD; ++u) {
            rxf[u] = make_float4(0, 0, 0, 0); rxb[u] = make_uint4(0, 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 4; ++i) r[u][i] = make_float4(0, 0, 0, 0);
        }
        float c = 0.0f;
        for (int g0 = -PD; g0 < nseg; g0 += PD) {
#pragma unroll
            for (int u = 0; u < PD; ++u) {
                const int g = g0 + u;
                if (g + PD < nseg) {                           // fetch segment g+PD into ring slot u
                    const int d0 = 32 * (g + PD);
                    const bool ein = d0 + epiece < D;
                    if (ok0 && ein) r[u][0] = *(const float4 *)(ep0 + d0);
                    if (ok1 && ein) r[u][1] = *(const float4 *)(ep1 + d0);
                    if (ok2 && ein) r[u][2] = *(const float4 *)(ep2 + d0);
                    if (ok3 && ein) r[u][3] = *(const float4 *)(ep3 + d0);
                    if (xok && d0 + xpiece < D) {
                        if constexpr (DT == 0) rxf[u] = *(const float4 *)((const float *)x + xoff + d0);
                        else rxb[u] = *(const uint4 *)((const uint16_t *)x + xoff + d0);
                    }
                }
                if (g >= 0 && g < nseg && chain) {             // chains over segment g from the tile
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        if (32 * g + 4 * k < D) {
                            const float4 a = *(const float4 *)(myx + 4 * k), bq = *(const float4 *)(mye + 4 * k);
                            c = fmaf(sx * a.x, bq.x, c); c = fmaf(sx * a.y, bq.y, c);
                            c = fmaf(sx * a.z, bq.z, c); c = fmaf(sx * a.w, bq.w, c);
                        }
                    }
                }
                __builtin_amdgcn_wave_barrier();
                if (g + 1 >= 0 && g + 1 < nseg) {              // park segment g+1 (ring slot (u+1) % PD)
                    constexpr int PDm = PD;
                    const int v = (u + 1) % PDm;
                    float *dst = te + (lane >> 3) * VQ_RR_STRIDE + epiece;
                    *(float4 *)dst = r[v][0];
                    if (ne > 1) *(float4 *)(dst + 8 * VQ_RR_STRIDE) = r[v][1];
                    if (ne > 2) { *(float4 *)(dst + 16 * VQ_RR_STRIDE) = r[v][2]; *(float4 *)(dst + 24 * VQ_RR_STRIDE) = r[v][3]; }
                    if (xr_row < 8) {
                        if constexpr (DT == 0) {
                            *(float4 *)(tx + xr_row * VQ_RR_STRIDE + xpiece) = rxf[v];
                        } else {
                            float xv[8];
                            RawVec<1>::unpack(rxb[v], xv);
                            float *dx = tx + xr_row * VQ_RR_STRIDE + xpiece;
                            *(float4 *)dx = make_float4(xv[0], xv[1], xv[2], xv[3]);
                            *(float4 *)(dx + 4) = make_float4(xv[4], xv[5], xv[6], xv[7]);
                        }
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
        }
        u64 key = ~0ull;
        if (cand) {
            float dist;
            if (VQ_IS_L2(metric)) {
                const float xn = xnorm[n];
                float t = VQ_SWAPPED(metric) ? (c + en[code]) + xn : (c + xn) + en[code];
                t = (t < 0.0f) ? 0.0f : t;
                dist = sqrtf(t);
            } else {
                dist = cos_distance(c, metric);
            }
            key = dist_key(dist, code);
        }
        for (int off = 1; off < S; off <<= 1) { u64 o = __shfl_xor(key, off, 64); key = o < key ? o : key; }
        if (chain && j == 0 && key != ~0ull) {
            const uint32_t best = (uint32_t)(key & 0xFFFFFFFFull);
            idx[n] = (int64_t)best;
            if (hist) atomicAdd(&hist[best], 1);
        }
    }
}

// One launch re-ranks both queues: blocks [0, g0) take the rows with several identified candidates (multi_list, S0
// slot lanes per row), the other blocks the rescanned rows (rescan_list, VQ_RESCAN_CAP slots).  g0 == gridDim.x or
// g0 == 0 runs one queue only.
template <int DT>
__global__ __launch_bounds__(256) void refine_rerank_kernel(const void *__restrict__ x, const float *__restrict__ e_exact,
                                                            const char *__restrict__ cb, VqCbLayout L, int D, int metric,
                                                            int nslices, int S0, int g0, const float *__restrict__ rec,
                                                            const float *__restrict__ xh2, const float *__restrict__ rho2,
                                                            const float *__restrict__ xnorm, int64_t Np,
                                                            int64_t *__restrict__ idx, int32_t *__restrict__ hist,
                                                            const int *__restrict__ multi_list,
                                                            const int *__restrict__ rescan_list, int *__restrict__ counters,
                                                            const int *__restrict__ rescan_cnt,
                                                            const int *__restrict__ cand_list, int *__restrict__ exact_list,
                                                            u64 *__restrict__ keys) {
    __shared__ __attribute__((aligned(16))) float tile_e[4][32 * VQ_RR_STRIDE];
    __shared__ __attribute__((aligned(16))) float tile_x[4][8 * VQ_RR_STRIDE];
    const int wave = threadIdx.x >> 6;
    if ((int)blockIdx.x < g0)
        rerank_rows<DT, 0>(tile_e[wave], tile_x[wave], (int64_t)blockIdx.x * 4 + wave, (int64_t)g0 * 4, x, e_exact, cb, L, D,
                           metric, nslices, S0, rec, xh2, rho2, xnorm, Np, idx, hist, multi_list, counters, nullptr, nullptr,
                           nullptr, nullptr);
    else
        rerank_rows<DT, 1>(tile_e[wave], tile_x[wave], (int64_t)(blockIdx.x - g0) * 4 + wave,
                           (int64_t)(gridDim.x - g0) * 4, x, e_exact, cb, L, D, metric, nslices, VQ_RESCAN_CAP, rec, xh2,
                           rho2, xnorm, Np, idx, hist, rescan_list, counters, rescan_cnt, cand_list, exact_list, keys);
}

// ------------------------------------------------------------------------------------------------
// exact fp32 MFMA pass (v_mfma_f32_32x32x2_f32 == k-ordered fmaf chain)
// ------------------------------------------------------------------------------------------------
// MODE 0: row argmin via 64-bit atomicMin keys[row]; MODE 1: column argmin keys[code]; MODE 2: store d[N,K]
// Work item = (tile of 32 rows, chunk of 4*CT*32 codes); persistent grid-stride loop over items.
template <int DT, int MODE, int CT>
__global__ __launch_bounds__(256) void exact_kernel(const void *__restrict__ x, const float *__restrict__ e,
                                                    const float *__restrict__ en_in,
                                                    const float *__restrict__ xn_in, int64_t N, int64_t K, int D,
                                                    int metric, const int *__restrict__ row_list,
                                                    const int *__restrict__ nrows_dev, u64 *__restrict__ keys,
                                                    float *__restrict__ dout, int *__restrict__ ticket = nullptr,
                                                    int64_t *__restrict__ fin_idx = nullptr,
                                                    int32_t *__restrict__ fin_hist = nullptr) {
    // CT = code tiles (32 codes) per wave: 4 for whole-batch passes, 1 when only a few flagged rows need the
    // whole codebook (more, smaller work items)
    // ticket != nullptr (row-list form): the workgroup that finishes last decodes keys -> idx (+hist) for the listed rows
    // itself, so the last-resort path is ONE launch; with an empty list every workgroup returns at once.
    constexpr int DB = 256;                     // dims per register block
    constexpr int CHUNK = 4 * CT * 32;          // codes per work item (4 waves)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 31, h = lane >> 5;
    const int64_t nrows = row_list ? (int64_t)(*nrows_dev) : N;
    const int64_t ntiles = (nrows + 31) / 32;
    const int64_t nchunks = (K + CHUNK - 1) / CHUNK;
    const float sx = (VQ_IS_L2(metric)) ? -2.0f : 1.0f;

    for (int64_t item = blockIdx.x; item < ntiles * nchunks; item += gridDim.x) {
        const int64_t tile = item / nchunks, chunk = item % nchunks;
        const int64_t slot = tile * 32 + j;
        const bool rvalid = slot < nrows;
        const int64_t row = rvalid ? (row_list ? (int64_t)row_list[slot] : slot) : 0;
        const int64_t kbase = chunk * CHUNK + (int64_t)wave * CT * 32;

        f32x16 acc[CT];
#pragma unroll
        for (int c = 0; c < CT; ++c)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[c][q] = 0.0f;

        // oracle-order |x|^2: precomputed for whole-batch passes, computed per lane on the (rare) last-resort path
        const float xn = (VQ_IS_L2(metric) && rvalid) ? (xn_in ? xn_in[row] : sqnorm_thread<DT>(x, row * D, D)) : 0.0f;
        for (int db = 0; db < D; db += DB) {
            // B fragments: lane (row j, k-parity h) holds sx * x[row][db + 2s + h], s = 0..DB/2-1
            float xfr[DB / 2];
#pragma unroll
            for (int s4 = 0; s4 < DB / 4; ++s4) {      // 4 consecutive dims per load
                int d = db + 4 * s4;
                float v0 = 0, v1 = 0, v2 = 0, v3 = 0;
                if (rvalid && d < D) {
                    if (DT == 0) {
                        if (d + 3 < D && (D % 4) == 0) {
                            float4 t = *(const float4 *)((const float *)x + row * D + d);
                            v0 = t.x; v1 = t.y; v2 = t.z; v3 = t.w;
                        } else {
                            v0 = load_elem<DT>(x, row * D + d);
                            if (d + 1 < D) v1 = load_elem<DT>(x, row * D + d + 1);
                            if (d + 2 < D) v2 = load_elem<DT>(x, row * D + d + 2);
                            if (d + 3 < D) v3 = load_elem<DT>(x, row * D + d + 3);
                        }
                    } else if (d + 3 < D && (D % 4) == 0) {
                        uint2 t = *(const uint2 *)((const uint16_t *)x + row * D + d);
                        v0 = __uint_as_float(t.x << 16); v1 = __uint_as_float(t.x & 0xFFFF0000u);
                        v2 = __uint_as_float(t.y << 16); v3 = __uint_as_float(t.y & 0xFFFF0000u);
                    } else {
                        v0 = load_elem<DT>(x, row * D + d);
                        if (d + 1 < D) v1 = load_elem<DT>(x, row * D + d + 1);
                        if (d + 2 < D) v2 = load_elem<DT>(x, row * D + d + 2);
                        if (d + 3 < D) v3 = load_elem<DT>(x, row * D + d + 3);
                    }
                }
                xfr[2 * s4] = sx * (h ? v1 : v0);
                xfr[2 * s4 + 1] = sx * (h ? v3 : v2);
            }
#pragma unroll
            for (int c = 0; c < CT; ++c) {
                const int64_t k = kbase + c * 32 + j;         // this lane's A row (code)
                const bool kvalid = k < K;
                const float *erow = e + (kvalid ? k : 0) * D;
#pragma unroll
                for (int s4 = 0; s4 < DB / 4; ++s4) {
                    int d = db + 4 * s4;
                    float a0 = 0, a1 = 0, a2 = 0, a3 = 0;
                    if (d < D) {
                        if (d + 3 < D && (D % 4) == 0) {
                            float4 t = *(const float4 *)(erow + d);
                            a0 = t.x; a1 = t.y; a2 = t.z; a3 = t.w;
                        } else {
                            a0 = erow[d];
                            if (d + 1 < D) a1 = erow[d + 1];
                            if (d + 2 < D) a2 = erow[d + 2];
                            if (d + 3 < D) a3 = erow[d + 3];
                        }
                        if (!kvalid) { a0 = a1 = a2 = a3 = 0.0f; }
                        acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(h ? a1 : a0, xfr[2 * s4], acc[c], 0, 0, 0);
                        acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(h ? a3 : a2, xfr[2 * s4 + 1], acc[c], 0, 0, 0);
                    }
                }
            }
        }

        // epilogue: C[code row][token col j]
        u64 best = ~0ull;
#pragma unroll
        for (int c = 0; c < CT; ++c) {
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int64_t k = kbase + c * 32 + mfma_row(q, h);
                float d;
                if (VQ_IS_L2(metric)) {
                    const float enk = (k < K) ? en_in[k] : 0.0f;
                    float t = VQ_SWAPPED(metric) ? (acc[c][q] + enk) + xn : (acc[c][q] + xn) + enk;
                    t = (t < 0.0f) ? 0.0f : t;
                    d = sqrtf(t);
                } else {
                    d = cos_distance(acc[c][q], metric);
                }
                if (MODE == 0) {
                    if (k < K) { u64 key = dist_key(d, (uint32_t)k); best = key < best ? key : best; }
                } else if (MODE == 1) {
                    // column argmin: reduce over the 32 token lanes of this half, one atomic per code
                    u64 key = (rvalid && k < K) ? dist_key(d, (uint32_t)row) : ~0ull;
#pragma unroll
                    for (int off = 16; off >= 1; off >>= 1) {
                        u64 o = __shfl_xor(key, off, 64);
                        key = o < key ? o : key;
                    }
                    if (j == 0 && k < K && key != ~0ull) atomicMin(&keys[k], key);
                } else {
                    if (rvalid && k < K) dout[row * K + k] = d;
                }
            }
        }
        if (MODE == 0) {
            u64 o = __shfl_xor(best, 32, 64);
            best = o < best ? o : best;
            if (h == 0 && rvalid && best != ~0ull) atomicMin(&keys[row], best);
        }
    }
    if (MODE == 0 && ticket != nullptr && nrows > 0) {
        // arrival counter (MI355X guide, Guideline 16): the key atomics execute at the memory side; every wave drains
        // its own, the workgroup meets, one lane publishes; whoever draws the last ticket reads the keys with loads that
        // bypass its L1 (agent-scope relaxed atomic loads)
        __shared__ int is_last;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            is_last = (atomicAdd(ticket, 1) == (int)gridDim.x - 1) ? 1 : 0;
        }
        __syncthreads();
        if (is_last) {
            for (int64_t i = threadIdx.x; i < nrows; i += blockDim.x) {
                const int64_t r = (int64_t)row_list[i];
                const u64 key = __hip_atomic_load(&keys[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const uint32_t k = (uint32_t)(key & 0xFFFFFFFFull);
                fin_idx[r] = (int64_t)k;
                if (fin_hist) atomicAdd(&fin_hist[k], 1);
            }
        }
    }
}

// Whole-batch fp32 pass (argmin_exact, col_argmin, distance): a workgroup = 4 waves x 32 rows against a chunk of CT code
// tiles.  Code tiles are staged through LDS once per workgroup (coalesced float4 loads, register prefetch of the next
// tile, 16-byte XOR swizzle -> conflict-free ds_read_b128) and shared by the 4 waves; accumulators of all CT tiles stay
// live so that the row fragments are loaded once per 256-dim block.  Same k-ordered fma chains as exact_kernel.
template <int DT, int MODE>
__global__ __launch_bounds__(256) void exact_tiled_kernel(const void *__restrict__ x, const float *__restrict__ e,
                                                          const float *__restrict__ en_in, const float *__restrict__ xn_in,
                                                          int64_t N, int64_t K, int D, int metric, u64 *__restrict__ keys,
                                                          float *__restrict__ dout) {
    constexpr int CT = 8;                        // code tiles (32 codes) per work item
    constexpr int DB = 128;                      // dims per register / LDS block
    constexpr int NPRE = 32 * (DB / 4) / 256;    // 16-byte chunks of a tile per thread
    extern __shared__ __attribute__((aligned(16))) char lds[];
    float4 *tile = (float4 *)lds;                // [2][32 rows][DB/4 chunks], chunk index XOR (row & 15)
    constexpr int CPR = DB / 4;                  // chunks per row
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 31, h = lane >> 5;
    const int64_t nrb = (N + 127) / 128;
    const int64_t nchunks = (K + CT * 32 - 1) / (CT * 32);
    const float sx = (VQ_IS_L2(metric)) ? -2.0f : 1.0f;

    for (int64_t item = blockIdx.x; item < nrb * nchunks; item += gridDim.x) {
        const int64_t rb = item / nchunks, chunk = item % nchunks;
        const int64_t row = rb * 128 + wave * 32 + j;
        const bool rvalid = row < N;
        const int64_t rrow = rvalid ? row : N - 1;
        const int64_t kbase = chunk * CT * 32;
        f32x16 acc[CT];
#pragma unroll
        for (int c = 0; c < CT; ++c)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[c][q] = 0.0f;

        for (int db = 0; db < D; db += DB) {
            // B fragments: lane (row j, k-parity h) holds sx * x[row][db + 2s + h], s = 0..DB/2-1
            float xfr[DB / 2];
#pragma unroll
            for (int s4 = 0; s4 < DB / 4; ++s4) {
                const int d = db + 4 * s4;
                float v0 = 0, v1 = 0, v2 = 0, v3 = 0;
                if (d < D) {
                    if (d + 3 < D && (D % 4) == 0) {
                        if (DT == 0) {
                            float4 t = *(const float4 *)((const float *)x + rrow * D + d);
                            v0 = t.x; v1 = t.y; v2 = t.z; v3 = t.w;
                        } else {
                            uint2 t = *(const uint2 *)((const uint16_t *)x + rrow * D + d);
                            v0 = __uint_as_float(t.x << 16); v1 = __uint_as_float(t.x & 0xFFFF0000u);
                            v2 = __uint_as_float(t.y << 16); v3 = __uint_as_float(t.y & 0xFFFF0000u);
                        }
                    } else {
                        v0 = load_elem<DT>(x, rrow * D + d);
                        if (d + 1 < D) v1 = load_elem<DT>(x, rrow * D + d + 1);
                        if (d + 2 < D) v2 = load_elem<DT>(x, rrow * D + d + 2);
                        if (d + 3 < D) v3 = load_elem<DT>(x, rrow * D + d + 3);
                    }
                }
                xfr[2 * s4] = sx * (h ? v1 : v0);
                xfr[2 * s4 + 1] = sx * (h ? v3 : v2);
            }
            // staging: thread t owns the 16-byte chunks t, t+256, ... of the 32 x DB tile
            float4 pre[NPRE];
            auto fetch = [&](int ct) {
#pragma unroll
                for (int i = 0; i < NPRE; ++i) {
                    const int c = threadIdx.x + 256 * i;
                    const int r = c / CPR, ch = c % CPR;
                    const int64_t k = kbase + ct * 32 + r;
                    const int d = db + 4 * ch;
                    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (k < K && d < D) {
                        if (d + 3 < D && (D % 4) == 0) v = *(const float4 *)(e + k * D + d);
                        else {
                            v.x = e[k * D + d];
                            if (d + 1 < D) v.y = e[k * D + d + 1];
                            if (d + 2 < D) v.z = e[k * D + d + 2];
                            if (d + 3 < D) v.w = e[k * D + d + 3];
                        }
                    }
                    pre[i] = v;
                }
            };
            auto stash = [&](int buf) {
#pragma unroll
                for (int i = 0; i < NPRE; ++i) {
                    const int c = threadIdx.x + 256 * i;
                    const int r = c / CPR, ch = c % CPR;
                    tile[(buf * 32 + r) * CPR + (ch ^ (r & 15))] = pre[i];
                }
            };
            __syncthreads();            // previous block / item is done with both buffers
            fetch(0);
            stash(0);
            __syncthreads();
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                if (ct + 1 < CT) fetch(ct + 1);
                const float4 *trow = tile + ((ct & 1) * 32 + j) * (DB / 4);
#pragma unroll
                for (int q = 0; q < DB / 4; ++q) {
                    if (db + 4 * q < D) {
                        const float4 v = trow[q ^ (j & 15)];
                        acc[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(h ? v.y : v.x, xfr[2 * q], acc[ct], 0, 0, 0);
                        acc[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(h ? v.w : v.z, xfr[2 * q + 1], acc[ct], 0, 0, 0);
                    }
                }
                if (ct + 1 < CT) stash((ct + 1) & 1);
                __syncthreads();
            }
        }

        const float xn = (VQ_IS_L2(metric) && rvalid) ? xn_in[row] : 0.0f;
        u64 best = ~0ull;
#pragma unroll
        for (int c = 0; c < CT; ++c) {
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int64_t k = kbase + c * 32 + mfma_row(q, h);
                float d;
                if (VQ_IS_L2(metric)) {
                    float t = (acc[c][q] + xn) + ((k < K) ? en_in[k] : 0.0f);
                    t = (t < 0.0f) ? 0.0f : t;
                    d = sqrtf(t);
                } else {
                    d = cos_distance(acc[c][q], metric);
                }
                if (MODE == 0) {
                    if (k < K) { u64 key = dist_key(d, (uint32_t)k); best = key < best ? key : best; }
                } else if (MODE == 1) {
                    u64 key = (rvalid && k < K) ? dist_key(d, (uint32_t)row) : ~0ull;
#pragma unroll
                    for (int off = 16; off >= 1; off >>= 1) {
                        u64 o = __shfl_xor(key, off, 64);
                        key = o < key ? o : key;
                    }
                    if (j == 0 && k < K && key != ~0ull) atomicMin(&keys[k], key);
                } else {
                    if (rvalid && k < K) dout[row * K + k] = d;
                }
            }
        }
        if (MODE == 0) {
            u64 o = __shfl_xor(best, 32, 64);
            best = o < best ? o : best;
            if (h == 0 && rvalid && best != ~0ull) atomicMin(&keys[row], best);
        }
    }
}

// decode keys -> idx (+hist, +dmin).  rows = flagged list (device count) or all N
__global__ void finalize_kernel(const u64 *keys, const int *row_list, const int *nrows_dev, int64_t N, int64_t *idx,
                                float *dmin, int32_t *hist) {
    const int64_t nrows = row_list ? (int64_t)(*nrows_dev) : N;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nrows; i += (int64_t)gridDim.x * blockDim.x) {
        int64_t row = row_list ? (int64_t)row_list[i] : i;
        u64 key = keys[row];
        uint32_t k = (uint32_t)(key & 0xFFFFFFFFull);
        idx[row] = (int64_t)k;
        if (hist) atomicAdd(&hist[k], 1);
        if (dmin) {
            u64 hi = key >> 32;
            float d;
            if (hi == 0) d = __uint_as_float(0x7FC00000u);
            else {
                uint32_t b = (uint32_t)(hi - 1ull);
                b = (b & 0x80000000u) ? (b & 0x7FFFFFFFu) : ~b;
                d = __uint_as_float(b);
            }
            dmin[row] = d;
        }
    }
}

__global__ void fill_u64_kernel(u64 *p, int64_t n, u64 v) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) p[i] = v;
}

// ------------------------------------------------------------------------------------------------
// decode / STE / loss partial sums, histogram, scatter-add, gathers, codebook updates
// ------------------------------------------------------------------------------------------------
// wave per token row, 4 elements (16 B) per lane and step, grid-stride over rows:
// z = e[idx], z_ste = x + (z - x), sse += (z-x)^2 (fp32 within a lane's 4 elements, double across; one atomic per block)
// streamed outputs: non-temporal stores keep the gathered codebook rows resident in L2 / Infinity Cache
__device__ __forceinline__ void nt_store4(float *q, float a, float b, float c, float d) {
    __builtin_nontemporal_store(a, q); __builtin_nontemporal_store(b, q + 1);
    __builtin_nontemporal_store(c, q + 2); __builtin_nontemporal_store(d, q + 3);   // merged into one dwordx4 ... nt
}

// NT: the outputs (and the latents) are larger than the Infinity Cache and are streamed with non-temporal accesses;
// smaller batches keep normal stores so that the consumer of z finds it in cache.
template <int DT, int NT>
// mse != nullptr: `sse` is a 16-byte scratch {double sum; int ticket; int pad} that is zero on entry; the workgroup that
// draws the last ticket writes mean((z - x)^2) as fp32 to mse[0] and mse[1] (the codebook and the commitment term share
// the value), mse[2] = mse[0] + beta * mse[1] (VQGANLoss), mse[3] = 0, and leaves the scratch zeroed for the next call — no zero-fill, division or cast kernels around the launch.
__global__ __launch_bounds__(1024) void gather_ste_loss_kernel(const void *x, const float *e, const int64_t *idx, int64_t N,
                                                              int D, float *z, float *zste, double *sse,
                                                              float *mse = nullptr, float beta = 0.0f) {
    __shared__ double red[16];                                // 16 waves per block: one atomic per 16 waves
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double s = 0.0;
    const bool vec = (D % 4) == 0;
    const int64_t stride = (int64_t)gridDim.x * 16;
    int64_t n = (int64_t)blockIdx.x * 16 + wave;
    for (; n < N; n += stride) {
        const float *er = e + idx[n] * D;
        if (vec) {
            for (int d = lane * 4; d < D; d += 256) {
                float4 zv = *(const float4 *)(er + d);
                float xv[4];
                if (DT == 0) {
                    float4 t = *(const float4 *)((const float *)x + n * D + d);
                    xv[0] = t.x; xv[1] = t.y; xv[2] = t.z; xv[3] = t.w;
                } else {
                    const uint32_t *px = (const uint32_t *)((const uint16_t *)x + n * D + d);
                    uint2 t;
                    if (NT) { t.x = __builtin_nontemporal_load(px); t.y = __builtin_nontemporal_load(px + 1); }
                    else t = *(const uint2 *)px;
                    xv[0] = __uint_as_float(t.x << 16); xv[1] = __uint_as_float(t.x & 0xFFFF0000u);
                    xv[2] = __uint_as_float(t.y << 16); xv[3] = __uint_as_float(t.y & 0xFFFF0000u);
                }
                float d0 = zv.x - xv[0], d1 = zv.y - xv[1], d2 = zv.z - xv[2], d3 = zv.w - xv[3];
                if (NT) {
                    if (z) nt_store4(z + n * D + d, zv.x, zv.y, zv.z, zv.w);
                    if (zste) nt_store4(zste + n * D + d, xv[0] + d0, xv[1] + d1, xv[2] + d2, xv[3] + d3);
                } else {
                    if (z) *(float4 *)(z + n * D + d) = zv;
                    if (zste) *(float4 *)(zste + n * D + d) = make_float4(xv[0] + d0, xv[1] + d1, xv[2] + d2, xv[3] + d3);
                }
                s += (double)((d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3));
            }
        } else {
            for (int d = lane; d < D; d += 64) {
                float xv = load_elem<DT>(x, n * D + d), zv = er[d];
                float df = zv - xv;
                if (z) z[n * D + d] = zv;
                if (zste) zste[n * D + d] = xv + df;
                s += (double)(df * df);
            }
        }
    }
    if (sse) {
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off, 64);
        if (lane == 0) red[wave] = s;
        __syncthreads();
        if (threadIdx.x == 0) {
            double t = 0.0;
#pragma unroll
            for (int i = 0; i < 16; ++i) t += red[i];
            if (!mse) {
                atomicAdd(sse, t);
            } else {
                int *ticket = (int *)(sse + 1);
                // the sum must be performed before the ticket is taken: a RETURNING atomic is complete when its value is
                // back, so waiting for the value orders the two without a release fence (an agent-scope __threadfence()
                // writes the XCD's dirty L2 lines back — this kernel's own 0.5 GB of output — at every workgroup's end)
                const double before = atomicAdd(sse, t);
                asm volatile("" :: "v"(before) : "memory");
                if (atomicAdd(ticket, 1) == (int)gridDim.x - 1) {
                    const double total = __hip_atomic_load(sse, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const float mean = (float)(total / ((double)N * (double)D));
                    mse[0] = mean; mse[1] = mean;
                    const float weighted = beta * mean;            // VQGANLoss: codebook + beta * commitment (losses.py:126),
                    mse[2] = mean + weighted; mse[3] = 0.0f;       // two roundings like the reference's two ops
                    __hip_atomic_store(sse, 0.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        }
    }
}

// The same pass with the OUTPUT written as the feature map [B, D, HW] the decoder side consumes — the reference's
// '(b h w) c -> b c h w' + .contiguous() of models/base.py:126-127 folded into the gather: a workgroup takes 64 tokens, reads
// codebook rows and latents token-major (256-byte rows per wave-instruction), turns 64 x 64 tiles through LDS and writes them
// with the tokens along the lanes (256 contiguous bytes per channel).  x == nullptr: plain decode (z = e[idx], no loss).
// Measured at 65 536 tokens x 256 channels: 43 us against 26-29 for the token-major kernel's fully contiguous rows — the
// strided 256-byte segments are the cost (whole 1 KiB codebook rows per instruction with 128-byte output segments: 62 us;
// non-temporal stores: 45 us; loading the next 64-channel chunk while the current one is in LDS: -1 us, kept).
// mse / sse scratch: as gather_ste_loss_kernel.
template <int DT>
__global__ __launch_bounds__(256) void gather_ste_map_kernel(const void *__restrict__ x, const float *__restrict__ e,
                                                             const int64_t *__restrict__ idx, int64_t N, int D, int64_t hw,
                                                             float *__restrict__ out_map, double *sse, float *mse, float beta) {
    __shared__ float tile[64][65];
    __shared__ double red[4];
    __shared__ int64_t code_s[64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double s = 0.0;
    const int64_t ntiles = (N + 63) / 64;
    // vector form: 4 channels per lane on the way in (16-byte loads of codebook rows and latents), 4 tokens per lane on the
    // way out (16-byte stores: 64 tokens of a channel = 256 contiguous bytes) — needs D % 4 == 0 and 4-token groups that stay
    // inside one image and aligned (hw % 4 == 0)
    const bool vec = (D % 4) == 0 && (hw % 4) == 0;
    for (int64_t tb = blockIdx.x; tb < ntiles; tb += gridDim.x) {
        const int64_t n0 = tb * 64;
        __syncthreads();
        if (threadIdx.x < 64) code_s[threadIdx.x] = (n0 + threadIdx.x < N) ? idx[n0 + threadIdx.x] : 0;
        __syncthreads();
        if (vec) {
            // chunk c0 + 64 is loaded while chunk c0 goes through LDS (two register sets)
            const int cl = 4 * (lane & 15);
            const int t4 = 4 * (lane & 15);
            const int64_t nw = n0 + t4;
            const int64_t wbase = (nw < N) ? ((nw / hw) * (int64_t)D * hw + (nw % hw)) : 0;
            float4 zc[4], zn[4];
            typename std::conditional<DT == 0, float4, uint2>::type xc[4], xnx[4];
            auto load_chunk = [&](int c0, float4 (&zr)[4], decltype(xc) &xr) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {                 // 16 tokens per pass of the workgroup: 4 per wave, 16 lanes each
                    const int tl = 16 * i + 4 * wave + (lane >> 4);
                    const int64_t n = n0 + tl;
                    if (n < N && c0 + cl < D) {
                        zr[i] = *(const float4 *)(e + code_s[tl] * D + c0 + cl);
                        if (x != nullptr) {
                            if constexpr (DT == 0) xr[i] = *(const float4 *)((const float *)x + n * D + c0 + cl);
                            else xr[i] = *(const uint2 *)((const uint16_t *)x + n * D + c0 + cl);
                        }
                    }
                }
            };
            load_chunk(0, zc, xc);
            for (int c0 = 0; c0 < D; c0 += 64) {
                if (c0 + 64 < D) load_chunk(c0 + 64, zn, xnx);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int tl = 16 * i + 4 * wave + (lane >> 4);
                    const int64_t n = n0 + tl;
                    float o[4] = {0.0f, 0.0f, 0.0f, 0.0f};
                    if (n < N && c0 + cl < D) {
                        const float4 zv = zc[i];
                        if (x != nullptr) {
                            float xv[4];
                            if constexpr (DT == 0) { xv[0] = xc[i].x; xv[1] = xc[i].y; xv[2] = xc[i].z; xv[3] = xc[i].w; }
                            else {
                                xv[0] = __uint_as_float(xc[i].x << 16); xv[1] = __uint_as_float(xc[i].x & 0xFFFF0000u);
                                xv[2] = __uint_as_float(xc[i].y << 16); xv[3] = __uint_as_float(xc[i].y & 0xFFFF0000u);
                            }
                            const float d0 = zv.x - xv[0], d1 = zv.y - xv[1], d2 = zv.z - xv[2], d3 = zv.w - xv[3];
                            s += (double)((d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3));
                            o[0] = xv[0] + d0; o[1] = xv[1] + d1; o[2] = xv[2] + d2; o[3] = xv[3] + d3;
                        } else {
                            o[0] = zv.x; o[1] = zv.y; o[2] = zv.z; o[3] = zv.w;
                        }
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j) tile[cl + j][tl] = o[j];      // bank (cl + j + tl) % 64: conflict-free
                }
                __syncthreads();
#pragma unroll
                for (int i = 0; i < 4; ++i) {                 // 16 channels per pass: 4 per wave, 16 lanes (64 tokens) each
                    const int dl = 16 * i + 4 * wave + (lane >> 4);
                    if (c0 + dl < D && nw < N) {
                        float *dst = out_map + wbase + (int64_t)(c0 + dl) * hw;
                        if (nw + 3 < N) *(float4 *)dst = make_float4(tile[dl][t4], tile[dl][t4 + 1], tile[dl][t4 + 2], tile[dl][t4 + 3]);
                        else
                            for (int j = 0; j < 4 && nw + j < N; ++j) dst[j] = tile[dl][t4 + j];
                    }
                }
                __syncthreads();
#pragma unroll
                for (int i = 0; i < 4; ++i) { zc[i] = zn[i]; xc[i] = xnx[i]; }
            }
            continue;
        }
        for (int c0 = 0; c0 < D; c0 += 64) {
            const int d = c0 + lane;
            const int64_t nw = n0 + lane;                     // this lane's token in the write phase: its position in the map
            const int64_t wbase = (nw < N) ? ((nw / hw) * (int64_t)D * hw + (nw % hw)) : 0;
#pragma unroll 4
            for (int i = 0; i < 16; ++i) {                    // wave w: tokens 16w .. 16w+15, lane = channel
                const int tl = wave * 16 + i;
                const int64_t n = n0 + tl;
                float o = 0.0f;
                if (n < N && d < D) {
                    const float zv = e[code_s[tl] * D + d];
                    if (x != nullptr) {
                        const float xv = load_elem<DT>(x, n * D + d);
                        const float df = zv - xv;
                        s += (double)(df * df);
                        o = xv + df;
                    } else {
                        o = zv;
                    }
                }
                tile[lane][tl] = o;
            }
            __syncthreads();
#pragma unroll 4
            for (int i = 0; i < 16; ++i) {                    // wave w: channels w, w+4, ...; lane = token
                const int dl = wave + 4 * i;
                if (nw < N && c0 + dl < D) out_map[wbase + (int64_t)(c0 + dl) * hw] = tile[dl][lane];
            }
            __syncthreads();
        }
    }
    if (sse != nullptr) {
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off, 64);
        if (lane == 0) red[wave] = s;
        __syncthreads();
        if (threadIdx.x == 0) {
            const double t = (red[0] + red[1]) + (red[2] + red[3]);
            int *ticket = (int *)(sse + 1);
            const double before = atomicAdd(sse, t);          // (returning atomic: complete before the ticket is taken)
            asm volatile("" :: "v"(before) : "memory");
            if (atomicAdd(ticket, 1) == (int)gridDim.x - 1) {
                const double total = __hip_atomic_load(sse, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const float mean = (float)(total / ((double)N * (double)D));
                mse[0] = mean; mse[1] = mean;
                const float weighted = beta * mean;
                mse[2] = mean + weighted; mse[3] = 0.0f;
                __hip_atomic_store(sse, 0.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
}

__global__ void hist_kernel(const int64_t *idx, int64_t N, int64_t K, int32_t *hist) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (int64_t)gridDim.x * blockDim.x) {
        int64_t k = idx[i];
        if (k >= 0 && k < K) atomicAdd(&hist[k], 1);
    }
}

// K <= 32768: block-private histogram in LDS, flushed with lane-contiguous atomics (a wave-instruction covers 64
// neighbouring bins = 256 bytes) instead of 64 scattered ones per wave-instruction
__global__ __launch_bounds__(1024) void hist_lds_kernel(const int64_t *__restrict__ idx, int64_t N, int K,
                                                        int32_t *__restrict__ hist) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    int *h = (int *)lds;
    for (int k = threadIdx.x; k < K; k += 1024) h[k] = 0;
    __syncthreads();
    for (int64_t i = (int64_t)blockIdx.x * 1024 + threadIdx.x; i < N; i += (int64_t)gridDim.x * 1024) {
        const int64_t k = idx[i];
        if (k >= 0 && k < K) atomicAdd(&h[k], 1);
    }
    __syncthreads();
    for (int k = threadIdx.x; k < K; k += 1024) {
        const int v = h[k];
        if (v) atomicAdd(&hist[k], v);
    }
}

// wave per source row; lanes sweep the row so each atomic wave-instruction adds 256 contiguous bytes
__global__ void scatter_add_rows_kernel(const float *src, const int64_t *idx, int64_t N, int64_t K, int D, float *dst) {
    int64_t n = (int64_t)blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6);
    int lane = threadIdx.x & 63;
    if (n >= N) return;
    int64_t k = idx[n];
    if (k < 0 || k >= K) return;
    for (int d = lane; d < D; d += 64) atomicAdd(&dst[k * D + d], src[n * D + d]);
}

template <int DT>
__global__ void gather_rows_kernel(const void *x, const int64_t *row_idx, int64_t K, int D, float *out) {
    int64_t k = (int64_t)blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6);
    int lane = threadIdx.x & 63;
    if (k >= K) return;
    int64_t n = row_idx[k];
    for (int d = lane; d < D; d += 64) out[k * D + d] = load_elem<DT>(x, n * D + d);
}

// VQ-KD codebook update, wave per code (callbacks.py:66-70,126-128,73-75)
__global__ void vqkd_update_kernel(float *w, const int64_t *hist, const float *sums, int64_t K, int D, float decay,
                                   int centroid_only) {
    int64_t k = (int64_t)blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6);
    int lane = threadIdx.x & 63;
    if (k >= K) return;
    int64_t occ = hist[k];
    float cnt = (float)(occ > 0 ? occ : 1);
    if (centroid_only) {   // VQKDCallback._kmeans alone (callbacks.py:66-70): where(occurred, sums/count, w)
        if (occ > 0)
            for (int d = lane; d < D; d += 64) w[k * D + d] = sums[k * D + d] / cnt;
        return;
    }
    // c = where(occurred, sums / max(count,1), w); then normalize
    float p = 0.0f;
    for (int d = lane; d < D; d += 64) {
        float c = (occ > 0) ? sums[k * D + d] / cnt : w[k * D + d];
        p = fmaf(c, c, p);
    }
    p = wave_sum_tree(p);
    float nrm = sqrtf(p), den = (nrm < 1e-12f) ? 1e-12f : nrm;
    float om = 1.0f - decay;
    float q = 0.0f;
    for (int d = lane; d < D; d += 64) {
        float c = (occ > 0) ? sums[k * D + d] / cnt : w[k * D + d];
        c = c / den;
        float v = w[k * D + d] * decay + c * om;       // todd.utils.ema
        q = fmaf(v, v, q);
    }
    q = wave_sum_tree(q);
    float nrm2 = sqrtf(q), den2 = (nrm2 < 1e-12f) ? 1e-12f : nrm2;
    for (int d = lane; d < D; d += 64) {
        float c = (occ > 0) ? sums[k * D + d] / cnt : w[k * D + d];
        c = c / den;
        float v = w[k * D + d] * decay + c * om;
        w[k * D + d] = v / den2;
    }
}

// CVQ-VAE update, wave per code (quantizer_callback.py:94-102)
__device__ __forceinline__ float cvq_decay_of(float pk, int64_t K, float ema_decay, float eps) {
    return 1.0f - expf(-pk * (float)K * 10.0f / (1.0f - ema_decay) - eps);
}

__global__ void cvq_update_kernel(float *w, float *p, const int64_t *hist, int64_t numel, const int64_t *numel_dev,
                                  const float *anchors, int64_t K, int D, float ema_decay, float eps, int stage) {
    int64_t k = (int64_t)blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6);
    int lane = threadIdx.x & 63;
    if (k >= K) return;
    // stage bit 0: p = ema(p, hist/numel); stage bit 1: w = ema(w, anchors, decay(p))
    float pk = p[k];
    if (stage & 1) {
        if (numel_dev) numel = *numel_dev;      // all-reduced token count left on the device (no host sync)
        float freq = (float)hist[k] / (float)numel;
        pk = pk * ema_decay + freq * (1.0f - ema_decay);
    }
    if (stage & 2) {
        float decay = cvq_decay_of(pk, K, ema_decay, eps);
        float om = 1.0f - decay;
        for (int d = lane; d < D; d += 64) w[k * D + d] = w[k * D + d] * decay + anchors[k * D + d] * om;
    }
    if (lane == 0 && (stage & 1)) p[k] = pk;
}

// The whole one-rank CVQ-VAE update in one launch, wave per code: probability EMA from the int32 epilogue histogram,
// decay, NearestAnchor's row gather x[col_idx[k]] and the blend — the same expressions, in the same order, as stage 1,
// vqhip_gather_rows and stage 2 above (bit-identical results); w_out may alias w_in and p_out may alias p_in.
template <int DT>
__global__ void cvq_step_kernel(const float *w_in, float *w_out, const float *p_in, float *p_out, const int32_t *hist,
                                int64_t numel, const void *x, const int64_t *col_idx, int64_t K, int D, float ema_decay,
                                float eps) {
    const int64_t k = (int64_t)blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (k >= K) return;
    const float freq = (float)hist[k] / (float)numel;
    const float pk = p_in[k] * ema_decay + freq * (1.0f - ema_decay);
    const float decay = cvq_decay_of(pk, K, ema_decay, eps), om = 1.0f - decay;
    const int64_t row = col_idx[k];
    for (int d = lane; d < D; d += 64) w_out[k * D + d] = w_in[k * D + d] * decay + load_elem<DT>(x, row * D + d) * om;
    if (lane == 0) p_out[k] = pk;
}

// decay_k of every code (the same expression, bit for bit): decay_k == 1.0f means the code's anchor is multiplied by 0
__global__ void cvq_decay_kernel(const float *p, int64_t K, float ema_decay, float eps, float *decay) {
    int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k < K) decay[k] = cvq_decay_of(p[k], K, ema_decay, eps);
}

// the w update restricted to the listed codes: w[rows[i]] = w[rows[i]]*decay + anchors_sub[i]*(1-decay)
__global__ void cvq_update_rows_kernel(float *w, const float *p, const int64_t *rows, const float *anchors_sub, int64_t M,
                                       int64_t K, int D, float ema_decay, float eps) {
    int64_t i = (int64_t)blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6);
    int lane = threadIdx.x & 63;
    if (i >= M) return;
    const int64_t k = rows[i];
    if (k < 0 || k >= K) return;
    const float decay = cvq_decay_of(p[k], K, ema_decay, eps), om = 1.0f - decay;
    for (int d = lane; d < D; d += 64) w[k * D + d] = w[k * D + d] * decay + anchors_sub[i * D + d] * om;
}

// ------------------------------------------------------------------------------------------------
// elementwise pieces of the autograd path (losses.py:50,62; utils/ste.py:10; F.normalize backward)
// ------------------------------------------------------------------------------------------------
// sse += sum (a-b)^2 (double accumulation across lanes/blocks), optional out = (a-b)*scale
template <int DTA, int DTB>
__global__ __launch_bounds__(256) void diff_kernel(const void *a, const void *b, int64_t n, float scale,
                                                   const float *scale_dev, float *out, double *sse) {
    __shared__ double red[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double s = 0.0;
    if (scale_dev) scale *= *scale_dev;          // upstream scalar gradient left on the device
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float df = load_elem<DTA>(a, i) - load_elem<DTB>(b, i);
        if (out) out[i] = df * scale;
        s += (double)(df * df);
    }
    if (sse) {
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off, 64);
        if (lane == 0) red[wave] = s;
        __syncthreads();
        if (threadIdx.x == 0) atomicAdd(sse, (red[0] + red[1]) + (red[2] + red[3]));
    }
}

// out = x + (z - x)
template <int DT>
__global__ void ste_kernel(const void *x, const float *z, int64_t n, float *out) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float xv = load_elem<DT>(x, i);
        out[i] = xv + (z[i] - xv);
    }
}

// backward of y = v / max(|v|, eps) per row: gv = (g - y*(y.g)) / max(|v|, eps)   (rows with |v| < eps: g / eps)
template <int DT>
__global__ void normalize_bwd_kernel(const void *v, const float *g, int64_t R, int D, float eps, float *gv) {
    int64_t r = (int64_t)blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6);
    int lane = threadIdx.x & 63;
    if (r >= R) return;
    float p = 0.0f;
    for (int d = lane; d < D; d += 64) { float a = load_elem<DT>(v, r * D + d); p = fmaf(a, a, p); }
    p = wave_sum_tree(p);
    float nrm = sqrtf(p);
    bool clamped = nrm < eps;
    float den = clamped ? eps : nrm;
    float dot = 0.0f;
    for (int d = lane; d < D; d += 64) dot = fmaf(load_elem<DT>(v, r * D + d) / den, g[r * D + d], dot);
    dot = wave_sum_tree(dot);
    for (int d = lane; d < D; d += 64) {
        float y = load_elem<DT>(v, r * D + d) / den;
        gv[r * D + d] = clamped ? g[r * D + d] / den : (g[r * D + d] - y * dot) / den;
    }
}

// fused backward of the quantizer forward, z = W[idx], z_ste = x + sg(z - x), m_cb = mse(z, sg x), m_cm = mse(sg z, x):
//   grad_x = g_zste + g_cm*(2/ND)*(x - z)        grad_W[idx] += g_cb*(2/ND)*(z - x)
// wave per token row; g_cb / g_cm are device scalars (upstream gradients of the two MSE values), nullable = 0.
template <int DT>
__global__ __launch_bounds__(256) void vq_backward_kernel(const void *x, const float *e, const int64_t *idx, int64_t N, int D,
                                                          const float *g_zste, const float *g_cb, const float *g_cm,
                                                          float *grad_x, float *grad_w, const float *g_comb = nullptr,
                                                          float beta = 0.0f) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float s = 2.0f / ((float)N * (float)D);
    // g_comb: upstream gradient of the combined value m_cb + beta * m_cm (VQGANLoss finished inside the forward kernel)
    const float gc = g_comb ? *g_comb : 0.0f;
    const float kx = ((g_cm ? *g_cm : 0.0f) + beta * gc) * s, kw = ((g_cb ? *g_cb : 0.0f) + gc) * s;
    const bool do_w = grad_w && kw != 0.0f;
    // float atomics want the 64 lanes on 256 contiguous bytes (measured: 4 consecutive floats per lane is 3.5x slower),
    // so the vector path is for the atomic-free case (grad_x only: the ordered route computes grad_w elsewhere)
    const bool vec = (D % 4) == 0 && !do_w;
    for (int64_t n = (int64_t)blockIdx.x * 4 + wave; n < N; n += (int64_t)gridDim.x * 4) {
        const int64_t k = idx[n];
        if (vec) {
            for (int d = 4 * lane; d < D; d += 256) {
                const float4 zv = *(const float4 *)(e + k * D + d);
                float xv[4];
                if (DT == 0) {
                    const float4 t = *(const float4 *)((const float *)x + n * D + d);
                    xv[0] = t.x; xv[1] = t.y; xv[2] = t.z; xv[3] = t.w;
                } else {
                    const uint2 t = *(const uint2 *)((const uint16_t *)x + n * D + d);
                    xv[0] = __uint_as_float(t.x << 16); xv[1] = __uint_as_float(t.x & 0xFFFF0000u);
                    xv[2] = __uint_as_float(t.y << 16); xv[3] = __uint_as_float(t.y & 0xFFFF0000u);
                }
                const float d0 = zv.x - xv[0], d1 = zv.y - xv[1], d2 = zv.z - xv[2], d3 = zv.w - xv[3];
                if (grad_x) {
                    float4 gz = make_float4(0, 0, 0, 0);
                    if (g_zste) gz = *(const float4 *)(g_zste + n * D + d);
                    *(float4 *)(grad_x + n * D + d) = make_float4(gz.x - kx * d0, gz.y - kx * d1, gz.z - kx * d2, gz.w - kx * d3);
                }
            }
        } else {
            for (int d = lane; d < D; d += 64) {
                float xv = load_elem<DT>(x, n * D + d), zv = e[k * D + d];
                float df = zv - xv;
                if (grad_x) grad_x[n * D + d] = (g_zste ? g_zste[n * D + d] : 0.0f) - kx * df;
                if (do_w) atomicAdd(&grad_w[k * D + d], kw * df);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// deterministic codebook-side sums: tokens ordered by code (stable), then one sequential sum per code
// ------------------------------------------------------------------------------------------------
// The atomic scatter-adds above sum in arrival order, which differs from run to run in the last bits (SURVEY.md §7
// hard part 9).  The ordered route fixes the order — ascending token index within every code — with a stable
// counting sort built from integer operations only, and replaces N*D floating-point atomics by one pass over the
// gathered rows.
#define VQ_SORT_CHUNK 1024          // tokens per block of the counting sort

// (1) per-chunk code histograms in LDS -> blockhist[chunk][K]
__global__ __launch_bounds__(VQ_SORT_CHUNK) void sort_hist_kernel(const int64_t *__restrict__ idx, int64_t N, int K,
                                                                    int *__restrict__ blockhist) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    int *h = (int *)lds;
    for (int k = threadIdx.x; k < K; k += VQ_SORT_CHUNK) h[k] = 0;
    __syncthreads();
    const int64_t n = (int64_t)blockIdx.x * VQ_SORT_CHUNK + threadIdx.x;
    if (n < N) {
        const int64_t c = idx[n];
        if (c >= 0 && c < K) atomicAdd(&h[c], 1);
    }
    __syncthreads();
    int *out = blockhist + (int64_t)blockIdx.x * K;
    for (int k = threadIdx.x; k < K; k += VQ_SORT_CHUNK) out[k] = h[k];
}

// (2) per code: exclusive scan over the chunks (in place) and the total count; 8 chunk loads in flight per thread
__global__ void sort_colscan_kernel(int *__restrict__ blockhist, int nchunks, int K, int *__restrict__ counts) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= K) return;
    int run = 0;
    for (int b0 = 0; b0 < nchunks; b0 += 8) {
        int t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) t[u] = b0 + u < nchunks ? blockhist[(int64_t)(b0 + u) * K + k] : 0;
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (b0 + u < nchunks) { blockhist[(int64_t)(b0 + u) * K + k] = run; run += t[u]; }
    }
    counts[k] = run;
}

// (3) exclusive scan of the counts over the codes -> offsets[K+1]; one block of 1024 threads, K <= 32768:
// coalesced load into LDS, 32 contiguous codes per thread, wave scans, coalesced store
__global__ __launch_bounds__(1024) void sort_offsets_kernel(const int *__restrict__ counts, int K, int *__restrict__ offsets) {
    __shared__ int buf[32768 + 1024];                      // padded: element i lives at i + i/32
    __shared__ int wsum[16];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    for (int i = t; i < 32768; i += 1024) buf[i + (i >> 5)] = i < K ? counts[i] : 0;
    __syncthreads();
    int sum = 0;
    int *mine = buf + t * 33;                              // codes 32t .. 32t+31
#pragma unroll
    for (int i = 0; i < 32; ++i) { const int v = mine[i]; mine[i] = sum; sum += v; }
    int incl = sum;                                        // inclusive scan of the per-thread sums: wave, then block
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const int v = __shfl_up(incl, off, 64); if (lane >= off) incl += v; }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    int base = 0;
    for (int w = 0; w < wave; ++w) base += wsum[w];
    const int excl = base + incl - sum;
#pragma unroll
    for (int i = 0; i < 32; ++i) mine[i] += excl;
    __syncthreads();
    for (int i = t; i < K; i += 1024) offsets[i] = buf[i + (i >> 5)];
    if (t == 1023) offsets[K] = base + incl;
}

// (4) placement: position = offsets[code] + (tokens of this code in earlier chunks) + (earlier tokens of this code in
// this chunk); the last term by comparing against the chunk's codes in LDS (broadcast reads)
__global__ __launch_bounds__(VQ_SORT_CHUNK) void sort_place_kernel(const int64_t *__restrict__ idx, int64_t N, int K,
                                                                     const int *__restrict__ blockhist,
                                                                     const int *__restrict__ offsets, int *__restrict__ order) {
    __shared__ __attribute__((aligned(16))) int codes[VQ_SORT_CHUNK];
    const int t = threadIdx.x;
    const int64_t n = (int64_t)blockIdx.x * VQ_SORT_CHUNK + t;
    int c = -1;
    if (n < N) { const int64_t v = idx[n]; c = (v >= 0 && v < K) ? (int)v : -1; }
    codes[t] = c;
    __syncthreads();
    if (c < 0) return;
    int r = 0;
    const int t4 = t & ~3;
    for (int j = 0; j < t4; j += 4) {
        const int4 q = *(const int4 *)(codes + j);
        r += (q.x == c) + (q.y == c) + (q.z == c) + (q.w == c);
    }
    for (int j = t4; j < t; ++j) r += codes[j] == c;
    order[offsets[c] + blockhist[(int64_t)blockIdx.x * K + c] + r] = (int)n;
}

// (5) ordered sums.  The sorted order is cut into ranges of 64 positions, one wave per range (balanced whatever the
// code frequencies are).  A wave adds the rows of its positions in order, 8 row loads in flight, and closes a sum
// whenever the code changes: a code that lies inside the range is written to dst directly; the piece of a code that
// began in an earlier range goes to partial[range][0] ("head"), the piece of a code that continues into the next
// range to partial[range][1] ("tail").  segsum_fixup_kernel then, per code: zero row if unused; for a code spanning
// ranges first..last: tail[first] + head[first+1] + ... + head[last], added in that order.  The association is a
// function of the counts only, hence reproducible.
//   MODE 0: rows = src[n]                                  (k-means centroid sums, callbacks.py:60-64)
//   MODE 1: rows = kw * (e_k - x_n), kw = g_cb * 2/(N*D)  (codebook gradient of the codebook loss)
#define VQ_SEG_RANGE 64
template <int MODE, int DT>
__global__ __launch_bounds__(256) void segsum_rows_kernel(const void *__restrict__ src, const float *__restrict__ e,
                                                          const int64_t *__restrict__ idx, const int *__restrict__ order,
                                                          const int *__restrict__ offsets, int64_t N, int K, int D,
                                                          const float *__restrict__ g_cb, float *__restrict__ dst,
                                                          float *__restrict__ partial) {
    const int lane = threadIdx.x & 63;
    const int64_t gw = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nw = ((int64_t)gridDim.x * blockDim.x) >> 6;
    const float kw = (MODE == 1) ? (g_cb ? *g_cb : 0.0f) * (2.0f / ((float)N * (float)D)) : 0.0f;
    const int total = offsets[K];                          // tokens with a valid code
    const int64_t nranges = (total + VQ_SEG_RANGE - 1) / VQ_SEG_RANGE;
    for (int64_t j = gw; j < nranges; j += nw) {
        const int p0 = (int)(j * VQ_SEG_RANGE);
        const int pos = p0 + lane;
        const int my_n = pos < total ? order[pos] : -1;
        const int my_c = my_n >= 0 ? (int)idx[my_n] : -1;
        const int c_first = __shfl(my_c, 0, 64);
        const bool starts_before = offsets[c_first] < p0;
        for (int d0 = 0; d0 < D; d0 += 256) {
            const int d = d0 + 4 * lane;
            const bool in = d < D;                        // D % 4 == 0 on this path
            float4 acc = make_float4(0, 0, 0, 0);
            int cur = c_first;
            auto flush = [&](int c) __attribute__((always_inline)) {
                if (!in) return;
                float *out;
                if (c == c_first && starts_before) out = partial + (j * 2 + 0) * D + d;
                else if (offsets[c + 1] > p0 + VQ_SEG_RANGE) out = partial + (j * 2 + 1) * D + d;
                else out = dst + (int64_t)c * D + d;
                *(float4 *)out = acc;
            };
            for (int b = 0; b < VQ_SEG_RANGE; b += 8) {
                float4 row[8], ek[8];
                int cc[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int n = __shfl(my_n, b + u, 64);
                    cc[u] = __shfl(my_c, b + u, 64);
                    row[u] = make_float4(0, 0, 0, 0); ek[u] = row[u];
                    if (cc[u] >= 0 && in) {
                        if (MODE == 0 || DT == 0) {
                            row[u] = *(const float4 *)((const float *)src + (int64_t)n * D + d);
                        } else {
                            const uint2 t = *(const uint2 *)((const uint16_t *)src + (int64_t)n * D + d);
                            row[u] = make_float4(__uint_as_float(t.x << 16), __uint_as_float(t.x & 0xFFFF0000u),
                                                 __uint_as_float(t.y << 16), __uint_as_float(t.y & 0xFFFF0000u));
                        }
                        if (MODE == 1) ek[u] = *(const float4 *)(e + (int64_t)cc[u] * D + d);
                    }
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    if (cc[u] < 0) continue;              // past the end of the order: wave-uniform
                    if (cc[u] != cur) { flush(cur); acc = make_float4(0, 0, 0, 0); cur = cc[u]; }
                    if (MODE == 0) { acc.x += row[u].x; acc.y += row[u].y; acc.z += row[u].z; acc.w += row[u].w; }
                    else {
                        acc.x += kw * (ek[u].x - row[u].x); acc.y += kw * (ek[u].y - row[u].y);
                        acc.z += kw * (ek[u].z - row[u].z); acc.w += kw * (ek[u].w - row[u].w);
                    }
                }
            }
            flush(cur);
        }
    }
}

__global__ __launch_bounds__(256) void segsum_fixup_kernel(const int *__restrict__ offsets, int K, int D,
                                                           const float *__restrict__ partial, float *__restrict__ dst) {
    const int lane = threadIdx.x & 63;
    const int64_t gw = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nw = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t k = gw; k < K; k += nw) {
        const int p0 = offsets[k], p1 = offsets[k + 1];
        if (p0 < p1 && p0 / VQ_SEG_RANGE == (p1 - 1) / VQ_SEG_RANGE) continue;   // written by its range
        const int first = p0 / VQ_SEG_RANGE, last = p0 < p1 ? (p1 - 1) / VQ_SEG_RANGE : first;
        for (int d = 4 * lane; d < D; d += 256) {
            float4 acc = make_float4(0, 0, 0, 0);
            if (p0 < p1) {
                acc = *(const float4 *)(partial + ((int64_t)first * 2 + 1) * D + d);
                for (int j = first + 1; j <= last; ++j) {
                    const float4 h = *(const float4 *)(partial + ((int64_t)j * 2 + 0) * D + d);
                    acc.x += h.x; acc.y += h.y; acc.z += h.z; acc.w += h.w;
                }
            }
            *(float4 *)(dst + k * D + d) = acc;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// callers of the path: BCHW <-> (BHW)C rearrangement and codebook metrics
// ------------------------------------------------------------------------------------------------
// 'b c h w -> (b h w) c' (models/base.py:124,140) as a 64x64 LDS-tiled transpose per image: in[b][c][p] -> out[b][p][c]
// (TO_TOKENS) or the inverse '(b h w) c -> b c h w' (base.py:126).  T = 2-byte or 4-byte element.
template <typename T>
__global__ __launch_bounds__(256) void transpose_kernel(const T *__restrict__ in, T *__restrict__ out, int64_t B, int R, int C) {
    // in: [B][R][C] -> out: [B][C][R]
    __shared__ T tile[64][65];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;     // 64 x 4
    const int64_t b = blockIdx.z;
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    const T *src = in + b * (int64_t)R * C;
    T *dst = out + b * (int64_t)R * C;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int r = r0 + ty + 4 * i, c = c0 + tx;
        if (r < R && c < C) tile[ty + 4 * i][tx] = src[(int64_t)r * C + c];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int c = c0 + ty + 4 * i, r = r0 + tx;
        if (r < R && c < C) dst[(int64_t)c * R + r] = tile[tx][ty + 4 * i];
    }
}

// CodebookUsageMetric / CodebookPPLMetric summaries (runners/metrics.py:58-73) from the accumulated counts:
// out[0] = #nonzero / K, out[1] = entropy of counts / sum(counts) in nats.  One block.
__global__ __launch_bounds__(1024) void codebook_metrics_kernel(const int64_t *counts, int64_t K, double *out) {
    __shared__ double red[3][16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double tot = 0.0, nz = 0.0;
    for (int64_t k = threadIdx.x; k < K; k += blockDim.x) { tot += (double)counts[k]; nz += counts[k] != 0 ? 1.0 : 0.0; }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) { tot += __shfl_xor(tot, off, 64); nz += __shfl_xor(nz, off, 64); }
    if (lane == 0) { red[0][wave] = tot; red[1][wave] = nz; }
    __syncthreads();
    tot = 0.0; nz = 0.0;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) { tot += red[0][i]; nz += red[1][i]; }
    double ent = 0.0;
    for (int64_t k = threadIdx.x; k < K; k += blockDim.x) {
        const double c = (double)counts[k];
        if (c > 0.0) { const double p = c / tot; ent -= p * log(p); }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) ent += __shfl_xor(ent, off, 64);
    __syncthreads();
    if (lane == 0) red[2][wave] = ent;
    __syncthreads();
    if (threadIdx.x == 0) {
        double e = 0.0;
        for (int i = 0; i < (int)(blockDim.x >> 6); ++i) e += red[2][i];
        out[0] = nz / (double)K;
        out[1] = tot > 0.0 ? e : 0.0;
    }
}

// bf16 -> fp32 copy (the column pass needs the latents as an fp32 "codebook")
__global__ void bf16_to_f32_kernel(const uint16_t *in, int64_t n, float *out) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        out[i] = bf16_to_f32(in[i]);
}

// ------------------------------------------------------------------------------------------------
// verification aid: the proposal scores of every (row, code) pair and the margin the decision uses
// ------------------------------------------------------------------------------------------------
// Same operands and MFMA sequence as coarse_kernel / rescan_kernel; one wave per (64 rows, stage).  Lets a test check
// |score - exact score| <= margin/2 for every pair against float64 (tests/test_gpu_parity.py::test_margin_holds).
template <int NSTEP, int TPS>
__global__ __launch_bounds__(256) void debug_scores_kernel(const char *__restrict__ ximg, const char *__restrict__ frag,
                                                           int64_t nstages, int64_t N, int64_t K, float *__restrict__ out) {
    constexpr int NS32 = NSTEP / 2;
    constexpr int NCH = TPS * NSTEP + VQ_AUX_CHUNKS(TPS);
    constexpr int STAGE_BYTES = NCH * VQ_CHUNK_BYTES;
    constexpr int TR = (NSTEP <= 16) ? 4 : (NSTEP <= 48 ? 2 : 1);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t ngroups = (N + 16 * TR - 1) / (16 * TR);
    for (int64_t item = (int64_t)blockIdx.x * 4 + wave; item < ngroups * nstages; item += (int64_t)gridDim.x * 4) {
        const int64_t fg = item / nstages, st = item % nstages;
        half8 xf[TR][NS32];
        int64_t tok[TR];
#pragma unroll
        for (int t = 0; t < TR; ++t) {
            tok[t] = fg * 16 * TR + t * 16 + (lane & 15);
            const int64_t tk = tok[t] < N ? tok[t] : N - 1;
            const char *xsrc = ximg + (tk >> 4) * (int64_t)(NS32 * VQ_CHUNK_BYTES) + ((lane >> 4) * 16 + (int)(tk & 15)) * 16;
#pragma unroll
            for (int s = 0; s < NS32; ++s) xf[t][s] = *(const half8 *)(xsrc + s * VQ_CHUNK_BYTES);
        }
        const char *base = frag + st * (int64_t)STAGE_BYTES;
        const char *aux = base + TPS * NSTEP * VQ_CHUNK_BYTES;
#pragma unroll 1
        for (int ti = 0; ti < TPS; ++ti) {
            f32x4 acc[2][TR];
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                f32x4 a4 = *(const f32x4 *)(aux + (ti * 32 + 16 * c + 4 * (lane >> 4)) * 4);
#pragma unroll
                for (int t = 0; t < TR; ++t) acc[c][t] = a4;
            }
#pragma unroll
            for (int ch = 0; ch < NSTEP; ++ch) {
                half8 a = *(const half8 *)(base + (ti * NSTEP + ch) * VQ_CHUNK_BYTES + lane * 16);
#pragma unroll
                for (int t = 0; t < TR; ++t)
                    acc[ch & 1][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, xf[t][ch >> 1], acc[ch & 1][t], 0, 0, 0);
            }
#pragma unroll
            for (int t = 0; t < TR; ++t)
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int64_t k = (st * TPS + ti) * 32 + tile_row16(e, lane);
                    if (tok[t] < N && k < K) out[tok[t] * K + k] = acc[e >> 2][t][e & 3];
                }
        }
    }
}

__global__ void debug_margin_kernel(const char *cb, VqCbLayout L, int64_t N, int metric, const float *xh2, const float *rho2,
                                    float *margin, float *scale) {
    int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const VqCbStats stv = cb_stats_view((const VqCbStats *)(cb + L.off_stats));
    const VqCbStats *st = &stv;
    if (n == 0) scale[0] = cb_scale(st);
    if (n < N) margin[n] = row_margin(st, L.Dp, metric, xh2[n], rho2[n]);
}

#include "vqhip_exchange_kernels.h"
