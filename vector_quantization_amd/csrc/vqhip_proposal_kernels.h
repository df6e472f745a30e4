// libvqhip device kernels, unit 3 of 8: the fp16-MFMA proposal pass (coarse_kernel) with its record merge and, optionally,
// the decision stage.  Included by vqhip_kernels.h.
#pragma once
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
    // coarse_kernel<..., XD != 0> (the token side made in the proposal prologue, no token image): the rows as the caller holds
    // them (row-major, D elements) and where slice 0 writes the per-row statistics x_prep_kernel would have written
    const void *xrows; float *xh2_w, *rho2_w, *xn_w;
};
template <bool AGENT>
__device__ __forceinline__ void decide_rows(int64_t n, bool oob, const VqCbStats *st, int Dp, int metric, int nslices,
                                            const float *rec, const float *xh2, const float *rho2, int64_t Np,
                                            const VqDecideOut &o, int *wcount, int *wbase, const float *rece2 = nullptr);

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
// Waves per SIMD the D <= 32 filtered forms WITHOUT group records are compiled for — the forms small batches take (below
// VQ_W32_MIN_N / VQ_GROUPS_MIN_N rows).  Four (128 registers) left them with 57-83 spilled registers and scratch traffic in the
// stream loop; two (256 registers) compile without a spill, and at these sizes there are at most two workgroups per CU to
// co-schedule anyway: 4096 x 8192 x 32 cosine 0.0705 -> 0.0538 ms per encode, 4096 x 16384 x 8 0.0768 -> 0.0592
// (profiles/r05_ab_small_d.txt; 12 544 rows: level).
#ifdef VQ_CLOCK_STAMPS
#define VQ_CLOCK_SLOTS 16384
__device__ unsigned long long vq_clock_dbg[2 * VQ_CLOCK_SLOTS];
#endif
#ifdef VQ_PHASE_STAMPS
// diagnostic build only (tools/phase_stamps.py): eight absolute s_memrealtime stamps (100 MHz) per workgroup, taken by its first
// lane — entry, first stages requested, first barrier passed, stream done, records stored, ticket drawn, end
#define VQ_PHASE_SLOTS 4096
__device__ unsigned long long vq_phase_dbg[8 * VQ_PHASE_SLOTS];
#define VQ_PHASE(i) do { if (threadIdx.x == 0 && blockIdx.x < VQ_PHASE_SLOTS) vq_phase_dbg[8 * blockIdx.x + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define VQ_PHASE(i) do { } while (0)
#endif
#ifdef VQ_STAGE_STAMPS
// diagnostic build only (tools/stage_stamps.py): s_memtime (shader clock) at five points of every iteration of the stage loop, kept in
// LDS behind the ring and dumped after the loop: [workgroup < 64][wave < 8][iteration < 24][5]
#define VQ_STAGE_WGS 64
#define VQ_STAGE_ITERS 24
__device__ unsigned long long vq_stage_dbg[VQ_STAGE_WGS * 8 * VQ_STAGE_ITERS * 5];
#define VQ_STAGE_LDS_EXTRA (8 * VQ_STAGE_ITERS * 5 * 8)
#define VQ_STAMP(k) do { if (lane == 0 && stamp_it < VQ_STAGE_ITERS) stamp_lds[(wave * VQ_STAGE_ITERS + stamp_it) * 5 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define VQ_STAGE_LDS_EXTRA 0
#define VQ_STAMP(k) do { } while (0)
#endif
#ifndef VQ_TICKET_NOFENCE
#define VQ_TICKET_NOFENCE 1
#endif
#ifndef VQ_D32_PLAIN_OCC
#define VQ_D32_PLAIN_OCC 2
#endif
// XD != 0 (1: bf16 rows, 2: fp32 rows; D == the padded dimension): the workgroup makes its token fragments itself, from the
// row-major latents — fp16 conversion with flush-to-zero, |xh|^2, |x - xh|^2 and the oracle-order |x|^2 exactly as x_prep_body
// computes them (same fma chains, same folding order: bit-identical statistics) — and slice 0 writes the statistics.  The
// 268 MB fp16 token image of the headline batch is then neither written nor read: pre_kernel's token side (85 us of HBM time
// at 524 288 x 256 bf16) is gone, for a prologue that converts what it loaded anyway.
template <int NSTEP, int TT, int WAVES, int TPS, int NBUF = 2, bool FILTER = false, bool NOAUX = false,
          bool GROUPS = false, int XD = 0>
__global__ __launch_bounds__(WAVES * 64, (FILTER && NSTEP <= 2) ? (GROUPS ? 4 : VQ_D32_PLAIN_OCC) : WAVES / 4) void coarse_kernel(
    const char *__restrict__ ximg, int64_t N, const char *__restrict__ frag, int64_t nstages, int nslices,
    float *__restrict__ rec, int64_t Np, const VqCbStats *__restrict__ cbst, const float *__restrict__ xh2,
    const float *__restrict__ rho2, int Dp, int metric, VqDecideOut dec, int pad_stage, int tpb) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    VQ_PHASE(0);
    static_assert(NSTEP % 2 == 0, "16x16x32 layout: 32-dim k-steps");
    static_assert(!GROUPS || FILTER, "group records are a form of the filtered epilogue");
    constexpr bool GBRANCH = TT >= VQ_GROUP_BRANCH_MIN_TT;
    constexpr bool PIPE = (TPS % 2) == 0;                // epilogue of tile t-1 in the MFMA shadow of tile t (ping-pong by parity)
    // One wave per SIMD (WAVES <= 4: D = 1024 ships with 48 tokens per wave where the two-wave form holds 16 — a third of the
    // LDS bytes per flop), one tile per stage: the two code halves of a tile run one after the other and each half's epilogue
    // sits in the MFMA shadow of the next half — the ping-pong of PIPE without a second accumulator set.
    constexpr bool PIPE_H = (TPS == 1) && (WAVES <= 4);
    static_assert(!(PIPE_H && (NOAUX || FILTER)), "half-tile ping-pong: plain large-D form only");
    constexpr int NS32 = NSTEP / 2;                      // k-steps of 32 dims
    constexpr int NCH = TPS * NSTEP + VQ_AUX_CHUNKS(TPS);   // chunks per stage (2 per k-step and tile, + aux)
    constexpr int STAGE_BYTES = NCH * VQ_CHUNK_BYTES;
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
    if (dec.n_dev != nullptr) {              // device-side row count: token blocks past it have nothing to do
        const int64_t nd = *dec.n_dev;
        N = nd < N ? nd : N;
        if ((int64_t)(blockIdx.x / nslices) * tpb * 16 >= N) return;
    }
    // Work assignment: workgroup = (token block tb, codebook slice sl of nslices).
    // tpb: 16-token tiles per workgroup, <= WAVES*TT (the host picks it so that the workgroups fill whole rounds of the
    // chip: launch_coarse).  Waves past it only help filling the ring; tiles past it belong to the next workgroup.
    // (A stream-K split of the (token block x stage) space into equal shares per CU was built for the D <= 32 kernels and
    //  measured slower, -4.6 % at configs[2], -14 % at small N: profiles/r02_ab_streamk.txt; removed.)
    const bool wave_active = wave * TT < tpb;
    cb_stats_publish(cbst);
    const int sl = blockIdx.x % nslices;
    const int64_t tb = blockIdx.x / nslices;
    const int64_t st0 = (nstages * sl) / nslices, st1 = (nstages * (sl + 1)) / nslices;
  // A loop that runs once.  It is what is left of the stream-K segment loop, and it stays because of what hipcc makes of the
  // kernel with it: without it the D = 1024 instantiation comes out with another schedule (165 instead of 193 VGPRs, the A
  // fragments fetched less far ahead) and runs 21 % slower (1.075 -> 1.301 ms at 65 536 x 8192, same-device A/B), every other
  // instantiation within 1 %.
  for (bool once = true; once; once = false) {

    // ---- prologue: this wave's token fragments straight from the fragment-major fp16 image ----
    half8 xf[TT][NS32];
    if constexpr (XD == 0) {
#pragma unroll
        for (int t = 0; t < TT; ++t) {
            int64_t tt = tb * tpb + wave * TT + t;
            tt = tt < ntt ? tt : ntt - 1;                    // out-of-range tiles read a valid tile and are never written
            const char *src = ximg + tt * (int64_t)(NS32 * VQ_CHUNK_BYTES) + lane * 16;
#pragma unroll
            for (int s = 0; s < NS32; ++s) xf[t][s] = *(const half8 *)(src + s * VQ_CHUNK_BYTES);
        }
    } else {
        static_assert(XD == 0 || (!FILTER && !GROUPS && !NOAUX && (NSTEP % 4) == 0), "token side in the prologue: plain D % 64 == 0 forms");
        const int q4 = lane >> 4, r16 = lane & 15;           // the fragment's own layout: token r16 of the tile, dims 32 s + 8 q4 .. + 8
        constexpr int Dr = NS32 * 32;
        // bf16 rows: a piece of 8 latents is 16 bytes, exactly the fragment's size — ALL the wave's pieces are requested into the
        // fragment registers themselves, then converted in place (one round trip to HBM for the whole prologue, as with the image:
        // a first form loaded into temporaries and hipcc, short of 128 more registers, made 32 round trips of it: +160 us per launch)
        if constexpr (XD == 1) {
            // Requested as whole rows — two 512-byte rows per wave-instruction, 1 KiB contiguous — and turned into the fragment
            // layout through a wave-private LDS tile (16 rows x 528 bytes: the 16-byte pad makes the fragment reads conflict-free)
            // in the ring's stages 2 and 3, which nothing writes before the first barrier of the stream.  (Requested in the fragment
            // layout itself a wave-instruction is 16 rows x 64 bytes: twice the requests for the same bytes, +5 us per workgroup.)
            static_assert(NS32 == 8 && NBUF >= 4 && 2 * STAGE_BYTES >= WAVES * 16 * 528, "row tile: D = 256, staged behind stage 1");
            char *stg = lds + 2 * STAGE_BYTES + wave * (16 * 528);
            const int hr = lane >> 5, c32 = lane & 31;
#pragma unroll
            for (int t = 0; t < TT; ++t) {
                const int64_t tok0 = (tb * tpb + wave * TT + t) * 16;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int64_t tok = tok0 + 2 * i + hr;
                    const int64_t trow = tok < N ? tok : N - 1;  // rows past the end repeat the last one and are never written
                    xf[t][i] = *(const half8 *)((const uint16_t *)dec.xrows + trow * Dr + 8 * c32);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < TT; ++t) {
#pragma unroll
                for (int i = 0; i < 8; ++i) *(half8 *)(stg + (2 * i + hr) * 528 + 16 * c32) = xf[t][i];
#pragma unroll
                for (int s = 0; s < NS32; ++s) xf[t][s] = *(const half8 *)(stg + r16 * 528 + 64 * s + 16 * q4);
            }
        }
#pragma unroll
        for (int t = 0; t < TT; ++t) {
            const int64_t tok = (tb * tpb + wave * TT + t) * 16 + r16;
            const int64_t trow = tok < N ? tok : N - 1;
            float raw32[XD == 2 ? NS32 : 1][8];              // fp32 rows: one token tile's pieces at a time (a round trip per tile)
            if constexpr (XD == 2) {
#pragma unroll
                for (int s = 0; s < NS32; ++s) load8<0>(dec.xrows, trow * Dr + 32 * s + 8 * q4, raw32[s]);
            }
            // x_prep_body's thread g = 4 (s & 1) + q4 owns the pieces of this lane with that parity: two running sums each.
            // The statistics are slice 0's business alone (wave-uniform branch): the other slices only convert.
            const bool stats = sl == 0;
            float sh[2] = {0.0f, 0.0f}, sr[2] = {0.0f, 0.0f}, pn[2][8];
#pragma unroll
            for (int j = 0; j < 8; ++j) { pn[0][j] = 0.0f; pn[1][j] = 0.0f; }
#pragma unroll
            for (int s = 0; s < NS32; ++s) {
                float v[8];
                half8 f;
                if constexpr (XD == 1) {
                    // bf16 -> fp16 two at a time (v_cvt_pk_f16_f32).  A bf16 value below 2^-14 converts to an fp16 subnormal or
                    // zero and never rounds up to 2^-14 (8 significant bits against 11), so flushing the INPUT where |v| < 2^-14
                    // is to_f16_ftz's flush of the result; everything else converts as (_Float16)v does (RNE, overflow to inf)
                    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
                    typedef _Float16 half2v __attribute__((ext_vector_type(2)));
                    typedef float float2v __attribute__((ext_vector_type(2)));
                    const u32x4 bits = __builtin_bit_cast(u32x4, xf[t][s]);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const uint32_t lo = bits[j] << 16, hi = bits[j] & 0xFFFF0000u;
                        v[2 * j] = __uint_as_float(lo); v[2 * j + 1] = __uint_as_float(hi);
                        float2v c;
                        c[0] = (lo & 0x7FFFFFFFu) < 0x38800000u ? 0.0f : v[2 * j];
                        c[1] = (hi & 0x7FFFFFFFu) < 0x38800000u ? 0.0f : v[2 * j + 1];
                        const half2v h = __builtin_convertvector(c, half2v);
                        f[2 * j] = h[0]; f[2 * j + 1] = h[1];
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < 8; ++j) { v[j] = raw32[s][j]; f[j] = to_f16_ftz(v[j]); }
                }
                if (stats) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const float b = (float)f[j], res = v[j] - b;
                        sh[s & 1] = fmaf(b, b, sh[s & 1]); sr[s & 1] = fmaf(res, res, sr[s & 1]);
                        pn[s & 1][j] = fmaf(v[j], v[j], pn[s & 1][j]);
                    }
                }
                xf[t][s] = f;
            }
            if (!stats) continue;
            // |xh|^2, |x - xh|^2: the eight threads' sums in thread order (g = 0 .. 7: pieces q4 = 0..3 of even s, then of odd s)
            float a = 0.0f, b = 0.0f;
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                a += __shfl(sh[g >> 2], (g & 3) * 16 + r16, 64);
                b += __shfl(sr[g >> 2], (g & 3) * 16 + r16, 64);
            }
            // oracle-order |x|^2: partial chain 8 g + j lives in the lane with q4 = g & 3; halving tree 32, 16, 8 across lanes, 4, 2, 1 inside
            float q[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) q[j] = pn[0][j] + pn[1][j];                    // level 32
#pragma unroll
            for (int j = 0; j < 8; ++j) q[j] = q[j] + __shfl_xor(q[j], 32, 64);        // level 16
#pragma unroll
            for (int j = 0; j < 8; ++j) q[j] = q[j] + __shfl_xor(q[j], 16, 64);        // level 8
            float a0 = q[0] + q[4], a1 = q[1] + q[5], a2 = q[2] + q[6], a3 = q[3] + q[7];
            a0 = a0 + a2; a1 = a1 + a3;
            a0 = a0 + a1;
            if (sl == 0 && q4 == 0 && tok < N && wave * TT + t < tpb) { dec.xh2_w[tok] = a; dec.rho2_w[tok] = b; dec.xn_w[tok] = a0; }
        }
    }
    if constexpr (PIPE_H) {
        // One wave per SIMD owns 512 registers, 256 of them accumulation registers — which the MFMA reads as operands just as
        // well.  Left alone, hipcc runs out of the first 256, parks fragments in the second as SPILLS and copies each back
        // (v_accvgpr_mov) in front of the MFMA that uses it: 4 copies per chunk, and the lone wave became bound by its own
        // instruction stream (profiles/r03_large_d_experiments.txt).  Told where the fragments live — the upper half of the
        // token tiles in accumulation registers, the lower half in vector registers — it emits the loop without a copy and
        // without a spill (500 registers at D = 768 with 64 tokens per wave, 496 at D = 1024 with 48).
#pragma unroll
        for (int t = 0; t < TT; ++t)
#pragma unroll
            for (int s = 0; s < NS32; ++s) {
                if (t >= TT / 2) asm volatile("" : "+a"(xf[t][s]));
                else asm volatile("" : "+v"(xf[t][s]));
            }
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
#ifndef VQ_LAG_MIN_STAGES
#define VQ_LAG_MIN_STAGES 48     // slices shorter than this run without the lag (one iteration less)
#endif
    const int lag = (NBUF >= 4 && wave >= WAVES / 2 && st1 - st0 >= VQ_LAG_MIN_STAGES) ? 1 : 0;
    if (st0 < st1) issue_stage(st0, 0);
    if (AHEAD >= 2 && st0 + 1 < st1) issue_stage(st0 + 1, 1);
    VQ_PHASE(1);
    vq_dma_barrier();  // drains the LDS-DMA (vmcnt(0)) and makes it visible to every wave
    VQ_PHASE(2);

    f32x4 accA[2][TT], accB[2][TT];
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int t = 0; t < TT; ++t)
#pragma unroll
            for (int q = 0; q < 4; ++q)   // "previous tile" of the very first tile: never wins (group records: never even registers)
                accB[c][t][q] = GROUPS ? -INFINITY : -3.0e38f;
    if constexpr (PIPE_H) {
#pragma unroll
        for (int t = 0; t < TT; ++t) accA[1][t] = f32x4{-3.0e38f, -3.0e38f, -3.0e38f, -3.0e38f};
    }

#ifdef VQ_CLOCK_STAMPS
    // diagnostic build only (tools/inkernel_clock.py; MI355X guide, DVFS item 6): the shader clock this wave's stage loop ran at =
    // delta s_memtime / delta s_memrealtime x 100 MHz.  The stamps go to a buffer of their own that nothing else reads.
    const unsigned long long clk_t0 = __builtin_amdgcn_s_memtime(), clk_r0 = __builtin_amdgcn_s_memrealtime();
#endif
#ifdef VQ_STAGE_STAMPS
    unsigned long long *stamp_lds = (unsigned long long *)(lds + NBUF * STAGE_BYTES);
    for (int i = threadIdx.x; i < 8 * VQ_STAGE_ITERS * 5; i += WAVES * 64) stamp_lds[i] = 0;
#endif
    for (int64_t it = st0; it < st1 + ((NBUF >= 4 && st1 - st0 >= VQ_LAG_MIN_STAGES) ? 1 : 0); ++it) {
#ifdef VQ_STAGE_STAMPS
        const int stamp_it = (int)(it - st0);
#endif
        VQ_STAMP(0);
#ifndef VQ_EXP_NO_DMA            // (timing-only diagnostic build: the ring is never refilled inside the loop)
        if (it + AHEAD < st1) issue_stage(it + AHEAD, (int)((it + AHEAD - st0) % NBUF));
#endif
        VQ_STAMP(1);
        const int64_t st = it - lag;
        if (st < st0 || st >= st1 || !wave_active) { vq_dma_barrier(); continue; }
        const int buf = (int)((st - st0) % NBUF);
        const char *base = lds + buf * STAGE_BYTES;
        const char *aux = base + TPS * NSTEP * VQ_CHUNK_BYTES;
      // the tiles of one stage; WITH_AUX false: accumulators start from the constant 0 (no aux read)
      auto run_stage = [&](auto with_aux_tag) __attribute__((always_inline)) {
        constexpr bool WITH_AUX = decltype(with_aux_tag)::value;
        if constexpr (PIPE_H) {
            constexpr int PFH = VQ_PIPEH_PF;                  // A fragments this many chunks (TT MFMAs each) ahead
            constexpr int TOTALH = 4 * TT;                    // accumulator elements per lane and code half
            const uint32_t tile = (uint32_t)st;
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const f32x4 a4 = *(const f32x4 *)(aux + (16 * c + 4 * (lane >> 4)) * 4);
                uint32_t old[TT];
#pragma unroll
                for (int t = 0; t < TT; ++t) { accA[c][t] = a4; old[t] = __float_as_uint(b1[t]); }
                half8 af[PFH + 1];
#pragma unroll
                for (int i = 0; i < PFH; ++i) af[i] = *(const half8 *)(base + (2 * i + c) * VQ_CHUNK_BYTES + lane * 16);
#pragma unroll
                for (int s = 0; s < NS32; ++s) {
                    if (s + PFH < NS32)
                        af[(s + PFH) % (PFH + 1)] = *(const half8 *)(base + (2 * (s + PFH) + c) * VQ_CHUNK_BYTES + lane * 16);
#pragma unroll
                    for (int t = 0; t < TT; ++t)
                        accA[c][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[s % (PFH + 1)], xf[t][s], accA[c][t], 0, 0, 0);
#pragma unroll
                    for (int id = s * TOTALH / NS32; id < (s + 1) * TOTALH / NS32; ++id) {
                        const int t = id / 4, q = id % 4, e = 4 * (1 - c) + q;
                        float v = __uint_as_float((__float_as_uint(accA[1 - c][t][q]) & 0xFFFFFFF0u) | (uint32_t)e);
                        b2[t] = __builtin_amdgcn_fmed3f(b1[t], b2[t], v);
                        b1[t] = vmax(b1[t], v);
                    }
                    // the statements of a chunk stay together: under this register pressure hipcc otherwise folds the fragment
                    // ring into one buffer (ds_read -> s_waitcnt -> MFMAs) and hoists the epilogue out of the MFMA shadow
                    __builtin_amdgcn_sched_barrier(0);
                }
                const uint32_t tgp = tile - (c == 0 ? 1u : 0u);
#pragma unroll
                for (int t = 0; t < TT; ++t) t1[t] = (__float_as_uint(b1[t]) != old[t]) ? tgp : t1[t];
            }
        } else {
        // A fragments PF chunks ahead of the MFMAs that consume them (ring of PF+1 register sets), across the tiles of the stage
        // (STAGE_PF; off: every tile starts with its own first reads exposed); chunk ch = 2*s32 + c of a tile feeds the TT MFMAs
        // of code half c at k-step s32
#ifndef VQ_PF_TT2
#define VQ_PF_TT2 3
#endif
#ifndef VQ_PF_TT4
#define VQ_PF_TT4 2
#endif
#ifndef VQ_CHUNK_SB
#define VQ_CHUNK_SB 2          // 1: TT <= 2 forms only, 2: every unfiltered form of NSTEP 4..16, 0: off (A/B builds)
#endif
#ifndef VQ_STAGE_PF
#define VQ_STAGE_PF 1
#endif
        // (D = 512, NSTEP = 32: measured 2.5 % slower with the chunks held together and the deeper ring — left as it was)
        // (the filtered forms of D = 64 / 128 lose 1-3 % with it and D = 768 / 1024 gain nothing at any ring depth: profiles/r05_ab_stage_loop.txt)
        constexpr bool CHUNK_SB = NSTEP >= 4 && NSTEP <= 16 && !FILTER && (VQ_CHUNK_SB == 2 || (VQ_CHUNK_SB == 1 && TT <= 2));
        constexpr int PF = NSTEP <= 32 ? (CHUNK_SB ? (TT <= 2 ? VQ_PF_TT2 : VQ_PF_TT4) : 1) : (NSTEP <= 48 ? 2 : 4);   // deeper where a chunk feeds fewer MFMAs
        constexpr bool STAGE_PF = (VQ_STAGE_PF != 0) && CHUNK_SB && TPS >= 2;
        half8 af[PF + 1];
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
            // ring slot of chunk ch of tile ti: by its index in the stage (STAGE_PF) or in the tile
            constexpr int RING = PF + 1;
            const int g0 = STAGE_PF ? ti * NSTEP : 0;
            if (!STAGE_PF || ti == 0) {
#pragma unroll
                for (int i = 0; i < PF; ++i)
                    if (i < NSTEP) af[i] = *(const half8 *)(base + (ti * NSTEP + i) * VQ_CHUNK_BYTES + lane * 16);
            }
#pragma unroll
            for (int ch = 0; ch < NSTEP; ++ch) {
#ifdef VQ_EXP_NO_LDS_READS      // timing-only diagnostic build: the A fragment of every chunk is the register set loaded first (opaque to the compiler)
                if (ch + PF < NSTEP) { af[(ch + PF) % RING] = af[0]; asm volatile("" : "+v"(af[(ch + PF) % RING])); }
#else
                if (ch + PF < NSTEP || (STAGE_PF && ti + 1 < TPS))
                    af[(g0 + ch + PF) % RING] = *(const half8 *)(base + (ti * NSTEP + ch + PF) * VQ_CHUNK_BYTES + lane * 16);
#endif
#pragma unroll
                for (int t = 0; t < TT; ++t) {
                    if constexpr (!WITH_AUX) {
                        if (ch < 2) {                      // first k-step of this code half: C = inline constant 0
                            cur[ch & 1][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[(g0 + ch) % RING], xf[t][ch >> 1], f32x4{0.0f, 0.0f, 0.0f, 0.0f}, 0, 0, 0);
                            continue;
                        }
                    }
                    cur[ch & 1][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[(g0 + ch) % RING], xf[t][ch >> 1], cur[ch & 1][t], 0, 0, 0);
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
#ifdef VQ_EXP_NO_EPILOGUE       // timing-only diagnostic build: the scores are never looked at.  Every accumulator of the previous tile is
                constexpr bool RETIRE = false;      // named as an asm input once per tile, so that not one MFMA is dead code (a first form that
                if (ch == 0) {                      // read ONE element per chunk let hipcc drop a quarter of the MFMAs: 96 of 128 per stage)
#pragma unroll
                    for (int u = 0; u < TT; ++u) asm volatile("" :: "v"(prv[0][u]), "v"(prv[1][u]));
                }
#else
                constexpr bool RETIRE = PIPE && !FILTER;
#endif
#pragma unroll
                for (int i = 0; RETIRE && i < (TOTAL + NSTEP - 1) / NSTEP; ++i) {
                    constexpr int EVERY = (NSTEP / TOTAL) > 0 ? NSTEP / TOTAL : 1;   // TOTAL < NSTEP: one element every EVERY chunks
                    const int id = (TOTAL >= NSTEP) ? ch * (TOTAL / NSTEP) + i : ((ch % EVERY == 0) ? ch / EVERY : -1);
                    if (id >= 0 && id < TOTAL) {
                        const int t = id / NE, e = id % NE;
                        float v = __uint_as_float((__float_as_uint(prv[e >> 2][t][e & 3]) & 0xFFFFFFF0u) | (uint32_t)e);
                        b2[t] = __builtin_amdgcn_fmed3f(b1[t], b2[t], v);
                        b1[t] = vmax(b1[t], v);
                    }
                }
                // the statements of a chunk stay together: left alone, hipcc sinks the fragment read of chunk ch + PF down to its
                // use (ds_read -> s_waitcnt lgkmcnt(0) -> MFMAs: the LDS latency of every chunk exposed, covered only by the
                // SIMD's other wave)
                if constexpr (CHUNK_SB) __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (!PIPE && !PIPE_H) {   // large D: one tile per stage, epilogue in place (the SIMD's other wave covers it)
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
        }
      };   // run_stage
#ifdef VQ_EXP_NO_COMPUTE         // timing-only diagnostic build: the ring, its requests and the barriers alone
        asm volatile("" :: "v"(xf[0][0]));
#else
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
        // next stage landed (vmcnt(0)) and everybody is done reading this one.  (A barrier that keeps the pieces of the
        // stage requested in this iteration in flight — s_waitcnt vmcnt(pieces) instead of 0 — was measured: 1-3 %
        // slower at D <= 128 and 2x slower at D = 256, profiles/r02_ring_partial_wait.txt; the full drain stays.)
#ifdef VQ_STAGE_STAMPS
        VQ_STAMP(2);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        VQ_STAMP(3);
        __syncthreads();
        VQ_STAMP(4);
#elif defined(VQ_EXP_NO_BARRIER)    // timing-only diagnostic build
        asm volatile("" ::: "memory");
#else
        vq_dma_barrier();
#endif
    }
#ifdef VQ_STAGE_STAMPS
    __syncthreads();
    if (blockIdx.x < VQ_STAGE_WGS)
        for (int i = threadIdx.x; i < 8 * VQ_STAGE_ITERS * 5; i += WAVES * 64) vq_stage_dbg[(size_t)blockIdx.x * 8 * VQ_STAGE_ITERS * 5 + i] = stamp_lds[i];
    __syncthreads();
#endif
#ifdef VQ_CLOCK_STAMPS
    {
        const unsigned long long clk_t1 = __builtin_amdgcn_s_memtime(), clk_r1 = __builtin_amdgcn_s_memrealtime();
        if (lane == 0 && wave_active) {
            const unsigned slot = ((unsigned)blockIdx.x * WAVES + (unsigned)wave) & (VQ_CLOCK_SLOTS - 1);
            vq_clock_dbg[2 * slot] = clk_t1 - clk_t0;
            vq_clock_dbg[2 * slot + 1] = clk_r1 - clk_r0;
        }
    }
#endif
    VQ_PHASE(3);
    if (PIPE_H && st1 > st0) {   // drain: second half of the last tile
#pragma unroll
        for (int t = 0; t < TT; ++t) {
            const uint32_t old = __float_as_uint(b1[t]);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float v = __uint_as_float((__float_as_uint(accA[1][t][q]) & 0xFFFFFFF0u) | (uint32_t)(4 + q));
                b2[t] = __builtin_amdgcn_fmed3f(b1[t], b2[t], v);
                b1[t] = vmax(b1[t], v);
            }
            t1[t] = (__float_as_uint(b1[t]) != old) ? (uint32_t)(st1 * TPS - 1) : t1[t];
        }
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
            if (VQ_TICKET_NOFENCE && dec.idx != nullptr && nslices > 1) {
                // read by ANOTHER workgroup of this launch (the one that draws the block's last ticket): agent-scope stores, which go
                // past this XCD's L2 on their own — no write-back of the whole L2 (the release fence) in front of the ticket
                __hip_atomic_store(rp, r.v1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(rp + Np, __uint_as_float(r.c1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(rp + 2 * Np, r.v2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(rp + 3 * Np, __uint_as_float(r.c2), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(rp + 4 * Np, r.v3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                rp[0] = r.v1; rp[Np] = __uint_as_float(r.c1); rp[2 * Np] = r.v2;
                rp[3 * Np] = __uint_as_float(r.c2); rp[4 * Np] = r.v3;
            }
        }
    }

    // ---- decision stage, by the workgroup that completes a token block (dec.idx == nullptr: left to refine_decide_kernel)
    // Arrival counter per token block (MI355X guide, Guideline 16): every wave drains its record stores, the workgroup
    // meets, one lane releases at agent scope and takes a ticket; the workgroup that draws the last ticket of the block
    // (nslices of them) acquires and merges the records of all slices — one launch less on the critical path.
    if (dec.idx != nullptr) {
        int *flags = (int *)lds;                         // the stage ring is free now (first barrier below)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        VQ_PHASE(4);
        if (threadIdx.x == 0) {
            int last = 1;
            if (nslices > 1) {
                // (every wave drained its record stores before the barrier above; the records are read back with agent-scope loads)
                if (!VQ_TICKET_NOFENCE) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
                last = (atomicAdd(&dec.arrive[tb], 1) == nslices - 1) ? 1 : 0;
                if (last && !VQ_TICKET_NOFENCE) {
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
            }
            flags[0] = last;
        }
        __syncthreads();
        const bool last = flags[0] != 0;
        __syncthreads();                                 // everybody has read the flag before the LDS words are reused
        VQ_PHASE(5);
#ifdef VQ_PHASE_STAMPS
        if (threadIdx.x == 0 && blockIdx.x < VQ_PHASE_SLOTS) vq_phase_dbg[8 * blockIdx.x + 7] = last ? 1 : 0;
#endif
        if (last) {
            int *wcount = (int *)lds, *wbase = wcount + 3 * 16;
            int64_t n = tb * (int64_t)(tpb * 16) + threadIdx.x;      // tpb*16 <= BM <= WAVES*64 threads: one token per thread
            const bool oob = (int)threadIdx.x >= tpb * 16 || n >= N;
            if (n >= N) n = N - 1;
            decide_rows<true>(n, oob, cbst, Dp, metric, nslices, rec, xh2, rho2, Np, dec, wcount, wbase);
        }
    }
    VQ_PHASE(6);
  }
}
