// libvqhip device kernels, unit 3b: the proposal pass for D <= 16 on v_mfma_f32_32x32x16_f16.  Included by vqhip_kernels.h.
//
// Why a second form.  At D <= 32 the proposal pass is bound by the vector instructions that look at the scores and by MFMA
// ISSUE, not by the matrix pipe (profiles/r03_d32_epilogue_shares.txt).  With D <= 16 one 32x32x16 instruction
// covers the whole inner dimension and 32 codes x 32 tokens = 1024 scores, where the 16x16x32 form needs four instructions
// (half of whose k-slots multiply padding): a quarter of the MFMA issue (8 cycles of the SIMD's vector port each), and a lane
// holds 16 scores of ONE token per instruction, so the group record (coarse_kernel, GROUPS) is updated per 16 scores instead
// of per 8 while the group stays ONE code tile — the identification replay does not grow (what sank the pair groups).
//
// Same images, same pipeline.  Nothing outside this kernel changes: the operands are read from the EXISTING fragment-major
// images (a 16-row x 32-dim chunk holds, for lane l', row (l' & 15), dims 8 (l' >> 4) .. +8; dims 16..31 are padding at
// D <= 16) with a per-lane address — lane l of the 32x32x16 operand wants row (l & 31) and dims 8 (l >> 5) .. +8, i.e. the
// 16-byte piece (l >> 5) * 16 + (l & 15) of chunk (l >> 4) & 1 — so codebook image, token image, LDS-DMA ring, records,
// decision stage, second pass and re-rank are the ones of the 16x16x32 form.  Scores: se * (xh . eh_k) - se |e_k|^2 / 2 with the
// accumulator initialised from the aux values, accumulated over 16 products instead of 32 (16 of them zeros) — within the
// same error bound B (row_margin is computed for the padded dimension 32); the replay uses this kernel's own instruction,
// hence the very scores of the stream; the second pass (16x16x32) re-derives scores under the same bound, which is all the
// margin argument needs of it.
#pragma once

// maximum of the 16 accumulator elements of one (token tile, code tile) in eight instructions, ordered behind the fake inputs
// after[] (results of the MFMAs of the CURRENT tile: >= TT 32-cycle MFMAs after the producers; see tile_max8)
template <int TT>
__device__ __forceinline__ void tile_max16(float &dst, float &sc1, float &sc2, const f32x16 &p, const float (&after)[TT]) {
    const float a0 = after[0], a1 = after[TT > 1 ? 1 : 0];
    asm("v_max3_f32 %0, %3, %4, %5\n\t"
        "v_max3_f32 %1, %6, %7, %8\n\t"
        "v_max3_f32 %2, %9, %10, %11\n\t"
        "v_max3_f32 %0, %0, %12, %13\n\t"
        "v_max3_f32 %1, %1, %14, %15\n\t"
        "v_max3_f32 %2, %2, %16, %17\n\t"
        "v_max3_f32 %0, %0, %1, %2\n\t"
        "v_max_f32 %0, %0, %18"
        : "+v"(dst), "+v"(sc1), "+v"(sc2)
        : "v"(p[0]), "v"(p[1]), "v"(p[2]), "v"(p[3]), "v"(p[4]), "v"(p[5]), "v"(p[6]), "v"(p[7]), "v"(p[8]), "v"(p[9]),
          "v"(p[10]), "v"(p[11]), "v"(p[12]), "v"(p[13]), "v"(p[14]), "v"(p[15]), "v"(a0), "v"(a1));
}
// the same folded into a running maximum `run` (the group so far): the seventeenth input rides in the eighth instruction
template <int TT>
__device__ __forceinline__ void tile_max17(float &run, float &sc0, float &sc1, float &sc2, const f32x16 &p, const float (&after)[TT]) {
    const float a0 = after[0], a1 = after[TT > 1 ? 1 : 0];
    asm("v_max3_f32 %1, %4, %5, %6\n\t"
        "v_max3_f32 %2, %7, %8, %9\n\t"
        "v_max3_f32 %3, %10, %11, %12\n\t"
        "v_max3_f32 %1, %1, %13, %14\n\t"
        "v_max3_f32 %2, %2, %15, %16\n\t"
        "v_max3_f32 %3, %3, %17, %18\n\t"
        "v_max3_f32 %1, %1, %2, %3\n\t"
        "v_max3_f32 %0, %0, %1, %19"
        : "+v"(run), "+v"(sc0), "+v"(sc1), "+v"(sc2)
        : "v"(p[0]), "v"(p[1]), "v"(p[2]), "v"(p[3]), "v"(p[4]), "v"(p[5]), "v"(p[6]), "v"(p[7]), "v"(p[8]), "v"(p[9]),
          "v"(p[10]), "v"(p[11]), "v"(p[12]), "v"(p[13]), "v"(p[14]), "v"(p[15]), "v"(a0), "v"(a1));
}
// max over the two lanes l, l ^ 32 that share a token in the 32x32 MFMA output
__device__ __forceinline__ float pair_rows_max(float v) {
    auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return vmax(__uint_as_float(b[0]), __uint_as_float(b[1]));
}

// where the identification requests of the group records go (workspace: vq_ws_layout): one counter and a list of `cap`
// entries (token | lane half << 30 | record slot << 31) per bucket, the token's B fragment beside it;
// rece2[slice][slot][token] receives the runner-up inside an identified group; idmask: the low mantissa bits that carry a
// group id in the stream kernel's running records (2^bits >= groups)
// (bucket = tile * R + token block % R: vqhip_layout.h; counter of bucket b at bcnt[b * VQ_GROUP_CNT_STRIDE])
struct VqGroupLists { int *bcnt; uint32_t *blist; char *bfrag; float *rece2; int cap, R; uint32_t idmask; };

// One workgroup = WAVES waves x TT wide token tiles of 32 tokens (B fragments in registers for the whole kernel); the codebook
// image streams through the LDS ring exactly as in coarse_kernel (NBUF stages of TPS tiles, filled two ahead, the second half
// of the waves one stage behind).  tpb: 16-token tiles per workgroup (even).  Group records as in coarse_kernel<..., GROUPS>,
// straight-line update (no skip test: TT <= 2); the identification of a lane's best group is a request to identify32_kernel.
// KS: k-steps of 16 dims — 1 for D <= 16, 2 for D <= 32 (two instructions accumulate; the chains of the TT token tiles are
// interleaved so that no instruction waits for the one it accumulates onto).
// GT: code tiles per group record (a divisor of TPS).  A lane keeps the running maximum of its scores over the GT tiles of a
// group (8 instructions per tile: the tree; the running maximum is the tree's seventeenth input) and runs the 4-instruction
// group update once per GROUP: 8 + 4 / GT vector instructions per 16 scores.  What a group costs elsewhere is GT tiles per
// request in identify32_kernel — GT MFMAs on 32 requests at a time — and nothing in second-pass rows: a lane identifies ONE
// candidate whatever the group size, everything else it has seen is a bound either way.
#ifndef VQ_W32_PF
#define VQ_W32_PF 1             // A fragments this many code tiles ahead of their MFMAs (0: the round-5 loop, read at use)
#endif
template <int TT, int WAVES, int TPS, int NBUF, bool NOAUX, int KS = 1, int GT = 1>
#ifndef VQ_W32_TT2_OCC
#define VQ_W32_TT2_OCC 4        // waves per SIMD the two-wide-tile form (>= 262 144 rows) is compiled for (A/B: 3)
#endif
__global__ __launch_bounds__(WAVES * 64, TT >= 2 ? VQ_W32_TT2_OCC : 4) void coarse32_kernel(
    const char *__restrict__ ximg, int64_t N, const char *__restrict__ frag, int64_t nstages, int nslices,
    float *__restrict__ rec, int64_t Np, const VqCbStats *__restrict__ cbst, const float *__restrict__ xh2,
    const float *__restrict__ rho2, int Dp, int metric, const int *__restrict__ n_dev, VqGroupLists grp, int pad_stage, int tpb) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    static_assert(TPS % 2 == 0 && NBUF >= 4, "ring of four stages, accumulators ping-pong by tile parity");
    static_assert(TPS % GT == 0, "a group lies inside one stage");
    constexpr int NSTEP = 2;                              // the images are those of the padded dimension 32
    constexpr int NCH = TPS * NSTEP + VQ_AUX_CHUNKS(TPS);
    constexpr int STAGE_BYTES = NCH * VQ_CHUNK_BYTES;
    constexpr int NE = 16;                                // accumulator elements per lane, token tile and code tile
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int half = lane >> 5;                           // dims 8 half .. +8 of the operands; output rows + 4 half
    const int col = lane & 31;                            // token of the wide tile (B, output column) / code of the tile (A)
    const int sub = (lane >> 4) & 1;                      // which 16-row chunk holds this lane's row
    const int piece = (half * 16 + (lane & 15)) * 16;     // byte offset of this lane's 16-byte piece inside that chunk
    const int64_t ntt = (N + 31) / 32 * 2;                // 16-token tiles in the fp16 token image (of the launch's capacity)
    if (n_dev != nullptr) {                               // device-side row count: token blocks past it have nothing to do
        const int64_t nd = *n_dev;
        N = nd < N ? nd : N;
        if ((int64_t)(blockIdx.x / nslices) * tpb * 16 >= N) return;
    }
    VQ_PHASE(0);
    cb_stats_publish(cbst);
    const bool wave_active = wave * TT * 2 < tpb;
    const int sl = blockIdx.x % nslices;
    const int64_t tb = blockIdx.x / nslices;
    const int64_t st0 = (nstages * sl) / nslices, st1 = (nstages * (sl + 1)) / nslices;

    // ---- prologue: the wide token tiles' B fragments, gathered from the 16-token chunks of the token image ----
    half8 xf[TT][KS];                                     // (k-step s: the pieces 32 further on, i.e. 512 bytes)
#pragma unroll
    for (int t = 0; t < TT; ++t) {
        int64_t t16 = tb * tpb + (wave * TT + t) * 2 + sub;
        t16 = t16 < ntt ? t16 : ntt - 1;                  // out-of-range tiles read a valid tile and are never written
#pragma unroll
        for (int s = 0; s < KS; ++s) xf[t][s] = *(const half8 *)(ximg + t16 * (int64_t)VQ_CHUNK_BYTES + piece + s * 512);
    }
    auto issue_stage = [&](int64_t st, int buf) {
        const char *src = frag + st * (int64_t)STAGE_BYTES;
        char *dstb = lds + buf * STAGE_BYTES;
        for (int c = wave; c < NCH; c += WAVES)
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void *)(src + c * VQ_CHUNK_BYTES + lane * 16),
                (__attribute__((address_space(3))) void *)(dstb + c * VQ_CHUNK_BYTES), 16, 0, 0);
    };
    constexpr int AHEAD = 2;
    const int lag = (wave >= WAVES / 2) ? 1 : 0;
    if (st0 < st1) issue_stage(st0, 0);
    if (st0 + 1 < st1) issue_stage(st0 + 1, 1);
    // (the first two stages are on their way before the statistics are folded and the margins computed: their round trips overlap)
    float b1[TT], b2[TT], b3[TT], mg[TT];                 // best / second / third group maximum, group id in the low bits
    float sc0 = 0.0f, sc1 = 0.0f, sc2 = 0.0f;             // destinations of the asm maxima: live across the whole loop
    float gm[TT];                                         // running maximum of the group being streamed
#pragma unroll
    for (int t = 0; t < TT; ++t) gm[t] = -INFINITY;
    bool const_norm;                                      // L2 on a constant-norm codebook: aux values of real codes are 0
    {
        const VqCbStats stv = cb_stats_view(cbst);
        const_norm = __builtin_amdgcn_readfirstlane((int)stv.l2_const_norm) != 0;
#pragma unroll
        for (int t = 0; t < TT; ++t) {
            b1[t] = -INFINITY; b2[t] = -INFINITY; b3[t] = -INFINITY;
            int64_t tokn = (tb * tpb + (wave * TT + t) * 2) * 16 + col;
            tokn = tokn < N ? tokn : N - 1;
            const float m = row_margin(&stv, Dp, metric, xh2[tokn], rho2[tokn]);
            mg[t] = (m > 0.0f) ? m : INFINITY;            // no usable bound: nothing is identified, the decision stage re-scans
        }
    }

    VQ_PHASE(1);
    vq_dma_barrier();  // drains the LDS-DMA (vmcnt(0)) and makes it visible to every wave
    VQ_PHASE(2);
#ifdef VQ_PHASE_STAMPS
    const unsigned long long phase_clk0 = __builtin_amdgcn_s_memtime();
#endif

    f32x16 accA[TT], accB[TT];
#pragma unroll
    for (int t = 0; t < TT; ++t)
#pragma unroll
        for (int q = 0; q < NE; ++q) accB[t][q] = -3.0e38f;    // "previous tile" of the very first tile: its group id lies outside
                                                               // the slice, so it never becomes a request (finite: the id bits
                                                               // would turn -inf into a signalling NaN)

#ifdef VQ_STAGE_STAMPS            // diagnostic build only (tools/stage_stamps.py): the five stamps of coarse_kernel's loop, same buffer
    unsigned long long *stamp_lds = (unsigned long long *)(lds + NBUF * STAGE_BYTES);
    for (int i = threadIdx.x; i < 8 * VQ_STAGE_ITERS * 5; i += WAVES * 64) stamp_lds[i] = 0;
    __syncthreads();
#endif
    for (int64_t it = st0; it < st1 + 1; ++it) {
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
        auto run_stage = [&](auto with_aux_tag) __attribute__((always_inline)) {
            constexpr bool WITH_AUX = decltype(with_aux_tag)::value;
            // A fragments PF tiles ahead of their MFMAs, through a ring of PF + 1 register sets; the statements of a tile stay
            // together (sched_barrier): left alone, hipcc sinks every fragment read down to its use — ds_read, s_waitcnt
            // lgkmcnt(0), MFMA: the LDS latency exposed once per code tile and wave (the shipped round-5 loop:
            // profiles/r06_c3_stage_stamps.txt)
            constexpr int PF = VQ_W32_PF, RING = PF + 1;
            half8 afr[RING][KS];
            auto load_af = [&](int tile, half8 (&dst)[KS]) __attribute__((always_inline)) {
#ifdef VQ_EXP_NO_LDS_READS      // timing-only diagnostic build: every tile multiplies the fragments of the stage's first tile
                if (tile != 0) {
#pragma unroll
                    for (int s = 0; s < KS; ++s) { dst[s] = afr[0][s]; asm volatile("" : "+v"(dst[s])); }
                    return;
                }
#endif
#pragma unroll
                for (int s = 0; s < KS; ++s) dst[s] = *(const half8 *)(base + (tile * NSTEP + sub) * VQ_CHUNK_BYTES + piece + s * 512);
            };
#pragma unroll
            for (int i = 0; i < PF; ++i) load_af(i, afr[i % RING]);
#pragma unroll
            for (int ti = 0; ti < TPS; ++ti) {
                f32x16 (&cur)[TT] = (ti & 1) ? accB : accA;
                f32x16 (&prv)[TT] = (ti & 1) ? accA : accB;
                if (ti + PF < TPS) load_af(ti + PF, afr[(ti + PF) % RING]);
                half8 (&af)[KS] = afr[ti % RING];
                f32x16 init;
                if constexpr (WITH_AUX) {                 // -se |e|^2 / 2 of this lane's code rows (r & 3) + 8 (r >> 2) + 4 half
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const f32x4 a4 = *(const f32x4 *)(aux + (ti * 32 + 8 * g + 4 * half) * 4);
#pragma unroll
                        for (int q = 0; q < 4; ++q) init[4 * g + q] = a4[q];
                    }
                } else {
#pragma unroll
                    for (int q = 0; q < NE; ++q) init[q] = 0.0f;
                }
                // the tile whose scores are retired below is tile st * TPS + ti - 1: number (ti - 1) mod GT of group gid
                const int pg = (ti + GT - 1) % GT;            // (a constant once the tile loop is unrolled)
                const uint32_t gid = (uint32_t)(st * (TPS / GT)) + (uint32_t)((ti + GT - 1) / GT) - 1u;
#ifdef VQ_EXP_NO_EPILOGUE       // timing-only diagnostic build: the scores are never looked at; every MFMA stays live because the
                // accumulators chain from tile to tile (prv -> cur) and the drain below reads the last ones
#pragma unroll
                for (int t = 0; t < TT; ++t) {
                    cur[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[0], xf[t][0], prv[t], 0, 0, 0);
#pragma unroll
                    for (int s = 1; s < KS; ++s) cur[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[s], xf[t][s], cur[t], 0, 0, 0);
                }
                (void)init; (void)pg; (void)gid;
#else
#pragma unroll
                for (int s = 0; s + 1 < KS; ++s)          // all but the last k-step, token tiles interleaved
#pragma unroll
                    for (int t = 0; t < TT; ++t)
                        cur[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[s], xf[t][s], s == 0 ? init : cur[t], 0, 0, 0);
#pragma unroll
                for (int t = 0; t < TT; ++t) {
                    cur[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[KS - 1], xf[t][KS - 1], KS == 1 ? init : cur[t], 0, 0, 0);
                    // token tile t of the PREVIOUS code tile: its maximum, then the group record (the lane keeps the best
                    // group maximum — its 16 codes of one code tile — the tile it came from and the best maximum of any
                    // other group; which of the 16 codes it was is found after the stream by replaying that one tile)
                    float after[TT];
#pragma unroll
                    for (int u = 0; u < TT; ++u) after[u] = cur[u <= t ? u : t][0];
                    if (pg == 0) tile_max16<TT>(gm[t], sc1, sc2, prv[t], after);
                    else tile_max17<TT>(gm[t], sc0, sc1, sc2, prv[t], after);
                    if (pg == GT - 1) {                    // the group is complete: its maximum with the group id in the low
                        // mantissa bits joins the lane's three best (one v_and_or, two v_med3, one v_max: the ids ride along)
                        // (the "group" in front of the slice's first one has id -1: masked, it holds -3e38 and never matters)
                        const float h = __uint_as_float((__float_as_uint(gm[t]) & ~grp.idmask) | (gid & grp.idmask));
                        b3[t] = __builtin_amdgcn_fmed3f(b2[t], b3[t], h);
                        b2[t] = __builtin_amdgcn_fmed3f(b1[t], b2[t], h);
                        b1[t] = vmax(b1[t], h);
                    }
                }
#endif
                if constexpr (PF > 0) __builtin_amdgcn_sched_barrier(0);
            }
        };
        if constexpr (NOAUX) {
            if (st == (int64_t)pad_stage) run_stage(std::true_type{}); else run_stage(std::false_type{});
        } else {       // the aux reads (four 16-byte LDS reads per lane and tile) are skipped where every value is zero anyway
            if (const_norm && st != (int64_t)pad_stage) run_stage(std::false_type{}); else run_stage(std::true_type{});
        }
#ifdef VQ_STAGE_STAMPS
        VQ_STAMP(2);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        VQ_STAMP(3);
        __syncthreads();
        VQ_STAMP(4);
#else
        vq_dma_barrier();  // next stage landed (vmcnt(0)) and everybody is done reading this one
#endif
    }
#ifdef VQ_STAGE_STAMPS
    __syncthreads();
    if (blockIdx.x < VQ_STAGE_WGS)
        for (int i = threadIdx.x; i < 8 * VQ_STAGE_ITERS * 5; i += WAVES * 64) vq_stage_dbg[(size_t)blockIdx.x * 8 * VQ_STAGE_ITERS * 5 + i] = stamp_lds[i];
    __syncthreads();
#endif
    VQ_PHASE(3);
#ifdef VQ_PHASE_STAMPS           // slot 7: shader cycles of this workgroup's stream loop (tools/phase_stamps.py: clock = cycles / (stamp 3 - stamp 2))
    if (threadIdx.x == 0 && blockIdx.x < VQ_PHASE_SLOTS) {
        vq_phase_dbg[8 * blockIdx.x + 7] = __builtin_amdgcn_s_memtime() - phase_clk0;
        // slot 5: where the workgroup ran — HW_REG_XCC_ID (id 20) << 32 | HW_REG_HW_ID (id 4: wave, simd, pipe, cu, sh, se)
        const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);
        vq_phase_dbg[8 * blockIdx.x + 5] = ((unsigned long long)xcc << 32) | hw;
    }
#endif
    // drain: the last tile (odd parity: it sits in accB)
    if (st1 > st0) {
#pragma unroll
        for (int t = 0; t < TT; ++t) {
            float g = accB[t][0];
#pragma unroll
            for (int e = 1; e < NE; ++e) g = __builtin_amdgcn_fmed3f(g, accB[t][e], INFINITY);   // max, NaN-transparent like v_max
            if constexpr (GT > 1) g = __builtin_amdgcn_fmed3f(g, gm[t], INFINITY);               // the last tile closes its group
            const float h = __uint_as_float((__float_as_uint(g) & ~grp.idmask) | (uint32_t)(st1 * (TPS / GT) - 1));
            b3[t] = __builtin_amdgcn_fmed3f(b2[t], b3[t], h);
            b2[t] = __builtin_amdgcn_fmed3f(b1[t], b2[t], h);
            b1[t] = vmax(b1[t], h);
        }
    }

    // ---- group records -> identification requests.  A lane's best group — and its second-best one — can matter when its
    // maximum lies within the row's margin of the best any lane of the token holds; the lane then files a request (token,
    // lane half, record slot) under that group, and identify32_kernel — 32 requests per MFMA, same fragments, same
    // instruction, same initial value: the very scores of the stream — writes the identified candidate into the record.
    // A record has two candidate slots per (token, slice): a lane's best group takes the slot of its half, its second-best
    // group the partner's slot when the partner files nothing.  Everything else the lane has seen stays a value bound, raised
    // to the largest value a score whose low bits were replaced (group id here, element id in the records) can have had.
    // (Round 3 replayed the best tile inside this kernel, one useful column per MFMA and ~55 vector instructions per token at
    // the end of every wave: 30 % of the kernel at BASELINE configs[2], profiles/r04_c3_shares.txt.)
    const uint32_t idm = grp.idmask;
    auto up = [idm](float v) {          // >= every score whose id-bit form is v
        const uint32_t b = __float_as_uint(v);
        return __uint_as_float((b & 0x80000000u) ? (b & ~idm) : (b | idm));
    };
    auto down = [idm](float v) {        // <= every such score
        const uint32_t b = __float_as_uint(v);
        return __uint_as_float((b & 0x80000000u) ? (b | idm) : (b & ~idm));
    };
    const uint32_t g0 = (uint32_t)(st0 * (TPS / GT)), g1 = (uint32_t)(st1 * (TPS / GT));
#pragma unroll
    for (int t = 0; t < TT; ++t) {
        const float top = pair_rows_max(down(b1[t]));           // a lower bound on the best score of the token in this slice
        const int64_t tokn = (tb * tpb + (wave * TT + t) * 2) * 16 + col;
        const bool live = tokn < N && (wave * TT + t) * 2 < tpb;
        const uint32_t id1 = __float_as_uint(b1[t]) & idm, id2 = __float_as_uint(b2[t]) & idm;
        const bool usable = live && (mg[t] < INFINITY) && (b1[t] > -INFINITY);
        bool need1 = usable && !(up(b1[t]) < top - mg[t]) && id1 >= g0 && id1 < g1;
        const bool pneed1 = __shfl_xor((int)need1, 32, 64) != 0;
        bool need2 = need1 && !pneed1 && (b2[t] > -INFINITY) && !(up(b2[t]) < top - mg[t]) && id2 >= g0 && id2 < g1;
        int bucket[2] = {0, 0}, pos[2] = {-1, -1};
        const uint32_t word = (uint32_t)tokn | ((uint32_t)half << 30);
        if (need1) {
            bucket[0] = (int)id1 * grp.R + (int)(tb & (grp.R - 1));
            pos[0] = atomicAdd(&grp.bcnt[(int64_t)bucket[0] * VQ_GROUP_CNT_STRIDE], 1);
        }
        if (need2) {
            bucket[1] = (int)id2 * grp.R + (int)(tb & (grp.R - 1));
            pos[1] = atomicAdd(&grp.bcnt[(int64_t)bucket[1] * VQ_GROUP_CNT_STRIDE], 1);
        }
        // a full list: the group stays a bound (the row takes the second pass)
        if (need1) { if (pos[0] < grp.cap) grp.blist[(int64_t)bucket[0] * grp.cap + pos[0]] = word | ((uint32_t)half << 31); else { need1 = false; pos[0] = -1; } }
        if (need2) { if (pos[1] < grp.cap) grp.blist[(int64_t)bucket[1] * grp.cap + pos[1]] = word | ((uint32_t)(1 - half) << 31); else { need2 = false; pos[1] = -1; } }
#pragma unroll
        for (int r = 0; r < 2; ++r) {   // the token's B fragment travels with the request: this lane's dims into its own entries and its partner's
            const int ppos = __shfl_xor(pos[r], 32, 64), pbucket = __shfl_xor(bucket[r], 32, 64);
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                if (pos[r] >= 0) *(half8 *)(grp.bfrag + ((((int64_t)bucket[r] * grp.cap + pos[r]) * 2 + half) * KS + s) * 16) = xf[t][s];
                if (ppos >= 0) *(half8 *)(grp.bfrag + ((((int64_t)pbucket * grp.cap + ppos) * 2 + half) * KS + s) * 16) = xf[t][s];
            }
        }
        float bound = up(b3[t]);
        if (!need1) bound = fmaxf(bound, up(b1[t]));
        if (!need2) bound = fmaxf(bound, up(b2[t]));
        bound = fmaxf(bound, __shfl_xor(bound, 32, 64));        // the two lanes that share the token
        if (lane < 32 && live) {
            float *rp = rec + (int64_t)sl * VQ_REC_FIELDS * Np + tokn;
            rp[0] = -INFINITY; rp[Np] = __uint_as_float(0xFFFFFFFFu); rp[2 * Np] = -INFINITY;
            rp[3 * Np] = __uint_as_float(0xFFFFFFFFu); rp[4 * Np] = bound;
            grp.rece2[(int64_t)(sl * 2) * Np + tokn] = -INFINITY;
            grp.rece2[(int64_t)(sl * 2 + 1) * Np + tokn] = -INFINITY;
        }
    }
#ifdef VQ_PHASE_STAMPS
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    VQ_PHASE(4);
    VQ_PHASE(6);
#endif
}

// Serves the identification requests of coarse32_kernel: one wave per bucket (group of GT code tiles x replica), four batches
// of 32 requests per trip — the group's A fragments are loaded once, the entries and the B fragments of the four batches (they
// travel with the requests: contiguous bytes) together: two dependent round trips per 128 requests.  Per batch and tile the
// wave runs the stream's own instruction (same fragments, same initial value: the very scores of the stream) and the
// per-element update (element id in the 4 low mantissa bits, runner-up) on the 16 scores each lane holds, folding the GT
// tiles' (best, runner-up) pairs; the lane that holds the requested half writes
//   rec[slice][2 slot .. 2 slot + 1][token] = (best score of the group, its code)      rece2[slice][slot][token] = runner-up.
template <int KS, int GT>
__global__ __launch_bounds__(256, 3) void identify32_kernel(const char *__restrict__ frag, int64_t nstages,
                                                            int nslices, int nbuckets, float *__restrict__ rec,
                                                            int64_t Np, const VqCbStats *__restrict__ cbst, VqGroupLists grp,
                                                            int pad_stage, int noaux) {
    constexpr int TPS = VQ_TPS_D32, NSTEP = 2, NE = 16, G = 4;
    constexpr int STAGE_BYTES = (TPS * NSTEP + VQ_AUX_CHUNKS(TPS)) * VQ_CHUNK_BYTES;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int bucket = (int)blockIdx.x * 4 + wave;
    if (bucket >= nbuckets) return;
    int cnt = grp.bcnt[(int64_t)bucket * VQ_GROUP_CNT_STRIDE];
    cnt = cnt < grp.cap ? cnt : grp.cap;
    if (cnt <= 0) return;
    const int half = lane >> 5, col = lane & 31, sub = (lane >> 4) & 1;
    const int piece = (half * 16 + (lane & 15)) * 16;
    const int gid = bucket / grp.R;
    const int64_t stage = (int64_t)gid * GT / TPS;
    const int ti0 = (gid * GT) % TPS;
    const int sl = (int)(((stage + 1) * nslices - 1) / nstages);
    const char *sb = frag + stage * (int64_t)STAGE_BYTES;
    const bool zero_init = (noaux || cbst->l2_const_norm != 0) && stage != (int64_t)pad_stage;
    const uint32_t *list = grp.blist + (int64_t)bucket * grp.cap;
    const char *bfrag = grp.bfrag + (int64_t)bucket * grp.cap * (2 * KS * 16);
    for (int base = 0; base < cnt; base += 32 * G) {
        // one trip: the entries and B fragments of up to G batches, then tile by tile (the tile's A fragments are fetched one
        // tile ahead) every batch's MFMA and per-element update, folded into the batch's (best, runner-up, tile) so far
        uint32_t entry[G];
        half8 xf[G][KS];
#pragma unroll
        for (int b = 0; b < G; ++b) {
            const int i = base + 32 * b + col;
            const int ic = i < cnt ? i : base;             // (padding columns repeat a valid request and never write)
            entry[b] = list[ic];
            const char *fp = bfrag + (((int64_t)ic * 2 + half) * KS) * 16;
#pragma unroll
            for (int s = 0; s < KS; ++s) xf[b][s] = *(const half8 *)(fp + s * 16);
        }
        float W1[G], W2[G];
        int Tb[G];
#pragma unroll
        for (int b = 0; b < G; ++b) { W1[b] = -INFINITY; W2[b] = -INFINITY; Tb[b] = 0; }
        half8 anext[KS];
#pragma unroll
        for (int s = 0; s < KS; ++s) anext[s] = *(const half8 *)(sb + (ti0 * NSTEP + sub) * VQ_CHUNK_BYTES + piece + s * 512);
#pragma unroll 1
        for (int j = 0; j < GT; ++j) {
            half8 af[KS];
#pragma unroll
            for (int s = 0; s < KS; ++s) af[s] = anext[s];
            if (j + 1 < GT) {
#pragma unroll
                for (int s = 0; s < KS; ++s)
                    anext[s] = *(const half8 *)(sb + ((ti0 + j + 1) * NSTEP + sub) * VQ_CHUNK_BYTES + piece + s * 512);
            }
            f32x16 init;
            if (zero_init) {
#pragma unroll
                for (int q = 0; q < NE; ++q) init[q] = 0.0f;
            } else {                   // -se |e|^2 / 2 of this lane's code rows (r & 3) + 8 (r >> 2) + 4 half
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 a4 = *(const f32x4 *)(sb + TPS * NSTEP * VQ_CHUNK_BYTES + ((ti0 + j) * 32 + 8 * g + 4 * half) * 4);
#pragma unroll
                    for (int q = 0; q < 4; ++q) init[4 * g + q] = a4[q];
                }
            }
#pragma unroll
            for (int b = 0; b < G; ++b) {
                if (base + 32 * b >= cnt) break;           // wave-uniform
                f32x16 acc = init;
#pragma unroll
                for (int s = 0; s < KS; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[s], xf[b][s], acc, 0, 0, 0);
                float w1 = -INFINITY, w2 = -INFINITY;
#pragma unroll
                for (int e = 0; e < NE; ++e) {             // element id = accumulator register (4 bits)
                    const float v = __uint_as_float((__float_as_uint(acc[e]) & 0xFFFFFFF0u) | (uint32_t)e);
                    w2 = __builtin_amdgcn_fmed3f(w1, w2, v);
                    w1 = vmax(w1, v);
                }
                // fold the tile's (best, runner-up) into the group's: second largest of {W1, W2, w1, w2} = med3(W1, w1, max(W2, w2))
                W2[b] = __builtin_amdgcn_fmed3f(W1[b], w1, fmaxf(W2[b], w2));
                Tb[b] = (w1 > W1[b]) ? j : Tb[b];
                W1[b] = fmaxf(W1[b], w1);
            }
        }
#pragma unroll
        for (int b = 0; b < G; ++b) {
            const int want = (int)((entry[b] >> 30) & 1u), slot = (int)(entry[b] >> 31);    // lane half that asked, record slot
            if (base + 32 * b + col < cnt && half == want) {
                const int64_t tk = (int64_t)(entry[b] & 0x3FFFFFFFu);
                const uint32_t code = (uint32_t)(gid * GT + Tb[b]) * 32u + (uint32_t)mfma_row((int)(__float_as_uint(W1[b]) & 15u), half);
                float *rp = rec + ((int64_t)sl * VQ_REC_FIELDS + 2 * slot) * Np + tk;
                rp[0] = W1[b];
                rp[Np] = __uint_as_float(code);
                grp.rece2[(int64_t)(sl * 2 + slot) * Np + tk] = W2[b];
            }
        }
    }
}
