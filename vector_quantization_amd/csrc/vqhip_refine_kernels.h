// libvqhip device kernels, unit 4 of 8: the row margin, the decision stage, the second proposal pass (rescan_kernel), the exact
// fp32 re-rank and the last-resort whole-codebook pass for listed rows.  Included by vqhip_kernels.h.
#pragma once
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
        // constant-norm codebook: the scores carry no -|e_k|^2/2 term; two codes' terms differ by at most spread/2 (twice that
        // is added: generous, and the spread of a normalised codebook is ~2^-22 of the norm)
        if (st->l2_const_norm != 0) m += __uint_as_float(st->en_spread_bits);
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
#ifndef VQ_RESCAN_LOCAL
#define VQ_RESCAN_LOCAL 8    // ... of which one (row block, slice) item of the second pass may contribute
#endif

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

// GROUPS (the D <= 32 group path: coarse32_kernel + identify32_kernel): fields (0, 1) and (2, 3) of a record hold the
// candidates the two lane halves of the token identified — in no particular order, either may be absent (code 0xFFFFFFFF) —
// and rece2[slice][half][n] the runner-up inside each identified group, a bound like v3.
template <int NSL, bool AGENT, bool GROUPS = false>
__device__ __forceinline__ void decide_rows_impl(int64_t n, bool oob, const VqCbStats *st, int Dp, int metric, int nslices,
                                                 const float *rec, const float *xh2, const float *rho2, int64_t Np,
                                                 const VqDecideOut &o, int *wcount, int *wbase, const float *rece2 = nullptr) {
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
        float v1[NSL], v2[NSL], v3[NSL], c1[NSL], c2[NSL];
#pragma unroll
        for (int s = 0; s < NSL; ++s) {
            const float *rp = rec + (int64_t)s * VQ_REC_FIELDS * Np + n;
            v1[s] = rec_load<AGENT>(rp); c1[s] = rec_load<AGENT>(rp + Np);
            v2[s] = rec_load<AGENT>(rp + 2 * Np); v3[s] = rec_load<AGENT>(rp + 4 * Np);
            if constexpr (GROUPS) {
                c2[s] = rec_load<AGENT>(rp + 3 * Np);
                v3[s] = fmaxf(v3[s], fmaxf(rece2[(int64_t)(2 * s) * Np + n], rece2[(int64_t)(2 * s + 1) * Np + n]));
            }
        }
#pragma unroll
        for (int s = 0; s < NSL; ++s) gbest = fmaxf(gbest, GROUPS ? fmaxf(v1[s], v2[s]) : v1[s]);
        if (!(gbest > -INFINITY) || !isfinite(gbest)) invalid = true;
        m = bf16_tight_margin(m, bf16_worst, gbest, &stv);
        thr = gbest - m;           // m > 0, so thr <= gbest and the best record always qualifies
#pragma unroll
        for (int s = 0; s < NSL; ++s) {
            if (v3[s] >= thr) unidentified = true;
            if (v1[s] >= thr) { ++nc; best = __float_as_uint(c1[s]); }
            if (v2[s] >= thr) { ++nc; if constexpr (GROUPS) best = __float_as_uint(c2[s]); }
        }
    } else {
        static_assert(!GROUPS, "the group path runs with 1, 2 or 4 slices");
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
                                            const VqDecideOut &o, int *wcount, int *wbase, const float *rece2) {
    if (rece2 != nullptr) {  // group path (uniform)
        switch (nslices) {
#define VQ_DECIDE_CASE(NSL) case NSL: decide_rows_impl<NSL, AGENT, true>(n, oob, st, Dp, metric, nslices, rec, xh2, rho2, Np, o, wcount, wbase, rece2); break;
            VQ_DECIDE_CASE(1) VQ_DECIDE_CASE(2) VQ_DECIDE_CASE(4)
#undef VQ_DECIDE_CASE
            default: break;  // (launch_coarse never picks another count on that path)
        }
        return;
    }
    switch (nslices) {      // every thread of the workgroup takes the same case (the list appends contain barriers)
#define VQ_DECIDE_CASE(NSL) case NSL: decide_rows_impl<NSL, AGENT>(n, oob, st, Dp, metric, nslices, rec, xh2, rho2, Np, o, wcount, wbase); break;
        VQ_DECIDE_CASE(1) VQ_DECIDE_CASE(2) VQ_DECIDE_CASE(4) VQ_DECIDE_CASE(8) VQ_DECIDE_CASE(16)
#undef VQ_DECIDE_CASE
        default: decide_rows_impl<0, AGENT>(n, oob, st, Dp, metric, nslices, rec, xh2, rho2, Np, o, wcount, wbase); break;
    }
}

// stand-alone form (one thread per token, 1024-thread workgroups): used when the proposal kernel does not decide itself
__global__ void refine_decide_kernel(const char *cb, VqCbLayout L, int64_t N, int metric, int nslices, const float *rec,
                                     const float *xh2, const float *rho2, int64_t Np, VqDecideOut o, const float *rece2 = nullptr) {
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
    decide_rows<false>(n, oob, (const VqCbStats *)(cb + L.off_stats), L.Dp, metric, nslices, rec, xh2, rho2, Np, o, wcount, wbase, rece2);
}

// Second proposal pass over the rows of rescan_list only: same fp16 MFMA scores as coarse_kernel (bitwise: same
// operands, same instruction sequence per accumulator), but every score >= the row's threshold is appended to the row's
// candidate list.  Same machinery as coarse_kernel — the rows' fragments (from the packed image) stay in registers,
// codebook stages arrive by LDS-DMA through the same ring and are shared by the 8 waves — as a persistent grid over
// (block of WAVES*TT*16 queued rows, slice of stages) items, the slice count chosen on the device from the queue
// length so that every workgroup gets an item.
// XD != 0 (1: bf16 rows, 2: fp32 rows): there is no token image (coarse_kernel<..., XD>) — the queued rows' fragments come from
// the row-major latents `xrows` (D == the padded dimension): 4 or 8 whole cache lines per row instead of 32 pieces in 32 lines,
// converted as the prologue of the proposal kernel converts them (the same fp16 values: the scores stay bitwise the stream's)
template <int NSTEP, int TT, int WAVES, int TPS, int NBUF = 2, int XD = 0>
__global__ __launch_bounds__(WAVES * 64) void rescan_kernel(const char *__restrict__ ximg, const char *__restrict__ frag,
                                                            int64_t nstages, const int *__restrict__ rescan_list,
                                                            const int *__restrict__ counters, const float *__restrict__ thr,
                                                            int *__restrict__ rescan_cnt, int *__restrict__ cand_list,
                                                            const void *__restrict__ xrows = nullptr) {
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
    // every item gathers its rows' fragments from the token image again — 16-byte pieces, each in a 128-byte line of its own:
    // 4 KB of L2 requests per row and item — so the slice count stops at the LAST power of two that still fits the grid in one
    // round (it used to go one further: 15 row blocks x 32 slices = 480 items on 256 workgroups, two rounds and twice the gathers)
    // (D = 768, 1498 queued rows: 70 -> 51 us; neutral at D <= 256, where the item is bound by the latency of its short stream)
    while (ntb * ns * 2 <= (int64_t)gridDim.x && ns * 2 <= nstages) ns <<= 1;

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
            if constexpr (XD == 0) {
                const char *src = ximg + (tk >> 4) * (int64_t)(NS32 * VQ_CHUNK_BYTES) + ((lane >> 4) * 16 + (int)(tk & 15)) * 16;
#pragma unroll
                for (int s = 0; s < NS32; ++s) xf[t][s] = *(const half8 *)(src + s * VQ_CHUNK_BYTES);
            } else {
#pragma unroll
                for (int s = 0; s < NS32; ++s) {
                    float v[8];
                    load8<(XD == 1 ? 1 : 0)>(xrows, tk * (int64_t)(NS32 * 32) + 32 * s + 8 * (lane >> 4), v);
                    half8 f;
#pragma unroll
                    for (int j = 0; j < 8; ++j) f[j] = to_f16_ftz(v[j]);
                    xf[t][s] = f;
                }
            }
        }
        vq_dma_barrier();  // stage st0 landed
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
            vq_dma_barrier();  // next stage landed and everybody is done reading this one
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
                                            u64 *__restrict__ keys, int groups = 0) {
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
            float gbest = ((j & 1) && !groups) ? -INFINITY : v;   // best first-field value over the row's slices (group path: the
                                                                  // two fields of a record are unordered, every slot counts)
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
#ifndef VQ_RR_PD
#define VQ_RR_PD 2
#endif
        constexpr int PD = VQ_RR_PD;
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
                                                            u64 *__restrict__ keys, int groups = 0) {
    __shared__ __attribute__((aligned(16))) float tile_e[4][32 * VQ_RR_STRIDE];
    __shared__ __attribute__((aligned(16))) float tile_x[4][8 * VQ_RR_STRIDE];
    const int wave = threadIdx.x >> 6;
    if ((int)blockIdx.x < g0)
        rerank_rows<DT, 0>(tile_e[wave], tile_x[wave], (int64_t)blockIdx.x * 4 + wave, (int64_t)g0 * 4, x, e_exact, cb, L, D,
                           metric, nslices, S0, rec, xh2, rho2, xnorm, Np, idx, hist, multi_list, counters, nullptr, nullptr,
                           nullptr, nullptr, groups);
    else
        rerank_rows<DT, 1>(tile_e[wave], tile_x[wave], (int64_t)(blockIdx.x - g0) * 4 + wave,
                           (int64_t)(gridDim.x - g0) * 4, x, e_exact, cb, L, D, metric, nslices, VQ_RESCAN_CAP, rec, xh2,
                           rho2, xnorm, Np, idx, hist, rescan_list, counters, rescan_cnt, cand_list, exact_list, keys);
}
