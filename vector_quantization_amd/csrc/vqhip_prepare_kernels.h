// libvqhip device kernels, unit 2 of 8: row norms / F.normalize, codebook image (cb_stats, cb_image), token image (x_prep,
// pre_kernel; NCHW-direct form).  Included by vqhip_kernels.h.
#pragma once
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

// Four codebook rows k0 .. k0 + 3 by one wave: |e_k|^2 in oracle order, optional normalisation into e_exact (cosine), max |e|,
// flags — folded into the caller's running maxima (amax per lane; m_e2 / m_en wave-uniform; bad per lane).
// The four rows are loaded together and reduced with interleaved shuffle trees.
// NI = 4 (D <= 256): the wave's four rows live in registers (lane l holds dims l, l + 64, l + 128, l + 192 of each): all 16
// loads are in flight together and the cosine form normalises from the registers instead of reading the rows again (19.7 -> us
// at K = 16 384, D = 256, cosine).  NI = 0 (larger D): the same 16 requests at a time over chunks of 256 dims, rows read a second
// time for the normalisation — element by element, these loops made cb_cos_kernel 54 us at D = 768 for the 63 MB it moves;
// holding the rows of D <= 1024 in registers instead cost the D <= 256 kernels that share the launch 1-2 us in occupancy.
// Same per-lane chains in the same order: same |e_k|^2 to the bit.
template <int NI>
__device__ __forceinline__ void cb_rows4_impl(int64_t k0, const float *e, int64_t K, int D, int metric, float *en, float *ex,
                                              float &amax, float &m_e2, float &m_en, bool &bad) {
    const int lane = threadIdx.x & 63;
    float p[4] = {0, 0, 0, 0};
    constexpr bool inreg = NI > 0;
    float av[4][inreg ? NI : 1];
    if constexpr (inreg) {
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int d = lane + 64 * i;
                av[c][i] = (d < D && k0 + c < K) ? e[(k0 + c) * D + d] : 0.0f;
            }
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float a = av[c][i];
                p[c] = fmaf(a, a, p[c]); amax = fmaxf(amax, fabsf(a)); bad |= !isfinite(a);     // (a = 0 past D: p + 0 = p exactly)
            }
    } else
    for (int d0 = lane; d0 < D; d0 += 256) {                    // 16 requests in flight (4 per row), consumed in the order of d
        float t[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int d = d0 + 64 * i;
                t[c][i] = (d < D && k0 + c < K) ? e[(k0 + c) * D + d] : 0.0f;
            }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float a = t[c][i];
                p[c] = fmaf(a, a, p[c]); amax = fmaxf(amax, fabsf(a)); bad |= !isfinite(a);     // (a = 0 past D: p + 0 = p exactly)
            }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1)
#pragma unroll
        for (int c = 0; c < 4; ++c) p[c] = p[c] + __shfl_xor(p[c], off, 64);
    if (VQ_IS_COS(metric)) {
        amax = 0.0f;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            if (k0 + c >= K) continue;
            float nrm = sqrtf(p[c]);
            float den = (nrm < 1e-12f) ? 1e-12f : nrm;
            float q2 = 0.0f;
            if constexpr (inreg) {
#pragma unroll
                for (int i = 0; i < NI; ++i) {
                    const int d = lane + 64 * i;
                    if (d < D) {
                        float a = av[c][i] / den;
                        if (VQ_IS_BF16(metric)) a = bf16_rne(a);
                        ex[(k0 + c) * D + d] = a;
                        amax = fmaxf(amax, fabsf(a)); bad |= !isfinite(a); q2 = fmaf(a, a, q2);
                    }
                }
            } else
            for (int d0 = lane; d0 < D; d0 += 256) {            // the row again (L2), four requests in flight
                float t[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) t[i] = (d0 + 64 * i < D) ? e[(k0 + c) * D + d0 + 64 * i] : 0.0f;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int d = d0 + 64 * i;
                    if (d < D) {
                        float a = t[i] / den;
                        if (VQ_IS_BF16(metric)) a = bf16_rne(a);    // bf16-autocast: the einsum sees bf16(normalize(e))
                        ex[(k0 + c) * D + d] = a;
                        amax = fmaxf(amax, fabsf(a)); bad |= !isfinite(a); q2 = fmaf(a, a, q2);
                    }
                }
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
            if (VQ_IS_L2(metric)) m_en = fmaxf(m_en, -p[c]);
        }
    }
}
__device__ __forceinline__ void cb_rows4(int64_t k0, const float *e, int64_t K, int D, int metric, float *en, float *ex,
                                         float &amax, float &m_e2, float &m_en, bool &bad) {
    if (D <= 256) cb_rows4_impl<4>(k0, e, K, D, metric, en, ex, amax, m_e2, m_en, bad);
    else cb_rows4_impl<0>(k0, e, K, D, metric, en, ex, amax, m_e2, m_en, bad);
}

// pass 1 (one wave per 4 codes): |e_k|^2 in oracle order, optional normalisation into e_exact, max|e|, flags.
// The four rows of a wave are loaded together and reduced with interleaved shuffle trees; maxima are reduced per
// block and written as one partial per block (same-line atomics from ~1000 concurrent blocks cost ~25 us).
__device__ __forceinline__ void cb_stats_body(int64_t blk, const float *e, int64_t K, int D, int metric, char *cb, const VqCbLayout &L) {
    __shared__ float red[4];
    const int wave = threadIdx.x >> 6;
    VqCbStats *st = (VqCbStats *)(cb + L.off_stats);
    float *en = (float *)(cb + L.off_en);
    float *ex = (float *)(cb + L.off_eexact);
    // the maxima slots cb_image_kernel (the next launch) raises start from zero
    if (blk == 0 && threadIdx.x < VQ_CB_SLOTS) {
        uint32_t *slot = (uint32_t *)(cb + L.off_stats + 256 + threadIdx.x * 128);
        slot[0] = 0u; slot[1] = 0u; slot[2] = 0u;
    }
    const int64_t k0 = (blk * 4 + wave) * 4;
    float amax = 0.0f;
    bool bad = false;
    float m_e2 = 0.0f, m_en = VQ_IS_L2(metric) ? -INFINITY : 0.0f;   // L2: m_en = -(smallest |e_k|^2) (max-reduced like the rest)
    cb_rows4(k0, e, K, D, metric, en, ex, amax, m_e2, m_en, bad);
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
// COSFUSED: the cosine preparation in ONE launch (cb_cos_body below calls this behind its own normalisation pass): the scale is
// the constant 2^13 (normalised rows: max |e_hat| <= 1, header maxabs = 1.0f), nothing is reduced from the statistics
// partials, and the tile's maxima go into its own float4 of the part2 array instead of the zero-initialised slots.
template <bool COSFUSED>
__device__ __forceinline__ void cb_image_body(int64_t tile, const float *e, int64_t K, int D, int metric, char *cb, const VqCbLayout &L,
                                              float tile_e2max = 0.0f, float tile_bad = 0.0f) {
    __shared__ float red[2][8][32];
    __shared__ float red4[4];
    const int64_t stage = tile / L.tps;
    const int ti = (int)(tile % L.tps);
    const int r = threadIdx.x & 31, g = threadIdx.x >> 5;
    VqCbStats *st = (VqCbStats *)(cb + L.off_stats);
    const float *src = (VQ_IS_COS(metric)) ? (const float *)(cb + L.off_eexact) : e;
    const float *en = (const float *)(cb + L.off_en);
    // every block reduces the statistics partials (16 KiB, L2-resident) to the global maxima -> the common scale
    VqCbStats g_st;
    if constexpr (COSFUSED) {
        g_st.maxabs_bits = __float_as_uint(1.0f); g_st.nonfinite = 0u; g_st.l2_const_norm = 0u;
        if (tile == 0 && threadIdx.x == 0) {
            st->maxabs_bits = g_st.maxabs_bits; st->e2max_bits = 0u; st->enmax_bits = 0u; st->nonfinite = 0u; st->metric = metric;
            st->finalized = 1u; st->en_spread_bits = 0u; st->l2_const_norm = 0u; st->r2max_bits = 0u; st->eh2max_bits = 0u;
            st->part2_n = (uint32_t)L.nblk2; st->part2_off = (uint32_t)(L.off_part2 - L.off_stats); st->folded = 0u;
        }
    } else {
        const f32x4 *part = (const f32x4 *)(cb + L.off_part1);
        float a0 = 0.0f, a1 = 0.0f, a2 = VQ_IS_L2(metric) ? -INFINITY : 0.0f, a3 = 0.0f;
        for (int64_t i = threadIdx.x; i < L.nblk1; i += 256) {
            f32x4 v = part[i];
            a0 = fmaxf(a0, v[0]); a1 = fmaxf(a1, v[1]); a2 = fmaxf(a2, v[2]); a3 = fmaxf(a3, v[3]);
        }
        a0 = wave_max(a0); a1 = wave_max(a1); a2 = wave_max(a2); a3 = wave_max(a3);
        a0 = block_max4(a0, red4); a1 = block_max4(a1, red4); a2 = block_max4(a2, red4); a3 = block_max4(a3, red4);
        g_st.maxabs_bits = __float_as_uint(a0); g_st.e2max_bits = __float_as_uint(a1);
        // L2: the largest |e_k|^2 is a1 (the same sums), the smallest is -a2.  Constant-norm codebook (NormalizeCallback:
        // norms equal to a few ulp): the -|e_k|^2/2 term of the proposal score is the same for every code up to se*spread/2 —
        // it is dropped from the scores (aux = 0 below, no aux reads in coarse32_kernel) and the spread goes into the margin
        const bool l2 = VQ_IS_L2(metric);
        const float spread = l2 ? a1 + a2 : 0.0f;
        const bool const_norm = l2 && a3 == 0.0f && a1 > 0.0f && spread >= 0.0f && spread <= a1 * 1.52587890625e-05f;
        g_st.enmax_bits = l2 ? __float_as_uint(a1) : 0u;
        g_st.en_spread_bits = const_norm ? __float_as_uint(spread) : 0u;
        g_st.l2_const_norm = const_norm ? 1u : 0u;
        g_st.nonfinite = a3 > 0.0f ? 1u : 0u;
        if (tile == 0 && threadIdx.x == 0) {
            st->maxabs_bits = g_st.maxabs_bits; st->e2max_bits = g_st.e2max_bits; st->enmax_bits = g_st.enmax_bits;
            st->nonfinite = g_st.nonfinite; st->metric = metric; st->finalized = 1u;
            st->en_spread_bits = g_st.en_spread_bits; st->l2_const_norm = g_st.l2_const_norm;
            st->r2max_bits = 0u; st->eh2max_bits = 0u;       // (the image's own maxima live in the slots: cb_stats_view)
            st->part2_n = 0u; st->part2_off = 0u; st->folded = 0u;
        }
    }
    const float se = cb_scale(&g_st), inv = 1.0f / se;
    const int64_t k = tile * VQ_TILE_CODES + r;
    char *stage_base = cb + L.off_frag + stage * L.stage_bytes;
    float r2 = 0.0f, h2 = 0.0f;
    // pieces of 8 dims: k-step of 32 dims s32 = piece/4, quarter q4 = piece%4; the tile's two 16-code halves go
    // to chunks (s32, 0) and (s32, 1); within a chunk lane = q4*16 + (code & 15)
    // (four pieces of a thread are requested before the first is converted: one piece at a time, the 12 pieces a thread has at
    //  D = 768 were 12 memory latencies in a row)
    const int npieces = L.nstep * 2;
    if (npieces <= 32) {                                       // D <= 256: at most 4 pieces per thread, one at a time (measured:
        for (int piece = g; piece < npieces; piece += 8) {     // requesting them together is 2 % slower per encode there)
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
    } else
    for (int p0 = g; p0 < npieces; p0 += 32) {
        float vv[4][8];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int piece = p0 + 8 * u;
            const int d0 = 32 * (piece >> 2) + 8 * (piece & 3);
            if (piece < npieces && k < K && d0 < D) {
                if (d0 + 8 <= D && (D % 4) == 0) { load8<0>(src, k * D + d0, vv[u]); }
                else {
#pragma unroll
                    for (int j = 0; j < 8; ++j) vv[u][j] = (d0 + j < D) ? src[k * D + d0 + j] : 0.0f;
                }
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int piece = p0 + 8 * u;
            if (piece >= npieces) break;
            const int s = piece >> 2, q4 = piece & 3;
            const int d0 = 32 * s + 8 * q4;
            half8 o;
            if (k < K && d0 < D) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    _Float16 q = to_f16_ftz(vv[u][j] * se);
                    float back = (float)q * inv, res = vv[u][j] - back;
                    r2 = fmaf(res, res, r2); h2 = fmaf(back, back, h2);
                    o[j] = q;
                }
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] = (_Float16)0.0f;
            }
            *(half8 *)(stage_base + (int64_t)(ti * L.nstep + 2 * s + (r >> 4)) * VQ_CHUNK_BYTES + (q4 * 16 + (r & 15)) * 16) = o;
        }
    }
    // aux chunk slice of this tile: -se*|e_k|^2/2 for its 32 codes (padded codes: a large FINITE negative score;
    // -inf with the register index or-ed into its mantissa would be a signalling NaN and poison v_max_f32)
    if (g == 0) {
        float v = (k < K) ? (g_st.l2_const_norm ? 0.0f : (-0.5f * en[k]) * se) : -3.0e38f;
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
    if constexpr (COSFUSED) {
        if (threadIdx.x == 0) ((f32x4 *)(cb + L.off_part2))[tile] = f32x4{a, b, fmaxf(badf, tile_bad), tile_e2max};
    } else if (threadIdx.x == 0) {
        uint32_t *slot = (uint32_t *)(cb + L.off_stats + 256 + (tile % VQ_CB_SLOTS) * 128);
        atomicMax(&slot[0], __float_as_uint(a));
        atomicMax(&slot[1], __float_as_uint(b));
        if (badf > 0.0f) atomicMax(&slot[2], 1u);
    }
}
__global__ __launch_bounds__(256) void cb_image_kernel(const float *e, int64_t K, int D, int metric, char *cb, VqCbLayout L) {
    cb_image_body<false>(blockIdx.x, e, K, D, metric, cb, L);
}

// Cosine (and DOT: unit rows as given, nothing normalised or copied) codebooks in ONE launch (one 256-thread block per tile of
// 32 codes): the tile's rows are normalised first — cb_rows4's
// arithmetic, a wave per 8 rows — into e_exact, then the block turns them into its piece of the fragment-major image.
// Possible because a normalised codebook needs no statistics pass for its scale (|e_hat| <= 1: the power-of-two scale is
// 2^13 whatever the data; fp16's relative precision does not depend on it), and because the tile's maxima go into a per-tile
// partial instead of slots somebody would have to zero first.  One launch less in front of every cosine encode; the blocks
// can share a launch with the token side (pre_kernel).
__device__ __forceinline__ void cb_cos_body(int64_t tile, const float *e, int64_t K, int D, int metric, char *cb, const VqCbLayout &L) {
    __shared__ float cred[4];
    const int wave = threadIdx.x >> 6;
    float *en = (float *)(cb + L.off_en);
    float *ex = (float *)(cb + L.off_eexact);
    float amax = 0.0f, m_e2 = 0.0f, m_en = 0.0f;
    bool bad = false;
    cb_rows4(tile * VQ_TILE_CODES + wave * 8, e, K, D, metric, en, ex, amax, m_e2, m_en, bad);
    cb_rows4(tile * VQ_TILE_CODES + wave * 8 + 4, e, K, D, metric, en, ex, amax, m_e2, m_en, bad);
    const float e2 = block_max4(m_e2, cred);
    const float badf = block_max4(__any(bad) ? 1.0f : 0.0f, cred);
    // the rows this block has just written are read back below by other waves of the block (same CU: workgroup scope)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    cb_image_body<true>(tile, e, K, D, metric, cb, L, e2, badf);
}
__global__ __launch_bounds__(256) void cb_cos_kernel(const float *e, int64_t K, int D, int metric, char *cb, VqCbLayout L) {
    cb_cos_body(blockIdx.x, e, K, D, metric, cb, L);
}

// Token side at a padded dimension of 32 (D <= 32: the VQ-KD and LlamaGen shapes): a wave = 16 tokens x 4 pieces of 8 dims,
// lane = piece * 16 + token — the fragment layout itself, so a wave's 64 image pieces are ONE contiguous 1 KiB chunk.  No LDS,
// no barrier: the oracle-order |x|^2 (64 interleaved partials of which 32 are live, then the halving tree 32, 16, ..., 1) is the
// same additions in the same order through three shuffle levels, and the residual sums fold the four pieces in piece order —
// every output is bit-identical to the general form below (which gives a token 8 threads of a 256-thread block, half of them
// idle at D = 32 and 7 of 8 at D = 8, behind three block barriers: 17.7 us for 100 352 x 32 rows with their normalisation).
template <int DT, bool XNORM>
__device__ __forceinline__ void x_prep_small(int64_t blk, const void *__restrict__ x, int64_t N, int D, char *__restrict__ ximg,
                                             float *__restrict__ xh2, float *__restrict__ rho2, float *__restrict__ xn,
                                             float *__restrict__ xq, float eps, int xround) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (wave >= 2) return;                                  // 32 tokens per block (the grid of the general form): two waves
    const int q4 = lane >> 4, r16 = lane & 15;
    const int64_t t = blk * 32 + wave * 16 + r16;
    const bool tvalid = t < N;
    const int64_t trow = tvalid ? t : (N - 1);
    const int d0 = 8 * q4;
    const bool have = tvalid && d0 < D;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = 0.0f;
    if (have) load8<DT>(x, trow * D + d0, v);
    // p[j] = partial 8 q4 + j of the 64 interleaved chains (one element each at D <= 32); result in every lane of the token
    auto tree = [&](const float (&p)[8]) -> float {
        float s[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) s[j] = p[j] + __shfl_xor(p[j], 32, 64);     // level 16: partials j and j + 16
#pragma unroll
        for (int j = 0; j < 8; ++j) s[j] = s[j] + __shfl_xor(s[j], 16, 64);     // level 8
        float a0 = s[0] + s[4], a1 = s[1] + s[5], a2 = s[2] + s[6], a3 = s[3] + s[7];   // level 4
        a0 = a0 + a2; a1 = a1 + a3;                                               // level 2
        a0 = a0 + a1;                                                             // level 1
        return __shfl(a0, r16, 64);                                               // piece 0's lane holds the row's sum
    };
    float pn[8];
    if constexpr (XNORM) {
#pragma unroll
        for (int j = 0; j < 8; ++j) pn[j] = fmaf(v[j], v[j], 0.0f);
        const float nrm = sqrtf(tree(pn));
        const float den = (nrm < eps) ? eps : nrm;
        if (have) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { v[j] = v[j] / den; if (xround) v[j] = bf16_rne(v[j]); }
            *(f32x4 *)(xq + trow * D + d0) = f32x4{v[0], v[1], v[2], v[3]};
            *(f32x4 *)(xq + trow * D + d0 + 4) = f32x4{v[4], v[5], v[6], v[7]};
        }
    }
    float s_h = 0.0f, s_r = 0.0f;
    half8 f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        _Float16 q = to_f16_ftz(v[j]);
        float b = (float)q, res = v[j] - b;
        s_h = fmaf(b, b, s_h); s_r = fmaf(res, res, s_r);
        pn[j] = fmaf(v[j], v[j], 0.0f);
        f[j] = have ? q : (_Float16)0.0f;
    }
    *(half8 *)(ximg + (blk * 2 + wave) * (int64_t)VQ_CHUNK_BYTES + lane * 16) = f;
    // |xh|^2 and |x - xh|^2: the four pieces' sums, folded in piece order (the general form adds its eight threads' in order)
    float a = __shfl(s_h, r16, 64), b = __shfl(s_r, r16, 64);
#pragma unroll
    for (int q = 1; q < 4; ++q) { a = a + __shfl(s_h, q * 16 + r16, 64); b = b + __shfl(s_r, q * 16 + r16, 64); }
    const float nn = tree(pn);
    if (q4 == 0 && tvalid) { xh2[t] = a; rho2[t] = b; xn[t] = nn; }
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
// GATHER (vqhip_col_argmin_rows: the rows are the LISTED codes of the codebook `x`): row t is x[grows[t]] for t < *gcount and
// zeros up to N (the launch is sized for a capacity, the count stays on the device); the gathered fp32 rows are written to
// `xrows` for the exact re-rank — the separate gather launch in front of the role-swapped pipeline is gone.
template <int DT, bool XNORM = false, bool NCHW = false, bool GATHER = false>
__device__ __forceinline__ void x_prep_body(int64_t blk, const void *__restrict__ x, int64_t N, int D, int nstep,
                                            char *__restrict__ ximg, float *__restrict__ xh2,
                                            float *__restrict__ rho2, float *__restrict__ xn,
                                            int *__restrict__ counters, int *__restrict__ arrive, int narrive,
                                            float *__restrict__ xq, float eps, int xround = 0,
                                            int32_t *__restrict__ hist_zero = nullptr, int64_t hist_len = 0,
                                            int64_t nblocks = 1, int64_t hw = 0, void *__restrict__ xrows = nullptr,
                                            const int32_t *__restrict__ grows = nullptr,
                                            const int32_t *__restrict__ gcount = nullptr, int tokens_off = 0) {
    static_assert(!GATHER || (DT == 0 && !XNORM && !NCHW), "GATHER: fp32 codebook rows, as given");
    __shared__ float red[2][8][32];
    // vqhip_encode(VQHIP_ENCODE_ZERO_HIST): the code-hit histogram the later kernels of this call add into starts from zero
    if (hist_zero != nullptr)
        for (int64_t i = blk * 256 + threadIdx.x; i < hist_len; i += nblocks * 256) hist_zero[i] = 0;
    __shared__ float part[64][32];   // the 64 interleaved partial sums of |x|^2 (oracle order), per token
    __shared__ float den_s[32];
    __shared__ float tile[NCHW ? 64 : 1][33];     // NCHW: 64 dims x 32 tokens of the map, turned here
    if (blk == 0 && threadIdx.x < 8) counters[threadIdx.x] = 0;   // housekeeping for the later kernels of this call (stream-ordered)
    // arrival counters of the proposal kernel's token blocks (at most one per 128 tokens: 4 blocks of this kernel)
    if (arrive != nullptr)                                      // every block zeroes its stride of the counter range
        for (int64_t i = blk * 256 + threadIdx.x; i < narrive; i += nblocks * 256) arrive[i] = 0;
    if (tokens_off) return;           // the proposal kernel makes its token fragments itself (coarse_kernel<..., XD>): housekeeping only
    if constexpr (!NCHW && !GATHER) {
        if (nstep == 2) {                                       // padded dimension 32: the wave-level form (no LDS, no barrier)
            x_prep_small<DT, XNORM>(blk, x, N, D, ximg, xh2, rho2, xn, xq, eps, xround);
            return;
        }
    }
    const int r = threadIdx.x & 31, g = threadIdx.x >> 5;
    const int64_t t = blk * 32 + r;
    const bool tvalid = t < N;
    const int64_t trow = tvalid ? t : (N - 1);
    const bool live = !GATHER || trow < (int64_t)gcount[0];
    const int64_t srow = GATHER ? (live ? (int64_t)grows[trow] : 0) : trow;      // the row the values come from
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
            load8<DT>(x, srow * D + 32 * (piece >> 2) + 8 * (piece & 3), v);
            if constexpr (GATHER) {
                if (!live) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = 0.0f;
                }
            }
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
            if constexpr (NCHW || GATHER) {        // the token-major rows as given, in the input's own dtype (exact: a copy)
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
                                                     int *__restrict__ arrive = nullptr, int narrive = 0, int tokens_off = 0) {
    x_prep_body<DT, false>(blockIdx.x, x, N, D, nstep, ximg, xh2, rho2, xn, counters, arrive, narrive, nullptr, 0.0f, 0, nullptr, 0,
                           (int64_t)gridDim.x, 0, nullptr, nullptr, nullptr, tokens_off);
}
// vqhip_encode / vqhip_col_argmin: the codebook statistics and the token side in ONE launch (they are independent;
// the image kernel that follows needs the former, the proposal kernel both)
// COSIMG: `nblk_stats` blocks of cb_cos_body (one per image tile: the whole cosine preparation) instead of statistics blocks —
// no image launch follows.
template <int DT, bool XNORM, bool NCHW = false, bool COSIMG = false, bool GATHER = false>
__global__ __launch_bounds__(256) void pre_kernel(const float *e, int64_t K, int metric, char *cb, VqCbLayout L, int nblk_stats,
                                                  const void *__restrict__ x, int64_t N, int D, int nstep,
                                                  char *__restrict__ ximg, float *__restrict__ xh2,
                                                  float *__restrict__ rho2, float *__restrict__ xn,
                                                  int *__restrict__ counters, int *__restrict__ arrive, int narrive,
                                                  float *__restrict__ xq, float eps, int32_t *__restrict__ hist_zero,
                                                  int64_t hw = 0, void *__restrict__ xrows = nullptr,
                                                  const int32_t *__restrict__ grows = nullptr,
                                                  const int32_t *__restrict__ gcount = nullptr, int tokens_off = 0) {
    // the smaller of the two groups of workgroups goes FIRST in the grid: dispatched behind the larger one it starts when that
    // one drains and its own latency (a chain of round trips either way) is added to the kernel — at 3072 tokens against 1024
    // statistics workgroups the token side started ~8 us into a 16 us kernel
    const int xgrid = (int)gridDim.x - nblk_stats;
    const bool x_first = xgrid < nblk_stats;
    const int b = (int)blockIdx.x;
    const bool is_x = x_first ? b < xgrid : b >= nblk_stats;
    if (!is_x) {
        if constexpr (COSIMG) cb_cos_body(x_first ? b - xgrid : b, e, K, D, metric, cb, L);
        else cb_stats_body(x_first ? b - xgrid : b, e, K, D, metric, cb, L);
    }
    else x_prep_body<DT, XNORM, NCHW, GATHER>((int64_t)(x_first ? b : b - nblk_stats), x, N, D, nstep, ximg, xh2, rho2, xn, counters, arrive, narrive,
                                              xq, eps, VQ_IS_BF16(metric) ? 1 : 0, hist_zero, K, (int64_t)xgrid, hw, xrows, grows, gcount, tokens_off);
}
