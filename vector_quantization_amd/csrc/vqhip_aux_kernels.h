// libvqhip device kernels, unit 8 of 8: transposes, codebook metrics, verification aids.  Included by vqhip_kernels.h.
#pragma once
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
