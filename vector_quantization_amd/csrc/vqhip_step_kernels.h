// libvqhip device kernels, unit 10: the pieces that only the one-call training forwards need (vqhip_vqkd_forward):
// the VQ-KD front (codebook normalised twice, latents normalised, exchange payload zeroed — one launch), centroid sums
// scattered straight into the packed exchange buffer, the EMA update read straight from it, and the decode / straight-through /
// normalised-MSE tail with its backward.  gfx950 only.  Reference: vq/algorithms/vq/callbacks/normalize.py:22-29,
// vq/algorithms/vqkd/quantizers/callbacks.py:44-75,114-129, vq/algorithms/vq/losses.py:37,53-62 (CommitmentLoss, mse norm=True).
#pragma once

// |v|^2 of one row in the oracle's order (lane l sums elements l, l + 64, ... with fma; halving tree) -> 1 / max(|v|, eps) is
// NOT formed: every consumer divides by the clamped norm like F.normalize does.
template <int DT>
__device__ __forceinline__ float row_den(const void *v, int64_t r, int D, int lane, float eps) {
    float p = 0.0f;
    for (int d = lane; d < D; d += 64) { const float a = load_elem<DT>(v, r * D + d); p = fmaf(a, a, p); }
    p = wave_sum_tree(p);
    const float nrm = sqrtf(p);
    return (nrm < eps) ? eps : nrm;
}

// One launch in front of a VQ-KD training forward, wave per row:
//   blocks [0, kblocks):  w_mid[k] = F.normalize(F.normalize(w_in[k]))   (NormalizeCallback.before_encode -> VQKDCallback.
//                         _update_embedding: callbacks/normalize.py:27, vqkd callbacks.py:73-75 — two normalisations)
//   blocks after:         xn[n] = F.normalize(x[n])                       (callbacks/normalize.py:24)
//   every block:          its share of `zero[0 .. nzero)` cleared (the payload of the packed exchange buffer)
// Bit-identical to normalize_rows_kernel applied twice / once.
template <int DT>
__global__ __launch_bounds__(256) void vqkd_front_kernel(const float *__restrict__ w_in, float *__restrict__ w_mid, int64_t K,
                                                         const void *__restrict__ x, float *__restrict__ xn, int64_t N, int D,
                                                         float eps, int kblocks, float *__restrict__ zero, int64_t nzero) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4; i < nzero; i += (int64_t)gridDim.x * 1024)
        *(float4 *)(zero + i) = make_float4(0.0f, 0.0f, 0.0f, 0.0f);          // nzero % 4 == 0 (K * D with D % 8 == 0)
    if ((int)blockIdx.x < kblocks) {
        const int64_t k = (int64_t)blockIdx.x * 4 + wave;
        if (k >= K) return;
        const float den1 = row_den<0>(w_in, k, D, lane, eps);
        float p = 0.0f;
        for (int d = lane; d < D; d += 64) { const float y = w_in[k * D + d] / den1; p = fmaf(y, y, p); }
        p = wave_sum_tree(p);
        const float n2 = sqrtf(p), den2 = (n2 < eps) ? eps : n2;
        for (int d = lane; d < D; d += 64) w_mid[k * D + d] = (w_in[k * D + d] / den1) / den2;
        return;
    }
    const int64_t n = (int64_t)(blockIdx.x - kblocks) * 4 + wave;
    if (n >= N) return;
    const float den = row_den<DT>(x, n, D, lane, eps);
    for (int d = lane; d < D; d += 64) xn[n * D + d] = load_elem<DT>(x, n * D + d) / den;
}

// Header of the packed exchange buffer from the epilogue histogram (blocks [0, header_blocks)) and the centroid sums
// payload[idx[n]] += F.normalize(xn[n]) (callbacks.py:124: the latents, already normalised once by NormalizeCallback, are
// normalised again; computed here from xn in the oracle's order — NOT read from the encode's xq, which holds bf16-rounded
// rows under the bf16-autocast metric).  fp32 atomics, wave per token: 256 contiguous bytes per wave-instruction; the payload
// was zeroed by vqkd_front_kernel — vqkd callbacks.py:52-62 as ONE launch in front of the collective.
__global__ __launch_bounds__(256) void vqkd_scatter_pack_kernel(const int32_t *__restrict__ hist, int64_t numel,
                                                                const float *__restrict__ xn, const int64_t *__restrict__ idx,
                                                                int64_t N, int64_t K, int D, float eps, float *packed,
                                                                int header_blocks, int scatter) {
    if ((int)blockIdx.x < header_blocks) {
        const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
        if (k < K) {
            const int64_t h = hist[k];
            packed[k] = (float)(h & 0xFFFF);
            packed[K + k] = (float)(h >> 16);
        }
        if (k == 0) {
            packed[2 * K] = (float)(numel & 0xFFFF);
            packed[2 * K + 1] = (float)((numel >> 16) & 0xFFFF);
            packed[2 * K + 2] = (float)(numel >> 32);
            packed[2 * K + 3] = 0.0f;
        }
        return;
    }
    if (!scatter) return;
    const int64_t n = (int64_t)(blockIdx.x - header_blocks) * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (n >= N) return;
    const int64_t k = idx[n];
    if (k < 0 || k >= K) return;
    float *dst = packed + VQ_PACK_HEADER(K) + k * D;
    const float den = row_den<0>(xn, n, D, lane, eps);
    for (int d = lane; d < D; d += 64) atomicAdd(&dst[d], xn[n * D + d] / den);
}

// vqkd_update_kernel on the (all-reduced) packed buffer, w_in -> w_out (may alias): counts from the header, sums from the
// payload — the expressions of vqkd_update_kernel in the same order (bit-identical), without the unpack launch in between.
__global__ void vqkd_update_packed_kernel(const float *w_in, float *w_out, const float *__restrict__ packed, int64_t K, int D,
                                          float decay) {
    const int64_t k = (int64_t)blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (k >= K) return;
    const int64_t occ = unpack_count(packed, K, k);
    const float cnt = (float)(occ > 0 ? occ : 1);
    const float *sums = packed + VQ_PACK_HEADER(K);
    float p = 0.0f;
    for (int d = lane; d < D; d += 64) {
        const float c = (occ > 0) ? sums[k * D + d] / cnt : w_in[k * D + d];
        p = fmaf(c, c, p);
    }
    p = wave_sum_tree(p);
    const float nrm = sqrtf(p), den = (nrm < 1e-12f) ? 1e-12f : nrm;
    const float om = 1.0f - decay;
    float q = 0.0f;
    for (int d = lane; d < D; d += 64) {
        float c = (occ > 0) ? sums[k * D + d] / cnt : w_in[k * D + d];
        c = c / den;
        const float v = w_in[k * D + d] * decay + c * om;
        q = fmaf(v, v, q);
    }
    q = wave_sum_tree(q);
    const float nrm2 = sqrtf(q), den2 = (nrm2 < 1e-12f) ? 1e-12f : nrm2;
    // (the row is re-read before it is overwritten: every lane writes only the elements it has read)
    for (int d = lane; d < D; d += 64) {
        float c = (occ > 0) ? sums[k * D + d] / cnt : w_in[k * D + d];
        c = c / den;
        const float v = w_in[k * D + d] * decay + c * om;
        w_out[k * D + d] = v / den2;
    }
}

// Tail of the VQ-KD forward, wave per token (grid-stride), 16 waves per workgroup and at most 256 workgroups (the double sum and
// the ticket are two same-address atomics per WORKGROUP: 1024 workgroups of 4 waves made this a 33 us kernel, 16 x 256 a 6 us one):
//   z = w[idx[n]];  z_ste = xn + (z - xn)  (utils/ste.py:10);  loss term: (F.normalize(z) - F.normalize(xn))^2
//   mse[0] = mean over N*D (CommitmentLoss with mse norm=True, losses.py:37,62), mse[1..3] = mse[0], 0, 0
// `sse`: the 16-byte zeroed scratch of gather_ste_loss_kernel (double sum + ticket), left zeroed.
__global__ __launch_bounds__(1024) void vqkd_tail_kernel(const float *__restrict__ xn,
                                                         const float *__restrict__ w, const int64_t *__restrict__ idx, int64_t N,
                                                         int D, float eps, float *__restrict__ z_ste, double *sse,
                                                         float *__restrict__ mse) {
    __shared__ double red[16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double s = 0.0;
    for (int64_t n = (int64_t)blockIdx.x * 16 + wave; n < N; n += (int64_t)gridDim.x * 16) {
        const int64_t k = idx[n];
        // both rows are requested before either norm is reduced
        float pz = 0.0f, px = 0.0f;
        for (int d = lane; d < D; d += 64) { const float a = w[k * D + d], b = xn[n * D + d]; pz = fmaf(a, a, pz); px = fmaf(b, b, px); }
        pz = wave_sum_tree(pz); px = wave_sum_tree(px);
        const float nz = sqrtf(pz), nx = sqrtf(px);
        const float den = (nz < eps) ? eps : nz, dnx = (nx < eps) ? eps : nx;
        for (int d = lane; d < D; d += 64) {
            const float zv = w[k * D + d], xv = xn[n * D + d];
            if (z_ste) z_ste[n * D + d] = xv + (zv - xv);
            const float df = zv / den - xv / dnx;
            s += (double)(df * df);
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off, 64);
    if (lane == 0) red[wave] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
#pragma unroll
        for (int i = 0; i < 16; ++i) t += red[i];
        int *ticket = (int *)(sse + 1);
        const double before = atomicAdd(sse, t);              // (returning atomic: complete before the ticket is taken)
        asm volatile("" :: "v"(before) : "memory");
        if (atomicAdd(ticket, 1) == (int)gridDim.x - 1) {
            const double total = __hip_atomic_load(sse, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const float mean = (float)(total / ((double)N * (double)D));
            mse[0] = mean; mse[1] = mean; mse[2] = 0.0f; mse[3] = 0.0f;
            __hip_atomic_store(sse, 0.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// Backward of   xn = F.normalize(x);  z_ste = xn + sg(z - xn);  loss = mean((F.normalize(sg z) - F.normalize(xn))^2)
// with respect to x, wave per token:
//   gt   = g_loss * 2/(N D) * (t - zn),  t = F.normalize(xn), zn = F.normalize(w[idx])          (d loss / d t)
//   g_xn = normalize_bwd(xn; gt) + g_zste                                                         (straight-through: identity)
//   gx   = normalize_bwd(x; g_xn)
// normalize_bwd(v; g) = (g - y (y.g)) / max(|v|, eps), y = v / max(|v|, eps)   (rows with |v| < eps: g / eps) — the expression
// of normalize_bwd_kernel.  g_zste / g_loss nullable (= 0).
template <int DT>
__global__ __launch_bounds__(256) void vqkd_backward_kernel(const void *__restrict__ x, const float *__restrict__ xn,
                                                            const float *__restrict__ w, const int64_t *__restrict__ idx,
                                                            int64_t N, int D, float eps, const float *__restrict__ g_zste,
                                                            const float *__restrict__ g_loss, float *__restrict__ gx) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float gl = g_loss ? *g_loss : 0.0f;
    const float sc = gl * (2.0f / ((float)N * (float)D));
    for (int64_t n = (int64_t)blockIdx.x * 4 + wave; n < N; n += (int64_t)gridDim.x * 4) {
        const int64_t k = idx[n];
        // norms (oracle order): |x|, |xn|, |z|
        float px = 0.0f, pn = 0.0f, pz = 0.0f;
        for (int d = lane; d < D; d += 64) {
            const float a = load_elem<DT>(x, n * D + d), b = xn[n * D + d], c = w[k * D + d];
            px = fmaf(a, a, px); pn = fmaf(b, b, pn); pz = fmaf(c, c, pz);
        }
        px = wave_sum_tree(px); pn = wave_sum_tree(pn); pz = wave_sum_tree(pz);
        const float nx = sqrtf(px), nn = sqrtf(pn), nz = sqrtf(pz);
        const bool cx = nx < eps, cn = nn < eps;
        const float dx = cx ? eps : nx, dn = cn ? eps : nn, dz = (nz < eps) ? eps : nz;
        // first normalize_bwd: dot1 = sum t * gt
        float dot1 = 0.0f;
        for (int d = lane; d < D; d += 64) {
            const float t = xn[n * D + d] / dn;
            const float gt = sc * (t - w[k * D + d] / dz);
            dot1 = fmaf(t, gt, dot1);
        }
        dot1 = wave_sum_tree(dot1);
        // g_xn and the second dot: dot2 = sum y * g_xn, y = x / dx
        float dot2 = 0.0f;
        for (int d = lane; d < D; d += 64) {
            const float t = xn[n * D + d] / dn;
            const float gt = sc * (t - w[k * D + d] / dz);
            const float g1 = cn ? gt / dn : (gt - t * dot1) / dn;
            const float gxn = g1 + (g_zste ? g_zste[n * D + d] : 0.0f);
            const float y = load_elem<DT>(x, n * D + d) / dx;
            dot2 = fmaf(y, gxn, dot2);
        }
        dot2 = wave_sum_tree(dot2);
        for (int d = lane; d < D; d += 64) {
            const float t = xn[n * D + d] / dn;
            const float gt = sc * (t - w[k * D + d] / dz);
            const float g1 = cn ? gt / dn : (gt - t * dot1) / dn;
            const float gxn = g1 + (g_zste ? g_zste[n * D + d] : 0.0f);
            const float y = load_elem<DT>(x, n * D + d) / dx;
            gx[n * D + d] = cx ? gxn / dx : (gxn - y * dot2) / dx;
        }
    }
}
