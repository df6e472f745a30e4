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
// w_passes: 2 = the VQ-KD front above; 1 = NormalizeCallback alone (w_mid = F.normalize(w_in), callbacks/normalize.py:27: the
// front of vqhip_vq_forward); kblocks == 0 / N == 0: that side is not wanted.
template <int DT>
__global__ __launch_bounds__(256) void vqkd_front_kernel(const float *__restrict__ w_in, float *__restrict__ w_mid, int64_t K,
                                                         const void *__restrict__ x, float *__restrict__ xn, int64_t N, int D,
                                                         float eps, int kblocks, float *__restrict__ zero, int64_t nzero,
                                                         int w_passes = 2) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4; i < nzero; i += (int64_t)gridDim.x * 1024)
        *(float4 *)(zero + i) = make_float4(0.0f, 0.0f, 0.0f, 0.0f);          // nzero % 4 == 0 (K * D with D % 8 == 0)
    if ((int)blockIdx.x < kblocks) {
        const int64_t k = (int64_t)blockIdx.x * 4 + wave;
        if (k >= K) return;
        const float den1 = row_den<0>(w_in, k, D, lane, eps);
        if (w_passes == 1) {
            for (int d = lane; d < D; d += 64) w_mid[k * D + d] = w_in[k * D + d] / den1;
            return;
        }
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

// The same launch at D <= 32 (the VQ-KD and LlamaGen shapes): L lanes per row, 64 / L rows per wave instead of a wave per row —
// at D = 32 half of the lanes of the wave-per-row form idle, at D = 8 seven of eight, and the launch is 20 000 waves of two or three
// dependent passes each (8.6 us for 8192 + 12 544 rows of 32 dims: 2.6 MB).  The halving tree runs inside the L-lane group: the very
// additions of the full-wave tree, whose upper levels only add the zeros of the lanes past D (row_small_kernel) — bit-identical.
template <int DT, int L>
__global__ __launch_bounds__(256) void vqkd_front_small_kernel(const float *__restrict__ w_in, float *__restrict__ w_mid, int64_t K,
                                                               const void *__restrict__ x, float *__restrict__ xn, int64_t N, int D,
                                                               float eps, int kblocks, float *__restrict__ zero, int64_t nzero,
                                                               int w_passes) {
    constexpr int RPW = 64 / L, RPB = 4 * RPW;             // rows per wave / per block
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4; i < nzero; i += (int64_t)gridDim.x * 1024)
        *(float4 *)(zero + i) = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    const int d = lane % L, rw = lane / L;
    auto group_sum = [](float p) {
#pragma unroll
        for (int off = L / 2; off >= 1; off >>= 1) p = p + __shfl_xor(p, off, 64);
        return p;
    };
    if ((int)blockIdx.x < kblocks) {
        const int64_t k = (int64_t)blockIdx.x * RPB + wave * RPW + rw;
        const bool live = k < K && d < D;
        const float a = live ? w_in[k * D + d] : 0.0f;
        const float n1 = sqrtf(group_sum(fmaf(a, a, 0.0f))), den1 = (n1 < eps) ? eps : n1;
        const float y = a / den1;
        if (w_passes == 1) { if (live) w_mid[k * D + d] = y; return; }
        const float n2 = sqrtf(group_sum(live ? fmaf(y, y, 0.0f) : 0.0f)), den2 = (n2 < eps) ? eps : n2;
        if (live) w_mid[k * D + d] = y / den2;
        return;
    }
    const int64_t n = (int64_t)(blockIdx.x - kblocks) * RPB + wave * RPW + rw;
    const bool live = n < N && d < D;
    const float a = live ? load_elem<DT>(x, n * D + d) : 0.0f;
    const float nrm = sqrtf(group_sum(fmaf(a, a, 0.0f))), den = (nrm < eps) ? eps : nrm;
    if (live) xn[n * D + d] = a / den;
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

// End of the tail kernels: the workgroup's 16 wave sums -> one partial per workgroup in `partials` (agent-scope store, drained),
// ONE ticket atomic per workgroup; the workgroup that draws the last ticket adds the partials in a fixed order (lane i takes
// partials i, i + 64, ...; halving tree) — the loss is bit-reproducible from run to run, which the earlier form (a double-
// precision atomicAdd per workgroup on one address, then the ticket: two serialised same-address atomics per workgroup, 4 us of
// a 10 us kernel at 196 workgroups) was not.  `sse`: the 16-byte zeroed scratch (ticket at sse + 1), left zeroed.
__device__ __forceinline__ void vqkd_tail_finish(double s, double *red, double *partials, double *sse, float *mse, int64_t N, int D) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __shared__ int last_s;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off, 64);
    if (lane == 0) red[wave] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
#pragma unroll
        for (int i = 0; i < 16; ++i) t += red[i];
        __hip_atomic_store(&partials[blockIdx.x], t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        last_s = (atomicAdd((int *)(sse + 1), 1) == (int)gridDim.x - 1) ? 1 : 0;
    }
    __syncthreads();
    if (last_s && wave == 0) {
        double t = 0.0;
        for (int i = lane; i < (int)gridDim.x; i += 64) t += __hip_atomic_load(&partials[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) t += __shfl_xor(t, off, 64);
        if (lane == 0) {
            const float mean = (float)(t / ((double)N * (double)D));
            mse[0] = mean; mse[1] = mean; mse[2] = 0.0f; mse[3] = 0.0f;
            __hip_atomic_store((int *)(sse + 1), 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// Tail of the VQ-KD forward, wave per token (grid-stride), 16 waves per workgroup and at most 256 workgroups (the double sum and
// the ticket are two same-address atomics per WORKGROUP: 1024 workgroups of 4 waves made this a 33 us kernel, 16 x 256 a 6 us one):
//   z = w[idx[n]];  z_ste = xn + (z - xn)  (utils/ste.py:10);  loss term: (F.normalize(z) - F.normalize(xn))^2
//   mse[0] = mean over N*D (CommitmentLoss with mse norm=True, losses.py:37,62), mse[1..3] = mse[0], 0, 0
// `sse`: the 16-byte zeroed scratch of gather_ste_loss_kernel (ticket), left zeroed; `partials`: >= gridDim.x doubles (vqkd_tail_finish).
__global__ __launch_bounds__(1024) void vqkd_tail_kernel(const float *__restrict__ xn,
                                                         const float *__restrict__ w, const int64_t *__restrict__ idx, int64_t N,
                                                         int D, float eps, float *__restrict__ z_ste, double *sse,
                                                         float *__restrict__ mse, double *__restrict__ partials) {
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
    vqkd_tail_finish(s, red, partials, sse, mse, N, D);
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

// The tail and the backward at D <= 32 (the shipped VQ-KD config has D = 32): L lanes per token, 64 / L tokens per wave, every
// element of a row in a register of its own — one trip to memory per token instead of two or three dependent ones by a wave whose
// upper lanes idle.  Same per-element expressions; the group sums are the full-wave trees without their all-zero upper levels
// (bit-identical gradients and straight-through outputs; the double-precision loss sum is regrouped, as it is between any two
// launches of the wave-per-token form: its cross-workgroup atomics already arrive in any order).
template <int L>
__device__ __forceinline__ float vq_group_sum(float p) {
#pragma unroll
    for (int off = L / 2; off >= 1; off >>= 1) p = p + __shfl_xor(p, off, 64);
    return p;
}
template <int L>
__global__ __launch_bounds__(1024) void vqkd_tail_small_kernel(const float *__restrict__ xn, const float *__restrict__ w,
                                                               const int64_t *__restrict__ idx, int64_t N, int D, float eps,
                                                               float *__restrict__ z_ste, double *sse, float *__restrict__ mse,
                                                               double *__restrict__ partials) {
    constexpr int RPW = 64 / L;
    __shared__ double red[16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int d = lane % L, rw = lane / L;
    double s = 0.0;
    for (int64_t base = ((int64_t)blockIdx.x * 16 + wave) * RPW; base < N; base += (int64_t)gridDim.x * 16 * RPW) {
        const int64_t n = base + rw;
        const bool live = n < N && d < D;
        const int64_t k = n < N ? idx[n] : 0;
        const float zv = live ? w[k * D + d] : 0.0f, xv = live ? xn[n * D + d] : 0.0f;
        const float nz = sqrtf(vq_group_sum<L>(fmaf(zv, zv, 0.0f))), nx = sqrtf(vq_group_sum<L>(fmaf(xv, xv, 0.0f)));
        const float den = (nz < eps) ? eps : nz, dnx = (nx < eps) ? eps : nx;
        if (live) {
            if (z_ste) z_ste[n * D + d] = xv + (zv - xv);
            const float df = zv / den - xv / dnx;
            s += (double)(df * df);
        }
    }
    vqkd_tail_finish(s, red, partials, sse, mse, N, D);
}

template <int DT, int L>
__global__ __launch_bounds__(256) void vqkd_backward_small_kernel(const void *__restrict__ x, const float *__restrict__ xn,
                                                                  const float *__restrict__ w, const int64_t *__restrict__ idx,
                                                                  int64_t N, int D, float eps, const float *__restrict__ g_zste,
                                                                  const float *__restrict__ g_loss, float *__restrict__ gx) {
    constexpr int RPW = 64 / L;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int d = lane % L, rw = lane / L;
    const float gl = g_loss ? *g_loss : 0.0f;
    const float sc = gl * (2.0f / ((float)N * (float)D));
    for (int64_t base = ((int64_t)blockIdx.x * 4 + wave) * RPW; base < N; base += (int64_t)gridDim.x * 4 * RPW) {
        const int64_t n = base + rw;
        const bool live = n < N && d < D;
        const int64_t k = n < N ? idx[n] : 0;
        const float a = live ? load_elem<DT>(x, n * D + d) : 0.0f, b = live ? xn[n * D + d] : 0.0f, c = live ? w[k * D + d] : 0.0f;
        const float gz = (live && g_zste) ? g_zste[n * D + d] : 0.0f;
        const float nx = sqrtf(vq_group_sum<L>(fmaf(a, a, 0.0f))), nn = sqrtf(vq_group_sum<L>(fmaf(b, b, 0.0f))),
                    nz = sqrtf(vq_group_sum<L>(fmaf(c, c, 0.0f)));
        const bool cx = nx < eps, cn = nn < eps;
        const float dx = cx ? eps : nx, dn = cn ? eps : nn, dz = (nz < eps) ? eps : nz;
        const float t = b / dn;
        const float gt = sc * (t - c / dz);
        const float dot1 = vq_group_sum<L>(fmaf(t, gt, 0.0f));
        const float g1 = cn ? gt / dn : (gt - t * dot1) / dn;
        const float gxn = g1 + gz;
        const float y = a / dx;
        const float dot2 = vq_group_sum<L>(live ? fmaf(y, gxn, 0.0f) : 0.0f);
        if (live) gx[n * D + d] = cx ? gxn / dx : (gxn - y * dot2) / dx;
    }
}

// ------------------------------------------------------------------------------------------------
// Backward of the quantizer call on the NCHW feature map, gradient of the latents written AS the map (round 5):
//   grad_map[b, d, p] = g_map[b, d, p] - kx * (e[idx[n]][d] - x_rows[n][d]),   n = b*hw + p,   kx = (g_cm + beta*g_comb) * 2/(N D)
// — vq_backward_kernel's grad_x with both rearrangements of vq/tasks/image_tokenization/models/base.py:124,126 folded in: the
// upstream gradient is read as the map it arrives in and the result is stored as the map the encoder's backward consumes, in the
// map's own dtype; round 4 transposed both through global memory (two vqhip_transpose launches and a cast around the kernel).
// Geometry of gather_ste_map256_kernel: a work item = (256 consecutive positions of one image, a share of the channels), 512
// threads, chunks of 32 channels turned through a swizzled LDS tile (two buffers, one barrier per chunk), codebook rows and
// latents requested one chunk ahead; every wave-load of g_map and wave-store of grad_map is one whole 1 KiB (fp32) channel row.
// Needs hw % 256 == 0 and D % 32 == 0.  The codebook gradient is not formed here (vqhip_vq_backward_ex with grad_x = NULL, or
// the ordered route).  ODT: 0 = fp32 map, 1 = bf16 map (round to nearest even).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t f32_to_bf16_bits(float v) {
    uint32_t b = __float_as_uint(v);
    if ((b & 0x7F800000u) == 0x7F800000u) return b >> 16;        // Inf / NaN: truncate (a NaN keeps a set mantissa bit or becomes Inf-safe)
    b += 0x7FFFu + ((b >> 16) & 1u);
    return b >> 16;
}

template <int DT, int ODT>
__global__ __launch_bounds__(512) void vq_backward_map256_kernel(const void *__restrict__ x, const float *__restrict__ e,
                                                                 const int64_t *__restrict__ idx, int64_t N, int D, int64_t hw,
                                                                 int csplit, const float *__restrict__ g_map,
                                                                 const float *__restrict__ g_cm, const float *__restrict__ g_comb,
                                                                 float beta, void *__restrict__ grad_map) {
    extern __shared__ __attribute__((aligned(16))) char bmap_lds[];
    float *tile = (float *)bmap_lds;                              // [2][32][256]
    __shared__ int64_t code_s[256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t ntiles = N / 256;
    const int nchunk = D / 32 / csplit;
    const float kx = ((g_cm ? *g_cm : 0.0f) + beta * (g_comb ? *g_comb : 0.0f)) * (2.0f / ((float)N * (float)D));
    for (int64_t item = blockIdx.x; item < ntiles * csplit; item += gridDim.x) {
        const int64_t tb = item / csplit;
        const int cbase = (int)(item % csplit) * nchunk * 32;
        const int64_t n0 = tb * 256;
        const int64_t obase = (n0 / hw) * (int64_t)D * hw + (n0 % hw);
        __syncthreads();
        if (threadIdx.x < 256) code_s[threadIdx.x] = idx[n0 + threadIdx.x];
        __syncthreads();
        float4 zc[4], zn[4];
        typename std::conditional<DT == 0, float4, uint2>::type xc[4], xnx[4];
        auto load_chunk = [&](int c0, float4 (&zr)[4], decltype(xc) &xr) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int q = threadIdx.x + 512 * i, tl = q >> 3, cp = q & 7;
                zr[i] = *(const float4 *)(e + code_s[tl] * D + c0 + 4 * cp);
                if constexpr (DT == 0) xr[i] = *(const float4 *)((const float *)x + (n0 + tl) * D + c0 + 4 * cp);
                else xr[i] = *(const uint2 *)((const uint16_t *)x + (n0 + tl) * D + c0 + 4 * cp);
            }
        };
        load_chunk(cbase, zc, xc);
        for (int c = 0; c < nchunk; ++c) {
            const int c0 = cbase + 32 * c;
            float *tb_lds = tile + (c & 1) * (32 * 256);
            // the upstream gradient of this chunk: requested before the barrier, consumed behind it
            float4 gv[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = wave + 8 * i;
                gv[i] = g_map ? *(const float4 *)(g_map + obase + (int64_t)(c0 + r) * hw + 4 * lane) : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            }
            if (c + 1 < nchunk) load_chunk(c0 + 32, zn, xnx);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int q = threadIdx.x + 512 * i, tl = q >> 3, cp = q & 7;
                const float4 zv = zc[i];
                float xv[4];
                if constexpr (DT == 0) { xv[0] = xc[i].x; xv[1] = xc[i].y; xv[2] = xc[i].z; xv[3] = xc[i].w; }
                else {
                    xv[0] = __uint_as_float(xc[i].x << 16); xv[1] = __uint_as_float(xc[i].x & 0xFFFF0000u);
                    xv[2] = __uint_as_float(xc[i].y << 16); xv[3] = __uint_as_float(xc[i].y & 0xFFFF0000u);
                }
                const float o[4] = {kx * (zv.x - xv[0]), kx * (zv.y - xv[1]), kx * (zv.z - xv[2]), kx * (zv.w - xv[3])};
                const int col = tl ^ (cp << 2);
#pragma unroll
                for (int j = 0; j < 4; ++j) tb_lds[(4 * cp + j) * 256 + col] = o[j];
            }
            __syncthreads();
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = wave + 8 * i;
                const float4 v = *(const float4 *)(tb_lds + r * 256 + 4 * (lane ^ ((r >> 2) & 7)));
                const float4 out = make_float4(gv[i].x - v.x, gv[i].y - v.y, gv[i].z - v.z, gv[i].w - v.w);
                const int64_t off = obase + (int64_t)(c0 + r) * hw + 4 * lane;
                if constexpr (ODT == 0) *(float4 *)((float *)grad_map + off) = out;
                else {
                    uint2 pk;
                    pk.x = f32_to_bf16_bits(out.x) | (f32_to_bf16_bits(out.y) << 16);
                    pk.y = f32_to_bf16_bits(out.z) | (f32_to_bf16_bits(out.w) << 16);
                    *(uint2 *)((uint16_t *)grad_map + off) = pk;
                }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) { zc[i] = zn[i]; xc[i] = xnx[i]; }
        }
    }
}
