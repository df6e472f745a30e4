// libvqhip device kernels, unit 7 of 8: deterministic (ordered) codebook-side sums.  Included by vqhip_kernels.h.
#pragma once
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
