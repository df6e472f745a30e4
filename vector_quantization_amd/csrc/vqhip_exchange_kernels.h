// Codebook-update kernels around the ONE exchange step of a training forward: the CVQ-VAE sparse anchor
// set, the packed all-reduce buffer (code counts as exactly-summable fp32 pairs next to the fp32 payload), and the update
// that consumes it.  gfx950 only.  Reference: vq/algorithms/cvqvae/quantizer_callback.py:85-103, anchors.py:50-67,83-84,
// vq/algorithms/vq/utils.py:26-52, vq/algorithms/vqkd/quantizers/callbacks.py:44-71.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// ------------------------------------------------------------------------------------------------
// Packed exchange buffer (fp32, one SUM all-reduce):
//   [0, K)       low 16 bits of this rank's code counts        [K, 2K)  the bits above
//   [2K, 2K+3)   token count in 16-bit pieces                   [2K+3]   zero (keeps the payload 16-byte aligned)
//   [2K+4, ...)  payload rows [M, D] fp32 (anchors / centroid sums)
// Every count piece is an integer below 2^16, so sums over up to 256 ranks stay below 2^24 and are EXACT in fp32 in any
// order — the histogram of vq/algorithms/vq/utils.py:34-35 survives the trip through a float collective bit for bit.
// ------------------------------------------------------------------------------------------------
#define VQ_PACK_HEADER(K) (2 * (int64_t)(K) + 4)
#define VQ_PACK_MAX_WORLD 256

__device__ __forceinline__ int64_t unpack_count(const float *packed, int64_t K, int64_t k) {
    return (int64_t)packed[K + k] * 65536 + (int64_t)packed[k];
}
__device__ __forceinline__ int64_t unpack_numel(const float *packed, int64_t K) {
    return ((int64_t)packed[2 * K + 2] * 65536 + (int64_t)packed[2 * K + 1]) * 65536 + (int64_t)packed[2 * K];
}

// header from an int32 (HT = 0) or int64 (HT = 1) histogram
template <int HT>
__global__ void pack_counts_kernel(const void *hist, int64_t numel, int64_t K, float *packed) {
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k < K) {
        const int64_t h = HT ? ((const int64_t *)hist)[k] : (int64_t)((const int32_t *)hist)[k];
        packed[k] = (float)(h & 0xFFFF);
        packed[K + k] = (float)(h >> 16);
    }
    if (k == 0) {
        packed[2 * K] = (float)(numel & 0xFFFF);
        packed[2 * K + 1] = (float)((numel >> 16) & 0xFFFF);
        packed[2 * K + 2] = (float)(numel >> 32);
        packed[2 * K + 3] = 0.0f;
    }
}
// the all-reduced header back as int64 [K + 1] = counts ‖ token count
__global__ void unpack_counts_kernel(const float *packed, int64_t K, int64_t *out) {
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k < K) out[k] = unpack_count(packed, K, k);
    if (k == 0) out[K] = unpack_numel(packed, K);
}

// ------------------------------------------------------------------------------------------------
// CVQ-VAE: which codes can need an anchor at all
// ------------------------------------------------------------------------------------------------
// quantizer_callback.py:97-102: decay_k = 1 - exp(-p_k K 10/(1-g) - eps), w_k = w_k decay_k + a_k (1 - decay_k).
// For every code in regular use decay_k is EXACTLY 1.0f and its anchor is multiplied by 0.  The coming probability
// p' = p g + freq (1-g) is not known before the histogram exchange, but p' >= fl(p g) (freq >= 0; every rounding involved
// is monotone), and the exponent is a monotone function of p' — so a code whose exponent, evaluated at fl(p g) with the
// update's own expression, is already <= -VQ_CVQ_SURE_ARG has decay == 1.0f whatever this step's histogram turns out to be:
// 1 - exp(-20) rounds to 1.0f with a factor 14 to spare (exp(-20) = 2.1e-9 against the rounding boundary 2^-25 = 3.0e-8),
// far beyond anything expf's last-bit behaviour could move.  The set depends on the SYNCHRONISED p only: identical on
// every rank, known before the exchange.  NaN / negative p: kept (the comparison fails).
#define VQ_CVQ_SURE_ARG 20.0f
__device__ __forceinline__ bool cvq_may_need_anchor(float p_old, int64_t K, float ema_decay, float eps) {
    const float lower = p_old * ema_decay;
    const float arg = -lower * (float)K * 10.0f / (1.0f - ema_decay) - eps;
    return !(arg <= -VQ_CVQ_SURE_ARG);
}

// rows[0..count) = those codes in ascending order, slot[k] = position of code k in rows or -1, count[0] = their number.
// One 1024-thread workgroup; thread t owns the consecutive codes [t*per, (t+1)*per), per = ceil(K / 1024): flags in a
// register, a wave scan of the per-thread counts, a 16-entry scan of the wave totals — two barriers in all (3 us at
// K = 16 384; a round-per-1024-codes form with three barriers per round took 14).
// count_host (nullable): a pinned HOST word that receives the count as well (system-scope store; the host reads it behind an
// event recorded after this launch).
__global__ __launch_bounds__(1024) void cvq_rows_kernel(const float *__restrict__ p, int64_t K, float ema_decay, float eps,
                                                        int32_t *__restrict__ rows, int32_t *__restrict__ slot,
                                                        int32_t *__restrict__ count, int32_t *count_host = nullptr) {
    __shared__ int wtot[16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t per = (K + 1023) / 1024;
    const int64_t k0 = (int64_t)threadIdx.x * per;
    // the thread's flags as a bit mask (per <= 64, i.e. K <= 65 536; beyond that the probabilities are read twice): p is read
    // ONCE, 16 bytes at a time where the run is aligned — the first form read it element by element in both loops and took
    // 23 us of a 300 us step at K = 16 384 (profiles/r03_cvq_timeline_*.txt)
    const bool masked = per <= 64;
    unsigned long long bits = 0;
    int mine = 0;
    if (masked && (per % 4) == 0 && k0 + per <= K) {
        for (int64_t i = 0; i < per; i += 4) {
            const float4 v = *(const float4 *)(p + k0 + i);
            bits |= (unsigned long long)(cvq_may_need_anchor(v.x, K, ema_decay, eps) ? 1 : 0) << i;
            bits |= (unsigned long long)(cvq_may_need_anchor(v.y, K, ema_decay, eps) ? 1 : 0) << (i + 1);
            bits |= (unsigned long long)(cvq_may_need_anchor(v.z, K, ema_decay, eps) ? 1 : 0) << (i + 2);
            bits |= (unsigned long long)(cvq_may_need_anchor(v.w, K, ema_decay, eps) ? 1 : 0) << (i + 3);
        }
        mine = __popcll(bits);
    } else {
        for (int64_t k = k0; k < k0 + per && k < K; ++k) {
            const bool f = cvq_may_need_anchor(p[k], K, ema_decay, eps);
            if (masked && f) bits |= 1ull << (k - k0);
            mine += f ? 1 : 0;
        }
    }
    int incl = mine;                                      // inclusive scan over the wave
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int v = __shfl_up(incl, off, 64);
        if (lane >= off) incl += v;
    }
    if (lane == 63) wtot[wave] = incl;
    __syncthreads();
    int base = 0, total = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) { if (i < wave) base += wtot[i]; total += wtot[i]; }
    int pos = base + incl - mine;
    if (masked && (per % 4) == 0 && k0 + per <= K) {          // slot written 16 bytes at a time as well
        for (int64_t i = 0; i < per; i += 4) {
            int sv[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const bool f = ((bits >> (i + q)) & 1ull) != 0;
                sv[q] = f ? pos : -1;
                if (f) { rows[pos] = (int32_t)(k0 + i + q); ++pos; }
            }
            *(int4 *)(slot + k0 + i) = make_int4(sv[0], sv[1], sv[2], sv[3]);
        }
        if (threadIdx.x == 0) {
            count[0] = total;
            if (count_host != nullptr) __hip_atomic_store(count_host, total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        return;
    }
    for (int64_t k = k0; k < k0 + per && k < K; ++k) {
        const bool f = masked ? ((bits >> (k - k0)) & 1ull) != 0 : cvq_may_need_anchor(p[k], K, ema_decay, eps);
        if (f) { rows[pos] = (int32_t)k; slot[k] = pos; ++pos; }
        else slot[k] = -1;
    }
    if (threadIdx.x == 0) {
        count[0] = total;
        if (count_host != nullptr) __hip_atomic_store(count_host, total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// The LENGTH of the next step's list as soon as this step's histogram is final — long before cvq_rows_kernel can build the list
// itself from the updated probabilities (that needs the column pass and the update first).  p' is the update's own expression
// (cvq_apply_kernel: bit-identical), the test is the list's own.  One workgroup.  The count goes to a pinned HOST word together
// with a sequence number the kernel advances on the device ({seq, count} as ONE 64-bit system-scope store): a host that replays
// this step from a HIP graph — where no event can be recorded in the middle — polls the word for the sequence number it expects
// and picks the next replay's capacity while the rest of this one is still running (graphs.GraphedQuantizer).
template <bool PACKED>
__global__ __launch_bounds__(1024) void cvq_count_next_kernel(const float *__restrict__ p_in, const int32_t *__restrict__ hist,
                                                              int64_t numel, const float *__restrict__ packed, int64_t K,
                                                              float ema_decay, float eps, int32_t *seq_dev,
                                                              unsigned long long *word_host) {
    __shared__ int wtot[16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float fn = PACKED ? (float)unpack_numel(packed, K) : (float)numel;
    int mine = 0;
    for (int64_t k = threadIdx.x; k < K; k += 1024) {
        const float freq = (PACKED ? (float)unpack_count(packed, K, k) : (float)hist[k]) / fn;
        const float pk = p_in[k] * ema_decay + freq * (1.0f - ema_decay);
        mine += cvq_may_need_anchor(pk, K, ema_decay, eps) ? 1 : 0;
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) mine += __shfl_xor(mine, off, 64);
    if (lane == 0) wtot[wave] = mine;
    __syncthreads();
    if (threadIdx.x == 0) {
        int total = 0;
#pragma unroll
        for (int i = 0; i < 16; ++i) total += wtot[i];
        const int seq = seq_dev[0] + 1;
        seq_dev[0] = seq;
        __hip_atomic_store(word_host, ((unsigned long long)(uint32_t)seq << 32) | (unsigned long long)(uint32_t)total, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// out[i] = e[rows[i]] for i < count, zeros up to cap (the role-swapped pipeline reads whole 32-row blocks)
__global__ void gather_listed_rows_kernel(const float *__restrict__ e, const int32_t *__restrict__ rows,
                                          const int32_t *__restrict__ count, int64_t cap, int D, float *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (i >= cap) return;
    const bool live = i < (int64_t)count[0];
    const int64_t k = live ? rows[i] : 0;
    for (int d = lane; d < D; d += 64) out[i * D + d] = live ? e[k * D + d] : 0.0f;
}

// Packed buffer of one rank for the CVQ-VAE exchange: header from the epilogue histogram, payload row i = the anchor
// x[col_idx[i]] of listed code i (NearestAnchor, anchors.py:83-84) for i < count, zeros up to cap.
template <int DT>
__global__ void cvq_pack_kernel(const int32_t *__restrict__ hist, int64_t numel, const void *__restrict__ x,
                                const int64_t *__restrict__ col_idx, const int32_t *__restrict__ count, int64_t cap,
                                int64_t K, int D, float *__restrict__ packed, int header_blocks) {
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
    const int64_t i = (int64_t)(blockIdx.x - header_blocks) * (blockDim.x / 64) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (i >= cap) return;
    float *dst = packed + VQ_PACK_HEADER(K) + i * D;
    if (i < (int64_t)count[0]) {
        const int64_t row = col_idx[i];
        for (int d = lane; d < D; d += 64) dst[d] = load_elem<DT>(x, row * D + d);
    } else {
        for (int d = lane; d < D; d += 64) dst[d] = 0.0f;
    }
}

// ------------------------------------------------------------------------------------------------
// NearestAnchor(sync=True) over more than one rank (anchors.py:50-57,83-84): the reference all-gathers the latents AND the
// [N, K] matrix and takes the column argmin of the concatenation.  Here every rank runs the column pass over its OWN tokens,
// and the ranks agree on the winner per listed code through one MIN all-reduce of 8-byte keys:
//   key = (distance of the definition, in torch.argmin's order: NaN first, -0 == +0) : rank (8 bits) : row (24 bits)
// — the smallest key is the smallest distance, and among equal distances the lowest (rank, row), i.e. the lowest index of
// the reference's concatenation `torch.cat(all_gather(d))`.  The keys travel as int64 with the top bit flipped, so that the
// SIGNED minimum every backend offers (gloo, RCCL) orders them as the unsigned values; slots past the count hold INT64_MAX.
// ------------------------------------------------------------------------------------------------
#define VQ_SYNC_ROW_BITS 24
#define VQ_SYNC_MAX_ROWS (1ll << VQ_SYNC_ROW_BITS)
// x / e: the operands of the definition (L2: the latents as given and the codebook; cosine: both normalised — what the column
// pass itself was given).  One lane per listed code: the k-ordered fma chain of the definition is sequential by nature, and
// a list is a few dozen codes in the steady state (every code in the first steps: ~30 us at K = 16 384, D = 256).
template <int DT>
__global__ __launch_bounds__(64) void cvq_col_keys_kernel(const void *__restrict__ x, const float *__restrict__ e,
                                                          const int32_t *__restrict__ rows, const int32_t *__restrict__ count,
                                                          int64_t cap, const int64_t *__restrict__ col_idx, int D, int metric,
                                                          int rank, int64_t *__restrict__ keys) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= cap) return;
    if (i >= (int64_t)count[0]) { keys[i] = INT64_MAX; return; }
    const int64_t k = rows[i], n = col_idx[i];
    float xn = 0.0f, en = 0.0f;
    if (VQ_IS_L2(metric)) {
        xn = sqnorm_thread<DT>(x, n * D, D);
        en = sqnorm_thread<0>(e, k * D, D);
    }
    const float d = oracle_distance<DT>(x, n * D, e + k * D, D, metric, xn, en);
    const u64 key = dist_key(d, ((uint32_t)rank << VQ_SYNC_ROW_BITS) | (uint32_t)n);
    keys[i] = (int64_t)(key ^ 0x8000000000000000ull);
}

// The packed buffer of one rank behind the MIN all-reduce of the keys: header as cvq_pack_kernel; payload row i = x[row] on
// the rank the reduced key names, -0.0f on every other rank (x + (-0) == x bit for bit for every x, -0 included: the SUM
// all-reduce then delivers the winner's row exactly, in any order), zeros past the count.
template <int DT>
__global__ void cvq_pack_sync_kernel(const int32_t *__restrict__ hist, int64_t numel, const void *__restrict__ x,
                                     const int64_t *__restrict__ keys, const int32_t *__restrict__ count, int64_t cap, int rank,
                                     int64_t K, int D, float *__restrict__ packed, int header_blocks) {
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
    const int64_t i = (int64_t)(blockIdx.x - header_blocks) * (blockDim.x / 64) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (i >= cap) return;
    float *dst = packed + VQ_PACK_HEADER(K) + i * D;
    if (i < (int64_t)count[0]) {
        const uint32_t who = (uint32_t)((u64)keys[i] & 0xFFFFFFFFull);
        if ((int)(who >> VQ_SYNC_ROW_BITS) == rank) {
            const int64_t row = who & (uint32_t)(VQ_SYNC_MAX_ROWS - 1);
            for (int d = lane; d < D; d += 64) dst[d] = load_elem<DT>(x, row * D + d);
        } else {
            for (int d = lane; d < D; d += 64) dst[d] = -0.0f;
        }
    } else {
        for (int d = lane; d < D; d += 64) dst[d] = 0.0f;
    }
}

// The CVQ-VAE update with anchors for the listed codes only, wave per code — the expressions of cvq_update_kernel /
// cvq_step_kernel in the same order (bit-identical results on finite data):
//   p' = p g + (hist/numel)(1-g);  decay = 1 - exp(-p' K 10/(1-g) - eps);  w' = w decay + a (1-decay)   for listed codes,
//   w' = w decay (decay == 1.0f by construction of the list)                                             for the others.
// PACKED: counts, token count and the anchor SUMS over `world` ranks come from the all-reduced buffer (a = sum / world:
// anchors.py:65-67); otherwise one rank: counts from the int32 epilogue histogram, a = x[col_idx[slot]].
template <int DT, bool PACKED>
__global__ void cvq_apply_kernel(const float *w_in, float *w_out, const float *p_in, float *p_out, const int32_t *hist,
                                 int64_t numel, const void *x, const int64_t *col_idx, const float *packed, int world,
                                 const int32_t *__restrict__ slot, int cap, int64_t K, int D, float ema_decay, float eps) {
    const int64_t k = (int64_t)blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (k >= K) return;
    float freq;
    if constexpr (PACKED) freq = (float)unpack_count(packed, K, k) / (float)unpack_numel(packed, K);
    else freq = (float)hist[k] / (float)numel;
    const float pk = p_in[k] * ema_decay + freq * (1.0f - ema_decay);
    const float decay = cvq_decay_of(pk, K, ema_decay, eps), om = 1.0f - decay;
    // a slot at or beyond the capacity the column pass and the pack were sized for has no anchor anywhere (a list longer than
    // the caller's capacity: include/vqhip.h): the code keeps w * decay, nothing outside col_idx / packed is read
    const int s = slot[k];
    if (s >= 0 && s < cap) {
        if constexpr (PACKED) {
            const float *a = packed + VQ_PACK_HEADER(K) + (int64_t)s * D;
            const float ws = (float)world;
            for (int d = lane; d < D; d += 64) w_out[k * D + d] = w_in[k * D + d] * decay + (a[d] / ws) * om;
        } else {
            const int64_t row = col_idx[s];
            for (int d = lane; d < D; d += 64) w_out[k * D + d] = w_in[k * D + d] * decay + load_elem<DT>(x, row * D + d) * om;
        }
    } else if (w_out != w_in || decay != 1.0f) {
        for (int d = lane; d < D; d += 64) w_out[k * D + d] = w_in[k * D + d] * decay;
    }
    if (lane == 0) p_out[k] = pk;
}
