// libvqhip device kernels, unit 6 of 8: gather / STE / loss (token-major and NCHW-map forms), histogram, scatter-add, VQ-KD and
// CVQ-VAE updates, elementwise autograd pieces, fused backward.  Included by vqhip_kernels.h.
#pragma once
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
// (Round 6 measured two more forms for bf16 latents on the headline's 805 MB — two or four rows in flight per wave with this access
//  pattern: 179-186 us against 178.6; eight consecutive elements per lane, half a wave per row of D = 256: 208-266 us, its 16-byte
//  accesses at a 32-byte stride cost more than the bytes in flight gain.  The kernel sits on a plateau of the memory system for a
//  1 : 2 read : write mix, not on a shortage of requests: profiles/r06_gather.txt.  Neither is kept.)
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

// gather_ste_map for maps whose images hold a multiple of 256 positions (16 x 16, 32 x 32, ...) and D % 32 == 0: a tile is
// 256 CONSECUTIVE positions of one image, so a chunk of 32 channels of it is ONE contiguous 32 KiB block of the output map
// (channel rows of 1 KiB, next to each other) — every wave-store is 1 KiB of one channel row, and DRAM sees the block being
// completed within the chunk's lifetime instead of 256-byte segments of 1024 tiles open across the whole launch.  Work item
// = (tile, share of the channels): 512 threads, 32 channels x 256 tokens per chunk through LDS, two chunk buffers (one
// barrier per chunk), the next chunk's codebook rows and latents requested before the current one is turned.
//   in:   thread -> (token tl = q >> 3, 16-byte piece cp = q & 7) for q = tid, tid + 512, ... : 8 lanes read the 128 bytes of
//         a token's codebook row (and 64 or 128 of its latent row) that belong to the chunk;
//   LDS:  tile[ch][256] with the token index XOR-ed by ((ch >> 2) & 7) << 2: the 4 values a thread writes and the
//         float4 a lane reads back are conflict-free (ds_write_b32 groups of 32 lanes, ds_read_b128 groups of 16);
//   out:  wave w, row r = w, w + 8, ...: lane l stores the float4 of tokens 4 l .. 4 l + 3.
template <int DT>
__global__ __launch_bounds__(512) void gather_ste_map256_kernel(const void *__restrict__ x, const float *__restrict__ e,
                                                                const int64_t *__restrict__ idx, int64_t N, int D, int64_t hw,
                                                                int csplit, float *__restrict__ out_map, double *sse,
                                                                float *mse, float beta) {
    extern __shared__ __attribute__((aligned(16))) char map_lds[];
    float *tile = (float *)map_lds;                               // [2][32][256]
    __shared__ double red[8];
    __shared__ int64_t code_s[256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t ntiles = N / 256;
    const int nchunk = D / 32 / csplit;                          // chunks per work item
    double s = 0.0;
    for (int64_t item = blockIdx.x; item < ntiles * csplit; item += gridDim.x) {
        const int64_t tb = item / csplit;
        const int cbase = (int)(item % csplit) * nchunk * 32;
        const int64_t n0 = tb * 256;
        const int64_t obase = (n0 / hw) * (int64_t)D * hw + (n0 % hw);         // + channel * hw + token of the tile
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
                if (x != nullptr) {
                    if constexpr (DT == 0) xr[i] = *(const float4 *)((const float *)x + (n0 + tl) * D + c0 + 4 * cp);
                    else xr[i] = *(const uint2 *)((const uint16_t *)x + (n0 + tl) * D + c0 + 4 * cp);
                }
            }
        };
        load_chunk(cbase, zc, xc);
        for (int c = 0; c < nchunk; ++c) {
            const int c0 = cbase + 32 * c;
            float *tb_lds = tile + (c & 1) * (32 * 256);
            if (c + 1 < nchunk) load_chunk(c0 + 32, zn, xnx);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int q = threadIdx.x + 512 * i, tl = q >> 3, cp = q & 7;
                const float4 zv = zc[i];
                float o[4];
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
                const int col = tl ^ (cp << 2);                   // (ch >> 2) & 7 == cp for ch = 4 cp + j
#pragma unroll
                for (int j = 0; j < 4; ++j) tb_lds[(4 * cp + j) * 256 + col] = o[j];
            }
            __syncthreads();
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = wave + 8 * i;
                const float4 v = *(const float4 *)(tb_lds + r * 256 + 4 * (lane ^ ((r >> 2) & 7)));
                *(float4 *)(out_map + obase + (int64_t)(c0 + r) * hw + 4 * lane) = v;
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) { zc[i] = zn[i]; xc[i] = xnx[i]; }
        }
    }
    if (sse) {
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off, 64);
        if (lane == 0) red[wave] = s;
        __syncthreads();
        if (threadIdx.x == 0) {
            const double t = ((red[0] + red[1]) + (red[2] + red[3])) + ((red[4] + red[5]) + (red[6] + red[7]));
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
    // grad_x: 16-byte accesses (4 consecutive floats per lane).  grad_w: float atomics want the 64 lanes of an instruction on 256
    // contiguous bytes (measured: 4 consecutive floats per lane is 3.5x slower), so the atomics run as a second sweep over the row
    // with one float per lane — its operands were loaded a moment ago and come from L1 / L2.  (Round 4 dropped to the one-float
    // form for BOTH outputs whenever the codebook gradient was wanted: quantize() forward + backward on a channels-last map
    // 0.674 -> see profiles/r05_shapes.txt.)
    const bool vec = (D % 4) == 0;
    for (int64_t n = (int64_t)blockIdx.x * 4 + wave; n < N; n += (int64_t)gridDim.x * 4) {
        const int64_t k = idx[n];
        if (vec) {
            if (grad_x)
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
                    float4 gz = make_float4(0, 0, 0, 0);
                    if (g_zste) gz = *(const float4 *)(g_zste + n * D + d);
                    *(float4 *)(grad_x + n * D + d) = make_float4(gz.x - kx * d0, gz.y - kx * d1, gz.z - kx * d2, gz.w - kx * d3);
                }
            if (do_w)
                for (int d = lane; d < D; d += 64) {
                    const float df = e[k * D + d] - load_elem<DT>(x, n * D + d);
                    atomicAdd(&grad_w[k * D + d], kw * df);
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
