// Would the D = 256 proposal loop gain from v_mfma_f32_32x32x16_f16 (32 cycles, holds the vector issue for 8 of them) instead of
// v_mfma_f32_16x16x32_f16 (16 cycles, 8 of them)?  Both loops in registers, no LDS, no memory, random operands, two waves per SIMD
// on every CU — one code tile of 32 codes against 64 tokens per wave and iteration, the per-score epilogue of coarse_kernel (id bits
// into the mantissa, v_med3 for the runner-up, v_max for the best: 3 vector instructions per score) on the PREVIOUS tile's scores:
//   shape 0  16x16x32: 64 MFMAs per tile (2 code halves x 4 token tiles x 8 k-steps), 64 scores per lane
//   shape 1  32x32x16: 32 MFMAs per tile (2 token tiles x 16 k-steps), 64 scores per lane (2 x 32 of the 32x32 outputs... 16 each)
// EPI 0: no epilogue (accumulators kept alive), EPI 1: the epilogue.  Prints cycles per tile per SIMD (both waves), matrix-pipe busy,
// clock and PFLOP/s.   build: hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_shape_epilogue.hip -o build/mfma_shape_epilogue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float vmax(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }

template <int SHAPE, int EPI>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void loop(float *out, long tiles, float seed, unsigned long long *cyc) {
    constexpr int NA = 4;                       // A fragment sets cycled through (as if freshly read from LDS)
    half8 xf[4][8];                             // 64 tokens x 256 dims of B fragments: 128 registers, as in the kernel
    half8 af[NA];
    unsigned h = (blockIdx.x * 1024u + threadIdx.x) * 2654435761u + 12345u;
    auto rnd = [&]() { h = h * 1664525u + 1013904223u; return (_Float16)(((int)(h >> 9) % 2001 - 1000) * (1.0f / 1024.0f) * seed); };
    for (int j = 0; j < 8; ++j) {
        for (int t = 0; t < 4; ++t) for (int s = 0; s < 8; ++s) xf[t][s][j] = rnd();
        for (int a = 0; a < NA; ++a) af[a][j] = rnd();
    }
    float b1[4], b2[4];
    for (int t = 0; t < 4; ++t) { b1[t] = -1e30f; b2[t] = -1e30f; }
    f32x4 a16A[2][4], a16B[2][4];
    f32x16 a32A[2], a32B[2];
    for (int c = 0; c < 2; ++c) for (int t = 0; t < 4; ++t) for (int q = 0; q < 4; ++q) { a16A[c][t][q] = 0.f; a16B[c][t][q] = -1.f; }
    for (int t = 0; t < 2; ++t) for (int q = 0; q < 16; ++q) { a32A[t][q] = 0.f; a32B[t][q] = -1.f; }
    __syncthreads();
    const unsigned long long c0 = clock64(), w0 = wall_clock64();
    for (long it = 0; it < tiles; it += 2) {
#pragma unroll
        for (int par = 0; par < 2; ++par) {
            if (SHAPE == 0) {
                f32x4 (&cur)[2][4] = par ? a16B : a16A;
                f32x4 (&prv)[2][4] = par ? a16A : a16B;
#pragma unroll
                for (int ch = 0; ch < 16; ++ch) {          // chunk = (k-step ch >> 1, code half ch & 1): one A fragment, four MFMAs
                    half8 &a = af[ch % NA];
                    asm volatile("" : "+v"(a));
#pragma unroll
                    for (int t = 0; t < 4; ++t)
                        cur[ch & 1][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, xf[t][ch >> 1], ch < 2 ? f32x4{0.f, 0.f, 0.f, 0.f} : cur[ch & 1][t], 0, 0, 0);
                    if (EPI) {                              // retire 2 scores of the previous tile per chunk step and token tile... 32 per chunk pair
#pragma unroll
                        for (int i = 0; i < 2; ++i) {
                            const int id = ch * 2 + i, t = id / 8, e = id % 8;       // 32 (t, e) pairs over 16 chunks: 4 tiles x 8 elements
                            const float v = __uint_as_float((__float_as_uint(prv[e >> 2][t][e & 3]) & 0xFFFFFFF0u) | (unsigned)e);
                            b2[t] = __builtin_amdgcn_fmed3f(b1[t], b2[t], v);
                            b1[t] = vmax(b1[t], v);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (!EPI) { for (int t = 0; t < 4; ++t) asm volatile("" :: "v"(prv[0][t]), "v"(prv[1][t])); }
            } else {
                f32x16 (&cur)[2] = par ? a32B : a32A;
                f32x16 (&prv)[2] = par ? a32A : a32B;
#pragma unroll
                for (int s = 0; s < 16; ++s) {             // k-step of 16 dims: one A fragment (32 codes x 16 dims), two MFMAs (two wide token tiles)
                    half8 &a = af[s % NA];
                    asm volatile("" : "+v"(a));
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        // the wide tile's B fragment of k-step s: tokens 32 t .. + 32, dims 16 s .. + 16 — one of the 64 register sets
                        const half8 &b = xf[2 * t + (s & 1)][s >> 1];
                        if (s == 0) { f32x16 z; for (int q = 0; q < 16; ++q) z[q] = 0.f; cur[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, z, 0, 0, 0); }
                        else cur[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, cur[t], 0, 0, 0);
                    }
                    if (EPI) {
#pragma unroll
                        for (int i = 0; i < 2; ++i) {
                            const int id = s * 2 + i, t = id / 16, e = id % 16;      // 32 (t, e) pairs: 2 wide tiles x 16 elements
                            const float v = __uint_as_float((__float_as_uint(prv[t][e]) & 0xFFFFFFF0u) | (unsigned)e);
                            b2[t] = __builtin_amdgcn_fmed3f(b1[t], b2[t], v);
                            b1[t] = vmax(b1[t], v);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (!EPI) asm volatile("" :: "v"(prv[0]), "v"(prv[1]));
            }
        }
    }
    const unsigned long long c1 = clock64(), w1 = wall_clock64();
    float s = 0;
    for (int t = 0; t < 4; ++t) s += b1[t] + b2[t];
    s += a16A[0][0][0] + a16B[1][3][2] + a32A[0][0] + a32B[1][7];
    if (s == 123.456f) out[0] = s;
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) { atomicMin(&cyc[2], c0); atomicMax(&cyc[3], c1); atomicMin(&cyc[4], w0); atomicMax(&cyc[5], w1); }
}

template <int SHAPE, int EPI>
static void run(const char *what, float *out, unsigned long long *cyc) {
    const long tiles = 60000;                  // ~3 ms launches
    for (int pass = 0; pass < 3; ++pass) {
        unsigned long long init[6] = {0, 0, ~0ull, 0, ~0ull, 0};
        CK(hipMemcpy(cyc, init, 48, hipMemcpyHostToDevice));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0));
        loop<SHAPE, EPI><<<256, 512>>>(out, tiles, 1.0f, cyc);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        unsigned long long h[6]; CK(hipMemcpy(h, cyc, 48, hipMemcpyDeviceToHost));
        if (pass == 2) {
            const double cyc_tile = (double)(h[3] - h[2]) / tiles;      // per tile-step of a SIMD: its two waves
            const double flops = 256.0 * 8 * tiles * 32.0 * 64 * 256 * 2;
            printf("%-34s: %.3f ms, %7.1f cycles per tile-step of a SIMD (MFMA cycles of its two waves: 2048): pipe %.1f %% busy, clock %.0f MHz, %.3f PFLOP/s\n",
                   what, ms, cyc_tile, 204800.0 / cyc_tile, (double)(h[3] - h[2]) / ((double)(h[5] - h[4]) * 0.01), flops / ms / 1e12);
        }
    }
}

int main() {
    float *out; unsigned long long *cyc;
    CK(hipMalloc(&out, 4)); CK(hipMalloc(&cyc, 48));
    run<0, 0>("16x16x32, no epilogue", out, cyc);
    run<0, 1>("16x16x32, per-score epilogue", out, cyc);
    run<1, 0>("32x32x16, no epilogue", out, cyc);
    run<1, 1>("32x32x16, per-score epilogue", out, cyc);
    return 0;
}
