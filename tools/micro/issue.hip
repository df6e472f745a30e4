// Micro-benchmark of the vector issue model used in DESIGN.md §4.3: cycles per loop iteration of a wave that issues M MFMAs
// (independent accumulators) and V v_max3_f32 fillers, for 1 / 2 / 4 waves per SIMD and both MFMA shapes.
// build: hipcc --offload-arch=gfx950 -O3 tools/micro/issue.hip -o build/issue ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int SHAPE, int M, int V>
__global__ void body(unsigned long long *out, int iters, float seed) {
    half8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (_Float16)(seed + threadIdx.x * 0.001f + j); b[j] = (_Float16)(seed * 0.5f + j); }
    f32x4 acc4[4] = {};
    f32x16 acc16[2] = {};
    float v0 = seed, v1 = seed + 1, v2 = seed + 2, v3 = seed + 3;
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < M; ++m) {
            if constexpr (SHAPE == 16) acc4[m & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc4[m & 3], 0, 0, 0);
            else acc16[m & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc16[m & 1], 0, 0, 0);
        }
#pragma unroll
        for (int k = 0; k < V; ++k) {
            if ((k & 1) == 0) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(v0) : "v"(v2), "v"(v3));
            else asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(v1) : "v"(v2), "v"(v3));
        }
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
    float sink = v0 + v1;
    for (int m = 0; m < 4; ++m) sink += acc4[m][0];
    for (int m = 0; m < 2; ++m) sink += acc16[m][0];
    if (sink == 123.456f) out[1] = 1;                      // keep everything alive
    // the SIMD favours its oldest wave: report the SLOWEST wave of workgroup 0 (and the fastest, out[2])
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) { atomicMax(&out[0], t1 - t0); atomicMin(&out[2], t1 - t0); }
}

template <int SHAPE, int M, int V>
static void run(const char *what, unsigned long long *dev) {
    const int iters = 2000;
    for (int waves_per_simd : {1, 2, 4}) {
        const int threads = 256 * waves_per_simd;                // one workgroup per CU, waves spread over its 4 SIMDs
        body<SHAPE, M, V><<<256, threads>>>(dev, iters, 1.0f);
        CK(hipDeviceSynchronize());
        unsigned long long init[3] = {0ull, 0ull, ~0ull}; CK(hipMemcpy(dev, init, 24, hipMemcpyHostToDevice));
        body<SHAPE, M, V><<<256, threads>>>(dev, iters, 1.0f);
        CK(hipDeviceSynchronize());
        unsigned long long h[3]; CK(hipMemcpy(h, dev, 24, hipMemcpyDeviceToHost));
        printf("%-14s M=%d V=%2d  waves/SIMD %d: slowest wave %8.2f cycles per iteration (fastest %7.2f) -> %7.2f per iteration and wave on the SIMD\n",
               what, M, V, waves_per_simd, (double)h[0] / iters, (double)h[2] / iters, (double)h[0] / iters / waves_per_simd);
    }
}

int main() {
    unsigned long long *dev; CK(hipMalloc(&dev, 64)); CK(hipMemset(dev, 0, 64));
    run<16, 4, 0>("16x16x32 f16", dev);  run<16, 4, 8>("16x16x32 f16", dev);  run<16, 4, 16>("16x16x32 f16", dev);  run<16, 4, 32>("16x16x32 f16", dev);
    run<32, 2, 0>("32x32x16 f16", dev);  run<32, 2, 8>("32x32x16 f16", dev);  run<32, 2, 12>("32x32x16 f16", dev); run<32, 2, 16>("32x32x16 f16", dev); run<32, 2, 32>("32x32x16 f16", dev);
    run<16, 0, 16>("VALU only", dev);
    return 0;
}
