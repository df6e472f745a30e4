// Sustained rate of bare v_mfma_f32_16x16x32_f16 on the whole chip (operands in registers, W waves per SIMD, A independent
// accumulators per wave) over a few milliseconds: the clock the chip holds under MFMA load and the FLOP/s that go with it —
// what the nominal 2.5 PFLOP/s (2.4 GHz) of bench.py's roofline turns into when nothing but the matrix pipe runs.
// build: hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_sustained.hip -o build/mfma_sustained ; run: build/mfma_sustained [waves per SIMD] [ms]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int A>
__global__ void mfma_loop(float *out, long iters, float seed, unsigned long long *cyc) {
    half8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (_Float16)(seed + threadIdx.x * 0.001f + j); b[j] = (_Float16)(seed * 0.5f + j); }
    f32x4 acc[A] = {};
    const unsigned long long c0 = clock64(), w0 = wall_clock64();
    for (long it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int m = 0; m < A; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[m], 0, 0, 0);
    }
    const unsigned long long c1 = clock64(), w1 = wall_clock64();
    float s = 0;
    for (int m = 0; m < A; ++m) s += acc[m][0];
    if (s == 123.456f) out[0] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) { cyc[0] = c1 - c0; cyc[1] = w1 - w0; }
}

int main(int argc, char **argv) {
    const int wps = argc > 1 ? atoi(argv[1]) : 2;
    const double target_ms = argc > 2 ? atof(argv[2]) : 3.0;
    float *out; unsigned long long *cyc;
    CK(hipMalloc(&out, 4)); CK(hipMalloc(&cyc, 16));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    constexpr int A = 4;
    const int grid = 256, threads = wps * 4 * 64;                         // one workgroup per CU, wps waves on each of its 4 SIMDs
    long iters = 2000;
    for (int pass = 0; pass < 5; ++pass) {
        CK(hipEventRecord(e0));
        mfma_loop<A><<<grid, threads>>>(out, iters, 1.0f, cyc);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        unsigned long long h[2]; CK(hipMemcpy(h, cyc, 16, hipMemcpyDeviceToHost));
        const double flops = (double)grid * (threads / 64) * iters * 4 * A * 16384.0;
        printf("%d waves per SIMD, %ld iterations: %.3f ms, %.0f TFLOP/s, shader clock %.0f MHz\n", wps, iters, ms, flops / ms / 1e9,
               (double)h[0] / ((double)h[1] * 0.01));
        iters = (long)(iters * target_ms / ms);                           // aim the next pass at the target duration
    }
    return 0;
}
