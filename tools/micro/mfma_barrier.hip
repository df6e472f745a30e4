// What a workgroup barrier per stage costs a bare MFMA stream: 8 waves per workgroup (two per SIMD), one workgroup per CU, every wave
// issues M v_mfma_f32_16x16x32_f16 (8 independent accumulators, operands in registers) per stage; with and without __syncthreads()
// between stages.  Prints the time per stage in shader cycles next to the M x 16 x 2 cycles the two waves of a SIMD need for their MFMAs.
// build: hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_barrier.hip -o build/mfma_barrier ; run: build/mfma_barrier
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int M, int BAR>
__global__ __launch_bounds__(512, 2) void stage_loop(float *out, long stages, float seed, unsigned long long *cyc) {
    constexpr int A = 8;
    half8 a[4], b[4];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 8; ++j) { a[i][j] = (_Float16)(seed + threadIdx.x * 0.001f + j + i); b[i][j] = (_Float16)(seed * 0.5f + j - i); }
    f32x4 acc[A] = {};
    const unsigned long long c0 = clock64(), w0 = wall_clock64();
    for (long st = 0; st < stages; ++st) {
#pragma unroll
        for (int r = 0; r < M / A; ++r)
#pragma unroll
            for (int m = 0; m < A; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[r & 3], b[m & 3], acc[m], 0, 0, 0);
        if (BAR == 1) __syncthreads();
        if (BAR == 2) { if (st & 1) __syncthreads(); }
    }
    const unsigned long long c1 = clock64(), w1 = wall_clock64();
    float s = 0;
    for (int m = 0; m < A; ++m) s += acc[m][0];
    if (s == 123.456f) out[0] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) { cyc[0] = c1 - c0; cyc[1] = w1 - w0; }
}

template <int M, int BAR>
static void run(const char *what, float *out, unsigned long long *cyc) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    long stages = 3000000 / M;
    for (int pass = 0; pass < 3; ++pass) {
        CK(hipEventRecord(e0));
        stage_loop<M, BAR><<<256, 512>>>(out, stages, 1.0f, cyc);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        unsigned long long h[2]; CK(hipMemcpy(h, cyc, 16, hipMemcpyDeviceToHost));
        if (pass == 2)
            printf("%3d MFMAs per wave and stage, %-22s: %.3f ms, %6.0f TFLOP/s, clock %.0f MHz, %6.0f cycles per stage (MFMA work of a SIMD: %d)\n", M, what, ms,
                   256.0 * 8 * stages * M * 16384.0 / ms / 1e9, (double)h[0] / ((double)h[1] * 0.01), (double)h[0] / stages, M * 32);
    }
}

int main() {
    float *out; unsigned long long *cyc;
    CK(hipMalloc(&out, 4)); CK(hipMalloc(&cyc, 16));
    run<64, 0>("no barrier", out, cyc);   run<64, 1>("barrier per stage", out, cyc);   run<64, 2>("barrier per two stages", out, cyc);
    run<128, 0>("no barrier", out, cyc);  run<128, 1>("barrier per stage", out, cyc);  run<128, 2>("barrier per two stages", out, cyc);
    run<256, 0>("no barrier", out, cyc);  run<256, 1>("barrier per stage", out, cyc);
    return 0;
}
