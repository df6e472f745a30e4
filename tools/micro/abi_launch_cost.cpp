// Host cost of ONE library entry point that is one small launch (vqhip_hist, vqhip_normalize_rows), next to a bare kernel launch of
// the same process: what the C ABI's argument checks, launch helpers and error polling add to hipLaunchKernel.
// build: hipcc --offload-arch=gfx950 -O3 -Iinclude tools/micro/abi_launch_cost.cpp -Lvector_quantization_amd -lvqhip -Wl,-rpath,$PWD/vector_quantization_amd -o build/abi_launch_cost
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include "vqhip.h"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
__global__ void touch(int *p) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1; }
int main() {
    const int64_t N = 1024, K = 1024; const int D = 32;
    int64_t *idx; int32_t *hist; float *x, *y; int *buf;
    CK(hipMalloc(&idx, N * 8)); CK(hipMemset(idx, 0, N * 8)); CK(hipMalloc(&hist, K * 4)); CK(hipMemset(hist, 0, K * 4));
    CK(hipMalloc(&x, N * D * 4)); CK(hipMemset(x, 0, N * D * 4)); CK(hipMalloc(&y, N * D * 4)); CK(hipMalloc(&buf, 64)); CK(hipMemset(buf, 0, 64));
    hipStream_t s; CK(hipStreamCreate(&s));
    const int n = 3000;
    for (int rep = 0; rep < 3; ++rep) {
        auto run = [&](const char *what, auto fn) {
            CK(hipStreamSynchronize(s));
            const auto t0 = std::chrono::steady_clock::now();
            for (int i = 0; i < n; ++i) fn();
            const auto t1 = std::chrono::steady_clock::now();
            CK(hipStreamSynchronize(s));
            if (rep == 2) printf("%-28s %.2f us of host time per call\n", what, std::chrono::duration<double, std::micro>(t1 - t0).count() / n);
        };
        run("bare kernel launch", [&] { touch<<<1, 64, 0, s>>>(buf); });
        run("vqhip_hist", [&] { if (vqhip_hist(idx, N, K, hist, s)) exit(2); });
        run("vqhip_normalize_rows", [&] { if (vqhip_normalize_rows(x, VQHIP_DTYPE_F32, N, D, 1e-12f, y, s)) exit(2); });
    }
    return 0;
}
