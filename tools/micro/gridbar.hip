// Micro-benchmark: 4 dependent short kernels vs ONE kernel of 256 workgroups with 3 grid-wide barriers.
// build: hipcc --offload-arch=gfx950 -O3 tools/micro/gridbar.hip -o build/gridbar ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ void short_kernel(const int *cnt, int *out) {
    if (cnt[0] > (int)(blockIdx.x * blockDim.x + threadIdx.x)) out[threadIdx.x] = 1;     // nothing to do: cnt[0] == 0
}

// two-level barrier: GROUPS first-level counters (one 128-byte line each), one top counter; monotonic targets
struct Bar { int *lvl1; int *top; };
__device__ __forceinline__ bool grid_barrier(const Bar &b, int phase, int ngroups, int per_group) {
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        const int g = blockIdx.x % ngroups;
        const int prev = atomicAdd(&b.lvl1[g * 32], 1);
        if (prev == phase * per_group + per_group - 1) atomicAdd(b.top, 1);
        const int want = (phase + 1) * ngroups;
        long spins = 0;
        while (__hip_atomic_load(b.top, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1l << 22)) { ok = false; break; }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
    return ok;
}
__global__ __launch_bounds__(512) void fused_kernel(const int *cnt, int *out, Bar b, int ngroups, int nbar) {
    for (int p = 0; p < nbar; ++p) {
        if (cnt[0] > (int)(blockIdx.x * blockDim.x + threadIdx.x)) out[threadIdx.x] = 1;
        if (!grid_barrier(b, p, ngroups, gridDim.x / ngroups)) { if (threadIdx.x == 0) out[1023] = -1; return; }
    }
    if (cnt[0] > (int)(blockIdx.x * blockDim.x + threadIdx.x)) out[threadIdx.x] = 1;
}
__global__ void zero_kernel(int *p, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] = 0; }

int main() {
    int *cnt, *out, *bar;
    CK(hipMalloc(&cnt, 4096)); CK(hipMalloc(&out, 8192)); CK(hipMalloc(&bar, 64 * 128 + 128));
    CK(hipMemset(cnt, 0, 4096)); CK(hipMemset(out, 0, 8192));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 300;
    float ms;
    for (int grid : {256, 1024}) {
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipEventRecord(e0));
            for (int i = 0; i < iters; ++i)
                for (int k = 0; k < 4; ++k) short_kernel<<<grid, 512>>>(cnt, out);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        }
        printf("4 dependent short kernels, grid %d: %.2f us per group of 4\n", grid, ms * 1e3 / iters);
    }
    for (int ngroups : {1, 8, 16, 32}) {
        for (int nbar : {3, 1}) {
            for (int rep = 0; rep < 2; ++rep) {
                CK(hipEventRecord(e0));
                for (int i = 0; i < iters; ++i) {
                    zero_kernel<<<9, 256>>>(bar, 64 * 32 + 32);        // (in the product another kernel of the chain zeroes the words)
                    Bar b{bar, bar + 64 * 32};
                    fused_kernel<<<256, 512>>>(cnt, out, b, ngroups, nbar);
                }
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
            }
            printf("zero + fused kernel (256 WGs, %d barriers, %d groups): %.2f us\n", nbar, ngroups, ms * 1e3 / iters);
        }
    }
    CK(hipEventRecord(e0));
    for (int i = 0; i < iters; ++i) zero_kernel<<<9, 256>>>(bar, 64 * 32 + 32);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
    printf("zero kernel alone: %.2f us\n", ms * 1e3 / iters);
    int h[2048]; CK(hipMemcpy(h, out, 8192, hipMemcpyDeviceToHost));
    printf("timeout flag: %d\n", h[1023]);
    return 0;
}
