// What one more launch costs a short training step: a chain of N dependent, (almost) empty kernels on one stream —
//   host: seconds per hipLaunchKernel call while the queue is never empty (the issue cost a step pays per launch),
//   device: events around the chain when the host is far ahead (the floor a dependent kernel adds on the GPU: dispatch, the
//           barrier between dependent packets, cache maintenance at the boundaries), eagerly and replayed from one HIP graph.
// build: hipcc --offload-arch=gfx950 -O3 tools/micro/launch_floor.hip -o build/launch_floor ; run: build/launch_floor
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void touch(int *p) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1; }
__global__ void touch_wide(int *p) { if (threadIdx.x == 0) atomicAdd(&p[1 + (blockIdx.x & 63)], 1); }

int main() {
    int *buf; CK(hipMalloc(&buf, 4096)); CK(hipMemset(buf, 0, 4096));
    hipStream_t s; CK(hipStreamCreate(&s));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int n = 2000;
    for (int wide = 0; wide < 2; ++wide) {
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipStreamSynchronize(s));
            CK(hipEventRecord(e0, s));
            const auto t0 = std::chrono::steady_clock::now();
            for (int i = 0; i < n; ++i) { if (wide) touch_wide<<<256, 256, 0, s>>>(buf); else touch<<<1, 64, 0, s>>>(buf); }
            const auto t1 = std::chrono::steady_clock::now();
            CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep == 2)
                printf("%-34s eager: host %.2f us per launch call, stream %.2f us per kernel\n", wide ? "256 workgroups x 256 threads" : "1 workgroup x 64 threads",
                       std::chrono::duration<double, std::micro>(t1 - t0).count() / n, ms * 1e3 / n);
        }
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
        for (int i = 0; i < 200; ++i) { if (wide) touch_wide<<<256, 256, 0, s>>>(buf); else touch<<<1, 64, 0, s>>>(buf); }
        CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipStreamSynchronize(s));
            CK(hipEventRecord(e0, s));
            for (int i = 0; i < 10; ++i) CK(hipGraphLaunch(ge, s));
            CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep == 2) printf("%-34s graph of 200 dependent kernels, replayed 10 times: %.2f us per kernel\n", "", ms * 1e3 / 2000);
        }
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    }
    return 0;
}
