// How fast can every CU stream an L2 / Infinity-Cache resident buffer into LDS by LDS-DMA (global_load_lds, 16 bytes per lane)?
// The proposal kernels stream the codebook image this way — 34 KB stages of 1 KB chunks through a ring of four, requested two
// ahead, one barrier per stage — so this is the ceiling of their stream with no MFMA work at all.
// build: hipcc --offload-arch=gfx950 -O3 tools/micro/stream_lds.hip -o build/stream_lds ; run on the GPU box:
//   build/stream_lds [MB of buffer] [waves per workgroup] [stages ahead] [slices]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
constexpr int CHUNK = 1024, NCH = 34, STAGE = NCH * CHUNK, NBUF = 4;

template <int AHEAD>
__global__ __launch_bounds__(512) void stream_kernel(const char *__restrict__ src, long nstages, int nslices, int reps, int *sink) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, waves = blockDim.x >> 6;
    const long per = nstages / nslices, st0 = (blockIdx.x % nslices) * per, st1 = st0 + per;
    auto issue = [&](long st, int buf) {
        const char *s = src + st * (long)STAGE;
        for (int c = wave; c < NCH; c += waves)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(s + c * CHUNK + lane * 16),
                                             (__attribute__((address_space(3))) void *)(lds + buf * STAGE + c * CHUNK), 16, 0, 0);
    };
    int acc = 0;
    for (int r = 0; r < reps; ++r) {
        for (int a = 0; a < AHEAD; ++a) if (st0 + a < st1) issue(st0 + a, a);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        for (long st = st0; st < st1; ++st) {
            if (st + AHEAD < st1) issue(st + AHEAD, (int)((st + AHEAD - st0) % NBUF));
            acc += *(const int *)(lds + ((st - st0) % NBUF) * STAGE + threadIdx.x * 4);     // touch the stage
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
    }
    if (acc == 0x12345678) *sink = acc;
}

int main(int argc, char **argv) {
    const double mb = argc > 1 ? atof(argv[1]) : 8.4;
    const int waves = argc > 2 ? atoi(argv[2]) : 8, ahead = argc > 3 ? atoi(argv[3]) : 2, nslices = argc > 4 ? atoi(argv[4]) : 1;
    long nstages = (long)(mb * 1e6 / STAGE) / nslices * nslices;
    char *buf; int *sink;
    CK(hipMalloc(&buf, nstages * (long)STAGE)); CK(hipMemset(buf, 1, nstages * (long)STAGE)); CK(hipMalloc(&sink, 4));
    const int lds = NBUF * STAGE, reps = 8 * nslices;
    auto kern = ahead == 3 ? stream_kernel<3> : (ahead == 1 ? stream_kernel<1> : stream_kernel<2>);
    CK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e9f;
    for (int i = 0; i < 6; ++i) {
        CK(hipEventRecord(e0));
        kern<<<256, waves * 64, lds>>>(buf, nstages, nslices, reps, sink);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (i > 0 && ms < best) best = ms;
    }
    const double bytes = 256.0 * (double)(nstages / nslices) * STAGE * reps;
    printf("buffer %.1f MB, %d waves per workgroup, %d stages ahead, %d slices: %.3f ms, %.2f TB/s into LDS over 256 workgroups (%.1f GB/s per CU)\n",
           nstages * (double)STAGE / 1e6, waves, ahead, nslices, best, bytes / best / 1e9, bytes / best / 1e6 / 256);
    return 0;
}
