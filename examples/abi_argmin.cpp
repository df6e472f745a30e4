// Standalone use of the C ABI (include/vqhip.h): no Python, no torch — HIP runtime + libvqhip.so only.
// Quantizes random latents with vqhip_codebook_prepare + vqhip_argmin, checks the indices against the all-fp32
// vqhip_argmin_exact, gathers z / straight-through output / squared error with vqhip_gather_ste_loss.
//   hipcc --offload-arch=gfx950 -Iinclude examples/abi_argmin.cpp -Lvector_quantization_amd -lvqhip \
//         -Wl,-rpath,$PWD/vector_quantization_amd -o /tmp/abi_argmin && /tmp/abi_argmin
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "vqhip.h"

#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
#define VQ_OK(x) do { int r_ = (x); if (r_ != VQHIP_OK) { std::fprintf(stderr, "%s -> %d (%s)\n", #x, r_, vqhip_last_error()); return 3; } } while (0)

int main() {
    const int64_t N = 10000, K = 4096;
    const int D = 64;
    std::vector<float> hx(N * D), he(K * D);
    uint64_t s = 0x9E3779B97F4A7C15ull;
    auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (float)((double)(s >> 11) / 9007199254740992.0 * 2.0 - 1.0); };
    for (auto &v : hx) v = rnd();
    for (auto &v : he) v = rnd();

    float *x, *e, *z, *zste; int64_t *idx, *idx2; double *sse; void *cb, *ws; int *stats;
    HIP_OK(hipMalloc(&x, hx.size() * 4)); HIP_OK(hipMalloc(&e, he.size() * 4));
    HIP_OK(hipMalloc(&z, hx.size() * 4)); HIP_OK(hipMalloc(&zste, hx.size() * 4));
    HIP_OK(hipMalloc(&idx, N * 8)); HIP_OK(hipMalloc(&idx2, N * 8)); HIP_OK(hipMalloc(&sse, 8)); HIP_OK(hipMalloc(&stats, 16));
    HIP_OK(hipMalloc(&cb, vqhip_codebook_bytes(K, D))); HIP_OK(hipMalloc(&ws, vqhip_workspace_bytes(N, K, D)));
    HIP_OK(hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(e, he.data(), he.size() * 4, hipMemcpyHostToDevice));
    HIP_OK(hipMemset(sse, 0, 8));
    hipStream_t st; HIP_OK(hipStreamCreate(&st));

    const int64_t cb_bytes = vqhip_codebook_bytes(K, D), ws_bytes = vqhip_workspace_bytes(N, K, D);
    VQ_OK(vqhip_codebook_prepare(e, K, D, VQHIP_METRIC_L2, cb, cb_bytes, st));
    VQ_OK(vqhip_argmin(x, VQHIP_DTYPE_F32, e, cb, cb_bytes, N, K, D, VQHIP_METRIC_L2, idx, nullptr, ws, ws_bytes, st));
    // an undersized workspace is refused before anything is launched
    if (vqhip_argmin(x, VQHIP_DTYPE_F32, e, cb, cb_bytes, N, K, D, VQHIP_METRIC_L2, idx, nullptr, ws, ws_bytes - 1, st) != VQHIP_EINVAL) { std::puts("undersized ws accepted"); return 4; }
    VQ_OK(vqhip_argmin_stats(ws, stats, st));
    VQ_OK(vqhip_gather_ste_loss(x, VQHIP_DTYPE_F32, e, idx, N, D, z, zste, sse, st));
    VQ_OK(vqhip_argmin_exact(x, VQHIP_DTYPE_F32, e, N, K, D, VQHIP_METRIC_L2, idx2, nullptr, nullptr, ws, ws_bytes, st));
    HIP_OK(hipStreamSynchronize(st));

    std::vector<int64_t> a(N), b(N); double hsse = 0; int hstats[4];
    HIP_OK(hipMemcpy(a.data(), idx, N * 8, hipMemcpyDeviceToHost)); HIP_OK(hipMemcpy(b.data(), idx2, N * 8, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(&hsse, sse, 8, hipMemcpyDeviceToHost)); HIP_OK(hipMemcpy(hstats, stats, 16, hipMemcpyDeviceToHost));
    int64_t bad = 0;
    for (int64_t n = 0; n < N; ++n) bad += a[n] != b[n];
    // host check of one row: the chosen code really is the nearest (double precision)
    const int64_t n0 = 1234; double best = 1e300; int64_t kbest = -1;
    for (int64_t k = 0; k < K; ++k) { double d = 0; for (int j = 0; j < D; ++j) { double t = (double)hx[n0 * D + j] - he[k * D + j]; d += t * t; } if (d < best) { best = d; kbest = k; } }
    std::printf("libvqhip %d: N=%lld K=%lld D=%d  mismatches vs fp32 path %lld  row %lld -> code %lld (host nearest %lld)  mse %.6f  "
                "second-pass rows %d, re-ranked rows %d, fp32-pass rows %d\n", vqhip_version(), (long long)N, (long long)K, D,
                (long long)bad, (long long)n0, (long long)a[n0], (long long)kbest, hsse / (double)(N * D), hstats[0], hstats[1], hstats[2]);
    const bool ok = bad == 0 && a[n0] == kbest;
    std::puts(ok ? "ABI OK" : "ABI FAILED");
    return ok ? 0 : 1;
}
