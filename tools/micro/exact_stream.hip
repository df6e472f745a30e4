// The all-fp32 route, register form (exact_tiled_kernel) against the streamed form (exact_stream_kernel): the same launch on the
// same operands, keys / distances compared bit for bit, best-of-n HIP-event times.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Iinclude -Ivector_quantization_amd/csrc
//        tools/micro/exact_stream.hip -o build/exact_stream ; run on the GPU box: build/exact_stream [N] [K] [D] [bf16 0/1] [mode] [metric]
#include "vqhip.h"
#include "vqhip_kernels.h"
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

static float gauss() {
    float u = (rand() + 1.0f) / (RAND_MAX + 2.0f), v = rand() / (float)RAND_MAX;
    return sqrtf(-2.0f * logf(u)) * cosf(6.2831853f * v);
}
static uint16_t to_bf16(float f) { uint32_t b; memcpy(&b, &f, 4); b += 0x7FFFu + ((b >> 16) & 1u); return (uint16_t)(b >> 16); }
static float from_bf16(uint16_t h) { uint32_t b = (uint32_t)h << 16; float f; memcpy(&f, &b, 4); return f; }

template <int DT, int MODE>
static int run(int64_t N, int64_t K, int D, int metric, int reps) {
    std::vector<float> hx((size_t)N * D), he((size_t)K * D), hen((K + 63) / 64 * 64, 0.0f), hxn(N);
    std::vector<uint16_t> hxb;
    srand(7);
    for (auto &v : hx) v = gauss();
    for (auto &v : he) v = gauss();
    // a few exact duplicates and near-ties so that the tie rules are exercised
    if (!getenv("XS_NODUP")) for (int64_t k = 0; k + 1 < K; k += 997) memcpy(&he[(size_t)(k + 1) * D], &he[(size_t)k * D], D * 4);
    if (DT) { hxb.resize(hx.size()); for (size_t i = 0; i < hx.size(); ++i) { hxb[i] = to_bf16(hx[i]); hx[i] = from_bf16(hxb[i]); } }
    for (int64_t k = 0; k < K; ++k) { float a = 0; for (int d = 0; d < D; ++d) a = fmaf(he[(size_t)k * D + d], he[(size_t)k * D + d], a); hen[k] = a; }
    for (int64_t n = 0; n < N; ++n) { float a = 0; for (int d = 0; d < D; ++d) a = fmaf(hx[(size_t)n * D + d], hx[(size_t)n * D + d], a); hxn[n] = a; }
    void *x; float *e, *en, *xn; u64 *keys[2]; float *dout[2] = {nullptr, nullptr};
    const int64_t nkeys = MODE == 1 ? K : N;
    CK(hipMalloc(&x, hx.size() * (DT ? 2 : 4))); CK(hipMalloc(&e, he.size() * 4)); CK(hipMalloc(&en, hen.size() * 4)); CK(hipMalloc(&xn, N * 4));
    for (int v = 0; v < 2; ++v) { CK(hipMalloc(&keys[v], nkeys * 8)); if (MODE == 2) CK(hipMalloc(&dout[v], (size_t)N * K * 4)); }
    CK(hipMemcpy(x, DT ? (void *)hxb.data() : (void *)hx.data(), hx.size() * (DT ? 2 : 4), hipMemcpyHostToDevice));
    CK(hipMemcpy(e, he.data(), he.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(en, hen.data(), hen.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(xn, hxn.data(), N * 4, hipMemcpyHostToDevice));
    const int64_t items = ((N + 127) / 128) * ((K + 255) / 256);
    const int lds_old = 2 * 32 * 128 * 4 + 8 * 32 * 4, lds_new = vq_xs_lds_bytes(DT);
    CK(hipFuncSetAttribute((const void *)exact_tiled_kernel<DT, MODE, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_old));
    CK(hipFuncSetAttribute((const void *)exact_stream_kernel<DT, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_new));
    int occ = 0;
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void *)exact_stream_kernel<DT, MODE>, 256, lds_new));
    const int wgs = getenv("XS_WGS") ? atoi(getenv("XS_WGS")) : 256 * (occ > 0 ? occ : 1);
    const int grid_old = (int)std::min<int64_t>(items, 1024), grid_new = (int)std::min<int64_t>(items, wgs);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best[2] = {1e9f, 1e9f};
    auto launch = [&](int v) {
        if (v == 0) exact_tiled_kernel<DT, MODE, true><<<grid_old, 256, lds_old>>>(x, e, en, xn, N, K, D, metric, keys[0], dout[0]);
        else exact_stream_kernel<DT, MODE><<<grid_new, 256, lds_new>>>(x, e, en, xn, N, K, D, metric, keys[1], dout[1]);
    };
    // per form: warm-up launches, then `reps` launches back to back between two events (clocks settled, as a caller's loop sees them)
    for (int round = 0; round < 2; ++round)
        for (int v = 0; v < 2; ++v) {
            CK(hipMemset(keys[v], 0xFF, nkeys * 8));
            for (int w = 0; w < 2; ++w) launch(v);
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0));
            for (int rep = 0; rep < reps; ++rep) launch(v);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            CK(hipGetLastError());
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            best[v] = std::min(best[v], ms / reps);
        }
    int64_t bad = 0;
    if (MODE != 2) {
        std::vector<u64> k0(nkeys), k1(nkeys);
        CK(hipMemcpy(k0.data(), keys[0], nkeys * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(k1.data(), keys[1], nkeys * 8, hipMemcpyDeviceToHost));
        for (int64_t i = 0; i < nkeys; ++i) if (k0[i] != k1[i]) { if (bad < 4) printf("  key %ld: tiled %016llx stream %016llx\n", (long)i, k0[i], k1[i]); ++bad; }
    } else {
        std::vector<uint32_t> d0((size_t)N * K), d1((size_t)N * K);
        CK(hipMemcpy(d0.data(), dout[0], d0.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(d1.data(), dout[1], d1.size() * 4, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < d0.size(); ++i) if (d0[i] != d1[i]) { if (bad < 4) printf("  d[%zu]: tiled %08x stream %08x\n", i, d0[i], d1[i]); ++bad; }
    }
    const double fl = 2.0 * N * K * D;
    printf("N %ld K %ld D %d %s mode %d metric %d: tiled %.3f ms %.1f TFLOP/s (grid %d) | stream %.3f ms %.1f TFLOP/s (grid %d, %d WG/CU) | mismatches %ld\n",
           (long)N, (long)K, D, DT ? "bf16" : "fp32", MODE, metric, best[0], fl / best[0] / 1e9, grid_old, best[1], fl / best[1] / 1e9, grid_new, occ, (long)bad);
    CK(hipFree(x)); CK(hipFree(e)); CK(hipFree(en)); CK(hipFree(xn));
    for (int v = 0; v < 2; ++v) { CK(hipFree(keys[v])); if (dout[v]) CK(hipFree(dout[v])); }
    return bad != 0;
}

int main(int argc, char **argv) {
    const int64_t N = argc > 1 ? atoll(argv[1]) : 65536, K = argc > 2 ? atoll(argv[2]) : 8192;
    const int D = argc > 3 ? atoi(argv[3]) : 256, bf = argc > 4 ? atoi(argv[4]) : 0, mode = argc > 5 ? atoi(argv[5]) : 0;
    const int metric = argc > 6 ? atoi(argv[6]) : VQHIP_METRIC_L2, reps = argc > 7 ? atoi(argv[7]) : 5;
    if (mode == 0) return bf ? run<1, 0>(N, K, D, metric, reps) : run<0, 0>(N, K, D, metric, reps);
    if (mode == 1) return bf ? run<1, 1>(N, K, D, metric, reps) : run<0, 1>(N, K, D, metric, reps);
    return bf ? run<1, 2>(N, K, D, metric, reps) : run<0, 2>(N, K, D, metric, reps);
}
