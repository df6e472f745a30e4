// Where the time of the last-resort fp32 pass over a few listed rows goes (exact_kernel, vqhip_exact_kernels.h): launches the
// library's kernel on synthetic data with VQ_EXACT_STAMPS and prints, over the workgroups, the 100 MHz stamps of its phases.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DVQ_EXACT_STAMPS -Iinclude -Ivector_quantization_amd/csrc
//        tools/micro/exact_rows.hip -o build/exact_rows ; run on the GPU box: build/exact_rows [rows] [K] [D] [few_max]
#include "vqhip.h"
#include "vqhip_kernels.h"
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

int main(int argc, char **argv) {
    const int rows = argc > 1 ? atoi(argv[1]) : 12, K = argc > 2 ? atoi(argv[2]) : 16384, D = argc > 3 ? atoi(argv[3]) : 256;
    const int few_max = argc > 4 ? atoi(argv[4]) : 1 << 20, N = 3072;
    std::vector<float> hx((size_t)N * D), he((size_t)K * D), hen(K), hxn(N);
    srand(1);
    for (auto &v : hx) v = rand() / (float)RAND_MAX - 0.5f;
    for (auto &v : he) v = rand() / (float)RAND_MAX - 0.5f;
    for (int k = 0; k < K; ++k) { float a = 0; for (int d = 0; d < D; ++d) a = fmaf(he[(size_t)k * D + d], he[(size_t)k * D + d], a); hen[k] = a; }
    for (int n = 0; n < N; ++n) { float a = 0; for (int d = 0; d < D; ++d) a = fmaf(hx[(size_t)n * D + d], hx[(size_t)n * D + d], a); hxn[n] = a; }
    std::vector<int> hl(rows);
    for (int i = 0; i < rows; ++i) hl[i] = (i * 977 + 13) % N;
    float *x, *e, *en, *xn; int *list, *cnt; u64 *keys; int64_t *idx;
    CK(hipMalloc(&x, hx.size() * 4)); CK(hipMalloc(&e, he.size() * 4)); CK(hipMalloc(&en, K * 4)); CK(hipMalloc(&xn, N * 4));
    CK(hipMalloc(&list, (rows + 64) * 4)); CK(hipMemset(list, 0x7f, (rows + 64) * 4)); CK(hipMalloc(&cnt, 16)); CK(hipMalloc(&keys, N * 8)); CK(hipMalloc(&idx, N * 8));
    CK(hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(e, he.data(), he.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(en, hen.data(), K * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(xn, hxn.data(), N * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(list, hl.data(), rows * 4, hipMemcpyHostToDevice));
    const int64_t ncb = (K + 63) / 64;
    const int grid = (int)(ncb < 256 ? 256 : (ncb > 1024 ? 1024 : ncb));
    const int lds = vq_few_lds_bytes(D);
    CK(hipFuncSetAttribute((const void *)exact_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e9f;
    for (int rep = 0; rep < 12; ++rep) {
        int hc[4] = {rows, 0, 0, 0};
        CK(hipMemcpy(cnt, hc, 16, hipMemcpyHostToDevice));
        CK(hipMemset(keys, 0xFF, N * 8));
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        exact_kernel<0><<<grid, 256, lds>>>(x, e, en, xn, N, K, D, VQHIP_METRIC_L2, list, cnt, keys, cnt + 1, idx, nullptr, few_max);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        best = std::min(best, ms);
    }
    std::vector<unsigned long long> st(1024 * 8);
    CK(hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(vq_exact_stamps), st.size() * 8));
    std::vector<unsigned long long> cy(1024 * 8);
    CK(hipMemcpyFromSymbol(cy.data(), HIP_SYMBOL(vq_exact_cycles), cy.size() * 8));
    printf("shader clock over workgroup 0's item: %.0f MHz\n", (double)(cy[3] - cy[0]) / ((double)(st[3] - st[0]) * 0.01));
    std::vector<int64_t> hidx(N);
    CK(hipMemcpy(hidx.data(), idx, N * 8, hipMemcpyDeviceToHost));
    // check row 0 of the list on the host (k-ordered fma chain)
    { const int r = hl[0]; double bd = 1e30; int bk = -1;
      for (int k = 0; k < K; ++k) { float a = 0; for (int d = 0; d < D; ++d) a = fmaf(he[(size_t)k * D + d], -2.0f * hx[(size_t)r * D + d], a);
          float t = (a + hxn[r]) + hen[k]; t = t < 0 ? 0 : t; float dd = sqrtf(t); if (dd < bd) { bd = dd; bk = k; } }
      printf("row %d: device %ld host %d\n", r, (long)hidx[r], bk); }
    unsigned long long t0 = ~0ull;
    for (int g = 0; g < grid; ++g) t0 = std::min(t0, st[g * 8]);
    const char *names[8] = {"item start", "x staged, ring requested", "block 0 landed", "blocks done", "keys sent", "atomics drained",
                            "ticket drawn", "end"};
    printf("rows %d K %d D %d few_max %d grid %d: event time (best of 12) %.1f us; stamps in us after the first workgroup's start, min / median / max over workgroups\n",
           rows, K, D, few_max, grid, best * 1e3);
    for (int i = 0; i < 8; ++i) {
        std::vector<double> v;
        for (int g = 0; g < grid; ++g) if (st[g * 8 + i] >= t0) v.push_back((st[g * 8 + i] - t0) * 0.01);
        if (v.empty()) continue;
        std::sort(v.begin(), v.end());
        printf("  %-26s %7.2f %7.2f %7.2f\n", names[i], v.front(), v[v.size() / 2], v.back());
    }
    return 0;
}
