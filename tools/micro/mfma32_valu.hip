// What the tile loop of coarse32_kernel (D = 32: two chained v_mfma_f32_32x32x16_f16 per code tile and wave, then 8-9 vector
// instructions that look at the PREVIOUS tile's 16 accumulators) can reach on one SIMD, and which ingredient costs what: operands in
// registers, no LDS, no memory; W waves per SIMD (workgroups of 4 W waves, one per CU), variants of the tile body:
//   0  MFMAs only (two per tile, the second accumulates onto the first: C = 0, then C = D)
//   1  + the 8-instruction v_max3 tree on the other accumulator set (the shipped epilogue: tile_max16)
//   2  + the same tree on 16 registers no MFMA ever writes (same instruction count, no MFMA result read)
//   3  MFMAs: ONE 32x32x16 per tile (D <= 16) + the tree on the other set
//   4  like 1, with the tree's first six instructions reading the accumulators through v_pk_max / v_max pairs instead (no VOP3 3-source)
//   5  like 1, the two MFMAs independent (two accumulator sets per tile, each C = 0: what a split over k would look like)
// Prints shader cycles per tile per SIMD (all W waves), next to the 64 (or 32) cycles the matrix pipe needs for W waves' MFMAs.
// build: hipcc --offload-arch=gfx950 -O3 tools/micro/mfma32_valu.hip -o build/mfma32_valu ; run: build/mfma32_valu
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ void tree(float &dst, float &s1, float &s2, const f32x16 &p, float after) {
    asm volatile("v_max3_f32 %0, %3, %4, %5\n\t"
        "v_max3_f32 %1, %6, %7, %8\n\t"
        "v_max3_f32 %2, %9, %10, %11\n\t"
        "v_max3_f32 %0, %0, %12, %13\n\t"
        "v_max3_f32 %1, %1, %14, %15\n\t"
        "v_max3_f32 %2, %2, %16, %17\n\t"
        "v_max3_f32 %0, %0, %1, %2\n\t"
        "v_max_f32 %0, %0, %18"
        : "+v"(dst), "+v"(s1), "+v"(s2)
        : "v"(p[0]), "v"(p[1]), "v"(p[2]), "v"(p[3]), "v"(p[4]), "v"(p[5]), "v"(p[6]), "v"(p[7]), "v"(p[8]), "v"(p[9]),
          "v"(p[10]), "v"(p[11]), "v"(p[12]), "v"(p[13]), "v"(p[14]), "v"(p[15]), "v"(after));      // `after`: a fake input that orders
}                                                                                                          // the tree behind the tile's MFMAs

template <int V, int WPS>
__global__ __launch_bounds__(256 * WPS) __attribute__((amdgpu_waves_per_eu(4, 4))) void loop(float *out, long tiles, float seed, unsigned long long *cyc) {
    // A fragments of NT different code tiles (as if freshly read from LDS: opaque to the compiler at every use), random values —
    // the matrix pipe's power draw, and with it the clock, depends on how the operands toggle (profiles/r05_mfma_random.txt)
    constexpr int NT = 8;
    half8 af[NT][2], b0, b1;
    unsigned h = (blockIdx.x * 1024u + threadIdx.x) * 2654435761u + 12345u;
    auto rnd = [&]() { h = h * 1664525u + 1013904223u; return (_Float16)(((int)(h >> 9) % 2001 - 1000) * (1.0f / 1024.0f) * seed); };
    for (int j = 0; j < 8; ++j) {
        for (int t = 0; t < NT; ++t) { af[t][0][j] = rnd(); af[t][1][j] = rnd(); }
        b0[j] = rnd(); b1[j] = rnd();
    }
    f32x16 accA, accB, other;
    for (int q = 0; q < 16; ++q) { accA[q] = 0.0f; accB[q] = -1.0f; other[q] = seed + q; }
    f32x16 zero;
    for (int q = 0; q < 16; ++q) zero[q] = 0.0f;
    float g = -1e30f, s1 = 0, s2 = 0;
    __syncthreads();
    const unsigned long long c0 = clock64(), w0 = wall_clock64();
    for (long t = 0; t < tiles; t += NT) {
#pragma unroll
        for (int ti = 0; ti < NT; ++ti) {
            const int par = ti & 1;
            f32x16 &cur = par ? accB : accA;
            f32x16 &prv = par ? accA : accB;
            half8 &a0 = af[ti][0], &a1 = af[ti][1];
            asm volatile("" : "+v"(a0), "+v"(a1));
            if (V == 5) {
                f32x16 tmp = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0, zero, 0, 0, 0);
                cur = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b1, zero, 0, 0, 0);
                asm volatile("" :: "v"(tmp));
            } else {
                cur = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0, zero, 0, 0, 0);
                if (V != 3) cur = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b1, cur, 0, 0, 0);
            }
            if (V == 1 || V == 3 || V == 5) tree(g, s1, s2, prv, cur[0]);
            if (V == 2) tree(g, s1, s2, other, cur[0]);
            if (V == 4) {
                float m[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) asm volatile("v_max_f32 %0, %1, %2" : "=v"(m[q]) : "v"(prv[2 * q]), "v"(prv[2 * q + 1]), "v"(cur[0]));
                asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(g) : "v"(m[0]), "v"(m[1]));
                asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(g) : "v"(m[2]), "v"(m[3]));
                asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(g) : "v"(m[4]), "v"(m[5]));
                asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(g) : "v"(m[6]), "v"(m[7]));
            }
            if (V == 0) asm volatile("" :: "v"(prv));
            __builtin_amdgcn_sched_barrier(0);        // a tile's statements stay together, in the order written (as in the kernel)
        }
    }
    const unsigned long long c1 = clock64(), w1 = wall_clock64();
    float s = g + accA[0] + accB[3] + s1 + s2;
    if (s == 123.456f) out[0] = s;
    // the waves of a SIMD are served oldest first: one wave's own clock says little — the block's span does (first start .. last end)
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) {
        atomicMin(&cyc[2], c0); atomicMax(&cyc[3], c1); atomicMin(&cyc[4], w0); atomicMax(&cyc[5], w1);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) { cyc[0] = c1 - c0; cyc[1] = w1 - w0; }
}

template <int V, int WPS>
static void run(const char *what, float *out, unsigned long long *cyc) {
    const long tiles = 40000;
    for (int pass = 0; pass < 3; ++pass) {
        unsigned long long init[6] = {0, 0, ~0ull, 0, ~0ull, 0};
        CK(hipMemcpy(cyc, init, 48, hipMemcpyHostToDevice));
        loop<V, WPS><<<256, 256 * WPS>>>(out, tiles, 1.0f, cyc);
        CK(hipDeviceSynchronize());
        unsigned long long h[6]; CK(hipMemcpy(h, cyc, 48, hipMemcpyDeviceToHost));
        if (pass == 2)
            printf("%d waves per SIMD, %-66s: %6.1f cycles per tile-step of the SIMD (its %d waves' MFMAs: %3d), oldest wave alone %6.1f, clock %.0f MHz\n", WPS, what,
                   (double)(h[3] - h[2]) / tiles, WPS, (V == 3 ? 32 : 64) * WPS, (double)h[0] / tiles, (double)(h[3] - h[2]) / ((double)(h[5] - h[4]) * 0.01));
    }
}

#define ALL(W) \
    run<0, W>("MFMAs only (2 chained per tile)", out, cyc); \
    run<1, W>("+ v_max3 tree on the previous tile's accumulators", out, cyc); \
    run<2, W>("+ the same tree on registers no MFMA writes", out, cyc); \
    run<3, W>("ONE MFMA per tile + the tree", out, cyc); \
    run<4, W>("2 MFMAs + 8 v_max + 4 v_max3 (2-source reads of the accumulators)", out, cyc); \
    run<5, W>("2 independent MFMAs (C = 0 both) + the tree", out, cyc);

int main() {
    float *out; unsigned long long *cyc;
    CK(hipMalloc(&out, 4)); CK(hipMalloc(&cyc, 48));
    ALL(1) ALL(2) ALL(3) ALL(4)
    return 0;
}
