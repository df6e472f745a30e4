// libvqhip device kernels, unit 5 of 8: the all-fp32 MFMA pass over whole batches (argmin_exact, col_argmin fallback, distance).
// Included by vqhip_kernels.h.
#pragma once
// ------------------------------------------------------------------------------------------------
// exact fp32 MFMA pass (v_mfma_f32_32x32x2_f32 == k-ordered fmaf chain)
// ------------------------------------------------------------------------------------------------
// Row argmin, L2: a lane meets its codes in increasing index order, so a later code can only replace the lane's best with a
// STRICTLY smaller distance; sqrt is correctly rounded, hence monotone: radicand t >= tb (the smallest radicand the lane has
// keyed) implies sqrt(t) >= sqrt(tb) and the code cannot win — its sqrt and key (the larger half of the epilogue's
// instructions) are skipped.  NaN radicands never compare >= and always go through (NaN sorts first in dist_key).
__device__ __forceinline__ bool l2_skip(float t, float tb) { return t >= tb; }

// Last-resort path of vqhip_argmin (MFMA form, long lists): LISTED rows against the whole codebook, row argmin via 64-bit atomicMin keys[row].
// Work item = (tile of 32 listed rows, chunk of 128 codes: one 32-code tile per wave); persistent grid-stride loop over items.
// A lane owns one latent row (as B operand: row j, k-parity h) and one code row (A operand) and walks them in batches of 32
// dims — 8 pieces of 16 bytes each — through a ring of RING batches requested ahead; the chain of D/2 dependent
// v_mfma_f32_32x32x2_f32 is the oracle's k-ordered fma chain.  The dims loop is a RUN-TIME loop on purpose: the first form
// unrolled all 64 pieces of a 256-dim block (11 600 instructions, 93 KB of code executed once per wave) and a handful of
// listed rows cost 29-36 us whatever was taken out of the data path — MFMAs, atomics, the ticket, the order of the loads
// (profiles/r03_exact_rows.txt): the waves were waiting for their own instruction stream.
template <int DT>
__device__ __forceinline__ void exact_rows_mfma(const void *__restrict__ x, const float *__restrict__ e,
                                                const float *__restrict__ en_in, const float *__restrict__ xn_in, int64_t N,
                                                int64_t K, int D, int metric, const int *__restrict__ row_list,
                                                const int *__restrict__ nrows_dev, u64 *__restrict__ keys) {
    constexpr int CHUNK = 4 * 32;               // codes per work item (4 waves)
    constexpr int RING = 4;                     // batches in flight
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 31, h = lane >> 5;
    const int64_t nrows = row_list ? (int64_t)(*nrows_dev) : N;
    const int64_t ntiles = (nrows + 31) / 32;
    const int64_t nchunks = (K + CHUNK - 1) / CHUNK;
    const float sx = (VQ_IS_L2(metric)) ? -2.0f : 1.0f;
    // D % 4 == 0 (the proposal route this path belongs to has D % 8 == 0).  An element-wise tail form in the same kernel made
    // hipcc hoist its ~250 loop-invariant `d + q < D` conditions in front of everything: thousands of scalar instructions and
    // lane spills per wave, 7 us before the first request went out.
    const int nb = (D + 31) / 32;

    for (int64_t item = blockIdx.x; item < ntiles * nchunks; item += gridDim.x) {
        const int64_t tile = item / nchunks, chunk = item % nchunks;
        const int64_t slot = tile * 32 + j;
        const bool rvalid = slot < nrows;
        const int64_t row = rvalid ? (row_list ? (int64_t)row_list[slot] : slot) : 0;
        const int64_t k = chunk * CHUNK + (int64_t)wave * 32 + j;     // this lane's A row (code)
        const bool kvalid = k < K;
        const float *erow = e + (kvalid ? k : 0) * D;
        // oracle-order |x|^2: precomputed by the token-side kernel, computed per lane otherwise
        const float xn = (VQ_IS_L2(metric) && rvalid) ? (xn_in ? xn_in[row] : sqnorm_thread<DT>(x, row * D, D)) : 0.0f;

        // piece = 4 consecutive dims of a row, as fp32, at a clamped address: no branch between the requests
        auto x_piece = [&](int d) -> float4 {
            const int64_t off = row * D + (d < D ? d : 0);
            if (DT == 0) return *(const float4 *)((const float *)x + off);
            const uint2 t = *(const uint2 *)((const uint16_t *)x + off);
            return float4{__uint_as_float(t.x << 16), __uint_as_float(t.x & 0xFFFF0000u),
                          __uint_as_float(t.y << 16), __uint_as_float(t.y & 0xFFFF0000u)};
        };
        auto e_piece = [&](int d) -> float4 { return *(const float4 *)(erow + (d < D ? d : 0)); };
        float4 xr[RING][8], er[RING][8];
#pragma unroll
        for (int u = 0; u < RING; ++u) {
            if (u < nb) {
#pragma unroll
                for (int i = 0; i < 8; ++i) { xr[u][i] = x_piece(32 * u + 4 * i); er[u][i] = e_piece(32 * u + 4 * i); }
            }
        }
        f32x16 acc;
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[q] = 0.0f;
        for (int b0 = 0; b0 < nb; b0 += RING) {
#pragma unroll
            for (int u = 0; u < RING; ++u) {
                const int b = b0 + u;
                if (b < nb) {
                    __builtin_amdgcn_sched_barrier(0);   // requests stay where they are written (hipcc sinks them to their uses)
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const int d = 32 * b + 4 * i;
                        // a whole piece past D is all zeros: +0 (or -0) added to an accumulator that is never -0 changes nothing
                        const bool okx = rvalid && d < D, oke = kvalid && d < D;
                        const float4 xv = xr[u][i], ev = er[u][i];
                        const float b0v = sx * (okx ? (h ? xv.y : xv.x) : 0.0f), b1v = sx * (okx ? (h ? xv.w : xv.z) : 0.0f);
                        const float a0v = oke ? (h ? ev.y : ev.x) : 0.0f, a1v = oke ? (h ? ev.w : ev.z) : 0.0f;
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0v, b0v, acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1v, b1v, acc, 0, 0, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    if (b + RING < nb) {
#pragma unroll
                        for (int i = 0; i < 8; ++i) { xr[u][i] = x_piece(32 * (b + RING) + 4 * i); er[u][i] = e_piece(32 * (b + RING) + 4 * i); }
                    }
                }
            }
        }

        // epilogue: C[code row][token col j]
        u64 best = ~0ull;
        float tb = INFINITY;                     // smallest radicand this lane has turned into a key (see l2_skip)
        const int64_t kbase = chunk * CHUNK + (int64_t)wave * 32;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int64_t kq = kbase + mfma_row(q, h);
            float d;
            if (VQ_IS_L2(metric)) {
                const float enk = (kq < K) ? en_in[kq] : 0.0f;
                float t = VQ_SWAPPED(metric) ? (acc[q] + enk) + xn : (acc[q] + xn) + enk;
                t = (t < 0.0f) ? 0.0f : t;
                if (l2_skip(t, tb)) continue;
                d = sqrtf(t);
                if (kq < K) tb = fminf(tb, t);
            } else {
                d = cos_distance(acc[q], metric);
            }
            if (kq < K) { u64 key = dist_key(d, (uint32_t)kq); best = key < best ? key : best; }
        }
        u64 o = __shfl_xor(best, 32, 64);
        best = o < best ? o : best;
#ifndef VQ_EXACT_NO_ATOMIC       // (timing-only diagnostic build, tools/micro/exact_rows.hip)
        if (h == 0 && rvalid && best != ~0ull) atomicMin(&keys[row], best);
#endif
    }
}

// The same last-resort pass on the VALU, for what it is nearly always asked to do: a HANDFUL of listed rows (a CVQ-VAE step
// sends 4-35 of 3072 rows here) against the whole codebook.  The MFMA form's work item is a chain of D dependent
// v_mfma_f32_32x32x2_f32 whatever the number of rows — 16 384 cycles at D = 256 before the first key exists, 24-28 us a launch
// with its load latency — and nearly all of the 32 x 32 outputs of each MFMA belong to rows that are not there.  Here a lane
// owns ONE code and a wave up to 4 of the listed rows (2 pairs): acc[i] = fmaf(e[k][d], sx * x[row_i][d], acc[i]) for
// d = 0 .. D-1, the oracle's k-ordered chain by definition (what the MFMA form was verified to equal), 4 independent chains
// per lane, two per v_pk_fma_f32.
// Work item = (64 codes, 16 listed rows): the 4 waves of a workgroup share the code tile and split the row pairs evenly.
// The x tile of the item — 16 rows x D, scaled by sx and widened to fp32, the two rows of a pair interleaved value by value —
// is staged once (D <= VQ_FEW_MAX_D) and read back as broadcasts (every lane the same address).  The e tile, 64 codes x 128 B
// per block of 32 dims, arrives by LDS-DMA in whole 128-byte lines through a ring of VQ_FEW_RING blocks, all but one of them
// requested ahead (at D = 256 the whole 64 KB tile is in flight before the first fma: with one block ahead the pass waited
// out a memory latency per block, 13 us per item): DMA j (2 per wave), lane i -> code 8 (i >> 3) + j, 16-byte piece
// (i & 7) ^ j.  Lane l reads piece p of its code l = 8 a + b at slot 64 b + 8 a + (p ^ b): within each of ds_read_b128's four
// 16-lane groups the 16 slots are distinct mod 16 (conflict-free).  One barrier per block.  Then the epilogue of the MFMA form
// per (code, row), the wave's smallest key and ONE atomicMin per (wave, row).
// Where a launch's time goes (tools/micro/exact_rows.hip, 12 rows, K = 16384, D = 256, us after the first workgroup starts,
// median over workgroups; profiles/r04_exact_rows_stamps.txt): x tile staged 2.3 (row ids -> x rows: two dependent memory
// latencies; the e tile lands meanwhile), blocks done 8.3 (one wave per SIMD: 1600 cycles per block for 40 LDS reads and 64
// v_pk_fma_f32), keys sent 10.0, atomics drained 11.1, last workgroup done 11.6 — 17.7 us by HIP events against 29.1 for the
// MFMA form.  Tried and not kept: the whole tile resident and no barrier between blocks, reads of the next half block ahead
// of the fmas (slower as compiled: 6.6 us of blocks for 2 pairs per wave; faster, 2.9 against 3.4, for 1).
typedef float vq_f32x2 __attribute__((ext_vector_type(2)));

// one block of NP pieces (4 dims each) for the M row pairs of a wave: v_pk_fma_f32 — the two halves are the two rows of a pair
// (each half a correctly rounded fma like v_fma_f32), the code's value feeds both.  NP == 8 (a whole block): every LDS read of
// the block is issued before the first fma (one wave per SIMD: nobody else hides a read's latency; read-then-use piece by
// piece took 2000 cycles per block, 6.6 us of a 9 us item).  NP < 8: the tail block of a D that is not a multiple of 32.
template <int M, int NP>
__device__ __forceinline__ void exact_few_block(const float4 *__restrict__ et, const float4 *__restrict__ xt, int npcp, int eslot,
                                                int sw, int np, vq_f32x2 (&acc)[2]) {
    if constexpr (NP == 8) {
        float4 ev[8], xa[M][8], xb[M][8];
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            ev[p] = et[eslot + (p ^ sw)];
#pragma unroll
            for (int i = 0; i < M; ++i) { xa[i][p] = xt[(i * npcp + p) * 2]; xb[i][p] = xt[(i * npcp + p) * 2 + 1]; }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int p = 0; p < 8; ++p) {
#pragma unroll
            for (int i = 0; i < M; ++i) {
                acc[i] = __builtin_elementwise_fma(vq_f32x2{ev[p].x, ev[p].x}, vq_f32x2{xa[i][p].x, xa[i][p].y}, acc[i]);
                acc[i] = __builtin_elementwise_fma(vq_f32x2{ev[p].y, ev[p].y}, vq_f32x2{xa[i][p].z, xa[i][p].w}, acc[i]);
                acc[i] = __builtin_elementwise_fma(vq_f32x2{ev[p].z, ev[p].z}, vq_f32x2{xb[i][p].x, xb[i][p].y}, acc[i]);
                acc[i] = __builtin_elementwise_fma(vq_f32x2{ev[p].w, ev[p].w}, vq_f32x2{xb[i][p].z, xb[i][p].w}, acc[i]);
            }
        }
    } else {
        for (int p = 0; p < np; ++p) {
            const float4 ev = et[eslot + (p ^ sw)];
#pragma unroll
            for (int i = 0; i < M; ++i) {
                const float4 xa = xt[(i * npcp + p) * 2], xb = xt[(i * npcp + p) * 2 + 1];   // (row0, row1) of dims 0,1 | 2,3
                acc[i] = __builtin_elementwise_fma(vq_f32x2{ev.x, ev.x}, vq_f32x2{xa.x, xa.y}, acc[i]);
                acc[i] = __builtin_elementwise_fma(vq_f32x2{ev.y, ev.y}, vq_f32x2{xa.z, xa.w}, acc[i]);
                acc[i] = __builtin_elementwise_fma(vq_f32x2{ev.z, ev.z}, vq_f32x2{xb.x, xb.y}, acc[i]);
                acc[i] = __builtin_elementwise_fma(vq_f32x2{ev.w, ev.w}, vq_f32x2{xb.z, xb.w}, acc[i]);
            }
        }
    }
}

#ifdef VQ_EXACT_STAMPS      // tools/micro/exact_rows.hip: where a workgroup's time goes (100 MHz counter, wave 0)
__device__ unsigned long long vq_exact_stamps[1024 * 8];
__device__ unsigned long long vq_exact_cycles[1024 * 8];
#define VQ_STAMP(i) do { if (threadIdx.x == 0) { vq_exact_stamps[(blockIdx.x & 1023) * 8 + (i)] = wall_clock64(); vq_exact_cycles[(blockIdx.x & 1023) * 8 + (i)] = clock64(); } } while (0)
#else
#define VQ_STAMP(i) do {} while (0)
#endif

__device__ __forceinline__ void wait_vm_pairs(int n) {       // s_waitcnt vmcnt(2 n), n = 0 .. 6 (the count is an immediate)
    switch (n) {
        case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        case 1: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
        case 2: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
        case 3: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
        case 4: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
        case 5: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
    }
}

#define VQ_FEW_RING 8                 // e tiles (blocks of 32 dims) in LDS: VQ_FEW_RING - 1 requested ahead
#define VQ_FEW_MAX_D 1024             // the x tile of an item is 16 x D floats of LDS
static inline int vq_few_lds_bytes(int D) { return VQ_FEW_RING * 8192 + 16 * ((D + 31) / 32 * 32) * 4; }

// Smallest dist_key of the wave when lane l holds the key of code k0 + l (so that among equal distances the lowest lane is
// the lowest code): the minimum of the high words — xor-1, xor-2, half-mirror and mirror DPP steps make each row of 16 lanes
// uniform, four v_readlane and scalar minima finish it — and the first lane that holds it.  Wave-uniform result.  (As a 64-bit
// __shfl_xor butterfly — two LDS-crossbar round trips per step and half — the four reductions of a wave took 2.8 us.)
__device__ __forceinline__ u64 wave_min_key(u64 key, int64_t k0) {
    const uint32_t hi = (uint32_t)(key >> 32);
    uint32_t h = hi;
#define VQ_DPP_MIN(ctrl) { const uint32_t o = (uint32_t)__builtin_amdgcn_update_dpp((int)h, (int)h, ctrl, 0xf, 0xf, false); h = o < h ? o : h; }
    VQ_DPP_MIN(0xB1) VQ_DPP_MIN(0x4E) VQ_DPP_MIN(0x141) VQ_DPP_MIN(0x140)
#undef VQ_DPP_MIN
    uint32_t m = (uint32_t)__builtin_amdgcn_readlane((int)h, 0);
#pragma unroll
    for (int r = 1; r < 4; ++r) { const uint32_t o = (uint32_t)__builtin_amdgcn_readlane((int)h, 16 * r); m = o < m ? o : m; }
    const u64 holders = __ballot(hi == m);
    return ((u64)m << 32) | (u64)(uint32_t)(k0 + (__ffsll((unsigned long long)holders) - 1));
}

template <int DT>
__device__ __forceinline__ void exact_rows_few(const void *__restrict__ x, const float *__restrict__ e,
                                               const float *__restrict__ en_in, const float *__restrict__ xn_in, int64_t K,
                                               int D, int metric, const int *__restrict__ row_list, int64_t nrows,
                                               u64 *__restrict__ keys) {
    constexpr int RT = 16;                       // listed rows per work item: 8 pairs, up to 2 pairs per wave
    extern __shared__ __attribute__((aligned(16))) char few_lds[];
    float4 *ering = (float4 *)few_lds;                               // [VQ_FEW_RING][512]
    float *xtile = (float *)(few_lds + VQ_FEW_RING * 8192);          // [pair][piece of the row][dim of the piece][row of the pair]
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t nrt = (nrows + RT - 1) / RT, ncb = (K + 63) / 64;
    const float sx = (VQ_IS_L2(metric)) ? -2.0f : 1.0f;
    const int nb = (D + 31) / 32, npc = D >> 2, npcp = nb * 8;      // pieces of a row: valid, padded
    const int eslot = 64 * (lane & 7) + 8 * (lane >> 3);          // + (p ^ (lane & 7))

    for (int64_t item = blockIdx.x; item < ncb * nrt; item += gridDim.x) {
        const int64_t rt = item / ncb, k0 = (item % ncb) * 64;
        const int nr = (int)((nrows - rt * RT) < RT ? (nrows - rt * RT) : RT);
        const int npair = (nr + 1) >> 1;
        const int pg = (npair + 3) >> 2;                           // pairs per wave (even split): 1 or 2
        const int p_lo = wave * pg;
        const int myn = npair - p_lo < pg ? (npair - p_lo > 0 ? npair - p_lo : 0) : pg;

        auto issue_e = [&](int b) {
            const int np = (D - 32 * b) >= 32 ? 8 : (D - 32 * b) >> 2;
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int j = 2 * wave + u;
                int64_t c = k0 + 8 * (lane >> 3) + j;
                int p = (lane & 7) ^ j;
                c = c < K ? c : K - 1;                             // clamped: what lands there is never used
                p = p < np ? p : 0;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(e + c * D + 32 * b + 4 * p),
                                                 (__attribute__((address_space(3))) void *)(&ering[(b % VQ_FEW_RING) * 512 + 64 * j]), 16, 0, 0);
            }
        };
        // The item's latency chain is kept to: row ids -> x rows -> fma.  Row ids first (one load, lane r holds row r);
        // then everything that depends on them or on nothing — |e|^2 of the lane's code and |x|^2 of the wave's rows for the
        // epilogue, the x tile — and behind those requests the e tiles of the first VQ_FEW_RING - 1 blocks; the x tile is
        // written (scaled, widened) while the e tiles are on their way.  Wave w stages rows w, w + 4, w + 8, w + 12, a lane
        // pieces lane, lane + 64, ... (4 x 4 requests cover D <= VQ_FEW_MAX_D).
        VQ_STAMP(0);
        const int rid_v = row_list[rt * RT + ((lane & 15) < nr ? (lane & 15) : 0)];
        const int64_t k = k0 + lane;
        const float enk = (VQ_IS_L2(metric) && k < K) ? en_in[k] : 0.0f;
        // (the ids move to scalar registers in one go: a v_readlane next to each request made hipcc wait for all earlier
        // requests before each of them)
        int64_t rx[4], re[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) rx[u] = (int64_t)__builtin_amdgcn_readlane(rid_v, (wave + 4 * u) & 15);
#pragma unroll
        for (int i = 0; i < 4; ++i) re[i] = (int64_t)__builtin_amdgcn_readlane(rid_v, ((p_lo + (i >> 1)) * 2 + (i & 1)) & 15);
        float xnv[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) xnv[i] = VQ_IS_L2(metric) ? xn_in[re[i]] : 0.0f;
        float4 xv[4][4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            if (t == 0 || 64 * t < npc) {                          // wave-uniform; inside, every request at a clamped address
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int pc = lane + 64 * t;
                    const int64_t off = rx[u] * D + 4 * (pc < npc ? pc : 0);
                    if (DT == 0) xv[t][u] = *(const float4 *)((const float *)x + off);
                    else {
                        const uint2 w2 = *(const uint2 *)((const uint16_t *)x + off);
                        xv[t][u] = float4{__uint_as_float(w2.x << 16), __uint_as_float(w2.x & 0xFFFF0000u),
                                          __uint_as_float(w2.y << 16), __uint_as_float(w2.y & 0xFFFF0000u)};
                    }
                }
            }
        }
        if (nb >= VQ_FEW_RING - 1) {
#pragma unroll
            for (int b = 0; b < VQ_FEW_RING - 1; ++b) issue_e(b);
        } else {
            for (int b = 0; b < nb; ++b) issue_e(b);
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int r = wave + 4 * u, pc = lane + 64 * t;
                if (r < 2 * npair && pc < npc) {                   // (the missing row of an odd last pair: zeros)
                    float *xs = xtile + ((r >> 1) * npcp + pc) * 8 + (r & 1);
                    xs[0] = r < nr ? sx * xv[t][u].x : 0.0f; xs[2] = r < nr ? sx * xv[t][u].y : 0.0f;
                    xs[4] = r < nr ? sx * xv[t][u].z : 0.0f; xs[6] = r < nr ? sx * xv[t][u].w : 0.0f;
                }
            }
        }

        VQ_STAMP(1);
        vq_f32x2 acc[2] = {vq_f32x2{0.0f, 0.0f}, vq_f32x2{0.0f, 0.0f}};
        for (int b = 0; b < nb; ++b) {
            // block b landed once the DMAs of the blocks requested after it (b + 1 .. b + RING - 2) are all that is pending
            const int after = nb - 1 - b < VQ_FEW_RING - 2 ? nb - 1 - b : VQ_FEW_RING - 2;
            wait_vm_pairs(after);
            __syncthreads();                                       // ... for every wave's part; everybody is done with block b - 1
            if (b == 0) VQ_STAMP(2);
            if (b + VQ_FEW_RING - 1 < nb) issue_e(b + VQ_FEW_RING - 1);   // into the buffer of block b - 1
            const int np = (D - 32 * b) >= 32 ? 8 : (D - 32 * b) >> 2;
            const float4 *et = ering + (b % VQ_FEW_RING) * 512, *xt = (const float4 *)xtile + (p_lo * npcp + b * 8) * 2;
            if (np == 8) {
                if (myn == 2) exact_few_block<2, 8>(et, xt, npcp, eslot, lane & 7, 8, acc);
                else if (myn == 1) exact_few_block<1, 8>(et, xt, npcp, eslot, lane & 7, 8, acc);
            } else {
                if (myn == 2) exact_few_block<2, 7>(et, xt, npcp, eslot, lane & 7, np, acc);
                else if (myn == 1) exact_few_block<1, 7>(et, xt, npcp, eslot, lane & 7, np, acc);
            }
        }
        VQ_STAMP(3);
        __syncthreads();                                           // the next item refills the ring and the x tile

        // keys of the wave's (up to) 4 rows and their wave-wide minima.  A run-time loop on purpose: this code runs once per
        // item, straight from a cold instruction cache — unrolled four times it was 2.4 us of a 13 us launch.
#pragma unroll 1
        for (int i = 0; i < 2 * myn; ++i) {
            const int lr = (p_lo + (i >> 1)) * 2 + (i & 1);
            if (lr >= nr) break;                                   // (the missing row of an odd last pair)
            const int64_t row = (i & 2) ? ((i & 1) ? re[3] : re[2]) : ((i & 1) ? re[1] : re[0]);
            const vq_f32x2 ap = (i & 2) ? acc[1] : acc[0];
            const float a = (i & 1) ? ap[1] : ap[0];
            float d;
            if (VQ_IS_L2(metric)) {
                const float xn = (i & 2) ? ((i & 1) ? xnv[3] : xnv[2]) : ((i & 1) ? xnv[1] : xnv[0]);   // (xn_in: always given here)
                float t = VQ_SWAPPED(metric) ? (a + enk) + xn : (a + xn) + enk;
                t = (t < 0.0f) ? 0.0f : t;
                d = sqrtf(t);
            } else {
                d = cos_distance(a, metric);
            }
            const u64 m = wave_min_key(k < K ? dist_key(d, (uint32_t)k) : ~0ull, k0);
            if (lane == 0) atomicMin(&keys[row], m);
        }
    }
}

// One launch for both forms, chosen on the device by the length of the list (the host does not know it): up to
// few_max (VQ_EXACT_FEW_MAX; 0 when D > VQ_FEW_MAX_D) listed rows take the VALU form, longer lists the MFMA form (twice the fma
// rate once its 32-row tiles are full).  ticket: the workgroup that finishes last decodes keys -> idx (+hist) for the listed rows itself, so the path is ONE
// launch; with an empty list every workgroup returns at once.
#ifndef VQ_EXACT_FEW_MAX
#define VQ_EXACT_FEW_MAX 16
#endif
template <int DT>
__global__ __launch_bounds__(256) void exact_kernel(const void *__restrict__ x, const float *__restrict__ e,
                                                    const float *__restrict__ en_in,
                                                    const float *__restrict__ xn_in, int64_t N, int64_t K, int D,
                                                    int metric, const int *__restrict__ row_list,
                                                    const int *__restrict__ nrows_dev, u64 *__restrict__ keys,
                                                    int *__restrict__ ticket, int64_t *__restrict__ fin_idx,
                                                    int32_t *__restrict__ fin_hist, int few_max, int fin_pos = 0) {
    const int64_t nrows = (int64_t)(*nrows_dev);
    if (nrows <= 0) return;
    if (nrows <= few_max) exact_rows_few<DT>(x, e, en_in, xn_in, K, D, metric, row_list, nrows, keys);
    else exact_rows_mfma<DT>(x, e, en_in, xn_in, N, K, D, metric, row_list, nrows_dev, keys);
    {
        // arrival counter (MI355X guide, Guideline 16, form R1): the only payload is the keys, written by agent-scope atomics —
        // performed at the memory side, past the XCD's L2 — so there is nothing for a release fence to write back (it cost
        // 1.5 us here): every wave drains its own atomics, the workgroup meets, one lane draws the ticket; whoever draws the
        // last one reads the keys with loads that bypass its L1 and L2 (agent-scope relaxed atomic loads).  This ordering — an atomic is
        // complete at the memory side once vmcnt has counted it down — is a property of gfx950's memory system this gfx950-only
        // library relies on, not of the HIP memory model (which would ask for the release fence)
        __shared__ int is_last;
        VQ_STAMP(4);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        VQ_STAMP(5);
        if (threadIdx.x == 0) is_last = (atomicAdd(ticket, 1) == (int)gridDim.x - 1) ? 1 : 0;
        __syncthreads();
        VQ_STAMP(6);
        if (is_last) {
            for (int64_t i = threadIdx.x; i < nrows; i += blockDim.x) {
                const int64_t r = (int64_t)row_list[i];
                const u64 key = __hip_atomic_load(&keys[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const uint32_t kk = (uint32_t)(key & 0xFFFFFFFFull);
                fin_idx[fin_pos ? i : r] = (int64_t)kk;        // fin_pos: by position in the list (the direct column pass)
                if (fin_hist) atomicAdd(&fin_hist[kk], 1);
            }
        }
        VQ_STAMP(7);
    }
}

// Front of the direct column pass (vqhip_col_argmin_rows on a short list): the keys of the listed rows start at "no code yet",
// the arrival ticket at zero.
__global__ void col_direct_init_kernel(const int32_t *__restrict__ rows, const int32_t *__restrict__ count, int64_t cap,
                                       u64 *__restrict__ keys, int *__restrict__ ticket) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) ticket[0] = 0;
    if (i < cap && i < (int64_t)count[0]) keys[rows[i]] = ~0ull;
}

// vqhip_set_tuning key 12 (verification aid): rows 0 .. V-1 of the batch join the list of the last-resort pass whatever the
// earlier stages decided for them (a row already listed is left alone); the pass then overwrites their indices.
__global__ void force_exact_rows_kernel(int V, int *__restrict__ exact_list, int *__restrict__ counters, u64 *__restrict__ keys) {
    const int n = threadIdx.x;
    if (n < V && keys[n] != ~0ull) {
        exact_list[atomicAdd(&counters[2], 1)] = n;
        keys[n] = ~0ull;
    }
}

// ------------------------------------------------------------------------------------------------
// Epilogue of a whole-batch work item (128 rows x 256 codes: acc[c][q] = the lane's column j of code tile c), both forms of the
// pass: row argmin -> the lane's best key (returned); column argmin -> atomicMin per code; distances -> stores.
// en_lds: |e_k|^2 of the item's 256 codes (L2).
// ------------------------------------------------------------------------------------------------
// HW > 0 (distances, exact_stream_kernel): `stage` is a wave-private LDS tile of 32 rows x HW codes (32, or 16 where the wave's
// share of a consumed row block is only 2 KiB: bf16 rows) through which the distances leave as ROWS — a lane's 16 values of a code
// tile are 4 codes x 4 runs down a column of d[N, K], and stored as they lie every wave-store touched 32 lines for 8 bytes each
// (8192 x 8192 x 64: 268 MB in 0.47 ms, slower than the register form); staged, a wave-store is 8 (16) rows x 128 (64) contiguous
// bytes.  K % 4 == 0 (16-byte row segments); other K keep the element stores.
template <int MODE, int HW = 0>
__device__ __forceinline__ u64 tiled_epilogue(const f32x16 (&acc)[8], const float4 *__restrict__ en_lds, float xn, int64_t kbase,
                                              int64_t K, int metric, int h, int j, bool rvalid, int64_t row,
                                              u64 *__restrict__ keys, float *__restrict__ dout, float4 *stage = nullptr,
                                              int64_t N = 0) {
    constexpr int CT = 8;
    // A lane's code of accumulator element (c, q) is kb + o with the CONSTANT
    // o = 32 c + mfma_row(q, 0): existence is `o < krem`, the winner is kept as its o (an inline constant in the select) —
    // no per-element 64-bit index is ever formed (128 of them, computed once for both passes below and kept, were the
    // register form's spills and, under this kernel's 256 registers, 316 more)
    u64 best = ~0ull;
    const int64_t kb = kbase + 4 * h;
    const int64_t left = K - kb;
    int krem = (int)(left < 0 ? 0 : (left > CT * 32 ? CT * 32 : left));      // this lane's codes kb + o exist for o < krem
    uint32_t kb32 = (uint32_t)kb;
    asm volatile("" : "+v"(krem), "+v"(kb32));
    float enr[2][16];
    auto request_en = [&](int c, float (&dst)[16]) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 v = en_lds[c * 8 + 2 * g + h];
            dst[4 * g] = v.x; dst[4 * g + 1] = v.y; dst[4 * g + 2] = v.z; dst[4 * g + 3] = v.w;
        }
    };
    if (VQ_IS_L2(metric)) request_en(0, enr[0]);
    if (MODE == 0 && VQ_IS_L2(metric)) {
        // row argmin, L2, in the radicands (exact_tiled_kernel): smallest radicand with its lowest index and the runner-up value
        // in one pass; a runner-up within 2^-21 of the smallest (near-ties, equal radicands, NaN) sends the wave through the
        // per-code sqrt + key loop.  A code that does not exist has t = +inf: fmaxf(inf, tmin) = inf leaves t2 alone.
        float tmin = INFINITY, t2 = INFINITY, tsum = 0.0f;
        int omin = -1;
        // (a chunk that lies wholly inside the codebook — all but the last — runs the pass without the existence selects)
        auto radicand_pass = [&](auto whole_chunk) {
            constexpr bool WHOLE = decltype(whole_chunk)::value;
#pragma unroll
            for (int c = 0; c < CT; ++c) {
                if (c + 1 < CT) request_en(c + 1, enr[(c + 1) & 1]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const int o = c * 32 + mfma_row(q, 0);
                    const bool kv = WHOLE || o < krem;
                    float t = (acc[c][q] + xn) + (kv ? enr[c & 1][q] : 0.0f);
                    t = (t < 0.0f) ? 0.0f : t;
                    t = kv ? t : INFINITY;
                    const bool upd = t < tmin;
                    t2 = fminf(t2, fmaxf(t, tmin));
                    tsum += t;                                   // NaN radicands (a code row holding inf or NaN): see below
                    omin = upd ? o : omin;
                    tmin = upd ? t : tmin;
                }
            }
        };
        if (kbase + CT * 32 <= K) radicand_pass(std::true_type{}); else radicand_pass(std::false_type{});
        // A NaN radicand must win (torch.argmin: NaN first) and is invisible to the comparisons above — fmaxf(NaN, tmin) = tmin
        // only flags it while tmin does not fall any further.  The radicands are >= 0 or NaN, so their sum is NaN exactly when
        // one of them is (round 6: a codebook row holding +inf, distances inf - inf for half the rows, lost to a NaN row of
        // higher index in a later chunk — both forms of the pass)
        const bool unique = t2 > tmin * (1.0f + 0x1p-21f) && tsum == tsum;   // (inf > inf is false: equal / all-inf radicands are not unique)
        // (no code selected although the lane has codes: every radicand is NaN or +inf — the exact loop sorts that out)
        if (__any(omin >= 0 ? !unique : krem > 0)) {
            float xn2 = xn;
            asm volatile("" : "+v"(xn2));     // the radicands are computed AGAIN: sharing them with the pass above would keep 128 values alive
            request_en(0, enr[0]);
#pragma unroll
            for (int c = 0; c < CT; ++c) {
                if (c + 1 < CT) request_en(c + 1, enr[(c + 1) & 1]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const int o = c * 32 + mfma_row(q, 0);
                    float t = (acc[c][q] + xn2) + ((o < krem) ? enr[c & 1][q] : 0.0f);
                    t = (t < 0.0f) ? 0.0f : t;
                    if (o < krem) { const u64 key = dist_key(sqrtf(t), kb32 + (uint32_t)o); best = key < best ? key : best; }
                }
            }
        } else if (omin >= 0) {
            best = dist_key(sqrtf(tmin), kb32 + (uint32_t)omin);
        }
    } else if (MODE == 2 && HW > 0 && (K & 3) == 0) {
        constexpr int HWS = HW > 0 ? HW : 32;                // (HW == 0 never reaches this branch)
        constexpr int NPR = HWS / 4, RPP = 64 / NPR;         // 16-byte pieces of a staged row; rows per read pass
        const int lane = j + 32 * h;
        const int rr = lane / NPR, rp = lane % NPR;
        const int64_t row0 = row - j;                        // the wave's first row
#pragma unroll
        for (int c = 0; c < CT; ++c) {
            if (VQ_IS_L2(metric) && c + 1 < CT) request_en(c + 1, enr[(c + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
            float dv[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int o = c * 32 + mfma_row(q, 0);
                if (VQ_IS_L2(metric)) {
                    float t = (acc[c][q] + xn) + ((o < krem) ? enr[c & 1][q] : 0.0f);
                    t = (t < 0.0f) ? 0.0f : t;
                    dv[q] = sqrtf(t);
                } else {
                    dv[q] = cos_distance(acc[c][q], metric);
                }
            }
#pragma unroll
            for (int hh = 0; hh < 32 / HWS; ++hh) {           // the tile's halves (HW == 16) or the whole tile
#pragma unroll
                for (int gl = 0; gl < HWS / 8; ++gl) {        // this lane's runs of four codes: 8 g + 4 h .. + 3 of the tile
                    const int g = hh * (HWS / 8) + gl, pc = 2 * gl + h;
                    stage[j * NPR + (pc ^ (j & (NPR - 1)))] = make_float4(dv[4 * g], dv[4 * g + 1], dv[4 * g + 2], dv[4 * g + 3]);
                }
#pragma unroll
                for (int pass = 0; pass < 32 / RPP; ++pass) {
                    const int r = pass * RPP + rr;
                    const float4 v = stage[r * NPR + (rp ^ (r & (NPR - 1)))];
                    const int64_t grow = row0 + r, k = kbase + c * 32 + hh * HWS + 4 * rp;
                    if (grow < N) {
                        float *dst = dout + grow * K + k;
                        if (k + 3 < K) *(float4 *)dst = v;
                        else { if (k < K) dst[0] = v.x; if (k + 1 < K) dst[1] = v.y; if (k + 2 < K) dst[2] = v.z; }
                    }
                }
            }
        }
    } else {
        float *dp = MODE == 2 ? dout + row * K + kb : nullptr;
#pragma unroll
        for (int c = 0; c < CT; ++c) {
            if (VQ_IS_L2(metric) && c + 1 < CT) request_en(c + 1, enr[(c + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int o = c * 32 + mfma_row(q, 0);
                const bool kv = o < krem;
                float d;
                if (VQ_IS_L2(metric)) {
                    float t = (acc[c][q] + xn) + (kv ? enr[c & 1][q] : 0.0f);
                    t = (t < 0.0f) ? 0.0f : t;
                    d = sqrtf(t);
                } else {
                    d = cos_distance(acc[c][q], metric);
                }
                if (MODE == 0) {
                    if (kv) { u64 key = dist_key(d, kb32 + (uint32_t)o); best = key < best ? key : best; }
                } else if (MODE == 1) {
                    u64 key = (rvalid && kv) ? dist_key(d, (uint32_t)row) : ~0ull;
#pragma unroll
                    for (int off = 16; off >= 1; off >>= 1) {
                        u64 o2 = __shfl_xor(key, off, 64);
                        key = o2 < key ? o2 : key;
                    }
                    if (j == 0 && kv && key != ~0ull) atomicMin(&keys[kb + o], key);
                } else {
                    if (rvalid && kv) dp[o] = d;
                }
            }
        }
    }
    return best;
}

// Whole-batch fp32 pass (argmin_exact, col_argmin, distance): a workgroup = 4 waves x 32 rows against a chunk of CT code
// tiles.  Code tiles are staged through LDS once per workgroup (coalesced float4 loads, register prefetch of the next
// tile, 16-byte XOR swizzle -> conflict-free ds_read_b128) and shared by the 4 waves; accumulators of all CT tiles stay
// live so that the row fragments are loaded once per 256-dim block.  Same k-ordered fma chains as exact_kernel.
// VEC4 (D % 4 == 0): operands move as whole 16-byte pieces at clamped addresses with zeros selected in past D / K, and the only
// branches are one per group of 8 pieces (32 dims) — straight-line groups of 16 MFMAs.  The element-wise tail form (any D) is
// a separate instantiation: in one kernel its per-piece conditions put every MFMA pair into a basic block of its own and
// hipcc hoisted some 250 loop-invariant comparisons in front of the item loop (25 000 instructions, 1 150 v_readlane).
template <int DT, int MODE, bool VEC4>
__global__ __launch_bounds__(256) void exact_tiled_kernel(const void *__restrict__ x, const float *__restrict__ e,
                                                          const float *__restrict__ en_in, const float *__restrict__ xn_in,
                                                          int64_t N, int64_t K, int D, int metric, u64 *__restrict__ keys,
                                                          float *__restrict__ dout) {
    constexpr int CT = 8;                        // code tiles (32 codes) per work item
    constexpr int DB = 128;                      // dims per register / LDS block
    constexpr int NPRE = 32 * (DB / 4) / 256;    // 16-byte chunks of a tile per thread
    extern __shared__ __attribute__((aligned(16))) char lds[];
    float4 *tile = (float4 *)lds;                // [2][32 rows][DB/4 chunks], chunk index XOR (row & 15)
    float4 *en_lds = tile + 2 * 32 * (DB / 4);   // |e_k|^2 of the item's CT * 32 codes (L2), parked at the start of the item
    constexpr int CPR = DB / 4;                  // chunks per row
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 31, h = lane >> 5;
    const int64_t nrb = (N + 127) / 128;
    const int64_t nchunks = (K + CT * 32 - 1) / (CT * 32);
    const float sx = (VQ_IS_L2(metric)) ? -2.0f : 1.0f;

    for (int64_t item = blockIdx.x; item < nrb * nchunks; item += gridDim.x) {
        const int64_t rb = item / nchunks, chunk = item % nchunks;
        const int64_t row = rb * 128 + wave * 32 + j;
        const bool rvalid = row < N;
        const int64_t rrow = rvalid ? row : N - 1;
        const int64_t kbase = chunk * CT * 32;
        // |e_k|^2 of the item's codes: one 16-byte piece per thread, requested here, parked in LDS behind the first barrier
        // below (the previous item's epilogue may still be reading) and read by the epilogue behind the barriers of the tile
        // loop — by then the round trip is long over (requested in the epilogue itself, a tile ahead, the norms cost 8 exposed
        // round trips per item).  en_in holds (K + 63) / 64 * 64 floats (vq_ws_layout).
        const bool en_owner = VQ_IS_L2(metric) && (int)threadIdx.x < CT * 8;
        float4 en_pre = make_float4(0.f, 0.f, 0.f, 0.f);
        if (en_owner) {
            const int64_t k0 = kbase + 4 * (int)threadIdx.x, kpad = (K + 63) / 64 * 64;
            en_pre = *(const float4 *)(en_in + (k0 < kpad ? k0 : 0));
        }
        f32x16 acc[CT];
#pragma unroll
        for (int c = 0; c < CT; ++c)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[c][q] = 0.0f;

        for (int db = 0; db < D; db += DB) {
            // B fragments: lane (row j, k-parity h) holds sx * x[row][db + 2s + h], s = 0..DB/2-1
            float xfr[DB / 2];
            if constexpr (VEC4) {
                // every piece of the block requested before the first one is used (a group that waited for its own 8 loads
                // before the next group's went out cost 17 000 cycles per block, profiles/r03_exact_rows.txt)
#pragma unroll
                for (int half = 0; half < 2; ++half) {       // two passes of 16 pieces (64 registers of requests in flight)
                    float4 xp[DB / 8];
#pragma unroll
                    for (int g = 0; g < DB / 64; ++g) {
                        if (db + 32 * (2 * half + g) < D) {
#pragma unroll
                            for (int i = 0; i < 8; ++i) {
                                const int s4 = 8 * (2 * half + g) + i, d = db + 4 * s4;
                                const int64_t off = rrow * D + (d < D ? d : 0);
                                if (DT == 0) {
                                    xp[8 * g + i] = *(const float4 *)((const float *)x + off);
                                } else {
                                    const uint2 t = *(const uint2 *)((const uint16_t *)x + off);
                                    xp[8 * g + i] = float4{__uint_as_float(t.x), __uint_as_float(t.y), 0.0f, 0.0f};
                                }
                            }
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int g = 0; g < DB / 64; ++g) {
                        if (db + 32 * (2 * half + g) < D) {
#pragma unroll
                            for (int i = 0; i < 8; ++i) {
                                const int s4 = 8 * (2 * half + g) + i, d = db + 4 * s4;
                                const float4 p = xp[8 * g + i];
                                float v0, v1, v2, v3;
                                if (DT == 0) {
                                    v0 = p.x; v1 = p.y; v2 = p.z; v3 = p.w;
                                } else {
                                    const uint32_t t0 = __float_as_uint(p.x), t1 = __float_as_uint(p.y);
                                    v0 = __uint_as_float(t0 << 16); v1 = __uint_as_float(t0 & 0xFFFF0000u);
                                    v2 = __uint_as_float(t1 << 16); v3 = __uint_as_float(t1 & 0xFFFF0000u);
                                }
                                xfr[2 * s4] = sx * (d < D ? (h ? v1 : v0) : 0.0f);
                                xfr[2 * s4 + 1] = sx * (d < D ? (h ? v3 : v2) : 0.0f);
                            }
                        }
                    }
                }
            } else {
#pragma unroll
            for (int s4 = 0; s4 < DB / 4; ++s4) {
                const int d = db + 4 * s4;
                float v0 = 0, v1 = 0, v2 = 0, v3 = 0;
                if (d < D) {
                    if (d + 3 < D && (D % 4) == 0) {
                        if (DT == 0) {
                            float4 t = *(const float4 *)((const float *)x + rrow * D + d);
                            v0 = t.x; v1 = t.y; v2 = t.z; v3 = t.w;
                        } else {
                            uint2 t = *(const uint2 *)((const uint16_t *)x + rrow * D + d);
                            v0 = __uint_as_float(t.x << 16); v1 = __uint_as_float(t.x & 0xFFFF0000u);
                            v2 = __uint_as_float(t.y << 16); v3 = __uint_as_float(t.y & 0xFFFF0000u);
                        }
                    } else {
                        v0 = load_elem<DT>(x, rrow * D + d);
                        if (d + 1 < D) v1 = load_elem<DT>(x, rrow * D + d + 1);
                        if (d + 2 < D) v2 = load_elem<DT>(x, rrow * D + d + 2);
                        if (d + 3 < D) v3 = load_elem<DT>(x, rrow * D + d + 3);
                    }
                }
                xfr[2 * s4] = sx * (h ? v1 : v0);
                xfr[2 * s4 + 1] = sx * (h ? v3 : v2);
            }
            }
            // staging: thread t owns the 16-byte chunks t, t+256, ... of the 32 x DB tile
            float4 pre[NPRE];
            auto fetch = [&](int ct) {
#pragma unroll
                for (int i = 0; i < NPRE; ++i) {
                    const int c = threadIdx.x + 256 * i;
                    const int r = c / CPR, ch = c % CPR;
                    const int64_t k = kbase + ct * 32 + r;
                    const int d = db + 4 * ch;
                    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                    if constexpr (VEC4) {
                        const float4 t = *(const float4 *)(e + (k < K ? k : 0) * D + (d < D ? d : 0));
                        if (k < K && d < D) v = t;
                    } else if (k < K && d < D) {
                        if (d + 3 < D && (D % 4) == 0) v = *(const float4 *)(e + k * D + d);
                        else {
                            v.x = e[k * D + d];
                            if (d + 1 < D) v.y = e[k * D + d + 1];
                            if (d + 2 < D) v.z = e[k * D + d + 2];
                            if (d + 3 < D) v.w = e[k * D + d + 3];
                        }
                    }
                    pre[i] = v;
                }
            };
            auto stash = [&](int buf) {
#pragma unroll
                for (int i = 0; i < NPRE; ++i) {
                    const int c = threadIdx.x + 256 * i;
                    const int r = c / CPR, ch = c % CPR;
                    tile[(buf * 32 + r) * CPR + (ch ^ (r & 15))] = pre[i];
                }
            };
            __syncthreads();            // previous block / item is done with both buffers
            if (db == 0 && en_owner) en_lds[threadIdx.x] = en_pre;
            fetch(0);
            stash(0);
            __syncthreads();
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                if (ct + 1 < CT) fetch(ct + 1);
                const float4 *trow = tile + ((ct & 1) * 32 + j) * (DB / 4);
                if constexpr (VEC4) {
                    // zero pieces past D (both operands) add +-0 to accumulators that are never -0: the chain is unchanged
                    // the tile's 16-byte pieces are read VQ_EXACT_PF pairs of MFMAs ahead of their use, through a ring of register
                    // sets, and every pair's statements stay together (sched_barrier): left alone, hipcc issues a piece's ds_read
                    // between the two MFMAs of the pair before and waits for it in front of the next pair — one wave per SIMD, 128
                    // cycles of MFMA against an LDS round trip: the pipe stood still a quarter of the time (round 6)
#ifndef VQ_EXACT_PF
#define VQ_EXACT_PF 2
#endif
#pragma unroll
                    for (int g = 0; g < DB / 32; ++g) {
                        if (db + 32 * g < D) {
                            constexpr int PF = VQ_EXACT_PF, RING = PF + 1;
                            float4 vr[RING];
#pragma unroll
                            for (int i = 0; i < PF; ++i) vr[i] = trow[(8 * g + i) ^ (j & 15)];
#pragma unroll
                            for (int i = 0; i < 8; ++i) {
                                const int q = 8 * g + i;
                                if (i + PF < 8) vr[(i + PF) % RING] = trow[(q + PF) ^ (j & 15)];
                                const float4 v = vr[i % RING];
                                acc[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(h ? v.y : v.x, xfr[2 * q], acc[ct], 0, 0, 0);
                                acc[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(h ? v.w : v.z, xfr[2 * q + 1], acc[ct], 0, 0, 0);
                                if constexpr (PF > 0) __builtin_amdgcn_sched_barrier(0);
                            }
                        }
                    }
                } else {
#pragma unroll
                for (int q = 0; q < DB / 4; ++q) {
                    if (db + 4 * q < D) {
                        const float4 v = trow[q ^ (j & 15)];
                        acc[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(h ? v.y : v.x, xfr[2 * q], acc[ct], 0, 0, 0);
                        acc[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(h ? v.w : v.z, xfr[2 * q + 1], acc[ct], 0, 0, 0);
                    }
                }
                }
                if (ct + 1 < CT) stash((ct + 1) & 1);
                __syncthreads();
            }
        }

        const float xn = (VQ_IS_L2(metric) && rvalid) ? xn_in[row] : 0.0f;
        u64 best = tiled_epilogue<MODE>(acc, en_lds, xn, kbase, K, metric, h, j, rvalid, row, keys, dout);
        if (MODE == 0) {
            u64 o = __shfl_xor(best, 32, 64);
            best = o < best ? o : best;
            if (h == 0 && rvalid && best != ~0ull) atomicMin(&keys[row], best);
        }
    }
}

// decode keys -> idx (+hist, +dmin).  rows = flagged list (device count) or all N
__global__ void finalize_kernel(const u64 *keys, const int *row_list, const int *nrows_dev, int64_t N, int64_t *idx,
                                float *dmin, int32_t *hist) {
    const int64_t nrows = row_list ? (int64_t)(*nrows_dev) : N;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nrows; i += (int64_t)gridDim.x * blockDim.x) {
        int64_t row = row_list ? (int64_t)row_list[i] : i;
        u64 key = keys[row];
        uint32_t k = (uint32_t)(key & 0xFFFFFFFFull);
        idx[row] = (int64_t)k;
        if (hist) atomicAdd(&hist[k], 1);
        if (dmin) {
            u64 hi = key >> 32;
            float d;
            if (hi == 0) d = __uint_as_float(0x7FC00000u);
            else {
                uint32_t b = (uint32_t)(hi - 1ull);
                b = (b & 0x80000000u) ? (b & 0x7FFFFFFFu) : ~b;
                d = __uint_as_float(b);
            }
            dmin[row] = d;
        }
    }
}

__global__ void fill_u64_kernel(u64 *p, int64_t n, u64 v) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) p[i] = v;
}
