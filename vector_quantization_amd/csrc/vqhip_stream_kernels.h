// libvqhip device kernels, unit 5b: the all-fp32 MFMA pass over whole batches, streamed form (round 6).
// Included by vqhip_kernels.h behind vqhip_exact_kernels.h (whose epilogue rules — l2_skip's monotone sqrt, dist_key — it shares).
#pragma once
// ------------------------------------------------------------------------------------------------
// exact_stream_kernel: exact_tiled_kernel's work item (128 rows x 256 codes, the same k-ordered chains of
// v_mfma_f32_32x32x2_f32 per (row, code), the same epilogue) with BOTH operands read from LDS at the MFMA that consumes them.
//
// What the register form paid for (exact_tiled_kernel, 76 of a sustainable 154 TFLOP/s): the row fragments of a
// 128-dim block live in 64 registers, so (a) their loads — 32 uncoalesced float4 per lane — stand in front of every block with
// nothing to overlap them (6 us of a 55 us item), (b) with 128 accumulators and the staging registers of the code tile the kernel
// needs the whole register file: ONE wave per SIMD, and whatever that wave waits for — the first tile of a block, a barrier per
// code tile, its own 2 500-instruction epilogue — the matrix pipe waits for too.  A v_mfma_f32_32x32x2_f32 occupies the pipe for
// 64 cycles and takes 8 bytes per lane: reading A AND B from LDS costs 2 ds_read_b128 per 128 cycles and wave, a quarter of
// the LDS array's 256 B/clk with four SIMDs busy.  So here nothing but the accumulators stays in registers:
//  * blocks of 32 dims.  The row block (128 rows x 128 B; bf16 rows: x 64 B) and the code tile (32 codes x 128 B) arrive by
//    LDS-DMA in 16-byte pieces, XOR-swizzled by row so that every ds_read_b128 lane group meets 16 distinct slots (mod 16):
//    piece p of row r sits at slot r * 8 + (p ^ ((r >> 1) & 7)) (bf16 rows, 4 pieces: r * 4 + (p ^ ((r >> 2) & 3)));
//  * a stage = (block, VQ_XS_TS = 4 code tiles): four independent accumulator chains interleaved MFMA by MFMA, one B operand for
//    the four, 64 MFMAs = 4096 matrix cycles between barriers.  The DMA of the next stage's code tiles (and, at the first stage of
//    a block, of the next block's rows) is requested at the top of the stage and drained IN FRONT OF the barrier at the top of the
//    next (vq_dma_barrier: the s_waitcnt vmcnt(0) is written out — __syncthreads() does not promise it): 65 KiB of LDS per
//    workgroup with fp32 rows (two row blocks, two stages of code tiles, the item's |e|^2), 49 KiB with bf16 rows;
//  * 128 accumulators + two small operand rings: under 256 registers, TWO workgroups per CU.  The second wave of a SIMD runs
//    its MFMAs through the other's barrier, LDS latency at the top of a stage, and epilogue;
//  * a workgroup takes a CONTIGUOUS span of the work items (row-block-major; column argmin: chunk-major), keeps the lane's best key
//    across the chunks of a row block and sends one atomicMin per row and row block: per-item atomics on keys shared by 32
//    concurrently running workgroups were the largest single loss of the first streamed form;
//  * the stream does not stop at an item's end: the first stage of the workgroup's next item is requested during the last stage.
// Shapes: D % 4 == 0 for fp32 rows, D % 8 == 0 for bf16 rows (whole 16-byte pieces); the tail block of a D that is not a multiple
// of 32 runs its valid pieces only (exact_tiled_kernel adds zeros for the rest: +-0 into accumulators that are never -0).
// Everything else keeps exact_tiled_kernel.
// ------------------------------------------------------------------------------------------------
#ifndef VQ_XS_PF
#define VQ_XS_PF 1                  // pieces read ahead of their MFMAs inside a stage
#endif

template <int DT>
__device__ __forceinline__ void xs_b_pair(const float4 &b, int piece_in_x, int h, int sh, float sx, float &b0, float &b1) {
    if (DT == 0) {
        b0 = sx * (h ? b.y : b.x);
        b1 = sx * (h ? b.w : b.z);
    } else {
        // a 16-byte piece of bf16 rows is 8 dims = two 4-dim pieces of the code tile: words (x, y) then (z, w);
        // dim 2 m + h of a word pair: low half for h = 0 (shift up), high half for h = 1 (mask)
        const uint32_t w0 = __float_as_uint(piece_in_x ? b.z : b.x), w1 = __float_as_uint(piece_in_x ? b.w : b.y);
        b0 = sx * __uint_as_float((w0 << sh) & 0xFFFF0000u);
        b1 = sx * __uint_as_float((w1 << sh) & 0xFFFF0000u);
    }
}

// LDS byte addresses (address space 3 pointers are 32-bit offsets).  The piece p of a row whose swizzle is z sits at
// row base + 16 (p ^ z) = (row base | 16 z) ^ 16 p — row bases are multiples of 128 (64: bf16 rows) — so a stage walks its
// pieces with ONE address register per operand and an XOR per piece; the 16 precomputed piece addresses that hipcc otherwise
// keeps across the whole kernel were what pushed the tile loop's own operands into scratch (4 reloads per stage, each behind the
// stage's LDS-DMA in vmcnt order).
typedef float xs_v4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) const xs_v4 *xs_lds_f4;
__device__ __forceinline__ uint32_t xs_lds_addr(const void *p) { return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void *)p; }
__device__ __forceinline__ float4 xs_lds_read(uint32_t a) { const xs_v4 v = *(xs_lds_f4)(uintptr_t)a; return float4{v.x, v.y, v.z, v.w}; }

// one full stage: the 8 pieces (32 dims) of row j of TS code tiles against the wave's row j — TS independent accumulator chains,
// interleaved MFMA by MFMA, one B operand for all of them.  ea / xa: the lane's swizzled row addresses (see above).
template <int DT, int TS>
__device__ __forceinline__ void xs_stage_full(uint32_t ea, uint32_t xa, int h, int sh, float sx, f32x16 *acc) {
    constexpr int PF = VQ_XS_PF, RING = PF + 1;
    asm volatile("" : "+v"(ea), "+v"(xa));       // per stage: nothing derived from them is worth keeping across stages
    float4 ar[RING][TS], br[DT ? 4 : RING];
    if (DT) {
#pragma unroll
        for (int i = 0; i < 4; ++i) br[i] = xs_lds_read(xa ^ (i << 4));         // bf16 rows: the block's four pieces up front
    }
#pragma unroll
    for (int i = 0; i < PF; ++i) {
#pragma unroll
        for (int t = 0; t < TS; ++t) ar[i][t] = xs_lds_read((ea ^ (i << 4)) + t * 4096);
        if (!DT) br[i] = xs_lds_read(xa ^ (i << 4));
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        if (i + PF < 8) {
#pragma unroll
            for (int t = 0; t < TS; ++t) ar[(i + PF) % RING][t] = xs_lds_read((ea ^ ((i + PF) << 4)) + t * 4096);
            if (!DT) br[(i + PF) % RING] = xs_lds_read(xa ^ ((i + PF) << 4));
        }
        float b0, b1;
        xs_b_pair<DT>(DT ? br[i >> 1] : br[i % RING], i & 1, h, sh, sx, b0, b1);
#pragma unroll
        for (int t = 0; t < TS; ++t) { const float4 a = ar[i % RING][t]; acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(h ? a.y : a.x, b0, acc[t], 0, 0, 0); }
#pragma unroll
        for (int t = 0; t < TS; ++t) { const float4 a = ar[i % RING][t]; acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(h ? a.w : a.z, b1, acc[t], 0, 0, 0); }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// the tail block of a D that is not a multiple of 32: np valid 4-dim pieces (bf16 rows: np even)
template <int DT, int TS>
__device__ __forceinline__ void xs_stage_tail(uint32_t ea, uint32_t xa, int h, int sh, float sx, int np, f32x16 *acc) {
#pragma unroll 1
    for (int i = 0; i < np; ++i) {
        const float4 b = xs_lds_read(xa ^ ((DT ? i >> 1 : i) << 4));
        float b0, b1;
        xs_b_pair<DT>(b, i & 1, h, sh, sx, b0, b1);
#pragma unroll
        for (int t = 0; t < TS; ++t) {
            const float4 a = xs_lds_read((ea ^ (i << 4)) + t * 4096);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(h ? a.y : a.x, b0, acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(h ? a.w : a.z, b1, acc[t], 0, 0, 0);
        }
    }
}

#ifndef VQ_XS_TS
#define VQ_XS_TS 4                  // code tiles per stage
#endif
static inline int vq_xs_lds_bytes(int bf16_rows) { return (2 * 128 * (bf16_rows ? 4 : 8) + 2 * VQ_XS_TS * 256 + 64) * 16; }

template <int DT, int MODE>
__global__ __launch_bounds__(256, 2) void exact_stream_kernel(const void *__restrict__ x, const float *__restrict__ e,
                                                              const float *__restrict__ en_in, const float *__restrict__ xn_in,
                                                              int64_t N, int64_t K, int D, int metric, u64 *__restrict__ keys,
                                                              float *__restrict__ dout) {
    constexpr int CT = 8;                        // code tiles (32 codes) per work item
    constexpr int TS = VQ_XS_TS, NS = CT / TS;   // tiles per stage, stages per block
    constexpr int XP = DT ? 4 : 8;               // 16-byte pieces of a row per 32-dim block
    constexpr int XSL = 128 * XP;                // slots of a row block
    constexpr int ESL = TS * 256;                // slots of a stage's code tiles
    typedef __attribute__((address_space(1))) const void *gptr_t;
    typedef __attribute__((address_space(3))) void *lptr_t;
    extern __shared__ __attribute__((aligned(128))) char lds[];
    float4 *xbuf = (float4 *)lds;                // [2][128 rows][XP pieces]
    float4 *ering = xbuf + 2 * XSL;              // [2][TS tiles][32 codes][8 pieces]
    float4 *en_lds = ering + 2 * ESL;            // |e_k|^2 of the item's 256 codes (L2)
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 31, h = lane >> 5, sh = h ? 0 : 16;
    const uint32_t nchunks = (uint32_t)((K + CT * 32 - 1) / (CT * 32));
    const uint32_t nitems = (uint32_t)((N + 127) / 128) * nchunks;               // (the host keeps this below 2^31)
    const float sx = (VQ_IS_L2(metric)) ? -2.0f : 1.0f;
    const int nbf = D >> 5, npt = (D & 31) >> 2, nb = nbf + (npt ? 1 : 0);

    // LDS-DMA: a wave-instruction fills 64 consecutive slots, lane i the slot base + i from the address the lane names
    const int re = 8 * wave + (lane >> 3), pe = (lane & 7) ^ ((re >> 1) & 7);     // code-tile row and piece of this lane's slot
    auto issue_e = [&](int64_t kb, int cs, int db, int slot) {    // the TS tiles of stage cs of block db
        int d = db * 32 + 4 * pe;
        d = d < D ? d : 0;                                        // pieces past D are never read
#pragma unroll
        for (int t = 0; t < TS; ++t) {
            int64_t k = kb + (cs * TS + t) * 32 + re;
            k = k < K ? k : K - 1;                                // clamped: what lands there is masked by the epilogue
            __builtin_amdgcn_global_load_lds((gptr_t)(e + k * D + d), (lptr_t)(&ering[slot * ESL + t * 256 + 64 * wave]), 16, 0, 0);
        }
    };
    auto issue_x = [&](int64_t rb, int db, int buf) {             // every wave fetches its own 32 rows
#pragma unroll
        for (int u = 0; u < XP / 2; ++u) {
            const int r = DT ? 16 * u + (lane >> 2) : 8 * u + (lane >> 3);        // row within the wave's 32
            const int p = DT ? (lane & 3) ^ ((r >> 2) & 3) : (lane & 7) ^ ((r >> 1) & 7);
            int64_t row = rb * 128 + 32 * wave + r;
            row = row < N ? row : N - 1;
            int d = db * 32 + (DT ? 8 : 4) * p;
            d = d < D ? d : 0;
            const void *src = DT ? (const void *)((const uint16_t *)x + row * D + d) : (const void *)((const float *)x + row * D + d);
            __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(&xbuf[buf * XSL + 32 * wave * XP + 64 * u]), 16, 0, 0);
        }
    };
    auto issue_en = [&](int64_t kb) {                             // en_in holds (K + 63) / 64 * 64 floats (vq_ws_layout)
        if (wave == 0) {
            const int64_t k0 = kb + 4 * lane, kpad = (K + 63) / 64 * 64;
            __builtin_amdgcn_global_load_lds((gptr_t)(en_in + (k0 < kpad ? k0 : 0)), (lptr_t)en_lds, 16, 0, 0);
        }
    };

    // Work: items in the order that keeps the atomics apart.  A workgroup takes a CONTIGUOUS span of that order (spans differ by
    // at most one item).  Row argmin / distances: row block major — the chunks of one row block follow each other, the lane keeps
    // its best key across them and sends ONE atomicMin per row at the end of the row block (or of the span).  With the items
    // dealt round-robin (exact_tiled_kernel) the 32 chunks of a row block ran on 32 workgroups at the same time: 1024 atomics on
    // each 128-byte line of keys within microseconds, serialised at the memory side, and every wave sat out its own at the next
    // barrier's vmcnt(0) — 23 us of a 55 us item.  Column argmin (keys per code): chunk major, for the same reason.
    const uint32_t nrb = (uint32_t)((N + 127) / 128);
    const uint32_t it0 = (uint32_t)((uint64_t)nitems * blockIdx.x / gridDim.x), it1 = (uint32_t)((uint64_t)nitems * (blockIdx.x + 1) / gridDim.x);
    if (it0 >= it1) return;
    auto item_rb = [&](uint32_t it) { return MODE == 1 ? it % nrb : it / nchunks; };
    auto item_chunk = [&](uint32_t it) { return MODE == 1 ? it / nrb : it % nchunks; };
    int xb = 0, eb = 0;
    issue_x(item_rb(it0), 0, 0);
    issue_e((int64_t)item_chunk(it0) * (CT * 32), 0, 0, 0);
    // the lane's rows with their swizzle folded in (xs_stage_full): code-tile row j, the wave's row j
    const uint32_t ea_row = xs_lds_addr(ering + j * 8) | (uint32_t)(((j >> 1) & 7) << 4);
    const uint32_t xa_row = xs_lds_addr(xbuf + (32 * wave + j) * XP) | (uint32_t)((DT ? (j >> 2) & 3 : (j >> 1) & 7) << 4);
    u64 run_best = ~0ull;                        // MODE 0: the lane's best key over the chunks of the current row block

    for (uint32_t item = it0; item < it1; ++item) {
        const int64_t rb = item_rb(item), chunk = item_chunk(item);
        const int64_t row = rb * 128 + wave * 32 + j;
        const bool rvalid = row < N;
        const int64_t kbase = chunk * CT * 32;
        const bool has_next = item + 1 < it1;
        const int64_t rb2 = has_next ? item_rb(item + 1) : 0, kbase2 = has_next ? (int64_t)item_chunk(item + 1) * (CT * 32) : 0;
        // requested here, used by the epilogue (its round trip hides behind the item's MFMAs)
        const float xn = (VQ_IS_L2(metric) && rvalid) ? xn_in[row] : 0.0f;
        f32x16 acc[CT];
#pragma unroll
        for (int c = 0; c < CT; ++c)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[c][q] = 0.0f;

        // top of a stage: its pieces have landed (the barrier drains vmcnt), everybody is done with the previous stage's;
        // the next stage's code tiles — and at a block's first stage the next block's rows — are requested
        auto stage_top = [&](int db, int cs) {
            const bool lastb = db + 1 == nb;
#ifndef VQ_XS_NO_BAR         // (timing-only diagnostic builds: VQ_XS_NO_BAR / _NO_DMA / _NO_EPI / _NO_ATOMIC)
            vq_dma_barrier();    // (the drain written out: __syncthreads() alone does not promise it — vqhip_kernels.h)
#endif
#ifndef VQ_XS_NO_DMA
            if (cs + 1 < NS) issue_e(kbase, cs + 1, db, eb ^ 1);
            else if (!lastb) issue_e(kbase, 0, db + 1, eb ^ 1);
            else if (has_next) issue_e(kbase2, 0, 0, eb ^ 1);
            if (cs == 0) {
                if (db == 0 && VQ_IS_L2(metric)) issue_en(kbase);             // (everybody is past the previous item's epilogue)
                if (!lastb) issue_x(rb, db + 1, xb ^ 1);
                else if (has_next) issue_x(rb2, 0, xb ^ 1);
            }
#endif
#ifdef VQ_XS_SYNC_DMA       // (diagnostic: nothing in flight while a stage computes)
            __syncthreads();
#endif
        };
        for (int db = 0; db < nbf; ++db) {
#pragma unroll
            for (int cs = 0; cs < NS; ++cs) {
                stage_top(db, cs);
                xs_stage_full<DT, TS>(ea_row + eb * (ESL * 16), xa_row + xb * (XSL * 16), h, sh, sx, &acc[cs * TS]);
                eb ^= 1;
            }
            xb ^= 1;
        }
        if (npt) {
#pragma unroll
            for (int cs = 0; cs < NS; ++cs) {
                stage_top(nbf, cs);
                xs_stage_tail<DT, TS>(ea_row + eb * (ESL * 16), xa_row + xb * (XSL * 16), h, sh, sx, npt, &acc[cs * TS]);
                eb ^= 1;
            }
            xb ^= 1;
        }

#ifdef VQ_XS_NO_EPI
        {
            float sacc = 0.0f;
#pragma unroll
            for (int c = 0; c < CT; ++c)
#pragma unroll
                for (int q = 0; q < 16; q += 4) sacc += acc[c][q];
            if (sacc == 123.456f) keys[row] = 0;
            continue;
        }
#endif
        // ---- epilogue (tiled_epilogue, vqhip_exact_kernels.h: shared with the register form) ----
        // (distances leave through the wave's own 32 rows of the row block just consumed: the other one receives the next item's)
        const u64 best = tiled_epilogue<MODE, (MODE == 2 ? (DT ? 16 : 32) : 0)>(acc, en_lds, xn, kbase, K, metric, h, j, rvalid, row, keys, dout,
                                                                               xbuf + (xb ^ 1) * XSL + 32 * wave * XP, N);
        if (MODE == 0) {
            run_best = best < run_best ? best : run_best;
            if (!has_next || rb2 != rb) {                            // last chunk of the row block in this span (wave-uniform)
                u64 o = __shfl_xor(run_best, 32, 64);
                run_best = o < run_best ? o : run_best;
#ifndef VQ_XS_NO_ATOMIC
                if (h == 0 && rvalid && run_best != ~0ull) atomicMin(&keys[row], run_best);
#endif
                run_best = ~0ull;
            }
        }
    }
}
