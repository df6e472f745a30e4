// Buffer layouts shared by host entry points and device kernels (gfx950 only).
#pragma once
#include <stdint.h>

#define VQ_WAVE 64
#define VQ_TILE_CODES 32           // codes per tile (two v_mfma_f32_16x16x32_f16 row blocks)
#define VQ_CHUNK_BYTES 1024        // one wave-instruction of global_load_lds_dwordx4
#define VQ_AUX_CHUNKS(tps) (((tps) * 128 + VQ_CHUNK_BYTES - 1) / VQ_CHUNK_BYTES)   // aux values: 32 floats per code tile
#define VQ_MAX_SLICES 16
#define VQ_CB_SLOTS 16             // slots (128-byte lines behind the 256-byte header) the image kernel's blocks raise their maxima into
#define VQ_REC_FIELDS 5            // v1, c1, v2, c2, v3

#if defined(__HIPCC__)
#define VQ_HD __host__ __device__ inline
#else
#define VQ_HD inline
#endif

// Padded inner dimension for the fp16 proposal pass (k-steps of 32): next power of two >= max(D,32) up to 512,
// then 768 and 1024 (one tile per LDS stage).
VQ_HD int vq_padded_d(int D) {
    if (D > 512) return D <= 768 ? 768 : 1024;
    int p = 32;
    while (p < D) p <<= 1;
    return p;
}
VQ_HD int vq_coarse_supported(int D) { return D >= 1 && D <= 1024 && (D % 8) == 0; }
// tiles (32 codes) staged per LDS stage
// (the proposal kernel keeps a ring of four stages up to D = 256: 4 x 33 KiB at D = 256, hence two tiles per stage there)
#define VQ_TPS16 2
#ifndef VQ_TPS_D32
#define VQ_TPS_D32 8               // D <= 32 (one 32-dim k-step): tiles per stage (the aux chunk holds at most 8 tiles' biases)
#endif
VQ_HD int vq_tiles_per_stage(int nstep) { return nstep <= 2 ? VQ_TPS_D32 : (nstep < 16 ? 4 : (nstep <= 32 ? 2 : 1)); }

struct VqCbLayout {
    int64_t K, Kp;          // codes, codes padded to a whole stage
    int D, Dp, nstep, tps;  // dims, padded dims, k-steps of 16, tiles per stage
    int64_t nstages;
    int64_t stage_bytes;    // (tps*nstep + aux) KiB: fragment chunks + the aux chunk(s) (-se*|e|^2/2, 32 floats per tile)
    int64_t off_stats, off_part1, off_part2, off_en, off_eexact, off_frag, total;
    int64_t nblk1, nblk2;   // blocks of the statistics / image kernels (their partial-result arrays)
};

VQ_HD VqCbLayout vq_cb_layout(int64_t K, int D) {
    VqCbLayout L;
    L.K = K; L.D = D;
    L.Dp = vq_padded_d(D); L.nstep = L.Dp / 16; L.tps = vq_tiles_per_stage(L.nstep);
    int64_t cps = (int64_t)L.tps * VQ_TILE_CODES;
    L.nstages = (K + cps - 1) / cps;
    L.Kp = L.nstages * cps;
    L.stage_bytes = ((int64_t)L.tps * L.nstep + VQ_AUX_CHUNKS(L.tps)) * VQ_CHUNK_BYTES;
    L.off_stats = 0;
    L.nblk1 = (K + 15) / 16;                 // cb_stats_kernel: 16 codes per block
    L.nblk2 = L.nstages * L.tps;             // cb_image_kernel: one tile per block
    // [256, 256 + 16*128): VQ_CB_SLOTS maxima slots of cb_image_kernel, one 128-byte line each (see cb_stats_view)
    L.off_part1 = 256 + VQ_CB_SLOTS * 128;   // float4 {max|e|, max e2, L2: -(min |e|^2) / else 0, bad} per stats block
    L.off_part2 = L.off_part1 + L.nblk1 * 16;  // cosine: one float4 per image tile (cb_cos_body); unused for L2 / DOT
    L.off_en = (L.off_part2 + L.nblk2 * 16 + 255) / 256 * 256;
    L.off_eexact = (L.off_en + L.Kp * 4 + 255) / 256 * 256;
    L.off_frag = (L.off_eexact + K * (int64_t)D * 4 + 1023) / 1024 * 1024;
    L.total = L.off_frag + L.nstages * L.stage_bytes;
    return L;
}

// Device-resident statistics of a prepared codebook (first 256 bytes of the image).
struct VqCbStats {
    uint32_t maxabs_bits;   // max |e_kd| (of the normalised codebook for COS), float bits
    uint32_t e2max_bits;    // max_k sum_d e_kd^2                (upper-bounds |e_k|^2)
    uint32_t r2max_bits;    // max_k sum_d (e_kd - ehat_kd)^2    (fp16 residual)
    uint32_t eh2max_bits;   // max_k sum_d ehat_kd^2
    uint32_t enmax_bits;    // max_k oracle |e_k|^2 (0 for COS)
    uint32_t nonfinite;     // !=0: some entry is NaN/Inf (or overflows the fp16 image)
    int32_t metric;
    uint32_t finalized;     // the image kernel has run (r2max/eh2max/nonfinite are raised by it with filtered atomics)
    uint32_t en_spread_bits;  // L2, constant-norm codebook (below): max_k |e_k|^2 - min_k |e_k|^2, else 0
    uint32_t folded;          // cosine: the partials below have been folded into r2max / eh2max / e2max / nonfinite of this header by
                              // the proposal kernel of the first argmin on the image (cb_stats_publish): later launches read the header
    uint32_t part2_n;         // cosine (one-launch preparation, cb_cos_body): the image's per-tile maxima are NOT in the slots
    uint32_t part2_off;       // but in part2_n float4 partials {max residual^2, max image norm^2, bad, max |e_hat|^2} at this byte
                              // offset from the header (no zeroing launch needed in front); 0: the slots (L2 / DOT)
    uint32_t l2_const_norm;   // L2 and the norms agree to 2^-16 relative (a NormalizeCallback codebook): the proposal scores
                              // carry NO -|e_k|^2/2 term (aux values of real codes are 0) and the row margin carries the
                              // spread instead — the aux reads were a third of the D <= 16 kernel's time (vqhip_proposal32_kernels.h)
};

struct VqWsLayout {
    int64_t N;
    int64_t off_counters, off_xh2, off_rho2, off_rec, off_flag, off_multi, off_exact, off_thr, off_rcnt, off_rlist, off_keys, off_en, off_xn, off_arrive, off_ximg, total;
    // group identification of the D <= 32 proposal kernels (vqhip_proposal32_kernels.h, identify32_kernel); zero-sized otherwise
    int64_t narrive;        // ints in the zeroed counter range at off_arrive: arrival counters, then the nbkt bucket counters
    int64_t nbkt;           // bucket counters: one per 32-code tile of the padded codebook
    int64_t off_bcnt, off_rece2, off_blist, off_bfrag, blist_entries;
};

// the D <= 32 group path: the stream kernel files one identification request per (token, slice, lane half) under the code
// tile of the lane's best group; identify32_kernel serves them bucket by bucket.  Up to VQ_GROUP_MAX_SLICES slices and
// VQ_GROUP_MAX_TILES code tiles (K <= 131 072): beyond, the per-element kernels are used.
// (four slices since round 5: 16 384 rows x 16 384 codes x 8 dims -12 %, nothing lost from 32 768 rows on, where fewer are picked
//  anyway: profiles/r05_ab_small_d.txt)
#ifndef VQ_GROUP_MAX_SLICES
#define VQ_GROUP_MAX_SLICES 4
#endif
#define VQ_GROUP_MAX_TILES 4096
// A group's requests are spread over R buckets (by the token block that files them; R = the largest power of two <= 128 with
// groups * R <= VQ_GROUP_MAX_BUCKETS): 100 352 returning atomics on the 256 counters of BASELINE configs[2] — eight cache
// lines — took 48 us at ~4 ns per atomic and line; with a line per counter and eight buckets per tile they take none
#define VQ_GROUP_MAX_BUCKETS 4096
#define VQ_GROUP_CNT_STRIDE 32         // ints between two bucket counters (128 bytes)
#ifndef VQ_GROUP_TILES
#define VQ_GROUP_TILES 4               // code tiles per group record of coarse32_kernel (a divisor of VQ_TPS_D32; 1, 2, 4, 8 measured: profiles/r04_group_tiles.txt)
#endif
#ifndef VQ_GROUPS_MIN_N
#define VQ_GROUPS_MIN_N 16384          // group records on the 16x16x32 form (in-kernel replay) from this many rows on
#endif
// ... and on the 32x32x16 form (coarse32_kernel + identify32_kernel, the request lists of this workspace) from this many: measured
// against the per-element form it replaces below 16 384 rows (profiles/r05_ab_small_d.txt, one-call encodes): 8192 x 8192 x 32
// cosine 0.066 -> 0.061 ms, 12 544 x 8192 x 32 (the VQ-KD per-rank batch) 0.077 -> 0.065, 8192 x 16384 x 8 0.077 -> 0.069; at 4096
// rows the per-element form — compiled for two waves per SIMD there, which ended its register spills — stays ahead
#ifndef VQ_W32_MIN_N
#define VQ_W32_MIN_N 8192
#endif
VQ_HD int vq_group_replicas(int64_t ngroups) { int r = 128; while (r > 1 && ngroups * r > VQ_GROUP_MAX_BUCKETS) r >>= 1; return r; }
VQ_HD bool vq_group_path_possible(int64_t K, int D) {
    return vq_coarse_supported(D) && D <= 32 && (K + VQ_TILE_CODES - 1) / VQ_TILE_CODES + VQ_TPS_D32 <= VQ_GROUP_MAX_TILES;
}

// counters: [0] rescanned rows, [1] rows with >1 identified candidate, [2] rows sent to the fp32 pass
VQ_HD VqWsLayout vq_ws_layout(int64_t N, int64_t K, int D) {
    VqWsLayout W;
    W.N = N;
    int64_t M = N > K ? N : K;      // col_argmin keys are per code
    int64_t Np = (N + 63) / 64 * 64;
    int64_t Mp = (M + 63) / 64 * 64;
    W.off_counters = 0;
    W.off_xh2 = 256;
    W.off_rho2 = W.off_xh2 + Np * 4;
    W.off_rec = W.off_rho2 + Np * 4;
    W.off_flag = W.off_rec + (int64_t)VQ_MAX_SLICES * VQ_REC_FIELDS * Np * 4;
    W.off_multi = W.off_flag + Np * 4;       // off_flag: rescan list
    W.off_exact = W.off_multi + Np * 4;
    W.off_thr = W.off_exact + Np * 4;
    W.off_rcnt = W.off_thr + Np * 4;
    W.off_rlist = W.off_rcnt + Np * 4;       // [Np][32] candidate codes of rescanned rows
    W.off_keys = (W.off_rlist + Np * 4 * 32 + 255) / 256 * 256;
    W.off_en = W.off_keys + Mp * 8;          // K floats: oracle |e_k|^2 for the fp32-only entry points
    W.off_xn = W.off_en + (K + 63) / 64 * 64 * 4;   // oracle-order |x_n|^2 of every row (x_prep_kernel)
    W.off_arrive = W.off_xn + Np * 4;          // arrival counters of the proposal kernel's token blocks (>= 128 tokens each)
    // the request lists exist only where the proposal pass can take the group path: enough rows (launch_coarse), and then one
    // counter per bucket the codebook really has — not 4096 x 128 bytes zeroed by every call's first launch whatever the shape
    const bool grp = vq_group_path_possible(K, D) && N >= VQ_W32_MIN_N;
    int64_t nbuckets = 0;
    if (grp) {
        const int64_t ngroups = ((K + (int64_t)VQ_TPS_D32 * VQ_TILE_CODES - 1) / ((int64_t)VQ_TPS_D32 * VQ_TILE_CODES)) * VQ_TPS_D32 / VQ_GROUP_TILES;
        nbuckets = ngroups * vq_group_replicas(ngroups);
    }
    W.nbkt = nbuckets * VQ_GROUP_CNT_STRIDE;                                    // one counter per bucket, each on a 128-byte line of its own
    W.off_bcnt = (W.off_arrive + (Np / 128 + 8) * 4 + 127) / 128 * 128;
    W.narrive = (W.off_bcnt - W.off_arrive) / 4 + W.nbkt;
    // second-best value inside an identified group, per (slice, lane half, token): the consumer folds it into the bound v3
    W.off_rece2 = (W.off_bcnt + W.nbkt * 4 + 255) / 256 * 256;
    // request lists: an equal share of one pool per bucket (4 N + 64 per bucket entries in all: >= 4x the mean at one slice);
    // an entry is the token word and, in a second array, the token's B fragment as the requesting lanes hold it (32 bytes
    // per 16 dims: identify32_kernel reads a batch's 32 fragments as 1-2 KiB of contiguous bytes instead of gathering 64
    // scattered 16-byte pieces from the token image — 35 -> 9 us at 524 288 requests)
    W.blist_entries = grp ? 4 * Np + 64 * nbuckets : 0;
    W.off_blist = W.off_rece2 + (grp ? (int64_t)VQ_GROUP_MAX_SLICES * 2 * Np * 4 : 0);
    W.off_bfrag = (W.off_blist + W.blist_entries * 4 + 255) / 256 * 256;
    const int64_t frag_bytes = W.blist_entries * (D <= 16 ? 32 : 64);
    W.off_ximg = (W.off_bfrag + frag_bytes + 1023) / 1024 * 1024;   // fp16 token image [N/32][nstep] KiB
    const int64_t img = vq_coarse_supported(D) ? ((N + 31) / 32) * (int64_t)(vq_padded_d(D) / 16) * VQ_CHUNK_BYTES : 0;
    W.total = W.off_ximg + img;
    return W;
}
