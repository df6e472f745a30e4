// libvqhip — host entry points (C ABI in include/vqhip.h).  gfx950 only.
#include "vqhip.h"

#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>

#include "vqhip_kernels.h"

static thread_local char g_err[256] = "";

static int fail(int code, const char *what, const char *detail = "") {
    snprintf(g_err, sizeof(g_err), "%s%s%s", what, detail[0] ? ": " : "", detail);
    return code;
}

#define VQ_CHECK_LAUNCH(name)                                            \
    do {                                                                 \
        hipError_t err__ = hipGetLastError();                            \
        if (err__ != hipSuccess) return fail(VQHIP_ELAUNCH, name, hipGetErrorString(err__)); \
    } while (0)

#define VQ_HIP(call)                                                     \
    do {                                                                 \
        hipError_t err__ = (call);                                       \
        if (err__ != hipSuccess) return fail(VQHIP_ELAUNCH, #call, hipGetErrorString(err__)); \
    } while (0)

// ---- optional per-launch timing of the proposal kernel (bench.py roofline) -------------------------------
// Events are recorded on the caller's stream right around the coarse_kernel launch; vqhip_profile_collect
// synchronises on them.  Disabled (zero overhead) unless vqhip_profile_enable(1) was called.
// Process-wide state is limited to this block and the tuning knobs below; all of it is safe to touch from several host
// threads (each with its own stream): the profiling list is mutex-protected (the mutex is taken only while profiling is
// on), the knobs and the per-device attribute cache are atomics.
#include <atomic>
#include <mutex>
#include <vector>
static std::atomic<bool> g_prof_on{false};
static std::mutex g_prof_mu;
static std::vector<std::pair<hipEvent_t, hipEvent_t>> g_prof_events;
static size_t g_prof_used = 0;

// returns the slot the matching prof_end must close, or -1 when profiling is off
static long prof_begin(hipStream_t s) {
    if (!g_prof_on.load(std::memory_order_relaxed)) return -1;
    std::lock_guard<std::mutex> lock(g_prof_mu);
    if (g_prof_used == g_prof_events.size()) {
        hipEvent_t a, b;
        if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) { g_prof_on = false; return -1; }
        g_prof_events.emplace_back(a, b);
    }
    const long slot = (long)g_prof_used++;
    (void)hipEventRecord(g_prof_events[slot].first, s);
    return slot;
}
static void prof_end(long slot, hipStream_t s) {
    if (slot < 0) return;
    std::lock_guard<std::mutex> lock(g_prof_mu);
    if ((size_t)slot < g_prof_events.size()) (void)hipEventRecord(g_prof_events[slot].second, s);
}

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is per device: remember what was set for (kernel, device).
// Devices beyond the cache simply set the attribute on every call.
#define VQ_MAX_DEVICES 64
typedef std::atomic<size_t> LdsCache[VQ_MAX_DEVICES];
static int ensure_dyn_lds(const void *kern, size_t bytes, LdsCache &set) {
    int dev = 0;
    VQ_HIP(hipGetDevice(&dev));
    const bool cached = dev >= 0 && dev < VQ_MAX_DEVICES;
    if (!cached || bytes > set[dev].load(std::memory_order_acquire)) {
        VQ_HIP(hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
        if (cached) {                                   // keep the maximum (another thread may have raised it meanwhile)
            size_t cur = set[dev].load(std::memory_order_relaxed);
            while (cur < bytes && !set[dev].compare_exchange_weak(cur, bytes, std::memory_order_release)) {}
        }
    }
    return VQHIP_OK;
}

// caller-owned buffers carry their size: an undersized workspace or image is refused here instead of becoming silent
// device-memory corruption inside a kernel
#define VQ_NEED(what, have, need)                                                                         \
    do {                                                                                                  \
        const int64_t need__ = (need);                                                                    \
        if ((have) < need__) {                                                                            \
            char msg__[160];                                                                              \
            snprintf(msg__, sizeof(msg__), "%lld bytes given, %lld needed", (long long)(have), (long long)need__); \
            return fail(VQHIP_EINVAL, what, msg__);                                                       \
        }                                                                                                 \
    } while (0)

static inline int waves_grid(int64_t rows, int waves_per_block) {
    return (int)((rows + waves_per_block - 1) / waves_per_block);
}

// ---- proposal-pass dispatch ------------------------------------------------------------------------
static std::atomic<int> g_tune_slices{0};   // proposal-kernel knob for A/B measurements (vqhip_set_tuning key 2)
// key 6: 1 = the proposal kernel runs the decision stage itself (last workgroup of a token block, arrival tickets).
// Measured neutral (tools/ab_key.py 6: -1.6 % .. +1.4 % over seven shapes; the release/acquire fences cost what the
// launch saves), so the stand-alone launch stays the default; results are identical either way.
// (2 = only where ONE slice covers the codebook: the workgroup then decides its own tokens, no ticket and no fence involved)
static std::atomic<int> g_tune_fused_decide{2};
// mode 2: the proposal kernel runs the decision stage itself where one slice covers the codebook, and for small batches
// whatever the slice count (a launch less on a chain of ~5 us launches: 12 images -2.6 %, 32 images -0.5 %, tools/ab_small_batch.py;
// at 256 images and more the last-arriving workgroup's merge is the longer tail)
#define VQ_FUSED_DECIDE_MAX_N 16384
static std::atomic<int> g_tune_col_direct{1};   // key 15: 0 = a short list of codes takes the proposal pipeline like a long one (A/B; results unchanged)
static std::atomic<int> g_tune_map256{1};   // key 13: 0 = maps of 256-position images keep the 64-token tiles of gather_ste_map_kernel (A/B; results unchanged)
static std::atomic<int> g_tune_force_exact{0};   // key 12 (verification aid): the first V rows of a batch also take the whole-codebook fp32 pass
static std::atomic<int> g_tune_w32{1};      // key 11: 0 = D <= 16 keeps the 16x16x32 proposal kernel (A/B; results unchanged)
static std::atomic<int> g_tune_groups{1};   // key 9: 0 = per-element update inside the stream of the D <= 32 kernels (A/B; results unchanged)
static std::atomic<int> g_tune_noaux{1};    // key 8: 0 = cosine / dot codebooks read the (all-zero) aux chunk like L2 ones (A/B; results unchanged)
static std::atomic<int> g_tune_filter{1};   // key 5: 0 = unfiltered epilogue on the small-D instantiations too (A/B; results unchanged)
static std::atomic<int> g_tune_gather_grid{0}, g_tune_gather_nt{0};   // gather kernel knobs (keys 3, 4)
static std::atomic<int> g_tune_stream{1};   // key 18: 0 = the whole-batch fp32 pass keeps its register form (exact_tiled_kernel) (A/B; results unchanged)
static std::atomic<int> g_tune_xdirect{1};  // key 17: 0 = D = 256 batches keep the fp16 token image (x_prep / pre_kernel token side) (A/B; results unchanged)

template <int NSTEP, int TT, int WAVES, int TPS, int NBUF = 2, bool FILTER = false, bool NOAUX = false, bool GROUPS = false, int XD = 0>
static int launch_coarse_cfg(const char *ximg, int64_t N, const char *frag, int64_t nstages, int nslices, float *rec,
                             int64_t Np, const VqCbStats *cbst, const float *xh2, const float *rho2, int Dp, int metric,
                             const VqDecideOut &dec, int pad_stage, int tpb, hipStream_t s) {
    constexpr int LDS = NBUF * (TPS * NSTEP + VQ_AUX_CHUNKS(TPS)) * VQ_CHUNK_BYTES + VQ_STAGE_LDS_EXTRA;
    auto kern = coarse_kernel<NSTEP, TT, WAVES, TPS, NBUF, FILTER, NOAUX, GROUPS, XD>;
    static LdsCache lds_set;
    if (int rc = ensure_dyn_lds((const void *)kern, LDS, lds_set)) return rc;
    const int64_t ntiles = (N + 15) / 16;
    const int64_t ntb = (ntiles + tpb - 1) / tpb;
    const int grid = (int)(ntb * nslices);
    const long slot = prof_begin(s);
    kern<<<grid, WAVES * 64, LDS, s>>>(ximg, N, frag, nstages, nslices, rec, Np, cbst, xh2, rho2, Dp, metric, dec, pad_stage, tpb);
    prof_end(slot, s);
    VQ_CHECK_LAUNCH("coarse_kernel");
    return VQHIP_OK;
}

// D <= 16 (and D <= 32 without aux reads): the proposal pass on v_mfma_f32_32x32x16_f16 (vqhip_proposal32_kernels.h), same
// images; its group records are identified by identify32_kernel (launched by argmin_pipeline)
template <int TT, int WAVES, int TPS, int NBUF, bool NOAUX, int KS>
static int launch_coarse32_cfg(const char *ximg, int64_t N, const char *frag, int64_t nstages, int nslices, float *rec,
                               int64_t Np, const VqCbStats *cbst, const float *xh2, const float *rho2, int Dp, int metric,
                               const int *n_dev, const VqGroupLists &grp, int pad_stage, int tpb, hipStream_t s) {
    constexpr int LDS = NBUF * (TPS * 2 + VQ_AUX_CHUNKS(TPS)) * VQ_CHUNK_BYTES + VQ_STAGE_LDS_EXTRA;
    auto kern = coarse32_kernel<TT, WAVES, TPS, NBUF, NOAUX, KS, VQ_GROUP_TILES>;
    static LdsCache lds_set;
    if (int rc = ensure_dyn_lds((const void *)kern, LDS, lds_set)) return rc;
    const int64_t ntiles = (N + 15) / 16;
    const int64_t ntb = (ntiles + tpb - 1) / tpb;
    const long slot = prof_begin(s);
    kern<<<(int)(ntb * nslices), WAVES * 64, LDS, s>>>(ximg, N, frag, nstages, nslices, rec, Np, cbst, xh2, rho2, Dp, metric, n_dev, grp, pad_stage, tpb);
    prof_end(slot, s);
    VQ_CHECK_LAUNCH("coarse32_kernel");
    return VQHIP_OK;
}

template <int NSTEP, int TT, int WAVES, int TPS, int NBUF = 2, int XD = 0>
static int launch_rescan_cfg(const char *ximg, const char *frag, int64_t nstages, const int *rescan_list, const int *counters,
                             const float *thr, int *rescan_cnt, int *cand_list, hipStream_t s, const void *xrows = nullptr) {
    constexpr int LDS = NBUF * (TPS * NSTEP + VQ_AUX_CHUNKS(TPS)) * VQ_CHUNK_BYTES + WAVES * TT * 16 * 4 * (1 + VQ_RESCAN_LOCAL);
    auto kern = rescan_kernel<NSTEP, TT, WAVES, TPS, NBUF, XD>;
    static LdsCache lds_set;
    if (int rc = ensure_dyn_lds((const void *)kern, LDS, lds_set)) return rc;
    kern<<<256, WAVES * 64, LDS, s>>>(ximg, frag, nstages, rescan_list, counters, thr, rescan_cnt, cand_list, xrows);
    VQ_CHECK_LAUNCH("rescan_kernel");
    return VQHIP_OK;
}

#ifndef VQ_W32_MAX_D
#define VQ_W32_MAX_D 32        // the 32x32x16 proposal kernel up to this D (16: one instruction per tile; 32: two)
#endif
#ifndef VQ_NBUF_D32
#define VQ_NBUF_D32 4          // LDS ring depth of the D <= 32 proposal kernels (2 and 3 measured: profiles/r02_smallD_ring.txt)
#endif
#ifndef VQ_D32_SMALL_TT
#define VQ_D32_SMALL_TT 2      // token tiles per wave / waves per workgroup of the D <= 32 kernels below 262 144 tokens
#define VQ_D32_SMALL_W 8
#endif
#ifndef VQ_MIN_SLICES_FILTER
#define VQ_MIN_SLICES_FILTER 1
#endif
// Tiles (of 16 tokens) per workgroup of the proposal kernel when the codebook is not sliced: the workgroups of one launch
// are spread over the 256 CUs by the dispatcher, so the kernel lasts as long as the CU with the most tiles —
// ceil(workgroups / 256) x tiles per workgroup.  Full workgroups are not always the minimum of that: 100 352 tokens
// (BASELINE configs[2]) make 392 workgroups of 16 tiles, two on 136 CUs and one on the other 120 (32 tiles at worst);
// 13 tiles per workgroup make 483, two per CU at worst (26 tiles).  Ties keep the larger workgroup; at least half the
// waves stay busy (and a workgroup covers >= 128 tokens: the arrival counters of the workspace are laid out for that).
static std::atomic<int> g_tune_balance{1};  // key 10: 0 = always full workgroups (A/B; results unchanged)
static int balanced_tiles_per_block(int64_t N, int full) {
    if (!g_tune_balance.load() || full < 16) return full;
    const int64_t ntiles = (N + 15) / 16;
    int best = full;
    int64_t best_cost = ((ntiles + full - 1) / full + 255) / 256 * full;
    for (int tpb = full - 1; tpb >= full / 2 && tpb >= 8; --tpb) {
        const int64_t cost = ((ntiles + tpb - 1) / tpb + 255) / 256 * tpb;
        if (cost < best_cost) { best_cost = cost; best = tpb; }
    }
    return best;
}

// The token side made inside the proposal kernel's prologue instead of a token image (coarse_kernel<..., XD>):
// the D = 256 form with 64 tokens per wave, rows as the caller holds them, the decision stage in its own launch.  One predicate
// for the front (which then skips its token side), the proposal launch and the second pass (which reads the rows instead).
static bool vq_xdirect(int64_t N, int64_t K, int D, int x_dtype, int metric, const int *n_dev) {
    if (!g_tune_xdirect.load() || D != 256 || n_dev != nullptr || vq_cb_layout(K, D).nstages < 2) return false;   // (one slice would decide in-kernel)
    // bf16 rows only: a piece of 8 latents has the fragment's own 16 bytes and is converted in place.  fp32 rows (XD = 2: twice the
    // bytes per piece, a round trip per token tile, 58 spilled registers) cost a workgroup +44 us of prologue at 20 000 x 16384 x 256
    // against +3 us for bf16 (profiles/r06_xdirect.txt): they keep the token image
    if (x_dtype != VQHIP_DTYPE_BF16 && g_tune_xdirect.load() != 2) return false;
    if (metric != VQHIP_METRIC_L2 && metric != VQHIP_METRIC_COS && metric != VQHIP_METRIC_COS_BF16) return false;   // (not the role-swapped column pass)
    if (N <= VQ_FUSED_DECIDE_MAX_N) return false;                                     // small batches decide inside the proposal kernel
    if (g_tune_fused_decide.load() == 1) return false;
    return !(N <= 4096 || (N <= 256 * 64 && K <= 4096));                               // the 64-tokens-per-wave form (launch_coarse: !small16)
}

static int pick_slices(int64_t ntb, int64_t nstages, int min_slices = 2) {
    if (const int forced = g_tune_slices.load(); forced > 0) { int ns = forced; while (ns > 1 && ns > nstages) ns >>= 1; return ns; }
    // Enough slices to put a workgroup on every CU, no more: fewer, longer workgroups amortise their prologue and
    // record write-back, re-read the token image fewer times and write fewer records.  (Rows whose candidates cannot
    // all be identified get a second proposal pass, so the number of candidate groups does not matter for speed.)
    // ... but at least two: a lane's stream then covers half the codebook, which halves the rows whose runner-up
    // cannot be identified (second proposal pass); measured +0.7 % at 524 288 tokens, nothing lost elsewhere.
    int64_t want = (256 + ntb - 1) / ntb;
    want = want < min_slices ? min_slices : want;
    int ns = 1;
    while (ns < want && ns < VQ_MAX_SLICES) ns <<= 1;
    while (ns > 1 && ns > nstages) ns >>= 1;
    return ns;
}

// grp (nullable lists: the workspace has none beyond VQ_GROUP_MAX_TILES code tiles): where coarse32_kernel files its
// identification requests; *group_path_out = 1 when that kernel ran (argmin_pipeline then launches identify32_kernel and the
// group form of the decision stage), with *group_ks / *group_noaux / *group_pad_stage the parameters identify32_kernel needs
struct VqGroupRun { int used, ks, noaux, pad_stage; };
static int launch_coarse(const char *ximg, int64_t N, const VqCbLayout &L, const char *frag, float *rec, int64_t Np,
                         const VqCbStats *cbst, const float *xh2, const float *rho2, int metric, const VqDecideOut &dec,
                         int *nslices_out, int *fused_decide_out, hipStream_t s, VqGroupLists grp = VqGroupLists{nullptr, nullptr, nullptr, nullptr, 0, 1, 0},
                         VqGroupRun *grun = nullptr, int xd = 0) {
    const int nstep = L.nstep;
    // small batches use fewer tokens per wave so that more workgroups exist
    const bool small = N <= 256 * 64;
    // D = 256: 64 tokens per wave already from 4097 tokens on when the codebook is long (measured at K = 16 384: proposal
    // kernel 76 -> 66 us at 8192 tokens, 140 -> 122 at 16 384, 74 -> 62 at 6144; at 3072 / 4096 tokens and for short
    // codebooks — NearestAnchor's role-swapped pass has K = batch size — 32 tokens per wave stay ahead;
    // profiles/r02_tokens_per_wave_d256.txt)
    const bool small16 = N <= 4096 || (N <= 256 * 64 && L.K <= 4096);
    // D <= 32: 32 tokens per wave until 64-token workgroups would number two per CU (measured at N = 100 352, K = 8192:
    // proposal kernel 128 -> 108 us with 8 tiles per stage, one slice and 32 tokens per wave; at N = 524 288 the
    // 64-token form is the faster one)
    const bool small32 = N < 512 * 512;
    // cosine / dot product: every real code's aux value is 0 (cb_stats_kernel), only padding codes need the aux chunk
    const bool noaux = !VQ_IS_L2(metric) && g_tune_noaux.load();
    const int pad_stage = (L.K % ((int64_t)L.tps * VQ_TILE_CODES)) ? (int)(L.nstages - 1) : -1;
#define VQ_CFG(NS, TT, W, ...)                                                                      \
    {                                                                                               \
        int64_t ntb = (N + (W) * (TT) * 16 - 1) / ((W) * (TT) * 16);                                \
        int ns = pick_slices(ntb, L.nstages, (NS) <= 8 ? VQ_MIN_SLICES_FILTER : 2);                 \
        *nslices_out = ns;                                                                          \
        const int tpb = (ns == 1) ? balanced_tiles_per_block(N, (W) * (TT)) : (W) * (TT);           \
        const int fmode = g_tune_fused_decide.load();                                               \
        VqDecideOut dsel = dec;                                                                     \
        if (!(fmode == 1 || (fmode == 2 && (ns == 1 || N <= VQ_FUSED_DECIDE_MAX_N)))) dsel.idx = nullptr;                            \
        *fused_decide_out = dsel.idx != nullptr ? 1 : 0;                                            \
        return launch_coarse_cfg<NS, TT, W, __VA_ARGS__>(ximg, N, frag, L.nstages, ns, rec, Np, cbst, xh2, rho2, L.Dp, metric, dsel, pad_stage, tpb, s); \
    }
    switch (nstep) {
        // D <= 128: VALU-issue-bound with the plain epilogue -> filtered epilogue (see coarse_kernel)
        // D <= 16 always; 16 < D <= 32 (two instructions per tile) only without aux reads — measured (tools/ab_w32.py,
        // profiles/r03_w32_ab.txt): D = 32 cosine +2.5..5 %, D = 32 L2 -4..-14 % (the form is LDS-bound once the four 16-byte aux
        // reads per lane and tile come on top of 2 KiB of fragments)
        case 2: if ((L.D <= 16 || (L.D <= VQ_W32_MAX_D && noaux)) && N >= VQ_W32_MIN_N && g_tune_w32.load() && g_tune_filter.load() && g_tune_groups.load() &&
                    grp.bcnt != nullptr && grun != nullptr) {
                    // one 32x32x16 instruction covers the whole inner dimension: a quarter of the MFMA issue, group update per
                    // 16 scores.  Wide token tiles of 32 tokens: 1 (below 262 144 tokens) or 2 per wave.
#ifdef VQ_W32_TT
                    const int tt = VQ_W32_TT;
#else
                    const int tt = small32 ? 1 : 2;
#endif
                    const int full = 8 * tt * 2;                                    // 16-token tiles per workgroup
                    const int64_t ntb0 = (N + full * 16 - 1) / (full * 16);
                    int ns = pick_slices(ntb0, L.nstages, VQ_MIN_SLICES_FILTER);
                    ns = ns > VQ_GROUP_MAX_SLICES ? VQ_GROUP_MAX_SLICES : ns;
                    *nslices_out = ns;
                    // (Workgroups of up to sixteen waves, one per CU, where the codebook is not sliced — 13 waves at configs[2]: 3.25 per
                    //  SIMD through ONE ring instead of two workgroups on 192 CUs and one on 64 — were built and measured in round 6:
                    //  -2.4 % at 100 352 x 8192 x 32, -1 % at 131 072, +2 % at D = 8 / 16 and +10 % once a launch needs two rounds of them.
                    //  Not kept: the stream is bound by what the matrix pipe may draw, not by where the waves sit: profiles/r06_c3_notes.txt)
                    int tpb = (ns == 1) ? balanced_tiles_per_block(N, full) : full;
                    tpb = (tpb + 1) & ~1;                                           // whole wide tiles
                    *fused_decide_out = 0;                                          // identification first: the decision stage is its own launch
                    grun->used = 1; grun->ks = L.D <= 16 ? 1 : 2; grun->noaux = noaux ? 1 : 0; grun->pad_stage = pad_stage;
#define VQ_CFG32(TTW, NOAUXV, KSV) return launch_coarse32_cfg<TTW, 8, VQ_TPS_D32, VQ_NBUF_D32, NOAUXV, KSV>(ximg, N, frag, L.nstages, ns, rec, Np, cbst, xh2, rho2, L.Dp, metric, dec.n_dev, grp, pad_stage, tpb, s)
                    if (L.D <= 16) {
                        if (tt == 1) { if (noaux) VQ_CFG32(1, true, 1); else VQ_CFG32(1, false, 1); }
                        else { if (noaux) VQ_CFG32(2, true, 1); else VQ_CFG32(2, false, 1); }
                    } else {
                        if (tt == 1) { if (noaux) VQ_CFG32(1, true, 2); else VQ_CFG32(1, false, 2); }
                        else { if (noaux) VQ_CFG32(2, true, 2); else VQ_CFG32(2, false, 2); }
                    }
#undef VQ_CFG32
                }
                if (!g_tune_filter.load()) { if (small32) VQ_CFG(2, 2, 8, VQ_TPS_D32, VQ_NBUF_D32) else VQ_CFG(2, 4, 8, VQ_TPS_D32, VQ_NBUF_D32) }
                // (N >= 262 144: at least 1024 workgroups of 64 tokens per wave — balance no longer matters and that form is the faster one)
                // group records from VQ_GROUPS_MIN_N tokens on: below, the identification replay at the end of a
                // workgroup (a few L2 round trips) costs more than the stream saves
                if (N >= VQ_GROUPS_MIN_N && g_tune_groups.load()) {
                    if (noaux) { if (small32) VQ_CFG(2, VQ_D32_SMALL_TT, VQ_D32_SMALL_W, VQ_TPS_D32, VQ_NBUF_D32, true, true, true) else VQ_CFG(2, 4, 8, VQ_TPS_D32, VQ_NBUF_D32, true, true, true) }
                    if (small32) VQ_CFG(2, VQ_D32_SMALL_TT, VQ_D32_SMALL_W, VQ_TPS_D32, VQ_NBUF_D32, true, false, true) else VQ_CFG(2, 4, 8, VQ_TPS_D32, VQ_NBUF_D32, true, false, true)
                }
                if (noaux) { if (small32) VQ_CFG(2, VQ_D32_SMALL_TT, VQ_D32_SMALL_W, VQ_TPS_D32, VQ_NBUF_D32, true, true) else VQ_CFG(2, 4, 8, VQ_TPS_D32, VQ_NBUF_D32, true, true) }
                if (small32) VQ_CFG(2, VQ_D32_SMALL_TT, VQ_D32_SMALL_W, VQ_TPS_D32, VQ_NBUF_D32, true) else VQ_CFG(2, 4, 8, VQ_TPS_D32, VQ_NBUF_D32, true)
        case 4: if (!g_tune_filter.load()) { if (small) VQ_CFG(4, 2, 8, 4, 4) else VQ_CFG(4, 4, 8, 4, 4) }
                if (small) VQ_CFG(4, 2, 8, 4, 4, true) else VQ_CFG(4, 4, 8, 4, 4, true)
        case 8: if (!g_tune_filter.load()) { if (small) VQ_CFG(8, 2, 8, 4, 4) else VQ_CFG(8, 4, 8, 4, 4) }
                if (small) VQ_CFG(8, 2, 8, 4, 4, true) else VQ_CFG(8, 4, 8, 4, 4, true)
#if defined(VQ_D256_GROUPS)
        case 16: if (small16) VQ_CFG(16, 2, 8, VQ_TPS16, 4) else VQ_CFG(16, 4, 8, VQ_TPS16, 4, true, false, true)
#elif defined(VQ_D256_FILTER)
        case 16: if (small16) VQ_CFG(16, 2, 8, VQ_TPS16, 4, true) else VQ_CFG(16, 4, 8, VQ_TPS16, 4, true)
#else
        case 16: if (xd == 1 && !small16) VQ_CFG(16, 4, 8, VQ_TPS16, 4, false, false, false, 1)
                 if (xd == 2 && !small16) VQ_CFG(16, 4, 8, VQ_TPS16, 4, false, false, false, 2)
                 if (small16) VQ_CFG(16, 2, 8, VQ_TPS16, 4) else VQ_CFG(16, 4, 8, VQ_TPS16, 4)
#endif
        // large D: the token fragments of a wave must stay in registers for the whole stream
#ifndef VQ_CFG_D512
#define VQ_CFG_D512 VQ_CFG(32, 2, 8, 2)
#endif
#ifndef VQ_CFG_D768
#define VQ_CFG_D768 VQ_CFG(48, 2, 8, 1)
#endif
        case 32: VQ_CFG_D512
        case 48: VQ_CFG_D768
        case 64: {
            // D = 1024, two forms: eight waves x 16 tokens (two waves per SIMD; one 1 KiB LDS read per MFMA), or four waves x 48
            // tokens (one wave per SIMD with the whole register file: coarse_kernel, PIPE_H; a third of the LDS bytes per flop).
            // Per unit of work the second is 1.06-1.29x faster (1.013 against 1.095 ms at 65 536 x 8192, 2.71 against 3.22 at
            // 196 608), but its 192-token workgroups fill the chip in other multiples than the 128-token ones: 16 384 and
            // 32 768 tokens are faster on the first form, 24 576 and 49 152 on the second (profiles/r03_large_d_experiments.txt).
            // Whole rounds of 256 workgroups times the work of one, the second form's divided by 1.18, decide.
            // (At D = 768 the one-wave form with 64 tokens per wave is 1-6 % BEHIND the eight-wave form: not taken.)
            auto cost = [&](int tokens, double eff) {
                const int64_t ntb = (N + tokens - 1) / tokens;
                const int ns = pick_slices(ntb, L.nstages, 2);
                return (double)((ntb * ns + 255) / 256) * tokens * ((double)L.nstages / ns) / eff;
            };
            if (cost(192, 1.18) < cost(128, 1.0)) VQ_CFG(64, 3, 4, 1) else VQ_CFG(64, 1, 8, 1)
        }
        default: break;
    }
#undef VQ_CFG
    return fail(VQHIP_EINVAL, "vqhip_argmin: unsupported padded D");
}

// Workgroups a persistent kernel can keep resident: compute units of the current device x workgroups per CU.
static int resident_grid(int per_cu) {
    static std::atomic<int> cus[VQ_MAX_DEVICES];
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 256 * per_cu;
    if (dev >= 0 && dev < VQ_MAX_DEVICES && (n = cus[dev].load(std::memory_order_relaxed)) > 0) return n * per_cu;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    if (dev >= 0 && dev < VQ_MAX_DEVICES) cus[dev].store(n, std::memory_order_relaxed);
    return n * per_cu;
}

// The all-fp32 MFMA pass over a whole batch.  Streamed form (exact_stream_kernel, both operands from LDS, two workgroups per CU,
// contiguous spans of work items) wherever rows and codes move as whole 16-byte pieces; the register form
// (exact_tiled_kernel) for the other D and on request (vqhip_set_tuning key 18 = 0, A/B: the results are identical).
template <int MODE>
static int run_exact_tiled(const void *x, int x_dtype, const float *e, const float *en, const float *xn, int64_t N, int64_t K,
                           int D, int metric, u64 *keys, float *dout, hipStream_t s) {
    const int bf = x_dtype == VQHIP_DTYPE_F32 ? 0 : 1;
    const int64_t items = ((N + 127) / 128) * ((K + 255) / 256);
    if (g_tune_stream.load() && D % (bf ? 8 : 4) == 0 && items < (int64_t)1 << 31) {
        static LdsCache ssets[2];
        const int lds = vq_xs_lds_bytes(bf);
        const void *kern = bf ? (const void *)exact_stream_kernel<1, MODE> : (const void *)exact_stream_kernel<0, MODE>;
        if (int rc = ensure_dyn_lds(kern, lds, ssets[bf])) return rc;
        const int cap = resident_grid(2);
        const int grid = (int)(items < cap ? items : cap);
        if (bf) exact_stream_kernel<1, MODE><<<grid, 256, lds, s>>>(x, e, en, xn, N, K, D, metric, keys, dout);
        else exact_stream_kernel<0, MODE><<<grid, 256, lds, s>>>(x, e, en, xn, N, K, D, metric, keys, dout);
        VQ_CHECK_LAUNCH("exact_stream_kernel");
        return VQHIP_OK;
    }
    constexpr int LDS = 2 * 32 * 128 * 4 + 8 * 32 * 4;      // two staged tiles + the |e|^2 of an item's 256 codes
    static LdsCache sets[4];
    const void *kerns[4] = {(const void *)exact_tiled_kernel<0, MODE, false>, (const void *)exact_tiled_kernel<1, MODE, false>,
                            (const void *)exact_tiled_kernel<0, MODE, true>, (const void *)exact_tiled_kernel<1, MODE, true>};
    const int v4 = (D % 4 == 0) ? 2 : 0;
    if (int rc = ensure_dyn_lds(kerns[v4 + bf], LDS, sets[v4 + bf])) return rc;
    int grid = (int)(items < 1024 ? items : 1024);
#define VQ_TILED(DT, V4) exact_tiled_kernel<DT, MODE, V4><<<grid, 256, LDS, s>>>(x, e, en, xn, N, K, D, metric, keys, dout)
    if (v4) { if (bf) VQ_TILED(1, true); else VQ_TILED(0, true); }
    else { if (bf) VQ_TILED(1, false); else VQ_TILED(0, false); }
#undef VQ_TILED
    VQ_CHECK_LAUNCH("exact_tiled_kernel");
    return VQHIP_OK;
}

// The few-rows (VALU) form of exact_kernel stages its operands in dynamic LDS: a 64 KiB ring + 64 D bytes.  The request applies to
// the LAUNCH, i.e. also when the device-side count picks the MFMA form: at D > 512 (96 KiB) that would leave the MFMA form one
// workgroup per CU, and a part with less LDS per workgroup than asked for would refuse the launch — so the form is offered only
// where the request stays within 96 KiB and within what the device grants a workgroup.
static int exact_few_max(int D) {
    if (D > 512 || D > VQ_FEW_MAX_D) return 0;
    static std::atomic<int> lds_limit{-1};
    int lim = lds_limit.load(std::memory_order_relaxed);
    if (lim < 0) {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) != hipSuccess) v = 65536;
        lim = v;
        lds_limit.store(lim);
    }
    return vq_few_lds_bytes(D) <= lim ? VQ_EXACT_FEW_MAX : 0;
}

static int run_exact_rows(const void *x, int x_dtype, const float *e, const float *en, const float *xn, int64_t N, int64_t K, int D,
                          int metric, const int *row_list, const int *nrows_dev, u64 *keys, int *ticket, int64_t *idx,
                          int32_t *hist, hipStream_t s) {
    // last-resort path of vqhip_argmin: a few listed rows against the whole codebook (small work items); the workgroup
    // that finishes last turns the keys into indices (ticket: zeroed by x_prep_kernel with the other counters)
    const int64_t ncb = (K + 63) / 64;           // work items of the few-rows form per 16 listed rows
    const int grid = (int)(ncb < 256 ? 256 : (ncb > 1024 ? 1024 : ncb));
    if (D % 4) return fail(VQHIP_EINVAL, "exact_kernel: D % 4 != 0 (the proposal route has D % 8 == 0)");
    const int few_max = exact_few_max(D);
    const int lds = few_max ? vq_few_lds_bytes(D) : 0;
    static LdsCache sets[2];
    const int bf = x_dtype == VQHIP_DTYPE_F32 ? 0 : 1;
    if (int rc = ensure_dyn_lds(bf ? (const void *)exact_kernel<1> : (const void *)exact_kernel<0>, lds, sets[bf])) return rc;
    if (!bf)
        exact_kernel<0><<<grid, 256, lds, s>>>(x, e, en, xn, N, K, D, metric, row_list, nrows_dev, keys, ticket, idx, hist, few_max);
    else
        exact_kernel<1><<<grid, 256, lds, s>>>(x, e, en, xn, N, K, D, metric, row_list, nrows_dev, keys, ticket, idx, hist, few_max);
    VQ_CHECK_LAUNCH("exact_kernel");
    return VQHIP_OK;
}

template <int MODE>
static int run_segsum(const void *src, int x_dtype, const float *e, const int64_t *idx, const int32_t *order,
                      const int32_t *offsets, int64_t N, int64_t K, int D, const float *g_cb, float *dst, void *ws,
                      hipStream_t s) {
    float *partial = (float *)ws;
    const int64_t nranges = (N + VQ_SEG_RANGE - 1) / VQ_SEG_RANGE;
    int grid = (int)((nranges + 3) / 4); grid = grid < 1 ? 1 : (grid > 4096 ? 4096 : grid);
    if (x_dtype == VQHIP_DTYPE_F32) segsum_rows_kernel<MODE, 0><<<grid, 256, 0, s>>>(src, e, idx, order, offsets, N, (int)K, D, g_cb, dst, partial);
    else segsum_rows_kernel<MODE, 1><<<grid, 256, 0, s>>>(src, e, idx, order, offsets, N, (int)K, D, g_cb, dst, partial);
    VQ_CHECK_LAUNCH("segsum_rows_kernel");
    int fgrid = (int)((K + 3) / 4); fgrid = fgrid > 2048 ? 2048 : fgrid;
    segsum_fixup_kernel<<<fgrid, 256, 0, s>>>(offsets, (int)K, D, partial, dst);
    VQ_CHECK_LAUNCH("segsum_fixup_kernel");
    return VQHIP_OK;
}

// |v|^2 / F.normalize for rows of at most 32 elements: 64 / L rows per wave (row_small_kernel)
template <bool NORMALIZE>
static int launch_row_small(const void *v, int dtype, int64_t R, int D, float eps, float *out, hipStream_t s) {
    const int L = D <= 8 ? 8 : (D <= 16 ? 16 : 32);
    const int grid = waves_grid((R + 64 / L - 1) / (64 / L), 4);
#define VQ_ROW_SMALL(DT, LL) row_small_kernel<DT, LL, NORMALIZE><<<grid, 256, 0, s>>>(v, R, D, eps, out)
    if (dtype == VQHIP_DTYPE_F32) { if (L == 8) VQ_ROW_SMALL(0, 8); else if (L == 16) VQ_ROW_SMALL(0, 16); else VQ_ROW_SMALL(0, 32); }
    else { if (L == 8) VQ_ROW_SMALL(1, 8); else if (L == 16) VQ_ROW_SMALL(1, 16); else VQ_ROW_SMALL(1, 32); }
#undef VQ_ROW_SMALL
    VQ_CHECK_LAUNCH("row_small_kernel");
    return VQHIP_OK;
}

extern "C" {

int vqhip_version(void) { return VQHIP_VERSION; }
const char *vqhip_last_error(void) { return g_err; }

int64_t vqhip_codebook_exact_offset(int64_t K, int D) {
    if (K <= 0 || D <= 0) return -1;
    return vq_cb_layout(K, D).off_eexact;
}

// front of the VQ-KD / NormalizeCallback forwards (vqhip_step_kernels.h): the L-lanes-per-row form at D <= 32
static int launch_vqkd_front(const float *w_in, float *w_mid, int64_t K, const void *x, int x_dtype, float *xn, int64_t N, int D, float eps,
                             float *zero, int64_t nzero, int w_passes, hipStream_t s) {
    if (D <= 32) {
        const int L = D <= 8 ? 8 : (D <= 16 ? 16 : 32), rpb = 4 * (64 / L);
        const int kblocks = (int)((K + rpb - 1) / rpb), xblocks = (int)((N + rpb - 1) / rpb);
#define VQ_FRONT_SMALL(DT, LL) vqkd_front_small_kernel<DT, LL><<<kblocks + xblocks, 256, 0, s>>>(w_in, w_mid, K, x, xn, N, D, eps, kblocks, zero, nzero, w_passes)
        if (x_dtype == VQHIP_DTYPE_F32) { if (L == 8) VQ_FRONT_SMALL(0, 8); else if (L == 16) VQ_FRONT_SMALL(0, 16); else VQ_FRONT_SMALL(0, 32); }
        else { if (L == 8) VQ_FRONT_SMALL(1, 8); else if (L == 16) VQ_FRONT_SMALL(1, 16); else VQ_FRONT_SMALL(1, 32); }
#undef VQ_FRONT_SMALL
    } else {
        const int kblocks = (int)((K + 3) / 4), xblocks = (int)((N + 3) / 4);
        if (x_dtype == VQHIP_DTYPE_F32) vqkd_front_kernel<0><<<kblocks + xblocks, 256, 0, s>>>(w_in, w_mid, K, x, xn, N, D, eps, kblocks, zero, nzero, w_passes);
        else vqkd_front_kernel<1><<<kblocks + xblocks, 256, 0, s>>>(w_in, w_mid, K, x, xn, N, D, eps, kblocks, zero, nzero, w_passes);
    }
    VQ_CHECK_LAUNCH("vqkd_front_kernel");
    return VQHIP_OK;
}

int64_t vqhip_codebook_bytes(int64_t K, int D) {
    if (K <= 0 || D <= 0) return 0;
    return vq_cb_layout(K, D).total;
}

int64_t vqhip_workspace_bytes(int64_t N, int64_t K, int D) {
    if (N < 0 || K <= 0 || D <= 0) return 0;
    return vq_ws_layout(N > 0 ? N : 1, K, D).total;
}

int vqhip_row_sqnorm(const void *v, int dtype, int64_t R, int D, float *out, void *stream) {
    if (!v || !out || D <= 0 || R < 0) return fail(VQHIP_EINVAL, "vqhip_row_sqnorm: bad argument");
    if (R == 0) return VQHIP_OK;
    hipStream_t s = (hipStream_t)stream;
    if (D <= 32 && (dtype == VQHIP_DTYPE_F32 || dtype == VQHIP_DTYPE_BF16)) return launch_row_small<false>(v, dtype, R, D, 0.0f, out, s);
    if (dtype == VQHIP_DTYPE_F32) row_sqnorm_kernel<0><<<waves_grid(R, 4), 256, 0, s>>>(v, R, D, out);
    else if (dtype == VQHIP_DTYPE_BF16) row_sqnorm_kernel<1><<<waves_grid(R, 4), 256, 0, s>>>(v, R, D, out);
    else return fail(VQHIP_EINVAL, "vqhip_row_sqnorm: dtype");
    VQ_CHECK_LAUNCH("row_sqnorm_kernel");
    return VQHIP_OK;
}

int vqhip_normalize_rows(const void *v, int dtype, int64_t R, int D, float eps, float *out, void *stream) {
    if (!v || !out || D <= 0 || R < 0) return fail(VQHIP_EINVAL, "vqhip_normalize_rows: bad argument");
    if (R == 0) return VQHIP_OK;
    hipStream_t s = (hipStream_t)stream;
    if (D <= 32 && (dtype == VQHIP_DTYPE_F32 || dtype == VQHIP_DTYPE_BF16)) return launch_row_small<true>(v, dtype, R, D, eps, out, s);
    if (dtype == VQHIP_DTYPE_F32) normalize_rows_kernel<0><<<waves_grid(R, 4), 256, 0, s>>>(v, R, D, eps, out);
    else if (dtype == VQHIP_DTYPE_BF16) normalize_rows_kernel<1><<<waves_grid(R, 4), 256, 0, s>>>(v, R, D, eps, out);
    else return fail(VQHIP_EINVAL, "vqhip_normalize_rows: dtype");
    VQ_CHECK_LAUNCH("normalize_rows_kernel");
    return VQHIP_OK;
}

// metric may carry the internal words (VQ_METRIC_DOT, VQ_METRIC_SWAP)
static int codebook_prepare_impl(const float *e, int64_t K, int D, int metric, void *cb, void *stream) {
    if (K >= (1ll << 31)) return fail(VQHIP_EINVAL, "vqhip_codebook_prepare: K too large");
    hipStream_t s = (hipStream_t)stream;
    VqCbLayout L = vq_cb_layout(K, D);
    char *c = (char *)cb;
    if (VQ_IS_COS(metric) && vq_coarse_supported(D)) {      // cosine: statistics and image in ONE launch (cb_cos_body)
        cb_cos_kernel<<<(int)(L.nstages * L.tps), 256, 0, s>>>(e, K, D, metric, c, L);
        VQ_CHECK_LAUNCH("cb_cos_kernel");
        return VQHIP_OK;
    }
    cb_stats_kernel<<<(int)((K + 15) / 16), 256, 0, s>>>(e, K, D, metric, c, L);
    VQ_CHECK_LAUNCH("cb_stats_kernel");
    if (vq_coarse_supported(D)) {
        cb_image_kernel<<<(int)(L.nstages * L.tps), 256, 0, s>>>(e, K, D, metric, c, L);
        VQ_CHECK_LAUNCH("cb_image_kernel");
    }
    return VQHIP_OK;
}

int vqhip_codebook_prepare(const float *e, int64_t K, int D, int metric, void *cb, int64_t cb_bytes, void *stream) {
    if (!e || !cb || K <= 0 || D <= 0 || (metric != VQHIP_METRIC_L2 && metric != VQHIP_METRIC_COS && metric != VQHIP_METRIC_COS_BF16))
        return fail(VQHIP_EINVAL, "vqhip_codebook_prepare: bad argument");
    VQ_NEED("vqhip_codebook_prepare: cb too small", cb_bytes, vqhip_codebook_bytes(K, D));
    return codebook_prepare_impl(e, K, D, metric, cb, stream);
}

int vqhip_argmin_exact(const void *x, int x_dtype, const float *e, int64_t N, int64_t K, int D, int metric, int64_t *idx,
                       float *dmin, int32_t *hist, void *ws, int64_t ws_bytes, void *stream);

// The proposal + decision pipeline: N rows `x` against the K codes whose prepared image is `cb` and whose fp32 rows
// (as used by the exact definition) are `e_exact`.  `metric` may carry the internal words (DOT, SWAP).
// x_prepared: the token side (x_prep) was already produced into `ws` by pre_kernel (encode_fused_front)
// n_dev (nullable DEVICE int): only rows [0, min(N, *n_dev)) are live; the launches are sized for N
static int argmin_pipeline(const void *x, int x_dtype, const float *e_exact, const void *cb, int64_t N, int64_t K, int D,
                           int metric, int64_t *idx, int32_t *hist, void *ws, void *stream, bool x_prepared = false,
                           const int *n_dev = nullptr) {
    hipStream_t s = (hipStream_t)stream;
    VqCbLayout L = vq_cb_layout(K, D);
    VqWsLayout W = vq_ws_layout(N, K, D);
    const char *c = (const char *)cb;
    char *w = (char *)ws;
    const int64_t Np = (N + 63) / 64 * 64;
    int *counters = (int *)(w + W.off_counters);
    float *xh2 = (float *)(w + W.off_xh2), *rho2 = (float *)(w + W.off_rho2), *rec = (float *)(w + W.off_rec);
    int *flag_list = (int *)(w + W.off_flag);
    u64 *keys = (u64 *)(w + W.off_keys);
    const float *en = (const float *)(c + L.off_en);

    int nslices = 1, rc;
    char *ximg = w + W.off_ximg;
    int *rescan_list = flag_list;
    int *multi_list = (int *)(w + W.off_multi), *exact_list = (int *)(w + W.off_exact);
    float *thr = (float *)(w + W.off_thr);
    int *rescan_cnt = (int *)(w + W.off_rcnt), *cand_list = (int *)(w + W.off_rlist);
    int *arrive = (int *)(w + W.off_arrive);
    const int narrive = (int)W.narrive;          // arrival counters + the group path's bucket counters (one zeroed range)
    int xgrid = (int)((N + 31) / 32);
    // no token image: the proposal kernel converts the rows it loads (coarse_kernel<..., XD>); the front only does the housekeeping
    const int xd = vq_xdirect(N, K, D, x_dtype, metric, n_dev) ? (x_dtype == VQHIP_DTYPE_BF16 ? 1 : 2) : 0;
    if (!x_prepared) {
        if (xd) xgrid = xgrid < 64 ? xgrid : 64;
        if (x_dtype == VQHIP_DTYPE_F32) x_prep_kernel<0><<<xgrid, 256, 0, s>>>(x, N, D, L.nstep, ximg, xh2, rho2, (float *)(w + W.off_xn), counters, (char *)cb, L, arrive, narrive, xd);
        else x_prep_kernel<1><<<xgrid, 256, 0, s>>>(x, N, D, L.nstep, ximg, xh2, rho2, (float *)(w + W.off_xn), counters, (char *)cb, L, arrive, narrive, xd);
        VQ_CHECK_LAUNCH("x_prep_kernel");
    }
    // the proposal kernel also runs the decision stage (the workgroup that completes a token block merges its slices)
    VqDecideOut dec{idx, hist, rescan_list, multi_list, exact_list, counters, keys, thr, rescan_cnt, arrive, n_dev,
                    x, xh2, rho2, (float *)(w + W.off_xn)};
    VqDecideOut dec_arg = dec;                   // launch_coarse decides (knob 6, slice count) whether the proposal kernel runs the
    int fused_done = 0;                          // decision stage itself and reports it here
    // D <= 32 group path: request lists of the proposal kernel (cap = an equal share of the pool per code tile, whole batches of 32)
    VqGroupLists grp{nullptr, nullptr, nullptr, nullptr, 0, 1, 0};
    VqGroupRun grun{0, 0, 0, -1};
    const int ntiles_cb = (int)(L.nstages * L.tps);
    int nbuckets = 0;
    if (W.nbkt > 0 && ntiles_cb <= VQ_GROUP_MAX_TILES && N < (1ll << 30)) {
        grp.R = vq_group_replicas(ntiles_cb / VQ_GROUP_TILES);
        nbuckets = ntiles_cb / VQ_GROUP_TILES * grp.R;
        grp.bcnt = (int *)(w + W.off_bcnt);
        grp.blist = (uint32_t *)(w + W.off_blist);
        grp.bfrag = w + W.off_bfrag;
        grp.rece2 = (float *)(w + W.off_rece2);
        grp.cap = (int)(W.blist_entries / nbuckets / 32 * 32);
        uint32_t bits = 1;                                   // group ids ride in the low mantissa bits of the running group maxima
        while ((1u << bits) < (uint32_t)(ntiles_cb / VQ_GROUP_TILES)) ++bits;
        grp.idmask = (1u << bits) - 1u;
    }
    rc = launch_coarse(ximg, N, L, c + L.off_frag, rec, Np, (const VqCbStats *)(c + L.off_stats), xh2, rho2, metric, dec_arg, &nslices, &fused_done, s, grp, &grun, xd);
    if (rc) return rc;
    const float *rece2 = nullptr;
    if (grun.used) {             // identify the group records: one candidate per request, written into the records
        const int igrid = (nbuckets + 3) / 4;                   // one wave per bucket
        if (grun.ks == 1) identify32_kernel<1, VQ_GROUP_TILES><<<igrid, 256, 0, s>>>(c + L.off_frag, L.nstages, nslices, nbuckets, rec, Np, (const VqCbStats *)(c + L.off_stats), grp, grun.pad_stage, grun.noaux);
        else identify32_kernel<2, VQ_GROUP_TILES><<<igrid, 256, 0, s>>>(c + L.off_frag, L.nstages, nslices, nbuckets, rec, Np, (const VqCbStats *)(c + L.off_stats), grp, grun.pad_stage, grun.noaux);
        VQ_CHECK_LAUNCH("identify32_kernel");
        rece2 = grp.rece2;
    }
    if (!fused_done) {
        // (1024-thread workgroups: 256 and 512 measured 1-2 % slower per encode at configs[2] and the tokenizer shape, level at D = 256)
        refine_decide_kernel<<<(int)((N + 1023) / 1024), 1024, 0, s>>>(c, L, N, metric, nslices, rec, xh2, rho2, Np, dec, rece2);
        VQ_CHECK_LAUNCH("refine_decide_kernel");
    }
    // second-chance proposals for rows with a possibly unidentified candidate (the kernel gathers their fragments from
    // the token image itself) ...
    float *xnorm = (float *)(w + W.off_xn);
    {
        const char *frag = c + L.off_frag;
        int rrc = VQHIP_OK;
        if (xd == 1) rrc = launch_rescan_cfg<16, 2, 8, VQ_TPS16, 4, 1>(ximg, frag, L.nstages, rescan_list, counters, thr, rescan_cnt, cand_list, s, x);
        else if (xd == 2) rrc = launch_rescan_cfg<16, 2, 8, VQ_TPS16, 4, 2>(ximg, frag, L.nstages, rescan_list, counters, thr, rescan_cnt, cand_list, s, x);
        else switch (L.nstep) {
#define VQ_RESCAN(NS, TT, ...) case NS: rrc = launch_rescan_cfg<NS, TT, 8, __VA_ARGS__>(ximg, frag, L.nstages, rescan_list, counters, thr, rescan_cnt, cand_list, s); break;
            VQ_RESCAN(2, 2, VQ_TPS_D32, 4) VQ_RESCAN(4, 2, 4, 4) VQ_RESCAN(8, 2, 4, 4) VQ_RESCAN(16, 2, VQ_TPS16, 4) VQ_RESCAN(32, 2, 2) VQ_RESCAN(48, 2, 1) VQ_RESCAN(64, 1, 1)
#undef VQ_RESCAN
            default: return fail(VQHIP_EINVAL, "vqhip_argmin: unsupported padded D");
        }
        if (rrc) return rrc;
        VQ_CHECK_LAUNCH("rescan_kernel");
    }
    // ... then ONE launch re-ranks exactly both the rows with several identified candidates (one lane per (row, record
    // slot)) and the rescanned rows' candidate lists
    {
        int S0 = 4;                                           // slot lanes per multi row: power of two >= max(4, 2*nslices)
        while (S0 < 2 * nslices) S0 <<= 1;
        // the re-rank is a latency chain per (row, candidate) pair: it wants one row per wave however short the queues
        // are (64 + 64 workgroups made it 80 us instead of 14 at N = 3072, cosine, 16 slices); idle workgroups exit at once
        const int64_t g0 = 2048, g1 = 1024;
        if (x_dtype == VQHIP_DTYPE_F32)
            refine_rerank_kernel<0><<<(int)(g0 + g1), 256, 0, s>>>(x, e_exact, c, L, D, metric, nslices, S0, (int)g0, rec, xh2, rho2,
                                                                  xnorm, Np, idx, hist, multi_list, rescan_list, counters,
                                                                  rescan_cnt, cand_list, exact_list, keys, grun.used);
        else
            refine_rerank_kernel<1><<<(int)(g0 + g1), 256, 0, s>>>(x, e_exact, c, L, D, metric, nslices, S0, (int)g0, rec, xh2, rho2,
                                                                  xnorm, Np, idx, hist, multi_list, rescan_list, counters,
                                                                  rescan_cnt, cand_list, exact_list, keys, grun.used);
        VQ_CHECK_LAUNCH("refine_rerank_kernel");
    }
    // last resort: whole-codebook fp32 pass (non-finite data, overflowing candidate lists)
    if (const int force = g_tune_force_exact.load()) {
        force_exact_rows_kernel<<<1, 1024, 0, s>>>((int)(force < N ? force : N), exact_list, counters, keys);
        VQ_CHECK_LAUNCH("force_exact_rows_kernel");
    }
    return run_exact_rows(x, x_dtype, e_exact, en, xnorm, N, K, D, metric, exact_list, counters + 2, keys, counters + 3, idx, hist, s);
}

// Front of an encode whose codebook image is made in the same call: ONE launch for the codebook statistics and the whole
// token side (they are independent), then the image kernel.  `rows` are the N rows to quantize (normalised into `xq`
// first when xnorm), `codes` the Kc rows the image is made from, `ws` the workspace of argmin_pipeline(N rows, Kc codes).
// hw > 0: `rows` is the feature map [N / hw, D, hw] (NCHW); the token-major rows go to `xrows` (input dtype; cosine: xq)
// grows != nullptr (vqhip_col_argmin_rows, fp32 `rows` = the codebook, no normalisation): row t of the call is rows[grows[t]] for
// t < *gcount and zeros up to N; the gathered rows are written to `xrows`
static int encode_fused_front(const void *rows, int rows_dtype, int64_t N, const float *codes, int64_t Kc, int D, int cb_metric,
                              void *cb, void *ws, bool xnorm, float *xq, hipStream_t s, int32_t *hist_zero = nullptr,
                              int64_t hw = 0, void *xrows = nullptr, const int32_t *grows = nullptr, const int32_t *gcount = nullptr) {
    VqCbLayout L = vq_cb_layout(Kc, D);
    VqWsLayout W = vq_ws_layout(N, Kc, D);
    char *w = (char *)ws, *c = (char *)cb;
    // cosine: the whole codebook preparation rides in the same launch (normalised rows need no statistics pass for their
    // scale).  DOT — the role-swapped NearestAnchor pass under the cosine metric, whose operands the caller has normalised
    // (include/vqhip.h: "COS: x and e already normalised") — takes the same form on the rows as given: constant scale 2^13,
    // the tile maxima from the actual data (rows that are not unit-norm keep exact results: values beyond fp16 range at that
    // scale raise the non-finite flag and the rows take the fp32 pass)
    const bool cosimg = VQ_IS_COS(cb_metric) || (cb_metric & 3) == VQ_METRIC_DOT;
    const int nblk_stats = cosimg ? (int)(L.nstages * L.tps) : (int)((Kc + 15) / 16);
    // the rows the pipeline will be handed are the rows given here (no normalisation, token-major, no gather): where the proposal
    // kernel makes its own fragments (vq_xdirect: the same predicate argmin_pipeline evaluates) the token side is housekeeping only
    const int toff = (!xnorm && hw == 0 && grows == nullptr && vq_xdirect(N, Kc, D, rows_dtype, cb_metric, nullptr)) ? 1 : 0;
    int xgrid = (int)((N + 31) / 32);
    if (toff) xgrid = xgrid < 64 ? xgrid : 64;
    const int narrive = (int)W.narrive;
    int *counters = (int *)(w + W.off_counters), *arrive = (int *)(w + W.off_arrive);
    float *xh2 = (float *)(w + W.off_xh2), *rho2 = (float *)(w + W.off_rho2), *xn = (float *)(w + W.off_xn);
    char *ximg = w + W.off_ximg;
#define VQ_PRE(DT, XN, MAP, COSI) pre_kernel<DT, XN, MAP, COSI><<<nblk_stats + xgrid, 256, 0, s>>>(codes, Kc, cb_metric, c, L, nblk_stats, rows, N, D, L.nstep, ximg, xh2, rho2, xn, counters, arrive, narrive, xq, 1e-12f, hist_zero, hw, xrows, nullptr, nullptr, toff)
#define VQ_PRE2(DT, XN, MAP) do { if (cosimg) VQ_PRE(DT, XN, MAP, true); else VQ_PRE(DT, XN, MAP, false); } while (0)
    if (grows != nullptr) {
        if (rows_dtype != VQHIP_DTYPE_F32 || xnorm || hw > 0 || L.nstep == 2) return fail(VQHIP_EINVAL, "encode_fused_front: gather form");
        if (cosimg) pre_kernel<0, false, false, true, true><<<nblk_stats + xgrid, 256, 0, s>>>(codes, Kc, cb_metric, c, L, nblk_stats, rows, N, D, L.nstep, ximg, xh2, rho2, xn, counters, arrive, narrive, xq, 1e-12f, hist_zero, hw, xrows, grows, gcount);
        else pre_kernel<0, false, false, false, true><<<nblk_stats + xgrid, 256, 0, s>>>(codes, Kc, cb_metric, c, L, nblk_stats, rows, N, D, L.nstep, ximg, xh2, rho2, xn, counters, arrive, narrive, xq, 1e-12f, hist_zero, hw, xrows, grows, gcount);
    } else
    if (hw > 0) {
        if (rows_dtype == VQHIP_DTYPE_F32) { if (xnorm) VQ_PRE2(0, true, true); else VQ_PRE2(0, false, true); }
        else { if (xnorm) VQ_PRE2(1, true, true); else VQ_PRE2(1, false, true); }
    } else {
        if (rows_dtype == VQHIP_DTYPE_F32) { if (xnorm) VQ_PRE2(0, true, false); else VQ_PRE2(0, false, false); }
        else { if (xnorm) VQ_PRE2(1, true, false); else VQ_PRE2(1, false, false); }
    }
#undef VQ_PRE2
#undef VQ_PRE
    VQ_CHECK_LAUNCH("pre_kernel");
    if (cosimg) return VQHIP_OK;
    cb_image_kernel<<<(int)(L.nstages * L.tps), 256, 0, s>>>(codes, Kc, D, cb_metric, c, L);
    VQ_CHECK_LAUNCH("cb_image_kernel");
    return VQHIP_OK;
}

int vqhip_encode(const void *x, int x_dtype, const float *e, int64_t N, int64_t K, int D, int metric, void *cb, int64_t cb_bytes,
                 int64_t *idx, int32_t *hist, float *xq, void *ws, int64_t ws_bytes, void *stream) {
    return vqhip_encode_ex(x, x_dtype, e, N, K, D, metric, cb, cb_bytes, idx, hist, xq, ws, ws_bytes, 0, stream);
}

int vqhip_encode_ex(const void *x, int x_dtype, const float *e, int64_t N, int64_t K, int D, int metric, void *cb, int64_t cb_bytes,
                    int64_t *idx, int32_t *hist, float *xq, void *ws, int64_t ws_bytes, int flags, void *stream) {
    const bool zero_hist = hist != nullptr && (flags & VQHIP_ENCODE_ZERO_HIST) != 0;
    if (N == 0 || !vq_coarse_supported(D)) {                 // paths without the fused front: plain memset
        if (zero_hist && K > 0) VQ_HIP(hipMemsetAsync(hist, 0, (size_t)K * 4, (hipStream_t)stream));
    }
    if (N == 0) return e && cb && K > 0 && D > 0 ? vqhip_codebook_prepare(e, K, D, metric, cb, cb_bytes, stream) : fail(VQHIP_EINVAL, "vqhip_encode: bad argument");
    if (!x || !e || !cb || !idx || !ws || N < 0 || K <= 0 || D <= 0) return fail(VQHIP_EINVAL, "vqhip_encode: bad argument");
    if (metric != VQHIP_METRIC_L2 && metric != VQHIP_METRIC_COS && metric != VQHIP_METRIC_COS_BF16) return fail(VQHIP_EINVAL, "vqhip_encode: metric");
    if (x_dtype != VQHIP_DTYPE_F32 && x_dtype != VQHIP_DTYPE_BF16) return fail(VQHIP_EINVAL, "vqhip_encode: x_dtype");
    if (VQ_IS_COS(metric) && !xq) return fail(VQHIP_EINVAL, "vqhip_encode: the cosine metric needs the xq buffer");
    if (N >= (1ll << 31) || K >= (1ll << 31)) return fail(VQHIP_EINVAL, "vqhip_encode: N or K too large");
    VQ_NEED("vqhip_encode: cb too small", cb_bytes, vqhip_codebook_bytes(K, D));
    VQ_NEED("vqhip_encode: ws too small", ws_bytes, vqhip_workspace_bytes(N, K, D));
    if (!vq_coarse_supported(D)) {          // no fp16 proposal image for this D: the separate entry points do the work
        if (int rc = vqhip_codebook_prepare(e, K, D, metric, cb, cb_bytes, stream)) return rc;
        const void *rows = x; int rows_dtype = x_dtype;
        if (VQ_IS_COS(metric)) {
            if (VQ_IS_BF16(metric)) return fail(VQHIP_EINVAL, "vqhip_encode: the bf16-autocast cosine metric needs D <= 1024, D % 8 == 0");
            if (int rc = vqhip_normalize_rows(x, x_dtype, N, D, 1e-12f, xq, stream)) return rc;
            rows = xq; rows_dtype = VQHIP_DTYPE_F32;
        }
        return vqhip_argmin(rows, rows_dtype, e, cb, cb_bytes, N, K, D, metric, idx, hist, ws, ws_bytes, stream);
    }
    hipStream_t s = (hipStream_t)stream;
    const bool cos = VQ_IS_COS(metric);
    if (int rc = encode_fused_front(x, x_dtype, N, e, K, D, metric, cb, ws, cos, xq, s, zero_hist ? hist : nullptr)) return rc;
    VqCbLayout L = vq_cb_layout(K, D);
    const float *e_exact = cos ? (const float *)((const char *)cb + L.off_eexact) : e;
    // from here on the rows are what vqhip_argmin would have been given: the normalised fp32 rows for cosine
    return argmin_pipeline(cos ? (const void *)xq : x, cos ? VQHIP_DTYPE_F32 : x_dtype, e_exact, cb, N, K, D, metric, idx, hist, ws,
                           stream, /*x_prepared=*/true);
}

int vqhip_encode_map(const void *x_map, int x_dtype, const float *e, int64_t B, int64_t HW, int64_t K, int D, int metric, void *cb,
                     int64_t cb_bytes, int64_t *idx, int32_t *hist, void *xrows, float *xq, void *ws, int64_t ws_bytes, int flags,
                     void *stream) {
    const int64_t N = B * HW;
    if (!x_map || !e || !cb || !idx || !xrows || !ws || B <= 0 || HW <= 0 || K <= 0 || D <= 0) return fail(VQHIP_EINVAL, "vqhip_encode_map: bad argument");
    if (VQ_IS_COS(metric) && !xq) return fail(VQHIP_EINVAL, "vqhip_encode_map: the cosine metric needs the xq buffer");
    if (metric != VQHIP_METRIC_L2 && metric != VQHIP_METRIC_COS && metric != VQHIP_METRIC_COS_BF16) return fail(VQHIP_EINVAL, "vqhip_encode_map: metric");
    if (x_dtype != VQHIP_DTYPE_F32 && x_dtype != VQHIP_DTYPE_BF16) return fail(VQHIP_EINVAL, "vqhip_encode_map: x_dtype");
    if (N >= (1ll << 31) || K >= (1ll << 31)) return fail(VQHIP_EINVAL, "vqhip_encode_map: N or K too large");
    if (!vq_coarse_supported(D)) return fail(VQHIP_EINVAL, "vqhip_encode_map: needs D <= 1024, D % 8 == 0 (transpose and use vqhip_encode)");
    VQ_NEED("vqhip_encode_map: cb too small", cb_bytes, vqhip_codebook_bytes(K, D));
    VQ_NEED("vqhip_encode_map: ws too small", ws_bytes, vqhip_workspace_bytes(N, K, D));
    hipStream_t s = (hipStream_t)stream;
    const bool cos = VQ_IS_COS(metric);
    const bool zero_hist = hist != nullptr && (flags & VQHIP_ENCODE_ZERO_HIST) != 0;
    if (int rc = encode_fused_front(x_map, x_dtype, N, e, K, D, metric, cb, ws, cos, cos ? xq : nullptr, s,
                                    zero_hist ? hist : nullptr, HW, xrows)) return rc;
    VqCbLayout L = vq_cb_layout(K, D);
    const float *e_exact = cos ? (const float *)((const char *)cb + L.off_eexact) : e;
    // from here on the rows are token-major: what the front wrote (L2: the copy in the input dtype; cosine: the normalised fp32 rows)
    return argmin_pipeline(cos ? (const void *)xq : (const void *)xrows, cos ? VQHIP_DTYPE_F32 : x_dtype, e_exact, cb, N, K, D, metric,
                           idx, hist, ws, stream, /*x_prepared=*/true);
}

int vqhip_argmin(const void *x, int x_dtype, const float *e, const void *cb, int64_t cb_bytes, int64_t N, int64_t K, int D, int metric,
                 int64_t *idx, int32_t *hist, void *ws, int64_t ws_bytes, void *stream) {
    if (N == 0) return VQHIP_OK;
    if (!x || !cb || !idx || !ws || N < 0 || K <= 0 || D <= 0) return fail(VQHIP_EINVAL, "vqhip_argmin: bad argument");
    if (metric != VQHIP_METRIC_L2 && metric != VQHIP_METRIC_COS && metric != VQHIP_METRIC_COS_BF16) return fail(VQHIP_EINVAL, "vqhip_argmin: metric");
    if (x_dtype != VQHIP_DTYPE_F32 && x_dtype != VQHIP_DTYPE_BF16) return fail(VQHIP_EINVAL, "vqhip_argmin: x_dtype");
    if (metric == VQHIP_METRIC_L2 && !e) return fail(VQHIP_EINVAL, "vqhip_argmin: e is required for L2");
    if (N >= (1ll << 31) || K >= (1ll << 31)) return fail(VQHIP_EINVAL, "vqhip_argmin: N or K too large");
    VQ_NEED("vqhip_argmin: cb too small", cb_bytes, vqhip_codebook_bytes(K, D));
    VQ_NEED("vqhip_argmin: ws too small", ws_bytes, vqhip_workspace_bytes(N, K, D));
    VqCbLayout L = vq_cb_layout(K, D);
    const float *e_exact = VQ_IS_COS(metric) ? (const float *)((const char *)cb + L.off_eexact) : e;
    if (!vq_coarse_supported(D)) {
        // no fp16 proposal image for this D: whole-codebook fp32 pass for every row
        VQ_HIP(hipMemsetAsync(ws, 0, 256, (hipStream_t)stream));
        return vqhip_argmin_exact(x, x_dtype, e_exact, N, K, D, metric, idx, nullptr, hist, ws, ws_bytes, stream);
    }
    return argmin_pipeline(x, x_dtype, e_exact, cb, N, K, D, metric, idx, hist, ws, stream);
}

int vqhip_argmin_exact(const void *x, int x_dtype, const float *e, int64_t N, int64_t K, int D, int metric, int64_t *idx,
                       float *dmin, int32_t *hist, void *ws, int64_t ws_bytes, void *stream) {
    if (N == 0) return VQHIP_OK;
    if (!x || !e || !idx || !ws || N < 0 || K <= 0 || D <= 0) return fail(VQHIP_EINVAL, "vqhip_argmin_exact: bad argument");
    if (x_dtype != VQHIP_DTYPE_F32 && x_dtype != VQHIP_DTYPE_BF16) return fail(VQHIP_EINVAL, "vqhip_argmin_exact: x_dtype");
    if (N >= (1ll << 31) || K >= (1ll << 31)) return fail(VQHIP_EINVAL, "vqhip_argmin_exact: N or K too large");
    VQ_NEED("vqhip_argmin_exact: ws too small", ws_bytes, vqhip_workspace_bytes(N, K, D));
    hipStream_t s = (hipStream_t)stream;
    VqWsLayout W = vq_ws_layout(N, K, D);
    char *w = (char *)ws;
    u64 *keys = (u64 *)(w + W.off_keys);
    float *en = (float *)(w + W.off_en);
    if (metric == VQHIP_METRIC_L2) {
        int rc = vqhip_row_sqnorm(e, VQHIP_DTYPE_F32, K, D, en, stream);
        if (rc) return rc;
    }
    fill_u64_kernel<<<256, 256, 0, s>>>(keys, N, ~0ull);
    VQ_CHECK_LAUNCH("fill_u64_kernel");
    float *xn = (float *)(w + W.off_xh2);
    if (metric == VQHIP_METRIC_L2) {
        int rc0 = vqhip_row_sqnorm(x, x_dtype, N, D, xn, stream);
        if (rc0) return rc0;
    }
    int rc = run_exact_tiled<0>(x, x_dtype, e, en, xn, N, K, D, metric, keys, nullptr, s);
    if (rc) return rc;
    finalize_kernel<<<256, 256, 0, s>>>(keys, nullptr, nullptr, N, idx, dmin, hist);
    VQ_CHECK_LAUNCH("finalize_kernel");
    return VQHIP_OK;
}

int64_t vqhip_col_workspace_bytes(int64_t N, int64_t K, int D) {
    if (N <= 0 || K <= 0 || D <= 0) return 0;
    // [pipeline workspace for K rows against N codes][image of the latents as codes][fp32 copy of bf16 latents]
    int64_t a = (vq_ws_layout(K, N, D).total + 1023) / 1024 * 1024;
    int64_t b = (vq_cb_layout(N, D).total + 1023) / 1024 * 1024;
    int64_t legacy = vq_ws_layout(N, K, D).total;        // fp32-only route (D without a proposal image)
    int64_t t = a + b + N * (int64_t)D * 4;
    return t > legacy ? t : legacy;
}

int vqhip_col_argmin(const void *x, int x_dtype, const float *e, int64_t N, int64_t K, int D, int metric,
                     int64_t *col_idx, void *ws, int64_t ws_bytes, void *stream) {
    if (!x || !e || !col_idx || !ws || N <= 0 || K <= 0 || D <= 0) return fail(VQHIP_EINVAL, "vqhip_col_argmin: bad argument");
    if (N >= (1ll << 31) || K >= (1ll << 31)) return fail(VQHIP_EINVAL, "vqhip_col_argmin: N or K too large");
    if (x_dtype != VQHIP_DTYPE_F32 && x_dtype != VQHIP_DTYPE_BF16) return fail(VQHIP_EINVAL, "vqhip_col_argmin: x_dtype");
    if (metric != VQHIP_METRIC_L2 && metric != VQHIP_METRIC_COS && metric != VQHIP_METRIC_COS_BF16) return fail(VQHIP_EINVAL, "vqhip_col_argmin: metric");
    VQ_NEED("vqhip_col_argmin: ws too small", ws_bytes, vqhip_col_workspace_bytes(N, K, D));
    hipStream_t s = (hipStream_t)stream;
    char *w = (char *)ws;
    if (vq_coarse_supported(D)) {
        // Roles swapped: the K codebook rows are the "rows", the N latents are the "codes"; the same proposal + exact
        // re-rank pipeline then returns for every code its nearest latent.  The exact finishing keeps the reference's
        // operand order: (chain + |x_n|^2) + |e_k|^2 = (chain + code norm) + row norm  (VQ_METRIC_SWAP); cosine uses
        // the operands as given (both normalised by the caller): VQ_METRIC_DOT.
        const int64_t a = (vq_ws_layout(K, N, D).total + 1023) / 1024 * 1024;
        const int64_t b = (vq_cb_layout(N, D).total + 1023) / 1024 * 1024;
        char *pipe_ws = w, *img = w + a;
        const float *codes = (const float *)x;
        if (x_dtype == VQHIP_DTYPE_BF16) {
            float *copy = (float *)(w + a + b);
            int64_t n = N * (int64_t)D;
            int grid = (int)((n + 255) / 256); grid = grid > 4096 ? 4096 : grid;
            bf16_to_f32_kernel<<<grid, 256, 0, s>>>((const uint16_t *)x, n, copy);
            VQ_CHECK_LAUNCH("bf16_to_f32_kernel");
            codes = copy;
        }
        const int m = (metric == VQHIP_METRIC_L2) ? (VQHIP_METRIC_L2 | VQ_METRIC_SWAP) : (VQ_METRIC_DOT | (metric & VQ_METRIC_BF16));
        // statistics of the latents-as-codebook and the token side of the codes-as-rows in one launch, then the image
        int rc = encode_fused_front(e, VQHIP_DTYPE_F32, K, codes, N, D, m, img, pipe_ws, false, nullptr, s);
        if (rc) return rc;
        return argmin_pipeline(e, VQHIP_DTYPE_F32, codes, img, K, N, D, m, col_idx, nullptr, pipe_ws, stream, /*x_prepared=*/true);
    }
    VqWsLayout W = vq_ws_layout(N, K, D);
    u64 *keys = (u64 *)(w + W.off_keys);
    float *en = (float *)(w + W.off_en);
    if (metric == VQHIP_METRIC_L2) {
        int rc = vqhip_row_sqnorm(e, VQHIP_DTYPE_F32, K, D, en, stream);
        if (rc) return rc;
    }
    float *xn = (float *)(w + W.off_xh2);
    if (metric == VQHIP_METRIC_L2) {
        int rc0 = vqhip_row_sqnorm(x, x_dtype, N, D, xn, stream);
        if (rc0) return rc0;
    }
    fill_u64_kernel<<<256, 256, 0, s>>>(keys, K, ~0ull);
    VQ_CHECK_LAUNCH("fill_u64_kernel");
    {
        int rc1 = run_exact_tiled<1>(x, x_dtype, e, en, xn, N, K, D, metric, keys, nullptr, s);
        if (rc1) return rc1;
    }
    finalize_kernel<<<256, 256, 0, s>>>(keys, nullptr, nullptr, K, col_idx, nullptr, nullptr);
    VQ_CHECK_LAUNCH("finalize_kernel");
    return VQHIP_OK;
}

// ---- the column pass over a SHORT list, directly in fp32 ----------------------------------------------------------------
// NearestAnchor over the listed codes is the proposal pipeline with the roles swapped: 5 launches, ~65 us at 3072 tokens x 256
// dims (83 at 6272 x 768) however few codes are listed — and a CVQ-VAE run in its steady state lists a few dozen.  For such a
// list the whole-codebook fp32 pass of the pipeline (exact_kernel: the oracle's k-ordered fma chains, the definition itself)
// IS the cheaper way to the same indices: listed rows x all tokens, one launch behind a tiny one that arms the keys.
// Limit: the pass's work items (32 listed rows x 128 tokens each, a chain of D/2 dependent fp32 MFMAs of 64 cycles) times D —
// up to about one item per SIMD of the chip at D = 256 the pass is one chain long (4-12 us); beyond, the proposal pipeline wins.
#ifndef VQ_COL_DIRECT_MAX_WORK
#define VQ_COL_DIRECT_MAX_WORK (1ll << 18)
#endif
static inline int64_t col_direct_bytes(int64_t N, int64_t cap, int D, int64_t K) {
    (void)cap; (void)D;
    return (K * 8 + 1023) / 1024 * 1024 + 1024 + (N * 4 + 1023) / 1024 * 1024 + (K * 4 + 1023) / 1024 * 1024;
}
static bool col_direct_ok(int x_dtype, int metric, int64_t cap, int64_t N, int64_t K, int D, int64_t ws_bytes) {
    if (!g_tune_col_direct.load() || cap <= 0 || (D % 4) != 0) return false;
    if (x_dtype != VQHIP_DTYPE_F32) return false;        // the pass reads the tokens as fp32 rows, whatever the metric
    if (((cap + 31) / 32) * ((N + 127) / 128) * (int64_t)D > VQ_COL_DIRECT_MAX_WORK) return false;
    return ws_bytes >= col_direct_bytes(N, cap, D, K);
}
// x: the tokens [N, D] as the exact definition consumes them (fp32), e: the codebook rows [K, D] likewise; tok_norm / row_norm:
// oracle-order |x_n|^2 [N] / |e_k|^2 [K] where a caller has them already (L2 only; nullable: computed here)
static int col_rows_direct(const void *x, const float *e, const int32_t *rows, const int32_t *count, int64_t cap, int64_t N,
                           int64_t K, int D, int metric, int64_t *col_idx, char *ws, const float *tok_norm, const float *row_norm,
                           hipStream_t s) {
    u64 *keys = (u64 *)ws;
    int *ticket = (int *)(ws + (K * 8 + 1023) / 1024 * 1024);
    float *en = (float *)((char *)ticket + 1024);
    float *xn = (float *)((char *)en + (N * 4 + 1023) / 1024 * 1024);
    col_direct_init_kernel<<<(int)((cap + 255) / 256), 256, 0, s>>>(rows, count, cap, keys, ticket);
    VQ_CHECK_LAUNCH("col_direct_init_kernel");
    const int m = (metric == VQHIP_METRIC_L2) ? (VQHIP_METRIC_L2 | VQ_METRIC_SWAP) : (VQ_METRIC_DOT | (metric & VQ_METRIC_BF16));
    if (metric == VQHIP_METRIC_L2) {
        if (!tok_norm) { if (int rc = vqhip_row_sqnorm(x, VQHIP_DTYPE_F32, N, D, en, s)) return rc; tok_norm = en; }
        if (!row_norm) { if (int rc = vqhip_row_sqnorm(e, VQHIP_DTYPE_F32, K, D, xn, s)) return rc; row_norm = xn; }
    }
    const int64_t ncb = (N + 63) / 64;
    const int grid = (int)(ncb < 256 ? 256 : (ncb > 1024 ? 1024 : ncb));
    const int few_max = exact_few_max(D);
    const int lds = few_max ? vq_few_lds_bytes(D) : 0;
    static LdsCache lds_set;
    if (int rc = ensure_dyn_lds((const void *)exact_kernel<0>, lds, lds_set)) return rc;
    // roles swapped: the listed codebook rows are the "rows", the tokens the "codes"
    exact_kernel<0><<<grid, 256, lds, s>>>(e, (const float *)x, tok_norm, row_norm, K, N, D, m, (const int *)rows, (const int *)count, keys, ticket,
                                           col_idx, nullptr, few_max, 1);
    VQ_CHECK_LAUNCH("exact_kernel (direct column pass)");
    return VQHIP_OK;
}

int64_t vqhip_col_rows_workspace_bytes(int64_t N, int64_t cap, int D) {
    if (N <= 0 || cap <= 0 || D <= 0 || !vq_coarse_supported(D)) return 0;
    // [pipeline workspace for cap rows against N codes][image of the latents as codes][the listed codebook rows][fp32 copy of bf16 latents]
    const int64_t a = (vq_ws_layout(cap, N, D).total + 1023) / 1024 * 1024;
    const int64_t b = (vq_cb_layout(N, D).total + 1023) / 1024 * 1024;
    const int64_t c = (cap * (int64_t)D * 4 + 1023) / 1024 * 1024;
    // (the direct form of a short list, col_rows_direct, lives in the same buffer: 12 K + 4 N bytes — it is taken only where the
    //  buffer it is handed is large enough for it, which the size below is unless K exceeds ~N D / 2)
    return a + b + c + N * (int64_t)D * 4;
}

int vqhip_col_argmin_rows(const void *x, int x_dtype, const float *e, const int32_t *rows, const int32_t *count, int64_t cap,
                          int64_t N, int64_t K, int D, int metric, int64_t *col_idx, void *ws, int64_t ws_bytes, void *stream) {
    if (!x || !e || !rows || !count || !col_idx || !ws || N <= 0 || K <= 0 || D <= 0 || cap < 0 || cap > K)
        return fail(VQHIP_EINVAL, "vqhip_col_argmin_rows: bad argument");
    if (cap == 0) return VQHIP_OK;
    if (!vq_coarse_supported(D)) return fail(VQHIP_EINVAL, "vqhip_col_argmin_rows: needs D <= 1024, D % 8 == 0 (use vqhip_col_argmin)");
    if (N >= (1ll << 31) || K >= (1ll << 31)) return fail(VQHIP_EINVAL, "vqhip_col_argmin_rows: N or K too large");
    if (x_dtype != VQHIP_DTYPE_F32 && x_dtype != VQHIP_DTYPE_BF16) return fail(VQHIP_EINVAL, "vqhip_col_argmin_rows: x_dtype");
    if (metric != VQHIP_METRIC_L2 && metric != VQHIP_METRIC_COS && metric != VQHIP_METRIC_COS_BF16) return fail(VQHIP_EINVAL, "vqhip_col_argmin_rows: metric");
    VQ_NEED("vqhip_col_argmin_rows: ws too small", ws_bytes, vqhip_col_rows_workspace_bytes(N, cap, D));
    hipStream_t s = (hipStream_t)stream;
    char *w = (char *)ws;
    if (col_direct_ok(x_dtype, metric, cap, N, K, D, ws_bytes))      // a short list: the fp32 pass itself, two launches (four for L2)
        return col_rows_direct(x, e, rows, count, cap, N, K, D, metric, col_idx, w, nullptr, nullptr, s);
    const int64_t a = (vq_ws_layout(cap, N, D).total + 1023) / 1024 * 1024;
    const int64_t b = (vq_cb_layout(N, D).total + 1023) / 1024 * 1024;
    const int64_t c = (cap * (int64_t)D * 4 + 1023) / 1024 * 1024;
    char *pipe_ws = w, *img = w + a;
    float *esub = (float *)(w + a + b);
    // the gather of the listed rows rides in the front launch of the pipeline (token side of pre_kernel) except at a padded
    // dimension of 32, whose token side is the wave-level form
    const bool fused_gather = vq_cb_layout(N, D).nstep != 2;
    if (!fused_gather) {
        gather_listed_rows_kernel<<<waves_grid(cap, 4), 256, 0, s>>>(e, rows, count, cap, D, esub);
        VQ_CHECK_LAUNCH("gather_listed_rows_kernel");
    }
    const float *codes = (const float *)x;
    if (x_dtype == VQHIP_DTYPE_BF16) {
        float *copy = (float *)(w + a + b + c);
        int64_t n = N * (int64_t)D;
        int grid = (int)((n + 255) / 256); grid = grid > 4096 ? 4096 : grid;
        bf16_to_f32_kernel<<<grid, 256, 0, s>>>((const uint16_t *)x, n, copy);
        VQ_CHECK_LAUNCH("bf16_to_f32_kernel");
        codes = copy;
    }
    // the role-swapped pipeline of vqhip_col_argmin on the listed codes: sized for `cap` rows, live for *count of them
    const int m = (metric == VQHIP_METRIC_L2) ? (VQHIP_METRIC_L2 | VQ_METRIC_SWAP) : (VQ_METRIC_DOT | (metric & VQ_METRIC_BF16));
    int rc = fused_gather ? encode_fused_front(e, VQHIP_DTYPE_F32, cap, codes, N, D, m, img, pipe_ws, false, nullptr, s, nullptr, 0, esub, rows, count)
                          : encode_fused_front(esub, VQHIP_DTYPE_F32, cap, codes, N, D, m, img, pipe_ws, false, nullptr, s);
    if (rc) return rc;
    return argmin_pipeline(esub, VQHIP_DTYPE_F32, codes, img, cap, N, D, m, col_idx, nullptr, pipe_ws, stream, /*x_prepared=*/true, count);
}

int vqhip_distance(const void *x, int x_dtype, const float *e, int64_t N, int64_t K, int D, int metric, float *d, void *ws,
                   int64_t ws_bytes, void *stream) {
    if (!x || !e || !d || !ws || N <= 0 || K <= 0 || D <= 0) return fail(VQHIP_EINVAL, "vqhip_distance: bad argument");
    VQ_NEED("vqhip_distance: ws too small", ws_bytes, vqhip_workspace_bytes(N, K, D));
    hipStream_t s = (hipStream_t)stream;
    VqWsLayout W = vq_ws_layout(N, K, D);
    char *w = (char *)ws;
    float *en = (float *)(w + W.off_en);
    if (metric == VQHIP_METRIC_L2) {
        int rc = vqhip_row_sqnorm(e, VQHIP_DTYPE_F32, K, D, en, stream);
        if (rc) return rc;
    }
    float *xn = (float *)(w + W.off_xh2);
    if (metric == VQHIP_METRIC_L2) {
        int rc0 = vqhip_row_sqnorm(x, x_dtype, N, D, xn, stream);
        if (rc0) return rc0;
    }
    if (x_dtype != VQHIP_DTYPE_F32 && x_dtype != VQHIP_DTYPE_BF16) return fail(VQHIP_EINVAL, "vqhip_distance: x_dtype");
    {
        int rc1 = run_exact_tiled<2>(x, x_dtype, e, en, xn, N, K, D, metric, nullptr, d, s);
        if (rc1) return rc1;
    }
    VQ_CHECK_LAUNCH("exact_kernel<dist>");
    return VQHIP_OK;
}

static int gather_ste_impl(const void *x, int x_dtype, const float *e, const int64_t *idx, int64_t N, int D, float *z,
                           float *z_ste, double *sse, float *mse, void *stream, float beta = 0.0f);

int vqhip_gather_ste_loss(const void *x, int x_dtype, const float *e, const int64_t *idx, int64_t N, int D, float *z,
                          float *z_ste, double *sse, void *stream) {
    if (!x || !e || !idx || N < 0 || D <= 0) return fail(VQHIP_EINVAL, "vqhip_gather_ste_loss: bad argument");
    return gather_ste_impl(x, x_dtype, e, idx, N, D, z, z_ste, sse, nullptr, stream);
}

int vqhip_gather_ste_mse(const void *x, int x_dtype, const float *e, const int64_t *idx, int64_t N, int D, float *z,
                         float *z_ste, float *mse, float beta, void *scratch16, void *stream) {
    if (!x || !e || !idx || !mse || !scratch16 || N <= 0 || D <= 0) return fail(VQHIP_EINVAL, "vqhip_gather_ste_mse: bad argument");
    return gather_ste_impl(x, x_dtype, e, idx, N, D, z, z_ste, (double *)scratch16, mse, stream, beta);
}

int vqhip_gather_ste_map(const void *x_rows, int x_dtype, const float *e, const int64_t *idx, int64_t B, int64_t HW, int D,
                         float *out_map, float *mse, float beta, void *scratch16, void *stream) {
    const int64_t N = B * HW;
    if (!e || !idx || !out_map || B <= 0 || HW <= 0 || D <= 0 || (x_rows && (!mse || !scratch16)))
        return fail(VQHIP_EINVAL, "vqhip_gather_ste_map: bad argument");
    hipStream_t s = (hipStream_t)stream;
    double *sse = x_rows ? (double *)scratch16 : nullptr;
    if (x_rows && x_dtype != VQHIP_DTYPE_F32 && x_dtype != VQHIP_DTYPE_BF16) return fail(VQHIP_EINVAL, "vqhip_gather_ste_map: x_dtype");
    if (HW % 256 == 0 && D % 32 == 0 && g_tune_map256.load()) {
        // images of a multiple of 256 positions: whole 1 KiB channel rows per wave-store (gather_ste_map256_kernel)
        const int64_t nt = N / 256;
        int csplit = 1;                                     // share the channels while that still leaves whole chunks and the grid is short
        while (nt * csplit < 512 && (D / 32) % (csplit * 2) == 0) csplit *= 2;
        const int64_t items = nt * csplit;
        const int grid256 = (int)(items < 1024 ? items : 1024);
        constexpr int LDS = 2 * 32 * 256 * 4;
        static LdsCache sets[2];
        const int bf = (x_rows && x_dtype == VQHIP_DTYPE_BF16) ? 1 : 0;
        if (int rc = ensure_dyn_lds(bf ? (const void *)gather_ste_map256_kernel<1> : (const void *)gather_ste_map256_kernel<0>, LDS, sets[bf])) return rc;
        if (!bf) gather_ste_map256_kernel<0><<<grid256, 512, LDS, s>>>(x_rows, e, idx, N, D, HW, csplit, out_map, sse, mse, beta);
        else gather_ste_map256_kernel<1><<<grid256, 512, LDS, s>>>(x_rows, e, idx, N, D, HW, csplit, out_map, sse, mse, beta);
        VQ_CHECK_LAUNCH("gather_ste_map256_kernel");
        return VQHIP_OK;
    }
    int64_t ntiles = (N + 63) / 64;
    const int grid = (int)(ntiles < 2048 ? ntiles : 2048);
    if (!x_rows || x_dtype == VQHIP_DTYPE_F32) gather_ste_map_kernel<0><<<grid, 256, 0, s>>>(x_rows, e, idx, N, D, HW, out_map, sse, mse, beta);
    else if (x_dtype == VQHIP_DTYPE_BF16) gather_ste_map_kernel<1><<<grid, 256, 0, s>>>(x_rows, e, idx, N, D, HW, out_map, sse, mse, beta);
    else return fail(VQHIP_EINVAL, "vqhip_gather_ste_map: x_dtype");
    VQ_CHECK_LAUNCH("gather_ste_map_kernel");
    return VQHIP_OK;
}

static int gather_ste_impl(const void *x, int x_dtype, const float *e, const int64_t *idx, int64_t N, int D, float *z,
                           float *z_ste, double *sse, float *mse, void *stream, float beta) {
    if (N == 0) return VQHIP_OK;
    hipStream_t s = (hipStream_t)stream;
    // outputs beyond the Infinity Cache (256 MiB) are streamed: non-temporal accesses and twice the waves in flight
    const int64_t out_bytes = (int64_t)N * D * 4 * ((z ? 1 : 0) + (z_ste ? 1 : 0));
    const int tune_nt = g_tune_gather_nt.load(), tune_grid = g_tune_gather_grid.load();
    const bool streamed = tune_nt ? tune_nt == 1 : out_bytes > (192ll << 20);
    int cap = tune_grid > 0 ? tune_grid : (streamed ? 512 : 256);      // blocks of 16 waves
    int grid = (int)((N + 15) / 16);
    grid = grid > cap ? cap : grid;
#define VQ_GATHER(DT, NT) gather_ste_loss_kernel<DT, NT><<<grid, 1024, 0, s>>>(x, e, idx, N, D, z, z_ste, sse, mse, beta)
    if (x_dtype == VQHIP_DTYPE_F32) { if (streamed) VQ_GATHER(0, 1); else VQ_GATHER(0, 0); }
    else if (x_dtype == VQHIP_DTYPE_BF16) { if (streamed) VQ_GATHER(1, 1); else VQ_GATHER(1, 0); }
    else return fail(VQHIP_EINVAL, "vqhip_gather_ste_loss: x_dtype");
#undef VQ_GATHER
    VQ_CHECK_LAUNCH("gather_ste_loss_kernel");
    return VQHIP_OK;
}

int vqhip_hist(const int64_t *idx, int64_t N, int64_t K, int32_t *hist, void *stream) {
    if (!idx || !hist || N < 0 || K <= 0) return fail(VQHIP_EINVAL, "vqhip_hist: bad argument");
    if (N == 0) return VQHIP_OK;
    if (K <= 32768 && N >= 16384) {
        // enough tokens per block that the K-bin flush pays: at most 256 blocks, >= 2048 tokens each
        static LdsCache lds_set;
        if (int rc = ensure_dyn_lds((const void *)hist_lds_kernel, (size_t)K * 4, lds_set)) return rc;
        int grid = (int)((N + 2047) / 2048); grid = grid > 256 ? 256 : grid;
        hist_lds_kernel<<<grid, 1024, (size_t)K * 4, (hipStream_t)stream>>>(idx, N, (int)K, hist);
    } else {
        int grid = (int)((N + 255) / 256); grid = grid > 2048 ? 2048 : grid;
        hist_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(idx, N, K, hist);
    }
    VQ_CHECK_LAUNCH("hist_kernel");
    return VQHIP_OK;
}

int vqhip_scatter_add_rows(const float *src, const int64_t *idx, int64_t N, int64_t K, int D, float *dst, void *stream) {
    if (!src || !idx || !dst || N < 0 || K <= 0 || D <= 0) return fail(VQHIP_EINVAL, "vqhip_scatter_add_rows: bad argument");
    if (N == 0) return VQHIP_OK;
    scatter_add_rows_kernel<<<waves_grid(N, 4), 256, 0, (hipStream_t)stream>>>(src, idx, N, K, D, dst);
    VQ_CHECK_LAUNCH("scatter_add_rows_kernel");
    return VQHIP_OK;
}

int vqhip_gather_rows(const void *x, int x_dtype, const int64_t *row_idx, int64_t K, int D, float *out, void *stream) {
    if (!x || !row_idx || !out || K < 0 || D <= 0) return fail(VQHIP_EINVAL, "vqhip_gather_rows: bad argument");
    if (K == 0) return VQHIP_OK;
    hipStream_t s = (hipStream_t)stream;
    if (x_dtype == VQHIP_DTYPE_F32) gather_rows_kernel<0><<<waves_grid(K, 4), 256, 0, s>>>(x, row_idx, K, D, out);
    else if (x_dtype == VQHIP_DTYPE_BF16) gather_rows_kernel<1><<<waves_grid(K, 4), 256, 0, s>>>(x, row_idx, K, D, out);
    else return fail(VQHIP_EINVAL, "vqhip_gather_rows: x_dtype");
    VQ_CHECK_LAUNCH("gather_rows_kernel");
    return VQHIP_OK;
}

int vqhip_vqkd_update(float *w, const int64_t *hist, const float *sums, int64_t K, int D, float decay, int centroid_only,
                      void *stream) {
    if (!w || !hist || !sums || K <= 0 || D <= 0) return fail(VQHIP_EINVAL, "vqhip_vqkd_update: bad argument");
    vqkd_update_kernel<<<waves_grid(K, 4), 256, 0, (hipStream_t)stream>>>(w, hist, sums, K, D, decay, centroid_only);
    VQ_CHECK_LAUNCH("vqkd_update_kernel");
    return VQHIP_OK;
}

int vqhip_cvq_update(float *w, float *p, const int64_t *hist, int64_t numel, const int64_t *numel_dev, const float *anchors,
                     int64_t K, int D, float ema_decay, float eps, int stage, void *stream) {
    if (!w || !p || K <= 0 || D <= 0 || stage < 1 || stage > 3 || ((stage & 1) && (!hist || (numel <= 0 && !numel_dev))) ||
        ((stage & 2) && !anchors)) return fail(VQHIP_EINVAL, "vqhip_cvq_update: bad argument");
    cvq_update_kernel<<<waves_grid(K, 4), 256, 0, (hipStream_t)stream>>>(w, p, hist, numel, numel_dev, anchors, K, D, ema_decay, eps, stage);
    VQ_CHECK_LAUNCH("cvq_update_kernel");
    return VQHIP_OK;
}

int vqhip_cvq_step(const float *w_in, float *w_out, const float *p_in, float *p_out, const int32_t *hist, int64_t numel,
                   const void *x, int x_dtype, const int64_t *col_idx, int64_t K, int D, float ema_decay, float eps,
                   void *stream) {
    if (!w_in || !w_out || !p_in || !p_out || !hist || !x || !col_idx || K <= 0 || D <= 0 || numel <= 0)
        return fail(VQHIP_EINVAL, "vqhip_cvq_step: bad argument");
    hipStream_t s = (hipStream_t)stream;
    if (x_dtype == VQHIP_DTYPE_F32) cvq_step_kernel<0><<<waves_grid(K, 4), 256, 0, s>>>(w_in, w_out, p_in, p_out, hist, numel, x, col_idx, K, D, ema_decay, eps);
    else if (x_dtype == VQHIP_DTYPE_BF16) cvq_step_kernel<1><<<waves_grid(K, 4), 256, 0, s>>>(w_in, w_out, p_in, p_out, hist, numel, x, col_idx, K, D, ema_decay, eps);
    else return fail(VQHIP_EINVAL, "vqhip_cvq_step: x_dtype");
    VQ_CHECK_LAUNCH("cvq_step_kernel");
    return VQHIP_OK;
}

int vqhip_cvq_decay(const float *p, int64_t K, float ema_decay, float eps, float *decay, void *stream) {
    if (!p || !decay || K <= 0) return fail(VQHIP_EINVAL, "vqhip_cvq_decay: bad argument");
    cvq_decay_kernel<<<(int)((K + 255) / 256), 256, 0, (hipStream_t)stream>>>(p, K, ema_decay, eps, decay);
    VQ_CHECK_LAUNCH("cvq_decay_kernel");
    return VQHIP_OK;
}

int vqhip_cvq_update_rows(float *w, const float *p, const int64_t *rows, const float *anchors_sub, int64_t M, int64_t K, int D,
                          float ema_decay, float eps, void *stream) {
    if (!w || !p || K <= 0 || D <= 0 || M < 0 || (M > 0 && (!rows || !anchors_sub)))
        return fail(VQHIP_EINVAL, "vqhip_cvq_update_rows: bad argument");
    if (M == 0) return VQHIP_OK;
    cvq_update_rows_kernel<<<waves_grid(M, 4), 256, 0, (hipStream_t)stream>>>(w, p, rows, anchors_sub, M, K, D, ema_decay, eps);
    VQ_CHECK_LAUNCH("cvq_update_rows_kernel");
    return VQHIP_OK;
}

int vqhip_cvq_rows(const float *p, int64_t K, float ema_decay, float eps, int32_t *rows, int32_t *slot, int32_t *count,
                   void *stream) {
    if (!p || !rows || !slot || !count || K <= 0 || K >= (1ll << 31)) return fail(VQHIP_EINVAL, "vqhip_cvq_rows: bad argument");
    cvq_rows_kernel<<<1, 1024, 0, (hipStream_t)stream>>>(p, K, ema_decay, eps, rows, slot, count);
    VQ_CHECK_LAUNCH("cvq_rows_kernel");
    return VQHIP_OK;
}

int64_t vqhip_pack_floats(int64_t K, int64_t M, int D) {
    if (K <= 0 || M < 0 || D <= 0) return 0;
    return VQ_PACK_HEADER(K) + M * (int64_t)D;
}

int vqhip_pack_counts(const void *hist, int hist_is_int64, int64_t numel, int64_t K, float *packed, void *stream) {
    if (!hist || !packed || K <= 0 || numel < 0) return fail(VQHIP_EINVAL, "vqhip_pack_counts: bad argument");
    const int grid = (int)((K + 255) / 256);
    if (hist_is_int64) pack_counts_kernel<1><<<grid, 256, 0, (hipStream_t)stream>>>(hist, numel, K, packed);
    else pack_counts_kernel<0><<<grid, 256, 0, (hipStream_t)stream>>>(hist, numel, K, packed);
    VQ_CHECK_LAUNCH("pack_counts_kernel");
    return VQHIP_OK;
}

int vqhip_unpack_counts(const float *packed, int64_t K, int64_t *out, void *stream) {
    if (!packed || !out || K <= 0) return fail(VQHIP_EINVAL, "vqhip_unpack_counts: bad argument");
    unpack_counts_kernel<<<(int)((K + 255) / 256), 256, 0, (hipStream_t)stream>>>(packed, K, out);
    VQ_CHECK_LAUNCH("unpack_counts_kernel");
    return VQHIP_OK;
}

int vqhip_cvq_pack(const int32_t *hist, int64_t numel, const void *x, int x_dtype, const int64_t *col_idx, const int32_t *count,
                   int64_t cap, int64_t K, int D, float *packed, void *stream) {
    if (!hist || !packed || K <= 0 || D <= 0 || numel <= 0 || cap < 0 || cap > K || (cap > 0 && (!x || !col_idx || !count)))
        return fail(VQHIP_EINVAL, "vqhip_cvq_pack: bad argument");
    hipStream_t s = (hipStream_t)stream;
    const int hb = (int)((K + 255) / 256);
    const int grid = hb + waves_grid(cap, 4);
    if (x_dtype == VQHIP_DTYPE_BF16) cvq_pack_kernel<1><<<grid, 256, 0, s>>>(hist, numel, x, col_idx, count, cap, K, D, packed, hb);
    else if (x_dtype == VQHIP_DTYPE_F32) cvq_pack_kernel<0><<<grid, 256, 0, s>>>(hist, numel, x, col_idx, count, cap, K, D, packed, hb);
    else return fail(VQHIP_EINVAL, "vqhip_cvq_pack: x_dtype");
    VQ_CHECK_LAUNCH("cvq_pack_kernel");
    return VQHIP_OK;
}

int vqhip_cvq_apply(const float *w_in, float *w_out, const float *p_in, float *p_out, const int32_t *hist, int64_t numel,
                    const void *x, int x_dtype, const int64_t *col_idx, const float *packed, int world, const int32_t *slot,
                    int64_t cap, int64_t K, int D, float ema_decay, float eps, void *stream) {
    if (!w_in || !w_out || !p_in || !p_out || !slot || K <= 0 || D <= 0 || cap < 0 || cap > K) return fail(VQHIP_EINVAL, "vqhip_cvq_apply: bad argument");
    hipStream_t s = (hipStream_t)stream;
    if (packed != nullptr) {
        if (world < 1 || world > VQ_PACK_MAX_WORLD) return fail(VQHIP_EINVAL, "vqhip_cvq_apply: world must be in [1, 256] (exact fp32 count sums)");
        cvq_apply_kernel<0, true><<<waves_grid(K, 4), 256, 0, s>>>(w_in, w_out, p_in, p_out, nullptr, 0, nullptr, nullptr, packed, world, slot, (int)cap, K, D, ema_decay, eps);
    } else {
        if (!hist || numel <= 0) return fail(VQHIP_EINVAL, "vqhip_cvq_apply: the one-rank form needs hist and numel");   // (x, col_idx: read for listed codes only)
        if (x_dtype == VQHIP_DTYPE_F32) cvq_apply_kernel<0, false><<<waves_grid(K, 4), 256, 0, s>>>(w_in, w_out, p_in, p_out, hist, numel, x, col_idx, nullptr, 1, slot, (int)cap, K, D, ema_decay, eps);
        else if (x_dtype == VQHIP_DTYPE_BF16) cvq_apply_kernel<1, false><<<waves_grid(K, 4), 256, 0, s>>>(w_in, w_out, p_in, p_out, hist, numel, x, col_idx, nullptr, 1, slot, (int)cap, K, D, ema_decay, eps);
        else return fail(VQHIP_EINVAL, "vqhip_cvq_apply: x_dtype");
    }
    VQ_CHECK_LAUNCH("cvq_apply_kernel");
    return VQHIP_OK;
}

// ---- NearestAnchor(sync=True) across ranks: keys of the local winners, the masked pack (include/vqhip.h) ---------------
int vqhip_cvq_col_keys(const void *x, int x_dtype, const float *e, const int32_t *rows, const int32_t *count, int64_t cap,
                       const int64_t *col_idx, int64_t N, int64_t K, int D, int metric, int rank, int64_t *keys, void *stream) {
    if (!x || !e || !rows || !count || !col_idx || !keys || N <= 0 || K <= 0 || D <= 0 || cap < 0 || cap > K)
        return fail(VQHIP_EINVAL, "vqhip_cvq_col_keys: bad argument");
    if (N > VQHIP_SYNC_MAX_ROWS || rank < 0 || rank >= VQ_PACK_MAX_WORLD)
        return fail(VQHIP_EINVAL, "vqhip_cvq_col_keys: a key holds 24 bits of row and 8 bits of rank (N <= 2^24, rank < 256)");
    if (metric != VQHIP_METRIC_L2 && metric != VQHIP_METRIC_COS && metric != VQHIP_METRIC_COS_BF16) return fail(VQHIP_EINVAL, "vqhip_cvq_col_keys: metric");
    if (cap == 0) return VQHIP_OK;
    hipStream_t s = (hipStream_t)stream;
    const int grid = (int)((cap + 63) / 64);
    if (x_dtype == VQHIP_DTYPE_F32) cvq_col_keys_kernel<0><<<grid, 64, 0, s>>>(x, e, rows, count, cap, col_idx, D, metric, rank, keys);
    else if (x_dtype == VQHIP_DTYPE_BF16) cvq_col_keys_kernel<1><<<grid, 64, 0, s>>>(x, e, rows, count, cap, col_idx, D, metric, rank, keys);
    else return fail(VQHIP_EINVAL, "vqhip_cvq_col_keys: x_dtype");
    VQ_CHECK_LAUNCH("cvq_col_keys_kernel");
    return VQHIP_OK;
}

int vqhip_cvq_pack_sync(const int32_t *hist, int64_t numel, const void *x, int x_dtype, const int64_t *keys, const int32_t *count,
                        int64_t cap, int rank, int64_t K, int D, float *packed, void *stream) {
    if (!hist || !packed || K <= 0 || D <= 0 || numel <= 0 || cap < 0 || cap > K || rank < 0 || rank >= VQ_PACK_MAX_WORLD ||
        (cap > 0 && (!x || !keys || !count)))
        return fail(VQHIP_EINVAL, "vqhip_cvq_pack_sync: bad argument");
    hipStream_t s = (hipStream_t)stream;
    const int hb = (int)((K + 255) / 256);
    const int grid = hb + waves_grid(cap, 4);
    if (x_dtype == VQHIP_DTYPE_BF16) cvq_pack_sync_kernel<1><<<grid, 256, 0, s>>>(hist, numel, x, keys, count, cap, rank, K, D, packed, hb);
    else if (x_dtype == VQHIP_DTYPE_F32) cvq_pack_sync_kernel<0><<<grid, 256, 0, s>>>(hist, numel, x, keys, count, cap, rank, K, D, packed, hb);
    else return fail(VQHIP_EINVAL, "vqhip_cvq_pack_sync: x_dtype");
    VQ_CHECK_LAUNCH("cvq_pack_sync_kernel");
    return VQHIP_OK;
}

// ---- one host call per training forward (include/vqhip.h) ------------------------------------------------------------
static inline int64_t vq_align1k(int64_t v) { return (v + 1023) / 1024 * 1024; }

// workspace of vqhip_cvq_forward: [row pass: vqhip_workspace_bytes(N, K, D)][col_idx: K int64][column pass over <= cap_max codes]
int64_t vqhip_cvq_forward_ws_bytes(int64_t N, int64_t K, int D, int64_t cap_max) {
    if (N <= 0 || K <= 0 || D <= 0 || cap_max < 0 || cap_max > K) return 0;
    return vq_align1k(vqhip_workspace_bytes(N, K, D)) + vq_align1k(K * 8) + (cap_max > 0 ? vqhip_col_rows_workspace_bytes(N, cap_max, D) : 0);
}

int vqhip_cvq_forward(vqhip_cvq_forward_t *a, void *stream) {
    if (!a || a->struct_bytes != (int64_t)sizeof(vqhip_cvq_forward_t)) return fail(VQHIP_EINVAL, "vqhip_cvq_forward: struct_bytes != sizeof(vqhip_cvq_forward_t)");
    const int64_t N = a->N, K = a->K;
    const int D = a->D, metric = a->metric;
    if (N <= 0 || K <= 0 || D <= 0 || N >= (1ll << 31) || K >= (1ll << 31) || !vq_coarse_supported(D))
        return fail(VQHIP_EINVAL, "vqhip_cvq_forward: needs N, K > 0 and D <= 1024, D % 8 == 0");
    if (metric != VQHIP_METRIC_L2 && metric != VQHIP_METRIC_COS && metric != VQHIP_METRIC_COS_BF16) return fail(VQHIP_EINVAL, "vqhip_cvq_forward: metric");
    if (a->x_dtype != VQHIP_DTYPE_F32 && a->x_dtype != VQHIP_DTYPE_BF16) return fail(VQHIP_EINVAL, "vqhip_cvq_forward: x_dtype");
    const bool sync = a->exchange && a->anchor_sync;
    if (a->phases < 1 || (a->phases > VQHIP_STEP_ALL && !(sync && a->phases == VQHIP_STEP_PACK_SYNC))) return fail(VQHIP_EINVAL, "vqhip_cvq_forward: phases");
    if (!a->x || !a->w_in || !a->p_in || !a->w_out || !a->p_out || !a->rows || !a->slot || !a->count || !a->cb || !a->idx || !a->hist || !a->ws)
        return fail(VQHIP_EINVAL, "vqhip_cvq_forward: null pointer");
    const bool cos = VQ_IS_COS(metric);
    if (cos && !a->xq) return fail(VQHIP_EINVAL, "vqhip_cvq_forward: the cosine metric needs the xq buffer");
    if ((a->z_ste || a->mse) && (!a->mse || !a->scratch16)) return fail(VQHIP_EINVAL, "vqhip_cvq_forward: the decode tail needs mse and scratch16");
    if (a->exchange) {
        if (!a->packed) return fail(VQHIP_EINVAL, "vqhip_cvq_forward: exchange without a packed buffer");
        if (a->world < 1 || a->world > VQ_PACK_MAX_WORLD) return fail(VQHIP_EINVAL, "vqhip_cvq_forward: world must be in [1, 256]");
        if (a->phases == VQHIP_STEP_ALL && !a->comm && a->world > 1)
            return fail(VQHIP_EINVAL, "vqhip_cvq_forward: VQHIP_STEP_ALL over more than one rank needs a communicator (or issue the collective between the two phases)");
    }
    if (sync) {
        if (!a->keys) return fail(VQHIP_EINVAL, "vqhip_cvq_forward: anchor_sync needs the keys buffer");
        if (a->rank < 0 || a->rank >= a->world) return fail(VQHIP_EINVAL, "vqhip_cvq_forward: anchor_sync needs this rank's number in [0, world)");
        if (N > VQHIP_SYNC_MAX_ROWS) return fail(VQHIP_EINVAL, "vqhip_cvq_forward: anchor_sync holds a row in 24 bits (N <= 2^24 per rank)");
    }
    hipStream_t s = (hipStream_t)stream;
    const int64_t enc_bytes = vq_align1k(vqhip_workspace_bytes(N, K, D));
    VQ_NEED("vqhip_cvq_forward: ws too small", a->ws_bytes, enc_bytes + vq_align1k(K * 8));
    char *w = (char *)a->ws;
    int64_t *col_idx = (int64_t *)(w + enc_bytes);
    char *col_ws = w + enc_bytes + vq_align1k(K * 8);
    const int64_t col_ws_have = a->ws_bytes - enc_bytes - vq_align1k(K * 8);
    if (a->phases != VQHIP_STEP_PACK_SYNC && (a->phases & VQHIP_STEP_BEFORE_EXCHANGE)) {
        if (int rc = vqhip_encode_ex(a->x, a->x_dtype, a->w_in, N, K, D, metric, a->cb, a->cb_bytes, a->idx, a->hist, a->xq, a->ws,
                                     enc_bytes, VQHIP_ENCODE_ZERO_HIST, stream)) return rc;
        if (!a->list_ready)
            if (int rc = vqhip_cvq_rows(a->p_in, K, a->ema_decay, a->eps, a->rows, a->slot, a->count, stream)) return rc;
        if (!a->exchange && a->early_word_host && a->early_seq_dev) {     // one rank: the histogram is final behind the encode
            cvq_count_next_kernel<false><<<1, 1024, 0, s>>>(a->p_in, a->hist, N, nullptr, K, a->ema_decay, a->eps, a->early_seq_dev, a->early_word_host);
            VQ_CHECK_LAUNCH("cvq_count_next_kernel");
        }
        int64_t cap = a->cap;
        if (cap < 0) {                   // the count the previous call's prefetch copied out: queued a whole step ago
            if (!a->count_host) return fail(VQHIP_EINVAL, "vqhip_cvq_forward: cap < 0 needs count_host");
            if (a->count_event) VQ_HIP(hipEventSynchronize((hipEvent_t)a->count_event));
            cap = (int64_t)*(volatile const int32_t *)a->count_host;
        }
        if (cap < 0 || cap > K) return fail(VQHIP_EINVAL, "vqhip_cvq_forward: capacity outside [0, K]");
        a->cap_used = cap;
        a->exchange_floats = a->exchange ? vqhip_pack_floats(K, cap, D) : 0;
        if (a->exchange) VQ_NEED("vqhip_cvq_forward: packed buffer too small (floats)", a->packed_floats, a->exchange_floats);
        if (cap > 0) {
            VQ_NEED("vqhip_cvq_forward: ws too small for the column pass", col_ws_have, vqhip_col_rows_workspace_bytes(N, cap, D));
            const VqCbLayout L = vq_cb_layout(K, D);
            const void *rows_x = cos ? (const void *)a->xq : a->x;
            const float *codes = cos ? (const float *)((const char *)a->cb + L.off_eexact) : a->w_in;
            if (int rc = vqhip_col_argmin_rows(rows_x, cos ? VQHIP_DTYPE_F32 : a->x_dtype, codes, a->rows, a->count, cap, N, K, D, metric,
                                               col_idx, col_ws, col_ws_have, stream)) return rc;
            if (sync)                    // the local winners' keys: the ranks agree on the global winner by a MIN all-reduce
                if (int rc = vqhip_cvq_col_keys(rows_x, cos ? VQHIP_DTYPE_F32 : a->x_dtype, codes, a->rows, a->count, cap, col_idx, N, K, D, metric,
                                                a->rank, a->keys, stream)) return rc;
        }
        if (a->exchange && !sync)
            if (int rc = vqhip_cvq_pack(a->hist, N, a->x, a->x_dtype, col_idx, a->count, cap, K, D, a->packed, stream)) return rc;
    }
    if (sync && (a->phases == VQHIP_STEP_ALL || a->phases == VQHIP_STEP_PACK_SYNC)) {
        if (a->cap_used < 0 || a->cap_used > K) return fail(VQHIP_EINVAL, "vqhip_cvq_forward: cap_used (the BEFORE phase writes it)");
        if (a->phases == VQHIP_STEP_ALL && a->comm && a->cap_used > 0)
            if (int rc = vqhip_allreduce_min_i64(a->keys, a->cap_used, a->comm, stream)) return rc;
        if (int rc = vqhip_cvq_pack_sync(a->hist, N, a->x, a->x_dtype, a->keys, a->count, a->cap_used, a->rank, K, D, a->packed, stream)) return rc;
    }
    if (a->phases == VQHIP_STEP_ALL && a->exchange && a->comm)
        if (int rc = vqhip_allreduce_packed(a->packed, a->exchange_floats, a->comm, stream)) return rc;
    if (a->phases != VQHIP_STEP_PACK_SYNC && (a->phases & VQHIP_STEP_AFTER_EXCHANGE)) {
        if (a->cap_used < 0 || a->cap_used > K) return fail(VQHIP_EINVAL, "vqhip_cvq_forward: cap_used (the BEFORE phase writes it)");
        if (a->exchange && a->early_word_host && a->early_seq_dev) {      // the next list's length, right behind the reduced histogram
            cvq_count_next_kernel<true><<<1, 1024, 0, s>>>(a->p_in, nullptr, 0, a->packed, K, a->ema_decay, a->eps, a->early_seq_dev, a->early_word_host);
            VQ_CHECK_LAUNCH("cvq_count_next_kernel");
        }
        // sync: the packed rows ARE the global winners' latents (every other rank added -0): no averaging (anchors.py:59-63)
        if (int rc = vqhip_cvq_apply(a->w_in, a->w_out, a->p_in, a->p_out, a->hist, N, a->x, a->x_dtype, col_idx, a->exchange ? a->packed : nullptr,
                                     sync ? 1 : a->world, a->slot, a->cap_used, K, D, a->ema_decay, a->eps, stream)) return rc;
        if (a->prefetch) {
            // the list of the NEXT step; its length also goes straight to the caller's pinned host word (a store of the kernel
            // itself — pinned host memory is device-visible at its own address — instead of a 4-byte copy launch behind it)
            cvq_rows_kernel<<<1, 1024, 0, s>>>(a->p_out, K, a->ema_decay, a->eps, a->rows, a->slot, a->count, a->count_host);
            VQ_CHECK_LAUNCH("cvq_rows_kernel");
            if (a->count_host && a->count_event) VQ_HIP(hipEventRecord((hipEvent_t)a->count_event, s));
        }
        if (a->mse)
            if (int rc = vqhip_gather_ste_mse(a->x, a->x_dtype, a->w_out, a->idx, N, D, nullptr, a->z_ste, a->mse, a->beta, a->scratch16, stream)) return rc;
    }
    return VQHIP_OK;
}

// workspace of vqhip_vqkd_forward: [encode: vqhip_workspace_bytes(N, K, D)][ordered sums: counts K, offsets K + 1, order N int32,
// vqhip_order_workspace_bytes, vqhip_segsum_workspace_bytes]
static inline void vqkd_ws_offsets(int64_t N, int64_t K, int D, int64_t *o_counts, int64_t *o_offsets, int64_t *o_order, int64_t *o_ows,
                                   int64_t *o_sws, int64_t *o_x2, int64_t *total) {
    int64_t p = vq_align1k(vq_ws_layout(N, K, D).total);
    *o_x2 = p; p += vq_align1k(N * (int64_t)D * 4);        // F.normalize(xn) for the ordered sums under the bf16-autocast metric (xq is rounded there)
    *o_counts = p; p += vq_align1k(K * 4);
    *o_offsets = p; p += vq_align1k((K + 1) * 4);
    *o_order = p; p += vq_align1k(N * 4);
    *o_ows = p; p += vq_align1k(vqhip_order_workspace_bytes(N, K));
    *o_sws = p; p += vq_align1k(vqhip_segsum_workspace_bytes(N, D));
    *total = p;
}

int64_t vqhip_vqkd_forward_ws_bytes(int64_t N, int64_t K, int D) {
    if (N <= 0 || K <= 0 || D <= 0) return 0;
    int64_t a, b, c, d, e, f, total;
    vqkd_ws_offsets(N, K, D, &a, &b, &c, &d, &e, &f, &total);
    return total;
}

int vqhip_vqkd_forward(vqhip_vqkd_forward_t *a, void *stream) {
    if (!a || a->struct_bytes != (int64_t)sizeof(vqhip_vqkd_forward_t)) return fail(VQHIP_EINVAL, "vqhip_vqkd_forward: struct_bytes != sizeof(vqhip_vqkd_forward_t)");
    const int64_t N = a->N, K = a->K;
    const int D = a->D;
    if (N <= 0 || K <= 0 || D <= 0 || N >= (1ll << 31) || K >= (1ll << 31) || !vq_coarse_supported(D))
        return fail(VQHIP_EINVAL, "vqhip_vqkd_forward: needs N, K > 0 and D <= 1024, D % 8 == 0");
    if (a->metric != VQHIP_METRIC_COS && a->metric != VQHIP_METRIC_COS_BF16) return fail(VQHIP_EINVAL, "vqhip_vqkd_forward: metric (cosine only)");
    if (a->x_dtype != VQHIP_DTYPE_F32 && a->x_dtype != VQHIP_DTYPE_BF16) return fail(VQHIP_EINVAL, "vqhip_vqkd_forward: x_dtype");
    if (a->phases < 1 || a->phases > VQHIP_STEP_ALL) return fail(VQHIP_EINVAL, "vqhip_vqkd_forward: phases");
    if (!a->x || !a->w_in || !a->w_mid || !a->w_out || !a->xn || !a->xq || !a->cb || !a->idx || !a->hist || !a->packed || !a->ws)
        return fail(VQHIP_EINVAL, "vqhip_vqkd_forward: null pointer");
    if (a->tail && (!a->mse || !a->scratch16)) return fail(VQHIP_EINVAL, "vqhip_vqkd_forward: the tail needs mse and scratch16");
    if (a->world < 1 || a->world > VQ_PACK_MAX_WORLD) return fail(VQHIP_EINVAL, "vqhip_vqkd_forward: world must be in [1, 256]");
    if (a->phases == VQHIP_STEP_ALL && a->exchange && !a->comm && a->world > 1)
        return fail(VQHIP_EINVAL, "vqhip_vqkd_forward: VQHIP_STEP_ALL over more than one rank needs a communicator (or issue the collective between the two phases)");
    if (a->ordered && (K > 32768 || (D % 4) != 0)) return fail(VQHIP_EINVAL, "vqhip_vqkd_forward: ordered sums need K <= 32768 and D % 4 == 0");
    const int64_t floats = vqhip_pack_floats(K, K, D);
    VQ_NEED("vqhip_vqkd_forward: packed buffer too small (floats)", a->packed_floats, floats);
    int64_t o_counts, o_offsets, o_order, o_ows, o_sws, o_x2, total;
    vqkd_ws_offsets(N, K, D, &o_counts, &o_offsets, &o_order, &o_ows, &o_sws, &o_x2, &total);
    VQ_NEED("vqhip_vqkd_forward: ws too small", a->ws_bytes, total);
    hipStream_t s = (hipStream_t)stream;
    char *w = (char *)a->ws;
    float *payload = a->packed + VQ_PACK_HEADER(K);
    if (a->phases & VQHIP_STEP_BEFORE_EXCHANGE) {
        a->exchange_floats = floats;
        // front: codebook normalised twice, latents normalised, payload zeroed (the atomic sums need it; the ordered sums write every row)
        int64_t nzero = a->ordered ? 0 : K * (int64_t)D;
        if (nzero && ((K & 1) || (nzero % 4))) {            // a payload that is not 16-byte aligned / sized: plain memset
            VQ_HIP(hipMemsetAsync(payload, 0, (size_t)nzero * 4, s));
            nzero = 0;
        }
        if (int rc = launch_vqkd_front(a->w_in, a->w_mid, K, a->x, a->x_dtype, a->xn, N, D, 1e-12f, payload, nzero, 2, s)) return rc;
        const int xblocks = (int)((N + 3) / 4);             // wave per token: the scatter below
        if (int rc = vqhip_encode_ex(a->xn, VQHIP_DTYPE_F32, a->w_mid, N, K, D, a->metric, a->cb, a->cb_bytes, a->idx, a->hist, a->xq, a->ws,
                                     vq_align1k(vq_ws_layout(N, K, D).total), VQHIP_ENCODE_ZERO_HIST, stream)) return rc;
        const int hb = (int)((K + 255) / 256);
        if (a->ordered) {
            int32_t *counts = (int32_t *)(w + o_counts), *offsets = (int32_t *)(w + o_offsets), *order = (int32_t *)(w + o_order);
            if (int rc = vqhip_token_order(a->idx, N, K, counts, offsets, order, w + o_ows, vqhip_order_workspace_bytes(N, K), stream)) return rc;
            const float *x2 = a->xq;                      // F.normalize(xn): the encode's by-product — except under the bf16-autocast
            if (VQ_IS_BF16(a->metric)) {                  // metric, where xq holds the ROUNDED rows (the operand of that metric)
                if (int rc = vqhip_normalize_rows(a->xn, VQHIP_DTYPE_F32, N, D, 1e-12f, (float *)(w + o_x2), stream)) return rc;
                x2 = (const float *)(w + o_x2);
            }
            if (int rc = vqhip_segsum_rows(x2, a->idx, order, offsets, N, K, D, payload, w + o_sws, vqhip_segsum_workspace_bytes(N, D), stream)) return rc;
            vqkd_scatter_pack_kernel<<<hb, 256, 0, s>>>(a->hist, N, a->xn, a->idx, N, K, D, 1e-12f, a->packed, hb, 0);
        } else {
            vqkd_scatter_pack_kernel<<<hb + xblocks, 256, 0, s>>>(a->hist, N, a->xn, a->idx, N, K, D, 1e-12f, a->packed, hb, 1);
        }
        VQ_CHECK_LAUNCH("vqkd_scatter_pack_kernel");
    }
    if (a->phases == VQHIP_STEP_ALL && a->exchange && a->comm)
        if (int rc = vqhip_allreduce_packed(a->packed, floats, a->comm, stream)) return rc;
    if (a->phases & VQHIP_STEP_AFTER_EXCHANGE) {
        vqkd_update_packed_kernel<<<waves_grid(K, 4), 256, 0, s>>>(a->w_mid, a->w_out, a->packed, K, D, a->ema_decay);
        VQ_CHECK_LAUNCH("vqkd_update_packed_kernel");
        if (a->tail) {
            // per-workgroup partial sums of the loss (<= 256 doubles): the record area of the encode's workspace, free by now
            double *tail_partials = (double *)((char *)a->ws + vq_ws_layout(N, K, D).off_rec);
            if (D <= 32) {                                  // L lanes per token (vqhip_step_kernels.h)
                const int L = D <= 8 ? 8 : (D <= 16 ? 16 : 32), rpb = 16 * (64 / L);
                int grid = (int)((N + rpb - 1) / rpb); grid = grid > 256 ? 256 : grid;
                if (L == 8) vqkd_tail_small_kernel<8><<<grid, 1024, 0, s>>>(a->xn, a->w_out, a->idx, N, D, 1e-12f, a->z_ste, (double *)a->scratch16, a->mse, tail_partials);
                else if (L == 16) vqkd_tail_small_kernel<16><<<grid, 1024, 0, s>>>(a->xn, a->w_out, a->idx, N, D, 1e-12f, a->z_ste, (double *)a->scratch16, a->mse, tail_partials);
                else vqkd_tail_small_kernel<32><<<grid, 1024, 0, s>>>(a->xn, a->w_out, a->idx, N, D, 1e-12f, a->z_ste, (double *)a->scratch16, a->mse, tail_partials);
            } else {
                int grid = (int)((N + 15) / 16); grid = grid > 256 ? 256 : grid;
                vqkd_tail_kernel<<<grid, 1024, 0, s>>>(a->xn, a->w_out, a->idx, N, D, 1e-12f, a->z_ste, (double *)a->scratch16, a->mse, tail_partials);
            }
            VQ_CHECK_LAUNCH("vqkd_tail_kernel");
        }
    }
    return VQHIP_OK;
}

int vqhip_vq_forward(vqhip_vq_forward_t *a, void *stream) {
    if (!a || a->struct_bytes != (int64_t)sizeof(vqhip_vq_forward_t)) return fail(VQHIP_EINVAL, "vqhip_vq_forward: struct_bytes != sizeof(vqhip_vq_forward_t)");
    const int64_t N = a->N, K = a->K;
    const int D = a->D;
    if (N <= 0 || K <= 0 || D <= 0 || N >= (1ll << 31) || K >= (1ll << 31)) return fail(VQHIP_EINVAL, "vqhip_vq_forward: N, K, D");
    if (a->metric != VQHIP_METRIC_L2 && a->metric != VQHIP_METRIC_COS && a->metric != VQHIP_METRIC_COS_BF16) return fail(VQHIP_EINVAL, "vqhip_vq_forward: metric");
    if (a->x_dtype != VQHIP_DTYPE_F32 && a->x_dtype != VQHIP_DTYPE_BF16) return fail(VQHIP_EINVAL, "vqhip_vq_forward: x_dtype");
    if (!a->x || !a->w_in || !a->cb || !a->idx || !a->ws || (a->normalize && (!a->w_out || !a->xn))) return fail(VQHIP_EINVAL, "vqhip_vq_forward: null pointer");
    if (VQ_IS_COS(a->metric) && !a->xq) return fail(VQHIP_EINVAL, "vqhip_vq_forward: the cosine metric needs the xq buffer");
    if ((a->z_ste || a->mse) && (!a->mse || !a->scratch16)) return fail(VQHIP_EINVAL, "vqhip_vq_forward: the decode tail needs mse and scratch16");
    VQ_NEED("vqhip_vq_forward: ws too small", a->ws_bytes, vqhip_workspace_bytes(N, K, D));
    hipStream_t s = (hipStream_t)stream;
    const void *rows = a->x;
    int rows_dtype = a->x_dtype;
    const float *codes = a->w_in;
    if (a->normalize) {
        if (int rc = launch_vqkd_front(a->w_in, a->w_out, K, a->x, a->x_dtype, a->xn, N, D, 1e-12f, nullptr, 0, 1, s)) return rc;
        rows = a->xn; rows_dtype = VQHIP_DTYPE_F32; codes = a->w_out;
    }
    if (int rc = vqhip_encode_ex(rows, rows_dtype, codes, N, K, D, a->metric, a->cb, a->cb_bytes, a->idx, a->hist, a->xq, a->ws, a->ws_bytes,
                                 a->hist ? VQHIP_ENCODE_ZERO_HIST : 0, stream)) return rc;
    if (a->mse)
        if (int rc = vqhip_gather_ste_mse(rows, rows_dtype, codes, a->idx, N, D, nullptr, a->z_ste, a->mse, a->beta, a->scratch16, stream)) return rc;
    return VQHIP_OK;
}

int vqhip_vqkd_backward(const void *x, int x_dtype, const float *xn, const float *w, const int64_t *idx, int64_t N, int D,
                        const float *g_zste, const float *g_loss, float *grad_x, void *stream) {
    if (!x || !xn || !w || !idx || !grad_x || N < 0 || D <= 0) return fail(VQHIP_EINVAL, "vqhip_vqkd_backward: bad argument");
    if (N == 0) return VQHIP_OK;
    hipStream_t s = (hipStream_t)stream;
    if (x_dtype != VQHIP_DTYPE_F32 && x_dtype != VQHIP_DTYPE_BF16) return fail(VQHIP_EINVAL, "vqhip_vqkd_backward: x_dtype");
    if (D <= 32) {                                          // L lanes per token (vqhip_step_kernels.h)
        const int L = D <= 8 ? 8 : (D <= 16 ? 16 : 32), rpb = 4 * (64 / L);
        int grid = (int)((N + rpb - 1) / rpb); grid = grid > 2048 ? 2048 : grid;
#define VQ_BWD_SMALL(DT, LL) vqkd_backward_small_kernel<DT, LL><<<grid, 256, 0, s>>>(x, xn, w, idx, N, D, 1e-12f, g_zste, g_loss, grad_x)
        if (x_dtype == VQHIP_DTYPE_F32) { if (L == 8) VQ_BWD_SMALL(0, 8); else if (L == 16) VQ_BWD_SMALL(0, 16); else VQ_BWD_SMALL(0, 32); }
        else { if (L == 8) VQ_BWD_SMALL(1, 8); else if (L == 16) VQ_BWD_SMALL(1, 16); else VQ_BWD_SMALL(1, 32); }
#undef VQ_BWD_SMALL
        VQ_CHECK_LAUNCH("vqkd_backward_kernel");
        return VQHIP_OK;
    }
    int grid = (int)((N + 3) / 4); grid = grid > 2048 ? 2048 : grid;
    if (x_dtype == VQHIP_DTYPE_F32) vqkd_backward_kernel<0><<<grid, 256, 0, s>>>(x, xn, w, idx, N, D, 1e-12f, g_zste, g_loss, grad_x);
    else vqkd_backward_kernel<1><<<grid, 256, 0, s>>>(x, xn, w, idx, N, D, 1e-12f, g_zste, g_loss, grad_x);
    VQ_CHECK_LAUNCH("vqkd_backward_kernel");
    return VQHIP_OK;
}

int vqhip_argmin_stats(const void *ws, int32_t *out, void *stream) {
    if (!ws || !out) return fail(VQHIP_EINVAL, "vqhip_argmin_stats: bad argument");
    VQ_HIP(hipMemcpyAsync(out, ws, 16, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return VQHIP_OK;
}

int vqhip_diff(const void *a, int a_dtype, const void *b, int b_dtype, int64_t n, float scale, const float *scale_dev,
               float *out, double *sse, void *stream) {
    if (!a || !b || n < 0 || (!out && !sse)) return fail(VQHIP_EINVAL, "vqhip_diff: bad argument");
    if (n == 0) return VQHIP_OK;
    hipStream_t s = (hipStream_t)stream;
    int grid = (int)((n + 255) / 256); grid = grid > 2048 ? 2048 : grid;
    if (a_dtype == 0 && b_dtype == 0) diff_kernel<0, 0><<<grid, 256, 0, s>>>(a, b, n, scale, scale_dev, out, sse);
    else if (a_dtype == 0 && b_dtype == 1) diff_kernel<0, 1><<<grid, 256, 0, s>>>(a, b, n, scale, scale_dev, out, sse);
    else if (a_dtype == 1 && b_dtype == 0) diff_kernel<1, 0><<<grid, 256, 0, s>>>(a, b, n, scale, scale_dev, out, sse);
    else if (a_dtype == 1 && b_dtype == 1) diff_kernel<1, 1><<<grid, 256, 0, s>>>(a, b, n, scale, scale_dev, out, sse);
    else return fail(VQHIP_EINVAL, "vqhip_diff: dtype");
    VQ_CHECK_LAUNCH("diff_kernel");
    return VQHIP_OK;
}

int vqhip_vq_backward(const void *x, int x_dtype, const float *e, const int64_t *idx, int64_t N, int D, const float *g_zste,
                      const float *g_cb, const float *g_cm, float *grad_x, float *grad_w, void *stream) {
    return vqhip_vq_backward_ex(x, x_dtype, e, idx, N, D, g_zste, g_cb, g_cm, nullptr, 0.0f, grad_x, grad_w, stream);
}

int vqhip_vq_backward_ex(const void *x, int x_dtype, const float *e, const int64_t *idx, int64_t N, int D, const float *g_zste,
                         const float *g_cb, const float *g_cm, const float *g_comb, float beta, float *grad_x, float *grad_w,
                         void *stream) {
    if (!x || !e || !idx || N < 0 || D <= 0 || (!grad_x && !grad_w)) return fail(VQHIP_EINVAL, "vqhip_vq_backward: bad argument");
    if (N == 0) return VQHIP_OK;
    hipStream_t s = (hipStream_t)stream;
    int grid = (int)((N + 3) / 4); grid = grid > 2048 ? 2048 : grid;
    if (x_dtype == VQHIP_DTYPE_F32)
        vq_backward_kernel<0><<<grid, 256, 0, s>>>(x, e, idx, N, D, g_zste, g_cb, g_cm, grad_x, grad_w, g_comb, beta);
    else if (x_dtype == VQHIP_DTYPE_BF16)
        vq_backward_kernel<1><<<grid, 256, 0, s>>>(x, e, idx, N, D, g_zste, g_cb, g_cm, grad_x, grad_w, g_comb, beta);
    else return fail(VQHIP_EINVAL, "vqhip_vq_backward: x_dtype");
    VQ_CHECK_LAUNCH("vq_backward_kernel");
    return VQHIP_OK;
}

int vqhip_vq_backward_map(const void *x_rows, int x_dtype, const float *e, const int64_t *idx, int64_t B, int64_t HW, int D,
                          const float *g_map, const float *g_cm, const float *g_comb, float beta, void *grad_map, int grad_dtype,
                          void *stream) {
    if (!x_rows || !e || !idx || !grad_map || B <= 0 || HW <= 0 || D <= 0) return fail(VQHIP_EINVAL, "vqhip_vq_backward_map: bad argument");
    if ((HW % 256) != 0 || (D % 32) != 0) return fail(VQHIP_EINVAL, "vqhip_vq_backward_map: needs HW % 256 == 0 and D % 32 == 0 (transpose and use vqhip_vq_backward_ex)");
    if ((x_dtype != VQHIP_DTYPE_F32 && x_dtype != VQHIP_DTYPE_BF16) || (grad_dtype != VQHIP_DTYPE_F32 && grad_dtype != VQHIP_DTYPE_BF16))
        return fail(VQHIP_EINVAL, "vqhip_vq_backward_map: dtype");
    const int64_t N = B * HW, nt = N / 256;
    int csplit = 1;
    while (nt * csplit < 512 && (D / 32) % (csplit * 2) == 0) csplit *= 2;
    const int64_t items = nt * csplit;
    const int grid = (int)(items < 1024 ? items : 1024);
    constexpr int LDS = 2 * 32 * 256 * 4;
    static LdsCache sets[4];
    const int v = (x_dtype == VQHIP_DTYPE_BF16 ? 1 : 0) + (grad_dtype == VQHIP_DTYPE_BF16 ? 2 : 0);
    const void *kerns[4] = {(const void *)vq_backward_map256_kernel<0, 0>, (const void *)vq_backward_map256_kernel<1, 0>,
                            (const void *)vq_backward_map256_kernel<0, 1>, (const void *)vq_backward_map256_kernel<1, 1>};
    if (int rc = ensure_dyn_lds(kerns[v], LDS, sets[v])) return rc;
    hipStream_t s = (hipStream_t)stream;
#define VQ_BMAP(DT, ODT) vq_backward_map256_kernel<DT, ODT><<<grid, 512, LDS, s>>>(x_rows, e, idx, N, D, HW, csplit, g_map, g_cm, g_comb, beta, grad_map)
    if (v == 0) VQ_BMAP(0, 0); else if (v == 1) VQ_BMAP(1, 0); else if (v == 2) VQ_BMAP(0, 1); else VQ_BMAP(1, 1);
#undef VQ_BMAP
    VQ_CHECK_LAUNCH("vq_backward_map256_kernel");
    return VQHIP_OK;
}

// ---- deterministic (ordered) codebook-side sums -----------------------------------------------------------------
int64_t vqhip_order_workspace_bytes(int64_t N, int64_t K) {
    if (N < 0 || K <= 0) return 0;
    const int64_t nchunks = (N + VQ_SORT_CHUNK - 1) / VQ_SORT_CHUNK;
    return (nchunks > 0 ? nchunks : 1) * K * 4;
}

int vqhip_token_order(const int64_t *idx, int64_t N, int64_t K, int32_t *counts, int32_t *offsets, int32_t *order, void *ws,
                      int64_t ws_bytes, void *stream) {
    if (!idx || !counts || !offsets || !order || !ws || N < 0 || K <= 0) return fail(VQHIP_EINVAL, "vqhip_token_order: bad argument");
    VQ_NEED("vqhip_token_order: ws too small", ws_bytes, vqhip_order_workspace_bytes(N, K));
    if (K > 32768) return fail(VQHIP_EINVAL, "vqhip_token_order: K > 32768 (per-chunk histogram must fit in LDS)");
    if (N >= (1ll << 31)) return fail(VQHIP_EINVAL, "vqhip_token_order: N too large");
    const int64_t nchunks = (N + VQ_SORT_CHUNK - 1) / VQ_SORT_CHUNK;
    if (nchunks * K >= (1ll << 31)) return fail(VQHIP_EINVAL, "vqhip_token_order: N*K too large for the ordered route");
    hipStream_t s = (hipStream_t)stream;
    int *blockhist = (int *)ws;
    const size_t lds = (size_t)K * 4;
    static LdsCache lds_set;
    if (int rc = ensure_dyn_lds((const void *)sort_hist_kernel, lds, lds_set)) return rc;
    if (nchunks > 0) {
        sort_hist_kernel<<<(int)nchunks, VQ_SORT_CHUNK, lds, s>>>(idx, N, (int)K, blockhist);
        VQ_CHECK_LAUNCH("sort_hist_kernel");
    }
    sort_colscan_kernel<<<(int)((K + 255) / 256), 256, 0, s>>>(blockhist, (int)nchunks, (int)K, counts);
    VQ_CHECK_LAUNCH("sort_colscan_kernel");
    sort_offsets_kernel<<<1, 1024, 0, s>>>(counts, (int)K, offsets);
    VQ_CHECK_LAUNCH("sort_offsets_kernel");
    if (nchunks > 0) {
        sort_place_kernel<<<(int)nchunks, VQ_SORT_CHUNK, 0, s>>>(idx, N, (int)K, blockhist, offsets, order);
        VQ_CHECK_LAUNCH("sort_place_kernel");
    }
    return VQHIP_OK;
}

int64_t vqhip_segsum_workspace_bytes(int64_t N, int D) {
    if (N < 0 || D <= 0) return 0;
    return ((N + VQ_SEG_RANGE - 1) / VQ_SEG_RANGE + 1) * 2 * (int64_t)D * 4;
}

int vqhip_segsum_rows(const float *src, const int64_t *idx, const int32_t *order, const int32_t *offsets, int64_t N, int64_t K,
                      int D, float *dst, void *ws, int64_t ws_bytes, void *stream) {
    if (!src || !idx || !order || !offsets || !dst || !ws || N < 0 || K <= 0 || D <= 0 || (D % 4) != 0)
        return fail(VQHIP_EINVAL, "vqhip_segsum_rows: bad argument (D must be a multiple of 4)");
    VQ_NEED("vqhip_segsum_rows: ws too small", ws_bytes, vqhip_segsum_workspace_bytes(N, D));
    return run_segsum<0>(src, VQHIP_DTYPE_F32, nullptr, idx, order, offsets, N, K, D, nullptr, dst, ws, (hipStream_t)stream);
}

int vqhip_vq_backward_w_ordered(const void *x, int x_dtype, const float *e, const int64_t *idx, const int32_t *order,
                                const int32_t *offsets, int64_t N, int64_t K, int D, const float *g_cb, float *grad_w, void *ws,
                                int64_t ws_bytes, void *stream) {
    if (!x || !e || !idx || !order || !offsets || !grad_w || !ws || N < 0 || K <= 0 || D <= 0 || (D % 4) != 0)
        return fail(VQHIP_EINVAL, "vqhip_vq_backward_w_ordered: bad argument (D must be a multiple of 4)");
    VQ_NEED("vqhip_vq_backward_w_ordered: ws too small", ws_bytes, vqhip_segsum_workspace_bytes(N, D));
    if (x_dtype != VQHIP_DTYPE_F32 && x_dtype != VQHIP_DTYPE_BF16) return fail(VQHIP_EINVAL, "vqhip_vq_backward_w_ordered: x_dtype");
    return run_segsum<1>(x, x_dtype, e, idx, order, offsets, N, K, D, g_cb, grad_w, ws, (hipStream_t)stream);
}

int vqhip_ste(const void *x, int x_dtype, const float *z, int64_t n, float *out, void *stream) {
    if (!x || !z || !out || n < 0) return fail(VQHIP_EINVAL, "vqhip_ste: bad argument");
    if (n == 0) return VQHIP_OK;
    hipStream_t s = (hipStream_t)stream;
    int grid = (int)((n + 255) / 256); grid = grid > 4096 ? 4096 : grid;
    if (x_dtype == VQHIP_DTYPE_F32) ste_kernel<0><<<grid, 256, 0, s>>>(x, z, n, out);
    else if (x_dtype == VQHIP_DTYPE_BF16) ste_kernel<1><<<grid, 256, 0, s>>>(x, z, n, out);
    else return fail(VQHIP_EINVAL, "vqhip_ste: dtype");
    VQ_CHECK_LAUNCH("ste_kernel");
    return VQHIP_OK;
}

int vqhip_normalize_rows_bwd(const void *v, int dtype, const float *g, int64_t R, int D, float eps, float *gv, void *stream) {
    if (!v || !g || !gv || R < 0 || D <= 0) return fail(VQHIP_EINVAL, "vqhip_normalize_rows_bwd: bad argument");
    if (R == 0) return VQHIP_OK;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == VQHIP_DTYPE_F32) normalize_bwd_kernel<0><<<waves_grid(R, 4), 256, 0, s>>>(v, g, R, D, eps, gv);
    else if (dtype == VQHIP_DTYPE_BF16) normalize_bwd_kernel<1><<<waves_grid(R, 4), 256, 0, s>>>(v, g, R, D, eps, gv);
    else return fail(VQHIP_EINVAL, "vqhip_normalize_rows_bwd: dtype");
    VQ_CHECK_LAUNCH("normalize_bwd_kernel");
    return VQHIP_OK;
}

int vqhip_transpose(const void *in, void *out, int elem_bytes, int64_t B, int R, int C, void *stream) {
    if (!in || !out || B < 0 || R <= 0 || C <= 0 || (elem_bytes != 2 && elem_bytes != 4))
        return fail(VQHIP_EINVAL, "vqhip_transpose: bad argument");
    if (B == 0) return VQHIP_OK;
    if (B > 65535) return fail(VQHIP_EINVAL, "vqhip_transpose: batch too large for one launch");
    dim3 grid((C + 63) / 64, (R + 63) / 64, (unsigned)B);
    hipStream_t s = (hipStream_t)stream;
    if (elem_bytes == 4) transpose_kernel<uint32_t><<<grid, 256, 0, s>>>((const uint32_t *)in, (uint32_t *)out, B, R, C);
    else transpose_kernel<uint16_t><<<grid, 256, 0, s>>>((const uint16_t *)in, (uint16_t *)out, B, R, C);
    VQ_CHECK_LAUNCH("transpose_kernel");
    return VQHIP_OK;
}

int vqhip_codebook_metrics(const int64_t *counts, int64_t K, double *out, void *stream) {
    if (!counts || !out || K <= 0) return fail(VQHIP_EINVAL, "vqhip_codebook_metrics: bad argument");
    codebook_metrics_kernel<<<1, 1024, 0, (hipStream_t)stream>>>(counts, K, out);
    VQ_CHECK_LAUNCH("codebook_metrics_kernel");
    return VQHIP_OK;
}

int vqhip_debug_proposal_scores(const void *x, int x_dtype, const void *cb, int64_t N, int64_t K, int D, int metric,
                                float *scores, float *margin, float *scale, void *ws, int64_t ws_bytes, void *stream) {
    if (!x || !cb || !scores || !margin || !scale || !ws || N <= 0 || K <= 0 || !vq_coarse_supported(D))
        return fail(VQHIP_EINVAL, "vqhip_debug_proposal_scores: bad argument");
    VQ_NEED("vqhip_debug_proposal_scores: ws too small", ws_bytes, vqhip_workspace_bytes(N, K, D));
    hipStream_t s = (hipStream_t)stream;
    VqCbLayout L = vq_cb_layout(K, D);
    VqWsLayout W = vq_ws_layout(N, K, D);
    const char *c = (const char *)cb;
    char *w = (char *)ws;
    int *counters = (int *)(w + W.off_counters);
    float *xh2 = (float *)(w + W.off_xh2), *rho2 = (float *)(w + W.off_rho2);
    char *ximg = w + W.off_ximg;
    const int xgrid = (int)((N + 31) / 32);
    if (x_dtype == VQHIP_DTYPE_F32) x_prep_kernel<0><<<xgrid, 256, 0, s>>>(x, N, D, L.nstep, ximg, xh2, rho2, (float *)(w + W.off_xn), counters, (char *)cb, L);
    else if (x_dtype == VQHIP_DTYPE_BF16) x_prep_kernel<1><<<xgrid, 256, 0, s>>>(x, N, D, L.nstep, ximg, xh2, rho2, (float *)(w + W.off_xn), counters, (char *)cb, L);
    else return fail(VQHIP_EINVAL, "vqhip_debug_proposal_scores: x_dtype");
    VQ_CHECK_LAUNCH("x_prep_kernel");
    const char *frag = c + L.off_frag;
    switch (L.nstep) {
#define VQ_DBG(NS, TPS) case NS: debug_scores_kernel<NS, TPS><<<512, 256, 0, s>>>(ximg, frag, L.nstages, N, K, scores); break;
        VQ_DBG(2, VQ_TPS_D32) VQ_DBG(4, 4) VQ_DBG(8, 4) VQ_DBG(16, VQ_TPS16) VQ_DBG(32, 2) VQ_DBG(48, 1) VQ_DBG(64, 1)
#undef VQ_DBG
        default: return fail(VQHIP_EINVAL, "vqhip_debug_proposal_scores: unsupported padded D");
    }
    VQ_CHECK_LAUNCH("debug_scores_kernel");
    debug_margin_kernel<<<(int)((N + 255) / 256), 256, 0, s>>>(c, L, N, metric, xh2, rho2, margin, scale);
    VQ_CHECK_LAUNCH("debug_margin_kernel");
    return VQHIP_OK;
}

#ifdef VQ_CLOCK_STAMPS
// diagnostic builds only (not declared in include/vqhip.h): the (delta s_memtime, delta s_memrealtime) pairs the proposal kernel's
// waves stamped around their stage loop, copied to the HOST buffer out[2 * n], n <= VQ_CLOCK_SLOTS
int vqhip_debug_clock_stamps(unsigned long long *out_host, int n) {
    if (!out_host || n <= 0 || n > VQ_CLOCK_SLOTS) return fail(VQHIP_EINVAL, "vqhip_debug_clock_stamps: bad argument");
    VQ_HIP(hipDeviceSynchronize());
    VQ_HIP(hipMemcpyFromSymbol(out_host, HIP_SYMBOL(vq_clock_dbg), (size_t)n * 16, 0, hipMemcpyDeviceToHost));
    return VQHIP_OK;
}
#endif

#ifdef VQ_STAGE_STAMPS
// diagnostic builds only (tools/stage_stamps.py): the per-iteration stamps of the first workgroups of the last coarse_kernel launch
int vqhip_debug_stage_stamps(unsigned long long *out_host, int wgs) {
    if (!out_host || wgs <= 0 || wgs > VQ_STAGE_WGS) return fail(VQHIP_EINVAL, "vqhip_debug_stage_stamps: bad argument");
    VQ_HIP(hipDeviceSynchronize());
    VQ_HIP(hipMemcpyFromSymbol(out_host, HIP_SYMBOL(vq_stage_dbg), (size_t)wgs * 8 * VQ_STAGE_ITERS * 5 * 8, 0, hipMemcpyDeviceToHost));
    return VQHIP_OK;
}
#endif
#ifdef VQ_PHASE_STAMPS
// diagnostic builds only (tools/phase_stamps.py): the per-workgroup phase stamps of the last coarse_kernel launch
int vqhip_debug_phase_stamps(unsigned long long *out_host, int n) {
    if (!out_host || n <= 0 || n > VQ_PHASE_SLOTS) return fail(VQHIP_EINVAL, "vqhip_debug_phase_stamps: bad argument");
    VQ_HIP(hipDeviceSynchronize());
    VQ_HIP(hipMemcpyFromSymbol(out_host, HIP_SYMBOL(vq_phase_dbg), (size_t)n * 64, 0, hipMemcpyDeviceToHost));
    return VQHIP_OK;
}
#endif

int vqhip_set_tuning(int key, int value) {
    if (key == 2) g_tune_slices = (value == 1 || value == 2 || value == 4 || value == 8 || value == 16) ? value : 0;
    else if (key == 3) g_tune_gather_grid = value > 0 ? value : 0;
    else if (key == 4) g_tune_gather_nt = (value == 1 || value == 2) ? value : 0;
    else if (key == 5) g_tune_filter = value != 0;
    else if (key == 6) g_tune_fused_decide = (value >= 0 && value <= 2) ? value : 0;
    else if (key == 8) g_tune_noaux = value != 0;
    else if (key == 9) g_tune_groups = value != 0;
    else if (key == 10) g_tune_balance = value != 0;
    else if (key == 11) g_tune_w32 = value != 0;
    else if (key == 13) g_tune_map256 = value != 0;
    else if (key == 15) g_tune_col_direct = value != 0;
    else if (key == 17) g_tune_xdirect = (value == 0 || value == 1 || value == 2) ? value : 1;       // 2: fp32 rows too (measurement)
    else if (key == 18) g_tune_stream = value != 0;
    else if (key == 12) g_tune_force_exact = value > 0 ? (value < 1024 ? value : 1024) : 0;
    else return fail(VQHIP_EINVAL, "vqhip_set_tuning: unknown key");
    return VQHIP_OK;
}

// ---- the packed all-reduce on the caller's stream (RCCL resolved at run time; include/vqhip.h) -----------------------
// libvqhip never links librccl: a process that already holds one (PyTorch-ROCm's) must not get a second copy, and a
// caller without RCCL must still be able to load the library.  The handful of symbols are looked up once.
#include <dlfcn.h>
struct VqRcclId { char bytes[VQHIP_RCCL_ID_BYTES]; };      // ncclUniqueId (rccl.h: 128 opaque bytes), passed by value
namespace {
struct RcclApi {
    void *handle = nullptr;
    int (*get_unique_id)(void *) = nullptr;
    int (*comm_init_rank)(void **, int, VqRcclId, int) = nullptr;
    int (*comm_destroy)(void *) = nullptr;
    int (*all_reduce)(const void *, void *, size_t, int, int, void *, hipStream_t) = nullptr;
    const char *(*error_string)(int) = nullptr;
};
}  // namespace
static RcclApi g_rccl;
static std::mutex g_rccl_mu;
static const int kNcclFloat32 = 7, kNcclInt64 = 4, kNcclSum = 0, kNcclMin = 3;        // rccl.h: ncclFloat32 = 7, ncclInt64 = 4, ncclSum = 0, ncclMin = 3

static int rccl_fail(const char *what, int rc) {
    return fail(VQHIP_ERCCL, what, g_rccl.error_string ? g_rccl.error_string(rc) : "RCCL error");
}

int vqhip_rccl_load(const char *path) {
    std::lock_guard<std::mutex> lock(g_rccl_mu);
    if (g_rccl.handle) return VQHIP_OK;
    void *h = nullptr;
    if (path && path[0]) {
        h = dlopen(path, RTLD_NOW | RTLD_GLOBAL);
        if (!h) return fail(VQHIP_ERCCL, "vqhip_rccl_load: librccl.so not loadable", dlerror());
    } else {
        // without a path: ONLY the copy the process already has (RTLD_NOLOAD).  Falling back to the loader's search path could map
        // a second RCCL (say /opt/rocm's next to PyTorch's) into the process — two copies must never coexist; a caller whose
        // copy was loaded under another name passes its path
        const char *names[] = {"librccl.so", "librccl.so.1"};
        for (const char *n : names)
            if (!h) h = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
        if (!h) return fail(VQHIP_ERCCL, "vqhip_rccl_load: no librccl.so is mapped into this process under that name; pass the path of the copy the application uses");
    }
    RcclApi api;
    api.handle = h;
    api.get_unique_id = (decltype(api.get_unique_id))dlsym(h, "ncclGetUniqueId");
    api.comm_init_rank = (decltype(api.comm_init_rank))dlsym(h, "ncclCommInitRank");
    api.comm_destroy = (decltype(api.comm_destroy))dlsym(h, "ncclCommDestroy");
    api.all_reduce = (decltype(api.all_reduce))dlsym(h, "ncclAllReduce");
    api.error_string = (decltype(api.error_string))dlsym(h, "ncclGetErrorString");
    if (!api.get_unique_id || !api.comm_init_rank || !api.comm_destroy || !api.all_reduce || !api.error_string)
        return fail(VQHIP_ERCCL, "vqhip_rccl_load: librccl.so lacks an nccl* symbol");
    g_rccl = api;
    return VQHIP_OK;
}

int vqhip_rccl_unique_id(void *id_host) {
    if (!id_host) return fail(VQHIP_EINVAL, "vqhip_rccl_unique_id: null buffer");
    if (!g_rccl.handle) return fail(VQHIP_ERCCL, "vqhip_rccl_unique_id: call vqhip_rccl_load first");
    static_assert(sizeof(VqRcclId) == 128, "ncclUniqueId is 128 bytes (rccl.h: NCCL_UNIQUE_ID_BYTES)");
    const int rc = g_rccl.get_unique_id(id_host);
    return rc == 0 ? VQHIP_OK : rccl_fail("ncclGetUniqueId", rc);
}

int vqhip_rccl_comm_init(void **comm, int nranks, const void *id_host, int rank) {
    if (!comm || !id_host || nranks < 1 || rank < 0 || rank >= nranks) return fail(VQHIP_EINVAL, "vqhip_rccl_comm_init: bad argument");
    if (!g_rccl.handle) return fail(VQHIP_ERCCL, "vqhip_rccl_comm_init: call vqhip_rccl_load first");
    VqRcclId id;
    memcpy(id.bytes, id_host, sizeof(id.bytes));
    *comm = nullptr;
    const int rc = g_rccl.comm_init_rank(comm, nranks, id, rank);
    return rc == 0 ? VQHIP_OK : rccl_fail("ncclCommInitRank", rc);
}

int vqhip_rccl_comm_destroy(void *comm) {
    if (!comm) return VQHIP_OK;
    if (!g_rccl.handle) return fail(VQHIP_ERCCL, "vqhip_rccl_comm_destroy: RCCL not loaded");
    const int rc = g_rccl.comm_destroy(comm);
    return rc == 0 ? VQHIP_OK : rccl_fail("ncclCommDestroy", rc);
}

int vqhip_allreduce_packed(float *buf, int64_t floats, void *comm, void *stream) {
    if (!buf || !comm || floats < 0) return fail(VQHIP_EINVAL, "vqhip_allreduce_packed: bad argument");
    if (!g_rccl.handle) return fail(VQHIP_ERCCL, "vqhip_allreduce_packed: RCCL not loaded");
    if (floats == 0) return VQHIP_OK;
    const int rc = g_rccl.all_reduce(buf, buf, (size_t)floats, kNcclFloat32, kNcclSum, comm, (hipStream_t)stream);
    return rc == 0 ? VQHIP_OK : rccl_fail("ncclAllReduce", rc);
}

int vqhip_allreduce_min_i64(int64_t *buf, int64_t n, void *comm, void *stream) {
    if (!buf || !comm || n < 0) return fail(VQHIP_EINVAL, "vqhip_allreduce_min_i64: bad argument");
    if (!g_rccl.handle) return fail(VQHIP_ERCCL, "vqhip_allreduce_min_i64: RCCL not loaded");
    if (n == 0) return VQHIP_OK;
    const int rc = g_rccl.all_reduce(buf, buf, (size_t)n, kNcclInt64, kNcclMin, comm, (hipStream_t)stream);
    return rc == 0 ? VQHIP_OK : rccl_fail("ncclAllReduce", rc);
}

int vqhip_profile_enable(int on) {
    std::lock_guard<std::mutex> lock(g_prof_mu);
    g_prof_on = on != 0;
    g_prof_used = 0;
    return VQHIP_OK;
}

int vqhip_profile_collect(double *ms_sum, int64_t *launches) {
    if (!ms_sum || !launches) return fail(VQHIP_EINVAL, "vqhip_profile_collect: bad argument");
    std::lock_guard<std::mutex> lock(g_prof_mu);
    double total = 0.0;
    for (size_t i = 0; i < g_prof_used; ++i) {
        VQ_HIP(hipEventSynchronize(g_prof_events[i].second));
        float ms = 0.0f;
        VQ_HIP(hipEventElapsedTime(&ms, g_prof_events[i].first, g_prof_events[i].second));
        total += ms;
    }
    *ms_sum = total;
    *launches = (int64_t)g_prof_used;
    g_prof_used = 0;
    return VQHIP_OK;
}

}  // extern "C"


