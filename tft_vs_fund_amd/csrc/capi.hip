// C ABI of libtftfund.so (see include/tftfund.h).  gfx950 only; there is no
// CPU path behind these entry points: without a HIP device they fail loudly.
#include <hip/hip_runtime.h>
#include <mutex>
#include <string>
#include <type_traits>
#include <vector>
#include <cstdio>
#include <cstring>
#include "../../include/tftfund.h"
#include "launch.h"

namespace {

thread_local std::string g_err;

int fail(int code, const char* what) {
    g_err = what;
    return code;
}
int hip_fail(hipError_t e, const char* where) {
    g_err = std::string(where) + ": " + hipGetErrorString(e);
    return -(int)e;
}
#define TFF_LOCK(c) std::lock_guard<std::recursive_mutex> lk__((c)->mu)
#define TFF_HIP(call)                                    \
    do {                                                 \
        hipError_t e__ = (call);                         \
        if (e__ != hipSuccess) return hip_fail(e__, #call); \
    } while (0)

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    int reserve(size_t bytes) {
        if (bytes <= cap) return 0;
        if (p) (void)hipFree(p);
        p = nullptr; cap = 0;
        hipError_t e = hipMalloc(&p, bytes);
        if (e != hipSuccess) return hip_fail(e, "hipMalloc(workspace)");
        cap = bytes;
        return 0;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};

}  // namespace

struct tff_ctx {
    std::recursive_mutex mu;               // serialises the entry points of one context (its workspaces are shared state; _host calls nest)
    int device = 0;
    hipStream_t own = nullptr;
    hipStream_t stream = nullptr;
    hipEvent_t handover = nullptr;         // orders work across a change of stream (tff_ctx_set_stream)
    int solver = 0;
    int exact_below = tff::EXACT_BELOW_N;   // TFF_OPT_EXACT_BELOW
    int stage = -1;
    DevBuf in, calm, out, idx, scratch_status, gh_rec, gh_topt, gh_init, spill, pre_rec, retry;
    const int32_t* sample_idx = nullptr;   // set around a *_sampled_dev call
    int32_t sample_ns = 0;                 //   size of the scene the indices refer to
    double* init_p = nullptr; double* init_x = nullptr;   // set around tff_pi_pose_batch_debug_dev
    int kernel_variant = 0;                // TFF_OPT_KERNEL
    int gh_exact = 0;                      // TFF_OPT_GH_EXACT
    int spill_only_if_needed = 0;          // TFF_OPT_SPILL
    int rows = 2;                          // TFF_OPT_ROWS: 0 never, 1 always, 2 by batch size (rows_for)
    int retry_parity = 0;                  // which of the two retry counters this call uses (launch_pose_rows)
    int pre = 0;                           // TFF_OPT_PRE: 0 never (default: measured slower, see pre_for), 1 always, 2 from N >= 48
    int dbg_fp_handover = 0;               // TFF_OPT_DEBUG_FP_HANDOVER
    int dbg_adaptive = 0;                  // TFF_OPT_DEBUG_ADAPTIVE
    int count_rows = 1;                    // TFF_OPT_COUNT_ROWS: inlier counts four hypotheses per wavefront (default) or one
};

namespace {

// Four triplets per wavefront or one?  The row kernels issue ~2.5x fewer instructions per triplet, but a wavefront of theirs lives ~1.3x (N = 200)
// to 1.7x (N = 500) as long as a one-triplet wavefront, and a batch that fits the device's 2048 wavefront slots in one go pays that latency
// for nothing.  Measured (tools/ab_rows_sweep.py, ms per batch, rows / one-triplet): N = 200: B = 256 0.075 / 0.059, 1024 0.077 / 0.079,
// 3072 0.085 / 0.125; N = 500: B = 1024 0.125 / 0.103, 2048 0.138 / 0.118, 3072 0.140 / 0.178; LinearF alike.
// Until the end of round 5 the default (TFF_OPT_ROWS = 2) went by batch size for the two linear methods.  The two routes agree to 1e-14 but not bit for
// bit, so a triplet's last bits depended on the batch it arrived in -- a hazard the reference (one deterministic call per triplet) does not have, for
// ~16 microseconds of latency on calls whose launch + transfer overhead is ten times that.  The default now is the row kernels at ANY batch size, for
// every method: same triplet, same bits, in a batch of one, of 1 023 or of a million, sampled or not, sharded or not.  TFF_OPT_ROWS = 0 still forces the
// one-triplet kernels (lowest latency for batches under ~1 000 triplets), 1 is the same as the default.
bool rows_for(const tff_ctx* c, int64_t /*B*/, int32_t /*N*/) { return c->rows != 0; }

// The iterative methods (Gauss-Helmert on T / F / the Pi matrices) amplify a last-bit difference of their start, so for them the route must not
// depend on the batch size: whatever B, the linear stage and the pose tail run four triplets per wavefront unless TFF_OPT_ROWS = 0 forces the
// one-triplet kernels (a few tens of microseconds of latency on a millisecond iteration).  Same triplet, same bits, same `iter` in any batch.
bool rows_for_iterative(const tff_ctx* c) { return c->rows != 0; }
// The normalisations and moment sums of the trifocal row kernels as a kernel of their own (tft_moments_kernel.h: one triplet per wavefront,
// correspondences read from HBM once, three wavefronts per SIMD)?  Built and measured in round 5 (profiles/r5_ab_pre.txt, tools/ab_pre.py,
// 10 000 triplets): N = 200 one batch at a time 0.179 -> 0.173 ms, two batches in flight 0.1225 -> 0.1284 ms; slower at every other N
// (N = 100: 0.136 -> 0.139 / 0.092 -> 0.103; N = 500: 0.312 -> 0.322 / 0.204 -> 0.252).  The two passes it removes from the row kernel were
// 31 % of a wavefront's CYCLES but memory waits that the SIMD's other wavefront filled with its compute-bound middle: the path is bound by fp64
// issue, and the pre-kernel only moves ~900 instructions per triplet to a launch of its own.  Hence OFF by default (TFF_OPT_PRE = 1 enables it,
// 2 = from N >= 48); sampled hypotheses (config 4) never take it.
bool pre_for(const tff_ctx* c, int32_t N) {
    if (c->sample_idx) return false;
    if (c->pre != 2) return c->pre != 0;
    return N >= 48;
}
// launches k_tft_moments on the context's stream; *pre_out = the B x PRE_DOUBLES records the row kernels' <true> variants read
int launch_moments(tff_ctx* c, const double* corresp, int64_t B, int32_t N, const double** pre_out) {
    if (int r = c->pre_rec.reserve((size_t)B * tff::PRE_DOUBLES * sizeof(double))) return r;
    tff::MomentArgs m{corresp, (long)B, N, (double*)c->pre_rec.p};
    const bool stage = N <= tff::PRE_STAGE_MAX_N;
    const size_t lds = tff::moments_lds_bytes(N, stage);
    if (stage) hipLaunchKernelGGL(tff::k_tft_moments<true>, dim3(tff::moments_grid(B)), dim3(64), lds, c->stream, m);
    else hipLaunchKernelGGL(tff::k_tft_moments<false>, dim3(tff::moments_grid(B)), dim3(64), lds, c->stream, m);
    TFF_HIP(hipGetLastError());
    *pre_out = (const double*)c->pre_rec.p;
    return 0;
}

int base_flags(const tff_ctx* c, bool reconst) {
    return (reconst ? tff::FLAG_RECONST : 0) | (c->gh_exact ? tff::FLAG_GH_EXACT : 0) | (c->dbg_fp_handover ? tff::FLAG_DBG_FP_HANDOVER : 0) |
           (c->dbg_adaptive ? tff::FLAG_DBG_ADAPTIVE : 0);
}
int staged_flags(const tff_ctx* c, int N, int flags, bool jacobi, int max_n = tff::STAGE_MAX_N_TFT) {
    if (c->stage < 0) return tff::pose_auto_flags(N, flags, jacobi, max_n);
    if (c->stage > 0) return flags | tff::FLAG_STAGE_LDS;
    return flags;
}

template <class K>
int ensure_lds(K kernel, size_t bytes) {
    if (bytes > 160 * 1024) return fail(TFF_E_INVALID, "N too large for the 160 KiB LDS workspace of this method");
    if (bytes > 64 * 1024) TFF_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    return 0;
}

// Iterative methods at large N: when the per-correspondence state does not fit the 160 KB of LDS it goes to a global
// workspace, one slice per resident block (the kernels loop over the batch with a grid stride).  lds_full / lds_fixed: the
// kernel's LDS request with and without the per-correspondence part.  Returns the LDS bytes to launch with.
constexpr size_t LDS_LIMIT = 160 * 1024;
// grid of the Jacobi fix-up pass: it scans the status array for ST_RETRY (almost always none), so it is sized to be resident in one go
constexpr long FIXUP_GRID = 1024;
// occupancy_cap > 0 (the workgroup kernels; the cap is what their registers allow): spill also when that lets more workgroups share
// the CU's LDS -- their wave-serial steps (KKT solve, pseudo-inverse) make workgroups per CU what counts.  Measured
// (tools/bench_n_sweep.py): Ressl 2.14 -> 3.26 M/s at N = 500, Pi 1.18 -> 1.79 M/s at N = 300, never slower.
int plan_spill(tff_ctx* c, size_t lds_full, size_t lds_fixed, unsigned* grid, double** spill, long* stride, size_t* lds, int occupancy_cap = 0) {
    *spill = nullptr; *stride = 0; *lds = lds_full;
    auto per_cu = [&](size_t bytes) { const size_t k = LDS_LIMIT / (bytes + 512); return (int)(k < (size_t)occupancy_cap ? k : (size_t)occupancy_cap); };
    const bool for_occupancy = !c->spill_only_if_needed && occupancy_cap > 0 && lds_fixed < lds_full && per_cu(lds_fixed) > per_cu(lds_full);
    if (lds_full <= LDS_LIMIT && !for_occupancy) return 0;
    if (lds_fixed > LDS_LIMIT) return fail(TFF_E_INVALID, "LDS workspace of this method exceeds 160 KiB");
    // + 16 doubles: the kernels carve their per-correspondence arrays with small alignment pads (e.g. OptimF's v = xi + 4N + 2), so a
    // slice of exactly lds_full - lds_fixed bytes would let the tail of one block's arrays overlap the head of its neighbour's
    const size_t per_block = lds_full - lds_fixed + 16 * sizeof(double);
    size_t blocks = ((size_t)512 << 20) / per_block;
    if (blocks < 256) blocks = 256;
    if (*grid > blocks) *grid = (unsigned)blocks;
    if (int r = c->spill.reserve((size_t)*grid * per_block)) return r;
    *spill = (double*)c->spill.p; *stride = (long)(per_block / sizeof(double)); *lds = lds_fixed;
    return 0;
}

int check_common(const tff_ctx* c, const void* corresp, const void* calm, int64_t calm_stride, int64_t B, int32_t N) {
    if (!c) return fail(TFF_E_INVALID, "null context");
    if (B < 0 || N < 0) return fail(TFF_E_INVALID, "negative batch or correspondence count");
    if (B > 0 && (!corresp || !calm)) return fail(TFF_E_INVALID, "null input pointer");
    if (calm_stride != 0 && calm_stride != 27) return fail(TFF_E_INVALID, "calm_stride must be 0 (shared CalM) or 27");
    return 0;
}

// Two launches on the context's stream: the inverse-iteration kernel for the whole
// batch, then the Jacobi kernel over the (rare) triplets it marked ST_RETRY.
// stage_max_n: largest N whose correspondences are staged in LDS (0: the kernel never stages); occupancy_cap: wavefronts per CU the
// kernel's registers allow (0: the kernel has no per-correspondence LDS state to spill), see plan_spill.
// With TFF_OPT_SOLVER = 1 only the Jacobi kernel runs, for every triplet.
typedef size_t (*lds_fn)(int N, int flags, bool jacobi);

template <class KMain, class KJac>
int launch_pose(tff_ctx* c, KMain kmain, KJac kjac, lds_fn ldsfn, int stage_max_n, int occupancy_cap, const double* corresp, const double* calm, int64_t calm_stride, int64_t B,
                int32_t N, double* Rt2, double* Rt3, double* T, double* reconst, int32_t* iter, int32_t* status, double* dbg, bool main_done = false) {
    if (int r = check_common(c, corresp, calm, calm_stride, B, N)) return r;
    TFF_LOCK(c);
    if (B == 0) return 0;
    if (!Rt2 || !Rt3 || !T) return fail(TFF_E_INVALID, "null output pointer");
    TFF_HIP(hipSetDevice(c->device));
    if (!status) {                         // the kernels hand ST_RETRY over through the status array
        if (int r = c->scratch_status.reserve((size_t)B * sizeof(int32_t))) return r;
        status = (int32_t*)c->scratch_status.p;
    }
    tff::LinearTftArgs a{corresp, calm, (long)calm_stride, (long)B, N, base_flags(c, reconst != nullptr),
                         Rt2, Rt3, T, reconst, iter, status, dbg, c->sample_idx, c->init_p, c->init_x, nullptr, 0, c->sample_ns};
    if (c->sample_idx) {                   // gathered samples always live in LDS
        if (!stage_max_n) return fail(TFF_E_INVALID, "sampled hypotheses are not supported by this method");
        a.flags |= tff::FLAG_STAGE_LDS;
        stage_max_n = 0;
    }
    const bool all_exact = !main_done && (c->solver != 0 || N < c->exact_below);
    if (main_done) a.flags |= tff::FLAG_ONLY_RETRY;                          // (the caller has run the fast stages: launch_optim_f)
    if (!all_exact && !main_done) {
        tff::LinearTftArgs m = a;
        m.flags = stage_max_n ? staged_flags(c, N, a.flags, false, stage_max_n) : a.flags;
        unsigned grid = tff::pose_grid(B);
        size_t lds;
        if ((m.flags & tff::FLAG_STAGE_LDS) && ldsfn(N, m.flags, false) > LDS_LIMIT) {   // staged correspondences would not fit the LDS
            if (c->sample_idx) return fail(TFF_E_INVALID, "sample too large for the LDS (sampled hypotheses are gathered into LDS)");
            m.flags &= ~tff::FLAG_STAGE_LDS;                                              // re-read them through L2 instead
        }
        if (int r = plan_spill(c, ldsfn(N, m.flags, false), ldsfn(0, m.flags, false), &grid, &m.spill, &m.spill_stride, &lds, occupancy_cap)) return r;
        if (int r = ensure_lds(kmain, lds)) return r;
        hipLaunchKernelGGL(kmain, dim3(grid), dim3(64), lds, c->stream, m);
        TFF_HIP(hipGetLastError());
        a.flags |= tff::FLAG_ONLY_RETRY;
    }
    if (stage_max_n) a.flags = staged_flags(c, N, a.flags, true, stage_max_n);
    unsigned grid = !all_exact ? (unsigned)(B < FIXUP_GRID ? B : FIXUP_GRID) : tff::pose_grid(B);
    size_t lds;
    if ((a.flags & tff::FLAG_STAGE_LDS) && ldsfn(N, a.flags, true) > LDS_LIMIT) {
        if (c->sample_idx) return fail(TFF_E_INVALID, "sample too large for the LDS (sampled hypotheses are gathered into LDS)");
        a.flags &= ~tff::FLAG_STAGE_LDS;
    }
    if (int r = plan_spill(c, ldsfn(N, a.flags, true), ldsfn(0, a.flags, true), &grid, &a.spill, &a.spill_stride, &lds)) return r;
    if (int r = ensure_lds(kjac, lds)) return r;
    hipLaunchKernelGGL(kjac, dim3(grid), dim3(64), lds, c->stream, a);
    TFF_HIP(hipGetLastError());
    return 0;
}

// LinearTFTPoseEstimation / LinearFPoseEstimation, default route: four triplets per wavefront (tft_rows_kernel.h / f_rows_kernel.h, fast
// tiers), then the exact kernel (one wavefront per triplet) over what they could not finish or certify.
// krows_pre (may be null): the variant of krows that starts from k_tft_moments' records; taken when pre_for() says so.
template <class KRows, class KExact>
int launch_pose_rows(tff_ctx* c, KRows krows, KExact kexact, lds_fn exact_lds, int stage_max_n, const double* corresp, const double* calm, int64_t calm_stride,
                     int64_t B, int32_t N, double* Rt2, double* Rt3, double* T, double* reconst, int32_t* iter, int32_t* status, double* dbg,
                     KRows krows_pre = nullptr) {
    if (int r = check_common(c, corresp, calm, calm_stride, B, N)) return r;
    TFF_LOCK(c);
    if (B == 0) return 0;
    if (!Rt2 || !Rt3 || !T) return fail(TFF_E_INVALID, "null output pointer");
    TFF_HIP(hipSetDevice(c->device));
    if (!status) {                         // the kernels hand ST_RETRY over through the status array
        if (int r = c->scratch_status.reserve((size_t)B * sizeof(int32_t))) return r;
        status = (int32_t*)c->scratch_status.p;
    }
    tff::LinearTftArgs a{corresp, calm, (long)calm_stride, (long)B, N, base_flags(c, reconst != nullptr),
                         Rt2, Rt3, T, reconst, iter, status, dbg, c->sample_idx, c->init_p, c->init_x, nullptr, 0, c->sample_ns};
    // the triplets the row kernel flags go to the exact kernel as a compact list: [count 0 | count 1 | B indices].  The row kernel appends to the
    // list itself (one atomic per flagged triplet) and zeroes the OTHER counter for the context's next call; this call's counter was zeroed
    // during the previous call (both at allocation).  No scan of the status array, no launch in between.
    {
        void* before = c->retry.p;
        if (int r = c->retry.reserve(((size_t)B + 2) * sizeof(int32_t))) return r;
        if (c->retry.p != before) { TFF_HIP(hipMemsetAsync(c->retry.p, 0, 2 * sizeof(int32_t), c->stream)); c->retry_parity = 0; }
    }
    a.retry_count = (int*)c->retry.p + c->retry_parity;
    a.retry_zero = (int*)c->retry.p + (1 - c->retry_parity);
    a.retry_list = (B < (1L << tff::RETRY_HINT_SHIFT)) ? (int*)c->retry.p + 2 : nullptr;   // (an entry is index | hints << 28; beyond, the exact kernel scans the status array as before)
    if (krows_pre && N >= 7 && pre_for(c, N)) {
        if (int r = launch_moments(c, corresp, B, N, &a.pre)) return r;
        hipLaunchKernelGGL(krows_pre, dim3(tff::rows_grid(B)), dim3(64), tff::rows_lds_bytes(), c->stream, a);
        a.pre = nullptr;
    } else {
        hipLaunchKernelGGL(krows, dim3(tff::rows_grid(B)), dim3(64), tff::rows_lds_bytes(), c->stream, a);
    }
    TFF_HIP(hipGetLastError());
    c->retry_parity ^= 1;                  // (only now: the row kernel that zeroes the next call's counter is on the stream)
    a.flags |= tff::FLAG_ONLY_RETRY;
    if (c->sample_idx) a.flags |= tff::FLAG_STAGE_LDS;   // the exact kernel gathers samples into LDS
    else a.flags = staged_flags(c, N, a.flags, true, stage_max_n);
    if ((a.flags & tff::FLAG_STAGE_LDS) && exact_lds(N, a.flags, true) > LDS_LIMIT) {
        if (c->sample_idx) return fail(TFF_E_INVALID, "sample too large for the LDS (sampled hypotheses are gathered into LDS)");
        a.flags &= ~tff::FLAG_STAGE_LDS;
    }
    const size_t lds = exact_lds(N, a.flags, true);
    if (int r = ensure_lds(kexact, lds)) return r;
    // the fix-up: one resident round of wavefronts walks the list the row kernel has filled (almost always empty; ~0.3 % of a million seven-point
    // samples of an outlier-ridden scene, config 4) -- one triplet per wavefront and round, where until round 5 every block scanned a fixed share
    // of the status array and redid what it found there one after the other (5 of 27 ms in config 4)
    const long fix = 2 * FIXUP_GRID;
    hipLaunchKernelGGL(kexact, dim3((unsigned)(B < fix ? B : fix)), dim3(64), lds, c->stream, a);
    TFF_HIP(hipGetLastError());
    return 0;
}

// LinearTFTPoseEstimation: one wavefront per triplet (fast tiers) + the exact kernel over what they could not finish.
int launch_linear_tft(tff_ctx* c, const double* corresp, const double* calm, int64_t calm_stride, int64_t B, int32_t N,
                      double* Rt2, double* Rt3, double* T, double* reconst, int32_t* iter, int32_t* status, double* dbg) {
    const bool rows = c && rows_for(c, B, N);
    if (rows && c->solver == 0 && N >= c->exact_below)
        return launch_pose_rows(c, tff::k_linear_tft_pose_rows<false>, tff::k_linear_tft_pose<true>, tff::pose_lds_bytes, tff::STAGE_MAX_N_TFT, corresp, calm, calm_stride,
                                B, N, Rt2, Rt3, T, reconst, iter, status, dbg, tff::k_linear_tft_pose_rows<true>);
    if (rows)             // whole batches for the exact tiers (minimal samples, TFF_OPT_SOLVER = 1): four triplets per wavefront there too (tft_rows_exact_kernel.h)
        return launch_pose_rows(c, tff::k_linear_tft_pose_rows_exact, tff::k_linear_tft_pose<true>, tff::pose_lds_bytes, tff::STAGE_MAX_N_TFT, corresp, calm,
                                calm_stride, B, N, Rt2, Rt3, T, reconst, iter, status, dbg);
    return launch_pose(c, tff::k_linear_tft_pose<false>, tff::k_linear_tft_pose<true>, tff::pose_lds_bytes, tff::STAGE_MAX_N_TFT, 0, corresp, calm, calm_stride,
                       B, N, Rt2, Rt3, T, reconst, iter, status, dbg);
}
int launch_linear_f(tff_ctx* c, const double* corresp, const double* calm, int64_t calm_stride, int64_t B, int32_t N,
                    double* Rt2, double* Rt3, double* T, double* reconst, int32_t* iter, int32_t* status, double* dbg) {
    if (c && rows_for(c, B, N))                                               // four triplets per wavefront (f_rows_kernel.h): fast tiers, or -- whole batches for the exact tiers -- the exact ones
        return (c->solver == 0 && N >= c->exact_below)
            ? launch_pose_rows(c, tff::k_linear_f_pose_rows, tff::k_f_pose<true, 0>, tff::f_pose_lds_bytes, tff::STAGE_MAX_N_F, corresp, calm, calm_stride,
                               B, N, Rt2, Rt3, T, reconst, iter, status, dbg)
            : launch_pose_rows(c, tff::k_linear_f_pose_rows_exact, tff::k_f_pose<true, 0>, tff::f_pose_lds_bytes, tff::STAGE_MAX_N_F, corresp, calm, calm_stride,
                               B, N, Rt2, Rt3, T, reconst, iter, status, dbg);
    return launch_pose(c, tff::k_f_pose<false, 0>, tff::k_f_pose<true, 0>, tff::f_pose_lds_bytes, tff::STAGE_MAX_N_F, 0, corresp, calm, calm_stride, B, N, Rt2, Rt3, T,
                       reconst, iter, status, dbg);
}
// OptimFPoseEstimation.  Large batches: three stages (optimf_rows_kernel.h) -- linear stage and pose tail four triplets per wavefront, the
// Gauss-Helmert refinement one wavefront per triplet -- then the exact kernel over what they could not finish.  Small batches, minimal
// samples, TFF_OPT_SOLVER = 1, debug records: the fused one-triplet kernel.
int launch_optim_f(tff_ctx* c, const double* corresp, const double* calm, int64_t calm_stride, int64_t B, int32_t N,
                   double* Rt2, double* Rt3, double* T, double* reconst, int32_t* iter, int32_t* status, double* dbg) {
    if (c && B > 0 && rows_for_iterative(c) && c->solver == 0 && N >= c->exact_below && N >= 8 && !dbg && !c->sample_idx && c->kernel_variant != 1) {
        if (int r = check_common(c, corresp, calm, calm_stride, B, N)) return r;
        TFF_LOCK(c);
        if (!Rt2 || !Rt3 || !T) return fail(TFF_E_INVALID, "null output pointer");
        TFF_HIP(hipSetDevice(c->device));
        if (!status) {
            if (int r = c->scratch_status.reserve((size_t)B * sizeof(int32_t))) return r;
            status = (int32_t*)c->scratch_status.p;
        }
        if (int r = c->gh_rec.reserve((size_t)B * tff::OPTIMF_REC_DOUBLES * sizeof(double))) return r;
        tff::OptimFStageArgs sa{};
        sa.la = tff::LinearTftArgs{corresp, calm, (long)calm_stride, (long)B, N, base_flags(c, reconst != nullptr),
                                   Rt2, Rt3, T, reconst, iter, status, nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0};
        sa.rec = (double*)c->gh_rec.p;
        hipLaunchKernelGGL(tff::k_optimf_linear_rows, dim3(tff::rows_grid(B)), dim3(64), tff::rows_lds_bytes(), c->stream, sa);
        TFF_HIP(hipGetLastError());
        {
            tff::OptimFStageArgs m = sa;
            unsigned grid = tff::pose_grid(B);
            size_t lds;
            // the normalised observations go to LDS with xi while eight wavefronts still fit a CU (N <= ~220); beyond, the passes read the correspondences
            // through L2 as the fused kernel does, and xi follows plan_spill's occupancy rule
            const bool stage_x = tff::optimf_refine_lds_bytes(N, true) + 512 <= LDS_LIMIT / (4 * tff::OPTIMF_REFINE_WAVES);
            if (stage_x) {
                lds = tff::optimf_refine_lds_bytes(N, true);
                if (int r = ensure_lds(tff::k_optimf_refine<tff::OPTIMF_REFINE_WAVES, true>, lds)) return r;
                hipLaunchKernelGGL((tff::k_optimf_refine<tff::OPTIMF_REFINE_WAVES, true>), dim3(grid), dim3(64), lds, c->stream, m);
            } else {
                if (int r = plan_spill(c, tff::optimf_refine_lds_bytes(N, false), tff::optimf_refine_lds_bytes(0, false), &grid, &m.spill, &m.spill_stride, &lds, 4 * tff::OPTIMF_REFINE_WAVES)) return r;
                if (int r = ensure_lds(tff::k_optimf_refine<tff::OPTIMF_REFINE_WAVES, false>, lds)) return r;
                hipLaunchKernelGGL((tff::k_optimf_refine<tff::OPTIMF_REFINE_WAVES, false>), dim3(grid), dim3(64), lds, c->stream, m);
            }
            TFF_HIP(hipGetLastError());
        }
        hipLaunchKernelGGL(tff::k_optimf_finish_rows, dim3(tff::rows_grid(B)), dim3(64), tff::rows_lds_bytes(), c->stream, sa);
        TFF_HIP(hipGetLastError());
        return launch_pose(c, tff::k_f_pose<false, 1>, tff::k_f_pose<true, 1>, tff::optimf_lds_bytes, 0, 12, corresp, calm, calm_stride, B, N, Rt2, Rt3, T,
                           reconst, iter, status, dbg, true);
    }
    return launch_pose(c, tff::k_f_pose<false, 1>, tff::k_f_pose<true, 1>, tff::optimf_lds_bytes, 0, 12, corresp, calm, calm_stride, B, N, Rt2, Rt3, T,
                       reconst, iter, status, dbg);
}

// Iterative TFT methods: three launches, a workgroup of four wavefronts per triplet for the iteration (gh_wg_kernel.h,
// pi_wg_kernel.h): k_gh_linear (+ Jacobi fix-up), the block kernel, k_gh_finish.  wg_lds(n): LDS bytes of the block kernel for n
// correspondences held in LDS.
template <class KBlock, class LdsFn>
int launch_wg(tff_ctx* c, KBlock kblock, LdsFn wg_lds, int occupancy_cap, const double* corresp, const double* calm, int64_t calm_stride, int64_t B, int32_t N,
              double* Rt2, double* Rt3, double* T, double* reconst, int32_t* iter, int32_t* status, double* dbg, bool fp_first = false,
              bool rows_linear = true, bool nordberg_pre = false, int block_threads = tff::GH_WG_THREADS, size_t xi_bytes_per_n = 0, bool rows_finish = true) {
    if (int r = check_common(c, corresp, calm, calm_stride, B, N)) return r;
    TFF_LOCK(c);
    if (B == 0) return 0;
    if (!Rt2 || !Rt3 || !T) return fail(TFF_E_INVALID, "null output pointer");
    TFF_HIP(hipSetDevice(c->device));
    if (!status) {
        if (int r = c->scratch_status.reserve((size_t)B * sizeof(int32_t))) return r;
        status = (int32_t*)c->scratch_status.p;
    }
    if (int r = c->gh_rec.reserve((size_t)B * tff::GH_REC_DOUBLES * sizeof(double))) return r;
    if (int r = c->gh_topt.reserve((size_t)B * 27 * sizeof(double))) return r;
    tff::GhWgArgs a{corresp, calm, (long)calm_stride, (long)B, N, base_flags(c, reconst != nullptr), (double*)c->gh_rec.p, (double*)c->gh_topt.p,
                    Rt2, Rt3, T, reconst, iter, status, dbg, nullptr, 0};
    {   // linear stage: fast tiers, then the exact kernel over the triplets they marked ST_RETRY (minimal samples: exact kernel for all)
        const bool all_exact = c->solver != 0 || N < c->exact_below;
        tff::GhWgArgs m = a;
        size_t lds;
        if (!all_exact && rows_linear && rows_for_iterative(c)) {                      // four triplets per wavefront (gh_rows_kernel.h), whatever the batch size
            if (N >= 7 && pre_for(c, N)) {                                             // normalisations + moment sums in their own kernel (tft_moments_kernel.h)
                if (int r = launch_moments(c, corresp, B, N, &m.pre)) return r;
                hipLaunchKernelGGL(tff::k_gh_linear_rows<true>, dim3(tff::rows_grid(B)), dim3(64), tff::rows_lds_bytes(), c->stream, m);
                m.pre = nullptr;
            } else {
                hipLaunchKernelGGL(tff::k_gh_linear_rows<false>, dim3(tff::rows_grid(B)), dim3(64), tff::rows_lds_bytes(), c->stream, m);
            }
            TFF_HIP(hipGetLastError());
        } else if (!all_exact) {
            m.flags = staged_flags(c, N, a.flags, false);
            lds = tff::pose_lds_bytes(N, m.flags, false);
            if (int r = ensure_lds(tff::k_gh_linear<false>, lds)) return r;
            hipLaunchKernelGGL(tff::k_gh_linear<false>, dim3(tff::pose_grid(B)), dim3(64), lds, c->stream, m);
            TFF_HIP(hipGetLastError());
        }
        m.flags = staged_flags(c, N, a.flags, true) | (all_exact ? 0 : tff::FLAG_ONLY_RETRY);
        lds = tff::pose_lds_bytes(N, m.flags, true);
        if (int r = ensure_lds(tff::k_gh_linear<true>, lds)) return r;
        hipLaunchKernelGGL(tff::k_gh_linear<true>, dim3(all_exact ? tff::pose_grid(B) : (unsigned)(B < FIXUP_GRID ? B : FIXUP_GRID)), dim3(64), lds, c->stream, m);
        TFF_HIP(hipGetLastError());
    }
    if (nordberg_pre) {   // the serial part of Nordberg's initial parameters, one triplet per lane (gh_wg_kernel.h::k_nordberg_init)
        if (int r = c->gh_init.reserve((size_t)B * tff::NordbergModel::PRE_DOUBLES * sizeof(double))) return r;
        a.init_rec = (double*)c->gh_init.p;
        hipLaunchKernelGGL(tff::k_nordberg_init, dim3((unsigned)((B + 63) / 64)), dim3(64), 0, c->stream, a);
        TFF_HIP(hipGetLastError());
    }
    if (fp_first) {   // FaugPapa's own block kernel (gh_fp_kernel.h); the generic one below then redoes what it handed back (ST_RETRY: almost always nothing)
        tff::GhWgArgs m = a;
        unsigned grid = tff::pose_grid(B);
        size_t lds;
        if (int r = plan_spill(c, tff::fp_lds_bytes(N), tff::fp_lds_bytes(0), &grid, &m.spill, &m.spill_stride, &lds, tff::FP_WG_PER_CU)) return r;
        if (m.spill) {
            if (int r = ensure_lds(tff::k_fp_block<false>, lds)) return r;
            hipLaunchKernelGGL(tff::k_fp_block<false>, dim3(grid), dim3(tff::FP_THREADS), lds, c->stream, m);
        } else {
            if (int r = ensure_lds(tff::k_fp_block<true>, lds)) return r;
            hipLaunchKernelGGL(tff::k_fp_block<true>, dim3(grid), dim3(tff::FP_THREADS), lds, c->stream, m);
        }
        TFF_HIP(hipGetLastError());
    }
    {
        tff::GhWgArgs m = a;
        unsigned grid = fp_first ? (unsigned)(B < FIXUP_GRID ? B : FIXUP_GRID) : tff::pose_grid(B);
        if (fp_first) m.flags |= tff::FLAG_ONLY_RETRY;
        size_t lds;
        if (int r = plan_spill(c, wg_lds(N), wg_lds(0), &grid, &m.spill, &m.spill_stride, &lds, occupancy_cap)) return r;
        if (m.spill && xi_bytes_per_n) {                                     // the state went to global slices for occupancy: does xi alone still fit in LDS?
            const size_t partial = wg_lds(0) + xi_bytes_per_n * (size_t)N;
            if (partial <= LDS_LIMIT && LDS_LIMIT / (partial + 512) >= (size_t)occupancy_cap) { lds = partial; m.flags |= tff::FLAG_XI_IN_LDS; }
        }
        if (int r = ensure_lds(kblock, lds)) return r;
        hipLaunchKernelGGL(kblock, dim3(grid), dim3(block_threads), lds, c->stream, m);
        TFF_HIP(hipGetLastError());
    }
    if (rows_finish && N >= 12 && rows_for_iterative(c)) {                             // four triplets per wavefront (gh_rows_kernel.h); minimal samples: the one-triplet kernel's ladder
        hipLaunchKernelGGL(tff::k_gh_finish_rows, dim3(tff::rows_grid(B)), dim3(64), tff::rows_lds_bytes(), c->stream, a);
        TFF_HIP(hipGetLastError());
    } else {
        const size_t lds = tff::pose_lds_bytes(N, 0, false);
        hipLaunchKernelGGL(tff::k_gh_finish, dim3(tff::pose_grid(B)), dim3(64), lds, c->stream, a);
        TFF_HIP(hipGetLastError());
    }
    return 0;
}
// TFF_OPT_KERNEL = 1 selects the fused single-wavefront kernels (gh_kernel.h; for the Pi methods also TFF_OPT_SOLVER = 1, pi_kernel.h).
template <class Model, class KFused, class KFusedJac>
int launch_gh(tff_ctx* c, KFused kfused, KFusedJac kfused_jac, const double* corresp, const double* calm, int64_t calm_stride, int64_t B, int32_t N,
              double* Rt2, double* Rt3, double* T, double* reconst, int32_t* iter, int32_t* status, double* dbg) {
    // Both kernels evaluate the weights in the factored form that reproduces the 50-digit iteration (tests/test_gpu_gh_noise.py runs
    // each of them on every fixture).  Until round 4 the fused single-wavefront kernel won below N = 80 (Ressl) / 72 (Nordberg): a workgroup of 256
    // threads idled on a few correspondences.  With TWO wavefronts per workgroup and four workgroups per CU (gh_wg_kernel.h::gh_wg_waves) the
    // workgroup path wins at every N -- tools/ab_wg_fused.py, 10 k triplets, workgroup / fused: Ressl 1.62 / 2.33 ms at N = 12, 1.58 / 2.34 at 60,
    // 1.77 / 2.85 at 100; Nordberg 2.00 / 2.96, 1.94 / 2.92, 2.10 / 3.74 -- so the fused kernels remain as TFF_OPT_KERNEL = 1 only.
    const bool small = false;
    if (c->kernel_variant == 1 || small)                                     // TFF_OPT_SOLVER = 1 is honoured by launch_wg's linear stage
        return launch_pose(c, kfused, kfused_jac, tff::gh_lds_bytes<Model>, 0, std::is_same<Model, tff::ResslModel>::value ? 8 : 4, corresp, calm, calm_stride, B, N, Rt2, Rt3, T, reconst, iter, status, dbg);
    auto wg_lds = [](int n) { return (size_t)(((tff::POSE_LDS_DOUBLES + 1) & ~1) + tff::gh_wg_lds_doubles(Model::U, Model::C, n, Model::REDUNDANT_CONSTRAINTS)) * sizeof(double); };
    // FaugPapa: the factored iteration of gh_fp_kernel.h (the threads stride over the correspondences: any N) unless an A/B switch asks for the generic kernel
    const bool fp_first = std::is_same<Model, tff::FaugPapaModel>::value && c->kernel_variant == 0 && !c->gh_exact;
    // occupancy policy of the per-correspondence state (plan_spill): Nordberg runs as fast with it in LDS at two workgroups per CU as with it in global
    // slices at three (3.89 vs 3.87 ms per 10 k x 200) -- without the state's HBM round trips (what is left of its 52x algorithmic traffic is scratch:
    // the 168-register build spills 368 registers, and is still faster than the 256-register one, 3.69 vs 3.90 ms)
    // (round 4: Ressl and Nordberg run two wavefronts per workgroup, four workgroups per CU at 256 registers -- gh_wg_kernel.h::gh_wg_waves)
    const int occupancy_cap = tff::gh_wg_per_cu<Model>::value;
    return launch_wg(c, tff::k_gh_block<Model>, wg_lds, occupancy_cap, corresp, calm, calm_stride, B, N, Rt2, Rt3, T, reconst, iter, status, dbg, fp_first, true,
                     std::is_same<Model, tff::NordbergModel>::value, tff::gh_wg_waves<Model>::value * tff::WAVE, tff::GH_XI * sizeof(double));
}
template <class Model>
int launch_pi_model(tff_ctx* c, const double* corresp, const double* calm, int64_t calm_stride, int64_t B, int32_t N,
                    double* Rt2, double* Rt3, double* T, double* reconst, int32_t* iter, int32_t* status, double* dbg) {
    // Pi: both kernels carry the factored weights; the fused one wins below N ~ 130 (2.49 vs 3.6 ms at N = 12 .. 64, 3.04 vs 3.54 ms at
    // N = 100, 3.73 vs 3.70 ms at N = 140; tools/time_methods.py)
    // (round 4: the two-wavefront workgroups of pi_wg_kernel.h win at every N -- Pi 2.12 / 2.42 ms at N = 12, 2.09 / 2.35 at 60, 2.28 / 3.04 at 100,
    // workgroup / fused, tools/ab_wg_fused.py; before, the fused kernel won below N = 128)
    const bool small = false;
    if (c->kernel_variant == 1 || c->solver != 0 || c->init_p || small)      // the debug outputs (init_p, init_x) come from the fused kernel
        return launch_pose(c, tff::k_pi_tft_pose<Model, false>, tff::k_pi_tft_pose<Model, true>, tff::pi_lds_bytes<Model>, 0, 4,
                           corresp, calm, calm_stride, B, N, Rt2, Rt3, T, reconst, iter, status, dbg);
    auto wg_lds = [](int n) { return (size_t)(((tff::POSE_LDS_DOUBLES + 1) & ~1) + tff::pi_wg_lds_doubles(Model::E, Model::C, n)) * sizeof(double); };
    // PiCol keeps the one-triplet-per-wavefront LINEAR stage: its scenes that take seven Gauss-Helmert iterations amplify a last-bit difference
    // of the start a million times (tools/diag_gh_noise_picol.py: 3.3e-10 from the 50-digit iteration with this start, 2.5e-9 with the rows
    // kernel's on the same N = 60 scene -- both draws of the same rounding noise, one of them over the 1e-9 gate of tests/test_gpu_gh_noise.py).
    // Its POSE TAIL (transform_TFT, R_t_from_TFT of the optimised tensor) is a fixed, well-conditioned function of that tensor and nothing
    // amplifies its rounding: it runs four triplets per wavefront like everyone else's since round 5 (k_gh_finish was 0.56 of PiCol's 5.4 ms).
    return launch_wg(c, tff::k_pi_block<Model>, wg_lds, 4, corresp, calm, calm_stride, B, N, Rt2, Rt3, T, reconst, iter, status, dbg,
                     false, !Model::PINV_KKT, false, tff::pi_wg_waves<Model>::value * tff::WAVE, 0, true);
}
int launch_ressl_tft(tff_ctx* c, const double* corresp, const double* calm, int64_t calm_stride, int64_t B, int32_t N,
                      double* Rt2, double* Rt3, double* T, double* reconst, int32_t* iter, int32_t* status, double* dbg) {
    return launch_gh<tff::ResslModel>(c, tff::k_gh_tft_pose<tff::ResslModel, false>, tff::k_gh_tft_pose<tff::ResslModel, true>,
                                      corresp, calm, calm_stride, B, N, Rt2, Rt3, T, reconst, iter, status, dbg);
}
int launch_nordberg_tft(tff_ctx* c, const double* corresp, const double* calm, int64_t calm_stride, int64_t B, int32_t N,
                        double* Rt2, double* Rt3, double* T, double* reconst, int32_t* iter, int32_t* status, double* dbg) {
    return launch_gh<tff::NordbergModel>(c, tff::k_gh_tft_pose<tff::NordbergModel, false>, tff::k_gh_tft_pose<tff::NordbergModel, true>,
                                         corresp, calm, calm_stride, B, N, Rt2, Rt3, T, reconst, iter, status, dbg);
}
int launch_faugpapa_tft(tff_ctx* c, const double* corresp, const double* calm, int64_t calm_stride, int64_t B, int32_t N,
                        double* Rt2, double* Rt3, double* T, double* reconst, int32_t* iter, int32_t* status, double* dbg) {
    return launch_gh<tff::FaugPapaModel>(c, tff::k_gh_tft_pose<tff::FaugPapaModel, false>, tff::k_gh_tft_pose<tff::FaugPapaModel, true>,
                                         corresp, calm, calm_stride, B, N, Rt2, Rt3, T, reconst, iter, status, dbg);
}
int launch_pi(tff_ctx* c, const double* corresp, const double* calm, int64_t calm_stride, int64_t B, int32_t N,
              double* Rt2, double* Rt3, double* T, double* reconst, int32_t* iter, int32_t* status, double* dbg) {
    return launch_pi_model<tff::PiModel>(c, corresp, calm, calm_stride, B, N, Rt2, Rt3, T, reconst, iter, status, dbg);
}
int launch_picol(tff_ctx* c, const double* corresp, const double* calm, int64_t calm_stride, int64_t B, int32_t N,
                 double* Rt2, double* Rt3, double* T, double* reconst, int32_t* iter, int32_t* status, double* dbg) {
    return launch_pi_model<tff::PiColModel>(c, corresp, calm, calm_stride, B, N, Rt2, Rt3, T, reconst, iter, status, dbg);
}

typedef int (*pose_launcher)(tff_ctx*, const double*, const double*, int64_t, int64_t, int32_t, double*, double*, double*, double*,
                             int32_t*, int32_t*, double*);

// host-pointer variant of any pose method: H2D, launch, D2H, synchronise
int pose_batch_host(pose_launcher launch, tff_ctx* c, const double* corresp, const double* calm, int64_t calm_stride, int64_t B,
                    int32_t N, double* Rt2, double* Rt3, double* T, double* reconst, int32_t* iter, int32_t* status) {
    if (int r = check_common(c, corresp, calm, calm_stride, B, N)) return r;
    TFF_LOCK(c);
    if (B == 0) return 0;
    if (!Rt2 || !Rt3 || !T) return fail(TFF_E_INVALID, "null output pointer");
    TFF_HIP(hipSetDevice(c->device));
    const size_t nin = (size_t)B * 6 * (size_t)N * sizeof(double);
    const size_t ncal = (calm_stride ? (size_t)B : 1) * 27 * sizeof(double);
    const size_t per_out = (12 + 12 + 27 + (reconst ? 3 * (size_t)N : 0)) * sizeof(double);
    if (int r = c->in.reserve(nin ? nin : 8)) return r;
    if (int r = c->calm.reserve(ncal)) return r;
    if (int r = c->out.reserve((size_t)B * per_out)) return r;
    if (int r = c->idx.reserve((size_t)B * 2 * sizeof(int32_t))) return r;
    double* d_in = (double*)c->in.p;
    double* d_cal = (double*)c->calm.p;
    double* d_Rt2 = (double*)c->out.p;
    double* d_Rt3 = d_Rt2 + (size_t)B * 12;
    double* d_T = d_Rt3 + (size_t)B * 12;
    double* d_rec = reconst ? d_T + (size_t)B * 27 : nullptr;
    int32_t* d_it = (int32_t*)c->idx.p;
    int32_t* d_st = d_it + B;
    if (nin) TFF_HIP(hipMemcpyAsync(d_in, corresp, nin, hipMemcpyHostToDevice, c->stream));
    TFF_HIP(hipMemcpyAsync(d_cal, calm, ncal, hipMemcpyHostToDevice, c->stream));
    if (int r = launch(c, d_in, d_cal, calm_stride, B, N, d_Rt2, d_Rt3, d_T, d_rec, d_it, d_st, nullptr)) return r;
    TFF_HIP(hipMemcpyAsync(Rt2, d_Rt2, (size_t)B * 12 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    TFF_HIP(hipMemcpyAsync(Rt3, d_Rt3, (size_t)B * 12 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    TFF_HIP(hipMemcpyAsync(T, d_T, (size_t)B * 27 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    if (reconst && N) TFF_HIP(hipMemcpyAsync(reconst, d_rec, (size_t)B * 3 * (size_t)N * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    if (iter) TFF_HIP(hipMemcpyAsync(iter, d_it, (size_t)B * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    if (status) TFF_HIP(hipMemcpyAsync(status, d_st, (size_t)B * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    TFF_HIP(hipStreamSynchronize(c->stream));
    return 0;
}

}  // namespace

// BundleAdjustment as the reference writes it, M = 2 .. 6 views, MATLAB's own array layouts (csrc/ba_views_kernel.h)
template <int M>
static int launch_bundle_adjust_views(tff_ctx* c, const tff::BavArgs& a) {
    const size_t lds = tff::bav_lds_bytes<M>(a.N);
    if (int r = ensure_lds(tff::k_bundle_adjust_views<M>, lds)) return r;
    hipLaunchKernelGGL(tff::k_bundle_adjust_views<M>, dim3(tff::pose_grid(a.B)), dim3(64), lds, c->stream, a);
    TFF_HIP(hipGetLastError());
    return 0;
}
static int check_views(const tff_ctx* c, int32_t M, const void* calm, int64_t calm_stride, const void* Rt_in, const void* corresp, int64_t B, int32_t N, const void* Rt) {
    if (!c) return fail(TFF_E_INVALID, "null context");
    if (M < tff::BAV_MIN_VIEWS || M > tff::BAV_MAX_VIEWS) return fail(TFF_E_INVALID, "bundle adjustment takes 2 .. 6 views");
    if (B < 0 || N < 0) return fail(TFF_E_INVALID, "negative batch or correspondence count");
    if (B > 0 && (!corresp || !calm || !Rt_in || !Rt)) return fail(TFF_E_INVALID, "null pointer");
    if (calm_stride != 0 && calm_stride != 9 * (int64_t)M) return fail(TFF_E_INVALID, "calm_stride must be 0 (shared CalM) or 9 M");
    if (B > 0 && N < 1) return fail(TFF_E_INVALID, "bundle adjustment needs at least one correspondence");
    return 0;
}

extern "C" {

int tff_version(void) { return 100; }
const char* tff_last_error(void) { return g_err.c_str(); }

int tff_ctx_create(tff_ctx** out, int device) {
    if (!out) return fail(TFF_E_INVALID, "null out pointer");
    *out = nullptr;
    int n = 0;
    TFF_HIP(hipGetDeviceCount(&n));
    if (device < 0 || device >= n) return fail(TFF_E_INVALID, "no such HIP device (libtftfund has no CPU path)");
    TFF_HIP(hipSetDevice(device));
    tff_ctx* c = new (std::nothrow) tff_ctx();
    if (!c) return fail(TFF_E_NOMEM, "out of host memory");
    c->device = device;
    hipError_t e = hipStreamCreateWithFlags(&c->own, hipStreamNonBlocking);
    if (e != hipSuccess) { delete c; return hip_fail(e, "hipStreamCreate"); }
    c->stream = c->own;
    *out = c;
    return 0;
}

void tff_ctx_destroy(tff_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->own) { (void)hipStreamSynchronize(c->own); (void)hipStreamDestroy(c->own); }
    if (c->handover) (void)hipEventDestroy(c->handover);
    c->in.release(); c->calm.release(); c->out.release(); c->idx.release(); c->scratch_status.release(); c->gh_rec.release(); c->gh_topt.release(); c->gh_init.release(); c->spill.release(); c->pre_rec.release(); c->retry.release();
    delete c;
}

// The workspaces of a context (status scratch, Gauss-Helmert records, spill slices, host-path staging) are reused by every
// call, so work enqueued on the previous stream must finish before work on the new one touches them: an event recorded on the
// old stream, waited for by the new one (no host synchronisation).
static int switch_stream(tff_ctx* c, hipStream_t s) {
    if (s == c->stream) return 0;
    // the hand-over touches the context's device; the calling thread's current device is the caller's business and is put back
    int caller_device = -1;
    (void)hipGetDevice(&caller_device);
    struct Restore { int d; ~Restore() { if (d >= 0) (void)hipSetDevice(d); } } restore{caller_device};
    TFF_HIP(hipSetDevice(c->device));
    if (!c->handover) TFF_HIP(hipEventCreateWithFlags(&c->handover, hipEventDisableTiming));
    // The previous stream must still be alive here (a caller that destroys its stream first hands the context a dangling handle).  If
    // recording on it fails all the same -- a destroyed user stream -- the new stream is adopted anyway, after draining the context's
    // own stream and the device: refusing would leave the context stuck on the dead stream for every later call.
    if (hipEventRecord(c->handover, c->stream) == hipSuccess) {
        TFF_HIP(hipStreamWaitEvent(s, c->handover, 0));
    } else {
        (void)hipGetLastError();
        (void)hipStreamSynchronize(c->own);
        (void)hipDeviceSynchronize();
    }
    c->stream = s;
    return 0;
}
int tff_ctx_set_stream(tff_ctx* c, void* s) {
    if (!c) return fail(TFF_E_INVALID, "null context");
    TFF_LOCK(c);
    return switch_stream(c, (hipStream_t)s);
}
int tff_ctx_use_own_stream(tff_ctx* c) {
    if (!c) return fail(TFF_E_INVALID, "null context");
    TFF_LOCK(c);
    return switch_stream(c, c->own);
}
void* tff_ctx_get_stream(tff_ctx* c) { return c ? (void*)c->stream : nullptr; }

int tff_ctx_set_option(tff_ctx* c, int option, long value) {
    if (!c) return fail(TFF_E_INVALID, "null context");
    TFF_LOCK(c);
    switch (option) {
        case TFF_OPT_SOLVER: if (value != 0 && value != 1) return fail(TFF_E_INVALID, "solver must be 0 or 1"); c->solver = (int)value; return 0;
        case TFF_OPT_EXACT_BELOW: if (value < 0 || value > (1L << 30)) return fail(TFF_E_INVALID, "exact_below must be >= 0"); c->exact_below = (int)value; return 0;
        case TFF_OPT_STAGE_LDS: if (value < -1 || value > 1) return fail(TFF_E_INVALID, "stage_lds must be -1, 0 or 1"); c->stage = (int)value; return 0;
        case TFF_OPT_GH_EXACT: c->gh_exact = value != 0; return 0;
        case TFF_OPT_SPILL: c->spill_only_if_needed = value != 0; return 0;
        case TFF_OPT_ROWS: if (value < 0 || value > 2) return fail(TFF_E_INVALID, "rows must be 0, 1 or 2"); c->rows = (int)value; return 0;
        case TFF_OPT_PRE: if (value < 0 || value > 2) return fail(TFF_E_INVALID, "pre must be 0, 1 or 2"); c->pre = (int)value; return 0;
        case TFF_OPT_COUNT_ROWS: c->count_rows = value != 0; return 0;
        case TFF_OPT_DEBUG_FP_HANDOVER: c->dbg_fp_handover = value != 0; return 0;
        case TFF_OPT_DEBUG_ADAPTIVE: c->dbg_adaptive = value != 0; return 0;
        case TFF_OPT_KERNEL: if (value < 0 || value > 2) return fail(TFF_E_INVALID, "kernel must be 0, 1 or 2"); c->kernel_variant = (int)value; return 0;
        default: return fail(TFF_E_INVALID, "unknown option");
    }
}

int tff_ctx_synchronize(tff_ctx* c) {
    if (!c) return fail(TFF_E_INVALID, "null context");
    TFF_LOCK(c);
    TFF_HIP(hipSetDevice(c->device));
    TFF_HIP(hipStreamSynchronize(c->stream));
    return 0;
}

int tff_linear_tft_pose_batch_dev(tff_ctx* c, const double* corresp, const double* calm, int64_t calm_stride, int64_t B,
                                  int32_t N, double* Rt2, double* Rt3, double* T, double* reconst, int32_t* iter,
                                  int32_t* status) {
    return launch_linear_tft(c, corresp, calm, calm_stride, B, N, Rt2, Rt3, T, reconst, iter, status, nullptr);
}

int tff_linear_tft_pose_batch_debug_dev(tff_ctx* c, const double* corresp, const double* calm, int64_t calm_stride,
                                        int64_t B, int32_t N, double* Rt2, double* Rt3, double* T, double* reconst,
                                        int32_t* iter, int32_t* status, double* dbg) {
    if (!dbg) return fail(TFF_E_INVALID, "null debug buffer");
    return launch_linear_tft(c, corresp, calm, calm_stride, B, N, Rt2, Rt3, T, reconst, iter, status, dbg);
}

int tff_linear_tft_pose_batch_host(tff_ctx* c, const double* corresp, const double* calm, int64_t calm_stride, int64_t B,
                                   int32_t N, double* Rt2, double* Rt3, double* T, double* reconst, int32_t* iter,
                                   int32_t* status) {
    return pose_batch_host(launch_linear_tft, c, corresp, calm, calm_stride, B, N, Rt2, Rt3, T, reconst, iter, status);
}

int tff_ressl_tft_pose_batch_dev(tff_ctx* c, const double* corresp, const double* calm, int64_t calm_stride, int64_t B,
                                  int32_t N, double* Rt2, double* Rt3, double* T, double* reconst, int32_t* iter,
                                  int32_t* status) {
    return launch_ressl_tft(c, corresp, calm, calm_stride, B, N, Rt2, Rt3, T, reconst, iter, status, nullptr);
}
int tff_ressl_tft_pose_batch_host(tff_ctx* c, const double* corresp, const double* calm, int64_t calm_stride, int64_t B,
                                   int32_t N, double* Rt2, double* Rt3, double* T, double* reconst, int32_t* iter,
                                   int32_t* status) {
    return pose_batch_host(launch_ressl_tft, c, corresp, calm, calm_stride, B, N, Rt2, Rt3, T, reconst, iter, status);
}
int tff_ressl_tft_pose_batch_debug_dev(tff_ctx* c, const double* corresp, const double* calm, int64_t calm_stride,
                                       int64_t B, int32_t N, double* Rt2, double* Rt3, double* T, double* reconst,
                                       int32_t* iter, int32_t* status, double* dbg) {
    if (!dbg) return fail(TFF_E_INVALID, "null debug buffer");
    return launch_ressl_tft(c, corresp, calm, calm_stride, B, N, Rt2, Rt3, T, reconst, iter, status, dbg);
}

int tff_nordberg_tft_pose_batch_dev(tff_ctx* c, const double* corresp, const double* calm, int64_t calm_stride, int64_t B,
                                     int32_t N, double* Rt2, double* Rt3, double* T, double* reconst, int32_t* iter,
                                     int32_t* status) {
    return launch_nordberg_tft(c, corresp, calm, calm_stride, B, N, Rt2, Rt3, T, reconst, iter, status, nullptr);
}
int tff_nordberg_tft_pose_batch_debug_dev(tff_ctx* c, const double* corresp, const double* calm, int64_t calm_stride, int64_t B,
                                           int32_t N, double* Rt2, double* Rt3, double* T, double* reconst, int32_t* iter,
                                           int32_t* status, double* dbg) {
    return launch_nordberg_tft(c, corresp, calm, calm_stride, B, N, Rt2, Rt3, T, reconst, iter, status, dbg);
}
int tff_nordberg_tft_pose_batch_host(tff_ctx* c, const double* corresp, const double* calm, int64_t calm_stride, int64_t B,
                                      int32_t N, double* Rt2, double* Rt3, double* T, double* reconst, int32_t* iter,
                                      int32_t* status) {
    return pose_batch_host(launch_nordberg_tft, c, corresp, calm, calm_stride, B, N, Rt2, Rt3, T, reconst, iter, status);
}
int tff_faugpapa_tft_pose_batch_dev(tff_ctx* c, const double* corresp, const double* calm, int64_t calm_stride, int64_t B,
                                     int32_t N, double* Rt2, double* Rt3, double* T, double* reconst, int32_t* iter,
                                     int32_t* status) {
    return launch_faugpapa_tft(c, corresp, calm, calm_stride, B, N, Rt2, Rt3, T, reconst, iter, status, nullptr);
}
int tff_faugpapa_tft_pose_batch_debug_dev(tff_ctx* c, const double* corresp, const double* calm, int64_t calm_stride, int64_t B,
                                           int32_t N, double* Rt2, double* Rt3, double* T, double* reconst, int32_t* iter,
                                           int32_t* status, double* dbg) {
    return launch_faugpapa_tft(c, corresp, calm, calm_stride, B, N, Rt2, Rt3, T, reconst, iter, status, dbg);
}
int tff_faugpapa_tft_pose_batch_host(tff_ctx* c, const double* corresp, const double* calm, int64_t calm_stride, int64_t B,
                                      int32_t N, double* Rt2, double* Rt3, double* T, double* reconst, int32_t* iter,
                                      int32_t* status) {
    return pose_batch_host(launch_faugpapa_tft, c, corresp, calm, calm_stride, B, N, Rt2, Rt3, T, reconst, iter, status);
}
int tff_pi_pose_batch_dev(tff_ctx* c, const double* corresp, const double* calm, int64_t calm_stride, int64_t B,
                           int32_t N, double* Rt2, double* Rt3, double* T, double* reconst, int32_t* iter, int32_t* status) {
    return launch_pi(c, corresp, calm, calm_stride, B, N, Rt2, Rt3, T, reconst, iter, status, nullptr);
}
int tff_pi_pose_batch_host(tff_ctx* c, const double* corresp, const double* calm, int64_t calm_stride, int64_t B,
                            int32_t N, double* Rt2, double* Rt3, double* T, double* reconst, int32_t* iter, int32_t* status) {
    return pose_batch_host(launch_pi, c, corresp, calm, calm_stride, B, N, Rt2, Rt3, T, reconst, iter, status);
}
int tff_picol_pose_batch_dev(tff_ctx* c, const double* corresp, const double* calm, int64_t calm_stride, int64_t B,
                              int32_t N, double* Rt2, double* Rt3, double* T, double* reconst, int32_t* iter, int32_t* status) {
    return launch_picol(c, corresp, calm, calm_stride, B, N, Rt2, Rt3, T, reconst, iter, status, nullptr);
}
int tff_picol_pose_batch_host(tff_ctx* c, const double* corresp, const double* calm, int64_t calm_stride, int64_t B,
                               int32_t N, double* Rt2, double* Rt3, double* T, double* reconst, int32_t* iter, int32_t* status) {
    return pose_batch_host(launch_picol, c, corresp, calm, calm_stride, B, N, Rt2, Rt3, T, reconst, iter, status);
}
int tff_pi_pose_batch_debug_dev(tff_ctx* c, int32_t collinear, const double* corresp, const double* calm, int64_t calm_stride, int64_t B,
                                 int32_t N, double* Rt2, double* Rt3, double* T, double* reconst, int32_t* iter, int32_t* status,
                                 double* init_p, double* init_x) {
    if (!c) return fail(TFF_E_INVALID, "null context");
    TFF_LOCK(c);
    if ((init_p == nullptr) != (init_x == nullptr)) return fail(TFF_E_INVALID, "init_p and init_x come together");
    c->init_p = init_p; c->init_x = init_x;
    const int r = (collinear ? launch_picol : launch_pi)(c, corresp, calm, calm_stride, B, N, Rt2, Rt3, T, reconst, iter, status, nullptr);
    c->init_p = nullptr; c->init_x = nullptr;
    return r;
}
int tff_optim_f_pose_batch_dev(tff_ctx* c, const double* corresp, const double* calm, int64_t calm_stride, int64_t B,
                                int32_t N, double* Rt2, double* Rt3, double* T, double* reconst, int32_t* iter,
                                int32_t* status) {
    return launch_optim_f(c, corresp, calm, calm_stride, B, N, Rt2, Rt3, T, reconst, iter, status, nullptr);
}
int tff_optim_f_pose_batch_host(tff_ctx* c, const double* corresp, const double* calm, int64_t calm_stride, int64_t B,
                                 int32_t N, double* Rt2, double* Rt3, double* T, double* reconst, int32_t* iter,
                                 int32_t* status) {
    return pose_batch_host(launch_optim_f, c, corresp, calm, calm_stride, B, N, Rt2, Rt3, T, reconst, iter, status);
}

int tff_linear_f_pose_batch_dev(tff_ctx* c, const double* corresp, const double* calm, int64_t calm_stride, int64_t B,
                                int32_t N, double* Rt2, double* Rt3, double* T, double* reconst, int32_t* iter,
                                int32_t* status) {
    return launch_linear_f(c, corresp, calm, calm_stride, B, N, Rt2, Rt3, T, reconst, iter, status, nullptr);
}
int tff_linear_f_pose_batch_debug_dev(tff_ctx* c, const double* corresp, const double* calm, int64_t calm_stride, int64_t B,
                                      int32_t N, double* Rt2, double* Rt3, double* T, double* reconst, int32_t* iter,
                                      int32_t* status, double* dbg) {
    if (!dbg) return fail(TFF_E_INVALID, "null debug buffer");
    return launch_linear_f(c, corresp, calm, calm_stride, B, N, Rt2, Rt3, T, reconst, iter, status, dbg);
}
int tff_linear_f_pose_batch_host(tff_ctx* c, const double* corresp, const double* calm, int64_t calm_stride, int64_t B,
                                 int32_t N, double* Rt2, double* Rt3, double* T, double* reconst, int32_t* iter,
                                 int32_t* status) {
    return pose_batch_host(launch_linear_f, c, corresp, calm, calm_stride, B, N, Rt2, Rt3, T, reconst, iter, status);
}


// ---------------------------------------------------------------------------------------------
// Building blocks (device pointers only)
// ---------------------------------------------------------------------------------------------
int tff_triangulate_batch_dev(tff_ctx* c, const double* cams, int64_t cam_stride, const double* pts, int64_t B, int32_t M,
                              int32_t N, double* X) {
    if (!c) return fail(TFF_E_INVALID, "null context");
    TFF_LOCK(c);
    if (B < 0 || N < 0 || (M != 2 && M != 3)) return fail(TFF_E_INVALID, "triangulate: M must be 2 or 3");    // triangulation3D.m:33,46
    if (cam_stride != 0 && cam_stride != 12 * M) return fail(TFF_E_INVALID, "cam_stride must be 0 or 12*M");
    if (B == 0 || N == 0) return 0;
    if (!cams || !pts || !X) return fail(TFF_E_INVALID, "null pointer");
    TFF_HIP(hipSetDevice(c->device));
    tff::TriangulateArgs a{cams, (long)cam_stride, pts, (long)B, M, N, X};
    hipLaunchKernelGGL(tff::k_triangulate, dim3(tff::pose_grid(B)), dim3(64), 0, c->stream, a);
    TFF_HIP(hipGetLastError());
    return 0;
}

int tff_repr_error_batch_dev(tff_ctx* c, const double* cams, int64_t cam_stride, const double* corresp, int64_t corresp_stride,
                             const double* pts3d, int64_t B, int32_t N, double* err) {
    if (!c) return fail(TFF_E_INVALID, "null context");
    TFF_LOCK(c);
    if (B < 0 || N < 0) return fail(TFF_E_INVALID, "negative size");
    if (cam_stride != 0 && cam_stride != 36) return fail(TFF_E_INVALID, "cam_stride must be 0 or 36");
    if (corresp_stride != 0 && corresp_stride != 6 * (int64_t)N) return fail(TFF_E_INVALID, "corresp_stride must be 0 or 6*N");
    if (B == 0) return 0;
    if (!cams || !corresp || !err) return fail(TFF_E_INVALID, "null pointer");
    TFF_HIP(hipSetDevice(c->device));
    tff::ReprErrorArgs a{cams, (long)cam_stride, nullptr, nullptr, nullptr, corresp, (long)corresp_stride, pts3d, (long)B, N, 0.0, err, nullptr};
    hipLaunchKernelGGL(tff::k_repr_error, dim3(tff::pose_grid(B)), dim3(64), 0, c->stream, a);
    TFF_HIP(hipGetLastError());
    return 0;
}

int tff_inlier_count_batch_dev(tff_ctx* c, const double* scene, int32_t Ns, const double* calm, const double* Rt2, const double* Rt3,
                               int64_t B, double threshold, int32_t* counts, double* err) {
    if (!c) return fail(TFF_E_INVALID, "null context");
    TFF_LOCK(c);
    if (B < 0 || Ns < 0) return fail(TFF_E_INVALID, "negative size");
    if (B == 0) return 0;
    if (!scene || !calm || !Rt2 || !Rt3 || !counts) return fail(TFF_E_INVALID, "null pointer");
    TFF_HIP(hipSetDevice(c->device));
    tff::ReprErrorArgs a{nullptr, 0, calm, Rt2, Rt3, scene, 0, nullptr, (long)B, Ns, threshold, err, counts};
    const size_t staged = ((size_t)6 * Ns + 36 * tff::INLIER_WG_WAVES) * sizeof(double);
    if (!err && staged <= 48 * 1024 && B >= 4096) {
        // counts only, many hypotheses, a scene that fits the LDS a few times over: stage it once per workgroup (blocks_kernel.h)
        const int per_cu = (int)((LDS_LIMIT / (staged + 512) < 4) ? LDS_LIMIT / (staged + 512) : 4);
        long grid = 256L * per_cu;
        if (grid * tff::INLIER_WG_WAVES > B) grid = (B + tff::INLIER_WG_WAVES - 1) / tff::INLIER_WG_WAVES;
        if (c->count_rows) {                                                 // four hypotheses per wavefront (blocks_kernel.h::k_inlier_count_rows): two workgroups per CU
            const size_t staged_rows = ((size_t)6 * Ns + 36 * 4 * tff::INLIER_WG_WAVES) * sizeof(double);
            long grid_rows = 256L * 2;
            const long per_wg = 4L * tff::INLIER_WG_WAVES;
            if (grid_rows * per_wg > B) grid_rows = (B + per_wg - 1) / per_wg;
            if (int r = ensure_lds(tff::k_inlier_count_rows, staged_rows)) return r;
            hipLaunchKernelGGL(tff::k_inlier_count_rows, dim3((unsigned)grid_rows), dim3(64 * tff::INLIER_WG_WAVES), staged_rows, c->stream, a);
        } else {
            hipLaunchKernelGGL(tff::k_inlier_count_staged, dim3((unsigned)grid), dim3(64 * tff::INLIER_WG_WAVES), staged, c->stream, a);
        }
    } else {
        hipLaunchKernelGGL(tff::k_repr_error, dim3(tff::pose_grid(B)), dim3(64), 0, c->stream, a);
    }
    TFF_HIP(hipGetLastError());
    return 0;
}

int tff_transform_tft_batch_dev(tff_ctx* c, const double* T, const double* M1, const double* M2, const double* M3, int64_t m_stride,
                                int64_t B, int32_t inverse, double* Tout) {
    if (!c) return fail(TFF_E_INVALID, "null context");
    TFF_LOCK(c);
    if (B < 0 || (m_stride != 0 && m_stride != 9) || (inverse != 0 && inverse != 1)) return fail(TFF_E_INVALID, "bad argument");
    if (B == 0) return 0;
    if (!T || !M1 || !M2 || !M3 || !Tout) return fail(TFF_E_INVALID, "null pointer");
    TFF_HIP(hipSetDevice(c->device));
    tff::TransformArgs a{T, M1, M2, M3, (long)m_stride, (long)B, inverse, Tout};
    hipLaunchKernelGGL(tff::k_transform_tft, dim3(tff::pose_grid(B)), dim3(64), 0, c->stream, a);
    TFF_HIP(hipGetLastError());
    return 0;
}

int tff_rt_from_tft_batch_dev(tff_ctx* c, const double* T, const double* calm, int64_t calm_stride, const double* corresp, int64_t B,
                              int32_t N, double* Rt2, double* Rt3, int32_t* status) {
    if (int r = check_common(c, corresp, calm, calm_stride, B, N)) return r;
    TFF_LOCK(c);
    if (B == 0) return 0;
    if (!T || !Rt2 || !Rt3) return fail(TFF_E_INVALID, "null pointer");
    TFF_HIP(hipSetDevice(c->device));
    tff::RtFromTftArgs a{T, calm, (long)calm_stride, corresp, (long)B, N, Rt2, Rt3, status};
    const size_t lds = tff::pose_lds_bytes(N, 0, false);
    hipLaunchKernelGGL(tff::k_rt_from_tft, dim3(tff::pose_grid(B)), dim3(64), lds, c->stream, a);
    TFF_HIP(hipGetLastError());
    return 0;
}

int tff_linear_tft_batch_dev(tff_ctx* c, const double* corresp, int64_t B, int32_t N, double* T, double* P2, double* P3,
                             int32_t* status) {
    if (!c) return fail(TFF_E_INVALID, "null context");
    TFF_LOCK(c);
    if (B < 0 || N < 0) return fail(TFF_E_INVALID, "negative size");
    if (B == 0) return 0;
    if (!corresp || !T || ((P2 == nullptr) != (P3 == nullptr))) return fail(TFF_E_INVALID, "null pointer (P2 and P3 come together)");
    TFF_HIP(hipSetDevice(c->device));
    if (!status) {
        if (int r = c->scratch_status.reserve((size_t)B * sizeof(int32_t))) return r;
        status = (int32_t*)c->scratch_status.p;
    }
    tff::LinearTftOnlyArgs a{corresp, (long)B, N, 0, T, P2, P3, status};
    const bool all_exact = c->solver != 0 || N < c->exact_below;
    if (!all_exact) {
        hipLaunchKernelGGL(tff::k_linear_tft<false>, dim3(tff::pose_grid(B)), dim3(64), tff::pose_lds_bytes(N, 0, false), c->stream, a);
        TFF_HIP(hipGetLastError());
        a.flags |= tff::FLAG_ONLY_RETRY;
    }
    const unsigned grid = !all_exact ? (unsigned)(B < FIXUP_GRID ? B : FIXUP_GRID) : tff::pose_grid(B);
    hipLaunchKernelGGL(tff::k_linear_tft<true>, dim3(grid), dim3(64), tff::pose_lds_bytes(N, 0, true), c->stream, a);
    TFF_HIP(hipGetLastError());
    return 0;
}

// BundleAdjustment for three views: refines (R_t_2, R_t_3) and the points
int tff_bundle_adjust_batch_dev(tff_ctx* c, const double* calm, int64_t calm_stride, const double* Rt2_in, const double* Rt3_in,
                                const double* corresp, int64_t B, int32_t N, const double* reconst0, double* Rt2, double* Rt3,
                                double* reconst, int32_t* iter, double* repr_err, int32_t* status) {
    if (int r = check_common(c, corresp, calm, calm_stride, B, N)) return r;
    TFF_LOCK(c);
    if (B == 0) return 0;
    if (!Rt2_in || !Rt3_in || !Rt2 || !Rt3) return fail(TFF_E_INVALID, "null pose pointer");
    if (N < 1) return fail(TFF_E_INVALID, "bundle adjustment needs at least one correspondence");
    TFF_HIP(hipSetDevice(c->device));
    const size_t lds = tff::ba_lds_bytes(N);
    if (int r = ensure_lds(tff::k_bundle_adjust, lds)) return r;
    tff::BaArgs a{calm, (long)calm_stride, Rt2_in, Rt3_in, corresp, (long)B, N, reconst0, Rt2, Rt3, reconst, iter, repr_err, status};
    hipLaunchKernelGGL(tff::k_bundle_adjust, dim3(tff::pose_grid(B)), dim3(64), lds, c->stream, a);
    TFF_HIP(hipGetLastError());
    return 0;
}

// host-pointer variant: H2D, launch, D2H, synchronise (what the MEX shim calls)
int tff_bundle_adjust_batch_host(tff_ctx* c, const double* calm, int64_t calm_stride, const double* Rt2_in, const double* Rt3_in,
                                 const double* corresp, int64_t B, int32_t N, const double* reconst0, double* Rt2, double* Rt3,
                                 double* reconst, int32_t* iter, double* repr_err, int32_t* status) {
    if (int r = check_common(c, corresp, calm, calm_stride, B, N)) return r;
    TFF_LOCK(c);
    if (B == 0) return 0;
    if (!Rt2_in || !Rt3_in || !Rt2 || !Rt3) return fail(TFF_E_INVALID, "null pose pointer");
    TFF_HIP(hipSetDevice(c->device));
    const size_t nin = (size_t)B * 6 * (size_t)N * sizeof(double), npt = (size_t)B * 3 * (size_t)N * sizeof(double);
    const size_t ncal = (calm_stride ? (size_t)B : 1) * 27 * sizeof(double), npose = (size_t)B * 12 * sizeof(double);
    if (int r = c->in.reserve(nin + npt + 2 * npose)) return r;
    if (int r = c->calm.reserve(ncal)) return r;
    if (int r = c->out.reserve(2 * npose + npt + (size_t)B * sizeof(double))) return r;
    if (int r = c->idx.reserve((size_t)B * 2 * sizeof(int32_t))) return r;
    char* din = (char*)c->in.p;
    double* d_C = (double*)din; double* d_X0 = (double*)(din + nin); double* d_r2 = (double*)(din + nin + npt); double* d_r3 = (double*)(din + nin + npt + npose);
    char* dout = (char*)c->out.p;
    double* d_o2 = (double*)dout; double* d_o3 = (double*)(dout + npose); double* d_rec = (double*)(dout + 2 * npose); double* d_err = (double*)(dout + 2 * npose + npt);
    int32_t* d_it = (int32_t*)c->idx.p; int32_t* d_st = d_it + B;
    TFF_HIP(hipMemcpyAsync(d_C, corresp, nin, hipMemcpyHostToDevice, c->stream));
    if (reconst0) TFF_HIP(hipMemcpyAsync(d_X0, reconst0, npt, hipMemcpyHostToDevice, c->stream));
    TFF_HIP(hipMemcpyAsync(d_r2, Rt2_in, npose, hipMemcpyHostToDevice, c->stream));
    TFF_HIP(hipMemcpyAsync(d_r3, Rt3_in, npose, hipMemcpyHostToDevice, c->stream));
    TFF_HIP(hipMemcpyAsync(c->calm.p, calm, ncal, hipMemcpyHostToDevice, c->stream));
    if (int r = tff_bundle_adjust_batch_dev(c, (double*)c->calm.p, calm_stride, d_r2, d_r3, d_C, B, N, reconst0 ? d_X0 : nullptr, d_o2, d_o3, d_rec, d_it, d_err, d_st)) return r;
    TFF_HIP(hipMemcpyAsync(Rt2, d_o2, npose, hipMemcpyDeviceToHost, c->stream));
    TFF_HIP(hipMemcpyAsync(Rt3, d_o3, npose, hipMemcpyDeviceToHost, c->stream));
    if (reconst) TFF_HIP(hipMemcpyAsync(reconst, d_rec, npt, hipMemcpyDeviceToHost, c->stream));
    if (iter) TFF_HIP(hipMemcpyAsync(iter, d_it, (size_t)B * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    if (repr_err) TFF_HIP(hipMemcpyAsync(repr_err, d_err, (size_t)B * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    if (status) TFF_HIP(hipMemcpyAsync(status, d_st, (size_t)B * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    TFF_HIP(hipStreamSynchronize(c->stream));
    return 0;
}

// BundleAdjustment as the reference writes it, M = 2 .. 6 views (launch_bundle_adjust_views above)
int tff_bundle_adjust_views_batch_dev(tff_ctx* c, int32_t M, const double* calm, int64_t calm_stride, const double* Rt_in, const double* corresp,
                                      int64_t B, int32_t N, const double* reconst0, double* Rt, double* reconst, int32_t* iter, double* repr_err,
                                      int32_t* status) {
    if (int r = check_views(c, M, calm, calm_stride, Rt_in, corresp, B, N, Rt)) return r;
    TFF_LOCK(c);
    if (B == 0) return 0;
    TFF_HIP(hipSetDevice(c->device));
    const tff::BavArgs a{calm, (long)calm_stride, Rt_in, corresp, (long)B, N, reconst0, Rt, reconst, iter, repr_err, status};
    switch (M) {
        case 2: return launch_bundle_adjust_views<2>(c, a);
        case 3: return launch_bundle_adjust_views<3>(c, a);
        case 4: return launch_bundle_adjust_views<4>(c, a);
        case 5: return launch_bundle_adjust_views<5>(c, a);
        default: return launch_bundle_adjust_views<6>(c, a);
    }
}
int tff_bundle_adjust_views_batch_host(tff_ctx* c, int32_t M, const double* calm, int64_t calm_stride, const double* Rt_in, const double* corresp,
                                       int64_t B, int32_t N, const double* reconst0, double* Rt, double* reconst, int32_t* iter, double* repr_err,
                                       int32_t* status) {
    if (int r = check_views(c, M, calm, calm_stride, Rt_in, corresp, B, N, Rt)) return r;
    TFF_LOCK(c);
    if (B == 0) return 0;
    TFF_HIP(hipSetDevice(c->device));
    const size_t nin = (size_t)B * 2 * M * (size_t)N * sizeof(double), npt = (size_t)B * 3 * (size_t)N * sizeof(double);
    const size_t ncal = (calm_stride ? (size_t)B : 1) * 9 * M * sizeof(double), npose = (size_t)B * 12 * M * sizeof(double);
    if (int r = c->in.reserve(nin + npt + npose)) return r;
    if (int r = c->calm.reserve(ncal)) return r;
    if (int r = c->out.reserve(npose + npt + (size_t)B * sizeof(double))) return r;
    if (int r = c->idx.reserve((size_t)B * 2 * sizeof(int32_t))) return r;
    char* din = (char*)c->in.p;
    double* d_C = (double*)din; double* d_X0 = (double*)(din + nin); double* d_r = (double*)(din + nin + npt);
    char* dout = (char*)c->out.p;
    double* d_o = (double*)dout; double* d_rec = (double*)(dout + npose); double* d_err = (double*)(dout + npose + npt);
    int32_t* d_it = (int32_t*)c->idx.p; int32_t* d_st = d_it + B;
    TFF_HIP(hipMemcpyAsync(d_C, corresp, nin, hipMemcpyHostToDevice, c->stream));
    if (reconst0) TFF_HIP(hipMemcpyAsync(d_X0, reconst0, npt, hipMemcpyHostToDevice, c->stream));
    TFF_HIP(hipMemcpyAsync(d_r, Rt_in, npose, hipMemcpyHostToDevice, c->stream));
    TFF_HIP(hipMemcpyAsync(c->calm.p, calm, ncal, hipMemcpyHostToDevice, c->stream));
    if (int r = tff_bundle_adjust_views_batch_dev(c, M, (double*)c->calm.p, calm_stride, d_r, d_C, B, N, reconst0 ? d_X0 : nullptr, d_o, d_rec, d_it, d_err, d_st)) return r;
    TFF_HIP(hipMemcpyAsync(Rt, d_o, npose, hipMemcpyDeviceToHost, c->stream));
    if (reconst) TFF_HIP(hipMemcpyAsync(reconst, d_rec, npt, hipMemcpyDeviceToHost, c->stream));
    if (iter) TFF_HIP(hipMemcpyAsync(iter, d_it, (size_t)B * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    if (repr_err) TFF_HIP(hipMemcpyAsync(repr_err, d_err, (size_t)B * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    if (status) TFF_HIP(hipMemcpyAsync(status, d_st, (size_t)B * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    TFF_HIP(hipStreamSynchronize(c->stream));
    return 0;
}

// linearF (refine = 0) / optimF (refine = 1) for the view pairs (1,2) and (1,3)
int tff_linear_f_batch_dev(tff_ctx* c, const double* corresp, int64_t B, int32_t N, int32_t refine, double* F21, double* F31,
                           int32_t* iter, int32_t* status) {
    if (!c) return fail(TFF_E_INVALID, "null context");
    TFF_LOCK(c);
    if (B < 0 || N < 0) return fail(TFF_E_INVALID, "negative size");
    if (B == 0) return 0;
    if (!corresp || !F21 || !F31) return fail(TFF_E_INVALID, "null pointer");
    TFF_HIP(hipSetDevice(c->device));
    if (!status) {
        if (int r = c->scratch_status.reserve((size_t)B * sizeof(int32_t))) return r;
        status = (int32_t*)c->scratch_status.p;
    }
    tff::LinearFOnlyArgs a{corresp, (long)B, N, 0, F21, F31, iter, status};
    const bool all_exact = c->solver != 0 || N < c->exact_below;
    const unsigned fix_grid = !all_exact ? (unsigned)(B < FIXUP_GRID ? B : FIXUP_GRID) : tff::pose_grid(B);
    if (refine) {
        if (!all_exact) {
            const size_t lds = tff::optimf_lds_bytes(N, 0, false);
            if (int r = ensure_lds(tff::k_linear_f<false, 1>, lds)) return r;
            hipLaunchKernelGGL((tff::k_linear_f<false, 1>), dim3(tff::pose_grid(B)), dim3(64), lds, c->stream, a);
            TFF_HIP(hipGetLastError());
            a.flags |= tff::FLAG_ONLY_RETRY;
        }
        const size_t lds = tff::optimf_lds_bytes(N, 0, true);
        if (int r = ensure_lds(tff::k_linear_f<true, 1>, lds)) return r;
        hipLaunchKernelGGL((tff::k_linear_f<true, 1>), dim3(fix_grid), dim3(64), lds, c->stream, a);
    } else {
        if (!all_exact) {
            hipLaunchKernelGGL((tff::k_linear_f<false, 0>), dim3(tff::pose_grid(B)), dim3(64), tff::f_pose_lds_bytes(N, 0, false), c->stream, a);
            TFF_HIP(hipGetLastError());
            a.flags |= tff::FLAG_ONLY_RETRY;
        }
        hipLaunchKernelGGL((tff::k_linear_f<true, 0>), dim3(fix_grid), dim3(64), tff::f_pose_lds_bytes(N, 0, true), c->stream, a);
    }
    TFF_HIP(hipGetLastError());
    return 0;
}

// Minimal-sample hypotheses (config 4): hypothesis b uses correspondences sample_idx[b*n .. b*n+n) of ONE shared scene.
int tff_linear_tft_pose_sampled_dev(tff_ctx* c, const double* scene, int32_t Ns, const double* calm, const int32_t* sample_idx, int64_t B,
                                    int32_t n, double* Rt2, double* Rt3, double* T, int32_t* status) {
    if (!c) return fail(TFF_E_INVALID, "null context");
    if (!sample_idx || Ns <= 0) return fail(TFF_E_INVALID, "null sample indices / empty scene");
    TFF_LOCK(c);
    c->sample_idx = sample_idx; c->sample_ns = Ns;
    const int r = launch_linear_tft(c, scene, calm, 0, B, n, Rt2, Rt3, T, nullptr, nullptr, status, nullptr);
    c->sample_idx = nullptr; c->sample_ns = 0;
    return r;
}
int tff_linear_f_pose_sampled_dev(tff_ctx* c, const double* scene, int32_t Ns, const double* calm, const int32_t* sample_idx, int64_t B,
                                  int32_t n, double* Rt2, double* Rt3, double* T, int32_t* status) {
    if (!c) return fail(TFF_E_INVALID, "null context");
    if (!sample_idx || Ns <= 0) return fail(TFF_E_INVALID, "null sample indices / empty scene");
    TFF_LOCK(c);
    c->sample_idx = sample_idx; c->sample_ns = Ns;
    const int r = launch_linear_f(c, scene, calm, 0, B, n, Rt2, Rt3, T, nullptr, nullptr, status, nullptr);
    c->sample_idx = nullptr; c->sample_ns = 0;
    return r;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------
// Multi-GPU behind the C ABI: ONE process, one context + one host thread + one stream per
// device; the batch is cut into contiguous shards of ceil(B / G) triplets; independent triplets need no
// collective on the data path.  The _host variant lands every shard directly in the caller's host
// arrays.  The _dev variant leaves shard g on device g and then gathers the fixed-size result records
// of all shards onto every device with ONE ncclAllGather over xGMI (RCCL, single-process communicators
// from ncclCommInitAll; librccl.so is opened on first use, libtftfund.so itself does not depend on it).
// ---------------------------------------------------------------------------------------------
#include <dlfcn.h>
#include <thread>

struct tff_multi {
    std::vector<tff_ctx*> ctx;
    std::vector<int> devices;
    // RCCL (lazily): communicators of the single-process clique, one per device
    void* rccl = nullptr;
    std::mutex rccl_mu;
    std::vector<void*> comms;
    int (*p_init_all)(void**, int, const int*) = nullptr;
    int (*p_allgather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
    int (*p_group_start)() = nullptr;
    int (*p_group_end)() = nullptr;
    int (*p_comm_destroy)(void*) = nullptr;
    const char* (*p_err)(int) = nullptr;
};

namespace {

pose_launcher method_launcher(int32_t method) {
    switch (method) {
        case TFF_METHOD_LINEAR_TFT: return launch_linear_tft;
        case TFF_METHOD_RESSL_TFT: return launch_ressl_tft;
        case TFF_METHOD_NORDBERG_TFT: return launch_nordberg_tft;
        case TFF_METHOD_FAUGPAPA_TFT: return launch_faugpapa_tft;
        case TFF_METHOD_PI: return launch_pi;
        case TFF_METHOD_PICOL: return launch_picol;
        case TFF_METHOD_LINEAR_F: return launch_linear_f;
        case TFF_METHOD_OPTIM_F: return launch_optim_f;
        default: return nullptr;
    }
}

int multi_load_rccl(tff_multi* m) {
    std::lock_guard<std::mutex> guard(m->rccl_mu);                             // concurrent first calls: one of them initialises the communicators
    if (m->rccl) return 0;
    void* h = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
    if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!h) h = dlopen("/opt/rocm/lib/librccl.so", RTLD_NOW | RTLD_LOCAL);
    if (!h) return fail(TFF_E_INVALID, "librccl.so not found (needed only by tff_pose_batch_dev_multi)");
    m->p_init_all = (int (*)(void**, int, const int*))dlsym(h, "ncclCommInitAll");
    m->p_allgather = (int (*)(const void*, void*, size_t, int, void*, hipStream_t))dlsym(h, "ncclAllGather");
    m->p_group_start = (int (*)())dlsym(h, "ncclGroupStart");
    m->p_group_end = (int (*)())dlsym(h, "ncclGroupEnd");
    m->p_comm_destroy = (int (*)(void*))dlsym(h, "ncclCommDestroy");
    m->p_err = (const char* (*)(int))dlsym(h, "ncclGetErrorString");
    if (!m->p_init_all || !m->p_allgather || !m->p_group_start || !m->p_group_end || !m->p_comm_destroy) {
        dlclose(h);
        return fail(TFF_E_INVALID, "librccl.so lacks the expected symbols");
    }
    m->comms.assign(m->ctx.size(), nullptr);
    const int rc = m->p_init_all(m->comms.data(), (int)m->devices.size(), m->devices.data());
    if (rc != 0) {
        g_err = std::string("ncclCommInitAll: ") + (m->p_err ? m->p_err(rc) : "error");
        dlclose(h);
        m->comms.clear();
        return TFF_E_INVALID;
    }
    m->rccl = h;
    return 0;
}

}  // namespace

extern "C" {

int tff_multi_create(tff_multi** out, const int32_t* devices, int32_t n_devices) {
    if (!out) return fail(TFF_E_INVALID, "null out pointer");
    *out = nullptr;
    int have = 0;
    TFF_HIP(hipGetDeviceCount(&have));
    if (n_devices <= 0) n_devices = have;                                    // all visible devices
    if (n_devices <= 0 || n_devices > have) return fail(TFF_E_INVALID, "no such set of HIP devices");
    tff_multi* m = new (std::nothrow) tff_multi();
    if (!m) return fail(TFF_E_NOMEM, "out of host memory");
    for (int g = 0; g < n_devices; ++g) {
        const int dev = devices ? devices[g] : g;
        for (int d : m->devices) if (d == dev) { tff_multi_destroy(m); return fail(TFF_E_INVALID, "duplicate device"); }
        tff_ctx* c = nullptr;
        const int rc = tff_ctx_create(&c, dev);
        if (rc != 0) { tff_multi_destroy(m); return rc; }
        m->ctx.push_back(c);
        m->devices.push_back(dev);
    }
    *out = m;
    return 0;
}

void tff_multi_destroy(tff_multi* m) {
    if (!m) return;
    if (m->rccl) {
        for (void* cm : m->comms) if (cm) (void)m->p_comm_destroy(cm);
        dlclose(m->rccl);
    }
    for (tff_ctx* c : m->ctx) tff_ctx_destroy(c);
    delete m;
}

int32_t tff_multi_size(const tff_multi* m) { return m ? (int32_t)m->ctx.size() : 0; }
tff_ctx* tff_multi_ctx(tff_multi* m, int32_t rank) { return (m && rank >= 0 && rank < (int32_t)m->ctx.size()) ? m->ctx[rank] : nullptr; }

void tff_multi_shard(const tff_multi* m, int64_t B, int32_t rank, int64_t* begin, int64_t* end) {
    const int64_t G = m ? (int64_t)m->ctx.size() : 1;
    const int64_t chunk = (B + G - 1) / G;
    int64_t b0 = chunk * rank, b1 = b0 + chunk;
    if (b0 > B) b0 = B;
    if (b1 > B) b1 = B;
    if (begin) *begin = b0;
    if (end) *end = b1;
}

int tff_pose_batch_host_multi(tff_multi* m, int32_t method, const double* corresp, const double* calm, int64_t calm_stride,
                              int64_t B, int32_t N, double* Rt2, double* Rt3, double* T, double* reconst, int32_t* iter,
                              int32_t* status) {
    if (!m) return fail(TFF_E_INVALID, "null multi-GPU handle");
    pose_launcher launch = method_launcher(method);
    if (!launch) return fail(TFF_E_INVALID, "unknown method id");
    if (B < 0 || N < 0) return fail(TFF_E_INVALID, "negative batch or correspondence count");
    if (calm_stride != 0 && calm_stride != 27) return fail(TFF_E_INVALID, "calm_stride must be 0 (shared CalM) or 27");
    const int G = (int)m->ctx.size();
    std::vector<int> rc(G, 0);
    std::vector<std::string> msg(G);
    std::vector<std::thread> th;
    for (int g = 0; g < G; ++g) {
        th.emplace_back([&, g]() {
            int64_t b0, b1;
            tff_multi_shard(m, B, g, &b0, &b1);
            if (b1 <= b0) return;
            rc[g] = pose_batch_host(launch, m->ctx[g], corresp + b0 * 6 * (int64_t)N, calm + b0 * calm_stride, calm_stride, b1 - b0, N,
                                    Rt2 + b0 * 12, Rt3 + b0 * 12, T + b0 * 27, reconst ? reconst + b0 * 3 * (int64_t)N : nullptr,
                                    iter ? iter + b0 : nullptr, status ? status + b0 : nullptr);
            if (rc[g] != 0) msg[g] = g_err;                                  // tff_last_error() is thread-local: carry it over
        });
    }
    for (auto& t : th) t.join();
    for (int g = 0; g < G; ++g) if (rc[g] != 0) { g_err = "device " + std::to_string(m->devices[g]) + ": " + msg[g]; return rc[g]; }
    return 0;
}

// Device-resident variant.  corresp[g], calm[g]: device pointers ON DEVICE g holding shard g of the batch (shard bounds:
// tff_multi_shard).  records[g]: device buffer on device g of G * chunk * 51 doubles, chunk = ceil(B / G); after the call
// EVERY device holds all shards: block r (chunk * 51 doubles) = [Rt2 (chunk x 12) | Rt3 (chunk x 12) | T (chunk x 27)] of
// shard r.  status[g] (optional): G * chunk int32 per device, gathered the same way.  Work is enqueued on each context's
// stream; tff_ctx_synchronize(tff_multi_ctx(m, g)) to wait.
int tff_pose_batch_dev_multi(tff_multi* m, int32_t method, const double* const* corresp, const double* const* calm,
                             int64_t calm_stride, int64_t B, int32_t N, double* const* records, int32_t* const* status) {
    if (!m) return fail(TFF_E_INVALID, "null multi-GPU handle");
    pose_launcher launch = method_launcher(method);
    if (!launch) return fail(TFF_E_INVALID, "unknown method id");
    if (!corresp || !calm || !records) return fail(TFF_E_INVALID, "null pointer array");
    if (B < 0 || N < 0) return fail(TFF_E_INVALID, "negative batch or correspondence count");
    if (calm_stride != 0 && calm_stride != 27) return fail(TFF_E_INVALID, "calm_stride must be 0 (shared CalM) or 27");
    if (B == 0) return 0;
    if (int r = multi_load_rccl(m)) return r;
    const int G = (int)m->ctx.size();
    const int64_t chunk = (B + G - 1) / G;
    int caller_device = 0;
    TFF_HIP(hipGetDevice(&caller_device));
    std::vector<int> rc(G, 0);
    std::vector<std::string> msg(G);
    std::vector<std::thread> th;
    for (int g = 0; g < G; ++g) {
        th.emplace_back([&, g]() {
            int64_t b0, b1;
            tff_multi_shard(m, B, g, &b0, &b1);
            double* blk = records[g] + (int64_t)g * chunk * 51;
            int32_t* sblk = status ? status[g] + (int64_t)g * chunk : nullptr;
            // The all-gather below sends the WHOLE block of every device: the part of it no triplet fills (an uneven last shard, or an
            // empty one when B < G) is defined first -- quiet NaN records with status TFF_ST_TOO_FEW -- on the stream the kernels use.
            if (b1 - b0 < chunk) {
                hipError_t e = hipSetDevice(m->devices[g]);
                if (e == hipSuccess) e = hipMemsetAsync(blk, 0xff, (size_t)chunk * 51 * sizeof(double), m->ctx[g]->stream);
                if (e == hipSuccess && sblk) e = hipMemsetD32Async((hipDeviceptr_t)sblk, TFF_ST_TOO_FEW, (size_t)chunk, m->ctx[g]->stream);
                if (e != hipSuccess) { rc[g] = hip_fail(e, "hipMemsetAsync(record block)"); msg[g] = g_err; return; }
            }
            if (b1 <= b0) return;
            rc[g] = launch(m->ctx[g], corresp[g], calm[g], calm_stride, b1 - b0, N, blk, blk + chunk * 12, blk + chunk * 24, nullptr, nullptr,
                           sblk, nullptr);
            if (rc[g] != 0) msg[g] = g_err;
        });
    }
    for (auto& t : th) t.join();
    for (int g = 0; g < G; ++g) if (rc[g] != 0) { g_err = "device " + std::to_string(m->devices[g]) + ": " + msg[g]; return rc[g]; }
    // one collective per result kind, all devices in one group (single-process clique)
    int nrc = m->p_group_start();
    hipError_t herr = hipSuccess;
    for (int g = 0; g < G && nrc == 0 && herr == hipSuccess; ++g) {
        herr = hipSetDevice(m->devices[g]);                                  // (no early return inside the group: it must be closed)
        if (herr != hipSuccess) break;
        nrc = m->p_allgather(records[g] + (int64_t)g * chunk * 51, records[g], (size_t)(chunk * 51), 8 /* ncclFloat64 */, m->comms[g], m->ctx[g]->stream);
        if (nrc == 0 && status) nrc = m->p_allgather(status[g] + (int64_t)g * chunk, status[g], (size_t)chunk, 2 /* ncclInt32 */, m->comms[g], m->ctx[g]->stream);
    }
    const int erc = m->p_group_end();
    (void)hipSetDevice(caller_device);                                       // the caller's current device is not ours to change
    if (herr != hipSuccess) return hip_fail(herr, "hipSetDevice");
    if (nrc == 0) nrc = erc;
    if (nrc != 0) { g_err = std::string("ncclAllGather: ") + (m->p_err ? m->p_err(nrc) : "error"); return TFF_E_INVALID; }
    return 0;
}

}  // extern "C"
