// TEST INFRASTRUCTURE ONLY: runs the HIP kernels of tft_vs_fund_amd/csrc on the
// CPU through tests/emu/hip_emu.h (one std::thread per lane) so that their
// logic is covered by `pytest -m "not gpu"` and can be run under sanitizers.
// Never linked into libtftfund.so.
#include <vector>
#include "../../tft_vs_fund_amd/csrc/launch.h"

// Optional cap on the grid of every launch below (0 = none): exercises the grid-stride loops, in which one block takes several
// batch items through the same LDS (stale state between items is what that catches).
static unsigned g_grid_cap = 0;
extern "C" void emu_set_grid_cap(int cap) { g_grid_cap = cap > 0 ? (unsigned)cap : 0u; }
static unsigned emu_rows_grid(long B) { const unsigned g = tff::rows_grid(B); return (g_grid_cap && g > g_grid_cap) ? g_grid_cap : g; }
static unsigned emu_grid(long B) { const unsigned g = tff::pose_grid(B); return (g_grid_cap && g > g_grid_cap) ? g_grid_cap : g; }

// Same two-pass structure as the C ABI: inverse-iteration kernel, then the
// Jacobi fix-up over ST_RETRY triplets (or Jacobi for everything with FLAG_JACOBI).
typedef size_t (*lds_fn)(int, int, bool);
template <class KMain, class KJac>
static int emu_pose(KMain kmain, KJac kjac, lds_fn ldsfn, bool may_stage, const double* corresp, const double* calm, long calm_stride, long B, int N,
                    int flags, double* Rt2, double* Rt3, double* T, double* reconst, int* iter, int* status, double* dbg,
                    double* init_p = nullptr, double* init_x = nullptr) {
    tff::LinearTftArgs a{corresp, calm, calm_stride, B, N, flags & ~tff::FLAG_JACOBI, Rt2, Rt3, T, reconst, iter, status, dbg, nullptr, init_p, init_x};
    if (reconst) a.flags |= tff::FLAG_RECONST;
    const bool all_jacobi = (flags & tff::FLAG_JACOBI) != 0 || N < tff::EXACT_BELOW_N;   // as the C ABI: minimal samples go to the exact kernel
    if (!all_jacobi) {
        if (may_stage) a.flags = tff::pose_auto_flags(N, a.flags, false);
        emu::launch(kmain, emu_grid(B), 64, ldsfn(N, a.flags, false), a);
        bool any = false;
        for (long b = 0; b < B; ++b) any = any || status[b] == tff::ST_RETRY;
        if (!any) return 0;
        a.flags |= tff::FLAG_ONLY_RETRY;
    }
    a.flags &= ~tff::FLAG_STAGE_LDS;
    if (may_stage) a.flags = tff::pose_auto_flags(N, a.flags, true);
    emu::launch(kjac, emu_grid(B), 64, ldsfn(N, a.flags, true), a);
    return all_jacobi ? 0 : 1;
}

extern "C" int emu_linear_tft_pose(const double* corresp, const double* calm, long calm_stride, long B, int N, int flags,
                                   double* Rt2, double* Rt3, double* T, double* reconst, int* iter, int* status, double* dbg) {
    return emu_pose(tff::k_linear_tft_pose<false>, tff::k_linear_tft_pose<true>, tff::pose_lds_bytes, true, corresp, calm, calm_stride, B, N, flags, Rt2, Rt3, T,
                    reconst, iter, status, dbg);
}
// the four-triplets-per-wavefront kernel (tft_rows_kernel.h), then the exact kernel over what it handed back
extern "C" int emu_linear_tft_pose_rows(const double* corresp, const double* calm, long calm_stride, long B, int N, int flags,
                                        double* Rt2, double* Rt3, double* T, double* reconst, int* iter, int* status, double* dbg) {
    tff::LinearTftArgs a{corresp, calm, calm_stride, B, N, flags & ~tff::FLAG_JACOBI, Rt2, Rt3, T, reconst, iter, status, dbg, nullptr, nullptr, nullptr};
    if (reconst) a.flags |= tff::FLAG_RECONST;
    std::vector<int> retry((size_t)B + 2, 0);                // [count | next call's count | indices]
    a.retry_count = retry.data(); a.retry_zero = retry.data() + 1; a.retry_list = retry.data() + 2;
    std::vector<double> pre;
    if (flags & 8192) {                                                      // test switch: moments + normalisations from k_tft_moments (as the C ABI does from N >= 48)
        pre.assign((size_t)B * tff::PRE_DOUBLES, 0.0);
        tff::MomentArgs m{corresp, B, N, pre.data()};
        const bool stage = !(flags & 16384);                                 // (16384: second pass from global memory, the large-N variant)
        if (stage) emu::launch(tff::k_tft_moments<true>, emu_grid(B), 64, tff::moments_lds_bytes(N, true), m);
        else emu::launch(tff::k_tft_moments<false>, emu_grid(B), 64, tff::moments_lds_bytes(N, false), m);
        a.pre = pre.data();
        a.flags &= ~(8192 | 16384);
        emu::launch(tff::k_linear_tft_pose_rows<true>, emu_rows_grid(B), 64, tff::rows_lds_bytes(), a);
        a.pre = nullptr;
    } else {
        emu::launch(tff::k_linear_tft_pose_rows<false>, emu_rows_grid(B), 64, tff::rows_lds_bytes(), a);
    }
    if (retry[0] == 0) return 0;                             // (as the C ABI: the row kernel has put the flagged triplets on the list)
    a.flags |= tff::FLAG_ONLY_RETRY;
    a.flags = tff::pose_auto_flags(N, a.flags, true);
    emu::launch(tff::k_linear_tft_pose<true>, emu_grid(B), 64, tff::pose_lds_bytes(N, a.flags, true), a);
    return 1;
}
// the exact tiers with four triplets per wavefront (tft_rows_exact_kernel.h), then the one-triplet exact kernel over what it handed back
extern "C" int emu_linear_tft_pose_rows_exact(const double* corresp, const double* calm, long calm_stride, long B, int N, int flags,
                                              double* Rt2, double* Rt3, double* T, double* reconst, int* iter, int* status, double* dbg) {
    tff::LinearTftArgs a{corresp, calm, calm_stride, B, N, flags & ~tff::FLAG_JACOBI, Rt2, Rt3, T, reconst, iter, status, dbg, nullptr, nullptr, nullptr};
    if (reconst) a.flags |= tff::FLAG_RECONST;
    emu::launch(tff::k_linear_tft_pose_rows_exact, emu_rows_grid(B), 64, tff::rows_lds_bytes(), a);
    bool any = false;
    for (long b = 0; b < B; ++b) any = any || status[b] == tff::ST_RETRY;
    if (!any) return 0;
    a.flags |= tff::FLAG_ONLY_RETRY;
    a.flags = tff::pose_auto_flags(N, a.flags, true);
    emu::launch(tff::k_linear_tft_pose<true>, emu_grid(B), 64, tff::pose_lds_bytes(N, a.flags, true), a);
    return 1;
}
// LinearF with four triplets per wavefront (f_rows_kernel.h), then the exact kernel over what it handed back
extern "C" int emu_linear_f_pose_rows(const double* corresp, const double* calm, long calm_stride, long B, int N, int flags,
                                      double* Rt2, double* Rt3, double* T, double* reconst, int* iter, int* status, double* dbg) {
    tff::LinearTftArgs a{corresp, calm, calm_stride, B, N, flags & ~tff::FLAG_JACOBI, Rt2, Rt3, T, reconst, iter, status, dbg, nullptr, nullptr, nullptr};
    if (reconst) a.flags |= tff::FLAG_RECONST;
    emu::launch(tff::k_linear_f_pose_rows, emu_rows_grid(B), 64, tff::rows_lds_bytes(), a);
    bool any = false;
    for (long b = 0; b < B; ++b) any = any || status[b] == tff::ST_RETRY;
    if (!any) return 0;
    a.flags |= tff::FLAG_ONLY_RETRY;
    a.flags = tff::pose_auto_flags(N, a.flags, true, tff::STAGE_MAX_N_F);
    emu::launch(tff::k_f_pose<true, 0>, emu_grid(B), 64, tff::f_pose_lds_bytes(N, a.flags, true), a);
    return 1;
}
// rows_qr.h: streaming Householder QR + inverse iteration in the row layout, four systems per wavefront.  A: B x rows x n (row-major) ->
// x: B x n (right singular vector of the smallest singular value), its, conv (1 = converged)
namespace {
struct RowsQrArgs { const double* A; long B; int rows; double* x; int* its; int* conv; double* Rout; };
template <int n, int M>
__global__ void k_emu_rows_qr(RowsQrArgs a) {
    TFF_DYNAMIC_LDS(double, lds);
    const int lane = tff::lane_id(), p = lane & 15, row = lane >> 4;
    constexpr int RP = tff::rows_up_doubles<n>();
    double* Rp = lds + row * (RP + M + n + 3);
    double* xch = Rp + RP; double* dinv = xch + M;
    long b = (long)blockIdx.x * 4 + row;
    if (b >= a.B) b = a.B - 1;
    tff::rows_qr_clear<n>(Rp);
    for (int base = 0; base < a.rows; base += M) {
        double a0[M], a1[M];
        for (int i = 0; i < M; ++i) {
            const bool have = base + i < a.rows;
            a0[i] = (have && p < n) ? a.A[(b * a.rows + base + i) * n + p] : 0.0;
            a1[i] = (have && 16 + p < n) ? a.A[(b * a.rows + base + i) * n + 16 + p] : 0.0;
        }
        tff::rows_qr_append<n, M>(a0, a1, Rp, xch);
    }
    if (a.Rout && (long)blockIdx.x * 4 + row < a.B) for (int e = p; e < RP; e += 16) a.Rout[b * RP + e] = Rp[e];
    int its; double r2, x0, x1;
    tff::rows_invit_from_R<n>(Rp, dinv, 300, &its, &r2, x0, x1);
    if ((long)blockIdx.x * 4 + row < a.B) {
        if (p < n) a.x[b * n + p] = x0;
        if (16 + p < n) a.x[b * n + 16 + p] = x1;
        if (p == 0) { a.its[b] = its; a.conv[b] = (r2 == 0.0) ? 1 : 0; }
    }
}
}
extern "C" int emu_rows_qr(const double* A, long B, int rows, int n, double* x, int* its, int* conv, double* Rout) {
    RowsQrArgs a{A, B, rows, x, its, conv, Rout};
    const unsigned grid = (unsigned)((B + 3) / 4);
    if (n == 27) emu::launch(k_emu_rows_qr<27, 28>, grid, 64, sizeof(double) * 4 * (tff::rows_up_doubles<27>() + 28 + 27 + 3), a);
    else if (n == 15) emu::launch(k_emu_rows_qr<15, 27>, grid, 64, sizeof(double) * 4 * (tff::rows_up_doubles<15>() + 27 + 15 + 3), a);
    else if (n == 9) emu::launch(k_emu_rows_qr<9, 16>, grid, 64, sizeof(double) * 4 * (tff::rows_up_doubles<9>() + 16 + 9 + 3), a);
    else return -1;
    return 0;
}
extern "C" int emu_linear_f_pose_rows_exact(const double* corresp, const double* calm, long calm_stride, long B, int N, int flags,
                                            double* Rt2, double* Rt3, double* T, double* reconst, int* iter, int* status, double* dbg) {
    tff::LinearTftArgs a{corresp, calm, calm_stride, B, N, flags & ~tff::FLAG_JACOBI, Rt2, Rt3, T, reconst, iter, status, dbg, nullptr, nullptr, nullptr};
    if (reconst) a.flags |= tff::FLAG_RECONST;
    emu::launch(tff::k_linear_f_pose_rows_exact, emu_rows_grid(B), 64, tff::rows_lds_bytes(), a);
    bool any = false;
    for (long b = 0; b < B; ++b) any = any || status[b] == tff::ST_RETRY;
    if (!any) return 0;
    a.flags |= tff::FLAG_ONLY_RETRY;
    a.flags = tff::pose_auto_flags(N, a.flags, true, tff::STAGE_MAX_N_F);
    emu::launch(tff::k_f_pose<true, 0>, emu_grid(B), 64, tff::f_pose_lds_bytes(N, a.flags, true), a);
    return 1;
}
#ifndef TFF_EMU_LINEAR_TFT_ONLY   // (the sanitizer build of tests/test_emulated_kernels.py compiles the linear trifocal kernels only: minutes less)
extern "C" int emu_linear_f_pose(const double* corresp, const double* calm, long calm_stride, long B, int N, int flags,
                                 double* Rt2, double* Rt3, double* T, double* reconst, int* iter, int* status, double* dbg) {
    return emu_pose(tff::k_f_pose<false, 0>, tff::k_f_pose<true, 0>, tff::f_pose_lds_bytes, true, corresp, calm, calm_stride, B, N, flags, Rt2, Rt3, T,
                    reconst, iter, status, dbg);
}
// OptimFPoseEstimation as the library runs large batches (capi.hip::launch_optim_f): linear stage and pose tail four triplets per wavefront,
// Gauss-Helmert refinement one wavefront per triplet, exact kernel over what they handed on
extern "C" int emu_optim_f_pose_staged(const double* corresp, const double* calm, long calm_stride, long B, int N, int flags,
                                       double* Rt2, double* Rt3, double* T, double* reconst, int* iter, int* status, double* dbg) {
    (void)dbg;
    std::vector<double> rec((size_t)B * tff::OPTIMF_REC_DOUBLES, 0.0);
    tff::OptimFStageArgs sa{};
    sa.la = tff::LinearTftArgs{corresp, calm, calm_stride, B, N, flags & ~tff::FLAG_JACOBI, Rt2, Rt3, T, reconst, iter, status, nullptr, nullptr, nullptr, nullptr};
    if (reconst) sa.la.flags |= tff::FLAG_RECONST;
    sa.rec = rec.data();
    emu::launch(tff::k_optimf_linear_rows, emu_rows_grid(B), 64, tff::rows_lds_bytes(), sa);
    if (N <= 64) emu::launch(tff::k_optimf_refine<tff::OPTIMF_REFINE_WAVES, true>, emu_grid(B), 64, tff::optimf_refine_lds_bytes(N, true), sa);
    else emu::launch(tff::k_optimf_refine<tff::OPTIMF_REFINE_WAVES, false>, emu_grid(B), 64, tff::optimf_refine_lds_bytes(N, false), sa);   // (both builds are exercised: the golden cases have N = 100)
    emu::launch(tff::k_optimf_finish_rows, emu_rows_grid(B), 64, tff::rows_lds_bytes(), sa);
    bool any = false;
    for (long b = 0; b < B; ++b) any = any || status[b] == tff::ST_RETRY;
    if (!any) return 0;
    tff::LinearTftArgs a = sa.la;
    a.flags |= tff::FLAG_ONLY_RETRY;
    emu::launch(tff::k_f_pose<true, 1>, emu_grid(B), 64, tff::optimf_lds_bytes(N, a.flags, true), a);
    return 1;
}
extern "C" int emu_optim_f_pose(const double* corresp, const double* calm, long calm_stride, long B, int N, int flags,
                                double* Rt2, double* Rt3, double* T, double* reconst, int* iter, int* status, double* dbg) {
    return emu_pose(tff::k_f_pose<false, 1>, tff::k_f_pose<true, 1>, tff::optimf_lds_bytes, false, corresp, calm, calm_stride, B, N, flags, Rt2, Rt3, T,
                    reconst, iter, status, dbg);
}
extern "C" int emu_pi_pose(const double* corresp, const double* calm, long calm_stride, long B, int N, int flags,
                           double* Rt2, double* Rt3, double* T, double* reconst, int* iter, int* status, double* dbg) {
    return emu_pose(tff::k_pi_tft_pose<tff::PiModel, false>, tff::k_pi_tft_pose<tff::PiModel, true>, tff::pi_lds_bytes<tff::PiModel>, false,
                    corresp, calm, calm_stride, B, N, flags, Rt2, Rt3, T, reconst, iter, status, dbg);
}
extern "C" int emu_picol_pose(const double* corresp, const double* calm, long calm_stride, long B, int N, int flags,
                              double* Rt2, double* Rt3, double* T, double* reconst, int* iter, int* status, double* dbg) {
    return emu_pose(tff::k_pi_tft_pose<tff::PiColModel, false>, tff::k_pi_tft_pose<tff::PiColModel, true>, tff::pi_lds_bytes<tff::PiColModel>, false,
                    corresp, calm, calm_stride, B, N, flags, Rt2, Rt3, T, reconst, iter, status, dbg);
}
// Pi / PiCol with the start of the Gauss-Helmert iteration exposed (tff_pi_pose_batch_debug_dev)
extern "C" int emu_pi_pose_debug(int collinear, const double* corresp, const double* calm, long calm_stride, long B, int N, int flags,
                                 double* Rt2, double* Rt3, double* T, double* reconst, int* iter, int* status, double* init_p, double* init_x) {
    if (collinear)
        return emu_pose(tff::k_pi_tft_pose<tff::PiColModel, false>, tff::k_pi_tft_pose<tff::PiColModel, true>, tff::pi_lds_bytes<tff::PiColModel>, false,
                        corresp, calm, calm_stride, B, N, flags, Rt2, Rt3, T, reconst, iter, status, nullptr, init_p, init_x);
    return emu_pose(tff::k_pi_tft_pose<tff::PiModel, false>, tff::k_pi_tft_pose<tff::PiModel, true>, tff::pi_lds_bytes<tff::PiModel>, false,
                    corresp, calm, calm_stride, B, N, flags, Rt2, Rt3, T, reconst, iter, status, nullptr, init_p, init_x);
}
// Gauss-Helmert methods through the three-launch workgroup path (gh_wg_kernel.h): model 0 Ressl, 1 Nordberg, 2 FaugPapa
template <class KBlock>
static int emu_wg_run(KBlock kblock, int block_threads, size_t lds_block, const double* corresp, const double* calm, long calm_stride, long B, int N, int flags,
                      double* Rt2, double* Rt3, double* T, double* reconst, int* iter, int* status) {
    std::vector<double> rec((size_t)B * tff::GH_REC_DOUBLES), topt((size_t)B * 27);
    tff::GhWgArgs a{corresp, calm, calm_stride, B, N, (flags & ~tff::FLAG_JACOBI) | (reconst ? tff::FLAG_RECONST : 0), rec.data(), topt.data(),
                    Rt2, Rt3, T, reconst, iter, status, nullptr, nullptr, 0};
    emu::launch(tff::k_gh_linear_rows<false>, emu_rows_grid(B), 64, tff::rows_lds_bytes(), a);     // (as the C ABI: four triplets per wavefront)
    tff::GhWgArgs m = a;
    m.flags |= tff::FLAG_ONLY_RETRY;
    emu::launch(tff::k_gh_linear<true>, emu_grid(B), 64, tff::pose_lds_bytes(N, m.flags, true), m);
    emu::launch(kblock, emu_grid(B), block_threads, lds_block, a);
    if (N >= 12) emu::launch(tff::k_gh_finish_rows, emu_rows_grid(B), 64, tff::rows_lds_bytes(), a);     // (as the C ABI)
    else emu::launch(tff::k_gh_finish, emu_grid(B), 64, tff::pose_lds_bytes(N, 0, false), a);
    return 0;
}
// Pi / PiCol through the workgroup path (pi_wg_kernel.h)
extern "C" int emu_pi_wg_pose(int collinear, const double* corresp, const double* calm, long calm_stride, long B, int N, int flags,
                              double* Rt2, double* Rt3, double* T, double* reconst, int* iter, int* status) {
    const size_t pre = (size_t)((tff::POSE_LDS_DOUBLES + 1) & ~1);
    if (collinear)
        return emu_wg_run(tff::k_pi_block<tff::PiColModel>, tff::pi_wg_waves<tff::PiColModel>::value * tff::WAVE, (pre + tff::pi_wg_lds_doubles(tff::PiColModel::E, tff::PiColModel::C, N)) * sizeof(double),
                          corresp, calm, calm_stride, B, N, flags, Rt2, Rt3, T, reconst, iter, status);
    return emu_wg_run(tff::k_pi_block<tff::PiModel>, tff::pi_wg_waves<tff::PiModel>::value * tff::WAVE, (pre + tff::pi_wg_lds_doubles(tff::PiModel::E, tff::PiModel::C, N)) * sizeof(double),
                      corresp, calm, calm_stride, B, N, flags, Rt2, Rt3, T, reconst, iter, status);
}
template <class Model>
static int emu_gh_wg_impl(const double* corresp, const double* calm, long calm_stride, long B, int N, int flags,
                          double* Rt2, double* Rt3, double* T, double* reconst, int* iter, int* status) {
    std::vector<double> rec((size_t)B * tff::GH_REC_DOUBLES), topt((size_t)B * 27);
    tff::GhWgArgs a{corresp, calm, calm_stride, B, N, (flags & ~tff::FLAG_JACOBI) | (reconst ? tff::FLAG_RECONST : 0), rec.data(), topt.data(),
                    Rt2, Rt3, T, reconst, iter, status, nullptr, nullptr, 0};
    emu::launch(tff::k_gh_linear_rows<false>, emu_rows_grid(B), 64, tff::rows_lds_bytes(), a);     // (as the C ABI: four triplets per wavefront)
    tff::GhWgArgs m = a;
    m.flags |= tff::FLAG_ONLY_RETRY;
    emu::launch(tff::k_gh_linear<true>, emu_grid(B), 64, tff::pose_lds_bytes(N, m.flags, true), m);
    const size_t lds = (size_t)(((tff::POSE_LDS_DOUBLES + 1) & ~1) + tff::gh_wg_lds_doubles(Model::U, Model::C, N, Model::REDUNDANT_CONSTRAINTS)) * sizeof(double);
    std::vector<double> pre;
    if (tff::gh_has_preinit<Model>::value && !(flags & 4096)) {              // (as the C ABI; flag 4096, a test switch: the block kernel computes it itself)
        pre.assign((size_t)B * 64, 0.0);
        a.init_rec = pre.data();
        emu::launch(tff::k_nordberg_init, (unsigned)((B + 63) / 64), 64, 0, a);
    }
    emu::launch(tff::k_gh_block<Model>, emu_grid(B), tff::gh_wg_waves<Model>::value * tff::WAVE, lds, a);
    if (N >= 12) emu::launch(tff::k_gh_finish_rows, emu_rows_grid(B), 64, tff::rows_lds_bytes(), a);     // (as the C ABI)
    else emu::launch(tff::k_gh_finish, emu_grid(B), 64, tff::pose_lds_bytes(N, 0, false), a);
    return 0;
}
// FaugPapa through its own block kernel (gh_fp_kernel.h), then the generic one over what it handed back
extern "C" int emu_fp_pose(const double* corresp, const double* calm, long calm_stride, long B, int N, int flags,
                           double* Rt2, double* Rt3, double* T, double* reconst, int* iter, int* status) {
    std::vector<double> rec((size_t)B * tff::GH_REC_DOUBLES), topt((size_t)B * 27);
    tff::GhWgArgs a{corresp, calm, calm_stride, B, N, (flags & ~tff::FLAG_JACOBI) | (reconst ? tff::FLAG_RECONST : 0), rec.data(), topt.data(),
                    Rt2, Rt3, T, reconst, iter, status, nullptr, nullptr, 0};
    emu::launch(tff::k_gh_linear_rows<false>, emu_rows_grid(B), 64, tff::rows_lds_bytes(), a);     // (as the C ABI: four triplets per wavefront)
    tff::GhWgArgs m = a;
    m.flags |= tff::FLAG_ONLY_RETRY;
    emu::launch(tff::k_gh_linear<true>, emu_grid(B), 64, tff::pose_lds_bytes(N, m.flags, true), m);
    emu::launch(tff::k_fp_block<true>, emu_grid(B), tff::FP_THREADS, tff::fp_lds_bytes(N), a);
    int handed = 0;
    for (long b = 0; b < B; ++b) handed += status[b] == tff::ST_RETRY;
    if (handed) {
        using Model = tff::FaugPapaModel;
        const size_t lds = (size_t)(((tff::POSE_LDS_DOUBLES + 1) & ~1) + tff::gh_wg_lds_doubles(Model::U, Model::C, N, Model::REDUNDANT_CONSTRAINTS)) * sizeof(double);
        emu::launch(tff::k_gh_block<Model>, emu_grid(B), tff::GH_WG_THREADS, lds, m);
    }
    if (N >= 12) emu::launch(tff::k_gh_finish_rows, emu_rows_grid(B), 64, tff::rows_lds_bytes(), a);     // (as the C ABI)
    else emu::launch(tff::k_gh_finish, emu_grid(B), 64, tff::pose_lds_bytes(N, 0, false), a);
    return handed;
}
extern "C" int emu_gh_wg_pose(int model, const double* corresp, const double* calm, long calm_stride, long B, int N, int flags,
                              double* Rt2, double* Rt3, double* T, double* reconst, int* iter, int* status) {
    if (model == 0) return emu_gh_wg_impl<tff::ResslModel>(corresp, calm, calm_stride, B, N, flags, Rt2, Rt3, T, reconst, iter, status);
    if (model == 1) return emu_gh_wg_impl<tff::NordbergModel>(corresp, calm, calm_stride, B, N, flags, Rt2, Rt3, T, reconst, iter, status);
    return emu_gh_wg_impl<tff::FaugPapaModel>(corresp, calm, calm_stride, B, N, flags, Rt2, Rt3, T, reconst, iter, status);
}
extern "C" int emu_bundle_adjust(const double* calm, long calm_stride, const double* Rt2_in, const double* Rt3_in, const double* corresp, long B, int N,
                                 const double* reconst0, double* Rt2, double* Rt3, double* reconst, int* iter, double* repr_err, int* status) {
    tff::BaArgs a{calm, calm_stride, Rt2_in, Rt3_in, corresp, B, N, reconst0, Rt2, Rt3, reconst, iter, repr_err, status};
    emu::launch(tff::k_bundle_adjust, emu_grid(B), 64, tff::ba_lds_bytes(N), a);
    return 0;
}
template <int M> static void emu_bav(const tff::BavArgs& a) { emu::launch(tff::k_bundle_adjust_views<M>, emu_grid(a.B), 64, tff::bav_lds_bytes<M>(a.N), a); }
extern "C" int emu_bundle_adjust_views(int M, const double* calm, long calm_stride, const double* Rt_in, const double* corresp, long B, int N,
                                       const double* reconst0, double* Rt, double* reconst, int* iter, double* repr_err, int* status) {
    const tff::BavArgs a{calm, calm_stride, Rt_in, corresp, B, N, reconst0, Rt, reconst, iter, repr_err, status};
    switch (M) {
        case 2: emu_bav<2>(a); break;
        case 3: emu_bav<3>(a); break;
        case 4: emu_bav<4>(a); break;
        case 5: emu_bav<5>(a); break;
        case 6: emu_bav<6>(a); break;
        default: return -1;
    }
    return 0;
}
// building block: linearF / optimF per view pair (tff_linear_f_batch_dev)
extern "C" int emu_linear_f(const double* corresp, long B, int N, int refine, double* F21, double* F31, int* iter, int* status) {
    tff::LinearFOnlyArgs a{corresp, B, N, 0, F21, F31, iter, status};
    if (refine) {
        emu::launch(tff::k_linear_f<false, 1>, emu_grid(B), 64, tff::optimf_lds_bytes(N, 0, false), a);
        a.flags |= tff::FLAG_ONLY_RETRY;
        emu::launch(tff::k_linear_f<true, 1>, emu_grid(B), 64, tff::optimf_lds_bytes(N, 0, true), a);
    } else {
        emu::launch(tff::k_linear_f<false, 0>, emu_grid(B), 64, tff::f_pose_lds_bytes(N, 0, false), a);
        a.flags |= tff::FLAG_ONLY_RETRY;
        emu::launch(tff::k_linear_f<true, 0>, emu_grid(B), 64, tff::f_pose_lds_bytes(N, 0, true), a);
    }
    return 0;
}
extern "C" int emu_ressl_tft_pose(const double* corresp, const double* calm, long calm_stride, long B, int N, int flags,
                                  double* Rt2, double* Rt3, double* T, double* reconst, int* iter, int* status, double* dbg) {
    return emu_pose(tff::k_gh_tft_pose<tff::ResslModel, false>, tff::k_gh_tft_pose<tff::ResslModel, true>, tff::gh_lds_bytes<tff::ResslModel>, false, corresp, calm, calm_stride, B, N,
                    flags, Rt2, Rt3, T, reconst, iter, status, dbg);
}
extern "C" int emu_faugpapa_tft_pose(const double* corresp, const double* calm, long calm_stride, long B, int N, int flags,
                                     double* Rt2, double* Rt3, double* T, double* reconst, int* iter, int* status, double* dbg) {
    return emu_pose(tff::k_gh_tft_pose<tff::FaugPapaModel, false>, tff::k_gh_tft_pose<tff::FaugPapaModel, true>,
                    tff::gh_lds_bytes<tff::FaugPapaModel>, false, corresp, calm, calm_stride, B, N, flags, Rt2, Rt3, T, reconst, iter, status, dbg);
}
extern "C" int emu_nordberg_tft_pose(const double* corresp, const double* calm, long calm_stride, long B, int N, int flags,
                                     double* Rt2, double* Rt3, double* T, double* reconst, int* iter, int* status, double* dbg) {
    return emu_pose(tff::k_gh_tft_pose<tff::NordbergModel, false>, tff::k_gh_tft_pose<tff::NordbergModel, true>,
                    tff::gh_lds_bytes<tff::NordbergModel>, false, corresp, calm, calm_stride, B, N, flags, Rt2, Rt3, T, reconst, iter, status, dbg);
}

// wave_eigh_ql / wave_pinv_solve_sym on caller-supplied symmetric matrices (one wavefront per matrix):
// Maug B x n x (n+1) row-major (column n = right-hand side) -> lam B x n, vecs B x n x n (row j = eigenvector of lam[j]), sol B x n
namespace {
struct EighArgs { const double* Maug; int n; double* lam; double* vecs; double* sol; };
__global__ void k_emu_eigh(EighArgs a) {
    TFF_DYNAMIC_LDS(double, lds);
    const int n = a.n, ld = n + 1, lane = tff::lane_id();
    const long b = blockIdx.x;
    double* M = lds; double* ZT = M + n * ld; double* scr = ZT + n * n; double* M2 = scr + 2 * n; double* sol = M2 + n * ld;
    for (int e = lane; e < n * ld; e += 64) { M[e] = a.Maug[b * n * ld + e]; M2[e] = M[e]; }
    tff::wave_sync();
    int fail;
    const double lam = tff::wave_eigh_ql(M, ld, ZT, n, n, scr, &fail);
    if (lane < n) {
        a.lam[b * n + lane] = fail ? 0.0 / 0.0 : lam;
        for (int r = 0; r < n; ++r) a.vecs[(b * n + lane) * n + r] = ZT[tff::eig_row(n, lane) * n + r];
    }
    tff::wave_sync();
    tff::wave_pinv_solve_sym(M2, ZT, n, sol, scr);
    if (lane < n) a.sol[b * n + lane] = sol[lane];
}
}
extern "C" int emu_eigh(const double* Maug, long B, int n, double* lam, double* vecs, double* sol) {
    EighArgs a{Maug, n, lam, vecs, sol};
    emu::launch(k_emu_eigh, emu_grid(B), 64, sizeof(double) * (size_t)(2 * n * (n + 1) + n * n + 3 * n), a);
    return 0;
}

// wave_pinv_solve_trid (wave_trid.h) on caller-supplied symmetric matrices, n <= 32: Maug B x n x (n+1) row-major (column n = right-hand
// side), tol B -> sol B x n, kept B, fail B
namespace {
struct TridArgs { const double* Maug; const double* tol; int n; double* sol; int* kept; int* fail; int variant; };
__global__ void k_emu_trid(TridArgs a) {
    TFF_DYNAMIC_LDS(double, lds);
    const int n = a.n, ld = n + 1, lane = tff::lane_id();
    const long b = blockIdx.x;
    double* M = lds; double* small = M + n * ld + 2; double* work = small + tff::TRID_SMALL_DOUBLES; double* sol = work + tff::TRID_WORK_DOUBLES;
    for (int e = lane; e < n * ld; e += 64) M[e] = a.Maug[b * n * ld + e];
    tff::wave_sync();
    int kept, fail;
    if (a.variant == 2) tff::wave_pinv_solve_trid<true, true>(M, ld, n, a.tol[b], sol, small, work, &kept, &fail);       // registers + minor form
    else if (a.variant == 1) tff::wave_pinv_solve_trid<false, false>(M, ld, n, a.tol[b], sol, small, work, &kept, &fail);   // LDS bursts + pivot form
    else tff::wave_pinv_solve_trid<true, false>(M, ld, n, a.tol[b], sol, small, work, &kept, &fail);                      // registers + pivot form
    if (lane < n) a.sol[b * n + lane] = sol[lane];
    if (lane == 0) { a.kept[b] = kept; a.fail[b] = fail; }
}
}
extern "C" int emu_trid_pinv(const double* Maug, const double* tol, long B, int n, double* sol, int* kept, int* fail, int variant) {
    TridArgs a{Maug, tol, n, sol, kept, fail, variant};
    emu::launch(k_emu_trid, emu_grid(B), 64, sizeof(double) * (size_t)(n * (n + 1) + 2 + tff::TRID_SMALL_DOUBLES + tff::TRID_WORK_DOUBLES + 64), a);
    return 0;
}

// row_min_eigvec<n> (csrc/row_eig.h: Cholesky + inverse iteration with DP-ALU DPP row_newbcast operands, two matrix rows per position of a
// row of 16 lanes) and wave_min_eigvec_reg<n> (csrc/wave_eig.h: the v_readlane form it replaced, one matrix row per lane) on the same
// symmetric positive semi-definite matrices G (B x n x n, row-major): smallest eigenvectors (B x n), iteration counts, convergence flags.
namespace {
struct RowEigArgs { const double* G; int n; double* x_row; double* x_lane; int* its_row; int* its_lane; int* conv_row; int* conv_lane; };
template <int n>
__device__ void emu_row_eig_one(const RowEigArgs& a, double* Lp) {
    const int b = blockIdx.x, lane = tff::lane_id(), p = lane & 15;
    const double* G = a.G + (long)b * n * n;
    {
        constexpr int N0 = tff::RowEigDims<n>::N0, N1 = tff::RowEigDims<n>::N1;
        double g0[N0], g1[N1], d0 = 0.0, d1 = 0.0;
        const bool v0 = p < n, v1 = n > 16 && 16 + p < n;
        for (int c = 0; c < N0; ++c) g0[c] = v0 ? G[(v0 ? p : 0) * n + c] : 0.0;
        for (int c = 0; c < N1; ++c) g1[c] = (v1 && c < n) ? G[(16 + p) * n + c] : 0.0;
        if (v0) d0 = G[p * n + p];
        if (v1) d1 = G[(16 + p) * n + 16 + p];
        int it = 0; double r2 = 1.0, risk = 0.0;
        const double x = tff::row_min_eigvec<n>(g0, g1, d0, d1, Lp, 40, &it, &r2, false, 0.0, 0.0, &risk);
        if (lane < n) a.x_row[(long)b * n + lane] = x;
        if (lane == 0) { a.its_row[b] = it; a.conv_row[b] = (r2 == 0.0) ? 1 : 0; }
    }
    tff::wave_sync();
    {
        double g[n], diag = 0.0;
        const int r = (lane < n) ? lane : 0;
        for (int c = 0; c < n; ++c) { g[c] = G[r * n + c]; if (c == r) diag = g[c]; }
        int it = 0; double r2 = 1.0, risk = 0.0;
        const double x = tff::wave_min_eigvec_reg<n>(g, diag, Lp, 40, &it, &r2, false, 0.0, &risk);
        if (lane < n) a.x_lane[(long)b * n + lane] = x;
        if (lane == 0) { a.its_lane[b] = it; a.conv_lane[b] = (r2 == 0.0) ? 1 : 0; }
    }
}
__global__ void k_emu_row_eig(RowEigArgs a) {
    TFF_DYNAMIC_LDS(double, smem);
    if (a.n == 27) emu_row_eig_one<27>(a, smem);
    else if (a.n == 15) emu_row_eig_one<15>(a, smem);
    else if (a.n == 9) emu_row_eig_one<9>(a, smem);
    else if (a.n == 16) emu_row_eig_one<16>(a, smem);
    else if (a.n == 17) emu_row_eig_one<17>(a, smem);
    else emu_row_eig_one<32>(a, smem);
}
}  // namespace
extern "C" int emu_row_eig(const double* G, long B, int n, double* x_row, double* x_lane, int* its_row, int* its_lane, int* conv_row, int* conv_lane) {
    if (n != 27 && n != 15 && n != 9 && n != 16 && n != 17 && n != 32) return -1;
    RowEigArgs a{G, n, x_row, x_lane, its_row, its_lane, conv_row, conv_lane};
    emu::launch(k_emu_row_eig, (unsigned)B, 64, sizeof(double) * (size_t)(n * n + 64), a);
    return 0;
}

// NordbergModel::init on caller-supplied linearTFT output (t 27, a 18, epipoles 6): the initial 19 parameters and the
// rank flag -- exercises the projective fix-up for a rank-deficient P2(:,1:3) / P3(:,1:3) (NordbergTFTPoseEstimation.m:56-62),
// which no correspondence set reaches through the whole pipeline (it needs sigma_3 <= 3 eps(sigma_1) in the linear solution).
namespace {
struct NordInitArgs { const double* t; const double* pa; const double* epi; double* p; int* bad; };
__global__ void k_emu_nordberg_init(NordInitArgs a) {
    TFF_DYNAMIC_LDS(double, smem);
    tff::PoseLds* w = reinterpret_cast<tff::PoseLds*>(smem);
    const int lane = tff::lane_id();
    if (lane < 27) w->t[lane] = a.t[lane];
    if (lane < 18) w->pa[lane] = a.pa[lane];
    if (lane < 6) w->epi[lane] = a.epi[lane];
    tff::wave_sync();
    tff::GhWork g = tff::gh_carve(smem + ((tff::POSE_LDS_DOUBLES + 1) & ~1), tff::NordbergModel::U, tff::NordbergModel::C, 0);
    tff::NordbergModel model;
    model.init(w, g);
    if (lane < 19) a.p[lane] = g.p[lane];
    if (lane == 0) *a.bad = model.bad;
}
}  // namespace
extern "C" int emu_nordberg_init(const double* t, const double* pa, const double* epi, double* p, int* bad) {
    NordInitArgs a{t, pa, epi, p, bad};
    const size_t lds = (size_t)(((tff::POSE_LDS_DOUBLES + 1) & ~1) + tff::gh_lds_doubles(tff::NordbergModel::U, tff::NordbergModel::C, 0)) * sizeof(double);
    emu::launch(k_emu_nordberg_init, 1, 64, lds, a);
    return 0;
}

// The two evaluations of pinv(W) for a 4 x 4 weight block with one direction under the tolerance (pi_kernel.h): the Cholesky
// shortcut pinv_one_null_packed and the Jacobi eigen-decomposition it replaces.  W: 4 x 4 symmetric (row-major, already shifted by
// 1e-12 I as pi_block_W delivers it); outputs: packed lower triangles (10 doubles each), *ok = the shortcut's own verdict.
extern "C" int emu_pinv_one_null(const double* W16, double tolW, double* wp_shortcut, double* wp_jacobi, int* ok) {
    double W[4][4], V[4][4], Wp[10];
    for (int a = 0; a < 4; ++a) for (int c = 0; c < 4; ++c) W[a][c] = W16[4 * a + c];
    *ok = tff::pinv_one_null_packed<4>(W, tolW, Wp) ? 1 : 0;
    for (int e = 0; e < 10; ++e) wp_shortcut[e] = Wp[e];
    tff::jacobi_small<4, true>(W, V);
    double inv[4];
    for (int a = 0; a < 4; ++a) inv[a] = (W[a][a] > tolW) ? 1.0 / W[a][a] : 0.0;
    for (int a = 0; a < 4; ++a)
        for (int b = 0; b <= a; ++b) {
            double acc = 0.0;
            for (int k = 0; k < 4; ++k) acc += V[a][k] * inv[k] * V[b][k];
            wp_jacobi[a * (a + 1) / 2 + b] = acc;
        }
    return 0;
}
#endif  // TFF_EMU_LINEAR_TFT_ONLY
