// TEST INFRASTRUCTURE ONLY: runs the HIP kernels of tft_vs_fund_amd/csrc on the
// CPU through tests/emu/hip_emu.h (one std::thread per lane) so that their
// logic is covered by `pytest -m "not gpu"` and can be run under sanitizers.
// Never linked into libtftfund.so.
#define TFF_CPU_EMU 1
#include "../../tft_vs_fund_amd/csrc/launch.h"

extern "C" int emu_linear_tft_pose(const double* corresp, const double* calm, long calm_stride, long B, int N,
                                   int flags, double* Rt2, double* Rt3, double* T, double* reconst, int* iter,
                                   int* status, double* dbg) {
    tff::LinearTftArgs a{corresp, calm, calm_stride, B, N, flags, Rt2, Rt3, T, reconst, iter, status, dbg};
    if (reconst) a.flags |= tff::FLAG_RECONST;
    a.flags = tff::pose_auto_flags(N, a.flags);
    emu::launch(tff::k_linear_tft_pose, tff::pose_grid(B), 64, tff::pose_lds_bytes(N, a.flags), a);
    return 0;
}
