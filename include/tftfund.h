/*
 * tftfund.h -- C ABI of the MI355X-native batched trifocal-tensor /
 * fundamental-matrix pose estimators (libtftfund.so, built by hipcc for gfx950).
 *
 * This is the drop-in boundary for the hot path of LauraFJulia/TFT_vs_Fund.
 * The reference has no FFI layer; its "operator API" is the MATLAB calling
 * convention of the method handles
 *     [R_t_2,R_t_3,Reconst,T,iter] = Method(Corresp,CalM)
 * (experiments.m:51-59,108; experiments_real.m:53-61,126; example.m:32-42).
 * Each tff_*_pose_batch entry point below replaces one such method for a batch
 * of B independent triplets; the MEX shim in matlab/ and the ctypes binding in
 * tft_vs_fund_amd/api.py bind exactly these symbols.
 *
 * Data layout (all IEEE double, MATLAB column-major, caller owns every buffer):
 *   corresp  B x (6 x N): triplet b, correspondence n = 6 contiguous doubles
 *            [x1 y1 x2 y2 x3 y3] at corresp[(b*N + n)*6]          (Corresp, 6xN)
 *   calm     27 doubles per triplet = 9x3 column-major [K1;K2;K3]; calm_stride
 *            is 27 (one per triplet) or 0 (one shared by the batch)   (CalM, 9x3)
 *   Rt2,Rt3  B x 12: 3x4 column-major [R|t], camera 1 = [I|0], |t2| = 1
 *   T        B x 27: T(j,k,i) at j + 3k + 9i, unit Frobenius norm, global sign free
 *   reconst  B x (3 x N) or NULL                                   (Reconst, 3xN)
 *   iter     B int32 or NULL  (0 for the linear methods, GH iterations otherwise)
 *   status   B int32 or NULL  (replaces MATLAB exceptions, see TFF_ST_*)
 *
 * Every function returns 0 on success or a negative code (-hipError_t for HIP
 * failures, TFF_E_* otherwise); tff_last_error() gives a thread-local message.
 * `_dev` variants take device pointers valid on the context's device and only
 * enqueue work on the context's stream (no synchronisation).  The context owns device
 * workspaces (status scratch when status == NULL, the records and spill slices of the
 * iterative methods) that grow on demand: a call with a larger B or N than any earlier
 * call of that method may hipMalloc / hipFree.  Inside a hipGraph capture use a `_dev`
 * entry point only after a warm-up call with the same method and B, N at least as large.
 * `_host` variants take host pointers and perform H2D, compute, D2H and a stream
 * synchronisation.
 * A context is bound to one device.  Its entry points are serialised by an internal lock
 * (the workspaces are shared state) and its work by its stream; when the stream is changed
 * (tff_ctx_set_stream) work on the new stream waits, on the device, for the work already
 * enqueued on the old one.  For concurrent streams or threads use one context each:
 * different contexts are independent.  No global state.
 */
#ifndef TFTFUND_H
#define TFTFUND_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct tff_ctx tff_ctx;

/* per-triplet status codes */
#define TFF_ST_OK 0
#define TFF_ST_TOO_FEW 1    /* N < 7 (TFT) or N < 8 (F): experiments.m:99, linearF.m:35-37 */
#define TFF_ST_NONFINITE 2  /* NaN/Inf in the result: Gauss_Helmert.m:53-55,63-65 */
#define TFF_ST_NO_POSE 3    /* no candidate with score >= 0: R_f unassigned in R_t_from_TFT.m:91-104 */
#define TFF_ST_NO_PARAM 5   /* PiColPoseEstimation.m:84-89: error('The minimal param could not be found') */
#define TFF_ST_RANK 4       /* Nordberg: P2(:,1:3) or P3(:,1:3) of rank < 2 (NordbergTFT...m:58,60 would fail: null() returns two columns) */

/* error codes (besides -hipError_t) */
#define TFF_E_INVALID (-10001)
#define TFF_E_NOMEM (-10002)

/* options for tff_ctx_set_option */
#define TFF_OPT_SOLVER 1    /* 0 (default): fast tiers (Gram matrix + Cholesky inverse iteration, certified sign-only cheirality votes) with
                             * the exact kernel (Householder QR of the explicit design matrix, one-sided Jacobi fall-backs: the accuracy
                             * of the reference's svd() calls) over the triplets they could not finish or certify; 1: exact kernel for all */
#define TFF_OPT_EXACT_BELOW 5 /* batches with N < value go to the exact kernel as a whole (default 12: minimal samples, where the two smallest
                             * singular values of the design matrix often nearly coincide); 0 = only the flagged triplets */
#define TFF_OPT_STAGE_LDS 2 /* -1 auto (default: staged in LDS while that costs no occupancy, N <= 200 for the TFT kernels, N <= 48 for LinearF),
                             * 0 re-read correspondences through L2, 1 stage them in LDS */

#define TFF_OPT_GH_EXACT 4  /* Gauss-Helmert methods: 1 = always form pinv(W) through per-block eigen-decompositions -- an A/B switch: it carries the
                             * 1e-6 .. 1e-4 noise of any fp64 pinv(W) (default 0: deflated block pseudo-inverse + factored strong direction, which
                             * reproduce a 50-digit evaluation of the reference's iteration to 1e-11 for Ressl, Nordberg and Pi) */
#define TFF_OPT_SPILL 6     /* per-correspondence state of the iterative methods: 0 (default) it leaves the LDS for the context's global slices whenever that lets
                             * more workgroups share a CU (measured faster, at the price of HBM traffic); 1 = only when the LDS cannot hold it (large N) */
#define TFF_OPT_KERNEL 3    /* Kernel variants of the iterative TFT methods: 0 automatic (default: a workgroup per triplet for the iteration -- two wavefronts
                             * for Ressl / Nordberg / Pi / PiCol, four for FaugPapa -- at every N since round 4);
                             * 1 the fused single-wavefront kernels (one wavefront per triplet from start to end);
                             * 2 workgroup kernels always (the same as 0 now) */
#define TFF_OPT_ROWS 7      /* LinearTFT / LinearF pose kernels, and the linear stage + pose tail of the iterative methods: 1 four triplets per wavefront,
                             * one per row of 16 lanes (csrc/tft_rows_kernel.h, f_rows_kernel.h, gh_rows_kernel.h, optimf_rows_kernel.h); 0 one triplet
                             * per wavefront (csrc/tft_kernel.h, f_kernel.h: the lowest latency for batches under ~1 000 triplets, ~16 us less per call); 2 (default)
                             * = 1 for every method and every batch size since the end of round 5.  (Before, the two linear methods went by batch size;
                             * the two routes agree to 1e-14 but not bit for bit, so a triplet's last bits depended on the batch it arrived in.)  With the
                             * default the same triplet gives the same bits -- and, for the iterative methods, the same `iter` -- in a batch of one, of 1 023
                             * or of a million, through the *_sampled_dev entry points in chunks of any size, and in any shard of a multi-GPU call.
                             * Whatever the route: a triplet with status != 0 has NaN in every output (T, R_t_2, R_t_3, Reconst) */
#define TFF_OPT_PRE 10      /* trifocal row kernels (TFF_OPT_ROWS route): where the three Normalize2Ddata calls and the 96 moment sums of linearTFT's system are
                             * computed.  0 (default) = inside the row kernels (two passes over the correspondences); 1 = in a kernel of their own, one triplet
                             * per wavefront, the correspondences read from HBM once and parked in LDS (csrc/tft_moments_kernel.h), the row kernels starting
                             * from its 112-double record; 2 = that kernel from N >= 48.  An A/B switch: measured slower than the fused passes on MI355X
                             * (profiles/r5_ab_pre.txt) -- the path is bound by fp64 issue, not by those passes' memory waits.  Results agree to rounding */
#define TFF_OPT_COUNT_ROWS 11 /* tff_inlier_count_batch_dev on many hypotheses of one scene: 1 (default) four hypotheses per wavefront, one per row of 16 lanes
                             * (the cameras composed once per row, 25 trips of 16 over a 400-correspondence scene); 0 one hypothesis per wavefront.  Identical counts */
#define TFF_OPT_DEBUG_FP_HANDOVER 8 /* test hook: 1 = FaugPapa's block kernel hands every third triplet back to the generic workgroup kernel, as it does when its
                             * pseudo-inverse reports a failure (exercises that production fall-back; results must not depend on it beyond the
                             * generic kernel's LAPACK-level noise) */
#define TFF_OPT_DEBUG_ADAPTIVE 9 /* profiling hook: 1 = the *_debug_dev entry points of the linear methods keep the production vote logic (main candidates only,
                             * scale sums during the votes) instead of evaluating all four cheirality scores for the debug record */
#define TFF_DEBUG_STRIDE 128 /* doubles per triplet written by the *_debug_dev entry points */

int tff_version(void);
const char* tff_last_error(void);

/* Context: device ordinal; owns a stream unless one is supplied. */
int tff_ctx_create(tff_ctx** out, int device);
void tff_ctx_destroy(tff_ctx* ctx);
int tff_ctx_set_stream(tff_ctx* ctx, void* hip_stream); /* borrow the caller's hipStream_t; NULL is the device's null stream */
int tff_ctx_use_own_stream(tff_ctx* ctx);               /* back to the context's own non-blocking stream */
void* tff_ctx_get_stream(tff_ctx* ctx);
int tff_ctx_set_option(tff_ctx* ctx, int option, long value);
int tff_ctx_synchronize(tff_ctx* ctx);

/* LinearTFTPoseEstimation (TFT_methods/LinearTFTPoseEstimation.m:44-62):
 * Normalize2Ddata x3 -> linearTFT -> transform_TFT -> R_t_from_TFT -> (Reconst). */
int tff_linear_tft_pose_batch_dev(tff_ctx* ctx, const double* corresp, const double* calm, int64_t calm_stride,
                                  int64_t B, int32_t N, double* Rt2, double* Rt3, double* T, double* reconst,
                                  int32_t* iter, int32_t* status);
int tff_linear_tft_pose_batch_host(tff_ctx* ctx, const double* corresp, const double* calm, int64_t calm_stride,
                                   int64_t B, int32_t N, double* Rt2, double* Rt3, double* T, double* reconst,
                                   int32_t* iter, int32_t* status);
/* same, additionally writing B x TFF_DEBUG_STRIDE intermediates (unconstrained tensor,
 * epipoles, constrained tensor, cheirality votes, t3 scale, solver iterations, normalisation) */
int tff_linear_tft_pose_batch_debug_dev(tff_ctx* ctx, const double* corresp, const double* calm, int64_t calm_stride,
                                        int64_t B, int32_t N, double* Rt2, double* Rt3, double* T, double* reconst,
                                        int32_t* iter, int32_t* status, double* dbg);

/* the same record for LinearFPoseEstimation (phase stamps at dbg[80 ..], iteration counts of the two view pairs at dbg[69], dbg[70]) */
int tff_linear_f_pose_batch_debug_dev(tff_ctx* ctx, const double* corresp, const double* calm, int64_t calm_stride,
                                      int64_t B, int32_t N, double* Rt2, double* Rt3, double* T, double* reconst,
                                      int32_t* iter, int32_t* status, double* dbg);

/* ResslTFTPoseEstimation (TFT_methods/ResslTFTPoseEstimation.m:47-177): linearTFT, Ressl's 20-parameter /
 * 2-constraint minimal parameterisation, Gauss-Helmert refinement (Optimization/Gauss_Helmert.m:38-83),
 * then transform_TFT -> R_t_from_TFT -> (Reconst).  iter = Gauss-Helmert iterations.  Any N (the per-correspondence state spills to a
 * global workspace when it exceeds the LDS). */
int tff_ressl_tft_pose_batch_dev(tff_ctx* ctx, const double* corresp, const double* calm, int64_t calm_stride,
                                 int64_t B, int32_t N, double* Rt2, double* Rt3, double* T, double* reconst,
                                 int32_t* iter, int32_t* status);
int tff_ressl_tft_pose_batch_host(tff_ctx* ctx, const double* corresp, const double* calm, int64_t calm_stride,
                                  int64_t B, int32_t N, double* Rt2, double* Rt3, double* T, double* reconst,
                                  int32_t* iter, int32_t* status);
int tff_ressl_tft_pose_batch_debug_dev(tff_ctx* ctx, const double* corresp, const double* calm, int64_t calm_stride,
                                       int64_t B, int32_t N, double* Rt2, double* Rt3, double* T, double* reconst,
                                       int32_t* iter, int32_t* status, double* dbg);

/* NordbergTFTPoseEstimation (TFT_methods/NordbergTFTPoseEstimation.m:47-222): three orthogonal matrices in
 * axis-angle form + a 10-entry sparse tensor (19 parameters, 1 constraint), Gauss-Helmert refinement.
 * The projective fix-up for rank-deficient P2/P3 (:56-62) is applied (rank by sigma_3 = |det| / (sigma_1 sigma_2)). */
int tff_nordberg_tft_pose_batch_dev(tff_ctx* ctx, const double* corresp, const double* calm, int64_t calm_stride,
                                    int64_t B, int32_t N, double* Rt2, double* Rt3, double* T, double* reconst,
                                    int32_t* iter, int32_t* status);
int tff_nordberg_tft_pose_batch_debug_dev(tff_ctx* ctx, const double* corresp, const double* calm, int64_t calm_stride,
                                          int64_t B, int32_t N, double* Rt2, double* Rt3, double* T, double* reconst,
                                          int32_t* iter, int32_t* status, double* dbg);
int tff_nordberg_tft_pose_batch_host(tff_ctx* ctx, const double* corresp, const double* calm, int64_t calm_stride,
                                     int64_t B, int32_t N, double* Rt2, double* Rt3, double* T, double* reconst,
                                     int32_t* iter, int32_t* status);

/* FaugPapaTFTPoseEstimation (TFT_methods/FaugPapaTFTPoseEstimation.m:48-159): all 27 tensor entries as
 * parameters, 12 algebraic constraints (3 determinants + 9 extended-rank), Gauss-Helmert refinement. */
int tff_faugpapa_tft_pose_batch_dev(tff_ctx* ctx, const double* corresp, const double* calm, int64_t calm_stride,
                                    int64_t B, int32_t N, double* Rt2, double* Rt3, double* T, double* reconst,
                                    int32_t* iter, int32_t* status);
int tff_faugpapa_tft_pose_batch_debug_dev(tff_ctx* ctx, const double* corresp, const double* calm, int64_t calm_stride,
                                          int64_t B, int32_t N, double* Rt2, double* Rt3, double* T, double* reconst,
                                          int32_t* iter, int32_t* status, double* dbg);
int tff_faugpapa_tft_pose_batch_host(tff_ctx* ctx, const double* corresp, const double* calm, int64_t calm_stride,
                                     int64_t B, int32_t N, double* Rt2, double* Rt3, double* T, double* reconst,
                                     int32_t* iter, int32_t* status);

/* PiPoseEstimation (TFT_methods/PiPoseEstimation.m:50-182): Ponce-Hebert Pi matrices from the linear solution
 * (27 parameters, 9 constraints), 3 epipolar equations + 1 trilinearity per correspondence, Gauss-Helmert.
 * PiColPoseEstimation (TFT_methods/PiColPoseEstimation.m:50-218): the variant for collinear camera centres
 * (11 constraints, 3 + 2 equations per correspondence); TFF_ST_NO_PARAM where the reference raises
 * 'The minimal param could not be found'. */
int tff_pi_pose_batch_dev(tff_ctx* ctx, const double* corresp, const double* calm, int64_t calm_stride,
                          int64_t B, int32_t N, double* Rt2, double* Rt3, double* T, double* reconst,
                          int32_t* iter, int32_t* status);
int tff_pi_pose_batch_host(tff_ctx* ctx, const double* corresp, const double* calm, int64_t calm_stride,
                           int64_t B, int32_t N, double* Rt2, double* Rt3, double* T, double* reconst,
                           int32_t* iter, int32_t* status);
int tff_picol_pose_batch_dev(tff_ctx* ctx, const double* corresp, const double* calm, int64_t calm_stride,
                             int64_t B, int32_t N, double* Rt2, double* Rt3, double* T, double* reconst,
                             int32_t* iter, int32_t* status);
int tff_picol_pose_batch_host(tff_ctx* ctx, const double* corresp, const double* calm, int64_t calm_stride,
                              int64_t B, int32_t N, double* Rt2, double* Rt3, double* T, double* reconst,
                              int32_t* iter, int32_t* status);

/* Pi / PiCol (collinear != 0) with the start of the Gauss-Helmert iteration exposed: init_p (B x 27, the vector `pi`
 * of PiPoseEstimation.m:86 / PiColPoseEstimation.m:113) and init_x (B x 6N, `x_est`).  The Pi matrices depend on sign
 * and basis choices the reference leaves to svd (null(P), null(M.')); PiCol's result depends on them
 * (PiColPoseEstimation.m:93-94 is not covariant), so parity for it is checked from this common start. */
int tff_pi_pose_batch_debug_dev(tff_ctx* ctx, int32_t collinear, const double* corresp, const double* calm,
                                int64_t calm_stride, int64_t B, int32_t N, double* Rt2, double* Rt3, double* T,
                                double* reconst, int32_t* iter, int32_t* status, double* init_p, double* init_x);

/* OptimFPoseEstimation (F_methods/OptimFPoseEstimation.m:44-73): two fundamental matrices, each refined by
 * optimF (F_methods/optimF.m:34-109: linearF start, 9 parameters, constraints det F = 0 and |F| = 1, one
 * epipolar equation per correspondence, Gauss-Helmert) -> recover_R_t x2 -> t3 scale -> (Reconst) ->
 * T = TFT_from_P.  iter = it1 + it2.  Needs N >= 8. */
int tff_optim_f_pose_batch_dev(tff_ctx* ctx, const double* corresp, const double* calm, int64_t calm_stride,
                               int64_t B, int32_t N, double* Rt2, double* Rt3, double* T, double* reconst,
                               int32_t* iter, int32_t* status);
int tff_optim_f_pose_batch_host(tff_ctx* ctx, const double* corresp, const double* calm, int64_t calm_stride,
                                int64_t B, int32_t N, double* Rt2, double* Rt3, double* T, double* reconst,
                                int32_t* iter, int32_t* status);

/* LinearFPoseEstimation (F_methods/LinearFPoseEstimation.m:42-109): Normalize2Ddata x3 ->
 * linearF x2 (F_methods/linearF.m:32-62) -> recover_R_t x2 -> t3 scale -> (Reconst) ->
 * T = TFT_from_P (TFT_methods/TFT_from_P.m:25-33).  Needs N >= 8 (status TFF_ST_TOO_FEW otherwise). */
int tff_linear_f_pose_batch_dev(tff_ctx* ctx, const double* corresp, const double* calm, int64_t calm_stride,
                                int64_t B, int32_t N, double* Rt2, double* Rt3, double* T, double* reconst,
                                int32_t* iter, int32_t* status);
int tff_linear_f_pose_batch_host(tff_ctx* ctx, const double* corresp, const double* calm, int64_t calm_stride,
                                 int64_t B, int32_t N, double* Rt2, double* Rt3, double* T, double* reconst,
                                 int32_t* iter, int32_t* status);

/* ---- building blocks (device pointers; all arrays MATLAB column-major) ------------------------- */

/* triangulation3D (auxiliar_functions/triangulation3D.m:32-64): M = 2 or 3 cameras (3x4 each, cam_stride 12*M per
 * item or 0 = shared), pts B x (2M x N); X B x (4 x N) unit-norm homogeneous points, sign free, not dehomogenised. */
int tff_triangulate_batch_dev(tff_ctx* ctx, const double* cams, int64_t cam_stride, const double* pts, int64_t B,
                              int32_t M, int32_t N, double* X);

/* ReprError (auxiliar_functions/ReprError.m:39-65) for three views: RMS over the 3N reprojected points; pts3d
 * (B x 3 x N) or NULL to triangulate first (ReprError.m:43-44).  corresp_stride 6*N or 0 (one shared scene). */
int tff_repr_error_batch_dev(tff_ctx* ctx, const double* cams, int64_t cam_stride, const double* corresp,
                             int64_t corresp_stride, const double* pts3d, int64_t B, int32_t N, double* err);

/* Inlier count of pose hypotheses against ONE shared scene (6 x Ns): cameras K1[I|0], K2 Rt2[b], K3 Rt3[b];
 * a correspondence is an inlier when all six reprojection residuals after triangulation are <= threshold in
 * absolute value (experiments_real.m:94-98).  counts B int32; err (B, RMS) optional. */
int tff_inlier_count_batch_dev(tff_ctx* ctx, const double* scene, int32_t Ns, const double* calm, const double* Rt2,
                               const double* Rt3, int64_t B, double threshold, int32_t* counts, double* err);

/* transform_TFT (TFT_methods/transform_TFT.m:32-49), inverse = 0 or 1; M1..M3 3x3, m_stride 9 or 0 (shared). */
int tff_transform_tft_batch_dev(tff_ctx* ctx, const double* T, const double* M1, const double* M2, const double* M3,
                                int64_t m_stride, int64_t B, int32_t inverse, double* Tout);

/* R_t_from_TFT (TFT_methods/R_t_from_TFT.m:40-106): pixel-coordinate tensor + CalM + Corresp -> poses. */
int tff_rt_from_tft_batch_dev(tff_ctx* ctx, const double* T, const double* calm, int64_t calm_stride,
                              const double* corresp, int64_t B, int32_t N, double* Rt2, double* Rt3, int32_t* status);

/* linearTFT (TFT_methods/linearTFT.m:33-91) on points used as given (rows x1;y1;x2;y2;x3;y3 of corresp):
 * T (27, unit norm), P2, P3 (3x4 each, P1 = [I|0]; both NULL to skip). */
int tff_linear_tft_batch_dev(tff_ctx* ctx, const double* corresp, int64_t B, int32_t N, double* T, double* P2,
                             double* P3, int32_t* status);

/* linearF (F_methods/linearF.m:32-62; refine = 0) or optimF (F_methods/optimF.m:34-78; refine = 1) for the view
 * pairs (1,2) and (1,3) of each item: F21, F31 (B x 9, 3x3 column-major, x2' F21 x1 = 0).  linearF normalises its
 * inputs itself; iter (optimF: it1 + it2) may be NULL.  Needs N >= 8. */
int tff_linear_f_batch_dev(tff_ctx* ctx, const double* corresp, int64_t B, int32_t N, int32_t refine, double* F21,
                           double* F31, int32_t* iter, int32_t* status);

/* BundleAdjustment (Optimization/BundleAdjustment.m:49-216) for three views: refines the poses (Rt2_in, Rt3_in: B x 12,
 * camera 1 = [I|0]) and the space points (reconst0: B x 3N, or NULL to triangulate them first, :59-77) by Levenberg-Marquardt on
 * the reprojection residual in per-view normalised coordinates.  Outputs: poses with |t2| = 1, points (NULL to skip),
 * successful LM iterations, repr_err = norm of the final residual vector (normalised units, as BundleAdjustment.m:105).
 * MATLAB's lsqnonlin is closed source: the loop follows its documented LM defaults (see csrc/ba_kernel.h); results are
 * comparable at the converged optimum.  All correspondences must be visible in all three views. */
int tff_bundle_adjust_batch_dev(tff_ctx* ctx, const double* calm, int64_t calm_stride, const double* Rt2_in,
                                const double* Rt3_in, const double* corresp, int64_t B, int32_t N, const double* reconst0,
                                double* Rt2, double* Rt3, double* reconst, int32_t* iter, double* repr_err, int32_t* status);
int tff_bundle_adjust_batch_host(tff_ctx* ctx, const double* calm, int64_t calm_stride, const double* Rt2_in,
                                 const double* Rt3_in, const double* corresp, int64_t B, int32_t N, const double* reconst0,
                                 double* Rt2, double* Rt3, double* reconst, int32_t* iter, double* repr_err, int32_t* status);

/* BundleAdjustment (Optimization/BundleAdjustment.m:49-216) as the reference writes it, for M = 2 .. 6 views, in MATLAB's own
 * array layouts so that a gateway passes its arguments through: calm = CalM (3M x 3, column-major; calm_stride 0 = shared, 9M =
 * one per item), Rt_in = R_t_0 (3M x 4 column-major per item, camera 1 included and NOT required to be [I|0]: the change of
 * coordinates of :80-86 is done on the device, after the optional initial triangulation of :59-77 in the given frame),
 * corresp = Corresp (2M x N column-major per item), reconst0 = Reconst0 (3 x N per item) or NULL.  Outputs: Rt = R_t (3M x 4
 * column-major per item, R_t(1:3,:) = eye(3,4), |t2| = 1), reconst (3 x N per item, or NULL), iter, repr_err, status.
 * Missing observations (:28-29, :165): a NaN entry is handled as the reference's code handles it -- Normalize2Ddata.m:34-37 turns
 * every point of that VIEW into NaN, :165 then skips the whole view: its camera keeps the initial angles and translation, the
 * other views are adjusted.  Without reconst0, fewer than two complete views: status TFF_ST_TOO_FEW and NaN outputs (the
 * reference stops with an error at :73-74).  Same Levenberg-Marquardt loop as tff_bundle_adjust_batch_dev. */
int tff_bundle_adjust_views_batch_dev(tff_ctx* ctx, int32_t M, const double* calm, int64_t calm_stride, const double* Rt_in,
                                      const double* corresp, int64_t B, int32_t N, const double* reconst0, double* Rt,
                                      double* reconst, int32_t* iter, double* repr_err, int32_t* status);
int tff_bundle_adjust_views_batch_host(tff_ctx* ctx, int32_t M, const double* calm, int64_t calm_stride, const double* Rt_in,
                                       const double* corresp, int64_t B, int32_t N, const double* reconst0, double* Rt,
                                       double* reconst, int32_t* iter, double* repr_err, int32_t* status);

/* Minimal-sample hypotheses (BASELINE.json config 4): hypothesis b = the n correspondences
 * sample_idx[b*n .. b*n+n) of one shared scene (6 x Ns); n >= 7 (TFT) / 8 (F); shared CalM (27). */
int tff_linear_tft_pose_sampled_dev(tff_ctx* ctx, const double* scene, int32_t Ns, const double* calm,
                                    const int32_t* sample_idx, int64_t B, int32_t n, double* Rt2, double* Rt3,
                                    double* T, int32_t* status);
int tff_linear_f_pose_sampled_dev(tff_ctx* ctx, const double* scene, int32_t Ns, const double* calm,
                                  const int32_t* sample_idx, int64_t B, int32_t n, double* Rt2, double* Rt3,
                                  double* T, int32_t* status);

/* ---- multi-GPU (one process, one host thread + stream per device) ----------------------------------
 * The reference runs its triplets one after the other in one MATLAB thread (experiments.m:91-108); they are independent,
 * so a batch is cut into contiguous shards of ceil(B / G) triplets, shard g on device g, with no collective on the data
 * path.  tff_pose_batch_host_multi lands every shard directly in the caller's host arrays (same argument meaning as the
 * single-device `_host` entry points).  tff_pose_batch_dev_multi works on device-resident shards and gathers the
 * fixed-size result records (51 doubles per triplet: Rt2 | Rt3 | T) of all shards onto every device with one
 * ncclAllGather (RCCL over xGMI; librccl.so is opened on first use). */
typedef struct tff_multi tff_multi;
#define TFF_METHOD_LINEAR_TFT 0   /* method ids: the order of experiments.m:51-59 */
#define TFF_METHOD_RESSL_TFT 1
#define TFF_METHOD_NORDBERG_TFT 2
#define TFF_METHOD_FAUGPAPA_TFT 3
#define TFF_METHOD_PI 4
#define TFF_METHOD_PICOL 5
#define TFF_METHOD_LINEAR_F 6
#define TFF_METHOD_OPTIM_F 7
int tff_multi_create(tff_multi** out, const int32_t* devices, int32_t n_devices);   /* devices NULL: 0..n-1; n_devices <= 0: all visible */
void tff_multi_destroy(tff_multi* m);
int32_t tff_multi_size(const tff_multi* m);
tff_ctx* tff_multi_ctx(tff_multi* m, int32_t rank);                                  /* per-device context (options, synchronisation) */
void tff_multi_shard(const tff_multi* m, int64_t B, int32_t rank, int64_t* begin, int64_t* end);   /* [rank*chunk, min(B,(rank+1)*chunk)), chunk = ceil(B/G) */
int tff_pose_batch_host_multi(tff_multi* m, int32_t method, const double* corresp, const double* calm, int64_t calm_stride,
                              int64_t B, int32_t N, double* Rt2, double* Rt3, double* T, double* reconst, int32_t* iter,
                              int32_t* status);
/* corresp[g], calm[g]: shard g on device g.  records[g]: G * chunk * 51 doubles on device g; afterwards block r of EVERY
 * device = [Rt2 (chunk x 12) | Rt3 (chunk x 12) | T (chunk x 27)] of shard r.  status[g] (or status == NULL): G * chunk int32. */
int tff_pose_batch_dev_multi(tff_multi* m, int32_t method, const double* const* corresp, const double* const* calm,
                             int64_t calm_stride, int64_t B, int32_t N, double* const* records, int32_t* const* status);

#ifdef __cplusplus
}
#endif
#endif
