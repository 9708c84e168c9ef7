"""
Host-side mirror of the reference's operator API on top of libtftfund.so.

The reference's drop-in surface is the MATLAB calling convention
    [R_t_2, R_t_3, Reconst, T, iter] = Method(Corresp, CalM)
(experiments.m:51-59,108).  This module exposes

  * the same names with the same argument meaning for ONE triplet
    (`LinearTFTPoseEstimation(Corresp, CalM)` with Corresp 6xN, CalM 9x3), and
  * `*_batch` variants for B triplets -- the form the GPU is built for --
    on numpy arrays (host path: the library does H2D/D2H) or on torch CUDA
    tensors (device path: only pointers and the current stream are passed; torch
    is plumbing for device memory and streams, nothing else).

There is no CPU fallback: if libtftfund.so is missing or no HIP device is
present, every call raises.
"""
import ctypes
import os
import threading

import numpy as np
# torch is plumbing here (device memory, streams, torch.distributed) -- and it must be
# imported BEFORE libtftfund.so is dlopen'ed: the torch wheel bundles its own
# libamdhip64.so (soname libamdhip64.so.7); loading it first makes the dynamic loader
# bind libtftfund.so to that same HIP runtime instead of starting a second one.
import torch  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libtftfund.so")

TFF_OPT_SOLVER = 1
TFF_OPT_STAGE_LDS = 2
TFF_OPT_KERNEL = 3
TFF_OPT_GH_EXACT = 4
TFF_OPT_EXACT_BELOW = 5
TFF_OPT_SPILL = 6
TFF_OPT_ROWS = 7
TFF_OPT_DEBUG_FP_HANDOVER = 8
TFF_OPT_DEBUG_ADAPTIVE = 9
TFF_OPT_PRE = 10
DEBUG_STRIDE = 128

ST_OK, ST_TOO_FEW, ST_NONFINITE, ST_NO_POSE, ST_RANK, ST_NO_PARAM = 0, 1, 2, 3, 4, 5

_c_dp = ctypes.c_void_p
_POSE_SIG = [ctypes.c_void_p, _c_dp, _c_dp, ctypes.c_int64, ctypes.c_int64, ctypes.c_int32,
             _c_dp, _c_dp, _c_dp, _c_dp, _c_dp, _c_dp]

_lib = None
_lib_lock = threading.Lock()


class TffError(RuntimeError):
    pass


def load_library(path=None):
    """dlopen libtftfund.so and declare its prototypes.  Raises if it is missing:
    the product has no other compute path."""
    global _lib
    with _lib_lock:
        if _lib is not None and path is None:
            return _lib
        p = path or _LIB_PATH
        if not os.path.exists(p):
            raise TffError("libtftfund.so not found at %s -- build it with `python -m tft_vs_fund_amd.build` "
                           "(hipcc --offload-arch=gfx950); there is no CPU fallback" % p)
        lib = ctypes.CDLL(p)
        lib.tff_version.restype = ctypes.c_int
        lib.tff_last_error.restype = ctypes.c_char_p
        lib.tff_ctx_create.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int]
        lib.tff_ctx_destroy.argtypes = [ctypes.c_void_p]
        lib.tff_ctx_destroy.restype = None
        lib.tff_ctx_set_stream.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        lib.tff_ctx_use_own_stream.argtypes = [ctypes.c_void_p]
        lib.tff_ctx_get_stream.argtypes = [ctypes.c_void_p]
        lib.tff_ctx_get_stream.restype = ctypes.c_void_p
        lib.tff_ctx_set_option.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_long]
        lib.tff_ctx_synchronize.argtypes = [ctypes.c_void_p]
        for name in POSE_METHODS.values():
            for suffix in ("_dev", "_host"):
                fn = getattr(lib, name + suffix, None)
                if fn is not None:
                    fn.argtypes = _POSE_SIG
                    fn.restype = ctypes.c_int
        for name in POSE_METHODS.values():
            fn = getattr(lib, name + "_debug_dev", None)
            if fn is not None:
                fn.argtypes = _POSE_SIG + [_c_dp]
                fn.restype = ctypes.c_int
        V, I64, I32, F64 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32, ctypes.c_double
        protos = {
            "tff_triangulate_batch_dev": [V, V, I64, V, I64, I32, I32, V],
            "tff_repr_error_batch_dev": [V, V, I64, V, I64, V, I64, I32, V],
            "tff_inlier_count_batch_dev": [V, V, I32, V, V, V, I64, F64, V, V],
            "tff_transform_tft_batch_dev": [V, V, V, V, V, I64, I64, I32, V],
            "tff_rt_from_tft_batch_dev": [V, V, V, I64, V, I64, I32, V, V, V],
            "tff_linear_tft_batch_dev": [V, V, I64, I32, V, V, V, V],
            "tff_linear_f_batch_dev": [V, V, I64, I32, I32, V, V, V, V],
            "tff_bundle_adjust_batch_dev": [V, V, I64, V, V, V, I64, I32, V, V, V, V, V, V, V],
            "tff_bundle_adjust_batch_host": [V, V, I64, V, V, V, I64, I32, V, V, V, V, V, V, V],
            "tff_bundle_adjust_views_batch_dev": [V, I32, V, I64, V, V, I64, I32, V, V, V, V, V, V],
            "tff_bundle_adjust_views_batch_host": [V, I32, V, I64, V, V, I64, I32, V, V, V, V, V, V],
            "tff_pi_pose_batch_debug_dev": [V, I32, V, V, I64, I64, I32, V, V, V, V, V, V, V, V],
            "tff_linear_tft_pose_sampled_dev": [V, V, I32, V, V, I64, I32, V, V, V, V],
            "tff_linear_f_pose_sampled_dev": [V, V, I32, V, V, I64, I32, V, V, V, V],
            "tff_multi_create": [ctypes.POINTER(ctypes.c_void_p), V, I32],
            "tff_pose_batch_host_multi": [V, I32, V, V, I64, I64, I32, V, V, V, V, V, V],
            "tff_pose_batch_dev_multi": [V, I32, V, V, I64, I64, I32, V, V],
        }
        for name, sig in protos.items():
            fn = getattr(lib, name)
            fn.argtypes = sig
            fn.restype = ctypes.c_int
        lib.tff_multi_destroy.argtypes = [V]; lib.tff_multi_destroy.restype = None
        lib.tff_multi_size.argtypes = [V]; lib.tff_multi_size.restype = I32
        lib.tff_multi_ctx.argtypes = [V, I32]; lib.tff_multi_ctx.restype = V
        lib.tff_multi_shard.argtypes = [V, I64, I32, ctypes.POINTER(I64), ctypes.POINTER(I64)]; lib.tff_multi_shard.restype = None
        if path is None:
            _lib = lib
        return lib


# reference method name -> C entry point stem
POSE_METHODS = {
    "LinearTFTPoseEstimation": "tff_linear_tft_pose_batch",
    "LinearFPoseEstimation": "tff_linear_f_pose_batch",
    "ResslTFTPoseEstimation": "tff_ressl_tft_pose_batch",
    "FaugPapaTFTPoseEstimation": "tff_faugpapa_tft_pose_batch",
    "NordbergTFTPoseEstimation": "tff_nordberg_tft_pose_batch",
    "OptimFPoseEstimation": "tff_optim_f_pose_batch",
    "PiPoseEstimation": "tff_pi_pose_batch",
    "PiColPoseEstimation": "tff_picol_pose_batch",
}

# every symbol include/tftfund.h declares (checked by the CPU test-suite)
EXPORTED_SYMBOLS = [
    "tff_version", "tff_last_error", "tff_ctx_create", "tff_ctx_destroy", "tff_ctx_set_stream",
    "tff_ctx_use_own_stream", "tff_ctx_get_stream", "tff_ctx_set_option", "tff_ctx_synchronize",
    "tff_linear_tft_pose_batch_dev", "tff_linear_tft_pose_batch_host", "tff_linear_tft_pose_batch_debug_dev", "tff_linear_f_pose_batch_debug_dev",
    "tff_linear_f_pose_batch_dev", "tff_linear_f_pose_batch_host",
    "tff_ressl_tft_pose_batch_dev", "tff_ressl_tft_pose_batch_host", "tff_ressl_tft_pose_batch_debug_dev",
    "tff_faugpapa_tft_pose_batch_dev", "tff_faugpapa_tft_pose_batch_host", "tff_faugpapa_tft_pose_batch_debug_dev",
    "tff_nordberg_tft_pose_batch_dev", "tff_nordberg_tft_pose_batch_host", "tff_nordberg_tft_pose_batch_debug_dev",
    "tff_optim_f_pose_batch_dev", "tff_optim_f_pose_batch_host",
    "tff_pi_pose_batch_dev", "tff_pi_pose_batch_host", "tff_picol_pose_batch_dev", "tff_picol_pose_batch_host",
    "tff_pi_pose_batch_debug_dev",
    "tff_triangulate_batch_dev", "tff_repr_error_batch_dev", "tff_inlier_count_batch_dev", "tff_transform_tft_batch_dev",
    "tff_rt_from_tft_batch_dev", "tff_linear_tft_batch_dev", "tff_linear_f_batch_dev", "tff_bundle_adjust_batch_dev", "tff_bundle_adjust_batch_host", "tff_bundle_adjust_views_batch_dev", "tff_bundle_adjust_views_batch_host", "tff_linear_tft_pose_sampled_dev", "tff_linear_f_pose_sampled_dev",
    "tff_multi_create", "tff_multi_destroy", "tff_multi_size", "tff_multi_ctx", "tff_multi_shard", "tff_pose_batch_host_multi", "tff_pose_batch_dev_multi",
]

# method ids of the multi-GPU entry points (include/tftfund.h TFF_METHOD_*: the order of experiments.m:51-59)
METHOD_IDS = {"LinearTFTPoseEstimation": 0, "ResslTFTPoseEstimation": 1, "NordbergTFTPoseEstimation": 2, "FaugPapaTFTPoseEstimation": 3,
              "PiPoseEstimation": 4, "PiColPoseEstimation": 5, "LinearFPoseEstimation": 6, "OptimFPoseEstimation": 7}


def _check(lib, rc, what):
    if rc != 0:
        msg = lib.tff_last_error()
        raise TffError("%s failed (%d): %s" % (what, rc, msg.decode() if msg else "?"))


class Context:
    """A tff_ctx: one device, one stream.  `solver`: 'invit' (fast tiers -- Gram matrix + Cholesky inverse iteration,
    certified sign-only votes -- with the exact kernel over what they cannot finish) or 'exact' (exact kernel for every
    triplet: Householder QR of the explicit design matrix, one-sided Jacobi fall-backs; 'jacobi' is the old name)."""

    def __init__(self, device=0, solver="invit", stage_lds=-1, lib_path=None):
        self.lib = load_library(lib_path)
        h = ctypes.c_void_p()
        _check(self.lib, self.lib.tff_ctx_create(ctypes.byref(h), int(device)), "tff_ctx_create")
        self.handle = h
        self.device = int(device)
        self.set_solver(solver)
        _check(self.lib, self.lib.tff_ctx_set_option(self.handle, TFF_OPT_STAGE_LDS, int(stage_lds)), "set_option")

    def set_solver(self, solver):
        v = {"invit": 0, "jacobi": 1, "exact": 1}[solver]
        _check(self.lib, self.lib.tff_ctx_set_option(self.handle, TFF_OPT_SOLVER, v), "set_option")

    def set_exact_below(self, n):
        """TFF_OPT_EXACT_BELOW: batches with fewer than n correspondences per triplet go to the exact kernel as a whole
        (default 12); 0 = only the triplets the fast tiers flag."""
        _check(self.lib, self.lib.tff_ctx_set_option(self.handle, TFF_OPT_EXACT_BELOW, int(n)), "set_option")

    def set_spill_only_if_needed(self, on):
        """TFF_OPT_SPILL: True = the per-correspondence state of the iterative methods stays in LDS whenever it fits (fewer workgroups per CU,
        HBM traffic near the algorithmic bytes); False (default) = it goes to global slices when that raises the occupancy."""
        _check(self.lib, self.lib.tff_ctx_set_option(self.handle, TFF_OPT_SPILL, int(bool(on))), "set_option")

    def set_count_rows(self, on):
        """TFF_OPT_COUNT_ROWS: inlier counts with four hypotheses per wavefront (default) or one."""
        _check(self.lib, self.lib.tff_ctx_set_option(self.handle, 11, int(bool(on))), "set_option")

    def set_rows(self, on):
        """TFF_OPT_ROWS: "auto" / 2 (default) and True / 1 = the row kernels (four triplets per wavefront, one per row of 16 lanes) at any batch size:
        a triplet's bits do not depend on the batch it arrives in; False / 0 = one triplet per wavefront (lowest latency for small batches)."""
        v = 2 if on == "auto" else (int(on) if isinstance(on, int) and not isinstance(on, bool) else int(bool(on)))
        _check(self.lib, self.lib.tff_ctx_set_option(self.handle, TFF_OPT_ROWS, v), "set_option")

    def set_pre(self, on):
        """TFF_OPT_PRE (A/B switch): False / 0 (default) = normalisations + moment sums inside the trifocal row kernels; True / 1 = in a kernel of
        their own (one triplet per wavefront, correspondences read once; measured slower: profiles/r5_ab_pre.txt); "auto" / 2 = that kernel from N >= 48."""
        v = 2 if on == "auto" else (int(on) if isinstance(on, int) and not isinstance(on, bool) else int(bool(on)))
        _check(self.lib, self.lib.tff_ctx_set_option(self.handle, TFF_OPT_PRE, v), "set_option")

    def set_debug_adaptive(self, on):
        """TFF_OPT_DEBUG_ADAPTIVE (profiling hook): debug entry points keep the production cheirality-vote logic."""
        _check(self.lib, self.lib.tff_ctx_set_option(self.handle, TFF_OPT_DEBUG_ADAPTIVE, int(bool(on))), "set_option")

    def set_debug_fp_handover(self, on):
        """TFF_OPT_DEBUG_FP_HANDOVER (test hook): FaugPapa's block kernel hands every third triplet back to the generic workgroup kernel."""
        _check(self.lib, self.lib.tff_ctx_set_option(self.handle, TFF_OPT_DEBUG_FP_HANDOVER, int(bool(on))), "set_option")

    def set_gh_exact(self, on):
        """Gauss-Helmert methods: True = pinv(W) always through per-block eigen-decompositions (A/B; slower)."""
        _check(self.lib, self.lib.tff_ctx_set_option(self.handle, TFF_OPT_GH_EXACT, int(bool(on))), "set_option")

    def set_kernel_variant(self, v):
        """TFF_OPT_KERNEL.  Iterative TFT methods:
        0 automatic (workgroup per triplet, fused single-wavefront kernel at small N), 1 fused always, 2 workgroup always."""
        _check(self.lib, self.lib.tff_ctx_set_option(self.handle, TFF_OPT_KERNEL, int(v)), "set_option")

    def set_stream(self, stream_ptr):
        """Enqueue on the caller's hipStream_t (0 / None = the device's null stream, torch's default)."""
        _check(self.lib, self.lib.tff_ctx_set_stream(self.handle, ctypes.c_void_p(stream_ptr or 0)), "set_stream")

    def use_own_stream(self):
        _check(self.lib, self.lib.tff_ctx_use_own_stream(self.handle), "use_own_stream")

    def stream_ptr(self):
        """The hipStream_t the context launches on (its own stream unless set_stream was called), as an integer."""
        return int(self.lib.tff_ctx_get_stream(self.handle) or 0)

    def synchronize(self):
        _check(self.lib, self.lib.tff_ctx_synchronize(self.handle), "synchronize")

    def close(self):
        if getattr(self, "handle", None):
            self.lib.tff_ctx_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- batched pose estimation ------------------------------------------
    def pose_batch(self, method, corresp, calm, reconst=True, debug=False):
        """corresp: (B, N, 6) float64 (== MATLAB 6 x N x B); calm: (9,3) shared or
        (B, 9, 3).  numpy in -> numpy out (host path); torch CUDA tensors in ->
        torch tensors out (device path, asynchronous on the current stream).
        Returns dict(R_t_2 (B,3,4), R_t_3 (B,3,4), T (B,3,3,3) indexed [b,j,k,i],
        Reconst (B,3,N) or None, iter (B,), status (B,))."""
        stem = POSE_METHODS[method]
        if isinstance(corresp, np.ndarray):
            return self._pose_batch_host(stem, corresp, calm, reconst)
        return self._pose_batch_dev(stem, corresp, calm, reconst, debug)

    @staticmethod
    def _calm_cm_np(calm, B):
        calm = np.asarray(calm, dtype=np.float64)
        if calm.shape == (9, 3):
            return np.ascontiguousarray(calm.T).reshape(27), 0
        if calm.shape == (B, 9, 3):
            return np.ascontiguousarray(calm.transpose(0, 2, 1)).reshape(B * 27), 27
        raise ValueError("CalM must be 9x3 or Bx9x3")

    def _pose_batch_host(self, stem, corresp, calm, reconst):
        corresp = np.ascontiguousarray(corresp, dtype=np.float64)
        if corresp.ndim != 3 or corresp.shape[2] != 6:
            raise ValueError("corresp must be (B, N, 6)")
        B, N, _ = corresp.shape
        calm_cm, stride = self._calm_cm_np(calm, B)
        Rt2 = np.empty((B, 12)); Rt3 = np.empty((B, 12)); T = np.empty((B, 27))
        rec = np.empty((B, N, 3)) if reconst else None
        it = np.zeros(B, dtype=np.int32); st = np.zeros(B, dtype=np.int32)
        fn = getattr(self.lib, stem + "_host")
        ptr = lambda a: ctypes.c_void_p(a.ctypes.data) if a is not None else None
        _check(self.lib, fn(self.handle, ptr(corresp), ptr(calm_cm), stride, B, N, ptr(Rt2), ptr(Rt3), ptr(T), ptr(rec),
                            ptr(it), ptr(st)), stem + "_host")
        return dict(R_t_2=Rt2.reshape(B, 4, 3).transpose(0, 2, 1), R_t_3=Rt3.reshape(B, 4, 3).transpose(0, 2, 1),
                    T=T.reshape(B, 3, 3, 3).transpose(0, 3, 2, 1),
                    Reconst=rec.transpose(0, 2, 1) if reconst else None, iter=it, status=st)

    def _pose_batch_dev(self, stem, corresp, calm, reconst, debug):
        if not (corresp.is_cuda and corresp.dtype == torch.float64 and corresp.is_contiguous()):
            raise ValueError("corresp must be a contiguous float64 CUDA tensor of shape (B, N, 6)")
        B, N, _ = corresp.shape
        dev = corresp.device
        if tuple(calm.shape) == (9, 3):
            calm_cm, stride = calm.t().contiguous().reshape(27), 0
        elif tuple(calm.shape) == (27,):
            calm_cm, stride = calm.contiguous(), 0          # already column-major
        else:
            calm_cm, stride = calm.transpose(1, 2).contiguous().reshape(B * 27), 27
        calm_cm = calm_cm.to(device=dev, dtype=torch.float64)
        Rt2 = torch.empty((B, 12), dtype=torch.float64, device=dev)
        Rt3 = torch.empty((B, 12), dtype=torch.float64, device=dev)
        T = torch.empty((B, 27), dtype=torch.float64, device=dev)
        rec = torch.empty((B, N, 3), dtype=torch.float64, device=dev) if reconst else None
        it = torch.zeros(B, dtype=torch.int32, device=dev)
        st = torch.zeros(B, dtype=torch.int32, device=dev)
        self.set_stream(torch.cuda.current_stream(dev).cuda_stream)
        p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
        out = dict()
        if debug and stem in ("tff_pi_pose_batch", "tff_picol_pose_batch"):
            # the start of the Gauss-Helmert iteration: `pi` (27) and `x_est` (6N) per triplet
            ip = torch.zeros((B, 27), dtype=torch.float64, device=dev)
            ix = torch.zeros((B, 6 * N), dtype=torch.float64, device=dev)
            _check(self.lib, self.lib.tff_pi_pose_batch_debug_dev(self.handle, int(stem == "tff_picol_pose_batch"), p(corresp), p(calm_cm), stride,
                                                                  B, N, p(Rt2), p(Rt3), p(T), p(rec), p(it), p(st), p(ip), p(ix)),
                   "tff_pi_pose_batch_debug_dev")
            out["init_p"] = ip
            out["init_x"] = ix
        elif debug:
            dbg = torch.zeros((B, DEBUG_STRIDE), dtype=torch.float64, device=dev)
            fn = getattr(self.lib, stem + "_debug_dev")
            _check(self.lib, fn(self.handle, p(corresp), p(calm_cm), stride, B, N, p(Rt2), p(Rt3), p(T), p(rec), p(it), p(st),
                                p(dbg)), stem + "_debug_dev")
            out["debug"] = dbg
        else:
            fn = getattr(self.lib, stem + "_dev")
            _check(self.lib, fn(self.handle, p(corresp), p(calm_cm), stride, B, N, p(Rt2), p(Rt3), p(T), p(rec), p(it), p(st)),
                   stem + "_dev")
        out.update(R_t_2=Rt2.reshape(B, 4, 3).transpose(1, 2), R_t_3=Rt3.reshape(B, 4, 3).transpose(1, 2),
                   T=T.reshape(B, 3, 3, 3).permute(0, 3, 2, 1),
                   Reconst=rec.transpose(1, 2) if reconst else None, iter=it, status=st,
                   _raw=(Rt2, Rt3, T, rec))
        return out


    # ---- building blocks (torch CUDA tensors or numpy arrays in; torch CUDA tensors out) ------------
    def _t(self, a, dtype=None):
        dtype = dtype or torch.float64
        if isinstance(a, np.ndarray):
            a = torch.from_numpy(np.ascontiguousarray(a))
        return a.to(device=torch.device("cuda", self.device), dtype=dtype).contiguous()

    def _begin(self):
        self.set_stream(torch.cuda.current_stream(torch.device("cuda", self.device)).cuda_stream)

    @staticmethod
    def _p(t):
        return ctypes.c_void_p(t.data_ptr()) if t is not None else None

    @staticmethod
    def _cams_cm(cams):
        """(..., 3, 4) row-major numpy/torch cameras -> column-major flat layout of the ABI."""
        return cams.transpose(-1, -2).contiguous()

    def triangulate(self, cams, pts):
        """triangulation3D: cams (B, M, 3, 4) or (M, 3, 4) shared; pts (B, N, 2M).  -> (B, 4, N) unit homogeneous."""
        self._begin()
        pts = self._t(pts); B, N, M2 = pts.shape; M = M2 // 2
        cams = self._cams_cm(self._t(cams))
        stride = 12 * M if cams.dim() == 4 else 0
        X = torch.empty((B, N, 4), dtype=torch.float64, device=pts.device)
        _check(self.lib, self.lib.tff_triangulate_batch_dev(self.handle, self._p(cams), stride, self._p(pts), B, M, N, self._p(X)),
               "tff_triangulate_batch_dev")
        return X.transpose(1, 2)

    def repr_error(self, cams, corresp, pts3d=None):
        """ReprError: cams (B, 3, 3, 4) or (3, 3, 4); corresp (B, N, 6) or (N, 6) shared; pts3d (B, 3, N) or None."""
        self._begin()
        cams = self._cams_cm(self._t(cams))
        cstride = 36 if cams.dim() == 4 else 0
        corresp = self._t(corresp)
        N = corresp.shape[-2]
        B = cams.shape[0] if cstride else (corresp.shape[0] if corresp.dim() == 3 else 1)
        pstride = 6 * N if corresp.dim() == 3 else 0
        p3 = self._t(pts3d).transpose(1, 2).contiguous() if pts3d is not None else None
        err = torch.empty(B, dtype=torch.float64, device=corresp.device)
        _check(self.lib, self.lib.tff_repr_error_batch_dev(self.handle, self._p(cams), cstride, self._p(corresp), pstride,
                                                           self._p(p3), B, N, self._p(err)), "tff_repr_error_batch_dev")
        return err

    def inlier_count(self, scene, calm, R_t_2, R_t_3, threshold=1.0, with_error=False):
        """Inlier counts (experiments_real.m:94-98 rule) of B pose hypotheses against one scene (Ns, 6)."""
        self._begin()
        scene = self._t(scene); Ns = scene.shape[0]
        calm = self._t(calm).t().contiguous().reshape(27)
        r2 = self._cams_cm(self._t(R_t_2)); r3 = self._cams_cm(self._t(R_t_3)); B = r2.shape[0]
        cnt = torch.empty(B, dtype=torch.int32, device=scene.device)
        err = torch.empty(B, dtype=torch.float64, device=scene.device) if with_error else None
        _check(self.lib, self.lib.tff_inlier_count_batch_dev(self.handle, self._p(scene), Ns, self._p(calm), self._p(r2), self._p(r3), B,
                                                             float(threshold), self._p(cnt), self._p(err)), "tff_inlier_count_batch_dev")
        return (cnt, err) if with_error else cnt

    def transform_tft(self, T, M1, M2, M3, inverse=0):
        """transform_TFT: T (B,3,3,3) indexed [b,j,k,i]; M1..M3 (3,3) shared or (B,3,3)."""
        self._begin()
        T = self._t(T); B = T.shape[0]
        Tv = T.permute(0, 3, 2, 1).contiguous()                               # -> flat index j + 3k + 9i
        Ms = [self._t(M).transpose(-1, -2).contiguous() for M in (M1, M2, M3)]
        stride = 9 if Ms[0].dim() == 3 else 0
        out = torch.empty((B, 27), dtype=torch.float64, device=T.device)
        _check(self.lib, self.lib.tff_transform_tft_batch_dev(self.handle, self._p(Tv), self._p(Ms[0]), self._p(Ms[1]), self._p(Ms[2]),
                                                              stride, B, int(inverse), self._p(out)), "tff_transform_tft_batch_dev")
        return out.reshape(B, 3, 3, 3).permute(0, 3, 2, 1)

    def rt_from_tft(self, T, calm, corresp):
        """R_t_from_TFT: T (B,3,3,3) [b,j,k,i] in pixel coordinates, calm (9,3), corresp (B,N,6)."""
        self._begin()
        T = self._t(T); B = T.shape[0]
        Tv = T.permute(0, 3, 2, 1).contiguous()
        corresp = self._t(corresp); N = corresp.shape[1]
        calm = self._t(calm).t().contiguous().reshape(27)
        Rt2 = torch.empty((B, 12), dtype=torch.float64, device=T.device); Rt3 = torch.empty_like(Rt2)
        st = torch.zeros(B, dtype=torch.int32, device=T.device)
        _check(self.lib, self.lib.tff_rt_from_tft_batch_dev(self.handle, self._p(Tv), self._p(calm), 0, self._p(corresp), B, N,
                                                            self._p(Rt2), self._p(Rt3), self._p(st)), "tff_rt_from_tft_batch_dev")
        return Rt2.reshape(B, 4, 3).transpose(1, 2), Rt3.reshape(B, 4, 3).transpose(1, 2), st

    def linear_tft(self, corresp):
        """linearTFT on the given (already normalised, if desired) points: corresp (B,N,6) -> T (B,3,3,3), P2, P3 (B,3,4)."""
        self._begin()
        corresp = self._t(corresp); B, N, _ = corresp.shape
        T = torch.empty((B, 27), dtype=torch.float64, device=corresp.device)
        P2 = torch.empty((B, 12), dtype=torch.float64, device=corresp.device); P3 = torch.empty_like(P2)
        st = torch.zeros(B, dtype=torch.int32, device=corresp.device)
        _check(self.lib, self.lib.tff_linear_tft_batch_dev(self.handle, self._p(corresp), B, N, self._p(T), self._p(P2), self._p(P3),
                                                           self._p(st)), "tff_linear_tft_batch_dev")
        return T.reshape(B, 3, 3, 3).permute(0, 3, 2, 1), P2.reshape(B, 4, 3).transpose(1, 2), P3.reshape(B, 4, 3).transpose(1, 2), st

    def linear_f(self, corresp, refine=False):
        """linearF (refine=False) / optimF (refine=True) for view pairs (1,2), (1,3): corresp (B,N,6) -> F21, F31 (B,3,3), iter, status."""
        self._begin()
        corresp = self._t(corresp); B, N, _ = corresp.shape
        F21 = torch.empty((B, 9), dtype=torch.float64, device=corresp.device); F31 = torch.empty_like(F21)
        it = torch.zeros(B, dtype=torch.int32, device=corresp.device); st = torch.zeros_like(it)
        _check(self.lib, self.lib.tff_linear_f_batch_dev(self.handle, self._p(corresp), B, N, int(bool(refine)), self._p(F21), self._p(F31),
                                                         self._p(it), self._p(st)), "tff_linear_f_batch_dev")
        return F21.reshape(B, 3, 3).transpose(1, 2), F31.reshape(B, 3, 3).transpose(1, 2), it, st

    def bundle_adjust(self, calm, R_t_2, R_t_3, corresp, reconst0=None):
        """BundleAdjustment for B triplets: calm (9,3) or (B,9,3); R_t_2, R_t_3 (B,3,4); corresp (B,N,6); reconst0 (B,3,N) or None.
        -> dict(R_t_2, R_t_3 (B,3,4), Reconst (B,3,N), iter, repr_err, status)."""
        self._begin()
        corresp = self._t(corresp); B, N, _ = corresp.shape
        dev = corresp.device
        calm = self._t(calm)
        if calm.dim() == 2:
            calm_cm, stride = calm.t().contiguous().reshape(27), 0
        else:
            calm_cm, stride = calm.transpose(1, 2).contiguous().reshape(B * 27), 27
        r2 = self._cams_cm(self._t(R_t_2)); r3 = self._cams_cm(self._t(R_t_3))
        x0 = self._t(reconst0).transpose(1, 2).contiguous() if reconst0 is not None else None
        o2 = torch.empty((B, 12), dtype=torch.float64, device=dev); o3 = torch.empty_like(o2)
        rec = torch.empty((B, N, 3), dtype=torch.float64, device=dev)
        it = torch.zeros(B, dtype=torch.int32, device=dev); st = torch.zeros_like(it)
        err = torch.empty(B, dtype=torch.float64, device=dev)
        _check(self.lib, self.lib.tff_bundle_adjust_batch_dev(self.handle, self._p(calm_cm), stride, self._p(r2), self._p(r3), self._p(corresp), B, N,
                                                              self._p(x0), self._p(o2), self._p(o3), self._p(rec), self._p(it), self._p(err), self._p(st)),
               "tff_bundle_adjust_batch_dev")
        return dict(R_t_2=o2.reshape(B, 4, 3).transpose(1, 2), R_t_3=o3.reshape(B, 4, 3).transpose(1, 2), Reconst=rec.transpose(1, 2),
                    iter=it, repr_err=err, status=st)

    def bundle_adjust_views(self, calm, R_t_0, corresp, reconst0=None):
        """BundleAdjustment for B problems of M = 2 .. 6 views (tff_bundle_adjust_views_batch_dev): calm (3M,3) or (B,3M,3); R_t_0 (B,3M,4),
        first camera included and free; corresp (B,N,2M), NaN = not seen (drops the whole view, as the reference's code does); reconst0 (B,3,N) or
        None.  -> dict(R_t (B,3M,4), Reconst (B,3,N), iter, repr_err, status)."""
        self._begin()
        corresp = self._t(corresp); B, N, M2 = corresp.shape
        M = M2 // 2
        dev = corresp.device
        calm = self._t(calm)
        if calm.dim() == 2:
            calm_cm, stride = calm.t().contiguous().reshape(9 * M), 0
        else:
            calm_cm, stride = calm.transpose(1, 2).contiguous().reshape(B * 9 * M), 9 * M
        rt = self._t(R_t_0)
        if M2 != 2 * M or tuple(rt.shape) != (B, 3 * M, 4) or calm_cm.numel() != (9 * M if stride == 0 else B * 9 * M):
            raise ValueError("M views: corresp (B,N,2M), R_t_0 (B,3M,4), calm (3M,3) or (B,3M,3)")
        rt_cm = rt.transpose(1, 2).contiguous()
        x0 = self._t(reconst0).transpose(1, 2).contiguous() if reconst0 is not None else None
        out = torch.empty((B, 4, 3 * M), dtype=torch.float64, device=dev)
        rec = torch.empty((B, N, 3), dtype=torch.float64, device=dev)
        it = torch.zeros(B, dtype=torch.int32, device=dev); st = torch.zeros_like(it)
        err = torch.empty(B, dtype=torch.float64, device=dev)
        _check(self.lib, self.lib.tff_bundle_adjust_views_batch_dev(self.handle, M, self._p(calm_cm), stride, self._p(rt_cm), self._p(corresp), B, N,
                                                                    self._p(x0), self._p(out), self._p(rec), self._p(it), self._p(err), self._p(st)),
               "tff_bundle_adjust_views_batch_dev")
        return dict(R_t=out.transpose(1, 2), Reconst=rec.transpose(1, 2), iter=it, repr_err=err, status=st)

    def pose_sampled(self, method, scene, calm, sample_idx):
        """Minimal-sample hypotheses (config 4): scene (Ns, 6), sample_idx (B, n) int32 -> R_t_2, R_t_3 (B,3,4), T, status."""
        self._begin()
        scene = self._t(scene); Ns = scene.shape[0]
        idx = self._t(sample_idx, torch.int32); B, n = idx.shape
        calm = self._t(calm).t().contiguous().reshape(27)
        Rt2 = torch.empty((B, 12), dtype=torch.float64, device=scene.device); Rt3 = torch.empty_like(Rt2)
        T = torch.empty((B, 27), dtype=torch.float64, device=scene.device)
        st = torch.zeros(B, dtype=torch.int32, device=scene.device)
        fn = {"LinearTFTPoseEstimation": self.lib.tff_linear_tft_pose_sampled_dev,
              "LinearFPoseEstimation": self.lib.tff_linear_f_pose_sampled_dev}[method]
        _check(self.lib, fn(self.handle, self._p(scene), Ns, self._p(calm), self._p(idx), B, n, self._p(Rt2), self._p(Rt3), self._p(T),
                            self._p(st)), "pose_sampled")
        return dict(R_t_2=Rt2.reshape(B, 4, 3).transpose(1, 2), R_t_3=Rt3.reshape(B, 4, 3).transpose(1, 2),
                    T=T.reshape(B, 3, 3, 3).permute(0, 3, 2, 1), status=st, _raw=(Rt2, Rt3, T))


class MultiContext:
    """A tff_multi: one process, one context + host thread + stream per device (include/tftfund.h, multi-GPU section).
    `devices`: list of HIP ordinals, or None for all visible devices."""

    def __init__(self, devices=None, lib_path=None):
        self.lib = load_library(lib_path)
        h = ctypes.c_void_p()
        if devices is None:
            arr, n = None, 0
        else:
            arr = (ctypes.c_int32 * len(devices))(*devices); n = len(devices)
        _check(self.lib, self.lib.tff_multi_create(ctypes.byref(h), arr, n), "tff_multi_create")
        self.handle = h
        self.size = int(self.lib.tff_multi_size(h))

    def shard(self, B, rank):
        b0, b1 = ctypes.c_int64(), ctypes.c_int64()
        self.lib.tff_multi_shard(self.handle, B, rank, ctypes.byref(b0), ctypes.byref(b1))
        return b0.value, b1.value

    def pose_batch(self, method, corresp, calm, reconst=True):
        """Host arrays in, host arrays out: corresp (B, N, 6), calm (9, 3) or (B, 9, 3); the shards run concurrently on the devices."""
        C = np.ascontiguousarray(corresp, dtype=np.float64)
        B, N, _ = C.shape
        calm = np.asarray(calm, dtype=np.float64)
        if calm.ndim == 2:
            cm, stride = np.ascontiguousarray(calm.T).reshape(27), 0
        else:
            cm, stride = np.ascontiguousarray(calm.transpose(0, 2, 1)).reshape(B, 27), 27
        Rt2 = np.empty((B, 12)); Rt3 = np.empty((B, 12)); T = np.empty((B, 27))
        Rec = np.empty((B, N, 3)) if reconst else None
        it = np.zeros(B, dtype=np.int32); st = np.zeros(B, dtype=np.int32)
        p = lambda a: ctypes.c_void_p(a.ctypes.data) if a is not None else None
        _check(self.lib, self.lib.tff_pose_batch_host_multi(self.handle, METHOD_IDS[method], p(C), p(cm), stride, B, N, p(Rt2), p(Rt3), p(T),
                                                            p(Rec), p(it), p(st)), "tff_pose_batch_host_multi")
        return dict(R_t_2=Rt2.reshape(B, 4, 3).transpose(0, 2, 1), R_t_3=Rt3.reshape(B, 4, 3).transpose(0, 2, 1),
                    T=T.reshape(B, 3, 3, 3).transpose(0, 3, 2, 1), Reconst=None if Rec is None else Rec.transpose(0, 2, 1), iter=it, status=st)

    def pose_batch_dev(self, method, corresp_shards, calm_shards, B):
        """Device-resident shards (torch CUDA tensors, shard g on device g, each (b1-b0, N, 6)); returns per device the gathered
        record tensor (G * chunk, 51) laid out as G blocks [Rt2 (chunk x 12) | Rt3 (chunk x 12) | T (chunk x 27)] and the status."""
        import torch
        G = self.size
        if B < 0 or len(corresp_shards) != G or len(calm_shards) != G:
            raise ValueError("pose_batch_dev needs B >= 0 and one shard per device")
        N = int(corresp_shards[0].shape[1])
        chunk = (B + G - 1) // G
        for g in range(G):
            b0, b1 = self.shard(B, g)
            c = corresp_shards[g]
            if c.dtype != torch.float64 or not c.is_contiguous() or tuple(c.shape) != (b1 - b0, N, 6):
                raise ValueError("shard %d must be a contiguous float64 tensor of shape (%d, %d, 6)" % (g, b1 - b0, N))
        recs = [torch.zeros(G * chunk * 51, dtype=torch.float64, device=corresp_shards[g].device) for g in range(G)]
        sts = [torch.zeros(G * chunk, dtype=torch.int32, device=corresp_shards[g].device) for g in range(G)]
        cms = [c.t().contiguous().reshape(27) for c in calm_shards]
        # torch enqueued the allocations / transposes above on ITS current stream of each device; the library works on each context's
        # stream.  Hand every context torch's stream of its device (tff_ctx_set_stream orders the hand-over with an event), so that
        # zero-fill -> kernels -> all-gather are one stream's program order and later torch work on the results is ordered too.
        for g in range(G):
            dev = corresp_shards[g].device
            _check(self.lib, self.lib.tff_ctx_set_stream(self.lib.tff_multi_ctx(self.handle, g), ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)),
                   "tff_ctx_set_stream")
        ptr = lambda ts: (ctypes.c_void_p * G)(*[t.data_ptr() for t in ts])
        _check(self.lib, self.lib.tff_pose_batch_dev_multi(self.handle, METHOD_IDS[method], ptr(corresp_shards), ptr(cms), 0, B, N, ptr(recs), ptr(sts)),
               "tff_pose_batch_dev_multi")
        for g in range(G):
            _check(self.lib, self.lib.tff_ctx_synchronize(self.lib.tff_multi_ctx(self.handle, g)), "synchronize")
        # everything is complete: give the borrowed torch streams back (a later call on these contexts -- pose_batch, another thread's
        # pose_batch_dev under a different torch stream -- must not run on a stream the caller may have destroyed meanwhile)
        for g in range(G):
            _check(self.lib, self.lib.tff_ctx_use_own_stream(self.lib.tff_multi_ctx(self.handle, g)), "tff_ctx_use_own_stream")
        return recs, sts, chunk

    def close(self):
        if getattr(self, "handle", None):
            self.lib.tff_multi_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_default_ctx = threading.local()


def default_context(device=0):
    """One context per (thread, device): a context's workspaces are shared state, so threads do not share one."""
    d = getattr(_default_ctx, "by_device", None)
    if d is None:
        d = _default_ctx.by_device = {}
    c = d.get(device)
    if c is None:
        c = d[device] = Context(device)
    return c


def _single(method, Corresp, CalM, nargout=5):
    """One triplet with the reference's shapes: Corresp 6xN, CalM 9x3 ->
    R_t_2 (3x4), R_t_3 (3x4), Reconst (3xN), T (3x3x3), iter."""
    Corresp = np.asarray(Corresp, dtype=np.float64)
    if Corresp.ndim != 2 or Corresp.shape[0] != 6:
        raise ValueError("Corresp must be 6xN")
    N = Corresp.shape[1]
    out = default_context().pose_batch(method, np.ascontiguousarray(Corresp.T).reshape(1, N, 6), np.asarray(CalM),
                                       reconst=nargout >= 3)
    st = int(out["status"][0])
    if st == ST_TOO_FEW:
        raise ValueError("not enough correspondences for %s (N=%d)" % (method, N))
    if st == ST_NO_POSE:
        raise RuntimeError("%s: no pose candidate with non-negative cheirality score" % method)
    if st == ST_NO_PARAM:
        raise ValueError("The minimal param could not be found")
    rec = out["Reconst"][0] if out["Reconst"] is not None else None
    return out["R_t_2"][0], out["R_t_3"][0], rec, out["T"][0], int(out["iter"][0])


def LinearTFTPoseEstimation(Corresp, CalM):
    """Drop-in for TFT_methods/LinearTFTPoseEstimation.m (same inputs/outputs)."""
    return _single("LinearTFTPoseEstimation", Corresp, CalM)


def LinearFPoseEstimation(Corresp, CalM):
    """Drop-in for F_methods/LinearFPoseEstimation.m (same inputs/outputs; needs N >= 8)."""
    return _single("LinearFPoseEstimation", Corresp, CalM)


def ResslTFTPoseEstimation(Corresp, CalM):
    """Drop-in for TFT_methods/ResslTFTPoseEstimation.m (iter = Gauss-Helmert iterations)."""
    return _single("ResslTFTPoseEstimation", Corresp, CalM)


def FaugPapaTFTPoseEstimation(Corresp, CalM):
    """Drop-in for TFT_methods/FaugPapaTFTPoseEstimation.m (iter = Gauss-Helmert iterations)."""
    return _single("FaugPapaTFTPoseEstimation", Corresp, CalM)


def NordbergTFTPoseEstimation(Corresp, CalM):
    """Drop-in for TFT_methods/NordbergTFTPoseEstimation.m (iter = Gauss-Helmert iterations)."""
    return _single("NordbergTFTPoseEstimation", Corresp, CalM)


def OptimFPoseEstimation(Corresp, CalM):
    """Drop-in for F_methods/OptimFPoseEstimation.m (iter = it1 + it2 Gauss-Helmert iterations of the two optimF calls)."""
    return _single("OptimFPoseEstimation", Corresp, CalM)


def PiPoseEstimation(Corresp, CalM):
    """Drop-in for TFT_methods/PiPoseEstimation.m (iter = Gauss-Helmert iterations)."""
    return _single("PiPoseEstimation", Corresp, CalM)


def PiColPoseEstimation(Corresp, CalM):
    """Drop-in for TFT_methods/PiColPoseEstimation.m (collinear camera centres; iter = Gauss-Helmert iterations)."""
    return _single("PiColPoseEstimation", Corresp, CalM)


def BundleAdjustment(CalM, R_t_0, Corresp, Reconst0=None):
    """Drop-in for Optimization/BundleAdjustment.m, M = 2 .. 6 views: CalM 3Mx3, R_t_0 3Mx4, Corresp 2MxN (NaN = not seen), Reconst0 3xN or None
    -> R_t (3Mx4, first camera [I|0], |t2| = 1), Reconst (3xN), iter, repr_err.  The initial triangulation (:59-77), the change of coordinates to
    camera 1 (:80-86) and the `isnan` branch (:165 -- it drops a whole VIEW, see csrc/ba_views_kernel.h) run on the device as the reference orders them.
    Raises ValueError where the reference stops with an error (fewer than two complete views to triangulate from, :73-74)."""
    CalM = np.asarray(CalM, dtype=np.float64); R_t_0 = np.asarray(R_t_0, dtype=np.float64); Corresp = np.asarray(Corresp, dtype=np.float64)
    if Corresp.ndim != 2 or Corresp.shape[0] % 2 or not 2 <= Corresp.shape[0] // 2 <= 6:
        raise ValueError("Corresp must be 2M x N with M = 2 .. 6 views")
    M = Corresp.shape[0] // 2
    if R_t_0.shape != (3 * M, 4) or CalM.shape != (3 * M, 3):
        raise ValueError("%d views: R_t_0 must be %dx4 and CalM %dx3" % (M, 3 * M, 3 * M))
    X0 = None if Reconst0 is None else np.asarray(Reconst0, dtype=np.float64)[None]
    out = default_context().bundle_adjust_views(CalM, R_t_0[None], np.ascontiguousarray(Corresp.T)[None], X0)
    if int(out["status"][0]) == 1:
        raise ValueError("fewer than two complete views: triangulation3D returns nothing (triangulation3D.m:36-38) and BundleAdjustment.m:73-74 stops")
    return out["R_t"][0].cpu().numpy(), out["Reconst"][0].cpu().numpy(), int(out["iter"][0]), float(out["repr_err"][0])
