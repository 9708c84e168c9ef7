"""TEST INFRASTRUCTURE ONLY: ctypes loader of oracle/liboracle_c.so (plain-C restatement)."""
import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "liboracle_c.so")


def build():
    src = os.path.join(HERE, "tft_oracle_c.c")
    if not os.path.exists(SO) or os.path.getmtime(SO) < os.path.getmtime(src):
        subprocess.run(["make", "-C", HERE, "liboracle_c.so"], check=True, stdout=subprocess.DEVNULL)
    return SO


def load():
    lib = ctypes.CDLL(build())
    lib.oracle_c_linear_tft_pose_batch.restype = ctypes.c_int
    return lib


def linear_f_pose_batch(C, CalM, reconst=True, threads=0):
    """LinearFPoseEstimation (plain C); same conventions as linear_tft_pose_batch."""
    return linear_tft_pose_batch(C, CalM, reconst, threads, entry="oracle_c_linear_f_pose_batch")


def linear_tft_pose_batch(C, CalM, reconst=True, threads=0, entry="oracle_c_linear_tft_pose_batch"):
    """C: (B,N,6); CalM (9,3).  Returns dict like api.Context.pose_batch plus 'threads'."""
    lib = load()
    C = np.ascontiguousarray(C, dtype=np.float64)
    B, N, _ = C.shape
    calm = np.ascontiguousarray(np.asarray(CalM, dtype=np.float64).T).reshape(27)
    Rt2 = np.zeros((B, 12)); Rt3 = np.zeros((B, 12)); T = np.zeros((B, 27))
    Rec = np.zeros((B, N, 3)) if reconst else None
    st = np.zeros(B, dtype=np.int32)
    p = lambda a: ctypes.c_void_p(a.ctypes.data) if a is not None else None
    fn = getattr(lib, entry)
    fn.restype = ctypes.c_int
    used = fn(p(C), p(calm), ctypes.c_long(0), ctypes.c_long(B), ctypes.c_int(N), p(Rt2), p(Rt3),
                                              p(T), p(Rec), p(st), ctypes.c_int(threads))
    return dict(R_t_2=Rt2.reshape(B, 4, 3).transpose(0, 2, 1), R_t_3=Rt3.reshape(B, 4, 3).transpose(0, 2, 1),
                T=T.reshape(B, 3, 3, 3).transpose(0, 3, 2, 1), Reconst=None if Rec is None else Rec.transpose(0, 2, 1),
                status=st, threads=used)
