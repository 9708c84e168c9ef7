"""Converter (build container, needs /root/reference and scipy): Data/<dataset>/Corresp_triplets.mat + *.camera of the reference
-> tests/golden/epfl_all.npz, an INPUTS-ONLY fixture of ALL non-empty triplets of fountain-P11 (150) and Herz-Jesu-P8 (56) in the
order of `indexes_sorted` (most matches first) -- experiments_real.m:75-82 takes the first 70 / 50 of that list.
Per dataset <d> in {fountain, herzjesu}:
  <d>_triplets  (T, 4) int32   im1, im2, im3 (1-based, as in the .mat), number of matches
  <d>_offsets   (T + 1,) int64 rows of <d>_corresp that belong to triplet t: [offsets[t], offsets[t+1])
  <d>_corresp   (sum N, 6) float64   x1 y1 x2 y2 x3 y3 in pixels   (Corresp{im1,im2,im3}, experiments_real.m:80)
  <d>_K, <d>_R, <d>_t  per IMAGE: calibration, rotation, translation  (Data/readCalibrationOrientation_EPFL.m)
Usage: python tests/golden/make_epfl_all.py"""
import os
import numpy as np
import scipy.io

REF = "/root/reference/Data"
HERE = os.path.dirname(os.path.abspath(__file__))


def read_camera(path):
    with open(path) as f:
        rows = [[float(v) for v in line.split()] for line in f.read().strip().splitlines()]
    K = np.array(rows[0:3]); R = np.array(rows[4:7]).T
    return K, R, -R @ np.array(rows[7])


out = {}
for ds, key in (("fountain-P11", "fountain"), ("Herz-Jesu-P8", "herzjesu")):
    m = scipy.io.loadmat(os.path.join(REF, ds, "Corresp_triplets.mat"))
    names = [str(x[0]) for x in m["im_names"].ravel()]
    order = np.asarray(m["indexes_sorted"]).astype(np.int64)
    trips, chunks = [], []
    for (i1, i2, i3, n) in order:
        c = np.asarray(m["Corresp"][i1 - 1, i2 - 1, i3 - 1], dtype=np.float64)
        if c.size == 0:
            continue
        assert c.shape == (n, 6)
        trips.append((i1, i2, i3, n)); chunks.append(c)
    cams = [read_camera(os.path.join(REF, ds, nm + ".camera")) for nm in names]
    out[key + "_triplets"] = np.array(trips, dtype=np.int32)
    out[key + "_offsets"] = np.concatenate([[0], np.cumsum([c.shape[0] for c in chunks])]).astype(np.int64)
    out[key + "_corresp"] = np.concatenate(chunks, axis=0)
    out[key + "_K"] = np.stack([c[0] for c in cams]); out[key + "_R"] = np.stack([c[1] for c in cams]); out[key + "_t"] = np.stack([c[2] for c in cams])
    print(ds, len(trips), "triplets,", out[key + "_corresp"].shape[0], "matches")
np.savez_compressed(os.path.join(HERE, "epfl_all.npz"), **out)
print(os.path.getsize(os.path.join(HERE, "epfl_all.npz")), "bytes")
