"""Per-dispatch evidence of the two-stream overlap behind bench.py's `value`.

  python tools/overlap_trace.py <dir with *_kernel_trace.csv> <kernel substring> <out prefix>

Reads a `rocprofv3 --kernel-trace --output-format csv` trace of `python3 bench.py ...` (the program directly after `--`), keeps the dispatches of
the named kernel and writes
  <out prefix>_dispatches.csv  : dispatch id, queue, start / end in ns relative to the first kept dispatch, duration, and the time during which ANOTHER
                                 dispatch of the same kernel was running (its neighbour on the other stream);
  <out prefix>_overlap.txt     : how many dispatches overlap a neighbour, by how much, and the resulting start-to-start period inside the
                                 two-stream bursts (what `ms_per_step` measures) against the duration of a dispatch alone (`roofline.kernel_ms`)."""
import csv
import glob
import os
import sys

import numpy as np


def main():
    d, kern, out = sys.argv[1], sys.argv[2], sys.argv[3]
    rows = []
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if kern in r["Kernel_Name"] and "exact" not in r["Kernel_Name"]:
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r["Queue_Id"]), int(r["Dispatch_Id"])))
    rows.sort()
    if not rows:
        print("no dispatch of", kern); return
    t0 = rows[0][0]
    S = np.array([r[0] for r in rows], dtype=np.int64) - t0
    E = np.array([r[1] for r in rows], dtype=np.int64) - t0
    Q = np.array([r[2] for r in rows])
    n = len(rows)
    ov = np.zeros(n)
    for i in range(n):                                   # time of dispatch i covered by other dispatches of the kernel
        lo = np.maximum(S, S[i]); hi = np.minimum(E, E[i])
        o = np.clip(hi - lo, 0, None).astype(float); o[i] = 0
        ov[i] = o.sum()
    dur = (E - S).astype(float)
    with open(out + "_dispatches.csv", "w") as f:
        f.write("dispatch_id,queue_id,start_ns,end_ns,duration_ns,overlapped_by_neighbours_ns\n")
        for i in range(n):
            f.write("%d,%d,%d,%d,%d,%d\n" % (rows[i][3], Q[i], S[i], E[i], dur[i], ov[i]))
    alone = ov == 0
    # bursts: maximal runs of dispatches in which each starts before the previous one ends
    bursts, cur = [], [0]
    for i in range(1, n):
        if S[i] < E[cur[0]:i].max():
            cur.append(i)
        else:
            bursts.append(cur); cur = [i]
    bursts.append(cur)
    lines = []
    lines.append("kernel %s: %d dispatches on %d queues" % (kern, n, len(set(Q.tolist()))))
    lines.append("dispatches that run alone: %d, mean duration %.1f us (min %.1f, max %.1f)" %
                 (alone.sum(), dur[alone].mean() / 1e3 if alone.any() else float("nan"), dur[alone].min() / 1e3 if alone.any() else float("nan"),
                  dur[alone].max() / 1e3 if alone.any() else float("nan")))
    lines.append("dispatches that overlap a neighbour: %d, mean duration %.1f us, of which a neighbour is resident for %.1f us on average (%.0f %%)" %
                 ((~alone).sum(), dur[~alone].mean() / 1e3 if (~alone).any() else float("nan"), ov[~alone].mean() / 1e3 if (~alone).any() else float("nan"),
                  100 * (ov[~alone] / dur[~alone]).mean() if (~alone).any() else float("nan")))
    big = [b for b in bursts if len(b) >= 4]
    for b in big[:12]:
        span = E[b].max() - S[b[0]]
        lines.append("burst of %3d overlapping dispatches (queues %s): %.1f us from first start to last end = %.1f us per dispatch; mean dispatch duration %.1f us"
                     % (len(b), sorted(set(Q[b].tolist())), span / 1e3, span / 1e3 / len(b), dur[b].mean() / 1e3))
    if big:
        per = np.array([(E[b].max() - S[b[0]]) / len(b) for b in big]) / 1e3
        lines.append("per-dispatch period inside the overlapping bursts: median %.1f us (this is what bench.py reports as ms_per_step); "
                     "a dispatch alone: %.1f us (roofline.kernel_ms); overlap factor %.2f"
                     % (np.median(per), dur[alone].mean() / 1e3 if alone.any() else float("nan"), (dur[alone].mean() / 1e3 / np.median(per)) if alone.any() else float("nan")))
    open(out + "_overlap.txt", "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
