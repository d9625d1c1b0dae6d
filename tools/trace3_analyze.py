"""Timeline of the three-batches-in-flight step from a rocprofv3 kernel trace (tools/trace3.sh): per kernel its duration under
contention, how much of the region each kernel family is running, the union of busy time and the gaps."""
import csv, sys
import numpy as np
rows = list(csv.DictReader(open(sys.argv[1])))
K = []
for r in rows:
    n = r["Kernel_Name"]
    fam = "softattn" if "softattn_kernel" in n else "maxpath" if "maxpath_pipelined" in n else "expand" if "expand_kernel" in n else None
    if fam: K.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), fam, r.get("Stream_Id", r.get("Queue_Id", "?"))))
K.sort()
# the timed region: the longest run of kernels whose starts are < 100 us apart; its middle 60 %
runs, cur = [], [K[0]]
for k in K[1:]:
    if k[0] - cur[-1][0] < 100_000: cur.append(k)
    else: runs.append(cur); cur = [k]
runs.append(cur)
R = max(runs, key=len)
t0, t1 = R[0][0], R[-1][1]
a, b = t0 + 0.2 * (t1 - t0), t0 + 0.8 * (t1 - t0)
S = [k for k in R if k[0] >= a and k[1] <= b]
print("kernels in the steady window:", len(S), "window us:", (b - a) / 1e3)
for fam in ("softattn", "maxpath", "expand"):
    d = np.array([k[1] - k[0] for k in S if k[2] == fam]) / 1e3
    print(f"{fam:9s}: n {len(d):4d}  duration us  median {np.median(d):6.1f}  p10 {np.percentile(d,10):6.1f}  p90 {np.percentile(d,90):6.1f}")
ev = sorted([(k[0], 1, k[2]) for k in S] + [(k[1], -1, k[2]) for k in S])
cnt = {"softattn": 0, "maxpath": 0, "expand": 0}
last = ev[0][0]; busy = 0; hist = {}
for t, d, fam in ev:
    key = (min(cnt["softattn"], 3), min(cnt["maxpath"], 3), min(cnt["expand"], 3))
    hist[key] = hist.get(key, 0) + (t - last)
    if sum(cnt.values()) > 0: busy += t - last
    last = t
    cnt[fam] += d
tot = ev[-1][0] - ev[0][0]
print("some kernel running: %.1f %% of the window" % (100 * busy / tot))
print("time share by (softattn, maxpath, expand) kernels running at once:")
for k, v in sorted(hist.items(), key=lambda kv: -kv[1])[:12]:
    print("   ", k, "%.1f %%" % (100 * v / tot))
nst = len([k for k in S if k[2] == "softattn"])
print("steps in window:", nst, "-> us per step:", (b - a) / 1e3 / max(nst, 1))
