"""In-kernel timeline of softattn_rt_kernel (row-tile form) at C2 and its time against the strip-per-wave form
(debug option softattn_strips): shader-clock stamps per wave (aligner_debug_set_stamps)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import aligner_amd
from aligner_amd import _lib
lib = _lib.load(); dev = torch.device("cuda:0")
B, C, Tx, Ty = (int(v) for v in os.environ.get("SA_SHAPE", "64,80,200,1000").split(","))
g = torch.Generator().manual_seed(0)
k = torch.randn(B, C, Tx, generator=g).to(dev); q = torch.randn(B, C, Ty, generator=g).to(dev)
out = torch.empty((B, Tx, Ty), device=dev); ref = torch.empty_like(out)
lib.aligner_debug_set_option(b"softattn_strips", 1)
aligner_amd.soft_attention(k, q, out=ref)
lib.aligner_debug_set_option(b"softattn_strips", 0)
aligner_amd.soft_attention(k, q, out=out)
torch.cuda.synchronize()
fin = torch.isfinite(ref)
print("max |row-tile - strips| =", (out[fin] - ref[fin]).abs().max().item(), " finite pattern equal:", bool(torch.equal(torch.isfinite(out), fin)))

def timeit(n=200):
    for _ in range(20): aligner_amd.soft_attention(k, q, out=out)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): aligner_amd.soft_attention(k, q, out=out)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for r in range(3):
    lib.aligner_debug_set_option(b"softattn_strips", 1); t_old = timeit()
    lib.aligner_debug_set_option(b"softattn_strips", 0); t_new = timeit()
    print(f"round {r}: strips {t_old:6.2f} us   row-tile {t_new:6.2f} us  (back-to-back calls, launch overhead included)")

NW = 8
nblk = 4096
st = torch.zeros((nblk, NW, 64), dtype=torch.int64, device=dev)
lib.aligner_debug_set_stamps(st.data_ptr())
aligner_amd.soft_attention(k, q, out=out)
torch.cuda.synchronize()
lib.aligner_debug_set_stamps(None)
s = st.cpu().numpy().astype(np.float64)
used = s[:, 0, 0] > 0
s = s[used]
names = {0: "entry", 1: "staged", 2: "first dots", 3: "first stores", 5: "last stores issued", 6: "drained"}
print("workgroups stamped:", s.shape[0], "(cycles since the workgroup's first entry stamp; median / max over workgroups)")
t0 = s[:, :, 0].min(axis=1)
for w in range(NW):
    row = []
    for kk in (0, 1, 2, 3, 5, 6):
        v = s[:, w, kk]
        ok = v > 0
        if ok.any():
            d = (v - t0)[ok]
            row.append(f"{names[kk]} {np.median(d):6.0f}/{d.max():6.0f}")
    print(f"wave {w}{' (loader)' if w == 0 else ''}: " + " | ".join(row))
rt0, rt1 = s[:, :, 7], s[:, 1:, 4]
print("100 MHz clock: first entry -> last entry %.2f us, first entry -> last drained %.2f us" %
      ((rt0.max() - rt0.min()) / 100.0, (rt1.max() - rt0.min()) / 100.0))
end = np.sort((rt1.max(axis=1) - rt0.min()) / 100.0)
print("workgroup drained times (us), deciles:", np.round(end[:: max(1, len(end) // 10)], 2))

# per-strip stamps: 8+4j before barrier j, 9+4j after it, 10+4j statistics merged, 11+4j stores issued
print("per strip (median cycles since entry): wave 1 | wave 5: logits + maximum done / slots done / arrive at barrier / leave | loader: arrive / leave")
for j in range(8):
    def med(w, kk):
        v = s[:, w, kk]; ok = v > 0
        return np.median((v - t0)[ok]) if ok.any() else float("nan")
    print(f"strip {j}: " + " | ".join(" ".join(f"{med(w, 8 + 4 * j + i):7.0f}" for i in ((2, 3, 0, 1) if w else (0, 1))) for w in (1, 5, 0)))

def medl(kk):
    v = s[:, 0, kk]; ok = v > 0
    return np.median((v - t0)[ok]) if ok.any() else float("nan")
print("loader prologue (median cycles): strip 0 in the ring, strip 1 asked for %.0f | strip 2 issued %.0f | strip 1 in the ring %.0f" % (medl(1), medl(11), medl(14)))
