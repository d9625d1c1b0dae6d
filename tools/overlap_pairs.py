"""Do two of the step's kernels slow each other down when they run on different streams?  Each kernel alone, then pairs
(N launches each, back to back on two streams): time of the pair against the longer of the two alone."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from aligner_amd import _lib
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
st = [bench.Step(dev, seed=1 + i, use_graph=False, stream_path=True) for i in range(2)]
for s in st: s.softattn(); s.forward(); s.expand()
torch.cuda.synchronize()
N = 40
streams = [torch.cuda.Stream(dev), torch.cuda.Stream(dev)]
def run(fa, fb):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for f, strm in ((fa, streams[0]), (fb, streams[1])):
        if f is None: continue
        with torch.cuda.stream(strm):
            for _ in range(N): f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / N * 1e6
K = {"similarity": (st[0].softattn, st[1].softattn), "search": (st[0].forward, st[1].forward), "path": (st[0].expand, st[1].expand)}
alone = {}
for k, (f0, _) in K.items():
    run(f0, None); alone[k] = min(run(f0, None) for _ in range(3)); print(f"{k:10s} alone: {alone[k]:6.1f} us per launch")
names = list(K)
for i, a in enumerate(names):
    for b in names[i:]:
        run(K[a][0], K[b][1]); t = min(run(K[a][0], K[b][1]) for _ in range(3))
        print(f"{a:10s} + {b:10s}: {t:6.1f} us per pair   (longer alone {max(alone[a], alone[b]):5.1f}, sum {alone[a] + alone[b]:5.1f})")
