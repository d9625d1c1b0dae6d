"""The similarity kernel with its log-probs at a row pitch of whole 128-byte lines (pitched_logp) against the contiguous
layout: same values, time back to back (HIP events), alternating."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import aligner_amd
from aligner_amd.softattn import pitched_logp
dev = torch.device("cuda:0")
B, C, Tx, Ty = (int(v) for v in os.environ.get("SA_SHAPE", "64,80,200,1000").split(","))
g = torch.Generator().manual_seed(0)
k = torch.randn(B, C, Tx, generator=g).to(dev); q = torch.randn(B, C, Ty, generator=g).to(dev)
flat = torch.empty((B, Tx, Ty), device=dev)
pit = pitched_logp(B, Tx, Ty, dev)
aligner_amd.soft_attention(k, q, out=flat); aligner_amd.soft_attention(k, q, out=pit)
torch.cuda.synchronize()
print("row pitch", pit.stride(1), "equal:", bool(torch.equal(flat, pit)))
def timeit(out, n=300):
    for _ in range(20): aligner_amd.soft_attention(k, q, out=out)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): aligner_amd.soft_attention(k, q, out=out)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for r in range(4):
    print(f"round {r}: contiguous {timeit(flat):6.2f} us   pitched {timeit(pit):6.2f} us")
