"""Experiment: does giving the DP kernel a high-priority stream improve pipelined throughput?"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from bench import Step, B, TX, TY
dev = torch.device("cuda:0")
def run(S, prio, steps=300, warm=30, only=None):
    st = [Step(dev, 1 + i, use_graph=False) for i in range(S)]
    lo = [torch.cuda.Stream(dev) for _ in range(S)]
    hi = [torch.cuda.Stream(dev, priority=-1) if prio else lo[i] for i in range(S)]
    for s in st: s.eager()
    torch.cuda.synchronize()
    def go(n):
        for i in range(n):
            k = i % S
            with torch.cuda.stream(lo[k]):
                if only in (None, "sa", "sa+ex"): st[k].softattn()
            if only in (None, "dp"):
                hi[k].wait_stream(lo[k])
                with torch.cuda.stream(hi[k]):
                    st[k].forward()
                lo[k].wait_stream(hi[k])
            with torch.cuda.stream(lo[k]):
                if only in (None, "sa+ex"): st[k].expand()
    go(warm); torch.cuda.synchronize()
    t0 = time.perf_counter(); go(steps); torch.cuda.synchronize(); t = time.perf_counter() - t0
    return t / steps * 1e6
for S in (1, 2, 3, 4):
    print("S", S, "equal prio us/step", round(run(S, False), 2), " DP high prio", round(run(S, True), 2),
          " | only softattn", round(run(S, False, only="sa"), 2), " only DP", round(run(S, False, only="dp"), 2),
          " softattn+expand", round(run(S, False, only="sa+ex"), 2), flush=True)
