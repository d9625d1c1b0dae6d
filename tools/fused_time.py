#!/usr/bin/env python3
"""Fused vs unfused similarity + alignment search at the headline shape: HIP events around the C-ABI calls,
alone and with several independent batches in flight (what bench.py's step does).  Development aid."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aligner_amd import _lib
lib = _lib.load(); dev = torch.device("cuda:0")
B, C, TX, TY = 64, 80, 200, 1000


class S:
    def __init__(self, seed):
        g = torch.Generator().manual_seed(seed)
        self.k = torch.randn(B, C, TX, generator=g).to(dev); self.q = torch.randn(B, C, TY, generator=g).to(dev)
        self.tx = torch.full((B,), TX, dtype=torch.int32, device=dev); self.ty = torch.full((B,), TY, dtype=torch.int32, device=dev)
        self.logp = torch.empty((B, TX, TY), device=dev); self.path = torch.empty((B, TX, TY), device=dev)
        self.tok = torch.empty((B, TY), dtype=torch.int32, device=dev); self.dur = torch.empty((B, TX), dtype=torch.int32, device=dev)
        self.ws = torch.zeros(lib.aligner_maxpath_workspace_bytes(B, TX, TY) + 256, dtype=torch.uint8, device=dev)
        self.sws = torch.empty(lib.aligner_softattn_workspace_bytes(B, C, TX) + 256, dtype=torch.uint8, device=dev)
    def st(self): return torch.cuda.current_stream(dev).cuda_stream
    def fused(self, logp=True):
        _lib.check(lib.aligner_fused_align_f32(self.k.data_ptr(), self.q.data_ptr(), self.tx.data_ptr(), self.ty.data_ptr(),
                                               self.logp.data_ptr() if logp else None, self.tok.data_ptr(), self.dur.data_ptr(),
                                               self.ws.data_ptr(), self.ws.numel(), B, C, TX, TY, 0.0005, 0, -1e9, self.st()))
    def unfused(self):
        _lib.check(lib.aligner_softattn_f32(self.k.data_ptr(), self.q.data_ptr(), self.tx.data_ptr(), None, self.logp.data_ptr(), None,
                                            self.sws.data_ptr(), self.sws.numel(), B, C, TX, TY, 0.0005, 0, self.st()))
        _lib.check(lib.aligner_maxpath_forward_f32(self.logp.data_ptr(), None, 0, self.tx.data_ptr(), self.ty.data_ptr(), self.tok.data_ptr(),
                                                   self.dur.data_ptr(), self.ws.data_ptr(), self.ws.numel(), B, TX, TY, -1e9, 0, self.st()))
    def expand(self):
        _lib.check(lib.aligner_maxpath_expand(self.ws.data_ptr(), self.path.data_ptr(), 0, B, TX, TY, self.st()))


def ev(fn, it=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / it * 1e3


a = S(1)
print("fused (logp written) %.1f us | fused (no logp) %.1f us | unfused softattn + DP %.1f us | expand %.1f us" % (
    ev(a.fused), ev(lambda: a.fused(False)), ev(a.unfused), ev(a.expand)))
for nst in (1, 2, 3, 4, 6):
    ss = [S(10 + i) for i in range(nst)]; strm = [torch.cuda.Stream(dev) for _ in range(nst)]
    for mode in ("fused", "unfused"):
        def step(i):
            with torch.cuda.stream(strm[i % nst]):
                (ss[i % nst].fused() if mode == "fused" else ss[i % nst].unfused()); ss[i % nst].expand()
        for i in range(3 * nst): step(i)
        torch.cuda.synchronize()
        import time
        t0 = time.perf_counter(); n = 300
        for i in range(n): step(i)
        torch.cuda.synchronize()
        print("  %d batches in flight, %-7s: %.1f us per step (eager launches)" % (nst, mode, (time.perf_counter() - t0) / n * 1e6))
