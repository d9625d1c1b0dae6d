import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import aligner_amd
dev = torch.device("cuda:0")
def t(fn, it=10):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / it * 1e3
for (B, Ci, Co, T, K) in [(64, 512, 1024, 200, 3), (64, 1024, 80, 200, 1), (64, 80, 160, 900, 3), (64, 160, 80, 900, 1), (64, 80, 80, 900, 1)]:
    x = torch.randn(B, Ci, T, device=dev); w = torch.randn(Co, Ci, K, device=dev); b = torch.randn(Co, device=dev)
    us = t(lambda: aligner_amd.conv1d(x, w, b, True))
    fl = 2.0 * B * T * Ci * Co * K
    us_t = t(lambda: torch.nn.functional.conv1d(x, w, b, padding=K // 2))
    print(f"conv B{B} {Ci}->{Co} T{T} k{K}: {us:9.1f} us  {fl/us/1e6:7.2f} TFLOP/s   (torch/MIOpen {us_t:9.1f} us {fl/us_t/1e6:7.2f} TF)")
