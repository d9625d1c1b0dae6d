import torch, sys
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from aligner_amd import _lib
lib = _lib.load(); dev = torch.device("cuda:0")
def ev(fn, it=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / it * 1e3
for (B, Ci, Co, T, K) in [(64, 512, 1024, 200, 3), (64, 1024, 80, 200, 1), (64, 80, 160, 900, 3), (64, 160, 80, 900, 1), (64, 80, 80, 900, 1)]:
    x = torch.randn(B, Ci, T, device=dev); w = torch.randn(Co, Ci, K, device=dev) / (Ci*K)**0.5; bias = torch.randn(Co, device=dev)
    y = torch.empty(B, Co, T, device=dev)
    n = lib.aligner_conv1d_prepared_bytes(Co, Ci, K); prep = torch.empty(n, dtype=torch.uint8, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    tp = ev(lambda: lib.aligner_conv1d_prepare_f32(w.data_ptr(), prep.data_ptr(), n, Co, Ci, K, st))
    tc = ev(lambda: lib.aligner_conv1d_prepared_f32(x.data_ptr(), prep.data_ptr(), bias.data_ptr(), y.data_ptr(), B, Ci, Co, T, K, 1, st))
    tr = ev(lambda: lib.aligner_conv1d_f32(x.data_ptr(), w.data_ptr(), bias.data_ptr(), y.data_ptr(), B, Ci, Co, T, K, 1, st))
    fl = 2.0 * B * T * Co * Ci * K
    print(f"[{B},{Ci}->{Co},T={T},k={K}] prep {tp:.1f} us, prepared conv {tc:.1f} us ({fl/tc/1e6:.1f} TFLOP/s fp32-equivalent), raw-weight kernel {tr:.1f} us")
