"""One wide convolution layer, timed: python tools/conv_one.py [B Cin Cout T K] (default: the C3 text encoder's 512 -> 1024 k=3).
Under rocprofv3 (--kernel-trace --stats, or --pmc) this is the launch set to read: the split pass + the GEMM kernel
(csrc/convgemm.hip) and, for comparison, round 3's conv1d_prepared_kernel on the same prepared buffer."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from aligner_amd import _lib
lib = _lib.load(); dev = torch.device("cuda:0")
B, Ci, Co, T, K = (int(a) for a in sys.argv[1:6]) if len(sys.argv) >= 6 else (64, 512, 1024, 200, 3)
x = torch.randn(B, Ci, T, device=dev); w = torch.randn(Co, Ci, K, device=dev) / (Ci*K)**0.5; bias = torch.randn(Co, device=dev)
y = torch.empty(B, Co, T, device=dev); y2 = torch.empty(B, Co, T, device=dev)
n = lib.aligner_conv1d_prepared_bytes(Co, Ci, K); prep = torch.empty(n, dtype=torch.uint8, device=dev)
nws = lib.aligner_conv1d_workspace_bytes(B, Ci, Co, T, K); ws = torch.empty(max(nws, 16), dtype=torch.uint8, device=dev)
st = torch.cuda.current_stream().cuda_stream
_lib.check(lib.aligner_conv1d_prepare_f32(w.data_ptr(), prep.data_ptr(), n, Co, Ci, K, st))
new = lambda: _lib.check(lib.aligner_conv1d_prepared_ws_f32(x.data_ptr(), prep.data_ptr(), bias.data_ptr(), y.data_ptr(), ws.data_ptr(), nws, B, Ci, Co, T, K, 1, st))
old = lambda: _lib.check(lib.aligner_conv1d_prepared_f32(x.data_ptr(), prep.data_ptr(), bias.data_ptr(), y2.data_ptr(), B, Ci, Co, T, K, 1, st))
def time_us(fn, it=20):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / it * 1e3
for r in range(3):
    print(f"round {r}: GEMM form (split + kernel) {time_us(new):8.1f} us   round-3 kernel {time_us(old):8.1f} us   workspace {nws} B", flush=True)
print("max |new - old| =", float((y - y2).abs().max()), " flops", 2.0 * B * Co * T * Ci * K)
