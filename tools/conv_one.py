import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from aligner_amd import _lib
lib = _lib.load(); dev = torch.device("cuda:0")
B, Ci, Co, T, K = 64, 512, 1024, 200, 3
x = torch.randn(B, Ci, T, device=dev); w = torch.randn(Co, Ci, K, device=dev) / (Ci*K)**0.5; bias = torch.randn(Co, device=dev)
y = torch.empty(B, Co, T, device=dev)
n = lib.aligner_conv1d_prepared_bytes(Co, Ci, K); prep = torch.empty(n, dtype=torch.uint8, device=dev)
st = torch.cuda.current_stream().cuda_stream
lib.aligner_conv1d_prepare_f32(w.data_ptr(), prep.data_ptr(), n, Co, Ci, K, st)
for _ in range(5):
    lib.aligner_conv1d_prepared_f32(x.data_ptr(), prep.data_ptr(), bias.data_ptr(), y.data_ptr(), B, Ci, Co, T, K, 1, st)
torch.cuda.synchronize()
