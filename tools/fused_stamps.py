import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aligner_amd import _lib
lib = _lib.load(); dev = torch.device("cuda:0")
B, C, TX, TY = 64, 80, 200, 1000
g = torch.Generator().manual_seed(1)
k = torch.randn(B, C, TX, generator=g).to(dev); q = torch.randn(B, C, TY, generator=g).to(dev)
tx = torch.full((B,), TX, dtype=torch.int32, device=dev); ty = torch.full((B,), TY, dtype=torch.int32, device=dev)
logp = torch.empty((B, TX, TY), device=dev); tok = torch.empty((B, TY), dtype=torch.int32, device=dev); dur = torch.empty((B, TX), dtype=torch.int32, device=dev)
ws = torch.zeros(lib.aligner_maxpath_workspace_bytes(B, TX, TY) + 256, dtype=torch.uint8, device=dev)
def run():
    _lib.check(lib.aligner_fused_align_f32(k.data_ptr(), q.data_ptr(), tx.data_ptr(), ty.data_ptr(), (None if os.environ.get("NOLOGP") else logp.data_ptr()), tok.data_ptr(), dur.data_ptr(), ws.data_ptr(), ws.numel(), B, C, TX, TY, 0.0005, 0, -1e9, torch.cuda.current_stream().cuda_stream))
for _ in range(3): run()
torch.cuda.synchronize()
st = torch.zeros((B, 16, 16), dtype=torch.int64, device=dev)
lib.aligner_debug_set_stamps(st.data_ptr()); run(); torch.cuda.synchronize(); lib.aligner_debug_set_stamps(None)
s = st.cpu().numpy()
for w in range(8):
    r = s[0, w]
    print("wave", w, "phase 12: start +", [int(r[kk] - r[8]) if r[kk] else None for kk in (9, 10, 11, 12)], " (producer: after F, after C, after mel write, after barrier; DP: -, -, before barrier, after)")
