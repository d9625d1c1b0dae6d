"""In-kernel timeline of conv_narrow_fused_kernel on BASELINE configs[2]'s mel encoder ([64, 80, 900]): shader-clock stamps per
wave (debug hook aligner_debug_set_stamps), cycles since the workgroup's entry; median / max over the workgroups."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import aligner_amd
from aligner_amd import _lib
from aligner_amd.softattn import encode
lib = _lib.load(); dev = torch.device("cuda:0")
params = aligner_amd.AlignmentEncoderParams.random(512, 80, 80, dev, seed=3)
mel = torch.randn(64, 80, 900, device=dev)
for _ in range(3): encode(mel, params.query_proj)
torch.cuda.synchronize()
st = torch.zeros((4096, 16, 8), dtype=torch.int64, device=dev)
lib.aligner_debug_set_stamps(st.data_ptr())
encode(mel, params.query_proj); torch.cuda.synchronize()
lib.aligner_debug_set_stamps(None)
s = st.cpu().numpy().astype(np.float64)
s = s[s[:, 0, 0] > 0]
names = ["entry", "staged + barrier", "layer 0 steps", "epilogue 0 -> LDS + barrier", "layer 1 steps", "epilogue 1 -> LDS + barrier", "layer 2 steps", "stores issued"]
print("workgroups stamped:", s.shape[0])
t0 = np.where(s[:, :, 0] > 0, s[:, :, 0], np.inf).min(axis=1)
for w in range(16):
    if not (s[:, w, 0] > 0).any(): continue
    row = []
    for k in range(1, 8):
        v = s[:, w, k]; ok = v > 0
        if ok.any():
            d = (v - t0)[ok]; row.append(f"{names[k]} {np.median(d):7.0f}/{d.max():7.0f}")
    print(f"wave {w}: " + " | ".join(row))
