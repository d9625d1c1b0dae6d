"""In-kernel timeline of the two-workgroup alignment search at C5 ([8,500,4000] bf16): both halves' stamps on the 100 MHz clock
all XCDs share (slots 6 / 7) and in shader cycles since each workgroup's entry; NOSPLIT=1: one walker for all rows."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aligner_amd import _lib, synth
lib = _lib.load(); dev = torch.device("cuda:0")
B, Tx, Ty = 8, 500, 4000
v = torch.from_numpy(synth.synth_value(B, Tx, Ty, 5, bits=8, denom=8.0)).to(dev).to(torch.bfloat16)
tx = torch.full((B,), Tx, dtype=torch.int32, device=dev); ty = torch.full((B,), Ty, dtype=torch.int32, device=dev)
dur = torch.empty((B, Tx), dtype=torch.int32, device=dev)
ws = torch.zeros(lib.aligner_maxpath_workspace_bytes(B, Tx, Ty) + 256, dtype=torch.uint8, device=dev)
st = torch.zeros((2 * B + 256, 16, 16), dtype=torch.int64, device=dev)
lib.aligner_debug_set_option(b"maxpath_no_split_walk", int(os.environ.get("NOSPLIT", "0")))
def run():
    _lib.check(lib.aligner_maxpath_forward(v.data_ptr(), _lib.DT_BF16, None, 0, tx.data_ptr(), ty.data_ptr(), None, dur.data_ptr(), ws.data_ptr(), ws.numel(), B, Tx, Ty, -1e9, 0, torch.cuda.current_stream().cuda_stream))
for _ in range(5): run()
torch.cuda.synchronize()
lib.aligner_debug_set_stamps(st.data_ptr()); run(); torch.cuda.synchronize(); lib.aligner_debug_set_stamps(None)
s = st.cpu().numpy().astype(np.float64)
names = {0: "entry", 4: "own sweep stream done", 1: "words in LDS / walk starts", 3: "walk done", 5: "outputs stored"}
for half, blk in (("first half", s[:B]), ("second half", s[B:2 * B])):
    print(half)
    for k in (4, 1, 3, 5):
        d = blk[:, 0, k] - blk[:, 0, 0]
        ok = blk[:, 0, k] > 0
        if ok.any(): print(f"   {names[k]:>28}: wave 0, cycles since entry: median {np.median(d[ok]):9.0f}  max {d[ok].max():9.0f}")
    rt = (blk[:, 0, 7] - blk[:, 0, 6]) * 10.0 / 1e3
    print(f"   entry -> exit on the common clock: median {np.median(rt):.1f} us, max {rt.max():.1f} us")
t0 = min(s[:2 * B, 0, 6].min(), 1e30)
print("launch: first entry -> last exit %.1f us" % ((s[:2 * B, 0, 7].max() - t0) * 10.0 / 1e3))
