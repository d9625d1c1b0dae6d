"""The two encoder stacks of BASELINE configs[2] ([64, 512 x 200 text, 80 x 900 mel]), timed: the whole stack in one
C-ABI call (aligner_conv_stack_f32) and the layers one by one; under rocprofv3 --kernel-trace --stats this is the launch
set to read for the per-kernel times."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import aligner_amd
from aligner_amd.softattn import encode
dev = torch.device("cuda:0")
from aligner_amd import _lib
if os.environ.get("FT"): _lib.load().aligner_debug_set_option(b"conv_narrow_ft", int(os.environ["FT"])); print("narrow frame tiles forced to", os.environ["FT"])
params = aligner_amd.AlignmentEncoderParams.random(512, 80, 80, dev, seed=3)
text = torch.randn(64, 512, 200, device=dev); mel = torch.randn(64, 80, 900, device=dev)
def time_us(fn, it=20):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / it * 1e3
def layers(x, stack):
    for n, (w, b) in enumerate(stack):
        x = aligner_amd.conv1d(x, w, b, relu=(n + 1 < len(stack)))
    return x
for r in range(3):
    print(f"round {r}: text stack {time_us(lambda: encode(text, params.key_proj)):7.1f} us (layer by layer {time_us(lambda: layers(text, params.key_proj)):7.1f})"
          f"   mel stack {time_us(lambda: encode(mel, params.query_proj)):7.1f} us (layer by layer {time_us(lambda: layers(mel, params.query_proj)):7.1f})", flush=True)
a, b = encode(text, params.key_proj), layers(text, params.key_proj)
c, d = encode(mel, params.query_proj), layers(mel, params.query_proj)
print("max |stack - layers|: text", float((a - b).abs().max()), " mel", float((c - d).abs().max()))
