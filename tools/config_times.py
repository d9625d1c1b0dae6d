#!/usr/bin/env python3
"""Per-config timings of the hot path on one MI355X (BASELINE.json configs C2..C5), HIP events on
the launch stream, every call through the C ABI wrappers.  Development/reporting aid for DESIGN.md."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import aligner_amd
from aligner_amd import synth
dev = torch.device("cuda:0")


def ev(fn, it=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / it * 1e3


def lengths(B, Tx, Tymin, Tymax, seed):
    import numpy as np
    tx, ty = synth.synth_lengths(B, Tx, Tymin, Tymax, seed)
    return torch.from_numpy(tx).to(dev), torch.from_numpy(ty).to(dev)


g = torch.Generator().manual_seed(0)
# C3: full OTA pipeline [64, 512-dim text emb x 200, 80-mel x 900]
B, Ct, Cm, Tx, Ty = 64, 512, 80, 200, 900
params = aligner_amd.AlignmentEncoderParams.random(Ct, Cm, 80, dev, seed=3)
text = torch.randn(B, Ct, Tx, generator=g).to(dev); mel = torch.randn(B, Cm, Ty, generator=g).to(dev)
tx = torch.full((B,), Tx, dtype=torch.int32, device=dev); ty = torch.full((B,), Ty, dtype=torch.int32, device=dev)
k = aligner_amd.softattn.encode(text, params.key_proj); q = aligner_amd.softattn.encode(mel, params.query_proj)
logp, _ = aligner_amd.soft_attention(k, q, t_x=tx)
print("C3 [64, 512x200 text, 80x900 mel]: text encoder %.1f us, mel encoder %.1f us, similarity+log-softmax %.1f us, DP (dense path + durations) %.1f us, whole pipeline %.1f us" % (
    ev(lambda: aligner_amd.softattn.encode(text, params.key_proj)), ev(lambda: aligner_amd.softattn.encode(mel, params.query_proj)),
    ev(lambda: aligner_amd.soft_attention(k, q, t_x=tx)), ev(lambda: aligner_amd.align(logp, tx, ty)),
    ev(lambda: aligner_amd.align(aligner_amd.alignment_encoder(text, mel, params, t_x=tx)[0], tx, ty))))
# C4: one 64-utterance shard of the 512 variable-length utterances, [64, 400, 2000]
B, Tx, Ty = 64, 400, 2000
v = torch.from_numpy(synth.synth_value(B, Tx, Ty, 40)).to(dev)
import numpy as np
txa, tya = synth.synth_lengths(512, 400, 200, 2000, 4)
tx, ty = torch.from_numpy(txa[:64].copy()).to(dev), torch.from_numpy(tya[:64].copy()).to(dev)
print("C4 shard [64,400,2000] ragged (sum t_y = %d frames): DP durations only %.1f us, with dense fp32 path %.1f us" % (
    int(tya[:64].sum()), ev(lambda: aligner_amd.align(v, tx, ty, want_path=False)), ev(lambda: aligner_amd.align(v, tx, ty))))
# C5: long-form [8, 500, 4000], bf16-exact scores
B, Tx, Ty = 8, 500, 4000
v = torch.from_numpy(synth.synth_value(B, Tx, Ty, 5, bits=8, denom=8.0)).to(dev).to(torch.bfloat16)
tx = torch.full((B,), Tx, dtype=torch.int32, device=dev); ty = torch.full((B,), Ty, dtype=torch.int32, device=dev)
print("C5 long-form [8,500,4000] bf16 scores: DP durations only %.1f us, with dense int32 path %.1f us" % (
    ev(lambda: aligner_amd.align(v, tx, ty, want_path=False)), ev(lambda: aligner_amd.align(v, tx, ty, path_dtype=torch.int32))))
v32 = v.float()                        # (cast once, outside the timed calls)
print("   ... kept on one CU per utterance (cus_per_utterance=1): %.1f us; fp32 scores, two CUs / one CU: %.1f / %.1f us" % (
    ev(lambda: aligner_amd.align(v, tx, ty, want_path=False, cus_per_utterance=1)),
    ev(lambda: aligner_amd.align(v32, tx, ty, want_path=False)),
    ev(lambda: aligner_amd.align(v32, tx, ty, want_path=False, cus_per_utterance=1))))
# C5 as BASELINE names it: bf16 similarity -> (a) maximum_path and (b) the MoBoAligner boundary search, int32 out
kk = torch.randn(B, 80, Tx, generator=g).to(dev); qq = torch.randn(B, 80, Ty, generator=g).to(dev)
lp16, _ = aligner_amd.soft_attention(kk, qq, t_x=tx, logp_dtype=torch.bfloat16)
print("C5 [8,80,500,4000]: similarity with bf16 log-probs %.1f us (fp32 log-probs %.1f us); DP on them, durations only %.1f us" % (
    ev(lambda: aligner_amd.soft_attention(kk, qq, t_x=tx, logp_dtype=torch.bfloat16)), ev(lambda: aligner_amd.soft_attention(kk, qq, t_x=tx)),
    ev(lambda: aligner_amd.align(lp16, tx, ty, want_path=False))))
for D in (16, 32, 64):
    print("C5 MoBoAligner boundary search [8,500,4000] bf16 energies, max duration %d: boundaries only %.1f us, with log_alpha %.1f us, with gamma %.1f us" % (
        D, ev(lambda: aligner_amd.boundary_search(lp16, tx, ty, D), it=5, warm=1),
        ev(lambda: aligner_amd.boundary_search(lp16, tx, ty, D, want_log_alpha=True), it=5, warm=1),
        ev(lambda: aligner_amd.boundary_search(lp16, tx, ty, D, want_gamma=True), it=5, warm=1)))
# widened rows at the C2 and C5 shapes
for (B, Tx, Ty, Cc) in ((64, 200, 1000, 512), (8, 500, 4000, 512)):
    lp = torch.log_softmax(torch.randn(B, Tx, Ty, generator=g), dim=1).to(dev)
    tx = torch.full((B,), Tx, dtype=torch.int32, device=dev); ty = torch.full((B,), Ty, dtype=torch.int32, device=dev)
    dur = aligner_amd.align(lp, tx, ty, want_path=False).durations
    h = torch.randn(B, Cc, Tx, generator=g).to(dev)
    print("[%d,%d,%d]: forward-sum loss %.1f us, with gradient %.1f us, beta-binomial prior %.1f us, length regulator (C=%d) %.1f us" % (
        B, Tx, Ty, ev(lambda: aligner_amd.forward_sum(lp, tx, ty, want_grad=False)), ev(lambda: aligner_amd.forward_sum(lp, tx, ty)),
        ev(lambda: aligner_amd.beta_binomial_prior(tx, ty, Tx, Ty)), Cc, ev(lambda: aligner_amd.regulate(h, dur, Ty))))

# the drop-in call itself: maximum_path(value, mask) on GPU tensors, as a training step calls it (the lengths
# come from the mask, the 0/1 path comes back in value's dtype) -- GPU time per call and host wall time per synchronous call
import time
B, Tx, Ty = 64, 200, 1000
v = torch.from_numpy(synth.synth_value(B, Tx, Ty, 2)).to(dev)
txr, tyr = lengths(B, Tx, 500, 1000, 2)
mask = ((torch.arange(Tx, device=dev)[None, :, None] < txr[:, None, None]) & (torch.arange(Ty, device=dev)[None, None, :] < tyr[:, None, None])).to(v.dtype)
full = torch.ones_like(v)
for name, m in (("full-length mask", full), ("ragged mask", mask)):
    t_gpu = ev(lambda: aligner_amd.maximum_path(v, m))
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20):
        aligner_amd.maximum_path(v, m); torch.cuda.synchronize()
    t_wall = (time.perf_counter() - t0) / 20 * 1e6
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(200): aligner_amd.maximum_path(v, m)
    t_host = (time.perf_counter() - t0) / 200 * 1e6; torch.cuda.synchronize()
    print("drop-in maximum_path(value, mask) [64,200,1000], %s: %.1f us of GPU time per call, %.1f us wall per synchronous call, %.1f us of host time per asynchronous call" % (name, t_gpu, t_wall, t_host))
