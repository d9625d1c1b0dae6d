#!/usr/bin/env python3
"""Development probe for a GPU box: correctness summary + per-kernel timings.
Not part of the product or the test suite; prints everything it learns."""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import aligner_amd  # noqa: E402
from aligner_amd import _lib, synth  # noqa: E402
from oracle import maxpath_oracle as O  # noqa: E402


def ev_time(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3   # us


def main():
    dev = torch.device("cuda:0")
    p = torch.cuda.get_device_properties(0)
    print("device", p.name, "CUs", p.multi_processor_count, "smem/block", p.shared_memory_per_block,
          getattr(p, "shared_memory_per_block_optin", None), flush=True)
    lib = _lib.load()

    # --- correctness on golden KATs, both kernels ---
    z = np.load(os.path.join(ROOT, "tests", "golden", "kat_small.npz"))
    n = int(z["n"])
    for generic in (False, True):
        bad = []
        for i in range(n):
            v, tx, ty = z[f"c{i}_value"], z[f"c{i}_tx"], z[f"c{i}_ty"]
            r = aligner_amd.align(torch.from_numpy(v).to(dev), torch.from_numpy(tx).to(dev), torch.from_numpy(ty).to(dev),
                                  path_dtype=torch.int32, max_neg_val=float(z[f"c{i}_neg"]), force_generic=(generic is True))
            torch.cuda.synchronize()
            if not np.array_equal(r.path.cpu().numpy(), z[f"c{i}_path"].astype(np.int32)):
                bad.append(str(z["tags"][i]))
        print("KATs", {False: "wide", True: "generic", "halo": "halo"}[generic], f"{n - len(bad)}/{n} ok", "BAD:", bad[:12], flush=True)
    print("status", aligner_amd.read_status(dev))

    # --- C2 ---
    rec = json.load(open(os.path.join(ROOT, "tests", "golden", "appendix_a.json")))
    v = synth.synth_value(*synth.CONFIGS["C2"])
    B, Tx, Ty = v.shape
    vd = torch.from_numpy(v).to(dev)
    txd = torch.full((B,), Tx, dtype=torch.int32, device=dev)
    tyd = torch.full((B,), Ty, dtype=torch.int32, device=dev)
    for generic in (False, True):
        r = aligner_amd.align(vd, txd, tyd, path_dtype=torch.int32, force_generic=generic)
        torch.cuda.synchronize()
        ok = synth.sha256_of(r.path.cpu().numpy()) == rec["C2-fixed"]["path_sha256"]
        print("C2-fixed", "generic" if generic else "pipelined", "hash ok" if ok else "HASH MISMATCH", flush=True)
        if not ok:
            want = np.zeros(v.shape, np.int32)
            vv = v.copy()
            O.maximum_path_c(want, vv, np.full(B, Tx, np.int32), np.full(B, Ty, np.int32))
            got = r.path.cpu().numpy()
            diff = np.argwhere((got != want).any(axis=(1,)))
            print("  first differing (b, y):", diff[:10].tolist())
            print("  dur got", r.durations[0, :16].tolist(), "want", want[0].sum(1)[:16].tolist())
    tx2, ty2 = synth.synth_lengths(64, 200, 500, 1000, 2)
    r = aligner_amd.align(vd, torch.from_numpy(tx2).to(dev), torch.from_numpy(ty2).to(dev), path_dtype=torch.int32)
    torch.cuda.synchronize()
    print("C2-varlen", "hash ok" if synth.sha256_of(r.path.cpu().numpy()) == rec["C2-varlen"]["path_sha256"] else "HASH MISMATCH")

    # --- timings (us) ---
    ws = torch.zeros(lib.aligner_maxpath_workspace_bytes(B, Tx, Ty) + 1024, dtype=torch.uint8, device=dev)
    tok = torch.empty((B, Ty), dtype=torch.int32, device=dev)
    dur = torch.empty((B, Tx), dtype=torch.int32, device=dev)
    path = torch.empty((B, Tx, Ty), dtype=torch.float32, device=dev)
    st = torch.cuda.current_stream().cuda_stream

    def fwd(flags=0):
        _lib.check(lib.aligner_maxpath_forward_f32(vd.data_ptr(), None, 0, txd.data_ptr(), tyd.data_ptr(), tok.data_ptr(),
                                                   dur.data_ptr(), ws.data_ptr(), ws.numel(), B, Tx, Ty, -1e9, flags, st))

    def expand():
        _lib.check(lib.aligner_maxpath_expand(ws.data_ptr(), path.data_ptr(), 0, B, Tx, Ty, st))

    print("forward pipelined us", round(ev_time(lambda: fwd(0)), 2))
    print("forward generic   us", round(ev_time(lambda: fwd(4), iters=5), 2))
    print("expand f32        us", round(ev_time(expand), 2))
    print("align (fwd+expand, python) us", round(ev_time(lambda: aligner_amd.align(vd, txd, tyd, want_durations=True)), 2))
    mask = torch.ones_like(vd)
    print("maximum_path strict us", round(ev_time(lambda: aligner_amd.maximum_path(vd, mask)), 2))
    print("maximum_path prefix us", round(ev_time(lambda: aligner_amd.maximum_path(vd, mask, mask_is_prefix=True)), 2))
    g = torch.Generator().manual_seed(0)
    k = torch.randn(B, 80, Tx, generator=g).to(dev)
    q = torch.randn(B, 80, Ty, generator=g).to(dev)
    out = torch.empty((B, Tx, Ty), dtype=torch.float32, device=dev)
    print("softattn C=80 us", round(ev_time(lambda: aligner_amd.soft_attention(k, q, out=out)), 2))
    # C5 and C4 shard timing
    for tag, (vv, a, b_) in {"C5": (synth.synth_value(*synth.CONFIGS["C5"]), np.full(8, 500, np.int32), np.full(8, 4000, np.int32)),
                             "C4s0": synth.c4_shard(0)}.items():
        vdd = torch.from_numpy(vv).to(dev)
        ad, bd = torch.from_numpy(a).to(dev), torch.from_numpy(b_).to(dev)
        r = aligner_amd.align(vdd, ad, bd, path_dtype=torch.int32)
        torch.cuda.synchronize()
        key = "C5-longform" if tag == "C5" else "C4-shard0"
        print(tag, "hash ok" if synth.sha256_of(r.path.cpu().numpy()) == rec[key]["path_sha256"] else "HASH MISMATCH",
              "us", round(ev_time(lambda: aligner_amd.align(vdd, ad, bd, want_path=False), iters=5), 2), flush=True)
    # CPU baseline quick
    vv = v.copy(); pp = np.zeros(v.shape, np.int32)
    t0 = time.time(); O.maximum_path_c(pp, vv, np.full(B, Tx, np.int32), np.full(B, Ty, np.int32)); t1 = time.time()
    print("cpu oracle 1 thread ms", round((t1 - t0) * 1e3, 2), "nproc", os.cpu_count())


if __name__ == "__main__":
    main()
