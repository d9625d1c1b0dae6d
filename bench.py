#!/usr/bin/env python3
"""bench.py -- headline benchmark of the alignment hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload (BASELINE.json configs[1]): per GPU one batch [B=64, T_text=200,
T_mel=1000] fp32 -- soft-attention log-likelihood matrix from encoded text/mel
(the "similarity GEMM") followed by the monotonic alignment search (forward sweep +
backtrack + dense 0/1 path + durations).  A step is one pass over one batch with
all inputs already resident in HBM.  With N > 1 every rank aligns its own batch
(weak scaling; utterances are independent, reference core.pyx:44-45) and the int32
duration vectors are all-gathered over RCCL in buckets on a side stream.

Prints ONE JSON line on rank 0 (see the contract in the task description) with two
extra objects: `roofline` (dominant kernel, timed with HIP events on the launch
stream) and `cpu_baseline` (the reference / its C restatement on the host cores).
"""
from __future__ import annotations

import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from aligner_amd import _lib, synth  # noqa: E402

B, C_ATT, TX, TY = 64, 80, 200, 1000
CONTIGUOUS_LOGP = False          # --contiguous-logp: the step's intermediate in the reference's contiguous layout (A-B)
SUSTAINED_STEPS = 20000          # the long region reported beside a short timed one (`sustained`), and the default K
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)
SIM_KERNEL = "softattn_rt_kernel"     # the similarity kernel configs[1] launches (csrc/softattn.hip: the row-tile form)


class Step:
    """One pass of the hot path through the C ABI with preallocated buffers."""

    def __init__(self, dev: torch.device, seed: int, use_graph: bool, stream_path: bool = False):
        self.dev = dev
        # "branch": the dense path in the reference's own two steps (zeros, then ones: __init__.py:15, core.pyx:33) with the
        # zeros on a second graph branch BESIDE the alignment search, so that only the ones follow it; "expand": one
        # streaming kernel behind the search.  The one-batch-at-a-time figure uses the branch form.
        self.path_mode = "expand"
        self.side = torch.cuda.Stream(dev)
        # ALIGNER_F_STREAM_PATH: non-temporal stores for the dense path (pays with several batches in flight)
        self.expand_flags = _lib.F_STREAM_PATH if stream_path else 0
        self.lib = _lib.load()
        g = torch.Generator().manual_seed(seed)
        self.keys = torch.randn(B, C_ATT, TX, generator=g).to(dev)       # encoded text  [B,C,Tx]
        self.queries = torch.randn(B, C_ATT, TY, generator=g).to(dev)    # encoded mel   [B,C,Ty]
        self.t_x = torch.full((B,), TX, dtype=torch.int32, device=dev)
        self.t_y = torch.full((B,), TY, dtype=torch.int32, device=dev)
        # the step's own intermediate: log-probs with rows on whole 128-byte lines (a pitch of 1024 elements for T_mel = 1000:
        # aligner_softattn_ld -> aligner_maxpath_ld); --contiguous-logp: the reference's [B,Tx,Ty] layout
        if CONTIGUOUS_LOGP:
            self.logp = torch.empty((B, TX, TY), dtype=torch.float32, device=dev)
        else:
            from aligner_amd.softattn import pitched_logp
            self.logp = pitched_logp(B, TX, TY, dev)
        self.ld = int(self.logp.stride(1))
        self.path = torch.empty((B, TX, TY), dtype=torch.float32, device=dev)
        self.tok = torch.empty((B, TY), dtype=torch.int32, device=dev)
        self.dur = torch.empty((B, TX), dtype=torch.int32, device=dev)
        nws = self.lib.aligner_maxpath_workspace_bytes(B, TX, TY)
        self.ws = torch.zeros(nws + 256, dtype=torch.uint8, device=dev)
        self.sa_ws = torch.empty(self.lib.aligner_softattn_workspace_bytes(B, C_ATT, TX) + 256, dtype=torch.uint8,
                                 device=dev)
        self.dur_out = self.dur          # where the DP kernel writes the durations (N>1: a slot of a gather bucket)
        self.graph = None
        self.slot_graphs = {}            # N>1: one captured step per bucket slot this stream serves
        self.use_graph = use_graph

    def stream(self) -> int:
        return torch.cuda.current_stream(self.dev).cuda_stream

    def softattn(self):
        _lib.check(self.lib.aligner_softattn_ld(self.keys.data_ptr(), self.queries.data_ptr(), self.t_x.data_ptr(),
                                                None, self.logp.data_ptr(), _lib.DT_F32, self.ld, None, self.sa_ws.data_ptr(),
                                                self.sa_ws.numel(), B, C_ATT, TX, TY, 0.0005,
                                                _lib.SIM_L2, self.stream()))

    def forward(self):
        _lib.check(self.lib.aligner_maxpath_ld(self.logp.data_ptr(), _lib.DT_F32, self.ld, self.t_x.data_ptr(),
                                               self.t_y.data_ptr(), None, 0, self.tok.data_ptr(), self.dur_out.data_ptr(),
                                               self.ws.data_ptr(), self.ws.numel(), B, TX, TY, -1e9, 0, self.stream()))

    def expand(self):
        _lib.check(self.lib.aligner_maxpath_expand_ex(self.ws.data_ptr(), self.path.data_ptr(), _lib.DT_F32, B, TX, TY,
                                                      self.expand_flags, self.stream()))

    def zero_path(self):
        _lib.check(self.lib.aligner_maxpath_zero_path(self.path.data_ptr(), _lib.DT_F32, B, TX, TY, self.expand_flags,
                                                      self.stream()))

    def scatter_path(self):
        _lib.check(self.lib.aligner_maxpath_scatter_path(self.ws.data_ptr(), self.path.data_ptr(), _lib.DT_F32, B, TX, TY,
                                                         self.stream()))

    def forward_with_ones(self):
        """The search with ALIGNER_F_PATH_PREZEROED: the kernel writes the path's ones into zeros that are already there."""
        _lib.check(self.lib.aligner_maxpath_ld(self.logp.data_ptr(), _lib.DT_F32, self.ld, self.t_x.data_ptr(), self.t_y.data_ptr(),
                                               self.path.data_ptr(), _lib.DT_F32, self.tok.data_ptr(),
                                               self.dur_out.data_ptr(), self.ws.data_ptr(), self.ws.numel(), B, TX, TY,
                                               -1e9, _lib.F_PATH_PREZEROED, self.stream()))

    def search_with_path(self):
        """aligner_maxpath_f32 as a caller uses it: ONE launch -- the search, and on the CUs the batch leaves idle the
        zeros of the dense path; the utterances' workgroups write the ones."""
        _lib.check(self.lib.aligner_maxpath_ld(self.logp.data_ptr(), _lib.DT_F32, self.ld, self.t_x.data_ptr(), self.t_y.data_ptr(),
                                               self.path.data_ptr(), _lib.DT_F32, self.tok.data_ptr(),
                                               self.dur_out.data_ptr(), self.ws.data_ptr(), self.ws.numel(), B, TX, TY,
                                               -1e9, self.expand_flags, self.stream()))

    def eager(self):
        if self.path_mode == "fused":
            self.softattn()
            self.search_with_path()
            return
        if self.path_mode == "inline":
            # zeros (np.zeros of __init__.py:15) on a second branch BESIDE the similarity kernel -- whose first stores
            # leave only after its 7 us prologue --, then the search writes its own ones (core.pyx:33): nothing of the
            # dense path is left on the step's serial chain
            cur = torch.cuda.current_stream(self.dev)
            self.side.wait_stream(cur)
            with torch.cuda.stream(self.side):
                self.zero_path()
            self.softattn()
            cur.wait_stream(self.side)
            self.forward_with_ones()
            return
        self.softattn()
        if self.path_mode == "branch":
            cur = torch.cuda.current_stream(self.dev)
            self.side.wait_stream(cur)                 # fork: the zeros need nothing of this step ...
            with torch.cuda.stream(self.side):
                self.zero_path()
            self.forward()
            cur.wait_stream(self.side)                 # ... join: the ones need the zeros and the search
            self.scatter_path()
        else:
            self.forward()
            self.expand()

    def capture(self):
        """Capture the three launches of a step into a HIP graph (launch-bound otherwise)."""
        if not self.use_graph:
            return
        try:
            s = torch.cuda.Stream(self.dev)
            s.wait_stream(torch.cuda.current_stream(self.dev))
            with torch.cuda.stream(s):
                self.eager()
            torch.cuda.current_stream(self.dev).wait_stream(s)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                self.eager()
            self.graph = g
        except Exception as e:  # noqa: BLE001  (fall back to eager launches, say so)
            print(f"[bench] graph capture unavailable ({type(e).__name__}: {e}); using eager launches",
                  file=sys.stderr)
            self.graph = None

    def capture_slot(self, key, dur_slot: torch.Tensor):
        """N>1: the same step with the durations written straight into `dur_slot` (a row of a gather
        bucket), so that no per-step copy (and no per-step host work beyond one graph launch) is needed."""
        self.dur_out = dur_slot
        self.capture()
        self.slot_graphs[key] = (self.graph, dur_slot)
        self.dur_out = self.dur
        self.graph = None

    def run_slot(self, key):
        g, dur_slot = self.slot_graphs[key]
        if g is not None:
            g.replay()
        else:
            self.dur_out = dur_slot
            self.eager()
            self.dur_out = self.dur

    def __call__(self):
        if self.graph is not None:
            self.graph.replay()
        else:
            self.eager()


def event_time_us(fn, iters: int, dev) -> float:
    """Average duration of `fn` (one launch) over `iters` back-to-back launches, HIP
    events recorded on the stream the kernel is launched on."""
    for _ in range(max(1, iters // 10)):         # (untimed: a tenth as many launches first)
        fn()
    torch.cuda.synchronize(dev)
    s = torch.cuda.Event(enable_timing=True)
    e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize(dev)
    return s.elapsed_time(e) / iters * 1e3


def cpu_baseline(dev, seconds: float = 12.0):
    """The reference's maximum_path_c on the host, 1 thread (the reference ships serial: setup.py passes no
    -fopenmp), on the same [64,200,1000] workload shape -- through oracle/libmaxpath_oracle.so, the C
    restatement of core.pyx:7-45 that is pinned bit-for-bit to the reference's outputs (same loops, same
    -O2 / no fast-math flags).  No compiled form of the reference travels to the GPU box."""
    from oracle import maxpath_oracle as O
    value = synth.synth_value(*synth.CONFIGS["C2"])
    tx = np.full(B, TX, np.int32)
    ty = np.full(B, TY, np.int32)
    paths = np.zeros(value.shape, np.int32)
    work = value.copy()
    O.maximum_path_c(paths, work, tx, ty)                 # warm-up
    reps, spent = 0, 0.0
    while spent < seconds and reps < 2000:
        np.copyto(work, value)                            # fresh scores, outside the timer
        paths.fill(0)
        t0 = time.perf_counter()
        O.maximum_path_c(paths, work, tx, ty)
        spent += time.perf_counter() - t0
        reps += 1
    ups = B * reps / spent
    out = {"value": round(ups, 1), "unit": "utterances/s", "cores": 1, "kind": "port",
           "sample": f"maximum_path_c core (C restatement, oracle/maxpath_oracle.c) on [64,200,1000] fp32 scores, "
                     f"{reps} batches in {spent:.1f}s ({spent / reps * 1e3:.1f} ms/batch), host has "
                     f"{os.cpu_count()} logical cores",
           "frames_per_s": round(ups * TY, 1)}
    # secondary line 1: what `prange` (core.pyx:44) would give if the reference were built with
    # -fopenmp -- the C restatement's OpenMP batch loop, one thread per utterance at most, ~4 s
    try:
        nthr = max(1, min(B, os.cpu_count() or 1))
        O.maximum_path_c(paths, work, tx, ty, num_threads=nthr)
        r2, s2 = 0, 0.0
        while s2 < 4.0 and r2 < 4000:
            np.copyto(work, value)
            paths.fill(0)
            t0 = time.perf_counter()
            O.maximum_path_c(paths, work, tx, ty, num_threads=nthr)
            s2 += time.perf_counter() - t0
            r2 += 1
        out["all_cores"] = {"value": round(B * r2 / s2, 1), "unit": "utterances/s", "cores": nthr, "kind": "port",
                            "sample": f"OpenMP batch loop of the C restatement, {r2} batches in {s2:.1f}s",
                            "why_this_many": f"one thread per utterance of the batch (the reference's prange is over the batch, "
                                             f"core.pyx:44): min(B = {B}, {os.cpu_count()} logical cores)"}
    except Exception as e:  # noqa: BLE001
        out["all_cores"] = {"error": f"{type(e).__name__}: {e}"}
    # secondary line 2: "wrapper-equivalent" -- what a caller of the reference's maximum_path(value,
    # mask) pays end to end for GPU-resident tensors: the mask multiply, 2 x D2H and 1 x H2D of the [B,Tx,Ty]
    # tensor, the host casts / allocations and the serial core (__init__.py:11-21), restated in
    # oracle.maximum_path on top of the same C core
    try:
        v_dev = torch.from_numpy(value).to(dev)
        m_dev = torch.ones_like(v_dev)
        O.maximum_path(v_dev, m_dev)
        torch.cuda.synchronize(dev)
        r3, s3 = 0, 0.0
        while s3 < 4.0 and r3 < 200:
            t0 = time.perf_counter()
            O.maximum_path(v_dev, m_dev)
            torch.cuda.synchronize(dev)
            s3 += time.perf_counter() - t0
            r3 += 1
        out["wrapper_equivalent"] = {"value": round(B * r3 / s3, 1), "unit": "utterances/s", "cores": 1, "kind": "port",
                                     "sample": f"maximum_path(value, mask) with GPU tensors as the reference wrapper "
                                               f"runs it (2 x D2H + H2D of 51.2 MB, host casts, serial core): "
                                               f"{r3} calls in {s3:.1f}s ({s3 / r3 * 1e3:.1f} ms/call)"}
    except Exception as e:  # noqa: BLE001
        out["wrapper_equivalent"] = {"error": f"{type(e).__name__}: {e}"}
    return out


def run_c4(args, world: int, rank: int, local: int):
    """BASELINE configs[3]: the 512 variable-length utterances ([<=400 tokens, <=2000 frames])
    aligned across the ranks of one node -- strong scaling: the job is fixed, each rank takes the utterances a
    longest-processing-time-first plan on the DP cost t_x*(t_y-t_x+1) hands it (aligner_amd/sharded.py), aligns
    them (durations only) and ONE all-gather of the int32 durations over RCCL restores utterance order on every
    rank.  A step = the whole job once; value = 512 * steps / time."""
    import hashlib
    from aligner_amd import sharded
    N_UTT, TXM, TYM = 512, 400, 2000
    dist = None
    if world > 1 or os.environ.get("ALIGNER_BENCH_FORCE_DIST") == "1":
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if world == 1:                                  # the one-rank form of the multi-rank path, started by hand
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
            os.environ.setdefault("MASTER_PORT", "29531")
        if os.environ.get("ALIGNER_BENCH_REHEARSE") == "1":
            local = 0
            torch.cuda.set_device(0)
            dist.init_process_group("gloo")
        else:
            torch.cuda.set_device(local)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    _lib.require_gpu()
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    lib = _lib.load()
    tx_all, ty_all = synth.synth_lengths(N_UTT, TXM, 200, TYM, 4)
    plan = sharded.lpt_partition(tx_all, ty_all, world)
    mine = plan[rank]
    cap = max(len(q) for q in plan)
    cost = sharded.dp_cost(tx_all, ty_all)
    loads = [int(cost[q].sum()) for q in plan]
    # this rank's utterances, resident in HBM (padded batch [cap, 400, 2000])
    value = torch.zeros((cap, TXM, TYM), dtype=torch.float32, device=dev)
    for n, gidx in enumerate(mine):
        value[n] = torch.from_numpy(synth.c4_utterance(int(gidx))[0]).to(dev)
    t_x = torch.ones(cap, dtype=torch.int32, device=dev)
    t_y = torch.ones(cap, dtype=torch.int32, device=dev)
    t_x[:len(mine)] = torch.from_numpy(tx_all[mine]).to(dev)
    t_y[:len(mine)] = torch.from_numpy(ty_all[mine]).to(dev)
    dur = torch.zeros((cap, TXM), dtype=torch.int32, device=dev)
    ws = torch.zeros(lib.aligner_maxpath_workspace_bytes(cap, TXM, TYM) + 256, dtype=torch.uint8, device=dev)
    gathered = torch.zeros((world * cap, TXM), dtype=torch.int32, device=dev)
    # every rank knows the whole plan: row r*cap + n of `gathered` is utterance plan[r][n]
    src = np.full(N_UTT, 0, np.int64)
    for r, q in enumerate(plan):
        src[q] = r * cap + np.arange(len(q))
    src_t = torch.from_numpy(src).to(dev)
    full = torch.zeros((N_UTT, TXM), dtype=torch.int32, device=dev)

    def step():
        _lib.check(lib.aligner_maxpath_forward_f32(value.data_ptr(), None, 0, t_x.data_ptr(), t_y.data_ptr(), None,
                                                   dur.data_ptr(), ws.data_ptr(), ws.numel(), cap, TXM, TYM, -1e9, 0,
                                                   torch.cuda.current_stream(dev).cuda_stream))
        if dist is not None:
            dist.all_gather_into_tensor(gathered, dur)
            torch.index_select(gathered, 0, src_t, out=full)
        else:
            torch.index_select(dur, 0, src_t, out=full)

    def timed(nsteps):
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(nsteps):
            step()
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)
        el = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([el], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el

    for _ in range(args.warmup):
        step()
    elapsed = timed(args.steps)
    # this rank's DP time alone (HIP events), for the load-balance line
    t_dp = event_time_us(lambda: _lib.check(lib.aligner_maxpath_forward_f32(
        value.data_ptr(), None, 0, t_x.data_ptr(), t_y.data_ptr(), None, dur.data_ptr(), ws.data_ptr(), ws.numel(),
        cap, TXM, TYM, -1e9, 0, torch.cuda.current_stream(dev).cuda_stream)), 10, dev)
    dp_all = [t_dp]
    if dist is not None:
        t = torch.zeros(world, dtype=torch.float64, device=dev)
        t[rank] = t_dp
        dist.all_reduce(t)
        dp_all = [float(x) for x in t.cpu()]
    # parity on the timed output: per contiguous 64-utterance shard, the reference's duration hashes
    ok = None
    try:
        with open(os.path.join(ROOT, "tests", "golden", "appendix_a.json")) as f:
            rec = json.load(f)
        fh = full.cpu().numpy().astype(np.int32)
        ok = all(hashlib.sha256(np.ascontiguousarray(fh[64 * s:64 * s + 64]).tobytes()).hexdigest() ==
                 rec[f"C4-shard{s}"]["dur_sha256"] for s in range(8))
    except (OSError, KeyError, ValueError):
        ok = None
    assert ok is not False, "gathered durations differ from the reference's hashes"
    if rank == 0:
        ups = N_UTT * args.steps / elapsed
        out = {
            "metric": "aligned utterances/sec, 512 variable-length utterances (T_mel <= 2000) batch-sharded over the GPUs",
            "value": round(ups, 1), "unit": "utterances/s", "frames_per_s": round(float(ty_all.sum()) * args.steps / elapsed, 1),
            "n_gpus": max(world, 1), "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 5), "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "configs[3]: 512 variable-length utterances [<=400 tokens, <=2000 frames] fp32, DP "
                                   "durations per rank + one RCCL all_gather of int32 durations per pass",
                       "utterances": N_UTT, "per_rank": cap, "parallelism": f"LPT batch shards x{max(world, 1)}"},
            "load_balance": {"dp_cost_max_over_mean": round(max(loads) / (sum(loads) / len(loads)), 4),
                             "dp_kernel_us_per_rank": [round(x, 1) for x in dp_all]},
            "durations_match_reference_hashes": ok,
        }
        _emit(out)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


MFMA_BF16_PEAK_TFLOPS = 2500.0      # MI355X_MICROARCH.md: dense bf16 matrix peak (not the 2:1-sparsity headline)


def _single_gpu_only(world: int, name: str):
    if world != 1:
        raise SystemExit(f"--config {name} is a one-GPU line (BASELINE names it for one MI355X); run it with --gpus 1")


def _timed_region(step, nsteps: int, dev) -> float:
    """EXACTLY nsteps steps bracketed by synchronize on both sides (one rank: no barrier partner)."""
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(nsteps):
        step()
    torch.cuda.synchronize(dev)
    return time.perf_counter() - t0


def _cpu_dp_baseline(value: np.ndarray, tx: np.ndarray, ty: np.ndarray, what: str, seconds: float = 8.0):
    """The reference's maximum_path_c (its pinned C restatement, one thread: the reference ships serial) on this config's
    score shape -- the part of the config the reference snapshot contains."""
    from oracle import maxpath_oracle as O
    paths = np.zeros(value.shape, np.int32)
    work = value.copy()
    O.maximum_path_c(paths, work, tx, ty)
    reps, spent = 0, 0.0
    while spent < seconds and reps < 500:
        np.copyto(work, value)
        paths.fill(0)
        t0 = time.perf_counter()
        O.maximum_path_c(paths, work, tx, ty)
        spent += time.perf_counter() - t0
        reps += 1
    nb = value.shape[0]
    return {"value": round(nb * reps / spent, 1), "unit": "utterances/s", "cores": 1, "kind": "port",
            "sample": f"maximum_path_c core (C restatement, oracle/maxpath_oracle.c) on {what}: {reps} batches in "
                      f"{spent:.1f}s ({spent / reps * 1e3:.1f} ms/batch); the conv / similarity / boundary-search stages "
                      f"have no counterpart in the reference snapshot", "host_logical_cores": os.cpu_count()}


def run_c3(args, world: int):
    """BASELINE configs[2]: the full OTA pipeline on an LJSpeech-shaped batch -- conv text / mel encoders -> similarity +
    log-softmax -> alignment search with dense path and durations; [64, 512-dim text embedding x 200, 80-mel x 900]."""
    _single_gpu_only(world, "c3")
    import aligner_amd
    _lib.require_gpu()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    Bc, Ct, Cm, Ca, Tx, Ty = 64, 512, 80, 80, 200, 900
    g = torch.Generator().manual_seed(33)
    params = aligner_amd.AlignmentEncoderParams.random(Ct, Cm, Ca, dev, seed=3)
    text = torch.randn(Bc, Ct, Tx, generator=g).to(dev)
    mel = torch.randn(Bc, Cm, Ty, generator=g).to(dev)
    tx = torch.full((Bc,), Tx, dtype=torch.int32, device=dev)
    ty = torch.full((Bc,), Ty, dtype=torch.int32, device=dev)
    out = {}

    def step():
        logp, _ = aligner_amd.alignment_encoder(text, mel, params, t_x=tx, pitched=not CONTIGUOUS_LOGP)
        out["logp"] = logp
        out["res"] = aligner_amd.align(logp, tx, ty, want_tok=True)

    for _ in range(max(args.warmup, 3)):
        step()
    elapsed = _timed_region(step, args.steps, dev)
    res, logp = out["res"], out["logp"]
    # guards on the timed outputs: every log-prob column is a distribution over the text, the path is one token per
    # frame, monotone, and its row sums are the durations
    col = torch.logsumexp(logp, dim=1)
    assert float(col.abs().max()) < 1e-3, "log-softmax columns do not sum to 1"
    assert int(res.durations.sum()) == Bc * Ty and bool((res.path.sum(1) == 1).all())
    assert bool((res.path.sum(2).to(torch.int32) == res.durations).all())
    assert bool((res.tok[:, 1:] - res.tok[:, :-1] >= 0).all()) and bool((res.tok[:, 1:] - res.tok[:, :-1] <= 1).all())
    # stage timings (HIP events on the launch stream, second pass over the same buffers)
    k1 = aligner_amd.softattn.conv1d(text, *params.key_proj[0], relu=True)
    k = aligner_amd.softattn.encode(text, params.key_proj)
    q = aligner_amd.softattn.encode(mel, params.query_proj)
    it = 20
    f1 = 2.0 * Bc * Tx * Ct * 2 * Ct * 3
    stages = {
        "text encoder, whole stack in one call (split pass + conv_gemm_kernel 512->1024 k3 + conv_narrow_ring_kernel 1024->80 k1)": (event_time_us(lambda: aligner_amd.softattn.encode(text, params.key_proj), it, dev), f1 + 2.0 * Bc * Tx * 2 * Ct * Ca),
        "text encoder conv 512->1024 k3 alone (split pass + conv_gemm_kernel, fp32 out)": (event_time_us(lambda: aligner_amd.softattn.conv1d(text, *params.key_proj[0], relu=True), it, dev), f1),
        "text encoder conv 1024->80 k1 alone (conv_narrow_kernel, fp32 input staged and split in the kernel)": (event_time_us(lambda: aligner_amd.softattn.conv1d(k1, *params.key_proj[1]), it, dev), 2.0 * Bc * Tx * 2 * Ct * Ca),
        "mel encoder, whole stack in one call (conv_narrow_fused_kernel: fp32 input staged and split in the kernel, the three layers in one kernel)": (event_time_us(lambda: aligner_amd.softattn.encode(mel, params.query_proj), it, dev), 2.0 * Bc * Ty * (Cm * 2 * Cm * 3 + 2 * Cm * Cm + Cm * Ca)),
        "similarity + log-softmax (softattn_rt_kernel)": (event_time_us(lambda: aligner_amd.soft_attention(k, q, t_x=tx, pitched=not CONTIGUOUS_LOGP), it, dev), 2.0 * Bc * Tx * Ty * Ca),
        "alignment search + dense path (maxpath_pipelined_kernel)": (event_time_us(lambda: aligner_amd.align(logp, tx, ty), it, dev), 0.0),
    }
    # beside the step: what an OTA training step adds on the same log-probs -- the forward-sum objective with its gradient
    # (published CTC form and plain form; both sweeps side by side in one launch + a combining pass)
    flat = logp.contiguous()             # (the objective's kernels take the contiguous layout)
    beside = {
        "forward-sum loss + gradient, CTC form (blank -1)": event_time_us(lambda: aligner_amd.forward_sum(flat, tx, ty, blank_logprob=-1.0), it, dev),
        "forward-sum loss only, CTC form": event_time_us(lambda: aligner_amd.forward_sum(flat, tx, ty, want_grad=False, blank_logprob=-1.0), it, dev),
        "forward-sum loss + gradient, plain form": event_time_us(lambda: aligner_amd.forward_sum(flat, tx, ty), it, dev),
    }
    dom = "text encoder conv 512->1024 k3 alone (split pass + conv_gemm_kernel, fp32 out)"     # the step's dominant launch pair
    tfl = stages[dom][1] / (stages[dom][0] * 1e-6) / 1e12
    ups = Bc * args.steps / elapsed
    line = {
        "metric": "aligned utterances/sec, full OTA pipeline (conv encoders -> similarity -> alignment search), "
                  "[B=64, 512-dim text x 200, 80-mel x 900]",
        "value": round(ups, 1), "unit": "utterances/s", "frames_per_s": round(ups * Ty, 1), "n_gpus": 1,
        "steps": args.steps, "warmup": max(args.warmup, 3), "ms_per_step": round(elapsed / args.steps * 1e3, 5),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "configs[2]: conv text encoder (512->1024 k3, ReLU, 1024->80 k1) + conv mel encoder (80->160 k3, "
                               "160->80, 80->80) + similarity (L2, C=80) + log-softmax + monotonic alignment search with dense "
                               "fp32 path and int32 durations; [64, 512x200 text embedding, 80x900 mel], eager launches",
                   "batch_per_gpu": Bc, "t_text": Tx, "t_mel": Ty},
        "roofline": {"bound": "mfma", "kernel": dom, "achieved": round(tfl, 1), "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                     "frac": round(tfl / MFMA_BF16_PEAK_TFLOPS, 4), "traffic": None, "kernel_us": round(stages[dom][0], 2),
                     "algorithmic_flops": stages[dom][1],
                     "note": "algorithmic flops of the fp32 convolution; the kernel issues three bf16 MFMAs per product "
                             "(hi*hi + hi*lo + lo*hi of the split operands), i.e. 3x these flops on the matrix cores",
                     "all_stages_us": {n: round(v[0], 2) for n, v in stages.items()},
                     "not_in_the_step_us": {n: round(v, 2) for n, v in beside.items()}},
    }
    if not args.no_cpu_baseline:
        line["cpu_baseline"] = _cpu_dp_baseline(logp.contiguous().cpu().numpy(), np.full(Bc, Tx, np.int32), np.full(Bc, Ty, np.int32),
                                                "this step's own [64,200,900] log-probs")
    _emit(line)


def run_c5(args, world: int):
    """BASELINE configs[4]: long-form [T_text=500, T_mel=4000] -- bf16 similarity, the alignment search on the bf16
    log-probs with an int32 path, and the MoBoAligner boundary search on the same energies."""
    _single_gpu_only(world, "c5")
    import hashlib
    import aligner_amd
    from aligner_amd import mobo
    _lib.require_gpu()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    Bc, Ca, Tx, Ty, D = 8, 80, 500, 4000, args.max_duration
    g = torch.Generator().manual_seed(55)
    kk = torch.randn(Bc, Ca, Tx, generator=g).to(dev)
    qq = torch.randn(Bc, Ca, Ty, generator=g).to(dev)
    tx = torch.full((Bc,), Tx, dtype=torch.int32, device=dev)
    ty = torch.full((Bc,), Ty, dtype=torch.int32, device=dev)
    out = {}

    def step():
        lp, _ = aligner_amd.soft_attention(kk, qq, t_x=tx, logp_dtype=torch.bfloat16)
        out["lp"] = lp
        out["res"] = aligner_amd.align(lp, tx, ty, path_dtype=torch.int32)
        out["bs"] = aligner_amd.boundary_search(lp, tx, ty, D)

    for _ in range(max(args.warmup, 3)):
        step()
    elapsed = _timed_region(step, args.steps, dev)
    lp, res, bs = out["lp"], out["res"], out["bs"]
    assert aligner_amd.read_status(dev) == 0 and mobo.read_status(dev) == 0
    # guards on the timed outputs
    assert int(res.durations.sum()) == Bc * Ty and bool((res.path.sum(1) == 1).all())
    dur = bs.durations
    assert bool((dur >= 1).all()) and bool((dur <= D).all()) and bool((dur.sum(1) == Ty).all())
    assert bool((bs.boundaries[:, -1] == Ty).all()) and bool(torch.isfinite(bs.map_score).all())
    # and the alignment search on the configuration's own synthetic scores: the reference's hash
    ref_ok = None
    try:
        with open(os.path.join(ROOT, "tests", "golden", "appendix_a.json")) as f:
            rec = json.load(f)
        v = torch.from_numpy(synth.synth_value(Bc, Tx, Ty, 5, bits=8, denom=8.0)).to(dev).to(torch.bfloat16)
        r5 = aligner_amd.align(v, tx, ty, path_dtype=torch.int32)
        ref_ok = hashlib.sha256(np.ascontiguousarray(r5.path.cpu().numpy().astype(np.int32)).tobytes()).hexdigest() == \
            rec["C5-longform"]["path_sha256"]
    except (OSError, KeyError, ValueError):
        ref_ok = None
    assert ref_ok is not False, "C5 path differs from the reference's hash"
    it = 10
    cells = Bc * Tx * Ty
    stages = {
        "similarity, bf16 log-probs (softattn_kernel)": (event_time_us(lambda: aligner_amd.soft_attention(kk, qq, t_x=tx, logp_dtype=torch.bfloat16), it, dev), 4 * Bc * Ca * (Tx + Ty) + 2 * cells),
        "alignment search on bf16 + int32 dense path (maxpath_pipelined_kernel, two workgroups per utterance, the path written inside the launch)": (event_time_us(lambda: aligner_amd.align(lp, tx, ty, path_dtype=torch.int32), it, dev), 2 * cells + 4 * cells),
        "alignment search, durations only": (event_time_us(lambda: aligner_amd.align(lp, tx, ty, want_path=False), it, dev), 2 * cells),
        f"boundary search, max duration {D} (norm + max-product chain + backtrack kernels)": (event_time_us(lambda: aligner_amd.boundary_search(lp, tx, ty, D), it, dev), 2 * cells + 4 * cells + 2 * cells + 4 * cells + 2 * cells),
    }
    # beside the step (a training step's extra): the search keeping log_alpha + gamma, and its gradient for a cotangent on
    # gamma -- guarded by the property every token row has (its energies only count up to a shift: the row's gradient sums to 0)
    soft = aligner_amd.boundary_search(lp, tx, ty, D, want_gamma=True)
    w = torch.randn(Bc, Tx, Ty, generator=g).to(dev)
    grad = aligner_amd.boundary_search_backward(lp, tx, ty, D, soft.log_alpha, None, w)
    gsc = float(grad.abs().max())
    assert mobo.read_status(dev) == 0 and bool(torch.isfinite(grad).all()) and gsc > 0
    assert float(grad.double().sum(2).abs().max()) < 2e-3 * gsc, "boundary search gradient: a token row does not sum to zero"
    training_extra = {
        "boundary search keeping log_alpha + gamma": round(event_time_us(lambda: aligner_amd.boundary_search(lp, tx, ty, D, want_gamma=True), it, dev), 2),
        "boundary search, log_alpha + gamma without the MAP sequence (a training step's forward pass)": round(event_time_us(lambda: aligner_amd.boundary_search(lp, tx, ty, D, want_gamma=True, want_map=False), it, dev), 2),
        "boundary search gradient, cotangent on gamma (norm + cotangent + chain + grad kernels)": round(event_time_us(lambda: aligner_amd.boundary_search_backward(lp, tx, ty, D, soft.log_alpha, None, w), it, dev), 2),
    }
    dom = max(stages, key=lambda n: stages[n][0])
    gbs = stages[dom][1] / (stages[dom][0] * 1e-6) / 1e9
    ups = Bc * args.steps / elapsed
    line = {
        "metric": "aligned utterances/sec, long-form [B=8, T_text=500, T_mel=4000]: bf16 similarity + alignment search (int32 "
                  "path) + MoBoAligner boundary search",
        "value": round(ups, 1), "unit": "utterances/s", "frames_per_s": round(ups * Ty, 1), "n_gpus": 1,
        "steps": args.steps, "warmup": max(args.warmup, 3), "ms_per_step": round(elapsed / args.steps * 1e3, 5),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16 scores, f32 arithmetic, i32 path",
        "data": "synthetic",
        "config": {"workload": f"configs[4]: similarity (L2, C=80) with bf16 log-probs -> maximum_path on them (dense int32 path + "
                               f"durations) -> boundary search with max duration {D} on the same energies; [8,500,4000], eager launches",
                   "batch_per_gpu": Bc, "t_text": Tx, "t_mel": Ty, "max_duration": D},
        "roofline": {"bound": "hbm", "kernel": dom, "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(gbs / HBM_PEAK_GBS, 4), "traffic": None, "kernel_us": round(stages[dom][0], 2),
                     "algorithmic_bytes": stages[dom][1],
                     "note": "boundary search bytes: energies read twice (2+2 B/cell: normalisers, chain), normalisers written and "
                             "read (4+4), durations written (2); it is bound by the dependent chain over the 500 token rows "
                             "(DESIGN 7.1), not by memory",
                     "all_stages_us": {n: round(v[0], 2) for n, v in stages.items()},
                     "not_in_the_step_us": training_extra},
        "alignment_path_matches_reference_hash_on_C5_scores": ref_ok,
    }
    if not args.no_cpu_baseline:
        line["cpu_baseline"] = _cpu_dp_baseline(synth.synth_value(Bc, Tx, Ty, 5, bits=8, denom=8.0), np.full(Bc, Tx, np.int32),
                                                np.full(Bc, Ty, np.int32), "the [8,500,4000] scores of SURVEY Appendix A")
    _emit(line)


_JSON_FD = None


def _keep_stdout_for_the_json_line() -> None:
    """Everything a library writes to file descriptor 1 from here on (RCCL prints its version banner there when the first
    communicator is made, gloo chats there too) goes to stderr; the JSON line goes to the descriptor stdout WAS."""
    global _JSON_FD
    if _JSON_FD is None:
        sys.stdout.flush()
        _JSON_FD = os.dup(1)
        os.dup2(2, 1)


def _emit(obj) -> None:
    data = (json.dumps(obj) + "\n").encode()
    if _JSON_FD is None:
        sys.stdout.write(data.decode())
        sys.stdout.flush()
    else:
        os.write(_JSON_FD, data)


def _self_launch(n: int) -> None:
    """`python bench.py --gpus N` without a launcher around it: run the N ranks as children of this process through
    torch.distributed.run (one rank per GPU, rendezvous on 127.0.0.1), let rank 0's JSON line through on stdout and leave
    with the children's exit code.  Nothing here touches the GPU -- the parent only counts devices -- so no process that
    has initialised HIP is ever replaced or forked."""
    import socket
    import subprocess
    if any(a in ("c3", "c5") or a.endswith("=c3") or a.endswith("=c5") for a in sys.argv[1:]):
        raise SystemExit("--config c3 / c5 are one-GPU lines (BASELINE names them for one MI355X); run them with --gpus 1")
    rehearse = os.environ.get("ALIGNER_BENCH_REHEARSE") == "1"
    have = torch.cuda.device_count()                # counts devices without creating a HIP context
    if have < n and not rehearse:
        raise SystemExit(f"--gpus {n} but only {have} GPU(s) visible (ALIGNER_BENCH_REHEARSE=1 runs all ranks on GPU 0 "
                         f"over gloo, for rehearsing the multi-rank path on a one-GPU box)")
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    # stderr inherited; of the ranks' stdout only rank 0's JSON line goes to ours (libraries chat on stdout: gloo does)
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    for ln in proc.stdout:
        if ln.startswith("{"):
            sys.stdout.write(ln)
            sys.stdout.flush()
        else:
            sys.stderr.write(ln)
    rc = proc.wait()
    if rc != 0:
        raise SystemExit(rc if rc > 0 else 1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=SUSTAINED_STEPS,
                    help="timed steps (default: a region long enough for the GPU's clocks and caches to have settled: ~0.6 s)")
    ap.add_argument("--warmup", type=int, default=2000)
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of a captured HIP graph")
    ap.add_argument("--gather-every", type=int, default=96,
                    help="steps per duration all-gather bucket (N>1): few, large collectives -- an all_gather every 18 steps "
                         "cost 8 %% of the step rate on one rank (the RCCL kernel takes CUs from the batches in flight), "
                         "every 96 steps 1.5 %%")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-side-kernels", action="store_true",
                    help="skip the event timings of the kernels the default step does not use (profiling passes: keeps the "
                         "per-kernel counters of the search kernel unmixed)")
    ap.add_argument("--no-repeats", dest="repeats", action="store_false",
                    help="skip the four extra timed regions (spread) and the one-batch-in-flight figure")
    ap.add_argument("--no-stream-path", action="store_true", help="ordinary stores for the dense path in every run")
    ap.add_argument("--path-mode", choices=["expand", "branch", "inline", "fused"], default="expand",
                    help="dense path of the batches-in-flight run: one streaming kernel behind the search (expand), or zeros on a "
                         "second graph branch beside the search + one 1 per frame behind it (branch).  The one-batch-at-a-time "
                         "figure always takes the branch form (--serial-path-mode)")
    ap.add_argument("--serial-path-mode", choices=["expand", "branch", "inline", "fused"], default="fused",
                    help="inline: zeros beside the similarity kernel, the search kernel writes the ones itself "
                         "(ALIGNER_F_PATH_PREZEROED)")
    ap.add_argument("--streams", type=int, default=3,
                    help="independent batches in flight (one HIP stream + buffer set each); 1 = strictly serial steps")
    ap.add_argument("--config", choices=["c2", "c3", "c4", "c5"], default="c2",
                    help="c2 (default): BASELINE configs[1], the headline; c3: configs[2], the full OTA pipeline; c4: configs[3], "
                         "the 512-utterance ragged job; c5: configs[4], long-form bf16 similarity + alignment + boundary search")
    ap.add_argument("--max-duration", type=int, default=32, help="c5: the boundary search's maximum-duration window")
    ap.add_argument("--contiguous-logp", action="store_true",
                    help="keep the step's intermediate log-probs contiguous ([B,Tx,Ty], rows of 4000 bytes) instead of at a row "
                         "pitch of whole 128-byte lines (A-B: the similarity kernel's stores and the search's loads then straddle lines)")
    args = ap.parse_args()
    global CONTIGUOUS_LOGP
    CONTIGUOUS_LOGP = bool(args.contiguous_logp)

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` started by hand (or by a driver that does not wrap it in torch.distributed.run): start
        # the N ranks ourselves, as fresh child processes, BEFORE anything in this process touches the GPU
        return _self_launch(args.gpus)

    _keep_stdout_for_the_json_line()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.config in ("c3", "c5"):
        if args.steps == SUSTAINED_STEPS and args.warmup == 2000:
            args.steps, args.warmup = 1000, 100          # a step is 0.2-0.6 ms of several launches: ~0.5 s (30 steps: 227-233 us at
                                                         # configs[2], the clocks still settling; 3 000: 211 us)
        return run_c3(args, world) if args.config == "c3" else run_c5(args, world)
    if args.config == "c4":
        if world != args.gpus:
            raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
        if args.steps == SUSTAINED_STEPS and args.warmup == 2000:
            args.steps, args.warmup = 500, 50            # a pass is ~0.1-1 ms: keep the default run short
        return run_c4(args, world, rank, local)
    dist = None
    # ALIGNER_BENCH_FORCE_DIST=1: take the multi-rank code path (RCCL group, duration buckets) with one rank
    if args.gpus > 1 or world > 1 or os.environ.get("ALIGNER_BENCH_FORCE_DIST") == "1":
        import torch.distributed as dist
        if world != args.gpus:
            raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if world == 1:                                  # the one-rank form of the multi-rank path, started by hand
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
            os.environ.setdefault("MASTER_PORT", "29531")
        # rehearsal hook: ALIGNER_BENCH_REHEARSE=1 runs all ranks on GPU 0 over gloo (a 1-GPU box cannot
        # host an RCCL group with more than one rank); the driver's real runs use RCCL, one GPU per rank
        if os.environ.get("ALIGNER_BENCH_REHEARSE") == "1":
            local = 0
            torch.cuda.set_device(0)
            dist.init_process_group("gloo")
        else:
            torch.cuda.set_device(local)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    _lib.require_gpu()
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)

    # S independent batches in flight: step i runs on stream i % S with its own buffers, so the
    # latency-bound DP of one batch (64 of 256 CUs) overlaps the bandwidth-bound kernels of the next
    nstreams = max(1, args.streams)
    # the dense path is an output nothing on the GPU reads back: with several batches in flight it is written with
    # non-temporal stores (ALIGNER_F_STREAM_PATH), so that it does not push the batches' score tensors out of the
    # caches; one batch at a time is faster with ordinary stores, and the serial figure is taken that way
    stream_path = nstreams > 1 and not args.no_stream_path
    steps = [Step(dev, seed=1234 + 17 * rank + 1000 * i, use_graph=not args.no_graph, stream_path=stream_path)
             for i in range(nstreams)]
    for st in steps:
        st.path_mode = args.path_mode
    streams = [torch.cuda.Stream(dev) for _ in range(nstreams)]
    for st, strm in zip(steps, streams):
        with torch.cuda.stream(strm):
            st.eager()
    torch.cuda.synchronize(dev)
    # duration gather (N>1): double-buffered buckets of `ge` steps, all-gathered over RCCL on a side
    # stream so the exchange overlaps the next bucket's compute.  The DP kernel of step i writes its
    # durations straight into buckets[(i // ge) % 2][i % ge]: one captured graph per (bucket, slot), `ge`
    # a multiple of the number of streams so that a slot always belongs to the same stream.  Per step the
    # host only launches a graph, as in the single-GPU run.
    comm_stream = torch.cuda.Stream(dev) if dist is not None else None
    buckets = gathered = None
    done = [None, None]
    ge = args.gather_every
    if dist is not None:
        # (a run shorter than a bucket gathers what it filled, not 96 slots of which 20 are used: the driver-shaped run's one
        # collective is a fifth of the bytes)
        ge = min(ge, max(args.steps, 1))
        ge = max(nstreams, (ge + nstreams - 1) // nstreams * nstreams)
        buckets = [torch.zeros((ge, B, TX), dtype=torch.int32, device=dev) for _ in range(2)]
        gathered = [torch.empty((world * ge, B, TX), dtype=torch.int32, device=dev) for _ in range(2)]
        for bi in range(2):
            for slot in range(ge):
                steps[slot % nstreams].capture_slot((bi, slot), buckets[bi][slot])
        torch.cuda.synchronize(dev)
    for st in steps:
        st.capture()
    step = steps[0]

    def run(nsteps: int, ns: int = 0):
        ns = ns or nstreams                        # batches in flight for this run (serial figure: 1)
        main = torch.cuda.current_stream(dev)
        for strm in streams:
            strm.wait_stream(main)
        if dist is None:
            for i in range(nsteps):
                with torch.cuda.stream(streams[i % ns]):
                    steps[i % ns]()
        else:
            for i in range(nsteps):
                k, bi, slot = i % nstreams, (i // ge) % 2, i % ge
                if slot < nstreams and done[bi] is not None:
                    streams[k].wait_event(done[bi])        # bucket bi's previous gather has read it
                with torch.cuda.stream(streams[k]):
                    steps[k].run_slot((bi, slot))
                if slot == ge - 1 or i == nsteps - 1:
                    for strm in streams:
                        comm_stream.wait_stream(strm)
                    with torch.cuda.stream(comm_stream):
                        dist.all_gather_into_tensor(gathered[bi], buckets[bi])
                        ev = torch.cuda.Event()
                        ev.record(comm_stream)
                    done[bi] = ev
        for strm in streams:
            main.wait_stream(strm)
        if dist is not None:
            main.wait_stream(comm_stream)

    def timed(nsteps: int, ns: int = 0) -> float:
        """EXACTLY nsteps steps bracketed by barrier + synchronize on both sides; max over ranks."""
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        run(nsteps, ns)
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)
        el = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([el], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el

    run(args.warmup)
    elapsed = timed(args.steps)                    # the reported figure: one region of exactly K steps
    # spread of that figure: the same region four more times (not part of `value`)
    extra = [timed(args.steps) for _ in range(4)] if args.repeats else []
    # ... and, where the reported region is short, one long one: a region of a few hundred steps is over within ~10 ms of the
    # first launch, while the GPU's clocks and caches are still settling (20 steps: 35-37 us a step, 200: 33-35, 200 000:
    # 28.7); not part of `value`
    sustained = None
    if args.repeats and dist is None and args.steps < SUSTAINED_STEPS:
        sustained = timed(SUSTAINED_STEPS)
    # strictly serial steps (one batch in flight: what a training step that waits for its alignment sees)
    serial_elapsed = None
    serial_expand_elapsed = None
    if dist is None and args.repeats:
        # one batch at a time: ordinary path stores (see above), and the dense path as zeros on a second graph branch
        # beside the alignment search + one 1 per frame behind it -- the 51 MB write is off the step's serial chain
        steps[0].expand_flags = 0
        steps[0].path_mode = args.serial_path_mode
        steps[0].capture()
        run(min(args.warmup, 10), 1)
        serial_elapsed = float(np.median([timed(args.steps, 1) for _ in range(3)]))
        if args.serial_path_mode != "expand":          # for comparison: the same step with the one-kernel path
            steps[0].path_mode = "expand"
            steps[0].capture()
            run(min(args.warmup, 10), 1)
            serial_expand_elapsed = float(np.median([timed(args.steps, 1) for _ in range(3)]))
        steps[0].expand_flags = _lib.F_STREAM_PATH if stream_path else 0
        steps[0].path_mode = args.path_mode
        steps[0].capture()

    # correctness guards on the timed outputs: every frame has exactly one token, durations sum to T_mel, the dense path
    # IS those durations; N>1: every bucket a gather has read holds complete steps, and what the gather delivered for
    # this rank is what this rank's kernels wrote
    guards = {"last_frame_on_last_token": all(bool((st.tok[:, -1] == TX - 1).all()) for st in steps)}
    if dist is None:
        guards["durations_sum_to_t_mel"] = all(int(st.dur.sum().item()) == B * TY for st in steps)
        guards["dense_path_is_the_durations"] = all(bool((st.path.sum(2).to(torch.int32) == st.dur).all()) for st in steps)
    else:
        filled = min(args.steps + args.warmup, ge)
        ok_sums, ok_gather = True, True
        for bi in range(2):
            sums = buckets[bi].sum(dim=(1, 2))
            ok_sums &= bool(((sums == B * TY) | (sums == 0)).all()) and (bi == 1 or int((sums == B * TY).sum()) >= min(filled, ge))
            if done[bi] is not None:
                ok_gather &= bool((gathered[bi][rank * ge:(rank + 1) * ge] == buckets[bi]).all())
                gs = gathered[bi].sum(dim=(1, 2))
                ok_gather &= bool(((gs == B * TY) | (gs == 0)).all())
        flags = torch.tensor([int(ok_sums), int(ok_gather)], dtype=torch.int32, device=dev)
        dist.all_reduce(flags, op=dist.ReduceOp.MIN)               # every rank's view
        guards["durations_sum_to_t_mel"] = bool(flags[0].item())
        guards["durations_match_gathered"] = bool(flags[1].item())
    # parity of the timed code path against the reference: one extra UNTIMED pass of the step's own search + dense-path
    # launches on the scores of SURVEY Appendix A's C2 case, whose path / duration hashes the reference produced
    # (tests/golden/appendix_a.json, made by tests/golden/make_golden.py from the real Cython core)
    ref_path = ref_dur = None
    if rank == 0:
        try:
            import hashlib
            with open(os.path.join(ROOT, "tests", "golden", "appendix_a.json")) as f:
                rec = json.load(f)["C2-fixed"]
            st = steps[0]
            torch.cuda.synchronize(dev)
            keep = st.logp.clone()
            st.logp.copy_(torch.from_numpy(synth.synth_value(*synth.CONFIGS["C2"])))
            st.forward()
            st.expand()
            torch.cuda.synchronize(dev)
            ref_path = hashlib.sha256(st.path.to(torch.int32).cpu().numpy().tobytes()).hexdigest() == rec["path_sha256"]
            ref_dur = hashlib.sha256(st.dur_out.cpu().numpy().astype(np.int32).tobytes()).hexdigest() == rec["dur_sha256"]
            st.search_with_path()                                   # the one-launch form of the serial figure, too
            torch.cuda.synchronize(dev)
            ref_path = ref_path and hashlib.sha256(st.path.to(torch.int32).cpu().numpy().tobytes()).hexdigest() == rec["path_sha256"]
            st.logp.copy_(keep)
            del keep
        except (OSError, KeyError, ValueError) as e:
            print(f"[bench] reference-hash pass skipped ({type(e).__name__}: {e})", file=sys.stderr)
    guards["path_matches_reference_hash"] = ref_path
    guards["durations_match_reference_hash"] = ref_dur
    bad = [k for k, v in guards.items() if v is False]
    assert not bad, f"guards failed on the timed outputs: {bad}"

    if rank == 0:
        n = max(world, 1)
        ups = n * B * args.steps / elapsed
        # ---- per-kernel timing with HIP events on the launch stream (second pass, same buffers) ----
        # (2 000 back-to-back launches each for the step's three kernels, a tenth as many untimed first.  Launched back to back
        # with itself the similarity kernel takes 14.4-14.5 us; between a search and a path kernel -- the one-batch-at-a-time
        # step under rocprofv3, 3 000 steps -- it averages 13.7: `roofline` reports the slower figure.  A HIP event pair per
        # launch inside the step does not resolve that: two events with nothing between them are 4.8 us apart.)
        it = 50
        t_sim = event_time_us(step.softattn, 2000, dev)
        t_fwd = event_time_us(step.forward, 2000, dev)
        t_exp = event_time_us(step.expand, 2000, dev)
        side_on = not args.no_side_kernels
        t_full = event_time_us(step.search_with_path, it, dev) if side_on else float("nan")
        t_zero = event_time_us(step.zero_path, it, dev) if side_on else float("nan")
        t_scat = event_time_us(step.scatter_path, it, dev) if side_on else float("nan")
        cells = B * TX * TY
        kernels = {
            SIM_KERNEL: {"us": t_sim, "bytes": 4 * B * C_ATT * (TX + TY) + 4 * cells},
            "maxpath_pipelined_kernel": {"us": t_fwd, "bytes": 4 * cells + 4 * B * (TX + TY)},
            "expand_kernel": {"us": t_exp, "bytes": 4 * cells + 4 * B * TY},
        }
        side = {"maxpath_pipelined_kernel (search + dense path in one launch)": {"us": t_full, "bytes": 8 * cells + 4 * B * (TX + TY)},
                "zero_path_kernel": {"us": t_zero, "bytes": 4 * cells},
                "scatter_path_kernel": {"us": t_scat, "bytes": 4 * B * TY + 4 * B * (TX + 1)}}
        # The roofline object is the similarity kernel's: north_star names THAT kernel for the HBM figure (it is
        # the HBM-bound one).  The kernel with the longest duration is the alignment search, which SURVEY 8d prices as bound
        # by neither HBM nor MFMA (a dependent chain of T_mel steps on B of the 256 CUs): it is reported beside it as
        # `dominant_by_time` with "bound": "latency", its bytes-per-second for information only.
        dom = SIM_KERNEL
        longest = max(kernels, key=lambda k: kernels[k]["us"])
        ach = kernels[dom]["bytes"] / (kernels[dom]["us"] * 1e-6) / 1e9
        # HBM bytes per launch from the PMC counters: collected in separate rocprofv3 --pmc passes of this same
        # command (tools/pmc_traffic.py; gfx950 FETCH_SIZE correction applied) and committed under profiles/ --
        # not measured by this run, so the line names the file (and with it the build) the figure comes from
        traffic, traffic_src, traffic_all = None, None, {}
        for name in ("r05_pmc_hbm_traffic.json",):     # (of THIS build's kernels: older files describe other kernels)
            try:
                with open(os.path.join(ROOT, "profiles", name)) as f:
                    traffic_all = json.load(f)["kernels"]
                traffic = traffic_all.get(dom, {}).get("hbm_bytes")
                traffic_src = "profiles/" + name
                break
            except (OSError, ValueError, KeyError):
                continue
        # wave occupancy of the search kernel from the SQ counters (rocprofv3 --pmc pass of this command,
        # tools/pmc_issue.py -> profiles/r05_pmc_issue_counters.json): resident waves over the chip's wave slots while
        # the kernel runs; falls back to the launch geometry (B workgroups x 8 waves / 8192 slots) when the file is absent
        occ, occ_src = round(B * 8 / (256 * 32), 4), "launch geometry (B workgroups x 8 waves / 8192 wave slots)"
        try:
            with open(os.path.join(ROOT, "profiles", "r05_pmc_issue_counters.json")) as f:
                pmc = json.load(f)["kernels"]["maxpath_pipelined_kernel"]
            occ, occ_src = pmc["wave_occupancy"], "profiles/r05_pmc_issue_counters.json (4 x SQ_WAVE_CYCLES / (GRBM_GUI_ACTIVE / 8) / 8192 wave slots)"
        except (OSError, ValueError, KeyError):
            pass
        dp_bytes, dp_us = kernels[longest]["bytes"], kernels[longest]["us"]
        roofline = {"bound": "hbm", "kernel": dom, "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_src,
                    "kernel_us": round(kernels[dom]["us"], 2), "algorithmic_bytes": kernels[dom]["bytes"],
                    "dominant_by_time": {
                        "kernel": longest, "bound": "latency", "kernel_us": round(dp_us, 2),
                        "algorithmic_bytes": dp_bytes, "achieved": round(dp_bytes / (dp_us * 1e-6) / 1e9, 1), "unit": "GB/s",
                        "frac_of_hbm_peak": round(dp_bytes / (dp_us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
                        "traffic": traffic_all.get(longest, {}).get("hbm_bytes"),
                        "wave_occupancy": occ, "wave_occupancy_source": occ_src,
                        # informational: this launch runs on B of the 256 CUs, and one CU takes in at most
                        # ~64 GB/s (the guide's 66-73 GB/s for L2-resident rows): its memory bound is B x 64 GB/s
                        "launch_fetch_bound_GBps": round(min(B, 256) * 64.0, 1),
                        "note": "SURVEY 8d: bound by neither HBM nor MFMA -- a dependent chain of T_mel frames on B of the "
                                "256 CUs; bytes per second are for information"},
                    "dp_wave_occupancy": occ,
                    # the whole step against the same peak: algorithmic bytes of its three launches over the measured
                    # time per step on ONE GPU (batches in flight overlap the kernels)
                    "step_aggregate": {
                        "algorithmic_bytes": int(sum(v["bytes"] for v in kernels.values())),
                        "achieved": round(sum(v["bytes"] for v in kernels.values()) / (elapsed / args.steps) / 1e9, 1),
                        "unit": "GB/s per GPU",
                        "frac": round(sum(v["bytes"] for v in kernels.values()) / (elapsed / args.steps) / 1e9
                                      / HBM_PEAK_GBS, 4),
                        "cu_time_us": round(sum(v["us"] * (min(B, 256) / 256.0 if k == longest else 1.0)
                                                for k, v in kernels.items()), 2),
                        "note": "cu_time_us = sum of the kernels' durations weighted by the share of the 256 CUs each one "
                                "occupies (the search: B of them): the step cannot be shorter than that however the batches "
                                "in flight interleave"},
                    "all_kernels": {k: {"us": round(v["us"], 2),
                                        "GBps": round(v["bytes"] / (v["us"] * 1e-6) / 1e9, 1),
                                        "frac_of_8000": round(v["bytes"] / (v["us"] * 1e-6) / 1e9 / HBM_PEAK_GBS, 3),
                                        "frac_of_6300_achievable": round(v["bytes"] / (v["us"] * 1e-6) / 1e9 / 6300.0, 3)}
                                    for k, v in {**kernels, **(side if side_on else {})}.items()}}
        out = {
            "metric": "aligned utterances/sec + frames/sec, [B=64,T_text=200,T_mel=1000]",
            "value": round(ups, 1), "unit": "utterances/s", "frames_per_s": round(ups * TY, 1),
            "n_gpus": n, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 5),
            "ms_per_step_repeats": ([round(elapsed / args.steps * 1e3, 5)] +
                                    [round(x / args.steps * 1e3, 5) for x in extra]) if extra else None,
            "ms_per_step_median": round(float(np.median([elapsed] + extra)) / args.steps * 1e3, 5) if extra else None,
            "sustained": ({"steps": SUSTAINED_STEPS, "ms_per_step": round(sustained / SUSTAINED_STEPS * 1e3, 5),
                           "value": round(B * n * SUSTAINED_STEPS / sustained, 1), "unit": "utterances/s",
                           "note": "one more region of that many steps behind the reported one (not part of `value`): the "
                                   "step rate once the GPU's clocks and caches have settled"} if sustained else None),
            "serial_ms_per_step": round(serial_elapsed / args.steps * 1e3, 5) if serial_elapsed else None,
            "serial_ms_per_step_expand_kernel": (round(serial_expand_elapsed / args.steps * 1e3, 5)
                                                 if serial_expand_elapsed else None),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "configs[1]: similarity (L2, C=80) + monotonic alignment search, "
                                   "[B=64,T_text=200,T_mel=1000] fp32 per GPU, dense fp32 path + int32 durations out",
                       "batch_per_gpu": B, "t_text": TX, "t_mel": TY, "c_att": C_ATT,
                       "launch": "hipGraph" if step.graph is not None else "eager",
                       "batches_in_flight": nstreams,
                       "intermediate": ("log-probs [B,Tx,Ty] contiguous (--contiguous-logp)" if CONTIGUOUS_LOGP else
                                        f"log-probs [B,Tx,Ty] at a row pitch of {step.ld} elements (rows on whole 128-byte lines: "
                                        "aligner_softattn_ld -> aligner_maxpath_ld); inputs and outputs contiguous"),
                       "path_stores": ("non-temporal (ALIGNER_F_STREAM_PATH) with the batches in flight, ordinary in the "
                                       "serial figure" if stream_path else "ordinary"),
                       "dense_path": {"batches_in_flight": args.path_mode, "serial": args.serial_path_mode,
                                      "note": "fused = aligner_maxpath: one launch, zero workgroups on the idle CUs beside the search, "
                                              "the utterances' workgroups write the ones; inline = zeros on a second graph branch beside the similarity kernel, the search "
                                              "kernel writes the ones (ALIGNER_F_PATH_PREZEROED); branch = zeros beside the "
                                              "search, then a scatter kernel; expand = one kernel behind the search"},
                       "parallelism": f"batch-sharded x{n}" + (f", RCCL all_gather of durations every "
                                                               f"{ge} steps" if n > 1 else "")},
            "roofline": roofline,
            "guards": guards,
        }
        if n == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(dev)
            out["speedup_vs_cpu_1thread"] = round(ups / out["cpu_baseline"]["value"], 1)
        _emit(out)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
