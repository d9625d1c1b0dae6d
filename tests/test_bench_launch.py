"""`python bench.py --gpus N` starts its own ranks (no torch.distributed.run around it) -- CPU-side checks.

The N-rank path itself needs GPUs; here: the parent never touches the GPU, spawns the ranks through
torch.distributed.run on 127.0.0.1, and propagates a failing child (on this GPU-less container every rank fails loudly in
`_lib.require_gpu()`, which is exactly the exit the parent must hand on)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra, timeout=300):
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    env.pop("LOCAL_RANK", None)
    env.update(env_extra)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, cwd=ROOT, capture_output=True,
                          text=True, timeout=timeout)


def _no_gpu():
    import torch
    return torch.cuda.device_count() == 0


@pytest.mark.skipif(not _no_gpu(), reason="checks the GPU-less behaviour")
def test_self_launch_refuses_more_ranks_than_gpus():
    r = _run(["--gpus", "2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"], {})
    assert r.returncode != 0
    assert "only 0 GPU(s) visible" in r.stderr


@pytest.mark.skipif(not _no_gpu(), reason="checks the GPU-less behaviour")
def test_self_launch_starts_ranks_and_propagates_their_failure():
    r = _run(["--gpus", "2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"], {"ALIGNER_BENCH_REHEARSE": "1"})
    assert r.returncode != 0                                    # the ranks' failure is the parent's
    # the children were really started by torch.distributed.run (its failure report names both ranks) ...
    assert "bench.py FAILED" in r.stderr or "ChildFailedError" in r.stderr
    # ... and died where the product refuses to run without a GPU (no fallback of any kind)
    assert "GPU" in r.stderr
    assert r.stdout.strip() == ""                              # no JSON line from a failed job


def test_one_gpu_configs_refuse_n_ranks():
    r = _run(["--gpus", "2", "--config", "c3"], {"ALIGNER_BENCH_REHEARSE": "1"})
    assert r.returncode != 0 and "one-GPU" in r.stderr


@pytest.mark.gpu
def test_self_launched_two_ranks_on_one_gpu_print_one_line():
    """On a GPU box: `bench.py --gpus 2` without a launcher starts two ranks itself (rehearsal form: both on GPU 0, the
    gather over gloo), rank 0 prints ONE JSON line for n_gpus = 2 and every guard on the timed outputs holds -- incl. that
    what the gather delivered for a rank is what that rank's kernels wrote."""
    import json
    r = _run(["--gpus", "2", "--steps", "6", "--warmup", "2", "--no-cpu-baseline", "--no-side-kernels"],
             {"ALIGNER_BENCH_REHEARSE": "1"}, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 6 and d["scaling"] == "weak"
    assert d["guards"] and all(v is not False for v in d["guards"].values()) and d["guards"]["durations_match_gathered"] is True


@pytest.mark.gpu
def test_one_rank_rccl_path_of_the_headline_bench():
    """The multi-rank code path of `bench.py` with the REAL backend on one rank (ALIGNER_BENCH_FORCE_DIST=1: a 1-rank
    RCCL process group, the durations gathered in buckets on the communication stream, captured graphs): the first
    `--gpus 8` run is the driver's, this is what can be exercised on one GPU.  One JSON line, every guard true, and the
    gather gave back what the kernels wrote."""
    import json
    r = _run(["--gpus", "1", "--steps", "20", "--warmup", "5", "--no-cpu-baseline", "--no-side-kernels"],
             {"ALIGNER_BENCH_FORCE_DIST": "1", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29561", "RANK": "0", "WORLD_SIZE": "1",
              "LOCAL_RANK": "0"}, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    assert r.stdout.strip() == lines[0], "stdout carries the JSON line and nothing else (RCCL's banner goes to stderr)"
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["steps"] == 20 and d["warmup"] == 5
    g = d["guards"]
    assert g and all(v is not False for v in g.values())
    assert g["durations_match_gathered"] is True and g["path_matches_reference_hash"] is True
    assert d["roofline"]["kernel"] == "softattn_rt_kernel" and d.get("cpu_baseline") is None


@pytest.mark.gpu
@pytest.mark.parametrize("layout", ["pitched", "contiguous"])
def test_headline_bench_line_in_both_layouts_of_its_intermediate(layout):
    """`bench.py` as the driver runs it (`--steps 20 --warmup 5`), with the step's log-probs at a row pitch of whole 128-byte
    lines (the default: aligner_softattn_ld -> aligner_maxpath_ld) and contiguous (`--contiguous-logp`): one JSON line, the
    reference's path and duration hashes in both, `roofline` and (skipped here) `cpu_baseline` in their places."""
    import json
    args = ["--steps", "20", "--warmup", "5", "--no-cpu-baseline", "--no-side-kernels", "--no-repeats"]
    r = _run(args + (["--contiguous-logp"] if layout == "contiguous" else []), {}, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["steps"] == 20 and d["unit"] == "utterances/s" and d["value"] > 0
    assert ("row pitch of 1024" in d["config"]["intermediate"]) == (layout == "pitched")
    g = d["guards"]
    assert g["path_matches_reference_hash"] is True and g["durations_match_reference_hash"] is True
    assert all(v is not False for v in g.values())
    assert d["roofline"]["kernel"] == "softattn_rt_kernel" and 0.2 < d["roofline"]["frac"] < 1.0


@pytest.mark.gpu
def test_c4_as_a_two_rank_job_matches_the_reference_hashes():
    """BASELINE configs[3] (512 ragged utterances, batch-sharded, durations gathered) self-launched with two ranks in the
    rehearsal form (both on GPU 0, gloo): every shard's gathered durations hash to SURVEY Appendix A's values."""
    import json
    r = _run(["--config", "c4", "--gpus", "2", "--steps", "3", "--warmup", "1", "--no-cpu-baseline"],
             {"ALIGNER_BENCH_REHEARSE": "1"}, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["config"]["utterances"] == 512
    assert d["durations_match_reference_hashes"] is True
    assert len(d["load_balance"]["dp_kernel_us_per_rank"]) == 2
