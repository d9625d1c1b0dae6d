"""Host logic of the multi-GPU path on CPU: world_size-2 gloo group.
The compute inside each rank is the ORACLE here (injected as align_fn) -- the test
checks partitioning, padding, the all-gather and the un-permutation, not the kernels;
the product align_fn (HIP) is exercised by the gpu-marked tests."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from aligner_amd import sharded, synth


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _oracle_align(value, tx, ty):
    from oracle import maxpath_oracle as O
    v = value.numpy().astype(np.float32).copy()
    p = np.zeros(v.shape, np.int32)
    O.maximum_path_c(p, v, tx.numpy().astype(np.int32).copy(), ty.numpy().astype(np.int32).copy())
    return torch.from_numpy(p.sum(2).astype(np.int32))


def _oracle_boundary_durations(D):
    def fn(value, tx, ty):
        from oracle import mobo_oracle as M
        out = np.zeros((value.shape[0], value.shape[1]), np.int32)
        for b in range(value.shape[0]):
            I, J = int(tx[b]), int(ty[b])
            out[b, :I] = M.boundary_search_fast(value[b, :I, :J].numpy().astype(np.float64), D)["durations"]
        return torch.from_numpy(out)
    return fn


def _worker_bs(rank, world, port, n, Tx, Ty, D, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        tx, ty = synth.synth_lengths(n, Tx, Ty // 2, Ty, 9)
        full_value = synth.synth_value(n, Tx, Ty, 78)
        got = sharded.sharded_align(lambda idx: torch.from_numpy(full_value[idx]), tx, ty, Tx,
                                    align_fn=_oracle_boundary_durations(D))
        np.save(os.path.join(out_dir, f"rank{rank}.npy"), got.numpy())
    finally:
        dist.destroy_process_group()


def test_sharded_boundary_search_equals_single_process(tmp_path):
    """The same host logic with the boundary search as the per-rank compute (the oracle here; the product's
    sharded.boundary_durations(D) runs the HIP kernels: tests/test_mobo.py)."""
    world, n, Tx, Ty, D = 2, 7, 12, 60, 16
    port = _free_port()
    mp.spawn(_worker_bs, args=(world, port, n, Tx, Ty, D, str(tmp_path)), nprocs=world, join=True)
    tx, ty = synth.synth_lengths(n, Tx, Ty // 2, Ty, 9)
    want = _oracle_boundary_durations(D)(torch.from_numpy(synth.synth_value(n, Tx, Ty, 78)), torch.from_numpy(tx),
                                         torch.from_numpy(ty)).numpy()
    assert (want.sum(1) == ty).all()
    for r in range(world):
        assert np.array_equal(np.load(os.path.join(tmp_path, f"rank{r}.npy")), want), f"rank {r}"


def _worker(rank, world, port, n, Tx, Ty, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        tx, ty = synth.synth_lengths(n, Tx, Ty // 4, Ty, 9)
        full_value = synth.synth_value(n, Tx, Ty, 77)

        def value_of(idx):
            return torch.from_numpy(full_value[idx])

        got = sharded.sharded_align(value_of, tx, ty, Tx, align_fn=_oracle_align)
        np.save(os.path.join(out_dir, f"rank{rank}.npy"), got.numpy())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n", [(2, 13), (2, 16), (4, 13), (4, 3)])
def test_sharded_equals_single_process(tmp_path, world, n):
    """world 2 and 4; 13 utterances (shards of 4/3/3/3: padded blocks, index column) and fewer utterances than
    ranks (an empty shard still takes part in the collective)."""
    Tx, Ty = 24, 96
    port = _free_port()
    mp.spawn(_worker, args=(world, port, n, Tx, Ty, str(tmp_path)), nprocs=world, join=True)
    tx, ty = synth.synth_lengths(n, Tx, Ty // 4, Ty, 9)
    want = _oracle_align(torch.from_numpy(synth.synth_value(n, Tx, Ty, 77)), torch.from_numpy(tx),
                         torch.from_numpy(ty)).numpy()
    for r in range(world):
        got = np.load(os.path.join(tmp_path, f"rank{r}.npy"))
        assert np.array_equal(got, want), f"rank {r}"


def test_lpt_partition_properties():
    tx, ty = synth.synth_lengths(512, 400, 200, 2000, 4)          # BASELINE config C4 lengths
    plan = sharded.lpt_partition(tx, ty, 8)
    allidx = np.sort(np.concatenate(plan))
    assert np.array_equal(allidx, np.arange(512))
    assert {len(p) for p in plan} == {64}
    cost = sharded.dp_cost(tx, ty)
    loads = np.array([cost[p].sum() for p in plan], dtype=np.float64)
    assert loads.max() / loads.mean() < 1.02                       # within 2 % of perfect balance
    naive = np.array([cost[64 * r:64 * r + 64].sum() for r in range(8)], dtype=np.float64)
    assert loads.max() <= naive.max()
    # deterministic
    plan2 = sharded.lpt_partition(tx, ty, 8)
    assert all(np.array_equal(a, b) for a, b in zip(plan, plan2))
    # ragged world sizes
    plan3 = sharded.lpt_partition(tx[:13], ty[:13], 4)
    assert sorted(len(p) for p in plan3) == [3, 3, 3, 4]
