"""world_size-2 gloo test (CPU) of the data-parallel decomposition in eigen-lstm_amd/dp.py:
two ranks, each running the window ops on its half of the streams and exchanging ONE SUM all-reduce
of the flat gradient block per window, follow the single-process full-batch run."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, N, S, B, windows, lr, out_dir):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    sys.path.insert(0, os.path.join(ROOT, "eigen-lstm_amd"))
    import torch
    import torch.distributed as dist
    import dp
    from oracle_lib import Oracle
    torch.set_num_threads(1)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    orc = Oracle("f32")
    text = (np.random.RandomState(7).randint(32, 127, size=600)).astype(np.uint8)
    first, per = dp.shard(rank, world, B)
    tr = orc.trainer(text, N, S, per, lr=lr, seed=1, stream0=first, streams_total=B)
    assert np.array_equal(dp.cursors(len(text), S, rank, world, B),
                          [S + ((first + b) * (len(text) - S)) // B for b in range(per)])
    # epoch state for the GLOBAL batch from one stream; each rank keeps its columns
    full = orc.trainer(text, N, S, B, lr=lr, seed=1)
    full.epoch_reset()
    tr.h[:] = full.h[:, first:first + per]
    tr.c[:] = full.c[:, first:first + per]
    losses = []
    for _ in range(windows):
        local = tr.window(update=False)
        g = torch.from_numpy(tr.grads)          # shares memory with the trainer's gradient block
        dist.all_reduce(g, op=dist.ReduceOp.SUM)
        orc.adagrad(tr.params, tr.grads, tr.mem, lr)
        l = torch.tensor([dp.local_loss_to_global(local, per, B)], dtype=torch.float64)
        dist.all_reduce(l, op=dist.ReduceOp.SUM)
        losses.append(float(l))
    np.save(os.path.join(out_dir, f"params_{rank}.npy"), np.array(tr.params))
    np.save(os.path.join(out_dir, f"losses_{rank}.npy"), np.array(losses))
    dist.destroy_process_group()


def test_two_rank_gloo_matches_full_batch(tmp_path, oracle32):
    import torch.multiprocessing as mp
    N, S, B, windows, lr = 16, 6, 8, 12, 0.1
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, N, S, B, windows, lr, str(tmp_path)), nprocs=2, join=True)
    p0, p1 = np.load(tmp_path / "params_0.npy"), np.load(tmp_path / "params_1.npy")
    assert np.array_equal(p0, p1)  # replicas stay bit-identical: same summed gradients, same Adagrad
    text = (np.random.RandomState(7).randint(32, 127, size=600)).astype(np.uint8)
    full = oracle32.trainer(text, N, S, B, lr=lr, seed=1)
    full.epoch_reset()
    want = np.array([full.window() for _ in range(windows)])
    got = np.load(tmp_path / "losses_0.npy")
    # only the cross-rank summation order differs from the single-process run
    assert np.abs(got[:3] - want[:3]).max() <= 1e-4
    assert np.abs(got - want).max() / (S - 1) <= 0.05
    d = np.abs(p0 - full.params)
    assert np.median(d) <= 1e-5


def test_shard_helpers():
    import dp
    assert dp.shard(3, 8, 512) == (192, 64)
    with pytest.raises(ValueError):
        dp.shard(0, 3, 64)
    assert dp.local_loss_to_global(8.0, 16, 64) == 2.0
