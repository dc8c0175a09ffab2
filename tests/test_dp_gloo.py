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


def _rdzv_worker(rank, world, tag, out_dir):
    sys.path.insert(0, os.path.join(ROOT, "eigen-lstm_amd"))
    import dp
    r = dp.Rendezvous(rank, world, tag, timeout=60.0)
    got = r.allgather((rank, rank * 1.5))
    b = r.broadcast(b"x" * 128 if rank == 0 else None)
    r.barrier()
    np.save(os.path.join(out_dir, f"r{rank}.npy"), np.array([g[1] for g in got] + [len(b)]))
    r.close()


def test_socket_rendezvous_used_by_the_bench(tmp_path):
    """bench.py's N>1 plumbing (dp.Rendezvous: rank-0 star over TCP, port published through a file)."""
    import torch.multiprocessing as mp
    world = 4
    mp.spawn(_rdzv_worker, args=(world, f"test_{os.getpid()}", str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        v = np.load(tmp_path / f"r{r}.npy")
        assert list(v) == [0.0, 1.5, 3.0, 4.5, 128.0]


def test_bench_multi_rank_plumbing_under_torchrun():
    """`python -m torch.distributed.run --nproc-per-node 2 bench.py --gpus 2` exactly as the driver launches it, with
    the GPU library replaced by tests/fake_gpu: rendezvous, RCCL-id broadcast, barriers, max-over-ranks, ONE JSON
    line on rank 0 with whole-job throughput."""
    import json
    import subprocess
    env = dict(os.environ, LSTM_BENCH_FAKE_GPU="1")
    port = 29700 + (os.getpid() % 200)
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"),
                          "--gpus", "2", "--steps", "5", "--warmup", "2", "--no-cpu-baseline"],
                         capture_output=True, text=True, env=env, timeout=240)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 5 and d["scaling"] == "weak" and d["unit"] == "chars/s"
    assert d["config"]["parallelism"] == "dp2" and "FAKE GPU" in d["data"]
    # whole-job aggregate: both ranks' characters over the slowest rank's time
    assert abs(d["value"] - 99 * 64 * 5 * 2 / (d["ms_per_step"] * 5e-3)) / d["value"] < 1e-3
    assert d["loss_first_last"] == [2.0, 2.0]  # sum over ranks of each rank's share
