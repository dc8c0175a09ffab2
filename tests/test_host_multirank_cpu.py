"""CPU: the C++ host program's --gpus G path (eigen-lstm_amd/host/lstm_main.cc) against a GPU-less stub of the C ABI
(tests/fake_gpu/lstm_hip_stub.c, built here as liblstm_hip.so and put first on the loader path).

What the real path must do and the stub lets us see: one process per GPU, forked BEFORE anything of the library is called
(a process that has touched the GPU must not fork or exec); the parent only relays the RCCL id and the epoch losses;
the epoch report is the sum of the ranks' shares of the global-batch loss; the mid-epoch report scales the lead's share to
the global batch; a rank that dies ends the whole job with a non-zero status instead of leaving its peers blocked in an
all-reduce.  Multi-GPU hardware is not available to this repo's own runs; this keeps the plumbing honest."""
import os
import re
import subprocess
import sys
import time

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST_DIR = os.path.join(ROOT, "eigen-lstm_amd", "host")


@pytest.fixture(scope="module")
def stub_env(tmp_path_factory):
    d = tmp_path_factory.mktemp("stub")
    so = d / "liblstm_hip.so"
    subprocess.check_call(["gcc", "-O1", "-fPIC", "-shared", "-Wall", os.path.join(ROOT, "tests", "fake_gpu", "lstm_hip_stub.c"),
                           "-o", str(so)])
    exe = d / "lstm_stub_linked"
    # the program itself, unchanged, linked against the stub (no rpath to the real library)
    subprocess.check_call(["g++", "-O1", "-std=c++17", os.path.join(HOST_DIR, "lstm_main.cc"), "-o", str(exe), "-L" + str(d),
                           "-llstm_hip", "-Wl,-rpath," + str(d)])
    text = d / "corpus.txt"
    np.random.RandomState(3).randint(97, 123, size=4000).astype(np.uint8).tofile(text)
    return d, str(exe), str(text)


def _log(path):
    rows = []
    for line in open(path):
        pid, rank, rest = line.rstrip("\n").split(" ", 2)
        rows.append((int(pid), int(rank), rest))
    return rows


def test_two_ranks_fork_before_any_library_call_and_sum_their_losses(stub_env):
    d, exe, text = stub_env
    log = d / "calls_ok.log"
    env = dict(os.environ, LSTM_STUB_LOG=str(log), LSTM_STUB_LOSS="2.0")
    N, S, B, windows = 16, 8, 6, 30
    p = subprocess.Popen([exe, text, str(N), str(S), str(B), "0.1", "--gpus", "2", "--epochs", "2", "--windows", str(windows),
                          "--sample", "0", "--quiet"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env)
    out, err = p.communicate(timeout=60)
    assert p.returncode == 0, err + out
    rows = _log(log)
    pids = {pid for pid, _, _ in rows}
    assert p.pid not in pids, "the parent process called into the library: it must only fork and relay"
    creates = [(pid, rest) for pid, _, rest in rows if rest.startswith("create")]
    assert len(creates) == 2 and len({pid for pid, _ in creates}) == 2            # one process per rank
    assert sorted(rest.split()[1] for _, rest in creates) == ["device=0", "device=1"]
    assert all("B=3" in rest for _, rest in creates)                               # streams sharded: 6 / 2 per rank
    inits = sorted(rest for _, _, rest in rows if rest.startswith("comm_init"))
    assert inits == ["comm_init nranks=2 rank=0 id_ok=1", "comm_init nranks=2 rank=1 id_ok=1"]   # the id was relayed intact
    assert sum(rest == "comm_unique_id" for _, _, rest in rows) == 1               # made once, by rank 0
    assert sorted(rest for _, _, rest in rows if rest.startswith("set_global_batch")) == ["set_global_batch 6"] * 2
    # every call of a rank comes from that rank's own process
    by_pid = {}
    for pid, rank, rest in rows:
        if rank >= 0:
            by_pid.setdefault(pid, set()).add(rank)
    assert all(len(r) == 1 for r in by_pid.values())
    # epoch report: each rank reports 2.0 * (S-1) * (3/6) bits per window; the relay adds the two shares
    m = re.findall(r"avg loss = ([\d.]+) bits/char", out)
    assert len(m) == 2, out
    want = 2.0 * (S - 1) * windows / (S * (windows + S))                            # R/lstm.cc:290: loss / (S * length)
    assert abs(float(m[0]) - want) < 1e-3 and abs(float(m[1]) - want) < 1e-3, (m, want)
    assert "2 GPUs" in out


def test_mid_epoch_report_is_scaled_to_the_global_batch(stub_env):
    d, exe, text = stub_env
    log = d / "calls_mid.log"
    env = dict(os.environ, LSTM_STUB_LOG=str(log), LSTM_STUB_LOSS="2.0")
    N, S, B = 16, 8, 8
    out = subprocess.run([exe, text, str(N), str(S), str(B), "0.1", "--gpus", "2", "--epochs", "1", "--windows", "300", "--sample",
                          "0", "--quiet", "--test-every", "0.000001", "--log", str(d / "mid")],
                         capture_output=True, text=True, env=env, timeout=60)
    assert out.returncode == 0, out.stderr + out.stdout
    errs = [float(x) for x in re.findall(r"Train error: ([\d.eE+-]+), Test error", out.stdout)]
    assert len(errs) >= 2, out.stdout
    # mid-epoch rows: the lead's share (half of the global-batch loss) times the rank count: 2.0 * (S-1) / S bits per char
    for e in errs[:-1]:
        assert abs(e - 2.0 * (S - 1) / S) < 1e-6, errs
    # only the lead writes the log's checkpoint and sample file (the sampler is local to one handle)
    samplers = {rank for _, rank, rest in _log(log) if rest.startswith("sample ")}
    assert samplers == {0}


def test_a_dying_rank_ends_the_job(stub_env):
    d, exe, text = stub_env
    log = d / "calls_fail.log"
    env = dict(os.environ, LSTM_STUB_LOG=str(log), LSTM_STUB_FAIL_RANK="1")
    t0 = time.time()
    out = subprocess.run([exe, text, "16", "8", "4", "0.1", "--gpus", "2", "--epochs", "1", "--windows", "250", "--sample", "0",
                          "--quiet"], capture_output=True, text=True, env=env, timeout=90)
    took = time.time() - t0
    assert out.returncode != 0, out.stdout
    assert took < 30, f"the surviving rank (asleep for 120 s inside its 'all-reduce') was waited for: {took:.0f} s"
    rows = _log(log)
    assert any(rest == "train_windows FAIL" and rank == 1 for _, rank, rest in rows)
    assert any(rest == "train_windows HANG" and rank == 0 for _, rank, rest in rows)
    assert "injected failure on rank 1" in out.stderr + out.stdout                 # the library's error text reaches the user
