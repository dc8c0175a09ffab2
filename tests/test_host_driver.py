"""-m gpu: the C++ host program (eigen-lstm_amd/lstm) end to end: command line in, the reference's
stdout report out, checked against the oracle's restatement of the same loop with the same seed."""
import os
import re
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LSTM = os.path.join(ROOT, "eigen-lstm_amd", "lstm")


def _text(n=3000, seed=11):
    rs = np.random.RandomState(seed)
    words = [bytes(rs.randint(97, 123, size=rs.randint(2, 8)).astype(np.uint8)) for _ in range(60)]
    out = b" ".join(words[i] for i in rs.randint(0, 60, size=n))
    return np.frombuffer(out[:n], dtype=np.uint8).copy()


def test_cli_epoch_report_matches_oracle(tmp_path, oracle32):
    N, S, B, lr, windows = 32, 8, 4, 0.1, 60
    text = _text()
    f = tmp_path / "corpus.txt"
    text.tofile(f)
    out = subprocess.run([LSTM, str(f), str(N), str(S), str(B), str(lr), "--epochs", "1", "--windows", str(windows),
                          "--seed", "1", "--sample", "50", "--save", str(tmp_path / "ck")],
                         capture_output=True, text=True, errors="replace", timeout=120)
    assert out.returncode == 0, out.stderr
    assert f"Read {len(text)} bytes ({f})" in out.stdout                        # R/lstm.cc:398
    m = re.search(r"Epoch 1/1, t = ([\d.]+) s, est GFLOP/s = ([\d.]+), avg loss = ([\d.]+) bits/char", out.stdout)
    assert m, out.stdout                                                          # R/lstm.cc:284-291
    assert "************ Generated text |" in out.stdout and "| Generated text END ************" in out.stdout
    tr = oracle32.trainer(text, N, S, B, lr=lr, seed=1)
    tr.epoch_reset()
    epoch_loss = sum(tr.window() for _ in range(windows))
    want = epoch_loss / (S * (windows + S))
    assert abs(float(m.group(3)) - want) <= 2e-3, (m.group(3), want)
    # checkpoint in the reference's text format (OV/lstm_eigen_class_CUDA/lstm.h:83-101): 5 files, row per line
    W = np.loadtxt(tmp_path / "ck_W.txt", ndmin=2)
    assert W.shape == (4 * N, 256)
    by = np.loadtxt(tmp_path / "ck_by.txt", ndmin=2)
    assert by.shape == (256, 1)
    # ... and it loads back: evaluating from the checkpoint reproduces the evaluator's number
    out2 = subprocess.run([LSTM, str(f), str(N), str(S), str(B), "0.0", "--epochs", "1", "--windows", "1", "--load",
                           str(tmp_path / "ck"), "--eval-file", str(f), "--sample", "0"],
                          capture_output=True, text=True, errors="replace", timeout=120)
    assert out2.returncode == 0, out2.stderr
    m2 = re.search(r"Test error: ([\d.]+) bits/char", out2.stdout)
    assert m2, out2.stdout
    from oracle_lib import Oracle
    P = np.concatenate([np.loadtxt(tmp_path / f"ck_{k}.txt", ndmin=2).astype(np.float32).flatten(order="F")
                        for k in ("W", "U", "b", "Why", "by")])
    assert abs(float(m2.group(1)) - oracle32.eval_bits(N, 256, P, text)) <= 1e-3


def test_cli_forget_bias_variant(tmp_path, oracle32):
    """--forget-bias 1: the later variants' b[2N:3N] = 1 at initialisation (OV/lstm_eigen_class_batch/lstm.cc:81), everything
    else as the root file; the epoch report follows the oracle's trainer started from the same parameters."""
    N, S, B, lr, windows = 32, 8, 4, 0.01, 40
    text = _text(seed=13)
    f = tmp_path / "corpus.txt"
    text.tofile(f)
    out = subprocess.run([LSTM, str(f), str(N), str(S), str(B), str(lr), "--epochs", "1", "--windows", str(windows), "--seed", "5",
                          "--sample", "0", "--forget-bias", "1", "--save", str(tmp_path / "fb")],
                         capture_output=True, text=True, errors="replace", timeout=120)
    assert out.returncode == 0, out.stderr
    m = re.search(r"avg loss = ([\d.]+) bits/char", out.stdout)
    assert m, out.stdout
    tr = oracle32.trainer(text, N, S, B, lr=lr, seed=5)
    off_b = 4 * N * 256 + 4 * N * N
    assert not tr.params[off_b:off_b + 4 * N].any()
    tr.params[off_b + 2 * N:off_b + 3 * N] = 1.0                                   # the f-gate rows (gate order i, o, f, u)
    tr.epoch_reset()
    want = sum(tr.window() for _ in range(windows)) / (S * (windows + S))
    assert abs(float(m.group(1)) - want) <= 2e-3, (m.group(1), want)
    # with the bias at 0 the same run reports a different loss: the flag reached the device
    tr0 = oracle32.trainer(text, N, S, B, lr=lr, seed=5)
    tr0.epoch_reset()
    other = sum(tr0.window() for _ in range(windows)) / (S * (windows + S))
    assert abs(other - want) > 2e-2
    b = np.loadtxt(tmp_path / "fb_b.txt", ndmin=2)[:, 0]
    assert np.all(np.abs(b[2 * N:3 * N] - 1.0) < 0.6) and np.all(np.abs(np.delete(b, np.s_[2 * N:3 * N])) < 0.6)


def test_cli_rejects_bad_arguments():
    out = subprocess.run([LSTM, "/nonexistent.txt", "24", "5", "2", "0.1"], capture_output=True, text=True, timeout=60)
    assert out.returncode != 0
    assert "fopen error" in out.stdout or "too short" in out.stderr           # R/lstm.cc:416 wording


def test_cli_results_log_holdout_and_resume(tmp_path, oracle32):
    """--test-percent / --log (the reference's results log, OV/lstm_eigen_class_CUDA/lstm.cc:76-86,186-236) and a
    resumable checkpoint: parameters + Adagrad memory + stream cursors."""
    N, S, B, lr = 32, 8, 4, 0.05
    text = _text(4000, seed=12)
    f = tmp_path / "corpus.txt"
    text.tofile(f)
    log, ck = tmp_path / "run", tmp_path / "ck"
    out = subprocess.run([LSTM, str(f), str(N), str(S), str(B), str(lr), "--epochs", "2", "--windows", "25", "--seed", "3",
                          "--sample", "0", "--test-percent", "5", "--log", str(log), "--save", str(ck), "--quiet"],
                         capture_output=True, text=True, errors="replace", timeout=120)
    assert out.returncode == 0, out.stderr + out.stdout
    cut = 95 * (len(text) // 100)
    assert f"Train set size: {cut}, Test set size: {len(text) - cut}, Total: {len(text)}" in out.stdout
    rows = np.loadtxt(str(log) + ".txt", ndmin=2)
    assert rows.shape == (2, 5) and list(rows[:, 0]) == [0.0, 1.0]              # one row per epoch end
    # the last row's test error is the held-out bits/char of the parameters saved beside it
    P = np.concatenate([np.loadtxt(str(log) + f"_{k}.txt", ndmin=2).astype(np.float32).flatten(order="F")
                        for k in ("W", "U", "b", "Why", "by")])
    assert abs(rows[1, 3] - oracle32.eval_bits(N, 256, P, text[cut:])) <= 2e-3
    assert os.path.getsize(str(log) + "_sample.txt") == 5000
    # resumable checkpoint: memory is non-negative and non-zero, cursors advanced by the windows done
    mem = np.loadtxt(str(ck) + "_mem_U.txt", ndmin=2)
    assert mem.shape == (4 * N, N) and mem.min() >= 0 and mem.max() > 0
    cur = np.loadtxt(str(ck) + "_cursors.txt", dtype=np.int64)
    start = S + (np.arange(B) * (cut - S)) // B
    assert list(cur) == list(S + (start - S + 50) % (cut - S))
    out2 = subprocess.run([LSTM, str(f), str(N), str(S), str(B), str(lr), "--epochs", "1", "--windows", "10", "--sample", "0",
                           "--test-percent", "5", "--load", str(ck), "--save", str(ck), "--quiet"],
                          capture_output=True, text=True, errors="replace", timeout=120)
    assert out2.returncode == 0, out2.stderr + out2.stdout
    assert "Loaded Adagrad memory" in out2.stdout and "Loaded stream cursors" in out2.stdout
    cur2 = np.loadtxt(str(ck) + "_cursors.txt", dtype=np.int64)
    assert list(cur2) == list(S + (cur - S + 10) % (cut - S))
    mem2 = np.loadtxt(str(ck) + "_mem_U.txt", ndmin=2)
    assert np.all(mem2 >= mem * (1 - 1e-6))                                    # Adagrad memory only grows
