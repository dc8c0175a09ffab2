"""CPU-side checks of the drop-in boundary: the C-ABI library loads, exports every symbol that
include/lstm_hip.h declares, and refuses to run without a gfx950 device (no CPU fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    text = open(os.path.join(ROOT, "include", "lstm_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(lstm_hip_[a-z_0-9]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    import lstm_hip
    lib = lstm_hip.load_library()
    declared = _header_symbols()
    assert len(declared) >= 30
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/lstm_hip.h but not exported"
    assert sorted(lstm_hip.SYMBOLS) == declared, "lstm_hip.SYMBOLS out of sync with the header"


def test_param_count_matches_reference_shapes():
    import lstm_hip
    # W 4N x M, U 4N x N, b 4N, Why M x N, by M   (R/lstm.cc:65-70)
    for N in (16, 128, 512, 1024):
        assert lstm_hip.param_count(N) == 4 * N * 256 + 4 * N * N + 4 * N + 256 * N + 256
    assert lstm_hip.param_count(512) == 1706240  # the all-reduce payload at N=512


def test_create_rejects_bad_shapes_and_missing_gpu():
    import lstm_hip
    lib = lstm_hip.load_library()
    h = C.c_void_p()
    for N, M, S, B in ((24, 256, 5, 1), (32, 128, 5, 1), (32, 256, 1, 1), (32, 256, 5, 0)):
        cfg = lstm_hip._Config(N, M, S, B, 0, 0)
        rc = lib.lstm_hip_create(C.byref(cfg), C.byref(h))
        assert rc == -1 and not h.value, (N, M, S, B, rc)
        assert lib.lstm_hip_last_error()
    try:
        import torch
        has_gpu = torch.cuda.device_count() > 0
    except Exception:
        has_gpu = False
    if not has_gpu:
        with pytest.raises(lstm_hip.LstmHipError):
            lstm_hip.Lstm(32, 5, 2)  # must fail loudly, never fall back to a CPU path


def test_host_rng_and_cursors_match_the_oracle_spec(oracle32):
    """The product's seeded init (lstm_hip.MT19937Normal/init_params/initial_cursors) and the
    oracle's (ref_rng_*, ref_init_params, trainer cursors) are two implementations of one spec."""
    import lstm_hip
    r = oracle32.rng(123)
    P0 = oracle32.init_params(r, 32)
    h0 = oracle32.randn(r, 32, 3, 0.0, 0.1)
    g = lstm_hip.MT19937Normal(123)
    P1 = lstm_hip.init_params(g, 32)
    h1 = g.randn(32, 3, 0.0, 0.1)
    assert np.array_equal(P0, P1)
    assert np.array_equal(np.asarray(h0).T, h1)
    text = np.arange(1000, dtype=np.uint8)
    tr = oracle32.trainer(text, 16, 7, 5, seed=1)
    pos = lstm_hip.initial_cursors(len(text), 7, 5)
    for _ in range(3):
        tr.slide()
    # after 3 slides the newest target of stream b is text[pos[b] + 2]
    assert [int(v) for v in tr.ti[6]] == [int(text[int(p) + 2]) for p in pos]


def test_forget_bias_init_variant():
    """init_params(forget_bias=1): b[2N:3N] = 1 (OV/lstm_eigen_class_batch/lstm.cc:81; gate rows are i, o, f, u), every other
    value and the random stream exactly as with the root file's b = 0."""
    import lstm_hip
    N = 48
    P0 = lstm_hip.init_params(lstm_hip.MT19937Normal(9), N)
    g1 = lstm_hip.MT19937Normal(9)
    P1 = lstm_hip.init_params(g1, N, forget_bias=1.0)
    off_b = 4 * N * 256 + 4 * N * N
    assert np.all(P1[off_b + 2 * N:off_b + 3 * N] == 1.0) and not P0[off_b:off_b + 4 * N].any()
    P1[off_b + 2 * N:off_b + 3 * N] = 0.0
    assert np.array_equal(P0, P1)
    g0 = lstm_hip.MT19937Normal(9)
    lstm_hip.init_params(g0, N)
    assert np.array_equal(g0.randn(4, 2, 0.0, 1.0), g1.randn(4, 2, 0.0, 1.0))    # the stream is where it would have been


def _reference_reader_rows(path):
    """The row count the reference's readMatrix would reach (OV/lstm_eigen_class_CUDA/io.h:36-74): it calls getline
    until eof is set, counting every line -- including the empty one a trailing newline produces."""
    data = open(path, "rb").read().decode()
    return len(data.split("\n"))


@pytest.mark.parametrize("name,rows,cols", [("ref_saved_test4_b.txt", 64, 1), ("ref_saved_test4_Why.txt", 256, 16)])
def test_checkpoint_text_layout_is_the_references(tmp_path, name, rows, cols):
    """Parameters::save_to_disk files written by the host program must be loadable by the reference (and vice versa):
    same layout as Eigen's operator<< (matrix_io.h).  A file the REFERENCE saved (tests/golden/ref_saved_*, data) is read
    by the host program's reader and written back by its writer: byte-identical, and the reference's eof/getline reader
    loop sees exactly `rows` rows (no trailing newline)."""
    import subprocess
    exe = tmp_path / "host_io_roundtrip"
    subprocess.check_call(["g++", "-O1", "-std=c++17", os.path.join(ROOT, "tests", "host_io_roundtrip.cc"), "-o", str(exe)])
    src = os.path.join(ROOT, "tests", "golden", name)
    out = tmp_path / "out.txt"
    dims = subprocess.check_output([str(exe), src, str(out)], text=True).split()
    assert [int(d) for d in dims] == [rows, cols]
    assert open(src, "rb").read() == open(out, "rb").read()
    assert _reference_reader_rows(out) == rows
    # 9 digits round-trip float32 exactly (the resumable checkpoint's Adagrad memory)
    out9, back = tmp_path / "out9.txt", tmp_path / "back.txt"
    subprocess.check_call([str(exe), src, str(out9), "9"])
    subprocess.check_call([str(exe), str(out9), str(back)])
    assert open(src, "rb").read() == open(back, "rb").read()
