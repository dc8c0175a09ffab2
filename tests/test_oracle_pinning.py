"""Pins the CPU oracle (oracle/lstm_ref.c) before anything is compared against it.

1. Known-answer fixtures A and B: weights the reference saved + the bits/char its own log
   recorded for them (tests/golden/make_fixtures.py).  Tolerance 1e-4 bits/char
   (the weight text carries 6 significant digits).
2. Finite-difference gradient check with the reference's own thresholds
   (OV/lstm_eigen_class/lstm.cc:250-304: rel = |a-n|/|a+n|, max <= 1e-1, mean <= 1e-3), fp64.
3. An independent torch-autograd model of the same recurrence (forward values and gradients).
4. The index-form window slide against a dense one-hot, line-by-line numpy restatement of
   R/lstm.cc:155-170.
"""
import os

import numpy as np
import pytest

from oracle_lib import split_params

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("name", ["A", "B"])
def test_known_answer_fixture(name, oracle32, oracle64):
    fx = np.load(os.path.join(GOLD, f"fixture_{name}.npz"))
    N, M = int(fx["N"]), int(fx["M"])
    for orc in (oracle32, oracle64):
        bits = orc.eval_bits(N, M, fx["params"], fx["text"])
        assert abs(bits - float(fx["expected_bits"])) <= 1e-4, (name, orc.kind, bits)


def chained_windows_bits(forward_fn, N, text, S):
    """The reference's test() (OV/lstm_eigen_class_CUDA/lstm.cc:661-720) expressed with the WINDOW operator instead of
    the one-stream loop: the text is cut into windows of S-1 steps (B = 1), h = c = 0 at the start (reset_std = 0,
    lstm.cc:45,676-677), the last column of a window is the carry of the next; past the end of the text the columns are
    empty (index -1: no input, no loss).  forward_fn(xi, ti, h0, c0) -> (window loss in bits, h_carry, c_carry) where
    the carry is the state after the window's last real character.  Returns bits/char."""
    h = np.zeros((1, N), np.float32)
    c = np.zeros((1, N), np.float32)
    total = 0.0
    for pos in range(0, len(text) - 1, S - 1):
        steps = min(S - 1, len(text) - 1 - pos)
        xi = np.full((S, 1), -1, np.int32)
        ti = np.full((S, 1), -1, np.int32)
        xi[1:steps + 1, 0] = text[pos:pos + steps]
        ti[1:steps + 1, 0] = text[pos + 1:pos + steps + 1]
        bits, h, c = forward_fn(xi, ti, h, c, steps)
        total += bits
    return total / (len(text) - 1)


@pytest.mark.parametrize("name,S", [("A", 26), ("B", 26), ("A", 101), ("B", 8)])
def test_known_answer_fixture_through_the_window_forward(name, S, oracle32):
    """Fixtures A/B through ref_forward -- the function every window parity test compares the HIP path with -- and not
    only through the separate one-stream loop ref_eval_bits: same logged bits/char, same 1e-4 tolerance."""
    fx = np.load(os.path.join(GOLD, f"fixture_{name}.npz"))
    N, M = int(fx["N"]), int(fx["M"])

    def fwd(xi, ti, h0, c0, steps):
        fw = oracle32.forward(N, M, S, 1, fx["params"], xi, ti, h0, c0)
        return fw["loss_bits"], fw["h"][steps].copy(), fw["c"][steps].copy()

    bits = chained_windows_bits(fwd, N, fx["text"], S)
    assert abs(bits - float(fx["expected_bits"])) <= 1e-4, (name, S, bits)
    # and the two oracle entry points agree with each other far more tightly than with the 6-digit weight text
    assert abs(bits - oracle32.eval_bits(N, M, fx["params"], fx["text"])) <= 2e-6


def _random_case(orc, N, S, B, seed, M=256, scale=0.3):
    rs = np.random.RandomState(seed)
    P = (rs.randn(orc.param_count(N, M)) * scale).astype(orc.np_t)
    xi = rs.randint(0, M, size=(S, B)).astype(np.int32)
    ti = rs.randint(0, M, size=(S, B)).astype(np.int32)
    h0 = (rs.randn(B, N) * 0.1).astype(orc.np_t)
    c0 = (rs.randn(B, N) * 0.1).astype(orc.np_t)
    return P, xi, ti, h0, c0


def test_gradient_check_reference_thresholds(oracle64):
    N, M, S, B = 8, 256, 5, 3
    P, xi, ti, h0, c0 = _random_case(oracle64, N, S, B, seed=3)
    xi[1, 0] = -1  # one empty input column (window not yet full)
    fw = oracle64.forward(N, M, S, B, P, xi, ti, h0, c0)
    dP = oracle64.backward(N, M, S, B, P, xi, ti, fw)
    rs = np.random.RandomState(0)
    o = 0
    for name, r, c in (("W", 4 * N, M), ("U", 4 * N, N), ("b", 4 * N, 1), ("Why", M, N), ("by", M, 1)):
        idx = o + rs.choice(r * c, size=min(r * c, 200), replace=False)
        num = oracle64.numgrad(N, M, S, B, P, xi, ti, h0, c0, idx)
        ana = dP[idx]
        den = np.abs(ana + num)
        rel = np.where(den > 0, np.abs(ana - num) / np.where(den > 0, den, 1), 0.0)
        assert rel.max() <= 1e-1 and rel.mean() <= 1e-3, (name, rel.max(), rel.mean())
        # far tighter than the reference asks, since this is fp64 end to end (absolute, because
        # the relative form blows up on structurally-zero entries such as unused W columns)
        assert np.abs(ana - num).max() <= 1e-7 * max(1.0, np.abs(ana).max()), (name, np.abs(ana - num).max())
        o += r * c


def _torch_model(P, N, M, S, B, xi, ti, h0, c0):
    """Same recurrence written with torch ops + autograd; nothing shared with the oracle."""
    import torch
    torch.set_num_threads(1)
    Pt = torch.tensor(P, dtype=torch.float64, requires_grad=True)
    o = 0
    mats = {}
    for name, r, c in (("W", 4 * N, M), ("U", 4 * N, N), ("b", 4 * N, 1), ("Why", M, N), ("by", M, 1)):
        mats[name] = Pt[o:o + r * c].reshape(c, r).T  # column-major
        o += r * c
    h = torch.tensor(np.asarray(h0, dtype=np.float64).T)  # N x B
    c = torch.tensor(np.asarray(c0, dtype=np.float64).T)
    nats = torch.zeros((), dtype=torch.float64)
    bits = 0.0
    hs, cs = [], []
    for t in range(1, S):
        x = torch.zeros(M, B, dtype=torch.float64)
        for b in range(B):
            if xi[t, b] >= 0:
                x[xi[t, b], b] = 1.0
        g = mats["W"] @ x + mats["U"] @ h + mats["b"]
        i, o_, f, u = g[:N], g[N:2 * N], g[2 * N:3 * N], g[3 * N:]
        i, o_, f, u = torch.sigmoid(i), torch.sigmoid(o_), torch.sigmoid(f), torch.tanh(u)
        c = torch.tanh(i * u + f * c)
        h = o_ * c
        y = mats["Why"] @ h + mats["by"]
        p = torch.exp(y) / torch.exp(y).sum(0, keepdim=True)
        for b in range(B):
            if ti[t, b] >= 0:
                nats = nats - torch.log(p[ti[t, b], b])
                bits += float(-torch.log2(p[ti[t, b], b]).detach()) / B
        hs.append(h.detach().numpy().T.copy())
        cs.append(c.detach().numpy().T.copy())
    nats.backward()
    return bits, float(nats), Pt.grad.numpy(), hs, cs


def test_against_torch_autograd(oracle64, oracle32):
    N, M, S, B = 12, 256, 6, 4
    P, xi, ti, h0, c0 = _random_case(oracle64, N, S, B, seed=11)
    xi[1, 2] = -1
    bits, nats, grad, hs, cs = _torch_model(P, N, M, S, B, xi, ti, h0, c0)
    fw = oracle64.forward(N, M, S, B, P, xi, ti, h0, c0)
    dP = oracle64.backward(N, M, S, B, P, xi, ti, fw)
    assert abs(fw["loss_nats"] - nats) <= 1e-9 * max(1, abs(nats))
    assert abs(fw["loss_bits"] - bits) <= 1e-9 * max(1, abs(bits))
    for t in range(1, S):
        np.testing.assert_allclose(fw["h"][t], hs[t - 1], rtol=1e-10, atol=1e-12)
        np.testing.assert_allclose(fw["c"][t], cs[t - 1], rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(dP, grad, rtol=1e-8, atol=1e-11)
    # the fp32 build of the same source agrees to fp32 accuracy
    fw32 = oracle32.forward(N, M, S, B, P, xi, ti, h0, c0)
    dP32 = oracle32.backward(N, M, S, B, P.astype(np.float32), xi, ti, fw32)
    assert abs(fw32["loss_bits"] - bits) <= 1e-4
    np.testing.assert_allclose(dP32, grad, rtol=2e-3, atol=2e-5)


def _dense_root_iteration(state, event, S, M):
    """R/lstm.cc:157-170 on dense one-hot columns (B = 1), numpy."""
    x, target, h, c = state
    for s in range(1, S):
        x[:, s - 1] = x[:, s]
        target[:, s - 1] = target[:, s]
        h[:, s - 1] = h[:, s]
        c[:, s - 1] = c[:, s]
    target[:, S - 1] = np.eye(M, dtype=np.float32)[event]
    x[:, S - 1] = target[:, S - 2]


def test_window_slide_matches_dense_onehot_form(oracle32):
    N, M, S, B = 4, 256, 5, 1
    text = (np.arange(40) * 7 % 251).astype(np.uint8)
    tr = oracle32.trainer(text, N, S, B, seed=5)
    tr.epoch_reset()
    x = np.zeros((M, S), np.float32)
    target = np.zeros((M, S), np.float32)
    h = tr.h[:, 0, :].T.copy()
    c = tr.c[:, 0, :].T.copy()
    pos = S
    for it in range(3 * len(text)):
        _dense_root_iteration((x, target, h, c), int(text[pos]), S, M)
        pos += 1
        if pos >= len(text):
            pos = S
        tr.slide()
        for s in range(S):
            for idx, dense in ((tr.xi[s, 0], x[:, s]), (tr.ti[s, 0], target[:, s])):
                want = np.zeros(M, np.float32)
                if idx >= 0:
                    want[idx] = 1
                assert np.array_equal(want, dense), (it, s)
        np.testing.assert_array_equal(tr.h[:, 0, :].T, h)
        # keep h/c in step: pretend forward wrote nothing (both sides only shift)


def test_adagrad_first_step_and_eps(oracle32):
    # first step: m = d^2 -> p -= lr * d / sqrt(d^2 + 1e-10) ~ lr * sign(d)   (R/lstm.cc:261-272)
    P = np.zeros(4, np.float32)
    d = np.array([1e-3, -2.0, 0.0, 1e-7], np.float32)
    mem = np.zeros(4, np.float32)
    oracle32.adagrad(P, d, mem, 0.1)
    want = -np.float32(0.1) * (d / np.sqrt((mem.astype(np.float64) + 1e-10).astype(np.float32)))
    np.testing.assert_array_equal(P, want.astype(np.float32))
    assert P[2] == 0.0 and abs(P[0] + 0.1) < 1e-5


def test_rng_is_mt19937(oracle32):
    # MT19937 known answers: seed 5489 -> first output 3499211612, 10000th 4123659995
    r = oracle32.rng(5489)
    first = oracle32.rng_u32(r)
    for _ in range(9998):
        oracle32.rng_u32(r)
    assert first == 3499211612 and oracle32.rng_u32(r) == 4123659995
    r = oracle32.rng(1)
    z = oracle32.randn(r, 200, 200, 0.0, 1.0)
    assert abs(z.mean()) < 0.02 and abs(z.std() - 1) < 0.02


def test_empty_target_column_is_faithful_not_a_gradient(oracle64):
    """Before the window has filled, target columns are all-zero (OV/lstm_eigen_opt/lstm.cc:122).
    The reference then still back-propagates dy = probs - 0 (opt:270) although that column adds
    nothing to the loss (opt:246): the restatement keeps this, it is NOT the loss gradient."""
    N, M, S, B = 6, 256, 3, 2
    P, xi, ti, h0, c0 = _random_case(oracle64, N, S, B, seed=2)
    ti[2, 1] = -1
    fw = oracle64.forward(N, M, S, B, P, xi, ti, h0, c0)
    dP = split_params(oracle64.backward(N, M, S, B, P, xi, ti, fw), N, M)
    dy = fw["probs"].copy()  # [S, B, M]
    for t in range(1, S):
        for b in range(B):
            if ti[t, b] >= 0:
                dy[t, b, ti[t, b]] -= 1.0
    np.testing.assert_allclose(dP["by"][:, 0], dy[1:].sum(axis=(0, 1)), rtol=1e-12, atol=1e-14)
    assert abs(dy[2, 1].sum() - 1.0) < 1e-12  # the empty column back-propagates the whole distribution


def test_correct_implementations_drift_apart(oracle32, oracle64):
    """The control experiment behind the free-running GPU tolerance (tests/trajectory_util.py): the oracle in float64,
    and in float32 from parameters one ulp away, against the float32 oracle over 60 windows at lr = 0.1.  All agree to
    1e-3 bits per window at first; none of them keeps that up (so "1e-3 for the first 100 windows", SURVEY 8d's
    proposal, is not a property of the algorithm), while the late averages stay within 0.005 bits/char."""
    import trajectory_util as tu
    N, S, B, windows, lr = 64, 10, 20, 60, 0.1
    text = tu.printable_text(4000, seed=5)
    base, controls = tu.oracle_trajectories(oracle32, oracle64, text, N, S, B, windows, lr)
    for c in controls:
        assert np.abs(c - base)[:10].max() <= 1e-3
    per_win, late = tu.envelope(base, controls, S)
    assert sum(np.abs(c - base).max() > 1e-3 for c in controls) >= len(controls) - 2
    assert 1e-3 < per_win < 0.25 and late < 0.005, (per_win, late)
