"""-m gpu: one window (forward, loss, BPTT) against the oracle over a grid of awkward shapes -- batches that are not a multiple
of a column group or of a half, one and two timesteps, an all-zero input column in the middle -- for the hidden sizes whose
recurrences change form with the batch: one pinned group, several pinned groups with one half per workgroup, two halves per
workgroup, two launches over column ranges (fp32: hidden 256 / 512; bf16 path: 256 / 512 / 1024).  Tolerances as in
test_hip_parity.py."""
import itertools

import numpy as np
import pytest

import gpu_util as gu

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def oracle_mt():
    from oracle_lib import Oracle
    return Oracle("f32_omp")  # the same restatement built with -fopenmp: the grid below is ~25 GFLOP of oracle work


def _window(lstm_hip, N, S, B, P, xi, ti, h0, c0, flags=0):
    L = lstm_hip.Lstm(N, S, B, flags=flags)
    L.set_params(P)
    L.set_state(0, h0, c0)
    L.set_window(xi, ti)
    L.forward()
    loss = L.loss()
    h, _ = L.get_state(S - 1)
    L.backward()
    g = L.get_grads()
    L.close()
    return loss, h, g


@pytest.mark.parametrize("N", [256, 512])
def test_fp32_forms_over_awkward_batches(N, oracle_mt):
    oracle32 = oracle_mt
    import lstm_hip
    for B, S in itertools.product((1, 2, 5, 7, 9, 13, 33, 37, 65, 70, 100, 129), (2, 3, 5)):
        P, xi, ti, h0, c0 = gu.random_case(N, S, B, seed=N + 7 * B + S, scale=0.05, empty=((1, B // 2),) if S > 2 else ())
        fw = oracle32.forward(N, 256, S, B, P, xi, ti, h0, c0)
        dref = oracle32.backward(N, 256, S, B, P, xi, ti, fw)
        loss, h, g = _window(lstm_hip, N, S, B, P, xi, ti, h0, c0)
        assert gu.max_rel(h, fw["h"][S - 1]) <= 2e-5, (N, B, S)
        assert abs(loss - fw["loss_bits"]) <= 2e-5 * (S - 1), (N, B, S)
        rep = gu.grads_report(g, dref, N)
        assert max(rep.values()) <= 2e-4, (N, B, S, rep)


@pytest.mark.parametrize("N", [256, 512, 1024])
def test_bf16_forms_over_awkward_batches(N, oracle_mt):
    oracle32 = oracle_mt
    import lstm_hip
    for B, S in itertools.product((8, 24, 40, 72, 136), (2, 3, 4)):
        P, xi, ti, h0, c0 = gu.random_case(N, S, B, seed=N + 7 * B + S, scale=0.05)
        oracle32.set_bf16_recurrence(True)
        oracle32.set_bf16_products(True)
        try:
            fw = oracle32.forward(N, 256, S, B, P, xi, ti, h0, c0)
            dref = oracle32.backward(N, 256, S, B, P, xi, ti, fw)
        finally:
            oracle32.set_bf16_recurrence(False)
            oracle32.set_bf16_products(False)
        loss, h, g = _window(lstm_hip, N, S, B, P, xi, ti, h0, c0, flags=lstm_hip.BF16_RECURRENCE)
        assert gu.max_rel(h, fw["h"][S - 1]) <= 2e-3, (N, B, S)
        rep = gu.grads_report(g, dref, N)
        assert max(rep.values()) <= 1e-2, (N, B, S, rep)
