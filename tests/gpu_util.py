"""Shared helpers for the -m gpu parity tests: build matching oracle / HIP inputs, compare tensors."""
import numpy as np

from oracle_lib import split_params


def random_case(N, S, B, seed, scale=0.08, M=256, empty=()):
    rs = np.random.RandomState(seed)
    n = 4 * N * M + 4 * N * N + 4 * N + M * N + M
    P = (rs.randn(n) * scale).astype(np.float32)
    xi = rs.randint(0, M, size=(S, B)).astype(np.int32)
    ti = rs.randint(0, M, size=(S, B)).astype(np.int32)
    for (t, b) in empty:
        xi[t, b] = -1
        ti[t, b] = -1
    h0 = (rs.randn(B, N) * 0.1).astype(np.float32)
    c0 = (rs.randn(B, N) * 0.1).astype(np.float32)
    return P, xi, ti, h0, c0


def max_rel(a, b):
    """max |a-b| relative to the tensor's own scale (max |b|)."""
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def grads_report(d_hip, d_ref, N, M=256):
    a, b = split_params(d_hip, N, M), split_params(d_ref, N, M)
    return {k: max_rel(a[k], b[k]) for k in a}
