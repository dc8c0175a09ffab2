"""How far do two CORRECT implementations of the training loop drift apart?  (test infrastructure)

The first Adagrad steps move every touched weight by +-lr*sign(d) (m = d^2, R/lstm.cc:261-272), so a rounding-level
difference in a near-zero gradient flips a weight by 2*lr and the trajectories of any two implementations that round
differently separate (hard part 2).  The free-running parity test therefore cannot use a fixed small
tolerance; it is calibrated against CONTROLS: the oracle itself run (a) in float64 from the same float32 start,
(b) in float32 with a few parameters, or all of them, moved by one ulp and (c) in float32 with every contraction summed
in descending instead of ascending index order (the reference leaves that order to Eigen / BLAS).  Every control is a
correct implementation of
OV/lstm_eigen_opt/lstm.cc:186-318; their distance from the float32 oracle is the yardstick for the HIP path.
"""
import numpy as np


def printable_text(n, seed):
    return np.random.RandomState(seed).choice(np.arange(32, 127), size=n).astype(np.uint8)


def oracle_trajectories(oracle32, oracle64, text, N, S, B, windows, lr, seed=1, n_ulp_controls=6, n_all_controls=3):
    """Returns (base, controls): window losses (bits, summed over S-1 steps) of the float32 oracle and of each control."""
    def fresh(orc):
        tr = orc.trainer(text, N, S, B, lr=lr, seed=seed)
        tr.epoch_reset()
        return tr

    tr = fresh(oracle32)
    start_p, start_h, start_c = tr.params.copy(), tr.h.copy(), tr.c.copy()
    base = np.array([tr.window() for _ in range(windows)])
    controls = []
    t64 = fresh(oracle64)  # float64 arithmetic from the float32 start
    t64.params[:] = start_p
    t64.h[:] = start_h
    t64.c[:] = start_c
    controls.append(np.array([t64.window() for _ in range(windows)]))
    for k in range(n_ulp_controls):  # float32 arithmetic, 50 parameters one ulp away
        t = fresh(oracle32)
        rs = np.random.RandomState(100 + k)
        idx = rs.choice(t.params.size, size=50, replace=False)
        p = t.params
        p[idx] = np.nextafter(p[idx], np.float32(np.inf if k % 2 else -np.inf))
        controls.append(np.array([t.window() for _ in range(windows)]))
    for k in range(n_all_controls):  # float32 arithmetic, EVERY parameter one ulp away in a random direction
        t = fresh(oracle32)
        rs = np.random.RandomState(7 + k)
        p = t.params
        p[:] = np.nextafter(p, np.where(rs.rand(p.size) < 0.5, np.float32(np.inf), np.float32(-np.inf)).astype(np.float32))
        controls.append(np.array([t.window() for _ in range(windows)]))
    # float32 arithmetic, every product of the window summed in descending index order (oracle/lstm_ref.c, ref_set_descending_sums)
    oracle32.lib.ref_set_descending_sums(1)
    try:
        t = fresh(oracle32)
        controls.append(np.array([t.window() for _ in range(windows)]))
    finally:
        oracle32.lib.ref_set_descending_sums(0)
    return base, controls


def envelope(base, controls, S, late=20):
    """per-char distances of the controls from the base trajectory: (max over controls and windows of the per-window
    distance, max over controls of the late-average distance), both in bits/char"""
    per_win = max(float(np.abs(c - base).max()) for c in controls) / (S - 1)
    late_avg = max(abs(float(c[-late:].mean() - base[-late:].mean())) for c in controls) / (S - 1)
    return per_win, late_avg
