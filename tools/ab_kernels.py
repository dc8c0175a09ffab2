"""A/B of library variants in ONE process (cdna_hip_programming.md 5.4 rule 24): one handle per variant, created under
that variant's environment (the switches below are read per handle at create), timed in interleaved rounds with the
library's own HIP-event profiling; prints median / min microseconds per kernel.

  python tools/ab_kernels.py [N S B] [--rounds R] [--windows W] name=ENV1:val,ENV2:val ...

e.g.  python tools/ab_kernels.py halves=LSTM_HIP_BWD_HALVES:7 one=LSTM_HIP_BWD_HALVES:0 poll0=LSTM_HIP_FWD_POLL:0
"""
import os
import statistics
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "eigen-lstm_amd"))
sys.path.insert(0, ROOT)
import lstm_hip  # noqa: E402
from bench import synthetic_text  # noqa: E402


def main():
    args = sys.argv[1:]
    shape = [512, 100, 64]
    rounds, windows = 5, 4
    variants = []
    i = 0
    nums = []
    while i < len(args):
        a = args[i]
        if a == "--rounds":
            rounds = int(args[i + 1]); i += 2; continue
        if a == "--windows":
            windows = int(args[i + 1]); i += 2; continue
        if "=" in a:
            name, spec = a.split("=", 1)
            env = dict(kv.split(":", 1) for kv in spec.split(",") if kv)
            variants.append((name, env))
        else:
            nums.append(int(a))
        i += 1
    if len(nums) == 3:
        shape = nums
    N, S, B = shape
    flags = int(os.environ.get("AB_FLAGS", "0"))
    text = synthetic_text(1_000_000, seed=0)
    handles = []
    for name, env in variants:
        saved = {k: os.environ.get(k) for k in env}
        os.environ.update(env)
        L = lstm_hip.Lstm(N, S, B, flags=flags)
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
        rng = lstm_hip.MT19937Normal(1)
        L.set_params(lstm_hip.init_params(rng, N))
        L.set_state(1, rng.randn(N, B, 0.0, 0.1), rng.randn(N, B, 0.0, 0.1))
        L.set_text(text)
        L.set_cursors(lstm_hip.initial_cursors(len(text), S, B))
        L.train_windows(S + 2, 0.0, want_losses=False)  # fill the window; lr 0 keeps every variant on the same weights
        handles.append((name, L))
    res = {name: {} for name, _ in handles}
    wall = {name: [] for name, _ in handles}
    for r in range(rounds):
        for name, L in handles:
            L.reset_kernel_stats()
            L.set_profiling(True)
            L.train_windows(windows, 0.001, want_losses=False)
            L.set_profiling(False)
            for k, (calls, ms) in L.kernel_stats().items():
                if calls:
                    res[name].setdefault(k, []).append(ms / calls * 1e3)
            _, dev_ms = L.train_windows(20, 0.001, want_losses=True, want_time=True)
            wall[name].append(dev_ms / 20 * 1e3)
    kernels = sorted({k for v in res.values() for k in v})
    print(f"shape N={N} S={S} B={B}; {rounds} rounds x {windows} windows; median (min) us per launch")
    print("%-18s" % "kernel" + "".join("%22s" % n for n, _ in handles))
    for k in kernels:
        row = "%-18s" % k
        for n, _ in handles:
            v = res[n].get(k)
            row += "%22s" % ("%.1f (%.1f)" % (statistics.median(v), min(v)) if v else "-")
        print(row)
    row = "%-18s" % "window (unprofiled)"
    for n, _ in handles:
        row += "%22s" % ("%.1f (%.1f)" % (statistics.median(wall[n]), min(wall[n])))
    print(row)
    # all variants saw the same weights and text: their parameters must still agree closely
    ref = handles[0][1].get_params()
    for n, L in handles[1:]:
        p = L.get_params()
        print(f"max |param diff| {n} vs {handles[0][0]}: {np.abs(p - ref).max():.3e}")
    for _, L in handles:
        L.close()


if __name__ == "__main__":
    main()
