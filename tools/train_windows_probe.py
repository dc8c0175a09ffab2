"""Diagnostic: how long a timed region of K windows takes right after a long leg of windows (clock / power state effects
on short measurements).  python tools/train_windows_probe.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "eigen-lstm_amd"))
sys.path.insert(0, ROOT)
import lstm_hip  # noqa: E402
from bench import synthetic_text  # noqa: E402

N, S, B = 512, 100, 64
text = synthetic_text(1_000_000, seed=0)
L = lstm_hip.Lstm(N, S, B)
L.set_params(lstm_hip.init_params(lstm_hip.MT19937Normal(1), N))
L.set_text(text)
L.set_cursors(lstm_hip.initial_cursors(len(text), S, B))
for leg, wl in ((300, True), (2300, False), (2300, True), (2300, True), (5000, False), (5000, True)):
    t0 = time.perf_counter()
    L.train_windows(leg, 0.01, want_losses=wl)
    L.synchronize()
    tl = (time.perf_counter() - t0) / leg * 1e3
    L.train_windows(5, 0.01, want_losses=False)
    L.synchronize()
    out = []
    for rep in range(3):
        t0 = time.perf_counter()
        _, dev_ms = L.train_windows(20, 0.01, want_losses=True, want_time=True)
        L.synchronize()
        out.append(f"{(time.perf_counter() - t0) / 20 * 1e3:.4f}/{dev_ms / 20:.4f}")
    print(f"losses={wl} leg of {leg} windows at {tl:.4f} ms; then 5; then 3 x 20 timed (wall/device ms per window): {' '.join(out)}")
L.close()
