"""Diagnostic: what the data-parallel code path costs before any bytes move -- the same window with and without a
1-rank RCCL communicator (folds as separate launches, the split all-reduce on two streams).  Headline shape.
  python tools/comm_overhead_probe.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "eigen-lstm_amd"))
sys.path.insert(0, ROOT)
import lstm_hip  # noqa: E402
from bench import synthetic_text  # noqa: E402

N, S, B = 512, 100, 64
text = synthetic_text(1_000_000, seed=0)
hs = []
for with_comm in (False, True):
    L = lstm_hip.Lstm(N, S, B)
    L.set_params(lstm_hip.init_params(lstm_hip.MT19937Normal(1), N))
    L.set_text(text)
    L.set_cursors(lstm_hip.initial_cursors(len(text), S, B))
    if with_comm:
        L.comm_init(lstm_hip.comm_unique_id(), 1, 0)
        L.set_global_batch(B)
    L.train_windows(300, 0.01, want_losses=False)
    hs.append(L)
for rep in range(3):
    for name, L in zip(("no communicator", "1-rank communicator"), hs):
        _, dev_ms = L.train_windows(300, 0.01, want_losses=True, want_time=True)
        print(f"{name:22s}: {dev_ms / 300 * 1e3:.1f} us per window")
L = hs[1]
L.reset_kernel_stats(); L.set_profiling(True); L.train_windows(4, 0.01, want_losses=False); L.set_profiling(False)
print({k: round(ms / c * 1e3, 1) for k, (c, ms) in L.kernel_stats().items() if c})
for L in hs:
    L.close()
