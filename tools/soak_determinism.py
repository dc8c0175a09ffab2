"""Soak: two handles, same seed and text, W windows each at the headline shape; the parameters must come out bit-identical
(every kernel of the path is deterministic, so a difference would mean a race in a hand-off).
  python tools/soak_determinism.py [windows [N S B [flags]]]      (flags 128: the bf16 recurrences)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "eigen-lstm_amd"))
sys.path.insert(0, ROOT)
import lstm_hip  # noqa: E402
from bench import synthetic_text  # noqa: E402

W = int(sys.argv[1]) if len(sys.argv) > 1 else 12000
N, S, B = (int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (512, 100, 64)
FLAGS = int(sys.argv[5]) if len(sys.argv) > 5 else 0
text = synthetic_text(1_000_000, seed=0)
out = []
for rep in range(2):
    L = lstm_hip.Lstm(N, S, B, flags=FLAGS)
    L.set_params(lstm_hip.init_params(lstm_hip.MT19937Normal(1), N))
    L.set_text(text)
    L.set_cursors(lstm_hip.initial_cursors(len(text), S, B))
    losses = []
    for _ in range(W // 2000):
        losses.append(L.train_windows(2000, 0.005))
        print(f"handle {rep}: {len(losses) * 2000} windows, loss {losses[-1][-1]:.4f}", flush=True)
    out.append((np.concatenate(losses), L.get_params()))
    L.close()
same_l = np.array_equal(out[0][0], out[1][0])
same_p = np.array_equal(out[0][1], out[1][1])
print("losses finite:", bool(np.all(np.isfinite(out[0][0]))), " losses identical:", same_l, " parameters identical:", same_p)
sys.exit(0 if same_l and same_p else 1)
