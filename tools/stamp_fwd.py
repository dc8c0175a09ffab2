"""Diagnostic: where does a step of the forward recurrence spend its time?  (LSTM_HIP_DEBUG_STAMPS build)"""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "eigen-lstm_amd")); sys.path.insert(0, ROOT)
import lstm_hip
from bench import synthetic_text
N, S, B = 512, 100, 64
L = lstm_hip.Lstm(N, S, B, flags=lstm_hip.DEBUG_STAMPS)
L.set_params(lstm_hip.init_params(lstm_hip.MT19937Normal(1), N))
text = synthetic_text(200000)
L.set_text(text); L.set_cursors(lstm_hip.initial_cursors(len(text), S, B))
L.train_windows(20, 0.01)
st = L.debug_stamps().astype(np.float64)
print("forward:")
for wg in range(2):
    s = st[wg, 2:S - 1]
    names = ["step top -> after poll+barrier", "loads+MFMA+LDS reduce+barrier", "gates (epilogue math)", "h store + drain", "signal -> next step top"]
    d = [s[:, 1] - s[:, 0], s[:, 2] - s[:, 1], s[:, 3] - s[:, 2], s[:, 4] - s[:, 3]]
    nxt = st[wg, 3:S, 0] - st[wg, 2:S - 1, 4]
    d.append(nxt)
    tot = st[wg, 3:S, 0] - st[wg, 2:S - 1, 0]
    print(f"workgroup {wg}: cycles per step median {np.median(tot):.0f}")
    for n, v in zip(names, d):
        print(f"   {n:38s} median {np.median(v):8.0f}  p90 {np.percentile(v, 90):8.0f}")

print("backward (steps run S-1 .. 1):")
for wg in (2, 3):
    # stamps: 0 step top, 1 after poll+barrier, 2 after loads+MFMA+reduce barrier, 3 after elementwise+stage barrier, 4 after store+drain+barrier
    s = st[wg, 3:S - 2]
    names = ["step top -> after poll+barrier", "dg loads + MFMA + LDS reduce", "elementwise + staging barrier", "dg store + drain + barrier", "signal -> next step top"]
    d = [s[:, 1] - s[:, 0], s[:, 2] - s[:, 1], s[:, 3] - s[:, 2], s[:, 4] - s[:, 3]]
    d.append(st[wg, 2:S - 3, 0] - st[wg, 3:S - 2, 4])
    tot = st[wg, 2:S - 3, 0] - st[wg, 3:S - 2, 0]
    print(f"workgroup {wg - 2}: cycles per step median {np.median(tot):.0f}")
    for n, v in zip(names, d):
        print(f"   {n:38s} median {np.median(v):8.0f}  p90 {np.percentile(v, 90):8.0f}")
