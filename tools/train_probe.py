"""Diagnostic: loss trajectory and a sample at the headline shape (synthetic word corpus)."""
import sys, os, subprocess
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "eigen-lstm_amd"))
import lstm_hip
N, S, B = 512, 100, 64
lr, warm, total = float(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "make_text.py"), "/tmp/corpus.txt", "1000000"])
text = np.fromfile("/tmp/corpus.txt", dtype=np.uint8)
L = lstm_hip.Lstm(N, S, B)
rng = lstm_hip.MT19937Normal(1)
L.set_params(lstm_hip.init_params(rng, N))
L.set_state(1, rng.randn(N, B, 0, 0.1), rng.randn(N, B, 0, 0.1))
L.set_text(text); L.set_cursors(lstm_hip.initial_cursors(len(text), S, B)); L.reset_window()
done = 0
while done < total:
    n = min(1000, total - done)
    losses = L.train_windows(n, 0.0 if done < warm else lr)
    done += n
    print(f"windows {done:6d}  mean bits/char {np.nanmean(losses) / (S - 1):.4f}  nan {int(np.isnan(losses).sum())}", flush=True)
u = np.random.RandomState(0).random_sample(200)
out, _, _ = L.sample(rng.randn(N, 1, 0, 0.1)[0], rng.randn(N, 1, 0, 0.1)[0], u)
print("sample:", bytes(out).decode("latin1"))
