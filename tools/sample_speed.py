import sys, os, time
import numpy as np
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, os.path.join(ROOT, "eigen-lstm_amd"))
import lstm_hip
for N in (128, 512):
    L = lstm_hip.Lstm(N, 4, 1)
    L.set_params(lstm_hip.init_params(lstm_hip.MT19937Normal(1), N))
    u = np.random.RandomState(0).rand(5000)
    h0 = np.zeros(N, np.float32); c0 = np.zeros(N, np.float32)
    L.sample(h0, c0, u[:100])
    t0 = time.perf_counter(); out = L.sample(h0, c0, u); dt = time.perf_counter() - t0
    print(N, "us/char", dt / 5000 * 1e6)
    L.close()
