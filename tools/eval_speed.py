import sys, os, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "eigen-lstm_amd"))
import lstm_hip
N = 512
text = np.random.RandomState(0).randint(32, 127, size=20000).astype(np.uint8)
for flags, name in ((0, "persistent chunks"), (lstm_hip.STEP_KERNELS, "single-workgroup kernel")):
    L = lstm_hip.Lstm(N, 4, 8, flags=flags)
    L.set_params(lstm_hip.init_params(lstm_hip.MT19937Normal(1), N) * 5)
    L.eval_bits(text[:300])
    t0 = time.perf_counter(); b = L.eval_bits(text); dt = time.perf_counter() - t0
    print(f"{name:26s} {b:.5f} bits/char  {dt / (len(text) - 1) * 1e6:7.2f} us/char")
    L.close()
