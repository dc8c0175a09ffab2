"""Stand-in for eigen-lstm_amd/lstm_hip.py with no GPU behind it: lets the CPU suite drive bench.py's
multi-rank plumbing (rendezvous, id broadcast, max-over-ranks, JSON line).  Test infrastructure only;
bench.py uses it only when LSTM_BENCH_FAKE_GPU=1 and says so in its output."""
import time

import numpy as np

STEP_KERNELS = 4
FAKE = True


def comm_unique_id():
    return bytes(range(128))


class MT19937Normal:
    def __init__(self, seed):
        self.rs = np.random.RandomState(seed)

    def randn(self, rows, cols, mean, std):
        return (mean + std * self.rs.randn(cols, rows)).astype(np.float32)


def init_params(rng, N, M=256):
    return np.zeros(4 * N * M + 4 * N * N + 4 * N + M * N + M, np.float32)


class Lstm:
    def __init__(self, N, S, B, device=0, flags=0):
        self.N, self.S, self.B, self.device = N, S, B, device
        self.comm = None

    def set_params(self, p): pass
    def set_state(self, t, h, c): pass
    def set_text(self, t): pass
    def set_cursors(self, pos): assert len(pos) == self.B
    def set_window(self, xi, ti): assert xi.shape == (self.S, self.B) and ti.shape == (self.S, self.B)
    def set_global_batch(self, gb): self.gb = gb
    def comm_init(self, uid, world, rank): assert uid == bytes(range(128)); self.comm = (world, rank)
    def synchronize(self): pass
    def set_profiling(self, on): pass
    def reset_kernel_stats(self): pass
    def kernel_stats(self): return {"fwd_persistent": (3, 1.2), "bwd_persistent": (3, 1.5)}
    def close(self): pass

    def train_windows(self, count, lr, want_losses=True, want_time=False):
        time.sleep(0.001 * count)
        losses = np.full(count, 1.0) if want_losses else None
        return (losses, float(count)) if want_time else losses
