"""Host-side mirror of the reference's device seam over the C ABI (include/lstm_hip.h).

The reference's driver (OV/lstm_eigen_class_CUDA/lstm.cc:99-114,156-163,273-377) owns
`cuParameters p, d, m` and a `cuLSTM<S>` and calls forward / calculate_loss / backward / cuda_adagrad
plus the copy helpers.  `Lstm` below is that set of objects behind one handle, same names and
argument meaning; arrays are numpy, column-major as Eigen's (an `N x B` matrix is passed as a
`[B, N]` C-order array, i.e. the same bytes).

There is no CPU fallback: if eigen-lstm_amd/liblstm_hip.so is missing or no gfx950 device is
visible, construction raises.
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("LSTM_HIP_LIB", os.path.join(HERE, "liblstm_hip.so"))  # env override: A/B builds only

FAST_MATH = 1
STEP_KERNELS = 4
DEBUG_STAMPS = 16
NO_FUSED_GRADS = 64
BF16_RECURRENCE = 128
LOSS_ALL_STEPS_BITS, LOSS_LAST_STEP_NATS, LOSS_LAST_STEP_BITS = 0, 1, 2
UNIQUE_ID_BYTES = 128
VOCAB = 256

P_PARAMS, P_GRADS, P_MEM = 0, 1, 2


class LstmHipError(RuntimeError):
    pass


class _Config(C.Structure):
    _fields_ = [("N", C.c_int32), ("M", C.c_int32), ("S", C.c_int32), ("B", C.c_int32), ("device", C.c_int32),
                ("flags", C.c_uint32)]


_lib = None

# every symbol include/lstm_hip.h declares (tests check the library exports all of them)
SYMBOLS = [
    "lstm_hip_create", "lstm_hip_destroy", "lstm_hip_last_error", "lstm_hip_param_count", "lstm_hip_set_params",
    "lstm_hip_get_params", "lstm_hip_set_state", "lstm_hip_get_state", "lstm_hip_get_activations",
    "lstm_hip_set_window", "lstm_hip_set_inputs_dense", "lstm_hip_slide_state", "lstm_hip_forward", "lstm_hip_loss", "lstm_hip_backward",
    "lstm_hip_adagrad", "lstm_hip_comm_unique_id", "lstm_hip_comm_init", "lstm_hip_allreduce_grads",
    "lstm_hip_set_text", "lstm_hip_set_cursors", "lstm_hip_get_cursors", "lstm_hip_reset_window",
    "lstm_hip_get_window", "lstm_hip_train_windows", "lstm_hip_set_global_batch", "lstm_hip_set_loss_mode", "lstm_hip_set_stride", "lstm_hip_eval_bits",
    "lstm_hip_sample", "lstm_hip_synchronize", "lstm_hip_set_profiling", "lstm_hip_kernel_stat_count",
    "lstm_hip_kernel_stat", "lstm_hip_reset_kernel_stats", "lstm_hip_device_info", "lstm_hip_debug_stamps",
]


def load_library():
    """dlopen the in-tree C-ABI library; raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise LstmHipError(f"{LIB_PATH} not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(there is no CPU fallback for the HIP path)")
    lib = C.CDLL(LIB_PATH)
    lib.lstm_hip_last_error.restype = C.c_char_p
    lib.lstm_hip_param_count.restype = C.c_size_t
    lib.lstm_hip_param_count.argtypes = [C.c_int32, C.c_int32]
    _lib = lib
    return lib


def param_count(N, M=VOCAB):
    return load_library().lstm_hip_param_count(N, M)


def _chk(rc):
    if rc != 0:
        raise LstmHipError(f"lstm_hip error {rc}: {load_library().lstm_hip_last_error().decode()}")


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _ptr(a, t=C.c_float):
    return a.ctypes.data_as(C.POINTER(t))


def device_info(device=0):
    lib = load_library()
    name = C.create_string_buffer(64)
    cus, mhz = C.c_int32(), C.c_int32()
    _chk(lib.lstm_hip_device_info(device, name, C.byref(cus), C.byref(mhz)))
    return name.value.decode(), cus.value, mhz.value


def comm_unique_id():
    buf = (C.c_uint8 * UNIQUE_ID_BYTES)()
    _chk(load_library().lstm_hip_comm_unique_id(buf))
    return bytes(buf)


class Lstm:
    """cuParameters p,d,m + cuLSTM<S> (OV/lstm_eigen_class_CUDA/cu_lstm.h) behind one handle."""

    def __init__(self, N, S, B, device=0, flags=0, M=VOCAB):
        self.lib = load_library()
        self.N, self.M, self.S, self.B = N, M, S, B
        self.np = param_count(N, M)
        self._h = C.c_void_p()
        cfg = _Config(N, M, S, B, device, flags)
        _chk(self.lib.lstm_hip_create(C.byref(cfg), C.byref(self._h)))

    def close(self):
        if self._h:
            self.lib.lstm_hip_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- copy_parameters_to_device / _to_host -------------------------------------------------
    def set_params(self, block, which=P_PARAMS):
        block = _f32(block)
        assert block.size == self.np, (block.size, self.np)
        _chk(self.lib.lstm_hip_set_params(self._h, which, _ptr(block)))

    def get_params(self, which=P_PARAMS):
        out = np.empty(self.np, np.float32)
        _chk(self.lib.lstm_hip_get_params(self._h, which, _ptr(out)))
        return out

    def get_grads(self):
        return self.get_params(P_GRADS)

    # ---- copy_lstm_to_device / copy_context_to_host -------------------------------------------
    def set_state(self, t, h=None, c=None):
        h = None if h is None else _f32(h)
        c = None if c is None else _f32(c)
        for a in (h, c):
            assert a is None or a.size == self.N * self.B
        _chk(self.lib.lstm_hip_set_state(self._h, t, _ptr(h) if h is not None else None,
                                         _ptr(c) if c is not None else None))

    def get_state(self, t):
        h = np.empty((self.B, self.N), np.float32)
        c = np.empty((self.B, self.N), np.float32)
        _chk(self.lib.lstm_hip_get_state(self._h, t, _ptr(h), _ptr(c)))
        return h, c

    def get_activations(self, t):
        g = np.empty((self.B, 4 * self.N), np.float32)
        p = np.empty((self.B, self.M), np.float32)
        _chk(self.lib.lstm_hip_get_activations(self._h, t, _ptr(g), _ptr(p)))
        return g, p

    # ---- copy_inputs_to_device ----------------------------------------------------------------
    def set_window(self, xi, ti):
        xi = np.ascontiguousarray(xi, dtype=np.int32)
        ti = np.ascontiguousarray(ti, dtype=np.int32)
        assert xi.shape == (self.S, self.B) and ti.shape == (self.S, self.B)
        _chk(self.lib.lstm_hip_set_window(self._h, _ptr(xi, C.c_int32), _ptr(ti, C.c_int32)))

    def set_inputs_dense(self, x, target, h0=None, c0=None):
        """copy_inputs_to_device with the reference's own operands: dense one-hot x[t], target[t] as [S, B, M] arrays
        (= M x B column-major matrices back to back) and optionally h[0], c[0] as [B, N]."""
        x, target = _f32(x), _f32(target)
        assert x.shape == (self.S, self.B, self.M) and target.shape == (self.S, self.B, self.M)
        h0 = None if h0 is None else _f32(h0)
        c0 = None if c0 is None else _f32(c0)
        _chk(self.lib.lstm_hip_set_inputs_dense(self._h, _ptr(h0) if h0 is not None else None,
                                                _ptr(c0) if c0 is not None else None, _ptr(x), _ptr(target)))

    def get_window(self):
        xi = np.empty((self.S, self.B), np.int32)
        ti = np.empty((self.S, self.B), np.int32)
        _chk(self.lib.lstm_hip_get_window(self._h, _ptr(xi, C.c_int32), _ptr(ti, C.c_int32)))
        return xi, ti

    def slide_state(self):
        _chk(self.lib.lstm_hip_slide_state(self._h))

    # ---- cuLSTM::forward / calculate_loss / backward, cuda_adagrad ------------------------------
    def forward(self):
        _chk(self.lib.lstm_hip_forward(self._h))

    def loss(self):
        out = C.c_double()
        _chk(self.lib.lstm_hip_loss(self._h, C.byref(out)))
        return out.value

    def backward(self):
        _chk(self.lib.lstm_hip_backward(self._h))

    def adagrad(self, lr):
        _chk(self.lib.lstm_hip_adagrad(self._h, C.c_double(lr)))

    # ---- data-parallel -----------------------------------------------------------------------
    def comm_init(self, unique_id, nranks, rank):
        buf = (C.c_uint8 * UNIQUE_ID_BYTES).from_buffer_copy(unique_id)
        _chk(self.lib.lstm_hip_comm_init(self._h, buf, nranks, rank))

    def allreduce_grads(self):
        _chk(self.lib.lstm_hip_allreduce_grads(self._h))

    def set_stride(self, stride, carry_col=1):
        _chk(self.lib.lstm_hip_set_stride(self._h, stride, carry_col))

    def set_global_batch(self, gb):
        _chk(self.lib.lstm_hip_set_global_batch(self._h, gb))

    def set_loss_mode(self, mode):
        """LOSS_ALL_STEPS_BITS (R/lstm.cc:204-207), LOSS_LAST_STEP_NATS (OV/lstm_eigen_class_CUDA/lstm.h:200-221) or
        LOSS_LAST_STEP_BITS (cuLSTM::calculate_loss, OV/lstm_eigen_class_CUDA/cu_lstm.h:203-215)."""
        _chk(self.lib.lstm_hip_set_loss_mode(self._h, mode))

    # ---- device-resident loop ------------------------------------------------------------------
    def set_text(self, text):
        text = np.ascontiguousarray(text, dtype=np.uint8)
        _chk(self.lib.lstm_hip_set_text(self._h, _ptr(text, C.c_uint8), C.c_size_t(text.size)))

    def set_cursors(self, pos):
        pos = np.ascontiguousarray(pos, dtype=np.uint64)
        assert pos.size == self.B
        _chk(self.lib.lstm_hip_set_cursors(self._h, _ptr(pos, C.c_uint64)))

    def get_cursors(self):
        pos = np.empty(self.B, np.uint64)
        _chk(self.lib.lstm_hip_get_cursors(self._h, _ptr(pos, C.c_uint64)))
        return pos

    def reset_window(self):
        _chk(self.lib.lstm_hip_reset_window(self._h))

    def train_windows(self, count, lr, want_losses=True, want_time=False):
        losses = np.zeros(count, np.float64) if want_losses else None
        ms = C.c_float(0)
        _chk(self.lib.lstm_hip_train_windows(self._h, C.c_int64(count), C.c_double(lr),
                                             _ptr(losses, C.c_double) if want_losses else None,
                                             C.byref(ms) if want_time else None))
        if want_time:
            return losses, ms.value
        return losses

    # ---- evaluator / sampler -------------------------------------------------------------------
    def eval_bits(self, text):
        text = np.ascontiguousarray(text, dtype=np.uint8)
        out = C.c_double()
        _chk(self.lib.lstm_hip_eval_bits(self._h, _ptr(text, C.c_uint8), C.c_size_t(text.size), C.byref(out)))
        return out.value

    def sample(self, h0, c0, u):
        h0, c0 = _f32(h0).copy(), _f32(c0).copy()
        u = np.ascontiguousarray(u, dtype=np.float64)
        out = np.zeros(u.size, np.uint8)
        _chk(self.lib.lstm_hip_sample(self._h, _ptr(h0), _ptr(c0), _ptr(u, C.c_double), int(u.size),
                                      _ptr(out, C.c_uint8)))
        return out, h0, c0

    def debug_stamps(self):
        out = np.zeros((4, self.S, 16), np.uint64)  # [fwd wg0, fwd wg1, bwd wg0, bwd wg1][step][slot]
        _chk(self.lib.lstm_hip_debug_stamps(self._h, _ptr(out, C.c_uint64), C.c_size_t(out.size)))
        return out

    # ---- measurement ---------------------------------------------------------------------------
    def synchronize(self):
        _chk(self.lib.lstm_hip_synchronize(self._h))

    def set_profiling(self, on):
        _chk(self.lib.lstm_hip_set_profiling(self._h, 1 if on else 0))

    def reset_kernel_stats(self):
        _chk(self.lib.lstm_hip_reset_kernel_stats(self._h))

    def kernel_stats(self):
        out = {}
        for i in range(self.lib.lstm_hip_kernel_stat_count(self._h)):
            name, n, ms = C.c_char_p(), C.c_int64(), C.c_double()
            _chk(self.lib.lstm_hip_kernel_stat(self._h, i, C.byref(name), C.byref(n), C.byref(ms)))
            out[name.value.decode()] = (n.value, ms.value)
        return out


# ---- the build's seeded stand-ins for the reference's unseeded randomness -----------------------
class MT19937Normal:
    """MT19937 (init_genrand seeding) + 53-bit uniforms (genrand_res53) + Marsaglia polar normals:
    the seeded replacement for the reference's `std::mt19937 mt(rd()); std::normal_distribution<>`
    (R/lstm.cc:370-372).  Spec shared with the C++ host driver (host/rng.h); independent of oracle/.

    numpy's legacy RandomState(seed).random_sample() IS init_genrand + genrand_res53, so the uniform
    stream comes from it; the polar transform is applied to consecutive uniform pairs in order and
    the normals are buffered, which is exactly the sequential algorithm (accepted pair -> u*m then v*m).
    """

    def __init__(self, seed):
        self._rs = np.random.RandomState(int(seed) & 0xFFFFFFFF)
        self._buf = np.empty(0, np.float64)

    def uniform(self):
        assert self._buf.size == 0, "uniform() after buffered normals would reorder the stream"
        return float(self._rs.random_sample())

    def _refill(self, need):
        chunks = [self._buf]
        have = self._buf.size
        while have < need:
            k = max(1024, int((need - have) * 0.7))
            uv = 2.0 * self._rs.random_sample(2 * k).reshape(k, 2) - 1.0
            s = (uv * uv).sum(axis=1)
            ok = (s < 1.0) & (s != 0.0)
            uv, s = uv[ok], s[ok]
            m = np.sqrt(-2.0 * np.log(s) / s)
            z = (uv * m[:, None]).ravel()  # u0*m0, v0*m0, u1*m1, ...
            chunks.append(z)
            have += z.size
        self._buf = np.concatenate(chunks)

    def normals(self, n):
        if self._buf.size < n:
            self._refill(n)
        out, self._buf = self._buf[:n], self._buf[n:]
        return out

    def randn(self, rows, cols, mean, std):
        """row-outer / column-inner fill order of R/lstm.cc:374-378; returns [cols, rows] C-order
        (= the bytes of a column-major rows x cols matrix)."""
        z = self.normals(rows * cols).reshape(rows, cols)
        return np.ascontiguousarray((mean + std * z).astype(np.float32).T)


def init_params(rng, N, M=VOCAB, forget_bias=0.0):
    """R/lstm.cc:113-119: W, U, Why ~ N(0, 0.01) in that order, b = by = 0; flat block.
    forget_bias = 1 reproduces the later variants' b[2N:3N] = 1 (OV/lstm_eigen_class_batch/lstm.cc:81)."""
    W = rng.randn(4 * N, M, 0.0, 0.01)
    U = rng.randn(4 * N, N, 0.0, 0.01)
    Why = rng.randn(M, N, 0.0, 0.01)
    b = np.zeros(4 * N, np.float32)
    b[2 * N:3 * N] = forget_bias
    return np.concatenate([W.ravel(), U.ravel(), b, Why.ravel(), np.zeros(M, np.float32)])


def initial_cursors(length, S, B, stream0=0, streams_total=None):
    """pos[b] = S + (b*(len-S))/B: deterministic stand-in for rand()%(len-S)+S (OV/lstm_eigen_opt/lstm.cc:140-144)."""
    total = B if streams_total is None else streams_total
    return np.array([S + ((stream0 + b) * (length - S)) // total for b in range(B)], dtype=np.uint64)
