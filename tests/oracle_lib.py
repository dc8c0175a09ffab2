"""ctypes binding of oracle/liblstm_ref_*.so -- the CPU checker.  Test infrastructure only.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")

_f = C.POINTER(C.c_float)
_d = C.POINTER(C.c_double)
_i32 = C.POINTER(C.c_int32)


def build_oracle():
    """Compile the oracle libraries if a .so is missing or older than the source."""
    src = os.path.join(ORACLE_DIR, "lstm_ref.c")
    libs = [os.path.join(ORACLE_DIR, n) for n in ("liblstm_ref_f32.so", "liblstm_ref_f64.so", "liblstm_ref_f32_omp.so")]
    stale = [l for l in libs if not os.path.exists(l) or os.path.getmtime(l) < os.path.getmtime(src)]
    if stale:
        subprocess.check_call(["make", "-C", ORACLE_DIR, "-s"])
    return libs


def cpu_share():
    """Cores this process may really use: the affinity mask, cut to the cgroup's CPU quota when there is one (a GPU box
    shows every host core but grants its containers a share; OpenMP's default of one thread per visible core then
    oversubscribes the share many times over and its spinning barriers crawl)."""
    cores = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            cores = max(1, min(cores, -(-int(quota) // int(period))))
    except (OSError, ValueError):
        pass
    return cores


class Oracle:
    """One precision of the oracle.  kind: 'f32', 'f64' or 'f32_omp'."""

    def __init__(self, kind="f32"):
        build_oracle()
        if kind.endswith("_omp"):
            os.environ.setdefault("OMP_NUM_THREADS", str(cpu_share()))  # read by libgomp when the library loads
        self.kind = kind
        self.suf = "_f64" if kind == "f64" else "_f32"
        self.np_t = np.float64 if kind == "f64" else np.float32
        self.c_t = C.c_double if kind == "f64" else C.c_float
        self.lib = C.CDLL(os.path.join(ORACLE_DIR, f"liblstm_ref_{kind}.so"))
        if kind.endswith("_omp"):  # also when libgomp was initialised before the variable was set
            try:
                C.CDLL("libgomp.so.1").omp_set_num_threads(int(os.environ["OMP_NUM_THREADS"]))
            except (OSError, AttributeError, ValueError):
                pass
        L = self.lib
        L.ref_rng_sizeof.restype = C.c_size_t
        self._fn("ref_param_count").restype = C.c_size_t
        self._fn("ref_eval_bits").restype = C.c_double
        self._fn("ref_trainer_create").restype = C.c_void_p
        self._fn("ref_trainer_window").restype = C.c_double
        for n in ("params", "grads", "mem", "h", "c", "g", "probs"):
            self._fn("ref_trainer_" + n).restype = C.POINTER(self.c_t)
        for n in ("xi", "ti"):
            self._fn("ref_trainer_" + n).restype = _i32
        self._fn("ref_trainer_rng").restype = C.c_void_p

    def _fn(self, name):
        return getattr(self.lib, name + self.suf)

    def _p(self, a):
        return a.ctypes.data_as(C.POINTER(self.c_t))

    def set_bf16_recurrence(self, on):
        """bf16 operands (RNE) in the two recurrent products, fp32 accumulate (BASELINE configs[4] semantics)."""
        self.lib.ref_set_bf16_recurrence(1 if on else 0)

    def set_bf16_products(self, on):
        """bf16 operands also in the four time-batched products (y, dWhy, Why^T*dy, dU): with set_bf16_recurrence(True)
        this is the complete bf16 MFMA path of BASELINE configs[4]."""
        self.lib.ref_set_bf16_products(1 if on else 0)

    # ---- RNG -------------------------------------------------------------------------------
    def rng(self, seed):
        buf = C.create_string_buffer(self.lib.ref_rng_sizeof())
        self.lib.ref_rng_seed(buf, C.c_uint32(seed))
        return buf

    def randn(self, rng, rows, cols, mean, std):
        m = np.zeros((rows, cols), dtype=self.np_t, order="F")
        self._fn("ref_randn")(rng, self._p(m), rows, cols, C.c_double(mean), C.c_double(std))
        return m

    def rng_u32(self, rng):
        self.lib.ref_rng_u32.restype = C.c_uint32
        return self.lib.ref_rng_u32(rng)

    def rng_uniform(self, rng):
        self.lib.ref_rng_uniform.restype = C.c_double
        return self.lib.ref_rng_uniform(rng)

    # ---- params ----------------------------------------------------------------------------
    def param_count(self, N, M=256):
        return self._fn("ref_param_count")(N, M)

    def init_params(self, rng, N, M=256):
        P = np.zeros(self.param_count(N, M), dtype=self.np_t)
        self._fn("ref_init_params")(rng, self._p(P), N, M)
        return P

    # ---- window ops ------------------------------------------------------------------------
    def forward(self, N, M, S, B, P, xi, ti, h0, c0):
        """Returns dict(h,c,g,probs as [S, rows, B] views of the col-major buffers, loss_bits, loss_nats)."""
        t = self.np_t
        h = np.zeros((S, B, N), dtype=t)
        c = np.zeros((S, B, N), dtype=t)
        g = np.zeros((S, B, 4 * N), dtype=t)
        probs = np.zeros((S, B, M), dtype=t)
        h[0] = np.asarray(h0, dtype=t).T if np.asarray(h0).shape == (N, B) else np.asarray(h0, dtype=t)
        c[0] = np.asarray(c0, dtype=t).T if np.asarray(c0).shape == (N, B) else np.asarray(c0, dtype=t)
        xi = np.ascontiguousarray(xi, dtype=np.int32)
        ti = np.ascontiguousarray(ti, dtype=np.int32)
        lb, ln = C.c_double(), C.c_double()
        P = np.ascontiguousarray(P, dtype=t)
        self._fn("ref_forward")(N, M, S, B, self._p(P), xi.ctypes.data_as(_i32), ti.ctypes.data_as(_i32),
                                self._p(h), self._p(c), self._p(g), self._p(probs), C.byref(lb), C.byref(ln))
        return dict(h=h, c=c, g=g, probs=probs, loss_bits=lb.value, loss_nats=ln.value)

    def backward(self, N, M, S, B, P, xi, ti, fw):
        t = self.np_t
        dP = np.zeros(self.param_count(N, M), dtype=t)
        xi = np.ascontiguousarray(xi, dtype=np.int32)
        ti = np.ascontiguousarray(ti, dtype=np.int32)
        P = np.ascontiguousarray(P, dtype=t)
        self._fn("ref_backward")(N, M, S, B, self._p(P), xi.ctypes.data_as(_i32), ti.ctypes.data_as(_i32),
                                 self._p(fw["h"]), self._p(fw["c"]), self._p(fw["g"]), self._p(fw["probs"]), self._p(dP))
        return dP

    def adagrad(self, P, dP, mem, lr):
        self._fn("ref_adagrad")(C.c_size_t(P.size), self._p(P), self._p(dP), self._p(mem), self.c_t(lr))

    def numgrad(self, N, M, S, B, P, xi, ti, h0, c0, which, delta=1e-5):
        t = self.np_t
        P = np.ascontiguousarray(P, dtype=t).copy()
        xi = np.ascontiguousarray(xi, dtype=np.int32)
        ti = np.ascontiguousarray(ti, dtype=np.int32)
        h0 = np.ascontiguousarray(np.asarray(h0, dtype=t))  # [B, N] = col-major N x B
        c0 = np.ascontiguousarray(np.asarray(c0, dtype=t))
        which = np.ascontiguousarray(which, dtype=np.int64)
        out = np.zeros(which.size, dtype=np.float64)
        self._fn("ref_numgrad")(N, M, S, B, self._p(P), xi.ctypes.data_as(_i32), ti.ctypes.data_as(_i32),
                                self._p(h0), self._p(c0), which.ctypes.data_as(C.POINTER(C.c_int64)),
                                int(which.size), C.c_double(delta), out.ctypes.data_as(_d))
        return out

    def eval_bits(self, N, M, P, text):
        text = np.ascontiguousarray(text, dtype=np.uint8)
        P = np.ascontiguousarray(P, dtype=self.np_t)
        return self._fn("ref_eval_bits")(N, M, self._p(P), text.ctypes.data_as(C.POINTER(C.c_uint8)), C.c_size_t(text.size))

    def sample(self, N, M, P, h, c, u):
        P = np.ascontiguousarray(P, dtype=self.np_t)
        h = np.ascontiguousarray(h, dtype=self.np_t).copy()
        c = np.ascontiguousarray(c, dtype=self.np_t).copy()
        u = np.ascontiguousarray(u, dtype=np.float64)
        out = np.zeros(u.size, dtype=np.uint8)
        self._fn("ref_sample")(N, M, self._p(P), self._p(h), self._p(c), u.ctypes.data_as(_d), int(u.size),
                               out.ctypes.data_as(C.POINTER(C.c_uint8)))
        return out, h, c

    def trainer(self, text, N, S, B, lr=0.1, seed=1, M=256, stream0=0, streams_total=None):
        return Trainer(self, text, N, M, S, B, lr, seed, stream0, B if streams_total is None else streams_total)


class Trainer:
    """The reference's outer loop (OV/lstm_eigen_opt/lstm.cc:172-332) around the oracle ops."""

    def __init__(self, orc, text, N, M, S, B, lr, seed, stream0, streams_total):
        self.o, self.N, self.M, self.S, self.B = orc, N, M, S, B
        self.text = np.ascontiguousarray(text, dtype=np.uint8)  # kept alive: C side holds the pointer
        self.T = C.c_void_p(orc._fn("ref_trainer_create")(
            self.text.ctypes.data_as(C.POINTER(C.c_uint8)), C.c_size_t(self.text.size), N, M, S, B,
            C.c_double(lr), C.c_uint32(seed), stream0, streams_total))
        self.np = orc.param_count(N, M)

    def __del__(self):
        try:
            self.o._fn("ref_trainer_destroy")(self.T)
        except Exception:
            pass

    def _arr(self, name, shape, dtype=None):
        ptr = self.o._fn("ref_trainer_" + name)(self.T)
        return np.ctypeslib.as_array(ptr, shape=shape)

    def epoch_reset(self):
        self.o._fn("ref_trainer_epoch_reset")(self.T)

    def slide(self):
        self.o._fn("ref_trainer_slide")(self.T)

    def window(self, update=True):
        return self.o._fn("ref_trainer_window")(self.T, 1 if update else 0)

    @property
    def params(self):
        return self._arr("params", (self.np,))

    @property
    def grads(self):
        return self._arr("grads", (self.np,))

    @property
    def mem(self):
        return self._arr("mem", (self.np,))

    @property
    def h(self):
        return self._arr("h", (self.S, self.B, self.N))

    @property
    def c(self):
        return self._arr("c", (self.S, self.B, self.N))

    @property
    def g(self):
        return self._arr("g", (self.S, self.B, 4 * self.N))

    @property
    def probs(self):
        return self._arr("probs", (self.S, self.B, self.M))

    @property
    def xi(self):
        return self._arr("xi", (self.S, self.B))

    @property
    def ti(self):
        return self._arr("ti", (self.S, self.B))


def split_params(P, N, M=256):
    """Flat block -> dict of column-major matrices (as [rows, cols] Fortran-order views)."""
    o = 0
    out = {}
    for name, r, c in (("W", 4 * N, M), ("U", 4 * N, N), ("b", 4 * N, 1), ("Why", M, N), ("by", M, 1)):
        out[name] = P[o:o + r * c].reshape((r, c), order="F")
        o += r * c
    assert o == P.size
    return out
