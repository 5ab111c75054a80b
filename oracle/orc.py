"""TEST INFRASTRUCTURE ONLY: ctypes front-end to oracle/liblpcnet_oracle.so (our plain-C
restatement of the reference's generic-C arithmetic).  Builds the library on first use with
gcc (available on the GPU box too).  Importable only from tests/, bench.py's cpu_baseline leg
and __graft_entry__.smoke()."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liblpcnet_oracle.so")

_f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
_i16p = np.ctypeslib.ndpointer(np.int16, flags="C_CONTIGUOUS")
_u32p = np.ctypeslib.ndpointer(np.uint32, flags="C_CONTIGUOUS")
_u8p = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")


def build(force=False):
    srcs = [os.path.join(_HERE, f) for f in ("lpcnet_oracle.c", "lpcnet_oracle.h", "orc_tables_gen.h")]
    if (not force and os.path.exists(_SO)
            and all(os.path.getmtime(_SO) >= os.path.getmtime(s) for s in srcs)):
        return _SO
    subprocess.check_call(["gcc", "-O2", "-fPIC", "-shared", "-ffp-contract=off", "-fno-fast-math",
                           "-Wall", "-o", _SO, srcs[0], "-lm"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    L = C.CDLL(build())
    vp = C.c_void_p
    L.orc_model_parse.argtypes = [C.c_char_p, C.c_int, C.c_float]
    L.orc_model_parse.restype = vp
    L.orc_model_set_end2end.argtypes = [vp, C.c_int]
    L.orc_model_set_end2end.restype = None
    L.orc_model_free.argtypes = [vp]
    L.orc_model_is_int8.argtypes = [vp]
    L.orc_model_nb_blocks.argtypes = [vp, C.c_int]
    L.orc_state_create.argtypes = [vp]
    L.orc_state_create.restype = vp
    L.orc_state_destroy.argtypes = [vp]
    L.orc_state_reset.argtypes = [vp]
    L.orc_frame_network.argtypes = [vp, _f32p, _f32p, _f32p, _f32p]
    L.orc_sample_network.argtypes = [vp, _f32p, _f32p, C.c_int, C.c_int, C.c_int]
    L.orc_synthesize.argtypes = [vp, _f32p, _i16p, C.c_int, C.c_int]
    L.orc_synthesize_tail.argtypes = [vp, _f32p, _f32p, _f32p, _i16p, C.c_int, C.c_int]
    L.orc_get_nnet_state.argtypes = [vp, _f32p, _f32p, _f32p, _f32p]
    L.orc_set_gru_state.argtypes = [vp, _f32p, _f32p]
    L.orc_get_frame_products.argtypes = [vp, _f32p, _f32p, _f32p]
    L.orc_get_signal_state.argtypes = [vp, _f32p, C.POINTER(C.c_int), C.POINTER(C.c_float),
                                       C.POINTER(C.c_int), _u32p]
    L.orc_force_frame_count.argtypes = [vp, C.c_int]
    L.orc_gru_a_input.argtypes = [vp, _f32p, _f32p, C.c_int, C.c_int, C.c_int]
    L.orc_sparse_gru_a.argtypes = [vp, _f32p, _f32p]
    L.orc_gru_b.argtypes = [vp, _f32p, _f32p, _f32p]
    L.orc_sample_mdense.argtypes = [vp, _f32p, _u32p]
    L.orc_mdense_f16_path.argtypes = [vp, _f32p, C.c_int, C.c_int, _f32p]
    L.orc_mdense_f16_path.restype = None
    L.orc_f16.argtypes = [C.c_float]
    L.orc_f16.restype = C.c_float
    L.orc_lin2ulaw.argtypes = [C.c_float]
    L.orc_ulaw2lin.argtypes = [C.c_int]
    L.orc_ulaw2lin.restype = C.c_float
    L.orc_tanh_approx.argtypes = [C.c_float]
    L.orc_tanh_approx.restype = C.c_float
    L.orc_sigmoid_approx.argtypes = [C.c_float]
    L.orc_sigmoid_approx.restype = C.c_float
    L.orc_kiss99_srand.argtypes = [_u32p, C.c_char_p, C.c_int]
    L.orc_kiss99_rand.argtypes = [_u32p]
    L.orc_kiss99_rand.restype = C.c_uint32
    L.orc_lpc_from_cepstrum.argtypes = [_f32p, _f32p]
    L.orc_fft320.argtypes = [_f32p, _f32p]
    L.orc_logit_table.argtypes = [C.c_int]
    L.orc_logit_table.restype = C.c_float
    L.orc_decode_packet.argtypes = [_f32p, _f32p, _u8p, _f32p, _f32p, _f32p, _f32p]
    _lib = L
    return L


class OracleModel:
    def __init__(self, blob: bytes, lpc_gamma: float = 1.0, end2end: bool = False):
        self.L = lib()
        self._blob = C.create_string_buffer(blob, len(blob))
        self.p = self.L.orc_model_parse(self._blob, len(blob), lpc_gamma)
        if not self.p:
            raise ValueError("oracle: malformed weight blob")
        if end2end:
            self.L.orc_model_set_end2end(self.p, 1)

    def __del__(self):
        try:
            self.L.orc_model_free(self.p)
        except Exception:
            pass

    @property
    def is_int8(self):
        return bool(self.L.orc_model_is_int8(self.p))

    def new_state(self):
        return OracleState(self)

    def mdense_f16_path(self, gru_b: np.ndarray, path: int, variant: int = 0) -> np.ndarray:
        """logits of the 8 tree nodes on `path` in the engine's fp16 dual-FC arithmetic (oracle-side restatement)"""
        out = np.zeros(8, np.float32)
        self.L.orc_mdense_f16_path(self.p, np.ascontiguousarray(gru_b, np.float32), int(path), int(variant), out)
        return out


class OracleState:
    def __init__(self, model: OracleModel):
        self.model = model
        self.L = model.L
        self.p = self.L.orc_state_create(model.p)

    def __del__(self):
        try:
            self.L.orc_state_destroy(self.p)
        except Exception:
            pass

    def synthesize(self, features: np.ndarray, preload_pcm=None) -> np.ndarray:
        T = features.shape[0]
        out = np.zeros(T * 160, np.int16)
        for t in range(T):
            f = np.ascontiguousarray(features[t, :20], np.float32)
            frame = out[t * 160:(t + 1) * 160]
            if preload_pcm is None:
                self.L.orc_synthesize(self.p, f, frame, 160, 0)
            else:
                frame[:] = preload_pcm[t * 160:(t + 1) * 160]
                self.L.orc_synthesize(self.p, f, frame, 160, 160)
        return out

    def frame_network(self, feat20):
        ca = np.zeros(1152, np.float32)
        cb = np.zeros(48, np.float32)
        lpc = np.zeros(16, np.float32)
        self.L.orc_frame_network(self.p, np.ascontiguousarray(feat20[:20], np.float32), ca, cb, lpc)
        return lpc, ca, cb

    def frame_products(self):
        lpc = np.zeros(16, np.float32)
        ca = np.zeros(1152, np.float32)
        cb = np.zeros(48, np.float32)
        self.L.orc_get_frame_products(self.p, lpc, ca, cb)
        return lpc, ca, cb

    def nnet_state(self):
        c1 = np.zeros(168, np.float32)
        c2 = np.zeros(256, np.float32)
        ga = np.zeros(384, np.float32)
        gb = np.zeros(16, np.float32)
        self.L.orc_get_nnet_state(self.p, c1, c2, ga, gb)
        return c1, c2, ga, gb

    def signal_state(self):
        ls = np.zeros(16, np.float32)
        le, dm, fc = C.c_int(), C.c_float(), C.c_int()
        rng = np.zeros(4, np.uint32)
        self.L.orc_get_signal_state(self.p, ls, C.byref(le), C.byref(dm), C.byref(fc), rng)
        return ls, le.value, dm.value, fc.value, rng


# ---- many independent streams on the host cores (tests and bench.py's checker leg) ---------------------------
def _synth_worker(args):
    blob, feats, lpc_gamma, end2end = args
    om = OracleModel(blob, lpc_gamma=lpc_gamma, end2end=end2end)
    return np.stack([om.new_state().synthesize(f) for f in feats])


def synthesize_many(blob: bytes, feats: np.ndarray, workers: int | None = None, lpc_gamma: float = 1.0, end2end: bool = False) -> np.ndarray:
    """feats (n, T, >=20) -> pcm (n, T*160): n fresh oracle states, spread over a pool of `workers` processes
    (spawned, so a parent that already initialised HIP is not forked)."""
    import multiprocessing as mp
    import os
    n = feats.shape[0]
    if workers is None:
        try:
            workers = len(os.sched_getaffinity(0))
        except AttributeError:
            workers = os.cpu_count() or 1
    workers = max(1, min(workers, n, 32))
    if workers == 1:
        return _synth_worker((blob, feats, lpc_gamma, end2end))
    build()
    chunks = np.array_split(np.arange(n), workers * 2)
    with mp.get_context("spawn").Pool(workers) as pool:
        parts = pool.map(_synth_worker, [(blob, np.ascontiguousarray(feats[c]), lpc_gamma, end2end) for c in chunks if len(c)])
    return np.concatenate(parts, axis=0)
