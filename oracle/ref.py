"""TEST INFRASTRUCTURE ONLY: ctypes front-end to oracle/_ref/liblpcnet_ref_<flavour>.so,
i.e. to the *real reference* compiled by oracle/Makefile (plus oracle/ref_harness.c accessors).
Imported only by tests/, bench.py's cpu_baseline leg and tools that generate golden fixtures.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
REF_DIR = os.path.join(_HERE, "_ref")

_f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
_i16p = np.ctypeslib.ndpointer(np.int16, flags="C_CONTIGUOUS")
_u32p = np.ctypeslib.ndpointer(np.uint32, flags="C_CONTIGUOUS")
_u8p = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")


def available(flavour="gf") -> bool:
    return os.path.exists(os.path.join(REF_DIR, f"liblpcnet_ref_{flavour}.so"))


class RefLib:
    """One loaded flavour of the compiled reference (gf, gi, af, ai)."""

    def __init__(self, flavour="gf"):
        path = os.path.join(REF_DIR, f"liblpcnet_ref_{flavour}.so")
        self.flavour = flavour
        self.lib = lib = C.CDLL(path)
        vp = C.c_void_p
        lib.lpcnet_create.restype = vp
        lib.lpcnet_destroy.argtypes = [vp]
        lib.lpcnet_get_size.restype = C.c_int
        lib.lpcnet_reset.argtypes = [vp]
        lib.lpcnet_load_model.argtypes = [vp, C.c_char_p, C.c_int]
        lib.lpcnet_load_model.restype = C.c_int
        lib.lpcnet_synthesize.argtypes = [vp, _f32p, _i16p, C.c_int]
        lib.lpcnet_decoder_create.restype = vp
        lib.lpcnet_decoder_destroy.argtypes = [vp]
        lib.lpcnet_decode.argtypes = [vp, _u8p, _i16p]
        lib.ref_flavour.restype = C.c_int
        lib.ref_set_codebooks.argtypes = [_f32p] * 4
        lib.ref_get_nnet_state.argtypes = [vp, _f32p, _f32p, _f32p, _f32p]
        lib.ref_set_gru_state.argtypes = [vp, _f32p, _f32p]
        lib.ref_get_frame_products.argtypes = [vp, _f32p, _f32p, _f32p]
        lib.ref_get_signal_state.argtypes = [vp, _f32p, C.POINTER(C.c_int), C.POINTER(C.c_float),
                                             C.POINTER(C.c_int), _u32p]
        lib.ref_get_logit_table.argtypes = [vp, _f32p]
        lib.ref_synthesize_impl.argtypes = [vp, _f32p, _i16p, C.c_int, C.c_int]
        lib.ref_run_frame_network.argtypes = [vp, _f32p, _f32p, _f32p, _f32p]
        lib.ref_run_sample_network.argtypes = [vp, _f32p, _f32p, C.c_int, C.c_int, C.c_int]
        lib.ref_run_sample_network.restype = C.c_int
        lib.ref_gru_a_input.argtypes = [vp, _f32p, _f32p, C.c_int, C.c_int, C.c_int]
        lib.ref_sparse_gru_a.argtypes = [vp, _f32p, _f32p]
        lib.ref_gru_b.argtypes = [vp, _f32p, _f32p, _f32p]
        lib.ref_sample_mdense.argtypes = [vp, _f32p, _u32p]
        lib.ref_sample_mdense.restype = C.c_int
        lib.ref_mdense_logits.argtypes = [vp, _f32p, _f32p]
        lib.ref_lin2ulaw.argtypes = [C.c_float]
        lib.ref_lin2ulaw.restype = C.c_int
        lib.ref_ulaw2lin.argtypes = [C.c_float]
        lib.ref_ulaw2lin.restype = C.c_float
        lib.ref_vec_tanh.argtypes = [_f32p, _f32p, C.c_int]
        lib.ref_vec_sigmoid.argtypes = [_f32p, _f32p, C.c_int]
        lib.ref_lpc_from_cepstrum.argtypes = [_f32p, _f32p]
        lib.ref_lpc_from_cepstrum.restype = C.c_float
        lib.ref_lpc_weighting.argtypes = [_f32p, C.c_float]
        lib.ref_decode_packet.argtypes = [_f32p, _f32p, _u8p]
        lib.kiss99_srand.argtypes = [_u32p, C.c_char_p, C.c_int]
        lib.kiss99_rand.argtypes = [_u32p]
        lib.kiss99_rand.restype = C.c_uint32

    # -- synthesis states ---------------------------------------------------------------
    def new_state(self, blob: bytes) -> "RefState":
        return RefState(self, blob)

    def synthesize_file(self, blob: bytes, features: np.ndarray) -> np.ndarray:
        """What `lpcnet_demo -synthesis` does (src/lpcnet_demo.c:202-219)."""
        st = self.new_state(blob)
        return st.synthesize(features)


class RefState:
    def __init__(self, ref: RefLib, blob: bytes):
        self.ref = ref
        self.lib = ref.lib
        self._blob = C.create_string_buffer(blob, len(blob))   # must outlive the state
        self.p = self.lib.lpcnet_create()
        ret = self.lib.lpcnet_load_model(self.p, self._blob, len(blob))
        if ret != 0:
            raise RuntimeError("reference lpcnet_load_model failed")

    def __del__(self):
        try:
            self.lib.lpcnet_destroy(self.p)
        except Exception:
            pass

    def synthesize(self, features: np.ndarray, preload_pcm: np.ndarray | None = None) -> np.ndarray:
        """features (T,>=20) -> int16 (T*160,).  With preload_pcm the loop is teacher-forced
        (src/lpcnet.c:256-259): the given samples are fed back instead of the sampled ones."""
        T = features.shape[0]
        out = np.zeros(T * 160, np.int16)
        for t in range(T):
            f = np.ascontiguousarray(features[t, :20], np.float32)
            frame = out[t * 160:(t + 1) * 160]
            if preload_pcm is None:
                self.lib.lpcnet_synthesize(self.p, f, frame, 160)
            else:
                frame[:] = preload_pcm[t * 160:(t + 1) * 160]
                self.lib.ref_synthesize_impl(self.p, f, frame, 160, 160)
        return out

    def frame_products(self):
        lpc = np.zeros(16, np.float32)
        ca = np.zeros(1152, np.float32)
        cb = np.zeros(48, np.float32)
        self.lib.ref_get_frame_products(self.p, lpc, ca, cb)
        return lpc, ca, cb

    def nnet_state(self):
        c1 = np.zeros(168, np.float32)
        c2 = np.zeros(256, np.float32)
        ga = np.zeros(384, np.float32)
        gb = np.zeros(16, np.float32)
        self.lib.ref_get_nnet_state(self.p, c1, c2, ga, gb)
        return c1, c2, ga, gb

    def signal_state(self):
        ls = np.zeros(16, np.float32)
        le = C.c_int()
        dm = C.c_float()
        fc = C.c_int()
        rng = np.zeros(4, np.uint32)
        self.lib.ref_get_signal_state(self.p, ls, C.byref(le), C.byref(dm), C.byref(fc), rng)
        return ls, le.value, dm.value, fc.value, rng
