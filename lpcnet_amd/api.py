"""Python mirror of the C API in include/lpcnet.h + include/lpcnet_batch.h (ctypes over the
C-ABI of liblpcnet_hip.so).  Same names and argument meaning as the reference's
include/lpcnet.h so that tests read like the reference's own driver (src/lpcnet_demo.c:202-219,
src/test_lpcnet.c:55-64).

There is no CPU fallback: importing works anywhere, but every entry point needs the HIP library
and a GPU, and fails loudly otherwise.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("LPCNET_HIP_LIB") or os.path.join(_HERE, "liblpcnet_hip.so")

NB_FEATURES = 20
NB_TOTAL_FEATURES = 36
LPCNET_FRAME_SIZE = 160
LPCNET_COMPRESSED_SIZE = 8
LPCNET_PACKET_SAMPLES = 640

_f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
_i16p = np.ctypeslib.ndpointer(np.int16, flags="C_CONTIGUOUS")
_u8p = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")

_lib = None


class LPCNetError(RuntimeError):
    pass


def load_library():
    """dlopen liblpcnet_hip.so (built by `python -m lpcnet_amd.build`).  Raises if it is absent:
    the product path never falls back to a CPU implementation."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise LPCNetError(f"{LIB_PATH} not built: run `python -m lpcnet_amd.build` (needs hipcc)")
    L = C.CDLL(LIB_PATH)
    vp = C.c_void_p
    L.lpcnet_get_size.restype = C.c_int
    L.lpcnet_init.argtypes = [vp]
    L.lpcnet_reset.argtypes = [vp]
    L.lpcnet_create.restype = vp
    L.lpcnet_destroy.argtypes = [vp]
    L.lpcnet_synthesize.argtypes = [vp, _f32p, _i16p, C.c_int]
    L.lpcnet_synthesize.restype = None
    L.lpcnet_load_model.argtypes = [vp, C.c_char_p, C.c_int]
    # the reference's internal entry points (src/lpcnet_private.h:125-132)
    L.lpcnet_reset_signal.argtypes = [vp]
    L.lpcnet_reset_signal.restype = None
    L.run_frame_network.argtypes = [vp, _f32p, _f32p, _f32p, _f32p]
    L.run_frame_network.restype = None
    L.run_frame_network_deferred.argtypes = [vp, _f32p]
    L.run_frame_network_deferred.restype = None
    L.run_frame_network_flush.argtypes = [vp]
    L.run_frame_network_flush.restype = None
    L.lpcnet_synthesize_tail_impl.argtypes = [vp, _i16p, C.c_int, C.c_int]
    L.lpcnet_synthesize_tail_impl.restype = None
    L.lpcnet_synthesize_impl.argtypes = [vp, _f32p, _i16p, C.c_int, C.c_int]
    L.lpcnet_synthesize_impl.restype = None
    L.lpcnet_decoder_get_size.restype = C.c_int
    L.lpcnet_decoder_init.argtypes = [vp]
    L.lpcnet_decoder_create.restype = vp
    L.lpcnet_decoder_destroy.argtypes = [vp]
    L.lpcnet_decode.argtypes = [vp, _u8p, _i16p]
    L.lpcnet_hip_last_error.restype = C.c_char_p
    L.lpcnet_hip_set_codebooks.argtypes = [_f32p] * 4
    L.lpcnet_hip_set_codebooks.restype = None
    L.lpcnet_hip_shutdown.restype = None
    L.lpcnet_hip_set_default_model.argtypes = [C.c_char_p, C.c_int]
    L.lpcnet_hip_decoder_load_model.argtypes = [vp, C.c_char_p, C.c_int]
    L.lpcnet_hip_set_device.argtypes = [C.c_int]
    L.lpcnet_batch_create_sharded.argtypes = [C.c_int, C.POINTER(C.c_int), C.c_int]
    L.lpcnet_batch_create_sharded.restype = vp
    L.lpcnet_batch_shards.argtypes = [vp]
    L.lpcnet_batch_shard_info.argtypes = [vp, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.lpcnet_batch_synthesize_device_shard.argtypes = [vp, C.c_int, vp, C.c_int, vp, C.c_int, vp]
    L.lpcnet_hip_check_model.argtypes = [C.c_char_p, C.c_int, C.POINTER(C.c_int)]
    L.lpcnet_batch_create.argtypes = [C.c_int, C.c_int]
    L.lpcnet_batch_create.restype = vp
    L.lpcnet_batch_destroy.argtypes = [vp]
    L.lpcnet_batch_streams.argtypes = [vp]
    L.lpcnet_batch_load_model.argtypes = [vp, C.c_char_p, C.c_int]
    L.lpcnet_batch_reset.argtypes = [vp, C.c_int, C.c_int]
    L.lpcnet_batch_synthesize.argtypes = [vp, _f32p, C.c_int, _i16p, C.c_int]
    L.lpcnet_batch_synthesize_device.argtypes = [vp, vp, C.c_int, vp, C.c_int, vp]
    L.lpcnet_batch_sync.argtypes = [vp]
    L.lpcnet_batch_synthesize_preload.argtypes = [vp, _f32p, C.c_int, _i16p, C.c_int, C.c_int]
    L.lpcnet_batch_synthesize_step.argtypes = [vp, vp, C.c_int, vp, vp, vp, vp]
    L.lpcnet_batch_decode.argtypes = [vp, _u8p, _i16p, C.c_int]
    L.lpcnet_batch_decode_device.argtypes = [vp, vp, vp, C.c_int, vp]
    L.lpcnet_batch_set_lpc_gamma.argtypes = [vp, C.c_float]
    L.lpcnet_batch_set_end2end.argtypes = [vp, C.c_int]
    L.lpcnet_batch_set_fast.argtypes = [vp, C.c_int]
    L.lpcnet_batch_export_state.argtypes = [vp, C.c_int, vp]
    L.lpcnet_batch_import_state.argtypes = [vp, C.c_int, vp]
    L.lpcnet_batch_set_streams_per_workgroup.argtypes = [vp, C.c_int]
    L.lpcnet_batch_get_streams_per_workgroup.argtypes = [vp]
    L.lpcnet_batch_tune.argtypes = [vp]
    L.lpcnet_batch_enable_timing.argtypes = [vp, C.c_int]
    L.lpcnet_batch_last_timing.argtypes = [vp, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    L.lpcnet_batch_last_error.restype = C.c_char_p
    L.lpcnet_batch_run_tail.argtypes = [vp, _f32p, _f32p, _f32p, _i16p, C.c_int, C.c_int]
    L.lpcnet_batch_run_frames.argtypes = [vp, _f32p, C.c_int, vp, vp, vp, C.c_int]
    L.lpcnet_batch_state_size.restype = C.c_int
    L.lpcnet_batch_get_raw_state.argtypes = [vp, C.c_int, vp]
    L.lpcnet_batch_set_raw_state.argtypes = [vp, C.c_int, vp]
    L.lpcnet_batch_debug_trace.argtypes = [vp, C.c_int, vp]
    L.lpcnet_batch_profile.argtypes = [vp, vp]
    _u32p = np.ctypeslib.ndpointer(np.uint32, flags="C_CONTIGUOUS")
    L.lpcnet_hip_arith_identities_device.argtypes = [_f32p, _f32p, _u32p, _u32p, _u32p, _u32p, C.c_int]
    L.lpcnet_hip_quant_sweep_device.argtypes = [C.POINTER(C.c_ulonglong)]
    L.lpcnet_hip_status.restype = C.c_int
    L.lpcnet_hip_clear_error.restype = None
    L.lpcnet_hip_model_status.argtypes = [vp, C.c_int]
    L.lpcnet_hip_build_info.restype = C.c_char_p
    L.lpcnet_hip_dispatch_stats.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
    _lib = L
    return L


def check_model(blob: bytes):
    """Host-only validation of a DNNw blob: returns (code, info[6]); see include/lpcnet.h."""
    info = (C.c_int * 6)()
    rc = load_library().lpcnet_hip_check_model(blob, len(blob), info)
    return rc, list(info)


def last_error() -> str:
    return load_library().lpcnet_hip_last_error().decode()


def status() -> int:
    """Sticky per-thread status of the void entry points (0 = no failure since clear_error())."""
    return load_library().lpcnet_hip_status()


def clear_error() -> None:
    load_library().lpcnet_hip_clear_error()


def build_info() -> dict:
    """{'src': hash of every source the library was built from, 'dev': hash of the device sources alone}"""
    txt = load_library().lpcnet_hip_build_info().decode()
    return dict(kv.split("=", 1) for kv in txt.split())


def dispatch_stats(reset=False):
    """(calls served, device passes, calls in the largest pass) of the combining dispatcher behind the per-state entry points"""
    out = (C.c_ulonglong * 3)()
    load_library().lpcnet_hip_dispatch_stats(out, 1 if reset else 0)
    return int(out[0]), int(out[1]), int(out[2])


def quant_sweep():
    """(mismatches over all finite |t| < 2^31, mismatches with |t| <= 127.5, a mismatching bit pattern) of v_cvt_rpi_i32_f32 vs floor(.5 + (double)t)"""
    out = (C.c_ulonglong * 3)()
    rc = load_library().lpcnet_hip_quant_sweep_device(out)
    if rc:
        raise LPCNetError(f"lpcnet_hip_quant_sweep_device failed ({rc}): " + last_error())
    return int(out[0]), int(out[1]), int(out[2])


def arith_identities(a: np.ndarray, b: np.ndarray):
    """Bit patterns of (mfma products, v_mul products, packed mul/add halves, scalar mul/add) for operand arrays a, b (float32, len % 64 == 0)."""
    a = np.ascontiguousarray(a, np.float32); b = np.ascontiguousarray(b, np.float32)
    n = a.size
    outs = [np.empty((n, 4), np.uint32) for _ in range(4)]
    rc = load_library().lpcnet_hip_arith_identities_device(a, b, *outs, n)
    if rc:
        raise LPCNetError(f"lpcnet_hip_arith_identities_device failed ({rc}): " + last_error())
    return outs


class StreamState(C.Structure):
    """Raw per-stream record = struct lpcn_stream_state (lpcnet_amd/csrc/lpcnet_engine.h)."""
    _fields_ = [("gru_a", C.c_float * 384), ("gru_b", C.c_float * 16),
                ("conv1_mem", C.c_float * 168), ("conv2_mem", C.c_float * 256),
                ("old_lpc", C.c_float * 32), ("last_sig", C.c_float * 16),
                ("deemph_mem", C.c_float), ("last_exc", C.c_int32), ("frame_count", C.c_int32),
                ("rng", C.c_uint32 * 4), ("lpc", C.c_float * 16), ("pad", C.c_int32 * 3)]


class LPCNetState:
    """lpcnet_create / lpcnet_load_model / lpcnet_synthesize / lpcnet_destroy (include/lpcnet.h)."""

    def __init__(self, blob: bytes | None = None):
        self.L = load_library()
        self.p = self.L.lpcnet_create()
        self._blob = None
        if blob is not None:
            self.load_model(blob)

    def load_model(self, blob: bytes) -> int:
        self._blob = blob
        ret = self.L.lpcnet_load_model(self.p, blob, len(blob))
        if ret != 0:
            raise LPCNetError("lpcnet_load_model failed: " + last_error())
        return ret

    def reset(self):
        self.L.lpcnet_reset(self.p)

    def synthesize(self, features: np.ndarray, n: int = LPCNET_FRAME_SIZE) -> np.ndarray:
        out = np.zeros(n, np.int16)
        self.L.lpcnet_synthesize(self.p, np.ascontiguousarray(features[:NB_FEATURES], np.float32), out, n)
        return out

    # ---- the reference's internal entry points (PLC-facing, src/lpcnet_private.h:125-132)
    def reset_signal(self):
        self.L.lpcnet_reset_signal(self.p)

    def run_frame_network(self, features: np.ndarray):
        ga, gb, lpc = np.zeros(1152, np.float32), np.zeros(48, np.float32), np.zeros(16, np.float32)
        self.L.run_frame_network(self.p, ga, gb, lpc, np.ascontiguousarray(features[:NB_FEATURES], np.float32))
        return ga, gb, lpc

    def run_frame_network_deferred(self, features: np.ndarray):
        self.L.run_frame_network_deferred(self.p, np.ascontiguousarray(features[:NB_FEATURES], np.float32))

    def run_frame_network_flush(self):
        self.L.run_frame_network_flush(self.p)

    def synthesize_tail_impl(self, n: int = LPCNET_FRAME_SIZE, preload_pcm=None) -> np.ndarray:
        out = np.zeros(n, np.int16)
        k = 0 if preload_pcm is None else len(preload_pcm)
        out[:k] = preload_pcm if k else 0
        self.L.lpcnet_synthesize_tail_impl(self.p, out, n, k)
        return out

    def synthesize_impl(self, features: np.ndarray, n: int = LPCNET_FRAME_SIZE, preload_pcm=None) -> np.ndarray:
        out = np.zeros(n, np.int16)
        k = 0 if preload_pcm is None else len(preload_pcm)
        out[:k] = preload_pcm if k else 0
        self.L.lpcnet_synthesize_impl(self.p, np.ascontiguousarray(features[:NB_FEATURES], np.float32), out, n, k)
        return out

    def raw_bytes(self) -> bytes:
        return C.string_at(self.p, self.L.lpcnet_get_size())

    def load_raw_bytes(self, raw: bytes):
        C.memmove(self.p, raw, self.L.lpcnet_get_size())

    def __del__(self):
        try:
            self.L.lpcnet_destroy(self.p)
        except Exception:
            pass


class LPCNetDecState:
    """lpcnet_decoder_create / lpcnet_decode; blob=None: the process-default model (what the reference's demo relies on)."""

    def __init__(self, blob: bytes | None = None):
        self.L = load_library()
        self.p = self.L.lpcnet_decoder_create()
        self._blob = blob
        if blob is not None and self.L.lpcnet_hip_decoder_load_model(self.p, blob, len(blob)) != 0:
            raise LPCNetError("lpcnet_hip_decoder_load_model failed: " + last_error())

    def decode(self, packet: np.ndarray) -> np.ndarray:
        pcm = np.zeros(LPCNET_PACKET_SAMPLES, np.int16)
        if self.L.lpcnet_decode(self.p, np.ascontiguousarray(packet, np.uint8), pcm) != 0:
            raise LPCNetError("lpcnet_decode failed: " + last_error())
        return pcm

    def __del__(self):
        try:
            self.L.lpcnet_decoder_destroy(self.p)
        except Exception:
            pass


def set_default_model(blob: bytes):
    if load_library().lpcnet_hip_set_default_model(blob, len(blob)) != 0:
        raise LPCNetError("lpcnet_hip_set_default_model failed: " + last_error())


def shutdown():
    load_library().lpcnet_hip_shutdown()


def set_codebooks(cb1, cb2, cb3, cbd):
    load_library().lpcnet_hip_set_codebooks(*[np.ascontiguousarray(x, np.float32).reshape(-1) for x in (cb1, cb2, cb3, cbd)])


class LPCNetBatch:
    """n independent streams on one GPU, or sharded over `devices` (include/lpcnet_batch.h)."""

    def __init__(self, n_streams: int, blob: bytes, device: int = 0, devices=None):
        self.L = load_library()
        self.n = n_streams
        if devices is None:
            self.p = self.L.lpcnet_batch_create(n_streams, device)
        else:
            arr = (C.c_int * len(devices))(*devices)
            self.p = self.L.lpcnet_batch_create_sharded(n_streams, arr, len(devices))
        if not self.p:
            raise LPCNetError("lpcnet_batch_create failed: " + last_error())
        if self.L.lpcnet_batch_load_model(self.p, blob, len(blob)) != 0:
            err = last_error()
            self.L.lpcnet_batch_destroy(self.p)
            self.p = None
            raise LPCNetError("lpcnet_batch_load_model failed: " + err)

    def _chk(self, rc, what):
        if rc != 0:
            raise LPCNetError(f"{what} failed ({rc}): " + last_error())

    def close(self):
        if getattr(self, "p", None):
            self.L.lpcnet_batch_destroy(self.p)
            self.p = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def reset(self, first=0, count=None):
        self._chk(self.L.lpcnet_batch_reset(self.p, first, self.n - first if count is None else count), "reset")

    def synthesize(self, features: np.ndarray, preload_pcm: np.ndarray | None = None, preload: int = 160) -> np.ndarray:
        """features (n, T, stride>=20) float32 -> pcm (n, T*160) int16."""
        n, T, stride = features.shape
        assert n == self.n
        f = np.ascontiguousarray(features, np.float32)
        if preload_pcm is None:
            pcm = np.zeros((n, T * 160), np.int16)
            self._chk(self.L.lpcnet_batch_synthesize(self.p, f.reshape(-1), stride, pcm.reshape(-1), T), "synthesize")
        else:
            pcm = np.ascontiguousarray(preload_pcm, np.int16).copy()
            self._chk(self.L.lpcnet_batch_synthesize_preload(self.p, f.reshape(-1), stride, pcm.reshape(-1), T, preload), "synthesize_preload")
        return pcm

    def synthesize_step(self, features: np.ndarray, pcm: np.ndarray, n_samples, preload, mode) -> np.ndarray:
        """one frame step with per-stream (mode, n_samples, preload): include/lpcnet_batch.h lpcnet_batch_synthesize_step;
        features (n, >=20), pcm (n, 160) int16 (in: the imposed samples; out: the synthesised ones)."""
        f = np.ascontiguousarray(features, np.float32)
        out = np.ascontiguousarray(pcm, np.int16).copy()
        ns, pr, md = (np.ascontiguousarray(x, np.int32) for x in (n_samples, preload, mode))
        assert f.shape[0] == self.n and out.shape == (self.n, LPCNET_FRAME_SIZE) and ns.size == pr.size == md.size == self.n
        self._chk(self.L.lpcnet_batch_synthesize_step(self.p, f.ctypes.data_as(C.c_void_p), f.shape[1], out.ctypes.data_as(C.c_void_p),
                                                      ns.ctypes.data_as(C.c_void_p), pr.ctypes.data_as(C.c_void_p), md.ctypes.data_as(C.c_void_p)),
                  "lpcnet_batch_synthesize_step")
        return out

    def synthesize_device(self, d_features_ptr: int, stride: int, d_pcm_ptr: int, n_frames: int, hip_stream: int = 0):
        self._chk(self.L.lpcnet_batch_synthesize_device(self.p, d_features_ptr, stride, d_pcm_ptr, n_frames, hip_stream or None), "synthesize_device")

    def sync(self):
        self._chk(self.L.lpcnet_batch_sync(self.p), "sync")

    @property
    def shards(self):
        """[(first, count, device)] of every shard"""
        out = []
        for k in range(self.L.lpcnet_batch_shards(self.p)):
            a, b, c = C.c_int(), C.c_int(), C.c_int()
            self._chk(self.L.lpcnet_batch_shard_info(self.p, k, C.byref(a), C.byref(b), C.byref(c)), "shard_info")
            out.append((a.value, b.value, c.value))
        return out

    def synthesize_device_shard(self, shard: int, d_features_ptr: int, stride: int, d_pcm_ptr: int, n_frames: int, hip_stream: int = 0):
        self._chk(self.L.lpcnet_batch_synthesize_device_shard(self.p, shard, d_features_ptr, stride, d_pcm_ptr, n_frames, hip_stream or None),
                  "synthesize_device_shard")

    def set_lpc_gamma(self, gamma: float):
        self._chk(self.L.lpcnet_batch_set_lpc_gamma(self.p, gamma), "set_lpc_gamma")

    def set_end2end(self, on: bool = True):
        self._chk(self.L.lpcnet_batch_set_end2end(self.p, int(on)), "set_end2end")

    def set_fast(self, on=True):
        """FAST arithmetic (FMA / int32 accumulation); default is PARITY (bit-exact).  on = 2: FAST with the dual FC in fp16."""
        self._chk(self.L.lpcnet_batch_set_fast(self.p, int(on)), "set_fast")

    def decode_device(self, d_packets_ptr: int, d_pcm_ptr: int, n_packets: int, hip_stream: int = 0):
        self._chk(self.L.lpcnet_batch_decode_device(self.p, d_packets_ptr, d_pcm_ptr, n_packets, hip_stream or None), "decode_device")

    def decode(self, packets: np.ndarray) -> np.ndarray:
        n, P, _ = packets.shape
        pcm = np.zeros((n, P * 640), np.int16)
        self._chk(self.L.lpcnet_batch_decode(self.p, np.ascontiguousarray(packets, np.uint8).reshape(-1), pcm.reshape(-1), P), "decode")
        return pcm

    def run_tail(self, cond_a, cond_b, lpc, preload_pcm=None, preload=0) -> np.ndarray:
        n, T, _ = cond_a.shape
        pcm = np.zeros((n, T * 160), np.int16) if preload_pcm is None else np.ascontiguousarray(preload_pcm, np.int16).copy()
        self._chk(self.L.lpcnet_batch_run_tail(self.p, np.ascontiguousarray(cond_a, np.float32).reshape(-1),
                                               np.ascontiguousarray(cond_b, np.float32).reshape(-1),
                                               np.ascontiguousarray(lpc, np.float32).reshape(-1), pcm.reshape(-1), T, preload), "run_tail")
        return pcm

    def run_frames(self, features: np.ndarray):
        n, T, stride = features.shape
        ca = np.zeros((n, T, 1152), np.float32)
        cb = np.zeros((n, T, 48), np.float32)
        lpc = np.zeros((n, T, 16), np.float32)
        self._chk(self.L.lpcnet_batch_run_frames(self.p, np.ascontiguousarray(features, np.float32).reshape(-1), stride,
                                                 ca.ctypes.data, cb.ctypes.data, lpc.ctypes.data, T), "run_frames")
        return ca, cb, lpc

    def get_state(self, stream: int) -> StreamState:
        st = StreamState()
        assert C.sizeof(st) == self.L.lpcnet_batch_state_size()
        self._chk(self.L.lpcnet_batch_get_raw_state(self.p, stream, C.byref(st)), "get_state")
        return st

    def set_state(self, stream: int, st: StreamState):
        self._chk(self.L.lpcnet_batch_set_raw_state(self.p, stream, C.byref(st)), "set_state")

    def tune(self):
        """measure the streams per workgroup now, on the engine's own stream (the enqueue-only device-pointer calls never do)"""
        self._chk(self.L.lpcnet_batch_tune(self.p), "tune")

    @property
    def streams_per_workgroup(self):
        return self.L.lpcnet_batch_get_streams_per_workgroup(self.p)

    @streams_per_workgroup.setter
    def streams_per_workgroup(self, s):
        self._chk(self.L.lpcnet_batch_set_streams_per_workgroup(self.p, s), "set_streams_per_workgroup")

    def enable_timing(self, on=True):
        self._chk(self.L.lpcnet_batch_enable_timing(self.p, int(on)), "enable_timing")

    def last_timing(self):
        a, b = C.c_float(), C.c_float()
        self._chk(self.L.lpcnet_batch_last_timing(self.p, C.byref(a), C.byref(b)), "last_timing")
        return a.value, b.value

    def debug_trace_alloc(self, n_samples):
        self._chk(self.L.lpcnet_batch_debug_trace(self.p, n_samples, None), "debug_trace")

    def debug_trace_fetch(self, n_samples):
        out = np.zeros((n_samples, 1600), np.float32)
        self._chk(self.L.lpcnet_batch_debug_trace(self.p, n_samples, out.ctypes.data), "debug_trace")
        return out

    def profile_reset(self):
        self._chk(self.L.lpcnet_batch_profile(self.p, None), "profile")

    def profile_fetch(self):
        out = np.zeros(96, np.uint64)
        self._chk(self.L.lpcnet_batch_profile(self.p, out.ctypes.data), "profile")
        return out
