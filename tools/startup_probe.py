#!/usr/bin/env python3
"""Where a fresh process spends its time before the first sample of a single stream comes out (the demo's 0.25-0.3 s start-up).
Run in a NEW process per flavour:  python tools/startup_probe.py [--int8]  -> one JSON line, milliseconds per stage."""
import json
import os
import sys
import time

t_proc = time.perf_counter()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ctypes as C                                          # noqa: E402
import numpy as np                                          # noqa: E402


def main():
    from lpcnet_amd import api, synth
    fl = "int8" if "--int8" in sys.argv else "float"
    blob = synth.blob_bytes(synth.make_model(flavour=fl))
    feats = synth.make_features(7, 20)[None]
    st = {}
    t = time.perf_counter()
    L = api.load_library()                                  # dlopen of the engine (+ libamdhip64)
    st["dlopen_library"] = time.perf_counter() - t
    f = L.lpcnet_hip_check_model
    f.restype = C.c_int; f.argtypes = [C.c_char_p, C.c_int, C.c_void_p]
    t = time.perf_counter(); f(blob, len(blob), None); st["host_parse_pack_selftest (also part of load_model)"] = time.perf_counter() - t
    t = time.perf_counter()
    p = L.lpcnet_batch_create(1, 0)                         # HIP runtime init, device, stream
    st["batch_create (HIP runtime init)"] = time.perf_counter() - t
    t = time.perf_counter()
    rc = L.lpcnet_batch_load_model(p, blob, len(blob))      # parse + pack + upload
    assert rc == 0
    st["load_model (pack + upload)"] = time.perf_counter() - t
    pcm = np.zeros((1, 20 * 160), dtype=np.int16)
    for k in ("first synthesize of 20 frames (code object load + kernels)", "second synthesize of 20 frames"):
        t = time.perf_counter()
        rc = L.lpcnet_batch_synthesize(p, np.ascontiguousarray(feats.reshape(-1), dtype=np.float32), 36, pcm.reshape(-1), 20)
        assert rc == 0
        st[k] = time.perf_counter() - t
    L.lpcnet_batch_destroy(p)
    print(json.dumps({"flavour": fl, "ms": {k: round(v * 1e3, 2) for k, v in st.items()}, "python_imports_ms": round((time.perf_counter() - t_proc) * 1e3 - sum(st.values()) * 1e3, 1)}))


if __name__ == "__main__":
    main()
