#!/usr/bin/env python3
"""BASELINE.json configs 0 -> 1: `lpcnet_demo -synthesis` on ONE 10-second feature file -- the reference's own program,
built (a) from its own sources with AVX2 (oracle/_ref/lpcnet_demo_af float, _ai int8 = the reference's default) and
(b) unmodified against liblpcnet_hip.so (oracle/_ref/lpcnet_demo_hip).  Prints one JSON object with wall seconds
(process start, model load and file I/O included, best of 3) and real-time factors."""
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from lpcnet_amd import synth  # noqa: E402

T = 1000


def run(exe, flavour, repeat=3, frames=T):
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "weights_blob.bin"), "wb").write(synth.blob_bytes(synth.make_model(flavour=flavour)))
        synth.make_features(1000, T)[:frames].astype(np.float32).tofile(os.path.join(d, "feat.f32"))
        best = None
        for _ in range(repeat):
            t0 = time.perf_counter()
            subprocess.check_call([exe, "-synthesis", "feat.f32", "out.pcm"], cwd=d)
            dt = time.perf_counter() - t0
            best = dt if best is None or dt < best else best
        return best


def main():
    ref = os.path.join(ROOT, "oracle", "_ref")
    out = {"audio_seconds": T / 100.0, "what": "wall seconds of `lpcnet_demo -synthesis` on one 10-s file, best of 3, process start included"}
    for name, exe, flavour in (("reference AVX2 float (A-f), 1 core", "lpcnet_demo_af", "float"), ("reference AVX2 int8 (A-i), 1 core", "lpcnet_demo_ai", "int8"),
                               ("HIP engine, float blob", "lpcnet_demo_hip", "float"), ("HIP engine, int8 blob", "lpcnet_demo_hip", "int8")):
        path = os.path.join(ref, exe)
        if not os.path.exists(path):
            continue
        s = run(path, flavour)
        out[name] = {"seconds": round(s, 3), "real_time_factor": round(T / 100.0 / s, 2)}
        # the split: the same program on the first 20 frames -- the difference is 980 frames of steady state,
        # what is left of the short run is process start + HIP initialisation + model packing and upload
        s20 = run(path, flavour, frames=20)
        per_frame = (s - s20) / (T - 20)
        out[name].update({"seconds_20_frames": round(s20, 3), "ms_per_frame_steady": round(per_frame * 1e3, 4), "startup_seconds": round(s20 - 20 * per_frame, 3),
                          "real_time_factor_steady": round(0.01 / per_frame, 2)})
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
