#!/usr/bin/env python3
"""tests/golden/simd_envelope_v1.json: how far the REFERENCE'S OWN SIMD builds (oracle/_ref af = AVX2 float, ai = AVX2
int8) drift from its generic-C builds (gf, gi) when both are teacher-forced with the same signal
(lpcnet_synthesize_impl preload, src/lpcnet.c:256-259) on the seeded synthetic model: per-frame max|dGRU-A| and
max|dGRU-B| over 4 streams x 200 frames.  This is the tolerance the FAST flavour of the HIP engine is held to
(tests/test_gpu_fast.py): a FAST kernel is "as good as the reference's own SIMD ports" if it stays inside it.

    make -C oracle ref && python tests/tools/make_envelope.py
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from lpcnet_amd import synth  # noqa: E402
from oracle import ref  # noqa: E402

T = 200
SEEDS = (8800, 8801, 8802, 8803)


def main():
    out = {"frames": T, "seeds": list(SEEDS), "metric": "per-frame max |state difference|, frames >= 3"}
    for g_name, s_name, flavour in (("gf", "af", "float"), ("gi", "ai", "int8")):
        blob = synth.blob_bytes(synth.make_model(flavour=flavour))
        G, Sx = ref.RefLib(g_name), ref.RefLib(s_name)
        da, db = [], []
        for seed in SEEDS:
            f = synth.make_features(seed, T)
            pcm = G.new_state(blob).synthesize(f)            # free run of the generic build = the forcing signal
            sg, ss = G.new_state(blob), Sx.new_state(blob)
            for t in range(T):
                fr = np.ascontiguousarray(f[t, :20], np.float32)
                x = pcm[t * 160:(t + 1) * 160].copy(); G.lib.ref_synthesize_impl(sg.p, fr, x, 160, 160)
                y = pcm[t * 160:(t + 1) * 160].copy(); Sx.lib.ref_synthesize_impl(ss.p, fr, y, 160, 160)
                _, _, ga1, gb1 = sg.nnet_state(); _, _, ga2, gb2 = ss.nnet_state()
                if t >= 3:
                    da.append(float(np.abs(ga1 - ga2).max())); db.append(float(np.abs(gb1 - gb2).max()))
        da, db = np.array(da), np.array(db)
        out[flavour] = {"builds": f"{g_name} vs {s_name}",
                        "gru_a": {"median": float(np.median(da)), "p99": float(np.percentile(da, 99)), "worst": float(da.max())},
                        "gru_b": {"median": float(np.median(db)), "p99": float(np.percentile(db, 99)), "worst": float(db.max())}}
        print(flavour, out[flavour])
    json.dump(out, open(os.path.join(ROOT, "tests", "golden", "simd_envelope_v1.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
