#!/usr/bin/env python3
"""Generate tests/golden/golden_demo_v1.npz: what the REFERENCE'S OWN demo binaries (oracle/_ref/lpcnet_demo_gf and
lpcnet_demo_gi, compiled from /root/reference/src by oracle/Makefile) write for `-synthesis` on the 10-second /
1000-frame synthetic feature file of BASELINE.json configs 0/1 (seed 1000), with the float resp. int8 weights_blob.bin
in the working directory.  320 000-byte PCM files are kept as SHA-256 + one CRC-32 per frame (pin-points the first
diverging frame) + the first three and the last frame verbatim.

    make -C oracle ref && python tests/tools/make_golden_demo.py
"""
import hashlib
import os
import subprocess
import sys
import tempfile
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from lpcnet_amd import synth  # noqa: E402

T = 1000
SEED = 1000


def digest(pcm):
    frames = pcm.reshape(-1, 160)
    return {"sha256": np.frombuffer(hashlib.sha256(pcm.tobytes()).digest(), np.uint8),
            "crc": np.array([zlib.crc32(f.tobytes()) for f in frames], np.uint32),
            "head": pcm[:480].copy(), "tail": pcm[-160:].copy()}


def main():
    out = {"n_frames": np.array(T), "seed": np.array(SEED)}
    feats = synth.make_features(SEED, T)
    for fl, flavour in (("gf", "float"), ("gi", "int8")):
        exe = os.path.join(ROOT, "oracle", "_ref", f"lpcnet_demo_{fl}")
        with tempfile.TemporaryDirectory() as d:
            open(os.path.join(d, "weights_blob.bin"), "wb").write(synth.blob_bytes(synth.make_model(flavour=flavour)))
            feats.astype(np.float32).tofile(os.path.join(d, "feat.f32"))
            subprocess.check_call([exe, "-synthesis", "feat.f32", "out.pcm"], cwd=d)
            pcm = np.fromfile(os.path.join(d, "out.pcm"), np.int16)
        assert pcm.size == T * 160
        for k, v in digest(pcm).items():
            out[f"{fl}_{k}"] = v
        print(fl, "pcm std", float(pcm.std()), "max", int(np.abs(pcm.astype(np.int32)).max()))
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "golden_demo_v1.npz"), **out)


if __name__ == "__main__":
    main()
