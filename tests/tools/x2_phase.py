#!/usr/bin/env python3
"""Phase table of the two-group sample kernel from its in-kernel shader-clock accounting (profiling build:
python -m lpcnet_amd.build --prof; LPCNET_HIP_LIB=lpcnet_amd/liblpcnet_hip_prof.so python tests/tools/x2_phase.py [frames] [streams])."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from lpcnet_amd import synth, api

def main():
    T = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
    blob = synth.blob_bytes(synth.make_model())
    feats = np.stack([synth.make_features(1000 + (s % 16), T) for s in range(n)])
    for S in (8, 4):
        b = api.LPCNetBatch(n, blob)
        b.streams_per_workgroup = S
        b.enable_timing(True)
        b.synthesize(feats)
        b.reset(); b.profile_reset()
        b.synthesize(feats)
        ms = b.last_timing()[0]
        prof = b.profile_fetch().astype(np.float64)
        steps = T * 160
        print(f"n={n} S={S} sample_kernel={ms:.2f} ms -> {n*steps/ms/1e3:.1f} M samples/s, us per sample step of a workgroup = {ms*1e3/steps:.2f}")
        if S == 8:
            print("   per-wave clk per HALF-step [lead chain|idxwait gatesB|p0issue heads start items close B1wait P2 tree B2wait p0wait | sum]:")
            for w in range(8):
                r = prof[w*12:w*12+12] / (2 * steps)
                print("   wave", w, " ".join("%6.0f" % x for x in r), "| %6.0f" % r.sum())
        else:
            print("   per-wave clk/step [B1wait P2 P3tail P4 P5 | gather close fcpre gruB items start P5a]:")
            for w in range(8): print("   wave", w, " ".join("%6.0f" % x for x in prof[w*12:w*12+12] / steps))
        b.close()

if __name__ == "__main__":
    main()
