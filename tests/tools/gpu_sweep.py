#!/usr/bin/env python3
"""Throughput sweep over (streams, streams-per-workgroup); checks a few streams against the oracle."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from lpcnet_amd import synth, api
from oracle import orc

def main():
    T = int(sys.argv[1]) if len(sys.argv) > 1 else 22
    configs = [(1, 1), (256, 1), (512, 2), (1024, 4), (1024, 2), (1024, 1), (2048, 4), (4096, 4)]
    if len(sys.argv) > 2:
        configs = [tuple(int(x) for x in c.split(":")) for c in sys.argv[2].split(",")]
    flavour = os.environ.get("LPCN_FLAVOUR", "float")
    blob = synth.blob_bytes(synth.make_model(flavour=flavour))
    print("flavour", flavour)
    om = orc.OracleModel(blob)
    base = np.stack([synth.make_features(1000 + s, T) for s in range(8)])
    ref = np.stack([om.new_state().synthesize(base[s]) for s in range(8)])
    for n, S in configs:
        feats = np.ascontiguousarray(base[np.arange(n) % 8])
        b = api.LPCNetBatch(n, blob)
        b.streams_per_workgroup = S
        if os.environ.get("LPCN_FAST"):
            b.set_fast(True)
        b.enable_timing(True)
        pcm = b.synthesize(feats)
        ok = np.array_equal(pcm, ref[np.arange(n) % 8])
        b.reset()
        b.profile_reset()
        t0 = time.time(); pcm = b.synthesize(feats); wall = time.time() - t0
        prof = b.profile_fetch().astype(np.float64) / ((T - 2) * 160)
        ms_s, ms_f = b.last_timing()
        live = n * (T - 2) * 160
        print(f"n={n:5d} S={S} exact={ok} sample_kernel={ms_s:8.2f} ms frame_kernels={ms_f:6.2f} ms wall={wall*1e3:8.1f} ms "
              f"-> {live/(ms_s*1e-3)/1e6:8.2f} M samples/s (kernel), us/step={ms_s*1e3/((T-2)*160):6.2f} ", flush=True)
        if prof.any():
            print("   per-wave clk/step [B1wait P2 P3tail P4 P5 | gather close fcpre gruB items start P5a]:")
            for w in range(8): print("   wave", w, " ".join("%6.0f" % x for x in prof[w*12:w*12+12]), flush=True)
        b.close()

if __name__ == "__main__":
    main()
