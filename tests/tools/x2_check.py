#!/usr/bin/env python3
"""Bring-up check of the two-group sample kernel (eight streams per workgroup) on a GPU box: PCM and final state against the CPU oracle
for a few stream counts, then its kernel time beside the four-stream kernel's.   python tests/tools/x2_check.py [frames] [timing streams]"""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from lpcnet_amd import synth, api
from oracle import orc


def main():
    T = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    nt = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
    kw = {}
    if os.environ.get('X2_SKEW'): kw['skew'] = float(os.environ['X2_SKEW'])                # trained-like sparsity: the streamed-item variants
    if os.environ.get('X2_DENS'): kw['densities'] = tuple(float(x) for x in os.environ['X2_DENS'].split(','))
    blob = synth.blob_bytes(synth.make_model(**kw))
    om = orc.OracleModel(blob)
    bad = 0
    for n in (() if os.environ.get('X2_SKIP_PARITY') else (8, 16, 13, 5)):
        feats = np.stack([synth.make_features(1000 + s, T) for s in range(n)])
        ref = np.stack([om.new_state().synthesize(feats[s]) for s in range(n)])
        b = api.LPCNetBatch(n, blob)
        b.streams_per_workgroup = 8
        t0 = time.time()
        pcm = b.synthesize(feats)
        d = np.nonzero(pcm != ref)
        print("n=%d S=%d: mismatching samples %d of %d, first %s (%.2fs)" % (n, b.streams_per_workgroup, d[0].size, pcm.size,
              (d[0][:4].tolist(), d[1][:4].tolist()) if d[0].size else None, time.time() - t0), flush=True)
        bad += d[0].size
        # a second call continues the streams (state round trip through global memory)
        feats2 = np.stack([synth.make_features(2000 + s, 3) for s in range(n)])
        refs = []
        for s in range(n):
            st = om.new_state(); st.synthesize(feats[s]); refs.append(st.synthesize(feats2[s]))
        pcm2 = b.synthesize(feats2)
        d2 = int((pcm2 != np.stack(refs)).sum())
        print("      continued over a second call: mismatches %d" % d2, flush=True)
        bad += d2
    if bad:
        print("X2 PARITY FAILED")
    # timing
    Tt = 10
    feats = np.stack([synth.make_features(1000 + (s % 64), Tt) for s in range(nt)])
    for S in ((8,) if os.environ.get('X2_SKIP_PARITY') else (4, 8)):
        b = api.LPCNetBatch(nt, blob)
        b.streams_per_workgroup = S
        b.enable_timing(True)
        b.synthesize(feats)
        best = 1e9
        for _ in range(3):
            b.synthesize(feats)
            best = min(best, b.last_timing()[0])
        print("timing n=%d S=%d: sample kernel %.2f ms -> %.1f M samples/s" % (nt, S, best, nt * Tt * 160 / best / 1e3), flush=True)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
