"""Clocks from the tree barrier to the publication of the next sample's indices (wave 0), from the debug trace."""
import sys, numpy as np
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from lpcnet_amd import synth, api
for fl in ("float", "int8"):
    blob = synth.blob_bytes(synth.make_model(flavour=fl))
    n, T = 1024, 6
    base = np.stack([synth.make_features(1000 + s, T) for s in range(8)])
    feats = np.ascontiguousarray(base[np.arange(n) % 8])
    b = api.LPCNetBatch(n, blob)
    b.debug_trace_alloc(T * 160)
    b.synthesize(feats)
    tr = b.debug_trace_fetch(T * 160)
    v = tr[320 + 10:, 405]
    v = v[v > 0]
    print(fl, "leader barrier->publish clk: median %.0f  p10 %.0f p90 %.0f" % (np.median(v), np.percentile(v, 10), np.percentile(v, 90)))
    b.close()
