#!/usr/bin/env python3
"""Bring-up check on a GPU box: HIP engine vs the CPU oracle, seam by seam."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from lpcnet_amd import synth, api
from oracle import orc

def main():
    T = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    S = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    blob = synth.blob_bytes(synth.make_model())
    om = orc.OracleModel(blob)
    feats = np.stack([synth.make_features(1000 + s, T) for s in range(n)])
    # oracle: full run, recording frame products
    ref_pcm = np.zeros((n, T * 160), np.int16)
    ca = np.zeros((n, T, 1152), np.float32); cb = np.zeros((n, T, 48), np.float32); lp = np.zeros((n, T, 16), np.float32)
    t0 = time.time()
    for s in range(n):
        st = om.new_state()
        for t in range(T):
            frame = ref_pcm[s, t*160:(t+1)*160]
            st.L.orc_synthesize(st.p, np.ascontiguousarray(feats[s, t, :20]), frame, 160, 0)
            lp[s, t], ca[s, t], cb[s, t] = st.frame_products()
    print("oracle %.1fs" % (time.time() - t0))
    b = api.LPCNetBatch(n, blob)
    if S: b.streams_per_workgroup = S
    print("streams/wg", b.streams_per_workgroup)
    # 1. frames seam
    gca, gcb, glp = b.run_frames(feats)
    print("frames: cond_a mismatches", int((gca != ca).sum()), "cond_b", int((gcb != cb).sum()), "lpc", int((glp != lp).sum()),
          "max|dlpc|", float(np.abs(glp - lp).max()))
    # 2. tail seam: needs frame_count > 2 => set via state; emulate by running frames first (frame_count advanced to T)
    b.reset()
    st = b.get_state(0)
    for s in range(n):
        st = b.get_state(s); st.frame_count = 3; b.set_state(s, st)
    # oracle tail with same convention: all frames live
    tail_ref = np.zeros((n, T * 160), np.int16)
    for s in range(n):
        o = om.new_state(); o.L.orc_force_frame_count(o.p, 3)
        for t in range(T):
            fr = tail_ref[s, t*160:(t+1)*160]
            o.L.orc_synthesize_tail(o.p, ca[s, t], cb[s, t], lp[s, t], fr, 160, 0)
    b.debug_trace_alloc(T * 160)
    tail = b.run_tail(ca, cb, lp)
    d = np.nonzero(tail != tail_ref)
    print("tail: mismatching samples", d[0].size, "first", (d[0][:3], d[1][:3]) if d[0].size else None)
    if d[0].size:
        tr = b.debug_trace_fetch(T * 160)
        # compare trace of stream 0 against oracle step by step
        o = om.new_state(); o.L.orc_force_frame_count(o.p, 3)
        for t in range(T):
            fr = np.zeros(160, np.int16)
            for i in range(160):
                pass
        print("trace[0:3] exc,sig,pred,pcm,pred:", tr[:3, 400:405])
        print("tail gpu", tail[0, :12], "ref", tail_ref[0, :12])
    b.debug_trace_alloc(0)
    # 3. end to end
    b.reset()
    pcm = b.synthesize(feats)
    d = np.nonzero(pcm != ref_pcm)
    print("e2e: mismatching samples", d[0].size, "of", pcm.size, "first", (d[0][:3], d[1][:3]) if d[0].size else None)
    # timing
    b.enable_timing(True); b.reset(); b.synthesize(feats)
    print("timing ms (sample, frame):", b.last_timing(), "samples/s: %.3g" % (n * T * 160 / (b.last_timing()[0] * 1e-3)))
    return 0

if __name__ == "__main__":
    sys.exit(main())
