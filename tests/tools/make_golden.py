#!/usr/bin/env python3
"""Generate tests/golden/golden_v1.npz by running the REAL reference (oracle/_ref, compiled from
/root/reference/src by oracle/Makefile) on the seeded synthetic model/features.  The reference
holds no golden vectors for this path, so these fixtures are its pinned outputs:
they travel to the GPU box, where /root/reference does not exist.

    make -C oracle ref && python tests/tools/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from lpcnet_amd import synth  # noqa: E402
from oracle import ref  # noqa: E402

T = 60            # frames per stream in the fixture (58 live frames = 9280 samples)
SEEDS = (1000, 1001, 1002)


def main():
    out = {}
    blob_f = synth.blob_bytes(synth.make_model(flavour="float"))
    blob_i = synth.blob_bytes(synth.make_model(flavour="int8"))
    gf, gi = ref.RefLib("gf"), ref.RefLib("gi")
    out["seeds"] = np.array(SEEDS)
    out["n_frames"] = np.array(T)
    for seed in SEEDS:
        f = synth.make_features(seed, T)
        st = gf.new_state(blob_f)
        pcm = np.zeros(T * 160, np.int16)
        lpcs, ca_sum, cb = [], [], []
        for t in range(T):
            frame = pcm[t * 160:(t + 1) * 160]
            gf.lib.lpcnet_synthesize(st.p, np.ascontiguousarray(f[t, :20]), frame, 160)
            lpc, ca, cbv = st.frame_products()
            lpcs.append(lpc); cb.append(cbv)
            ca_sum.append(np.array([ca[::97].astype(np.float64).sum(), np.abs(ca).astype(np.float64).sum()]))
        out[f"pcm_gf_{seed}"] = pcm
        out[f"lpc_gf_{seed}"] = np.stack(lpcs)
        out[f"condb_gf_{seed}"] = np.stack(cb)
        out[f"conda_sums_gf_{seed}"] = np.stack(ca_sum)
        c1, c2, ga, gb = st.nnet_state()
        out[f"gru_a_gf_{seed}"] = ga
        out[f"gru_b_gf_{seed}"] = gb
        out[f"pcm_gi_{seed}"] = gi.synthesize_file(blob_i, f)
    # full conditioning vectors of a few frames (seed 1000)
    st = gf.new_state(blob_f)
    f = synth.make_features(1000, T)
    cas = []
    for t in range(8):
        frame = np.zeros(160, np.int16)
        gf.lib.lpcnet_synthesize(st.p, np.ascontiguousarray(f[t, :20]), frame, 160)
        cas.append(st.frame_products()[1])
    out["conda_gf_1000_first8"] = np.stack(cas)
    # teacher forcing (lpcnet_synthesize_impl preload, src/lpcnet.c:256-259): drive the loop with a given signal
    tt = np.arange(20 * 160)
    forced = (3000 * np.sin(tt * 0.05) + 800 * np.sin(tt * 0.31 + 1.0)).astype(np.int16)
    st = gf.new_state(blob_f)
    f = synth.make_features(1000, 20)
    st.synthesize(f, preload_pcm=forced)
    out["forced_pcm_in"] = forced
    out["forced_gru_a"], out["forced_gru_b"] = st.nnet_state()[2], st.nnet_state()[3]
    ls, le, dm, fc, rng = st.signal_state()
    out["forced_last_sig"], out["forced_last_exc"], out["forced_deemph"], out["forced_rng"] = ls, np.array(le), np.array(dm, np.float32), rng
    # half-forced frames: first 80 samples of each frame forced, rest sampled
    st = gf.new_state(blob_f)
    half = np.zeros(20 * 160, np.int16)
    for t in range(20):
        frame = half[t * 160:(t + 1) * 160]
        frame[:80] = forced[t * 160:t * 160 + 80]
        gf.lib.ref_synthesize_impl(st.p, np.ascontiguousarray(f[t, :20]), frame, 160, 80)
    out["half_forced_pcm"] = half
    # scalar known-answer tables
    xs = np.concatenate([np.linspace(-40000, 40000, 4001), np.linspace(-3, 3, 601), [0.0, -0.0, 1e-8, 32767.0, -32768.0]]).astype(np.float32)
    out["ulaw_x"] = xs
    out["lin2ulaw"] = np.array([gf.lib.ref_lin2ulaw(float(x)) for x in xs], np.int32)
    out["ulaw2lin"] = np.array([gf.lib.ref_ulaw2lin(float(u)) for u in range(256)], np.float32)
    ax = np.concatenate([np.linspace(-12, 12, 4801), [0.0, -0.0, 8.0, -8.0, 100.0]]).astype(np.float32)
    th = np.zeros_like(ax); sg = np.zeros_like(ax)
    gf.lib.ref_vec_tanh(th, ax, ax.size); gf.lib.ref_vec_sigmoid(sg, ax, ax.size)
    out["act_x"], out["tanh"], out["sigmoid"] = ax, th, sg
    rng = np.zeros(4, np.uint32)
    gf.lib.kiss99_srand(rng, b"LPCNet", 6)
    out["kiss99_seeded"] = rng.copy()
    out["kiss99_first32"] = np.array([gf.lib.kiss99_rand(rng) for _ in range(32)], np.uint32)
    table = np.zeros(256, np.float32)
    gf.lib.ref_get_logit_table(gf.new_state(blob_f).p, table)
    out["logit_table"] = table
    # lpc_from_cepstrum on random cepstra
    r = np.random.default_rng(7)
    ceps = (r.standard_normal((64, 18)) * np.array([3.0] + [1.0] * 17)).astype(np.float32)
    lp = np.zeros((64, 16), np.float32)
    for i in range(64):
        gf.lib.ref_lpc_from_cepstrum(lp[i], ceps[i])
    out["lpc_ceps_in"], out["lpc_ceps_out"] = ceps, lp
    # codec front-end with seeded codebooks
    cbs = synth.make_codebooks(5)
    gf.lib.ref_set_codebooks(*cbs)
    pk = r.integers(0, 256, size=(12, 8), dtype=np.uint8)
    feats = np.zeros((12, 4, 36), np.float32); vq = np.zeros(18, np.float32)
    for i in range(12):
        gf.lib.ref_decode_packet(feats[i].reshape(-1), vq, pk[i])
    out["packets"], out["packet_features"] = pk, feats
    dec = gf.lib.lpcnet_decoder_create()
    import ctypes as C
    buf = C.create_string_buffer(blob_f, len(blob_f))
    gf.lib.lpcnet_load_model(dec, buf, len(blob_f))
    pcm = np.zeros((12, 640), np.int16)
    for i in range(12):
        gf.lib.lpcnet_decode(dec, pk[i], pcm[i])
    out["packet_pcm_gf"] = pcm
    path = os.path.join(ROOT, "tests", "golden", "golden_v1.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
