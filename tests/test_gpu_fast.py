"""FAST arithmetic flavour (lpcnet_batch_set_fast): fused multiply-add for float blobs, int32 block accumulation for int8
blobs -- what the reference's own SIMD builds do (src/vec_avx.h:690-858) instead of the generic-C order.  FAST is not
bit-identical to any reference build (those differ among themselves), and free-running synthesis is
chaotic, so it is validated the way SURVEY.md §8c prescribes: TEACHER-FORCED (lpcnet_synthesize_impl's preload,
src/lpcnet.c:256-259) against the bit-exact PARITY engine fed the same signal, frame by frame, and the per-frame state
deviation must stay inside the envelope the reference's own AVX2 builds show against its generic-C builds under the very
same protocol on the very same model: tests/golden/simd_envelope_v1.json, generated from oracle/_ref (af vs gf, ai vs gi)
by tests/tools/make_envelope.py.  (int8 blobs re-quantise the state every sample, so a last-bit difference flips
quantisation levels and the teacher-forced state trajectories drift apart by ~1e-2 -- for the reference's own int8
builds just as for FAST.)"""
import json
import os

import numpy as np
import pytest

from lpcnet_amd import api, synth

pytestmark = pytest.mark.gpu
ENVELOPE = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "simd_envelope_v1.json")))


def forced_states(blob, feats, pcm, fast, S):
    n, T, _ = feats.shape
    b = api.LPCNetBatch(n, blob)
    b.streams_per_workgroup = S
    b.set_fast(fast)
    ga, gb = np.zeros((T, n, 384), np.float32), np.zeros((T, n, 16), np.float32)
    for t in range(T):
        b.synthesize(np.ascontiguousarray(feats[:, t:t + 1]), preload_pcm=np.ascontiguousarray(pcm[:, t * 160:(t + 1) * 160]), preload=160)
        for s in range(n):
            st = b.get_state(s)
            ga[t, s] = np.array(st.gru_a, np.float32)
            gb[t, s] = np.array(st.gru_b, np.float32)
    b.close()
    return ga, gb


@pytest.mark.parametrize("flavour,S,kw", [("float", 4, {}), ("float", 2, {}), ("float", 1, {}), ("int8", 4, {}), ("int8", 2, {}),
                                          ("float", 4, dict(grub_density=0.4)), ("float", 2, dict(grub_density=0.4)),
                                          ("int8", 4, dict(grub_density=0.4))],
                         ids=["f32-S4", "f32-S2", "f32-S1", "int8-S4", "int8-S2", "f32-S4-sparseB", "f32-S2-sparseB", "int8-S4-sparseB"])
def test_fast_teacher_forced_inside_reference_simd_envelope(flavour, S, kw, hip_lib):
    """(S = 2 runs the matrix-pipe items with two dead accumulator columns; a block-sparse GRU-B switches the
    split / dense fast paths of GRU-B off; 16 streams x 100 frames = every lane position of four workgroups)"""
    env = ENVELOPE[flavour]
    p99_a, worst_a, worst_b = env["gru_a"]["p99"], env["gru_a"]["worst"], env["gru_b"]["worst"]
    blob = synth.blob_bytes(synth.make_model(flavour=flavour, **kw))
    n, T = 16, 100
    feats = np.stack([synth.make_features(8800 + s, T) for s in range(n)])
    ref = api.LPCNetBatch(n, blob)
    pcm = ref.synthesize(feats)                              # the signal both engines are driven with
    ref.close()
    ga_p, gb_p = forced_states(blob, feats, pcm, False, S)
    ga_f, gb_f = forced_states(blob, feats, pcm, True, S)
    live = slice(3, T)                                       # frames that produce samples (two start-up frames)
    da = np.abs(ga_f - ga_p)[live].max(axis=2).reshape(-1)   # per (frame, stream) maximum
    db = np.abs(gb_f - gb_p)[live].max(axis=2).reshape(-1)
    assert np.abs(ga_p[live]).max() > 0.3                    # the states are alive
    assert da.max() > 0 or flavour == "float"                # FAST is a different arithmetic (float FMA may round alike on a grid model)
    # (100 frames here against 200 in the envelope: 1.25 x covers the sampling noise of a p99 / maximum of a chaotic quantity)
    assert np.percentile(da, 99) <= 1.25 * p99_a and da.max() <= 1.25 * worst_a, (np.percentile(da, 99), da.max(), env)
    assert db.max() <= 1.25 * worst_b, (db.max(), env)
    if flavour == "float":                                  # FMA only changes last bits: far inside the AVX2 build's own drift
        assert da.max() <= 0.1 * p99_a
    # forcing all 160 samples must return them untouched, exactly like PARITY does
    b = api.LPCNetBatch(n, blob)
    b.set_fast(True)
    out = b.synthesize(feats, preload_pcm=pcm, preload=160)
    assert np.array_equal(out[:, 320:], pcm[:, 320:]) and np.all(out[:, :320] == 0)
    b.close()


@pytest.mark.parametrize("flavour", ["float", "int8"])
def test_fast_free_running_is_sane_and_switchable(flavour, hip_lib):
    """free-running FAST output: same start-up behaviour, same signal statistics as PARITY; switching FAST off restores
    bit-exact PARITY on the same batch object"""
    blob = synth.blob_bytes(synth.make_model(flavour=flavour))
    n, T = 8, 40
    feats = np.stack([synth.make_features(8900 + s, T) for s in range(n)])
    b = api.LPCNetBatch(n, blob)
    parity = b.synthesize(feats)
    b.reset()
    b.set_fast(True)
    fast = b.synthesize(feats)
    assert np.all(fast[:, :320] == 0)
    rp, rf = parity[:, 320:].astype(np.float64).std(), fast[:, 320:].astype(np.float64).std()
    assert 0.7 < rf / rp < 1.4, (rp, rf)
    assert np.array_equal(fast[:, 320:480], parity[:, 320:480]) or np.abs(fast[:, 320:480].astype(np.int32) - parity[:, 320:480]).max() < 2000
    b.reset()
    b.set_fast(False)
    assert np.array_equal(b.synthesize(feats), parity)
    b.close()


def _tree_decisions(blob, feats, pcm, fast):
    """teacher-forced run of ONE stream; the sampler's own decision per sample (the trace keeps it although the forced
    signal overrides it) and the GRU-B state it was drawn from"""
    T = feats.shape[1]
    b = api.LPCNetBatch(1, blob)
    b.set_fast(fast)
    b.debug_trace_alloc(T * 160)
    b.synthesize(feats, preload_pcm=pcm, preload=160)
    tr = b.debug_trace_fetch(T * 160)
    b.debug_trace_alloc(0)
    b.close()
    return tr[320:, 406].astype(np.int32), tr[320:, 384:400]


@pytest.mark.parametrize("flavour", ["float", "int8"])
def test_fp16_dual_fc_decision_flips(flavour, hip_lib):
    """BASELINE.json config 4 names an fp16 dual FC.  It is a FAST sub-option (lpcnet_batch_set_fast(b, 2)); the reference
    has nothing to pin it to, so it is measured against FAST with the fp32 tree, teacher-forced on the same signal (same
    GRU-B states up to FAST's own last-bit effects, same thresholds): the tree decision may change only where a logit
    sits within fp16 rounding of its threshold.  For scale, the share of decisions FAST itself changes against PARITY is
    measured the same way."""
    blob = synth.blob_bytes(synth.make_model(flavour=flavour))
    T = 60
    feats = synth.make_features(9100, T)[None]
    ref = api.LPCNetBatch(1, blob)
    pcm = ref.synthesize(feats)
    ref.close()
    d_par, hb_par = _tree_decisions(blob, feats, pcm, 0)
    d_fast, hb_fast = _tree_decisions(blob, feats, pcm, 1)
    d_f16, hb_f16 = _tree_decisions(blob, feats, pcm, 2)
    n = d_par.size
    assert n == (T - 2) * 160 and np.unique(d_par).size > 20           # a live sampler, not a stuck one
    assert np.array_equal(hb_fast, hb_f16)                               # the fp16 tree changes nothing upstream of the tree
    flips_fast = float((d_fast != d_par).mean())
    flips_f16 = float((d_f16 != d_fast).mean())
    # fp16 weights and state perturb a logit by ~1e-3 of its range: a few per cent of the decisions sit that close to a threshold
    assert flips_f16 < 0.06, (flips_f16, flips_fast)
    # and a changed decision is a NEIGHBOURING one far more often than not (the low bits of the mu-law code)
    moved = np.abs(d_f16 - d_fast)[d_f16 != d_fast]
    assert moved.size == 0 or np.median(moved) <= 8, (np.median(moved), moved[:20])
    print("decision flips: FAST vs PARITY %.4f, fp16 tree vs FAST %.4f (%d samples)" % (flips_fast, flips_f16, n))
    # free running: sane statistics
    b = api.LPCNetBatch(4, blob)
    b.set_fast(2)
    f4 = np.stack([synth.make_features(9200 + s, 30) for s in range(4)])
    out = b.synthesize(f4)
    b.set_fast(0); b.reset()
    par = b.synthesize(f4)
    b.close()
    r = out[:, 320:].astype(np.float64).std() / par[:, 320:].astype(np.float64).std()
    assert np.all(out[:, :320] == 0) and 0.7 < r < 1.4, r
