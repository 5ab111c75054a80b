"""CPU tests of the checker itself: the plain-C oracle (oracle/lpcnet_oracle.c) against
 (1) the committed golden fixtures, which are outputs of the REAL reference compiled from its own
     sources (tests/tools/make_golden.py), and
 (2) the real reference directly, when oracle/_ref is present (this container; on the GPU box the
     prebuilt .so travels with the snapshot).
The reference has no golden vectors of its own for this path."""
import ctypes as C

import numpy as np
import pytest

from lpcnet_amd import synth
from oracle import orc, ref


def _state(blob):
    return orc.OracleModel(blob).new_state()


@pytest.mark.parametrize("seed", [1000, 1001, 1002])
def test_pcm_float_matches_reference_golden(blob_f32, golden, seed):
    T = int(golden["n_frames"])
    f = synth.make_features(seed, T)
    st = _state(blob_f32)
    pcm = st.synthesize(f)
    assert np.array_equal(pcm, golden[f"pcm_gf_{seed}"])          # bit-exact, integer output
    assert np.all(pcm[:320] == 0)                                  # FEATURES_DELAY frames are silent (src/lpcnet.c:239-243)
    assert np.any(pcm[320:480] != 0)
    _, _, ga, gb = st.nnet_state()
    assert np.array_equal(ga, golden[f"gru_a_gf_{seed}"])
    assert np.array_equal(gb, golden[f"gru_b_gf_{seed}"])


@pytest.mark.parametrize("seed", [1000, 1001])
def test_pcm_int8_matches_reference_golden(blob_i8, golden, seed):
    T = int(golden["n_frames"])
    pcm = _state(blob_i8).synthesize(synth.make_features(seed, T))
    assert np.array_equal(pcm, golden[f"pcm_gi_{seed}"])


def test_frame_products_match_golden(blob_f32, golden):
    T = int(golden["n_frames"])
    for seed in (1000, 1001):
        f = synth.make_features(seed, T)
        st = _state(blob_f32)
        for t in range(T):
            lpc, ca, cb = st.frame_network(f[t])
            assert np.array_equal(lpc, golden[f"lpc_gf_{seed}"][t])
            assert np.array_equal(cb, golden[f"condb_gf_{seed}"][t])
            sums = np.array([ca[::97].astype(np.float64).sum(), np.abs(ca).astype(np.float64).sum()])
            assert np.array_equal(sums, golden[f"conda_sums_gf_{seed}"][t])
            if seed == 1000 and t < 8:
                assert np.array_equal(ca, golden["conda_gf_1000_first8"][t])


def test_scalar_known_answers(golden, oracle_lib):
    L = oracle_lib
    got = np.array([L.orc_lin2ulaw(float(x)) for x in golden["ulaw_x"]], np.int32)
    assert np.array_equal(got, golden["lin2ulaw"])
    assert L.orc_lin2ulaw(0.0) == 128                                   # KAT named in SURVEY.md §8c
    assert np.array_equal(np.array([L.orc_ulaw2lin(u) for u in range(256)], np.float32), golden["ulaw2lin"])
    assert np.array_equal(np.array([L.orc_tanh_approx(float(x)) for x in golden["act_x"]], np.float32), golden["tanh"])
    assert np.array_equal(np.array([L.orc_sigmoid_approx(float(x)) for x in golden["act_x"]], np.float32), golden["sigmoid"])
    assert np.array_equal(np.array([L.orc_logit_table(i) for i in range(256)], np.float32), golden["logit_table"])


def test_kiss99(golden, oracle_lib):
    rng = np.zeros(4, np.uint32)
    oracle_lib.orc_kiss99_srand(rng, b"LPCNet", 6)
    assert np.array_equal(rng, golden["kiss99_seeded"])
    got = np.array([oracle_lib.orc_kiss99_rand(rng) for _ in range(32)], np.uint32)
    assert np.array_equal(got, golden["kiss99_first32"])


def test_lpc_from_cepstrum(golden, oracle_lib):
    out = np.zeros(16, np.float32)
    for c, want in zip(golden["lpc_ceps_in"], golden["lpc_ceps_out"]):
        oracle_lib.orc_lpc_from_cepstrum(out, np.ascontiguousarray(c))
        assert np.array_equal(out, want)


def test_decode_packet(golden, oracle_lib):
    cbs = [np.ascontiguousarray(c.reshape(-1)) for c in synth.make_codebooks(5)]
    vq = np.zeros(18, np.float32)
    for pk, want in zip(golden["packets"], golden["packet_features"]):
        feats = np.zeros(4 * 36, np.float32)
        oracle_lib.orc_decode_packet(feats, vq, np.ascontiguousarray(pk), *cbs)
        assert np.array_equal(feats.reshape(4, 36), want)


def test_teacher_forcing_matches_reference(blob_f32, golden):
    """preload semantics (src/lpcnet.c:256-259): the excitation is recomputed from the forced samples."""
    f = synth.make_features(1000, 20)
    st = _state(blob_f32)
    st.synthesize(f, preload_pcm=golden["forced_pcm_in"])
    assert np.array_equal(st.nnet_state()[2], golden["forced_gru_a"])
    assert np.array_equal(st.nnet_state()[3], golden["forced_gru_b"])
    ls, le, dm, fc, rng = st.signal_state()
    assert np.array_equal(ls, golden["forced_last_sig"]) and le == int(golden["forced_last_exc"])
    assert np.float32(dm) == golden["forced_deemph"] and np.array_equal(rng, golden["forced_rng"])
    # partially forced frames
    st = _state(blob_f32)
    half = np.zeros(20 * 160, np.int16)
    for t in range(20):
        frame = half[t * 160:(t + 1) * 160]
        frame[:80] = golden["forced_pcm_in"][t * 160:t * 160 + 80]
        st.L.orc_synthesize(st.p, np.ascontiguousarray(f[t, :20]), frame, 160, 80)
    assert np.array_equal(half, golden["half_forced_pcm"])


def test_malformed_blobs_rejected(blob_f32):
    with pytest.raises(ValueError):
        orc.OracleModel(blob_f32[:-64])                 # truncated last record
    bad = bytearray(blob_f32)
    bad[12:16] = (10 ** 9).to_bytes(4, "little")        # record size larger than the blob
    with pytest.raises(ValueError):
        orc.OracleModel(bytes(bad))


# ---------------------------------------------------------------- against the real reference ------
needs_ref = pytest.mark.skipif(not ref.available("gf"), reason="oracle/_ref not built (needs /root/reference)")


@needs_ref
@pytest.mark.parametrize("flavour,kind", [("gf", "float"), ("gi", "int8")])
def test_oracle_equals_compiled_reference(flavour, kind):
    blob = synth.blob_bytes(synth.make_model(flavour=kind))
    f = synth.make_features(1234, 40)
    want = ref.RefLib(flavour).synthesize_file(blob, f)
    got = _state(blob).synthesize(f)
    assert np.array_equal(got, want)


@needs_ref
def test_oracle_layers_equal_reference_layers(blob_f32):
    r = ref.RefLib("gf")
    rs = r.new_state(blob_f32)
    om = orc.OracleModel(blob_f32)
    L = om.L
    rng = np.random.default_rng(3)
    for _ in range(5):
        h = (rng.standard_normal(384) * 0.4).astype(np.float32)
        x = (rng.standard_normal(1152) * 0.8).astype(np.float32)
        a, b = h.copy(), h.copy()
        r.lib.ref_sparse_gru_a(rs.p, a, x)
        L.orc_sparse_gru_a(om.p, b, x)
        assert np.array_equal(a, b)
        hb = (rng.standard_normal(16) * 0.4).astype(np.float32)
        cb = (rng.standard_normal(48) * 0.5).astype(np.float32)
        a2, b2 = hb.copy(), hb.copy()
        r.lib.ref_gru_b(rs.p, cb, a2, a)
        L.orc_gru_b(om.p, cb, b2, b)
        assert np.array_equal(a2, b2)
        k1 = np.array([1, 2, 3, 4], np.uint32); k2 = k1.copy()
        assert r.lib.ref_sample_mdense(rs.p, a2, k1) == L.orc_sample_mdense(om.p, b2, k2)
        assert np.array_equal(k1, k2)


@needs_ref
def test_reference_flavours_diverge_as_documented(blob_f32):
    """the AVX2 float build is NOT bit-compatible with the generic-C build, which
    is why parity is defined against the generic-C flavour."""
    if not ref.available("af"):
        pytest.skip("AVX2 flavour not built")
    f = synth.make_features(1000, 150)
    a = ref.RefLib("gf").synthesize_file(blob_f32, f)
    b = ref.RefLib("af").synthesize_file(blob_f32, f)
    assert a.shape == b.shape and not np.array_equal(a, b)


@pytest.mark.skipif(not ref.available("gg"), reason="oracle/_ref gamma flavour not built (needs /root/reference)")
def test_lpc_gamma_variant_matches_reference(blob_f32):
    """LPC_GAMMA 0.9 (lpc_weighting, src/freq.c:299-308): the oracle's parameter == the reference compiled with that #define"""
    f = synth.make_features(1234, 40)
    want = ref.RefLib("gg").new_state(blob_f32).synthesize(f)
    got = orc.OracleModel(blob_f32, lpc_gamma=0.9).new_state().synthesize(f)
    assert np.array_equal(got, want)
    assert not np.array_equal(got, orc.OracleModel(blob_f32).new_state().synthesize(f))


@pytest.mark.skipif(not ref.available("ge"), reason="oracle/_ref END2END flavour not built (needs /root/reference)")
def test_end2end_variant_matches_reference(blob_f32):
    """END2END models (LPC from the network's reflection coefficients, src/lpcnet.c:56-80,107-108)"""
    f = synth.make_features(4321, 40)
    want = ref.RefLib("ge").new_state(blob_f32).synthesize(f)
    got = orc.OracleModel(blob_f32, end2end=True).new_state().synthesize(f)
    assert np.array_equal(got, want)
    assert not np.array_equal(got, orc.OracleModel(blob_f32).new_state().synthesize(f))


def test_fp16_restatement_rounds_like_ieee_binary16():
    """the oracle-side statement of the engine's fp16 dual-FC option rounds to binary16 exactly like IEEE (numpy float16):
    normals, subnormals, ties, overflow"""
    L = orc.lib()
    rng = np.random.default_rng(5)
    xs = np.concatenate([rng.standard_normal(4000).astype(np.float32) * s for s in (1e-8, 1e-6, 6e-5, 1e-3, 1.0, 300.0, 7e4)])
    ties = (np.arange(1, 2000, dtype=np.float32) * 2 + 1) * np.float32(2.0 ** -11)       # exactly half way between two halves near 1..2
    xs = np.concatenate([xs, ties + 1.0, -(ties + 1.0), np.array([0.0, -0.0, 65504.0, 65519.9, 65520.0, 2.0 ** -24, 2.0 ** -25, 1.5 * 2.0 ** -25], np.float32)])
    got = np.array([L.orc_f16(float(x)) for x in xs], np.float32)
    with np.errstate(over="ignore"):
        want = xs.astype(np.float16).astype(np.float32)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
