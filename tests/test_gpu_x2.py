"""GPU parity tests of the two-group sample kernel (lpcnet_amd/csrc/sample_kernel_x2.hip.h: eight float streams per workgroup, two groups of four half a
step apart; streams-per-workgroup = 8), through the C ABI, against the plain-C oracle and the reference-generated golden fixtures.  Same bar as
tests/test_gpu_parity.py: PCM, GRU states, LPC history, RNG state bit for bit (tolerance 0)."""
import numpy as np
import pytest

from lpcnet_amd import api, synth
from oracle import orc

pytestmark = pytest.mark.gpu


def feats_for(seeds, T):
    return np.stack([synth.make_features(s, 60)[:T] if T <= 60 else synth.make_features(s, T) for s in seeds])


def check_states(b, states, which):
    for s in which:
        st = b.get_state(s)
        c1, c2, ga, gb = states[s].nnet_state()
        ls, le, dm, fc, rng = states[s].signal_state()
        assert np.array_equal(np.array(st.gru_a, np.float32), ga) and np.array_equal(np.array(st.gru_b, np.float32), gb), s
        assert np.array_equal(np.array(st.conv1_mem, np.float32), c1) and np.array_equal(np.array(st.conv2_mem, np.float32), c2), s
        assert np.array_equal(np.array(st.last_sig, np.float32), ls) and st.last_exc == le and st.frame_count == fc, s
        assert np.float32(st.deemph_mem) == np.float32(dm) and np.array_equal(np.array(st.rng, np.uint32), rng), s


def oracle_states(blob, feats):
    om = orc.OracleModel(blob)
    pcm, states = [], []
    for f in feats:
        st = om.new_state()
        pcm.append(st.synthesize(f))
        states.append(st)
    return np.stack(pcm), states


@pytest.mark.parametrize("n", [8, 16, 13, 5, 1, 24])
def test_eight_streams_per_workgroup_match_the_oracle(n, blob_f32, hip_lib):
    """full, ragged (the second group of the last workgroup partly / entirely empty) and tiny batches; complete state compared"""
    T = 7
    feats = feats_for(range(2600, 2600 + n), T)
    want, states = oracle_states(blob_f32, feats)
    b = api.LPCNetBatch(n, blob_f32)
    b.streams_per_workgroup = 8
    assert b.streams_per_workgroup == 8
    got = b.synthesize(feats)
    assert np.array_equal(got, want)
    check_states(b, states, range(n))
    b.close()


def test_golden_reference_pcm_on_the_two_group_kernel(blob_f32, golden, hip_lib):
    """the reference's own generic-C float output (tests/golden), every position of a workgroup: stream s of 8 gets golden file 1000 + s % 3"""
    T = int(golden["n_frames"])
    seeds = [1000 + s % 3 for s in range(8)]
    b = api.LPCNetBatch(8, blob_f32)
    b.streams_per_workgroup = 8
    pcm = b.synthesize(feats_for(seeds, T))
    for s, seed in enumerate(seeds):
        assert np.array_equal(pcm[s], golden[f"pcm_gf_{seed}"]), s
        st = b.get_state(s)
        assert np.array_equal(np.array(st.gru_a, np.float32), golden[f"gru_a_gf_{seed}"])
        assert np.array_equal(np.array(st.gru_b, np.float32), golden[f"gru_b_gf_{seed}"])
    b.close()


def test_streaming_calls_and_chunks_equal_one_call(blob_f32, hip_lib):
    """state carried across calls (frame boundaries of the two groups are half a step apart inside a launch, not between launches), a launch
    longer than the 100-frame chunk, one-frame launches, and the same result as four streams per workgroup"""
    n, T = 11, 104
    feats = feats_for(range(4100, 4100 + n), T)
    b = api.LPCNetBatch(n, blob_f32)
    b.streams_per_workgroup = 8
    whole = b.synthesize(feats)
    b.reset()
    parts = [b.synthesize(np.ascontiguousarray(feats[:, a:z])) for a, z in ((0, 1), (1, 2), (2, 3), (3, 50), (50, 104))]
    assert np.array_equal(np.concatenate(parts, axis=1), whole)
    b4 = api.LPCNetBatch(n, blob_f32)
    b4.streams_per_workgroup = 4
    assert np.array_equal(b4.synthesize(feats), whole)
    want, _ = oracle_states(blob_f32, feats[:3, :12])
    assert np.array_equal(whole[:3, :12 * 160], want)
    assert np.all(whole[:, :320] == 0) and np.any(whole[:, 320:480] != 0)
    b.close(); b4.close()


def test_partial_reset_streams_of_one_group_at_different_frame_counts(blob_f32, hip_lib):
    """per-stream liveness inside a group and across the two groups of a workgroup: some streams freshly reset (two silent frames), others live"""
    n, T = 8, 5
    feats = feats_for(range(5100, 5100 + n), T)
    b = api.LPCNetBatch(n, blob_f32)
    b.streams_per_workgroup = 8
    first = b.synthesize(feats)
    b.reset(1, 2)                                           # streams 1, 2 (group 0) start over
    b.reset(4, 4)                                           # all of group 1 starts over
    second = b.synthesize(feats)
    om = orc.OracleModel(blob_f32)
    for s in range(n):
        st = om.new_state()
        assert np.array_equal(st.synthesize(feats[s]), first[s])
        if s in (1, 2, 4, 5, 6, 7):
            st = om.new_state()
        assert np.array_equal(st.synthesize(feats[s]), second[s]), s
    b.close()


def test_teacher_forcing_and_short_frames(blob_f32, golden, hip_lib):
    """preload (src/lpcnet.c:256-259) against the reference's golden run, whole and half frames, on every stream of a workgroup; then N < 160 samples
    per frame through the per-stream step call"""
    f = np.repeat(synth.make_features(1000, 20)[None], 8, axis=0)
    b = api.LPCNetBatch(8, blob_f32)
    b.streams_per_workgroup = 8
    forced = np.repeat(golden["forced_pcm_in"][None, :], 8, axis=0)
    out = b.synthesize(f, preload_pcm=forced, preload=160)
    want = forced.copy()
    want[:, :320] = 0
    assert np.array_equal(out, want)
    for s in (0, 3, 4, 7):
        st = b.get_state(s)
        assert np.array_equal(np.array(st.gru_a, np.float32), golden["forced_gru_a"])
        assert np.array_equal(np.array(st.gru_b, np.float32), golden["forced_gru_b"])
        assert np.array_equal(np.array(st.last_sig, np.float32), golden["forced_last_sig"])
        assert st.last_exc == int(golden["forced_last_exc"]) and np.array_equal(np.array(st.rng, np.uint32), golden["forced_rng"])
    b.reset()
    half_in = np.zeros((8, 20 * 160), np.int16)
    for t in range(20):
        half_in[:, t * 160:t * 160 + 80] = golden["forced_pcm_in"][t * 160:t * 160 + 80]
    half = b.synthesize(f, preload_pcm=half_in, preload=80)
    for s in range(8):
        assert np.array_equal(half[s], golden["half_forced_pcm"]), s
    b.close()


@pytest.mark.parametrize("N", [160, 40, 1])
def test_frames_of_n_samples_like_lpcnet_synthesize_with_n(N, blob_f32, hip_lib):
    """lpcnet_synthesize(st, features, out, N) with N != 160 for every stream (lpcnet_batch_synthesize_step, equal arguments: one group of the step
    call = one launch): the two-group kernel's frame bookkeeping with frame_len < 160, incl. frame_len = 1 (every sample a frame boundary)"""
    n, T = 9, 6
    feats = feats_for(range(5300, 5300 + n), T)
    om = orc.OracleModel(blob_f32)
    b = api.LPCNetBatch(n, blob_f32)
    b.streams_per_workgroup = 8
    sts = [om.new_state() for _ in range(n)]
    for t in range(T):
        pcm = np.zeros((n, 160), np.int16)
        got = b.synthesize_step(np.ascontiguousarray(feats[:, t]), pcm, [N] * n, [0] * n, [1] * n)
        for s in range(n):
            ref = np.zeros(160, np.int16)
            sts[s].L.orc_synthesize(sts[s].p, np.ascontiguousarray(feats[s, t, :20]), ref, N, 0)
            assert np.array_equal(got[s, :N], ref[:N]), (t, s)
    b.close()


@pytest.mark.parametrize("kw", [
    dict(densities=(0.03, 0.03, 0.12)),                                 # 24-item variant
    dict(densities=(0.045, 0.045, 0.18)),                               # 28-item variant
    dict(shaped=False, densities=(0.04, 0.06, 0.15), seed=77),
    dict(off_grid=True),
], ids=["sparseA", "midA", "unshaped", "offgrid"])
def test_other_models_on_the_two_group_kernel(kw, hip_lib):
    """other sparsity patterns -> other item-count variants and slot dealings; float weights off the k/128 grid"""
    blob = synth.blob_bytes(synth.make_model(**kw))
    rc, info = api.check_model(blob)
    assert rc == 0 and info[5] == 0
    n, T = 10, 6
    feats = feats_for(range(2700, 2700 + n), T)
    want, states = oracle_states(blob, feats)
    b = api.LPCNetBatch(n, blob)
    try:
        b.streams_per_workgroup = 8
    except api.LPCNetError:
        pytest.skip("model does not fit the two-group kernel (more than 32 items per lane)")
    got = b.synthesize(feats)
    assert np.array_equal(got, want), kw
    check_states(b, states, (0, 3, 4, 7, 9))
    b.close()


def test_models_the_two_group_kernel_cannot_run_are_refused_or_fall_back(blob_i8, hip_lib):
    """int8 blobs, block-sparse GRU-B and FAST arithmetic keep four (or fewer) streams per workgroup: asking for eight is an error, and a pinned
    eight falls back when the arithmetic flavour changes under it"""
    b = api.LPCNetBatch(8, blob_i8)
    with pytest.raises(api.LPCNetError):
        b.streams_per_workgroup = 8
    b.close()
    blob = synth.blob_bytes(synth.make_model(grub_density=0.5, seed=5))
    b = api.LPCNetBatch(8, blob)
    with pytest.raises(api.LPCNetError):
        b.streams_per_workgroup = 8
    b.close()
    blob = synth.blob_bytes(synth.make_model())
    feats = feats_for(range(2800, 2808), 5)
    b = api.LPCNetBatch(8, blob)
    b.streams_per_workgroup = 8
    want = b.synthesize(feats)
    b.reset()
    b.set_fast(True)                                        # FAST has no two-group kernel: the launch falls back to four streams per workgroup
    fast = b.synthesize(feats)
    assert fast.shape == want.shape and np.any(fast != 0)
    b.set_fast(False)
    b.reset()
    assert np.array_equal(b.synthesize(feats), want)
    b.close()


def test_the_engine_picks_eight_streams_per_workgroup_only_beyond_four_per_cu(blob_f32, hip_lib):
    """1024 streams = four per CU: the four-stream kernel; 2048: the two-group kernel (table value without a measurement)"""
    import os
    os.environ["LPCNET_HIP_NO_AUTOTUNE"] = "1"
    try:
        b = api.LPCNetBatch(1024, blob_f32)
        assert b.streams_per_workgroup == 4
        b.close()
        b = api.LPCNetBatch(2048, blob_f32)
        assert b.streams_per_workgroup == 8
        b.close()
    finally:
        del os.environ["LPCNET_HIP_NO_AUTOTUNE"]


def test_2048_distinct_streams_full_occupancy_continued_over_a_second_call(blob_f32, hip_lib):
    """the headline shape: 256 workgroups x 8 distinct streams, two calls (state round trip through global memory at full occupancy); every 16th
    stream and every stream of three whole workgroups replayed on the oracle in a process pool"""
    n, T = 2048, 6
    feats = np.stack([synth.make_features(7000 + s, 2 * T) for s in range(n)])
    b = api.LPCNetBatch(n, blob_f32)
    b.streams_per_workgroup = 8
    a1 = b.synthesize(np.ascontiguousarray(feats[:, :T]))
    a2 = b.synthesize(np.ascontiguousarray(feats[:, T:]))
    got = np.concatenate([a1, a2], axis=1)
    pick = sorted(set(range(0, n, 16)) | set(range(0, 8)) | set(range(1016, 1024)) | set(range(2040, 2048)))
    want = orc.synthesize_many(blob_f32, np.ascontiguousarray(feats[pick]))
    assert np.array_equal(got[pick], want)
    nz = (got != 0).mean()
    assert nz > 0.5
    b.close()
