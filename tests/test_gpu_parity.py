"""GPU parity tests: the HIP engine, called through its C ABI, against
  * the committed golden fixtures (outputs of the real reference, generic-C float build), and
  * the plain-C oracle on the same seeded inputs.
Integer/index results (PCM, mu-law codes, RNG state) and -- because the engine reproduces the
reference's accumulation order without FMA -- also every float state are compared BIT-EXACTLY
(tolerance 0; north_star allows +-1 mu-law level, which free-running chaotic synthesis cannot use:
SURVEY.md fact 8)."""
import ctypes as C
import os

import numpy as np
import pytest

from lpcnet_amd import api, synth
from oracle import orc

pytestmark = pytest.mark.gpu


def oracle_run(blob, feats, with_products=False):
    om = orc.OracleModel(blob)
    n, T, _ = feats.shape
    pcm = np.zeros((n, T * 160), np.int16)
    ca = np.zeros((n, T, 1152), np.float32); cb = np.zeros((n, T, 48), np.float32); lp = np.zeros((n, T, 16), np.float32)
    states = []
    for s in range(n):
        st = om.new_state()
        for t in range(T):
            st.L.orc_synthesize(st.p, np.ascontiguousarray(feats[s, t, :20]), pcm[s, t * 160:(t + 1) * 160], 160, 0)
            if with_products:
                lp[s, t], ca[s, t], cb[s, t] = st.frame_products()
        states.append(st)
    return (pcm, ca, cb, lp, states) if with_products else (pcm, states)


def feats_for(seeds, T):
    """first T frames of the 60-frame feature files the golden fixtures were made from (longer: own file)"""
    return np.stack([synth.make_features(s, 60)[:T] if T <= 60 else synth.make_features(s, T) for s in seeds])


@pytest.fixture(scope="module")
def batch4(blob_f32, hip_lib):
    b = api.LPCNetBatch(4, blob_f32)
    yield b
    b.close()


def test_golden_reference_pcm_single_stream(blob_f32, golden, hip_lib):
    """config 1: one stream on the GPU == the reference's generic-C float output, bit for bit."""
    T = int(golden["n_frames"])
    for seed in (1000, 1001, 1002):
        b = api.LPCNetBatch(1, blob_f32)
        pcm = b.synthesize(feats_for([seed], T))
        assert np.array_equal(pcm[0], golden[f"pcm_gf_{seed}"])
        st = b.get_state(0)
        assert np.array_equal(np.array(st.gru_a, np.float32), golden[f"gru_a_gf_{seed}"])
        assert np.array_equal(np.array(st.gru_b, np.float32), golden[f"gru_b_gf_{seed}"])
        b.close()


@pytest.mark.parametrize("S", [1, 2, 4])
def test_multi_stream_matches_oracle_all_interleavings(blob_f32, S, hip_lib):
    """streams-per-workgroup 1/2/4 (DPP broadcast paths), stream count not a multiple of S."""
    n, T = 7, 9
    feats = feats_for(range(2000, 2000 + n), T)
    want, states = oracle_run(blob_f32, feats)
    b = api.LPCNetBatch(n, blob_f32)
    b.streams_per_workgroup = S
    got = b.synthesize(feats)
    assert np.array_equal(got, want)
    for s in range(n):
        st = b.get_state(s)
        c1, c2, ga, gb = states[s].nnet_state()
        ls, le, dm, fc, rng = states[s].signal_state()
        assert np.array_equal(np.array(st.gru_a, np.float32), ga) and np.array_equal(np.array(st.gru_b, np.float32), gb)
        assert np.array_equal(np.array(st.conv1_mem, np.float32), c1) and np.array_equal(np.array(st.conv2_mem, np.float32), c2)
        assert np.array_equal(np.array(st.last_sig, np.float32), ls) and st.last_exc == le and st.frame_count == fc
        assert np.float32(st.deemph_mem) == np.float32(dm) and np.array_equal(np.array(st.rng, np.uint32), rng)
    b.close()


def test_frame_network_and_lpc_seam(blob_f32, golden, batch4):
    """frame-rate kernels alone (conv/dense/LPC incl. the 320-point FFT and double-precision pow)."""
    T = 24
    feats = feats_for([1000, 1001, 3000, 3001], T)
    _, ca, cb, lp, _ = oracle_run(blob_f32, feats, with_products=True)
    batch4.reset()
    gca, gcb, glp = batch4.run_frames(feats)
    assert np.array_equal(gca, ca) and np.array_equal(gcb, cb) and np.array_equal(glp, lp)
    assert np.array_equal(glp[0], golden["lpc_gf_1000"][:T]) and np.array_equal(gcb[1], golden["condb_gf_1001"][:T])
    assert np.array_equal(gca[0, :8], golden["conda_gf_1000_first8"])


def test_sample_loop_seam(blob_f32, batch4):
    """sample kernel alone on frame products supplied by the oracle (SURVEY.md §7 hard part 9)."""
    T = 6
    feats = feats_for([1000, 1001, 1002, 1003], T)
    _, ca, cb, lp, _ = oracle_run(blob_f32, feats, with_products=True)
    om = orc.OracleModel(blob_f32)
    want = np.zeros((4, T * 160), np.int16)
    for s in range(4):
        o = om.new_state()
        o.L.orc_force_frame_count(o.p, 3)
        for t in range(T):
            o.L.orc_synthesize_tail(o.p, ca[s, t], cb[s, t], lp[s, t], want[s, t * 160:(t + 1) * 160], 160, 0)
    batch4.reset()
    for s in range(4):
        st = batch4.get_state(s)
        st.frame_count = 3
        batch4.set_state(s, st)
    got = batch4.run_tail(ca, cb, lp)
    assert np.array_equal(got, want)


@pytest.mark.parametrize("n", [1, 2, 3, 4, 5, 9])
def test_one_frame_steps_for_small_batches_match_oracle(n, blob_f32, hip_lib):
    """round 5: the frame kernels' tile size is a template parameter (1 / 2 / 4 / 8 streams of one frame each, 1 / 2 / 4 / 8 (stream, frame)
    items in the projection, whose row blocks are separate workgroups for small batches): every instantiation, frame by frame and in
    chunks of 2 / 3 frames, PCM and frame products against the oracle"""
    T = 7
    feats = feats_for(range(7300, 7300 + n), T)
    want, ca, cb, lp, _ = oracle_run(blob_f32, feats, with_products=True)
    b = api.LPCNetBatch(n, blob_f32)
    got = np.concatenate([b.synthesize(np.ascontiguousarray(feats[:, t:t + 1])) for t in range(T)], axis=1)
    assert np.array_equal(got, want), n
    b.reset()
    got = np.concatenate([b.synthesize(np.ascontiguousarray(feats[:, a:z])) for a, z in ((0, 2), (2, 5), (5, 7))], axis=1)
    assert np.array_equal(got, want), n
    b.reset()
    a_, b_, l_ = b.run_frames(feats)
    assert np.array_equal(a_, ca) and np.array_equal(b_, cb) and np.array_equal(l_, lp)
    b.reset()
    for t in range(T):                                         # ... and the frame network alone, one frame per call
        a_, b_, l_ = b.run_frames(np.ascontiguousarray(feats[:, t:t + 1]))
        assert np.array_equal(a_[:, 0], ca[:, t]) and np.array_equal(b_[:, 0], cb[:, t]) and np.array_equal(l_[:, 0], lp[:, t]), (n, t)
    b.close()


@pytest.mark.parametrize("S", [4, 8])
def test_a_step_captured_in_a_hip_graph_replays_bit_exactly(S, blob_f32, hip_lib):
    """the device-pointer call is enqueue-only and CAPTURABLE: one real-time step (frame kernels + sample kernel + the sample kernel's argument block,
    which travels as the by-value parameter of a one-lane kernel inside the graph) is captured into a HIP graph on a side stream and replayed frame
    after frame on a static feature / PCM buffer -- every replay must equal the oracle, and the batch must stay usable.  Round 6: the
    argument block no longer comes from a pool of 32 pinned slots that every captured launch used up for good -- a server may re-capture as often as it
    likes (45 captures below); four streams per workgroup and the two-group kernel (eight)."""
    import torch
    n, T = 64, 8
    feats = feats_for(range(8800, 8800 + n), T)
    want, _ = oracle_run(blob_f32, feats[:6])
    dev = torch.device("cuda:0")
    b = api.LPCNetBatch(n, blob_f32)
    b.streams_per_workgroup = S
    d_feat = torch.zeros((n, 36), dtype=torch.float32, device=dev)
    d_pcm = torch.zeros((n, 160), dtype=torch.int16, device=dev)
    s = torch.cuda.Stream()
    out = []
    with torch.cuda.stream(s):                                   # frame 0 eagerly (first launch: function attributes, code-object load)
        d_feat.copy_(torch.from_numpy(np.ascontiguousarray(feats[:, 0])))
        b.synthesize_device(d_feat.data_ptr(), 36, d_pcm.data_ptr(), 1, s.cuda_stream)
        s.synchronize()
        out.append(d_pcm.cpu().numpy().copy())
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        b.synthesize_device(d_feat.data_ptr(), 36, d_pcm.data_ptr(), 1, torch.cuda.current_stream().cuda_stream)
    for t in range(1, 4):
        d_feat.copy_(torch.from_numpy(np.ascontiguousarray(feats[:, t])))
        torch.cuda.synchronize()
        g.replay()
        torch.cuda.synchronize()
        out.append(d_pcm.cpu().numpy().copy())
    del g
    for k in range(45):                                          # re-capture over and over (a captured launch used to cost one of 32 slots for the batch's lifetime)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            b.synthesize_device(d_feat.data_ptr(), 36, d_pcm.data_ptr(), 1, torch.cuda.current_stream().cuda_stream)
        if k < 44:
            del g
    for t in range(4, T):
        d_feat.copy_(torch.from_numpy(np.ascontiguousarray(feats[:, t])))
        torch.cuda.synchronize()
        g.replay()
        torch.cuda.synchronize()
        out.append(d_pcm.cpu().numpy().copy())
    got = np.concatenate(out, axis=1)
    assert np.array_equal(got[:6], want)
    st = b.get_state(5)                                          # host-side access after the replays: the batch's own bookkeeping is intact
    assert st.frame_count == T
    del g
    b.close()


def test_streaming_calls_equal_one_call_and_chunking(blob_f32, hip_lib):
    """state carried across calls; n_frames above the internal 100-frame chunk; single-frame calls."""
    n, T = 2, 104
    feats = feats_for([4000, 4001], T)
    b = api.LPCNetBatch(n, blob_f32)
    whole = b.synthesize(feats)
    b.reset()
    parts = [b.synthesize(np.ascontiguousarray(feats[:, a:z])) for a, z in ((0, 1), (1, 2), (2, 50), (50, 104))]
    assert np.array_equal(np.concatenate(parts, axis=1), whole)
    want, _ = oracle_run(blob_f32, feats[:, :12])
    assert np.array_equal(whole[:, :12 * 160], want)
    assert np.all(whole[:, :320] == 0) and np.any(whole[:, 320:480] != 0)
    # determinism: run twice from reset
    b.reset()
    assert np.array_equal(b.synthesize(feats), whole)
    b.close()


def test_reset_subset_and_mixed_start(blob_f32, hip_lib):
    """streams of one workgroup at different frame counts (one freshly reset): per-stream liveness."""
    n, T = 4, 5
    feats = feats_for([5000, 5001, 5002, 5003], T)
    b = api.LPCNetBatch(n, blob_f32)
    b.streams_per_workgroup = 4
    first = b.synthesize(feats)
    b.reset(1, 2)                                           # streams 1,2 start over; 0,3 continue
    second = b.synthesize(feats)
    om = orc.OracleModel(blob_f32)
    for s in range(n):
        st = om.new_state()
        a = st.synthesize(feats[s])
        assert np.array_equal(a, first[s])
        if s in (1, 2):
            st = om.new_state()
        assert np.array_equal(st.synthesize(feats[s]), second[s])
    b.close()


def test_teacher_forcing_matches_reference_golden(blob_f32, golden, hip_lib):
    f = synth.make_features(1000, 20)[None]                  # the 20-frame file the fixture was made from
    b = api.LPCNetBatch(1, blob_f32)
    forced = golden["forced_pcm_in"][None, :]
    out = b.synthesize(f, preload_pcm=forced, preload=160)
    want = forced.copy()
    want[:, :320] = 0                                        # start-up frames are cleared even when forced (src/lpcnet.c:239-243)
    assert np.array_equal(out, want)                         # later forced samples are left untouched
    st = b.get_state(0)
    assert np.array_equal(np.array(st.gru_a, np.float32), golden["forced_gru_a"])
    assert np.array_equal(np.array(st.gru_b, np.float32), golden["forced_gru_b"])
    assert np.array_equal(np.array(st.last_sig, np.float32), golden["forced_last_sig"])
    assert st.last_exc == int(golden["forced_last_exc"]) and np.array_equal(np.array(st.rng, np.uint32), golden["forced_rng"])
    b.reset()
    half_in = np.zeros((1, 20 * 160), np.int16)
    for t in range(20):
        half_in[0, t * 160:t * 160 + 80] = golden["forced_pcm_in"][t * 160:t * 160 + 80]
    half = b.synthesize(f, preload_pcm=half_in, preload=80)
    assert np.array_equal(half[0], golden["half_forced_pcm"])
    b.close()


def test_single_stream_c_api_like_lpcnet_demo(blob_f32, golden, hip_lib):
    """lpcnet_create / lpcnet_load_model / lpcnet_synthesize per frame (src/lpcnet_demo.c:202-219),
    POD state copy (PLC-style snapshot / rollback) and N < 160."""
    T = 12
    f = synth.make_features(1000, int(golden["n_frames"]))[:T]
    st = api.LPCNetState(blob_f32)
    pcm = np.concatenate([st.synthesize(f[t]) for t in range(T)])
    assert np.array_equal(pcm, golden["pcm_gf_1000"][:T * 160])
    # snapshot by value, run ahead, roll back, run again -> identical
    L = hip_lib
    size = L.lpcnet_get_size()
    snap = C.string_at(st.p, size)
    a = st.synthesize(f[3])
    C.memmove(st.p, snap, size)
    assert np.array_equal(st.synthesize(f[3]), a)
    # N < 160: one frame-network step, N samples
    om = orc.OracleModel(blob_f32)
    o = om.new_state()
    s2 = api.LPCNetState(blob_f32)
    for t in range(6):
        n = 160 if t % 2 == 0 else 57
        want = np.zeros(n, np.int16)
        o.L.orc_synthesize(o.p, np.ascontiguousarray(f[t, :20]), want, n, 0)
        assert np.array_equal(s2.synthesize(f[t], n), want)


def test_state_export_import_between_apis(blob_f32, hip_lib):
    T = 8
    feats = feats_for([6000, 6001], T)
    b = api.LPCNetBatch(2, blob_f32)
    b.synthesize(feats[:, :4])
    st = api.LPCNetState(blob_f32)
    assert hip_lib.lpcnet_batch_export_state(b.p, 1, st.p) == 0
    tail_single = np.concatenate([st.synthesize(feats[1, t]) for t in range(4, T)])
    tail_batch = b.synthesize(np.ascontiguousarray(feats[:, 4:]))
    assert np.array_equal(tail_single, tail_batch[1])
    b2 = api.LPCNetBatch(1, blob_f32)
    st2 = api.LPCNetState(blob_f32)
    assert hip_lib.lpcnet_batch_export_state(b.p, 0, st2.p) == 0 and hip_lib.lpcnet_batch_import_state(b2.p, 0, st2.p) == 0
    more = feats_for([6002], 3)
    assert np.array_equal(b2.synthesize(more)[0], np.concatenate([st2.synthesize(more[0, t]) for t in range(3)]))
    b.close(); b2.close()


def test_codec_path_matches_reference_golden(blob_f32, golden, hip_lib):
    """lpcnet_decode: 8 bytes -> 640 samples (src/lpcnet.c:310-319) with seeded VQ codebooks."""
    api.set_codebooks(*synth.make_codebooks(5))
    dec = api.LPCNetDecState(blob_f32)
    pcm = np.stack([dec.decode(p) for p in golden["packets"]])
    assert np.array_equal(pcm, golden["packet_pcm_gf"])
    b = api.LPCNetBatch(3, blob_f32)
    pk = np.stack([golden["packets"]] * 3)
    out = b.decode(pk)
    for s in range(3):
        assert np.array_equal(out[s], golden["packet_pcm_gf"].reshape(-1))
    # streaming: the VQ memory lives on the device between calls; reset clears it; device-pointer entry point
    import torch
    b.reset()
    P = pk.shape[1]
    part = np.concatenate([b.decode(pk[:, :2]), b.decode(pk[:, 2:])], axis=1)
    assert np.array_equal(part, out)
    b.reset()
    d_pk = torch.from_numpy(np.ascontiguousarray(pk)).cuda()
    d_pcm = torch.zeros((3, P * 640), dtype=torch.int16, device="cuda")
    b.decode_device(d_pk.data_ptr(), d_pcm.data_ptr(), P, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert np.array_equal(d_pcm.cpu().numpy(), out)
    # random packets (every bit pattern is a valid packet): device unpacking == the single-stream host path
    rng = np.random.default_rng(9)
    rp = rng.integers(0, 256, size=(2, 5, 8), dtype=np.uint8)
    b2 = api.LPCNetBatch(2, blob_f32)
    got = b2.decode(rp)
    for s in range(2):
        d1 = api.LPCNetDecState(blob_f32)
        want = np.concatenate([d1.decode(rp[s, k]) for k in range(5)])
        assert np.array_equal(got[s], want)
    b2.close()
    b.close()


def test_device_pointer_api_with_torch(blob_f32, hip_lib):
    """HBM-resident inputs/outputs on a caller stream (what bench.py times)."""
    import torch
    n, T = 8, 6
    feats = feats_for(range(7000, 7000 + n), T)
    want, _ = oracle_run(blob_f32, feats)
    b = api.LPCNetBatch(n, blob_f32)
    d_f = torch.from_numpy(feats).cuda()
    d_p = torch.zeros((n, T * 160), dtype=torch.int16, device="cuda")
    b.synthesize_device(d_f.data_ptr(), 36, d_p.data_ptr(), T, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert np.array_equal(d_p.cpu().numpy(), want)
    b.close()


def test_lpc_kernel_random_cepstra(blob_f32, hip_lib, oracle_lib):
    """wide random cepstra through the device LPC path (pow in double, 320-pt FFT, Levinson)."""
    rng = np.random.default_rng(11)
    n, T = 16, 40
    feats = np.zeros((n, T, 36), np.float32)
    feats[:, :, :18] = rng.standard_normal((n, T, 18)) * np.array([3.0] + [1.2] * 17)
    feats[:, :, 18] = rng.uniform(-1.3, 3.0, (n, T))
    b = api.LPCNetBatch(n, blob_f32)
    _, _, glp = b.run_frames(feats)
    out = np.zeros(16, np.float32)
    for s in range(n):
        for t in range(T - 2):
            oracle_lib.orc_lpc_from_cepstrum(out, np.ascontiguousarray(feats[s, t, :18]))
            assert np.array_equal(glp[s, t + 2], out), (s, t)       # two-frame delay line (src/lpcnet.c:110-112)
        assert np.all(glp[s, :2] == 0)
    b.close()


def test_full_size_1024_streams_properties(blob_f32, golden, hip_lib):
    """BASELINE config 2 size: 1024 concurrent streams.  Size-independent properties: streams fed the
    same features are bit-identical to each other and to the reference golden; start-up frames are
    silent; every stream is finite and non-degenerate."""
    n, T = 1024, 12
    base = feats_for([1000, 1001, 1002, 1003], T)
    feats = np.ascontiguousarray(base[np.arange(n) % 4])
    b = api.LPCNetBatch(n, blob_f32)
    assert b.streams_per_workgroup == 4
    pcm = b.synthesize(feats)
    for k in range(4):
        grp = pcm[k::4]
        assert np.all(grp == grp[0])
    for k, seed in enumerate((1000, 1001, 1002)):
        assert np.array_equal(pcm[k], golden[f"pcm_gf_{seed}"][:T * 160])
    want, _ = oracle_run(blob_f32, base)
    assert np.array_equal(pcm[:4], want)
    assert np.all(pcm[:, :320] == 0) and np.all(np.abs(pcm.astype(np.int32)).max(axis=1) > 0)
    b.close()


def test_int8_golden_reference_pcm(blob_i8, golden, hip_lib):
    """config 4: the int8 (DOT_PROD) blob on the GPU == the reference's generic-C int8 build, bit for bit."""
    T = int(golden["n_frames"])
    for seed in (1000, 1001, 1002):
        b = api.LPCNetBatch(1, blob_i8)
        pcm = b.synthesize(feats_for([seed], T))
        assert np.array_equal(pcm[0], golden[f"pcm_gi_{seed}"])
        b.close()


@pytest.mark.parametrize("S", [1, 2, 4])
def test_int8_multi_stream_matches_oracle(blob_i8, S, hip_lib):
    """int8 engine, streams-per-workgroup 1/2/4, ragged stream count, full state comparison."""
    n, T = 7, 9
    feats = feats_for(range(2100, 2100 + n), T)
    want, states = oracle_run(blob_i8, feats)
    b = api.LPCNetBatch(n, blob_i8)
    b.streams_per_workgroup = S
    got = b.synthesize(feats)
    assert np.array_equal(got, want)
    for s in range(n):
        st = b.get_state(s)
        _, _, ga, gb = states[s].nnet_state()
        ls, le, dm, fc, rng = states[s].signal_state()
        assert np.array_equal(np.array(st.gru_a, np.float32), ga) and np.array_equal(np.array(st.gru_b, np.float32), gb)
        assert np.array_equal(np.array(st.last_sig, np.float32), ls) and st.last_exc == le and st.frame_count == fc
        assert np.array_equal(np.array(st.rng, np.uint32), rng)
    b.close()


def test_int8_streaming_state_roundtrip(blob_i8, hip_lib):
    """the quantised activations are rebuilt from the float state at every launch: split runs == one run."""
    feats = feats_for([4100, 4101, 4102], 12)
    b = api.LPCNetBatch(3, blob_i8)
    whole = b.synthesize(feats)
    b.reset()
    parts = np.concatenate([b.synthesize(feats[:, :5]), b.synthesize(feats[:, 5:6]), b.synthesize(feats[:, 6:])], axis=1)
    assert np.array_equal(whole, parts)
    b.close()


@pytest.mark.parametrize("kw", [
    dict(densities=(0.02, 0.02, 0.1)),                                  # sparse GRU-A: smallest item-count variant
    dict(densities=(0.07, 0.07, 0.25)),                                 # denser GRU-A: a larger register-resident variant
    dict(grub_density=0.3),                                             # block-sparse GRU-B input matrix (indexed path)
    dict(flavour="int8", densities=(0.07, 0.07, 0.25), grub_density=0.4),
    dict(shaped=False, densities=(0.04, 0.06, 0.15), grub_density=0.6, seed=77),
    dict(densities=(0.08, 0.08, 0.3)),                                  # 40 items per lane: items past the 28th are streamed from L2 (round 5), in P1 and in the heads
    dict(densities=(0.1, 0.1, 0.35)),                                   # 48 items per lane: 1.83 x the benchmark model's blocks (refused before round 5)
    dict(densities=(0.1, 0.1, 0.35), grub_density=0.5, seed=5),         # ... with the indexed GRU-B path
], ids=["sparseA", "denseA", "sparseB", "int8-denseA-sparseB", "unshaped", "denseA40-streamed", "denseA48-streamed", "denseA48-sparseB"])
def test_other_model_shapes_match_oracle(kw, hip_lib):
    """the slot packing, the item-count variants and the GRU-B paths depend on the model: other sparsity patterns"""
    blob = synth.blob_bytes(synth.make_model(**kw))
    rc, info = api.check_model(blob)
    assert rc == 0 and info[5] == 0
    if kw.get("densities", (0,))[0] >= 0.08:
        assert info[3] > 32                                              # (the model really needs a streamed variant)
    n, T = 5, 7
    feats = feats_for(range(2300, 2300 + n), T)
    want, states = oracle_run(blob, feats)
    for S in ((1, 2, 4) if info[3] > 32 else (1, 4)):
        b = api.LPCNetBatch(n, blob)
        b.streams_per_workgroup = S
        got = b.synthesize(feats)
        assert np.array_equal(got, want), (kw, S)
        st = b.get_state(n - 1)
        _, _, ga, gb = states[n - 1].nnet_state()
        assert np.array_equal(np.array(st.gru_a, np.float32), ga) and np.array_equal(np.array(st.gru_b, np.float32), gb)
        b.close()


@pytest.mark.parametrize("kw", [dict(skew=0.1), dict(skew=0.15), dict(skew=0.1, flavour="int8"), dict(skew=0.4)],
                         ids=["skew0.1", "skew0.15", "skew0.1-int8", "skew0.4-full-rows"])
def test_trained_like_skewed_sparsity_loads_and_matches_the_oracle(kw, hip_lib):
    """the sparsifier acts on TRAINED weights in the reference (training_tf2/lpcnet.py:73-129) and the block count per row group is
    heavy-tailed there; the synthetic default (i.i.d. weights) gives every group of a gate about the same count.  synth.make_model(skew=...) keeps the
    number of blocks per gate and makes their distribution over the row groups log-normal: groups with 65 / 74 / all 96 of their possible blocks,
    dozens of empty ones.  Such models used to be REFUSED at load (more than 48 items on a lane); they run on the 64 / 80 / 96-item variants (28 items
    resident, the rest and their block indices streamed from L2) and must match the oracle bit for bit like any other."""
    m = synth.make_model(**kw)
    blob = synth.blob_bytes(m)
    rc, info = api.check_model(blob)
    assert rc == 0 and info[5] == 0 and info[1] == 1382
    assert info[3] > 48                                                  # (the model really needs one of the new variants)
    n, T = 5, 6
    feats = feats_for(range(2900, 2900 + n), T)
    want, states = oracle_run(blob, feats)
    for S in (1, 4):
        b = api.LPCNetBatch(n, blob)
        b.streams_per_workgroup = S
        got = b.synthesize(feats)
        assert np.array_equal(got, want), (kw, S)
        st = b.get_state(n - 1)
        _, _, ga, gb = states[n - 1].nnet_state()
        assert np.array_equal(np.array(st.gru_a, np.float32), ga) and np.array_equal(np.array(st.gru_b, np.float32), gb)
        b.close()
    assert np.any(want[:, 320:] != 0)


def test_plc_facing_entry_points(blob_f32, hip_lib):
    """the reference's internal entry points (src/lpcnet_private.h:125-132), which src/lpcnet_plc.c drives:
    split frame-network / tail calls, the deferred feature queue, teacher forcing and reset_signal"""
    T = 9
    feats = feats_for([5100], T)[0]
    a = api.LPCNetState(blob_f32)
    whole = np.concatenate([a.synthesize(feats[t]) for t in range(T)])
    # 1. run_frame_network + lpcnet_synthesize_tail_impl == lpcnet_synthesize, products == oracle's
    om = orc.OracleModel(blob_f32)
    o = om.new_state()
    b = api.LPCNetState(blob_f32)
    split = []
    for t in range(T):
        frame = np.zeros(160, np.int16)
        o.L.orc_synthesize(o.p, np.ascontiguousarray(feats[t, :20]), frame, 160, 0)
        lp, ca, cb = o.frame_products()
        split.append(b.synthesize_impl(feats[t]))
        raw = b.raw_bytes()
        off = 8 + C.sizeof(api.StreamState)
        got_ca = np.frombuffer(raw, np.float32, 1152, off)
        got_cb = np.frombuffer(raw, np.float32, 48, off + 1152 * 4)
        assert np.array_equal(got_ca, ca) and np.array_equal(got_cb, cb)
        assert np.array_equal(split[-1], frame)
    assert np.array_equal(np.concatenate(split), whole)
    # 2. deferred queue: frames pushed and flushed advance the network exactly like direct calls
    c, d = api.LPCNetState(blob_f32), api.LPCNetState(blob_f32)
    for t in range(3):
        c.run_frame_network(feats[t])
        d.run_frame_network_deferred(feats[t])
    d.run_frame_network_flush()
    nstate = 8 + C.sizeof(api.StreamState)               # (the queue itself keeps its stale frames, like the reference's)
    assert c.raw_bytes()[:nstate] == d.raw_bytes()[:nstate]
    for t in range(6):                                    # more than 4 queued: the oldest are dropped (src/lpcnet.c:126-127)
        d.run_frame_network_deferred(feats[t])
    d.run_frame_network_flush()
    for t in range(2, 6):
        c.run_frame_network(feats[t])
    assert c.raw_bytes()[:nstate] == d.raw_bytes()[:nstate]
    # 3. teacher forcing through lpcnet_synthesize_impl == the batch preload path; N < 160
    e = api.LPCNetState(blob_f32)
    bt = api.LPCNetBatch(1, blob_f32)
    forced = (np.arange(40) * 37 % 2000 - 1000).astype(np.int16)
    for t in range(4):
        pe = e.synthesize_impl(feats[t], 160, forced)
        buf = np.zeros((1, 160), np.int16); buf[0, :40] = forced
        pb = bt.synthesize(feats[None, t:t + 1], preload_pcm=buf, preload=40)
        assert np.array_equal(pe, pb[0])
    bt.close()
    # 4. reset_signal clears the signal path and both GRU states, nothing else (src/lpcnet.c:226-233)
    before = api.StreamState.from_buffer_copy(e.raw_bytes()[8:8 + C.sizeof(api.StreamState)])
    e.reset_signal()
    after = api.StreamState.from_buffer_copy(e.raw_bytes()[8:8 + C.sizeof(api.StreamState)])
    assert not any(after.gru_a) and not any(after.gru_b) and not any(after.last_sig) and after.deemph_mem == 0 and after.last_exc == 128
    assert list(after.conv1_mem) == list(before.conv1_mem) and after.frame_count == before.frame_count and list(after.rng) == list(before.rng)


def test_lpc_gamma_variant(blob_f32, hip_lib):
    """LPC_GAMMA != 1 (a #define of the reference's generated header): bandwidth-expanded LPC on the device == oracle"""
    n, T = 2, 8
    feats = feats_for([8100, 8101], T)
    om = orc.OracleModel(blob_f32, lpc_gamma=0.9)
    want = np.stack([om.new_state().synthesize(feats[s]) for s in range(n)])
    b = api.LPCNetBatch(n, blob_f32)
    b.set_lpc_gamma(0.9)
    got = b.synthesize(feats)
    assert np.array_equal(got, want)
    plain, _ = oracle_run(blob_f32, feats)
    assert not np.array_equal(plain, want)
    with pytest.raises(api.LPCNetError):
        b.set_lpc_gamma(1.5)
    b.close()


def test_end2end_variant(blob_f32, hip_lib):
    """END2END models (src/lpcnet.c:56-80,107-108): LPC from the conditioning network's reflection coefficients"""
    n, T = 3, 8
    feats = feats_for([8200, 8201, 8202], T)
    om = orc.OracleModel(blob_f32, lpc_gamma=0.95, end2end=True)
    want = np.stack([om.new_state().synthesize(feats[s]) for s in range(n)])
    b = api.LPCNetBatch(n, blob_f32)
    b.set_end2end(True)
    b.set_lpc_gamma(0.95)
    got = np.concatenate([b.synthesize(feats[:, :3]), b.synthesize(feats[:, 3:])], axis=1)      # also across calls
    assert np.array_equal(got, want)
    b.close()


@pytest.mark.parametrize("kw", [dict(), dict(grub_density=0.4), dict(densities=(0.07, 0.07, 0.25))], ids=["default", "sparseB", "denseA"])
def test_off_grid_float_model(kw, hip_lib):
    """float blobs whose GRU weights are NOT multiples of 1/128 (a model trained without quantisation; the synthetic
    default and the int8 flavour sit on that grid): arbitrary float weights, same bar -- bit-exact against the oracle."""
    blob = synth.blob_bytes(synth.make_model(off_grid=True, **kw))
    rc, info = api.check_model(blob)
    assert rc == 0 and info[0] == 0
    n, T = 7, 9
    feats = feats_for(range(2400, 2400 + n), T)
    want, states = oracle_run(blob, feats)
    for S in (1, 2, 4):
        b = api.LPCNetBatch(n, blob)
        b.streams_per_workgroup = S
        got = b.synthesize(feats)
        assert np.array_equal(got, want), (kw, S)
        for s in (0, n - 1):
            st = b.get_state(s)
            _, _, ga, gb = states[s].nnet_state()
            assert np.array_equal(np.array(st.gru_a, np.float32), ga) and np.array_equal(np.array(st.gru_b, np.float32), gb)
        b.close()


def test_registry_keeps_handles_valid_when_device_sides_are_recycled(hip_lib):
    """the single-stream registry keeps at most 16 device sides resident.  A 17th distinct blob releases the
    DEVICE side of the least recently used idle slot only: its handle and blob copy stay valid, so a long-lived state of
    that model keeps working bit for bit at its next call (it used to stop the process), states of resident models are
    untouched, and binding the same blob again finds the same slot."""
    feats = synth.make_features(1000, 4)
    run = lambda st: np.concatenate([st.synthesize(f) for f in feats])
    blobs = [synth.blob_bytes(synth.make_model(seed=2000 + i)) for i in range(19)]
    first = api.LPCNetState(blobs[0])
    want0 = run(first)
    assert np.any(want0 != 0)
    states = [api.LPCNetState(b) for b in blobs[1:]]             # 19 distinct blobs: the device sides of the oldest ones are recycled
    wants = [run(st) for st in states[:3]]
    cont = api.LPCNetState(blobs[0])
    assert np.array_equal(run(cont), want0)                       # the recycled model, bound again
    first.reset()
    assert np.array_equal(run(first), want0)                      # ... and the state that held its handle all along
    for st, w in zip(states[:3], wants):
        st.reset()
        assert np.array_equal(run(st), w)


def test_batch_step_with_per_stream_arguments_like_a_batched_plc(blob_f32, hip_lib):
    """VERDICT r2 (missing 5): the PLC's call pattern per frame is lpcnet_synthesize_impl(N = TRAINING_OFFSET, preload) /
    lpcnet_synthesize_tail_impl(N, preload) / nothing, per stream (src/lpcnet_plc.c:224-239,308-320,405-421).
    lpcnet_batch_synthesize_step takes (mode, n_samples, preload) PER STREAM: 12 streams, every call a different mix of
    skipped streams, full / partial frames with and without teacher forcing and tail-only steps; each stream must equal the
    oracle driven alone with the same calls, bit for bit, including the samples it must leave untouched."""
    rng = np.random.default_rng(7)
    n, calls = 12, 14
    om = orc.OracleModel(blob_f32)
    ost = [om.new_state() for _ in range(n)]
    b = api.LPCNetBatch(n, blob_f32)
    had_frame = [False] * n
    for c in range(calls):
        feats = np.stack([synth.make_features(5000 + 31 * c + s, 1)[0] for s in range(n)])
        pcm_in = rng.integers(-3000, 3000, size=(n, 160)).astype(np.int16)
        mode = rng.integers(0, 3, size=n)
        if c < 3:
            mode[:] = 1                                      # start-up frames for everybody
        mode = np.where((mode == 2) & ~np.array(had_frame), 1, mode)
        ns = rng.choice([160, 120, 40, 1, 77], size=n)
        pre = np.array([int(rng.integers(0, k + 1)) if rng.random() < 0.6 else 0 for k in ns])
        want = pcm_in.copy()
        for s in range(n):
            if mode[s] == 0:
                continue
            frame = want[s]
            if mode[s] == 1:
                ost[s].L.orc_synthesize(ost[s].p, np.ascontiguousarray(feats[s, :20]), frame, int(ns[s]), int(pre[s]))
                had_frame[s] = True
            else:
                lpc, ca, cb = ost[s].frame_products()
                ost[s].L.orc_synthesize_tail(ost[s].p, ca, cb, lpc, frame, int(ns[s]), int(pre[s]))
        got = b.synthesize_step(feats, pcm_in, ns, pre, mode)
        bad = np.argwhere(got != want)
        assert bad.size == 0, (c, bad[:4].tolist(), mode.tolist(), ns.tolist(), pre.tolist())
    b.close()


def test_tail_only_step_needs_a_frame_step_of_the_same_call(blob_f32, hip_lib):
    """lpcnet_batch_synthesize_step keeps frame products per stream only for its own mode-1 steps; a tail-only step of
    a stream that was advanced by the ordinary calls (or never) is refused up front instead of synthesising from stale products"""
    n = 3
    b = api.LPCNetBatch(n, blob_f32)
    feats = np.stack([synth.make_features(6100 + s, 3) for s in range(n)])
    one = np.ascontiguousarray(feats[:, 0])
    pcm = np.zeros((n, 160), np.int16)
    ns, pre = np.full(n, 160, np.int32), np.zeros(n, np.int32)
    with pytest.raises(api.LPCNetError):
        b.synthesize_step(one, pcm, ns, pre, np.array([2, 0, 0], np.int32))
    b.synthesize_step(one, pcm, ns, pre, np.array([1, 1, 0], np.int32))
    b.synthesize_step(one, pcm, ns, pre, np.array([2, 2, 0], np.int32))          # fine: both had their frame step
    with pytest.raises(api.LPCNetError):
        b.synthesize_step(one, pcm, ns, pre, np.array([0, 0, 2], np.int32))
    b.synthesize(feats[:, 1:2])                                                  # the ordinary path: the kept products are stale now
    with pytest.raises(api.LPCNetError):
        b.synthesize_step(one, pcm, ns, pre, np.array([2, 0, 0], np.int32))
    b.close()


def test_matrix_pipe_multiplier_and_packed_math_are_exact(hip_lib):
    """PARITY's products come from v_mfma_f32_4x4x1 with C = -0.0 (S >= 2) and its sums / GRU-B products from v_pk_add_f32 /
    v_pk_mul_f32: all three must be the vector unit's v_mul_f32 / v_add_f32 bit for bit, and v_mul_f32 must be IEEE binary32
    multiplication -- over > 10^7 operand pairs stratified to subnormal inputs and products, underflow, +-0, the largest
    finite products, overflow and infinities (src/vec.h:347-404: product, then add, separately rounded)."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import arith_identities
    rep = arith_identities.run(262144)
    assert rep["total"]["products"] >= 10 ** 7
    bad = {k: v for k, v in rep["strata"].items() if v["mfma_c_negzero_vs_v_mul_f32_mismatches"] or v["v_mul_f32_vs_ieee_host_mismatches"] or v["v_pk_mul_add_vs_scalar_mismatches"]}
    assert not bad, bad
    # the strata do reach the corners they are named after
    assert rep["strata"]["subnormal state x normal weight in [1/128, 8)"]["results_subnormal"] > 10 ** 5
    assert rep["strata"]["normal x normal, product in the subnormal range"]["results_subnormal"] > 10 ** 5
    assert rep["strata"]["largest finite products and overflow"]["results_inf"] > 10 ** 5


@pytest.mark.parametrize("S", [4, 2, 1])
def test_gru_state_in_the_subnormal_range_end_to_end(blob_f32, S, hip_lib):
    """A GRU state that has decayed into the subnormal range (reachable on silence): every GRU-A product of the next sample is
    subnormal or underflows -- on the matrix pipe at S >= 2 -- and the engine must still follow the reference bit for bit."""
    n, T = 4, 3
    feats = feats_for(range(7300, 7300 + n), 2 * T)
    rng = np.random.default_rng(11)
    ga = (rng.uniform(0.2, 9.0, (n, 384)) * 1e-39 * rng.choice([-1.0, 1.0], (n, 384))).astype(np.float32)
    gb = (rng.uniform(0.2, 9.0, (n, 16)) * 1e-39 * rng.choice([-1.0, 1.0], (n, 16))).astype(np.float32)
    ga[:, ::7] = 0.0; ga[:, 3::11] = -0.0; ga[:, 5::13] = np.float32(1e-45)
    assert (np.abs(ga[ga != 0]) < np.finfo(np.float32).tiny).all()
    om = orc.OracleModel(blob_f32)
    b = api.LPCNetBatch(n, blob_f32)
    b.streams_per_workgroup = S
    b.synthesize(np.ascontiguousarray(feats[:, :T]))              # past the start-up frames
    want = np.zeros((n, T * 160), np.int16)
    states = []
    for s in range(n):
        o = om.new_state()
        o.synthesize(feats[s, :T])
        o.L.orc_set_gru_state(o.p, np.ascontiguousarray(ga[s]), np.ascontiguousarray(gb[s]))
        want[s] = o.synthesize(feats[s, T:])
        states.append(o)
        st = b.get_state(s)
        st.gru_a[:] = ga[s].tolist(); st.gru_b[:] = gb[s].tolist()
        assert np.array_equal(np.array(st.gru_a, np.float32).view(np.uint32), ga[s].view(np.uint32))      # (ctypes keeps subnormals and -0.0)
        b.set_state(s, st)
    got = b.synthesize(np.ascontiguousarray(feats[:, T:]))
    assert np.array_equal(got, want)
    for s in range(n):
        st = b.get_state(s)
        _, _, oa, ob = states[s].nnet_state()
        assert np.array_equal(np.array(st.gru_a, np.float32).view(np.uint32), oa.view(np.uint32))
        assert np.array_equal(np.array(st.gru_b, np.float32).view(np.uint32), ob.view(np.uint32))
    b.close()


def test_void_entry_points_fail_soft_with_a_sticky_status(blob_f32, hip_lib, monkeypatch):
    """SURVEY §8b "Errors": a void entry point that cannot run zero-fills its output, leaves the state alone and records a sticky
    per-thread status instead of aborting the process (here: a state with no model and no default model anywhere)."""
    monkeypatch.chdir(os.path.dirname(os.path.abspath(__file__)))     # no ./weights_blob.bin here
    monkeypatch.delenv("LPCNET_HIP_MODEL", raising=False)
    monkeypatch.setenv("LPCNET_HIP_QUIET", "1")
    code = (
        "import sys, numpy as np; sys.path.insert(0, %r)\n"
        "import torch\n"
        "from lpcnet_amd import api\n"
        "st = api.LPCNetState()\n"
        "before = st.raw_bytes()\n"
        "out = np.full(160, 77, np.int16)\n"
        "api.load_library().lpcnet_synthesize(st.p, np.zeros(20, np.float32), out, 160)\n"
        "assert not out.any(), 'output must be zero-filled'\n"
        "assert api.status() == -5 and 'no model' in api.last_error(), (api.status(), api.last_error())\n"
        "assert st.raw_bytes() == before\n"
        "ga, gb, lpc = st.run_frame_network(np.zeros(20, np.float32))\n"
        "assert not ga.any() and not lpc.any() and api.status() == -5\n"
        "o2 = st.synthesize_tail_impl(40, np.arange(8, dtype=np.int16))\n"
        "assert list(o2[:8]) == list(range(8)) and not o2[8:].any()\n"
        "api.clear_error(); assert api.status() == 0 and api.last_error() == ''\n"
        "print('soft-fail ok')\n" % os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import subprocess
    import sys
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "soft-fail ok" in r.stdout, (r.stdout[-500:], r.stderr[-1500:])
    # the old behaviour stays available for callers that prefer to stop
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=dict(os.environ, LPCNET_HIP_ABORT_ON_ERROR="1"))
    assert r.returncode != 0 and "no model bound" in r.stderr
    # a model that works keeps working afterwards, with a clean status
    st = api.LPCNetState(blob_f32)
    api.clear_error()
    assert st.synthesize(feats_for([7400], 3)[0, 0]).shape == (160,) and api.status() == 0
    assert hip_lib.lpcnet_hip_model_status(st.p, 0) == 0


def test_int8_requantisation_instruction_is_the_reference_formula_on_every_float(hip_lib):
    """src/vec.h:311-316 quantises the GRU states as (int)floor(.5 + 127 x) with the sum in double; the int8 kernels may use ONE instruction,
    v_cvt_rpi_i32_f32 (LPCN_QUANT_RPI).  The device compares both on all 2^32 float bit patterns: not one may differ where the C
    expression is defined (finite, |t| < 2^31), in particular none inside the reachable range |t| <= 127.5."""
    bad_all, bad_reachable, pattern = api.quant_sweep()
    assert bad_reachable == 0, (bad_reachable, hex(pattern))
    print("v_cvt_rpi_i32_f32 vs floor(.5 + (double)t): mismatches over all finite |t| < 2^31:", bad_all, hex(pattern))
