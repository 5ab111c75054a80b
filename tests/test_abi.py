"""The drop-in boundary without a GPU: liblpcnet_hip.so loads, exports every symbol that
include/lpcnet.h and include/lpcnet_batch.h declare, validates blobs like the reference loader and
fails loudly (never silently falls back) when no HIP device is usable."""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from lpcnet_amd import api, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"LPCNET_EXPORT[^;(]*?\b((?:lpcnet_|run_frame_network)\w*)\s*\(", text)))


def test_every_declared_symbol_is_exported(hip_lib):
    names = _declared("lpcnet.h") + _declared("lpcnet_batch.h")
    assert len(names) > 30
    out = subprocess.check_output(["nm", "-D", "--defined-only", api.LIB_PATH], text=True)
    exported = {line.split()[-1] for line in out.splitlines() if line.strip()}
    missing = [n for n in names if n not in exported]
    assert not missing, missing
    # the reference's synthesis + decoder entry points (include/lpcnet.h:67-96,160-214 of the reference)
    for n in ("lpcnet_get_size", "lpcnet_init", "lpcnet_reset", "lpcnet_create", "lpcnet_destroy", "lpcnet_synthesize",
              "lpcnet_load_model", "lpcnet_decoder_get_size", "lpcnet_decoder_init", "lpcnet_decoder_create",
              "lpcnet_decoder_destroy", "lpcnet_decode"):
        assert n in exported
    # nothing of the internal engine ABI leaks out of the shared object
    assert not [s for s in exported if s.startswith("lpcn_")]


def test_state_sizes_and_init(hip_lib):
    L = hip_lib
    assert L.lpcnet_get_size() > 3000 and L.lpcnet_decoder_get_size() > L.lpcnet_get_size()
    assert L.lpcnet_batch_state_size() == C.sizeof(api.StreamState)
    buf = C.create_string_buffer(L.lpcnet_get_size())
    assert L.lpcnet_init(buf) == 0                                   # caller-allocated state (no create/destroy)
    st = api.LPCNetState()
    raw = (C.c_char * L.lpcnet_get_size()).from_address(st.p)
    s = api.StreamState.from_buffer_copy(raw.raw[8:8 + C.sizeof(api.StreamState)])
    assert s.last_exc == 128 and s.frame_count == 0                  # lin2ulaw(0); src/lpcnet.c:180
    assert list(s.rng) != [0, 0, 0, 0]


def test_reset_seeds_rng_like_reference(hip_lib, golden):
    st = api.LPCNetState()
    raw = (C.c_char * hip_lib.lpcnet_get_size()).from_address(st.p)
    s = api.StreamState.from_buffer_copy(raw.raw[8:8 + C.sizeof(api.StreamState)])
    assert np.array_equal(np.array(list(s.rng), np.uint32), golden["kiss99_seeded"])


def test_check_model_accepts_and_rejects(blob_f32, blob_i8):
    rc, info = api.check_model(blob_f32)
    assert rc == 0 and info[:3] == [0, 1382, 576] and info[5] == 0 and 0 < info[3] <= 40
    rc, info = api.check_model(synth.blob_bytes(synth.make_model(off_grid=True)))                 # weights off the k/128 grid
    assert rc == 0 and info[:3] == [0, 1382, 576] and info[5] == 0 and 0 < info[3] <= 40
    rc, info = api.check_model(blob_i8)
    assert rc == 0 and info[:3] == [1, 1382, 576] and info[5] == 0 and 0 < info[3] <= 64      # int8 packing self-check
    assert api.check_model(blob_f32[:-64])[0] == -1                 # last record truncated
    assert api.check_model(b"")[0] == -1
    bad = bytearray(blob_f32)
    bad[63] = 65                                                    # record name not NUL-terminated
    assert api.check_model(bytes(bad))[0] == -1
    # an index stream whose positions are not multiples of 4 (src/parse_lpcnet_weights.c:104)
    m = synth.make_model()
    idx = m.get("gru_b_weights_idx").copy()
    idx[1] = 2
    m.add("gru_b_weights_idx", idx, synth.WEIGHT_TYPE_INT)
    assert api.check_model(synth.blob_bytes(m))[0] == -1
    # a missing array
    m = synth.make_model()
    del m.arrays["dual_fc_factor"]
    assert api.check_model(synth.blob_bytes(m))[0] == -1


@pytest.mark.parametrize("seed,dens", [(7, (0.05, 0.05, 0.2)), (8, (0.02, 0.1, 0.25)), (9, (0.1, 0.1, 0.1))])
def test_packing_roundtrip_other_models(seed, dens):
    blob = synth.blob_bytes(synth.make_model(seed=seed, densities=dens))
    rc, info = api.check_model(blob)
    assert rc == 0 and info[5] == 0


def _layout(blob):
    out = (C.c_int * 65)()
    assert api.load_library().lpcnet_hip_model_layout(blob, len(blob), out) == 0
    o = list(out)
    waves = [dict(bound=o[1 + w * 7:1 + w * 7 + 4], allh=o[1 + w * 7 + 4:1 + w * 7 + 7], head=o[57 + w]) for w in range(8)]
    return o[0], waves


def test_candidate_heads_sit_where_the_kernel_runs_them(blob_f32, blob_i8):
    """the dealing of GRU-A (model_pack.c): the head of a wave's candidate chains is stored end-aligned behind its other items
    (bound[3] + head <= items per lane); float blobs give heads only to the waves that never run GRU-B (4..7); int8 blobs (two
    streams per workgroup is their operating point) to every wave except GRU-B's chain waves there -- waves 2 and 3 since round 5
    (lpcnet_engine.h: LPCN_I8_GBWA / _GBWB); a head belongs to a candidate-only first slot."""
    for blob, headless, headed in ((blob_f32, (0, 1, 2, 3), (4, 5, 6, 7)), (blob_i8, (2, 3), (0, 1, 4, 5, 6, 7))):
        nw, waves = _layout(blob)
        assert 1 <= nw <= 64
        heads = [w["head"] for w in waves]
        assert all(heads[w] == 0 for w in headless) and all(0 < heads[w] <= 24 for w in headed), heads
        for w in waves:
            b = w["bound"]
            assert b[0] == 0 and b[0] <= b[1] <= b[2] <= b[3] and b[3] + w["head"] <= nw, (w, nw)
            if w["head"]:
                assert w["allh"][0] == 1, w


def test_no_gpu_means_loud_failure_not_fallback(hip_lib, blob_f32):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    st = api.LPCNetState()
    with pytest.raises(api.LPCNetError) as e:
        st.load_model(blob_f32)
    assert "no HIP device" in str(e.value) or "hip" in str(e.value).lower()
    with pytest.raises(api.LPCNetError):
        api.LPCNetBatch(4, blob_f32)


def test_product_does_not_link_or_import_the_oracle():
    out = subprocess.check_output(["ldd", api.LIB_PATH], text=True)
    assert "oracle" not in out and "lpcnet_ref" not in out
    for root, _, files in os.walk(os.path.join(ROOT, "lpcnet_amd")):
        for f in files:
            if f.endswith((".py", ".c", ".h", ".hip")):
                text = open(os.path.join(root, f), errors="ignore").read()
                assert "import oracle" not in text and "from oracle" not in text and "lpcnet_oracle.h" not in text, f


def test_index_stream_with_hostile_count_is_rejected(blob_f32):
    """an index stream whose block count is INT_MAX must not overflow the bounds check"""
    m = synth.make_model()
    idx = m.get("gru_b_weights_idx").copy()
    idx[0] = 0x7FFFFFFF
    m.add("gru_b_weights_idx", idx, synth.WEIGHT_TYPE_INT)
    assert api.check_model(synth.blob_bytes(m))[0] == -1


def test_bench_refuses_a_rank_count_it_cannot_get():
    """`python bench.py --gpus N` must end up with N ranks or fail loudly -- never print an n_gpus: 1 line for N > 1"""
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "64"], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode != 0 and "HIP device" in (r.stderr + r.stdout) and '"n_gpus"' not in r.stdout
    env["WORLD_SIZE"] = "2"; env["RANK"] = "0"; env["LOCAL_RANK"] = "0"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode != 0 and "WORLD_SIZE" in (r.stderr + r.stdout)


@pytest.mark.gpu
def test_bench_multi_rank_path_rehearsed_on_one_gpu():
    """`bench.py --gpus 2 --share-device`: the launcher, the rank environment, the barrier + max-over-ranks timing, the
    per-rank feature seeds and the rank-0 JSON line of the multi-rank branch, executed on the one GPU the box has (both
    ranks on device 0, gloo control plane).  The first real SCALE run must not be the first execution of this code."""
    import json
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    common = ["--steps", "2", "--warmup", "1", "--streams", "256", "--frames", "10", "--no-cpu-baseline"]
    r1 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + common, capture_output=True, text=True, env=env, timeout=900)
    assert r1.returncode == 0, r1.stderr[-2000:]
    one = json.loads(r1.stdout.strip().splitlines()[-1])
    r2 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--share-device"] + common,
                        capture_output=True, text=True, env=env, timeout=900)
    assert r2.returncode == 0, (r2.stdout[-1000:], r2.stderr[-2000:])
    lines = [ln for ln in r2.stdout.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1                                   # rank 0 only
    two = json.loads(lines[0])
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2 and two["parity_checked"] == 8 and "rehearsal" in two
    assert two["config"]["sharding"].startswith("2 x 256")
    # two ranks share one GPU: the aggregate is the one-GPU figure again (256 streams leave most CUs idle, so two such
    # batches overlap almost perfectly).  A sanity bound on a shared GPU, deliberately loose: the structural
    # asserts above are the gate.
    assert 0.5 * one["value"] < two["value"] < 3.0 * one["value"], (one["value"], two["value"])


@pytest.mark.gpu
@pytest.mark.parametrize("flag", [[], ["--int8"]], ids=["config3-fp32", "config4-int8"])
def test_bench_eight_rank_shape_rehearsed_on_one_gpu(flag):
    """BASELINE configs 3 / 4 are 8 192 streams over 8 ranks; nothing of that shape had ever executed.  All eight
    ranks on the one GPU the box has (gloo control plane, 8 x 1024 distinct streams, every rank its own engine and model
    copy): the launcher, the rendezvous, the barrier + max-over-ranks timing, rank 0's parity check and JSON line.  No
    scaling number can come out of eight processes time-sharing one device -- the line says so (`rehearsal`)."""
    import json
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--share-device", "--steps", "2", "--warmup", "1",
                        "--streams", "1024", "--no-cpu-baseline"] + flag, capture_output=True, text=True, env=env, timeout=1500)      # (configs 3 / 4: 8 x 1024; the bench's default is 2048 per GPU since round 6)
    assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-2000:])
    lines = [ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["parity_checked"] == 8 and "rehearsal" in d and d["config"]["sharding"].startswith("8 x 1024")
    assert d["value"] > 0 and d["scaling"] == "weak"


def test_library_carries_the_hash_of_the_sources_it_was_built_from(hip_lib):
    """the .so travels to the GPU box prebuilt; the source hashes baked into it (and echoed by bench.py) prove which
    tree it was built from.  `dev` covers the device sources only, so a host-only change does not orphan the profiles."""
    from lpcnet_amd import build
    src, dev = build.source_hashes()
    assert build.baked_hashes(api.LIB_PATH) == (src, dev)
    assert api.build_info() == {"src": src, "dev": dev}
    import bench
    assert bench.kernel_source_hash() == dev
    # a host-only edit moves `src` but not `dev`
    import hashlib
    host_only = [f for f in os.listdir(build.CSRC) if f.endswith(".c")]
    assert sorted(host_only) == ["api.c", "model_pack.c"]
    h = hashlib.sha1()
    for f in sorted(os.listdir(build.CSRC)):
        if f.endswith((".h", ".hip", ".inc")):
            h.update(f.encode() + open(os.path.join(build.CSRC, f), "rb").read())
    flags = " ".join(x for x in build.HIP_FLAGS if not x.startswith("-I")) + " | "
    h.update(flags.encode())
    assert h.hexdigest()[:16] == dev


def test_void_entry_points_do_not_abort_without_a_device_or_model(hip_lib, blob_f32, tmp_path):
    """SURVEY 8b "Errors": lpcnet_synthesize & co. return void in the reference; on this engine a failing call zero-fills its output and
    sets a sticky status instead of aborting the process (a child process: an abort would take the test run down)."""
    code = (
        "import sys, os, numpy as np; sys.path.insert(0, %r)\n"
        "from lpcnet_amd import api\n"
        "st = api.LPCNetState()\n"
        "out = np.full(160, 77, np.int16)\n"
        "api.load_library().lpcnet_synthesize(st.p, np.zeros(20, np.float32), out, 160)\n"
        "assert not out.any() and api.status() < 0 and api.last_error(), (api.status(), api.last_error())\n"
        "o2 = st.synthesize_tail_impl(40, np.arange(8, dtype=np.int16))\n"
        "assert list(o2[:8]) == list(range(8)) and not o2[8:].any()\n"
        "o3 = np.full(10, 5, np.int16)\n"
        "api.load_library().lpcnet_synthesize_tail_impl(st.p, o3, 10, 11)\n"        # preload > N: refused, zero-filled, no abort
        "assert not o3.any() and 'preload' in api.last_error()\n"
        "first = api.status()\n"
        "api.clear_error(); assert api.status() == 0 and api.last_error() == ''\n"
        "print('ok', first)\n" % ROOT)
    env = dict(os.environ, LPCNET_HIP_QUIET="1")
    env.pop("LPCNET_HIP_MODEL", None)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, cwd=str(tmp_path), env=env)
    assert r.returncode == 0 and r.stdout.startswith("ok -"), (r.stdout[-300:], r.stderr[-1500:])
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, cwd=str(tmp_path), env=dict(env, LPCNET_HIP_ABORT_ON_ERROR="1"))
    assert r.returncode != 0 and "lpcnet_synthesize" in r.stderr            # the old stop-with-a-message, on request
