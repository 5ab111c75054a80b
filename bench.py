#!/usr/bin/env python3
"""Benchmark of the LPCNet synthesis hot path on MI355X (driver contract: one JSON line on rank 0).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

step     = one pass of the hot path (frame network + LPC + 160-sample loop per frame) over one
           batch of synthetic feature frames: STREAMS streams x FRAMES frames per GPU, features and
           PCM resident in HBM.  Round 6: 2048 concurrent streams per GPU -- the two-group sample kernel carries EIGHT float
           streams per CU (256 CUs x 8), so BASELINE.json config 2's 1024 streams fill only half of its stream slots; the
           1024-stream figure of the same build rides along as `also.config2_1024_streams` (four streams per CU, the
           round-5 kernel).  Weak scaling: every GPU gets its own streams.
also     = (N = 1) the same step with BASELINE config 4's int8 GRU weights (`also.int8_parity`, bit-exact against the reference's
           generic int8 build) and config 2's 1024 streams, each oracle-checked like the headline.
rt       = (N = 1) the half of the metric that is a DEADLINE: frame-at-a-time synthesis at the stream counts that bracket the 10-ms
           frame period (`rt.probe`), host wall time AND device time of every step; `realtime_streams_sustained` is the largest
           measured count whose p99.9 step stays under 10 ms.
value    = whole-job 16 kHz samples per second (sum over GPUs / max-over-ranks time);
           concurrent real-time streams = value / 16000.
           Every stream of every rank has its own seeded feature file (1000 + rank*streams + s); after the timed loop
           `--check-streams` streams of the timed output are replayed on the plain-C oracle (checker leg) and
           compared bit for bit: `parity_checked` in the JSON line.
           `--gpus N` without a launcher re-executes itself under `python -m torch.distributed.run` with N ranks;
           `--gpus 2 --share-device` rehearses that path on ONE GPU (both ranks on device 0, gloo control plane).
roofline = the sample kernel (dominant, >98 % of the step).  SURVEY.md §8(d): not MFMA; the primary bound
           is on-chip operand bandwidth, so `achieved` = algorithmic operand bytes per stream-sample
           (286 704 B fp32 / 96 432 B int8: every weight and table entry once per sample) x samples per
           launch / launch time, against the 150 TB/s LDS peak; the fp32-VALU flop fraction (129 698 flop
           per sample vs 157.3 TFLOP/s) and the L2-side gather rate (13 857 B per sample, L2 -> CU, NOT HBM)
           ride along; `traffic` = HBM bytes per launch from the rocprofv3 PMC passes of this very command
           (profiles/, FETCH_SIZE x2 + WRITE_SIZE; it cannot be collected from inside the process) and
           `hbm` = that traffic over the live launch time against the 8 TB/s peak.  Launch time is
           measured live with HIP events on the stream the kernel runs on.
cpu_baseline = the reference's own AVX2 builds (oracle/_ref, `kind: reference`): A-f = float (-DDISABLE_DOT_PROD, the
           arithmetic the fp32 line is compared with) and A-i = int8, the reference's default x86 build; `value`
           is A-f for the fp32 line and A-i for the --int8 line, both are listed under `flavours`.  If the
           prebuilt libraries are absent: the plain-C oracle (`kind: port`).  Timed here on the host cores over a
           bounded sample.  The oracle is only the baseline/checker, never the product.
"""
import argparse
import json
import multiprocessing as mp
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

STREAMS_PER_GPU = 2048                  # eight float streams per CU x 256 CUs (round 6; BASELINE config 2 = 1024 rides along under `also`)
CONFIG2_STREAMS = 1024
FRAMES_PER_STEP = 25                    # 0.25 s of audio per stream and step
FLOP_PER_SAMPLE = 129698                # 64 849 MAC
L2_BYTES_PER_SAMPLE = 13857             # embedding gather 13 824 (L2 -> CU) + PCM 2 + frame I/O 31
LDS_OPERAND_BYTES_PER_SAMPLE = 286704   # every fp32 operand once per stream-sample
LDS_OPERAND_BYTES_PER_SAMPLE_I8 = 96432 # int8 GRU-A 59 544 + int8 GRU-B 21 912 + fp32 tree 1 152 + embedding rows 13 824
PEAK_FP32_TFLOPS = 157.3                # MI355X_MICROARCH.md: fp32 vector peak
PEAK_HBM_GBS = 8000.0
PEAK_LDS_TBS = 150.0
MEASURED_FP32_MUL_ADD_TFLOPS = 66.9      # tools/ubench/peaks.hip (profiles/r04_roofline_measured.json): separately rounded multiply + add, no FMA,
                                        # as PACKED math (v_pk_mul_f32 + v_pk_add_f32: 66.9; scalar v_mul_f32 + v_add_f32: 59.5) -- the VALU-only
                                        # ceiling of the PARITY arithmetic (the kernel also forms exact products on the matrix pipe)


def _cpu_worker(args):
    """One single-threaded reference process pinned to its own core: untimed warm-up, then best of `reps` timed passes."""
    kind, frames, seed, flavour, cpu, reps = args
    if cpu is not None:
        try:
            os.sched_setaffinity(0, {cpu})
        except Exception:
            pass
    from lpcnet_amd import synth
    blob = synth.blob_bytes(synth.make_model(flavour="int8" if flavour in ("ai", "ni") else "float"))
    f = synth.make_features(seed, frames)
    if kind == "reference":
        from oracle import ref
        lib = ref.RefLib(flavour)
        new_state = lambda: lib.new_state(blob)
    else:
        from oracle import orc
        om = orc.OracleModel(blob)
        new_state = om.new_state
    new_state().synthesize(f[:min(frames, 100)])             # warm-up: page in the model, the tables, the code
    best = None
    for _ in range(reps):
        st = new_state()
        t0 = time.perf_counter()
        st.synthesize(f)
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    return best


def usable_cores():
    """Host cores this process may actually use: affinity mask capped by the cgroup CPU quota."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if q > 0:
                    n = min(n, max(1, q // per))
        except Exception:
            pass
    return n


def cpu_model():
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.lower().startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


def _flavour_runs(flavour):
    """Can this box execute the flavour at all?  (the -march=native builds carry the BUILD host's instruction set)"""
    import subprocess
    code = ("import sys; sys.path.insert(0, %r); import bench; "
            "print(bench._cpu_worker(('reference', 4, 1000, %r, None, 1)))" % (ROOT, flavour))
    try:
        return subprocess.run([sys.executable, "-c", code], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=120).returncode == 0
    except Exception:
        return False


def cpu_baseline(int8_line=False):
    """Reference CPU path on this box's host cores.  The library is single-threaded, so the aggregate is `procs` independent
    processes, each PINNED to its own core (an unpinned pool under a CPU quota gave a 1.7x run-to-run spread), one untimed
    warm-up pass and the best of three timed passes per process.  Reported per flavour: the sum of the per-process rates, the
    median rate x procs and their ratio (`unstable` when they differ by more than 20 %), and one process alone on one core.
    Flavours: A-f / A-i = -O3 -mavx2 -mfma float / int8 (the reference's default x86 build, src/vec_avx.h:39-41);
    N-f / N-i = -Ofast -march=native as README.md:55-57 recommends (native to the host they were BUILT on; skipped when this
    box cannot execute them)."""
    from oracle import ref
    have = ref.available("af") and ref.available("ai")
    kind = "reference" if have else "port"
    cores = usable_cores()
    try:
        allowed = sorted(os.sched_getaffinity(0))
    except AttributeError:
        allowed = list(range(os.cpu_count() or 1))
    procs = max(1, min(cores, 32, len(allowed)))
    if procs > 2 and procs == cores:
        procs -= 1                                           # leave the quota's last core to this (parent) process
    pins = [allowed[(i * len(allowed)) // procs] for i in range(procs)]       # spread over the mask (distinct physical cores where possible)
    # the parent (pool bookkeeping, result pickling) moves to a core the pool does not use, so that no worker shares its core with it
    # (one straggler per int8 pool); restored afterwards
    spare = [c for c in allowed if c not in pins]
    old_mask = None
    if spare:
        try:
            old_mask = os.sched_getaffinity(0)
            os.sched_setaffinity(0, {spare[-1]})
        except Exception:
            old_mask = None
    names = {"af": "reference AVX2+FMA float build (oracle/_ref af: -O3 -mavx2 -mfma -DDISABLE_DOT_PROD)",
             "ai": "reference AVX2 int8 build (oracle/_ref ai: -O3 -mavx2 -mfma, the reference's default on x86)",
             "nf": "reference float build with -Ofast -march=native (oracle/_ref nf, README.md:55-57; native to the build host)",
             "ni": "reference int8 build with -Ofast -march=native (oracle/_ref ni, README.md:55-57; native to the build host)",
             "port": "plain-C oracle"}
    todo = [fl for fl in ("af", "ai", "nf", "ni") if ref.available(fl)] if have else ["port"]
    flav = {}
    for fl in todo:
        if fl in ("nf", "ni") and not _flavour_runs(fl):
            flav[fl] = {"value": None, "build": names[fl], "note": "not executable on this host (built with -march=native elsewhere)"}
            continue
        frames = 1000 if have else 250                       # ~1.3 s (A-f) / ~0.7 s (A-i) / ~1.5 s (plain C) per pass
        t0 = time.perf_counter()
        with mp.get_context("spawn").Pool(procs) as pool:
            times = pool.map(_cpu_worker, [(kind, frames, 1000 + i, fl, pins[i], 3) for i in range(procs)])
        wall = time.perf_counter() - t0
        with mp.get_context("spawn").Pool(1) as pool:        # one process alone on one pinned core
            single = pool.map(_cpu_worker, [(kind, frames, 1000, fl, pins[0], 3)])[0]
        n_s = (frames - 2) * 160
        rates = [n_s / tt for tt in times]
        total, med = float(sum(rates)), float(np.median(rates)) * procs
        flav[fl] = {"value": total, "median_x_procs": med, "sum_over_median_x_procs": total / med, "unstable": bool(abs(total / med - 1.0) > 0.2),
                    "per_core": float(np.median(rates)), "per_core_min": float(min(rates)), "per_core_max": float(max(rates)),
                    "single_process_pinned": n_s / single, "realtime_factor_single": n_s / single / 16000.0,
                    "frames_per_process": frames, "passes": "1 warm-up + best of 3", "pool_wall_s": round(wall, 1), "build": names[fl]}
    if old_mask is not None:
        try:
            os.sched_setaffinity(0, old_mask)
        except Exception:
            pass
    main = ("ai" if int8_line else "af") if have else "port"
    frames = flav[main]["frames_per_process"]
    return {"value": flav[main]["value"], "unit": "samples/s", "cores": procs, "kind": kind, "per_core": flav[main]["per_core"],
            "unstable": flav[main]["unstable"], "cpu_model": cpu_model(), "flavour": main, "flavours": flav,
            "sample": f"{procs} independent single-threaded processes, each pinned to its own core, x {frames} frames ({frames / 100:.1f} s of audio), "
                      f"warm-up + best of 3, {names[main]}; {cores} usable host cores of {os.cpu_count()} ({cpu_model()})"}


def kernel_source_hash():
    """sha1 over the DEVICE sources (*.hip, *.hip.h, *.inc, the headers they include, the device compile flags): ties a PMC
    traffic file under profiles/ to the kernels it was measured with.  Host-only changes (api.c, model_pack.c) do not move it
    (a comment-only commit to api.c orphaned the round's HBM figures)."""
    from lpcnet_amd import build
    return build.source_hashes()[1]


def fast_envelope_check(blob, feats, int8, fp16_fc, spw, device):
    """FAST lines carry their own check: FAST is not bit-identical to any reference build, so it is validated TEACHER-FORCED -- a PARITY engine
    and a FAST engine are fed the same signal (PARITY's free-running output for these very streams' feature files) frame by frame through the preload of
    lpcnet_synthesize_impl (src/lpcnet.c:256-259), and the per-frame maximum state deviation must stay inside what the reference's own AVX2 builds show against its
    generic-C builds under the same protocol (tests/golden/simd_envelope_v1.json, tests/tools/make_envelope.py; 1.25 x for the shorter run, like tests/test_gpu_fast.py).
    Returns (streams checked, worst GRU-A deviation, worst GRU-B deviation); raises if the envelope is exceeded."""
    from lpcnet_amd import api
    env = json.load(open(os.path.join(ROOT, "tests", "golden", "simd_envelope_v1.json")))["int8" if int8 else "float"]
    k, T = feats.shape[0], feats.shape[1]
    ref = api.LPCNetBatch(k, blob, device=device)
    pcm = ref.synthesize(feats)
    ref.close()

    def forced(fast):
        b = api.LPCNetBatch(k, blob, device=device)
        if spw in (1, 2, 4):
            b.streams_per_workgroup = spw
        if fast:
            b.set_fast(2 if fp16_fc else 1)
        ga, gb = np.zeros((T, k, 384), np.float32), np.zeros((T, k, 16), np.float32)
        for t in range(T):
            b.synthesize(np.ascontiguousarray(feats[:, t:t + 1]), preload_pcm=np.ascontiguousarray(pcm[:, t * 160:(t + 1) * 160]), preload=160)
            for i in range(k):
                st = b.get_state(i)
                ga[t, i] = np.array(st.gru_a, np.float32)
                gb[t, i] = np.array(st.gru_b, np.float32)
        b.close()
        return ga, gb
    (ga_p, gb_p), (ga_f, gb_f) = forced(False), forced(True)
    da = float(np.abs(ga_f - ga_p)[3:].max())
    db = float(np.abs(gb_f - gb_p)[3:].max())
    if not (np.abs(ga_p[3:]).max() > 0.3 and da <= 1.25 * env["gru_a"]["worst"] and (fp16_fc or db <= 1.25 * env["gru_b"]["worst"])):
        raise SystemExit(f"bench.py --fast: teacher-forced state deviation outside the reference's own SIMD envelope (GRU-A {da:.3g} / {env['gru_a']['worst']:.3g}, GRU-B {db:.3g} / {env['gru_b']['worst']:.3g})")
    return k, da, db


FRAME_DEADLINE_MS = 10.0                # one frame = 160 samples at 16 kHz
RT_PROBE_STREAMS = [7168, 8192]         # the default line's real-time probe: the counts that bracket the frame deadline (eight streams per CU: capacity moves in rounds of 2048)
RT_PROBE_STEPS = 200


def rt_measure(a, counts, steps, warm, world, rank, local, dev, blob, int8):
    """Real-time operating point: n streams advance ONE frame per step, like the reference's callers do (src/lpcnet_demo.c:203-219: one
    lpcnet_synthesize per 10-ms frame); the step is enqueued on the device-pointer API (frame kernels + sample kernel) and waited for.  Two
    clocks per step: the host's wall time around enqueue + wait -- what a server thread sees, the one the 10-ms frame period is held against
    -- and the DEVICE time of the same step (HIP events on the kernels' stream, lpcnet_batch_last_timing), so that a late step can be told
    apart: device time normal = the host (scheduler, driver) was late; device time long = the GPU was."""
    import torch
    import torch.distributed as dist
    from lpcnet_amd import api, synth
    stream = torch.cuda.current_stream().cuda_stream
    results = []
    for n in counts:
        batch = api.LPCNetBatch(n, blob, device=local)
        if a.fast:
            batch.set_fast(2 if a.fp16_fc else 1)
        if a.spw:
            batch.streams_per_workgroup = a.spw
        else:
            batch.tune()
        T = warm + steps
        k = 0 if a.fast else min(a.check_streams, n)
        pick = sorted({(i * n) // k + (i % 4 if n >= 4 * k else 0) for i in range(k)}) if k else []
        # every stream its own feature sequence; laid out [frame][stream][36] so that one step's features are contiguous.
        # (distinct files for the streams that are checked, and a pool of 64 files cycled over the rest: generating 8192 x 500-frame files
        # on the host would take longer than the measurement)
        pool = np.stack([synth.make_features(5000 + rank * 64 + i, T) for i in range(64)])
        d_feat = torch.from_numpy(pool).to(dev)[torch.arange(n, device=dev) % 64].permute(1, 0, 2).contiguous()      # [T][n][36]
        picked = {sidx: synth.make_features(9000 + rank * n + sidx, T) for sidx in pick}
        for sidx, f in picked.items():
            d_feat[:, sidx] = torch.from_numpy(f).to(dev)
        d_pcm = torch.zeros((T, n, 160), dtype=torch.int16, device=dev)
        if world > 1:
            dist.barrier()
        lat, devms = np.zeros(T), np.zeros(T)
        batch.enable_timing(True)                            # HIP events around the frame kernels and the sample kernel of EVERY step
        for t in range(T):
            t0 = time.perf_counter()
            batch.synthesize_device(d_feat[t].data_ptr(), 36, d_pcm[t].data_ptr(), 1, stream)
            torch.cuda.synchronize()
            lat[t] = (time.perf_counter() - t0) * 1e3
            ks = batch.last_timing()
            devms[t] = ks[0] + ks[1]
        batch.enable_timing(False)
        timed, tdev = lat[warm:], devms[warm:]
        parity = 0
        if rank == 0 and pick:
            from oracle import orc
            want = orc.synthesize_many(blob, np.stack([picked[sidx] for sidx in pick]))      # [k][T*160] from reset
            got = d_pcm[:, pick].cpu().numpy().transpose(1, 0, 2).reshape(len(pick), T * 160)
            if not np.array_equal(got, want):
                raise SystemExit(f"bench.py --rt: output of the timed steps differs from the CPU oracle ({n} streams)")
            parity = len(pick)
        p50, p99, mx = float(np.percentile(timed, 50)), float(np.percentile(timed, 99)), float(timed.max())
        late = np.nonzero(timed >= FRAME_DEADLINE_MS)[0]
        p999 = float(np.percentile(timed, 99.9))
        results.append({"streams": n, "steps": steps, "step_ms_p50": p50, "step_ms_p99": p99, "step_ms_p999": p999, "step_ms_max": mx,
                        "step_ms_mean": float(timed.mean()),
                        "deadline_ms": FRAME_DEADLINE_MS, "meets_deadline_p99": bool(p99 < FRAME_DEADLINE_MS), "meets_deadline_p999": bool(p999 < FRAME_DEADLINE_MS),
                        "over_deadline_steps": int(late.size),
                        "device_ms_p50": float(np.percentile(tdev, 50)), "device_ms_p99": float(np.percentile(tdev, 99)), "device_ms_max": float(tdev.max()),
                        # every late step with both clocks: [step, wall ms, device ms] (at most 16 listed)
                        "late_steps": [[int(i), float(timed[i]), float(tdev[i])] for i in late[:16]],
                        "late_steps_with_normal_device_time": int(sum(1 for i in late if tdev[i] < 1.05 * np.percentile(tdev, 50))),
                        "realtime_factor_p99": FRAME_DEADLINE_MS / p99,
                        "samples_per_s_at_p50": n * 160 / (p50 * 1e-3),
                        "streams_per_workgroup": batch.streams_per_workgroup, "parity_checked": parity})
        batch.close()
        del d_feat, d_pcm
    if world > 1:
        for r in results:                                    # every rank must hold the deadline: the worst rank's figures
            t = torch.tensor([r["step_ms_p50"], r["step_ms_p99"], r["step_ms_max"], r["step_ms_p999"]], dtype=torch.float64, device="cpu" if a.share_device else dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            r["step_ms_p50"], r["step_ms_p99"], r["step_ms_max"], r["step_ms_p999"] = (float(x) for x in t.tolist())
            r["meets_deadline_p99"] = bool(r["step_ms_p99"] < FRAME_DEADLINE_MS)
            r["meets_deadline_p999"] = bool(r["step_ms_p999"] < FRAME_DEADLINE_MS)
        dist.barrier()
    return results


def rt_main(a, world, rank, local, dev):
    """`bench.py --rt`: the real-time operating point as a line of its own (see rt_measure)."""
    import torch.distributed as dist
    from lpcnet_amd import synth
    counts = [int(x) for x in a.rt_sweep.split(",") if x] or [a.streams]
    steps, warm = max(a.steps, 1), max(a.warmup, 3)
    blob = synth.blob_bytes(synth.make_model(flavour="int8" if a.int8 else "float"))
    results = rt_measure(a, counts, steps, warm, world, rank, local, dev, blob, a.int8)
    if rank == 0:
        ok = [r for r in results if r["meets_deadline_p999"]]
        best = max(ok, key=lambda r: r["streams"]) if ok else None
        head = best or min(results, key=lambda r: r["streams"])
        from lpcnet_amd import api as _api
        out = {"metric": "16 kHz samples/sec & concurrent real-time streams, 1/2/4/8 MI355X",
               "value": world * head["streams"] * 160 / (head["step_ms_p50"] * 1e-3), "unit": "samples/s",
               "realtime_streams_sustained": world * best["streams"] if best else 0,
               "realtime_streams_sustained_note": "largest measured stream count per GPU whose p99.9 step time (one 10-ms frame for every stream, enqueue + wait; the worst step of a run shorter than 1000 steps) "
                                                  "stays under the 10-ms frame period, x GPUs; 0 = none of the measured counts does",
               "n_gpus": world, "steps": steps, "warmup": warm, "ms_per_step": head["step_ms_mean"],
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": "int8 weights/activations x f32 accumulate (GRU-A/GRU-B), f32 elsewhere" if a.int8 else "f32", "data": "synthetic",
               "parity_checked": head["parity_checked"],
               "config": {"workload": f"REAL-TIME operating point: {head['streams']} concurrent streams per GPU, ONE 10-ms frame per step and stream "
                                      "(frame network + LPC + 160 samples), each step enqueued and waited for; "
                                      + ("int8 GRU weights" if a.int8 else "fp32 weights") + ", " + ("FAST arithmetic" if a.fast else "bit-exact (PARITY) arithmetic"),
                          "arithmetic": "fast" if a.fast else "parity", "frames_per_step": 1, "streams_per_gpu": head["streams"],
                          "sharding": f"{world} x n independent streams, no data-path collective"},
               "rt": results, "library_build_info": _api.build_info(), "kernel_source_hash": kernel_source_hash(),
               "library_matches_sources": _api.build_info().get("dev") == kernel_source_hash()}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


def _self_launch(a):
    """`python bench.py --gpus N` without a launcher: become N ranks (one per GPU) under torch.distributed.run."""
    import socket
    import subprocess
    import torch
    have = torch.cuda.device_count()
    if have < a.gpus and not a.share_device:
        raise SystemExit(f"bench.py: --gpus {a.gpus} but only {have} HIP device(s) are visible")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    raise SystemExit(subprocess.call(cmd))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--streams", type=int, default=STREAMS_PER_GPU, help="streams per GPU")
    ap.add_argument("--frames", type=int, default=FRAMES_PER_STEP, help="frames per step")
    ap.add_argument("--spw", type=int, default=0, help="streams per workgroup (0 = auto)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="the default command without its `also` (1024 streams, int8 weights) and `rt` (real-time probe) records")
    ap.add_argument("--check-streams", type=int, default=8, help="streams of the timed output replayed on the CPU oracle (0 = none)")
    ap.add_argument("--fast", action="store_true",
                    help="FAST arithmetic (fused multiply-add / int32 accumulation like the reference's AVX2 builds): a separate, "
                         "non-bit-exact flavour; the headline is the default PARITY arithmetic")
    ap.add_argument("--fp16-fc", action="store_true", help="with --fast: the sampler's dual FC in fp16 (BASELINE.json config 4's wording)")
    ap.add_argument("--int8", action="store_true",
                    help="BASELINE.json config 4: int8 (DOT_PROD) GRU-A/GRU-B weights, bit-exact vs the reference's generic int8 build "
                         "(default: float32 weights, the configuration the metric is quoted on)")
    ap.add_argument("--densities", default="", help="GRU-A block densities 'z,r,h' of the synthetic model (default 0.05,0.05,0.2 = SURVEY.md section 8d); "
                    "e.g. 0.07,0.07,0.25 loads the 36-items-per-lane kernel, 0.08,0.08,0.3 the 40-item one (NOT the headline workload)")
    ap.add_argument("--skew", type=float, default=0.0, help="trained-like (heavy-tailed) block distribution of the synthetic GRU-A: synth.make_model(skew=...), "
                    "e.g. 0.1 -> row groups with up to 65 of 96 blocks, 48 empty ones (NOT the headline workload)")
    ap.add_argument("--rt", action="store_true",
                    help="the operating point the metric is NAMED after: frame-at-a-time synthesis (one lpcnet_synthesize per 10-ms frame and "
                         "stream, src/lpcnet_demo.c:203-219) of --streams concurrent streams against the 10-ms frame deadline -- every step is one "
                         "frame for every stream (frame network + LPC + 160 samples), enqueued and WAITED for; reports the per-step wall time "
                         "(p50 / p99 / max over --steps steps) and whether the batch keeps up with real time (p99 < 10 ms)")
    ap.add_argument("--rt-sweep", default="", help="with --rt: comma-separated stream counts to measure in one run (the line carries all of them; "
                    "`value` is the rate at the largest count whose p99 meets the deadline)")
    ap.add_argument("--share-device", action="store_true",
                    help="rehearsal of the multi-rank path on ONE GPU: every rank uses device 0 and the control plane is gloo "
                         "(RCCL cannot put two ranks on one device); the line it prints is a plumbing check, not a scaling number")
    a = ap.parse_args()

    import torch
    import torch.distributed as dist
    from lpcnet_amd import api, synth

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        _self_launch(a)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus != world:
        raise SystemExit(f"bench.py: --gpus {a.gpus} but WORLD_SIZE {world}: launch one rank per GPU "
                         f"(python -m torch.distributed.run --nproc-per-node {a.gpus} ... bench.py --gpus {a.gpus})")
    if a.share_device:
        local = 0
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo" if a.share_device else "nccl", rank=rank, world_size=world)      # RCCL: control plane only
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the LPCNet HIP engine has no CPU fallback")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    if a.rt:
        return rt_main(a, world, rank, local, dev)

    def run_workload(n, F, int8, steps, warmup, fast=False, fp16_fc=False, spw=0, densities="", check_streams=8, seed0=1000, skew=0.0):
        """`warmup` untimed + `steps` timed passes of the hot path over n streams x F frames (features / PCM resident in HBM), then the live kernel
        timing and the checker leg; returns the figures of one workload."""
        mk = dict(flavour="int8" if int8 else "float")
        if densities:
            mk["densities"] = tuple(float(x) for x in densities.split(","))
        if skew:
            mk["skew"] = skew
        blob = synth.blob_bytes(synth.make_model(**mk))
        batch = api.LPCNetBatch(n, blob, device=local)
        if fast:
            batch.set_fast(2 if fp16_fc else 1)
            check_streams = 0                                # FAST is validated teacher-forced (tests/test_gpu_fast.py), not bit for bit
        if spw:
            batch.streams_per_workgroup = spw
        else:
            batch.tune()                # measured now (PARITY) / table value (FAST): the timed calls below are enqueue-only and never measure
        # synthetic features: every stream of every rank has its own seeded feature file, resident in HBM
        feats = np.stack([synth.make_features(seed0 + rank * n + s, F) for s in range(n)])
        d_feat = torch.from_numpy(feats).to(dev)
        d_pcm = torch.zeros((n, F * 160), dtype=torch.int16, device=dev)
        stream = torch.cuda.current_stream().cuda_stream

        def step():
            batch.synthesize_device(d_feat.data_ptr(), 36, d_pcm.data_ptr(), F, stream)

        def sync_all():
            torch.cuda.synchronize()
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize()

        # untimed warm-up (starts from reset, so it also absorbs the two silent start-up frames: every timed frame is a live frame)
        for _ in range(max(warmup, 1)):
            step()
        sync_all()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        sync_all()
        elapsed = time.perf_counter() - t0
        timed_pcm = d_pcm.clone()                            # output of the last timed step (checked below)
        if world > 1:
            t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if a.share_device else dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        # live kernel timing for the roofline: HIP events around the sample kernel on its own stream
        batch.enable_timing(True)
        ks = []
        for _ in range(3):
            step()
            torch.cuda.synchronize()
            ks.append(batch.last_timing())
        batch.enable_timing(False)
        ms_sample = float(np.median([k[0] for k in ks]))
        ms_frame = float(np.median([k[1] for k in ks]))
        nonzero = int(torch.count_nonzero(d_pcm).item())
        assert nonzero > 0.5 * d_pcm.numel(), "benchmark output is degenerate"
        # checker leg (rank 0): replay a few streams of the timed region on the plain-C oracle -- the same features fed
        # warmup + steps times in a row from reset -- and compare the last step's PCM bit for bit
        parity_checked = 0
        if rank == 0 and check_streams > 0:
            from oracle import orc
            passes = max(warmup, 1) + steps
            k = min(check_streams, n)
            pick = sorted({(i * n) // k + (i % 4 if n >= 4 * k else 0) for i in range(k)})
            want = orc.synthesize_many(blob, np.tile(feats[pick], (1, passes, 1)))[:, -F * 160:]
            got = timed_pcm[pick].cpu().numpy()
            if not np.array_equal(got, want):
                raise SystemExit(f"bench.py: timed output differs from the CPU oracle on streams {[p for p, g, w in zip(pick, got, want) if not np.array_equal(g, w)]}")
            parity_checked = len(pick)
        spw_used = batch.streams_per_workgroup
        batch.close()
        envelope = None
        if fast and rank == 0 and a.check_streams > 0:       # FAST: its own check on a few streams of this very batch (their feature files), teacher-forced
            kk = min(4, n)
            pick = sorted({(i * n) // kk for i in range(kk)})
            ek, eda, edb = fast_envelope_check(blob, np.ascontiguousarray(feats[pick]), int8, fp16_fc, spw_used, local)
            envelope = {"envelope_checked": ek, "max_gru_a_deviation": eda, "max_gru_b_deviation": edb,
                        "protocol": "teacher-forced against the PARITY engine, limits = the reference's AVX2-vs-generic-C envelope (tests/golden/simd_envelope_v1.json x 1.25)"}
        samples_per_step = n * F * 160
        return {"value": world * samples_per_step * steps / elapsed, "ms_per_step": elapsed / steps * 1e3, "ms_sample": ms_sample, "ms_frame": ms_frame,
                "parity_checked": parity_checked, "streams_per_workgroup": spw_used, "samples_per_step": samples_per_step, "blob": blob, "nb_a": int(api.check_model(blob)[1][1]),
                "envelope": envelope}

    n, F = a.streams, a.frames
    head = run_workload(n, F, a.int8, a.steps, a.warmup, fast=a.fast, fp16_fc=a.fp16_fc, spw=a.spw, densities=a.densities, check_streams=a.check_streams, skew=a.skew)
    value, ms_sample, ms_frame, parity_checked, samples_per_step, nb_a = (head[k] for k in ("value", "ms_sample", "ms_frame", "parity_checked", "samples_per_step", "nb_a"))

    def lds_frac(rec, int8):
        return rec["samples_per_step"] / (rec["ms_sample"] * 1e-3) * (LDS_OPERAND_BYTES_PER_SAMPLE_I8 if int8 else LDS_OPERAND_BYTES_PER_SAMPLE) / 1e9 / (PEAK_LDS_TBS * 1e3)

    # ---- (N = 1, the default command only) what else the metric and BASELINE's configs name: config 2's 1024 streams, config 4's int8 weights, the deadline
    also, rt = None, None
    default_cmd = world == 1 and not (a.fast or a.int8 or a.densities or a.spw or a.skew) and (n, F) == (STREAMS_PER_GPU, FRAMES_PER_STEP) and not a.no_extras
    if default_cmd:
        also = {}
        r1 = run_workload(CONFIG2_STREAMS, F, False, max(a.steps // 2, 2), 1, check_streams=a.check_streams, seed0=3000)
        also["config2_1024_streams"] = {"value": r1["value"], "unit": "samples/s", "ms_per_step": r1["ms_per_step"], "launch_ms": r1["ms_sample"], "frac": lds_frac(r1, False),
                                        "streams_per_workgroup": r1["streams_per_workgroup"], "parity_checked": r1["parity_checked"],
                                        "workload": f"BASELINE config 2: {CONFIG2_STREAMS} concurrent streams x {F} frames per step, fp32 weights, bit-exact arithmetic (four streams per CU)"}
        r2 = run_workload(n, F, True, max(a.steps // 2, 2), 1, check_streams=a.check_streams, seed0=5000)
        also["int8_parity"] = {"value": r2["value"], "unit": "samples/s", "ms_per_step": r2["ms_per_step"], "launch_ms": r2["ms_sample"], "frac": lds_frac(r2, True),
                               "operand_bytes_per_sample": LDS_OPERAND_BYTES_PER_SAMPLE_I8,
                               "streams_per_workgroup": r2["streams_per_workgroup"], "parity_checked": r2["parity_checked"],
                               "workload": f"BASELINE config 4's weights: {n} concurrent streams x {F} frames per step, int8 GRU-A / GRU-B (bit-exact against the reference's generic int8 build), fp32 dual FC"}
        probe = rt_measure(a, RT_PROBE_STREAMS, RT_PROBE_STEPS, 5, world, rank, local, dev, head["blob"], False)
        ok = [r for r in probe if r["meets_deadline_p999"]]
        best = max(ok, key=lambda r: r["streams"]) if ok else None
        rt = {"sustained_streams": best["streams"] if best else 0,
              "note": "frame-at-a-time synthesis (one 10-ms frame for every stream per step, enqueue + wait, src/lpcnet_demo.c:203-219); `sustained_streams` = the "
                      f"largest of the probed counts {RT_PROBE_STREAMS} whose p99.9 step over {RT_PROBE_STEPS} steps (in effect the slowest of them) stays under the 10-ms frame period (0: none); "
                      "`probe` carries host wall time and device time of the steps, every late step with both (tools: bench.py --rt --rt-sweep for other counts / longer runs)",
              "p50": best["step_ms_p50"] if best else None, "p99": best["step_ms_p99"] if best else None, "p999": best["step_ms_p999"] if best else None, "max": best["step_ms_max"] if best else None,
              "over_deadline_steps": best["over_deadline_steps"] if best else None, "device_ms_max": best["device_ms_max"] if best else None,
              "parity_checked": best["parity_checked"] if best else 0, "probe": probe}

    if world > 1:
        dist.barrier()                                       # the last collective: rank 0's CPU baseline below (~90 s) must not leave the other ranks inside one
    if rank == 0:
        x2 = head["streams_per_workgroup"] == 8
        launch_flop = samples_per_step * FLOP_PER_SAMPLE
        achieved_tflops = launch_flop / (ms_sample * 1e-3) / 1e12
        kernel_rate = samples_per_step / (ms_sample * 1e-3)
        op_bytes = LDS_OPERAND_BYTES_PER_SAMPLE_I8 if a.int8 else LDS_OPERAND_BYTES_PER_SAMPLE
        if nb_a != 1382:                                     # --densities: another GRU-A (SURVEY.md section 8d's formula: 32 weights + 1 index per block)
            op_bytes += (nb_a - 1382) * ((32 + 4) if a.int8 else (128 + 4))
        op_gbs = kernel_rate * op_bytes / 1e9
        traffic, traffic_src = None, None
        khash = kernel_source_hash()
        for rnd in ("r06", "r05", "r04", "r03", "r02", "r01"):      # PMC passes of this command, newest round first
            tag = ("_int8" if a.int8 else "") + ("_fast" if a.fast else "")
            tpath = os.path.join(ROOT, "profiles", f"{rnd}_hbm_traffic{tag}.json")
            if os.path.exists(tpath) and (n, F) == (STREAMS_PER_GPU, FRAMES_PER_STEP):
                try:
                    rec = json.load(open(tpath))
                    # only a measurement of THIS kernel counts: the file records the hash of the kernel sources it was taken with
                    if rec.get("kernel_source_hash") == khash:
                        traffic = rec.get("hbm_bytes_per_launch")
                        traffic_src = os.path.relpath(tpath, ROOT)
                    else:
                        traffic_src = f"stale: {os.path.relpath(tpath, ROOT)} was measured with other kernel sources"
                    break
                except Exception:
                    traffic = None
        out = {
            "metric": "16 kHz samples/sec & concurrent real-time streams, 1/2/4/8 MI355X",
            "value": value, "unit": "samples/s",
            "realtime_streams_by_division": value / 16000.0,
            "realtime_streams_sustained": rt["sustained_streams"] if rt else None,
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": head["ms_per_step"],
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "int8 weights/activations x f32 accumulate (GRU-A/GRU-B), f32 elsewhere" if a.int8 else "f32", "data": "synthetic",
            "parity_checked": parity_checked,
            **({"envelope_checked": head["envelope"]["envelope_checked"], "envelope": head["envelope"]} if head.get("envelope") else {}),
            "config": {"workload": f"{n} concurrent streams (each with its own feature file) per GPU x {F} frames ({F * 160} samples) per step, "
                                   + ("int8 GRU weights (v_dot4_i32_i8)" if a.int8 else "fp32 weights")
                                   + ", register-resident block-sparse GRU-A, "
                                   + (("FAST arithmetic (FMA / int32 accumulation, not bit-exact)" + (", fp16 dual FC" if a.fp16_fc else "")) if a.fast else "bit-exact (PARITY) arithmetic")
                                   + ("; eight streams per CU as two groups of four half a step apart (BASELINE config 2's 1024 streams fill half of these stream slots: "
                                      "`also.config2_1024_streams`)" if x2 else ""),
                       "arithmetic": "fast" if a.fast else "parity",
                       "gru_a_blocks": nb_a, "gru_a_densities": a.densities or "0.05,0.05,0.2 (benchmark model)",
                       "gru_a_skew": a.skew, "gru_a_items_per_lane": int(api.check_model(head["blob"])[1][3]),
                       "streams_per_gpu": n, "frames_per_step": F, "streams_per_workgroup": head["streams_per_workgroup"],
                       "sharding": f"{world} x {n} independent streams, no data-path collective"},
            "roofline": {"bound": "lds_operand_bandwidth", "kernel": "lpcn::sample_kernel_x2" if x2 else "lpcn::sample_kernel",
                         "achieved": op_gbs, "peak": PEAK_LDS_TBS * 1e3, "unit": "GB/s",
                         "frac": op_gbs / (PEAK_LDS_TBS * 1e3), "traffic": traffic, "traffic_source": traffic_src,
                         "launch_ms": ms_sample, "frame_kernels_ms": ms_frame,
                         "operand_bytes_per_sample": op_bytes,
                         "note": "algorithmic operand bytes (each weight/table entry once per stream-sample); the engine keeps "
                                 "GRU-A weights in VGPRs and shares every LDS read among the workgroup's streams, so realised LDS bytes are lower; "
                                 "`traffic` comes from the rocprofv3 PMC passes recorded under profiles/ (it cannot be collected from inside the "
                                 "process) and is null unless that file was measured with these very kernel sources",
                         "kernel_source_hash": khash, "library_build_info": api.build_info(),
                         "library_matches_sources": api.build_info().get("dev") == khash,
                         "valu_fp32": {"achieved_TFLOPs": achieved_tflops, "peak_TFLOPs": PEAK_FP32_TFLOPS,
                                       "frac": achieved_tflops / PEAK_FP32_TFLOPS, "flop_per_sample": FLOP_PER_SAMPLE,
                                       "measured_packed_mul_add_no_fma_TFLOPs": MEASURED_FP32_MUL_ADD_TFLOPS,
                                       "frac_of_measured_no_fma": achieved_tflops / MEASURED_FP32_MUL_ADD_TFLOPS},
                         "l2_gather": {"achieved_GBs": kernel_rate * L2_BYTES_PER_SAMPLE / 1e9, "bytes_per_sample": L2_BYTES_PER_SAMPLE,
                                       "note": "embedding rows + frame products, L2 -> CU; the tables stay L2-resident, this is not HBM traffic"},
                         "hbm": None if traffic is None else {"achieved_GBs": traffic / (ms_sample * 1e-3) / 1e9, "peak_GBs": PEAK_HBM_GBS,
                                                              "frac": traffic / (ms_sample * 1e-3) / 1e9 / PEAK_HBM_GBS,
                                                              "note": "measured PMC traffic per launch / live launch time: HBM is not a bound of this kernel"}},
        }
        if also is not None:
            out["also"] = also
        if rt is not None:
            out["rt"] = rt
        if a.share_device:
            out["rehearsal"] = "all ranks on device 0, gloo control plane: a plumbing check of the multi-rank path, not a scaling number"
        if not a.no_cpu_baseline:                            # (N > 1: rank 0 only, behind the timed region; the other ranks wait at the final barrier)
            try:
                out["cpu_baseline"] = cpu_baseline(a.int8)
            except Exception as e:  # the baseline must never take the GPU number down with it
                out["cpu_baseline"] = {"value": None, "unit": "samples/s", "cores": 0, "kind": "port", "sample": f"failed: {e}"}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
