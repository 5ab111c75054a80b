#!/usr/bin/env python3
"""What a 1 -> 8 GPU curve would expose and ONE GPU can still measure: host feeding and launch serialisation of the sharded
batch.  `lpcnet_batch_create_sharded` with 8 shards -- all on device 0 here -- of 896 streams each (the node-level real-time shape: 8 x 896 = 7168); every step, eight
host threads enqueue ONE 10-ms frame for their shard on the shard's own HIP stream (lpcnet_batch_synthesize_device_shard: three frame kernels, the argument block, the
sample kernel) and the step ends with one synchronisation.  Reported per shard thread: the host time inside the enqueue call (p50 / p99 / max, us), how much the eight
calls overlap in wall time (sum of the calls' durations / the span from the first call's start to the last call's end: 1 = serialised by a lock, up to 8 = concurrent), and
the step's wall time.  The eight shards time-share one GPU here, so the step time is NOT a scaling number; the enqueue figures are what carries over to eight devices.

    python tools/shard_rt.py [--shards 8] [--streams 896] [--steps 300]  ->  one JSON line
"""
import argparse
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shards", type=int, default=8)
    ap.add_argument("--streams", type=int, default=896, help="streams per shard")
    ap.add_argument("--steps", type=int, default=300)
    a = ap.parse_args()
    import torch
    from lpcnet_amd import api, synth
    dev = torch.device("cuda", 0)
    blob = synth.blob_bytes(synth.make_model())
    K, n = a.shards, a.streams
    b = api.LPCNetBatch(K * n, blob, devices=[0] * K)
    shards = b.shards
    assert len(shards) == K
    T = a.steps + 5
    pool = np.stack([synth.make_features(5000 + i, T) for i in range(32)])
    d_feat = [torch.from_numpy(pool).to(dev)[torch.arange(c, device=dev) % 32].permute(1, 0, 2).contiguous() for (_, c, _) in shards]      # per shard [T][count][36]
    d_pcm = [torch.zeros((c, 160), dtype=torch.int16, device=dev) for (_, c, _) in shards]
    t_in = np.zeros((T, K, 2))
    go = [threading.Event() for _ in range(K)]
    done = [threading.Event() for _ in range(K)]
    stop = False
    step_box = [0]

    def worker(k):
        while True:
            go[k].wait(); go[k].clear()
            if stop:
                return
            t = step_box[0]
            t0 = time.perf_counter()
            b.synthesize_device_shard(k, d_feat[k][t].data_ptr(), 36, d_pcm[k].data_ptr(), 1, 0)      # the shard's own stream
            t_in[t, k] = (t0, time.perf_counter())
            done[k].set()

    ths = [threading.Thread(target=worker, args=(k,), daemon=True) for k in range(K)]
    for th in ths:
        th.start()
    wall = np.zeros(T)
    for t in range(T):
        step_box[0] = t
        t0 = time.perf_counter()
        for k in range(K):
            go[k].set()
        for k in range(K):
            done[k].wait(); done[k].clear()
        b.sync()
        wall[t] = (time.perf_counter() - t0) * 1e3
    stop = True
    for k in range(K):
        go[k].set()
    w = slice(5, T)
    dur = (t_in[w, :, 1] - t_in[w, :, 0]) * 1e6                     # us inside the enqueue call, per (step, shard)
    span = (t_in[w, :, 1].max(axis=1) - t_in[w, :, 0].min(axis=1)) * 1e6
    overlap = dur.sum(axis=1) / span
    out = {"tool": "tools/shard_rt.py", "shards": K, "streams_per_shard": n, "steps": a.steps, "device": "all shards on device 0 (time-shared: step time is not a scaling number)",
           "enqueue_us_per_shard_thread": {"p50": float(np.percentile(dur, 50)), "p99": float(np.percentile(dur, 99)), "max": float(dur.max()),
                                           "per_shard_p50": [float(np.percentile(dur[:, k], 50)) for k in range(K)]},
           "enqueue_span_us_p50": float(np.percentile(span, 50)),
           "overlap_factor": {"p50": float(np.percentile(overlap, 50)), "min": float(overlap.min()),
                              "note": "sum of the eight calls' durations / wall span of the eight calls; 1 = serialised (a global lock on the step path), > 1 = the shard threads enqueue concurrently"},
           "step_ms": {"p50": float(np.percentile(wall[w], 50)), "p99": float(np.percentile(wall[w], 99)), "max": float(wall[w].max())},
           "streams_per_workgroup": b.streams_per_workgroup}
    print(json.dumps(out))
    b.close()


if __name__ == "__main__":
    main()
