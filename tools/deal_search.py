#!/usr/bin/env python3
"""Measured local search over the dealing of GRU-A's slots to waves (run on a GPU box; a tool, not part of the product).

    python tools/deal_search.py [--rounds 3] [--int8]

model_pack.c deals the 18 slots of 64 rows to the 8 waves with a cost model fitted to phase clocks.  This tool asks the GPU
instead: starting from the model's dealing (LPCN_DEAL_PRINT=1), it moves one update / reset slot to another wave or swaps two of
them (LPCN_DEAL_FORCE), times the sample kernel of a fresh 1024-stream batch for each neighbour (in process; results are bit-exact
by construction: the dealing never changes arithmetic) and keeps what is faster.  Output: the best map found and its gain -- evidence for (or against) the cost
model, see DESIGN.md section 6.
"""
import argparse
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


_CTX = {}


X2 = False          # --x2: the two-group kernel (eight streams per workgroup, 2048 streams, LPCN_DEAL_FORCE_X2)


def bench(force, int8):
    """kernel rate (M samples/s) of a fresh 1024-stream batch dealt with `force` (None = the cost model's dealing), in process"""
    if X2:
        return bench_x2(force)
    import numpy as np
    sys.path.insert(0, ROOT)
    from lpcnet_amd import api, synth
    if "blob" not in _CTX:
        _CTX["blob"] = synth.blob_bytes(synth.make_model(flavour="int8" if int8 else "float"))
        base = np.stack([synth.make_features(1000 + s, 12) for s in range(8)])
        _CTX["feats"] = np.ascontiguousarray(base[np.arange(1024) % 8])
    if force is None:
        os.environ.pop("LPCN_DEAL_FORCE", None)
    else:
        os.environ["LPCN_DEAL_FORCE"] = ",".join(str(w) for w in force)
    try:
        b = api.LPCNetBatch(1024, _CTX["blob"])
    except Exception:
        return None
    b.streams_per_workgroup = 2 if int8 else 4
    b.enable_timing(True)
    b.synthesize(_CTX["feats"])
    best = None
    for _ in range(4):
        b.reset()
        b.synthesize(_CTX["feats"])
        t = b.last_timing()[0]
        best = t if best is None else min(best, t)
    b.close()
    return 1024 * 10 * 160 / (best * 1e-3) / 1e6


def bench_x2(force):
    import numpy as np
    sys.path.insert(0, ROOT)
    from lpcnet_amd import api, synth
    if "blob" not in _CTX:
        _CTX["blob"] = synth.blob_bytes(synth.make_model())
        base = np.stack([synth.make_features(1000 + s, 8) for s in range(16)])
        _CTX["feats"] = np.ascontiguousarray(base[np.arange(2048) % 16])
    if force is None:
        os.environ.pop("LPCN_DEAL_FORCE_X2", None)
    else:
        os.environ["LPCN_DEAL_FORCE_X2"] = ",".join(str(w) for w in force)
    try:
        b = api.LPCNetBatch(2048, _CTX["blob"])
        b.streams_per_workgroup = 8
    except Exception:
        return None
    b.enable_timing(True)
    b.synthesize(_CTX["feats"])
    best = None
    for _ in range(3):
        b.synthesize(_CTX["feats"])
        t = b.last_timing()[0]
        best = t if best is None else min(best, t)
    b.close()
    return 2048 * 8 * 160 / (best * 1e-3) / 1e6


def current_map(int8):
    env = dict(os.environ, LPCN_DEAL_PRINT="1")
    code = ("import sys; sys.path.insert(0, %r); from lpcnet_amd import api, synth; "
            "api.check_model(synth.blob_bytes(synth.make_model(flavour=%r)))" % (ROOT, "int8" if int8 else "float"))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=300)
    m = re.search(r"LPCN_DEAL slots \(length:wave, two-group kernel\):(.*)" if X2 else r"LPCN_DEAL slots \(length:wave[^)]*\):(.*)", r.stderr)
    slots = m.group(1).split()
    return [(s.startswith("c"), int(s.lstrip("c").split(":")[0]), int(s.split(":")[1])) for s in slots]


def legal(slots, wv):
    for w in range(8):
        idx = [i for i in range(len(slots)) if wv[i] == w]
        if len(idx) > 3 or sum(1 for i in idx if slots[i][0]) > 1 or sum(slots[i][1] for i in idx) > 30:
            return False
    return True


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--int8", action="store_true")
    ap.add_argument("--x2", action="store_true", help="the two-group kernel's dealing (LPCN_DEAL_FORCE_X2, 2048 streams, eight per workgroup); candidate slots move too")
    ap.add_argument("--start", default="", help="comma-separated start map instead of the model's dealing")
    a = ap.parse_args()
    global X2
    X2 = a.x2
    slots = current_map(a.int8)
    wv = [s[2] for s in slots]
    base = max(bench(None, a.int8), bench(None, a.int8))
    if a.start:
        wv = [int(x) for x in a.start.split(",")]
        print("start map %s -> %.2f M (the model's dealing: %.2f M)" % (a.start, bench(wv, a.int8), base), flush=True)
        base = max(bench(wv, a.int8), bench(wv, a.int8))
    print("model's dealing:", " ".join(("c" if c else "") + f"{n}:{w}" for c, n, w in slots), "-> %.2f M" % base, flush=True)
    best, best_v = list(wv), base
    zr = [i for i, s in enumerate(slots) if not s[0] or X2]
    for rnd in range(a.rounds):
        improved = False
        cands = []
        for i in zr:                                        # move one update / reset slot
            for w in range(8):
                if w != best[i]:
                    t = list(best); t[i] = w
                    cands.append(t)
        for x in range(len(zr)):                            # swap two of different length
            for y in range(x + 1, len(zr)):
                i, j = zr[x], zr[y]
                if best[i] != best[j] and slots[i][1] != slots[j][1]:
                    t = list(best); t[i], t[j] = t[j], t[i]
                    cands.append(t)
        seen = set()
        for t in cands:
            if tuple(t) in seen or not legal(slots, t):
                continue
            seen.add(tuple(t))
            v = bench(t, a.int8)
            if v is None:
                continue
            if v > best_v * 1.004:                          # beyond run-to-run noise (~0.3 %)
                v2 = bench(t, a.int8)                       # confirm
                if v2 and min(v, v2) > best_v * 1.003:
                    best, best_v, improved = t, min(v, v2), True
                    print("round %d: %s -> %.2f M" % (rnd, ",".join(map(str, best)), best_v), flush=True)
                    break                                   # restart the neighbourhood from the new point
        if not improved:
            break
    print("best:", ",".join(map(str, best)), "%.2f M (model's dealing %.2f M, %+.1f %%)" % (best_v, base, 100 * (best_v / base - 1)))


if __name__ == "__main__":
    main()
