#!/usr/bin/env python3
"""On-chip (SQ / LDS) counters of the sample kernel, collected on a GPU box.

    python tools/profile_sq.py [--int8] [--tag r02] [--extra "<bench.py flags>"]

rocprofv3 --pmc passes (at most 8 SQ counters each, --kernel-trace only: never mixed with other trace
domains) around `python bench.py --steps 2 --warmup 1 --no-cpu-baseline`, one pass per counter group; the
counters actually present on the box are taken from `rocprofv3 -L`.  Output: gpurun_out/sq/<tag>_sq_<flavour>.csv
(one row per counter: sum over the sample kernel's dispatches / number of dispatches) which
tools/profile_summarize.py copies into profiles/.
"""
import argparse
import collections
import csv
import glob
import re
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WISH = [
    # pass 1: time and instruction mix
    ["SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_SMEM"],
    # pass 2: where the wave cycles go (quad-cycles; WAIT_ANY + WAIT_INST_ANY + ACTIVE_INST_ANY ~ WAVE_CYCLES)
    ["SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_SCA", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS",
     "SQ_ACTIVE_INST_VMEM"],
    # pass 3: LDS array
    ["SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_LDS_ADDR_CONFLICT", "SQ_LDS_UNALIGNED_STALL", "SQ_LDS_MEM_VIOLATIONS", "SQ_INSTS_VALU_MFMA_MOPS_F32",
     "SQ_INST_LEVEL_LDS", "SQ_INST_LEVEL_VMEM"],
    # pass 4: busy
    ["SQ_VALU_MFMA_BUSY_CYCLES", "SQ_THREAD_CYCLES_VALU", "SQ_IFETCH", "SQ_INSTS_BRANCH", "SQ_INSTS_SENDMSG", "SQ_INST_CYCLES_SALU", "SQ_INSTS_VSKIPPED",
     "SQ_INSTS_WAVE32_LDS"],
    ["GRBM_GUI_ACTIVE", "GRBM_COUNT"],
]


def available():
    try:
        txt = subprocess.run(["rocprofv3", "-L"], capture_output=True, text=True, timeout=120).stdout
    except Exception as e:
        print("rocprofv3 -L failed:", e)
        return None
    return txt, txt


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--int8", action="store_true")
    ap.add_argument("--tag", default="r02")
    ap.add_argument("--extra", default="")
    a = ap.parse_args()
    fl = ("i8" if a.int8 else "f32") + ("_fast" if "--fast" in a.extra else "")
    out_dir = os.path.join(ROOT, "gpurun_out", "sq")
    os.makedirs(out_dir, exist_ok=True)
    os.environ["TMPDIR"] = "/tmp"
    os.environ["LPCNET_HIP_NO_AUTOTUNE"] = "1"            # only the workload's launches: the auto-tune's short trial launches of every variant would be averaged in
    av = available()
    have = None
    if av:
        have, txt = av
        open(os.path.join(out_dir, "counters_available.txt"), "w").write(txt)
    rows = collections.OrderedDict()
    ndisp = 0
    bench = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-extras"] + (["--int8"] if a.int8 else []) + a.extra.split()
    for gi, group in enumerate(WISH):
        ctrs = [c for c in group if have is None or len(have) < 1000 or re.search(r"\b" + c + r"\b", have)]
        if not ctrs:
            continue
        d = os.path.join(out_dir, f"pass{gi}_{fl}")
        subprocess.run(["rm", "-rf", d])
        cmd = ["rocprofv3", "--pmc"] + ctrs + ["--kernel-trace", "--output-format", "csv", "-d", d, "--"] + bench
        print("+", " ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd="/tmp")
        open(d + ".log", "w").write(r.stdout[-4000:] + "\n---\n" + r.stderr[-4000:])
        files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
        if not files:
            print("  no counter file (rc %d)" % r.returncode)
            continue
        acc = collections.defaultdict(float)
        cnt = collections.defaultdict(int)
        for rec in csv.DictReader(open(files[0])):
            if "sample_kernel" in rec["Kernel_Name"]:
                acc[rec["Counter_Name"]] += float(rec["Counter_Value"])
                cnt[rec["Counter_Name"]] += 1
        for c in ctrs:
            if cnt[c]:
                rows[c] = acc[c] / cnt[c]
                ndisp = cnt[c]
    path = os.path.join(out_dir, f"{a.tag}_sq_{fl}.csv")
    with open(path, "w") as o:
        o.write(f"# rocprofv3 --pmc <<=8 SQ counters per pass> --kernel-trace -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras{' --int8' if a.int8 else ''} {a.extra}\n")
        o.write(f"# lpcn::sample_kernel*, per-dispatch average over {ndisp} dispatches (bench.py's workload: 2048 streams x 25 frames x 160 samples per dispatch unless --streams says otherwise); SQ_*_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles\n")
        o.write("counter,value_per_dispatch\n")
        for k, v in rows.items():
            o.write(f"{k},{v:.1f}\n")
    print(open(path).read())


if __name__ == "__main__":
    main()
