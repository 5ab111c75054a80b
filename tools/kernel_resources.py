#!/usr/bin/env python3
"""Register / scratch accounting of the sample kernel's variants from the compiler's own output (no GPU needed).

    python tools/kernel_resources.py [--s 4] [--asm-dir DIR] [--json]

Compiles lpcnet_amd/csrc/sample_variants.hip for gfx950 to assembly (device only, same flags as lpcnet_amd/build.py),
then reports per kernel: VGPR / SGPR counts, spill counts, scratch bytes (the .amdhsa metadata), and how many scratch
accesses sit INSIDE the per-sample loop.  The loop is delimited in the assembly by the kernel's own marker comments
(`; LPCN_SAMPLE_LOOP_END` at the bottom of the loop body; the loop header is the target of the backward branch that
follows it).  tests/test_kernel_resources.py asserts on the result so a spill in the hot loop cannot come back unnoticed.
"""
import argparse
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def compile_asm(s_value, out_path):
    """s_value 1 / 2 / 4: sample_variants.hip with -DLPCN_S; 8: the two-group kernel (sample_x2.hip)"""
    from lpcnet_amd import build
    src = "sample_x2.hip" if s_value == 8 else "sample_variants.hip"
    cmd = [build.HIPCC] + build.HIP_FLAGS + ([] if s_value == 8 else [f"-DLPCN_S={s_value}"]) + ["--cuda-device-only", "-S", os.path.join(build.CSRC, src), "-o", out_path]
    subprocess.check_call(cmd, stderr=subprocess.DEVNULL)


def demangle(name):
    m = re.match(r"_ZN4lpcn13sample_kernelILi(\d+)ELi(\d+)ELb(\d)ELb(\d)ELb(\d)EEE", name)
    if m:
        return dict(S=int(m.group(1)), NW=int(m.group(2)), int8=bool(int(m.group(3))), fast=bool(int(m.group(4))), pack2=bool(int(m.group(5))))
    m = re.match(r"_ZN4lpcn16sample_kernel_x2ILi(\d+)EEE", name)          # eight streams per workgroup (two groups of four)
    return None if not m else dict(S=8, NW=int(m.group(1)), int8=False, fast=False, pack2=False)


def analyse(asm_path):
    text = open(asm_path).read()
    lines = text.split("\n")
    # ---- function bodies
    bodies = {}
    cur = None
    for i, ln in enumerate(lines):
        m = re.match(r"^(_ZN4lpcn1[36]sample_kernel\w+):\s*(;.*)?$", ln)
        if m:
            cur = m.group(1)
            bodies[cur] = [i, None]
        elif cur and ".end_amdhsa_kernel" in ln:
            bodies[cur][1] = i
            cur = None
    # ---- metadata (YAML at the end): one record per kernel
    meta = {}
    for rec in re.split(r"\n\s+- \.agpr_count:", text)[1:]:
        nm = re.search(r"\.name:\s+(\S+)", rec)
        if not nm:
            continue
        get = lambda key: int(re.search(r"\." + key + r":\s+(\d+)", rec).group(1))
        meta[nm.group(1)] = dict(vgpr=get("vgpr_count"), sgpr=get("sgpr_count"), vgpr_spill=get("vgpr_spill_count"),
                                 sgpr_spill=get("sgpr_spill_count"), scratch_bytes=get("private_segment_fixed_size"))
    out = []
    for name, (a, b) in bodies.items():
        info = demangle(name)
        if info is None or b is None:
            continue
        body = lines[a:b]
        labels = {m.group(1): k for k, ln in enumerate(body) for m in [re.match(r"^(\.LBB\d+_\d+):", ln)] if m}
        ends = [k for k, ln in enumerate(body) if "LPCN_SAMPLE_LOOP_END" in ln]
        in_loop = None
        loop_lines = None
        sgpr_reloads = None
        flat_in_loop = None
        if ends:
            end = ends[-1]
            # the loop's backward branch: first branch after the marker whose target label lies before the marker
            head = None
            for k in range(end, len(body)):
                m = re.search(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", body[k])
                if m and m.group(1) in labels and labels[m.group(1)] < end:
                    head = labels[m.group(1)]
                    break
            if head is not None:
                region = body[head:end]
                in_loop = sum(1 for ln in region if re.search(r"\bscratch_(load|store)", ln))
                loop_lines = len(region)
                # spilled SGPRs live in VGPR lanes: every reload in the loop is a v_readlane_b32 (plus its hazard wait states)
                sgpr_reloads = sum(1 for ln in region if re.search(r"\bv_readlane_b32", ln))
                # a FLAT access in the loop makes the compiler force every later LDS wait to lgkmcnt(0) (see sample_kernel.hip.h)
                flat_in_loop = sum(1 for ln in region if re.search(r"^\s+flat_(load|store|atomic)", ln))
        # DPP read-after-VALU-write hazard of the hand-written v_fmac_f32_dpp (the assembler's hazard recogniser cannot
        # see inside inline asm): its DPP source must not be written by a VALU instruction in the two preceding slots
        dpp_viol = 0
        insts = [ln.strip() for ln in body if ln.startswith("\t") and not ln.strip().startswith((".", ";"))]
        for k, ins in enumerate(insts):
            m = re.match(r"v_fmac_f32_dpp (v\d+), (v\d+),", ins)
            if not m:
                continue
            src = int(m.group(2)[1:])
            for prev in insts[max(0, k - 2):k]:             # VALU write -> DPP read of the same VGPR: 2 wait states
                pm = re.match(r"(v_\w+) (?:v(\d+)|v\[(\d+):(\d+)\])", prev)
                if pm and not prev.startswith("v_fmac_f32_dpp"):
                    lo, hi = (int(pm.group(2)),) * 2 if pm.group(2) else (int(pm.group(3)), int(pm.group(4)))
                    if lo <= src <= hi:                     # (a register RANGE that contains the DPP source counts too)
                        dpp_viol += 1
            for prev in insts[max(0, k - 5):k]:             # EXEC write -> DPP: 5 wait states
                if re.match(r"v_cmpx", prev) or re.match(r"s_\w+ exec(_lo|_hi)?,", prev) or re.match(r"s_\w+saveexec", prev):
                    dpp_viol += 1
        total = sum(1 for ln in body if re.search(r"\bscratch_(load|store)", ln))
        rec = dict(name=name, **info, **meta.get(name, {}), scratch_insts=total, scratch_insts_in_sample_loop=in_loop, sample_loop_asm_lines=loop_lines, sgpr_reloads_in_sample_loop=sgpr_reloads, flat_insts_in_sample_loop=flat_in_loop,
                   fmac_dpp=sum(1 for i in insts if i.startswith("v_fmac_f32_dpp")), dpp_hazard_violations=dpp_viol)
        out.append(rec)
    return sorted(out, key=lambda r: (r["S"], r["int8"], r["fast"], r["pack2"], r["NW"]))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--s", type=int, default=4, help="streams per workgroup: 1, 2, 4, or 8 = the two-group kernel")
    ap.add_argument("--asm-dir", default=None)
    ap.add_argument("--json", action="store_true")
    a = ap.parse_args()
    d = a.asm_dir or tempfile.mkdtemp(prefix="lpcn_asm_")
    path = os.path.join(d, f"sample_s{a.s}.s")
    if not os.path.exists(path):
        compile_asm(a.s, path)
    rows = analyse(path)
    if a.json:
        print(json.dumps(rows, indent=1))
        return
    print(f"# {path}")
    print("S NW int8 fast (+ = two workgroups per CU) | vgpr sgpr vgpr_spill sgpr_spill scratch_B | scratch insts: total / in sample loop (loop asm lines) | SGPR reloads (v_readlane), FLAT instructions in sample loop")
    for r in rows:
        print(f"{r['S']} {r['NW']:2d} {int(r['int8'])}    {int(r['fast'])}{'+' if r['pack2'] else ' '}   | {r.get('vgpr', -1):4d} {r.get('sgpr', -1):4d} {r.get('vgpr_spill', -1):6d} {r.get('sgpr_spill', -1):10d} "
              f"{r.get('scratch_bytes', -1):9d} | {r['scratch_insts']:3d} / {r['scratch_insts_in_sample_loop']} ({r['sample_loop_asm_lines']}) | {r['sgpr_reloads_in_sample_loop']}, {r['flat_insts_in_sample_loop']}"
              + (f" | v_fmac_f32_dpp {r['fmac_dpp']}, hazard violations {r['dpp_hazard_violations']}" if r['fmac_dpp'] else ""))


if __name__ == "__main__":
    main()
