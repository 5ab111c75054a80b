#!/usr/bin/env python3
"""Per-phase instruction census of the benchmark's PARITY float kernel (sample_kernel<4, 30, false, false>), derived from the
compiler's assembly (no GPU needed) -- VERDICT r3 item 3a: give the issued VALU work names.

    python tools/valu_census.py [--csv profiles/r04_valu_census.csv]

The sample loop is cut at its four workgroup barriers (P1 GRU-A rows | P2 gates | P3 GRU-B + candidate heads | P4 tree |
P5 leader).  P2, P4 and P5 are straight-line code: their static counts ARE the executed counts per wave (P5: wave 0 only).
P1 and P3 contain the unrolled item chains, of which a wave executes only its own items: an item is recognised as the code
between two consecutive state fetches (ds_read_b128 of the GRU-A state) that contains the four matrix-pipe products, its cost
is the median over the chain, and the executed count is  (everything that is not an item: start values, gather, slot
boundaries, close)  +  items of the wave (from the dealing, lpcnet_hip_model_layout)  x  cost of an item.  GRU-B's block
loop is the hand-written assembly (tools/gen_grub_asm.py): 96 blocks of its body per stream.
Units: wave-instructions per sample step and workgroup (4 streams); x 64 / 4 = lane-operations per stream-sample.
"""
import argparse
import ctypes as C
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def compile_asm(path, int8=False):
    from lpcnet_amd import build
    sel = ["-DLPCN_S=2", "-DLPCN_ONLY_BENCH_VARIANT=2"] if int8 else ["-DLPCN_S=4", "-DLPCN_ONLY_BENCH_VARIANT=1"]
    cmd = [build.HIPCC] + build.HIP_FLAGS + sel + ["--cuda-device-only", "-S", os.path.join(build.CSRC, "sample_variants.hip"), "-o", path]
    subprocess.check_call(cmd, stderr=subprocess.DEVNULL)


def classify(ln):
    t = ln.strip().split()[0] if ln.strip() else ""
    if t.startswith("v_mfma"):
        return "mfma"
    if t.startswith("v_"):
        return "valu"
    if t.startswith("ds_"):
        return "lds"
    if t.startswith(("global_", "buffer_", "scratch_", "flat_")):
        return "vmem"
    if t.startswith("s_"):
        return "salu"
    return None


def count(lines):
    c = dict(valu=0, mfma=0, lds=0, vmem=0, salu=0, pk=0, dpp=0)
    for ln in lines:
        k = classify(ln)
        if k:
            c[k] += 1
            if k == "valu" and ln.strip().startswith("v_pk_"):
                c["pk"] += 1
            if k == "valu" and ("quad_perm" in ln or "row_" in ln):
                c["dpp"] += 1
    return c


def items_of(lines):
    """split a region into item bodies (>= 4 v_mfma between two ds_read_b128) and the rest"""
    idx = [i for i, ln in enumerate(lines) if ln.strip().startswith("ds_read_b128")]
    item_cost, used = [], set()
    for a, b in zip(idx, idx[1:]):
        body = lines[a:b]
        if sum(1 for ln in body if ln.strip().startswith("v_mfma")) == 4 and len(body) < 60:
            item_cost.append(count(body))
            used.update(range(a, b))
    rest = [ln for i, ln in enumerate(lines) if i not in used]
    return item_cost, rest


def layout(int8=False):
    from lpcnet_amd import api, synth
    L = api.load_library()
    blob = synth.blob_bytes(synth.make_model(flavour="int8" if int8 else "float"))
    out = (C.c_int * 65)()
    L.lpcnet_hip_model_layout.argtypes = [C.c_char_p, C.c_int, C.POINTER(C.c_int)]
    assert L.lpcnet_hip_model_layout(blob, len(blob), out) == 0
    o = list(out)
    waves = []
    for w in range(8):
        bounds = o[1 + w * 7:1 + w * 7 + 4]
        head = o[57 + w]
        waves.append(dict(p1_items=bounds[3], head=head))
    return waves


def loop_regions(lines, head, end):
    """the sample loop cut at its workgroup barriers: [P1, P2, P3, P4, P5] as line lists"""
    bars = [i for i, ln in enumerate(lines[:end]) if ln.strip() == "s_barrier" and i > head]
    b1, b2, b3, b4 = bars[-4:]
    return dict(P1=lines[head:b1], P2=lines[b1:b2], P3=lines[b2:b3], P4=lines[b3:b4], P5=lines[b4:end])


def find_loop(lines):
    end = next(i for i, ln in enumerate(lines) if "LPCN_SAMPLE_LOOP_END" in ln)
    for ln in lines[end:end + 80]:                          # the first branch behind the marker that goes BACKWARD is the loop's back edge
        m = re.match(r"\s+s_c?branch\w*\s+(\.LBB\d+_\d+)", ln)
        if m:
            tgt = next((i for i, l2 in enumerate(lines) if l2.startswith(m.group(1) + ":")), None)
            if tgt is not None and tgt < end:
                return tgt, end
    raise SystemExit("sample loop not found")


def split_at_mfma(lines, opcode):
    """int8 items: the chain is cut at every matrix-pipe instruction; a segment = [mfma, next mfma) holds the tail of one item (wait states,
    2 conversions, packed add) and the front of the next (offset unpack, state fetch, exit test).  Returns (segments' counts, lines outside)."""
    idx = [i for i, ln in enumerate(lines) if ln.strip().startswith(opcode)]
    segs = [count(lines[a:b]) for a, b in zip(idx, idx[1:])]
    return idx, segs


def inner_loops(lines):
    """(start, end) line ranges of the loops INSIDE a region: a label followed later by a conditional branch back to it"""
    labels = {m.group(1): i for i, ln in enumerate(lines) for m in [re.match(r"^(\.LBB\d+_\d+):", ln)] if m}
    out = []
    for i, ln in enumerate(lines):
        m = re.match(r"\s+s_cbranch\w*\s+(\.LBB\d+_\d+)", ln)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            out.append((labels[m.group(1)], i + 1))
    return out


def main_int8(a):
    """sample_kernel<2, 32, true, false, true>: int8 blobs, PARITY, two workgroups per CU."""
    path = a.asm
    if not path:
        path = os.path.join(tempfile.mkdtemp(prefix="lpcn_census_"), "k.s")
        compile_asm(path, int8=True)
    text = open(path).read()
    m = re.search(r"^(_ZN4lpcn13sample_kernelILi2ELi32ELb1ELb0ELb1EEE\w+):", text, re.M)
    lines = text[m.start():].split("\n") if m else text.split("\n")
    lines = lines[:next(i for i, ln in enumerate(lines) if ".end_amdhsa_kernel" in ln)]
    head, end = find_loop(lines)
    R = loop_regions(lines, head, end)
    waves = layout(int8=True)
    S = 2
    rows, tot = [], dict(valu=0, mfma=0, lds=0)

    def add(phase, what, per, units, note):
        rows.append((phase, what, per["valu"], per["mfma"], per["lds"], per["salu"], units, round(per["valu"] * units), round(per["mfma"] * units), note))
        tot["valu"] += per["valu"] * units; tot["mfma"] += per["mfma"] * units; tot["lds"] += per["lds"] * units

    med = lambda xs, k: sorted(x[k] for x in xs)[len(xs) // 2]
    zero = dict(valu=0, mfma=0, lds=0, vmem=0, salu=0, pk=0, dpp=0)
    sub = lambda c, d, n: {k: c[k] - d[k] * n for k in c}
    # ---- P1
    idx1, seg1 = split_at_mfma(R["P1"], "v_mfma_i32")
    item = {k: med(seg1, k) for k in seg1[0]}
    item["mfma"] = 1
    n1 = sum(w["p1_items"] for w in waves)
    add("P1", "GRU-A item: 1 v_mfma_i32_4x4x4i8 (the item's block products of both streams), 2 conversions, 1 packed add, offset unpack (2), state fetch, exit test",
        item, n1, f"{n1} wave-items per step outside the heads ({'; '.join(str(w['p1_items']) for w in waves)} per wave)")
    c1 = count(R["P1"])
    rest1 = sub(c1, item, len(idx1))
    half = {k: max(v, 0) / 2 for k, v in rest1.items()}
    add("P1", "start values (3 slots x 2 streams: bias + diag*h, x 128*127), gather issue + 9 gather adds per stream and slot, slot boundaries, close (static count / 2: a wave runs one of the two start paths)",
        half, 8, f"static: {rest1['valu']} VALU, {rest1['lds']} LDS, {rest1['vmem']} VMEM outside the {len(idx1)} item bodies")
    # ---- P2
    c2 = count(R["P2"])
    add("P2", "gate stage: 768 (neuron, stream) items on 512 lanes = 2 rounds (the second half empty), 2 sigmoid + 1 tanh (table) each, blend, re-quantisation floor(.5 + 127 x) in double, byte stores", c2, 8, "straight-line")
    # ---- P3: GRU-B's quad loop on the two chain waves (2, 3 since round 5), heads on the six others, the rest
    loops = [(s0, e0) for s0, e0 in inner_loops(R["P3"]) if any("v_dot4" in ln for ln in R["P3"][s0:e0])]
    gb = zero
    gb_lines = set()
    if loops:
        s0, e0 = max(loops, key=lambda t: t[1] - t[0])
        body = count(R["P3"][s0:e0])
        nq = sum(1 for ln in R["P3"][s0:e0] if ln.strip().startswith("v_dot4")) // 4
        gb = {k: v / nq for k, v in body.items()}
        gb_lines = set(range(s0, e0))
        add("P3", "GRU-B quad of 4 input blocks: 4 v_dot4_i32_i8 + 4 conversions + 4 dependent adds + 2 LDS reads (compiler's loop, two quads per trip)", gb, 24 * S,
            f"24 quads x {S} streams; loop body of {nq} quads: {body['valu']} VALU")
    p3 = [ln for i, ln in enumerate(R["P3"]) if i not in gb_lines]
    idx3, seg3 = split_at_mfma(p3, "v_mfma_i32")
    n_head = sum(w["head"] for w in waves)
    it3 = {k: med(seg3, k) for k in seg3[0]} if seg3 else item
    it3["mfma"] = 1
    add("P3", "candidate HEAD item (same body as P1's)", it3, n_head, f"{n_head} wave-items per step ({'; '.join(str(w['head']) for w in waves)} per wave)")
    rest3 = sub(count(p3), it3, len(idx3))
    add("P3", "recurrent part + GRU-B gates + re-quantisation (chain waves 2, 3), head start values (the six other waves), dual-FC row prefetch (all) (static count / 2: gate and head waves run different halves)",
        {k: max(v, 0) / 2 for k, v in rest3.items()}, 8, f"static: {rest3['valu']} VALU outside the {len(idx3)} head item bodies and the quad loop")
    add("P4", "tree: 255 nodes x 2 channels x 2 streams speculatively (16 mul + 16 add + table tanh per stream and lane), ballots", count(R["P4"]), 8, "straight-line")
    add("P5", "leader (wave 0): tree walk, mu-law, LPC chain, publish; thresholds (wave 1)", count(R["P5"]), 1, "straight-line, one wave")
    hdr = "phase,what,VALU_per_unit,MFMA_per_unit,LDS_per_unit,SALU_per_unit,units_per_step,VALU_wave_insts_per_step,MFMA_wave_insts_per_step,note"
    out = [hdr] + [",".join(str(x).replace(",", ";") for x in r) for r in rows]
    lane_ops = tot["valu"] * 64 / S
    gemv = (n1 + n_head) * (item["valu"] - 2) + 24 * S * gb.get("valu", 0)
    out.append(f"TOTAL,,,,,,,{round(tot['valu'])},{round(tot['mfma'])},\"= {lane_ops / 1000:.0f} k VALU lane-operations per stream-sample ({S} streams per workgroup; + {tot['mfma'] * 64 / S / 1000:.0f} k lanes x matrix-pipe instructions); "
               f"of which the two mat-vecs (items without their address arithmetic + GRU-B quads): {gemv * 64 / S / 1000:.0f} k; measured SQ_INSTS_VALU x 64 / samples = 192 k (profiles/r04_sq_counters_int8.csv)\"")
    txt = "\n".join(out)
    print(txt)
    if a.csv:
        with open(a.csv, "w") as f:
            f.write("# tools/valu_census.py --int8: per-phase instruction census of sample_kernel<2,32,true,false,true> (int8 blobs, PARITY, two workgroups per CU) from the compiler's assembly; wave-instructions per sample step and workgroup (2 streams)\n")
            f.write(txt + "\n")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--csv", default=None)
    ap.add_argument("--asm", default=None)
    ap.add_argument("--int8", action="store_true", help="the int8 PARITY kernel of BASELINE config 4 (sample_kernel<2,32,true,false,true>) instead of the float one")
    a = ap.parse_args()
    if a.int8:
        return main_int8(a)
    path = a.asm
    if not path:
        path = os.path.join(tempfile.mkdtemp(prefix="lpcn_census_"), "k.s")
        compile_asm(path)
    lines = open(path).read().split("\n")
    end = next(i for i, ln in enumerate(lines) if "LPCN_SAMPLE_LOOP_END" in ln)
    bars = [i for i, ln in enumerate(lines[:end]) if ln.strip() == "s_barrier"]
    b1, b2, b3, b4 = bars[-4:]
    # loop header: the target of the backward branch behind the end marker
    head = None
    for ln in lines[end:end + 60]:                          # the first branch behind the marker that goes BACKWARD is the loop's back edge
        m = re.match(r"\s+s_c?branch\w*\s+(\.LBB\d+_\d+)", ln)
        if m:
            tgt = next((i for i, l2 in enumerate(lines) if l2.startswith(m.group(1) + ":")), None)
            if tgt is not None and tgt < end:
                head = tgt
                break
    assert head is not None
    regions = dict(P1=lines[head:b1], P2=lines[b1:b2], P3=lines[b2:b3], P4=lines[b3:b4], P5=lines[b4:end])
    waves = layout()
    rows = []
    med = lambda xs, k: sorted(x[k] for x in xs)[len(xs) // 2]
    tot = dict(valu=0, mfma=0, lds=0)

    def add(phase, what, per_wave, nwaves, note):
        rows.append((phase, what, per_wave["valu"], per_wave["mfma"], per_wave["lds"], per_wave["salu"], nwaves,
                     per_wave["valu"] * nwaves, per_wave["mfma"] * nwaves, note))
        tot["valu"] += per_wave["valu"] * nwaves; tot["mfma"] += per_wave["mfma"] * nwaves; tot["lds"] += per_wave["lds"] * nwaves

    # ---- P1: items + the rest
    it1, rest1 = items_of(regions["P1"])
    item = {k: med(it1, k) for k in it1[0]} if it1 else count([])
    n_items1 = sum(w["p1_items"] for w in waves)
    add("P1", "GRU-A item (4 exact products on the matrix pipe + 8 packed adds + state fetch + control)", item, n_items1,
        f"{n_items1} wave-items per step outside the heads ({', '.join(str(w['p1_items']) for w in waves)} per wave)")
    r1 = count(rest1)
    # the rest of P1 is executed once per wave, but it holds both start-value paths (waves that wait for the indices / run-ahead waves) and
    # three slots' worth of code of which a wave runs its own: half of it is the working estimate
    half = {k: v // 2 for k, v in r1.items()}
    add("P1", "start values, gather, slot boundaries, close (static count / 2: a wave runs one of the two start paths)", half, 8,
        f"static: {r1['valu']} VALU, {r1['lds']} LDS, {r1['vmem']} VMEM in the region outside the item bodies")
    add("P2", "gate stage: 3 (neuron, stream) items per lane, 2 sigmoid + 1 tanh (table) each, blend, state stores", count(regions["P2"]), 8, "straight-line")
    # ---- P3: GRU-B block loop (assembly), recurrent part + gates, heads
    it3, rest3 = items_of(regions["P3"])
    n_head = sum(w["head"] for w in waves)
    add("P3", "candidate HEAD item (same body as P1's)", {k: med(it3, k) for k in it3[0]} if it3 else item, n_head,
        f"{n_head} wave-items per step ({', '.join(str(w['head']) for w in waves)} per wave)")
    blk = dict(valu=6, mfma=0, lds=2, vmem=0, salu=0, pk=2, dpp=0)          # grub_lds_loop: 4 v_add_f32 + 2 v_pk_mul_f32 + 2 ds_read_b128 (+ 1 wait) per block
    add("P3", "GRU-B block (hand-scheduled: 2 packed multiplies + 4 dependent adds + 2 LDS reads)", blk, 96 * 4, "96 blocks x 4 streams")
    r3 = count([ln for ln in rest3 if "v[216:" not in ln and "v[22" not in ln and "v[23" not in ln and "v[24" not in ln and "v[25" not in ln])
    quarter = {k: v // 2 for k, v in r3.items()}
    add("P3", "recurrent part, GRU-B gates, head start values, dual-FC prefetch (static count / 2: gate waves and head waves run different halves)", quarter, 8,
        f"static: {r3['valu']} VALU outside the item bodies and the block loop")
    add("P4", "tree: 255 nodes x 2 channels x 4 streams speculatively (16 mul + 16 add + tanh per stream and lane), ballots", count(regions["P4"]), 8, "straight-line")
    add("P5", "leader (wave 0): tree walk, mu-law, LPC chain, publish; thresholds (wave 1)", count(regions["P5"]), 1, "straight-line, one wave")
    hdr = "phase,what,VALU_per_unit,MFMA_per_unit,LDS_per_unit,SALU_per_unit,units_per_step,VALU_wave_insts_per_step,MFMA_wave_insts_per_step,note"
    out = [hdr] + [",".join(str(x).replace(",", ";") for x in r) for r in rows]
    lane_ops = tot["valu"] * 64 / 4
    out.append(f"TOTAL,,,,,,,{tot['valu']},{tot['mfma']},\"= {lane_ops / 1000:.0f} k VALU lane-operations per stream-sample (+ {tot['mfma'] * 64 / 4 / 1000:.0f} k lanes x matrix-pipe instructions); "
               f"PARITY minimum 129.7 k (one multiply + one add per MAC)\"")
    txt = "\n".join(out)
    print(txt)
    if a.csv:
        with open(a.csv, "w") as f:
            f.write("# tools/valu_census.py: per-phase instruction census of sample_kernel<4,30,false,false> from the compiler's assembly; wave-instructions per sample step and workgroup\n")
            f.write(txt + "\n")


if __name__ == "__main__":
    main()
