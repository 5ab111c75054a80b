"""Register / scratch budget of the benchmarked sample-kernel variants, from the compiler's own assembly (no GPU).
VERDICT r1: `sample_kernel<4,30,false>` reports 15 spilled VGPRs; the spills are loop-invariant values of the FRAME loop
(spilled once before it, reloaded per frame) -- not one scratch access may sit inside the per-sample loop, and the int8
variants must not spill at all.  Also checks the DPP hazard of the hand-written v_fmac_f32_dpp of the FAST flavour."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


@pytest.fixture(scope="module")
def analysed(tmp_path_factory):
    """every streams-per-workgroup value compiled to assembly ONCE (device code only, the three in parallel: ~2 minutes) and analysed"""
    import kernel_resources as kr
    from concurrent.futures import ThreadPoolExecutor
    d = tmp_path_factory.mktemp("asm")
    paths = {sv: str(d / f"sample_s{sv}.s") for sv in (1, 2, 4, 8)}      # (8 = the two-group kernel, sample_x2.hip)
    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(lambda sv: kr.compile_asm(sv, paths[sv]), (1, 2, 4, 8)))
    return {sv: kr.analyse(paths[sv]) for sv in (1, 2, 4, 8)}


@pytest.fixture(scope="module")
def rows(analysed):
    return {(r["NW"], r["int8"], r["fast"]): r for r in analysed[4] if not r["pack2"]}


def test_no_scratch_access_inside_the_sample_loop(rows):
    assert len(rows) == 28                                  # float 24..96 (10 variants) and int8 32..96 (4), PARITY and FAST
    # round 5: EVERY instantiated variant, float ones included (more than 32 items per lane: the items past the 28th are streamed from L2
    # instead of being held -- round 4's 40-item kernel had 94 scratch accesses in the loop)
    for key in sorted(rows):
        r = rows[key]
        assert r["sample_loop_asm_lines"] and r["sample_loop_asm_lines"] > 3000, key      # the loop was found
        assert r["scratch_insts_in_sample_loop"] == 0, (key, r)
        assert r["vgpr"] <= 256
    for nw in (32, 48, 64):                                 # int8 weights are one VGPR per item: no spills at all
        assert rows[(nw, True, False)]["vgpr_spill"] == 0 and rows[(nw, True, False)]["scratch_bytes"] == 0
    # the benchmarked fp32 variant: its spills (frame-loop invariants) stay bounded
    assert rows[(30, False, False)]["vgpr_spill"] <= 40


def test_two_workgroups_per_cu_variants_stay_out_of_scratch_in_the_loop(analysed):
    """the 128-VGPR int8 variants (S <= 2, 32 items per lane): spills allowed outside, none inside the sample loop"""
    packed = [r for r in analysed[2] if r["pack2"]]
    assert len(packed) == 2                                 # PARITY and FAST
    for r in packed:
        assert r["int8"] and r["NW"] == 32 and r["vgpr"] <= 128 and r["scratch_insts_in_sample_loop"] == 0, r


def test_two_group_kernel_keeps_its_half_step_loop_out_of_scratch(analysed):
    """round 6, eight float streams per workgroup (sample_kernel_x2.hip.h): every items-per-lane variant; no scratch access inside the half-step loop"""
    r8 = {r["NW"]: r for r in analysed[8]}
    assert sorted(r8) == [24, 28, 30, 32]
    for nw, r in r8.items():
        assert r["S"] == 8 and r["sample_loop_asm_lines"] > 3000, nw       # the loop was found
        assert r["scratch_insts_in_sample_loop"] == 0 and r["flat_insts_in_sample_loop"] == 0 and r["vgpr"] <= 256, (nw, r)
        assert r["sgpr_reloads_in_sample_loop"] <= 48, (nw, r)
    assert r8[30]["vgpr_spill"] <= 8 and r8[30]["scratch_bytes"] <= 32      # (the tree on the matrix pipe: six values of the launch prologue live in scratch, none touched inside the loop)


def test_no_experiment_switches_are_left_in_the_kernels():
    """~15 compile-time experiment switches lived in the hot kernel, built and tested by nobody.  The measured alternatives are in the history
    and in EXPERIMENTS.md; what may still be overridden from the command line is the profiling build (LPCN_ENABLE_PROF / LPCN_PROF_MASK, built by
    `python -m lpcnet_amd.build --prof` and used by tools/) and the quick-listing switch of sample_variants.hip."""
    import re
    csrc = os.path.join(ROOT, "lpcnet_amd", "csrc")
    allowed = {"LPCN_ENABLE_PROF", "LPCN_PROF_MASK", "LPCN_S", "LPCN_ONLY_BENCH_VARIANT", "LPCN_HD", "LPCN_MAX_MODELS", "LPCN_MAX_RESIDENT", "LPCN_SOURCE_HASH", "LPCN_DEVICE_SOURCE_HASH", "LPCN_EXP10_TABLE_QUAL"}
    found = set()
    for f in sorted(os.listdir(csrc)):
        if f.endswith((".h", ".hip", ".c")) and not f.endswith("_gen.h"):
            found |= set(re.findall(r"#\s*ifn?def\s+(LPCN_\w+)", open(os.path.join(csrc, f)).read()))
    assert found <= allowed, sorted(found - allowed)


def test_fast_fmac_dpp_hazards(rows, analysed):
    """the hand-written v_fmac_f32_dpp of FAST float GRU-B sits inside inline assembly, where LLVM's hazard recogniser does not
    look: VALU-write -> DPP-read (2 slots, register ranges included) and EXEC-write -> DPP (5 slots) are checked on the
    assembly of every streams-per-workgroup value (S = 1 and 2 were unchecked)."""
    for key, r in rows.items():
        if key[2] and not key[1]:
            assert r["dpp_hazard_violations"] == 0, key     # (S >= 2: GRU-B and GRU-A both run on the matrix pipe, the DPP loop is gone)
    for sv in (1, 2):
        fast_float = [r for r in analysed[sv] if r["fast"] and not r["int8"]]
        assert len(fast_float) == 10
        for r in fast_float:
            assert (r["fmac_dpp"] >= 16 or sv > 1) and r["dpp_hazard_violations"] == 0, (sv, r["NW"], r["dpp_hazard_violations"])


def test_nothing_in_the_sample_loop_forces_full_lds_waits(rows):
    """round 3: a FLAT store in the loop (the trace, through a generic pointer) made the compiler wait with lgkmcnt(0) in every
    item of the GRU-A chains; and a register allocation that reloads ~110 spilled SGPRs per sample (v_readlane) instead of
    ~15 costs the benchmarked float kernel 2.3 % -- it has come and gone with one-line changes, so it is pinned here"""
    for key, r in rows.items():
        assert r["flat_insts_in_sample_loop"] == 0, key
    assert rows[(30, False, False)]["sgpr_reloads_in_sample_loop"] <= 40, rows[(30, False, False)]
    for key in [(32, True, False), (32, True, True), (30, False, True)]:
        assert rows[key]["sgpr_reloads_in_sample_loop"] <= 10, (key, rows[key])


def test_generated_assembly_loops_are_what_the_generator_emits():
    """GRU-B's hand-scheduled loops are generated (tools/gen_grub_asm.py) and committed; the single-stream pair -- the chain
    wave's first blocks, then the products its helper waves leave in LDS -- must be generated for the split the kernel is
    compiled with (LPCN_PROD_BLOCKS), or blocks would be summed twice or not at all."""
    import re
    import subprocess
    csrc = os.path.join(ROOT, "lpcnet_amd", "csrc")
    gen = os.path.join(ROOT, "tools", "gen_grub_asm.py")
    m = re.search(r"#define LPCN_PROD_BLOCKS (\d+)", open(os.path.join(csrc, "sample_kernel.hip.h")).read())
    prod = int(m.group(1))
    assert 0 < prod < 96 and prod % 8 == 0 and (96 - prod) % 8 == 0
    assert (prod + 7) * 48 * 16 + 113392 - 6144 <= 160 * 1024      # the products (+ the 7 blocks the chain wave's ring reads ahead) + the rest of the single-stream carve-up (107 248 B since the unused row table went) fit the CU's LDS
    cases = [(["--lds", "1"], "grub_lds_loop_s1.inc"), (["--lds", "2"], "grub_lds_loop_s2.inc"), (["--lds", "4"], "grub_lds_loop_s4.inc"),
             (["--lds", "1", "--blocks", str(96 - prod), "--name", "LPCN_GRUB_LDS32_CLOBBERS"], "grub_lds_loop_s1_first.inc"),
             (["--prod", str(prod)], "grub_prod_loop.inc")]
    # (round 6: the loops of forms that were measured and not kept -- the scalar-state loop, the product ring of four streams, the int8 loop, two / four
    # streams per chain wave -- are no longer committed; the generator still emits them for tools/ubench and the record in EXPERIMENTS.md)
    for args, name in cases:
        out = subprocess.run([sys.executable, gen] + args, capture_output=True, text=True, check=True).stdout
        assert out == open(os.path.join(csrc, name)).read(), name
