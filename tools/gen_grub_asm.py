#!/usr/bin/env python3
"""Generate lpcnet_amd/csrc/grub_scalar_loop.inc: the hand-scheduled GRU-B input mat-vec loop of the PARITY float kernel
with the state operand in SGPRs (sample_kernel.hip.h, `gb_scalar`).

    python tools/gen_grub_asm.py > lpcnet_amd/csrc/grub_scalar_loop.inc

Why one assembly block: the state arrives through s_load_dwordx16 (SMEM returns out of order -> every wait is lgkmcnt(0)),
and a scalar load must never be in flight across an inline-asm boundary -- the register allocator may spill or copy an
SGPR tuple it believes defined while the load is still filling it.  So loads, waits and consumers live in ONE block that
names its registers itself (declared as clobbers).

Per output row (lane) and stream: zrh += sum over 96 blocks x 4 columns of w * h, every product and every sum rounded
separately, in block order / column order (src/vec.h:355-401).  The chain of 384 dependent v_add_f32 is the floor
(~6.6 clk each); everything else sits in its shadow:
  * products of block b+1 (2 v_pk_mul_f32: SGPR pair x VGPR pair) are formed during the adds of block b;
  * a group = 4 blocks = one s_load_dwordx16 (64 B of state) + four ds_read_b128 (the lane's weights); two buffers;
    the loads of group g+2 are issued at block 3 of group g, right after the ONE s_waitcnt lgkmcnt(0) of the group,
    which at that point guarantees group g+1 -- a full group (~110 clk) of lookahead for both LDS and scalar cache.
Register use (fixed, clobbered): s[36:51] / s[52:67] state tuples, s68/s69 offsets, s70 trip counter, s71 touch target;
v[216:231] / v[232:247] weights, v[248:251] / v[252:255] products.
"""
GROUPS = 24            # 96 blocks
GROUPS_PER_TRIP = 4    # 6 trips; the taken branch at the end of a trip costs ~35 clk
T = {0: 36, 1: 52}     # SGPR tuple base per buffer
W = {0: 216, 1: 232}   # VGPR weight base per buffer (4 blocks x float4)
P = {0: 248, 1: 252}   # product registers (alternate per block)
OFF, OFFC, CNT, DUMMY = 68, 69, 70, 71
import sys
PHASE = int(sys.argv[sys.argv.index("--phase") + 1]) if "--phase" in sys.argv else 0
MASK48 = "--mask48" in sys.argv
RING = int(sys.argv[sys.argv.index("--ring") + 1]) if "--ring" in sys.argv else 4
NBLK = int(sys.argv[sys.argv.index("--blocks") + 1]) if "--blocks" in sys.argv else 96     # --lds: blocks 0..NBLK-1 only (the rest arrives as products, --prod)
NAME = sys.argv[sys.argv.index("--name") + 1] if "--name" in sys.argv else "LPCN_GRUB_LDS_CLOBBERS"


def pk_mul(dst, sreg, vreg):
    return f"v_pk_mul_f32 v[{dst}:{dst + 1}], s[{sreg}:{sreg + 1}], v[{vreg}:{vreg + 1}]"


def products(pbuf, tbuf, blk):
    """products of block `blk` (0..3) of the group in buffer tbuf -> product set pbuf"""
    return [pk_mul(P[pbuf], T[tbuf] + 4 * blk, W[tbuf] + 4 * blk), pk_mul(P[pbuf] + 2, T[tbuf] + 4 * blk + 2, W[tbuf] + 4 * blk + 2)]


def adds(pbuf):
    return [f"v_add_f32 %[z], %[z], v{P[pbuf] + k}" for k in range(4)]


def interleave(a, fill):
    """one filler after each dependent add (the add's latency is the slot)"""
    out = []
    for i, x in enumerate(a):
        out.append(x)
        if i < len(fill):
            out.append(fill[i])
    out += fill[len(a):]
    return out


def group(k, buf, lds_off_next2):
    """group in buffer `buf`; entering: product set 0 holds block 0's products.  lds_off_next2 = byte offset (from %[wp]) of the
    weights of the group two ahead."""
    o = 1 - buf
    code = []
    # blocks 0..2: adds of block b, products of block b+1 (same group)
    pset = 0
    for b in range(3):
        code += interleave(adds(pset), products(1 - pset, buf, b + 1))
        pset = 1 - pset
    # block 3: the group's one wait; products of the NEXT group's block 0; loads of the group two ahead into this group's buffers
    loads = [f"s_add_u32 s{OFF}, s{OFF}, 64",
             f"s_min_u32 s{OFFC}, s{OFF}, 1472",
             f"s_load_dwordx16 s[{T[buf]}:{T[buf] + 15}], %[hb], s{OFFC}"]
    loads += [f"ds_read_b128 v[{W[buf] + 4 * j}:{W[buf] + 4 * j + 3}], %[wp] offset:{lds_off_next2 + 128 * j}" for j in range(4)]
    a = adds(pset)
    nxt = products(1 - pset, o, 0)
    code.append("s_waitcnt lgkmcnt(0)")
    code += [a[0], nxt[0], a[1], nxt[1], loads[0], loads[1], loads[2], a[2], loads[3], loads[4], a[3], loads[5], loads[6]]
    assert pset == 1            # after three alternations block 3's products sit in set 1, the next group's block 0 in set 0
    return code


def main():
    lines = []
    # prologue: groups 0 and 1 -> buffers 0 and 1, products of (group 0, block 0)
    # The scalar cache still holds the previous sample's copy of these addresses: drop it, then touch all 24 lines of the
    # stream at once so that the L2 round trips (~280 clk each) overlap; the touches land in a register of this block
    # (a load must not be in flight into a register the compiler may reuse).
    lines += ["s_dcache_inv",
              f"s_mov_b32 s{OFF}, 64",
              f"s_mov_b32 s{CNT}, {GROUPS // GROUPS_PER_TRIP}",
              f"s_load_dwordx16 s[{T[0]}:{T[0] + 15}], %[hb], 0x0"]
    lines += [f"s_load_dwordx16 s[{T[1]}:{T[1] + 15}], %[hb], 0x40"]
    lines += [f"s_load_dword s{DUMMY}, %[hb], {hex(64 * k)}" for k in range(2, GROUPS)]
    lines += [f"ds_read_b128 v[{W[0] + 4 * j}:{W[0] + 4 * j + 3}], %[wp] offset:{128 * j}" for j in range(4)]
    lines += [f"ds_read_b128 v[{W[1] + 4 * j}:{W[1] + 4 * j + 3}], %[wp] offset:{512 + 128 * j}" for j in range(4)]
    lines += ["s_waitcnt lgkmcnt(0)"] + products(0, 0, 0)
    # Code placement: a hand-written stream is sensitive to where its loop body starts modulo 8 bytes (MI355X_MICROARCH.md,
    # "Code-placement sensitivity": a byte-identical stream lost 13 % under a 4-mod-8 shift; here an unrelated edit that moved
    # the block by 4 bytes cost the whole kernel 3.5 %).  The loop head is therefore pinned to a 16-byte boundary
    # (+ PHASE x 4 bytes, chosen by measurement: see LPCN_GRUB_PHASE in sample_kernel.hip.h).
    lines += [".p2align 4"] + ["s_nop 0"] * PHASE
    lines += ["1:"]
    for k in range(GROUPS_PER_TRIP):
        lines += group(k, k & 1, (k + 2) * 512)
    lines += [f"v_add_u32 %[wp], {GROUPS_PER_TRIP * 512}, %[wp]",
              f"s_sub_u32 s{CNT}, s{CNT}, 1",
              f"s_cmp_lg_u32 s{CNT}, 0",
              "s_cbranch_scc1 1b",
              "s_waitcnt lgkmcnt(0)"]
    print("// generated by tools/gen_grub_asm.py -- do not edit")
    print("// operands: %[z] float accumulator (in/out VGPR), %[wp] LDS byte address of the lane's row, block 0 (in/out VGPR), %[hb] state mirror (SGPR pair)")
    clob = [f"s{i}" for i in range(36, 72)] + [f"v{i}" for i in range(216, 256)]
    print("#undef LPCN_GRUB_SCALAR_CLOBBERS")
    print("#define LPCN_GRUB_SCALAR_CLOBBERS " + ", ".join('"%s"' % c for c in clob) + ', "scc", "memory"')
    for ln in lines:
        print('"%s\\n\\t"' % ln)


def main_lds(S):
    """GRU-B input mat-vec with the state operand read from LDS as a BROADCAST (every lane the same 16 bytes: block b of the
    wave's stream) -- no L2 mirror, no scalar cache: the state a GRU-B wave needs is the one the gate stage has just written
    to LDS for GRU-A's items.  Two ds_read_b128 per block (the lane's weights + the state block), ring of 4 blocks each,
    products of block b+1 formed during the adds of block b.  LDS returns in order, so the waits are exact counts.
    Layout of the state: [block][stream][4] floats, 16 B pad every 4 blocks (Lds<S>::ha_off)."""
    stride = 16 * S
    ha_off = lambda p: p * stride + (p >> 2) * 16
    R = RING                       # ring slots = blocks in flight + 1
    base = 256 - 8 - 8 * R
    PS = [base, base + 4]          # product sets
    WR = [base + 8 + 4 * i for i in range(R)]              # weight ring (float4 per block)
    HR = [base + 8 + 4 * R + 4 * i for i in range(R)]      # state ring
    CNT = 70
    BPT = 16 if 96 % R else (R * (16 // R) if 96 % (R * (16 // R)) == 0 else 12)     # blocks per trip: a multiple of the ring size that divides 96
    if NBLK % BPT and R == 4 and NBLK % 8 == 0: BPT = 8
    if BPT % R: BPT = {4: 16, 5: 15, 6: 12, 3: 12, 8: 16}.get(R, 16)
    TAIL = NBLK % BPT              # blocks behind the last full trip, straight-line (--blocks 86: 5 trips of 16 + 6)
    assert TAIL == 0 or (R == 4 and TAIL % 2 == 0) or R == 5
    LA = R - 1
    lines = []
    rdw = lambda slot, blk: f"ds_read_b128 v[{WR[slot]}:{WR[slot] + 3}], %[wp] offset:{blk * 128}"
    rdh = lambda slot, blk: f"ds_read_b128 v[{HR[slot]}:{HR[slot] + 3}], %[hp] offset:{ha_off(blk)}"
    prod = lambda pset, slot: [f"v_pk_mul_f32 v[{PS[pset]}:{PS[pset] + 1}], v[{HR[slot]}:{HR[slot] + 1}], v[{WR[slot]}:{WR[slot] + 1}]",
                               f"v_pk_mul_f32 v[{PS[pset] + 2}:{PS[pset] + 3}], v[{HR[slot] + 2}:{HR[slot] + 3}], v[{WR[slot] + 2}:{WR[slot] + 3}]"]
    # (entry: nothing of the compiler's may be in flight under the partial waits below -- a scalar load returns out of order)
    lines += ["s_waitcnt lgkmcnt(0)", f"s_mov_b32 s{CNT}, {NBLK // BPT}"]
    if MASK48:          # only the 48 row lanes take part: a quarter less LDS return traffic per read
        lines += ["s_mov_b64 s[72:73], exec", "s_bfm_b64 exec, 48, 0"]
    for b in range(LA):
        lines += [rdw(b % R, b), rdh(b % R, b)]
    lines += [f"s_waitcnt lgkmcnt({2 * (LA - 1)})"] + prod(0, 0)
    lines += [".p2align 4"] + ["s_nop 0"] * PHASE
    lines += ["1:"]
    for k in range(BPT):
        pr = prod((k + 1) & 1, (k + 1) % R)
        a = [f"v_add_f32 %[z], %[z], v{PS[k & 1] + j}" for j in range(4)]
        lines += [rdw((k + LA) % R, k + LA), rdh((k + LA) % R, k + LA), f"s_waitcnt lgkmcnt({2 * (LA - 1)})",
                  a[0], pr[0], a[1], pr[1], a[2], a[3]]
    assert BPT % 2 == 0 and BPT % R == 0
    lines += [f"v_add_u32 %[wp], {BPT * 128}, %[wp]",
              f"v_add_u32 %[hp], {ha_off(BPT)}, %[hp]",
              f"s_sub_u32 s{CNT}, s{CNT}, 1",
              f"s_cmp_lg_u32 s{CNT}, 0",
              "s_cbranch_scc1 1b"]
    for k in range(TAIL if R == 4 else 0):     # (the pointers have been advanced: the tail's offsets start over; it reads LA blocks past its end like the last trip)
        pr = prod((k + 1) & 1, (k + 1) % R)
        a = [f"v_add_f32 %[z], %[z], v{PS[k & 1] + j}" for j in range(4)]
        lines += [rdw((k + LA) % R, k + LA), rdh((k + LA) % R, k + LA), f"s_waitcnt lgkmcnt({2 * (LA - 1)})",
                  a[0], pr[0], a[1], pr[1], a[2], a[3]]
    lines += ["s_waitcnt lgkmcnt(0)"]
    if MASK48:
        lines += ["s_mov_b64 exec, s[72:73]"]
    print("// generated by tools/gen_grub_asm.py --lds %d -- do not edit" % S)
    print("// operands: %[z] float accumulator (in/out VGPR), %[wp] LDS byte address of the lane's row, block 0 (in/out VGPR), %[hp] LDS byte address of the stream's state, block 0 (in/out VGPR)")
    clob = [f"s{CNT}"] + (["s72", "s73"] if MASK48 else []) + [f"v{i}" for i in range(base, 256)]
    name = NAME
    print("#undef " + name)
    print("#define " + name + " " + ", ".join('"%s"' % c for c in clob) + ', "scc", "memory"')
    for ln in lines:
        print('"%s\\n\\t"' % ln)


def main_lds2(S):
    """GRU-B input mat-vec of TWO streams on one wave (round 6, sample_kernel_x2.hip.h): the lane's weight read of a block serves both streams, whose
    state blocks -- neighbours in the [block][stream][4] layout -- arrive as two broadcast reads: three ds_read_b128 per block for two streams
    instead of four, and two independent chains of dependent adds per lane, so the adds of the one sit in the latency of the other.  Per block: 8 v_add_f32
    (two chains of four, interleaved), 4 v_pk_mul_f32 for the next block, 3 reads for the block RING - 1 ahead.  Every product and every sum is
    rounded on its own, per stream in block order, columns 0..3 (src/vec.h:355-401).
    Registers: product sets base+0..15 (set x {A, B} x 4), weight ring, state ring (A and B of a block side by side)."""
    stride = 16 * S
    ha_off = lambda p: p * stride + (p >> 2) * 16
    R = RING
    base = 256 - 16 - 12 * R
    PS = [[base, base + 4], [base + 8, base + 12]]          # PS[set][stream]
    WR = [base + 16 + 4 * i for i in range(R)]
    HR = [[base + 16 + 4 * R + 8 * i, base + 16 + 4 * R + 8 * i + 4] for i in range(R)]      # HR[slot][stream]
    CNT = 70
    BPT = {3: 12, 4: 16, 6: 12}[R]
    assert 96 % BPT == 0 and BPT % R == 0 and BPT % 2 == 0
    LA = R - 1
    rd = lambda slot, blk: [f"ds_read_b128 v[{WR[slot]}:{WR[slot] + 3}], %[wp] offset:{blk * 128}",
                            f"ds_read_b128 v[{HR[slot][0]}:{HR[slot][0] + 3}], %[hp] offset:{ha_off(blk)}",
                            f"ds_read_b128 v[{HR[slot][1]}:{HR[slot][1] + 3}], %[hp] offset:{ha_off(blk) + 16}"]
    def prod(pset, slot):
        out = []
        for st in range(2):
            out += [f"v_pk_mul_f32 v[{PS[pset][st]}:{PS[pset][st] + 1}], v[{HR[slot][st]}:{HR[slot][st] + 1}], v[{WR[slot]}:{WR[slot] + 1}]",
                    f"v_pk_mul_f32 v[{PS[pset][st] + 2}:{PS[pset][st] + 3}], v[{HR[slot][st] + 2}:{HR[slot][st] + 3}], v[{WR[slot] + 2}:{WR[slot] + 3}]"]
        return out
    def block(k):
        pr = prod((k + 1) & 1, (k + 1) % R)
        a = [[f"v_add_f32 %[z{'ab'[st]}], %[z{'ab'[st]}], v{PS[k & 1][st] + j}" for j in range(4)] for st in range(2)]
        return rd((k + LA) % R, k + LA) + [f"s_waitcnt lgkmcnt({3 * (LA - 1)})",
                a[0][0], a[1][0], pr[0], a[0][1], a[1][1], pr[1], a[0][2], a[1][2], pr[2], a[0][3], a[1][3], pr[3]]
    lines = ["s_waitcnt lgkmcnt(0)", f"s_mov_b32 s{CNT}, {96 // BPT}"]
    for b in range(LA):
        lines += rd(b % R, b)
    lines += [f"s_waitcnt lgkmcnt({3 * (LA - 1)})"] + prod(0, 0)
    lines += [".p2align 4"] + ["s_nop 0"] * PHASE
    lines += ["1:"]
    for k in range(BPT):
        lines += block(k)
    lines += [f"v_add_u32 %[wp], {BPT * 128}, %[wp]",
              f"v_add_u32 %[hp], {ha_off(BPT)}, %[hp]",
              f"s_sub_u32 s{CNT}, s{CNT}, 1",
              f"s_cmp_lg_u32 s{CNT}, 0",
              "s_cbranch_scc1 1b",
              "s_waitcnt lgkmcnt(0)"]
    print("// generated by tools/gen_grub_asm.py --lds2 %d --ring %d -- do not edit" % (S, R))
    print("// operands: %[za], %[zb] float accumulators of streams s and s + 1 (in/out VGPRs), %[wp] LDS byte address of the lane's row, block 0 (in/out VGPR), %[hp] LDS byte address of stream s's state, block 0 (in/out VGPR)")
    clob = [f"s{CNT}"] + [f"v{i}" for i in range(base, 256)]
    print("#undef LPCN_GRUB_LDS2_CLOBBERS")
    print("#define LPCN_GRUB_LDS2_CLOBBERS " + ", ".join('"%s"' % c for c in clob) + ', "scc", "memory"')
    for ln in lines:
        print('"%s\\n\\t"' % ln)


def main_lds2p(S):
    """GRU-B input mat-vec of TWO streams on one wave, PACKED over the stream pair (round 6, sample_kernel_x2.hip.h): the two accumulators live in one
    register pair, a block is 4 v_pk_add_f32 (z_a, z_b) += (p_a, p_b) -- each half rounded on its own, per stream in block order, columns 0..3
    (src/vec.h:355-401) -- and 4 v_pk_mul_f32 for the next block: (h_a, h_b)[column j] x the lane's weight w_j broadcast to both halves through
    op_sel.  The state comes from the NEURON-major copy of the GRU-A state ([neuron][stream]: the pair of a column is 8 adjacent bytes, the four
    columns of a block two ds_read2_b64), the weights as before: 3 LDS reads + 8 packed VALU instructions per block for two streams, against
    2 + 6 for one stream in the single-stream loop and 3 + 12 in the unpacked two-stream loop (--lds2: 92 clk per block measured).
    Registers: product sets base + 0..15, weight ring 4 per slot, state ring 8 per slot."""
    R = RING
    base = 256 - 16 - 12 * R
    PS = [base, base + 8]                                    # PS[set] + 2 j = pair (stream a, stream b) of column j
    WR = [base + 16 + 4 * i for i in range(R)]
    HR = [base + 16 + 4 * R + 8 * i for i in range(R)]      # HR[slot] + 2 j = (h_a, h_b) of column j
    CNT = 70
    BPT = {3: 12, 4: 16, 6: 12}[R]
    assert 96 % BPT == 0 and BPT % R == 0 and BPT % 2 == 0
    LA = R - 1
    stride = 4 * S * 4                                       # bytes per block in the [neuron][stream] layout
    assert stride % 8 == 0 and (stride * (BPT + LA)) // 8 + 6 < 256
    rd = lambda slot, blk: [f"ds_read_b128 v[{WR[slot]}:{WR[slot] + 3}], %[wp] offset:{blk * 128}",
                            f"ds_read2_b64 v[{HR[slot]}:{HR[slot] + 3}], %[hp] offset0:{blk * stride // 8} offset1:{blk * stride // 8 + 4 * S // 8 * 1}",
                            f"ds_read2_b64 v[{HR[slot] + 4}:{HR[slot] + 7}], %[hp] offset0:{blk * stride // 8 + 2 * (4 * S // 8)} offset1:{blk * stride // 8 + 3 * (4 * S // 8)}"]
    def prod(pset, slot):
        out = []
        for j in range(4):
            wpair = WR[slot] + 2 * (j >> 1)
            sel = "op_sel:[0,0] op_sel_hi:[1,0]" if (j & 1) == 0 else "op_sel:[0,1] op_sel_hi:[1,1]"
            out.append(f"v_pk_mul_f32 v[{PS[pset] + 2 * j}:{PS[pset] + 2 * j + 1}], v[{HR[slot] + 2 * j}:{HR[slot] + 2 * j + 1}], v[{wpair}:{wpair + 1}] {sel}")
        return out
    def block(k):
        pr = prod((k + 1) & 1, (k + 1) % R)
        a = [f"v_pk_add_f32 %[z], %[z], v[{PS[k & 1] + 2 * j}:{PS[k & 1] + 2 * j + 1}]" for j in range(4)]
        return rd((k + LA) % R, k + LA) + [f"s_waitcnt lgkmcnt({3 * (LA - 1)})", a[0], pr[0], a[1], pr[1], a[2], pr[2], a[3], pr[3]]
    lines = ["s_waitcnt lgkmcnt(0)", f"s_mov_b32 s{CNT}, {96 // BPT}"]
    for b in range(LA):
        lines += rd(b % R, b)
    lines += [f"s_waitcnt lgkmcnt({3 * (LA - 1)})"] + prod(0, 0)
    lines += [".p2align 4"] + ["s_nop 0"] * PHASE
    lines += ["1:"]
    for k in range(BPT):
        lines += block(k)
    lines += [f"v_add_u32 %[wp], {BPT * 128}, %[wp]",
              f"v_add_u32 %[hp], {BPT * stride}, %[hp]",
              f"s_sub_u32 s{CNT}, s{CNT}, 1",
              f"s_cmp_lg_u32 s{CNT}, 0",
              "s_cbranch_scc1 1b",
              "s_waitcnt lgkmcnt(0)"]
    print("// generated by tools/gen_grub_asm.py --lds2p %d --ring %d -- do not edit" % (S, R))
    print("// operands: %[z] accumulator pair (stream s, stream s + 1) (in/out, 64-bit VGPR pair), %[wp] LDS byte address of the lane's row, block 0 (in/out VGPR), %[hp] LDS byte address of (neuron 0, stream s) in the [neuron][stream] state (in/out VGPR)")
    clob = [f"s{CNT}"] + [f"v{i}" for i in range(base, 256)]
    print("#undef LPCN_GRUB_LDS2P_CLOBBERS")
    print("#define LPCN_GRUB_LDS2P_CLOBBERS " + ", ".join('"%s"' % c for c in clob) + ', "scc", "memory"')
    for ln in lines:
        print('"%s\\n\\t"' % ln)


def main_lds4m(S):
    """GRU-B input mat-vec of ALL FOUR streams of a group on ONE wave, products on the matrix pipe (round 6, sample_kernel_x2.hip.h).
    What bounds the single-stream loop (--lds) is the wave's DS issue cadence -- ~25 clk per LDS instruction whatever its width
    (profiles/r03_ubench_grub_scalar.txt), two reads per block -- not its 4 dependent adds.  So the reads have to serve more streams: lane (quad q,
    i) fetches the 16 bytes of stream i's state block (the [block][stream][4] layout GRU-A's items use) and its own row's weights, and
    v_mfma_f32_4x4x1 with C = -0.0 gives lane j, in register i, h_i[c] * w_j[c] rounded once -- bit for bit v_mul_f32's product
    (tests: the matrix-pipe identity) -- for the four streams of the quad: TWO reads + 4 MFMA + 8 v_pk_add_f32 per block for four streams.  The sums
    are two packed chains (streams 0 / 1, streams 2 / 3), each half rounded on its own, per stream in block order, columns 0..3 (src/vec.h:355-401);
    the MFMAs of block b + 1 are issued between the adds of block b.
    Registers: -0.0 quad, two product sets of 4 quads, weight ring, state ring."""
    assert S == 4
    stride = 16 * S
    ha_off = lambda p: p * stride + (p >> 2) * 16
    R = RING
    base = 256 - 4 - 32 - 8 * R
    NZ = base
    PS = [base + 4, base + 20]                               # PS[set] + 4 c = product quad of column c (register i = stream i)
    WR = [base + 36 + 4 * i for i in range(R)]
    HR = [base + 36 + 4 * R + 4 * i for i in range(R)]
    CNT = 70
    BPT = {3: 12, 4: 16, 6: 12}[R]
    assert 96 % BPT == 0 and BPT % R == 0 and BPT % 2 == 0
    LA = R - 1
    rd = lambda slot, blk: [f"ds_read_b128 v[{WR[slot]}:{WR[slot] + 3}], %[wp] offset:{blk * 128}",
                            f"ds_read_b128 v[{HR[slot]}:{HR[slot] + 3}], %[hp] offset:{ha_off(blk)}"]
    mf = lambda pset, slot, c: f"v_mfma_f32_4x4x1_16b_f32 v[{PS[pset] + 4 * c}:{PS[pset] + 4 * c + 3}], v{HR[slot] + c}, v{WR[slot] + c}, v[{NZ}:{NZ + 3}]"
    def block(k):
        out = rd((k + LA) % R, k + LA) + [f"s_waitcnt lgkmcnt({2 * (LA - 1)})"]
        for c in range(4):
            out += [mf((k + 1) & 1, (k + 1) % R, c),
                    f"v_pk_add_f32 %[za], %[za], v[{PS[k & 1] + 4 * c}:{PS[k & 1] + 4 * c + 1}]",
                    f"v_pk_add_f32 %[zb], %[zb], v[{PS[k & 1] + 4 * c + 2}:{PS[k & 1] + 4 * c + 3}]"]
        return out
    lines = ["s_waitcnt lgkmcnt(0)", f"s_mov_b32 s{CNT}, {96 // BPT}"]
    lines += [f"v_mov_b32 v{NZ + i}, 0x80000000" for i in range(4)]
    for b in range(LA):
        lines += rd(b % R, b)
    lines += [f"s_waitcnt lgkmcnt({2 * (LA - 1)})"] + [mf(0, 0, c) for c in range(4)]
    lines += [".p2align 4"] + ["s_nop 0"] * PHASE
    lines += ["1:"]
    for k in range(BPT):
        lines += block(k)
    lines += [f"v_add_u32 %[wp], {BPT * 128}, %[wp]",
              f"v_add_u32 %[hp], {ha_off(BPT)}, %[hp]",
              f"s_sub_u32 s{CNT}, s{CNT}, 1",
              f"s_cmp_lg_u32 s{CNT}, 0",
              "s_cbranch_scc1 1b",
              "s_waitcnt lgkmcnt(0)",
              "s_nop 7"]                                     # (the last trip's look-ahead MFMAs drain before the compiler's code may touch the clobbered registers)
    print("// generated by tools/gen_grub_asm.py --lds4m %d --ring %d -- do not edit" % (S, R))
    print("// operands: %[za] accumulator pair (streams 0, 1), %[zb] (streams 2, 3) (in/out, 64-bit VGPR pairs), %[wp] LDS byte address of the lane's row, block 0 (in/out VGPR), %[hp] LDS byte address of block 0 of stream (lane & 3) in the [block][stream][4] state (in/out VGPR)")
    clob = [f"s{CNT}"] + [f"v{i}" for i in range(base, 256)]
    print("#undef LPCN_GRUB_LDS4M_CLOBBERS")
    print("#define LPCN_GRUB_LDS4M_CLOBBERS " + ", ".join('"%s"' % c for c in clob) + ', "scc", "memory"')
    for ln in lines:
        print('"%s\\n\\t"' % ln)


def main_rl(S):
    """GRU-B input mat-vec with the state operand in SGPRs again -- but filled from LDS, not from L2: ONE ds_read_b128 per 16
    blocks brings block b0 + L of the wave's stream to lane L (L < 16), and v_readlane_b32 moves the four values of the block
    about to be used into an SGPR quad, which v_pk_mul_f32 takes as its scalar operand pair.  Per block: 1 LDS read (the lane's
    weights) instead of 2, + 4 v_readlane issued two blocks ahead in the shadow of the dependent adds.
    MEASURED AND REJECTED (round 4): bit-exact, 117.6 vs 125.2 M samples/s -- four v_readlane per block cost more issue time on the
    GRU-B wave than the broadcast LDS read they replace (the kernel does not include this variant; the generator is kept as the record).
    Registers: weights ring v[WR], state double buffer v[HS], products v[PS], SGPR quads s[36:51]."""
    stride = 16 * S
    ha_off = lambda p: p * stride + (p >> 2) * 16
    R = 4
    base = 256 - 8 - 4 * R - 8
    PS = [base, base + 4]
    WR = [base + 8 + 4 * i for i in range(R)]
    HS = [base + 8 + 4 * R, base + 8 + 4 * R + 4]         # state of 16 blocks per buffer: lane L = block b0 + L
    SQ = [36, 40, 44, 48]                                  # SGPR quads (a ring of four: the 32-block loop body must return to quad 0)
    CNT = 70
    BPT = 16
    LA = R - 1
    lines = []
    rdw = lambda slot, blk: f"ds_read_b128 v[{WR[slot]}:{WR[slot] + 3}], %[wp] offset:{blk * 128}"
    rl = lambda q, buf, lane: [f"v_readlane_b32 s{SQ[q] + j}, v{HS[buf] + j}, {lane}" for j in range(4)]
    prod = lambda pset, slot, q: [f"v_pk_mul_f32 v[{PS[pset]}:{PS[pset] + 1}], s[{SQ[q]}:{SQ[q] + 1}], v[{WR[slot]}:{WR[slot] + 1}]",
                                  f"v_pk_mul_f32 v[{PS[pset] + 2}:{PS[pset] + 3}], s[{SQ[q] + 2}:{SQ[q] + 3}], v[{WR[slot] + 2}:{WR[slot] + 3}]"]
    # %[hl] = LDS address of (stream's state, block = lane & 15) ; advanced by 16 blocks per trip
    lines += [f"s_mov_b32 s{CNT}, {96 // BPT}",
              f"ds_read_b128 v[{HS[0]}:{HS[0] + 3}], %[hl]",
              f"ds_read_b128 v[{HS[1]}:{HS[1] + 3}], %[hl] offset:{ha_off(16)}"]
    for b in range(LA):
        lines += [rdw(b % R, b)]
    lines += [f"s_waitcnt lgkmcnt({LA})"]                   # both state buffers are in (LDS returns in order); weights of block 0.. still in flight
    lines += rl(0, 0, 0) + rl(1, 0, 1)
    lines += [f"s_waitcnt lgkmcnt({LA - 1})"] + prod(0, 0, 0)
    lines += [".p2align 4"] + ["s_nop 0"] * PHASE
    # two trips per loop iteration so that the state double buffer alternates statically
    lines += ["1:"]
    for half in range(2):
        buf = half
        for k in range(BPT):
            kk = half * BPT + k                            # block index inside the iteration (0..31)
            pr = prod((kk + 1) & 1, (kk + 1) % R, (kk + 1) % 4)
            a = [f"v_add_f32 %[z], %[z], v{PS[kk & 1] + j}" for j in range(4)]
            nb, nl = (buf, k + 2) if k + 2 < BPT else (1 - buf, k + 2 - BPT)       # the quad two blocks ahead: buffer, lane
            r2 = rl((kk + 2) % 4, nb, nl)
            seq = [rdw((kk + LA) % R, kk + LA)]
            if k == 2:                                     # this buffer's successor: the state of the 16 blocks two trips ahead is fetched
                pass
            seq += [f"s_waitcnt lgkmcnt({LA - 1})", a[0], pr[0], r2[0], a[1], pr[1], r2[1], a[2], r2[2], a[3], r2[3]]
            lines += seq
            if k == BPT - 1:
                # the buffer just finished is refilled with the state two trips ahead (its readlanes are all done: the last quad of
                # this buffer was read two blocks ago)
                lines += [f"ds_read_b128 v[{HS[buf]}:{HS[buf] + 3}], %[hl] offset:{ha_off(16 * (half + 2))}", "s_waitcnt lgkmcnt(%d)" % LA]
    lines += [f"v_add_u32 %[wp], {2 * BPT * 128}, %[wp]",
              f"v_add_u32 %[hl], {ha_off(2 * BPT)}, %[hl]",
              f"s_sub_u32 s{CNT}, s{CNT}, 2",
              f"s_cmp_lg_u32 s{CNT}, 0",
              "s_cbranch_scc1 1b",
              "s_waitcnt lgkmcnt(0)"]
    print("// generated by tools/gen_grub_asm.py --rl %d -- do not edit" % S)
    print("// operands: %[z] float accumulator (in/out VGPR), %[wp] LDS byte address of the lane's row, block 0 (in/out VGPR), %[hl] LDS byte address of the state block (lane & 15) of the wave's stream (in/out VGPR)")
    clob = [f"s{CNT}"] + [f"s{i}" for i in range(36, 52)] + [f"v{i}" for i in range(base, 256)]
    print("#undef LPCN_GRUB_RL_CLOBBERS")
    print("#define LPCN_GRUB_RL_CLOBBERS " + ", ".join('"%s"' % c for c in clob) + ', "scc", "memory"')
    for ln in lines:
        print('"%s\\n\\t"' % ln)


def main_prod(nblk):
    """The tail of the single-stream GRU-B chain with the PRODUCTS ready in LDS: idle waves of the workgroup have multiplied
    weight x state for blocks 96-nblk..95 (one float4 per row and block, 768 bytes per block: 48 rows) while the chain wave
    summed the first blocks itself.  Per block: ONE ds_read_b128 and the four dependent v_add_f32 -- the chain at its floor.
    Ring of 8 blocks in flight (7 x 27 clk of lookahead > the LDS latency)."""
    R = 8
    base = 256 - 4 * R
    RG = [base + 4 * i for i in range(R)]
    CNT = 70
    BPT = 16 if nblk % 16 == 0 else 8
    assert nblk % BPT == 0
    LA = R - 1
    rd = lambda slot, blk: f"ds_read_b128 v[{RG[slot]}:{RG[slot] + 3}], %[pp] offset:{blk * 768}"
    lines = ["s_waitcnt lgkmcnt(0)", f"s_mov_b32 s{CNT}, {nblk // BPT}"]      # (entry: see --lds)
    for b in range(LA):
        lines += [rd(b % R, b)]
    lines += [f"s_waitcnt lgkmcnt({LA - 1})", ".p2align 4", "1:"]
    for k in range(BPT):
        a = [f"v_add_f32 %[z], %[z], v{RG[k % R] + j}" for j in range(4)]
        # the read of the block seven ahead issues in the shadow of the first dependent add; the wait for the NEXT block's
        # products sits behind the last add (LDS returns in order: at most LA - 1 younger reads may still be out)
        lines += [a[0], rd((k + LA) % R, k + LA), a[1], a[2], a[3], f"s_waitcnt lgkmcnt({LA - 1})"]          # (the last trip reads LA blocks past the end: Lds<1>::prod_sz pads for them)
    lines += [f"v_add_u32 %[pp], {BPT * 768}, %[pp]",
              f"s_sub_u32 s{CNT}, s{CNT}, 1",
              f"s_cmp_lg_u32 s{CNT}, 0",
              "s_cbranch_scc1 1b",
              "s_waitcnt lgkmcnt(0)"]
    print("// generated by tools/gen_grub_asm.py --prod %d -- do not edit" % nblk)
    print("// operands: %[z] float accumulator (in/out VGPR), %[pp] LDS byte address of the lane's row in the first product block (in/out VGPR)")
    clob = [f"s{CNT}"] + [f"v{i}" for i in range(base, 256)]
    print("#undef LPCN_GRUB_PROD_CLOBBERS")
    print("#define LPCN_GRUB_PROD_CLOBBERS " + ", ".join('"%s"' % c for c in clob) + ', "scc", "memory"')
    for ln in lines:
        print('"%s\\n\\t"' % ln)


RING_SEG = (4, 2, 4)     # blocks of a stream's product ring per LDS region (update/reset pre-activations, candidate inputs, free tail)


def main_ring_fill(S, first):
    """S = 4, float PARITY: the PRODUCER side of GRU-B's product ring.  Wave 4 + s (idle until its candidate heads start) multiplies
    weight x state for blocks first..95 of stream s -- one row per lane, the same two 16-byte reads per block as the chain wave's loop
    -- and stores the four products of (row, block) where the chain wave picks them up (main_ring_sum).  The ring lives in cells that
    are dead while GRU-B runs: the update / reset pre-activations (4 blocks per stream), the candidate inputs (2) and the free tail (4).
    Uses the chain loop's register block (v216..v255 are reserved by it anyway): ring of four blocks, products formed in place."""
    stride = 16 * S
    ha_off = lambda p: p * stride + (p >> 2) * 16
    n = 96 - first
    assert sum(RING_SEG) == n
    R = 4
    base = 256 - 8 * R
    WR = [base + 4 * i for i in range(R)]
    HR = [base + 4 * R + 4 * i for i in range(R)]
    seg_of = []
    for si, cnt in enumerate(RING_SEG):
        seg_of += [(si, k) for k in range(cnt)]
    rdw = lambda b: f"ds_read_b128 v[{WR[b % R]}:{WR[b % R] + 3}], %[wp] offset:{(first + b) * 128}"
    rdh = lambda b: f"ds_read_b128 v[{HR[b % R]}:{HR[b % R] + 3}], %[hp] offset:{ha_off(first + b)}"
    def body(b):
        si, k = seg_of[b]
        return [f"v_pk_mul_f32 v[{HR[b % R]}:{HR[b % R] + 1}], v[{HR[b % R]}:{HR[b % R] + 1}], v[{WR[b % R]}:{WR[b % R] + 1}]",
                f"v_pk_mul_f32 v[{HR[b % R] + 2}:{HR[b % R] + 3}], v[{HR[b % R] + 2}:{HR[b % R] + 3}], v[{WR[b % R] + 2}:{WR[b % R] + 3}]",
                f"ds_write_b128 %[q{si}], v[{HR[b % R]}:{HR[b % R] + 3}] offset:{k * 768}"]
    lines = ["s_waitcnt lgkmcnt(0)"]
    queue = []                                     # LDS operations issued so far, in order (they complete in order): ("r", block) / ("w", block)
    def issue(kind, b, text):
        queue.append((kind, b))
        lines.append(text)
    for b in range(R - 1):
        issue("r", b, rdw(b)); issue("r", b, rdh(b))
    for b in range(n):
        if b + R - 1 < n:
            # (slot (b + R - 1) % R was stored from by block b - 1: that ds_write has been issued, and its data left the registers long before
            # a read's data can come back)
            issue("r", b + R - 1, rdw(b + R - 1)); issue("r", b + R - 1, rdh(b + R - 1))
        last = max(i for i, (kind, blk) in enumerate(queue) if kind == "r" and blk == b)
        lines.append(f"s_waitcnt lgkmcnt({len(queue) - 1 - last})")      # everything issued after block b's second read may still be out
        m = body(b)
        lines += m[:2]
        issue("w", b, m[2])
    print("// generated by tools/gen_grub_asm.py --ring-fill %d --first %d -- do not edit" % (S, first))
    print("// operands: %[wp] LDS byte address of the lane's row, block 0; %[hp] LDS byte address of the stream's state, block 0; %[q0] %[q1] %[q2] LDS byte addresses of the lane's row in the ring's three regions")
    clob = [f"v{i}" for i in range(base, 256)]
    print("#undef LPCN_GRUB_RING_FILL_CLOBBERS")
    print("#define LPCN_GRUB_RING_FILL_CLOBBERS " + ", ".join('"%s"' % c for c in clob) + ', "memory"')
    for ln in lines:
        print('"%s\\n\\t"' % ln)


def main_ring_sum(first):
    """The CONSUMER side: the chain wave has summed blocks 0..first-1 itself (--lds 4 --blocks first); the products of the others wait in
    the ring.  Per block ONE ds_read_b128 and the four dependent v_add_f32; all reads of the ten blocks are issued up front (40 VGPRs)."""
    n = 96 - first
    assert sum(RING_SEG) == n
    base = 256 - 4 * n
    RG = [base + 4 * i for i in range(n)]
    seg_of = []
    for si, cnt in enumerate(RING_SEG):
        seg_of += [(si, k) for k in range(cnt)]
    lines = ["s_waitcnt lgkmcnt(0)"]
    for b in range(n):
        si, k = seg_of[b]
        lines += [f"ds_read_b128 v[{RG[b]}:{RG[b] + 3}], %[q{si}] offset:{k * 768}"]
    for b in range(n):
        lines += [f"s_waitcnt lgkmcnt({n - 1 - b})"] + [f"v_add_f32 %[z], %[z], v{RG[b] + j}" for j in range(4)]
    print("// generated by tools/gen_grub_asm.py --ring-sum --first %d -- do not edit" % first)
    print("// operands: %[z] float accumulator (in/out VGPR); %[q0] %[q1] %[q2] LDS byte addresses of the lane's row in the ring's three regions")
    clob = [f"v{i}" for i in range(base, 256)]
    print("#undef LPCN_GRUB_RING_SUM_CLOBBERS")
    print("#define LPCN_GRUB_RING_SUM_CLOBBERS " + ", ".join('"%s"' % c for c in clob) + ', "memory"')
    for ln in lines:
        print('"%s\\n\\t"' % ln)


def main_dpp(S):
    """GRU-B input mat-vec with ONE state read per FOUR blocks: lane k of a quad (four consecutive rows: same stream, same row
    group) fetches block 4q + k of the wave's stream, and the four rows take each block's values through quad broadcasts folded
    into the multiplies (v_mul_f32_dpp quad_perm:[j,j,j,j]).  1.25 LDS instructions per block instead of 2; eight full-rate
    VALU instructions per block (4 DPP multiplies + 4 dependent adds) instead of 2 packed multiplies + 4 adds -- the same issue
    time.  Products of block b+1 are formed in the shadow of the adds of block b.
    %[hp] = LDS address of (stream's state, block = lane & 3); weights as in --lds.
    MEASURED AND REJECTED (round 4): bit-exact at the first attempt, 123.1 vs 126.7 M samples/s at S = 4, 72.7 vs 75.4 M at
    512 streams (S = 2) -- the loop is bound by the issue time of the chain wave's own instructions, and four DPP multiplies
    take more of it than two packed multiplies + the second read (the kernel does not include this variant)."""
    stride = 16 * S
    qstep = 4 * stride + 16                        # bytes from one quad of blocks to the next (16 B pad every 4 blocks)
    base = 256 - 32
    PS = [base, base + 4]
    WR = [base + 8 + 4 * i for i in range(4)]
    HS = [base + 24, base + 28]
    CNT = 70
    BPT = 16
    lines = []
    rdw = lambda blk: f"ds_read_b128 v[{WR[blk % 4]}:{WR[blk % 4] + 3}], %[wp] offset:{blk * 128}"
    rdh = lambda q: f"ds_read_b128 v[{HS[q & 1]}:{HS[q & 1] + 3}], %[hp] offset:{q * qstep}"
    def prod(blk):                                 # products of block `blk` (index inside the trip, may be 16 = next trip's block 0)
        q, j = blk // 4, blk % 4
        return [f"v_mul_f32_dpp v{PS[blk & 1] + c}, v{HS[q & 1] + c}, v{WR[blk % 4] + c} quad_perm:[{j},{j},{j},{j}] row_mask:0xf bank_mask:0xf" for c in range(4)]
    lines += [f"s_mov_b32 s{CNT}, {96 // BPT}", rdh(0), rdw(0), rdw(1), rdw(2), "s_waitcnt lgkmcnt(2)"] + prod(0)
    lines += [".p2align 4"] + ["s_nop 0"] * PHASE + ["1:"]
    for k in range(BPT):
        j = k % 4
        lines += [rdw(k + 3)]
        if j == 0:
            lines += [rdh(k // 4 + 1)]
        lines += [f"s_waitcnt lgkmcnt({2 if j == 3 else 3})"]
        a = [f"v_add_f32 %[z], %[z], v{PS[k & 1] + c}" for c in range(4)]
        m = prod(k + 1)
        lines += [a[0], m[0], a[1], m[1], a[2], m[2], a[3], m[3]]
    # (the trip's last products were formed from ring slot (16 % 4) = 0 and state buffer (16 // 4) & 1 = 0: the next trip's block 0)
    lines += [f"v_add_u32 %[wp], {BPT * 128}, %[wp]",
              f"v_add_u32 %[hp], {4 * qstep}, %[hp]",
              f"s_sub_u32 s{CNT}, s{CNT}, 1",
              f"s_cmp_lg_u32 s{CNT}, 0",
              "s_cbranch_scc1 1b",
              "s_waitcnt lgkmcnt(0)"]
    print("// generated by tools/gen_grub_asm.py --dpp %d -- do not edit" % S)
    print("// operands: %[z] float accumulator (in/out VGPR), %[wp] LDS byte address of the lane's row, block 0 (in/out VGPR), %[hp] LDS byte address of the stream's state, block (lane & 3) (in/out VGPR)")
    clob = [f"s{CNT}"] + [f"v{i}" for i in range(base, 256)]
    print("#undef LPCN_GRUB_DPP_CLOBBERS")
    print("#define LPCN_GRUB_DPP_CLOBBERS " + ", ".join('"%s"' % c for c in clob) + ', "scc", "memory"')
    for ln in lines:
        print('"%s\\n\\t"' % ln)


def main_i8(base):
    """GRU-B input mat-vec of the int8 PARITY kernels, dense input matrix: per block one exact v_dot4_i32_i8 (4 int8 weights x 4
    quantised state values), converted to float and added to the running sum in block order (src/vec.h:306-339).  One 16-byte
    read brings the lane's weights of FOUR blocks, another the stream's quantised state of the same four: a "quad".  Ring of
    --ring R quads (R - 1 in flight), the dots and conversions of quad q+1 in the shadow of the four dependent adds of quad q.
    The compiler's version of this loop waits for each pair of reads right behind their issue (~150 clk exposed per quad).
    gfx90a+ hazard: 3 wait states between a DOT write and another VALU reading it -- the four conversions follow the four dots,
    with the adds in between.   `base` = first clobbered VGPR (a 128-VGPR kernel needs them below v128).
    MEASURED AND NOT KEPT (round 4): bit-exact; GRU-B 3.9 k -> 2.8 k clk per step in the phase table of the int8 kernel with two
    workgroups per CU, throughput unchanged (158.0 vs 158.7 M samples/s; 137.0 vs 137.1 M at S = 4): the second workgroup
    fills the wait either way."""
    R = RING
    WR = [base + 4 * i for i in range(R)]
    XR = [base + 4 * R + 4 * i for i in range(R)]
    D = base + 8 * R
    F = base + 8 * R + 4
    top = base + 8 * R + 8
    CNT = 70
    NQ = 24
    assert NQ % R == 0
    LA = R - 1
    rdw = lambda q: f"ds_read_b128 v[{WR[q % R]}:{WR[q % R] + 3}], %[wp] offset:{q * 128}"
    rdx = lambda q: f"ds_read_b128 v[{XR[q % R]}:{XR[q % R] + 3}], %[xp] offset:{q * 16}"
    dots = lambda q: [f"v_dot4_i32_i8 v{D + k}, v{WR[q % R] + k}, v{XR[q % R] + k}, 0" for k in range(4)]
    cvts = [f"v_cvt_f32_i32 v{F + k}, v{D + k}" for k in range(4)]
    lines = ["s_waitcnt lgkmcnt(0)", f"s_mov_b32 s{CNT}, {NQ // R}"]      # (entry: see --lds)
    for q in range(LA):
        lines += [rdw(q), rdx(q)]
    lines += [f"s_waitcnt lgkmcnt({2 * (LA - 1)})"] + dots(0) + cvts
    lines += [".p2align 4", "1:"]
    for q in range(R):
        a = [f"v_add_f32 %[z], %[z], v{F + k}" for k in range(4)]
        d = dots(q + 1)
        # (the conversions of quad q+1 overwrite the four floats of quad q behind its last add)
        lines += [rdw(q + LA), rdx(q + LA), f"s_waitcnt lgkmcnt({2 * (LA - 1)})",
                  a[0], d[0], a[1], d[1], a[2], d[2], a[3], d[3]] + cvts
    lines += [f"v_add_u32 %[wp], {R * 128}, %[wp]",
              f"v_add_u32 %[xp], {R * 16}, %[xp]",
              f"s_sub_u32 s{CNT}, s{CNT}, 1",
              f"s_cmp_lg_u32 s{CNT}, 0",
              "s_cbranch_scc1 1b",
              "s_waitcnt lgkmcnt(0)"]
    print("// generated by tools/gen_grub_asm.py --i8 %d --ring %d -- do not edit" % (base, R))
    print("// operands: %[z] float accumulator (in/out VGPR), %[wp] LDS byte address of the lane's row, quad 0 (in/out VGPR), %[xp] LDS byte address of the stream's quantised state (in/out VGPR)")
    clob = [f"s{CNT}"] + [f"v{i}" for i in range(base, top)]
    name = "LPCN_GRUB_I8_CLOBBERS_%d" % base
    print("#undef " + name)
    print("#define " + name + " " + ", ".join('"%s"' % c for c in clob) + ', "scc", "memory"')
    for ln in lines:
        print('"%s\\n\\t"' % ln)


if __name__ == "__main__":
    FIRST = int(sys.argv[sys.argv.index("--first") + 1]) if "--first" in sys.argv else 86
    if "--ring-fill" in sys.argv:
        main_ring_fill(int(sys.argv[sys.argv.index("--ring-fill") + 1]), FIRST)
    elif "--ring-sum" in sys.argv:
        main_ring_sum(FIRST)
    elif "--i8" in sys.argv:
        main_i8(int(sys.argv[sys.argv.index("--i8") + 1]))
    elif "--dpp" in sys.argv:
        main_dpp(int(sys.argv[sys.argv.index("--dpp") + 1]))
    elif "--prod" in sys.argv:
        main_prod(int(sys.argv[sys.argv.index("--prod") + 1]))
    elif "--rl" in sys.argv:
        main_rl(int(sys.argv[sys.argv.index("--rl") + 1]))
    elif "--lds4m" in sys.argv:
        main_lds4m(int(sys.argv[sys.argv.index("--lds4m") + 1]))
    elif "--lds2p" in sys.argv:
        main_lds2p(int(sys.argv[sys.argv.index("--lds2p") + 1]))
    elif "--lds2" in sys.argv:
        main_lds2(int(sys.argv[sys.argv.index("--lds2") + 1]))
    elif "--lds" in sys.argv:
        main_lds(int(sys.argv[sys.argv.index("--lds") + 1]))
    else:
        main()
