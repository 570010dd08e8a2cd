#!/usr/bin/env python3
"""Generates ctc_amd/csrc/noblank_km_asm.hpp: the four hand-scheduled chain loops of noblank_km.hpp
(K = exponent wave, M = mantissa wave; forward and backward) as inline-asm strings.

    python tools/gen_km_asm.py > ctc_amd/csrc/noblank_km_asm.hpp

A loop iteration is a DOUBLE block: 2 x 8 steps (4 pairs of cells each) on two register sets -- while one
set is worked on, the 16-byte LDS reads of the next block land in the other.  Buffers and temporaries are
explicit physical VGPRs (v48..v127, listed as clobbers): an asm statement that issues its own LDS reads
must not hand their destinations to the compiler (it would be free to copy them before the data is there).
Every DPP read is >= 2 instructions behind the VALU write of its source (no s_nop inside the loops).
"""

DPP = {True: "wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1", False: "wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1"}


def regs4(base):
    return "v[%d:%d]" % (base, base + 3)


def off(fwd, q, unit):          # byte offset of pair q (processing order) inside a block
    return unit * (q if fwd else 3 - q)


def gen_m(fwd):
    """operands: %[ka] %[ea] %[oa] addresses (kc cells, em cells, mantissas) of the block's lowest pair,
    %[m] %[kp] chain state, %[pa] address of the progress word, %[pc] progress value, %[nb] double blocks (SGPR)"""
    dpp = DPP[fwd]
    step = 64 if fwd else -64
    mstep = 32 if fwd else -32
    KC = {"X": [64, 68, 72, 76], "Y": [96, 100, 104, 108]}
    EE = {"X": [80, 84, 88, 92], "Y": [112, 116, 120, 124]}
    D = [[48, 49, 50, 51], [52, 53, 54, 55]]
    MP = [56, 58, 60, 62]                                     # (lower row, higher row) mantissa pairs
    out = []
    A = out.append

    def first(s, q):      # (k, c, pm) registers of the pair's first / second step
        kc, ee = KC[s][q], EE[s][q]
        lo = (kc, kc + 1, ee)
        hi = (kc + 2, kc + 3, ee + 2)
        return (lo, hi) if fwd else (hi, lo)

    def mregs(q):         # mantissa registers of the first / second step of pair q
        return (MP[q], MP[q] + 1) if fwd else (MP[q] + 1, MP[q])

    def E(s, q, kp, d):
        (k1, c1, p1), (k2, c2, p2) = first(s, q)
        return ["v_sub_u32 v%d, %s, v%d" % (d[0], kp, c1),
                "v_sub_u32_dpp v%d, %s, v%d %s" % (d[1], kp, c1, dpp),
                "v_ldexp_f32 v%d, v%d, v%d" % (d[0], p1, d[0]),
                "v_ldexp_f32 v%d, v%d, v%d" % (d[1], p1, d[1]),
                "v_sub_u32 v%d, v%d, v%d" % (d[2], k1, c2),
                "v_sub_u32_dpp v%d, v%d, v%d %s" % (d[3], k1, c2, dpp),
                "v_ldexp_f32 v%d, v%d, v%d" % (d[2], p2, d[2]),
                "v_ldexp_f32 v%d, v%d, v%d" % (d[3], p2, d[3])]

    def M(q, min_, d):
        ma, mb = mregs(q)
        return ["v_mul_f32 v%d, v%d, v%d" % (ma, min_, d[0]),
                "v_fmac_f32_dpp v%d, v%d, v%d %s" % (ma, min_, d[1], dpp),
                "v_mul_f32 v%d, v%d, v%d" % (mb, ma, d[2]),
                "v_fmac_f32_dpp v%d, v%d, v%d %s" % (mb, ma, d[3], dpp)]

    def kp_of(s, q):      # k of the second step of pair q
        return "v%d" % first(s, q)[1][0]

    def loads(s, masked=True):
        # LDS instructions run on the live lanes only (two thirds of a chain wave's lanes are idle, and the chains'
        # LDS traffic competes with the workers').  The idle lanes keep what the unmasked loads of the prologue
        # gave them: the cells of the spare row, which never change.
        if masked:
            A("s_mov_b64 exec, %[mk]")
        for q in range(4):
            A("ds_read_b128 %s, %%[ka] offset:%d" % (regs4(KC[s][q]), off(fwd, q, 16)))
            A("ds_read_b128 %s, %%[ea] offset:%d" % (regs4(EE[s][q]), off(fwd, q, 16)))
        if masked:
            A("s_mov_b64 exec, -1")

    def block(cur, nxt):
        # the next block's cells into the other register set
        A("v_add_u32 %%[ka], %d, %%[ka]" % step)
        A("v_add_u32 %%[ea], %d, %%[ea]" % step)
        loads(nxt)
        for q in range(4):
            min_ = mregs(3)[1] if q == 0 else mregs(q - 1)[1]
            mm = M(q, min_, D[q & 1])
            if q < 3:
                ee = E(cur, q + 1, kp_of(cur, q), D[(q + 1) & 1])
            else:
                A("s_waitcnt lgkmcnt(0)")                     # the next block's reads
                ee = E(nxt, 0, kp_of(cur, 3), D[0])
            order = [mm[0], ee[0], ee[1], mm[1], ee[2], ee[3], mm[2], ee[4], ee[5], mm[3], ee[6], ee[7]]
            for ins in order:
                A(ins)
        A("v_add_u32 %[pc], 8, %[pc]")
        A("s_mov_b64 exec, %[mk]")
        for q in range(4):
            A("ds_write_b64 %%[oa], v[%d:%d] offset:%d" % (MP[q], MP[q] + 1, off(fwd, q, 8)))
        A("ds_write_b32 %[pa], %[pc]")
        A("s_mov_b64 exec, -1")
        A("v_add_u32 %%[oa], %d, %%[oa]" % mstep)

    A("v_mov_b32 v%d, %%[m]" % mregs(3)[1])
    loads("Y", masked=False)
    loads("X", masked=False)
    A("s_waitcnt lgkmcnt(0)")
    for ins in E("X", 0, "%[kp]", D[0]):
        A(ins)
    A("Lkm_m_%s_%%=:" % ("f" if fwd else "b"))
    block("X", "Y")
    block("Y", "X")
    A("s_sub_u32 %[nb], %[nb], 1")
    A("s_cmp_lg_u32 %[nb], 0")
    A("s_cbranch_scc1 Lkm_m_%s_%%=" % ("f" if fwd else "b"))
    A("s_waitcnt lgkmcnt(0)")
    A("v_mov_b32 %%[m], v%d" % mregs(3)[1])
    A("v_mov_b32 %%[kp], %s" % kp_of("Y", 3))
    return out


def gen_k(fwd):
    """operands: %[ea] %[oa] addresses (em cells read, kc cells written) of the block's lowest pair, %[kf] chain state,
    %[pa] %[pc] progress word / value, %[nb] double blocks (SGPR)"""
    dpp = DPP[fwd]
    step = 64 if fwd else -64
    EE = {"X": [64, 68, 72, 76], "Y": [80, 84, 88, 92]}
    OUT = [96, 100, 104, 108]
    T, KA, KB, P1, P2 = 48, 49, 50, 51, 52
    out = []
    A = out.append

    def loads(s, masked=True):
        if masked:
            A("s_mov_b64 exec, %[mk]")
        for q in range(4):
            A("ds_read_b128 %s, %%[ea] offset:%d" % (regs4(EE[s][q]), off(fwd, q, 16)))
        if masked:
            A("s_mov_b64 exec, -1")

    def pair(s, q, kin):
        ee, o = EE[s][q], OUT[q]
        e1, e2 = (ee + 1, ee + 3) if fwd else (ee + 3, ee + 1)
        (k1, c1), (k2, c2) = ((o, o + 1), (o + 2, o + 3)) if fwd else ((o + 2, o + 3), (o, o + 1))
        A("v_max_f32_dpp v%d, %s, %s %s" % (T, kin, kin, dpp))
        A("v_cvt_flr_i32_f32 v%d, v%d" % (P1, e1))
        A("v_add_f32 v%d, v%d, v%d" % (KA, T, e1))
        A("v_cvt_flr_i32_f32 v%d, v%d" % (P2, e2))
        A("v_cvt_flr_i32_f32 v%d, v%d" % (k1, KA))
        A("v_sub_u32 v%d, v%d, v%d" % (c1, k1, P1))
        A("v_max_f32_dpp v%d, v%d, v%d %s" % (T, KA, KA, dpp))
        A("v_add_f32 v%d, v%d, v%d" % (KB, T, e2))
        A("v_cvt_flr_i32_f32 v%d, v%d" % (k2, KB))
        A("v_sub_u32 v%d, v%d, v%d" % (c2, k2, P2))

    def block(cur, nxt):
        A("v_add_u32 %%[ea], %d, %%[ea]" % step)
        loads(nxt)
        A("s_waitcnt lgkmcnt(4)")                             # this block's cells (the next block's four reads behind them)
        for q in range(4):
            pair(cur, q, "v%d" % KB)
        A("v_add_u32 %[pc], 8, %[pc]")
        A("s_mov_b64 exec, %[mk]")
        for q in range(4):
            A("ds_write_b128 %%[oa], %s offset:%d" % (regs4(OUT[q]), off(fwd, q, 16)))
        A("ds_write_b32 %[pa], %[pc]")
        A("s_mov_b64 exec, -1")
        A("v_add_u32 %%[oa], %d, %%[oa]" % step)

    A("v_mov_b32 v%d, %%[kf]" % KB)
    loads("Y", masked=False)
    loads("X", masked=False)
    A("s_nop 1")
    A("Lkm_k_%s_%%=:" % ("f" if fwd else "b"))
    block("X", "Y")
    block("Y", "X")
    A("s_sub_u32 %[nb], %[nb], 1")
    A("s_cmp_lg_u32 %[nb], 0")
    A("s_cbranch_scc1 Lkm_k_%s_%%=" % ("f" if fwd else "b"))
    A("s_waitcnt lgkmcnt(0)")
    A("v_mov_b32 %%[kf], v%d" % KB)
    return out


def emit(name, lines):
    print("#define %s \\" % name)
    for i, ins in enumerate(lines):
        print('    "%s\\n\\t"%s' % (ins, " \\" if i + 1 < len(lines) else ""))
    print()


print("// GENERATED by tools/gen_km_asm.py -- do not edit; the loops are described there and in noblank_km.hpp.")
print("#pragma once")
print()
emit("CTC_KM_M_FWD", gen_m(True))
emit("CTC_KM_M_BWD", gen_m(False))
emit("CTC_KM_K_FWD", gen_k(True))
emit("CTC_KM_K_BWD", gen_k(False))
print('#define CTC_KM_CLOBBERS \\')
print("    " + ", ".join('"v%d"' % r for r in range(48, 128)) + ', "scc", "memory"')
