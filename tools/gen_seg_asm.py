#!/usr/bin/env python3
"""Generates deeprecsys_amd/csrc/seg_asm.inc: the instruction streams of stream4_kernel (mlp_stream4.hip).

One asm statement runs a whole SEGMENT -- every 64-k chunk of one (layer, pass) for the T = 4 / 2 / 1
column tiles a wave owns -- as an unbroken run of MFMAs with the weight reloads, the operand prefetch
and the loop control placed between them.  Everything it touches lives in accumulation registers
under fixed names (the kernel's C++ never uses AGPRs; the build checks that):

    a[0:15]     accumulators, tile j = a[4j:4j+3]
    a[16:79]    S0: weight slot 0     tile j, k-group q = base + 16 j + 4 q
    a[80:143]   S1: weight slot 1
    a[144:159]  A0: activation operands that go with slot 0 (k-group q = base + 4 q)
    a[160:175]  A1: ... with slot 1
(32 rows per workgroup, SEG2_*: accumulators a[0:31] -- tile j of half h = a[4 (j + T h) ...] --, slots at
a[32:95] / a[96:159], operand sets of slot p, half h at a[160 + 32 p + 16 h ...]: 224 registers.)
Chunk c of a segment that starts in slot p uses slot (p + c) % 2; while the LAST chunk runs, chunk 0 of
the NEXT segment is requested into the other slot, which that segment then starts in (`par`).

Operands of a segment statement (T tiles): %0..%(T-1) R_j (in/out: byte offset from the weight arena
of tile j's 4-KB block in the chunk the next reload fetches, + 16 lane), then a_addr (in/out: this
lane's LDS byte address of the next operand read), rem (in/out: chunks left), N_0..N_3 (offsets of the
next segment's chunk 0, all four tiles), wbase (64-bit scalar), par (scalar: the slot chunk 0 sits in).

    python tools/gen_seg_asm.py > deeprecsys_amd/csrc/seg_asm.inc
"""

# register maps by rows per workgroup: R = 1 (16 rows) and R = 2 (32 rows: two 16-row halves that share
# every weight operand -- twice the MFMAs per byte of weights)
MAPS = {
    1: dict(SB=(16, 80), AB=((144,), (160,)), NREG=176),
    2: dict(SB=(32, 96), AB=((160, 176), (192, 208)), NREG=224),
}
CHUNK_BYTES = 32768


def areg(b, n=4):
    return "a[%d:%d]" % (b, b + n - 1)


def gen(T, R=1):
    SB, AB = MAPS[R]["SB"], MAPS[R]["AB"]
    oR = list(range(T))                 # %0..%T-1
    oAs = [T + h for h in range(R)]     # a_addr of each 16-row half
    oREM = T + R
    oN = [T + R + 1 + j for j in range(4)]
    oW, oPAR = T + R + 5, T + R + 6
    L = []
    e = L.append

    def S(base, j, q):
        return base + 16 * j + 4 * q

    def loads(base, offs, tiles, q_list=(0, 1, 2, 3)):
        for q in q_list:
            for j in range(tiles):
                e("global_load_dwordx4 %s, %%%d, %%%d offset:%d" % (areg(S(base, j, q)), offs[j], oW, 1024 * q))

    def read_a(bases):
        for h in range(R):
            for q in range(4):
                e("ds_read_b128 %s, %%%d offset:%d" % (areg(bases[h] + 4 * q), oAs[h], 64 * q))
        for h in range(R):
            e("v_add_u32 %%%d, 256, %%%d" % (oAs[h], oAs[h]))

    def mfmas(sb, ab, q):
        for s in range(4):
            for h in range(R):
                for j in range(T):
                    acc = areg(4 * (j + T * h))
                    e("v_mfma_f32_16x16x4_f32 %s, a%d, a%d, %s" % (acc, ab[h] + 4 * q + s, S(sb, j, q) + s, acc))

    def body(kind, sb, ab, sbo, abo):
        # kind: steady | penult_w | penult_nw | last_w | last_nw;  sb / ab: this chunk's slot and operand
        # set, sbo / abo: the other ones
        if kind == "last_w":
            e("s_waitcnt vmcnt(0)")
        if kind == "penult_w":
            e("s_waitcnt vmcnt(%d)" % (4 * T))
        e("s_waitcnt lgkmcnt(0)")
        if kind.startswith("last"):
            loads(sbo, oN, 4)                      # the next segment's chunk 0 (all four tiles) -> the free slot
        else:
            read_a(abo)
        for q in range(4):
            if kind == "steady":
                e("s_waitcnt vmcnt(%d)" % (7 * T))
            mfmas(sb, ab, q)
            if kind == "steady":
                loads(sb, oR, T, (q,))
        if kind == "steady":
            for j in range(T):
                e("v_add_u32 %%%d, 0x%x, %%%d" % (oR[j], CHUNK_BYTES, oR[j]))
            e("s_sub_u32 %%%d, %%%d, 1" % (oREM, oREM))

    def variant(p):
        X, Y = (SB[p], AB[p]), (SB[1 - p], AB[1 - p])     # chunk 0 sits in slot p
        t = "P%d" % p
        read_a(X[1])
        for i in range(4 * T * R):
            e("v_accvgpr_write_b32 a%d, 0" % i)
        e("s_cmp_eq_u32 %%%d, 1" % oREM)
        e("s_cbranch_scc1 LE1%s_%%=" % t)
        loads(Y[0], oR, T)
        for j in range(T):
            e("v_add_u32 %%%d, 0x%x, %%%d" % (oR[j], CHUNK_BYTES, oR[j]))
        e("s_cmp_eq_u32 %%%d, 2" % oREM)
        e("s_cbranch_scc1 LE2%s_%%=" % t)
        e("LLOOP%s_%%=:" % t)
        body("steady", X[0], X[1], Y[0], Y[1])
        e("s_cmp_le_u32 %%%d, 2" % oREM)
        e("s_cbranch_scc1 LTODD%s_%%=" % t)
        body("steady", Y[0], Y[1], X[0], X[1])
        e("s_cmp_gt_u32 %%%d, 2" % oREM)
        e("s_cbranch_scc1 LLOOP%s_%%=" % t)
        body("penult_w", X[0], X[1], Y[0], Y[1])
        body("last_w", Y[0], Y[1], X[0], X[1])
        e("s_branch LDONE_%=")
        e("LTODD%s_%%=:" % t)
        body("penult_w", Y[0], Y[1], X[0], X[1])
        body("last_w", X[0], X[1], Y[0], Y[1])
        e("s_branch LDONE_%=")
        e("LE2%s_%%=:" % t)
        body("penult_nw", X[0], X[1], Y[0], Y[1])
        body("last_w", Y[0], Y[1], X[0], X[1])
        e("s_branch LDONE_%=")
        e("LE1%s_%%=:" % t)
        body("last_nw", X[0], X[1], Y[0], Y[1])

    # ---- entry: chunk 0 of this segment was requested into slot `par` by the segment before
    e("s_waitcnt vmcnt(0)")
    e("s_cmp_eq_u32 %%%d, 0" % oPAR)
    e("s_cbranch_scc0 LPAR1_%=")
    variant(0)
    e("s_branch LDONE_%=")
    e("LPAR1_%=:")
    variant(1)
    e("LDONE_%=:")
    e("s_nop 15")
    e("s_nop 7")
    return L


def emit(name, lines):
    print("#define %s \\" % name)
    for i, l in enumerate(lines):
        print('  "%s\\n\\t"%s' % (l, " \\" if i + 1 < len(lines) else ""))
    print()


def main():
    print("// GENERATED by tools/gen_seg_asm.py -- do not edit.  See that file for the register map.")
    for R in (1, 2):
        tag = "SEG" if R == 1 else "SEG2"
        for T in (4, 2, 1):
            emit("%s_ASM_T%d" % (tag, T), gen(T, R))
        # stand-alone request of a segment's chunk 0 into slot 0 / 1 (kernel prologue; waves that sit a
        # segment out): %0..%3 offsets of the four tiles, %4 wbase
        for p in (0, 1):
            pf = []
            for q in range(4):
                for j in range(4):
                    pf.append("global_load_dwordx4 %s, %%%d, %%4 offset:%d" % (areg(MAPS[R]["SB"][p] + 16 * j + 4 * q), j, 1024 * q))
            emit("%s_PREFETCH%d_ASM" % (tag, p), pf)
        regs = ", ".join('"a%d"' % i for i in range(MAPS[R]["NREG"]))
        print("#define %s_AGPR_CLOBBER %s" % (tag, regs))


if __name__ == "__main__":
    main()
