#!/usr/bin/env python3
"""Static check of the instruction streams tools/gen_seg_asm.py generates for stream4_kernel.

Interprets every generated statement (T = 4 / 2 / 1 tiles per wave, R = 1 / 2 row halves) for segments
of 1 .. 7 chunks entered in either ring slot, with a model of the hardware's counters -- vector-memory
loads and LDS reads retire IN ORDER, `s_waitcnt vmcnt(n)` / `lgkmcnt(n)` block until at most n are
outstanding -- and verifies that

  * every MFMA reads operands whose loads have retired and that hold exactly the data the k-ordered
    chain wants: accumulator (tile j, half h) sees (chunk c, k-group q, step s) in increasing order with
    the activations of (c, h, q, s) and the weights of (c, j, q, s);
  * the statement issues 16 T R MFMAs per chunk, requests the NEXT segment's chunk 0 (all four tiles)
    exactly once, into the slot its last chunk leaves free, and leaves nothing else in flight.

Exit status 0 when all cases pass.  (Run by tests/test_isa.py.)
"""
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gen_seg_asm as G


def regs(tok):
    m = re.match(r"a\[(\d+):(\d+)\]", tok)
    if m:
        return list(range(int(m.group(1)), int(m.group(2)) + 1))
    return [int(tok[1:])]


def run(T, R, nch, par):
    prog = G.gen(T, R)
    SB, AB = G.MAPS[R]["SB"], G.MAPS[R]["AB"]
    oR = list(range(T))
    oAs = [T + h for h in range(R)]
    oREM = T + R
    oN = [T + R + 1 + j for j in range(4)]
    oPAR = T + R + 6
    labels = {l[:-1].replace("%=", ""): i for i, l in enumerate(prog) if l.endswith(":")}
    reg = {}                 # AGPR -> tag of its retired content
    vm, lds = [], []         # in-flight (regs, tags), oldest first
    # chunk 0 of this segment: requested by the segment before (all four tiles), still in flight
    for q in range(4):
        for j in range(4):
            base = SB[par] + 16 * j + 4 * q
            vm.append(([base + i for i in range(4)], [("W", 0, j, q, i) for i in range(4)]))
    r_chunk = [1] * T        # chunk the reload offsets point at
    a_chunk = [0] * R        # chunk the next operand read fetches
    rem = nch
    scc = False
    acc_next = {}            # (j, h) -> next expected (c, q, s)
    n_mfma = 0
    next_req = []            # registers the next segment's chunk 0 went to
    pc = 0
    steps = 0

    def retire(queue, upto):
        while len(queue) > upto:
            rs, tags = queue.pop(0)
            for r_, t_ in zip(rs, tags):
                reg[r_] = t_

    def pending(queue, r_):
        return any(r_ in rs for rs, _ in queue)

    while pc < len(prog):
        steps += 1
        assert steps < 200000, "runaway"
        ins = prog[pc]
        pc += 1
        if ins.endswith(":"):
            continue
        op, _, rest = ins.partition(" ")
        args = [a.strip() for a in rest.split(",")] if rest else []
        if op == "s_waitcnt":
            m = re.match(r"(vmcnt|lgkmcnt)\((\d+)\)", rest)
            retire(vm if m.group(1) == "vmcnt" else lds, int(m.group(2)))
        elif op == "global_load_dwordx4":
            dst = regs(args[0])
            src = int(args[1][1:])
            q = int(re.search(r"offset:(\d+)", ins).group(1)) // 1024
            if src in oN:
                j = oN.index(src)
                tags = [("N", j, q, i) for i in range(4)]
                next_req.extend(dst)
            else:
                j = oR.index(src)
                tags = [("W", r_chunk[j], j, q, i) for i in range(4)]
            # (a reload that lands on data still to be consumed shows up as a wrong tag at the MFMA)
            vm.append((dst, tags))
        elif op == "ds_read_b128":
            dst = regs(args[0])
            h = oAs.index(int(args[1].split()[0][1:]))
            q = int(re.search(r"offset:(\d+)", ins).group(1)) // 64
            lds.append((dst, [("A", a_chunk[h], h, q, i) for i in range(4)]))
        elif op == "v_add_u32":
            o = int(args[0][1:])
            if o in oR:
                assert args[1] == "0x%x" % G.CHUNK_BYTES
                r_chunk[oR.index(o)] += 1
            else:
                assert args[1] == "256"
                a_chunk[oAs.index(o)] += 1
        elif op == "v_accvgpr_write_b32":
            reg[int(args[0][1:])] = ("Z",)
        elif op == "v_mfma_f32_16x16x4_f32":
            acc = regs(args[0])
            ra, rb = int(args[1][1:]), int(args[2][1:])
            assert not pending(lds, ra), "MFMA reads a%d while its LDS read is in flight (pc %d)" % (ra, pc)
            assert not pending(vm, rb), "MFMA reads a%d while its load is in flight (pc %d)" % (rb, pc)
            ta, tb = reg.get(ra), reg.get(rb)
            assert ta and ta[0] == "A" and tb and tb[0] == "W", (ta, tb, pc)
            j_h = acc[0] // 4
            h, j = divmod(j_h, T)
            want = acc_next.get((j, h), (0, 0, 0))
            c, q, s_ = want
            assert ta == ("A", c, h, q, s_), "acc(%d,%d): operand %r, wanted chunk %d q %d s %d (pc %d)" % (j, h, ta, c, q, s_, pc)
            assert tb == ("W", c, j, q, s_), "acc(%d,%d): weights %r, wanted chunk %d q %d s %d (pc %d)" % (j, h, tb, c, q, s_, pc)
            s_ += 1
            if s_ == 4:
                s_, q = 0, q + 1
            if q == 4:
                q, c = 0, c + 1
            acc_next[(j, h)] = (c, q, s_)
            n_mfma += 1
        elif op in ("s_cmp_eq_u32", "s_cmp_le_u32", "s_cmp_gt_u32", "s_cmp_lt_u32"):
            o = int(args[0][1:])
            v = rem if o == oREM else par
            k = int(args[1])
            scc = {"eq": v == k, "le": v <= k, "gt": v > k, "lt": v < k}[op.split("_")[2]]
        elif op == "s_sub_u32":
            rem -= 1
        elif op in ("s_cbranch_scc1", "s_cbranch_scc0"):
            if scc == (op[-1] == "1"):
                pc = labels[args[0].replace("%=", "")]
        elif op == "s_branch":
            pc = labels[args[0].replace("%=", "")]
        elif op == "s_nop":
            pass
        else:
            raise AssertionError("unknown instruction " + ins)
    assert n_mfma == 16 * T * R * nch, (n_mfma, T, R, nch)
    for (j, h), nx in acc_next.items():
        assert nx == (nch, 0, 0), ((j, h), nx)
    free = SB[(par + nch) & 1]
    assert sorted(next_req) == list(range(free, free + 64)), "next segment's chunk 0 went to %r" % sorted(next_req)[:4]
    left = [t for _, tags in vm for t in tags if t[0] != "N"]
    assert not left and not lds, "left in flight: %r" % left[:4]
    return n_mfma


def main():
    n = 0
    for R in (1, 2):
        for T in (4, 2, 1):
            for nch in range(1, 8):
                for par in (0, 1):
                    n += run(T, R, nch, par)
    print("seg_asm: %d cases, %d MFMAs checked" % (2 * 3 * 7 * 2, n))


if __name__ == "__main__":
    main()
