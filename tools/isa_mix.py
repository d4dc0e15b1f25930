#!/usr/bin/env python3
"""Static instruction mix of one kernel in a hipcc -S listing: per basic block and for the largest
loop (the RK4 hot loop).  usage: isa_mix.py file.s <substring of the mangled kernel name> [--blocks]"""
import re
import sys
from collections import Counter


def classify(m):
    if m.startswith("v_accvgpr"):
        return "accvgpr"
    if m.startswith(("v_fma_f64", "v_mul_f64", "v_add_f64", "v_fma_f32", "v_mul_f32", "v_add_f32", "v_sub_f32",
                     "v_fmac", "v_mac", "v_pk_")):
        return "fp_arith"
    if m.startswith(("v_rcp", "v_rsq", "v_sqrt", "v_exp", "v_log", "v_sin", "v_cos")):
        return "trans"
    if m.startswith(("v_cndmask",)):
        return "cndmask"
    if m.startswith(("v_cmp", "v_cmpx")):
        return "vcmp"
    if m.startswith(("v_mov", "v_readlane", "v_writelane", "v_readfirstlane")):
        return "vmov"
    if m.startswith(("v_max", "v_min", "v_med3", "v_bfi", "v_and", "v_or", "v_xor", "v_ldexp", "v_frexp",
                     "v_rndne", "v_cvt", "v_trunc", "v_floor", "v_fract", "v_div")):
        return "v_other_fp"
    if m.startswith("v_"):
        return "v_int_other"
    if m.startswith(("s_waitcnt", "s_nop")):
        return "wait"
    if m.startswith(("s_cbranch", "s_branch")):
        return "branch"
    if m.startswith(("s_load", "s_buffer_load")):
        return "smem"
    if m.startswith("s_"):
        return "salu"
    if m.startswith("ds_"):
        return "lds"
    if m.startswith(("global_", "flat_", "buffer_", "scratch_")):
        return "vmem"
    return "other"


def main():
    path, key = sys.argv[1], sys.argv[2]
    show_blocks = "--blocks" in sys.argv
    lines = open(path).read().split("\n")
    start = None
    for i, l in enumerate(lines):
        if re.match(r"^[A-Za-z_]\S*:", l) and key in l.split(":")[0]:
            start = i
            break
    assert start is not None, "kernel not found"
    blocks, cur, order = {}, "entry", ["entry"]
    blocks[cur] = []
    for l in lines[start + 1:]:
        s = l.strip()
        if s.startswith(".Lfunc_end"):
            break
        m = re.match(r"^(\.LBB\d+_\d+):", s)
        if m:
            cur = m.group(1)
            blocks[cur] = []
            order.append(cur)
            continue
        if not s or s.startswith((";", ".")):
            continue
        blocks[cur].append(s.split(";")[0].strip())
    pos = {b: i for i, b in enumerate(order)}
    loops = []
    for b in order:
        for ins in blocks[b]:
            mm = re.match(r"^s_c?branch\S*\s+(\.LBB\d+_\d+)", ins)
            if mm and pos[mm.group(1)] <= pos[b]:
                loops.append((pos[mm.group(1)], pos[b]))
    total = Counter()
    for b in order:
        for ins in blocks[b]:
            total[classify(ins.split()[0])] += 1
    print("whole kernel:", sum(total.values()), dict(total.most_common()))
    loops.sort(key=lambda ab: -(sum(len(blocks[order[i]]) for i in range(ab[0], ab[1] + 1))))
    for a, b in loops[:4]:
        c = Counter()
        for i in range(a, b + 1):
            for ins in blocks[order[i]]:
                c[classify(ins.split()[0])] += 1
        print("loop %s..%s: %d blocks, %d instr" % (order[a], order[b], b - a + 1, sum(c.values())), dict(c.most_common()))
    if show_blocks and loops:
        a, b = loops[0]
        for i in range(a, b + 1):
            bl = blocks[order[i]]
            c = Counter(classify(x.split()[0]) for x in bl)
            print("  %-12s %5d  %s" % (order[i], len(bl), dict(c.most_common(6))))


if __name__ == "__main__":
    main()
