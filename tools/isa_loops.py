#!/usr/bin/env python3
"""Instruction histogram of the big loops of one kernel in a hipcc -S listing (tools: which instructions a Griffin-Lim
iteration really issues).  usage: isa_loops.py listing.s kernel-substring [min_body]"""
import re
import sys
from collections import Counter


def classify(op):
    if op.startswith("v_pk_"): return "v_pk"
    if op.startswith("v_mfma"): return "mfma"
    if op.startswith("v_readlane") or op.startswith("v_writelane") or op.startswith("v_readfirstlane"): return "v_lane"
    if op.startswith("v_mov") or op.startswith("v_accvgpr"): return "v_mov"
    if op.startswith("v_cndmask"): return "v_cndmask"
    if op.startswith("v_"): return "v_other"
    if op.startswith("ds_bpermute"): return "ds_bperm"
    if op.startswith("ds_"): return "ds"
    if op.startswith("global_") or op.startswith("buffer_") or op.startswith("flat_"): return "vmem"
    if op.startswith("scratch_"): return "scratch"
    if op.startswith("s_waitcnt"): return "s_waitcnt"
    if op.startswith("s_barrier"): return "s_barrier"
    if op.startswith("s_load") or op.startswith("s_buffer"): return "smem"
    if op.startswith("s_nop"): return "s_nop"
    if op.startswith("s_"): return "s_other"
    return "other"


def main():
    path, kern = sys.argv[1], sys.argv[2]
    min_body = int(sys.argv[3]) if len(sys.argv) > 3 else 150
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and kern in l and l.rstrip().split(":")[0].endswith(l.split(":")[0]))
    end = next(i for i in range(start + 1, len(lines)) if lines[i].startswith(".Lfunc_end"))
    body = lines[start:end]
    # blocks
    blocks, cur = [], None
    for l in body:
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            cur = [m.group(1), l, []]
            blocks.append(cur)
            continue
        s = l.strip()
        if not s or s.startswith(";") or s.startswith("."): continue
        if cur is None:
            cur = ["entry", "", []]
            blocks.append(cur)
        cur[2].append(s.split()[0])
    total = Counter()
    for name, hdr, ops in blocks:
        for o in ops: total[classify(o)] += 1
    print("kernel total:", sum(total.values()), dict(total.most_common()))
    # loops: a block whose terminator branches back to itself or an earlier block (single-block loops are what the unrolled code gives)
    idx = {b[0]: i for i, b in enumerate(blocks)}
    text = {b[0]: b for b in blocks}
    for i, (name, hdr, ops) in enumerate(blocks):
        if len(ops) < min_body: continue
        c = Counter(classify(o) for o in ops)
        print(f"{name:12s} n={len(ops):5d} {'LOOP' if 'Loop' in hdr else '    '} " + " ".join(f"{k}={v}" for k, v in c.most_common()))


if __name__ == "__main__":
    main()
