#!/usr/bin/env python3
"""Instruction mix of one kernel in a hipcc -save-temps .s file, per basic block.

usage: isa_stats.py file.s kernel_name_substring [--blocks] [--dump LABEL]

Classes: mfma, valu (incl. v_cmp / v_cndmask), trans (v_exp/v_log/v_rcp/v_rsq/v_sqrt), vmem, lds, salu, smem, wait, branch.
The per-block table is the tool used for DESIGN.md's "VALU instructions per score element" figures: find the loop
body (the block that branches back to itself or to an earlier label) and read its row.
"""
import re
import sys
from collections import Counter, OrderedDict


def classify(op):
    if op.startswith("v_mfma") or op.startswith("v_smfma"):
        return "mfma"
    if op in ("v_exp_f32", "v_log_f32", "v_rcp_f32", "v_rsq_f32", "v_sqrt_f32", "v_sin_f32", "v_cos_f32", "v_rcp_iflag_f32"):
        return "trans"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("s_waitcnt") or op.startswith("s_nop") or op.startswith("s_barrier") or op.startswith("s_sleep"):
        return "wait"
    if op.startswith("s_cbranch") or op.startswith("s_branch") or op.startswith("s_endpgm") or op.startswith("s_setpc"):
        return "branch"
    if op.startswith(("s_load", "s_buffer_load", "s_store", "s_memtime", "s_dcache")):
        return "smem"
    if op.startswith("s_"):
        return "salu"
    return "other"


def main():
    path, name = sys.argv[1], sys.argv[2]
    show_blocks = "--blocks" in sys.argv
    dump = sys.argv[sys.argv.index("--dump") + 1] if "--dump" in sys.argv else None
    lines = open(path).read().split("\n")
    start = None
    for i, l in enumerate(lines):
        if re.match(r"^[A-Za-z_][\w$.]*:", l) and name in l.split(":")[0]:
            start = i
            break
    if start is None:
        sys.exit("kernel not found")
    blocks = OrderedDict()
    cur = "entry"
    blocks[cur] = []
    for l in lines[start + 1:]:
        s = l.strip()
        if s.startswith(".Lfunc_end") or s.startswith(".section") or s.startswith(".rodata"):
            break
        if re.match(r"^\.LBB\d+_\d+:", s):
            cur = s.split(":")[0]
            blocks[cur] = []
            continue
        if not s or s.startswith((";", ".", "//")):
            continue
        op = s.split()[0]
        blocks[cur].append((op, s))
    total = Counter()
    for b, ins in blocks.items():
        for op, _ in ins:
            total[classify(op)] += 1
    print("kernel:", lines[start][:-1][:100])
    print("static total:", dict(total))
    if dump:
        for op, s in blocks[dump]:
            print("   ", s)
        return
    if show_blocks:
        print(f"{'block':<14}{'n':>6}{'mfma':>6}{'valu':>6}{'trans':>6}{'lds':>5}{'vmem':>5}{'salu':>6}{'wait':>5}  branch-to")
        for b, ins in blocks.items():
            c = Counter(classify(op) for op, _ in ins)
            tgt = [s.split()[-1] for op, s in ins if op.startswith(("s_cbranch", "s_branch"))]
            if len(ins) >= 8:
                print(f"{b:<14}{len(ins):>6}{c['mfma']:>6}{c['valu']:>6}{c['trans']:>6}{c['lds']:>5}{c['vmem']:>5}{c['salu']:>6}{c['wait']:>5}  {','.join(tgt)}")
    ops = Counter()
    for b, ins in blocks.items():
        for op, _ in ins:
            ops[op] += 1
    print("top ops:", ops.most_common(40))


if __name__ == "__main__":
    main()
