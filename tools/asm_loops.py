"""Per-loop instruction mix of one kernel in a hipcc -S .s file: python tools/asm_loops.py file.s kernel_prefix [min_len]"""
import collections
import re
import sys

lines = open(sys.argv[1]).read().split("\n")
key = sys.argv[2]
minlen = int(sys.argv[3]) if len(sys.argv) > 3 else 40
start = next(i for i, l in enumerate(lines) if l.startswith(key) and ":" in l)
end = next(i for i in range(start, len(lines)) if ".Lfunc_end" in lines[i])
body = lines[start:end]
labels = {}
for i, l in enumerate(body):
    m = re.match(r"(\.LBB\d+_\d+):", l)
    if m:
        labels[m.group(1)] = i


def klass(op):
    if op.startswith("v_") and "f64" in op:
        return "f64"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith("global_") or op.startswith("buffer_") or op.startswith("scratch_") or op.startswith("flat_"):
        return "vmem"
    if op == "s_barrier":
        return "barrier"
    if op.startswith("s_waitcnt") or op == "s_nop":
        return "wait"
    if op.startswith("s_cbranch") or op == "s_branch":
        return "branch"
    if op.startswith("s_"):
        return "salu"
    return "other"


for lab, i in sorted(labels.items(), key=lambda x: x[1]):
    if "Loop Header" not in body[i]:
        continue
    last = max((j for j, l in enumerate(body) if re.search(r"s_c?branch\S*\s+" + re.escape(lab) + r"\b", l)), default=-1)
    if last - i < minlen:
        continue
    c = collections.Counter()
    ops = collections.Counter()
    for l in body[i:last + 1]:
        m = re.match(r"\t([a-z_0-9]+)", l)
        if m:
            c[klass(m.group(1))] += 1
            ops[m.group(1)] += 1
    depth = re.search(r"Depth=(\d+)", body[i]).group(1)
    print("%s @%d..%d depth %s  total %d  %s" % (lab, i, last, depth, sum(c.values()), dict(c)))
    if len(sys.argv) > 4:
        print("    ", sorted(ops.items(), key=lambda x: -x[1])[:30])
