"""Instruction mix between the s_barrier marks of one kernel in a hipcc -S file:
   python tools/asm_classes.py file.s kernel_prefix
classes: mfma, f64 (other fp64 VALU), valu (everything else vector: moves, selects, integer), lds, vmem, scratch, salu, branch, wait."""
import collections
import re
import sys


def klass(op):
    if op.startswith("v_mfma"):
        return "mfma"
    if op.startswith("scratch_"):
        return "scratch"
    if op.startswith("v_") and "f64" in op:
        return "f64"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_")):
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


def regions(path, key):
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith(key) and ":" in l)
    end = next(i for i in range(start, len(lines)) if ".Lfunc_end" in lines[i])
    body = lines[start:end]
    out, cur, first = [], collections.Counter(), 0
    for i, l in enumerate(body):
        m = re.match(r"\t([a-z_0-9]+)", l)
        if not m:
            continue
        op = m.group(1)
        if op == "s_barrier":
            out.append((first, i, cur))
            cur, first = collections.Counter(), i
        else:
            cur[klass(op)] += 1
    out.append((first, len(body), cur))
    return out


if __name__ == "__main__":
    for a, b, c in regions(sys.argv[1], sys.argv[2]):
        tot = sum(c.values())
        if tot < 30:
            continue
        print("%6d-%6d total %5d  " % (a, b, tot) + "  ".join("%s %d" % (k, c[k]) for k in ("mfma", "f64", "valu", "lds", "vmem", "scratch", "salu", "branch", "wait") if c[k]))
