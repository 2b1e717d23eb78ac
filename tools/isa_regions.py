"""Per-region instruction accounting of the fused step kernel's update loops.

   hipcc ... -DVIEKF_INST_GROUP=3 -DVIEKF_ISA_MARKS -S --cuda-device-only -o marks.s viekf_inst.hip     (RES_MARK fences)
   hipcc ... -DVIEKF_INST_GROUP=3                   -S --cuda-device-only -o plain.s viekf_inst.hip     (the shipped code)
   python tools/isa_regions.py marks.s plain.s 'k_step_residentILi7ELi3ELb0ELi1ELb1E' out.json

Counts are STATIC wave-instructions of the listing between two marks (the marks are scheduling fences, so nothing crosses them);
`trips` says how often a region's inner loop runs per update where it has one.  The plain listing gives the totals of the same
two loops in the shipped build (the fences cost a few instructions)."""
import collections
import json
import re
import sys

CLASSES = ("f64_fma_mul_add", "f64_other", "v_cndmask", "v_readlane", "valu_other", "lds", "vmem", "scratch", "salu", "s_waitcnt", "s_nop",
           "branch", "barrier", "other")


def klass(op):
    if op in ("v_fma_f64", "v_mul_f64", "v_add_f64"):
        return "f64_fma_mul_add"
    if op.startswith("v_") and "f64" in op:
        return "f64_other"
    if op.startswith("v_cndmask"):
        return "v_cndmask"
    if op.startswith("v_readlane") or op.startswith("v_readfirstlane"):
        return "v_readlane"
    if op.startswith("v_"):
        return "valu_other"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith("scratch_"):
        return "scratch"
    if op.startswith(("global_", "buffer_", "flat_")):
        return "vmem"
    if op == "s_barrier":
        return "barrier"
    if op.startswith("s_waitcnt"):
        return "s_waitcnt"
    if op == "s_nop":
        return "s_nop"
    if op.startswith("s_cbranch") or op == "s_branch":
        return "branch"
    if op.startswith("s_"):
        return "salu"
    return "other"


def kernel_body(path, key):
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if key in l and l.startswith("_Z") and l.rstrip().split(":")[0].endswith("Pi"))
    end = next(i for i in range(start, len(lines)) if ".Lfunc_end" in lines[i])
    return lines[start:end]


def count(seg, cold=None):
    """class histogram of a listing segment; instructions inside @@COLD begin/end brackets go to `cold` when given"""
    c = collections.Counter()
    depth = 0
    for l in seg:
        if "@@COLD begin" in l:
            depth += 1
        elif "@@COLD end" in l:
            depth -= 1
        m = re.match(r"\t([a-z_0-9]+)", l)
        if m:
            (cold if (depth > 0 and cold is not None) else c)[klass(m.group(1))] += 1
    return c


def marked(path, key):
    body = kernel_body(path, key)
    marks = [(i, l.split("@@MARK")[1].strip()) for i, l in enumerate(body) if "@@MARK" in l]
    out = collections.OrderedDict()
    for (i, name), (j, nxt) in zip(marks, marks[1:]):
        if name.endswith("loop_end"):
            continue
        seg = body[i:j]
        if nxt.endswith("loop_end"):   # the tail runs to the loop's back edge, the rest (up to the mark) is the exit path
            back = [k for k, l in enumerate(seg) if re.match(r"\ts_c?branch", l)]
            seg = seg[:back[0] + 1] if back else seg
        cold = collections.Counter()
        out[name] = (count(seg, cold), cold)
    return out


def loops(path, key, minlen=150):
    """the loops of the plain listing that hold exactly one s_barrier (the update loops), largest first"""
    body = kernel_body(path, key)
    labels = {}
    for i, l in enumerate(body):
        m = re.match(r"(\.LBB\d+_\d+):", l)
        if m:
            labels[m.group(1)] = i
    found = []
    for lab, i in labels.items():
        if "Loop Header" not in body[i] or "Depth=1" not in body[i]:
            continue
        last = max((j for j, l in enumerate(body) if re.search(r"s_c?branch\S*\s+" + re.escape(lab) + r"\b", l)), default=-1)
        seg = body[i:last + 1]
        if last - i >= minlen and sum(1 for l in seg if "\ts_barrier" in l) == 1:
            found.append((i, count(seg)))
    return found


if __name__ == "__main__":
    marks_s, plain_s, key, out = sys.argv[1:5]
    reg = marked(marks_s, key)
    trips = {"worker.body_items": "2 or 3 per thread (8 N = 400 items on 192 threads, one item = the inner loop body)"}
    rows = []
    notes = {"worker.column_extraction": "7 predicated group bodies behind s_cbranch_execz; with the 8x8 tile map 2.6 of them run per wave and update on average (at most 4)",
             "worker.body_block": "the top 128 threads only (waves 1 and 2)",
             "service.predict": "includes the result store (lane 0) and the static table read of the measurement after next"}
    for name, (c, cold) in reg.items():
        rows.append({"region": name, "total": sum(c.values()), **{k: c[k] for k in CLASSES if c[k]}, "cold_path_total": sum(cold.values()),
                     **({"trips": trips[name]} if name in trips else {}), **({"note": notes[name]} if name in notes else {})})
    tot = {w: sum(r["total"] for r in rows if r["region"].startswith(w)) for w in ("worker", "service")}
    f64 = {w: sum(r.get("f64_fma_mul_add", 0) + r.get("f64_other", 0) for r in rows if r["region"].startswith(w)) for w in ("worker", "service")}
    plain = [{"first_line": i, "total": sum(c.values()), **{k: c[k] for k in CLASSES if c[k]}} for i, c in loops(plain_s, key)]
    j = {"kernel": key, "what": "static wave-instructions per update phase by region (fenced accounting build; `total` and the classes are the "
                                 "usually executed path, `cold_path_total` the bracketed code a usual update skips: medium / large-angle exp, fix_depth "
                                 "edits), and of the same loops in the shipped build",
         "regions": rows, "static_total_per_update": tot, "static_f64_per_update": f64, "shipped_build_update_loops": plain}
    json.dump(j, open(out, "w"), indent=1)
    hdr = "%-28s %6s %5s " % ("region", "total", "cold") + " ".join("%9s" % k[:9] for k in CLASSES)
    print(hdr)
    for r in rows:
        print("%-28s %6d %5d " % (r["region"], r["total"], r["cold_path_total"]) + " ".join("%9d" % r.get(k, 0) for k in CLASSES))
    print("static per update: worker %d (f64 %d)   service %d (f64 %d)" % (tot["worker"], f64["worker"], tot["service"], f64["service"]))
    for p in plain:
        print("shipped loop @%d: total %d  " % (p["first_line"], p["total"]) + "  ".join("%s %d" % (k, p[k]) for k in CLASSES if k in p))
