"""Counts scratch (spill) ops between s_barrier marks of one kernel in a -save-temps .s file."""
import sys

lines = open(sys.argv[1]).read().split("\n")
key = sys.argv[2]
start = next(i for i, l in enumerate(lines) if l.startswith(key) and ":" in l)
end = next(i for i in range(start, len(lines)) if lines[i].startswith("\t.section") or "s_endpgm" in lines[i] and i > start + 50)
end = next(i for i in range(start, len(lines)) if ".Lfunc_end" in lines[i])
body = lines[start:end]
prev = 0
marks = [i for i, l in enumerate(body) if "s_barrier" in l] + [len(body)]
for b in marks:
    seg = body[prev:b]
    sc = sum(1 for l in seg if "scratch_" in l)
    st = sum(1 for l in seg if "scratch_store" in l)
    gl = sum(1 for l in seg if "global_load" in l or "global_store" in l)
    ds = sum(1 for l in seg if "\tds_" in l)
    fma = sum(1 for l in seg if "v_fma_f64" in l or "v_mul_f64" in l or "v_add_f64" in l)
    print("%6d-%6d len %5d  scratch %4d (st %3d)  global %3d  ds %4d  f64ops %4d" % (prev, b, b - prev, sc, st, gl, ds, fma))
    prev = b
