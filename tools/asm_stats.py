"""Prints per-kernel register / spill / scratch numbers from a hipcc -save-temps .s file."""
import re
import sys

txt = open(sys.argv[1]).read()
for blk in txt.split("  - .agpr_count:")[1:]:
    g = lambda k: (re.search(r"\.%s:\s+(\S+)" % k, blk) or [None, "?"])[1]
    print("%-90s vgpr %4s spill_v %4s spill_s %4s scratch %5s lds %6s" % (
        g("name")[:90], g("vgpr_count"), g("vgpr_spill_count"), g("sgpr_spill_count"),
        g("private_segment_fixed_size"), g("group_segment_fixed_size")))
