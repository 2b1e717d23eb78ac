"""In-tree build of libviekf_hip.so (hipcc, gfx950).  Used by __graft_entry__.build()."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "libviekf_hip.so")
CSRC = os.path.join(HERE, "csrc")


def sources():
    out = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith((".hip", ".cpp", ".hpp", "Makefile"))]
    out.append(os.path.join(os.path.dirname(HERE), "include", "viekf.h"))
    return out


def stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(s) > t for s in sources())


def build(force=False):
    if force or stale():
        jobs = str(max(1, min(8, os.cpu_count() or 1)))
        subprocess.check_call(["make", "-C", CSRC, "-s", "-j", jobs] + (["-B"] if force else []))
    return LIB
