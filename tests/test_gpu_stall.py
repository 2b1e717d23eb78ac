"""The one bounded wait of the fused kernel -- the body wave polling the feature wave's mailbox (two service waves,
N + 14 > 64) -- must give up, say so (VIEKF_FLAG_INTERNAL) and let the launch end instead of hanging.  A -DVIEKF_TEST_STALL
build drops one hand-over of filter 0; the test runs it in a child process under a timeout."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "vi_ekf_amd", "csrc")

CHILD = r'''
import sys, numpy as np
sys.path.insert(0, %r)
import vi_ekf_amd as v
from vi_ekf_amd import scene
B, N = 3, 52
sc = scene.make_scene(B, N, 1, seed=8)
g = v.BatchVIEKF(B, N, sc["params"])
for i in range(N):
    g.init_feature(sc["pix"][:, i, :].copy(), np.full(B, np.nan))
res = g.step(sc["u"][0], sc["dt"], sc["z"][0], sc["slot"], sc["R"])
st = g.get_status()
x = g.get_state()
print("FLAGS", int(st[0]), int(st[1]), int(st[2]))
np.save(sys.argv[1], x)
'''


@pytest.mark.gpu
def test_stalled_mailbox_raises_internal_flag_and_returns(tmp_path):
    import numpy as np
    lib = str(tmp_path / "libviekf_stall.so")
    subprocess.check_call(["make", "-C", CSRC, "-s", "-j", str(min(8, os.cpu_count() or 1)), "OUT=" + lib,
                           "OBJDIR=" + str(tmp_path / "obj"), "EXTRA=-DVIEKF_TEST_STALL"])
    script = str(tmp_path / "child.py")
    open(script, "w").write(CHILD % ROOT)
    outs = {}
    for name, env_lib in (("stall", lib), ("normal", None)):
        env = dict(os.environ)
        if env_lib:
            env["VIEKF_LIB"] = env_lib
        else:
            env.pop("VIEKF_LIB", None)
        xf = str(tmp_path / (name + ".npy"))
        r = subprocess.run([sys.executable, script, xf], capture_output=True, text=True, timeout=120, env=env)   # must not hang
        assert r.returncode == 0, r.stderr[-2000:]
        flags = [int(t) for t in r.stdout.split("FLAGS")[1].split()[:3]]
        outs[name] = (flags, np.load(xf))
    assert outs["normal"][0] == [0, 0, 0]
    assert outs["stall"][0][0] & 8, "the stalled filter must carry VIEKF_FLAG_INTERNAL"
    assert outs["stall"][0][1] == 0 and outs["stall"][0][2] == 0
    # the other filters of the launch are untouched by it.  The two builds differ in the hand-over only: the listings of this
    # instance (hipcc -S of viekf_inst.hip, group 5, with and without -DVIEKF_TEST_STALL) hold the same 3,081 fp64 instructions,
    # three v_fma_f64 scheduled a few slots apart, the rest of the difference is scalar moves and lane reads -- so: bit for bit
    ref = outs["normal"][1][1:]
    d = np.abs(outs["stall"][1][1:] - ref).max()
    assert d == 0.0, d
