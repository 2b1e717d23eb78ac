import sys, numpy as np
sys.path.insert(0, '.')
import vi_ekf_amd as v
from oracle import oracle as orc, seq_oracle as so
from tests.test_gpu_sequencer import _feed, _params
N, steps = 5, int(sys.argv[1]) if len(sys.argv) > 1 else 48
p = dict(_params(0), keyframe_overlap_threshold=0.8, name="ind")
src = dict(seed=31, clock0=0.0, dt_imu=0.004, delay=0.0, alt_every=14)
out = {}
for mode in ("indep", "lock"):
    g1 = v.BatchVIEKF(1, N, p)
    s1 = v.SeqVIEKF(g1, state_hist=64, meas_hist=200, independent=(mode == "indep"))
    for ev in _feed(s1, [0], 1, N, steps=steps, **src):
        if ev[0] == "frame":
            s1.handle_measurements()
    out[mode] = (g1.get_state()[0], g1.get_covariance()[0], s1.tracked_features()[0], g1.get_len_features()[0])
    print(mode, "tracked", out[mode][2], "len", out[mode][3], s1.status())
o = so.SeqOracle(orc.OracleFilter(N).init(**_params(0)), 0.8, state_hist=64)
rng = np.random.default_rng(src["seed"])
pix = rng.uniform(120, 480, (N, 2))
for k in range(steps):
    t = src["clock0"] + src["dt_imu"] * k
    u1 = np.array([0.1, -0.05, -9.80665, 0.01, 0.0, 0.02]) + rng.normal(0, 0.3, 6) * np.array([1, 1, 1, .05, .05, .05])
    o.propagate_state(u1, t)
    if k % 7 == 3:
        tz = t - src["delay"]
        for i in range(N):
            r = o.add_measurement(tz, pix[i] + rng.normal(0, 0.5, 2), orc.FEAT, np.eye(2) * 10.0, True, i, float("nan"))
        if src["alt_every"] and k % src["alt_every"] == 3:
            o.add_measurement(tz + 0.001, np.array([rng.normal(2.0, 0.05)]), orc.ALT, np.array([[0.01]]), True)
        o.handle_measurements()
print("oracle tracked", list(o.f.feature_ids), o.f.len_features, o.log[:3])
for mode in out:
    print(mode, "x err vs oracle", np.abs(out[mode][0] - o.f.x).max(), "P err", np.abs(out[mode][1] - o.f.P).max())
print("indep vs lock x", np.abs(out["indep"][0] - out["lock"][0]).max())
