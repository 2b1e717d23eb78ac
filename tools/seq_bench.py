"""IMU steps/s through the DROP-IN route: viekf_seq_propagate / _add_frame / _handle_measurements, 250 Hz IMU : 30 Hz camera
(params/sim_params.yaml:149,160) with the camera's 30 ms delay (:161) -- every frame rewinds the filter and replays the inputs
since its time stamp (src/vi_ekf/vi_ekf_meas.cpp:45-116), which is the NORMAL path of test/vi_ekf_test.cpp.

Before the timed run the SAME call sequence is flown for `parity_frames` frames on a fresh batch and a strided sample of the
filters (every dispatch round) is compared with the restated reference plumbing (oracle/seq_oracle.SeqOracle over the C oracle:
vi_ekf_meas.cpp:6-127, vi_ekf.cpp:262-318) -- x, P, tracked ids, ring index -- so the figure is about work that has been checked
at the size it is timed at.  The oracle is the checker here, never the thing timed.
usage: python tools/seq_bench.py [B] [N] [independent 0|1] [frames] [per_feature_calls 0|1]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402,F401

import vi_ekf_amd as v  # noqa: E402
from vi_ekf_amd import scene  # noqa: E402
from vi_ekf_amd.seq import SeqVIEKF  # noqa: E402

DT = 0.004
FEAT = 6


def calls(sc, frames, delay):
    """the call sequence of the run: ("prop", k, t) | ("frame", kk, stamp) | ("handle",) | ("mark", frames handled so far)"""
    yield ("prop", 0, 0.0)
    yield ("frame", None, 0.0)        # first frame at t = 0 initialises the features (unknown ids: vi_ekf_meas.cpp:140-147)
    k, nframes = 1, 0
    next_cam = 1.0 / 30.0
    pending = []                      # (arrival time, stamp, frame index)
    while nframes < frames:
        t = k * DT
        yield ("prop", k, t)
        if t >= next_cam:
            pending.append((next_cam + delay, next_cam, k))
            next_cam += 1.0 / 30.0
        while pending and pending[0][0] <= t:
            _, stamp, kk = pending.pop(0)
            yield ("frame", kk, stamp)
            yield ("handle",)
            nframes += 1
            yield ("mark", nframes)
        k += 1


def parity_check(sc, B, N, independent, frames, delay, state_hist, nsample=6):
    """-> {max_rel_err, filters, frames, ...}: HIP sequencer vs SeqOracle on a strided sample, same calls (checker leg, untimed)"""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import oracle as orc
    from oracle import seq_oracle as so
    which = [int(b) for b in np.unique(np.linspace(0, B - 1, nsample).round().astype(int))]
    g = v.BatchVIEKF(B, N, sc["params"])
    s = SeqVIEKF(g, state_hist=state_hist, meas_hist=4 * N, independent=independent)
    ids = np.arange(N, dtype=np.int32)
    R = np.asarray(sc["R"]).reshape(2, 2)
    seq = list(calls(sc, frames, delay))
    for c in seq:
        if c[0] == "prop":
            s.propagate_state(sc["u"][c[1] % 32], c[2])
        elif c[0] == "frame":
            s.add_frame(c[2], sc["pix"] if c[1] is None else sc["z"][c[1] % 32], R, ids)
        elif c[0] == "handle":
            s.handle_measurements(want_gated=False)
    g.sync()
    x, P = g.get_state(), g.get_covariance()
    tracked = s.tracked_features()
    st = s.status()
    keys = ("x0", "P0", "Qx", "lam", "Qu", "P0_feat", "Qx_feat", "lam_feat", "cam_center", "focal_len", "q_b_c",
            "p_b_c", "q_b_u", "min_depth", "use_drag_term", "use_partial_update", "use_keyframe_reset")
    p = sc["params"]

    def fly(b):
        o = so.SeqOracle(orc.OracleFilter(N).init(**{k: p[k] for k in keys}), 0.8, state_hist=state_hist, meas_hist=4 * N)
        for c in seq:
            if c[0] == "prop":
                o.propagate_state(sc["u"][c[1] % 32][b], c[2])
            elif c[0] == "frame":
                zf = sc["pix"][b] if c[1] is None else sc["z"][c[1] % 32][b]
                for i in range(N):
                    o.add_measurement(c[2], zf[i], FEAT, R, True, i, float("nan"))
            elif c[0] == "handle":
                o.handle_measurements()
        return o
    t0 = time.perf_counter()
    with ThreadPoolExecutor(max_workers=min(len(which), os.cpu_count() or 1)) as ex:
        os_ = list(ex.map(fly, which))
    secs = time.perf_counter() - t0
    err = 0.0
    for j, b in enumerate(which):
        o = os_[j]
        if o.log:
            raise SystemExit("seq parity: the restated plumbing complained for filter %d: %s" % (b, o.log[:3]))
        if tracked[b] != list(o.f.feature_ids):
            raise SystemExit("seq parity: tracked feature ids of filter %d differ" % b)
        err = max(err, float(np.abs(x[b] - o.f.x).max() / np.abs(o.f.x).max()), float(np.abs(P[b] - o.f.P).max() / np.abs(o.f.P).max()))
    if st["ring_index"] != os_[0].i or abs(st["t"] - os_[0].t[os_[0].i]) > 1e-12 or st["inputs"] != len(os_[0].u):
        raise SystemExit("seq parity: ring index / time / input count of filter %d differ: %s vs %d, %.6f, %d"
                         % (which[0], st, os_[0].i, os_[0].t[os_[0].i], len(os_[0].u)))
    bad = int((g.get_status() & (1 | 2 | 8) != 0).sum())
    del s
    g.close()
    return {"max_rel_err": err, "filters": which, "frames": frames, "oracle_seconds": round(secs, 2), "bad_filters": bad,
            "checked": "x, P, tracked ids of the sampled filters, ring index / time / input count of the first, vs oracle/seq_oracle.SeqOracle"}


def run(B=1024, N=50, independent=False, frames=12, per_feature=False, delay=0.03, state_hist=24, parity_frames=0):
    sc = scene.make_scene(B, N, 32, seed=77)
    par = None
    if parity_frames > 0:
        par = parity_check(sc, B, N, independent, parity_frames, delay, state_hist)
        if not (par["max_rel_err"] <= 1e-6):
            raise SystemExit("PARITY FAILURE of the sequencer route vs the restated plumbing: rel err %.3e" % par["max_rel_err"])
    g = v.BatchVIEKF(B, N, sc["params"])
    s = SeqVIEKF(g, state_hist=state_hist, meas_hist=4 * N, independent=independent)
    ids = np.arange(N, dtype=np.int32)
    R = np.asarray(sc["R"]).reshape(2, 2)
    imu, nframes = 0, 0
    torch.cuda.synchronize()
    t0 = None
    host = {"propagate": 0.0, "add_frame": 0.0, "handle": 0.0}   # time spent INSIDE the calls (they only queue device work)
    for c in calls(sc, frames + 2, delay):
        if c[0] == "prop":
            h0 = time.perf_counter()
            s.propagate_state(sc["u"][c[1] % 32], c[2])
            host["propagate"] += time.perf_counter() - h0
            imu += 1
        elif c[0] == "frame":
            zf = sc["pix"] if c[1] is None else sc["z"][c[1] % 32]
            if per_feature and c[1] is not None:
                for i in range(N):
                    s.add_measurement(c[2], zf[:, i, :], FEAT, R, True, int(i))
            else:
                h0 = time.perf_counter()
                s.add_frame(c[2], zf, R, ids)
                host["add_frame"] += time.perf_counter() - h0
        elif c[0] == "handle":
            h0 = time.perf_counter()
            s.handle_measurements(want_gated=False)
            host["handle"] += time.perf_counter() - h0
        elif c[0] == "mark":
            nframes = c[1]
            if nframes == 2:          # warm-up done
                g.sync()
                t0, imu = time.perf_counter(), 0
                host = {k2: 0.0 for k2 in host}
    g.sync()
    secs = time.perf_counter() - t0
    st = g.get_status()
    out = {"imu_steps_per_s": B * imu / secs, "frames_per_s": B * (nframes - 2) / secs, "imu_steps": imu, "frames": nframes - 2,
           "seconds": secs, "host_seconds_in_calls": {k2: round(v2, 5) for k2, v2 in host.items()}, "bad_filters": int((st & (1 | 2 | 8) != 0).sum()),
           "what": "B=%d N=%d, %s clock, 250 Hz IMU : 30 Hz frames stamped %.0f ms back (rewind + replay every frame), %s"
                   % (B, N, "one per filter" if independent else "shared", delay * 1e3,
                      "one add_measurement per feature" if per_feature else "viekf_seq_add_frame")}
    if par is not None:
        out["parity_max_rel_err"] = par["max_rel_err"]
        out["parity"] = par
    return out


if __name__ == "__main__":
    a = sys.argv[1:]
    B = int(a[0]) if len(a) > 0 else 1024
    N = int(a[1]) if len(a) > 1 else 50
    ind = bool(int(a[2])) if len(a) > 2 else False
    fr = int(a[3]) if len(a) > 3 else 12
    pf = bool(int(a[4])) if len(a) > 4 else False
    print(run(B, N, ind, fr, pf, parity_frames=int(a[5]) if len(a) > 5 else 0))
