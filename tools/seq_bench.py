"""IMU steps/s through the DROP-IN route: viekf_seq_propagate / _add_frame / _handle_measurements, 250 Hz IMU : 30 Hz camera
(params/sim_params.yaml:149,160) with the camera's 30 ms delay (:161) -- every frame rewinds the filter and replays the inputs
since its time stamp (src/vi_ekf/vi_ekf_meas.cpp:45-116), which is the NORMAL path of test/vi_ekf_test.cpp.
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


def run(B=1024, N=50, independent=False, frames=12, per_feature=False, delay=0.03, state_hist=24):
    sc = scene.make_scene(B, N, 32, seed=77)
    g = v.BatchVIEKF(B, N, sc["params"])
    s = SeqVIEKF(g, state_hist=state_hist, meas_hist=4 * N, independent=independent)
    ids = np.arange(N, dtype=np.int32)
    R = np.asarray(sc["R"]).reshape(2, 2)
    dt = 0.004
    # first frame at t = 0 initialises the features (unknown ids: vi_ekf_meas.cpp:140-147)
    s.propagate_state(sc["u"][0], 0.0)
    s.add_frame(0.0, sc["pix"], R, ids)
    k, imu, nframes = 1, 0, 0
    next_cam = 1.0 / 30.0
    pending = []          # (arrival time, stamp, frame index)
    torch.cuda.synchronize()
    t0 = None
    host = {"propagate": 0.0, "add_frame": 0.0, "handle": 0.0}   # time spent INSIDE the calls (they only queue device work)
    while nframes < frames + 2:
        t = k * dt
        h0 = time.perf_counter()
        s.propagate_state(sc["u"][k % 32], t)
        host["propagate"] += time.perf_counter() - h0
        imu += 1
        if t >= next_cam:
            pending.append((next_cam + delay, next_cam, k))
            next_cam += 1.0 / 30.0
        while pending and pending[0][0] <= t:
            _, stamp, kk = pending.pop(0)
            zf = sc["z"][kk % 32]
            if per_feature:
                for i in range(N):
                    s.add_measurement(stamp, zf[:, i, :], 6, R, True, int(i))
            else:
                h0 = time.perf_counter()
                s.add_frame(stamp, zf, R, ids)
                host["add_frame"] += time.perf_counter() - h0
            h0 = time.perf_counter()
            s.handle_measurements(want_gated=False)
            host["handle"] += time.perf_counter() - h0
            nframes += 1
            if nframes == 2:          # warm-up done
                g.sync()
                t0, imu = time.perf_counter(), 0
                host = {k2: 0.0 for k2 in host}
        k += 1
    g.sync()
    secs = time.perf_counter() - t0
    st = g.get_status()
    return {"imu_steps_per_s": B * imu / secs, "frames_per_s": B * (nframes - 2) / secs, "imu_steps": imu, "frames": nframes - 2,
            "seconds": secs, "host_seconds_in_calls": {k2: round(v2, 5) for k2, v2 in host.items()}, "bad_filters": int((st & (1 | 2 | 8) != 0).sum()),
            "what": "B=%d N=%d, %s clock, 250 Hz IMU : 30 Hz frames stamped %.0f ms back (rewind + replay every frame), %s"
                    % (B, N, "one per filter" if independent else "shared", delay * 1e3,
                       "one add_measurement per feature" if per_feature else "viekf_seq_add_frame")}


if __name__ == "__main__":
    a = sys.argv[1:]
    B = int(a[0]) if len(a) > 0 else 1024
    N = int(a[1]) if len(a) > 1 else 50
    ind = bool(int(a[2])) if len(a) > 2 else False
    fr = int(a[3]) if len(a) > 3 else 12
    pf = bool(int(a[4])) if len(a) > 4 else False
    print(run(B, N, ind, fr, pf))
