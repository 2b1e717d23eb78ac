"""Host sequencer (viekf_seq_*) on the device core vs the line-by-line restatement of the reference's plumbing
(oracle/seq_oracle.py): delayed measurements, rewind / replay, new features, keyframe trigger.  SURVEY 8(f)-1 / config 1."""
import numpy as np
import pytest

from oracle import oracle as orc
from oracle import seq_oracle as so
from tests.test_gpu_parity import assert_close


def _params(seed, identity_qbu=False):
    p = dict(orc.EKF_YAML)
    if identity_qbu:
        p["q_b_u"] = [1.0, 0.0, 0.0, 0.0]
    return p


def _run(B, N, seed, delay, identity_qbu=False, nan_filter=None, steps=60, hist=64, log_root=None, log_filter=0, frames=False,
         want_gated=True, late_alt=None, late_at=5, init_at_start=False):
    """frames=True: every camera frame goes in through ONE viekf_seq_add_frame call (one queue entry per frame) instead of one
    add_measurement per feature; want_gated=False: handle_measurements() without the optional list (nothing waited for)"""
    import vi_ekf_amd as v
    p = _params(seed, identity_qbu)
    g = v.BatchVIEKF(B, N, dict(p, keyframe_overlap_threshold=0.8, name="seq"))
    sg = v.SeqVIEKF(g, state_hist=hist, meas_hist=200)
    os_ = [so.SeqOracle(orc.OracleFilter(N).init(**p), 0.8, state_hist=hist) for _ in range(B)]
    if log_root is not None:
        sg.init_logger(log_root, "ekf", log_filter)
        os_[log_filter].enable_records()
    rng = np.random.default_rng(seed)
    pix = rng.uniform(120, 480, (B, N, 2))
    R = np.eye(2) * 10.0
    gated_g = [[] for _ in range(B)]
    gated_o = [[] for _ in range(B)]
    for k in range(steps):
        t = 0.004 * k
        u = np.tile(np.array([0, 0, -9.80665, 0, 0, 0.0]), (B, 1)) + rng.normal(0, 0.3, (B, 6)) * np.array([1, 1, 1, .05, .05, .05])
        sg.propagate_state(u, t)
        for b in range(B):
            os_[b].propagate_state(u[b], t)
        if init_at_start and k == 0:
            # every feature starts at t = 0 (unknown ids: initialised at the current state, nothing queued, vi_ekf_meas.cpp:140-147), so
            # that no later rewind goes back past a feature's initialisation -- a feature first seen by a DELAYED frame is initialised in
            # the newest slot only and the frame's own rewind drops it again (the reference's behaviour; the flights above have it)
            sg.add_frame(t, pix, R, np.arange(N))
            for b in range(B):
                for i in range(N):
                    os_[b].add_measurement(t, pix[b, i], orc.FEAT, R, True, i, float("nan"))
        if k % 7 == 3:   # a camera frame, time-stamped `delay` seconds ago: FEAT for every feature + an altimeter reading
            tz = t - delay
            zf = np.zeros((B, N, 2))
            for i in range(N):
                z = pix[:, i, :] + rng.normal(0, 0.5, (B, 2))
                if nan_filter is not None and i == 1 and k > 20:
                    z[nan_filter, 0] = np.nan
                zf[:, i, :] = z
                rg = None if frames else sg.add_measurement(tz, z, orc.FEAT, R, True, id=i)
                for b in range(B):
                    ro = os_[b].add_measurement(tz, z[b], orc.FEAT, R, True, i, float("nan"))
                    assert frames or rg[b] == ro, (k, i, b, rg[b], ro)
            if frames:
                rf = sg.add_frame(tz, zf, R, np.arange(N))
                assert rf.shape == (B, N)
            alt = rng.normal(2.0, 0.05, (B, 1))
            sg.add_measurement(tz + 0.001, alt, orc.ALT, np.array([[0.01]]), True)
            for b in range(B):
                os_[b].add_measurement(tz + 0.001, alt[b], orc.ALT, np.array([[0.01]]), True)
            gg = sg.handle_measurements(want_gated=want_gated)
            for b in range(B):
                if want_gated:
                    gated_g[b] += gg[b]
                gated_o[b] += os_[b].handle_measurements()
        if late_alt is not None and k % 7 == late_at and k > 7:   # a second, slower sensor: its reading is stamped INSIDE the span the
            tz = t - late_alt                                # last frame's replay covered
            alt = rng.normal(2.0, 0.05, (B, 1))
            sg.add_measurement(tz, alt, orc.ALT, np.array([[0.01]]), True)
            sg.handle_measurements(want_gated=False)
            for b in range(B):
                os_[b].add_measurement(tz, alt[b], orc.ALT, np.array([[0.01]]), True)
                os_[b].handle_measurements()
    return g, sg, os_, gated_g, gated_o


@pytest.mark.gpu
@pytest.mark.parametrize("delay", [0.0, 0.0105, 0.03])
def test_sequencer_matches_reference_plumbing(delay):
    B, N = 3, 6
    g, sg, os_, gg, go = _run(B, N, seed=3, delay=delay)
    x = g.get_state()
    P = g.get_covariance()
    for b in range(B):
        assert sg.tracked_features()[b] == list(os_[b].f.feature_ids)
        assert not os_[b].log, os_[b].log
    assert_close(x, np.stack([o.f.x for o in os_]), "x")
    assert_close(P, np.stack([o.f.P for o in os_]), "P")
    assert gg == go
    st = sg.status()
    assert st["ring_index"] == os_[0].i and abs(st["t"] - os_[0].t[os_[0].i]) < 1e-12
    assert st["queued"] == len(os_[0].zbuf) and st["inputs"] == len(os_[0].u)


@pytest.mark.gpu
@pytest.mark.parametrize("delay,late", [(0.03, 0.010), (0.03, 0.0185), (0.0105, 0.006)])
def test_rewind_into_a_fused_replay(delay, late):
    """The replay that closes handle_measurements runs as ONE fused launch and writes only its last ring slot
    (viekf_batch_propagate_n_to); a later measurement stamped inside that span makes the sequencer re-create the slot it rewinds to
    from the nearest written one.  Against the restated reference plumbing, which keeps every slot."""
    B, N = 3, 6
    g, sg, os_, gg, go = _run(B, N, seed=11, delay=delay, frames=True, late_alt=late)
    for b in range(B):
        assert sg.tracked_features()[b] == list(os_[b].f.feature_ids)
    assert_close(g.get_state(), np.stack([o.f.x for o in os_]), "x")
    assert_close(g.get_covariance(), np.stack([o.f.P for o in os_]), "P")
    assert gg == go
    st = sg.status()
    assert st["ring_index"] == os_[0].i and abs(st["t"] - os_[0].t[os_[0].i]) < 1e-12


@pytest.mark.gpu
@pytest.mark.parametrize("hist,late_at,late", [(12, 2, 0.038), (12, 2, 0.034), (12, 1, 0.038), (13, 2, 0.042), (11, 2, 0.030)])
def test_rewind_into_a_fused_replay_whose_start_left_the_ring(hist, late_at, late):
    """a ring that wraps past the slot a fused replay started from while steps of that replay are still unmaterialised,
    then a slower sensor stamped inside the span.  The reference's ring still holds x_ / P_ of that step (vi_ekf_meas.cpp:46-57) and
    fuses the reading; the sequencer serves the rewind from the buffer it took out of the ring when the start slot was due to be
    overwritten (viekf_seq.cpp: before_overwrite / rewind_to).  Frames every 7 steps stamped 30 ms back, ring of `hist` slots, an
    altimeter reading 5 - 6 steps after each frame stamped `late` s back.  Against the restated plumbing, which keeps every slot."""
    B, N = 3, 6
    g, sg, os_, gg, go = _run(B, N, seed=13, delay=0.03, frames=True, late_alt=late, late_at=late_at, hist=hist, steps=80,
                              init_at_start=True)
    for b in range(B):
        assert sg.tracked_features()[b] == list(os_[b].f.feature_ids)
        assert not [m for m in os_[b].log if "state buffer" in m], os_[b].log      # (the reference does fuse these readings)
    assert_close(g.get_state(), np.stack([o.f.x for o in os_]), "x")
    assert_close(g.get_covariance(), np.stack([o.f.P for o in os_]), "P")
    st = sg.status()
    assert st["ring_index"] == os_[0].i and abs(st["t"] - os_[0].t[os_[0].i]) < 1e-12
    assert st["inputs"] == len(os_[0].u)        # (a frame is ONE queue entry here: the queue lengths are not comparable)


@pytest.mark.gpu
@pytest.mark.parametrize("delay,want_gated", [(0.0, True), (0.0105, True), (0.03, True), (0.03, False)])
def test_whole_frames_in_one_call_equal_one_call_per_feature(delay, want_gated):
    """viekf_seq_add_frame queues a camera frame as ONE entry: bit for bit the filter that was fed one add_measurement per feature
    (same launches, same order), the restated reference plumbing to the parity tolerance, the same gated ids; a NaN pixel of one
    filter skips that filter's entry only.  want_gated=False: handle_measurements without the list waits for nothing."""
    B, N = 3, 6
    # (q_b_u = identity: a filter that skips an entry still takes part in the others' replays, which is exact only then)
    ga, sa, os_, gga, go = _run(B, N, seed=3, delay=delay, nan_filter=2, identity_qbu=True)
    gb, sb, _, ggb, _ = _run(B, N, seed=3, delay=delay, nan_filter=2, identity_qbu=True, frames=True, want_gated=want_gated)
    assert np.array_equal(ga.get_state(), gb.get_state()) and np.array_equal(ga.get_covariance(), gb.get_covariance())
    assert sa.tracked_features() == sb.tracked_features()
    if want_gated:
        assert gga == ggb == go
    assert_close(gb.get_state(), np.stack([o.f.x for o in os_]), "x")
    assert_close(gb.get_covariance(), np.stack([o.f.P for o in os_]), "P")


@pytest.mark.gpu
def test_sequencer_lockstep_with_a_skipping_filter():
    """one filter drops a measurement (NaN pixel): with q_b_u = identity the lockstep replay is exact for everybody"""
    B, N = 3, 5
    g, sg, os_, gg, go = _run(B, N, seed=5, delay=0.0105, identity_qbu=True, nan_filter=1)
    assert_close(g.get_state(), np.stack([o.f.x for o in os_]), "x")
    assert_close(g.get_covariance(), np.stack([o.f.P for o in os_]), "P")


@pytest.mark.gpu
def test_sequencer_keep_only_features_and_keyframe_reset():
    B, N = 2, 6
    g, sg, os_, _, _ = _run(B, N, seed=7, delay=0.0, steps=30)
    keep1 = np.array([[0, 1, 2, 3, 4, 5]] * B)          # first call only records the keyframe features
    did, _ = sg.keep_only_features(keep1)
    for b in range(B):
        os_[b].keep_only_features(keep1[b])
    assert not did.any()
    keep2 = np.array([[0, 2, 5, -1, -1, -1]] * B)        # 3 of 6 overlap < 0.8 -> drop 1,3,4 and reset
    did, edges = sg.keep_only_features(keep2)
    for b in range(B):
        os_[b].keep_only_features([0, 2, 5])
    assert did.all()
    for b in range(B):
        assert sg.tracked_features()[b] == list(os_[b].f.feature_ids) == [0, 2, 5]
        assert_close(edges[b], os_[b].keyframe_edges[-1], "edge")
    assert_close(g.get_state(), np.stack([o.f.x for o in os_]), "x")
    assert_close(g.get_covariance(), np.stack([o.f.P for o in os_]), "P")


@pytest.mark.gpu
@pytest.mark.parametrize("delay", [0.0, 0.0105])
def test_log_writer_matches_reference_records(tmp_path, delay):
    """viekf_seq_init_logger: the binary files of vi_ekf_log.cpp (record layouts of log_state / log_measurement, what
    matlab/plot_ekf.m reads) against the same records from the restated plumbing"""
    B, N, lf = 3, 5, 1
    root = str(tmp_path) + "/"
    g, sg, os_, _, _ = _run(B, N, seed=11, delay=delay, steps=40, log_root=root, log_filter=lf)
    sg.disable_logger()
    nx, n = 17 + 5 * N, 16 + 3 * N
    rec = os_[lf].rec

    def load(name, width):
        a = np.fromfile(root + "ekf_" + name, dtype=np.float64)
        assert a.size % width == 0, (name, a.size, width)
        return a.reshape(-1, width)

    st = load("state.log", 1 + nx)
    assert st.shape[0] == len(rec["state"]) > 10
    for name, width, key in (("state.log", 1 + nx, "x"), ("cov.log", 1 + n, "Pd"), ("input.log", 7, "u"),
                             ("xdot.log", 1 + n, "xdot"), ("feat_id.log", 1 + N, "ids")):
        a = load(name, width)
        ref = np.stack([np.concatenate([[r["t"]], r[key]]) for r in rec["state"]])
        assert_close(a, ref, name)
    gp = load("global_pose.log", 8)
    ref = np.stack([np.concatenate([[r["t"]], r["gpose"]]) for r in rec["state"]])
    assert_close(gp, ref, "global_pose.log")     # (no keyframe reset in this run: node pose = identity)
    assert (ref[:, 1:4] == np.stack([r["x"][0:3] for r in rec["state"]])).all()
    for mtype, name, width in ((orc.FEAT, "FEAT.log", 1 + 2 + 2 + 1 + 1), (orc.ALT, "ALT.log", 1 + 1 + 1 + 1)):
        a = load(name, width)
        ref = np.stack(rec["meas"][mtype])
        assert a.shape == ref.shape and a.shape[0] >= 4
        assert_close(a, ref, name)
    import os
    for name in ("ACC.log", "kf.log", "debug.txt"):
        assert os.path.getsize(root + "ekf_" + name) == 0
    cfg = open(root + "ekf_config.txt").read().splitlines()
    assert cfg[0].startswith("Test Num: ") and cfg[-1].startswith("min_depth: ") and len(cfg) == 17
    assert cfg[-2] == "num features: %d" % N


@pytest.mark.gpu
def test_global_pose_and_covariance_through_keyframe_resets(tmp_path):
    """VIEKF::get_global_pose / get_global_cov and the node update at the end of keyframe_reset (vi_ekf_kfr.cpp:14-53,147-150)
    through three resets, against the restated plumbing; the global_pose log record is global after a reset"""
    B, N, lf = 2, 6, 1
    root = str(tmp_path) + "/"
    g, sg, os_, _, _ = _run(B, N, seed=21, delay=0.0, steps=25, log_root=root, log_filter=lf)
    t = 0.004 * 24
    rng = np.random.default_rng(3)
    R = np.eye(2) * 10.0

    def fly(k):
        nonlocal t
        for _ in range(k):
            t += 0.004
            u = np.tile(np.array([0.3, -0.2, -9.80665, 0.02, 0.01, 0.2]), (B, 1)) + rng.normal(0, 0.3, (B, 6))
            sg.propagate_state(u, t)
            for b in range(B):
                os_[b].propagate_state(u[b], t)

    def keep(ids):
        arr = np.array([list(ids) + [-1] * (6 - len(ids))] * B)
        did, _ = sg.keep_only_features(arr)
        for b in range(B):
            os_[b].keep_only_features(list(ids))
        return bool(did.all())

    def check(what):
        pose, node = sg.get_global_pose()
        cov = sg.get_global_cov()
        for b in range(B):
            assert_close(node[b], os_[b].node, "node pose " + what)
            assert_close(pose[b], os_[b].get_global_pose(), "global pose " + what)
            assert_close(cov[b], os_[b].get_global_cov(), "global covariance " + what)
        return node

    assert not keep([0, 1, 2, 3, 4, 5])            # first call only records the keyframe features
    check("before any reset")
    fly(5)
    assert keep([0, 2, 5])                         # 3 of 6 < 0.8: reset 1
    fly(7)
    check("after reset 1")
    assert keep([0])                               # 1 of 3: reset 2
    fly(4)
    check("after reset 2")
    for i in range(3):                             # three new features (the filter numbers them 6, 7, 8 itself)
        z = rng.uniform(150, 450, (B, 2))
        rg = sg.add_measurement(t, z, orc.FEAT, R, True, id=6 + i)
        for b in range(B):
            assert os_[b].add_measurement(t, z[b], orc.FEAT, R, True, 6 + i, float("nan")) == rg[b] == orc.MEAS_NEW_FEATURE
    assert keep([6, 7, 8])                         # 0 of 1: reset 3, feature 0 dropped
    fly(6)
    node = check("after reset 3")
    assert np.abs(node[:, :3]).max() > 1e-4 and np.abs(node[:, 3] - 1.0).max() > 1e-9   # the node really moved and turned
    sg.disable_logger()
    gp = np.fromfile(root + "ekf_global_pose.log", dtype=np.float64).reshape(-1, 8)
    ref = np.stack([np.concatenate([[r["t"]], r["gpose"]]) for r in os_[lf].rec["state"]])
    assert_close(gp, ref, "global_pose.log through resets")
    assert np.abs(ref[-1, 1:4] - os_[lf].rec["state"][-1]["x"][0:3]).max() > 1e-6   # global != node-relative by now


def _feed(sg, sel, B, N, seed, clock0, dt_imu, delay, steps, alt_every=0, frames=False):
    """one source (rosbag) per filter: its own clock origin, IMU period and camera delay.  `sel` = the filters of the batch
    this source feeds (all others are masked out of its calls); returns nothing -- the sequencer holds the state"""
    rng = np.random.default_rng(seed)
    pix = rng.uniform(120, 480, (N, 2))
    R = np.eye(2) * 10.0
    mask = np.zeros(B, dtype=np.uint8)
    mask[sel] = 1
    ind = getattr(sg, "independent", False)
    tt = (lambda t: np.full(B, t)) if ind else (lambda t: t)     # (a lock-step sequencer takes the shared stamp)
    if not ind:
        mask = None
    for k in range(steps):
        t = clock0 + dt_imu * k
        u1 = np.array([0.1, -0.05, -9.80665, 0.01, 0.0, 0.02]) + rng.normal(0, 0.3, 6) * np.array([1, 1, 1, .05, .05, .05])
        u = np.zeros((B, 6))
        u[sel] = u1
        yield ("imu", k)
        sg.propagate_state(u, tt(t), mask=mask)
        if k % 7 == 3:
            tz = t - delay
            zf = np.zeros((B, N, 2))
            for i in range(N):
                z = np.zeros((B, 2))
                z[sel] = pix[i] + rng.normal(0, 0.5, 2)
                zf[:, i, :] = z
                if not frames:
                    sg.add_measurement(tt(tz), z, orc.FEAT, R, True, id=i, mask=mask)
            if frames:   # the whole frame in one call (independent clocks: every filter's entries go into its queue at once)
                sg.add_frame(tt(tz), zf, R, np.arange(N), mask=mask)
            if alt_every and k % alt_every == 3:
                alt = np.zeros((B, 1))
                alt[sel] = rng.normal(2.0, 0.05)
                sg.add_measurement(tt(tz + 0.001), alt, orc.ALT, np.array([[0.01]]), True, mask=mask)
            yield ("frame", k)


@pytest.mark.gpu
@pytest.mark.parametrize("frames", [False, True])
def test_independent_clocks_equal_separate_filters_bit_for_bit(frames):
    """Filters fed from different sources share a batch (north_star: one filter per trajectory / rosbag): different clock
    origins, IMU periods and camera delays (0 and 30 ms) -- per-filter deferral, rewind target and replay length
    (vi_ekf_meas.cpp:6-127 per filter).  Each filter must equal, bit for bit, a batch of one fed the same source."""
    import vi_ekf_amd as v
    N, steps = 5, 48
    p = dict(_params(0), keyframe_overlap_threshold=0.8, name="ind")
    sources = [dict(seed=31, clock0=0.0, dt_imu=0.004, delay=0.0, alt_every=14),
               dict(seed=32, clock0=1000.0, dt_imu=0.005, delay=0.03, alt_every=0),
               dict(seed=33, clock0=-50.0, dt_imu=0.004, delay=0.0105, alt_every=7)]
    B = len(sources)
    # together: one batch, the sources interleaved call by call (each call carries one source's sample, the others masked out)
    g = v.BatchVIEKF(B, N, p)
    sg = v.SeqVIEKF(g, state_hist=64, meas_hist=200, independent=True)
    gens = [_feed(sg, [b], B, N, steps=steps, frames=frames, **src) for b, src in enumerate(sources)]
    live = list(range(B))
    while live:
        for b in list(live):
            try:
                ev = next(gens[b])
                if ev[0] == "frame":
                    sg.handle_measurements()     # (every filter with queued work takes its own steps)
            except StopIteration:
                live.remove(b)
    x_all, P_all = g.get_state(), g.get_covariance()
    tracked = sg.tracked_features()
    # apart: a batch of one per source, same sequencer mode
    for b, src in enumerate(sources):
        g1 = v.BatchVIEKF(1, N, p)
        s1 = v.SeqVIEKF(g1, state_hist=64, meas_hist=200, independent=True)
        for ev in _feed(s1, [0], 1, N, steps=steps, **src):
            if ev[0] == "frame":
                s1.handle_measurements()
        assert np.array_equal(g1.get_state()[0], x_all[b]), "filter %d: state differs from its own batch of one" % b
        assert np.array_equal(g1.get_covariance()[0], P_all[b]), "filter %d: covariance differs" % b
        assert s1.tracked_features()[0] == tracked[b]
        # ... and the independent-clock plumbing equals the restated reference plumbing fed the same source
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
                    o.add_measurement(tz, pix[i] + rng.normal(0, 0.5, 2), orc.FEAT, np.eye(2) * 10.0, True, i, float("nan"))
                if src["alt_every"] and k % src["alt_every"] == 3:
                    o.add_measurement(tz + 0.001, np.array([rng.normal(2.0, 0.05)]), orc.ALT, np.array([[0.01]]), True)
                o.handle_measurements()
        assert not o.log, o.log
        assert_close(x_all[b], o.f.x, "filter %d vs restated plumbing: x" % b)
        assert_close(P_all[b], o.f.P, "filter %d vs restated plumbing: P" % b)
    assert np.abs(x_all[0] - x_all[1]).max() > 1e-6   # (the sources really differ)
