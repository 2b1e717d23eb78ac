"""Synthetic "static hover" scene for parity tests and the benchmark.

multirotor_sim (the reference's input generator, test/vi_ekf_test.cpp:18-37) is an absent
submodule, so inputs are generated here: truth pose fixed at x0, N landmarks per filter seen
at random pixels, IMU = hover specific force + noise (pre-rotated so the filter's q_b_u.rota
recovers it, reference vi_ekf.cpp:265-267), pixel measurements = landmark pixel + noise.
Pure data generation: no filter arithmetic happens here.
"""
import numpy as np

# reference params/ekf.yaml
EKF_YAML = dict(
    x0=[0, 0, -2, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0.1],
    P0=[1e-3] * 9 + [2e-1] * 3 + [1e-1] * 3 + [1e-6],
    Qx=[0.0] * 16,
    lam=[1.0] * 9 + [0.1] * 6 + [0.01],
    Qu=[1.0, 1.0, 1.0, 4e-4, 4e-4, 4e-4],
    P0_feat=[0.01, 0.01, 0.3],
    Qx_feat=[0.0, 0.0, 0.0],
    lam_feat=[1.0, 1.0, 0.4],
    cam_center=[315.83184814453125, 242.1165771484375],
    focal_len=[611.1864013671875, 611.5557861328125],
    q_b_c=[0.9974335273839892, 0.019768487288642146, -0.03564306227555538, 0.05886541830542371],
    p_b_c=[0.17363129, -0.02205945, 0.05490228],
    q_b_u=[0.993760669165504, 0.0497294816014604, 0.0997086508721388, 0.00498959122946198],
    min_depth=1.5, keyframe_overlap_threshold=0.8,
    use_drag_term=1, use_partial_update=1, use_keyframe_reset=0, name="ekf1",
)
DT = 0.004          # 250 Hz IMU (reference params/sim_params.yaml:149)
R_PIX = np.diag([10.0, 10.0])   # feat_R (reference params/ekf.yaml:47)


def _rotp(q, v):
    """R(q) v for the passive matrix of the reference's quaternion convention (vectorised)."""
    w, qv = q[0], np.asarray(q[1:4])
    t = -2.0 * np.cross(qv, v)
    return v + w * t - np.cross(qv, t)


def make_scene(batch, num_features, steps, seed=0x5EED, accel_sigma=1.0, gyro_sigma=0.02, pix_sigma=0.5,
               params=None):
    """-> dict(params, pix [B,N,2], u [steps,B,6], z [steps,B,N,2], slot [B,N], dt [B], R [2,2])

    Measurements are listed in the order the reference would process them (slot N-1 ... 0: the
    reverse in-frame order of vi_ekf_meas.cpp:150-176,16-18).
    """
    p = dict(EKF_YAML)
    if params:
        p.update(params)
    r = np.random.default_rng(seed)
    B, N = batch, num_features
    pix = np.stack([r.uniform(40, 600, (B, N)), r.uniform(40, 440, (B, N))], axis=-1)
    u_body = np.zeros((steps, B, 6))
    u_body[..., 2] = -9.80665
    u_body[..., 0:3] += r.normal(0, accel_sigma, (steps, B, 3))
    u_body[..., 3:6] += r.normal(0, gyro_sigma, (steps, B, 3))
    q_b_u = np.asarray(p["q_b_u"], float)
    u = np.concatenate([_rotp(q_b_u, u_body[..., 0:3]), _rotp(q_b_u, u_body[..., 3:6])], axis=-1)
    slot = np.tile(np.arange(N - 1, -1, -1, dtype=np.int32), (B, 1))
    z = pix[None, :, ::-1, :] + r.normal(0, pix_sigma, (steps, B, N, 2))  # z[s,b,m] belongs to slot[b,m] = N-1-m
    return dict(params=p, pix=pix, u=np.ascontiguousarray(u), z=np.ascontiguousarray(z), slot=slot,
                dt=np.full(B, DT), R=R_PIX.copy())


def algorithmic_bytes_per_step(num_features):
    """compulsory HBM traffic of one fused step of one filter."""
    N = num_features
    n, nx = 16 + 3 * N, 17 + 5 * N
    return 2 * 8 * n * n + 2 * 8 * nx + 56 + 24 * N
