"""A small multirotor simulator with the callback shape of the reference's input generator (test harness only).

multirotor_sim (reference test/vi_ekf_test.cpp:18-37,61: `Simulator`, `EstimatorWrapper::register_imu_cb / register_feat_cb`,
`run()`) is an absent submodule, so its role is played here by a closed-loop kinematic model written against the SAME
equations the filter integrates (reference src/vi_ekf/vi_ekf_dyn.cpp:36-52 with the drag term):

    p_dot = R(q)^T v          v_dot = a_z e_z + R(q) g e_z - w x v - mu v_xy          q_dot = q [+] w

so an estimator that restates the reference correctly must track it.  A cascaded position / tilt / yaw controller flies a
slow circle over a field of ground landmarks; the camera (p_b_c, q_b_c, pinhole focal_len / cam_center) reports the pixels
of up to `num_features` tracked landmarks with their ids; the IMU reports specific force and rates in the IMU frame
(q_b_u), with constant biases and white noise.  Pure data generation: no filter arithmetic happens here.

    sim = Simulator(params, num_features=8, seed=1)
    sim.register_imu_cb(lambda t, z, R: ekf.propagate_state(z, t))
    sim.register_feat_cb(lambda t, pixs, ids, R_pix: ...)
    ekf.propagate_state(sim.imu(), sim.t)          # as vi_ekf_test.cpp:57
    while sim.run(): pass
"""
import math

import numpy as np

G = 9.80665


def q_otimes(a, b):
    aw, av = a[0], a[1:4]
    bw, bv = b[0], b[1:4]
    return np.concatenate([[aw * bw - av @ bv], aw * bv + bw * av + np.cross(av, bv)])


def q_exp(v):
    th = np.linalg.norm(v)
    if th < 1e-9:
        q = np.concatenate([[1.0], 0.5 * v])
    else:
        q = np.concatenate([[math.cos(th / 2)], math.sin(th / 2) * v / th])
    return q / np.linalg.norm(q)


def q_rotp(q, v):
    """passive rotation R(q) v (inertial -> body for the attitude quaternion), reference src/quat.cpp:285-289"""
    w, qv = q[0], q[1:4]
    t = -2.0 * np.cross(qv, v)
    return v + w * t - np.cross(qv, t)


def q_rota(q, v):
    """active rotation R(q)^T v (body -> inertial), reference src/quat.cpp:279-283"""
    w, qv = q[0], q[1:4]
    t = 2.0 * np.cross(qv, v)
    return v + w * t + np.cross(qv, t)


class Simulator:
    def __init__(self, params, num_features=8, seed=1, tmax=8.0, imu_rate=250.0, cam_rate=25.0, radius=0.35, period=8.0,
                 accel_sigma=0.3, gyro_sigma=0.01, pix_sigma=0.5, accel_bias=(0.05, -0.04, 0.03), gyro_bias=(0.004, -0.003, 0.002)):
        self.p = dict(params)
        self.N = int(num_features)
        self.rng = np.random.default_rng(seed)
        self.dt = 1.0 / imu_rate
        self.cam_every = int(round(imu_rate / cam_rate))
        self.tmax = float(tmax)
        self.radius, self.period = float(radius), float(period)
        self.accel_sigma, self.gyro_sigma, self.pix_sigma = accel_sigma, gyro_sigma, pix_sigma
        self.accel_bias_ = np.asarray(accel_bias, float)
        self.gyro_bias_ = np.asarray(gyro_bias, float)
        x0 = np.asarray(self.p["x0"], float)
        self.pos = x0[0:3].copy()
        self.vel = x0[3:6].copy()                # body frame
        self.q = x0[6:10].copy()
        self.mu = float(x0[16])
        self.t = 0.0
        self.k = 0
        self.w = np.zeros(3)
        self.az = -G
        self._home = self.pos.copy()
        self.q_b_c = np.asarray(self.p["q_b_c"], float)
        self.p_b_c = np.asarray(self.p["p_b_c"], float)
        self.q_b_u = np.asarray(self.p["q_b_u"], float)
        self.f = np.asarray(self.p["focal_len"], float)
        self.c = np.asarray(self.p["cam_center"], float)
        g = np.arange(-3.0, 3.0001, 0.22)
        gx, gy = np.meshgrid(g, g, indexing="ij")
        jit = self.rng.uniform(-0.08, 0.08, (gx.size, 2))
        self.landmarks = np.stack([gx.ravel() + jit[:, 0], gy.ravel() + jit[:, 1], np.zeros(gx.size)], axis=1)   # ground, z = 0
        self.tracked = []                        # landmark indices being tracked
        self.feat_id = {}                        # landmark index -> feature id: a running count in order of first sight, the
        self.next_feat_id = 0                    # numbering the reference gives its features itself (vi_ekf_feat.cpp:29-30)
        self._imu_cb = None
        self._feat_cb = None
        self.R_imu = np.diag([accel_sigma ** 2] * 3 + [gyro_sigma ** 2] * 3)
        self.R_pix = np.diag([10.0, 10.0])       # feat_R of the reference's params/ekf.yaml:47
        self._control()

    # -- reference EstimatorWrapper::register_* -------------------------------------------------------------------------
    def register_imu_cb(self, cb):
        self._imu_cb = cb

    def register_feat_cb(self, cb):
        self._feat_cb = cb

    # -- truth ----------------------------------------------------------------------------------------------------------
    def state(self):
        """[pos(3), att(4), vel_body(3), omega(3)] like multirotor_sim's state().arr (test/vi_ekf_test.cpp:43)"""
        return np.concatenate([self.pos, self.q, self.vel, self.w])

    def commanded(self, t):
        a = 2.0 * math.pi * t / self.period
        ramp = min(1.0, t / 2.0)
        pc = self._home + ramp * np.array([self.radius * (math.cos(a) - 1.0), self.radius * math.sin(a), -0.3 * math.sin(0.5 * a)])
        return pc, 0.25 * ramp * math.sin(0.7 * a)

    def _control(self):
        """position -> desired specific force -> tilt + thrust; yaw -> yaw rate (ideal rate tracking)"""
        pc, yaw_c = self.commanded(self.t)
        v_i = q_rota(self.q, self.vel)
        a_des = 2.0 * (pc - self.pos) - 2.5 * v_i
        a_des = np.clip(a_des, -2.0, 2.0)
        f_i = a_des - np.array([0.0, 0.0, G])                    # specific force wanted, inertial
        thrust = np.linalg.norm(f_i)
        zb_des = q_rotp(self.q, -f_i / thrust)                   # desired body z axis (down), seen from the body
        ez = np.array([0.0, 0.0, 1.0])
        axis = np.cross(ez, zb_des)
        s = np.linalg.norm(axis)
        ang = math.atan2(s, ez @ zb_des)
        w_xy = 6.0 * ang * axis / s if s > 1e-12 else np.zeros(3)
        xb_i = q_rota(self.q, np.array([1.0, 0.0, 0.0]))
        yaw = math.atan2(xb_i[1], xb_i[0])
        self.w = np.array([w_xy[0], w_xy[1], 2.0 * (yaw_c - yaw)])
        self.w = np.clip(self.w, -1.5, 1.5)
        self.az = -thrust

    def _step_truth(self):
        n = 4
        h = self.dt / n
        ez = np.array([0.0, 0.0, 1.0])
        for _ in range(n):
            gB = q_rotp(self.q, G * ez)
            vxy = np.array([self.vel[0], self.vel[1], 0.0])
            vdot = self.az * ez + gB - np.cross(self.w, self.vel) - self.mu * vxy
            self.pos = self.pos + h * q_rota(self.q, self.vel)
            self.vel = self.vel + h * vdot
            self.q = q_otimes(self.q, q_exp(h * self.w))
            self.q /= np.linalg.norm(self.q)

    # -- sensors --------------------------------------------------------------------------------------------------------
    def imu(self, noise=True):
        """[accel, gyro] in the IMU frame: the filter rotates it back with q_b_u.rota (reference vi_ekf.cpp:265-267)"""
        acc_b = np.array([-self.mu * self.vel[0], -self.mu * self.vel[1], self.az]) + self.accel_bias_
        gyr_b = self.w + self.gyro_bias_
        if noise:
            acc_b = acc_b + self.rng.normal(0, self.accel_sigma, 3)
            gyr_b = gyr_b + self.rng.normal(0, self.gyro_sigma, 3)
        return np.concatenate([q_rotp(self.q_b_u, acc_b), q_rotp(self.q_b_u, gyr_b)])

    def project(self, ids=None):
        """-> (pixels [n,2], depth [n], visible [n]) of the landmarks `ids` (all if None), pinhole model of h_feat"""
        L = self.landmarks if ids is None else self.landmarks[np.asarray(ids, int)]
        pc = np.array([q_rotp(self.q_b_c, q_rotp(self.q, l - self.pos) - self.p_b_c) for l in L]).reshape(-1, 3)
        z = pc[:, 2]
        ok = z > 0.2
        zs = np.where(ok, z, 1.0)
        pix = np.stack([self.f[0] * pc[:, 0] / zs + self.c[0], self.f[1] * pc[:, 1] / zs + self.c[1]], axis=1)
        vis = ok & (pix[:, 0] > 15) & (pix[:, 0] < 625) & (pix[:, 1] > 15) & (pix[:, 1] < 465)
        return pix, np.linalg.norm(pc, axis=1), vis

    def _camera(self):
        pix, depth, vis = self.project()
        for i in self.tracked:
            if not vis[i]:
                del self.feat_id[i]              # a lost landmark comes back under a new id, like a real tracker
        self.tracked = [i for i in self.tracked if vis[i]]
        if len(self.tracked) < self.N:
            cand = [i for i in np.argsort(np.linalg.norm(pix - self.c, axis=1)) if vis[i] and i not in self.tracked]
            # spread the picks: every third candidate first, so that the features are not all at the image centre
            order = cand[::3] + cand[1::3] + cand[2::3]
            for i in order[: self.N - len(self.tracked)]:
                self.tracked.append(int(i))
                self.feat_id[int(i)] = self.next_feat_id
                self.next_feat_id += 1
        lm = list(self.tracked)
        z = pix[lm] + self.rng.normal(0, self.pix_sigma, (len(lm), 2))
        return z, [self.feat_id[i] for i in lm], depth[lm]

    # -- reference Simulator::run ----------------------------------------------------------------------------------------
    def run(self):
        """one IMU period: truth, then the IMU callback, then (at the camera rate) the feature callback"""
        if self.t >= self.tmax:
            return False
        self._control()
        self._step_truth()
        self.k += 1
        self.t = self.k * self.dt
        if self._imu_cb:
            self._imu_cb(self.t, self.imu(), self.R_imu)
        if self.k % self.cam_every == 0 and self._feat_cb:
            z, ids, depth = self._camera()
            self._feat_cb(self.t, z, ids, self.R_pix)
        return True
