"""TEST INFRASTRUCTURE (parity unpinned): the host plumbing of vi_ekf::VIEKF around ONE
oracle filter -- state ring, input deque, measurement queue with rewind / replay -- restated line by line from the
reference:

  propagate_state bookkeeping    src/vi_ekf/vi_ekf.cpp:262-318
  add_measurement                src/vi_ekf/vi_ekf_meas.cpp:130-194
  handle_measurements            src/vi_ekf/vi_ekf_meas.cpp:6-127
  keep_only_features             src/vi_ekf/vi_ekf_feat.cpp:81-142
  ring sizes                     include/vi_ekf.h:50-51

Only tests/ import this.  Pure Python over oracle.OracleFilter: meant for the small functional configuration (SURVEY 8d,
config 1).  Quirks kept: the input deque stores the ROTATED input and the replay rotates it again (vi_ekf.cpp:265-271 with
vi_ekf_meas.cpp:78,102,111,118); a measurement newer than the newest input is left in the queue (:24-28); the rewind picks
the newest ring slot with t <= the input before the measurement (:45-56); init_feature numbers features itself
(vi_ekf_feat.cpp:29-30).
"""
import math
from collections import deque

import numpy as np

from . import oracle as orc

LEN_STATE_HIST = 250
LEN_MEAS_HIST = 200


class _Meas:
    __slots__ = ("t", "type", "z", "R", "active", "id", "depth", "handled")


# ---- SE(3) algebra of the global node frame (vi_ekf_kfr.cpp:14-53,147-150).  The reference's Xformd comes from its absent
# `geometry` submodule; the convention is the one stated in include/viekf.h: T = {t, q}, q passive (src/quat.cpp),
#   T1 * T2 = {t1 + q1.rota(t2), q1 (x) q2},   Adj(T) = [[R, [t]x R], [0, R]],  R = q.R().
# Written with rotation matrices (oracle/np_twin.py), not a transliteration of the product's quaternion code.
def xform_mul(T1, T2):
    from oracle import np_twin as tw
    T1, T2 = np.asarray(T1, float), np.asarray(T2, float)
    return np.concatenate([T1[:3] + tw.Rmat(T1[3:]).T @ T2[:3], tw.qmul(T1[3:], T2[3:])])


def xform_adj(T):
    from oracle import np_twin as tw
    T = np.asarray(T, float)
    R = tw.Rmat(T[3:])
    A = np.zeros((6, 6))
    A[:3, :3] = R
    A[:3, 3:] = tw.skew(T[:3]) @ R
    A[3:, 3:] = R
    return A


class SeqOracle:
    def __init__(self, filt, keyframe_overlap_threshold=0.8, state_hist=LEN_STATE_HIST, meas_hist=LEN_MEAS_HIST):
        self.f = filt
        self.H = int(state_hist)
        self.MH = int(meas_hist)
        self.xr = [None] * self.H
        self.Pr = [None] * self.H
        self.lr = [0] * self.H              # len_features per slot is NOT in the reference ring (x_, P_, t_ only) -- see _load
        self.t = [math.nan] * self.H
        self.i = 0
        self.u = deque()                    # (t, ub) newest first
        self.zbuf = deque()                 # newest first
        self.start_t = math.nan
        self.kf_thresh = float(keyframe_overlap_threshold)
        self.keyframe_features = []
        self.log = []                       # what would go to cerr
        self.keyframe_edges = []
        self.node = np.array([0.0, 0, 0, 1, 0, 0, 0])      # current_node_global_pose_ = Identity (vi_ekf.cpp:38)
        self.node_cov = np.zeros((6, 6))                    # global_pose_cov_ (vi_ekf.cpp:39)
        self.rec = None                     # what the reference's log writer would record (enable_records)
        self._save()

    # -- vi_ekf_log.cpp:6-67: the records of log_state / log_measurement, kept as Python lists instead of files ----------
    def enable_records(self):
        self.rec = {"state": [], "meas": {}}
        self._rec_state(0.0, np.zeros(6), np.zeros(self.f.n))        # the constructor's first record, vi_ekf.cpp:154

    def _rec_state(self, t, ub, xdot):
        ids = list(self.f.feature_ids)
        idv = np.array([float(ids[i]) if i < len(ids) else -1.0 for i in range(self.f.N)])
        self.rec["state"].append(dict(t=t, x=self.f.x.copy(), Pd=np.diag(self.f.P).copy(), u=np.asarray(ub).copy(),
                                      xdot=np.asarray(xdot).copy(), ids=idv, gpose=self.get_global_pose()))

    # -- vi_ekf_kfr.cpp:14-53 -------------------------------------------------------------------------------------
    def get_global_pose(self):
        x = self.f.x
        return xform_mul(self.node, np.concatenate([x[0:3], x[6:10]]))

    def get_global_cov(self):
        P = self.f.P
        idx = [0, 1, 2, 6, 7, 8]                            # the POS and ATT blocks (vi_ekf_kfr.cpp:28-31)
        A = xform_adj(self.node)
        return self.node_cov + A.T @ P[np.ix_(idx, idx)] @ A

    def _node_update(self, edge):
        """end of keyframe_reset (vi_ekf_kfr.cpp:147-150): covariance first, with the node pose before it moves"""
        C = np.zeros((6, 6))
        C[:3, :3] = np.asarray(edge[7:16]).reshape(3, 3, order="F")
        C[5, 5] = edge[16]
        A = xform_adj(self.node)
        self.node_cov = self.node_cov + A.T @ C @ A
        self.node = xform_mul(self.node, edge[:7])

    # -- ring <-> live filter -------------------------------------------------------------------------------------
    def _save(self):
        self.xr[self.i] = self.f.x.copy()
        self.Pr[self.i] = self.f.P.copy()

    def _load(self):
        # the reference ring holds x and P only: feature bookkeeping (len_features_, ids) is not rewound
        self.f.x[:] = self.xr[self.i]
        self.f.P[:] = self.Pr[self.i]

    def _rot(self, u):
        q = np.array(self.f.q_b_u)
        u = np.asarray(u, dtype=np.float64)
        return np.concatenate([orc.q_rota(q, u[0:3]), orc.q_rota(q, u[3:6])])

    # -- vi_ekf.cpp:262-318 -------------------------------------------------------------------------------------------
    def propagate_state(self, u, t, save_input=True):
        ub = self._rot(u)                                   # :265-267
        if save_input:
            self.u.appendleft((t, ub))                      # :269-272  (the ROTATED input is stored)
        if math.isnan(self.start_t):                        # :274-279
            self.start_t = t
            self.t[self.i] = t
            return
        dt = t - self.t[self.i]
        if abs(dt) < 1e-6:                                  # :281-283
            return
        if dt < 0:                                          # :285-289
            self.log.append("propagate backwards")
            return
        self._save()                                        # (x_[i_], P_[i_] stay behind as history)
        xdot = self.f.dynamics(self.f.x.copy(), ub)[0] if (save_input and self.rec is not None) else None   # dx_ of :293
        self.f.propagate(u, dt)                             # :295-304 (+ fix_depth :311); rotates u itself, like the reference
        ip = (self.i + 1) % self.H
        self.t[ip] = t
        self.i = ip
        self._save()
        if xdot is not None:                                # :316-317
            self._rec_state(t, ub, xdot)

    # -- vi_ekf_meas.cpp:130-194 ----------------------------------------------------------------------------------------
    def add_measurement(self, t, z, mtype, R, active=False, id=-1, depth=math.nan):
        z = np.atleast_1d(np.asarray(z, dtype=np.float64))
        if t < self.start_t:                                # :133-134 (false while start_t is NaN)
            return orc.MEAS_INVALID
        if np.isnan(z).any():                               # :136-137
            return orc.MEAS_NAN
        if mtype == orc.FEAT and id >= 0:                   # :140-147
            if self.f.global_to_local_feature_id(id) < 0:
                self.f.init_feature(z, id, depth)
                self._save()
                return orc.MEAS_NEW_FEATURE
        k = 0                                               # :150-156 first entry with z.t < t
        while k < len(self.zbuf) and not (self.zbuf[k].t < t):
            k += 1
        m = _Meas()
        m.t, m.type, m.z, m.R = t, mtype, z.copy(), np.atleast_2d(np.asarray(R, dtype=np.float64)).copy()
        m.active, m.id, m.depth, m.handled = bool(active), int(id), depth, False
        self.zbuf.insert(k, m)                              # :169-175
        return orc.MEAS_SUCCESS

    def _update(self, m):
        m.handled = True                                    # :198
        zhat = self.f.h(m.type, None, m.id)[0] if self.rec is not None else None
        res = self.f.update(m.type, m.z, m.R, m.active, m.id)
        if self.rec is not None and res == orc.MEAS_SUCCESS:   # log_measurement sits at the end of update(), :276
            r = [self.t[self.i] - self.start_t] + list(m.z) + list(zhat[:m.z.size]) + [1.0 if m.active else 0.0]
            if m.type in (orc.FEAT, orc.QZETA, orc.DEPTH, orc.INV_DEPTH):
                r.append(float(m.id))
            self.rec["meas"].setdefault(m.type, []).append(np.array(r))
        return res

    # -- vi_ekf_meas.cpp:6-127 --------------------------------------------------------------------------------------------
    def handle_measurements(self):
        gated = []
        if len(self.zbuf) == 0:                             # :12-13
            return gated
        zi = len(self.zbuf) - 1                             # :16-18 oldest unhandled
        while self.zbuf[zi].handled and zi != 0:
            zi -= 1
        if zi == 0 and self.zbuf[zi].handled:               # :21-22
            return gated
        if self.zbuf[zi].t > self.u[0][0]:                  # :24-28 measurement from the future
            return gated
        ui = 0                                              # :32-38 input just before the measurement
        while ui != len(self.u):
            if self.zbuf[zi].t > self.u[ui][0]:
                break
            ui += 1
        if ui == len(self.u) or self.zbuf[zi].t <= self.u[ui][0]:   # :39-43 (u_.end() dereference guarded here)
            self.log.append("not enough history in input buffer")
            return gated
        i = self.H                                          # :46-57 rewind
        while i > 0:
            if self.t[(self.i + i) % self.H] <= self.u[ui][0]:
                self.i = (self.i + i) % self.H
                break
            i -= 1
        if i == 0:                                          # :59-64
            self.log.append("not enough history in state buffer")
            del self.zbuf[zi]
            return gated
        self._load()
        if self.t[self.i] > self.zbuf[zi].t or abs(self.t[self.i] - self.u[ui][0]) > 1e-8 or self.u[ui][0] > self.zbuf[zi].t:
            self.log.append("time history misaligned")      # :67-70
        ui -= 1                                             # :74
        while ui != 0:                                      # :75
            while self.zbuf[zi].t <= self.u[ui][0]:         # :78
                z = self.zbuf[zi]
                if self.t[self.i] < z.t:                    # :81-82
                    self.propagate_state(self.u[ui][1], z.t, False)
                elif self.t[self.i] > z.t:
                    self.log.append("can't propagate backwards")
                if z.handled:                               # :87-95
                    self.log.append("trying to handle measurement again")
                else:
                    res = self._update(z)
                    self._save()
                    if res == orc.MEAS_GATED and z.type == orc.FEAT:
                        gated.append(z.id)
                if zi != 0:                                 # :97-105
                    zi -= 1
                    while self.u[ui][0] < self.zbuf[zi].t and ui != 0:
                        self.propagate_state(self.u[ui][1], self.u[ui][0], False)
                        ui -= 1
                else:                                       # :106-115
                    while ui != 0:
                        self.propagate_state(self.u[ui][1], self.u[ui][0], False)
                        ui -= 1
                    break
            else:
                # the reference's outer `while (u_it != u_.begin())` spins here when the next measurement is newer than this
                # input; with well-formed queues the inner loop always consumes inputs, so leave as the reference would
                # after the inner propagates
                break
        self.propagate_state(self.u[ui][1], self.u[ui][0], False)   # :118
        while len(self.zbuf) > self.MH:                     # :121-122
            self.zbuf.pop()
        while len(self.u) > self.H:                         # :125-126
            self.u.pop()
        return gated

    # -- setters and the explicit keyframe reset that callers use (include/vi_ekf.h:288-292,324) ----------------------------
    def set_x0(self, x0):                                   # vi_ekf.cpp:157-160: x_[i_].topRows(xZ) = x0
        self.f.x[:17] = np.asarray(x0, dtype=np.float64)
        self._save()

    def set_imu_bias(self, b_g, b_a):                       # vi_ekf.cpp:179-183
        self.f.x[13:16] = np.asarray(b_g, dtype=np.float64)
        self.f.x[10:13] = np.asarray(b_a, dtype=np.float64)
        self._save()

    def set_drag_term(self, on):                            # include/vi_ekf.h:290
        self.f.set_drag_term(bool(on))

    def keyframe_reset(self):                               # vi_ekf_kfr.cpp:56-157 (state, N P N^T, the node frame's move)
        self.keyframe_edges.append(self.f.keyframe_reset_edge())
        self._node_update(self.keyframe_edges[-1])
        self._save()

    def clear_feature(self, gid):                           # vi_ekf_feat.cpp:50-73
        self.f.clear_feature(int(gid))
        self._save()

    # -- vi_ekf_feat.cpp:81-142 -----------------------------------------------------------------------------------------
    def keep_only_features(self, features):
        features = [int(v) for v in features]
        ids = list(self.f.feature_ids)
        remove, overlap = [], 0
        for gid in ids:
            if gid in features:
                if self.f.use_keyframe_reset and gid in self.keyframe_features:
                    overlap += 1
            else:
                remove.append(gid)
        for gid in remove:
            self.f.clear_feature(gid)
        if self.f.use_keyframe_reset and len(self.keyframe_features) > 0 and \
                overlap / float(len(self.keyframe_features)) < self.kf_thresh:
            self.keyframe_edges.append(self.f.keyframe_reset_edge())
            self._node_update(self.keyframe_edges[-1])
            self.keyframe_features = list(features)
        elif self.f.use_keyframe_reset and len(self.keyframe_features) == 0:
            self.keyframe_features = list(features)
        self._save()
