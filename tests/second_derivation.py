"""A SECOND, independent derivation of the hot path in plain numpy / scipy — test infrastructure only.

Written against the text of the reference (balazs-bamer/cdpr-simulation, paths relative to src/cdpr_gazebo/), NOT
against oracle/cdpr_oracle.c: different language, different data structures, library solvers instead of hand-written
ones (np.linalg.lstsq for Eigen's QR, scipy's Rotation exponential map for the orientation update,
scipy.optimize.least_squares for the forward kinematics).  Its only job is to catch a shared misreading between
the HIP kernels and the C oracle, which were written by one author from one reading of the reference.

  RefTextPid           src/Pid.cpp:100-247   (reset, update, derive, fitPolynomial; filters bypassed: cascade 0)
  RefTextJointForce    src/JointForceCalculator.cpp:59-119 (mode machine, hold branch, Pid resets)
  SecondRobot          src/CdprGazeboPlugin.cpp:202-246 (update() ordering) + the reduced world step of SURVEY.md
                       8(a) rows 8-9 (IK from sdf/gen_cdpr.py:113-118's geometry statement, Joint::SetForce clamp
                       and damping from sdf/cube.sdf:438,442, semi-implicit Euler with the exact rotation update)
"""
import numpy as np
from scipy.optimize import least_squares
from scipy.spatial.transform import Rotation


class RefTextPid:
    """gazebo::common::Pid with both CascadeFilters at cascade 0 (identity)."""

    def __init__(self, kf, kp, ki, kd, degree, buflen, i_limit, cmd_limit, centred=False):
        self.centred = centred  # True: the same least-squares problem posed in centred time, in units of the sample spacing (well conditioned)
        self.kf, self.kp, self.ki, self.kd = kf, kp, ki, kd
        self.degree, self.buflen = int(degree), int(buflen)
        self.imax, self.imin = abs(i_limit), -abs(i_limit)      # Pid.cpp:70-71
        self.cmax, self.cmin = abs(cmd_limit), -abs(cmd_limit)  # Pid.cpp:72-73
        self.last_time = 0.0
        self.terms = {}
        self.reset()

    def reset(self):  # Pid.cpp:100-115
        self.was_last = False
        self.perr = self.ierr = self.derr = self.cmd = 0.0
        self.bx = [0.0] * self.buflen
        self.by = [0.0] * self.buflen
        self.missing = self.buflen

    def fit(self, shift=0.0, scale=1.0):  # Pid.cpp:219-247, absolute abscissae (shift 0, scale 1), pow(), normal equations
        d1, d2 = self.degree + 1, 2 * self.degree + 1
        bx = [(t - shift) / scale for t in self.bx]
        x = [sum(pow(t, i) for t in bx) for i in range(d2)]
        a = np.array([[x[i + j] for j in range(d1)] for i in range(d1)])
        b = np.array([sum(pow(t, i) * y for t, y in zip(bx, self.by)) for i in range(d1)])
        return np.linalg.lstsq(a, b, rcond=None)[0]  # stands in for colPivHouseholderQr().solve (Pid.cpp:246)

    def derive(self, value, now):  # Pid.cpp:193-217
        self.bx = self.bx[1:] + [now]
        self.by = self.by[1:] + [value]
        if self.missing > 0:
            self.missing -= 1
        derived = 0.0
        if self.missing == 0:
            shift = float(np.mean(self.bx)) if self.centred else 0.0
            scale = (self.bx[-1] - self.bx[0]) / (self.buflen - 1) if self.centred else 1.0  # sample spacing
            c = list(self.fit(shift, scale))
            dc = [(i + 1) * c[i + 1] for i in range(self.degree)] + [0.0]
            for i in range(self.degree, 0, -1):
                derived = (now - shift) / scale * (derived + dc[i])
            derived = (derived + dc[0]) / scale
        return derived

    def update(self, desired, actual, now):  # Pid.cpp:122-191
        if not self.was_last:
            self.was_last = True
            self.cmd = 0.0
        else:
            f_term = self.kf * desired
            error = desired - actual
            dt = now - self.last_time
            self.perr = error
            p_term = self.kp * self.perr
            prev_ierr = self.ierr
            self.ierr += dt * error
            i_term = self.ki * self.ierr
            self.terms["p"], self.terms["i"] = p_term, i_term  # pidMsg.axes[0..1]: I before the clamp
            if i_term > self.imax:
                i_term = self.imax
                self.ierr = i_term / self.ki
            elif i_term < self.imin:
                i_term = self.imin
                self.ierr = i_term / self.ki
            if dt > 0.0:
                self.derr = self.derive(error, now)
                self.terms["desired"] = desired
            d_term = self.kd * self.derr
            self.terms["d"] = d_term
            cmd = f_term + p_term + i_term + d_term
            if self.cmax > self.cmin:
                self.cmd = max(min(cmd, self.cmax), self.cmin)
            if self.cmd != cmd:
                self.ierr = prev_ierr
                self.cmd += dt * error * self.ki
        self.last_time = now
        return self.cmd


def pid_from_params(p, centred=False):
    """From a cdpr_pid_params_t (ctypes) or anything with the same attribute names."""
    return RefTextPid(p.forward_gain, p.p_gain, p.i_gain, p.d_gain, p.d_degree, p.d_buffer_length, p.i_limit, p.cmd_limit, centred)


class RefTextJointForce:
    """gazebo::physics::JointForceCalculator for one joint (JFC.cpp:59-119); Force mode left out (nothing in the
    plugin reaches setForce)."""

    POSITION, VELOCITY = 1, 2

    def __init__(self, pos_pid, vel_pid, eps):
        self.pos_pid, self.vel_pid, self.eps = pos_pid, vel_pid, eps
        self.mode = self.POSITION  # after Load: Position, target 0 (PLG.cpp:153-157, JFC.cpp:38-51)
        self.pos_target = self.vel_target = 0.0
        self.last_pos = 0.0
        self.last_update = 0.0

    def set_position_target(self, t):  # JFC.cpp:99-107
        self.pos_target = t
        if self.mode != self.POSITION:
            self.pos_pid.reset()
        self.mode = self.POSITION

    def set_velocity_target(self, t):  # JFC.cpp:111-119
        self.vel_target = t
        if self.mode != self.VELOCITY:
            self.vel_pid.reset()
        self.mode = self.VELOCITY

    def update(self, q, qd, now):  # JFC.cpp:59-96
        step = now - self.last_update
        self.last_update = now
        force = 0.0
        if step > 0:
            if self.mode == self.VELOCITY:
                if abs(self.vel_target) > self.eps:
                    self.last_pos = q
                    force = self.vel_pid.update(self.vel_target, qd, now)
                else:
                    force = self.pos_pid.update(self.last_pos, q, now)
            else:
                self.last_pos = q
                force = self.pos_pid.update(self.pos_target, q, now)
        return force


def ik(frame_anchors, platform_anchors, ref_lengths, p, rot, v, w):
    """Joint::Position / GetVelocity restated (gen_cdpr.py:113-118): returns q, qdot, unit vectors u, lever arms R b."""
    rb = rot.apply(platform_anchors)
    l = p + rb - frame_anchors
    length = np.linalg.norm(l, axis=1)
    u = l / length[:, None]
    q = ref_lengths - length                    # positive = cable shortening (gen_cdpr.py:181)
    qdot = -np.einsum("ij,ij->i", u, v + np.cross(w, rb))  # rate of the anchor-to-anchor distance, sign flipped
    return q, qdot, u, rb


def world_step(model, dt, gravity, p, rot, v, w, tension_axial, u, rb):
    """One semi-implicit Euler step of the free platform under the cable forces (each cable pulls its platform anchor
    towards its frame anchor with its axial tension) and gravity; rotation advanced by the exact exponential map."""
    m = model.mass
    ixx, iyy, izz, ixy, ixz, iyz = model.inertia
    ib = np.array([[ixx, ixy, ixz], [ixy, iyy, iyz], [ixz, iyz, izz]])
    forces = -tension_axial[:, None] * u
    f = forces.sum(axis=0) + m * np.asarray(gravity)
    tau = np.cross(rb, forces).sum(axis=0)
    r = rot.as_matrix()
    iw = r @ ib @ r.T
    v2 = v + dt * f / m
    w2 = w + dt * np.linalg.solve(iw, tau - np.cross(w, iw @ w))
    p2 = p + dt * v2
    rot2 = Rotation.from_rotvec(dt * w2) * rot
    return p2, rot2, v2, w2


def lumped_leg_terms(model, p, rot, v, w, frame_anchors, platform_anchors):
    """The lumped legs from ENERGY functions instead of force balances (the oracle and the kernels use closed-form
    forces and a closed-form mass matrix, HISTORY.md section 1): kinetic energy T(xi) and Rayleigh dissipation D(xi) of
    the leg links and passive joint dampers as quadratic functions of the platform twist xi = [v; omega] at the current
    pose; the added mass matrix is the Hessian of T, the damper wrench is -grad D, both by central differences (exact
    for quadratics).  Leg i turns about its frame anchor with angular velocity (u x vP)/L, vP = v + omega x rb."""
    rb = rot.apply(platform_anchors)
    l = p + rb - frame_anchors
    L = np.linalg.norm(l, axis=1)
    u = l / L[:, None]
    n = len(L)

    def rates(xi):
        vp = xi[:3] + np.cross(xi[3:], rb)
        w_leg = np.cross(u, vp) / L[:, None]
        return vp, w_leg

    def kinetic(xi):
        vp, w_leg = rates(xi)
        t = 0.5 * model.leg_inertia * (w_leg**2).sum()                       # virt_X, virt_Y, cable, virt_Ypf turn with the leg
        t += 0.5 * model.cable_axial_mass * (np.einsum("ij,ij->i", u, vp) ** 2).sum()   # cable link slides along the axis
        t += 0.5 * model.anchor_point_mass * (vp**2).sum()                   # virt_Xpf, virt_Ypf ride on the anchor
        t += 0.5 * n * model.anchor_inertia * (xi[3:] ** 2).sum()            # virt_Xpf turns with the platform
        return t

    def dissipation(xi):
        _, w_leg = rates(xi)
        # universal pair at the frame: the leg's own rate; spherical triple at the platform: leg relative to platform
        return 0.5 * model.passive_damping * ((w_leg**2).sum() + ((w_leg - xi[3:]) ** 2).sum())

    xi0 = np.concatenate([v, w])
    eye = np.eye(6)
    grad_d = np.array([(dissipation(xi0 + eye[k]) - dissipation(xi0 - eye[k])) / 2.0 for k in range(6)])
    hess_t = np.empty((6, 6))
    for a in range(6):
        for b in range(6):
            hess_t[a, b] = (kinetic(eye[a] + eye[b]) - kinetic(eye[a] - eye[b]) - kinetic(-eye[a] + eye[b]) + kinetic(-eye[a] - eye[b])) / 4.0
    return hess_t, -grad_d, rb


def world_step_lumped(model, dt, gravity, p, rot, v, w, tension_axial, u, rb, frame_anchors, platform_anchors):
    """world_step with the lumped legs: M(q) xi_dot = wrench, semi-implicit Euler, exact rotation update."""
    m = model.mass
    ixx, iyy, izz, ixy, ixz, iyz = model.inertia
    ib = np.array([[ixx, ixy, ixz], [ixy, iyy, iyz], [ixz, iyz, izz]])
    r = rot.as_matrix()
    iw = r @ ib @ r.T
    added, damper, _ = lumped_leg_terms(model, p, rot, v, w, frame_anchors, platform_anchors)
    mass = np.zeros((6, 6))
    mass[:3, :3] = m * np.eye(3)
    mass[3:, 3:] = iw
    mass += added
    forces = -tension_axial[:, None] * u
    g = np.asarray(gravity, dtype=np.float64)
    f = forces.sum(axis=0) + m * g + len(u) * model.anchor_point_mass * g
    tau = np.cross(rb, forces).sum(axis=0) + np.cross(rb, model.anchor_point_mass * g).sum(axis=0)
    i_tot = iw + len(u) * model.anchor_inertia * np.eye(3)
    rhs = np.concatenate([f, tau - np.cross(w, i_tot @ w)]) + damper
    acc = np.linalg.solve(mass, rhs)
    v2, w2 = v + dt * acc[:3], w + dt * acc[3:]
    return p + dt * v2, Rotation.from_rotvec(dt * w2) * rot, v2, w2


class SecondRobot:
    """One robot advanced the way CdprGazeboPlugin::update + the world step do it, on the pieces above."""

    def __init__(self, cfg, centred=False):
        s = cfg.to_struct()
        self.model, self.n, self.dt = cfg.model, cfg.n_cables, float(s.dt)
        self.fa = np.asarray(cfg.model.frame_anchors, dtype=np.float64)
        self.pb = np.asarray(cfg.model.platform_anchors, dtype=np.float64)
        self.l0 = np.array([s.cable_ref_length[i] for i in range(self.n)])
        self.gravity = np.array([s.gravity[i] for i in range(3)])
        self.damping, self.effort = float(s.joint_damping), float(s.effort_limit)
        hp = [s.home_pose[i] for i in range(7)]
        self.p, self.rot = np.array(hp[:3]), Rotation.from_quat(hp[3:])
        self.v, self.w = np.zeros(3), np.zeros(3)
        self.jfc = [RefTextJointForce(pid_from_params(s.position_pid, centred), pid_from_params(s.velocity_pid, centred), float(s.velocity_epsilon))
                    for _ in range(self.n)]
        self.pending_vel = self.pending_pos = None
        self.step = 0
        self.obs = None

    def set_state(self, pose7, twist6=None):
        self.p, self.rot = np.array(pose7[:3], dtype=np.float64), Rotation.from_quat(pose7[3:])
        if twist6 is not None:
            self.v, self.w = np.array(twist6[:3], dtype=np.float64), np.array(twist6[3:], dtype=np.float64)

    def set_velocity_command(self, axes):  # PLG.cpp:67-74: accepted iff axes.size() == n
        if len(axes) == self.n:
            self.pending_vel = [float(np.float32(a)) for a in axes]

    def set_position_command(self, axes):  # PLG.cpp:76-83
        if len(axes) == self.n:
            self.pending_pos = [float(np.float32(a)) for a in axes]

    def update(self, nsteps=1):
        for _ in range(nsteps):
            now_ns = self.step * int(round(self.dt * 1e9))
            now = (now_ns // 10**9) + (now_ns % 10**9) * 1e-9  # gazebo::common::Time::Double()
            if self.pending_vel is not None:  # PLG.cpp:206-212, velocity first
                for j, a in zip(self.jfc, self.pending_vel):
                    j.set_velocity_target(a)
                self.pending_vel = None
            if self.pending_pos is not None:  # PLG.cpp:213-219
                for j, a in zip(self.jfc, self.pending_pos):
                    j.set_position_target(a)
                self.pending_pos = None
            q, qd, u, rb = ik(self.fa, self.pb, self.l0, self.p, self.rot, self.v, self.w)
            raw = np.array([j.update(q[i], qd[i], now) for i, j in enumerate(self.jfc)])  # PLG.cpp:222-228
            applied = np.clip(raw, -self.effort, self.effort) if self.effort >= 0 else raw  # Joint::SetForce (cube.sdf:438)
            self.obs = dict(q=q, qd=qd, effort=applied, pose=np.concatenate([self.p, self.rot.as_quat()]),
                            twist=np.concatenate([self.v, self.w]))
            axial = applied - self.damping * qd  # explicit joint damping (cube.sdf:442)
            m = self.model
            if m.passive_damping or m.leg_inertia or m.cable_axial_mass or m.anchor_point_mass or m.anchor_inertia:
                self.p, self.rot, self.v, self.w = world_step_lumped(m, self.dt, self.gravity, self.p, self.rot, self.v, self.w, axial, u, rb,
                                                                     self.fa, self.pb)
            else:
                self.p, self.rot, self.v, self.w = world_step(m, self.dt, self.gravity, self.p, self.rot, self.v, self.w, axial, u, rb)
            self.step += 1


def fk_least_squares(frame_anchors, platform_anchors, lengths, seed_p, seed_rot):
    """Forward kinematics as a generic nonlinear least-squares problem on the length residual (scipy trust region),
    pose parametrised as position + rotation vector on top of the seed orientation."""
    def residual(x):
        rot = Rotation.from_rotvec(x[3:]) * seed_rot
        return np.linalg.norm(x[:3] + rot.apply(platform_anchors) - frame_anchors, axis=1) - lengths

    sol = least_squares(residual, np.concatenate([seed_p, np.zeros(3)]), xtol=1e-15, ftol=1e-15, gtol=1e-15)
    return sol.x[:3], Rotation.from_rotvec(sol.x[3:]) * seed_rot, float(np.abs(sol.fun).max())
