"""Scene compiler: flattened bodies + addon ops -> the flat "scene blob".

The blob layout is defined in ``include/diygym_scene.h``; the constants are read
from that header at import time so Python and C cannot drift apart.  This is the
host half of what the reference does through ``p.loadURDF`` /
``p.setPhysicsEngineParameter`` / ``p.setGravity`` while constructing an
environment (reference: diy_gym/diy_gym.py:74-91, diy_gym/model.py:53-83).
"""
import os
import re

import numpy as np

from .mathx import Transform, quat_from_mat
from .urdf import SHAPE_BOX, SHAPE_CAPSULE, SHAPE_POINTS, SHAPE_SPHERE

_HEADER = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'include', 'diygym_scene.h')
_HEADER_PKG = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'csrc', 'diygym_scene.h')


def _parse_header(path):
    """Evaluate the enums / #defines of diygym_scene.h into a dict."""
    text = open(path).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    consts = {}
    for m in re.finditer(r'#define\s+(DG_\w+)\s+(\S+)', text):
        consts[m.group(1)] = int(m.group(2), 0)
    for m in re.finditer(r'enum\s*\{(.*?)\}', text, flags=re.S):
        value = -1
        for item in m.group(1).split(','):
            item = item.strip()
            if not item:
                continue
            if '=' in item:
                name, expr = (t.strip() for t in item.split('=', 1))
                value = int(eval(expr, {}, consts))
            else:
                name, value = item, value + 1
            consts[name] = value
    return consts


C = _parse_header(_HEADER if os.path.isfile(_HEADER) else _HEADER_PKG)


class K:
    """Namespace holding the DG_* constants without the prefix."""


for _k, _v in C.items():
    setattr(K, _k[3:], _v)

# Solver / engine defaults.  Values tagged [R] are pybullet defaults restated from
# recollection (DESIGN.md "Restated pybullet behaviours").
DEFAULTS = dict(
    residual_threshold=1e-7,  # [R] solverResidualThreshold
    contact_erp=0.08,  # [R] m_erp2 set by pybullet's world setup
    limit_erp=0.08,
    linear_slop=1e-5,  # [R]
    linear_damping=0.04,  # [R] btMultiBody default
    angular_damping=0.04,  # [R]
    max_coordinate_velocity=100.0,  # [R]
    default_motor_impulse=1.0,  # [R] velocity motor created for every joint at load
    ik_iterations=20,  # [R] maxNumIterations
    ik_lambda_sq=0.36,  # [R] Jacobian::DefaultDampingLambda = 0.6, squared
    ik_joint_damping=0.1,  # [R] jointDamping default
    ik_residual=1e-4,  # [R] residualThreshold
    ik_max_angle=np.radians(45.0),  # [R] MaxAngleDLS
    ik_null_rest_gain=0.001,  # [R] stayCloseToZeroGain
    ik_null_limit_gain=10.0,  # [R] stayAwayFromLimitsGain
    contact_margin=0.02,  # [R] contact breaking threshold
    # contact warm starting: a contact that persists from one substep to the next (same candidate pair, same feature)
    # starts its normal row from `warmstart` x the impulse it ended with, its friction rows from `warmstart_friction` x
    # theirs.  Bullet: m_warmstartingFactor 0.85 on the normal row, friction rows from zero [R]; 0 switches it off.
    # Default here: the whole normal impulse (a resting stack then needs 26 sweeps where 0.85 needs 109 and a cold start
    # 150+; the trajectories of the three agree to well below their distance from the converged solution),
    # friction from zero (a warm-started friction row keeps whatever split of a statically indeterminate support it
    # landed with -- tests/test_oracle_kat.py::test_force_torque_sensor_sees_contact_forces_on_the_child_side).
    warmstart=1.0,
    warmstart_friction=0.0,
    # motor rows start from the clamped direct solution of their body's unclamped motor system instead of zero (same
    # fixed point; 35 -> 6 sweeps for ur_high_5, 26 -> 1 for from_the_readme); 0 = Bullet's cold start [R]
    motor_guess=1.0,
    # with motor_guess: a joint whose motor target lies beyond an ACTIVE joint-limit row starts with its motor saturated into
    # the limit and the limit row holding the balance (one unknown in the guess's linear system) -- DG_HF_LIMIT_GUESS; 0 = off
    limit_guess=1.0,
    # time base of a motor row's impulse bound: 'substep' (max force x fixedTimeStep / numSubSteps) or 'step' (max force x
    # fixedTimeStep: the other reading of pybullet's maxAppliedImpulse [R]) -- DG_HF_MOTOR_IMPULSE_SCALE
    motor_impulse_timebase='substep',
    # two convex hulls (URDF collision meshes; boxes on moving bodies) collide as hulls: GJK closest points, an expanding polytope
    # for the depth once they overlap, one contact per pair -- what Bullet's btConvexConvexAlgorithm does per call [R]; 0 = through
    # the capsule fitted to each hull (the narrow phase of rounds 1-3) -- DG_HF_HULL_CONTACTS
    hull_contacts=1.0,
    hull_margin=0.001,  # [R] gUrdfDefaultCollisionMargin: a hull is inflated by this radius -- DG_HF_HULL_MARGIN
)


def fit_capsule(points):
    """Bounding capsule of a vertex cloud along its principal axis: (Transform, r, half_len)."""
    pts = np.asarray(points, dtype=np.float64)
    mean = pts.mean(axis=0)
    cov = np.cov((pts - mean).T) if len(pts) > 2 else np.eye(3)
    w, v = np.linalg.eigh(cov)
    axis = v[:, np.argmax(w)]
    if axis[np.argmax(np.abs(axis))] < 0:
        axis = -axis
    t = (pts - mean) @ axis
    perp = (pts - mean) - np.outer(t, axis)
    r = float(np.max(np.linalg.norm(perp, axis=1)))
    tmin, tmax = float(t.min()), float(t.max())
    half = max(0.5 * (tmax - tmin) - r, 0.0)
    centre = mean + axis * 0.5 * (tmin + tmax)
    # frame with z = axis
    helper = np.array([1.0, 0.0, 0.0]) if abs(axis[0]) < 0.9 else np.array([0.0, 1.0, 0.0])
    x = np.cross(helper, axis)
    x /= np.linalg.norm(x)
    y = np.cross(axis, x)
    R = np.stack([x, y, axis], axis=1)
    return Transform(R, centre), max(r, 1e-4), half


def as_box(points, tol=1e-6):
    """If the 8 points are the corners of a box return (Transform, half_extents), else None."""
    pts = np.asarray(points, dtype=np.float64)
    if pts.shape != (8, 3):
        return None
    c = pts.mean(axis=0)
    d = pts - c
    # the three edge directions at corner 0: differences to the three nearest corners
    dist = np.linalg.norm(pts - pts[0], axis=1)
    order = np.argsort(dist)[1:4]
    e = pts[order] - pts[0]
    lens = np.linalg.norm(e, axis=1)
    if np.any(lens < tol):
        return None
    u = e / lens[:, None]
    if abs(u[0] @ u[1]) > 1e-6 or abs(u[0] @ u[2]) > 1e-6 or abs(u[1] @ u[2]) > 1e-6:
        return None
    if np.linalg.det(u) < 0:
        u[2] = -u[2]
    R = u.T
    local = d @ R
    half = 0.5 * lens
    scale = max(1.0, float(np.abs(local).max()))
    if not np.allclose(np.abs(local), half[None, :], atol=1e-6 * scale):
        return None
    return Transform(R, c), half


def box_corners(T, half):
    s = np.array([[sx, sy, sz] for sx in (-1, 1) for sy in (-1, 1) for sz in (-1, 1)], dtype=np.float64)
    return (s * half[None, :]) @ T.R.T + T.p


def hull_planes(points):
    """Unique face planes (n, d) of the convex hull, n.x + d <= 0 inside."""
    pts = np.asarray(points, dtype=np.float64)
    try:
        from scipy.spatial import ConvexHull
        eq = ConvexHull(pts).equations
    except Exception:
        return np.zeros((0, 4))
    keys, out = set(), []
    for row in eq:
        k = tuple(np.round(row, 6))
        if k not in keys:
            keys.add(k)
            out.append(row)
    return np.asarray(out, dtype=np.float64)


def shape_bound(sh):
    """Radius of a sphere around the shape frame origin (or the hull points' centroid frame) containing the shape."""
    if sh.kind == SHAPE_SPHERE:
        return float(sh.params[0])
    if sh.kind == SHAPE_BOX:
        return float(np.linalg.norm(sh.params))
    if sh.kind == SHAPE_CAPSULE:
        return float(sh.params[0] + sh.params[1])
    return 0.0


def body_bound(flat):
    """Conservative radius around the base origin that contains every collision shape of the body whatever the
    joint angles: joint offsets (and prismatic travel) add up along the chain."""
    reach = []
    for fl in flat.links:
        r = (reach[fl.parent] if fl.parent >= 0 else 0.0) + float(np.linalg.norm(fl.origin.p))
        if fl.joint_type == 1 and fl.lower <= fl.upper:
            r += max(abs(fl.lower), abs(fl.upper))
        reach.append(r)
    out = 0.0
    for sh in flat.shapes:
        base = reach[sh.link] if sh.link >= 0 else 0.0
        if sh.kind == SHAPE_POINTS:
            ext = float(np.max(np.linalg.norm(np.asarray(sh.points), axis=1)))
        else:
            ext = float(np.linalg.norm(sh.T.p)) + shape_bound(sh)
        out = max(out, base + ext)
    return out


class OpHandle:
    def __init__(self, index, kind, io_off, io_dim, state_off, slot):
        self.index = index
        self.kind = kind
        self.io_off = io_off
        self.io_dim = io_dim
        self.state_off = state_off
        self.slot = slot


class SceneBuilder:
    def __init__(self, timestep=1.0 / 240.0, substeps=2, solver_iterations=150, gravity=(0.0, 0.0, -9.81),
                 max_episode_steps=None, hot_start=1, rew_mode=0, term_mode=0, max_contacts=None, **overrides):
        self.timestep = float(timestep)
        self.substeps = max(int(substeps), 1)
        self.solver_iterations = int(solver_iterations)
        self.gravity = [float(g) for g in gravity]
        self.max_episode_steps = -1 if max_episode_steps is None else int(max_episode_steps)
        self.hot_start = int(hot_start)
        self.rew_mode = rew_mode
        self.term_mode = term_mode
        self.max_contacts = max_contacts
        self.params = dict(DEFAULTS)
        for k, v in overrides.items():
            if k not in self.params:
                raise KeyError('unknown engine parameter: ' + k)
            self.params[k] = v
        self.bodies = []
        self.aliases = {}  # (FlatBody, pos, quat)
        self.colors = []      # per body: the YAML `color` (reference model.py:82-83), default grey
        self.color_set = []   # ... whether the config gave one: it then overrides the base link's URDF material, like changeVisualShape(uid, -1)
        self.constraints = []  # ([body_a, local link_a, body_b, local link_b], [pivot_a pos quat, pivot_b pos quat, max force, 0])
        self.cameras = []  # (body, frame, width, height, flags, Transform, fov, near, far)
        self.ops = []
        self.ilist = []
        self.flist = []
        self.dims = {'act': 0, 'obs': 0, 'rew': 0, 'term': 0}
        self.addon_state = 0
        self.n_slots = 0
        self.term_groups = []

    # -- bodies ---------------------------------------------------------
    ALIAS_BASE = 10000  # uids of attached child models (they own no body of their own)

    def resolve(self, uid):
        """``uid`` -> (body index, link offset, frame offset, frame that stands for the base or -1)."""
        return self.aliases[uid] if uid in self.aliases else (uid, 0, 0, -1)

    def attach_child(self, parent_uid, parent_frame, child_flat, pos, quat):
        """Child model (reference model.py:69-77): merged rigidly into the parent's body (FlatBody.attach)."""
        body, _, foff, basef = self.resolve(parent_uid)
        pf = (basef if parent_frame < 0 else foff + parent_frame)
        link_off, frame_off, base_frame = self.bodies[body][0].attach(child_flat, pf, Transform.from_xyz_quat(pos, quat))
        uid = self.ALIAS_BASE + len(self.aliases)
        self.aliases[uid] = (body, link_off, frame_off, base_frame)
        return uid

    def add_constrained_child(self, parent_uid, parent_frame, child_flat, pos, quat, max_force):
        """Child model coupled to its parent the way the reference does it (model.py:69-77): a body of its own, held by
        ``createConstraint(JOINT_FIXED)`` -- here six solver rows (``DG_KI_*`` / ``DG_KF_*``).  ``pos`` / ``quat``: the pivot in
        the parent link's INERTIAL frame; the pivot on the child is its base inertial frame [RECOLLECTION: the frames
        createConstraint takes its pivots in].  [decision] The child is loaded AT the pivot for the parent's load
        configuration (all joints at zero); the reference loads it at the parent frame and lets the constraint pull it
        over.  The two bodies do not collide with each other."""
        body, _, foff, basef = self.resolve(parent_uid)
        flat, p_link, q_link = self.bodies[body]
        pf = basef if parent_frame < 0 else foff + parent_frame
        anchor, T_anchor_pf = (flat.frames[pf].link, flat.frames[pf].T_com) if pf >= 0 else (-1, flat.T_base_report)
        T = Transform()  # anchor link frame in the base link frame, joints at zero
        k = anchor
        while k >= 0:
            T = flat.links[k].origin * T
            k = flat.links[k].parent
        T_rel = Transform.from_xyz_quat(pos, quat)
        T_world = Transform.from_xyz_quat(p_link, q_link) * T * T_anchor_pf * T_rel
        child = self.add_body(child_flat, T_world.p, T_world.quat)
        T_a = T_anchor_pf * T_rel  # pivot in the anchor's link frame
        T_b = child_flat.T_base_report  # the child's base inertial frame in its base link frame
        self.constraints.append(([body, -1 if anchor < 0 else anchor, child, -1], [*T_a.p, *T_a.quat, *T_b.p, *T_b.quat, float(max_force), 0.0]))
        return child

    def add_body(self, flat, pos, quat):
        """``pos``/``quat``: pose of the root inertial frame, as passed to
        ``p.resetBasePositionAndOrientation`` (reference model.py:68)."""
        q_link = quat_from_mat(Transform.from_xyz_quat([0, 0, 0], quat).R @ flat.T_base_report.R.T)
        T_link = Transform.from_xyz_quat([0, 0, 0], q_link)
        p_link = np.asarray(pos, dtype=np.float64) - T_link.R @ flat.T_base_report.p
        self.bodies.append((flat, p_link, q_link))
        self.colors.append([0.8, 0.8, 0.8, 1.0])
        self.color_set.append(False)
        return len(self.bodies) - 1

    def set_color(self, body, rgba):
        if body in self.aliases:
            return  # colours are per body; an attached child keeps its parent's
        self.colors[body] = [float(v) for v in rgba]
        self.color_set[body] = True

    def add_camera(self, body, frame, width, height, flags, T_parent_cam, fov, near, far):
        self.cameras.append((self.resolve(body)[0] if body >= 0 else body, self.global_frame(body, frame) if body >= 0 else -1, int(width), int(height), int(flags),
                             T_parent_cam, float(fov), float(near), float(far)))
        return len(self.cameras) - 1

    def prev_velocity_slots(self, body):
        """Addon-state floats a force_torque_sensor on ``body`` needs for the body's velocities at the start of the last
        substep; 0 when an earlier sensor on the same body already owns them."""
        b = self.resolve(body)[0]
        if any(row[K.OI_CODE] == K.OP_OBS_FT and row[K.OI_BODY] == b for row, _ in self.ops):
            return 0
        flat = self.bodies[b][0]
        return len(flat.links) + (0 if flat.fixed_base else 6)

    def shapes_of(self, body, urdf_links, moving_dofs):
        """Global indices of the shapes of ``body`` that belong to the given pybullet link indices or sit on one of the
        given moving links (local DoF indices)."""
        b = self.resolve(body)[0]
        base = sum(len(x[0].shapes) for x in self.bodies[:b])
        flat = self.bodies[b][0]
        return [base + i for i, sh in enumerate(flat.shapes) if sh.urdf_link in urdf_links or sh.link in moving_dofs]

    def link_base(self, body):
        return sum(len(b[0].links) for b in self.bodies[:body])

    def frame_base(self, body):
        return sum(len(b[0].frames) for b in self.bodies[:body])

    def global_link(self, body, dof):
        b, loff, _, _ = self.resolve(body)
        return self.link_base(b) + loff + dof

    def global_frame(self, body, frame_id):
        b, _, foff, basef = self.resolve(body)
        if frame_id < 0:
            return -1 if basef < 0 else self.frame_base(b) + basef
        return self.frame_base(b) + foff + frame_id

    # -- ops --------------------------------------------------------------
    def add_op(self, code, kind, body=-1, frame=-1, body2=-1, frame2=-1, flags=0, ilist=(), flist=(), fparams=(),
               io_dim=0, state_dim=0, group=None, n=None):
        """``kind`` in act|obs|rew|term|reset.  Returns an :class:`OpHandle`."""
        io_off = 0
        if kind in self.dims:
            io_off = self.dims[kind]
            self.dims[kind] += io_dim
        state_off = self.addon_state
        self.addon_state += state_dim
        slot = 0
        if kind == 'act':
            slot = self.n_slots
            self.n_slots += 1
            if self.n_slots > 64:
                raise ValueError('more than 64 controller addons in one environment')
        if kind == 'term':
            if group not in self.term_groups:
                if len(self.term_groups) >= 64:  # terminal_if_all folds the groups into a 64-bit mask on the device
                    raise ValueError('more than 64 receptors with terminal addons in one environment')
                self.term_groups.append(group)
            slot = self.term_groups.index(group)
        ioff, foff = len(self.ilist), len(self.flist)
        self.ilist.extend(int(v) for v in ilist)
        self.flist.extend(float(v) for v in flist)
        fp = [float(v) for v in fparams]
        if len(fp) > K.OF_STRIDE:
            raise ValueError('too many float parameters for one op')
        fp = fp + [0.0] * (K.OF_STRIDE - len(fp))
        row = [0] * K.OI_STRIDE
        row[K.OI_CODE] = code
        if body in self.aliases and code in (K.OP_RESPAWN, K.OP_EXTERNAL_FORCE, K.OP_PROPELLOR, K.OP_REW_ELECTRICITY, K.OP_TERM_TILT):
            raise NotImplementedError('this addon acts on a whole body; an attached child model has none of its own')
        row[K.OI_BODY] = self.resolve(body)[0] if body >= 0 else body
        row[K.OI_FRAME] = self.global_frame(body, frame) if body >= 0 else -1
        row[K.OI_BODY2] = self.resolve(body2)[0] if body2 >= 0 else body2
        row[K.OI_FRAME2] = self.global_frame(body2, frame2) if body2 >= 0 else -1
        row[K.OI_FLAGS] = flags
        row[K.OI_N] = len(ilist) if n is None else int(n)
        row[K.OI_ILIST] = ioff
        row[K.OI_FLIST] = foff
        row[K.OI_IO_OFF] = io_off
        row[K.OI_STATE_OFF] = state_off
        row[K.OI_SLOT] = slot
        self.ops.append((row, fp))
        return OpHandle(len(self.ops) - 1, kind, io_off, io_dim, state_off, slot)

    # -- emit -----------------------------------------------------------
    def finalize(self):
        nb = len(self.bodies)
        body_i, body_f, link_i, link_f, frame_i, frame_f = [], [], [], [], [], []
        shape_i, shape_f, points, planes = [], [], [], []
        shape_dyn, shape_anchor = [], []
        state_off = K.ST_PREFIX
        link_state_offs = []
        gl = 0
        respawned = {row[K.OI_BODY] for row, _ in self.ops if row[K.OI_CODE] == K.OP_RESPAWN}
        for b, (flat, p_link, q_link) in enumerate(self.bodies):
            first = gl
            dynamic = not (flat.fixed_base and len(flat.links) == 0)
            frozen = (not dynamic) and b not in respawned
            flags = (K.BODY_FIXED if flat.fixed_base else 0) | (K.BODY_FROZEN if frozen else 0)
            # frozen bodies have no per-env state at all: their pose is the load pose in the body table
            body_i.append([flags, first, len(flat.links), -1 if frozen else state_off, -1, -1, -1])
            if not frozen:
                state_off += (K.BS_FIXED_END if flat.fixed_base else K.BS_FLOAT_END) + K.EXT_STRIDE
            I = flat.base_inertia
            rep_q = quat_from_mat(flat.T_base_report.R)
            body_f.append([flat.base_mass, *flat.base_com, I[0, 0], I[0, 1], I[0, 2], I[1, 1], I[1, 2], I[2, 2], *p_link,
                           *q_link, *flat.T_base_report.p, *rep_q, *self.colors[b], body_bound(flat), 0.0, 0.0, 0.0])
            for i, fl in enumerate(flat.links):
                parent = -1 if fl.parent < 0 else first + fl.parent
                link_i.append([parent, fl.joint_type, b, state_off, -1])
                link_state_offs.append(state_off)
                state_off += K.LS_STRIDE
                I = fl.inertia
                row = [*fl.origin.p, *fl.origin.R.reshape(-1), *fl.axis, fl.mass, *fl.com, I[0, 0], I[0, 1], I[0, 2],
                       I[1, 1], I[1, 2], I[2, 2], fl.damping, fl.lower, fl.upper, fl.max_force, fl.max_velocity]
                link_f.append(row + [0.0] * (K.LF_STRIDE - len(row)))
                gl += 1
            for fr in flat.frames:
                frame_i.append([b, -1 if fr.link < 0 else first + fr.link])
                frame_f.append([*fr.T.p, *fr.T.quat, *fr.T_com.p, *fr.T_com.quat])
            T_body = Transform.from_xyz_quat(p_link, q_link)
            # reach of each link origin from the base origin, whatever the joint angles (as in body_bound)
            link_reach = []
            for fl in flat.links:
                r = (link_reach[fl.parent] if fl.parent >= 0 else 0.0) + float(np.linalg.norm(fl.origin.p))
                if fl.joint_type == 1:
                    r += max(abs(fl.lower), abs(fl.upper)) if fl.lower <= fl.upper else np.inf
                link_reach.append(r)
            anchored = flat.fixed_base and b not in respawned  # the base never leaves its load pose
            for sh in flat.shapes:
                kind, T, prm, pts, hull_half = sh.kind, sh.T, np.zeros(3), None, 0.0
                if kind == SHAPE_SPHERE:
                    prm[0] = sh.params[0]
                elif kind == SHAPE_BOX:
                    prm[:] = sh.params
                elif kind == SHAPE_CAPSULE:
                    prm[:2] = sh.params
                elif kind == SHAPE_POINTS:
                    pts = np.asarray(sh.points, dtype=np.float64)
                    box = as_box(pts)
                    if box is not None:  # a box given as a mesh (wall.obj): use the analytic box
                        kind, (T, half), pts = SHAPE_BOX, box, None
                        prm[:] = half
                # Boxes on bodies that move are collided through their corners (there is no box-box routine);
                # boxes of the static world stay analytic.
                if kind == SHAPE_BOX and dynamic:
                    pts = box_corners(T, prm)
                    kind = SHAPE_POINTS
                if kind == SHAPE_POINTS:
                    T, r, half = fit_capsule(pts)
                    # (third parameter: radius of the sphere around the capsule's centre that holds every hull point -- the
                    # fitted capsule itself lets points near its caps stick out; the cull of the hull-hull narrow phase)
                    prm = np.array([r, half, float(np.max(np.linalg.norm(pts - T.p, axis=1)))])
                    # (and the half length at which a capsule of that radius about that axis contains every point, DG_SF_HULL_HALF)
                    t_ax = (pts - T.p) @ T.R[:, 2]; rho2 = np.sum((pts - T.p) ** 2, axis=1) - t_ax ** 2
                    hull_half = float(max(np.max(np.abs(t_ax) - np.sqrt(np.maximum(r * r - rho2, 0.0))), 0.0)) + 1e-9
                wflag = (K.SHAPE_WORLD if frozen else 0) | (K.SHAPE_NO_COLLIDE if getattr(sh, 'visual_only', False) else 0)
                wflag |= ((sh.urdf_link + 1) & 0xFFFF) << 8  # pybullet link index + 1, for segmentation masks
                if frozen:  # bake the body pose in
                    T = T_body * T
                    if pts is not None:
                        pts = pts @ T_body.R.T + T_body.p
                poff, npts, ploff, npl = 0, 0, 0, 0
                if pts is not None:
                    poff, npts = len(points), len(pts)
                    points.extend(pts.tolist())
                    pl = hull_planes(pts)
                    ploff, npl = len(planes), len(pl)
                    planes.extend(pl.tolist())
                shape_i.append([kind, b, -1 if sh.link < 0 else first + sh.link, poff, npts, ploff, npl, wflag])
                # colour in camera images: the YAML `color` on the base link's shapes (changeVisualShape(uid, -1, rgbaColor), reference
                # model.py:82-83), else the link's URDF material, else the body's default grey
                mat = getattr(sh, 'color', None)
                rgb3 = self.colors[b][:3] if ((sh.urdf_link < 0 and self.color_set[b]) or mat is None) else mat
                shape_f.append([*T.p, *T.R.reshape(-1), *prm, sh.friction, *rgb3, hull_half])
                shape_dyn.append(dynamic)
                # bounding sphere that holds the shape in EVERY reachable configuration, or None (floating / respawned base)
                if not anchored:
                    shape_anchor.append(None)
                elif frozen:
                    c = np.asarray(T.p, dtype=np.float64)
                    rad = float(np.max(np.linalg.norm(pts - c, axis=1))) if pts is not None else float(
                        prm[0] if kind == SHAPE_SPHERE else np.linalg.norm(prm) if kind == SHAPE_BOX else prm[0] + prm[1])
                    shape_anchor.append((c, rad))
                else:
                    ext = float(np.max(np.linalg.norm(pts, axis=1))) if pts is not None else float(np.linalg.norm(T.p)) + float(
                        prm[0] if kind == SHAPE_SPHERE else np.linalg.norm(prm) if kind == SHAPE_BOX else prm[0] + prm[1])
                    shape_anchor.append((np.asarray(p_link, dtype=np.float64), (link_reach[sh.link] if sh.link >= 0 else 0.0) + ext))
        addon_off = state_off
        state_dim = state_off + self.addon_state
        # per-env dynamics parameters (dynamics_randomizer): the links / bodies they belong to point at the addon state
        for row, _ in self.ops:
            if row[K.OI_CODE] == K.OP_RANDOMIZE_DYNAMICS:
                base = addon_off + row[K.OI_STATE_OFF]
                links = self.ilist[row[K.OI_ILIST]:row[K.OI_ILIST] + row[K.OI_N]]
                for k, gl_ in enumerate(links):
                    if link_i[gl_][K.LI_MASS_SCALE] >= 0:
                        raise ValueError('two dynamics_randomizer addons on the same joint')
                    link_i[gl_][K.LI_MASS_SCALE] = base + k
                body_i[row[K.OI_BODY]][K.BI_DYN_OFF] = base + len(links)
            if row[K.OI_CODE] == K.OP_RANDOMIZE_COLOR:
                body_i[row[K.OI_BODY]][K.BI_COLOR_OFF] = addon_off + row[K.OI_STATE_OFF]
            if row[K.OI_CODE] == K.OP_OBS_FT and row[K.OI_STATE_OFF] >= 0:  # the sensor that owns its body's saved velocities
                if body_i[row[K.OI_BODY]][K.BI_PREV_OFF] < 0:
                    body_i[row[K.OI_BODY]][K.BI_PREV_OFF] = addon_off + row[K.OI_STATE_OFF]

        # candidate collision pairs: different bodies, at least one of them able to move,
        # and a narrow-phase routine exists for the pair (no box-box)
        cand, max_contacts, static_pairs = [], 0, 0
        coupled = {(k[0][0], k[0][2]) for k in self.constraints}
        for a in range(len(shape_i)):
            for c in range(a + 1, len(shape_i)):
                if shape_i[a][1] == shape_i[c][1] or not (shape_dyn[a] or shape_dyn[c]):
                    continue
                if (shape_i[a][1], shape_i[c][1]) in coupled or (shape_i[c][1], shape_i[a][1]) in coupled:
                    continue  # two bodies held together by a fixed constraint
                if (shape_i[a][7] | shape_i[c][7]) & K.SHAPE_NO_COLLIDE:
                    continue
                ta, tc = shape_i[a][0], shape_i[c][0]
                if ta == SHAPE_BOX and tc == SHAPE_BOX:
                    continue
                # static pruning: two shapes whose bases are bolted down can be too far apart to ever touch
                # (the shoulders of two arms a metre apart); such a pair can produce no contact in any backend
                if shape_anchor[a] is not None and shape_anchor[c] is not None:
                    (ca, ra), (cc, rc) = shape_anchor[a], shape_anchor[c]
                    if np.linalg.norm(ca - cc) - ra - rc > self.params['contact_margin'] + 1e-3:
                        continue
                kinds = {ta, tc}
                per_pair = 4 if kinds == {SHAPE_POINTS, SHAPE_BOX} else 2 if kinds == {SHAPE_CAPSULE, SHAPE_BOX} else 1
                max_contacts += per_pair
                if not (shape_dyn[a] and shape_dyn[c]):
                    static_pairs += per_pair
                # broad-phase key: (moving body, static shape) or (moving body, moving body)
                if shape_dyn[a] and shape_dyn[c]:
                    key = (1, shape_i[a][1], shape_i[c][1], -1)
                else:
                    dyn, sta = (a, c) if shape_dyn[a] else (c, a)
                    key = (0, shape_i[dyn][1], -1, sta)
                cand.append((key, a, c))
        cand.sort(key=lambda t: (t[0], t[1], t[2]))
        pairs, groups = [], []
        for key, a, c in cand:
            if not groups or groups[-1][5] != key:
                groups.append([len(pairs), 0, key[1], key[2], key[3], key])
            groups[-1][1] += 1
            pairs.append([a, c])
        groups = [g[:5] for g in groups]
        # Contact budget per env (rows live in LDS).  Default: every contact a moving shape can have
        # with the static world, plus a small pool for moving-vs-moving contacts; `max_contacts`
        # in the env config overrides it.  Contacts beyond the budget are dropped in pair order --
        # by the oracle and the kernels alike.
        n_dyn = sum(1 for b in self.bodies if not (b[0].fixed_base and len(b[0].links) == 0))
        budget = self.max_contacts if self.max_contacts is not None else max(3, static_pairs + n_dyn)
        max_contacts = min(max_contacts, int(budget), 32)

        def arr(rows, width, dtype):
            if not rows:
                return np.zeros((0, width), dtype=dtype)
            return np.asarray(rows, dtype=dtype).reshape(len(rows), width)

        tables_i = [
            ('OFF_BODY_I', arr(body_i, K.BI_STRIDE, np.int32)),
            ('OFF_LINK_I', arr(link_i, K.LI_STRIDE, np.int32)),
            ('OFF_FRAME_I', arr(frame_i, K.FI_STRIDE, np.int32)),
            ('OFF_SHAPE_I', arr(shape_i, K.SI_STRIDE, np.int32)),
            ('OFF_PAIR_I', arr(pairs, K.PI_STRIDE, np.int32)),
            ('OFF_GROUP_I', arr(groups, K.GI_STRIDE, np.int32)),
            ('OFF_CAMERA_I', arr([[c[0], c[1], c[2], c[3], c[4]] for c in self.cameras], K.CI_STRIDE, np.int32)),
            ('OFF_OP_I', arr([o[0] for o in self.ops], K.OI_STRIDE, np.int32)),
            ('OFF_ILIST', np.asarray(self.ilist, dtype=np.int32).reshape(-1, 1)),
            ('OFF_CONS_I', arr([[k[0][0], -1 if k[0][1] < 0 else body_i[k[0][0]][1] + k[0][1], k[0][2], -1 if k[0][3] < 0 else body_i[k[0][2]][1] + k[0][3]]
                                for k in self.constraints], K.KI_STRIDE, np.int32)),
        ]
        tables_f = [
            ('OFF_BODY_F', arr(body_f, K.BF_STRIDE, np.float64)),
            ('OFF_LINK_F', arr(link_f, K.LF_STRIDE, np.float64)),
            ('OFF_FRAME_F', arr(frame_f, K.FF_STRIDE, np.float64)),
            ('OFF_SHAPE_F', arr(shape_f, K.SF_STRIDE, np.float64)),
            ('OFF_POINT_F', arr(points, 3, np.float64)),
            ('OFF_PLANE_F', arr(planes, 4, np.float64)),
            ('OFF_CAMERA_F', arr([[*c[5].p, *c[5].quat, c[6], c[7], c[8], float(np.tan(np.radians(0.5 * c[6]))), 0.0] for c in self.cameras], K.CF_STRIDE, np.float64)),
            ('OFF_OP_F', arr([o[1] for o in self.ops], K.OF_STRIDE, np.float64)),
            ('OFF_FLIST', np.asarray(self.flist, dtype=np.float64).reshape(-1, 1)),
            ('OFF_CONS_F', arr([k[1] for k in self.constraints], K.KF_STRIDE, np.float64)),
        ]
        H = np.zeros(K.H_INT_COUNT, dtype=np.int32)
        H[K.H_MAGIC] = K.MAGIC
        H[K.H_VERSION] = K.VERSION
        H[K.H_N_BODIES] = nb
        H[K.H_N_LINKS] = len(link_i)
        H[K.H_N_FRAMES] = len(frame_i)
        H[K.H_N_SHAPES] = len(shape_i)
        H[K.H_N_POINTS] = len(points)
        H[K.H_N_PLANES] = len(planes)
        H[K.H_N_CAMERAS] = len(self.cameras)
        H[K.H_N_PAIRS] = len(pairs)
        H[K.H_N_GROUPS] = len(groups)
        H[K.H_N_OPS] = len(self.ops)
        H[K.H_N_ILIST] = len(self.ilist)
        H[K.H_N_FLIST] = len(self.flist)
        H[K.H_N_CONSTRAINTS] = len(self.constraints)
        H[K.H_ACT_DIM] = self.dims['act']
        H[K.H_OBS_DIM] = self.dims['obs']
        H[K.H_REW_DIM] = self.dims['rew']
        H[K.H_TERM_DIM] = self.dims['term']
        H[K.H_SUBSTEPS] = self.substeps
        H[K.H_SOLVER_ITERS] = self.solver_iterations
        H[K.H_MAX_EPISODE_STEPS] = self.max_episode_steps
        H[K.H_HOT_START] = self.hot_start
        H[K.H_IK_ITERS] = int(self.params['ik_iterations'])
        # contact impulse cache (warm starting), behind the addon state: [count][key normal t1 t2] x max_contacts
        warm_off = -1
        if len(pairs) > 0 and max_contacts > 0 and (self.params['warmstart'] > 0 or self.params['warmstart_friction'] > 0):
            warm_off = state_dim
            state_dim += 1 + K.WS_STRIDE * max_contacts
        H[K.H_WARM_OFF] = warm_off
        H[K.H_STATE_DIM] = state_dim
        H[K.H_ADDON_STATE_OFF] = addon_off
        H[K.H_N_ADDON_STATE] = self.addon_state
        H[K.H_MAX_CONTACTS] = max_contacts
        H[K.H_REW_MODE] = self.rew_mode
        H[K.H_TERM_MODE] = self.term_mode
        H[K.H_N_TERM_GROUPS] = len(self.term_groups)
        off = K.H_INT_COUNT
        chunks_i = [H]
        for name, t in tables_i:
            H[getattr(K, 'H_' + name)] = off
            chunks_i.append(t.reshape(-1))
            off += t.size
        HF = np.zeros(K.HF_FLOAT_COUNT, dtype=np.float64)
        p = self.params
        HF[K.HF_DT] = self.timestep / self.substeps
        HF[K.HF_GRAV_X:K.HF_GRAV_X + 3] = self.gravity
        HF[K.HF_RESIDUAL_THRESHOLD] = p['residual_threshold']
        HF[K.HF_CONTACT_ERP] = p['contact_erp']
        HF[K.HF_LIMIT_ERP] = p['limit_erp']
        HF[K.HF_LINEAR_SLOP] = p['linear_slop']
        HF[K.HF_LIN_DAMPING] = p['linear_damping']
        HF[K.HF_ANG_DAMPING] = p['angular_damping']
        HF[K.HF_MAX_COORD_VEL] = p['max_coordinate_velocity']
        HF[K.HF_DEFAULT_MOTOR_IMPULSE] = p['default_motor_impulse']
        HF[K.HF_IK_LAMBDA_SQ] = p['ik_lambda_sq']
        HF[K.HF_IK_JOINT_DAMPING] = p['ik_joint_damping']
        HF[K.HF_IK_RESIDUAL] = p['ik_residual']
        HF[K.HF_IK_MAX_ANGLE] = p['ik_max_angle']
        HF[K.HF_IK_NULL_REST_GAIN] = p['ik_null_rest_gain']
        HF[K.HF_IK_NULL_LIMIT_GAIN] = p['ik_null_limit_gain']
        HF[K.HF_CONTACT_MARGIN] = p['contact_margin']
        HF[K.HF_WARMSTART] = p['warmstart']
        HF[K.HF_WARMSTART_FRICTION] = p['warmstart_friction']
        HF[K.HF_MOTOR_GUESS] = p['motor_guess']
        HF[K.HF_LIMIT_GUESS] = p['limit_guess'] if p['motor_guess'] > 0 else 0.0
        if p['motor_impulse_timebase'] not in ('substep', 'step'):
            raise ValueError("motor_impulse_timebase must be 'substep' or 'step'")
        HF[K.HF_HULL_CONTACTS] = p['hull_contacts']
        HF[K.HF_HULL_MARGIN] = p['hull_margin']
        HF[K.HF_MOTOR_IMPULSE_SCALE] = float(self.substeps) if p['motor_impulse_timebase'] == 'step' else 1.0
        off = K.HF_FLOAT_COUNT
        chunks_f = [HF]
        for name, t in tables_f:
            H[getattr(K, 'H_' + name)] = off
            chunks_f.append(t.reshape(-1))
            off += t.size
        I = np.ascontiguousarray(np.concatenate(chunks_i).astype(np.int32))
        F = np.ascontiguousarray(np.concatenate(chunks_f).astype(np.float64))
        layout = SceneLayout(self, I, F, body_i, link_state_offs, addon_off, state_dim, max_contacts)
        return layout


class SceneLayout:
    """The finished blob plus the host-side offsets addons need for state access."""
    def __init__(self, builder, I, F, body_i, link_state_offs, addon_off, state_dim, max_contacts):
        self.I = I
        self.F = F
        self.body_state_off = [r[3] for r in body_i]
        self.body_fixed = [bool(r[0] & K.BODY_FIXED) for r in body_i]
        self.body_first_link = [r[1] for r in body_i]
        self.body_n_links = [r[2] for r in body_i]
        self.link_state_off = link_state_offs
        self.addon_off = addon_off
        self.state_dim = state_dim
        self.max_contacts = max_contacts
        self.aliases = dict(builder.aliases)  # uid of an attached child model -> (body, link offset, frame offset, frame that stands for its base)
        self.warm_off = int(I[K.H_WARM_OFF])  # state offset of the contact impulse cache (warm starting), -1: none
        self.physical_dim = self.warm_off if self.warm_off >= 0 else state_dim  # state columns before that cache
        self.act_dim = builder.dims['act']
        self.obs_dim = builder.dims['obs']
        self.rew_dim = builder.dims['rew']
        self.term_dim = builder.dims['term']
        self.n_links = len(link_state_offs)
        self.n_bodies = len(body_i)
        self.n_slots = builder.n_slots
        self.substeps = builder.substeps
        self.dt = builder.timestep / builder.substeps

    def resolve_frame(self, uid, frame):
        """``(body index, body-local frame index)`` behind a Model's ``uid`` and one of ITS frame ids (-1: its base).  A model
        attached to its parent with the default ``attach: merge`` owns no body: its uid is an alias, its frames live in the
        parent's body behind an offset and its base is one of the parent's frames."""
        uid, frame = int(uid), int(frame)
        if uid in self.aliases:
            body, _, foff, basef = self.aliases[uid]
            return body, (basef if frame < 0 else foff + frame)
        return uid, frame
