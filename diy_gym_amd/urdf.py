"""URDF reader and multibody flattener.

Replaces ``p.loadURDF`` + ``p.getNumJoints`` / ``p.getJointInfo`` for this
backend (reference call sites: diy_gym/model.py:65, :94-96;
joint_controller.py:21-33; joint_state_sensor.py:29-43).  Parsing uses
``xml.etree`` only.

Conventions that are restated from pybullet/Bullet **from recollection** (the
wheel is not available to check; see DESIGN.md "Restated pybullet behaviours"):

* joint indices follow a depth-first walk from the root link, children in file
  order, fixed joints included; link index ``i`` is the child link of joint ``i``;
* a link without ``<inertial>`` gets mass 1 and unit inertia, except a link
  named ``world`` which gets mass 0 (that is what makes a ``world`` root link a
  fixed base: ur5_robot.urdf:315-320);
* ``q_index`` (``getJointInfo()[3]``) is -1 for fixed joints;
* ``maxForce`` / ``maxVelocity`` come from ``<limit effort= velocity=>`` and are
  0 when the tag is absent; ``continuous`` joints are revolute without limits.

Flattening merges every fixed-joint subtree into the moving link (or base) it
is rigidly attached to -- inertias are lumped exactly, named frames keep their
offsets -- so the device kernels only see 1-DoF joints.
"""
import os
import xml.etree.ElementTree as ET

import numpy as np

from .mathx import Transform, mat_from_euler, quat_from_mat

JOINT_REVOLUTE = 0
JOINT_PRISMATIC = 1
JOINT_FIXED = 4  # pybullet's numeric value; only used for reporting

SHAPE_SPHERE = 0
SHAPE_BOX = 1
SHAPE_CAPSULE = 2
SHAPE_POINTS = 3  # convex vertex cloud


def _floats(text, n=None, default=None):
    if text is None:
        return None if default is None else np.array(default, dtype=np.float64)
    vals = np.array([float(t) for t in text.replace(',', ' ').split()], dtype=np.float64)
    if n is not None and vals.size != n:
        raise ValueError('expected %d numbers, got %r' % (n, text))
    return vals


def _origin(elem):
    if elem is None:
        return Transform()
    o = elem.find('origin')
    if o is None:
        return Transform()
    return Transform.from_xyz_rpy(_floats(o.get('xyz'), 3, [0, 0, 0]), _floats(o.get('rpy'), 3, [0, 0, 0]))


def resolve_mesh(urdf_dir, filename):
    """Path of a ``<mesh filename=...>``.  Plain names are relative to the URDF.  ``package://pkg/rest`` (ROS style,
    used by ``ur5_2f.urdf`` / ``ur5_3f.urdf`` for the gripper meshes) is looked up like Bullet's URDF importer does
    [RECOLLECTION]: the part after ``package://`` is tried under the URDF's directory and each of its parents; since the
    reference's data tree has no ``robotiq/`` level (``package://robotiq/robotiq_2f/...`` lives at
    ``data/robotiq_2f/...``) the package name is also dropped as a second attempt [decision]."""
    if not filename.startswith('package://'):
        return os.path.join(urdf_dir, filename)
    rest = filename[len('package://'):]
    tails = [rest] + ([rest.split('/', 1)[1]] if '/' in rest else [])
    d = os.path.abspath(urdf_dir)
    for _ in range(6):
        for t in tails:
            cand = os.path.join(d, t)
            if os.path.isfile(cand):
                return cand
        d = os.path.dirname(d)
    return os.path.join(urdf_dir, rest)


class UrdfShape:
    def __init__(self, kind, origin, size=None, radius=0.0, length=0.0, mesh=None, mesh_scale=None):
        self.kind = kind  # 'sphere' | 'box' | 'cylinder' | 'capsule' | 'mesh' | 'plane'
        self.origin = origin
        self.size = size
        self.radius = radius
        self.length = length
        self.mesh = mesh
        self.mesh_scale = mesh_scale


class UrdfLink:
    def __init__(self, name):
        self.name = name
        self.mass = 0.0
        self.inertial_origin = Transform()
        self.inertia = np.zeros((3, 3))
        self.has_inertial = False
        self.collisions = []
        self.lateral_friction = None
        self.color = None  # rgba of the first <visual>'s <material>, if the file has one
        self.child_joints = []
        self.parent_joint = None


class UrdfJoint:
    def __init__(self, name, jtype):
        self.name = name
        self.type = jtype
        self.parent = None
        self.child = None
        self.origin = Transform()
        self.axis = np.array([1.0, 0.0, 0.0])
        self.lower = 0.0
        self.upper = -1.0  # lower > upper == "no limit", like Bullet
        self.effort = 0.0
        self.velocity = 0.0
        self.damping = 0.0
        self.friction = 0.0
        self.index = -1
        self.q_index = -1

    @property
    def movable(self):
        return self.type in ('revolute', 'continuous', 'prismatic')


def _parse_geometry(geom, origin):
    if geom is None:
        return None
    for child in geom:
        tag = child.tag
        if tag == 'sphere':
            return UrdfShape('sphere', origin, radius=float(child.get('radius')))
        if tag == 'box':
            return UrdfShape('box', origin, size=_floats(child.get('size'), 3))
        if tag == 'cylinder':
            return UrdfShape('cylinder', origin, radius=float(child.get('radius')), length=float(child.get('length')))
        if tag == 'capsule':
            return UrdfShape('capsule', origin, radius=float(child.get('radius')), length=float(child.get('length')))
        if tag == 'mesh':
            return UrdfShape('mesh', origin, mesh=child.get('filename'),
                             mesh_scale=_floats(child.get('scale'), 3, [1, 1, 1]))
        if tag == 'plane':
            return UrdfShape('plane', origin, size=_floats(child.get('normal'), 3, [0, 0, 1]))
    return None


class UrdfRobot:
    """Parsed URDF with pybullet-style joint numbering."""
    def __init__(self, path):
        self.path = path
        self.dir = os.path.dirname(os.path.abspath(path))
        root = ET.parse(path).getroot()
        self.name = root.get('name', os.path.basename(path))
        self.links = {}
        self.link_order = []
        # <material name=...><color rgba=.../></material> at the top level, referenced by name from the visuals
        materials = {}
        for me in root.findall('material'):
            ce = me.find('color')
            if ce is not None and me.get('name'):
                materials[me.get('name')] = _floats(ce.get('rgba'), 4)
        for le in root.findall('link'):
            link = UrdfLink(le.get('name'))
            for ve in le.findall('visual'):  # the colour of the link's first visual (what pybullet shows for an untextured mesh [R])
                me = ve.find('material')
                if me is None:
                    continue
                ce = me.find('color')
                if ce is not None:
                    link.color = _floats(ce.get('rgba'), 4)
                    materials.setdefault(me.get('name'), link.color)
                elif me.get('name') in materials:
                    link.color = materials[me.get('name')]
                if link.color is not None:
                    break
            inertial = le.find('inertial')
            if inertial is not None:
                link.has_inertial = True
                link.inertial_origin = _origin(inertial)
                m = inertial.find('mass')
                link.mass = float(m.get('value')) if m is not None else 0.0
                ie = inertial.find('inertia')
                if ie is not None:
                    g = lambda k: float(ie.get(k, 0.0))
                    link.inertia = np.array([[g('ixx'), g('ixy'), g('ixz')], [g('ixy'), g('iyy'), g('iyz')],
                                             [g('ixz'), g('iyz'), g('izz')]])
            elif link.name == 'world':
                link.mass = 0.0
            else:
                # Bullet's URDF importer: "No inertial data for link, using mass=1,
                # localinertiadiagonal = 1,1,1, identity local inertial frame" [RECOLLECTION]
                link.mass = 1.0
                link.inertia = np.eye(3)
            for ce in le.findall('collision'):
                shape = _parse_geometry(ce.find('geometry'), _origin(ce))
                if shape is not None:
                    link.collisions.append(shape)
            contact = le.find('contact')
            if contact is not None:
                lf = contact.find('lateral_friction')
                if lf is not None:
                    link.lateral_friction = float(lf.get('value'))
            self.links[link.name] = link
            self.link_order.append(link.name)

        self.joints_by_name = {}
        file_joints = []
        for je in root.findall('joint'):
            j = UrdfJoint(je.get('name'), je.get('type'))
            j.parent = je.find('parent').get('link')
            j.child = je.find('child').get('link')
            j.origin = _origin(je)
            ax = je.find('axis')
            if ax is not None:
                a = _floats(ax.get('xyz'), 3)
                n = np.linalg.norm(a)
                j.axis = a / n if n > 0 else a
            lim = je.find('limit')
            if lim is not None:
                j.effort = float(lim.get('effort', 0.0))
                j.velocity = float(lim.get('velocity', 0.0))
                if j.type != 'continuous':
                    j.lower = float(lim.get('lower', 0.0))
                    j.upper = float(lim.get('upper', 0.0))
            dyn = je.find('dynamics')
            if dyn is not None:
                j.damping = float(dyn.get('damping', 0.0))
                j.friction = float(dyn.get('friction', 0.0))
            if j.type not in ('revolute', 'continuous', 'prismatic', 'fixed'):
                raise ValueError('joint %s: type %r is not supported by this backend' % (j.name, j.type))
            file_joints.append(j)
            self.joints_by_name[j.name] = j
            self.links[j.parent].child_joints.append(j)
            self.links[j.child].parent_joint = j

        roots = [n for n in self.link_order if self.links[n].parent_joint is None]
        if len(roots) != 1:
            raise ValueError('%s: expected exactly one root link, found %r' % (path, roots))
        self.root = roots[0]

        self._number_joints()  # depth-first numbering, children in file order

    def _number_joints(self):
        """pybullet's numbering: depth-first from the root, children in file order; q_index counts movable joints."""
        self.joints = []
        stack = list(reversed(self.links[self.root].child_joints))
        q = 0
        while stack:
            j = stack.pop()
            j.index = len(self.joints)
            j.q_index = -1
            if j.movable:
                j.q_index = q
                q += 1
            self.joints.append(j)
            stack.extend(reversed(self.links[j.child].child_joints))
        self.num_dofs = q

    def rerooted(self, new_root):
        """The same mechanism described from link ``new_root`` (a copy; this robot is untouched).  Used for child
        models attached by one of their links (``child_frame``, reference model.py:71-77): the tree then hangs from
        that link.  Every joint on the path old root -> new root is reversed: the old parent becomes the child, its
        link frame moves to the joint (so that the URDF rule 'child frame = joint frame' still holds; everything
        attached to it is re-expressed), the axis changes sign so that the joint coordinate keeps its meaning, sign
        and limits.  Joint numbering follows the new tree."""
        import copy
        if new_root not in self.links:
            raise ValueError('no link %r' % new_root)
        r = copy.deepcopy(self)
        if new_root == r.root:
            return r
        path = []  # joints from the new root up to the old root
        link = r.links[new_root]
        while link.parent_joint is not None:
            path.append(link.parent_joint)
            link = r.links[link.parent_joint.parent]
        shift_of_child = Transform()  # how the frame of the joint's (old) child was re-expressed: x_new = S x_old
        for j in path:
            P, C = r.links[j.parent], r.links[j.child]
            S = j.origin.inverse()  # P's frame moves to this joint: x_in_new_P_frame = origin^-1 x_in_old_P_frame
            P.inertial_origin = S * P.inertial_origin
            for sh in P.collisions:
                sh.origin = S * sh.origin
            for cj in P.child_joints:
                if cj is not j:
                    cj.origin = S * cj.origin  # (the upstream path joint, if any, is P's PARENT joint and is handled in its own turn)
            P.child_joints = [cj for cj in P.child_joints if cj is not j]
            C.child_joints.append(j)
            # the reversed joint: its frame sits where C's old frame was, seen from C's (possibly moved) new frame
            j.origin = shift_of_child
            j.parent, j.child = C.name, P.name
            j.axis = -j.axis
            C.parent_joint = None if C.name == new_root else C.parent_joint
            P.parent_joint = j
            shift_of_child = S
        r.links[new_root].parent_joint = None
        r.root = new_root
        r._number_joints()
        return r

    @property
    def joint_names(self):
        return [j.name for j in self.joints]

    def joint_info(self, i):
        """Subset of ``p.getJointInfo`` fields that the reference reads."""
        j = self.joints[i]
        return {
            'index': j.index,
            'name': j.name,
            'type': {'revolute': JOINT_REVOLUTE, 'continuous': JOINT_REVOLUTE, 'prismatic': JOINT_PRISMATIC}.get(
                j.type, JOINT_FIXED),
            'q_index': j.q_index,
            'damping': j.damping,
            'friction': j.friction,
            'lower': j.lower,
            'upper': j.upper,
            'max_force': j.effort,
            'max_velocity': j.velocity,
            'link_name': j.child,
        }


class FlatLink:
    """One moving (1-DoF) link after fixed-joint merging."""
    def __init__(self):
        self.parent = -1  # index of parent moving link, -1 = base
        self.joint_type = JOINT_REVOLUTE
        self.joint_index = -1  # pybullet joint index that moves this link
        self.name = ''
        self.origin = Transform()  # joint frame (q = 0) in the parent's reference frame
        self.axis = np.array([0.0, 0.0, 1.0])
        self.mass = 0.0
        self.com = np.zeros(3)
        self.inertia = np.zeros((3, 3))  # about the COM, link axes
        self.damping = 0.0
        self.lower = 0.0
        self.upper = -1.0
        self.max_force = 0.0
        self.max_velocity = 0.0


class FlatFrame:
    """A named frame rigidly attached to a moving link (or the base)."""
    def __init__(self, name, link, T_link_frame, T_link_com):
        self.name = name
        self.link = link  # moving link index, -1 = base
        self.T = T_link_frame  # URDF link frame (getLinkState items 4, 5)
        self.T_com = T_link_com  # inertial frame (getLinkState items 0, 1)


class FlatShape:
    def __init__(self, kind, link, T, params, points=None, friction=1.0, urdf_link=-1):
        self.color = None  # rgb of the owning URDF link's material (camera images), None: the body's colour
        self.urdf_link = urdf_link  # pybullet link index of the URDF link that owns the shape (-1 = base)
        self.kind = kind
        self.link = link
        self.T = T  # shape frame in the link's reference frame
        self.params = params  # sphere: [r]; box: half extents; capsule: [r, half_len]
        self.points = points  # SHAPE_POINTS: [n,3] in link reference frame
        self.friction = friction


class FlatBody:
    """A URDF flattened into base + 1-DoF links + frames + collision shapes."""
    def __init__(self, robot, scale=1.0, fixed_base=False, mass_override=None, mesh_loader=None,
                 max_hull_points=32):
        self.robot = robot
        self.scale = float(scale)
        s = self.scale
        links = robot.links
        root = links[robot.root]

        root_mass = root.mass
        root_inertia = root.inertia.copy()
        if mass_override is not None:
            # p.changeDynamics(uid, -1, mass=...) touches the root link only (reference
            # model.py:79-80); inertia is rescaled by the mass ratio [decision, DESIGN.md].
            ratio = (float(mass_override) / root_mass) if root_mass > 0 else 1.0
            root_mass = float(mass_override)
            root_inertia = root_inertia * ratio
        self.fixed_base = bool(fixed_base) or root_mass == 0.0

        self.links = []
        self.frames = []  # index == pybullet joint/link index
        self.shapes = []
        # what every URDF link contributed and where, keyed by the pybullet index of its parent joint: (anchor moving
        # link, mass, COM in the anchor frame, inertia about that COM in anchor axes) -- the force/torque sensor needs
        # the rigid cluster on the child side of one joint, which the merged links no longer show
        self.parts = {}
        # accumulators: link index (-1 base) -> [m, h(3), I_O(3x3)]
        acc = {-1: [0.0, np.zeros(3), np.zeros((3, 3))]}

        def add_inertia(anchor, T_anchor_link, link, mass, inertia, joint_index=None):
            if mass <= 0.0:
                return
            Tc = T_anchor_link * Transform(link.inertial_origin.R, link.inertial_origin.p * s)
            c = Tc.p
            Ic = Tc.R @ (inertia * (s * s)) @ Tc.R.T
            if joint_index is not None:
                self.parts[joint_index] = (anchor, float(mass), c.copy(), Ic.copy())
            a = acc[anchor]
            a[0] += mass
            a[1] += mass * c
            a[2] += Ic + mass * (np.dot(c, c) * np.eye(3) - np.outer(c, c))

        def add_shapes(anchor, T_anchor_link, link, urdf_link=-1):
            mu = link.lateral_friction if link.lateral_friction is not None else 0.5
            n_before = len(self.shapes)
            for sh in link.collisions:
                T = T_anchor_link * Transform(sh.origin.R, sh.origin.p * s)
                if sh.kind == 'sphere':
                    self.shapes.append(FlatShape(SHAPE_SPHERE, anchor, T, np.array([sh.radius * s]), friction=mu))
                elif sh.kind == 'box':
                    self.shapes.append(FlatShape(SHAPE_BOX, anchor, T, 0.5 * sh.size * s, friction=mu))
                elif sh.kind in ('capsule', 'cylinder'):
                    # cylinders are approximated by the inscribed capsule of the same radius
                    # (documented deviation; Bullet uses a true cylinder)
                    half = max(0.5 * sh.length * s - (sh.radius * s if sh.kind == 'cylinder' else 0.0), 0.0)
                    self.shapes.append(
                        FlatShape(SHAPE_CAPSULE, anchor, T, np.array([sh.radius * s, half]), friction=mu))
                elif sh.kind == 'mesh':
                    if mesh_loader is None:
                        continue
                    pts = mesh_loader(resolve_mesh(robot.dir, sh.mesh), max_hull_points)
                    if pts is None or len(pts) == 0:
                        continue
                    pts = (pts * sh.mesh_scale[None, :] * s) @ T.R.T + T.p
                    self.shapes.append(FlatShape(SHAPE_POINTS, anchor, Transform(), np.zeros(3), points=pts, friction=mu))
            for shp in self.shapes[n_before:]:
                shp.urdf_link = urdf_link
                shp.color = None if link.color is None else [float(c) for c in link.color[:3]]

        # root
        self.base_name = root.name
        self.T_base_report = Transform(root.inertial_origin.R, root.inertial_origin.p * s)  # root inertial frame
        add_inertia(-1, Transform(), root, root_mass, root_inertia)
        add_shapes(-1, Transform(), root)
        anchor_of = {robot.root: (-1, Transform())}

        for j in robot.joints:  # DFS order guarantees the parent was visited
            p_anchor, T_anchor_parent = anchor_of[j.parent]
            child = links[j.child]
            T_joint = T_anchor_parent * Transform(j.origin.R, j.origin.p * s)
            if j.movable:
                fl = FlatLink()
                fl.parent = p_anchor
                fl.joint_type = JOINT_PRISMATIC if j.type == 'prismatic' else JOINT_REVOLUTE
                fl.joint_index = j.index
                fl.name = j.name
                fl.origin = T_joint
                fl.axis = j.axis.copy()
                fl.damping = j.damping
                fl.lower, fl.upper = (j.lower * s, j.upper * s) if j.type == 'prismatic' else (j.lower, j.upper)
                fl.max_force = j.effort
                fl.max_velocity = j.velocity
                idx = len(self.links)
                self.links.append(fl)
                acc[idx] = [0.0, np.zeros(3), np.zeros((3, 3))]
                anchor, T_anchor_child = idx, Transform()
            else:
                anchor, T_anchor_child = p_anchor, T_joint
            anchor_of[j.child] = (anchor, T_anchor_child)
            add_inertia(anchor, T_anchor_child, child, child.mass, child.inertia, j.index)
            add_shapes(anchor, T_anchor_child, child, j.index)
            T_com = T_anchor_child * Transform(child.inertial_origin.R, child.inertial_origin.p * s)
            self.frames.append(FlatFrame(j.name, anchor, T_anchor_child, T_com))

        def finish(a):
            m, h, IO = a
            if m <= 0.0:
                return 0.0, np.zeros(3), np.zeros((3, 3))
            c = h / m
            Ic = IO - m * (np.dot(c, c) * np.eye(3) - np.outer(c, c))
            return m, c, 0.5 * (Ic + Ic.T)

        self.base_mass, self.base_com, self.base_inertia = finish(acc[-1])
        for i, fl in enumerate(self.links):
            fl.mass, fl.com, fl.inertia = finish(acc[i])
            if fl.mass <= 0.0:
                raise ValueError('moving link %s has zero mass' % fl.name)
        if self.fixed_base:
            self.base_mass = 0.0

        self.joint_to_dof = [j.q_index for j in robot.joints]
        self.dof_to_joint = [fl.joint_index for fl in self.links]

    def attach(self, child, parent_frame, T_rel):
        """Bolt another flattened body onto this one (child models, reference model.py:69-77, where pybullet
        couples the two with ``createConstraint(JOINT_FIXED)``).  [decision] The coupling is made rigid: the child's
        base becomes part of the link that carries ``parent_frame`` (the base when negative), its joints, frames and
        shapes are appended to this body.  ``T_rel`` is the pose of the child's base INERTIAL frame in the parent
        link's INERTIAL frame -- the two frames ``createConstraint`` takes its pivots in [RECOLLECTION].

        Returns ``(link_offset, frame_offset, base_frame)``: where the child's joints / frames start in this body and
        the index of a new frame that stands for the child's base."""
        import copy
        if parent_frame >= 0:
            anchor, T_anchor_pf = self.frames[parent_frame].link, self.frames[parent_frame].T_com
        else:
            anchor, T_anchor_pf = -1, self.T_base_report
        T_attach = T_anchor_pf * T_rel * child.T_base_report.inverse()  # anchor link frame -> child's root link frame
        link_off, frame_off = len(self.links), len(self.frames)

        # inertia of the child's base joins the anchor link's
        def combine(m1, c1, I1, m2, c2, I2):
            m = m1 + m2
            if m <= 0.0:
                return 0.0, np.zeros(3), np.zeros((3, 3))
            c = (m1 * c1 + m2 * c2) / m
            shift = lambda mm, cc: mm * (np.dot(cc - c, cc - c) * np.eye(3) - np.outer(cc - c, cc - c))
            return m, c, I1 + shift(m1, c1) + I2 + shift(m2, c2)
        m2, c2 = child.base_mass, T_attach.apply(child.base_com)
        I2 = T_attach.R @ child.base_inertia @ T_attach.R.T
        if anchor >= 0:
            fl = self.links[anchor]
            fl.mass, fl.com, fl.inertia = combine(fl.mass, fl.com, fl.inertia, m2, c2, I2)
        elif not self.fixed_base:
            self.base_mass, self.base_com, self.base_inertia = combine(self.base_mass, self.base_com, self.base_inertia, m2, c2, I2)
        for sh in child.shapes:
            sh2 = copy.copy(sh)
            if sh.link < 0:
                sh2.link = anchor
                sh2.T = T_attach * sh.T
                if sh.points is not None:
                    sh2.points = np.asarray(sh.points) @ T_attach.R.T + T_attach.p
            else:
                sh2.link = sh.link + link_off
            self.shapes.append(sh2)
        for fl in child.links:
            fl2 = copy.copy(fl)
            if fl.parent < 0:
                fl2.parent, fl2.origin = anchor, T_attach * fl.origin
            else:
                fl2.parent = fl.parent + link_off
            self.links.append(fl2)
        for fr in child.frames:
            if fr.link < 0:
                self.frames.append(FlatFrame(fr.name, anchor, T_attach * fr.T, T_attach * fr.T_com))
            else:
                self.frames.append(FlatFrame(fr.name, fr.link + link_off, fr.T, fr.T_com))
        self.frames.append(FlatFrame(child.base_name + '__attached_base', anchor, T_attach, T_attach * child.T_base_report))
        return link_off, frame_off, len(self.frames) - 1

    def ft_cluster(self, joint_index):
        """Everything on the CHILD side of URDF joint ``joint_index`` (pybullet numbering), for the force/torque sensor:

        ``anchor``   moving link the joint's child link is part of (-1 = base),
        ``rigid``    (mass, com, inertia about the com) of the URDF links rigidly attached on the child side -- the
                     joint's child link and everything reached from it through fixed joints only -- in the anchor
                     link's frame; the whole moving link when the joint itself is the link's (movable) joint,
        ``moving``   moving links hanging off that rigid cluster (their whole subtrees are on the child side too),
        ``urdf_links`` pybullet link indices of the rigid cluster (shapes owned by them belong to the child side).
        """
        robot = self.robot
        if not 0 <= joint_index < len(robot.joints):
            raise ValueError('joint index %d out of range' % joint_index)
        if joint_index >= len(self.frames) or len(self.frames) != len(robot.joints):
            raise NotImplementedError('force/torque sensing across the joints of an attached child model')
        J = robot.joints[joint_index]
        anchor = self.frames[joint_index].link
        cluster, moving = [], []
        stack = [J]
        while stack:
            j = stack.pop()
            cluster.append(j.index)
            for cj in robot.links[j.child].child_joints:
                if cj.movable:
                    moving.append(cj.q_index)
                else:
                    stack.append(cj)
        m, h, IO = 0.0, np.zeros(3), np.zeros((3, 3))
        for ji in cluster:
            if ji in self.parts:
                _, pm, pc, pI = self.parts[ji]
                m += pm; h += pm * pc; IO += pI + pm * (np.dot(pc, pc) * np.eye(3) - np.outer(pc, pc))
        if m > 0.0:
            c = h / m
            Ic = IO - m * (np.dot(c, c) * np.eye(3) - np.outer(c, c))
        else:
            c, Ic = np.zeros(3), np.zeros((3, 3))
        # every moving link below: the listed ones and their descendants
        sub = set(moving)
        for i, fl in enumerate(self.links):
            if fl.parent in sub:
                sub.add(i)
        return dict(anchor=anchor, rigid=(m, c, 0.5 * (Ic + Ic.T)), moving=sorted(sub), urdf_links=sorted(cluster))

    @property
    def num_dofs(self):
        return len(self.links)

    def frame_id(self, name):
        names = self.robot.joint_names
        return names.index(name) if name in names else -1
