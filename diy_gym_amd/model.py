"""``Model``: one URDF body of the scene (reference: diy_gym/model.py:11-106).

Same config keys (``model``, ``xyz``, ``rpy``, ``scale``, ``use_fixed_base``,
``mass``, ``color``) and the same receptor role; instead of ``p.loadURDF`` the
URDF is parsed and flattened on the host and registered with the environment's
:class:`~diy_gym_amd.scene.SceneBuilder`.  ``uid`` is the body index in the
scene (pybullet hands out body ids in load order too).
"""
import os
from collections import OrderedDict

from . import mesh
from .addons.addon import AddonFactory, Receptor
from .mathx import quat_from_euler
from .urdf import FlatBody, UrdfRobot

_PKG_DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'data')


def urdf_search_path():
    """Same three-entry idea as the reference (model.py:8): the path as given,
    the package data directory, then a pybullet_data directory -- here the
    stand-ins authored under ``data/pybullet_data`` (plane, sphere2, ...),
    or the real one when ``DIYGYM_PYBULLET_DATA`` points to it."""
    extra = os.environ.get('DIYGYM_PYBULLET_DATA')
    paths = ['', _PKG_DATA]
    if extra:
        paths.append(extra)
    paths.append(os.path.join(_PKG_DATA, 'pybullet_data'))
    return paths


class Model(Receptor):
    def __init__(self, config, parent=None, env=None):
        Receptor.__init__(self)
        self.env = env if env is not None else getattr(parent, 'env', None)
        if self.env is None:
            raise ValueError('Model needs the environment it belongs to')
        self.name = config.name
        self.position = [float(v) for v in config.get('xyz', [0., 0., 0.])]
        self.orientation = [float(v) for v in quat_from_euler(config.get('rpy', [0., 0., 0.]))]
        use_fixed_base = config.get('use_fixed_base', False)
        scale = config.get('scale', 1.0)
        urdf = config.get('model')

        full = None
        for root in urdf_search_path() + [getattr(self.env, 'config_dir', '')]:
            cand = os.path.join(root, urdf)
            if os.path.isfile(cand):
                full = cand
                break
        if full is None:
            raise ValueError('Could not find URDF: ' + urdf)  # reference model.py:63

        self.robot = UrdfRobot(full)
        if parent is not None and 'child_frame' in config:
            # Child model attached by one of its LINKS (reference model.py:71-77: createConstraint pins the child link of
            # joint `child_frame` to the parent frame).  The child is described from that link instead of its URDF root
            # (UrdfRobot.rerooted: same mechanism, same joint coordinates); the rest of it hangs from there.  Joint
            # indices of the child then follow the re-rooted tree.
            names = self.robot.joint_names
            if config.get('child_frame') not in names:
                raise ValueError('child_frame: model %r has no joint %r' % (config.name, config.get('child_frame')))
            self.robot = self.robot.rerooted(self.robot.joints[names.index(config.get('child_frame'))].child)
        self.flat = FlatBody(self.robot, scale=scale, fixed_base=use_fixed_base,
                             mass_override=config.get('mass') if 'mass' in config else None,
                             mesh_loader=mesh.load_convex, max_hull_points=self.env.max_hull_points)
        if parent is None:
            self.uid = self.env.builder.add_body(self.flat, self.position, self.orientation)
        else:
            # Child model (reference model.py:69-77): the reference loads it as its own body and couples it to the
            # parent with a fixed constraint whose pivot is this model's xyz / rpy in the parent frame.  Here the
            # coupling is rigid -- the child is merged into the parent's body (FlatBody.attach) and ``uid`` is an
            # alias the scene builder resolves to (parent body, link / frame offsets).
            # ``attach: constraint`` keeps the reference's arrangement instead: the child is a body of its own and the
            # fixed constraint becomes six solver rows (``constraint_max_force``: createConstraint's default 500 N [R]).
            parent_frame_id = parent.get_frame_id(config.get('parent_frame')) if 'parent_frame' in config else -1
            how = config.get('attach', 'merge')
            if how not in ('merge', 'constraint'):
                raise ValueError("attach: 'merge' or 'constraint', not %r" % (how, ))
            if how == 'constraint':
                self.uid = self.env.builder.add_constrained_child(parent.uid, parent_frame_id, self.flat, self.position, self.orientation,
                                                                  config.get('constraint_max_force', 500.0))
            else:
                self.uid = self.env.builder.attach_child(parent.uid, parent_frame_id, self.flat, self.position, self.orientation)
        self.color = config.get('color') if 'color' in config else None  # visual only (camera rgb)
        if self.color is not None:
            self.env.builder.set_color(self.uid, list(self.color) + [1.0] * (4 - len(self.color)))

        self.addons = OrderedDict(
            sorted(((child.name, AddonFactory.build(child.get('addon'), self, child)) for child in config.find_all('addon')),
                   key=lambda kv: kv[0]))
        # child models (reference model.py:90-92).  As in the reference they are NOT receptors: their own addons are
        # constructed but never stepped; they exist physically and can be named by the parent's sensors
        # (``source_model`` / ``target_model``).
        self.models = OrderedDict(sorted(((child.name, Model(child, parent=self)) for child in config.find_all('model')),
                                         key=lambda kv: kv[0]))

    def get_frame_id(self, frame):
        """Joint index of the named frame, -1 when absent (reference model.py:94-96)."""
        return self.flat.frame_id(frame)

    def get_transform(self, frame_id=-1):
        """World pose of the URDF link frame (``getLinkState`` items 4, 5) or of the
        base (reference model.py:98-106) for every env: ``(xyz [B,3], quat [B,4])``."""
        body, _, foff, basef = self.env.builder.resolve(self.uid)
        if body != self.uid:  # attached child: its frames live in the parent's body
            st = self.env.sim.frame_state(body, basef if frame_id < 0 else foff + frame_id, com=frame_id < 0)
        else:
            st = self.env.sim.frame_state(self.uid, frame_id, com=frame_id < 0)
        return st[:, 0:3], st[:, 3:7]
