"""Reset-time addons."""
from ..scene import K
from .addon import Addon


class Respawn(Addon):
    """On reset put the base back at its load pose plus uniform jitter
    (reference: diy_gym/addons/misc/respawn.py:7-39).  The reference draws from
    the global numpy RNG; here every env draws from its own counter-based stream
    keyed by (seed, env index, episode), so shards are reproducible."""
    def __init__(self, parent, config):
        super().__init__(parent, config)
        self.uid = parent.uid
        self.initial_pose = (list(parent.position), list(parent.orientation))
        self.position_range = list(config.get('position_range', [0., 0., 0.]))
        self.rotation_range = list(config.get('rotation_range', [0., 0., 0.]))
        self.once = config.get('once', False)

    def compile(self, builder):
        fp = self.initial_pose[0] + self.initial_pose[1] + self.position_range + self.rotation_range
        self.op = builder.add_op(K.OP_RESPAWN, 'reset', body=self.uid, flags=K.RS_ONCE if self.once else 0, fparams=fp)


class SpawnMultiple(Addon):
    """Clones a model ``num_models`` times into ``parent.models`` (reference:
    diy_gym/addons/misc/spawn_multiple.py:6-12).  Like there, every clone is a root-level body built from the same
    config (same pose, same ``name``; only the dictionary key ``<name>_<i>`` differs)."""
    def __init__(self, parent, config):
        super().__init__(parent, config)
        from ..model import Model
        child_config = config.find('model')
        for i in range(config.get('num_models')):
            parent.models[child_config.name + '_%d' % i] = Model(child_config, env=self.env)

    def compile(self, builder):
        pass


class DynamicsRandomizer(Addon):
    """Per-env link masses and body angular damping re-drawn at every reset (reference:
    diy_gym/addons/misc/dynamics_randomizer.py:8-32; the sim-to-real randomisation the README advertises).

    The reference formula, kept literally: for every movable joint ``k`` of the parent model, in joint order,

    * ``mass_k <- log(U(mass_range)) * getDynamicsInfo(uid, k)[0]`` -- it reads the CURRENT mass back, so the factors
      compound from reset to reset;
    * ``changeDynamics(angularDamping = log(U(damping_range)) * getJointInfo(uid, k)[6])`` -- pybullet's
      ``angularDamping`` is a property of the whole multibody [R], so the draw of the last joint is what stays; the
      second factor is the URDF joint damping (0 for most robot descriptions, which switches Bullet's default
      angular damping of 0.04 off).

    ``log(U(0.25, 4))`` is negative for a fifth of the draws; a negative link mass has no defined behaviour in
    Bullet.  Guards chosen here (``mass_scale_limits`` is an extension key): the mass factor is ``|log U|``, the
    accumulated scale relative to the URDF mass is clamped to ``mass_scale_limits`` (default [1e-3, 1e3]); the
    angular damping is clamped at 0.  A link's inertia tensor is scaled with its mass (same convention as the
    ``mass`` key of a model).  The reference draws once at construction and once more in the constructor's
    ``reset()``; so does this addon at an env's first reset.  Draws come from the per-env counter-based stream
    (seed, global env index, episode) instead of python's global ``random``.
    """
    def __init__(self, parent, config):
        super().__init__(parent, config)
        self.uid = parent.uid
        robot = parent.robot
        self.joint_ids = [j.index for j in robot.joints if j.q_index > -1]
        if not self.joint_ids:
            # the reference falls back to joint -1 and then fails in p.getJointInfo(uid, -1) (dynamics_randomizer.py:30)
            raise ValueError('dynamics_randomizer: model %r has no movable joint (the reference raises here too: '
                             'getJointInfo(uid, -1) is out of range)' % parent.name)
        self.mass_range = [float(v) for v in config.get('mass_range', [0.25, 4.0])]
        self.damping_range = [float(v) for v in config.get('damping_range', [0.2, 20])]
        self.mass_scale_limits = [float(v) for v in config.get('mass_scale_limits', [1e-3, 1e3])]
        if min(self.mass_range) <= 0 or min(self.damping_range) <= 0:
            raise ValueError('dynamics_randomizer: ranges must be positive (log of the draw is taken)')

    def compile(self, builder):
        robot = self.parent.robot
        dofs = [builder.global_link(self.uid, robot.joints[j].q_index) for j in self.joint_ids]
        self.op = builder.add_op(K.OP_RANDOMIZE_DYNAMICS, 'reset', body=self.uid, ilist=dofs,
                                 flist=[robot.joints[j].damping for j in self.joint_ids],
                                 fparams=self.mass_range + self.damping_range + self.mass_scale_limits, state_dim=len(dofs) + 1)

    def mass_scales(self):
        """[B, n_joints] current mass scale of every randomised link (host copy, for inspection / tests)."""
        import torch
        o = self.env.layout.addon_off + self.op.state_off
        return torch.as_tensor(self.env.sim.get_state()[:, o:o + len(self.joint_ids)])

    def angular_damping(self):
        import torch
        o = self.env.layout.addon_off + self.op.state_off + len(self.joint_ids)
        return torch.as_tensor(self.env.sim.get_state()[:, o])


class VisualRandomizer(Addon):
    """A new look for the parent model at every reset (reference: diy_gym/addons/misc/visual_randomizer.py:14-46).

    The reference picks a random TEXTURE of the "describable textures" data set for every link (``p.loadTexture`` /
    ``p.changeVisualShape``) and downloads the 600 MB data set over HTTP when it is missing (:48-77).  The data set is
    out of scope; what is randomised here is a PROCEDURAL texture of the model -- two colours, a frequency and a pattern
    (checker, stripes or per-cell blends, ``DG_TEX_*`` in diygym_scene.h) per env and episode from the per-env
    counter-based stream, drawn at construction and at every reset like the reference does, evaluated in the shapes'
    own frames (the pattern sticks to the object) and visible in the camera addon's ``rgb`` only.  Depth and
    segmentation are unaffected.  Deviation: one texture per model, not one per link."""
    def __init__(self, parent, config):
        super().__init__(parent, config)
        self.uid = parent.uid

    def compile(self, builder):
        self.op = builder.add_op(K.OP_RANDOMIZE_COLOR, 'reset', body=self.uid, state_dim=K.TX_STRIDE)

    def textures(self):
        """[B, 8] current texture of the model in every env: colour A, colour B, frequency, kind (host copy)."""
        import torch
        o = self.env.layout.addon_off + self.op.state_off
        return torch.as_tensor(self.env.sim.get_state()[:, o:o + K.TX_STRIDE])

    def colors(self):
        """[B, 3] colour A of the model's texture in every env (host copy)."""
        return self.textures()[:, :3]
