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
