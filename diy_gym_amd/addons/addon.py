"""Addon plugin API: ``AddonFactory``, ``Addon``, ``Receptor``.

This is the drop-in boundary the reference exposes to users (reference:
diy_gym/addons/addon.py:5-81 registry, :91-186 hooks, :189-210 receptor).  The
registry names, the constructor signature ``(parent, config)``, the five hooks
and the ``action_space`` / ``observation_space`` / ``hide`` attributes are the
same.  What is new is one optional method:

``compile(self, builder)``
    Called once, after every model and addon has been constructed, in the same
    (receptor-sorted, addon-sorted) order the reference walks addons.  A
    built-in addon emits its *ops* into the scene's addon program here, so that
    its ``update`` / ``observe`` / ``reward`` / ``is_terminal`` / ``reset`` work
    runs inside the batched HIP step kernel instead of through per-env pybullet
    calls.  An addon without ``compile`` (a user's custom class) keeps working:
    its Python hooks are called once per step with ``[B, ...]`` tensors and may
    read simulation state through ``parent.env.sim``.
"""
from collections import OrderedDict

from .. import spaces


class AddonFactory:
    """Name -> class registry, singleton like the reference's."""
    _instance = None

    class _Registry:
        def __init__(self):
            from .controllers import AdmittanceController, InverseKinematicsController, JointController, ExternalForce
            from .sensors import Camera, ForceTorqueSensor, JointStateSensor, ObjectStateSensor
            from .rewards import ReachTarget, ElectricityCost, TimePenalty
            from .misc import DynamicsRandomizer, Respawn, SpawnMultiple, VisualRandomizer
            from .unsupported import StuckJointCost, DrawCoords
            # same 17 keys as reference addon.py:36-54
            self.addons = {
                'ik_controller': InverseKinematicsController,
                'joint_controller': JointController,
                'admittance_controller': AdmittanceController,
                'camera': Camera,
                'joint_state_sensor': JointStateSensor,
                'object_state_sensor': ObjectStateSensor,
                'force_torque_sensor': ForceTorqueSensor,
                'reach_target': ReachTarget,
                'stuck_joint_cost': StuckJointCost,
                'electricity_cost': ElectricityCost,
                'time_penalty': TimePenalty,
                'respawn': Respawn,
                'spawn_multiple': SpawnMultiple,
                'draw_coords': DrawCoords,
                'external_force': ExternalForce,
                'visual_randomizer': VisualRandomizer,
                'dynamics_randomizer': DynamicsRandomizer,
            }

    @staticmethod
    def get():
        if AddonFactory._instance is None:
            AddonFactory._instance = AddonFactory._Registry()
        return AddonFactory._instance

    @staticmethod
    def build(name, parent, config):
        """``KeyError`` for an unknown name, like the reference (addon.py:77)."""
        return AddonFactory.get().addons[name](parent, config)

    @staticmethod
    def register_addon(name, cls):
        AddonFactory.get().addons[name] = cls


class Addon:
    """Base class.  Hooks default to "nothing to say" (``None``), which the
    environment leaves out of its dictionaries (reference diy_gym.py:216-218)."""
    def __init__(self, parent, config):
        self.parent = parent
        self.action_space = None
        self.observation_space = None
        self.hide = config.get('hide', False)
        self.name = getattr(config, 'name', None)

    # the environment that owns this addon (parent is a Model or the env itself)
    @property
    def env(self):
        return getattr(self.parent, 'env', self.parent)

    def update(self, action):
        pass

    def reset(self):
        pass

    def observe(self):
        pass

    def reward(self):
        pass

    def is_terminal(self):
        pass


class Receptor:
    """Anything addons can be attached to: a model or the environment."""
    def __init__(self):
        self.addons = OrderedDict()

    def build_spaces(self):
        obs_space, act_space = spaces.Dict(OrderedDict()), spaces.Dict(OrderedDict())
        for name, addon in self.addons.items():
            if addon.hide:
                continue
            if addon.observation_space is not None:
                obs_space.spaces[name] = addon.observation_space
            if addon.action_space is not None:
                act_space.spaces[name] = addon.action_space
        return obs_space, act_space
