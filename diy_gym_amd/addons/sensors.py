"""Sensor addons, compiled to observe-phase ops of the batched step kernel."""
from collections import OrderedDict

import numpy as np

from .. import spaces
from ..scene import K
from .addon import Addon


class JointStateSensor(Addon):
    """Joint position (+ velocity, default ON; + effort) (reference:
    diy_gym/addons/sensors/joint_state_sensor.py:15-57).  Effort is the motor
    torque applied during the last solver pass (``getJointStates`` item 3)."""
    def __init__(self, parent, config):
        super().__init__(parent, config)
        self.uid = parent.uid
        robot = parent.robot
        if 'joints' in config:
            names = robot.joint_names
            self.joint_ids = [names.index(j) for j in config.get('joints')]
        else:
            self.joint_ids = [j.index for j in robot.joints if j.q_index > -1]
        for j in self.joint_ids:
            if robot.joints[j].q_index < 0:
                raise ValueError('joint_state_sensor: joint %s is fixed and has no state' % robot.joints[j].name)
        self.include_velocity = config.get('include_velocity', True)
        self.include_effort = config.get('include_effort', False)
        info = [robot.joints[j] for j in self.joint_ids]
        sp = OrderedDict(position=spaces.Box(low=np.array([j.lower for j in info]), high=np.array([j.upper for j in info]),
                                             dtype='float32'))
        if self.include_velocity:
            vmax = np.array([j.velocity for j in info])
            sp['velocity'] = spaces.Box(low=-vmax, high=vmax, dtype='float32')
        if self.include_effort:
            tmax = np.array([j.effort for j in info])
            sp['effort'] = spaces.Box(low=-tmax, high=tmax, dtype='float32')
        self.observation_space = spaces.Dict(sp)

    def compile(self, builder):
        dofs = [builder.global_link(self.uid, self.parent.robot.joints[j].q_index) for j in self.joint_ids]
        n = len(dofs)
        flags = (K.JS_VELOCITY if self.include_velocity else 0) | (K.JS_EFFORT if self.include_effort else 0)
        self.op = builder.add_op(K.OP_OBS_JOINT_STATE, 'obs', body=self.uid, flags=flags, ilist=dofs,
                                 io_dim=n * (1 + bool(self.include_velocity) + bool(self.include_effort)))
        self._n = n

    def observe(self):
        env, off, n = self.env, self.op.io_off, self._n
        obs = OrderedDict(position=env._obs_view(off, n))
        k = off + n
        if self.include_velocity:
            obs['velocity'] = env._obs_view(k, n)
            k += n
        if self.include_effort:
            obs['effort'] = env._obs_view(k, n)
        return obs


class ObjectStateSensor(Addon):
    """Pose / twist of a model's base or link, optionally minus a source frame's
    (reference: diy_gym/addons/sensors/object_state_sensor.py:8-83).  Kept quirks:
    the link path reads the *inertial* frame (items 0,1,6,7); with a source the
    subtraction is done in the world frame and ``rotation`` is
    ``q_source (x) q_target``, not a relative rotation; ``angular_velocity`` needs
    both ``include_rotation`` and ``include_velocity``."""
    def __init__(self, parent, config):
        super().__init__(parent, config)
        self.source_model = parent.models[config.get('source_model')] if 'source_model' in config else None
        self.target_model = parent.models[config.get('target_model')] if 'target_model' in config else parent
        self.source_frame_id = self.source_model.get_frame_id(config.get('source_frame')) if 'source_frame' in config else -1
        self.target_frame_id = self.target_model.get_frame_id(config.get('target_frame')) if 'target_frame' in config else -1
        self.include_rotation = config.get('include_rotation', False)
        self.include_velocity = config.get('include_velocity', False)
        box = lambda: spaces.Box(-10, 10, shape=(3, ), dtype='float32')
        sp = OrderedDict(position=box())
        if self.include_rotation:
            sp['rotation'] = box()
        if self.include_velocity:
            sp['velocity'] = box()
        if self.include_rotation and self.include_velocity:
            sp['angular_velocity'] = box()
        self.observation_space = spaces.Dict(sp)

    def compile(self, builder):
        flags = (K.OS_ROTATION if self.include_rotation else 0) | (K.OS_VELOCITY if self.include_velocity else 0)
        n = 3 * (1 + bool(self.include_rotation) + bool(self.include_velocity) +
                 bool(self.include_rotation and self.include_velocity))
        src = self.source_model
        self.op = builder.add_op(K.OP_OBS_OBJECT_STATE, 'obs', body=self.target_model.uid, frame=self.target_frame_id,
                                 body2=src.uid if src is not None else -1, frame2=self.source_frame_id, flags=flags,
                                 io_dim=n)

    def observe(self):
        # dict order as built by the reference's observe(): position, velocity, rotation, angular_velocity
        env, k = self.env, self.op.io_off
        obs = OrderedDict(position=env._obs_view(k, 3))
        k += 3
        if self.include_velocity:
            obs['velocity'] = env._obs_view(k, 3)
            k += 3
        if self.include_rotation:
            obs['rotation'] = env._obs_view(k, 3)
            k += 3
        if self.include_rotation and self.include_velocity:
            obs['angular_velocity'] = env._obs_view(k, 3)
        return obs
