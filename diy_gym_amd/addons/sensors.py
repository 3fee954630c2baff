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


class ForceTorqueSensor(Addon):
    """Reaction wrench across one joint of the parent model: ``force`` and ``torque`` (reference:
    diy_gym/addons/sensors/force_torque_sensor.py:8-23 -- ``enableJointForceTorqueSensor`` + ``getJointState()[2]``).

    What is reported [R: Bullet's joint feedback, ``I^A a + Z^A`` of the child link in its own frame]: the force and
    the torque the PARENT side exerts on the CHILD side through the joint -- gravity, inertial loads and the contact
    forces acting on the child side all show -- expressed in the child link's inertial frame, torque about its
    origin.  Computed in the output phase by Newton-Euler over the child side (the rigid cluster of URDF links behind
    the joint plus every moving link hanging off it), with the accelerations of the step's LAST substep,
    ``(v_end - v_start) / h``, and that substep's contact impulses.  ``frame`` names the joint (fixed or movable); it
    is required: the reference's default of -1 makes ``getJointState`` fail."""
    def __init__(self, parent, config):
        super().__init__(parent, config)
        if 'frame' not in config:
            raise ValueError("force_torque_sensor needs a 'frame' (the reference's default, joint -1, is rejected by pybullet)")
        self.uid = parent.uid
        self.frame_id = parent.get_frame_id(config.get('frame'))
        if self.frame_id < 0:
            raise ValueError('force_torque_sensor: model %r has no joint %r' % (parent.name, config.get('frame')))
        box = lambda: spaces.Box(-10, 10, shape=(3, ), dtype='float32')
        self.observation_space = spaces.Dict(OrderedDict(force=box(), torque=box()))

    def compile(self, builder):
        if self.uid in builder.aliases:
            raise NotImplementedError('force_torque_sensor on an attached child model')
        flat = self.parent.flat
        cl = flat.ft_cluster(self.frame_id)
        m, c, I = cl['rigid']
        whole = self.parent.robot.joints[self.frame_id].movable
        shapes = builder.shapes_of(self.uid, cl['urdf_links'], cl['moving'])
        moving = [builder.global_link(self.uid, d) for d in cl['moving']]
        self.op = builder.add_op(K.OP_OBS_FT, 'obs', body=self.uid, frame=self.frame_id, flags=K.FT_WHOLE_LINK if whole else 0,
                                 ilist=[len(moving)] + moving + [len(shapes)] + shapes,
                                 flist=[m, c[0], c[1], c[2], I[0, 0], I[0, 1], I[0, 2], I[1, 1], I[1, 2], I[2, 2]], io_dim=6,
                                 state_dim=builder.prev_velocity_slots(self.uid))

    def observe(self):
        off = self.op.io_off
        return OrderedDict(force=self.env._obs_view(off, 3), torque=self.env._obs_view(off + 3, 3))


class Camera(Addon):
    """RGB (+ depth, default ON; + segmentation) from a pinhole camera attached to a model frame or fixed in
    the world (reference: diy_gym/addons/sensors/camera.py:26-98).  Same config keys and defaults
    (``clipping_boundaries`` [0.01, 100], ``field_of_view`` 70, ``resolution`` [640, 480], ``frame``, ``xyz``,
    ``rpy``, ``use_depth`` True, ``use_segmentation_mask`` False).  Rendered by its own kernel launch
    (``dg_world_render``), lazily, the first time ``observe()`` is called after a step.

    * ``depth`` is what the reference's formula (:82-85) yields: the eye-space z of the nearest surface,
      i.e. NEGATIVE values in [-far, -near]; -far where the ray hits nothing.
    * ``segmentation_mask`` is ``uid + ((link + 1) << 24)``, -1 for background.
    * ``rgb`` is flat-shaded collision geometry -- not comparable with pybullet's lit visual meshes.
    * Images are the row-major ``height x width`` buffer viewed with shape ``resolution`` (= [w, h]),
      exactly like the reference's ``reshape`` (:77), so non-square images are scrambled there too.
    """
    def __init__(self, parent, config):
        super().__init__(parent, config)
        from ..mathx import Transform, quat_from_euler
        from ..model import Model
        self.near, self.far = config.get('clipping_boundaries', [0.01, 100])
        self.fov = config.get('field_of_view', 70.0)
        self.resolution = list(config.get('resolution', [640, 480]))
        self.aspect = self.resolution[0] / self.resolution[1]
        self.uid = parent.uid if isinstance(parent, Model) else -1
        self.frame_id = parent.get_frame_id(config.get('frame')) if 'frame' in config else -1
        xyz = config.get('xyz', [0., 0., 0.])
        rpy = config.get('rpy', [0., 0., 0.])
        self.use_depth = config.get('use_depth', True)
        self.use_seg_mask = config.get('use_segmentation_mask', False)
        self.T_parent_cam = Transform.from_xyz_quat(xyz, quat_from_euler(rpy))
        sp = OrderedDict(rgb=spaces.Box(0., 1., shape=self.resolution + [3], dtype='float32'))
        if self.use_depth:
            sp['depth'] = spaces.Box(0., 10., shape=self.resolution, dtype='float32')
        if self.use_seg_mask:
            sp['segmentation_mask'] = spaces.Box(0., 10., shape=self.resolution, dtype='float32')
        self.observation_space = spaces.Dict(sp)
        self._tick = None
        self._buffers = None

    def compile(self, builder):
        from ..scene import K as _K
        flags = (_K.CAM_DEPTH if self.use_depth else 0) | (_K.CAM_SEGMENTATION if self.use_seg_mask else 0)
        self.camera_index = builder.add_camera(self.uid, self.frame_id, self.resolution[0], self.resolution[1], flags,
                                               self.T_parent_cam, self.fov, self.near, self.far)

    def observe(self):
        import torch
        env = self.env
        B, (w, h) = env.num_envs, self.resolution
        if self._buffers is None:
            dev = env.device
            self._buffers = (torch.zeros((B, w, h, 3), dtype=torch.float32, device=dev),
                             torch.zeros((B, w, h), dtype=torch.float32, device=dev) if self.use_depth else None,
                             torch.zeros((B, w, h), dtype=torch.int32, device=dev) if self.use_seg_mask else None)
        rgb, depth, seg = self._buffers
        if self._tick != env._tick:
            env.sim.render(self.camera_index, rgb, depth, seg)
            self._tick = env._tick
        obs = OrderedDict(rgb=env._out(rgb))
        if self.use_depth:
            obs['depth'] = env._out(depth)
        if self.use_seg_mask:
            obs['segmentation_mask'] = env._out(seg)
        return obs
