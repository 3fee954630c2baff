"""Reward / terminal addons, compiled to reward- and terminal-phase ops."""
import math

from ..scene import K
from .addon import Addon


class ReachTarget(Addon):
    """``-|target - source| * multiplier`` reward, terminal when closer than
    ``tolerance`` (reference: diy_gym/addons/rewards/reach_target.py:7-36; link
    positions are the URDF link frame, item 4; bases the reported base position)."""
    def __init__(self, parent, config):
        super().__init__(parent, config)
        self.source_model = parent.models[config.get('source_model')]
        self.target_model = parent.models[config.get('target_model')]
        self.source_frame_id = self.source_model.get_frame_id(config.get('source_frame')) if 'source_frame' in config else -1
        self.target_frame_id = self.target_model.get_frame_id(config.get('target_frame')) if 'target_frame' in config else -1
        self.multiplier = config.get('multiplier', 1.0)
        self.tolerance = config.get('tolerance', 0.05)

    def compile(self, builder):
        kw = dict(body=self.target_model.uid, frame=self.target_frame_id, body2=self.source_model.uid,
                  frame2=self.source_frame_id, fparams=[self.multiplier, self.tolerance], io_dim=1)
        self.rew_op = builder.add_op(K.OP_REW_REACH, 'rew', **kw)
        self.term_op = builder.add_op(K.OP_TERM_REACH, 'term', group=id(self.parent), **kw)

    def reward(self):
        return self.env._rew_view(self.rew_op.io_off)

    def is_terminal(self):
        return self.env._term_view(self.term_op.io_off)


class ElectricityCost(Addon):
    """``-sum |motor torque * joint velocity| * multiplier`` over the movable
    joints (reference: diy_gym/addons/rewards/electricity_cost.py:8-18)."""
    def __init__(self, parent, config):
        super().__init__(parent, config)
        self.uid = parent.uid
        self.multiplier = config.get('multiplier', 1.0)
        self.joint_ids = [j.index for j in parent.robot.joints if j.q_index > -1]

    def compile(self, builder):
        self.rew_op = builder.add_op(K.OP_REW_ELECTRICITY, 'rew', body=self.uid, fparams=[self.multiplier], io_dim=1)

    def reward(self):
        return self.env._rew_view(self.rew_op.io_off)


class TimePenalty(Addon):
    """Constant reward per step (reference: diy_gym/addons/rewards/time_penalty.py:7-12)."""
    def __init__(self, parent, config):
        super().__init__(parent, config)
        self.penalty = config.get('penalty', -1)

    def compile(self, builder):
        self.rew_op = builder.add_op(K.OP_REW_CONST, 'rew', fparams=[self.penalty], io_dim=1)

    def reward(self):
        return self.env._rew_view(self.rew_op.io_off)


class FellOver(Addon):
    """Terminal when the base is tilted more than 10 degrees (reference:
    examples/drone_pilot/drone_pilot.py:43-55, a user addon there)."""
    def __init__(self, parent, config):
        super().__init__(parent, config)
        self.uid = parent.uid
        self.max_tilt = math.radians(config.get('max_tilt_degrees', 10))

    def compile(self, builder):
        self.term_op = builder.add_op(K.OP_TERM_TILT, 'term', body=self.uid, fparams=[self.max_tilt], io_dim=1,
                                      group=id(self.parent))

    def is_terminal(self):
        return self.env._term_view(self.term_op.io_off)
