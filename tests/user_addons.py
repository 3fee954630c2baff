"""User addons written in plain Python against the batched ``env.sim`` API -- what a user of the reference ports when the
addon acts on the world from ``update()``.  No ``compile()``: the environment calls the hooks once per step with
``[B, ...]`` values.  Used by the tests that show these equal the compiled ops."""
import torch

from diy_gym_amd import spaces
from diy_gym_amd.addons.addon import Addon


class PyPropellor(Addon):
    """The reference's ``Propellor`` (examples/drone_pilot/drone_pilot.py:10-40) line by line, batched: the two pybullet
    calls become ``sim.apply_external_force`` / ``sim.apply_external_torque`` (or one ``apply_external_wrench``)."""
    one_call = True

    def __init__(self, parent, config):
        super().__init__(parent, config)
        self.uid = parent.uid
        self.frame_id = parent.get_frame_id(config.get('frame'))
        self.max_thrust = config.get('max_thrust', 20.0)
        self.max_torque = config.get('max_torque', 0.1) * (1.0 if config.get('rotor_direction') == 'CCW' else -1.0)
        self.spool_up_rate = 0.1
        self.rotor_speed = None
        self.observation_space = spaces.Box(0.0, 1.0, shape=(1, ), dtype='float32')
        self.action_space = spaces.Box(0.0, 1.0, shape=(1, ), dtype='float32')

    def update(self, action):
        sim = self.env.sim
        a = torch.as_tensor(action, dtype=torch.float32).to(sim.device).reshape(-1)
        if a.numel() == 1:
            a = a.expand(sim.num_envs)
        if self.rotor_speed is None:
            self.rotor_speed = torch.zeros(sim.num_envs, dtype=torch.float32, device=sim.device)
            # (the constants as fp32 tensors: a Python float operand is a double, and which precision the product with it is
            # rounded in is the framework's business -- this way it is one fp32 multiply, like the compiled op's)
            self._k, self._thrust, self._torque = (torch.tensor(v, dtype=torch.float32, device=sim.device) for v in (self.spool_up_rate, self.max_thrust, self.max_torque))
        self.rotor_speed = self.rotor_speed + (a - self.rotor_speed) * self._k
        zero = torch.zeros_like(self.rotor_speed)
        force = torch.stack([zero, zero, self._thrust * self.rotor_speed], dim=1)
        torque = torch.stack([zero, zero, self._torque * self.rotor_speed], dim=1)
        if self.one_call:
            sim.apply_external_wrench(self.uid, self.frame_id, force, [0.0, 0.0, 0.0], torque, sim.LINK_FRAME)
        else:
            sim.apply_external_force(self.uid, self.frame_id, force, [0.0, 0.0, 0.0], sim.LINK_FRAME)
            sim.apply_external_torque(self.uid, self.frame_id, torque, sim.LINK_FRAME)

    def observe(self):
        if self.rotor_speed is None:
            return torch.zeros((self.env.sim.num_envs, 1), dtype=torch.float32, device=self.env.sim.device)
        return self.rotor_speed.reshape(-1, 1)


class PyPropellorTwoCalls(PyPropellor):
    one_call = False


class PyExternalForce(Addon):
    """The reference's ``ExternalForce`` (diy_gym/addons/controllers/external_force.py:9-24) as a Python hook addon."""
    def __init__(self, parent, config):
        super().__init__(parent, config)
        self.uid = parent.uid
        self.xyz = config.get('xyz', [0.0, 0.0, 0.0])
        self.action_space = spaces.Box(-10.0, 10.0, shape=(3, ), dtype='float32')

    def update(self, action):
        sim = self.env.sim
        sim.apply_external_force(self.uid, -1, torch.as_tensor(action, dtype=torch.float32), self.xyz, sim.WORLD_FRAME)
