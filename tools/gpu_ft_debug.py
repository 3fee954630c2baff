import os, sys, tempfile, pathlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
from diy_gym_amd import DIYGym
from oracle_backend import OracleBackend
G = os.path.join(ROOT, 'tests', 'golden', 'urdf')
cfg = pathlib.Path(tempfile.mkdtemp()) / 'pend_tool.yaml'
cfg.write_text('render: no\nprop:\n  model: %s\n  use_fixed_base: yes\n  xyz: [-0.7, 0, 0.0]\n'
               'pend:\n  model: %s\n  xyz: [0, 0, 0]\n  wrist: {addon: force_torque_sensor, frame: mount}\n'
               '  shoulder: {addon: force_torque_sensor, frame: hinge}\n  q: {addon: joint_state_sensor}\n'
               % (os.path.join(G, 'ft_prop.urdf'), os.path.join(G, 'pendulum_tool.urdf')))
B = 5
eng = dict(residual_threshold=1e-13)
gpu = DIYGym(str(cfg), num_envs=B, device='cuda:0', engine=eng); cpu = DIYGym(str(cfg), num_envs=B, backend_factory=OracleBackend, engine=eng)
print('lanes', gpu.sim.lanes, gpu.sim.kernel_name)
qo = gpu.layout.link_state_off[0]
st = cpu.sim.get_state(); st[:, qo] = np.linspace(0.9, 1.1, B); cpu.sim.set_state(st); gpu.sim.set_state(st)
m = np.array([[0.0, 1.0, 0.0]]); gpu.sim.set_motor_cfg(m); cpu.sim.set_motor_cfg(m)
d = gpu.sim.enable_diagnostics()
np.set_printoptions(precision=4, suppress=True, linewidth=200)
for step in range(300):
    gpu.sim.step(0); cpu.sim.step(0)
    if step in (2, 150, 299):
        print('step', step, 'contacts gpu', d[:, 0].tolist(), 'cpu', [cpu.sim.contacts(e) for e in range(B)], 'iters', d[:, 1].tolist(), [cpu.sim.iterations(e) for e in range(B)])
        print(' gpu', gpu.sim.obs.cpu().numpy()[:2]); print(' cpu', cpu.sim.obs.numpy()[:2])
