"""Where does a step spend its cycles?  Runs the stamped (diagnostic) step kernel and prints section shares."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
import diy_gym_amd.examples
from diy_gym_amd import DIYGym
import test_parity_gpu as T
name = sys.argv[1] if len(sys.argv) > 1 else 'ur_ik'
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
eng = {kv.split('=')[0]: float(kv.split('=')[1]) for kv in os.environ.get('ENGINE', '').split(',') if kv}  # e.g. ENGINE=motor_guess=0,warmstart=0
env = DIYGym(T.CONFIGS[name], num_envs=B, device='cuda:0', engine=eng)
if os.environ.get('CROSSED'):  # ur_ik from the crossed-forearms pose (bench.py's in_contact leg)
    from diy_gym_amd.scene import K
    for arm in range(2):
        for j, q in enumerate([1.35, -1.08, 1.03, -0.01, 0.09, 0.86]):
            o = env.layout.link_state_off[6 * arm + j]
            env.sim.state[o + K.LS_Q, :] = q; env.sim.state[o + K.LS_QD, :] = 0.0; env.sim.state[o + K.LS_TARGET_POS, :] = q
lo, hi = T.action_bounds(env)
gen = torch.Generator().manual_seed(1)
ring = [(lo + (hi - lo) * torch.rand((B, lo.numel()), generator=gen)).to('cuda:0') for _ in range(8)]
for i in range(int(os.environ.get("SETTLE", "10"))): env.sim.step(env._all_slots, ring[i % 8] * float(os.environ.get("ACT_SCALE", "1")))
cyc = env.sim.enable_stamps()
tot = torch.zeros(len(env.sim.SECTIONS) + 12, dtype=torch.float64)
scale = float(os.environ.get('ACT_SCALE', '1'))
ring = [r * scale for r in ring]
n = 16
for i in range(n):
    env.sim.step(env._all_slots, ring[i % 8]); torch.cuda.synchronize()
    tot += cyc.double().mean(0).cpu()
tot /= n
waves = tot[len(env.sim.SECTIONS):].tolist(); tot = tot[:len(env.sim.SECTIONS)]
s = float(tot.sum())
print('%s B=%d: %.0f cycles per wave per step (median wave)' % (name, B, s))
for k, v in zip(env.sim.SECTIONS, tot.tolist()):
    print('  %-13s %9.0f  %5.1f %%' % (k, v, 100 * v / s))
if any(waves):
    for w in range(3):
        print('  wavefront %d reached: pose hand-over %.0f | end of update phase %.0f | final hand-over %.0f | its end %.0f' % ((w + 1,) + tuple(waves[4 * w:4 * w + 4])))
env.sim.enable_stamps(False)
d = env.sim.enable_diagnostics()
for i in range(4): env.sim.step(env._all_slots, ring[i % 8])
torch.cuda.synchronize()
it = d[:, 1].float()
wave_max = it.reshape(-1, env.sim.envs_per_wave).max(1).values
print('PGS iterations (last substep): mean %.1f  p50 %.0f  p99 %.0f  max %.0f | per-wave max: mean %.1f max %.0f | contacts max %d' % (
    it.mean(), it.median(), it.quantile(0.99), it.max(), wave_max.mean(), wave_max.max(), int(d[:, 0].max())))
