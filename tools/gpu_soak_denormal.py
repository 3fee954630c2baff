"""Base twists and joint velocities pushed into the range whose squares are denormal floats (1e-16 .. 1e-24 of what they were), in random envs,
every few steps: every parity config must stay finite.  (v_rsq_f32 / v_rcp_f32 answer +inf for denormal arguments.)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
import diy_gym_amd.examples  # noqa: F401
from diy_gym_amd import DIYGym
import test_parity_gpu as T
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
for name in T.CONFIGS:
    B = 256
    env = DIYGym(T.CONFIGS[name], num_envs=B, device='cuda:0', seed=3)
    lo, hi = T.action_bounds(env)
    gen = torch.Generator().manual_seed(1)
    bad_at = None
    for i in range(120):
        act = (lo + (hi - lo) * torch.rand((B, lo.numel()), generator=gen)).to('cuda:0')
        env.sim.step(env._all_slots, act)
        if i % 6 == 5:
            st = np.array(env.sim.get_state())
            if not np.isfinite(st).all():
                bad_at = i; break
            pick = rng.random(B) < 0.5
            scale = 10.0 ** (-rng.uniform(16, 24, size=(B, 1)))
            st2 = st.copy(); L = env.layout
            for b in range(L.n_bodies):                            # base twists
                so = L.body_state_off[b]
                if so >= 0 and not L.body_fixed[b]:
                    st2[:, so + 7:so + 13] = np.where(pick[:, None], st[:, so + 7:so + 13] * scale, st[:, so + 7:so + 13])
            for o in L.link_state_off:                             # joint velocities
                st2[:, o + 1] = np.where(pick, st[:, o + 1] * scale[:, 0], st[:, o + 1])
            env.sim.set_state(st2)
    ok = bad_at is None and bool(torch.isfinite(env.sim.state[:, :B]).all()) and bool(torch.isfinite(env.sim.obs).all())
    print('%-12s lanes %3d: %s' % (name, env.sim.lanes, 'finite' if ok else 'NON-FINITE (first seen at step %s)' % bad_at), flush=True)
