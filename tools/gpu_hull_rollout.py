#!/usr/bin/env python3
"""Diagnostic: ur_arms_touching with hull contacts, GPU against the fp64 checker, free-running and teacher-forced (the GPU restarted
from the checker's state before every step): per-step largest observation difference and contact counts.
    python tools/gpu_hull_rollout.py [--steps 30] [--envs 8] [--engine hull_contacts=0]"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))


def main():
    ap = argparse.ArgumentParser(); ap.add_argument('--steps', type=int, default=30); ap.add_argument('--envs', type=int, default=8)
    ap.add_argument('--scene', default='tests/golden/ur_arms_touching.yaml'); ap.add_argument('--engine', default=''); ap.add_argument('--scale', type=float, default=0.3); ap.add_argument('--press', action='store_true', help='hold the rest pose and turn the first shoulder joint by 0.06 .. 0.14 rad (per env): the forearms press against each other')
    a = ap.parse_args()
    engine = {k: float(v) for k, v in (kv.split('=') for kv in a.engine.split(',') if kv)}
    from diy_gym_amd import DIYGym
    from oracle_backend import OracleBackend
    cfg = os.path.join(ROOT, a.scene); B = a.envs
    free = DIYGym(cfg, num_envs=B, device='cuda:0', seed=5, engine=engine); forced = DIYGym(cfg, num_envs=B, device='cuda:0', seed=5, engine=engine)
    cpu = DIYGym(cfg, num_envs=B, seed=5, backend_factory=OracleBackend, engine=engine)
    diag = free.sim.enable_diagnostics(); dforced = forced.sim.enable_diagnostics()
    gen = torch.Generator().manual_seed(2)
    for step in range(a.steps):
        act = (torch.rand((B, free.layout.act_dim), generator=gen) * 2 - 1) * a.scale
        if a.press:
            act = torch.tensor([1.35, -1.08, 1.03, -0.01, 0.09, 0.86] * 2)[None].repeat(B, 1); act[:, 0] += 0.06 + 0.08 * torch.arange(B) / max(B - 1, 1)
        forced.sim.set_state(np.asarray(cpu.sim.get_state(), dtype=np.float32))
        for e in (free, forced):
            e.sim.step(e._all_slots, act.to('cuda:0'))
        cpu.sim.step(cpu._all_slots, act)
        ef = (free.sim.obs.cpu() - cpu.sim.obs).abs().max(1).values.numpy(); et = (forced.sim.obs.cpu() - cpu.sim.obs).abs().max(1).values.numpy()
        cc = [cpu.sim.contacts(e) for e in range(B)]
        imp = max([cpu.sim.contact(e, k)[7] for e in range(B) for k in range(cc[e])] + [0.0])
        print('step %2d free %.2e forced %.2e | contacts cpu %s gpu-forced %s | max|obs| %.1f imp %.3f sweeps %s' % (step, ef.max(), et.max(), cc, dforced[:, 0].tolist(), float(cpu.sim.obs.abs().max()), imp, [cpu.sim.iterations(e) for e in range(B)]))


if __name__ == '__main__':
    main()
