#!/usr/bin/env python3
"""CPU experiment behind the aged-rollout tail of ur_high_5: the bench's loop -- a ring of 8 random action
batches replayed, masked auto-reset -- on the fp32 OpenMP oracle; prints the Gauss-Seidel iteration histogram of the
last substep every --every steps and, for the envs at the iteration cap, which joints sit on a limit.
    python tools/aged_tail_oracle.py [--envs 1024] [--steps 4500] [--engine motor_guess=1.0,limit_guess=1.0]"""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--envs', type=int, default=1024)
    ap.add_argument('--steps', type=int, default=4500)
    ap.add_argument('--every', type=int, default=500)
    ap.add_argument('--engine', default='')
    ap.add_argument('--config', default='examples/ur_high_5/ur_high_5.yaml')
    args = ap.parse_args()
    os.environ.setdefault('OMP_NUM_THREADS', str(os.cpu_count()))
    import oracle_backend
    from diy_gym_amd import DIYGym
    from diy_gym_amd.utils import flatten, get_bounds_for_space
    engine = {k: float(v) for k, v in (kv.split('=') for kv in args.engine.split(',') if kv)}
    B = args.envs
    env = DIYGym(os.path.join(ROOT, args.config), num_envs=B, seed=1234, backend_factory=oracle_backend.flavour(os.environ.get('DG_ORACLE_FLAVOUR', 'f32_omp')), engine=engine)
    lo = torch.as_tensor(flatten(get_bounds_for_space(env.action_space, True)), dtype=torch.float32)
    hi = torch.as_tensor(flatten(get_bounds_for_space(env.action_space, False)), dtype=torch.float32)
    gen = torch.Generator().manual_seed(1234)
    ring = [lo + (hi - lo) * torch.rand((B, lo.numel()), generator=gen) for _ in range(8)]
    sim = env.sim
    L = env.layout
    t0 = time.time()
    for s in range(args.steps):
        sim.step(env._all_slots, ring[s % 8])
        sim.reset(sim.term_flag)
        if (s + 1) % args.every == 0 or s == args.steps - 1:
            it = np.array([sim.iterations(e) for e in range(B)])
            print('step %5d  iters mean %.1f p50 %d p90 %d p99 %d max %d  at cap: %d   per-64 max mean %.1f  (%.0f s)' % (
                s + 1, it.mean(), np.median(it), np.percentile(it, 90), np.percentile(it, 99), it.max(), int((it >= 150).sum()),
                it.reshape(-1, 64).max(1).mean() if B % 64 == 0 else -1, time.time() - t0), flush=True)
    st = np.asarray(sim.get_state())
    it = np.array([sim.iterations(e) for e in range(B)])
    worst = np.argsort(-it)[:6]
    F = L.F; from diy_gym_amd.scene import K
    lf = F[L.I[K.H_OFF_LINK_F]:].reshape(-1)[:L.n_links * K.LF_STRIDE].reshape(L.n_links, K.LF_STRIDE)
    for e in worst:
        q = np.array([st[e, o + K.LS_Q] for o in L.link_state_off]); tp = np.array([st[e, o + K.LS_TARGET_POS] for o in L.link_state_off])
        ap_ = np.array([st[e, o + K.LS_APPLIED] for o in L.link_state_off])
        lim = [(j, 'lo' if q[j] - lf[j, K.LF_LOWER] < 0.01 else 'hi') for j in range(L.n_links) if lf[j, K.LF_LOWER] <= lf[j, K.LF_UPPER] and (q[j] - lf[j, K.LF_LOWER] < 0.01 or lf[j, K.LF_UPPER] - q[j] < 0.01)]
        print('env %4d iters %3d  joints on a limit %s  target-q there %s  effort/max of every joint %s' % (e, it[e], lim, [round(float(tp[j] - q[j]), 4) for j, _ in lim],
              [round(float(ap_[j] / max(lf[j, K.LF_MAX_FORCE], 1e-9)), 2) for j in range(L.n_links)]))


if __name__ == '__main__':
    main()
