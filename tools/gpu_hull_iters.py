#!/usr/bin/env python3
"""Diagnostic: GJK iterations per lane of the device routine (dg_hull.h, through dg_debug_hull_hull) on pairs of UR5 link hulls at
random relative poses in contact range, next to the fp64 checker's.    python tools/gpu_hull_iters.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import test_hull_contacts as H  # noqa: E402


def main():
    rng = np.random.default_rng(11); hulls = H.ur5_hulls(); L = H.hull_lib('f64')
    dev, ora, dist = [], [], []
    for i, j in ((1, 2), (3, 5), (2, 6), (4, 4), (5, 6), (2, 3)):
        a, b = hulls[i], hulls[j]; poses, Ts = [], []
        for _ in range(512):
            Ta = H.random_pose(rng, spread=1.0); Tb = H.random_pose(rng, Ta[1] + rng.normal(size=3) * rng.choice([0.05, 0.08, 0.12]))
            poses.append(H._pose_rows(Ta, Tb)); Ts.append((Ta, Tb))
        out = H.device_pairs(a, b, np.stack(poses), max_dist=0.022)
        for k, (Ta, Tb) in enumerate(Ts):
            hit, ref, st = H.oracle_pair(L, a, Ta, b, Tb, max_dist=0.022)
            dev.append(out[k, 11]); ora.append(st[0]); dist.append(ref[9] if hit else 9.0)
    dev, ora, dist = np.array(dev), np.array(ora), np.array(dist)
    near = dist < 0.022
    print('all pairs      : device iterations mean %.1f p99 %.0f max %.0f | checker mean %.1f max %.0f' % (dev.mean(), np.percentile(dev, 99), dev.max(), ora.mean(), ora.max()))
    print('in contact range (%d): device mean %.1f p90 %.0f p99 %.0f max %.0f, at the 32-iteration cap: %d | checker mean %.1f max %.0f' % (near.sum(), dev[near].mean(), np.percentile(dev[near], 90), np.percentile(dev[near], 99), dev[near].max(), int((dev[near] >= 32).sum()), ora[near].mean(), ora[near].max()))
    w = dev.reshape(-1, 64).max(1)
    print('slowest lane of a wavefront (64 poses of one pair): mean %.1f max %.0f' % (w.mean(), w.max()))


if __name__ == '__main__':
    main()
