#!/usr/bin/env python3
"""Diagnostic: what a GJK iteration costs a lone wavefront.  65 536 contact-range poses of one pair of UR5 link hulls = 1 024
wavefronts, one per SIMD, through dg_debug_hull_hull; run under `rocprofv3 --kernel-trace --stats` and divide hull_pair_kernel's
duration by the mean of the wavefronts' slowest lanes (printed here).    python tools/gpu_hull_time.py [max_dist]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import test_hull_contacts as H  # noqa: E402

rng = np.random.default_rng(3); hulls = H.ur5_hulls(); a, b = hulls[2], hulls[6]
max_dist = float(sys.argv[1]) if len(sys.argv) > 1 else 0.022
poses = []
while len(poses) < 65536:
    Ta = H.random_pose(rng, spread=1.0); Tb = H.random_pose(rng, Ta[1] + rng.normal(size=3) * 0.08)
    poses.append(H._pose_rows(Ta, Tb))
poses = np.stack(poses)
out = H.device_pairs(a, b, poses, max_dist=max_dist)
if os.environ.get('SHALLOW'):  # keep the poses the polytope search does not decide (resampled to the same count), five more runs
    keep = ~((out[:, 10] > 0) & (out[:, 9] < 0.0005))
    poses = poses[keep][np.arange(len(poses)) % int(keep.sum())]
    for rep in range(5):
        out = H.device_pairs(a, b, poses, max_dist=max_dist)
it = out[:, 11].reshape(-1, 64)
print('poses %d, hit %.2f, deep (dist < -0.002) %.3f | iterations: lane mean %.2f, slowest lane of a wavefront mean %.2f max %d' % (
    len(poses), out[:, 10].mean(), float((out[out[:, 10] > 0, 9] < -0.002).mean()), it.mean(), it.max(1).mean(), it.max()))
