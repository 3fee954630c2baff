#!/usr/bin/env python3
"""How far is the geometry the camera draws -- the convex hulls of the COLLISION meshes -- from the hulls of the VISUAL meshes that
pybullet's renderer draws?  For every link of the robot descriptions that has both, the largest difference of the two hulls'
support functions over 4 000 directions (a Hausdorff-type distance between convex bodies), in millimetres and relative to the
link's size.  Reads the third-party mesh files of a pybullet-style data tree (not part of this repo):
    python tools/visual_vs_collision_hulls.py /path/to/data > profiles/r4_visual_vs_collision_hulls.txt"""
import os
import sys
import xml.etree.ElementTree as ET

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from diy_gym_amd import mesh  # noqa: E402
from diy_gym_amd.urdf import resolve_mesh  # noqa: E402

ASSETS = ['ur5/ur5_robot.urdf', 'jaco/j2s7s300_standalone.urdf', 'hector_quadrotor/quadrotor.urdf', 'robotiq_2f/gripper.urdf', 'robotiq_3f/gripper.urdf']


def rpy(r, p, y):
    cr, sr, cp, sp, cy, sy = np.cos(r), np.sin(r), np.cos(p), np.sin(p), np.cos(y), np.sin(y)
    return np.array([[cy * cp, cy * sp * sr - sy * cr, cy * sp * cr + sy * sr], [sy * cp, sy * sp * sr + cy * cr, sy * sp * cr - cy * sr], [-sp, cp * sr, cp * cr]])


def points_of(elem, base):
    g = elem.find('geometry')
    m = g.find('mesh') if g is not None else None
    if m is None:
        return None
    path = resolve_mesh(base, m.get('filename'))
    if not os.path.isfile(path):
        return None
    v = mesh.read_vertices(path) * np.array([float(t) for t in m.get('scale', '1 1 1').split()])[None, :]
    o = elem.find('origin')
    xyz = np.array([float(t) for t in (o.get('xyz', '0 0 0') if o is not None else '0 0 0').split()])
    r = [float(t) for t in (o.get('rpy', '0 0 0') if o is not None else '0 0 0').split()]
    return v @ rpy(*r).T + xyz


def main():
    src = sys.argv[1]
    rng = np.random.RandomState(0)
    D = rng.normal(size=(4000, 3)); D /= np.linalg.norm(D, axis=1)[:, None]
    print('link: support-function gap between hull(visual mesh) and hull(collision mesh), max over 4000 directions')
    worst = 0.0
    for rel in ASSETS:
        path = os.path.join(src, rel)
        if not os.path.isfile(path):
            continue
        root = ET.parse(path).getroot()
        for le in root.findall('link'):
            vis = [points_of(e, os.path.dirname(path)) for e in le.findall('visual')]
            col = [points_of(e, os.path.dirname(path)) for e in le.findall('collision')]
            vis = [p for p in vis if p is not None]; col = [p for p in col if p is not None]
            if not vis or not col:
                continue
            V, C = np.concatenate(vis), np.concatenate(col)
            gap = np.abs((V @ D.T).max(0) - (C @ D.T).max(0)).max()
            size = np.linalg.norm(C.max(0) - C.min(0))
            worst = max(worst, gap / size)
            print('%-28s %-26s gap %7.2f mm  = %5.2f %% of the link\'s extent (%.0f mm)' % (rel.split('/')[0], le.get('name'), gap * 1e3, 100 * gap / size, size * 1e3))
    print('worst relative gap: %.2f %%' % (100 * worst))


if __name__ == '__main__':
    main()
