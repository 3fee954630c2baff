"""Collision-mesh readers: STL / OBJ / DAE -> convex vertex cloud.

Bullet turns a URDF ``<mesh>`` collision element into a convex hull of the mesh
vertices (btConvexHullShape) [RECOLLECTION].  The device kernels collide convex
vertex clouds, so at load time every mesh is reduced to the vertices of its
convex hull, and -- to bound LDS/constant storage -- thinned to at most
``max_points`` support points chosen as the extreme vertices along a fixed
spherical direction set (the thinned hull is inscribed in the true hull; the
worst-case support-distance error is reported by :func:`hull_error`).
"""
import os
import struct
import xml.etree.ElementTree as ET

import numpy as np


def _read_stl(path):
    with open(path, 'rb') as fh:
        data = fh.read()
    if len(data) >= 84:
        ntri = struct.unpack_from('<I', data, 80)[0]
        if 84 + 50 * ntri == len(data):
            rec = np.frombuffer(data, dtype=np.dtype([('n', '<f4', 3), ('v', '<f4', (3, 3)), ('a', '<u2')]),
                                count=ntri, offset=84)
            return rec['v'].reshape(-1, 3).astype(np.float64)
    verts = []
    for line in data.decode('ascii', errors='ignore').splitlines():
        parts = line.split()
        if len(parts) == 4 and parts[0] == 'vertex':
            verts.append([float(parts[1]), float(parts[2]), float(parts[3])])
    return np.array(verts, dtype=np.float64)


def _read_obj(path):
    verts = []
    with open(path, 'r', errors='ignore') as fh:
        for line in fh:
            if line.startswith('v '):
                parts = line.split()
                verts.append([float(parts[1]), float(parts[2]), float(parts[3])])
    return np.array(verts, dtype=np.float64)


def _read_dae(path):
    root = ET.parse(path).getroot()
    ns = root.tag[:root.tag.index('}') + 1] if root.tag.startswith('{') else ''
    unit = 1.0
    u = root.find('%sasset/%sunit' % (ns, ns))
    if u is not None and u.get('meter'):
        unit = float(u.get('meter'))
    clouds = []
    for mesh in root.iter(ns + 'mesh'):
        verts_elem = mesh.find(ns + 'vertices')
        if verts_elem is None:
            continue
        src_id = None
        for inp in verts_elem.findall(ns + 'input'):
            if inp.get('semantic') == 'POSITION':
                src_id = inp.get('source', '').lstrip('#')
        for src in mesh.findall(ns + 'source'):
            if src.get('id') == src_id:
                fa = src.find(ns + 'float_array')
                if fa is not None and fa.text:
                    clouds.append(np.array(fa.text.split(), dtype=np.float64).reshape(-1, 3) * unit)
    if not clouds:
        return np.zeros((0, 3))
    return np.concatenate(clouds, axis=0)


def read_vertices(path):
    ext = os.path.splitext(path)[1].lower()
    if ext == '.stl':
        return _read_stl(path)
    if ext == '.obj':
        return _read_obj(path)
    if ext == '.dae':
        return _read_dae(path)
    raise ValueError('unsupported mesh format: ' + path)


def _directions(n):
    """Deterministic, roughly uniform directions (Fibonacci sphere) plus the axes."""
    k = np.arange(n, dtype=np.float64) + 0.5
    phi = np.arccos(1.0 - 2.0 * k / n)
    theta = np.pi * (1.0 + 5.0**0.5) * k
    d = np.stack([np.cos(theta) * np.sin(phi), np.sin(theta) * np.sin(phi), np.cos(phi)], axis=1)
    axes = np.array([[1, 0, 0], [-1, 0, 0], [0, 1, 0], [0, -1, 0], [0, 0, 1], [0, 0, -1]], dtype=np.float64)
    return np.concatenate([axes, d], axis=0)


def convex_points(verts, max_points=32):
    verts = np.unique(np.round(np.asarray(verts, dtype=np.float64), 9), axis=0)
    if len(verts) > 4:
        try:
            from scipy.spatial import ConvexHull
            verts = verts[ConvexHull(verts).vertices]
        except Exception:
            pass
    if len(verts) <= max_points:
        return verts
    chosen = None
    ndir = max(max_points - 6, 1)
    while True:
        idx = np.unique(np.argmax(verts @ _directions(ndir).T, axis=0))
        if len(idx) > max_points:
            if chosen is None:
                chosen = idx[:max_points]
            break
        chosen = idx
        if len(idx) == max_points or ndir > 64 * max_points:
            break
        ndir += max(1, max_points // 4)
    return verts[np.sort(chosen)]


def hull_error(full, thin):
    """Largest support-function gap between two vertex sets over 512 directions."""
    d = _directions(512)
    return float(np.max(np.max(full @ d.T, axis=0) - np.max(thin @ d.T, axis=0)))


_cache = {}


def load_convex(path, max_points=32):
    key = (os.path.abspath(path), max_points)
    if key not in _cache:
        if not os.path.isfile(path):
            _cache[key] = None
        else:
            _cache[key] = convex_points(read_vertices(path), max_points)
    return _cache[key]
