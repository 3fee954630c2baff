"""Nested-dict helpers: flatten / unflatten / bounds / reductions.

Same behaviour as the reference helpers (reference: diy_gym/utils.py:6-95) with
one extension: every function takes ``batch_dims`` so that leaves shaped
``[B, ...]`` (torch tensors or numpy arrays) are flattened per environment into
``[B, n]`` instead of into one long vector.  With ``batch_dims=0`` and numpy
leaves the results are the reference's.

Reference quirk kept on purpose: ``walk_dict`` applies ``func`` only at the top
level and always *sums* inside nested dicts (reference utils.py:42-43), so
``terminal_if_all`` means "all receptors have any addon terminal".
"""
from collections import OrderedDict

import numpy as np

from . import spaces

try:
    import torch
except Exception:  # pragma: no cover
    torch = None


def _is_tensor(x):
    return torch is not None and isinstance(x, torch.Tensor)


def get_bounds_for_space(space, low_not_high):
    if isinstance(space, spaces.Discrete):
        return 0 if low_not_high else space.n
    if isinstance(space, spaces.MultiDiscrete):
        return np.zeros(space.nvec.shape) if low_not_high else np.ones(space.nvec.shape) * space.nvec
    if isinstance(space, spaces.MultiBinary):
        return np.zeros(space.n) if low_not_high else np.ones(space.n)
    if isinstance(space, spaces.Box):
        return space.low if low_not_high else space.high
    if isinstance(space, spaces.Dict):
        return OrderedDict(
            sorted(((k, get_bounds_for_space(s, low_not_high)) for k, s in space.spaces.items()),
                   key=lambda kv: kv[0]))
    if isinstance(space, spaces.Tuple):
        return tuple(get_bounds_for_space(s, low_not_high) for s in space.spaces)
    try:
        return space.low if low_not_high else space.high
    except AttributeError:
        raise AttributeError("Could not find a bound for this space; custom spaces must define .low and .high "
                             "so that they can be flattened")


def get_desc_for_space(space, prepend=''):
    names = []
    for key, sub in space.spaces.items():
        if isinstance(sub, spaces.Dict):
            names.extend(get_desc_for_space(sub, prepend + '/' + key))
        else:
            names.append(prepend + '/' + key)
    return names


def walk_dict(d, func=sum):
    """Collapse a nested dict of scalars (or ``[B]`` tensors).

    ``func`` is one of the builtins ``sum`` / ``any`` / ``all``; nested levels
    always use ``sum`` exactly like the reference.
    """
    vals = [walk_dict(e) if isinstance(e, dict) else e for e in d.values()]
    if vals and any(_is_tensor(v) for v in vals):
        stacked = torch.stack([v if _is_tensor(v) else torch.as_tensor(v) for v in vals], dim=0)
        if func is sum:
            return stacked.sum(dim=0) if stacked.dtype != torch.bool else stacked.to(torch.int32).sum(dim=0)
        if func is any:
            return stacked.to(torch.bool).any(dim=0)
        if func is all:
            return stacked.to(torch.bool).all(dim=0)
        return func(stacked)
    return func(vals)


def _leaves(tree):
    if isinstance(tree, dict):
        for v in tree.values():
            yield from _leaves(v)
    elif isinstance(tree, tuple):
        for v in tree:
            yield from _leaves(v)
    else:
        yield tree


def flatten(tree, batch_dims=0):
    """Depth-first concatenation of every leaf along the feature axis."""
    leaves = list(_leaves(tree))
    if any(_is_tensor(v) for v in leaves):
        ref = next(v for v in leaves if _is_tensor(v))
        parts = []
        for v in leaves:
            t = v if _is_tensor(v) else torch.as_tensor(np.asarray(v), device=ref.device)
            lead = t.shape[:batch_dims]
            parts.append(t.reshape(*lead, -1).to(ref.dtype if t.dtype != ref.dtype and t.is_floating_point() else t.dtype))
        return torch.cat(parts, dim=-1)
    parts = []
    for v in leaves:
        a = np.asarray(v)
        parts.append(a.reshape(a.shape[:batch_dims] + (-1, )))
    return np.concatenate(parts, axis=-1)


class _Cursor:
    def __init__(self, arr):
        self.arr = arr
        self.i = 0

    def pop(self, n):
        n = int(n)
        out = self.arr[..., self.i:self.i + n]
        self.i += n
        return out


def unflatten(flat, space, batch_dims=0):
    """Inverse of :func:`flatten` driven by the shapes/dtypes in ``space``."""
    lead = tuple(flat.shape[:batch_dims])

    def rec(cur, sp):
        if isinstance(sp, spaces.Dict):
            return OrderedDict(sorted(((k, rec(cur, s)) for k, s in sp.spaces.items()), key=lambda kv: kv[0]))
        if isinstance(sp, spaces.Tuple):
            return tuple(rec(cur, s) for s in sp.spaces)
        if isinstance(sp, spaces.Discrete):
            v = cur.pop(1)
            if _is_tensor(v):
                return v.round().to(torch.int64).reshape(lead)
            return int(round(float(v[0]))) if not lead else np.round(v).astype(np.int64).reshape(lead)
        if isinstance(sp, spaces.MultiDiscrete):
            v = cur.pop(sp.nvec.size)
            return (v.round().to(torch.uint8) if _is_tensor(v) else np.round(v).astype(np.uint8)).reshape(lead + sp.nvec.shape)
        if isinstance(sp, spaces.MultiBinary):
            v = cur.pop(sp.n)
            return v.round().to(torch.uint8) if _is_tensor(v) else np.round(v).astype(np.uint8)
        if isinstance(sp, spaces.Box):
            v = cur.pop(sp.low.size)
            if _is_tensor(v):
                return v.reshape(lead + sp.low.shape)
            return v.astype(sp.low.dtype).reshape(lead + sp.low.shape)
        raise AttributeError("Unrecognised space type in unflatten; only the built-in space kinds are supported")

    # Dict sub-trees are popped in *space* order (the reference iterates space.spaces.items()
    # while building the sorted dict), which equals sorted order for spaces built by DIYGym.
    return rec(_Cursor(flat), space)
