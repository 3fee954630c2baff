"""Minimal Gym-compatible space classes.

``gym`` is not a dependency of this backend (it is absent from the build image);
the reference builds its spaces from ``gym.spaces`` (reference:
diy_gym/addons/addon.py:196-210, diy_gym/diy_gym.py:103-122).  These classes
cover exactly what the reference touches: ``Box(low, high, shape, dtype)``,
``Dict(...).spaces``, ``Dict[key]``, ``sample()``, ``contains()``, plus the
discrete kinds that ``utils.get_bounds_for_space`` / ``unflatten`` know about.
If ``gym`` is importable the classes register as virtual subclasses so that
``isinstance(space, gym.spaces.Box)`` style checks in user code keep working.
"""
from collections import OrderedDict

import numpy as np

_rng = np.random.default_rng()


def seed(value=None):
    """Seed the module-level generator used by ``Space.sample()``."""
    global _rng
    _rng = np.random.default_rng(value)


class Space:
    shape = None
    dtype = None

    def sample(self):
        raise NotImplementedError

    def contains(self, x):
        raise NotImplementedError

    def __contains__(self, x):
        return self.contains(x)


class Box(Space):
    def __init__(self, low, high, shape=None, dtype='float32'):
        self.dtype = np.dtype(dtype)
        if shape is None:
            low = np.asarray(low)
            high = np.asarray(high)
            shape = np.broadcast(low, high).shape
        shape = tuple(int(s) for s in shape)
        self.shape = shape
        self.low = np.broadcast_to(np.asarray(low, dtype=self.dtype), shape).copy()
        self.high = np.broadcast_to(np.asarray(high, dtype=self.dtype), shape).copy()

    def sample(self):
        lo = np.where(np.isfinite(self.low), self.low, -1.0)
        hi = np.where(np.isfinite(self.high), self.high, 1.0)
        return (lo + (hi - lo) * _rng.random(self.shape)).astype(self.dtype)

    def contains(self, x):
        x = np.asarray(x)
        return x.shape == self.shape and bool(np.all(x >= self.low) and np.all(x <= self.high))

    def __repr__(self):
        return 'Box(%s, %s, %s, %s)' % (self.low.min() if self.low.size else 0,
                                        self.high.max() if self.high.size else 0, self.shape, self.dtype)

    def __eq__(self, other):
        return (isinstance(other, Box) and self.shape == other.shape and np.array_equal(self.low, other.low)
                and np.array_equal(self.high, other.high))


class Discrete(Space):
    def __init__(self, n):
        self.n = int(n)
        self.shape = ()
        self.dtype = np.dtype('int64')

    def sample(self):
        return int(_rng.integers(self.n))

    def contains(self, x):
        return 0 <= int(x) < self.n


class MultiDiscrete(Space):
    def __init__(self, nvec):
        self.nvec = np.asarray(nvec, dtype=np.int64)
        self.shape = self.nvec.shape
        self.dtype = np.dtype('int64')

    def sample(self):
        return (_rng.random(self.nvec.shape) * self.nvec).astype(self.dtype)

    def contains(self, x):
        x = np.asarray(x)
        return x.shape == self.shape and bool(np.all(x >= 0) and np.all(x < self.nvec))


class MultiBinary(Space):
    def __init__(self, n):
        self.n = int(n)
        self.shape = (self.n, )
        self.dtype = np.dtype('int8')

    def sample(self):
        return _rng.integers(0, 2, size=self.n).astype(self.dtype)

    def contains(self, x):
        x = np.asarray(x)
        return x.shape == self.shape and bool(np.all((x == 0) | (x == 1)))


class Tuple(Space):
    def __init__(self, spaces):
        self.spaces = tuple(spaces)

    def sample(self):
        return tuple(s.sample() for s in self.spaces)

    def contains(self, x):
        return len(x) == len(self.spaces) and all(s.contains(v) for s, v in zip(self.spaces, x))


class Dict(Space):
    """Ordered mapping of sub-spaces.  Like old gym, a plain ``dict`` argument is
    key-sorted; an ``OrderedDict`` keeps its order."""
    def __init__(self, spaces=None, **kwargs):
        if spaces is None:
            spaces = kwargs
        if isinstance(spaces, dict) and not isinstance(spaces, OrderedDict):
            spaces = OrderedDict(sorted(spaces.items(), key=lambda kv: kv[0]))
        self.spaces = OrderedDict(spaces)

    def sample(self):
        return OrderedDict((k, s.sample()) for k, s in self.spaces.items())

    def contains(self, x):
        return (isinstance(x, dict) and set(x.keys()) == set(self.spaces.keys())
                and all(self.spaces[k].contains(v) for k, v in x.items()))

    def __getitem__(self, key):
        return self.spaces[key]

    def __setitem__(self, key, value):
        self.spaces[key] = value

    def __iter__(self):
        return iter(self.spaces)

    def __len__(self):
        return len(self.spaces)

    def keys(self):
        return self.spaces.keys()

    def items(self):
        return self.spaces.items()

    def values(self):
        return self.spaces.values()

    def __repr__(self):
        return 'Dict(' + ', '.join('%s: %r' % kv for kv in self.spaces.items()) + ')'


try:  # pragma: no cover - gym is absent in the build image
    import gym.spaces as _gs
    for _mine, _theirs in ((Box, 'Box'), (Dict, 'Dict'), (Discrete, 'Discrete'), (MultiDiscrete, 'MultiDiscrete'),
                           (MultiBinary, 'MultiBinary'), (Tuple, 'Tuple')):
        try:
            getattr(_gs, _theirs).register(_mine)
        except Exception:
            pass
except Exception:
    pass
