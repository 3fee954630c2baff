"""flatten / unflatten / walk_dict / bounds (reference diy_gym/utils.py:6-95,
diy_gym/tests/test_utils.py:13-20) and the gym-free space classes."""
from collections import OrderedDict

import numpy as np
import torch

from diy_gym_amd import spaces
from diy_gym_amd.utils import flatten, get_bounds_for_space, get_desc_for_space, unflatten, walk_dict


def space():
    return spaces.Dict(OrderedDict(
        blue_marble=spaces.Dict(OrderedDict(force=spaces.Box(-10.0, 10.0, shape=(3, ), dtype='float32'))),
        red_marble=spaces.Dict(OrderedDict(force=spaces.Box(-10.0, 10.0, shape=(3, ), dtype='float32'))),
        arm=spaces.Dict(OrderedDict(ctl=spaces.Dict(OrderedDict(linear=spaces.Box(-0.01, 0.01, shape=(3, )),
                                                                rotation=spaces.Box(-0.01, 0.01, shape=(3, ))))))))


def test_flatten_unflatten_round_trip():
    sp = space()
    action = sp.sample()
    flat = flatten(action)
    assert flat.shape == (12, )
    back = unflatten(flat, sp)
    assert np.all(action['red_marble']['force'] == back['red_marble']['force'])
    assert np.all(action['blue_marble']['force'] == back['blue_marble']['force'])
    assert np.all(action['arm']['ctl']['rotation'] == back['arm']['ctl']['rotation'])


def test_flatten_unflatten_batched_tensors():
    sp = space()
    B = 7
    # leaves are popped in the space's own order (reference utils.py:66-69), so build the tree in that order
    tree = OrderedDict(blue_marble=OrderedDict(force=torch.rand(B, 3)), red_marble=OrderedDict(force=torch.rand(B, 3)),
                       arm=OrderedDict(ctl=OrderedDict(linear=torch.rand(B, 3), rotation=torch.rand(B, 3))))
    flat = flatten(tree, batch_dims=1)
    assert flat.shape == (B, 12)
    back = unflatten(flat, sp, batch_dims=1)
    for k in ('blue_marble', 'red_marble'):
        assert torch.equal(back[k]['force'], tree[k]['force'])
    assert torch.equal(back['arm']['ctl']['linear'], tree['arm']['ctl']['linear'])


def test_bounds_and_desc():
    sp = space()
    lo = flatten(get_bounds_for_space(sp, True))
    hi = flatten(get_bounds_for_space(sp, False))
    assert lo.shape == hi.shape == (12, ) and np.all(lo < hi)
    assert get_desc_for_space(sp) == ['/blue_marble/force', '/red_marble/force', '/arm/ctl/linear', '/arm/ctl/rotation']


def test_walk_dict_matches_reference_quirk():
    # nested levels always SUM (reference utils.py:42-43): all() only applies across receptors
    d = OrderedDict(a=OrderedDict(x=True, y=False), b=OrderedDict(z=False))
    assert walk_dict(d, any) is True
    assert walk_dict(d, all) is False
    d2 = OrderedDict(a=OrderedDict(x=True, y=False), b=OrderedDict(z=True))
    assert walk_dict(d2, all) is True
    r = OrderedDict(a=OrderedDict(x=-1.0, y=-2.5), b=OrderedDict(z=0.5))
    assert walk_dict(r, sum) == -3.0


def test_walk_dict_batched():
    t = OrderedDict(a=OrderedDict(x=torch.tensor([True, False]), y=torch.tensor([False, False])), b=OrderedDict(z=torch.tensor([True, True])))
    assert walk_dict(t, any).tolist() == [True, True]
    assert walk_dict(t, all).tolist() == [True, False]
    r = OrderedDict(a=OrderedDict(x=torch.tensor([1.0, 2.0])), b=OrderedDict(z=torch.tensor([0.5, 0.5])))
    assert walk_dict(r, sum).tolist() == [1.5, 2.5]


def test_spaces_basics():
    b = spaces.Box(-0.5, 0.5, shape=(4, ), dtype='float32')
    s = b.sample()
    assert s.shape == (4, ) and s.dtype == np.float32 and b.contains(s)
    assert not b.contains(np.ones(4))
    d = spaces.Dict({'z': b, 'a': spaces.Discrete(3)})
    assert list(d.spaces.keys()) == ['a', 'z']  # plain dicts are key-sorted like old gym
    assert 'a' in d.spaces and d['z'] is b
    assert d.contains(d.sample())
