"""The ctypes binding hands raw device pointers to the kernels, so it must reject anything whose device, dtype,
shape or strides are not what the kernels index.  The checks are plain Python and run without a GPU:
a HipBackend is assembled by hand around a library stub that records whether a call got through."""
import numpy as np
import pytest
import torch

from diy_gym_amd.backend import HipBackend


class _Lib:
    def __init__(self):
        self.calls = []

    def __getattr__(self, name):
        def fn(*args):
            self.calls.append(name)
            return 0
        return fn


def make(num_envs=8, act_dim=12):
    b = HipBackend.__new__(HipBackend)
    b.lib, b.handle, b.num_envs, b.act_dim, b.device = _Lib(), 1, num_envs, act_dim, torch.device('cpu')
    b.state_dim, b.stride = 5, 64
    b.state = torch.zeros((5, 64))
    b.act = torch.zeros((num_envs, act_dim))
    b.obs = torch.zeros((num_envs, 3)); b.rew = torch.zeros((num_envs, 1)); b.term = torch.zeros((num_envs, 1), dtype=torch.uint8)
    b.rew_sum = torch.zeros(num_envs); b.term_flag = torch.zeros(num_envs, dtype=torch.uint8)
    b._stream = lambda: None

    class L:  # one 4 x 2 camera
        pass
    from diy_gym_amd.scene import K
    I = np.zeros(K.H_INT_COUNT + K.CI_STRIDE, dtype=np.int32)
    I[K.H_N_CAMERAS] = 1; I[K.H_OFF_CAMERA_I] = K.H_INT_COUNT
    I[K.H_INT_COUNT + K.CI_WIDTH] = 4; I[K.H_INT_COUNT + K.CI_HEIGHT] = 2
    b.layout = L(); b.layout.I = I
    return b


def test_good_arguments_reach_the_library():
    b = make()
    b.step(1, torch.zeros((8, 12)))
    b.reset(torch.ones(8, dtype=torch.bool))
    b.reset(torch.ones(8, dtype=torch.uint8))
    b.reset([1, 0, 0, 0, 0, 0, 0, 1])
    b.render(0, rgb=torch.zeros((8, 4, 2, 3)), depth=torch.zeros((8, 4, 2)), seg=torch.zeros((8, 4, 2), dtype=torch.int32))
    assert b.lib.calls == ['dg_world_step', 'dg_world_reset', 'dg_world_reset', 'dg_world_reset', 'dg_world_render']


@pytest.mark.parametrize('bad', [
    torch.zeros((8, 11)),                       # wrong width
    torch.zeros((7, 12)),                       # wrong batch
    torch.zeros((8, 12), dtype=torch.float64),  # wrong dtype
    torch.zeros((8, 24))[:, ::2],               # right shape, strided
    torch.zeros((12, 8)).t(),                   # right shape, transposed
    np.zeros((8, 12), dtype=np.float32),        # not a tensor
])
def test_step_rejects_bad_actions(bad):
    b = make()
    with pytest.raises(ValueError):
        b.step(1, bad)
    assert b.lib.calls == []


def test_step_rejects_a_tensor_on_another_device():
    b = make()
    b.device = torch.device('meta')  # stands for cuda:0 with a CPU tensor coming in
    with pytest.raises(ValueError, match='is on cpu'):
        b.step(1, torch.zeros((8, 12)))
    assert b.lib.calls == []


def test_reset_rejects_a_short_or_long_mask():
    b = make()
    for n in (7, 9, 0):
        with pytest.raises(ValueError, match='one element per env'):
            b.reset(torch.ones(n, dtype=torch.uint8))
    assert b.lib.calls == []


def test_render_rejects_wrong_buffers_and_cameras():
    b = make()
    with pytest.raises(ValueError):
        b.render(0, rgb=torch.zeros((8, 4, 2)))                      # one channel instead of three
    with pytest.raises(ValueError):
        b.render(0, depth=torch.zeros((8, 4, 2), dtype=torch.float64))
    with pytest.raises(ValueError):
        b.render(0, seg=torch.zeros((8, 4, 2)))                      # float instead of int32
    with pytest.raises(ValueError):
        b.render(0, depth=torch.zeros((4, 4, 2)))                    # half the envs
    with pytest.raises(ValueError, match='out of range'):
        b.render(1, depth=torch.zeros((8, 4, 2)))
    assert b.lib.calls == []


def test_set_state_checks_the_shape():
    b = make()
    with pytest.raises(ValueError):
        b.set_state(np.zeros((8, 4)))
    b.set_state(np.zeros((8, 5)))


def test_more_than_64_terminal_groups_are_rejected_when_the_scene_is_built():
    from diy_gym_amd.scene import K, SceneBuilder
    sb = SceneBuilder()
    for g in range(64):
        sb.add_op(K.OP_TERM_TIMER, 'term', fparams=[10], io_dim=1, group=g)
    with pytest.raises(ValueError, match='more than 64 receptors'):
        sb.add_op(K.OP_TERM_TIMER, 'term', fparams=[10], io_dim=1, group=64)
