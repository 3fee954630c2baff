"""ctypes binding of the C-ABI in ``include/diygym_hip.h`` (``libdiygym_hip.so``).

This is the only compute path of the package.  There is no CPU fallback: if the
HIP library has not been built, or no GPU is visible, constructing a backend
raises.  (The CPU oracle under ``oracle/`` is test infrastructure and is never
imported from here.)
"""
import ctypes
import os

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'csrc', 'libdiygym_hip.so')

_lib = None

_c_i32p = ctypes.POINTER(ctypes.c_int32)
_c_f64p = ctypes.POINTER(ctypes.c_double)
_vp = ctypes.c_void_p

# every symbol include/diygym_hip.h declares: name -> (restype, argtypes)
SYMBOLS = {
    'dg_version': (ctypes.c_int32, []),
    'dg_last_error': (ctypes.c_char_p, []),
    'dg_world_create': (ctypes.c_int32, [_c_i32p, ctypes.c_int64, _c_f64p, ctypes.c_int64, ctypes.c_int32, ctypes.c_int32,
                                         ctypes.c_int32, ctypes.c_uint64, ctypes.c_int64, ctypes.POINTER(_vp)]),
    'dg_world_destroy': (None, [_vp]),
    'dg_world_dims': (ctypes.c_int32, [_vp, _c_i32p]),
    'dg_world_get_motor_cfg': (ctypes.c_int32, [_vp, _c_f64p]),
    'dg_world_set_motor_cfg': (ctypes.c_int32, [_vp, _c_f64p]),
    'dg_world_init_state': (ctypes.c_int32, [_vp, _vp, _vp]),
    'dg_world_reset': (ctypes.c_int32, [_vp, _vp, _vp, _vp, _vp]),
    'dg_world_step': (ctypes.c_int32, [_vp, _vp, _vp, ctypes.c_uint64, _vp, _vp, _vp, _vp, _vp, _vp]),
    'dg_world_observe': (ctypes.c_int32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    'dg_world_frame_state': (ctypes.c_int32, [_vp, _vp, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, _vp, _vp]),
    'dg_world_render': (ctypes.c_int32, [_vp, _vp, ctypes.c_int32, _vp, _vp, _vp, _vp]),
    'dg_world_set_diag_buffer': (ctypes.c_int32, [_vp, _vp]),
    'dg_world_set_profile_buffer': (ctypes.c_int32, [_vp, _vp]),
}


def load_library(path=None):
    """Load ``libdiygym_hip.so`` and type every exported entry point."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    path = path or LIB_PATH
    if not os.path.isfile(path):
        raise RuntimeError('HIP library not built: %s is missing. Run `python -c "import __graft_entry__ as g; g.build()"` '
                           '(or `make -j8 -C diy_gym_amd/csrc`).  There is no CPU fallback.' % path)
    lib = ctypes.CDLL(path)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if the library does not export it
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def _ptr(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


class HipBackend:
    """Owns one ``dg_world`` and the device tensors of one shard of envs."""
    def __init__(self, layout, num_envs, device=None, seed=0, env_index_base=0):
        if not torch.cuda.is_available():
            raise RuntimeError('diy_gym_amd needs a ROCm GPU (torch.cuda.is_available() is False); there is no CPU fallback')
        self.lib = load_library()
        self.layout = layout
        self.num_envs = int(num_envs)
        self.device = torch.device(device if device is not None else 'cuda:0')
        if self.device.type != 'cuda':
            raise RuntimeError('diy_gym_amd runs on ROCm devices only, got %s' % self.device)
        dev_index = self.device.index if self.device.index is not None else torch.cuda.current_device()
        self.stride = ((self.num_envs + 63) // 64) * 64
        I, F = layout.I, layout.F
        handle = _vp()
        rc = self.lib.dg_world_create(I.ctypes.data_as(_c_i32p), I.size, F.ctypes.data_as(_c_f64p), F.size, self.num_envs,
                                      self.stride, dev_index, ctypes.c_uint64(seed), ctypes.c_int64(env_index_base),
                                      ctypes.byref(handle))
        self._check(rc)
        self.handle = handle
        dims = (ctypes.c_int32 * 8)()
        self._check(self.lib.dg_world_dims(self.handle, dims))
        self.state_dim, self.act_dim, self.obs_dim, self.rew_dim, self.term_dim, self.n_links, self.lds_bytes, self.lanes = list(dims)
        B, dev = self.num_envs, self.device
        with torch.cuda.device(dev):
            self.state = torch.zeros((self.state_dim, self.stride), dtype=torch.float32, device=dev)
            self.act = torch.zeros((B, max(self.act_dim, 1)), dtype=torch.float32, device=dev)
            self.obs = torch.zeros((B, max(self.obs_dim, 1)), dtype=torch.float32, device=dev)
            self.rew = torch.zeros((B, max(self.rew_dim, 1)), dtype=torch.float32, device=dev)
            self.term = torch.zeros((B, max(self.term_dim, 1)), dtype=torch.uint8, device=dev)
            self.rew_sum = torch.zeros((B, ), dtype=torch.float32, device=dev)
            self.term_flag = torch.zeros((B, ), dtype=torch.uint8, device=dev)
        self._check(self.lib.dg_world_init_state(self.handle, _ptr(self.state), self._stream()))

    def _stream(self):
        return ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _check(self, rc):
        if rc != 0:
            raise RuntimeError('diygym_hip error %d: %s' % (rc, self.lib.dg_last_error().decode()))

    def close(self):
        if getattr(self, 'handle', None):
            self.lib.dg_world_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- step path --------------------------------------------------------
    def reset(self, mask=None):
        if mask is not None:
            mask = mask.to(device=self.device, dtype=torch.uint8).contiguous()
        self._check(self.lib.dg_world_reset(self.handle, _ptr(self.state), _ptr(mask), _ptr(self.obs), self._stream()))

    def step(self, update_mask, actions=None):
        act = self.act if actions is None else actions
        self._check(
            self.lib.dg_world_step(self.handle, _ptr(self.state), _ptr(act) if self.act_dim else None,
                                   ctypes.c_uint64(update_mask), _ptr(self.obs), _ptr(self.rew), _ptr(self.term),
                                   _ptr(self.rew_sum), _ptr(self.term_flag), self._stream()))

    def observe(self):
        self._check(
            self.lib.dg_world_observe(self.handle, _ptr(self.state), _ptr(self.obs), _ptr(self.rew), _ptr(self.term),
                                      _ptr(self.rew_sum), _ptr(self.term_flag), self._stream()))

    def frame_state(self, body, frame=-1, com=False):
        out = torch.empty((self.num_envs, 13), dtype=torch.float32, device=self.device)
        self._check(self.lib.dg_world_frame_state(self.handle, _ptr(self.state), int(body), int(frame), int(bool(com)), _ptr(out),
                                                  self._stream()))
        return out

    def render(self, camera, rgb=None, depth=None, seg=None):
        self._check(self.lib.dg_world_render(self.handle, _ptr(self.state), int(camera), _ptr(rgb), _ptr(depth), _ptr(seg), self._stream()))

    def motor_cfg(self):
        cfg = np.zeros((self.n_links, 3), dtype=np.float64)
        self._check(self.lib.dg_world_get_motor_cfg(self.handle, cfg.ctypes.data_as(_c_f64p)))
        return cfg

    def set_motor_cfg(self, cfg):
        cfg = np.ascontiguousarray(cfg, dtype=np.float64)
        self._check(self.lib.dg_world_set_motor_cfg(self.handle, cfg.ctypes.data_as(_c_f64p)))

    def enable_diagnostics(self):
        self.diag = torch.zeros((self.num_envs, 2), dtype=torch.int32, device=self.device)
        self._check(self.lib.dg_world_set_diag_buffer(self.handle, _ptr(self.diag)))
        return self.diag

    SECTIONS = ['update_ops', 'kinematics', 'narrow_phase', 'aba', 'minv', 'rows', 'pgs_other', 'integrate', 'outputs', 'pgs_motor', 'pgs_limit',
                'pgs_contact']

    @property
    def envs_per_wave(self):
        """Envs per wavefront of the workspace mode (``lanes``: 64/32/16 LDS modes, 0 and -16 global-workspace modes)."""
        return self.lanes if self.lanes > 0 else (-self.lanes if self.lanes < 0 else 64)

    def enable_stamps(self, on=True):
        """Diagnostic: per-wavefront shader cycles per section of the step (see diygym_hip.h)."""
        if on:
            n_waves = (self.num_envs + self.envs_per_wave - 1) // self.envs_per_wave
            self.cycles = torch.zeros((n_waves, len(self.SECTIONS)), dtype=torch.int64, device=self.device)
            self._check(self.lib.dg_world_set_profile_buffer(self.handle, _ptr(self.cycles)))
        else:
            self._check(self.lib.dg_world_set_profile_buffer(self.handle, None))
        return getattr(self, 'cycles', None)

    # state as [num_envs, state_dim] host array (tests / checkpoints)
    def get_state(self):
        return self.state[:, :self.num_envs].t().contiguous().cpu().numpy()

    def set_state(self, arr):
        t = torch.as_tensor(np.asarray(arr, dtype=np.float32), device=self.device)
        self.state[:, :self.num_envs] = t.t()
