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
    'dg_world_kernel_name': (ctypes.c_char_p, [_vp]),
    'dg_world_get_motor_cfg': (ctypes.c_int32, [_vp, _c_f64p]),
    'dg_world_set_motor_cfg': (ctypes.c_int32, [_vp, _c_f64p]),
    'dg_world_init_state': (ctypes.c_int32, [_vp, _vp, _vp]),
    'dg_world_reset': (ctypes.c_int32, [_vp, _vp, _vp, _vp, _vp]),
    'dg_world_step': (ctypes.c_int32, [_vp, _vp, _vp, ctypes.c_uint64, _vp, _vp, _vp, _vp, _vp, _vp]),
    'dg_world_observe': (ctypes.c_int32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    'dg_world_frame_state': (ctypes.c_int32, [_vp, _vp, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, _vp, _vp]),
    'dg_world_apply_wrench': (ctypes.c_int32, [_vp, _vp, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, _vp, _vp, _vp, _vp]),
    'dg_world_render': (ctypes.c_int32, [_vp, _vp, ctypes.c_int32, _vp, _vp, _vp, _vp]),
    'dg_world_set_render_diag': (ctypes.c_int32, [_vp, ctypes.c_int32]),
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


def _scene_constants():
    from .scene import K
    return K


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
        # always indexed: tensors allocated on 'cuda' report 'cuda:<n>', and device('cuda') != device('cuda:0')
        self.device = torch.device('cuda', dev_index)
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

    # -- argument checks ------------------------------------------------------
    # The kernels index raw device pointers: a CPU tensor, a strided view, a wrong width or a short mask would be an
    # out-of-bounds device access, not a Python error -- so every caller-supplied tensor is checked here.
    def _require(self, name, t, shape, dtype):
        if not isinstance(t, torch.Tensor):
            raise ValueError('%s must be a torch.Tensor, got %s' % (name, type(t).__name__))
        if t.device != self.device:
            raise ValueError('%s is on %s, this backend runs on %s' % (name, t.device, self.device))
        if t.dtype != dtype:
            raise ValueError('%s must be %s, got %s' % (name, dtype, t.dtype))
        if tuple(t.shape) != tuple(shape):
            raise ValueError('%s must have shape %s, got %s' % (name, tuple(shape), tuple(t.shape)))
        if not t.is_contiguous():
            raise ValueError('%s must be contiguous' % name)
        return t

    def _require_numel(self, name, t, numel, dtype):
        if not isinstance(t, torch.Tensor):
            raise ValueError('%s must be a torch.Tensor, got %s' % (name, type(t).__name__))
        if t.device != self.device or t.dtype != dtype or not t.is_contiguous() or t.numel() != numel:
            raise ValueError('%s must be a contiguous %s tensor of %d elements on %s, got %s %s on %s' %
                             (name, dtype, numel, self.device, t.dtype, tuple(t.shape), t.device))
        return t

    # -- step path --------------------------------------------------------
    def reset(self, mask=None):
        """``mask``: None (all envs) or one flag per env; bool / uint8 masks on the backend's device are used as they
        are (a bool tensor is reinterpreted, not copied), anything else is converted."""
        if mask is not None:
            if not isinstance(mask, torch.Tensor):
                mask = torch.as_tensor(mask)
            if mask.numel() != self.num_envs:
                raise ValueError('reset mask must have one element per env (%d), got shape %s' % (self.num_envs, tuple(mask.shape)))
            if mask.dtype == torch.bool and mask.device == self.device and mask.is_contiguous():
                mask = mask.view(torch.uint8)
            elif mask.dtype != torch.uint8 or mask.device != self.device or not mask.is_contiguous():
                mask = (mask != 0).to(device=self.device, dtype=torch.uint8).contiguous()
        self._check(self.lib.dg_world_reset(self.handle, _ptr(self.state), _ptr(mask), _ptr(self.obs), self._stream()))

    def step(self, update_mask, actions=None):
        act = self.act if actions is None else self._require('actions', actions, (self.num_envs, max(self.act_dim, 1)), torch.float32)
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

    # -- batched p.applyExternalForce / p.applyExternalTorque for addons written in Python ----------------------------
    LINK_FRAME, WORLD_FRAME = 1, 2   # pybullet's flag values

    def _rows3(self, name, v):
        """``v`` as a contiguous float32 ``[num_envs, 3]`` tensor on this device: one 3-vector for every env, or one per env."""
        if v is None:
            return None
        t = torch.as_tensor(v, dtype=torch.float32, device=self.device) if not isinstance(v, torch.Tensor) else v.to(device=self.device, dtype=torch.float32)
        if t.numel() == 3:
            t = t.reshape(1, 3).expand(self.num_envs, 3)
        if t.numel() != 3 * self.num_envs:
            raise ValueError('%s must have 3 or %d x 3 elements, got shape %s' % (name, self.num_envs, tuple(t.shape)))
        return t.reshape(self.num_envs, 3).contiguous()

    def apply_external_force(self, body, frame, force, pos=None, flags=2):
        """``p.applyExternalForce(uid, linkIndex, forceObj, posObj, flags)`` for every env at once (``force``: ``[B, 3]`` or
        one 3-vector; ``pos``: likewise, default the origin).  ``body`` is a Model's ``uid`` (an attached child model's alias
        uid included), ``frame`` the index ``Model.get_frame_id`` returns (-1: the base).  ``LINK_FRAME`` means the link's
        INERTIAL frame, as in pybullet.  The force acts during the next ``step`` only; see ``dg_world_apply_wrench``."""
        body, frame = self.layout.resolve_frame(body, frame)   # (the uid of a merged child model is an alias into its parent's body)
        self._check(self.lib.dg_world_apply_wrench(self.handle, _ptr(self.state), int(body), int(frame), int(flags), _ptr(self._rows3('force', force)),
                                                   _ptr(self._rows3('pos', pos)), None, self._stream()))

    def apply_external_wrench(self, body, frame, force, pos, torque, flags=2):
        """Force at ``pos`` and torque in ONE launch (what the compiled ``propellor`` op does: its base torque is
        ``r x F + T`` summed before it is added to the state, so this form reproduces that op bit for bit)."""
        body, frame = self.layout.resolve_frame(body, frame)
        self._check(self.lib.dg_world_apply_wrench(self.handle, _ptr(self.state), int(body), int(frame), int(flags), _ptr(self._rows3('force', force)),
                                                   _ptr(self._rows3('pos', pos)), _ptr(self._rows3('torque', torque)), self._stream()))

    def apply_external_torque(self, body, frame, torque, flags=2):
        """``p.applyExternalTorque(uid, linkIndex, torqueObj, flags)`` for every env at once."""
        body, frame = self.layout.resolve_frame(body, frame)
        self._check(self.lib.dg_world_apply_wrench(self.handle, _ptr(self.state), int(body), int(frame), int(flags), None, None,
                                                   _ptr(self._rows3('torque', torque)), self._stream()))

    def camera_resolution(self, camera):
        I, K = self.layout.I, _scene_constants()
        if not 0 <= int(camera) < int(I[K.H_N_CAMERAS]):
            raise ValueError('camera %d out of range (scene has %d)' % (camera, int(I[K.H_N_CAMERAS])))
        ci = I[I[K.H_OFF_CAMERA_I] + int(camera) * K.CI_STRIDE:]
        return int(ci[K.CI_WIDTH]), int(ci[K.CI_HEIGHT])

    def render(self, camera, rgb=None, depth=None, seg=None):
        w, h = self.camera_resolution(camera)
        px = self.num_envs * w * h
        if rgb is not None:
            self._require_numel('rgb', rgb, 3 * px, torch.float32)
        if depth is not None:
            self._require_numel('depth', depth, px, torch.float32)
        if seg is not None:
            self._require_numel('seg', seg, px, torch.int32)
        self._check(self.lib.dg_world_render(self.handle, _ptr(self.state), int(camera), _ptr(rgb), _ptr(depth), _ptr(seg), self._stream()))

    def set_render_diag(self, flags):
        """Diagnostic switches of ``render`` (1: no culling -- the brute-force picture; see dg_world_set_render_diag)."""
        self._check(self.lib.dg_world_set_render_diag(self.handle, int(flags)))

    def motor_cfg(self):
        cfg = np.zeros((self.n_links, 3), dtype=np.float64)
        self._check(self.lib.dg_world_get_motor_cfg(self.handle, cfg.ctypes.data_as(_c_f64p)))
        return cfg

    def set_motor_cfg(self, cfg):
        cfg = np.ascontiguousarray(cfg, dtype=np.float64)
        self._check(self.lib.dg_world_set_motor_cfg(self.handle, cfg.ctypes.data_as(_c_f64p)))

    # columns of the diagnostics buffer (include/diygym_hip.h DG_DIAG_*)
    DIAG_STRIDE, DIAG_CONTACTS, DIAG_PGS_ITERS, DIAG_PGS_ITERS_FIRST, DIAG_CONTACTS_FIRST, DIAG_IK_ITERS, DIAG_N_IK = 8, 0, 1, 2, 3, 4, 4

    def enable_diagnostics(self):
        self.diag = torch.zeros((self.num_envs, self.DIAG_STRIDE), dtype=torch.int32, device=self.device)
        self._check(self.lib.dg_world_set_diag_buffer(self.handle, _ptr(self.diag)))
        return self.diag

    def disable_diagnostics(self):
        self._check(self.lib.dg_world_set_diag_buffer(self.handle, None))

    @property
    def kernel_name(self):
        return self.lib.dg_world_kernel_name(self.handle).decode()

    @property
    def par(self):
        """True when the step runs as the four-wavefront helper-wave kernel."""
        return self.kernel_name.startswith('step_kernel_par')

    SECTIONS = ['update_ops', 'kinematics', 'narrow_phase', 'aba', 'minv', 'rows', 'pgs_other', 'integrate', 'outputs', 'pgs_motor', 'pgs_limit',
                'pgs_contact']

    @property
    def envs_per_wave(self):
        """Envs per wavefront of the workspace mode (``lanes``: 64/32/16/8/4/1 LDS modes, 0 and -16 global-workspace modes).

        The mode is chosen per scene AND batch size (and, for one-env-per-wavefront scenes, the GPU's CU count): each
        mode sums in its own order, so a rollout replays bit for bit only within one mode.  Shards of a job that must
        equal the whole batch bit for bit pin the mode with the environment variable ``DG_MAX_LANES`` (32 / 16 / 8 / 4 /
        1) before constructing their worlds; ``sim.lanes`` tells which mode a world got."""
        return self.lanes if self.lanes > 0 else (-self.lanes if self.lanes < 0 else 64)

    def enable_stamps(self, on=True):
        """Diagnostic: per-wavefront shader cycles per section of the step (see diygym_hip.h)."""
        if on:
            n_waves = (self.num_envs + self.envs_per_wave - 1) // self.envs_per_wave
            self.cycles = torch.zeros((n_waves, len(self.SECTIONS) + 12), dtype=torch.int64, device=self.device)
            self._check(self.lib.dg_world_set_profile_buffer(self.handle, _ptr(self.cycles)))
        else:
            self._check(self.lib.dg_world_set_profile_buffer(self.handle, None))
        return getattr(self, 'cycles', None)

    # state as [num_envs, state_dim] host array (tests / checkpoints)
    def get_state(self):
        return self.state[:, :self.num_envs].t().contiguous().cpu().numpy()

    def set_state(self, arr):
        t = torch.as_tensor(np.asarray(arr, dtype=np.float32), device=self.device)
        if tuple(t.shape) != (self.num_envs, self.state_dim):
            raise ValueError('state must have shape %s, got %s' % ((self.num_envs, self.state_dim), tuple(t.shape)))
        self.state[:, :self.num_envs] = t.t()
