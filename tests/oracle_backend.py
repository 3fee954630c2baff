"""Adapter that gives the CPU oracle the interface of ``HipBackend``.

Lives under tests/ on purpose: the product package never sees the oracle.  It is
used (a) to run host-side logic tests (addon compilation, spaces, dict plumbing,
sharding) without a GPU and (b) as the checker in the GPU parity tests.
"""
import ctypes
import os

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_LIBS = {}
ORACLE_LIB = os.path.join(ROOT, 'oracle', 'libdgsim_oracle.so')   # fp64, serial: the checker of the parity tests
# other flavours of the same source (oracle/Makefile), used by bench.py's cpu_baseline only
FLAVOURS = {'f64': 'libdgsim_oracle.so', 'f64_omp': 'libdgsim_oracle_omp.so', 'f32': 'libdgsim_oracle_f32.so', 'f32_omp': 'libdgsim_oracle_f32_omp.so'}


def lib(path=None):
    path = path or ORACLE_LIB
    if path not in _LIBS:
        L = ctypes.CDLL(path)
        vp, i32, i64, u64 = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64, ctypes.c_uint64
        L.dgo_create.restype = vp
        L.dgo_create.argtypes = [vp, i64, vp, i64, i32, u64, i64]
        L.dgo_destroy.argtypes = [vp]
        L.dgo_last_error.restype = ctypes.c_char_p
        L.dgo_state_dim.restype = i32
        L.dgo_state_dim.argtypes = [vp]
        L.dgo_real_bytes.restype = i32
        L.real = {8: np.float64, 4: np.float32}[L.dgo_real_bytes()]   # every `real*` of oracle/dgsim_oracle.h
        creal = ctypes.c_double if L.real is np.float64 else ctypes.c_float
        L.dgo_state.restype = ctypes.POINTER(creal)
        L.dgo_state.argtypes = [vp]
        L.dgo_motor_cfg.restype = ctypes.POINTER(creal)
        L.dgo_motor_cfg.argtypes = [vp]
        L.dgo_reset.argtypes = [vp, vp, vp]
        L.dgo_step.argtypes = [vp, vp, u64, vp, vp, vp, vp, vp]
        L.dgo_observe.argtypes = [vp, vp, vp, vp, vp, vp]
        L.dgo_frame_state.argtypes = [vp, i32, i32, i32, i32, vp]
        L.dgo_last_contact_count.restype = i32
        L.dgo_last_contact_count.argtypes = [vp, i32]
        L.dgo_last_iterations.restype = i32
        L.dgo_last_iterations.argtypes = [vp, i32]
        L.dgo_last_contact.restype = i32
        L.dgo_last_contact.argtypes = [vp, i32, i32, vp]
        L.dgo_forward_dynamics.argtypes = [vp, i32, i32, vp, vp]
        L.dgo_unit_response.argtypes = [vp, i32, i32, i32, vp]
        L.dgo_ik.argtypes = [vp, i32, i32, vp, vp]
        L.dgo_render.argtypes = [vp, i32, vp, vp, vp]
        L.dgo_apply_wrench.argtypes = [vp, i32, i32, i32, vp, vp, vp]
        _LIBS[path] = L
    return _LIBS[path]


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


class OracleBackend:
    """The oracle (fp64 unless ``lib_path`` names an fp32 flavour) behind the HipBackend interface (CPU tensors, fp32
    views for outputs)."""
    lib_path = None

    def __init__(self, layout, num_envs, device=None, seed=0, env_index_base=0):
        self.L = lib(self.lib_path)
        real = self.real = self.L.real
        self.layout = layout
        self.num_envs = int(num_envs)
        self.device = torch.device('cpu')
        I, F = layout.I, layout.F
        self.handle = self.L.dgo_create(_p(I), I.size, _p(F), F.size, self.num_envs, seed, env_index_base)
        if not self.handle:
            raise RuntimeError('oracle: ' + self.L.dgo_last_error().decode())
        self.state_dim = self.L.dgo_state_dim(self.handle)
        self.act_dim, self.obs_dim, self.rew_dim, self.term_dim = layout.act_dim, layout.obs_dim, layout.rew_dim, layout.term_dim
        self.n_links = layout.n_links
        B = self.num_envs
        self._state = np.ctypeslib.as_array(self.L.dgo_state(self.handle), shape=(B, self.state_dim))
        self._mcfg = np.ctypeslib.as_array(self.L.dgo_motor_cfg(self.handle), shape=(max(self.n_links, 1), 3))
        self.act = torch.zeros((B, max(self.act_dim, 1)), dtype=torch.float32)
        self.obs64 = np.zeros((B, max(self.obs_dim, 1)), dtype=real)   # ("64": the oracle's own precision)
        self.rew64 = np.zeros((B, max(self.rew_dim, 1)), dtype=real)
        self.term8 = np.zeros((B, max(self.term_dim, 1)), dtype=np.uint8)
        self.rsum64 = np.zeros(B, dtype=real)
        self.tflag8 = np.zeros(B, dtype=np.uint8)
        self.obs = torch.zeros((B, max(self.obs_dim, 1)), dtype=torch.float32)
        self.rew = torch.zeros((B, max(self.rew_dim, 1)), dtype=torch.float32)
        self.term = torch.zeros((B, max(self.term_dim, 1)), dtype=torch.uint8)
        self.rew_sum = torch.zeros(B, dtype=torch.float32)
        self.term_flag = torch.zeros(B, dtype=torch.uint8)

    def _publish(self):
        self.obs.copy_(torch.from_numpy(self.obs64).float())
        self.rew.copy_(torch.from_numpy(self.rew64).float())
        self.term.copy_(torch.from_numpy(self.term8))
        self.rew_sum.copy_(torch.from_numpy(self.rsum64).float())
        self.term_flag.copy_(torch.from_numpy(self.tflag8))

    def close(self):
        if self.handle:
            self.L.dgo_destroy(self.handle)
            self.handle = None

    def reset(self, mask=None):
        m = None if mask is None else np.ascontiguousarray(mask.cpu().numpy().astype(np.uint8))
        self.L.dgo_reset(self.handle, _p(m), _p(self.obs64))
        self._publish()

    def step(self, update_mask, actions=None):
        act = (self.act if actions is None else actions).detach().cpu().numpy().astype(self.real)
        act = np.ascontiguousarray(act)
        self.L.dgo_step(self.handle, _p(act) if self.act_dim else None, update_mask, _p(self.obs64), _p(self.rew64), _p(self.term8),
                        _p(self.rsum64), _p(self.tflag8))
        self._publish()

    def observe(self):
        self.L.dgo_observe(self.handle, _p(self.obs64), _p(self.rew64), _p(self.term8), _p(self.rsum64), _p(self.tflag8))
        self._publish()

    def frame_state(self, body, frame=-1, com=False):
        out = np.zeros((self.num_envs, 13), dtype=self.real)
        gf = -1 if frame < 0 else self.layout.I[0:0].size + self._global_frame(body, frame)
        for e in range(self.num_envs):
            self.L.dgo_frame_state(self.handle, e, body, gf, int(bool(com)), _p(out[e]))
        return torch.from_numpy(out).float()

    def frame_state64(self, body, frame=-1, com=False):
        out = np.zeros((self.num_envs, 13), dtype=self.real)
        gf = -1 if frame < 0 else self._global_frame(body, frame)
        for e in range(self.num_envs):
            self.L.dgo_frame_state(self.handle, e, body, gf, int(bool(com)), _p(out[e]))
        return out

    def _global_frame(self, body, frame):
        from diy_gym_amd.scene import K
        I = self.layout.I
        FI = I[I[K.H_OFF_FRAME_I]:I[K.H_OFF_FRAME_I] + I[K.H_N_FRAMES] * K.FI_STRIDE].reshape(-1, K.FI_STRIDE)
        idx = np.nonzero(FI[:, K.FI_BODY] == body)[0]
        return int(idx[frame])

    def render(self, camera, rgb=None, depth=None, seg=None):
        r64 = np.zeros(tuple(rgb.shape), dtype=self.real) if rgb is not None else None
        d64 = np.zeros(tuple(depth.shape), dtype=self.real) if depth is not None else None
        s32 = np.zeros(tuple(seg.shape), dtype=np.int32) if seg is not None else None
        rc = self.L.dgo_render(self.handle, int(camera), _p(r64), _p(d64), _p(s32))
        if rc:
            raise RuntimeError(self.L.dgo_last_error().decode())
        if rgb is not None:
            rgb.copy_(torch.from_numpy(r64).float())
        if depth is not None:
            depth.copy_(torch.from_numpy(d64).float())
        if seg is not None:
            seg.copy_(torch.from_numpy(s32))
        self.last_render64 = (r64, d64, s32)

    LINK_FRAME, WORLD_FRAME = 1, 2

    def _rows3(self, v):
        if v is None:
            return None
        a = np.asarray(v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else v, dtype=self.real)
        return np.ascontiguousarray(np.broadcast_to(a.reshape(-1, 3), (self.num_envs, 3)))

    def apply_external_force(self, body, frame, force, pos=None, flags=2):
        body, frame = self.layout.resolve_frame(body, frame)
        gf = -1 if frame < 0 else self._global_frame(body, frame)
        f, p = self._rows3(force), self._rows3(pos)
        if self.L.dgo_apply_wrench(self.handle, int(body), gf, int(flags == self.LINK_FRAME), _p(f), _p(p), None):
            raise RuntimeError(self.L.dgo_last_error().decode())

    def apply_external_wrench(self, body, frame, force, pos, torque, flags=2):
        body, frame = self.layout.resolve_frame(body, frame)
        gf = -1 if frame < 0 else self._global_frame(body, frame)
        f, p, t = self._rows3(force), self._rows3(pos), self._rows3(torque)
        if self.L.dgo_apply_wrench(self.handle, int(body), gf, int(flags == self.LINK_FRAME), _p(f), _p(p), _p(t)):
            raise RuntimeError(self.L.dgo_last_error().decode())

    def apply_external_torque(self, body, frame, torque, flags=2):
        body, frame = self.layout.resolve_frame(body, frame)
        gf = -1 if frame < 0 else self._global_frame(body, frame)
        t = self._rows3(torque)
        if self.L.dgo_apply_wrench(self.handle, int(body), gf, int(flags == self.LINK_FRAME), None, None, _p(t)):
            raise RuntimeError(self.L.dgo_last_error().decode())

    def motor_cfg(self):
        return self._mcfg[:self.n_links].copy()

    def set_motor_cfg(self, cfg):
        self._mcfg[:self.n_links] = cfg

    def get_state(self):
        return self._state.copy()

    def set_state(self, arr):
        self._state[:] = arr


    def contacts(self, env=0):
        return self.L.dgo_last_contact_count(self.handle, env)

    def contact(self, env, k):
        """[point 3, normal 3 (B -> A), signed distance, normal impulse] of contact k of the env's most recent substep."""
        out = np.zeros(8, dtype=self.real)
        if not self.L.dgo_last_contact(self.handle, env, k, _p(out)):
            raise IndexError(k)
        return out.astype(np.float64)

    def iterations(self, env=0):
        return self.L.dgo_last_iterations(self.handle, env)

    def forward_dynamics(self, env, body, n):
        qdd = np.zeros(max(n, 1), dtype=self.real)
        a0 = np.zeros(6, dtype=self.real)
        self.L.dgo_forward_dynamics(self.handle, env, body, _p(qdd), _p(a0))
        return qdd[:n], a0

    def unit_response(self, env, body, dof, n):
        dv = np.zeros(6 + n, dtype=self.real)
        self.L.dgo_unit_response(self.handle, env, body, dof, _p(dv))
        return dv

    def ik(self, env, op_index, action, n):
        q = np.zeros(n, dtype=self.real)
        a = np.ascontiguousarray(np.asarray(action, dtype=self.real))
        self.L.dgo_ik(self.handle, env, op_index, _p(a), _p(q))
        return q


def flavour(name):
    """OracleBackend subclass bound to another build of the oracle ('f32', 'f32_omp', 'f64_omp'); None if it was not built."""
    path = os.path.join(ROOT, 'oracle', FLAVOURS[name])
    if not os.path.isfile(path):
        return None
    return type('OracleBackend_' + name, (OracleBackend, ), {'lib_path': path})
