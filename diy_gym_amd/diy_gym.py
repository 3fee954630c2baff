"""``DIYGym``: the Gym-style environment shell over the batched MI355X backend.

Drop-in for the reference class (reference: diy_gym/diy_gym.py:14-225): same
constructor argument (a YAML config file), same config keys, same
``reset / step / observe / reward / is_terminal / seed / close`` surface and the
same ``models / addons / receptors / observation_space / action_space``
attributes.  Differences, all additive:

* ``num_envs=B`` runs B independent copies of the scene on one GPU.  Actions
  are ``[B, ...]`` tensors (or one ``[B, act_dim]`` tensor with
  ``flatten_actions``), observations / rewards / terminals come back as
  ``[B, ...]`` torch tensors that are *views* of three persistent device
  buffers (no per-step allocation, no host sync).
* without ``num_envs`` the environment behaves like the reference: one env,
  numpy in, numpy out (this path synchronises every step and exists for
  compatibility and tests, not for speed).
* ``reset(mask)`` resets a subset of envs (the reference can only reset all).
* config key ``auto_reset: true`` (needs ``terminal_if_any`` / ``terminal_if_all``): ``step()`` resets the envs
  whose collapsed terminal fired before it returns -- their row of the returned observation is the first
  observation of the new episode, reward and terminal are those of the step that ended the old one (the usual
  vector-env convention).  Respawn jitter is drawn per (seed, global env index, episode), so a shard replays.
* ``render`` is accepted and ignored: there is no GUI.

``step()`` never calls back into a per-env engine: controller addons, the
physics step and the sensor / reward / terminal addons all run inside one HIP
kernel launch (``dg_world_step``).
"""
import os
from collections import OrderedDict

import numpy as np
import torch

from . import spaces
from .addons.addon import Addon, AddonFactory, Receptor
from .config import Configuration
from .model import Model
from .scene import K, SceneBuilder
from .utils import flatten, get_bounds_for_space, unflatten, walk_dict


class DIYGym(Receptor):
    metadata = {'render.modes': []}

    def __init__(self, config_file, num_envs=None, device=None, seed=0, env_index_base=0, backend_factory=None,
                 max_hull_points=32, engine=None):
        Receptor.__init__(self)
        config = config_file if isinstance(config_file, Configuration) else Configuration.from_file(config_file)
        self.env = self
        self.config_dir = config.source_dir
        self.name = config.name
        self.compat = num_envs is None
        self.num_envs = 1 if num_envs is None else int(num_envs)
        if not 4 <= int(max_hull_points) <= 256:
            raise ValueError('max_hull_points must be in [4, 256] (a contact is identified by its hull vertex, DG_CONTACT_KEY)')
        self.max_hull_points = max_hull_points
        self._max_episode_steps = config.get('max_episode_steps') if 'max_episode_steps' in config else None
        self.hot_start = config.get('hot_start', 1)

        # physics parameters, exactly the reference's derivation (diy_gym.py:76-82)
        timestep = config.get('timestep', 1 / 240.)
        sub_steps = int(1. / config.get('update_freq', 100) / timestep)
        iterations = config.get('solver_iterations', 150)
        gravity = config.get('gravity', [0.0, 0.0, -9.81])
        self.collapse_rewards_func = sum if config.get('sum_rewards', False) else None
        self.collapse_terminals_func = any if config.get('terminal_if_any', False) else all if config.get(
            'terminal_if_all', False) else None
        self.flatten_observations = config.get('flatten_observations', False)
        self.flatten_actions = config.get('flatten_actions', False)
        self.auto_reset = bool(config.get('auto_reset', False))
        if self.auto_reset and self.collapse_terminals_func is None:
            raise ValueError('auto_reset needs one terminal flag per env: set terminal_if_any or terminal_if_all')
        term_mode = {None: K.COLLAPSE_NONE, any: K.COLLAPSE_ANY, all: K.COLLAPSE_ALL}[self.collapse_terminals_func]
        self.builder = SceneBuilder(timestep=timestep, substeps=sub_steps, solver_iterations=iterations, gravity=gravity,
                                    max_episode_steps=self._max_episode_steps, hot_start=self.hot_start,
                                    rew_mode=K.COLLAPSE_SUM if self.collapse_rewards_func else K.COLLAPSE_NONE,
                                    term_mode=term_mode, max_contacts=config.get('max_contacts', None), **(engine or {}))

        # models in YAML order (body ids follow it), stored sorted by name (diy_gym.py:84-86)
        built = [(child.name, Model(child, env=self)) for child in config.find_all('model')]
        self.models = OrderedDict(sorted(built, key=lambda kv: kv[0]))
        self.addons = OrderedDict(
            sorted(((child.name, AddonFactory.build(child.get('addon'), self, child)) for child in config.find_all('addon')),
                   key=lambda kv: kv[0]))
        self.receptors = OrderedDict(sorted({**self.models, self.name: self}.items(), key=lambda kv: kv[0]))

        # compile the addon program in the order the reference walks addons
        self._hook_addons = []
        for receptor in self.receptors.values():
            for addon in receptor.addons.values():
                if hasattr(addon, 'compile'):
                    addon.compile(self.builder)
                else:
                    self._hook_addons.append(addon)
        self._timer_op = None
        if self._max_episode_steps is not None:
            self._timer_op = self.builder.add_op(K.OP_TERM_TIMER, 'term', fparams=[self._max_episode_steps], io_dim=1,
                                                 group=id(self))
        self.layout = self.builder.finalize()

        if backend_factory is None:
            from .backend import HipBackend
            backend_factory = HipBackend
        self.sim = backend_factory(self.layout, self.num_envs, device=device, seed=seed, env_index_base=env_index_base)
        self.device = self.sim.device
        self._mask = 0
        self._staged = None   # list of (column offset, width, tensors) while step() collects the addons' actions
        self._tick = 0  # bumps whenever the simulation state changes (cameras render lazily per tick)
        self._all_slots = (1 << self.layout.n_slots) - 1 if self.layout.n_slots else 0
        self._has_hook_rewards = any(type(a).reward is not Addon.reward for a in self._hook_addons)
        self._has_hook_terminals = any(type(a).is_terminal is not Addon.is_terminal for a in self._hook_addons)

        # zero-copy flat paths are valid when every addon is compiled and none is hidden
        visible = all(not a.hide for r in self.receptors.values() for a in r.addons.values())
        self._flat_fast = visible and not self._hook_addons
        # camera images live in their own buffers, not in the observation rows: with a camera the flat observation is
        # assembled by flatten() (reference utils.py:46-60), images included, instead of being handed out as a view
        self._flat_obs_fast = self._flat_fast and not self.builder.cameras
        if self.auto_reset and self._has_hook_terminals:
            raise ValueError('auto_reset needs every terminal addon compiled into the step kernel')

        self.seed(seed)
        self.reset()

        self.observation_space, self.action_space = spaces.Dict(OrderedDict()), spaces.Dict(OrderedDict())
        for name, receptor in self.receptors.items():
            obs_space, act_space = receptor.build_spaces()
            if len(obs_space.spaces):
                self.observation_space.spaces[name] = obs_space
            if len(act_space.spaces):
                self.action_space.spaces[name] = act_space
        if self.flatten_observations:
            lows, highs = [flatten(get_bounds_for_space(self.observation_space, opt)) for opt in [True, False]]
            self.original_observation_space = self.observation_space
            self.observation_space = spaces.Box(low=lows, high=highs)
        if self.flatten_actions:
            lows, highs = (flatten(get_bounds_for_space(self.action_space, opt)) for opt in [True, False])
            self.original_action_space = self.action_space
            self.action_space = spaces.Box(low=lows, high=highs)

    # ------------------------------------------------------------ plumbing
    def _out(self, t):
        if not self.compat:
            return t
        a = np.array(t[0].detach().cpu().numpy(), copy=True)  # a snapshot, like the reference's fresh arrays
        return a if a.ndim else a.item()

    def _obs_view(self, off, n):
        return self._out(self.sim.obs[:, off:off + n])

    def _rew_view(self, off):
        return self._out(self.sim.rew[:, off])

    def _term_view(self, off):
        return self._out(self.sim.term[:, off].view(torch.bool))  # 0 / 1 bytes: a reinterpreting view, no copy

    def _as_batch(self, value, width):
        t = torch.as_tensor(np.asarray(value, dtype=np.float32) if not isinstance(value, torch.Tensor) else value,
                            device=self.device).to(torch.float32)
        return t.reshape(1, width).expand(self.num_envs, width) if t.numel() == width else t.reshape(self.num_envs, width)

    def _stage_action(self, addon, action):
        """One controller addon's action for its columns of the action buffer.  Inside ``step()`` the pieces are
        collected and written by ONE concatenation into the buffer when they cover all of it (``_flush_actions``: the
        reference's dict API then costs one small kernel per step instead of two per addon); otherwise, and outside
        ``step()`` (an addon's ``update`` called by hand), each addon's columns are written on the spot."""
        op = addon.op
        if isinstance(action, dict):
            parts = [self._as_batch(action[k], int(np.prod(sp.shape))) for k, sp in addon.action_space.spaces.items()]
        else:
            parts = [self._as_batch(action, op.io_dim)]
        self._mask |= 1 << op.slot
        if self._staged is not None:
            self._staged.append((op.io_off, op.io_dim, parts))
        else:
            self.sim.act[:, op.io_off:op.io_off + op.io_dim] = parts[0] if len(parts) == 1 else torch.cat(parts, dim=1)

    def _flush_actions(self):
        staged, self._staged = self._staged, None
        if not staged:
            return
        staged.sort(key=lambda t: t[0])
        at = 0
        for off, dim, _ in staged:
            if off != at:
                at = -1
                break
            at += dim
        if at == self.layout.act_dim and at > 0:
            torch.cat([p for _, _, parts in staged for p in parts], dim=1, out=self.sim.act)   # every column, in order: one kernel
        else:
            for off, dim, parts in staged:
                self.sim.act[:, off:off + dim] = parts[0] if len(parts) == 1 else torch.cat(parts, dim=1)

    # -------------------------------------------------------------- gym API
    def seed(self, seed=None):
        """Seeds ``action_space.sample()``; the per-env respawn streams are keyed by
        the constructor's ``seed`` (the reference's ``np_random`` is never read:
        diy_gym.py:124-128)."""
        spaces.seed(seed)
        self.np_random = np.random.default_rng(seed)
        return [seed]

    def reset(self, mask=None):
        """Reference diy_gym.py:130-148 (per-env ``mask`` is an extension)."""
        for addon in self._hook_addons:
            addon.reset()
        self.sim.reset(mask)
        self._tick += 1
        return self.observe(_refresh=False)

    def observe(self, _refresh=True):
        if _refresh:
            self.sim.observe()
        if self.flatten_observations and self._flat_obs_fast:
            return self._out(self.sim.obs[:, :self.layout.obs_dim])
        ret = self.walk_addons(lambda addon: addon.observe())
        return flatten(ret, batch_dims=0 if self.compat else 1) if self.flatten_observations else ret

    def reward(self):
        if self.collapse_rewards_func is not None and not self._has_hook_rewards:
            return self._out(self.sim.rew_sum)
        ret = self.walk_addons(lambda addon: addon.reward())
        return walk_dict(ret, self.collapse_rewards_func) if self.collapse_rewards_func is not None else ret

    def is_terminal(self):
        if self.collapse_terminals_func is not None and not self._has_hook_terminals:
            return self._out(self.sim.term_flag.view(torch.bool))
        ret = self.walk_addons(lambda addon: addon.is_terminal())
        if self._timer_op is not None:
            if self.name not in ret:
                ret[self.name] = OrderedDict()
            ret[self.name]['episode_timer'] = self._term_view(self._timer_op.io_off)
        return walk_dict(ret, self.collapse_terminals_func) if self.collapse_terminals_func is not None else ret

    def step(self, action):
        """Reference diy_gym.py:187-209: only the addons named in ``action`` are updated.

        ALIASING CONTRACT (batched use, ``num_envs`` given): the observations, rewards and terminals returned here are
        zero-copy VIEWS of the backend's output buffers (terminals: a ``bool`` reinterpretation of 0 / 1 bytes).  The
        next ``step()`` / ``reset()`` overwrites them in place -- a trainer that keeps them in a rollout buffer must
        ``clone()`` them.  The single-env compatibility mode (no ``num_envs``) returns fresh numpy snapshots, like the
        reference."""
        if self.flatten_actions:
            if self._flat_fast and isinstance(action, torch.Tensor) and not self.compat:
                # the flat action tensor already has the kernel's column order: hand it over as is
                if action.numel() != self.num_envs * self.layout.act_dim:
                    raise ValueError('flat action must have %d x %d elements, got shape %s' % (self.num_envs, self.layout.act_dim, tuple(action.shape)))
                act = action.to(device=self.device, dtype=torch.float32).reshape(self.num_envs, self.layout.act_dim).contiguous() if self.layout.act_dim else None
                self.sim.step(self._all_slots, act)
                self._tick += 1
                return self._finish_step()
            action = unflatten(torch.as_tensor(np.asarray(action, dtype=np.float32)) if not isinstance(action, torch.Tensor)
                               else action, self.original_action_space, batch_dims=0 if self.compat else 1)
        self._mask = 0
        self._staged = []
        try:
            for receptor_name, receptor_action in action.items():
                for addon_name, addon_action in receptor_action.items():
                    self.receptors[receptor_name].addons[addon_name].update(addon_action)
        except BaseException:
            self._staged = None
            raise
        self._flush_actions()
        self.sim.step(self._mask)
        self._tick += 1
        return self._finish_step()

    def _finish_step(self):
        if not self.auto_reset:
            return self.observe(_refresh=False), self.reward(), self.is_terminal(), {}
        # reward / terminal of the finished step first (they are views of buffers the reset does not write), then the
        # masked reset, which rewrites the observation rows of the envs it restarted
        rew, term = self.reward(), self.is_terminal()
        self.sim.reset(self.sim.term_flag)
        self._tick += 1
        return self.observe(_refresh=False), rew, term, {}

    def walk_addons(self, func):
        ret = OrderedDict()
        for receptor_name, receptor in self.receptors.items():
            receptor_ret = OrderedDict()
            for addon_name, addon in receptor.addons.items():
                addon_ret = func(addon)
                if addon_ret is not None:
                    receptor_ret[addon_name] = addon_ret
            if len(receptor_ret):
                ret[receptor_name] = receptor_ret
        return ret

    def close(self):
        self.sim.close()
