"""Drop-in ``CollectiveCrossingEnv``: the reference's dict / MultiAgentEnv API over libccx.

Same constructor, ``reset`` / ``step`` signatures, agent ids, key-presence rules and attributes as
``collectivecrossing.CollectiveCrossingEnv`` (collectivecrossing.py:30-783), so
``lambda env_config: CollectiveCrossingEnv(config=CollectiveCrossingConfig(**env_config))``
(examples/training_script.py:26-29) keeps working.  One instance = a batch of ONE env on the GPU:
``step`` builds the action / move-order arrays from the dict, launches ``ccx_step`` and decodes the
flag bytes back into the five dicts.  There is no CPU implementation of ``step`` in this class --
without the MI355X and libccx it raises.  For throughput use :class:`BatchedCollectiveCrossing`.

Deliberate differences (documented in DESIGN.md):
  * actions are validated for the WHOLE dict before anything moves (the reference validates
    interleaved with moving, :197-202, so a bad entry leaves earlier agents moved and
    ``_step_count`` incremented); the ``ValueError`` messages keep the reference's wording.
  * the key order of ``observations`` / ``infos`` is slot order (the reference iterates a ``set``
    of strings, :243, so its order changes with PYTHONHASHSEED); values are identical.
"""

from __future__ import annotations

import os

import numpy as np

from . import _abi
from .actions import ACTION_TO_DIRECTION
from .batched import BatchedCollectiveCrossing
from .configs import CollectiveCrossingConfig
from .params import agent_ids, calculate_tram_boundaries
from .reset import make_generator, sample_initial_positions
from .spaces import Discrete
from .strategies import (get_observation_function, get_reward_function, get_terminated_function,
                         get_truncated_function)
from .types import Agent, AgentType

try:  # RLlib is optional; with it the class is a real MultiAgentEnv
    from ray.rllib.env.multi_agent_env import MultiAgentEnv as _Base  # type: ignore
except Exception:
    _Base = object


class _Mirror:
    """Host copy of the one-env SoA state; ``dirty`` = must be uploaded before the next launch.

    ``step()`` does not update the arrays eagerly: it parks the step's outputs with ``defer`` and the
    arrays are brought up to date on first access (the ``env._agents[...]`` views, ``_upload``, the
    helper predicates) -- a plain rollout loop never pays for them.  ``version`` counts external
    writes so the env knows when its own per-step flag lists are out of date.
    """

    FIELDS = ("x", "y", "active", "terminated", "truncated")

    def __init__(self, n: int):
        self._arr = {"x": np.zeros(n, np.int32), "y": np.zeros(n, np.int32), "active": np.ones(n, np.uint8),
                     "terminated": np.zeros(n, np.uint8), "truncated": np.zeros(n, np.uint8)}
        self._pending = None
        self.step_count = 0
        self.dirty = True
        self.version = 0

    def defer(self, obs: np.ndarray, active: list, terminated: list, truncated: list) -> None:
        """Park the latest post-step state: obs columns 0/1 are x/y, the lists are cumulative flags."""
        self._pending = (obs, active, terminated, truncated)

    def _sync(self) -> None:
        obs, active, terminated, truncated = self._pending
        self._pending = None
        a = self._arr
        xy = obs[:, :2].astype(np.int32)
        a["x"][:], a["y"][:] = xy[:, 0], xy[:, 1]
        a["active"][:], a["terminated"][:], a["truncated"][:] = active, terminated, truncated

    def _field(name):  # noqa: N805
        def get(self) -> np.ndarray:
            if self._pending is not None:
                self._sync()
            return self._arr[name]
        return property(get)

    x, y, active = _field("x"), _field("y"), _field("active")
    terminated, truncated = _field("terminated"), _field("truncated")
    del _field

    def touch(self) -> None:
        self.version += 1

    def write(self, name: str, index: int, value: int) -> None:
        getattr(self, name)[index] = value
        self.dirty = True
        self.version += 1


def decode_step(ids, obs, reward, agent_flags, env_flag, agent_types):
    """Flag bytes + arrays of ONE env -> the reference's five dicts (collectivecrossing.py:214-261).

    Pure function (the CPU test-suite exercises it without a GPU):
    ``rewards`` / ``truncateds`` only hold agents that were live before the step (CCX_AF_LIVE),
    ``terminateds`` holds everybody, ``observations`` / ``infos`` hold CCX_AF_OBS agents, and both
    flag dicts get ``"__all__"``.  The observation rows are views of ONE fresh copy of ``obs``.
    """
    observations, rewards, terminateds, truncateds, infos = {}, {}, {}, {}, {}
    flags = agent_flags.tolist() if hasattr(agent_flags, "tolist") else [int(f) for f in agent_flags]
    rew = reward.tolist() if hasattr(reward, "tolist") else [float(r) for r in reward]
    rows = np.array(obs, dtype=np.float32)
    for i, aid in enumerate(ids):
        f = flags[i]
        terminateds[aid] = (f & 1) != 0                     # CCX_AF_TERMINATED
        if f & 4:                                           # CCX_AF_LIVE
            rewards[aid] = rew[i]
            truncateds[aid] = (f & 2) != 0                  # CCX_AF_TRUNCATED
        if f & 8:                                           # CCX_AF_OBS
            observations[aid] = rows[i]
            infos[aid] = {"agent_type": agent_types[i], "in_tram_area": (f & 0x10) != 0,
                          "at_door": (f & 0x20) != 0, "active": (f & 0x40) != 0,
                          "at_destination": (f & 0x80) != 0}
    terminateds["__all__"] = bool(env_flag & _abi.EF_ALL_TERMINATED)
    truncateds["__all__"] = bool(env_flag & _abi.EF_ALL_TRUNCATED)
    return observations, rewards, terminateds, truncateds, infos


def _encode_lists(ids, slot, action_dict):
    n = len(ids)
    actions = [_abi.ACTION_ABSENT] * n
    order = []
    for aid, action in action_dict.items():
        i = slot.get(aid)
        if i is None:
            raise ValueError(f"Unknown agent ID: {aid} in action_dict. The action_dict keys must be a "
                             f"subset of the agents. Current agents: {dict.fromkeys(ids).keys()}")
        if action not in ACTION_TO_DIRECTION:
            raise ValueError(f"Invalid action: {action} for agent {aid}. Valid actions are: "
                             f"{list(ACTION_TO_DIRECTION)}")
        actions[i] = int(action)
        order.append(i)
    if len(order) != n:
        listed = set(order)
        order += [i for i in range(n) if i not in listed]
    return actions, order


def encode_actions(ids, action_dict):
    """``action_dict`` -> (actions u8 [N] with 255 = absent, move order u8 [N]); raises the
    reference's ``ValueError``s (collectivecrossing.py:685-711), agent check before action check."""
    actions, order = _encode_lists(ids, {aid: i for i, aid in enumerate(ids)}, action_dict)
    return np.asarray(actions, np.uint8), np.asarray(order, np.uint8)


class CollectiveCrossingEnv(_Base):
    """Multi-agent tram boarding / exiting grid world, one env instance on the GPU."""

    metadata = {"render_modes": ["human", "rgb_array"], "render_fps": 4}

    def __init__(self, config: CollectiveCrossingConfig, device: int | str | None = None, *,
                 _host_view: bool = False):
        self._config = config
        self._tram_boundaries = calculate_tram_boundaries(config)
        self._action_to_direction = ACTION_TO_DIRECTION
        # unknown strategy names raise ValueError here, like the reference (:69-78)
        self._observation_function = get_observation_function(config.observation_config)
        self._reward_function = get_reward_function(config.reward_config)
        self._terminated_function = get_terminated_function(config.terminated_config)
        self._truncated_function = get_truncated_function(config.truncated_config)
        self._ids = agent_ids(config)
        nb = config.num_boarding_agents
        self._types = [AgentType.BOARDING if i < nb else AgentType.EXITING for i in range(len(self._ids))]
        self._type_names = [t.value for t in self._types]
        self._slot = {aid: i for i, aid in enumerate(self._ids)}
        self._identity_order = list(range(len(self._ids)))
        self._mirror = _Mirror(len(self._ids))
        self._flags_version = -1     # version of the mirror the per-step flag lists were taken from
        self._agents: dict[str, Agent] = {aid: Agent(self._mirror, i, aid, self._types[i])
                                          for i, aid in enumerate(self._ids)}
        self._agents_truncated_or_terminated_this_step: set[str] = set()
        self._setup_spaces()
        if _Base is not object:
            super().__init__()
        # Strategies without a kernel_mode are user-registered Python classes (rewards.py:186-216,
        # terminateds.py:86-114, truncateds.py:99-128, observations.py:122-149 accept any registered
        # class).  The GPU then runs the move / deactivate phases with built-in stand-ins for them and
        # step() evaluates the user's strategies on the host mirror in the reference's phase order
        # (_step_host_strategies: the documented slow path).
        self._host_strategies = any(fn.kernel_mode is None for fn in (
            self._reward_function, self._terminated_function, self._truncated_function,
            self._observation_function))
        gpu_config = config
        if self._host_strategies:
            from .configs import (DefaultObservationConfig, DefaultRewardConfig,
                                  IndividualAtDestinationTerminatedConfig, MaxStepsTruncatedConfig)
            update = {}
            if self._reward_function.kernel_mode is None:
                update["reward_config"] = DefaultRewardConfig()
            if self._terminated_function.kernel_mode is None:
                update["terminated_config"] = IndividualAtDestinationTerminatedConfig()
            if self._truncated_function.kernel_mode is None:
                update["truncated_config"] = MaxStepsTruncatedConfig()
            if self._observation_function.kernel_mode is None:
                update["observation_config"] = DefaultObservationConfig()
            gpu_config = config.model_copy(update=update)
        # _host_view: no GPU handle -- only the host mirror and the predicates / strategy objects on
        # it (inspect recorded states, drive host-side policies); reset() and step() need the GPU
        self._batch = None if _host_view else BatchedCollectiveCrossing(gpu_config, 1, device=device)
        self.np_random: np.random.Generator | None = None
        self._window = None
        self._clock = None

    # ------------------------------------------------------------------ reference properties
    config = property(lambda self: self._config)
    tram_boundaries = property(lambda self: self._tram_boundaries)
    tram_door_left = property(lambda self: self._tram_boundaries.tram_door_left)
    tram_door_right = property(lambda self: self._tram_boundaries.tram_door_right)
    tram_left = property(lambda self: self._tram_boundaries.tram_left)
    tram_right = property(lambda self: self._tram_boundaries.tram_right)
    action_spaces = property(lambda self: self._action_spaces)
    observation_spaces = property(lambda self: self._observation_spaces)

    @property
    def _step_count(self) -> int:
        return self._mirror.step_count

    @_step_count.setter
    def _step_count(self, value: int) -> None:
        self._mirror.step_count = int(value)
        self._mirror.dirty = True

    @property
    def agents(self) -> list[str]:
        """Ids that are neither terminated nor truncated (collectivecrossing.py:743-768)."""
        t, u = self._flag_lists()[1:]
        return [aid for i, aid in enumerate(self._ids) if not t[i] and not u[i]]

    def _flag_lists(self):
        """(active, terminated, truncated) as python lists, cumulative, refreshed after external
        writes to the mirror (``version``)."""
        m = self._mirror
        if self._flags_version != m.version:
            self._flags = (m.active.tolist(), m.terminated.tolist(), m.truncated.tolist())
            self._flags_version = m.version
        return self._flags

    @property
    def possible_agents(self) -> list[str]:
        return list(self._ids)

    def get_observation_space(self, agent_id):
        return self.observation_space

    def get_action_space(self, agent_id):
        return self.action_space

    def _setup_spaces(self) -> None:
        self._action_spaces = {aid: Discrete(5) for aid in self._ids}
        self._observation_spaces = {
            aid: self._observation_function.return_agent_observation_space(aid, self) for aid in self._ids}
        if self._ids:
            self.action_space = self._action_spaces[self._ids[0]]
            self.observation_space = self._observation_spaces[self._ids[0]]

    # ------------------------------------------------------------------ state sync
    @classmethod
    def host_view(cls, config: CollectiveCrossingConfig) -> "CollectiveCrossingEnv":
        """An env object WITHOUT a GPU handle: the agent views, geometry predicates and strategy
        objects work on the host mirror (set states through ``env._agents[...]``); ``reset`` /
        ``step`` raise.  Used to evaluate host-side policies on recorded states."""
        return cls(config, _host_view=True)

    def _need_gpu(self) -> None:
        if self._batch is None:
            raise RuntimeError("this CollectiveCrossingEnv is a host view (no GPU handle): reset() and "
                               "step() are unavailable")

    def _upload(self) -> None:
        self._need_gpu()
        m = self._mirror
        if m.dirty:
            self._batch.set_state(x=m.x, y=m.y, active=m.active, terminated=m.terminated,
                                  truncated=m.truncated, step_count=[m.step_count])
            m.dirty = False

    def _download(self) -> None:
        st = self._batch.get_state()
        m = self._mirror
        for k in _Mirror.FIELDS:
            getattr(m, k)[:] = st[k][0]
        m.step_count = int(st["step_count"][0])
        m.dirty = False
        m.touch()

    # ------------------------------------------------------------------ reset / step
    def reset(self, *, seed: int | None = None, options: dict | None = None):
        """Seeded rejection-sampled placement (collectivecrossing.py:91-159), bit-identical."""
        self._need_gpu()
        if seed is not None or self.np_random is None:
            self.np_random = make_generator(seed)        # gymnasium.Env.reset(seed=...)
        pos = sample_initial_positions(self._config, self.np_random)
        m = self._mirror
        m._pending = None            # everything is overwritten below
        m.x[:], m.y[:] = pos[:, 0], pos[:, 1]
        m.active[:] = 1
        m.terminated[:] = 0
        m.truncated[:] = 0
        m.step_count = 0
        m.dirty = True
        m.touch()
        self._upload()
        obs = self._batch.observe().cpu().numpy()[0]
        observations = {aid: np.array(obs[i]) for i, aid in enumerate(self._ids)}
        infos = {aid: {"agent_type": self._type_names[i]} for i, aid in enumerate(self._ids)}
        return observations, infos

    def _alloc_io(self) -> None:
        """One device buffer + one pinned host buffer for all outputs of a step (a single D2H copy
        per step), one of each for the inputs."""
        import ctypes as C

        import torch

        n, L = len(self._ids), 6 + 4 * len(self._ids)
        a16 = lambda v: (v + 15) & ~15  # noqa: E731
        o_obs, o_rew = 0, a16(n * L * 4)
        o_af = o_rew + a16(n * 8)
        o_ef = o_af + a16(n)
        total = o_ef + 16
        dev = self._batch.device
        self._out_host = torch.empty(total, dtype=torch.uint8).pin_memory()
        self._in_host = torch.empty(2 * n, dtype=torch.uint8).pin_memory()
        # zero-copy (default): pinned host memory is mapped into the GPU's address space at the same
        # address, so the step kernel reads the actions from it and writes its ~1.5 KB of outputs
        # straight into it over the host link -- a step is ONE launch + one stream sync, no memcpy
        # calls.  CCX_ENV_STAGED=1 keeps device staging buffers and two async copies instead.
        self._zero_copy = os.environ.get("CCX_ENV_STAGED", "0") != "1"
        if self._zero_copy:
            b, ptrs = self._batch, []
            for t in (self._out_host, self._in_host):
                d = C.c_void_p()
                rc = b._lib.ccx_host_device_pointer(b._h, C.c_void_p(t.data_ptr()), C.byref(d))
                ptrs.append(d.value if rc == 0 else None)
            self._zero_copy = None not in ptrs
        if self._zero_copy:
            base, self._in_base = ptrs
        else:
            self._out_dev = torch.empty(total, dtype=torch.uint8, device=dev)
            self._in_dev = torch.empty(2 * n, dtype=torch.uint8, device=dev)
            base, self._in_base = self._out_dev.data_ptr(), self._in_dev.data_ptr()
        self._step_out = _abi.CcxStepOut(base + o_obs, base + o_rew, base + o_af, base + o_ef)
        h = self._out_host.numpy()
        self._h_obs = h[o_obs:o_obs + n * L * 4].view(np.float32).reshape(n, L)
        self._h_rew = h[o_rew:o_rew + n * 8].view(np.float64)
        self._h_af = h[o_af:o_af + n]
        self._h_ef = h[o_ef:o_ef + 1]
        self._h_in = self._in_host.numpy()
        self._step_out_ref = C.byref(self._step_out)

    def step(self, action_dict):
        """One tick (collectivecrossing.py:161-261) on the GPU: one ``ccx_step`` launch that reads
        the actions / move order from and writes obs + rewards + flag bytes to pinned host memory
        (zero-copy; two staging copies instead with CCX_ENV_STAGED=1), one stream sync."""
        import ctypes as C

        ids = self._ids
        n = len(ids)
        actions, order = _encode_lists(ids, self._slot, action_dict)
        self._upload()
        if not hasattr(self, "_in_base"):
            self._alloc_io()
        self._h_in[:] = actions + order
        if not self._zero_copy:
            self._in_dev.copy_(self._in_host, non_blocking=True)
        b = self._batch
        base = self._in_base
        # dict order = slot order (the usual case: `{a: ... for a in env.agents}`): no move-order array, the
        # kernel then runs its plain path without the rank exchange (same results, a shorter step)
        rc = b._lib.ccx_step(b._h, base, None if order == self._identity_order else base + n, self._step_out_ref)
        if rc:
            from ._lib import check
            check(rc)
        if not self._zero_copy:
            self._out_host.copy_(self._out_dev, non_blocking=True)
        b.synchronize()
        obs, ef = self._h_obs.copy(), int(self._h_ef[0])
        flags, rew = self._h_af.tolist(), self._h_rew.tolist()
        if self._host_strategies:
            return self._step_host_strategies(obs, flags, rew)
        # the new state is fully determined by the outputs: no state read-back; the mirror arrays are
        # brought up to date lazily (see _Mirror), only the flag lists are maintained per step
        m = self._mirror
        act_l, term_l, trunc_l = self._flag_lists()
        types = self._type_names
        observations, rewards, terminateds, truncateds, infos = {}, {}, {}, {}, {}
        new_done = set()
        for i, aid in enumerate(ids):
            f = flags[i]
            t = f & 1                                           # CCX_AF_TERMINATED
            terminateds[aid] = t != 0
            if f & 4:                                           # CCX_AF_LIVE
                rewards[aid] = rew[i]
                tr = f & 2                                      # CCX_AF_TRUNCATED
                truncateds[aid] = tr != 0
                if t | tr:
                    if not (term_l[i] | trunc_l[i]):
                        new_done.add(aid)
                    if tr:
                        trunc_l[i] = 1
            if t:
                term_l[i] = 1
            act_l[i] = (f >> 6) & 1
            if f & 8:                                           # CCX_AF_OBS
                observations[aid] = obs[i]
                infos[aid] = {"agent_type": types[i], "in_tram_area": (f & 0x10) != 0,
                              "at_door": (f & 0x20) != 0, "active": (f & 0x40) != 0,
                              "at_destination": (f & 0x80) != 0}
        terminateds["__all__"] = (ef & _abi.EF_ALL_TERMINATED) != 0
        truncateds["__all__"] = (ef & _abi.EF_ALL_TRUNCATED) != 0
        m.defer(obs, act_l, term_l, trunc_l)
        m.step_count += 1
        self._agents_truncated_or_terminated_this_step = new_done
        return observations, rewards, terminateds, truncateds, infos

    def _step_host_strategies(self, obs, flags, rew):
        """Slow path for user-registered strategies: the GPU has done the moves and the deactivation
        (collectivecrossing.py:197-212, the O(N^2) part); phases :214-259 run here on the host mirror
        in the reference's order -- rewards, terminated and truncated are all evaluated against the
        PRE-step flags (a reward written as `if agent.terminated: return None` still pays on the step
        the agent finishes), then the flags are applied once, then observations / infos are emitted
        for `agents | finished this step`.  Built-in strategies keep the GPU's values (rewards: the
        f64 from the kernel; observations: the kernel's rows); built-in terminated / truncated rules
        are evaluated through their host methods so that any mix with user classes sees one state.
        The GPU's own flag state is overwritten from the mirror before the next launch."""
        m = self._mirror
        ids, n = self._ids, len(self._ids)
        m._pending = None
        xy = obs[:, :2].astype(np.int32)
        m.x[:], m.y[:] = xy[:, 0], xy[:, 1]
        m.active[:] = [(f >> 6) & 1 for f in flags]
        m.step_count += 1                                        # :188 (the kernel counted it too)
        m.touch()
        rewards, terminateds, truncateds = {}, {}, {}
        gpu_reward = self._reward_function.kernel_mode is not None
        for i, aid in enumerate(ids):                            # :214-217
            if gpu_reward:
                if flags[i] & 4:                                 # CCX_AF_LIVE = not done before the step
                    rewards[aid] = rew[i]
            else:
                r = self._reward_function.calculate_reward(aid, self)
                if r is not None:
                    rewards[aid] = r
        for aid in ids:                                          # :219-222
            t = self._terminated_function.calculate_terminated(aid, self)
            if t is not None:
                terminateds[aid] = t
        for aid in ids:                                          # :224-227
            t = self._truncated_function.calculate_truncated(aid, self)
            if t is not None:
                truncateds[aid] = t
        new_done = set()
        for i, aid in enumerate(ids):                            # :229-241
            if terminateds.get(aid) and not m.terminated[i]:
                m.terminated[i] = 1
                new_done.add(aid)
        for i, aid in enumerate(ids):
            if truncateds.get(aid) and not m.truncated[i]:
                m.truncated[i] = 1
                new_done.add(aid)
        m.dirty = True                                           # the GPU applied the stand-in rules
        m.touch()
        self._agents_truncated_or_terminated_this_step = new_done
        observations, infos = {}, {}
        gpu_obs = self._observation_function.kernel_mode is not None
        types = self._type_names
        for i, aid in enumerate(ids):                            # :243-254
            if (m.terminated[i] or m.truncated[i]) and aid not in new_done:
                continue
            observations[aid] = obs[i] if gpu_obs else self._observation_function.get_agent_observation(aid, self)
            f = flags[i]
            infos[aid] = {"agent_type": types[i], "in_tram_area": (f & 0x10) != 0, "at_door": (f & 0x20) != 0,
                          "active": (f & 0x40) != 0, "at_destination": (f & 0x80) != 0}
        terminateds["__all__"] = all(terminateds.values()) if terminateds else False   # :256-259
        truncateds["__all__"] = all(truncateds.values()) if truncateds else False
        return observations, rewards, terminateds, truncateds, infos

    def close(self) -> None:
        batch = getattr(self, "_batch", None)
        if batch is not None:
            batch.close()

    def render(self, mode: str = "rgb_array"):
        """Rendering is matplotlib drawing in the reference (rendering.py) and never on the step
        path; it is out of scope here."""
        if mode not in ("rgb_array", "human"):
            raise NotImplementedError(f"Render mode {mode} not supported")
        raise NotImplementedError("rendering is out of scope of collectivecrossing_amd (SURVEY 2, row 9)")

    # ------------------------------------------------------------------ host views used by callers
    def _get_agent(self, agent_id) -> Agent:
        if agent_id in self._agents:
            return self._agents[agent_id]
        raise ValueError(f"Unknown agent ID: {agent_id}")

    def _get_agent_position(self, agent_id) -> np.ndarray:
        return self._get_agent(agent_id).position

    def _get_agents_by_type(self, agent_type):
        return [a for a in self._agents.values() if a.agent_type == agent_type]

    def _get_boarding_agents(self):
        return self._get_agents_by_type(AgentType.BOARDING)

    def _get_exiting_agents(self):
        return self._get_agents_by_type(AgentType.EXITING)

    def _get_agent_observation(self, agent_id) -> np.ndarray:
        return self._observation_function.get_agent_observation(agent_id, self)

    def _check_action_and_agent_validity(self, agent_id, action) -> None:
        encode_actions(self._ids, {agent_id: action})

    def _is_valid_position(self, pos) -> bool:
        x, y = int(pos[0]), int(pos[1])
        c, tb = self._config, self._tram_boundaries
        if not (0 <= x <= c.width and 0 <= y <= c.height):
            return False
        if y == c.division_y and not (tb.tram_door_left < x < tb.tram_door_right):
            return False
        return not (y >= c.division_y and not (tb.tram_left < x < tb.tram_right))

    def _is_position_occupied(self, pos, exclude_agent=None) -> bool:
        x, y = int(pos[0]), int(pos[1])
        return any(a.active and a.id != exclude_agent and a.x == x and a.y == y
                   for a in self._agents.values())

    def _would_hit_tram_wall(self, current_pos, new_pos) -> bool:
        x, y = int(new_pos[0]), int(new_pos[1])
        c, tb = self._config, self._tram_boundaries
        if y == c.division_y:
            return not (tb.tram_door_left < x < tb.tram_door_right)
        return y > c.division_y and x in (tb.tram_left, tb.tram_right)

    def _is_move_valid(self, agent_id, current_pos, new_pos) -> bool:
        return (self._is_valid_position(new_pos) and
                not self._is_position_occupied(new_pos, exclude_agent=agent_id) and
                not self._would_hit_tram_wall(current_pos, new_pos))

    def _calculate_new_position(self, agent_id, action) -> np.ndarray:
        return self._get_agent_position(agent_id) + self._action_to_direction[action]

    def is_in_boarding_destination_area(self, agent_id) -> bool:
        return self._get_agent(agent_id).y == self._config.boarding_destination_area_y

    def is_in_exiting_destination_area(self, agent_id) -> bool:
        return self._get_agent(agent_id).y == self._config.exiting_destination_area_y

    def is_in_tram_area(self, agent_id) -> bool:
        a = self._get_agent(agent_id)
        return a.y >= self._config.division_y and self.tram_left <= a.x <= self.tram_right

    def is_at_tram_door(self, agent_id) -> bool:
        a = self._get_agent(agent_id)
        return a.y == self._config.division_y and a.x in (self.tram_door_left - 1, self.tram_door_right + 1)

    def get_agent_destination_position(self, agent_id):
        if self._agents[agent_id].is_boarding:
            return (None, self._config.boarding_destination_area_y)
        return (None, self._config.exiting_destination_area_y)

    def has_agent_reached_destination(self, agent_id) -> bool:
        if self._agents[agent_id].is_boarding:
            return self.is_in_boarding_destination_area(agent_id)
        return self.is_in_exiting_destination_area(agent_id)

    def _calculate_reward(self, agent_id):
        return self._reward_function.calculate_reward(agent_id, self)

    def _calculate_terminated(self, agent_id):
        return self._terminated_function.calculate_terminated(agent_id, self)

    def _calculate_truncated(self, agent_id):
        return self._truncated_function.calculate_truncated(agent_id, self)
