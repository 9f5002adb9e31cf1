"""An object RLlib can drive: E envs of one GPU batch as ONE ``MultiAgentEnv`` with flat agent ids.

The reference's callers do ``register_env(name, lambda env_config: CollectiveCrossingEnv(config=
CollectiveCrossingConfig(**env_config)))`` + ``DQNConfig().environment(env=name, env_config=...)`` and route
agents to two policies by the id prefix (examples/training_script.py:26-29, 33-47, 69-86); at inference they
stack the rows of the ids containing "boarding" / "exiting" (examples/evaluation_script.py:149-195).  Every
EnvRunner process then owns ONE env (training_script.py:84).  Here one process owns E envs on one MI355X and
hands them to RLlib as one multi-agent env whose agents are ``"{e}/boarding_{i}"`` / ``"{e}/exiting_{j}"``:

    register_env("collective_crossing_batch", BatchedMultiAgentEnv.from_env_config)
    DQNConfig().environment(env="collective_crossing_batch", env_config={**env_config, "num_envs": 256})
               .multi_agent(policies={"boarding", "exiting"}, policy_mapping_fn=policy_mapping_fn)

Per env the dicts follow the reference's key-presence rules (collectivecrossing.py:214-261) -- ``rewards`` /
``truncateds`` for agents that were live before the step, ``terminateds`` for every agent, ``observations`` /
``infos`` for live agents and those that finished this step -- with the env index in front of every key.  An env
whose own ``__all__`` has been raised is FINISHED: its agents leave ``agents``, entries for them are no longer
returned and actions for them are ignored until the next ``reset``.  The flat ``"__all__"`` flags are raised when
every env of the batch has finished: ``terminateds["__all__"]`` if all of them ended by termination,
``truncateds["__all__"]`` otherwise.

``reset(seed=s)`` places env e with the reference's own ``reset(seed=s + e)`` (bit-identical, on the device);
``seed=None`` draws a fresh base seed from the adapter's generator (the reference would continue one PCG64
stream; a batch needs one seed per env).

This is the dict path: it builds O(E x N) Python objects per step.  A learner that lives on the GPU takes
``env.vector`` (:class:`VectorCollectiveCrossing`: device tensors, ``policy_inputs()``, DLPack) instead.
"""

from __future__ import annotations

import numpy as np

from .configs import CollectiveCrossingConfig
from .params import agent_ids
from .spaces import Box, Discrete
from .vector import VectorCollectiveCrossing

try:  # RLlib is optional; with it the class is a real MultiAgentEnv (same pattern as env.py)
    from ray.rllib.env.multi_agent_env import MultiAgentEnv as _Base  # type: ignore
except Exception:
    _Base = object

SEP = "/"


def flat_id(env_index: int, agent_id: str) -> str:
    return f"{int(env_index)}{SEP}{agent_id}"


def split_id(flat_agent_id: str) -> tuple[int, str]:
    """``"17/boarding_3"`` -> ``(17, "boarding_3")``; anything else raises the reference's ``ValueError`` text."""
    head, sep, tail = str(flat_agent_id).partition(SEP)
    if not sep or not head.isdigit():
        raise ValueError(f"Unknown agent ID: {flat_agent_id} in action_dict. Batched ids look like "
                         f"'<env index>{SEP}boarding_<i>' / '<env index>{SEP}exiting_<j>'")
    return int(head), tail


def policy_mapping_fn(agent_id: str, *args, **kwargs) -> str:
    """The reference's mapping (examples/training_script.py:33-47: ``agent_id.startswith("boarding_")``) for plain
    AND flat ids: the type prefix is looked at after the env index."""
    return "boarding" if str(agent_id).rpartition(SEP)[2].startswith("boarding_") else "exiting"


class BatchedMultiAgentEnv(_Base):
    """E independent CollectiveCrossing envs behind one RLlib ``MultiAgentEnv`` (flat agent ids).

    ``auto_reset=True``: the batch never drains.  An env whose own ``__all__`` rises is restarted on the device right
    behind that step (the reference's ``reset(seed = seed0 + episode * num_envs + e)``): the step returns the finished
    step's rewards / terminateds / truncateds for its agents, the NEW episode's first observations under the same flat
    ids (every agent of the env) and the finished episode's own five dicts (local ids) under ``infos["<e>/__final__"]``
    -- gymnasium's same-step convention, what ``VectorCollectiveCrossing.step_dicts(auto_reset=True)`` does per env;
    RLlib restarts each env when its ``__all__`` rises (examples/training_script.py:26-29, 69-86), and the flat
    ``"__all__"`` flags stay False."""

    metadata = {"render_modes": [], "render_fps": 4}

    def __init__(self, config: CollectiveCrossingConfig, num_envs: int, device=None, auto_reset: bool = False,
                 seed0: int = 0, *, _host_only: bool = False):
        # (_host_only: the id / dict plumbing without a GPU batch behind it -- CPU tests of `_encode` / `_dicts`; such an
        # object cannot reset or step)
        self.vector = None if _host_only else VectorCollectiveCrossing(config, int(num_envs), device=device)
        self.config = config
        self.num_envs = int(num_envs)
        self.auto_reset = bool(auto_reset)
        self.seed0 = int(seed0)
        self._local_ids = list(agent_ids(config))
        self._slot = {a: i for i, a in enumerate(self._local_ids)}
        E, N = self.num_envs, len(self._local_ids)
        self._flat = [[flat_id(e, a) for a in self._local_ids] for e in range(E)]
        # the flat ids as ONE object array [E, N] (the dicts of a step are built from slices of it: the very same str
        # objects every step) and their inverse
        self._keys = np.empty((E, N), dtype=object)
        for e, row in enumerate(self._flat):
            self._keys[e, :] = row
        self._key_index = {f: k for k, f in enumerate(self._keys.ravel().tolist())}
        self._possible_agents = self._keys.ravel().tolist()
        # spaces of the reference (collectivecrossing.py:445-477), one per flat id
        self.action_space = Discrete(5)
        self.observation_space = Box(low=-1, high=max(config.width, config.height) - 1, shape=(6 + 4 * N,), dtype=np.float32)
        self.action_spaces = {f: self.action_space for f in self._possible_agents}
        self.observation_spaces = {f: self.observation_space for f in self._possible_agents}
        self.np_random: np.random.Generator | None = None
        self._started = False                                # no episode is running before the first reset()
        self._finished = np.zeros(E, bool)
        self._ended_by_termination = np.zeros(E, bool)
        self._done_agents = np.zeros((E, N), bool)           # terminated | truncated, per agent (host copy)
        self._episodes = np.zeros(E, np.int64)
        nb = config.num_boarding_agents
        types = ["boarding" if i < nb else "exiting" for i in range(N)]
        # infos[id] has 2 x 16 possible values (agent type x the four info bits of the flag byte): templates, copied per agent
        self._info_tpl = {(t, b): {"agent_type": t, "in_tram_area": bool(b & 1), "at_door": bool(b & 2), "active": bool(b & 4),
                                   "at_destination": bool(b & 8)} for t in ("boarding", "exiting") for b in range(16)}
        self._type_row = np.array(types, dtype=object)
        self._is_exiting = np.array([t == "exiting" for t in types])
        self.last_step_host_us: dict[str, float] = {}
        if _Base is not object:
            super().__init__()

    @classmethod
    def from_env_config(cls, env_config) -> "BatchedMultiAgentEnv":
        """RLlib env factory: the reference's ``env_config`` dict plus ``num_envs`` (and optionally ``device``,
        ``auto_reset``, ``seed0``)."""
        cfg = dict(env_config)
        num_envs = int(cfg.pop("num_envs", 1))
        device = cfg.pop("device", None)
        auto_reset = bool(cfg.pop("auto_reset", False))
        seed0 = int(cfg.pop("seed0", 0))
        return cls(CollectiveCrossingConfig(**cfg), num_envs, device=device, auto_reset=auto_reset, seed0=seed0)

    # ------------------------------------------------------------------ attribute surface
    @property
    def possible_agents(self) -> list[str]:
        return list(self._possible_agents)

    @possible_agents.setter
    def possible_agents(self, value) -> None:   # (RLlib's MultiAgentEnv.__init__ may assign its own inference: ignored)
        pass

    @property
    def agents(self) -> list[str]:
        """Flat ids that are neither terminated nor truncated, of the envs still running (:743-768 per env).  Before the
        first ``reset()`` every possible agent is listed, like the reference's dummy agents (:80-86): RLlib's
        ``MultiAgentEnv.__init__`` infers its agent set from a non-empty ``agents``."""
        if not self._started:
            return list(self._possible_agents)
        live = ~self._done_agents & ~self._finished[:, None]
        return self._keys[live].tolist()

    @agents.setter
    def agents(self, value) -> None:            # (assigned by RLlib's base class when it infers the agent set: ignored)
        pass

    def get_observation_space(self, agent_id):
        return self.observation_space

    def get_action_space(self, agent_id):
        return self.action_space

    # ------------------------------------------------------------------ reset / step
    def reset(self, *, seed: int | None = None, options: dict | None = None):
        if seed is not None or self.np_random is None:
            self.np_random = np.random.Generator(np.random.PCG64(np.random.SeedSequence(seed)))
            base = int(seed) if seed is not None else int(self.np_random.integers(0, 2**62))
        else:
            base = int(self.np_random.integers(0, 2**62))
        seeds = (np.uint64(base) + np.arange(self.num_envs, dtype=np.uint64)).astype(np.uint64)
        rows = self.vector.reset(seeds).cpu().numpy()
        self._started = True
        self._finished[:] = False
        self._ended_by_termination[:] = False
        self._done_agents[:] = False
        self._episodes[:] = 0
        keys = self._possible_agents
        observations = dict(zip(keys, rows.reshape(len(keys), -1)))
        infos = dict(zip(keys, [{"agent_type": t} for t in np.broadcast_to(self._type_row, self._keys.shape).ravel().tolist()]))   # reset() :153-159
        return observations, infos

    def set_state(self, **state) -> None:
        """Forced state for every env (``BatchedCollectiveCrossing.set_state``: x, y, active, terminated, truncated,
        step_count as [E, N] / [E] arrays); the adapter's own done flags follow the terminated / truncated arrays."""
        self.vector.batch.set_state(**state)
        import torch
        done = np.zeros(self._keys.shape, bool)
        for k in ("terminated", "truncated"):
            if state.get(k) is not None:
                done |= np.asarray(state[k]).astype(bool)
        self._done_agents[:] = done
        self.vector._done.copy_(torch.from_numpy(done))

    def _encode(self, action_dict):
        """Flat action dict -> (actions u8 [E, N] with 255 = absent, move order u8 [E, N] or None); raises the reference's
        ``ValueError``s (collectivecrossing.py:685-711: unknown id first, then the action's value) before anything moves."""
        E, N = self._keys.shape
        n = len(action_dict)
        a = np.full(E * N, 255, np.uint8)
        if n == 0:
            return a.reshape(E, N), None
        try:
            idx = np.fromiter(map(self._key_index.__getitem__, action_dict), dtype=np.int64, count=n)
        except KeyError:
            bad = next(f for f in action_dict if f not in self._key_index)
            split_id(bad)                                     # (ids without an env index get their own message)
            raise ValueError(f"Unknown agent ID: {bad} in action_dict. The action_dict keys must be a subset of "
                             f"the agents. Current agents: {self.agents}") from None
        vals = list(action_dict.values())
        act = None
        try:
            arr = np.asarray(vals)
            if arr.dtype.kind in "iu" and arr.ndim == 1 and (arr >= 0).all() and (arr <= 4).all():
                act = arr.astype(np.uint8)
        except Exception:
            act = None
        if act is None:                                        # slow path: the reference's own membership test, first offender
            from .actions import ACTION_TO_DIRECTION
            for f, v in action_dict.items():
                if v not in ACTION_TO_DIRECTION:
                    raise ValueError(f"Invalid action: {v} for agent {f}. Valid actions are: {list(ACTION_TO_DIRECTION)}")
            act = np.asarray([int(v) for v in vals], np.uint8)
        env_of, slot_of = np.divmod(idx, N)
        keep = ~self._finished[env_of]                         # entries of finished envs are ignored
        a[idx[keep]] = act[keep]
        # move order (collectivecrossing.py:197: dict order, per env).  Slot order within every env = no order tensor.
        perm = np.argsort(env_of, kind="stable")
        es, ss = env_of[perm], slot_of[perm]
        same = es[1:] == es[:-1]
        if not np.any(same & (ss[1:] <= ss[:-1])):
            return a.reshape(E, N), None
        o = np.empty((E, N), np.uint8)
        o[:] = np.arange(N, dtype=np.uint8)
        starts = np.flatnonzero(np.r_[True, ~same])
        counts = np.diff(np.r_[starts, len(es)])
        pos = np.arange(len(es)) - np.repeat(starts, counts)
        full = np.repeat(counts == N, counts)
        o[es[full], pos[full]] = ss[full]
        for k in np.flatnonzero(counts != N):                  # envs with absent agents: listed first, the rest in slot order
            e, listed = int(es[starts[k]]), ss[starts[k]:starts[k] + counts[k]].tolist()
            seen = set(listed)
            o[e] = listed + [i for i in range(N) if i not in seen]
        return a.reshape(E, N), o

    def _dicts(self, env_sel: np.ndarray, af: np.ndarray, rew: np.ndarray, obs: np.ndarray, keys: np.ndarray):
        """The reference's five dicts (collectivecrossing.py:214-261, without ``__all__``) of the envs selected by the
        boolean mask, built from array slices: keys = ``keys[env, slot]``."""
        sel = env_sel[:, None]
        emit = ((af & 8) != 0) & sel
        live = ((af & 4) != 0) & sel
        everyone = np.broadcast_to(sel, af.shape)
        k_emit = keys[emit].tolist()
        observations = dict(zip(k_emit, obs[emit]))
        rewards = dict(zip(keys[live].tolist(), rew[live].tolist()))
        terminateds = dict(zip(keys[everyone].tolist(), ((af & 1) != 0)[everyone].tolist()))
        truncateds = dict(zip(keys[live].tolist(), ((af & 2) != 0)[live].tolist()))
        tpl = self._info_tpl
        bits = (af >> 4)[emit].tolist()
        types = np.broadcast_to(self._type_row, af.shape)[emit].tolist()
        infos = dict(zip(k_emit, [tpl[t, b].copy() for t, b in zip(types, bits)]))
        return observations, rewards, terminateds, truncateds, infos

    def step(self, action_dict):
        """Flat ``{"<e>/<agent>": action}`` in, the reference's five dicts out (flat keys).  The order of the entries
        of one env is that env's move order (collectivecrossing.py:197); entries of different envs may interleave."""
        import time
        t0 = time.perf_counter()
        if not self._started:
            raise RuntimeError("step() before reset()")
        a, o = self._encode(action_dict)
        t1 = time.perf_counter()
        v = self.vector
        v.step(a, o)
        v._pull()
        t2 = time.perf_counter()
        af, ef, rew, obs = v._h_af, v._h_ef, v._h_rew, v._h_obs
        running = ~self._finished
        observations, rewards, terminateds, truncateds, infos = self._dicts(running, af, rew, obs, self._keys)
        self._done_agents |= ((af & 3) != 0) & running[:, None]
        env_done = ((ef & 3) != 0) & running
        if env_done.any():
            if self.auto_reset:
                E = self.num_envs
                for e in np.flatnonzero(env_done).tolist():    # the finished episode's own dicts, local ids
                    one = np.zeros(E, bool)
                    one[e] = True
                    fo, fr, fte, ftr, finf = self._dicts(one, af, rew, obs, np.broadcast_to(np.array(self._local_ids, dtype=object), af.shape))
                    fte["__all__"], ftr["__all__"] = bool(ef[e] & 1), bool(ef[e] & 2)
                    infos[f"{e}{SEP}__final__"] = (fo, fr, fte, ftr, finf)
                self._episodes[env_done] += 1
                seeds = (self.seed0 + self._episodes * E + np.arange(E)).astype(np.uint64)
                rows = v.reset(seeds, env_mask=env_done.astype(np.uint8)).cpu().numpy()
                self._done_agents[env_done] = False
                sel = np.broadcast_to(env_done[:, None], af.shape)
                k_new = self._keys[sel].tolist()
                observations.update(zip(k_new, rows[env_done].reshape(len(k_new), -1)))
                infos.update(zip(k_new, [{"agent_type": t} for t in np.broadcast_to(self._type_row, af.shape)[sel].tolist()]))
            else:
                self._finished |= env_done
                self._ended_by_termination[env_done] = (ef[env_done] & 1) != 0
        done = bool(self._finished.all())
        terminateds["__all__"] = done and bool(self._ended_by_termination.all())
        truncateds["__all__"] = done and not terminateds["__all__"]
        t3 = time.perf_counter()
        self.last_step_host_us = {"encode": (t1 - t0) * 1e6, "launch_and_copy": (t2 - t1) * 1e6, "dicts": (t3 - t2) * 1e6,
                                  "total": (t3 - t0) * 1e6}
        return observations, rewards, terminateds, truncateds, infos

    def close(self) -> None:
        if self.vector is not None:
            self.vector.close()

    def render(self):
        raise NotImplementedError("rendering is out of scope of collectivecrossing_amd (SURVEY 2, row 9)")
