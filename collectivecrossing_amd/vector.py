"""Vector-env adapter: E envs stepped as ONE batch on the GPU, RLlib-shaped access on top.

The reference parallelises by giving each RLlib EnvRunner process its own single env
(examples/training_script.py:84); callers then stack the per-agent observations of one env and route
them to a policy by id prefix (examples/evaluation_script.py:45-87, 149-195; training_script.py:33-47).
Here one process owns E envs on one GPU:

* **device path** -- ``step(actions)`` takes / returns device tensors; ``last.obs`` ([E, N, L] f32) is a
  torch tensor, so it is DLPack-exportable as it is (``obs_dlpack()``), and ``policy_inputs()`` hands the
  rows out per agent type ("boarding" / "exiting", the two policies of the reference's
  ``policy_mapping_fn``) with the mask of rows the reference would have emitted -- no host round trip;
* **dict path** -- ``step_dicts([...])`` takes one ``action_dict`` per env (reference semantics, dict
  order = move order) and the reference-shaped dicts (``observations, rewards, terminateds, truncateds,
  infos`` with the key-presence rules of collectivecrossing.py:214-261) are built lazily per env from ONE
  pinned host buffer that a single asynchronous device-to-host copy fills after each step
  (``view(e)`` / ``envs[e].last()``); building 32 768 Python dicts per step would cost more than the
  whole GPU step.  ``auto_reset=True`` restarts finished envs on the device with the
  reference's own seeded placement (gymnasium "same-step" convention: the step returns the new
  episode's first observations, the finished episode's last dicts ride along under ``"__final__"``);
* **per-env views** -- ``envs[e]`` exposes the attribute surface callers use on a single reference env:
  ``agents``, ``possible_agents``, ``observation_space(s)``, ``action_space(s)``,
  ``get_observation_space`` / ``get_action_space``, ``config``.
"""

from __future__ import annotations

import numpy as np
import torch

from . import _abi
from .batched import BatchedCollectiveCrossing, StepResult
from .configs import CollectiveCrossingConfig
from .env import decode_step, encode_actions
from .spaces import Box, Discrete


class EnvView:
    """One env of the batch, seen the way callers look at a single ``CollectiveCrossingEnv``."""

    def __init__(self, vec: "VectorCollectiveCrossing", index: int):
        self._vec, self.index = vec, int(index)

    config = property(lambda self: self._vec.config)
    possible_agents = property(lambda self: list(self._vec.agent_ids))
    observation_space = property(lambda self: self._vec.observation_space)
    action_space = property(lambda self: self._vec.action_space)
    observation_spaces = property(lambda self: dict(self._vec.observation_spaces))
    action_spaces = property(lambda self: dict(self._vec.action_spaces))

    def get_observation_space(self, agent_id):
        return self._vec.observation_space

    def get_action_space(self, agent_id):
        return self._vec.action_space

    @property
    def agents(self) -> list[str]:
        """Ids that are neither terminated nor truncated (collectivecrossing.py:743-768)."""
        done = self._vec._done_flags()[self.index]
        return [a for a, d in zip(self._vec.agent_ids, done) if not d]

    def last(self):
        """The five reference dicts of this env for the last step."""
        return self._vec.view(self.index)

    def observations(self) -> dict[str, np.ndarray]:
        return self._vec.view(self.index)[0]

    # A view is not a steppable env: the E envs of the batch advance TOGETHER in one kernel launch.
    def _batch_only(self, what: str):
        raise RuntimeError(
            f"EnvView.{what}() is not available: env {self.index} is one of {self._vec.num_envs} envs of a batch that is "
            "stepped as a whole on the GPU.  Use VectorCollectiveCrossing.step / step_dicts / reset (all envs at once), "
            "collectivecrossing_amd.rllib.BatchedMultiAgentEnv (one RLlib MultiAgentEnv over the batch, flat agent ids), "
            "or collectivecrossing_amd.CollectiveCrossingEnv for a single env with the reference's reset()/step().")

    def reset(self, *, seed=None, options=None):
        self._batch_only("reset")

    def step(self, action_dict):
        self._batch_only("step")


class VectorCollectiveCrossing:
    """E independent envs; array API in, array API out, dict views on demand."""

    def __init__(self, config: CollectiveCrossingConfig, num_envs: int, device=None,
                 env_offset: int = 0, total_envs: int | None = None, check_inputs: bool | None = None):
        self.batch = BatchedCollectiveCrossing(config, num_envs, device, env_offset, total_envs,
                                               check_inputs=check_inputs)
        self.config = config
        self.num_envs = self.batch.num_envs
        self.agent_ids = self.batch.agent_ids
        self.possible_agents = list(self.agent_ids)
        nb = config.num_boarding_agents
        self.num_boarding = nb
        N, L = len(self.agent_ids), self.batch.obs_len
        self._types = ["boarding" if i < nb else "exiting" for i in range(N)]
        # spaces of the reference (collectivecrossing.py:445-477, observations.py:96-118): one per agent id
        self.action_space = Discrete(5)
        self.observation_space = Box(low=-1, high=max(config.width, config.height) - 1, shape=(L,), dtype=np.float32)
        self.action_spaces = {a: self.action_space for a in self.agent_ids}
        self.observation_spaces = {a: self.observation_space for a in self.agent_ids}
        self.envs = [EnvView(self, e) for e in range(self.num_envs)]
        self.last: StepResult | None = None
        # ONE device buffer and ONE pinned host buffer hold all outputs of a step, 16-byte aligned segments
        # [obs f32 E*N*L | reward f64 E*N | agent_flags u8 E*N | env_flags u8 E]
        E = self.num_envs
        a16 = lambda v: (v + 15) & ~15  # noqa: E731
        self._o_rew = a16(E * N * L * 4)
        self._o_af = self._o_rew + a16(E * N * 8)
        self._o_ef = self._o_af + a16(E * N)
        total = self._o_ef + a16(E)
        dev = self.batch.device
        self._dev = torch.empty(total, dtype=torch.uint8, device=dev)
        self._host = torch.empty(total, dtype=torch.uint8).pin_memory()
        self._copied = torch.cuda.Event()
        self._host_valid = False
        d = self._dev
        self._out = StepResult(d[:E * N * L * 4].view(torch.float32).view(E, N, L),
                               d[self._o_rew:self._o_rew + E * N * 8].view(torch.float64).view(E, N),
                               d[self._o_af:self._o_af + E * N].view(E, N), d[self._o_ef:self._o_ef + E])
        h = self._host.numpy()
        self._h_obs = h[:E * N * L * 4].view(np.float32).reshape(E, N, L)
        self._h_rew = h[self._o_rew:self._o_rew + E * N * 8].view(np.float64).reshape(E, N)
        self._h_af = h[self._o_af:self._o_af + E * N].reshape(E, N)
        self._h_ef = h[self._o_ef:self._o_ef + E]
        # cumulative done flags per agent (terminated | truncated), maintained on the device
        self._done = torch.zeros((E, N), dtype=torch.bool, device=dev)
        self._done_host: np.ndarray | None = None
        self._episodes = np.zeros(E, np.int64)
        self._final: dict[int, tuple] = {}
        self._reset_rows = None
        # envs restarted by step_dicts(auto_reset=True) after the last step: their rows in `last.obs` are the NEW
        # episode's first observations and every agent's row counts as handed out (policy_inputs' mask)
        self._restarted = torch.zeros((E,), dtype=torch.bool, device=dev)

    # ------------------------------------------------------------------ batch API (device)
    def reset(self, seeds, env_mask=None) -> torch.Tensor:
        """``reset(seed=seeds[e])`` of the (masked) envs on the device; obs tensor [E, N, L]."""
        obs = self.batch.reset(seeds, env_mask)
        if env_mask is None:
            self._done.zero_()
        else:
            m = torch.as_tensor(np.asarray(env_mask) if not isinstance(env_mask, torch.Tensor) else env_mask)
            self._done[m.to(self._done.device).bool()] = False
        self.last, self._host_valid, self._done_host = None, False, None
        self._reset_rows = None
        self._restarted.zero_()
        return obs

    def step(self, actions, order=None) -> StepResult:
        """``actions`` u8 [E, N] (255 = agent absent), optional move order; device tensors out (views of
        one buffer, overwritten by the next step)."""
        b = self.batch
        a = b._as_dev_u8(actions, (self.num_envs, len(self.agent_ids)))
        o = None if order is None else b._as_dev_u8(order, (self.num_envs, len(self.agent_ids)))
        out = self._out
        so = _abi.CcxStepOut(out.obs.data_ptr(), out.reward.data_ptr(), out.agent_flags.data_ptr(),
                             out.env_flags.data_ptr())
        import ctypes as C
        from ._lib import check
        check(b._lib.ccx_step(b._h, C.c_void_p(a.data_ptr()), C.c_void_p(None if o is None else o.data_ptr()), C.byref(so)))
        self.last = out
        with torch.cuda.stream(b._stream):
            self._done |= (out.agent_flags & (_abi.AF_TERMINATED | _abi.AF_TRUNCATED)) != 0
            self._restarted.zero_()
        self._host_valid, self._done_host = False, None
        self._final = {}
        return out

    def done_mask(self) -> torch.Tensor:
        """u8 [E]: envs whose last step raised ``__all__`` terminated or truncated."""
        if self.last is None:
            raise RuntimeError("no step yet")
        return ((self.last.env_flags & (_abi.EF_ALL_TERMINATED | _abi.EF_ALL_TRUNCATED)) != 0).to(torch.uint8)

    def reset_done(self, seeds) -> torch.Tensor:
        """Restart exactly the envs that finished (seeded, on the device); returns the done mask."""
        m = self.done_mask()
        self.batch.reset(seeds, env_mask=m)
        with torch.cuda.stream(self.batch._stream):
            self._done[m.bool()] = False
        self._done_host = None
        return m

    def obs_dlpack(self):
        """The last step's observation tensor [E, N, L] as a DLPack capsule (zero-copy hand-over to any
        framework; inside torch simply use ``last.obs``)."""
        if self.last is None:
            raise RuntimeError("no step yet")
        self.batch.synchronize()
        return torch.utils.dlpack.to_dlpack(self.last.obs)

    def policy_inputs(self, obs: torch.Tensor | None = None) -> dict[str, dict[str, torch.Tensor]]:
        """Observation rows grouped the way the reference's callers feed its two policies
        (evaluation_script.py:45-87: rows of ids containing "boarding" / "exiting", stacked):
        ``{"boarding": {"obs": [E, Nb, L], "mask": [E, Nb]}, "exiting": {...}}`` on the device; ``mask`` marks
        the rows the reference would have handed out (agents not done before the step, CCX_AF_OBS)."""
        if obs is None:
            if self.last is None:
                raise RuntimeError("no step yet")
            obs = self.last.obs
            emitted = ((self.last.agent_flags & _abi.AF_OBS) != 0) | self._restarted[:, None]
        else:
            emitted = torch.ones(obs.shape[:2], dtype=torch.bool, device=obs.device)
        nb = self.num_boarding
        return {"boarding": {"obs": obs[:, :nb], "mask": emitted[:, :nb]},
                "exiting": {"obs": obs[:, nb:], "mask": emitted[:, nb:]}}

    # ------------------------------------------------------------------ dict API
    def step_dicts(self, action_dicts, auto_reset: bool = False, seed0: int = 0):
        """One ``action_dict`` per env (reference semantics incl. dict order = move order; unknown ids and
        actions outside 0..4 raise the reference's ``ValueError`` before anything moves).  With
        ``auto_reset`` every env whose step raised ``__all__`` is restarted on the device with
        ``reset(seed = seed0 + episode * num_envs + e)``; ``view(e)`` then returns the new episode's first
        observations (rewards / flags of the finished step) and the finished episode's own five dicts under
        ``infos["__final__"]``."""
        E, N = self.num_envs, len(self.agent_ids)
        if len(action_dicts) != E:
            raise ValueError(f"need {E} action dicts, got {len(action_dicts)}")
        a = np.empty((E, N), np.uint8)
        o = np.empty((E, N), np.uint8)
        for e, d in enumerate(action_dicts):
            a[e], o[e] = encode_actions(self.agent_ids, d)
        # every dict in slot order (the usual `{a: ... for a in env.agents}`): no move-order tensor, plain kernel path
        out = self.step(a, None if bool((o == np.arange(N, dtype=np.uint8)).all()) else o)
        if auto_reset:
            self._pull()
            done = (self._h_ef & (_abi.EF_ALL_TERMINATED | _abi.EF_ALL_TRUNCATED)) != 0
            if done.any():
                for e in np.flatnonzero(done):
                    self._final[int(e)] = self._decode(int(e))
                self._episodes[done] += 1
                seeds = (seed0 + self._episodes * E + np.arange(E)).astype(np.uint64)
                obs = self.batch.reset(seeds, env_mask=done.astype(np.uint8))
                with torch.cuda.stream(self.batch._stream):
                    m = torch.from_numpy(done).to(self._done.device)
                    self._done[m] = False
                    # the device path sees the new episode too: the restarted envs' rows of `last.obs` (what
                    # policy_inputs() / obs_dlpack() hand out) become the reset observations, like view(e) on the host;
                    # the terminal rows stay available under infos["__final__"] only
                    self._out.obs.copy_(torch.where(m[:, None, None], obs, self._out.obs))
                    self._restarted.copy_(m)
                self._done_host = None
                self._reset_rows = (done, obs.cpu().numpy())
            else:
                self._reset_rows = None
        else:
            self._reset_rows = None
        return out

    # ------------------------------------------------------------------ lazy host views
    def _pull(self) -> None:
        """ONE asynchronous device-to-host copy of all outputs into the pinned buffer, then wait for it."""
        if self._host_valid:
            return
        if self.last is None:
            raise RuntimeError("no step yet")
        with torch.cuda.stream(self.batch._stream):
            self._host.copy_(self._dev, non_blocking=True)
            self._copied.record(self.batch._stream)
        self._copied.synchronize()
        self._host_valid = True

    def _done_flags(self) -> np.ndarray:
        if self._done_host is None:
            self.batch.synchronize()
            self._done_host = self._done.cpu().numpy()
        return self._done_host

    def _decode(self, e: int):
        return decode_step(self.agent_ids, self._h_obs[e], self._h_rew[e], self._h_af[e], int(self._h_ef[e]), self._types)

    def view(self, env_index: int):
        """The five reference dicts of env ``env_index`` for the last step."""
        self._pull()
        e = int(env_index)
        if e in self._final and self._reset_rows is not None:
            observations, rewards, terminateds, truncateds, infos = self._final[e]
            done, rows = self._reset_rows
            new_obs = {aid: np.array(rows[e, i]) for i, aid in enumerate(self.agent_ids)}
            new_infos = {aid: {"agent_type": self._types[i]} for i, aid in enumerate(self.agent_ids)}   # reset() :153-159
            new_infos["__final__"] = (observations, rewards, terminateds, truncateds, infos)
            return new_obs, rewards, terminateds, truncateds, new_infos
        return self._decode(e)

    def close(self) -> None:
        self.batch.close()
