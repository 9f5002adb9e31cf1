"""Vector-env adapter: E envs stepped as ONE batch on the GPU, dict views built lazily per env.

The reference parallelises by giving each RLlib EnvRunner process its own single env
(examples/training_script.py:84).  Here one process owns E envs on one GPU; policies consume the
observation tensor directly on the device (``last.obs`` is a torch tensor, zero-copy), and the
reference-shaped per-env dicts (``observations, rewards, terminateds, truncateds, infos`` with
the key-presence rules of collectivecrossing.py:214-261) are only materialised for the envs somebody
asks for -- building 32 768 Python dicts per step would cost more than the whole GPU step
(SURVEY 8 f-3).
"""

from __future__ import annotations

import numpy as np
import torch

from . import _abi
from .batched import BatchedCollectiveCrossing, StepResult
from .configs import CollectiveCrossingConfig
from .env import decode_step, encode_actions


class VectorCollectiveCrossing:
    """E independent envs; array API in, array API out, dict views on demand."""

    def __init__(self, config: CollectiveCrossingConfig, num_envs: int, device=None,
                 env_offset: int = 0, total_envs: int | None = None):
        self.batch = BatchedCollectiveCrossing(config, num_envs, device, env_offset, total_envs)
        self.config = config
        self.num_envs = self.batch.num_envs
        self.agent_ids = self.batch.agent_ids
        nb = config.num_boarding_agents
        self._types = ["boarding" if i < nb else "exiting" for i in range(len(self.agent_ids))]
        self.last: StepResult | None = None
        self._host: tuple | None = None

    # ------------------------------------------------------------------ batch API
    def reset(self, seeds, env_mask=None) -> torch.Tensor:
        """``reset(seed=seeds[e])`` of the (masked) envs on the device; obs tensor [E, N, L]."""
        self.last, self._host = None, None
        return self.batch.reset(seeds, env_mask)

    def step(self, actions, order=None) -> StepResult:
        """``actions`` u8 [E, N] (255 = agent absent), optional move order; device tensors out."""
        self.last = self.batch.step(actions, order)
        self._host = None
        return self.last

    def step_dicts(self, action_dicts) -> StepResult:
        """One ``action_dict`` per env (reference semantics incl. dict order = move order)."""
        E, N = self.num_envs, len(self.agent_ids)
        if len(action_dicts) != E:
            raise ValueError(f"need {E} action dicts, got {len(action_dicts)}")
        a = np.empty((E, N), np.uint8)
        o = np.empty((E, N), np.uint8)
        for e, d in enumerate(action_dicts):
            a[e], o[e] = encode_actions(self.agent_ids, d)
        return self.step(a, o)

    def done_mask(self) -> torch.Tensor:
        """u8 [E]: envs whose last step raised ``__all__`` terminated or truncated."""
        if self.last is None:
            raise RuntimeError("no step yet")
        return ((self.last.env_flags & (_abi.EF_ALL_TERMINATED | _abi.EF_ALL_TRUNCATED)) != 0).to(torch.uint8)

    def reset_done(self, seeds) -> torch.Tensor:
        """Restart exactly the envs that finished (seeded, on the device); returns the done mask."""
        m = self.done_mask()
        self.batch.reset(seeds, env_mask=m)
        return m

    # ------------------------------------------------------------------ lazy dict views
    def _pull(self) -> tuple:
        if self._host is None:
            if self.last is None:
                raise RuntimeError("no step yet")
            r = self.last
            self._host = (r.obs.cpu().numpy(), r.reward.cpu().numpy(), r.agent_flags.cpu().numpy(),
                          r.env_flags.cpu().numpy())
        return self._host

    def view(self, env_index: int):
        """The five reference dicts of env ``env_index`` for the last step."""
        obs, rew, af, ef = self._pull()
        e = int(env_index)
        return decode_step(self.agent_ids, obs[e], rew[e], af[e], int(ef[e]), self._types)

    def close(self) -> None:
        self.batch.close()
