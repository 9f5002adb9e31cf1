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
    """E independent CollectiveCrossing envs behind one RLlib ``MultiAgentEnv`` (flat agent ids)."""

    metadata = {"render_modes": [], "render_fps": 4}

    def __init__(self, config: CollectiveCrossingConfig, num_envs: int, device=None):
        self.vector = VectorCollectiveCrossing(config, int(num_envs), device=device)
        self.config = config
        self.num_envs = self.vector.num_envs
        self._local_ids = list(self.vector.agent_ids)
        self._slot = {a: i for i, a in enumerate(self._local_ids)}
        self._flat = [[flat_id(e, a) for a in self._local_ids] for e in range(self.num_envs)]
        self.possible_agents = [f for row in self._flat for f in row]
        # spaces of the reference (collectivecrossing.py:445-477), one per flat id
        self.action_space = self.vector.action_space
        self.observation_space = self.vector.observation_space
        self.action_spaces = {f: self.action_space for f in self.possible_agents}
        self.observation_spaces = {f: self.observation_space for f in self.possible_agents}
        self.np_random: np.random.Generator | None = None
        self._finished = np.ones(self.num_envs, bool)        # no episode is running before the first reset()
        self._ended_by_termination = np.zeros(self.num_envs, bool)
        if _Base is not object:
            super().__init__()

    @classmethod
    def from_env_config(cls, env_config) -> "BatchedMultiAgentEnv":
        """RLlib env factory: the reference's ``env_config`` dict plus ``num_envs`` (and optionally ``device``)."""
        cfg = dict(env_config)
        num_envs = int(cfg.pop("num_envs", 1))
        device = cfg.pop("device", None)
        return cls(CollectiveCrossingConfig(**cfg), num_envs, device=device)

    # ------------------------------------------------------------------ attribute surface
    @property
    def agents(self) -> list[str]:
        """Flat ids that are neither terminated nor truncated, of the envs still running (:743-768 per env)."""
        out = []
        for e in np.flatnonzero(~self._finished):
            out += [flat_id(e, a) for a in self.vector.envs[e].agents]
        return out

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
        self._finished[:] = False
        self._ended_by_termination[:] = False
        types = self.vector._types
        observations, infos = {}, {}
        for e, ids in enumerate(self._flat):
            for i, f in enumerate(ids):
                observations[f] = np.array(rows[e, i])
                infos[f] = {"agent_type": types[i]}                       # reset() :153-159
        return observations, infos

    def step(self, action_dict):
        """Flat ``{"<e>/<agent>": action}`` in, the reference's five dicts out (flat keys).  The order of the entries
        of one env is that env's move order (collectivecrossing.py:197); entries of different envs may interleave."""
        per_env: list[dict] = [{} for _ in range(self.num_envs)]
        for f, action in action_dict.items():
            e, a = split_id(f)
            if not 0 <= e < self.num_envs or a not in self._slot:
                raise ValueError(f"Unknown agent ID: {f} in action_dict. The action_dict keys must be a subset of "
                                 f"the agents. Current agents: {self.agents}")
            if not self._finished[e]:
                per_env[e][a] = action                                   # (bad action values raise in step_dicts)
        self.vector.step_dicts(per_env)
        observations, rewards, terminateds, truncateds, infos = {}, {}, {}, {}, {}
        for e in np.flatnonzero(~self._finished):
            o, r, te, tr, inf = self.vector.view(int(e))
            pre = f"{e}{SEP}"
            all_te, all_tr = te.pop("__all__"), tr.pop("__all__")
            observations.update({pre + k: v for k, v in o.items()})
            rewards.update({pre + k: v for k, v in r.items()})
            terminateds.update({pre + k: v for k, v in te.items()})
            truncateds.update({pre + k: v for k, v in tr.items()})
            infos.update({pre + k: v for k, v in inf.items()})
            if all_te or all_tr:
                self._finished[e] = True
                self._ended_by_termination[e] = bool(all_te)
        done = bool(self._finished.all())
        terminateds["__all__"] = done and bool(self._ended_by_termination.all())
        truncateds["__all__"] = done and not terminateds["__all__"]
        return observations, rewards, terminateds, truncateds, infos

    def close(self) -> None:
        self.vector.close()

    def render(self):
        raise NotImplementedError("rendering is out of scope of collectivecrossing_amd (SURVEY 2, row 9)")
