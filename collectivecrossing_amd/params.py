"""Lowering of a validated config to the POD ``ccx_params`` of ``include/ccx.h``.

Geometry follows utils/geometry.py:20-47 of the reference; strategy selection follows the
registries (rewards.py:186-216, terminateds.py:86-114, truncateds.py:99-128,
observations.py:122-149): an unknown strategy name is a ``ValueError`` raised when the env is
built, with the reference's message prefix.
"""

from __future__ import annotations

from dataclasses import dataclass

from ._abi import REWARD_MODES, TERMINATED_MODES, TRUNCATED_MODES, CcxParams
from .configs import CollectiveCrossingConfig


@dataclass(frozen=True)
class TramBoundaries:
    """Absolute tram / door columns (utils/geometry.py:10-17)."""

    tram_door_left: int
    tram_door_right: int
    tram_left: int
    tram_right: int


def calculate_tram_boundaries(config: CollectiveCrossingConfig) -> TramBoundaries:
    centre = config.width // 2
    half = config.tram_length // 2
    left = centre - half
    return TramBoundaries(tram_door_left=left + config.tram_door_left,
                          tram_door_right=left + config.tram_door_right,
                          tram_left=left, tram_right=centre + half)


def _lookup(kind: str, name: str, table: dict[str, int]) -> int:
    if name not in table:
        shown = [k for k in table if k != "custom"] if kind != "truncation" else list(table)
        raise ValueError(f"Unknown {kind} function '{name}'. Available: {', '.join(shown)}")
    return table[name]


def _user_strategy(kind: str, name: str):
    """The registered class behind a strategy name that is not a kernel mode, or None."""
    from . import strategies
    table = {"reward": strategies.REWARD_FUNCTIONS, "termination": strategies.TERMINATED_FUNCTIONS}[kind]
    return table.get(name)


def lower_config(config: CollectiveCrossingConfig, *, allow_position_only: bool = False) -> CcxParams:
    """Config -> ``ccx_params`` (what ``CollectiveCrossingEnv.__init__`` resolves at :59-78).

    ``allow_position_only``: a registered user reward / terminated class that declares ``position_only = True`` is
    accepted -- the params carry a built-in stand-in for it and the caller installs the class's table
    (:func:`position_only_tables`); any other user class keeps raising, with a message that names the single-env path."""
    tb = calculate_tram_boundaries(config)
    obs_name = config.observation_config.get_observation_function_name()
    if obs_name != "default":
        raise ValueError(f"Unknown observation function '{obs_name}'. Available: default")
    rc, tc, uc = config.reward_config, config.terminated_config, config.truncated_config
    p = CcxParams()
    p.width, p.height, p.division_y = config.width, config.height, config.division_y
    p.tram_left, p.tram_right = tb.tram_left, tb.tram_right
    p.door_left, p.door_right = tb.tram_door_left, tb.tram_door_right
    p.num_boarding, p.num_exiting = config.num_boarding_agents, config.num_exiting_agents
    p.boarding_dest_y = config.boarding_destination_area_y
    p.exiting_dest_y = config.exiting_destination_area_y
    p.reward_mode = _mode_or_stand_in("reward", rc.get_reward_function_name(), REWARD_MODES, "constant_negative", allow_position_only)
    p.terminated_mode = _mode_or_stand_in("termination", tc.get_terminated_function_name(), TERMINATED_MODES,
                                          "individual_at_destination", allow_position_only)
    p.truncated_mode = _lookup("truncation", uc.get_truncated_function_name(), TRUNCATED_MODES)
    p.max_steps = int(getattr(uc, "max_steps"))
    # defaults of the strategy configs that the selected strategy does not read
    p.boarding_destination_reward = float(getattr(rc, "boarding_destination_reward", 15.0))
    p.tram_door_reward = float(getattr(rc, "tram_door_reward", 10.0))
    p.tram_area_reward = float(getattr(rc, "tram_area_reward", 5.0))
    p.distance_penalty_factor = float(getattr(rc, "distance_penalty_factor", 0.1))
    p.goal_reward = float(getattr(rc, "goal_reward", 1.0))
    p.no_goal_reward = float(getattr(rc, "no_goal_reward", 0.0))
    p.step_penalty = float(getattr(rc, "step_penalty", -1.0))
    return p


def _mode_or_stand_in(kind: str, name: str, table: dict[str, int], stand_in: str, allow_position_only: bool) -> int:
    if name in table and name != "custom":
        return table[name]
    cls = _user_strategy(kind, name)
    if cls is None:
        return _lookup(kind, name, table)                     # the reference's "Unknown ... function" ValueError
    if getattr(cls, "kernel_mode", None) is not None:
        return int(cls.kernel_mode)
    if allow_position_only and getattr(cls, "position_only", False):
        return table[stand_in]
    raise ValueError(
        f"{kind} function '{name}' is a user-registered class ({cls.__name__}): the batched GPU path runs the built-in "
        f"strategies and user classes that declare `position_only = True` (evaluated once per agent type and cell, "
        f"ccx_set_reward_table / ccx_set_terminated_table).  Anything else is evaluated per step on the host by the "
        f"single-env class collectivecrossing_amd.CollectiveCrossingEnv (E = 1, the documented slow path)")


def position_only_tables(config: CollectiveCrossingConfig):
    """Tables of the config's position-only user strategies: ``(reward, terminated)``, each ``None`` (built-in strategy)
    or a pair of arrays ``[height + 1, width + 1]`` (boarding, exiting) -- f64 rewards / u8 terminated values.

    The user's own ``calculate_reward`` / ``calculate_terminated`` (rewards.py:16-38, terminateds.py:16-36) is CALLED for a
    live probe agent of each type on every cell of the grid, on a host-only env view (no GPU).  What the declaration
    promises is checked as far as a probe can: the value must not change when the other agents stand elsewhere or the step
    counter differs, and a reward must be ``None`` for a terminated agent (the built-in convention, rewards.py:64, which
    the flag byte's LIVE bit reproduces)."""
    import numpy as np

    from . import strategies
    from .env import CollectiveCrossingEnv
    rname = config.reward_config.get_reward_function_name()
    tname = config.terminated_config.get_terminated_function_name()
    rcls = None if rname in REWARD_MODES and rname != "custom" else strategies.REWARD_FUNCTIONS.get(rname)
    tcls = None if tname in TERMINATED_MODES and tname != "custom" else strategies.TERMINATED_FUNCTIONS.get(tname)
    want_r = rcls is not None and rcls.kernel_mode is None and getattr(rcls, "position_only", False)
    want_t = tcls is not None and tcls.kernel_mode is None and getattr(tcls, "position_only", False)
    if not (want_r or want_t):
        return None, None
    env = CollectiveCrossingEnv.host_view(config)
    ids = list(env._agents)
    W, H = config.width, config.height
    probes = {"boarding": next((a for a in ids if a.startswith("boarding_")), None),
              "exiting": next((a for a in ids if a.startswith("exiting_")), None)}

    def place_others(variant: int) -> None:
        for k, a in enumerate(ids):
            ag = env._agents[a]
            ag.position = np.array([(k * 3 + variant * 5) % (W + 1), (k * 2 + variant * 3) % (H + 1)])
        env._mirror.step_count = 1 + 7 * variant

    def evaluate(fn, method: str, dtype, what: str):
        out = []
        for kind in ("boarding", "exiting"):
            tab = np.zeros((H + 1, W + 1), dtype)
            aid = probes[kind]
            if aid is not None:
                for variant in (0, 1):
                    place_others(variant)
                    for y in range(H + 1):
                        for x in range(W + 1):
                            env._agents[aid].position = np.array([x, y])
                            v = getattr(fn, method)(aid, env)
                            if v is None:
                                raise ValueError(f"{type(fn).__name__}.{method} returned None for a live {kind} agent on cell ({x}, {y}): "
                                                 f"a position-only {what} has a value on every cell")
                            v = dtype(v)
                            if variant and v != tab[y, x]:
                                raise ValueError(f"{type(fn).__name__} declares position_only but its {what} on cell ({x}, {y}) changed "
                                                 f"with the other agents' positions / the step counter ({tab[y, x]!r} vs {v!r})")
                            tab[y, x] = v
            out.append(tab)
        return out

    rew = term = None
    if want_r:
        fn = env._reward_function
        rew = evaluate(fn, "calculate_reward", np.float64, "reward")
        for kind, aid in probes.items():                          # the built-in convention: None once done (rewards.py:64)
            if aid is None:
                continue
            ag = env._agents[aid]
            env._mirror.terminated[ids.index(aid)] = 1
            try:
                if fn.calculate_reward(aid, env) is not None:
                    raise ValueError(f"{type(fn).__name__} declares position_only but pays a terminated agent: the batch path "
                                     "hands out rewards for agents that were live before the step only (rewards.py:64)")
            finally:
                env._mirror.terminated[ids.index(aid)] = 0
            del ag
    if want_t:
        term = [t.astype(np.uint8) for t in evaluate(env._terminated_function, "calculate_terminated", np.bool_, "terminated value")]
    return rew, term


def agent_ids(config_or_params) -> list[str]:
    """``boarding_0.. exiting_0..`` in slot order (collectivecrossing.py:101-150)."""
    nb = getattr(config_or_params, "num_boarding_agents", None)
    if nb is None:
        nb, ne = config_or_params.num_boarding, config_or_params.num_exiting
    else:
        ne = config_or_params.num_exiting_agents
    return [f"boarding_{i}" for i in range(nb)] + [f"exiting_{j}" for j in range(ne)]
