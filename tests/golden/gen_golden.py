#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by RUNNING THE REFERENCE in the build container.

The reference (pure Python, /root/reference) is imported here and only here; it never travels to
the GPU box.  Its module tops import ``gymnasium`` and ``ray`` which are not installed, so the
tiny stand-ins of ``_refshim/`` (base classes + numpy seeding, see its README) go first on
``sys.path``.  Before writing anything the script replays the reference's OWN committed goldens
(seed 42) through the imported reference, which proves the stand-in seeding consumes the PCG64
stream exactly like real gymnasium did when those goldens were recorded.

Every fixture is DATA: the config, the initial state, the per-step inputs (actions, move order) and
the reference's outputs re-encoded in the array contract of include/ccx.h:

  config_json                       str   kwargs of CollectiveCrossingConfig (+ "_relaxed" flag)
  init_{x,y}[E,N] i32, init_{active,terminated,truncated}[E,N] u8, init_step_count[E] i32
  actions[K,E,N] u8 (255 = agent not in action_dict), order[K,E,N] u8 (dict iteration order)
  x,y[K,E,N] i32; active,terminated,truncated[K,E,N] u8; step_count[K,E] i32   (post-step state)
  reward[K,E,N] f64 (0 where not LIVE), agent_flags[K,E,N] u8, env_flags[K,E] u8
  obs[K,E,N,L] f32 (DefaultObservation of every agent, from the reference's observation function)
  (rollout fixtures also) pool_xy[P,N,2] u8 = reference reset(seed=seed0+p) placements, seed0

Usage: python tests/golden/gen_golden.py            (no-op when /root/reference is absent)
"""

from __future__ import annotations

import json
import os
import sys
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
REF = Path(os.environ.get("CCX_REFERENCE", "/root/reference"))

AF = dict(TERMINATED=1, TRUNCATED=2, LIVE=4, OBS=8, IN_TRAM=16, AT_DOOR=32, ACTIVE=64, AT_DEST=128)
EF = dict(ALL_TERM=1, ALL_TRUNC=2, RESET=4)
ABSENT = 255


def import_reference():
    sys.path.insert(0, str(REF / "src"))
    sys.path.insert(0, str(HERE / "_refshim"))
    os.environ.setdefault("MPLBACKEND", "Agg")
    import collectivecrossing as cc  # noqa: F401  (the reference)
    from collectivecrossing import configs, observation_configs, reward_configs  # noqa: F401
    from collectivecrossing import terminated_configs, truncated_configs  # noqa: F401
    return cc


def build_ref_config(cfg: dict):
    """dict (our fixture format) -> reference CollectiveCrossingConfig."""
    from collectivecrossing.configs import CollectiveCrossingConfig
    from collectivecrossing.reward_configs import get_reward_config
    from collectivecrossing.terminated_configs import get_terminated_config
    from collectivecrossing.truncated_configs import get_truncated_config

    kw = {k: v for k, v in cfg.items() if not k.endswith("_config") and not k.startswith("_")}
    rc = dict(cfg.get("reward_config", {"reward_function": "default"}))
    tc = dict(cfg.get("terminated_config", {"terminated_function": "individual_at_destination"}))
    uc = dict(cfg.get("truncated_config", {"truncated_function": "max_steps"}))
    from collectivecrossing.reward_configs import REWARD_CONFIGS, CustomRewardConfig
    from collectivecrossing.terminated_configs import TERMINATED_CONFIGS, CustomTerminatedConfig
    # (a name outside the config registry = a user-registered strategy class: the reference's Custom*Config carries the name)
    kw["reward_config"] = (get_reward_config(rc.pop("reward_function"), **rc) if rc["reward_function"] in REWARD_CONFIGS
                           else CustomRewardConfig(**rc))
    kw["terminated_config"] = (get_terminated_config(tc.pop("terminated_function"), **tc)
                               if tc["terminated_function"] in TERMINATED_CONFIGS else CustomTerminatedConfig(**tc))
    kw["truncated_config"] = get_truncated_config(uc.pop("truncated_function"), **uc)
    if cfg.get("_relaxed"):
        # reference-illegal agent counts (BASELINE config 5): bypass validation, SURVEY 8c
        from collectivecrossing.observation_configs import DefaultObservationConfig
        kw.setdefault("observation_config", DefaultObservationConfig())
        kw.setdefault("render_mode", None)
        return CollectiveCrossingConfig.model_construct(**kw)
    return CollectiveCrossingConfig(**kw)


def ids_of(cfg: dict) -> list[str]:
    return ([f"boarding_{i}" for i in range(cfg["num_boarding_agents"])] +
            [f"exiting_{j}" for j in range(cfg["num_exiting_agents"])])


class Recorder:
    """Drives E reference envs for K steps and re-encodes everything as arrays."""

    def __init__(self, cfg: dict, E: int, K: int):
        from collectivecrossing import CollectiveCrossingEnv
        self.cfg, self.E, self.K = cfg, E, K
        self.ids = ids_of(cfg)
        self.N = N = len(self.ids)
        self.L = 6 + 4 * N
        self.envs = [CollectiveCrossingEnv(config=build_ref_config(cfg)) for _ in range(E)]
        z = lambda *s, dt=np.uint8: np.zeros(s, dt)  # noqa: E731
        self.a = dict(
            init_x=z(E, N, dt=np.int32), init_y=z(E, N, dt=np.int32), init_active=z(E, N),
            init_terminated=z(E, N), init_truncated=z(E, N), init_step_count=z(E, dt=np.int32),
            actions=np.full((K, E, N), ABSENT, np.uint8), order=z(K, E, N),
            x=z(K, E, N, dt=np.int32), y=z(K, E, N, dt=np.int32), active=z(K, E, N),
            terminated=z(K, E, N), truncated=z(K, E, N), step_count=z(K, E, dt=np.int32),
            reward=z(K, E, N, dt=np.float64), agent_flags=z(K, E, N), env_flags=z(K, E),
            obs=z(K, E, N, self.L, dt=np.float32))

    def snapshot_init(self, e: int) -> None:
        env = self.envs[e]
        for i, aid in enumerate(self.ids):
            ag = env._agents[aid]
            self.a["init_x"][e, i], self.a["init_y"][e, i] = int(ag.position[0]), int(ag.position[1])
            self.a["init_active"][e, i] = ag.active
            self.a["init_terminated"][e, i] = ag.terminated
            self.a["init_truncated"][e, i] = ag.truncated
        self.a["init_step_count"][e] = env._step_count

    def step(self, s: int, e: int, action_dict: dict[str, int]) -> tuple[bool, bool]:
        env, ids, a = self.envs[e], self.ids, self.a
        slot = {aid: i for i, aid in enumerate(ids)}
        # move order = dict iteration order; agents not in the dict fill the tail (they don't move)
        listed = [slot[k] for k in action_dict]
        a["order"][s, e] = listed + [i for i in range(self.N) if i not in listed]
        for k, v in action_dict.items():
            a["actions"][s, e, slot[k]] = v
        obs, rew, term, trunc, infos = env.step(dict(action_dict))
        for i, aid in enumerate(ids):
            ag = env._agents[aid]
            a["x"][s, e, i], a["y"][s, e, i] = int(ag.position[0]), int(ag.position[1])
            a["active"][s, e, i] = ag.active
            a["terminated"][s, e, i] = ag.terminated
            a["truncated"][s, e, i] = ag.truncated
            f = 0
            assert aid in term, "terminateds lists every agent every step"
            f |= AF["TERMINATED"] if term[aid] else 0
            assert (aid in rew) == (aid in trunc)
            if aid in rew:
                f |= AF["LIVE"]
                a["reward"][s, e, i] = float(rew[aid])
                f |= AF["TRUNCATED"] if trunc[aid] else 0
            assert (aid in obs) == (aid in infos)
            if aid in obs:
                f |= AF["OBS"]
                info = infos[aid]
                assert info["in_tram_area"] == env.is_in_tram_area(aid)
                assert info["active"] == ag.active
            f |= AF["IN_TRAM"] if env.is_in_tram_area(aid) else 0
            f |= AF["AT_DOOR"] if env.is_at_tram_door(aid) else 0
            f |= AF["ACTIVE"] if ag.active else 0
            f |= AF["AT_DEST"] if env.has_agent_reached_destination(aid) else 0
            a["agent_flags"][s, e, i] = f
            full = env._get_agent_observation(aid)  # the reference's observation function
            if aid in obs:
                assert np.array_equal(full, obs[aid])
            a["obs"][s, e, i] = full
        a["step_count"][s, e] = env._step_count
        at, au = bool(term["__all__"]), bool(trunc["__all__"])
        a["env_flags"][s, e] = (EF["ALL_TERM"] if at else 0) | (EF["ALL_TRUNC"] if au else 0)
        return at, au

    def save(self, name: str, **extra) -> None:
        out = HERE / f"{name}.npz"
        np.savez_compressed(out, config_json=np.array(json.dumps(self.cfg)), **self.a, **extra)
        print(f"wrote {out.name}: E={self.E} K={self.K} N={self.N} {out.stat().st_size / 1024:.0f} KiB")


# ------------------------------------------------------------------------------------ configs
def cfg_c1(**over):
    """BASELINE config 1/2/4 geometry = the reference README quick-start (README.md:48-64)."""
    c = dict(width=12, height=8, division_y=4, tram_door_left=5, tram_door_right=7, tram_length=9,
             num_boarding_agents=5, num_exiting_agents=3, exiting_destination_area_y=0,
             boarding_destination_area_y=8,
             truncated_config=dict(truncated_function="max_steps", max_steps=100))
    c.update(over)
    return c


def cfg_c3(**over):
    """BASELINE config 3: 20x12, 16+16, SimpleDistance."""
    c = dict(width=20, height=12, division_y=6, tram_door_left=6, tram_door_right=10,
             tram_length=16, num_boarding_agents=16, num_exiting_agents=16,
             exiting_destination_area_y=0, boarding_destination_area_y=12,
             reward_config=dict(reward_function="simple_distance", distance_penalty_factor=0.1),
             truncated_config=dict(truncated_function="max_steps", max_steps=100))
    c.update(over)
    return c


def cfg_c5(nb, ne, max_steps=500, **over):
    """BASELINE config 5: 32x16, AllAtDestination, MaxSteps; 32+32 is reference-illegal."""
    c = dict(width=32, height=16, division_y=8, tram_door_left=10, tram_door_right=16,
             tram_length=26, num_boarding_agents=nb, num_exiting_agents=ne,
             exiting_destination_area_y=0, boarding_destination_area_y=16,
             terminated_config=dict(terminated_function="all_at_destination"),
             truncated_config=dict(truncated_function="max_steps", max_steps=max_steps))
    if nb + ne > 50:
        c["_relaxed"] = True
    c.update(over)
    return c


def cfg_big(**over):
    """The largest legal grid (configs.py:39-40: 100 x 100): its per-env occupancy tables exceed the LDS budget of the
    kernels, which then resolve conflicts -- and, round 4, the in-kernel policies' `busy` bits -- through all-pairs compares."""
    c = dict(width=100, height=100, division_y=50, tram_door_left=25, tram_door_right=35, tram_length=60,
             num_boarding_agents=10, num_exiting_agents=10, exiting_destination_area_y=0,
             boarding_destination_area_y=100,
             truncated_config=dict(truncated_function="max_steps", max_steps=400))
    c.update(over)
    return c


def cfg_small(**over):
    """The 10x6 / 2+1 env of the reference's VCR tests (test_trajectory_vcr.py:323-340)."""
    c = dict(width=10, height=6, division_y=3, tram_door_left=3, tram_door_right=5, tram_length=8,
             num_boarding_agents=2, num_exiting_agents=1, exiting_destination_area_y=0,
             boarding_destination_area_y=5,
             truncated_config=dict(truncated_function="max_steps", max_steps=50))
    c.update(over)
    return c


def cfg_fuzz(index: int) -> dict:
    """Random but reference-VALID config number ``index`` (the reference's own validators decide):
    geometry, door width (incl. sealed), agent counts, destination rows, every reward / termination
    strategy with random parameters, short max_steps so truncation happens inside the recording."""
    rng = np.random.default_rng(9000 + index)
    for _ in range(10000):
        W, H = int(rng.integers(4, 27)), int(rng.integers(4, 19))
        div = int(rng.integers(1, H))
        Lt = int(rng.integers(2, W + 1))
        dl = int(rng.integers(0, Lt))
        dr = int(rng.integers(dl + 1, Lt + 1))
        tl, tr = W // 2 - Lt // 2, W // 2 + Lt // 2
        cap_b = W * div - (dr - dl + 1)                       # reset() also rejects cells under the door
        cap_e = max(0, tr - tl - 1) * max(0, H - div - 1)
        nb = int(rng.integers(0, max(1, min(14, cap_b // 3) + 1)))
        ne = int(rng.integers(0, max(1, min(12, cap_e // 3) + 1)))
        if nb + ne < 1:
            continue
        reward = [dict(reward_function="default",
                       boarding_destination_reward=float(np.round(rng.uniform(-20, 20), 3)),
                       tram_door_reward=float(np.round(rng.uniform(-20, 20), 3)),
                       tram_area_reward=float(np.round(rng.uniform(-20, 20), 3)),
                       distance_penalty_factor=float(rng.choice([0.0, 0.1, 0.37, 1.0, 2.5, 7.3]))),
                  dict(reward_function="simple_distance",
                       distance_penalty_factor=float(rng.choice([0.0, 0.1, 0.7, 3.3]))),
                  dict(reward_function="binary", goal_reward=float(np.round(rng.uniform(0, 9), 2)),
                       no_goal_reward=float(np.round(rng.uniform(-3, 3), 2))),
                  dict(reward_function="constant_negative",
                       step_penalty=float(np.round(rng.uniform(-5, 0), 3)))][int(rng.integers(0, 4))]
        cfg = dict(width=W, height=H, division_y=div, tram_door_left=dl, tram_door_right=dr, tram_length=Lt,
                   num_boarding_agents=nb, num_exiting_agents=ne,
                   exiting_destination_area_y=int(rng.integers(0, div)),
                   boarding_destination_area_y=int(rng.integers(div, H + 1)),
                   reward_config=reward,
                   terminated_config=dict(terminated_function=str(rng.choice(
                       ["individual_at_destination", "all_at_destination"]))),
                   truncated_config=dict(truncated_function=str(rng.choice(["max_steps", "custom"])),
                                         max_steps=int(rng.integers(5, 40))))
        try:
            build_ref_config(cfg)
        except Exception:   # noqa: BLE001 -- the reference's pydantic validators reject it: draw again
            continue
        return cfg
    raise RuntimeError("no valid fuzz config found")


# ------------------------------------------------------------------------------------ drivers
def run_random(name, cfg, seeds, K, shuffle=False, p_absent=0.0, act_done=True):
    """reset(seed=s) then K steps of uniform random actions (keeps stepping after __all__)."""
    rec = Recorder(cfg, len(seeds), K)
    for e, seed in enumerate(seeds):
        env = rec.envs[e]
        env.reset(seed=int(seed))
        rec.snapshot_init(e)
        rng = np.random.default_rng(1000 + int(seed))
        for s in range(K):
            ids = list(rec.ids) if act_done else list(env.agents)
            if shuffle:
                ids = [ids[i] for i in rng.permutation(len(ids))]
            acts = {aid: int(rng.integers(0, 5)) for aid in ids if rng.random() >= p_absent}
            rec.step(s, e, acts)
    rec.save(name, seeds=np.asarray(seeds, np.int64))


def run_greedy(name, cfg, seeds, K, policy="greedy", epsilon=0.0):
    """reset(seed) then the reference's GreedyPolicy / WaitingPolicy (epsilon=0 unless given; one
    policy object = one RandomState(42) per env) for live agents in index order."""
    from baseline_policies import GreedyPolicy, WaitingPolicy
    rec = Recorder(cfg, len(seeds), K)
    for e, seed in enumerate(seeds):
        env = rec.envs[e]
        obs, _ = env.reset(seed=int(seed))
        rec.snapshot_init(e)
        pol = (GreedyPolicy if policy == "greedy" else WaitingPolicy)(randomness_factor=epsilon, seed=42)
        for s in range(K):
            acts = {aid: int(pol.get_action(aid, None, env)) for aid in env.agents}
            rec.step(s, e, acts)
    rec.save(name, seeds=np.asarray(seeds, np.int64), **({"epsilon": np.float64(epsilon)} if epsilon else {}))


def run_scenarios(name, cfg, scenarios):
    """Forced initial states (the reference's tests poke env._agents the same way)."""
    K = max(len(sc["steps"]) for sc in scenarios)
    rec = Recorder(cfg, len(scenarios), K)
    for e, sc in enumerate(scenarios):
        env = rec.envs[e]
        env.reset(seed=0)
        for aid, st in sc["state"].items():
            ag = env._agents[aid]
            ag.position = np.array(st["pos"], dtype=np.int32)
            ag.active = bool(st.get("active", True))
            ag.terminated = bool(st.get("terminated", False))
            ag.truncated = bool(st.get("truncated", False))
        env._step_count = int(sc.get("step_count", 0))
        rec.snapshot_init(e)
        steps = list(sc["steps"]) + [{}] * (K - len(sc["steps"]))
        for s, acts in enumerate(steps):
            rec.step(s, e, acts)
    rec.save(name, labels=np.array([sc["label"] for sc in scenarios]))


def run_rollout(name, cfg, E, K, P, seed0, total_envs=None, env_offset=0):
    """Auto-reset rollout: `if __all__: reset(seed=seed0 + (g + episode*stride) % P)` with
    stride = total % P, or 1 when P divides total (every env still walks through the pool)."""
    from collectivecrossing import CollectiveCrossingEnv
    total = total_envs or E
    ids = ids_of(cfg)
    N = len(ids)
    pool = np.zeros((P, N, 2), np.uint8)
    penv = CollectiveCrossingEnv(config=build_ref_config(cfg))
    for pidx in range(P):
        penv.reset(seed=seed0 + pidx)
        for i, aid in enumerate(ids):
            pool[pidx, i] = penv._agents[aid].position
    rec = Recorder(cfg, E, K)
    rng = np.random.default_rng(seed0)
    all_actions = rng.integers(0, 5, size=(K, E, N), dtype=np.uint8)
    episodes = np.zeros(E, np.int32)
    for e in range(E):
        env = rec.envs[e]
        g = env_offset + e
        env.reset(seed=seed0 + (g % P))
        for i, aid in enumerate(ids):
            assert tuple(env._agents[aid].position) == tuple(pool[g % P, i])
        rec.snapshot_init(e)
        for s in range(K):
            acts = {aid: int(all_actions[s, e, i]) for i, aid in enumerate(ids)}
            at, au = rec.step(s, e, acts)
            if at or au:
                episodes[e] += 1
                env.reset(seed=seed0 + int((g + int(episodes[e]) * ((total % P) or 1)) % P))
                rec.a["env_flags"][s, e] |= EF["RESET"]
    rec.save(name, pool_xy=pool, seed0=np.int64(seed0), final_episode=episodes,
             total_envs=np.int64(total), env_offset=np.int64(env_offset))


# --------------------------------------------------------------------------- shim validation
def replay_reference_goldens():
    """The reference's own goldens through the imported reference (validates the stand-ins)."""
    from collectivecrossing import CollectiveCrossingEnv
    for fn in ("golden_basic_trajectory.json", "regression_test.json"):
        d = json.loads((REF / "tests/fixtures/trajectories/golden" / fn).read_text())
        env = CollectiveCrossingEnv(config=build_ref_config(
            {k: v for k, v in d["config"].items() if k != "render_mode"}))
        obs, _ = env.reset(seed=42)
        for k, v in d["initial_observations"].items():
            assert np.array_equal(obs[k], np.asarray(v, np.float32)), (fn, "initial", k)
        for st in d["steps"]:
            obs, rew, term, trunc, _ = env.step(st["active_actions"])
            for k, v in st["next_observations"].items():
                assert np.array_equal(obs[k], np.asarray(v, np.float32)), (fn, st["step"], k)
            for k, v in st["next_rewards"].items():
                assert float(rew[k]) == v, (fn, st["step"], k, rew[k], v)
            assert {k: bool(v) for k, v in term.items()} == st["next_terminated"]
            assert {k: bool(v) for k, v in trunc.items()} == st["next_truncated"]
        print(f"reference golden {fn}: replayed bit-exactly through the imported reference")


def edge_scenarios():
    """G5 edge probes on the 10x8 env used all over the reference's unit tests."""
    cfg = dict(width=10, height=8, division_y=4, tram_door_left=3, tram_door_right=5,
               tram_length=8, num_boarding_agents=2, num_exiting_agents=2,
               exiting_destination_area_y=0, boarding_destination_area_y=8,
               truncated_config=dict(truncated_function="max_steps", max_steps=6))
    # geometry: tram 1..9, door 4..6 -> single passable door cell x=5
    far = {"boarding_0": dict(pos=[0, 0]), "boarding_1": dict(pos=[9, 1]),
           "exiting_0": dict(pos=[2, 6]), "exiting_1": dict(pos=[8, 6])}

    def st(**kw):
        s = {k: dict(v) for k, v in far.items()}
        for k, v in kw.items():
            s[k] = v
        return s

    R, U, Lf, D, Wt = 0, 1, 2, 3, 4
    sc = [
        dict(label="walk_to_x_eq_W", state=st(boarding_0=dict(pos=[8, 2])),
             steps=[{"boarding_0": R}] * 4),
        dict(label="walk_up_to_y_eq_H_through_door", state=st(boarding_0=dict(pos=[5, 3])),
             steps=[{"boarding_0": U}] * 6),
        dict(label="wall_blocks_non_door", state=st(boarding_0=dict(pos=[4, 3]), exiting_0=dict(pos=[6, 5])),
             steps=[{"boarding_0": U, "exiting_0": D}] * 3),
        dict(label="side_walls", state=st(exiting_0=dict(pos=[2, 6]), exiting_1=dict(pos=[8, 5])),
             steps=[{"exiting_0": Lf, "exiting_1": R}] * 2),
        dict(label="forced_at_door_reward", state=st(boarding_0=dict(pos=[3, 4]), boarding_1=dict(pos=[7, 4])),
             steps=[{"boarding_0": Wt, "boarding_1": Wt}, {"boarding_0": U, "boarding_1": D}]),
        dict(label="arrive_exactly_at_max_steps", step_count=5,
             state=st(exiting_0=dict(pos=[5, 1])), steps=[{"exiting_0": D}, {"exiting_0": D}, {}]),
        dict(label="step_after_all_done", step_count=5, state=st(), steps=[{}, {"boarding_0": R}, {}]),
        dict(label="action_for_terminated_agent",
             state=st(exiting_0=dict(pos=[5, 0], active=False, terminated=True)),
             steps=[{"exiting_0": U, "boarding_0": R}, {"exiting_0": U}]),
        dict(label="inactive_agent_does_not_block",
             state=st(exiting_0=dict(pos=[3, 0], active=False, terminated=True), exiting_1=dict(pos=[3, 1])),
             steps=[{"exiting_1": D}, {"exiting_1": Wt}]),
        dict(label="swap_blocked_and_chain_order_forward",
             state=st(boarding_0=dict(pos=[2, 2]), boarding_1=dict(pos=[3, 2])),
             steps=[{"boarding_0": R, "boarding_1": R}, {"boarding_0": R, "boarding_1": Lf}]),
        dict(label="chain_order_reversed",
             state=st(boarding_0=dict(pos=[2, 2]), boarding_1=dict(pos=[3, 2])),
             steps=[{"boarding_1": R, "boarding_0": R}, {"boarding_1": Lf, "boarding_0": R}]),
        dict(label="same_target_first_wins",
             state=st(boarding_0=dict(pos=[2, 2]), boarding_1=dict(pos=[4, 2])),
             steps=[{"boarding_0": R, "boarding_1": Lf}, {"boarding_1": Lf, "boarding_0": R}]),
        dict(label="truncated_agent_still_blocks",
             state=st(boarding_0=dict(pos=[2, 2], truncated=True), boarding_1=dict(pos=[3, 2])),
             steps=[{"boarding_1": Lf}, {"boarding_0": R, "boarding_1": Lf}]),
        dict(label="omitted_agents_do_not_move", state=st(), steps=[{"boarding_1": Lf}, {}]),
        dict(label="out_of_bounds_moves", state=st(boarding_0=dict(pos=[0, 0]), boarding_1=dict(pos=[10, 0])),
             steps=[{"boarding_0": Lf, "boarding_1": R}, {"boarding_0": D, "boarding_1": D}]),
        dict(label="exiting_through_door_and_outside_reward", state=st(exiting_0=dict(pos=[5, 5])),
             steps=[{"exiting_0": D}] * 4),
    ]
    return cfg, sc


def run_custom_strategies():
    """g12_custom_strategies.json.gz: user-registered reward / terminated / truncated classes
    (custom_strategies.py) stepped through the imported reference, every dict recorded as data."""
    import custom_strategies as cs
    from collectivecrossing import CollectiveCrossingEnv, configs, reward_configs, rewards
    from collectivecrossing import terminated_configs, terminateds, truncated_configs, truncateds

    plugins = cs.make(rewards.RewardFunction, terminateds.TerminatedFunction, truncateds.TruncatedFunction)
    rewards.REWARD_FUNCTIONS[cs.NAMES["reward"]] = plugins["reward"]
    terminateds.TERMINATED_FUNCTIONS[cs.NAMES["terminated"]] = plugins["terminated"]
    truncateds.TRUNCATED_FUNCTIONS[cs.NAMES["truncated"]] = plugins["truncated"]
    out = {}
    for mix_name, mix in cs.MIXES.items():
        episodes = []
        for seed in (1200, 1201, 1202):
            env = CollectiveCrossingEnv(config=cs.build_config(configs, reward_configs, terminated_configs,
                                                               truncated_configs, mix))
            obs, _ = env.reset(seed=seed)
            ids = list(env._agents)
            # two agents start one or two cells from their destination row and walk straight to it, so
            # that arrivals (and the reward of the finishing step) happen inside the recording
            forced = {"exiting_0": [3, 1 + seed % 2], "boarding_0": [8, 7 - seed % 2]}
            for a, pos in forced.items():
                env._agents[a].position = np.array(pos)
            rng = np.random.default_rng(seed)
            steps = []
            for k in range(24):
                acting = [a for a in ids if rng.random() > 0.1]
                rng.shuffle(acting)
                acts = {a: int(rng.integers(0, 5)) for a in acting}
                if k < 3:
                    acts.update({a: v for a, v in (("exiting_0", 3), ("boarding_0", 1)) if a in acts})
                o, r, te, tr, inf = env.step(dict(acts))
                steps.append(dict(
                    actions=acts, observations={k: np.asarray(v, np.float32).tolist() for k, v in sorted(o.items())},
                    rewards={k: float(v) for k, v in r.items()}, terminateds={k: bool(v) for k, v in te.items()},
                    truncateds={k: bool(v) for k, v in tr.items()},
                    infos={k: {kk: (vv if isinstance(vv, str) else bool(vv)) for kk, vv in v.items()} for k, v in inf.items()},
                    agents=list(env.agents), step_count=int(env._step_count),
                    flags={a: [bool(env._agents[a].active), bool(env._agents[a].terminated), bool(env._agents[a].truncated)]
                           for a in ids}))
            episodes.append(dict(seed=seed, forced=forced, initial={k: np.asarray(v, np.float32).tolist() for k, v in obs.items()}, steps=steps))
        out[mix_name] = episodes
    import gzip
    f = HERE / "g12_custom_strategies.json.gz"
    with gzip.GzipFile(f, "wb", mtime=0) as z:
        z.write(json.dumps(out).encode())
    print(f"wrote {f.name}: {len(out)} strategy mixes x 3 episodes x 24 steps, {f.stat().st_size / 1024:.0f} KiB")


def run_position_only():
    """g13_position_only_*: user-registered reward / terminated classes that depend on the agent's own type and cell only
    (custom_strategies.make_position_only), registered with the imported reference and recorded in the array contract --
    what collectivecrossing_amd lowers to ccx_set_reward_table / ccx_set_terminated_table."""
    import custom_strategies as cs
    from collectivecrossing import rewards, terminateds

    plugins = cs.make_position_only(rewards.RewardFunction, terminateds.TerminatedFunction)
    rewards.REWARD_FUNCTIONS[cs.PO_NAMES["reward"]] = plugins["reward"]
    terminateds.TERMINATED_FUNCTIONS[cs.PO_NAMES["terminated"]] = plugins["terminated"]
    both = dict(reward_config=dict(reward_function=cs.PO_NAMES["reward"]),
                terminated_config=dict(terminated_function=cs.PO_NAMES["terminated"]),
                truncated_config=dict(truncated_function="max_steps", max_steps=40))
    run_random("g13_position_only_both", cfg_c1(**both), seeds=range(1300, 1308), K=60, shuffle=True, p_absent=0.1)
    run_random("g13_position_only_reward", cfg_c1(reward_config=both["reward_config"], truncated_config=both["truncated_config"]),
               seeds=range(1310, 1314), K=60)
    run_random("g13_position_only_terminated_c3", cfg_c3(terminated_config=both["terminated_config"]), seeds=range(1320, 1322), K=50)


ONLY = [a.split("=", 1)[1] for a in sys.argv[1:] if a.startswith("--only=")]


def _selected(fn):
    """``--only=<prefix>`` regenerates just the fixtures whose name starts with the prefix."""
    def wrapped(name, *a, **kw):
        if ONLY and not any(name.startswith(p) for p in ONLY):
            return
        return fn(name, *a, **kw)
    return wrapped


def main() -> int:
    if not (REF / "src" / "collectivecrossing").is_dir():
        print(f"reference not found at {REF}: nothing to do (fixtures are committed)")
        return 0
    import_reference()
    replay_reference_goldens()
    global run_random, run_greedy, run_scenarios, run_rollout
    run_random, run_greedy, run_scenarios, run_rollout = (_selected(f) for f in (
        run_random, run_greedy, run_scenarios, run_rollout))

    if not ONLY or any("g12_custom_strategies".startswith(p) for p in ONLY):
        run_custom_strategies()
    if not ONLY or any("g13_position_only".startswith(p) or p.startswith("g13") for p in ONLY):
        run_position_only()
    # G14: policies on a grid without LDS occupancy tables (100 x 100)
    run_greedy("g14_greedy_100x100", cfg_big(), seeds=range(1400, 1402), K=230)
    run_greedy("g14_waiting_100x100", cfg_big(num_boarding_agents=6, num_exiting_agents=5), seeds=range(1410, 1411), K=200, policy="waiting")
    # G1 / G2: BASELINE config-1 geometry, random actions, identity and shuffled move order
    run_random("g1_c1_random", cfg_c1(), seeds=range(0, 24), K=110)
    run_random("g2_c1_shuffled_absent", cfg_c1(), seeds=range(100, 116), K=110, shuffle=True,
               p_absent=0.15)
    # G3: dense collisions, SimpleDistance
    run_random("g3_c3_dense_simple_distance", cfg_c3(), seeds=range(200, 204), K=110)
    run_random("g3_c3_dense_shuffled", cfg_c3(), seeds=range(210, 212), K=60, shuffle=True)
    # G4: AllAtDestination, greedy actions; legal 25+25 and reference-illegal 32+32
    run_greedy("g4_c5_all_at_dest_greedy_25_25", cfg_c5(25, 25, max_steps=60), seeds=[300, 301], K=64)
    run_greedy("g4_c5_all_at_dest_greedy_32_32", cfg_c5(32, 32, max_steps=40), seeds=[310], K=44)
    run_greedy("g4_small_all_at_dest_greedy", cfg_small(
        terminated_config=dict(terminated_function="all_at_destination")), seeds=range(320, 328), K=56)
    run_greedy("g4_c1_individual_greedy", cfg_c1(), seeds=range(330, 338), K=110)
    # G5: edge probes on forced states, under several strategy combinations
    ecfg, esc = edge_scenarios()
    run_scenarios("g5_edges_default", ecfg, esc)
    run_scenarios("g5_edges_all_at_dest_binary", dict(
        ecfg, terminated_config=dict(terminated_function="all_at_destination"),
        reward_config=dict(reward_function="binary", goal_reward=3.0, no_goal_reward=-0.25)), esc)
    run_scenarios("g5_edges_constant_negative", dict(
        ecfg, reward_config=dict(reward_function="constant_negative", step_penalty=-1.5)), esc)
    run_scenarios("g5_edges_simple_distance_zero_factor", dict(
        ecfg, reward_config=dict(reward_function="simple_distance", distance_penalty_factor=0.0)), esc)
    # sealed door (door_right - door_left == 1): nobody can ever cross
    run_greedy("g5_sealed_door_greedy", dict(
        width=10, height=8, division_y=4, tram_door_left=4, tram_door_right=5, tram_length=8,
        num_boarding_agents=2, num_exiting_agents=1, exiting_destination_area_y=1,
        boarding_destination_area_y=7,
        truncated_config=dict(truncated_function="max_steps", max_steps=30)), seeds=[340, 341], K=34)
    # G7: shape matrix -- N=1, odd N, padded lane groups (N=5, 12, 50), other rewards
    run_random("g7_n1_boarding_only", cfg_small(num_boarding_agents=1, num_exiting_agents=0), seeds=range(400, 404), K=56)
    run_random("g7_n1_exiting_only", cfg_small(num_boarding_agents=0, num_exiting_agents=1), seeds=range(410, 414), K=56)
    run_random("g7_n3_small", cfg_small(), seeds=range(420, 436), K=56)
    run_random("g7_n5_odd", cfg_c1(num_boarding_agents=3, num_exiting_agents=2), seeds=range(440, 448), K=110)
    run_random("g7_n12_constant_negative", cfg_c1(
        num_boarding_agents=7, num_exiting_agents=5,
        reward_config=dict(reward_function="constant_negative", step_penalty=-0.5)), seeds=range(450, 454), K=110)
    run_random("g7_n50_padded_group", cfg_c5(25, 25, max_steps=40, terminated_config=dict(
        terminated_function="individual_at_destination")), seeds=[460], K=44, shuffle=True)
    run_random("g7_default_reward_big_factor", cfg_c1(reward_config=dict(
        reward_function="default", boarding_destination_reward=-7.25, tram_door_reward=3.5,
        tram_area_reward=0.3, distance_penalty_factor=9.7)), seeds=range(470, 474), K=110)
    run_random("g7_live_agents_only_actions", cfg_c1(), seeds=range(480, 484), K=110, act_done=False)
    # G8: auto-reset rollouts (pool = reference reset placements), incl. a 2-way shard view
    run_rollout("g8_rollout_c1", cfg_c1(truncated_config=dict(truncated_function="max_steps", max_steps=25)),
                E=12, K=90, P=64, seed0=5000)
    run_rollout("g8_rollout_c1_shard1of2", cfg_c1(truncated_config=dict(truncated_function="max_steps", max_steps=25)),
                E=6, K=90, P=64, seed0=5000, total_envs=12, env_offset=6)
    run_rollout("g8_rollout_small_all_at_dest", cfg_small(
        terminated_config=dict(terminated_function="all_at_destination"),
        truncated_config=dict(truncated_function="max_steps", max_steps=12)), E=8, K=60, P=32, seed0=6000)
    # G9: WaitingPolicy (epsilon=0): boarding agents outside the tram wait for the exiting ones
    run_greedy("g9_c1_waiting_policy", cfg_c1(), seeds=range(500, 508), K=110, policy="waiting")
    run_greedy("g9_small_all_at_dest_waiting_policy", cfg_small(
        terminated_config=dict(terminated_function="all_at_destination")), seeds=range(510, 518), K=56,
        policy="waiting")
    run_greedy("g9_c5_waiting_policy_25_25", cfg_c5(25, 25, max_steps=70), seeds=[520], K=74, policy="waiting")
    # G10: random reference-valid configs (geometry x strategies), shuffled dict order, omitted agents
    for i in range(20):
        run_random(f"g10_fuzz_{i:02d}", cfg_fuzz(i), seeds=range(700 + 10 * i, 703 + 10 * i), K=44,
                   shuffle=bool(i & 1), p_absent=0.1 if i % 3 == 0 else 0.0)
    # ... and the same family driven by the reference's greedy / waiting policies (arrivals, door
    # traffic, terminations), max_steps long enough to finish
    for i in range(20, 32):
        cfg = dict(cfg_fuzz(i), truncated_config=dict(truncated_function="max_steps", max_steps=70))
        pol = "waiting" if i % 3 == 2 else "greedy"
        run_greedy(f"g10_fuzz_{pol}_{i:02d}", cfg, seeds=[900 + 2 * i, 901 + 2 * i], K=74, policy=pol)
    # G11: epsilon > 0 -- the policies' RandomState(42) stream decides when and what to randomise
    run_greedy("g11_epsilon_policy_g_c1", cfg_c1(), seeds=range(1100, 1104), K=110, policy="greedy", epsilon=0.3)
    run_greedy("g11_epsilon_policy_w_small", cfg_small(
        terminated_config=dict(terminated_function="all_at_destination")), seeds=range(1110, 1116), K=56,
        policy="waiting", epsilon=0.2)
    run_greedy("g11_epsilon_policy_g_fuzz", dict(cfg_fuzz(25), truncated_config=dict(
        truncated_function="max_steps", max_steps=70)), seeds=[1120, 1121], K=74, policy="greedy", epsilon=0.5)
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
