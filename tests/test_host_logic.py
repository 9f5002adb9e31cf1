"""Host-side logic that needs no GPU: config validation, lowering, seeded placement, the
dict <-> array codecs of the drop-in env (fed with oracle-produced arrays), sharding arithmetic."""

import json

import numpy as np
import pytest
from _fixtures import GOLDEN, Golden, config_from_dict

from collectivecrossing_amd import configs as C
from collectivecrossing_amd.params import agent_ids, calculate_tram_boundaries, lower_config
from collectivecrossing_amd.reset import build_reset_pool, make_generator, sample_initial_positions, seeded_positions


def _base(**kw):
    d = dict(width=12, height=8, division_y=4, tram_door_left=5, tram_door_right=7, tram_length=9,
             num_boarding_agents=5, num_exiting_agents=3, exiting_destination_area_y=0,
             boarding_destination_area_y=8)
    d.update(kw)
    return d


# ---- configs (reference configs.py:39-195, utils/pydantic.py) ----------------------------------
def test_config_defaults_and_strictness():
    c = C.CollectiveCrossingConfig(**_base())
    assert c.reward_config.get_reward_function_name() == "default"
    assert c.terminated_config.get_terminated_function_name() == "individual_at_destination"
    assert c.truncated_config.max_steps == 1000 and c.render_mode is None
    assert c.reward_config.boarding_destination_reward == 15.0 and c.reward_config.distance_penalty_factor == 0.1
    with pytest.raises(Exception):
        c.width = 3                                    # frozen
    with pytest.raises(Exception):
        C.CollectiveCrossingConfig(**_base(bogus=1))   # extra="forbid"
    assert c.is_valid() and c.get_validation_errors() == []


@pytest.mark.parametrize("over,needle", [
    (dict(tram_length=13), "Tram length"), (dict(tram_door_left=9), "Tram door left boundary"),
    (dict(tram_door_left=7, tram_door_right=5), "cannot be greater"),
    (dict(exiting_destination_area_y=4), "Exiting destination"), (dict(boarding_destination_area_y=3), "Boarding destination"),
    (dict(division_y=8, boarding_destination_area_y=8), "Division line"),
    (dict(num_boarding_agents=20, num_exiting_agents=10), "exceeds reasonable limit"),
    (dict(render_mode="ascii"), "Invalid render_mode"), (dict(width=0), "width"), (dict(height=101), "height"),
    (dict(num_boarding_agents=101), "num_boarding_agents")])
def test_config_validation_rules(over, needle):
    with pytest.raises(Exception) as ei:
        C.CollectiveCrossingConfig(**_base(**over))
    assert needle in str(ei.value)


def test_strategy_config_factories_and_bounds():
    assert C.get_reward_config("simple_distance", distance_penalty_factor=0.2).distance_penalty_factor == 0.2
    assert C.get_truncated_config("max_steps", max_steps=7).max_steps == 7
    assert C.get_terminated_config("all_at_destination").get_terminated_function_name() == "all_at_destination"
    for fn, kind in ((C.get_reward_config, "reward"), (C.get_terminated_config, "termination"),
                     (C.get_truncated_config, "truncation"), (C.get_observation_config, "observation")):
        with pytest.raises(ValueError, match=f"Unknown {kind} function"):
            fn("nope")
    with pytest.raises(Exception):
        C.DefaultRewardConfig(distance_penalty_factor=11.0)
    with pytest.raises(Exception):
        C.MaxStepsTruncatedConfig(max_steps=0)
    with pytest.raises(Exception):
        C.ConstantNegativeRewardConfig(step_penalty=0.5)


def test_lowering_matches_oracle_geometry(oracle):
    for cfg in (_base(), _base(width=10, height=6, division_y=3, tram_door_left=3, tram_door_right=5, tram_length=8,
                               num_boarding_agents=2, num_exiting_agents=1, boarding_destination_area_y=5)):
        c = C.CollectiveCrossingConfig(**cfg)
        p = lower_config(c)
        tb = calculate_tram_boundaries(c)
        assert (p.tram_left, p.tram_right, p.door_left, p.door_right) == \
            oracle.tram_boundaries(c.width, c.tram_length, c.tram_door_left, c.tram_door_right) == \
            (tb.tram_left, tb.tram_right, tb.tram_door_left, tb.tram_door_right)
    custom = C.CollectiveCrossingConfig(**_base(reward_config=C.CustomRewardConfig(reward_function="mine")))
    with pytest.raises(ValueError, match="Unknown reward function 'mine'"):
        lower_config(custom)
    assert agent_ids(c) == ["boarding_0", "boarding_1", "exiting_0"]


# ---- seeded placement (reference reset(), collectivecrossing.py:91-150) --------------------------
def test_seed_42_placement_of_the_reference_goldens():
    d = json.loads((GOLDEN / "reference" / "golden_basic_trajectory.json").read_text())
    cfg = config_from_dict({k: v for k, v in d["config"].items() if k != "render_mode"})
    pos = sample_initial_positions(cfg, make_generator(42))
    assert pos.tolist() == [[0, 2], [6, 1], [4, 5]]       # seed 42 -> (0,2),(6,1),(4,5)
    for i, a in enumerate(agent_ids(cfg)):
        assert d["initial_observations"][a][:2] == pos[i].tolist()


@pytest.mark.parametrize("name", ["g1_c1_random", "g2_c1_shuffled_absent", "g3_c3_dense_simple_distance",
                                  "g4_c5_all_at_dest_greedy_25_25", "g4_c5_all_at_dest_greedy_32_32",
                                  "g7_n1_exiting_only", "g7_n5_odd", "g7_n50_padded_group"])
def test_seeded_placement_matches_reference_for_all_recorded_seeds(name):
    g = Golden(name)
    pos = seeded_positions(g.config, g["seeds"])
    np.testing.assert_array_equal(pos[..., 0], g["init_x"])
    np.testing.assert_array_equal(pos[..., 1], g["init_y"])


def test_reset_pool_matches_reference_pool():
    g = Golden("g8_rollout_c1")
    np.testing.assert_array_equal(build_reset_pool(g.config, int(g["seed0"]), len(g["pool_xy"])), g["pool_xy"])


# ---- dict <-> array codecs of the drop-in env ------------------------------------------------------
def test_encode_actions_and_errors():
    from collectivecrossing_amd.env import encode_actions

    ids = ["boarding_0", "boarding_1", "exiting_0"]
    a, o = encode_actions(ids, {"exiting_0": 3, "boarding_0": 1})
    assert a.tolist() == [1, 255, 3] and o.tolist() == [2, 0, 1]
    with pytest.raises(ValueError) as ei:
        encode_actions(ids, {"boarding_0": 1, "ghost": 0})
    assert "Unknown agent ID" in str(ei.value) and "ghost" in str(ei.value)
    assert "'boarding_0', 'boarding_1', 'exiting_0" in str(ei.value)
    with pytest.raises(ValueError) as ei:
        encode_actions(ids, {"boarding_1": 999})
    assert "Invalid action" in str(ei.value) and "999" in str(ei.value) and "boarding_1" in str(ei.value)
    with pytest.raises(ValueError, match="Unknown agent ID"):      # agent check precedes action check
        encode_actions(ids, {"nobody": 999})
    for bad in (None, "", "boarding 0"):
        with pytest.raises(ValueError, match="Unknown agent ID"):
            encode_actions(ids, {bad: 0})
    for bad in (-1, 5, 10):
        with pytest.raises(ValueError, match="Invalid action"):
            encode_actions(ids, {"boarding_0": bad})


@pytest.mark.parametrize("fn", ["golden_basic_trajectory.json", "regression_test.json"])
def test_decode_step_reproduces_reference_dicts(oracle, fn):
    """encode -> (oracle arrays) -> decode gives exactly the dicts the reference recorded."""
    from collectivecrossing_amd.env import decode_step, encode_actions

    d = json.loads((GOLDEN / "reference" / fn).read_text())
    cfg = config_from_dict({k: v for k, v in d["config"].items() if k != "render_mode"})
    ids = agent_ids(cfg)
    types = ["boarding"] * cfg.num_boarding_agents + ["exiting"] * cfg.num_exiting_agents
    b = oracle.OracleBatch(lower_config(cfg), 1)
    init = np.array([d["initial_observations"][a][:2] for a in ids], np.int32)
    b.set_state(x=init[:, 0], y=init[:, 1])
    for st in d["steps"]:
        acts, order = encode_actions(ids, st["active_actions"])
        obs, rew, af, ef = b.step(acts[None], order[None])
        o, r, te, tr, inf = decode_step(ids, obs[0], rew[0], af[0], int(ef[0]), types)
        assert {k: v.tolist() for k, v in o.items()} == st["next_observations"]
        assert r == st["next_rewards"] and te == st["next_terminated"] and tr == st["next_truncated"]
        assert inf == st["next_infos"]
        assert all(v.dtype == np.float32 for v in o.values())


def test_decode_key_presence_rules(oracle):
    """rewards/truncateds omit done agents, terminateds lists everyone, obs only for live or
    just-finished agents, __all__ rules on empty dicts (collectivecrossing.py:214-259)."""
    from collectivecrossing_amd.env import decode_step

    g = Golden("g5_edges_default")
    e = list(g["labels"]).index("step_after_all_done")
    ids = agent_ids(g.config)
    types = ["boarding"] * 2 + ["exiting"] * 2
    outs = [decode_step(ids, g["obs"][s, e], g["reward"][s, e], g["agent_flags"][s, e], int(g["env_flags"][s, e]), types)
            for s in range(3)]
    o, r, te, tr, inf = outs[0]            # step 6 of max_steps 6: everyone truncates now
    assert set(r) == set(ids) and all(tr[a] for a in ids) and tr["__all__"] and not te["__all__"]
    assert set(o) == set(ids) == set(inf)
    o, r, te, tr, inf = outs[1]            # stepping after __all__: nothing but terminateds
    assert o == {} and r == {} and inf == {} and tr == {"__all__": False}
    assert set(te) == set(ids) | {"__all__"} and not any(te.values())


# ---- sharding arithmetic -----------------------------------------------------------------------------
def test_shard_ranges_cover_the_batch():
    from collectivecrossing_amd.sharding import shard_range

    for total, world in ((32768, 8), (4096, 1), (10, 4), (7, 8)):
        spans = [shard_range(total, world, r) for r in range(world)]
        assert spans[0][0] == 0 and sum(n for _, n in spans) == total
        for (o0, n0), (o1, _) in zip(spans, spans[1:]):
            assert o0 + n0 == o1
    assert shard_range(32768, 8, 3) == (3 * 4096, 4096)
    with pytest.raises(ValueError):
        shard_range(8, 2, 2)


def test_host_placement_gives_up_where_the_reference_would_spin():
    """The only row of this tram is the division row and its door is sealed -- no legal cell for an exiting agent: the reference's
    rejection loop never returns (collectivecrossing.py:127-150); ours reports it after 2^16 draws."""
    import pytest as _pytest

    from collectivecrossing_amd import configs as C
    from collectivecrossing_amd.reset import make_generator, sample_initial_positions

    cfg = C.CollectiveCrossingConfig.model_construct(
        width=8, height=4, division_y=3, tram_door_left=1, tram_door_right=2, tram_length=2,
        num_boarding_agents=1, num_exiting_agents=1, exiting_destination_area_y=0, boarding_destination_area_y=4,
        reward_config=C.DefaultRewardConfig(), terminated_config=C.IndividualAtDestinationTerminatedConfig(),
        truncated_config=C.MaxStepsTruncatedConfig(max_steps=10), observation_config=C.DefaultObservationConfig(),
        render_mode=None)
    with _pytest.raises(RuntimeError, match="no free legal cell"):
        sample_initial_positions(cfg, make_generator(0))


def test_strict_reference_limits_switch_lifts_only_the_total_agent_cap():
    """the reference caps the total at min(w*h//4, 50) agents (configs.py:166); BASELINE configs[4]
    needs 64.  `strict_reference_limits=False` lifts that one cap to the library's 64, nothing else."""
    import pytest
    from pydantic import ValidationError

    from collectivecrossing_amd import configs as C
    from collectivecrossing_amd.params import lower_config
    kw = dict(width=32, height=16, division_y=8, tram_door_left=10, tram_door_right=16, tram_length=26,
              exiting_destination_area_y=0, boarding_destination_area_y=16)
    with pytest.raises(ValidationError, match="exceeds reasonable limit"):
        C.CollectiveCrossingConfig(**kw, num_boarding_agents=32, num_exiting_agents=32)
    cfg = C.CollectiveCrossingConfig(**kw, num_boarding_agents=32, num_exiting_agents=32, strict_reference_limits=False)
    assert lower_config(cfg).num_agents == 64 and cfg.is_valid()
    with pytest.raises(ValidationError, match="exceeds reasonable limit"):
        C.CollectiveCrossingConfig(**kw, num_boarding_agents=33, num_exiting_agents=32, strict_reference_limits=False)
    with pytest.raises(ValidationError, match="Tram length"):       # the other rules still hold
        C.CollectiveCrossingConfig(**dict(kw, tram_length=40), num_boarding_agents=2, num_exiting_agents=2,
                                   strict_reference_limits=False)
    assert C.CollectiveCrossingConfig(**kw, num_boarding_agents=2, num_exiting_agents=2).strict_reference_limits is True
