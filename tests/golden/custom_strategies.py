"""User-style strategy plugins for the g12 fixtures (test tooling, written for this repo).

The reference accepts ANY class registered in its strategy registries (rewards.py:186-216,
terminateds.py:86-114, truncateds.py:99-128).  ``make(RewardBase, TerminatedBase, TruncatedBase)``
builds the same three plugins on top of whichever package's ABCs are passed in, so gen_golden.py
registers them with the imported reference (to record what the reference does with them) and the GPU
test registers them with collectivecrossing_amd (to replay).  They only use the surface both envs
offer: ``env._agents[id]`` (``terminated``, ``truncated``, ``agent_type``), ``env._step_count``,
``env.has_agent_reached_destination``, ``env.is_in_tram_area``.
"""


def make(reward_base, terminated_base, truncated_base):
    class ArrivalBonusReward(reward_base):
        """Written the way the reference's own rewards are (`None` once done): pays a bonus on the step
        the agent finishes -- which only works if rewards are evaluated BEFORE this step's flags."""

        def calculate_reward(self, agent_id, env):
            a = env._agents[agent_id]
            if a.terminated or a.truncated:
                return None
            if env.has_agent_reached_destination(agent_id):
                return 100.0 - env._step_count
            return -0.25 * env._step_count - (1.0 if env.is_in_tram_area(agent_id) else 0.0)

    class TramAreaTerminated(terminated_base):
        """Boarding agents are done as soon as they are inside the tram area, exiting agents on their
        destination row; no entry at all (None) for an agent that is already terminated."""

        def calculate_terminated(self, agent_id, env):
            a = env._agents[agent_id]
            if a.terminated:
                return None
            if a.agent_type.value == "boarding":
                return bool(env.is_in_tram_area(agent_id))
            return bool(env.has_agent_reached_destination(agent_id))

    class PerTypeBudgetTruncated(truncated_base):
        """Boarding agents run out of steps at max_steps, exiting agents 4 steps later."""

        def calculate_truncated(self, agent_id, env):
            a = env._agents[agent_id]
            if a.terminated or a.truncated:
                return None
            budget = self.truncated_config.max_steps + (0 if a.agent_type.value == "boarding" else 4)
            return env._step_count >= budget

    return {"reward": ArrivalBonusReward, "terminated": TramAreaTerminated, "truncated": PerTypeBudgetTruncated}


def make_position_only(reward_base, terminated_base):
    """Two plugins whose value depends on the agent's own type and cell ONLY (g13: what the batch path lowers to tables,
    ccx_set_reward_table / ccx_set_terminated_table).  They say so with ``position_only = True`` -- an attribute the
    reference ignores."""

    class CellValueReward(reward_base):
        position_only = True

        def calculate_reward(self, agent_id, env):
            a = env._agents[agent_id]
            if a.terminated or a.truncated:                      # the built-in convention (rewards.py:64)
                return None
            x, y = int(a.position[0]), int(a.position[1])
            v = 0.125 * x - 0.3 * y + (2.5 if env.is_in_tram_area(agent_id) else 0.0)
            if env.has_agent_reached_destination(agent_id):
                v += 7.0
            return v if a.agent_type.value == "boarding" else -v

    class TramAreaCellTerminated(terminated_base):
        """Boarding agents are done inside the tram area (they stay active and keep blocking), exiting agents on their
        destination row; a value on every step, like the built-in strategies."""
        position_only = True

        def calculate_terminated(self, agent_id, env):
            a = env._agents[agent_id]
            if a.agent_type.value == "boarding":
                return bool(env.is_in_tram_area(agent_id))
            return bool(env.has_agent_reached_destination(agent_id))

    return {"reward": CellValueReward, "terminated": TramAreaCellTerminated}


PO_NAMES = {"reward": "cell_value", "terminated": "tram_area_cell"}


# the four recorded mixes: which of the three strategies is the user's (the rest stay built-in)
MIXES = {"all": ("reward", "terminated", "truncated"), "reward": ("reward",), "terminated": ("terminated",),
         "truncated": ("truncated",)}
NAMES = {"reward": "arrival_bonus", "terminated": "tram_area", "truncated": "per_type_budget"}


def build_config(cfg_mod, reward_cfg_mod, term_cfg_mod, trunc_cfg_mod, mix, max_steps=9):
    """The C1 geometry (reference README quick-start) with the mix's strategies swapped in; works with
    the reference's config modules and with collectivecrossing_amd.configs alike."""
    kw = dict(width=12, height=8, division_y=4, tram_door_left=5, tram_door_right=7, tram_length=9,
              num_boarding_agents=5, num_exiting_agents=3, exiting_destination_area_y=0,
              boarding_destination_area_y=8)
    kw["truncated_config"] = (trunc_cfg_mod.CustomTruncatedConfig(truncated_function=NAMES["truncated"], max_steps=max_steps)
                              if "truncated" in mix else trunc_cfg_mod.MaxStepsTruncatedConfig(max_steps=max_steps + 6))
    if "terminated" in mix:
        kw["terminated_config"] = term_cfg_mod.CustomTerminatedConfig(terminated_function=NAMES["terminated"])
    if "reward" in mix:
        kw["reward_config"] = reward_cfg_mod.CustomRewardConfig(reward_function=NAMES["reward"])
    return cfg_mod.CollectiveCrossingConfig(**kw)
