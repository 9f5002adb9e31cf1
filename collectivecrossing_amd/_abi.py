"""ctypes mirror of ``include/ccx.h`` (the C-ABI of libccx).

Only declarations live here: structures, constants and the prototype table used by
``_lib.load()`` to type-check every exported symbol.  No compute, no fallbacks.
"""

from __future__ import annotations

import ctypes as C

ABI_VERSION = 5

# ccx_status
OK, EINVAL, ENOMEM, EHIP, ENODEVICE = 0, -1, -2, -3, -4

# reward / terminated / truncated modes (include/ccx.h enums)
REWARD_MODES = {"default": 0, "simple_distance": 1, "binary": 2, "constant_negative": 3}
TERMINATED_MODES = {"individual_at_destination": 0, "all_at_destination": 1}
TRUNCATED_MODES = {"max_steps": 0, "custom": 0}  # truncateds.py:64-95: same arithmetic

ACTION_ABSENT = 255
POLICY_GREEDY = 1
POLICY_WAITING = 2
POLICY_RANDOM = 3
POLICIES = {"greedy": POLICY_GREEDY, "waiting": POLICY_WAITING, "random": POLICY_RANDOM}

# agent_flags bits
AF_TERMINATED, AF_TRUNCATED, AF_LIVE, AF_OBS = 0x01, 0x02, 0x04, 0x08
AF_IN_TRAM_AREA, AF_AT_DOOR, AF_ACTIVE, AF_AT_DEST = 0x10, 0x20, 0x40, 0x80
# env_flags bits
EF_ALL_TERMINATED, EF_ALL_TRUNCATED, EF_RESET = 0x01, 0x02, 0x04


class CcxParams(C.Structure):
    _fields_ = [
        ("width", C.c_int32), ("height", C.c_int32), ("division_y", C.c_int32),
        ("tram_left", C.c_int32), ("tram_right", C.c_int32),
        ("door_left", C.c_int32), ("door_right", C.c_int32),
        ("num_boarding", C.c_int32), ("num_exiting", C.c_int32),
        ("boarding_dest_y", C.c_int32), ("exiting_dest_y", C.c_int32),
        ("reward_mode", C.c_int32), ("terminated_mode", C.c_int32), ("truncated_mode", C.c_int32),
        ("max_steps", C.c_int32), ("_pad0", C.c_int32),
        ("boarding_destination_reward", C.c_double), ("tram_door_reward", C.c_double),
        ("tram_area_reward", C.c_double), ("distance_penalty_factor", C.c_double),
        ("goal_reward", C.c_double), ("no_goal_reward", C.c_double), ("step_penalty", C.c_double),
    ]

    @property
    def num_agents(self) -> int:
        return self.num_boarding + self.num_exiting


class CcxState(C.Structure):
    _fields_ = [
        ("x", C.c_void_p), ("y", C.c_void_p), ("active", C.c_void_p), ("terminated", C.c_void_p),
        ("truncated", C.c_void_p), ("step_count", C.c_void_p), ("episode", C.c_void_p),
    ]


class CcxStepOut(C.Structure):
    _fields_ = [("obs", C.c_void_p), ("reward", C.c_void_p), ("agent_flags", C.c_void_p),
                ("env_flags", C.c_void_p), ("obs_compact", C.c_void_p)]


class CcxRolloutOut(C.Structure):
    _fields_ = [("obs", C.c_void_p), ("reward", C.c_void_p), ("agent_flags", C.c_void_p),
                ("env_flags", C.c_void_p), ("obs_compact", C.c_void_p)]


class CcxCounters(C.Structure):
    _fields_ = [("env_steps", C.c_uint64), ("agent_steps", C.c_uint64),
                ("live_agent_steps", C.c_uint64), ("episodes", C.c_uint64),
                ("moves", C.c_uint64), ("arrivals", C.c_uint64)]

    def as_dict(self) -> dict[str, int]:
        return {name: int(getattr(self, name)) for name, _ in self._fields_}


COUNTER_FIELDS = tuple(name for name, _ in CcxCounters._fields_)

_H = C.c_void_p  # ccx_handle*

# symbol -> (restype, argtypes): every function include/ccx.h declares
PROTOTYPES: dict[str, tuple] = {
    "ccx_abi_version": (C.c_int, []),
    "ccx_build_info": (C.c_char_p, []),
    "ccx_last_error": (C.c_char_p, []),
    "ccx_obs_len": (C.c_int32, [C.c_int32]),
    "ccx_create": (C.c_int, [C.POINTER(CcxParams), C.c_int32, C.c_int64, C.c_int64, C.c_int,
                             C.c_void_p, C.POINTER(_H)]),
    "ccx_destroy": (None, [_H]),
    "ccx_num_envs": (C.c_int32, [_H]),
    "ccx_num_agents": (C.c_int32, [_H]),
    "ccx_state_view": (C.c_int, [_H, C.POINTER(CcxState)]),
    "ccx_set_state_host": (C.c_int, [_H, C.POINTER(CcxState)]),
    "ccx_get_state_host": (C.c_int, [_H, C.POINTER(CcxState)]),
    "ccx_set_reset_pool": (C.c_int, [_H, C.c_void_p, C.c_int64]),
    "ccx_reset_from_pool": (C.c_int, [_H, C.c_void_p]),
    "ccx_fill_reset_pool_seeded": (C.c_int, [_H, C.c_void_p, C.c_int64, C.c_uint64]),
    "ccx_reset_seeded": (C.c_int, [_H, C.c_void_p, C.c_void_p]),
    "ccx_greedy_actions": (C.c_int, [_H, C.c_void_p]),
    "ccx_policy_actions": (C.c_int, [_H, C.c_int32, C.c_void_p]),
    "ccx_observe": (C.c_int, [_H, C.c_void_p]),
    "ccx_expand_observations": (C.c_int, [_H, C.c_void_p, C.c_int64, C.c_void_p]),
    "ccx_step": (C.c_int, [_H, C.c_void_p, C.c_void_p, C.POINTER(CcxStepOut)]),
    "ccx_rollout": (C.c_int, [_H, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32,
                              C.POINTER(CcxRolloutOut)]),
    "ccx_rollout_policy": (C.c_int, [_H, C.c_int32, C.c_int32, C.c_int32, C.POINTER(CcxRolloutOut),
                                     C.c_void_p]),
    "ccx_set_check_inputs": (C.c_int, [_H, C.c_int32]),
    "ccx_check_inputs": (C.c_int, [_H]),
    "ccx_set_rng_seed": (C.c_int, [_H, C.c_uint64]),
    "ccx_set_policy_epsilon": (C.c_int, [_H, C.c_double]),
    "ccx_set_policy_stream": (C.c_int, [_H, C.c_int32, C.c_void_p, C.c_uint32]),
    "ccx_get_policy_stream": (C.c_int, [_H, C.POINTER(C.c_int32), C.c_void_p]),
    "ccx_zero_counters": (C.c_int, [_H]),
    "ccx_read_counters": (C.c_int, [_H, C.POINTER(CcxCounters)]),
    "ccx_counters_device_ptr": (C.c_int, [_H, C.POINTER(C.c_void_p)]),
    "ccx_rccl_unique_id": (C.c_int, [C.c_void_p]),
    "ccx_rccl_comm_create": (C.c_int, [C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.POINTER(C.c_void_p)]),
    "ccx_rccl_comm_destroy": (C.c_int, [C.c_void_p]),
    "ccx_rccl_allreduce_counters": (C.c_int, [_H, C.c_void_p, C.c_void_p, C.POINTER(C.c_int32)]),
    "ccx_set_timing": (C.c_int, [_H, C.c_int32]),
    "ccx_last_launch_ms": (C.c_int, [_H, C.POINTER(C.c_float)]),
    "ccx_set_launch_shape": (C.c_int, [_H, C.c_int32, C.c_int32]),
    "ccx_set_writers": (C.c_int, [_H, C.c_int32]),
    "ccx_set_store_throttle": (C.c_int, [_H, C.c_int32]),
    "ccx_set_step_pace": (C.c_int, [_H, C.c_int32]),
    "ccx_get_step_pace": (C.c_int, [_H, C.POINTER(C.c_float)]),
    "ccx_get_pace_state": (C.c_int, [_H, C.POINTER(C.c_float)]),
    "ccx_set_step_pace_start": (C.c_int, [_H, C.c_float]),
    "ccx_set_pace_calibration": (C.c_int, [_H, C.c_int32]),
    "ccx_get_pace_start": (C.c_int, [_H, C.POINTER(C.c_float), C.POINTER(C.c_int32), C.POINTER(C.c_float)]),
    "ccx_set_tunable": (C.c_int, [_H, C.c_char_p, C.c_int32]),
    "ccx_get_residency": (C.c_int, [_H, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "ccx_get_writer_shape": (C.c_int, [_H, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "ccx_get_step_shape": (C.c_int, [_H] + [C.POINTER(C.c_int32)] * 5),
    "ccx_set_reward_table": (C.c_int, [_H, C.c_void_p, C.c_void_p]),
    "ccx_set_terminated_table": (C.c_int, [_H, C.c_void_p, C.c_void_p]),
    "ccx_get_launch_shape": (C.c_int, [_H, C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                                       C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "ccx_host_device_pointer": (C.c_int, [_H, C.c_void_p, C.POINTER(C.c_void_p)]),
    "ccx_set_stream": (C.c_int, [_H, C.c_void_p]),
    "ccx_synchronize": (C.c_int, [_H]),
}
