"""numpy front-end of the CPU oracle (``oracle/ccx_oracle.c``).

TEST INFRASTRUCTURE ONLY -- see the header of ``ccx_oracle.c``.  Imported by ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg, never by the product package.
The array contract is the one of ``include/ccx.h`` (env-major SoA), so a test can hand the same
numpy arrays to the oracle and (through torch) to libccx and compare bit for bit.
"""

from __future__ import annotations

import ctypes as C
import subprocess
from pathlib import Path

import numpy as np

from collectivecrossing_amd._abi import CcxCounters, CcxParams

_HERE = Path(__file__).resolve().parent
_LIB_PATH = _HERE / "_build" / "libccx_oracle.so"
_lib = None


def build(force: bool = False) -> Path:
    """Compile the oracle with gcc (seconds)."""
    src = _HERE / "ccx_oracle.c"
    hdr = _HERE.parent / "include" / "ccx.h"
    stale = (not _LIB_PATH.exists()) or _LIB_PATH.stat().st_mtime < max(
        src.stat().st_mtime, hdr.stat().st_mtime)
    if force or stale:
        subprocess.run(["make", "-C", str(_HERE), "-s", "-B"], check=True)
    return _LIB_PATH


def _p(a: np.ndarray | None, dtype) -> C.c_void_p:
    if a is None:
        return C.c_void_p(None)
    assert a.dtype == dtype and a.flags.c_contiguous, (a.dtype, dtype, a.flags)
    return C.c_void_p(a.ctypes.data)


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(str(_LIB_PATH))
        PP = C.POINTER(CcxParams)
        V = C.c_void_p
        L.ccxo_tram_boundaries.argtypes = [C.c_int32] * 4 + [C.POINTER(C.c_int32 * 4)]
        L.ccxo_tram_boundaries.restype = None
        L.ccxo_obs_len.argtypes = [C.c_int32]
        L.ccxo_obs_len.restype = C.c_int32
        L.ccxo_is_valid_position.argtypes = [PP, C.c_int32, C.c_int32]
        L.ccxo_is_valid_position.restype = C.c_int
        L.ccxo_would_hit_tram_wall.argtypes = [PP, C.c_int32, C.c_int32]
        L.ccxo_would_hit_tram_wall.restype = C.c_int
        L.ccxo_reward.argtypes = [PP, C.c_int32, C.c_int32, C.c_int32]
        L.ccxo_reward.restype = C.c_double
        L.ccxo_observe.argtypes = [PP, C.c_int32, V, V, V, V]
        L.ccxo_observe.restype = None
        L.ccxo_reset_from_pool.argtypes = [PP, C.c_int32, C.c_int64, C.c_int64] + [V] * 8 + [V, C.c_int64]
        L.ccxo_reset_from_pool.restype = None
        L.ccxo_step.argtypes = [PP, C.c_int32] + [V] * 6 + [V, V] + [V] * 4 + [C.POINTER(CcxCounters)]
        L.ccxo_step.restype = None
        L.ccxo_rollout.argtypes = ([PP, C.c_int32, C.c_int64, C.c_int64] + [V] * 7 +
                                   [C.c_int32, V, V, C.c_int32, V, C.c_int64] + [V] * 4 +
                                   [C.POINTER(CcxCounters)])
        L.ccxo_rollout.restype = None
        L.ccxo_rng_probe.argtypes = [C.c_uint64, C.c_int, V, V, C.c_int64, C.c_int64, V]
        L.ccxo_rng_probe.restype = None
        L.ccxo_seeded_placements.argtypes = [PP, C.c_int32, V, V, C.c_int32]
        L.ccxo_seeded_placements.restype = C.c_int
        L.ccxo_rollout_policy.argtypes = ([PP, C.c_int32, C.c_int32, C.c_int64, C.c_int64] + [V] * 7 +
                                          [C.c_int32, V, C.c_int32, V, C.c_int64] + [V] * 4 +
                                          [C.POINTER(CcxCounters)])
        L.ccxo_rollout_policy.restype = None
        L.ccxo_policy_actions.argtypes = [PP, C.c_int32, C.c_int32] + [V] * 6
        L.ccxo_policy_actions.restype = None
        L.ccxo_policy_actions_eps.argtypes = [PP, C.c_int32, C.c_int32, C.c_int64] + [V] * 8
        L.ccxo_policy_actions_eps.restype = None
        L.ccxo_set_user_tables.argtypes = [V, V, V, V]
        L.ccxo_set_user_tables.restype = None
        L.ccxo_set_rng_seed.argtypes = [C.c_uint64]
        L.ccxo_set_rng_seed.restype = None
        L.ccxo_set_policy_epsilon.argtypes = [C.c_double]
        L.ccxo_set_policy_epsilon.restype = None
        L.ccxo_greedy_actions.argtypes = [PP, C.c_int32] + [V] * 6
        L.ccxo_greedy_actions.restype = None
        L.ccxo_mt_seed_streams.argtypes = [V, C.c_int32, V]
        L.ccxo_mt_seed_streams.restype = None
        L.ccxo_mt_probe.argtypes = [C.c_uint32, C.c_int32, V, V, V, C.c_int32, V]
        L.ccxo_mt_probe.restype = None
        L.ccxo_set_policy_stream_mt19937.argtypes = [V, C.c_double]
        L.ccxo_set_policy_stream_mt19937.restype = None
        _lib = L
    return _lib


def tram_boundaries(width: int, tram_length: int, door_left_rel: int, door_right_rel: int):
    """(tram_left, tram_right, door_left, door_right), utils/geometry.py:20-47."""
    out = (C.c_int32 * 4)()
    lib().ccxo_tram_boundaries(width, tram_length, door_left_rel, door_right_rel, C.byref(out))
    return tuple(int(v) for v in out)


def is_valid_position(params: CcxParams, x: int, y: int) -> bool:
    return bool(lib().ccxo_is_valid_position(C.byref(params), x, y))


def would_hit_tram_wall(params: CcxParams, x: int, y: int) -> bool:
    return bool(lib().ccxo_would_hit_tram_wall(C.byref(params), x, y))


def reward(params: CcxParams, slot: int, x: int, y: int) -> float:
    return float(lib().ccxo_reward(C.byref(params), slot, x, y))


def rng_probe(seed: int, n: int, low: int, high: int):
    """(raw 64-bit draws, raw 32-bit draws, Generator.integers(low, high) draws) after seeding."""
    r64, r32, b = np.zeros(n, np.uint64), np.zeros(n, np.uint32), np.zeros(n, np.int64)
    lib().ccxo_rng_probe(seed, n, r64.ctypes.data, r32.ctypes.data, low, high, b.ctypes.data)
    return r64, r32, b


MT_STATE_WORDS = 625      # ccxo_mt19937: key[624] + pos


def mt_probe(seed: int, n: int, counts):
    """n raw 32-bit words, n `random()` doubles and n `choice` indices (list lengths cycling through `counts`) of the
    oracle's restatement of numpy's RandomState(seed) -- each from a freshly seeded generator."""
    raw, dbl, picks = np.empty(n, np.uint32), np.empty(n, np.float64), np.empty(n, np.int32)
    counts = np.ascontiguousarray(counts, np.int32)
    lib().ccxo_mt_probe(C.c_uint32(int(seed)), n, _p(raw, np.uint32), _p(dbl, np.float64), _p(counts, np.int32),
                        len(counts), _p(picks, np.int32))
    return raw, dbl, picks


def seeded_placements(params: CcxParams, seeds, max_tries: int = 1 << 20) -> np.ndarray:
    """reset(seed=s) placements, u8 [len(seeds), N, 2], from the C restatement of numpy's stream."""
    seeds = np.ascontiguousarray(seeds, np.uint64)
    out = np.zeros((len(seeds), params.num_agents, 2), np.uint8)
    rc = lib().ccxo_seeded_placements(C.byref(params), len(seeds), seeds.ctypes.data, out.ctypes.data, max_tries)
    if rc != 0:
        raise RuntimeError("placement did not converge")
    return out


class OracleBatch:
    """E independent envs stepped by the C oracle; mirrors the libccx handle's array contract."""

    def __init__(self, params: CcxParams, num_envs: int, env_offset: int = 0,
                 total_envs: int | None = None):
        self.params = params
        self.E = int(num_envs)
        self.N = params.num_agents
        self.L = 6 + 4 * self.N
        self.env_offset = int(env_offset)
        self.total_envs = int(total_envs if total_envs is not None else num_envs)
        E, N = self.E, self.N
        self.x = np.zeros((E, N), np.int32)
        self.y = np.zeros((E, N), np.int32)
        self.active = np.ones((E, N), np.uint8)
        self.terminated = np.zeros((E, N), np.uint8)
        self.truncated = np.zeros((E, N), np.uint8)
        self.step_count = np.zeros((E,), np.int32)
        self.episode = np.zeros((E,), np.int32)
        self.pool: np.ndarray | None = None
        self.counters = CcxCounters()
        self._mt: np.ndarray | None = None          # [E, 625] u32: one numpy-RandomState restatement per env
        self._mt_epsilon = 0.0
        self._reward_tables = None                  # (boarding, exiting) f64 [H + 1, W + 1]: a position-only user reward
        self._term_tables = None                    # (boarding, exiting) u8  [H + 1, W + 1]: a position-only user terminated value

    def set_user_tables(self, reward=None, terminated=None) -> None:
        """Tables of position-only user strategies (what ``ccx_set_reward_table`` / ``ccx_set_terminated_table`` install on the
        GPU side); ``None`` = the built-in strategy."""
        shape = (self.params.height + 1, self.params.width + 1)
        self._reward_tables = None if reward is None else tuple(np.ascontiguousarray(t, np.float64).reshape(shape) for t in reward)
        self._term_tables = None if terminated is None else tuple(
            np.ascontiguousarray(np.asarray(t) != 0, np.uint8).reshape(shape) for t in terminated)

    def _bind_tables(self) -> None:
        r, t = self._reward_tables or (None, None), self._term_tables or (None, None)
        lib().ccxo_set_user_tables(_p(r[0], np.float64), _p(r[1], np.float64), _p(t[0], np.uint8), _p(t[1], np.uint8))

    # -- state ------------------------------------------------------------------------------
    def set_state(self, x=None, y=None, active=None, terminated=None, truncated=None,
                  step_count=None, episode=None) -> None:
        for name, val, dt in (("x", x, np.int32), ("y", y, np.int32), ("active", active, np.uint8),
                              ("terminated", terminated, np.uint8),
                              ("truncated", truncated, np.uint8),
                              ("step_count", step_count, np.int32), ("episode", episode, np.int32)):
            if val is not None:
                getattr(self, name)[...] = np.asarray(val, dt).reshape(getattr(self, name).shape)

    def set_reset_pool(self, pool_xy: np.ndarray) -> None:
        pool_xy = np.ascontiguousarray(pool_xy, np.uint8)
        assert pool_xy.ndim == 3 and pool_xy.shape[1:] == (self.N, 2), pool_xy.shape
        self.pool = pool_xy

    def reset_from_pool(self, env_mask: np.ndarray | None = None) -> None:
        assert self.pool is not None
        m = None if env_mask is None else np.ascontiguousarray(env_mask, np.uint8)
        lib().ccxo_reset_from_pool(
            C.byref(self.params), self.E, self.env_offset, self.total_envs,
            _p(self.x, np.int32), _p(self.y, np.int32), _p(self.active, np.uint8),
            _p(self.terminated, np.uint8), _p(self.truncated, np.uint8),
            _p(self.step_count, np.int32), _p(self.episode, np.int32), _p(m, np.uint8),
            _p(self.pool, np.uint8), len(self.pool))

    # -- compute ----------------------------------------------------------------------------
    def observe(self) -> np.ndarray:
        obs = np.empty((self.E, self.N, self.L), np.float32)
        lib().ccxo_observe(C.byref(self.params), self.E, _p(self.x, np.int32), _p(self.y, np.int32),
                           _p(self.active, np.uint8), _p(obs, np.float32))
        return obs

    POLICIES = {"greedy": 1, "waiting": 2, "random": 3}

    @staticmethod
    def set_rng_seed(seed: int) -> None:
        """Seed of CCX_POLICY_RANDOM (process-wide in the oracle)."""
        lib().ccxo_set_rng_seed(C.c_uint64(int(seed) & (2**64 - 1)))

    @staticmethod
    def set_policy_epsilon(epsilon: float) -> None:
        """randomness_factor of the greedy / waiting policies in rollout_greedy (process-wide in the oracle)."""
        lib().ccxo_set_policy_epsilon(C.c_double(float(epsilon)))

    def set_policy_stream_mt19937(self, seeds, epsilon: float) -> None:
        """The reference's own epsilon stream: env e draws from `np.random.RandomState(seeds[e])` (one int = every env
        the same seed, as one policy object per env has in the reference), consumed by env.agents in index order.
        `None` switches back to the counter-based draws."""
        if seeds is None:
            self._mt = None
            return
        seeds = np.broadcast_to(np.asarray(seeds, np.uint32), (self.E,)).copy()
        self._mt = np.zeros((self.E, MT_STATE_WORDS), np.uint32)
        lib().ccxo_mt_seed_streams(_p(self._mt, np.uint32), self.E, _p(seeds, np.uint32))
        self._mt_epsilon = float(epsilon)

    def _bind_stream(self) -> None:   # (process-wide pointer in the oracle: set it for every call)
        lib().ccxo_set_policy_stream_mt19937(_p(self._mt, np.uint32), C.c_double(self._mt_epsilon))

    def policy_actions(self, policy: str = "greedy", with_epsilon: bool = False) -> np.ndarray:
        """GreedyPolicy / WaitingPolicy action of every live agent, u8 [E, N]: epsilon 0, or (with_epsilon) with
        the exploration draws of set_policy_epsilon, as ccx_policy_actions does."""
        out = np.empty((self.E, self.N), np.uint8)
        if with_epsilon:
            self._bind_stream()
            lib().ccxo_policy_actions_eps(C.byref(self.params), self.POLICIES[policy], self.E, int(self.env_offset),
                                          _p(self.x, np.int32), _p(self.y, np.int32),
                                          _p(self.active, np.uint8), _p(self.terminated, np.uint8),
                                          _p(self.truncated, np.uint8), _p(self.step_count, np.int32),
                                          _p(self.episode, np.int32), _p(out, np.uint8))
            return out
        lib().ccxo_policy_actions(C.byref(self.params), self.POLICIES[policy], self.E,
                                  _p(self.x, np.int32), _p(self.y, np.int32),
                                  _p(self.active, np.uint8), _p(self.terminated, np.uint8),
                                  _p(self.truncated, np.uint8), _p(out, np.uint8))
        return out

    def greedy_actions(self) -> np.ndarray:
        return self.policy_actions("greedy")

    def step(self, actions: np.ndarray, order: np.ndarray | None = None, want_obs: bool = True):
        E, N = self.E, self.N
        actions = np.ascontiguousarray(actions, np.uint8).reshape(E, N)
        order = None if order is None else np.ascontiguousarray(order, np.uint8).reshape(E, N)
        obs = np.empty((E, N, self.L), np.float32) if want_obs else None
        reward_ = np.empty((E, N), np.float64)
        af = np.empty((E, N), np.uint8)
        ef = np.empty((E,), np.uint8)
        self._bind_tables()
        lib().ccxo_step(C.byref(self.params), E, _p(self.x, np.int32), _p(self.y, np.int32),
                        _p(self.active, np.uint8), _p(self.terminated, np.uint8),
                        _p(self.truncated, np.uint8), _p(self.step_count, np.int32),
                        _p(actions, np.uint8), _p(order, np.uint8), _p(obs, np.float32),
                        _p(reward_, np.float64), _p(af, np.uint8), _p(ef, np.uint8),
                        C.byref(self.counters))
        return obs, reward_, af, ef

    def rollout(self, actions: np.ndarray, order: np.ndarray | None = None,
                auto_reset: bool = False, want_obs: bool = True, want_traj: bool = True):
        E, N = self.E, self.N
        actions = np.ascontiguousarray(actions, np.uint8)
        K = actions.shape[0]
        assert actions.shape == (K, E, N), actions.shape
        order = None if order is None else np.ascontiguousarray(order, np.uint8)
        if auto_reset:
            assert self.pool is not None, "auto_reset needs a reset pool"
        obs = np.empty((K, E, N, self.L), np.float32) if (want_obs and want_traj) else None
        reward_ = np.empty((K, E, N), np.float64) if want_traj else None
        af = np.empty((K, E, N), np.uint8) if want_traj else None
        ef = np.empty((K, E), np.uint8) if want_traj else None
        pool = self.pool
        self._bind_tables()
        lib().ccxo_rollout(C.byref(self.params), E, self.env_offset, self.total_envs,
                           _p(self.x, np.int32), _p(self.y, np.int32), _p(self.active, np.uint8),
                           _p(self.terminated, np.uint8), _p(self.truncated, np.uint8),
                           _p(self.step_count, np.int32), _p(self.episode, np.int32), K,
                           _p(actions, np.uint8), _p(order, np.uint8), int(bool(auto_reset)),
                           _p(pool, np.uint8), 0 if pool is None else len(pool),
                           _p(obs, np.float32), _p(reward_, np.float64), _p(af, np.uint8),
                           _p(ef, np.uint8), C.byref(self.counters))
        return obs, reward_, af, ef

    def rollout_greedy(self, num_steps: int, auto_reset: bool = False, want_obs: bool = True,
                       policy: str = "greedy"):
        """K steps of policy -> step with the epsilon-0 greedy / waiting policy; returns (actions,
        obs, reward, agent_flags, env_flags)."""
        E, N, K = self.E, self.N, int(num_steps)
        if auto_reset:
            assert self.pool is not None, "auto_reset needs a reset pool"
        acts = np.empty((K, E, N), np.uint8)
        obs = np.empty((K, E, N, self.L), np.float32) if want_obs else None
        reward_ = np.empty((K, E, N), np.float64)
        af = np.empty((K, E, N), np.uint8)
        ef = np.empty((K, E), np.uint8)
        pool = self.pool
        self._bind_stream()
        self._bind_tables()
        lib().ccxo_rollout_policy(C.byref(self.params), self.POLICIES[policy], E, self.env_offset, self.total_envs,
                                  _p(self.x, np.int32), _p(self.y, np.int32), _p(self.active, np.uint8),
                                  _p(self.terminated, np.uint8), _p(self.truncated, np.uint8),
                                  _p(self.step_count, np.int32), _p(self.episode, np.int32), K,
                                  _p(acts, np.uint8), int(bool(auto_reset)), _p(pool, np.uint8),
                                  0 if pool is None else len(pool), _p(obs, np.float32),
                                  _p(reward_, np.float64), _p(af, np.uint8), _p(ef, np.uint8),
                                  C.byref(self.counters))
        return acts, obs, reward_, af, ef
