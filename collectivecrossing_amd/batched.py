"""Array-native batched environment: E independent CollectiveCrossing envs on one MI355X.

Thin Python over the C-ABI of libccx (``include/ccx.h``).  torch is used for plumbing only --
device buffers, the HIP stream, and (in :mod:`.sharding`) the RCCL process group; every compute
call goes through ``ctypes`` into the HIP library.  Layouts are the library's: env-major SoA,
``[E, N]`` per-agent arrays, observations ``[E, N, L]`` with ``L = 6 + 4N``.

The reference has no batched API; the per-env semantics are those of
``CollectiveCrossingEnv.reset/step`` (collectivecrossing.py:91-261).  The dict API of the
reference is layered on top of this class in :mod:`.env`.
"""

from __future__ import annotations

import ctypes as C
import json
import os
from dataclasses import dataclass
from pathlib import Path

import numpy as np
import torch

from . import _abi
from ._lib import check, load
from .configs import CollectiveCrossingConfig
from .params import agent_ids, lower_config, position_only_tables
from .reset import build_reset_pool, seeded_positions


@dataclass
class StepResult:
    """Device tensors of one step (views into buffers owned by the env, overwritten next step)."""

    obs: torch.Tensor | None  # f32 [E, N, L]
    reward: torch.Tensor      # f64 [E, N]
    agent_flags: torch.Tensor  # u8 [E, N]  (_abi.AF_*)
    env_flags: torch.Tensor   # u8 [E]     (_abi.EF_*)
    obs_compact: torch.Tensor | None = None   # f32 [E, N, 4]  (x, y, type, active), CCX_OBS_COMPACT


@dataclass
class RolloutResult:
    obs: torch.Tensor | None   # f32 [K, E, N, L]
    reward: torch.Tensor | None  # f64 [K, E, N]
    agent_flags: torch.Tensor | None  # u8 [K, E, N]
    env_flags: torch.Tensor | None    # u8 [K, E]
    obs_compact: torch.Tensor | None = None   # f32 [K, E, N, 4], CCX_OBS_COMPACT


def _ptr(t: torch.Tensor | None) -> C.c_void_p:
    return C.c_void_p(None if t is None else t.data_ptr())


class BatchedCollectiveCrossing:
    """E envs sharing one config, resident on one GPU for their whole life."""

    def __init__(self, config: CollectiveCrossingConfig, num_envs: int, device: int | str | None = None,
                 env_offset: int = 0, total_envs: int | None = None, check_inputs: bool | None = None):
        self._lib = load()
        self.config = config
        # (position-only user strategies -- strategies.RewardFunction.position_only -- run inside the kernels as tables)
        self.params = lower_config(config, allow_position_only=True)
        self._user_tables = position_only_tables(config)
        self.num_envs = int(num_envs)
        self.num_agents = self.params.num_agents
        self.obs_len = 6 + 4 * self.num_agents
        self.agent_ids = agent_ids(config)
        self.env_offset = int(env_offset)
        self.total_envs = int(total_envs if total_envs is not None else num_envs)
        if not torch.cuda.is_available():
            raise RuntimeError("collectivecrossing_amd needs an MI355X: torch sees no GPU and libccx "
                               "has no CPU path")
        dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        if dev.type != "cuda":
            raise ValueError(f"device must be a GPU, got {dev}")
        self.device = torch.device("cuda", dev.index if dev.index is not None else torch.cuda.current_device())
        self._stream = torch.cuda.current_stream(self.device)
        self._stream_raw = self._stream.cuda_stream
        handle = C.c_void_p()
        check(self._lib.ccx_create(C.byref(self.params), self.num_envs, self.env_offset, self.total_envs,
                                   self.device.index, C.c_void_p(self._stream.cuda_stream),
                                   C.byref(handle)))
        self._h = handle
        self._pool: torch.Tensor | None = None
        self._step_bufs: StepResult | None = None
        self._step_out_cache: dict = {}
        self._rollouts_with_obs = 0
        if check_inputs is None:
            check_inputs = os.environ.get("CCX_CHECK_INPUTS", "0") not in ("", "0")
        if check_inputs:
            self.set_check_inputs(True)
        rew, term = self._user_tables
        if term is not None:
            self.set_terminated_table(*term)
        if rew is not None:
            self.set_reward_table(*rew)
        self._apply_pace_memory()

    # ------------------------------------------------------------------ lifetime
    def close(self) -> None:
        if getattr(self, "_h", None):
            try:
                self._remember_pace()
            except Exception:
                pass
            self._lib.ccx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ helpers
    def _new(self, shape, dtype) -> torch.Tensor:
        return torch.empty(shape, dtype=dtype, device=self.device)

    def _as_dev_u8(self, a, shape) -> torch.Tensor:
        if isinstance(a, torch.Tensor):
            # (fast path first: a step-wise loop hands over a ready device tensor every few microseconds)
            t = a if (a.dtype is torch.uint8 and a.device == self.device) else a.to(device=self.device, dtype=torch.uint8)
        else:
            t = torch.from_numpy(np.ascontiguousarray(a, np.uint8)).to(self.device)
        if tuple(t.shape) != tuple(shape):
            raise ValueError(f"expected shape {tuple(shape)}, got {tuple(t.shape)}")
        if not t.is_contiguous():
            t = t.contiguous()
        self._order_after_current_stream(t)
        return t

    def _order_after_current_stream(self, *tensors) -> None:
        """The library launches on the stream captured at construction (or `use_stream`).  Inputs are
        produced / copied on torch's CURRENT stream: when that is another stream, make the launch stream
        wait for it and tell the caching allocator that the launch stream uses the buffers (otherwise
        nothing orders the H2D copy before the kernel, and a temporary could be recycled while the
        kernel still reads it)."""
        raw = torch._C._cuda_getCurrentRawStream(self.device.index)      # (an int; ~0.2 us, this is the per-step path)
        if raw != self._stream_raw:
            self._stream.wait_stream(torch.cuda.current_stream(self.device))
            for t in tensors:
                if t is not None and t.is_cuda:
                    t.record_stream(self._stream)

    # ------------------------------------------------------------------ state
    def set_state(self, x=None, y=None, active=None, terminated=None, truncated=None,
                  step_count=None, episode=None) -> None:
        """Overwrite (parts of) the SoA state from host arrays (``ccx_set_state_host``)."""
        E, N = self.num_envs, self.num_agents
        keep = []

        def conv(a, dt, shape):
            if a is None:
                return None
            arr = np.ascontiguousarray(np.asarray(a, dt).reshape(shape))
            keep.append(arr)
            return arr.ctypes.data

        st = _abi.CcxState(conv(x, np.int32, (E, N)), conv(y, np.int32, (E, N)),
                           conv(active, np.uint8, (E, N)), conv(terminated, np.uint8, (E, N)),
                           conv(truncated, np.uint8, (E, N)), conv(step_count, np.int32, (E,)),
                           conv(episode, np.int32, (E,)))
        check(self._lib.ccx_set_state_host(self._h, C.byref(st)))

    def get_state(self) -> dict[str, np.ndarray]:
        E, N = self.num_envs, self.num_agents
        out = dict(x=np.empty((E, N), np.int32), y=np.empty((E, N), np.int32),
                   active=np.empty((E, N), np.uint8), terminated=np.empty((E, N), np.uint8),
                   truncated=np.empty((E, N), np.uint8), step_count=np.empty((E,), np.int32),
                   episode=np.empty((E,), np.int32))
        st = _abi.CcxState(*[out[k].ctypes.data for k in
                             ("x", "y", "active", "terminated", "truncated", "step_count", "episode")])
        check(self._lib.ccx_get_state_host(self._h, C.byref(st)))
        return out

    # ------------------------------------------------------------------ reset
    def reset(self, seeds, env_mask=None) -> torch.Tensor:
        """``reset(seed=seeds[e])`` for every (masked) env ON THE DEVICE (``ccx_reset_seeded``:
        numpy's SeedSequence -> PCG64 -> bounded-integer stream and the reference's rejection
        sampling, bit-identical); returns the observations [E,N,L]."""
        if isinstance(seeds, torch.Tensor):
            s = seeds.to(device=self.device).view(-1)
            s = s.view(torch.int64) if s.dtype == torch.uint64 else s.to(torch.int64)
        else:
            s = torch.from_numpy(np.asarray(seeds, dtype=np.uint64).reshape(-1).view(np.int64)).to(self.device)
        if s.numel() != self.num_envs:
            raise ValueError(f"need {self.num_envs} seeds, got {s.numel()}")
        m = None if env_mask is None else self._as_dev_u8(env_mask, (self.num_envs,))
        check(self._lib.ccx_reset_seeded(self._h, _ptr(s.contiguous()), _ptr(m)))
        return self.observe()

    def reset_host(self, seeds) -> torch.Tensor:
        """Same placements computed with numpy on the host (``reset.py``) and uploaded."""
        seeds = np.asarray(seeds).reshape(-1)
        if len(seeds) != self.num_envs:
            raise ValueError(f"need {self.num_envs} seeds, got {len(seeds)}")
        pos = seeded_positions(self.config, seeds)
        E, N = self.num_envs, self.num_agents
        self.set_state(x=pos[..., 0], y=pos[..., 1], active=np.ones((E, N), np.uint8),
                       terminated=np.zeros((E, N), np.uint8), truncated=np.zeros((E, N), np.uint8),
                       step_count=np.zeros(E, np.int32))
        return self.observe()

    def set_reset_pool(self, pool_xy) -> None:
        """Install seeded placements ``u8 [P, N, 2]`` for ``reset_from_pool`` / auto-reset."""
        if isinstance(pool_xy, torch.Tensor):
            t = pool_xy.to(device=self.device, dtype=torch.uint8).contiguous()
        else:
            t = torch.from_numpy(np.ascontiguousarray(pool_xy, np.uint8)).to(self.device)
        if t.ndim != 3 or tuple(t.shape[1:]) != (self.num_agents, 2):
            raise ValueError(f"pool must be [P, {self.num_agents}, 2], got {tuple(t.shape)}")
        self._pool = t
        check(self._lib.ccx_set_reset_pool(self._h, _ptr(t), t.shape[0]))

    def make_reset_pool(self, seed0: int, size: int, on_device: bool = True) -> None:
        """Pool of ``reset(seed=seed0 + p)`` placements, p < size; generated on the GPU by default
        (``ccx_fill_reset_pool_seeded``), or with numpy on the host."""
        if not on_device:
            self.set_reset_pool(build_reset_pool(self.config, seed0, size))
            return
        t = self._new((size, self.num_agents, 2), torch.uint8)
        check(self._lib.ccx_fill_reset_pool_seeded(self._h, _ptr(t), size, seed0))
        self._pool = t
        check(self._lib.ccx_set_reset_pool(self._h, _ptr(t), size))

    def reset_pool(self) -> torch.Tensor | None:
        return self._pool

    def reset_from_pool(self, env_mask=None) -> None:
        m = None if env_mask is None else self._as_dev_u8(env_mask, (self.num_envs,))
        check(self._lib.ccx_reset_from_pool(self._h, _ptr(m)))

    # ------------------------------------------------------------------ scripted policy
    def policy_actions(self, policy: str = "greedy", out: torch.Tensor | None = None) -> torch.Tensor:
        """``GreedyPolicy`` / ``WaitingPolicy`` action of every live agent for the current state, u8 [E, N]
        (255 for agents that are terminated or truncated) -- ``ccx_policy_actions``; epsilon 0 unless
        :meth:`set_policy_epsilon` is in effect (then the same draws as inside ``rollout_policy``)."""
        if out is None:
            out = self._new((self.num_envs, self.num_agents), torch.uint8)
        check(self._lib.ccx_policy_actions(self._h, _abi.POLICIES[policy], _ptr(out)))
        return out

    def greedy_actions(self, out: torch.Tensor | None = None) -> torch.Tensor:
        """``GreedyPolicy(epsilon=0)`` for every live agent (``ccx_greedy_actions``; never explores)."""
        if out is None:
            out = self._new((self.num_envs, self.num_agents), torch.uint8)
        check(self._lib.ccx_greedy_actions(self._h, _ptr(out)))
        return out

    # ------------------------------------------------------------------ compute
    def observe(self, out: torch.Tensor | None = None) -> torch.Tensor:
        if out is None:
            out = self._new((self.num_envs, self.num_agents, self.obs_len), torch.float32)
        check(self._lib.ccx_observe(self._h, _ptr(out)))
        return out

    def step(self, actions, order=None, want_obs: bool = True, want_compact: bool = False) -> StepResult:
        E, N = self.num_envs, self.num_agents
        a = self._as_dev_u8(actions, (E, N))
        o = None if order is None else self._as_dev_u8(order, (E, N))
        if self._step_bufs is None:
            self._step_bufs = StepResult(self._new((E, N, self.obs_len), torch.float32),
                                         self._new((E, N), torch.float64),
                                         self._new((E, N), torch.uint8), self._new((E,), torch.uint8))
        b = self._step_bufs
        if want_compact and b.obs_compact is None:
            b.obs_compact = self._new((E, N, 4), torch.float32)
        # (the output buffers are static: their ccx_step_out struct and the result tuple are built once per output
        # selection -- this is the per-step path of a policy-in-the-loop caller, a few microseconds end to end)
        key = (bool(want_obs), bool(want_compact))
        cached = self._step_out_cache.get(key)
        if cached is None:
            so = _abi.CcxStepOut(_ptr(b.obs if want_obs else None).value, _ptr(b.reward).value,
                                 _ptr(b.agent_flags).value, _ptr(b.env_flags).value,
                                 _ptr(b.obs_compact if want_compact else None).value)
            cached = (so, C.byref(so), StepResult(b.obs if want_obs else None, b.reward, b.agent_flags, b.env_flags,
                                                  b.obs_compact if want_compact else None))
            self._step_out_cache[key] = cached
        check(self._lib.ccx_step(self._h, a.data_ptr(), None if o is None else o.data_ptr(), cached[1]))
        return cached[2]

    def rows_alignment(self) -> int:
        """Smallest batch-size multiple for which one env-step's slab of observation rows ([E][N][L] floats) is a whole number
        of 128-byte lines.  A batch that is not such a multiple still gives bit-identical results, but every step's slab then
        starts somewhere else within a line: wide rows (about 40 agents and more) take a per-step layout and lose ~5 %, narrow
        ones keep the region-relative layout and lose 10-30 %."""
        row_bytes, m = self.num_agents * self.obs_len * 4, 1
        while (row_bytes * m) % 128:
            m *= 2
        return m

    def alloc_rollout(self, num_steps: int, want_obs: bool = True, want_compact: bool = False) -> RolloutResult:
        K, E, N = num_steps, self.num_envs, self.num_agents
        if want_obs and K >= 32 and E % self.rows_alignment() and not getattr(self, "_warned_alignment", False):
            import warnings
            self._warned_alignment = True
            warnings.warn(f"{E} envs x {N} agents: a step's observation rows are not a whole number of 128-byte lines; rollouts with "
                          f"rows run faster with a batch size that is a multiple of {self.rows_alignment()} (results are identical)",
                          RuntimeWarning, stacklevel=2)
        return RolloutResult(self._new((K, E, N, self.obs_len), torch.float32) if want_obs else None,
                             self._new((K, E, N), torch.float64), self._new((K, E, N), torch.uint8),
                             self._new((K, E), torch.uint8),
                             self._new((K, E, N, 4), torch.float32) if want_compact else None)

    def expand_observations(self, obs_compact: torch.Tensor, out: torch.Tensor | None = None) -> torch.Tensor:
        """Compact rows ``[..., N, 4]`` -> DefaultObservation rows ``[..., N, L]`` on the device
        (``ccx_expand_observations``: the gather of ``ccx_observe``, bit for bit; observations.py:43-94)."""
        c = obs_compact.to(device=self.device, dtype=torch.float32).contiguous()
        if c.shape[-2:] != (self.num_agents, 4):
            raise ValueError(f"expected [..., {self.num_agents}, 4], got {tuple(c.shape)}")
        rows = int(c.numel() // (self.num_agents * 4))
        if out is None:
            out = self._new((*c.shape[:-1], self.obs_len), torch.float32)
        self._order_after_current_stream(c, out)
        check(self._lib.ccx_expand_observations(self._h, _ptr(c), rows, _ptr(out)))
        return out

    def rollout(self, actions, order=None, auto_reset: bool = False,
                out: RolloutResult | None = None, want_obs: bool = True,
                want_traj: bool = True, want_compact: bool = False) -> RolloutResult | None:
        """K fused steps (``ccx_rollout``); ``actions`` u8 [K, E, N] on the device."""
        K = int(actions.shape[0])
        E, N = self.num_envs, self.num_agents
        a = self._as_dev_u8(actions, (K, E, N))
        o = None if order is None else self._as_dev_u8(order, (K, E, N))
        if out is None and want_traj:
            out = self.alloc_rollout(K, want_obs, want_compact)
        if out is not None:
            ro = _abi.CcxRolloutOut(_ptr(out.obs).value, _ptr(out.reward).value,
                                    _ptr(out.agent_flags).value, _ptr(out.env_flags).value, _ptr(out.obs_compact).value)
            check(self._lib.ccx_rollout(self._h, K, _ptr(a), _ptr(o), int(bool(auto_reset)), C.byref(ro)))
            self._rollouts_with_obs += int(out.obs is not None and K >= 64)
        else:
            check(self._lib.ccx_rollout(self._h, K, _ptr(a), _ptr(o), int(bool(auto_reset)), None))
        return out

    def rollout_greedy(self, num_steps: int, auto_reset: bool = False, out: RolloutResult | None = None,
                       want_obs: bool = True, actions_out: torch.Tensor | None = None,
                       want_actions: bool = True, policy: str = "greedy"):
        """K fused steps driven by an on-device scripted policy ("greedy" or "waiting",
        ``ccx_rollout_policy``); returns ``(RolloutResult, actions u8 [K, E, N])``."""
        K, E, N = int(num_steps), self.num_envs, self.num_agents
        if out is None:
            out = self.alloc_rollout(K, want_obs)
        if actions_out is None and want_actions:
            actions_out = self._new((K, E, N), torch.uint8)
        ro = _abi.CcxRolloutOut(_ptr(out.obs).value, _ptr(out.reward).value,
                                _ptr(out.agent_flags).value, _ptr(out.env_flags).value, _ptr(out.obs_compact).value)
        check(self._lib.ccx_rollout_policy(self._h, K, _abi.POLICIES[policy], int(bool(auto_reset)),
                                           C.byref(ro), _ptr(actions_out)))
        self._rollouts_with_obs += int(out.obs is not None and K >= 64)
        return out, actions_out

    def rollout_policy(self, num_steps: int, policy: str, auto_reset: bool = False, **kw):
        """``ccx_rollout_policy`` under its general name: "greedy", "waiting" or "random" (uniform actions drawn
        on the device, seeded by :meth:`set_rng_seed`)."""
        return self.rollout_greedy(num_steps, auto_reset=auto_reset, policy=policy, **kw)

    def set_rng_seed(self, seed: int) -> None:
        """Seed of the on-device action RNG (``CCX_POLICY_RANDOM``)."""
        check(self._lib.ccx_set_rng_seed(self._h, int(seed) & (2**64 - 1)))

    def set_policy_epsilon(self, epsilon: float) -> None:
        """``randomness_factor`` of the on-device greedy / waiting policies (``ccx_set_policy_epsilon``): with
        probability epsilon an agent takes one of its valid actions uniformly (the reference's
        ``create_greedy_policy(epsilon=0.1)``), drawn from the counter-based RNG seeded by :meth:`set_rng_seed`
        -- not numpy's stream; the host classes in ``baseline_policies`` keep that one.  0 = deterministic."""
        check(self._lib.ccx_set_policy_epsilon(self._h, float(epsilon)))

    def set_policy_stream(self, kind: str = "mt19937", seeds=42) -> None:
        """Where the policies' exploration draws come from (``ccx_set_policy_stream``): ``"counter"`` (default) or
        ``"mt19937"`` -- every env owns numpy's ``RandomState(seeds[e])`` (one int: the same seed for every env, like one
        ``create_greedy_policy()`` object per env, seed 42) and the device walks it exactly as the reference's
        ``get_action`` calls do, so epsilon episodes of the reference replay action for action.  Policy rollouts then
        run policy and step as separate launches (a validation mode, not the fast path)."""
        kinds = {"counter": 0, "mt19937": 1}
        if kind not in kinds:
            raise ValueError(f"unknown epsilon stream {kind!r}; one of {sorted(kinds)}")
        if np.ndim(seeds) == 0:
            check(self._lib.ccx_set_policy_stream(self._h, kinds[kind], None, int(seeds) & 0xFFFFFFFF))
            return
        arr = np.ascontiguousarray(np.asarray(seeds, np.uint32).reshape(self.num_envs))
        check(self._lib.ccx_set_policy_stream(self._h, kinds[kind], arr.ctypes.data, 0))

    def policy_stream_state(self) -> np.ndarray:
        """u32 [E, 625]: every env's MT19937 generator (key[624], pos) as ``RandomState.get_state()`` has it."""
        out = np.empty((self.num_envs, 625), np.uint32)
        check(self._lib.ccx_get_policy_stream(self._h, None, out.ctypes.data))
        return out

    # ------------------------------------------------------------------ counters / timing / shape
    def zero_counters(self) -> None:
        check(self._lib.ccx_zero_counters(self._h))

    def counters(self) -> dict[str, int]:
        c = _abi.CcxCounters()
        check(self._lib.ccx_read_counters(self._h, C.byref(c)))
        return c.as_dict()

    def counters_tensor(self) -> torch.Tensor:
        """Zero-copy int64 view of the 6 device counters (for the RCCL all-reduce)."""
        p = C.c_void_p()
        check(self._lib.ccx_counters_device_ptr(self._h, C.byref(p)))
        return _device_view_i64(p.value, len(_abi.COUNTER_FIELDS), self.device)

    def set_check_inputs(self, enabled: bool = True) -> None:
        """Opt-in validation of action / move-order tensors on the device (``ccx_set_check_inputs``): what
        the reference's ``_check_action_and_agent_validity`` raises on (collectivecrossing.py:685-711).
        Violations surface as :class:`CcxInputError` (a ``ValueError``) from the next ``synchronize()``,
        ``counters()`` or ``check_inputs()``.  Also enabled by ``CCX_CHECK_INPUTS=1``."""
        check(self._lib.ccx_set_check_inputs(self._h, int(bool(enabled))))

    def check_inputs(self) -> None:
        """Synchronise and raise ``CcxInputError`` if a checked launch saw invalid actions / orders."""
        check(self._lib.ccx_check_inputs(self._h))

    def set_tunable(self, name: str, value: int) -> None:
        """Performance experiment knobs (``ccx_set_tunable``); results never depend on them."""
        check(self._lib.ccx_set_tunable(self._h, name.encode(), int(value)))

    def set_step_pace_start(self, ns_per_env_step: float) -> None:
        """Start value of the adaptive pace controller (``ccx_set_step_pace_start``), 0 = library default."""
        check(self._lib.ccx_set_step_pace_start(self._h, float(ns_per_env_step)))

    # -- where the pace controller starts.  By default the library MEASURES it in-process (ccx_set_pace_calibration: a
    # ~2.5 ms write probe of the caller's own trajectory buffer at the first long rollout).  A pace memory across processes
    # is opt-in: only when CCX_PACE_CACHE names a file does a handle start from the pace an earlier handle of the same
    # shape settled at, and only then is the file updated on close() (never from a multi-rank job: N ranks would
    # read-modify-write one file).  Nothing is ever read from or written to $HOME implicitly.  Never affects results.
    PACE_START_SOURCES = {0: "unpaced", 1: "assumed", 2: "caller", 3: "calibration", 4: "fixed"}

    @staticmethod
    def _pace_cache_path() -> Path | None:
        p = os.environ.get("CCX_PACE_CACHE")
        try:
            world = int(os.environ.get("WORLD_SIZE", "1") or "1")
        except ValueError:                     # (an empty or non-numeric value must not break handle construction: ADVICE r3)
            world = 1
        if not p or world > 1:
            return None
        return Path(p)

    def _pace_key(self) -> str:
        s = self.launch_shape()
        c = self.config
        props = torch.cuda.get_device_properties(self.device)
        name = str(getattr(props, "gcnArchName", "") or props.name).split(":")[0]    # "gfx950" (the marketing name varies)
        return (f"{name}|grid{c.width}x{c.height}|E{self.num_envs}|N{self.num_agents}|lanes{s['lanes_per_wave']}"
                f"|tpb{s['waves_per_block']}|w{s['writers_per_tile']}")

    def _apply_pace_memory(self) -> None:
        """(Re)applied whenever the launch shape changes: the key names the shape."""
        self._pace_from_cache = False
        path = self._pace_cache_path()
        if path is None:
            return
        try:
            ns = float(json.loads(path.read_text())[self._pace_key()]["pace_ns"])
        except Exception:
            return
        if ns > 0:
            self.set_step_pace_start(ns * 1.01)      # start a notch on the safe side of the remembered pace
            self._pace_from_cache = True

    def _remember_pace(self) -> None:
        path = self._pace_cache_path()
        if path is None or self._rollouts_with_obs < 40:
            return
        ns = self.step_pace_ns()
        if not ns > 0:
            return
        try:
            data = json.loads(path.read_text())
        except Exception:
            data = {}
        data[self._pace_key()] = {"pace_ns": round(ns, 2), "launches": self._rollouts_with_obs}
        path.parent.mkdir(parents=True, exist_ok=True)
        tmp = path.with_suffix(f".tmp{os.getpid()}")
        tmp.write_text(json.dumps(data, indent=1, sort_keys=True))
        tmp.replace(path)

    def pace_start(self) -> dict:
        """Where the adaptive pace controller started (``ccx_get_pace_start``): ``ns`` per env-step, ``source`` in
        {"unpaced", "assumed", "caller", "calibration", "fixed", "user_cache"} ("user_cache" = a caller's value that came
        from the CCX_PACE_CACHE file), ``probe_GBs`` = the write rate the calibration probe measured (0 = it has not run)."""
        ns, src, gbs = C.c_float(), C.c_int32(), C.c_float()
        check(self._lib.ccx_get_pace_start(self._h, C.byref(ns), C.byref(src), C.byref(gbs)))
        source = self.PACE_START_SOURCES.get(int(src.value), str(src.value))
        if source == "caller" and getattr(self, "_pace_from_cache", False):
            source = "user_cache"
        return {"ns": float(ns.value), "source": source, "probe_GBs": float(gbs.value)}

    def set_pace_calibration(self, enabled: bool = True) -> None:
        """In-process start-up calibration of the pace controller (``ccx_set_pace_calibration``; default on)."""
        check(self._lib.ccx_set_pace_calibration(self._h, int(bool(enabled))))

    def set_timing(self, enabled: bool = True) -> None:
        """Record HIP events around every launch so that ``last_launch_ms`` works (off by default)."""
        check(self._lib.ccx_set_timing(self._h, int(bool(enabled))))

    def last_launch_ms(self) -> float:
        ms = C.c_float()
        check(self._lib.ccx_last_launch_ms(self._h, C.byref(ms)))
        return float(ms.value)

    def set_launch_shape(self, lanes_per_wave: int = 0, waves_per_block: int = 0) -> None:
        check(self._lib.ccx_set_launch_shape(self._h, lanes_per_wave, waves_per_block))
        self._shape_changed()

    def set_writers(self, writers_per_tile: int = 0) -> None:
        check(self._lib.ccx_set_writers(self._h, writers_per_tile))
        self._shape_changed()

    def _shape_changed(self) -> None:
        """A remembered start value belongs to ONE launch shape: drop it, look the new shape up."""
        if getattr(self, "_pace_from_cache", False):
            self.set_step_pace_start(0.0)
        self._apply_pace_memory()

    def set_store_throttle(self, max_stores_in_flight: int = 0) -> None:
        check(self._lib.ccx_set_store_throttle(self._h, max_stores_in_flight))

    def set_step_pace(self, ns_per_env_step: int = 0) -> None:
        """0 = adaptive (default), -1 = off, > 0 = fixed nanoseconds per env-step (``ccx_set_step_pace``)."""
        check(self._lib.ccx_set_step_pace(self._h, int(ns_per_env_step)))

    def step_pace_ns(self) -> float:
        ns = C.c_float()
        check(self._lib.ccx_get_step_pace(self._h, C.byref(ns)))
        return float(ns.value)

    def pace_state(self) -> dict[str, float]:
        """The pace controller's state (``ccx_get_pace_state``): next pace / floor in ns, launches since the
        last collapse, whether rollouts of this shape are paced at all."""
        v = (C.c_float * 6)()
        check(self._lib.ccx_get_pace_state(self._h, v))
        return dict(zip(("next_pace_ns", "floor_ns", "calm_launches", "paced", "cliff_ns", "cliff_confirmations"),
                        (float(x) for x in v)))

    def launch_shape(self) -> dict[str, int]:
        v = [C.c_int32() for _ in range(4)]
        check(self._lib.ccx_get_launch_shape(self._h, *[C.byref(x) for x in v]))
        w = [C.c_int32() for _ in range(2)]
        check(self._lib.ccx_get_writer_shape(self._h, *[C.byref(x) for x in w]))
        r = [C.c_int32() for _ in range(2)]
        check(self._lib.ccx_get_residency(self._h, *[C.byref(x) for x in r]))
        return dict(zip(("lanes_per_wave", "waves_per_block", "group_lanes", "num_blocks",
                         "writers_per_tile", "store_throttle", "resident_blocks"),
                        (int(x.value) for x in v + w + r[:1])))

    def set_reward_table(self, boarding, exiting) -> None:
        """Per-(agent type, cell) rewards, f64 ``[height + 1, width + 1]`` each (``ccx_set_reward_table``): what a
        position-only user ``RewardFunction`` (rewards.py:16-38) is lowered to.  ``None, None`` restores the built-in reward."""
        if boarding is None and exiting is None:
            check(self._lib.ccx_set_reward_table(self._h, None, None))
        else:
            shape = (self.config.height + 1, self.config.width + 1)
            b, e = (np.ascontiguousarray(t, np.float64) for t in (boarding, exiting))
            if b.shape != shape or e.shape != shape:
                raise ValueError(f"reward tables must be {shape} (rows y, columns x), got {b.shape} / {e.shape}")
            check(self._lib.ccx_set_reward_table(self._h, b.ctypes.data, e.ctypes.data))
        self._shape_changed()

    def set_terminated_table(self, boarding, exiting) -> None:
        """Per-(agent type, cell) ``terminateds[id]`` values, u8 ``[height + 1, width + 1]`` each (``ccx_set_terminated_table``)."""
        if boarding is None and exiting is None:
            check(self._lib.ccx_set_terminated_table(self._h, None, None))
            return
        shape = (self.config.height + 1, self.config.width + 1)
        b, e = (np.ascontiguousarray(np.asarray(t) != 0, np.uint8) for t in (boarding, exiting))
        if b.shape != shape or e.shape != shape:
            raise ValueError(f"terminated tables must be {shape} (rows y, columns x), got {b.shape} / {e.shape}")
        check(self._lib.ccx_set_terminated_table(self._h, b.ctypes.data, e.ctypes.data))

    def step_shape(self) -> dict[str, int]:
        """Launch shape of the short-launch kernel (``ccx_get_step_shape``): ``ok`` = 1 when ``step`` / rollouts of at most
        16 steps without a move order run the short-launch kernel (csrc/ccx_step.hip: a sim wave + ``row_waves`` row waves per tile)."""
        v = [C.c_int32() for _ in range(5)]
        check(self._lib.ccx_get_step_shape(self._h, *[C.byref(x) for x in v]))
        return dict(zip(("ok", "lanes_per_wave", "row_waves", "num_blocks", "lds_bytes"), (int(x.value) for x in v)))

    def use_stream(self, stream: "torch.cuda.Stream | None" = None) -> None:
        """Launch on ``stream`` (default: torch's current stream of the device) from now on."""
        self._stream = stream if stream is not None else torch.cuda.current_stream(self.device)
        self._stream_raw = self._stream.cuda_stream
        check(self._lib.ccx_set_stream(self._h, C.c_void_p(self._stream_raw)))

    def synchronize(self) -> None:
        check(self._lib.ccx_synchronize(self._h))


class _CudaArrayView:
    """Minimal ``__cuda_array_interface__`` carrier so torch can wrap library-owned memory."""

    def __init__(self, ptr: int, n: int):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<i8", "data": (ptr, False),
                                         "version": 3, "strides": None}


def _device_view_i64(ptr: int, n: int, device: torch.device) -> torch.Tensor:
    return torch.as_tensor(_CudaArrayView(ptr, n), device=device)
