"""Seeded initial placement, bit-identical to the reference's ``reset`` (host side).

``CollectiveCrossingEnv.reset`` (collectivecrossing.py:91-150) seeds
``np_random = Generator(PCG64(SeedSequence(seed)))`` through gymnasium and rejection-samples one
agent at a time: boarding agents draw ``x = integers(0, width)``, ``y = integers(0, division_y)``
and are rejected on invalid cells, occupied cells and the row directly under the door
(``door_left <= x <= door_right and y == division_y - 1``, :110-117); exiting agents draw
``x = integers(tram_left, tram_right + 1)``, ``y = integers(division_y, height)`` (:132-140).
The draw ORDER (x then y, boarding first, index order) fixes the stream consumption, so the same
numpy calls in the same order reproduce the reference's positions for every seed.  numpy is a
dependency of both code bases, not reference code.

Placement is the row "next" of the hot-path scope: it runs on the host and feeds
either ``ccx_set_state_host`` or the device reset pool consumed by ``ccx_rollout``.
"""

from __future__ import annotations

import numpy as np

from .params import calculate_tram_boundaries


def make_generator(seed: int | None) -> np.random.Generator:
    """gymnasium.utils.seeding.np_random(seed)."""
    return np.random.Generator(np.random.PCG64(np.random.SeedSequence(seed)))


def _cell_ok(x: int, y: int, W: int, H: int, div: int, tl: int, tr: int, dl: int, dr: int) -> bool:
    """collectivecrossing.py:509-534 (inclusive bounds, strict door / tram interiors)."""
    if not (0 <= x <= W and 0 <= y <= H):
        return False
    if y == div and not (dl < x < dr):
        return False
    if y >= div and not (tl < x < tr):
        return False
    return True


def sample_initial_positions(config, rng: np.random.Generator) -> np.ndarray:
    """One env's placement, ``[N, 2]`` int32 (x, y), consuming ``rng`` like the reference."""
    tb = calculate_tram_boundaries(config)
    W, H, div = config.width, config.height, config.division_y
    tl, tr, dl, dr = tb.tram_left, tb.tram_right, tb.tram_door_left, tb.tram_door_right
    taken: set[tuple[int, int]] = set()
    out = []
    for _ in range(config.num_boarding_agents):
        for _draw in range(_MAX_DRAWS):
            x = int(rng.integers(0, W))
            y = int(rng.integers(0, div))
            if (_cell_ok(x, y, W, H, div, tl, tr, dl, dr) and (x, y) not in taken
                    and not (dl <= x <= dr and y == div - 1)):
                break
        else:
            raise RuntimeError(_NO_ROOM.format("boarding"))
        taken.add((x, y))
        out.append((x, y))
    for _ in range(config.num_exiting_agents):
        for _draw in range(_MAX_DRAWS):
            x = int(rng.integers(tl, tr + 1))
            y = int(rng.integers(div, H))
            if _cell_ok(x, y, W, H, div, tl, tr, dl, dr) and (x, y) not in taken:
                break
        else:
            raise RuntimeError(_NO_ROOM.format("exiting"))
        taken.add((x, y))
        out.append((x, y))
    return np.asarray(out, np.int32).reshape(-1, 2)


# The reference's rejection loop (collectivecrossing.py:100-150) never gives up: a config without a free
# legal cell spins forever.  Same draws here, but a placement that needs more than 2^16 draws for one
# agent -- the bound of the device kernel (ccx_reset.hip) -- is reported instead.
_MAX_DRAWS = 1 << 16
_NO_ROOM = "no free legal cell for a(n) {} agent after 65536 draws: the config leaves no room (the reference would loop forever)"


def seeded_positions(config, seeds) -> np.ndarray:
    """``reset(seed=s)`` placements for many seeds: ``[len(seeds), N, 2]`` int32."""
    return np.stack([sample_initial_positions(config, make_generator(int(s))) for s in seeds])


def build_reset_pool(config, seed0: int, size: int) -> np.ndarray:
    """Reset pool for ``ccx_set_reset_pool``: u8 ``[size, N, 2]`` for seeds seed0..seed0+size-1."""
    return seeded_positions(config, range(seed0, seed0 + size)).astype(np.uint8)
