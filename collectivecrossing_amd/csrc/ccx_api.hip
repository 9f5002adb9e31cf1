// ccx_api.hip -- the C-ABI of libccx (include/ccx.h) on top of the gfx950 kernels.
//
// Host-side only: argument validation, device memory for the SoA state, launch-shape selection,
// HIP-event timing.  No compute happens here and there is NO CPU fallback: every entry point that
// steps envs launches a kernel or fails with a ccx_status.
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include "ccx_internal.h"

namespace {

// a tile's observation region has at most 64 lanes x (3 + 2 * 64) float2 units (+ pad)
constexpr size_t kMaxObsUnits = 64u * (3u + 2u * 64u) + 2u;

thread_local char g_err[512] = "";

int ceil_log2(int n) {
    int g = 0;
    while ((1 << g) < n) ++g;
    return g;
}

}  // namespace

namespace ccxi {
int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}
}  // namespace ccxi
using ccxi::fail;

namespace {

// Per-cell geometry table of the padded grid (layout: ccx_kernels.hip, struct CellInfo).  This is
// config lowering, done once per handle: the reference evaluates the same predicates per agent
// per step (collectivecrossing.py:509-534, 551-563, 663-683; rewards.py:44-182).
std::vector<unsigned long long> build_cell_table(const ccx_params& p, const std::vector<uint8_t>* term_table = nullptr) {
    const int Wp = p.width + 3, Hp = p.height + 3;
    const int dc = (p.door_left + p.door_right) / 2;
    auto cell_ok = [&](int x, int y) {   // collectivecrossing.py:509-534
        if (!(x >= 0 && x <= p.width && y >= 0 && y <= p.height)) return false;
        if (y == p.division_y && !(p.door_left < x && x < p.door_right)) return false;
        if (y >= p.division_y && !(p.tram_left < x && x < p.tram_right)) return false;
        return true;
    };
    static const int DX[4] = {1, 0, -1, 0}, DY[4] = {0, 1, 0, -1};   // actions.py:18-24
    std::vector<unsigned long long> tab((size_t)Wp * Hp, 0ull);
    for (int y = 0; y <= p.height; ++y)
        for (int x = 0; x <= p.width; ++x) {   // border cells stay 0: never occupied
            unsigned nv = 0;
            for (int a = 0; a < 4; ++a) nv |= (cell_ok(x + DX[a], y + DY[a]) ? 1u : 0u) << a;
            const bool in_area = y >= p.division_y && p.tram_left <= x && x <= p.tram_right;
            const bool at_door = y == p.division_y && (x == p.door_left - 1 || x == p.door_right + 1);
            const bool dest_b = y == p.boarding_dest_y, dest_e = y == p.exiting_dest_y;
            const int adx = x > dc ? x - dc : dc - x;
            unsigned cls_b = 1, cls_e = 1;   // binary / constant_negative: always the constant rA
            int sd_b = 0, sd_e = 0;
            if (p.reward_mode == CCX_REWARD_DEFAULT) {
                cls_b = dest_b ? 1 : at_door ? 2 : in_area ? 3 : 0;
                cls_e = dest_e ? 1 : !in_area ? 3 : 0;
                sd_b = -(adx + (p.division_y - y));
                sd_e = adx + (y - p.division_y);
            } else if (p.reward_mode == CCX_REWARD_SIMPLE_DISTANCE) {
                cls_b = cls_e = 0;
                sd_b = -(y > p.boarding_dest_y ? y - p.boarding_dest_y : p.boarding_dest_y - y);
                sd_e = -(y > p.exiting_dest_y ? y - p.exiting_dest_y : p.exiting_dest_y - y);
            }
            // terminateds[id] on this cell (terminateds.py:66-82): the destination row, or the user's table
            const size_t ci = (size_t)y * (size_t)(p.width + 1) + (size_t)x;
            const bool term_b = (term_table && !term_table[0].empty()) ? term_table[0][ci] != 0 : dest_b;
            const bool term_e = (term_table && !term_table[1].empty()) ? term_table[1][ci] != 0 : dest_e;
            unsigned lo = nv | (in_area ? ccx::kCellInTram : 0u) | (at_door ? ccx::kCellAtDoor : 0u) |
                          ((dest_b ? 1u : 0u) << 8) | (cls_b << 9) | ((term_b ? 1u : 0u) << (8 + ccx::kCellTermShift)) |
                          ((dest_e ? 1u : 0u) << 12) | (cls_e << 13) | ((term_e ? 1u : 0u) << (12 + ccx::kCellTermShift)) |
                          ((unsigned)x << 16) | ((unsigned)y << 24);
            unsigned hi = ((unsigned)sd_b & 0xFFFFu) | (((unsigned)sd_e & 0xFFFFu) << 16);
            tab[(size_t)(y + 1) * Wp + (x + 1)] = (unsigned long long)lo | ((unsigned long long)hi << 32);
        }
    return tab;
}

// Default launch shape.  A tile (EW envs) is served by a sim wave and, when outputs are written,
// a writer wave on another SIMD; one step of a tile is a latency chain of ~1.5k cycles whatever
// EW is, so the batch should be cut into at least ~512 tiles (1024 waves = one per SIMD of the
// 256 CUs) before tiles are made fuller.  Measured on 4096 envs x 8 agents: 64 lanes/wave
// (512 tiles) 0.252 ms per 250 steps, 32 lanes 0.280 ms, 16 lanes 0.344 ms.
// rows: the shape of launches that write observation rows (h->shape / h->kp: paced, sized for the row stream);
// !rows: the shape of launches without them (h->shape_small / h->kp_small: rewards, flag bytes, compact rows -- bound by
// the sim chain and by how many tiles are resident, profiles/r04_shape_sweep.json).
int derive_shape(ccx_handle* h, const bool rows, ccx::LaunchShape& s, ccx::KParams& k) {
    const int glog = ceil_log2(h->N);
    const int G = 1 << glog;
    const int max_ew = 64 / G;
    int ew;
    bool drop_tables = false;
    if (h->lanes_per_wave > 0) {
        ew = h->lanes_per_wave / G;
    } else {
        const int target_tiles = 512;
        ew = max_ew;
        while (ew > 1 && (h->E + ew - 1) / ew < target_tiles) ew >>= 1;
    }
    if (ew < 1) ew = 1;
    if (ew > max_ew) ew = max_ew;
    if (h->lanes_per_wave == 0) {
        // Prefer the O(1) occupancy-table conflict masks: carry fewer envs per wave when that makes
        // the per-env tables fit in LDS (only grids too large even for one env per wave fall back
        // to the all-pairs compare).
        auto up = [](size_t v) { return (v + 15u) & ~(size_t)15u; };
        const size_t cells_ = (size_t)(h->params.width + 3) * (size_t)(h->params.height + 3);
        const size_t msz_ = (glog == 6) ? 8u : 4u;
        auto need = [&](int e) {
            const size_t units_ = (size_t)e * h->N * (3 + 2 * h->N);
            return up(cells_ * 8u) + (h->reward_table ? up(cells_ * 16u) : 0u) +
                   up(ccx::tile_head_bytes(16) + 7u * 1056u + up((size_t)e * 2u * (cells_ + 1u) * msz_)) + up((units_ + 2u) * 2u);
        };
        // The tables cost LDS -- one per env, (cells + 1) x 8 bytes -- and LDS is what bounds how many tiles a CU holds: a 24 x 16
        // grid keeps three 8-env tiles per CU where five fit without tables, a 64 x 48 grid ONE two-env tile; single-agent envs
        // on 12 x 8 one 64-env tile.  A batch that needs more ROUNDS because of them pays a round's time for each (64 x 48,
        // 8 agents: 0.18-0.24 of the HBM peak with tables, 0.51-0.63 without; 24 x 16, 16 384 envs: 0.69 vs 0.93;
        // profiles/r04_big_grid_scan.txt).  The all-pairs masks cost the sim chain ~5.5 % per lane of the group (8 agents:
        // 0.47 vs 0.34 us per env-step; 32: 2.8 x).  For groups of <= 16 lanes the tables go (and the tile keeps its lanes)
        // where the rounds saved outweigh that (16 agents on 64 x 48: 0.25-0.35 of the peak with tables -- one 16-lane tile per
        // CU --, 0.77-0.88 without).
        const int ew_fit = [&] { int e = ew; if (need(1) <= 96u * 1024u) while (e > 1 && need(e) > 96u * 1024u) e >>= 1; return e; }();
        if (glog <= 4 && h->tun_occ_tables < 0) {
            auto rounds = [&](int e, size_t bytes_per_tile) {
                const size_t per_cu = std::max<size_t>(1, std::min<size_t>(160u * 1024u / std::max<size_t>(bytes_per_tile, 1), 5u));
                const size_t tiles_ = (size_t)(h->E + e - 1) / (size_t)e;
                return (tiles_ + per_cu * h->num_cus - 1) / (per_cu * h->num_cus);
            };
            const size_t with_tables = rounds(ew_fit, need(ew_fit));
            const size_t without = rounds(ew, need(ew) - up((size_t)ew * 2u * (cells_ + 1u) * msz_));
            if ((double)without * (1.0 + 0.055 * G) < (double)with_tables) drop_tables = true;
        }
        if (!drop_tables) ew = ew_fit;
    }
    // Batches too small to be memory-bound (round 2): one env-step of a tile takes the sim chain's ~0.5 us
    // whatever the tile holds, so what counts is that no writer wave takes longer than that and that every
    // wave has a SIMD of its own.  Full 64-lane tiles with TWO writer waves each do both as long as
    // 3 waves x tiles stays near the 1024 SIMDs: C2 geometry, us per env-step, full tiles + 2 writers vs the
    // half-empty single-writer tiles chosen before: 1024 envs 0.53 vs 0.58, 2048 envs 0.54 vs 0.59 (0.63 vs
    // 0.58 of the HBM peak), 3072 envs 0.63 vs 0.72 (0.81 vs 0.70).
    bool small_batch = false, half_tiles = false;
    if (h->lanes_per_wave == 0 && h->writers != 1 && h->waves_per_block == 0) {
        const long long full_tiles = (h->E + max_ew - 1) / max_ew;
        const int full_n4 = max_ew * h->N * (3 + 2 * h->N) / ((h->N % 2 == 0) ? 2 : 1);
        if (full_n4 <= 64 * 12 && full_tiles * (1 + (h->writers > 0 ? h->writers : 2)) <= 1400 && full_tiles * max_ew >= 256) {
            auto up = [](size_t v) { return (v + 15u) & ~(size_t)15u; };   // (only if the LDS tables of a full tile fit)
            const size_t cells_ = (size_t)(h->params.width + 3) * (size_t)(h->params.height + 3);
            const size_t msz_ = (glog == 6) ? 8u : 4u;
            const size_t units_ = (size_t)max_ew * h->N * (3 + 2 * h->N);
            // (with the writer slots of the writer count chosen below -- up to four for small batches: a check against
            // two let a 6 x 16 grid with one agent per env through whose full tile then lost its occupancy tables, and
            // with them the in-kernel policies; found by the round-3 hypothesis soak)
            const size_t wslots_ = (size_t)(h->writers > 0 ? h->writers : 4) * 1056u;
            if (up(cells_ * 8u) + (h->reward_table ? up(cells_ * 16u) : 0u) +
                    up(ccx::tile_head_bytes(16) + wslots_ + up((size_t)max_ew * 2u * (cells_ + 1u) * msz_)) +
                    up((units_ + 2u) * 2u) <= 96u * 1024u) {
                small_batch = true;
                ew = max_ew;
                // Up to 128 full tiles (C2 geometry: 1024 envs) half the CUs would stand idle: HALF tiles with four writers
                // each put a tile on every CU and halve each writer's share (round 3, us per env-step with full outputs:
                // 1024 envs 0.355 vs 0.398-0.402 with full tiles and 3-4 writers, 512 envs 0.353 vs 0.395).  What remains
                // is the step's own latency chain sim -> hand-off -> writer (~0.35 us), not bytes.
                // (round 4 sweep: only where a full tile has >= 9 store iterations per step -- with fewer the row work is too
                // small to be worth two half-width sim waves: 3 agents, 256-2048 envs 0.37-0.39 vs 0.34 us per env-step)
                if (full_tiles <= 128 && max_ew >= 2 && h->writers == 0 && (!rows || full_n4 >= 64 * 9)) {
                    ew = max_ew / 2;
                    half_tiles = true;
                }
            }
        }
    }
    const int tiles = (h->E + ew - 1) / ew;
    // writer waves per tile: enough that a writer handles <= ~6 store iterations per step
    const int units = ew * h->N * (3 + 2 * h->N);
    const int n4 = (h->N % 2 == 0) ? units / 2 : units;
    // Small tiles (<= 12 store iterations per step, e.g. C2's 9.5): ONE writer wave whose stores in
    // flight are bounded (many small tiles oversubscribe the HBM write queues; the bound is the
    // regulator when step pacing is off and a safety net when it is on).
    // Larger tiles: 2-3 writers, no throttle (measured: no effect).
    const bool small_tiles = n4 <= 64 * 12;
    // small batches: writer 0 = small outputs, the others share the observation rows (KParams::writer0_small);
    // three writers while 4 waves x tiles still fit the 1024 SIMDs (2048-env C2: 0.54 -> 0.485 us per step); round 3, with
    // the sim chain at 0.27 us and the writer loops compiled per role: four writers up to 160 tiles (1024 envs 0.39 -> 0.41 of
    // the peak), three beyond (2048 envs: 0.81-0.82 with three, 0.80-0.81 with four)
    // Small tiles beyond that (C2's 4096 x 8: 9.5 store iterations per tile and step): TWO writers split by role
    // as well, as long as 3 waves per tile fit one round (16 wavefronts per CU).  With everything on one writer
    // that wave had 1580 clocks of work per step next to the sim's 1200 and held the tile at ~0.72 us per step --
    // C2 was bound by its writer wave, not by memory; with two the pace follows the memory side down to
    // ~0.70 us: 0.86 -> 0.92 of the HBM peak in one call (round 2; round 1 measured two writers as no gain,
    // but its sim wave was 20 % slower and hid the difference).
    // Writer waves per tile (round 4: re-derived from the E x N x output-mode sweep, profiles/r04_shape_sweep.json, which
    // measures every (lanes, writers) candidate per point; the tables of DESIGN.md 4 are that file):
    //   launches WITHOUT observation rows (rewards / flag bytes / compact rows: bound by the sim chain and by how many tiles
    //     are resident): two writers split by role, ONE from 1024 tiles on (8 agents, 16 384 envs: 0.55 vs 0.91 us per
    //     env-step with two; 32 agents, 32 768 envs: 4.1 vs 6.3) -- lane groups of 32 / 64 only from 16 384 tiles on (their
    //     compact rows keep a second writer busy), single-agent envs never (one writer: +30-40 %); with 257-512 tiles -- one
    //     per two SIMDs, C2's 4096 envs -- FOUR writers in one-tile workgroups (the two spare waves only poll): every agent
    //     count of the sweep is 4-6 % faster than with two (8 agents, 4096 envs: 0.319 vs 0.338 us per env-step;
    //     profiles/scratch/calls/r04_call19.sh, r04_call21.sh);
    //   small batches (unpaced, full or half tiles): 3-4 writers split by role, as measured in round 3;
    //   large tiles (> 24 store iterations per step): 3;
    //   small tiles (<= 12 iterations): with <= 8 iterations per tile (1-3 agents) THREE writers whatever the batch (3
    //     agents, 8192 envs 0.38 vs 0.47 us with two; N = 1, 65 536 envs: 1.17 vs 2.40 us with the single throttled writer of
    //     rounds 1-3); with 9-12 iterations (C2's class) two writers split by role up to four tiles per CU (two six-wave
    //     workgroups; the rule used to read "while 3 waves per tile fit one round", i.e. up to 1365 tiles, and every batch of
    //     8193 .. 10 920 envs of C2 ran at 0.49 of the peak with a third workgroup on some CUs: profiles/r04_ragged_c2.txt) --
    //     in pairs up to two tiles per CU (C2's 4096 envs), THREE in one-tile workgroups from there to four (5000 .. 8192 envs:
    //     0.92-0.94 vs 0.89-0.93 with two, two runs of that table) --, beyond that ONE throttled writer as before: the sweep's short launches put three within 2 % of it, but the bench's settled
    //     launches do not (C2 geometry, secondary.workloads: 16 384 envs 0.881 vs 0.845 of the peak with three, 65 536 envs
    //     0.818 vs 0.747).  TWO writers are a cliff in several rounds (6-wave workgroups: +35-80 %);
    //   tiles in between: 2, or 1 + four tiles per workgroup when that makes the batch fit one round (C3, below).
    const long long cap_waves = 16ll * h->num_cus;
    int writers = h->writers > 0 ? h->writers
                  : !rows ? ((tiles > 256 && tiles <= 512) ? 4   // one tile per two SIMDs: five-wave workgroups, one tile each (below)
                             : glog <= 2 ? 2 : glog <= 4 ? (tiles > 1280 ? 1 : 2) : (tiles >= 1500 ? 1 : 2))
                  : small_batch ? (half_tiles || tiles <= 160 ? 4 : 3)
                  : n4 > 64 * 24 ? 3
                  : small_tiles ? (n4 <= 64 * 8 ? 3 : tiles <= 2 * h->num_cus ? 2 : tiles <= 4 * h->num_cus ? 3 : 1) : 2;
    if (writers > 7) writers = 7;
    // tiles per workgroup: two small tiles share one cell table / one CU slot (with the throttle:
    // 4.43e9 vs 4.23e9 env-steps/s on C2; 3 or 4 per workgroup leave CUs idle and lose 5-10 %)
    int tpb = h->waves_per_block > 0 ? h->waves_per_block
              : small_batch ? 1 : (!rows && writers == 4) ? 1
              // without rows (profiles/r04_noobs_scan.txt): two-writer tiles NEVER in pairs (six-wave workgroups: 0.45 vs 0.36 us
              // per env-step from 375 tiles on -- every batch between 2049 and 8191 envs of C2 but 4096), one-writer tiles in
              // pairs from 1500 tiles on (32 agents, 8192 envs: 1.07 vs 1.31 us)
              : (!rows && writers == 2) ? 1
              : (!rows && writers == 1) ? (tiles >= 1500 ? 2 : 1)
              : (rows && small_tiles && writers == 1 && tiles <= 8 * h->num_cus) ? 1   // (one round of two-wave workgroups: +1-3 % over pairs, 10 000 .. 16 384 envs of C2)
              : (rows && small_tiles && n4 > 64 * 8 && writers == 3) ? 1
              : ((tiles > 8192 && writers != 2) || (small_tiles && tiles >= 512)) ? 2 : 1;   // (never two writers in pairs: six-wave workgroups, 16 agents x 36 032 envs ran at 0.47)
    // One round beats two (round 2): a CU holds 16 wavefronts of this kernel (4 per SIMD at its ~100
    // VGPRs).  If the batch needs more than that with the writer count above but fits with ONE writer
    // wave per tile, and that writer's share stays <= 36 store iterations per step, every tile is
    // resident for the whole launch and the tiles of a step sweep the slab exactly once -- C3 (4096 x 32:
    // 2048 tiles x 4 waves = two rounds) 0.84 -> 0.87 of the HBM peak with 1 writer and 4 tiles per
    // workgroup.  (C5, 67 KB per tile, already fits one round with 3 writers; one writer is too slow there.)
    if (rows && h->writers == 0 && h->waves_per_block == 0 && !small_tiles) {
        const long long cap = cap_waves;
        if ((long long)tiles * (1 + writers) > cap && (long long)tiles * 2 <= cap && n4 <= 64 * 36) {
            writers = 1;
            tpb = 4;
        } else if ((long long)tiles * (1 + writers) > cap && writers == 3 && n4 <= 64 * 36) {
            // several rounds anyway: two writers in one-tile workgroups keep five tiles per CU resident instead of four (C3
            // geometry, 8192 envs: 21.7 vs 25.2 us per env-step; 32 768 envs 1.01).  NOT with two tiles per workgroup:
            // six-wave workgroups leave a quarter of a CU's 16 wave slots empty (+40 % at 32 768 envs)
            // From 8192 tiles on ONE writer (eight tiles per CU resident): 16 384 envs 44.2 vs 49.3 us, 32 768 envs 90 vs 100.
            if (tiles < 8192) {
                writers = 2;
                tpb = 1;
            } else {
                writers = 1;
            }
        } else if (writers == 2 && tiles >= 8192 && n4 <= 64 * 36) {
            // the two-writer class (13-24 store iterations per tile: 12-20 agents) in many rounds: the same one writer, in
            // pairs (20 agents on 24 x 16, 32 768 envs: 0.74 vs 0.42 of the peak; 40 x 30: 0.70 vs 0.65;
            // profiles/r04_big_grid_scan.txt)
            writers = 1;
            tpb = 2;
        }
    }
    while (tpb > 1 && tpb * (1 + writers) > 8) --tpb;   // <= 512 threads per workgroup
    s.glog = glog;
    s.envs_per_wave = ew;
    s.waves_per_block = tpb;
    s.writers = writers;
    // with step pacing (the default) the throttle is only a safety net against a collapse of the
    // drain rate when the pace is too fast; without pacing it is the regulator (16, see above)
    s.store_throttle = h->store_throttle > 0 ? h->store_throttle
                       : (h->store_throttle == 0 && small_tiles && writers == 1)
                             ? (h->step_pace_ns == -1 ? 16 : 48) : 0;
    s.num_blocks = (tiles + tpb - 1) / tpb;

    // LDS carve-up (see ccx_kernels.hip): [cell table][tiles][u16 obs table]
    auto up16 = [](size_t v) { return (v + 15u) & ~(size_t)15u; };
    const size_t cells = (size_t)(h->params.width + 3) * (size_t)(h->params.height + 3);
    const size_t msz = (glog == 6) ? 8u : 4u;
    const size_t occ_bytes = up16((size_t)ew * 2u * (cells + 1u) * msz);
    const size_t table = up16((size_t)(units + 2) * 2u);
    const size_t rtab_bytes = h->reward_table ? up16(cells * 16u) : 0u;        // user reward table behind the cell table
    const size_t off_tiles = up16(cells * 8u) + rtab_bytes;
    // The hand-off ring takes 32 slots (8 KB) -- or 16 where that costs a CU a resident workgroup: LDS is what bounds the
    // residency of the big-tile shapes (C5-64: 40 KB per workgroup with 4 KB of ring, four per CU; with 8 KB only three).
    // (The sim wave proves "slot free" from a progress value it reads once per 16-step burst: 16 slots are the minimum
    // for which that stale value suffices almost always.)
    const size_t lds_cu = 160u * 1024u;
    size_t off_ws = 0, off_occ = 0, tile_stride = 0, total = 0;
    uint32_t slots = ccx::kMaxStageSlots;
    // Small batches (unpaced, writers split by role): a second staging slot per writer lets a row writer take TWO steps
    // per iteration whenever the sim wave is that far ahead (ccx_rollout_body.inc) -- if the LDS has the room.
    size_t wsw = 1;
    auto lay_out = [&](uint32_t nslots, int tiles_pb, bool with_occ) {
        off_ws = ccx::tile_head_bytes(nslots);                   // xch + hand-off words + stage ring
        off_occ = off_ws + (size_t)writers * wsw * 1056u;        // WSlot(s) per writer
        tile_stride = up16(off_occ + (with_occ ? occ_bytes : 0));
        total = off_tiles + (size_t)tiles_pb * tile_stride + table;
    };
    // Tile PAIRS hold fewer tiles per CU than single tiles where the waves are what bounds them (three writers: two
    // eight-wave workgroups = 4 tiles against five four-wave ones): a batch that fits one round of singles but not of pairs
    // takes singles (4 agents x 17 776 envs: 0.94 vs 1.40 us per env-step; profiles/r04_rows_456.txt).
    if (h->writers == 0 && h->waves_per_block == 0 && tpb == 2 && off_tiles < 24u * 1024u) {
        auto rounds_tpb = [&](int t_) {
            lay_out(16, t_, !drop_tables);
            const size_t blocks = std::max<size_t>(1, std::min<size_t>(lds_cu / std::max<size_t>(total, 1), (size_t)(20 / (t_ * (1 + writers)))));
            const size_t at_once = blocks * (size_t)t_ * (size_t)h->num_cus;
            return ((size_t)tiles + at_once - 1) / at_once;
        };
        if (rounds_tpb(1) < rounds_tpb(2)) {
            tpb = 1;
            s.waves_per_block = tpb;
            s.num_blocks = tiles;
        }
    }
    // A CELL table that fills much of the LDS by itself (64 x 48: 27 KB, 80 x 60: 39 KB, 100 x 100: 85 KB -- one per
    // workgroup) limits the workgroups per CU, and then the tiles per workgroup decide how much of the batch is resident:
    // 100 x 100, 8 agents, 16 384 envs in one-tile workgroups = 256 tiles at a time, 0.35 of the HBM peak and 3.9 us per
    // env-step without rows; four one-writer tiles per workgroup = 1024 at a time, 0.85 and 0.90 us
    // (profiles/r04_big_grid_scan.txt).  Among the rule's own (writers, tiles) and the fuller workgroups -- the same writers
    // with more tiles, ONE writer with up to four -- the one that needs the fewest rounds is taken (ties: the rule's writers,
    // then fewer tiles per workgroup).
    if (h->writers == 0 && h->waves_per_block == 0 && off_tiles >= 24u * 1024u) {
        auto rounds_with = [&](int w_, int t_) {
            const int w_saved = writers;
            writers = w_;
            lay_out(16, t_, !drop_tables);
            if (total > 96u * 1024u) lay_out(16, t_, false);      // (tables that do not fit are not kept: below)
            writers = w_saved;
            const size_t blocks = std::max<size_t>(1, std::min<size_t>(lds_cu / std::max<size_t>(total, 1), (size_t)(16 / (t_ * (1 + w_)))));
            const size_t at_once = blocks * (size_t)t_ * (size_t)h->num_cus;
            return total > lds_cu ? (size_t)1 << 30 : ((size_t)tiles + at_once - 1) / at_once;
        };
        int best_w = writers, best_t = tpb;
        size_t best = rounds_with(writers, tpb);
        for (int pass = 0; pass < 2; ++pass) {
            const int w_ = pass == 0 ? writers : 1;
            for (int t_ = (pass == 0 ? tpb + 1 : 2); t_ * (1 + w_) <= 8; ++t_) {
                const size_t r_ = rounds_with(w_, t_);
                if (r_ < best) { best = r_; best_w = w_; best_t = t_; }
            }
        }
        writers = best_w;
        tpb = best_t;
        s.waves_per_block = tpb;
        s.writers = writers;
        s.num_blocks = (tiles + tpb - 1) / tpb;
        s.store_throttle = h->store_throttle > 0 ? h->store_throttle
                           : (h->store_throttle == 0 && small_tiles && writers == 1) ? (h->step_pace_ns == -1 ? 16 : 48) : 0;
    }
    // (by default only with half tiles, <= 1024 envs of the C2 geometry: +3 % there; at 2048 envs the unpaced write stream is
    // the limit and two steps' stores back to back cost it 5-7 %: tunable pair_rows = 1 forces it for every small batch)
    if (small_batch && writers >= 2 && (h->tun_pair_rows == 1 || (h->tun_pair_rows < 0 && half_tiles))) {
        wsw = 2;
        lay_out(16, tpb, !drop_tables);
        if (total > 96u * 1024u) wsw = 1;
    }
    lay_out(16, 1, !drop_tables);
    const bool tables_can_fit = total <= 96u * 1024u;    // (with one tile per workgroup; a 100 x 100 grid: never)
    lay_out(16, tpb, !drop_tables);
    if (h->waves_per_block == 0 && tables_can_fit)        // a default never costs the occupancy tables their LDS
        while (tpb > 1 && total > 96u * 1024u) {
            --tpb;
            lay_out(16, tpb, !drop_tables);
        }
    s.waves_per_block = tpb;
    s.num_blocks = (tiles + tpb - 1) / tpb;
    s.occ = 1;
    if (total > 96u * 1024u || h->tun_occ_tables == 0 || drop_tables) {          // tables too big (or switched off, or given up for residency): all-pairs conflict masks instead
        s.occ = 0;
        lay_out(16, tpb, false);
    }
    // Single-agent envs: a table of (cells + 1) entries per env, 64 envs per wave -- 73 KB for a 12 x 8 grid, ONE tile per CU:
    // a batch of more than 64 x CUs envs ran in rounds at twice the time per env-step (17 768 envs: 0.83 us against 0.43 at
    // 15 800).  An agent alone in its env collides with nobody; the all-pairs masks cost it three VALU ops (+4 % while the
    // tables fit one round, half the time when they do not: profiles/r04_occ_tables.txt).
    if (s.occ && h->tun_occ_tables < 0 && (drop_tables || (glog == 0 && (size_t)s.num_blocks > (lds_cu / total) * (size_t)h->num_cus))) {
        s.occ = 0;
        lay_out(16, tpb, false);
    }
    const size_t fit16 = std::min<size_t>(lds_cu / total, 16u);
    lay_out(slots, tpb, s.occ != 0);
    if (total > 150u * 1024u || std::min<size_t>(lds_cu / total, 16u) < fit16) slots = 16;
    lay_out(slots, tpb, s.occ != 0);
    s.lds_bytes = total;
    s.lds_bytes_observe = up16((size_t)tpb * 1056u + table);

    const ccx_params& p = h->params;
    k.W = p.width; k.H = p.height; k.div = p.division_y;
    k.tl = p.tram_left; k.tr = p.tram_right; k.dl = p.door_left; k.dr = p.door_right;
    k.dc = (p.door_left + p.door_right) / 2;  // observations.py:70-71, rewards.py:81
    k.Nb = p.num_boarding; k.N = h->N; k.bdy = p.boarding_dest_y; k.edy = p.exiting_dest_y;
    k.reward_mode = p.reward_mode; k.term_mode = p.terminated_mode; k.max_steps = p.max_steps;
    k.E = h->E; k.EW = ew; k.waves_per_block = tpb; k.units_per_wave = units; k.writers = writers;
    k.off_tiles = (uint32_t)off_tiles; k.tile_stride = (uint32_t)tile_stride;
    k.reward_table = h->reward_table;
    k.off_rtab = h->reward_table ? (uint32_t)up16(cells * 8u) : 0u;
    k.user_tables = (h->reward_table || !h->term_table[0].empty() || !h->term_table[1].empty()) ? 1u : 0u;
    k.off_ws = (uint32_t)off_ws; k.off_occ = (uint32_t)off_occ;
    k.occ_words = s.occ ? (uint32_t)(occ_bytes / 4u) : 0u;
    k.off_table = (uint32_t)(off_tiles + (size_t)tpb * tile_stride);
    k.stage_slots = slots;
    k.ws_per_writer = (uint32_t)wsw;
    k.wp_magic = (uint32_t)((0x100000000ull + (unsigned long long)(p.width + 3) - 1ull) / (unsigned long long)(p.width + 3));
    k.writer_vmcnt = (uint32_t)s.store_throttle;
    k.writer0_small = (h->tun_writer_roles >= 0 ? h->tun_writer_roles != 0 : (small_batch || small_tiles)) && writers >= 2 ? 1u : 0u;

    // Step pacing (ccx_kernels.hip).  The schedule limits the rate at which the resident
    // workgroups inject observation stores; its start value assumes a drain rate of 6.8 TB/s and the
    // kernel retunes it after every long launch (bounds: 7.8 TB/s .. a sixth of the start rate).
    CCX_HIP(hipSetDevice(h->device));
    int per_cu = ccx::rollout_blocks_per_cu(s, h->N);
    if (per_cu < 1) per_cu = 1;
    s.resident_blocks = std::min(s.num_blocks, per_cu * h->num_cus);
    const double tile_bytes = (double)ew * h->N * (4.0 * (6 + 4 * h->N) + 10.0) + ew;
    s.step_bytes = tile_bytes * tpb * s.resident_blocks;
    auto to_fp = [](double ns) { double v = ns / 10.0 * 256.0; return (uint32_t)(v < 1.0 ? 1.0 : (v > 4.0e9 ? 4.0e9 : v)); };
    // A batch whose resident tiles cannot even fill the drain rate at a fast 0.40 us per env-step is
    // bound by the step chain, not by memory: pacing could only cost it (a clock read per step).  (0.45 us until the end of
    // round 4: the chain has become faster than that, and C2 batches of 2160 .. 2430 envs ran unpaced INTO the cliff -- 2304
    // envs at 0.62 of the peak between 0.82 at 2048 and 0.83 at 2500; profiles/r04_small_batch.txt.)
    const bool can_saturate = s.step_bytes / 7000.0 >= 400.0;
    k.pace_state = (!rows || h->step_pace_ns == -1 || (h->step_pace_ns == 0 && !can_saturate)) ? nullptr : h->pace_state;
    if (rows) h->ring_when_paced = s.step_bytes / 7000.0 < 550.0;
    k.pace_adapt = (h->step_pace_ns == 0) ? 1u : 0u;
    {   // how many steps make a launch worth pacing (~12 us) / long enough to judge its lateness (~50 us), at the assumed rate
        const double step_ns = s.step_bytes / 6800.0;
        const double a = std::ceil(50000.0 / (step_ns > 1.0 ? step_ns : 1.0));
        k.adapt_min_k = (uint32_t)(a < 4.0 ? 4.0 : a > 64.0 ? 64.0 : a);
        k.pace_min_k = k.adapt_min_k / 4u < 2u ? 2u : k.adapt_min_k / 4u;
    }
    k.resident_blocks = (uint32_t)s.resident_blocks;

    // observation address table of this shape (ccx_kernels.h: obs_unit_addr).  Shape changes are rare and
    // never happen inside a graph capture: wait for launches that still read the old table, then copy.
    if (rows) {
        h->obs_table_host.assign((size_t)units + 2u, 0);
        for (int w = 0; w < units; ++w) h->obs_table_host[(size_t)w] = ccx::obs_unit_addr((uint32_t)w, h->N, glog);
        CCX_HIP(hipStreamSynchronize(h->stream));
        CCX_HIP(hipMemcpy(h->obs_table, h->obs_table_host.data(), h->obs_table_host.size() * sizeof(uint16_t),
                          hipMemcpyHostToDevice));
    }
    k.obs_table = h->obs_table;
    k.pace_min_fp = to_fp(s.step_bytes / 8000.0);    // (the spec peak; round 2 stopped at 7.8 TB/s, which multi-tile-per-CU shapes now reach)
    k.pace_max_fp = to_fp(s.step_bytes / 1100.0);
    if (rows) {
        h->pace_init_fp = h->step_pace_ns > 0 ? to_fp((double)h->step_pace_ns)
                          : h->pace_start_ns > 0.0f ? to_fp((double)h->pace_start_ns) : to_fp(s.step_bytes / 6800.0);
        h->pace_start_source = h->step_pace_ns > 0 ? CCX_PACE_START_FIXED
                               : h->pace_start_ns > 0.0f ? CCX_PACE_START_CALLER : CCX_PACE_START_ASSUMED;
        h->pace_dirty = true;
        h->pace_needs_calibration = false;
    }
    k.r_dest = p.boarding_destination_reward; k.r_door = p.tram_door_reward;
    k.r_area = p.tram_area_reward; k.r_f = p.distance_penalty_factor;
    k.r_nogoal = p.no_goal_reward; k.r_pen = p.step_penalty;
    k.env_offset = h->env_offset;
    k.pool_size = h->pool_size;
    // cursor stride of the reset pool: entry (global_env + episode * stride) mod P.  total_envs mod P keeps
    // the walk independent of the world size; when P divides total_envs that would be 0 and every env
    // would restart from ONE placement forever, so the stride is 1 then (env e walks e, e+1, e+2, ...)
    k.pool_stride = h->pool_size > 0 ? (long long)(h->total_envs % h->pool_size) : 0;
    if (h->pool_size > 0 && k.pool_stride == 0) k.pool_stride = 1 % h->pool_size;
    // Write-window defaults (measured round 2).  Large tiles (tens of KB per tile and step:
    // C3, C5) drain 5-10 % faster when the tiles of a round are phased over the step
    // period in tile order and groups of 16 adjacent tiles go to one XCD, dealt round-robin: the chip then
    // writes one window that sweeps through the slab instead of 1000+ regions at once (C3 0.80 -> 0.86,
    // C5-64 0.79 -> 0.89, C5-50 0.77 -> 0.83 of the HBM peak in one call).  Small tiles (C2: 10 KB per tile
    // and step) show no difference while there are two of them per CU (4096 envs) and keep the common phase and
    // the XCD-contiguous mapping; LARGER batches of small tiles are 1000+ regions at once again and gain the same
    // way (C2 geometry, one call: 16 384 envs 0.879 -> 0.908, 32 768 envs in two rounds 0.830 -> 0.882).
    const bool big_tiles = !small_tiles || (long long)tiles >= 4ll * h->num_cus;
    k.pace_phase = (uint32_t)(h->tun_pace_phase >= 0 ? h->tun_pace_phase : (big_tiles ? 1 : 0));
    // groups of ~1 MiB of one step's slab per XCD: g workgroups with g * (bytes a workgroup writes per step)
    // closest to 1 MiB, a power of two in 1..32 (C5-64: 16 x 67 KB, C3: 8 x 137 KB)
    int auto_map = 0;
    if (big_tiles) {
        const double wg_bytes = tile_bytes * tpb;
        int g = 1;
        while (g < 32 && wg_bytes * g * 1.41 < 1048576.0) g <<= 1;
        auto_map = 1;
        while ((1 << (auto_map - 1)) < g) ++auto_map;
    }
    k.tile_map = (uint32_t)(h->tun_tile_map >= 0 ? h->tun_tile_map : auto_map);

    if (!rows) return CCX_OK;
    // Short launches (<= 16 steps, ccx_step.hip): a workgroup = one tile = a sim wave + row waves, as many tiles as the
    // batch fills, halved only while there are fewer than one per two CUs (a one-step launch is bound by how fast its waves are
    // dispatched and drained: with the row waves' table words preloaded, 512 tiles of 64 lanes + 2 row waves step C2's 4096 envs
    // in 3.64 us, 1024 tiles of 32 + 1 in 3.72; 512 envs: 128 tiles of 32 lanes 3.04 us, 512 tiles of 8 lanes 3.30;
    // profiles/r04_step_k1_shapes_*.txt, r04_step_scan.txt), never more envs per wave than the rollout shape (the observation address
    // table of a smaller tile is a prefix of the rollout's).  Row waves: enough that one handles <= ~6 store iterations per step
    // (small tiles) or 10-14 (large ones), at most 5 (C5-64: 14.9 us with five, 15.5 with seven).  Grids whose tables exceed
    // the LDS keep the rollout kernel.
    {
        ccx::StepShape& ss = h->step_shape;
        int sew = ew;
        if (h->tun_step_lanes > 0) sew = std::max(1, std::min(ew, h->tun_step_lanes / G));
        else while (sew > 1 && ((h->E + sew - 1) / sew < h->num_cus / 2 ||      // (profiles/r04_step_scan.txt)
                                ccx::step_lds_bytes(glog, sew, h->N, (int)cells, h->reward_table != nullptr) > 96u * 1024u))
            sew >>= 1;                                                          // (... and until the tile's tables fit: 40 x 30, 4096 envs took the rollout kernel for 8 KB)
        ss.glog = glog;
        ss.envs_per_wave = sew;
        ss.lds_bytes = ccx::step_lds_bytes(glog, sew, h->N, (int)cells, h->reward_table != nullptr);
        const int sunits = sew * h->N * (3 + 2 * h->N);
        const int sits = ((h->N % 2 == 0 ? sunits / 2 : sunits) + 63) / 64;
        ss.row_waves = h->tun_step_rows > 0 ? h->tun_step_rows
                       : sits <= 6 ? 1 : sits <= 12 ? 2 : sits <= 33 ? 3 : sits <= 45 ? 4 : 5;
        ss.num_blocks = (h->E + sew - 1) / sew;
        // (the step kernel carries its own tables whatever the rollout shape does about its; EVERY workgroup stages the cell
        //  table and zeroes its tables per launch: past ~24 MB of that per step the rollout kernel is the faster one -- 40 x 30,
        //  4096 envs: 60 MB, 10.2 vs 5.8 us per step; 80 x 60: 113 vs 6.8; profiles/r04_step_big_grid.txt)
        //  -- or three times the step's rows where those are the larger part: 20 agents on 40 x 30, 4096 envs: 11.9 vs 16.4 us)
        const double step_rows_bytes = (double)h->E * h->N * (6.0 + 4.0 * h->N) * 4.0;
        ss.ok = (ss.lds_bytes <= 96u * 1024u &&
                 (double)ss.num_blocks * (double)ss.lds_bytes <= std::max(24.0e6, 3.0 * step_rows_bytes)) ? 1 : 0;
    }
    return CCX_OK;
}

int choose_shape(ccx_handle* h) {
    // (seeds / epsilon live in h->kp and survive a re-derivation)
    const uint32_t rng_lo = h->kp.rng_lo, rng_hi = h->kp.rng_hi, eps_thr = h->kp.eps_thr;
    int rc = derive_shape(h, true, h->shape, h->kp);
    if (rc) return rc;
    rc = derive_shape(h, false, h->shape_small, h->kp_small);
    h->kp.rng_lo = rng_lo; h->kp.rng_hi = rng_hi; h->kp.eps_thr = eps_thr;
    return rc;
}

int validate_params(const ccx_params* p) {
    if (!p) return fail(CCX_EINVAL, "params is NULL");
    const int n = p->num_boarding + p->num_exiting;
    if (p->num_boarding < 0 || p->num_exiting < 0 || n < 1)
        return fail(CCX_EINVAL, "need at least one agent (got %d boarding + %d exiting)",
                    p->num_boarding, p->num_exiting);
    if (n > 64)
        return fail(CCX_EINVAL, "%d agents per env: libccx supports at most 64 (one wavefront "
                    "lane per agent); the reference itself caps the total at 50 (configs.py:166)", n);
    if (p->width < 1 || p->width > 100 || p->height < 1 || p->height > 100)
        return fail(CCX_EINVAL, "grid %dx%d outside 1..100 (configs.py:39-40)", p->width, p->height);
    if (p->division_y < 1 || p->division_y > 100) return fail(CCX_EINVAL, "division_y out of range");
    if (p->reward_mode < 0 || p->reward_mode > 3) return fail(CCX_EINVAL, "unknown reward_mode %d", p->reward_mode);
    if (p->terminated_mode < 0 || p->terminated_mode > 1)
        return fail(CCX_EINVAL, "unknown terminated_mode %d", p->terminated_mode);
    if (p->truncated_mode != 0) return fail(CCX_EINVAL, "unknown truncated_mode %d", p->truncated_mode);
    if (p->max_steps < 1) return fail(CCX_EINVAL, "max_steps must be >= 1");
    return CCX_OK;
}

int begin_timed(ccx_handle* h) {
    if (h->timing) CCX_HIP(hipEventRecord(h->ev_start, h->stream));
    return CCX_OK;
}
int end_timed(ccx_handle* h) {
    if (h->timing) {
        CCX_HIP(hipEventRecord(h->ev_stop, h->stream));
        h->timed = true;
    }
    return CCX_OK;
}

#ifdef CCX_LAG_TRACE
int* g_lag_buf = nullptr;
#endif

// Start-up calibration of the pace controller.  The adaptive controller needs a start value within
// a few per cent of the rate at which THIS box drains THIS process's write stream; round 2 shipped four learned values
// for the four bench shapes and started every other shape from an assumed 6.8 TB/s.  Now the first paced launch of a
// handle / launch shape measures it: a pure write stream (the writer waves' own store instruction) into the caller's
// trajectory buffer -- the memory the rollout is about to overwrite anyway -- for ~2.5 ms (which also takes the memory
// side through its start-up transient), best pass wins.  The paced rollouts of all shapes settle at 1.05-1.15 x that
// rate (round 3: C2 7.3-7.6 TB/s vs fills of 6.4-7.1 in the same process), so the controller starts 5 % above the
// measured rate and keeps descending at its fast rate until a launch comes in late.  Synchronises the stream once; skipped inside a stream capture, with a caller's start value
// (ccx_set_step_pace_start), with a fixed pace, or with ccx_set_pace_calibration(h, 0) / CCX_PACE_CALIBRATION=0.
int calibrate_pace(ccx_handle* h, float* obs, size_t obs_bytes) {
    const size_t bytes = std::min<size_t>(obs_bytes & ~(size_t)15, (size_t)3 << 30);
    if (bytes < ((size_t)16 << 20)) return CCX_OK;            // too small a buffer to say anything about a stream
    hipEvent_t e0 = nullptr, e1 = nullptr;
    CCX_HIP(hipEventCreate(&e0));
    if (hipError_t ee = hipEventCreate(&e1); ee != hipSuccess) {
        (void)hipEventDestroy(e0);
        return fail(CCX_EHIP, "hipEventCreate failed: %s", hipGetErrorString(ee));
    }
    double best_gbs = 0.0, spent_ms = 0.0;
    int rc = CCX_OK;
    for (int pass = 0; pass < 24 && spent_ms < 2.5 && rc == CCX_OK; ++pass) {
        float ms = 0.0f;
        hipError_t e = hipEventRecord(e0, h->stream);
        if (e == hipSuccess) e = ccx::launch_write_probe(h->stream, obs, bytes, h->num_cus * 8);
        if (e == hipSuccess) e = hipEventRecord(e1, h->stream);
        if (e == hipSuccess) e = hipEventSynchronize(e1);
        if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
        if (e != hipSuccess) rc = fail(CCX_EHIP, "pace calibration probe failed: %s", hipGetErrorString(e));
        else if (ms > 0.0f) {
            spent_ms += ms;
            best_gbs = std::max(best_gbs, (double)bytes / ((double)ms * 1.0e6));
        }
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (rc) return rc;
    if (best_gbs > 1000.0 && best_gbs < 12000.0) {            // (a sane measurement; else keep the assumption)
        auto to_fp = [](double ns) { double v = ns / 10.0 * 256.0; return (uint32_t)(v < 1.0 ? 1.0 : (v > 4.0e9 ? 4.0e9 : v)); };
        h->pace_probe_gbs = (float)best_gbs;
        h->pace_init_fp = std::min(std::max(to_fp(h->shape.step_bytes / (best_gbs * 1.05)), h->kp.pace_min_fp), h->kp.pace_max_fp);
        h->pace_start_source = CCX_PACE_START_CALIBRATED;
    }
    return CCX_OK;
}

int run_rollout(ccx_handle* h, int K, const uint8_t* actions, const uint8_t* order, int auto_reset,
                const ccx::KOut& out, int policy = 0, uint8_t* actions_out = nullptr) {
    // The writer waves address the small output streams (rewards, flag bytes, compact rows, chosen actions) with 32-bit
    // byte offsets from the stream's base: one launch must stay below 4 GiB per stream.  Longer rollouts are cut into
    // launches on the same stream (bit-identical: an env's trajectory does not depend on how a rollout is split).
    {
        const unsigned long long per_step = (unsigned long long)h->E * (unsigned long long)h->N * 16ull;   // the widest stream
        const unsigned long long fit = 0xFFFFFFFFull / per_step;
        int max_k = (int)std::min<unsigned long long>(fit > 1 ? fit - 1 : 1, 0x7FFFFFFFull);
        if (h->tun_max_launch_steps > 0) max_k = std::min(max_k, h->tun_max_launch_steps);   // (tests: force the cut)
        // L = 6 + 4N is 2 mod 4: one step's observation slab (E x N x L floats) is a multiple of 16 bytes only when E x N is
        // even; with E x N odd every sub-launch must start on an EVEN step so that its slice of the tensor stays 16-byte
        // aligned (an odd cut failed on "obs buffer must be 16-byte aligned" after the first sub-launch had
        // already advanced the state)
        const bool odd_slab = out.obs && ((((size_t)h->E * h->N * (size_t)(6 + 4 * h->N) * sizeof(float)) & 15u) != 0);
        if (odd_slab && max_k > 1) max_k &= ~1;
        if (K > max_k && odd_slab && max_k == 1)
            return fail(CCX_EINVAL, "a rollout of an odd number of agent slots (E x N = %lld) cannot be cut into launches of "
                        "one step: its per-step observation slabs are not 16-byte aligned (raise max_launch_steps to >= 2)",
                        (long long)h->E * h->N);
        if (K > max_k) {
            const size_t EN = (size_t)h->E * h->N, L = (size_t)(6 + 4 * h->N);
            for (int k0 = 0; k0 < K; k0 += max_k) {
                const int kk = std::min(max_k, K - k0);
                ccx::KOut o = out;
                if (o.obs) o.obs += (size_t)k0 * EN * L;
                if (o.reward) o.reward += (size_t)k0 * EN;
                if (o.agent_flags) o.agent_flags += (size_t)k0 * EN;
                if (o.env_flags) o.env_flags += (size_t)k0 * (size_t)h->E;
                if (o.obs_compact) o.obs_compact += (size_t)k0 * EN * 4u;
                const int rc = run_rollout(h, kk, actions ? actions + (size_t)k0 * EN : nullptr,
                                           order ? order + (size_t)k0 * EN : nullptr, auto_reset, o, policy,
                                           actions_out ? actions_out + (size_t)k0 * EN : nullptr);
                if (rc) return rc;
            }
            return CCX_OK;
        }
    }
    if (out.obs && (reinterpret_cast<uintptr_t>(out.obs) & 15u))
        return fail(CCX_EINVAL, "obs buffer must be 16-byte aligned");
    if (out.reward && (reinterpret_cast<uintptr_t>(out.reward) & 7u))
        return fail(CCX_EINVAL, "reward buffer must be 8-byte aligned");
    if (out.obs_compact && (reinterpret_cast<uintptr_t>(out.obs_compact) & 15u))
        return fail(CCX_EINVAL, "obs_compact buffer must be 16-byte aligned");
    CCX_HIP(hipSetDevice(h->device));
    if (K <= ccx::kStepMaxK && actions && policy == 0 && !actions_out && h->step_shape.ok && h->tun_step_kernel != 0) {
        // the short-launch kernel (ccx_step.hip): CollectiveCrossingEnv.step itself, no pacing, no controller state
        if (h->check_inputs) {
            hipError_t ce = ccx::launch_check_inputs(h->stream, actions, order, (size_t)K * (size_t)h->E, h->N, h->input_errors);
            if (ce != hipSuccess) return fail(CCX_EHIP, "input check kernel launch failed: %s", hipGetErrorString(ce));
        }
        int rc = begin_timed(h);
        if (rc) return rc;
        hipError_t e = ccx::launch_step(h->step_shape, h->stream, h->kp, h->st_slab, h->cell_info, actions, order, K, auto_reset,
                                        h->pool, out, h->counters);
        if (e != hipSuccess) return fail(CCX_EHIP, "step kernel launch failed: %s", hipGetErrorString(e));
        return end_timed(h);
    }
    const bool writes_obs = out.obs != nullptr;
    bool capturing = false;
    {
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(h->stream, &cap) != hipSuccess) (void)hipGetLastError();
        else capturing = cap != hipStreamCaptureStatusNone;
    }
    {   // (re)start the pace controller: new handle, new launch shape or new setting
        // Never inside a stream capture: the memsets would become graph nodes and every replay would re-zero the
        // controller's state, and the calibration must synchronise.  Only a launch that would actually run them
        // is refused while capturing; a controller that has been started (by ANY eager paced launch of the shape) and merely
        // waits for its calibration captures fine and runs at the pace in effect (K = 16..63 on the C2 shape).
        const bool paced_launch = ccx::launch_is_paced(h->kp.pace_state != nullptr, writes_obs, K, h->kp.pace_min_k);
        const bool adaptive_launch =
            ccx::launch_is_adaptive(h->kp.pace_state != nullptr, h->kp.pace_adapt != 0u, writes_obs, K, h->kp.pace_min_k, h->kp.adapt_min_k);
        if (h->pace_dirty && !h->kp.pace_state) {
            h->pace_dirty = false;              // rollouts of this shape are not paced at all
            h->pace_needs_calibration = false;
        }
        if (h->pace_dirty && paced_launch) {    // (launches that are not paced never look at the controller: nothing to do yet)
            if (capturing)
                return fail(CCX_EINVAL, "the pace controller of this handle must be (re)started (new handle, launch shape or "
                            "setting): run one eager rollout of this shape (any paced length: >= %u steps) before capturing it "
                            "into a graph, or switch pacing off (ccx_set_step_pace(h, -1))", h->kp.pace_min_k);
            h->pace_needs_calibration = h->kp.pace_adapt != 0u && h->pace_calibrate && h->pace_start_ns <= 0.0f;
        }
        const bool calibrate_now = h->pace_needs_calibration && adaptive_launch && !capturing;
        if ((h->pace_dirty && paced_launch) || calibrate_now) {
            if (calibrate_now) {
                const size_t obs_bytes = (size_t)K * (size_t)h->E * h->N * (size_t)(6 + 4 * h->N) * sizeof(float);
                const int rc = calibrate_pace(h, out.obs, obs_bytes);
                if (rc) return rc;
                h->pace_needs_calibration = false;
            }
            CCX_HIP(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(h->pace_state), 0, 8, h->stream));   // floor, cliff memory = 0
            CCX_HIP(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(h->pace_state + h->pace_slot),
                                      (int)h->pace_init_fp, 1, h->stream));
            // A start value from the caller (ccx_set_step_pace_start: the pace a previous handle of this shape settled at)
            // is also the controller's first FLOOR: a process that already knows its pace skips the descent into a first
            // collapse; the floor decays as always.  A CALIBRATED start gets no floor: the probe is a lower bound of what
            // the paced stream reaches (measured: the paced rollouts settle 5-15 % above a plain fill of the same buffer),
            // so the controller descends from it at its fast rate (1.6 % per launch) until the memory side says stop.
            if (h->step_pace_ns == 0 && h->pace_start_ns > 0.0f)
                CCX_HIP(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(h->pace_state + 2), (int)h->pace_init_fp, 1, h->stream));
            // (a paced launch too short to calibrate on runs from the assumption; the first adaptive eager one calibrates
            // and restarts: pace_needs_calibration)
            h->pace_dirty = false;
        }
    }
    h->kp.pace_slot = h->pace_slot;
    if (h->check_inputs && actions) {
        hipError_t ce = ccx::launch_check_inputs(h->stream, actions, order, (size_t)K * (size_t)h->E, h->N,
                                                 h->input_errors);
        if (ce != hipSuccess) return fail(CCX_EHIP, "input check kernel launch failed: %s", hipGetErrorString(ce));
    }
    // A launch that is being captured into a HIP graph is replayed with these very arguments: the controller
    // cannot adapt across replays (the slot flip below happens once, at capture time), so such a launch runs
    // at the pace in effect and neither votes nor touches the controller's state.
    const bool small_launch = !writes_obs && h->tun_small_shape != 0;      // rewards / flag bytes / compact rows only
    const ccx::LaunchShape& shape = small_launch ? h->shape_small : h->shape;
    ccx::KParams kp = small_launch ? h->kp_small : h->kp;
    kp.rng_lo = h->kp.rng_lo; kp.rng_hi = h->kp.rng_hi; kp.eps_thr = h->kp.eps_thr;
    kp.pace_slot = h->pace_slot;
    kp.launch_flags = 0u;
    // launches the kernel will not pace hand steps to the writer waves through sequence words (ccx_kernels.h: one
    // definition of the launch modes for the host and the kernel)
    kp.hand_flags = ccx::launch_uses_flags(kp.pace_state != nullptr, writes_obs, K, kp.pace_min_k, h->tun_hand2) ? 1u : 0u;
    // Paced batches whose step period is close to the sim wave's own chain (C2: 2200 .. 3000 envs, 0.41-0.52 us per env-step)
    // keep the ring too: a barrier per step couples the chain to the writers and such tiles step in 0.46 us instead of 0.40
    // (2304 envs 0.77 -> 0.83 of the peak, 2816 envs 0.88 -> 0.91; from 3072 envs on there is nothing in it, and saturating
    // batches want the barrier: DESIGN 3.7; profiles/r04_small_batch.txt)
    if (h->tun_hand2 == 1 && writes_obs && !small_launch && h->ring_when_paced) kp.hand_flags = 1u;
#ifdef CCX_LAG_TRACE
    {
        static int* lag_buf = nullptr;
        if (!lag_buf) CCX_HIP(hipMalloc(&lag_buf, 16 * 4096 * sizeof(int)));
        CCX_HIP(hipMemsetAsync(lag_buf, 0x80, 16 * 4096 * sizeof(int), h->stream));
        kp.lag_trace = lag_buf;
        const int tiles_ = shape.num_blocks * shape.waves_per_block;
        kp.lag_every = tiles_ >= 16 ? tiles_ / 16 : 1;
        g_lag_buf = lag_buf;
    }
#endif
    const bool adaptive = ccx::launch_is_adaptive(kp.pace_state != nullptr, kp.pace_adapt != 0u, writes_obs, K, kp.pace_min_k, kp.adapt_min_k);
    if (adaptive && capturing) kp.pace_adapt = 0u;
    int rc = begin_timed(h);
    if (rc) return rc;
    // A grid of more workgroups than the device holds is launched ROUND BY ROUND: one launch per `resident_blocks`
    // workgroups, back to back on the stream.  In ONE launch the workgroups of the second round start as slots come free, each on
    // a schedule of its own, and the chip's write front -- one narrow window per env-step while the tiles march in step --
    // frays over several slabs.  A launch boundary between the rounds (~2 us against rounds of 100+ us) restarts every round in step.  The kernel sees
    // its place in the grid through block_base (env indices, tile phases) and knows that it is one round (launch_flags: the pace of a partial last round); the
    // controller's owner is tile 0 of the first launch and all rounds vote into the same slot, as in one launch.
    // Only where the rows of the launch exceed the ~4 GB footprint knee (a stream through more memory than that
    // sustains 5-8 % less, whatever the kernel does; profiles/r04_output_size.txt):
    // below it the boundary costs 1-2 % (32 768 envs x 64 steps: 0.887 -> 0.873), beyond it the rounds in step win 3-5 %
    // (65 536 x 64: 0.750 -> 0.785).
    // The rounds are made EQUAL (R = ceil(blocks / resident) launches of ceil(blocks / R) workgroups, each on a schedule
    // scaled to its fill): a grid just above a whole number of rounds -- 17 768 envs of C2: 1111 workgroups on 1024 slots --
    // otherwise spends a full round's time on its last few tiles (0.74-0.77 of the peak where 15 800 envs reach 0.91; balanced:
    // 0.86-0.87).  That alone is worth the boundary for a paced grid of TWO rounds whose second would be less than 30 % full
    // (17 000 .. 20 500 envs of C2: +4 .. +12 %; fuller second rounds and grids of three or more rounds: nothing or a loss --
    // profiles/r04_balanced_rounds.txt).
    const int resident = shape.resident_blocks;
    const double rows_bytes = writes_obs ? (double)K * (double)h->E * h->N * (double)(6 + 4 * h->N) * 4.0 : 0.0;
    const int n_rounds = resident > 0 ? (shape.num_blocks + resident - 1) / resident : 1;
    // ... and whose tiles can FOLLOW the scaled schedule: a tile with one throttled writer steps in ~1 us at best; where the
    // common pace is already short (5 agents: 1.3 us for a full device) half-full rounds would be asked for 0.7 us, be late,
    // vote, and drag the common pace up (N = 5, 17 776 envs: 0.52 of the peak that way; profiles/r04_cliff_scan2.txt)
    const bool thin_second_round = n_rounds == 2 && (double)(shape.num_blocks - resident) < 0.3 * resident &&
                                   shape.step_bytes / 7000.0 * (0.5 * shape.num_blocks / resident) >= 1200.0;
    const bool paced_rows = ccx::launch_is_paced(kp.pace_state != nullptr, writes_obs, K, kp.pace_min_k);
    const bool by_rounds = resident > 0 && shape.num_blocks > resident &&
                           (h->tun_round_launches >= 2 ||
                            (h->tun_round_launches == 1 && (rows_bytes > 3.5e9 || (paced_rows && thin_second_round))));
    hipError_t e = hipSuccess;
    if (by_rounds) {
        ccx::LaunchShape round = shape;
        kp.launch_flags |= ccx::CCX_K_LAUNCH_ROUND;
        const int per_round = (shape.num_blocks + n_rounds - 1) / n_rounds;
        for (int b0 = 0; b0 < shape.num_blocks && e == hipSuccess; b0 += per_round) {
            round.num_blocks = std::min(per_round, shape.num_blocks - b0);
            kp.block_base = (uint32_t)b0;
            e = ccx::launch_rollout(round, h->stream, kp, h->st, h->cell_info, actions,
                                    order, K, auto_reset, h->pool, out, h->counters, policy, actions_out);
        }
    } else {
        e = ccx::launch_rollout(shape, h->stream, kp, h->st, h->cell_info, actions,
                                order, K, auto_reset, h->pool, out, h->counters, policy, actions_out);
    }
    if (e != hipSuccess) return fail(CCX_EHIP, "rollout kernel launch failed: %s", hipGetErrorString(e));
    // the kernel collected the votes for the next pace in the other slot (same condition as in the kernel)
    if (adaptive && !capturing) h->pace_slot ^= 1u;
    return end_timed(h);
}

}  // namespace

extern "C" {

int ccx_abi_version(void) { return CCX_ABI_VERSION; }

const char* ccx_build_info(void) {
    static char buf[128];
    snprintf(buf, sizeof(buf), "libccx 0.5.0 abi %d gfx950 hip %d.%d", CCX_ABI_VERSION,
             HIP_VERSION_MAJOR, HIP_VERSION_MINOR);
    return buf;
}

const char* ccx_last_error(void) { return g_err; }

int32_t ccx_obs_len(int32_t num_agents) { return 2 + 4 + 4 * num_agents; }

int ccx_create(const ccx_params* params, int32_t num_envs, int64_t env_offset, int64_t total_envs,
               int device, void* stream, ccx_handle** out) {
    if (!out) return fail(CCX_EINVAL, "out is NULL");
    *out = nullptr;
    int rc = validate_params(params);
    if (rc) return rc;
    if (num_envs < 1) return fail(CCX_EINVAL, "num_envs must be >= 1 (got %d)", num_envs);
    if ((long long)num_envs * (params->num_boarding + params->num_exiting) >= (1ll << 28))
        return fail(CCX_EINVAL, "num_envs x agents = %lld exceeds 2^28 (per-lane byte offsets are 32-bit)",
                    (long long)num_envs * (params->num_boarding + params->num_exiting));
    if (total_envs <= 0) total_envs = num_envs;
    if (env_offset < 0 || env_offset + num_envs > total_envs)
        return fail(CCX_EINVAL, "env_offset %lld + num_envs %d exceeds total_envs %lld",
                    (long long)env_offset, num_envs, (long long)total_envs);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
        return fail(CCX_ENODEVICE, "no HIP device visible: libccx has no CPU path");
    if (device < 0 || device >= ndev) return fail(CCX_EINVAL, "device %d out of range (%d visible)", device, ndev);
    hipDeviceProp_t prop;
    CCX_HIP(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(CCX_ENODEVICE, "device %d is %s; libccx carries gfx950 (MI355X) code only", device,
                    prop.gcnArchName);
    CCX_HIP(hipSetDevice(device));

    ccx_handle* h = new (std::nothrow) ccx_handle();
    if (!h) return fail(CCX_ENOMEM, "host allocation failed");
    h->params = *params;
    h->E = num_envs;
    h->N = params->num_boarding + params->num_exiting;
    h->env_offset = env_offset;
    h->total_envs = total_envs;
    h->device = device;
    h->stream = reinterpret_cast<hipStream_t>(stream);
    const size_t en = (size_t)h->E * h->N;
    hipError_t e = hipSuccess;
    auto alloc = [&](void** p, size_t bytes) {
        if (e == hipSuccess) e = hipMalloc(p, bytes);
    };
    {   // the seven state arrays: ONE allocation (ccx_kernels.h: StateSlab), so that the short-launch kernel needs one pointer
        const ccx::StateSlab sl = ccx::state_slab(h->E, h->N);
        alloc((void**)&h->st_slab, sl.total);
        if (e == hipSuccess) {
            h->st.x = reinterpret_cast<int32_t*>(h->st_slab + sl.x);
            h->st.y = reinterpret_cast<int32_t*>(h->st_slab + sl.y);
            h->st.active = h->st_slab + sl.active;
            h->st.terminated = h->st_slab + sl.terminated;
            h->st.truncated = h->st_slab + sl.truncated;
            h->st.step_count = reinterpret_cast<int32_t*>(h->st_slab + sl.step_count);
            h->st.episode = reinterpret_cast<int32_t*>(h->st_slab + sl.episode);
        }
    }
    alloc((void**)&h->counters, ccx::counter_words(h->E) * sizeof(unsigned long long));
    alloc((void**)&h->placement_scratch, en * 2);
    alloc((void**)&h->pace_state, 8 * sizeof(uint32_t));
    alloc((void**)&h->input_errors, 2 * sizeof(unsigned long long));
    alloc((void**)&h->obs_table, kMaxObsUnits * sizeof(uint16_t));
    if (hipDeviceGetAttribute(&h->num_cus, hipDeviceAttributeMultiprocessorCount, h->device) != hipSuccess ||
        h->num_cus < 1)
        h->num_cus = 256;
    const std::vector<unsigned long long> cell_tab = build_cell_table(*params);
    alloc((void**)&h->cell_info, cell_tab.size() * sizeof(unsigned long long));
    if (e == hipSuccess)
        e = hipMemcpy(h->cell_info, cell_tab.data(), cell_tab.size() * sizeof(unsigned long long),
                      hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipEventCreate(&h->ev_start);
    if (e == hipSuccess) e = hipEventCreate(&h->ev_stop);
    if (e == hipSuccess) e = hipMemsetAsync(h->st.x, 0, en * 4, h->stream);
    if (e == hipSuccess) e = hipMemsetAsync(h->st.y, 0, en * 4, h->stream);
    if (e == hipSuccess) e = hipMemsetAsync(h->st.active, 1, en, h->stream);
    if (e == hipSuccess) e = hipMemsetAsync(h->st.terminated, 0, en, h->stream);
    if (e == hipSuccess) e = hipMemsetAsync(h->st.truncated, 0, en, h->stream);
    if (e == hipSuccess) e = hipMemsetAsync(h->st.step_count, 0, (size_t)h->E * 4, h->stream);
    if (e == hipSuccess) e = hipMemsetAsync(h->st.episode, 0, (size_t)h->E * 4, h->stream);
    if (e == hipSuccess)
        e = hipMemsetAsync(h->counters, 0, ccx::counter_words(h->E) * sizeof(unsigned long long), h->stream);
    if (e == hipSuccess) e = hipMemsetAsync(h->input_errors, 0, 2 * sizeof(unsigned long long), h->stream);
    if (e != hipSuccess) {
        int code = (e == hipErrorOutOfMemory) ? CCX_ENOMEM : CCX_EHIP;
        fail(code, "ccx_create: %s", hipGetErrorString(e));
        ccx_destroy(h);
        return code;
    }
    if (const char* ev = getenv("CCX_PACE_CALIBRATION")) h->pace_calibrate = !(ev[0] == '0' && ev[1] == 0);
    rc = choose_shape(h);
    if (rc) {
        ccx_destroy(h);
        return rc;
    }
    *out = h;
    return CCX_OK;
}

void ccx_destroy(ccx_handle* h) {
    if (!h) return;
    (void)hipSetDevice(h->device);
    (void)hipStreamSynchronize(h->stream);
    (void)hipFree(h->st_slab);
    (void)hipFree(h->counters);
    (void)hipFree(h->pace_state);
    (void)hipFree(h->input_errors);
    (void)hipFree(h->obs_table);
    (void)hipFree(h->cell_info);
    (void)hipFree(h->reward_table);
    (void)hipFree(h->placement_scratch);
    (void)hipFree(h->mt_state);
    (void)hipFree(h->stream_actions);
    (void)hipFree(h->stream_obs);
    if (h->ev_start) (void)hipEventDestroy(h->ev_start);
    if (h->ev_stop) (void)hipEventDestroy(h->ev_stop);
    delete h;
}

int32_t ccx_num_envs(const ccx_handle* h) { return h ? h->E : 0; }
int32_t ccx_num_agents(const ccx_handle* h) { return h ? h->N : 0; }

int ccx_state_view(ccx_handle* h, ccx_state* out) {
    if (!h || !out) return fail(CCX_EINVAL, "NULL argument");
    out->x = h->st.x;
    out->y = h->st.y;
    out->active = h->st.active;
    out->terminated = h->st.terminated;
    out->truncated = h->st.truncated;
    out->step_count = h->st.step_count;
    out->episode = h->st.episode;
    return CCX_OK;
}

int ccx_set_state_host(ccx_handle* h, const ccx_state* src) {
    if (!h || !src) return fail(CCX_EINVAL, "NULL argument");
    const size_t en = (size_t)h->E * h->N;
    if (src->x)
        for (size_t t = 0; t < en; ++t)
            if (src->x[t] < 0 || src->x[t] > h->params.width)
                return fail(CCX_EINVAL, "x[%zu] = %d outside 0..width=%d (collectivecrossing.py:515)", t,
                            src->x[t], h->params.width);
    if (src->y)
        for (size_t t = 0; t < en; ++t)
            if (src->y[t] < 0 || src->y[t] > h->params.height)
                return fail(CCX_EINVAL, "y[%zu] = %d outside 0..height=%d (collectivecrossing.py:515)", t,
                            src->y[t], h->params.height);
    const uint8_t* flags[3] = {src->active, src->terminated, src->truncated};
    for (const uint8_t* f : flags)
        if (f)
            for (size_t t = 0; t < en; ++t)
                if (f[t] > 1) return fail(CCX_EINVAL, "flag value %u at %zu is not 0/1", f[t], t);
    if (src->step_count)
        for (int32_t e = 0; e < h->E; ++e)
            if (src->step_count[e] < 0) return fail(CCX_EINVAL, "negative step_count");
    if (src->episode)
        for (int32_t e = 0; e < h->E; ++e)
            if (src->episode[e] < 0) return fail(CCX_EINVAL, "negative episode");
    CCX_HIP(hipSetDevice(h->device));
    CCX_HIP(hipStreamSynchronize(h->stream));
    if (src->x) CCX_HIP(hipMemcpy(h->st.x, src->x, en * 4, hipMemcpyHostToDevice));
    if (src->y) CCX_HIP(hipMemcpy(h->st.y, src->y, en * 4, hipMemcpyHostToDevice));
    if (src->active) CCX_HIP(hipMemcpy(h->st.active, src->active, en, hipMemcpyHostToDevice));
    if (src->terminated) CCX_HIP(hipMemcpy(h->st.terminated, src->terminated, en, hipMemcpyHostToDevice));
    if (src->truncated) CCX_HIP(hipMemcpy(h->st.truncated, src->truncated, en, hipMemcpyHostToDevice));
    if (src->step_count)
        CCX_HIP(hipMemcpy(h->st.step_count, src->step_count, (size_t)h->E * 4, hipMemcpyHostToDevice));
    if (src->episode) CCX_HIP(hipMemcpy(h->st.episode, src->episode, (size_t)h->E * 4, hipMemcpyHostToDevice));
    return CCX_OK;
}

int ccx_get_state_host(ccx_handle* h, ccx_state* dst) {
    if (!h || !dst) return fail(CCX_EINVAL, "NULL argument");
    const size_t en = (size_t)h->E * h->N;
    CCX_HIP(hipSetDevice(h->device));
    CCX_HIP(hipStreamSynchronize(h->stream));
    if (dst->x) CCX_HIP(hipMemcpy(dst->x, h->st.x, en * 4, hipMemcpyDeviceToHost));
    if (dst->y) CCX_HIP(hipMemcpy(dst->y, h->st.y, en * 4, hipMemcpyDeviceToHost));
    if (dst->active) CCX_HIP(hipMemcpy(dst->active, h->st.active, en, hipMemcpyDeviceToHost));
    if (dst->terminated) CCX_HIP(hipMemcpy(dst->terminated, h->st.terminated, en, hipMemcpyDeviceToHost));
    if (dst->truncated) CCX_HIP(hipMemcpy(dst->truncated, h->st.truncated, en, hipMemcpyDeviceToHost));
    if (dst->step_count)
        CCX_HIP(hipMemcpy(dst->step_count, h->st.step_count, (size_t)h->E * 4, hipMemcpyDeviceToHost));
    if (dst->episode) CCX_HIP(hipMemcpy(dst->episode, h->st.episode, (size_t)h->E * 4, hipMemcpyDeviceToHost));
    return CCX_OK;
}

int ccx_set_reset_pool(ccx_handle* h, const uint8_t* pool_xy, int64_t pool_size) {
    if (!h) return fail(CCX_EINVAL, "NULL handle");
    if ((pool_xy == nullptr) != (pool_size == 0) || pool_size < 0 || pool_size >= (1ll << 31))
        return fail(CCX_EINVAL, "bad reset pool (ptr %p, size %lld)", (const void*)pool_xy, (long long)pool_size);
    if (reinterpret_cast<uintptr_t>(pool_xy) & 1u) return fail(CCX_EINVAL, "pool must be 2-byte aligned");
    h->pool = pool_xy;
    h->pool_size = pool_size;
    return choose_shape(h);
}

int ccx_reset_from_pool(ccx_handle* h, const uint8_t* env_mask) {
    if (!h) return fail(CCX_EINVAL, "NULL handle");
    if (!h->pool || h->pool_size <= 0) return fail(CCX_EINVAL, "no reset pool set (ccx_set_reset_pool)");
    CCX_HIP(hipSetDevice(h->device));
    hipError_t e = ccx::launch_reset_from_pool(h->stream, h->kp, h->st, env_mask, h->pool);
    if (e != hipSuccess) return fail(CCX_EHIP, "reset kernel launch failed: %s", hipGetErrorString(e));
    return CCX_OK;
}

namespace {
// counters[7] collects placements that did not converge (the reference would spin forever)
int finish_placement(ccx_handle* h, const char* what) {
    unsigned long long fails = 0;
    CCX_HIP(hipStreamSynchronize(h->stream));
    CCX_HIP(hipMemcpy(&fails, h->counters + 7, sizeof(fails), hipMemcpyDeviceToHost));
    if (fails) {
        CCX_HIP(hipMemset(h->counters + 7, 0, sizeof(fails)));
        return fail(CCX_EINVAL, "%s: %llu placement(s) found no free cell within %d draws per agent "
                    "(the reference's rejection sampling would not terminate either)", what, fails, 1 << 16);
    }
    return CCX_OK;
}
}  // namespace

int ccx_fill_reset_pool_seeded(ccx_handle* h, uint8_t* pool_xy, int64_t pool_size, uint64_t seed0) {
    if (!h || !pool_xy) return fail(CCX_EINVAL, "NULL argument");
    if (pool_size < 1 || pool_size >= (1ll << 31)) return fail(CCX_EINVAL, "bad pool_size %lld", (long long)pool_size);
    CCX_HIP(hipSetDevice(h->device));
    hipError_t e = ccx::launch_seeded_placement(h->stream, h->kp, (int)pool_size, nullptr, seed0, pool_xy,
                                                h->st, nullptr, nullptr, 1 << 16, h->counters + 7);
    if (e != hipSuccess) return fail(CCX_EHIP, "placement kernel launch failed: %s", hipGetErrorString(e));
    return finish_placement(h, "ccx_fill_reset_pool_seeded");
}

int ccx_reset_seeded(ccx_handle* h, const uint64_t* seeds, const uint8_t* env_mask) {
    if (!h || !seeds) return fail(CCX_EINVAL, "NULL argument");
    CCX_HIP(hipSetDevice(h->device));
    hipError_t e = ccx::launch_seeded_placement(h->stream, h->kp, h->E, seeds, 0, nullptr, h->st, env_mask,
                                                h->placement_scratch, 1 << 16, h->counters + 7);
    if (e != hipSuccess) return fail(CCX_EHIP, "placement kernel launch failed: %s", hipGetErrorString(e));
    return finish_placement(h, "ccx_reset_seeded");
}

int ccx_policy_actions(ccx_handle* h, int32_t policy, uint8_t* actions) {
    if (!h || !actions) return fail(CCX_EINVAL, "NULL argument");
    if (policy != CCX_POLICY_GREEDY && policy != CCX_POLICY_WAITING) return fail(CCX_EINVAL, "unknown policy %d", policy);
    CCX_HIP(hipSetDevice(h->device));
    hipError_t e = h->eps_stream == CCX_EPS_STREAM_MT19937
        ? ccx::launch_policy_stream_actions(h->stream, h->kp, h->st, h->cell_info, actions, policy, h->mt_state, h->epsilon)
        : ccx::launch_greedy_actions(h->stream, h->kp, h->st, h->cell_info, actions, policy);
    if (e != hipSuccess) return fail(CCX_EHIP, "policy kernel launch failed: %s", hipGetErrorString(e));
    return CCX_OK;
}

int ccx_greedy_actions(ccx_handle* h, uint8_t* actions) {
    if (!h || !actions) return fail(CCX_EINVAL, "NULL argument");
    CCX_HIP(hipSetDevice(h->device));
    ccx::KParams kp = h->kp;
    kp.eps_thr = 0u;                               // (always the deterministic policy; ccx_policy_actions has the epsilon)
    hipError_t e = ccx::launch_greedy_actions(h->stream, kp, h->st, h->cell_info, actions, CCX_POLICY_GREEDY);
    if (e != hipSuccess) return fail(CCX_EHIP, "greedy kernel launch failed: %s", hipGetErrorString(e));
    return CCX_OK;
}

int ccx_observe(ccx_handle* h, float* obs) {
    if (!h || !obs) return fail(CCX_EINVAL, "NULL argument");
    if (reinterpret_cast<uintptr_t>(obs) & 15u) return fail(CCX_EINVAL, "obs buffer must be 16-byte aligned");
    CCX_HIP(hipSetDevice(h->device));
    hipError_t e = ccx::launch_observe(h->shape, h->stream, h->kp, h->st, obs);
    if (e != hipSuccess) return fail(CCX_EHIP, "observe kernel launch failed: %s", hipGetErrorString(e));
    return CCX_OK;
}

int ccx_expand_observations(ccx_handle* h, const float* obs_compact, int64_t rows, float* obs) {
    if (!h || !obs_compact || !obs) return fail(CCX_EINVAL, "NULL argument");
    if (rows < 0 || rows * (int64_t)h->N >= (1ll << 31)) return fail(CCX_EINVAL, "rows = %lld out of range", (long long)rows);
    if ((reinterpret_cast<uintptr_t>(obs) | reinterpret_cast<uintptr_t>(obs_compact)) & 15u)
        return fail(CCX_EINVAL, "obs and obs_compact must be 16-byte aligned");
    CCX_HIP(hipSetDevice(h->device));
    hipError_t e = ccx::launch_expand(h->shape, h->stream, h->kp, obs_compact, rows, obs);
    if (e != hipSuccess) return fail(CCX_EHIP, "expand kernel launch failed: %s", hipGetErrorString(e));
    return CCX_OK;
}

int ccx_step(ccx_handle* h, const uint8_t* actions, const uint8_t* order, const ccx_step_out* out) {
    if (!h || !actions) return fail(CCX_EINVAL, "NULL argument");
    ccx::KOut ko{};
    if (out) {
        ko.obs = out->obs;
        ko.reward = out->reward;
        ko.agent_flags = out->agent_flags;
        ko.env_flags = out->env_flags;
        ko.obs_compact = out->obs_compact;
    }
    return run_rollout(h, 1, actions, order, 0, ko);
}

int ccx_rollout(ccx_handle* h, int32_t num_steps, const uint8_t* actions, const uint8_t* order,
                int32_t auto_reset, const ccx_rollout_out* out) {
    if (!h || !actions) return fail(CCX_EINVAL, "NULL argument");
    if (num_steps < 1) return fail(CCX_EINVAL, "num_steps must be >= 1");
    if (auto_reset && (!h->pool || h->pool_size <= 0))
        return fail(CCX_EINVAL, "auto_reset needs a reset pool (ccx_set_reset_pool)");
    ccx::KOut ko{};
    if (out) {
        ko.obs = out->obs;
        ko.reward = out->reward;
        ko.agent_flags = out->agent_flags;
        ko.env_flags = out->env_flags;
        ko.obs_compact = out->obs_compact;
    }
    return run_rollout(h, num_steps, actions, order, auto_reset ? 1 : 0, ko);
}

int ccx_rollout_policy(ccx_handle* h, int32_t num_steps, int32_t policy, int32_t auto_reset,
                       const ccx_rollout_out* out, uint8_t* actions_out) {
    if (!h) return fail(CCX_EINVAL, "NULL handle");
    if (num_steps < 1) return fail(CCX_EINVAL, "num_steps must be >= 1");
    if (policy != CCX_POLICY_GREEDY && policy != CCX_POLICY_WAITING && policy != CCX_POLICY_RANDOM)
        return fail(CCX_EINVAL, "unknown policy %d", policy);
    if (auto_reset && (!h->pool || h->pool_size <= 0))
        return fail(CCX_EINVAL, "auto_reset needs a reset pool (ccx_set_reset_pool)");
    ccx::KOut ko{};
    if (out) {
        ko.obs = out->obs;
        ko.reward = out->reward;
        ko.agent_flags = out->agent_flags;
        ko.env_flags = out->env_flags;
        ko.obs_compact = out->obs_compact;
    }
    if (h->eps_stream == CCX_EPS_STREAM_MT19937 && policy != CCX_POLICY_RANDOM && h->epsilon > 0.0) {
        // The reference's stream is sequential per env: the policy runs as its own kernel between the steps (policy -> step
        // -> policy ..., each a launch on the handle's stream) instead of inside the fused one.  Meant for replaying the
        // reference's epsilon episodes action for action, not for throughput.
        CCX_HIP(hipSetDevice(h->device));
        const size_t EN = (size_t)h->E * h->N, L = (size_t)(6 + 4 * h->N);
        // (the loop's scratch buffers are allocated on first use: never inside a stream capture)
        const bool misaligned_steps = ko.obs && ((EN * L * sizeof(float)) & 15u) && num_steps > 1;
        if ((!actions_out && !h->stream_actions) || (misaligned_steps && !h->stream_obs)) {
            hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
            if (hipStreamIsCapturing(h->stream, &cap) != hipSuccess) (void)hipGetLastError();
            else if (cap != hipStreamCaptureStatusNone)
                return fail(CCX_EINVAL, "the stepwise policy loop allocates scratch buffers on first use: run one eager "
                            "policy rollout of this shape before capturing it into a graph");
            if (!actions_out && !h->stream_actions) CCX_HIP(hipMalloc(&h->stream_actions, EN));
            if (misaligned_steps && !h->stream_obs) CCX_HIP(hipMalloc(&h->stream_obs, EN * L * sizeof(float)));
        }
        for (int s = 0; s < num_steps; ++s) {
            uint8_t* acts = actions_out ? actions_out + (size_t)s * EN : h->stream_actions;
            hipError_t e = ccx::launch_policy_stream_actions(h->stream, h->kp, h->st, h->cell_info, acts, policy, h->mt_state,
                                                             h->epsilon);
            if (e != hipSuccess) return fail(CCX_EHIP, "policy kernel launch failed: %s", hipGetErrorString(e));
            ccx::KOut o = ko;
            if (o.obs) o.obs += (size_t)s * EN * L;
            if (o.reward) o.reward += (size_t)s * EN;
            if (o.agent_flags) o.agent_flags += (size_t)s * EN;
            if (o.env_flags) o.env_flags += (size_t)s * (size_t)h->E;
            if (o.obs_compact) o.obs_compact += (size_t)s * EN * 4u;
            // (a step's slice of the observation tensor starts on a 16-byte boundary only if E x N x L x 4 is a multiple of
            // 16: the others go through an aligned staging slab and a device-to-device copy)
            float* const obs_dst = o.obs;
            const bool staged = obs_dst && (reinterpret_cast<uintptr_t>(obs_dst) & 15u);
            if (staged) {
                if (!h->stream_obs) return fail(CCX_EINVAL, "obs buffer must be 16-byte aligned");   // (a misaligned BASE)
                o.obs = h->stream_obs;
            }
            const int rc = run_rollout(h, 1, acts, nullptr, auto_reset ? 1 : 0, o);
            if (rc) return rc;
            if (staged)
                CCX_HIP(hipMemcpyAsync(obs_dst, h->stream_obs, EN * L * sizeof(float), hipMemcpyDeviceToDevice, h->stream));
        }
        return CCX_OK;
    }
    // (grids whose occupancy tables exceed the LDS run the in-kernel policies through the all-pairs exchange: round 4)
    return run_rollout(h, num_steps, nullptr, nullptr, auto_reset ? 1 : 0, ko, policy, actions_out);
}

namespace {
// CCX_CHECK_INPUTS: after a stream sync, turn what check_inputs_kernel counted into CCX_EINVAL (once)
int report_input_errors(ccx_handle* h) {
    if (!h->check_inputs) return CCX_OK;
    unsigned long long bad[2] = {0, 0};
    CCX_HIP(hipMemcpy(bad, h->input_errors, sizeof(bad), hipMemcpyDeviceToHost));
    if (!bad[0] && !bad[1]) return CCX_OK;
    CCX_HIP(hipMemset(h->input_errors, 0, sizeof(bad)));
    return fail(CCX_EINVAL, "invalid inputs since the last check: Invalid action: %llu action byte(s) outside "
                "{0,1,2,3,4} and not CCX_ACTION_ABSENT; Unknown agent ID / duplicate: %llu move-order row(s) that are "
                "not a permutation of 0..N-1 (collectivecrossing.py:685-711); such entries were stepped as "
                "'no move' / an undefined order", bad[0], bad[1]);
}
}  // namespace

int ccx_set_reward_table(ccx_handle* h, const double* boarding_per_cell, const double* exiting_per_cell) {
    if (!h) return fail(CCX_EINVAL, "NULL handle");
    if ((boarding_per_cell == nullptr) != (exiting_per_cell == nullptr))
        return fail(CCX_EINVAL, "give a table for both agent types, or NULL for both (back to the built-in reward)");
    CCX_HIP(hipSetDevice(h->device));
    CCX_HIP(hipStreamSynchronize(h->stream));               // (launches in flight still read the old table)
    if (!boarding_per_cell) {
        (void)hipFree(h->reward_table);
        h->reward_table = nullptr;
        return choose_shape(h);
    }
    const int W1 = h->params.width + 1, H1 = h->params.height + 1, Wp = h->params.width + 3, Hp = h->params.height + 3;
    const size_t cells = (size_t)Wp * (size_t)Hp;
    std::vector<double> tab(2u * cells, 0.0);                // padded grid, border cells 0 (never occupied)
    const double* src[2] = {boarding_per_cell, exiting_per_cell};
    for (int t = 0; t < 2; ++t)
        for (int y = 0; y < H1; ++y)
            for (int x = 0; x < W1; ++x)
                tab[(size_t)t * cells + (size_t)(y + 1) * Wp + (size_t)(x + 1)] = src[t][(size_t)y * W1 + x];
    if (!h->reward_table) CCX_HIP(hipMalloc(&h->reward_table, tab.size() * sizeof(double)));
    CCX_HIP(hipMemcpy(h->reward_table, tab.data(), tab.size() * sizeof(double), hipMemcpyHostToDevice));
    const int rc = choose_shape(h);
    if (rc) return rc;
    if (h->shape.lds_bytes > 150u * 1024u || h->shape_small.lds_bytes > 150u * 1024u) {
        (void)hipFree(h->reward_table);
        h->reward_table = nullptr;
        (void)choose_shape(h);
        return fail(CCX_EINVAL, "the reward table of a %d x %d grid (%zu bytes of LDS) does not fit next to the kernel's tables",
                    h->params.width, h->params.height, cells * 16u);
    }
    return CCX_OK;
}

int ccx_set_terminated_table(ccx_handle* h, const uint8_t* boarding_per_cell, const uint8_t* exiting_per_cell) {
    if (!h) return fail(CCX_EINVAL, "NULL handle");
    if ((boarding_per_cell == nullptr) != (exiting_per_cell == nullptr))
        return fail(CCX_EINVAL, "give a table for both agent types, or NULL for both (back to the built-in strategy)");
    if (boarding_per_cell && h->params.terminated_mode != CCX_TERM_INDIVIDUAL_AT_DESTINATION)
        return fail(CCX_EINVAL, "a terminated table replaces individual_at_destination (terminated_mode 0): per agent, from its own cell");
    const size_t n = (size_t)(h->params.width + 1) * (size_t)(h->params.height + 1);
    const uint8_t* src[2] = {boarding_per_cell, exiting_per_cell};
    for (int t = 0; t < 2; ++t) {
        h->term_table[t].clear();
        if (src[t]) h->term_table[t].assign(src[t], src[t] + n);
    }
    CCX_HIP(hipSetDevice(h->device));
    CCX_HIP(hipStreamSynchronize(h->stream));
    const std::vector<unsigned long long> cell_tab = build_cell_table(h->params, h->term_table);
    CCX_HIP(hipMemcpy(h->cell_info, cell_tab.data(), cell_tab.size() * sizeof(unsigned long long), hipMemcpyHostToDevice));
    return choose_shape(h);        // (KParams::user_tables)
}

int ccx_set_check_inputs(ccx_handle* h, int32_t enabled) {
    if (!h) return fail(CCX_EINVAL, "NULL handle");
    h->check_inputs = enabled != 0;
    return CCX_OK;
}

int ccx_check_inputs(ccx_handle* h) {
    if (!h) return fail(CCX_EINVAL, "NULL handle");
    CCX_HIP(hipSetDevice(h->device));
    CCX_HIP(hipStreamSynchronize(h->stream));
    return report_input_errors(h);
}

int ccx_set_rng_seed(ccx_handle* h, uint64_t seed) {
    if (!h) return fail(CCX_EINVAL, "NULL handle");
    h->kp.rng_lo = (uint32_t)seed;
    h->kp.rng_hi = (uint32_t)(seed >> 32);
    return CCX_OK;
}

int ccx_set_policy_epsilon(ccx_handle* h, double epsilon) {
    if (!h) return fail(CCX_EINVAL, "NULL handle");
    if (!(epsilon >= 0.0 && epsilon <= 1.0)) return fail(CCX_EINVAL, "epsilon must be in [0, 1] (got %g)", epsilon);
    const double t = epsilon * 4294967296.0;
    h->kp.eps_thr = t >= 4294967295.0 ? 0xFFFFFFFFu : (uint32_t)t;
    h->epsilon = epsilon;
    return CCX_OK;
}

int ccx_set_policy_stream(ccx_handle* h, int32_t kind, const uint32_t* seeds, uint32_t seed) {
    if (!h) return fail(CCX_EINVAL, "NULL handle");
    if (kind != CCX_EPS_STREAM_COUNTER && kind != CCX_EPS_STREAM_MT19937)
        return fail(CCX_EINVAL, "unknown epsilon stream %d", kind);
    if (kind == CCX_EPS_STREAM_MT19937) {
        if (h->N > 64) return fail(CCX_EINVAL, "the MT19937 stream walks an env's agents with one 64-lane wave: <= 64 agents");
        CCX_HIP(hipSetDevice(h->device));
        if (!h->mt_state) CCX_HIP(hipMalloc(&h->mt_state, (size_t)h->E * ccx::kMtStateWords * sizeof(uint32_t)));
        uint32_t* seeds_dev = nullptr;
        if (seeds) {   // (host array: a handful of words per env, staged through the placement scratch's neighbour)
            CCX_HIP(hipMalloc(&seeds_dev, (size_t)h->E * sizeof(uint32_t)));
            hipError_t ce = hipMemcpyAsync(seeds_dev, seeds, (size_t)h->E * sizeof(uint32_t), hipMemcpyHostToDevice, h->stream);
            if (ce != hipSuccess) {
                (void)hipFree(seeds_dev);
                return fail(CCX_EHIP, "seed upload failed: %s", hipGetErrorString(ce));
            }
        }
        hipError_t e = ccx::launch_policy_stream_seed(h->stream, h->mt_state, seeds_dev, seed, h->E);
        hipError_t se = hipStreamSynchronize(h->stream);      // (the seeds array may be freed by the caller on return)
        if (seeds_dev) (void)hipFree(seeds_dev);
        if (e != hipSuccess || se != hipSuccess) {
            // (a half-seeded generator array must not be handed out by ccx_get_policy_stream or walked by a policy kernel)
            (void)hipFree(h->mt_state);
            h->mt_state = nullptr;
            h->eps_stream = CCX_EPS_STREAM_COUNTER;
            return fail(CCX_EHIP, "stream seeding failed: %s", hipGetErrorString(e != hipSuccess ? e : se));
        }
    }
    h->eps_stream = kind;
    return CCX_OK;
}

int ccx_get_policy_stream(ccx_handle* h, int32_t* kind, uint32_t* state) {
    if (!h) return fail(CCX_EINVAL, "NULL handle");
    if (kind) *kind = h->eps_stream;
    if (state) {
        if (!h->mt_state) return fail(CCX_EINVAL, "no MT19937 stream has been seeded on this handle (ccx_set_policy_stream)");
        CCX_HIP(hipSetDevice(h->device));
        CCX_HIP(hipStreamSynchronize(h->stream));
        CCX_HIP(hipMemcpy(state, h->mt_state, (size_t)h->E * ccx::kMtStateWords * sizeof(uint32_t), hipMemcpyDeviceToHost));
    }
    return CCX_OK;
}

int ccx_zero_counters(ccx_handle* h) {
    if (!h) return fail(CCX_EINVAL, "NULL handle");
    CCX_HIP(hipSetDevice(h->device));
    CCX_HIP(hipMemsetAsync(h->counters, 0, ccx::counter_words(h->E) * sizeof(unsigned long long), h->stream));
    return CCX_OK;
}

int ccx_read_counters(ccx_handle* h, ccx_counters* out_host) {
    if (!h || !out_host) return fail(CCX_EINVAL, "NULL argument");
    static_assert(sizeof(ccx_counters) == 6 * sizeof(unsigned long long), "counter layout");
    CCX_HIP(hipSetDevice(h->device));
    CCX_HIP(ccx::launch_reduce_counters(h->stream, h->counters, h->E));
    CCX_HIP(hipStreamSynchronize(h->stream));
    CCX_HIP(hipMemcpy(out_host, h->counters, sizeof(ccx_counters), hipMemcpyDeviceToHost));
    return report_input_errors(h);
}

int ccx_counters_device_ptr(ccx_handle* h, uint64_t** out) {
    if (!h || !out) return fail(CCX_EINVAL, "NULL argument");
    CCX_HIP(hipSetDevice(h->device));
    CCX_HIP(ccx::launch_reduce_counters(h->stream, h->counters, h->E));
    *out = reinterpret_cast<uint64_t*>(h->counters);
    return CCX_OK;
}

int ccx_set_timing(ccx_handle* h, int32_t enabled) {
    if (!h) return fail(CCX_EINVAL, "NULL handle");
    h->timing = enabled != 0;
    if (!h->timing) h->timed = false;
    return CCX_OK;
}

int ccx_last_launch_ms(ccx_handle* h, float* ms) {
    if (!h || !ms) return fail(CCX_EINVAL, "NULL argument");
    if (!h->timed) return fail(CCX_EINVAL, "no timed launch yet (enable with ccx_set_timing(h, 1))");
    CCX_HIP(hipSetDevice(h->device));
    CCX_HIP(hipEventSynchronize(h->ev_stop));
    CCX_HIP(hipEventElapsedTime(ms, h->ev_start, h->ev_stop));
    return CCX_OK;
}

int ccx_set_launch_shape(ccx_handle* h, int32_t lanes_per_wave, int32_t waves_per_block) {
    if (!h) return fail(CCX_EINVAL, "NULL handle");
    const int G = 1 << ceil_log2(h->N);
    if (lanes_per_wave != 0 && (lanes_per_wave < G || lanes_per_wave > 64 || lanes_per_wave % G))
        return fail(CCX_EINVAL, "lanes_per_wave %d must be a multiple of the env lane group %d and <= 64",
                    lanes_per_wave, G);
    if (waves_per_block < 0 || waves_per_block > 4) return fail(CCX_EINVAL, "waves_per_block must be 0..4");
    h->lanes_per_wave = lanes_per_wave;
    h->waves_per_block = waves_per_block;
    return choose_shape(h);
}

int ccx_set_writers(ccx_handle* h, int32_t writers_per_tile) {
    if (!h) return fail(CCX_EINVAL, "NULL handle");
    if (writers_per_tile < 0 || writers_per_tile > 7) return fail(CCX_EINVAL, "writers_per_tile must be 0..7");
    h->writers = writers_per_tile;
    return choose_shape(h);
}

int ccx_set_store_throttle(ccx_handle* h, int32_t max_stores_in_flight) {
    if (!h) return fail(CCX_EINVAL, "NULL handle");
    if (max_stores_in_flight < -1 || max_stores_in_flight > 63)
        return fail(CCX_EINVAL, "max_stores_in_flight must be -1 (off), 0 (default) or 1..63");
    h->store_throttle = max_stores_in_flight;
    return choose_shape(h);
}

int ccx_set_step_pace(ccx_handle* h, int32_t ns_per_env_step) {
    if (!h) return fail(CCX_EINVAL, "NULL handle");
    if (ns_per_env_step < -1) return fail(CCX_EINVAL, "ns_per_env_step must be -1 (off), 0 (adaptive) or > 0");
    h->step_pace_ns = ns_per_env_step;
    return choose_shape(h);
}

int ccx_set_step_pace_start(ccx_handle* h, float ns_per_env_step) {
    if (!h) return fail(CCX_EINVAL, "NULL handle");
    if (!(ns_per_env_step >= 0.0f) || ns_per_env_step > 1.0e7f)
        return fail(CCX_EINVAL, "ns_per_env_step must be 0 (library default) or a positive number of nanoseconds");
    h->pace_start_ns = ns_per_env_step;
    return choose_shape(h);
}

int ccx_set_pace_calibration(ccx_handle* h, int32_t enabled) {
    if (!h) return fail(CCX_EINVAL, "NULL handle");
    h->pace_calibrate = enabled != 0;
    return choose_shape(h);
}

int ccx_get_pace_start(ccx_handle* h, float* ns_per_env_step, int32_t* source, float* probe_gbs) {
    if (!h) return fail(CCX_EINVAL, "NULL handle");
    if (ns_per_env_step) *ns_per_env_step = h->kp.pace_state ? (float)((double)h->pace_init_fp / 256.0 * 10.0) : 0.0f;
    if (source) *source = h->kp.pace_state ? h->pace_start_source : CCX_PACE_START_UNPACED;
    if (probe_gbs) *probe_gbs = h->pace_probe_gbs;
    return CCX_OK;
}

int ccx_set_tunable(ccx_handle* h, const char* name, int32_t value) {
    if (!h || !name) return fail(CCX_EINVAL, "NULL argument");
    struct { const char* name; int* slot; int lo, hi; } table[] = {
        {"pace_phase", &h->tun_pace_phase, -1, 3},
        {"tile_map", &h->tun_tile_map, -1, 6},
        {"hand2", &h->tun_hand2, 0, 2},
        {"writer_roles", &h->tun_writer_roles, -1, 1},
        {"max_launch_steps", &h->tun_max_launch_steps, 0, 0x7FFFFFFF},
        {"pair_rows", &h->tun_pair_rows, -1, 1},
        {"small_shape", &h->tun_small_shape, 0, 1},
        {"round_launches", &h->tun_round_launches, 0, 2},
        {"occ_tables", &h->tun_occ_tables, -1, 1},
        {"step_kernel", &h->tun_step_kernel, -1, 1},
        {"step_rows", &h->tun_step_rows, 0, 7},
        {"step_lanes", &h->tun_step_lanes, 0, 64},
    };
    for (auto& t : table)
        if (strcmp(name, t.name) == 0) {
            if (value < t.lo || value > t.hi)
                return fail(CCX_EINVAL, "tunable %s must be %d..%d (got %d)", name, t.lo, t.hi, value);
            *t.slot = value;
            return choose_shape(h);
        }
    return fail(CCX_EINVAL, "unknown tunable '%s' (pace_phase, tile_map, hand2, writer_roles, max_launch_steps, pair_rows, small_shape, round_launches, occ_tables, step_kernel, step_rows, step_lanes)", name);
}

int ccx_get_step_pace(ccx_handle* h, float* ns_per_env_step) {
    if (!h || !ns_per_env_step) return fail(CCX_EINVAL, "NULL argument");
    CCX_HIP(hipSetDevice(h->device));
    uint32_t fp = h->pace_init_fp;
    if (!h->kp.pace_state) {   // off, or adaptive on a batch too small to be memory-bound
        fp = 0;
    } else if (!h->pace_dirty) {
        CCX_HIP(hipStreamSynchronize(h->stream));
        uint32_t st[4];
        CCX_HIP(hipMemcpy(st, h->pace_state, sizeof(st), hipMemcpyDeviceToHost));
        fp = st[h->pace_slot] > st[2] ? st[h->pace_slot] : st[2];
    }
    *ns_per_env_step = (float)((double)fp / 256.0 * 10.0);
    return CCX_OK;
}

int ccx_get_pace_state(ccx_handle* h, float* out6) {
    if (!h || !out6) return fail(CCX_EINVAL, "NULL argument");
    CCX_HIP(hipSetDevice(h->device));
    CCX_HIP(hipStreamSynchronize(h->stream));
    uint32_t st[8];
    CCX_HIP(hipMemcpy(st, h->pace_state, sizeof(st), hipMemcpyDeviceToHost));
    auto ns = [](uint32_t fp) { return (float)((double)fp / 256.0 * 10.0); };
    out6[0] = h->pace_dirty ? ns(h->pace_init_fp) : ns(st[h->pace_slot]);   // the vote the next launch reads
    out6[1] = h->pace_dirty ? 0.0f : ns(st[2]);                             // floor
    out6[2] = h->pace_dirty ? 0.0f : (float)st[3];                          // launches since the last collapse
    out6[3] = h->kp.pace_state ? 1.0f : 0.0f;
    out6[4] = h->pace_dirty ? 0.0f : ns(st[4]);                             // pace of the last collapse (the cliff)
    out6[5] = h->pace_dirty ? 0.0f : (float)st[5];                          // confirmations of that cliff
    return CCX_OK;
}

int ccx_get_residency(ccx_handle* h, int32_t* resident_workgroups, int32_t* workgroups) {
    if (!h) return fail(CCX_EINVAL, "NULL handle");
    if (resident_workgroups) *resident_workgroups = h->shape.resident_blocks;
    if (workgroups) *workgroups = h->shape.num_blocks;
    return CCX_OK;
}

int ccx_get_launch_shape(ccx_handle* h, int32_t* lanes_per_wave, int32_t* waves_per_block,
                         int32_t* group_lanes, int32_t* num_blocks) {
    if (!h) return fail(CCX_EINVAL, "NULL handle");
    if (lanes_per_wave) *lanes_per_wave = h->shape.envs_per_wave << h->shape.glog;
    if (waves_per_block) *waves_per_block = h->shape.waves_per_block;
    if (group_lanes) *group_lanes = 1 << h->shape.glog;
    if (num_blocks) *num_blocks = h->shape.num_blocks;
    return CCX_OK;
}

int ccx_get_step_shape(ccx_handle* h, int32_t* ok, int32_t* lanes_per_wave, int32_t* row_waves,
                       int32_t* num_blocks, int32_t* lds_bytes) {
    if (!h) return fail(CCX_EINVAL, "NULL handle");
    const ccx::StepShape& ss = h->step_shape;
    if (ok) *ok = (ss.ok && h->tun_step_kernel != 0) ? 1 : 0;
    if (lanes_per_wave) *lanes_per_wave = ss.envs_per_wave << ss.glog;
    if (row_waves) *row_waves = ss.row_waves;
    if (num_blocks) *num_blocks = ss.num_blocks;
    if (lds_bytes) *lds_bytes = (int32_t)ss.lds_bytes;
    return CCX_OK;
}

int ccx_get_writer_shape(ccx_handle* h, int32_t* writers_per_tile, int32_t* max_stores_in_flight) {
    if (!h) return fail(CCX_EINVAL, "NULL handle");
    if (writers_per_tile) *writers_per_tile = h->shape.writers;
    if (max_stores_in_flight) *max_stores_in_flight = h->shape.store_throttle;
    return CCX_OK;
}

int ccx_host_device_pointer(ccx_handle* h, void* pinned_host, void** device_ptr) {
    if (!h || !pinned_host || !device_ptr) return fail(CCX_EINVAL, "NULL argument");
    CCX_HIP(hipSetDevice(h->device));
    hipPointerAttribute_t attr{};
    hipError_t e = hipPointerGetAttributes(&attr, pinned_host);
    if (e != hipSuccess || attr.type != hipMemoryTypeHost) {
        (void)hipGetLastError();
        return fail(CCX_EINVAL, "pointer is not page-locked host memory registered with HIP");
    }
    void* d = nullptr;
    CCX_HIP(hipHostGetDevicePointer(&d, pinned_host, 0));
    *device_ptr = d;
    return CCX_OK;
}

int ccx_set_stream(ccx_handle* h, void* stream) {
    if (!h) return fail(CCX_EINVAL, "NULL handle");
    CCX_HIP(hipSetDevice(h->device));
    CCX_HIP(hipStreamSynchronize(h->stream));   // work queued on the old stream finishes first
    h->stream = reinterpret_cast<hipStream_t>(stream);
    return CCX_OK;
}

int ccx_synchronize(ccx_handle* h) {
    if (!h) return fail(CCX_EINVAL, "NULL handle");
    CCX_HIP(hipSetDevice(h->device));
    CCX_HIP(hipStreamSynchronize(h->stream));
    return report_input_errors(h);
}

#ifdef CCX_LAG_TRACE
int ccx_debug_lag_trace(int* host) {   // diagnostic build only (make variant DEFS=-DCCX_LAG_TRACE): [16][4096] ints of the last launch
    if (!g_lag_buf) return -1;
    return (int)hipMemcpy(host, g_lag_buf, 16 * 4096 * sizeof(int), hipMemcpyDeviceToHost);
}
#endif
}  // extern "C"
