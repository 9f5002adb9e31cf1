// ccx_kernels.h -- types shared by the kernel translation unit and the C-ABI layer (internal).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ccx {

enum : uint32_t { CCX_K_ABSENT = 255u };
enum : int { CCX_K_REWARD_DEFAULT = 0, CCX_K_REWARD_SIMPLE_DISTANCE = 1, CCX_K_REWARD_BINARY = 2,
             CCX_K_REWARD_CONSTANT_NEGATIVE = 3 };
enum : int { CCX_K_TERM_INDIVIDUAL = 0, CCX_K_TERM_ALL = 1 };
enum : int { CCX_K_POLICY_GREEDY = 1, CCX_K_POLICY_WAITING = 2, CCX_K_POLICY_RANDOM = 3 };

// Counter-based random word of agent `agent` of global env `genv` at step `step` (0-based) of its episode
// `episode` (lowbias32 rounds, restated by the CPU checker).  CCX_POLICY_RANDOM: action = word * 5 >> 32, uniform
// over 0..4 up to 2^-32.
__host__ __device__ inline uint32_t mix32(uint32_t k) {
    k ^= k >> 16; k *= 0x7FEB352Du; k ^= k >> 15; k *= 0x846CA68Bu; k ^= k >> 16;
    return k;
}
__host__ __device__ inline uint32_t random_word(uint32_t seed_lo, uint32_t seed_hi, uint32_t genv, uint32_t episode,
                                                uint32_t step, uint32_t agent) {
    uint32_t k = genv * 0x9E3779B1u + episode * 0x85EBCA77u + step * 0xC2B2AE3Du + agent * 0x27D4EB2Fu + seed_lo;
    return mix32(mix32(k) ^ seed_hi);
}
__host__ __device__ inline uint32_t random_action(uint32_t seed_lo, uint32_t seed_hi, uint32_t genv, uint32_t episode,
                                                  uint32_t step, uint32_t agent) {
    return (uint32_t)(((unsigned long long)random_word(seed_lo, seed_hi, genv, episode, step, agent) * 5ull) >> 32);
}
// Epsilon-greedy (greedy_policy.py:48-59, waiting_policy.py:48-59): the reference draws random() < epsilon and
// then choice(valid_actions) from ONE RandomState shared by all agents of an env (a sequential stream); here the
// two draws come from the agent's own counter-based word u (seed_hi ^ kEpsStream): explore iff u < epsilon * 2^32,
// and the exploring action is the k-th (ascending) of the valid actions -- the free directions plus wait --
// with k = mix32(u + 0x9E3779B9) * count >> 32.
enum : uint32_t { kEpsStream = 0x5BD1E995u };
__host__ __device__ inline uint32_t explore_action(uint32_t u, uint32_t free_dirs) {
    uint32_t vm = (free_dirs & 0xFu) | 0x10u;          // wait is always valid (greedy_policy.py:251)
#if defined(__HIP_DEVICE_COMPILE__)
    const uint32_t count = (uint32_t)__popc(vm);
#else
    const uint32_t count = (uint32_t)__builtin_popcount(vm);
#endif
    const uint32_t k = (uint32_t)(((unsigned long long)mix32(u + 0x9E3779B9u) * count) >> 32);
    for (uint32_t j = 0; j < 4u; ++j)
        if (j < k) vm &= vm - 1u;                       // drop the k lowest valid actions
#if defined(__HIP_DEVICE_COMPILE__)
    return (uint32_t)(__ffs((int)vm) - 1);
#else
    return (uint32_t)__builtin_ctz(vm);
#endif
}
// Per-tile LDS head (byte offsets from the tile's base): [xch u32 x 64][hand-off words u32 x 16][stage ring].
// Hand-off ring between the sim wave and the writer waves of a tile: slots of 64 words, one per lane (kHw* below).  In flags mode
// (KParams::hand_flags) word kSyncSeq counts the env-steps the sim wave has staged and word kSyncProg + w the env-steps
// writer w has read; neither side ever meets the other at a barrier.
// The ring has KParams::stage_slots slots (32, or fewer where 32 would cost a workgroup per CU its LDS).
constexpr uint32_t kMaxStageSlots = 32, kStageSlotBytes = 256u;
// hand-off word of one agent and step: bits 0-16 LDS address of the agent's cell word (< 2^17: 103 x 103 cells x 8 bytes
// behind at most a few KB), 17-18 the flags the step raises (terminated, truncated), 19 active, 20-22 the env byte
// (CCX_EF_*), 23-30 the action taken (CCX_ACTION_ABSENT = 255)
constexpr uint32_t kHwCellMask = 0x1FFFFu, kHwOut2Shift = 17, kHwActShift = 19, kHwEfShift = 20, kHwActionShift = 23;
constexpr uint32_t kSyncOff = 256u, kSyncBytes = 64u, kStageOff = kSyncOff + kSyncBytes;
constexpr uint32_t tile_head_bytes(uint32_t stage_slots) { return kStageOff + stage_slots * kStageSlotBytes; }
constexpr uint32_t kSyncSeq = 0u, kSyncProg = 4u;          // (kSyncProg + writer index, at most 7 writers)
// bits of the low word of a cell's geometry entry (ccx_kernels.hip: per-cell geometry table); bit 4 is always 0
constexpr uint32_t kCellInTram = 0x20u, kCellAtDoor = 0x40u;
// bits 11 / 15: terminateds[id] of a boarding / exiting agent on this cell (terminateds.py:66-82: = the destination-row
// bits 8 / 12 for the built-in strategies; ccx_set_terminated_table overrides them)
constexpr uint32_t kCellTermShift = 3u;    // (relative to the type's destination bit)

// ONE definition of how a launch is driven, shared by the host (run_rollout) and the kernel: a launch is PACED when the
// handle paces its shape, it writes observation rows and is long enough to be worth the clock reads; the pace controller
// ADAPTS (votes, slot flip on the host) only in paced launches of at least 64 steps; launches that are not paced hand
// steps from the sim wave to the writer waves through sequence words instead of a barrier per step (tunable "hand2").
// The two step counts are per launch SHAPE (KParams::pace_min_k / adapt_min_k, set by the host: a launch must last ~12 us
// to be worth pacing and ~50 us for its lateness to be judged -- 16 / 64 steps of C2's 0.78-us steps, 2 / 5 of C3's 10-us
// ones; rounds 1-2 used 16 / 64 for every shape, which left short rollouts of large tiles at their start pace).
__host__ __device__ inline bool launch_is_paced(bool handle_paces, bool writes_obs, int K, uint32_t pace_min_k) {
    return handle_paces && writes_obs && K >= (int)pace_min_k;
}
__host__ __device__ inline bool launch_is_adaptive(bool handle_paces, bool handle_adapts, bool writes_obs, int K,
                                                   uint32_t pace_min_k, uint32_t adapt_min_k) {
    return launch_is_paced(handle_paces, writes_obs, K, pace_min_k) && handle_adapts && K >= (int)adapt_min_k;
}
__host__ __device__ inline bool launch_uses_flags(bool handle_paces, bool writes_obs, int K, uint32_t pace_min_k, int tunable) {
    return tunable >= 2 || (tunable == 1 && !launch_is_paced(handle_paces, writes_obs, K, pace_min_k));   // 0 = never, 1 = unpaced launches, 2 = always
}
enum : uint32_t { CCX_K_LAUNCH_ROUND = 1u };
enum : uint32_t { CCX_K_EF_ALL_TERM = 1u, CCX_K_EF_ALL_TRUNC = 2u, CCX_K_EF_RESET = 4u };

// kernel parameters (passed by value; lives in SGPRs / the kernarg segment)
struct KParams {
    int W, H, div, tl, tr, dl, dr, dc;   // grid + absolute tram/door columns, door centre
    int Nb, N, bdy, edy;                 // boarding count, agents per env, destination rows
    int reward_mode, term_mode, max_steps;
    int E;                               // envs of this handle
    int EW;                              // envs carried by one wavefront
    int waves_per_block;
    int units_per_wave;                  // EW * N * (3 + 2N) float2 units of observation per wave
    int writers;                         // writer waves per tile (rollout kernel with outputs)
    // LDS carve-up of the rollout kernel (bytes): cell table at 0, tiles, obs table
    uint32_t off_tiles, tile_stride;     // first tile, distance between tiles of a block
    uint32_t off_ws, off_occ;            // within a tile: WSlot array, occupancy/proposal tables
    uint32_t occ_words;                  // 32-bit words of one tile's occupancy+proposal tables
    uint32_t off_table;                  // u16 obs address table
    uint32_t pace_adapt;                 // 1: the kernel retunes *pace_state after every long launch
    uint32_t writer_vmcnt;               // >0: a writer starts a step only with <= this many of its stores in flight
    // step pacing (rollouts that write observations): every tile starts env-step s no earlier than
    // t0 + s * pace on the 100 MHz s_memrealtime clock; pace = *pace_state in ticks x 256 (0 = off)
    uint32_t* pace_state;                // [pace_slot] is read, [pace_slot ^ 1] collects the votes, [2] = floor, [3] launches since the last collapse, [4] pace of the last collapse, [5] confirmations, [6] collapse mark of the running launch, [7] adaptive launches
    uint32_t pace_min_fp, pace_max_fp;
    uint32_t pace_slot;
    uint32_t resident_blocks;            // workgroups the device holds at once (0 = unknown)
    // u16 LDS source address of every float2 unit of a tile's observation region, built on the host
    // for this (glog, EW, N) and copied to LDS at kernel start (units_per_wave + 1 entries)
    const uint16_t* obs_table;
    double r_dest, r_door, r_area, r_f, r_nogoal, r_pen;
    long long env_offset;                // global index of env 0 (sharding)
    long long pool_size;                 // reset-pool entries (0 = none)
    long long pool_stride;               // cursor stride: total_envs mod pool_size, 1 when that is 0
    // tunables (ccx_set_tunable): how the tiles' step schedules are phased, how workgroups map to tiles
    uint32_t pace_phase, tile_map;
    uint32_t wp_magic;                   // ceil(2^32 / (W + 3)): cell index / row length by one multiply-high (exact below 2^17)
    uint32_t stage_slots;                // slots of the sim -> writer hand-off ring (a power of two: 16 or 32, ccx_api.hip: choose_shape)
    uint32_t hand_flags;                 // 1: sequence-word hand-off between sim and writer waves (unpaced launches), 0: one barrier per step
    uint32_t writer0_small;              // 1: writer 0 writes the small outputs only, writers 1.. the observation rows
    uint32_t rng_lo, rng_hi;             // seed of CCX_POLICY_RANDOM and of the epsilon draws (ccx_set_rng_seed)
    uint32_t eps_thr;                    // epsilon * 2^32 of the scripted policies (ccx_set_policy_epsilon), 0 = greedy
    uint32_t ws_per_writer;              // staging slots (WSlot) per writer wave: 2 where row writers may take two steps per iteration
    uint32_t pace_min_k, adapt_min_k;    // a launch is paced from pace_min_k steps on, the controller adapts from adapt_min_k on
    // user reward table (ccx_set_reward_table: position-only RewardFunction plugins, rewards.py:16-38): f64 [2][cells of the
    // padded grid] (boarding, exiting), staged in LDS at off_rtab behind the cell table; off_rtab = 0: the built-in classes
    const double* reward_table;
    uint32_t off_rtab;
    uint32_t user_tables;                // 1: a user reward / terminated table is set: launch the instantiations that are not PLAIN
    // A grid of more workgroups than the device holds is launched ROUND BY ROUND (ccx_api.hip: run_rollout): this launch
    // carries workgroups block_base .. block_base + gridDim.x - 1 of the grid (launch_flags bit 0: it is one round of a larger grid).
    uint32_t block_base, launch_flags;
#ifdef CCX_LAG_TRACE
    int* lag_trace;                      // diagnostic build: [16 traced tiles][4096 steps] lag behind the schedule, 10-ns ticks
    int lag_every;                       // every lag_every-th tile is traced
#endif
};

struct KState {
    int32_t* x;
    int32_t* y;
    uint8_t* active;
    uint8_t* terminated;
    uint8_t* truncated;
    int32_t* step_count;
    int32_t* episode;
};

// The handle's state arrays live in ONE device allocation (ccx_api.hip: ccx_create): the short-launch kernel gets one base
// pointer (a preloaded kernel argument) and derives the seven arrays from E and N.
struct StateSlab { size_t x, y, active, terminated, truncated, step_count, episode, total; };
__host__ __device__ inline StateSlab state_slab(int E, int N) {
    const size_t en = ((size_t)E * (size_t)N + 63u) & ~(size_t)63u, e = ((size_t)E + 63u) & ~(size_t)63u;
    StateSlab s;
    s.x = 0; s.y = 4u * en; s.active = 8u * en; s.terminated = 9u * en; s.truncated = 10u * en;
    s.step_count = 11u * en; s.episode = 11u * en + 4u * e; s.total = 11u * en + 8u * e;
    return s;
}

struct KOut {
    float* obs;
    double* reward;
    uint8_t* agent_flags;
    uint8_t* env_flags;
    float* obs_compact;   // [..][E][N][4]: (x, y, type, active) of every agent slot, once per env and step
};

// LDS tile a writer wave gathers observation rows from: float4 (x, y, type, active) per lane, then
// 8 floats of constants.  obs_unit_addr: byte offset (from that tile) of the 8 bytes that go to float2
// unit `w` of a tile's observation region.  Row layout (observations.py:64-92): unit 0 = (x_i, y_i);
// unit 1 = (door_centre, division_y); unit 2 = (door_left, door_right); unit 3+2j = (x_j, y_j),
// unit 4+2j = (type_j, active_j), or (-1, -1) for j == i.
constexpr uint32_t kObsCstOff = 1024;
__host__ __device__ inline uint16_t obs_unit_addr(uint32_t w, int N, int glog) {
    const uint32_t U = 3u + 2u * (uint32_t)N;
    const uint32_t row = w / U, u = w - row * U;
    const uint32_t el = row / (uint32_t)N, i = row - el * (uint32_t)N;
    const uint32_t gb = el << glog;
    if (u == 0) return (uint16_t)((gb + i) * 16u);
    if (u < 3) return (uint16_t)(kObsCstOff + (u - 1u) * 8u);
    const uint32_t j = (u - 3u) >> 1, hh = (u - 3u) & 1u;
    if (j == i) return (uint16_t)(kObsCstOff + 16u);
    return (uint16_t)((gb + j) * 16u + hh * 8u);
}

// Device counters: [0..5] totals (ccx_counters), [7] failed placements, [8..15] diagnostic builds, then
// one slot of kCounterSlot words per env tile that the rollout kernel adds to (a tile index never
// exceeds the env count); launch_reduce_counters sums the slots into the totals.
constexpr unsigned kCounterTotals = 16, kCounterSlot = 8;
inline size_t counter_words(int num_envs) { return kCounterTotals + (size_t)kCounterSlot * (size_t)num_envs; }

struct LaunchShape {
    int glog;             // log2 of the per-env lane group
    int envs_per_wave;    // EW
    int waves_per_block;  // env tiles per block (each: 1 sim wave + `writers` writer waves)
    int writers;
    int store_throttle;   // max stores a writer keeps in flight when it starts a step (0 = unlimited)
    int resident_blocks;  // workgroups of the rollout kernel the device holds at once (<= num_blocks)
    double step_bytes;    // bytes the resident workgroups write per env-step
    int occ;              // 1: occupancy-table conflict masks fit in LDS
    int num_blocks;
    size_t lds_bytes;        // rollout kernel: cell table + per-wave tiles + obs table
    size_t lds_bytes_observe;  // observe kernel: per-wave tiles + obs table
};

// launch shape of the short-launch kernel (ccx_step.hip): a workgroup = one tile = a sim wave + row_waves row waves
struct StepShape {
    int ok;               // 0: this handle's short launches take the rollout kernel (tables too large for LDS)
    int glog, envs_per_wave, row_waves, num_blocks;
    size_t lds_bytes;
};
constexpr int kStepMaxK = 16;   // env-steps per launch of the short-launch kernel (one burst of action loads)
size_t step_lds_bytes(int glog, int ew, int N, int cells, bool reward_table);
hipError_t launch_step(const StepShape& ss, hipStream_t stream, const KParams& p, uint8_t* st_base,
                       const unsigned long long* cell_info, const uint8_t* actions, const uint8_t* order, int K, int auto_reset,
                       const uint8_t* pool, const KOut& out, unsigned long long* counters);

hipError_t launch_rollout(const LaunchShape& ls, hipStream_t stream, const KParams& p,
                          const KState& st, const unsigned long long* cell_info,
                          const uint8_t* actions, const uint8_t* order, int K,
                          int auto_reset, const uint8_t* pool, const KOut& out,
                          unsigned long long* counters, int policy = 0, uint8_t* actions_out = nullptr);
int rollout_blocks_per_cu(const LaunchShape& ls, int agents);
hipError_t launch_reduce_counters(hipStream_t stream, unsigned long long* counters, int slots);
hipError_t launch_observe(const LaunchShape& ls, hipStream_t stream, const KParams& p,
                          const KState& st, float* obs);
// DefaultObservation rows [rows][N][L] from compact rows [rows][N][4] (rows = envs x steps)
hipError_t launch_expand(const LaunchShape& ls, hipStream_t stream, const KParams& p, const float* compact,
                         long long rows, float* obs);
hipError_t launch_reset_from_pool(hipStream_t stream, const KParams& p, const KState& st,
                                  const uint8_t* env_mask, const uint8_t* pool);
// streams `bytes` (a multiple of 16) of filler into `dst` with the rollout's own store instruction: the pace calibration probe
hipError_t launch_write_probe(hipStream_t stream, void* dst, size_t bytes, int blocks);

// CCX_CHECK_INPUTS: counts action bytes outside {0..4, 255} into bad[0] and move-order rows that are not a
// permutation of 0..N-1 into bad[1] (collectivecrossing.py:685-711), one thread per (step, env) row
hipError_t launch_check_inputs(hipStream_t stream, const uint8_t* actions, const uint8_t* order, size_t rows,
                               int N, unsigned long long* bad);
hipError_t launch_greedy_actions(hipStream_t stream, const KParams& p, const KState& st,
                                 const unsigned long long* cell_info, uint8_t* actions, int policy);
// the reference's own epsilon stream (ccx_policy.hip): per env numpy RandomState = MT19937 key[624] + pos, u32 [E][625]
constexpr int kMtStateWords = 625;
hipError_t launch_policy_stream_seed(hipStream_t stream, uint32_t* mt_state, const uint32_t* seeds_dev, uint32_t seed_all,
                                     int E);
hipError_t launch_policy_stream_actions(hipStream_t stream, const KParams& p, const KState& st,
                                        const unsigned long long* cell_info, uint8_t* actions, int policy,
                                        uint32_t* mt_state, double epsilon);
hipError_t launch_seeded_placement(hipStream_t stream, const KParams& p, int n, const uint64_t* seeds,
                                   uint64_t seed0, uint8_t* pool_out, const KState& st,
                                   const uint8_t* env_mask, uint8_t* scratch_xy, int max_tries,
                                   unsigned long long* fail_counter);

}  // namespace ccx
