// ccx_internal.h -- what the translation units of the C-ABI layer share (not installed, not part of the ABI).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <vector>

#include "../../include/ccx.h"
#include "ccx_kernels.h"

namespace ccxi {

// records the thread-local message returned by ccx_last_error() and hands the code back
int fail(int code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));

}  // namespace ccxi

#define CCX_HIP(call)                                                                         \
    do {                                                                                      \
        hipError_t e_ = (call);                                                               \
        if (e_ != hipSuccess)                                                                 \
            return ccxi::fail(CCX_EHIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), \
                              __FILE__, __LINE__);                                            \
    } while (0)

struct ccx_handle {
    ccx_params params{};
    int32_t E = 0, N = 0;
    int64_t env_offset = 0, total_envs = 0;
    int device = 0;
    hipStream_t stream = nullptr;
    ccx::KState st{};
    uint8_t* st_slab = nullptr;              // ONE allocation behind the seven state arrays (ccx_kernels.h: StateSlab)
    unsigned long long* cell_info = nullptr; // per-cell geometry table (see ccx_kernels.hip: CellInfo)
    double* reward_table = nullptr;          // device f64 [2][cells of the padded grid]: user reward table (ccx_set_reward_table), null = built-in
    std::vector<uint8_t> term_table[2];      // host u8 [(H+1)(W+1)] per agent type: user terminated table (ccx_set_terminated_table), empty = built-in
    uint8_t* placement_scratch = nullptr;    // u8 [E][N][2] work area of ccx_reset_seeded
    unsigned long long* counters = nullptr;  // 6 x u64 (+ 10 spare words used by diagnostic builds)
    const uint8_t* pool = nullptr;
    int64_t pool_size = 0;
    hipEvent_t ev_start = nullptr, ev_stop = nullptr;
    bool timed = false;
    bool timing = false;                                       // record HIP events around launches
    int lanes_per_wave = 0, waves_per_block = 0, writers = 0;  // user overrides (0 = default)
    int store_throttle = 0;                                    // 0 = default, -1 = off, >0 = stores in flight
    int step_pace_ns = 0;                                      // 0 = adaptive, -1 = off, >0 = fixed ns per env-step
    uint32_t* pace_state = nullptr;                            // device: current pace (ticks x 256)
    uint32_t pace_init_fp = 0;                                 // value to (re)start the controller from
    bool pace_dirty = true;                                    // pace_state must be rewritten before a launch
    bool pace_needs_calibration = false;                       // the controller runs from the assumed start value: the first ADAPTIVE eager launch calibrates and restarts it
    uint16_t* obs_table = nullptr;                             // device: obs address table of the current shape
    std::vector<uint16_t> obs_table_host;
    uint32_t pace_slot = 0;                                    // slot of pace_state the next launch reads
    int num_cus = 256;
    float pace_start_ns = 0.0f;                                // > 0: the adaptive controller starts here (ccx_set_step_pace_start)
    bool pace_calibrate = true;                                // measure the start value in-process (ccx_set_pace_calibration)
    int pace_start_source = 0;                                 // CCX_PACE_START_*: where the controller's start value came from
    float pace_probe_gbs = 0.0f;                               // write rate the calibration probe measured (GB/s), 0 = not run
    int tun_pace_phase = -1, tun_tile_map = -1;                 // -1 = the library's choice for the launch shape
    int tun_writer_roles = -1;                                  // -1 = by batch size, 0 = writers share everything, 1 = writer 0 small outputs only
    int tun_hand2 = 1;                                          // sim -> writer hand-off: 0 barrier per step, 1 sequence words in unpaced launches, 2 always
    int tun_pair_rows = -1;                                     // -1 / 1: row writers of small batches may take two steps per iteration (a second staging slot each), 0: never
    int tun_max_launch_steps = 0;                               // > 0: cut rollouts into launches of at most this many steps
    double epsilon = 0.0;                                      // ccx_set_policy_epsilon, as given (the MT19937 stream compares doubles)
    int eps_stream = 0;                                        // CCX_EPS_STREAM_*: where the policies' exploration draws come from
    uint32_t* mt_state = nullptr;                              // device u32 [E][625]: one numpy RandomState per env (CCX_EPS_STREAM_MT19937)
    uint8_t* stream_actions = nullptr;                         // device u8 [E][N]: one step's actions of the unfused policy loop
    float* stream_obs = nullptr;                               // device f32 [E][N][L]: aligned staging slab of that loop
    bool check_inputs = false;                                 // ccx_set_check_inputs
    unsigned long long* input_errors = nullptr;                // device [2]: bad action bytes, bad order rows
    int tun_step_kernel = -1;                                   // launches of <= 16 steps: -1 / 1 the short-launch kernel (ccx_step.hip) where it applies, 0 always the rollout kernel
    int tun_step_rows = 0, tun_step_lanes = 0;                  // short-launch kernel: row waves per tile (0 = default), lanes per wave carrying agents (0 = default)
    ccx::StepShape step_shape{};
    bool ring_when_paced = false;                               // paced launches of this shape hand steps over through the sequence-word ring (step period close to the sim chain)
    int tun_occ_tables = -1;                                    // 0: the rollout kernel compares all pairs of an env instead of keeping LDS occupancy tables (-1: the library's choice)
    int tun_round_launches = 1;                                 // 1: a grid of several rounds of workgroups is launched round by round when its rows exceed ~3.5 GB, 2: always, 0: one launch
    int tun_small_shape = 1;                                    // 1: launches without observation rows take their own launch shape (shape_small), 0: the rows shape
    ccx::LaunchShape shape{};                                   // launches that write observation rows
    ccx::KParams kp{};
    ccx::LaunchShape shape_small{};                             // launches without them (rewards / flag bytes / compact rows)
    ccx::KParams kp_small{};
};
