/*
 * ccx.h -- C-ABI of libccx, the MI355X (gfx950) batched CollectiveCrossing step library.
 *
 * The reference (nima-siboni/collectivecrossing v0.1.3) has NO FFI: its hot path is the pure
 * Python method CollectiveCrossingEnv.step (src/collectivecrossing/collectivecrossing.py:161-261)
 * plus the strategy objects it calls.  Every entry point below names the reference function(s) it
 * replaces (file:line relative to the reference root); INTEGRATION.md shows the ctypes binding a
 * maintainer of the reference would add.
 *
 * Conventions
 *   - plain C, no torch / HIP types in signatures; device pointers travel as void* / typed pointers.
 *   - every function returns 0 (CCX_OK) or a negative ccx_status; ccx_last_error() returns the
 *     message of the last failure on the calling thread.
 *   - all array arguments are DEVICE pointers unless the name ends in _host.
 *   - array layout is struct-of-arrays, ENV-MAJOR: index [e*N + a] for per-agent arrays ("[E][N]"),
 *     agent slot a in 0..N-1, boarding agents first (a < num_boarding  <=> "boarding_{a}",
 *     otherwise "exiting_{a-num_boarding}"), exactly the reference's dict insertion order
 *     (collectivecrossing.py:101-150).
 *   - a handle is bound to one device and one HIP stream; calls are asynchronous on that stream
 *     unless stated otherwise; a handle is not thread-safe (neither is the reference env).
 */
#ifndef CCX_H
#define CCX_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CCX_ABI_VERSION 5

typedef enum ccx_status {
    CCX_OK = 0,
    CCX_EINVAL = -1,     /* bad argument / unsupported configuration            */
    CCX_ENOMEM = -2,     /* device or host allocation failed                    */
    CCX_EHIP = -3,       /* a HIP runtime call failed (message has the detail)  */
    CCX_ENODEVICE = -4   /* no usable gfx950 device                             */
} ccx_status;

/* reward_configs.py registry names -> enum (rewards.py:186-191) */
enum { CCX_REWARD_DEFAULT = 0, CCX_REWARD_SIMPLE_DISTANCE = 1, CCX_REWARD_BINARY = 2,
       CCX_REWARD_CONSTANT_NEGATIVE = 3 };
/* terminateds.py:86-89 */
enum { CCX_TERM_INDIVIDUAL_AT_DESTINATION = 0, CCX_TERM_ALL_AT_DESTINATION = 1 };
/* truncateds.py:99-102 ("custom" has the same arithmetic as "max_steps", truncateds.py:64-95) */
enum { CCX_TRUNC_MAX_STEPS = 0 };

/* scripted policies evaluated on the device (src/baseline_policies/) */
enum { CCX_POLICY_GREEDY = 1, CCX_POLICY_WAITING = 2, CCX_POLICY_RANDOM = 3 };

/* action codes, actions.py:8-24; CCX_ACTION_ABSENT = agent not in action_dict (it does not move,
 * collectivecrossing.py:197-202 only iterates over the dict's items) */
enum { CCX_ACTION_RIGHT = 0, CCX_ACTION_UP = 1, CCX_ACTION_LEFT = 2, CCX_ACTION_DOWN = 3,
       CCX_ACTION_WAIT = 4, CCX_ACTION_ABSENT = 255 };

/*
 * Lowered, POD form of CollectiveCrossingConfig (configs.py:15-77) + the four strategy configs.
 * Geometry is ABSOLUTE (already passed through utils/geometry.py:34-40):
 *   tram_left = width/2 - tram_length/2, tram_right = width/2 + tram_length/2,
 *   door_left/right = tram_left + cfg.tram_door_left/right.
 */
typedef struct ccx_params {
    int32_t width, height, division_y;
    int32_t tram_left, tram_right, door_left, door_right;
    int32_t num_boarding, num_exiting;
    int32_t boarding_dest_y, exiting_dest_y;
    int32_t reward_mode, terminated_mode, truncated_mode;
    int32_t max_steps;
    int32_t _pad0;
    /* DefaultRewardConfig (reward_configs.py:24-57); distance_penalty_factor is shared with
     * SimpleDistanceRewardConfig (reward_configs.py:60-77) */
    double boarding_destination_reward, tram_door_reward, tram_area_reward, distance_penalty_factor;
    /* BinaryRewardConfig (reward_configs.py:80-101), ConstantNegativeRewardConfig (:104-121) */
    double goal_reward, no_goal_reward, step_penalty;
} ccx_params;

typedef struct ccx_handle ccx_handle;

/* per-agent result byte of one step ("agent_flags", u8 [E][N]) */
#define CCX_AF_TERMINATED   0x01u /* terminateds[id] (always present, terminateds.py:40-82)             */
#define CCX_AF_TRUNCATED    0x02u /* truncateds[id]; meaningful only with CCX_AF_LIVE (truncateds.py:56)  */
#define CCX_AF_LIVE         0x04u /* agent was neither terminated nor truncated BEFORE this step:         */
                                  /*   rewards[id] and truncateds[id] exist (rewards.py:64, truncateds.py:56) */
#define CCX_AF_OBS          0x08u /* observations[id]/infos[id] emitted (collectivecrossing.py:243-254)   */
#define CCX_AF_IN_TRAM_AREA 0x10u /* infos[id]["in_tram_area"]   (collectivecrossing.py:551-554)          */
#define CCX_AF_AT_DOOR      0x20u /* infos[id]["at_door"]        (collectivecrossing.py:556-563)          */
#define CCX_AF_ACTIVE       0x40u /* infos[id]["active"]                                                  */
#define CCX_AF_AT_DEST      0x80u /* infos[id]["at_destination"] (collectivecrossing.py:663-683)          */

/* per-env result byte of one step ("env_flags", u8 [E]) */
#define CCX_EF_ALL_TERMINATED 0x01u /* terminateds["__all__"] (collectivecrossing.py:256,258) */
#define CCX_EF_ALL_TRUNCATED  0x02u /* truncateds["__all__"]  (collectivecrossing.py:257,259) */
#define CCX_EF_RESET          0x04u /* rollout only: the env was auto-reset after this step    */

/* SoA state of the batch: the Agent dataclass (types.py:16-83) + env._step_count. */
typedef struct ccx_state {
    int32_t* x;           /* [E][N] */
    int32_t* y;           /* [E][N] */
    uint8_t* active;      /* [E][N] Agent.active      */
    uint8_t* terminated;  /* [E][N] Agent.terminated  */
    uint8_t* truncated;   /* [E][N] Agent.truncated   */
    int32_t* step_count;  /* [E]    env._step_count   */
    int32_t* episode;     /* [E]    number of auto-resets so far (selects the reset-pool entry) */
} ccx_state;

/* outputs of one step for the whole batch; any pointer may be NULL to skip that output */
typedef struct ccx_step_out {
    float*   obs;          /* [E][N][L], L = 6 + 4N, DefaultObservation (observations.py:43-94) */
    double*  reward;       /* [E][N] f64, meaningful where CCX_AF_LIVE (rewards.py:44-182)      */
    uint8_t* agent_flags;  /* [E][N] CCX_AF_*                                                   */
    uint8_t* env_flags;    /* [E]    CCX_EF_*                                                   */
    float*   obs_compact;  /* [E][N][4] CCX_OBS_COMPACT, see below                                 */
} ccx_step_out;

/* trajectory outputs of a K-step rollout; step s of env e lives at [s][e]...; NULL skips */
typedef struct ccx_rollout_out {
    float*   obs;          /* [K][E][N][L] */
    double*  reward;       /* [K][E][N]    */
    uint8_t* agent_flags;  /* [K][E][N]    */
    uint8_t* env_flags;    /* [K][E]       */
    float*   obs_compact;  /* [K][E][N][4] */
} ccx_rollout_out;

/*
 * CCX_OBS_COMPACT -- an OPTIONAL second observation output for consumers that live on the GPU (next to,
 * never instead of, the DefaultObservation layout).  A DefaultObservation row (observations.py:79-92) is six
 * per-row numbers plus, for EVERY agent j of the env, (x_j, y_j, type_j, active_j): the N rows of an env
 * repeat the same 4N numbers N times (152 of the 162 bytes a C2 agent-step writes).  obs_compact holds each
 * agent's four numbers ONCE per env and step: f32 [..][E][N][4] = (x, y, type 0 boarding / 1 exiting,
 * active 0/1), 16 bytes per agent-step instead of 16N + 24.  ccx_expand_observations turns compact rows back
 * into the DefaultObservation rows with the very gather ccx_observe uses, bit for bit (rows = envs, or
 * steps x envs for a trajectory): a policy network can consume the compact tensor directly, or expand just
 * the mini-batch it samples.  The row constants (door centre, division_y, door_left, door_right) are those
 * of the handle.
 */
int ccx_expand_observations(ccx_handle* h, const float* obs_compact /* [rows][N][4] */, int64_t rows,
                            float* obs /* [rows][N][L] */);

/* device-side counters accumulated by ccx_rollout (u64 each; ccx_read_counters copies to host) */
typedef struct ccx_counters {
    uint64_t env_steps;        /* env-steps executed                                      */
    uint64_t agent_steps;      /* env_steps * N (all slots)                               */
    uint64_t live_agent_steps; /* agent-steps of agents with CCX_AF_LIVE                  */
    uint64_t episodes;         /* auto-resets performed                                   */
    uint64_t moves;            /* successful position changes (collectivecrossing.py:408) */
    uint64_t arrivals;         /* deactivations (collectivecrossing.py:210-212)           */
} ccx_counters;


/* library / build info ------------------------------------------------------------------------ */
int         ccx_abi_version(void);
const char* ccx_build_info(void);      /* "libccx <ver> gfx950 hip <ver>" */
const char* ccx_last_error(void);      /* thread-local message of the last failing call */

/* observation length L = 2 + 4 + 4N (observations.py:113-118) */
int32_t ccx_obs_len(int32_t num_agents);

/*
 * Create a batch of num_envs independent environments on `device`, using HIP stream `stream`
 * (a hipStream_t passed as void*; NULL = the device's default stream).  env_offset is the global
 * index of this handle's env 0 and total_envs the global batch size (multi-GPU sharding: the
 * reset-pool entry of env e, episode j depends on env_offset + e, j, total_envs and the pool size only
 * (see ccx_set_reset_pool), so the trajectory of a global env does not depend on how many GPUs the
 * batch is split over).
 * Replaces CollectiveCrossingEnv.__init__ (collectivecrossing.py:44-89) for E instances.
 * All agents start at (0,0), active, step_count 0: call ccx_set_state / ccx_reset_from_pool next.
 */
int ccx_create(const ccx_params* params, int32_t num_envs, int64_t env_offset, int64_t total_envs,
               int device, void* stream, ccx_handle** out);
void ccx_destroy(ccx_handle* h);

int32_t ccx_num_envs(const ccx_handle* h);
int32_t ccx_num_agents(const ccx_handle* h);

/* pointers to the handle's own device-resident SoA state (valid until ccx_destroy) */
int ccx_state_view(ccx_handle* h, ccx_state* out);

/*
 * Overwrite / read back the state (tests reach into env._agents[...] the same way, e.g. the
 * reference's test_collective_crossing.py:139-143).  Host pointers; NULL members are skipped.
 * Synchronous.  set_state validates 0 <= x <= width, 0 <= y <= height (cells that exist,
 * collectivecrossing.py:515) and flags in {0,1}.
 */
int ccx_set_state_host(ccx_handle* h, const ccx_state* src);
int ccx_get_state_host(ccx_handle* h, ccx_state* dst);

/*
 * Reset pool: P seeded initial placements, u8 xy pairs [P][N][2], produced on the host by the
 * reference's rejection-sampling reset (collectivecrossing.py:91-150) for seeds seed0..seed0+P-1.
 * Device pointer; the library keeps the pointer (caller keeps the allocation alive).
 * Cursor: env e (global index g = env_offset + e) starts episode j from entry (g + j * stride) mod P
 * with stride = total_envs mod P, or 1 when P divides total_envs (so that every env still walks
 * through the pool instead of restarting from one placement forever).
 */
int ccx_set_reset_pool(ccx_handle* h, const uint8_t* pool_xy, int64_t pool_size);

/* (Re)start every env whose mask byte is non-zero (mask NULL = all) from its pool entry for the
 * env's current episode index; clears flags and step_count.  = reset() body :97-150 */
int ccx_reset_from_pool(ccx_handle* h, const uint8_t* env_mask);

/*
 * Seeded placement ON THE DEVICE, bit-identical to reset(seed=s) of the reference INCLUDING its
 * random stream (gymnasium np_random = numpy Generator(PCG64(SeedSequence(s))), rejection sampling
 * of collectivecrossing.py:100-150).  Synchronous; CCX_EINVAL if a placement finds no free cell
 * within 65536 draws per agent (the reference would loop forever).
 *   ccx_fill_reset_pool_seeded: pool_xy[p] = placement of seed seed0 + p, p < pool_size (device
 *     buffer u8 [pool_size][N][2] of the caller; pass it to ccx_set_reset_pool afterwards).
 *   ccx_reset_seeded: env e (mask NULL or mask[e] != 0) restarts from reset(seed=seeds[e]):
 *     positions, active = 1, flags and step_count cleared (seeds: device u64 [E]).
 */
int ccx_fill_reset_pool_seeded(ccx_handle* h, uint8_t* pool_xy, int64_t pool_size, uint64_t seed0);
int ccx_reset_seeded(ccx_handle* h, const uint64_t* seeds, const uint8_t* env_mask);

/*
 * Batched GreedyPolicy with epsilon = 0 (src/baseline_policies/greedy_policy.py:33-449): the action
 * the reference's scripted policy picks for every agent of env.agents from the CURRENT state,
 * actions u8 [E][N]; agents that are terminated or truncated get CCX_ACTION_ABSENT (the rollout loop
 * of scripts/run_greedy_policy_demo.py:67-109 only asks the policy for env.agents).
 */
int ccx_greedy_actions(ccx_handle* h, uint8_t* actions);
/* same for any scripted policy: CCX_POLICY_GREEDY, or CCX_POLICY_WAITING = WaitingPolicy with
 * epsilon = 0 (src/baseline_policies/waiting_policy.py:33-131: boarding agents outside the tram area
 * wait until every exiting agent that is still in env.agents stands on its destination row) */
int ccx_policy_actions(ccx_handle* h, int32_t policy, uint8_t* actions);

/* DefaultObservation of the CURRENT state for every agent (what reset() returns, :153-159). */
int ccx_observe(ccx_handle* h, float* obs /* [E][N][L] */);

/*
 * One step of every env = CollectiveCrossingEnv.step (collectivecrossing.py:161-261):
 *   actions u8 [E][N] (CCX_ACTION_*), order u8 [E][N] or NULL: order[e][k] = slot of the agent
 *   moved k-th (a permutation of 0..N-1; NULL = 0,1,..,N-1 = dict order of possible_agents).
 */
int ccx_step(ccx_handle* h, const uint8_t* actions, const uint8_t* order, const ccx_step_out* out);

/*
 * K fused steps in one launch (state stays in registers between steps).
 *   actions u8 [K][E][N]; order u8 [K][E][N] or NULL.
 *   auto_reset != 0: an env whose step raised terminateds["__all__"] or truncateds["__all__"] is
 *   restarted from the reset pool before its next step (the rollout loop
 *   `if done: env.reset(seed=...)` of the reference's demos, scripts/run_greedy_policy_demo.py:67-109).
 *   out may be NULL (no trajectory); counters are accumulated into the handle.
 */
int ccx_rollout(ccx_handle* h, int32_t num_steps, const uint8_t* actions, const uint8_t* order,
                int32_t auto_reset, const ccx_rollout_out* out);

/*
 * K fused steps with the actions chosen ON THE DEVICE by a scripted policy from the pre-step state
 * (policy -> step -> policy ..., the loop of scripts/run_greedy_policy_demo.py:67-109 inside one
 * launch).  CCX_POLICY_GREEDY = GreedyPolicy with epsilon = 0 (greedy_policy.py:33-449); move order
 * is slot order (env.agents order).  actions_out (u8 [K][E][N], may be NULL) receives the chosen
 * actions, CCX_ACTION_ABSENT for agents outside env.agents.
 */
int ccx_rollout_policy(ccx_handle* h, int32_t num_steps, int32_t policy, int32_t auto_reset,
                       const ccx_rollout_out* out, uint8_t* actions_out);
/*
 * CCX_POLICY_RANDOM: uniform random actions drawn on the device -- the random-action rollouts of the
 * reference's tests and demos (e.g. tests/.../test_trajectory_vcr.py) without an action tensor (SURVEY 8b:
 * `rng_seed` of ccx_rollout).  The action of agent slot a of global env g at step t (0-based) of its episode j is
 *     k = g*0x9E3779B1 + j*0x85EBCA77 + t*0xC2B2AE3D + a*0x27D4EB2F + seed_lo        (u32 arithmetic)
 *     k = mix(k);  k ^= seed_hi;  k = mix(k);       mix: k^=k>>16; k*=0x7FEB352D; k^=k>>15; k*=0x846CA68B; k^=k>>16
 *     action = (k * 5) >> 32
 * for every agent of env.agents (others: CCX_ACTION_ABSENT in actions_out).  It depends on the env's own
 * counters only, so the stream of an env is the same for any split into launches and any world size.  This is
 * NOT numpy's stream: trajectories driven by it are pinned by the oracle's restatement of the same function.
 */
int ccx_set_rng_seed(ccx_handle* h, uint64_t seed);
/*
 * Epsilon-greedy for ccx_rollout_policy's CCX_POLICY_GREEDY / CCX_POLICY_WAITING: the `randomness_factor` of the
 * reference's policies (greedy_policy.py:25-59, waiting_policy.py:25-59; create_greedy_policy's default is 0.1).
 * With probability epsilon an agent takes one of its VALID actions uniformly -- the directions the pre-step state
 * lets it enter (env._is_move_valid) plus wait -- instead of the policy's choice.  The reference draws from one
 * numpy RandomState shared by all agents of an env (a sequential stream); the device draws are counter-based
 * like CCX_POLICY_RANDOM's ("replace with a documented counter-based RNG"):
 *     u = word(seed_lo, seed_hi ^ 0x5BD1E995; g, j, t, a)     (the k of ccx_set_rng_seed's formula, before "* 5")
 *     explore  iff  u < floor(epsilon * 2^32)                (epsilon = 1: 2^32 - 1)
 *     action   =  the ((mix(u + 0x9E3779B9) * count) >> 32)-th valid action in ascending order, count = #valid
 * pinned by the oracle's restatement (ccxo_set_policy_epsilon).  0 (the default) = the deterministic policies.
 * The host classes of collectivecrossing_amd/baseline_policies.py keep the reference's own RandomState stream, and so does
 * the device with ccx_set_policy_stream(CCX_EPS_STREAM_MT19937) (below).
 * ccx_policy_actions honours it as well (same draws: a loop of ccx_policy_actions + ccx_step takes the actions
 * ccx_rollout_policy takes); ccx_greedy_actions is always epsilon = 0.
 */
int ccx_set_policy_epsilon(ccx_handle* h, double epsilon);
/*
 * Where the exploration draws of the scripted policies come from.
 *   CCX_EPS_STREAM_COUNTER (default): the counter-based draws documented above.
 *   CCX_EPS_STREAM_MT19937: the REFERENCE's own stream.  The reference's policies hold one
 *     `np.random.RandomState(seed)` (greedy_policy.py:31, waiting_policy.py:31; create_greedy_policy /
 *     create_waiting_policy seed it with 42, :452-465) and every get_action call draws from it in turn --
 *     `random_state.random() < randomness_factor` (:49), then `random_state.choice(valid_actions)` (:57) -- for the
 *     agents of env.agents in index order (scripts/run_greedy_policy_demo.py:67-109).  With this stream every env owns
 *     one such generator (numpy's legacy MT19937: init_genrand seeding, 53-bit doubles from two outputs, choice =
 *     masked rejection on 32-bit outputs, a one-entry list draws nothing), seeded here with seeds[e] (HOST array u32 [E])
 *     or, seeds NULL, with `seed` for every env (= one policy object per env, as the reference's callers have), and
 *     walked on the device by one wave per env: an epsilon episode of the reference replays ACTION FOR ACTION
 *     (tests: the reference-recorded g11_epsilon_policy_* episodes).  epsilon is compared as the double given to
 *     ccx_set_policy_epsilon.  The stream is sequential per env, so
 *       - ccx_policy_actions draws from it (every call advances the generators; epsilon 0 draws nothing);
 *       - ccx_rollout_policy (greedy / waiting, epsilon > 0) runs policy -> step -> policy ... as separate launches
 *         on the handle's stream instead of the fused kernel: the mode for replaying / validating against the
 *         reference, not for throughput (any batch size works; the fused kernel keeps the counter-based draws);
 *       - the generators run on across episodes (auto_reset), as a policy object of the reference does;
 *       - ccx_greedy_actions and CCX_POLICY_RANDOM never touch them.
 *     Calling it again re-seeds.  Synchronous.
 * ccx_get_policy_stream: the kind in effect and (state != NULL, host u32 [E][625]) every env's generator as numpy's
 * RandomState.get_state() has it: key[624] then pos.  Synchronises the handle's stream.
 */
#define CCX_EPS_STREAM_COUNTER 0
#define CCX_EPS_STREAM_MT19937 1
int ccx_set_policy_stream(ccx_handle* h, int32_t kind, const uint32_t* seeds, uint32_t seed);
int ccx_get_policy_stream(ccx_handle* h, int32_t* kind, uint32_t* state);

/*
 * Opt-in input validation for the array API (the dict API validates on the host).  The reference raises
 * ValueError for an action outside 0..4 and for an agent id the env does not have
 * (_check_action_and_agent_validity, collectivecrossing.py:685-711); ccx_step / ccx_rollout by default
 * treat any action byte other than 0..3 as "no move" and trust `order` to be a permutation.  With
 * checking enabled every launch that takes an action tensor is preceded by an elementwise kernel that
 * counts action bytes outside {0..4, CCX_ACTION_ABSENT} and move-order rows that are not a permutation
 * of 0..N-1; the next synchronising call (ccx_synchronize, ccx_read_counters, ccx_check_inputs) then
 * fails ONCE with CCX_EINVAL and a message naming both counts ("Invalid action", "Unknown agent ID").
 * The offending entries have been stepped as "no move" / an undefined order by then: a caller that
 * gets the error must discard those envs.  Costs one extra read of the action tensor (1 of 16N+34
 * bytes per agent-step).
 */
int ccx_set_check_inputs(ccx_handle* h, int32_t enabled);
int ccx_check_inputs(ccx_handle* h);     /* synchronises; CCX_EINVAL if anything was counted */

int ccx_zero_counters(ccx_handle* h);
int ccx_read_counters(ccx_handle* h, ccx_counters* out_host);   /* synchronous */
/* device pointer to the 6 u64 counters (for an RCCL all-reduce by the caller).  The rollout kernel
 * accumulates per-tile partial counters; this call enqueues their reduction on the handle's stream, so
 * the totals cover every launch enqueued BEFORE it (call it again after later launches). */
int ccx_counters_device_ptr(ccx_handle* h, uint64_t** out);

/*
 * The one collective of the multi-GPU layout, on RCCL directly: contiguous env shards never
 * exchange data; once per measurement window the six u64 counters are summed over the ranks.
 * ccx_rccl_allreduce_counters enqueues, on the handle's stream, the reduction of the per-tile partial
 * counters and ONE ncclAllReduce(ncclSum, 6 x u64) over `rccl_comm` (an ncclComm_t passed as void*)
 * into out_device (device memory, 6 x u64); *num_ranks (may be NULL) receives the communicator size.
 * The handle's own counters keep the per-rank values.  The communicator is the caller's: either one it
 * already has, or one made with the helpers below (rank 0 draws the 128-byte unique id, ships it to the
 * other ranks by any means -- bench.py uses the torch.distributed store -- and every rank calls
 * ccx_rccl_comm_create with the device its handle lives on).  librccl.so.1 is bound at run time; without
 * it these calls fail with CCX_ENODEVICE and everything else keeps working.
 * No reference counterpart: its parallelism is one env per RLlib EnvRunner process
 * (examples/training_script.py:84).
 */
#define CCX_RCCL_UNIQUE_ID_BYTES 128
int ccx_rccl_unique_id(void* id_out_128 /* host */);
int ccx_rccl_comm_create(int32_t num_ranks, const void* id_128 /* host */, int32_t rank, int32_t device,
                         void** comm_out);
int ccx_rccl_comm_destroy(void* comm);
int ccx_rccl_allreduce_counters(ccx_handle* h, void* rccl_comm, uint64_t* out_device, int32_t* num_ranks);

/* Launch timing, off by default (two event records per launch cost a step-wise loop several
 * microseconds per step): when enabled, HIP events are recorded on the handle's stream around every
 * ccx_step / ccx_rollout kernel and ccx_last_launch_ms returns the duration of the most recent one
 * (it synchronises on the stop event). */
int ccx_set_timing(ccx_handle* h, int32_t enabled);
int ccx_last_launch_ms(ccx_handle* h, float* ms);

/* launch-shape tuning (0 = library default): lanes of each 64-wide wavefront that carry agents
 * (a multiple of the per-env lane group), and wavefronts per workgroup. */
int ccx_set_launch_shape(ccx_handle* h, int32_t lanes_per_wave, int32_t waves_per_block);
/* writer wavefronts per env tile (0 = library default, 1..7): the wavefronts that turn a step into
 * reward / flag bytes / observation rows next to the wavefront that simulates it */
int ccx_set_writers(ccx_handle* h, int32_t writers_per_tile);
/* store throttle: stores a writer wavefront may still have in flight when it starts the next step
 * (0 = library default, -1 = unlimited, 1..63).  Many small env tiles oversubscribe the HBM write
 * queues; bounding the in-flight stores raises the sustained write rate. */
int ccx_set_store_throttle(ccx_handle* h, int32_t max_stores_in_flight);
/* Step pacing of rollouts that write observations: every env tile starts env-step s no earlier than
 * t0 + s * pace on the GPU's 100 MHz clock, which turns the output into a smooth stream at (just under)
 * the HBM drain rate instead of bursts that oversubscribe the write queues.
 *   0  = adaptive (default): starts from the write rate measured in-process at the first long rollout
 *        (ccx_set_pace_calibration; an assumed 6.8 TB/s without it) and is retuned by the kernel after every
 *        launch that lasts ~50 us or more (64 steps of the 4096 x 8 shape, 5 of a 4096 x 32 one; a launch
 *        shorter than ~12 us is not paced at all: late => slower, on time => 0.4 % faster, a collapse of the drain
 *        rate => +3 % and a decaying floor); batches too small to fill the drain rate are not paced at all
 *   -1 = off;   > 0 = fixed pace in nanoseconds per env-step.
 * ccx_get_step_pace returns the pace in effect (synchronises).  Results never depend on it. */
int ccx_set_step_pace(ccx_handle* h, int32_t ns_per_env_step);
int ccx_get_step_pace(ccx_handle* h, float* ns_per_env_step);
/* The controller's state for diagnostics and tests (synchronises): out6 = { the pace the next launch starts
 * from (ns), the floor just above the last collapse (ns), launches since that collapse, 1 if rollouts of this
 * shape are paced at all, the pace of the last collapse = the cliff (ns), how often that cliff was confirmed } */
int ccx_get_pace_state(ccx_handle* h, float* out6);
/* Where the ADAPTIVE controller starts (ns per env-step; 0 = the library's assumption of 6.8 TB/s): a
 * caller that remembers the pace a previous handle of the same shape converged to (ccx_get_step_pace)
 * skips the descent of the first launches; the value is also the controller's first floor (it decays like any
 * floor), so that the start of a process is not spent probing below a pace that is already known.  Restarts the
 * controller. */
int ccx_set_step_pace_start(ccx_handle* h, float ns_per_env_step);
/* Start-up calibration (default on; CCX_PACE_CALIBRATION=0 in the environment turns it off for new handles).  The first
 * adaptive paced rollout of a handle / launch shape -- one that writes observations for ~50 us or more -- first streams
 * filler into the caller's own observation buffer (which that rollout overwrites anyway) for ~2.5 ms with the writer
 * wavefronts' store type, takes the best pass as a lower bound of the drain rate of this box and starts the controller
 * 5 % above it (it descends further by itself until a launch comes in late).  One stream synchronisation; never inside a stream capture (a rollout that would have to restart
 * the controller while capturing fails with CCX_EINVAL: run one eager rollout of the shape first).  Without it -- or
 * before the first such rollout -- the start value is an assumed 6.8 TB/s. */
int ccx_set_pace_calibration(ccx_handle* h, int32_t enabled);
/* where the pace controller started: the value (ns per env-step; 0 if rollouts of this shape are not paced), its
 * source (CCX_PACE_START_*) and the write rate the calibration probe measured (GB/s, 0 = it did not run).  Any pointer
 * may be NULL. */
#define CCX_PACE_START_UNPACED    0  /* rollouts of this shape are not paced (too small to be memory-bound, or pacing off) */
#define CCX_PACE_START_ASSUMED    1  /* the library's assumption (6.8 TB/s): no calibration has run (yet)                  */
#define CCX_PACE_START_CALLER     2  /* ccx_set_step_pace_start                                                           */
#define CCX_PACE_START_CALIBRATED 3  /* measured in-process at the first adaptive rollout                                 */
#define CCX_PACE_START_FIXED      4  /* ccx_set_step_pace(h, ns > 0): no controller                                       */
int ccx_get_pace_start(ccx_handle* h, float* ns_per_env_step, int32_t* source, float* probe_gbs);
/* Performance experiments without an ABI change; results never depend on a tunable.  -1 = the library's
 * choice for the launch shape (pace_phase, tile_map).
 *   "pace_phase"   0 = every tile starts env-step s at t0 + s * pace, 1 = tiles are phased over the step
 *                  period in tile order (one write window sweeps through the slab), 2 = hashed phases,
 *                  3 = in tile order within each XCD's share (with tile_map 0)
 *   "tile_map"     0 = workgroups of one XCD take adjacent tiles, v >= 1 = groups of 2^(v-1) adjacent tiles
 *                  per XCD dealt round-robin (1 = tile = workgroup index)
 *   "writer_roles" 1 = writer wave 0 of a tile writes the small outputs only and the others share the observation
 *                  rows, 0 = every writer takes a share of the rows (writer 0 the small outputs on top), -1 = by
 *                  launch shape (split wherever a tile of at most 12 store iterations per step has two or more writers)
 *   "hand2"        how the simulating wavefront of a tile hands an env-step to its writer wavefronts: 1 (default) = launches
 *                  that are not paced use a ring of hand-off words with a sequence word and per-writer progress words in
 *                  LDS (no barrier; paced launches keep one workgroup barrier per step, which regularises their store
 *                  stream), 0 = a barrier per step always, 2 = the ring always
 *   "max_launch_steps"  > 0: ccx_rollout cuts a rollout into kernel launches of at most this many env-steps (the library
 *                  does so by itself where one launch would exceed 4 GiB per small output stream); 0 = automatic
 *   "pair_rows"    in small batches (role-split writers, launches that are not paced) every writer can get a second
 *                  staging slot in LDS and a row writer then takes TWO env-steps per iteration whenever the simulating
 *                  wavefront is that far ahead: -1 (default) = where it pays (half-tile shapes: up to 128 full tiles),
 *                  1 = in every small batch, 0 = never */
int ccx_set_tunable(ccx_handle* h, const char* name, int32_t value);
/* workgroups of a rollout launch with outputs, and how many of them the device holds at once (a grid
 * larger than that runs in rounds; the pace of a partial last round is scaled accordingly) */
int ccx_get_residency(ccx_handle* h, int32_t* resident_workgroups, int32_t* workgroups);
/* the writers per tile and the store throttle in effect (0 = unlimited) */
int ccx_get_writer_shape(ccx_handle* h, int32_t* writers_per_tile, int32_t* max_stores_in_flight);
int ccx_get_launch_shape(ccx_handle* h, int32_t* lanes_per_wave, int32_t* waves_per_block,
                         int32_t* group_lanes, int32_t* num_blocks);
/* Plugin point for position-only user strategies on the batch path.  The reference accepts ANY registered RewardFunction /
 * TerminatedFunction class (rewards.py:16-38, 186-216; terminateds.py:16-36, 86-114); a class whose value depends on nothing
 * but the agent's own type and cell is lowered to a table by the host (collectivecrossing_amd/params.py: position_only_tables)
 * and evaluated inside the kernels like the built-in strategies, at the same speed:
 *   ccx_set_reward_table      rewards[id] of an agent that was live before the step and stands on cell (x, y) after it =
 *                             table[type][y][x]: f64 [height + 1][width + 1] per agent type (boarding, exiting), host memory.
 *                             Staged in LDS next to the cell table (16 bytes per cell of the padded grid): grids whose tables do
 *                             not fit are refused with CCX_EINVAL.
 *   ccx_set_terminated_table  terminateds[id] = table[type][y][x] != 0 (u8), for individual_at_destination handles;
 *                             terminateds["__all__"] stays all(values) (collectivecrossing.py:256).  Deactivation on the destination
 *                             row (:210-212) is env logic, not strategy, and does not change.
 * NULL for both types restores the built-in strategy of the handle's params.  Both synchronise the handle's stream. */
int ccx_set_reward_table(ccx_handle* h, const double* boarding_per_cell, const double* exiting_per_cell);
int ccx_set_terminated_table(ccx_handle* h, const uint8_t* boarding_per_cell, const uint8_t* exiting_per_cell);

/* Launch shape of the SHORT-LAUNCH kernel (csrc/ccx_step.hip): ccx_step and ccx_rollout calls of at most 16 steps with an
 * action tensor (with or without a move order) are CollectiveCrossingEnv.step itself (collectivecrossing.py:161-261) with one workgroup
 * per env tile -- a sim wave plus *row_waves waves that gather the observation rows, one LDS barrier per step, no ring, no
 * pacing.  *ok = 0: this handle's short launches take the rollout kernel (grid too large for the LDS tables).  Tunables:
 * "step_kernel" (0 = always the rollout kernel), "step_rows" (row waves per tile), "step_lanes" (lanes per wave carrying
 * agents). */
int ccx_get_step_shape(ccx_handle* h, int32_t* ok, int32_t* lanes_per_wave, int32_t* row_waves,
                       int32_t* num_blocks, int32_t* lds_bytes);

/* Zero-copy I/O for single-env stepping (the dict API of collectivecrossing.py:161-261 needs every
 * output on the host after each step): the device address of page-locked host memory (hipHostMalloc /
 * hipHostRegister, e.g. a torch pinned tensor).  Pass it to ccx_step as actions / outputs and the
 * kernel reads and writes the host buffer over the host link -- one launch + one sync per step,
 * no memcpy.  Fails with CCX_EINVAL for pageable or unregistered memory. */
int ccx_host_device_pointer(ccx_handle* h, void* pinned_host, void** device_ptr);
/* rebind the handle to another HIP stream of its device (synchronises the old one first) */
int ccx_set_stream(ccx_handle* h, void* stream);
int ccx_synchronize(ccx_handle* h);

#ifdef __cplusplus
}
#endif
#endif /* CCX_H */
