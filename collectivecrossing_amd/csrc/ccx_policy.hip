// ccx_policy.hip -- the reference's GreedyPolicy (epsilon = 0, or epsilon-greedy with the device's counter-based draws)
// (src/baseline_policies/greedy_policy.py:33-449) for every agent of every env, one thread per
// (env, agent).  The rule reads only the PRE-step state (the reference computes all actions before
// env.step, scripts/run_greedy_policy_demo.py:67-109), so agents are independent:
//   head for the door centre column on the row next to the division line, cross, then run along y
//   to the destination row (:96-165); if that move is not possible (wall, bounds, a cell held by
//   another ACTIVE agent: env._is_move_valid, collectivecrossing.py:345-369) take the first legal
//   move of the preference list (:311-449), else wait.
// Agents that are terminated or truncated get CCX_ACTION_ABSENT (the reference only asks the policy
// for env.agents).  Legality of a move w.r.t. walls/bounds is the 4 neighbour bits of the per-cell
// table (see ccx_kernels.hip); occupancy is an O(N) scan of the env's agents (this kernel is not on
// the rollout's critical path; the fused rollout uses the LDS occupancy tables instead).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ccx_kernels.h"
#include "ccx_greedy.h"

namespace ccx {

// The scripted policy's own choice for live agent i of env `env` (epsilon aside) and, in free_dirs, the directions it
// could move in: the neighbour bit of its cell and no other ACTIVE agent on the target (env._is_move_valid,
// collectivecrossing.py:345-369) -- what both the policy's fallback list and an epsilon draw choose from.
__device__ __forceinline__ uint32_t scripted_choice(const KParams& p, const KState& st,
                                                    const unsigned long long* __restrict__ cell_info, const int env,
                                                    const int i, const int policy, uint32_t& free_dirs) {
    const size_t base = (size_t)env * p.N, t = base + (size_t)i;
    const int cx = st.x[t], cy = st.y[t];
    const int Wp = p.W + 3;
    const uint32_t cw = (uint32_t)cell_info[(cy + 1) * Wp + cx + 1];
    const uint32_t nv = cw & 0xFu;   // enterable neighbours
    const uint32_t cand = greedy_candidates(p, i < p.Nb, cx, cy);
    bool waits = false;
    if (policy == CCX_K_POLICY_WAITING && i < p.Nb && !(cw & kCellInTram)) {   // waiting_policy.py:92-100
        for (int b = p.Nb; b < p.N; ++b)                                  // :119-129
            waits |= !(st.terminated[base + b] || st.truncated[base + b]) && st.y[base + b] != p.edy;
    }
    free_dirs = 0u;
    for (uint32_t a = 0; a < 4u; ++a) {
        if (!((nv >> a) & 1u)) continue;
        const int nx = cx + (a == 0u) - (a == 2u), ny = cy + (a == 1u) - (a == 3u);
        bool taken = false;
        for (int b = 0; b < p.N; ++b)
            taken |= (b != i) && st.active[base + b] && st.x[base + b] == nx && st.y[base + b] == ny;
        if (!taken) free_dirs |= 1u << a;
    }
    uint32_t chosen = 4u;
    for (int k = 0; k < 6 && !waits; ++k) {   // candidate 0 = primary, 1..4 preference list, 5 = wait
        const uint32_t a = k < 5 ? ((cand >> (4 * k)) & 0xFu) : 4u;
        if (a == 4u) break;
        if ((free_dirs >> a) & 1u) {
            chosen = a;
            break;
        }
    }
    return chosen;
}

// policy: CCX_K_POLICY_GREEDY, or CCX_K_POLICY_WAITING = WaitingPolicy(epsilon = 0)
// (baseline_policies/waiting_policy.py:33-131): a boarding agent outside the tram area waits while
// an exiting agent that is neither terminated nor truncated is not on its destination row yet.
__global__ void greedy_actions_kernel(const KParams p, const KState st,
                                      const unsigned long long* __restrict__ cell_info,
                                      uint8_t* __restrict__ actions, const int policy) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t total = (size_t)p.E * p.N;
    if (t >= total) return;
    const int env = (int)(t / p.N), i = (int)(t % p.N);
    if (st.terminated[t] || st.truncated[t]) {
        actions[t] = (uint8_t)CCX_K_ABSENT;
        return;
    }
    uint32_t free_dirs;
    uint32_t chosen = scripted_choice(p, st, cell_info, env, i, policy, free_dirs);
    if (p.eps_thr) {   // epsilon-greedy with the device's counter-based draws (include/ccx.h: ccx_set_policy_epsilon)
        const uint32_t u = random_word(p.rng_lo, p.rng_hi ^ kEpsStream, (uint32_t)(p.env_offset + env), (uint32_t)st.episode[env],
                                       (uint32_t)st.step_count[env], (uint32_t)i);
        if (u < p.eps_thr) chosen = explore_action(u, free_dirs);
    }
    actions[t] = (uint8_t)chosen;
}

// ---------------------------------------------------------------------------------------------
// The reference's OWN epsilon stream (include/ccx.h: ccx_set_policy_stream, CCX_EPS_STREAM_MT19937).  The reference's
// policies hold one `np.random.RandomState(seed)` (greedy_policy.py:31, waiting_policy.py:31) and every get_action
// call draws from it in turn (:49 `random_state.random() < randomness_factor`, :57 `random_state.choice(valid_actions)`),
// agents in env.agents order (scripts/run_greedy_policy_demo.py:67-109): a sequential stream per env.  One wave per env:
// lane i prepares agent i (live?  its valid directions, the policy's own choice -- all from the pre-step state, so
// independent), then the WHOLE wave walks the agents in index order with the env's generator in LDS, wave-uniformly:
//   MT19937 (numpy's legacy RandomState): key[624] + pos per env in global memory, staged through LDS; the twist runs
//     in place, 64 words at a time (word i reads old i, old i+1 and i+397 -- old below 227, new from 227 on -- none of
//     which another lane of the same 64-word chunk writes before the chunk's reads are done);
//   random()  = ((a >> 5) * 2^26 + (b >> 6)) / 2^53 from two outputs; compared with epsilon AS DOUBLES (exact);
//   choice(v) = randint(0, len(v)): a list of one entry draws nothing, otherwise masked rejection on 32-bit outputs
//     (mask = smallest 2^k - 1 >= len - 1); v = the valid actions in ascending order, wait (4) always among them.
// The tests' CPU checker restates the same generator and is pinned against numpy itself and against the
// reference-recorded epsilon episodes g11_epsilon_policy_*; tests/test_gpu_policy_stream.py replays those on the device.
// ---------------------------------------------------------------------------------------------
constexpr int kMtWords = 624;

__global__ __launch_bounds__(64) void policy_stream_kernel(const KParams p, const KState st,
                                                           const unsigned long long* __restrict__ cell_info,
                                                           uint8_t* __restrict__ actions, const int policy,
                                                           uint32_t* __restrict__ mt_state, const double epsilon) {
    __shared__ uint32_t key[kMtWords];
    const int env = (int)blockIdx.x, lane = (int)threadIdx.x;
    uint32_t* g = mt_state + (size_t)env * (kMtWords + 1);
    for (int i = lane; i < kMtWords; i += 64) key[i] = g[i];
    int pos = (int)g[kMtWords];                                   // wave-uniform
    pos = pos < 0 || pos > kMtWords ? kMtWords : pos;             // (a state this library did not write: twist first)
    __syncthreads();

    const bool agent = lane < p.N;
    const size_t t = (size_t)env * p.N + (size_t)(agent ? lane : 0);
    const bool live = agent && !(st.terminated[t] || st.truncated[t]);
    uint32_t free_dirs = 0u, chosen = (uint32_t)CCX_K_ABSENT;
    if (live) chosen = scripted_choice(p, st, cell_info, env, lane, policy, free_dirs);
    const uint64_t live_b = __ballot(live);

    auto next32 = [&]() -> uint32_t {                             // every lane computes the same word
        if (pos == kMtWords) {
            for (int c = 0; c < kMtWords; c += 64) {
                const int i = c + lane;
                uint32_t v = 0u;
                if (i < kMtWords) {
                    const uint32_t y = (key[i] & 0x80000000u) | (key[i + 1 == kMtWords ? 0 : i + 1] & 0x7fffffffu);
                    v = key[i + 397 < kMtWords ? i + 397 : i - 227] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
                }
                __syncthreads();
                if (i < kMtWords) key[i] = v;
                __syncthreads();
            }
            pos = 0;
        }
        uint32_t y = key[pos++];
        y ^= y >> 11;
        y ^= (y << 7) & 0x9d2c5680u;
        y ^= (y << 15) & 0xefc60000u;
        y ^= y >> 18;
        return y;
    };

    if (epsilon > 0.0) {                                          // (:49: the stream is untouched when randomness_factor is 0)
        for (int a = 0; a < p.N; ++a) {
            if (!((live_b >> a) & 1ull)) continue;                // only env.agents are asked
            const uint32_t w0 = next32() >> 5, w1 = next32() >> 6;
            const double u = ((double)w0 * 67108864.0 + (double)w1) / 9007199254740992.0;
            if (!(u < epsilon)) continue;
            const uint32_t valid = (uint32_t)__shfl((int)free_dirs, a, 64) | 16u;   // ascending list incl. wait
            const uint32_t rng = (uint32_t)__popc(valid) - 1u;
            uint32_t idx = 0u;
            if (rng != 0u) {
                uint32_t mask = rng;
                mask |= mask >> 1;
                mask |= mask >> 2;
                do idx = next32() & mask; while (idx > rng);
            }
            uint32_t rest = valid;
            for (uint32_t k = 0; k < idx; ++k) rest &= rest - 1u;  // drop the idx lowest entries
            if (lane == a) chosen = (uint32_t)__ffs((int)rest) - 1u;
        }
    }
    if (agent) actions[t] = (uint8_t)chosen;
    __syncthreads();
    for (int i = lane; i < kMtWords; i += 64) g[i] = key[i];
    if (lane == 0) g[kMtWords] = (uint32_t)pos;
}

// numpy's RandomState(seed) for an integer seed = init_genrand (numpy/random/src/mt19937/mt19937.c: mt19937_seed);
// pos = 624: the first draw twists.
__global__ void policy_stream_seed_kernel(uint32_t* __restrict__ mt_state, const uint32_t* __restrict__ seeds,
                                          const uint32_t seed_all, const int E) {
    const int env = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (env >= E) return;
    uint32_t* g = mt_state + (size_t)env * (kMtWords + 1);
    uint32_t s = seeds ? seeds[env] : seed_all;
    for (int i = 0; i < kMtWords; ++i) {
        g[i] = s;
        s = 1812433253u * (s ^ (s >> 30)) + (uint32_t)i + 1u;
    }
    g[kMtWords] = (uint32_t)kMtWords;
}

hipError_t launch_policy_stream_seed(hipStream_t stream, uint32_t* mt_state, const uint32_t* seeds_dev, uint32_t seed_all,
                                     int E) {
    if (E == 0) return hipSuccess;
    hipLaunchKernelGGL(policy_stream_seed_kernel, dim3((unsigned)((E + 63) / 64)), dim3(64), 0, stream, mt_state, seeds_dev,
                       seed_all, E);
    return hipGetLastError();
}

hipError_t launch_policy_stream_actions(hipStream_t stream, const KParams& p, const KState& st,
                                        const unsigned long long* cell_info, uint8_t* actions, int policy,
                                        uint32_t* mt_state, double epsilon) {
    if (p.E == 0) return hipSuccess;
    hipLaunchKernelGGL(policy_stream_kernel, dim3((unsigned)p.E), dim3(64), 0, stream, p, st, cell_info, actions, policy,
                       mt_state, epsilon);
    return hipGetLastError();
}

hipError_t launch_greedy_actions(hipStream_t stream, const KParams& p, const KState& st,
                                 const unsigned long long* cell_info, uint8_t* actions, int policy) {
    const size_t total = (size_t)p.E * p.N;
    if (total == 0) return hipSuccess;
    const unsigned blocks = (unsigned)((total + 255) / 256);
    hipLaunchKernelGGL(greedy_actions_kernel, dim3(blocks), dim3(256), 0, stream, p, st, cell_info, actions, policy);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// CCX_CHECK_INPUTS (opt-in): what the reference's _check_action_and_agent_validity raises on
// (collectivecrossing.py:685-711: action not in 0..4, agent id not in the env), restated for the array
// inputs of ccx_step / ccx_rollout: an action byte must be 0..4 or CCX_ACTION_ABSENT, a move-order row
// must name every slot exactly once.  Counted here, reported as CCX_EINVAL by the next synchronising
// call; a separate elementwise kernel so that the hot kernel carries none of it.
// ---------------------------------------------------------------------------------------------
__global__ void check_inputs_kernel(const uint8_t* __restrict__ actions, const uint8_t* __restrict__ order,
                                    const size_t rows, const int N, unsigned long long* bad) {
    const size_t r = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t bad_a = 0, bad_o = 0;
    if (r < rows) {
        const uint8_t* a = actions + r * (size_t)N;
        for (int i = 0; i < N; ++i) bad_a += (a[i] > 4u && a[i] != (uint8_t)CCX_K_ABSENT) ? 1u : 0u;
        if (order) {
            const uint8_t* o = order + r * (size_t)N;
            unsigned long long seen = 0;
            for (int i = 0; i < N; ++i) {
                const uint32_t v = o[i];
                if (v >= (uint32_t)N || ((seen >> v) & 1ull)) bad_o = 1u;
                else seen |= 1ull << v;
            }
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        bad_a += __shfl_xor(bad_a, off, 64);
        bad_o += __shfl_xor(bad_o, off, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        if (bad_a) atomicAdd(&bad[0], (unsigned long long)bad_a);
        if (bad_o) atomicAdd(&bad[1], (unsigned long long)bad_o);
    }
}

hipError_t launch_check_inputs(hipStream_t stream, const uint8_t* actions, const uint8_t* order, size_t rows,
                               int N, unsigned long long* bad) {
    if (rows == 0) return hipSuccess;
    const unsigned blocks = (unsigned)((rows + 255) / 256);
    hipLaunchKernelGGL(check_inputs_kernel, dim3(blocks), dim3(256), 0, stream, actions, order, rows, N, bad);
    return hipGetLastError();
}

}  // namespace ccx
