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
    const int cx = st.x[t], cy = st.y[t];
    const int Wp = p.W + 3;
    const uint32_t cw = (uint32_t)cell_info[(cy + 1) * Wp + cx + 1];
    const uint32_t nv = cw & 0xFu;   // enterable neighbours
    uint32_t cand = greedy_candidates(p, i < p.Nb, cx, cy);
    const size_t base = (size_t)env * p.N;
    bool waits = false;
    if (policy == CCX_K_POLICY_WAITING && i < p.Nb && !(cw & kCellInTram)) {   // waiting_policy.py:92-100
        for (int b = p.Nb; b < p.N; ++b)                                  // :119-129
            waits |= !(st.terminated[base + b] || st.truncated[base + b]) && st.y[base + b] != p.edy;
    }
    // directions this agent could move in: the neighbour bit of its cell and no other ACTIVE agent on the target
    // (env._is_move_valid, collectivecrossing.py:345-369) -- what both the policy's fallback list and an
    // epsilon draw choose from
    uint32_t free_dirs = 0u;
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
    if (p.eps_thr) {   // epsilon-greedy with the device's counter-based draws (include/ccx.h: ccx_set_policy_epsilon)
        const uint32_t u = random_word(p.rng_lo, p.rng_hi ^ kEpsStream, (uint32_t)(p.env_offset + env), (uint32_t)st.episode[env],
                                       (uint32_t)st.step_count[env], (uint32_t)i);
        if (u < p.eps_thr) chosen = explore_action(u, free_dirs);
    }
    actions[t] = (uint8_t)chosen;
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
