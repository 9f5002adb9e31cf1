// ccx_kernels.hip -- gfx950 (MI355X, CDNA4) kernels of libccx.
//
// Replaces, for E independent envs at once, the reference's CollectiveCrossingEnv.step
// (src/collectivecrossing/collectivecrossing.py:161-261) and the strategy methods it calls
// (rewards.py:44-182, terminateds.py:40-82, truncateds.py:40-61, observations.py:43-94).
//
// Mapping (wave64): one LANE per (env, agent).  An env owns a group of G = 2^GLOG >= N consecutive
// lanes; a SIM wavefront carries EW <= 64/G envs and keeps their state in registers for all K
// steps of a rollout.  Env tiles never talk to each other (no inter-workgroup traffic).
//
// A workgroup is a small TEAM of wavefronts per env tile (1 sim + 1..3 writers), split by role along
// the reference's own phase boundary (collectivecrossing.py:197-212 vs :214-261):
//   * the SIM wave does the state transition: action decode, ordered move resolution, arrival /
//     termination / truncation flags, auto-reset.  It issues no global stores at all; per step it
//     hands 16 bytes per lane (cell word, distance word, flag byte, env byte) to its partner
//     through a double-buffered LDS tile and one s_barrier.
//   * the WRITER wave materialises the outputs of that step: the f64 reward, the flag bytes and
//     the observation rows (the byte-dominant part, 4*(6+4N) B per agent-step).  It stages
//     (x, y, type, active) as float4 per agent in LDS; the tile's rows form ONE contiguous region
//     of the [E][N][L] output, written with full-width global_store_dwordx4 whose two 8-byte
//     halves come from LDS addresses that never change (kept in VGPRs / a u16 LDS table).
//   The two waves sit on different SIMDs of one CU, so a step costs max(sim, writer) instead of
//   the sum, and the stores drain asynchronously: the writer never waits on vmcnt.
//
// Ordered move resolution without a serial agent loop ("agent k sees earlier agents at their new
// cell and later agents at their old cell", collectivecrossing.py:197-202,536-541):
//   1. every lane proposes its target cell in parallel; validity of the target (bounds, walls,
//      door) and everything else the step needs to know about a cell is ONE ds_read_b64 from a
//      per-cell table precomputed on the host (see "per-cell geometry table" below);
//   2. every active agent ORs its move-rank bit into a per-env OCCUPANCY bit table (LDS, one mask
//      per cell) at its current cell and, if its proposal is legal, into a PROPOSAL table at the
//      target cell; reading both tables at the target gives, in O(1) per agent,
//   3. C = earlier ranks standing on my target, P = earlier ranks proposing my target, and whether
//      a later rank still stands there (grids whose tables exceed LDS fall back to an all-pairs
//      compare through a 256-byte exchange tile, O(G) per agent);
//   4. with M = mask of earlier ranks that DID move, rank k moves iff ((M & P) | (~M & C)) == 0.
//      M is a wave ballot; F(M) has a unique fixed point reached in <= N rounds (bit k of F
//      depends on bits < k only); the loop is skipped when no lane depends on an earlier rank and
//      exits as soon as two consecutive ballots agree.
//
// Everything is integer / index work; the only floating-point operation on the path is ONE f64
// multiply per reward (compiled with -ffp-contract=off).  No MFMA on purpose.
#include "ccx_rollout_dev.h"

namespace ccx {

// totals[q] = sum over the per-tile partial slots (q = 0..5); one workgroup
__global__ void __launch_bounds__(256) reduce_counters_kernel(unsigned long long* counters, int slots) {
    __shared__ unsigned long long part[4][6];
    unsigned long long acc[6] = {0, 0, 0, 0, 0, 0};
    for (int t = threadIdx.x; t < slots; t += 256) {
        const unsigned long long* slot = counters + kCounterTotals + (size_t)t * kCounterSlot;
#pragma unroll
        for (int q = 0; q < 6; ++q) acc[q] += slot[q];
    }
#pragma unroll
    for (int q = 0; q < 6; ++q) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) acc[q] += __shfl_xor(acc[q], off, 64);
        if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6][q] = acc[q];
    }
    __syncthreads();
    if (threadIdx.x < 6)
        counters[threadIdx.x] = part[0][threadIdx.x] + part[1][threadIdx.x] + part[2][threadIdx.x] +
                                part[3][threadIdx.x];
}

hipError_t launch_reduce_counters(hipStream_t stream, unsigned long long* counters, int slots) {
    hipLaunchKernelGGL(reduce_counters_kernel, dim3(1), dim3(256), 0, stream, counters, slots);
    return hipGetLastError();
}

// DefaultObservation of the current state (what reset() returns, collectivecrossing.py:153-159)
// compact != nullptr: the rows come from compact observations [E][N][4] (the inverse of CCX_OBS_COMPACT: E =
// envs x steps) instead of the handle's state -- the same gather, hence the same bits.
template <int GLOG, bool PAIR>
__global__ void __launch_bounds__(256)
observe_kernel(const KParams p, const KState st, float* __restrict__ obs, const float4* __restrict__ compact) {
    constexpr int G = 1 << GLOG;
    extern __shared__ __align__(16) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const int wib = threadIdx.x >> 6;
    const int wave = blockIdx.x * p.waves_per_block + wib;
    const int g = lane >> GLOG, i = lane & (G - 1);
    const int env0 = wave * p.EW, env = env0 + g;
    const bool valid = (g < p.EW) && (env < p.E) && (i < p.N);
    const int N = p.N, L = 6 + 4 * N;
    WaveLds* wl = reinterpret_cast<WaveLds*>(smem) + wib;
    uint16_t* table = reinterpret_cast<uint16_t*>(smem + sizeof(WaveLds) * p.waves_per_block);
    build_obs_table<GLOG>(table, p);
    init_wave_consts(wl, p, lane);
    float4 me = make_float4(0.0f, 0.0f, (i < p.Nb) ? 0.0f : 1.0f, 0.0f);
    if (valid) {
        const size_t idx = (size_t)env * N + i;
        if (compact) {
            me = compact[idx];
        } else {
            me.x = (float)st.x[idx];
            me.y = (float)st.y[idx];
            me.w = st.active[idx] != 0 ? 1.0f : 0.0f;
        }
    }
    wl->slot[lane] = me;
    __syncthreads();
    int envs_here = p.E - env0;
    envs_here = envs_here < 0 ? 0 : (envs_here > p.EW ? p.EW : envs_here);
    const int units = envs_here * N * (3 + 2 * N);
    emit_obs<PAIR>(wl, table, reinterpret_cast<char*>(obs + (size_t)env0 * N * L), 0,
                   PAIR ? (units >> 1) : units, lane);
}

// (re)start masked envs from their pool entry (reset() :97-150, placements precomputed on host)
__global__ void reset_from_pool_kernel(const KParams p, const KState st,
                                       const uint8_t* __restrict__ env_mask,
                                       const uint8_t* __restrict__ pool) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t total = (size_t)p.E * p.N;
    if (t >= total) return;
    const int env = (int)(t / p.N), i = (int)(t % p.N);
    if (env_mask && !env_mask[env]) return;
    const unsigned long long P = (unsigned long long)p.pool_size;
    const unsigned long long gi = (unsigned long long)(p.env_offset + env) % P;
    const unsigned long long ep = (unsigned long long)st.episode[env] % P;
    const unsigned long long pi = (gi + ep * (unsigned long long)p.pool_stride) % P;
    const uint8_t* src = pool + ((size_t)pi * p.N + i) * 2;
    st.x[t] = src[0];
    st.y[t] = src[1];
    st.active[t] = 1;
    st.terminated[t] = 0;
    st.truncated[t] = 0;
    if (i == 0) st.step_count[env] = 0;
}

// ---------------------------------------------------------------------------------------------
// host-side dispatch
// ---------------------------------------------------------------------------------------------
// the rollout kernels live in ccx_rollout_g.hip, one translation unit per lane-group size
#define CCX_DECL_GLOG(n)                                                                                              \
    hipError_t launch_rollout_glog##n(const LaunchShape&, hipStream_t, const KParams&, const KState&,                 \
                                      const unsigned long long*, const uint8_t*, const uint8_t*, int, int,            \
                                      const uint8_t*, const KOut&, unsigned long long*, int, uint8_t*);               \
    int blocks_per_cu_glog##n(const LaunchShape&, bool);
CCX_DECL_GLOG(0) CCX_DECL_GLOG(1) CCX_DECL_GLOG(2) CCX_DECL_GLOG(3) CCX_DECL_GLOG(4) CCX_DECL_GLOG(5) CCX_DECL_GLOG(6)
#undef CCX_DECL_GLOG

// workgroups of the rollout kernel (with outputs) one CU holds at once; 0 = unknown
int rollout_blocks_per_cu(const LaunchShape& ls, int agents) {
    const bool pair = (agents % 2) == 0;
    switch (ls.glog) {
#if !defined(CCX_ONLY_GLOG) || CCX_ONLY_GLOG == 0
    case 0: return blocks_per_cu_glog0(ls, pair);
#endif
#if !defined(CCX_ONLY_GLOG) || CCX_ONLY_GLOG == 1
    case 1: return blocks_per_cu_glog1(ls, pair);
#endif
#if !defined(CCX_ONLY_GLOG) || CCX_ONLY_GLOG == 2
    case 2: return blocks_per_cu_glog2(ls, pair);
#endif
#if !defined(CCX_ONLY_GLOG) || CCX_ONLY_GLOG == 3
    case 3: return blocks_per_cu_glog3(ls, pair);
#endif
#if !defined(CCX_ONLY_GLOG) || CCX_ONLY_GLOG == 4
    case 4: return blocks_per_cu_glog4(ls, pair);
#endif
#if !defined(CCX_ONLY_GLOG) || CCX_ONLY_GLOG == 5
    case 5: return blocks_per_cu_glog5(ls, pair);
#endif
#if !defined(CCX_ONLY_GLOG) || CCX_ONLY_GLOG == 6
    default: return blocks_per_cu_glog6(ls, pair);
#endif
    }
    return 0;
}

template <int GLOG>
static hipError_t launch_observe_g(const LaunchShape& ls, hipStream_t stream, const KParams& p,
                                   const KState& st, float* obs, const float4* compact, unsigned blocks) {
    const bool pair = (p.N % 2) == 0;
    dim3 grid(blocks), block(64 * ls.waves_per_block);
    if (pair)
        hipLaunchKernelGGL((observe_kernel<GLOG, true>), grid, block, ls.lds_bytes_observe, stream, p, st, obs, compact);
    else
        hipLaunchKernelGGL((observe_kernel<GLOG, false>), grid, block, ls.lds_bytes_observe, stream, p, st, obs, compact);
    return hipGetLastError();
}

hipError_t launch_rollout(const LaunchShape& ls, hipStream_t stream, const KParams& p,
                          const KState& st, const unsigned long long* cell_info,
                          const uint8_t* actions, const uint8_t* order, int K,
                          int auto_reset, const uint8_t* pool, const KOut& out,
                          unsigned long long* counters, int policy, uint8_t* actions_out) {
    switch (ls.glog) {
#if !defined(CCX_ONLY_GLOG) || CCX_ONLY_GLOG == 0
    case 0: return launch_rollout_glog0(ls, stream, p, st, cell_info, actions, order, K, auto_reset, pool, out, counters, policy, actions_out);
#endif
#if !defined(CCX_ONLY_GLOG) || CCX_ONLY_GLOG == 1
    case 1: return launch_rollout_glog1(ls, stream, p, st, cell_info, actions, order, K, auto_reset, pool, out, counters, policy, actions_out);
#endif
#if !defined(CCX_ONLY_GLOG) || CCX_ONLY_GLOG == 2
    case 2: return launch_rollout_glog2(ls, stream, p, st, cell_info, actions, order, K, auto_reset, pool, out, counters, policy, actions_out);
#endif
#if !defined(CCX_ONLY_GLOG) || CCX_ONLY_GLOG == 3
    case 3: return launch_rollout_glog3(ls, stream, p, st, cell_info, actions, order, K, auto_reset, pool, out, counters, policy, actions_out);
#endif
#if !defined(CCX_ONLY_GLOG) || CCX_ONLY_GLOG == 4
    case 4: return launch_rollout_glog4(ls, stream, p, st, cell_info, actions, order, K, auto_reset, pool, out, counters, policy, actions_out);
#endif
#if !defined(CCX_ONLY_GLOG) || CCX_ONLY_GLOG == 5
    case 5: return launch_rollout_glog5(ls, stream, p, st, cell_info, actions, order, K, auto_reset, pool, out, counters, policy, actions_out);
#endif
#if !defined(CCX_ONLY_GLOG) || CCX_ONLY_GLOG == 6
    case 6: return launch_rollout_glog6(ls, stream, p, st, cell_info, actions, order, K, auto_reset, pool, out, counters, policy, actions_out);
#endif
    }
    return hipErrorInvalidValue;
}

static hipError_t launch_observe_any(const LaunchShape& ls, hipStream_t stream, const KParams& p, const KState& st,
                                     float* obs, const float4* compact, unsigned blocks) {
    switch (ls.glog) {
#if !defined(CCX_ONLY_GLOG) || CCX_ONLY_GLOG == 0
    case 0: return launch_observe_g<0>(ls, stream, p, st, obs, compact, blocks);
#endif
#if !defined(CCX_ONLY_GLOG) || CCX_ONLY_GLOG == 1
    case 1: return launch_observe_g<1>(ls, stream, p, st, obs, compact, blocks);
#endif
#if !defined(CCX_ONLY_GLOG) || CCX_ONLY_GLOG == 2
    case 2: return launch_observe_g<2>(ls, stream, p, st, obs, compact, blocks);
#endif
#if !defined(CCX_ONLY_GLOG) || CCX_ONLY_GLOG == 3
    case 3: return launch_observe_g<3>(ls, stream, p, st, obs, compact, blocks);
#endif
#if !defined(CCX_ONLY_GLOG) || CCX_ONLY_GLOG == 4
    case 4: return launch_observe_g<4>(ls, stream, p, st, obs, compact, blocks);
#endif
#if !defined(CCX_ONLY_GLOG) || CCX_ONLY_GLOG == 5
    case 5: return launch_observe_g<5>(ls, stream, p, st, obs, compact, blocks);
#endif
#if !defined(CCX_ONLY_GLOG) || CCX_ONLY_GLOG == 6
    case 6: return launch_observe_g<6>(ls, stream, p, st, obs, compact, blocks);
#endif
    }
    return hipErrorInvalidValue;
}

hipError_t launch_observe(const LaunchShape& ls, hipStream_t stream, const KParams& p,
                          const KState& st, float* obs) {
    return launch_observe_any(ls, stream, p, st, obs, nullptr, (unsigned)ls.num_blocks);
}

hipError_t launch_expand(const LaunchShape& ls, hipStream_t stream, const KParams& p, const float* compact,
                         long long rows, float* obs) {
    if (rows <= 0) return hipSuccess;
    KParams q = p;                 // same lane layout, `rows` envs instead of the handle's E
    q.E = (int)rows;
    const long long per_block = (long long)p.EW * p.waves_per_block;
    const long long blocks = (rows + per_block - 1) / per_block;
    if (rows > 0x7FFFFFFFll || blocks > 0x7FFFFFFFll) return hipErrorInvalidValue;
    return launch_observe_any(ls, stream, q, KState{}, obs, reinterpret_cast<const float4*>(compact), (unsigned)blocks);
}

// Pace calibration probe (ccx_api.hip: calibrate_pace): a plain fill of the caller's own trajectory buffer -- one
// workgroup per 16 KiB, four 16-byte stores per lane, cached stores like a library fill (measured: streaming `nt` stores
// issued flat out, without the rollout's pacing, drain 25 % slower than the paced rollout does and are no yardstick for
// it; a plain fill lands within a few per cent of it on every box seen).  What it writes is overwritten by the rollout
// that follows.
__global__ void __launch_bounds__(256) write_probe_kernel(v4f* __restrict__ dst, const size_t n16) {
    const v4f filler = {0.0f, 0.0f, 0.0f, 0.0f};
    const size_t base = (size_t)blockIdx.x * 1024u + threadIdx.x;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const size_t q = base + (size_t)k * 256u;
        if (q < n16) dst[q] = filler;
    }
}

hipError_t launch_write_probe(hipStream_t stream, void* dst, size_t bytes, int) {
    if (bytes < 16) return hipSuccess;
    const size_t n16 = bytes / 16, blocks = (n16 + 1023) / 1024;
    hipLaunchKernelGGL(write_probe_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, reinterpret_cast<v4f*>(dst), n16);
    return hipGetLastError();
}

hipError_t launch_reset_from_pool(hipStream_t stream, const KParams& p, const KState& st,
                                  const uint8_t* env_mask, const uint8_t* pool) {
    const size_t total = (size_t)p.E * p.N;
    if (total == 0) return hipSuccess;
    const unsigned blocks = (unsigned)((total + 255) / 256);
    hipLaunchKernelGGL(reset_from_pool_kernel, dim3(blocks), dim3(256), 0, stream, p, st, env_mask, pool);
    return hipGetLastError();
}

}  // namespace ccx
