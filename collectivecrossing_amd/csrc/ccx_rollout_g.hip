// ccx_rollout_g.hip -- the rollout kernel instantiations of ONE lane-group size (compiled once per CCX_GLOG = 0..6:
// csrc/Makefile) and their host-side launch.  See ccx_rollout_dev.h / ccx_rollout_body.inc for the kernel itself.
#include "ccx_rollout_dev.h"
#include <algorithm>

#ifndef CCX_GLOG
#error "compile with -DCCX_GLOG=0..6"
#endif

namespace ccx {

template <int GLOG, bool PAIR, int OUT, bool OCC, bool PLAIN>
static hipError_t launch_rollout_v(const LaunchShape& ls, hipStream_t stream, const KParams& p,
                                   const KState& st, const unsigned long long* cell_info,
                                   const uint8_t* actions, const uint8_t* order, int K,
                                   int auto_reset, const uint8_t* pool, const KOut& out,
                                   unsigned long long* counters, int policy, uint8_t* actions_out) {
    const void* entry = rollout_entry<GLOG, PAIR, OUT, OCC, PLAIN>();
    if (ls.lds_bytes > 60 * 1024) {
        // big grids / many envs per tile need more than the default 64 KiB of dynamic LDS
        hipError_t e = hipFuncSetAttribute(entry, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
    }
    dim3 grid(ls.num_blocks), block(64 * ls.waves_per_block * (OUT ? 1 + ls.writers : 1));
    void* args[] = {const_cast<KParams*>(&p), const_cast<KState*>(&st), &cell_info, &actions, &order, &K, &auto_reset,
                    &pool, const_cast<KOut*>(&out), &counters, &policy, &actions_out};
    return hipLaunchKernel(entry, grid, block, args, ls.lds_bytes, stream);
}

template <int GLOG>
static hipError_t launch_rollout_g(const LaunchShape& ls, hipStream_t stream, const KParams& p,
                                   const KState& st, const unsigned long long* cell_info,
                                   const uint8_t* actions, const uint8_t* order, int K,
                                   int auto_reset, const uint8_t* pool, const KOut& out,
                                   unsigned long long* counters, int policy, uint8_t* actions_out) {
    const bool pair = (p.N % 2) == 0;
    const bool want_out = out.obs || out.reward || out.agent_flags || out.env_flags || out.obs_compact || actions_out;
    // edge iterations: the tile regions of the observation output share 128-byte lines with their neighbours
    const size_t tile_region = (size_t)p.EW * p.N * (6 + 4 * p.N) * 4u, slab = (size_t)p.E * p.N * (6 + 4 * p.N) * 4u;
    const bool edges = out.obs && (((tile_region | slab) & 127u) != 0 || (reinterpret_cast<uintptr_t>(out.obs) & 127u) != 0);
    // (3: the slab stride itself is not a multiple of 128 bytes -- `lead` per step, ccx_rollout_body.inc: VARLEAD -- and a row
    //  writer has more store iterations than it caches source addresses for: 50 agents x 532 envs 0.50 -> 0.88 of the peak.  Narrow
    //  rows, all of whose iterations are register-cached, are faster with the launch-wide layout: 3 agents 0.28 vs 0.19.)
    const int row_writers = std::max(1, ls.writers - ((p.writer0_small && ls.writers > 1) ? 1 : 0));
    const int its = ((p.units_per_wave >> (pair ? 1 : 0)) + 63) / 64;
    const bool long_rows = (its + row_writers - 1) / row_writers > kFastObsIters;
    const int outm = want_out ? (edges ? ((slab & 127u) != 0 && long_rows ? 3 : 2) : 1) : 0;
    const bool plain = order == nullptr && policy == 0 && p.user_tables == 0u;   // (user reward / terminated tables: the general instantiations)
#define CCX_GO2(P_, O_, C_)                                                                                  \
    return plain ? launch_rollout_v<GLOG, P_, O_, C_, true>(ls, stream, p, st, cell_info, actions, order, K, \
                                                            auto_reset, pool, out, counters, policy, actions_out) \
                 : launch_rollout_v<GLOG, P_, O_, C_, false>(ls, stream, p, st, cell_info, actions, order, K, \
                                                             auto_reset, pool, out, counters, policy, actions_out)
#define CCX_GO(P_, C_)                \
    switch (outm) {                   \
    case 3: CCX_GO2(P_, 3, C_);       \
    case 2: CCX_GO2(P_, 2, C_);       \
    case 1: CCX_GO2(P_, 1, C_);       \
    default: CCX_GO2(P_, 0, C_);      \
    }
    if (pair && ls.occ) { CCX_GO(true, true) }
    else if (pair) { CCX_GO(true, false) }
    else if (ls.occ) { CCX_GO(false, true) }
    else { CCX_GO(false, false) }
#undef CCX_GO2
#undef CCX_GO
}

template <int GLOG>
static int blocks_per_cu_g(const LaunchShape& ls, bool pair) {
    const int threads = 64 * ls.waves_per_block * (1 + ls.writers);
    int n = 0;
    hipError_t e = hipSuccess;
#define CCX_OCCQ(P_, C_)                                                                          \
    do {                                                                                          \
        const void* f = rollout_entry<GLOG, P_, 1, C_, true>();       \
        if (ls.lds_bytes > 60 * 1024)                                                             \
            (void)hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
        e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, f, threads, ls.lds_bytes);           \
    } while (0)
    if (pair && ls.occ) CCX_OCCQ(true, true);
    else if (pair) CCX_OCCQ(true, false);
    else if (ls.occ) CCX_OCCQ(false, true);
    else CCX_OCCQ(false, false);
#undef CCX_OCCQ
    if (e != hipSuccess) { (void)hipGetLastError(); return 0; }
    return n;
}

#define CCX_CAT2(a, b) a##b
#define CCX_CAT(a, b) CCX_CAT2(a, b)
hipError_t CCX_CAT(launch_rollout_glog, CCX_GLOG)(const LaunchShape& ls, hipStream_t stream, const KParams& p, const KState& st,
                                                 const unsigned long long* cell_info, const uint8_t* actions, const uint8_t* order,
                                                 int K, int auto_reset, const uint8_t* pool, const KOut& out,
                                                 unsigned long long* counters, int policy, uint8_t* actions_out) {
    return launch_rollout_g<CCX_GLOG>(ls, stream, p, st, cell_info, actions, order, K, auto_reset, pool, out, counters, policy, actions_out);
}
int CCX_CAT(blocks_per_cu_glog, CCX_GLOG)(const LaunchShape& ls, bool pair) { return blocks_per_cu_g<CCX_GLOG>(ls, pair); }

}  // namespace ccx
