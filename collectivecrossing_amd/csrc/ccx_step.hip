// ccx_step.hip -- the SHORT-LAUNCH kernel: CollectiveCrossingEnv.step itself (collectivecrossing.py:161-261), K = 1..16
// env-steps per launch, for callers with a policy in the loop (the reference's literal API: one step() per call).
//
// ccx::rollout_kernel is built for hundreds of steps per launch: LDS hand-off ring, sequence words, pace controller, 450
// bytes of kernel arguments read through the scalar cache, ~350 prologue instructions, tables staged for every role.  A
// one-step launch of it takes 5.8 us in a replayed HIP graph for 0.7 us of bytes.  What a launch costs
// on this chip (profiles/r04_launch_floor.txt, graph chains): an EMPTY kernel 1.56 us per dependent launch, a load ->
// LDS -> store skeleton 1.75 us, the same plus 5.6 MB of streaming stores 2.45 us: that is the floor of a C2 step.
// This kernel is the same state transition and the same outputs (bit for bit: tests/test_gpu_step_kernel.py) with
// everything that serves long launches removed:
//   * a workgroup is ONE tile: a SIM wave (state in, moves, flags, rewards, flag bytes, state out) and, when observation
//     rows are asked for, 1-3 ROW waves that do nothing but the gather (observations.py:43-94).  They meet at ONE
//     LDS-only barrier per step (float4 per agent, double-buffered); no ring, no sequence words, no pacing, no controller.
//   * every wave issues all its global loads in its first instructions -- the sim wave state + actions + cell table, a
//     row wave the address-table words of its own store iterations (straight into registers: no LDS copy of that table)
//     -- so the whole entry is ONE memory round trip, and the row waves' set-up runs beside the sim wave's step.
//   * the handle's state lives in ONE slab (ccx_kernels.h: StateSlab): one base pointer, offsets from E and N; plain
//     kernel arguments (two cache lines), 32-bit lane offsets from scalar bases.
//   * state write-back behind the last hand-off: everything drains together.
// Tiles are smaller than the rollout's (a short launch is bound by latency, not by issue).
//
// Scope: actions from a tensor with or without a move order (no in-kernel policy), K <= 16 (one burst of action loads), LDS
// occupancy tables (grids whose tables do not fit take the rollout kernel: ccx_api.hip decides).  Auto-reset from the
// pool is supported (ccx_rollout with few steps).
// The first 14 argument dwords (state slab, actions, both tables, obs, E, shape words, max_steps) are preloaded into SGPRs
// (csrc/Makefile: -amdgpu-kernarg-preload-count=14): -0.05 us per step, measured on this kernel.
#include "ccx_rollout_dev.h"

namespace ccx {

#ifdef CCX_TSTAMPS   // diagnostic (profiles/scratch/step_tstamps.py): s_memrealtime (10-ns ticks) at fixed points of tile 0
#define CCX_ST(q) do { if (counters && tile == 0 && lane == 0) { unsigned long long t_; \
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory"); counters[8 + (q)] = t_; } } while (0)
#define CCX_ST_CLK0() unsigned long long clk0_ = 0; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(clk0_)::"memory")
#define CCX_ST_CLK1() do { if (counters && tile == 0 && lane == 0) { unsigned long long t_; \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory"); counters[6] = t_ - clk0_; } } while (0)   /* shader clocks ST(0) .. ST(5) */
#define CCX_ST_DRAIN(q) do { if (counters && tile == 0) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); CCX_ST(q); } } while (0)
#else
#define CCX_ST(q) do { } while (0)
#define CCX_ST_CLK0() do { } while (0)
#define CCX_ST_CLK1() do { } while (0)
#define CCX_ST_DRAIN(q) do { } while (0)
#endif

#ifndef CCX_STEP_STORE_BITS      // cache policy of the observation-row stores of the step kernel
#ifdef CCX_STEP_PLAIN_ROWS       // (experiment: cached stores)
#define CCX_STEP_STORE_BITS ""
#else
#define CCX_STEP_STORE_BITS "sc1 nt"
#endif
#endif
// row stores: wave-uniform base in SGPRs + 32-bit lane offset (a tile's rows span < 4 GiB)
__device__ __forceinline__ void step_store_obs(v4f v, const char* base, uint32_t voff) {
    asm volatile("global_store_dwordx4 %0, %1, %2 " CCX_STEP_STORE_BITS "\n\ts_nop 1" ::"v"(voff), "v"(v), "s"(base) : "memory");
}
__device__ __forceinline__ void step_store_obs(v2f v, const char* base, uint32_t voff) {
    asm volatile("global_store_dwordx2 %0, %1, %2 " CCX_STEP_STORE_BITS ::"v"(voff), "v"(v), "s"(base) : "memory");
}

// store iterations of a row wave whose table words / LDS reads are in flight together: 8 for small lane groups (C2: a row
// wave has ~5 iterations), 16 for the large ones (C3 / C5: 10-11 iterations per row wave, one batch instead of two -- the
// second batch's table words would be a memory round trip of their own)
template <int GLOG> struct RowBatch { static constexpr int value = GLOG <= 3 ? 8 : 16; };

// K1: the launch is ONE env-step (ccx_step): one action load per lane instead of a burst of sixteen
// ORD: the caller passed a move order (collectivecrossing.py:197: agents move in the order of `action_dict`)
template <int GLOG, bool PAIR, bool K1, bool ORD>
__global__ void __launch_bounds__(512)
step_kernel(uint8_t* __restrict__ st_base,                      // StateSlab layout
            const uint8_t* __restrict__ actions,                // u8 [K][E][N]
            const unsigned long long* __restrict__ cell_info,   // per-cell geometry table
            const uint16_t* __restrict__ obs_table,             // u16 LDS source address per float2 unit
            float* __restrict__ obs,                            // f32 [K][E][N][L] or null
            const int E,
            const uint32_t shape,                               // N | Nb << 8 | EW << 16 | K << 24
            const uint32_t grid,                                // cells of the padded grid | (W + 3) << 16 | row waves per tile << 24 | (reward table ? 1 : 0) << 31
            const int max_steps,
            double* __restrict__ reward, uint8_t* __restrict__ agent_flags, uint8_t* __restrict__ env_flags,
            float* __restrict__ obs_compact, unsigned long long* __restrict__ counters,
            const uint8_t* __restrict__ pool, const uint32_t pool_size, const uint32_t pool_stride,
            const uint32_t env_offset_mod_pool, const int dc, const int div, const int dl, const int dr,
            const int term_all, const int auto_reset,
            const double rA, const double rB, const double rC, const double rF,
            const double* __restrict__ reward_table,                   // user reward table f64 [2][cells] or null
            const uint8_t* __restrict__ order) {                       // u8 [K][E][N]: slot of the agent that moves k-th, or null
    using mask_t = typename GroupMask<GLOG>::type;
    constexpr int G = 1 << GLOG;
    constexpr uint32_t msz = sizeof(mask_t);
    constexpr uint32_t TS = (GLOG == 6) ? 1u : 0u;     // a table entry {occ, prp} is 8 bytes (32-bit masks) or 16
    typedef mask_t mask2_t __attribute__((ext_vector_type(2)));
    extern __shared__ __align__(16) unsigned char smem[];

    const int lane = threadIdx.x & 63;
    const int wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);          // 0 = sim wave, 1.. = row waves
    const int N = (int)(shape & 0xFFu), Nb = (int)((shape >> 8) & 0xFFu), EW = (int)((shape >> 16) & 0xFFu);
    const int K = K1 ? 1 : (int)(shape >> 24);
    const uint32_t cells = grid & 0xFFFFu;
    const int Wp = (int)((grid >> 16) & 0xFFu);
    const int RW = (int)((grid >> 24) & 0x7Fu);                                // row waves of this launch (0: no observation rows)
    const bool has_rtab = (grid >> 31) != 0u;        // (a preloaded bit: the LDS layout must not wait for the pointer's scalar load)
    const int tile = (int)blockIdx.x;
    const int env0 = tile * EW;
    const int L = 6 + 4 * N;
    const uint32_t EN = (uint32_t)E * (uint32_t)N;
    // LDS of the workgroup: [cell table][occupancy / proposal tables][two staging slots (float4 per lane + row constants)]
    auto up16 = [](uint32_t v) { return (v + 15u) & ~15u; };
    const uint32_t off_rtab = up16(cells * 8u);                               // (the region exists only with a user reward table)
    const uint32_t off_occ = off_rtab + (has_rtab ? cells * 16u : 0u);
    const uint32_t occ_bytes = up16((uint32_t)EW * 2u * (cells + 1u) * msz);
    const uint32_t off_ws = off_occ + occ_bytes;
    const uint32_t off_xch = off_ws + 2u * (uint32_t)sizeof(WSlot);          // 64 words: move-order exchange (ORD)
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
    int envs_here = E - env0;
    envs_here = envs_here < 0 ? 0 : (envs_here > EW ? EW : envs_here);
    const size_t obs_stride = (size_t)EN * (size_t)L * 4u;
    constexpr uint32_t vbytes = PAIR ? 16u : 8u;
    constexpr int kRowBatch = RowBatch<GLOG>::value;

    // =========================================================================================================
    // ROW waves: the tile's observation rows (ONE contiguous region of [E][N][L]), laid out on 128-byte lines of
    // global memory (`lead`), gathered from the staging slot through the u16 address table (ccx_kernels.h:
    // obs_unit_addr).  Row wave w of RW takes the w-th contiguous share of the store iterations.
    // =========================================================================================================
    if (wib > 0) {
        const int w = wib - 1;
        const uint32_t n4 = (uint32_t)(envs_here * N * (3 + 2 * N)) >> (PAIR ? 1 : 0);     // vector units of the tile's rows
        const char* obs_s = reinterpret_cast<const char*>(obs) + (size_t)env0 * (size_t)N * (size_t)L * 4u;
        const uint32_t lead0 = (uint32_t)(reinterpret_cast<uintptr_t>(obs_s) & 127u) / vbytes;
        // my table words for the first batch of iterations, straight from global memory into registers
        auto table_word = [&](uint32_t q) -> uint32_t {           // PAIR: two u16 addresses; else one
            if constexpr (PAIR) return reinterpret_cast<const uint32_t*>(obs_table)[q];
            else return (uint32_t)obs_table[q];
        };
        uint32_t tw0[kRowBatch];
        {
            const uint32_t its = (n4 + lead0 + 63u) >> 6, per_w = (its + (uint32_t)RW - 1u) / (uint32_t)RW;
            const uint32_t it0 = (uint32_t)w * per_w;
#pragma unroll
            for (int j = 0; j < kRowBatch; ++j) {
                const uint32_t q = (uint32_t)lane + 64u * (it0 + j) - lead0;
                tw0[j] = table_word(q < n4 ? q : 0u);
            }
            // (pinned: left alone the compiler sinks these loads below the barrier, to their use -- a memory round trip on the
            //  row wave's path AFTER its release instead of beside the sim wave's work)
#pragma unroll
            for (int j = 0; j < kRowBatch; ++j) tw0[j] = in_vgpr(tw0[j]);
        }
        for (int s = 0; s < K; ++s) {
            const uint32_t lead = (uint32_t)(reinterpret_cast<uintptr_t>(obs_s) & 127u) / vbytes;
            const char* const base = obs_s - lead * vbytes;
            const uint32_t its = (n4 + lead + 63u) >> 6, per_w = (its + (uint32_t)RW - 1u) / (uint32_t)RW;
            const uint32_t it0 = (uint32_t)w * per_w, it_end = (it0 + per_w) < its ? (it0 + per_w) : its;
            const uint32_t wl_abs = lds0 + off_ws + ((uint32_t)s & 1u) * (uint32_t)sizeof(WSlot);
            lds_barrier();                                             // the sim wave has staged step s
            if (w == 0) CCX_ST(7);
            typedef __attribute__((address_space(3))) const v2f lds_f2;
            for (uint32_t itb = it0; itb < it_end; itb += kRowBatch) {
                uint32_t q[kRowBatch], tw[kRowBatch];
                const bool cached = itb == it0 && lead == lead0;       // wave-uniform
#pragma unroll
                for (int j = 0; j < kRowBatch; ++j) {
                    q[j] = (uint32_t)lane + 64u * (itb + j) - lead;    // wraps below the region
                    tw[j] = cached ? tw0[j] : table_word(q[j] < n4 ? q[j] : 0u);
                }
                v2f va[kRowBatch], vb[kRowBatch];
#pragma unroll
                for (int j = 0; j < kRowBatch; ++j) {
                    va[j] = *(lds_f2*)(uintptr_t)(wl_abs + (tw[j] & 0xFFFFu));
                    if constexpr (PAIR) vb[j] = *(lds_f2*)(uintptr_t)(wl_abs + (tw[j] >> 16));
                }
#pragma unroll
                for (int j = 0; j < kRowBatch; ++j) {
                    if (itb + j < it_end && q[j] < n4) {
                        const uint32_t voff = ((uint32_t)lane + 64u * (itb + j)) * vbytes;
                        if constexpr (PAIR) step_store_obs(v4f{va[j].x, va[j].y, vb[j].x, vb[j].y}, base, voff);
                        else step_store_obs(va[j], base, voff);
                    }
                }
            }
            obs_s += obs_stride;
        }
        return;
    }

    // =========================================================================================================
    // SIM wave
    // =========================================================================================================
    CCX_ST(0);
    CCX_ST_CLK0();
    const int g = lane >> GLOG, i = lane & (G - 1);
    const int gbase = g << GLOG;
    const int env = env0 + g;
    const bool valid_env = (g < EW) && (env < E);
    const bool valid = valid_env && (i < N);
    const uint32_t validbit = in_vgpr(valid ? 1u : 0u);
    const uint32_t idx = ((uint32_t)env * (uint32_t)N + (uint32_t)i) & 0x0FFFFFFFu;   // (E x N < 2^28: ccx_create; keeps byte offsets 32-bit)
    const uint32_t idx_ld = valid ? idx : 0u;
    const uint32_t env_ld = valid_env ? (uint32_t)env : 0u;
    const bool boarding = i < Nb;
    const uint32_t tsh = boarding ? 8u : 12u, tsh2 = boarding ? 0u : 16u;

    // ---- every global load of the wave's entry, issued before anything is waited for ---------------------------
    const StateSlab sl = state_slab(E, N);
    const int s_x = reinterpret_cast<const int32_t*>(st_base + sl.x)[idx_ld];
    const int s_y = reinterpret_cast<const int32_t*>(st_base + sl.y)[idx_ld];
    const uint32_t s_active = (st_base + sl.active)[idx_ld], s_term = (st_base + sl.terminated)[idx_ld];
    const uint32_t s_trunc = (st_base + sl.truncated)[idx_ld];
    const int s_stepc = reinterpret_cast<const int32_t*>(st_base + sl.step_count)[env_ld];
    const int s_episode = reinterpret_cast<const int32_t*>(st_base + sl.episode)[env_ld];
    uint32_t araw[kActBatch];
    if constexpr (K1) {
        araw[0] = (uint32_t)actions[idx_ld];
#pragma unroll
        for (int d = 1; d < kActBatch; ++d) araw[d] = 4u;
    } else {
        uint32_t off = idx_ld;
        const uint32_t lim = idx_ld + (uint32_t)(K - 1) * EN;
#pragma unroll
        for (int d = 0; d < kActBatch; ++d) {
            araw[d] = (uint32_t)actions[off];
            const uint32_t nx = off + EN;
            off = nx < lim ? nx : lim;
        }
    }
    // move order: one byte per step and lane, packed 8 bits per step like the actions (16 steps = two 64-bit words)
    unsigned long long ord_lo = 0, ord_hi = 0;
    if constexpr (ORD) {
        uint32_t oraw[kActBatch];
        uint32_t off = idx_ld;
        const uint32_t lim = idx_ld + (uint32_t)(K - 1) * EN;
#pragma unroll
        for (int d = 0; d < (K1 ? 1 : kActBatch); ++d) {
            oraw[d] = (uint32_t)order[off];
            const uint32_t nx = off + EN;
            off = nx < lim ? nx : lim;
        }
#pragma unroll
        for (int d = 0; d < (K1 ? 1 : kActBatch); ++d) {
            if (d < 8) ord_lo |= (unsigned long long)(oraw[d] & 0xFFu) << (8 * d);
            else ord_hi |= (unsigned long long)(oraw[d] & 0xFFu) << (8 * (d - 8));
        }
    }
    // the first chunks of the cell table travel with the state (small grids need nothing more)
    constexpr int kCellFirst = 3;
    unsigned long long c_first[kCellFirst];
#pragma unroll
    for (int r = 0; r < kCellFirst; ++r) {
        const uint32_t cw = (uint32_t)lane + 64u * r;
        c_first[r] = cell_info[cw < cells ? cw : 0u];
    }
    const bool use_pool = auto_reset != 0 && pool != nullptr && pool_size > 0u;
    CCX_ST(1);
    // ---- in the shadow of that round trip: zero the occupancy tables, the row constants of both staging slots ---
    {
        uint4* occ = reinterpret_cast<uint4*>(smem + off_occ);
        for (uint32_t ow = (uint32_t)lane; ow < occ_bytes / 16u; ow += 64u) occ[ow] = make_uint4(0u, 0u, 0u, 0u);
    }
    WSlot* const wl = reinterpret_cast<WSlot*>(smem + off_ws);
    const bool want_obs = RW > 0;
    if (want_obs && lane < 16) {
        const int c = lane & 7;
        float v = -1.0f;
        if (c == 0) v = (float)dc;
        if (c == 1) v = (float)div;
        if (c == 2) v = (float)dl;
        if (c == 3) v = (float)dr;
        wl[lane >> 3].cst[c] = v;
    }
    // lane constants of the step: computed HERE, while the loads are in flight (pinned with empty asm: the compiler would
    // otherwise sink them behind the wait, next to their first use -- 280 ns between "tables in LDS" and "state in
    // registers" in the first stamps of this kernel, profiles/r04_step_k1_tstamps.txt)
    const mask_t full = full_mask<GLOG>();
    const mask_t lo_m = in_vgpr(low_mask<mask_t>(i));
    const mask_t later_m = in_vgpr((mask_t)(~lo_m & ~(mask_t(1) << i)));
    const mask_t mybit = in_vgpr((mask_t)(mask_t(1) << i));
    const uint32_t gsh = (uint32_t)lane & ~(uint32_t)(G - 1);
    const uint32_t cells1 = cells + 1u;
    const uint32_t g_tab = (g < EW) ? (uint32_t)g : 0u;
    const uint32_t tab_abs = lds0 + off_occ + g_tab * cells1 * 2u * msz;
    const uint32_t tab_rel = in_vgpr(tab_abs - (lds0 << TS));
    const uint32_t dump_addr = in_vgpr(tab_abs + ((valid ? cells : (uint32_t)(lane % Wp)) << (3u + TS)));
    const unsigned long long lut64 = (unsigned long long)(uint16_t)8 | ((unsigned long long)(uint16_t)(Wp * 8) << 16) |
                                     ((unsigned long long)(uint16_t)(-8) << 32) |
                                     ((unsigned long long)(uint16_t)(-Wp * 8) << 48);
    const uint32_t o_rew = in_vgpr(idx * 8u), o_af = idx, o_ef = (uint32_t)env & 0x0FFFFFFFu, o_cmp = in_vgpr(idx * 16u);
    const float type_f = boarding ? 0.0f : 1.0f;
    unsigned long long* const cinfo = reinterpret_cast<unsigned long long*>(smem);
#pragma unroll
    for (int r = 0; r < kCellFirst; ++r) {
        const uint32_t cw = (uint32_t)lane + 64u * r;
        if (cw < cells) cinfo[cw] = c_first[r];
    }
    // (larger grids: eight loads in flight per lane -- one at a time a 64 x 48 table was fifty memory round trips in a row)
    for (uint32_t cw0 = (uint32_t)lane + 64u * kCellFirst; cw0 < cells; cw0 += 64u * 8u) {
        unsigned long long c_more[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const uint32_t cw = cw0 + 64u * (uint32_t)r;
            c_more[r] = cell_info[cw < cells ? cw : 0u];
        }
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const uint32_t cw = cw0 + 64u * (uint32_t)r;
            if (cw < cells) cinfo[cw] = c_more[r];
        }
    }
    if (has_rtab) {
        double* rt = reinterpret_cast<double*>(smem + off_rtab);
        for (uint32_t t = (uint32_t)lane; t < 2u * cells; t += 64u) rt[t] = reward_table[t];
    }
    const uint32_t rt_add = off_rtab + (boarding ? 0u : cells * 8u);
    wave_lds_sync();
    CCX_ST(2);

    // ---- state -> registers (as in ccx_rollout_body.inc: position = LDS address of the agent's cell word) -------
    auto lds_or = [](uint32_t addr, mask_t bits) {
        __hip_atomic_fetch_or((__attribute__((address_space(3))) mask_t*)(uintptr_t)addr, bits, __ATOMIC_RELAXED,
                              __HIP_MEMORY_SCOPE_WAVEFRONT);
    };
    auto lds_mask_st = [](uint32_t addr, mask_t v) { *(__attribute__((address_space(3))) mask_t*)(uintptr_t)addr = v; };
    auto lds_cell = [](int addr) { return *(__attribute__((address_space(3))) const unsigned long long*)(uintptr_t)(uint32_t)addr; };
    auto group_raw = [&](uint64_t bb) -> mask_t { return GLOG == 6 ? (mask_t)bb : (mask_t)(bb >> gsh); };

    int c8 = (int)lds0 + (Wp + 1) * 8;
    uint32_t act = 0, tt = 1u;                       // lanes without an agent count as done
    int stepc = 0, episode = 0;
    if (valid) {
        c8 = (int)lds0 + ((s_y + 1) * Wp + s_x + 1) * 8;
        act = s_active != 0u;
        tt = (s_term != 0u ? 1u : 0u) | (s_trunc != 0u ? 2u : 0u);
    }
    if (valid_env) {
        stepc = s_stepc;
        episode = s_episode;
    }
    const int max_steps_m1 = max_steps - 1;
    int left1 = max_steps_m1 - stepc;                // negative from the step on which step_count reaches max_steps
    unsigned long long ci = lds_cell(c8);
    uint32_t ilo = (uint32_t)ci, ihi = (uint32_t)(ci >> 32);
    const uint64_t may_reset_b = __builtin_amdgcn_ballot_w64(use_pool && valid_env);
    const uint64_t slot0_b = __builtin_amdgcn_ballot_w64(valid_env && i == 0);
    const uint32_t reset_bit = use_pool ? (uint32_t)CCX_K_EF_RESET : 0u;

    // throughput counters (wave totals, scalar): live agent-steps, moves, episodes; arrivals as in the rollout kernel
    uint32_t c_moves = 0, c_live = 0, c_episodes = 0;
    uint32_t c_arrivals = (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(act != 0));

    // small outputs: scalar bases + 32-bit lane offsets of step 0, advanced by the streams' strides (a launch of <= 16
    // steps stays below 4 GiB per stream: 16 x 2^28 x 16 B would not, so the compact rows advance their BASE)
    const bool has_rew = reward != nullptr, has_af = agent_flags != nullptr, has_ef = env_flags != nullptr;
    const bool has_cmp = obs_compact != nullptr;
    typedef __attribute__((address_space(1))) char gchar;
    gchar* b_rew = (gchar*)reward;
    gchar* b_af = (gchar*)agent_flags;
    gchar* b_ef = (gchar*)env_flags;
    gchar* b_cmp = (gchar*)obs_compact;

    unsigned long long acur = 0;
    {
        uint32_t apk[2] = {0, 0};
#pragma unroll
        for (int d = 0; d < kActBatch; ++d) {
            const uint32_t a4 = araw[d] < 4u ? araw[d] : 4u;
            apk[d >> 3] |= a4 << (4 * (d & 7));
        }
        acur = (unsigned long long)apk[0] | ((unsigned long long)apk[1] << 32);
    }
    CCX_ST(3);

    for (int s = 0; s < K; ++s) {
        const uint32_t a = (uint32_t)acur & 0xFu;
        acur >>= 4;
        const uint32_t tt_before = tt;
        left1 -= 1;                                                    // collectivecrossing.py:188
        // ---- 1. proposal (:371-376, 509-534): legality is bit a of the current cell's word
        const int np8 = c8 + (int)(int16_t)(uint16_t)(lut64 >> ((a & 3u) << 4));     // (wait / absent: any neighbour, never entered -- its legality bit 4 is 0)
        const uint32_t ok = (ilo >> a) & act;
        const uint32_t nok = ok ^ 1u;
        // ---- move rank of this agent (dict order of action_dict, collectivecrossing.py:197); identity without ORD
        int rank = i;
        uint32_t src_lane = (uint32_t)lane;            // the lane whose agent has move rank i (= my lane index in the group)
        mask_t my_rbit = mybit, lo_r = lo_m, later_r = later_m;
        if constexpr (ORD) {
            const uint32_t o_cur = valid ? (uint32_t)(ord_lo & 0xFFu) : (uint32_t)i;    // slot of the agent that moves i-th
            ord_lo = (ord_lo >> 8) | (ord_hi << 56);
            ord_hi >>= 8;
            uint32_t* xch = reinterpret_cast<uint32_t*>(smem + off_xch);
            xch[gbase + (int)(o_cur & (uint32_t)(G - 1))] = (uint32_t)i;
            wave_lds_sync();
            rank = (int)xch[lane];
            wave_lds_sync();
            src_lane = (uint32_t)gbase + (o_cur & (uint32_t)(G - 1));
            my_rbit = mask_t(1) << rank;
            lo_r = low_mask<mask_t>(rank);
            later_r = ~lo_r & ~my_rbit;
        }
        // ---- 2. conflict masks from the occupancy / proposal tables (:536-541 in O(1) per agent), bits = move ranks
        const uint32_t ca = act ? tab_rel + ((uint32_t)c8 << TS) : dump_addr;
        const uint32_t ta = tab_rel + ((uint32_t)np8 << TS);
        const uint32_t qa = (ok ? ta : dump_addr) + msz;
        const unsigned long long pci = lds_cell(np8);
        lds_or(ca, my_rbit);
        lds_or(qa, my_rbit);
        wave_lds_sync();
        const mask2_t tt2 = *(__attribute__((address_space(3))) const mask2_t*)(uintptr_t)ta;
        wave_lds_sync();
        const bool live_lane = tt == 0u;
        const uint64_t live_b = __builtin_amdgcn_ballot_w64(live_lane);
        const uint32_t ge_m = (uint32_t)(left1 >> 31);                 // all ones from the step that reaches max_steps
        const uint32_t trunc2 = live_lane ? (ge_m & 2u) : 0u;
        uint32_t live_grp;
        if constexpr (GLOG == 6) live_grp = (uint32_t)live_b | (uint32_t)(live_b >> 32);
        else live_grp = (uint32_t)(live_b >> gsh) & (uint32_t)full;
        mask_t Cm = tt2.x & lo_r, Pm = tt2.y & lo_r;
        mask_t H = (tt2.x & later_r) | (mask_t)nok;
        if constexpr (ORD) {
            // the fixed point below runs with lane = move rank: lane i takes over the masks of the agent that moves i-th
            // (three cross-lane reads), and every agent finds its own verdict at bit `rank` of the final ballot
            auto pull = [&](mask_t v) -> mask_t {
                if constexpr (GLOG == 6) {
                    const uint32_t lo32 = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(src_lane << 2), (int)(uint32_t)v);
                    const uint32_t hi32 = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(src_lane << 2), (int)(uint32_t)(v >> 32));
                    return (mask_t)lo32 | ((mask_t)hi32 << 32);
                } else {
                    return (mask_t)(uint32_t)__builtin_amdgcn_ds_bpermute((int)(src_lane << 2), (int)(uint32_t)v);
                }
            };
            Cm = pull(Cm);
            Pm = pull(Pm);
            H = pull(H);
        }
        // ---- 3. ballot fixed point over "who moved" (ccx_rollout_body.inc, step 3)
        const uint64_t b0 = __builtin_amdgcn_ballot_w64((H | Cm) == 0);
        const mask_t M0 = group_raw(b0);
        uint64_t b_prev = __builtin_amdgcn_ballot_w64((H | (M0 & Pm) | (~M0 & Cm)) == 0);
        const mask_t M1 = group_raw(b_prev);
        bool mv_lane = (H | (M1 & Pm) | (~M1 & Cm)) == 0;
        uint64_t b = __builtin_amdgcn_ballot_w64(mv_lane);
        lds_mask_st(ca, 0);
        lds_mask_st(qa, 0);
        if (b != b_prev) {   // a chain of three or more agents somewhere in the wave: iterate
            for (int it = 3; it <= N; ++it) {
                const mask_t M = group_raw(b);
                mv_lane = (H | (M & Pm) | (~M & Cm)) == 0;
                const uint64_t b2 = __builtin_amdgcn_ballot_w64(mv_lane);
                if (b2 == b) break;
                b = b2;
            }
        }
        if constexpr (ORD) mv_lane = ((group_raw(b) >> rank) & mask_t(1)) != 0;    // (the ballot is rank-indexed)
        if (mv_lane) {  // :408
            c8 = np8;
            ilo = (uint32_t)pci;
            ihi = (uint32_t)(pci >> 32);
        }
        c_moves += (uint32_t)__builtin_popcountll(b);
        c_live += (uint32_t)__builtin_popcountll(live_b);
        // ---- 4. deactivate on arrival (:210-212), terminated (terminateds.py:40-82), truncated, __all__ (:256-259)
        const uint32_t dest = (ilo >> tsh) & 1u;
        act &= ~dest;
        // ---- 5. hand the agents' (x, y, type, active) to the row waves: ONE barrier per step, two slots
        const float4 me = make_float4((float)((ilo >> 16) & 0xFFu), (float)(ilo >> 24), type_f, (float)act);
        if (want_obs) {
            wl[s & 1].slot[lane] = me;
            lds_barrier();
        }
        const uint32_t tind = (ilo >> (tsh + kCellTermShift)) & 1u;            // terminateds[id] as the cell says (ccx_kernels.h)
        const uint64_t ndest_b = __builtin_amdgcn_ballot_w64((validbit & ~tind) != 0);
        uint32_t ndest_grp;
        if constexpr (GLOG == 6) ndest_grp = (uint32_t)ndest_b | (uint32_t)(ndest_b >> 32);
        else ndest_grp = (uint32_t)(ndest_b >> gsh) & (uint32_t)full;
        const uint32_t all_dest = ndest_grp == 0u ? 1u : 0u;
        const uint32_t term_out = term_all ? all_dest : tind;
        const uint32_t out2 = term_out | trunc2;
        tt |= out2;
        const uint32_t ef = all_dest | ((live_grp != 0u ? 2u : 0u) & ge_m);
        const uint64_t reset_b = __builtin_amdgcn_ballot_w64(ef != 0u) & may_reset_b;
        const uint32_t efw = ef + (ef < 1u ? ef : 1u) * reset_bit;
        // ---- 6. this step's small outputs (collectivecrossing.py:214-261)
        const uint32_t live = tt_before == 0u ? 1u : 0u;                       // rewards.py:64, truncateds.py:56
        const uint32_t emit = (live | (out2 & ~tt_before)) != 0u ? 1u : 0u;    // :243, :763-767
        const uint32_t af = out2 | (live << 2) | (emit << 3) | ((ilo >> 1) & 0x30u) | (act << 6) | (dest << 7);
        if (valid) {
            if (has_rew) {
                // rewards.py:44-182: the INTEGER is negated before the one f64 multiply (d == 0 gives +0.0)
                double r;
                if (has_rtab) {          // position-only user reward: one f64 per (type, cell)
                    r = *(__attribute__((address_space(3))) const double*)(uintptr_t)((uint32_t)c8 + rt_add);
                } else {
                    const uint32_t cls = (ilo >> (tsh + 1u)) & 3u;
                    const int sd = (int)(int16_t)(uint16_t)(ihi >> tsh2);
                    r = (double)sd * rF;
                    r = (cls == 1u) ? rA : r;
                    r = (cls == 2u) ? rB : r;
                    r = (cls == 3u) ? rC : r;
                }
                r = live ? r : 0.0;
                *(__attribute__((address_space(1))) double*)(b_rew + o_rew) = r;
            }
            if (has_af) *(__attribute__((address_space(1))) uint8_t*)(b_af + o_af) = (uint8_t)af;
            if (has_ef && i == 0) *(__attribute__((address_space(1))) uint8_t*)(b_ef + o_ef) = (uint8_t)efw;
            if (has_cmp) *(__attribute__((address_space(1))) v4f*)(b_cmp + o_cmp) = v4f{me.x, me.y, me.z, me.w};
        }
        b_rew += (size_t)EN * 8u;
        b_af += EN;
        b_ef += (uint32_t)E;
        b_cmp += (size_t)EN * 16u;
        // ---- 7. auto-reset from the pool (reset() :97-150 with precomputed placements); the cursor is only worked out
        //         when an env really restarts (rare inside <= 16 steps): entry (global_env + episode * stride) mod P
        if (reset_b != 0) {
            const bool do_reset = ((reset_b >> lane) & 1ull) != 0;
            c_episodes += (uint32_t)__builtin_popcountll(reset_b & slot0_b);
            c_arrivals += (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(do_reset && valid && act == 0u));
            if (do_reset) {
                episode += 1;
                left1 = max_steps_m1;
                if (valid) {
                    const unsigned long long P = (unsigned long long)pool_size;
                    const unsigned long long gi = ((unsigned long long)env_offset_mod_pool + (unsigned long long)env) % P;
                    const unsigned long long ep = (unsigned long long)(uint32_t)episode % P;
                    const uint32_t pool_idx = (uint32_t)((gi + ep * (unsigned long long)pool_stride) % P);
                    const uint32_t pn = *reinterpret_cast<const uint16_t*>(pool + ((size_t)pool_idx * N + i) * 2);
                    c8 = (int)lds0 + ((int)(pn >> 8) * Wp + (int)(pn & 0xFFu) + Wp + 1) * 8;
                    const unsigned long long rci = lds_cell(c8);
                    ilo = (uint32_t)rci;
                    ihi = (uint32_t)(rci >> 32);
                    act = 1;
                    tt = 0;
                }
            }
        }
    }
    CCX_ST(4);
    // ---- registers -> state --------------------------------------------------------------------------------------
    if (valid) {
        reinterpret_cast<int32_t*>(st_base + sl.x)[idx] = (int)((ilo >> 16) & 0xFFu);
        reinterpret_cast<int32_t*>(st_base + sl.y)[idx] = (int)(ilo >> 24);
        (st_base + sl.active)[idx] = (uint8_t)act;
        (st_base + sl.terminated)[idx] = (uint8_t)(tt & 1u);
        (st_base + sl.truncated)[idx] = (uint8_t)(tt >> 1);
    }
    if (valid_env && i == 0) {
        reinterpret_cast<int32_t*>(st_base + sl.step_count)[env] = max_steps_m1 - left1;
        reinterpret_cast<int32_t*>(st_base + sl.episode)[env] = episode;
    }
    CCX_ST(5);
    CCX_ST_CLK1();
    c_arrivals -= (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(act != 0));
    if (counters) {
        // one partial slot per tile (ccx_kernels.h: kCounterSlot), six lanes -> six words, one instruction
        const uint32_t nenv = (uint32_t)__builtin_popcountll(slot0_b);
        unsigned long long v = 0;
        if (lane == 0) v = (unsigned long long)nenv * (unsigned long long)K;
        if (lane == 1) v = (unsigned long long)nenv * (unsigned long long)K * (unsigned long long)N;
        if (lane == 2) v = c_live;
        if (lane == 3) v = c_episodes;
        if (lane == 4) v = c_moves;
        if (lane == 5) v = c_arrivals;
        if (lane < 6 && nenv)
            atomicAdd(counters + kCounterTotals + (size_t)tile * kCounterSlot + lane, v);
    }
    CCX_ST_DRAIN(6);
}

// ---------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------
size_t step_lds_bytes(int glog, int ew, int N, int cells, bool reward_table) {
    auto up16 = [](size_t v) { return (v + 15u) & ~(size_t)15u; };
    const size_t msz = glog == 6 ? 8u : 4u;
    (void)N;
    return up16((size_t)cells * 8u) + (reward_table ? (size_t)cells * 16u : 0u) +
           up16((size_t)ew * 2u * ((size_t)cells + 1u) * msz) + 2u * sizeof(WSlot) + 256u /* move-order exchange */;
}

template <int GLOG>
static hipError_t launch_step_g(const StepShape& ss, hipStream_t stream, const KParams& p, uint8_t* st_base,
                                const unsigned long long* cell_info, const uint8_t* actions, const uint8_t* order, int K,
                                int auto_reset, const uint8_t* pool, const KOut& out, unsigned long long* counters) {
    const bool pair = (p.N % 2) == 0;
    auto pick = [&](auto ord_c) -> const void* {
        constexpr bool ORD = decltype(ord_c)::value;
        return K == 1 ? (pair ? reinterpret_cast<const void*>(&step_kernel<GLOG, true, true, ORD>)
                              : reinterpret_cast<const void*>(&step_kernel<GLOG, false, true, ORD>))
                      : (pair ? reinterpret_cast<const void*>(&step_kernel<GLOG, true, false, ORD>)
                              : reinterpret_cast<const void*>(&step_kernel<GLOG, false, false, ORD>));
    };
    const void* entry = order ? pick(std::true_type{}) : pick(std::false_type{});
    int E = p.E;
    const int row_waves = out.obs ? ss.row_waves : 0;
    uint32_t shape = (uint32_t)p.N | ((uint32_t)p.Nb << 8) | ((uint32_t)ss.envs_per_wave << 16) | ((uint32_t)K << 24);
    const uint32_t cells = (uint32_t)((p.W + 3) * (p.H + 3));
    const double* reward_table = p.off_rtab ? p.reward_table : nullptr;
    uint32_t grid_w = cells | ((uint32_t)(p.W + 3) << 16) | ((uint32_t)row_waves << 24) | (reward_table ? 0x80000000u : 0u);
    int max_steps = p.max_steps;
    const uint16_t* obs_table = p.obs_table;
    float* obs = out.obs;
    double* reward = out.reward;
    uint8_t *af = out.agent_flags, *ef = out.env_flags;
    float* cmp = out.obs_compact;
    uint32_t pool_size = (uint32_t)p.pool_size, pool_stride = (uint32_t)p.pool_stride;
    uint32_t env_offset_mod_pool = p.pool_size > 0 ? (uint32_t)(p.env_offset % p.pool_size) : 0u;
    int dc = p.dc, div = p.div, dl = p.dl, dr = p.dr, term_all = p.term_mode == CCX_K_TERM_ALL ? 1 : 0;
    double rA = p.reward_mode == CCX_K_REWARD_BINARY ? p.r_nogoal
                : p.reward_mode == CCX_K_REWARD_CONSTANT_NEGATIVE ? p.r_pen : p.r_dest;
    double rB = p.r_door, rC = p.r_area, rF = p.r_f;
    void* args[] = {&st_base, &actions, &cell_info, &obs_table, &obs, &E, &shape, &grid_w, &max_steps,
                    &reward, &af, &ef, &cmp, &counters, &pool, &pool_size, &pool_stride, &env_offset_mod_pool,
                    &dc, &div, &dl, &dr, &term_all, &auto_reset, &rA, &rB, &rC, &rF, &reward_table, &order};
    if (ss.lds_bytes > 60 * 1024) {
        hipError_t e = hipFuncSetAttribute(entry, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
    }
    return hipLaunchKernel(entry, dim3((unsigned)ss.num_blocks), dim3(64u * (unsigned)(1 + row_waves)), args, ss.lds_bytes, stream);
}

hipError_t launch_step(const StepShape& ss, hipStream_t stream, const KParams& p, uint8_t* st_base,
                       const unsigned long long* cell_info, const uint8_t* actions, const uint8_t* order, int K, int auto_reset,
                       const uint8_t* pool, const KOut& out, unsigned long long* counters) {
    switch (ss.glog) {
    case 0: return launch_step_g<0>(ss, stream, p, st_base, cell_info, actions, order, K, auto_reset, pool, out, counters);
    case 1: return launch_step_g<1>(ss, stream, p, st_base, cell_info, actions, order, K, auto_reset, pool, out, counters);
    case 2: return launch_step_g<2>(ss, stream, p, st_base, cell_info, actions, order, K, auto_reset, pool, out, counters);
    case 3: return launch_step_g<3>(ss, stream, p, st_base, cell_info, actions, order, K, auto_reset, pool, out, counters);
    case 4: return launch_step_g<4>(ss, stream, p, st_base, cell_info, actions, order, K, auto_reset, pool, out, counters);
    case 5: return launch_step_g<5>(ss, stream, p, st_base, cell_info, actions, order, K, auto_reset, pool, out, counters);
    case 6: return launch_step_g<6>(ss, stream, p, st_base, cell_info, actions, order, K, auto_reset, pool, out, counters);
    }
    return hipErrorInvalidValue;
}

}  // namespace ccx
