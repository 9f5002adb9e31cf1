// ccx_kernels.hip -- gfx950 (MI355X, CDNA4) kernels of libccx.
//
// Replaces, for E independent envs at once, the reference's CollectiveCrossingEnv.step
// (src/collectivecrossing/collectivecrossing.py:161-261) and the strategy methods it calls
// (rewards.py:44-182, terminateds.py:40-82, truncateds.py:40-61, observations.py:43-94).
//
// Mapping (wave64): one LANE per (env, agent).  An env owns a group of G = 2^GLOG >= N consecutive
// lanes, a wavefront carries EW <= 64/G envs and keeps their state in registers for all K steps of
// a rollout; waves never talk to each other (no barrier in the step loop, no inter-workgroup
// traffic), so the grid is embarrassingly parallel over env tiles.
//
// Ordered move resolution without a serial agent loop ("agent k sees earlier agents at their new
// cell and later agents at their old cell", collectivecrossing.py:197-202,536-541):
//   1. every lane proposes its target cell in parallel (bounds + wall predicate, closed form);
//   2. proposals are exchanged through a 256-byte LDS tile indexed by move RANK, so lane k of a
//      group plays "the agent moved k-th";
//   3. each rank k builds two bit masks over earlier ranks k' < k:  P[k'] = prop_k' == prop_k,
//      C[k'] = active_k' && cur_k' == prop_k, and tests later ranks' current cells once;
//   4. with M = mask of earlier ranks that DID move, rank k moves iff ((M & P) | (~M & C)) == 0,
//      one v_bfi_b32.  M is a wave ballot (v_cmp writes the lane mask to SGPRs for free); F(M)
//      has a unique fixed point reached in <= N rounds (bit k of F depends on bits < k only), and
//      the loop exits as soon as two consecutive ballots agree -- typically after 2 rounds.
//
// Observation gather (the byte-dominant part, 4*(6+4N) B per agent-step): lanes stage
// (x, y, type, active) as float4 per agent in LDS; the wave's rows form ONE contiguous region of
// the [E][N][L] output, written with full-width global_store_dwordx4 whose two 8-byte halves come
// from LDS addresses pre-computed once per workgroup (a u16 table in LDS), so the per-step cost of
// a 1 KiB store is one ds_read_b32 + two ds_read_b64.
//
// Everything is integer / index work; the only floating-point operation on the path is ONE f64
// multiply per reward (compiled with -ffp-contract=off).  No MFMA on purpose.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ccx_kernels.h"

namespace ccx {

// ---------------------------------------------------------------------------------------------
// wave-level helpers
// ---------------------------------------------------------------------------------------------

// LDS traffic of ONE wave is executed in program order by the hardware; this keeps the compiler
// from moving LDS accesses across the hand-off point (no instruction is emitted for the barrier).
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <int GLOG> struct GroupMask { using type = uint32_t; };
template <> struct GroupMask<6> { using type = uint64_t; };

// bits of a 64-lane ballot that belong to this lane's group, shifted down to bit 0
template <int GLOG>
__device__ __forceinline__ typename GroupMask<GLOG>::type group_bits(uint64_t ballot, int lane) {
    if constexpr (GLOG == 6) {
        return ballot;
    } else if constexpr (GLOG == 5) {
        return (lane & 32) ? (uint32_t)(ballot >> 32) : (uint32_t)ballot;
    } else {
        constexpr int G = 1 << GLOG;
        uint32_t half = (lane & 32) ? (uint32_t)(ballot >> 32) : (uint32_t)ballot;
        return (half >> (lane & 31 & ~(G - 1))) & ((1u << G) - 1u);
    }
}

template <typename T> __device__ __forceinline__ T low_mask(int i) {
    return (T(1) << i) - T(1);
}

template <int GLOG> __device__ __forceinline__ constexpr typename GroupMask<GLOG>::type full_mask() {
    using T = typename GroupMask<GLOG>::type;
    if constexpr (GLOG >= 5) return ~T(0);
    else return (T(1) << (1 << GLOG)) - T(1);
}

typedef float v4f __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// per-wave LDS tile
struct WaveLds {
    float4 slot[64];   // (x, y, type, active) of the agent on each lane, as floats
    float cst[8];      // (door_centre, division_y) (door_left, door_right) (-1,-1) pad
    uint32_t xch[64];  // move proposals, indexed [group_base + rank]: cur_key | prop_key << 16
};
static_assert(sizeof(WaveLds) == 1024 + 32 + 256, "WaveLds layout");
static constexpr uint32_t kCstOff = 1024;  // byte offset of cst[] from slot[]

// u16 table entry for float2 unit `w` of a wave's observation region: byte offset (from the wave's
// WaveLds) of the 8 bytes to copy there.  Row layout (observations.py:64-92):
//   unit 0 = (x_i, y_i); unit 1 = (door_centre, division_y); unit 2 = (door_left, door_right);
//   unit 3+2j = (x_j, y_j), unit 4+2j = (type_j, active_j), or (-1,-1) for j == i.
template <int GLOG>
__device__ __forceinline__ uint16_t obs_unit_addr(uint32_t w, int N, int U) {
    uint32_t row = w / (uint32_t)U, u = w - row * (uint32_t)U;
    uint32_t el = row / (uint32_t)N, i = row - el * (uint32_t)N;
    uint32_t gb = el << GLOG;
    if (u == 0) return (uint16_t)((gb + i) * 16u);
    if (u < 3) return (uint16_t)(kCstOff + (u - 1u) * 8u);
    uint32_t j = (u - 3u) >> 1, h = (u - 3u) & 1u;
    if (j == i) return (uint16_t)(kCstOff + 16u);
    return (uint16_t)((gb + j) * 16u + h * 8u);
}

template <int GLOG>
__device__ __forceinline__ void build_obs_table(uint16_t* table, const KParams& p) {
    const int U = 3 + 2 * p.N;
    for (uint32_t w = threadIdx.x; w < (uint32_t)p.units_per_wave; w += blockDim.x)
        table[w] = obs_unit_addr<GLOG>(w, p.N, U);
    if (threadIdx.x == 0 && (p.units_per_wave & 1)) table[p.units_per_wave] = 0;
}

__device__ __forceinline__ void init_wave_consts(WaveLds* wl, const KParams& p, int lane) {
    if (lane < 8) {
        float v = -1.0f;
        if (lane == 0) v = (float)p.dc;
        if (lane == 1) v = (float)p.div;
        if (lane == 2) v = (float)p.dl;
        if (lane == 3) v = (float)p.dr;
        wl->cst[lane] = v;
    }
}

// copy the wave's observation region out of LDS: `units` float2 units starting at `dst`
template <bool PAIR>
__device__ __forceinline__ void emit_obs(const WaveLds* wl, const uint16_t* table, float* dst,
                                         int units, int lane) {
    const char* sbase = reinterpret_cast<const char*>(wl);
    if constexpr (PAIR) {
        // N even: region start and length are multiples of 16 bytes
        const uint32_t* t32 = reinterpret_cast<const uint32_t*>(table);
        v4f* d4 = reinterpret_cast<v4f*>(dst);
        const int n4 = units >> 1;
        for (int q = lane; q < n4; q += 64) {
            uint32_t t = t32[q];
            float2 a = *reinterpret_cast<const float2*>(sbase + (t & 0xFFFFu));
            float2 b = *reinterpret_cast<const float2*>(sbase + (t >> 16));
            v4f v = {a.x, a.y, b.x, b.y};
            __builtin_nontemporal_store(v, &d4[q]);
        }
    } else {
        float2* d2 = reinterpret_cast<float2*>(dst);
        for (int w = lane; w < units; w += 64) {
            float2 a = *reinterpret_cast<const float2*>(sbase + table[w]);
            d2[w] = a;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// the fused rollout / step kernel
// ---------------------------------------------------------------------------------------------

// Keep a wave-uniform value in VGPRs on purpose: the step loop needs ~60 scalars (geometry,
// pointers, reward constants) next to the ballot masks, which overflows the 102 SGPRs and makes
// hipcc spill SGPRs through v_writelane/v_readlane inside the loop.  VGPRs are plentiful here
// (one or two waves per SIMD), so the loop-invariant values that are only used by vector
// instructions anyway are pinned there.
template <typename T> __device__ __forceinline__ T in_vgpr(T v) {
    if constexpr (sizeof(T) == 8) {
        unsigned long long u = __builtin_bit_cast(unsigned long long, v);
        asm volatile("" : "+v"(u));
        return __builtin_bit_cast(T, u);
    } else {
        asm volatile("" : "+v"(v));
        return v;
    }
}

// Diagnostic build only (-DCCX_STAMPS): s_memtime stamps around the segments of one step; wave 0
// adds its per-segment cycle sums to counters[8..15].  Never compiled into libccx.so.
#ifdef CCX_STAMPS
#define CCX_STAMP(slot)                                                                     \
    do {                                                                                    \
        unsigned long long t_;                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                  \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");          \
        __builtin_amdgcn_sched_barrier(0);                                                  \
        stamp_sum[slot] += t_ - t_prev;                                                     \
        t_prev = t_;                                                                        \
    } while (0)
#else
#define CCX_STAMP(slot) do { } while (0)
#endif

constexpr int kActBatch = 16;  // env-steps of actions fetched per global-load burst

template <int GLOG, bool PAIR>
__global__ void __launch_bounds__(256)
rollout_kernel(const KParams p, const KState st, const uint8_t* __restrict__ actions,
               const uint8_t* __restrict__ order, const int K, const int auto_reset,
               const uint8_t* __restrict__ pool, const KOut out, unsigned long long* counters) {
    using mask_t = typename GroupMask<GLOG>::type;
    constexpr int G = 1 << GLOG;
    extern __shared__ __align__(16) unsigned char smem[];

    const int lane = threadIdx.x & 63;
    const int wib = threadIdx.x >> 6;
    const int wave = blockIdx.x * p.waves_per_block + wib;
    const int g = lane >> GLOG;
    const int i = lane & (G - 1);
    const int gbase = g << GLOG;
    const int env0 = wave * p.EW;                 // first env of this wave
    const int env = env0 + g;
    const bool valid_env = (g < p.EW) && (env < p.E);
    const bool valid = valid_env && (i < p.N);
    const int N = p.N;
    const int L = 6 + 4 * N;
    const size_t EN = (size_t)p.E * N;
    const size_t idx = (size_t)env * N + i;
    const bool boarding = i < p.Nb;
    const int dest_y = boarding ? p.bdy : p.edy;
    const mask_t full = full_mask<GLOG>();
    const mask_t lo = low_mask<mask_t>(i);
    const mask_t later = ~lo & ~(mask_t(1) << i);

    // LDS carve-up: [WaveLds x waves_per_block][u16 table]
    WaveLds* wl = reinterpret_cast<WaveLds*>(smem) + wib;
    uint16_t* table = reinterpret_cast<uint16_t*>(smem + sizeof(WaveLds) * p.waves_per_block);
    const bool want_obs = out.obs != nullptr;
    if (want_obs) build_obs_table<GLOG>(table, p);
    init_wave_consts(wl, p, lane);
    __syncthreads();  // the only workgroup barrier: table is read-only from here on

    int envs_here = p.E - env0;
    envs_here = envs_here < 0 ? 0 : (envs_here > p.EW ? p.EW : envs_here);
    const int units = envs_here * N * (3 + 2 * N);

    // ---- state -> registers ------------------------------------------------------------------
    int x = 0, y = 0, stepc = 0, episode = 0;
    bool active = false, term = false, trunc = false;
    if (valid) {
        x = st.x[idx];
        y = st.y[idx];
        active = st.active[idx] != 0;
        term = st.terminated[idx] != 0;
        trunc = st.truncated[idx] != 0;
    }
    if (valid_env) {
        stepc = st.step_count[env];
        episode = st.episode[env];
    }
    // final write-back addresses, parked in VGPRs for the duration of the loop
    int32_t* const fx = in_vgpr(st.x + idx);
    int32_t* const fy = in_vgpr(st.y + idx);
    uint8_t* const fact = in_vgpr(st.active + idx);
    uint8_t* const fterm = in_vgpr(st.terminated + idx);
    uint8_t* const ftrunc = in_vgpr(st.truncated + idx);
    int32_t* const fstep = in_vgpr(st.step_count + env);
    int32_t* const fepi = in_vgpr(st.episode + env);
    unsigned long long* const ctr = in_vgpr(counters);

    // reward constants, selected once (rewards.py:44-182):  r = c1 ? rA : c2 ? rB : c3 ? rC : d*rF
    const int rmode = p.reward_mode;
    const double rA = in_vgpr(rmode == CCX_K_REWARD_BINARY ? p.r_nogoal
                              : rmode == CCX_K_REWARD_CONSTANT_NEGATIVE ? p.r_pen : p.r_dest);
    const double rB = in_vgpr(p.r_door);
    const double rC = in_vgpr(p.r_area);
    const double rF = in_vgpr(p.r_f);

    // reset-pool cursor of this env: entry (global_env + episode*total) mod P, advanced by
    // total mod P per episode; the NEXT placement is prefetched right after every reset and only
    // decoded when it is consumed (so no wait sits behind the load).
    uint32_t pool_idx = 0, pnext = 0;
    int px = 0, py = 0;
    bool pnext_pending = false;   // pnext holds a load that has not been decoded into px/py yet
    const bool use_pool = auto_reset && pool != nullptr && p.pool_size > 0;
    const uint32_t pool_size = (uint32_t)p.pool_size, pool_stride = (uint32_t)p.pool_stride;
    const uint8_t* const pool_v = pool;   // stays a global-address-space pointer (SGPR pair)
    if (use_pool && valid) {
        unsigned long long P = (unsigned long long)p.pool_size;
        unsigned long long gi = (unsigned long long)(p.env_offset + env) % P;
        unsigned long long ep = (unsigned long long)(episode + 1) % P;
        pool_idx = (uint32_t)((gi + ep * (unsigned long long)p.pool_stride) % P);
        pnext = *reinterpret_cast<const uint16_t*>(pool_v + ((size_t)pool_idx * N + i) * 2);
        pnext_pending = true;
    }

    // per-lane output cursors, advanced by one step's stride per iteration
    double* rew_p = out.reward ? out.reward + idx : nullptr;
    uint8_t* af_p = out.agent_flags ? out.agent_flags + idx : nullptr;
    uint8_t* ef_p = out.env_flags ? out.env_flags + env : nullptr;
    float* obs_p = want_obs ? out.obs + (size_t)env0 * N * L : nullptr;
    const size_t obs_stride = EN * (size_t)L;
    const uint8_t* act_p = actions + idx;
    const uint8_t* ord_p = order ? order + idx : nullptr;

    uint32_t c_moves = 0, c_arrivals = 0, c_live = 0, c_episodes = 0;
#ifdef CCX_STAMPS
    unsigned long long stamp_sum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t_prev = 0;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_prev)::"memory");
#endif

    // ---- actions: bursts of kActBatch steps, 4 bits per step in a u64 --------------------------
    // One s_waitcnt vmcnt per burst instead of one per step (vmcnt also counts the stores).
    // Loads are unconditional (steps past K re-read the last valid step) and addressed from a
    // VGPR stride, so the burst is straight-line code with no scalar address table.
    uint32_t araw[kActBatch];
    const size_t EN_v = in_vgpr(EN);
    auto fetch_actions = [&](int s_first) {
        const int last = K - 1 - s_first;   // >= 0 whenever this is called
#pragma unroll
        for (int d = 0; d < kActBatch; ++d) {
            const int dd = d < last ? d : last;
            araw[d] = valid ? (uint32_t)act_p[(size_t)dd * EN_v] : CCX_K_ABSENT;
        }
        act_p += (size_t)kActBatch * EN_v;
    };
    fetch_actions(0);

    for (int s0 = 0; s0 < K; s0 += kActBatch) {
        // the burst issued one batch ago is consumed here: ONE vmcnt wait per kActBatch steps
        unsigned long long apack = 0;
#pragma unroll
        for (int d = 0; d < kActBatch; ++d)
            apack |= (unsigned long long)(araw[d] & 0xFu) << (4 * d);   // 255 -> 15: no move
        apack = in_vgpr(apack);
        if (pnext_pending) {   // decode behind the wait that just happened: costs nothing
            px = (int)(pnext & 0xFFu);
            py = (int)(pnext >> 8);
            pnext_pending = false;
        }
        __builtin_amdgcn_sched_barrier(0);   // keep the next burst BEHIND the wait above
        if (s0 + kActBatch < K) fetch_actions(s0 + kActBatch);
        __builtin_amdgcn_sched_barrier(0);
        const int dmax = (K - s0) < kActBatch ? (K - s0) : kActBatch;
        CCX_STAMP(0);   // burst wait + pack + next burst issue

        for (int d = 0; d < dmax; ++d) {
            const uint32_t a = (uint32_t)(apack >> (4 * d)) & 0xFu;

            // ---- move rank of this agent (dict order of action_dict, collectivecrossing.py:197)
            int rank = i;
            if (ord_p != nullptr) {
                uint32_t ok_ = valid ? (uint32_t)*ord_p : (uint32_t)i;  // agent moved i-th
                ord_p += EN;
                wl->xch[gbase + (ok_ & (G - 1))] = (uint32_t)i;
                wave_lds_sync();
                rank = (int)wl->xch[lane];
                wave_lds_sync();
            }

            stepc += 1;  // collectivecrossing.py:188
            CCX_STAMP(7);   // loop overhead / order path

            // ---- 1. proposal (collectivecrossing.py:371-376, 509-534; 565-588 adds nothing)
            const int dx = (a == 0u) - (a == 2u);
            const int dy = (a == 1u) - (a == 3u);
            const int nx = x + dx, ny = y + dy;
            bool ok = valid && active && (a < 4u);
            ok = ok && ((unsigned)nx <= (unsigned)p.W) && ((unsigned)ny <= (unsigned)p.H);
            ok = ok && !(ny == p.div && !(p.dl < nx && nx < p.dr));
            ok = ok && !(ny >= p.div && !(p.tl < nx && nx < p.tr));
            const uint32_t curkey = (valid && active) ? (uint32_t)(x | (y << 8)) : 0x8000u;
            const uint32_t propkey = ok ? (uint32_t)(nx | (ny << 8)) : 0xFFFFu;
            wl->xch[gbase + rank] = curkey | (propkey << 16);
            wave_lds_sync();

            // ---- 2. this lane now plays move-rank i of its group.  Entries of unused lanes hold
            //         cur 0x8000 / prop 0xFFFF and can never match a real proposal.
            const uint32_t myprop = wl->xch[lane] >> 16;
            mask_t call = 0, pall = 0;
#pragma unroll
            for (int k2 = 0; k2 < G; ++k2) {
                const uint32_t v = wl->xch[gbase + k2];
                call |= (mask_t)((v & 0xFFFFu) == myprop) << k2;
                pall |= (mask_t)((v >> 16) == myprop) << k2;
            }
            CCX_STAMP(1);   // proposal + LDS exchange + pair masks
            const mask_t Cm = call & lo, Pm = pall & lo;
            const bool okr = (myprop != 0xFFFFu) && ((call & later) == 0);  // later ranks: old cells

            // ---- 3. ballot fixed point over "who moved"
            uint64_t b = __ballot(okr && (Cm == 0));
            for (int it = 1; it < N; ++it) {
                const mask_t M = group_bits<GLOG>(b, lane);
                const uint64_t b2 = __ballot(okr && (((M & Pm) | (~M & Cm)) == 0));
                if (b2 == b) break;
                b = b2;
            }
            wave_lds_sync();  // xch is rewritten next step
            const bool moved = (group_bits<GLOG>(b, lane) >> rank) & 1;
            if (moved) {  // collectivecrossing.py:408
                x = nx;
                y = ny;
            }
            c_moves += moved;
            CCX_STAMP(2);   // ballot fixed point + position update

            // ---- 4. tail: deactivate, reward, terminated, truncated, flags (:210-241)
            const bool at_dest = (y == dest_y);                       // :663-683
            const bool arrive = valid && active && at_dest;           // :210-212
            active = active && !arrive;
            c_arrivals += arrive;
            const bool live = valid && !(term || trunc);              // rewards.py:64, truncateds.py:56
            c_live += live;
            const bool in_area = (y >= p.div) && (p.tl <= x) && (x <= p.tr);         // :551-554
            const bool at_door = (y == p.div) && (x == p.dl - 1 || x == p.dr + 1);   // :556-563

            // rewards.py:44-182.  Distances are integers and the reference negates the INTEGER
            // before the one f64 multiply, so d == 0 gives +0.0 (never -0.0).
            bool c1 = true, c2 = false, c3 = false;
            int sd = 0;
            if (rmode == CCX_K_REWARD_DEFAULT) {
                const int adx = x > p.dc ? x - p.dc : p.dc - x;
                c1 = at_dest;
                c2 = boarding && at_door;
                c3 = boarding ? in_area : !in_area;
                sd = boarding ? -(adx + (p.div - y)) : (adx + (y - p.div));
            } else if (rmode == CCX_K_REWARD_SIMPLE_DISTANCE) {
                c1 = false;
                sd = -(y > dest_y ? y - dest_y : dest_y - y);
            }
            double r = c1 ? rA : c2 ? rB : c3 ? rC : (double)sd * rF;
            if (!live) r = 0.0;

            bool term_out = at_dest;                                   // terminateds.py:66-82
            const mask_t dest_bits = group_bits<GLOG>(__ballot(at_dest || !valid), lane);
            if (p.term_mode == CCX_K_TERM_ALL) term_out = (dest_bits == full);   // terminateds.py:40-60
            const bool trunc_out = live && (stepc >= p.max_steps);     // truncateds.py:40-61
            const bool done_now = (term_out && !term) || (trunc_out && !trunc);  // :229-241
            term = term || term_out;
            trunc = trunc || trunc_out;
            const bool emit = done_now || !(term || trunc);            // :243, :763-767

            const mask_t term_bits = (p.term_mode == CCX_K_TERM_ALL)
                                         ? (term_out ? full : mask_t(0))
                                         : dest_bits;
            const bool all_term = (term_bits == full);                 // :256
            const mask_t live_bits = group_bits<GLOG>(__ballot(live), lane);
            const mask_t tr_bits = group_bits<GLOG>(__ballot(!live || trunc_out), lane);
            const bool all_trunc = (live_bits != 0) && (tr_bits == full);  // :257
            uint32_t ef = (all_term ? CCX_K_EF_ALL_TERM : 0u) | (all_trunc ? CCX_K_EF_ALL_TRUNC : 0u);

            const uint32_t af = (term_out ? 0x01u : 0u) | (trunc_out ? 0x02u : 0u) |
                                (live ? 0x04u : 0u) | (emit ? 0x08u : 0u) | (in_area ? 0x10u : 0u) |
                                (at_door ? 0x20u : 0u) | (active ? 0x40u : 0u) | (at_dest ? 0x80u : 0u);

            CCX_STAMP(3);   // tail: reward / flags / ballots
            // ---- 5. outputs
            if (valid) {
                if (rew_p) {
                    *rew_p = r;
                    rew_p += EN;
                }
                if (af_p) {
                    *af_p = (uint8_t)af;
                    af_p += EN;
                }
            }
            CCX_STAMP(4);   // reward + flag stores
            if (want_obs) {
                wl->slot[lane] = make_float4((float)x, (float)y, boarding ? 0.0f : 1.0f,
                                             active ? 1.0f : 0.0f);
                wave_lds_sync();
                emit_obs<PAIR>(wl, table, obs_p, units, lane);
                obs_p += obs_stride;
                wave_lds_sync();
            }

            CCX_STAMP(5);   // observation gather + stores
            // ---- 6. auto-reset from the pool (reset() :97-150 with host-computed placements)
            const bool do_reset = use_pool && valid_env && (ef != 0u);
            if (do_reset) {
                ef |= CCX_K_EF_RESET;
                episode += 1;
                stepc = 0;
                c_episodes += (i == 0);
                if (valid) {
                    if (pnext_pending) {   // second reset inside one action burst (rare): wait here
                        asm volatile("; rare: reset twice within one action burst");
                        px = (int)(pnext & 0xFFu);
                        py = (int)(pnext >> 8);
                    }
                    x = px;
                    y = py;
                    active = true;
                    term = false;
                    trunc = false;
                    pool_idx += pool_stride;
                    if (pool_idx >= pool_size) pool_idx -= pool_size;
                    pnext = *reinterpret_cast<const uint16_t*>(pool_v + ((size_t)pool_idx * N + i) * 2);
                    pnext_pending = true;
                }
            }
            if (ef_p && valid_env && i == 0) *ef_p = (uint8_t)ef;
            if (ef_p) ef_p += p.E;
            CCX_STAMP(6);   // auto-reset + env flag store
        }
    }
#ifdef CCX_STAMPS
    if (ctr && wave == 0 && lane == 0)
        for (int q = 0; q < 8; ++q) atomicAdd(&ctr[8 + q], stamp_sum[q]);
#endif

    // ---- registers -> state ------------------------------------------------------------------
    if (valid) {
        *fx = x;
        *fy = y;
        *fact = active;
        *fterm = term;
        *ftrunc = trunc;
    }
    if (valid_env && i == 0) {
        *fstep = stepc;
        *fepi = episode;
    }
    if (ctr) {
        const uint32_t nenv = wave_sum_u32((valid_env && i == 0) ? 1u : 0u);
        const uint32_t moves = wave_sum_u32(c_moves), arrivals = wave_sum_u32(c_arrivals);
        const uint32_t lives = wave_sum_u32(c_live), eps = wave_sum_u32(c_episodes);
        if (lane == 0 && nenv) {
            atomicAdd(&ctr[0], (unsigned long long)nenv * (unsigned long long)K);
            atomicAdd(&ctr[1], (unsigned long long)nenv * (unsigned long long)K * N);
            atomicAdd(&ctr[2], (unsigned long long)lives);
            atomicAdd(&ctr[3], (unsigned long long)eps);
            atomicAdd(&ctr[4], (unsigned long long)moves);
            atomicAdd(&ctr[5], (unsigned long long)arrivals);
        }
    }
}

// DefaultObservation of the current state (what reset() returns, collectivecrossing.py:153-159)
template <int GLOG, bool PAIR>
__global__ void __launch_bounds__(256)
observe_kernel(const KParams p, const KState st, float* __restrict__ obs) {
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
    int x = 0, y = 0;
    bool active = false;
    if (valid) {
        const size_t idx = (size_t)env * N + i;
        x = st.x[idx];
        y = st.y[idx];
        active = st.active[idx] != 0;
    }
    wl->slot[lane] = make_float4((float)x, (float)y, (i < p.Nb) ? 0.0f : 1.0f, active ? 1.0f : 0.0f);
    __syncthreads();
    int envs_here = p.E - env0;
    envs_here = envs_here < 0 ? 0 : (envs_here > p.EW ? p.EW : envs_here);
    emit_obs<PAIR>(wl, table, obs + (size_t)env0 * N * L, envs_here * N * (3 + 2 * N), lane);
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
template <int GLOG>
static hipError_t launch_rollout_g(const LaunchShape& ls, hipStream_t stream, const KParams& p,
                                   const KState& st, const uint8_t* actions, const uint8_t* order,
                                   int K, int auto_reset, const uint8_t* pool, const KOut& out,
                                   unsigned long long* counters) {
    const bool pair = (p.N % 2) == 0;
    dim3 grid(ls.num_blocks), block(64 * ls.waves_per_block);
    if (pair)
        hipLaunchKernelGGL((rollout_kernel<GLOG, true>), grid, block, ls.lds_bytes, stream, p, st,
                           actions, order, K, auto_reset, pool, out, counters);
    else
        hipLaunchKernelGGL((rollout_kernel<GLOG, false>), grid, block, ls.lds_bytes, stream, p, st,
                           actions, order, K, auto_reset, pool, out, counters);
    return hipGetLastError();
}

template <int GLOG>
static hipError_t launch_observe_g(const LaunchShape& ls, hipStream_t stream, const KParams& p,
                                   const KState& st, float* obs) {
    const bool pair = (p.N % 2) == 0;
    dim3 grid(ls.num_blocks), block(64 * ls.waves_per_block);
    if (pair)
        hipLaunchKernelGGL((observe_kernel<GLOG, true>), grid, block, ls.lds_bytes, stream, p, st, obs);
    else
        hipLaunchKernelGGL((observe_kernel<GLOG, false>), grid, block, ls.lds_bytes, stream, p, st, obs);
    return hipGetLastError();
}

hipError_t launch_rollout(const LaunchShape& ls, hipStream_t stream, const KParams& p,
                          const KState& st, const uint8_t* actions, const uint8_t* order, int K,
                          int auto_reset, const uint8_t* pool, const KOut& out,
                          unsigned long long* counters) {
    switch (ls.glog) {
    case 0: return launch_rollout_g<0>(ls, stream, p, st, actions, order, K, auto_reset, pool, out, counters);
    case 1: return launch_rollout_g<1>(ls, stream, p, st, actions, order, K, auto_reset, pool, out, counters);
    case 2: return launch_rollout_g<2>(ls, stream, p, st, actions, order, K, auto_reset, pool, out, counters);
    case 3: return launch_rollout_g<3>(ls, stream, p, st, actions, order, K, auto_reset, pool, out, counters);
    case 4: return launch_rollout_g<4>(ls, stream, p, st, actions, order, K, auto_reset, pool, out, counters);
    case 5: return launch_rollout_g<5>(ls, stream, p, st, actions, order, K, auto_reset, pool, out, counters);
    case 6: return launch_rollout_g<6>(ls, stream, p, st, actions, order, K, auto_reset, pool, out, counters);
    }
    return hipErrorInvalidValue;
}

hipError_t launch_observe(const LaunchShape& ls, hipStream_t stream, const KParams& p,
                          const KState& st, float* obs) {
    switch (ls.glog) {
    case 0: return launch_observe_g<0>(ls, stream, p, st, obs);
    case 1: return launch_observe_g<1>(ls, stream, p, st, obs);
    case 2: return launch_observe_g<2>(ls, stream, p, st, obs);
    case 3: return launch_observe_g<3>(ls, stream, p, st, obs);
    case 4: return launch_observe_g<4>(ls, stream, p, st, obs);
    case 5: return launch_observe_g<5>(ls, stream, p, st, obs);
    case 6: return launch_observe_g<6>(ls, stream, p, st, obs);
    }
    return hipErrorInvalidValue;
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
