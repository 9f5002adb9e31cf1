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
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ccx_kernels.h"
#include "ccx_greedy.h"

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

// wait until at most n vector-memory operations of this wave are in flight (n rounded down to the
// next available immediate)
__device__ __forceinline__ void wait_vm_at_most(uint32_t n) {
    if (n >= 48) asm volatile("s_waitcnt vmcnt(48)" ::: "memory");
    else if (n >= 32) asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
    else if (n >= 24) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
    else if (n >= 20) asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
    else if (n >= 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else if (n >= 13) asm volatile("s_waitcnt vmcnt(13)" ::: "memory");
    else if (n >= 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
    else if (n >= 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (n >= 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else if (n >= 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if (n >= 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
}

// Workgroup barrier that orders LDS traffic ONLY: __syncthreads() would also wait for vmcnt(0),
// i.e. drain the writer wave's global stores at every step.
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
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

// Streaming stores of observation vectors (written once, never re-read by the kernel).  Cache policy
// "sc1 nt": under step pacing the stream drains 5 % faster than with plain `nt` (in-call, C2: 0.901 vs
// 0.856 of the HBM peak; "sc0 sc1 nt" the same).  The compiler has no builtin for the
// sc1 bit, hence the inline asm.
#ifndef CCX_STORE_BITS
#define CCX_STORE_BITS "sc1 nt"
#endif
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void store_obs(v4f v, v4f* dst) {
#ifdef CCX_PLAIN_STORES
    *dst = v;
#elif defined(CCX_BUILTIN_NT_STORES)   /* diagnostic: the compiler's nontemporal store (`nt` only) */
    __builtin_nontemporal_store(v, dst);
#else
    // s_nop: the "VMEM store of more than 64 bits followed by a VALU write of its data registers"
    // hazard is the compiler's job for its own instructions; it cannot see into this asm
    asm volatile("global_store_dwordx4 %0, %1, off " CCX_STORE_BITS "\n\ts_nop 1" ::"v"(dst), "v"(v) : "memory");
#endif
}
__device__ __forceinline__ void store_obs(float2 v, float2* dst) {   // odd agent counts: 8-byte units
#if defined(CCX_PLAIN_STORES) || defined(CCX_BUILTIN_NT_STORES)
    *dst = v;
#else
    v2f w = {v.x, v.y};
    asm volatile("global_store_dwordx2 %0, %1, off " CCX_STORE_BITS ::"v"(dst), "v"(w) : "memory");
#endif
}

__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// Keep a wave-uniform value in VGPRs on purpose: the step loop needs ~60 scalars (geometry,
// pointers, reward constants) next to the ballot masks, which overflows the 102 SGPRs and makes
// hipcc spill SGPRs through v_writelane/v_readlane inside the loop.  VGPRs are plentiful here
// (one or two waves per SIMD), so loop-invariant values that are only used by vector
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

// LDS tiles.  WSlot: what a wave that writes observation rows gathers from (one per writer wave /
// per wave of the observe kernel).  The rollout kernel's per-tile carve-up (byte offsets in
// KParams): [xch u32 x 64][stage uint4 x 2 x 64][WSlot x writers][occ | prp masks x EW x (cells+1)].
struct WSlot {
    float4 slot[64];   // (x, y, type, active) of the agent on each lane, as floats
    float cst[8];      // (door_centre, division_y) (door_left, door_right) (-1,-1) pad
};
static_assert(sizeof(WSlot) == 1024 + 32, "WSlot layout");
static constexpr uint32_t kCstOff = kObsCstOff;  // byte offset of cst[] from slot[]
using WaveLds = WSlot;

// The u16 observation address table (ccx_kernels.h: obs_unit_addr) comes from the host, like the cell
// table: computing it per launch cost ~1 us of integer divisions in every workgroup.
template <int GLOG>
__device__ __forceinline__ void build_obs_table(uint16_t* table, const KParams& p) {
    const uint32_t words = ((uint32_t)p.units_per_wave + 2u) >> 1;          // u16 pairs, incl. the pad entry
    const uint32_t* src = reinterpret_cast<const uint32_t*>(p.obs_table);
    uint32_t* dst = reinterpret_cast<uint32_t*>(table);
    for (uint32_t w = threadIdx.x; w < words; w += blockDim.x) dst[w] = src[w];
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

// copy a tile's observation region out of LDS, table-driven: vector units [first, n) of `dst`
template <bool PAIR>
__device__ __forceinline__ void emit_obs(const WaveLds* wl, const uint16_t* table, char* dst,
                                         int first, int n, int lane) {
    const char* sbase = reinterpret_cast<const char*>(wl);
    for (int q = first + lane; q < n; q += 64) {
        if constexpr (PAIR) {
            // N even: region start and length are multiples of 16 bytes
            const uint32_t t = reinterpret_cast<const uint32_t*>(table)[q];
            float2 a = *reinterpret_cast<const float2*>(sbase + (t & 0xFFFFu));
            float2 b = *reinterpret_cast<const float2*>(sbase + (t >> 16));
            v4f v = {a.x, a.y, b.x, b.y};
            store_obs(v, reinterpret_cast<v4f*>(dst + (size_t)q * 16));
        } else {
            store_obs(*reinterpret_cast<const float2*>(sbase + table[q]),
                      reinterpret_cast<float2*>(dst + (size_t)q * 8));
        }
    }
}

// Diagnostic build only (-DCCX_STAMPS): s_memtime stamps around the segments of one step; the
// waves of block 0 add their per-segment cycle sums to counters[8..15].  Never in libccx.so.
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
#define CCX_STAMP_DECL                                                                      \
    unsigned long long stamp_sum[4] = {0, 0, 0, 0}, t_prev = 0;                             \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_prev)::"memory")
#define CCX_STAMP_FLUSH(ctr, base)                                                          \
    if ((ctr) && blockIdx.x == 0 && lane == 0)                                              \
        for (int q_ = 0; q_ < 4; ++q_) atomicAdd(&(ctr)[8 + (base) + q_], stamp_sum[q_])
#else
#define CCX_STAMP(slot) do { } while (0)
#define CCX_STAMP_DECL do { } while (0)
#define CCX_STAMP_FLUSH(ctr, base) do { } while (0)
#endif

constexpr int kActBatch = 16;      // env-steps of actions fetched per global-load burst
constexpr int kFastObsIters = 10;  // observation store iterations whose LDS addresses live in VGPRs
constexpr int kObsBatch = 5;       // LDS reads issued back to back before their stores

// ---- per-cell geometry table ------------------------------------------------------------------
// Everything the step needs to know about a grid cell is precomputed once per handle on the host
// (ccx_api.hip: build_cell_table) for the padded grid x in [-1, W+1], y in [-1, H+1]
// (cell = (y+1)*(W+3) + (x+1)) and copied to LDS at kernel start:
//   lo: bits 0-3  move a (right, up, left, down) from this cell lands on a cell that is in the
//                 grid and not a wall                       (collectivecrossing.py:509-534)
//       bit4 IN_TRAM_AREA (:551-554)  bit5 AT_DOOR (:556-563)       -- same bits as CCX_AF_*
//       bit8  boarding: on destination row (:663-683)   bits 9-10  boarding reward class
//       bit12 exiting:  on destination row              bits 13-14 exiting reward class
//       byte2 = x, byte3 = y  (0 for border cells)
//   hi: int16 signed distance term of the boarding reward | int16 of the exiting reward << 16
// reward class (rewards.py:44-182): 0 = (double)sd * distance_penalty_factor, 1/2/3 = constants
// rA/rB/rC chosen per reward mode.  The word of the agent's CURRENT cell is carried in registers,
// so the legality of a move is a bit test; the word of the proposed cell is fetched off the
// critical path and only consumed once the move is known to happen.
//
// sim -> writer hand-off (uint4 per lane and step): x = cell lo, y = cell hi, z = the CCX_AF_*
// byte of the agent, w = the CCX_EF_* byte of its env.

// ---------------------------------------------------------------------------------------------
// the fused rollout / step kernel.
//   OUT  trajectory outputs requested: a tile is served by 1 sim wave + p.writers writer waves;
//        OUT = false: sim waves only (counters only).
//   OCC  conflict masks come from per-env occupancy / proposal bit tables in LDS (O(1) per agent);
//        OCC = false: all-pairs compare through the xch tile (grids whose tables exceed LDS).
// wave index in block -> role = wib / tiles_per_block (0 = sim, 1.. = writer), tile = wib % tpb.
// ---------------------------------------------------------------------------------------------
#ifdef CCX_TSTAMPS   // diagnostic (profiles/scratch/tstamps.py): raw s_memtime at fixed points of tile 0's sim wave
#define CCX_T(q) do { if (counters && blockIdx.x == 0 && threadIdx.x == 0) { unsigned long long t_; \
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory"); \
    counters[8 + (q)] = t_; } } while (0)
#else
#define CCX_T(q) do { } while (0)
#endif

//   PLAIN  the caller passed neither a move order nor a policy (the bench line, plain RL stepping): the step
//        loop is compiled without those branches (12 % fewer cycles per env-step on the sim chain).
//   OUTM 0 = no trajectory outputs, 1 = outputs, 2 = outputs whose tile regions do not begin / end on 128-byte
//        lines (edge iterations, below; a separate instantiation because the single-writer C2 path loses 4-5 %
//        to ANY extra instruction in its store loop, even a never-taken branch)
template <int GLOG, bool PAIR, int OUTM, bool OCC, bool PLAIN>
__global__ void __launch_bounds__(512)
rollout_kernel(const KParams p, const KState st, const unsigned long long* __restrict__ cell_info,
               const uint8_t* __restrict__ actions, const uint8_t* __restrict__ order, const int K,
               const int auto_reset, const uint8_t* __restrict__ pool, const KOut out,
               unsigned long long* counters, const int policy_arg, uint8_t* __restrict__ actions_out) {
    constexpr bool OUT = OUTM != 0, EDGE = OUTM == 2;
    const int policy = PLAIN ? 0 : policy_arg;
    using mask_t = typename GroupMask<GLOG>::type;
    constexpr int G = 1 << GLOG;
    extern __shared__ __align__(16) unsigned char smem[];

    CCX_T(0);
    const int lane = threadIdx.x & 63;
    const int wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform -> SGPR
    const int tpb = p.waves_per_block;                                  // tiles per block
    const int role = OUT ? wib / tpb : 0;                               // 0 sim, 1.. writers
    const int tile_in_block = OUT ? wib - role * tpb : wib;
    // XCD-aware tile mapping: workgroups are dealt round-robin to the 8 XCDs (blockIdx % 8), each
    // with its own L2 and path to memory.  Workgroups of one XCD take ADJACENT tiles, so an XCD
    // writes one contiguous eighth of every step's output slab instead of every eighth chunk --
    // the pace at which the output stream collapses moves from ~740 to ~700 ns per env-step on C2
    // (profiles/scratch/tile_times.py).
    // tile_map: 0 = XCD-contiguous (above); v >= 1 = groups of 2^(v-1) adjacent tiles per XCD, dealt
    // round-robin (1 = the plain blockIdx -> tile mapping): with the tiles phased in tile order
    // (pace_phase 1) the chip writes ONE window that sweeps through the slab, fed by all eight XCDs,
    // while the small partial-line outputs of a group still meet in one L2.
    int bid = blockIdx.x;
    if (p.tile_map == 0u) {
        const int nb = gridDim.x, xcd = blockIdx.x & 7, q8 = nb >> 3, r8 = nb & 7;
        bid = xcd * q8 + (xcd < r8 ? xcd : r8) + (int)(blockIdx.x >> 3);
    } else if (p.tile_map > 1u) {
        const uint32_t m = p.tile_map - 1u, span = 8u << m, b = blockIdx.x;
        if (b < (gridDim.x / span) * span)       // (a tail that does not fill a span keeps the plain mapping)
            bid = (int)((b & ~(span - 1u)) + ((b & 7u) << m) + ((b >> 3) & ((1u << m) - 1u)));
    }
    const int tile = bid * tpb + tile_in_block;
    const int g = lane >> GLOG;
    const int i = lane & (G - 1);
    const int gbase = g << GLOG;
    const int env0 = tile * p.EW;                 // first env of this tile
    const int env = env0 + g;
    const bool valid_env = (g < p.EW) && (env < p.E);
    const bool valid = valid_env && (i < p.N);
    const uint32_t validbit = valid ? 1u : 0u;
    const int N = p.N;
    const int L = 6 + 4 * N;
    const size_t EN = (size_t)p.E * N;
    const size_t idx = (size_t)env * N + i;
    const bool boarding = i < p.Nb;
    const uint32_t tsh = boarding ? 8u : 12u;     // type-specific nibble of the cell word
    const uint32_t tsh2 = boarding ? 0u : 16u;    // type-specific half of the distance word
    const int Wp = p.W + 3;

    // LDS carve-up: [cell table][tile 0 .. tile tpb-1][u16 obs table]; offsets from the host
    const uint32_t cells = (uint32_t)(Wp * (p.H + 3));
    unsigned long long* cinfo = reinterpret_cast<unsigned long long*>(smem);
    unsigned char* tbase = smem + p.off_tiles + (uint32_t)tile_in_block * p.tile_stride;
    uint32_t* xch = reinterpret_cast<uint32_t*>(tbase);
    uint4* stage = reinterpret_cast<uint4*>(tbase + 256);
    uint16_t* table = reinterpret_cast<uint16_t*>(smem + p.off_table);
    const bool want_obs = OUT && out.obs != nullptr;
    // Prologue: both host-built tables come from global memory (L2-resident after the first
    // workgroup).  The first chunk of each is requested before anything else so that ONE memory
    // latency covers both and the LDS zeroing below; small grids (C2: 143 cells, 609 table words)
    // need nothing more.
    const uint32_t tw = want_obs ? (((uint32_t)p.units_per_wave + 2u) >> 1) : 0u;   // u16 pairs of the obs table
    const uint32_t* tsrc = reinterpret_cast<const uint32_t*>(p.obs_table);
    const unsigned long long c_first = threadIdx.x < cells ? cell_info[threadIdx.x] : 0ull;
    uint32_t t_first[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const uint32_t w = threadIdx.x + (uint32_t)r * blockDim.x;
        t_first[r] = w < tw ? tsrc[w] : 0u;
    }
    const bool sim_wave = !OUT || role == 0;
    if constexpr (OCC) {   // zero the occupancy / proposal tables of every tile of the block
        for (int ti = 0; ti < tpb; ++ti) {
            uint32_t* occ = reinterpret_cast<uint32_t*>(smem + p.off_tiles + (uint32_t)ti * p.tile_stride + p.off_occ);
            for (uint32_t w = threadIdx.x; w < p.occ_words; w += blockDim.x) occ[w] = 0u;
        }
    }
    if (threadIdx.x < cells) cinfo[threadIdx.x] = c_first;
    for (uint32_t t = threadIdx.x + blockDim.x; t < cells; t += blockDim.x) cinfo[t] = cell_info[t];
    {
        uint32_t* tdst = reinterpret_cast<uint32_t*>(table);
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const uint32_t w = threadIdx.x + (uint32_t)r * blockDim.x;
            if (w < tw) tdst[w] = t_first[r];
        }
        for (uint32_t w = threadIdx.x + 3u * blockDim.x; w < tw; w += blockDim.x) tdst[w] = tsrc[w];
    }
    CCX_T(1);
    __syncthreads();  // tables are read-only / zeroed from here on
    CCX_T(2);

    int envs_here = p.E - env0;
    envs_here = envs_here < 0 ? 0 : (envs_here > p.EW ? p.EW : envs_here);
    const int units = envs_here * N * (3 + 2 * N);
    const int n4 = PAIR ? (units >> 1) : units;     // vector units (16 B or 8 B) of the obs region

    // =========================================================================================
    // WRITER waves: reward, flag bytes, observation rows of every step (reference :214-261).
    // Writer w of nw takes the store iterations it = w, w + nw, ...; writer 0 also writes the
    // reward and the flag bytes.
    // =========================================================================================
    if constexpr (OUT) {
        if (role > 0) {
            const int w = role - 1, nw = p.writers;
            WSlot* wl = reinterpret_cast<WSlot*>(tbase + p.off_ws) + w;
            init_wave_consts(wl, p, lane);
            const float type_f = boarding ? 0.0f : 1.0f;  // observations.py:85
            // reward constants (rewards.py:44-182): class 1/2/3 -> rA/rB/rC, class 0 -> sd * rF
            const int rmode = p.reward_mode;
            const double rA = in_vgpr(rmode == CCX_K_REWARD_BINARY ? p.r_nogoal
                                      : rmode == CCX_K_REWARD_CONSTANT_NEGATIVE ? p.r_pen : p.r_dest);
            const double rB = in_vgpr(p.r_door);
            const double rC = in_vgpr(p.r_area);
            const double rF = in_vgpr(p.r_f);
            // wave-uniform base pointers advanced per step on the scalar unit + per-lane 32-bit
            // byte offsets that never change
            const bool small_out = (w == 0);
            const uint32_t rew_off = (uint32_t)idx * 8u, af_off = (uint32_t)idx, ef_off = (uint32_t)env;
            char* rew_s = small_out ? reinterpret_cast<char*>(out.reward) : nullptr;
            char* af_s = small_out ? reinterpret_cast<char*>(out.agent_flags) : nullptr;
            char* ef_s = small_out ? reinterpret_cast<char*>(out.env_flags) : nullptr;
            char* act_s = small_out ? reinterpret_cast<char*>(actions_out) : nullptr;   // policy rollouts
            // compact observation: the (x, y, type, active) of this lane's agent, 16 bytes per agent-step
            // instead of the 16N + 24 of the DefaultObservation rows that repeat it N times
            char* cmp_s = small_out ? reinterpret_cast<char*>(out.obs_compact) : nullptr;
            const uint32_t cmp_off = (uint32_t)idx * 16u;
            char* obs_s = reinterpret_cast<char*>(out.obs) + (size_t)env0 * N * L * 4;
            const size_t obs_stride = EN * (size_t)L * 4;
            // Store iterations are laid out on 128-byte lines of GLOBAL memory, not from the start of
            // the tile's region: a region that starts mid-line (most agent counts; N = 50: 41200 B
            // per env) would otherwise make every 1-KiB wave store straddle two partially written
            // lines (measured 4.1 vs 6.1 TB/s).  `lead` = vector units between the line boundary
            // below the region and its start; constant over the steps when the slab stride is a
            // multiple of 128 bytes (else 0: the old, region-relative layout).
            const uint32_t vbytes = PAIR ? 16u : 8u;
            const uint32_t lead = __builtin_amdgcn_readfirstlane(
                ((obs_stride & 127u) == 0 && want_obs)
                    ? (uint32_t)(reinterpret_cast<uintptr_t>(obs_s) & 127u) / vbytes : 0u);
            obs_s -= lead * vbytes;                       // line-aligned base; unit q lives at (q + lead)
            const int n4l = n4 + (int)lead;               // one past the last slot of the shifted layout
            // my store iterations: it0, it0 + it_step, ... below it_end (one iteration = 64 vector units)
            const int total_its = (n4l + 63) >> 6;
            // each writer takes one contiguous share of the region (+0.5 % on C3 / C5 under pacing; interleaving
            // the writers' 1-KiB iterations instead was +1 % on C3 with 3 writers, -1..2 % on C5-50, and the
            // run-time stride cost the single-writer C2 path 3-5 %: not kept)
            const int per_w = (total_its + nw - 1) / nw;
            const int it0 = w * per_w;
            constexpr int it_step = 1;
            const int it_end = (it0 + per_w) < total_its ? (it0 + per_w) : total_its;
            // Edge iterations.  A region that does not begin / end on a 128-byte line (N = 50: rows of 824 bytes)
            // shares its first and last line with the neighbouring tiles' regions: two partial writes of one
            // line from two workgroups.  Streamed (`nt`) they reach memory as two partial-line writes; written
            // with plain stores the first and the last iteration of a region stay in L2, where the halves
            // merge (the grouped tile map keeps neighbours on one XCD) -- C5-50 0.83 -> 0.86 of the HBM peak,
            // aligned shapes (C2, C3, C5-64) have no edge iteration and take the unchanged path.
            const int edge_first = (EDGE && lead != 0u && it0 == 0) ? 0 : -1;
            const int edge_last = (EDGE && (((uint32_t)n4l * vbytes) & 127u) != 0u && it_end == total_its) ? total_its - 1 : -1;
            uint32_t edge_mask = 0;   // over my register-cached iterations
            if (EDGE && want_obs) {
                if (edge_first >= 0) edge_mask |= 1u;
                if (edge_last >= it0 && edge_last - it0 < kFastObsIters) edge_mask |= 1u << (edge_last - it0);
            }
            edge_mask = __builtin_amdgcn_readfirstlane(edge_mask);
            // LDS source addresses of this lane's first kFastObsIters observation stores
            uint32_t oa0[kFastObsIters], oa1[kFastObsIters];
#pragma unroll
            for (int j = 0; j < kFastObsIters; ++j) {
                const uint32_t q = (uint32_t)(lane + 64 * (it0 + it_step * j)) - lead;   // wraps below the region
                oa0[j] = oa1[j] = kCstOff + 16u;
                if (want_obs && (it0 + it_step * j) < it_end && q < (uint32_t)n4) {
                    if constexpr (PAIR) {
                        const uint32_t t = reinterpret_cast<const uint32_t*>(table)[q];
                        oa0[j] = t & 0xFFFFu;
                        oa1[j] = t >> 16;
                    } else {
                        oa0[j] = table[q];
                    }
                }
            }
            const uint32_t q0_off = (uint32_t)(lane + 64 * it0) * vbytes;       // byte offset of my first iteration
            const uint32_t it_stride = 64u * (uint32_t)it_step * vbytes;        // between my iterations
            const char* sbase = reinterpret_cast<const char*>(wl);
            CCX_STAMP_DECL;

            for (int s = 0; s < K; ++s) {
                lds_barrier();                       // the sim wave has staged step s
                const uint4 e = stage[(s & 1) * 64 + lane];
                CCX_STAMP(0);                        // wait for the sim wave
                const uint32_t ilo = e.x, ihi = e.y, af = e.z;
                if (want_obs || cmp_s) {
                    // (x, y) are bytes 2 and 3 of the cell word: v_cvt_f32_ubyte2 / ubyte3
                    const float4 me = make_float4((float)((ilo >> 16) & 0xFFu), (float)(ilo >> 24), type_f,
                                                  (float)((af >> 6) & 1u));
                    if (want_obs) wl->slot[lane] = me;
                    if (cmp_s) {
                        if (valid) *reinterpret_cast<float4*>(cmp_s + cmp_off) = me;
                        cmp_s += EN * 16;
                    }
                }
                if (small_out) {
                    // rewards.py:44-182.  Distances are integers and the reference negates the
                    // INTEGER before the one f64 multiply, so d == 0 gives +0.0 (never -0.0).
                    const uint32_t cls = (ilo >> (tsh + 1u)) & 3u;
                    const int sd = (int)(int16_t)(uint16_t)(ihi >> tsh2);
                    double r = (double)sd * rF;
                    r = (cls == 1u) ? rA : r;
                    r = (cls == 2u) ? rB : r;
                    r = (cls == 3u) ? rC : r;
                    r = (af & 0x04u) ? r : 0.0;      // rewards.py:64: None unless live
                    if (valid) {
                        if (rew_s) *reinterpret_cast<double*>(rew_s + rew_off) = r;
                        if (af_s) *reinterpret_cast<uint8_t*>(af_s + af_off) = (uint8_t)af;
                        if (ef_s && i == 0) *reinterpret_cast<uint8_t*>(ef_s + ef_off) = (uint8_t)e.w;
                        if (act_s) *reinterpret_cast<uint8_t*>(act_s + af_off) = (uint8_t)(af >> 8);
                    }
                    if (act_s) act_s += EN;
                    if (rew_s) rew_s += EN * 8;
                    if (af_s) af_s += EN;
                    if (ef_s) ef_s += p.E;
                }
                CCX_STAMP(1);                        // reward + flag bytes
                if (want_obs) {
                    wave_lds_sync();
                    // all LDS reads of a batch first (idle lanes read the constant slot), so the
                    // wave pays one LDS latency per batch, then the stores
#pragma unroll
                    for (int j0 = 0; j0 < kFastObsIters; j0 += kObsBatch) {
                        if ((it0 + it_step * j0) < it_end) {
                            float2 va[kObsBatch], vb[kObsBatch];
#pragma unroll
                            for (int j = 0; j < kObsBatch; ++j) {
                                va[j] = *reinterpret_cast<const float2*>(sbase + oa0[j0 + j]);
                                if constexpr (PAIR)
                                    vb[j] = *reinterpret_cast<const float2*>(sbase + oa1[j0 + j]);
                            }
                            const uint32_t em = EDGE ? (edge_mask >> j0) & ((1u << kObsBatch) - 1u) : 0u;   // wave-uniform
                            if (!EDGE || em == 0u) {
#pragma unroll
                                for (int j = 0; j < kObsBatch; ++j) {
                                    const uint32_t q = (uint32_t)(lane + 64 * (it0 + it_step * (j0 + j))) - lead;
                                    if ((it0 + it_step * (j0 + j)) < it_end && q < (uint32_t)n4) {
                                        char* dst = obs_s + (q0_off + (uint32_t)(j0 + j) * it_stride);
                                        if constexpr (PAIR) {
                                            v4f v = {va[j].x, va[j].y, vb[j].x, vb[j].y};
                                            store_obs(v, reinterpret_cast<v4f*>(dst));
                                        } else {
                                            store_obs(va[j], reinterpret_cast<float2*>(dst));
                                        }
                                    }
                                }
                            } else {   // a batch that holds the region's first or last iteration (misaligned regions only)
#pragma unroll
                                for (int j = 0; j < kObsBatch; ++j) {
                                    const uint32_t q = (uint32_t)(lane + 64 * (it0 + it_step * (j0 + j))) - lead;
                                    if ((it0 + it_step * (j0 + j)) < it_end && q < (uint32_t)n4) {
                                        char* dst = obs_s + (q0_off + (uint32_t)(j0 + j) * it_stride);
                                        const bool edge = (em >> j) & 1u;
                                        if constexpr (PAIR) {
                                            v4f v = {va[j].x, va[j].y, vb[j].x, vb[j].y};
                                            if (edge) *reinterpret_cast<v4f*>(dst) = v;
                                            else store_obs(v, reinterpret_cast<v4f*>(dst));
                                        } else {
                                            if (edge) *reinterpret_cast<float2*>(dst) = va[j];
                                            else store_obs(va[j], reinterpret_cast<float2*>(dst));
                                        }
                                    }
                                }
                            }
                        }
                    }
                    // beyond the register-cached iterations: table-driven
                    for (int it = it0 + it_step * kFastObsIters; it < it_end; it += it_step) {
                        const int ql = lane + 64 * it;
                        const uint32_t q = (uint32_t)ql - lead;
                        if (q >= (uint32_t)n4) continue;
                        if constexpr (PAIR) {
                            const uint32_t t = reinterpret_cast<const uint32_t*>(table)[q];
                            float2 a2 = *reinterpret_cast<const float2*>(sbase + (t & 0xFFFFu));
                            float2 b2 = *reinterpret_cast<const float2*>(sbase + (t >> 16));
                            v4f v = {a2.x, a2.y, b2.x, b2.y};
                            if (EDGE && it == edge_last) *reinterpret_cast<v4f*>(obs_s + (size_t)ql * 16) = v;
                            else store_obs(v, reinterpret_cast<v4f*>(obs_s + (size_t)ql * 16));
                        } else {
                            const float2 v2 = *reinterpret_cast<const float2*>(sbase + table[q]);
                            if (EDGE && it == edge_last) *reinterpret_cast<float2*>(obs_s + (size_t)ql * 8) = v2;
                            else store_obs(v2, reinterpret_cast<float2*>(obs_s + (size_t)ql * 8));
                        }
                    }
#ifndef CCX_SAME_SLAB   /* diagnostic: every step overwrites slab 0 (L2-resident) */
                    obs_s += obs_stride;
#endif
                    wave_lds_sync();
                }
                CCX_STAMP(2);                        // observation gather + stores
                // store throttle: many small tiles oversubscribe the HBM write queues and the drain
                // rate of the whole chip drops (measured: DESIGN.md 3.6); bounding the stores a
                // writer keeps in flight keeps the memory side in its efficient regime
                if (p.writer_vmcnt) wait_vm_at_most(p.writer_vmcnt);
            }
            if (w == 0) { CCX_STAMP_FLUSH(counters, 4); }
            return;
        }
    }

    // =========================================================================================
    // SIM wave: the state transition (reference phases :188-212, :219-241 and the auto-reset)
    // =========================================================================================
    const mask_t full = full_mask<GLOG>();
    const mask_t lo_m = low_mask<mask_t>(i);
    const mask_t later_m = ~lo_m & ~(mask_t(1) << i);
    // occupancy / proposal bit tables of this lane's env: [cells + 1] masks each, the last entry is a dump
    // slot for agents that have nothing to publish.  Lanes WITHOUT an agent (half-empty waves of small
    // batches, padded lane groups) get dump words of their own on the top border row (y = -1, never
    // occupied): 32+ lanes hammering one dump word with same-address LDS atomics cost such tiles 10 % per
    // step (4096 x 8 at 2048 envs: 0.61 -> 0.55 us), while steering the few idle agents of an env to its one
    // dump word is cheaper than letting them OR a zero into their own cells (0.54 vs 0.565 us per step).
    constexpr uint32_t msz = sizeof(mask_t);
    const uint32_t cells1 = cells + 1u;
    const uint32_t g_tab = (g < p.EW) ? (uint32_t)g : 0u;   // lanes beyond the tile's envs use env 0's tables (zeros only)
    const uint32_t occ_base = p.off_tiles + (uint32_t)tile_in_block * p.tile_stride + p.off_occ +
                              g_tab * 2u * cells1 * msz;
    const uint32_t prp_base = occ_base + cells1 * msz;
    const uint32_t dump_off = valid ? cells * msz : (uint32_t)(lane % Wp) * msz;

    // ---- state -> registers ------------------------------------------------------------------
    int c = Wp + 1;               // cell index of (0,0)
    int stepc = 0, episode = 0;
    uint32_t act = 0, term = 0, trunc = 0;   // 0/1
    if (valid) {
        c = (st.y[idx] + 1) * Wp + st.x[idx] + 1;
        act = st.active[idx] != 0;
        term = st.terminated[idx] != 0;
        trunc = st.truncated[idx] != 0;
    }
    if (valid_env) {
        stepc = st.step_count[env];
        episode = st.episode[env];
    }
    unsigned long long ci = cinfo[c];
    uint32_t ilo = (uint32_t)ci, ihi = (uint32_t)(ci >> 32);
    // final write-back addresses, parked in VGPRs for the duration of the loop
    int32_t* const fx = in_vgpr(st.x + idx);
    int32_t* const fy = in_vgpr(st.y + idx);
    uint8_t* const fact = in_vgpr(st.active + idx);
    uint8_t* const fterm = in_vgpr(st.terminated + idx);
    uint8_t* const ftrunc = in_vgpr(st.truncated + idx);
    int32_t* const fstep = in_vgpr(st.step_count + env);
    int32_t* const fepi = in_vgpr(st.episode + env);
    unsigned long long* const ctr = in_vgpr(counters);
    const uint32_t term_all = (p.term_mode == CCX_K_TERM_ALL) ? 1u : 0u;

    // move deltas in the padded grid, one signed byte per action 0..4 (actions.py:18-24)
    const unsigned long long lut = (unsigned long long)(uint8_t)1 | ((unsigned long long)(uint8_t)Wp << 8) |
                                   ((unsigned long long)(uint8_t)(-1) << 16) |
                                   ((unsigned long long)(uint8_t)(-Wp) << 24);

    // reset-pool cursor of this env: entry (global_env + episode*total) mod P, advanced by
    // total mod P per episode; the NEXT placement is prefetched right after every reset and only
    // decoded when it is consumed (so no wait sits behind the load).
    uint32_t pool_idx = 0, pnext = 0;
    int pcell = Wp + 1;
    bool pnext_pending = false;   // pnext holds a load that has not been decoded into pcell yet
    const bool use_pool = auto_reset && pool != nullptr && p.pool_size > 0;
    const uint32_t pool_size = (uint32_t)p.pool_size, pool_stride = (uint32_t)p.pool_stride;
    if (use_pool && valid) {
        unsigned long long P = (unsigned long long)p.pool_size;
        unsigned long long gi = (unsigned long long)(p.env_offset + env) % P;
        unsigned long long ep = (unsigned long long)(episode + 1) % P;
        pool_idx = (uint32_t)((gi + ep * (unsigned long long)p.pool_stride) % P);
        pnext = *reinterpret_cast<const uint16_t*>(pool + ((size_t)pool_idx * N + i) * 2);
        pnext_pending = true;
    }

    const bool has_order = PLAIN ? false : order != nullptr;   // wave-uniform
    const uint8_t* ord_p = order + idx;

    uint32_t c_moves = 0, c_arrivals = 0, c_live = 0, c_episodes = 0;
    CCX_STAMP_DECL;

    const uint8_t* act_p = actions + idx;
    // ---- actions: bursts of kActBatch steps, 4 bits per step ----------------------------------
    // One s_waitcnt vmcnt per burst instead of one per step, addressed from a VGPR stride.
    uint32_t araw[kActBatch];
    const size_t EN_v = in_vgpr(EN);
    auto fetch_actions = [&](int s_first) {
        const int last = K - 1 - s_first;   // >= 0 whenever this is called
#pragma unroll
        for (int d = 0; d < kActBatch; ++d) {
            // (steps past K are never consumed: no load for them -- a single-step launch issues ONE action
            // load, not sixteen; the branch is wave-uniform)
            araw[d] = 4u;
            if (d <= last && sim_wave && valid) araw[d] = (uint32_t)act_p[(size_t)d * EN_v];
        }
        act_p += (size_t)kActBatch * EN_v;
    };
    if (!policy) fetch_actions(0);
    CCX_T(3);

    // ---- step pacing: a smooth, absolute schedule for the output stream ----------
    uint32_t pace = 0, pace_base = 0, pace_floor = 0, pace_skip = 0;
    unsigned long long pace_t0 = 0, pace_due = 0;   // ticks, ticks x 256 since t0
    if (want_obs && p.pace_state && K >= 16) {   // (short launches: not worth the load)
        // pace_state: [0], [1] = the pace slots (one is read, the other collects this launch's votes),
        // [2] = floor: the pace just above the last collapse; it decays by 0.1 % per launch at first
        // and twice as fast after every 8 further launches without a collapse ([3] counts them), so
        // that a transient (the first milliseconds of a process collapse at paces that are fine
        // later) does not hold the pace up for long while a persistent cliff is approached slowly;
        // [4]..[7] = cliff memory (below)
        const uint32_t voted = *reinterpret_cast<volatile const uint32_t*>(p.pace_state + p.pace_slot);
        const uint32_t floor_fp = *reinterpret_cast<volatile const uint32_t*>(p.pace_state + 2);
        pace_base = __builtin_amdgcn_readfirstlane(voted > floor_fp ? voted : floor_fp);
        pace_floor = __builtin_amdgcn_readfirstlane(floor_fp);
        // The common pace is per workgroup for a FULL device (p.resident_blocks at once).  A grid of
        // more workgroups runs in rounds; the workgroups of a partial last round share the same
        // memory among fewer, so their schedule is proportionally faster (else 1024 workgroups on 768
        // slots would spend their second round at a third of the drain rate).
        pace = pace_base;
        if (p.resident_blocks && gridDim.x > p.resident_blocks) {
            const uint32_t first = (blockIdx.x / p.resident_blocks) * p.resident_blocks;
            const uint32_t left = gridDim.x - first;
            if (left < p.resident_blocks) {
                pace = (uint32_t)(((unsigned long long)pace_base * left) / p.resident_blocks);
                pace = pace ? pace : 1u;
            }
        }
        if (p.pace_adapt && K >= 64 && tile == 0 && lane == 0) {
            p.pace_state[p.pace_slot ^ 1u] = 0u;               // the votes of this launch are collected here
            const uint32_t streak = *reinterpret_cast<volatile const uint32_t*>(p.pace_state + 3);
            // Cliff memory: the collapsed tiles of the previous launch left its pace in [6].  A collapse
            // within 3 % of the one before ([4]) confirms where the cliff of this box and shape is ([5]
            // counts confirmations); from then on the floor is let down 40 times more slowly (0.024 % per
            // launch, no acceleration), so the controller sits ~2 % above a cliff it knows instead of
            // walking back into it every 15-20 launches.  Collapses during the first 16 launches of a
            // controller do not count (a process collapses at paces that are fine a few milliseconds later).
            const uint32_t mark = *reinterpret_cast<volatile const uint32_t*>(p.pace_state + 6);
            uint32_t confirmed = *reinterpret_cast<volatile const uint32_t*>(p.pace_state + 5);
            const uint32_t launches = *reinterpret_cast<volatile const uint32_t*>(p.pace_state + 7);
            if (mark) {
                const uint32_t cliff = *reinterpret_cast<volatile const uint32_t*>(p.pace_state + 4);
                const uint32_t gap = mark > cliff ? mark - cliff : cliff - mark;
                if (launches >= 16u) confirmed = (cliff && gap < (cliff >> 5)) ? (confirmed < 8u ? confirmed + 1u : 8u) : 1u;
                p.pace_state[4] = mark;
                p.pace_state[5] = confirmed;
                p.pace_state[6] = 0u;
            }
            p.pace_state[7] = launches < 100000u ? launches + 1u : launches;
            const uint32_t sh = confirmed >= 2u ? 12u : 10u - (streak >= 32u ? 4u : streak >> 3);
            p.pace_state[2] = floor_fp - (floor_fp >> sh);
            p.pace_state[3] = streak < 1000u ? streak + 1u : streak;
        }
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(pace_t0)::"memory");
        if (p.pace_phase) {
            // tunable: the tiles of a round do not start their steps together but spread over the step
            // period -- 1: in tile order (the chip then writes one narrow window that sweeps through the
            // step's slab), 2: hashed.  Only the tile's own t0 moves; the schedule and the controller's
            // lateness test are relative to it.
            const uint32_t per_round = (p.resident_blocks ? p.resident_blocks : gridDim.x) * (uint32_t)tpb;
            const uint32_t tr = (uint32_t)tile % per_round;
            uint32_t frac16;
            if (p.pace_phase == 1u) {
                frac16 = (tr << 16) / per_round;                       // one window for the whole chip
            } else if (p.pace_phase == 3u) {
                const uint32_t per_xcd = per_round >> 3;               // one window per XCD (tile_map 0)
                frac16 = per_xcd ? ((tr % per_xcd) << 16) / per_xcd : 0u;
            } else {
                frac16 = __brev(tr * 2654435761u) & 0xFFFFu;
            }
            pace_t0 += ((unsigned long long)pace * frac16) >> 24;
        }
    }

    int s = 0;
    for (int s0 = 0; s0 < K; s0 += kActBatch) {
        // the burst issued one batch ago is consumed here: ONE vmcnt wait per kActBatch steps.
        // Actions are clamped to 0..4 (4 = wait; 255 = absent and anything else: no move).
        uint32_t apk[2] = {0, 0};
#pragma unroll
        for (int d = 0; d < kActBatch; ++d) {
            const uint32_t a4 = araw[d] < 4u ? araw[d] : 4u;
            apk[d >> 3] |= a4 << (4 * (d & 7));
        }
        apk[0] = in_vgpr(apk[0]);
        apk[1] = in_vgpr(apk[1]);
        if (pnext_pending) {   // decode behind the wait that just happened: costs nothing
            pcell = (int)(pnext >> 8) * Wp + (int)(pnext & 0xFFu) + Wp + 1;
            pnext_pending = false;
        }
        __builtin_amdgcn_sched_barrier(0);   // keep the next burst BEHIND the wait above
        if (!policy && s0 + kActBatch < K) fetch_actions(s0 + kActBatch);
        __builtin_amdgcn_sched_barrier(0);
        const int dmax = (K - s0) < kActBatch ? (K - s0) : kActBatch;

        uint32_t acur = apk[0];
        for (int d = 0; d < dmax; ++d, ++s) {
            if (pace) {   // env-step s is due at t0 + s * pace; a late tile does not wait (it catches up)
                const unsigned long long due = pace_t0 + (pace_due >> 8);
                pace_due += pace;
                if (pace_skip) {
                    --pace_skip;             // clearly behind schedule a moment ago: do not even look
                } else {
                    unsigned long long now;
                    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(now)::"memory");
                    // reading the clock stalls the step chain (scalar-memory round trip): a tile that is
                    // more than half a step behind (batches too small to be memory-bound, or a collapse)
                    // checks again only three steps later
                    if ((long long)(now - due) > (long long)(pace >> 9)) pace_skip = 3;
                    while ((long long)(due - now) > 0) {
                        __builtin_amdgcn_s_sleep(1);
                        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(now)::"memory");
                    }
                }
            }
            if (d == 8) acur = apk[1];
            uint32_t a = acur & 0xFu;
            acur >>= 4;

            // ---- move rank of this agent (dict order of action_dict, collectivecrossing.py:197)
            int rank = i;
            if (has_order) {
                uint32_t ok_ = valid ? (uint32_t)*ord_p : (uint32_t)i;  // agent moved i-th
                ord_p += EN;
                xch[gbase + (ok_ & (G - 1))] = (uint32_t)i;
                wave_lds_sync();
                rank = (int)xch[lane];
                wave_lds_sync();
            }

            // ---- 0. policy-driven rollout: the reference's GreedyPolicy(epsilon = 0) picks the action
            //         from the pre-step state (greedy_policy.py:33-449).  The occupancy bits are
            //         published first; a direction is free if the current cell's word says the
            //         neighbour is enterable and no other active agent's bit sits on it.
            uint32_t a_out = a;
            const bool occ_policy = policy == CCX_K_POLICY_GREEDY || policy == CCX_K_POLICY_WAITING;
            if (policy == CCX_K_POLICY_RANDOM) {
                // uniform random actions drawn ON THE DEVICE (ccx_rollout's rng_seed): a counter-based
                // hash of (seed, global env, episode, step of the episode, agent slot), so the stream of an env does
                // not depend on how launches are split or on the world size
                const uint32_t asked = validbit & ~(term | trunc);
                const uint32_t r = random_action(p.rng_lo, p.rng_hi, (uint32_t)(p.env_offset + env), (uint32_t)episode,
                                                 (uint32_t)stepc, (uint32_t)i);
                a = asked ? r : 4u;
                a_out = asked ? r : (uint32_t)CCX_K_ABSENT;
            }
            if constexpr (OCC) {
                if (occ_policy) {
                    const mask_t mybit0 = mask_t(1) << rank;
                    mask_t* const o_p0 = reinterpret_cast<mask_t*>(smem + occ_base + (act ? (uint32_t)c * msz : dump_off));
                    __hip_atomic_fetch_or(o_p0, mybit0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                    wave_lds_sync();
                    const mask_t* onb = reinterpret_cast<const mask_t*>(smem + occ_base);
                    const uint32_t busy = (onb[c + 1] != 0 ? 1u : 0u) | (onb[c + Wp] != 0 ? 2u : 0u) |
                                          (onb[c - 1] != 0 ? 4u : 0u) | (onb[c - Wp] != 0 ? 8u : 0u);
                    wave_lds_sync();
                    const uint32_t cand = greedy_candidates(p, boarding, (int)((ilo >> 16) & 0xFFu), (int)(ilo >> 24));
                    uint32_t pick = greedy_pick(cand, ilo & 0xFu & ~busy);
                    const uint32_t asked = validbit & ~(term | trunc);   // policy is asked for env.agents only
                    if (policy == CCX_K_POLICY_WAITING) {
                        // waiting_policy.py:74-131: boarding agents outside the tram area wait while
                        // a live exiting agent is not on its destination row yet
                        const uint32_t pend = (boarding ? 0u : asked) & (((ilo >> 12) & 1u) ^ 1u);
                        const mask_t pend_bits = group_bits<GLOG>(__builtin_amdgcn_ballot_w64(pend != 0), lane);
                        if (boarding && !(ilo & 0x10u) && pend_bits != 0) pick = 4u;
                    }
                    a = asked ? pick : 4u;
                    a_out = asked ? pick : (uint32_t)CCX_K_ABSENT;
                }
            }

            stepc += 1;  // collectivecrossing.py:188

            // ---- 1. proposal (collectivecrossing.py:371-376, 509-534; :565-588 adds nothing):
            //         legality is a bit of the CURRENT cell's word; the target's word is fetched
            //         now and consumed after the move is decided
            const int delta = (int)(int8_t)(uint8_t)(lut >> (a * 8u));
            const int np = c + delta;
            const unsigned long long pci = cinfo[np];
            const uint32_t ok = act & (((ilo & 0xFu) >> a) & 1u);   // a == 4 (wait/absent): 0

            // ---- 2. conflict masks over move ranks: Cm = earlier ranks standing on my target,
            //         Pm = earlier ranks proposing my target, hard != 0 = cannot move whatever the
            //         earlier ranks do (no legal proposal, or a later rank still on the target)
            mask_t Cm, Pm, hard;
            if constexpr (OCC) {
                const mask_t mybit = mask_t(1) << rank;
                const uint32_t o_addr = occ_base + (act ? (uint32_t)c * msz : dump_off);
                const uint32_t q_addr = prp_base + (ok ? (uint32_t)np * msz : dump_off);
                const uint32_t r_off = (uint32_t)np * msz;
                mask_t* const o_p = reinterpret_cast<mask_t*>(smem + o_addr);
                mask_t* const q_p = reinterpret_cast<mask_t*>(smem + q_addr);
                if (!occ_policy)   // (already published for the scripted policies)
                    __hip_atomic_fetch_or(o_p, mybit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                __hip_atomic_fetch_or(q_p, mybit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                wave_lds_sync();
                const mask_t occ_t = *reinterpret_cast<const mask_t*>(smem + occ_base + r_off);
                const mask_t prp_t = *reinterpret_cast<const mask_t*>(smem + prp_base + r_off);
                wave_lds_sync();
                *o_p = 0;                      // leave the tables clean for the next step
                *q_p = 0;
                // ranks are positions in the move order: rank r's masks are compared in RANK space
                const mask_t lo_r = has_order ? low_mask<mask_t>(rank) : lo_m;
                const mask_t later_r = has_order ? (~lo_r & ~mybit) : later_m;
                Cm = occ_t & lo_r;
                Pm = prp_t & lo_r;
                hard = (occ_t & later_r) | (mask_t)(ok ^ 1u);
            } else {
                // all-pairs through the xch tile, indexed by rank: this lane plays move-rank i.
                // Entries of unused lanes hold cur 0x8000 / prop 0xFFFF and never match.
                const uint32_t curkey = act ? (uint32_t)c : 0x8000u;
                const uint32_t propkey = ok ? (uint32_t)np : 0xFFFFu;
                xch[gbase + rank] = curkey | (propkey << 16);
                wave_lds_sync();
                const uint32_t myprop = xch[lane] >> 16;
                mask_t call, pall;
                if constexpr (GLOG <= 4) {
                    // both 16-bit compares of an entry in 3 VALU ops: xor with (prop,prop), clamp
                    // each half to 0/1 ("differs"), shift into a packed accumulator
                    const uint32_t mp2 = myprop | (myprop << 16);
                    uint32_t acc = 0;
#pragma unroll
                    for (int k2 = G - 1; k2 >= 0; --k2) {
                        const uint32_t t = xch[gbase + k2] ^ mp2;
                        uint32_t ne;   // per 16-bit half: 1 if it differs, 0 if equal
                        asm("v_pk_min_u16 %0, %1, %2" : "=v"(ne) : "v"(t), "v"(0x00010001u));
                        acc = (acc << 1) | ne;
                    }
                    const uint32_t eq = ~acc;
                    call = (mask_t)(eq & 0xFFFFu) & full;
                    pall = (mask_t)(eq >> 16) & full;
                } else {
                    call = 0;
                    pall = 0;
#pragma unroll
                    for (int k2 = 0; k2 < G; ++k2) {
                        const uint32_t v = xch[gbase + k2];
                        call |= (mask_t)((v & 0xFFFFu) == myprop) << k2;
                        pall |= (mask_t)((v >> 16) == myprop) << k2;
                    }
                }
                wave_lds_sync();  // xch is rewritten next step
                Cm = call & lo_m;
                Pm = pall & lo_m;
                hard = (call & later_m) | (mask_t)((myprop + 1u) >> 16);
            }
            CCX_STAMP(0);   // loop top + proposal + conflict masks

            // ---- 3. ballot fixed point over "who moved".  Integer tests keep every ballot a
            //         single v_cmp; F(M) is constant for lanes without any earlier-rank
            //         dependency, so the loop only runs when some lane has one.
            //         OCC: lane = agent, bits = ranks.  !OCC: lane = rank, bits = ranks.
            uint64_t b = __builtin_amdgcn_ballot_w64((hard | Cm) == 0);
            const mask_t dep = hard ? mask_t(0) : (Cm | Pm);
            if (__builtin_amdgcn_ballot_w64(dep != 0) != 0) {
                if constexpr (OCC) {
                    if (has_order) {
                        // ballots are lane(=agent)-indexed here but the masks are rank-indexed:
                        // resolve through the per-rank view kept in xch (rank -> moved bit)
                        for (int it = 1; it < N; ++it) {
                            xch[gbase + rank] = (uint32_t)((b >> lane) & 1ull);
                            wave_lds_sync();
                            mask_t M = 0;
                            for (int k2 = 0; k2 < N; ++k2) M |= (mask_t)xch[gbase + k2] << k2;
                            wave_lds_sync();
                            const uint64_t b2 = __builtin_amdgcn_ballot_w64((hard | (M & Pm) | (~M & Cm)) == 0);
                            if (b2 == b) break;
                            b = b2;
                        }
                    } else {
                        for (int it = 1; it < N; ++it) {
                            const mask_t M = group_bits<GLOG>(b, lane);
                            const uint64_t b2 = __builtin_amdgcn_ballot_w64((hard | (M & Pm) | (~M & Cm)) == 0);
                            if (b2 == b) break;
                            b = b2;
                        }
                    }
                } else {
                    for (int it = 1; it < N; ++it) {
                        const mask_t M = group_bits<GLOG>(b, lane);
                        const uint64_t b2 = __builtin_amdgcn_ballot_w64((hard | (M & Pm) | (~M & Cm)) == 0);
                        if (b2 == b) break;
                        b = b2;
                    }
                }
            }
            // OCC: the ballot bit of my own lane; !OCC: the bit of the lane playing my rank
            const uint32_t moved = OCC ? (uint32_t)((b >> lane) & 1ull)
                                       : ((uint32_t)(group_bits<GLOG>(b, lane) >> rank) & 1u);
            if (moved) {  // collectivecrossing.py:408
                c = np;
                ilo = (uint32_t)pci;
                ihi = (uint32_t)(pci >> 32);
            }
            c_moves += moved;
            CCX_STAMP(1);   // ballot fixed point + position update

            // ---- 4. tail in 0/1 integer arithmetic: deactivate, terminated, truncated, flags
            //         (collectivecrossing.py:210-259)
            const uint32_t dest = (ilo >> tsh) & 1u;                 // :663-683
            const uint32_t arrive = act & dest;                      // :210-212 (act is 0 off-grid)
            act &= ~dest;
            const uint32_t live = validbit & ~(term | trunc);        // rewards.py:64, truncateds.py:56
            c_arrivals += arrive;
            c_live += live;
            const mask_t dest_bits =
                group_bits<GLOG>(__builtin_amdgcn_ballot_w64((dest | (validbit ^ 1u)) != 0), lane);
            const mask_t live_bits = group_bits<GLOG>(__builtin_amdgcn_ballot_w64(live != 0), lane);
            const uint32_t all_dest = dest_bits == full;             // terminateds.py:40-60 and :256
            const uint32_t term_out = term_all ? all_dest : dest;    // terminateds.py:40-82
            const uint32_t ge = stepc >= p.max_steps;                // truncateds.py:40-61
            const uint32_t trunc_out = live & ge;
            const uint32_t done_now = (term_out & ~term) | (trunc_out & ~trunc);  // :229-241
            term |= term_out;
            trunc |= trunc_out;
            const uint32_t emit = (done_now | ~(term | trunc)) & 1u; // :243, :763-767
            // every live agent truncates at once, so __all__ = (any live) && ge  (:257)
            uint32_t ef = all_dest | (((live_bits != 0) & ge) << 1);
            const uint32_t af = term_out | (trunc_out << 1) | (live << 2) | (emit << 3) |
                                (ilo & 0x30u) | (act << 6) | (dest << 7);
            const bool do_reset = use_pool && valid_env && (ef != 0u);
            if (do_reset) ef |= CCX_K_EF_RESET;

            // ---- 5. hand the step to the writer waves
            if constexpr (OUT) {
                stage[(s & 1) * 64 + lane] = make_uint4(ilo, ihi, af | (a_out << 8), ef);
                CCX_STAMP(2);   // tail
                lds_barrier();
                CCX_STAMP(3);   // wait for the writer waves (they may lag one step at most)
            }

            // ---- 6. auto-reset from the pool (reset() :97-150 with host-computed placements)
            if (do_reset) {
                episode += 1;
                stepc = 0;
                c_episodes += (i == 0);
                if (valid) {
                    if (pnext_pending) {   // second reset inside one action burst (rare): wait here
                        asm volatile("; rare: reset twice within one action burst");
                        pcell = (int)(pnext >> 8) * Wp + (int)(pnext & 0xFFu) + Wp + 1;
                    }
                    c = pcell;
                    const unsigned long long rci = cinfo[c];
                    ilo = (uint32_t)rci;
                    ihi = (uint32_t)(rci >> 32);
                    act = 1;
                    term = 0;
                    trunc = 0;
                    pool_idx += pool_stride;
                    if (pool_idx >= pool_size) pool_idx -= pool_size;
                    pnext = *reinterpret_cast<const uint16_t*>(pool + ((size_t)pool_idx * N + i) * 2);
                    pnext_pending = true;
                }
            }
        }
    }
    CCX_STAMP_FLUSH(ctr, 0);
    CCX_T(4);

    // ---- pace control: every tile compares its elapsed time with the schedule.
    //   on time (<= 1.5 % over)   tile 0 votes for a pace 0.4 % faster (1.6 % while no collapse has
    //                             been seen: quick descent from the conservative start value)
    //   slightly late (<= 5 %)    the memory side is at its limit: vote for pace + half the overshoot
    //   collapse (> 5 %)          the pace was beyond the cliff: vote +3 % and raise the floor to
    //                             2.5 % above this pace, so that the probing does not walk straight
    //                             back into it (the floor decays slowly at first, see above)
    // The slowest vote wins (atomicMax into the slot the next launch reads); only late tiles and
    // tile 0 touch the words, so a healthy launch costs one store.  (A healthy launch has NO late
    // tile: all of them finish within 1 % of the schedule; a collapse delays whole XCDs, 64+ tiles.)
    if (pace && p.pace_adapt && K >= 64) {
        unsigned long long now;
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(now)::"memory");
        const unsigned long long elapsed_fp = (now - pace_t0) << 8, planned_fp = (unsigned long long)K * pace;
        const bool late = elapsed_fp > planned_fp + (planned_fp >> 6);
        const bool collapse = elapsed_fp > planned_fp + planned_fp / 20u;
        if ((late || tile == 0) && lane == 0) {
            uint32_t next;                                   // a vote is for the COMMON (full-round) pace
            if (collapse) {
                next = pace_base + pace_base / 33u;
                atomicMax(&p.pace_state[2], pace_base + pace_base / 40u);
                p.pace_state[3] = 0u;
                p.pace_state[6] = pace_base;                 // (every collapsed tile writes the same value)
            } else if (late) {
                // half the relative overshoot, at least 0.5 %
                unsigned long long over16 = ((elapsed_fp - planned_fp) << 16) / planned_fp;   // overshoot x 2^16
                over16 = over16 > 65536ull ? 65536ull : over16;
                uint32_t inc = (uint32_t)(((unsigned long long)pace_base * over16) >> 17);
                const uint32_t lo = pace_base / 200u;
                next = pace_base + (inc < lo ? lo : inc);
            } else {
                next = pace_base - (pace_base >> (pace_floor ? 8 : 6));   // no collapse seen yet: descend quickly
            }
            next = next < p.pace_min_fp ? p.pace_min_fp : (next > p.pace_max_fp ? p.pace_max_fp : next);
            atomicMax(&p.pace_state[p.pace_slot ^ 1u], next);
        }
    }

#ifdef CCX_TILE_TIMES   // diagnostic (profiles/scratch/tile_times.py): elapsed 10-ns ticks of every tile
    if (pace && ctr && lane == 0) {
        unsigned long long now;
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(now)::"memory");
        ctr[kCounterTotals + (size_t)tile * kCounterSlot + 6] = now - pace_t0;
    }
#endif

    CCX_T(5);
    // ---- registers -> state ------------------------------------------------------------------
    if (valid) {
        *fx = (int)((ilo >> 16) & 0xFFu);
        *fy = (int)(ilo >> 24);
        *fact = (uint8_t)act;
        *fterm = (uint8_t)term;
        *ftrunc = (uint8_t)trunc;
    }
    if (valid_env && i == 0) {
        *fstep = stepc;
        *fepi = episode;
    }
    if (ctr) {
        const uint32_t nenv = wave_sum_u32((valid_env && i == 0) ? 1u : 0u);
        const uint32_t moves = wave_sum_u32(c_moves), arrivals = wave_sum_u32(c_arrivals);
        const uint32_t lives = wave_sum_u32(c_live), eps = wave_sum_u32(c_episodes);
        // One partial slot per tile: every wave adding to the SAME six words costs ~35 us per
        // launch when all tiles finish together (3072 same-address device-scope atomics serialise
        // at the memory side); distinct lines are free.  reduce_counters_kernel sums the slots.
        if (lane == 0 && nenv) {
            unsigned long long* slot = ctr + kCounterTotals + (size_t)tile * kCounterSlot;
            atomicAdd(&slot[0], (unsigned long long)nenv * (unsigned long long)K);
            atomicAdd(&slot[1], (unsigned long long)nenv * (unsigned long long)K * N);
            atomicAdd(&slot[2], (unsigned long long)lives);
            atomicAdd(&slot[3], (unsigned long long)eps);
            atomicAdd(&slot[4], (unsigned long long)moves);
            atomicAdd(&slot[5], (unsigned long long)arrivals);
        }
    }
    CCX_T(6);
}

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
template <int GLOG, bool PAIR, int OUT, bool OCC, bool PLAIN>
static hipError_t launch_rollout_v(const LaunchShape& ls, hipStream_t stream, const KParams& p,
                                   const KState& st, const unsigned long long* cell_info,
                                   const uint8_t* actions, const uint8_t* order, int K,
                                   int auto_reset, const uint8_t* pool, const KOut& out,
                                   unsigned long long* counters, int policy, uint8_t* actions_out) {
    if (ls.lds_bytes > 60 * 1024) {
        // big grids / many envs per tile need more than the default 64 KiB of dynamic LDS
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&rollout_kernel<GLOG, PAIR, OUT, OCC, PLAIN>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
    }
    dim3 grid(ls.num_blocks), block(64 * ls.waves_per_block * (OUT ? 1 + ls.writers : 1));
    hipLaunchKernelGGL((rollout_kernel<GLOG, PAIR, OUT, OCC, PLAIN>), grid, block, ls.lds_bytes, stream, p, st,
                       cell_info, actions, order, K, auto_reset, pool, out, counters, policy, actions_out);
    return hipGetLastError();
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
    const int outm = want_out ? (edges ? 2 : 1) : 0;
    const bool plain = order == nullptr && policy == 0;
#define CCX_GO2(P_, O_, C_)                                                                                  \
    return plain ? launch_rollout_v<GLOG, P_, O_, C_, true>(ls, stream, p, st, cell_info, actions, order, K, \
                                                            auto_reset, pool, out, counters, policy, actions_out) \
                 : launch_rollout_v<GLOG, P_, O_, C_, false>(ls, stream, p, st, cell_info, actions, order, K, \
                                                             auto_reset, pool, out, counters, policy, actions_out)
#define CCX_GO(P_, C_)                \
    switch (outm) {                   \
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
        const void* f = reinterpret_cast<const void*>(&rollout_kernel<GLOG, P_, 1, C_, true>);       \
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

// workgroups of the rollout kernel (with outputs) one CU holds at once; 0 = unknown
int rollout_blocks_per_cu(const LaunchShape& ls, int agents) {
    const bool pair = (agents % 2) == 0;
    switch (ls.glog) {
    case 0: return blocks_per_cu_g<0>(ls, pair);
    case 1: return blocks_per_cu_g<1>(ls, pair);
    case 2: return blocks_per_cu_g<2>(ls, pair);
    case 3: return blocks_per_cu_g<3>(ls, pair);
    case 4: return blocks_per_cu_g<4>(ls, pair);
    case 5: return blocks_per_cu_g<5>(ls, pair);
    default: return blocks_per_cu_g<6>(ls, pair);
    }
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
    case 0: return launch_rollout_g<0>(ls, stream, p, st, cell_info, actions, order, K, auto_reset, pool, out, counters, policy, actions_out);
    case 1: return launch_rollout_g<1>(ls, stream, p, st, cell_info, actions, order, K, auto_reset, pool, out, counters, policy, actions_out);
    case 2: return launch_rollout_g<2>(ls, stream, p, st, cell_info, actions, order, K, auto_reset, pool, out, counters, policy, actions_out);
    case 3: return launch_rollout_g<3>(ls, stream, p, st, cell_info, actions, order, K, auto_reset, pool, out, counters, policy, actions_out);
    case 4: return launch_rollout_g<4>(ls, stream, p, st, cell_info, actions, order, K, auto_reset, pool, out, counters, policy, actions_out);
    case 5: return launch_rollout_g<5>(ls, stream, p, st, cell_info, actions, order, K, auto_reset, pool, out, counters, policy, actions_out);
    case 6: return launch_rollout_g<6>(ls, stream, p, st, cell_info, actions, order, K, auto_reset, pool, out, counters, policy, actions_out);
    }
    return hipErrorInvalidValue;
}

static hipError_t launch_observe_any(const LaunchShape& ls, hipStream_t stream, const KParams& p, const KState& st,
                                     float* obs, const float4* compact, unsigned blocks) {
    switch (ls.glog) {
    case 0: return launch_observe_g<0>(ls, stream, p, st, obs, compact, blocks);
    case 1: return launch_observe_g<1>(ls, stream, p, st, obs, compact, blocks);
    case 2: return launch_observe_g<2>(ls, stream, p, st, obs, compact, blocks);
    case 3: return launch_observe_g<3>(ls, stream, p, st, obs, compact, blocks);
    case 4: return launch_observe_g<4>(ls, stream, p, st, obs, compact, blocks);
    case 5: return launch_observe_g<5>(ls, stream, p, st, obs, compact, blocks);
    case 6: return launch_observe_g<6>(ls, stream, p, st, obs, compact, blocks);
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

hipError_t launch_reset_from_pool(hipStream_t stream, const KParams& p, const KState& st,
                                  const uint8_t* env_mask, const uint8_t* pool) {
    const size_t total = (size_t)p.E * p.N;
    if (total == 0) return hipSuccess;
    const unsigned blocks = (unsigned)((total + 255) / 256);
    hipLaunchKernelGGL(reset_from_pool_kernel, dim3(blocks), dim3(256), 0, stream, p, st, env_mask, pool);
    return hipGetLastError();
}

}  // namespace ccx
