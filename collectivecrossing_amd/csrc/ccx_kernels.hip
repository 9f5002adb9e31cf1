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
#include <stddef.h>
#include <type_traits>
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
// The same stores addressed as (wave-uniform base in SGPRs) + (32-bit lane offset) + (immediate < 4096): no 64-bit vector
// add per store, and the ten per-iteration lane offsets of a writer collapse into three registers.
template <int IMM>
__device__ __forceinline__ void store_obs_at(v4f v, const char* base, uint32_t voff) {
    static_assert(IMM >= 0 && IMM < 4096, "global_store immediate offset");
#if defined(CCX_PLAIN_STORES) || defined(CCX_BUILTIN_NT_STORES)
    store_obs(v, reinterpret_cast<v4f*>(const_cast<char*>(base) + voff + IMM));
#else
    asm volatile("global_store_dwordx4 %0, %1, %2 offset:%3 " CCX_STORE_BITS "\n\ts_nop 1" ::"v"(voff), "v"(v), "s"(base), "n"(IMM) : "memory");
#endif
}
template <int IMM>
__device__ __forceinline__ void store_obs_at(v2f w, const char* base, uint32_t voff) {
    static_assert(IMM >= 0 && IMM < 4096, "global_store immediate offset");
#if defined(CCX_PLAIN_STORES) || defined(CCX_BUILTIN_NT_STORES)
    *reinterpret_cast<v2f*>(const_cast<char*>(base) + voff + IMM) = w;
#else
    asm volatile("global_store_dwordx2 %0, %1, %2 offset:%3 " CCX_STORE_BITS ::"v"(voff), "v"(w), "s"(base), "n"(IMM) : "memory");
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
// KParams): [xch u32 x 64][hand-off words u32 x 16][stage uint4 x 8 x 64][WSlot x writers][{occ, prp} masks x (cells+1) x EW].
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

template <typename P>   // (KParams by value, or the kernel-argument segment's copy)
__device__ __forceinline__ void init_wave_consts(WaveLds* wl, const P& p, int lane) {
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
//       bit4 = 0 always (the "legality bit" of action 4 = wait: `(lo >> a) & 1` needs no clamp)
//       bit5 IN_TRAM_AREA (:551-554)  bit6 AT_DOOR (:556-563)       -- CCX_AF_* bits 4/5, shifted up by one
//       bit8  boarding: on destination row (:663-683)   bits 9-10  boarding reward class
//       bit12 exiting:  on destination row              bits 13-14 exiting reward class
//       byte2 = x, byte3 = y  (0 for border cells)
//   hi: int16 signed distance term of the boarding reward | int16 of the exiting reward << 16
// reward class (rewards.py:44-182): 0 = (double)sd * distance_penalty_factor, 1/2/3 = constants
// rA/rB/rC chosen per reward mode.  The word of the agent's CURRENT cell is carried in registers,
// so the legality of a move is a bit test; the word of the proposed cell is fetched off the
// critical path and only consumed once the move is known to happen.
//
// sim -> writer hand-off (uint4 per lane and step): x = cell lo, y = cell hi, z = the CCX_AF_* bits the writer
// cannot derive from the cell word (terminated, truncated, live, obs, active) | chosen action << 8, w = the CCX_EF_*
// byte of its env.

// ---------------------------------------------------------------------------------------------
// the fused rollout / step kernel.
//   OUT  trajectory outputs requested: a tile is served by 1 sim wave + p.writers writer waves;
//        OUT = false: sim waves only (counters only).
//   OCC  conflict masks come from per-env occupancy / proposal bit tables in LDS (O(1) per agent);
//        OCC = false: all-pairs compare through the xch tile (grids whose tables exceed LDS).
// wave index in block -> role = wib / tiles_per_block (0 = sim, 1.. = writer), tile = wib % tpb.
// ---------------------------------------------------------------------------------------------
#ifdef CCX_TSTAMPS   // diagnostic (profiles/scratch/tstamps.py): raw s_memrealtime (10-ns ticks) at fixed points of tile 0's waves
#define CCX_T(q) do { if (rollout_kernarg_tail().counters && blockIdx.x == 0 && (threadIdx.x & 63) == 0 && \
                          ((q) >= 7 || threadIdx.x == 0)) { unsigned long long t_; \
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory"); \
    rollout_kernarg_tail().counters[8 + (q)] = t_; } } while (0)
#else
#define CCX_T(q) do { } while (0)
#endif

// The tail of the rollout kernels' argument list as it lies in the kernel-argument segment (AMDGPU ABI: by-value
// arguments in declaration order, each at its natural alignment): the epilogue re-reads the state / counter pointers
// from there instead of carrying them through the step loop.  tests/test_kernel_resources.py checks these offsets
// against the `.args` metadata of the built code objects.
struct KernargTail {
    KState st;                                   // argument 1
    const unsigned long long* cell_info;         // 2
    const uint8_t* actions;                      // 3
    const uint8_t* order;                        // 4
    int K, auto_reset;                           // 5, 6
    const uint8_t* pool;                         // 7
    KOut out;                                    // 8
    unsigned long long* counters;                // 9
    int policy;                                  // 10
    uint8_t* actions_out;                        // 11
};
static_assert(sizeof(KParams) % 8 == 0 && alignof(KParams) == 8 && alignof(KState) == 8, "kernarg layout");
static_assert(offsetof(KernargTail, counters) == sizeof(KState) + 3 * 8 + 8 + 8 + sizeof(KOut), "kernarg layout");
typedef __attribute__((address_space(4))) const KernargTail KernargTailC;     // (the constant address space: scalar loads)
// The kernel parameters themselves are read the same way: `p` in the kernel body is a reference into the kernel-argument
// segment, not the by-value argument.  62 dwords of KParams preloaded into SGPRs at kernel entry and held until their
// last use were most of the 60-150 SGPR spills of the rollout kernels; a field is now a scalar load next to its use (or
// hoisted in front of the loop that needs it), and the register allocator re-loads instead of spilling.
typedef __attribute__((address_space(4))) const KParams KParamsC;
__device__ __forceinline__ KParamsC& rollout_kernarg_params() {
#if defined(__HIP_DEVICE_COMPILE__)
    return *(KParamsC*)__builtin_amdgcn_kernarg_segment_ptr();
#else
    static KParams host_dummy{};
    return *(KParamsC*)(uintptr_t)&host_dummy;
#endif
}
// Reading fields where they are used makes the kernel entry a CHAIN of scalar-cache misses (seven 64-byte lines, each
// first touch a memory round trip of its own: a single-step launch spent ~1.6 us before its first global load).  One
// dword of every line is requested at entry, all at once, and waited for once; the values are never used, the lines are in
// the scalar cache for the loads that follow.
__device__ __forceinline__ void rollout_kernarg_prefetch() {
#if defined(__HIP_DEVICE_COMPILE__)
    static_assert(sizeof(KParams) + sizeof(KernargTail) <= 7 * 64, "one s_load per 64-byte line of the explicit arguments");
    uint32_t d0, d1, d2, d3, d4, d5, d6;
    asm volatile("s_load_dword %0, %7, 0x0\n\ts_load_dword %1, %7, 0x40\n\ts_load_dword %2, %7, 0x80\n\t"
                 "s_load_dword %3, %7, 0xc0\n\ts_load_dword %4, %7, 0x100\n\ts_load_dword %5, %7, 0x140\n\t"
                 "s_load_dword %6, %7, 0x180\n\ts_waitcnt lgkmcnt(0)"   // (the compiler must not reuse a destination in flight)
                 : "=&s"(d0), "=&s"(d1), "=&s"(d2), "=&s"(d3), "=&s"(d4), "=&s"(d5), "=&s"(d6)
                 : "s"(__builtin_amdgcn_kernarg_segment_ptr()));
#endif
}
__device__ __forceinline__ KernargTailC& rollout_kernarg_tail() {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef __attribute__((address_space(4))) const char kchar;
    kchar* base = (kchar*)__builtin_amdgcn_kernarg_segment_ptr();
    return *(KernargTailC*)(base + sizeof(KParams));
#else
    static KernargTail host_dummy{};   // (device-only; the host pass just needs the declaration)
    return *(KernargTailC*)(uintptr_t)&host_dummy;
#endif
}

// template parameters of the kernel (its body: ccx_rollout_body.inc):
//   PLAIN  the caller passed neither a move order nor a policy (the bench line, plain RL stepping): the step
//        loop is compiled without those branches (12 % fewer cycles per env-step on the sim chain).
//   OUTM 0 = no trajectory outputs, 1 = outputs, 2 = outputs whose tile regions do not begin / end on 128-byte
//        lines (edge iterations; a separate instantiation because the single-writer C2 path loses 4-5 %
//        to ANY extra instruction in its store loop, even a never-taken branch)
//
// The kernel proper.  A CU must hold 16 wavefronts of it (4 per SIMD: the launch shapes count on that), i.e. at most
// 128 VGPRs.  The compiler stays below that by itself for the plain instantiations (115-118); those with the policy /
// move-order branches drift to 129-141 (12 wavefronts per CU: C5 fell from 0.87 to 0.49 of the peak when the random
// policy was added) and carry amdgpu_waves_per_eu(4) (127 VGPRs; the smaller lane groups pay 12 bytes of scratch per
// lane for it).  The attribute on the plain instantiations as well, or a lower bound on all of them, only made the
// allocator's choices worse (scratch there, or 137-153 VGPRs).
template <int GLOG, bool PAIR, int OUTM, bool OCC, bool PLAIN>
__global__ void __launch_bounds__(512)
rollout_kernel(const KParams p_by_value, const KState st_by_value, const unsigned long long* __restrict__ cell_info_by_value,
               const uint8_t* __restrict__ actions_by_value, const uint8_t* __restrict__ order_by_value, const int K_by_value,
               const int auto_reset_by_value, const uint8_t* __restrict__ pool_by_value, const KOut out_by_value,
               unsigned long long* counters_by_value, const int policy_by_value, uint8_t* __restrict__ actions_out_by_value) {
#include "ccx_rollout_body.inc"
}
template <int GLOG, bool PAIR, int OUTM, bool OCC, bool PLAIN>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4)))
rollout_kernel_v128(const KParams p_by_value, const KState st_by_value, const unsigned long long* __restrict__ cell_info_by_value,
                    const uint8_t* __restrict__ actions_by_value, const uint8_t* __restrict__ order_by_value, const int K_by_value,
                    const int auto_reset_by_value, const uint8_t* __restrict__ pool_by_value, const KOut out_by_value,
                    unsigned long long* counters_by_value, const int policy_by_value, uint8_t* __restrict__ actions_out_by_value) {
#include "ccx_rollout_body.inc"
}
template <int GLOG, bool PAIR, int OUTM, bool OCC, bool PLAIN>
static const void* rollout_entry() {
    if constexpr (!PLAIN)
        return reinterpret_cast<const void*>(&rollout_kernel_v128<GLOG, PAIR, OUTM, OCC, PLAIN>);
    else
        return reinterpret_cast<const void*>(&rollout_kernel<GLOG, PAIR, OUTM, OCC, PLAIN>);
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

// workgroups of the rollout kernel (with outputs) one CU holds at once; 0 = unknown
int rollout_blocks_per_cu(const LaunchShape& ls, int agents) {
    const bool pair = (agents % 2) == 0;
    switch (ls.glog) {
#if !defined(CCX_ONLY_GLOG) || CCX_ONLY_GLOG == 0
    case 0: return blocks_per_cu_g<0>(ls, pair);
#endif
#if !defined(CCX_ONLY_GLOG) || CCX_ONLY_GLOG == 1
    case 1: return blocks_per_cu_g<1>(ls, pair);
#endif
#if !defined(CCX_ONLY_GLOG) || CCX_ONLY_GLOG == 2
    case 2: return blocks_per_cu_g<2>(ls, pair);
#endif
#if !defined(CCX_ONLY_GLOG) || CCX_ONLY_GLOG == 3
    case 3: return blocks_per_cu_g<3>(ls, pair);
#endif
#if !defined(CCX_ONLY_GLOG) || CCX_ONLY_GLOG == 4
    case 4: return blocks_per_cu_g<4>(ls, pair);
#endif
#if !defined(CCX_ONLY_GLOG) || CCX_ONLY_GLOG == 5
    case 5: return blocks_per_cu_g<5>(ls, pair);
#endif
#if !defined(CCX_ONLY_GLOG) || CCX_ONLY_GLOG == 6
    default: return blocks_per_cu_g<6>(ls, pair);
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
    case 0: return launch_rollout_g<0>(ls, stream, p, st, cell_info, actions, order, K, auto_reset, pool, out, counters, policy, actions_out);
#endif
#if !defined(CCX_ONLY_GLOG) || CCX_ONLY_GLOG == 1
    case 1: return launch_rollout_g<1>(ls, stream, p, st, cell_info, actions, order, K, auto_reset, pool, out, counters, policy, actions_out);
#endif
#if !defined(CCX_ONLY_GLOG) || CCX_ONLY_GLOG == 2
    case 2: return launch_rollout_g<2>(ls, stream, p, st, cell_info, actions, order, K, auto_reset, pool, out, counters, policy, actions_out);
#endif
#if !defined(CCX_ONLY_GLOG) || CCX_ONLY_GLOG == 3
    case 3: return launch_rollout_g<3>(ls, stream, p, st, cell_info, actions, order, K, auto_reset, pool, out, counters, policy, actions_out);
#endif
#if !defined(CCX_ONLY_GLOG) || CCX_ONLY_GLOG == 4
    case 4: return launch_rollout_g<4>(ls, stream, p, st, cell_info, actions, order, K, auto_reset, pool, out, counters, policy, actions_out);
#endif
#if !defined(CCX_ONLY_GLOG) || CCX_ONLY_GLOG == 5
    case 5: return launch_rollout_g<5>(ls, stream, p, st, cell_info, actions, order, K, auto_reset, pool, out, counters, policy, actions_out);
#endif
#if !defined(CCX_ONLY_GLOG) || CCX_ONLY_GLOG == 6
    case 6: return launch_rollout_g<6>(ls, stream, p, st, cell_info, actions, order, K, auto_reset, pool, out, counters, policy, actions_out);
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
