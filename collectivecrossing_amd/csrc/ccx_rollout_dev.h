// ccx_rollout_dev.h -- device-side helpers and the rollout kernel templates of libccx (gfx950), shared by
// ccx_kernels.hip (observe / reset / reduction kernels, dispatch) and ccx_rollout_g.hip (the rollout kernel
// instantiations of ONE lane-group size per translation unit, so that the 168 instantiations compile in parallel).
#pragma once
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
//        to ANY extra instruction in its store loop, even a never-taken branch), 3 = the same when the slab STRIDE is not a
//        multiple of 128 bytes either (batch sizes off rows_alignment(): the region's place in its line moves from step to step
//        and the layout is worked out per step; its own instantiation so that 2 -- C5-50's bench kernel -- stays as it was)
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

}  // namespace ccx
