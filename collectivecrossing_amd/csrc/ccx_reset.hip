// ccx_reset.hip -- seeded initial placement on the device, bit-identical to the reference's
// reset(seed) (src/collectivecrossing/collectivecrossing.py:91-150) INCLUDING its random stream.
//
// The reference draws from gymnasium's np_random = numpy Generator(PCG64(SeedSequence(seed))):
//   boarding agent: x = integers(0, width), y = integers(0, division_y), rejected on invalid cells,
//   occupied cells and the row under the door (:101-117); exiting agent:
//   x = integers(tram_left, tram_right + 1), y = integers(division_y, height) (:130-140).
// numpy is a third-party dependency (numpy>=1.24; the stream is stable across those versions);
// its published algorithms are implemented here for the GPU:
//   SeedSequence  numpy/random/bit_generator.pyx: hashmix/mix over a pool of 4 uint32 words
//   PCG64         numpy/random/src/pcg64: 128-bit LCG (64-bit limbs here), XSL-RR 128/64 output,
//                 next_uint32 = low half, then the buffered high half of one 64-bit draw
//   integers      distributions.c: buffered_bounded_lemire_uint32 (multiply-shift + rejection);
//                 a one-value range consumes nothing
// One thread per seed: the rejection loop is data dependent and short; this kernel runs once per
// pool fill / reset, not per step.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ccx_kernels.h"

namespace ccx {

struct Pcg64 {
    uint64_t s_hi, s_lo, i_hi, i_lo;
    uint32_t buffered;
    bool has_buffered;
};

__device__ __forceinline__ uint32_t ss_hashmix(uint32_t v, uint32_t& hc) {
    v ^= hc;
    hc *= 0x931e8875u;
    v *= hc;
    v ^= v >> 16;
    return v;
}
__device__ __forceinline__ uint32_t ss_mix(uint32_t x, uint32_t y) {
    uint32_t r = 0xca01f9ddu * x - 0x4973f715u * y;
    return r ^ (r >> 16);
}

// state = state * MULT + inc  (mod 2^128), MULT = 0x2360ED051FC65DA44385DF649FCCF645
__device__ __forceinline__ void pcg_step(Pcg64& g) {
    const uint64_t MH = 2549297995355413924ull, ML = 4865540595714422341ull;
    const uint64_t lo = g.s_lo * ML;
    const uint64_t hi = __umul64hi(g.s_lo, ML) + g.s_hi * ML + g.s_lo * MH;
    const uint64_t nlo = lo + g.i_lo;
    g.s_hi = hi + g.i_hi + (nlo < lo ? 1ull : 0ull);
    g.s_lo = nlo;
}

__device__ void pcg_seed(Pcg64& g, uint64_t seed) {
    uint32_t entropy[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
    const int n_ent = (seed >> 32) ? 2 : 1;
    uint32_t pool[4], hc = 0x43b0d7e5u;
#pragma unroll
    for (int i = 0; i < 4; ++i) pool[i] = ss_hashmix(i < n_ent ? entropy[i] : 0u, hc);
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int d = 0; d < 4; ++d)
            if (s != d) pool[d] = ss_mix(pool[d], ss_hashmix(pool[s], hc));
    uint32_t w[8], hb = 0x8b51f9ddu;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        uint32_t v = pool[i & 3] ^ hb;
        hb *= 0x58f38dedu;
        v *= hb;
        w[i] = v ^ (v >> 16);
    }
    const uint64_t v0 = (uint64_t)w[0] | ((uint64_t)w[1] << 32), v1 = (uint64_t)w[2] | ((uint64_t)w[3] << 32);
    const uint64_t v2 = (uint64_t)w[4] | ((uint64_t)w[5] << 32), v3 = (uint64_t)w[6] | ((uint64_t)w[7] << 32);
    // initstate = v0:v1 (high:low), initseq = v2:v3; inc = (initseq << 1) | 1
    g.i_hi = (v2 << 1) | (v3 >> 63);
    g.i_lo = (v3 << 1) | 1ull;
    g.s_hi = 0;
    g.s_lo = 0;
    pcg_step(g);
    const uint64_t nlo = g.s_lo + v1;
    g.s_hi = g.s_hi + v0 + (nlo < g.s_lo ? 1ull : 0ull);
    g.s_lo = nlo;
    pcg_step(g);
    g.has_buffered = false;
    g.buffered = 0;
}

__device__ __forceinline__ uint64_t pcg_next64(Pcg64& g) {
    pcg_step(g);
    const uint64_t x = g.s_hi ^ g.s_lo;
    const unsigned rot = (unsigned)(g.s_hi >> 58);
    return (x >> rot) | (x << ((64u - rot) & 63u));
}

__device__ __forceinline__ uint32_t pcg_next32(Pcg64& g) {
    if (g.has_buffered) {
        g.has_buffered = false;
        return g.buffered;
    }
    const uint64_t n = pcg_next64(g);
    g.has_buffered = true;
    g.buffered = (uint32_t)(n >> 32);
    return (uint32_t)n;
}

// Generator.integers(low, high), 0 <= high - low - 1 < 2^32 - 1
__device__ __forceinline__ int pcg_integers(Pcg64& g, int low, int high) {
    const uint32_t rng = (uint32_t)(high - 1 - low);
    if (rng == 0u) return low;
    const uint32_t rng_excl = rng + 1u;
    uint64_t m = (uint64_t)pcg_next32(g) * rng_excl;
    uint32_t leftover = (uint32_t)m;
    if (leftover < rng_excl) {
        const uint32_t threshold = (0xFFFFFFFFu - rng) % rng_excl;
        while (leftover < threshold) {
            m = (uint64_t)pcg_next32(g) * rng_excl;
            leftover = (uint32_t)m;
        }
    }
    return low + (int)(m >> 32);
}

// One thread per seed.  seeds == nullptr: seed of thread t is seed0 + t.  Destination: either a pool
// (u8 [n][N][2]) or the SoA state (x, y, flags, step_count) of the envs with a non-zero mask byte.
__global__ void seeded_placement_kernel(const KParams p, const int n, const uint64_t* __restrict__ seeds,
                                        const uint64_t seed0, uint8_t* __restrict__ pool_out,
                                        const KState st, const uint8_t* __restrict__ env_mask,
                                        uint8_t* __restrict__ scratch_xy, const int max_tries,
                                        unsigned long long* fail_counter) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    if (!pool_out && env_mask && !env_mask[t]) return;
    const int N = p.N;
    // placements are written as they are accepted and re-read for the occupancy test
    uint8_t* mine = (pool_out ? pool_out : scratch_xy) + (size_t)t * N * 2;
    Pcg64 g;
    pcg_seed(g, seeds ? seeds[t] : seed0 + (uint64_t)t);
    bool failed = false;
    for (int a = 0; a < N && !failed; ++a) {
        const bool boarding = a < p.Nb;
        int x = 0, y = 0, tries = 0;
        for (;;) {
            if (++tries > max_tries) {   // the reference would spin forever here
                failed = true;
                break;
            }
            if (boarding) {
                x = pcg_integers(g, 0, p.W);
                y = pcg_integers(g, 0, p.div);
            } else {
                x = pcg_integers(g, p.tl, p.tr + 1);
                y = pcg_integers(g, p.div, p.H);
            }
            // collectivecrossing.py:509-534
            if (!(x >= 0 && x <= p.W && y >= 0 && y <= p.H)) continue;
            if (y == p.div && !(p.dl < x && x < p.dr)) continue;
            if (y >= p.div && !(p.tl < x && x < p.tr)) continue;
            bool taken = false;                                   // :536-541
            for (int b = 0; b < a; ++b) taken |= (mine[2 * b] == x) && (mine[2 * b + 1] == y);
            if (taken) continue;
            if (boarding && p.dl <= x && x <= p.dr && y == p.div - 1) continue;   // :110-117
            break;
        }
        mine[2 * a] = (uint8_t)x;
        mine[2 * a + 1] = (uint8_t)y;
    }
    if (failed) {
        atomicAdd(fail_counter, 1ull);
        return;
    }
    if (!pool_out) {   // reset() :97-98, :118-126, :141-149
        for (int a = 0; a < N; ++a) {
            const size_t idx = (size_t)t * N + a;
            st.x[idx] = mine[2 * a];
            st.y[idx] = mine[2 * a + 1];
            st.active[idx] = 1;
            st.terminated[idx] = 0;
            st.truncated[idx] = 0;
        }
        st.step_count[t] = 0;
    }
}

hipError_t launch_seeded_placement(hipStream_t stream, const KParams& p, int n, const uint64_t* seeds,
                                   uint64_t seed0, uint8_t* pool_out, const KState& st,
                                   const uint8_t* env_mask, uint8_t* scratch_xy, int max_tries,
                                   unsigned long long* fail_counter) {
    if (n <= 0) return hipSuccess;
    const unsigned blocks = (unsigned)((n + 63) / 64);
    hipLaunchKernelGGL(seeded_placement_kernel, dim3(blocks), dim3(64), 0, stream, p, n, seeds, seed0,
                       pool_out, st, env_mask, scratch_xy, max_tries, fail_counter);
    return hipGetLastError();
}

}  // namespace ccx
