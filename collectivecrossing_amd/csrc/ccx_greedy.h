// ccx_greedy.h -- the greedy rule shared by the stand-alone policy kernel (ccx_policy.hip) and the
// policy-driven rollout (ccx_kernels.hip).  Device code only.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ccx_kernels.h"

namespace ccx {

// (primary action, preference list) of the greedy rule: returns 5 candidate actions packed 4 bits
// each, candidate 0 first; candidate 4 is always "wait"
template <typename P>   // (KParams by value, or the kernel-argument segment's copy)
__device__ __forceinline__ uint32_t greedy_candidates(const P& p, bool boarding, int cx, int cy) {
    const int div = p.div, dcx = p.dc;
    const int dest_y = boarding ? p.bdy : p.edy;
    const bool before_door = boarding ? (cy < div) : (cy > div);       // greedy_policy.py:118, :139
    const int door_level = boarding ? div - 1 : div + 1;               // :122, :143
    const uint32_t fwd = boarding ? 1u : 3u, back = boarding ? 3u : 1u;
    int dx = 0, dy = 0;
    if (before_door) {
        if (cy == door_level) {
            if (cx == dcx) dy = boarding ? 1 : -1;                     // :126-127, :148-149
            else dx = (dcx > cx) - (dcx < cx);                         // :129-130
        } else {
            dy = (door_level > cy) - (door_level < cy);                // :133-134, :155-156
        }
    } else {
        dy = (dest_y > cy) - (dest_y < cy);                            // :137, :159
    }
    const uint32_t primary = dx == 1 ? 0u : dy == 1 ? 1u : dx == -1 ? 2u : dy == -1 ? 3u : 4u;  // :202-234
    uint32_t p0, p1, p2;
    if (before_door && cx != dcx) {                                    // :334-349, :396-411
        p0 = cx < dcx ? 0u : 2u;
        p1 = fwd;
        p2 = cx < dcx ? 2u : 0u;
    } else {                                                           // :350-389, :412-449
        p0 = fwd;
        p1 = 0u;
        p2 = 2u;
    }
    return primary | (p0 << 4) | (p1 << 8) | (p2 << 12) | (back << 16) | (4u << 20);
}


// first candidate that is "wait" or whose direction bit is set in `free4` (bit a = move a is legal
// and the target cell holds no other active agent)
__device__ __forceinline__ uint32_t greedy_pick(uint32_t cand, uint32_t free4) {
    uint32_t chosen = 4u;
#pragma unroll
    for (int k = 4; k >= 0; --k) {
        const uint32_t a = (cand >> (4 * k)) & 0xFu;
        const bool ok = (a == 4u) || ((free4 >> a) & 1u);
        chosen = ok ? a : chosen;
    }
    return chosen;
}

}  // namespace ccx
