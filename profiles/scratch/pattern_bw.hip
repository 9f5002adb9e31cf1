// scratch: the rollout kernel's observation WRITE PATTERN without any of its work.
// tiles x writers waves; per step each tile writes its `tile_bytes` region of the step slab
// (interleaved 1 KiB chunks between the tile's writers), then moves one slab forward.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v4f __attribute__((ext_vector_type(4)));
__global__ void pattern(v4f* dst, int steps, int tiles, int writers, int tile_v4, size_t slab_v4, int sleep) {
    const int tile = blockIdx.x, w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    v4f v = {1.f, 2.f, 3.f, (float)lane};
    v4f* p = dst + (size_t)tile * tile_v4;
    for (int s = 0; s < steps; ++s) {
        for (int q = lane + 64 * w; q < tile_v4; q += 64 * writers) __builtin_nontemporal_store(v, &p[q]);
        p += slab_v4;
        for (int k = 0; k < sleep; ++k) __builtin_amdgcn_s_sleep(8);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
}
int main() {
    const int tiles = 512, steps = 500, tile_v4 = 9728 / 16;
    const size_t slab_v4 = (size_t)tiles * tile_v4;
    v4f* d;
    hipMalloc(&d, slab_v4 * 16 * steps);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    for (int writers : {1, 2, 4})
        for (int sleep : {0, 2, 4, 8}) {
            float best = 1e9;
            for (int r = 0; r < 4; ++r) {
                hipEventRecord(a);
                hipLaunchKernelGGL(pattern, dim3(tiles), dim3(64 * writers), 0, 0, d, steps, tiles, writers, tile_v4, slab_v4, sleep);
                hipEventRecord(b);
                hipEventSynchronize(b);
                float ms;
                hipEventElapsedTime(&ms, a, b);
                if (ms < best) best = ms;
            }
            printf("writers=%d sleep=%d  %.3f ms  %.3f us/step  %.2f TB/s\n", writers, sleep, best, best * 1e3 / steps,
                   slab_v4 * 16.0 * steps / (best * 1e-3) / 1e12);
        }
    return 0;
}
