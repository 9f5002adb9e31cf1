// scratch micro-benchmark: achievable pure-WRITE bandwidth with float4 stores, for comparison with
// the rollout kernel's observation stream.  hipcc --offload-arch=gfx950 -O3 write_bw.hip -o write_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float v4f __attribute__((ext_vector_type(4)));
template <bool NT>
__global__ void fill(v4f* dst, size_t n4, int chunk4) {
    // each wave writes `chunk4` consecutive float4 per step-like iteration, waves interleaved
    const size_t wave = (size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const size_t nwaves = (size_t)gridDim.x * (blockDim.x >> 6);
    const int lane = threadIdx.x & 63;
    v4f v = {1.f, 2.f, 3.f, (float)lane};
    for (size_t base = wave * chunk4; base < n4; base += nwaves * chunk4)
        for (int q = lane; q < chunk4 && base + q < n4; q += 64) {
            if (NT) __builtin_nontemporal_store(v, &dst[base + q]);
            else dst[base + q] = v;
        }
}
int main() {
    const size_t bytes = 1327104000ull / 16 * 16;
    v4f* d;
    hipMalloc(&d, bytes);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    for (int nt = 0; nt < 2; ++nt)
        for (int waves : {512, 1024, 2048, 4096, 16384})
            for (int chunk4 : {64, 608, 2144}) {
                float best = 1e9;
                for (int r = 0; r < 5; ++r) {
                    hipEventRecord(a);
                    if (nt) hipLaunchKernelGGL(fill<true>, dim3(waves), dim3(64), 0, 0, d, bytes / 16, chunk4);
                    else hipLaunchKernelGGL(fill<false>, dim3(waves), dim3(64), 0, 0, d, bytes / 16, chunk4);
                    hipEventRecord(b);
                    hipEventSynchronize(b);
                    float ms;
                    hipEventElapsedTime(&ms, a, b);
                    if (ms < best) best = ms;
                }
                printf("nt=%d waves=%5d chunk=%5d B  %.3f ms  %.2f TB/s\n", nt, waves, chunk4 * 16, best,
                       bytes / (best * 1e-3) / 1e12);
            }
    return 0;
}
