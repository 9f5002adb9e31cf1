// launch_floor.hip -- what does a dependent kernel boundary cost on this box, and what do the pieces of a one-step launch
// cost on top of it?  (round 4: the floor of ccx_step.)  Every variant is a chain of 200 launches of ONE kernel on one
// stream, captured into a HIP graph and replayed 20 times; printed: us per launch.
//   null          empty kernel
//   ld_st         one dword load -> store per lane on a 128 KB array (one dependent round trip through memory)
//   ld_ld_st      load -> dependent load (L2-resident 1 KB table) -> store (two round trips)
//   lds_table     load state + table into LDS, barrier-free, ds_read, store (the step kernel's skeleton)
//   rows          the above + B bytes of streaming stores per launch (the observation rows of a step)
// build: hipcc -O3 --offload-arch=gfx950 [-mllvm -amdgpu-kernarg-preload-count=14] launch_floor.hip -o launch_floor
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

typedef float v4f __attribute__((ext_vector_type(4)));

__global__ void k_null() {}

__global__ void k_ld_st(const int* __restrict__ a, int* __restrict__ b, int n) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) b[t] = a[t] + 1;
}

__global__ void k_ld_ld_st(const int* __restrict__ a, const int* __restrict__ tab, int* __restrict__ b, int n) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) b[t] = tab[a[t] & 255] + 1;
}

// seven state-like loads + a 143-entry table staged in LDS by every wave, then an LDS lookup, a store per lane
__global__ void k_lds_table(const int* __restrict__ x, const int* __restrict__ y, const unsigned char* __restrict__ f0,
                            const unsigned char* __restrict__ f1, const unsigned char* __restrict__ f2,
                            const unsigned char* __restrict__ act, const unsigned long long* __restrict__ tab,
                            int* __restrict__ ox, int n) {
    __shared__ unsigned long long lt[4][192];
    const int t = blockIdx.x * blockDim.x + threadIdx.x, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int tc = t < n ? t : 0;
    const int vx = x[tc], vy = y[tc];
    const unsigned v0 = f0[tc], v1 = f1[tc], v2 = f2[tc], va = act[tc];
    const unsigned long long t0 = tab[lane], t1 = tab[lane + 64], t2 = tab[(lane + 128) < 143 ? lane + 128 : 0];
    lt[w][lane] = t0; lt[w][lane + 64] = t1; lt[w][lane + 128] = t2;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const unsigned long long c = lt[w][(vx + vy * 15 + va) % 143];
    if (t < n) ox[t] = (int)(c >> 16) + (int)(v0 + v1 + v2);
}

// the same + rows: every wave streams `its` x 1 KiB of `sc1 nt` stores
__global__ void k_rows(const int* __restrict__ x, const int* __restrict__ y, const unsigned char* __restrict__ f0,
                       const unsigned char* __restrict__ f1, const unsigned char* __restrict__ f2,
                       const unsigned char* __restrict__ act, const unsigned long long* __restrict__ tab,
                       int* __restrict__ ox, int n, char* __restrict__ rows, int its) {
    __shared__ unsigned long long lt[4][192];
    const int t = blockIdx.x * blockDim.x + threadIdx.x, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int tc = t < n ? t : 0;
    const int vx = x[tc], vy = y[tc];
    const unsigned v0 = f0[tc], v1 = f1[tc], v2 = f2[tc], va = act[tc];
    const unsigned long long t0 = tab[lane], t1 = tab[lane + 64], t2 = tab[(lane + 128) < 143 ? lane + 128 : 0];
    lt[w][lane] = t0; lt[w][lane + 64] = t1; lt[w][lane + 128] = t2;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const unsigned long long c = lt[w][(vx + vy * 15 + va) % 143];
    const float f = (float)(c >> 16);
    const v4f v = {f, f + 1.0f, (float)v0, (float)(v1 + v2)};
    const size_t wave = (size_t)(t >> 6);
    char* dst = rows + wave * (size_t)its * 1024u + (size_t)lane * 16u;
    for (int i = 0; i < its; ++i)
        asm volatile("global_store_dwordx4 %0, %1, off sc1 nt\n\ts_nop 1" ::"v"(dst + (size_t)i * 1024u), "v"(v) : "memory");
    if (t < n) ox[t] = (int)(c >> 16) + (int)(v0 + v1 + v2);
}

template <typename F>
static double per_launch_us(hipStream_t s, int chain, int reps, F launch) {
    launch();                                   // warm (code object load)
    CK(hipStreamSynchronize(s));
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
    for (int i = 0; i < chain; ++i) launch();
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CK(hipGraphLaunch(ge, s));
    CK(hipStreamSynchronize(s));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    double best = 1e30;
    for (int pass = 0; pass < 3; ++pass) {
        CK(hipEventRecord(e0, s));
        for (int r = 0; r < reps; ++r) CK(hipGraphLaunch(ge, s));
        CK(hipEventRecord(e1, s));
        CK(hipEventSynchronize(e1));
        float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
        const double us = ms * 1e3 / ((double)chain * reps);
        if (us < best) best = us;
    }
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
    return best;
}

int main() {
    hipStream_t s; CK(hipStreamCreate(&s));
    const int n = 4096 * 8;
    int *a, *b, *tab; unsigned char *f0, *f1, *f2, *act; unsigned long long* ctab; char* rows;
    CK(hipMalloc(&a, n * 4)); CK(hipMalloc(&b, n * 4)); CK(hipMalloc(&tab, 1024));
    CK(hipMalloc(&f0, n)); CK(hipMalloc(&f1, n)); CK(hipMalloc(&f2, n)); CK(hipMalloc(&act, n));
    CK(hipMalloc(&ctab, 192 * 8)); CK(hipMalloc(&rows, (size_t)96 << 20));
    CK(hipMemset(a, 0, n * 4)); CK(hipMemset(b, 0, n * 4)); CK(hipMemset(tab, 0, 1024));
    CK(hipMemset(f0, 0, n)); CK(hipMemset(f1, 0, n)); CK(hipMemset(f2, 0, n)); CK(hipMemset(act, 1, n)); CK(hipMemset(ctab, 0, 192 * 8));
    const int chain = 200, reps = 20;
    struct { int grid, block; } shapes[] = {{256, 64}, {512, 64}, {1024, 64}, {2048, 64}, {128, 256}, {256, 256}, {512, 128}, {256, 384}, {256, 128}};
    for (auto sh : shapes)
        printf("null        grid %4d x %3d : %.2f us\n", sh.grid, sh.block,
               per_launch_us(s, chain, reps, [&] { hipLaunchKernelGGL(k_null, dim3(sh.grid), dim3(sh.block), 0, s); }));
    struct { int grid, block; } cover[] = {{512, 64}, {256, 128}, {128, 256}, {1024, 64}};   // (n = 32768 lanes; the last: half-empty)
    for (auto sh : cover) {
        const int nn = sh.grid * sh.block >= n ? n : sh.grid * sh.block;
        printf("ld_st       grid %4d x %3d : %.2f us\n", sh.grid, sh.block,
               per_launch_us(s, chain, reps, [&] { hipLaunchKernelGGL(k_ld_st, dim3(sh.grid), dim3(sh.block), 0, s, a, b, nn); }));
        // a chain that READS what the previous launch WROTE (the state write-back -> next step's state load)
        printf("ld_st RAW   grid %4d x %3d : %.2f us\n", sh.grid, sh.block,
               per_launch_us(s, chain, reps, [&] { hipLaunchKernelGGL(k_ld_st, dim3(sh.grid), dim3(sh.block), 0, s, a, a, nn); }));
        printf("ld_ld_st    grid %4d x %3d : %.2f us\n", sh.grid, sh.block,
               per_launch_us(s, chain, reps, [&] { hipLaunchKernelGGL(k_ld_ld_st, dim3(sh.grid), dim3(sh.block), 0, s, a, tab, b, nn); }));
        printf("lds_table   grid %4d x %3d : %.2f us\n", sh.grid, sh.block,
               per_launch_us(s, chain, reps, [&] { hipLaunchKernelGGL(k_lds_table, dim3(sh.grid), dim3(sh.block), 0, s, a, b, f0, f1, f2, act, ctab, a, nn); }));
        const int waves = sh.grid * sh.block / 64;
        for (int total_kb : {0, 1216, 5189, 20756}) {   // (5189 KB = one C2 step's rows)
            const int its = (total_kb + waves - 1) / waves;
            printf("rows %5d KB (%3d x 1 KiB per wave) grid %4d x %3d : %.2f us\n", its * waves, its, sh.grid, sh.block,
                   per_launch_us(s, chain, reps, [&] { hipLaunchKernelGGL(k_rows, dim3(sh.grid), dim3(sh.block), 0, s, a, b, f0, f1, f2, act, ctab, a, nn, rows, its); }));
        }
    }
    return 0;
}
