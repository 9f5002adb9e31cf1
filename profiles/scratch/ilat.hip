// ilat.hip -- instruction latency / issue microbenchmark for gfx950 (one wave per SIMD, like the sim wave of
// ccx::rollout_kernel).  Every case is REP back-to-back copies of a short instruction sequence between two
// s_memtime reads; prints shader clocks per copy.   hipcc --offload-arch=gfx950 -O2 ilat.hip -o ilat && ./ilat
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <string>

#define REP 64
#define STR2(x) #x
#define STR(x) STR2(x)

#define T0 asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory")
#define T1 asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory")

__global__ void __launch_bounds__(256) k(unsigned long long* out, uint32_t seed, int which) {
    extern __shared__ uint32_t lds[];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = (i * 4 + 64) & 0x3FFC;   // pointer-chase table
    __syncthreads();
    unsigned long long t0 = 0, t1 = 0;
    uint32_t x = seed + lane, y = seed * 3 + 1, z = lane * 4;
    unsigned long long q = ((unsigned long long)x << 32) | y;
    uint32_t a0 = x, a1 = y, a2 = z, a3 = x ^ y;
    switch (which) {
    case 0:   // dependent v_add_u32
        T0; asm volatile(".rept " STR(REP) "\n v_add_u32 %0, %0, %1\n .endr" : "+v"(x) : "v"(y)); T1; break;
    case 1:   // 4 independent v_add_u32 chains (issue rate)
        T0; asm volatile(".rept " STR(REP) "\n v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4\n .endr"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(y)); T1; x = a0 + a1 + a2 + a3; break;
    case 2:   // dependent v_lshrrev_b64
        T0; asm volatile(".rept " STR(REP) "\n v_lshrrev_b64 %0, 1, %0\n .endr" : "+v"(q)); T1; x = (uint32_t)q; break;
    case 3:   // dependent v_bitop3 / v_and_or
        T0; asm volatile(".rept " STR(REP) "\n v_and_or_b32 %0, %0, %1, %2\n .endr" : "+v"(x) : "v"(y), "v"(z)); T1; break;
    case 4:   // v_cmp -> vcc -> v_cndmask (VALU -> VCC -> VALU)
        T0; asm volatile(".rept " STR(REP) "\n v_cmp_ne_u32 vcc, 0, %0\n v_cndmask_b32 %0, %1, %0, vcc\n .endr" : "+v"(x) : "v"(y) : "vcc"); T1; break;
    case 5:   // v_cmp -> sgpr pair -> s_and -> v_cndmask (VALU -> SALU -> VALU)
        T0; asm volatile(".rept " STR(REP) "\n v_cmp_ne_u32 s[20:21], 0, %0\n s_and_b64 s[22:23], s[20:21], exec\n v_cndmask_b32 %0, %1, %0, s[22:23]\n .endr"
                         : "+v"(x) : "v"(y) : "s20", "s21", "s22", "s23"); T1; break;
    case 6:   // v_cmp -> vcc -> s_cbranch_vccz not taken
        T0; asm volatile(".rept " STR(REP) "\n v_cmp_ne_u32 vcc, 0x7fffffff, %0\n s_cbranch_vccz 1f\n v_add_u32 %0, %0, %1\n1:\n .endr" : "+v"(x) : "v"(y) : "vcc"); T1; break;
    case 7:   // v_cmp -> vcc -> s_cbranch_vccnz taken (skips one instruction)
        T0; asm volatile(".rept " STR(REP) "\n v_cmp_ne_u32 vcc, 0x7fffffff, %0\n s_cbranch_vccnz 1f\n v_add_u32 %0, %0, %1\n1:\n v_add_u32 %0, %0, %1\n .endr" : "+v"(x) : "v"(y) : "vcc"); T1; break;
    case 8:   // ballot style: v_cmp -> sgpr -> v_lshrrev_b64 by lane -> v_and (group_bits)
        T0; asm volatile(".rept " STR(REP) "\n v_cmp_ne_u32 s[20:21], 0, %0\n v_lshrrev_b64 v[40:41], %1, s[20:21]\n v_and_b32 %0, 0xff, v40\n v_or_b32 %0, 1, %0\n .endr"
                         : "+v"(x) : "v"(z) : "s20", "s21", "v40", "v41"); T1; break;
    case 9:   // v_cmp -> sgpr -> s_cmp_eq_u64 -> s_cbranch_scc (SALU reads a VALU-written SGPR, branch not taken)
        T0; asm volatile(".rept " STR(REP) "\n v_cmp_ne_u32 s[20:21], 0, %0\n s_cmp_eq_u64 s[20:21], 0\n s_cbranch_scc1 1f\n v_add_u32 %0, %0, %1\n1:\n .endr"
                         : "+v"(x) : "v"(y) : "s20", "s21", "scc"); T1; break;
    case 10:  // ds_read_b32 pointer chase (LDS latency)
        T0; asm volatile(".rept " STR(REP) "\n ds_read_b32 %0, %0\n s_waitcnt lgkmcnt(0)\n .endr" : "+v"(z)); T1; x = z; break;
    case 11:  // ds_read_b64 pointer chase
        T0; asm volatile(".rept " STR(REP) "\n ds_read_b64 v[40:41], %0\n s_waitcnt lgkmcnt(0)\n v_mov_b32 %0, v40\n .endr" : "+v"(z) :: "v40", "v41"); T1; x = z; break;
    case 12:  // ds_or_b32 x2 + ds_read_b32 x2 + wait (the occupancy-table round trip), distinct addresses per lane
        T0; asm volatile(".rept " STR(REP) "\n ds_or_b32 %1, %2\n ds_or_b32 %1, %2 offset:8192\n ds_read_b32 v40, %1\n ds_read_b32 v41, %1 offset:8192\n s_waitcnt lgkmcnt(0)\n v_and_b32 %0, v40, v41\n .endr"
                         : "+v"(x) : "v"(z), "v"(y) : "v40", "v41", "memory"); T1; break;
    case 13:  // ds_write_b128 + lgkmcnt(0) (the hand-off write and its acknowledgement)
        T0; asm volatile(".rept " STR(REP) "\n ds_write_b128 %0, v[40:43]\n s_waitcnt lgkmcnt(0)\n .endr" :: "v"(z * 4) : "v40", "v41", "v42", "v43", "memory"); T1; break;
    case 14:  // s_barrier, all waves of the block in lock-step
        T0; asm volatile(".rept " STR(REP) "\n s_barrier\n .endr" ::: "memory"); T1; break;
    case 15:  // v_readfirstlane -> s_add -> v_add (VALU -> SGPR -> SALU -> VALU)
        T0; asm volatile(".rept " STR(REP) "\n v_readfirstlane_b32 s20, %0\n s_add_u32 s20, s20, 1\n v_add_u32 %0, s20, %0\n .endr" : "+v"(x) :: "s20", "scc"); T1; break;
    case 16:  // DPP dependent (quad_perm) v_add
        T0; asm volatile(".rept " STR(REP) "\n s_nop 1\n v_add_u32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n .endr" : "+v"(x)); T1; break;
    case 17:  // ds_bpermute_b32 dependent
        T0; asm volatile(".rept " STR(REP) "\n ds_bpermute_b32 %0, %1, %0\n s_waitcnt lgkmcnt(0)\n .endr" : "+v"(x) : "v"(z)); T1; break;
    case 18:  // s_memrealtime + wait
        T0; asm volatile(".rept " STR(REP) "\n s_memrealtime s[20:21]\n s_waitcnt lgkmcnt(0)\n .endr" ::: "s20", "s21"); T1; break;
    case 19:  // v_perm_b32 dependent
        T0; asm volatile(".rept " STR(REP) "\n v_perm_b32 %0, %0, %1, %2\n .endr" : "+v"(x) : "v"(y), "v"(z)); T1; break;
    case 20:  // ds_write_b128 then ds_write_b32 (payload + flag), no wait: issue cost only
        T0; asm volatile(".rept " STR(REP) "\n ds_write_b128 %0, v[40:43]\n ds_write_b32 %0, v40 offset:16384\n .endr" :: "v"(z * 4) : "v40", "v41", "v42", "v43", "memory"); T1; break;
    case 21:  // global_store_dwordx4 issue cost (64 lanes x 16 B, nt), no wait
        T0; asm volatile(".rept " STR(REP) "\n global_store_dwordx4 %0, v[40:43], off nt\n .endr" :: "v"(out + 4096 + lane * 2) : "v40", "v41", "v42", "v43", "memory"); T1; break;
    case 22:  // global_store_dwordx2 + byte store issue cost, no wait
        T0; asm volatile(".rept " STR(REP) "\n global_store_dwordx2 %0, v[40:41], off\n global_store_byte %0, v40, off offset:2048\n .endr" :: "v"(out + 8192 + lane) : "v40", "v41", "memory"); T1; break;
    case 23:  // s_and_saveexec + s_cbranch_execz (taken: all lanes off) + restore
        T0; asm volatile(".rept " STR(REP) "\n v_cmp_eq_u32 vcc, 0x7fffffff, %0\n s_and_saveexec_b64 s[20:21], vcc\n s_cbranch_execz 1f\n v_add_u32 %0, %0, %1\n1:\n s_or_b64 exec, exec, s[20:21]\n v_add_u32 %0, %0, %1\n .endr"
                         : "+v"(x) : "v"(y) : "vcc", "s20", "s21"); T1; break;
    case 24:  // v_cvt_f64_i32 + v_mul_f64 (the reward)
        T0; asm volatile(".rept " STR(REP) "\n v_cvt_f64_i32 v[40:41], %0\n v_mul_f64 v[40:41], v[40:41], v[40:41]\n v_mov_b32 %0, v40\n .endr" : "+v"(x) :: "v40", "v41"); T1; break;
    case 25:  // dependent v_mad_u32_u24 / v_lshl_add_u32
        T0; asm volatile(".rept " STR(REP) "\n v_lshl_add_u32 %0, %0, 2, %1\n .endr" : "+v"(x) : "v"(y)); T1; break;
    case 26:  // v_cmp_e64 sgpr -> v_cndmask using that sgpr directly (no SALU in between)
        T0; asm volatile(".rept " STR(REP) "\n v_cmp_ne_u32 s[20:21], 0, %0\n v_cndmask_b32 %0, %1, %0, s[20:21]\n .endr" : "+v"(x) : "v"(y) : "s20", "s21"); T1; break;
    case 27:  // SDWA add with sext byte
        T0; asm volatile(".rept " STR(REP) "\n v_add_u32_sdwa %0, %0, sext(%1) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0\n .endr" : "+v"(x) : "v"(y)); T1; break;
    case 28:  // ds_or_b32 x2 + reads + clears, SAME bank conflicts: all lanes of an 8-lane group hit one word
        T0; asm volatile(".rept " STR(REP) "\n ds_or_b32 %1, %2\n ds_or_b32 %1, %2 offset:8192\n ds_read_b32 v40, %1\n ds_read_b32 v41, %1 offset:8192\n s_waitcnt lgkmcnt(0)\n v_and_b32 %0, v40, v41\n .endr"
                         : "+v"(x) : "v"((lane >> 3) * 4), "v"(y) : "v40", "v41", "memory"); T1; break;
    case 29:  // dependent s_add (SALU chain)
        T0; asm volatile(".rept " STR(REP) "\n s_add_u32 s20, s20, 1\n .endr" ::: "s20", "scc"); T1; break;
    case 30:  // one VALU + one independent SALU per copy: do they share the wave's issue slots?
        T0; asm volatile(".rept " STR(REP) "\n v_add_u32 %0, %0, %1\n s_add_u32 s20, s20, 1\n .endr" : "+v"(x) : "v"(y) : "s20", "scc"); T1; break;
    case 31:  // one VALU + three independent SALU per copy
        T0; asm volatile(".rept " STR(REP) "\n v_add_u32 %0, %0, %1\n s_add_u32 s20, s20, 1\n s_bcnt1_i32_b64 s21, s[22:23]\n s_add_u32 s24, s24, s21\n .endr"
                         : "+v"(x) : "v"(y) : "s20", "s21", "s24", "scc"); T1; break;
    case 32:  // SALU-only branch, not taken, operands ready
        T0; asm volatile(".rept " STR(REP) "\n s_cmp_eq_u32 s20, 63\n s_cbranch_scc1 1f\n v_add_u32 %0, %0, %1\n1:\n .endr" : "+v"(x) : "v"(y) : "s20", "scc"); T1; break;
    case 33:  // ds_read_b64 chase, 8-byte aligned addresses
        T0; asm volatile(".rept " STR(REP) "\n ds_read_b64 v[40:41], %0\n s_waitcnt lgkmcnt(0)\n v_and_b32 %0, 0x3ff8, v40\n .endr" : "+v"(z) :: "v40", "v41"); T1; x = z; break;
    case 34:  // ds_write_b128 (distinct) + ds_write_b32 to ONE word from all lanes, issue only
        T0; asm volatile(".rept " STR(REP) "\n ds_write_b128 %0, v[40:43]\n ds_write_b32 %1, v40\n .endr" :: "v"(z * 4), "v"(16384u) : "v40", "v41", "v42", "v43", "memory"); T1; break;
    case 35:  // the occupancy round trip with 12 independent VALU ops issued in its shadow
        T0; asm volatile(".rept " STR(REP) "\n ds_or_b32 %1, %2\n ds_or_b32 %1, %2 offset:8192\n ds_read_b32 v40, %1\n ds_read_b32 v41, %1 offset:8192\n"
                         " .rept 12\n v_add_u32 %3, %3, %2\n .endr\n s_waitcnt lgkmcnt(0)\n v_and_b32 %0, v40, v41\n .endr"
                         : "+v"(x) : "v"(z), "v"(y), "v"(a1) : "v40", "v41", "memory"); T1; break;
    case 36:  // v_cmp -> sgpr, 6 independent VALU, then s_cmp + branch (deferred test)
        T0; asm volatile(".rept " STR(REP) "\n v_cmp_ne_u32 s[20:21], -1, %0\n .rept 6\n v_add_u32 %2, %2, %1\n .endr\n s_cmp_eq_u64 s[20:21], 0\n s_cbranch_scc1 1f\n v_add_u32 %0, %0, %1\n1:\n .endr"
                         : "+v"(x) : "v"(y), "v"(a1) : "s20", "s21", "scc"); T1; break;
    case 37:  // 7 dependent VALU (the cost of the 6 fillers + 1 of case 36, for comparison)
        T0; asm volatile(".rept " STR(REP) "\n .rept 7\n v_add_u32 %0, %0, %1\n .endr\n .endr" : "+v"(x) : "v"(y)); T1; break;
    case 38:  // ds_read_b128 of own slot + ds_read_b32 broadcast poll
        T0; asm volatile(".rept " STR(REP) "\n ds_read_b32 v44, %1\n s_waitcnt lgkmcnt(0)\n v_readfirstlane_b32 s20, v44\n s_cmp_lg_u32 s20, 63\n s_cbranch_scc0 1f\n ds_read_b128 v[40:43], %0\n s_waitcnt lgkmcnt(0)\n1:\n .endr"
                         :: "v"(z * 4), "v"(16384u) : "v40", "v41", "v42", "v43", "v44", "s20", "scc", "memory"); T1; break;
    }
    if (lane == 0) out[which * 8 + (threadIdx.x >> 6)] = t1 - t0;
    if (x == 0x12345678u) out[1000] = x;   // keep results alive
}

int main() {
    const char* names[] = {"dep v_add_u32", "4 indep v_add_u32 (per 4)", "dep v_lshrrev_b64", "dep v_and_or_b32", "v_cmp vcc -> v_cndmask",
        "v_cmp sgpr -> s_and -> v_cndmask", "v_cmp vcc -> s_cbranch not taken + v_add", "v_cmp vcc -> s_cbranch TAKEN + v_add",
        "v_cmp sgpr -> v_lshrrev_b64 -> v_and -> v_or (group_bits)", "v_cmp sgpr -> s_cmp_eq_u64 -> s_cbranch_scc nt + v_add",
        "ds_read_b32 chase", "ds_read_b64 chase + mov", "ds_or x2 + ds_read x2 + wait + v_and", "ds_write_b128 + lgkmcnt(0)", "s_barrier",
        "v_readfirstlane -> s_add -> v_add", "s_nop 1 + v_add dpp quad_perm", "ds_bpermute + wait", "s_memrealtime + wait", "dep v_perm_b32",
        "ds_write_b128 + ds_write_b32 issue", "global_store_dwordx4 nt issue", "global_store_dwordx2 + byte issue",
        "v_cmp + saveexec + cbranch_execz taken + restore + 1 v_add", "v_cvt_f64_i32 + v_mul_f64 + mov", "dep v_lshl_add_u32",
        "v_cmp sgpr -> v_cndmask(sgpr)", "dep v_add_u32_sdwa sext", "ds_or x2 + reads, 8-way same word", "dep s_add_u32",
        "v_add + indep s_add", "v_add + 3 indep SALU", "s_cmp + s_cbranch_scc nt (ready) + v_add", "ds_read_b64 chase aligned + and",
        "ds_write_b128 + same-word ds_write_b32 issue", "occupancy round trip + 12 VALU in shadow", "v_cmp, 6 VALU, s_cmp + branch nt + v_add",
        "7 dep VALU", "poll word + readfirstlane + branch + ds_read_b128"};
    const int ncase = sizeof(names) / sizeof(names[0]);
    unsigned long long* d;
    hipMalloc(&d, 1 << 20);
    std::vector<unsigned long long> h(4096);
    for (int waves = 1; waves <= 4; waves *= 2) {
        if (waves == 2) continue;
        printf("---- %d wave(s) per workgroup (one per SIMD), %d copies per measurement, shader clocks per copy ----\n", waves, REP);
        for (int c = 0; c < ncase; ++c) {
            double best = 1e30;
            for (int r = 0; r < 5; ++r) {
                hipMemset(d, 0, 32768);
                hipLaunchKernelGGL(k, dim3(1), dim3(64 * waves), 65536, 0, d, 12345u + r, c);
                hipDeviceSynchronize();
                hipMemcpy(h.data(), d, 32768, hipMemcpyDeviceToHost);
                double v = (double)h[c * 8] / REP;
                if (v < best) best = v;
            }
            printf("%2d %-60s %8.1f\n", c, names[c], best);
        }
    }
    // empty measurement overhead: two timestamps back to back
    return 0;
}
