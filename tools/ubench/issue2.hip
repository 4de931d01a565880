// VALU issue-rate probe for gfx950 (diagnostic; not part of the product).
// Question: how many cycles does one wave64 VALU instruction occupy its SIMD when W waves share the SIMD
// and each wave's stream is (a) one dependent chain, (b) 8 independent chains?  lat.hip measured ONE wave: 4 cycles.
// Every CU gets exactly 4*W single-wave workgroups (LDS sized so that no more fit), each runs REPS x 64 VALU instructions
// between two s_memtime stamps; cycles per wave-instruction per SIMD = elapsed / (REPS * 64 * W).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define REP4(x) x x x x
#define REP8(x) REP4(x) REP4(x)
#define REPS 1000
__device__ __forceinline__ unsigned long long now() { unsigned long long t; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory"); return t; }
__device__ __forceinline__ unsigned long long realnow() { unsigned long long t; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory"); return t; }

template <int KIND> __global__ void __launch_bounds__(64) k(long long* out, float* buf, int lds_words)
{
    extern __shared__ float lds[];
    const int lane = threadIdx.x;
    if (lds_words > 0) lds[lane] = 0.0f;
    __syncthreads();
    float a0 = buf[lane], a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    double d0 = a0, d1 = a1, d2 = a2, d3 = a3;
    int s0 = __builtin_amdgcn_readfirstlane(lane & 1);
    const unsigned long long r0 = realnow(), t0 = now();
    for (int r = 0; r < REPS; r++) {
#define R8(x) REP8(x)
#define C8(op) asm volatile(op " %0, %0, %1\n " op " %1, %1, %2\n " op " %2, %2, %3\n " op " %3, %3, %4\n " op " %4, %4, %5\n " op " %5, %5, %6\n " op " %6, %6, %7\n " op " %7, %7, %0" \
                           : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        if (KIND == 0) { R8(C8("v_add_f32")) }
        else if (KIND == 1) { R8(C8("v_mul_f32")) }
        else if (KIND == 2) { R8(C8("v_max_f32")) }
        else if (KIND == 3) { R8(C8("v_and_b32")) }
        else if (KIND == 4) { R8(C8("v_add_u32")) }
        else if (KIND == 5) { R8(C8("v_lshlrev_b32")) }
        else if (KIND == 6) {    // v_cndmask VOP2 with vcc, independent destinations
            R8(asm volatile("v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %1, %1, %2, vcc\n v_cndmask_b32 %2, %2, %3, vcc\n v_cndmask_b32 %3, %3, %4, vcc\n"
                              "v_cndmask_b32 %4, %4, %5, vcc\n v_cndmask_b32 %5, %5, %6, vcc\n v_cndmask_b32 %6, %6, %7, vcc\n v_cndmask_b32 %7, %7, %0, vcc"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) :: );)
        } else if (KIND == 7) {  // v_cndmask VOP3 with an SGPR pair as mask
            R8(asm volatile("v_cndmask_b32 %0, %0, %1, s[20:21]\n v_cndmask_b32 %1, %1, %2, s[20:21]\n v_cndmask_b32 %2, %2, %3, s[20:21]\n v_cndmask_b32 %3, %3, %4, s[20:21]\n"
                              "v_cndmask_b32 %4, %4, %5, s[20:21]\n v_cndmask_b32 %5, %5, %6, s[20:21]\n v_cndmask_b32 %6, %6, %7, s[20:21]\n v_cndmask_b32 %7, %7, %0, s[20:21]"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) :: "s20", "s21");)
        } else if (KIND == 8) {  // compare + select pairs (4 pairs = 8 instr)
            R8(asm volatile("v_cmp_lt_f32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc\n v_cmp_lt_f32 vcc, %2, %3\n v_cndmask_b32 %2, %2, %3, vcc\n"
                              "v_cmp_lt_f32 vcc, %4, %5\n v_cndmask_b32 %4, %4, %5, vcc\n v_cmp_lt_f32 vcc, %6, %7\n v_cndmask_b32 %6, %6, %7, vcc"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) :: "vcc");)
        } else if (KIND == 9) {  // f32 <-> f64 conversions (4 round trips = 8 instr)
            R8(asm volatile("v_cvt_f64_f32 %4, %0\n v_cvt_f32_f64 %0, %4\n v_cvt_f64_f32 %5, %1\n v_cvt_f32_f64 %1, %5\n v_cvt_f64_f32 %6, %2\n v_cvt_f32_f64 %2, %6\n v_cvt_f64_f32 %7, %3\n v_cvt_f32_f64 %3, %7"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3));)
        } else if (KIND == 10) { // v_readlane to SGPR (8 per group)
            R8(asm volatile("v_readlane_b32 s20, %0, 1\n v_readlane_b32 s21, %1, 2\n v_readlane_b32 s22, %2, 3\n v_readlane_b32 s23, %3, 4\n"
                              "v_readlane_b32 s24, %4, 5\n v_readlane_b32 s25, %5, 6\n v_readlane_b32 s26, %6, 7\n v_readlane_b32 s27, %7, 8"
                              :: "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6), "v"(a7) : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");)
        } else if (KIND == 11) { // LDS reads, 8 independent b32 per group then one wait
            R8(asm volatile("ds_read_b32 %0, %8\n ds_read_b32 %1, %8 offset:256\n ds_read_b32 %2, %8 offset:512\n ds_read_b32 %3, %8 offset:768\n"
                              "ds_read_b32 %4, %8 offset:1024\n ds_read_b32 %5, %8 offset:1280\n ds_read_b32 %6, %8 offset:1536\n ds_read_b32 %7, %8 offset:1792\n s_waitcnt lgkmcnt(0)"
                              : "=v"(a0), "=v"(a1), "=v"(a2), "=v"(a3), "=v"(a4), "=v"(a5), "=v"(a6), "=v"(a7) : "v"(lane * 4) : "memory");)
        }         else if (KIND == 13) {   // SALU only: 8 dependent s_add
            R8(asm volatile("s_add_i32 %0, %0, 1\n s_add_i32 %0, %0, 1\n s_add_i32 %0, %0, 1\n s_add_i32 %0, %0, 1\n s_add_i32 %0, %0, 1\n s_add_i32 %0, %0, 1\n s_add_i32 %0, %0, 1\n s_add_i32 %0, %0, 1" : "+s"(s0) :: "scc");)
        } else if (KIND == 14) { // v_fma_f32 with 2 independent chains only
            R8(asm volatile("v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %1, %1, %1, %1\n v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %1, %1, %1, %1\n v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %1, %1, %1, %1\n v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %1, %1, %1, %1" : "+v"(a0), "+v"(a1));)
        } else if (KIND == 15) { // dependent v_add_f32 interleaved with a dependent SALU chain and LDS read (mixed stream like the codec)
            R8(asm volatile("v_add_f32 %0, %0, %0\n s_add_i32 %2, %2, 1\n v_add_f32 %0, %0, %0\n s_add_i32 %2, %2, 1\n v_add_f32 %0, %0, %0\n ds_read_b32 %1, %3\n v_add_f32 %0, %0, %0\n s_add_i32 %2, %2, 1\n s_waitcnt lgkmcnt(0)"
                              : "+v"(a0), "=v"(a1), "+s"(s0) : "v"(lane * 4) : "scc", "memory");)
        }
    }
    const unsigned long long t1 = now(), r1 = realnow();
    if (lane == 0) { out[2 * blockIdx.x] = (long long)(t1 - t0); out[2 * blockIdx.x + 1] = (long long)(r1 - r0); }
    buf[64 + lane] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (float)(d0 + d1 + d2 + d3) + (float)s0;
}

template <int KIND> static void run(const char* name, long long* out, float* buf)
{
    for (int W : {1, 2, 4, 8}) {
        const int per_cu = 4 * W, nblk = 256 * per_cu;
        // LDS per block such that exactly per_cu blocks fit a CU (160 KiB): forces an even spread over the CUs
        size_t lds = (160 * 1024) / per_cu; if (lds > 64 * 1024) lds = 64 * 1024;
        if (W == 1) lds = 40 * 1024;    // 4 blocks x 40 KiB = 160 KiB
        hipFuncSetAttribute((const void*)k<KIND>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        for (int rep = 0; rep < 2; rep++) k<KIND><<<nblk, 64, lds>>>(out, buf, 1);
        hipDeviceSynchronize();
        std::vector<long long> h(2 * nblk);
        hipMemcpy(h.data(), out, sizeof(long long) * 2 * nblk, hipMemcpyDeviceToHost);
        std::vector<double> cyc(nblk), clk(nblk);
        for (int i = 0; i < nblk; i++) { cyc[i] = (double)h[2 * i]; clk[i] = (double)h[2 * i] / ((double)h[2 * i + 1] / 100.0); }   // s_memrealtime: 100 MHz
        std::sort(cyc.begin(), cyc.end()); std::sort(clk.begin(), clk.end());
        const double med = cyc[nblk / 2];
        printf("%-34s W=%d waves/SIMD: median %8.0f cycles per wave for %d instr -> %.2f cycles per wave-instr per wave, %.2f per SIMD slot; clock %.2f GHz (MHz %.0f)\n",
               name, W, med, REPS * 64, med / (REPS * 64.0), med / (REPS * 64.0 * W), clk[nblk / 2] / 1000.0, clk[nblk / 2]);
    }
}

int main()
{
    long long* out; float* buf;
    hipMalloc(&out, sizeof(long long) * 2 * 256 * 32); hipMalloc(&buf, 1024);
    hipMemset(buf, 0, 1024);
    run<0>("v_add_f32 8 chains", out, buf);
    run<1>("v_mul_f32 8 chains", out, buf);
    run<2>("v_max_f32 8 chains", out, buf);
    run<3>("v_and_b32 8 chains", out, buf);
    run<4>("v_add_u32 8 chains (ring)", out, buf);
    run<5>("v_lshlrev_b32 8 chains", out, buf);
    run<6>("v_cndmask vcc 8 chains", out, buf);
    run<7>("v_cndmask s[20:21] 8 chains", out, buf);
    run<8>("v_cmp + v_cndmask pairs", out, buf);
    run<9>("cvt f32->f64->f32 x4", out, buf);
    run<10>("v_readlane x8", out, buf);
    run<11>("ds_read_b32 x8 + wait", out, buf);
    run<13>("s_add_i32 dependent x8 (SALU only)", out, buf);
    run<14>("v_fma_f32 2 chains", out, buf);
    run<15>("4 dep v_add + 3 s_add + ds_read + wait (8)", out, buf);
    return 0;
}
