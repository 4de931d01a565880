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
#define REPS 2000
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
        if (KIND == 0) {           // 64 dependent f32 ops
            REP8(REP8(asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(a0));))
        } else if (KIND == 1) {    // 64 f32 ops, 8 independent chains
            REP8(asm volatile("v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %1, %1, %1, %1\n v_fma_f32 %2, %2, %2, %2\n v_fma_f32 %3, %3, %3, %3\n"
                              "v_fma_f32 %4, %4, %4, %4\n v_fma_f32 %5, %5, %5, %5\n v_fma_f32 %6, %6, %6, %6\n v_fma_f32 %7, %7, %7, %7"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
        } else if (KIND == 2) {    // 64 f64 adds, 4 independent chains
            REP8(REP4(asm volatile("v_add_f64 %0, %0, %0\n v_add_f64 %1, %1, %1" : "+v"(d0), "+v"(d1));) )
        } else if (KIND == 3) {    // 64 dependent f64 adds
            REP8(REP8(asm volatile("v_add_f64 %0, %0, %0" : "+v"(d0));))
        } else if (KIND == 4) {    // 32 VALU (4 chains) interleaved with 32 SALU (dependent): does the scalar unit run beside the vector unit?
            REP8(asm volatile("v_fma_f32 %0, %0, %0, %0\n s_add_i32 %4, %4, 1\n v_fma_f32 %1, %1, %1, %1\n s_add_i32 %4, %4, 1\n"
                              "v_fma_f32 %2, %2, %2, %2\n s_add_i32 %4, %4, 1\n v_fma_f32 %3, %3, %3, %3\n s_add_i32 %4, %4, 1"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+s"(s0) :: "scc");)
        } else if (KIND == 5) {    // 64 int32 ops (v_add_u32 / v_and), 8 chains
            int* p = (int*)&a0; (void)p;
            REP8(asm volatile("v_add_u32 %0, %0, %0\n v_add_u32 %1, %1, %1\n v_add_u32 %2, %2, %2\n v_add_u32 %3, %3, %3\n"
                              "v_add_u32 %4, %4, %4\n v_add_u32 %5, %5, %5\n v_add_u32 %6, %6, %6\n v_add_u32 %7, %7, %7"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
        } else if (KIND == 6) {    // DPP-modified ops (row_shr), 8 chains: do they issue at the plain rate?
            REP8(asm volatile("v_add_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %1, %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                              "v_add_f32_dpp %2, %2, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %3, %3, %3 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                              "v_add_f32_dpp %4, %4, %4 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %5, %5, %5 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                              "v_add_f32_dpp %6, %6, %6 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %7, %7, %7 row_shr:1 row_mask:0xf bank_mask:0xf"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
        } else if (KIND == 7) {    // 64 v_cndmask (select), 8 chains
            REP8(asm volatile("v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %1, %1, %2, vcc\n v_cndmask_b32 %2, %2, %3, vcc\n v_cndmask_b32 %3, %3, %4, vcc\n"
                              "v_cndmask_b32 %4, %4, %5, vcc\n v_cndmask_b32 %5, %5, %6, vcc\n v_cndmask_b32 %6, %6, %7, vcc\n v_cndmask_b32 %7, %7, %0, vcc"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
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
    run<0>("v_fma_f32 dependent chain", out, buf);
    run<1>("v_fma_f32 8 independent chains", out, buf);
    run<2>("v_add_f64 2 independent chains", out, buf);
    run<3>("v_add_f64 dependent chain", out, buf);
    run<4>("v_fma_f32 x32 + s_add_i32 x32", out, buf);
    run<5>("v_add_u32 8 chains", out, buf);
    run<6>("v_add_f32_dpp row_shr 8 chains", out, buf);
    run<7>("v_cndmask 8 chains", out, buf);
    return 0;
}
