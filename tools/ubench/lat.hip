// Dependent-chain latency probes for gfx950 (diagnostic; not part of the product).  One wave, s_memtime around N dependent ops.
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)
#define REP64(x) REP16(x) REP16(x) REP16(x) REP16(x)
#define N 64
__device__ __forceinline__ unsigned long long now() { unsigned long long t; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory"); return t; }

__global__ void k(long long* out, float* buf, int one)
{
    __shared__ float lds[256];
    const int lane = threadIdx.x;
    lds[lane] = 0.0f; lds[lane + 64] = 0; lds[lane+128]=0; lds[lane+192]=0;
    __syncthreads();
    float f = buf[lane]; double d = (double)buf[lane + 64]; int i = lane; int s = one;
    unsigned long long t0, t1; int id = 0;
#define RUN(name, body) { t0 = now(); REP64(body) t1 = now(); if (lane == 0) out[id] = (long long)(t1 - t0); id++; }
    RUN("v_add_f32", asm volatile("v_add_f32 %0, %0, %0" : "+v"(f));)
    RUN("v_mul_f32+v_add_f32", asm volatile("v_mul_f32 %0, %0, %0\n v_add_f32 %0, %0, %0" : "+v"(f));)
    RUN("v_add_f64", asm volatile("v_add_f64 %0, %0, %0" : "+v"(d));)
    RUN("v_mul_f64", asm volatile("v_mul_f64 %0, %0, %0" : "+v"(d));)
    RUN("v_fma_f64", asm volatile("v_fma_f64 %0, %0, %0, %0" : "+v"(d));)
    RUN("cvt f32->f64->f32", asm volatile("v_cvt_f64_f32 %1, %0\n v_cvt_f32_f64 %0, %1" : "+v"(f), "+v"(d));)
    RUN("cvt64,add64,cvt32", asm volatile("v_cvt_f64_f32 %1, %0\n v_add_f64 %1, %1, %1\n v_cvt_f32_f64 %0, %1" : "+v"(f), "+v"(d));)
    RUN("v_cndmask", asm volatile("v_cndmask_b32 %0, %0, %0, vcc" : "+v"(i) :: );)
    RUN("v_cmp+v_cndmask", asm volatile("v_cmp_lt_f32 vcc, %0, %0\n v_cndmask_b32 %0, %0, %0, vcc" : "+v"(f) :: "vcc");)
    RUN("s_add_i32", asm volatile("s_add_i32 %0, %0, %0" : "+s"(s));)
    RUN("s_mul_i32", asm volatile("s_mul_i32 %0, %0, %0" : "+s"(s));)
    RUN("s_lshr+s_mul+s_cmp+s_cselect", asm volatile("s_lshr_b32 %0, %0, 1\n s_mul_i32 %0, %0, %0\n s_cmp_lt_u32 %0, 5\n s_cselect_b32 %0, %0, 7" : "+s"(s) :: "scc");)
    RUN("v_readlane(sidx)->s", asm volatile("v_readlane_b32 %0, %1, %0" : "+s"(s) : "v"(i));)
    RUN("v_readfirstlane->v_mov", asm volatile("v_readfirstlane_b32 %1, %0\n s_nop 0\n v_mov_b32 %0, %1" : "+v"(i), "+s"(s));)
    RUN("ds_read_b32 dep", asm volatile("ds_read_b32 %0, %0\n s_waitcnt lgkmcnt(0)" : "+v"(i) :: "memory");)
    RUN("ds_bpermute dep", asm volatile("ds_bpermute_b32 %0, %0, %0\n s_waitcnt lgkmcnt(0)" : "+v"(i) :: "memory");)
    RUN("dpp row_shr add", asm volatile("s_nop 1\n v_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(i));)
    RUN("v_mov dpp wave_shr? (row_bcast15)", asm volatile("s_nop 1\n v_add_u32_dpp %0, %0, %0 row_bcast:15 row_mask:0xf bank_mask:0xf" : "+v"(i));)
    RUN("s_cbranch taken", asm volatile("s_cmp_eq_u32 %0, %0\n s_cbranch_scc1 1f\n s_nop 0\n1:" : "+s"(s) :: "scc");)
    RUN("saveexec branch", asm volatile("v_cmp_eq_u32 vcc, %0, %0\n s_and_saveexec_b64 s[20:21], vcc\n s_cbranch_execz 1f\n v_add_u32 %0, %0, %0\n1:\n s_or_b64 exec, exec, s[20:21]" : "+v"(i) :: "vcc", "s20", "s21");)
    RUN("v_exp_f32", asm volatile("v_exp_f32 %0, %0" : "+v"(f));)
    RUN("v_rcp_f64", asm volatile("v_rcp_f64 %0, %0" : "+v"(d));)
    RUN("v_mul_lo_u32", asm volatile("v_mul_lo_u32 %0, %0, %0" : "+v"(i));)
    RUN("v_cvt_f32_i32,v_cvt_i32_f32", asm volatile("v_cvt_f32_i32 %0, %0\n v_cvt_i32_f32 %0, %0" : "+v"(i));)
    RUN("lds write+read same addr", asm volatile("ds_write_b32 %1, %0\n ds_read_b32 %0, %1\n s_waitcnt lgkmcnt(0)" : "+v"(f) : "v"(lane * 4) : "memory");)
    RUN("ballot->s->v", asm volatile("v_cmp_ne_u32 vcc, 0, %0\n s_nop 0\n v_mov_b32 %0, vcc_lo" : "+v"(i) :: "vcc");)
    buf[lane] = f + (float)d + (float)i + (float)s;
}
int main()
{
    long long* out; float* buf;
    hipMalloc(&out, 64 * 8); hipMalloc(&buf, 256 * 4);
    hipMemset(buf, 0, 1024); hipMemset(out, 0, 512);
    for (int r = 0; r < 3; r++) k<<<1, 64>>>(out, buf, 1);
    hipDeviceSynchronize();
    long long h[64]; hipMemcpy(h, out, 512, hipMemcpyDeviceToHost);
    const char* names[] = {"v_add_f32", "v_mul_f32+v_add_f32 (2)", "v_add_f64", "v_mul_f64", "v_fma_f64", "cvt f32->f64->f32 (2)", "cvt64,add64,cvt32 (3)", "v_cndmask",
        "v_cmp+v_cndmask (2)", "s_add_i32", "s_mul_i32", "s_lshr,s_mul,s_cmp,s_cselect (4)", "v_readlane(sidx)->s", "v_readfirstlane->v_mov (2)", "ds_read_b32 dep", "ds_bpermute dep",
        "dpp row_shr add", "dpp row_bcast15 add", "s_cbranch taken (cmp+br)", "saveexec branch seq", "v_exp_f32", "v_rcp_f64", "v_mul_lo_u32", "cvt_f32_i32+cvt_i32_f32 (2)",
        "lds write+read", "vcc->v_mov"};
    for (int i = 0; i < 26; i++) printf("%-36s %7.1f cycles per rep (64 reps: %lld)\n", names[i], (h[i] - 40) / 64.0, h[i]);
    return 0;
}
