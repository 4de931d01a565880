/* lc3_kernels.hip -- gfx950 (MI355X / CDNA4) LC3plus encode / decode kernels + the C-ABI device shim.
 *
 * This file holds (1) the stage functions of the encoder - one per stage of the reference, each citing its lines - and lc3_encode_kernel, which runs
 * all of them with ONE CHANNEL-STREAM PER WAVEFRONT, frames in time order, everything in that wave's LDS slice: the path of traced launches, of calls of
 * a few frames and of the single-stream lc3_enc_* API; (2) the device shim (contexts, launches, streams, events).  The product path for batches is the
 * PIPELINE of kernels in the lc3_enc_*.inc files included below - resampler, HP50, pitch chain, MDCT front, scale factors, SNS quantiser, shaping + TNS,
 * rate chain, bitstream writer - each with the unit of work its dependences allow (one frame per lane, four frames per wave, one or two channel-streams
 * per wave), hand-overs in HBM, three HIP streams, up to three calls in flight.  PCM is read from HBM with coalesced loads, bytes are
 * written back coalesced.  No MFMA (nothing here is a dense contraction), no collectives.
 *
 * Numerics contract: every floating-point expression keeps the ETSI reference's evaluation order and
 * C promotions (R = LC3plus_ETSI_src_v17171_20200723/src/floating_point, cited per stage), compiled with
 * -ffp-contract=off, so that decisions (argmax, thresholds, quantisation) match the reference bit for bit.
 * Independent serial sums (autocorrelation lags, FIR taps, band energies ...) are mapped one sum per lane,
 * which keeps the reference's summation order AND fills the wave.  Strictly serial chains (biquad, normalised
 * correlations, bisection, range coder) run on wave-uniform values: operands are fetched from lane registers
 * with v_readlane (no LDS round trip) and integer state lives in scalar registers.  Run-time libm calls of the
 * reference (log2f, log10f, powf(2, .)) are evaluated as (float)f((double)x): lc3_fastmath.h, identical to glibc's double functions for every float argument.
 *
 * Stage functions are deliberately NOT inlined: each gets its own register allocation, which keeps the kernel
 * at <= 3 waves' worth of VGPRs per SIMD instead of the union of all live ranges.
 */
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <stddef.h>
#include <type_traits>

#define LC3T_QUAL static __device__ const
#include "lc3_fastmath.h"
#include "lc3_tables.h"
#include "lc3_plan.h"
#include "lc3_shim.h"

/* Two LDS layouts of the same code.  Standard: frames up to 480 samples with an MDCT memory of at most 300 (every operating
 * point except two), 10 KB per wave, 4 waves per SIMD.  Large (-DLC3_BIG, kernel lc3_encode_kernel_big): 96 kHz / 10 ms (N = 960)
 * and 96 kHz / 5 ms (N = 480, MDCT memory 360), 17 KB per wave, 2 waves per SIMD. */
#ifdef LC3_BIG
#define MAXN 960
#define MEMCAP LC3D_MEMCAP_BIG
#define SMW 1088                /* sm[]: holds the resampler's scaled input (120 + 960 samples) */
#define KERNEL_NAME lc3_encode_kernel_big
#define KERNEL_WAVES 2
#else
#define MAXN 480
#define MEMCAP LC3D_MEMCAP_STD  /* MDCT overlap memory: N - la_zeros <= 300 for every N <= 480 except 96 kHz / 5 ms */
#define SMW 548
#define KERNEL_NAME lc3_encode_kernel
#ifndef KERNEL_WAVES
#define KERNEL_WAVES 4
#endif
#endif
#define NQL ((MAXN / 4 + 63) / 64)   /* bisection energies (4 bins each) per lane: 2 or 4 */
#define WAVE 64
#define LSYNC() __syncthreads()
#define STAGE __device__ __attribute__((noinline))

/* Diagnostic build only (-DLC3_STAGE_TIMING, tools/stage_timing.py): per-stage wave-latency accounting with s_memtime.
 * The product library is built without it; no stamp executes there. */
#ifdef LC3_STAGE_TIMING
#define NSTAGE 64                       /* 0..23: stages, 24..63: sub-stage stamps */
#define TICK(id) do { long long now_ = clock64(); if (lane == 0) L.tacc[id] += now_ - tlast; tlast = now_; } while (0)
#define SUB_BEGIN() long long ts_ = clock64()
#define SUB(id) do { long long n_ = clock64(); if (lane == 0) L.tacc[24 + (id)] += n_ - ts_; ts_ = n_; } while (0)
#elif defined(LC3_STOP_AFTER)
/* Diagnostic build only (tools/stage_counts.sh): every frame ends after stage LC3_STOP_AFTER, so that instruction counters of
 * consecutive variants differ by exactly one stage.  Output bytes are meaningless in such a build. */
#define TICK(id) if ((id) == LC3_STOP_AFTER) continue
#define SUB_BEGIN() do { } while (0)
#define SUB(id) do { } while (0)
#else
#define TICK(id) do { } while (0)
#define SUB_BEGIN() do { } while (0)
#define SUB(id) do { } while (0)
#endif

/* ------------------------------------------------------------------------------------------------ */
/* LDS slice of one wave (~12.8 KB -> 12 waves per CU)                                                */
/* ------------------------------------------------------------------------------------------------ */
struct __attribute__((aligned(16))) WaveLds {
    static constexpr bool CDW = true;       /* st_quantize leaves the per-tuple coder words st_bitstream reads */
    static constexpr int MISC = 368;   /* = SM_MISC: where the scratch vectors of this layout's sm[] start */
    static constexpr int XOFF = MEMCAP; /* the frame half of xbuf (XCUR / XQ) */
    static constexpr int TNS0 = 368, TNS1 = 240;   /* TNS work areas in sm[]: filter 0 from SM_MISC, filter 1 in the (idle) SM_PVQ area */
    static constexpr int RESO = MAXN / 2, LSTO = MAXN / 2 + 2;   /* residual bits / noise-level list behind the coder words */
    float xbuf[MEMCAP + MAXN];  /* [MDCT/resampler memory right-aligned in 0..MEMCAP | current frame X]; once the MDCT fold has consumed the
                                   frame, X is scratch (DFT ping buffer, TNS output) and finally the quantised spectrum xq */
    float A[MAXN];              /* scratch, then the MDCT spectrum (shaped / TNS-filtered in place); the output frame during the bitstream stage */
    float h12[384];             /* HP-filtered 12.8 kHz stream, newest sample at [383] */
    float h6[194];              /* 6.4 kHz stream, newest at [193] */
    float sm[SMW];              /* small vectors (SM_*); from quantisation on: cdw[MAXN/2] | residual / LSB bits (160 words) */
    int   pc[LC3D_PLAN_HEAD_WORDS];  /* the scalar head of the plan (lc3d_plan up to pad0), copied once: stage code reads it from LDS
                                   instead of through a flat pointer, and readfirstlane makes the values scalar */
    int   cc[14];               /* this channel-stream's lc3d_chan */
    float fsc[12];              /* float scalars: cross-frame state + values passed between stages */
    int   isc[56];              /* integer scalars */
#ifdef LC3_STAGE_TIMING
    long long tacc[NSTAGE];
#endif
};
static_assert(offsetof(lc3d_plan, tw1) == 4 * LC3D_PLAN_HEAD_WORDS, "plan head size");
static_assert(offsetof(WaveLds, A) % 16 == 0 && (offsetof(WaveLds, sm) + (MAXN / 2 + 2) * 4) % 16 == 0 && offsetof(WaveLds, xbuf) == 0, "16-byte aligned LDS rows");
/* the slice of lc3_enc_front_kernel: what the stateless front needs, nothing of the pitch buffers or the coder's work areas -> 6 KB, six waves per SIMD */
struct __attribute__((aligned(16))) FrontLds {
    static constexpr int MISC = 96;
    static constexpr int XOFF = MEMCAP;
    float xbuf[MEMCAP + MAXN];
    float A[MAXN];
    float sm[160];              /* band energies [0..63], scale factors [64..79], scratch from MISC */
    int   pc[LC3D_PLAN_HEAD_WORDS];
    int   cc[14];
    float fsc[12];
    int   isc[56];
#ifdef LC3_STAGE_TIMING
    long long tacc[NSTAGE];
#endif
};
#define PI(f) uni(L.pc[offsetof(lc3d_plan, f) / 4])
#define PF(f) __int_as_float(uni(L.pc[offsetof(lc3d_plan, f) / 4]))
#define CI(f) uni(L.cc[offsetof(lc3d_chan, f) / 4])
#define XCUR(L) (&(L).xbuf[(L).XOFF])
#define XQ(L)   ((int*)&(L).xbuf[(L).XOFF])
#define SPEC(L) ((L).A)
#define BYTES(L) ((uint8_t*)(L).A)            /* up to 640 bytes, valid from the bitstream stage to the copy-out */
#define CDW(L)  ((uint32_t*)&(L).sm[0])      /* per 2-tuple: ctx(10) | maxlev+1 (6) | pki of the final symbol (6) | sym (5) */
#define RESB(L) ((uint8_t*)&(L).sm[(L).RESO]) /* 640 bytes: residual bits / LSB-mode list (bit-packed, LSB first) */

/* sm[] map (floats) before quantisation */
#define SM_ENER   0     /* 64  band energies, later the interpolated SNS gains */
#define SM_GI     0
#define SM_SCF    64    /* 16 */
#define SM_SCFQ   80    /* 16 */
#define SM_TGT    96    /* 16 pvq target (dct domain) */
#define SM_TGTP   112   /* 16 pvq target pre */
#define SM_ST1    128   /* 16 */
#define SM_VEC    144   /* 6*16 = 96: candidate vectors */
#define SM_PVQ    240   /* 4 searches x (y[16] int, ynorm[16]) = 128 */
#define SM_MISC   368   /* 180: R0 (98), cor, tns r[], idct in/out ... */

/* fsc[] map */
enum { F_HP0 = 0, F_HP1, F_LTPF_NC1, F_LTPF_NC2, F_LTPF_PITCH, F_ATT_M0, F_ATT_M1, F_ATT_ACC, F_TBITS_OFF, F_NC, F_GAIN };
/* isc[] map */
enum { I_OLPA_PITCH = 0, I_LTPF_ON, I_ATT_POS, I_ATT_FLAG, I_MEM_TARGET, I_MEM_SPEC,
       I_T0, I_LTPF0, I_LTPF1, I_LTPF2, I_LTPF_BITS, I_BW, I_SCF0, I_SCF1, I_SCF2, I_SCF3, I_SCF4, I_SCF5, I_SCF6,
       I_TNS_NF, I_TNS_ORD0, I_TNS_ORD1, I_TNS_BITS, I_TNS_IDX0 /* 16 entries */, I_GG = I_TNS_IDX0 + 16, I_GGMIN, I_NBITS, I_NBITS2,
       I_LASTNZ, I_LSB, I_CHANGE, I_FACNS, I_NRES, I_BP_SIDE, I_MASK_SIDE, I_BUDGET /* st_bitstream: side information + coder bits exceed the frame */, I_COUNT };
static_assert(I_COUNT <= 56, "isc[] holds 56 words");

/* ------------------------------------------------------------------------------------------------ */
/* small helpers                                                                                     */
/* ------------------------------------------------------------------------------------------------ */
/* (float)f((double)x) for f = log2, log10, 2^x: lc3_fastmath.h - a dozen double-precision fused multiply-adds behind a table, bit-identical to glibc's
 * log2 / log10 / exp2 / pow(2, .) for EVERY float argument (tools/fastmath_check.c), instead of the device library's 40 ... 80 fp64 instructions.  The
 * one-frame-per-lane kernels read the tables from LDS copies (m_*_t), everything else from global memory (a gather per call). */
#ifdef LC3_OCML_MATH      /* diagnostic builds (tools/variants.sh): the device library's double functions, as until round 3 */
__device__ __forceinline__ float m_log2f(float x) { return (float)log2((double)x); }
__device__ __forceinline__ float m_log10f(float x) { return (float)log10((double)x); }
__device__ __forceinline__ float m_pow2f(float y) { return (float)exp2((double)y); }
#else
__device__ __forceinline__ float m_log2f(float x) { return lc3m_log2f(x, lc3m_log2_tab); }
__device__ __forceinline__ float m_log10f(float x) { return lc3m_log10f(x, lc3m_log10_tab); }
__device__ __forceinline__ float m_pow2f(float y) { return lc3m_exp2f(y, lc3m_exp2_tab); }
#endif
__device__ __forceinline__ float m_powf(float x, float y) { return (float)pow((double)x, (double)y); }
__device__ __forceinline__ float mul_d(float a, double c) { return (float)((double)a * c); }
__device__ __forceinline__ int imin(int a, int b) { return a < b ? a : b; }
__device__ __forceinline__ int imax(int a, int b) { return a > b ? a : b; }
__device__ __forceinline__ int ilog2(unsigned v) { return 31 - __clz((int)v); }
/* wave-uniform value -> scalar register (lets the compiler use SALU + scalar branches for serial code) */
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }

/* floor(log2f((float)v)) as glibc evaluates it: log2f rounds to an integer for the few v just below 2^b
 * (SURVEY 9): 2^21-1, 2^22-{1,2}, 2^23-{1..5}, 2^24-{1..11}. */
__device__ __forceinline__ int flog2f_int(unsigned v)
{
    int e = ilog2(v);
    if (v >= (1u << 20)) {                          /* the only values the rounding can lift; a masked region the wave skips otherwise */
        const int b = e + 1;
        const unsigned k = (1u << b) - v;
        if ((b == 21 && k <= 1) || (b == 22 && k <= 2) || (b == 23 && k <= 5) || (b == 24 && k <= 11)) e = b;
    }
    return e;
}

/* Cross-lane primitives on DPP (row_shr 1/2/4/8 inside a row of 16, row_bcast:15 / :31 across rows): ~6 dependent VALU ops per
 * wave-wide scan or reduction instead of six LDS-crossbar round trips (ds_bpermute, ~60 cycles each).  A lane without a source
 * keeps `old`. */
template <int CTRL, int RM = 0xF> __device__ __forceinline__ int dpp_i(int old, int v) { return __builtin_amdgcn_update_dpp(old, v, CTRL, RM, 0xF, false); }
template <int CTRL, int RM = 0xF> __device__ __forceinline__ float dpp_f(float old, float v)
{ return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(v), CTRL, RM, 0xF, false)); }
#define DPP_SHR1 0x111
#define DPP_SHR2 0x112
#define DPP_SHR4 0x114
#define DPP_SHR8 0x118
#define DPP_BC15 0x142
#define DPP_BC31 0x143
#define DPP_WSHR1 0x138      /* wave_shr:1: lane n takes lane n-1 across row boundaries */
__device__ __forceinline__ int wave_incl_scan_i(int v, int lane)
{
    (void)lane;
    v += dpp_i<DPP_SHR1>(0, v); v += dpp_i<DPP_SHR2>(0, v); v += dpp_i<DPP_SHR4>(0, v); v += dpp_i<DPP_SHR8>(0, v);
    v += dpp_i<DPP_BC15, 0xA>(0, v); v += dpp_i<DPP_BC31, 0xC>(0, v);
    return v;
}
/* reductions return the wave-uniform result (lane 63 of the scan) */
__device__ __forceinline__ int wave_sum_i(int v) { return __builtin_amdgcn_readlane(wave_incl_scan_i(v, 0), 63); }
/* float sum over the wave in DPP tree order (NOT the reference's serial order: only for values that are allowed to be approximate) */
__device__ __forceinline__ float wave_sum_f_tree(float v)
{
    v += dpp_f<DPP_SHR1>(0.0f, v); v += dpp_f<DPP_SHR2>(0.0f, v); v += dpp_f<DPP_SHR4>(0.0f, v); v += dpp_f<DPP_SHR8>(0.0f, v);
    v += dpp_f<DPP_BC15, 0xA>(0.0f, v); v += dpp_f<DPP_BC31, 0xC>(0.0f, v);
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ float wave_max_f(float v)
{
    v = fmaxf(v, dpp_f<DPP_SHR1>(v, v)); v = fmaxf(v, dpp_f<DPP_SHR2>(v, v)); v = fmaxf(v, dpp_f<DPP_SHR4>(v, v)); v = fmaxf(v, dpp_f<DPP_SHR8>(v, v));
    v = fmaxf(v, dpp_f<DPP_BC15, 0xA>(v, v)); v = fmaxf(v, dpp_f<DPP_BC31, 0xC>(v, v));
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ int wave_max_i(int v)
{
    v = imax(v, dpp_i<DPP_SHR1>(v, v)); v = imax(v, dpp_i<DPP_SHR2>(v, v)); v = imax(v, dpp_i<DPP_SHR4>(v, v)); v = imax(v, dpp_i<DPP_SHR8>(v, v));
    v = imax(v, dpp_i<DPP_BC15, 0xA>(v, v)); v = imax(v, dpp_i<DPP_BC31, 0xC>(v, v));
    return __builtin_amdgcn_readlane(v, 63);
}
/* first index of the maximum / minimum (scan order with strict compare): ties resolve to the lowest index.  WIDTH 64: one result,
 * returned uniform in (v, i).  WIDTH 16 / 32: the result of group 0 is returned uniform (lane 15 / 31); ARG32B additionally hands
 * back the second 32-lane group's index. */
#define ARG_STEP(CTRL, RM, CMP) do { const float ov = dpp_f<CTRL, RM>(v, v); const int oi = dpp_i<CTRL, RM>(i, i); \
        const bool t = (ov CMP v) || (ov == v && oi < i); v = t ? ov : v; i = t ? oi : i; } while (0)
template <int WIDTH> __device__ __forceinline__ void wave_argmax_first(float& v, int& i)
{
    ARG_STEP(DPP_SHR1, 0xF, >); ARG_STEP(DPP_SHR2, 0xF, >); ARG_STEP(DPP_SHR4, 0xF, >); ARG_STEP(DPP_SHR8, 0xF, >);
    if (WIDTH >= 32) ARG_STEP(DPP_BC15, 0xA, >);
    if (WIDTH >= 64) ARG_STEP(DPP_BC31, 0xC, >);
    v = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), WIDTH - 1)); i = __builtin_amdgcn_readlane(i, WIDTH - 1);
}
/* two independent 32-lane argmin searches (lanes 0-31 and 32-63); returns both winning indices, uniform */
__device__ __forceinline__ void wave_argmin_first_2x32(float v, int i, int& i_lo, int& i_hi)
{
    ARG_STEP(DPP_SHR1, 0xF, <); ARG_STEP(DPP_SHR2, 0xF, <); ARG_STEP(DPP_SHR4, 0xF, <); ARG_STEP(DPP_SHR8, 0xF, <);
    ARG_STEP(DPP_BC15, 0xA, <);
    i_lo = __builtin_amdgcn_readlane(i, 31); i_hi = __builtin_amdgcn_readlane(i, 63);
}
#undef ARG_STEP

/* ------------------------------------------------------------------------------------------------ */
/* DFT kernels (register resident).  Exact operand order of R/fft/fft_15_16.h and R/fft/fft_2_9.h.   */
/* ------------------------------------------------------------------------------------------------ */
__device__ __forceinline__ void dft16(float* v)
{
    const float S = 7.071067811865475e-1f, C1 = 9.238795325112867e-1f, C3 = 3.826834323650898e-1f;
    const float SP = 2.414213562373095f, SM = 4.142135623730952e-1f;
    float E[16], O[16], P[16], Q[16];
#pragma unroll
    for (int i = 0; i < 16; i++) { E[i] = v[i] + v[i + 16]; O[i] = v[i] - v[i + 16]; }
#pragma unroll
    for (int k = 0; k < 4; k++) {
        P[4 * k + 0] = E[2 * k] + E[2 * k + 8];     P[4 * k + 2] = E[2 * k] - E[2 * k + 8];
        P[4 * k + 1] = E[2 * k + 1] + E[2 * k + 9]; P[4 * k + 3] = E[2 * k + 1] - E[2 * k + 9];
    }
    Q[0] = P[0] + P[8];   Q[4] = P[0] - P[8];   Q[1] = P[1] + P[9];   Q[5] = P[1] - P[9];
    Q[8] = P[2] - P[11];  Q[10] = P[2] + P[11]; Q[9] = P[3] + P[10];  Q[11] = P[3] - P[10];
    Q[2] = P[4] + P[12];  Q[7] = P[4] - P[12];  Q[3] = P[5] + P[13];  Q[6] = P[13] - P[5];
    {
        float a1 = P[6] + P[14], a2 = P[6] - P[14], a0 = P[7] + P[15], a3 = P[7] - P[15];
        Q[12] = (a0 + a2) * S; Q[14] = (a0 - a2) * S; Q[13] = (a3 - a1) * S; Q[15] = (a1 + a3) * -S;
    }
    float g9 = (O[2] + O[14]) * -C3, g10 = (O[2] - O[14]) * C1, g8 = (O[3] + O[15]) * C3, g11 = (O[3] - O[15]) * C1;
    float g5 = (O[4] + O[12]) * -S,  g6 = (O[4] - O[12]) * S,   g4 = (O[5] + O[13]) * S,  g7 = (O[5] - O[13]) * S;
    float g13 = (O[6] + O[10]) * -C1, g14 = (O[6] - O[10]) * C3, g12 = (O[7] + O[11]) * C1, g15 = (O[7] - O[11]) * C3;
    float u2 = g8 * SP - g12 * SM, u3 = g9 * SP - g13 * SM, u4 = g10 * SM - g14 * SP, u5 = g11 * SM - g15 * SP;
    g8 += g12; g9 += g13; g10 += g14; g11 += g15;
    float w6 = O[0] + g4, w10 = O[0] - g4, w7 = O[1] + g5, w11 = O[1] - g5;
    float w12 = g6 - O[9], w14 = g6 + O[9], w13 = O[8] + g7, w15 = O[8] - g7;
    float r10 = w6 - w14, r12 = w6 + w14, r11 = w7 + w15, r13 = w7 - w15;
    float r14 = w10 + w12, r16 = w10 - w12, r15 = w11 + w13, r17 = w11 - w13;
    float h10 = g8 + g10, d10 = g8 - g10, h11 = g9 + g11, d11 = g9 - g11;
    float s12 = u2 + u4, d12 = u2 - u4, s13 = u3 + u5, d13 = u3 - u5;
    v[0] = Q[0] + Q[2];    v[1] = Q[1] + Q[3];    v[2] = r12 + h10;      v[3] = r13 + h11;
    v[4] = Q[10] + Q[12];  v[5] = Q[11] + Q[13];  v[6] = r10 + s12;      v[7] = r11 + s13;
    v[8] = Q[4] - Q[6];    v[9] = Q[5] - Q[7];    v[10] = r16 + d12;     v[11] = r17 + d13;
    v[12] = Q[8] + Q[14];  v[13] = Q[9] + Q[15];  v[14] = r14 + d10;     v[15] = r15 + d11;
    v[16] = Q[0] - Q[2];   v[17] = Q[1] - Q[3];   v[18] = r12 - h10;     v[19] = r13 - h11;
    v[20] = Q[10] - Q[12]; v[21] = Q[11] - Q[13]; v[22] = r10 - s12;     v[23] = r11 - s13;
    v[24] = Q[4] + Q[6];   v[25] = Q[5] + Q[7];   v[26] = r16 - d12;     v[27] = r17 - d13;
    v[28] = Q[8] - Q[14];  v[29] = Q[9] - Q[15];  v[30] = r14 - d10;     v[31] = r15 - d11;
}

__device__ __forceinline__ void dft15(float* v)
{
    float a[2][18], t[2][11];
#pragma unroll
    for (int c = 0; c < 2; c++) {
#define X(k) v[2 * (k) + c]
        a[c][1] = X(1) + X(4);   a[c][2] = X(1) - X(4);   a[c][3] = X(2) + X(8);   a[c][4] = X(2) - X(8);
        a[c][5] = X(3) + X(12);  a[c][6] = X(3) - X(12);  a[c][7] = X(5) + X(10);  a[c][8] = X(5) - X(10);
        a[c][9] = X(6) + X(9);   a[c][10] = X(6) - X(9);  a[c][11] = X(7) + X(13); a[c][12] = X(7) - X(13);
        a[c][13] = X(11) + X(14); a[c][14] = X(11) - X(14);
#undef X
        t[c][1] = a[c][1] + a[c][3];    t[c][2] = a[c][1] - a[c][3];
        t[c][3] = a[c][2] + a[c][14];   t[c][4] = a[c][2] - a[c][14];
        t[c][5] = a[c][4] + a[c][12];   t[c][6] = a[c][4] - a[c][12];
        t[c][7] = a[c][5] + a[c][9];    t[c][8] = a[c][5] - a[c][9];
        t[c][9] = a[c][11] + a[c][13];  t[c][10] = a[c][11] - a[c][13];
    }
    float* r = a[0]; float* i = a[1]; float* tr = t[0]; float* ti = t[1];
    float t28 = tr[2] + tr[10], t29 = ti[2] + ti[10];
    r[4] = tr[1] + tr[9];                 i[4] = ti[1] + ti[9];
    r[3] = mul_d(r[4] + tr[7], -1.25);    i[3] = mul_d(i[4] + ti[7], -1.25);
    r[2] = mul_d(t29 - i[8], -8.660254037844387e-1);
    i[2] = mul_d(t28 - r[8], 8.660254037844387e-1);
    r[1] = r[4] + r[7];                   i[1] = i[4] + i[7];
    r[0] = r[1] + v[0] + tr[7];           i[0] = i[1] + v[1] + ti[7];
    r[7] = tr[2] - tr[10];                i[7] = ti[2] - ti[10];
    r[8] = mul_d(ti[1] - ti[9], -4.841229182759272e-1);
    i[8] = mul_d(tr[1] - tr[9], 4.841229182759272e-1);
    float t0 = tr[3] + r[10], t1 = ti[3] + i[10], t2 = r[6] - tr[5], t3 = i[6] - ti[5];
    r[10] = mul_d(ti[3], -2.308262652881440);  i[10] = mul_d(tr[3], 2.308262652881440);
    r[11] = mul_d(tr[4], 1.332676064001459);   i[11] = mul_d(ti[4], 1.332676064001459);
    r[6] = mul_d(r[7] - tr[8], 5.590169943749475e-1);
    i[6] = mul_d(i[7] - ti[8], 5.590169943749475e-1);
    r[12] = mul_d(t1 + t3, 5.877852522924733e-1);        i[12] = mul_d(t0 + t2, -5.877852522924733e-1);
    r[13] = mul_d(ti[3] - ti[5], -8.816778784387098e-1); i[13] = mul_d(tr[3] - tr[5], 8.816778784387098e-1);
    r[14] = mul_d(tr[4] + tr[6], 5.090369604551274e-1);  i[14] = mul_d(ti[4] + ti[6], 5.090369604551274e-1);
    r[16] = mul_d(ti[5], 5.449068960040204e-1);          i[16] = mul_d(tr[5], -5.449068960040204e-1);
    r[17] = mul_d(tr[6], 3.146021430912046e-1);          i[17] = mul_d(ti[6], 3.146021430912046e-1);
    r[4] = mul_d(r[4], 1.875);   i[4] = mul_d(i[4], 1.875);
    r[1] = mul_d(r[1], -1.5);    i[1] = mul_d(i[1], -1.5);
    r[7] = mul_d(r[7], -8.385254915624212e-1); i[7] = mul_d(i[7], -8.385254915624212e-1);
    r[5] = mul_d(t29, 1.082531754730548);      i[5] = mul_d(t28, -1.082531754730548);
    r[9] = mul_d(t1, 1.538841768587627);       i[9] = mul_d(t0, -1.538841768587627);
    r[15] = mul_d(t3, 3.632712640026803e-1);   i[15] = mul_d(t2, -3.632712640026803e-1);
#pragma unroll
    for (int c = 0; c < 2; c++) {
        float* q = a[c];
        float T2 = q[0] + q[1], T4 = q[3] + q[6], T6 = q[3] - q[6], T8 = q[4] + q[5], T10 = q[4] - q[5];
        float T12 = q[7] + q[8], T14 = q[7] - q[8], T16 = q[13] + q[16], T18 = q[14] + q[17];
        float T20 = q[10] - q[13], T22 = q[11] - q[14], T24 = q[12] + q[15], T26 = q[12] - q[9];
        float o1 = T2 + q[2], o2 = T2 - q[2], o3 = T4 + T26, o4 = T4 - T26, o5 = T6 + T24, o6 = T6 - T24;
        float o7 = T16 + T18, o8 = T16 - T18, o9 = T20 - T22, o10 = T20 + T22;
        float o11 = o1 + T8, o12 = o2 + T10, o13 = o11 - T12, o14 = o12 - T14, o15 = o12 + T14, o16 = o11 + T12;
        float o0 = q[0];
        v[0 + c] = o0;              v[2 + c] = o13 + o5 + o7;    v[4 + c] = o15 + o3 - o9;   v[6 + c] = o0 + o4;
        v[8 + c] = o13 + o6 - o7;   v[10 + c] = o2;              v[12 + c] = o0 + o5;        v[14 + c] = o16 + o3 - o10;
        v[16 + c] = o15 + o4 + o9;  v[18 + c] = o0 + o6;         v[20 + c] = o1;             v[22 + c] = o14 + o5 + o8;
        v[24 + c] = o0 + o3;        v[26 + c] = o16 + o4 + o10;  v[28 + c] = o14 + o6 - o8;
    }
}

__device__ __forceinline__ void dft8(float* v)
{
    const float S = 7.071067811865475e-1f;
    float P[16], Q[16];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        P[4 * k + 0] = v[2 * k] + v[2 * k + 8];     P[4 * k + 2] = v[2 * k] - v[2 * k + 8];
        P[4 * k + 1] = v[2 * k + 1] + v[2 * k + 9]; P[4 * k + 3] = v[2 * k + 1] - v[2 * k + 9];
    }
    Q[0] = P[0] + P[8];   Q[4] = P[0] - P[8];   Q[1] = P[1] + P[9];   Q[5] = P[1] - P[9];
    Q[8] = P[2] - P[11];  Q[10] = P[2] + P[11]; Q[9] = P[3] + P[10];  Q[11] = P[3] - P[10];
    Q[2] = P[4] + P[12];  Q[7] = P[4] - P[12];  Q[3] = P[5] + P[13];  Q[6] = P[13] - P[5];
    float a1 = P[6] + P[14], a2 = P[6] - P[14], a0 = P[7] + P[15], a3 = P[7] - P[15];
    Q[12] = (a0 + a2) * S; Q[14] = (a0 - a2) * S; Q[13] = (a3 - a1) * S; Q[15] = (a1 + a3) * -S;
    v[0] = Q[0] + Q[2];   v[8] = Q[0] - Q[2];    v[1] = Q[1] + Q[3];   v[9] = Q[1] - Q[3];
    v[4] = Q[4] - Q[6];   v[12] = Q[4] + Q[6];   v[5] = Q[5] - Q[7];   v[13] = Q[5] + Q[7];
    v[6] = Q[8] + Q[14];  v[14] = Q[8] - Q[14];  v[7] = Q[9] + Q[15];  v[15] = Q[9] - Q[15];
    v[2] = Q[10] + Q[12]; v[10] = Q[10] - Q[12]; v[3] = Q[11] + Q[13]; v[11] = Q[11] - Q[13];
}
__device__ __forceinline__ void dft3(float* v)
{
    const float C1 = 0.5f, C2 = 0.866025403784439f;
    float r1 = v[0], i1 = v[1];
    float sr = v[2] + v[4], si = v[3] + v[5], dr = v[2] - v[4], di = v[3] - v[5];
    v[0] = r1 + sr;                   v[1] = i1 + si;
    v[2] = r1 - C1 * sr + C2 * di;    v[4] = r1 - C1 * sr - C2 * di;
    v[3] = i1 - C2 * dr - C1 * si;    v[5] = i1 + C2 * dr - C1 * si;
}
__device__ __forceinline__ void dft4(float* v)      /* R/fft/fft_2_9.h:69-92 (forms im(x3) - im(x1), not its negative) */
{
    const float sr = v[0] + v[4], dr = v[0] - v[4], si = v[1] + v[5], di = v[1] - v[5];
    const float tr = v[2] + v[6], ur = v[2] - v[6], ti = v[7] + v[3], ui = v[7] - v[3];
    v[0] = sr + tr; v[1] = si + ti; v[2] = dr - ui; v[3] = di - ur;
    v[4] = sr - tr; v[5] = si - ti; v[6] = dr + ui; v[7] = di + ur;
}
__device__ __forceinline__ void dft5(float* v)
{
    const float C1 = 0.309016994374947f, C2 = 0.951056516295154f, C3 = 0.809016994374947f, C4 = 0.587785252292473f;
    float r1 = v[0], i1 = v[1];
    float a = v[2] + v[8], b = v[2] - v[8], c = v[3] + v[9], d = v[3] - v[9];
    float e = v[4] + v[6], f = v[4] - v[6], g = v[5] + v[7], h = v[5] - v[7];
    v[0] = r1 + a + e;                                v[1] = i1 + c + g;
    v[2] = r1 + C1 * a - C3 * e + C2 * d + C4 * h;    v[8] = r1 + C1 * a - C3 * e - C2 * d - C4 * h;
    v[3] = i1 - C2 * b - C4 * f + C1 * c - C3 * g;    v[9] = i1 + C2 * b + C4 * f + C1 * c - C3 * g;
    v[4] = r1 - C3 * a + C1 * e + C4 * d - C2 * h;    v[6] = r1 - C3 * a + C1 * e - C4 * d + C2 * h;
    v[5] = i1 - C4 * b + C2 * f - C3 * c + C1 * g;    v[7] = i1 + C4 * b - C2 * f - C3 * c + C1 * g;
}

/* ------------------------------------------------------------------------------------------------ */
__device__ __forceinline__ float unif(float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); }
__device__ __forceinline__ float rl_f(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }
__device__ __forceinline__ double rl_d(double v, int l)
{
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}

/* ------------------------------------------------------------------------------------------------ */
/* time-domain analysis                                                                              */
/* ------------------------------------------------------------------------------------------------ */

/* ---- 12.8 kHz resampler + 50 Hz high-pass: R/resamp12k8.c:13-84.  Appends len12 samples to h12. ---- */
template <class LdsT> STAGE void st_resample(const lc3d_plan* __restrict__ P, LdsT& L, int lane, const float* __restrict__ yin /* the frame's HP-filtered 12.8 kHz samples from the pre-kernels, or null */)
{
    const int mlen = PI(rs_mem_in_len), stride = PI(rs_stride), n12 = PI(n12), len12 = PI(len12), N = PI(N);
    const float sf = PF(rs_scale);
    const float* buf = &L.xbuf[MEMCAP - mlen];      /* [mem_in | x] */
    SUB_BEGIN();
    float y[2];
    if (yin) {                                       /* lc3_enc_resample_kernel + lc3_enc_hp50_kernel (lc3_enc_pre.inc) have done the work */
        y[0] = lane < len12 ? yin[lane] : 0.0f; y[1] = lane + 64 < len12 ? yin[lane + 64] : 0.0f;
        (void)buf; (void)stride; (void)n12; (void)N; (void)sf;
    } else {
    /* polyphase FIR, R/resamp12k8.c:48-57: out[n] = sum_m (buf[.]*sf) * lp[.] in the reference's tap order.  The scaled samples
     * are formed once (sm is idle here); a lane's two outputs (n = lane, lane + 64) share one phase, whose taps are held in
     * registers 10 at a time. */
    float* xs = L.sm;
    for (int j = lane; j < mlen + N; j += WAVE) xs[j] = buf[j] * sf;
    LSYNC();
    float d[2] = {0, 0};
    {
        const int T = 240 / stride;
        const int i0 = 15 * lane, r = i0 % stride, start = r ? stride - r : 0;
        const float* tp = &P->rs_taps[start * T];
        const float* b0 = xs + (i0 + start) / stride; const float* b1 = xs + (i0 + 15 * 64 + start) / stride;
        const bool on0 = lane < n12, on1 = lane + 64 < n12;
        if (!on0) b0 = xs;                            /* idle lanes read in bounds */
        if (!on1) b1 = xs;
        float m0 = 0, m1 = 0;
        for (int tb = 0; tb < T; tb += 10) {          /* T = 240 / stride = 10 ... 120; 10 taps at a time keep the stage inside its register budget */
            float tap[10];
#pragma unroll
            for (int m = 0; m < 10; m++) tap[m] = tp[tb + m];
#pragma unroll
            for (int m = 0; m < 10; m++) { m0 += b0[tb + m] * tap[m]; m1 += b1[tb + m] * tap[m]; }
        }
        d[0] = on0 ? m0 : 0.0f; d[1] = on1 ? m1 : 0.0f;
    }
    SUB(0);
    /* biquad in double, strictly serial (R/resamp12k8.c:60-74): the x-only products b_k*x are formed per lane and parked in LDS
     * (A and sm are idle here), the recurrence streams them back with uniform-address reads, one group of four steps ahead of
     * the arithmetic; each output goes back through LDS (a uniform-address store) instead of a per-step lane select */
    const double b0 = lc3t_hp50_b[0], b1 = lc3t_hp50_b[1], b2 = lc3t_hp50_b[2], a1 = lc3t_hp50_a[1], a2 = lc3t_hp50_a[2];
    double* q0 = (double*)L.A; double* q1 = (double*)L.sm; double* q2 = q1 + 128; float* yo = &L.A[256];
    LSYNC();                                          /* xs (in sm) is dead */
#pragma unroll
    for (int h = 0; h < 2; h++) {
        const int n = lane + 64 * h;
        if (n < len12) { const double x = (double)d[h]; q0[n] = b0 * x; q1[n] = b1 * x; q2[n] = b2 * x; }
    }
    LSYNC();
    double u11 = (double)L.fsc[F_HP0], u21 = (double)L.fsc[F_HP1];
    SUB(1);
    {
        double c0[4], c1[4], c2[4];
#pragma unroll
        for (int k = 0; k < 4; k++) { c0[k] = q0[k]; c1[k] = q1[k]; c2[k] = q2[k]; }
        for (int g = 0; g < len12; g += 4) {
            const int gn = g + 4 < len12 ? g + 4 : g;      /* the last group re-reads itself: no branch around the prefetch */
            double n0[4], n1[4], n2[4];
#pragma unroll
            for (int k = 0; k < 4; k++) { n0[k] = q0[gn + k]; n1[k] = q1[gn + k]; n2[k] = q2[gn + k]; }
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const double y1 = c0[k] + u11;
                const double u1 = (c1[k] + u21) - a1 * y1;
                const double u2 = c2[k] - a2 * y1;
                u11 = u1; u21 = u2;
                yo[g + k] = (float)y1;
            }
#pragma unroll
            for (int k = 0; k < 4; k++) { c0[k] = n0[k]; c1[k] = n1[k]; c2[k] = n2[k]; }
        }
    }
    LSYNC();
    y[0] = lane < len12 ? yo[lane] : 0.0f; y[1] = lane + 64 < len12 ? yo[lane + 64] : 0.0f;
    if (lane == 0) { L.fsc[F_HP0] = (float)u11; L.fsc[F_HP1] = (float)u21; }
    }
    SUB(2);
    float keep[6];
#pragma unroll
    for (int k = 0; k < 6; k++) { const int i = lane + 64 * k; keep[k] = (i + len12 < 384) ? L.h12[i + len12] : 0.0f; }
    LSYNC();
#pragma unroll
    for (int k = 0; k < 6; k++) { const int i = lane + 64 * k; if (i + len12 < 384) L.h12[i] = keep[k]; }
    if (lane < len12) L.h12[384 - len12 + lane] = y[0];
    if (lane + 64 < len12) L.h12[384 - len12 + 64 + lane] = y[1];
    LSYNC();
    SUB(3);
}

/* normalised correlation at lag T over acf <= 64 samples (R/olpa.c:104-114): the three serial float sums run in all lanes,
 * the products come back from LDS scratch (A) with uniform-address reads */
/* Serial float sums, one per lane: lane s < nsums adds base[s*stride + 0 .. count-1] in index order (the reference's order) and
 * keeps the total.  A dependent add issues every ~6 cycles whether one lane or all of them run it, so independent sums belong in
 * different lanes rather than in one uniform loop.  Rows are 16-byte aligned and padded to a multiple of 4 readable floats. */
__device__ __forceinline__ float lane_serial_sum(const float* base, int stride, int nsums, int count, int lane)
{
    const float4* p = (const float4*)(base + (lane < nsums ? lane : 0) * stride);
    float acc = 0;
    int i = 0;
    for (; i + 8 <= count; i += 8) {
        const float4 u = p[i >> 2], v = p[(i >> 2) + 1];
        acc += u.x; acc += u.y; acc += u.z; acc += u.w; acc += v.x; acc += v.y; acc += v.z; acc += v.w;
    }
    for (; i < count; i += 4) {                      /* tail: terms past `count` are skipped by selects, not by branches */
        const float4 u = p[i >> 2];
        acc += u.x; acc += (i + 1 < count) ? u.y : 0.0f; acc += (i + 2 < count) ? u.z : 0.0f; acc += (i + 3 < count) ? u.w : 0.0f;
    }
    return acc;
}

/* normalised correlations at lags T_a and T_b over acf <= 64 samples (R/olpa.c:104-114), both at once: five serial sums
 * (ab_a, bb_a, aa, ab_b, bb_b) in five lanes, products parked in LDS scratch (A) */
template <class LdsT> __device__ __forceinline__ void olpa_normcorr2(LdsT& L, const float* s6, int acf, int Ta, int Tb, int lane, float eps, float& nca, float& ncb)
{
    float* pr = L.A;
    if (lane < acf) {
        const float a = s6[lane], ba = s6[lane - Ta], bb = s6[lane - Tb];
        pr[lane] = a * ba; pr[64 + lane] = ba * ba; pr[128 + lane] = a * a; pr[192 + lane] = a * bb; pr[256 + lane] = bb * bb;
    }
    LSYNC();
    const float acc = lane_serial_sum(pr, 64, 5, acf, lane);
    LSYNC();
    const float s_aa = rl_f(acc, 2);
    { float s1 = rl_f(acc, 1) * s_aa; s1 = sqrtf(s1) + eps; const float nc = rl_f(acc, 0) / s1; nca = 0 > nc ? 0 : nc; }
    { float s1 = rl_f(acc, 4) * s_aa; s1 = sqrtf(s1) + eps; const float nc = rl_f(acc, 3) / s1; ncb = 0 > nc ? 0 : nc; }
}

/* ---- open-loop pitch: R/olpa.c:52-143 ---- */
template <class LdsT> STAGE void st_olpa(const lc3d_plan* __restrict__ P, LdsT& L, int lane)
{
    const int len = PI(len12), len2 = len >> 1;
    SUB_BEGIN();
    int acf = len2, back = 0;
    if (PI(dms) == 25) { acf += 16; back = 16; }
    float nd = 0;
    if (lane < len2) {                              /* 2:1 decimation (filter_olpa R/olpa.c:16-31) */
        const float* in12 = &L.h12[384 - len - 27];
        const int i = 4 + 2 * lane;
        float sum = 0;
#pragma unroll
        for (int k = 0; k < 5; k++) sum += lc3t_olpa_dec[k] * in12[i - k];
        nd = sum;
    }
    float keep[4];
#pragma unroll
    for (int k = 0; k < 4; k++) { const int i = lane + 64 * k; keep[k] = (i + len2 < 194) ? L.h6[i + len2] : 0.0f; }
    LSYNC();
#pragma unroll
    for (int k = 0; k < 4; k++) { const int i = lane + 64 * k; if (i + len2 < 194) L.h6[i] = keep[k]; }
    if (lane < len2) L.h6[194 - len2 + lane] = nd;
    LSYNC();
    SUB(25);
    const float* s6 = &L.h6[194 - len2 - back];
    float* R0 = &L.sm[L.MISC];                     /* 98 unweighted autocorrelations */
    const bool two = lane < 34;
    const float* q0 = s6 - (17 + lane); const float* q1 = s6 - (two ? 81 + lane : 17 + lane);
    /* the two ordered sums of a lane advance together in packed fp32 (v_pk_mul_f32 / v_pk_add_f32: two IEEE operations per instruction,
     * each rounded like the scalar one) */
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x2 rr = {0.0f, 0.0f};
#pragma unroll 8
    for (int j = 0; j < acf; j++) { const float a = s6[j]; const f32x2 av = {a, a}, qv = {q0[j], q1[j]}; rr = rr + av * qv; }
    const float r0 = rr.x, r1 = rr.y;
    R0[lane] = r0;
    if (two) R0[64 + lane] = r1;
    float best = r0 * lc3t_olpa_w[lane]; int besti = lane;
    if (two) { const float w1 = r1 * lc3t_olpa_w[64 + lane]; if (w1 > best) { best = w1; besti = 64 + lane; } }
    wave_argmax_first<64>(best, besti);
    int T0 = uni(besti) + 17;
    LSYNC();
    SUB(26);
    const int old = uni(L.isc[I_OLPA_PITCH]);
    const int lo = imax(17, old - 4), hi = imin(114, old + 4), cnt = hi - lo + 1;
    float v = (lane & 15) < cnt ? R0[lo - 17 + (lane & 15)] : -INFINITY; int vi = lane & 15;
    wave_argmax_first<16>(v, vi);
    const int T02 = uni(vi) + lo;
    float nc, nc2;
    olpa_normcorr2(L, s6, acf, T0, T02, lane, PF(c_1em5_a), nc, nc2);
    if (T02 != T0 && (double)nc2 > ((double)nc * 0.85)) { T0 = T02; nc = nc2; }
    if (lane == 0) { L.isc[I_OLPA_PITCH] = T0; L.isc[I_T0] = (int)(T0 * 2.0); L.fsc[F_NC] = nc; }
    LSYNC();
    SUB(27);
}

/* ---- LTPF parameter coder: R/ltpf_coder.c:34-263 ---- */
template <class LdsT> STAGE void st_ltpf(const lc3d_plan* __restrict__ P, const lc3d_chan* __restrict__ C, LdsT& L, int lane)
{
    const int len = PI(len12);                 /* N of the reference = xLen - 1 */
    const float* x = &L.h12[384 - len - 24];
    const int pitch_ol = uni(L.isc[I_T0]); const float ol_nc = unif(L.fsc[F_NC]);
    const int mem_on = uni(L.isc[I_LTPF_ON]);
    const float nc1 = unif(L.fsc[F_LTPF_NC1]), nc2m = unif(L.fsc[F_LTPF_NC2]), mem_pitch = unif(L.fsc[F_LTPF_PITCH]);
    int active = 0, pitch_index = 0, gain = 0;
    float norm_corr = 0, pitch = 0;
    if ((double)ol_nc > 0.6) {
        const int t0_min = imax(pitch_ol - 4, 32), t0_max = imin(pitch_ol + 4, 228);
        int acf = len;
        if (PI(dms) == 25) { acf = 2 * len; x = x - len; }
        const int t_min = t0_min - 4, t_max = t0_max + 4, nl = t_max - t_min + 1;
        float sum1 = 0, sum2 = 0;
        {   /* R/ltpf_coder.c:74-78: two serial sums over acf <= 128 terms, products per lane */
            float* pr = L.A;
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const int j = lane + 64 * h;
                if (j < acf) { const float a = x[j], b = x[j - t_min]; pr[j] = a * a; pr[128 + j] = b * b; }
            }
            LSYNC();
            const float acc = lane_serial_sum(pr, 128, 2, acf, lane);
            sum1 = rl_f(acc, 0); sum2 = rl_f(acc, 1);
            LSYNC();
        }
        float* cor = &L.sm[L.MISC];           /* up to 17 */
        float* cor_int = &L.sm[L.MISC + 32];  /* up to 36 */
        if (lane < nl) {
            const int lag = t_min + lane;
            const float* xl = x - lag;
            float sum = 0;
#pragma unroll 8
            for (int j = 0; j < acf; j++) sum += x[j] * xl[j];
            float s2 = sum2;
            for (int k = t_min + 1; k <= lag; k++) s2 = s2 + x[-k] * x[-k] - x[acf - 1 - (k - 1)] * x[acf - 1 - (k - 1)];
            const float sum3 = sqrtf(sum1 * s2) + PF(c_1em5_b);
            float nc = sum / sum3;
            nc = 0 > nc ? 0 : nc;
            cor[lane] = nc;
        }
        LSYNC();
        int tsel;
        /* searchMaxIndice of R/ltpf_coder.c:15-32 starts from max = 0 and takes strictly greater values: a NaN correlation (the
         * running energy can round below zero on an all-zero history) is never taken - here it is mapped to -inf first */
        { float v = lane < 16 && lane < t_max - t_min - 8 + 1 ? cor[4 + lane] : -INFINITY; v = v == v ? v : -INFINITY; int vi = lane & 15; wave_argmax_first<16>(v, vi);
          tsel = unif(v) > 0 ? uni(vi) : 0; }
        const int t1 = tsel + t0_min;
        int pitch_int, pitch_fr;
        if (t1 >= 157) { pitch_int = t1; pitch_fr = 0; }
        else {
            const int nint = 4 * (t0_max - t0_min + 1);
            if (lane < nint) {
                /* cor_up is cor zero-stuffed by 4; its zero taps only add +-0 to the accumulator: skipped */
                float sum = 0;
                for (int k = (4 - (lane & 3)) & 3; k < 32; k += 4) {
                    const int m = (lane + k) >> 2;
                    if (m <= t_max - t_min) sum += cor[m] * lc3t_ltpf_int4[k];
                }
                cor_int[lane] = sum;
            }
            LSYNC();
            const int step = t1 >= 127 ? 2 : 1;
            const int mid = 4 * (t1 - t0_min) + 1, up = 4 - step, down = t1 == t0_min ? 0 : 4 - step;
            const int cnt = ((mid + up) - (mid - down)) / step + 1;
            int ksel;
            { float v = (lane & 15) < cnt ? cor_int[mid - down - 1 + (lane & 15) * step] : -INFINITY; v = v == v ? v : -INFINITY; int vi = lane & 15; wave_argmax_first<16>(v, vi);
              ksel = unif(v) > 0 ? uni(vi) : 0; }
            pitch_fr = ksel * step - down;
            if (pitch_fr >= 0) pitch_int = t1; else { pitch_int = t1 - 1; pitch_fr = 4 + pitch_fr; }
        }
        if (pitch_int < 127) pitch_index = pitch_int * 4 + pitch_fr - 128;
        else if (pitch_int < 157) pitch_index = pitch_int * 2 + (pitch_fr / 2) - 254 + 380;
        else pitch_index = pitch_int - 157 + 380 + 60;
        pitch = (float)((double)(float)pitch_int + (double)(float)pitch_fr / 4.0);
        const float* f0 = &lc3t_ltpf_frac[0]; const float* fp = &lc3t_ltpf_frac[4 * pitch_fr];
        float a = 0, b = 0, c = 0;
        {
            float* pq = L.A;                        /* 3 x 128 products */
#pragma unroll
            for (int h = 0; h < 2; h++) {           /* R/ltpf_coder.c:190-216 */
                const int n = lane + 64 * h;
                if (n < acf) {
                    const float cu = x[n + 1] * f0[0] + x[n] * f0[1] + x[n - 1] * f0[2];
                    const float pr = x[n - pitch_int + 1] * fp[0] + x[n - pitch_int] * fp[1] + x[n - pitch_int - 1] * fp[2] + x[n - pitch_int - 2] * fp[3];
                    pq[n] = cu * pr; pq[128 + n] = cu * cu; pq[256 + n] = pr * pr;
                }
            }
            LSYNC();
            const float acc = lane_serial_sum(pq, 128, 3, acf, lane);
            a = rl_f(acc, 0); b = rl_f(acc, 1); c = rl_f(acc, 2);
        }
        b = sqrtf(b * c) + PF(c_1em5_b);
        norm_corr = a / b;
        { const float lo = -1 > norm_corr ? -1 : norm_corr; norm_corr = 1 < lo ? 1 : lo; }
        if (norm_corr < 0) norm_corr = 0;
        if (CI(ltpf_enable) == 1) {
            if ((mem_on == 0 && (PI(dms) == 100 || (double)nc2m > 0.94) && (double)nc1 > 0.94 && (double)norm_corr > 0.94) ||
                (mem_on == 1 && (double)norm_corr > 0.9) ||
                (mem_on == 1 && fabsf(pitch - mem_pitch) < 2 && (double)(norm_corr - nc1) > -0.1 && (double)norm_corr > 0.84))
                active = 1;
        }
        gain = 4;
    } else { gain = 0; norm_corr = ol_nc; pitch = 0; }
    LSYNC();
    if (lane == 0) {
        if (gain > 0) { L.isc[I_LTPF0] = 1; L.isc[I_LTPF1] = active; L.isc[I_LTPF2] = pitch_index; L.isc[I_LTPF_BITS] = 11; }
        else { L.isc[I_LTPF0] = 0; L.isc[I_LTPF1] = 0; L.isc[I_LTPF2] = 0; L.isc[I_LTPF_BITS] = 1; }
        if (PI(dms) < 100) L.fsc[F_LTPF_NC2] = nc1;
        L.fsc[F_LTPF_NC1] = norm_corr; L.isc[I_LTPF_ON] = active; L.fsc[F_LTPF_PITCH] = pitch;
    }
    LSYNC();
}

/* ---- attack detector: R/attack_detector.c:13-104 (only when attack_handling) ---- */
/* first half (:26-77): decimation to 16 kHz, high-pass with the filter memory (m0, m1) of the previous frame, block energies: lane b < nb
 * returns block b's energy; (nm0, nm1) is the filter memory this frame leaves.  Stateless given the previous frame's last samples. */
__device__ __forceinline__ float attack_energies_at(const int nb, const int fs_hz, const float* in /* the frame's PCM */, float* scr /* 362 floats of scratch */, int lane, float m0, float m1, float& nm0, float& nm1)
{
    const int n16 = nb * 40;
    float* p = &scr[2];
    for (int j = lane; j < n16; j += WAVE) {
        float v;
        if (fs_hz == 96000) { const float* q = &in[6 * j]; v = q[0] + q[1] + q[2] + q[3] + q[4] + q[5]; }
        else if (fs_hz == 48000) { const float* q = &in[3 * j]; v = (q[0] + q[1] + q[2]); }
        else if (fs_hz == 32000) { const float* q = &in[2 * j]; v = (q[0] + q[1]); }
        else { const float* q = &in[3 * j]; v = (float)((double)q[0] + ((double)(q[1] + q[2])) / 2.0); }
        p[j] = v;
    }
    if (lane == 0) { p[-2] = m0; p[-1] = m1; }
    LSYNC();
    nm0 = p[n16 - 2]; nm1 = p[n16 - 1];
    float* fs = &scr[200];
    for (int i = lane; i < 160; i += WAVE) {
        float t = 0;
        t = (float)((double)t + (double)p[i] * 0.375);
        t = (float)((double)t + (double)p[i - 1] * (-0.5));
        t = (float)((double)t + (double)p[i - 2] * (0.125));
        fs[i] = t;
    }
    LSYNC();
    float e = 0;
    if (lane < nb) { for (int k = 0; k < 40; k++) { const float v = fs[k + lane * 40]; e += v * v; } }
    return e;
}
template <class LdsT> __device__ __forceinline__ float attack_energies(LdsT& L, int lane, float m0, float m1, float& nm0, float& nm1)
{
    return attack_energies_at(PI(att_nblocks), PI(fs), XCUR(L), L.A, lane, m0, m1, nm0, nm1);
}
/* second half (:79-102): the decision over the blocks, sequential from frame to frame through (acc, last position) */
__device__ __forceinline__ void attack_decide(float e0, float e1, float e2, float e3, int nb, float mval, int hang, float& acc, int& last_pos, int& flag)
{
    const float ev[4] = {e0, e1, e2, e3};
    flag = 0; int pos = -1;
#pragma unroll
    for (int b = 0; b < 4; b++) {
        if (b < nb) {
            const float nrg = ev[b];
            const float t = (float)((double)nrg / 8.5);
            if (t > (acc > mval ? acc : mval)) { flag = 1; pos = b + 1; }
            const double q = 0.25 * (double)acc;
            acc = (double)nrg > q ? nrg : (float)q;
        }
    }
    if (last_pos > hang) flag = 1;
    last_pos = pos;
}
template <class LdsT> STAGE void st_attack(const lc3d_plan* __restrict__ P, LdsT& L, int lane)
{
    float nm0, nm1;
    const float e = attack_energies(L, lane, unif(L.fsc[F_ATT_M0]), unif(L.fsc[F_ATT_M1]), nm0, nm1);
    float acc = unif(L.fsc[F_ATT_ACC]); int pos = uni(L.isc[I_ATT_POS]), flag;
    attack_decide(rl_f(e, 0), rl_f(e, 1), rl_f(e, 2), rl_f(e, 3), PI(att_nblocks), PI(fs) == 96000 ? 1e-5f : 0.0f, PI(att_hang), acc, pos, flag);
    LSYNC();
    if (lane == 0) { L.fsc[F_ATT_M0] = nm0; L.fsc[F_ATT_M1] = nm1; L.fsc[F_ATT_ACC] = acc; L.isc[I_ATT_FLAG] = flag; L.isc[I_ATT_POS] = pos; }
    LSYNC();
}

/* ------------------------------------------------------------------------------------------------ */
/* MDCT: R/mdct.c:103-124 + R/dct4.c:75-95 + R/fft/fft_240_480.h:16-88 / R/fft/fft_generic.h:634-699  */
/* ------------------------------------------------------------------------------------------------ */
/* One half of the 16-point kernel (R/fft/fft_15_16.h:214-401): ODD = false produces the 8 even-indexed outputs from the sums
 * E[i] = v[i] + v[i+16], ODD = true the 8 odd-indexed outputs from the differences O[i].  Operation for operation identical to
 * dft16(); splitting it lets two lanes share one transform (half the registers, half the latency). */
__device__ __forceinline__ void dft16_oddblock(const float* O, float* o);
template <bool ODD> __device__ __forceinline__ void dft16_half(const float* v, float* o /* 8 complex outputs: k = ODD + 2j */)
{
    const float S = 7.071067811865475e-1f;
    if (!ODD) {
        float E[16], P[16], Q[16];
#pragma unroll
        for (int i = 0; i < 16; i++) E[i] = v[i] + v[i + 16];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            P[4 * k + 0] = E[2 * k] + E[2 * k + 8];     P[4 * k + 2] = E[2 * k] - E[2 * k + 8];
            P[4 * k + 1] = E[2 * k + 1] + E[2 * k + 9]; P[4 * k + 3] = E[2 * k + 1] - E[2 * k + 9];
        }
        Q[0] = P[0] + P[8];   Q[4] = P[0] - P[8];   Q[1] = P[1] + P[9];   Q[5] = P[1] - P[9];
        Q[8] = P[2] - P[11];  Q[10] = P[2] + P[11]; Q[9] = P[3] + P[10];  Q[11] = P[3] - P[10];
        Q[2] = P[4] + P[12];  Q[7] = P[4] - P[12];  Q[3] = P[5] + P[13];  Q[6] = P[13] - P[5];
        const float a1 = P[6] + P[14], a2 = P[6] - P[14], a0 = P[7] + P[15], a3 = P[7] - P[15];
        Q[12] = (a0 + a2) * S; Q[14] = (a0 - a2) * S; Q[13] = (a3 - a1) * S; Q[15] = (a1 + a3) * -S;
        o[0] = Q[0] + Q[2];    o[1] = Q[1] + Q[3];      /* k = 0  */
        o[2] = Q[10] + Q[12];  o[3] = Q[11] + Q[13];    /* k = 2  */
        o[4] = Q[4] - Q[6];    o[5] = Q[5] - Q[7];      /* k = 4  */
        o[6] = Q[8] + Q[14];   o[7] = Q[9] + Q[15];     /* k = 6  */
        o[8] = Q[0] - Q[2];    o[9] = Q[1] - Q[3];      /* k = 8  */
        o[10] = Q[10] - Q[12]; o[11] = Q[11] - Q[13];   /* k = 10 */
        o[12] = Q[4] + Q[6];   o[13] = Q[5] + Q[7];     /* k = 12 */
        o[14] = Q[8] - Q[14];  o[15] = Q[9] - Q[15];    /* k = 14 */
    } else {
        float O[16];
#pragma unroll
        for (int i = 0; i < 16; i++) O[i] = v[i] - v[i + 16];
        dft16_oddblock(O, o);
    }
}
/* odd half of the 16-point kernel on 8 complex differences O (R/fft/fft_15_16.h:289-372; the same block is the "fft8even" part of
 * the 32-point kernel, R/fft/fft_32.h:330-412): o[m] = bin 2m+1 */
__device__ __forceinline__ void dft16_oddblock(const float* O, float* o)
{
    const float S = 7.071067811865475e-1f, C1 = 9.238795325112867e-1f, C3 = 3.826834323650898e-1f;
    const float SP = 2.414213562373095f, SM = 4.142135623730952e-1f;
    {
        float g9 = (O[2] + O[14]) * -C3, g10 = (O[2] - O[14]) * C1, g8 = (O[3] + O[15]) * C3, g11 = (O[3] - O[15]) * C1;
        const float g5 = (O[4] + O[12]) * -S,  g6 = (O[4] - O[12]) * S,   g4 = (O[5] + O[13]) * S,  g7 = (O[5] - O[13]) * S;
        const float g13 = (O[6] + O[10]) * -C1, g14 = (O[6] - O[10]) * C3, g12 = (O[7] + O[11]) * C1, g15 = (O[7] - O[11]) * C3;
        const float u2 = g8 * SP - g12 * SM, u3 = g9 * SP - g13 * SM, u4 = g10 * SM - g14 * SP, u5 = g11 * SM - g15 * SP;
        g8 += g12; g9 += g13; g10 += g14; g11 += g15;
        const float w6 = O[0] + g4, w10 = O[0] - g4, w7 = O[1] + g5, w11 = O[1] - g5;
        const float w12 = g6 - O[9], w14 = g6 + O[9], w13 = O[8] + g7, w15 = O[8] - g7;
        const float r10 = w6 - w14, r12 = w6 + w14, r11 = w7 + w15, r13 = w7 - w15;
        const float r14 = w10 + w12, r16 = w10 - w12, r15 = w11 + w13, r17 = w11 - w13;
        const float h10 = g8 + g10, d10 = g8 - g10, h11 = g9 + g11, d11 = g9 - g11;
        const float s12 = u2 + u4, d12 = u2 - u4, s13 = u3 + u5, d13 = u3 - u5;
        o[0] = r12 + h10;  o[1] = r13 + h11;    /* k = 1  */
        o[2] = r10 + s12;  o[3] = r11 + s13;    /* k = 3  */
        o[4] = r16 + d12;  o[5] = r17 + d13;    /* k = 5  */
        o[6] = r14 + d10;  o[7] = r15 + d11;    /* k = 7  */
        o[8] = r12 - h10;  o[9] = r13 - h11;    /* k = 9  */
        o[10] = r10 - s12; o[11] = r11 - s13;   /* k = 11 */
        o[12] = r16 - d12; o[13] = r17 - d13;   /* k = 13 */
        o[14] = r14 - d10; o[15] = r15 - d11;   /* k = 15 */
    }
}

template <class LdsT> STAGE void mdct_dft240_cols(LdsT& L, int lane)   /* 15 transforms of length 16, in place in X; two lanes per transform */
{
    float* X = XCUR(L);
    const int col = lane >> 1, odd = lane & 1;
    float o[16];
    if (lane < 30) {
        float v[32];
#pragma unroll
        for (int l = 0; l < 16; l++) { const int s = (225 * l + 16 * col) % 240; v[2 * l] = X[2 * s]; v[2 * l + 1] = X[2 * s + 1]; }
        if (odd) dft16_half<true>(v, o); else dft16_half<false>(v, o);
    }
    LSYNC();
    if (lane < 30) {
#pragma unroll
        for (int j = 0; j < 8; j++) { const int l = odd + 2 * j, s = (225 * l + 16 * col) % 240; X[2 * s] = o[2 * j]; X[2 * s + 1] = o[2 * j + 1]; }
    }
    LSYNC();
}
template <class LdsT> STAGE void mdct_dft240_rows(LdsT& L, int lane)   /* 16 transforms of length 15, X -> A in natural order */
{
    const float* X = XCUR(L);
    if (lane < 16) {
        float v[30];
#pragma unroll
        for (int l = 0; l < 15; l++) { const int s = (225 * lane + 16 * l) % 240; v[2 * l] = X[2 * s]; v[2 * l + 1] = X[2 * s + 1]; }
        dft15(v);
#pragma unroll
        for (int l = 0; l < 15; l++) { const int d = (15 * lane + 16 * l) % 240; L.A[2 * d] = v[2 * l]; L.A[2 * d + 1] = v[2 * l + 1]; }
    }
    LSYNC();
}
/* Prime-factor DFT of N/2 in {10, 20, 30, 40, 80, 120, 160} (R/fft/fft_generic.h:634-699 pfaDFT with fft_n leaves): two or three
 * stages of small DFTs whose gather maps the host derived by running the reference's index logic on slot labels.  One lane per
 * small transform; a stage reads its inputs, the wave synchronises, then writes consecutive slots, so it may work in place.
 * X -> (X ->) A, the last stage scattering to natural order. */
/* 32-point kernel (R/fft/fft_32.h:16-467) split over two lanes: with s_k = x_k + x_{k+16}, d_k = x_k - x_{k+16}, the even bins are the
 * 16-point kernel on s (lane 0 of the pair); odd bin 2m+1 = T_m + F_m and bin 2m+17 = T_m - F_m, T = odd block of the 16-point kernel on
 * (d_0, d_2, .., d_14), F = the 4x4 rotations of (d_1, d_3, .., d_15) (lane 1).  sd: 16 complex sums or differences; o: 16 complex bins. */
template <bool ODD> __device__ __forceinline__ void dft32_half(float* sd, float* o)
{
    if (!ODD) {
        dft16(sd);
#pragma unroll
        for (int i = 0; i < 32; i++) o[i] = sd[i];
    } else {
        const float c0 = 9.807852804032304e-1f, c1 = 8.314696123025452e-1f, c2 = 5.555702330196023e-1f, c3 = 1.950903220161283e-1f;
        float de[16], T[16], A[4], B[4], C[4], D[4];
#pragma unroll
        for (int j = 0; j < 8; j++) { de[2 * j] = sd[4 * j]; de[2 * j + 1] = sd[4 * j + 1]; }
#pragma unroll
        for (int j = 0; j < 4; j++) {               /* pairs (d_{2j+1}, d_{15-2j}) */
            const float er = sd[4 * j + 2], ei = sd[4 * j + 3], fr = sd[4 * (7 - j) + 2], fi = sd[4 * (7 - j) + 3];
            B[j] = -(er + fr); C[j] = er - fr; A[j] = ei + fi; D[j] = ei - fi;
        }
        const float a0 = A[0] * c3 + A[1] * c2 + A[2] * c1 + A[3] * c0, a1 = A[0] * c2 + A[1] * c0 + A[2] * c3 - A[3] * c1;
        const float a2 = A[0] * c1 + A[1] * c3 - A[2] * c0 + A[3] * c2, a3 = A[0] * c0 - A[1] * c1 + A[2] * c2 - A[3] * c3;
        const float b0 = B[0] * c3 + B[1] * c2 + B[2] * c1 + B[3] * c0, b1 = B[0] * c2 + B[1] * c0 + B[2] * c3 - B[3] * c1;
        const float b2 = B[0] * c1 + B[1] * c3 - B[2] * c0 + B[3] * c2, b3 = B[0] * c0 - B[1] * c1 + B[2] * c2 - B[3] * c3;
        const float g0 = C[0] * c0 + C[1] * c1 + C[2] * c2 + C[3] * c3, g1 = C[0] * c1 - C[1] * c3 - C[2] * c0 - C[3] * c2;
        const float g2 = C[0] * c2 - C[1] * c0 + C[2] * c3 + C[3] * c1, g3 = C[0] * c3 - C[1] * c2 + C[2] * c1 - C[3] * c0;
        const float h0 = D[0] * c0 + D[1] * c1 + D[2] * c2 + D[3] * c3, h1 = D[0] * c1 - D[1] * c3 - D[2] * c0 - D[3] * c2;
        const float h2 = D[0] * c2 - D[1] * c0 + D[2] * c3 + D[3] * c1, h3 = D[0] * c3 - D[1] * c2 + D[2] * c1 - D[3] * c0;
        float F[16];
        F[0] = a0 + g0;  F[1] = b0 + h0;   F[14] = a0 - g0; F[15] = b0 - h0;
        F[2] = a1 + g1;  F[3] = b1 + h1;   F[12] = a1 - g1; F[13] = b1 - h1;
        F[4] = a2 + g2;  F[5] = b2 + h2;   F[10] = a2 - g2; F[11] = b2 - h2;
        F[6] = a3 + g3;  F[7] = b3 + h3;   F[8] = a3 - g3;  F[9] = b3 - h3;
        dft16_oddblock(de, T);
#pragma unroll
        for (int m = 0; m < 8; m++) {
            o[2 * m] = T[2 * m] + F[2 * m];        o[2 * m + 1] = T[2 * m + 1] + F[2 * m + 1];          /* bin 2m+1  */
            o[16 + 2 * m] = T[2 * m] - F[2 * m];   o[16 + 2 * m + 1] = T[2 * m + 1] - F[2 * m + 1];     /* bin 2m+17 */
        }
    }
}
/* first stage of the 160-point prime-factor DFT (32 kHz / 10 ms): five 32-point transforms, two lanes each; in place with a barrier.
 * Its own function: it needs more registers than every other DFT stage. */
template <class LdsT> STAGE void mdct_dft160_stage1(const lc3d_plan* __restrict__ P, LdsT& L, int lane)
{
    float* X = XCUR(L);
    const uint8_t* map = P->pfa_src;
    const bool on = lane < 10;
    const int t = on ? lane >> 1 : 0, half = lane & 1;
    float sd[32], o[32];
#pragma unroll
    for (int k = 0; k < 16; k++) {
        const int sa = map[t * 32 + k], sb = map[t * 32 + k + 16];
        const float ar = X[2 * sa], ai = X[2 * sa + 1], br = X[2 * sb], bi = X[2 * sb + 1];
        sd[2 * k] = half ? ar - br : ar + br; sd[2 * k + 1] = half ? ai - bi : ai + bi;
    }
    if (half) dft32_half<true>(sd, o); else dft32_half<false>(sd, o);
    LSYNC();
    if (on) {
        /* even lane: o[m] = bin 2m; odd lane: o[m] = bin 2m+1 (m < 8), o[8+m] = bin 2m+17 */
#pragma unroll
        for (int m = 0; m < 16; m++) {
            const int bin = half ? (m < 8 ? 2 * m + 1 : 2 * (m - 8) + 17) : 2 * m;
            const int d = t * 32 + bin;
            X[2 * d] = o[2 * m]; X[2 * d + 1] = o[2 * m + 1];
        }
    }
    LSYNC();
}
#ifdef LC3_BIG
/* 480 = 15 x 32 Good-Thomas (R/fft/fft_240_480.h:90-185): index tables table1[k + 15 l] = (256 k + 225 l) mod 480 for the fifteen
 * 32-point column transforms (two lanes each, in place), the same table as (225 k + 256 l) mod 480 for the thirty-two 15-point row
 * transforms, output table2[15 k + l] = (15 k + 32 l) mod 480.  X -> A. */
template <class LdsT> STAGE void mdct_dft480_cols(LdsT& L, int lane)
{
    float* X = XCUR(L);
    const bool on = lane < 30;
    const int k = on ? lane >> 1 : 0, half = lane & 1;
    float sd[32], o[32];
#pragma unroll
    for (int l = 0; l < 16; l++) {
        const int sa = (256 * k + 225 * l) % 480, sb = (256 * k + 225 * (l + 16)) % 480;
        const float ar = X[2 * sa], ai = X[2 * sa + 1], br = X[2 * sb], bi = X[2 * sb + 1];
        sd[2 * l] = half ? ar - br : ar + br; sd[2 * l + 1] = half ? ai - bi : ai + bi;
    }
    if (half) dft32_half<true>(sd, o); else dft32_half<false>(sd, o);
    LSYNC();
    if (on) {
#pragma unroll
        for (int m = 0; m < 16; m++) {
            const int bin = half ? (m < 8 ? 2 * m + 1 : 2 * (m - 8) + 17) : 2 * m;
            const int d = (256 * k + 225 * bin) % 480;
            X[2 * d] = o[2 * m]; X[2 * d + 1] = o[2 * m + 1];
        }
    }
    LSYNC();
}
template <class LdsT> STAGE void mdct_dft480_rows(LdsT& L, int lane)
{
    const float* X = XCUR(L);
    if (lane < 32) {
        float v[30];
#pragma unroll
        for (int l = 0; l < 15; l++) { const int s = (225 * lane + 256 * l) % 480; v[2 * l] = X[2 * s]; v[2 * l + 1] = X[2 * s + 1]; }
        dft15(v);
#pragma unroll
        for (int l = 0; l < 15; l++) { const int d = (15 * lane + 32 * l) % 480; L.A[2 * d] = v[2 * l]; L.A[2 * d + 1] = v[2 * l + 1]; }
    }
    LSYNC();
}
#endif
template <int RAD> __device__ __forceinline__ void pfa_stage_r(const uint8_t* __restrict__ map, const uint8_t* __restrict__ dst, const float* in, float* out, int cnt, int lane)
{
    float v[2 * RAD];
    const bool on = lane < cnt;
    const int base = on ? lane * RAD : 0;
#pragma unroll
    for (int j = 0; j < RAD; j++) { const int s = map[base + j]; v[2 * j] = in[2 * s]; v[2 * j + 1] = in[2 * s + 1]; }
    if (RAD == 2) { const float r1 = v[0], i1 = v[1], r2 = v[2], i2 = v[3]; v[0] = r1 + r2; v[1] = i1 + i2; v[2] = r1 - r2; v[3] = i1 - i2; }   /* R/fft/fft_2_9.h:22-37 */
    else if (RAD == 3) dft3(v);
    else if (RAD == 4) dft4(v);
    else if (RAD == 5) dft5(v);
    else if (RAD == 8) dft8(v);
    else dft16(v);
    LSYNC();
    if (on) {
#pragma unroll
        for (int j = 0; j < RAD; j++) { const int d = dst ? dst[base + j] : base + j; out[2 * d] = v[2 * j]; out[2 * d + 1] = v[2 * j + 1]; }
    }
    LSYNC();
}
__device__ __forceinline__ void pfa_stage(const uint8_t* __restrict__ map, const uint8_t* __restrict__ dst, const float* in, float* out, int rad, int cnt, int lane)
{
    switch (rad) {                                  /* wave-uniform */
    case 2: pfa_stage_r<2>(map, dst, in, out, cnt, lane); break;
    case 3: pfa_stage_r<3>(map, dst, in, out, cnt, lane); break;
    case 4: pfa_stage_r<4>(map, dst, in, out, cnt, lane); break;
    case 5: pfa_stage_r<5>(map, dst, in, out, cnt, lane); break;
    default: pfa_stage_r<8>(map, dst, in, out, cnt, lane); break;
    }
}
/* first stage of the 80-point prime-factor DFT (16 x 5): its own function, so that the 16-point kernel's register needs (and the
 * callee-saved spills they cause) stay out of the other lengths' path */
template <class LdsT> STAGE void mdct_dft80_stage1(const lc3d_plan* __restrict__ P, LdsT& L, int lane)
{
    float* X = XCUR(L);
    pfa_stage_r<16>(P->pfa_src, nullptr, X, X, 5, lane);
}
template <class LdsT> STAGE void mdct_dft_pfa(const lc3d_plan* __restrict__ P, LdsT& L, int lane)
{
    float* X = XCUR(L);
    const int len = PI(N) >> 1, nst = PI(pfa_nst);
    const int r0 = PI(pfa_rad[0]), r1 = PI(pfa_rad[1]), r2 = PI(pfa_rad[2]);
    if (nst == 3) {
        pfa_stage(P->pfa_src, nullptr, X, L.A, r0, len / r0, lane);
        pfa_stage(P->pfa_src + LC3D_PFA_STRIDE, nullptr, L.A, X, r1, len / r1, lane);
        pfa_stage(P->pfa_src + 2 * LC3D_PFA_STRIDE, P->pfa_dst, X, L.A, r2, len / r2, lane);
    } else {
        if (r0 < 16) pfa_stage(P->pfa_src, nullptr, X, X, r0, len / r0, lane);       /* 16, 32: mdct_dft80_stage1 / mdct_dft160_stage1 have run */
        pfa_stage(P->pfa_src + LC3D_PFA_STRIDE, P->pfa_dst, X, L.A, r1, len / r1, lane);
    }
}

/* 60 = 4 x 15 Good-Thomas (R/fft/fft_60_128.h:16-66): four 15-point transforms in place over the map (45k + 16l) % 60, then
 * fifteen 4-point transforms scattering to (15k + 4l) % 60.  X -> A. */
template <class LdsT> STAGE void mdct_dft60(const lc3d_plan* __restrict__ P, LdsT& L, int lane)
{
    float* X = XCUR(L);
    const uint8_t* m = P->pfa_src;                  /* m[k + 4 l] = (45k + 16l) % 60 */
    {
        float v[30];
        const int k = lane < 4 ? lane : 0;
#pragma unroll
        for (int l = 0; l < 15; l++) { const int s = m[k + 4 * l]; v[2 * l] = X[2 * s]; v[2 * l + 1] = X[2 * s + 1]; }
        dft15(v);
        LSYNC();
        if (lane < 4) {
#pragma unroll
            for (int l = 0; l < 15; l++) { const int s = m[k + 4 * l]; X[2 * s] = v[2 * l]; X[2 * s + 1] = v[2 * l + 1]; }
        }
        LSYNC();
    }
    {
        float v[8];
        const int l = lane < 15 ? lane : 0;
#pragma unroll
        for (int k = 0; k < 4; k++) { const int s = m[k + 4 * l]; v[2 * k] = X[2 * s]; v[2 * k + 1] = X[2 * s + 1]; }
        dft4(v);
        if (lane < 15) {
#pragma unroll
            for (int k = 0; k < 4; k++) { const int d = P->pfa_dst[k + 4 * l]; L.A[2 * d] = v[2 * k]; L.A[2 * d + 1] = v[2 * k + 1]; }
        }
        LSYNC();
    }
}

template <class LdsT> STAGE void mdct_pre(const lc3d_plan* __restrict__ P, LdsT& L, int lane)
{
    const int N = PI(N), h = N >> 1, la = PI(la), ml = N - la;
    const float* w = &lc3t_win_pool[PI(win_off)];
    const float* t = &L.xbuf[MEMCAP - ml];          /* t[j] = [memory | frame], j < 2N-la ; zero beyond */
    float* X = XCUR(L);
    const int lim = 2 * N - la;
    constexpr int NK = (MAXN / 2 + WAVE - 1) / WAVE;    /* unrolled with a compile-time bound: the window / twiddle loads of all rounds are in flight together */
#pragma unroll
    for (int k = 0; k < NK; k++) {                  /* window + fold (R/mdct.c:113-119) -> A */
        const int i = lane + 64 * k;
        if (i < h) {
            const int j0 = 3 * h - i - 1, j1 = 3 * h + i, j2 = i, j3 = 2 * h - i - 1;
            const float a0 = (j0 < lim ? t[j0] : 0.0f) * w[j0];
            const float a1 = (j1 < lim ? t[j1] : 0.0f) * w[j1];
            const float a2 = t[j2] * w[j2];
            const float a3 = t[j3] * w[j3];
            L.A[i] = -a0 - a1;
            L.A[h + i] = a2 - a3;
        }
    }
    LSYNC();
    /* the frame's tail becomes the next frame's MDCT / resampler memory; the frame half of xbuf (X) is scratch from here on */
    for (int i = lane; i < ml; i += WAVE) L.xbuf[MEMCAP - ml + i] = L.xbuf[MEMCAP + N - ml + i];
    LSYNC();
#pragma unroll
    for (int k = 0; k < NK; k++) {                  /* pre-twiddle R/dct4.c:84-86: A -> X */
        const int i = lane + 64 * k;
        if (i < h) {
            const float ar = L.A[2 * i], ai = L.A[N - 2 * i - 1], br = P->tw1[2 * i], bi = P->tw1[2 * i + 1];
            X[2 * i] = ar * br - ai * bi;
            X[2 * i + 1] = ai * br + ar * bi;
        }
    }
    LSYNC();
}
/* post-twiddle (leaf stage) */
template <class LdsT> STAGE void mdct_post(const lc3d_plan* __restrict__ P, LdsT& L, int lane)
{
    const int N = PI(N), h = N >> 1;
    const float norm = PF(dct4_norm);
    constexpr int NK = (MAXN / 2 + WAVE - 1) / WAVE;
    float o0[NK], o1[NK];                            /* post-twiddle R/dct4.c:90-94, in place in A through registers */
#pragma unroll
    for (int k = 0; k < NK; k++) {
        const int i = lane + 64 * k;
        if (i < h) {
            const float ar = L.A[2 * i], ai = L.A[2 * i + 1], br = P->tw2[2 * i], bi = P->tw2[2 * i + 1];
            const float tr = ar * br - ai * bi, ti = ai * br + ar * bi;
            o0[k] = tr * norm; o1[k] = -ti * norm;
        }
    }
    LSYNC();
#pragma unroll
    for (int k = 0; k < NK; k++) {
        const int i = lane + 64 * k;
        if (i < h) { L.A[2 * i] = o0[k]; L.A[N - 2 * i - 1] = o1[k]; }
    }
    LSYNC();
}

/* ------------------------------------------------------------------------------------------------ */
/* spectral shaping                                                                                  */
/* ------------------------------------------------------------------------------------------------ */

/* ---- per-band energy R/per_band_energy.c:13-30, bandwidth detector R/detect_cutoff_warped.c:13-83 ---- */
template <class LdsT> STAGE void st_energy_bw(const lc3d_plan* __restrict__ P, LdsT& L, int lane)
{
    const uint16_t* be = &lc3t_band_pool[PI(band_off)];
    float* en = &L.sm[SM_ENER];
    if (lane < PI(nbands)) {
        const int a = be[lane], b = be[lane + 1];
        float sum = 0;
        for (int j = a; j < b; j++) { const float v = L.A[j]; sum += v * v; }
        en[lane] = sum / (float)(b - a);
    }
    LSYNC();
    int bw = PI(fs_idx);
    if (PI(fs_idx) > 0 && PI(hrmode) == 0) {
        const int f = PI(fs_idx);
        const uint8_t* st = &lc3t_bw_start[(PI(bw_cls) * 4 + f - 1) * 4]; const uint8_t* sp = &lc3t_bw_stop[(PI(bw_cls) * 4 + f - 1) * 4];
        const float ev = en[lane];
        int counter = f;
        float sum = 0;
        for (int i = st[counter - 1]; i <= sp[counter - 1]; i++) sum += rl_f(ev, i);
        float mean = sum / (float)(sp[counter - 1] - st[counter - 1] + 1);
        while (mean < (float)lc3t_bw_quiet_thr[counter - 1]) {
            counter--;
            if (counter == 0) break;
            sum = 0;
            for (int i = st[counter - 1]; i <= sp[counter - 1]; i++) sum += rl_f(ev, i);
            mean = sum / (float)(sp[counter - 1] - st[counter - 1] + 1);
        }
        bw = counter;
        if (bw < f) {
            const float thr = (float)lc3t_bw_brick_thr[counter];
            const int stop = st[counter], dist = lc3t_bw_brick_dist[counter];
            int brick = 0;
            for (int i = stop; i >= stop - dist; i--) {
                const float ediff = (float)(10.0 * (double)m_log10f(rl_f(ev, i - dist + 1) + 1.1920928955078125e-07f) -
                                            10.0 * (double)m_log10f(rl_f(ev, i + 1) + 1.1920928955078125e-07f));
                if (ediff > thr) { brick = 1; break; }
            }
            if (!brick) bw = f;
        }
    }
    if (lane == 0) L.isc[I_BW] = bw;
    LSYNC();
}

/* ---- SNS scale factors R/sns_compute_scf.c:13-176 ---- */
template <class LdsT> STAGE void st_sns_scf(const lc3d_plan* __restrict__ P, LdsT& L, int lane)
{
    float* x = &L.sm[SM_ENER];
    const int smooth = uni(L.isc[I_ATT_FLAG]);
    int nb = PI(nbands);
    float c = x[lane < nb ? lane : 0];
    if (nb < 64) {
        const int d = 64 - nb;
        if (d < nb) c = lane < 2 * d ? x[lane >> 1] : x[lane - d];
        else {
            const float ratio = fabsf((float)(1.0 - 32.0 / (double)(float)nb));
            const int n4 = (int)round((double)(ratio * (float)nb));
            c = x[lane < 4 * n4 ? (lane >> 2) : n4 + ((lane - 4 * n4) >> 1)];
        }
        nb = 64;
    }
    /* smoothing + pre-emphasis, neighbours through DPP-free shuffles */
    const float c_up = __shfl_up(c, 1), c_dn = __shfl_down(c, 1);       /* shuffles stay outside the selects: every lane must execute them */
    const float mm = lane > 0 ? c_up : c, pp = lane < 63 ? c_dn : c;
    float s = (float)(0.5 * (double)c + 0.25 * (double)mm + 0.25 * (double)pp);
    s = s * P->sns_preemph[lane];
    float sum = 0;
    for (int i = 0; i < 64; i++) sum += rl_f(s, i);
    float mean = sum / (float)64;
    float nf = mean * PF(c_1em4);
    nf = nf > PF(c_2m32) ? nf : PF(c_2m32);
    if (s < nf) s = nf;
    const float xl = (float)((double)m_log2f(s) / 2.0);
    float* tmp = &L.sm[LdsT::MISC];
    LSYNC();
    x[lane] = s;                                    /* the reference overwrites the energies in place */
    tmp[lane] = xl;
    LSYNC();
    float v4 = 0;
    if (lane < 16) {
        const float W[6] = {(float)(1.0 / 12.0), (float)(2.0 / 12.0), (float)(3.0 / 12.0), (float)(3.0 / 12.0), (float)(2.0 / 12.0), (float)(1.0 / 12.0)};
        float t[6];
#pragma unroll
        for (int i = 0; i < 6; i++) { int q = lane * 4 - 1 + i; q = q < 0 ? 0 : q > 63 ? 63 : q; t[i] = tmp[q]; }
        float a = 0;
#pragma unroll
        for (int i = 0; i < 6; i++) a += t[i] * W[i];
        v4 = a;
    }
    sum = 0;
    for (int i = 0; i < 16; i++) sum += rl_f(v4, i);
    mean = (float)((double)sum / ((double)(float)nb / 4.0));
    float g = PF(sns_damping) * (v4 - mean);
    if (smooth) {
        const float gm2 = __shfl_up(g, 2), gm1 = __shfl_up(g, 1), gp1 = __shfl_down(g, 1), gp2 = __shfl_down(g, 2);
        float gs;
        if (lane == 0) gs = (float)((double)(g + gp1 + gp2) / 3.0);
        else if (lane == 1) gs = (float)((double)(gm1 + g + gp1 + gp2) / 4.0);
        else if (lane == 14) gs = (float)((double)(gm2 + gm1 + g + gp1) / 4.0);
        else if (lane == 15) gs = (float)((double)(gm2 + gm1 + g) / 3.0);
        else gs = (float)((double)(gm2 + gm1 + g + gp1 + gp2) / 5.0);
        sum = 0;
        for (int i = 0; i < 16; i++) sum += rl_f(gs, i);
        mean = sum / (float)16;
        g = PF(att_damping) * (gs - mean);
    }
    if (lane < 16) L.sm[SM_SCF + lane] = g;
    LSYNC();
}

/* ---- PVQ pulse search R/sns_quantize_scf.c:43-136: one search per lane, everything in registers ---- */
/* PVQ pulse searches R/sns_quantize_scf.c:43-136: the four searches (N=10,K=10 | N=6,K=1 on tgt+10 | N=16,K=8 | N=16,K=6) run side
 * by side, one per 16-lane row, lane i of a row owning dimension i.  What the reference does serially over the dimensions is
 * kept in its order: the sums xsum and xy are re-added by every lane from row-wide LDS arrays, and the candidate scan
 *     for i: if (a_i * cden > b_i * cnum) take i
 * (a float cross-multiplication, not a total order, so it cannot become a tree reduction) is replayed by every lane over the
 * row's (a, b) pairs read back from LDS - 6 VALU per candidate instead of 10, for four searches at once.  Dimensions beyond a
 * search's N carry |x| = 0 and b = +INF: a*cden > INF*cnum is false for every cnum >= 0, and cnum >= 0 from candidate 0 on. */
template <class LdsT> __device__ __forceinline__ void pvq_search_rows(LdsT& L, int lane, const float* tgt, float* pv)
{
    const int s = lane >> 4, i = lane & 15;
    const int dim = s == 0 ? 10 : s == 1 ? 6 : 16, K = s == 0 ? 10 : s == 1 ? 1 : s == 2 ? 8 : 6;
    float* scr = XCUR(L);                            /* X is scratch between the MDCT and TNS: 4 arrays of 4 rows x 16 */
    float* xs = scr + 16 * s; float* ys = scr + 64 + 16 * s; float* as = scr + 128 + 16 * s; float* bs = scr + 192 + 16 * s;
    const float xv = tgt[(s == 1 ? 10 : 0) + i];
    const bool valid = i < dim, neg = !(xv >= 0);
    const float xa = valid ? fabsf(xv) : 0.0f;
    xs[i] = xa;
    LSYNC();
    float xsum = 0;
#pragma unroll
    for (int j = 0; j < 16; j += 4) { const float4 u = *(const float4*)&xs[j]; xsum += u.x; xsum += u.y; xsum += u.z; xsum += u.w; }
    const bool live = xsum > PF(c_2m24);
    const float proj = live ? (float)(K - 1) / xsum : 0.0f;
    float y = floorf(xa * proj);                     /* this dimension's pulse count: a small integer, exact in float */
    ys[i] = y; as[i] = xa * y;
    LSYNC();
    float totf = 0, yy = 0, xy = 0;
#pragma unroll
    for (int j = 0; j < 16; j += 4) {
        const float4 u = *(const float4*)&ys[j], v = *(const float4*)&as[j];
        totf += u.x; totf += u.y; totf += u.z; totf += u.w;
        yy = yy + u.x * u.x; yy = yy + u.y * u.y; yy = yy + u.z * u.z; yy = yy + u.w * u.w;
        xy = xy + v.x; xy = xy + v.y; xy = xy + v.z; xy = xy + v.w;
    }
    int tot = live ? (int)totf : K;
    yy = yy * 0.5f;
    LSYNC();
    while (__ballot(tot < K)) {
        const bool act = tot < K;
        const float yyt = yy + 0.5f;
        float a = xy + xa; a = a * a;
        as[i] = a; bs[i] = valid ? yyt + y : INFINITY;
        LSYNC();
        float cnum, cden; int best = 0;
        {
            const float4 ua = *(const float4*)&as[0], ub = *(const float4*)&bs[0];
            cnum = ua.x; cden = ub.x;                /* candidate 0 always beats the initial (-2^15, 0) */
#define PVQ_CAND(aj, bj, j) do { const bool t = (aj) * cden > (bj) * cnum; cnum = t ? (aj) : cnum; cden = t ? (bj) : cden; best = t ? (j) : best; } while (0)
            PVQ_CAND(ua.y, ub.y, 1); PVQ_CAND(ua.z, ub.z, 2); PVQ_CAND(ua.w, ub.w, 3);
#pragma unroll
            for (int j = 4; j < 16; j += 4) {
                const float4 va = *(const float4*)&as[j], vb = *(const float4*)&bs[j];
                PVQ_CAND(va.x, vb.x, j); PVQ_CAND(va.y, vb.y, j + 1); PVQ_CAND(va.z, vb.z, j + 2); PVQ_CAND(va.w, vb.w, j + 3);
            }
#undef PVQ_CAND
        }
        const float xb = xs[best], yb = ys[best];
        LSYNC();
        if (act) {
            if (i == best) { y = y + 1.0f; ys[i] = y; }
            xy = xy + xb; yy = yyt + yb; tot++;
        }
        LSYNC();
    }
    yy = yy * 2.0f;
    /* all-zero target: the reference puts the pulses at y[0] and (out of range) y[dim]; only y[0] is ever read back */
    const int y0z = K / 2, ydz = -(K - K / 2);
    if (!live) yy = (float)(y0z * y0z + ydz * ydz);
    const float g = (float)(1.0 / (double)sqrtf(yy));
    int ya = valid ? (int)y : 0;
    if (!live) ya = i == 0 ? y0z : 0;
    const int yi = neg ? -ya : ya;
    ((int*)(pv + s * 32))[i] = yi; pv[s * 32 + 16 + i] = (float)yi * g;
}

/* MPVQ enumeration R/sns_quantize_scf.c:138-163 (integer), lane-parallel.  The reference walks pos = len-1 .. 0 with
 *   if (ls >= 0 && pv != 0) idx = 2*idx + ls;  ls = sign(pv) if pv != 0;  idx += offs[(len-pos-1)*11 + k];  k += |pv|
 * i.e. an affine map idx -> m*idx + c per step with m in {1, 2}.  Lane n of a 16-lane row takes step n: k is an exclusive
 * prefix sum, the sign in force is that of the nearest earlier non-zero, and idx = sum_n c_n << (doublings after n).
 * Row 0 enumerates pulses[0..len0), row 1 pulses[10..16) (the 6-dimensional part of the split shape); results are uniform. */
__device__ __forceinline__ void mpvq_index_rows(const int* pulses, int lane, int len0, int& ls0, int& idx0, int& ls1, int& idx1)
{
    const int grp = lane >> 4, n = lane & 15;
    const int len = grp == 0 ? len0 : 6;
    const bool on = lane < 32 && n < len;
    const int pv = on ? pulses[(grp ? 10 : 0) + len - 1 - n] : 0;
    const int sh = 16 * grp;
    const unsigned nzm = (unsigned)(__ballot(pv != 0) >> sh) & 0xFFFFu, ngm = (unsigned)(__ballot(pv < 0) >> sh) & 0xFFFFu;
    const unsigned below = nzm & ((1u << n) - 1u);
    const int ls_prev = below ? (int)((ngm >> (31 - __clz((int)below))) & 1u) : -1;
    const int apv = pv < 0 ? -pv : pv;
    int k = apv;                                      /* inclusive prefix sum inside the row */
    k += dpp_i<DPP_SHR1>(0, k); k += dpp_i<DPP_SHR2>(0, k); k += dpp_i<DPP_SHR4>(0, k); k += dpp_i<DPP_SHR8>(0, k);
    k -= apv;
    const unsigned off = on ? lc3t_mpvq_offs[n * 11 + k] : 0u;
    const bool dbl = below != 0 && pv != 0;
    const unsigned dbm = (unsigned)(__ballot(dbl) >> sh) & 0xFFFFu;
    const int d = __popc(dbm >> (n + 1));
    int c = (int)((off + (dbl ? (unsigned)ls_prev : 0u)) << d);
    c += dpp_i<DPP_SHR1>(0, c); c += dpp_i<DPP_SHR2>(0, c); c += dpp_i<DPP_SHR4>(0, c); c += dpp_i<DPP_SHR8>(0, c);
    const int lsf = nzm ? (int)((ngm >> (31 - __clz((int)nzm))) & 1u) : -1;
    idx0 = __builtin_amdgcn_readlane(c, 15); idx1 = __builtin_amdgcn_readlane(c, 31);
    ls0 = __builtin_amdgcn_readlane(lsf, 0); ls1 = __builtin_amdgcn_readlane(lsf, 16);
}

/* ---- SNS vector quantiser R/sns_quantize_scf.c:165-430 (+ DCT-II(16) R/dct4.c:28-48, IDCT-II :19-41) ---- */
template <class LdsT> STAGE void st_sns_vq(const lc3d_plan* __restrict__ P, LdsT& L, int lane)
{
    const float* env = &L.sm[SM_SCF];
    float* st1 = &L.sm[SM_ST1]; float* tgt = &L.sm[SM_TGT]; float* tgtp = &L.sm[SM_TGTP];
    float* vec = &L.sm[SM_VEC];
    int* isc = L.isc;
    SUB_BEGIN();
    {   /* stage 1: lane = sec*32 + codeword */
        const int sec = lane >> 5, c = lane & 31;
        const float* cb = sec ? lc3t_sns_hf : lc3t_sns_lf;
        float sum = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) { const float d = env[8 * sec + i] - cb[c * 8 + i]; sum += d * d; }
        int bi0, bi1;
        wave_argmin_first_2x32(sum, c, bi0, bi1);
        const int bi = sec ? bi1 : bi0;
        if (c == 0) isc[I_SCF0 + sec] = bi;
        if (c < 8) { const float s = cb[bi * 8 + c]; st1[8 * sec + c] = s; tgtp[8 * sec + c] = env[8 * sec + c] - s; }
    }
    LSYNC();
    SUB(4);
    {   /* DCT-II(16): every lane runs the 16-point DFT on the same data, lanes < 16 keep one output */
        float z[32];
#pragma unroll
        for (int i = 0; i < 8; i++) { z[2 * i] = tgtp[2 * i]; z[2 * i + 1] = 0; z[2 * (15 - i)] = tgtp[2 * i + 1]; z[2 * (15 - i) + 1] = 0; }
        dft16(z);
        float zr = 0, zi = 0;
#pragma unroll
        for (int i = 0; i < 16; i++) { zr = (lane == i) ? z[2 * i] : zr; zi = (lane == i) ? z[2 * i + 1] : zi; }
        const float twr = P->dct2_tw[2 * (lane & 15)], twi = P->dct2_tw[2 * (lane & 15) + 1];
        float o = zr * twr - zi * twi;
        if (lane == 0) o = o / PF(c_sqrt2);
        if (lane < 16) tgt[lane] = o;
    }
    LSYNC();
    SUB(5);
    /* four pulse searches, one 16-lane row each: 0:(N=10,K=10) 1:(N=6,K=1 on tgt+10) 2:(N=16,K=8) 3:(N=16,K=6) */
    float* pv = &L.sm[SM_PVQ];
    pvq_search_rows(L, lane, tgt, pv);
    LSYNC();
    SUB(6);
    const int* pA = (const int*)(pv); const int* pB = (const int*)(pv + 32);
    const int* pN = (const int*)(pv + 64); const int* pF = (const int*)(pv + 96);
    const float* nA = pv + 16; const float* nN = pv + 64 + 16; const float* nF = pv + 96 + 16;
    /* yC = [pA(10) | pB(6)], normalised */
    const int yCl = lane < 16 ? (lane < 10 ? pA[lane] : pB[lane - 10]) : 0;
    const float ysq = (float)(yCl * yCl);
    float sumy = 0;
    for (int i = 0; i < 16; i++) sumy += rl_f(ysq, i);
    const float gf = (float)(1.0 / (double)sqrtf(sumy));
    const float yCn = (float)yCl * gf;
    const float nz = (lane < 10) ? nA[lane] : 0.0f;
    if (lane < 16) {
        vec[0 * 16 + lane] = lc3t_sns_gain_reg[0] * yCn; vec[1 * 16 + lane] = lc3t_sns_gain_reg[1] * yCn;
#pragma unroll
        for (int k = 0; k < 4; k++) vec[(2 + k) * 16 + lane] = lc3t_sns_gain_reg_lf[k] * nz;
    }
    LSYNC();
    int idx = 0; float glob;
    {
        float err = INFINITY;
        if (lane < 6) { float s = 0; for (int j = 0; j < 16; j++) { const float d = tgt[j] - vec[lane * 16 + j]; s += d * d; } err = s; }
        float min_err = PF(c_2p15);
        for (int i = 0; i < 6; i++) { const float e = rl_f(err, i); if (e < min_err) { min_err = e; idx = i; } }
        glob = lc3t_sns_gain_q[idx];
    }
    /* three inverse DCTs in parallel: lanes 0-15 split candidate, 16-31 near, 32-47 far */
    float* idc = &L.sm[SM_MISC];      /* 48 inputs then 48 outputs at +48 */
    if (lane < 16) idc[lane] = vec[idx * 16 + lane] / glob;
    else if (lane < 32) idc[lane] = nN[lane - 16];
    else if (lane < 48) idc[lane] = nF[lane - 32];
    LSYNC();
    if (lane < 48) {
        const float* in = &idc[lane & ~15]; const int i = lane & 15;
        float sum = 0;
#pragma unroll
        for (int j = 0; j < 16; j++) {
            float t = (float)((double)in[j] * P->idct_cos[i * 16 + j]);
            if (j == 0) t *= PF(c_idct_n2);
            sum += t;
        }
        idc[48 + lane] = PF(c_idct_n1) * sum;
    }
    LSYNC();
    SUB(7);
    const float* split = &idc[48]; const float* subN = &idc[64]; const float* subF = &idc[80];
    /* error of the split candidate and of the 4 near / 8 far gains: one serial 16-term sum per lane (0 split, 1-4 near, 5-12 far) */
    float err = INFINITY;
    if (lane < 13) {
        const float g = lane == 0 ? glob : lane < 5 ? lc3t_sns_gain_near[lane - 1] : lc3t_sns_gain_far[lane - 5];
        const float* sb = lane == 0 ? split : lane < 5 ? subN : subF;
        float s = 0;
        if (lane == 0) { for (int j = 0; j < 16; j++) { const float d = tgtp[j] - g * sb[j]; s += d * d; } }
        else { for (int j = 0; j < 16; j++) s += (tgtp[j] - g * sb[j]) * (tgtp[j] - g * sb[j]); }
        err = s;
    }
    const float e_split = rl_f(err, 0);
    int sub_mode = 0, sub_gain = 0, shape = 0;    /* shape: 0 = yC, 1 = pA only, 2 = near, 3 = far */
    float e_sofar = PF(c_2p15), g_sel = 0; const float* v_sel = split;
    bool have = false;
    if (e_split < e_sofar) {
        if (idx <= 1) { sub_mode = 0; sub_gain = idx; shape = 0; } else { sub_mode = 1; sub_gain = idx - 2; shape = 1; }
        g_sel = glob; v_sel = split; e_sofar = e_split; have = true;
    }
    {
        float min_err = PF(c_2p15); int gi = idx; float gg = glob;
        for (int i = 0; i < 4; i++) { const float e = rl_f(err, 1 + i); if (e < min_err) { gi = i; min_err = e; gg = lc3t_sns_gain_near[i]; } }
        if (min_err < e_sofar) { sub_mode = 2; sub_gain = gi; shape = 2; g_sel = gg; v_sel = subN; e_sofar = min_err; have = true; }
        min_err = PF(c_2p15);
        for (int i = 0; i < 8; i++) { const float e = rl_f(err, 5 + i); if (e < min_err) { gi = i; min_err = e; gg = lc3t_sns_gain_far[i]; } }
        if (min_err < e_sofar) { sub_mode = 3; sub_gain = gi; shape = 3; g_sel = gg; v_sel = subF; have = true; }
    }
    if (lane < 16) {
        const float st2 = have ? g_sel * v_sel[lane] : 0.0f;
        L.sm[SM_SCFQ + lane] = st1[lane] + st2;
        /* selected pulse vector for the MPVQ enumeration */
        int pl = shape == 0 ? (lane < 10 ? pA[lane] : pB[lane - 10]) : shape == 1 ? (lane < 10 ? pA[lane] : 0) : shape == 2 ? pN[lane] : pF[lane];
        if (!have) pl = 0;
        ((int*)vec)[lane] = pl;
    }
    LSYNC();
    SUB(8);
    {
        int ls, mi, a6, b6;
        mpvq_index_rows((const int*)vec, lane, sub_mode < 2 ? 10 : 16, ls, mi, a6, b6);
        const int i6 = sub_mode == 0 ? b6 * 2 + a6 : sub_mode == 2 ? -1 : -2;
        if (lane == 0) { isc[I_SCF2] = sub_mode; isc[I_SCF3] = sub_gain; isc[I_SCF4] = ls; isc[I_SCF5] = mi; isc[I_SCF6] = i6; }
    }
    LSYNC();
    SUB(9);
}

/* ---- SNS interpolation R/sns_interpolate_scf.c:13-89 and spectral shaping R/mdct_shaping.c:13-22 ---- */
template <class LdsT> STAGE void st_sns_apply(const lc3d_plan* __restrict__ P, LdsT& L, int lane, const float* src /* the MDCT spectrum: L.A, or where the front kernel's copy was parked */,
                        unsigned bob0, unsigned bob1, unsigned bob2, unsigned bob3 /* band index of bin lane + 64 k in byte k: constant for the launch, fetched once */)
{
    const float* g = &L.sm[SM_SCFQ];
    float* gi = &L.sm[SM_GI];
    float v;
    if (lane < 2) v = g[0];
    else if (lane < 62) {
        const int n = (lane - 2) >> 2, r = (lane - 2) & 3;
        const double dd = (double)(g[n + 1] - g[n]);
        if (r == 0) v = (float)((double)g[n] + dd / 8.0);
        else if (r == 1) v = (float)((double)g[n] + 3.0 * dd / 8.0);
        else if (r == 2) v = (float)((double)g[n] + 5.0 * dd / 8.0);
        else v = (float)((double)g[n] + 7.0 * dd / 8.0);
    } else {
        const double dd = (double)(g[15] - g[14]);
        v = lane == 62 ? (float)((double)g[15] + dd / 8.0) : (float)((double)g[15] + 3.0 * dd / 8.0);
    }
    const int nb = PI(nbands);
    if (nb < 64) {
        gi[lane] = v;
        LSYNC();
        const int d = 64 - nb;
        v = 0;
        if (d < 32) {
            if (lane < d) v = (float)((double)(gi[2 * lane] + gi[2 * lane + 1]) / 2.0);
            else if (lane < nb) v = gi[lane + d];
        } else {
            const float ratio = fabsf((float)(1.0 - 32.0 / (double)(float)nb));
            const int n4 = (int)round((double)(ratio * (float)nb));
            if (lane < n4) v = (float)((double)(gi[4 * lane] + gi[4 * lane + 1] + gi[4 * lane + 2] + gi[4 * lane + 3]) / 4.0);
            else if (lane < nb) { const int i = lane - n4; v = (float)((double)(gi[4 * n4 + 2 * i] + gi[4 * n4 + 2 * i + 1]) / 2.0); }
        }
        LSYNC();
    }
    if (lane < nb) gi[lane] = m_pow2f(-v);
    LSYNC();
    const unsigned bob[4] = {bob0, bob1, bob2, bob3};
#pragma unroll
    for (int k = 0; k < (MAXN + WAVE - 1) / WAVE; k++) {
        const int j = lane + 64 * k;
        if (j < PI(N)) {
            const int b = (int)((bob[k >> 2] >> (8 * (k & 3))) & 255u);
            const float x = src[j];
            L.A[j] = b < nb ? x * gi[b] : x;
        }
    }
    LSYNC();
}

/* ------------------------------------------------------------------------------------------------ */
/* TNS: R/tns_coder.c:170-362                                                                        */
/* ------------------------------------------------------------------------------------------------ */
struct TnsGeom { int numfilters, maxOrder, nSub, start0, start1, stop0, stop1; float maxPG; int obits_off; };   /* scalars only: no stack object */

template <class LdsT> __device__ __forceinline__ TnsGeom tns_geom(LdsT& L, int bw_idx, int bw_bin)
{
    TnsGeom g;
    int fs = PI(fs), N = PI(N); const int nBits = CI(total_bits), dms = PI(dms);
    g.numfilters = (fs >= 32000 && dms >= 50) ? 2 : 1;
    g.start0 = g.start1 = g.stop0 = g.stop1 = 0;
    if ((double)N > 40 * ((double)(float)dms / 10.0)) { N = (int)(40 * ((double)(float)dms / 10.0)); fs = 40000; }
    g.start0 = (600 * N * 2 / fs) + 1;
    if (g.numfilters == 1) g.stop0 = N; else { g.start1 = N / 2 + 1; g.stop0 = N / 2; g.stop1 = N; }
    g.maxOrder = dms == 100 ? 8 : 4; g.nSub = dms == 100 ? 3 : 2;
    g.maxPG = 2; g.obits_off = 8;
    if ((dms >= 50 && (double)nBits >= 48 * ((double)(float)dms / 10.0)) || dms == 25) { g.maxPG = 1.5f; g.obits_off = 0; }
    if (bw_idx >= 3 && g.numfilters == 2) { g.start1 = bw_bin / 2 + 1; g.stop0 = bw_bin / 2; g.stop1 = bw_bin; }
    else { g.numfilters = 1; g.stop0 = bw_bin; }
    return g;
}

/* LPC weighting (total_bits < 480 only): polynomial weighting + step-down back to reflection coefficients,
 * R/tns_coder.c:91-155,279-287.  Rare and index-heavy: lane 0 works in LDS scratch sc[]: a_in[9] at 36, rc_in[8] at 46. */
/* per-filter TNS scratch: work area sc[64] (LPC weighting 0..35, a[] 36.., rc[] 46.., predGain 63) and quantised rc[8] */
#define TNS_SC(L, f)  (&(L).sm[(f) ? (L).TNS1 : (L).TNS0 + 112])
#define TNS_RCS(L, f) (&(L).sm[(f) ? (L).TNS1 + 64 : (L).TNS0 + 104])
#define TNS_RACC(L)   (&(L).sm[(L).TNS0])          /* [f][sub][k] sums (54), sub-division energies (6) at +54, r[f][9] at +64, lattice state (8) at +96 */

template <class LdsT> STAGE void tns_lpc_weight(LdsT& L, int lane, int f, int maxOrder, float maxPG, float predGain)
{
    float* sc = TNS_SC(L, f);
    if (lane == 0) {
        float* pa = sc + 56;                /* 9 */
        pa[0] = 1;
        for (int j = 1; j <= maxOrder - 1; j++) pa[j] = -sc[36 + maxOrder - j];
        pa[maxOrder] = sc[46 + maxOrder - 1];
        const float alpha = (float)((double)((maxPG - predGain)) * (0.85f - 1.0) / (double)(maxPG - 1.5f) + 1.0);
        for (int i = 0; i <= maxOrder; i++) sc[i] = pa[i] * m_powf(alpha, (float)i);
        int len = maxOrder + 1; const int len0 = len;
        float* pa_ = sc; float* out = sc + 9; float* t0 = sc + 18; float* bf = sc + 26;
        for (int i = 0; i < len - 1; i++) out[i] = 0;
        { float a0 = pa_[0]; for (int i = 0; i < len; i++) { pa_[i] = pa_[i] / a0; a0 = pa_[0]; } }
        out[len - 1] = pa_[len - 1];
        for (int k = len0 - 2; k >= 0; k--) {
            for (int i = 0; i < len - 1; i++) t0[i] = pa_[1 + i];
            int l = len - 1;
            const float knxt = t0[l - 1];
            l = l - 1;
            bf[0] = 1;
            for (int i = 0; i < l; i++) {
                const float t2 = knxt * t0[l - 1 - i];
                bf[i + 1] = (float)((double)(t0[i] - t2) / (1.0 - (double)(fabsf(knxt) * fabsf(knxt))));
            }
            len = l + 1;
            out[k] = bf[len - 1];
            for (int i = 0; i < len; i++) pa_[i] = bf[i];
        }
        for (int i = 0; i < len0 - 1; i++) out[i] = out[i + 1];
    }
    LSYNC();
}

/* Levinson-Durbin (R/tns_coder.c:41-89) and prediction gain; the two filters are independent up to here, so lane f runs filter f
 * (fully unrolled for register residency).  Code 0: filter off, 1: on, 2: on and LPC weighting required; lane f leaves a[] at
 * sc[36..], rc[] at sc[46..], predGain at sc[63] of its filter's scratch.  Returns code0 | code1 << 2. */
template <class LdsT> STAGE int tns_levinson(LdsT& L, int lane, int nf, int maxOrder, float maxPG)
{
    const int f = lane & 1;
    const float* racc = TNS_RACC(L) + 64 + f * 9;
    float r[9], a[9], rc[8], buf[9];
#pragma unroll
    for (int i = 0; i < 9; i++) { r[i] = racc[i]; a[i] = 0; }
    float g = r[1] / r[0];
    a[0] = g;
    float v = (float)((1.0 - (double)(g * g)) * (double)r[0]);
    rc[0] = -g;
#pragma unroll
    for (int t = 1; t < 8; t++) {
        if (t < maxOrder) {
            float sum = 0;
#pragma unroll
            for (int i = 1; i <= t; i++) sum += a[i - 1] * r[i];
            g = (r[t + 1] - sum) / v;
#pragma unroll
            for (int j = 1; j <= t; j++) buf[j] = a[j - 1] - g * a[t - j];
#pragma unroll
            for (int j = 1; j <= t; j++) a[j] = buf[j];
            a[0] = g;
            v = v * (1 - g * g);
            rc[t] = -g;
        } else rc[t] = 0;
    }
    const float predGain = r[0] / v;
    const int tns = predGain > 1.5f;
    if (lane < nf) {
        float* sc = TNS_SC(L, f);
#pragma unroll
        for (int j = 0; j < 9; j++) sc[36 + j] = a[j];
#pragma unroll
        for (int j = 0; j < 8; j++) sc[46 + j] = rc[j];
        sc[63] = predGain;
    }
    const int code = tns ? (predGain < maxPG ? 2 : 1) : 0;
    LSYNC();
    return __builtin_amdgcn_readlane(code, 0) | (nf > 1 ? __builtin_amdgcn_readlane(code, 1) << 2 : 0);
}

/* reflection-coefficient quantisation, order and bit count (R/tns_coder.c:289-336) of both filters: lanes 0-31 serve filter 0,
 * lanes 32-63 filter 1; `codes` from tns_levinson.  Results: quantised rc -> TNS_RCS(f), order / indices -> isc.  Returns the
 * bits the filters add (flags included). */
template <class LdsT> STAGE int tns_quant(LdsT& L, int lane, int nf, int maxOrder, int obits_off, int codes)
{
    /* lane i < 8 of half f owns reflection coefficient i of filter f: its interval among the 17 (R/tns_coder.c:157-168 findRC_idx: the
     * intervals (thr[q], thr[q+1]] are disjoint, none -> 0), its quantised value, its Huffman bits; order = last non-zero coefficient
     * (ballot), bit count = row sum */
    const int f = lane >> 5, i = lane & 31;
    const int code = (codes >> (2 * f)) & 3;
    const bool live = f < nf, mine = live && i < maxOrder && i < 8;
    float* rcs = TNS_RCS(L, f);
    const float* sc = TNS_SC(L, f);
    const float rc = mine ? sc[(code == 2 ? 9 : 46) + i] : 0.0f;
    int tns = live && code != 0;
    int idx = 0;
#pragma unroll
    for (int q = 0; q < 17; q++) idx = (rc <= lc3t_tns_rc_thr[q + 1] && rc > lc3t_tns_rc_thr[q]) ? q : idx;
    if (!mine) idx = 0;
    const float qv = (mine && tns) ? lc3t_tns_rc_pts[idx] : 0.0f;
    const unsigned long long nzm = __ballot(qv != 0);
    const unsigned nzh = (f ? (unsigned)(nzm >> 32) : (unsigned)nzm) & 0xFFu;
    const int ord = nzh ? 32 - __clz((int)nzh) : 0;
    if (ord == 0) tns = 0;            /* would be undefined behaviour in the reference (R/tns_coder.c:311-321); filter off */
    int term = (tns && i < ord && i < 8) ? lc3t_tns_coef_bits[i * 17 + idx] : 0;
    term += dpp_i<DPP_SHR1>(0, term); term += dpp_i<DPP_SHR2>(0, term); term += dpp_i<DPP_SHR4>(0, term);      /* lane 7 of the row: coefficients 0..7 */
    const int tsum = __builtin_amdgcn_readlane(term, 7), tsum1 = __builtin_amdgcn_readlane(term, 39);
    int bits = live ? 1 : 0;
    if (tns) bits += ((f ? tsum1 : tsum) + lc3t_tns_order_bits[obits_off + ord - 1] + 2047) >> 11;
    if (live && i < 8) {
        rcs[i] = qv;
        if (tns && i < ord) L.isc[I_TNS_IDX0 + f * 8 + i] = idx;
        if (i == 0) L.isc[I_TNS_ORD0 + f] = tns ? ord : 0;
    }
    LSYNC();
    return __builtin_amdgcn_readlane(bits, 0) + __builtin_amdgcn_readlane(bits, 32);
}

/* Lattice MA filter of one TNS filter (R/tns_coder.c:339-357), lane-parallel with exact replay: each lane owns a run of
 * consecutive bins and first replays the 8 preceding inputs; bins before the filter start come from the carried state. */
template <class LdsT> STAGE void tns_lattice(LdsT& L, int lane, int f, int b_first, int cnt, int ord)
{
    f = uni(f); b_first = uni(b_first); cnt = uni(cnt); ord = uni(ord);
    float* stt = TNS_RACC(L) + 96;
    const float* rcs = TNS_RCS(L, f);
    float* X = XCUR(L);
    float rc[8], st[8], carried[8];
#pragma unroll
    for (int j = 0; j < 8; j++) { rc[j] = unif(rcs[j]); carried[j] = unif(stt[j]); st[j] = carried[j]; }
    const int chunk = (cnt + WAVE - 1) / WAVE;
    const int b0 = b_first + lane * chunk;
    int nmine = imin(chunk, b_first + cnt - b0); if (nmine < 0) nmine = 0;
    for (int t = b0 - 8; nmine > 0 && t < b0 + nmine; t++) {
        if (t < b_first) {
#pragma unroll
            for (int j = 0; j < 8; j++) st[j] = carried[j];
            continue;
        }
        float s = L.A[t], save = s;
#pragma unroll
        for (int j = 0; j < 7; j++) {
            if (j < ord - 1) { const float tt = rc[j] * s + st[j]; s += rc[j] * st[j]; st[j] = save; save = tt; }
        }
#pragma unroll
        for (int j = 0; j < 8; j++) if (j == ord - 1) { s += rc[j] * st[j]; st[j] = save; }
        if (t >= b0) X[t] = s;
    }
    const int lastLane = (cnt - 1) / chunk;
    LSYNC();
    if (lane == lastLane) {
#pragma unroll
        for (int j = 0; j < 8; j++) stt[j] = st[j];
    }
    for (int t = b_first + lane; t < b_first + cnt; t += WAVE) L.A[t] = X[t];
    LSYNC();
}

/* sub-division autocorrelations and the lag-windowed r[f][0..8] of both filters (R/tns_coder.c:258-281) */
template <class LdsT> STAGE void tns_sums(LdsT& L, int lane, int bw_idx, int bw_bin)
{
    const TnsGeom G = tns_geom(L, bw_idx, bw_bin);
    float* racc = TNS_RACC(L);                 /* [f][sub][k] 2*3*9 = 54, sub-division energies 6 at +54, r[f][9] at +64 */
    {   /* one serial sum per lane: lane (f, sub, k) < 54 owns autocorrelation lag k of a sub-division (R/tns_coder.c:18-39,258-277);
         * the sub-division energies are the lag-0 sums (same terms, same order).  Eight terms per block; the operands of the next
         * block are read before the arithmetic of the current one. */
        const int f = lane / 27, r = lane % 27, sub = r / 9, k = r % 9;
        const bool on = lane < 54 && f < G.numfilters && sub < G.nSub && k <= G.maxOrder;
        const int fstart = f ? G.start1 : G.start0, fstop = f ? G.stop1 : G.stop0;
        const float sublen = (float)(((double)(float)fstop + 1.0 - (double)(float)fstart) / (double)(float)G.nSub);
        const int lo = (int)(floor((double)(sublen * (float)sub)) + fstart - 1);
        const int hi = (int)(floor((double)(sublen * (float)(sub + 1))) + fstart - 1);
        const int cnt = on ? hi - lo - k : 0;                         /* terms x[i] * x[i-k], i = k .. n-1 */
        const float* pb = on ? &L.A[lo] : &L.A[0]; const float* pa = pb + (on ? k : 0);
        const int cmax = wave_max_i(cnt), cmin = -wave_max_i(on ? -cnt : -0x7fff);
        float acc = 0;
        int t0 = 0;
        if (cmin >= 8) {
            float ca[8], cb[8];
#pragma unroll
            for (int j = 0; j < 8; j++) { ca[j] = pa[j]; cb[j] = pb[j]; }
            for (; t0 + 8 <= cmin; t0 += 8) {
                float na[8], nb[8];
#pragma unroll
                for (int j = 0; j < 8; j++) { na[j] = pa[t0 + 8 + j]; nb[j] = pb[t0 + 8 + j]; }     /* at most 15 floats past a run: inside A */
#pragma unroll
                for (int j = 0; j < 8; j++) acc += ca[j] * cb[j];
#pragma unroll
                for (int j = 0; j < 8; j++) { ca[j] = na[j]; cb[j] = nb[j]; }
            }
        }
        for (; t0 < cmax; t0 += 8) {
#pragma unroll
            for (int j = 0; j < 8; j++) { const float pr = pa[t0 + j] * pb[t0 + j]; acc += (t0 + j < cnt) ? pr : 0.0f; }
        }
        if (on) racc[lane] = acc;
    }
    LSYNC();
    if (lane < 6) racc[54 + lane] = racc[(lane / 3) * 27 + (lane % 3) * 9];
    LSYNC();
    if (lane < 18) {   /* r[f][k] with the reference's zero-energy escape, lag window */
        const int f = lane / 9, k = lane % 9;
        float r = 0;
        if (f < G.numfilters && k <= G.maxOrder) {
            for (int sub = 0; sub < G.nSub; sub++) {
                const float e = racc[54 + f * 3 + sub];
                if (e == 0) { r = (k == 0) ? 1.0f : 0.0f; break; }
                r = r + racc[f * 27 + sub * 9 + k] / e;
            }
            r = r * lc3t_tns_lagwin[k];
        }
        racc[64 + lane] = r;
    }
    if (lane >= 32 && lane < 40) TNS_RACC(L)[96 + lane - 32] = 0;     /* lattice state, persists from filter 0 to 1 */
    LSYNC();
}

/* TNS driver, inlined into the kernel so that the STAGE functions stay leaf calls (a nested call level makes the callee
 * save registers to scratch, i.e. HBM write traffic that is not part of the algorithm) */
template <class LdsT> __device__ __forceinline__ void st_tns(const lc3d_plan* __restrict__ P, LdsT& L, int lane, int bw_idx, int bw_bin)
{
    SUB_BEGIN();
    tns_sums(L, lane, bw_idx, bw_bin);
    SUB(21);
    const TnsGeom G = tns_geom(L, bw_idx, bw_bin);
    const int codes = tns_levinson(L, lane, G.numfilters, G.maxOrder, G.maxPG);
    for (int f = 0; f < G.numfilters; f++)
        if (((codes >> (2 * f)) & 3) == 2) tns_lpc_weight(L, lane, f, G.maxOrder, G.maxPG, unif(TNS_SC(L, f)[63]));
    SUB(22);
    const int bits = tns_quant(L, lane, G.numfilters, G.maxOrder, G.obits_off, codes);
    SUB(23);
    for (int f = 0; f < G.numfilters; f++) {
        const int ord = uni(L.isc[I_TNS_ORD0 + f]);
        const int fstart = f ? G.start1 : G.start0, fstop = f ? G.stop1 : G.stop0;
        if (ord > 0) tns_lattice(L, lane, f, fstart - 1, fstop - fstart + 1, ord);
    }
    SUB(24);
    if (lane == 0) { L.isc[I_TNS_NF] = G.numfilters; L.isc[I_TNS_BITS] = bits; }
    LSYNC();
}

/* ------------------------------------------------------------------------------------------------ */
/* quantisation                                                                                      */
/* ------------------------------------------------------------------------------------------------ */

/* one bisection probe of R/estimate_global_gain.c:97-124 for this lane's candidate offset; the energies come from
 * lane registers (e0: j < 64, e1: j >= 64) through readlane, so the 100-step serial chain never touches LDS */
/* keep a value opaque to the optimiser: stops it from sinking the (double) arithmetic of a select operand into a divergent
 * branch (a taken branch costs ~30 cycles on this chain, a v_cndmask 8) */
__device__ __forceinline__ float opaque(float v) { asm volatile("" : "+v"(v)); return v; }
__device__ __forceinline__ double opaque_d(double v) { asm volatile("" : "+v"(v)); return v; }
__device__ __forceinline__ int opaque_i(int v) { asm volatile("" : "+v"(v)); return v; }

/* number of leading (low-j) energies a probe has to visit: while even the smallest candidate of the wave sees
 * en[j] - cand < thr7 and no lane has left the all-zero state, a step is a no-op, so the trailing run of such j is skipped */
__device__ __forceinline__ int gain_probe_len(float thr7, const float* ev /* [NQL]: energies of j = lane + 64 h */, int lane, int nq, int cand_min)
{
    const float fmin = (float)cand_min;
    int len = 0;
#pragma unroll
    for (int h = 0; h < NQL; h++) {
        const unsigned long long m = __ballot(lane + 64 * h < nq && !(ev[h] - fmin < thr7));
        if (m) len = 64 * (h + 1) - (int)__clzll((long long)m);
    }
    return uni(len);
}

/* en[j] = enb[LC3D_ROW_OFF(e0 + j, RT)]: the energies of a tiled spectrum row (lc3_plan.h), or a plain array with e0 = 0, RT = 1 */
__device__ __forceinline__ bool gain_probe(float thr7, float thr50, const float* enb, int e0, int RT, int nq, int cand, float target)
{
#define en_(j_) enb[LC3D_ROW_OFF(e0 + (j_), RT)]
    float ener = 0; bool iszero = true;
    const float fc = (float)cand;
    /* one step of R/estimate_global_gain.c:103-121, branch-free.  All four outcomes have the form (float)(((double)ener + X) + Y):
     * unchanged (0, 0); + 2.7*1.4 (c, 0); + 2t - 50*1.4 (2t, -70); float ener + t == (float)((double)ener + (double)t) because a
     * double holds more than 2*24+2 bits (double rounding is innocuous).  X (a float: 0, t or 2t) and Y (0, c_lo or c_hi) are
     * selected off the serial chain, which is then cvt - add - add - cvt; double-precision instructions cost twice a float one, so
     * as much of the selection as possible happens in float. */
    const double c_lo = (2.7) * (28.0 / 20.0), c_hi = -((50.0) * (28.0 / 20.0));
    /* Y in halves: lo and hi exclude each other, c_hi has a zero low word */
    const int clo_h = __double2hiint(c_lo), clo_l = __double2loint(c_lo), chi_h = __double2hiint(c_hi);
#define GSTEP(ev) do { const float t = (ev) - fc; const bool lo = t < thr7, hi = t > thr50, lonz = lo && !iszero; \
        const float x_hi = opaque(hi ? t + t : t), xf = opaque(lo ? 0.0f : x_hi);            /* 2t is exact in float */ \
        const int y_h0 = opaque_i(hi ? chi_h : 0), y_h = opaque_i(lonz ? clo_h : y_h0), y_l = opaque_i(lonz ? clo_l : 0); \
        ener = (float)(((double)ener + (double)xf) + __hiloint2double(y_h, y_l)); iszero = iszero && lo; } while (0)
    int j = nq - 1;
    for (; j >= 3; j -= 4) { const float v0 = en_(j), v1 = en_(j - 1), v2 = en_(j - 2), v3 = en_(j - 3); GSTEP(v0); GSTEP(v1); GSTEP(v2); GSTEP(v3); }
    for (; j >= 0; j--) GSTEP(en_(j));
#undef GSTEP
#undef en_
    return ener > target && !iszero;
}

/* ---- global gain estimate R/estimate_global_gain.c:30-137 ---- */
template <class LdsT> STAGE void st_gain_estimate(const lc3d_plan* __restrict__ P, const lc3d_chan* __restrict__ C, LdsT& L, int lane, int nbitsSQ)
{
    const int lg = PI(ylen), off = CI(gg_off), nq = lg >> 2;
    SUB_BEGIN();
    float tbits_off = unif(L.fsc[F_TBITS_OFF]);
    int mem_target = uni(L.isc[I_MEM_TARGET]); const int mem_spec = uni(L.isc[I_MEM_SPEC]);
    if (mem_target < 0) tbits_off = 0;
    else {
        float v = tbits_off + (float)mem_target - (float)mem_spec;
        v = -40 > v ? -40 : v; v = 40 < v ? 40 : v;
        tbits_off = (float)(0.8 * (double)tbits_off + 0.2 * (double)v);
    }
    mem_target = nbitsSQ;
    nbitsSQ = (int)((double)nbitsSQ + round((double)tbits_off));
    float xm = 0;
#pragma unroll
    for (int k = 0; k < MAXN / WAVE + 1; k++) { const int i = lane + 64 * k; if (i < lg) xm = fmaxf(xm, fabsf(L.A[i])); }
    const float x_max = unif(wave_max_f(xm));
    float reg_val = 0;
    if (PI(hrmode) && CI(reg_bits) > 0) {
        float M0 = 1e-5f, M1 = 1e-5f; const float thresh = 2 * PF(frame_ms);
        for (int i0 = 0; i0 < lg; i0 += WAVE) {                      /* R/estimate_global_gain.c:58-62, serial in double */
            const int i = i0 + lane;
            const double ax = i < lg ? fabs((double)L.A[i]) : 0.0, ix = (double)i * ax;
            const int cnt = imin(WAVE, lg - i0);
            for (int k = 0; k < cnt; k++) { M0 = (float)((double)M0 + rl_d(ax, k)); M1 = (float)((double)M1 + rl_d(ix, k)); }
        }
        const float q = M1 / M0;
        const float rB = 8 * (1 - (q < thresh ? q : thresh) / thresh);
        reg_val = x_max * m_powf(2.0f, (float)(-CI(reg_bits)) - rB);
    }
    float ind = 0, ind_min = 0;
    if (x_max == 0) { ind_min = (float)off; ind = 0; mem_target = -1; }
    else {
        const float g_min = PI(hrmode) == 1 ? x_max / (float)(32768 * 256 - 2) : (float)((double)x_max / (32768 - 0.375));
        ind_min = (float)ceil(28.0 * (double)m_log10f(g_min));
        float* en = XCUR(L);                         /* X is scratch between TNS and quantisation */
        SUB(10);
        float ev[NQL];
#pragma unroll
        for (int h = 0; h < NQL; h++) {
            const int j = lane + 64 * h;
            ev[h] = 0;
            if (j < nq) {
                const float* x = &L.A[4 * j];
                float t = x[0] * x[0];
                t += x[1] * x[1]; t += x[2] * x[2]; t += x[3] * x[3];
                en[j] = ev[h] = (float)((28.0 / 20.0) * (7 + 10.0 * (double)m_log10f(t + reg_val + PF(c_2m31))));
            }
        }
        LSYNC();
        SUB(11);
        const float target = (float)((28.0 / 20.0) * (1.4) * (double)nbitsSQ);
        const float thr7 = PF(c_thr7_up), thr50 = PF(c_thr50_dn);
        const int offset0 = 255 + off;
        /* 8-step bisection R/estimate_global_gain.c:97-124.  A probe compares an ORDERED float sum of up to 240 non-negative terms with
         * the target, so only the sign of (sum - target) is needed: every probe is first evaluated lane-parallel (one term per lane,
         * the iszero latch = "no band above this one has left the quiet state" from a ballot, tree sum in float).  The serial float
         * sum of n non-negative terms differs from the exact sum by at most n * 2^-24 * sum, the tree sum of rounded terms by at most
         * (log2(n) + 3) * 2^-24 * sum: when the tree sum is further from the target than 4e-5 * sum + 1e-4 (> twice both bounds at
         * n = 240) the reference's comparison is decided.  Otherwise (measured: well under 1 % of the probes) that probe is repeated with
         * the reference's serial chain (gain_probe).  ~40 instructions per step instead of two 100-step double-precision chains. */
        int m = 0;
        const float c_lo_f = (float)((2.7) * (28.0 / 20.0));
        for (int i = 0; i < 8; i++) {
            const int fac = 128 >> i, cand = offset0 - m - fac;
            const float fc = (float)cand;
            float tsum = 0; int top = -1;
            bool lo_[NQL]; float tv[NQL];
#pragma unroll
            for (int h = NQL - 1; h >= 0; h--) {
                const bool valid = lane + 64 * h < nq;
                const float t = ev[h] - fc;
                lo_[h] = !valid || t < thr7;
                tv[h] = t > thr50 ? (t + t) - 70.0f : t;
                const unsigned long long nz = __ballot(!lo_[h]);
                if (nz && top < 0) top = 64 * h + 63 - (int)__clzll((long long)nz);
            }
#pragma unroll
            for (int h = 0; h < NQL; h++) tsum += lo_[h] ? (lane + 64 * h < top ? c_lo_f : 0.0f) : tv[h];
            const float S = wave_sum_f_tree(tsum);
            bool addback;
            if (top < 0) addback = false;                                         /* iszero stays set: never added back */
            else {
                const float margin = 4e-5f * S + 1e-4f;
                if (S - margin > target) addback = true;
                else if (S + margin < target) addback = false;
                else addback = gain_probe(thr7, thr50, en, 0, 1, nq, cand, target);         /* too close to call: the reference's serial sum */
            }
            if (!addback) m += fac;
        }
        SUB(12);
        SUB(13);
        const int offset = offset0 - m;
        if ((float)offset < ind_min) mem_target = -1;
        ind = (ind_min > (float)offset ? ind_min : (float)offset) - (float)off;
    }
    if (lane == 0) {
        L.fsc[F_TBITS_OFF] = tbits_off; L.isc[I_MEM_TARGET] = mem_target;
        L.isc[I_GGMIN] = (int)ind_min; L.isc[I_GG] = (int)ind;
        L.fsc[F_GAIN] = P->gain_est[(int)(ind + (float)off) + 256];
    }
    LSYNC();
    SUB(14);
}

/* ---- the stateless half of the global-gain estimate (R/estimate_global_gain.c:52-91): spectrum maximum -> smallest gain index,
 * the high-resolution regulariser, the per-4-bin log energies en[].  Runs in lc3_enc_shape_kernel for all frames at once; the
 * half that reads what the previous frame left (the rate loop and the bisection against its target, :42-50, :93-137) is
 * lc3_enc_rate_kernel.  en[] goes behind the ylen lines of the frame's row, ind_min and the all-zero flag to the record. ---- */
template <class LdsT> STAGE void st_gain_prep(const lc3d_plan* __restrict__ P, const lc3d_chan* __restrict__ C, LdsT& L, int lane, float* __restrict__ row /* tiled (lc3_plan.h) */, int e0, int RT, float* __restrict__ rec)
{
    const int lg = PI(ylen), off = CI(gg_off), nq = lg >> 2;
    float xm = 0;
#pragma unroll
    for (int k = 0; k < MAXN / WAVE + 1; k++) { const int i = lane + 64 * k; if (i < lg) xm = fmaxf(xm, fabsf(L.A[i])); }
    const float x_max = unif(wave_max_f(xm));
    float reg_val = 0;
    if (PI(hrmode) && CI(reg_bits) > 0) {
        float M0 = 1e-5f, M1 = 1e-5f; const float thresh = 2 * PF(frame_ms);
        for (int i0 = 0; i0 < lg; i0 += WAVE) {                      /* R/estimate_global_gain.c:58-62, serial in double */
            const int i = i0 + lane;
            const double ax = i < lg ? fabs((double)L.A[i]) : 0.0, ix = (double)i * ax;
            const int cnt = imin(WAVE, lg - i0);
            for (int k = 0; k < cnt; k++) { M0 = (float)((double)M0 + rl_d(ax, k)); M1 = (float)((double)M1 + rl_d(ix, k)); }
        }
        const float q = M1 / M0;
        const float rB = 8 * (1 - (q < thresh ? q : thresh) / thresh);
        reg_val = x_max * m_powf(2.0f, (float)(-CI(reg_bits)) - rB);
    }
    float ind_min = (float)off;
    if (x_max != 0) {
        const float g_min = PI(hrmode) == 1 ? x_max / (float)(32768 * 256 - 2) : (float)((double)x_max / (32768 - 0.375));
        ind_min = (float)ceil(28.0 * (double)m_log10f(g_min));
#pragma unroll
        for (int h = 0; h < NQL; h++) {
            const int j = lane + 64 * h;
            if (j < nq) {
                const float* x = &L.A[4 * j];
                float t = x[0] * x[0];
                t += x[1] * x[1]; t += x[2] * x[2]; t += x[3] * x[3];
                row[LC3D_ROW_OFF(e0 + j, RT)] = (float)((28.0 / 20.0) * (7 + 10.0 * (double)m_log10f(t + reg_val + PF(c_2m31))));
            }
        }
    }
    if (lane == 0) { rec[FR_GGMIN] = ind_min; ((int*)rec)[FR_XZERO] = x_max == 0 ? 1 : 0; }
}

/* ---- quantisation + exact bit estimate R/quantize_spec.c:26-197 ---- */
template <class LdsT> STAGE void st_quantize(const lc3d_plan* __restrict__ P, const lc3d_chan* __restrict__ C, LdsT& L, int lane, int mode, int target)
{
    mode = uni(mode); target = uni(target);       /* arguments arrive in vector registers: wave-uniform ones go to scalar registers, or every test of them is a masked region */
    const int nt = PI(ylen), fs = PI(fs), tb = CI(total_bits);
    const float offs = PI(hrmode) ? 0.5f : 0.375f;
    const float gain = unif(L.fsc[F_GAIN]);
    int* xq = XQ(L); uint32_t* cdw = CDW(L);
    SUB_BEGIN();
    /* xq = (int)(x / gain + offs * sign(x)), R/quantize_spec.c:37-47.  The division is an IEEE one per line in the reference; here the line
     * is first multiplied by 1 / gain (one division per frame): product and sum are within 3e-7 relative of the reference's, so the
     * truncation agrees unless the sum lies that close to an integer - then (a few lines per thousand frames) the line is divided. */
    const float rgain = 1.0f / gain;
    unsigned amb = 0;
#pragma unroll
    for (int k = 0; k < MAXN / WAVE + 1; k++) {
        const int i = lane + 64 * k;
        if (i < nt) {
            const float x = L.A[i];
            const float o = x > 0 ? offs : x < 0 ? -offs : 0.0f;
            const float v = x * rgain + o;
            xq[i] = (int)v;
            if (x != 0 && fabsf(v - rintf(v)) <= fabsf(v) * 5e-7f) amb |= 1u << k;
        }
    }
    if (__ballot(amb != 0)) {
#pragma unroll
        for (int k = 0; k < MAXN / WAVE + 1; k++) {
            if ((amb >> k) & 1u) { const int i = lane + 64 * k; const float x = L.A[i]; xq[i] = (int)truncf(x / gain + (x > 0 ? offs : -offs)); }
        }
    }
    int rate = 0;
    if ((fs < 48000 && tb > 320 + (fs / 8000 - 2) * 160) || (fs == 48000 && tb > 800)) rate = 512;
    if (mode == 0 && ((fs < 48000 && tb >= 640 + (fs / 8000 - 2) * 160) || (fs == 48000 && tb >= 1120))) mode = 1;
    LSYNC();
    int lp = 0;
#pragma unroll
    for (int k = 0; k < MAXN / 2 / WAVE + 1; k++) { const int p = lane + 64 * k; if (p < (nt >> 1) && p >= 1 && (xq[2 * p] != 0 || xq[2 * p + 1] != 0)) lp = p; }
    lp = uni(wave_max_i(lp));
    const int lastnz = lp >= 1 ? 2 * lp + 1 : 1;
    if (mode == -2) {                             /* lc3_enc_tail_kernel, gain unchanged: the lines only; the bit count of this quantisation came from lc3_enc_rate_kernel */
        if (lane == 0) { L.isc[I_LSB] = 0; L.isc[I_LASTNZ] = lastnz + 1; }
        LSYNC();
        return;
    }
    const int ntup = (lastnz + 1) >> 1;
    int lastnz2 = mode < 0 ? lastnz + 1 : 2;
    int nbits2 = 0, base = 0, nlsb = 0, ct1 = 0, ct2 = 0;
    SUB(28);
    /* Exact bit count of the 2-tuples in coding order.  Three phases over the (at most NP) groups of 64 tuples, so that the two dependent
     * rounds of table reads are paid once per call and not once per group:
     *   A  per tuple: escape count, final symbol, context (the two previous tuples' values: wave_shr:1, lanes 62 / 63 carry into the next
     *      group); reads of the context's model indices (one word for the four escape classes) and escape costs (plan tables q_lut4, q_esc);
     *   B  read of the final symbol's cost under the model of its class;
     *   C  bits of the tuple, running sum, truncation point (R/quantize_spec.c:150-166). */
    constexpr int NP = (MAXN / 2 + WAVE - 1) / WAVE;
    int meta[NP], mlev[NP]; unsigned lutw[NP]; uint2 escw[NP]; int bsym[NP];
    const uint32_t* __restrict__ qlut = P->q_lut4; const uint2* __restrict__ qesc = (const uint2*)P->q_esc; const uint16_t* __restrict__ qbits = P->q_bits;
    const int half = nt / 2;
#pragma unroll
    for (int k = 0; k < NP; k++) {
        meta[k] = 0; mlev[k] = 0; lutw[k] = 0; escw[k] = make_uint2(0, 0);
        if (64 * k < ntup) {
            const int p_ = 64 * k + lane; const bool act_ = p_ < ntup;
            const int pc_ = imin(p_, MAXN / 2 - 1);
            const int x0 = act_ ? xq[2 * pc_] : 0, x1 = act_ ? xq[2 * pc_ + 1] : 0;
            const int a_ = x0 < 0 ? -x0 : x0, b_ = x1 < 0 ? -x1 : x1, mx_ = imax(a_, b_);
            const int nsh = mx_ >= 4 ? ilog2((unsigned)mx_) - 1 : 0;
            const int af_ = a_ >> nsh, bf_ = b_ >> nsh;
            const int lev1 = imin(nsh, 3), s_ = af_ + bf_;
            const int tval_ = lev1 <= 1 ? 1 + (s_ << lev1) : 12 + lev1;
            const int t1_ = dpp_i<DPP_WSHR1>(ct1, tval_), t2_ = dpp_i<DPP_WSHR1>(ct2, t1_);
            ct2 = __builtin_amdgcn_readlane(tval_, 62); ct1 = __builtin_amdgcn_readlane(tval_, 63);
            int tin = 16 * (t2_ & 15) + t1_ + rate; if (2 * p_ > half) tin += 256;
            if (!act_) tin = 0;
            lutw[k] = qlut[tin]; escw[k] = qesc[tin];
            const int sh1 = nsh > 0 ? 1 : 0, am_ = a_ >> sh1, bm_ = b_ >> sh1;
            /* nsh | lev1 << 5 | sym << 7 | non-zero values << 11 | non-zero values without their first LSB << 13 | values that are only that LSB << 15 | act << 17 | tin << 18 */
            meta[k] = nsh | (lev1 << 5) | ((af_ + 4 * bf_) << 7) | ((imin(a_, 1) + imin(b_, 1)) << 11) | ((imin(am_, 1) + imin(bm_, 1)) << 13)
                      | (((am_ == 0 && a_ != 0) + (bm_ == 0 && b_ != 0)) << 15) | ((act_ ? 1 : 0) << 17) | (tin << 18);
            if constexpr (LdsT::CDW) mlev[k] = mx_ == 0 ? 0 : flog2f_int((unsigned)imax(mx_, 3));     /* max level + 1 */
        }
    }
#pragma unroll
    for (int k = 0; k < NP; k++) {
        bsym[k] = 0;
        if (64 * k < ntup) {
            const int lev1 = (meta[k] >> 5) & 3, sym = (meta[k] >> 7) & 15;
            const int pkf = (int)((lutw[k] >> (8 * lev1)) & 255u);
            bsym[k] = qbits[pkf * 17 + sym];
        }
    }
#pragma unroll
    for (int k = 0; k < NP; k++) {
        if (64 * k < ntup) {
            const int m_ = meta[k], nsh = m_ & 31, act = (m_ >> 17) & 1;
            const unsigned w_ = nsh >= 3 ? escw[k].y : escw[k].x;
            const int cv = (int)(nsh == 2 ? w_ >> 16 : w_ & 0xffffu), e3 = (int)(escw[k].y >> 16);
            int bits = nsh == 0 ? 0 : cv + e3 * imax(nsh - 3, 0);
            int lsbc = 0;
            if (mode <= 0) bits += ((m_ >> 11) & 3) * 2048 + 4096 * nsh;
            else { bits += ((m_ >> 13) & 3) * 2048 + (nsh > 0 ? 4096 * (nsh - 1) : 0); lsbc = (nsh > 0 ? 2 : 0) + ((m_ >> 15) & 3); }
            bits += bsym[k];
            if (!act) { bits = 0; lsbc = 0; }
            if constexpr (LdsT::CDW) {
                if (act) {
                    const int cls_f = imin(imax(mlev[k] - 1, 0), 3);          /* class of the final symbol in the coder (R/ari_codec.c:723-727) */
                    const unsigned pkc = (lutw[k] >> (8 * cls_f)) & 255u;
                    cdw[64 * k + lane] = (uint32_t)(m_ >> 18) | ((uint32_t)mlev[k] << 10) | (pkc << 16) | ((uint32_t)((m_ >> 7) & 15) << 22);
                }
            }
            const int incl = wave_incl_scan_i(bits, lane) + base;
            const unsigned long long ok = __ballot(act && mode >= 0 && ((m_ >> 11) & 3) != 0 && incl <= target * 2048);
            if (ok) { const int hl = 63 - __clzll((long long)ok); lastnz2 = 2 * (64 * k + hl) + 2; nbits2 = __builtin_amdgcn_readlane(incl, hl); }
            base = __builtin_amdgcn_readlane(incl, 63);
            if (mode > 0) nlsb += wave_sum_i(lsbc);
        }
    }
    SUB(29);
    int nbits = (base + 2047) >> 11;
    if (mode >= 0) nbits2 = (nbits2 + 2047) >> 11; else nbits2 = nbits;
    if (mode > 0) { nbits += nlsb; nbits2 += nlsb; }
    LSYNC();
    for (int i = lastnz2 + lane; i <= lastnz; i += WAVE) xq[i] = 0;
    if (lane == 0) { L.isc[I_LSB] = (mode > 0 && nbits > target) ? 1 : 0; L.isc[I_LASTNZ] = lastnz2; L.isc[I_NBITS] = nbits; L.isc[I_NBITS2] = nbits2; }
    LSYNC();
}

/* ---- R/adjust_global_gain.c:13-50 (wave-uniform scalars) ---- */
template <class LdsT> __device__ __forceinline__ void gain_adjust(const lc3d_plan* __restrict__ P, LdsT& L, int& gg, int gg_min, float& gain, int target, int nBits, int& change)
{
    const int f = PI(fs_idx), off = CI(gg_off);
    float delta;
    if (nBits < lc3t_gg_p1[f]) delta = (float)(((double)nBits + 48.0) / 16.0);
    else if (nBits < lc3t_gg_p2[f]) delta = ((float)nBits + lc3t_gg_d[f]) * lc3t_gg_c[f];
    else if (nBits < lc3t_gg_p3[f]) delta = (float)((double)nBits / 48.0);
    else delta = (float)((double)lc3t_gg_p3[f] / 48.0);
    delta = (float)round((double)delta);
    const int delta2 = (int)(delta + 2);
    change = 0;
    if (gg == 255 && nBits > target) change = 1;
    if ((gg < 255 && nBits > target) || (gg > 0 && nBits < target - delta2)) {
        if (nBits < target - delta2) gg = gg - 1;
        else if (gg == 254 || (float)nBits < (float)target + delta) gg = gg + 1;
        else gg = gg + 2;
        gg = imax(gg, gg_min - off);
        gain = P->gain_adj[gg + off + 256];
        change = 1;
    }
}

/* ---- noise factor R/noise_factor.c:13-108 ----
 * Pass 1 finds the zero lines (ballots) and their count / index sum; pass 2 accumulates |x/gg| over them in index order,
 * the serial float sums fed from lane registers with readlane (no list in LDS). */
template <class LdsT> STAGE void st_noise_factor(const lc3d_plan* __restrict__ P, LdsT& L, int lane, int bw_bin)
{
    bw_bin = uni(bw_bin);
    const int dms = PI(dms);
    const int width = dms == 100 ? 8 : 4, first = dms == 100 ? 24 : dms == 50 ? 12 : 6, hw = (width - 2) / 2;
    const float gg = unif(L.fsc[F_GAIN]);
    const int* xq = XQ(L);
    SUB_BEGIN();
    int nz = 0;
    /* nzb[c]: non-zero lines among bins 64c .. 64c+63 below the bandwidth cut-off (R/noise_factor.c:47-60 looks at xq only there) */
    unsigned long long nzb[10];
#pragma unroll
    for (int c = 0; c < 8; c++) {
        const int k = 64 * c + lane;
        const int q = xq[k];                        /* k < 512: in bounds of the LDS slice; lines >= bw_bin are masked */
        nzb[c] = __ballot(k < bw_bin && q != 0);
    }
    nzb[8] = nzb[9] = 0;
    /* zm[c]: zero-line masks (lines whose +-hw neighbourhood is all zero) of chunk c = bins first + 64c .. +63.  The 2hw+1 window
     * bits of lane l start at bit (first - hw) + 64c + l of the 512-bit mask: a uniform 128-bit pre-shift, then a per-lane shift. */
    unsigned long long zm[8];
    const int o = first - hw; const unsigned wm = (1u << (2 * hw + 1)) - 1u;
#pragma unroll
    for (int c = 0; c < 8; c++) {
        zm[c] = 0;
        if (first + 64 * c < bw_bin) {
            const unsigned long long lo64 = (nzb[c] >> o) | (nzb[c + 1] << (64 - o)), hi64 = (nzb[c + 1] >> o) | (nzb[c + 2] << (64 - o));
            const unsigned w = (unsigned)(lo64 >> lane) | (unsigned)((hi64 << 1) << (63 - lane));
            zm[c] = __ballot(first + 64 * c + lane < bw_bin && (w & wm) == 0);
            nz += __popcll(zm[c]);
        }
    }
    SUB(31);
    int sumz = nz;                                  /* only its sign matters unless the spectrum is split (below) */
    const bool split = CI(nbytes) <= 20 && dms == 100 && nz > 0;
    if (split) {
        sumz = 0;
#pragma unroll
        for (int c = 0; c < 8; c++) sumz += wave_sum_i(((zm[c] >> lane) & 1ull) ? first + 64 * c + lane + 1 : 0);
    }
    const int msplit = split ? sumz / nz : 0x7fffffff;
    float m1 = 0, m2 = 0; int j1 = 0;
    /* The level index is round(8 - 16 * mean) clamped to 0..7, the mean an ORDERED float sum of up to ~400 non-negative terms: first a
     * lane-parallel tree sum; the serial sum of n terms differs from it by less than (n + 10) * 2^-24 of the sum (see st_gain_estimate),
     * so unless 8 - 16 * mean comes within that of a rounding boundary the index is decided without the serial chain. */
    bool decided = false; float idx_fast = 0;
    if (!split && nz > 0) {
        float ts = 0;
#pragma unroll
        for (int c = 0; c < 8; c++) { const int k = first + 64 * c + lane; if ((zm[c] >> lane) & 1ull) ts += fabsf(L.A[k] / gg); }
        const float S = wave_sum_f_tree(ts);
        const float v = 8.0f - 16.0f * (S / (float)nz), d = 16.0f * (S / (float)nz) * 6e-5f + 2e-5f;
        float a = (float)round((double)(v - d)), b = (float)round((double)(v + d));
        a = a > 0 ? a : 0; a = a < 7 ? a : 7; b = b > 0 ? b : 0; b = b < 7 ? b : 7;
        decided = a == b; idx_fast = a;
    }
    if (decided) {
        if (lane == 0) L.isc[I_FACNS] = (int)idx_fast;
        LSYNC();
        SUB(32);
        return;
    }
    if (!split) {
        float* lst = &L.sm[L.LSTO];               /* >= 208 free words (16-byte aligned): the residual-bit area is not in use yet */
        j1 = nz;
#pragma unroll
        for (int g = 0; g < 8; g += 3) {
            int cntg = 0;
#pragma unroll
            for (int c = g; c < g + 3 && c < 8; c++) {
                const int k = first + 64 * c + lane;
                const unsigned long long m = zm[c];
                if ((m >> lane) & 1ull) lst[cntg + __popcll(m & ((1ull << lane) - 1ull))] = fabsf(L.A[k] / gg);
                cntg += __popcll(m);
            }
            if (lane < 8) lst[cntg + lane] = 0.0f;  /* pad to a whole block: adding +0 is exact */
            LSYNC();
            for (int j = 0; j < cntg; j += 8) {
                const float4 u = *(const float4*)&lst[j], v = *(const float4*)&lst[j + 4];
                m1 += u.x; m1 += u.y; m1 += u.z; m1 += u.w; m1 += v.x; m1 += v.y; m1 += v.z; m1 += v.w;
            }
            LSYNC();
        }
    } else
#pragma unroll
    for (int c = 0; c < 8; c++) {
        const int k = first + 64 * c + lane;
        const float v = k < bw_bin ? fabsf(L.A[k] / gg) : 0.0f;
        unsigned long long m = zm[c];
        const int kbase = first + 64 * c;
        while (m) {
            const int b = __ffsll((long long)m) - 1; m &= m - 1;
            const float t = rl_f(v, b);
            if (kbase + b + 1 <= msplit) { m1 += t; j1++; } else m2 += t;
        }
    }
    float fac = 0;
    if (sumz > 0) fac = m1 / (float)nz;              /* without the split every line is in the "low" group */
    if (split) { const float n1 = m1 / (float)j1, n2 = m2 / (float)(nz - j1); fac = n1 < n2 ? n1 : n2; }
    float idx = (float)round((double)(8 - 16 * fac));
    { const float t = idx > 0 ? idx : 0; idx = t < 7 ? t : 7; }
    if (lane == 0) L.isc[I_FACNS] = (int)idx;
    LSYNC();
    SUB(32);
}

/* ---- residual coding R/residual_coding.c:13-75 ----
 * The n-th non-zero coefficient (in bin order) owns residual bit n; a ballot prefix count gives n, so every lane decides and
 * stores its own bit.  High-resolution mode repeats the sweep with a halved offset (up to 20 times). */
template <class LdsT> STAGE void st_residual(const lc3d_plan* __restrict__ P, LdsT& L, int lane, int targetBits, int nBits)
{
    const int* xq = XQ(L);
    const float gain = unif(L.fsc[F_GAIN]);
    unsigned* res = (unsigned*)RESB(L);
    const int ylen = PI(ylen), hr = PI(hrmode);
    int m = targetBits - nBits + 4;
    if (hr) m += 10;
    const int iter_max = hr ? 20 : 1;
    for (int i = lane; i < 160; i += WAVE) res[i] = 0;
    LSYNC();
    int n = 0, iter = 0; float offset = .25f;
    while (iter < iter_max && n < m) {
        for (int k0 = 0; k0 < ylen && n < m; k0 += WAVE) {
            const int k = k0 + lane;
            const int q = k < ylen ? xq[k] : 0;
            const unsigned long long mk = __ballot(q != 0);
            const int pos = n + __popcll(mk & ((1ull << lane) - 1ull));
            if (q != 0 && pos < m) {
                const float x = L.A[k];
                if (x >= (float)q * gain) { atomicOr(&res[pos >> 5], 1u << (pos & 31)); L.A[k] = x - gain * offset; }
                else L.A[k] = x + gain * offset;
            }
            n = imin(m, n + __popcll(mk));
        }
        iter++; offset *= .5f;
    }
    LSYNC();
    if (lane == 0) L.isc[I_NRES] = n;
    LSYNC();
}

/* ---- bitstream: side information R/enc_entropy.c:13-115, range coder R/ari_codec.c:511-800 (integer, bit-exact).
 *
 * The frame has two cursors: the range coder writes bytes FORWARD from byte 0, everything else (side info, escape LSBs,
 * signs, residual) is written BACKWARD bit by bit from the last byte.  The backward stream does not depend on the coder
 * state, so it is assembled in parallel: every lane builds the bit string of its 2-tuple, a wave scan gives the bit
 * offsets and the strings are OR-ed into LDS.  Only the range coder itself is serial; it runs on wave-uniform values
 * (readlane -> scalar registers), its output bytes are collected with writelane and stored 64 at a time. ---- */

/* OR the n (<= 56) low bits of v into the backward stream at bit position q (bit 0 = LSB of the last frame byte) */
/* The backward part of a frame (side information, escape LSBs, signs, residual bits) is written from the last byte towards the
 * first, LSB first (write_bit_backward_fl, R/enc_entropy.c:101-115).  Seen from the end of the frame that is one little-endian
 * bit string: backward bit q is bit (q & 31) of word q >> 5 of `rb`, whose byte j is frame byte nbytes-1-j.  OR n (<= 64) bits. */
__device__ __forceinline__ void or_bits_back(unsigned* rb, int q, unsigned long long v, int n)
{
    if (n <= 0) return;
    v &= (n >= 64) ? ~0ull : ((1ull << n) - 1ull);
    const int w = q >> 5, sh = q & 31;
    const unsigned long long lo = v << sh;
    const unsigned hi = sh ? (unsigned)(v >> (64 - sh)) : 0u;
    if ((unsigned)lo) atomicOr(&rb[w], (unsigned)lo);
    if ((unsigned)(lo >> 32)) atomicOr(&rb[w + 1], (unsigned)(lo >> 32));
    if (hi) atomicOr(&rb[w + 2], hi);
}
/* OR n (<= 32) bits into a forward little-endian bit buffer (LSB mode list) */
__device__ __forceinline__ void or_bits_fwd(uint8_t* buf, int q, unsigned v, int n)
{
    if (n <= 0) return;
    const unsigned long long V = (unsigned long long)v << (q & 31);
    unsigned* W = (unsigned*)buf;
    if ((unsigned)V) atomicOr(&W[q >> 5], (unsigned)V);
    if ((unsigned)(V >> 32)) atomicOr(&W[(q >> 5) + 1], (unsigned)(V >> 32));
}

/* ---- range coder, restated for a wavefront (R/ari_codec.c:511-660) ----
 * The reference's coder carries (low, range, cache, carry, carry_count, bp) from symbol to symbol.  Only `range` is inherently
 * serial: range' = (range >> 10) * freq, renormalised by whole bytes.  `low` is a sum: symbol j adds c_j = (range_j >> 10) * cum_j
 * at the byte position s_j (the number of renormalisation shifts so far), and cache / carry / carry_count are just a streaming
 * carry-propagation over that sum.  So:
 *   serial   (SALU, ~9 instructions per symbol): the range chain; r_j = range_j >> 10 is left in lane j of a VGPR;
 *   parallel (one lane per symbol): s_j by a prefix sum of the shift counts, c_j = r_j * cum_j added into the code value, a
 *            big-endian byte string held as a multi-word integer in LDS (atomic add, carries rippled by the lane that caused them).
 * Finalisation (ac_finalize_fl) is then an add of (val - low) at the window plus the top `bits` bits of the result. */
#define BIGW 162                                 /* words of the code value: stream byte p is byte (BIGB - 1 - p) of big[] */
#define BIGB (4 * BIGW)
#define BIG(L)  ((unsigned*)&(L).A[160])         /* BIGW + 1 words (guard word on top) */
#define SYML(L) ((unsigned*)&(L).A[324])         /* flattened symbol list of one group of tuples: cum | freq << 16 */
#define SYMCAP 156

/* v_writelane_b32 with a variable lane select: two different SGPR operands would break the constant-bus limit, M0 is exempt */
__device__ __forceinline__ int writelane(int vec, int val, int lane_sel)
{
    asm volatile("s_mov_b32 m0, %2\n\tv_writelane_b32 %0, %1, m0" : "+v"(vec) : "s"(val), "s"(lane_sel));   /* M0 is a reserved scratch register: the compiler never keeps a value in it across other code */
    return vec;
}

/* add v (< 2^25) into the code value with its bits 23..16 at stream byte s */
__device__ __forceinline__ void big_add(unsigned* big, unsigned v, int s)
{
    const int q = 8 * (BIGB - 3 - s); int w = q >> 5;
    const unsigned long long V = (unsigned long long)v << (q & 31);
    unsigned add = (unsigned)V;
    unsigned old = atomicAdd(&big[w], add);
    add = (unsigned)(V >> 32) + ((old + add) < add ? 1u : 0u); w++;
    while (add) { old = atomicAdd(&big[w], add); add = (old + add) < add ? 1u : 0u; w++; }
}

struct AriSt { int range, s8; int pa; };         /* range, 8 * shifts so far; pa (per lane): the c_j this lane added since the last shift */

template <int LANE> __device__ __forceinline__ int writelane_c(int vec, int val)
{ asm volatile("v_writelane_b32 %0, %1, %2" : "+v"(vec) : "s"(val), "n"(LANE)); return vec; }

/* code up to 64 symbols: lane j < cnt holds symbol j as cum | freq << 16.
 * Serial part, per symbol: r = range >> 10; range = r * freq, shifted left by whole bytes until >= 2^16.  `R` carries range << 8, so
 * the renormalising shift is simply clz(r * freq) & 24 (clz 8..15 -> 8, 16..23 -> 16, 24..25 -> 24).  Groups of eight symbols run
 * with compile-time lane numbers; the rest of a chunk takes the generic loop. */
__device__ __forceinline__ void ari_chunk(AriSt& w, unsigned* big, int lane, unsigned vsym, int cnt)
{
    const int vfreq = (int)(vsym >> 16), vcum = (int)(vsym & 0xffffu);
    int vr = 0, k = 0;
    unsigned R = (unsigned)w.range << 8;
#define ARI_STEP_C(LN) do { const int f_ = __builtin_amdgcn_readlane(vfreq, (LN)); const int r_ = (int)(R >> 18); \
        vr = writelane_c<(LN)>(vr, r_); const unsigned rp_ = (unsigned)(r_ * f_); R = rp_ << (__builtin_clz(rp_) & 24); } while (0)
#define ARI_GROUP(G) if (cnt >= 8 * (G) + 8) { ARI_STEP_C(8 * (G)); ARI_STEP_C(8 * (G) + 1); ARI_STEP_C(8 * (G) + 2); ARI_STEP_C(8 * (G) + 3); \
        ARI_STEP_C(8 * (G) + 4); ARI_STEP_C(8 * (G) + 5); ARI_STEP_C(8 * (G) + 6); ARI_STEP_C(8 * (G) + 7); k = 8 * (G) + 8;
    ARI_GROUP(0) ARI_GROUP(1) ARI_GROUP(2) ARI_GROUP(3) ARI_GROUP(4) ARI_GROUP(5) ARI_GROUP(6) ARI_GROUP(7) }}}}}}}}
#undef ARI_GROUP
#undef ARI_STEP_C
    for (; k < cnt; k++) {
        const int f_ = __builtin_amdgcn_readlane(vfreq, k); const int r_ = (int)(R >> 18);
        vr = writelane(vr, r_, k); const unsigned rp_ = (unsigned)(r_ * f_); R = rp_ << (__builtin_clz(rp_) & 24);
    }
    const int rg = (int)(R >> 8);
    w.range = rg;
    const bool on = lane < cnt;
    const int rng2 = on ? vr * vfreq : 0x10000;                         /* this symbol's range before renormalisation */
    const int n8 = (__builtin_clz(rng2) - 8) & 24;
    const int incl = wave_incl_scan_i(n8, lane);
    const int tot = __builtin_amdgcn_readlane(incl, 63);
    const int sk8 = w.s8 + incl - n8, s8_end = w.s8 + tot;
    const unsigned c = on ? (unsigned)(vr * vcum) : 0u;
    if (c) big_add(big, c, sk8 >> 3);
    w.pa = (tot == 0 ? w.pa : 0) + (sk8 == s8_end ? (int)c : 0);
    w.s8 = s8_end;
}

template <class LdsT> STAGE void st_bitstream(const lc3d_plan* __restrict__ P, const lc3d_chan* __restrict__ C, LdsT& L, int lane)
{
    int* isc = L.isc;
    unsigned* rb = (unsigned*)BYTES(L);              /* backward bit string while coding; the finished frame at the end */
    unsigned* big = BIG(L); unsigned* syml = SYML(L);
    SUB_BEGIN();
    const int nbytes = CI(nbytes);
    /* the spectrum in A is dead from here on: clear the frame (whole words up to nbytes) and the words of the code value a
     * frame of nbytes can reach */
    for (int i = lane; i < (nbytes >> 2) + 3; i += WAVE) { rb[i] = 0; big[BIGW - i] = 0; }
    LSYNC();
    const int nfilt = uni(isc[I_TNS_NF]);
    const int lastnz = uni(isc[I_LASTNZ]), lsbMode = uni(isc[I_LSB]), nres = uni(isc[I_NRES]);
    const int bw_idx = uni(isc[I_BW]), gg = uni(isc[I_GG]), fac_ns = uni(isc[I_FACNS]);
    const int* xq = XQ(L); const uint32_t* cdw = CDW(L); uint8_t* resb = RESB(L);
    /* ---- side information, R/enc_entropy.c:25-87: lane i forms field i (value, width); a prefix sum places it ---- */
    int Q;
    {
        const int s2 = isc[I_SCF2], s3 = isc[I_SCF3], s5 = isc[I_SCF5], s6 = isc[I_SCF6], ltpf0 = isc[I_LTPF0];
        const int sub_msb = s2 >> 1, sub_lsb = s2 & 1, g_lsb = s3 & 1;
        int joint, jbits;
        if (sub_msb == 0) { joint = (sub_lsb == 0 ? s6 + 2 : g_lsb) * 2390004 + s5; jbits = 25; }
        else { joint = sub_lsb != 0 ? 2 * s5 + g_lsb + 15158272 : s5; jbits = 24; }
        const int o0 = isc[I_TNS_ORD0], o1 = isc[I_TNS_ORD1], sc0 = isc[I_SCF0], sc1 = isc[I_SCF1], sc4 = isc[I_SCF4], lt1 = isc[I_LTPF1], lt2 = isc[I_LTPF2];
        int v = 0, n = 0;
#define FIELD(k_, val_, width_) do { const bool m_ = lane == (k_); v = m_ ? (val_) : v; n = m_ ? (width_) : n; } while (0)
        FIELD(0, bw_idx, PI(bw_bits));
        FIELD(1, lastnz / 2 - 1, ilog2((unsigned)(PI(ylen) / 2 - 1)) + 1);
        FIELD(2, lsbMode, 1);
        FIELD(3, gg, 8);
        FIELD(4, imin(1, o0), nfilt > 0 ? 1 : 0);
        FIELD(5, imin(1, o1), nfilt > 1 ? 1 : 0);
        FIELD(6, ltpf0, 1);
        FIELD(7, sc0, 5);
        FIELD(8, sc1, 5);
        FIELD(9, sub_msb, 1);
        FIELD(10, s3 >> (s2 & 1), 1 + sub_msb);                        /* gain msbs: 1, 1, 2, 2 bits for sub-modes 0..3 */
        FIELD(11, sc4, 1);
        FIELD(12, joint, jbits);
        FIELD(13, lt1, ltpf0 == 1 ? 1 : 0);
        FIELD(14, lt2, ltpf0 == 1 ? 9 : 0);
        FIELD(15, fac_ns, 3);
#undef FIELD
        const int incl = wave_incl_scan_i(n, lane);
        or_bits_back(rb, incl - n, (unsigned long long)(unsigned)v, n);
        Q = __builtin_amdgcn_readlane(incl, 63);
    }
    if (lane == 0) { isc[I_BP_SIDE] = nbytes - 1 - (Q >> 3); isc[I_MASK_SIDE] = 1 << (Q & 7); }
    SUB(15);

    /* ---- range coder: TNS symbols (order, then coefficients, per filter) ---- */
    AriSt w; w.range = 0xFFFFFF; w.s8 = 0; w.pa = 0;
    {
        const int ord0 = nfilt > 0 ? uni(isc[I_TNS_ORD0]) : 0, ord1 = nfilt > 1 ? uni(isc[I_TNS_ORD1]) : 0;
        const int n0 = ord0 > 0 ? ord0 + 1 : 0, n1 = ord1 > 0 ? ord1 + 1 : 0, nt = n0 + n1;
        if (nt > 0) {
            const int f = lane < n0 ? 0 : 1, j = (f ? lane - n0 : lane) - 1, ord = f ? ord1 : ord0;
            unsigned sv = 0;
            if (lane < nt) sv = j < 0 ? lc3t_tns_order_sym[CI(lpc_weighting) * 8 + ord - 1] : lc3t_tns_coef_sym[j * 17 + isc[I_TNS_IDX0 + f * 8 + j]];
            ari_chunk(w, big, lane, sv, nt);
        }
    }
    SUB(16);
    /* ---- spectrum: 64 tuples per pass ---- */
    const int ntup = (lastnz + 1) >> 1;
    int nl = 0;                                   /* LSB-mode list length */
    /* The table reads of a pass (final symbol; escape models: context LUT, then the symbol) have long latencies and feed only the
     * symbol list, so they are issued one pass ahead: pass p+1's reads are in flight while pass p is coded. */
    uint32_t cdvN = 0; unsigned fsymN = 0; int x0N = 0, x1N = 0, lu0N = 0, lu1N = 0, lu2N = 0, lu3N = 0;
#define LOAD_PASS(c0_) do { const int p_ = (c0_) + lane; const bool act_ = p_ < ntup; \
        cdvN = act_ ? cdw[p_] : 0u; x0N = act_ ? xq[2 * p_] : 0; x1N = act_ ? xq[2 * p_ + 1] : 0; \
        fsymN = act_ ? lc3t_ac_sym[((cdvN >> 16) & 63) * 17 + ((cdvN >> 22) & 31)] : 0u; \
        const int ctx_ = cdvN & 1023, ne_ = act_ ? (int)((cdvN >> 10) & 63) - 1 : 0; \
        if (__ballot(ne_ > 0)) { if (ne_ > 0) lu0N = lc3t_ac_ctx_lut[ctx_]; \
            if (__ballot(ne_ > 1)) { if (ne_ > 1) lu1N = lc3t_ac_ctx_lut[ctx_ + 1024]; \
                if (__ballot(ne_ > 2)) { if (ne_ > 2) lu2N = lc3t_ac_ctx_lut[ctx_ + 2048]; if (ne_ > 3) lu3N = lc3t_ac_ctx_lut[ctx_ + 3072]; } } } } while (0)
    if (ntup > 0) LOAD_PASS(0);
    for (int c0 = 0; c0 < ntup; c0 += WAVE) {
        const int p = c0 + lane; const bool act = p < ntup;
        const uint32_t cdv = cdvN; const unsigned fsym = fsymN; const int x0 = x0N, x1 = x1N;
        const int a0 = x0 < 0 ? -x0 : x0, b0 = x1 < 0 ? -x1 : x1;
        const int maxlev = act ? (int)((cdv >> 10) & 63) - 1 : -1;
        const int nesc = maxlev > 0 ? maxlev : 0;
        /* escape symbol (16) of the models of level classes 0..3 */
        unsigned e0 = 0, e1 = 0, e2 = 0, e3 = 0;
        if (__ballot(nesc > 0)) {
            if (nesc > 0) e0 = lc3t_ac_sym[lu0N * 17 + 16];
            if (__ballot(nesc > 1)) {
                if (nesc > 1) e1 = lc3t_ac_sym[lu1N * 17 + 16];
                if (__ballot(nesc > 2)) { if (nesc > 2) e2 = lc3t_ac_sym[lu2N * 17 + 16]; if (nesc > 3) e3 = lc3t_ac_sym[lu3N * 17 + 16]; }
            }
        }
        if (c0 + WAVE < ntup) LOAD_PASS(c0 + WAVE);
        /* backward bits of this tuple: escape LSB pairs, then signs (R/ari_codec.c:700-757) */
        unsigned long long bits = 0; int n = 0;
        for (int lev = 0; lev < maxlev; lev++) {
            if (!(lsbMode == 1 && lev == 0)) {
                bits |= (unsigned long long)((a0 >> lev) & 1) << n; n++;
                bits |= (unsigned long long)((b0 >> lev) & 1) << n; n++;
            }
        }
        int a = a0, b = b0; unsigned lsbv = 0; int ln = 0;
        if (lsbMode == 1 && maxlev > 0) {
            a >>= 1; lsbv |= (unsigned)(a0 & 1) << ln; ln++;
            if (a == 0 && x0 != 0) { lsbv |= (unsigned)(x0 < 0) << ln; ln++; }
            b >>= 1; lsbv |= (unsigned)(b0 & 1) << ln; ln++;
            if (b == 0 && x1 != 0) { lsbv |= (unsigned)(x1 < 0) << ln; ln++; }
        }
        if (a != 0) { bits |= (unsigned long long)(x0 < 0) << n; n++; }
        if (b != 0) { bits |= (unsigned long long)(x1 < 0) << n; n++; }
        const int incl = wave_incl_scan_i(n, lane);
        or_bits_back(rb, Q + incl - n, bits, n);
        Q += __builtin_amdgcn_readlane(incl, 63);
        if (lsbMode == 1) {
            const int li = wave_incl_scan_i(ln, lane);
            or_bits_fwd(resb, nl + li - ln, lsbv, ln);
            nl += __builtin_amdgcn_readlane(li, 63);
        }
        /* flatten the pass into the symbol list: per tuple its escape symbols in level order, then the final symbol.  The list
         * holds SYMCAP symbols; a pass with more (many escapes) goes through it in groups of 16 or 4 tuples. */
        const int ns = act ? 1 + nesc : 0;
        const int sincl = wave_incl_scan_i(ns, lane);
        const int M = __builtin_amdgcn_readlane(sincl, 63);
        int G = WAVE;
        if (M > SYMCAP) {
            const int q0 = __builtin_amdgcn_readlane(sincl, 15), q1 = __builtin_amdgcn_readlane(sincl, 31), q2 = __builtin_amdgcn_readlane(sincl, 47);
            G = (q0 <= SYMCAP && q1 - q0 <= SYMCAP && q2 - q1 <= SYMCAP && M - q2 <= SYMCAP) ? 16 : 4;
        }
        SUB(17);
        const int cntT = imin(WAVE, ntup - c0);
        for (int g0 = 0; g0 < cntT; g0 += G) {
            const int gl = g0 + G - 1;
            const int base = g0 ? __builtin_amdgcn_readlane(sincl, g0 - 1) : 0, Mg = __builtin_amdgcn_readlane(sincl, gl) - base;
            if (act && lane >= g0 && lane <= gl) {
                const int o = sincl - ns - base;
                for (int lev = 0; lev < nesc; lev++) syml[o + lev] = lev == 0 ? e0 : lev == 1 ? e1 : lev == 2 ? e2 : e3;
                syml[o + nesc] = fsym;
            }
            LSYNC();
            for (int k0 = 0; k0 < Mg; k0 += WAVE) {
                const int cnt = imin(WAVE, Mg - k0);
                ari_chunk(w, big, lane, lane < cnt ? syml[k0 + lane] : 0u, cnt);
            }
            LSYNC();
        }
        SUB(18);
    }
#undef LOAD_PASS
    /* ---- residual / LSB bits (R/ari_codec.c:764-797); bp + pending bytes of the reference = number of shifts ---- */
    const int total = CI(total_bits);
    const int bp_side = nbytes - 1 - (Q >> 3), mask_log = Q & 7;
    const int nbits_side = total - (8 * (bp_side + 1) + 8 - mask_log);
    const int S = w.s8 >> 3;
    const int nbits_ari = 8 * S + 33 - flog2f_int((unsigned)w.range);
    int nres_enc = total - (nbits_side + nbits_ari);
    if (lane == 0) isc[I_BUDGET] = nres_enc < 0 ? 1 : 0;      /* R/ari_codec.c:777 asserts this; the caller flags the frame */
    nres_enc = imin(nres_enc, lsbMode == 0 ? nres : nl);
    LSYNC();                                        /* LSB list / residual bits and the code value are complete */
    for (int k0 = 32 * lane; k0 < nres_enc; k0 += 32 * WAVE) {
        const unsigned wv = ((const unsigned*)resb)[k0 >> 5];
        or_bits_back(rb, Q + k0, wv, imin(32, nres_enc - k0));
    }
    SUB(19);
    /* ---- finalise (R/ari_codec.c:573-647) ---- */
    const uint8_t* bb = (const uint8_t*)big;
    const int low = (bb[BIGB - 1 - S] << 16) | (bb[BIGB - 2 - S] << 8) | bb[BIGB - 3 - S];
    /* carry flag of the reference at this point: did the adds since the last shift leave the 24-bit window? */
    const unsigned a0 = (unsigned)wave_sum_i(w.pa);       /* sum of the c_j added at the final shift count: < 2^25 */
    const int c_pending = (int)(((((unsigned)low - a0) & 0xFFFFFFu) + a0) >> 24);
    int bits = 24 - flog2f_int((unsigned)w.range);
    int mask = 0xFFFFFF >> bits, val = low + mask; const int over1 = val >> 24;
    val &= 0xFFFFFF;
    int high = low + w.range; const int over2 = high >> 24;
    high &= 0xFFFFFF;
    val &= (0xFFFFFF - mask);
    int cf = 0;
    if (over1 == over2) {
        if (val + mask >= high) { bits++; mask >>= 1; val = ((low + mask) & 0xFFFFFF) & (0xFFFFFF - mask); }
        if (val < low) cf = 1;
    }
    LSYNC();                                        /* `low` has been read by every lane */
    if (lane == 0) { const unsigned D = (unsigned)(val - low + (cf << 24)); if (D) big_add(big, D, S); }
    /* the final shifts: 1, or ceil(bits / 8) when bits > 8; nb = bits of the last byte that belong to the coder.  The reference
     * ends in its carry_count > 0 branch iff the last shift found 0xFF on top of `low` with no carry pending; it then takes the
     * bits from 255 << (nb - 8), a negative shift count that the hardware reduces mod 32: nb < 8 yields zeros. */
    int n_f = 1, nb = bits;
    if (bits > 8) { n_f = (bits + 7) >> 3; nb = bits - 8 * n_f; }
    if (nb < 0) nb += 8;
    bool ff_last = false;
    { int lv = val, cin = c_pending | cf;
      for (int i = 0; i < n_f; i++) { ff_last = (lv >= 0xFF0000) && cin == 0; cin = 0; lv = (lv << 8) & 0xFFFFFF; } }
    const int F = S + n_f;                          /* forward bytes, the last one partial */
    unsigned lastm = nb ? (0xFFu << (8 - nb)) & 0xFFu : 0u;
    if (ff_last && nb < 8) lastm = 0;
    LSYNC();
    /* assemble the frame in place: word k = frame bytes 4k..4k+3 = forward bytes (code value, byte-reversed words of big[]) OR
     * backward bytes (rb bytes nbytes-1-4k .. nbytes-4-4k, reversed).  Every lane reads all its sources before anyone writes. */
    {
        const uint8_t* rbb = (const uint8_t*)rb;
        unsigned outw[3]; const int nw = (nbytes + 3) >> 2;
#pragma unroll
        for (int r = 0; r < 3; r++) {
            const int k = lane + WAVE * r;
            unsigned o = 0;
            if (k < nw) {
                if (4 * k < F) {
                    const unsigned fw = __builtin_bswap32(big[BIGW - 1 - k]);
                    const int full = imin(4, imax(0, F - 1 - 4 * k));
                    unsigned mw = full >= 4 ? 0xFFFFFFFFu : ((1u << (8 * full)) - 1u);
                    if (full < 4 && 4 * k + full == F - 1) mw |= lastm << (8 * full);
                    o = fw & mw;
                }
                /* rb bytes at offsets b0 .. b0+3 (b0 = nbytes-4-4k, may be -3..-1 for the last word: those frame bytes do not exist) */
                const int b0 = nbytes - 4 - 4 * k;
                const int wq = b0 >> 2, sh = 8 * (b0 & 3);                 /* arithmetic shift: wq = -1 for negative b0 */
                const unsigned w0 = wq >= 0 ? rb[wq] : 0u, w1 = rb[wq + 1];
                const unsigned v = sh ? (w0 >> sh) | (w1 << (32 - sh)) : w0;
                o |= __builtin_bswap32(v);
            }
            outw[r] = o;
        }
        (void)rbb;
        LSYNC();
#pragma unroll
        for (int r = 0; r < 3; r++) { const int k = lane + WAVE * r; if (k < nw) rb[k] = outw[r]; }
    }
    LSYNC();
    SUB(20);
}


/* ------------------------------------------------------------------------------------------------ */
/* the kernel: one wave per channel-stream, frames in time order  (frame driver R/enc_lc3_fl.c:13-160) */
/* ------------------------------------------------------------------------------------------------ */
extern "C" __global__ void __launch_bounds__(WAVE) __attribute__((amdgpu_waves_per_eu(KERNEL_WAVES, KERNEL_WAVES)))
KERNEL_NAME(const lc3d_plan* __restrict__ P, const lc3d_chan* __restrict__ chans, float* __restrict__ state,
                  const void* __restrict__ pcm, int bitdepth, int T, uint8_t* __restrict__ out, int out_stride, int ncs,
                  lc3d_trace* __restrict__ trace, int* __restrict__ dump /* [cs][T][dstride] hand-over to lc3_enc_pack_kernel, or null: write the bytes here */, int dstride,
                  const float* __restrict__ y12 /* [cs][T][128] HP-filtered 12.8 kHz signal from the pre-kernels, or null: resample here */,
                  uint8_t* __restrict__ status /* [cs][dT] LC3D_ENC_ST_* bits (zeroed by the host), or null */,
                  int dT, int dt0 /* the hand-over and the status rows hold dT frames per channel-stream; this launch's frame t is their frame dt0 + t */,
                  const float* __restrict__ spec /* [cs][T][N] MDCT spectra from lc3_enc_front_kernel, or null: transform here */,
                  const float* __restrict__ frec /* [cs][T][FR_WORDS] with the SNS result of lc3_enc_snsvq_kernel */,
                  const float* __restrict__ xnext /* [cs][MEMCAP] MDCT memory after the last frame */)
{
    __shared__ WaveLds L;
    const int lane = threadIdx.x;
    const int cs = blockIdx.x;
    if (cs >= ncs) return;
    if (lane < LC3D_PLAN_HEAD_WORDS) L.pc[lane] = ((const int*)P)[lane];
    if (lane < 14) L.cc[lane] = ((const int*)&chans[cs])[lane];
    LSYNC();
    const lc3d_chan* __restrict__ C = &chans[cs];
    const int N = PI(N), channels = PI(channels), ml = N - PI(la);
    const int strm = cs / channels, ch = cs - strm * channels;

    /* ---- load cross-frame state ---- */
    float* stp = state + (size_t)cs * LC3D_STATE_WORDS(MEMCAP);
    for (int i = lane; i < MEMCAP; i += WAVE) L.xbuf[i] = stp[LC3D_ST_XPREV + i];
    for (int i = lane; i < 384; i += WAVE) L.h12[i] = stp[LC3D_ST_H12(MEMCAP) + i];
    for (int i = lane; i < 194; i += WAVE) L.h6[i] = stp[LC3D_ST_H6(MEMCAP) + i];
    if (lane < 12) L.fsc[lane] = stp[LC3D_ST_SCAL(MEMCAP) + lane];
    if (lane < 16) L.isc[lane] = ((const int*)stp)[LC3D_ST_SCAL(MEMCAP) + 16 + lane];
    LSYNC();
    if (CI(reset_attack) && lane == 0) { L.fsc[F_ATT_M0] = 0; L.fsc[F_ATT_M1] = 0; L.fsc[F_ATT_ACC] = 0; L.isc[I_ATT_POS] = 0; L.isc[I_ATT_FLAG] = 0; }
    LSYNC();
#ifdef LC3_STAGE_TIMING
    if (lane < NSTAGE) L.tacc[lane] = 0;
    LSYNC();
    long long tlast = clock64();
#endif

    /* The next frame's PCM (16 bytes per lane when the layout allows) and 12.8 kHz samples are requested one frame ahead and wait
     * in registers: a wave has nothing else to hide a global-memory round trip with at the top of a frame. */
    const bool fast16 = bitdepth == 16 && (N & 7) == 0 && N <= 8 * WAVE && ((((size_t)pcm) + (((size_t)strm * T) * channels + ch) * N * 2) & 15) == 0 && ((N * 2 * channels) & 15) == 0;
    uint4 nv = make_uint4(0, 0, 0, 0); float ny0 = 0, ny1 = 0;
    constexpr int SPK = (MAXN + WAVE - 1) / WAVE;
    unsigned bob[4] = {0, 0, 0, 0};                  /* band of this lane's bins (lane + 64 k), one byte each: the table look-up of the SNS shaping, once per launch */
#pragma unroll
    for (int k = 0; k < SPK; k++) { const int j = lane + 64 * k; bob[k >> 2] |= (unsigned)(j < N ? P->band_of_bin[j] : 255) << (8 * (k & 3)); }
    float sp[SPK]; float rq = 0; int ri = 0;
#pragma unroll
    for (int k = 0; k < SPK; k++) sp[k] = 0;
#define SPEC_PREFETCH(t_) do { const float* sr_ = spec + ((size_t)cs * T + (t_)) * N; const float* fr_ = frec + ((size_t)cs * T + (t_)) * FR_WORDS; \
        _Pragma("unroll") for (int k = 0; k < SPK; k++) sp[k] = lane + 64 * k < N ? sr_[lane + 64 * k] : 0.0f; \
        if (lane < 16) rq = fr_[FR_SCFQ + lane]; \
        if (lane < 8) ri = ((const int*)fr_)[FR_IDX + lane];   /* seven indices, then the bandwidth index */ } while (0)
    if (spec && T > 0) SPEC_PREFETCH(0);
    if (T > 0) {
        if (fast16 && lane < (N >> 3)) nv = ((const uint4*)((const int16_t*)pcm + (((size_t)strm * T) * channels + ch) * N))[lane];
        if (y12) { const float* yp = y12 + ((size_t)cs * T) * 128; ny0 = lane < PI(len12) ? yp[lane] : 0.0f; ny1 = lane + 64 < PI(len12) ? yp[lane + 64] : 0.0f; }
    }
    for (int t = 0; t < T; t++) {
#ifdef LC3_STAGE_TIMING
        lc3d_trace* tr = nullptr;
#else
        lc3d_trace* tr = trace ? &trace[(size_t)cs * T + t] : nullptr;
#endif
        /* ---- PCM in (R/enc_lc3_fl.c:30-42) ---- */
        const size_t fidx = ((size_t)strm * T + t) * channels + ch;
        /* the frame's spectrum and SNS record from the frame-parallel front were requested at the end of the previous frame: park them in
         * LDS (the frame half of xbuf is free until the quantiser needs it) */
        if (spec) {
#pragma unroll
            for (int k = 0; k < SPK; k++) if (lane + 64 * k < N) XCUR(L)[lane + 64 * k] = sp[k];
            if (lane < 16) L.sm[SM_SCFQ + lane] = rq;
            if (lane < 7) L.isc[I_SCF0 + lane] = ri;
            if (lane == 7) L.isc[I_BW] = ri;
        } else if (fast16) {
            if (lane < (N >> 3)) {
                const uint4 v = nv;
                float* d = &XCUR(L)[8 * lane];
                d[0] = (float)(int16_t)(v.x & 0xffff); d[1] = (float)(int16_t)(v.x >> 16);
                d[2] = (float)(int16_t)(v.y & 0xffff); d[3] = (float)(int16_t)(v.y >> 16);
                d[4] = (float)(int16_t)(v.z & 0xffff); d[5] = (float)(int16_t)(v.z >> 16);
                d[6] = (float)(int16_t)(v.w & 0xffff); d[7] = (float)(int16_t)(v.w >> 16);
            }
            if (t + 1 < T && lane < (N >> 3)) nv = ((const uint4*)((const int16_t*)pcm + (fidx + channels) * N))[lane];
        } else if (bitdepth == 16) {
            const int16_t* p = (const int16_t*)pcm + fidx * N;
            for (int i = lane; i < N; i += WAVE) XCUR(L)[i] = (float)p[i];
        } else {
            const int32_t* p = (const int32_t*)pcm + fidx * N;
            const float sc = bitdepth == 24 ? 256.0f : 65536.0f;
            for (int i = lane; i < N; i += WAVE) XCUR(L)[i] = (float)p[i] / sc;
        }
        LSYNC();
        TICK(0);

        if (y12) {                                       /* lc3_enc_resample_kernel + lc3_enc_hp50_kernel (lc3_enc_pre.inc) have done the work */
            const int len12 = PI(len12);
            const float y0 = ny0, y1 = ny1;
            if (t + 1 < T) { const float* yp = y12 + ((size_t)cs * T + t + 1) * 128; ny0 = lane < len12 ? yp[lane] : 0.0f; ny1 = lane + 64 < len12 ? yp[lane + 64] : 0.0f; }
            float keep[6];
#pragma unroll
            for (int k = 0; k < 6; k++) { const int i = lane + 64 * k; keep[k] = (i + len12 < 384) ? L.h12[i + len12] : 0.0f; }
            LSYNC();
#pragma unroll
            for (int k = 0; k < 6; k++) { const int i = lane + 64 * k; if (i + len12 < 384) L.h12[i] = keep[k]; }
            if (lane < len12) L.h12[384 - len12 + lane] = y0;
            if (lane + 64 < len12) L.h12[384 - len12 + 64 + lane] = y1;
            LSYNC();
        } else st_resample(P, L, lane, nullptr);
        TICK(2);
        if (tr) for (int i = lane; i < PI(len12) + 1; i += WAVE) tr->s12k8[i] = L.h12[384 - PI(len12) - 24 + i];
        st_olpa(P, L, lane);
        TICK(3);
        st_ltpf(P, C, L, lane);
        TICK(4);
        if (spec) {
            TICK(5); TICK(1); TICK(6); TICK(7); TICK(8);
        } else {
        if (CI(attack_handling)) st_attack(P, L, lane);
        TICK(5);
        mdct_pre(P, L, lane);
#ifdef LC3_BIG
        if (PI(N) == 960) { mdct_dft480_cols(L, lane); mdct_dft480_rows(L, lane); } else
#endif
        if (PI(N) == 480) { mdct_dft240_cols(L, lane); mdct_dft240_rows(L, lane); }
        else if (PI(N) == 120) mdct_dft60(P, L, lane);
        else { if (PI(N) == 320) mdct_dft160_stage1(P, L, lane); else if (PI(N) == 160) mdct_dft80_stage1(P, L, lane); mdct_dft_pfa(P, L, lane); }
        mdct_post(P, L, lane);
        TICK(1);
        if (tr) for (int i = lane; i < N; i += WAVE) tr->spec_mdct[i] = L.A[i];
        st_energy_bw(P, L, lane);
        TICK(6);
        if (tr) { if (lane == 0) { tr->T0 = L.isc[I_T0]; tr->normcorr = L.fsc[F_NC]; tr->ltpf_param[0] = L.isc[I_LTPF0]; tr->ltpf_param[1] = L.isc[I_LTPF1];
                                   tr->ltpf_param[2] = L.isc[I_LTPF2]; tr->ltpf_bits = L.isc[I_LTPF_BITS]; tr->attack = L.isc[I_ATT_FLAG]; }
                  tr->ener[lane] = lane < PI(nbands) ? L.sm[SM_ENER + lane] : 0; }
        LSYNC();
        st_sns_scf(P, L, lane);
        TICK(7);
        if (tr && lane < 16) tr->scf[lane] = L.sm[SM_SCF + lane];
        st_sns_vq(P, L, lane);
        TICK(8);
        }
        st_sns_apply(P, L, lane, spec ? XCUR(L) : L.A, bob[0], bob[1], bob[2], bob[3]);
        TICK(9);
        if (tr) { if (lane < 16) tr->scf_q[lane] = L.sm[SM_SCFQ + lane]; if (lane < 7) tr->scf_idx[lane] = L.isc[I_SCF0 + lane];
                  for (int i = lane; i < N; i += WAVE) tr->spec_shaped[i] = L.A[i]; }
        int bw = uni(L.isc[I_BW]);
        if (CI(bandwidth)) {                                  /* R/cutoff_bandwidth.c:13-26 */
            const int bin = CI(bw_cut_bin);
            if (PI(ylen) > bin) {
                if (lane < 4) { const float sc4[4] = {0.5f, 0.25f, 0.125f, 0.0625f}; L.A[bin - 1 + lane] = L.A[bin - 1 + lane] * sc4[lane]; }
                for (int i = bin + 3 + lane; i < PI(ylen); i += WAVE) L.A[i] = 0;
            }
            bw = imin(bw, CI(bw_index));
            if (lane == 0) L.isc[I_BW] = bw;
        }
        const int bw_bin = lc3t_bw_bins[PI(bw_cls) * 6 + bw];
        if (lane < 16) L.isc[I_TNS_IDX0 + lane] = 0;
        if (lane < 2) L.isc[I_TNS_ORD0 + lane] = 0;
        LSYNC();
        st_tns(P, L, lane, bw, bw_bin);
        TICK(10);
        const int tns_bits = uni(L.isc[I_TNS_BITS]);
        if (tr) { if (lane == 0) { tr->bw_idx = bw; tr->tns_nfilt = L.isc[I_TNS_NF]; tr->tns_order[0] = L.isc[I_TNS_ORD0]; tr->tns_order[1] = L.isc[I_TNS_ORD1]; tr->tns_bits = tns_bits; }
                  if (lane < 16) tr->tns_rc_idx[lane] = L.isc[I_TNS_IDX0 + lane];
                  for (int i = lane; i < N; i += WAVE) tr->spec_tns[i] = L.A[i]; }
        const int tbq = CI(target_bits_init) - (tns_bits + uni(L.isc[I_LTPF_BITS]));
        st_gain_estimate(P, C, L, lane, tbq);
        TICK(11);
        if (tr && lane == 0) { tr->target_bits_quant = tbq; tr->gain0 = L.fsc[F_GAIN]; tr->gg_idx0 = L.isc[I_GG]; tr->gg_min = L.isc[I_GGMIN]; }
        st_quantize(P, C, L, lane, -1, tbq);
        TICK(12);
        {
            int gg = uni(L.isc[I_GG]), change; float gain = unif(L.fsc[F_GAIN]);
            const int nbits0 = uni(L.isc[I_NBITS]);
            if (tr && lane == 0) tr->nbits0 = nbits0;
            gain_adjust(P, L, gg, uni(L.isc[I_GGMIN]), gain, tbq, nbits0, change);
            LSYNC();
            if (lane == 0) { L.isc[I_MEM_SPEC] = nbits0; L.isc[I_GG] = gg; L.fsc[F_GAIN] = gain; L.isc[I_CHANGE] = change; }
            LSYNC();
            if (change) st_quantize(P, C, L, lane, 0, tbq);
        }
        TICK(13);
        st_noise_factor(P, L, lane, bw_bin);
        TICK(14);
        if (tr) { if (lane == 0) { tr->gain = L.fsc[F_GAIN]; tr->gg_idx = L.isc[I_GG]; tr->gain_change = L.isc[I_CHANGE]; tr->nbits = L.isc[I_NBITS]; tr->nbits2 = L.isc[I_NBITS2];
                                   tr->lastnz = L.isc[I_LASTNZ]; tr->lsb_mode = L.isc[I_LSB]; tr->fac_ns = L.isc[I_FACNS]; }
                  for (int i = lane; i < N; i += WAVE) tr->xq[i] = i < PI(ylen) ? XQ(L)[i] : 0; }
        if (uni(L.isc[I_LSB]) == 0) st_residual(P, L, lane, tbq, uni(L.isc[I_NBITS2]));
        else { for (int i = lane; i < 160; i += WAVE) ((uint32_t*)RESB(L))[i] = 0; if (lane == 0) L.isc[I_NRES] = 0; LSYNC(); }
        TICK(15);
        if (spec && t + 1 < T) SPEC_PREFETCH(t + 1);
        if (dump) {
            /* the bitstream of a frame depends on nothing but this: scalars, residual bits, quantised lines up to lastnz.  The
             * serial writer runs one frame per lane in lc3_enc_pack_kernel. */
            int* r = dump + ((size_t)cs * dT + dt0 + t) * dstride;
            if (lane < 56) r[lane] = L.isc[lane];
            const int lastnz = uni(L.isc[I_LASTNZ]), nresw = uni(L.isc[I_LSB]) == 0 ? (uni(L.isc[I_NRES]) + 31) >> 5 : 0;
            for (int i = lane; i < nresw; i += WAVE) r[PK_RES + i] = (int)((const uint32_t*)RESB(L))[i];
            const int* xq = XQ(L);
            if (PI(hrmode)) { for (int i = lane; i < ((lastnz + 1) & ~1); i += WAVE) r[PK_XQ + i] = xq[i]; }
            else {
                /* R/quantize_spec.c:50 asserts that a quantised line fits 16 bits outside the high-resolution mode; here the frame is
                 * flagged instead (the hand-over keeps the low 16 bits) */
                bool ovf = false;
                for (int p = lane; p < ((lastnz + 1) >> 1); p += WAVE) {
                    const int q0 = xq[2 * p], q1 = xq[2 * p + 1];
                    ovf |= q0 != (int)(int16_t)q0 || q1 != (int)(int16_t)q1;
                    r[PK_XQ + p] = (q0 & 0xFFFF) | (q1 << 16);
                }
                if (status && __ballot(ovf) && lane == 0) status[(size_t)cs * dT + dt0 + t] |= LC3D_ENC_ST_QUANT_RANGE;
            }
            LSYNC();
            TICK(16);
            TICK(17);
            continue;
        }
        st_bitstream(P, C, L, lane);
        LSYNC();
        TICK(16);
        if (tr && lane == 0) { tr->n_res_bits = L.isc[I_NRES]; tr->bp_side = L.isc[I_BP_SIDE]; tr->mask_side = L.isc[I_MASK_SIDE]; }
        /* ---- bytes out ---- */
        uint8_t* o = out + ((size_t)strm * T + t) * out_stride + CI(out_off);
        const int nby = CI(nbytes);
        if (((nby | (int)(size_t)o) & 3) == 0) { for (int i = lane; i < (nby >> 2); i += WAVE) ((uint32_t*)o)[i] = ((const uint32_t*)BYTES(L))[i]; }
        else for (int i = lane; i < nby; i += WAVE) o[i] = BYTES(L)[i];
        LSYNC();
        TICK(17);
    }
#ifdef LC3_STAGE_TIMING
    if (trace && lane < NSTAGE) ((long long*)&trace[(size_t)cs * T])[lane] = L.tacc[lane];
#endif
    /* ---- store cross-frame state ---- */
    for (int i = lane; i < MEMCAP; i += WAVE) stp[LC3D_ST_XPREV + i] = spec ? xnext[(size_t)cs * MEMCAP + i] : L.xbuf[i];
    for (int i = lane; i < 384; i += WAVE) stp[LC3D_ST_H12(MEMCAP) + i] = L.h12[i];
    for (int i = lane; i < 194; i += WAVE) stp[LC3D_ST_H6(MEMCAP) + i] = L.h6[i];
    /* the HP50 state belongs to lc3_enc_hp50_kernel when it runs, the attack detector's to lc3_enc_attack_kernel on the split path */
    if (lane < 12 && !(y12 && lane < 2) && !(spec && lane >= F_ATT_M0 && lane <= F_ATT_ACC)) stp[LC3D_ST_SCAL(MEMCAP) + lane] = L.fsc[lane];
    if (lane < 16 && !(spec && (lane == I_ATT_POS || lane == I_ATT_FLAG))) ((int*)stp)[LC3D_ST_SCAL(MEMCAP) + 16 + lane] = L.isc[lane];
    (void)ml;
}

/* ------------------------------------------------------------------------------------------------ */
/* C-ABI device shim (lc3_shim.h): context, uploads, launch                                          */
/* ------------------------------------------------------------------------------------------------ */
#include "lc3_enc_front.inc"       /* lc3_enc_front_kernel (or _big): the stateless front, frame-parallel */
#ifndef LC3_BIG
#include "lc3_enc_front4.inc"      /* lc3_enc_front4_kernel: the same for N = 480, four frames per wave */
#include "lc3_enc_pitch2.inc"      /* lc3_enc_pitch2_kernel: the pitch chain, two streams per wave */
#include "lc3_enc_frontm.inc"      /* lc3_enc_frontm_kernel: the front for the short prime-factor frame lengths, several frames per wave */
#endif
#include "lc3_enc_seq.inc"         /* lc3_enc_pitch_kernel: the pitch chain of the pipelined path */
#include "lc3_enc_rate.inc"        /* lc3_enc_shape_kernel, lc3_enc_rate_kernel, lc3_enc_tail_kernel (or _big): the rate chain and its frame-parallel neighbours */
#include "lc3_dec_kernels.inc"     /* lc3_dec_{plc,imdct,synth}_kernel, or the _big imdct / synth kernels in the large-layout object */
#ifndef LC3_BIG                 /* the large-layout object holds only its kernels */
#include "lc3_dec_parse.inc"
#include "lc3_dec_imdct4.inc"      /* lc3_dec_imdct4_kernel: the decoder's IMDCT for N = 480, four frames per wave */
extern "C" __global__ void lc3_encode_kernel_big(const lc3d_plan* __restrict__ P, const lc3d_chan* __restrict__ chans, float* __restrict__ state,
                                                 const void* __restrict__ pcm, int bitdepth, int T, uint8_t* __restrict__ out, int out_stride, int ncs,
                                                 lc3d_trace* __restrict__ trace, int* __restrict__ dump, int dstride, const float* __restrict__ y12,
                                                 uint8_t* __restrict__ status, int dT, int dt0, const float* __restrict__ spec, const float* __restrict__ frec,
                                                 const float* __restrict__ xnext);
extern "C" __global__ void lc3_enc_front_kernel_big(const lc3d_plan* __restrict__ P, const lc3d_chan* __restrict__ chans, const float* __restrict__ state, const void* __restrict__ pcm,
                                                    int bitdepth, int T, int tb, int nt, int fpw, int ncs, float* __restrict__ spec, int srow, int RT, int r0, float* __restrict__ rec, float* __restrict__ xnext, const float* __restrict__ xprev, int xprev_stride, int do_scf);
extern "C" __global__ void lc3_enc_shape_kernel_big(const lc3d_plan* __restrict__ P, const lc3d_chan* __restrict__ chans, int T, int tb, int nt, int fpw, int ncs,
                                                    float* __restrict__ rows, int srow, float* __restrict__ frec);
extern "C" __global__ void lc3_enc_tailw_kernel_big(const lc3d_plan* __restrict__ P, const lc3d_chan* __restrict__ chans, int T, int nt, int fpw, int ncs, const float* __restrict__ rows, int srow,
                                                     const float* __restrict__ frec, uint8_t* __restrict__ out, int out_stride, uint8_t* __restrict__ status, int min_bytes);
extern "C" __global__ void lc3_enc_rate_kernel_big(const lc3d_plan* __restrict__ P, const lc3d_chan* __restrict__ chans, float* __restrict__ state, int T, int t0, int nt, int ncs,
                                                   const float* __restrict__ rows, int srow, float* __restrict__ frec, const float* __restrict__ xnext, int last);
#include "lc3_enc_pack.inc"
#include "lc3_enc_snsvq.inc"
#include "lc3_enc_shapel.inc"
#include "lc3_enc_pre.inc"
#define LC3D_MAX_RUNS 16
#ifndef LC3D_SETS
#define LC3D_SETS 3                     /* sets of hand-over buffers under the input-ready promise: that many calls may be in flight */
#endif
#define LC3D_AHEAD_MAX_FRAMES 256     /* lc3hip_set_input_ready: calls of up to this many frames overlap with their predecessor */
#define LC3D_RUN_FRAMES 16            /* frames per run when consecutive calls do not overlap (measured, 4096 streams x 64 frames: 8: 58.1, 16: 64.9, 32: 62.8, 64: 58.6 Mframes/s) */
#define LC3D_RUN_FRAMES_READY 64      /* under the input-ready promise (calls overlap, a call's own pipeline matters less: 8: 62.4, 16: 70.1, 32: 72.6, 64: 73.0) */
/* The diagnostic switches (LC3PLUS_* environment variables), read ONCE per context in lc3hip_create / lc3hip_dec_create: no function-local statics, so two threads
 * that drive two batches never race on them, and a context's behaviour does not change under it. */
struct lc3hip_opts {
    int fused, no_split, streams5, run_frames, runs, ahead_max, rate_stream /* -1 rule, 0, 1 */, pre_runs, pitch2, scf_wave, front4, shape_fpw, shape_on_s, shape_wave,
        pack_wpg, pack_stream /* -1 off (default), 0, 1 */, resample48, resample96, dec_imdct4, check_ready, tailw_bytes, dec_parse_pad_kb, pack_pad_kb, pack_split, pack_w5, fuse_vq, stream_order, stream_skip, rate_on, dec_plc_stream, shape_on_pitch, side_prio;
};
static int env_int(const char* name, int lo, int hi, int dflt) { const char* e = getenv(name); if (!e || !*e) return dflt; const int v = atoi(e); return v >= lo && v <= hi ? v : dflt; }
static void read_opts(lc3hip_opts* o)
{
    o->fused = env_int("LC3PLUS_ENC_FUSED", 0, 1, 0);                 /* the bitstream writer inside lc3_encode_kernel */
    o->no_split = env_int("LC3PLUS_ENC_NO_SPLIT", 0, 1, 0);           /* everything in lc3_encode_kernel */
    o->streams5 = env_int("LC3PLUS_ENC_STREAMS", 0, 8, 0) >= 5;       /* the pitch kernel and the one-frame-per-lane kernels on streams of their own (GPU_MAX_HW_QUEUES >= 6) */
    o->run_frames = env_int("LC3PLUS_ENC_RUN_FRAMES", 1, 1 << 20, 0);
    o->runs = env_int("LC3PLUS_ENC_RUNS", 1, 16, 0);
    o->ahead_max = env_int("LC3PLUS_ENC_AHEAD_MAX", 1, 1 << 20, 0);
    o->rate_stream = env_int("LC3PLUS_ENC_RATE_STREAM", 0, 1, -1);
    o->pre_runs = env_int("LC3PLUS_ENC_PRE_RUNS", 1, 64, 3);
    o->pitch2 = env_int("LC3PLUS_ENC_PITCH2", 0, 1, 1);              /* 0 = one stream per wave */
    o->scf_wave = env_int("LC3PLUS_ENC_SCF_WAVE", 0, 1, 0);          /* energies / scale factors in the front kernel */
    o->front4 = env_int("LC3PLUS_ENC_FRONT4", 0, 1, 1);              /* 0 = the one-frame-at-a-time front for every frame length */
    o->shape_fpw = env_int("LC3PLUS_ENC_SHAPE_FPW", 1, 64, 0);
    o->shape_on_s = env_int("LC3PLUS_ENC_SHAPE_ON_S", 0, 1, 0);
    o->shape_wave = env_int("LC3PLUS_ENC_SHAPE_WAVE", 0, 1, 0);      /* the wave-per-frame shape kernel */
    o->pack_wpg = env_int("LC3PLUS_ENC_PACK_WPG", 1, 4, 4);          /* waves per workgroup of the writer */
    o->pack_stream = env_int("LC3PLUS_ENC_PACK_STREAM", 0, 1, -1);   /* 1 = the writers of consecutive calls on two side streams (deployment switch, see enc_launch) */
    o->resample48 = env_int("LC3PLUS_ENC_RESAMPLE48", 0, 1, 1);      /* 0 = the two-outputs-per-lane resampler for 48 kHz / 10 ms too */
    o->resample96 = env_int("LC3PLUS_ENC_RESAMPLE96", 0, 2, 1);      /* the four-outputs-per-lane resampler for 96 kHz: 0 never, 1 standard kernel layout (2.5 ms frames), 2 every frame length */
    /* frames of this size and more: tail + writer a frame per wave (lc3_enc_tailw_kernel).  Off (0) by default - measured, Mframes/s: c96 (320-byte frames) 32.5 without,
     * 27.3 with; c5 (20 ... 400 bytes) 86.5 without, 68.6 / 73.6 / 78.4 from 120 / 200 / 320 bytes: the wave-parallel writer shortens the longest wave of the call but
     * costs several times the instructions per frame, and the call is bound by instructions, not by that latency. */
    o->tailw_bytes = env_int("LC3PLUS_ENC_TAILW_BYTES", 0, 1 << 20, 0);
    o->dec_parse_pad_kb = env_int("LC3PLUS_DEC_PARSE_PAD_KB", 0, 60, -1);  /* LDS padding per parse workgroup = fewer resident parse waves; -1: the rule in lc3hip_dec_decode */
    o->pack_pad_kb = env_int("LC3PLUS_ENC_PACK_PAD_KB", 0, 60, -1);      /* LDS padding per writer workgroup = fewer resident writer waves; -1: the rule in enc_launch */
    o->pack_split = env_int("LC3PLUS_ENC_PACK_SPLIT", 0, 1, -1);        /* the writer as two kernels (head, coder); -1: the rule in enc_launch */
    o->pack_w5 = env_int("LC3PLUS_ENC_PACK_W5", 0, 1, -1);              /* the writer under a 96-register budget; -1: the rule in enc_launch (long calls of small 10 ms frames) */
    o->fuse_vq = env_int("LC3PLUS_ENC_FUSE_VQ", 0, 1, 0);               /* the SNS quantiser at the tail of the scale-factor kernel where no stream has attack handling */
    o->stream_order = env_int("LC3PLUS_ENC_STREAM_ORDER", 0, 1, 1);     /* diagnostic: 0 = the pitch stream is created before the front stream */
    o->stream_skip = env_int("LC3PLUS_ENC_STREAM_SKIP", 0, 8, 0);
    o->rate_on = env_int("LC3PLUS_ENC_RATE_ON", 0, 1, -1);                /* a rate chain that leaves the caller's stream runs on the front stream (0) / the pitch stream (1); -1: the rule in enc_launch */
    o->dec_plc_stream = env_int("LC3PLUS_DEC_PLC_STREAM", 0, 1, 1);        /* 0 = the decoder's concealment bookkeeping on the caller's stream (round 3) */
    o->shape_on_pitch = env_int("LC3PLUS_ENC_SHAPE_ON_PITCH", 0, 1, -1);  /* the shape kernel on the pitch stream; -1: the rule in enc_launch (long calls of 2.5 ms high-resolution frames only) */
    o->side_prio = env_int("LC3PLUS_ENC_SIDE_PRIO", 0, 2, 0);             /* diagnostic: 1 = the side streams at the lowest HIP stream priority, 2 = at the highest */
    o->check_ready = env_int("LC3PLUS_CHECK_READY", 0, 1, 0);        /* debug aid for lc3plus_enc_batch_set_input_ready: refuse a call made while foreign work is pending on the caller's stream */
    o->dec_imdct4 = env_int("LC3PLUS_DEC_IMDCT4", 0, 1, 1);          /* 0 = the one-frame-at-a-time IMDCT for N = 480 too */
}
struct lc3hip_ctx {
    lc3hip_opts opt;
    int device, ncs, n_streams, channels, N, big, state_words, rs48, rs96;
    lc3d_plan* d_plan; lc3d_chan* d_chans; float* d_state;
    void* d_pcm; size_t pcm_cap; uint8_t* d_out; size_t out_cap;
    lc3d_trace* d_trace; size_t trace_cap;
    int* d_dumpv[LC3D_SETS]; size_t dump_capv[LC3D_SETS]; int hr, fused; float* d_y12[LC3D_SETS]; size_t y12_cap[LC3D_SETS];
    uint8_t* d_status; uint8_t* d_statusv[LC3D_SETS]; size_t status_capv[LC3D_SETS]; int status_frames;      /* d_status: the set of the last call */
    hipStream_t s_pk[2]; hipEvent_t ev_pk[2]; int pk_par;       /* the bitstream writers of consecutive calls beside each other (enc_launch) */
    float* d_spec[LC3D_SETS]; size_t spec_cap[LC3D_SETS]; float* d_frec[LC3D_SETS]; size_t frec_cap[LC3D_SETS]; hipEvent_t ev_done[LC3D_SETS]; float* d_xnext[LC3D_SETS + 1]; int xn_par, row_par; uint8_t* h_attack; int any_attack;
    int input_ready, ahead_ok, ahead_T, ahead_R;   /* lc3hip_set_input_ready: side kernels of a call beside the previous call's tail */   /* split path (lc3_enc_front.inc) */      /* per channel-frame status bits of the last call (LC3D_ENC_ST_*) */
    /* host-pointer pipeline (lc3hip_encode_host): two chunk slots, each with device staging and (for pageable callers) pinned staging */
    void* hp_dpcm[2]; void* hp_pin_in[2]; size_t hp_pcm_cap, hp_pin_in_cap;
    hipStream_t s_h2d; hipEvent_t ev_h2d[2], ev_k[2];
    hipStream_t s_pre, s_fr, s_pit, s_ln; hipEvent_t ev_rate; int rate_armed, mean_nbytes, min_nbytes, max_nbytes; int* h_nb; hipEvent_t ev_fork, ev_p[LC3D_MAX_RUNS], ev_f[LC3D_MAX_RUNS], ev_h[LC3D_MAX_RUNS], ev_m[LC3D_MAX_RUNS], ev_v[LC3D_MAX_RUNS];   /* side streams: pitch chain, frame-parallel front, frame-parallel tail */
    int ylen, srow, la, len12, fm_frames; const float* last_frec; int last_frec_frames;      /* the records of the last pipelined call (lc3hip_last_records) */
    hipStream_t stream, last_stream; hipEvent_t ev0, ev1; float last_ms;
    hipEvent_t ev_ours, ev_now; int ours_armed;       /* LC3PLUS_CHECK_READY: the tail of the library's own work on the caller's stream */
};

#define LC3D_FUSED_MAX_T 8
#define LC3D_FUSED_MAX_T_READY 5     /* measured under the promise (Mframes/s, pipelined / in-kernel writer): 4 frames 31.6 / 38, 6: 46.2 / 40, 8: 52.4 / 41 */
#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "lc3plus_hip: %s failed: %s (%s:%d)\n", #x, hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)
/* inside the create functions: release what has been allocated so far (the caller only sees ctx == NULL) */
#define HIPCHK_OR(x, cleanup) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "lc3plus_hip: %s failed: %s (%s:%d)\n", #x, hipGetErrorString(e_), __FILE__, __LINE__); cleanup; return 1; } } while (0)

/* test hook (tests/test_gpu_parity.py::test_device_fastmath_equals_host): lc3_fastmath.h as the kernels evaluate it, over an array.  kind 0 log2, 1 log10, 2 2^x */
extern "C" __global__ void lc3_fastmath_test_kernel(int kind, const float* __restrict__ x, float* __restrict__ y, long long n)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = kind == 0 ? m_log2f(x[i]) : kind == 1 ? m_log10f(x[i]) : m_pow2f(x[i]);
}
extern "C" int lc3hip_test_fastmath(int kind, const float* x_host, float* y_host, long long n)
{
    float *dx = nullptr, *dy = nullptr;
    HIPCHK(hipMalloc((void**)&dx, (size_t)n * 4)); HIPCHK(hipMalloc((void**)&dy, (size_t)n * 4));
    HIPCHK(hipMemcpy(dx, x_host, (size_t)n * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(lc3_fastmath_test_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, kind, dx, dy, n);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpy(y_host, dy, (size_t)n * 4, hipMemcpyDeviceToHost));
    HIPCHK(hipFree(dx)); HIPCHK(hipFree(dy));
    return 0;
}
extern "C" int lc3hip_destroy(void* ctx);
extern "C" int lc3hip_create(void** out_ctx, const lc3d_plan* plan, int n_streams, int device)
{
    int ndev = 0;
    *out_ctx = nullptr;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { fprintf(stderr, "lc3plus_hip: no HIP device available (this engine has no CPU fallback)\n"); return 1; }
    lc3hip_ctx* c = (lc3hip_ctx*)calloc(1, sizeof *c);
    if (!c) return 1;
    if (device < 0) { if (hipGetDevice(&device) != hipSuccess) device = 0; }
    c->device = device;
    HIPCHK_OR(hipSetDevice(device), free(c));
    c->n_streams = n_streams; c->channels = plan->channels; c->ncs = n_streams * plan->channels; c->N = plan->N;
    c->big = LC3D_LAYOUT_BIG(plan->N, plan->la);
    c->hr = plan->hrmode; c->ylen = plan->ylen; c->la = plan->la; c->len12 = plan->len12;
    c->fm_frames = (!c->big && plan->pfa_nst >= 2 && plan->pfa_rad[0] <= 8 && plan->pfa_rad[1] <= 8 && (plan->pfa_nst < 3 || plan->pfa_rad[2] <= 8) && plan->N <= 240) ? (plan->N > 120 ? FM_F240 : 8) : 0;      /* lc3_enc_frontm_kernel */
    c->srow = LC3D_SROW(plan->ylen);
    c->rs48 = plan->N == 480 && plan->rs_stride == 4 && plan->n12 == 128 && plan->rs_mem_in_len == 60;      /* lc3_enc_resample48_kernel */
    read_opts(&c->opt);
    if (!c->opt.resample48) c->rs48 = 0;
    /* lc3_enc_resample96_kernel_n*: 96 kHz.  By default for the standard kernel layout only (2.5 ms frames): c4 125.0 -> 137.0 Mframes/s; beside the large-layout kernels its
     * 256 registers per wave cost more than its shorter run gives (c96 36.6 -> 33.4) */
    c->rs96 = c->opt.resample96 && plan->rs_stride == 2 && plan->rs_mem_in_len == 120 && (plan->N == 960 || plan->N == 480 || plan->N == 240) && plan->n12 * 15 == plan->N * 2
              && (!c->big || c->opt.resample96 == 2);
    c->fused = c->opt.fused;
    c->state_words = LC3D_STATE_WORDS(c->big ? LC3D_MEMCAP_BIG : LC3D_MEMCAP_STD);
    HIPCHK_OR(hipMalloc((void**)&c->d_plan, sizeof(lc3d_plan)), lc3hip_destroy(c));
    HIPCHK_OR(hipMemcpy(c->d_plan, plan, sizeof(lc3d_plan), hipMemcpyHostToDevice), lc3hip_destroy(c));
    HIPCHK_OR(hipMalloc((void**)&c->d_chans, sizeof(lc3d_chan) * c->ncs), lc3hip_destroy(c));
    HIPCHK_OR(hipMalloc((void**)&c->d_state, sizeof(float) * c->state_words * (size_t)c->ncs), lc3hip_destroy(c));
    /* the library's own launch stream is created when a call first needs it (a caller that brings its stream never does): HIP maps streams
     * onto a few hardware queues, and the pipelined path wants its three side streams on queues of their own */
    HIPCHK_OR(hipEventCreate(&c->ev0), lc3hip_destroy(c)); HIPCHK_OR(hipEventCreate(&c->ev1), lc3hip_destroy(c));
    *out_ctx = c;
    return 0;
}

extern "C" int lc3hip_reset_state(void* ctx, const float* init_state_one /* LC3D_STATE_WORDS floats */)
{
    lc3hip_ctx* c = (lc3hip_ctx*)ctx;
    HIPCHK(hipSetDevice(c->device));
    const size_t sw = (size_t)c->state_words;
    float* h = (float*)malloc(sizeof(float) * sw * (size_t)c->ncs);
    if (!h) return 1;
    for (int i = 0; i < c->ncs; i++) memcpy(h + (size_t)i * sw, init_state_one, sizeof(float) * sw);
    hipError_t e = hipMemcpy(c->d_state, h, sizeof(float) * sw * (size_t)c->ncs, hipMemcpyHostToDevice);
    free(h);
    HIPCHK(e);
    c->ahead_ok = 0;                       /* the MDCT memory is in the state again, not in the hand-over of a previous call */
    return 0;
}

extern "C" int lc3hip_upload_chans(void* ctx, const lc3d_chan* chans, int first, int count)
{
    lc3hip_ctx* c = (lc3hip_ctx*)ctx;
    HIPCHK(hipSetDevice(c->device));
    /* a launch with sync = 0 may still be reading d_chans, possibly on a caller's non-blocking stream that a plain hipMemcpy does not
     * wait for: drain the stream the last launch went to first */
    if (c->last_stream) { HIPCHK(hipStreamSynchronize(c->last_stream)); c->last_stream = nullptr; }
    HIPCHK(hipMemcpy(c->d_chans + first, chans, sizeof(lc3d_chan) * count, hipMemcpyHostToDevice));
    /* streams with attack handling need lc3_enc_attack_kernel between the front and the quantiser */
    if (!c->h_attack) { c->h_attack = (uint8_t*)calloc((size_t)c->ncs, 1); if (!c->h_attack) return 1; }
    for (int i = 0; i < count; i++) c->h_attack[first + i] = chans[i].attack_handling != 0 || chans[i].reset_attack != 0;   /* a pending reset needs the kernel too */
    c->any_attack = 0;
    for (int i = 0; i < c->ncs; i++) c->any_attack |= c->h_attack[i];
    /* mean frame size: decides where the rate chain runs (enc_launch) */
    if (!c->h_nb) { c->h_nb = (int*)calloc((size_t)c->ncs, sizeof(int)); if (!c->h_nb) return 1; }
    for (int i = 0; i < count; i++) c->h_nb[first + i] = chans[i].nbytes;
    { long long sum = 0; int mn = 1 << 30, mx = 0; for (int i = 0; i < c->ncs; i++) { sum += c->h_nb[i]; if (c->h_nb[i] < mn) mn = c->h_nb[i]; if (c->h_nb[i] > mx) mx = c->h_nb[i]; }
      c->mean_nbytes = (int)(sum / (c->ncs > 0 ? c->ncs : 1)); c->min_nbytes = mn; c->max_nbytes = mx; }
    return 0;
}

/* the kernels of one call (or of one run of frames of a call) on stream s, PCM and output in device memory.  n_frames frames from
 * dpcm [stream][n_frames][channel][N]; the hand-over records and status bytes are rows of dT frames per channel-stream in which this
 * launch fills frames dt0 ... dt0 + n_frames - 1; with `pack` the bitstream writer then runs over all dT frames into dout [stream][dT][out_stride]. */
#ifdef LC3_DUP
/* diagnostic build (tools/variants.sh dup "-DLC3_DUP", tools/dup_run.sh): LC3PLUS_ENC_DUP=<letters> launches the named kernels of the pipelined
 * path twice (r resampler, h HP50, p pitch, f front, v quantiser, s rate, k pack) - what a kernel costs in the co-resident mix.  Output is
 * wrong for the kernels that carry state (h, p, s); never built into the product library. */
static int dup_of(char k) { static const char* e = nullptr; static bool rd = false; if (!rd) { e = getenv("LC3PLUS_ENC_DUP"); rd = true; } return e && strchr(e, k) ? 2 : 1; }
#define DUPL(k) for (int dup_ = 0; dup_ < dup_of(k); dup_++)
#else
#define DUPL(k)
#endif
/* the 12.8 kHz polyphase FIR of frames hb ... hb + hn - 1 of every channel-stream on stream st: four outputs per lane where the shape allows */
static void launch_resample(lc3hip_ctx* c, hipStream_t st, const void* dpcm, int bitdepth, int n_frames, int hb, int hn, int mc, float* dy12, const float* xprev, int xprev_stride)
{
    const unsigned pruns = (unsigned)((hn + PRE_FPW - 1) / PRE_FPW);
    if (c->rs48 && bitdepth == 16 && (((size_t)dpcm) & 15) == 0)
        hipLaunchKernelGGL(lc3_enc_resample48_kernel, dim3((unsigned)c->ncs * pruns), dim3(WAVE), 0, st, c->d_plan, (const int16_t*)dpcm, c->channels, mc, n_frames, hb, hn, c->ncs, dy12, xprev, xprev_stride);
    else if (c->rs96 && bitdepth == 16 && (((size_t)dpcm) & 15) == 0) {
        auto k = c->N == 960 ? lc3_enc_resample96_kernel_n960 : c->N == 480 ? lc3_enc_resample96_kernel_n480 : lc3_enc_resample96_kernel_n240;
        const int fpb = (1920 / c->N) * PRE96_ITERS;                       /* frames per workgroup: PRE96_ITERS steps of 1 920 samples */
        hipLaunchKernelGGL(k, dim3((unsigned)c->ncs * (unsigned)((hn + fpb - 1) / fpb)), dim3(WAVE), 0, st, c->d_plan, (const int16_t*)dpcm, c->channels, mc, n_frames, hb, hn, c->ncs, dy12, xprev, xprev_stride);
    }
    else
        hipLaunchKernelGGL(lc3_enc_resample_kernel, dim3((unsigned)c->ncs * pruns), dim3(WAVE), 0, st, c->d_plan, c->d_state, c->state_words, mc, dpcm, bitdepth, n_frames, hb, hn, c->ncs, dy12, xprev, xprev_stride);
}
static int enc_launch(lc3hip_ctx* c, const void* dpcm, int bitdepth, int n_frames, uint8_t* dout, int out_stride, hipStream_t s, lc3d_trace* dtr,
                      int dT, int dt0, bool pack)
{
    /* two kernels: lc3_encode_kernel (one wave per channel-stream, frames in order) leaves each frame's parameters and quantised
     * spectrum in a record; lc3_enc_pack_kernel (one channel-frame per lane, any frame size) writes the bytes.  With stage traces,
     * or with LC3PLUS_ENC_FUSED=1 (diagnostic), the first kernel writes the bytes itself. */
    /* A call of very few frames is latency bound and the one-frame-per-lane writer is the longest chain in it (~0.19 ms for a frame of
     * 80 bytes whatever the batch size): up to LC3D_FUSED_MAX_T frames per call the wave-parallel writer inside the first kernel
     * (st_bitstream, ~6 us per frame) is used instead - the single-stream lc3_enc_* API and T = 1 batches live here. */
    int* ddump = nullptr; int dstride = 0;
    const int set = c->input_ready ? c->row_par : 0;        /* the set of hand-over buffers of this call (rows, records, writer scratch, status bytes) */
    /* with the input-ready promise consecutive short calls overlap on the pipelined path, which then wins from 4 frames per call
     * (4096 streams, Mframes/s pipelined / in-kernel writer: 3 frames 31.9 / 36.3, 4: 39.8 / 38.0, 6: 48.6 / 40.2, 8: 53.9 / 41.2; without the
     * promise 8: 40.2 / 41.2) */
    const bool in_kernel_writer = dtr || c->fused || dT <= (c->input_ready ? LC3D_FUSED_MAX_T_READY : LC3D_FUSED_MAX_T);
    if (!in_kernel_writer) {
        dstride = PK_STRIDE(c->N, c->hr);
        const size_t need = (size_t)c->ncs * dT * dstride;
        /* under the input-ready promise every set is sized on the first call that needs it: no allocation inside a later (timed, overlapped) call */
        for (int i = c->input_ready ? 0 : set; i < (c->input_ready ? LC3D_SETS : set + 1); i++)
            if (c->dump_capv[i] < need) { if (c->d_dumpv[i]) HIPCHK(hipFree(c->d_dumpv[i])); c->d_dumpv[i] = nullptr; c->dump_capv[i] = 0; HIPCHK(hipMalloc((void**)&c->d_dumpv[i], need * sizeof(int))); c->dump_capv[i] = need; }
        ddump = c->d_dumpv[set];
    }
    /* ahead of it: the 12.8 kHz resampler of all frames at once and its HP50 recurrence one stream per lane (lc3_enc_pre.inc) */
    float* dy12 = nullptr;
    if (!dtr && !c->fused) {
        const size_t need = (size_t)c->ncs * n_frames * 128;
        const int yb = c->input_ready ? c->row_par : 0;      /* two buffers under the input-ready promise: the next call's resampler may run beside this call's pitch kernel */
        for (int i = c->input_ready ? 0 : yb; i < (c->input_ready ? LC3D_SETS : yb + 1); i++)
            if (c->y12_cap[i] < need) { if (c->d_y12[i]) HIPCHK(hipFree(c->d_y12[i])); c->d_y12[i] = nullptr; c->y12_cap[i] = 0; HIPCHK(hipMalloc((void**)&c->d_y12[i], need * sizeof(float))); c->y12_cap[i] = need; }
        dy12 = c->d_y12[yb];
    }
    const bool split = dy12 && ddump && !c->opt.no_split;
    if (dt0 == 0) {   /* per channel-frame status bits (LC3D_ENC_ST_*), cleared per call: by the stream that runs the kernel that sets them (the writer's, on the pipelined path) */
        const size_t need = (size_t)c->ncs * dT;
        for (int i = c->input_ready ? 0 : set; i < (c->input_ready ? LC3D_SETS : set + 1); i++)
            if (c->status_capv[i] < need) { if (c->d_statusv[i]) HIPCHK(hipFree(c->d_statusv[i])); c->d_statusv[i] = nullptr; c->status_capv[i] = 0; HIPCHK(hipMalloc((void**)&c->d_statusv[i], need)); c->status_capv[i] = need; }
        c->d_status = c->d_statusv[set];
        if (!split) HIPCHK(hipMemsetAsync(c->d_status, 0, need, s));
        c->status_frames = dT;
    }
    bool rate_on_side = false;
    float* rows_for_pack = nullptr; const float* frec_for_pack = nullptr;      /* pipelined path: the bitstream writer starts from the shaped spectra (frame-parallel tail, one frame per lane) */
    const int mc = c->big ? LC3D_MEMCAP_BIG : LC3D_MEMCAP_STD;
    if (!split) {
        c->ahead_ok = 0; c->last_frec = nullptr; c->last_frec_frames = 0;
        /* everything in lc3_encode_kernel (traced, diagnostic and very short launches), behind the 12.8 kHz pre-kernels when they apply */
        if (dy12) {
            launch_resample(c, s, dpcm, bitdepth, n_frames, 0, n_frames, mc, dy12, c->d_state + LC3D_ST_XPREV, c->state_words);
            hipLaunchKernelGGL(lc3_enc_hp50_kernel, dim3((unsigned)((c->ncs + WAVE - 1) / WAVE)), dim3(WAVE), 0, s, c->d_plan, c->d_state, c->state_words, LC3D_ST_SCAL(mc), n_frames, 0, n_frames, c->ncs, dy12);
            HIPCHK(hipGetLastError());
        }
        if (c->big) hipLaunchKernelGGL(lc3_encode_kernel_big, dim3(c->ncs), dim3(WAVE), 0, s, c->d_plan, c->d_chans, c->d_state, dpcm, bitdepth, n_frames,
                                       dout, out_stride, c->ncs, dtr, ddump, dstride, dy12, c->d_status, dT, dt0, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr);
        else hipLaunchKernelGGL(lc3_encode_kernel, dim3(c->ncs), dim3(WAVE), 0, s, c->d_plan, c->d_chans, c->d_state, dpcm, bitdepth, n_frames,
                                dout, out_stride, c->ncs, dtr, ddump, dstride, dy12, c->d_status, dT, dt0, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr);
    } else {
        /* The pipelined path.  Per run of frames: on one side stream the pitch chain (resampler per frame, HP50 one stream per lane, OLPA +
         * LTPF one stream per wave); on another the frame-parallel front (MDCT ... scale factors), the attack decision and the SNS quantiser
         * (one frame per lane); on the launch stream the shape kernel (SNS shaping, TNS, log energies: frame-parallel) and behind it the rate
         * chain (lc3_enc_rate_kernel: rate loop, bisection, first quantisation), which also waits for the pitch chain.  Everything behind the
         * chain - gain adjustment, second quantisation, noise level, residual - is frame-parallel again and runs one frame per lane at the head
         * of the bitstream writer, once per call.  The side kernels of run k+1 are resident beside the launch stream's kernels of run k. */
        /* spectrum rows and records of all dT frames of the call (a call through host pointers comes in pieces: rows dt0 ...): two sets under
         * the input-ready promise (consecutive calls overlap: the side kernels of a call write one set while the bitstream writer of the call
         * before still reads the other), one otherwise */
        const int hb_ = set;
        const size_t ns = (size_t)c->ncs * dT * c->srow, nr = (size_t)c->ncs * dT * FR_WORDS;
        for (int i = c->input_ready ? 0 : hb_; i < (c->input_ready ? LC3D_SETS : hb_ + 1); i++) {
            if (c->spec_cap[i] < ns) { if (c->d_spec[i]) HIPCHK(hipFree(c->d_spec[i])); c->d_spec[i] = nullptr; c->spec_cap[i] = 0; HIPCHK(hipMalloc((void**)&c->d_spec[i], ns * sizeof(float))); c->spec_cap[i] = ns; }
            if (c->frec_cap[i] < nr) { if (c->d_frec[i]) HIPCHK(hipFree(c->d_frec[i])); c->d_frec[i] = nullptr; c->frec_cap[i] = 0; HIPCHK(hipMalloc((void**)&c->d_frec[i], nr * sizeof(float))); c->frec_cap[i] = nr; }
        }
        for (int i = 0; i < LC3D_SETS + 1; i++) if (!c->d_xnext[i]) HIPCHK(hipMalloc((void**)&c->d_xnext[i], (size_t)c->ncs * mc * sizeof(float)));
        if (!c->s_pre) {
            {   /* Two side streams, no third.  HIP (four hardware queues by default) gave the FIRST side stream a batch creates a queue of its own and put all later ones
                 * together on another - and kernels of two streams on one queue run one after the other.  Round 3's separate rate stream therefore shared the front stream's
                 * queue (timeline of c5: 2.9 of the call's 3.0 ms on that one queue).  A rate chain that leaves the caller's stream now runs ON one of the two side streams,
                 * chosen per call (below): the same packets in the same queue, by choice instead of by creation order. */
                for (int i = 0; i < c->opt.stream_skip; i++) { hipStream_t d; HIPCHK(hipStreamCreateWithFlags(&d, hipStreamNonBlocking)); }      /* diagnostic: shifts the assignment (never destroyed) */
                int plo = 0, phi = 0; (void)hipDeviceGetStreamPriorityRange(&plo, &phi);       /* least, greatest */
                const int prio = c->opt.side_prio == 1 ? plo : c->opt.side_prio == 2 ? phi : 0;
                if (c->opt.stream_order) { HIPCHK(hipStreamCreateWithPriority(&c->s_fr, hipStreamNonBlocking, prio)); HIPCHK(hipStreamCreateWithPriority(&c->s_pre, hipStreamNonBlocking, prio)); }
                else { HIPCHK(hipStreamCreateWithPriority(&c->s_pre, hipStreamNonBlocking, prio)); HIPCHK(hipStreamCreateWithPriority(&c->s_fr, hipStreamNonBlocking, prio)); }
            }
            /* LC3PLUS_ENC_STREAMS=5: the pitch kernel and the one-frame-per-lane kernels on streams of their own (pays only where the HIP runtime has
             * hardware queues for them: GPU_MAX_HW_QUEUES >= 6) */
            { c->s_pit = c->s_pre; c->s_ln = c->s_fr;
              if (c->opt.streams5) { HIPCHK(hipStreamCreateWithFlags(&c->s_pit, hipStreamNonBlocking)); HIPCHK(hipStreamCreateWithFlags(&c->s_ln, hipStreamNonBlocking)); } }
            for (int i = 0; i < LC3D_MAX_RUNS; i++) { HIPCHK(hipEventCreateWithFlags(&c->ev_h[i], hipEventDisableTiming)); HIPCHK(hipEventCreateWithFlags(&c->ev_m[i], hipEventDisableTiming)); HIPCHK(hipEventCreateWithFlags(&c->ev_v[i], hipEventDisableTiming)); }
            for (int i = 0; i < 2; i++) { c->s_pk[i] = NULL; HIPCHK(hipEventCreateWithFlags(&c->ev_pk[i], hipEventDisableTiming)); }
            HIPCHK(hipEventCreateWithFlags(&c->ev_rate, hipEventDisableTiming));
            HIPCHK(hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
            for (int i = 0; i < LC3D_SETS; i++) HIPCHK(hipEventCreateWithFlags(&c->ev_done[i], hipEventDisableTiming));
            for (int i = 0; i < LC3D_MAX_RUNS; i++) { HIPCHK(hipEventCreateWithFlags(&c->ev_p[i], hipEventDisableTiming)); HIPCHK(hipEventCreateWithFlags(&c->ev_f[i], hipEventDisableTiming)); }
        }
        float* dspec = c->d_spec[hb_]; float* dfrec = c->d_frec[hb_];
        rows_for_pack = dspec; frec_for_pack = dfrec;
        c->last_frec = dfrec; c->last_frec_frames = dT;
        const int runf = c->opt.run_frames ? c->opt.run_frames : c->input_ready ? LC3D_RUN_FRAMES_READY : LC3D_RUN_FRAMES;
        int R = (n_frames + runf - 1) / runf;              /* runs of frames */
        if (R > LC3D_MAX_RUNS) R = LC3D_MAX_RUNS;
        if (R < 1) R = 1;
        if (c->opt.runs) R = c->opt.runs;
        const int Tr = (n_frames + R - 1) / R;
        /* Where the side kernels of this call may start.  Normally behind everything the caller queued on s before the call (the PCM may
         * come from there).  With lc3hip_set_input_ready - the PCM of a call is complete when the call is made - and a previous call of
         * the same shape on the same stream, they need not wait for that call's chain and bitstream writer: their streams carry on in
         * their own order, writing the other set of rows and records (the set they write now was last read by the writer of the call before
         * the previous one: ev_done); the MDCT memory before frame 0 is read from the previous call's hand-over (two alternating buffers),
         * not from the state that call's last rate kernel is still to update. */
        const int amax = c->opt.ahead_max ? c->opt.ahead_max : LC3D_AHEAD_MAX_FRAMES;
        const bool ahead = c->input_ready && n_frames <= amax && c->ahead_ok && c->ahead_T == n_frames && c->ahead_R == R && c->last_stream == s && dt0 == 0 && dT == n_frames && pack;
        float* xn_w = c->d_xnext[c->xn_par];                         /* written by this call's front kernel */
        /* one buffer more than calls in flight: the one written now was last read by the call LC3D_SETS back (its resampler and front) and by the
         * rate kernel of the call before that, all finished before the bitstream writer this call's side streams have waited for */
        const float* xprev = ahead ? c->d_xnext[(c->xn_par + LC3D_SETS) % (LC3D_SETS + 1)] : c->d_state + LC3D_ST_XPREV;
        const int xprev_stride = ahead ? mc : c->state_words;
        const bool five = c->s_pit != c->s_pre;
        /* The rate chain on a stream of its own, so that the rate kernel of call k+1 runs beside the bitstream writer of call k (which stays on the
         * caller's stream: it is what the caller waits for).  It pays where the caller's stream - rate kernel + writer - is the longest of the three:
         * large frames (the writer's work grows with the bytes: c96 22 -> 32 Mframes/s, c5 77 -> 83) and short calls (c3 +3 %); on 80-byte frames in
         * calls of 64 (c1) a fourth side stream costs 0 ... 11 % (it shares one of HIP's four hardware queues with another, depending on what else the
         * process created), and on c4 4 %.  LC3PLUS_ENC_RATE_STREAM=0 / 1 forces the choice (diagnostic). */
        const int rt_env = c->opt.rate_stream;
        const bool want_rt = rt_env == 1 || (rt_env < 0 && !c->big && (c->mean_nbytes >= 120 || n_frames <= 32));      /* large layout (c96, with round 4's writer): 36.6 on the caller's stream against 33.9 / 34.5 on the front / pitch stream */
        /* ... and then on which side stream: behind the pitch kernel (it waits for the shape kernel's event) or behind the shape kernel (it waits for the pitch kernel's).
         * Measured (Mframes/s, front stream / pitch stream): 48 kHz / 10 ms x 64 frames at 120 bytes 91.6 / 102.5, 160: 88.5 / 98.0, 240: 82.0 / 88.0, 400: 72.8 / 75.2, c5 88.3 / 97.1 (calls of
         * 32: 82.7 / 88.5); 80-byte frames in calls of 6: 51.3 / 60.4, 8: 62.3 / 68.1, 12: 72.4 / 88.6, 14: 80.6 / 88.4, 18: 85.3 / 90.8, 28: 93.4 / 95.5, 32: 94.6 / 100.9 - but of 16: 94.2 / 92.6 (c3 94.1 / 89.8),
         * 20: 93.8 / 91.0, 24: 96.4 / 92.9, and c96 32.8 / 30.8.  The pitch stream is the lighter one; behind the shape kernel the rate kernel blocks nothing while it waits, which wins where the
         * front stream is at its best (calls of 16 ... 24 frames in whole groups of four - lc3_enc_front4_kernel's unit) and in the large layout. */
        const bool on_pre = c->opt.rate_on >= 0 ? c->opt.rate_on == 1 : !(c->big || (c->mean_nbytes < 120 && n_frames >= 16 && n_frames <= 24 && (n_frames & 3) == 0));
        hipStream_t rts = want_rt ? (on_pre ? c->s_pit : c->s_ln) : NULL;          /* NULL: the rate kernels run on the caller's stream */
        if (!ahead) {
            HIPCHK(hipEventRecord(c->ev_fork, s)); HIPCHK(hipStreamWaitEvent(c->s_pre, c->ev_fork, 0)); HIPCHK(hipStreamWaitEvent(c->s_fr, c->ev_fork, 0));
            if (five) { HIPCHK(hipStreamWaitEvent(c->s_pit, c->ev_fork, 0)); HIPCHK(hipStreamWaitEvent(c->s_ln, c->ev_fork, 0)); }
        } else {
            HIPCHK(hipStreamWaitEvent(c->s_pre, c->ev_m[R - 1], 0));    /* the resampler reads the hand-over the previous call's last front kernel wrote */
            HIPCHK(hipStreamWaitEvent(c->s_pre, c->ev_done[hb_], 0)); HIPCHK(hipStreamWaitEvent(c->s_fr, c->ev_done[hb_], 0));
            if (five) { HIPCHK(hipStreamWaitEvent(c->s_pit, c->ev_done[hb_], 0)); HIPCHK(hipStreamWaitEvent(c->s_ln, c->ev_done[hb_], 0));
                        if (c->any_attack) HIPCHK(hipStreamWaitEvent(c->s_fr, c->ev_f[R - 1], 0)); }      /* the front reads the attack detector's filter memory the previous call's attack kernel leaves */
        }
        /* The 12.8 kHz pre-kernels run ahead in larger pieces than the runs: the HP50 kernel (one stream per lane, B / 64 waves) costs ~0.1 ms
         * per launch whatever the frame count, which per run would make its stream the slowest.  First piece = the first run (the rate
         * kernel should start early), then three runs at a time, in stream order between the pitch kernels that need them.  (More side
         * streams than these two do not help: HIP multiplexes streams onto a few hardware queues and kernels of two streams that share
         * one run back to back.) */
        hipStream_t rs = s;                                  /* where the rate kernels run */
        for (int k = 0, tb = 0, hb = 0, hk = 0; tb < n_frames; k++, tb += Tr) {
            const int nt = n_frames - tb < Tr ? n_frames - tb : Tr;
            if (tb >= hb) {
                const int prn = c->opt.pre_runs;
                const int hn0 = hk == 0 ? Tr : prn * Tr, hn = n_frames - hb < hn0 ? n_frames - hb : hn0;
                DUPL('r') launch_resample(c, c->s_pre, dpcm, bitdepth, n_frames, hb, hn, mc, dy12, xprev, xprev_stride);
                DUPL('h') hipLaunchKernelGGL(lc3_enc_hp50_kernel, dim3((unsigned)((c->ncs + WAVE - 1) / WAVE)), dim3(WAVE), 0, c->s_pre, c->d_plan, c->d_state, c->state_words, LC3D_ST_SCAL(mc), n_frames, hb, hn, c->ncs, dy12);
                HIPCHK(hipGetLastError());
                hb += hn; hk++;
                if (five) { HIPCHK(hipEventRecord(c->ev_h[k], c->s_pre)); HIPCHK(hipStreamWaitEvent(c->s_pit, c->ev_h[k], 0)); }
            }
            const int p2 = c->opt.pitch2;
            if (p2 && (c->len12 == 128 || c->len12 == 64 || c->len12 == 32)) {
                auto pk = c->len12 == 128 ? lc3_enc_pitch2_kernel : c->len12 == 64 ? lc3_enc_pitch2_kernel_l64 : lc3_enc_pitch2_kernel_l32;
                DUPL('p') hipLaunchKernelGGL(pk, dim3((unsigned)((c->ncs + 1) / 2)), dim3(WAVE), 0, c->s_pit, c->d_plan, c->d_chans, c->d_state, c->state_words, mc, dy12, n_frames, tb, nt, c->ncs, dfrec, dT, dt0);
            }
            else DUPL('p') hipLaunchKernelGGL(lc3_enc_pitch_kernel, dim3(c->ncs), dim3(WAVE), 0, c->s_pit, c->d_plan, c->d_chans, c->d_state, c->state_words, mc, dy12, n_frames, tb, nt, c->ncs, dfrec, dT, dt0);
            HIPCHK(hipGetLastError());
            HIPCHK(hipEventRecord(c->ev_p[k], c->s_pit));
            const int scf_wave = c->opt.scf_wave;
            const int fpw = nt < FRONT_FPW ? nt : FRONT_FPW;
            const unsigned fruns = (unsigned)((nt + fpw - 1) / fpw);
            const int f4 = c->opt.front4;
            if (f4 && !c->big && !scf_wave && c->N == 480 && c->la == 180 && (c->ylen & 15) == 0)
                DUPL('f') hipLaunchKernelGGL(lc3_enc_front4_kernel, dim3((unsigned)c->ncs * (unsigned)((nt + 3) / 4)), dim3(WAVE), 0, c->s_fr, c->d_plan, c->d_chans, c->d_state, dpcm, bitdepth, n_frames, tb, nt, c->ncs, dspec, c->srow, dT, dt0, dfrec, xn_w, xprev, xprev_stride);
            else if (f4 && c->fm_frames && !scf_wave)
                hipLaunchKernelGGL(lc3_enc_frontm_kernel, dim3((unsigned)c->ncs * (unsigned)((nt + c->fm_frames - 1) / c->fm_frames)), dim3(WAVE), 0, c->s_fr, c->d_plan, c->d_chans, c->d_state, dpcm, bitdepth, n_frames, tb, nt, c->fm_frames, c->ncs, dspec, c->srow, dT, dt0, dfrec, xn_w, xprev, xprev_stride);
            else if (c->big) hipLaunchKernelGGL(lc3_enc_front_kernel_big, dim3((unsigned)c->ncs * fruns), dim3(WAVE), 0, c->s_fr, c->d_plan, c->d_chans, c->d_state, dpcm, bitdepth, n_frames, tb, nt, fpw, c->ncs, dspec, c->srow, dT, dt0, dfrec, xn_w, xprev, xprev_stride, scf_wave);
            else DUPL('f') hipLaunchKernelGGL(lc3_enc_front_kernel, dim3((unsigned)c->ncs * fruns), dim3(WAVE), 0, c->s_fr, c->d_plan, c->d_chans, c->d_state, dpcm, bitdepth, n_frames, tb, nt, fpw, c->ncs, dspec, c->srow, dT, dt0, dfrec, xn_w, xprev, xprev_stride, scf_wave);
            HIPCHK(hipEventRecord(c->ev_m[k], c->s_fr));                 /* the MDCT memory hand-over and the spectrum rows of the run are written */
            if (five) HIPCHK(hipStreamWaitEvent(c->s_ln, c->ev_m[k], 0));
            const int fuse_vq = !scf_wave && !c->any_attack && c->opt.fuse_vq;
            if (!scf_wave) DUPL('e') hipLaunchKernelGGL(lc3_enc_scf_lane_kernel, dim3((unsigned)(((long long)c->ncs * nt + WAVE - 1) / WAVE)), dim3(WAVE), 0, c->s_ln, c->d_plan, dT, dt0 + tb, nt, c->ncs, dspec, c->srow, dfrec, fuse_vq);
            if (c->any_attack)
                hipLaunchKernelGGL(lc3_enc_attack_kernel, dim3((unsigned)((c->ncs + WAVE - 1) / WAVE)), dim3(WAVE), 0, c->s_ln, c->d_plan, c->d_chans, c->d_state, c->state_words, LC3D_ST_SCAL(mc), dfrec, dT, dt0, tb, nt, c->ncs);
            const long long nfr = (long long)c->ncs * nt;
            if (!fuse_vq) DUPL('v') hipLaunchKernelGGL(lc3_enc_snsvq_kernel, dim3((unsigned)((nfr + WAVE - 1) / WAVE)), dim3(WAVE), 0, c->s_ln, c->d_plan, dfrec, dT, dt0, tb, nt, c->ncs, c->any_attack);
            {   /* shaping, TNS and the stateless half of the gain estimate: frame-parallel, behind the quantiser (LC3PLUS_ENC_SHAPE_ON_S=1, diagnostic: on the
                 * launch stream in front of the rate kernel instead) */
                const int sfpw = c->opt.shape_fpw ? c->opt.shape_fpw : SHAPE_FPW, son = c->opt.shape_on_s;
                const int spw = nt < sfpw ? nt : sfpw;
                const unsigned sruns = (unsigned)((nt + spw - 1) / spw);
                /* LC3PLUS_ENC_SHAPE_ON_PITCH=1 (diagnostic): the shape kernel on the pitch stream, behind the pitch kernel, waiting for the quantiser's event.  Tried for c96, whose
                 * front stream is the longest (3.6 of a 3.6 ms call) and whose pitch stream the lightest (1.4): the next call's pitch chain then queues behind a shape kernel that
                 * waits for the front stream - 37.3 -> 31.0 Mframes/s; c1 111 -> 101, c5 97 -> 81, c3 93 -> 83; only c4 - long calls of 2.5 ms high-resolution frames, four runs per call, the front stream 8.5 of the 8.6 ms - gains (122.0 -> 126.1): on for that shape only. */
                const bool sop = !son && (c->opt.shape_on_pitch >= 0 ? c->opt.shape_on_pitch == 1 : (c->hr && c->N == 240 && n_frames >= 128));      /* c4's shape: calls of 128 / 256 frames 123.8 -> 126.0, 122.0 -> 126.1 */
                hipStream_t ss = son ? s : sop ? c->s_pit : c->s_ln;
                rs = (son || !rts) ? s : rts;
                if (son) { HIPCHK(hipEventRecord(c->ev_f[k], c->s_ln)); HIPCHK(hipStreamWaitEvent(s, c->ev_f[k], 0)); }
                if (sop) { HIPCHK(hipEventRecord(c->ev_v[k], c->s_ln)); HIPCHK(hipStreamWaitEvent(ss, c->ev_v[k], 0)); }
                const int swave = c->opt.shape_wave;
                if (!swave) DUPL('a') hipLaunchKernelGGL(lc3_enc_shape_lane_kernel, dim3((unsigned)(((long long)c->ncs * nt + WAVE - 1) / WAVE)), dim3(WAVE), 0, ss, c->d_plan, c->d_chans, dT, dt0 + tb, nt, c->ncs, dspec, c->srow, dfrec);
                else if (c->big) hipLaunchKernelGGL(lc3_enc_shape_kernel_big, dim3((unsigned)c->ncs * sruns), dim3(WAVE), 0, ss, c->d_plan, c->d_chans, dT, dt0 + tb, nt, spw, c->ncs, dspec, c->srow, dfrec);
                else DUPL('a') hipLaunchKernelGGL(lc3_enc_shape_kernel, dim3((unsigned)c->ncs * sruns), dim3(WAVE), 0, ss, c->d_plan, c->d_chans, dT, dt0 + tb, nt, spw, c->ncs, dspec, c->srow, dfrec);
                HIPCHK(hipGetLastError());
                if (!son) { HIPCHK(hipEventRecord(c->ev_f[k], ss)); HIPCHK(hipStreamWaitEvent(rs, c->ev_f[k], 0)); }
            }
            HIPCHK(hipStreamWaitEvent(rs, c->ev_p[k], 0));
            if (k == 0 && c->rate_armed) HIPCHK(hipStreamWaitEvent(rs, c->ev_rate, 0));      /* the rate chain is a chain: behind the previous call's, whichever stream that ran on */
            const int last = tb + nt >= n_frames;            /* behind the last frame of this launch the MDCT memory goes into the state */
            if (c->big) hipLaunchKernelGGL(lc3_enc_rate_kernel_big, dim3((unsigned)((c->ncs + RATE_WG - 1) / RATE_WG)), dim3(RATE_WG * WAVE), 0, rs, c->d_plan, c->d_chans, c->d_state, dT, dt0 + tb, nt, c->ncs, dspec, c->srow, dfrec, xn_w, last);
            else DUPL('s') hipLaunchKernelGGL(lc3_enc_rate_kernel, dim3((unsigned)((c->ncs + RATE_WG - 1) / RATE_WG)), dim3(RATE_WG * WAVE), 0, rs, c->d_plan, c->d_chans, c->d_state, dT, dt0 + tb, nt, c->ncs, dspec, c->srow, dfrec, xn_w, last);
            HIPCHK(hipGetLastError());
        }
        HIPCHK(hipEventRecord(c->ev_rate, rs)); c->rate_armed = 1; rate_on_side = rs != s;
        if (rs != s) HIPCHK(hipStreamWaitEvent(s, c->ev_rate, 0));      /* the writer (and whatever the caller queues next) behind the rate chain */
        c->ahead_ok = (dt0 == 0 && dT == n_frames && pack) ? 1 : 0; c->ahead_T = n_frames; c->ahead_R = R;
        c->xn_par = (c->xn_par + 1) % (LC3D_SETS + 1);
    }
    if (ddump && pack) {
        HIPCHK(hipGetLastError());
        const int wpg = c->opt.pack_wpg;                  /* waves per workgroup of the writer (they share the coder's tables in LDS) */
        const size_t per_wave = (size_t)PK_XBUF * WAVE * sizeof(unsigned);
        const long long tasks = (long long)c->ncs * dT, per_wg = (long long)wpg * WAVE;
        /* The writer codes one frame per lane: its duration is the latency of the LARGEST frame of the batch (c5: 1.7 ms for 400 bytes, c96: 3.2 ms), whatever the
         * batch size, and on the caller's stream the writers of consecutive calls run one after the other.  Where that is the longest stream (the rule that moves the
         * rate chain off the caller's stream: large frames, short calls) and calls overlap, the writers CAN alternate between two side streams - writer k + 1 beside
         * writer k, each with its own set of scratch rows and status bytes, the caller's stream waiting for their events in call order.  Measured (Mframes/s,
         * off / on): with HIP's default four hardware queues c5 87.0 / 80.0, c96 32.1 / 28.7, c3 85.1 / 73.0 - six streams share four queues and kernels of two
         * streams on one queue run back to back; with GPU_MAX_HW_QUEUES=8 c5 90.9 / 92.4, c96 28.8 / 33.9, c3 85.5 / 85.8.  So it is a deployment switch
         * (LC3PLUS_ENC_PACK_STREAM=1 together with GPU_MAX_HW_QUEUES >= 6), off by default. */
        const int pk_env = c->opt.pack_stream;
        const bool side = split && rate_on_side && c->input_ready && dt0 == 0 && dT == n_frames && pk_env == 1;
        hipStream_t ps = s;
        if (side) {
            /* behind this call's rate chain only - NOT behind the caller's stream, whose tail is the writer of the call before: under the input-ready promise the
             * output buffer of a call, like its PCM, is the caller's to have ready (include/lc3plus_batch.h) */
            if (!c->s_pk[c->pk_par]) HIPCHK(hipStreamCreateWithFlags(&c->s_pk[c->pk_par], hipStreamNonBlocking));
            ps = c->s_pk[c->pk_par];
            HIPCHK(hipStreamWaitEvent(ps, c->ev_rate, 0));
        }
        if (split) HIPCHK(hipMemsetAsync(c->d_status, 0, (size_t)c->ncs * dT, ps));
        /* large frames: tail + writer a frame per wave (lc3_enc_tailw_kernel, lc3_enc_rate.inc), launched first - its waves are the long ones */
        const int big_from = (split && c->opt.tailw_bytes && c->max_nbytes >= c->opt.tailw_bytes) ? c->opt.tailw_bytes : 0;
        if (big_from) {
            const int fpw = dT < 4 ? dT : 4;
            const unsigned wruns = (unsigned)((dT + fpw - 1) / fpw);
            if (c->big) hipLaunchKernelGGL(lc3_enc_tailw_kernel_big, dim3((unsigned)c->ncs * wruns), dim3(WAVE), 0, ps, c->d_plan, c->d_chans, dT, dT, fpw, c->ncs, rows_for_pack, c->srow, frec_for_pack, dout, out_stride, c->d_status, big_from);
            else hipLaunchKernelGGL(lc3_enc_tailw_kernel, dim3((unsigned)c->ncs * wruns), dim3(WAVE), 0, ps, c->d_plan, c->d_chans, dT, dT, fpw, c->ncs, rows_for_pack, c->srow, frec_for_pack, dout, out_stride, c->d_status, big_from);
            HIPCHK(hipGetLastError());
        }
        const bool two = split && c->opt.pack_split == 1;
        if (!big_from || c->min_nbytes < big_from) {
            const dim3 grid((unsigned)((tasks + per_wg - 1) / per_wg)), block(wpg * WAVE);
            const size_t dyn = per_wave * wpg + ((size_t)(c->opt.pack_pad_kb > 0 ? c->opt.pack_pad_kb : 0) << 10);
            if (two) {
                hipLaunchKernelGGL(lc3_enc_pack_head_kernel, grid, block, 0, ps, c->d_plan, c->d_chans, ddump, dstride, dT, 0, dT, c->ncs, dout, out_stride, c->d_status, rows_for_pack, c->srow, frec_for_pack, big_from);
                hipLaunchKernelGGL(lc3_enc_pack_code_kernel, grid, block, dyn, ps, c->d_plan, c->d_chans, ddump, dstride, dT, 0, dT, c->ncs, dout, out_stride, c->d_status, rows_for_pack, c->srow, frec_for_pack, big_from);
            } else {
                /* 96 or 128 registers (lc3_enc_pack.inc, the table at lc3_enc_pack_kernel_w5): five waves per SIMD pay for long calls of small 10 ms frames */
                const bool w5 = c->opt.pack_w5 >= 0 ? c->opt.pack_w5 == 1 : (split && !c->big && !c->hr && c->N == 480 && dT >= 48 && c->max_nbytes <= 100);
                auto pk = w5 ? lc3_enc_pack_kernel_w5 : lc3_enc_pack_kernel;
                DUPL('k') hipLaunchKernelGGL(pk, grid, block, dyn, ps, c->d_plan, c->d_chans, ddump, dstride,
                                   dT, 0, dT, c->ncs, dout, out_stride, c->d_status, rows_for_pack, c->srow, frec_for_pack, big_from);
            }
        }
        if (split && c->input_ready) { HIPCHK(hipEventRecord(c->ev_done[c->row_par], ps)); c->row_par = (c->row_par + 1) % LC3D_SETS; }      /* this call's set of rows and records is free again */
        if (side) { HIPCHK(hipEventRecord(c->ev_pk[c->pk_par], ps)); HIPCHK(hipStreamWaitEvent(s, c->ev_pk[c->pk_par], 0)); c->pk_par ^= 1; }
    }
    HIPCHK(hipGetLastError());
    c->last_stream = s;
    return 0;
}

static bool host_ptr_is_pinned(const void* p)
{
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return false; }
    return a.type == hipMemoryTypeHost;
}

/* Host pointers on both sides (SURVEY 8d "wall-clock over the encode() call including H2D of PCM and D2H of bitstreams"): the PCM of a
 * call is cut into runs of frames (all streams advance together, so every run fills the GPU like the whole call would; cutting by
 * streams would not).  Run k+1 goes up on a copy stream while run k is encoded on the launch stream: PCM of a run is a strided block of
 * the caller's [stream][frame][channel][N] array - a 2-D copy (SDMA) straight from the caller's memory when that is pinned
 * (hipHostMalloc / hipHostRegister), otherwise rows are staged through the library's own pinned slots by the calling thread, which
 * overlaps with the GPU work of the previous run.  The bitstream writer runs ONCE behind the last run over all frames of the call (one
 * frame per lane makes it latency bound: per run it would cost as much as for the whole call), then the frames come down in one
 * linear copy.  State stays on the device between runs.  The first run is short so that the kernels start early. */
static int encode_host(lc3hip_ctx* c, const void* pcm, int bitdepth, int n_frames, void* out, int out_stride, hipStream_t s)
{
    const size_t bps = bitdepth == 16 ? 2 : 4;
    const size_t fr_in = (size_t)c->channels * c->N * bps;                    /* bytes of one stream-frame of PCM */
    const size_t pcm_bytes = (size_t)c->n_streams * n_frames * fr_in, out_bytes = (size_t)c->n_streams * n_frames * out_stride;
    int K = (int)(pcm_bytes >> 25);                                           /* ~32 MB of PCM per run */
    if (K < 1) K = 1; if (K > 8) K = 8; if (K > n_frames) K = n_frames;
    if (c->fused || n_frames <= LC3D_FUSED_MAX_T) K = 1;                                                      /* diagnostic single-kernel path: the first kernel addresses the output by its own frame count */
    const int Tc = (n_frames + K - 1) / K, T0 = K > 1 ? (Tc + 1) / 2 : Tc;   /* first run: half a run */
    const bool pin_in = host_ptr_is_pinned(pcm);
    const size_t cin = (size_t)c->n_streams * Tc * fr_in;
    if (!c->s_h2d) {
        HIPCHK(hipStreamCreateWithFlags(&c->s_h2d, hipStreamNonBlocking));
        for (int i = 0; i < 2; i++) { HIPCHK(hipEventCreateWithFlags(&c->ev_h2d[i], hipEventDisableTiming)); HIPCHK(hipEventCreateWithFlags(&c->ev_k[i], hipEventDisableTiming)); }
    }
    if (c->hp_pcm_cap < cin) { for (int i = 0; i < 2; i++) { if (c->hp_dpcm[i]) HIPCHK(hipFree(c->hp_dpcm[i])); c->hp_dpcm[i] = nullptr; } c->hp_pcm_cap = 0;
                               for (int i = 0; i < 2; i++) HIPCHK(hipMalloc(&c->hp_dpcm[i], cin)); c->hp_pcm_cap = cin; }
    if (!pin_in && c->hp_pin_in_cap < cin) { for (int i = 0; i < 2; i++) { if (c->hp_pin_in[i]) HIPCHK(hipHostFree(c->hp_pin_in[i])); c->hp_pin_in[i] = nullptr; } c->hp_pin_in_cap = 0;
                                             for (int i = 0; i < 2; i++) HIPCHK(hipHostMalloc(&c->hp_pin_in[i], cin, hipHostMallocDefault)); c->hp_pin_in_cap = cin; }
    if (c->out_cap < out_bytes) { if (c->d_out) HIPCHK(hipFree(c->d_out)); c->d_out = nullptr; c->out_cap = 0; HIPCHK(hipMalloc((void**)&c->d_out, out_bytes)); c->out_cap = out_bytes; }
    const size_t in_pitch = (size_t)n_frames * fr_in;
    HIPCHK(hipEventRecord(c->ev0, s));
    HIPCHK(hipMemsetAsync(c->d_out, 0, out_bytes, s));
    for (int k = 0, t0 = 0; t0 < n_frames; k++) {
        const int i = k & 1, want = k == 0 ? T0 : Tc, tc = n_frames - t0 < want ? n_frames - t0 : want;
        const size_t w_in = (size_t)tc * fr_in;
        if (k >= 2) HIPCHK(hipEventSynchronize(c->ev_k[i]));                   /* run k - 2 has been encoded: its staging slot is free */
        const uint8_t* src = (const uint8_t*)pcm + (size_t)t0 * fr_in;
        if (pin_in) HIPCHK(hipMemcpy2DAsync(c->hp_dpcm[i], w_in, src, in_pitch, w_in, (size_t)c->n_streams, hipMemcpyHostToDevice, c->s_h2d));
        else {
            for (int st = 0; st < c->n_streams; st++) memcpy((uint8_t*)c->hp_pin_in[i] + st * w_in, src + st * in_pitch, w_in);
            HIPCHK(hipMemcpyAsync(c->hp_dpcm[i], c->hp_pin_in[i], w_in * c->n_streams, hipMemcpyHostToDevice, c->s_h2d));
        }
        HIPCHK(hipEventRecord(c->ev_h2d[i], c->s_h2d));
        HIPCHK(hipStreamWaitEvent(s, c->ev_h2d[i], 0));
        if (enc_launch(c, c->hp_dpcm[i], bitdepth, tc, c->d_out, out_stride, s, nullptr, n_frames, t0, t0 + tc >= n_frames)) return 1;
        HIPCHK(hipEventRecord(c->ev_k[i], s));
        t0 += tc;
    }
    HIPCHK(hipEventRecord(c->ev1, s));
    HIPCHK(hipMemcpyAsync(out, c->d_out, out_bytes, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    float ms = 0; if (hipEventElapsedTime(&ms, c->ev0, c->ev1) == hipSuccess) c->last_ms = ms;
    return 0;
}

extern "C" int lc3hip_encode(void* ctx, const void* pcm, int pcm_on_device, int bitdepth, int n_frames, void* out, int out_stride,
                             int out_on_device, void* hip_stream, int sync, void* trace_host)
{
    lc3hip_ctx* c = (lc3hip_ctx*)ctx;
    HIPCHK(hipSetDevice(c->device));
    if (!hip_stream && !c->stream) HIPCHK(hipStreamCreate(&c->stream));
    hipStream_t s = hip_stream ? (hipStream_t)hip_stream : c->stream;
    if (!pcm_on_device && !out_on_device && !trace_host) return encode_host(c, pcm, bitdepth, n_frames, out, out_stride, s);
    const size_t bps = bitdepth == 16 ? 2 : 4;
    const size_t pcm_bytes = (size_t)c->n_streams * n_frames * c->channels * c->N * bps;
    const size_t out_bytes = (size_t)c->n_streams * n_frames * out_stride;
    const void* dpcm = pcm; uint8_t* dout = (uint8_t*)out;
    if (!pcm_on_device) {
        if (c->pcm_cap < pcm_bytes) { if (c->d_pcm) HIPCHK(hipFree(c->d_pcm)); c->d_pcm = nullptr; c->pcm_cap = 0; HIPCHK(hipMalloc(&c->d_pcm, pcm_bytes)); c->pcm_cap = pcm_bytes; }
        HIPCHK(hipMemcpyAsync(c->d_pcm, pcm, pcm_bytes, hipMemcpyHostToDevice, s));
        dpcm = c->d_pcm;
    }
    if (!out_on_device) {
        if (c->out_cap < out_bytes) { if (c->d_out) HIPCHK(hipFree(c->d_out)); c->d_out = nullptr; c->out_cap = 0; HIPCHK(hipMalloc((void**)&c->d_out, out_bytes)); c->out_cap = out_bytes; }
        dout = c->d_out;
        HIPCHK(hipMemsetAsync(dout, 0, out_bytes, s));
    }
    lc3d_trace* dtr = nullptr;
    if (trace_host) {
        const size_t tb = sizeof(lc3d_trace) * (size_t)c->ncs * n_frames;
        if (c->trace_cap < tb) { if (c->d_trace) HIPCHK(hipFree(c->d_trace)); c->d_trace = nullptr; c->trace_cap = 0; HIPCHK(hipMalloc((void**)&c->d_trace, tb)); c->trace_cap = tb; }
        HIPCHK(hipMemsetAsync(c->d_trace, 0, tb, s));
        dtr = c->d_trace;
    }
    if (c->opt.check_ready && c->input_ready && pcm_on_device) {
        /* The promise says the PCM is complete NOW.  What can be checked: if everything this library queued on s has finished and s still has work pending, that work is
         * the caller's - possibly the producer of this PCM.  (While our own work is pending nothing can be told apart; the check is a debug aid, not a proof.) */
        if (!c->ev_ours) { HIPCHK(hipEventCreateWithFlags(&c->ev_ours, hipEventDisableTiming)); HIPCHK(hipEventCreateWithFlags(&c->ev_now, hipEventDisableTiming)); }
        if ((!c->ours_armed || c->last_stream != s || hipEventQuery(c->ev_ours) == hipSuccess) && hipStreamQuery(s) == hipErrorNotReady) {
            fprintf(stderr, "lc3plus_hip: LC3PLUS_CHECK_READY: lc3plus_enc_batch_set_input_ready(1) is in force, but work queued by the caller is still pending on the stream "
                            "of this call - the PCM (or the output buffer) may not be ready; call refused\n");
            (void)hipGetLastError();
            return 1;
        }
        (void)hipGetLastError();
    }
    HIPCHK(hipEventRecord(c->ev0, s));
    if (enc_launch(c, dpcm, bitdepth, n_frames, dout, out_stride, s, dtr, n_frames, 0, true)) return 1;
    HIPCHK(hipEventRecord(c->ev1, s));
    if (c->opt.check_ready && c->ev_ours) { HIPCHK(hipEventRecord(c->ev_ours, s)); c->ours_armed = 1; }
    if (!out_on_device) HIPCHK(hipMemcpyAsync(out, dout, out_bytes, hipMemcpyDeviceToHost, s));
    if (trace_host) HIPCHK(hipMemcpyAsync(trace_host, dtr, sizeof(lc3d_trace) * (size_t)c->ncs * n_frames, hipMemcpyDeviceToHost, s));
    if (sync || !out_on_device || trace_host) {
        HIPCHK(hipStreamSynchronize(s));
        float ms = 0; if (hipEventElapsedTime(&ms, c->ev0, c->ev1) == hipSuccess) c->last_ms = ms;
    }
    return 0;
}

/* status bits of the last call, [channel-stream][frame] (n = ncs * frames of that call), to host memory */
extern "C" int lc3hip_last_status(void* ctx, uint8_t* status_host, int n)
{
    lc3hip_ctx* c = (lc3hip_ctx*)ctx;
    HIPCHK(hipSetDevice(c->device));
    if (n > c->ncs * c->status_frames) n = c->ncs * c->status_frames;
    if (c->last_stream) HIPCHK(hipStreamSynchronize(c->last_stream));
    if (n > 0) HIPCHK(hipMemcpy(status_host, c->d_status, (size_t)n, hipMemcpyDeviceToHost));
    return n;
}

/* the per-frame records of the last call of the pipelined path (FR_* in lc3_plan.h: scale factors, SNS indices, bandwidth, LTPF and TNS parameters, gain floor,
 * the rate kernel's four words), [channel-stream][frame][FR_WORDS] to host memory: stage-level parity tests of the product path read them */
extern "C" int lc3hip_last_records(void* ctx, float* rec_host, int max_words)
{
    lc3hip_ctx* c = (lc3hip_ctx*)ctx;
    if (!c || !c->last_frec) return 0;
    HIPCHK(hipSetDevice(c->device));
    if (c->last_stream) HIPCHK(hipStreamSynchronize(c->last_stream));
    long long n = (long long)c->ncs * c->last_frec_frames * FR_WORDS;
    if (n > max_words) n = max_words;
    if (n > 0) HIPCHK(hipMemcpy(rec_host, c->last_frec, (size_t)n * sizeof(float), hipMemcpyDeviceToHost));
    return (int)n;
}

/* checkpoint / resume: the cross-frame state of every channel-stream (LC3D_STATE_WORDS words each, the layout of lc3_plan.h) as one host
 * array; everything else a batch holds is derived from its configuration.  Both wait for the last call to finish. */
extern "C" size_t lc3hip_state_bytes(void* ctx) { lc3hip_ctx* c = (lc3hip_ctx*)ctx; return c ? sizeof(float) * (size_t)c->state_words * (size_t)c->ncs : 0; }
extern "C" int lc3hip_get_state(void* ctx, void* host, size_t bytes)
{
    lc3hip_ctx* c = (lc3hip_ctx*)ctx;
    if (!c || !host || bytes != lc3hip_state_bytes(ctx)) return 1;
    HIPCHK(hipSetDevice(c->device));
    if (c->last_stream) HIPCHK(hipStreamSynchronize(c->last_stream));
    HIPCHK(hipMemcpy(host, c->d_state, bytes, hipMemcpyDeviceToHost));
    return 0;
}
extern "C" int lc3hip_set_state(void* ctx, const void* host, size_t bytes)
{
    lc3hip_ctx* c = (lc3hip_ctx*)ctx;
    if (!c || !host || bytes != lc3hip_state_bytes(ctx)) return 1;
    HIPCHK(hipSetDevice(c->device));
    if (c->last_stream) HIPCHK(hipStreamSynchronize(c->last_stream));
    HIPCHK(hipMemcpy(c->d_state, host, bytes, hipMemcpyHostToDevice));
    c->ahead_ok = 0;                       /* the MDCT memory is in the state, not in a previous call's hand-over */
    return 0;
}

extern "C" int lc3hip_set_input_ready(void* ctx, int ready)
{
    lc3hip_ctx* c = (lc3hip_ctx*)ctx;
    if (!c) return 1;
    c->input_ready = ready != 0; c->ahead_ok = 0;
    return 0;
}

extern "C" float lc3hip_last_ms(void* ctx)
{
    lc3hip_ctx* c = (lc3hip_ctx*)ctx;
    float ms = 0;
    if (hipEventSynchronize(c->ev1) == hipSuccess && hipEventElapsedTime(&ms, c->ev0, c->ev1) == hipSuccess) c->last_ms = ms;
    return c->last_ms;
}

extern "C" int lc3hip_destroy(void* ctx)
{
    lc3hip_ctx* c = (lc3hip_ctx*)ctx;
    if (!c) return 0;
    hipSetDevice(c->device);
    hipDeviceSynchronize();
    if (c->d_plan) hipFree(c->d_plan);
    if (c->d_chans) hipFree(c->d_chans);
    if (c->d_state) hipFree(c->d_state);
    if (c->d_pcm) hipFree(c->d_pcm);
    if (c->d_out) hipFree(c->d_out);
    for (int i = 0; i < LC3D_SETS; i++) { if (c->d_dumpv[i]) hipFree(c->d_dumpv[i]); if (c->d_statusv[i]) hipFree(c->d_statusv[i]); }
    for (int i = 0; i < LC3D_SETS; i++) if (c->d_y12[i]) hipFree(c->d_y12[i]);
    if (c->d_trace) hipFree(c->d_trace);
    for (int i = 0; i < LC3D_SETS; i++) { if (c->d_spec[i]) hipFree(c->d_spec[i]); if (c->d_frec[i]) hipFree(c->d_frec[i]); }
    for (int i = 0; i < LC3D_SETS + 1; i++) if (c->d_xnext[i]) hipFree(c->d_xnext[i]);
    free(c->h_attack); free(c->h_nb);
    for (int i = 0; i < 2; i++) {
        if (c->hp_dpcm[i]) hipFree(c->hp_dpcm[i]);
        if (c->hp_pin_in[i]) hipHostFree(c->hp_pin_in[i]);
        if (c->ev_h2d[i]) hipEventDestroy(c->ev_h2d[i]);
        if (c->ev_k[i]) hipEventDestroy(c->ev_k[i]);
    }
    if (c->s_h2d) hipStreamDestroy(c->s_h2d);
    if (c->s_pre) { if (c->s_pit != c->s_pre) { hipStreamDestroy(c->s_pit); hipStreamDestroy(c->s_ln); } hipStreamDestroy(c->s_pre); hipStreamDestroy(c->s_fr); for (int i = 0; i < 2; i++) { if (c->s_pk[i]) hipStreamDestroy(c->s_pk[i]); hipEventDestroy(c->ev_pk[i]); } hipEventDestroy(c->ev_rate);
                    for (int i = 0; i < LC3D_MAX_RUNS; i++) { hipEventDestroy(c->ev_h[i]); hipEventDestroy(c->ev_m[i]); hipEventDestroy(c->ev_v[i]); } hipEventDestroy(c->ev_fork); for (int i = 0; i < LC3D_SETS; i++) hipEventDestroy(c->ev_done[i]);
                    for (int i = 0; i < LC3D_MAX_RUNS; i++) { hipEventDestroy(c->ev_p[i]); hipEventDestroy(c->ev_f[i]); } }
    if (c->ev0) hipEventDestroy(c->ev0);
    if (c->ev1) hipEventDestroy(c->ev1);
    if (c->ev_ours) { hipEventDestroy(c->ev_ours); hipEventDestroy(c->ev_now); }
    if (c->stream) { hipStreamSynchronize(c->stream); hipStreamDestroy(c->stream); }
    free(c);
    return 0;
}
/* ---- decoder shim ---- */
extern "C" __global__ void lc3_dec_imdct_kernel_big(const lc3d_plan* __restrict__ P, const float* __restrict__ state, const int* __restrict__ rec, const float* __restrict__ ws,
                                                    int T, int ncs, float* __restrict__ ov, lc3d_dec_trace* __restrict__ trace);
extern "C" __global__ void lc3_dec_synth_kernel_big(const lc3d_plan* __restrict__ P, const lc3d_dchan* __restrict__ chans, float* __restrict__ state, const int* __restrict__ rec,
                                                    const float* __restrict__ ws, const float* __restrict__ ov, int T, void* __restrict__ pcm, int bps, int ncs,
                                                    uint8_t* __restrict__ status, lc3d_dec_trace* __restrict__ trace);
#ifndef DEC_SETS
#define DEC_SETS 3                      /* sets of hand-over buffers (records, spectrum rows) under the decoder's input-ready promise: the parser of call k + 2 may write while call k is synthesised */
#endif
struct lc3hip_dctx {
    lc3hip_opts opt;
    int device, ncs, n_streams, channels, N, big;
    lc3d_plan* d_plan; lc3d_dchan* d_chans; float* d_state;
    uint8_t* d_in; size_t in_cap; void* d_pcm; size_t pcm_cap; uint8_t* d_bfi; size_t bfi_cap;
    lc3d_dec_trace* d_trace; size_t trace_cap; uint8_t* d_status; size_t status_cap;
    int* d_rec; float* d_ws; float* d_ov; size_t hand_cap; int max_nbytes; int* h_nbytes;
    hipStream_t stream, last_stream; hipEvent_t ev0, ev1; float last_ms;
    /* lc3hip_dec_set_input_ready: the parse kernel of a call runs on a stream of its own beside the transform and synthesis of the call before; a
     * second set of hand-over buffers (records, spectrum rows), alternating */
    int input_ready, set; int* d_recx[DEC_SETS - 1]; float* d_wsx[DEC_SETS - 1]; size_t handx_cap; hipStream_t s_par, s_plc; hipEvent_t ev_par[DEC_SETS], ev_free[DEC_SETS], ev_plc; int free_armed[DEC_SETS];
};
extern "C" int lc3hip_dec_destroy(void* ctx);
extern "C" int lc3hip_dec_create(void** out_ctx, const lc3d_plan* plan, int n_streams, int device)
{
    int ndev = 0;
    *out_ctx = nullptr;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { fprintf(stderr, "lc3plus_hip: no HIP device available (this engine has no CPU fallback)\n"); return 1; }
    lc3hip_dctx* c = (lc3hip_dctx*)calloc(1, sizeof *c);
    if (!c) return 1;
    if (device < 0) { if (hipGetDevice(&device) != hipSuccess) device = 0; }
    c->device = device;
    HIPCHK_OR(hipSetDevice(device), free(c));
    c->n_streams = n_streams; c->channels = plan->channels; c->ncs = n_streams * plan->channels; c->N = plan->N;
    c->big = LC3D_LAYOUT_BIG(plan->N, plan->la);
    read_opts(&c->opt);
    HIPCHK_OR(hipMalloc((void**)&c->d_plan, sizeof(lc3d_plan)), lc3hip_dec_destroy(c));
    HIPCHK_OR(hipMemcpy(c->d_plan, plan, sizeof(lc3d_plan), hipMemcpyHostToDevice), lc3hip_dec_destroy(c));
    HIPCHK_OR(hipMalloc((void**)&c->d_chans, sizeof(lc3d_dchan) * c->ncs), lc3hip_dec_destroy(c));
    HIPCHK_OR(hipMalloc((void**)&c->d_state, sizeof(float) * DST_WORDS * (size_t)c->ncs), lc3hip_dec_destroy(c));
    {   /* initial state: zeros; ltpf_mem_beta_idx = -1, cum_alpha = 1, PLC seed 24607 (R/setup_dec_lc3.c:170-183) */
        float* h = (float*)calloc((size_t)DST_WORDS * c->ncs, sizeof(float));
        if (!h) { lc3hip_dec_destroy(c); return 1; }
        for (int i = 0; i < c->ncs; i++) {
            int* sc = (int*)(h + (size_t)i * DST_WORDS + DST_SCAL);
            sc[DS_BETA_IDX] = -1; ((float*)sc)[DS_CUM_ALPHA] = 1.0f; sc[DS_PLC_SEED] = 24607;
        }
        hipError_t e = hipMemcpy(c->d_state, h, sizeof(float) * DST_WORDS * (size_t)c->ncs, hipMemcpyHostToDevice);
        free(h);
        HIPCHK_OR(e, lc3hip_dec_destroy(c));
    }
    HIPCHK_OR(hipStreamCreate(&c->stream), lc3hip_dec_destroy(c));
    HIPCHK_OR(hipEventCreate(&c->ev0), lc3hip_dec_destroy(c)); HIPCHK_OR(hipEventCreate(&c->ev1), lc3hip_dec_destroy(c));
    *out_ctx = c;
    return 0;
}
extern "C" int lc3hip_dec_upload_chans(void* ctx, const lc3d_dchan* chans, int first, int count)
{
    lc3hip_dctx* c = (lc3hip_dctx*)ctx;
    HIPCHK(hipSetDevice(c->device));
    if (c->last_stream) { HIPCHK(hipStreamSynchronize(c->last_stream)); c->last_stream = nullptr; }
    HIPCHK(hipMemcpy(c->d_chans + first, chans, sizeof(lc3d_dchan) * count, hipMemcpyHostToDevice));
    /* the largest frame of the batch selects the parse kernel's staging: keep it exact when sizes shrink again */
    if (!c->h_nbytes) { c->h_nbytes = (int*)calloc((size_t)c->ncs, sizeof(int)); if (!c->h_nbytes) return 1; }
    for (int i = 0; i < count; i++) c->h_nbytes[first + i] = chans[i].nbytes;
    c->max_nbytes = 0;
    for (int i = 0; i < c->ncs; i++) if (c->h_nbytes[i] > c->max_nbytes) c->max_nbytes = c->h_nbytes[i];
    return 0;
}
extern "C" int lc3hip_dec_decode(void* ctx, const void* frames, int frames_on_device, int in_stride, const uint8_t* bfi_host, int n_frames,
                                 void* pcm, int pcm_on_device, int bps, uint8_t* status_host, void* hip_stream, int sync, void* trace_host)
{
    lc3hip_dctx* c = (lc3hip_dctx*)ctx;
    HIPCHK(hipSetDevice(c->device));
    hipStream_t s = hip_stream ? (hipStream_t)hip_stream : c->stream;
    const size_t in_bytes = (size_t)c->n_streams * n_frames * in_stride;
    const size_t pcm_bytes = (size_t)c->ncs * n_frames * c->N * (bps == 16 ? 2 : 4);
    const uint8_t* din = (const uint8_t*)frames; void* dpcm = pcm; const uint8_t* dbfi = nullptr; lc3d_dec_trace* dtr = nullptr;
    if (!frames_on_device) {
        if (c->in_cap < in_bytes) { if (c->d_in) HIPCHK(hipFree(c->d_in)); HIPCHK(hipMalloc((void**)&c->d_in, in_bytes)); c->in_cap = in_bytes; }
        HIPCHK(hipMemcpyAsync(c->d_in, frames, in_bytes, hipMemcpyHostToDevice, s));
        din = c->d_in;
    }
    if (!pcm_on_device) {
        if (c->pcm_cap < pcm_bytes) { if (c->d_pcm) HIPCHK(hipFree(c->d_pcm)); HIPCHK(hipMalloc((void**)&c->d_pcm, pcm_bytes)); c->pcm_cap = pcm_bytes; }
        dpcm = c->d_pcm;
    }
    if (bfi_host) {
        const size_t fb = (size_t)c->n_streams * n_frames;
        if (c->bfi_cap < fb) { if (c->d_bfi) HIPCHK(hipFree(c->d_bfi)); HIPCHK(hipMalloc((void**)&c->d_bfi, fb)); c->bfi_cap = fb; }
        HIPCHK(hipMemcpyAsync(c->d_bfi, bfi_host, fb, hipMemcpyHostToDevice, s));
        dbfi = c->d_bfi;
    }
    if (trace_host) {
        const size_t tb = sizeof(lc3d_dec_trace) * (size_t)c->ncs * n_frames;
        if (c->trace_cap < tb) { if (c->d_trace) HIPCHK(hipFree(c->d_trace)); HIPCHK(hipMalloc((void**)&c->d_trace, tb)); c->trace_cap = tb; }
        HIPCHK(hipMemsetAsync(c->d_trace, 0, tb, s));
        dtr = c->d_trace;
    }
    uint8_t* dst = nullptr;
    if (status_host) {
        const size_t fb = (size_t)c->n_streams * n_frames;
        if (c->status_cap < fb) { if (c->d_status) HIPCHK(hipFree(c->d_status)); HIPCHK(hipMalloc((void**)&c->d_status, fb)); c->status_cap = fb; }
        dst = c->d_status;
    }
    {   /* hand-over buffers between the two kernels: records and spectrum rows of every channel-frame of this call */
        const size_t cf = (size_t)c->ncs * n_frames;
        if (c->hand_cap < cf) {
            if (c->d_rec) HIPCHK(hipFree(c->d_rec));
            if (c->d_ws) HIPCHK(hipFree(c->d_ws));
            if (c->d_ov) HIPCHK(hipFree(c->d_ov));
            c->d_rec = nullptr; c->d_ws = nullptr; c->d_ov = nullptr; c->hand_cap = 0;
            HIPCHK(hipMalloc((void**)&c->d_rec, cf * PR_WORDS * sizeof(int)));
            HIPCHK(hipMalloc((void**)&c->d_ws, cf * WS_ROW(c->N) * sizeof(float)));
            HIPCHK(hipMalloc((void**)&c->d_ov, cf * (c->big ? OV_ROW_BIG : OV_ROW_STD) * sizeof(float)));
            c->hand_cap = cf;
        }
    }
    /* Under the input-ready promise (the frames of a call are complete on the device when the call is made) the parse kernel - stateless: a frame's
     * record and spectrum row depend on that frame's bytes only - does not wait for what is queued on s: it runs on its own stream into the other set of
     * hand-over buffers while the concealment bookkeeping, transform and synthesis of the call before (the stateful part, in order on s) read theirs. */
    const bool ahead = c->input_ready && frames_on_device && pcm_on_device && !bfi_host && !trace_host && !status_host;
    int* rec_w = c->d_rec; float* ws_w = c->d_ws;
    if (ahead) {
        const size_t cf = (size_t)c->ncs * n_frames;
        if (!c->s_par) {
            HIPCHK(hipStreamCreateWithFlags(&c->s_par, hipStreamNonBlocking));
            for (int i = 0; i < DEC_SETS; i++) { HIPCHK(hipEventCreateWithFlags(&c->ev_par[i], hipEventDisableTiming)); HIPCHK(hipEventCreateWithFlags(&c->ev_free[i], hipEventDisableTiming)); }
            HIPCHK(hipStreamCreateWithFlags(&c->s_plc, hipStreamNonBlocking)); HIPCHK(hipEventCreateWithFlags(&c->ev_plc, hipEventDisableTiming));
        }
        if (c->handx_cap < cf) {
            HIPCHK(hipDeviceSynchronize());
            for (int i = 0; i < DEC_SETS - 1; i++) {
                if (c->d_recx[i]) HIPCHK(hipFree(c->d_recx[i]));
                if (c->d_wsx[i]) HIPCHK(hipFree(c->d_wsx[i]));
                c->d_recx[i] = nullptr; c->d_wsx[i] = nullptr;
            }
            c->handx_cap = 0;
            for (int i = 0; i < DEC_SETS - 1; i++) {
                HIPCHK(hipMalloc((void**)&c->d_recx[i], cf * PR_WORDS * sizeof(int)));
                HIPCHK(hipMalloc((void**)&c->d_wsx[i], cf * WS_ROW(c->N) * sizeof(float)));
            }
            c->handx_cap = cf;
        }
        if (c->set) { rec_w = c->d_recx[c->set - 1]; ws_w = c->d_wsx[c->set - 1]; }
    }
    hipStream_t sp = ahead ? c->s_par : s;
    /* frames of up to 128 bytes are staged in LDS; larger ones would cut the waves per workgroup and are read from global memory */
    const int nw_max = c->max_nbytes > 128 ? 0 : c->max_nbytes > 0 ? (c->max_nbytes + 3) / 4 : 1;
    const int nlw = (WS_ROW(c->N) / 2 + 31) / 32;                          /* >= (ylen / 2 + 31) / 32 of the plan */
    const size_t per_wave = (size_t)(nw_max + nlw) * WAVE * sizeof(unsigned);
    int wpg = (int)((64 * 1024 - sizeof(ParseLds)) / per_wave);            /* waves per workgroup: they share the model tables */
    if (wpg > 4) wpg = 4;
    if (wpg < 1) { fprintf(stderr, "lc3plus_hip: frame of %d bytes exceeds the parse kernel's LDS staging\n", c->max_nbytes); return 1; }
    const long long tasks = (long long)c->n_streams * n_frames, per_wg = (long long)wpg * WAVE;
    HIPCHK(hipEventRecord(c->ev0, s));
    /* parse: one stream-frame per lane; concealment bookkeeping: one channel-stream per lane; IMDCT: one channel-frame per wave;
     * synthesis: one channel-stream per wave (lc3_dec_kernels.inc) */
    if (ahead && c->free_armed[c->set]) HIPCHK(hipStreamWaitEvent(sp, c->ev_free[c->set], 0));      /* this set was last read by the synthesis of the call DEC_SETS back */
    /* How many parse waves a CU holds.  The kernel for frames of more than 128 bytes reads its frames from global memory and needs little LDS, so its 4 096
     * waves of 128 registers fill every SIMD, and the 64-wave concealment kernel and the transform of the call before wait for parse waves to retire; 24 KB of
     * padding per workgroup leave room beside them: d5 81.3 -> 88.7 Mframes/s (20 KB: 87.1, 28 KB: 77.1).  The kernel that stages its frames in LDS (d1) loses
     * with any padding (129 -> 117 at 16 KB): none there. */
    /* (the rule in bytes: the workgroup's LDS - tables, its waves' slices, padding - is a quarter of the CU's 160 KB, so that exactly four of them fit; that was 24 KB of padding
     * with the tables of the time) */
    const size_t quarter = (160u << 10) / 4, used = sizeof(ParseLds) + per_wave * wpg;
    const size_t pad = c->opt.dec_parse_pad_kb >= 0 ? (size_t)c->opt.dec_parse_pad_kb << 10 : (nw_max || used >= quarter ? 0 : quarter - used);
    if (nw_max) hipLaunchKernelGGL(lc3_dec_parse_kernel, dim3((unsigned)((tasks + per_wg - 1) / per_wg)), dim3(wpg * WAVE), per_wave * wpg + pad, sp, c->d_plan, c->d_chans, din, in_stride,
                                   dbfi, n_frames, c->n_streams, nw_max, rec_w, ws_w, WS_ROW(c->N));
    else hipLaunchKernelGGL(lc3_dec_parse_kernel_g, dim3((unsigned)((tasks + per_wg - 1) / per_wg)), dim3(wpg * WAVE), per_wave * wpg + pad, sp, c->d_plan, c->d_chans, din, in_stride,
                            dbfi, n_frames, c->n_streams, nw_max, rec_w, ws_w, WS_ROW(c->N));
    HIPCHK(hipGetLastError());
    /* The concealment bookkeeping needs its call's parser and the bookkeeping of the call before - NOT the transform or the synthesis of the call before.  On the caller's stream it
     * became runnable at the moment the NEXT call's parser did (both behind the previous synthesis; the parser waits for its set of hand-over buffers), lost the race for the SIMDs to
     * 4 096 parse waves of 128 registers, and took 0.9 ms for 0.05 ms of work - on the stream that bounds the call (timeline in profiles/experiments/r04_what_bounds.md, section 8).
     * On a stream of its own it runs the moment its parser ends, while the chip has room. */
    hipStream_t spl = s;
    if (ahead && c->opt.dec_plc_stream) {
        spl = c->s_plc;
        HIPCHK(hipEventRecord(c->ev_par[c->set], sp)); HIPCHK(hipStreamWaitEvent(spl, c->ev_par[c->set], 0));
    } else if (ahead) { HIPCHK(hipEventRecord(c->ev_par[c->set], sp)); HIPCHK(hipStreamWaitEvent(s, c->ev_par[c->set], 0)); }
    else if (c->s_plc) HIPCHK(hipStreamWaitEvent(s, c->ev_plc, 0));      /* an ordered call behind ahead calls: the bookkeeping is a chain (the event of the last one, if any: waiting on a fresh event is a no-op) */
    hipLaunchKernelGGL(lc3_dec_plc_kernel, dim3((unsigned)((c->ncs + WAVE - 1) / WAVE)), dim3(WAVE), 0, spl, c->d_plan, c->d_state, rec_w, n_frames, c->ncs);
    HIPCHK(hipGetLastError());
    if (spl != s) { HIPCHK(hipEventRecord(c->ev_plc, spl)); HIPCHK(hipStreamWaitEvent(s, c->ev_plc, 0)); }
    const unsigned ncf = (unsigned)((size_t)c->ncs * ((n_frames + IMDCT_FPW - 1) / IMDCT_FPW));     /* runs of IMDCT_FPW frames */
    if (c->big) {
        hipLaunchKernelGGL(lc3_dec_imdct_kernel_big, dim3(ncf), dim3(WAVE), 0, s, c->d_plan, c->d_state, rec_w, ws_w, n_frames, c->ncs, c->d_ov, dtr);
        hipLaunchKernelGGL(lc3_dec_synth_kernel_big, dim3(c->ncs), dim3(WAVE), 0, s, c->d_plan, c->d_chans, c->d_state, rec_w, ws_w, c->d_ov, n_frames, dpcm, bps, c->ncs, dst, dtr);
    } else {
        const int i4 = c->opt.dec_imdct4;
        if (i4 && !dtr && c->N == 480)
            hipLaunchKernelGGL(lc3_dec_imdct4_kernel, dim3((unsigned)((size_t)c->ncs * ((n_frames + 3) / 4))), dim3(WAVE), 0, s, c->d_plan, c->d_state, rec_w, ws_w, n_frames, c->ncs, c->d_ov);
        else
        hipLaunchKernelGGL(lc3_dec_imdct_kernel, dim3(ncf), dim3(WAVE), 0, s, c->d_plan, c->d_state, rec_w, ws_w, n_frames, c->ncs, c->d_ov, dtr);
        hipLaunchKernelGGL(lc3_dec_synth_kernel, dim3(c->ncs), dim3(WAVE), 0, s, c->d_plan, c->d_chans, c->d_state, rec_w, ws_w, c->d_ov, n_frames, dpcm, bps, c->ncs, dst, dtr);
    }
    HIPCHK(hipGetLastError());
    if (ahead) { HIPCHK(hipEventRecord(c->ev_free[c->set], s)); c->free_armed[c->set] = 1; c->set = (c->set + 1) % DEC_SETS; }
    else if (c->s_par) { HIPCHK(hipEventRecord(c->ev_free[0], s)); c->free_armed[0] = 1; }      /* an ordered call reads the first set: a later parse-ahead into it waits for this one */
    c->last_stream = s;
    HIPCHK(hipEventRecord(c->ev1, s));
    if (!pcm_on_device) HIPCHK(hipMemcpyAsync(pcm, dpcm, pcm_bytes, hipMemcpyDeviceToHost, s));
    if (trace_host) HIPCHK(hipMemcpyAsync(trace_host, dtr, sizeof(lc3d_dec_trace) * (size_t)c->ncs * n_frames, hipMemcpyDeviceToHost, s));
    if (status_host) HIPCHK(hipMemcpyAsync(status_host, dst, (size_t)c->n_streams * n_frames, hipMemcpyDeviceToHost, s));
    if (sync || !pcm_on_device || !frames_on_device || trace_host || bfi_host || status_host) {
        HIPCHK(hipStreamSynchronize(s));
        float ms = 0; if (hipEventElapsedTime(&ms, c->ev0, c->ev1) == hipSuccess) c->last_ms = ms;
    }
    return 0;
}
extern "C" int lc3hip_dec_set_input_ready(void* ctx, int ready)
{
    lc3hip_dctx* c = (lc3hip_dctx*)ctx;
    if (!c) return 1;
    /* 0 -> 1: an ordered call made before the promise may still be reading the first set of hand-over buffers, and it recorded no event a parse-ahead
     * could wait for (the side stream and its events exist from the first ahead call on): drain it once, as the encoder side does by clearing ahead_ok */
    if (ready && !c->input_ready && c->last_stream) { HIPCHK(hipSetDevice(c->device)); HIPCHK(hipStreamSynchronize(c->last_stream)); }
    c->input_ready = ready != 0;
    return 0;
}
extern "C" size_t lc3hip_dec_state_bytes(void* ctx) { lc3hip_dctx* c = (lc3hip_dctx*)ctx; return c ? sizeof(float) * (size_t)DST_WORDS * (size_t)c->ncs : 0; }
extern "C" int lc3hip_dec_get_state(void* ctx, void* host, size_t bytes)
{
    lc3hip_dctx* c = (lc3hip_dctx*)ctx;
    if (!c || !host || bytes != lc3hip_dec_state_bytes(ctx)) return 1;
    HIPCHK(hipSetDevice(c->device));
    if (c->last_stream) HIPCHK(hipStreamSynchronize(c->last_stream));
    HIPCHK(hipMemcpy(host, c->d_state, bytes, hipMemcpyDeviceToHost));
    return 0;
}
extern "C" int lc3hip_dec_set_state(void* ctx, const void* host, size_t bytes)
{
    lc3hip_dctx* c = (lc3hip_dctx*)ctx;
    if (!c || !host || bytes != lc3hip_dec_state_bytes(ctx)) return 1;
    HIPCHK(hipSetDevice(c->device));
    if (c->last_stream) HIPCHK(hipStreamSynchronize(c->last_stream));
    HIPCHK(hipMemcpy(c->d_state, host, bytes, hipMemcpyHostToDevice));
    return 0;
}
extern "C" float lc3hip_dec_last_ms(void* ctx) { return ctx ? ((lc3hip_dctx*)ctx)->last_ms : 0.0f; }
extern "C" int lc3hip_dec_destroy(void* ctx)
{
    lc3hip_dctx* c = (lc3hip_dctx*)ctx;
    if (!c) return 0;
    hipSetDevice(c->device);
    hipDeviceSynchronize();
    void* bufs[] = {c->d_plan, c->d_chans, c->d_state, c->d_in, c->d_pcm, c->d_bfi, c->d_trace, c->d_status, c->d_rec, c->d_ws, c->d_ov};
    for (int i = 0; i < DEC_SETS - 1; i++) { if (c->d_recx[i]) hipFree(c->d_recx[i]); if (c->d_wsx[i]) hipFree(c->d_wsx[i]); }
    if (c->s_par) { hipStreamDestroy(c->s_par); for (int i = 0; i < DEC_SETS; i++) { hipEventDestroy(c->ev_par[i]); hipEventDestroy(c->ev_free[i]); } hipStreamDestroy(c->s_plc); hipEventDestroy(c->ev_plc); }
    for (void* p : bufs) if (p) hipFree(p);
    if (c->stream) hipStreamDestroy(c->stream);
    if (c->ev0) hipEventDestroy(c->ev0);
    if (c->ev1) hipEventDestroy(c->ev1);
    free(c->h_nbytes);
    free(c);
    return 0;
}
#endif /* !LC3_BIG */
