/* lc3_kernels.hip -- gfx950 (MI355X / CDNA4) LC3plus encode kernels + the C-ABI device shim.
 *
 * One 64-lane wavefront encodes one channel-stream and walks its frames in time order; the spectrum,
 * the 12.8 kHz / 6.4 kHz pitch-analysis histories, the quantised spectrum, the entropy-coder symbol
 * list and the output frame all live in that wave's LDS slice.  PCM is read from HBM with coalesced
 * loads, bytes are written back coalesced; cross-frame state is read once per launch and written once.
 * No MFMA (nothing here is a dense contraction), no collectives.
 *
 * Numerics contract: every floating-point expression keeps the ETSI reference's evaluation order and
 * C promotions (R = LC3plus_ETSI_src_v17171_20200723/src/floating_point, cited per stage), compiled with
 * -ffp-contract=off, so that decisions (argmax, thresholds, quantisation) match the reference bit for bit.
 * Independent serial sums (autocorrelation lags, FIR taps, band energies ...) are mapped one sum per lane,
 * which keeps the reference's summation order AND fills the wave.  Run-time libm calls of the reference
 * (log2f, log10f, powf) are evaluated as (float)f((double)x) with the device's double libm.
 */
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#define LC3T_QUAL static __device__ const
#include "lc3_tables.h"
#include "lc3_plan.h"
#include "lc3_shim.h"

#define MAXN LC3D_MAX_N
#define WAVE 64
#define LSYNC() __syncthreads()

/* ------------------------------------------------------------------------------------------------ */
/* LDS slice of one wave                                                                             */
/* ------------------------------------------------------------------------------------------------ */
struct __attribute__((aligned(16))) WaveLds {
    float xbuf[2 * MAXN];   /* previous frame right-aligned in [0,MAXN), current frame in [MAXN, MAXN+N) */
    float za[MAXN];         /* DFT in  / scratch */
    float zb[MAXN];         /* DFT out / scratch */
    float spec[MAXN];       /* MDCT spectrum, shaped / TNS-filtered in place */
    float h12[384];         /* HP-filtered 12.8 kHz stream, newest sample at [383] */
    float h6[196];          /* 6.4 kHz stream, newest at [193] */
    float sm[704];          /* small vectors, see SM_* */
    int   xq[MAXN];
    uint32_t cd[MAXN / 2];  /* per 2-tuple: ctx | (maxlev+1)<<10 | sym<<16 */
    uint32_t cf[MAXN / 2];  /* per 2-tuple: cumfreq | symfreq<<16 of the final symbol */
    int   isc[64];          /* integer scalars passed between phases */
    uint8_t bytes[416];
    uint8_t res[640];
};

/* sm[] map (floats) */
#define SM_ENER   0     /* 64  band energies (modified in place by SNS) */
#define SM_GI     64    /* 64  interpolated SNS gains */
#define SM_SCF    128   /* 16 */
#define SM_SCFQ   144   /* 16 */
#define SM_TGT    160   /* 16 pvq target (dct domain) */
#define SM_TGTP   176   /* 16 pvq target pre */
#define SM_ST1    192   /* 16 */
#define SM_VEC    208   /* 6*16 = 96: candidate vectors / idct outputs */
#define SM_PVQ    304   /* 4 * 52 per-search scratch: xabs[16], y[17] (int), ynorm[16] -> 208 */
#define PVQ_STRIDE 52
#define SM_MISC   512   /* 192: R0 (98), cor, tns r[], ... */

/* isc[] map */
enum { I_T0 = 0, I_LTPF0, I_LTPF1, I_LTPF2, I_LTPF_BITS, I_BW, I_SCF0, I_SCF1, I_SCF2, I_SCF3, I_SCF4, I_SCF5, I_SCF6,
       I_TNS_NF, I_TNS_ORD0, I_TNS_ORD1, I_TNS_BITS, I_TNS_IDX0 /* 16 entries */, I_NEXT = I_TNS_IDX0 + 16 };

/* ------------------------------------------------------------------------------------------------ */
/* small helpers                                                                                     */
/* ------------------------------------------------------------------------------------------------ */
__device__ __forceinline__ float m_log2f(float x) { return (float)log2((double)x); }
__device__ __forceinline__ float m_log10f(float x) { return (float)log10((double)x); }
__device__ __forceinline__ float m_powf(float x, float y) { return (float)pow((double)x, (double)y); }
__device__ __forceinline__ float mul_d(float a, double c) { return (float)((double)a * c); }
__device__ __forceinline__ int imin(int a, int b) { return a < b ? a : b; }
__device__ __forceinline__ int imax(int a, int b) { return a > b ? a : b; }
__device__ __forceinline__ int ilog2(unsigned v) { return 31 - __clz((int)v); }

/* floor(log2f((float)v)) as glibc evaluates it: log2f rounds to an integer for the few v just below 2^b
 * (SURVEY 9): 2^21-1, 2^22-{1,2}, 2^23-{1..5}, 2^24-{1..11}. */
__device__ __forceinline__ int flog2f_int(unsigned v)
{
    int e = ilog2(v), b = e + 1;
    unsigned k = (1u << b) - v;
    if ((b == 21 && k <= 1) || (b == 22 && k <= 2) || (b == 23 && k <= 5) || (b == 24 && k <= 11)) return b;
    return e;
}

__device__ __forceinline__ float wave_max_f(float v) { for (int o = 32; o; o >>= 1) v = fmaxf(v, __shfl_xor(v, o)); return v; }
__device__ __forceinline__ int wave_max_i(int v) { for (int o = 32; o; o >>= 1) v = imax(v, __shfl_xor(v, o)); return v; }
__device__ __forceinline__ int wave_sum_i(int v) { for (int o = 32; o; o >>= 1) v += __shfl_xor(v, o); return v; }
__device__ __forceinline__ int wave_incl_scan_i(int v, int lane)
{
    for (int o = 1; o < 64; o <<= 1) { int t = __shfl_up(v, o); if (lane >= o) v += t; }
    return v;
}
/* first index of the maximum (strict '>' scan order): ties resolve to the lowest index */
__device__ __forceinline__ void wave_argmax_first(float& v, int& i, int width)
{
    for (int o = width >> 1; o; o >>= 1) {
        float ov = __shfl_xor(v, o); int oi = __shfl_xor(i, o);
        if (ov > v || (ov == v && oi < i)) { v = ov; i = oi; }
    }
}
__device__ __forceinline__ void wave_argmin_first(float& v, int& i, int width)
{
    for (int o = width >> 1; o; o >>= 1) {
        float ov = __shfl_xor(v, o); int oi = __shfl_xor(i, o);
        if (ov < v || (ov == v && oi < i)) { v = ov; i = oi; }
    }
}

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
/* frame context                                                                                     */
/* ------------------------------------------------------------------------------------------------ */
struct Scal {   /* wave-uniform cross-frame scalars (R/setup_enc_lc3.h:18-52) kept in registers */
    float hp0, hp1;
    int olpa_pitch;
    float ltpf_nc1, ltpf_nc2, ltpf_pitch; int ltpf_on;
    float att_m0, att_m1, att_acc; int att_pos, att_flag;
    float tbits_off; int mem_target, mem_spec;
};

/* ---- MDCT: R/mdct.c:103-124 + R/dct4.c:75-95 + R/fft/fft_240_480.h:16-88 / R/fft/fft_generic.h:634-699 ---- */
__device__ void st_mdct(const lc3d_plan* __restrict__ P, WaveLds& L, int lane)
{
    const int N = P->N, h = N >> 1, la = P->la;
    const float* w = &lc3t_win_pool[P->win_off];
    const float* t = &L.xbuf[MAXN - N + la];       /* t[j], j < 2N-la ; zero beyond */
    const int lim = 2 * N - la;
    for (int i = lane; i < h; i += WAVE) {
        int j0 = 3 * h - i - 1, j1 = 3 * h + i, j2 = i, j3 = 2 * h - i - 1;
        float a0 = (j0 < lim ? t[j0] : 0.0f) * w[j0];
        float a1 = (j1 < lim ? t[j1] : 0.0f) * w[j1];
        float a2 = t[j2] * w[j2];
        float a3 = t[j3] * w[j3];
        L.zb[i] = -a0 - a1;
        L.zb[h + i] = a2 - a3;
    }
    LSYNC();
    for (int i = lane; i < h; i += WAVE) {          /* pre-twiddle R/dct4.c:84-86 */
        float ar = L.zb[2 * i], ai = L.zb[N - 2 * i - 1], br = P->tw1[2 * i], bi = P->tw1[2 * i + 1];
        L.za[2 * i] = ar * br - ai * bi;
        L.za[2 * i + 1] = ai * br + ar * bi;
    }
    LSYNC();
    if (h == 240) {
        if (lane < 15) {
            float v[32];
#pragma unroll
            for (int l = 0; l < 16; l++) { int s = (225 * l + 16 * lane) % 240; v[2 * l] = L.za[2 * s]; v[2 * l + 1] = L.za[2 * s + 1]; }
            dft16(v);
#pragma unroll
            for (int l = 0; l < 16; l++) { int s = (225 * l + 16 * lane) % 240; L.za[2 * s] = v[2 * l]; L.za[2 * s + 1] = v[2 * l + 1]; }
        }
        LSYNC();
        if (lane < 16) {
            float v[30];
#pragma unroll
            for (int l = 0; l < 15; l++) { int s = (225 * lane + 16 * l) % 240; v[2 * l] = L.za[2 * s]; v[2 * l + 1] = L.za[2 * s + 1]; }
            dft15(v);
#pragma unroll
            for (int l = 0; l < 15; l++) { int d = (15 * lane + 16 * l) % 240; L.zb[2 * d] = v[2 * l]; L.zb[2 * d + 1] = v[2 * l + 1]; }
        }
        LSYNC();
    } else {   /* h == 120: prime-factor 8 x 3 x 5, index maps precomputed on the host (lc3_host.c: pfa_plan) */
        const uint8_t* m1 = P->pfa_src; const uint8_t* m2 = P->pfa_src + 120; const uint8_t* m3 = P->pfa_src + 240;
        if (lane < 15) {
            float v[16];
#pragma unroll
            for (int j = 0; j < 8; j++) { int s = m1[lane * 8 + j]; v[2 * j] = L.za[2 * s]; v[2 * j + 1] = L.za[2 * s + 1]; }
            dft8(v);
#pragma unroll
            for (int j = 0; j < 8; j++) { int d = lane * 8 + j; L.zb[2 * d] = v[2 * j]; L.zb[2 * d + 1] = v[2 * j + 1]; }
        }
        LSYNC();
        if (lane < 40) {
            float v[6];
#pragma unroll
            for (int j = 0; j < 3; j++) { int s = m2[lane * 3 + j]; v[2 * j] = L.zb[2 * s]; v[2 * j + 1] = L.zb[2 * s + 1]; }
            dft3(v);
#pragma unroll
            for (int j = 0; j < 3; j++) { int d = lane * 3 + j; L.za[2 * d] = v[2 * j]; L.za[2 * d + 1] = v[2 * j + 1]; }
        }
        LSYNC();
        float v[10];
        if (lane < 24) {
#pragma unroll
            for (int j = 0; j < 5; j++) { int s = m3[lane * 5 + j]; v[2 * j] = L.za[2 * s]; v[2 * j + 1] = L.za[2 * s + 1]; }
            dft5(v);
        }
        LSYNC();
        if (lane < 24) {
#pragma unroll
            for (int j = 0; j < 5; j++) { int d = P->pfa_dst[lane * 5 + j]; L.zb[2 * d] = v[2 * j]; L.zb[2 * d + 1] = v[2 * j + 1]; }
        }
        LSYNC();
    }
    const float norm = P->dct4_norm;
    for (int i = lane; i < h; i += WAVE) {          /* post-twiddle R/dct4.c:90-94 */
        float ar = L.zb[2 * i], ai = L.zb[2 * i + 1], br = P->tw2[2 * i], bi = P->tw2[2 * i + 1];
        float tr = ar * br - ai * bi, ti = ai * br + ar * bi;
        L.spec[2 * i] = tr * norm;
        L.spec[N - 2 * i - 1] = -ti * norm;
    }
    LSYNC();
}

/* ---- 12.8 kHz resampler + 50 Hz high-pass: R/resamp12k8.c:13-84.  Appends len12 samples to h12. ---- */
__device__ void st_resample(const lc3d_plan* __restrict__ P, WaveLds& L, Scal& S, int lane)
{
    const int mlen = P->rs_mem_in_len, stride = P->rs_stride, n12 = P->n12, len12 = P->len12;
    const float sf = P->rs_scale;
    const float* buf = &L.xbuf[MAXN - mlen];        /* [mem_in | x] */
    float* down = L.za;                             /* scratch: n12 floats */
    for (int n = lane; n < n12; n += WAVE) {
        int i = 15 * n, start = (-i) % stride;
        if (start < 0) start += stride;
        float mac = 0;
        for (int j = start; j < 240; j += stride) mac += buf[(i + j) / stride] * sf * lc3t_rs_lp[240 - j - 1];
        down[n] = mac;
    }
    /* shift the 12.8 kHz history while the FIR results settle */
    float keep[6];
#pragma unroll
    for (int k = 0; k < 6; k++) { int i = lane + 64 * k; keep[k] = (i + len12 < 384) ? L.h12[i + len12] : 0.0f; }
    LSYNC();
#pragma unroll
    for (int k = 0; k < 6; k++) { int i = lane + 64 * k; if (i + len12 < 384) L.h12[i] = keep[k]; }
    /* biquad in double, strictly serial (uniform across lanes) */
    double u11 = S.hp0, u21 = S.hp1;
    const double b0 = lc3t_hp50_b[0], b1 = lc3t_hp50_b[1], b2 = lc3t_hp50_b[2], a1 = lc3t_hp50_a[1], a2 = lc3t_hp50_a[2];
    for (int i = 0; i < len12; i++) {
        double x = (double)down[i];
        double y1 = (b0 * x + u11);
        double u1 = (b1 * x + u21) - a1 * y1;
        double u2 = b2 * x - a2 * y1;
        u11 = u1; u21 = u2;
        if (lane == 0) L.h12[384 - len12 + i] = (float)y1;
    }
    S.hp0 = (float)u11; S.hp1 = (float)u21;
    LSYNC();
}

/* ---- open-loop pitch: R/olpa.c:52-143 ---- */
__device__ void st_olpa(const lc3d_plan* __restrict__ P, WaveLds& L, Scal& S, int lane, int& T0_out, float& nc_out)
{
    const int len = P->len12, len2 = len >> 1;
    int acf = len2, back = 0;
    if (P->dms == 25) { acf += 16; back = 16; }
    /* decimate: d6[j] = sum_k dec[k] * in12[4+2j-k], in12[i] = h12[384 - len - 27 + i] */
    float nd = 0;
    if (lane < len2) {
        const float* in12 = &L.h12[384 - len - 27];
        int i = 4 + 2 * lane;
        float sum = 0;
#pragma unroll
        for (int k = 0; k < 5; k++) sum += lc3t_olpa_dec[k] * in12[i - k];
        nd = sum;
    }
    float keep[4];
#pragma unroll
    for (int k = 0; k < 4; k++) { int i = lane + 64 * k; keep[k] = (i + len2 < 194) ? L.h6[i + len2] : 0.0f; }
    LSYNC();
#pragma unroll
    for (int k = 0; k < 4; k++) { int i = lane + 64 * k; if (i + len2 < 194) L.h6[i] = keep[k]; }
    if (lane < len2) L.h6[194 - len2 + lane] = nd;
    LSYNC();
    const float* s6 = &L.h6[194 - len2 - back];
    float* R0 = &L.sm[SM_MISC];        /* 98 unweighted */
    float best = -INFINITY; int besti = 0x7fffffff;
    for (int q = lane; q < 98; q += WAVE) {
        int lag = 17 + q;
        float sum = 0;
        for (int j = 0; j < acf; j++) sum += s6[j] * s6[j - lag];
        R0[q] = sum;
        float wv = sum * lc3t_olpa_w[q];
        if (wv > best) { best = wv; besti = q; }       /* lane-local: lower q first */
    }
    wave_argmax_first(best, besti, 64);
    int T0 = besti + 17;
    LSYNC();
    float s0 = 0, s1 = 0, s2 = 0;
    for (int i = 0; i < acf; i++) { float a = s6[i], b = s6[i - T0]; s0 += a * b; s1 += b * b; s2 += a * a; }
    s1 = s1 * s2;
    s1 = sqrtf(s1) + P->c_1em5_a;
    float nc = s0 / s1;
    nc = 0 > nc ? 0 : nc;
    int lo = imax(17, S.olpa_pitch - 4), hi = imin(114, S.olpa_pitch + 4);
    int bi = 0; float bm = R0[lo - 17];
    for (int i = 0; i < hi - lo + 1; i++) { float v = R0[lo - 17 + i]; if (v > bm) { bm = v; bi = i; } }
    int T02 = bi + lo;
    if (T02 != T0) {
        s0 = s1 = s2 = 0;
        for (int i = 0; i < acf; i++) { float a = s6[i], b = s6[i - T02]; s0 += a * b; s1 += b * b; s2 += a * a; }
        s1 = s1 * s2;
        s1 = sqrtf(s1) + P->c_1em5_a;
        float nc2 = s0 / s1;
        nc2 = 0 > nc2 ? 0 : nc2;
        if ((double)nc2 > ((double)nc * 0.85)) { T0 = T02; nc = nc2; }
    }
    S.olpa_pitch = T0;
    T0_out = (int)(T0 * 2.0);
    nc_out = nc;
    LSYNC();
}

/* ---- LTPF parameter coder: R/ltpf_coder.c:34-263 ---- */
__device__ void st_ltpf(const lc3d_plan* __restrict__ P, const lc3d_chan& C, WaveLds& L, Scal& S, int lane, int pitch_ol, float ol_nc,
                        int* param, int& bits)
{
    const int len = P->len12;                 /* N of the reference = xLen - 1 */
    const float* x = &L.h12[384 - len - 24];
    int active = 0, pitch_index = 0, gain = 0;
    float norm_corr = 0, pitch = 0;
    if ((double)ol_nc > 0.6) {
        int t0_min = imax(pitch_ol - 4, 32), t0_max = imin(pitch_ol + 4, 228), acf = len;
        if (P->dms == 25) { acf = 2 * len; x = x - len; }
        const int t_min = t0_min - 4, t_max = t0_max + 4, nl = t_max - t_min + 1;
        float sum1 = 0, sum2 = 0;
        for (int j = 0; j < acf; j++) { float a = x[j], b = x[j - t_min]; sum1 += a * a; sum2 += b * b; }
        float* cor = &L.sm[SM_MISC];           /* up to 17 */
        float* cor_int = &L.sm[SM_MISC + 32];  /* up to 36 */
        if (lane < nl) {
            const int lag = t_min + lane;
            float sum = 0;
            for (int j = 0; j < acf; j++) sum += x[j] * x[j - lag];
            float s2 = sum2;
            for (int k = t_min + 1; k <= lag; k++) s2 = s2 + x[-k] * x[-k] - x[acf - 1 - (k - 1)] * x[acf - 1 - (k - 1)];
            float sum3 = sqrtf(sum1 * s2) + P->c_1em5_b;
            float nc = sum / sum3;
            nc = 0 > nc ? 0 : nc;
            cor[lane] = nc;
        }
        LSYNC();
        int tsel = 0; { float m = 0; for (int i = 0; i < t_max - t_min - 8 + 1; i++) { float v = cor[4 + i]; if (v > m) { m = v; tsel = i; } } }
        const int t1 = tsel + t0_min;
        int pitch_int, pitch_fr;
        if (t1 >= 157) { pitch_int = t1; pitch_fr = 0; }
        else {
            const int nint = 4 * (t0_max - t0_min + 1);
            if (lane < nint) {
                /* cor_up is cor zero-stuffed by 4; zero taps add +-0 to a non-negative-zero accumulator: skipped */
                float sum = 0;
                for (int k = (4 - (lane & 3)) & 3; k < 32; k += 4) {
                    int m = (lane + k) >> 2;
                    if (m <= t_max - t_min) sum += cor[m] * lc3t_ltpf_int4[k];
                }
                cor_int[lane] = sum;
            }
            LSYNC();
            const int step = t1 >= 127 ? 2 : 1;
            const int mid = 4 * (t1 - t0_min) + 1, up = 4 - step, down = t1 == t0_min ? 0 : 4 - step;
            const int cnt = ((mid + up) - (mid - down)) / step + 1;
            int ksel = 0; { float m = 0; for (int q = 0; q < cnt; q++) { float v = cor_int[mid - down - 1 + q * step]; if (v > m) { m = v; ksel = q; } } }
            pitch_fr = ksel * step - down;
            if (pitch_fr >= 0) pitch_int = t1; else { pitch_int = t1 - 1; pitch_fr = 4 + pitch_fr; }
        }
        if (pitch_int < 127) pitch_index = pitch_int * 4 + pitch_fr - 128;
        else if (pitch_int < 157) pitch_index = pitch_int * 2 + (pitch_fr / 2) - 254 + 380;
        else pitch_index = pitch_int - 157 + 380 + 60;
        pitch = (float)((double)(float)pitch_int + (double)(float)pitch_fr / 4.0);
        const float* f0 = &lc3t_ltpf_frac[0]; const float* fp = &lc3t_ltpf_frac[4 * pitch_fr];
        float* cur = L.za; float* pred = L.zb;
        for (int n = lane; n < acf; n += WAVE) {
            cur[n] = x[n + 1] * f0[0] + x[n] * f0[1] + x[n - 1] * f0[2];
            pred[n] = x[n - pitch_int + 1] * fp[0] + x[n - pitch_int] * fp[1] + x[n - pitch_int - 1] * fp[2] + x[n - pitch_int - 2] * fp[3];
        }
        LSYNC();
        float a = 0, b = 0, c = 0;
        for (int i = 0; i < acf; i++) { float cu = cur[i], pr = pred[i]; a += cu * pr; b += cu * cu; c += pr * pr; }
        b = sqrtf(b * c) + P->c_1em5_b;
        norm_corr = a / b;
        { float lo = -1 > norm_corr ? -1 : norm_corr; norm_corr = 1 < lo ? 1 : lo; }
        if (norm_corr < 0) norm_corr = 0;
        if (C.ltpf_enable == 1) {
            if ((S.ltpf_on == 0 && (P->dms == 100 || (double)S.ltpf_nc2 > 0.94) && (double)S.ltpf_nc1 > 0.94 && (double)norm_corr > 0.94) ||
                (S.ltpf_on == 1 && (double)norm_corr > 0.9) ||
                (S.ltpf_on == 1 && fabsf(pitch - S.ltpf_pitch) < 2 && (double)(norm_corr - S.ltpf_nc1) > -0.1 && (double)norm_corr > 0.84))
                active = 1;
        }
        gain = 4;
        LSYNC();
    } else { gain = 0; norm_corr = ol_nc; pitch = 0; }
    if (gain > 0) { param[0] = 1; param[1] = active; param[2] = pitch_index; bits = 11; }
    else { param[0] = param[1] = param[2] = 0; bits = 1; }
    if (P->dms < 100) S.ltpf_nc2 = S.ltpf_nc1;
    S.ltpf_nc1 = norm_corr; S.ltpf_on = active; S.ltpf_pitch = pitch;
}

/* ---- attack detector: R/attack_detector.c:13-104 (only when attack_handling) ---- */
__device__ void st_attack(const lc3d_plan* __restrict__ P, const lc3d_chan& C, WaveLds& L, Scal& S, int lane)
{
    if (!C.attack_handling) return;
    const int N = P->N, nb = P->att_nblocks, n16 = nb * 40;
    const float* in = &L.xbuf[MAXN];
    float* p = &L.za[2];
    float mval = 0;
    for (int j = lane; j < n16; j += WAVE) {
        float v;
        if (P->fs == 96000) { const float* q = &in[6 * j]; v = q[0] + q[1] + q[2] + q[3] + q[4] + q[5]; }
        else if (P->fs == 48000) { const float* q = &in[3 * j]; v = (q[0] + q[1] + q[2]); }
        else if (P->fs == 32000) { const float* q = &in[2 * j]; v = (q[0] + q[1]); }
        else { const float* q = &in[3 * j]; v = (float)((double)q[0] + ((double)(q[1] + q[2])) / 2.0); }
        p[j] = v;
    }
    if (P->fs == 96000) mval = 1e-5f;
    if (lane == 0) { p[-2] = S.att_m0; p[-1] = S.att_m1; }
    LSYNC();
    S.att_m0 = p[n16 - 2]; S.att_m1 = p[n16 - 1];
    float* fs = L.zb;
    for (int i = lane; i < 160; i += WAVE) {
        float t = 0;
        t = (float)((double)t + (double)p[i] * 0.375);
        t = (float)((double)t + (double)p[i - 1] * (-0.5));
        t = (float)((double)t + (double)p[i - 2] * (0.125));
        fs[i] = t;
    }
    LSYNC();
    float e = 0;
    if (lane < nb) { for (int k = 0; k < 40; k++) { float v = fs[k + lane * 40]; e += v * v; } }
    int flag = 0, pos = -1;
    float acc = S.att_acc;
    for (int b = 0; b < nb; b++) {
        float nrg = __shfl(e, b);
        float t = (float)((double)nrg / 8.5);
        if (t > (acc > mval ? acc : mval)) { flag = 1; pos = b + 1; }
        double q = 0.25 * (double)acc;
        acc = (double)nrg > q ? nrg : (float)q;
    }
    S.att_acc = acc;
    if (S.att_pos > P->att_hang) flag = 1;
    S.att_flag = flag;
    S.att_pos = pos;
    LSYNC();
    (void)N;
}

/* ---- per-band energy R/per_band_energy.c:13-30, bandwidth detector R/detect_cutoff_warped.c:13-83 ---- */
__device__ int st_energy_bw(const lc3d_plan* __restrict__ P, WaveLds& L, int lane)
{
    const uint16_t* be = &lc3t_band_pool[P->band_off];
    float* en = &L.sm[SM_ENER];
    if (lane < P->nbands) {
        int a = be[lane], b = be[lane + 1];
        float sum = 0;
        for (int j = a; j < b; j++) { float v = L.spec[j]; sum += v * v; }
        en[lane] = sum / (float)(b - a);
    }
    LSYNC();
    int bw = P->fs_idx;
    if (P->fs_idx > 0 && P->hrmode == 0) {
        const int f = P->fs_idx;
        const uint8_t* st = &lc3t_bw_start[(P->bw_cls * 4 + f - 1) * 4]; const uint8_t* sp = &lc3t_bw_stop[(P->bw_cls * 4 + f - 1) * 4];
        int counter = f;
        float sum = 0;
        for (int i = st[counter - 1]; i <= sp[counter - 1]; i++) sum += en[i];
        float mean = sum / (float)(sp[counter - 1] - st[counter - 1] + 1);
        while (mean < (float)lc3t_bw_quiet_thr[counter - 1]) {
            counter--;
            if (counter == 0) break;
            sum = 0;
            for (int i = st[counter - 1]; i <= sp[counter - 1]; i++) sum += en[i];
            mean = sum / (float)(sp[counter - 1] - st[counter - 1] + 1);
        }
        bw = counter;
        if (bw < f) {
            float thr = (float)lc3t_bw_brick_thr[counter];
            int stop = st[counter], dist = lc3t_bw_brick_dist[counter], brick = 0;
            for (int i = stop; i >= stop - dist; i--) {
                float ediff = (float)(10.0 * (double)m_log10f(en[i - dist + 1] + 1.1920928955078125e-07f) -
                                      10.0 * (double)m_log10f(en[i + 1] + 1.1920928955078125e-07f));
                if (ediff > thr) { brick = 1; break; }
            }
            if (!brick) bw = f;
        }
    }
    return bw;
}

/* ---- SNS scale factors R/sns_compute_scf.c:13-176 ---- */
__device__ void st_sns_scf(const lc3d_plan* __restrict__ P, WaveLds& L, int lane, int smooth)
{
    float* x = &L.sm[SM_ENER];
    float* tmp = &L.sm[SM_MISC];
    int nb = P->nbands;
    if (nb < 64) {
        int d = 64 - nb; float v;
        if (d < nb) { v = lane < 2 * d ? x[lane >> 1] : x[lane - d]; }
        else {
            float ratio = fabsf((float)(1.0 - 32.0 / (double)(float)nb));
            int n4 = (int)round((double)(ratio * (float)nb));
            int idx = lane < 4 * n4 ? (lane >> 2) : n4 + ((lane - 4 * n4) >> 1);
            v = x[idx];
        }
        LSYNC();
        x[lane] = v;
        LSYNC();
        nb = 64;
    }
    {   /* smoothing + pre-emphasis */
        float c = x[lane], m = lane > 0 ? x[lane - 1] : x[0], p = lane < 63 ? x[lane + 1] : x[63];
        float s = (float)(0.5 * (double)c + 0.25 * (double)m + 0.25 * (double)p);
        s = s * P->sns_preemph[lane];
        LSYNC();
        x[lane] = s;
        LSYNC();
    }
    float sum = 0;
    for (int i = 0; i < 64; i++) sum += x[i];
    float mean = sum / (float)64;
    float nf = mean * P->c_1em4;
    nf = nf > P->c_2m32 ? nf : P->c_2m32;
    float xv = x[lane];
    if (xv < nf) xv = nf;
    float xl = (float)((double)m_log2f(xv) / 2.0);
    tmp[lane] = xl;
    LSYNC();
    float* xl4 = &L.sm[SM_MISC + 64];
    if (lane < 16) {
        const float W[6] = {(float)(1.0 / 12.0), (float)(2.0 / 12.0), (float)(3.0 / 12.0), (float)(3.0 / 12.0), (float)(2.0 / 12.0), (float)(1.0 / 12.0)};
        float t[6];
        if (lane == 0) { t[0] = tmp[0]; for (int i = 0; i < 5; i++) t[1 + i] = tmp[i]; }
        else if (lane == 15) { for (int i = 0; i < 5; i++) t[i] = tmp[59 + i]; t[5] = tmp[63]; }
        else for (int i = 0; i < 6; i++) t[i] = tmp[lane * 4 - 1 + i];
        float s = 0;
#pragma unroll
        for (int i = 0; i < 6; i++) s += t[i] * W[i];
        xl4[lane] = s;
    }
    LSYNC();
    sum = 0;
    for (int i = 0; i < 16; i++) sum += xl4[i];
    mean = (float)((double)sum / ((double)(float)nb / 4.0));
    float* g = &L.sm[SM_SCF];
    if (lane < 16) g[lane] = P->sns_damping * (xl4[lane] - mean);
    LSYNC();
    if (smooth) {
        float gs = 0;
        if (lane < 16) {
            if (lane == 0) gs = (float)((double)(g[0] + g[1] + g[2]) / 3.0);
            else if (lane == 1) gs = (float)((double)(g[0] + g[1] + g[2] + g[3]) / 4.0);
            else if (lane == 14) gs = (float)((double)(g[12] + g[13] + g[14] + g[15]) / 4.0);
            else if (lane == 15) gs = (float)((double)(g[13] + g[14] + g[15]) / 3.0);
            else gs = (float)((double)(g[lane - 2] + g[lane - 1] + g[lane] + g[lane + 1] + g[lane + 2]) / 5.0);
            xl4[lane] = gs;
        }
        LSYNC();
        sum = 0;
        for (int i = 0; i < 16; i++) sum += xl4[i];
        mean = sum / (float)16;
        if (lane < 16) g[lane] = P->att_damping * (gs - mean);
        LSYNC();
    }
}

/* ---- PVQ pulse search R/sns_quantize_scf.c:43-136; one lane per search, scratch in LDS ---- */
__device__ void pvq_search_lane(const lc3d_plan* __restrict__ P, const float* x_in, int dim, int pulses, float* xabs, int* y, float* ynorm)
{
    float xsum = 0, yy = 0, xy = 0;
    for (int i = 0; i < dim; i++) xabs[i] = fabsf(x_in[i]);
    for (int i = 0; i < dim; i++) xsum += xabs[i];
    for (int i = 0; i < 17; i++) y[i] = 0;
    if (xsum > P->c_2m24) {
        int tot = 0;
        float proj = (float)(pulses - 1) / xsum;
        for (int i = 0; i < dim; i++) {
            int yi = (int)floorf(xabs[i] * proj);
            y[i] = yi; tot += yi;
            yy = yy + (float)(yi * yi);
            xy = xy + xabs[i] * (float)yi;
        }
        yy = yy * 0.5f;
        while (tot < pulses) {
            int imx = 0; float cnum = -P->c_2p15, cden = 0;
            yy = yy + 0.5f;
            for (int i = 0; i < dim; i++) {
                float a = xy + xabs[i]; a = a * a;
                float b = yy + (float)y[i];
                if (a * cden > b * cnum) { cnum = a; cden = b; imx = i; }
            }
            xy = xy + xabs[imx]; yy = yy + (float)y[imx]; y[imx] = y[imx] + 1; tot++;
        }
        yy = yy * 2.0f;
    } else {
        if (dim > 1) { y[0] = pulses / 2; y[dim] = -(pulses - pulses / 2); yy = (float)(y[0] * y[0] + y[dim] * y[dim]); }
        else { y[1] = pulses; yy = (float)(pulses * pulses); }
    }
    float g = (float)(1.0 / (double)sqrtf(yy));
    for (int i = 0; i < dim; i++) { int s = x_in[i] >= 0 ? 1 : -1; int yi = y[i] * s; y[i] = yi; ynorm[i] = (float)yi * g; }
}

/* MPVQ enumeration R/sns_quantize_scf.c:138-163 (integer) */
__device__ void mpvq_index(const int* pulses, int len, int& ls, int& idx)
{
    int k = 0; ls = -1; idx = 0;
    for (int pos = len - 1; pos >= 0; pos--) {
        int pv = pulses[pos];
        if (ls >= 0 && pv != 0) idx = 2 * idx + ls;
        if (pv > 0) ls = 0;
        if (pv < 0) ls = 1;
        idx = idx + (int)lc3t_mpvq_offs[(len - pos - 1) * 11 + k];
        k += pv < 0 ? -pv : pv;
    }
}

/* ---- SNS vector quantiser R/sns_quantize_scf.c:165-430 (+ DCT-II(16) R/dct4.c:28-48, IDCT-II :19-41) ---- */
__device__ void st_sns_vq(const lc3d_plan* __restrict__ P, WaveLds& L, int lane)
{
    const float* env = &L.sm[SM_SCF];
    float* st1 = &L.sm[SM_ST1]; float* tgt = &L.sm[SM_TGT]; float* tgtp = &L.sm[SM_TGTP];
    float* vec = &L.sm[SM_VEC];
    int* isc = L.isc;
    {   /* stage 1: lane = sec*32 + codeword */
        const int sec = lane >> 5, c = lane & 31;
        const float* cb = sec ? lc3t_sns_hf : lc3t_sns_lf;
        float sum = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) { float d = env[8 * sec + i] - cb[c * 8 + i]; sum += d * d; }
        int bi = c;
        wave_argmin_first(sum, bi, 32);
        if (c == 0) isc[I_SCF0 + sec] = bi;
        if (c < 8) { float s = cb[bi * 8 + c]; st1[8 * sec + c] = s; tgtp[8 * sec + c] = env[8 * sec + c] - s; }
    }
    LSYNC();
    {   /* DCT-II(16): every lane runs the 16-point DFT on the same data, lanes < 16 keep one output */
        float z[32];
#pragma unroll
        for (int i = 0; i < 8; i++) { z[2 * i] = tgtp[2 * i]; z[2 * i + 1] = 0; z[2 * (15 - i)] = tgtp[2 * i + 1]; z[2 * (15 - i) + 1] = 0; }
        dft16(z);
        float o = 0;
#pragma unroll
        for (int i = 0; i < 16; i++) if (lane == i) o = z[2 * i] * P->dct2_tw[2 * i] - z[2 * i + 1] * P->dct2_tw[2 * i + 1];
        if (lane == 0) o = o / P->c_sqrt2;
        if (lane < 16) tgt[lane] = o;
    }
    LSYNC();
    /* four pulse searches, one lane each: 0:(N=10,K=10) 1:(N=6,K=1 on tgt+10) 2:(N=16,K=8) 3:(N=16,K=6) */
    float* pv = &L.sm[SM_PVQ];
    if (lane < 4) {
        float* xabs = pv + lane * PVQ_STRIDE; int* y = (int*)(pv + lane * PVQ_STRIDE + 16); float* yn = pv + lane * PVQ_STRIDE + 33;
        const int dim = lane == 0 ? 10 : lane == 1 ? 6 : 16, K = lane == 0 ? 10 : lane == 1 ? 1 : lane == 2 ? 8 : 6;
        for (int i = 0; i < 16; i++) yn[i] = 0;
        pvq_search_lane(P, lane == 1 ? tgt + 10 : tgt, dim, K, xabs, y, yn);
    }
    LSYNC();
    const int* pA = (const int*)(pv + 16); const int* pB = (const int*)(pv + PVQ_STRIDE + 16);
    const int* pN = (const int*)(pv + 2 * PVQ_STRIDE + 16); const int* pF = (const int*)(pv + 3 * PVQ_STRIDE + 16);
    const float* nA = pv + 33; const float* nN = pv + 2 * PVQ_STRIDE + 33; const float* nF = pv + 3 * PVQ_STRIDE + 33;
    /* yC = [pA(10) | pB(6)], normalised */
    float sumy = 0;
    for (int i = 0; i < 16; i++) { int yi = i < 10 ? pA[i] : pB[i - 10]; sumy += (float)(yi * yi); }
    const float gf = (float)(1.0 / (double)sqrtf(sumy));
    const int yCl = lane < 16 ? (lane < 10 ? pA[lane] : pB[lane - 10]) : 0;
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
        if (lane < 6) { float s = 0; for (int j = 0; j < 16; j++) { float d = tgt[j] - vec[lane * 16 + j]; s += d * d; } err = s; }
        float min_err = P->c_2p15;
        for (int i = 0; i < 6; i++) { float e = __shfl(err, i); if (e < min_err) { min_err = e; idx = i; } }
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
        for (int j = 0; j < 16; j++) {
            float t = (float)((double)in[j] * P->idct_cos[i * 16 + j]);
            if (j == 0) t *= P->c_idct_n2;
            sum += t;
        }
        idc[48 + lane] = P->c_idct_n1 * sum;
    }
    LSYNC();
    const float* split = &idc[48]; const float* subN = &idc[64]; const float* subF = &idc[80];
    float e_split = 0;
    for (int i = 0; i < 16; i++) { float d = tgtp[i] - glob * split[i]; e_split += d * d; }
    int sub_mode = 0, sub_gain = 0, shape = 0;    /* shape: 0 = yC, 1 = pA only, 2 = near, 3 = far */
    float e_sofar = P->c_2p15, g_sel = 0; const float* v_sel = split;
    bool have = false;
    if (e_split < e_sofar) {
        if (idx <= 1) { sub_mode = 0; sub_gain = idx; shape = 0; } else { sub_mode = 1; sub_gain = idx - 2; shape = 1; }
        g_sel = glob; v_sel = split; e_sofar = e_split; have = true;
    }
    {   /* near (4 gains, lanes 0-3) and far (8 gains, lanes 4-11) errors */
        float err = INFINITY;
        if (lane < 12) {
            const float g = lane < 4 ? lc3t_sns_gain_near[lane] : lc3t_sns_gain_far[lane - 4];
            const float* sb = lane < 4 ? subN : subF;
            float s = 0;
            for (int j = 0; j < 16; j++) { float d = tgtp[j] - g * sb[j]; s += d * d; }
            err = s;
        }
        float min_err = P->c_2p15; int gi = idx; float gg = glob;
        for (int i = 0; i < 4; i++) { float e = __shfl(err, i); if (e < min_err) { gi = i; min_err = e; gg = lc3t_sns_gain_near[i]; } }
        if (min_err < e_sofar) { sub_mode = 2; sub_gain = gi; shape = 2; g_sel = gg; v_sel = subN; e_sofar = min_err; have = true; }
        min_err = P->c_2p15;
        for (int i = 0; i < 8; i++) { float e = __shfl(err, 4 + i); if (e < min_err) { gi = i; min_err = e; gg = lc3t_sns_gain_far[i]; } }
        if (min_err < e_sofar) { sub_mode = 3; sub_gain = gi; shape = 3; g_sel = gg; v_sel = subF; have = true; }
    }
    if (lane < 16) {
        float st2 = have ? g_sel * v_sel[lane] : 0.0f;
        L.sm[SM_SCFQ + lane] = st1[lane] + st2;
    }
    if (lane == 0) {
        int pulses[16];
        for (int i = 0; i < 16; i++)
            pulses[i] = shape == 0 ? (i < 10 ? pA[i] : pB[i - 10]) : shape == 1 ? (i < 10 ? pA[i] : 0) : shape == 2 ? pN[i] : pF[i];
        if (!have) for (int i = 0; i < 16; i++) pulses[i] = 0;
        int ls, mi;
        if (sub_mode < 2) mpvq_index(pulses, 10, ls, mi); else mpvq_index(pulses, 16, ls, mi);
        int i6;
        if (sub_mode == 0) { int a, b; mpvq_index(pulses + 10, 6, a, b); i6 = b * 2 + a; }
        else if (sub_mode == 2) i6 = -1; else i6 = -2;
        isc[I_SCF2] = sub_mode; isc[I_SCF3] = sub_gain; isc[I_SCF4] = ls; isc[I_SCF5] = mi; isc[I_SCF6] = i6;
    }
    LSYNC();
}

/* ---- SNS interpolation R/sns_interpolate_scf.c:13-89 and spectral shaping R/mdct_shaping.c:13-22 ---- */
__device__ void st_sns_apply(const lc3d_plan* __restrict__ P, WaveLds& L, int lane)
{
    const float* g = &L.sm[SM_SCFQ];
    float* gi = &L.sm[SM_GI];
    float* tmp = &L.sm[SM_MISC];
    {
        float v;
        if (lane < 2) v = g[0];
        else if (lane < 62) {
            int n = (lane - 2) >> 2, r = (lane - 2) & 3;
            float d = g[n + 1] - g[n];
            double dd = (double)d;
            if (r == 0) v = (float)((double)g[n] + dd / 8.0);
            else if (r == 1) v = (float)((double)g[n] + 3.0 * dd / 8.0);
            else if (r == 2) v = (float)((double)g[n] + 5.0 * dd / 8.0);
            else v = (float)((double)g[n] + 7.0 * dd / 8.0);
        } else {
            double dd = (double)(g[15] - g[14]);
            v = lane == 62 ? (float)((double)g[15] + dd / 8.0) : (float)((double)g[15] + 3.0 * dd / 8.0);
        }
        gi[lane] = v;
    }
    LSYNC();
    const int nb = P->nbands;
    if (nb < 64) {
        const int d = 64 - nb;
        float v = 0;
        if (d < 32) {
            if (lane < d) v = (float)((double)(gi[2 * lane] + gi[2 * lane + 1]) / 2.0);
            else if (lane < nb) v = gi[lane + d];
        } else {
            float ratio = fabsf((float)(1.0 - 32.0 / (double)(float)nb));
            int n4 = (int)round((double)(ratio * (float)nb));
            if (lane < n4) v = (float)((double)(gi[4 * lane] + gi[4 * lane + 1] + gi[4 * lane + 2] + gi[4 * lane + 3]) / 4.0);
            else if (lane < nb) { int i = lane - n4; v = (float)((double)(gi[4 * n4 + 2 * i] + gi[4 * n4 + 2 * i + 1]) / 2.0); }
        }
        LSYNC();
        gi[lane] = v;
        LSYNC();
    }
    if (lane < nb) { float v = -gi[lane]; tmp[lane] = m_powf(2.0f, v); }
    LSYNC();
    if (lane < nb) gi[lane] = tmp[lane];
    LSYNC();
    for (int j = lane; j < P->N; j += WAVE) {
        int b = P->band_of_bin[j];
        if (b < nb) L.spec[j] = L.spec[j] * gi[b];
    }
    LSYNC();
}

/* ---- TNS analysis + lattice filter R/tns_coder.c:170-362 ---- */
__device__ void st_tns(const lc3d_plan* __restrict__ P, const lc3d_chan& C, WaveLds& L, int lane, int bw_idx, int bw_bin)
{
    int fs = P->fs, N = P->N; const int nBits = C.total_bits, dms = P->dms;
    int numfilters = (fs >= 32000 && dms >= 50) ? 2 : 1;
    int start[2] = {0, 0}, stop[2] = {0, 0};
    if ((double)N > 40 * ((double)(float)dms / 10.0)) { N = (int)(40 * ((double)(float)dms / 10.0)); fs = 40000; }
    start[0] = (600 * N * 2 / fs) + 1;
    if (numfilters == 1) stop[0] = N; else { start[1] = N / 2 + 1; stop[0] = N / 2; stop[1] = N; }
    const int maxOrder = dms == 100 ? 8 : 4; const int nSub = dms == 100 ? 3 : 2;
    float maxPG = 2; const float minPG = 1.5f;
    const uint16_t* obits = &lc3t_tns_order_bits[8];
    if ((dms >= 50 && (double)nBits >= 48 * ((double)(float)dms / 10.0)) || dms == 25) { maxPG = minPG; obits = &lc3t_tns_order_bits[0]; }
    if (bw_idx >= 3 && numfilters == 2) { start[1] = bw_bin / 2 + 1; stop[0] = bw_bin / 2; stop[1] = bw_bin; }
    else { numfilters = 1; stop[0] = bw_bin; }
    float* racc = &L.sm[SM_MISC];              /* [f][sub][k] 2*3*9 = 54, then energies 6 at +54, r[f][9] at +64 */
    int* isc = L.isc;
    {   /* one serial sum per lane: lanes 0..53 autocorrelation terms, 54..59 sub-division energies */
        int f, sub, k = -1;
        if (lane < 54) { f = lane / 27; int r = lane % 27; sub = r / 9; k = r % 9; }
        else { int r = lane - 54; f = r / 3; sub = r % 3; }
        if (lane < 60 && f < numfilters && sub < nSub && (k <= maxOrder)) {
            float sublen = (float)(((double)(float)stop[f] + 1.0 - (double)(float)start[f]) / (double)(float)nSub);
            int lo = (int)(floor((double)(sublen * (float)sub)) + start[f] - 1);
            int hi = (int)(floor((double)(sublen * (float)(sub + 1))) + start[f] - 1);
            const float* x = &L.spec[lo]; const int n = hi - lo;
            float acc = 0;
            if (k < 0) { for (int i = 0; i < n; i++) acc += x[i] * x[i]; racc[54 + f * 3 + sub] = acc; }
            else { for (int i = k; i < n; i++) acc += x[i] * x[i - k]; racc[lane] = acc; }
        }
    }
    LSYNC();
    if (lane < 18) {   /* r[f][k] */
        int f = lane / 9, k = lane % 9;
        float r = 0;
        if (f < numfilters && k <= maxOrder) {
            for (int sub = 0; sub < nSub; sub++) {
                float e = racc[54 + f * 3 + sub];
                if (e == 0) { r = (k == 0) ? 1.0f : 0.0f; break; }
                r = r + racc[f * 27 + sub * 9 + k] / e;
            }
            r = r * lc3t_tns_lagwin[k];
        }
        racc[64 + lane] = r;
    }
    LSYNC();
    int bits = 0;
    float* stt = &L.sm[SM_MISC + 96];          /* lattice state between filters: 8 */
    float* rcs = &L.sm[SM_MISC + 104];         /* quantised rc of current filter: 8 */
    if (lane < 8) stt[lane] = 0;
    LSYNC();
    for (int f = 0; f < numfilters; f++) {
        /* Levinson-Durbin R/tns_coder.c:41-89, uniform across lanes, fully unrolled for register residency */
        float r[9], a[9], rc[8], buf[9];
#pragma unroll
        for (int i = 0; i < 9; i++) { r[i] = racc[64 + f * 9 + i]; a[i] = 0; }
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
        const float err = v;
        const float predGain = r[0] / err;
        int tns = predGain > minPG;
        bits++;
        int ord = 0; int idxq[8];
        if (tns) {
            if (predGain < maxPG) {
                /* LPC weighting (low rates): reorder to the polynomial, weight, back to reflection coeffs R/tns_coder.c:91-155,279-287.
                 * Small, rare and index-heavy: lane 0 works in LDS scratch. */
                float* sc = &L.sm[SM_MISC + 112];   /* a[9] at 0, out[9] at 9, t0[8] at 18, buf[9] at 26, a_in[9] at 36, rc_in[8] at 46 */
                if (lane == 0) {
#pragma unroll
                    for (int j = 0; j < 9; j++) sc[36 + j] = a[j];
#pragma unroll
                    for (int j = 0; j < 8; j++) sc[46 + j] = rc[j];
                    float* pa = sc + 56;                /* 9 */
                    pa[0] = 1;
                    for (int j = 1; j <= maxOrder - 1; j++) pa[j] = -sc[36 + maxOrder - j];
                    pa[maxOrder] = sc[46 + maxOrder - 1];
                    float alpha = (float)((double)((maxPG - predGain)) * (0.85f - 1.0) / (double)(maxPG - minPG) + 1.0);
                    for (int i = 0; i <= maxOrder; i++) sc[i] = pa[i] * m_powf(alpha, (float)i);
                    int len = maxOrder + 1; const int len0 = len;
                    float* pa_ = sc; float* out = sc + 9; float* t0 = sc + 18; float* bf = sc + 26;
                    for (int i = 0; i < len - 1; i++) out[i] = 0;
                    { float a0 = pa_[0]; for (int i = 0; i < len; i++) { pa_[i] = pa_[i] / a0; a0 = pa_[0]; } }
                    out[len - 1] = pa_[len - 1];
                    for (int k = len0 - 2; k >= 0; k--) {
                        for (int i = 0; i < len - 1; i++) t0[i] = pa_[1 + i];
                        int l = len - 1;
                        float knxt = t0[l - 1];
                        l = l - 1;
                        bf[0] = 1;
                        for (int i = 0; i < l; i++) {
                            float t2 = knxt * t0[l - 1 - i];
                            bf[i + 1] = (float)((double)(t0[i] - t2) / (1.0 - (double)(fabsf(knxt) * fabsf(knxt))));
                        }
                        len = l + 1;
                        out[k] = bf[len - 1];
                        for (int i = 0; i < len; i++) pa_[i] = bf[i];
                    }
                    for (int i = 0; i < len0 - 1; i++) out[i] = out[i + 1];
                }
                LSYNC();
#pragma unroll
                for (int i = 0; i < 8; i++) rc[i] = i < maxOrder ? sc[9 + i] : 0.0f;
                LSYNC();
            }
#pragma unroll
            for (int i = 0; i < 8; i++) {
                int ret = 0;
                if (i < maxOrder) for (int q = 0; q < 17; q++) if (rc[i] <= lc3t_tns_rc_thr[q + 1] && rc[i] > lc3t_tns_rc_thr[q]) ret = q;
                idxq[i] = ret;
            }
#pragma unroll
            for (int i = 0; i < 8; i++) { float q = i < maxOrder ? lc3t_tns_rc_pts[idxq[i]] : 0.0f; rc[i] = q; if (i < maxOrder && q != 0) ord = i + 1; }
            if (ord == 0) tns = 0;
        }
        if (lane == 0) isc[I_TNS_ORD0 + f] = tns ? ord : 0;
        if (tns) {
            int tmp = obits[ord - 1];
#pragma unroll
            for (int i = 0; i < 8; i++) if (i < ord) tmp += lc3t_tns_coef_bits[i * 17 + idxq[i]];
            bits = bits + ((tmp + 2047) >> 11);
#pragma unroll
            for (int i = 0; i < 8; i++) if (lane == 0 && i < ord) isc[I_TNS_IDX0 + f * 8 + i] = idxq[i];
            /* lattice MA filter, lane-parallel with exact replay: each lane owns a run of consecutive bins and first
             * replays the 8 preceding inputs (values before the filter start come from the carried state). */
            const int b_first = start[f] - 1, cnt = stop[f] - start[f] + 1;
            const int chunk = (cnt + WAVE - 1) / WAVE;
            const int b0 = b_first + lane * chunk;
            int nmine = imin(chunk, b_first + cnt - b0); if (nmine < 0) nmine = 0;
            float st[8];
#pragma unroll
            for (int j = 0; j < 8; j++) st[j] = stt[j];
            float carried[8];
#pragma unroll
            for (int j = 0; j < 8; j++) carried[j] = st[j];
            for (int t = b0 - 8; nmine > 0 && t < b0 + nmine; t++) {
                if (t < b_first) {
#pragma unroll
                    for (int j = 0; j < 8; j++) st[j] = carried[j];
                    continue;
                }
                float s = L.spec[t], save = s;
#pragma unroll
                for (int j = 0; j < 7; j++) {
                    if (j < ord - 1) { float tt = rc[j] * s + st[j]; s += rc[j] * st[j]; st[j] = save; save = tt; }
                }
#pragma unroll
                for (int j = 0; j < 8; j++) if (j == ord - 1) { s += rc[j] * st[j]; st[j] = save; }
                if (t >= b0) L.zb[t] = s;
            }
            const int lastLane = (cnt - 1) / chunk;
            LSYNC();
            if (lane == lastLane) {
#pragma unroll
                for (int j = 0; j < 8; j++) stt[j] = st[j];
            }
            for (int t = b_first + lane; t < b_first + cnt; t += WAVE) L.spec[t] = L.zb[t];
            LSYNC();
        }
    }
    if (lane == 0) { isc[I_TNS_NF] = numfilters; isc[I_TNS_BITS] = bits; }
    LSYNC();
    (void)rcs;
}

/* ---- global gain estimate R/estimate_global_gain.c:30-137 ---- */
__device__ void st_gain_estimate(const lc3d_plan* __restrict__ P, const lc3d_chan& C, WaveLds& L, Scal& S, int lane, int nbitsSQ,
                                 float& gain, int& qgain, int& qmin)
{
    const int lg = P->ylen, off = C.gg_off, nq = lg >> 2;
    if (S.mem_target < 0) S.tbits_off = 0;
    else {
        float v = S.tbits_off + (float)S.mem_target - (float)S.mem_spec;
        v = -40 > v ? -40 : v; v = 40 < v ? 40 : v;
        S.tbits_off = (float)(0.8 * (double)S.tbits_off + 0.2 * (double)v);
    }
    S.mem_target = nbitsSQ;
    nbitsSQ = (int)((double)nbitsSQ + round((double)S.tbits_off));
    float xm = 0;
    for (int i = lane; i < lg; i += WAVE) xm = fmaxf(xm, fabsf(L.spec[i]));
    const float x_max = wave_max_f(xm);
    float reg_val = 0;
    if (P->hrmode && C.reg_bits > 0) {
        float M0 = 1e-5f, M1 = 1e-5f; const float thresh = 2 * P->frame_ms;
        for (int i = 0; i < lg; i++) { double ax = fabs((double)L.spec[i]); M0 = (float)((double)M0 + ax); M1 = (float)((double)M1 + (double)i * ax); }
        float q = M1 / M0;
        float rB = 8 * (1 - (q < thresh ? q : thresh) / thresh);
        reg_val = x_max * m_powf(2.0f, (float)(-C.reg_bits) - rB);
    }
    float ind = 0, ind_min = 0;
    if (x_max == 0) { ind_min = (float)off; ind = 0; S.mem_target = -1; }
    else {
        float g_min = P->hrmode == 1 ? x_max / (float)(32768 * 256 - 2) : (float)((double)x_max / (32768 - 0.375));
        ind_min = (float)ceil(28.0 * (double)m_log10f(g_min));
        float* en = L.za;
        for (int j = lane; j < nq; j += WAVE) {
            const float* x = &L.spec[4 * j];
            float t = x[0] * x[0];
            t += x[1] * x[1]; t += x[2] * x[2]; t += x[3] * x[3];
            en[j] = (float)((28.0 / 20.0) * (7 + 10.0 * (double)m_log10f(t + reg_val + P->c_2m31)));
        }
        LSYNC();
        const float target = (float)((28.0 / 20.0) * (1.4) * (double)nbitsSQ);
        const int offset0 = 255 + off;
        /* 8-step bisection, evaluated speculatively: lanes 1..63 are the decision-tree nodes of the first six steps */
        int m = 0;
        {
            const int lvl = lane ? ilog2((unsigned)lane) : 0, p = lane - (1 << lvl);
            const int cand = offset0 - (p << (8 - lvl)) - (128 >> lvl);
            float ener = 0; int iszero = 1;
            for (int j = nq - 1; j >= 0; j--) {
                float t = en[j] - (float)cand;
                if ((double)t < (7.0) * (28.0 / 20.0)) { if (iszero == 0) ener = (float)((double)ener + (2.7) * (28.0 / 20.0)); }
                else {
                    if ((double)t > (50.0) * (28.0 / 20.0)) ener = (float)((double)ener + 2.0 * (double)t - (50.0) * (28.0 / 20.0));
                    else ener = ener + t;
                    iszero = 0;
                }
            }
            const unsigned long long addback = __ballot(ener > target && iszero == 0);
            int node = 1;
            for (int i = 0; i < 6; i++) { int nb = ((addback >> node) & 1ull) ? 0 : 1; m += nb << (7 - i); node = 2 * node + nb; }
        }
        {   /* last two steps: lane 0: step 6; lane 1: step 7 if step 6 added back; lane 2: step 7 otherwise */
            const int cand = lane == 0 ? offset0 - m - 2 : lane == 1 ? offset0 - m - 1 : offset0 - m - 3;
            float ener = 0; int iszero = 1;
            for (int j = nq - 1; j >= 0; j--) {
                float t = en[j] - (float)cand;
                if ((double)t < (7.0) * (28.0 / 20.0)) { if (iszero == 0) ener = (float)((double)ener + (2.7) * (28.0 / 20.0)); }
                else {
                    if ((double)t > (50.0) * (28.0 / 20.0)) ener = (float)((double)ener + 2.0 * (double)t - (50.0) * (28.0 / 20.0));
                    else ener = ener + t;
                    iszero = 0;
                }
            }
            const unsigned long long addback = __ballot(ener > target && iszero == 0);
            if (addback & 1ull) { if (!(addback & 2ull)) m += 1; }
            else { m += 2; if (!(addback & 4ull)) m += 1; }
        }
        const int offset = offset0 - m;
        if ((float)offset < ind_min) S.mem_target = -1;
        ind = (ind_min > (float)offset ? ind_min : (float)offset) - (float)off;
        LSYNC();
    }
    qmin = (int)ind_min; qgain = (int)ind;
    gain = P->gain_est[(int)(ind + (float)off) + 256];
}

/* ---- quantisation + exact bit estimate R/quantize_spec.c:26-197 ---- */
__device__ void st_quantize(const lc3d_plan* __restrict__ P, const lc3d_chan& C, WaveLds& L, int lane, float gain, int mode, int target,
                            int& nbits_o, int& nbits2_o, int& lastnz_o, int& lsb_o)
{
    const int nt = P->ylen, fs = P->fs, tb = C.total_bits;
    const float offs = P->hrmode ? 0.5f : 0.375f;
    for (int i = lane; i < nt; i += WAVE) {
        float x = L.spec[i];
        int sg = x > 0 ? 1 : x < 0 ? -1 : 0;
        L.xq[i] = (int)truncf(x / gain + offs * (float)sg);
    }
    int rate = 0;
    if ((fs < 48000 && tb > 320 + (fs / 8000 - 2) * 160) || (fs == 48000 && tb > 800)) rate = 512;
    if (mode == 0 && ((fs < 48000 && tb >= 640 + (fs / 8000 - 2) * 160) || (fs == 48000 && tb >= 1120))) mode = 1;
    LSYNC();
    int lp = 0;
    for (int p = lane; p < (nt >> 1); p += WAVE) if (p >= 1 && (L.xq[2 * p] != 0 || L.xq[2 * p + 1] != 0)) lp = p;
    lp = wave_max_i(lp);
    const int lastnz = lp >= 1 ? 2 * lp + 1 : 1;
    const int ntup = (lastnz + 1) >> 1;
    int lastnz2 = mode < 0 ? lastnz + 1 : 2;
    int nbits2 = 0, base = 0, nlsb = 0, ct1 = 0, ct2 = 0;
    for (int c0 = 0; c0 < ntup; c0 += WAVE) {
        const int p = c0 + lane; const bool act = p < ntup;
        const int x0 = act ? L.xq[2 * p] : 0, x1 = act ? L.xq[2 * p + 1] : 0;
        const int a0 = x0 < 0 ? -x0 : x0, b0 = x1 < 0 ? -x1 : x1, mx = imax(a0, b0);
        const int nsh = mx >= 4 ? ilog2((unsigned)mx) - 1 : 0;
        const int af = a0 >> nsh, bf = b0 >> nsh;
        const int lev1 = imin(nsh, 3), levm = lev1 - 1;
        const int tval = levm <= 0 ? 1 + (af + bf) * (levm + 2) : 13 + levm;
        int t1 = __shfl_up(tval, 1), t2 = __shfl_up(tval, 2);
        if (lane == 0) { t1 = ct1; t2 = ct2; } else if (lane == 1) t2 = ct1;
        int tin = 16 * (t2 & 15) + t1 + rate;
        if (2 * p > nt / 2) tin += 256;
        const int maxlev = mx == 0 ? -1 : flog2f_int((unsigned)imax(mx, 3)) - 1;
        int bits = 0, lsbc = 0;
        if (act) {
            if (mode <= 0) bits += (imin(a0, 1) + imin(b0, 1)) * 2048;
            for (int lev = 0; lev < nsh; lev++) {
                int pki = lc3t_ac_ctx_lut[tin + imin(lev, 3) * 1024];
                bits += lc3t_ac_bits[pki * 17 + 16];
                if (lev == 0 && mode > 0) lsbc += 2; else bits += 2 * 2048;
            }
            const int pki = lc3t_ac_ctx_lut[tin + lev1 * 1024], sym = af + 4 * bf;
            bits += lc3t_ac_bits[pki * 17 + sym];
            if (mode > 0) {
                int am = a0, bm = b0;
                if (lev1 > 0) { am >>= 1; bm >>= 1; if (am == 0 && x0 != 0) lsbc++; if (bm == 0 && x1 != 0) lsbc++; }
                bits += (imin(am, 1) + imin(bm, 1)) * 2048;
            }
            L.cd[p] = (uint32_t)tin | ((uint32_t)(maxlev + 1) << 10) | ((uint32_t)sym << 16);
            const int pk2 = lc3t_ac_ctx_lut[tin + imin(imax(maxlev, 0), 3) * 1024];
            const uint32_t cl = lc3t_ac_cum[pk2 * 18 + sym], ch = lc3t_ac_cum[pk2 * 18 + sym + 1];
            L.cf[p] = cl | ((ch - cl) << 16);
        }
        const int incl = wave_incl_scan_i(bits, lane) + base;
        const unsigned long long ok = __ballot(act && mode >= 0 && (a0 != 0 || b0 != 0) && incl <= target * 2048);
        if (ok) { int hl = 63 - __clzll((long long)ok); lastnz2 = 2 * (c0 + hl) + 2; nbits2 = __shfl(incl, hl); }
        base = __shfl(incl, 63);
        nlsb += wave_sum_i(lsbc);
        ct2 = __shfl(tval, 62); ct1 = __shfl(tval, 63);
    }
    int nbits = (base + 2047) >> 11;
    if (mode >= 0) nbits2 = (nbits2 + 2047) >> 11; else nbits2 = nbits;
    if (mode > 0) { nbits += nlsb; nbits2 += nlsb; }
    LSYNC();
    for (int i = lastnz2 + lane; i <= lastnz; i += WAVE) L.xq[i] = 0;
    lsb_o = (mode > 0 && nbits > target) ? 1 : 0;
    lastnz_o = lastnz2; nbits_o = nbits; nbits2_o = nbits2;
    LSYNC();
}

/* ---- R/adjust_global_gain.c:13-50 ---- */
__device__ void st_gain_adjust(const lc3d_plan* __restrict__ P, const lc3d_chan& C, int& gg, int gg_min, float& gain, int target, int nBits, int& change)
{
    const int f = P->fs_idx, off = C.gg_off;
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

/* ---- noise factor R/noise_factor.c:13-108 ---- */
__device__ int st_noise_factor(const lc3d_plan* __restrict__ P, const lc3d_chan& C, WaveLds& L, int lane, float gg, int bw_bin)
{
    const int width = P->dms == 100 ? 8 : 4, first = P->dms == 100 ? 24 : P->dms == 50 ? 12 : 6, hw = (width - 2) / 2;
    float* val = L.za; int* zk = (int*)L.zb;
    int nz = 0, sumz = 0;
    for (int k0 = first; k0 < bw_bin; k0 += WAVE) {
        const int k = k0 + lane;
        bool allz = false;
        if (k < bw_bin) {
            allz = true;
            const int lo = k - hw, hi = imin(bw_bin - 1, k + hw);
            for (int i = lo; i <= hi; i++) if (L.xq[i] != 0) allz = false;
        }
        const unsigned long long mk = __ballot(allz);
        if (allz) {
            const int pos = nz + __popcll(mk & ((1ull << lane) - 1ull));
            val[pos] = fabsf(L.spec[k] / gg); zk[pos] = k + 1;
        }
        nz += __popcll(mk);
        sumz += wave_sum_i(allz ? k + 1 : 0);
    }
    LSYNC();
    float fac = 0;
    if (sumz > 0) { float mean = 0; for (int j = 0; j < nz; j++) mean += val[j]; fac = mean / (float)nz; }
    if (C.nbytes <= 20 && P->dms == 100 && nz > 0) {
        const int m = sumz / nz; int j = 0, k = 0; float m1 = 0, m2 = 0;
        for (int i = 0; i < nz; i++) { if (zk[i] <= m) { m1 += val[i]; j++; } else { m2 += val[i]; k++; } }
        float n1 = m1 / (float)j, n2 = m2 / (float)k;
        fac = n1 < n2 ? n1 : n2;
    }
    float idx = (float)round((double)(8 - 16 * fac));
    { float t = idx > 0 ? idx : 0; idx = t < 7 ? t : 7; }
    LSYNC();
    return (int)idx;
}

/* ---- residual coding R/residual_coding.c:13-75 ---- */
__device__ int st_residual(const lc3d_plan* __restrict__ P, WaveLds& L, int lane, float gain, int targetBits, int nBits)
{
    int* nzi = (int*)L.za;
    int nnz = 0;
    for (int k0 = 0; k0 < P->ylen; k0 += WAVE) {
        const int k = k0 + lane;
        const bool nzq = k < P->ylen && L.xq[k] != 0;
        const unsigned long long mk = __ballot(nzq);
        if (nzq) nzi[nnz + __popcll(mk & ((1ull << lane) - 1ull))] = k;
        nnz += __popcll(mk);
    }
    int m = targetBits - nBits + 4;
    if (P->hrmode) m += 10;
    const int iter_max = P->hrmode ? 20 : 1;
    for (int i = lane; i < 160; i += WAVE) ((uint32_t*)L.res)[i] = 0;
    LSYNC();
    int n = 0, iter = 0; float offset = .25f;
    while (iter < iter_max && n < m) {
        for (int k0 = 0; k0 < nnz && n < m; k0 += WAVE) {
            const int k = k0 + lane;
            const bool act = k < nnz && (n + lane) < m;
            int bit = 0;
            if (act) {
                const int id = nzi[k];
                const float x = L.spec[id];
                if (x >= (float)L.xq[id] * gain) { bit = 1; L.spec[id] = x - gain * offset; } else { L.spec[id] = x + gain * offset; }
            }
            const unsigned long long bm = __ballot(bit);
            const unsigned long long am = __ballot(act);
            const int cnt = __popcll(am);
            /* n is a multiple of 64 here except across iterations in hrmode; handle the general bit offset */
            if (lane < 9) {
                const int sh = n & 7; const int byte0 = n >> 3;
                unsigned long long lo = bm << sh; unsigned hi = sh ? (unsigned)(bm >> (64 - sh)) : 0u;
                unsigned v = lane < 8 ? (unsigned)((lo >> (8 * lane)) & 0xff) : (hi & 0xff);
                if (byte0 + lane < 640 && v) L.res[byte0 + lane] |= (uint8_t)v;
            }
            n += cnt;
            LSYNC();
        }
        iter++; offset *= .5f;
    }
    return n;
}

/* ---- bitstream writers: side information R/enc_entropy.c:13-115, range coder R/ari_codec.c:511-800.
 * Integer code; runs on lane 0 with the frame bytes in LDS. ---- */
struct BitW { uint8_t* p; int bp_side, mask_side; int bp, low, range, cache, carry, carry_count; };

__device__ __forceinline__ void put_bit_back(BitW& w, int bit)
{
    uint8_t v = w.p[w.bp_side];
    v = bit ? (uint8_t)(v | w.mask_side) : (uint8_t)(v & (255 - w.mask_side));
    w.p[w.bp_side] = v;
    if (w.mask_side == 128) { w.mask_side = 1; w.bp_side--; } else w.mask_side *= 2;
}
__device__ __forceinline__ void put_uint_back(BitW& w, int val, int nbits) { for (int k = 0; k < nbits; k++) { put_bit_back(w, val & 1); val = val / 2; } }
__device__ __forceinline__ void ac_shift(BitW& w)
{
    if (w.low < 16711680 || w.carry == 1) {
        if (w.cache >= 0) { w.p[w.bp] = (uint8_t)(w.cache + w.carry); w.bp++; }
        while (w.carry_count > 0) { w.p[w.bp] = (uint8_t)((w.carry + 255) & 255); w.bp++; w.carry_count--; }
        w.cache = w.low >> 16; w.carry = 0;
    } else w.carry_count++;
    w.low = (w.low << 8) & 0xFFFFFF;
}
__device__ __forceinline__ void ac_encode(BitW& w, int freq, int cum)
{
    int r = w.range >> 10;
    w.low += r * cum;
    if ((w.low >> 24) == 1) w.carry = 1;
    w.low &= 0xFFFFFF;
    w.range = r * freq;
    while (w.range < 65536) { w.range <<= 8; ac_shift(w); }
}
__device__ void ac_finish(BitW& w)
{
    int bits = 24 - flog2f_int((unsigned)w.range);
    int mask = 0xFFFFFF >> bits, val = w.low + mask, over1 = val >> 24;
    val &= 0xFFFFFF;
    int high = w.low + w.range, over2 = high >> 24;
    high &= 0xFFFFFF;
    val &= (0xFFFFFF - mask);
    if (over1 == over2) {
        if (val + mask >= high) { bits++; mask >>= 1; val = ((w.low + mask) & 0xFFFFFF) & (0xFFFFFF - mask); }
        if (val < w.low) w.carry = 1;
    }
    w.low = val;
    int b = bits;
    if (bits > 8) { for (; b >= 1; b -= 8) ac_shift(w); } else ac_shift(w);
    bits = b; if (bits < 0) bits += 8;
    int last; const int nb = bits;
    if (w.carry_count > 0) {
        w.p[w.bp++] = (uint8_t)w.cache;
        for (int c = w.carry_count; c >= 2; c--) w.p[w.bp++] = 255;
        last = 255 << (bits - 8);
    } else last = w.cache;
    uint8_t v = w.p[w.bp];
    for (int k = 0, m = 128; k < nb; k++, m >>= 1) { if ((last & m) == 0) v &= (uint8_t)(255 - m); else v |= (uint8_t)m; }
    w.p[w.bp] = v;
}

__device__ void st_bitstream(const lc3d_plan* __restrict__ P, const lc3d_chan& C, WaveLds& L, int bw_idx, int lastnz, int lsbMode, int gg,
                             int fac_ns, int nres, int* dbg_bp_side, int* dbg_mask_side)
{
    const int* isc = L.isc;
    BitW w; w.p = L.bytes;
    w.bp_side = C.nbytes - 1; w.mask_side = 1;
    const int nfilt = isc[I_TNS_NF];
    {   /* side information */
        const int gain_msb_bits[4] = {1, 1, 2, 2}, gain_lsb_bits[4] = {0, 1, 0, 1};
        if (P->bw_bits > 0) put_uint_back(w, bw_idx, P->bw_bits);
        put_uint_back(w, lastnz / 2 - 1, ilog2((unsigned)(P->ylen / 2 - 1)) + 1);       /* ceil(log2(ylen/2)) */
        put_bit_back(w, lsbMode);
        put_uint_back(w, gg, 8);
        for (int i = 0; i < nfilt; i++) put_bit_back(w, imin(1, isc[I_TNS_ORD0 + i]));
        put_bit_back(w, isc[I_LTPF0]);
        put_uint_back(w, isc[I_SCF0], 5); put_uint_back(w, isc[I_SCF1], 5);
        const int s2 = isc[I_SCF2], s3 = isc[I_SCF3];
        const int sub_msb = s2 / 2, sub_lsb = s2 & 1;
        put_bit_back(w, sub_msb);
        const int g_msb = s3 >> gain_lsb_bits[s2], g_lsb = s3 & 1;
        put_uint_back(w, g_msb, gain_msb_bits[s2]);
        put_bit_back(w, isc[I_SCF4]);
        if (sub_msb == 0) {
            int t = sub_lsb == 0 ? isc[I_SCF6] + 2 : g_lsb;
            t = t * 2390004 + isc[I_SCF5];
            put_uint_back(w, t, 25);
        } else {
            int t = isc[I_SCF5];
            if (sub_lsb != 0) t = 2 * t + g_lsb + 15158272;
            put_uint_back(w, t, 24);
        }
        if (isc[I_LTPF0] == 1) { put_uint_back(w, isc[I_LTPF1], 1); put_uint_back(w, isc[I_LTPF2], 9); }
        put_uint_back(w, fac_ns, 3);
    }
    if (dbg_bp_side) { *dbg_bp_side = w.bp_side; *dbg_mask_side = w.mask_side; }
    /* range coder */
    w.bp = 0; w.low = 0; w.range = 0xFFFFFF; w.cache = -1; w.carry = 0; w.carry_count = 0;
    for (int i = 0; i < nfilt; i++) {
        const int ord = isc[I_TNS_ORD0 + i];
        if (ord > 0) {
            const uint16_t* oc = &lc3t_tns_order_cum[C.lpc_weighting * 9];
            ac_encode(w, oc[ord] - oc[ord - 1], oc[ord - 1]);
            for (int j = 0; j < ord; j++) {
                const uint16_t* cc = &lc3t_tns_coef_cum[j * 18]; const int id = isc[I_TNS_IDX0 + i * 8 + j];
                ac_encode(w, cc[id + 1] - cc[id], cc[id]);
            }
        }
    }
    uint8_t* lsbs = L.res + 0;   /* lsbMode==1: residual bits are not produced, reuse the buffer bit-packed from byte 0 */
    int nl = 0, lsb1 = 0, lsb2 = 0;
    for (int k = 0; k < lastnz; k += 2) {
        const uint32_t cdv = L.cd[k >> 1];
        const int ctx = cdv & 1023, maxlev = (int)((cdv >> 10) & 63) - 1;
        const int x0 = L.xq[k], x1 = L.xq[k + 1];
        const int a0 = x0 < 0 ? -x0 : x0, b0 = x1 < 0 ? -x1 : x1;
        for (int lev = 0; lev < maxlev; lev++) {
            const int pki = lc3t_ac_ctx_lut[ctx + imin(lev, 3) * 1024];
            const uint16_t* cfp = &lc3t_ac_cum[pki * 18];
            ac_encode(w, cfp[17] - cfp[16], cfp[16]);
            const int b1 = (a0 >> lev) & 1, b2 = (b0 >> lev) & 1;
            if (lsbMode == 1 && lev == 0) { lsb1 = b1; lsb2 = b2; }
            else { put_bit_back(w, b1); put_bit_back(w, b2); }
        }
        const uint32_t cfv = L.cf[k >> 1];
        ac_encode(w, (int)(cfv >> 16), (int)(cfv & 0xffff));
        int a = a0, b = b0;
        if (lsbMode == 1 && maxlev > 0) {
#define PUSH_LSB(bitv) do { int bb = (bitv); if (bb) lsbs[nl >> 3] |= (uint8_t)(1 << (nl & 7)); nl++; } while (0)
            a >>= 1; PUSH_LSB(lsb1);
            if (a == 0 && x0 != 0) PUSH_LSB(x0 < 0);
            b >>= 1; PUSH_LSB(lsb2);
            if (b == 0 && x1 != 0) PUSH_LSB(x1 < 0);
#undef PUSH_LSB
        }
        if (a != 0) put_bit_back(w, x0 < 0);
        if (b != 0) put_bit_back(w, x1 < 0);
    }
    const int total = C.total_bits;
    const int nbits_side = total - (8 * (w.bp_side + 1) + 8 - ilog2((unsigned)w.mask_side));
    int nbits_ari = (w.bp + 1) * 8 + 25 - flog2f_int((unsigned)w.range);
    if (w.cache >= 0) nbits_ari += 8;
    if (w.carry_count > 0) nbits_ari += w.carry_count * 8;
    int nres_enc = total - (nbits_side + nbits_ari);
    nres_enc = imin(nres_enc, lsbMode == 0 ? nres : nl);
    for (int k = 0; k < nres_enc; k++) put_bit_back(w, (L.res[k >> 3] >> (k & 7)) & 1);
    ac_finish(w);
}

/* ------------------------------------------------------------------------------------------------ */
/* the kernel: one wave per channel-stream, frames in time order  (frame driver R/enc_lc3_fl.c:13-160) */
/* ------------------------------------------------------------------------------------------------ */
extern "C" __global__ void __launch_bounds__(WAVE)
lc3_encode_kernel(const lc3d_plan* __restrict__ P, const lc3d_chan* __restrict__ chans, float* __restrict__ state,
                  const void* __restrict__ pcm, int bitdepth, int T, uint8_t* __restrict__ out, int out_stride, int ncs,
                  lc3d_trace* __restrict__ trace)
{
    __shared__ WaveLds L;
    const int lane = threadIdx.x;
    const int cs = blockIdx.x;
    if (cs >= ncs) return;
    const lc3d_chan C = chans[cs];
    const int N = P->N, channels = P->channels;
    const int strm = cs / channels, ch = cs - strm * channels;

    /* ---- load cross-frame state ---- */
    float* stp = state + (size_t)cs * LC3D_STATE_WORDS;
    for (int i = lane; i < MAXN; i += WAVE) L.xbuf[i] = stp[LC3D_ST_XPREV + i];
    for (int i = lane; i < 384; i += WAVE) L.h12[i] = stp[LC3D_ST_H12 + i];
    for (int i = lane; i < 194; i += WAVE) L.h6[i] = stp[LC3D_ST_H6 + i];
    Scal S;
    {
        const float* sc = stp + LC3D_ST_SCAL; const int* si = (const int*)sc;
        S.hp0 = sc[LC3D_S_HP0]; S.hp1 = sc[LC3D_S_HP1]; S.olpa_pitch = si[LC3D_S_OLPA_PITCH];
        S.ltpf_nc1 = sc[LC3D_S_LTPF_NC1]; S.ltpf_nc2 = sc[LC3D_S_LTPF_NC2]; S.ltpf_pitch = sc[LC3D_S_LTPF_PITCH]; S.ltpf_on = si[LC3D_S_LTPF_ON];
        S.att_m0 = sc[LC3D_S_ATT_M0]; S.att_m1 = sc[LC3D_S_ATT_M1]; S.att_acc = sc[LC3D_S_ATT_ACC];
        S.att_pos = si[LC3D_S_ATT_POS]; S.att_flag = si[LC3D_S_ATT_FLAG];
        S.tbits_off = sc[LC3D_S_TBITS_OFF]; S.mem_target = si[LC3D_S_MEM_TARGET]; S.mem_spec = si[LC3D_S_MEM_SPEC];
        if (C.reset_attack) { S.att_m0 = S.att_m1 = S.att_acc = 0; S.att_pos = 0; S.att_flag = 0; }
    }
    LSYNC();

    for (int t = 0; t < T; t++) {
        lc3d_trace* tr = trace ? &trace[(size_t)cs * T + t] : nullptr;
        /* ---- PCM in (R/enc_lc3_fl.c:30-42) ---- */
        const size_t fidx = ((size_t)strm * T + t) * channels + ch;
        if (bitdepth == 16) {
            const int16_t* p = (const int16_t*)pcm + fidx * N;
            for (int i = lane; i < N; i += WAVE) L.xbuf[MAXN + i] = (float)p[i];
        } else {
            const int32_t* p = (const int32_t*)pcm + fidx * N;
            const float sc = bitdepth == 24 ? 256.0f : 65536.0f;
            for (int i = lane; i < N; i += WAVE) L.xbuf[MAXN + i] = (float)p[i] / sc;
        }
        for (int i = lane; i < 104; i += WAVE) ((uint32_t*)L.bytes)[i] = 0;
        LSYNC();

        st_mdct(P, L, lane);
        if (tr) for (int i = lane; i < N; i += WAVE) tr->spec_mdct[i] = L.spec[i];
        st_resample(P, L, S, lane);
        if (tr) for (int i = lane; i < P->len12 + 1; i += WAVE) tr->s12k8[i] = L.h12[384 - P->len12 - 24 + i];
        int T0; float nc;
        st_olpa(P, L, S, lane, T0, nc);
        int ltpf[3], ltpf_bits;
        st_ltpf(P, C, L, S, lane, T0, nc, ltpf, ltpf_bits);
        st_attack(P, C, L, S, lane);
        int bw = st_energy_bw(P, L, lane);
        if (tr) { if (lane == 0) { tr->T0 = T0; tr->normcorr = nc; tr->ltpf_param[0] = ltpf[0]; tr->ltpf_param[1] = ltpf[1]; tr->ltpf_param[2] = ltpf[2];
                                   tr->ltpf_bits = ltpf_bits; tr->attack = S.att_flag; }
                  tr->ener[lane] = lane < P->nbands ? L.sm[SM_ENER + lane] : 0; }
        LSYNC();
        st_sns_scf(P, L, lane, S.att_flag);
        if (tr && lane < 16) tr->scf[lane] = L.sm[SM_SCF + lane];
        st_sns_vq(P, L, lane);
        st_sns_apply(P, L, lane);
        if (tr) { if (lane < 16) tr->scf_q[lane] = L.sm[SM_SCFQ + lane]; if (lane < 7) tr->scf_idx[lane] = L.isc[I_SCF0 + lane];
                  for (int i = lane; i < N; i += WAVE) tr->spec_shaped[i] = L.spec[i]; }
        if (C.bandwidth) {                                  /* R/cutoff_bandwidth.c:13-26 */
            const int bin = C.bw_cut_bin;
            if (P->ylen > bin) {
                if (lane < 4) { const float sc4[4] = {0.5f, 0.25f, 0.125f, 0.0625f}; L.spec[bin - 1 + lane] = L.spec[bin - 1 + lane] * sc4[lane]; }
                for (int i = bin + 3 + lane; i < P->ylen; i += WAVE) L.spec[i] = 0;
            }
            bw = imin(bw, C.bw_index);
            LSYNC();
        }
        const int bw_bin = lc3t_bw_bins[P->bw_cls * 6 + bw];
        if (lane < 16) L.isc[I_TNS_IDX0 + lane] = 0;
        if (lane < 2) L.isc[I_TNS_ORD0 + lane] = 0;
        if (lane == 0) { L.isc[I_LTPF0] = ltpf[0]; L.isc[I_LTPF1] = ltpf[1]; L.isc[I_LTPF2] = ltpf[2]; }
        LSYNC();
        st_tns(P, C, L, lane, bw, bw_bin);
        const int tns_bits = L.isc[I_TNS_BITS];
        if (tr) { if (lane == 0) { tr->bw_idx = bw; tr->tns_nfilt = L.isc[I_TNS_NF]; tr->tns_order[0] = L.isc[I_TNS_ORD0]; tr->tns_order[1] = L.isc[I_TNS_ORD1]; tr->tns_bits = tns_bits; }
                  if (lane < 16) tr->tns_rc_idx[lane] = L.isc[I_TNS_IDX0 + lane];
                  for (int i = lane; i < N; i += WAVE) tr->spec_tns[i] = L.spec[i]; }
        const int tbq = C.target_bits_init - (tns_bits + ltpf_bits);
        float gain; int gg, ggmin, nbits, nbits2, lastnz, lsb, change;
        st_gain_estimate(P, C, L, S, lane, tbq, gain, gg, ggmin);
        if (tr && lane == 0) { tr->target_bits_quant = tbq; tr->gain0 = gain; tr->gg_idx0 = gg; tr->gg_min = ggmin; }
        st_quantize(P, C, L, lane, gain, -1, tbq, nbits, nbits2, lastnz, lsb);
        S.mem_spec = nbits;
        if (tr && lane == 0) tr->nbits0 = nbits;
        st_gain_adjust(P, C, gg, ggmin, gain, tbq, nbits, change);
        if (change) st_quantize(P, C, L, lane, gain, 0, tbq, nbits, nbits2, lastnz, lsb);
        const int fac_ns = st_noise_factor(P, C, L, lane, gain, bw_bin);
        int nres = 0;
        if (lsb == 0) nres = st_residual(P, L, lane, gain, tbq, nbits2);
        else { for (int i = lane; i < 160; i += WAVE) ((uint32_t*)L.res)[i] = 0; }
        LSYNC();
        if (tr) { if (lane == 0) { tr->gain = gain; tr->gg_idx = gg; tr->gain_change = change; tr->nbits = nbits; tr->nbits2 = nbits2; tr->lastnz = lastnz;
                                   tr->lsb_mode = lsb; tr->fac_ns = fac_ns; tr->n_res_bits = nres; }
                  for (int i = lane; i < N; i += WAVE) tr->xq[i] = i < P->ylen ? L.xq[i] : 0; }
        if (lane == 0) st_bitstream(P, C, L, bw, lastnz, lsb, gg, fac_ns, nres, tr ? &tr->bp_side : nullptr, tr ? &tr->mask_side : nullptr);
        LSYNC();
        /* ---- bytes out ---- */
        uint8_t* o = out + ((size_t)strm * T + t) * out_stride + C.out_off;
        for (int i = lane; i < C.nbytes; i += WAVE) o[i] = L.bytes[i];
        /* ---- slide the input history ---- */
        float keep[8];
#pragma unroll
        for (int k = 0; k < 8; k++) { int i = lane + 64 * k; keep[k] = i < N ? L.xbuf[MAXN + i] : 0.0f; }
        LSYNC();
#pragma unroll
        for (int k = 0; k < 8; k++) { int i = lane + 64 * k; if (i < N) L.xbuf[MAXN - N + i] = keep[k]; }
        LSYNC();
    }

    /* ---- store cross-frame state ---- */
    for (int i = lane; i < MAXN; i += WAVE) stp[LC3D_ST_XPREV + i] = L.xbuf[i];
    for (int i = lane; i < 384; i += WAVE) stp[LC3D_ST_H12 + i] = L.h12[i];
    for (int i = lane; i < 194; i += WAVE) stp[LC3D_ST_H6 + i] = L.h6[i];
    if (lane == 0) {
        float* sc = stp + LC3D_ST_SCAL; int* si = (int*)sc;
        sc[LC3D_S_HP0] = S.hp0; sc[LC3D_S_HP1] = S.hp1; si[LC3D_S_OLPA_PITCH] = S.olpa_pitch;
        sc[LC3D_S_LTPF_NC1] = S.ltpf_nc1; sc[LC3D_S_LTPF_NC2] = S.ltpf_nc2; sc[LC3D_S_LTPF_PITCH] = S.ltpf_pitch; si[LC3D_S_LTPF_ON] = S.ltpf_on;
        sc[LC3D_S_ATT_M0] = S.att_m0; sc[LC3D_S_ATT_M1] = S.att_m1; sc[LC3D_S_ATT_ACC] = S.att_acc;
        si[LC3D_S_ATT_POS] = S.att_pos; si[LC3D_S_ATT_FLAG] = S.att_flag;
        sc[LC3D_S_TBITS_OFF] = S.tbits_off; si[LC3D_S_MEM_TARGET] = S.mem_target; si[LC3D_S_MEM_SPEC] = S.mem_spec;
    }
}

/* ------------------------------------------------------------------------------------------------ */
/* C-ABI device shim (lc3_shim.h): context, uploads, launch                                          */
/* ------------------------------------------------------------------------------------------------ */
struct lc3hip_ctx {
    int device, ncs, n_streams, channels, N;
    lc3d_plan* d_plan; lc3d_chan* d_chans; float* d_state;
    void* d_pcm; size_t pcm_cap; uint8_t* d_out; size_t out_cap;
    lc3d_trace* d_trace; size_t trace_cap;
    hipStream_t stream; hipEvent_t ev0, ev1; float last_ms;
};

#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "lc3plus_hip: %s failed: %s (%s:%d)\n", #x, hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

extern "C" int lc3hip_create(void** out_ctx, const lc3d_plan* plan, int n_streams, int device)
{
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { fprintf(stderr, "lc3plus_hip: no HIP device available (this engine has no CPU fallback)\n"); return 1; }
    lc3hip_ctx* c = (lc3hip_ctx*)calloc(1, sizeof *c);
    if (!c) return 1;
    if (device < 0) { if (hipGetDevice(&device) != hipSuccess) device = 0; }
    c->device = device;
    HIPCHK(hipSetDevice(device));
    c->n_streams = n_streams; c->channels = plan->channels; c->ncs = n_streams * plan->channels; c->N = plan->N;
    HIPCHK(hipMalloc((void**)&c->d_plan, sizeof(lc3d_plan)));
    HIPCHK(hipMemcpy(c->d_plan, plan, sizeof(lc3d_plan), hipMemcpyHostToDevice));
    HIPCHK(hipMalloc((void**)&c->d_chans, sizeof(lc3d_chan) * c->ncs));
    HIPCHK(hipMalloc((void**)&c->d_state, sizeof(float) * LC3D_STATE_WORDS * (size_t)c->ncs));
    HIPCHK(hipStreamCreate(&c->stream));
    HIPCHK(hipEventCreate(&c->ev0)); HIPCHK(hipEventCreate(&c->ev1));
    *out_ctx = c;
    return 0;
}

extern "C" int lc3hip_reset_state(void* ctx, const float* init_state_one /* LC3D_STATE_WORDS floats */)
{
    lc3hip_ctx* c = (lc3hip_ctx*)ctx;
    HIPCHK(hipSetDevice(c->device));
    float* h = (float*)malloc(sizeof(float) * LC3D_STATE_WORDS * (size_t)c->ncs);
    if (!h) return 1;
    for (int i = 0; i < c->ncs; i++) memcpy(h + (size_t)i * LC3D_STATE_WORDS, init_state_one, sizeof(float) * LC3D_STATE_WORDS);
    hipError_t e = hipMemcpy(c->d_state, h, sizeof(float) * LC3D_STATE_WORDS * (size_t)c->ncs, hipMemcpyHostToDevice);
    free(h);
    HIPCHK(e);
    return 0;
}

extern "C" int lc3hip_upload_chans(void* ctx, const lc3d_chan* chans, int first, int count)
{
    lc3hip_ctx* c = (lc3hip_ctx*)ctx;
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipMemcpy(c->d_chans + first, chans, sizeof(lc3d_chan) * count, hipMemcpyHostToDevice));
    return 0;
}

extern "C" int lc3hip_encode(void* ctx, const void* pcm, int pcm_on_device, int bitdepth, int n_frames, void* out, int out_stride,
                             int out_on_device, void* hip_stream, int sync, void* trace_host)
{
    lc3hip_ctx* c = (lc3hip_ctx*)ctx;
    HIPCHK(hipSetDevice(c->device));
    hipStream_t s = hip_stream ? (hipStream_t)hip_stream : c->stream;
    const size_t bps = bitdepth == 16 ? 2 : 4;
    const size_t pcm_bytes = (size_t)c->n_streams * n_frames * c->channels * c->N * bps;
    const size_t out_bytes = (size_t)c->n_streams * n_frames * out_stride;
    const void* dpcm = pcm; uint8_t* dout = (uint8_t*)out;
    if (!pcm_on_device) {
        if (c->pcm_cap < pcm_bytes) { if (c->d_pcm) HIPCHK(hipFree(c->d_pcm)); HIPCHK(hipMalloc(&c->d_pcm, pcm_bytes)); c->pcm_cap = pcm_bytes; }
        HIPCHK(hipMemcpyAsync(c->d_pcm, pcm, pcm_bytes, hipMemcpyHostToDevice, s));
        dpcm = c->d_pcm;
    }
    if (!out_on_device) {
        if (c->out_cap < out_bytes) { if (c->d_out) HIPCHK(hipFree(c->d_out)); HIPCHK(hipMalloc((void**)&c->d_out, out_bytes)); c->out_cap = out_bytes; }
        dout = c->d_out;
        HIPCHK(hipMemsetAsync(dout, 0, out_bytes, s));
    }
    lc3d_trace* dtr = nullptr;
    if (trace_host) {
        const size_t tb = sizeof(lc3d_trace) * (size_t)c->ncs * n_frames;
        if (c->trace_cap < tb) { if (c->d_trace) HIPCHK(hipFree(c->d_trace)); HIPCHK(hipMalloc((void**)&c->d_trace, tb)); c->trace_cap = tb; }
        HIPCHK(hipMemsetAsync(c->d_trace, 0, tb, s));
        dtr = c->d_trace;
    }
    HIPCHK(hipEventRecord(c->ev0, s));
    hipLaunchKernelGGL(lc3_encode_kernel, dim3(c->ncs), dim3(WAVE), 0, s, c->d_plan, c->d_chans, c->d_state, dpcm, bitdepth, n_frames,
                       dout, out_stride, c->ncs, dtr);
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(c->ev1, s));
    if (!out_on_device) HIPCHK(hipMemcpyAsync(out, dout, out_bytes, hipMemcpyDeviceToHost, s));
    if (trace_host) HIPCHK(hipMemcpyAsync(trace_host, dtr, sizeof(lc3d_trace) * (size_t)c->ncs * n_frames, hipMemcpyDeviceToHost, s));
    if (sync || !out_on_device || trace_host) {
        HIPCHK(hipStreamSynchronize(s));
        float ms = 0; if (hipEventElapsedTime(&ms, c->ev0, c->ev1) == hipSuccess) c->last_ms = ms;
    }
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
    hipStreamSynchronize(c->stream);
    if (c->d_plan) hipFree(c->d_plan);
    if (c->d_chans) hipFree(c->d_chans);
    if (c->d_state) hipFree(c->d_state);
    if (c->d_pcm) hipFree(c->d_pcm);
    if (c->d_out) hipFree(c->d_out);
    if (c->d_trace) hipFree(c->d_trace);
    hipEventDestroy(c->ev0); hipEventDestroy(c->ev1);
    hipStreamDestroy(c->stream);
    free(c);
    return 0;
}
