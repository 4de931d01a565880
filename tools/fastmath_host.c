/* tools/fastmath_host.c -- TEST INFRASTRUCTURE: audio_codec_amd/csrc/lc3_fastmath.h compiled for the host as array functions (tests/test_fastmath.py,
 * tests/test_gpu_parity.py::test_device_fastmath_equals_host): kind 0 = log2, 1 = log10, 2 = 2^x through the header, 3 / 4 / 5 the same through glibc.
 *   gcc -O2 -ffp-contract=off -mfma -shared -fPIC -Iaudio_codec_amd/csrc tools/fastmath_host.c -o <out>.so -lm */
#include "lc3_fastmath.h"
void lc3m_host_eval(int kind, const float* x, float* y, long n)
{
    for (long i = 0; i < n; i++) {
        switch (kind) {
        case 0: y[i] = lc3m_log2f(x[i], lc3m_log2_tab); break;
        case 1: y[i] = lc3m_log10f(x[i], lc3m_log10_tab); break;
        case 2: y[i] = lc3m_exp2f(x[i], lc3m_exp2_tab); break;
        case 3: y[i] = (float)log2((double)x[i]); break;
        case 4: y[i] = (float)log10((double)x[i]); break;
        default: y[i] = (float)pow(2.0, (double)x[i]); break;
        }
    }
}
