// Host <-> device copy rates for the shapes lc3hip_encode's host path uses (diagnostic; not part of the product).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <chrono>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
// strided rows host -> compact device buffer, 16 bytes per lane (zero-copy read of pinned host memory by a kernel)
__global__ void gather_rows(const uint4* __restrict__ src, size_t src_pitch16, uint4* __restrict__ dst, int w16, int rows)
{
    for (int r = blockIdx.x; r < rows; r += gridDim.x)
        for (int i = threadIdx.x; i < w16; i += blockDim.x) dst[(size_t)r * w16 + i] = src[(size_t)r * src_pitch16 + i];
}
int main()
{
    const size_t rows = 4096, T = 64, fr = 960, pitch = T * fr, total = rows * pitch;       // the c1 PCM block: 252 MB
    void *h, *d; hipHostMalloc(&h, total, hipHostMallocDefault); hipMalloc(&d, total);
    memset(h, 1, total);
    hipStream_t s; hipStreamCreate(&s);
    for (int rep = 0; rep < 2; rep++) {
        double t0 = now(); hipMemcpyAsync(d, h, total, hipMemcpyHostToDevice, s); hipStreamSynchronize(s); double t1 = now();
        printf("linear H2D %zu MB: %.2f ms = %.1f GB/s\n", total >> 20, (t1 - t0) * 1e3, total / (t1 - t0) / 1e9);
    }
    for (int K : {4, 8, 16}) {
        const size_t w = pitch / K;
        double t0 = now();
        for (int k = 0; k < K; k++) hipMemcpy2DAsync((char*)d + k * w * rows, w, (char*)h + k * w, pitch, w, rows, hipMemcpyHostToDevice, s);
        hipStreamSynchronize(s); double t1 = now();
        printf("2-D H2D in %d column blocks of %zu B rows: %.2f ms = %.1f GB/s\n", K, w, (t1 - t0) * 1e3, total / (t1 - t0) / 1e9);
    }
    for (int K : {8}) {
        const size_t w = pitch / K;
        double t0 = now();
        for (int k = 0; k < K; k++) gather_rows<<<1024, 256, 0, s>>>((const uint4*)((char*)h + k * w), pitch / 16, (uint4*)((char*)d + k * w * rows), (int)(w / 16), (int)rows);
        hipStreamSynchronize(s); double t1 = now();
        printf("kernel gather (zero-copy reads of pinned memory) in %d column blocks: %.2f ms = %.1f GB/s\n", K, (t1 - t0) * 1e3, total / (t1 - t0) / 1e9);
    }
    {   // D2H of the frames: 4096 rows x 64 x 80 B
        const size_t op = 64 * 80, ot = rows * op;
        double t0 = now(); hipMemcpyAsync(h, d, ot, hipMemcpyDeviceToHost, s); hipStreamSynchronize(s); double t1 = now();
        printf("linear D2H %zu MB: %.2f ms = %.1f GB/s\n", ot >> 20, (t1 - t0) * 1e3, ot / (t1 - t0) / 1e9);
        t0 = now();
        for (int k = 0; k < 8; k++) hipMemcpy2DAsync((char*)h + k * op / 8, op, (char*)d + k * (op / 8) * rows, op / 8, op / 8, rows, hipMemcpyDeviceToHost, s);
        hipStreamSynchronize(s); t1 = now();
        printf("2-D D2H in 8 column blocks of %zu B rows: %.2f ms = %.1f GB/s\n", op / 8, (t1 - t0) * 1e3, ot / (t1 - t0) / 1e9);
    }
    {   // pageable source for comparison
        void* p = malloc(total); memset(p, 2, total);
        double t0 = now(); hipMemcpyAsync(d, p, total, hipMemcpyHostToDevice, s); hipStreamSynchronize(s); double t1 = now();
        printf("linear H2D from pageable memory: %.2f ms = %.1f GB/s\n", (t1 - t0) * 1e3, total / (t1 - t0) / 1e9);
        t0 = now(); memcpy(h, p, total); t1 = now();
        printf("host memcpy pageable -> pinned (one thread): %.2f ms = %.1f GB/s\n", (t1 - t0) * 1e3, total / (t1 - t0) / 1e9);
    }
    return 0;
}
