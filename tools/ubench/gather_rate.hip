// gather_rate.hip — microbenchmark: how many SCATTERED 4-byte loads per second the chip answers from its L2s (and from the
// Infinity Cache behind them), on gfx950.  k_join is 4 M such loads per step at 100k rows — one bitmap word per token
// occurrence, each in a cache line of its own — plus the table probes behind the hits; its "roof" is this rate, not HBM bytes.
//   hipcc -O3 --offload-arch=gfx950 -o gather_rate gather_rate.hip && ./gather_rate
// Every lane loads words at pseudo-random indices of a table of F bytes, U loads in flight per lane (independent address
// streams), W waves per SIMD, all 256 CUs; the table is replicated by the hardware in every XCD's L2 that touches it.
// Output: G loads/s for footprints 256 KiB .. 256 MiB, U = 1, 4, 8 and 4 or 8 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CHECK(x)                                                                  \
    do {                                                                          \
        hipError_t e_ = (x);                                                      \
        if (e_ != hipSuccess) {                                                   \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));               \
            return 1;                                                             \
        }                                                                         \
    } while (0)

template <int U>
__global__ __launch_bounds__(256) void k_gather(const unsigned *__restrict__ tab, unsigned mask, int iters, unsigned *out) {
    unsigned idx[U], acc = 0;
    const unsigned tid = blockIdx.x * 256u + threadIdx.x;
#pragma unroll
    for (int u = 0; u < U; u++) idx[u] = (tid * 2654435761u + u * 0x9E3779B9u) | 1u;
    for (int i = 0; i < iters; i++) {
        unsigned v[U];
#pragma unroll
        for (int u = 0; u < U; u++) v[u] = tab[(idx[u] >> 4) & mask];  // (16 words = one 64-byte line per index step: a line per lane)
#pragma unroll
        for (int u = 0; u < U; u++) {
            acc += v[u];
            idx[u] = idx[u] * 1664525u + 1013904223u + (v[u] & 1u);  // the next address depends on the loaded word
        }
    }
    if (acc == 0x12345678u) out[tid & 1023u] = acc;
}

template <int U>
static int run(const unsigned *d_tab, size_t bytes, int waves_per_simd, unsigned *d_out) {
    const unsigned mask = (unsigned)(bytes / 4 - 1) & ~15u;  // index of a line's first word
    const int blocks = 256 * waves_per_simd;                 // 256 CUs x (4 SIMDs x waves) / 4 waves per block
    const int iters = 2048;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k_gather<U>, dim3(blocks), dim3(256), 0, 0, d_tab, mask, 64, d_out);  // warm the caches
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_gather<U>, dim3(blocks), dim3(256), 0, 0, d_tab, mask, iters, d_out);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double loads = (double)blocks * 256 * iters * U;
    printf("  footprint %8.2f MiB  U %d  %d waves/SIMD : %8.1f G loads/s  (%.3f ms)\n", bytes / 1048576.0, U, waves_per_simd, loads / ms / 1e6, ms);
    return 0;
}

int main() {
    const size_t max_bytes = (size_t)256 << 20;
    unsigned *d_tab, *d_out;
    CHECK(hipMalloc(&d_tab, max_bytes));
    CHECK(hipMalloc(&d_out, 4096));
    std::vector<unsigned> h(max_bytes / 4);
    for (size_t i = 0; i < h.size(); i++) h[i] = (unsigned)(i * 2654435761u) >> 7;
    CHECK(hipMemcpy(d_tab, h.data(), max_bytes, hipMemcpyHostToDevice));
    for (size_t bytes : {(size_t)256 << 10, (size_t)1 << 20, (size_t)2 << 20, (size_t)4 << 20, (size_t)16 << 20, (size_t)64 << 20, (size_t)256 << 20}) {
        for (int w : {4, 8}) {
            if (run<1>(d_tab, bytes, w, d_out)) return 1;
            if (run<4>(d_tab, bytes, w, d_out)) return 1;
            if (run<8>(d_tab, bytes, w, d_out)) return 1;
        }
    }
    return 0;
}
