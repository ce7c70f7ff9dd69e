// h2d.hip — microbenchmark: what does it cost to bring the profile text (31 MB at 100k rows, 330 MB at 1M)
// from PAGEABLE host memory (the C-ABI borrows the caller's buffer) into HBM?  Sizes the device tokeniser's
// input stage.
//   hipcc -O3 --offload-arch=gfx950 -o h2d h2d.hip -pthread && ./h2d
// Variants: (a) hipMemcpy from pageable memory; (b) hipHostRegister + hipMemcpyAsync + hipHostUnregister;
// (c) host threads copy into pinned staging chunks, each chunk goes out as soon as it is filled;
// (d) a kernel reads the registered host buffer directly (zero copy over PCIe).
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#define CK(x)                                                                      \
    do {                                                                           \
        hipError_t e_ = (x);                                                       \
        if (e_ != hipSuccess) {                                                    \
            printf("%s: %s\n", #x, hipGetErrorString(e_));                         \
            exit(1);                                                               \
        }                                                                          \
    } while (0)

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

__global__ void k_sum(const uint4 *p, size_t n16, unsigned *out) {
    unsigned acc = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) {
        uint4 v = p[i];
        acc += v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) out[0] = acc;
}

int main(int argc, char **argv) {
    const int threads = argc > 1 ? atoi(argv[1]) : 8;
    CK(hipSetDevice(0));
    hipStream_t st, st2;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&st2, hipStreamNonBlocking));
    unsigned *d_out;
    CK(hipMalloc(&d_out, 64));
    for (size_t bytes : {(size_t)31 << 20, (size_t)330 << 20}) {
        char *h = (char *)malloc(bytes + 4096);
        for (size_t i = 0; i < bytes; i++) h[i] = (char)(i * 131 + (i >> 9));
        char *d;
        CK(hipMalloc(&d, bytes));
        CK(hipMemset(d, 0, bytes));
        CK(hipDeviceSynchronize());
        printf("---- %zu MiB, %d host threads\n", bytes >> 20, threads);
        for (int rep = 0; rep < 4; rep++) {
            double t0 = now();
            CK(hipMemcpy(d, h, bytes, hipMemcpyHostToDevice));
            double t1 = now();
            printf("(a) hipMemcpy pageable            %8.3f ms  %6.1f GB/s\n", (t1 - t0) * 1e3, bytes / (t1 - t0) / 1e9);
        }
        for (int rep = 0; rep < 4; rep++) {
            double t0 = now();
            CK(hipHostRegister(h, bytes, hipHostRegisterDefault));
            double t1 = now();
            CK(hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, st));
            CK(hipStreamSynchronize(st));
            double t2 = now();
            CK(hipHostUnregister(h));
            double t3 = now();
            printf("(b) register %7.3f + copy %7.3f (%5.1f GB/s) + unregister %7.3f = %8.3f ms\n", (t1 - t0) * 1e3, (t2 - t1) * 1e3,
                   bytes / (t2 - t1) / 1e9, (t3 - t2) * 1e3, (t3 - t0) * 1e3);
        }
        for (size_t chunk : {(size_t)1 << 20, (size_t)4 << 20, (size_t)16 << 20}) {
            const int n_stage = 8;
            char *stage;
            CK(hipHostMalloc(&stage, chunk * n_stage, hipHostMallocDefault));
            hipEvent_t ev[n_stage];
            for (auto &e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
            for (int rep = 0; rep < 3; rep++) {
                double t0 = now();
                const size_t n_chunks = (bytes + chunk - 1) / chunk;
                for (size_t c = 0; c < n_chunks; c++) {
                    const int s = (int)(c % n_stage);
                    if (c >= (size_t)n_stage) CK(hipEventSynchronize(ev[s]));
                    const size_t o = c * chunk, len = std::min(chunk, bytes - o);
                    // the chunk is cut into `threads` pieces copied side by side
                    std::vector<std::thread> th;
                    const int nt = std::max(1, std::min<int>(threads, (int)(len >> 18)));
                    for (int t = 1; t < nt; t++)
                        th.emplace_back([=] { memcpy(stage + s * chunk + len * t / nt, h + o + len * t / nt, len * (t + 1) / nt - len * t / nt); });
                    memcpy(stage + s * chunk, h + o, len / nt);
                    for (auto &x : th) x.join();
                    CK(hipMemcpyAsync(d + o, stage + s * chunk, len, hipMemcpyHostToDevice, st));
                    CK(hipEventRecord(ev[s], st));
                }
                CK(hipStreamSynchronize(st));
                double t1 = now();
                printf("(c) staged, %2zu MiB chunks          %8.3f ms  %6.1f GB/s\n", chunk >> 20, (t1 - t0) * 1e3, bytes / (t1 - t0) / 1e9);
            }
            for (auto &e : ev) CK(hipEventDestroy(e));
            CK(hipHostFree(stage));
        }
        {   // persistent copier threads (no thread start per chunk): each thread owns every threads-th chunk of 1 MiB,
            // copies it into its own pinned slot pair and enqueues the DMA on its own stream
            const size_t chunk = (size_t)1 << 20;
            const int nt = threads;
            char *stage;
            CK(hipHostMalloc(&stage, chunk * 2 * nt, hipHostMallocDefault));
            std::vector<hipStream_t> sts(nt);
            for (auto &s : sts) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
            for (int rep = 0; rep < 4; rep++) {
                double t0 = now();
                const size_t n_chunks = (bytes + chunk - 1) / chunk;
                std::vector<std::thread> th;
                for (int t = 0; t < nt; t++)
                    th.emplace_back([&, t] {
                        (void)hipSetDevice(0);
                        hipEvent_t ev[2];
                        for (auto &e : ev) (void)hipEventCreateWithFlags(&e, hipEventDisableTiming);
                        int k = 0;
                        for (size_t c = t; c < n_chunks; c += nt, k++) {
                            const int s = k & 1;
                            if (k >= 2) (void)hipEventSynchronize(ev[s]);
                            const size_t o = c * chunk, len = std::min(chunk, bytes - o);
                            char *dst = stage + ((size_t)t * 2 + s) * chunk;
                            memcpy(dst, h + o, len);
                            (void)hipMemcpyAsync(d + o, dst, len, hipMemcpyHostToDevice, sts[t]);
                            (void)hipEventRecord(ev[s], sts[t]);
                        }
                        (void)hipStreamSynchronize(sts[t]);
                        for (auto &e : ev) (void)hipEventDestroy(e);
                    });
                for (auto &x : th) x.join();
                double t1 = now();
                printf("(c') %d copier threads, own streams  %8.3f ms  %6.1f GB/s\n", nt, (t1 - t0) * 1e3, bytes / (t1 - t0) / 1e9);
            }
            for (auto &s : sts) CK(hipStreamDestroy(s));
            CK(hipHostFree(stage));
        }
        for (int rep = 0; rep < 3; rep++) {
            double t0 = now();
            CK(hipHostRegister(h, bytes, hipHostRegisterMapped));
            void *dp;
            CK(hipHostGetDevicePointer(&dp, h, 0));
            double t1 = now();
            hipLaunchKernelGGL(k_sum, dim3(1024), dim3(256), 0, st, (const uint4 *)dp, bytes / 16, d_out);
            CK(hipStreamSynchronize(st));
            double t2 = now();
            CK(hipHostUnregister(h));
            double t3 = now();
            printf("(d) register %7.3f + zero-copy kernel %7.3f (%5.1f GB/s) + unregister %7.3f = %8.3f ms\n", (t1 - t0) * 1e3,
                   (t2 - t1) * 1e3, bytes / (t2 - t1) / 1e9, (t3 - t2) * 1e3, (t3 - t0) * 1e3);
        }
        {   // pinned source (what a caller that owns its buffers could hand over): the DMA rate itself
            char *p;
            CK(hipHostMalloc(&p, bytes, hipHostMallocDefault));
            memcpy(p, h, bytes);
            for (int rep = 0; rep < 3; rep++) {
                double t0 = now();
                CK(hipMemcpyAsync(d, p, bytes, hipMemcpyHostToDevice, st));
                CK(hipStreamSynchronize(st));
                double t1 = now();
                printf("(e) pinned source                  %8.3f ms  %6.1f GB/s\n", (t1 - t0) * 1e3, bytes / (t1 - t0) / 1e9);
            }
            for (int rep = 0; rep < 2; rep++) {
                double t0 = now();
                hipLaunchKernelGGL(k_sum, dim3(1024), dim3(256), 0, st, (const uint4 *)p, bytes / 16, d_out);
                CK(hipStreamSynchronize(st));
                double t1 = now();
                printf("(e') kernel reads pinned source    %8.3f ms  %6.1f GB/s\n", (t1 - t0) * 1e3, bytes / (t1 - t0) / 1e9);
            }
            CK(hipHostFree(p));
        }
        {   // host memcpy rate alone, 1 and `threads` threads
            char *q = (char *)malloc(bytes);
            memset(q, 1, bytes);
            for (int nt : {1, threads}) {
                double t0 = now();
                std::vector<std::thread> th;
                for (int t = 0; t < nt; t++) th.emplace_back([=] { memcpy(q + bytes * t / nt, h + bytes * t / nt, bytes * (t + 1) / nt - bytes * t / nt); });
                for (auto &x : th) x.join();
                double t1 = now();
                printf("(f) host memcpy, %2d thread(s)       %8.3f ms  %6.1f GB/s\n", nt, (t1 - t0) * 1e3, bytes / (t1 - t0) / 1e9);
            }
            free(q);
        }
        CK(hipFree(d));
        free(h);
    }
    return 0;
}
