// Host-to-device copy rate of this box from pinned memory: one stream, and several streams at once
// (the ceiling of the upload path).  hipcc --offload-arch=gfx950 -O2 -o tools/ubench_h2d tools/ubench_h2d.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
int main()
{
    const size_t total = (size_t)1200 << 20;
    void *h = nullptr, *d = nullptr;
    if (hipHostMalloc(&h, total, hipHostMallocDefault) != hipSuccess || hipMalloc(&d, total) != hipSuccess) return 1;
    for (size_t i = 0; i < total; i += 4096) ((char*)h)[i] = 1;
    for (int n_streams : {1, 2, 4, 8}) {
        std::vector<hipStream_t> st(n_streams);
        for (auto& s : st) hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
        for (size_t chunk : {(size_t)4 << 20, (size_t)32 << 20, total / n_streams}) {
            double best = 1e9;
            for (int rep = 0; rep < 3; ++rep) {
                hipDeviceSynchronize();
                auto t0 = std::chrono::steady_clock::now();
                size_t k = 0;
                for (size_t off = 0; off < total; off += chunk, ++k)
                    hipMemcpyAsync((char*)d + off, (char*)h + off, std::min(chunk, total - off), hipMemcpyHostToDevice, st[k % n_streams]);
                hipDeviceSynchronize();
                best = std::min(best, std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
            }
            printf("%d streams, %zu MiB chunks: %.1f GB/s\n", n_streams, chunk >> 20, total / best / 1e9);
        }
        for (auto& s : st) hipStreamDestroy(s);
    }
    return 0;
}
