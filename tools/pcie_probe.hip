// tools/pcie_probe.hip -- development probe: host<->device copy rates for pageable, registered and pinned memory.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
    const size_t N = (size_t)1 << 30;
    char *pageable = (char *)malloc(N); memset(pageable, 1, N);
    char *pinned; hipHostMalloc((void **)&pinned, N, hipHostMallocDefault); memset(pinned, 2, N);
    void *d; hipMalloc(&d, N);
    hipStream_t s; hipStreamCreate(&s);
    for (int rep = 0; rep < 2; rep++) {
        double t = now(); hipMemcpy(d, pageable, N, hipMemcpyHostToDevice); printf("H2D pageable  %.1f GB/s\n", N / (now() - t) / 1e9);
        t = now(); hipMemcpy(pageable, d, N, hipMemcpyDeviceToHost); printf("D2H pageable  %.1f GB/s\n", N / (now() - t) / 1e9);
        t = now(); hipMemcpyAsync(d, pinned, N, hipMemcpyHostToDevice, s); hipStreamSynchronize(s); printf("H2D pinned    %.1f GB/s\n", N / (now() - t) / 1e9);
        t = now(); hipMemcpyAsync(pinned, d, N, hipMemcpyDeviceToHost, s); hipStreamSynchronize(s); printf("D2H pinned    %.1f GB/s\n", N / (now() - t) / 1e9);
        t = now(); hipError_t e = hipHostRegister(pageable, N, hipHostRegisterDefault); double tr = now() - t;
        printf("register 1 GiB: %.1f ms (%s)\n", tr * 1e3, hipGetErrorString(e));
        t = now(); hipMemcpyAsync(d, pageable, N, hipMemcpyHostToDevice, s); hipStreamSynchronize(s); printf("H2D registered %.1f GB/s\n", N / (now() - t) / 1e9);
        t = now(); hipHostUnregister(pageable); printf("unregister: %.1f ms\n", (now() - t) * 1e3);
        for (int nt : {1, 2, 4, 8}) {
            t = now();
            std::vector<std::thread> th;
            for (int i = 0; i < nt; i++) th.emplace_back([=] { memcpy(pinned + N / nt * i, pageable + N / nt * i, N / nt); });
            for (auto &x : th) x.join();
            printf("memcpy pageable->pinned %d threads %.1f GB/s\n", nt, N / (now() - t) / 1e9);
        }
    }
    return 0;
}
