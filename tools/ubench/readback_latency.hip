// Host read-back of a few device integers after a kernel: hipMemcpyAsync + hipStreamSynchronize against a kernel that
// writes them to coherent pinned host memory and a host that polls a sequence word. Build + run:
//   hipcc --offload-arch=gfx950 -O2 tools/ubench/readback_latency.hip -o tools/ubench/readback_latency && tools/ubench/readback_latency
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void k_work(int* s, int v) { if (threadIdx.x == 0) s[0] = v; }
__global__ void k_publish(const int* s, int n, volatile int* host, int seq) {
    if ((int)threadIdx.x < n) host[threadIdx.x] = s[threadIdx.x];
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) host[63] = seq;
}
int main() {
    int *d, *h, *hp;
    CK(hipMalloc(&d, 256));
    CK(hipHostMalloc(&h, 256, hipHostMallocDefault));
    CK(hipHostMalloc(&hp, 256, hipHostMallocCoherent | hipHostMallocMapped));
    hp[63] = 0;
    hipStream_t st;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    const int K = 2000;
    for (int mode = 0; mode < 4; mode++) {
        auto t0 = std::chrono::steady_clock::now();
        for (int i = 1; i <= K; i++) {
            k_work<<<1, 64, 0, st>>>(d, i);
            if (mode == 0) {
                int v;
                CK(hipMemcpyAsync(&v, d, 4, hipMemcpyDeviceToHost, st));
                CK(hipStreamSynchronize(st));
                if (v != i) return 2;
            } else if (mode == 1) {
                CK(hipMemcpyAsync(h, d, 57 * 4, hipMemcpyDeviceToHost, st));
                CK(hipStreamSynchronize(st));
                if (h[0] != i) return 2;
            } else if (mode == 2) {
                k_publish<<<1, 64, 0, st>>>(d, 57, hp, i);
                while (__atomic_load_n(&hp[63], __ATOMIC_ACQUIRE) != i) __builtin_ia32_pause();
                if (hp[0] != i) return 2;
            } else {
                CK(hipStreamSynchronize(st));
            }
        }
        const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / K;
        const char* names[] = {"memcpyAsync 4 B to pageable + sync", "memcpyAsync 228 B to pinned + sync", "publish kernel + host poll", "stream sync only"};
        printf("%-40s %7.1f us per kernel+readback\n", names[mode], us);
    }
    return 0;
}
