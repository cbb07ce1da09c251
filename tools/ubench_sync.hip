// Completion latency of a stream's last kernel as the host sees it: hipStreamSynchronize against polling a word the kernel writes
// into pinned host memory (system-scope store).  hipcc --offload-arch=gfx950 -O2 tools/ubench_sync.hip -o tools/ubench_sync
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdint>
__global__ void spin_then_flag(volatile uint32_t* flag, uint32_t value, long long cycles) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < cycles) {}
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        __threadfence_system();
        *flag = value;
    }
}
int main() {
    uint32_t* flag;
    hipHostMalloc((void**)&flag, 64, hipHostMallocDefault);
    *flag = 0;
    hipStream_t s;
    hipStreamCreate(&s);
    const long long cyc = 100 * 100;  // ~100 us at the 100 MHz wall clock
    for (int mode = 0; mode < 2; ++mode) {
        double tot = 0;
        const int reps = 200;
        for (int r = 1; r <= reps + 20; ++r) {
            auto t0 = std::chrono::steady_clock::now();
            hipLaunchKernelGGL(spin_then_flag, dim3(1), dim3(64), 0, s, flag, (uint32_t)(mode * 100000 + r), cyc);
            if (mode == 0) hipStreamSynchronize(s);
            else { while (*(volatile uint32_t*)flag != (uint32_t)(mode * 100000 + r)) {} }
            auto t1 = std::chrono::steady_clock::now();
            if (r > 20) tot += std::chrono::duration<double, std::micro>(t1 - t0).count();
            if (mode == 1) hipStreamSynchronize(s);
        }
        printf("%s: %.1f us per launch + 100 us kernel + completion\n", mode == 0 ? "hipStreamSynchronize" : "poll pinned flag", tot / reps);
    }
    return 0;
}
