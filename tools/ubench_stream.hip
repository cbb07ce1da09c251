// What a streaming kernel shaped like the quotient's pointwise kernel can pull from HBM on this box: K input arrays of
// N 32-byte elements + one output, one element per thread, with the load patterns under test.
//   mode 0: lane l reads its element as two 16-byte loads 32 bytes apart (the kernel's pattern: every load instruction
//           of a wavefront touches 2 KiB and uses half of it)
//   mode 1: every load instruction of a wavefront covers 1 KiB contiguously (lane l: bytes [16 l, 16 l + 16) of the
//           first / second KiB of the wavefront's 2 KiB) -- the data a lane gets are NOT its element: timing only
//   mode 2: as 0 with non-temporal loads and stores
//   mode 3: as 0, two elements per thread (i and i + N/2)
// build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench_stream tools/ubench_stream.hip ; run: tools/ubench_stream
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>

#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

struct Args { const uint4* in[16]; uint4* out; uint64_t n; int k; };

template <int MODE>
__global__ __launch_bounds__(256) void stream_kernel(Args a) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    const uint64_t n = MODE == 3 ? a.n / 2 : a.n;
    if (i >= n) return;
    uint4 s0 = make_uint4(0, 0, 0, 0), s1 = s0;
    const uint64_t wave_base = (i & ~63ull) * 2;         // in uint4 units
    const uint32_t lane = threadIdx.x & 63;
#pragma unroll 4
    for (int j = 0; j < a.k; ++j) {
        uint4 x, y;
        if (MODE == 1) {
            x = a.in[j][wave_base + lane];
            y = a.in[j][wave_base + 64 + lane];
        } else if (MODE == 2) {
            typedef uint32_t v4 __attribute__((ext_vector_type(4)));
            const v4 vx = __builtin_nontemporal_load((const v4*)(a.in[j] + 2 * i));
            const v4 vy = __builtin_nontemporal_load((const v4*)(a.in[j] + 2 * i + 1));
            x = make_uint4(vx.x, vx.y, vx.z, vx.w);
            y = make_uint4(vy.x, vy.y, vy.z, vy.w);
        } else {
            x = a.in[j][2 * i];
            y = a.in[j][2 * i + 1];
            if (MODE == 3) {
                const uint4 x2 = a.in[j][2 * (i + n)], y2 = a.in[j][2 * (i + n) + 1];
                x.x ^= x2.x; x.y += x2.y; y.z ^= y2.z; y.w += y2.w;
            }
        }
        s0.x ^= x.x; s0.y += x.y; s0.z ^= x.z; s0.w += x.w;
        s1.x ^= y.x; s1.y += y.y; s1.z ^= y.z; s1.w += y.w;
    }
    if (MODE == 2) {
        typedef uint32_t v4 __attribute__((ext_vector_type(4)));
        v4 t0 = {s0.x, s0.y, s0.z, s0.w}, t1 = {s1.x, s1.y, s1.z, s1.w};
        __builtin_nontemporal_store(t0, (v4*)(a.out + 2 * i));
        __builtin_nontemporal_store(t1, (v4*)(a.out + 2 * i + 1));
    } else {
        a.out[2 * i] = s0;
        a.out[2 * i + 1] = s1;
        if (MODE == 3) { a.out[2 * (i + n)] = s1; a.out[2 * (i + n) + 1] = s0; }
    }
}

int main() {
    const uint64_t n = 1ull << 22;
    Args a{};
    a.n = n;
    for (int j = 0; j < 16; ++j) { CHK(hipMalloc((void**)&a.in[j], n * 32)); CHK(hipMemset((void*)a.in[j], j + 1, n * 32)); }
    CHK(hipMalloc((void**)&a.out, n * 32));
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    for (int k : {1, 2, 4, 8, 14}) {
        a.k = k;
        for (int mode = 0; mode < 4; ++mode) {
            const unsigned blocks = (unsigned)((mode == 3 ? n / 2 : n) / 256);
            float best = 1e9f;
            for (int rep = 0; rep < 6; ++rep) {
                CHK(hipEventRecord(e0));
                for (int it = 0; it < 5; ++it) {
                    if (mode == 0) hipLaunchKernelGGL(stream_kernel<0>, dim3(blocks), dim3(256), 0, 0, a);
                    if (mode == 1) hipLaunchKernelGGL(stream_kernel<1>, dim3(blocks), dim3(256), 0, 0, a);
                    if (mode == 2) hipLaunchKernelGGL(stream_kernel<2>, dim3(blocks), dim3(256), 0, 0, a);
                    if (mode == 3) hipLaunchKernelGGL(stream_kernel<3>, dim3(blocks), dim3(256), 0, 0, a);
                }
                CHK(hipEventRecord(e1));
                CHK(hipEventSynchronize(e1));
                float ms;
                CHK(hipEventElapsedTime(&ms, e0, e1));
                if (ms / 5 < best) best = ms / 5;
            }
            printf("STREAM k=%2d mode=%d  %.4f ms  %.0f GB/s\n", k, mode, best, (k + 1) * n * 32.0 / best / 1e6);
        }
    }
    return 0;
}
