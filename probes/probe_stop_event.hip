// Probe (GPU box): what a fork of the side stream costs the MAIN queue per kernel, by how the dependency is expressed:
//   0  no fork at all (kernels back to back)
//   1  hipEventRecord between the kernels (what _lib.hip_fork does: event without timing / system fence), side stream waits + runs a kernel
//   2  the event rides on the kernel dispatch itself: hipExtLaunchKernelGGL(..., stopEvent), side stream waits + runs a kernel
// build: hipcc -O3 --offload-arch=gfx950 probes/probe_stop_event.hip -o probes/probe_stop_event
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <vector>
__global__ void k_busy(int iters, float* sink) {
    float a = threadIdx.x;
    for (int i = 0; i < iters; ++i) a = a * 1.0001f + 0.5f;
    if (a == 12345.f) *sink = a;
}
int main() {
    hipStream_t m, s; hipStreamCreateWithFlags(&m, hipStreamNonBlocking); hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    float* sink; hipMalloc(&sink, 4);
    const int N = 200, iters = 20000;   // ~tens of us per kernel
    std::vector<hipEvent_t> ev(N);
    for (auto& e : ev) hipEventCreateWithFlags(&e, hipEventDisableTiming | 0x20000000 /* hipEventDisableSystemFence */);
    hipEvent_t t0, t1; hipEventCreate(&t0); hipEventCreate(&t1);
    for (int mode = 0; mode < 3; ++mode) {
        float best = 1e9f;
        for (int rep = 0; rep < 3; ++rep) {
            hipDeviceSynchronize();
            hipEventRecord(t0, m);
            for (int i = 0; i < N; ++i) {
                if (mode == 2) hipExtLaunchKernelGGL(k_busy, dim3(128), dim3(256), 0, m, nullptr, ev[i], 0, iters, sink);
                else hipLaunchKernelGGL(k_busy, dim3(128), dim3(256), 0, m, iters, sink);
                if (mode == 1) hipEventRecord(ev[i], m);
                if (mode >= 1) { hipStreamWaitEvent(s, ev[i], 0); hipLaunchKernelGGL(k_busy, dim3(128), dim3(256), 0, s, iters * 8 / 10, sink); }
            }
            hipEventRecord(t1, m); hipEventSynchronize(t1); hipDeviceSynchronize();
            float ms; hipEventElapsedTime(&ms, t0, t1); if (ms < best) best = ms;
        }
        printf("mode %d: %d kernels on the main stream: %.3f ms = %.2f us per kernel\n", mode, N, best, best * 1e3 / N);
    }
    return 0;
}
