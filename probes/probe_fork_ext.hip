// probe (round 5): what a dependency between two HIP streams of one device costs the RECORDING queue, by mechanism.
// Main stream: NK spin kernels (~30 us each) back to back; behind every second one the side stream is released for one spin kernel.
//   0 no dependency            1 hipEventRecord (no timing, no system fence) + hipStreamWaitEvent      (what unet._Side does)
//   2 the event rides on the producing kernel's own dispatch packet: hipExtLaunchKernelGGL(..., stopEvent) + hipStreamWaitEvent
//   3 (round 6) stream memory operations: hipStreamWriteValue32 on the main stream + hipStreamWaitValue32 on the side stream (signal memory)
//   4 (round 6) no packets at all: the producing kernel's last workgroup stores a sequence number, the consuming kernel's workgroups spin on it
//     (what an in-kernel dependency would cost the two queues: the bound of profiles/r06/fork_bound.txt, measured in isolation)
// build: hipcc --offload-arch=gfx950 -O2 probes/probe_fork_ext.hip -o probes/probe_fork_ext ; run on the GPU box
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>
#include <vector>
__global__ void spin(unsigned long long cycles, unsigned* sink) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();   // 100 MHz
    while (__builtin_amdgcn_s_memrealtime() - t0 < cycles) {}
    if (sink && threadIdx.x == 9999) *sink = 1;
}
// mode 4: flag[0] = sequence number of the last finished producer, flag[1] = arrival counter
__global__ void spin_sig(unsigned long long cycles, unsigned* flag, unsigned seq) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < cycles) {}
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0 && atomicAdd(flag + 1, 1u) == gridDim.x - 1) {
        flag[1] = 0;
        __threadfence();
        atomicExch(flag, seq);
    }
}
__global__ void spin_wait(unsigned long long cycles, const unsigned* flag, unsigned seq) {
    if (threadIdx.x == 0)
        while ((int)(__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - seq) < 0) __builtin_amdgcn_s_sleep(4);
    __syncthreads();
    __threadfence();
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < cycles) {}
}
#define CK(x) do { hipError_t err__ = (x); if (err__ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(err__)); return 1; } } while (0)
int main() {
    const int NK = 40, REP = 20;
    hipStream_t ms, ss;
    CK(hipStreamCreateWithFlags(&ms, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&ss, hipStreamNonBlocking));
    std::vector<hipEvent_t> ev(64);
    for (auto& evx : ev) CK(hipEventCreateWithFlags(&evx, hipEventDisableTiming | 0x20000000));
    hipEvent_t t0, t1;
    CK(hipEventCreate(&t0)); CK(hipEventCreate(&t1));
    const unsigned long long ticks = 3000;   // 30 us
    unsigned* sigmem = nullptr;
    const bool have_sig = hipExtMallocWithFlags((void**)&sigmem, 64, hipMallocSignalMemory) == hipSuccess;
    unsigned* flag = nullptr;
    CK(hipMalloc((void**)&flag, 64));
    CK(hipMemset(flag, 0, 64));
    if (have_sig) CK(hipMemset(sigmem, 0, 64));
    unsigned seq = 0;
    for (int mode = 0; mode < 5; ++mode) {
        if (mode == 3 && !have_sig) { printf("mode 3: no signal memory\n"); continue; }
        float best = 1e30f, sum = 0.f;
        for (int rep = 0; rep < REP + 2; ++rep) {
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(t0, ms));
            int ei = 0;
            for (int k = 0; k < NK; ++k) {
                const bool fork = (k & 1) == 1;
                if (mode == 2 && fork) {
                    hipExtLaunchKernelGGL(spin, dim3(128), dim3(64), 0, ms, nullptr, ev[ei], 0, ticks, (unsigned*)nullptr);
                } else if (mode == 4 && fork) {
                    hipLaunchKernelGGL(spin_sig, dim3(128), dim3(64), 0, ms, ticks, flag, ++seq);
                } else {
                    hipLaunchKernelGGL(spin, dim3(128), dim3(64), 0, ms, ticks, (unsigned*)nullptr);
                }
                if (fork) {
                    if (mode == 1) CK(hipEventRecord(ev[ei], ms));
                    if (mode == 1 || mode == 2) CK(hipStreamWaitEvent(ss, ev[ei], 0));
                    if (mode == 3) {
                        ++seq;
                        CK(hipStreamWriteValue32(ms, sigmem, seq, 0));
                        CK(hipStreamWaitValue32(ss, sigmem, seq, hipStreamWaitValueGte, 0xffffffffu));
                    }
                    ei = (ei + 1) % 64;
                    if (mode == 4) hipLaunchKernelGGL(spin_wait, dim3(128), dim3(64), 0, ss, ticks, flag, seq);
                    else hipLaunchKernelGGL(spin, dim3(128), dim3(64), 0, ss, ticks, (unsigned*)nullptr);
                }
            }
            hipEvent_t j = ev[63];
            CK(hipEventRecord(j, ss));
            CK(hipStreamWaitEvent(ms, j, 0));
            CK(hipEventRecord(t1, ms));
            CK(hipEventSynchronize(t1));
            float msf = 0.f;
            CK(hipEventElapsedTime(&msf, t0, t1));
            if (rep >= 2) { sum += msf; if (msf < best) best = msf; }
        }
        printf("mode %d: %8.1f us per pass (best %8.1f) = %5.2f us per main-stream kernel beyond its 30 us\n", mode, sum / REP * 1e3f, best * 1e3f,
               (sum / REP * 1e3f - NK * 30.f) / NK);
    }
    return 0;
}
