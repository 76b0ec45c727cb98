// Probe (developer tool, GPU box; VERDICT r5 "Next round" item 1, step A): what does a kernel on the OTHER half of the chip cost the
// backward-data launches -- its power, or its traffic? Three synthetic neighbours, each one workgroup per CU (150 KB of LDS reserved: no
// second workgroup, of this or any conv kernel, fits beside it), built into probes/libprobe_corun.so and driven by tools/corun_split.py:
//   corun_burn    register-only v_mfma_f32_16x16x32_bf16 on random operands, no memory: power without traffic. `sleep` > 0 puts
//                 an s_sleep behind every round of 48 MFMAs (duty cycle of the matrix pipe); `nread` > 0 adds nread x 8 ds_read_b128 per
//                 round (3 = the fragment reads of a conv's stage)
//   corun_stream  every workgroup streams its own slice of a large buffer through L2 (16-byte loads, optionally stored back), no MFMA:
//                 traffic without power. `sleep` throttles it
//   corun_clock   eight idle workgroups that note s_memtime over s_memrealtime (the shader clock of whatever CU they land on)
// Every workgroup of the first two writes {s_memtime delta, s_memrealtime delta} so the host gets the clock it ran at.
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o probes/libprobe_corun.so probes/probe_corun.hip
#include <hip/hip_runtime.h>

typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

#define RESERVE_LDS (150 * 1024)

__device__ __forceinline__ void mfma16(f32x4& acc, const bf16x8& a, const bf16x8& b) {
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
}

__global__ void __launch_bounds__(512) k_burn(const bf16x8* __restrict__ src, float* __restrict__ out, unsigned long long* __restrict__ clk,
                                              int rounds, int sleep, int nread) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __attribute__((address_space(3))) char* lds = (__attribute__((address_space(3))) char*)smem;
    for (int i = threadIdx.x; i < 8192; i += 512) ((__attribute__((address_space(3))) bf16x8*)lds)[i] = src[i & 4095];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    bf16x8 fa[4], fb[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        fa[i] = src[(threadIdx.x * 8 + i) & 4095];
        fb[i] = src[(threadIdx.x * 8 + 4 + i) & 4095];
    }
    f32x4 acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < rounds; ++it) {
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) mfma16(acc[i], fa[i & 3], fb[i >> 2]);
        if (nread > 0) {   // nread x 8 fragment reads (conflict-free 1-KiB rows), the operands of the next round
            for (int r = 0; r < nread; ++r) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    fa[i] = *(const __attribute__((address_space(3))) bf16x8*)(lds + (((it + r) & 7) * 8 + i) * 1024 + lane * 16 + (wave & 1) * 65536);
                    fb[i] = *(const __attribute__((address_space(3))) bf16x8*)(lds + (((it + r) & 7) * 8 + 4 + i) * 1024 + lane * 16 + (wave & 1) * 65536);
                }
            }
        }
        if (sleep > 0)
            for (int s = 0; s < sleep; ++s) __builtin_amdgcn_s_sleep(8);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    f32x4 s = acc[0];
#pragma unroll
    for (int i = 1; i < 16; ++i) s += acc[i];
    if (s[0] == 12345.678f) out[threadIdx.x] = s[1] + s[2] + s[3];   // (keeps the accumulators live; never true in practice)
    if (threadIdx.x == 0) {
        clk[blockIdx.x * 2] = t1 - t0;
        clk[blockIdx.x * 2 + 1] = r1 - r0;
    }
}

// mode 0: read only, 1: copy. n16: 16-byte pieces per workgroup per pass; the workgroup walks its own slice `passes` times, each pass
// over a fresh slice (slice index = pass * gridDim + block) so nothing is re-read from L2 unless the host makes the buffer small.
__global__ void __launch_bounds__(512) k_stream(const u32x4* __restrict__ src, u32x4* __restrict__ dst, unsigned long long* __restrict__ clk,
                                                long n16, int passes, long nslices, int mode, int sleep) {
    extern __shared__ char smem[];
    u32x4 x = {0u, 0u, 0u, 0u};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int p = 0; p < passes; ++p) {
        const long slice = ((long)p * gridDim.x + blockIdx.x) % nslices;
        const u32x4* s = src + slice * n16;
        u32x4* d = dst + slice * n16;
        for (long i = threadIdx.x; i < n16; i += 512 * 4) {
            u32x4 v[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = (i + q * 512 < n16) ? __builtin_nontemporal_load(s + i + q * 512) : u32x4{0u, 0u, 0u, 0u};
            if (mode == 1) {
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (i + q * 512 < n16) __builtin_nontemporal_store(v[q], d + i + q * 512);
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q) x ^= v[q];
            }
            if (sleep > 0)
                for (int sl = 0; sl < sleep; ++sl) __builtin_amdgcn_s_sleep(4);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if ((x[0] ^ x[1] ^ x[2] ^ x[3]) == 0x12345677u) dst[0] = x;
    if (threadIdx.x == 0) {
        clk[blockIdx.x * 2] = t1 - t0;
        clk[blockIdx.x * 2 + 1] = r1 - r0;
    }
    if (smem[threadIdx.x] == 77 && passes < 0) dst[1] = x;
}

__global__ void __launch_bounds__(64) k_clock(unsigned long long* __restrict__ clk, unsigned long long ticks) {
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long r1 = r0;
    while (r1 - r0 < ticks) {
        __builtin_amdgcn_s_sleep(32);
        r1 = __builtin_amdgcn_s_memrealtime();
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) {
        clk[blockIdx.x * 2] = t1 - t0;
        clk[blockIdx.x * 2 + 1] = r1 - r0;
    }
}

// where does a workgroup run? {XCC_ID register, HW_ID register} per workgroup; every workgroup spins `ticks` of s_memrealtime (100 MHz) so that
// the workgroups of one launch are resident together instead of re-using one CU. `lds` bytes of LDS keep a second workgroup off the CU.
__global__ void __launch_bounds__(64) k_whereami(unsigned* __restrict__ out, unsigned long long ticks) {
    extern __shared__ char smem[];
    unsigned xcc, hwid;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - r0 < ticks) __builtin_amdgcn_s_sleep(16);
    if (threadIdx.x == 0) {
        out[blockIdx.x * 2] = xcc;
        out[blockIdx.x * 2 + 1] = hwid;
    }
    if (smem[threadIdx.x] == 77 && ticks == 0) out[0] = 1;
}

static bool g_attr_done = false;
static int set_attrs() {
    if (g_attr_done) return 0;
    if (hipFuncSetAttribute((const void*)k_burn, hipFuncAttributeMaxDynamicSharedMemorySize, RESERVE_LDS) != hipSuccess) return -1;
    if (hipFuncSetAttribute((const void*)k_stream, hipFuncAttributeMaxDynamicSharedMemorySize, RESERVE_LDS) != hipSuccess) return -1;
    g_attr_done = true;
    return 0;
}

extern "C" int corun_burn(const void* src, float* out, unsigned long long* clk, int nwg, int rounds, int sleep, int nread, void* stream) {
    if (set_attrs()) return -1;
    hipLaunchKernelGGL(k_burn, dim3(nwg), dim3(512), RESERVE_LDS, (hipStream_t)stream, (const bf16x8*)src, out, clk, rounds, sleep, nread);
    return (int)hipGetLastError();
}
extern "C" int corun_stream(const void* src, void* dst, unsigned long long* clk, int nwg, long n16, int passes, long nslices, int mode, int sleep,
                            void* stream) {
    if (set_attrs()) return -1;
    hipLaunchKernelGGL(k_stream, dim3(nwg), dim3(512), RESERVE_LDS, (hipStream_t)stream, (const u32x4*)src, (u32x4*)dst, clk, n16, passes, nslices, mode,
                       sleep);
    return (int)hipGetLastError();
}
extern "C" int corun_whereami(unsigned* out, int nwg, unsigned long long ticks, int lds, void* stream) {
    if (hipFuncSetAttribute((const void*)k_whereami, hipFuncAttributeMaxDynamicSharedMemorySize, RESERVE_LDS) != hipSuccess) return -1;
    hipLaunchKernelGGL(k_whereami, dim3(nwg), dim3(64), lds, (hipStream_t)stream, out, ticks);
    return (int)hipGetLastError();
}
extern "C" int corun_clock(unsigned long long* clk, int nwg, unsigned long long ticks, void* stream) {
    hipLaunchKernelGGL(k_clock, dim3(nwg), dim3(64), 0, (hipStream_t)stream, clk, ticks);
    return (int)hipGetLastError();
}
