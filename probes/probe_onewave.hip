// Probe (developer tool, GPU box; round 6, DESIGN.md section 8 "open" (2)): what would the OTHER main-loop shape sustain -- ONE wave per SIMD on
// v_mfma_f32_32x32x16_bf16 with a 128 x 128 wave tile (256 accumulator registers in AGPRs), the fragment reads of the next k-step (8 ds_read_b128 per
// 16 MFMAs) and the LDS-DMA pieces (ND per 16 MFMAs) issued in the gaps BETWEEN the wave's own MFMAs, instead of igemm_pp's ping-pong between two
// waves per SIMD on 16x16x32 with 64 x 64 wave tiles (probe_mfma_rate.hip modes 6-9)? Synthetic: random operands, L2-hot DMA source, no epilogue.
//   hipcc --offload-arch=gfx950 -O3 -o probes/probe_onewave probes/probe_onewave.hip && probes/probe_onewave
// Prints shader cycles per 16-MFMA step (ideal 16 x 32 = 512), the share of the matrix pipe's slots used, wall-clock TFLOP/s on all 256 CUs and the
// clock the counter ran at. Compare with profiles/r06/r_interval_order.txt (ping-pong: 67-73 % of the slots, ~1.5 PFLOP/s wall).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

__device__ __forceinline__ void mfma32(f32x16& acc, const bf16x8& a, const bf16x8& b) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
}
template <int OFF>
__device__ __forceinline__ void rd(bf16x8& dst, unsigned addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF) : "memory");
}

// NRD: ds_read_b128 per 16 MFMAs (8 = the fragments of a 128 x 128 wave tile's next k-step), ND: LDS-DMA pieces per 16 MFMAs and wave, NS: scalar
// instructions per MFMA gap
template <int NRD, int ND, int NS>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) k_onewave(const bf16x8* src, float* out, unsigned long long* cyc, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __attribute__((address_space(3))) char* lds = (__attribute__((address_space(3))) char*)smem;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) ((__attribute__((address_space(3))) bf16x8*)lds)[i] = src[i & 4095];
    __syncthreads();
    bf16x8 fa[2][4], fb[2][4];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            fa[h][i] = src[(threadIdx.x * 8 + i + 16 * h) & 4095];
            fb[h][i] = src[(threadIdx.x * 8 + 4 + i + 16 * h) & 4095];
        }
    f32x16 acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    const unsigned ldsa = (unsigned)(lane * 16 + wave * 8192);
    int sc = wave;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {   // two k-steps per trip: fragments double-buffered, buffer h multiplies while buffer 1 - h is read
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                mfma32(acc[i], fa[h][i & 3], fb[h][i >> 2]);
                if constexpr (NRD > 0) {
                    if ((i * NRD) / 16 != ((i + 1) * NRD) / 16) {   // NRD reads spread evenly over the 16 gaps
                        const int r = (i * NRD) / 16;
                        if (r < 4) {
                            if (r == 0) rd<0>(fa[1 - h][0], ldsa); else if (r == 1) rd<1024>(fa[1 - h][1], ldsa);
                            else if (r == 2) rd<2048>(fa[1 - h][2], ldsa); else rd<3072>(fa[1 - h][3], ldsa);
                        } else {
                            if (r == 4) rd<4096>(fb[1 - h][0], ldsa); else if (r == 5) rd<5120>(fb[1 - h][1], ldsa);
                            else if (r == 6) rd<6144>(fb[1 - h][2], ldsa); else rd<7168>(fb[1 - h][3], ldsa);
                        }
                    }
                }
                if constexpr (ND > 0) {
                    if ((i * ND) / 16 != ((i + 1) * ND) / 16) {
                        const int d = (i * ND) / 16;
                        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (((it * 2 + h) * ND + d) * 64 + lane) % 4096),
                                                         (__attribute__((address_space(3))) void*)(lds + 65536 + (wave * 8 + d) * 1024), 16, 0, 0);
                        asm volatile("" ::: "memory");
                    }
                }
#pragma unroll
                for (int s = 0; s < NS; ++s) asm volatile("s_add_u32 %0, %0, 3" : "+s"(sc)::"scc");
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the next k-step's fragments
            if constexpr (ND > 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(ND) : "memory");
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) asm volatile("" ::"a"(acc[i]));   // (all accumulators live to the end, none copied to vector registers for a sum)
#pragma unroll
    for (int j = 0; j < 16; ++j) s += acc[0][j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s + (float)sc;
    if (lane == 0) cyc[blockIdx.x * 4 + wave] = t1 - t0;
}

template <int NRD, int ND, int NS>
static void run(const char* name, int blocks, const bf16x8* src, float* out, unsigned long long* cyc) {
    hipFuncSetAttribute((const void*)k_onewave<NRD, ND, NS>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    const int iters = 400, long_iters = 20000;
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL((k_onewave<NRD, ND, NS>), dim3(blocks), dim3(256), 150 * 1024, 0, src, out, cyc, iters);
        hipDeviceSynchronize();
    }
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k_onewave<NRD, ND, NS>), dim3(blocks), dim3(256), 150 * 1024, 0, src, out, cyc, long_iters);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(blocks * 4);
    hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    const double flops = (double)blocks * 4 * long_iters * 32 * 32768.0;
    const double per_step = (double)h[0] / (long_iters * 2.0);
    printf("%-74s blocks %3d  cycles per 16 MFMAs %7.1f (ideal 512)  pipe %5.1f %% | wall %7.1f TFLOP/s, counter %.2f GHz\n", name, blocks, per_step,
           100.0 * 512.0 / per_step, flops / (ms * 1e-3) / 1e12, (double)h[0] / (ms * 1e-3) / 1e9);
    fflush(stdout);
}

int main() {
    bf16x8* src;
    float* out;
    unsigned long long* cyc;
    hipMalloc(&src, 4096 * 16);
    hipMalloc(&out, 256 * 256 * 4);
    hipMalloc(&cyc, 256 * 4 * 8);
    std::vector<unsigned short> h(4096 * 8);
    for (auto& v : h) v = (unsigned short)(0x3c00 + (rand() & 0x3ff)) ^ ((rand() & 1) << 15);
    hipMemcpy(src, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    for (int blocks : {1, 256}) {
        run<0, 0, 0>("one wave per SIMD, 32x32x16, 128 x 128 wave tile: MFMAs only", blocks, src, out, cyc);
        run<8, 0, 0>("... + 8 ds_read_b128 per 16 MFMAs (the next k-step's fragments)", blocks, src, out, cyc);
        run<8, 0, 1>("... + 1 scalar instruction per gap", blocks, src, out, cyc);
        run<8, 2, 1>("... + 2 LDS-DMA pieces per 16 MFMAs and wave", blocks, src, out, cyc);
        run<8, 4, 1>("... + 4 LDS-DMA pieces per 16 MFMAs and wave (a 256 x 256 workgroup tile's own traffic)", blocks, src, out, cyc);
        run<8, 4, 3>("... + 3 scalar instructions per gap", blocks, src, out, cyc);
    }
    return 0;
}
