// Probe (developer tool, GPU box): what a SIMD's matrix pipe sustains under the wave arrangements the conv kernels use.
//   hipcc --offload-arch=gfx950 -O3 -o probes/probe_mfma_rate probes/probe_mfma_rate.hip && probes/probe_mfma_rate
// Every variant runs ITER rounds of NM v_mfma_f32_16x16x32_bf16 (or NM/2 32x32x16) per wave on independent accumulators and
// reports shader cycles (s_memtime) per round for wave 0 of block 0, plus the ideal (16 cycles per 16x16x32 MFMA per SIMD).
//   mode 0: free-running, no barrier           mode 1: one s_barrier per round, all waves multiply together
//   mode 2: ping-pong: groups G0 = waves 0-3, G1 = waves 4-7 alternate M rounds and idle rounds (two barriers per round)
//   mode 3: ping-pong with NR ds_read_b128 in the idle round         mode 4: as 3 plus ND LDS-DMA pieces in the idle round
//   mode 6: a phase shaped like igemm_pp's: 24 live ds_read_b128 (3 taps x (4 A + 4 B) fragments), 48 MFMAs, ND LDS-DMA pieces (L2-hot
//           source) and 30 vector + 40 scalar instructions in the R round
//   mode 5: as 3 plus ND vector (v_add) and 2*ND scalar instructions in the idle round (the address arithmetic of a real kernel)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <type_traits>

typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

__device__ __forceinline__ void mfma16(f32x4& acc, const bf16x8& a, const bf16x8& b) {
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
}
__device__ __forceinline__ void mfma32(f32x16& acc, const bf16x8& a, const bf16x8& b) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
}
__device__ __forceinline__ void mfma16a(f32x4& acc, const bf16x8& a, const bf16x8& b) {
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
}

// ---- round 6: an R interval with its instruction order pinned (modes 7 / 8 below)
template <int R>
__device__ __forceinline__ void order_rd(bf16x8 (&ka)[3][4], bf16x8 (&kb)[3][4], unsigned ldsa) {
    constexpr int off = R < 12 ? R * 1024 : 12288 + (R - 12) * 1024;
    if constexpr (R < 12) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(ka[R / 4][R % 4]) : "v"(ldsa), "n"(off) : "memory");
    else asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(kb[(R - 12) / 4][R % 4]) : "v"(ldsa), "n"(off) : "memory");
}
__device__ __forceinline__ void order_alu(int& vv, int& sc) {
    asm volatile("v_add_u32 %0, %0, %1\n\tv_add_u32 %0, %0, %1\n\tv_add_u32 %0, %0, %1\n\ts_add_u32 %1, %1, 3\n\ts_xor_b32 %1, %1, 5\n\ts_add_u32 %1, %1, 3\n\ts_xor_b32 %1, %1, 5" : "+v"(vv), "+s"(sc)::"memory");
}
template <int ND>
__device__ __forceinline__ void order_dma(const bf16x8* src, __attribute__((address_space(3))) char* lds, int it, int lane, int wave, int d) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + ((it * ND + d) * 64 + lane) % 4096),
                                     (__attribute__((address_space(3))) void*)(lds + 65536 + (wave * 8 + d) * 1024), 16, 0, 0);
    asm volatile("" ::: "memory");
}
template <int MODE, int ND, int R>
__device__ __forceinline__ void order_seq(bf16x8 (&ka)[3][4], bf16x8 (&kb)[3][4], unsigned ldsa, int& vv, int& sc, const bf16x8* src,
                                          __attribute__((address_space(3))) char* lds, int it, int lane, int wave) {
    if constexpr (R < 24) {
        order_rd<R>(ka, kb, ldsa);
        if constexpr (MODE == 7) {
            if constexpr ((R & 1) == 1 && R / 2 < 10) order_alu(vv, sc);
            if constexpr ((R & 3) == 3 && R / 4 < ND) order_dma<ND>(src, lds, it, lane, wave, R / 4);
        }
        order_seq<MODE, ND, R + 1>(ka, kb, ldsa, vv, sc, src, lds, it, lane, wave);
    }
}

template <int MODE, int NM, int NR, int ND, bool BIG, bool AGPR>
__global__ void __launch_bounds__(512) probe(const bf16x8* src, float* out, unsigned long long* cyc, int iters, int nthreads_active) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __attribute__((address_space(3))) char* lds = (__attribute__((address_space(3))) char*)smem;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // operands: random-ish data from memory
    bf16x8 fa[4], fb[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        fa[i] = src[(threadIdx.x * 8 + i) & 4095];
        fb[i] = src[(threadIdx.x * 8 + 4 + i) & 4095];
    }
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) ((__attribute__((address_space(3))) bf16x8*)lds)[i & 4095] = src[i & 4095];
    __syncthreads();
    constexpr int NACC = BIG ? NM / 2 : NM;
    f32x4 acc[BIG ? 1 : (MODE >= 6 ? 16 : NM)];
    f32x16 accb[BIG ? NM / 2 : 1];
    if constexpr (!BIG) {
#pragma unroll
        for (int i = 0; i < (MODE >= 6 ? 16 : NM); ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    } else {
#pragma unroll
        for (int i = 0; i < NACC; ++i)
#pragma unroll
            for (int j = 0; j < 16; ++j) accb[i][j] = 0.f;
    }
    bf16x8 ka[3][4], kb[3][4];
    if constexpr (MODE >= 6) {
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                ka[t][i] = fa[i];
                kb[t][i] = fb[i];
            }
    }
    int mit = 0;
    auto mround = [&]() {
        if constexpr (MODE == 9) {   // round 6: the multiplying wave issues the LDS-DMA pieces itself, one behind every 48 / ND MFMAs
#pragma unroll
            for (int t = 0; t < 3; ++t)
#pragma unroll
                for (int pt = 0; pt < 4; ++pt)
#pragma unroll
                    for (int ct = 0; ct < 4; ++ct) {
                        mfma16(acc[ct * 4 + pt], ka[t][ct], kb[t][pt]);
                        constexpr int every = ND > 0 ? 48 / ND : 1000;
                        const int idx = (t * 4 + pt) * 4 + ct;
                        if (ND > 0 && idx % every == every - 1 && idx / every < ND) order_dma<ND>(src, lds, mit, lane, wave, idx / every);
                    }
            return;
        }
        if constexpr (MODE == 6 || MODE == 7 || MODE == 8) {
#pragma unroll
            for (int t = 0; t < 3; ++t)
#pragma unroll
                for (int pt = 0; pt < 4; ++pt)
#pragma unroll
                    for (int ct = 0; ct < 4; ++ct) mfma16(acc[ct * 4 + pt], ka[t][ct], kb[t][pt]);
            return;
        }
        if constexpr (!BIG) {
#pragma unroll
            for (int i = 0; i < NM; ++i) {
                if constexpr (AGPR) mfma16a(acc[i], fa[i & 3], fb[(i >> 2) & 3]);
                else mfma16(acc[i], fa[i & 3], fb[(i >> 2) & 3]);
            }
        } else {
#pragma unroll
            for (int i = 0; i < NACC; ++i) mfma32(accb[i], fa[i & 3], fb[(i >> 2) & 3]);
        }
    };
    auto bar = [&]() {
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };
    const int grp = wave >> 2;
    unsigned long long t0 = 0, t1 = 0;
    if (MODE >= 2 && grp == 1) bar();
    t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        mit = it;
        if constexpr (MODE == 0) {
            mround();
        } else if constexpr (MODE == 1) {
            mround();
            bar();
        } else {
            // idle / R round
            if constexpr (MODE >= 3) {
#pragma unroll
                for (int r = 0; r < NR; ++r) {
                    const bf16x8 v = *(const __attribute__((address_space(3))) bf16x8*)(lds + ((r * 64 + lane) * 16 + (it & 3) * 8192));
                    if (r < 4) fa[r] = v; else fb[r & 3] = v;
                }
            }
            if constexpr (MODE == 6) {
                const int base = (it & 1) * 32768;
#pragma unroll
                for (int t = 0; t < NR / 8; ++t) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) ka[t][i] = *(const __attribute__((address_space(3))) bf16x8*)(lds + base + ((t * 4 + i) * 64 + lane) * 16);
#pragma unroll
                    for (int i = 0; i < 4; ++i) kb[t][i] = *(const __attribute__((address_space(3))) bf16x8*)(lds + base + 12288 + ((t * 4 + i) * 64 + lane) * 16);
                }
                int vv = lane + it;
                int sc = __builtin_amdgcn_readfirstlane(it);
#pragma unroll
                for (int d = 0; d < (AGPR ? 0 : (NM == 48 ? 10 : NM)); ++d)
                    asm volatile("v_add_u32 %0, %0, %1\n\tv_add_u32 %0, %0, %1\n\tv_add_u32 %0, %0, %1\n\ts_add_u32 %1, %1, 3\n\ts_xor_b32 %1, %1, 5\n\ts_add_u32 %1, %1, 3\n\ts_xor_b32 %1, %1, 5" : "+v"(vv), "+s"(sc));
                asm volatile("" ::"v"(vv), "s"(sc));
#pragma unroll
                for (int d = 0; d < ND; ++d)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + ((it * ND + d) * 64 + lane) % 4096),
                                                     (__attribute__((address_space(3))) void*)(lds + 65536 + (wave * 8 + d) * 1024), 16, 0, 0);
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(ND) : "memory");
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
            if constexpr (MODE == 9) {
                const unsigned ldsa = (unsigned)((it & 1) * 32768 + lane * 16);
                int vv = lane + it;
                int sc = __builtin_amdgcn_readfirstlane(it);
                order_seq<8, 0, 0>(ka, kb, ldsa, vv, sc, src, lds, it, lane, wave);
#pragma unroll
                for (int d = 0; d < 10; ++d) order_alu(vv, sc);
                asm volatile("" ::"v"(vv), "s"(sc));
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
            if constexpr (MODE == 7 || MODE == 8) {
                // round 6: the same R interval as mode 6 (24 ds_read_b128, 10 ALU blocks of 3 vector + 4 scalar, ND LDS-DMA pieces) with the ORDER pinned by
                // inline asm: mode 8 = the order igemm_pp's intervals compile to (all reads, then the ALU, then the pieces), mode 7 = interleaved (an ALU
                // block behind every second read, a piece behind every fourth): does a wave stalled on the LDS queue lose issue time other categories could use?
                const unsigned ldsa = (unsigned)((it & 1) * 32768 + lane * 16);
                int vv = lane + it;
                int sc = __builtin_amdgcn_readfirstlane(it);
                order_seq<MODE, ND, 0>(ka, kb, ldsa, vv, sc, src, lds, it, lane, wave);
                if constexpr (MODE == 8) {
#pragma unroll
                    for (int d = 0; d < 10; ++d) order_alu(vv, sc);
#pragma unroll
                    for (int d = 0; d < ND; ++d) order_dma<ND>(src, lds, it, lane, wave, d);
                }
                asm volatile("" ::"v"(vv), "s"(sc));
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(ND) : "memory");
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
            if constexpr (MODE == 5) {
                int vv = lane + it;
                int sc = __builtin_amdgcn_readfirstlane(it);
#pragma unroll
                for (int d = 0; d < ND; ++d) asm volatile("v_add_u32 %0, %0, %1\n\ts_add_u32 %1, %1, 3\n\ts_xor_b32 %1, %1, 5" : "+v"(vv), "+s"(sc));
                asm volatile("" ::"v"(vv), "s"(sc));
            }
            if constexpr (MODE == 4) {
#pragma unroll
                for (int d = 0; d < ND; ++d)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + ((it * ND + d) * 64 + lane) % 4096),
                                                     (__attribute__((address_space(3))) void*)(lds + 65536 + (wave * ND + d) * 1024), 16, 0, 0);
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(ND) : "memory");
            }
            bar();
            __builtin_amdgcn_s_setprio(1);
            mround();
            __builtin_amdgcn_s_setprio(0);
            bar();
        }
    }
    t1 = __builtin_amdgcn_s_memtime();
    if (MODE >= 2 && grp == 0) bar();
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    float s = 0.f;
    if constexpr (!BIG) {
#pragma unroll
        for (int i = 0; i < (MODE >= 6 ? 16 : NM); ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    } else {
#pragma unroll
        for (int i = 0; i < NACC; ++i)
#pragma unroll
            for (int j = 0; j < 16; ++j) s += accb[i][j];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int MODE, int NM, int NR, int ND, bool BIG, bool AGPR>
static void run(const char* name, int threads, int blocks, const bf16x8* src, float* out, unsigned long long* cyc) {
    const int iters = 400;
    hipFuncSetAttribute((const void*)probe<MODE, NM, NR, ND, BIG, AGPR>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL((probe<MODE, NM, NR, ND, BIG, AGPR>), dim3(blocks), dim3(threads), 96 * 1024, 0, src, out, cyc, iters, threads);
        hipDeviceSynchronize();
    }
    // wall clock of a long run (all blocks): TFLOP/s by hipEvents, to read next to the cycle counter
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int long_iters = 20000;
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((probe<MODE, NM, NR, ND, BIG, AGPR>), dim3(blocks), dim3(threads), 96 * 1024, 0, src, out, cyc, long_iters, threads);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> hl(blocks * 8);
    hipMemcpy(hl.data(), cyc, hl.size() * 8, hipMemcpyDeviceToHost);
    const double flops = (double)blocks * (threads / 64) * long_iters * (MODE >= 6 ? 48 : NM) * 16384.0;
    const double tf = flops / (ms * 1e-3) / 1e12;
    const double ghz = (double)hl[0] / (ms * 1e-3) / 1e9;  // counter ticks per second over the long run
    hipLaunchKernelGGL((probe<MODE, NM, NR, ND, BIG, AGPR>), dim3(blocks), dim3(threads), 96 * 1024, 0, src, out, cyc, iters, threads);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(blocks * 8);
    hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    const int waves = threads / 64;
    // MFMA work per SIMD per round: waves/4 waves x NM MFMAs x 16 cycles; ping-pong modes run 2 rounds (G0's and G1's) per iteration
    const double per_iter = (double)h[0] / iters;
    const double ideal = MODE >= 6 ? 1536.0 : MODE >= 2 ? 2.0 * NM * 16.0 : (double)(waves > 4 ? waves / 4 : 1) * NM * 16.0;
    printf("%-52s blocks %3d  cyc/iter %8.1f  ideal %5.0f  util %5.1f %% | wall %7.1f TFLOP/s, counter %.2f GHz\n", name, blocks, per_iter, ideal,
           100.0 * ideal / per_iter, tf, ghz);
}

int main() {
    bf16x8* src;
    float* out;
    unsigned long long* cyc;
    hipMalloc(&src, 4096 * 16);
    hipMalloc(&out, 256 * 512 * 4);
    hipMalloc(&cyc, 256 * 8 * 8);
    std::vector<unsigned short> h(4096 * 8);
    for (auto& v : h) v = (unsigned short)(0x3c00 + (rand() & 0x3ff)) ^ ((rand() & 1) << 15);  // bf16 values of magnitude ~1, random sign
    hipMemcpy(src, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    if (getenv("PROBE_SHAPES")) {   // 16x16x32 against 32x32x16 beside a partner that reads / computes addresses (same FLOPs per M round)
        for (int blocks : {256}) {
            run<2, 16, 0, 0, false, false>("ping-pong, 16 x 16x16x32 per M round", 512, blocks, src, out, cyc);
            run<2, 16, 0, 0, true, false>("ping-pong, 8 x 32x32x16 per M round", 512, blocks, src, out, cyc);
            run<3, 16, 8, 0, false, false>("ping-pong, 16 x 16x16x32 | 8 ds_read_b128", 512, blocks, src, out, cyc);
            run<3, 16, 8, 0, true, false>("ping-pong, 8 x 32x32x16 | 8 ds_read_b128", 512, blocks, src, out, cyc);
            run<5, 16, 8, 12, false, false>("ping-pong, 16 x 16x16x32 | 8 ds_read + 12 VALU + 24 SALU", 512, blocks, src, out, cyc);
            run<5, 16, 8, 12, true, false>("ping-pong, 8 x 32x32x16 | 8 ds_read + 12 VALU + 24 SALU", 512, blocks, src, out, cyc);
            run<5, 16, 8, 4, false, false>("ping-pong, 16 x 16x16x32 | 8 ds_read + 4 VALU + 8 SALU", 512, blocks, src, out, cyc);
            run<5, 16, 8, 4, true, false>("ping-pong, 8 x 32x32x16 | 8 ds_read + 4 VALU + 8 SALU", 512, blocks, src, out, cyc);
            run<4, 16, 8, 2, false, false>("ping-pong, 16 x 16x16x32 | 8 ds_read + 2 LDS-DMA", 512, blocks, src, out, cyc);
            run<4, 16, 8, 2, true, false>("ping-pong, 8 x 32x32x16 | 8 ds_read + 2 LDS-DMA", 512, blocks, src, out, cyc);
            run<4, 16, 8, 4, false, false>("ping-pong, 16 x 16x16x32 | 8 ds_read + 4 LDS-DMA", 512, blocks, src, out, cyc);
            run<4, 16, 8, 4, true, false>("ping-pong, 8 x 32x32x16 | 8 ds_read + 4 LDS-DMA", 512, blocks, src, out, cyc);
        }
        return 0;
    }
    if (getenv("PROBE_ORDER")) {   // round 6: does the ORDER of an R interval's instructions matter? (profiles/r06/r_interval_order.txt)
        for (int blocks : {1, 256}) {
            run<6, 48, 24, 6, false, false>("mode 6 (compiler's order): 24 ds_read + 70 ALU + 6 LDS-DMA", 512, blocks, src, out, cyc);
            run<8, 48, 24, 6, false, false>("asm, grouped: 24 ds_read | 70 ALU | 6 LDS-DMA", 512, blocks, src, out, cyc);
            run<7, 48, 24, 6, false, false>("asm, interleaved: ALU behind every 2nd read, DMA every 4th", 512, blocks, src, out, cyc);
            run<9, 48, 24, 6, false, false>("asm, 6 LDS-DMA issued by the MULTIPLYING wave (1 per 8 MFMAs)", 512, blocks, src, out, cyc);
            run<9, 48, 24, 12, false, false>("asm, 12 LDS-DMA issued by the multiplying wave (1 per 4 MFMAs)", 512, blocks, src, out, cyc);
            run<8, 48, 24, 0, false, false>("asm, grouped, no LDS-DMA", 512, blocks, src, out, cyc);
            run<7, 48, 24, 0, false, false>("asm, interleaved, no LDS-DMA", 512, blocks, src, out, cyc);
        }
        return 0;
    }
    for (int blocks : {1, 256}) {
        run<0, 16, 0, 0, false, false>("free-running, 1 wave/SIMD, 16 x 16x16x32", 256, blocks, src, out, cyc);
        run<0, 16, 0, 0, false, true>("free-running, 1 wave/SIMD, 16 x 16x16x32, AGPR acc", 256, blocks, src, out, cyc);
        run<0, 16, 0, 0, true, false>("free-running, 1 wave/SIMD, 8 x 32x32x16", 256, blocks, src, out, cyc);
        run<0, 16, 0, 0, false, false>("free-running, 2 waves/SIMD, 16 x 16x16x32 each", 512, blocks, src, out, cyc);
        run<1, 16, 0, 0, false, false>("barrier per round, 2 waves/SIMD, 16 each", 512, blocks, src, out, cyc);
        run<1, 48, 0, 0, false, false>("barrier per round, 2 waves/SIMD, 48 each", 512, blocks, src, out, cyc);
        run<2, 8, 0, 0, false, false>("ping-pong, 8 MFMA per M round", 512, blocks, src, out, cyc);
        run<2, 16, 0, 0, false, false>("ping-pong, 16 MFMA per M round", 512, blocks, src, out, cyc);
        run<2, 32, 0, 0, false, false>("ping-pong, 32 MFMA per M round", 512, blocks, src, out, cyc);
        run<2, 16, 0, 0, true, false>("ping-pong, 8 x 32x32x16 per M round", 512, blocks, src, out, cyc);
        run<3, 16, 8, 0, false, false>("ping-pong, 16 MFMA, 8 ds_read_b128 in R", 512, blocks, src, out, cyc);
        run<3, 32, 8, 0, false, false>("ping-pong, 32 MFMA, 8 ds_read_b128 in R", 512, blocks, src, out, cyc);
        run<3, 48, 8, 0, false, false>("ping-pong, 48 MFMA, 8 ds_read_b128 in R", 512, blocks, src, out, cyc);
        run<5, 48, 8, 12, false, false>("ping-pong, 48 MFMA, 8 ds_read + 12 VALU + 24 SALU in R", 512, blocks, src, out, cyc);
        run<5, 48, 8, 40, false, false>("ping-pong, 48 MFMA, 8 ds_read + 40 VALU + 80 SALU in R", 512, blocks, src, out, cyc);
        run<5, 48, 8, 100, false, false>("ping-pong, 48 MFMA, 8 ds_read + 100 VALU + 200 SALU in R", 512, blocks, src, out, cyc);
        run<6, 48, 24, 0, false, false>("kernel-like phase: 48 MFMA | 24 ds_read + 30 VALU + 40 SALU", 512, blocks, src, out, cyc);
        run<6, 48, 24, 0, false, true>("kernel-like phase: 48 MFMA | 24 ds_read, no ALU", 512, blocks, src, out, cyc);
        run<6, 48, 16, 0, false, true>("kernel-like phase: 48 MFMA | 16 ds_read, no ALU", 512, blocks, src, out, cyc);
        run<6, 48, 8, 0, false, true>("kernel-like phase: 48 MFMA | 8 ds_read, no ALU", 512, blocks, src, out, cyc);
        run<6, 48, 8, 0, false, false>("kernel-like phase: 48 MFMA | 8 ds_read + 30 VALU + 40 SALU", 512, blocks, src, out, cyc);
        run<6, 48, 0, 0, false, false>("kernel-like phase: 48 MFMA | 0 ds_read + 30 VALU + 40 SALU", 512, blocks, src, out, cyc);
        run<6, 48, 0, 6, false, true>("kernel-like phase: 48 MFMA | 6 LDS-DMA only", 512, blocks, src, out, cyc);
        run<6, 3, 24, 4, false, false>("lean phase: 48 MFMA | 24 ds_read + 21 ALU + 4 LDS-DMA", 512, blocks, src, out, cyc);
        run<6, 3, 24, 6, false, false>("lean phase: 48 MFMA | 24 ds_read + 21 ALU + 6 LDS-DMA", 512, blocks, src, out, cyc);
        run<6, 5, 24, 4, false, false>("lean phase: 48 MFMA | 24 ds_read + 35 ALU + 4 LDS-DMA", 512, blocks, src, out, cyc);
        run<6, 3, 24, 2, false, false>("lean phase: 48 MFMA | 24 ds_read + 21 ALU + 2 LDS-DMA", 512, blocks, src, out, cyc);
        run<6, 48, 24, 3, false, false>("kernel-like phase + 3 LDS-DMA pieces per wave", 512, blocks, src, out, cyc);
        run<6, 48, 24, 6, false, false>("kernel-like phase + 6 LDS-DMA pieces per wave", 512, blocks, src, out, cyc);
        run<4, 16, 8, 2, false, false>("ping-pong, 16 MFMA, 8 ds_read + 2 LDS-DMA in R", 512, blocks, src, out, cyc);
        run<4, 32, 8, 2, false, false>("ping-pong, 32 MFMA, 8 ds_read + 2 LDS-DMA in R", 512, blocks, src, out, cyc);
        run<4, 32, 8, 4, false, false>("ping-pong, 32 MFMA, 8 ds_read + 4 LDS-DMA in R", 512, blocks, src, out, cyc);
    }
    return 0;
}
