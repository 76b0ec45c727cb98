// igemm_ct: the 2x2 stride-2 transposed convolution (forward and backward-data) as a ping-pong GEMM.
//
// Both directions are plain matrix products over the pixels m = (n, y, x) of the LOW-resolution grid:
//   forward        y[n][2y+a][2x+b][co] = bias[co] + sum_ci x[m][ci] * K[a][b][co][ci]
//                  = for each output row phase a: C[m][n''] with n'' = b * Cout + co, 2*Cout CONTIGUOUS output elements per pixel
//   backward-data  dx[m][ci] = mask * sum_{a,b,co} dy[n][2y+a][2x+b][co] * K[a][b][co][ci]   (K = 4 taps x Cout, gathered)
// igemm_fwd2 ran them as 1-tap / 4-tap convolutions with one 32-channel k-step per pipeline stage: 16 MFMAs per wave between two
// rounds of stage bookkeeping, 350 TFLOP/s. Here a stage is 64 channels (two k-steps: 16 fragment reads, 32 MFMAs per wave), the
// eight waves work as the two ping-pong groups of igemm_pp.hip (G0 = waves 0-3 multiply while G1 = waves 4-7 read and prefetch,
// and the other way round, one workgroup barrier per interval), pixels are tiled along the FLAT index m (256 per workgroup tile,
// no padding to strips: the levels are small), and all address arithmetic that does not depend on the lane is scalar.
//
// Workgroup tile 256 pixels x 128 output columns (wave tile 64 x 64: acc[4][4]); packed weights, fragment layouts, the swizzled
// [pixel][64 bytes] LDS image of a 32-channel plane and the channel-permuted 16-byte stores are igemm_fwd2's.
// LDS: ring of 3 stage slots, each [A plane 0][A plane 1][W plane 0][W plane 1] = 2 x 16 KiB + 2 x 8 KiB. In R(s) every wave
// reads its fragments of stage s and issues its six pieces (4 of A, 2 of W; plane = its group) of stage s+2 into the slot of stage
// s-1; the pieces of stage s+1 are waited for (counted vmcnt) at the end of G1's R(s) / behind G0's MFMAs of M(s).
#include <type_traits>

#include "igemm.h"

#define RSU_SENT 0x80000000u

namespace {
__device__ __forceinline__ void bdma16c(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned soff, void* lds_wave_base) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds_wave_base, 16, voff, soff, 0, 0);
}
}  // namespace

// MODE 0 = forward (scatter store, bias), 1 = backward-data (gathered taps, ReLU mask)
template <int MODE>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) igemm_ct_kernel(const IgCtParams p) {
    constexpr int NW = 8, CT = 4, PT = 4, TM = 256, TN = 128, WPX = 4;
    constexpr int APL = TM * 64, WPL = (TN / 16) * 1024;   // one 32-channel plane of the A tile / of the W tile
    constexpr int SLOT = 2 * APL + 2 * WPL, NSLOT = 3;
    constexpr int NST = (CT / 2) * PT;                     // epilogue stores per wave per tile
    constexpr int NPW = 6;                                 // DMA pieces per wave per stage
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __attribute__((address_space(3))) char* lds = (__attribute__((address_space(3))) char*)smem;
    constexpr int bias_base = NSLOT * SLOT;

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int grp = wave >> 2, w4 = wave & 3;
    const int wco = wave / WPX, wpx = wave % WPX;
    const int g4 = lane >> 4, l15 = lane & 15, lq = lane >> 2;
    auto sgpr = [](auto v) { return __builtin_amdgcn_readfirstlane(v); };

    int vid = xcd_contiguous_id(blockIdx.x, gridDim.x);
    const int cob = vid % p.ncob;                 // column block (forward: row phase a = cob / nnb, block of 128 columns cob % nnb)
    const int tile0 = vid / p.ncob, tstride = gridDim.x / p.ncob;
    const int M = p.N * p.H * p.W;
    const int ntile_m = (M + TM - 1) / TM;
    const int my_tiles = tile0 < ntile_m ? (ntile_m - tile0 + tstride - 1) / tstride : 0;
    if (my_tiles == 0) return;
    const int pha = MODE == 0 ? cob / p.nnb : 0, nb = MODE == 0 ? cob - pha * p.nnb : cob;
    const int ncols = MODE == 0 ? 2 * p.Cn : p.Cn;   // columns of C this launch produces per row phase
    const int NS = MODE == 0 ? (p.nchunk + 1) / 2 : 4 * ((p.nchunk + 1) / 2);   // stages per tile
    const int GS = my_tiles * NS;

    // ---- per-lane constants
    int boff[PT];    // byte offset of this lane's 16-byte fragment piece inside an A plane, per pixel fragment
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) {
        const int hp = (wpx * PT + pt) * 16 + l15;
        boff[pt] = (hp << 6) + ((g4 ^ ((hp >> 1) & 2)) << 4);
    }
    const int afrag = 2 * APL + (wco * CT * 64 + lane) * 16;   // this lane's 16 bytes inside weight tile 0 of the wave (plane 0, slot 0)
    auto mk = [&](const void* ptr) { return __builtin_amdgcn_make_buffer_rsrc((void*)ptr, 0, 0x7fffffff, 0x00020000); };

    // ---- prefetch stream. A wave's six pieces of a stage: A pieces w4*4 .. w4*4+3 (16 pixels x 64 bytes each) and the two weight
    // tiles of 32-column group w4 of plane `grp` (k-step grp of the stage).
    // weights: column group G = nb * 4 + w4 of the launch; forward: (b, local group) = divmod(G * 32, Cout) inside phase (a, b)
    const int wG = nb * 4 + w4;
    const bool wvalid = wG * 32 < ncols;
    unsigned w_base;   // byte offset of the wave's first weight tile inside a (chunk[, tap]) block
    if (MODE == 0) {
        const int b = (wG * 32) / p.Cn, lg = (wG * 32 - b * p.Cn) >> 5;
        w_base = sgpr((unsigned)(((long)(pha * 2 + b) * p.wp_phase_stride) * 2 + (long)lg * 2048));
    } else {
        w_base = sgpr((unsigned)(wG * 2048));
    }
    const unsigned w_blk = sgpr((unsigned)(p.ntiles_w * 1024));   // one (chunk[, tap]) block of the packed weights
    const unsigned w_lane = wvalid ? (unsigned)(lane * 16) : RSU_SENT;
    // A: per-lane part of the source offset (pixel lq of a piece, 16-byte chunk (lane & 3) ^ swizzle); forward: pixels are rows of
    // x[M][Ca]; backward-data: pixel (n, y, x) -> dy[n][2y][2x], recomputed per tile
    const int kg8 = ((lane & 3) ^ ((lq >> 1) & 2)) * 8;
    unsigned a_lane[MODE == 0 ? 1 : 4];
    if (MODE == 0) a_lane[0] = (unsigned)((lq * p.Ca + kg8) * 2);
    const int HW = p.H * p.W;
    auto pix_decode = [&](int m, int& n, int& y, int& x) {   // m < 2^24: the multiply-high quotient is at most one short
        n = (int)__umulhi((unsigned)m, p.magic_hw);
        int r = m - n * HW;
        if (r >= HW) { ++n; r -= HW; }
        y = (int)__umulhi((unsigned)r, p.magic_w);
        x = r - y * p.W;
        if (x >= p.W) { ++y; x -= p.W; }
    };
    int pf_tile = 0;          // tile (index in this workgroup's list) the prefetch stream is in
    int pf_s = 0;             // its next stage
    int pf_slot = 0;
    auto pf_m0 = [&]() { return sgpr((tile0 + pf_tile * tstride) * TM); };
    auto pf_setup_tile = [&]() {
        if (MODE == 1) {
            const int m0 = pf_m0();
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int m = m0 + (w4 * 4 + q) * 16 + lq;
                int n, y, x;
                pix_decode(m < M ? m : 0, n, y, x);
                a_lane[q] = m < M ? (unsigned)((((n * 2 * p.H + 2 * y) * 2 * p.W + 2 * x) * p.Ca + kg8) * 2) : RSU_SENT;
            }
        }
    };
    auto pf_issue = [&]() {
        // stage pf_s of tile pf_tile -> slot pf_slot; this wave: plane grp
        const int st = MODE == 0 ? pf_s : (pf_s >> 2) ;   // backward-data: stage = (chunk pair, tap), taps innermost
        const int tap = MODE == 0 ? 0 : (pf_s & 3);
        const int chunk = 2 * st + grp;
        const bool cvalid = chunk < p.nchunk;             // (odd chunk count: the last stage's second plane reads as zeros)
        const int crem = p.Ca - chunk * 32;               // channels left from this chunk on
        const __amdgpu_buffer_rsrc_t ra = mk(p.a), rw = mk(p.wp);
        const int dstA = pf_slot * SLOT + grp * APL, dstW = pf_slot * SLOT + 2 * APL + grp * WPL;
        const int m0 = pf_m0();
        if (MODE == 0) {
            const unsigned soff = sgpr((unsigned)(((long)(m0 + w4 * 64) * p.Ca + chunk * 32) * 2));
            const unsigned pstep = (unsigned)(16 * p.Ca * 2);
            const bool full = cvalid && crem >= 32 && m0 + TM <= M;   // wave-uniform
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                unsigned vo = a_lane[0];
                if (!full) vo = (cvalid && kg8 < crem && m0 + (w4 * 4 + q) * 16 + lq < M) ? a_lane[0] : RSU_SENT;
                bdma16c(ra, vo, soff + q * pstep, (void*)(lds + dstA + (w4 * 4 + q) * 1024));
            }
        } else {
            const int ta = tap >> 1, tb = tap & 1;
            const unsigned soff = sgpr((unsigned)(((ta * 2 * p.W + tb) * p.Ca + chunk * 32) * 2));
            const bool full = cvalid && crem >= 32;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                unsigned vo = a_lane[q];
                if (!full) vo = (cvalid && kg8 < crem) ? a_lane[q] : RSU_SENT;
                bdma16c(ra, vo, soff, (void*)(lds + dstA + (w4 * 4 + q) * 1024));
            }
        }
        {
            const unsigned soff = sgpr(w_base + (unsigned)(MODE == 0 ? chunk : chunk * 4 + tap) * w_blk);
            const unsigned vo = cvalid ? w_lane : RSU_SENT;
            bdma16c(rw, vo, soff, (void*)(lds + dstW + (w4 * 2) * 1024));
            bdma16c(rw, vo == RSU_SENT ? RSU_SENT : vo + 1024, soff, (void*)(lds + dstW + (w4 * 2 + 1) * 1024));
        }
        // advance; behind the last tile the stream stages that tile again (valid memory, slots nobody reads): the counted waits hold
        pf_slot = pf_slot == NSLOT - 1 ? 0 : pf_slot + 1;
        if (++pf_s == NS) {
            pf_s = 0;
            if (pf_tile + 1 < my_tiles) ++pf_tile;
            pf_setup_tile();
        }
    };

    // ---- epilogue
    if (threadIdx.x < TN) {
        const int col = nb * TN + threadIdx.x;   // forward: column n'' = b * Cout + co
        float bv = 0.f;
        if (MODE == 0 && p.bias && col < ncols) bv = p.bias[col >= p.Cn ? col - p.Cn : col];
        *(__attribute__((address_space(3))) float*)(lds + bias_base + threadIdx.x * 4) = bv;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#define CT_STORE(R, VOFF) \
    asm volatile("s_nop 4\n\tbuffer_store_dwordx4 %0, %1, %2, %3 offen\n\ts_nop 3" ::"v"(R), "v"(VOFF), "s"(orsrc), "s"(sbase) : "memory")
    // output offsets of this lane's PT pixels: backward-data = rows of dx[M][outC] (constant per lane, the tile is a scalar offset);
    // forward = pixel (n, 2y + a, 2x) of y, recomputed per tile
    unsigned ovoff[PT];
    const int colw = nb * TN + wco * (CT / 2) * 32;   // first column of this wave
    auto epi_setup_tile = [&](int m0) {
#pragma unroll
        for (int pt = 0; pt < PT; ++pt) {
            const int m = m0 + (wpx * PT + pt) * 16 + l15;
            if (MODE == 0) {
                int n, y, x;
                pix_decode(m < M ? m : 0, n, y, x);
                ovoff[pt] = m < M ? (unsigned)((((n * 2 * p.H + 2 * y + pha) * 2 * p.W + 2 * x) * p.outC + colw + 8 * g4) * 2) : RSU_SENT;
            } else {
                ovoff[pt] = m < M ? (unsigned)((((wpx * PT + pt) * 16 + l15) * p.outC + colw + 8 * g4) * 2) : RSU_SENT;
            }
        }
    };
    auto epilogue = [&](int m0, f32x4(&acc)[CT][PT]) {
        const __amdgpu_buffer_rsrc_t orsrc = mk(p.out);
        const __amdgpu_buffer_rsrc_t mrsrc = mk(p.mask_src ? (const void*)p.mask_src : (const void*)p.out);
        const unsigned sbase = MODE == 0 ? 0u : sgpr((unsigned)((long)m0 * p.outC * 2));
        unsigned ones_pk = 0x00010001u;
        asm volatile("" : "+v"(ones_pk));
        unsigned voffs[NST];
        u32x4 mk4[NST];
#pragma unroll
        for (int e = 0; e < NST; ++e) {
            const int pt = e / (CT / 2), pp = e % (CT / 2);
            const bool cok = colw + pp * 32 + 8 * g4 < ncols;
            voffs[e] = (cok && ovoff[pt] != RSU_SENT) ? ovoff[pt] + pp * 64 : RSU_SENT;
        }
        if (MODE == 1 && p.mask_src) {
#pragma unroll
            for (int e = 0; e < NST; ++e) mk4[e] = __builtin_amdgcn_raw_buffer_load_b128(mrsrc, voffs[e], sbase, 0);
        }
#pragma unroll
        for (int e = 0; e < NST; ++e) {
            const int pt = e / (CT / 2), pp = e % (CT / 2);
            u32x4 r;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                r[i] = pack_bf2(acc[2 * pp][pt][2 * i], acc[2 * pp][pt][2 * i + 1]);
                r[2 + i] = pack_bf2(acc[2 * pp + 1][pt][2 * i], acc[2 * pp + 1][pt][2 * i + 1]);
            }
            if (MODE == 1 && p.mask_src) {
#pragma unroll
                for (int i = 0; i < 4; ++i) r[i] &= pos_mask_pk_bf16(mk4[e][i], ones_pk);
            }
            CT_STORE(r, voffs[e]);
        }
    };
    auto bar = [&]() {
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };

    // ---- prologue: stages 0 and 1 of the first tile
    pf_setup_tile();
    pf_issue();
    pf_issue();
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    bar();
    if (grp) bar();  // G1 sits out interval 0

    f32x4 acc[CT][PT];
    auto run_stream = [&](auto gconst) {
        constexpr int G = decltype(gconst)::value;
        int sit = 0;   // stage inside the current tile
        int ck = 0;    // current tile
        bool after_epi = false;
        epi_setup_tile((tile0 + 0 * tstride) * TM);
        auto phase = [&](auto slotc) {
            constexpr int SL = decltype(slotc)::value;
            if (sit == 0) {
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) {
                    const f32x4 bv = *(const __attribute__((address_space(3))) f32x4*)(lds + bias_base +
                                                                                         ((wco * (CT / 2) + (ct >> 1)) * 32 + 8 * g4 + (ct & 1) * 4) * 4);
#pragma unroll
                    for (int pt = 0; pt < PT; ++pt) {
                        acc[ct][pt] = bv;
                        asm volatile("" : "+v"(acc[ct][pt]));
                    }
                }
            }
            // ================= R interval
            bf16x8 fa[2][CT], fb[2][PT];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
                for (int ct = 0; ct < CT; ++ct)
                    fa[ks][ct] = *(const __attribute__((address_space(3))) bf16x8*)(lds + afrag + (SL * SLOT + ks * WPL + ct * 1024));
#pragma unroll
                for (int pt = 0; pt < PT; ++pt)
                    fb[ks][pt] = *(const __attribute__((address_space(3))) bf16x8*)(lds + boff[pt] + (SL * SLOT + ks * APL));
            }
            pf_issue();   // this wave's pieces of stage s+2 -> the slot of stage s-1
            if constexpr (G == 1) {
                if (after_epi) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPW + NST) : "memory");
                else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPW) : "memory");
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            bar();
            // ================= M interval
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int pt = 0; pt < PT; ++pt)
#pragma unroll
                    for (int ct = 0; ct < CT; ++ct) mfma_bf16_inplace(acc[ct][pt], fa[ks][ct], fb[ks][pt]);
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (G == 0) {
                if (after_epi) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPW + NST) : "memory");
                else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPW) : "memory");
            }
            after_epi = false;
            if (sit == NS - 1) mfma_results_fence();
            bar();
            if (sit == NS - 1) {
                // the finished tile's epilogue opens the wave's next R interval (its partner multiplies meanwhile)
                const int m0 = sgpr((tile0 + ck * tstride) * TM);
                epilogue(m0, acc);
                after_epi = true;
                sit = 0;
                ++ck;
                if (ck < my_tiles) epi_setup_tile((tile0 + ck * tstride) * TM);
            } else {
                ++sit;
            }
        };
        for (int gs = 0; gs < GS; gs += 3) {
            phase(std::integral_constant<int, 0>{});
            if (gs + 1 >= GS) break;
            phase(std::integral_constant<int, 1>{});
            if (gs + 2 >= GS) break;
            phase(std::integral_constant<int, 2>{});
        }
    };
    if (grp) run_stream(std::integral_constant<int, 1>{}); else run_stream(std::integral_constant<int, 0>{});
    if (!grp) bar();  // G0 sits out the last interval (G1's last epilogue)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // nothing may land in this workgroup's LDS after it has gone
}

static size_t ct_lds_bytes() { return 3 * (2 * 256 * 64 + 2 * 8 * 1024) + 512; }

template <int MODE>
static hipError_t ct_launch_one(const IgCtParams& p, int gx, hipStream_t st) {
    auto kern = igemm_ct_kernel<MODE>;
    static bool set = false;
    if (!set) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ct_lds_bytes());
        if (e != hipSuccess) return e;
        set = true;
    }
    hipLaunchKernelGGL(kern, dim3(gx), dim3(512), ct_lds_bytes(), st, p);
    return hipGetLastError();
}
// the launches igemm_ct is built for: whole 32-column groups (the channel-permuted stores) and 24-bit pixel indices
bool igemm_ct_supports(int mode, int N, int H, int W, int Ca, int Cn) {
    if (mode == 0 && (Cn % 32)) return false;   // a 32-column group must not straddle the two column phases b
    return (long)N * H * W < (1L << 24) && Ca % 8 == 0 && Cn % 8 == 0 && H >= 1 && W >= 1;
}
hipError_t igemm_ct_launch(int mode, const IgCtParams& p, int gx, hipStream_t st) {
    return mode == 0 ? ct_launch_one<0>(p, gx, st) : ct_launch_one<1>(p, gx, st);
}
