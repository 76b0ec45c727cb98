// igemm_fwd: NHWC bf16 implicit-GEMM convolution on v_mfma_f32_16x16x32_bf16 (gfx950).
//
// One workgroup = TN output channels x TM output pixels of one image. The GEMM is oriented
// D[co][pixel] = sum_k A[co][k] * B[k][pixel]:
//   A (weights) : pre-packed in MFMA fragment order by rsu_pack_* (pack.hip): each (chunk, tap, 16-row
//                 tile) is one 1-KiB block that is copied verbatim global -> LDS by one
//                 global_load_lds_dwordx4 wave-instruction and read back with ds_read_b128 (lane-linear,
//                 conflict-free).
//   B (pixels)  : the input rows/cols the tile touches ("halo tile") are staged ONCE per 32-channel chunk
//                 as a dense [R][CW] pixel image, 64 B per pixel, and re-used by all taps: tap (ky,kx)
//                 only adds (ky*CW + kx)*dil pixels to the fragment's base pixel. 16-byte channel groups
//                 are XOR-swizzled by pixel bit 2 (tools/lds_bank_sim.py: conflict-free for every tap
//                 shift); the swizzle is applied on the SOURCE address of the LDS-DMA and again on read.
// Rows of A are permuted inside each pair of 16-row tiles so that a lane ends up with 8 consecutive
// output channels of one pixel -> one 16-byte NHWC store per lane.
// Pipeline: stage = (chunk, tap group); weights double-buffered per stage, halo tile per chunk; all
// global->LDS traffic is LDS-DMA issued one stage ahead; one __syncthreads() per stage.
#include "igemm.h"

// occupancy target: the register-hungry shapes run one workgroup per CU, the others two (tells the scheduler not to
// serialise the fragment prefetch to save registers)
template <int WCO, int WPX, int CT, int PT, int NTAP, int KW, int TPS>
__global__ void __launch_bounds__(WCO* WPX * 64) __attribute__((amdgpu_waves_per_eu(1, (CT * PT > 16 ? 1 : 2))))
igemm_fwd_kernel(const IgFwdParams p) {
    constexpr int NW = WCO * WPX;
    constexpr int TN = WCO * CT * 16, TM = WPX * PT * 16;
    constexpr int WT = TN / 16;  // weight tiles per tap held in LDS
    constexpr int SPC = NTAP / TPS;
    constexpr int WBUF = TPS * WT * 1024;
    constexpr int KH = NTAP / KW;
    static_assert(NTAP % TPS == 0 && (CT % 2) == 0, "bad config");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __attribute__((address_space(3))) char* lds = (__attribute__((address_space(3))) char*)smem;
    const int ABUF = p.g.npix_max * 64;
    const int a_base = 2 * WBUF;

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wco = wave / WPX, wpx = wave % WPX;
    const int g4 = lane >> 4, l15 = lane & 15;

    // ---- tile decode (all wave-uniform)
    const int cob = blockIdx.x % p.ncob;
    int t = blockIdx.x / p.ncob;
    const int tpi = p.g.nstrips * p.g.tiles_per_strip;
    const int n = t / tpi;
    t -= n * tpi;
    const int strip = t / p.g.tiles_per_strip;
    const int mt = t - strip * p.g.tiles_per_strip;
    const int SW = p.g.SW, CW = p.g.CW;
    const int x0 = strip * SW;
    const int sw = min(SW, p.Wo - x0);
    const int m0 = mt * TM;
    const int y_first = m0 / SW;
    int y_last = (m0 + TM - 1) / SW;
    if (y_last > p.Ho - 1) y_last = p.Ho - 1;
    const int iy0 = y_first * p.stride - p.pad, ix0 = x0 * p.stride - p.pad;
    const int R = (y_last - y_first) * p.stride + (KH - 1) * p.dil + 1;
    const int npix = R * CW;
    const int npieces = (npix + 15) >> 4;
    const int ph = blockIdx.y;  // transposed-conv output phase (a*2+b), 0 otherwise
    const bf16_t* wp = p.wp + (long)ph * p.wp_y_stride;

    // ---- per-lane base pixel (in halo-tile coordinates) of each of this wave's pixel fragments
    int hpb[PT];
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) {
        const int m = m0 + (wpx * PT + pt) * 16 + l15;
        const int y = div_magic(m, p.g.inv_SW);
        const int tx = m - y * SW;
        const bool valid = (y < p.Ho) && (tx < sw);
        hpb[pt] = valid ? ((y - y_first) * p.stride * CW + tx * p.stride) : 0;
    }

    const int nchunks = p.nchunk[0] + p.nchunk[1] + p.nchunk[2];
    const int nstage = nchunks * SPC;

    auto issue_w = [&](int s) {
        const int chunk = s / SPC, tg = s - chunk * SPC;
        const int dst = (s & 1) * WBUF;
#pragma unroll
        for (int i0 = 0; i0 < TPS * WT; i0 += NW) {
            const int i = i0 + wave;
            if (i < TPS * WT) {
                const int tap_l = i / WT, tl = i - tap_l * WT;
                const int tap = tg * TPS + tap_l;
                const int tile = p.tile_off + cob * WT + tl;
                const bf16_t* src = tile < p.ntiles_w ? wp + ((long)(chunk * NTAP + tap) * p.ntiles_w + tile) * 512 + lane * 8
                                                      : (const bf16_t*)p.zero_page + lane * 8;
                dma16(src, (void*)(lds + dst + i * 1024));
            }
        }
    };
    auto issue_a = [&](int chunk) {
        int si = 0, cl = chunk;
        if (cl >= p.nchunk[0]) {
            cl -= p.nchunk[0];
            si = 1;
            if (cl >= p.nchunk[1]) {
                cl -= p.nchunk[1];
                si = 2;
            }
        }
        const bf16_t* sptr = si == 0 ? p.src[0].ptr : (si == 1 ? p.src[1].ptr : p.src[2].ptr);
        const int sH = si == 0 ? p.src[0].H : (si == 1 ? p.src[1].H : p.src[2].H);
        const int sW = si == 0 ? p.src[0].W : (si == 1 ? p.src[1].W : p.src[2].W);
        const int sC = si == 0 ? p.src[0].C : (si == 1 ? p.src[1].C : p.src[2].C);
        const int soy = si == 0 ? p.src[0].oy : (si == 1 ? p.src[1].oy : p.src[2].oy);
        const int sox = si == 0 ? p.src[0].ox : (si == 1 ? p.src[1].ox : p.src[2].ox);
        const int dst = a_base + (chunk & 1) * ABUF;
        const int c0 = cl * 32;
        for (int j = wave; j < npieces; j += NW) {
            const int hp = j * 16 + (lane >> 2);
            const int kg = (lane & 3) ^ ((hp >> 1) & 2);
            const int rr = div_magic(hp, p.g.inv_CW);
            const int cc = hp - rr * CW;
            const int iy = iy0 + rr, ix = ix0 + cc;
            const int ch = c0 + kg * 8;
            const bool ok = (hp < npix) && (iy >= 0) && (iy < p.Hin) && (ix >= 0) && (ix < p.Win) && (ch < sC);
            const bf16_t* src =
                ok ? sptr + ((long)(n * sH + iy + soy) * sW + (ix + sox)) * sC + ch : (const bf16_t*)p.zero_page;
            dma16(src, (void*)(lds + dst + j * 1024));
        }
    };

    f32x4 acc[CT][PT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int pt = 0; pt < PT; ++pt) acc[ct][pt] = f32x4{0.f, 0.f, 0.f, 0.f};

    issue_a(0);
    issue_w(0);
    __syncthreads();

    for (int s = 0; s < nstage; ++s) {
        const int chunk = s / SPC, tg = s - chunk * SPC;
        if (s + 1 < nstage) {
            issue_w(s + 1);
            if (tg == 0 && chunk + 1 < nchunks) issue_a(chunk + 1);
        }
        const int wb = (s & 1) * WBUF;
        const int ab = a_base + (chunk & 1) * ABUF;
        // software pipeline over the taps of this stage: the fragments of tap tl+1 are requested from LDS
        // before the MFMAs of tap tl are issued (two register sets, statically indexed after unrolling)
        bf16x8 fa[2][CT], fb[2][PT];
        auto load_tap = [&](int tl, bf16x8(&a)[CT], bf16x8(&b)[PT]) {
            const int tap = tg * TPS + tl;
            const int ky = tap / KW, kx = tap - ky * KW;
            const int toff = (ky * CW + kx) * p.dil;
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
                a[ct] = *(const __attribute__((address_space(3))) bf16x8*)(lds + wb + ((tl * WT + wco * CT + ct) * 64 + lane) * 16);
#pragma unroll
            for (int pt = 0; pt < PT; ++pt) {
                const int hp = hpb[pt] + toff;
                const int off = (hp << 6) + ((g4 ^ ((hp >> 1) & 2)) << 4);
                b[pt] = *(const __attribute__((address_space(3))) bf16x8*)(lds + ab + off);
            }
        };
        load_tap(0, fa[0], fb[0]);
#pragma unroll
        for (int tl = 0; tl < TPS; ++tl) {
            if (tl + 1 < TPS) load_tap(tl + 1, fa[(tl + 1) & 1], fb[(tl + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);  // keep the prefetch ahead of this tap's MFMAs
#pragma unroll
            for (int pt = 0; pt < PT; ++pt)
#pragma unroll
                for (int ct = 0; ct < CT; ++ct)
                    acc[ct][pt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[tl & 1][ct], fb[tl & 1][pt], acc[ct][pt], 0, 0, 0);
        }
        __syncthreads();
    }

    // ---- epilogue: lane holds channels co..co+7 of one pixel per (row pair, pixel fragment)
    const int ooffy = ph >> 1, ooffx = ph & 1;
#pragma unroll
    for (int pp = 0; pp < CT / 2; ++pp) {
        const int co = cob * TN + (wco * (CT / 2) + pp) * 32 + 8 * g4;
        if (co >= p.Cout) continue;
        float bv[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) bv[i] = p.bias ? p.bias[co + i] : 0.f;
#pragma unroll
        for (int pt = 0; pt < PT; ++pt) {
            const int m = m0 + (wpx * PT + pt) * 16 + l15;
            const int y = div_magic(m, p.g.inv_SW);
            const int tx = m - y * SW;
            if (y >= p.Ho || tx >= sw) continue;
            const long idx = ((long)(n * p.oH + y * p.ostride + ooffy) * p.oW + (x0 + tx) * p.ostride + ooffx) * p.outC + co;
            float v[8];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                v[i] = acc[2 * pp][pt][i] + bv[i];
                v[4 + i] = acc[2 * pp + 1][pt][i] + bv[4 + i];
            }
            if (p.mask_src) {
                const u32x4 mk = *(const u32x4*)(p.mask_src + idx);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if (!(bf_lo(mk[i]) > 0.f)) v[2 * i] = 0.f;
                    if (!(bf_hi(mk[i]) > 0.f)) v[2 * i + 1] = 0.f;
                }
            }
            if (p.accumulate) {
                const u32x4 o = *(const u32x4*)(p.out + idx);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    v[2 * i] += bf_lo(o[i]);
                    v[2 * i + 1] += bf_hi(o[i]);
                }
            }
            if (p.relu) {
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = fmaxf(v[i], 0.f);
            }
            u32x4 r;
#pragma unroll
            for (int i = 0; i < 4; ++i) r[i] = pack_bf2(v[2 * i], v[2 * i + 1]);
            *(u32x4*)(p.out + idx) = r;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// configurations: (WCO, WPX, CT, PT) -> TN = WCO*CT*16 channels, TM = WPX*PT*16 pixels
// ---------------------------------------------------------------------------------------------
template <int CFG> struct FwdCfg;
template <> struct FwdCfg<IGF_CFG_64x256> { static constexpr int WCO = 1, WPX = 4, CT = 4, PT = 4; };
template <> struct FwdCfg<IGF_CFG_128x256> { static constexpr int WCO = 2, WPX = 2, CT = 4, PT = 8; };
template <> struct FwdCfg<IGF_CFG_128x128> { static constexpr int WCO = 2, WPX = 2, CT = 4, PT = 4; };
template <> struct FwdCfg<IGF_CFG_128x64> { static constexpr int WCO = 4, WPX = 1, CT = 2, PT = 4; };
template <> struct FwdCfg<IGF_CFG_64x128> { static constexpr int WCO = 1, WPX = 4, CT = 4, PT = 2; };

static constexpr int tps_for(int TN, int ntap) {
    return ntap == 9 ? (TN <= 64 ? 9 : 3) : (ntap == 4 ? (TN <= 64 ? 4 : 2) : 1);
}

IgFwdCfgInfo igemm_fwd_cfg_info(int cfg) {
    switch (cfg) {
#define CASE(C) \
    case C: return IgFwdCfgInfo{FwdCfg<C>::WCO * FwdCfg<C>::CT * 16, FwdCfg<C>::WPX * FwdCfg<C>::PT * 16, FwdCfg<C>::WCO * FwdCfg<C>::WPX * 64};
        CASE(IGF_CFG_64x256)
        CASE(IGF_CFG_128x256)
        CASE(IGF_CFG_128x128)
        CASE(IGF_CFG_128x64)
        CASE(IGF_CFG_64x128)
#undef CASE
    }
    return IgFwdCfgInfo{0, 0, 0};
}

size_t igemm_fwd_lds_bytes(int cfg, int ntap, int npix_max) {
    const IgFwdCfgInfo ci = igemm_fwd_cfg_info(cfg);
    const int tps = tps_for(ci.TN, ntap);
    return (size_t)2 * tps * (ci.TN / 16) * 1024 + (size_t)2 * npix_max * 64;
}

template <int CFG, int NTAP, int KW>
static hipError_t launch_one(const IgFwdParams& p, int gx, int gy, hipStream_t st) {
    using C = FwdCfg<CFG>;
    constexpr int TN = C::WCO * C::CT * 16;
    constexpr int TPS = tps_for(TN, NTAP);
    auto kern = igemm_fwd_kernel<C::WCO, C::WPX, C::CT, C::PT, NTAP, KW, TPS>;
    const size_t lds = igemm_fwd_lds_bytes(CFG, NTAP, p.g.npix_max);
    static size_t lds_set = 0;
    if (lds > lds_set) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        lds_set = lds;
    }
    hipLaunchKernelGGL(kern, dim3(gx, gy), dim3(C::WCO * C::WPX * 64), lds, st, p);
    return hipGetLastError();
}

template <int CFG>
static hipError_t launch_cfg(int ntap, const IgFwdParams& p, int gx, int gy, hipStream_t st) {
    switch (ntap) {
        case 9: return launch_one<CFG, 9, 3>(p, gx, gy, st);
        case 4: return launch_one<CFG, 4, 2>(p, gx, gy, st);
        case 1: return launch_one<CFG, 1, 1>(p, gx, gy, st);
    }
    return hipErrorInvalidValue;
}

hipError_t igemm_fwd_launch(int cfg, int ntap, const IgFwdParams& p, int gx, int gy, hipStream_t st) {
    switch (cfg) {
        case IGF_CFG_64x256: return launch_cfg<IGF_CFG_64x256>(ntap, p, gx, gy, st);
        case IGF_CFG_128x256: return launch_cfg<IGF_CFG_128x256>(ntap, p, gx, gy, st);
        case IGF_CFG_128x128: return launch_cfg<IGF_CFG_128x128>(ntap, p, gx, gy, st);
        case IGF_CFG_128x64: return launch_cfg<IGF_CFG_128x64>(ntap, p, gx, gy, st);
        case IGF_CFG_64x128: return launch_cfg<IGF_CFG_64x128>(ntap, p, gx, gy, st);
    }
    return hipErrorInvalidValue;
}
