// igemm_wgrad_body.h -- the generic weight-gradient implicit-GEMM kernel body, shared by igemm_wgrad.hip (one launch per layer) and
// igemm_wgpp.hip (the grouped launch).
#pragma once
// igemm_wgrad: weight-gradient implicit GEMM, reduction over pixels, on v_mfma_f32_16x16x32_bf16.
//
//   slab[z][tap][cs][cf] = sum over the pixels of split z of  S[pix*stride + tap*dil][cs] * F[pix][cf]
//
// Both operands are channel-contiguous (NHWC) in memory but the MFMA wants 8 consecutive REDUCTION
// indices (pixels) per lane, so both are staged pixel-major in LDS and read with the gfx950 transposed
// LDS read ds_read_b64_tr_b16 (4 pixel rows x 16 channels -> channel-per-lane, 4 pixels per lane).
// The reduction order inside a 32-pixel MFMA step is permuted identically for both operands
// (lane group g takes pixels 4g..4g+3 and 16+4g..16+4g+3), which makes each 32-lane half of a tr-read
// touch 8 consecutive pixels: with 128-byte pixels and the 32-byte-block XOR swizzle (pixel>>1)&3 that
// is bank-conflict free (tools/lds_bank_sim.py).
// The S halo tile is staged once per pixel tile and re-used by all taps (tap = pixel offset), F once.
// Each workgroup owns a 64(cf) x CSB(cs) block for ALL taps (9*4 = 36 accumulator tiles per wave at the
// default shape), walks the pixel tiles of its split with LDS-DMA staging through a ring of p.nbuf (2 or 3) buffers --
// with 3 the loads run TWO tiles ahead and each wave waits with a counted s_waitcnt vmcnt(pieces per tile) -- and writes
// one fp32 slab; reduce_slabs (elementwise.hip) sums the splits deterministically.
#include "igemm.h"


static __device__ __forceinline__ void bdma16(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned soff, void* lds_wave_base) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds_wave_base, 16, voff, soff, 0, 0);
}
#define RSU_WG_WAIT(N) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory")
#define RSU_SENT 0x80000000u   // voffset the range check always rejects (num_records = 0x7fffffff): the lane deposits zeros

// Tiles are ALIGNED (strip width SW = 2^lsw divides TMK; a tile = TMK/SW full rows of one strip), so every per-lane LDS
// read offset is a workgroup constant and a pixel tile only contributes scalar bases + edge validity.
// KG = 1: four waves, one per SIMD. KG = 2: eight waves; wave group g = wave/4 reduces k-steps [g*KS/2, (g+1)*KS/2) of every
// pixel tile into its OWN accumulators, so two waves share each SIMD and cover each other's LDS latency / DMA bookkeeping;
// the two partial sums meet once, through LDS, after the last tile.
// The kernel body as a device function: `lds` = the workgroup's dynamic LDS, (cfb, csb, z) = the (F channel block, S channel block,
// pixel split) this call works on. igemm_wgrad_kernel calls it once per workgroup; the grouped launch (igemm_wgpp.hip,
// igemm_wg_group_kernel) calls it for every unit of its workgroup's list.
template <int WCF, int WCS, int CFT, int CST, int NTAP, int KW, int TMK, int KG, class P>
__device__ __forceinline__ void igemm_wgrad_body(const P& p, __attribute__((address_space(3))) char* lds, const int cfb, const int csb,
                                                 const int z, const unsigned tid) {
    constexpr int NW = WCF * WCS * KG;
    constexpr int KS = TMK / 32;          // 32-pixel MFMA steps per tile
    constexpr int KSG = KS / KG;          // ... per wave group
    static_assert(KS % KG == 0, "k-steps must split evenly over the wave groups");
    constexpr int CFB = WCF * CFT * 16;  // 64 or 128 (staged as CFB/64 planes of 64 channels)
    constexpr int NPL = CFB / 64;
    constexpr int CSB = WCS * CST * 16;  // 64 or 16
    constexpr int KH = NTAP / KW;
    constexpr int SPITCH = CSB * 2;      // bytes per S pixel in LDS
    constexpr int LPP = CSB / 8;         // lanes (16-B pieces) per S pixel
    constexpr int PPP = 64 / LPP;        // S pixels per DMA piece
    constexpr int FPL = TMK * 128;       // bytes of one F plane
    constexpr int FBUF = NPL * FPL;
    static_assert(CFB % 64 == 0, "F block is a multiple of 64 channels");
    const int nbuf = p.nbuf, nsw = p.nsw;      // ring depth; S pieces per wave per tile (every wave issues exactly nsw)
    const int SBUF = nsw * NW * 1024;
    const int s_base = nbuf * FBUF;

    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kgrp = wave / (WCF * WCS), wave4 = wave % (WCF * WCS);
    const int wcf = wave4 / WCS, wcs = wave4 % WCS;
    const int g4 = lane >> 4, l15 = lane & 15, q4 = l15 >> 2, p4 = lane & 3;
    const int zs = z;  // slab of this workgroup
    const int SW = p.g.SW, CW = p.g.CW, lsw = p.lsw, TR = TMK >> lsw;
    const int tpi = p.g.nstrips * p.g.tiles_per_strip;
    const int Hs = (p.Hf - 1) * p.stride + (KH - 1) * p.dil + 1;  // S window extent
    const int Ws = (p.Wf - 1) * p.stride + (KW - 1) * p.dil + 1;

    f32x4 acc[NTAP][CFT][CST];
#pragma unroll
    for (int t = 0; t < NTAP; ++t)
#pragma unroll
        for (int a = 0; a < CFT; ++a)
#pragma unroll
            for (int b = 0; b < CST; ++b) acc[t][a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    // BiasAddGrad rides along: sum_pix F[pix][cf] = F^T x ones, one extra MFMA per k-step in the waves of cs block 0
    // (which cs wave of a cf row carries them is free -- all see the same F fragments: pick different SIMDs for the different cf rows /
    // wave groups, wave w runs on SIMD w % 4; with wcs == 0 everywhere both bias waves sat on SIMD 0: 80 MFMAs per k-step against 72)
    const bool do_bias = (p.bslab != nullptr) && (csb == 0) && (wcs == (wcf + kgrp) % WCS);
    f32x4 accb[CFT];
#pragma unroll
    for (int a = 0; a < CFT; ++a) accb[a] = f32x4{0.f, 0.f, 0.f, 0.f};
    // ... and the column sums of S over every tap (ones x S): the bias gradient of the transposed conv (S = dy, taps = the four
    // disjoint output phases), in the waves of cf block 0
    const bool do_sbias = NTAP < 9 && (p.sbslab != nullptr) && (cfb == 0) && (wcf == 0);  // (only the transposed conv asks for them: no per-step branch in the 3x3 kernels)
    f32x4 accs[CST];
#pragma unroll
    for (int b = 0; b < CST; ++b) accs[b] = f32x4{0.f, 0.f, 0.f, 0.f};

    // per-lane byte offsets of the transposed LDS reads (workgroup constants) for the lane's pixel INSIDE a 32-pixel k-step;
    // the k-step itself adds a compile-time constant to the F offset (32 pixels x 128 bytes) and a scalar to the S offset
    // (whole rows, or half a row when SW = 64): both are multiples of 8 pixels, so the swizzle terms do not change
    int f0[2];              // F tile: [read], cf tile 0 of this wave; tile ct is the same address with bits 5-6 XOR ct (the swizzle)
    int soff[2][KW][CST];   // S halo tile: [read][kx][cs tile]; ky adds whole rows (CW % 8 == 0 keeps the swizzle)
#pragma unroll
    for (int rd = 0; rd < 2; ++rd) {
        const int ml = rd * 16 + 4 * g4 + q4;
        {
            // channel of cf tile ct: ch = (wcf*CFT + ct)*16 + 4*p4; its 32-byte block inside the 128-byte pixel is
            // ((ch>>4)&3) ^ ((ml>>1)&3). CFT = 4: wcf*CFT is a multiple of 4, so block(ct) = ct ^ m  ->  address(ct) = address(0) ^ (ct<<5).
            // CFT < 4 keeps the general form (the XOR below then needs (wcf*CFT)&3 folded in, which address(0) already has)
            const int ch = (wcf * CFT) * 16 + 4 * p4;
            const int chl = ch & 63;
            f0[rd] = (ch >> 6) * FPL + ml * 128 + ((((chl >> 4) ^ ((ml >> 1) & 3))) << 5) + (chl & 15) * 2;
        }
        const int ty = ml >> lsw, tx = ml & (SW - 1);
        const int hp0 = ty * p.stride * CW + tx * p.stride;
#pragma unroll
        for (int kx = 0; kx < KW; ++kx) {
            const int hp = hp0 + kx * p.dil;
#pragma unroll
            for (int st = 0; st < CST; ++st) {
                const int ch = (wcs * CST + st) * 16 + 4 * p4;
                soff[rd][kx][st] = (LPP == 8) ? (hp * 128 + ((((ch >> 4) ^ ((hp >> 1) & 3))) << 5) + (ch & 15) * 2)
                                              : (hp * SPITCH + ch * 2);
            }
        }
    }
    // S byte offset of k-step k (of the whole tile): scalar
    auto sdelta = [&](int k) {
        const int m0 = k * 32;
        return (((m0 >> lsw) * CW + (m0 & (SW - 1))) * p.stride) * SPITCH;
    };

    struct Tile { int n, x0, y0; };
    auto decode = [&](int tile) {
        Tile T;
        T.n = tile / tpi;
        int r = tile - T.n * tpi;
        const int strip = r / p.g.tiles_per_strip;
        T.x0 = strip * SW;
        T.y0 = (r - strip * p.g.tiles_per_strip) * TR;
        return T;
    };
    auto mk = [&](const void* ptr) { return __builtin_amdgcn_make_buffer_rsrc((void*)ptr, 0, 0x7fffffff, 0x00020000); };
    static_assert((NPL * TMK / 8) % NW == 0, "F pieces per wave must be a constant");
    constexpr int NFW = NPL * TMK / 8 / NW;
    // Per-lane source offsets of the staging pieces do not depend on the tile (tiles are aligned): they are computed once (NFW + up to
    // 5 VGPRs) and an interior tile -- every pixel of the F tile and of the S halo inside the image -- issues its pieces with no
    // per-lane arithmetic at all (the address math was ~60 % of the kernel's VALU instructions, all of them in front of a DMA
    // instruction the wave's MFMAs wait behind). Edge tiles (last strip / last rows) redo the validity tests per lane; `ln` = lane
    // id made opaque there keeps that arithmetic from being hoisted out of the tile loop into spilled registers.
    // (Measured: +5-8 % on the layers that run the two-wave-group shapes; the 128x64 shape, which has no VGPRs to spare for the
    // cache, lost 4-7 % with it and keeps computing its offsets per tile.)
    constexpr bool CACHED = KG == 2;
    constexpr int NSW_MAX = 5;
    unsigned fvo[NFW], svo[NSW_MAX];
#pragma unroll
    for (int q = 0; q < (CACHED ? NFW : 0); ++q) {
        const int j = q * NW + wave;
        const int pl = j / (TMK / 8), jj = j - pl * (TMK / 8);
        const int ml = jj * 8 + (lane >> 3);
        const int c = (lane & 7) ^ (((ml >> 1) & 3) << 1);
        const int ty = ml >> lsw, tx = ml & (SW - 1);
        fvo[q] = (cfb * CFB + pl * 64 + c * 8 < p.Cf) ? (unsigned)(((ty * p.Wf + tx) * p.Cf + pl * 64 + c * 8) * 2) : RSU_SENT;
    }
#pragma unroll
    for (int q = 0; q < (CACHED ? NSW_MAX : 0); ++q) {
        const int j = q * NW + wave;
        const int hp = j * PPP + lane / LPP;
        const int pc = lane % LPP;
        const int c = (LPP == 8) ? (pc ^ (((hp >> 1) & 3) << 1)) : pc;
        const int rr = div_magic(hp, p.g.inv_CW);
        const int cc = hp - rr * CW;
        svo[q] = (q < nsw && hp < p.g.npix_max && csb * CSB + c * 8 < p.S.C) ? (unsigned)(((rr * p.S.W + cc) * p.S.C + c * 8) * 2) : RSU_SENT;
    }
    const int halo_rows = div_magic(p.g.npix_max - 1, p.g.inv_CW) + 1;  // rows of the staged S halo tile
    auto issue = [&](const Tile& T, int buf) {
        if constexpr (CACHED) {
        // F tile: TMK pixels x CFB channels as planes of 64 channels, 8 pixels per piece; rows/cols beyond the image come back as zeros
        {
            const __amdgpu_buffer_rsrc_t rf = mk(p.F);
            const unsigned soffF = (unsigned)((((long)(T.n * p.Hf + T.y0) * p.Wf + T.x0) * p.Cf + cfb * CFB) * 2);
            const bool inside = (T.y0 + TR <= p.Hf) && (T.x0 + SW <= p.Wf);  // wave-uniform
            if (inside) {
#pragma unroll
                for (int q = 0; q < NFW; ++q) bdma16(rf, fvo[q], soffF, (void*)(lds + buf * FBUF + (q * NW + wave) * 1024));
            } else {
                int ln = lane;
                asm volatile("" : "+v"(ln));
#pragma unroll
                for (int q = 0; q < NFW; ++q) {
                    const int j = q * NW + wave;
                    const int pl = j / (TMK / 8), jj = j - pl * (TMK / 8);
                    const int ml = jj * 8 + (ln >> 3);
                    const int ty = ml >> lsw, tx = ml & (SW - 1);
                    const bool ok = (T.y0 + ty < p.Hf) && (T.x0 + tx < p.Wf);
                    unsigned vo;
                    if constexpr (CACHED) {
                        vo = fvo[q];
                    } else {
                        const int c = (ln & 7) ^ (((ml >> 1) & 3) << 1);
                        vo = (cfb * CFB + pl * 64 + c * 8 < p.Cf) ? (unsigned)(((ty * p.Wf + tx) * p.Cf + pl * 64 + c * 8) * 2) : RSU_SENT;
                    }
                    bdma16(rf, ok ? vo : RSU_SENT, soffF, (void*)(lds + buf * FBUF + j * 1024));
                }
            }
        }
        // S halo tile
        {
            const __amdgpu_buffer_rsrc_t rs = mk(p.S.ptr);
            const int iy0 = T.y0 * p.stride, ix0 = T.x0 * p.stride;
            const unsigned soffS = (unsigned)((((long)(T.n * p.S.H + iy0 + p.S.oy) * p.S.W + ix0 + p.S.ox) * p.S.C + csb * CSB) * 2);
            const bool inside = (iy0 + halo_rows <= Hs) && (ix0 + CW <= Ws);  // wave-uniform
            if (inside) {
#pragma unroll
                for (int q = 0; q < NSW_MAX; ++q)
                    if (q < nsw) bdma16(rs, svo[q], soffS, (void*)(lds + s_base + buf * SBUF + (q * NW + wave) * 1024));
            } else {
                int ln = lane;
                asm volatile("" : "+v"(ln));
#pragma unroll
                for (int q = 0; q < NSW_MAX; ++q) {  // pieces past the halo tile (hp >= npix_max) carry RSU_SENT already or are never read
                    if (q >= nsw) break;
                    const int j = q * NW + wave;
                    const int hp = j * PPP + ln / LPP;
                    const int rr = div_magic(hp, p.g.inv_CW);
                    const int cc = hp - rr * CW;
                    const bool ok = (iy0 + rr < Hs) && (ix0 + cc < Ws);
                    unsigned vo;
                    if constexpr (CACHED) {
                        vo = svo[q];
                    } else {
                        const int pc = ln % LPP;
                        const int c = (LPP == 8) ? (pc ^ (((hp >> 1) & 3) << 1)) : pc;
                        vo = (hp < p.g.npix_max && csb * CSB + c * 8 < p.S.C) ? (unsigned)(((rr * p.S.W + cc) * p.S.C + c * 8) * 2) : RSU_SENT;
                    }
                    bdma16(rs, ok ? vo : RSU_SENT, soffS, (void*)(lds + s_base + buf * SBUF + j * 1024));
                }
            }
            if (nsw > NSW_MAX) {  // big halo tiles of the two-buffer ring: the pieces beyond the cached ones, in full
                int ln = lane;
                asm volatile("" : "+v"(ln));
                for (int q = NSW_MAX; q < nsw; ++q) {
                    const int j = q * NW + wave;
                    const int hp = j * PPP + ln / LPP;
                    const int pc = ln % LPP;
                    const int c = (LPP == 8) ? (pc ^ (((hp >> 1) & 3) << 1)) : pc;
                    const int rr = div_magic(hp, p.g.inv_CW);
                    const int cc = hp - rr * CW;
                    const bool ok = (hp < p.g.npix_max) && (iy0 + rr < Hs) && (ix0 + cc < Ws) && (csb * CSB + c * 8 < p.S.C);
                    const unsigned voff = ok ? (unsigned)(((rr * p.S.W + cc) * p.S.C + c * 8) * 2) : RSU_SENT;
                    bdma16(rs, voff, soffS, (void*)(lds + s_base + buf * SBUF + j * 1024));
                }
            }
        }
        } else {
        int ln = lane;
        asm volatile("" : "+v"(ln));
        // F tile: TMK pixels x CFB channels as planes of 64 channels, 8 pixels per piece; rows/cols beyond the image come back as zeros
        {
            const __amdgpu_buffer_rsrc_t rf = mk(p.F);
            const unsigned soffF = (unsigned)((((long)(T.n * p.Hf + T.y0) * p.Wf + T.x0) * p.Cf + cfb * CFB) * 2);
#pragma unroll
            for (int q = 0; q < NFW; ++q) {
                const int j = q * NW + wave;
                const int pl = j / (TMK / 8), jj = j - pl * (TMK / 8);
                const int ml = jj * 8 + (ln >> 3);
                const int c = (ln & 7) ^ (((ml >> 1) & 3) << 1);
                const int ty = ml >> lsw, tx = ml & (SW - 1);
                const bool ok = (T.y0 + ty < p.Hf) && (T.x0 + tx < p.Wf) && (cfb * CFB + pl * 64 + c * 8 < p.Cf);
                const unsigned voff = ok ? (unsigned)(((ty * p.Wf + tx) * p.Cf + pl * 64 + c * 8) * 2) : RSU_SENT;
                bdma16(rf, voff, soffF, (void*)(lds + buf * FBUF + j * 1024));
            }
        }
        // S halo tile
        {
            const __amdgpu_buffer_rsrc_t rs = mk(p.S.ptr);
            const int iy0 = T.y0 * p.stride, ix0 = T.x0 * p.stride;
            const unsigned soffS = (unsigned)((((long)(T.n * p.S.H + iy0 + p.S.oy) * p.S.W + ix0 + p.S.ox) * p.S.C + csb * CSB) * 2);
            for (int q = 0; q < nsw; ++q) {  // pieces past the halo tile (hp >= npix_max) fail the window test below or are never read
                const int j = q * NW + wave;
                const int hp = j * PPP + ln / LPP;
                const int pc = ln % LPP;
                const int c = (LPP == 8) ? (pc ^ (((hp >> 1) & 3) << 1)) : pc;
                const int rr = div_magic(hp, p.g.inv_CW);
                const int cc = hp - rr * CW;
                const bool ok = (hp < p.g.npix_max) && (iy0 + rr < Hs) && (ix0 + cc < Ws) && (csb * CSB + c * 8 < p.S.C);
                const unsigned voff = ok ? (unsigned)(((rr * p.S.W + cc) * p.S.C + c * 8) * 2) : RSU_SENT;
                bdma16(rs, voff, soffS, (void*)(lds + s_base + buf * SBUF + j * 1024));
            }
        }
        }
    };

    // ---- prologue: the first nbuf-1 tiles of this split
    int tile = z;
    {
        int t = tile;
        for (int d = 0; d < nbuf - 1 && t < p.ntiles_total; ++d, t += p.nsplit) issue(decode(t), d);
    }
    int buf = 0;                      // ring slot of the tile being reduced
    int ibuf = nbuf - 1;              // ring slot the next prefetch goes to
    for (; tile < p.ntiles_total; tile += p.nsplit) {
        // tile `tile` has landed once this wave's own pieces are done (all but the NFW + nsw pieces of the tile after it,
        // when that one is in flight) and everybody passed the barrier; the barrier also retires the slot read last round
        if (nbuf == 3 && tile + p.nsplit < p.ntiles_total) {
            switch (nsw) {
                case 1: RSU_WG_WAIT(NFW + 1); break;
                case 2: RSU_WG_WAIT(NFW + 2); break;
                case 3: RSU_WG_WAIT(NFW + 3); break;
                case 4: RSU_WG_WAIT(NFW + 4); break;
                case 5: RSU_WG_WAIT(NFW + 5); break;
                default: RSU_WG_WAIT(0); break;
            }
        } else {
            RSU_WG_WAIT(0);
        }
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        // staging loads of the tile nbuf-1 ahead: the first half of the waves issues them right behind the barrier, the second half
        // (the SIMD partners) behind its first k-step -- each LDS-DMA instruction holds a wave's issue for ~100 cycles, and with
        // both partners issuing at once the matrix pipe of the SIMD sat idle for that long every tile
        const int nxt = tile + (nbuf - 1) * p.nsplit;
        const bool have_nxt = nxt < p.ntiles_total && p.dbg != 1;
        const int ibuf_now = ibuf;
        const bool late = wave >= NW / 2;
        if (have_nxt && !late) issue(decode(p.dbg == 2 ? z : nxt), ibuf_now);
        ibuf = ibuf + 1 == nbuf ? 0 : ibuf + 1;
        const int fb = buf * FBUF, sb = s_base + buf * SBUF;
        // A operand: F^T (rows = cf), two transposed reads per 16-channel tile
        auto tr_read = [&](int off) {
            return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bf16x4*)(lds + off));
        };
        auto load_F = [&](int ks, bf16x8(&fa)[CFT]) {
            const int fk = fb + (kgrp * KSG + ks) * (32 * 128);  // wave-uniform
            const int a0 = fk + f0[0], a1 = fk + f0[1];  // fk is a multiple of 128: bits 5-6 stay the swizzled block index
#pragma unroll
            for (int ct = 0; ct < CFT; ++ct) {
                // tiles of one wave never straddle a 64-channel plane and start at a multiple of CFT <= 4 tiles: XOR-ing the tile
                // index into the block bits is exact ((b0 + ct) ^ m == (b0 ^ m) ^ ct when b0 is a multiple of CFT, CFT a power of two)
                const bf16x4 lo = tr_read(a0 ^ (ct << 5)), hi = tr_read(a1 ^ (ct << 5));
                fa[ct] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
            }
        };
        // B operand: S (cols = cs) shifted by the tap
        auto load_S = [&](int ks, int tap, bf16x8(&sv)[CST]) {
            const int ky = tap / KW, kx = tap - ky * KW;
            const int rowoff = sb + ky * CW * p.dil * SPITCH + sdelta(kgrp * KSG + ks);  // wave-uniform
#pragma unroll
            for (int st = 0; st < CST; ++st) {
                const bf16x4 lo = tr_read(rowoff + soff[0][kx][st]), hi = tr_read(rowoff + soff[1][kx][st]);
                sv[st] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
            }
        };
        // one software-pipelined sequence over (k-step, tap): operands of step+1 are requested before the MFMAs of step
        constexpr int NS = KSG * NTAP;
        // F fragments are double-buffered across k-steps unless the wave already holds 64 F channels x all k-steps (128x64
        // shape: 16 more VGPRs would spill); then they are fetched at the head of each k-step and the partner wave covers the wait
        constexpr int FB = 2;
        bf16x8 fa[FB][CFT], sv[2][CST];
        load_F(0, fa[0]);
        load_S(0, 0, sv[0]);
#pragma unroll
        for (int step = 0; step < NS; ++step) {
            const int ks = step / NTAP, tap = step % NTAP;
            if (FB == 1 && tap == 0 && step > 0) load_F(ks, fa[0]);
            if (step + 1 < NS) {
                const int ks1 = (step + 1) / NTAP, tap1 = (step + 1) % NTAP;
                if (FB == 2 && tap1 == 0) load_F(ks1, fa[ks1 & (FB - 1)]);
                load_S(ks1, tap1, sv[(step + 1) & 1]);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ct = 0; ct < CFT; ++ct)
#pragma unroll
                for (int st = 0; st < CST; ++st)
                    mfma_bf16_inplace(acc[tap][ct][st], fa[ks & (FB - 1)][ct], sv[step & 1][st]);  // vDst == SrcC by construction (rsu_common.h)
            if (do_sbias) {
                unsigned o1 = 0x3f803f80u;
                asm volatile("" : "+v"(o1));
                const u32x4 o4 = {o1, o1, o1, o1};
                bf16x8 ones = __builtin_bit_cast(bf16x8, o4);
                asm volatile("s_nop 3" : "+v"(ones));  // VALU-written operand -> (asm) MFMA read: the hazard recogniser cannot see it
#pragma unroll
                for (int st = 0; st < CST; ++st) mfma_bf16_inplace(accs[st], ones, sv[step & 1][st]);
            }
            if (step == (KSG >= 4 ? 2 : 1) * NTAP - 1 && late && have_nxt) issue(decode(p.dbg == 2 ? z : nxt), ibuf_now);
            if (tap == 0 && do_bias) {
                // the all-ones operand is re-materialised here (four v_mov per k-step) instead of living in four VGPRs
                unsigned o1 = 0x3f803f80u;
                asm volatile("" : "+v"(o1));
                const u32x4 o4 = {o1, o1, o1, o1};
                bf16x8 ones = __builtin_bit_cast(bf16x8, o4);
                asm volatile("s_nop 3" : "+v"(ones));  // VALU-written operand -> (asm) MFMA read: the hazard recogniser cannot see it
#pragma unroll
                for (int ct = 0; ct < CFT; ++ct)
                    mfma_bf16_inplace(accb[ct], fa[ks & (FB - 1)][ct], ones);
            }
        }
        buf = buf + 1 == nbuf ? 0 : buf + 1;
    }
    mfma_results_fence();
    __syncthreads();  // the staging buffers are dead from here on

    // ---- KG = 2: wave group 1 hands its partial sums to group 0 through LDS (the staging buffers are dead by now; the launch
    // reserves NTAP*CFB*CSB*4 + 4 KiB bytes), fixed order group0 + group1: one slab per workgroup, still deterministic
    if constexpr (KG == 2) {
        f32x4 __attribute__((address_space(3)))* red = (f32x4 __attribute__((address_space(3)))*)lds;
        const int slot = wave4 * 64 + lane;
        if (kgrp == 1) {
#pragma unroll
            for (int t = 0; t < NTAP; ++t)
#pragma unroll
                for (int a = 0; a < CFT; ++a)
#pragma unroll
                    for (int b = 0; b < CST; ++b) red[((t * CFT + a) * CST + b) * (WCF * WCS * 64) + slot] = acc[t][a][b];
            // bias sums: the wave that carries them in this group (a different cs wave than in group 0, see do_bias) parks them in the
            // slot of group 0's bias wave of the same cf row
            if (wcs == (wcf + 1) % WCS) {
                const int slot_b = (wcf * WCS + wcf % WCS) * 64 + lane;
#pragma unroll
                for (int a = 0; a < CFT; ++a) red[((NTAP * CFT + a) * CST) * (WCF * WCS * 64) + slot_b] = accb[a];
            }
            if (NTAP < 9) {  // (the 9-tap hand-off already fills the 160 KiB; only the transposed conv carries S sums)
#pragma unroll
                for (int b = 0; b < CST; ++b) red[((NTAP * CFT + CFT) * CST + b) * (WCF * WCS * 64) + slot] = accs[b];
            }
        }
        __syncthreads();
        if (kgrp == 1) return;
#pragma unroll
        for (int t = 0; t < NTAP; ++t)
#pragma unroll
            for (int a = 0; a < CFT; ++a)
#pragma unroll
                for (int b = 0; b < CST; ++b) acc[t][a][b] += red[((t * CFT + a) * CST + b) * (WCF * WCS * 64) + slot];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int a = 0; a < CFT; ++a) accb[a] += red[((NTAP * CFT + a) * CST) * (WCF * WCS * 64) + slot];
        if (NTAP < 9) {
#pragma unroll
            for (int b = 0; b < CST; ++b) accs[b] += red[((NTAP * CFT + CFT) * CST + b) * (WCF * WCS * 64) + slot];
        }
    }
    if (do_sbias && g4 == 0) {  // every row of accs holds the same sums: row 0 (lanes 0..15, element 0) writes them
#pragma unroll
        for (int st = 0; st < CST; ++st) {
            const int cs = csb * CSB + (wcs * CST + st) * 16 + l15;
            if (cs < p.S.C) p.sbslab[(long)zs * p.slab_stride + cs] = accs[st][0];
        }
    }
    if (do_bias && l15 == 0) {  // every column of accb holds the same sums: column 0 writes them
#pragma unroll
        for (int ct = 0; ct < CFT; ++ct) {
            const int cf = cfb * CFB + (wcf * CFT + ct) * 16 + 4 * g4;
            if (cf < p.Cf) *(f32x4*)(p.bslab + (long)zs * p.slab_stride + cf) = accb[ct];
        }
    }
    // ---- write this split's slab: D rows = cf (4 consecutive per lane), cols = cs
#pragma unroll
    for (int tap = 0; tap < NTAP; ++tap)
#pragma unroll
        for (int st = 0; st < CST; ++st) {
            const int cs = csb * CSB + (wcs * CST + st) * 16 + l15;
            if (cs >= p.S.C) continue;
#pragma unroll
            for (int ct = 0; ct < CFT; ++ct) {
                const int cf = cfb * CFB + (wcf * CFT + ct) * 16 + 4 * g4;
                if (cf >= p.Cf) continue;
                float* dst = p.slab + (long)zs * p.slab_stride + (((long)tap * p.CsOut + p.cs_off + cs) * p.CfOut + cf);
                *(f32x4*)dst = acc[tap][ct][st];
            }
        }
}

template <int CFG> struct WgCfg;
template <> struct WgCfg<IGW_CFG_64x64> { static constexpr int WCF = 1, WCS = 4, CFT = 4, CST = 1, TMK = 128, KG = 2; };
template <> struct WgCfg<IGW_CFG_64x16> { static constexpr int WCF = 4, WCS = 1, CFT = 1, CST = 1, TMK = 128, KG = 2; };
// 128 F channels per workgroup, eight waves with their own 64 x 16 output blocks (all k-steps each): a third fewer staging
// bytes per MFMA than 64x64 (the S halo tile is shared by twice the F channels)
template <> struct WgCfg<IGW_CFG_128x64> { static constexpr int WCF = 2, WCS = 4, CFT = 4, CST = 1, TMK = 128, KG = 1; };

