// igemm_pp: third generation of the NHWC bf16 implicit-GEMM 3x3 convolution (forward and backward-data): the data layout, tile
// shapes and summation order of igemm_fwd2 (fragment-ordered weights, pixel-major swizzled halo tile, channel-permuted 16-byte
// stores; results are bit-identical) run as a PING-PONG between the two waves of every SIMD.
//
// What the measurements said (probes/probe_mfma_rate.hip, tools/pp_stamps.py): a SIMD's matrix pipe sustains 97-99 % of its
// issue slots when ONE wave feeds it back to back while its partner does everything else, but a single wave issues only about
// one instruction per 5 cycles -- LDS reads, scalar bookkeeping and LDS-DMA alike. igemm_fwd2 mixes all of that into every wave's
// MFMA stream (37 % of the pipe). Here the eight waves form two groups, G0 = waves 0-3 and G1 = waves 4-7 (waves w and w+4 share
// a SIMD), and time is cut into INTERVALS by workgroup barriers. A PHASE is one stage = the three taps of one kernel row of one
// 32-channel chunk: 3*(CT+PT) fragment reads and 3*CT*PT MFMAs per wave. Every wave runs
//        R(0) | M(0) | R(1) | M(1) | ...                  ('|' = barrier)
// where R(s) holds the LDS fragment reads of stage s and ALL bookkeeping (LDS-DMA prefetch, counted waits, tile changes, the
// epilogue of the finished tile) and M(s) is nothing but the stage's MFMAs (48 at the 64x64 wave tile: 768 pipe cycles, enough
// to cover the ~70 instructions of an R interval). G1 starts one interval late, so in every interval one wave of each SIMD
// multiplies while its partner reads and prefetches. A wave holds one stage of fragments and never reads while it multiplies.
//
// Both prefetch streams are spread evenly over the R intervals of BOTH groups: the LDS-DMA path of a CU moves ~29 bytes per clock
// from L2, and a wave that issues into a full queue stalls (a G1 wave issuing a whole stage of weights spent its whole R
// interval there, a G0 wave issuing a whole halo tile twice that). Every wave ends its R interval with s_waitcnt lgkmcnt(0):
// its fragments have landed before the barrier, so the slots it read may be overwritten from the NEXT interval on.
// In global intervals (G0 runs R(s) in interval 2s, M(s) in 2s+1; G1 one later):
//   * weights: ring of 3 stage slots, stage j of every chunk in slot j. In R(s) every wave issues its share of stage s+2 into the
//     slot of stage s-1 (last read in interval 2s-1). The pieces of stage s+1 (issued in R(s-1)) must be published by the barrier
//     closing interval 2s+1: G1 waits for its share at the end of R(s), G0 behind its MFMAs of M(s) -- counted vmcnt, the issues
//     of R(s) stay in flight.
//   * halo: ring of 2 chunk slots; the slot of chunk c-1 is free from interval 6c on. The pieces of chunk c+1 are issued in G0's
//     R(3c), R(3c+1), R(3c+2) and G1's R(3c), R(3c+1) (always in front of the interval's weight pieces) and are covered by the
//     same counted waits in interval 6c+5.
#include <cstring>
#include <type_traits>

#include "igemm.h"

// PP_DIL: the dilation this translation unit is built for (1: igemm_pp.hip itself; 2: igemm_pp_d2.hip, which includes this file with the
// exported names changed -- the dilated twin blocks of unet.py:32-39). The halo tile is 2 * DIL wider and higher, taps are DIL pixels apart.
#ifndef PP_TRS
#define PP_TRS 1      // developer A/B switch: backward-data's epilogue stores transposed across the lanes (see TRS below)
#endif
#ifndef PP_CLIP_PAD
#define PP_CLIP_PAD 1   // developer A/B switch: the padding pixels of a halo block are not fetched (setup_a)
#endif
#ifndef PP_TRS_LA
#define PP_TRS_LA 8    // ... with the exchanges of this many stores in flight ahead of the store being issued
#endif
#ifndef PP_TRS_FWD
#define PP_TRS_FWD 0   // developer A/B switch: the transposed stores in the epilogue without a mask too (forward; measured slower in round 4)
#endif
#ifndef PP_DIL
#define PP_DIL 1
#endif

#define RSU_WAIT_VMCNT(N) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory")
#define RSU_SENT 0x80000000u   // voffset that the range check always rejects (num_records = 0x7fffffff)

namespace {
__device__ __forceinline__ void bdma16(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned soff, void* lds_wave_base) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds_wave_base, 16, voff, soff, 0, 0);
}
__device__ __forceinline__ unsigned relu_pk_bf16(unsigned x) {
    typedef __attribute__((ext_vector_type(2))) short s2;
    s2 h = __builtin_bit_cast(s2, x);
    h = __builtin_elementwise_max(h, s2{0, 0});
    return __builtin_bit_cast(unsigned, h);
}
}  // namespace

// STAMP: diagnostic build (RSU_FWD_DBG bit 7 + RSU_STAMP_PTR): every wave notes s_memtime in front of and behind each barrier
// (in LDS, dumped to p.stamps at the end: [block][wave][PP_NSTAMP]); tools/pp_stamps.py prints the interval lengths. Never on the
// product path.
#define PP_NSTAMP 640
// The halo DMA pieces of a chunk are dealt over five (group, stage) slots G0/0, G1/0, G0/1, G1/1, G0/2 (NAW: pieces per wave in
// each slot; their sum * 4 * 16 = halo pixels at most); wave w4 of the slot's group issues piece (first(slot) + t) * 4 + w4.
// WP0 = weight pieces of a stage issued by each G0 wave (the G1 waves issue the rest)
namespace {
// (NAW packs the five per-slot counts, 4 bits each, slot 0 in the lowest nibble)
constexpr int pp_na_slot(int NAW, int k) { return (NAW >> (4 * k)) & 15; }
constexpr int pp_na(int NAW, int g, int j) { return g == 0 ? pp_na_slot(NAW, 2 * j) : (j < 2 ? pp_na_slot(NAW, 2 * j + 1) : 0); }
constexpr int pp_na_first(int NAW, int g, int j) {  // per-wave piece index of the slot's first piece
    int n = 0;
    for (int k = 0; k < (g == 0 ? 2 * j : 2 * j + 1); ++k) n += pp_na_slot(NAW, k);
    return n;
}
constexpr int pp_na_total(int NAW, int g) { return pp_na(NAW, g, 0) + pp_na(NAW, g, 1) + pp_na(NAW, g, 2); }
constexpr int pp_na_idx(int NAW, int g, int j) {  // index of the slot's first piece in the wave's own offset array
    int n = 0;
    for (int k = 0; k < j; ++k) n += pp_na(NAW, g, k);
    return n;
}
}  // namespace
// KS: the split-K instantiations (IgFwdParams::ksplit > 1; the others ignore it -- their reduction bookkeeping stays compile-time lean)
template <int WCO, int WPX, int CT, int PT, int LSW, int NAW, int WP0, bool STAMP, bool DBG, bool POOLK = false, bool KS = false, int DIL = PP_DIL>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2)))
igemm_pp_kernel(const IgFwdParams p) {
    constexpr int NW = 8, NG = 4, KW = 3, TPS = 3;
    static_assert(WCO * WPX == NW, "eight waves: two per SIMD");
    constexpr int TN = WCO * CT * 16, TM = WPX * PT * 16;
    constexpr int WT = TN / 16;
    constexpr int WBUF = TPS * WT * 1024;
    constexpr int NWB = 3, NAB = 2;
    // weight pieces of a stage (3 * WT): waves of G0 issue WP0 each (pieces q*4 + w4), waves of G1 WP1 each (pieces (WP0 + q)*4 + w4)
    constexpr int WP1 = TPS * WT / 4 - WP0;
    constexpr int WPM = WP1 > WP0 ? WP1 : (WP0 > 0 ? WP0 : 1);
    constexpr int NAV = pp_na_total(NAW, 0) > pp_na_total(NAW, 1) ? pp_na_total(NAW, 0) : pp_na_total(NAW, 1);
    static_assert((WP0 + WP1) * 4 == TPS * WT && WP0 >= 0 && WP1 >= 1, "weight pieces per stage");
    constexpr int NST = (CT / 2) * PT;   // epilogue buffer stores per wave per tile (always issued)
    static_assert((CT % 2) == 0, "bad config");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __attribute__((address_space(3))) char* lds = (__attribute__((address_space(3))) char*)smem;
    // geometry is a compile-time constant of the instantiation (strip width 2^LSW, 3x3 taps, dilation 1, stride 1): halo addresses
    // become immediate offsets of the LDS reads and most of the scalar arithmetic of an R interval disappears (an R-interval
    // instruction costs ~6 cycles of a budget of ~290: probes/probe_mfma_rate.hip)
    constexpr int SW = 1 << LSW, lsw = LSW, TR = TM >> LSW;
    constexpr int CW = (SW + 2 * DIL + 7) / 8 * 8;              // halo row pitch in pixels (plan_geo_aligned)
    constexpr int NPIX = ((TR + 2 * DIL) * CW + 31) / 32 * 32;  // halo pixels per slot
    constexpr int ABUF = NPIX * 64;
    constexpr int ROWB = DIL * CW * 64;                         // one kernel row (DIL image rows) down in the halo tile
    constexpr int a_base = 0;                             // LDS: [halo slot 0][halo slot 1][weight slots 0..2][scratch][bias]
    constexpr int WBASE = NAB * ABUF;
    constexpr int dummy_base = WBASE + NWB * WBUF;        // 1 KiB scratch slot for padding loads
    static_assert(ABUF + 2 * ROWB < 65536, "halo offsets must fit the 16-bit offset field of ds_read");

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int grp = wave >> 2, w4 = wave & 3;
    const int wco = wave / WPX, wpx = wave % WPX;
    const int g4 = lane >> 4, l15 = lane & 15;

    auto sgpr = [](int v) { return __builtin_amdgcn_readfirstlane(v); };
    // ---- this workgroup's tile list (XCD-aware numbering as in igemm_fwd2)
    int vid = blockIdx.x;
    {
        const int q = gridDim.x >> 3, r = gridDim.x & 7, x = vid & 7;
        vid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (vid >> 3);
    }
    // (channel block, reduction slice) units per pixel tile; `cgrp` of them on neighbouring ids, then the pixel tiles of the workgroup
    // column, then the next group of units: cgrp = nck is the numbering of rounds 1-3 (all units of a tile together: they share its halo),
    // cgrp = 1 keeps a weight slice inside one XCD's run
    const int ksplit = (KS && p.ksplit > 1) ? p.ksplit : 1;
    const int nck = p.ncob * ksplit;
    const int cgrp = (p.cob_group > 0 && p.cob_group < nck) ? p.cob_group : nck;
    const int tstride = sgpr(gridDim.x / nck);
    const int u_in = vid % cgrp, u_r = vid / cgrp;
    const int unit = sgpr((u_r / tstride) * cgrp + u_in);
    const int tile0 = sgpr(u_r % tstride);
    const int cob = sgpr(unit % p.ncob), ks = KS ? sgpr(unit / p.ncob) : 0;
    const int tpi = p.g.nstrips * p.g.tiles_per_strip;
    const int ntile_m = p.N * tpi;
    const int my_tiles = (unit < nck && tile0 < ntile_m) ? (ntile_m - tile0 + tstride - 1) / tstride : 0;
    if (my_tiles == 0) return;
    const int nchunks = p.nchunk[0] + p.nchunk[1] + p.nchunk[2];
    // this workgroup's slice of the reduction: chunks [c_lo, c_hi) of every one of its tiles (all of them without split-K)
    const int c_lo = KS ? sgpr(ks * nchunks / ksplit) : 0, c_hi = KS ? sgpr((ks + 1) * nchunks / ksplit) : nchunks;
    const int nloc = c_hi - c_lo;
    // the concat source that holds chunk c_lo, the chunk's index inside it, and the chunk index at which the next source begins
    int src_lo = 0, coff_lo = c_lo, next_lo = p.nchunk[0];
    if (KS && coff_lo >= p.nchunk[0] && p.nsrc > 1) {
        coff_lo -= p.nchunk[0];
        src_lo = 1;
        next_lo += p.nchunk[1];
        if (coff_lo >= p.nchunk[1] && p.nsrc > 2) {
            coff_lo -= p.nchunk[1];
            src_lo = 2;
            next_lo += p.nchunk[2];
        }
    }
    const int next_rel = next_lo - c_lo;   // ... counted from c_lo, as the prefetch stream counts
    const int GC = my_tiles * nloc;  // chunks in this workgroup's stream

    // this workgroup's tiles are tile0, tile0 + tstride, ...: (image, strip, row in strip) of the first one by division, then stepped --
    // a run-time division costs ~20 vector instructions, and a vector instruction of an R interval waits for a gap between the
    // partner wave's MFMAs (~17 cycles each, tools/pp_stamps.py)
    struct Tile { int n, x0, y0; };
    struct Pos { int n, strip, row; };
    auto split = [&](int t) {
        const int n = t / tpi, r = t - n * tpi;
        const int strip = r / p.g.tiles_per_strip;
        return Pos{sgpr(n), sgpr(strip), sgpr(r - strip * p.g.tiles_per_strip)};
    };
    const Pos tstep = split(tstride);
    auto advance = [&](Pos& q) {
        q.row += tstep.row;
        if (q.row >= p.g.tiles_per_strip) { q.row -= p.g.tiles_per_strip; ++q.strip; }
        q.strip += tstep.strip;
        if (q.strip >= p.g.nstrips) { q.strip -= p.g.nstrips; ++q.n; }
        q.n += tstep.n;
    };
    auto tile_at = [&](const Pos& q) { return Tile{q.n, q.strip * SW, q.row * TR}; };

    // ---- workgroup constants (per lane)
    int boff[PT][KW];   // byte offset (inside a halo slot) of this lane's 16-byte fragment piece, per pixel fragment and kx
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) {
        const int ml = (wpx * PT + pt) * 16 + l15;
        const int ty = ml >> lsw, tx = ml & (SW - 1);
        const int hp0 = ty * CW + tx;
#pragma unroll
        for (int kx = 0; kx < KW; ++kx) {
            const int hp = hp0 + kx * DIL;
            boff[pt][kx] = (hp << 6) + ((g4 ^ ((hp >> 1) & 2)) << 4);  // CW % 8 == 0: a ky shift keeps the swizzle
        }
    }
    const int afrag = WBASE + (wco * CT * 64 + lane) * 16;  // this lane's 16 bytes inside weight tile 0 of the wave (slot 0)
    constexpr int npieces = NPIX >> 4;
    const int lq = lane >> 2;
    auto mk = [&](const void* ptr) { return __builtin_amdgcn_make_buffer_rsrc((void*)ptr, 0, 0x7fffffff, 0x00020000); };

    // ---- weight prefetch stream (G1). Stage s of a tile is the contiguous block [s*3 .. s*3+3) x [all 16-row tiles] of the packed
    // weights; a wave's WPS pieces have constant offsets inside it (folded into the per-lane voffset), the stage is one scalar
    // offset that advances by a constant and rewinds per tile. The stream never ends: behind the last stage it simply starts the
    // tile's weights again (valid memory, slots nobody reads), so the counted waits hold to the very end.
    const int nstage_tile = nloc * 3;
    const unsigned stage_bytes = (unsigned)TPS * p.ntiles_w * 1024;
    const unsigned w_tile_soff = (unsigned)(p.tile_off + cob * WT) * 1024u + (unsigned)(c_lo * 3) * stage_bytes;
    unsigned w_soff = w_tile_soff;     // stage to prefetch next
    int w_sit = 0;                     // its stage index inside the tile
    unsigned w_voff[WPM];              // per-lane byte offset of this wave's piece q inside a stage block
#pragma unroll
    for (int q = 0; q < WPM; ++q) {
        const int i = ((grp ? WP0 : 0) + q) * NG + w4;
        const int tap_l = i / WT, tl = i - tap_l * WT;
        const bool real = (q < (grp ? WP1 : WP0)) && (p.tile_off + cob * WT + tl < p.ntiles_w);
        w_voff[q] = real ? (unsigned)((tap_l * p.ntiles_w + tl) * 1024 + lane * 16) : RSU_SENT;
    }
    auto issue_w = [&](auto gc_, int slot) {
        constexpr int G = decltype(gc_)::value;
        const __amdgpu_buffer_rsrc_t rw = mk(p.wp);
#pragma unroll
        for (int q = 0; q < (G ? WP1 : WP0); ++q) bdma16(rw, w_voff[q], w_soff, (void*)(lds + WBASE + slot * WBUF + (((G ? WP0 : 0) + q) * NG + w4) * 1024));
        if (++w_sit == nstage_tile) {
            w_sit = 0;
            w_soff = w_tile_soff;
        } else {
            w_soff += stage_bytes;
        }
    };
    // ---- halo prefetch stream: NAW * 4 pieces per chunk; clipped / padded pixels come back as zeros. Every wave keeps the stream
    // state (scalars) and the per-lane offsets of ITS pieces.
    unsigned a_voff[NAV > 0 ? NAV : 1];  // per-lane byte offset of this wave's piece inside the current source, or RSU_SENT
    const char* a_ptr = nullptr;   // current source, shifted back by the padding so that every in-window offset is >= 0
    unsigned a_soff = 0;           // byte offset of the (padded) halo origin + channel chunk in that source
    int a_crem = 0;                // channels left in the current source (>= 32 except in a partial last chunk)
    int a_cl = 0;                  // chunk (inside this workgroup's slice of its tile: 0 = chunk c_lo) of the next halo to prefetch
    int a_next_src = 0;            // chunk index at which the next source begins
    int a_si = src_lo;             // current source
    int ia_slot = 0;               // ring slot of the next halo
    auto my_piece = [&](int idx) {  // halo piece (per chunk) behind entry idx of this wave's offset array
        int pw = 0;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int i0 = grp ? pp_na_idx(NAW, 1, j) : pp_na_idx(NAW, 0, j);
            const int n = grp ? pp_na(NAW, 1, j) : pp_na(NAW, 0, j);
            const int f = grp ? pp_na_first(NAW, 1, j) : pp_na_first(NAW, 0, j);
            if (idx >= i0 && idx < i0 + n) pw = f + idx - i0;
        }
        return pw * NG + w4;
    };
    auto setup_a = [&](const Tile& T, int si, int coff) {   // source si from its chunk coff on
        const bf16_t* sptr = si == 0 ? p.src[0].ptr : (si == 1 ? p.src[1].ptr : p.src[2].ptr);
        const int sH = si == 0 ? p.src[0].H : (si == 1 ? p.src[1].H : p.src[2].H);
        const int sW = si == 0 ? p.src[0].W : (si == 1 ? p.src[1].W : p.src[2].W);
        const int sC = si == 0 ? p.src[0].C : (si == 1 ? p.src[1].C : p.src[2].C);
        const int soy = si == 0 ? p.src[0].oy : (si == 1 ? p.src[1].oy : p.src[2].oy);
        const int sox = si == 0 ? p.src[0].ox : (si == 1 ? p.src[1].ox : p.src[2].ox);
        a_ptr = (const char*)(sptr - ((long)p.pad * sW + p.pad) * sC);
        a_soff = (unsigned)((((long)(T.n * sH + T.y0 + soy) * sW + (T.x0 + sox)) * sC) * 2) + (unsigned)coff * 64u;
        a_crem = sC - coff * 32;
        const int iy0 = T.y0 - p.pad, ix0 = T.x0 - p.pad;
#pragma unroll
        for (int q = 0; q < NAV; ++q) {
            const int hp = my_piece(q) * 16 + lq;
            const int kg8 = ((lane & 3) ^ ((hp >> 1) & 2)) * 8;
            const int rr0 = hp / CW;   // compile-time divisor
            const int cc = hp - rr0 * CW;
            // (PP_CLIP_PAD: the columns that pad a halo row to CW pixels and the rows that pad the block to NPIX are never read by a
            // fragment -- boff stays below column SW + 2*DIL -- so their lanes ask for nothing: 15 % (strips of 32) to 28 % (strips of
            // 16) fewer requests in the halo stream. A per-lane constant folded into the row: the range test below rejects it.)
            const int rr = (!PP_CLIP_PAD || (cc < SW + 2 * DIL && rr0 < TR + 2 * DIL)) ? rr0 : 0x20000000;
            const bool ok = ((unsigned)(iy0 + rr) < (unsigned)p.Hin) && ((unsigned)(ix0 + cc) < (unsigned)p.Win) &&
                            (q < (grp ? pp_na_total(NAW, 1) : pp_na_total(NAW, 0)));
            a_voff[q] = ok ? (unsigned)(((rr * sW + cc) * sC + kg8) * 2) : RSU_SENT;
        }
    };
    Pos ppos = split(tile0);
    Tile ptile = tile_at(ppos);     // tile whose halo is being prefetched
    int pk = 0;                 // index of ptile in this workgroup's list
    bool a_started = false;
    auto a_begin = [&]() {  // the next chunk of the stream: moves on to the next source / tile where one ends
        if (a_cl == 0) {
            if (a_started) {
                if (pk + 1 < my_tiles) {  // behind the last tile the stream prefetches that tile again (valid memory, a slot nobody reads)
                    ++pk;
                    advance(ppos);
                }
                if (!(DBG && (p.dbg & 256))) {   // (dbg bit 8: timing without the prefetch stream's tile change)
                    ptile = tile_at(ppos);
                    setup_a(ptile, src_lo, coff_lo);
                }
                a_si = src_lo;
                a_next_src = next_rel;
            }
        } else if (a_cl == a_next_src) {
            ++a_si;
            a_next_src += a_si == 1 ? p.nchunk[1] : p.nchunk[2];
            setup_a(ptile, a_si, 0);
        }
        a_started = true;
    };
    auto a_pieces = [&](auto gc_, auto jc_) {  // this wave's pieces of slot (G, J)
        constexpr int G = decltype(gc_)::value, J = decltype(jc_)::value;
        constexpr int I0 = pp_na_idx(NAW, G, J), N = pp_na(NAW, G, J), F = pp_na_first(NAW, G, J);
        if constexpr (N > 0) {
            const __amdgpu_buffer_rsrc_t rs = mk(a_ptr);
            const int dst = a_base + ia_slot * ABUF;
            if (a_crem >= 32) {
#pragma unroll
                for (int t = 0; t < N; ++t) {
                    const int j = (F + t) * NG + w4;
                    bdma16(rs, a_voff[I0 + t], a_soff, (void*)(lds + (j < npieces ? dst + j * 1024 : dummy_base)));
                }
            } else {  // partial last chunk of a source whose channel count is not a multiple of 32: the missing channels read as zeros
#pragma unroll
                for (int t = 0; t < N; ++t) {
                    const int j = (F + t) * NG + w4;
                    const int hp = j * 16 + lq;
                    const int kg8 = ((lane & 3) ^ ((hp >> 1) & 2)) * 8;
                    bdma16(rs, kg8 < a_crem ? a_voff[I0 + t] : RSU_SENT, a_soff, (void*)(lds + (j < npieces ? dst + j * 1024 : dummy_base)));
                }
            }
        }
    };
    auto a_end = [&]() {
        ia_slot ^= 1;
        a_soff += 64;
        a_crem -= 32;
        a_cl = a_cl + 1 == nloc ? 0 : a_cl + 1;
    };

    // bias of this workgroup's TN channels lives in LDS (behind the scratch slot)
    const int bias_base = dummy_base + 1024;
    const int stamp_base = bias_base + 512;
    int stamp_i = 0;
    // STAMP builds also sum the cycles of named segments of the R intervals (reported in the last stamps of the wave)
    unsigned long long seg_sum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, seg_t0 = 0;
    auto seg_begin = [&]() {
        if constexpr (STAMP) seg_t0 = __builtin_amdgcn_s_memtime();
    };
    auto seg_end = [&](int k) {
        if constexpr (STAMP) {
            const unsigned long long t = __builtin_amdgcn_s_memtime();
            seg_sum[k] += t - seg_t0;
            seg_t0 = t;
        }
    };
    auto stamp = [&]() {
        if constexpr (STAMP) {
            if (stamp_i < PP_NSTAMP - 10) {
                const unsigned t = (unsigned)__builtin_amdgcn_s_memtime();
                if (lane == 0) *(__attribute__((address_space(3))) unsigned*)(lds + stamp_base + (wave * PP_NSTAMP + stamp_i) * 4) = t;
            }
            ++stamp_i;
        }
    };
    if (threadIdx.x < TN) {
        const int co = cob * TN + threadIdx.x;
        const float bvv = (p.bias && co < p.Cout && ks == 0) ? p.bias[co] : 0.f;   // (split-K: the bias rides in slice 0's partial sums)
        *(__attribute__((address_space(3))) float*)(lds + bias_base + threadIdx.x * 4) = bvv;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // retire the ordinary loads before the LDS-DMA stream starts

    // The epilogue of backward-data (the one with the ReLU mask) stores TRANSPOSED across the lanes (TRS): the MFMA leaves lane (g4, l15) with
    // pixel l15 of a fragment and the 16 bytes g4 of that pixel's 64-byte channel group -- stored as they lie, neighbouring lanes write 16
    // bytes each one pixel (outC * 2 bytes) apart and the address unit, which merges the 4 x 16 bytes of a lane quad into one 64-byte
    // request, gets nothing to merge: 64 requests per store instead of 16 (probes/probe_store_shapes.hip: a tile's burst of stores costs
    // ~3200 cycles in this shape, ~1100 with quads of 64 contiguous bytes). Four ds_bpermute_b32 per store hand lane L the words of lane
    // (L & 3) * 16 + (L >> 2): lane L then covers pixel L >> 2, bytes 16 * (L & 3); the mask is loaded in the same layout (coalesced as
    // well) and applied behind the exchange. Measured per layer at fixed tile shapes (tools/pp_fixed.py, profiles/r04/epilogue_variants.txt):
    // backward-data -2 ... -8 %; the forward epilogue (no mask to wait for, its stores spread between the conversions) loses 2-4 % to the
    // exchanges and keeps the MFMA layout, as does the pooling epilogue (its 2x2 windows pair l15 neighbours).
    // scalar byte offset of a tile's first output pixel (+ this wave's first channel) and per-lane offset of epilogue store e
    auto out_sbase = [&](const Tile& T) {
        return (unsigned)((((long)(T.n * p.oH + T.y0) * p.oW + T.x0) * p.outC + cob * TN + wco * (CT / 2) * 32) * 2);
    };
    auto out_voff = [&](const Tile& T, int e, const int ql15, const int qg4) {   // (ql15, qg4): pixel and 16-byte group this lane stores
        const int pt = e / (CT / 2), pp = e % (CT / 2);
        const int ml = (wpx * PT + pt) * 16 + ql15;
        const int ty = ml >> lsw, tx = ml & (SW - 1);
        const bool pok = (T.y0 + ty < p.Ho) && (T.x0 + tx < p.Wo);
        const int ovoff_pt = ((ty * p.oW + tx) * p.outC + 8 * qg4) * 2;
        const int co = cob * TN + (wco * (CT / 2) + pp) * 32 + 8 * qg4;
        return (pok && co < p.Cout) ? (unsigned)(ovoff_pt + pp * 64) : RSU_SENT;
    };
// store + its wait states as ONE asm statement (the >64-bit store / VALU-write hazard). The leading s_nop 4:
// under register pressure the compiler parks descriptor words in VGPR lanes and brings them back through v_readlane right in front
// of the store -- a VALU write of an SGPR that a VMEM instruction reads needs five wait states, which hipcc does not add for asm
#define PP_STORE(R, VOFF) \
    asm volatile("s_nop 4\n\tbuffer_store_dwordx4 %0, %1, %2, %3 offen\n\ts_nop 3" ::"v"(R), "v"(VOFF), "s"(orsrc), "s"(sbase) : "memory")
#define PP_STORE64(R, VOFF) \
    asm volatile("s_nop 4\n\tbuffer_store_dwordx4 %0, %1, %2, %3 offen offset:64\n\ts_nop 3" ::"v"(R), "v"(VOFF), "s"(orsrc), "s"(sbase) : "memory")
    // tiles that lie inside the output (all but the last row / strip of an image) store through per-lane offsets computed once
    unsigned ovoff[PT];
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) {
        const int ml = (wpx * PT + pt) * 16 + l15;
        const int ty = ml >> lsw, tx = ml & (SW - 1);
        ovoff[pt] = (unsigned)(((ty * p.oW + tx) * p.outC + 8 * g4) * 2);
    }
    const bool co_inside = cob * TN + TN <= p.Cout;
    // ---- 2x2 max-pool folded into the epilogue (unet.py:44-52: conv2 + ReLU, then max_pooling2d). A wave's PT fragments are PT * 16
    // consecutive pixels of the tile = whole rows of the strip when the strip is 16 or 32 wide: the two rows of a 2x2 window are two
    // fragments of the SAME wave (PT = 4: fragments (0,2),(1,3) at width 32, (0,1),(2,3) at width 16; PT = 2, width 16: (0,1)), its two
    // columns neighbouring lanes. On the packed, post-ReLU bf16 words (non-negative: integer order = float order) the vertical max is
    // one v_pk_max_i16 per dword, the horizontal one a DPP lane swap + max; the even lanes store the pooled pixel (16 bytes) and its
    // code bytes (k_maxpool_fwd's: bits 0-3 window element > 0, bits 4-5 first maximum in row-major order).
    // (POOLK: the pooling epilogue lives in instantiations of its own -- inside the others it cost the main loop 5-6 spilled registers)
    constexpr bool POOL_OK = POOLK && ((PT == 4 && (LSW == 4 || LSW == 5)) || (PT == 2 && LSW == 4));
    auto pool_body = [&](const Tile& T, f32x4(&acc)[CT][PT], const unsigned (&voffs)[NST], const __amdgpu_buffer_rsrc_t orsrc, const unsigned sbase)
                         __attribute__((always_inline)) {
        typedef __attribute__((ext_vector_type(2))) short s2;
        typedef __attribute__((ext_vector_type(2))) unsigned short us2;
        constexpr int NPAIR = PT / 2, PSTEP = (LSW == 5) ? 2 : 1;   // fragment pa's row partner is fragment pa + PSTEP
        const int pH = p.oH >> 1, pW = p.oW >> 1;
        const __amdgpu_buffer_rsrc_t prsrc = mk(p.pool_out);
        const __amdgpu_buffer_rsrc_t crsrc = mk(p.pool_code ? (const void*)p.pool_code : (const void*)p.pool_out);
        const unsigned psbase = (unsigned)((((long)(T.n * pH + (T.y0 >> 1)) * pW + (T.x0 >> 1)) * p.outC + cob * TN + wco * (CT / 2) * 32) * 2);
        const short fl = p.relu ? (short)0 : (short)-32768;
        const s2 floor2 = {fl, fl};
        auto pk = [&](int ct2, int pt) {
            u32x4 r;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const unsigned lo = pack_bf2(acc[ct2][pt][2 * i], acc[ct2][pt][2 * i + 1]);
                const unsigned hi = pack_bf2(acc[ct2 + 1][pt][2 * i], acc[ct2 + 1][pt][2 * i + 1]);
                r[i] = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s2, lo), floor2));
                r[2 + i] = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s2, hi), floor2));
            }
            return r;
        };
        auto swap1 = [](unsigned v) { return (unsigned)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xF, 0xF, true); };   // quad_perm [1,0,3,2]: lane ^ 1
        auto umin1 = [](unsigned v) {   // per 16-bit half: 1 where the half is non-zero
            const us2 one = {1, 1};
            return __builtin_bit_cast(unsigned, __builtin_elementwise_min(__builtin_bit_cast(us2, v), one));
        };
        auto sub16 = [](unsigned a, unsigned b) { return __builtin_bit_cast(unsigned, (us2)(__builtin_bit_cast(us2, a) - __builtin_bit_cast(us2, b))); };
        auto mul16 = [](unsigned a, unsigned b) { return __builtin_bit_cast(unsigned, (us2)(__builtin_bit_cast(us2, a) * __builtin_bit_cast(us2, b))); };
        auto add16 = [](unsigned a, unsigned b) { return __builtin_bit_cast(unsigned, (us2)(__builtin_bit_cast(us2, a) + __builtin_bit_cast(us2, b))); };
#pragma unroll
        for (int q = 0; q < NPAIR; ++q) {
            const int pa = (LSW == 5) ? q : 2 * q, pb = pa + PSTEP;
            // pooled pixel of this lane's window (even lanes): row ty / 2, column tx / 2 of the tile's pooled block
            const int ml = (wpx * PT + pa) * 16 + l15;
            const int ty = ml >> lsw, tx = ml & (SW - 1);
            const bool pok = ((l15 & 1) == 0) && (T.y0 + ty < p.Ho) && (T.x0 + tx < p.Wo);
            const unsigned pvoff = (unsigned)((((ty >> 1) * pW + (tx >> 1)) * p.outC + 8 * g4) * 2);
#pragma unroll
            for (int pp = 0; pp < CT / 2; ++pp) {
                const u32x4 ra = pk(2 * pp, pa), rb = pk(2 * pp, pb);
                PP_STORE(ra, voffs[pa * (CT / 2) + pp]);
                PP_STORE(rb, voffs[pb * (CT / 2) + pp]);
                u32x4 pooled;
                unsigned cw[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const unsigned v00 = ra[i], v10 = rb[i], v01 = swap1(ra[i]), v11 = swap1(rb[i]);
                    const unsigned mv = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s2, v00), __builtin_bit_cast(s2, v10)));
                    const unsigned mh = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s2, v01), __builtin_bit_cast(s2, v11)));
                    const unsigned mx = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s2, mv), __builtin_bit_cast(s2, mh)));
                    pooled[i] = mx;
                    // code: (element > 0) bits, then the first element that equals the maximum: ne_k = (mx - v_k != 0)
                    const unsigned g = umin1(v00) | (umin1(v01) << 1) | (umin1(v10) << 2) | (umin1(v11) << 3);
                    const unsigned ne0 = umin1(sub16(mx, v00)), ne1 = umin1(sub16(mx, v01)), ne2 = umin1(sub16(mx, v10));
                    const unsigned idx = mul16(ne0, add16(0x00010001u, mul16(ne1, add16(0x00010001u, ne2))));
                    cw[i] = g | (idx << 4);
                }
                const bool cok = cob * TN + (wco * (CT / 2) + pp) * 32 + 8 * g4 < p.Cout;
                const unsigned pv = (pok && cok) ? pvoff + pp * 64 : RSU_SENT;
                asm volatile("s_nop 4\n\tbuffer_store_dwordx4 %0, %1, %2, %3 offen\n\ts_nop 3" ::"v"(pooled), "v"(pv), "s"(prsrc), "s"(psbase) : "memory");
                if (p.pool_code) {
                    u32x2 cd = {__builtin_amdgcn_perm(cw[1], cw[0], 0x06040200u), __builtin_amdgcn_perm(cw[3], cw[2], 0x06040200u)};
                    const unsigned cv = (pok && cok) ? (pvoff + pp * 64) >> 1 : RSU_SENT;
                    asm volatile("s_nop 4\n\tbuffer_store_dwordx2 %0, %1, %2, %3 offen\n\ts_nop 3" ::"v"(cd), "v"(cv), "s"(crsrc), "s"(psbase >> 1) : "memory");
                }
            }
        }
    };
    auto epilogue = [&](const Tile& T, const int tidx, f32x4(&acc)[CT][PT]) {
        if constexpr (KS) {   // split-K: the slice's fp32 partial sums in register order, 1 KiB per store; k_pp_splitk_finish sums them
            const unsigned long long kaddr = (unsigned long long)(p.kslab + (long)ks * p.kslab_stride);
            const __amdgpu_buffer_rsrc_t krsrc = mk((const void*)(((unsigned long long)(unsigned)sgpr((int)(kaddr >> 32)) << 32) | (unsigned)sgpr((int)kaddr)));
            const unsigned kbase = (unsigned)sgpr((((tidx * p.ncob + cob) * NW + wave) * NST) * 2048);
            const unsigned kvoff = (unsigned)lane * 16u;
#pragma unroll
            for (int e = 0; e < NST; ++e) {
                const int pt = e / (CT / 2), pp = e % (CT / 2);
                asm volatile("s_nop 4\n\tbuffer_store_dwordx4 %0, %1, %2, %3 offen\n\ts_nop 3" ::"v"(acc[2 * pp][pt]), "v"(kvoff), "s"(krsrc),
                             "s"(kbase + (unsigned)(e * 2048)) : "memory");
                asm volatile("s_nop 4\n\tbuffer_store_dwordx4 %0, %1, %2, %3 offen\n\ts_nop 3" ::"v"(acc[2 * pp + 1][pt]), "v"(kvoff), "s"(krsrc),
                             "s"(kbase + (unsigned)(e * 2048 + 1024)) : "memory");
            }
            return;
        }
        const __amdgpu_buffer_rsrc_t orsrc = mk(p.out);
        const __amdgpu_buffer_rsrc_t mrsrc = mk(p.mask_src ? (const void*)p.mask_src : (const void*)p.out);
        const unsigned sbase = out_sbase(T);
        const bool inside = co_inside && (T.y0 + TR <= p.Ho) && (T.x0 + SW <= p.Wo);  // wave-uniform
        {
            // forward and backward-data without an AddN: everything on the packed bf16 result. ReLU is a packed int16 max against 0
            // (or, switched off, against the most negative int16: bf16 sign bit == int16 sign bit); the ReLU mask of backward-data
            // (relu_src > 0) is a packed 0 / 0xffff word ANDed onto it: max(x, 0) -> min(., 1) -> 0 - . (a positive NaN in relu_src
            // counts as > 0 here; TensorFlow's ReluGrad lets nothing through there -- no finite activation is affected).
            // One straight-line instance with and one without the mask, chosen by one wave-uniform branch up front: tested per store,
            // the conditions (mask? inside? 1-bit masks?) cost the R interval that holds the epilogue ~2000 cycles in branches alone
            // (tools/pp_stamps.py; the partner group, done with its MFMAs, waits at the barrier).
            typedef __attribute__((ext_vector_type(2))) short s2;
            const short fl = p.relu ? (short)0 : (short)-32768;
            const s2 floor2 = {fl, fl};
            unsigned voffs[NST];
            auto fill_voffs = [&](const bool tr) __attribute__((always_inline)) {   // tr: the transposed lane layout (computed here, no registers of the loop)
                const int ql15 = tr ? (lane >> 2) : l15, qg4 = tr ? (lane & 3) : g4;
                if (inside) {   // (wave-uniform)
#pragma unroll
                    for (int e = 0; e < NST; ++e) {
                        const int pt = e / (CT / 2);
                        unsigned o = ovoff[pt];
                        if (tr) {
                            const int ml = (wpx * PT + pt) * 16 + ql15;
                            o = (unsigned)((((ml >> lsw) * p.oW + (ml & (SW - 1))) * p.outC + 8 * qg4) * 2);
                        }
                        voffs[e] = o + (e % (CT / 2)) * 64;
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < NST; ++e) voffs[e] = out_voff(T, e, ql15, qg4);
                }
            };
            auto body = [&](const bool MASK) __attribute__((always_inline)) {   // (called with a constant: two straight-line copies)
                const bool TRS = PP_TRS && (MASK || PP_TRS_FWD);
                const int tr_src = (((lane & 3) << 4) | (lane >> 2)) << 2;   // ds_bpermute byte index of the lane whose words this lane stores
                fill_voffs(TRS);
                seg_end(6);
                unsigned ones_pk = 0x00010001u;
                asm volatile("" : "+v"(ones_pk));
                u32x4 mk4[NST];
                if (MASK) {
                    // every mask load of the tile in flight at once (the stage fragments are dead by now), then ONE wait. The loads are
                    // asm like the stores: hipcc counts only the memory instructions it sees, and its own vmcnt(7 - e) in front of use e
                    // would also wait for the e stores issued meanwhile -- store acknowledgements, thousands of cycles
#pragma unroll
                    for (int e = 0; e < NST; ++e)
                        asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=&v"(mk4[e]) : "v"(voffs[e]), "s"(mrsrc), "s"(sbase) : "memory");
                }
                // The packed results of LA stores ahead, their (TRS) lane exchanges in flight: a ds_bpermute_b32 comes back after ~150 cycles,
                // four of them waited for in front of each store would cost the interval 8 x that. The exchanges are asm with counted
                // waits of our own (LDS operations return in order; the "+v" of the wait statement keeps the uses behind it)
                const int LA = TRS ? (NST < PP_TRS_LA ? NST : PP_TRS_LA) : 1;   // (without exchanges: convert, store, convert, store ... as before)
                u32x4 rr[NST];
                auto pack = [&](const int e) __attribute__((always_inline)) {
                    const int pt = e / (CT / 2), pp = e % (CT / 2);
                    unsigned q[4];
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        const unsigned lo = pack_bf2(acc[2 * pp][pt][2 * i], acc[2 * pp][pt][2 * i + 1]);
                        const unsigned hi = pack_bf2(acc[2 * pp + 1][pt][2 * i], acc[2 * pp + 1][pt][2 * i + 1]);
                        q[i] = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s2, lo), floor2));
                        q[2 + i] = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s2, hi), floor2));
                    }
                    if (TRS) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) asm volatile("ds_bpermute_b32 %0, %1, %2" : "=v"(q[i]) : "v"(tr_src), "v"(q[i]));
                    }
                    rr[e] = u32x4{q[0], q[1], q[2], q[3]};
                };
#pragma unroll
                for (int e = 0; e < NST; ++e) {
                    if (e == 0) {
#pragma unroll
                        for (int k = 0; k < LA; ++k) pack(k);
                        if (MASK) {
                            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
                            for (int k = 0; k < NST; ++k) asm volatile("" : "+v"(mk4[k]));
                        }
                    } else if (e + LA - 1 < NST) {
                        pack(e + LA - 1);
                    }
                    u32x4 r = rr[e];
                    if (TRS) {   // the exchanges of stores 0..e are back (lgkmcnt is a 4-bit counter)
                        const int ahead = (e + LA < NST ? e + LA : NST) - e - 1;
                        asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(r) : "n"(4 * ahead < 15 ? 4 * ahead : 15));
                    }
                    if (MASK) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) r[i] &= pos_mask_pk_bf16(mk4[e][i], ones_pk);
                    }
                    if (DBG && (p.dbg & 512)) {   // (dbg bit 9: timing of the epilogue without its stores)
                        asm volatile("" ::"v"(r));
                    } else {
                        PP_STORE(r, voffs[e]);
                    }
                }
            };
            if constexpr (POOL_OK) {
                if (p.pool_out) {   // (wave-uniform) forward conv2 of an encoder level: y, the 2x2 max-pool of y and its code bytes
                    fill_voffs(false);
                    pool_body(T, acc, voffs, orsrc, sbase);
                    return;
                }
            }
            if (p.mask_src) body(true); else body(false);
            return;
        }
        // (launches with an AddN -- backward-data into a tensor that already holds a gradient -- stay with igemm_fwd2: igemm_pp_supports)
    };
    auto bar = [&]() {
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };

    // ---- prologue: every wave issues its share of the halo of chunk 0 and of the weights of stages 0 and 1 and waits for all of
    // it; one barrier publishes the lot (raw barrier: __syncthreads() would drain the LDS-DMA stream on every later use; the bias
    // words above are the only ordinary LDS stores)
    setup_a(ptile, src_lo, coff_lo);
    a_next_src = next_rel;
    a_begin();
    if (!(DBG && (p.dbg & 2))) {
        if (grp) {
            a_pieces(std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{});
            a_pieces(std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{});
        } else {
            a_pieces(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
            a_pieces(std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{});
            a_pieces(std::integral_constant<int, 0>{}, std::integral_constant<int, 2>{});
        }
    }
    a_end();
    if (!(DBG && (p.dbg & 1))) {
        if (grp) {
            issue_w(std::integral_constant<int, 1>{}, 0);
            issue_w(std::integral_constant<int, 1>{}, 1);
        } else {
            issue_w(std::integral_constant<int, 0>{}, 0);
            issue_w(std::integral_constant<int, 0>{}, 1);
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    bar();
    if (grp) bar();  // G1 sits out interval 0

    unsigned long long clk0 = 0, rt0 = 0;
    if constexpr (STAMP) {
        clk0 = __builtin_amdgcn_s_memtime();
        rt0 = __builtin_amdgcn_s_memrealtime();
    }
    // ---- the chunk stream, one copy of the loop per wave group (no group tests inside), two chunks per iteration (the halo slot of a
    // chunk is its parity: every LDS address of an R interval is a per-lane constant plus an immediate). Accumulators live across
    // the loop; a tile starts them at the bias and ends with its epilogue, which opens the wave's next R interval.
    f32x4 acc[CT][PT];
    auto run_stream = [&](auto gconst) {
        constexpr int G = decltype(gconst)::value;
        int c = 0;               // chunk inside the current tile
        int ck = 0;              // current tile
        Pos cpos = split(tile0);
        Tile ctile = tile_at(cpos);
        // one stage of fragments per wave; they live across a tile boundary in G0 (whose first R interval of the next tile runs BEFORE
        // the finished tile's epilogue, see below)
        bf16x8 fa[TPS][CT], fb[TPS][PT];
        auto bias_init = [&]() {
            if (!(DBG && (p.dbg & 16))) {   // (dbg bit 4: timing without the bias initialisation)
                // every accumulator straight from the bias words in LDS: 16 LDS reads instead of 4 reads + 60 register moves (a vector
                // instruction of an R interval waits for a gap between the partner's MFMAs; an LDS read does not). The reads land before
                // the s_waitcnt lgkmcnt(0) that ends the interval.
                const unsigned baddr = (unsigned)(bias_base + (wco * (CT / 2) * 32 + 8 * g4) * 4);
#pragma unroll
                for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                    for (int pt = 0; pt < PT; ++pt)
                        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(acc[ct][pt]) : "v"(baddr), "n"(((ct >> 1) * 32 + (ct & 1) * 4) * 4) : "memory");
            }
        };
        // ================= R interval of stage J of a chunk whose halo sits in slot PAR: the stage's fragment reads (r_reads) + the
        // bookkeeping for the stages ahead (r_issue). after_epi: the NST stores of the previous tile sit in front of this interval's
        // issues (G1 only)
        auto r_reads = [&](auto parc, auto jc) {
            constexpr int PAR = decltype(parc)::value, J = decltype(jc)::value;
            seg_begin();
            if (!(DBG && (p.dbg & 32))) {  // (dbg bit 5: timing experiment without fragment reads and MFMAs -- the LDS-DMA streams alone)
#pragma unroll
                for (int tl = 0; tl < TPS; ++tl) {
#pragma unroll
                    for (int ct = 0; ct < CT; ++ct)
                        fa[tl][ct] = *(const __attribute__((address_space(3))) bf16x8*)(lds + afrag + J * WBUF + (tl * WT + ct) * 1024);
#pragma unroll
                    for (int pt = 0; pt < PT; ++pt)
                        fb[tl][pt] = *(const __attribute__((address_space(3))) bf16x8*)(lds + boff[pt][tl] + (a_base + PAR * ABUF + J * ROWB));
                }
            }
            seg_end(3);
        };
        auto r_issue = [&](auto jc, const bool after_epi) {
            constexpr int J = decltype(jc)::value;
            // prefetch issues of this interval: halo pieces of the next chunk first, then this wave's share of stage s+2
            if constexpr (pp_na(NAW, G, J) > 0 || J == 0) {
                if (!(DBG && (p.dbg & 2))) {
                    if constexpr (J == 0) {
                        a_begin();
                        seg_end(7);
                    }
                    a_pieces(gconst, jc);
                }
            }
            if constexpr (J == 2) a_end();
            if constexpr ((G ? WP1 : WP0) > 0) {
                if (!(DBG && (p.dbg & 1))) issue_w(gconst, (J + 2) % NWB);
            }
            seg_end(4);
            if constexpr (G == 1) {
                // G1's share of the next stage (issued one phase ago) must be in LDS behind this interval's barrier
                if (J == 0 && after_epi) {
                    RSU_WAIT_VMCNT(pp_na(NAW, 1, J) + WP1 + NST);
                } else {
                    RSU_WAIT_VMCNT(pp_na(NAW, 1, J) + WP1);
                }
            }
        };
        auto r_part = [&](auto parc, auto jc, const bool after_epi) {
            r_reads(parc, jc);
            r_issue(jc, after_epi);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this wave's fragments are in registers: the slots they came from may be refilled
            seg_end(5);
        };
        // ================= M interval: the MFMAs of the stage, nothing else
        auto m_part = [&](auto jc, const bool after_epi, const bool last_of_tile) {
            constexpr int J = decltype(jc)::value;
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(1);
            if (!(DBG && (p.dbg & 32)))
#pragma unroll
            for (int tl = 0; tl < TPS; ++tl)
#pragma unroll
                for (int pt = 0; pt < PT; ++pt)
#pragma unroll
                    for (int ct = 0; ct < CT; ++ct) mfma_bf16_inplace(acc[ct][pt], fa[tl][ct], fb[tl][pt]);
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (G == 0) {  // G0's share of the next stage (and, in stage 2, of the next halo) must be in LDS behind this interval's barrier
                // (J == 0 behind a tile boundary: G0 issued R(0)'s pieces BEFORE the finished tile's NST stores -- the same NST + pieces may stay in flight)
                if (J == 0 && after_epi) {
                    RSU_WAIT_VMCNT(pp_na(NAW, 0, J) + WP0 + NST);
                } else if (J == 2) {
                    RSU_WAIT_VMCNT(WP0);
                } else {
                    RSU_WAIT_VMCNT(pp_na(NAW, 0, J) + WP0);
                }
            }
            if (J == 2 && last_of_tile) mfma_results_fence();  // straight behind the tile's last MFMA
        };
        auto chunk = [&](auto parc, int gc) {
            constexpr int PAR = decltype(parc)::value;
            const bool after_epi = (c == 0) && gc > 0;  // this chunk opens a tile that is not the workgroup's first
            const bool last_of_tile = c == nloc - 1;
            // phase 0. G0 has run R(0) of a tile that is not its first already (in front of the previous tile's epilogue, below)
            if (!(G == 0 && after_epi)) {
                seg_begin();
                if (c == 0) bias_init();
                seg_end(2);
                r_part(parc, std::integral_constant<int, 0>{}, after_epi);
                stamp();
                bar();
                stamp();
            }
            m_part(std::integral_constant<int, 0>{}, after_epi, last_of_tile);
            stamp();
            bar();
            stamp();
            r_part(parc, std::integral_constant<int, 1>{}, false);
            stamp();
            bar();
            stamp();
            m_part(std::integral_constant<int, 1>{}, false, last_of_tile);
            stamp();
            bar();
            stamp();
            r_part(parc, std::integral_constant<int, 2>{}, false);
            stamp();
            bar();
            stamp();
            m_part(std::integral_constant<int, 2>{}, false, last_of_tile);
            stamp();
            bar();
            stamp();
            if (last_of_tile) {
                // Tile boundary. Both groups store the finished tile in ONE interval of their own, with no MFMAs beside it: a vector
                // instruction next to the partner's MFMA stream waits ~17 cycles for an issue gap (tools/pp_stamps.py), so an epilogue
                // folded into an R interval made that interval ~2500 cycles longer than the partner's MFMAs. In global intervals, with S
                // stages per tile (G0 runs M(S-1) in interval 2S-1, G1 in 2S):
                //     interval 2S     G1: M(S-1)           G0: the prefetch issues of R(0) of its NEXT tile (what they overwrite was last read
                //                                              in interval 2S-1)
                //     interval 2S+1   both: epilogue; G0 then starts its accumulators at the bias and reads the fragments of stage 0 (published
                //                                              by the barrier closing 2S-1), with no MFMAs beside it
                //     interval 2S+2   G0: M(0)             G1: R(0)
                // -- ONE interval without MFMAs per tile. (Rounds 2-3 ran G0's whole R(0) behind the epilogue in an interval of its own, G1
                // sitting it out: two idle intervals per tile. Reading G0's fragments in interval 2S as well keeps 96 more registers live
                // across the epilogue: ~100 spilled registers in the four-fragment shapes.)
                const bool more = ck + 1 < my_tiles;
                if constexpr (G == 0) {
                    if (more) r_issue(std::integral_constant<int, 0>{}, false);
                    stamp();
                    bar();
                    stamp();
                }
                seg_begin();
                if (!(DBG && (p.dbg & 8))) epilogue(ctile, KS ? (cpos.n * p.g.nstrips + cpos.strip) * p.g.tiles_per_strip + cpos.row : 0, acc);
                seg_end(0);
                if constexpr (G == 0) {
                    if (more) {
                        bias_init();
                        if constexpr (PAR == 0) r_reads(std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{});
                        else r_reads(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    } else {
                        // (no tile follows: the fragment registers hold no value the loop could still use -- said explicitly, or the compiler
                        // keeps the previous stage's 96 registers alive across the epilogue for a next iteration that never comes)
#pragma unroll
                        for (int tl = 0; tl < TPS; ++tl) {
#pragma unroll
                            for (int ct = 0; ct < CT; ++ct) fa[tl][ct] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
                            for (int pt = 0; pt < PT; ++pt) fb[tl][pt] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
                        }
                    }
                }
                stamp();
                bar();
                stamp();
                c = 0;
                ++ck;
                if (ck < my_tiles && !(DBG && (p.dbg & 64))) {   // (dbg bit 6: timing without the tile change)
                    advance(cpos);
                    ctile = tile_at(cpos);
                }
                seg_end(1);
            } else {
                ++c;
            }
        };
        for (int gc = 0; gc < GC; gc += 2) {
            chunk(std::integral_constant<int, 0>{}, gc);
            if (gc + 1 >= GC) break;
            chunk(std::integral_constant<int, 1>{}, gc + 1);
        }
    };
    if (grp) run_stream(std::integral_constant<int, 1>{}); else run_stream(std::integral_constant<int, 0>{});
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // nothing may land in this workgroup's LDS after it has gone
    if constexpr (STAMP) {
        if (p.stamps) {
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            const unsigned dclk = (unsigned)(__builtin_amdgcn_s_memtime() - clk0), drt = (unsigned)(__builtin_amdgcn_s_memrealtime() - rt0);
            for (int i = lane; i < PP_NSTAMP; i += 64)
                p.stamps[((long)blockIdx.x * NW + wave) * PP_NSTAMP + i] =
                    i == PP_NSTAMP - 2 ? dclk : (i == PP_NSTAMP - 1 ? drt :  // shader cycles and 100-MHz ticks of the main loop
                    (i >= PP_NSTAMP - 10 && i < PP_NSTAMP - 2) ? (unsigned)seg_sum[i - (PP_NSTAMP - 10)] :
                    (i < stamp_i ? *(__attribute__((address_space(3))) unsigned*)(lds + stamp_base + (wave * PP_NSTAMP + i) * 4) : 0u));
        }
    }
}

#if PP_DIL == 1
// ---------------------------------------------------------------------------------------------
// split-K finish: out[pixel][8 channels] = bf16(epilogue(sum over the slices, in slice order, of the partial sums)) -- one thread per
// (tile slot, wave, store e, lane) of the conv kernel's epilogue, i.e. per 16-byte output store: it reads its two float4 of every slice
// (1 KiB per wavefront and slice, contiguous), sums them in a fixed order (deterministic), applies ReLU (or not) and the ReLU mask of
// backward-data exactly as the conv epilogue does (packed bf16), and stores where that epilogue would have stored.
struct PpFinishParams {
    const float* slab;
    long slice_stride;     // floats
    int ksplit, nslots, ncob;
    int WCO, WPX, CT, PT, lsw, TR;
    int nstrips, tiles_per_strip;
    bf16_t* out;
    const bf16_t* mask_src;
    int Ho, Wo, oH, oW, outC, Cout, relu;
};
__global__ void __launch_bounds__(256) k_pp_splitk_finish(const PpFinishParams q) {
    const int NST = (q.CT / 2) * q.PT, TN = q.WCO * q.CT * 16;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    const int lane = (int)(idx & 63);
    long r = idx >> 6;
    const int e = (int)(r % NST); r /= NST;
    const int wave = (int)(r & 7); r >>= 3;
    const int slot = (int)r;
    if (slot >= q.nslots) return;
    const float* src = q.slab + ((long)(slot * 8 + wave) * NST + e) * 512 + lane * 4;   // 2 KiB = 512 floats per (wave, e): two 1-KiB halves
    f32x4 a = *(const f32x4*)src, b = *(const f32x4*)(src + 256);
    for (int z = 1; z < q.ksplit; ++z) {
        const float* sz = src + (long)z * q.slice_stride;
        a += *(const f32x4*)sz;
        b += *(const f32x4*)(sz + 256);
    }
    const int cob = slot % q.ncob, t = slot / q.ncob;
    const int tpi = q.nstrips * q.tiles_per_strip;
    const int n = t / tpi, tr = t - n * tpi;
    const int strip = tr / q.tiles_per_strip, row = tr - strip * q.tiles_per_strip;
    const int wco = wave / q.WPX, wpx = wave % q.WPX;
    const int pt = e / (q.CT / 2), pp = e % (q.CT / 2);
    const int g4 = lane >> 4, l15 = lane & 15;
    const int ml = (wpx * q.PT + pt) * 16 + l15;
    const int SW = 1 << q.lsw;
    const int y = row * q.TR + (ml >> q.lsw), x = strip * SW + (ml & (SW - 1));
    const int co = cob * TN + (wco * (q.CT / 2) + pp) * 32 + 8 * g4;
    if (y >= q.Ho || x >= q.Wo || co >= q.Cout) return;
    typedef __attribute__((ext_vector_type(2))) short s2;
    const short fl = q.relu ? (short)0 : (short)-32768;
    const s2 floor2 = {fl, fl};
    auto pk = [&](float lo, float hi) {   // two results -> packed bf16, ReLU (or not) as the conv epilogue applies it: a packed int16 max
        return __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s2, pack_bf2(lo, hi)), floor2));
    };
    u32x4 v = {pk(a[0], a[1]), pk(a[2], a[3]), pk(b[0], b[1]), pk(b[2], b[3])};
    const long off = (((long)n * q.oH + y) * q.oW + x) * q.outC + co;
    if (q.mask_src) {
        const u32x4 m = *(const u32x4*)(q.mask_src + off);
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] &= pos_mask_pk_bf16(m[i]);
    }
    *(u32x4*)(q.out + off) = v;
}

#endif   // PP_DIL == 1
// ---------------------------------------------------------------------------------------------
template <int CFG> struct PpCfg;
#define PP_NAS(a, b, c, d, e) ((a) | ((b) << 4) | ((c) << 8) | ((d) << 12) | ((e) << 16))
// NAS: halo pieces per wave in the slots G0/0, G1/0, G0/1, G1/1, G0/2 (sum * 4 = igemm_fwd2's halo pieces per chunk: the same LDS budget)
// WP0: the G1 waves' R intervals are the long ones (1300-1600 cycles against G0's 750-950, tools/pp_stamps_raw.py): two of a wave's six weight
// pieces per stage go to G0 in the 128-channel shapes (tools/pp_fixed.py: forward / backward-data -2 %; three is no better; the 64-channel
// shapes, HBM-bound, gain nothing from one of three)
template <> struct PpCfg<IGF2_CFG_128x256> { static constexpr int WCO = 2, WPX = 4, CT = 4, PT = 4, NAS = PP_NAS(0, 2, 3, 1, 2), WP0 = 2; };
template <> struct PpCfg<IGF2_CFG_64x512> { static constexpr int WCO = 1, WPX = 8, CT = 4, PT = 4, NAS = PP_NAS(0, 4, 4, 2, 2), WP0 = 0; };
template <> struct PpCfg<IGF2_CFG_128x128> { static constexpr int WCO = 2, WPX = 4, CT = 4, PT = 2, NAS = PP_NAS(0, 1, 2, 1, 2), WP0 = 2; };
template <> struct PpCfg<IGF2_CFG_64x256> { static constexpr int WCO = 1, WPX = 8, CT = 4, PT = 2, NAS = PP_NAS(0, 2, 3, 1, 2), WP0 = 0; };
template <> struct PpCfg<IGF2_CFG_128x192> { static constexpr int WCO = 2, WPX = 4, CT = 4, PT = 3, NAS = PP_NAS(0, 2, 3, 1, 2), WP0 = 2; };
template <> struct PpCfg<IGF2_CFG_64x384> { static constexpr int WCO = 1, WPX = 8, CT = 4, PT = 3, NAS = PP_NAS(0, 3, 3, 2, 2), WP0 = 0; };
// (128x320 / 64x640, five pixel fragments per wave: measured 20-40 % slower than igemm_fwd2's -- three taps of fragments beside 80
// accumulators leave no registers for the tile bookkeeping; not instantiated)
template <int WCO, int WPX, int CT, int PT, int LSW, int NAS, int WP0, bool STAMP, bool DBG, bool POOLK = false, bool KS = false>
static hipError_t pp_launch_kernel3(int cfg, const IgFwdParams& p, int gx, hipStream_t st) {
    auto kern = igemm_pp_kernel<WCO, WPX, CT, PT, LSW, NAS, WP0, STAMP, DBG, POOLK, KS, PP_DIL>;
    const size_t lds = igemm_fwd2_lds_bytes(cfg, 9, p.g.npix_max) + (STAMP ? 8 * PP_NSTAMP * 4 : 0);  // same rings as igemm_fwd2's 9-tap kernels
    static size_t lds_set = 0;
    if (lds > lds_set) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        lds_set = lds;
    }
    hipLaunchKernelGGL(kern, dim3(gx, 1), dim3(512), lds, st, p);
    return hipGetLastError();
}
template <int WCO, int WPX, int CT, int PT, int LSW, int NAS, int WP0, bool STAMP, bool DBG>
static hipError_t pp_launch_kernel2(int cfg, const IgFwdParams& p, int gx, hipStream_t st) {
    if constexpr (PP_DIL == 1 && !STAMP && !DBG && ((PT == 4 && (LSW == 4 || LSW == 5)) || (PT == 2 && LSW == 4))) {
        if (p.pool_out) return pp_launch_kernel3<WCO, WPX, CT, PT, LSW, NAS, WP0, false, false, true>(cfg, p, gx, st);
    }
    if (p.pool_out) return hipErrorInvalidValue;   // (the planner only asks for the pool where igemm_pp_pool_lsw_mask allows it)
    if (p.ksplit > 1) {   // split-K: instantiated for the 128-channel shapes (the deep layers, which are the ones short of pixel tiles)
        if constexpr (!STAMP && !DBG && WCO == 2) {
            if (!p.kslab) return hipErrorInvalidValue;
            return pp_launch_kernel3<WCO, WPX, CT, PT, LSW, NAS, WP0, false, false, false, true>(cfg, p, gx, st);
        }
        return hipErrorInvalidValue;
    }
    return pp_launch_kernel3<WCO, WPX, CT, PT, LSW, NAS, WP0, STAMP, DBG, false>(cfg, p, gx, st);
}
// strip widths an instantiation exists for: the halo tile must fit the DMA pieces of a chunk and the 16-bit offsets of the LDS reads
constexpr bool pp_geo_ok(int TM, int LSW, int NAS) {
    const int SW = 1 << LSW, TR = TM >> LSW;
    if (TR < 1 || TR * SW != TM) return false;
    const int CW = (SW + 2 * PP_DIL + 7) / 8 * 8, NPIX = ((TR + 2 * PP_DIL) * CW + 31) / 32 * 32;
    int pieces = 0;
    for (int k = 0; k < 5; ++k) pieces += pp_na_slot(NAS, k);
    return NPIX <= pieces * 4 * 16 && NPIX * 64 + 2 * PP_DIL * CW * 64 < 65536;
}
// the kernel is instantiated per strip width (2^LSW = 8 .. 64): the planner's geometry must be the one the instantiation assumes
template <int WCO, int WPX, int CT, int PT, int NAS, int WP0, bool STAMP, bool DBG = false>
static hipError_t pp_launch_kernel(int cfg, const IgFwdParams& p, int gx, hipStream_t st) {
    constexpr int TM = WPX * PT * 16;
    const int SW = 1 << p.lsw, TR = TM >> p.lsw;
    const int CW = (SW + 2 * PP_DIL + 7) / 8 * 8, NPIX = ((TR + 2 * PP_DIL) * CW + 31) / 32 * 32;
    if (p.dil != PP_DIL || p.g.SW != SW || p.g.CW != CW || p.g.npix_max != NPIX || TR < 1) return hipErrorInvalidValue;
    switch (p.lsw) {
        case 3: if constexpr (pp_geo_ok(TM, 3, NAS)) return pp_launch_kernel2<WCO, WPX, CT, PT, 3, NAS, WP0, STAMP, DBG>(cfg, p, gx, st); break;
        case 4: if constexpr (pp_geo_ok(TM, 4, NAS)) return pp_launch_kernel2<WCO, WPX, CT, PT, 4, NAS, WP0, STAMP, DBG>(cfg, p, gx, st); break;
        case 5: if constexpr (pp_geo_ok(TM, 5, NAS)) return pp_launch_kernel2<WCO, WPX, CT, PT, 5, NAS, WP0, STAMP, DBG>(cfg, p, gx, st); break;
        case 6: if constexpr (pp_geo_ok(TM, 6, NAS)) return pp_launch_kernel2<WCO, WPX, CT, PT, 6, NAS, WP0, STAMP, DBG>(cfg, p, gx, st); break;
    }
    return hipErrorInvalidValue;
}
template <int CFG, bool STAMP = false, bool DBG = false>
static hipError_t pp_launch_one(const IgFwdParams& p, int gx, hipStream_t st) {
    using C = PpCfg<CFG>;
    return pp_launch_kernel<C::WCO, C::WPX, C::CT, C::PT, C::NAS, C::WP0, STAMP, DBG>(CFG, p, gx, st);
}
// (the five-fragment shapes are instantiated but lose: three taps of fragments beside 80 accumulators leave no room for the tile
// bookkeeping, whose spills cost more than the lean loop gains; igemm_fwd2 keeps them)
#if PP_DIL == 1
bool igemm_pp_has(int cfg) { return cfg >= 0 && cfg < IGF2_NCFG && cfg != IGF2_CFG_128x320 && cfg != IGF2_CFG_64x640; }
#endif
// this launch, planned with this geometry, is one the ping-pong kernels are instantiated for (3x3 taps, stride 1, dilation 1, the
// planner's halo tile for strip width 2^lsw)
#if PP_DIL == 1
bool igemm_pp_has_ksplit(int cfg) { return cfg == IGF2_CFG_128x256 || cfg == IGF2_CFG_128x128 || cfg == IGF2_CFG_128x192; }
#endif
bool igemm_pp_supports(int cfg, const IgFwdParams& p) {
    if (!igemm_pp_has(cfg) || p.stride != 1 || p.ostride != 1 || p.dil != PP_DIL || p.lsw < 3 || p.lsw > 6 || p.accumulate) return false;
    if (p.ksplit > 1 && (!igemm_pp_has_ksplit(cfg) || p.pool_out || !p.kslab)) return false;
    if (p.pool_out && PP_DIL != 1) return false;
    const int TM = igemm_fwd2_cfg_info(cfg).TM;
    const int SW = 1 << p.lsw, TR = TM >> p.lsw;
    const int CW = (SW + 2 * PP_DIL + 7) / 8 * 8, NPIX = ((TR + 2 * PP_DIL) * CW + 31) / 32 * 32;
    if (TR < 1 || p.g.SW != SW || p.g.CW != CW || p.g.npix_max != NPIX) return false;
    int nas = 0;
    switch (cfg) {
#define PP_CASE(C) case C: nas = PpCfg<C>::NAS; break;
        PP_CASE(IGF2_CFG_128x256) PP_CASE(IGF2_CFG_64x512) PP_CASE(IGF2_CFG_128x128) PP_CASE(IGF2_CFG_64x256)
        PP_CASE(IGF2_CFG_128x192) PP_CASE(IGF2_CFG_64x384)
#undef PP_CASE
    }
    return pp_geo_ok(TM, p.lsw, nas);
}
#if PP_DIL == 1
// (PT = 4: whole rows per wave at widths 16 and 32; PT = 2: at width 16 -- igemm_pp_kernel POOL_OK)
int igemm_pp_pool_lsw_mask(int cfg) {
    switch (cfg) {
        case IGF2_CFG_128x256: case IGF2_CFG_64x512: return (1 << 4) | (1 << 5);
        case IGF2_CFG_128x128: case IGF2_CFG_64x256: return 1 << 4;
    }
    return 0;
}
#endif
// 3x3 taps, stride 1 only (forward and backward-data of the conv3x3 layers)
hipError_t igemm_pp_launch(int cfg, const IgFwdParams& p, int gx, hipStream_t st) {
    if (p.stride != 1 || p.ostride != 1) return hipErrorInvalidValue;
#ifdef RSU_DEV_KERNELS   // developer build only (make DEV=1 -> build_ab/): the time-stamping and timing-ablation instantiations
    if ((p.dbg & 128) && p.stamps) {  // diagnostic build with interval time stamps
        if (cfg == IGF2_CFG_128x256 && (p.dbg & ~128)) return pp_launch_one<IGF2_CFG_128x256, true, true>(p, gx, st);
        if (cfg == IGF2_CFG_128x256) return pp_launch_one<IGF2_CFG_128x256, true>(p, gx, st);
        if (cfg == IGF2_CFG_64x512) return pp_launch_one<IGF2_CFG_64x512, true>(p, gx, st);
    }
    if (p.dbg & (63 | 64 | 256 | 512)) {  // timing experiments (RSU_FWD_DBG bits 0-6, 8): the build that tests those bits
        if (cfg == IGF2_CFG_128x256) return pp_launch_one<IGF2_CFG_128x256, false, true>(p, gx, st);
        if (cfg == IGF2_CFG_64x512) return pp_launch_one<IGF2_CFG_64x512, false, true>(p, gx, st);
    }
#endif
    switch (cfg) {
        case IGF2_CFG_128x256: return pp_launch_one<IGF2_CFG_128x256>(p, gx, st);
        case IGF2_CFG_64x512: return pp_launch_one<IGF2_CFG_64x512>(p, gx, st);
        case IGF2_CFG_128x128: return pp_launch_one<IGF2_CFG_128x128>(p, gx, st);
        case IGF2_CFG_64x256: return pp_launch_one<IGF2_CFG_64x256>(p, gx, st);
        case IGF2_CFG_128x192: return pp_launch_one<IGF2_CFG_128x192>(p, gx, st);
        case IGF2_CFG_64x384: return pp_launch_one<IGF2_CFG_64x384>(p, gx, st);
    }
    return hipErrorInvalidValue;
}

#if PP_DIL == 1
// ---- split-K support (host)
static void pp_cfg_dims(int cfg, int& WCO, int& WPX, int& CT, int& PT) {
    WCO = WPX = CT = PT = 0;
    switch (cfg) {
#define PP_CASE(C) case C: WCO = PpCfg<C>::WCO; WPX = PpCfg<C>::WPX; CT = PpCfg<C>::CT; PT = PpCfg<C>::PT; break;
        PP_CASE(IGF2_CFG_128x256) PP_CASE(IGF2_CFG_64x512) PP_CASE(IGF2_CFG_128x128) PP_CASE(IGF2_CFG_64x256)
        PP_CASE(IGF2_CFG_128x192) PP_CASE(IGF2_CFG_64x384)
#undef PP_CASE
    }
}
size_t igemm_pp_slab_floats(int cfg, const IgFwdParams& p) {
    const IgFwdCfgInfo ci = igemm_fwd2_cfg_info(cfg);
    return (size_t)p.N * p.g.nstrips * p.g.tiles_per_strip * p.ncob * ci.TN * ci.TM;
}
hipError_t igemm_pp_finish_launch(int cfg, const IgFwdParams& p, hipStream_t st) {
    PpFinishParams q;
    memset(&q, 0, sizeof(q));
    pp_cfg_dims(cfg, q.WCO, q.WPX, q.CT, q.PT);
    if (!q.CT || p.ksplit < 2 || !p.kslab) return hipErrorInvalidValue;
    q.slab = p.kslab; q.slice_stride = p.kslab_stride; q.ksplit = p.ksplit;
    q.nslots = p.N * p.g.nstrips * p.g.tiles_per_strip * p.ncob; q.ncob = p.ncob;
    q.lsw = p.lsw; q.TR = (q.WPX * q.PT * 16) >> p.lsw;
    q.nstrips = p.g.nstrips; q.tiles_per_strip = p.g.tiles_per_strip;
    q.out = p.out; q.mask_src = p.mask_src;
    q.Ho = p.Ho; q.Wo = p.Wo; q.oH = p.oH; q.oW = p.oW; q.outC = p.outC; q.Cout = p.Cout; q.relu = p.relu;
    const long threads = (long)q.nslots * 8 * (q.CT / 2) * q.PT * 64;
    hipLaunchKernelGGL(k_pp_splitk_finish, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st, q);
    return hipGetLastError();
}
#endif   // PP_DIL == 1
