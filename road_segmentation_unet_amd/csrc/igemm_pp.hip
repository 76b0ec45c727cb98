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
// The two groups also split the prefetch streams: G1 owns the weights, G0 the halo tiles. In global intervals (G0 runs R(s) in
// interval 2s, M(s) in 2s+1; G1 one later; a wave's reads of R(s) have landed when its M(s) has issued its last MFMA):
//   * weights: ring of 3 stage slots, stage j of every chunk in slot j. In R(s) (interval 2s+1) a G1 wave issues its pieces of
//     stage s+2 into the slot stage s-1 used (last read: G1's own R(s-1), interval 2s-1, retired in 2s), then waits with a
//     counted vmcnt for its pieces of stage s+1 (issued in 2s-1); the barrier closing 2s+1 publishes them to G0's R(s+1) (2s+2).
//   * halo: ring of 2 chunk slots. In R(3c+1) (interval 6c+2) a G0 wave issues its pieces of chunk c+1 into the slot chunk c-1
//     used (last read: G1's R(3c-1), interval 6c-1, retired in 6c) and waits for them behind its MFMAs of M(3c+2) (interval 6c+5);
//     the barrier closing 6c+5 publishes them to G0's R(3c+3) (6c+6).
#include <type_traits>

#include "igemm.h"

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
// NA = halo DMA pieces per G0 wave per chunk (NA * 4 * 16 halo pixels at most)
template <int WCO, int WPX, int CT, int PT, int NA, bool STAMP>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2)))
igemm_pp_kernel(const IgFwdParams p) {
    constexpr int NW = 8, NG = 4, KW = 3, TPS = 3;
    static_assert(WCO * WPX == NW, "eight waves: two per SIMD");
    constexpr int TN = WCO * CT * 16, TM = WPX * PT * 16;
    constexpr int WT = TN / 16;
    constexpr int WBUF = TPS * WT * 1024;
    constexpr int NWB = 3, NAB = 2;
    static_assert((TPS * WT) % NG == 0, "weight pieces per stage split evenly over the four G1 waves");
    constexpr int WPS = TPS * WT / NG;   // weight DMA instructions per G1 wave per stage
    constexpr int NST = (CT / 2) * PT;   // epilogue buffer stores per wave per tile (always issued)
    static_assert((CT % 2) == 0, "bad config");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __attribute__((address_space(3))) char* lds = (__attribute__((address_space(3))) char*)smem;
    const int ABUF = p.g.npix_max * 64;
    const int a_base = NWB * WBUF;
    const int dummy_base = a_base + NAB * ABUF;  // 1 KiB scratch slot for padding loads

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int grp = wave >> 2, w4 = wave & 3;
    const int wco = wave / WPX, wpx = wave % WPX;
    const int g4 = lane >> 4, l15 = lane & 15;
    const int SW = p.g.SW, CW = p.g.CW, lsw = p.lsw, TR = TM >> lsw;

    // ---- this workgroup's tile list (XCD-aware numbering as in igemm_fwd2)
    int vid = blockIdx.x;
    {
        const int q = gridDim.x >> 3, r = gridDim.x & 7, x = vid & 7;
        vid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (vid >> 3);
    }
    const int cob = vid % p.ncob;
    const int tile0 = vid / p.ncob, tstride = gridDim.x / p.ncob;
    const int tpi = p.g.nstrips * p.g.tiles_per_strip;
    const int ntile_m = p.N * tpi;
    const int my_tiles = tile0 < ntile_m ? (ntile_m - tile0 + tstride - 1) / tstride : 0;
    if (my_tiles == 0) return;
    const int nchunks = p.nchunk[0] + p.nchunk[1] + p.nchunk[2];
    const int GC = my_tiles * nchunks;  // chunks in this workgroup's stream

    struct Tile { int n, x0, y0; };
    auto decode = [&](int k) {
        Tile T;
        int t = tile0 + k * tstride;
        T.n = t / tpi;
        t -= T.n * tpi;
        const int strip = t / p.g.tiles_per_strip;
        T.x0 = strip * SW;
        T.y0 = (t - strip * p.g.tiles_per_strip) * TR;
        return T;
    };

    // ---- workgroup constants (per lane)
    int boff[PT][KW];   // byte offset (inside a halo slot) of this lane's 16-byte fragment piece, per pixel fragment and kx
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) {
        const int ml = (wpx * PT + pt) * 16 + l15;
        const int ty = ml >> lsw, tx = ml & (SW - 1);
        const int hp0 = ty * CW + tx;
#pragma unroll
        for (int kx = 0; kx < KW; ++kx) {
            const int hp = hp0 + kx * p.dil;
            boff[pt][kx] = (hp << 6) + ((g4 ^ ((hp >> 1) & 2)) << 4);  // CW % 8 == 0: a ky shift keeps the swizzle
        }
    }
    const int afrag = (wco * CT * 64 + lane) * 16;  // this lane's 16 bytes inside weight tile 0 of the wave
    const int row_bytes = (CW * p.dil) << 6;        // one kernel row down in the halo tile
    const int npieces = p.g.npix_max >> 4;
    const int lq = lane >> 2;
    auto mk = [&](const void* ptr) { return __builtin_amdgcn_make_buffer_rsrc((void*)ptr, 0, 0x7fffffff, 0x00020000); };

    // ---- weight prefetch stream (G1). Stage s of a tile is the contiguous block [s*3 .. s*3+3) x [all 16-row tiles] of the packed
    // weights; a wave's WPS pieces have constant offsets inside it (folded into the per-lane voffset), the stage is one scalar
    // offset that advances by a constant and rewinds per tile. The stream never ends: behind the last stage it simply starts the
    // tile's weights again (valid memory, slots nobody reads), so the counted waits hold to the very end.
    const int nstage_tile = nchunks * 3;
    const unsigned stage_bytes = (unsigned)TPS * p.ntiles_w * 1024;
    const unsigned w_tile_soff = (unsigned)(p.tile_off + cob * WT) * 1024u;
    unsigned w_soff = w_tile_soff;     // stage to prefetch next
    int w_sit = 0;                     // its stage index inside the tile
    unsigned w_voff[WPS];              // per-lane byte offset of piece q inside a stage block
#pragma unroll
    for (int q = 0; q < WPS; ++q) {
        const int i = q * NG + w4;
        const int tap_l = i / WT, tl = i - tap_l * WT;
        const bool real = p.tile_off + cob * WT + tl < p.ntiles_w;
        w_voff[q] = real ? (unsigned)((tap_l * p.ntiles_w + tl) * 1024 + lane * 16) : RSU_SENT;
    }
    auto issue_w = [&](int slot) {
        const __amdgpu_buffer_rsrc_t rw = mk(p.wp);
#pragma unroll
        for (int q = 0; q < WPS; ++q) bdma16(rw, w_voff[q], w_soff, (void*)(lds + slot * WBUF + (q * NG + w4) * 1024));
        if (++w_sit == nstage_tile) {
            w_sit = 0;
            w_soff = w_tile_soff;
        } else {
            w_soff += stage_bytes;
        }
    };
    // ---- halo prefetch stream (G0): exactly NA pieces per wave per chunk; clipped / padded pixels come back as zeros
    unsigned a_voff[NA];           // per-lane byte offset of piece q inside the current source, or RSU_SENT
    const char* a_ptr = nullptr;   // current source, shifted back by the padding so that every in-window offset is >= 0
    unsigned a_soff = 0;           // byte offset of the (padded) halo origin + channel chunk in that source
    int a_crem = 0;                // channels left in the current source (>= 32 except in a partial last chunk)
    int a_cl = 0;                  // chunk (inside its tile) of the next halo to prefetch
    int a_next_src = 0;            // chunk index at which the next source begins
    int a_si = 0;                  // current source
    int ia_slot = 0;               // ring slot of the next halo
    auto setup_a = [&](const Tile& T, int si) {
        const bf16_t* sptr = si == 0 ? p.src[0].ptr : (si == 1 ? p.src[1].ptr : p.src[2].ptr);
        const int sH = si == 0 ? p.src[0].H : (si == 1 ? p.src[1].H : p.src[2].H);
        const int sW = si == 0 ? p.src[0].W : (si == 1 ? p.src[1].W : p.src[2].W);
        const int sC = si == 0 ? p.src[0].C : (si == 1 ? p.src[1].C : p.src[2].C);
        const int soy = si == 0 ? p.src[0].oy : (si == 1 ? p.src[1].oy : p.src[2].oy);
        const int sox = si == 0 ? p.src[0].ox : (si == 1 ? p.src[1].ox : p.src[2].ox);
        a_ptr = (const char*)(sptr - ((long)p.pad * sW + p.pad) * sC);
        a_soff = (unsigned)((((long)(T.n * sH + T.y0 + soy) * sW + (T.x0 + sox)) * sC) * 2);
        a_crem = sC;
        const int iy0 = T.y0 - p.pad, ix0 = T.x0 - p.pad;
#pragma unroll
        for (int q = 0; q < NA; ++q) {
            const int hp = (q * NG + w4) * 16 + lq;
            const int kg8 = ((lane & 3) ^ ((hp >> 1) & 2)) * 8;
            const int rr = div_magic(hp, p.g.inv_CW);
            const int cc = hp - rr * CW;
            const bool ok = ((unsigned)(iy0 + rr) < (unsigned)p.Hin) && ((unsigned)(ix0 + cc) < (unsigned)p.Win);
            a_voff[q] = ok ? (unsigned)(((rr * sW + cc) * sC + kg8) * 2) : RSU_SENT;
        }
    };
    Tile ptile = decode(0);     // tile whose halo is being prefetched
    int pk = 0;                 // index of ptile in this workgroup's list
    bool a_started = false;
    auto issue_a = [&]() {  // the next chunk of the stream (moves on to the next source / tile where one ends)
        if (a_cl == 0) {
            if (a_started) {
                ++pk;
                ptile = decode(pk);
                a_si = 0;
                a_next_src = p.nchunk[0];
                setup_a(ptile, 0);
            }
        } else if (a_cl == a_next_src) {
            ++a_si;
            a_next_src += a_si == 1 ? p.nchunk[1] : p.nchunk[2];
            setup_a(ptile, a_si);
        }
        a_started = true;
        const __amdgpu_buffer_rsrc_t rs = mk(a_ptr);
        const int dst = a_base + ia_slot * ABUF;
        if (a_crem >= 32) {
#pragma unroll
            for (int q = 0; q < NA; ++q) {
                const int j = q * NG + w4;
                bdma16(rs, a_voff[q], a_soff, (void*)(lds + (j < npieces ? dst + j * 1024 : dummy_base)));
            }
        } else {  // partial last chunk of a source whose channel count is not a multiple of 32: the missing channels read as zeros
#pragma unroll
            for (int q = 0; q < NA; ++q) {
                const int j = q * NG + w4;
                const int hp = j * 16 + lq;
                const int kg8 = ((lane & 3) ^ ((hp >> 1) & 2)) * 8;
                bdma16(rs, kg8 < a_crem ? a_voff[q] : RSU_SENT, a_soff, (void*)(lds + (j < npieces ? dst + j * 1024 : dummy_base)));
            }
        }
        ia_slot ^= 1;
        a_soff += 64;
        a_crem -= 32;
        a_cl = a_cl + 1 == nchunks ? 0 : a_cl + 1;
    };

    // bias of this workgroup's TN channels lives in LDS (behind the scratch slot)
    const int bias_base = dummy_base + 1024;
    const int stamp_base = bias_base + 512;
    int stamp_i = 0;
    auto stamp = [&]() {
        if constexpr (STAMP) {
            if (stamp_i < PP_NSTAMP - 2) {
                const unsigned t = (unsigned)__builtin_amdgcn_s_memtime();
                if (lane == 0) *(__attribute__((address_space(3))) unsigned*)(lds + stamp_base + (wave * PP_NSTAMP + stamp_i) * 4) = t;
            }
            ++stamp_i;
        }
    };
    if (threadIdx.x < TN) {
        const int co = cob * TN + threadIdx.x;
        const float bvv = (p.bias && co < p.Cout) ? p.bias[co] : 0.f;
        *(__attribute__((address_space(3))) float*)(lds + bias_base + threadIdx.x * 4) = bvv;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // retire the ordinary loads before the LDS-DMA stream starts

    // scalar byte offset of a tile's first output pixel (+ this wave's first channel) and per-lane offset of epilogue store e
    auto out_sbase = [&](const Tile& T) {
        return (unsigned)((((long)(T.n * p.oH + T.y0) * p.oW + T.x0) * p.outC + cob * TN + wco * (CT / 2) * 32) * 2);
    };
    auto out_voff = [&](const Tile& T, int e) {
        const int pt = e / (CT / 2), pp = e % (CT / 2);
        const int ml = (wpx * PT + pt) * 16 + l15;
        const int ty = ml >> lsw, tx = ml & (SW - 1);
        const bool pok = (T.y0 + ty < p.Ho) && (T.x0 + tx < p.Wo);
        const int ovoff_pt = ((ty * p.oW + tx) * p.outC + 8 * g4) * 2;
        const int co = cob * TN + (wco * (CT / 2) + pp) * 32 + 8 * g4;
        return (pok && co < p.Cout) ? (unsigned)(ovoff_pt + pp * 64) : RSU_SENT;
    };
// store + its wait states as ONE asm statement (the >64-bit store / VALU-write hazard); the diagnostic build
// spills descriptor words, which come back through v_readlane right in front of the store: five more wait states there
#define PP_STORE(R, VOFF)                                                                                                                       \
    do {                                                                                                                                        \
        if constexpr (STAMP)                                                                                                                    \
            asm volatile("s_nop 4\n\tbuffer_store_dwordx4 %0, %1, %2, %3 offen\n\ts_nop 3" ::"v"(R), "v"(VOFF), "s"(orsrc), "s"(sbase) : "memory"); \
        else                                                                                                                                    \
            asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen\n\ts_nop 3" ::"v"(R), "v"(VOFF), "s"(orsrc), "s"(sbase) : "memory");         \
    } while (0)
    auto epilogue = [&](const Tile& T, f32x4(&acc)[CT][PT]) {
        const __amdgpu_buffer_rsrc_t orsrc = mk(p.out);
        const __amdgpu_buffer_rsrc_t mrsrc = mk(p.mask_src ? (const void*)p.mask_src : (const void*)p.out);
        const unsigned sbase = out_sbase(T);
        if (!p.mask_src && !p.accumulate) {
            typedef __attribute__((ext_vector_type(2))) short s2;
            const short fl = p.relu ? (short)0 : (short)-32768;
            const s2 floor2 = {fl, fl};
#pragma unroll
            for (int e = 0; e < NST; ++e) {
                const int pt = e / (CT / 2), pp = e % (CT / 2);
                const unsigned voff = out_voff(T, e);
                u32x4 r;
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const unsigned lo = pack_bf2(acc[2 * pp][pt][2 * i], acc[2 * pp][pt][2 * i + 1]);
                    const unsigned hi = pack_bf2(acc[2 * pp + 1][pt][2 * i], acc[2 * pp + 1][pt][2 * i + 1]);
                    r[i] = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s2, lo), floor2));
                    r[2 + i] = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s2, hi), floor2));
                }
                PP_STORE(r, voff);
            }
            return;
        }
        constexpr int EB = NST % 4 == 0 ? 4 : 2;
        static_assert(NST % EB == 0, "epilogue batches");
#pragma unroll
        for (int b0 = 0; b0 < NST; b0 += EB) {
            unsigned voffs[EB];
            u32x4 mk4[EB], ob4[EB];
#pragma unroll
            for (int e = 0; e < EB; ++e) {
                voffs[e] = out_voff(T, b0 + e);
                if (p.mask_src) mk4[e] = __builtin_amdgcn_raw_buffer_load_b128(mrsrc, voffs[e], sbase, 0);
                if (p.accumulate) ob4[e] = __builtin_amdgcn_raw_buffer_load_b128(orsrc, voffs[e], sbase, 0);
            }
#pragma unroll
            for (int e = 0; e < EB; ++e) {
                const int pt = (b0 + e) / (CT / 2), pp = (b0 + e) % (CT / 2);
                float v[8];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    v[i] = acc[2 * pp][pt][i];
                    v[4 + i] = acc[2 * pp + 1][pt][i];
                }
                if (p.mask_src) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        if (!(bf_lo(mk4[e][i]) > 0.f)) v[2 * i] = 0.f;
                        if (!(bf_hi(mk4[e][i]) > 0.f)) v[2 * i + 1] = 0.f;
                    }
                }
                if (p.accumulate) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        v[2 * i] += bf_lo(ob4[e][i]);
                        v[2 * i + 1] += bf_hi(ob4[e][i]);
                    }
                }
                unsigned r0 = pack_bf2(v[0], v[1]), r1 = pack_bf2(v[2], v[3]), r2 = pack_bf2(v[4], v[5]), r3 = pack_bf2(v[6], v[7]);
                if (p.relu) {
                    r0 = relu_pk_bf16(r0);
                    r1 = relu_pk_bf16(r1);
                    r2 = relu_pk_bf16(r2);
                    r3 = relu_pk_bf16(r3);
                }
                const u32x4 r = {r0, r1, r2, r3};
                PP_STORE(r, voffs[e]);
            }
            __builtin_amdgcn_sched_barrier(0);  // keep the batches apart
        }
    };
    auto bar = [&]() {
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };

    // ---- prologue: G1 issues W(0), W(1) and waits for W(0); G0 issues A(0) and waits for it; one barrier publishes both
    // (raw barrier: __syncthreads() would drain the LDS-DMA stream; the bias words above are the only ordinary LDS stores)
    if (grp) {
        if (!(p.dbg & 1)) {
            issue_w(0);
            issue_w(1);
        }
        RSU_WAIT_VMCNT(WPS);
    } else {
        setup_a(ptile, 0);
        a_next_src = p.nchunk[0];
        if (!(p.dbg & 2)) issue_a();
        RSU_WAIT_VMCNT(0);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    bar();
    if (grp) bar();  // G1 sits out interval 0

    unsigned long long clk0 = 0, rt0 = 0;
    if constexpr (STAMP) {
        clk0 = __builtin_amdgcn_s_memtime();
        rt0 = __builtin_amdgcn_s_memrealtime();
    }
    int gc = 0;       // stream chunk counter
    int ca_slot = 0;  // halo ring slot of the chunk being computed
    for (int ck = 0; ck < my_tiles; ++ck) {
        const Tile ctile = decode(ck);
        // accumulators live for exactly one tile; they start at the bias
        f32x4 acc[CT][PT];
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
        for (int c = 0; c < nchunks; ++c, ++gc) {
            const bool after_epi = (c == 0) && gc > 0;  // the NST stores of the previous tile sit in front of this chunk's issues
            const int ab = a_base + ca_slot * ABUF;
            auto phase = [&](auto jc) {
                constexpr int J = decltype(jc)::value;
                // ================= R interval: the stage's fragment reads + bookkeeping for the stages ahead
                bf16x8 fa[TPS][CT], fb[TPS][PT];
                if (!(p.dbg & 32)) {  // (dbg bit 5: timing experiment without fragment reads and MFMAs -- the LDS-DMA streams alone)
                    const int rowoff = ab + J * row_bytes;  // wave-uniform
#pragma unroll
                    for (int tl = 0; tl < TPS; ++tl) {
#pragma unroll
                        for (int ct = 0; ct < CT; ++ct)
                            fa[tl][ct] = *(const __attribute__((address_space(3))) bf16x8*)(lds + afrag + J * WBUF + (tl * WT + ct) * 1024);
#pragma unroll
                        for (int pt = 0; pt < PT; ++pt)
                            fb[tl][pt] = *(const __attribute__((address_space(3))) bf16x8*)(lds + (boff[pt][tl] + rowoff));
                    }
                }
                if (grp) {
                    if (!(p.dbg & 1)) issue_w((J + 2) % NWB);
                    // this wave's pieces of the next stage (issued one phase ago) are the oldest loads still allowed in flight
                    if (J == 0 && after_epi) {
                        RSU_WAIT_VMCNT(WPS + NST);
                    } else {
                        RSU_WAIT_VMCNT(WPS);
                    }
                } else if (J == 1) {
                    if (gc + 1 < GC && !(p.dbg & 2)) issue_a();
                }
                stamp();
                bar();
                stamp();
                // ================= M interval: the MFMAs of this stage, nothing else
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_setprio(1);
                if (!(p.dbg & 32))
#pragma unroll
                for (int tl = 0; tl < TPS; ++tl)
#pragma unroll
                    for (int pt = 0; pt < PT; ++pt)
#pragma unroll
                        for (int ct = 0; ct < CT; ++ct) mfma_bf16_inplace(acc[ct][pt], fa[tl][ct], fb[tl][pt]);
                __builtin_amdgcn_s_setprio(0);
                __builtin_amdgcn_sched_barrier(0);
                if (J == 2) {
                    if (!grp) RSU_WAIT_VMCNT(0);                  // G0: the next chunk's halo pieces (nothing younger is in flight)
                    if (c == nchunks - 1) mfma_results_fence();  // straight behind the tile's last MFMA
                }
                stamp();
                bar();
                stamp();
            };
            phase(std::integral_constant<int, 0>{});
            phase(std::integral_constant<int, 1>{});
            phase(std::integral_constant<int, 2>{});
            ca_slot ^= 1;
        }
        // the finished tile's epilogue opens the wave's next R interval (its partner is in an M interval meanwhile)
        if (!(p.dbg & 8)) epilogue(ctile, acc);
    }
    if (!grp) bar();  // G0 sits out the last interval (G1's last epilogue)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // nothing may land in this workgroup's LDS after it has gone
    if constexpr (STAMP) {
        if (p.stamps) {
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            const unsigned dclk = (unsigned)(__builtin_amdgcn_s_memtime() - clk0), drt = (unsigned)(__builtin_amdgcn_s_memrealtime() - rt0);
            for (int i = lane; i < PP_NSTAMP; i += 64)
                p.stamps[((long)blockIdx.x * NW + wave) * PP_NSTAMP + i] =
                    i == PP_NSTAMP - 2 ? dclk : (i == PP_NSTAMP - 1 ? drt :  // shader cycles and 100-MHz ticks of the main loop
                    (i < stamp_i ? *(__attribute__((address_space(3))) unsigned*)(lds + stamp_base + (wave * PP_NSTAMP + i) * 4) : 0u));
        }
    }
}

// ---------------------------------------------------------------------------------------------
template <int CFG> struct PpCfg;
// NA = halo DMA pieces per G0 wave per chunk: igemm_fwd2's per-wave counts, doubled (four waves carry the halo stream here)
template <> struct PpCfg<IGF2_CFG_128x256> { static constexpr int WCO = 2, WPX = 4, CT = 4, PT = 4, NA = 8; };
template <> struct PpCfg<IGF2_CFG_64x512> { static constexpr int WCO = 1, WPX = 8, CT = 4, PT = 4, NA = 12; };
template <> struct PpCfg<IGF2_CFG_128x128> { static constexpr int WCO = 2, WPX = 4, CT = 4, PT = 2, NA = 6; };
template <> struct PpCfg<IGF2_CFG_64x256> { static constexpr int WCO = 1, WPX = 8, CT = 4, PT = 2, NA = 8; };
template <> struct PpCfg<IGF2_CFG_128x192> { static constexpr int WCO = 2, WPX = 4, CT = 4, PT = 3, NA = 8; };
template <> struct PpCfg<IGF2_CFG_64x384> { static constexpr int WCO = 1, WPX = 8, CT = 4, PT = 3, NA = 10; };
template <> struct PpCfg<IGF2_CFG_128x320> { static constexpr int WCO = 2, WPX = 4, CT = 4, PT = 5, NA = 10; };
template <> struct PpCfg<IGF2_CFG_64x640> { static constexpr int WCO = 1, WPX = 8, CT = 4, PT = 5, NA = 14; };

template <int CFG, bool STAMP = false>
static hipError_t pp_launch_one(const IgFwdParams& p, int gx, hipStream_t st) {
    using C = PpCfg<CFG>;
    auto kern = igemm_pp_kernel<C::WCO, C::WPX, C::CT, C::PT, C::NA, STAMP>;
    const size_t lds = igemm_fwd2_lds_bytes(CFG, 9, p.g.npix_max) + (STAMP ? 8 * PP_NSTAMP * 4 : 0);  // same rings as igemm_fwd2's 9-tap kernels
    static size_t lds_set = 0;
    if (lds > lds_set) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        lds_set = lds;
    }
    hipLaunchKernelGGL(kern, dim3(gx, 1), dim3(512), lds, st, p);
    return hipGetLastError();
}
// 3x3 taps, stride 1 only (forward and backward-data of the conv3x3 layers)
hipError_t igemm_pp_launch(int cfg, const IgFwdParams& p, int gx, hipStream_t st) {
    if (p.stride != 1 || p.ostride != 1) return hipErrorInvalidValue;
    if ((p.dbg & 128) && p.stamps) {  // diagnostic build with interval time stamps
        if (cfg == IGF2_CFG_128x256) return pp_launch_one<IGF2_CFG_128x256, true>(p, gx, st);
        if (cfg == IGF2_CFG_64x512) return pp_launch_one<IGF2_CFG_64x512, true>(p, gx, st);
    }
    switch (cfg) {
        case IGF2_CFG_128x256: return pp_launch_one<IGF2_CFG_128x256>(p, gx, st);
        case IGF2_CFG_64x512: return pp_launch_one<IGF2_CFG_64x512>(p, gx, st);
        case IGF2_CFG_128x128: return pp_launch_one<IGF2_CFG_128x128>(p, gx, st);
        case IGF2_CFG_64x256: return pp_launch_one<IGF2_CFG_64x256>(p, gx, st);
        case IGF2_CFG_128x192: return pp_launch_one<IGF2_CFG_128x192>(p, gx, st);
        case IGF2_CFG_64x384: return pp_launch_one<IGF2_CFG_64x384>(p, gx, st);
        case IGF2_CFG_128x320: return pp_launch_one<IGF2_CFG_128x320>(p, gx, st);
        case IGF2_CFG_64x640: return pp_launch_one<IGF2_CFG_64x640>(p, gx, st);
    }
    return hipErrorInvalidValue;
}
