// igemm_pp: third generation of the NHWC bf16 implicit-GEMM 3x3 convolution (forward and backward-data): the persistent stage
// stream of igemm_fwd2 (same data layout: fragment-ordered weights, pixel-major swizzled halo tile, channel-permuted 16-byte
// stores, same tile shapes and the same summation order, so results are bit-identical) run as a PING-PONG between the two
// waves of every SIMD.
//
// The eight waves of a workgroup form two groups, G0 = waves 0-3 and G1 = waves 4-7 (waves w and w+4 share a SIMD). Time is cut
// into INTERVALS by workgroup barriers. A PHASE is one tap of one 32-channel chunk: CT+PT fragment reads and CT*PT MFMAs per
// wave. Every wave runs   R(0) | M(0) | R(1) | M(1) | ...   with a barrier at every '|', where R(p) issues the LDS fragment
// reads of phase p plus all the bookkeeping (LDS-DMA prefetches, counted waits, tile changes, the epilogue of the finished tile)
// and M(p) is nothing but the phase's MFMAs. G1 starts one interval late, so in every interval one wave of each SIMD feeds the
// matrix pipe while its partner does everything else: LDS latency, the issue cost of the LDS-DMA instructions (60-185 cycles
// each, what bounded igemm_fwd2) and the address arithmetic no longer sit in a wave's own MFMA stream, and a wave needs only ONE
// set of fragment registers (it never reads and multiplies at the same time).
//
// Ring discipline (stage = 3 taps of a chunk, weight ring of 3 stage slots = slot j for stage j of every chunk; halo ring of 2
// chunk slots), in global intervals (G0 runs R(p) in interval 2p, G1 in 2p+1; a wave's reads of R(p) have landed at the latest
// when its M(p) has issued its last MFMA):
//   * the weights of stage s+2 go to the slot stage s-1 used: its last reader is G1's R(3s-1) in interval 6s-1, retired by the end
//     of interval 6s, so the slot may be written from interval 6s+1 on: the waves issue in R(3s+1) (G0: 6s+2, G1: 6s+3);
//   * the halo of chunk c+1 goes to the slot of chunk c-1, free from interval 18c+1 on: issued in R(9c+2) ... R(9c+6);
//   * a wave waits for its own pieces of stage s+1 (and, in the chunk's last phase, of the next halo) with a COUNTED
//     s_waitcnt vmcnt(N) at the end of R(3s+2): a barrier lies between that wait and the first read of the data (G0's R(3s+3)).
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

// halo pieces of the next chunk are issued in the R intervals of phases 2, 3, 5 and 6 (never beside the weight pieces of
// phases 1, 4, 7): slot i of {2,3,5,6} carries NA/4 pieces, the first NA%4 slots one more
constexpr int a_slot_of(int pp) { return pp == 2 ? 0 : (pp == 3 ? 1 : (pp == 5 ? 2 : (pp == 6 ? 3 : -1))); }
constexpr int a_cnt_slot(int NA, int i) { return NA / 4 + (i < NA % 4 ? 1 : 0); }
constexpr int a_cnt(int NA, int pp) { return a_slot_of(pp) < 0 ? 0 : a_cnt_slot(NA, a_slot_of(pp)); }
constexpr int a_first(int NA, int pp) {
    int n = 0;
    for (int i = 0; i < a_slot_of(pp); ++i) n += a_cnt_slot(NA, i);
    return n;
}

}  // namespace

// STAMP: diagnostic build (RSU_FWD_DBG bit 7 + RSU_STAMP_PTR): every wave notes s_memtime behind each barrier (in LDS, dumped to
// p.stamps at the end: [block][wave][PP_NSTAMP]); tools/pp_stamps.py prints the interval lengths. Never on the product path.
#define PP_NSTAMP 640
template <int WCO, int WPX, int CT, int PT, int NA, bool STAMP>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2)))
igemm_pp_kernel(const IgFwdParams p) {
    constexpr int NW = 8, NTAP = 9, KW = 3, TPS = 3;
    static_assert(WCO * WPX == NW, "eight waves: two per SIMD");
    constexpr int TN = WCO * CT * 16, TM = WPX * PT * 16;
    constexpr int WT = TN / 16;
    constexpr int WBUF = TPS * WT * 1024;
    constexpr int NWB = 3, NAB = 2, DW = 2;
    constexpr int WPS = (TPS * WT + NW - 1) / NW;  // weight DMA instructions per wave per stage (padded to a constant)
    constexpr int NST = (CT / 2) * PT;             // epilogue buffer stores per wave per tile (always issued)
    static_assert((CT % 2) == 0, "bad config");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __attribute__((address_space(3))) char* lds = (__attribute__((address_space(3))) char*)smem;
    const int ABUF = p.g.npix_max * 64;
    const int a_base = NWB * WBUF;
    const int dummy_base = a_base + NAB * ABUF;  // 1 KiB scratch slot for padding loads

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wco = wave / WPX, wpx = wave % WPX;
    const int g4 = lane >> 4, l15 = lane & 15;
    const int SW = p.g.SW, CW = p.g.CW, lsw = p.lsw, TR = TM >> lsw;

    // ---- this workgroup's tile list (XCD-aware numbering as in igemm_fwd2)
    int vid = blockIdx.x;
    if (!(p.dbg & 64)) {
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
    const int npieces = p.g.npix_max >> 4;
    const int lq = lane >> 2;
    auto mk = [&](const void* ptr) { return __builtin_amdgcn_make_buffer_rsrc((void*)ptr, 0, 0x7fffffff, 0x00020000); };

    // ---- weight prefetch stream (one scalar source pointer that advances by a constant per stage and rewinds per tile)
    const int nstage_tile = nchunks * 3;
    const long stage_bytes = (long)TPS * p.ntiles_w * 1024;
    const char* const w_tile_base = (const char*)p.wp + (long)(p.tile_off + cob * WT) * 1024;
    const char* w_cur = w_tile_base;   // source of the next stage to prefetch
    int w_sit = 0;                     // its stage index inside the tile
    int wpo[WPS];                      // byte offset of piece q inside a stage block, or -1 (padding piece / beyond the packed rows)
#pragma unroll
    for (int q = 0; q < WPS; ++q) {
        const int i = q * NW + wave;
        const int tap_l = i / WT, tl = i - tap_l * WT;
        const bool real = (i < TPS * WT) && (p.tile_off + cob * WT + tl < p.ntiles_w);
        wpo[q] = real ? (tap_l * p.ntiles_w + tl) * 1024 : -1;
    }
    auto issue_w = [&](int slot) {
        const int dst = slot * WBUF;
#pragma unroll
        for (int q = 0; q < WPS; ++q) {
            const int i = q * NW + wave;
            const char* base = wpo[q] >= 0 ? w_cur + wpo[q] : (const char*)p.zero_page;
            dma16(base + lane * 16, (void*)(lds + (i < TPS * WT ? dst + i * 1024 : dummy_base)));
        }
        if (++w_sit == nstage_tile) {
            w_sit = 0;
            w_cur = w_tile_base;
        } else {
            w_cur += stage_bytes;
        }
    };
    // ---- halo prefetch stream: exactly NA pieces per wave per chunk; clipped / padded pixels come back as zeros
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
            const int hp = (q * NW + wave) * 16 + lq;
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
    auto a_begin = [&]() {  // the next chunk of the stream moves on to the next source / tile where one ends
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
    };
    auto a_pieces = [&](auto q0c, auto nc) {
        constexpr int Q0 = decltype(q0c)::value, NQ = decltype(nc)::value;
        const __amdgpu_buffer_rsrc_t rs = mk(a_ptr);
        const int dst = a_base + ia_slot * ABUF;
        if (a_crem >= 32) {
#pragma unroll
            for (int q = Q0; q < Q0 + NQ; ++q) {
                const int j = q * NW + wave;
                bdma16(rs, a_voff[q], a_soff, (void*)(lds + (j < npieces ? dst + j * 1024 : dummy_base)));
            }
        } else {  // partial last chunk of a source whose channel count is not a multiple of 32: the missing channels read as zeros
#pragma unroll
            for (int q = Q0; q < Q0 + NQ; ++q) {
                const int j = q * NW + wave;
                const int hp = j * 16 + lq;
                const int kg8 = ((lane & 3) ^ ((hp >> 1) & 2)) * 8;
                bdma16(rs, kg8 < a_crem ? a_voff[q] : RSU_SENT, a_soff, (void*)(lds + (j < npieces ? dst + j * 1024 : dummy_base)));
            }
        }
    };
    auto a_end = [&]() {
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
            if (stamp_i < PP_NSTAMP) {
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
                if constexpr (STAMP)  // (the diagnostic build spills descriptor words: they come back through v_readlane right in front of the store)
                    asm volatile("s_nop 4\n\tbuffer_store_dwordx4 %0, %1, %2, %3 offen\n\ts_nop 3" ::"v"(r), "v"(voff), "s"(orsrc), "s"(sbase) : "memory");
                else
                    asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen\n\ts_nop 3" ::"v"(r), "v"(voff), "s"(orsrc), "s"(sbase) : "memory");
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
                // store + its wait states as ONE asm statement (the >64-bit store / VALU-write hazard)
                if constexpr (STAMP)
                    asm volatile("s_nop 4\n\tbuffer_store_dwordx4 %0, %1, %2, %3 offen\n\ts_nop 3" ::"v"(r), "v"(voffs[e]), "s"(orsrc), "s"(sbase)
                                 : "memory");
                else
                    asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen\n\ts_nop 3" ::"v"(r), "v"(voffs[e]), "s"(orsrc), "s"(sbase)
                                 : "memory");
            }
            __builtin_amdgcn_sched_barrier(0);  // keep the batches apart
        }
    };

    // ---- prologue: W(0), A(0), W(1); everybody waits for its share of W(0) and A(0), one barrier publishes them
    setup_a(ptile, 0);
    a_next_src = p.nchunk[0];
    issue_w(0);
    a_begin();
    a_pieces(std::integral_constant<int, 0>{}, std::integral_constant<int, NA>{});
    a_end();
    if (GC * 3 > 1) {
        issue_w(1);
        RSU_WAIT_VMCNT(WPS);
    } else {
        RSU_WAIT_VMCNT(0);
    }
    // (raw barrier: __syncthreads() would drain the LDS-DMA stream; the bias words above are the only ordinary LDS stores)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (wave >= NW / 2) {  // G1 sits out interval 0
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
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
            const bool steady = gc + 1 < GC;          // the issues of this chunk's R intervals all take place
            const bool after_epi = (c == 0) && gc > 0;  // the NST stores of the previous tile sit in front of this chunk's issues
            const int ab = a_base + ca_slot * ABUF;
            auto phase = [&](auto ppc) {
                constexpr int PP = decltype(ppc)::value;
                constexpr int J = PP / 3, TL = PP % 3;
                const int st = gc * 3 + J;
                // ================= R interval: fragment reads of this phase + bookkeeping for the phases ahead
                bf16x8 fa[CT], fb[PT];
                {
                    constexpr int ky = PP / KW, kx = PP - ky * KW;
                    const int rowoff = ab + ((ky * CW * p.dil) << 6);  // wave-uniform
#pragma unroll
                    for (int ct = 0; ct < CT; ++ct)
                        fa[ct] = *(const __attribute__((address_space(3))) bf16x8*)(lds + J * WBUF + ((TL * WT + wco * CT + ct) * 64 + lane) * 16);
#pragma unroll
                    for (int pt = 0; pt < PT; ++pt)
                        fb[pt] = *(const __attribute__((address_space(3))) bf16x8*)(lds + (boff[pt][kx] + rowoff));
                }
                if constexpr (TL == 1) {
                    if (st + DW < GC * 3 && !(p.dbg & 1)) issue_w((J + DW) % NWB);
                }
                if constexpr (a_cnt(NA, PP) > 0) {
                    if (gc + 1 < GC && !(p.dbg & 2)) {
                        if constexpr (a_first(NA, PP) == 0) a_begin();
                        a_pieces(std::integral_constant<int, a_first(NA, PP)>{}, std::integral_constant<int, a_cnt(NA, PP)>{});
                        if constexpr (a_first(NA, PP) + a_cnt(NA, PP) == NA) a_end();
                    }
                }
                if constexpr (TL == 2) {
                    // needed: the weights of the next stage (issued four phases ago) and, in the chunk's last phase, the next halo
                    constexpr int ALLOWED = PP == 2 ? WPS + a_cnt(NA, 2)
                                                    : (PP == 5 ? a_cnt(NA, 2) + a_cnt(NA, 3) + WPS + a_cnt(NA, 5) : WPS);
                    if (!steady) {
                        RSU_WAIT_VMCNT(0);
                    } else if (PP == 2 && after_epi) {
                        RSU_WAIT_VMCNT(ALLOWED + NST);
                    } else {
                        RSU_WAIT_VMCNT(ALLOWED);
                    }
                }
                stamp();
                asm volatile("" ::: "memory");
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                stamp();
                // ================= M interval: the MFMAs of this phase, nothing else
                __builtin_amdgcn_sched_barrier(0);
                if (!(p.dbg & 16)) __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int pt = 0; pt < PT; ++pt)
#pragma unroll
                    for (int ct = 0; ct < CT; ++ct) mfma_bf16_inplace(acc[ct][pt], fa[ct], fb[pt]);
                if (!(p.dbg & 16)) __builtin_amdgcn_s_setprio(0);
                __builtin_amdgcn_sched_barrier(0);
                if (PP == 8 && c == nchunks - 1) mfma_results_fence();  // straight behind the tile's last MFMA
                stamp();
                asm volatile("" ::: "memory");
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                stamp();
            };
            phase(std::integral_constant<int, 0>{});
            phase(std::integral_constant<int, 1>{});
            phase(std::integral_constant<int, 2>{});
            phase(std::integral_constant<int, 3>{});
            phase(std::integral_constant<int, 4>{});
            phase(std::integral_constant<int, 5>{});
            phase(std::integral_constant<int, 6>{});
            phase(std::integral_constant<int, 7>{});
            phase(std::integral_constant<int, 8>{});
            ca_slot ^= 1;
        }
        // the finished tile's epilogue opens the wave's next R interval (its partner is in an M interval meanwhile)
        if (!(p.dbg & 8)) epilogue(ctile, acc);
    }
    if (wave < NW / 2) {  // G0 sits out the last interval (G1's last epilogue)
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    if constexpr (STAMP) {
        if (p.stamps) {
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            for (int i = lane; i < PP_NSTAMP; i += 64)
                p.stamps[((long)blockIdx.x * NW + wave) * PP_NSTAMP + i] =
                    i < stamp_i ? *(__attribute__((address_space(3))) unsigned*)(lds + stamp_base + (wave * PP_NSTAMP + i) * 4) : 0u;
        }
    }
}

// ---------------------------------------------------------------------------------------------
template <int CFG> struct PpCfg;
// NA = halo DMA pieces per wave per chunk (NA * 8 * 16 halo pixels at most), as in igemm_fwd2
template <> struct PpCfg<IGF2_CFG_128x256> { static constexpr int WCO = 2, WPX = 4, CT = 4, PT = 4, NA = 4; };
template <> struct PpCfg<IGF2_CFG_64x512> { static constexpr int WCO = 1, WPX = 8, CT = 4, PT = 4, NA = 6; };
template <> struct PpCfg<IGF2_CFG_128x128> { static constexpr int WCO = 2, WPX = 4, CT = 4, PT = 2, NA = 3; };
template <> struct PpCfg<IGF2_CFG_64x256> { static constexpr int WCO = 1, WPX = 8, CT = 4, PT = 2, NA = 4; };
template <> struct PpCfg<IGF2_CFG_128x192> { static constexpr int WCO = 2, WPX = 4, CT = 4, PT = 3, NA = 4; };
template <> struct PpCfg<IGF2_CFG_64x384> { static constexpr int WCO = 1, WPX = 8, CT = 4, PT = 3, NA = 5; };
template <> struct PpCfg<IGF2_CFG_128x320> { static constexpr int WCO = 2, WPX = 4, CT = 4, PT = 5, NA = 5; };
template <> struct PpCfg<IGF2_CFG_64x640> { static constexpr int WCO = 1, WPX = 8, CT = 4, PT = 5, NA = 7; };

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
