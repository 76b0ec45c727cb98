// igemm_fwd2: persistent, deeply pipelined version of the NHWC bf16 implicit-GEMM convolution (see igemm_fwd.hip for
// the data layout: fragment-ordered weights, pixel-major swizzled halo tile, channel-permuted 16-byte stores).
//
// What changes against igemm_fwd:
//   * one workgroup per CU walks a list of output tiles; the (chunk, tap-group) STAGE STREAM runs continuously across
//     tile boundaries, so the first stages of the next tile are already in flight while the current tile finishes
//     (no exposed prologue, the epilogue is the only per-tile bubble);
//   * weights are prefetched TWO stages ahead into a 3-slot LDS ring, the halo tile of the next chunk one chunk
//     ahead into a 2-slot ring; nothing ever drains: each wave waits with a COUNTED s_waitcnt vmcnt(N) for exactly
//     the LDS-DMA loads its next stage needs, then one raw s_barrier per stage publishes everybody's pieces;
//   * every wave issues a CONSTANT number of LDS-DMA instructions per stage position (halo pieces are padded with
//     loads of the zero page into a scratch slot), which is what makes N an immediate;
//   * epilogue stores go through a buffer descriptor with out-of-range offsets for masked lanes: always the same
//     number of VMEM instructions per tile, so the counters stay exact across tiles (CDNA4 vmcnt counts stores too).
#include <type_traits>

#include "igemm.h"

#define RSU_WAIT_VMCNT(N) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory")

template <int SPC, int WPS, int NA, int DA, int DW>
__device__ __forceinline__ constexpr int vm_allowed(int j) {
    // loads that may still be in flight when stage position j starts (issue order: see the kernel body).
    // SPC == 1 (one stage per chunk, halo DA chunks ahead, weights DA+1 stages ahead): everything issued after the halo of
    // this stage = the weights of that same stage position plus the complete issues of the DA-1 stages since
    // SPC > 1 (halo one chunk ahead, issued behind the weights of stage position 0; weights DW <= SPC stages ahead): the weights
    // of the DW-1 stages since, plus the next chunk's halo unless this is position 0 (whose own halo is the youngest load it needs)
    return SPC == 1 ? WPS + (DA - 1) * (NA + WPS) : (j == 0 ? (DW - 1) * WPS : (DW - 1) * WPS + NA);
}

// buffer -> LDS copy of 16 bytes per lane: out-of-range lanes (voffset + soffset >= num_records) deposit ZEROS, which is
// exactly the zero padding / window clipping the halo tile needs (probes/probe_buffer_lds.hip pins the semantics)
__device__ __forceinline__ void bdma16(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned soff, void* lds_wave_base) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds_wave_base, 16, voff, soff, 0, 0);
}
__device__ __forceinline__ unsigned relu_pk_bf16(unsigned x) {
    typedef __attribute__((ext_vector_type(2))) short s2;
    s2 h = __builtin_bit_cast(s2, x);
    h = __builtin_elementwise_max(h, s2{0, 0});
    return __builtin_bit_cast(unsigned, h);
}
#define RSU_SENT 0x80000000u   // voffset that the range check always rejects (num_records = 0x7fffffff)

// Tiles are ALIGNED: strip width SW = 2^lsw divides TM, a tile is TR = TM/SW full rows of one strip. Every per-lane
// offset (fragment reads, halo pieces, output pixels) is therefore a workgroup constant; a tile only contributes
// scalar bases (soffset) and edge validity.
// NW = WCO*WPX waves: 4 (one per SIMD, big wave tiles) or 8 (two per SIMD: one wave's address arithmetic, waits and
// LDS latency overlap its partner's MFMAs)
// NAB = halo ring slots (prefetch distance NAB-1 chunks), NAB+1 weight slots (distance NAB stages). 2 for the 3x3 kernels
// (a chunk is three stages long, one chunk ahead is plenty); deeper for the one-stage-per-chunk kernels (1x1 / transposed
// conv), whose stages are too short to cover a load's latency
template <int WCO, int WPX, int CT, int PT, int NTAP, int KW, int TPS, int NA, int NAB, int DWS>
__global__ void __launch_bounds__(WCO* WPX * 64) __attribute__((amdgpu_waves_per_eu(WCO* WPX / 4, WCO* WPX / 4)))
igemm_fwd2_kernel(const IgFwdParams p) {
    constexpr int NW = WCO * WPX;
    static_assert(NW == 8, "eight waves: two per SIMD");
    constexpr int TN = WCO * CT * 16, TM = WPX * PT * 16;
    constexpr int WT = TN / 16;
    constexpr int SPC = NTAP / TPS;
    constexpr int WBUF = TPS * WT * 1024;
    constexpr int DA = NAB - 1, DW = DWS;      // prefetch distances: halo (chunks), weights (stages)
    constexpr int NWB = DW + 1;                // weight ring slots
    constexpr int WPS = (TPS * WT + NW - 1) / NW;  // weight DMA instructions per wave per stage (padded to a constant)
    constexpr int NST = (CT / 2) * PT;         // epilogue buffer stores per wave per tile (always issued)
    static_assert(NTAP % TPS == 0 && (CT % 2) == 0 && SPC <= 3, "bad config");
    static_assert(SPC == 1 || NAB == 2, "deep halo rings only for one-stage-per-chunk kernels");
    static_assert(SPC == 1 ? DW == NAB : (DW >= 2 && DW <= SPC), "weight distance");
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
    const int ph = blockIdx.y;
    const bf16_t* wp = p.wp + (long)ph * p.wp_y_stride;
    const int ooffy = ph >> 1, ooffx = ph & 1;

    // ---- this workgroup's tile list: fixed channel block, m-tiles first, first+stride, ...
    // XCD-aware numbering: blocks whose ids agree modulo 8 are observed to share an XCD (and its L2). Renumber so that such a group
    // owns a contiguous run of (pixel tile, channel block) pairs: the channel blocks of one pixel tile read the same halo, and
    // neighbouring tiles overlap in theirs. Speed only -- any placement computes the same values. (dbg bit 6: plain numbering)
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
        const int hp0 = ty * p.stride * CW + tx * p.stride;
#pragma unroll
        for (int kx = 0; kx < KW; ++kx) {
            const int hp = hp0 + kx * p.dil;
            boff[pt][kx] = (hp << 6) + ((g4 ^ ((hp >> 1) & 2)) << 4);  // CW % 8 == 0: a ky shift keeps the swizzle
        }
    }
    const int npieces = p.g.npix_max >> 4;
    const int lq = lane >> 2;
    // buffer descriptors are rebuilt from the (scalar) kernel-argument pointers right where they are used: hoisting them
    // makes hipcc park them in VGPRs/scratch and wrap every buffer op in a waterfall loop (cdna guide T20)
    auto mk = [&](const void* ptr) { return __builtin_amdgcn_make_buffer_rsrc((void*)ptr, 0, 0x7fffffff, 0x00020000); };

    // ---- weight prefetch stream. Stage s of a tile reads the contiguous block [s*TPS .. s*TPS+TPS) x [all tiles] of the
    // packed weights, so the source is one scalar pointer that advances by a constant per stage and rewinds per tile;
    // the pieces a wave copies have per-wave constant offsets. No divisions, ~5 scalar ops per piece.
    const int nstage_tile = nchunks * SPC;
    const long stage_bytes = (long)TPS * p.ntiles_w * 1024;
    const char* const w_tile_base = (const char*)wp + (long)(p.tile_off + cob * WT) * 1024;
    const char* w_cur = w_tile_base;   // source of the next stage to prefetch
    int w_sit = 0;                     // its stage index inside the tile
    int w_slot = 0;                    // its ring slot
    int wpo[WPS];                      // byte offset of piece q inside a stage block, or -1 (padding piece / beyond the packed rows)
#pragma unroll
    for (int q = 0; q < WPS; ++q) {
        const int i = q * NW + wave;
        const int tap_l = i / WT, tl = i - tap_l * WT;
        const bool real = (i < TPS * WT) && (p.tile_off + cob * WT + tl < p.ntiles_w);
        wpo[q] = real ? (tap_l * p.ntiles_w + tl) * 1024 : -1;
    }
    auto issue_w = [&]() {
        const int dst = w_slot * WBUF;
#pragma unroll
        for (int q = 0; q < WPS; ++q) {
            const int i = q * NW + wave;
            const char* base = wpo[q] >= 0 ? w_cur + wpo[q] : (const char*)p.zero_page;
            dma16(base + lane * 16, (void*)(lds + (i < TPS * WT ? dst + i * 1024 : dummy_base)));
        }
        w_slot = w_slot == NWB - 1 ? 0 : w_slot + 1;
        if (++w_sit == nstage_tile) {
            w_sit = 0;
            w_cur = w_tile_base;
        } else {
            w_cur += stage_bytes;
        }
    };
    // ---- halo prefetch stream: exactly NA pieces per wave per chunk; clipped / padded pixels come back as zeros.
    // Everything that depends only on (tile, source) is computed once per (tile, source) by setup_a and carried in registers:
    // the per-lane byte offsets of the NA pieces (already RSU_SENT where the pixel falls outside the window), the source base
    // and the scalar offset of the halo origin. A chunk then costs one scalar add and, per piece, an M0 write and the DMA.
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
        a_soff = (unsigned)((((long)(T.n * sH + T.y0 * p.stride + soy) * sW + (T.x0 * p.stride + sox)) * sC) * 2);
        a_crem = sC;
        const int iy0 = T.y0 * p.stride - p.pad, ix0 = T.x0 * p.stride - p.pad;
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
    auto issue_a = [&]() {
        const __amdgpu_buffer_rsrc_t rs = mk(a_ptr);
        const int dst = a_base + ia_slot * ABUF;
        ia_slot = ia_slot == NAB - 1 ? 0 : ia_slot + 1;
        if (a_crem >= 32) {
#pragma unroll
            for (int q = 0; q < NA; ++q) {
                const int j = q * NW + wave;
                bdma16(rs, a_voff[q], a_soff, (void*)(lds + (j < npieces ? dst + j * 1024 : dummy_base)));
            }
        } else {  // partial last chunk of a source whose channel count is not a multiple of 32: the missing channels read as zeros
#pragma unroll
            for (int q = 0; q < NA; ++q) {
                const int j = q * NW + wave;
                const int hp = j * 16 + lq;
                const int kg8 = ((lane & 3) ^ ((hp >> 1) & 2)) * 8;
                bdma16(rs, kg8 < a_crem ? a_voff[q] : RSU_SENT, a_soff, (void*)(lds + (j < npieces ? dst + j * 1024 : dummy_base)));
            }
        }
        a_soff += 64;
        a_crem -= 32;
        a_cl = a_cl + 1 == nchunks ? 0 : a_cl + 1;
    };

    // bias of this workgroup's TN channels lives in LDS (behind the scratch slot): read back at every tile start with
    // ds_read (lgkmcnt) -- keeping it in registers costs 16 VGPRs, re-loading it from memory would touch vmcnt
    const int bias_base = dummy_base + 1024;
    if (threadIdx.x < TN) {
        const int co = cob * TN + threadIdx.x;
        const float bvv = (p.bias && co < p.Cout) ? p.bias[co] : 0.f;
        *(__attribute__((address_space(3))) float*)(lds + bias_base + threadIdx.x * 4) = bvv;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // retire the ordinary loads before the LDS-DMA stream starts
    __syncthreads();

    // scalar byte offset of a tile's first output pixel (+ this wave's first channel) and per-lane offset of epilogue store e
    auto out_sbase = [&](const Tile& T) {
        return (unsigned)((((long)(T.n * p.oH + T.y0 * p.ostride + ooffy) * p.oW + T.x0 * p.ostride + ooffx) * p.outC + cob * TN +
                           wco * (CT / 2) * 32) * 2);
    };
    auto out_voff = [&](const Tile& T, int e) {
        const int pt = e / (CT / 2), pp = e % (CT / 2);
        const int ml = (wpx * PT + pt) * 16 + l15;
        const int ty = ml >> lsw, tx = ml & (SW - 1);
        const bool pok = (T.y0 + ty < p.Ho) && (T.x0 + tx < p.Wo);
        const int ovoff_pt = ((ty * p.ostride * p.oW + tx * p.ostride) * p.outC + 8 * g4) * 2;
        const int co = cob * TN + (wco * (CT / 2) + pp) * 32 + 8 * g4;
        return (pok && co < p.Cout) ? (unsigned)(ovoff_pt + pp * 64) : RSU_SENT;
    };
    auto epilogue = [&](const Tile& T, f32x4(&acc)[CT][PT]) {
        const __amdgpu_buffer_rsrc_t orsrc = mk(p.out);
        const __amdgpu_buffer_rsrc_t mrsrc = mk(p.mask_src ? (const void*)p.mask_src : (const void*)p.out);
        const unsigned sbase = out_sbase(T);
        if (!p.accumulate) {
            // forward and backward-data without an AddN, without a single branch per store: everything on the packed bf16 result.
            // ReLU is a packed int16 max against 0 or, switched off, against the most negative int16 (bf16 sign bit == int16 sign
            // bit); the ReLU mask of backward-data (relu_src > 0) is a packed 0 / 0xffff word ANDed onto it: max(x, 0) -> min(., 1)
            // -> 0 - . (a positive NaN in relu_src counts as > 0 here; no finite activation is affected). The mask loads go out in
            // batches of four (one memory latency per batch)
            typedef __attribute__((ext_vector_type(2))) short s2;
            const short fl = p.relu ? (short)0 : (short)-32768;
            const s2 floor2 = {fl, fl};
            unsigned ones_pk = 0x00010001u;
            asm volatile("" : "+v"(ones_pk));
            constexpr int MB = NST % 4 == 0 ? 4 : 2;
#pragma unroll
            for (int b0 = 0; b0 < NST; b0 += MB) {
                unsigned voffs[MB];
                u32x4 mk4[MB];
#pragma unroll
                for (int e = 0; e < MB; ++e) {
                    voffs[e] = out_voff(T, b0 + e);
                    if (p.mask_src) mk4[e] = __builtin_amdgcn_raw_buffer_load_b128(mrsrc, voffs[e], sbase, 0);
                }
#pragma unroll
                for (int e = 0; e < MB; ++e) {
                    const int pt = (b0 + e) / (CT / 2), pp = (b0 + e) % (CT / 2);
                    u32x4 r;
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        const unsigned lo = pack_bf2(acc[2 * pp][pt][2 * i], acc[2 * pp][pt][2 * i + 1]);
                        const unsigned hi = pack_bf2(acc[2 * pp + 1][pt][2 * i], acc[2 * pp + 1][pt][2 * i + 1]);
                        r[i] = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s2, lo), floor2));
                        r[2 + i] = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s2, hi), floor2));
                    }
                    if (p.mask_src) {
#pragma unroll
                            for (int i = 0; i < 4; ++i) r[i] &= pos_mask_pk_bf16(mk4[e][i], ones_pk);
                    }
                    asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen\n\ts_nop 3" ::"v"(r), "v"(voffs[e]), "s"(orsrc), "s"(sbase) : "memory");
                }
                if (p.mask_src) __builtin_amdgcn_sched_barrier(0);  // keep the batches apart
            }
            return;
        }
        // NST stores in batches of four: the mask / accumulate loads of a batch are requested together (one memory latency per
        // batch instead of one per store); larger batches would need more than the 16-32 VGPRs that are free here
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
                if (p.relu) {  // ReLU on the packed result: bf16 sign bit == int16 sign bit (v_pk_max_i16)
                    r0 = relu_pk_bf16(r0);
                    r1 = relu_pk_bf16(r1);
                    r2 = relu_pk_bf16(r2);
                    r3 = relu_pk_bf16(r3);
                }
                const u32x4 r = {r0, r1, r2, r3};
                // the store's data registers are dead from here on and the compiler reuses them at once (a VALU write in the very
                // next instruction); with an SGPR offset it sees no hazard in that, but the last lanes of the 128-bit store were
                // observed to pick up the NEW value. Store and four wait states are ONE asm statement, so
                // nothing can be scheduled in between. (The descriptor and offset SGPRs must not come from a VALU instruction --
                // v_readlane of a spilled word -- within five wait states: rule (4) of tools/check_mfma_hazards.py watches that.)
                asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen\n\ts_nop 3" ::"v"(r), "v"(voffs[e]), "s"(orsrc), "s"(sbase)
                             : "memory");
            }
            __builtin_amdgcn_sched_barrier(0);  // keep the batches apart
        }
    };

    // ---- prologue: A(0), W(0), W(1)
    Tile ptile = decode(0);     // tile whose halo is being prefetched
    int pk = 0;                 // index of ptile in this workgroup's list
    setup_a(ptile, 0);
    a_next_src = p.nchunk[0];
    bool a_started = false;
    auto issue_a_next = [&]() {  // halo of the next chunk of the stream (moves on to the next source / tile where one ends)
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
        issue_a();
    };
    if constexpr (SPC == 1) {
        // the issues of the DW stages "before" stage 0, in steady-state order (halo of stage k+DA, then weights of stage k+DW)
#pragma unroll
        for (int k = -DW; k < 0; ++k) {
            if (k + DA >= 0 && k + DA < GC) issue_a_next();
            if (k + DW < GC) issue_w();
        }
    } else {
        issue_w();
        issue_a_next();
#pragma unroll
        for (int k = 1; k < DW; ++k)
            if (k < GC * SPC) issue_w();
    }

    int gc = 0;      // stream chunk counter
    int c_slot = 0;  // weight ring slot of the stage being computed
    int ca_slot = 0; // halo ring slot of the chunk being computed
    for (int ck = 0; ck < my_tiles; ++ck) {
        const Tile ctile = decode(ck);
        // accumulators live for exactly one tile (no loop-carried copies across the epilogue); they start at the bias
        f32x4 acc[CT][PT];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            const f32x4 bv = *(const __attribute__((address_space(3))) f32x4*)(lds + bias_base +
                                                                                 ((wco * (CT / 2) + (ct >> 1)) * 32 + 8 * g4 + (ct & 1) * 4) * 4);
#pragma unroll
            for (int pt = 0; pt < PT; ++pt) {
                acc[ct][pt] = bv;
                // materialise the copy HERE: left to itself the compiler sinks it to just in front of the first (inline-asm) MFMA
                // that reads it, closer than the matrix pipe tolerates (tools/check_mfma_hazards.py)
                asm volatile("" : "+v"(acc[ct][pt]));
            }
        }
        for (int c = 0; c < nchunks; ++c, ++gc) {
            // every wave may rely on the constant per-stage counts only while the two chunks ahead exist
            const bool steady = SPC == 1 ? (gc + DW < GC && nchunks >= DA) : (gc + 2 < GC);
            // first stages after an epilogue: its 16-byte stores sit in the VMEM queue behind the loads these stages wait for
            const bool after_epi = SPC == 1 ? (c < DA && ck > 0) : ((c == 0) && gc > 0);
            auto stage = [&](auto jc) {
                constexpr int j = decltype(jc)::value;
                const int st = gc * SPC + j;
                constexpr int ALLOWED = vm_allowed<SPC, WPS, NA, DA, DW>(j);
                if (!steady) {
                    RSU_WAIT_VMCNT(0);
                } else if (after_epi && j < DW) {
                    RSU_WAIT_VMCNT(ALLOWED + NST);
                } else {
                    RSU_WAIT_VMCNT(ALLOWED);
                }
                asm volatile("" ::: "memory");
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                // ---- compute stage st from weight slot st%NWB and halo slot gc&1
                const int wb = c_slot * WBUF;
                const int ab = a_base + ca_slot * ABUF;
                bf16x8 fa[2][CT], fb[2][PT];
                auto load_tap = [&](int tl, bf16x8(&a)[CT], bf16x8(&b)[PT]) {
                    const int tap = j * TPS + tl;
                    const int ky = tap / KW, kx = tap - ky * KW;
                    const int rowoff = ab + ((ky * CW * p.dil) << 6);  // wave-uniform
#pragma unroll
                    for (int ct = 0; ct < CT; ++ct)
                        a[ct] = *(const __attribute__((address_space(3))) bf16x8*)(lds + wb + ((tl * WT + wco * CT + ct) * 64 + lane) * 16);
#pragma unroll
                    for (int pt = 0; pt < PT; ++pt)
                        b[pt] = *(const __attribute__((address_space(3))) bf16x8*)(lds + (boff[pt][kx] + rowoff));
                };
                auto prefetch = [&]() {
                // prefetch: (SPC == 1: halo first, then weights; otherwise weights, then halo at position 0)
                    if (SPC == 1) {
                        if (gc + DA < GC && !(p.dbg & 2)) issue_a_next();
                        if (st + DW < GC * SPC && !(p.dbg & 1)) issue_w();
                    } else {
                        if (st + DW < GC * SPC && !(p.dbg & 1)) issue_w();
                        if (j == 0 && gc + 1 < GC && !(p.dbg & 2)) issue_a_next();
                    }
                };
                load_tap(0, fa[0], fb[0]);
#pragma unroll
                for (int tl = 0; tl < TPS; ++tl) {
                    if (tl + 1 < TPS) load_tap(tl + 1, fa[(tl + 1) & 1], fb[(tl + 1) & 1]);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int pt = 0; pt < PT; ++pt)
#pragma unroll
                        for (int ct = 0; ct < CT; ++ct)
                            mfma_bf16_inplace(acc[ct][pt], fa[tl & 1][ct], fb[tl & 1][pt]);
                    // LDS-DMA bookkeeping for the stages ahead, placed BEHIND a block of MFMAs and staggered between the two
                    // waves of a SIMD (waves w and w+4): while one does scalar address work its partner feeds the matrix pipe
                    if (TPS == 1) {
                        if (tl == 0) prefetch();
                    } else {
                        if (tl == 0 && wave < NW / 2) prefetch();
                        if (tl == 1 && wave >= NW / 2) prefetch();
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            };
            stage(std::integral_constant<int, 0>{});
            c_slot = c_slot == NWB - 1 ? 0 : c_slot + 1;
            if constexpr (SPC > 1) {
                stage(std::integral_constant<int, 1>{});
                c_slot = c_slot == NWB - 1 ? 0 : c_slot + 1;
            }
            if constexpr (SPC > 2) {
                stage(std::integral_constant<int, 2>{});
                c_slot = c_slot == NWB - 1 ? 0 : c_slot + 1;
            }
            ca_slot = ca_slot == NAB - 1 ? 0 : ca_slot + 1;
            // the results fence sits INSIDE the loop, straight behind the tile's last MFMA: register copies the compiler makes on
            // the loop exit (they read the accumulators) then come after it
            if (c == nchunks - 1) mfma_results_fence();
        }
        if (!(p.dbg & 8)) epilogue(ctile, acc);  // dbg bit 3: timing experiment without the epilogue (results are not written)
    }
}

// ---------------------------------------------------------------------------------------------
template <int CFG> struct Fwd2Cfg;
// na(ntap) = halo DMA pieces per wave per chunk (every wave issues exactly that many: na * NW * 16 halo pixels at most). 3x3: a
// TM-pixel tile plus its halo; 2x2 stride 2 (transposed-conv backward-data): 4*TM pixels; 1x1: TM pixels
template <> struct Fwd2Cfg<IGF2_CFG_128x256> {
    static constexpr int WCO = 2, WPX = 4, CT = 4, PT = 4;
    static constexpr int na(int ntap) { return ntap == 1 ? 2 : 4; }
};
template <> struct Fwd2Cfg<IGF2_CFG_64x512> {
    static constexpr int WCO = 1, WPX = 8, CT = 4, PT = 4;
    static constexpr int na(int ntap) { return ntap == 1 ? 4 : 6; }
};
template <> struct Fwd2Cfg<IGF2_CFG_128x128> {
    static constexpr int WCO = 2, WPX = 4, CT = 4, PT = 2;
    static constexpr int na(int ntap) { return ntap == 1 ? 1 : (ntap == 4 ? 4 : 3); }
};
template <> struct Fwd2Cfg<IGF2_CFG_64x256> {
    static constexpr int WCO = 1, WPX = 8, CT = 4, PT = 2;
    static constexpr int na(int ntap) { return ntap == 1 ? 2 : 4; }
};
// five pixel fragments per wave (80 accumulator registers): a fifth fewer weight pieces and LDS reads per MFMA than 128x256
// three pixel fragments per wave: between the 256/512- and the 128/256-pixel shapes, for layers whose tile count fills the last
// round badly with either
template <> struct Fwd2Cfg<IGF2_CFG_128x192> {
    static constexpr int WCO = 2, WPX = 4, CT = 4, PT = 3;
    static constexpr int na(int ntap) { return ntap == 1 ? 2 : 4; }
};
template <> struct Fwd2Cfg<IGF2_CFG_64x384> {
    static constexpr int WCO = 1, WPX = 8, CT = 4, PT = 3;
    static constexpr int na(int ntap) { return ntap == 1 ? 3 : 5; }
};
template <> struct Fwd2Cfg<IGF2_CFG_128x320> {
    static constexpr int WCO = 2, WPX = 4, CT = 4, PT = 5;
    static constexpr int na(int ntap) { return ntap == 1 ? 3 : 5; }
};
template <> struct Fwd2Cfg<IGF2_CFG_64x640> {
    static constexpr int WCO = 1, WPX = 8, CT = 4, PT = 5;
    static constexpr int na(int ntap) { return ntap == 1 ? 5 : 7; }
};

static constexpr int tps2_for(int TN, int ntap) { return ntap == 9 ? 3 : (ntap == 4 ? (TN <= 64 ? 4 : 2) : 1); }
static constexpr int nab_for(int TN, int ntap) { return ntap / tps2_for(TN, ntap) == 1 ? 4 : 2; }
// weight prefetch distance in stages: = halo ring depth for the one-stage-per-chunk kernels, else 2 (3 was measured on the PT = 2
// 3x3 shapes: no gain -- their waits are barrier skew, not load latency)
static constexpr int dws_for(int TN, int ntap, int PT) { return ntap / tps2_for(TN, ntap) == 1 ? nab_for(TN, ntap) : 2; }

IgFwdCfgInfo igemm_fwd2_cfg_info(int cfg) {
    switch (cfg) {
#define CASE(C) \
    case C: return IgFwdCfgInfo{Fwd2Cfg<C>::WCO * Fwd2Cfg<C>::CT * 16, Fwd2Cfg<C>::WPX * Fwd2Cfg<C>::PT * 16, Fwd2Cfg<C>::WCO * Fwd2Cfg<C>::WPX * 64};
        CASE(IGF2_CFG_128x256)
        CASE(IGF2_CFG_64x512)
        CASE(IGF2_CFG_128x128)
        CASE(IGF2_CFG_64x256)
        CASE(IGF2_CFG_128x192)
        CASE(IGF2_CFG_64x384)
        CASE(IGF2_CFG_128x320)
        CASE(IGF2_CFG_64x640)
#undef CASE
    }
    return IgFwdCfgInfo{0, 0, 0};
}
int igemm_fwd2_max_pieces(int cfg, int ntap) {
    switch (cfg) {
#define CASE(C) case C: return Fwd2Cfg<C>::na(ntap) * Fwd2Cfg<C>::WCO * Fwd2Cfg<C>::WPX;
        CASE(IGF2_CFG_128x256)
        CASE(IGF2_CFG_64x512)
        CASE(IGF2_CFG_128x128)
        CASE(IGF2_CFG_64x256)
        CASE(IGF2_CFG_128x192)
        CASE(IGF2_CFG_64x384)
        CASE(IGF2_CFG_128x320)
        CASE(IGF2_CFG_64x640)
#undef CASE
    }
    return 0;
}
size_t igemm_fwd2_lds_bytes(int cfg, int ntap, int npix_max) {
    const IgFwdCfgInfo ci = igemm_fwd2_cfg_info(cfg);
    const int tps = tps2_for(ci.TN, ntap), nab = nab_for(ci.TN, ntap);
    const int pt = ci.TM / 16 / (ci.threads / 64 / (ci.TN / 64));  // PT = TM / 16 / WPX, WPX = waves / WCO, WCO = TN / 64
    const int dws = dws_for(ci.TN, ntap, pt);
    return (size_t)(dws + 1) * tps * (ci.TN / 16) * 1024 + (size_t)nab * npix_max * 64 + 1024 + 512;
}

template <int CFG, int NTAP, int KW>
static hipError_t launch2_one(const IgFwdParams& p, int gx, int gy, hipStream_t st) {
    using C = Fwd2Cfg<CFG>;
    constexpr int TN = C::WCO * C::CT * 16;
    constexpr int TPS = tps2_for(TN, NTAP);
    auto kern = igemm_fwd2_kernel<C::WCO, C::WPX, C::CT, C::PT, NTAP, KW, TPS, C::na(NTAP), nab_for(TN, NTAP), dws_for(TN, NTAP, C::PT)>;
    const size_t lds = igemm_fwd2_lds_bytes(CFG, NTAP, p.g.npix_max);
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
static hipError_t launch2_cfg(int ntap, const IgFwdParams& p, int gx, int gy, hipStream_t st) {
    switch (ntap) {
        case 9: return launch2_one<CFG, 9, 3>(p, gx, gy, st);
        case 4: return launch2_one<CFG, 4, 2>(p, gx, gy, st);
        case 1: return launch2_one<CFG, 1, 1>(p, gx, gy, st);
    }
    return hipErrorInvalidValue;
}
hipError_t igemm_fwd2_launch(int cfg, int ntap, const IgFwdParams& p, int gx, int gy, hipStream_t st) {
    switch (cfg) {
        case IGF2_CFG_128x256: return launch2_cfg<IGF2_CFG_128x256>(ntap, p, gx, gy, st);
        case IGF2_CFG_64x512: return launch2_cfg<IGF2_CFG_64x512>(ntap, p, gx, gy, st);
        case IGF2_CFG_128x128: return launch2_cfg<IGF2_CFG_128x128>(ntap, p, gx, gy, st);
        case IGF2_CFG_64x256: return launch2_cfg<IGF2_CFG_64x256>(ntap, p, gx, gy, st);
        case IGF2_CFG_128x192: return launch2_cfg<IGF2_CFG_128x192>(ntap, p, gx, gy, st);
        case IGF2_CFG_64x384: return launch2_cfg<IGF2_CFG_64x384>(ntap, p, gx, gy, st);
        case IGF2_CFG_128x320: return launch2_cfg<IGF2_CFG_128x320>(ntap, p, gx, gy, st);
        case IGF2_CFG_64x640: return launch2_cfg<IGF2_CFG_64x640>(ntap, p, gx, gy, st);
    }
    return hipErrorInvalidValue;
}
