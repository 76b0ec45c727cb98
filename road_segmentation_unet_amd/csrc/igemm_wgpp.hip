// igemm_wgpp: the 3x3 weight-gradient implicit GEMM of igemm_wgrad.hip (128 F channels x 64 S channels x 9 taps per workgroup,
// pixel tiles of 128, same LDS images, same slabs, same summation order: bit-identical results) run as a PING-PONG between the two
// waves of every SIMD, like igemm_pp.hip does for the forward convolution.
//
// igemm_wgrad's eight waves each read, multiply, compute staging addresses and issue LDS-DMA in one stream: 2.1 non-MFMA vector
// instructions per MFMA, the matrix pipe ~35-44 % busy. Here a PHASE is one 32-pixel k-step of a pixel tile: 26 transposed LDS reads
// (4 F fragments, 9 tap-shifted S fragments) and 36 MFMAs (+4 for the bias sums) per wave. Waves 0-3 (G0: F channels 0-63) and
// waves 4-7 (G1: F channels 64-127; wave w+4 shares its SIMD with wave w) run  R(0) | M(0) | R(1) | M(1) ...  ('|' = workgroup
// barrier) one interval apart: in every interval one wave of a SIMD does nothing but MFMAs while its partner reads the next k-step's
// fragments and stages the next tile. Geometry (strip width 2^LSW) is a template constant and the loop is unrolled over the two
// staging slots, so every LDS address is a per-lane register plus an immediate.
//
// Staging: 2 slots (pixel tile t in slot t & 1). Every wave ends an R interval with s_waitcnt lgkmcnt(0), so a slot may be refilled
// from the interval after its last reader's R: tile t+1 goes into the slot of tile t-1 during the R intervals of tile t's k-steps
// 0, 1, 2 (three pieces per wave each); the waves wait for all of their pieces (vmcnt(0): nothing else is in flight) at the end of
// G1's R(3) / behind G0's MFMAs of M(3) -- the same global interval, whose barrier publishes the tile to G0's next R(0).
#include <type_traits>

#include "igemm_wgrad_body.h"

#define RSU_SENT 0x80000000u

namespace {
// SCH (phases per 32-pixel k-step and what they hold): 0 = two, taps 0-4 (+ the F fragments, the bias sums) | taps 5-8; 3 = two, taps
// 0-3 | 4-8; 4 = one phase of all nine taps. Pieces (of the 9 a wave stages per tile) issued in the R interval of phase ph:
constexpr int sched_nph(int sch) { return sch >= 4 ? 1 : 2; }
constexpr int sched_count(int sch, int ph) {
    constexpr int T2[8] = {0, 3, 0, 3, 0, 3, 0, 0}, T1[4] = {3, 3, 3, 0};
    return sch >= 4 ? T1[ph] : T2[ph];
}
constexpr int sched_first(int sch, int ph) {
    int n = 0;
    for (int i = 0; i < ph; ++i) n += sched_count(sch, i);
    return n;
}
// WGPP_NT / WG64_NT (developer A/B switches, round 6; VERDICT r5 item 1 "try the nt policy on the weight-gradient kernels' once-read activation
// pieces"): the operand streams of igemm_wgpp / igemm_wgp64 loaded with the non-temporal hint, so that they displace less of what the
// backward-data kernel on the same XCD keeps in its L2. Measured: profiles/r06/ab_wgnt.txt.
#ifndef WGPP_NT
#define WGPP_NT 0
#endif
#ifndef WG64_NT
#define WG64_NT 0
#endif
template <int NT = 0>
__device__ __forceinline__ void bdma16w(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned soff, void* lds_wave_base) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds_wave_base, 16, voff, soff, 0, NT ? 2 : 0);
}
}  // namespace

// DBG (timing ablations, RSU_WG_DBG): 1 = no staging after the first tile, 2 = no LDS reads, 4 = no MFMAs, 8 = S reads of taps 0-2 only
// SCH: phases per k-step and where its taps are split (sched_nph)
// (the kernel body as a device function: `lds` = the workgroup's dynamic LDS, (cfb, csb, z) = the unit it works on; igemm_wgpp_kernel
// calls it once per workgroup, the grouped launch igemm_wg_group_kernel once per unit of its workgroup's list)
template <int LSW, int DBG, int SCH, class P>
__device__ __forceinline__ void igemm_wgpp_body(const P& p, __attribute__((address_space(3))) char* lds, const int cfb, const int csb,
                                                const int z, const unsigned tid) {
    constexpr int NW = 8, NTAP = 9, KW = 3, TMK = 128, CFT = 4;
    constexpr int SW = 1 << LSW, TR = TMK >> LSW;
    constexpr int CW = (SW + 2 + 7) / 8 * 8;
    constexpr int NPIX = ((TR + 2) * CW + 31) / 32 * 32;   // S halo pixels (plan_geo_aligned)
    constexpr int NSW = ((NPIX + 7) / 8 + 7) / 8;          // S pieces (8 pixels x 128 bytes) per wave per tile
    constexpr int NFW = 4;                                 // F pieces per wave per tile: 2 planes x 128 px / 8 px / 8 waves
    constexpr int FPL = TMK * 128, FBUF = 2 * FPL, SBUF = NSW * NW * 1024;
    constexpr int SLOT = FBUF + SBUF;                      // LDS: [F tile | S halo tile] x 2 slots
    static_assert(2 * SLOT <= 160 * 1024 && NFW + NSW <= 9, "staging budget");

    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, wcs = wave & 3;   // wcf == grp
    const int g4 = lane >> 4, l15 = lane & 15, q4 = l15 >> 2, p4 = lane & 3;
    const int tpi = p.g.nstrips * p.g.tiles_per_strip;
    // (the S window is the F image plus a border of one: 3x3 taps, stride 1, dilation 1)

    f32x4 acc[NTAP][CFT];
#pragma unroll
    for (int t = 0; t < NTAP; ++t)
#pragma unroll
        for (int a = 0; a < CFT; ++a) acc[t][a] = f32x4{0.f, 0.f, 0.f, 0.f};
    // BiasAddGrad rides along (F^T x ones, one MFMA per F tile and k-step), dealt over the four waves of a group: wave wcs keeps the
    // sums of F tile wcs -- one extra MFMA per k-step on every SIMD and 4 accumulator registers, not four MFMAs on one SIMD and 16.
    // So that this tile sits in a fixed register, a wave numbers its F tiles from its own: fragment / accumulator column i belongs to
    // F tile (i + wcs) & 3.
    const bool do_bias = (p.bslab != nullptr) && (csb == 0);
    f32x4 accb = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- per-lane LDS read offsets inside slot 0 (k-step 0, kernel row 0); slot, k-step and row are immediates
    // (addresses of the CURRENT slot: the offsets of slot 1 do not fit the 16-bit immediate of ds_read, so the registers move by
    // +-SLOT at every tile switch -- 14 additions per 8 phases -- instead of the loop being unrolled over the slots)
    int fo[2][CFT];   // F: [half of the k-step][cf tile]
    int so[2][KW];    // S: [half][kx]
#pragma unroll
    for (int rd = 0; rd < 2; ++rd) {
        const int ml = rd * 16 + 4 * g4 + q4;
        const int ch = grp * 64 + 4 * p4;   // F channel of tile 0 of this wave: plane grp, channel 4*p4 inside it
        const int f0 = (ch >> 6) * FPL + ml * 128 + ((((ch & 63) >> 4) ^ ((ml >> 1) & 3)) << 5) + (ch & 15) * 2;
#pragma unroll
        for (int ct = 0; ct < CFT; ++ct) fo[rd][ct] = f0 ^ (((ct + wcs) & 3) << 5);
        const int ty = ml >> LSW, tx = ml & (SW - 1);
        const int hp0 = ty * CW + tx;
#pragma unroll
        for (int kx = 0; kx < KW; ++kx) {
            const int hp = hp0 + kx;
            const int chs = wcs * 16 + 4 * p4;
            so[rd][kx] = FBUF + hp * 128 + ((((chs >> 4) ^ ((hp >> 1) & 3))) << 5) + (chs & 15) * 2;
        }
    }
    // S byte offset of k-step k inside the halo tile
    auto sdelta = [](int k) constexpr {
        const int m0 = k * 32;
        return ((m0 >> LSW) * CW + (m0 & (SW - 1))) * 128;
    };

    // ---- staging pieces of this wave: NFW of the F tile, NSW of the S halo tile (8 pixels x 64 channels = 1 KiB each, one LDS-DMA
    // instruction). The 8 pixels of a piece lie in one row of the tile (SW and CW are multiples of 8) and the channel swizzle depends
    // on the pixel's low bits only, so a lane's source offset splits into a per-lane part that is the same for every piece and tile
    // (lvF / lvS: one register each) and a wave-uniform part that rides in the instruction's scalar offset. Interior tiles need no
    // vector instruction per piece; edge tiles two compares and a select.
    const int l8 = lane >> 3;
    const int cswz8 = ((lane & 7) ^ (((l8 >> 1) & 3) << 1)) * 8;   // first channel (of the 64 of the block) this lane loads
    unsigned lvF = (unsigned)((l8 * p.Cf + cswz8) * 2), lvS = (unsigned)((l8 * p.S.C + cswz8) * 2);
    asm volatile("" : "+v"(lvF), "+v"(lvS));  // (keep them as registers: rematerialising costs vector instructions per piece)
    const int limcS = p.S.C - csb * 64, limcF = p.Cf - cfb * 128;   // channels of this block that exist (S: of 64, F: of 128)
    auto sgpr = [](auto v) { return __builtin_amdgcn_readfirstlane(v); };  // (wave-uniform by construction; said out loud so it stays scalar)
    // wave-uniform parts of the source offsets that do not depend on the tile: F piece I covers tile row wty + (I & 1) * (64 >> LSW)
    // from column wtx0 of channel plane I >> 1; S piece Q starts at halo pixel Q * 64 + wave * 8 = row rrQ, column ccQ
    const int wty = (wave * 8) >> LSW, wtx0 = (wave * 8) & (SW - 1);
    const unsigned rowF = sgpr((unsigned)(p.Wf * p.Cf * 2) * (64 >> LSW));
    const unsigned wbF = sgpr((unsigned)((wty * p.Wf + wtx0) * p.Cf * 2));
    unsigned wbS[NSW];
#pragma unroll
    for (int q = 0; q < NSW; ++q) {
        const int hp0 = q * 64 + wave * 8, rr = hp0 / CW, cc0 = hp0 - rr * CW;
        wbS[q] = sgpr((unsigned)((rr * p.S.W + cc0) * p.S.C * 2));
    }
    constexpr int HROWS = (NPIX - 1) / CW + 1;  // rows of the staged halo image (its last, partial row is rounding)
    // per tile: byte offsets of its F and S windows, whether the whole tile lies inside the tensors, and what is left of them from (y0, x0) on
    struct Tile { unsigned bF, bS; int insF, insS, ry, rx; };
    // this workgroup's tiles are z, z + nsplit, ...: (image, strip, row in strip) of the first one by division, then stepped -- a
    // run-time division costs ~40 vector instructions, which an R interval does not have
    struct Pos { int n, strip, row; };
    auto split = [&](int t) {
        Pos q;
        q.n = t / tpi;
        const int r = t - q.n * tpi;
        q.strip = r / p.g.tiles_per_strip;
        q.row = r - q.strip * p.g.tiles_per_strip;
        return Pos{sgpr(q.n), sgpr(q.strip), sgpr(q.row)};
    };
    const Pos step = split(p.nsplit);
    auto advance = [&](Pos& q) {
        q.row += step.row;
        if (q.row >= p.g.tiles_per_strip) { q.row -= p.g.tiles_per_strip; ++q.strip; }
        q.strip += step.strip;
        if (q.strip >= p.g.nstrips) { q.strip -= p.g.nstrips; ++q.n; }
        q.n += step.n;
    };
    auto decode = [&](const Pos& q) {
        Tile T;
        const int x0 = q.strip * SW, y0 = q.row * TR;
        T.bF = sgpr((unsigned)((((long)(q.n * p.Hf + y0) * p.Wf + x0) * p.Cf + cfb * 128) * 2) + wbF);
        T.bS = sgpr((unsigned)((((long)(q.n * p.S.H + y0 + p.S.oy) * p.S.W + x0 + p.S.ox) * p.S.C + csb * 64) * 2));
        T.ry = sgpr(p.Hf - y0);
        T.rx = sgpr(p.Wf - x0);
        T.insF = (int)(TR <= T.ry) & (int)(SW <= T.rx) & (int)(limcF >= 128);
        T.insS = (int)(HROWS <= T.ry + 2) & (int)(CW <= T.rx + 2) & (int)(limcS >= 64);
        return T;
    };
    auto mk = [&](const void* ptr) { return __builtin_amdgcn_make_buffer_rsrc((void*)ptr, 0, 0x7fffffff, 0x00020000); };
    // piece i (0 .. NFW+NSW-1) of tile T into slot `buf`: F pieces first
    auto issue_piece = [&](const Tile& T, auto ic, int buf) {
        constexpr int I = decltype(ic)::value;
        if constexpr (I < NFW) {
            const __amdgpu_buffer_rsrc_t rf = mk(p.F);
            const unsigned soff = T.bF + (I & 1) * rowF + (I >> 1) * 128;
            unsigned vo = lvF;
            if (!T.insF) {
                const int ty = wty + (I & 1) * (64 >> LSW);
                const int limx = ty < T.ry ? T.rx - wtx0 : 0, limc = limcF - (I >> 1) * 64;
                vo = (l8 < limx && cswz8 < limc) ? lvF : RSU_SENT;
            }
            bdma16w<WGPP_NT>(rf, vo, soff, (void*)(lds + buf * SLOT + (I * NW + wave) * 1024));
        } else if constexpr (I < NFW + NSW) {
            constexpr int Q = I - NFW;
            const __amdgpu_buffer_rsrc_t rs = mk(p.S.ptr);
            const unsigned soff = T.bS + wbS[Q];
            constexpr bool whole = NPIX >= (Q + 1) * 64;   // every wave's piece Q lies inside the halo tile
            unsigned vo = lvS;
            if (!whole || !T.insS) {
                const int hp0 = Q * 64 + wave * 8, rr = hp0 / CW, cc0 = hp0 - rr * CW;
                const int limx = (hp0 < NPIX && rr < T.ry + 2) ? T.rx + 2 - cc0 : 0;
                vo = (l8 < limx && cswz8 < limcS) ? lvS : RSU_SENT;
            }
            bdma16w<WGPP_NT>(rs, vo, soff, (void*)(lds + buf * SLOT + FBUF + (Q * NW + wave) * 1024));
        }
    };
    auto bar = [&]() {
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };
    auto tr_read = [&](int off) {
        return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bf16x4*)(lds + off));
    };

    // ---- prologue: the first tile of this split, all of it; G1 then sits out interval 0
    int tile = z;
    Pos pos = split(tile);   // of the tile staged last
    {
        const Tile T0 = decode(pos);
        issue_piece(T0, std::integral_constant<int, 0>{}, 0);
        issue_piece(T0, std::integral_constant<int, 1>{}, 0);
        issue_piece(T0, std::integral_constant<int, 2>{}, 0);
        issue_piece(T0, std::integral_constant<int, 3>{}, 0);
        issue_piece(T0, std::integral_constant<int, 4>{}, 0);
        issue_piece(T0, std::integral_constant<int, 5>{}, 0);
        issue_piece(T0, std::integral_constant<int, 6>{}, 0);
        issue_piece(T0, std::integral_constant<int, 7>{}, 0);
        issue_piece(T0, std::integral_constant<int, 8>{}, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    bar();
    if (grp) bar();

    auto run = [&](auto gconst) {
        constexpr int G = decltype(gconst)::value;
        int buf = 0;  // slot of the current tile (wave-uniform)
        // the tile after the current one (staged while the current one is multiplied); looked up in the LAST R interval of the tile
        // before, which has no staging to do, together with the move of the read addresses to the other slot
        int nxt = 0;
        bool have_nxt = false;
        Tile TN = Tile{0, 0, 0, 0, 0, 0};
        auto look_ahead = [&]() {
            nxt = tile + p.nsplit;
            have_nxt = nxt < p.ntiles_total && !(DBG & 1);
            if (have_nxt) {
                advance(pos);
                TN = decode(pos);
            }
        };
        look_ahead();
        auto do_tile = [&]() {
            // (two-phase schedules: the F fragments of a k-step are read in its first phase and stay for the second)
            bf16x8 fa[CFT];
            auto phase = [&](auto ksc, auto hc) {
                constexpr int K = decltype(ksc)::value, HB = decltype(hc)::value;
                constexpr int NPH = sched_nph(SCH), TSPL = SCH == 3 ? 4 : 5;
                constexpr int T0 = NPH == 1 ? 0 : (HB ? TSPL : 0), T1 = NPH == 1 ? NTAP : (HB ? NTAP : TSPL), NSV = NPH == 1 ? NTAP : 5;
                constexpr int PH = NPH * K + HB, LAST = NPH * 4 - 1;
                constexpr int P0 = sched_first(SCH, PH), PN = sched_count(SCH, PH);
                // ================= R interval
                bf16x8 sv[NSV];
                if constexpr (DBG & 2) {
                    if constexpr (HB == 0) {
#pragma unroll
                        for (int ct = 0; ct < CFT; ++ct) asm volatile("" : "=v"(fa[ct]));
                    }
#pragma unroll
                    for (int i = 0; i < NSV; ++i) asm volatile("" : "=v"(sv[i]));
                } else {
                    if constexpr (HB == 0) {
#pragma unroll
                        for (int ct = 0; ct < CFT; ++ct) {
                            const bf16x4 lo = tr_read(fo[0][ct] + K * 32 * 128), hi = tr_read(fo[1][ct] + K * 32 * 128);
                            fa[ct] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                        }
                    }
#pragma unroll
                    for (int tap = T0; tap < T1; ++tap) {
                        if ((DBG & 8) && tap >= 3) {  // (timing: a third of the S reads)
                            asm volatile("" : "=v"(sv[tap - T0]));
                            continue;
                        }
                        const int ky = tap / KW, kx = tap - ky * KW;
                        const int off = ky * CW * 128 + sdelta(K);
                        const bf16x4 lo = tr_read(so[0][kx] + off), hi = tr_read(so[1][kx] + off);
                        sv[tap - T0] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                    }
                }
                if constexpr (PN > 0) {
                    if (have_nxt) {
                        issue_piece(TN, std::integral_constant<int, P0>{}, buf ^ 1);
                        if constexpr (PN > 1) issue_piece(TN, std::integral_constant<int, P0 + 1>{}, buf ^ 1);
                        if constexpr (PN > 2) issue_piece(TN, std::integral_constant<int, P0 + 2>{}, buf ^ 1);
                    }
                }
                if constexpr (PH == LAST) {
                    // (the reads above carry the old addresses; everything below belongs to the next tile)
                    tile = nxt;
                    const int mv = buf ? -SLOT : SLOT;
                    buf ^= 1;
#pragma unroll
                    for (int rd = 0; rd < 2; ++rd) {
#pragma unroll
                        for (int ct = 0; ct < CFT; ++ct) {
                            fo[rd][ct] += mv;
                            asm volatile("" : "+v"(fo[rd][ct]));  // (here, not in the next tile's busier first R interval)
                        }
#pragma unroll
                        for (int kx = 0; kx < KW; ++kx) {
                            so[rd][kx] += mv;
                            asm volatile("" : "+v"(so[rd][kx]));
                        }
                    }
                    look_ahead();
                    if constexpr (G == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // G1's pieces of the next tile (nothing else is in flight)
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // fragments in registers: this slot may be refilled from the next interval on
                bar();
                // ================= M interval: MFMAs only
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_setprio(1);
                if constexpr (DBG & 4) {
#pragma unroll
                    for (int tap = T0; tap < T1; ++tap) asm volatile("" :: "v"(sv[tap - T0]), "v"(fa[tap & 3]));
                } else {
#pragma unroll
                    for (int tap = T0; tap < T1; ++tap)
#pragma unroll
                        for (int ct = 0; ct < CFT; ++ct) mfma_bf16_inplace(acc[tap][ct], fa[ct], sv[tap - T0]);
                }
                if (!(DBG & 4) && HB == 0 && do_bias) {
                    unsigned o1 = 0x3f803f80u;
                    asm volatile("" : "+v"(o1));
                    const u32x4 o4 = {o1, o1, o1, o1};
                    bf16x8 ones = __builtin_bit_cast(bf16x8, o4);
                    asm volatile("s_nop 3" : "+v"(ones));  // VALU-written operand -> (asm) MFMA read: the hazard recogniser cannot see it
                    mfma_bf16_inplace(accb, fa[0], ones);
                }
                __builtin_amdgcn_s_setprio(0);
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (PH == LAST && G == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // G0's pieces of the next tile
                bar();
            };
            using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
            using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;
            phase(I0{}, I0{});
            if constexpr (sched_nph(SCH) == 2) phase(I0{}, I1{});
            phase(I1{}, I0{});
            if constexpr (sched_nph(SCH) == 2) phase(I1{}, I1{});
            phase(I2{}, I0{});
            if constexpr (sched_nph(SCH) == 2) phase(I2{}, I1{});
            phase(I3{}, I0{});
            if constexpr (sched_nph(SCH) == 2) phase(I3{}, I1{});
        };
        while (tile < p.ntiles_total) do_tile();
    };
    if (grp) run(std::integral_constant<int, 1>{}); else run(std::integral_constant<int, 0>{});
    if (!grp) bar();  // G0 sits out G1's last M interval
    mfma_results_fence();

    // ---- outputs: bias sums (every column of accb holds the same sums: column 0 writes them) and this split's slab
    const int zs = z;
    if (do_bias && l15 == 0) {
        const int cf = cfb * 128 + (grp * CFT + wcs) * 16 + 4 * g4;
        if (cf < p.Cf) *(f32x4*)(p.bslab + (long)zs * p.slab_stride + cf) = accb;
    }
#pragma unroll
    for (int tap = 0; tap < NTAP; ++tap) {
        const int cs = csb * 64 + wcs * 16 + l15;
        if (cs >= p.S.C) continue;
#pragma unroll
        for (int ct = 0; ct < CFT; ++ct) {
            const int cf = cfb * 128 + (grp * CFT + ((ct + wcs) & 3)) * 16 + 4 * g4;
            if (cf >= p.Cf) continue;
            float* dst = p.slab + (long)zs * p.slab_stride + (((long)tap * p.CsOut + p.cs_off + cs) * p.CfOut + cf);
            *(f32x4*)dst = acc[tap][ct];
        }
    }
}

// ---------------------------------------------------------------------------------------------
// igemm_wgp64: the 64 x 64 shape (layers with 64 gradient channels: one workgroup holds the whole 9 x 64 x 64 block of dW for its
// pixel split) as a ping-pong. Both groups read the same 64 F channels; the NINE TAPS are dealt over them: G0 = waves 0-3 taps 0-4,
// G1 = waves 4-7 taps 5-8 and the bias sums; wave w of a group owns S channels 16*(w&3).. as in igemm_wgpp. A phase is TWO 32-pixel
// k-steps (36 / 32 transposed reads, 40 / 34 MFMAs per wave), a pixel tile two phases. Staging: ring of THREE slots ([F 16 KiB][S
// halo]), tile t in slot t % 3; the 2 + NSW pieces per wave of tile t+2 are issued in the R intervals of tile t, tile t+1 is waited
// for (counted vmcnt: tile t+2's pieces stay in flight) at the end of G1's last R interval / behind G0's last MFMAs of tile t. The
// stream never ends: behind the last tile it stages that tile again into a slot nobody reads, so the counts always hold.
// The summation order differs from igemm_wgrad's 64x64 shape (which split the k-steps over its wave groups): results agree to fp32
// rounding, not bit for bit.
template <int LSW, class P>
__device__ __forceinline__ void igemm_wgp64_body(const P& p, __attribute__((address_space(3))) char* lds, const int cfb, const int csb,
                                                 const int z, const unsigned tid) {
    constexpr int NW = 8, KW = 3, TMK = 128, CFT = 4;
    constexpr int SW = 1 << LSW, TR = TMK >> LSW;
    constexpr int CW = (SW + 2 + 7) / 8 * 8;
    constexpr int NPIX = ((TR + 2) * CW + 31) / 32 * 32;
    constexpr int NSW = ((NPIX + 7) / 8 + 7) / 8;
    constexpr int NFW = 2;                                 // F pieces per wave per tile: 128 px x 1 plane / 8 px / 8 waves
    constexpr int NP = NFW + NSW;
    constexpr int FBUF = TMK * 128, SBUF = NSW * NW * 1024, SLOT = FBUF + SBUF, NSLOT = 3;
    static_assert(NSLOT * SLOT <= 160 * 1024, "staging budget");

    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, wcs = wave & 3;
    const int g4 = lane >> 4, l15 = lane & 15, q4 = l15 >> 2, p4 = lane & 3;
    const int tpi = p.g.nstrips * p.g.tiles_per_strip;
    auto sgpr = [](auto v) { return __builtin_amdgcn_readfirstlane(v); };

    // ---- per-lane LDS read offsets inside the CURRENT slot (they move by SLOT / -2 SLOT per tile)
    int fo[2][CFT], so[2][KW];
#pragma unroll
    for (int rd = 0; rd < 2; ++rd) {
        const int ml = rd * 16 + 4 * g4 + q4;
        const int ch = 4 * p4;
        const int f0 = ml * 128 + ((((ch & 63) >> 4) ^ ((ml >> 1) & 3)) << 5) + (ch & 15) * 2;
#pragma unroll
        for (int ct = 0; ct < CFT; ++ct) fo[rd][ct] = f0 ^ (((ct + wcs) & 3) << 5);   // F tiles numbered from the wave's own (bias tile = column 0)
        const int ty = ml >> LSW, tx = ml & (SW - 1);
        const int hp0 = ty * CW + tx;
#pragma unroll
        for (int kx = 0; kx < KW; ++kx) {
            const int hp = hp0 + kx;
            const int chs = wcs * 16 + 4 * p4;
            so[rd][kx] = FBUF + hp * 128 + ((((chs >> 4) ^ ((hp >> 1) & 3))) << 5) + (chs & 15) * 2;
        }
    }
    auto sdelta = [](int k) constexpr {
        const int m0 = k * 32;
        return ((m0 >> LSW) * CW + (m0 & (SW - 1))) * 128;
    };
    // ---- staging (scalar addressing as in igemm_wgpp)
    const int l8 = lane >> 3;
    const int cswz8 = ((lane & 7) ^ (((l8 >> 1) & 3) << 1)) * 8;
    unsigned lvF = (unsigned)((l8 * p.Cf + cswz8) * 2), lvS = (unsigned)((l8 * p.S.C + cswz8) * 2);
    asm volatile("" : "+v"(lvF), "+v"(lvS));
    const int limcS = p.S.C - csb * 64, limcF = p.Cf - cfb * 64;
    const int wty = (wave * 8) >> LSW, wtx0 = (wave * 8) & (SW - 1);
    const unsigned rowF = sgpr((unsigned)(p.Wf * p.Cf * 2) * (64 >> LSW));
    const unsigned wbF = sgpr((unsigned)((wty * p.Wf + wtx0) * p.Cf * 2));
    unsigned wbS[NSW];
#pragma unroll
    for (int q = 0; q < NSW; ++q) {
        const int hp0 = q * 64 + wave * 8, rr = hp0 / CW, cc0 = hp0 - rr * CW;
        wbS[q] = sgpr((unsigned)((rr * p.S.W + cc0) * p.S.C * 2));
    }
    constexpr int HROWS = (NPIX - 1) / CW + 1;
    struct Tile { unsigned bF, bS; int insF, insS, ry, rx; };
    struct Pos { int n, strip, row; };
    auto split = [&](int t) {
        const int n = t / tpi, r = t - n * tpi;
        const int strip = r / p.g.tiles_per_strip;
        return Pos{sgpr(n), sgpr(strip), sgpr(r - strip * p.g.tiles_per_strip)};
    };
    const Pos step = split(p.nsplit);
    auto advance = [&](Pos& q) {
        q.row += step.row;
        if (q.row >= p.g.tiles_per_strip) { q.row -= p.g.tiles_per_strip; ++q.strip; }
        q.strip += step.strip;
        if (q.strip >= p.g.nstrips) { q.strip -= p.g.nstrips; ++q.n; }
        q.n += step.n;
    };
    auto decode = [&](const Pos& q) {
        Tile T;
        const int x0 = q.strip * SW, y0 = q.row * TR;
        T.bF = sgpr((unsigned)((((long)(q.n * p.Hf + y0) * p.Wf + x0) * p.Cf + cfb * 64) * 2) + wbF);
        T.bS = sgpr((unsigned)((((long)(q.n * p.S.H + y0 + p.S.oy) * p.S.W + x0 + p.S.ox) * p.S.C + csb * 64) * 2));
        T.ry = sgpr(p.Hf - y0);
        T.rx = sgpr(p.Wf - x0);
        T.insF = (int)(TR <= T.ry) & (int)(SW <= T.rx) & (int)(limcF >= 64);
        T.insS = (int)(HROWS <= T.ry + 2) & (int)(CW <= T.rx + 2) & (int)(limcS >= 64);
        return T;
    };
    auto mk = [&](const void* ptr) { return __builtin_amdgcn_make_buffer_rsrc((void*)ptr, 0, 0x7fffffff, 0x00020000); };
    auto issue_piece = [&](const Tile& T, auto ic, int buf) {
        constexpr int I = decltype(ic)::value;
        if constexpr (I < NFW) {
            const __amdgpu_buffer_rsrc_t rf = mk(p.F);
            const unsigned soff = T.bF + I * rowF;
            unsigned vo = lvF;
            if (!T.insF) {
                const int ty = wty + I * (64 >> LSW);
                const int limx = ty < T.ry ? T.rx - wtx0 : 0;
                vo = (l8 < limx && cswz8 < limcF) ? lvF : RSU_SENT;
            }
            bdma16w<WG64_NT>(rf, vo, soff, (void*)(lds + buf * SLOT + (I * NW + wave) * 1024));
        } else if constexpr (I < NP) {
            constexpr int Q = I - NFW;
            const __amdgpu_buffer_rsrc_t rs = mk(p.S.ptr);
            const unsigned soff = T.bS + wbS[Q];
            constexpr bool whole = NPIX >= (Q + 1) * 64;
            unsigned vo = lvS;
            if (!whole || !T.insS) {
                const int hp0 = Q * 64 + wave * 8, rr = hp0 / CW, cc0 = hp0 - rr * CW;
                const int limx = (hp0 < NPIX && rr < T.ry + 2) ? T.rx + 2 - cc0 : 0;
                vo = (l8 < limx && cswz8 < limcS) ? lvS : RSU_SENT;
            }
            bdma16w<WG64_NT>(rs, vo, soff, (void*)(lds + buf * SLOT + FBUF + (Q * NW + wave) * 1024));
        }
    };
    auto bar = [&]() {
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };
    auto tr_read = [&](int off) {
        return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bf16x4*)(lds + off));
    };
    using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>;
    using I3 = std::integral_constant<int, 3>; using I4 = std::integral_constant<int, 4>; using I5 = std::integral_constant<int, 5>;
    using I6 = std::integral_constant<int, 6>;
    auto issue_half = [&](const Tile& T, auto hc, int buf) {   // the first / second half of a tile's pieces of this wave
        constexpr int H = decltype(hc)::value;
        if constexpr (H == 0) {
            issue_piece(T, I0{}, buf);
            issue_piece(T, I1{}, buf);
            issue_piece(T, I2{}, buf);
        } else {
            issue_piece(T, I3{}, buf);
            issue_piece(T, I4{}, buf);
            issue_piece(T, I5{}, buf);
            issue_piece(T, I6{}, buf);
        }
    };

    // ---- prefetch stream: tile index (in this split) and position of the tile staged next
    const int ntile_mine = (p.ntiles_total - z + p.nsplit - 1) / p.nsplit;   // tiles z, z + nsplit, ...
    Pos ppos = split(z);
    int pf_k = 0;        // index (in this workgroup's list) of the tile the stream stages next; stays at the last tile
    int pf_slot = 0;
    Tile PT = decode(ppos);
    auto pf_next = [&]() {
        if (pf_k + 1 < ntile_mine) {
            ++pf_k;
            advance(ppos);
            PT = decode(ppos);
        }
        pf_slot = pf_slot == NSLOT - 1 ? 0 : pf_slot + 1;
    };
    // prologue: tiles 0 and 1 (or tile 0 twice), all of them
    issue_half(PT, I0{}, 0);
    issue_half(PT, I1{}, 0);
    pf_next();
    issue_half(PT, I0{}, 1);
    issue_half(PT, I1{}, 1);
    pf_next();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    bar();
    if (grp) bar();

    const bool do_bias = (p.bslab != nullptr) && (csb == 0);
    auto run = [&](auto gconst) {
        constexpr int G = decltype(gconst)::value;
        constexpr int T0 = G ? 5 : 0, NT = G ? 4 : 5;
        f32x4 acc[NT][CFT];
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int a = 0; a < CFT; ++a) acc[t][a] = f32x4{0.f, 0.f, 0.f, 0.f};
        f32x4 accb = f32x4{0.f, 0.f, 0.f, 0.f};
        int cur = 0;   // slot of the current tile
        for (int k = 0; k < ntile_mine; ++k) {
            auto phase = [&](auto phc) {
                constexpr int PH = decltype(phc)::value;
                // ================= R interval: the fragments of k-steps 2 PH and 2 PH + 1, half of the pieces of tile k + 2
                bf16x8 fa[2][CFT], sv[2][NT];
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    constexpr int dummy = 0;
                    (void)dummy;
                    const int K = 2 * PH + ks;
#pragma unroll
                    for (int ct = 0; ct < CFT; ++ct) {
                        const bf16x4 lo = tr_read(fo[0][ct] + K * 32 * 128), hi = tr_read(fo[1][ct] + K * 32 * 128);
                        fa[ks][ct] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                    }
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
                        const int tap = T0 + t, ky = tap / KW, kx = tap - ky * KW;
                        const int off = ky * CW * 128 + sdelta(K);
                        const bf16x4 lo = tr_read(so[0][kx] + off), hi = tr_read(so[1][kx] + off);
                        sv[ks][t] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                    }
                }
                issue_half(PT, phc, pf_slot);
                if constexpr (PH == 1) {
                    pf_next();
                    // the reads above carry the old addresses: move them to the next tile's slot
                    const int mv = cur == NSLOT - 1 ? -(NSLOT - 1) * SLOT : SLOT;
                    cur = cur == NSLOT - 1 ? 0 : cur + 1;
#pragma unroll
                    for (int rd = 0; rd < 2; ++rd) {
#pragma unroll
                        for (int ct = 0; ct < CFT; ++ct) {
                            fo[rd][ct] += mv;
                            asm volatile("" : "+v"(fo[rd][ct]));
                        }
#pragma unroll
                        for (int kx = 0; kx < KW; ++kx) {
                            so[rd][kx] += mv;
                            asm volatile("" : "+v"(so[rd][kx]));
                        }
                    }
                    if constexpr (G == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NP) : "memory");   // tile k + 1 has landed (k + 2 may be in flight)
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                bar();
                // ================= M interval
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
                    for (int t = 0; t < NT; ++t)
#pragma unroll
                        for (int ct = 0; ct < CFT; ++ct) mfma_bf16_inplace(acc[t][ct], fa[ks][ct], sv[ks][t]);
                    if (G == 1 && do_bias) {
                        unsigned o1 = 0x3f803f80u;
                        asm volatile("" : "+v"(o1));
                        const u32x4 o4 = {o1, o1, o1, o1};
                        bf16x8 ones = __builtin_bit_cast(bf16x8, o4);
                        asm volatile("s_nop 3" : "+v"(ones));
                        mfma_bf16_inplace(accb, fa[ks][0], ones);
                    }
                }
                __builtin_amdgcn_s_setprio(0);
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (PH == 1 && G == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NP) : "memory");
                bar();
            };
            phase(I0{});
            phase(I1{});
        }
        mfma_results_fence();
        // ---- outputs of this group: its taps of this split's slab (and, G1, the bias sums: every column of accb holds them)
        if (G == 1 && do_bias && l15 == 0) {
            const int cf = cfb * 64 + wcs * 16 + 4 * g4;
            if (cf < p.Cf) *(f32x4*)(p.bslab + (long)z * p.slab_stride + cf) = accb;
        }
        const int cs = csb * 64 + wcs * 16 + l15;
        if (cs < p.S.C) {
#pragma unroll
            for (int t = 0; t < NT; ++t) {
#pragma unroll
                for (int ct = 0; ct < CFT; ++ct) {
                    const int cf = cfb * 64 + ((ct + wcs) & 3) * 16 + 4 * g4;
                    if (cf >= p.Cf) continue;
                    float* dst = p.slab + (long)z * p.slab_stride + (((long)(T0 + t) * p.CsOut + p.cs_off + cs) * p.CfOut + cf);
                    *(f32x4*)dst = acc[t][ct];
                }
            }
        }
    };
    if (grp) {
        run(I1{});
    } else {
        run(I0{});
        bar();  // G0 sits out G1's last M interval
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the stream's last pieces: nothing may land in this workgroup's LDS after it has gone
}

// ---------------------------------------------------------------------------------------------
// one launch per layer: (cfb, csb, z) from an XCD-contiguous numbering -- the gx * gy workgroups of one pixel split z read the same F
// and S pixel tiles (each F tile gy times, each S tile gx times): on one XCD all but the first of those reads are L2 hits
#define WG_UNIT_FROM_GRID()                                                                                                                   \
    extern __shared__ __attribute__((aligned(16))) char smem[];                                                                              \
    __attribute__((address_space(3))) char* lds = (__attribute__((address_space(3))) char*)smem;                                             \
    const int lid = xcd_contiguous_id(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z), gridDim.x * gridDim.y * gridDim.z);  \
    const int cfb = lid % gridDim.x, csb = (lid / gridDim.x) % gridDim.y, z = lid / (gridDim.x * gridDim.y)
template <int LSW, int DBG, int SCH>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) igemm_wgpp_kernel(const IgWgradParams p) {
    WG_UNIT_FROM_GRID();
    igemm_wgpp_body<LSW, DBG, SCH>(p, lds, cfb, csb, z, threadIdx.x);
}
template <int LSW>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) igemm_wgp64_kernel(const IgWgradParams p) {
    WG_UNIT_FROM_GRID();
    igemm_wgp64_body<LSW>(p, lds, cfb, csb, z, threadIdx.x);
}

// ---------------------------------------------------------------------------------------------
// igemm_wg_group: the weight gradients of SEVERAL layers in one launch. A layer launched alone on 256 CUs writes one fp32 partial
// result (a 128 x 64 x 9 slab, 295 KB) per workgroup whatever its size -- 75 MB per layer, read back by the reduce kernel -- and a
// layer with few pixel tiles leaves part of the chip idle. Here every layer (job) gets a share of the workgroups in proportion to its
// work (rsu_api.hip plans it): few pixel splits for the shallow layers (the slabs of the WHOLE group are about one per CU), none for the
// deep ones, whose (channel block) units reduce over all pixels and write the gradient in place.
// A unit u < gx * gy * gz of job j is (cfb, csb, z) = (u % gx, (u / gx) % gy, u / (gx * gy)). The launch is PERSISTENT -- exactly one
// workgroup per budgeted CU, so that it keeps to its share of the chip beside the other stream's kernel -- and workgroup b walks the
// units unit[wg_first[b] .. wg_first[b+1]) the planner dealt to it (longest-first greedy: every workgroup gets about the same work).
// The unit bodies are the very kernels above (families: igemm_wgpp per strip width, igemm_wgp64, and the generic igemm_wgrad shapes for the 2x2
// stride-2 taps of the transposed convs, dilated convs and the 16-channel input of level 0).
template <int FAM, class P>
__device__ __forceinline__ void wg_group_unit(const P& p, __attribute__((address_space(3))) char* lds, int cfb, int csb, int z, unsigned tid) {
    if constexpr (FAM == IGW_FAM_WGPP3) igemm_wgpp_body<3, 0, 4>(p, lds, cfb, csb, z, tid);
    else if constexpr (FAM == IGW_FAM_WGPP4) igemm_wgpp_body<4, 0, 4>(p, lds, cfb, csb, z, tid);
    else if constexpr (FAM == IGW_FAM_WGPP5) igemm_wgpp_body<5, 0, 4>(p, lds, cfb, csb, z, tid);
    else if constexpr (FAM == IGW_FAM_WGPP6) igemm_wgpp_body<6, 0, 4>(p, lds, cfb, csb, z, tid);
    else if constexpr (FAM == IGW_FAM_WGP64_4) igemm_wgp64_body<4>(p, lds, cfb, csb, z, tid);
    else if constexpr (FAM == IGW_FAM_WGP64_5) igemm_wgp64_body<5>(p, lds, cfb, csb, z, tid);
    else {
        constexpr int CFG = (FAM - IGW_FAM_GENERIC) >> 1, NTAP = ((FAM - IGW_FAM_GENERIC) & 1) ? 4 : 9, KW = NTAP == 4 ? 2 : 3;
        using C = WgCfg<CFG>;
        igemm_wgrad_body<C::WCF, C::WCS, C::CFT, C::CST, NTAP, KW, C::TMK, C::KG>(p, lds, cfb, csb, z, tid);
    }
}
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) igemm_wg_group_kernel(const IgWgGroupParams* table) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __attribute__((address_space(3))) char* lds = (__attribute__((address_space(3))) char*)smem;
    // persistent: one workgroup per budgeted CU walks the unit list the planner made for it (longest units first).
    // The loop state -- position in the list, its end, the table pointer -- lives in LDS (one 16-byte slot per wave in the last 256
    // bytes of the 160 KiB, which no unit body uses; IGW_GROUP_LDS_BYTES) and is read back through an opaque asm at the top of every
    // iteration: the unit bodies run at the limit of both register files, and four scalars carried across them in registers cost
    // 60 spilled vector registers in the tightest one.
    {
        const unsigned state = IGW_GROUP_LDS_BYTES - 256 + (threadIdx.x >> 6) * 16;
        const u32x4 st0 = {(unsigned)__ldg(&table->wg_first[blockIdx.x]), (unsigned)__ldg(&table->wg_first[blockIdx.x + 1]),
                           (unsigned)(uintptr_t)table, (unsigned)((uintptr_t)table >> 32)};
        asm volatile("ds_write_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" ::"v"(state), "v"(st0) : "memory");
    }
    for (;;) {
        // (the thread id through an opaque asm, once per iteration: otherwise the compiler hoists everything the unit bodies derive from
        // it -- lane, wave, LDS offsets -- out of the loop and keeps it in registers across bodies that have none to spare)
        unsigned tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
        const unsigned state = IGW_GROUP_LDS_BYTES - 256 + (tid >> 6) * 16;
        u32x4 st;
        asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(st) : "v"(state) : "memory");
        const int i = __builtin_amdgcn_readfirstlane((int)st[0]), iend = __builtin_amdgcn_readfirstlane((int)st[1]);
        if (i >= iend) break;
        {
            const unsigned nxt = (unsigned)i + 1u;
            asm volatile("ds_write_b32 %0, %1" ::"v"(state), "v"(nxt) : "memory");   // (every lane of the wave: same word, same value)
        }
        // the job table is read through the constant address space (scalar loads, like kernel arguments): it is written once, when the
        // group is planned, and never while a launch that reads it is in flight
        typedef const __attribute__((address_space(4))) IgWgGroupParams* CTab;
        CTab g = (CTab)(((uintptr_t)(unsigned)__builtin_amdgcn_readfirstlane((int)st[3]) << 32) | (uintptr_t)(unsigned)__builtin_amdgcn_readfirstlane((int)st[2]));
        const unsigned e = g->unit[i];
        const int j = (int)(e >> 24), u = (int)(e & 0xffffffu);
        const int gx = g->job[j].gx, gy = g->job[j].gy, family = g->job[j].family;
        const int cfb = u % gx, csb = (u / gx) % gy, z = u / (gx * gy);
        const auto& p = g->job[j].p;   // (stays in the constant address space: the body loads a field where it uses it, like a kernel argument)
        switch (family) {
#define WGG_CASE(F) case F: wg_group_unit<F>(p, lds, cfb, csb, z, tid); break;
            WGG_CASE(IGW_FAM_WGPP3) WGG_CASE(IGW_FAM_WGPP4) WGG_CASE(IGW_FAM_WGPP5) WGG_CASE(IGW_FAM_WGPP6)
            WGG_CASE(IGW_FAM_WGP64_4) WGG_CASE(IGW_FAM_WGP64_5)
            WGG_CASE(IGW_FAM_GENERIC + 2 * IGW_CFG_64x64 + 1)   // (the 9-tap 64x64 generic shape spills registers: it stays a launch of its own)
            WGG_CASE(IGW_FAM_GENERIC + 2 * IGW_CFG_64x16)
            WGG_CASE(IGW_FAM_GENERIC + 2 * IGW_CFG_128x64) WGG_CASE(IGW_FAM_GENERIC + 2 * IGW_CFG_128x64 + 1)
#undef WGG_CASE
        }
        // a unit leaves nothing in flight (every body ends behind s_waitcnt vmcnt(0)); the barrier keeps a fast wave's staging of the
        // next unit out of LDS the slowest wave still reads
        __syncthreads();
    }
}
// family of a planned launch (the kernel the single-layer path would pick), or -1
int igemm_wg_group_family(int cfg, int ntap, const IgWgradParams& p) {
    if (igemm_wgpp_supports(cfg, ntap, p)) return IGW_FAM_WGPP3 + (p.lsw - 3);
    if (igemm_wgp64_supports(cfg, ntap, p)) return p.lsw == 4 ? IGW_FAM_WGP64_4 : IGW_FAM_WGP64_5;
    if (ntap == 9 && cfg != IGW_CFG_64x64) return IGW_FAM_GENERIC + 2 * cfg;
    if (ntap == 4 && cfg != IGW_CFG_64x16) return IGW_FAM_GENERIC + 2 * cfg + 1;
    return -1;
}
// dynamic LDS a job of this family needs (the launch asks for the largest of its jobs)
size_t igemm_wg_group_lds_bytes(int family, const IgWgradParams& p) {
    const size_t nsw = (size_t)(((p.g.npix_max + 7) / 8 + 7) / 8);
    if (family >= IGW_FAM_WGPP3 && family <= IGW_FAM_WGPP6) return 2 * (2 * 128 * 128 + nsw * 8 * 1024);
    if (family == IGW_FAM_WGP64_4 || family == IGW_FAM_WGP64_5) return 3 * (128 * 128 + nsw * 8 * 1024);
    return igemm_wgrad_lds_bytes((family - IGW_FAM_GENERIC) >> 1, p.g.npix_max, p.nbuf);
}
hipError_t igemm_wg_group_launch(const IgWgGroupParams* dev_table, int nwg_total, hipStream_t st) {
    static bool set = false;
    if (!set) {
        hipError_t e = hipFuncSetAttribute((const void*)igemm_wg_group_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, IGW_GROUP_LDS_BYTES);
        if (e != hipSuccess) return e;
        set = true;
    }
    hipLaunchKernelGGL(igemm_wg_group_kernel, dim3(nwg_total), dim3(512), IGW_GROUP_LDS_BYTES, st, dev_table);
    return hipGetLastError();
}

template <int LSW>
static hipError_t wgp64_launch_one(const IgWgradParams& p, int gx, int gy, int gz, hipStream_t st) {
    constexpr int SW = 1 << LSW, TR = 128 >> LSW;
    constexpr int CW = (SW + 2 + 7) / 8 * 8, NPIX = ((TR + 2) * CW + 31) / 32 * 32, NSW = ((NPIX + 7) / 8 + 7) / 8;
    if (p.g.SW != SW || p.g.CW != CW || p.g.npix_max != NPIX) return hipErrorInvalidValue;
    auto kern = igemm_wgp64_kernel<LSW>;
    const size_t lds = 3 * (size_t)(128 * 128 + NSW * 8 * 1024);
    static bool set = false;
    if (!set) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        set = true;
    }
    hipLaunchKernelGGL(kern, dim3(gx, gy, gz), dim3(512), lds, st, p);
    return hipGetLastError();
}
// the launches igemm_wgp64 is built for: the 64x64 shape of the 3x3 stride-1 dilation-1 weight gradient, strip widths 16 and 32
bool igemm_wgp64_supports(int cfg, int ntap, const IgWgradParams& p) {
    if (cfg != IGW_CFG_64x64 || ntap != 9 || p.stride != 1 || p.dil != 1 || (p.lsw != 4 && p.lsw != 5) || p.sbslab) return false;
    const int SW = 1 << p.lsw, TR = 128 >> p.lsw;
    const int CW = (SW + 2 + 7) / 8 * 8, NPIX = ((TR + 2) * CW + 31) / 32 * 32, NSW = ((NPIX + 7) / 8 + 7) / 8;
    return p.g.SW == SW && p.g.CW == CW && p.g.npix_max == NPIX && NSW + 2 <= 7 && 3 * (128 * 128 + NSW * 8 * 1024) <= 160 * 1024;
}
hipError_t igemm_wgp64_launch(const IgWgradParams& p, int gx, int gy, int gz, hipStream_t st) {
    return p.lsw == 4 ? wgp64_launch_one<4>(p, gx, gy, gz, st) : wgp64_launch_one<5>(p, gx, gy, gz, st);
}

template <int LSW, int DBG = 0, int SCH = 4>
static hipError_t wgpp_launch_one(const IgWgradParams& p, int gx, int gy, int gz, hipStream_t st) {
    constexpr int SW = 1 << LSW, TR = 128 >> LSW;
    constexpr int CW = (SW + 2 + 7) / 8 * 8, NPIX = ((TR + 2) * CW + 31) / 32 * 32, NSW = ((NPIX + 7) / 8 + 7) / 8;
    if (p.g.SW != SW || p.g.CW != CW || p.g.npix_max != NPIX || p.nsw != NSW) return hipErrorInvalidValue;
    auto kern = igemm_wgpp_kernel<LSW, DBG, SCH>;
    const size_t lds = 2 * (size_t)(2 * 128 * 128 + NSW * 8 * 1024);
    static size_t lds_set = 0;
    if (lds > lds_set) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        lds_set = lds;
    }
    hipLaunchKernelGGL(kern, dim3(gx, gy, gz), dim3(512), lds, st, p);
    return hipGetLastError();
}
// the launches igemm_wgpp is built for: the 128x64 shape of the 3x3 stride-1 dilation-1 weight gradient with the planner's halo tile
bool igemm_wgpp_supports(int cfg, int ntap, const IgWgradParams& p) {
    if (cfg != IGW_CFG_128x64 || ntap != 9 || p.stride != 1 || p.dil != 1 || p.lsw < 3 || p.lsw > 6) return false;
    const int SW = 1 << p.lsw, TR = 128 >> p.lsw;
    const int CW = (SW + 2 + 7) / 8 * 8, NPIX = ((TR + 2) * CW + 31) / 32 * 32, NSW = ((NPIX + 7) / 8 + 7) / 8;
    return p.g.SW == SW && p.g.CW == CW && p.g.npix_max == NPIX && p.nsw == NSW && NSW + 4 <= 9 &&
           2 * (2 * 128 * 128 + NSW * 8 * 1024) <= 160 * 1024;
}
hipError_t igemm_wgpp_launch(const IgWgradParams& p, int gx, int gy, int gz, hipStream_t st) {
#ifdef RSU_DEV_KERNELS   // developer build only (make DEV=1 -> build_ab/)
    if (p.dbg) {  // timing ablations / schedule variants (strip width 16 and 32 only): RSU_WG_DBG = 16 * SCH + DBG
#define WGPP_CASE(L, S, D) if (p.lsw == L && p.dbg == 16 * S + D) return wgpp_launch_one<L, D, S>(p, gx, gy, gz, st)
#define WGPP_CASES(L, S) WGPP_CASE(L, S, 0); WGPP_CASE(L, S, 1); WGPP_CASE(L, S, 2); WGPP_CASE(L, S, 3); WGPP_CASE(L, S, 4); WGPP_CASE(L, S, 5); WGPP_CASE(L, S, 7); WGPP_CASE(L, S, 9)
        WGPP_CASES(4, 0); WGPP_CASES(5, 0); WGPP_CASES(4, 4); WGPP_CASES(5, 4); WGPP_CASE(4, 3, 0); WGPP_CASE(5, 3, 0);
#undef WGPP_CASES
#undef WGPP_CASE
    }
#endif
    switch (p.lsw) {
        case 3: return wgpp_launch_one<3>(p, gx, gy, gz, st);
        case 4: return wgpp_launch_one<4>(p, gx, gy, gz, st);
        case 5: return wgpp_launch_one<5>(p, gx, gy, gz, st);
        case 6: return wgpp_launch_one<6>(p, gx, gy, gz, st);
    }
    return hipErrorInvalidValue;
}
