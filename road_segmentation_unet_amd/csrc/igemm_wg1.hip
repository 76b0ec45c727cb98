// igemm_wg1: weight gradient (+ bias gradient) of the network's FIRST 3x3 convolution over the 16-channel input tensor
// (rsu_conv_first_bwd_weight: the rows of tmp[9][16][Cout] are dW1 and the sums the colour-adjust gradient is made of).
//   tmp[tap][ci][co] = sum_m in16[n][y + ky*dil][x + kx*dil][ci] * dz[m][co],   db[co] = sum_m dz[m][co],   m = (n, y, x) output pixels
// 24 GFLOP over 208 MB (dz 166 MB, in16 42 MB at B = 4): a bandwidth-bound reduction. Rounds 1-3a ran it through the generic igemm_wgrad
// kernel in its 64x16 shape: 72 us (2.9 TB/s), 3.1 vector + 7.7 scalar instructions and 2 LDS reads per MFMA, 15 % of the MFMA slots.
// Built like igemm_wgt (the transposed conv's weight gradient):
//   * the nine taps are nine PHASE IMAGES of in16, gathered by the LDS-DMA while staging (a lane fetches pixel (y + ky*dil, x + kx*dil)
//     of its output pixel; the 42-MB input comes from L2 eight times out of nine): a tap is an immediate LDS offset, a transposed
//     read a contiguous 512-byte block of a 32-byte-per-pixel image (no bank conflicts);
//   * a workgroup owns 64 co x 16 ci x 9 taps and EVERY wave holds all of it (36 accumulator tiles + 4 for the bias sums): a wave
//     reduces one 32-pixel k-step of a 128-pixel tile -- 36 + 4 MFMAs behind 26 transposed reads -- the four waves of a group the four
//     k-steps of one tile; wave group G0 takes the even tiles of the workgroup's pixel split, G1 the odd ones, one group reading while
//     the other multiplies, one barrier per interval; the eight partial sums meet once, through LDS, behind the last tile (fixed tree);
//   * tiles are flat runs of the output pixel index; three ring slots of 52 KiB (dz plane 16 KiB + 9 x 4 KiB); in every interval
//     every wave issues its seven pieces of the tile two ahead and waits with one counted s_waitcnt vmcnt(7).
// Slabs ([9][16][Cout] + a bias row per pixel split) and their reduction are igemm_wgrad's.
#include <type_traits>

#include "igemm_wgrad_body.h"

struct IgWg1Params {
    const bf16_t* in16;   // [N][H][W][16]
    const bf16_t* dz;     // [N][Ho][Wo][Cout]
    float* slab;          // [nsplit] x ([9][16][Cout] + bias row)
    float* bslab;         // per-split column sums of dz, or null
    long slab_stride;
    int N, H, W, Ho, Wo, Cout, dil;
    int nsplit, ntiles;   // pixel splits (grid.y) and 128-pixel tiles in all
    unsigned wo_magic, ho_magic;   // floor(2^32 / Wo), floor(2^32 / Ho)
};

namespace {
constexpr int WG1_FPL = 128 * 128;              // dz plane: 128 pixels x 64 channels
constexpr int WG1_SPH = 128 * 32;               // one phase image of in16: 128 pixels x 16 channels
constexpr int WG1_SLOT = WG1_FPL + 9 * WG1_SPH; // 52 KiB
constexpr int WG1_NSLOT = 3;
constexpr int WG1_SCRATCH = WG1_NSLOT * WG1_SLOT;   // 1 KiB behind the ring for the padding piece
}  // namespace

__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) igemm_wg1_kernel(const IgWg1Params p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __attribute__((address_space(3))) char* lds = (__attribute__((address_space(3))) char*)smem;
    const int lid = xcd_contiguous_id(blockIdx.x + gridDim.x * blockIdx.y, gridDim.x * gridDim.y);
    const int cfb = lid % gridDim.x, z = lid / gridDim.x;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int grp = wave >> 2, w4 = wave & 3;
    const int g4 = lane >> 4, l15 = lane & 15, q4 = l15 >> 2, p4 = lane & 3;
    const int P = p.N * p.Ho * p.Wo;
    const int n_mine = z < p.ntiles ? (p.ntiles - z + p.nsplit - 1) / p.nsplit : 0;   // tiles z, z + nsplit, ... of this split
    if (n_mine == 0) return;

    f32x4 acc[9][4];   // [tap][co tile]; cols = the 16 input channels
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int a = 0; a < 4; ++a) acc[t][a] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 accb[4];     // column sums of dz (F^T x ones)
#pragma unroll
    for (int a = 0; a < 4; ++a) accb[a] = f32x4{0.f, 0.f, 0.f, 0.f};
    const bool do_bias = p.bslab != nullptr;

    // ---- transposed reads of this wave's k-step (pixels w4*32 .. w4*32+31 of the tile)
    int fx[2][4], sx[2];
#pragma unroll
    for (int rd = 0; rd < 2; ++rd) {
        const int ml = w4 * 32 + rd * 16 + 4 * g4 + q4;
        const int rb = ml * 128 + (((ml >> 1) & 3) << 5) + p4 * 8;
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) fx[rd][ct] = rb ^ (ct << 5);
        sx[rd] = WG1_FPL + ml * 32 + p4 * 8;
    }
    // ---- staging: 52 pieces per tile, seven per wave (the last four of the 56 are padding into the scratch slot):
    //   piece id = j * 8 + wave; 0..15 = the dz plane (8 pixels x 128 bytes each); 16..51 = phase (id - 16) >> 2, quarter (id - 16) & 3
    //   (32 pixels x 32 bytes): the S pieces of a wave all hold quarter wave & 3 -- one pixel decode per lane and interval
    const int fpos = lane & 7;
    const __amdgpu_buffer_rsrc_t rF = __builtin_amdgcn_make_buffer_rsrc((void*)p.dz, 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rS = __builtin_amdgcn_make_buffer_rsrc((void*)p.in16, 0, 0x7fffffff, 0x00020000);
    auto issue = [&](int i, int slot) {   // this wave's seven pieces of the split's tile i
        int ln = lane;
        asm volatile("" : "+v"(ln));      // (keeps the pixel arithmetic inside the interval)
        const int m0 = (z + i * p.nsplit) * 128;
        __attribute__((address_space(3))) char* sl = lds + slot * WG1_SLOT;
        // dz pieces `wave` and `wave + 8`
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int px = (j * 8 + wave) * 8 + (ln >> 3);
            const int m = m0 + px;
            const int c = fpos ^ (((px >> 1) & 3) << 1);
            const bool ok = m < P && cfb * 64 + c * 8 < p.Cout;
            bdma16(rF, ok ? (unsigned)((m * p.Cout + cfb * 64 + c * 8) * 2) : RSU_SENT, 0u, (void*)(sl + (j * 8 + wave) * 1024));
        }
        // in16 pieces: quarter wave & 3 of phases 2 * (j - 2) + (wave >> 2), j = 2 .. 6 (phase 9 of the last one does not exist: padding)
        const int m = m0 + (wave & 3) * 32 + (ln >> 1);
        const bool ok = m < P;
        const int mc = ok ? m : 0;
        unsigned q = __umulhi((unsigned)mc, p.wo_magic);   // q = n * Ho + y (floor(2^32 / d) may fall one short)
        int xx = mc - (int)q * p.Wo;
        if (xx >= p.Wo) { ++q; xx -= p.Wo; }
        unsigned nn = __umulhi(q, p.ho_magic);
        int yy = (int)q - (int)nn * p.Ho;
        if (yy >= p.Ho) { ++nn; yy -= p.Ho; }
        const unsigned so = ok ? (unsigned)(((((int)nn * p.H + yy) * p.W + xx) * 16 + (ln & 1) * 8) * 2) : RSU_SENT;
#pragma unroll
        for (int j = 2; j < 7; ++j) {
            const int ph = 2 * (j - 2) + (wave >> 2);   // (wave-uniform)
            if (ph < 9) {
                const int ky = ph / 3, kx = ph - 3 * ky;
                bdma16(rS, so, (unsigned)(((ky * p.W + kx) * p.dil) * 32), (void*)(sl + WG1_FPL + ph * WG1_SPH + (wave & 3) * 1024));
            } else {
                bdma16(rS, RSU_SENT, 0u, (void*)(lds + WG1_SCRATCH));
            }
        }
    };
    auto bar = [&]() {
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };

    // ---- prologue: tiles 0 and 1 of the split complete before anybody reads
    issue(0, 0);
    issue(n_mine > 1 ? 1 : 0, 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    bar();

    // interval k: group (k & 1) reads tile k, the other group multiplies tile k - 1, everybody stages tile k + 2 (behind the last tile:
    // that tile again, into a slot nobody reads any more) and waits for everything but those seven pieces: tile k + 1 is complete
    // behind the barrier. A wave walks its group's tiles i = grp, grp + 2, ...: R in interval i, M in interval i + 1; intervals in
    // which its group has nothing to do (the first for G1, the last for the group that did not get the last tile) only stage.
    auto stage_and_sync = [&](int k) {
        const int nx = k + 2 < n_mine ? k + 2 : n_mine - 1;
        issue(nx, (k + 2) % WG1_NSLOT);
        RSU_WG_WAIT(7);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        bar();
    };
    if (grp) stage_and_sync(0);
    int i = grp;
    for (; i < n_mine; i += 2) {
        bf16x8 fa[4], sv[9];
        // ================= R interval
        {
            __attribute__((address_space(3))) char* sl = lds + (i % WG1_NSLOT) * WG1_SLOT;
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) {
                const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bf16x4*)(sl + fx[0][ct]));
                const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bf16x4*)(sl + fx[1][ct]));
                fa[ct] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
            }
            __attribute__((address_space(3))) char* s0 = sl + sx[0];
            __attribute__((address_space(3))) char* s1 = sl + sx[1];
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bf16x4*)(s0 + t * WG1_SPH));
                const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bf16x4*)(s1 + t * WG1_SPH));
                sv[t] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
            }
        }
        stage_and_sync(i);
        // ================= M interval: 36 MFMAs + 4 for the bias sums
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) mfma_bf16_inplace(acc[t][ct], fa[ct], sv[t]);
        if (do_bias) {
            unsigned o1 = 0x3f803f80u;
            asm volatile("" : "+v"(o1));
            const u32x4 o4 = {o1, o1, o1, o1};
            bf16x8 ones = __builtin_bit_cast(bf16x8, o4);
            asm volatile("s_nop 3" : "+v"(ones));   // VALU-written operand -> (asm) MFMA read
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) mfma_bf16_inplace(accb[ct], fa[ct], ones);
        }
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        stage_and_sync(i + 1);
    }
    if (i == n_mine) stage_and_sync(n_mine);   // (i ends at n_mine or n_mine + 1: the group without the last tile sits out the last interval)
    mfma_results_fence();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // nothing may land in LDS after this point
    __syncthreads();

    // ---- the eight partial sums meet in wave 0 through LDS (the ring is dead): waves 4-7 -> 0-3 (in two halves: 4 x 40 KiB do not fit),
    // 2-3 -> 0-1, 1 -> 0; fixed order
    f32x4 __attribute__((address_space(3)))* red = (f32x4 __attribute__((address_space(3)))*)lds;
    auto handoff = [&](auto src_lo_c, auto nsrc_c, auto t_lo_c, auto t_hi_c, auto bias_c) {   // waves src_lo .. src_lo+nsrc-1 add into src_lo-nsrc .. src_lo-1
        constexpr int SRC = decltype(src_lo_c)::value, NS = decltype(nsrc_c)::value, TLO = decltype(t_lo_c)::value, THI = decltype(t_hi_c)::value;
        constexpr bool BIAS = decltype(bias_c)::value;
        const int k = wave - SRC;
        if (k >= 0 && k < NS) {
#pragma unroll
            for (int t = TLO; t < THI; ++t)
#pragma unroll
                for (int a = 0; a < 4; ++a) red[((t - TLO) * 4 + a) * (NS * 64) + k * 64 + lane] = acc[t][a];
            if (BIAS) {
#pragma unroll
                for (int a = 0; a < 4; ++a) red[((THI - TLO) * 4 + a) * (NS * 64) + k * 64 + lane] = accb[a];
            }
        }
        __syncthreads();
        const int d = wave - (SRC - NS);
        if (d >= 0 && d < NS) {
#pragma unroll
            for (int t = TLO; t < THI; ++t)
#pragma unroll
                for (int a = 0; a < 4; ++a) acc[t][a] += red[((t - TLO) * 4 + a) * (NS * 64) + d * 64 + lane];
            if (BIAS) {
#pragma unroll
                for (int a = 0; a < 4; ++a) accb[a] += red[((THI - TLO) * 4 + a) * (NS * 64) + d * 64 + lane];
            }
        }
        __syncthreads();
    };
    using std::integral_constant;
    handoff(integral_constant<int, 4>{}, integral_constant<int, 4>{}, integral_constant<int, 0>{}, integral_constant<int, 5>{}, integral_constant<bool, false>{});
    handoff(integral_constant<int, 4>{}, integral_constant<int, 4>{}, integral_constant<int, 5>{}, integral_constant<int, 9>{}, integral_constant<bool, true>{});
    handoff(integral_constant<int, 2>{}, integral_constant<int, 2>{}, integral_constant<int, 0>{}, integral_constant<int, 9>{}, integral_constant<bool, true>{});
    handoff(integral_constant<int, 1>{}, integral_constant<int, 1>{}, integral_constant<int, 0>{}, integral_constant<int, 9>{}, integral_constant<bool, true>{});
    if (wave != 0) return;

    if (do_bias && l15 == 0) {   // every column of accb holds the same sums: column 0 writes them (4 consecutive channels per lane)
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) {
            const int cf = cfb * 64 + ct * 16 + 4 * g4;
            if (cf < p.Cout) *(f32x4*)(p.bslab + (long)z * p.slab_stride + cf) = accb[ct];
        }
    }
    // ---- this split's slab: rows = co (4 consecutive per lane), cols = ci
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) {
            const int cf = cfb * 64 + ct * 16 + 4 * g4;
            if (cf >= p.Cout) continue;
            *(f32x4*)(p.slab + (long)z * p.slab_stride + (((long)t * 16 + l15) * p.Cout + cf)) = acc[t][ct];
        }
}

bool igemm_wg1_supports(int N, int H, int W, int Cout, int dil) {
    const int Ho = H - 2 * dil, Wo = W - 2 * dil;
    return Cout % 8 == 0 && Ho >= 1 && Wo >= 1 && (long)N * Ho * Wo < (1l << 28) && (long)N * Ho * Wo * Cout * 2 < 0x7ffffff0L &&
           (long)N * H * W * 32 < 0x7ffffff0L;
}
int igemm_wg1_blocks(int Cout) { return (Cout + 63) / 64; }
int igemm_wg1_tiles(int N, int Ho, int Wo) { return (int)(((long)N * Ho * Wo + 127) / 128); }
static unsigned wg1_magic(int d) { return d <= 1 ? 0xffffffffu : (unsigned)(0x100000000ull / (unsigned)d); }
hipError_t igemm_wg1_launch(const void* in16, const void* dz, float* slab, float* bslab, long slab_stride, int N, int H, int W, int Cout, int dil,
                            int nsplit, hipStream_t st) {
    IgWg1Params p;
    p.in16 = (const bf16_t*)in16; p.dz = (const bf16_t*)dz; p.slab = slab; p.bslab = bslab; p.slab_stride = slab_stride;
    p.N = N; p.H = H; p.W = W; p.Ho = H - 2 * dil; p.Wo = W - 2 * dil; p.Cout = Cout; p.dil = dil;
    p.ntiles = igemm_wg1_tiles(N, p.Ho, p.Wo);
    p.nsplit = nsplit < 1 ? 1 : (nsplit > p.ntiles ? p.ntiles : nsplit);
    p.wo_magic = wg1_magic(p.Wo);
    p.ho_magic = wg1_magic(p.Ho);
    const size_t lds = (size_t)WG1_NSLOT * WG1_SLOT + 1024;   // 157 KiB
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)igemm_wg1_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    hipLaunchKernelGGL(igemm_wg1_kernel, dim3((Cout + 63) / 64, p.nsplit), dim3(512), lds, st, p);
    return hipGetLastError();
}
