// conv_first: forward of the network's first 3x3 convolution (unet.py:35-38 on the colour-adjusted input; rsu_conv_first_fwd).
//
// The layer has 16 input channels (in16 of k_color_adjust: 3 colours + the 13 helper channels of the colour-adjust gradient, whose
// forward weights are zero) and K = 9 x 16 = 144: ~2 us of MFMA work against 208 MB of traffic (42 MB read, 166 MB written at
// B = 4, 572 px). It is an HBM-bound copy with a few MFMAs inside, so it is built like one: no LDS, no persistent tile stream, no
// barriers -- many independent waves, each with the WHOLE weight matrix in registers.
//   * a wave owns 64 output channels (the layer's Cout = root; more channels = more blocks in y) and walks 16-pixel pieces of
//     output rows; MFMA 16x16x32 with A = weights, B = pixels, like every other kernel of the library (same packed weights, same
//     channel-permuted 16-byte stores);
//   * a k-step of 32 is TWO taps x 16 channels: lane (l15, g4) supplies pixel l15 of tap 2j + (g4 >> 1), channels 8 * (g4 & 1) .. + 8:
//     one 16-byte load straight from global memory (the 42-MB input stays in L1 / L2 over its 9 taps); 5 k-steps, the second half
//     of the last one has zero weights;
//   * the next piece's five loads are in flight while the current piece's 20 MFMAs and two stores run.
// Summation order: taps in pairs (0,1) (2,3) ... inside one fp32 accumulator -- igemm_fwd2 sums tap by tap; the results differ by
// fp32 rounding of a 144-term sum, far below the bf16 rounding of the output.
#include "igemm.h"

namespace {
__device__ __forceinline__ unsigned relu_pk(unsigned x) {
    typedef __attribute__((ext_vector_type(2))) short s2;
    return __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s2, x), s2{0, 0}));
}
}  // namespace

struct ConvFirstParams {
    const float* x;       // FUSED: the network input, f32 [N][H][W][3]; colour adjust (unet.py:22-23) is computed on the fly
    const float* cw;      // FUSED: color_space_adjust kernel [ci][cj] (device)
    const float* cb;      // FUSED: color_space_adjust bias [3] (device)
    const bf16_t* in16;   // [N][H][W][16]
    const bf16_t* wp;     // packed forward weights [1 chunk][9 taps][ntiles_w][64 lanes][8]
    const float* bias;    // [Cout] or null
    bf16_t* y;            // [N][Ho][Wo][Cout]
    int N, H, W, Ho, Wo, Cout, dil, relu, ntiles_w;
    int ntx;              // 16-pixel pieces per output row
    unsigned ntx_magic, ho_magic;   // floor(2^32 / ntx), floor(2^32 / Ho) (0xffffffff for a divisor of 1)
    int npieces;          // N * Ho * ntx
};

// FUSED: the B operand is made from the f32 input instead of read from in16 (k_color_adjust's output): a lane of the low channel half
// loads its pixel's three colours (12 bytes), applies net0 = (x - 0.5) W0 + b0 with k_color_adjust's own expression (same fmaf chain, same
// bf16 rounding: the kernel's results are bit-identical to the two-launch path) and supplies {net0, 0 x 5}; the high half supplies zeros
// (its weights are zero in either path). in16 is then neither written nor read in a forward-only net (-58 MB and one launch per forward
// pass); a training step still needs it for the first conv's weight gradient, but writes it off the critical path.
// No dropout here (keep == 1 only: the mask of keep < 1 would be hashed nine times per pixel).
#ifndef CF_TRS
#define CF_TRS 1   // developer A/B switch: the output piece transposed through LDS into whole-pixel stores (0: the MFMA layout's stores of rounds 3-5)
#endif
template <bool FUSED>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3))) k_conv_first_fwd(const ConvFirstParams p) {
#if CF_TRS
    __shared__ __attribute__((aligned(16))) char trs[4 * 2048];
#endif
    const int lane = threadIdx.x & 63, l15 = lane & 15, g4 = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + (threadIdx.x >> 6)));
    const int nwaves = gridDim.x * 4;
    const int cb = blockIdx.y;   // block of 64 output channels
    const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc((void*)p.in16, 0, (int)((long)p.N * p.H * p.W * 32), 0x00020000);
    const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc((void*)p.y, 0, 0x7fffffff, 0x00020000);
    // ---- the weights of this channel block: 5 k-steps x 4 tiles of 16 channels, 80 registers for the life of the wave
    bf16x8 fa[5][4];
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        const int tap = 2 * j + (g4 >> 1);
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) {
            const int tile = cb * 4 + ct;
            bf16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
            if (tap < 9 && tile < p.ntiles_w) v = *(const bf16x8*)(p.wp + ((long)(tap * p.ntiles_w + tile) * 64 + l15 + 16 * (g4 & 1)) * 8);
            fa[j][ct] = v;
        }
    }
    // accumulator [ct][i] = channel (ct >> 1) * 32 + 8 * g4 + (ct & 1) * 4 + i of the block (the packed weights' row order)
    f32x4 binit[4];
#pragma unroll
    for (int ct = 0; ct < 4; ++ct)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int co = cb * 64 + (ct >> 1) * 32 + 8 * g4 + (ct & 1) * 4 + i;
            binit[ct][i] = (p.bias && co < p.Cout) ? p.bias[co] : 0.f;
        }
    // per-lane byte offset of the lane's 16 bytes of k-step j, relative to the piece's first pixel at tap (0, 0)
    unsigned voff[5];
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        const int tap = min(2 * j + (g4 >> 1), 8), ky = tap / 3, kx = tap - 3 * ky;
        voff[j] = (unsigned)((((ky * p.W + kx) * p.dil + l15) * 16 + 8 * (g4 & 1)) * 2);
    }
    auto decode = [&](int t, int& n, int& y, int& x0) {
        unsigned row = __umulhi((unsigned)t, p.ntx_magic);
        int xt = t - (int)row * p.ntx;
        if (xt >= p.ntx) { ++row; xt -= p.ntx; }
        unsigned nn = __umulhi(row, p.ho_magic);
        int yy = (int)row - (int)nn * p.Ho;
        if (yy >= p.Ho) { ++nn; yy -= p.Ho; }
        n = (int)nn; y = yy; x0 = xt * 16;
    };
    typedef __attribute__((ext_vector_type(3))) float f32x3;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, FUSED ? (int)((long)p.N * p.H * p.W * 12) : 0, 0x00020000);
    unsigned xoff[5];     // FUSED: byte offset of the lane's pixel (3 floats) of k-step j; the high channel half asks for nothing
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        const int tap = min(2 * j + (g4 >> 1), 8), ky = tap / 3, kx = tap - 3 * ky;
        xoff[j] = (g4 & 1) ? 0x80000000u : (unsigned)(((ky * p.W + kx) * p.dil + l15) * 12);
    }
    auto load = [&](int t, bf16x8 (&fb)[5]) {
        int n, y, x0;
        decode(t, n, y, x0);
        if constexpr (FUSED) {
            const unsigned soff = (unsigned)(((long)(n * p.H + y) * p.W + x0) * 12);
#pragma unroll
            for (int j = 0; j < 5; ++j) {
                // (the three colours ride in the fragment's registers until the MFMAs need them: convert() below)
                const f32x3 v = __builtin_bit_cast(f32x3, __builtin_amdgcn_raw_buffer_load_b96(rx, xoff[j], soff, 0));
                fb[j] = __builtin_bit_cast(bf16x8, f32x4{v[0], v[1], v[2], 0.f});
            }
        } else {
            const unsigned soff = (unsigned)((((long)(n * p.H + y) * p.W + x0) * 16) * 2);
#pragma unroll
            for (int j = 0; j < 5; ++j) fb[j] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rin, voff[j], soff, 0));
        }
    };
    float cw[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, cbias[3] = {0.f, 0.f, 0.f};
    if constexpr (FUSED) {   // (uniform addresses: scalar loads, twelve SGPRs for the life of the wave)
#pragma unroll
        for (int i = 0; i < 9; ++i) cw[i] = p.cw[i];
#pragma unroll
        for (int i = 0; i < 3; ++i) cbias[i] = p.cb[i];
    }
    auto convert = [&](bf16x8 (&fb)[5]) {   // FUSED: colours -> {bf16 net0[0..2], 0 x 5} (k_color_adjust's arithmetic, keep == 1)
        if constexpr (FUSED) {
#pragma unroll
            for (int j = 0; j < 5; ++j) {
                const f32x4 v = __builtin_bit_cast(f32x4, fb[j]);
                const float xc[3] = {v[0] - 0.5f, v[1] - 0.5f, v[2] - 0.5f};
                float f[3];
#pragma unroll
                for (int c = 0; c < 3; ++c) f[c] = fmaf(xc[2], cw[6 + c], fmaf(xc[1], cw[3 + c], fmaf(xc[0], cw[c], cbias[c])));   // (x (m * inv) = 1 at keep == 1)
                const u32x4 o = {pack_bf2(f[0], f[1]), pack_bf2(f[2], 0.f), 0u, 0u};
                fb[j] = (g4 & 1) ? bf16x8{0, 0, 0, 0, 0, 0, 0, 0} : __builtin_bit_cast(bf16x8, o);
            }
        }
    };
    int t = wave;
    if (t >= p.npieces) return;
    bf16x8 fb[5], fbn[5];
    load(t, fb);
    while (true) {
        const int tn = t + nwaves;
        const bool more = tn < p.npieces;   // (wave-uniform)
        // (always issued -- behind the last piece the same piece again: with a branch around the loads the compiler's s_waitcnt in front of
        // the MFMAs has to assume that nothing was issued, and waits for the prefetch as well)
        load(more ? tn : t, fbn);
        f32x4 acc[4];
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) acc[ct] = binit[ct];
        convert(fb);
#pragma unroll
        for (int j = 0; j < 5; ++j)
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) acc[ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[j][ct], fb[j], acc[ct], 0, 0, 0);
        int n, y, x0;
        decode(t, n, y, x0);
        const unsigned sbase = (unsigned)((((long)(n * p.Ho + y) * p.Wo + x0) * p.Cout + cb * 64) * 2);
#if CF_TRS
        // The MFMA's lane layout has lane (l15, g4) hold 16 bytes of pixel l15: a store instruction then writes sixteen 64-byte runs one pixel
        // apart (the slow pattern of probes/probe_store_shapes.hip: ~3x the cycles of whole lines). This kernel is all stores (166 of 182 MB), so
        // the piece's 16 pixels x 128 bytes take a turn through 2 KiB of wave-private LDS (16-byte slots XOR-swizzled by the pixel) and leave as
        // two instructions of eight whole pixels each: 8 consecutive lanes = the 128 bytes of one pixel, 1 KiB contiguous where Cout = 64.
        {
            __attribute__((address_space(3))) char* wl = (__attribute__((address_space(3))) char*)trs + (threadIdx.x >> 6) * 2048;
#pragma unroll
            for (int pp = 0; pp < 2; ++pp) {
                u32x4 r = {pack_bf2(acc[2 * pp][0], acc[2 * pp][1]), pack_bf2(acc[2 * pp][2], acc[2 * pp][3]),
                           pack_bf2(acc[2 * pp + 1][0], acc[2 * pp + 1][1]), pack_bf2(acc[2 * pp + 1][2], acc[2 * pp + 1][3])};
                if (p.relu) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) r[i] = relu_pk(r[i]);
                }
                *(__attribute__((address_space(3))) u32x4*)(wl + l15 * 128 + (((pp * 4 + g4) ^ (l15 & 7)) << 4)) = r;
            }
            // (a wave's LDS instructions execute in order: the reads below see the writes above; the asm keeps the COMPILER from moving a
            // lane's read -- of another lane's slot -- in front of its own write, and the next piece's writes in front of these reads)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const int q8 = lane & 7;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int P = (lane >> 3) + 8 * h;
                const u32x4 r = *(const __attribute__((address_space(3))) u32x4*)(wl + P * 128 + ((q8 ^ (P & 7)) << 4));
                const bool ok = (x0 + P < p.Wo) && (cb * 64 + q8 * 8 < p.Cout);
                const unsigned vo = ok ? (unsigned)((P * p.Cout + q8 * 8) * 2) : 0x80000000u;
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_nop 4\n\tbuffer_store_dwordx4 %0, %1, %2, %3 offen\n\ts_nop 3" ::"v"(r), "v"(vo), "s"(rout), "s"(sbase) : "memory");
            }
        }
#else
        const bool pok = x0 + l15 < p.Wo;
#pragma unroll
        for (int pp = 0; pp < 2; ++pp) {
            u32x4 r = {pack_bf2(acc[2 * pp][0], acc[2 * pp][1]), pack_bf2(acc[2 * pp][2], acc[2 * pp][3]),
                       pack_bf2(acc[2 * pp + 1][0], acc[2 * pp + 1][1]), pack_bf2(acc[2 * pp + 1][2], acc[2 * pp + 1][3])};
            if (p.relu) {
#pragma unroll
                for (int i = 0; i < 4; ++i) r[i] = relu_pk(r[i]);
            }
            const bool cok = cb * 64 + pp * 32 + 8 * g4 < p.Cout;
            const unsigned vo = (pok && cok) ? (unsigned)((l15 * p.Cout + pp * 32 + 8 * g4) * 2) : 0x80000000u;
            // (store + its wait states as one asm statement: on gfx950 a VALU write of the data registers right behind a 128-bit store with
            // an SGPR offset can reach the store's last lanes, and hipcc adds no wait state there -- DESIGN.md section 4)
            asm volatile("s_nop 4\n\tbuffer_store_dwordx4 %0, %1, %2, %3 offen\n\ts_nop 3" ::"v"(r), "v"(vo), "s"(rout), "s"(sbase) : "memory");
        }
#endif
        if (!more) break;
        t = tn;
#pragma unroll
        for (int j = 0; j < 5; ++j) fb[j] = fbn[j];
    }
}

static unsigned floor_magic(int d) { return d <= 1 ? 0xffffffffu : (unsigned)(0x100000000ull / (unsigned)d); }

// in16 [N][H][W][16] -> y [N][H - 2 dil][W - 2 dil][Cout]; wp = the layer's packed forward weights (ntiles_w 16-row tiles per tap)
// x / cw / cb != null: the fused form (the colour adjust computed from the f32 input; cw [3][3], cb [3] device pointers); in16 is then unused
hipError_t conv_first_fwd_launch(const void* in16, const void* wp, int ntiles_w, const float* bias, void* y, int N, int H, int W, int Cout, int dil,
                                 int relu, int ncu, hipStream_t st, const float* x, const float* cw, const float* cb) {
    ConvFirstParams p;
    p.x = x; p.cw = cw; p.cb = cb;
    p.in16 = (const bf16_t*)in16; p.wp = (const bf16_t*)wp; p.bias = bias; p.y = (bf16_t*)y;
    p.N = N; p.H = H; p.W = W; p.Ho = H - 2 * dil; p.Wo = W - 2 * dil; p.Cout = Cout; p.dil = dil; p.relu = relu; p.ntiles_w = ntiles_w;
    p.ntx = (p.Wo + 15) / 16;
    p.ntx_magic = floor_magic(p.ntx);
    p.ho_magic = floor_magic(p.Ho);
    const long np = (long)N * p.Ho * p.ntx;
    if (np >= (1l << 31) / 2 || p.Ho < 1 || p.Wo < 1) return hipErrorInvalidValue;
    p.npieces = (int)np;
    long blocks = (long)ncu * 3;                    // 3 blocks of 4 waves per CU: ~160 registers per lane
    if (blocks * 4 > np) blocks = (np + 3) / 4;
    if (x) hipLaunchKernelGGL(k_conv_first_fwd<true>, dim3((unsigned)blocks, (unsigned)((Cout + 63) / 64)), dim3(256), 0, st, p);
    else hipLaunchKernelGGL(k_conv_first_fwd<false>, dim3((unsigned)blocks, (unsigned)((Cout + 63) / 64)), dim3(256), 0, st, p);
    return hipGetLastError();
}
