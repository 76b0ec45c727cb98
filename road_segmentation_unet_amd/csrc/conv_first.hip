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
    const bf16_t* in16;   // [N][H][W][16]
    const bf16_t* wp;     // packed forward weights [1 chunk][9 taps][ntiles_w][64 lanes][8]
    const float* bias;    // [Cout] or null
    bf16_t* y;            // [N][Ho][Wo][Cout]
    int N, H, W, Ho, Wo, Cout, dil, relu, ntiles_w;
    int ntx;              // 16-pixel pieces per output row
    unsigned ntx_magic, ho_magic;   // floor(2^32 / ntx), floor(2^32 / Ho) (0xffffffff for a divisor of 1)
    int npieces;          // N * Ho * ntx
};

__global__ void __launch_bounds__(256) k_conv_first_fwd(const ConvFirstParams p) {
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
    auto load = [&](int t, bf16x8 (&fb)[5]) {
        int n, y, x0;
        decode(t, n, y, x0);
        const unsigned soff = (unsigned)((((long)(n * p.H + y) * p.W + x0) * 16) * 2);
#pragma unroll
        for (int j = 0; j < 5; ++j) fb[j] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rin, voff[j], soff, 0));
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
#pragma unroll
        for (int j = 0; j < 5; ++j)
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) acc[ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[j][ct], fb[j], acc[ct], 0, 0, 0);
        int n, y, x0;
        decode(t, n, y, x0);
        const bool pok = x0 + l15 < p.Wo;
        const unsigned sbase = (unsigned)((((long)(n * p.Ho + y) * p.Wo + x0) * p.Cout + cb * 64) * 2);
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
        if (!more) break;
        t = tn;
#pragma unroll
        for (int j = 0; j < 5; ++j) fb[j] = fbn[j];
    }
}

static unsigned floor_magic(int d) { return d <= 1 ? 0xffffffffu : (unsigned)(0x100000000ull / (unsigned)d); }

// in16 [N][H][W][16] -> y [N][H - 2 dil][W - 2 dil][Cout]; wp = the layer's packed forward weights (ntiles_w 16-row tiles per tap)
hipError_t conv_first_fwd_launch(const void* in16, const void* wp, int ntiles_w, const float* bias, void* y, int N, int H, int W, int Cout, int dil,
                                 int relu, int ncu, hipStream_t st) {
    ConvFirstParams p;
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
    hipLaunchKernelGGL(k_conv_first_fwd, dim3((unsigned)blocks, (unsigned)((Cout + 63) / 64)), dim3(256), 0, st, p);
    return hipGetLastError();
}
