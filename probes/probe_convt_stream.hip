// convt_stream: the 2x2 stride-2 transposed convolution (unet.py:67-68), forward and backward-data, as a STREAMING GEMM.
//
// Both directions are GEMMs over the low-resolution pixels m = (n, y, x) with a short reduction (forward K = Cin = 128 .. 1024,
// backward-data K = 4 taps x Cout) whose k index is the CHANNEL -- contiguous in NHWC. A lane's MFMA fragment (one pixel, 8
// consecutive channels) is therefore one 16-byte load from global memory, and the packed weights are stored in fragment order
// (1 KiB per fragment, contiguous). igemm_ct stages both through LDS in a persistent ping-pong stream: for these shapes that is
// bound by the LDS-DMA path (~29 B/clk/CU) and by tile boundaries (2-16 stages per tile), and the small levels leave CUs idle
// (104-208 tiles). Here nothing is staged: 12 independent waves per CU, each computes a 64-pixel x 64-column block -- per 32-channel
// k-step four weight fragments and four pixel fragments straight into registers (the next step's eight loads in flight during this
// step's 16 MFMAs), then eight 16-byte stores. Weights come from L2 (<= 4 MB per layer), the pixel block of a wave is shared by
// the waves beside it (the column blocks of one pixel block are consecutive units): L1 / L2 hits.
//   mode 0, forward:        y[n][2y+a][2x+b][co] = relu(bias[co] + sum_ci x[m][ci] K[a][b][co][ci]); unit = (pixel block, tap, 64 co)
//   mode 1, backward-data:  dx[m][ci] = (x[m][ci] > 0) * sum_{a,b,co} dy[n][2y+a][2x+b][co] K[a][b][co][ci]; unit = (pixel block, 64 ci)
// Same packed weights, same accumulation order inside a k-step and over the k-steps as igemm_ct (chunk by chunk; backward: chunk
// outer, tap inner as packed).
#include "igemm.h"

namespace {
__device__ __forceinline__ unsigned relu_pk(unsigned x) {
    typedef __attribute__((ext_vector_type(2))) short s2;
    return __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s2, x), s2{0, 0}));
}
}  // namespace

struct CtStreamParams {
    const bf16_t* src;      // mode 0: x [N][H][W][Cin]; mode 1: dy [N][2H][2W][Cout]
    const bf16_t* wp;       // packed weights (rsu_pack_convT_fwd: four 1-tap matrices `tap_stride` elements apart; rsu_pack_convT_bwd: one 4-tap matrix)
    const float* bias;      // mode 0: [Cout] or null
    const bf16_t* mask;     // mode 1: ReLU source x [N][H][W][Cin] or null
    bf16_t* out;            // mode 0: y [N][2H][2W][Cout]; mode 1: dx [N][H][W][Cin]
    long tap_stride;        // mode 0: elements between the per-tap matrices
    int H, W, P;            // low-resolution geometry, P = N * H * W
    int Ck, Cn;             // channels of src (the k side) and of out (the column side)
    int nchunk;             // 32-channel chunks of Ck
    int ntiles_w;           // 16-row tiles per (chunk, tap) of the packed weights
    int ncb;                // 64-column blocks of Cn
    int ncol;               // units per pixel block: mode 0 4 * ncb, mode 1 ncb
    int nunits;
    unsigned w_magic;       // floor(2^32 / W)
    long src_bytes;
};

template <int MODE>
__global__ void __launch_bounds__(256) k_convt_stream(const CtStreamParams p) {
    const int lane = threadIdx.x & 63, l15 = lane & 15, g4 = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + (threadIdx.x >> 6)));
    const int nwaves = gridDim.x * 4;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.src, 0, (int)p.src_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc((void*)p.out, 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rmask = __builtin_amdgcn_make_buffer_rsrc((void*)(p.mask ? p.mask : p.out), 0, 0x7fffffff, 0x00020000);
    const int W2 = 2 * p.W;
    for (int u = wave; u < p.nunits; u += nwaves) {
        const int pb = u / p.ncol, col = u - pb * p.ncol;          // (wave-uniform)
        const int tap = MODE == 0 ? col / p.ncb : 0, cb = MODE == 0 ? col - tap * p.ncb : col;
        const int m0 = pb * 64;
        // ---- this lane's four pixels: byte offset inside src (tap (0,0) for mode 1) and inside out
        unsigned boff[4], ooff[4];
        bool pok[4];
#pragma unroll
        for (int pt = 0; pt < 4; ++pt) {
            const int m = m0 + pt * 16 + l15;
            pok[pt] = m < p.P;
            const int mc = pok[pt] ? m : p.P - 1;
            unsigned q = __umulhi((unsigned)mc, p.w_magic);        // q = n * H + y (floor(2^32 / W) may fall one short)
            int x = mc - (int)q * p.W;
            if (x >= p.W) { ++q; x -= p.W; }
            const unsigned hi = (unsigned)((2 * (int)q) * W2 + 2 * x);   // high-resolution pixel (2y, 2x) of image n, flat
            if (MODE == 0) {
                boff[pt] = (unsigned)mc * (unsigned)(p.Ck * 2) + (unsigned)(g4 * 16);
                ooff[pt] = (hi + (unsigned)((tap >> 1) * W2 + (tap & 1))) * (unsigned)(p.Cn * 2);
            } else {
                boff[pt] = hi * (unsigned)(p.Ck * 2) + (unsigned)(g4 * 16);
                ooff[pt] = (unsigned)mc * (unsigned)(p.Cn * 2);
            }
        }
        // ---- accumulators: [column tile][pixel group]; element i of tile ct = column (ct >> 1) * 32 + 8 * g4 + (ct & 1) * 4 + i of the block
        f32x4 acc[4][4];
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) {
            f32x4 b = {0.f, 0.f, 0.f, 0.f};
            if (MODE == 0 && p.bias) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int co = cb * 64 + (ct >> 1) * 32 + 8 * g4 + (ct & 1) * 4 + i;
                    b[i] = co < p.Cn ? p.bias[co] : 0.f;
                }
            }
#pragma unroll
            for (int pt = 0; pt < 4; ++pt) acc[ct][pt] = b;
        }
        // ---- the k-steps: mode 0 chunk c of the unit's tap; mode 1 (chunk c, tap t), t fastest (the packed order)
        const int nsteps = MODE == 0 ? p.nchunk : 4 * p.nchunk;
        const bf16_t* wbase = p.wp + (MODE == 0 ? (long)tap * p.tap_stride : 0l) + ((long)cb * 4 * 64 + lane) * 8;
        const long wstep = (long)p.ntiles_w * 512;   // elements per (chunk, tap)
        auto load = [&](int s, bf16x8 (&fa)[4], bf16x8 (&fb)[4]) {
            const bf16_t* wq = wbase + s * wstep;
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) {
                bf16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
                if (cb * 4 + ct < p.ntiles_w) v = *(const bf16x8*)(wq + ct * 512);   // (wave-uniform test)
                fa[ct] = v;
            }
            unsigned soff;
            if (MODE == 0) {
                soff = (unsigned)(s * 64);
            } else {
                const int c = s >> 2, t = s & 3;
                soff = (unsigned)(((t >> 1) * W2 + (t & 1)) * p.Ck * 2 + c * 64);
            }
#pragma unroll
            for (int pt = 0; pt < 4; ++pt) fb[pt] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rsrc, boff[pt], soff, 0));
        };
        auto mma = [&](const bf16x8 (&fa)[4], const bf16x8 (&fb)[4]) {
#pragma unroll
            for (int pt = 0; pt < 4; ++pt)
#pragma unroll
                for (int ct = 0; ct < 4; ++ct) acc[ct][pt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[ct], fb[pt], acc[ct][pt], 0, 0, 0);
        };
        bf16x8 a0[4], b0[4], a1[4], b1[4];
        load(0, a0, b0);
        int s = 0;
        for (; s + 2 <= nsteps - 1; s += 2) {   // two register sets, the next step's loads in flight during this step's MFMAs
            load(s + 1, a1, b1);
            mma(a0, b0);
            load(s + 2, a0, b0);
            mma(a1, b1);
        }
        if (s == nsteps - 1) {
            mma(a0, b0);
        } else {
            load(s + 1, a1, b1);
            mma(a0, b0);
            mma(a1, b1);
        }
        // ---- epilogue: 16 bytes (8 consecutive columns) per lane and (pixel group, column half)
        const unsigned cbase = (unsigned)(cb * 64 * 2);
#pragma unroll
        for (int pt = 0; pt < 4; ++pt)
#pragma unroll
            for (int pp = 0; pp < 2; ++pp) {
                u32x4 r = {pack_bf2(acc[2 * pp][pt][0], acc[2 * pp][pt][1]), pack_bf2(acc[2 * pp][pt][2], acc[2 * pp][pt][3]),
                           pack_bf2(acc[2 * pp + 1][pt][0], acc[2 * pp + 1][pt][1]), pack_bf2(acc[2 * pp + 1][pt][2], acc[2 * pp + 1][pt][3])};
                const bool cok = cb * 64 + pp * 32 + 8 * g4 < p.Cn;
                const unsigned vo = (pok[pt] && cok) ? ooff[pt] + (unsigned)((pp * 32 + 8 * g4) * 2) : 0x80000000u;
                if (MODE == 0) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) r[i] = relu_pk(r[i]);
                } else if (p.mask) {
                    const u32x4 mk = __builtin_amdgcn_raw_buffer_load_b128(rmask, vo, cbase, 0);
#pragma unroll
                    for (int i = 0; i < 4; ++i) r[i] &= pos_mask_pk_bf16(mk[i]);
                }
                // (store + its wait states as one asm statement: DESIGN.md section 4, the 128-bit store / VALU-write hazard with an SGPR offset)
                asm volatile("s_nop 4\n\tbuffer_store_dwordx4 %0, %1, %2, %3 offen\n\ts_nop 3" ::"v"(r), "v"(vo), "s"(rout), "s"(cbase) : "memory");
            }
    }
}

static unsigned floor_magic_ct(int d) { return d <= 1 ? 0xffffffffu : (unsigned)(0x100000000ull / (unsigned)d); }

bool convt_stream_supports(int mode, int N, int H, int W, int Ck, int Cn) {
    (void)mode;
    return Ck % 8 == 0 && Cn % 8 == 0 && (long)N * H * W < (1l << 28) && (long)N * 4 * H * W * (Ck > Cn ? Ck : Cn) * 2 < 0x7ffffff0L;
}
// mode 0: src = x, Ck = Cin, Cn = Cout, wp = rsu_pack_convT_fwd's buffer (tap_stride elements per tap); mode 1: src = dy, Ck = Cout, Cn = Cin,
// wp = rsu_pack_convT_bwd's buffer, mask = the ReLU source or null
hipError_t convt_stream_launch(int mode, const void* src, const void* wp, long tap_stride, int ntiles_w, const float* bias, const void* mask, void* out,
                               int N, int H, int W, int Ck, int Cn, int ncu, hipStream_t st) {
    CtStreamParams p;
    p.src = (const bf16_t*)src; p.wp = (const bf16_t*)wp; p.bias = bias; p.mask = (const bf16_t*)mask; p.out = (bf16_t*)out;
    p.tap_stride = tap_stride;
    p.H = H; p.W = W; p.P = N * H * W;
    p.Ck = Ck; p.Cn = Cn;
    p.nchunk = (Ck + 31) / 32;
    p.ntiles_w = ntiles_w;
    p.ncb = (Cn + 63) / 64;
    p.ncol = mode == 0 ? 4 * p.ncb : p.ncb;
    const long nu = (long)((p.P + 63) / 64) * p.ncol;
    if (nu >= (1l << 30)) return hipErrorInvalidValue;
    p.nunits = (int)nu;
    p.w_magic = floor_magic_ct(W);
    p.src_bytes = (long)N * H * W * (mode == 0 ? 1 : 4) * Ck * 2;
    long blocks = (long)ncu * 2;   // 2 blocks of 4 waves per CU (176-220 registers per lane: two waves per SIMD)
    if (blocks * 4 > nu) blocks = (nu + 3) / 4;
    if (mode == 0)
        hipLaunchKernelGGL(k_convt_stream<0>, dim3((unsigned)blocks), dim3(256), 0, st, p);
    else
        hipLaunchKernelGGL(k_convt_stream<1>, dim3((unsigned)blocks), dim3(256), 0, st, p);
    return hipGetLastError();
}
