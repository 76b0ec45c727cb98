// HBM-bound kernels of the U-Net path: 16-byte (8 x bf16) accesses per lane everywhere, coalesced along
// the channel axis of NHWC; wave-level shuffles for the per-pixel reductions of the head.
#include "elementwise.h"

static __device__ __forceinline__ void unpack8(const u32x4 v, float* f) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        f[2 * i] = bf_lo(v[i]);
        f[2 * i + 1] = bf_hi(v[i]);
    }
}
static __device__ __forceinline__ u32x4 pack8(const float* f) {
    u32x4 r;
#pragma unroll
    for (int i = 0; i < 4; ++i) r[i] = pack_bf2(f[2 * i], f[2 * i + 1]);
    return r;
}

// ---------------------------------------------------------------------------------------------
// unet.py:22-23  color_space_adjust
// ---------------------------------------------------------------------------------------------
// tf.nn.dropout (unet.py:29-30, 64-65): y = x / keep * floor(keep + U[0,1)). TensorFlow's Philox stream cannot be reproduced, so
// U comes from a counter-based hash of (key, element index) that the test-side CPU restatement reproduces bit for bit. key = f(seed, dropout site, step) is chosen by the host; element index = NHWC linear index of the tensor being dropped.
__device__ __forceinline__ float drop_keep(unsigned key, unsigned idx, float keep) {
    unsigned h = idx ^ key;
    h ^= h >> 16;
    h *= 0x85ebca6bu;
    h ^= h >> 13;
    h *= 0xc2b2ae35u;
    h ^= h >> 16;
    const float u = (float)(h >> 8) * (1.0f / 16777216.0f);
    return floorf(keep + u);  // 1 = kept, 0 = dropped
}

// out16 = { dropout(net0)[0..2], 0, m[cj] * (x-0.5)[ci] at 4 + 3*ci + cj, m[0..2] at 13..15 } with m = 1 without dropout
__global__ void k_color_adjust(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ b,
                               bf16_t* __restrict__ out16, long npix, float keep, unsigned key) {
    const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= npix) return;
    const float xc[3] = {x[3 * p] - 0.5f, x[3 * p + 1] - 0.5f, x[3 * p + 2] - 0.5f};
    float m[3] = {1.f, 1.f, 1.f};
    float inv = 1.f;
    if (keep < 1.f) {
        inv = 1.f / keep;
#pragma unroll
        for (int c = 0; c < 3; ++c) m[c] = drop_keep(key, (unsigned)(3 * p + c), keep);
    }
    float f[16];
#pragma unroll
    for (int c = 0; c < 3; ++c) f[c] = fmaf(xc[2], w[6 + c], fmaf(xc[1], w[3 + c], fmaf(xc[0], w[c], b[c]))) * (m[c] * inv);
    f[3] = 0.f;
#pragma unroll
    for (int ci = 0; ci < 3; ++ci)
#pragma unroll
        for (int cj = 0; cj < 3; ++cj) f[4 + 3 * ci + cj] = xc[ci] * m[cj];
#pragma unroll
    for (int cj = 0; cj < 3; ++cj) f[13 + cj] = m[cj];
    u32x4* o = (u32x4*)(out16 + 16 * p);
    o[0] = pack8(f);
    o[1] = pack8(f + 8);
}

// generic dropout of a bf16 tensor (decoder: input of each transposed conv), 8 elements per thread
__global__ void k_dropout(const bf16_t* __restrict__ x, bf16_t* __restrict__ y, long n8, float keep, unsigned key) {
    const float inv = 1.f / keep;
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < n8; t += (long)gridDim.x * blockDim.x) {
        float v[8];
        unpack8(*(const u32x4*)(x + 8 * t), v);
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] *= drop_keep(key, (unsigned)(8 * t + i), keep) * inv;
        *(u32x4*)(y + 8 * t) = pack8(v);
    }
}

// ---------------------------------------------------------------------------------------------
// level-0 conv1 gradients: rows of the 16-channel wgrad result -> dW1 / colour-adjust helper sums
// ---------------------------------------------------------------------------------------------
__global__ void k_scatter_first_grads(const float* __restrict__ tmp, float* __restrict__ dw1, float* __restrict__ gx, int Cout) {
    // tmp [9][16][Cout] -> dw1 [9][3][Cout] (rows 0..2) and gx [9][12][Cout] (rows 4..15: 9 masked (x-0.5) products, 3 mask sums)
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 9 * 12 * Cout) return;
    const int co = i % Cout, r = i / Cout, row = r % 12, tap = r / 12;
    if (gx) gx[i] = tmp[((long)tap * 16 + 4 + row) * Cout + co];
    if (row < 3) dw1[((long)tap * 3 + row) * Cout + co] = tmp[((long)tap * 16 + row) * Cout + co];
}

// ---------------------------------------------------------------------------------------------
// 2x2 max pool forward
// ---------------------------------------------------------------------------------------------
// `code` (optional, [N][H/2][W/2][C] bytes): per pooled element, bits 0-3 = (window element k > 0) for k = 2*dy + dx, bits 4-5 =
// the window's first maximum in row-major order -- everything the MaxPoolGrad + ReluGrad junction needs from the activation
// (k_pool_skip_relu_bwd then reads one byte per pooled element instead of four bf16 activations)
__global__ void k_maxpool_fwd(const bf16_t* __restrict__ x, bf16_t* __restrict__ y, unsigned char* __restrict__ code, int N, int H, int W, int C,
                              float keep, unsigned key) {
    const float inv = keep < 1.f ? 1.f / keep : 1.f;
    const int Ho = H >> 1, Wo = W >> 1, ncg = C >> 3;
    const long total = (long)N * Ho * Wo * ncg;
    for (long tid = (long)blockIdx.x * blockDim.x + threadIdx.x; tid < total; tid += (long)gridDim.x * blockDim.x) {
        const int cg = (int)(tid % ncg);
        long r = tid / ncg;
        const int ox = (int)(r % Wo);
        r /= Wo;
        const int oy = (int)(r % Ho), n = (int)(r / Ho);
        const bf16_t* p = x + ((long)(n * H + 2 * oy) * W + 2 * ox) * C + cg * 8;
        float a[8], b[8], c[8], d[8];
        unpack8(*(const u32x4*)p, a);
        unpack8(*(const u32x4*)(p + C), b);
        unpack8(*(const u32x4*)(p + (long)W * C), c);
        unpack8(*(const u32x4*)(p + (long)W * C + C), d);
        const long oidx = ((long)(n * Ho + oy) * Wo + ox) * C + cg * 8;
        if (code) {
            unsigned cw[2] = {0u, 0u};
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                int best = 0;
                float m = a[i];
                if (b[i] > m) { m = b[i]; best = 1; }
                if (c[i] > m) { m = c[i]; best = 2; }
                if (d[i] > m) { m = d[i]; best = 3; }
                const unsigned byte = (a[i] > 0.f ? 1u : 0u) | (b[i] > 0.f ? 2u : 0u) | (c[i] > 0.f ? 4u : 0u) | (d[i] > 0.f ? 8u : 0u) | ((unsigned)best << 4);
                cw[i >> 2] |= byte << (8 * (i & 3));
            }
            *(uint2*)(code + oidx) = make_uint2(cw[0], cw[1]);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) a[i] = fmaxf(fmaxf(a[i], b[i]), fmaxf(c[i], d[i]));
        if (keep < 1.f) {  // dropout of the next level's input (unet.py:29-30), fused
#pragma unroll
            for (int i = 0; i < 8; ++i) a[i] *= drop_keep(key, (unsigned)(oidx + i), keep) * inv;
        }
        *(u32x4*)(y + oidx) = pack8(a);
    }
}

// ---------------------------------------------------------------------------------------------
// gradient junction at an encoder output, the training path (code bytes, whole windows, a pooled gradient): one thread = 8 channels
// of one 2x2 window, blockIdx.y = (image, window row), so the index arithmetic is one multiply-high per thread (the general kernel
// below spends ~500 vector instructions per thread on 64-bit divisions and builds float stand-ins of the activation from the code:
// it runs at the vector ALU's pace, 4.6 TB/s alone on the chip and half of that beside a weight-gradient kernel). Same arithmetic:
// dz[k] = (bit k of the code) ? dskip[k] + (k == first maximum ? dpool : 0) : 0, the sum in fp32, rounded once.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_pool_skip_relu_bwd_code(const unsigned char* __restrict__ code, const bf16_t* __restrict__ dpool,
                                                                 const bf16_t* __restrict__ dskip, bf16_t* __restrict__ dz, int N, int H, int W,
                                                                 int C, int Hs, int Ws, unsigned ncg, unsigned ncg_magic, float keep, unsigned key) {
    const int Hp = H >> 1, Wp = W >> 1;
    const unsigned x = blockIdx.x * 256u + threadIdx.x;
    if (x >= (unsigned)Wp * ncg) return;
    unsigned wx = __umulhi(x, ncg_magic), cg = x - wx * ncg;   // floor(2^32 / ncg) may fall one short
    if (cg >= ncg) { ++wx; cg -= ncg; }
    const int oy0 = (H - Hs) / 2, ox0 = (W - Ws) / 2;
  for (int row = (int)blockIdx.y; row < N * Hp; row += (int)gridDim.y) {
    const int n = row / Hp, wy = row - n * Hp;
    const unsigned pidx = (unsigned)(((n * Hp + wy) * Wp + (int)wx) * C) + cg * 8;
    const uint2 c2 = *(const uint2*)(code + pidx);
    const u32x4 dp4 = *(const u32x4*)(dpool + pidx);
    u32x4 g4[4];
    bool have[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int sy = 2 * wy + (k >> 1) - oy0, sx = 2 * (int)wx + (k & 1) - ox0;
        have[k] = dskip && (unsigned)sy < (unsigned)Hs && (unsigned)sx < (unsigned)Ws;
        g4[k] = have[k] ? *(const u32x4*)(dskip + ((long)(n * Hs + sy) * Ws + sx) * C + cg * 8) : u32x4{0u, 0u, 0u, 0u};
    }
    float dp[8];
    unpack8(dp4, dp);
    if (keep < 1.f) {  // the pooled tensor went through dropout: same mask, same scale
        const float inv = 1.f / keep;
#pragma unroll
        for (int i = 0; i < 8; ++i) dp[i] *= drop_keep(key, pidx + i, keep) * inv;
    }
    unsigned best[8], bits[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const unsigned byte = ((i < 4 ? c2.x : c2.y) >> (8 * (i & 3))) & 0xffu;
        best[i] = byte >> 4;
        bits[i] = byte;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        float g[8];
        unpack8(g4[k], g);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float v = g[i] + (best[i] == (unsigned)k ? dp[i] : 0.f);
            g[i] = (bits[i] >> k) & 1u ? v : 0.f;
        }
        *(u32x4*)(dz + ((long)(n * H + 2 * wy + (k >> 1)) * W + 2 * (int)wx + (k & 1)) * C + cg * 8) = pack8(g);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// gradient junction at an encoder output: MaxPoolGrad + zero-padded skip gradient + ReluGrad.
// One thread = 8 channels of one 2x2 window.
// ---------------------------------------------------------------------------------------------
__global__ void k_pool_skip_relu_bwd(const bf16_t* __restrict__ yact, const unsigned char* __restrict__ code, const bf16_t* __restrict__ dpool,
                                     const bf16_t* __restrict__ dskip, bf16_t* __restrict__ dz, int N, int H, int W, int C,
                                     int Hs, int Ws, float keep, unsigned key) {
    const float inv = keep < 1.f ? 1.f / keep : 1.f;
    const int Hw = (H + 1) >> 1, Ww = (W + 1) >> 1, ncg = C >> 3;
    const int Hp = H >> 1, Wp = W >> 1;
    const int oy0 = (H - Hs) / 2, ox0 = (W - Ws) / 2;
    const long total = (long)N * Hw * Ww * ncg;
    for (long tid = (long)blockIdx.x * blockDim.x + threadIdx.x; tid < total; tid += (long)gridDim.x * blockDim.x) {
        const int cg = (int)(tid % ncg);
        long r = tid / ncg;
        const int wx = (int)(r % Ww);
        r /= Ww;
        const int wy = (int)(r % Hw), n = (int)(r / Hw);
        float v[4][8], g[4][8];
        bool inb[4];
        // with a code tensor (H, W even: every window is whole) the activation is not read: v[k] = 1 where it was > 0, and 2 at the
        // window's first maximum -- the same comparisons below then pick the same elements
        unsigned cw[2] = {0u, 0u};
        if (code) {
            const uint2 c2 = *(const uint2*)(code + ((long)(n * Hp + wy) * Wp + wx) * C + cg * 8);
            cw[0] = c2.x;
            cw[1] = c2.y;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int yy = 2 * wy + (k >> 1), xx = 2 * wx + (k & 1);
            inb[k] = (yy < H) && (xx < W);
            if (code) {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const unsigned byte = (cw[i >> 2] >> (8 * (i & 3))) & 0xffu;
                    v[k][i] = ((byte >> k) & 1u) ? ((int)(byte >> 4) == k ? 2.f : 1.f) : ((int)(byte >> 4) == k ? 0.f : -1.f);
                }
            } else if (inb[k]) {
                unpack8(*(const u32x4*)(yact + ((long)(n * H + yy) * W + xx) * C + cg * 8), v[k]);
            } else {
#pragma unroll
                for (int i = 0; i < 8; ++i) v[k][i] = 0.f;
            }
            const int sy = yy - oy0, sx = xx - ox0;
            if (dskip && inb[k] && sy >= 0 && sy < Hs && sx >= 0 && sx < Ws) {
                unpack8(*(const u32x4*)(dskip + ((long)(n * Hs + sy) * Ws + sx) * C + cg * 8), g[k]);
            } else {
#pragma unroll
                for (int i = 0; i < 8; ++i) g[k][i] = 0.f;
            }
        }
        if (dpool && wy < Hp && wx < Wp) {
            float dp[8];
            const long pidx = ((long)(n * Hp + wy) * Wp + wx) * C + cg * 8;
            unpack8(*(const u32x4*)(dpool + pidx), dp);
            if (keep < 1.f) {  // the pooled tensor went through dropout: same mask, same scale
#pragma unroll
                for (int i = 0; i < 8; ++i) dp[i] *= drop_keep(key, (unsigned)(pidx + i), keep) * inv;
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                int best = 0;
                float m = v[0][i];
                if (v[1][i] > m) { m = v[1][i]; best = 1; }
                if (v[2][i] > m) { m = v[2][i]; best = 2; }
                if (v[3][i] > m) { m = v[3][i]; best = 3; }
#pragma unroll
                for (int k = 0; k < 4; ++k) g[k][i] += (best == k) ? dp[i] : 0.f;
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (!inb[k]) continue;
            const int yy = 2 * wy + (k >> 1), xx = 2 * wx + (k & 1);
#pragma unroll
            for (int i = 0; i < 8; ++i) g[k][i] = v[k][i] > 0.f ? g[k][i] : 0.f;
            *(u32x4*)(dz + ((long)(n * H + yy) * W + xx) * C + cg * 8) = pack8(g[k]);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// BiasAddGrad: per-channel sums over pixels, two deterministic stages
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_colsum_partial(const bf16_t* __restrict__ dz, float* __restrict__ partial, long npix, int C) {
    __shared__ float red[256 * 8];
    const int ncg = C >> 3;
    const int lanes_p = 256 / ncg;  // pixel lanes per block (ncg divides 256)
    const int cg = threadIdx.x % ncg, pl = threadIdx.x / ncg;
    float s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (pl < lanes_p) {
        for (long p = (long)blockIdx.x * lanes_p + pl; p < npix; p += (long)gridDim.x * lanes_p) {
            float f[8];
            unpack8(*(const u32x4*)(dz + p * C + cg * 8), f);
#pragma unroll
            for (int i = 0; i < 8; ++i) s[i] += f[i];
        }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) red[threadIdx.x * 8 + i] = s[i];
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {
        const int g = c >> 3, i = c & 7;
        float t = 0.f;
        for (int q = 0; q < lanes_p; ++q) t += red[(q * ncg + g) * 8 + i];
        partial[(long)blockIdx.x * C + c] = t;
    }
}
// second stage: 32 channels x 8 partial-lanes per block, LDS tree over the lanes (fixed order: deterministic)
__global__ void __launch_bounds__(256) k_colsum_final(const float* __restrict__ partial, float* __restrict__ out, int nblk, int C) {
    __shared__ float red[256];
    const int cl = threadIdx.x & 31, part = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + cl;
    float t = 0.f;
    if (c < C)
        for (int b = part; b < nblk; b += 8) t += partial[(long)b * C + c];
    red[threadIdx.x] = t;
    __syncthreads();
    if (part == 0 && c < C) {
        float r = 0.f;
#pragma unroll
        for (int q = 0; q < 8; ++q) r += red[q * 32 + cl];
        out[c] = r;
    }
}

// ---------------------------------------------------------------------------------------------
// split-K slab reduction: out[row][col] = sum_z slab[z][row][col] for the rows of one source
// rows are (tap, cs) pairs: row index -> tap*CsOut + cs_off + cs
// ---------------------------------------------------------------------------------------------
// an optional extra row behind the taps of every slab (column sums of F = bias gradient) is reduced into out2 by the same launch
struct ReduceItem { long e; float* dst; };
__device__ __forceinline__ bool reduce_item(ReduceItem& it, long tid, long total, float* out, float* out2, int n2, int ntap, int CsOut,
                                            int cs_off, int cs_cnt, int CfOut) {
    const int cf4 = CfOut >> 2;
    if (tid < total) {
        const int c4 = (int)(tid % cf4);
        long r = tid / cf4;
        const int cs = (int)(r % cs_cnt), tap = (int)(r / cs_cnt);
        it.e = ((long)tap * CsOut + cs_off + cs) * CfOut + c4 * 4;
        it.dst = out + it.e;
        return true;
    }
    if (out2 && tid < total + n2) {
        const int c4 = (int)(tid - total);
        it.e = (long)ntap * CsOut * CfOut + c4 * 4;
        it.dst = out2 + c4 * 4;
        return true;
    }
    return false;
}
__global__ void k_reduce_slabs(const float* __restrict__ slab, float* __restrict__ out, float* __restrict__ out2, int n2, int nsplit,
                               long slab_elems, int ntap, int CsOut, int cs_off, int cs_cnt, int CfOut) {
    const long total = (long)ntap * cs_cnt * (CfOut >> 2);
    const long total2 = total + (out2 ? n2 : 0);
    for (long tid = (long)blockIdx.x * blockDim.x + threadIdx.x; tid < total2; tid += (long)gridDim.x * blockDim.x) {
        ReduceItem it;
        if (!reduce_item(it, tid, total, out, out2, n2, ntap, CsOut, cs_off, cs_cnt, CfOut)) continue;
        f32x4 t = {0.f, 0.f, 0.f, 0.f};
        for (int z = 0; z < nsplit; ++z) t += *(const f32x4*)(slab + z * slab_elems + it.e);
        *(f32x4*)it.dst = t;
    }
}
// many splits, few outputs: 32 float4 outputs x 8 split-lanes per block, fixed-order LDS tree
__global__ void __launch_bounds__(256) k_reduce_slabs_wide(const float* __restrict__ slab, float* __restrict__ out, float* __restrict__ out2,
                                                           int n2, int nsplit, long slab_elems, int ntap, int CsOut, int cs_off, int cs_cnt,
                                                           int CfOut) {
    __shared__ f32x4 red[256];
    const long total = (long)ntap * cs_cnt * (CfOut >> 2);
    const int el = threadIdx.x & 31, zp = threadIdx.x >> 5;
    const long tid = (long)blockIdx.x * 32 + el;
    f32x4 t = {0.f, 0.f, 0.f, 0.f};
    ReduceItem it;
    const bool have = reduce_item(it, tid, total, out, out2, n2, ntap, CsOut, cs_off, cs_cnt, CfOut);
    if (have)
        for (int z = zp; z < nsplit; z += 8) t += *(const f32x4*)(slab + z * slab_elems + it.e);
    red[threadIdx.x] = t;
    __syncthreads();
    if (zp == 0 && have) {
        f32x4 r = red[el];
#pragma unroll
        for (int q = 1; q < 8; ++q) r += red[q * 32 + el];
        *(f32x4*)it.dst = r;
    }
}

// the slab reductions of a whole group of weight-gradient launches (igemm_wg_group) in ONE launch: block b serves job r with
// block_begin[r] <= b < block_begin[r+1]; per job either form above (wide: 32 outputs x 8 split lanes per block), same summation
// orders as the single launches
__global__ void __launch_bounds__(256) k_reduce_slabs_many(const ReduceJob* __restrict__ jobs, int njobs) {
    __shared__ f32x4 red[256];
    // (njobs <= 16 <= 64 lanes: every lane tests one job, a ballot counts those that begin at or before this block -- one L2 round trip
    // instead of up to 16 dependent ones)
    const int lj = threadIdx.x & 63;
    const bool mine = lj < njobs && jobs[lj].block_begin <= (int)blockIdx.x;
    const int r = __popcll(__ballot(mine)) - 1;
    const ReduceJob J = jobs[r];
    const int b = blockIdx.x - J.block_begin;
    const long total = (long)J.ntap * J.cs_cnt * (J.CfOut >> 2);
    ReduceItem it;
    if (J.wide) {
        const int el = threadIdx.x & 31, zp = threadIdx.x >> 5;
        const long tid = (long)b * 32 + el;
        f32x4 t = {0.f, 0.f, 0.f, 0.f};
        const bool have = reduce_item(it, tid, total, J.out, J.out2, J.n2, J.ntap, J.CsOut, J.cs_off, J.cs_cnt, J.CfOut);
        if (have)
            for (int z = zp; z < J.nsplit; z += 8) t += *(const f32x4*)(J.slab + z * J.slab_elems + it.e);
        red[threadIdx.x] = t;
        __syncthreads();
        if (zp == 0 && have) {
            f32x4 q = red[el];
#pragma unroll
            for (int k = 1; k < 8; ++k) q += red[k * 32 + el];
            *(f32x4*)it.dst = q;
        }
    } else {
        const long tid = (long)b * 256 + threadIdx.x;
        if (!reduce_item(it, tid, total, J.out, J.out2, J.n2, J.ntap, J.CsOut, J.cs_off, J.cs_cnt, J.CfOut)) return;
        f32x4 t = {0.f, 0.f, 0.f, 0.f};
        for (int z = 0; z < J.nsplit; ++z) t += *(const f32x4*)(J.slab + z * J.slab_elems + it.e);
        *(f32x4*)it.dst = t;
    }
}

// ---------------------------------------------------------------------------------------------
// head: 1x1 conv (C -> 2) + softmax[...,1] (+ mean CE loss and its backward)
// LP = C/8 lanes per pixel, partial logits reduced with wave shuffles
// ---------------------------------------------------------------------------------------------
template <bool TRAIN>
__global__ void __launch_bounds__(256) k_head(const bf16_t* __restrict__ act, const float* __restrict__ w, const float* __restrict__ b,
                                              const int64_t* __restrict__ labels, float* __restrict__ prob, float* __restrict__ logits,
                                              bf16_t* __restrict__ dact, float* __restrict__ partial, long npix, int C, float inv_count) {
    const int LP = C >> 3;
    const int sub = threadIdx.x % LP;
    const int ppb = 256 / LP;  // pixels per block iteration
    float w0[8], w1[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        w0[i] = w[(sub * 8 + i) * 2];
        w1[i] = w[(sub * 8 + i) * 2 + 1];
    }
    const float b0 = b[0], b1 = b[1];
    float gw0[8], gw1[8], gb0 = 0.f, gb1 = 0.f, lsum = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) gw0[i] = gw1[i] = 0.f;
    const long niter = (npix + (long)gridDim.x * ppb - 1) / ((long)gridDim.x * ppb);
    // HU pixels per thread and trip, all loads requested before any is used: the HBM latency (~2 us x 6 TB/s = 47 KB per CU in flight) wants more
    // than the 16-32 KB that one or two 16-byte loads per lane of 16-20 resident waves give (round 5: two, 3.87 TB/s; round 6: four). The sums keep
    // the order of the pixels: same bits.
    constexpr int HU = 4;
    for (long it0 = 0; it0 < niter; it0 += HU) {
        long pp[HU];
        bool okk[HU];
        u32x4 raw[HU];
        int labv[HU];
#pragma unroll
        for (int u = 0; u < HU; ++u) {
            labv[u] = 0;
            pp[u] = ((it0 + u) * gridDim.x + blockIdx.x) * ppb + threadIdx.x / LP;
            okk[u] = (it0 + u < niter) && pp[u] < npix;
            raw[u] = okk[u] ? *(const u32x4*)(act + pp[u] * C + sub * 8) : u32x4{0u, 0u, 0u, 0u};
            if (TRAIN) labv[u] = okk[u] ? (int)labels[pp[u]] : 0;
        }
#pragma unroll
        for (int u = 0; u < HU; ++u) {
            const long p = pp[u];
            const bool ok = okk[u];
            float a[8];
            unpack8(raw[u], a);
            float l0 = 0.f, l1 = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                l0 = fmaf(a[i], w0[i], l0);
                l1 = fmaf(a[i], w1[i], l1);
            }
            for (int o = 1; o < LP; o <<= 1) {
                l0 += __shfl_xor(l0, o);
                l1 += __shfl_xor(l1, o);
            }
            l0 += b0;
            l1 += b1;
            const float m = fmaxf(l0, l1);
            const float e0 = expf(l0 - m), e1 = expf(l1 - m);
            const float s = e0 + e1;
            const float p1 = e1 / s, p0 = e0 / s;
            if (ok && sub == 0) {
                prob[p] = p1;
                if (logits) {
                    logits[2 * p] = l0;
                    logits[2 * p + 1] = l1;
                }
            }
            if (TRAIN) {
                if (ok) {
                    const int lab = labv[u];
                    const float d0 = (p0 - (lab == 0 ? 1.f : 0.f)) * inv_count;
                    const float d1 = (p1 - (lab == 1 ? 1.f : 0.f)) * inv_count;
                    float da[8];
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        gw0[i] = fmaf(a[i], d0, gw0[i]);
                        gw1[i] = fmaf(a[i], d1, gw1[i]);
                        da[i] = a[i] > 0.f ? fmaf(d0, w0[i], d1 * w1[i]) : 0.f;
                    }
                    *(u32x4*)(dact + p * C + sub * 8) = pack8(da);
                    if (sub == 0) {
                        gb0 += d0;
                        gb1 += d1;
                        lsum += -((lab ? l1 : l0) - m - logf(s));
                    }
                }
            }
        }
    }
    if (TRAIN) {
        // block reduce: threads with equal `sub` hold partials for the same 8 channels
        __shared__ float red[256 * 19];
        float* my = red + threadIdx.x * 19;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            my[i] = gw0[i];
            my[8 + i] = gw1[i];
        }
        my[16] = gb0; my[17] = gb1; my[18] = lsum;
        __syncthreads();
        const int nout = 2 * C + 3;  // [C][2] dw, db[2], loss
        for (int o = threadIdx.x; o < nout; o += 256) {
            float t = 0.f;
            if (o < 2 * C) {
                const int c = o >> 1, k = o & 1, sg = c >> 3, i = c & 7;
                for (int q = 0; q < ppb; ++q) t += red[(q * LP + sg) * 19 + k * 8 + i];
            } else {
                const int j = 16 + (o - 2 * C);
                for (int q = 0; q < ppb; ++q) t += red[(q * LP) * 19 + j];
            }
            partial[(long)blockIdx.x * nout + o] = t;
        }
    }
}
template __global__ void k_head<true>(const bf16_t*, const float*, const float*, const int64_t*, float*, float*, bf16_t*, float*, long, int, float);
template __global__ void k_head<false>(const bf16_t*, const float*, const float*, const int64_t*, float*, float*, bf16_t*, float*, long, int, float);

__global__ void __launch_bounds__(256) k_head_final(const float* __restrict__ partial, float* __restrict__ dw, float* __restrict__ db,
                                                    float* __restrict__ loss_sum, int nblk, int C) {
    // one wave per output (four per workgroup): lane l sums partials l, l + 64, ... (independent loads), then a fixed-order butterfly;
    // 2 C + 3 outputs of ~1000 partials each -- with 32 lanes per output in 17 workgroups the kernel was a chain of ~32 dependent
    // L2 round trips (11 us)
    const int nout = 2 * C + 3;
    const int o = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (o >= nout) return;
    float t = 0.f;
    for (int bk = lane; bk < nblk; bk += 64) t += partial[(long)bk * nout + o];
    for (int m = 32; m > 0; m >>= 1) t += __shfl_xor(t, m);
    if (lane == 0) {
        if (o < 2 * C) dw[o] = t;
        else if (o < 2 * C + 2) db[o - 2 * C] = t;
        else loss_sum[0] += t;
    }
}

// color_space_adjust gradients from the first conv's scatter buffer (include/rsu.h, rsu_color_adjust_bwd): 12 outputs, each a
// 9 x Cout sum in a fixed order (lane-strided partials, then an LDS tree): one block of 12 x 64 threads
__global__ void __launch_bounds__(768) k_color_adjust_bwd(const float* __restrict__ gx, const float* __restrict__ w1, float* __restrict__ dW0,
                                                           float* __restrict__ db0, int Cout, float scale, int accumulate) {
    __shared__ float red[12][64];
    const int o = threadIdx.x >> 6, l = threadIdx.x & 63;  // o < 9: dW0[ci][cj] with o = 3*ci + cj; o >= 9: db0[cj]
    const int cj = o < 9 ? o % 3 : o - 9;
    float t = 0.f;
    for (int tap = 0; tap < 9; ++tap)
        for (int co = l; co < Cout; co += 64) t += w1[(tap * 3 + cj) * Cout + co] * gx[(tap * 12 + o) * Cout + co];
    red[o][l] = t;
    __syncthreads();
    if (l == 0) {
        float r = 0.f;
        for (int q = 0; q < 64; ++q) r += red[o][q];
        r *= scale;
        float* dst = o < 9 ? dW0 + o : db0 + (o - 9);
        *dst = accumulate ? *dst + r : r;
    }
}

// ---------------------------------------------------------------------------------------------
// momentum SGD (tf_aerial_images.py:116-121)
// ---------------------------------------------------------------------------------------------
__global__ void k_momentum(float* __restrict__ w, float* __restrict__ acc, const float* __restrict__ g, float lr, float mu, float gscale, long n) {
    const long n4 = n >> 2;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        f32x4 a = ((f32x4*)acc)[i], gv = ((const f32x4*)g)[i], wv = ((f32x4*)w)[i];
        a = mu * a + gscale * gv;
        wv -= lr * a;
        ((f32x4*)acc)[i] = a;
        ((f32x4*)w)[i] = wv;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const long i = (n4 << 2) + threadIdx.x;
        const float a = mu * acc[i] + gscale * g[i];
        acc[i] = a;
        w[i] -= lr * a;
    }
}

// ---------------------------------------------------------------------------------------------
// weight packing into MFMA fragment order: packed[chunk][tap][tile][lane][8]
//   element (chunk, tap, tile T, lane, j): row rho = lane&15 of tile T, k = chunk*32 + 8*(lane>>4) + j
//   tile pair P = T/2, t = T%2: output channel of the row = 32P + 8*(rho>>2) + 4t + (rho&3)
//   k runs over the 32-padded concatenation of the K segments (concat sources)
// value = src[tapmap(tap)*s_tap + row*s_row + kreal*s_k]
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void pack_range(const float* __restrict__ src, bf16_t* __restrict__ dst, const PackParams& pp, long e0, long estride) {
    const long total = (long)pp.nchunks * pp.ntap * pp.ntiles * 512;
    for (long e = e0; e < total; e += estride) {
        const int j = (int)(e & 7), lane = (int)((e >> 3) & 63);
        long r = e >> 9;
        const int T = (int)(r % pp.ntiles);
        r /= pp.ntiles;
        const int tap = (int)(r % pp.ntap), chunk = (int)(r / pp.ntap);
        const int rho = lane & 15;
        const int row = 32 * (T >> 1) + 8 * (rho >> 2) + 4 * (T & 1) + (rho & 3);
        int k = chunk * 32 + 8 * (lane >> 4) + j;
        // locate the K segment
        int kreal = -1, base_pad = 0, base_real = 0;
        for (int s = 0; s < pp.nseg; ++s) {
            const int cpad = (pp.seg_c[s] + 31) & ~31;
            if (k >= base_pad && k < base_pad + cpad) {
                const int c = k - base_pad;
                if (c < pp.seg_c[s]) kreal = base_real + c;
            }
            base_pad += cpad;
            base_real += pp.seg_c[s];
        }
        float v = 0.f;
        if (kreal >= 0 && row < pp.rows) {
            const int tsrc = pp.flip ? (pp.ntap - 1 - tap) : tap;
            v = src[(long)tsrc * pp.s_tap + (long)row * pp.s_row + (long)kreal * pp.s_k];
        }
        dst[e] = f2bf(v);
    }
}
__global__ void k_pack(const float* __restrict__ src, bf16_t* __restrict__ dst, PackParams pp) {
    pack_range(src, dst, pp, (long)blockIdx.x * blockDim.x + threadIdx.x, (long)gridDim.x * blockDim.x);
}
// all weight tensors of the network in ONE launch: each job owns a contiguous range of blocks (PACK_EPB elements per block)
#define PACK_EPB 4096
__global__ void __launch_bounds__(256) k_pack_many(const PackJob* __restrict__ jobs, int njobs) {
    __shared__ int sj;
    if (threadIdx.x == 0) {
        int lo = 0, hi = njobs - 1;
        while (lo < hi) {  // last job whose block_start <= blockIdx.x
            const int mid = (lo + hi + 1) >> 1;
            if (jobs[mid].block_start <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
        }
        sj = lo;
    }
    __syncthreads();
    const PackJob jb = jobs[sj];
    const long total = (long)jb.pp.nchunks * jb.pp.ntap * jb.pp.ntiles * 512;
    const long b0 = (long)(blockIdx.x - jb.block_start) * PACK_EPB;
    long hi = b0 + PACK_EPB;
    if (hi > total) hi = total;
    PackParams pp = jb.pp;
    const unsigned g0 = (unsigned)(b0 >> 3), g1 = (unsigned)((hi + 7) >> 3);
    // rows contiguous in the source (HWIO conv kernels packed for the forward pass, the bulk of the bytes): one thread per FOUR
    // neighbouring groups = 4 consecutive rows: eight 16-byte reads (one per k, full sectors) and four 16-byte stores that form
    // one 64-byte run, instead of eight 4-byte reads per group that use a quarter of every sector they touch
    if (pp.s_row == 1 && (pp.rows & 3) == 0 && (pp.s_k & 3) == 0 && (pp.s_tap & 3) == 0 && ((unsigned long)jb.src & 15) == 0 && (g0 & 3) == 0 &&
        (g1 & 3) == 0) {
        for (unsigned q = (g0 >> 2) + threadIdx.x; q < (g1 >> 2); q += 256) {
            const unsigned gidx = q << 2;
            const int lane = (int)(gidx & 63);
            unsigned r = gidx >> 6;
            const int T = (int)(r % (unsigned)pp.ntiles);
            r /= (unsigned)pp.ntiles;
            const int tap = (int)(r % (unsigned)pp.ntap), chunk = (int)(r / (unsigned)pp.ntap);
            const int rho = lane & 15;  // a multiple of 4
            const int row = 32 * (T >> 1) + 8 * (rho >> 2) + 4 * (T & 1);
            const int k0 = chunk * 32 + 8 * (lane >> 4);
            int kreal0 = 0, nvalid = 0, base_pad = 0, base_real = 0;
            for (int sgi = 0; sgi < pp.nseg; ++sgi) {
                const int cpad = (pp.seg_c[sgi] + 31) & ~31;
                if (k0 >= base_pad && k0 < base_pad + cpad) {
                    const int c = k0 - base_pad;
                    kreal0 = base_real + c;
                    nvalid = pp.seg_c[sgi] - c;
                }
                base_pad += cpad;
                base_real += pp.seg_c[sgi];
            }
            if (row >= pp.rows) nvalid = 0;
            const int tsrc = pp.flip ? (pp.ntap - 1 - tap) : tap;
            const float* sp = jb.src + (long)tsrc * pp.s_tap + row + (long)kreal0 * pp.s_k;
            f32x4 v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = j < nvalid ? *(const f32x4*)(sp + (long)j * pp.s_k) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int r3 = 0; r3 < 4; ++r3) {
                const u32x4 o = {pack_bf2(v[0][r3], v[1][r3]), pack_bf2(v[2][r3], v[3][r3]), pack_bf2(v[4][r3], v[5][r3]),
                                 pack_bf2(v[6][r3], v[7][r3])};
                *(u32x4*)(jb.dst + (long)(gidx + r3) * 8) = o;
            }
        }
        return;
    }
    // general case: one thread per 16-byte group (lane, j = 0..7): 32-bit index math once per group, eight strided reads (two 16-byte
    // reads when the k are contiguous in the source), one 16-byte store
    for (unsigned gidx = g0 + threadIdx.x; gidx < g1; gidx += 256) {
        const int lane = (int)(gidx & 63);
        unsigned r = gidx >> 6;
        const int T = (int)(r % (unsigned)pp.ntiles);
        r /= (unsigned)pp.ntiles;
        const int tap = (int)(r % (unsigned)pp.ntap), chunk = (int)(r / (unsigned)pp.ntap);
        const int rho = lane & 15;
        const int row = 32 * (T >> 1) + 8 * (rho >> 2) + 4 * (T & 1) + (rho & 3);
        const int k0 = chunk * 32 + 8 * (lane >> 4);  // the 8 k of a group lie in one (32-padded) segment
        int kreal0 = 0, nvalid = 0, base_pad = 0, base_real = 0;
        for (int sgi = 0; sgi < pp.nseg; ++sgi) {
            const int cpad = (pp.seg_c[sgi] + 31) & ~31;
            if (k0 >= base_pad && k0 < base_pad + cpad) {
                const int c = k0 - base_pad;
                kreal0 = base_real + c;
                nvalid = pp.seg_c[sgi] - c;
            }
            base_pad += cpad;
            base_real += pp.seg_c[sgi];
        }
        if (row >= pp.rows) nvalid = 0;
        const int tsrc = pp.flip ? (pp.ntap - 1 - tap) : tap;
        const float* sp = jb.src + (long)tsrc * pp.s_tap + (long)row * pp.s_row + (long)kreal0 * pp.s_k;
        float v[8];
        if (pp.s_k == 1 && nvalid >= 8 && ((unsigned long)sp & 15) == 0) {
            const f32x4 lo = *(const f32x4*)sp, hi4 = *(const f32x4*)(sp + 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                v[j] = lo[j];
                v[4 + j] = hi4[j];
            }
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = j < nvalid ? sp[(long)j * pp.s_k] : 0.f;
        }
        const u32x4 o = {pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]), pack_bf2(v[4], v[5]), pack_bf2(v[6], v[7])};
        *(u32x4*)(jb.dst + (long)gidx * 8) = o;
    }
}

// ---------------------------------------------------------------------------------------------
// tiler: mirror_border + extract_patches fused (images.py:269-281,35-85), overlap-add (images.py:131-164)
// ---------------------------------------------------------------------------------------------
__global__ void k_extract_tiles(const float* __restrict__ imgs, float* __restrict__ tiles, int H, int S, int P, int stride,
                                int pps, long t0, long ntiles) {
    const int off = (S - P) / 2;
    const long total = ntiles * S * S;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int x = (int)(e % S);
        long r = e / S;
        const int yy = (int)(r % S);
        const long tl = r / S;
        const long tg = t0 + tl;
        const int img = (int)(tg / (pps * pps)), tt = (int)(tg % (pps * pps));
        const int x0 = (tt / pps) * stride, y0 = (tt % pps) * stride;  // x outer, y inner
        int sy = y0 + yy - off, sx = x0 + x - off;                       // coordinates in the un-padded image
        sy = sy < 0 ? -sy - 1 : (sy >= H ? 2 * H - 1 - sy : sy);         // np.pad 'symmetric'
        sx = sx < 0 ? -sx - 1 : (sx >= H ? 2 * H - 1 - sx : sx);
        const float* s = imgs + ((long)(img * H + sy) * H + sx) * 3;
        float* d = tiles + e * 3;
        d[0] = s[0]; d[1] = s[1]; d[2] = s[2];
    }
}

// gather form (deterministic, no atomics): one thread per output pixel sums, in increasing tile index, the tiles of
// [t0, t0+ntiles) that cover it; the in-call sum is kept in double and added to the float accumulator once.
__global__ void k_overlap_add(const float* __restrict__ prob, float* __restrict__ acc, float* __restrict__ hits, int nimg, int H, int P,
                              int stride, int pps, long t0, long ntiles) {
    const long total = (long)nimg * H * H;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int x = (int)(e % H);
        long r = e / H;
        const int y = (int)(r % H), img = (int)(r / H);
        int xi_lo = (x - P + stride) / stride; if (x - P + 1 <= 0) xi_lo = 0;
        int yi_lo = (y - P + stride) / stride; if (y - P + 1 <= 0) yi_lo = 0;
        int xi_hi = x / stride; if (xi_hi > pps - 1) xi_hi = pps - 1;
        int yi_hi = y / stride; if (yi_hi > pps - 1) yi_hi = pps - 1;
        double s = 0.0;
        int cnt = 0;
        for (int xi = xi_lo; xi <= xi_hi; ++xi)
            for (int yi = yi_lo; yi <= yi_hi; ++yi) {
                const long t = (long)img * pps * pps + (long)xi * pps + yi;
                if (t < t0 || t >= t0 + ntiles) continue;
                s += (double)prob[((t - t0) * P + (y - yi * stride)) * P + (x - xi * stride)];
                ++cnt;
            }
        if (cnt) {
            acc[e] += (float)s;
            hits[e] += (float)cnt;
        }
    }
}

__global__ void k_overlap_finish(const float* __restrict__ acc, const float* __restrict__ hits, float* __restrict__ out, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) out[i] = acc[i] / hits[i];
}

// ---------------------------------------------------------------------------------------------
// host-side launchers
// ---------------------------------------------------------------------------------------------
static inline int grid_for(long total, int block, int cap = 256 * 16) {
    long g = (total + block - 1) / block;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (int)g;
}

hipError_t ew_color_adjust(const float* x, const float* w, const float* b, void* out16, long npix, float keep, unsigned key, hipStream_t st) {
    hipLaunchKernelGGL(k_color_adjust, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, st, x, w, b, (bf16_t*)out16, npix, keep, key);
    return hipGetLastError();
}
__global__ void k_scale_bf16(bf16_t* __restrict__ x, long n8, float s) {
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < n8; t += (long)gridDim.x * blockDim.x) {
        float v[8];
        unpack8(*(const u32x4*)(x + 8 * t), v);
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] *= s;
        *(u32x4*)(x + 8 * t) = pack8(v);
    }
}
hipError_t ew_scale_bf16(void* x, long n, float s, hipStream_t st) {
    hipLaunchKernelGGL(k_scale_bf16, dim3(grid_for(n / 8, 256)), dim3(256), 0, st, (bf16_t*)x, n / 8, s);
    return hipGetLastError();
}
hipError_t ew_dropout(const void* x, void* y, long n, float keep, unsigned key, hipStream_t st) {
    hipLaunchKernelGGL(k_dropout, dim3(grid_for(n / 8, 256)), dim3(256), 0, st, (const bf16_t*)x, (bf16_t*)y, n / 8, keep, key);
    return hipGetLastError();
}
hipError_t ew_scatter_first_grads(const float* tmp, float* dw1, float* gxc, int Cout, hipStream_t st) {
    hipLaunchKernelGGL(k_scatter_first_grads, dim3((9 * 12 * Cout + 255) / 256), dim3(256), 0, st, tmp, dw1, gxc, Cout);
    return hipGetLastError();
}
hipError_t ew_maxpool_fwd(const void* x, void* y, void* code, int N, int H, int W, int C, float keep, unsigned key, hipStream_t st) {
    const long total = (long)N * (H / 2) * (W / 2) * (C / 8);
    hipLaunchKernelGGL(k_maxpool_fwd, dim3(grid_for(total, 256)), dim3(256), 0, st, (const bf16_t*)x, (bf16_t*)y, (unsigned char*)code, N, H, W, C, keep,
                       key);
    return hipGetLastError();
}
hipError_t ew_pool_skip_relu_bwd(const void* yact, const void* code, const void* dpool, const void* dskip, void* dz, int N, int H, int W, int C,
                                 int Hs, int Ws, float keep, unsigned key, hipStream_t st) {
    if (code && dpool && !((H | W) & 1) && (long)N * H * W * C < (1l << 31)) {   // whole windows, a code byte per pooled element: the lean kernel
        const unsigned ncg = (unsigned)(C / 8), per_row = (unsigned)(W / 2) * ncg;
        const long rows = (long)N * (H / 2);
        hipLaunchKernelGGL(k_pool_skip_relu_bwd_code, dim3((per_row + 255) / 256, (unsigned)(rows < 32768 ? rows : 32768)), dim3(256), 0, st,
                           (const unsigned char*)code, (const bf16_t*)dpool, (const bf16_t*)dskip, (bf16_t*)dz, N, H, W, C, dskip ? Hs : 0, dskip ? Ws : 0, ncg,
                           ncg == 1 ? 0xffffffffu : (unsigned)(0x100000000ull / ncg), keep, key);
        return hipGetLastError();
    }
    const long total = (long)N * ((H + 1) / 2) * ((W + 1) / 2) * (C / 8);
    hipLaunchKernelGGL(k_pool_skip_relu_bwd, dim3(grid_for(total, 256)), dim3(256), 0, st, (const bf16_t*)yact, (const unsigned char*)code,
                       (const bf16_t*)dpool,
                       (const bf16_t*)dskip, (bf16_t*)dz, N, H, W, C, Hs, Ws, keep, key);
    return hipGetLastError();
}
int ew_colsum_blocks(long npix, int C) {
    const int lanes_p = 256 / (C / 8);
    long nb = (npix + (long)lanes_p * 64 - 1) / ((long)lanes_p * 64);
    if (nb > 256) nb = 256;
    if (nb < 1) nb = 1;
    return (int)nb;
}
hipError_t ew_colsum(const void* dz, float* db, float* ws, long npix, int C, hipStream_t st) {
    const int nb = ew_colsum_blocks(npix, C);
    hipLaunchKernelGGL(k_colsum_partial, dim3(nb), dim3(256), 0, st, (const bf16_t*)dz, ws, npix, C);
    hipLaunchKernelGGL(k_colsum_final, dim3((C + 31) / 32), dim3(256), 0, st, ws, db, nb, C);
    return hipGetLastError();
}
hipError_t ew_reduce_slabs(const float* slab, float* out, float* out2, int n2, int nsplit, long slab_elems, int ntap, int CsOut, int cs_off,
                           int cs_cnt, int CfOut, hipStream_t st) {
    const long total = (long)ntap * cs_cnt * (CfOut / 4) + (out2 ? n2 : 0);
    if (nsplit >= 16 && (total + 31) / 32 < 0x7fffffffL)
        hipLaunchKernelGGL(k_reduce_slabs_wide, dim3((unsigned)((total + 31) / 32)), dim3(256), 0, st, slab, out, out2, n2, nsplit, slab_elems, ntap,
                           CsOut, cs_off, cs_cnt, CfOut);
    else
        hipLaunchKernelGGL(k_reduce_slabs, dim3(grid_for(total, 256)), dim3(256), 0, st, slab, out, out2, n2, nsplit, slab_elems, ntap, CsOut,
                           cs_off, cs_cnt, CfOut);
    return hipGetLastError();
}
int ew_reduce_job_blocks(ReduceJob& j) {
    const long total = (long)j.ntap * j.cs_cnt * (j.CfOut / 4) + (j.out2 ? j.n2 : 0);
    j.wide = j.nsplit >= 16 ? 1 : 0;
    return (int)(j.wide ? (total + 31) / 32 : (total + 255) / 256);
}
hipError_t ew_reduce_slabs_many(const ReduceJob* jobs_dev, int njobs, int total_blocks, hipStream_t st) {
    hipLaunchKernelGGL(k_reduce_slabs_many, dim3(total_blocks), dim3(256), 0, st, jobs_dev, njobs);
    return hipGetLastError();
}
int ew_head_blocks(long npix, int C) {
    const int ppb = 256 / (C / 8);
    long nb = (npix + (long)ppb * 8 - 1) / ((long)ppb * 8);
    if (nb > 1024) nb = 1024;
    if (nb < 1) nb = 1;
    return (int)nb;
}
hipError_t ew_head(bool train, const void* act, const float* w, const float* b, const int64_t* labels, float* prob, float* logits, void* dact,
                   float* dw, float* db, float* loss_sum, float* ws, long npix, int C, float inv_count, hipStream_t st) {
    const int nb = ew_head_blocks(npix, C);
    if (train) {
        hipLaunchKernelGGL(k_head<true>, dim3(nb), dim3(256), 0, st, (const bf16_t*)act, w, b, labels, prob, logits, (bf16_t*)dact, ws, npix, C, inv_count);
        hipLaunchKernelGGL(k_head_final, dim3((2 * C + 3 + 3) / 4), dim3(256), 0, st, ws, dw, db, loss_sum, nb, C);
    } else {
        hipLaunchKernelGGL(k_head<false>, dim3(nb), dim3(256), 0, st, (const bf16_t*)act, w, b, nullptr, prob, logits, nullptr, nullptr, npix, C, 0.f);
    }
    return hipGetLastError();
}
hipError_t ew_color_adjust_bwd(const float* gx, const float* w1, float* dW0, float* db0, int Cout, float scale, int accumulate, hipStream_t st) {
    hipLaunchKernelGGL(k_color_adjust_bwd, dim3(1), dim3(768), 0, st, gx, w1, dW0, db0, Cout, scale, accumulate);
    return hipGetLastError();
}
hipError_t ew_momentum(float* w, float* acc, const float* g, float lr, float mu, float gscale, long n, hipStream_t st) {
    hipLaunchKernelGGL(k_momentum, dim3(grid_for((n + 3) / 4, 256)), dim3(256), 0, st, w, acc, g, lr, mu, gscale, n);
    return hipGetLastError();
}
hipError_t ew_pack(const float* src, void* dst, const PackParams& pp, hipStream_t st) {
    const long total = (long)pp.nchunks * pp.ntap * pp.ntiles * 512;
    hipLaunchKernelGGL(k_pack, dim3(grid_for(total, 256)), dim3(256), 0, st, src, (bf16_t*)dst, pp);
    return hipGetLastError();
}
int ew_pack_blocks(const PackParams& pp) {
    const long total = (long)pp.nchunks * pp.ntap * pp.ntiles * 512;
    return (int)((total + PACK_EPB - 1) / PACK_EPB);
}
hipError_t ew_pack_many(const PackJob* jobs_dev, int njobs, int total_blocks, hipStream_t st) {
    hipLaunchKernelGGL(k_pack_many, dim3(total_blocks), dim3(256), 0, st, jobs_dev, njobs);
    return hipGetLastError();
}
// ---------------------------------------------------------------------------------------------
// Momentum step + re-pack in ONE pass over the parameters (k_update_pack_many). k_momentum moves 20 B per parameter and k_pack_many
// reads every conv kernel twice more (once per packed layout, 12 B per weight, at half the copy rate: its gathers use a quarter of
// a sector for one of the two layouts). Here a workgroup owns 32 x 128 source elements of one tap of one tensor -- four 32 x 32
// blocks: it reads w, acc, g once (full 128-byte rows), updates them, keeps the new weights in LDS and writes BOTH packed layouts
// from there, each block as one contiguous 2-KiB run of 16-byte pieces: 24 B per weight, every access a full sector. Small
// variables that no MFMA kernel reads (biases, colour adjust, the 1x1 head) are plain ranges of the same launch.
// Source tensor = [ntap][R1][R2] float32, R2 contiguous (conv kernels HWIO: R1 = ci, R2 = co; transposed-conv kernels
// [a][b][co][ci]: R1 = co, R2 = ci). A packed layout is [chunk of 32 k][tap][16-row tile][lane][8] (pack_range above); a
// destination takes its ROWS from R2 and k from R1 ("orientation A": conv forward, transposed-conv backward) or rows from R1 and k
// from R2 ("B": conv backward-data, one buffer per concat source; transposed-conv forward, one buffer per tap).
// ---------------------------------------------------------------------------------------------
#define UP_EPB 4096   // floats per workgroup of a plain range
// UP_NT (developer A/B switch): 1 = the gradient is read and the packed layouts are written with non-temporal hints (streamed once per
// step); 2 = w and acc too
#ifndef UP_NT
#define UP_NT 0
#endif
namespace {
__device__ __forceinline__ f32x4 up_ld_stream(const f32x4* p) { return UP_NT >= 1 ? __builtin_nontemporal_load(p) : *p; }
__device__ __forceinline__ f32x4 up_ld_state(const f32x4* p) { return UP_NT >= 2 ? __builtin_nontemporal_load(p) : *p; }
__device__ __forceinline__ void up_st_state(f32x4* p, f32x4 v) { if (UP_NT >= 2) __builtin_nontemporal_store(v, p); else *p = v; }
__device__ __forceinline__ void up_st_stream(u32x4* p, u32x4 v) { if (UP_NT >= 1) __builtin_nontemporal_store(v, p); else *p = v; }
}  // namespace
// One workgroup of the packed-tensor path: block (tap, 32-row block rb, group cg of four 32-column blocks) of tensor J. `grad(i)` returns
// the gradient float4 at float4 index i of the source tensor (k_update_pack_many: the gradient buffer; k_update_pack_seg: the ordered sum
// of the weight-gradient slabs).
template <typename GradFn>
__device__ __forceinline__ void update_pack_block(const UpJob& J, float (*tile)[32][33], int tap, int rb, int cg, float lr, float mu, float gscale,
                                                  GradFn grad) {
    int seg = 0;
    while (seg + 1 < J.nseg && rb >= J.seg_blk0[seg + 1]) ++seg;
    const int rbs = rb - J.seg_blk0[seg];                // 32-row block inside its segment
    const int r0 = J.seg_r0[seg] + rbs * 32;              // first source row
    const int vr = min(32, J.seg_c[seg] - rbs * 32);      // real rows of the block
    const int r = threadIdx.x >> 3, c4 = (threadIdx.x & 7) * 4;
    // every load of the workgroup's four blocks is issued before the first store: w / acc are read and written through the same (non-restrict)
    // pointers, so a store in front of a later block's loads would order them -- four dependent round trips to memory instead of one
    f32x4 av[4], gv[4], wv[4];
    long idx[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int cb = cg * 4 + q, c0 = cb * 32;
        idx[q] = -1;
        av[q] = gv[q] = wv[q] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (cb < J.ncb && r < vr && c0 + c4 < J.R2) {   // (R2 is a multiple of 4: a float4 is inside or outside as a whole)
            const long i = (((long)tap * J.R1 + r0 + r) * J.R2 + c0 + c4) >> 2;
            idx[q] = i;
            av[q] = up_ld_state((const f32x4*)J.acc + i);
            gv[q] = grad(i);
            wv[q] = up_ld_state((const f32x4*)J.w + i);
        }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        if (idx[q] >= 0) {
            av[q] = mu * av[q] + gscale * gv[q];
            wv[q] -= lr * av[q];
            up_st_state((f32x4*)J.acc + idx[q], av[q]);
            up_st_state((f32x4*)J.w + idx[q], wv[q]);
        }
        tile[q][r][c4] = wv[q][0]; tile[q][r][c4 + 1] = wv[q][1]; tile[q][r][c4 + 2] = wv[q][2]; tile[q][r][c4 + 3] = wv[q][3];
    }
    __syncthreads();
    // ---- both packed layouts from LDS: per (column block, destination) 128 pieces of 16 bytes = the 2 tiles of a 32-row pair
    const int npiece = 4 * J.ndest * 128;
    for (int pi = threadIdx.x; pi < npiece; pi += 256) {
        const int p = pi & 127, d = (pi >> 7) % J.ndest, q = pi / (128 * J.ndest);
        const int cb = cg * 4 + q;
        if (cb >= J.ncb) continue;
        const UpDest& D = J.d[d];
        const int tl = p >> 6, lane = p & 63, rho = lane & 15, k0 = 8 * (lane >> 4);
        const int rip = 8 * (rho >> 2) + 4 * tl + (rho & 3);   // row inside the pair of 16-row tiles
        float v[8];
        int chunk, pair;
        if (D.orient == 0) {   // rows from R2 (columns of the block), k from R1 (its rows)
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = tile[q][k0 + j][rip];
            chunk = D.chunk0[seg] + rbs;
            pair = cb;
        } else {               // rows from R1, k from R2
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = tile[q][rip][k0 + j];
            chunk = cb;
            pair = rbs;
        }
        const int tapd = D.tapmode == 0 ? tap : (D.tapmode == 1 ? J.ntap - 1 - tap : 0);
        bf16_t* base = D.base[D.orient == 0 ? 0 : (D.tapmode == 2 ? 0 : seg)] + (D.tapmode == 2 ? (long)tap * D.tap_buf_stride : 0);
        const int ntl = D.ntiles[D.orient == 0 || D.tapmode == 2 ? 0 : seg];
        const long e = ((((long)chunk * D.ntap + tapd) * ntl + 2 * pair + tl) << 9) + lane * 8;
        u32x4 o = {pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]), pack_bf2(v[4], v[5]), pack_bf2(v[6], v[7])};
        up_st_stream((u32x4*)(base + e), o);
    }
}
__global__ void __launch_bounds__(256) k_update_pack_many(const UpJob* __restrict__ jobs, int njobs, float lr, float mu, float gscale) {
    __shared__ int sj;
    __shared__ float tile[4][32][33];
    if (threadIdx.x < 64) {
        // last job whose block_start <= blockIdx.x = (number of such jobs) - 1 (block_start ascends): every lane tests its share, ONE round
        // trip to L2 instead of the seven dependent ones of a binary search by lane 0 (a third of a workgroup's short life)
        int cnt = 0;
        for (int j = threadIdx.x; j < njobs; j += 64) cnt += jobs[j].block_start <= (int)blockIdx.x ? 1 : 0;
        for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o);
        if (threadIdx.x == 0) sj = cnt - 1;
    }
    __syncthreads();
    const UpJob& J = jobs[sj];
    const int b = blockIdx.x - J.block_start;
    if (J.kind == 0) {   // plain Momentum over a contiguous range (k_momentum's arithmetic)
        const long n = J.n, n4 = n >> 2;
        const long i0 = (long)b * (UP_EPB / 4);
        for (int q = 0; q < UP_EPB / 4 / 256; ++q) {
            const long i = i0 + q * 256 + threadIdx.x;
            if (i < n4) {
                f32x4 a = ((f32x4*)J.acc)[i], gv = ((const f32x4*)J.g)[i], wv = ((f32x4*)J.w)[i];
                a = mu * a + gscale * gv;
                wv -= lr * a;
                ((f32x4*)J.acc)[i] = a;
                ((f32x4*)J.w)[i] = wv;
            }
        }
        if (b == 0 && threadIdx.x < (n & 3)) {
            const long i = (n4 << 2) + threadIdx.x;
            const float a = mu * J.acc[i] + gscale * J.g[i];
            J.acc[i] = a;
            J.w[i] -= lr * a;
        }
        return;
    }
    // ---- a packed tensor: workgroup b = (tap, row block rb, group of four column blocks cg)
    const int ncg = (J.ncb + 3) >> 2;
    const int cg = b % ncg, rb = (b / ncg) % J.nrb, tap = b / (ncg * J.nrb);
    const float* gp = J.g;
    update_pack_block(J, tile, tap, rb, cg, lr, mu, gscale, [gp](long i) { return up_ld_stream((const f32x4*)gp + i); });
}
// The same pass for ONE concat source (R1 segment `seg`) of ONE conv kernel, fed by the weight-gradient slabs of that source instead of a
// finished gradient: the reduce launch of the slabs IS the update (the fp32 gradient is neither written nor re-read,
// 8 of 28 B per weight; and the Momentum pass of these tensors leaves the tail of the step). Summation orders are those of
// k_reduce_slabs / k_reduce_slabs_wide (sequential below 16 splits; else eight interleaved partial sums combined in order), so w, acc and
// the packed copies equal reduce -> k_update_pack_many bit for bit. nsplit == 1: `slab` is the gradient the weight-gradient kernel wrote
// in place. gout (optional): the reduced gradient is stored too (tests, hosts that log gradient norms). The last blocks reduce the n2
// float4 items of the bias row (behind the taps of every slab) into out2, as k_reduce_slabs does.
__device__ __forceinline__ f32x4 slab_sum(const float* __restrict__ slab, long stride, int nsplit, long e) {
    if (nsplit < 16) {
        f32x4 t = {0.f, 0.f, 0.f, 0.f};
        for (int z = 0; z < nsplit; ++z) t += *(const f32x4*)(slab + z * stride + e);
        return t;
    }
    f32x4 part[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) part[q] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int z0 = 0; z0 < nsplit; z0 += 8) {
#pragma unroll
        for (int q = 0; q < 8; ++q)
            if (z0 + q < nsplit) part[q] += *(const f32x4*)(slab + (z0 + q) * stride + e);
    }
    f32x4 r = part[0];
#pragma unroll
    for (int q = 1; q < 8; ++q) r += part[q];
    return r;
}
__global__ void __launch_bounds__(256) k_update_pack_seg(const UpJob J, int seg, const float* __restrict__ slab, long stride, int nsplit,
                                                         float* __restrict__ gout, float* __restrict__ out2, int n2, int tensor_blocks, float lr, float mu,
                                                         float gscale) {
    __shared__ float tile[4][32][33];
    const int b = blockIdx.x;
    if (b >= tensor_blocks) {   // the bias row
        const int c4 = (b - tensor_blocks) * 256 + threadIdx.x;
        if (c4 < n2) *(f32x4*)(out2 + c4 * 4) = slab_sum(slab, stride, nsplit, (long)J.ntap * J.R1 * J.R2 + c4 * 4);
        return;
    }
    const int ncg = (J.ncb + 3) >> 2;
    const int nrb_seg = (J.seg_c[seg] + 31) >> 5;
    const int cg = b % ncg, rb = J.seg_blk0[seg] + (b / ncg) % nrb_seg, tap = b / (ncg * nrb_seg);
    update_pack_block(J, tile, tap, rb, cg, lr, mu, gscale, [=](long i) {
        const f32x4 g = slab_sum(slab, stride, nsplit, i * 4);
        if (gout && nsplit > 1) *(f32x4*)(gout + i * 4) = g;
        return g;
    });
}
hipError_t ew_update_pack_seg(const UpJob& J, int seg, const float* slab, long stride, int nsplit, float* gout, float* out2, int n2, float lr, float mu,
                              float gscale, hipStream_t st) {
    const int tensor_blocks = J.ntap * ((J.seg_c[seg] + 31) / 32) * ((J.ncb + 3) >> 2);
    const int bias_blocks = (out2 && nsplit > 1) ? (n2 + 255) / 256 : 0;
    hipLaunchKernelGGL(k_update_pack_seg, dim3(tensor_blocks + bias_blocks), dim3(256), 0, st, J, seg, slab, stride, nsplit, gout, out2, n2, tensor_blocks, lr, mu,
                       gscale);
    return hipGetLastError();
}
int ew_update_job_blocks(const UpJob& j) {
    if (j.kind == 0) return (int)((j.n + UP_EPB - 1) / UP_EPB);
    return j.ntap * j.nrb * ((j.ncb + 3) >> 2);
}
hipError_t ew_update_pack_many(const UpJob* jobs_dev, int njobs, int total_blocks, float lr, float mu, float gscale, hipStream_t st) {
    hipLaunchKernelGGL(k_update_pack_many, dim3(total_blocks), dim3(256), 0, st, jobs_dev, njobs, lr, mu, gscale);
    return hipGetLastError();
}
hipError_t ew_extract_tiles(const float* imgs, float* tiles, int H, int S, int P, int stride, int pps, long t0, long ntiles, hipStream_t st) {
    hipLaunchKernelGGL(k_extract_tiles, dim3(grid_for(ntiles * S * S, 256)), dim3(256), 0, st, imgs, tiles, H, S, P, stride, pps, t0, ntiles);
    return hipGetLastError();
}
hipError_t ew_overlap_add(const float* prob, float* acc, float* hits, int nimg, int H, int P, int stride, int pps, long t0, long ntiles, hipStream_t st) {
    hipLaunchKernelGGL(k_overlap_add, dim3(grid_for((long)nimg * H * H, 256)), dim3(256), 0, st, prob, acc, hits, nimg, H, P, stride, pps, t0, ntiles);
    return hipGetLastError();
}
hipError_t ew_overlap_finish(const float* acc, const float* hits, float* out, long n, hipStream_t st) {
    hipLaunchKernelGGL(k_overlap_finish, dim3(grid_for(n, 256)), dim3(256), 0, st, acc, hits, out, n);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// post-processing (src/images.py:88-99, 256-266; src/summary.py:134-147)
// ---------------------------------------------------------------------------------------------
// One 256-thread workgroup per patch_size x patch_size block of a mask [nimg][S][S] (patch_size <= 64): the block statistic
//   mode 0 (quantize_mask, images.py:256-266):       label = mean(mask >= 0.5) > threshold, written over the whole block of `out`
//   mode 1 (labels_for_patches, images.py:88-99):    label = mean(mask) > threshold, written to labels[img][bx][by] (x outer: the
//                                                    patch order of extract_patches) as int64
// Blocks cut by the image edge average over the pixels they hold (numpy slicing does the same).
// (mask and out may be the SAME buffer -- quantize_mask in place, rsu.h: no __restrict__ on them; a block's stores follow the barrier of
// its reduction, and blocks do not overlap)
__global__ void __launch_bounds__(256) k_block_label(const float* mask, float* out, int64_t* __restrict__ labels,
                                                     int S, int ps, int nb, float thr, int mode) {
    const int b = blockIdx.x, by = b % nb, bx = (b / nb) % nb, img = b / (nb * nb);
    const int y0 = by * ps, x0 = bx * ps;
    const int hh = min(ps, S - y0), ww = min(ps, S - x0);
    float sum = 0.f;
    for (int i = threadIdx.x; i < hh * ww; i += 256) {
        const float v = mask[((long)img * S + y0 + i / ww) * S + x0 + i % ww];
        sum += mode == 0 ? (v >= 0.5f ? 1.f : 0.f) : v;
    }
    __shared__ float red[256];
    red[threadIdx.x] = sum;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    const float lab = (red[0] / (float)(hh * ww)) > thr ? 1.f : 0.f;
    if (mode == 0) {
        for (int i = threadIdx.x; i < hh * ww; i += 256) out[((long)img * S + y0 + i / ww) * S + x0 + i % ww] = lab;
    } else if (threadIdx.x == 0) {
        labels[((long)img * nb + bx) * nb + by] = (int64_t)lab;
    }
}
hipError_t ew_block_label(const float* mask, float* out, int64_t* labels, int nimg, int S, int ps, float thr, int mode, hipStream_t st) {
    const int nb = (S + ps - 1) / ps;
    hipLaunchKernelGGL(k_block_label, dim3(nimg * nb * nb), dim3(256), 0, st, mask, out, labels, S, ps, nb, thr, mode);
    return hipGetLastError();
}
// tf.metrics.{accuracy, recall, precision} counters (summary.py:141-147): counts[0..3] += TP, FP, FN, TN over n int64 labels
__global__ void __launch_bounds__(256) k_confusion(const int64_t* __restrict__ pred, const int64_t* __restrict__ truth, long n,
                                                   unsigned long long* __restrict__ counts) {
    unsigned c[4] = {0, 0, 0, 0};
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const int pr = pred[i] != 0, t = truth[i] != 0;
        ++c[pr ? (t ? 0 : 1) : (t ? 2 : 3)];
    }
    __shared__ unsigned red[4][256];
    for (int k = 0; k < 4; ++k) red[k][threadIdx.x] = c[k];
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s)
            for (int k = 0; k < 4; ++k) red[k][threadIdx.x] += red[k][threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x < 4) atomicAdd(&counts[threadIdx.x], (unsigned long long)red[threadIdx.x][0]);
}
hipError_t ew_confusion(const int64_t* pred, const int64_t* truth, long n, unsigned long long* counts, hipStream_t st) {
    int grid = (int)((n + 255) / 256);
    if (grid > 1024) grid = 1024;
    hipLaunchKernelGGL(k_confusion, dim3(grid), dim3(256), 0, st, pred, truth, n, counts);
    return hipGetLastError();
}
