// igemm_wgt: weight gradient (+ bias gradient) of the 2x2 stride-2 transposed convolution (unet.py:67-68) as a ping-pong kernel.
//   dK[a][b][co][ci] = sum_m dy[n][2y+a][2x+b][co] * x[m][ci],   db[co] = sum over all dy pixels,   m = (n, y, x) low-resolution pixels
// Rounds 1-3a ran this through the generic igemm_wgrad kernel (4 taps, stride 2, 64x64 shape): 16 MFMAs per k-step and wave behind
// 16 transposed LDS reads with a 4-way bank conflict on the stride-2 pixels, every wave reading, multiplying and staging at once --
// 260-290 TFLOP/s, 12 % of the MFMA issue slots (profiles/r03/pmc_step_kernels.csv). What is different here:
//   * the four taps are four PHASE IMAGES of dy: the LDS-DMA gather de-interleaves them while staging (a lane fetches pixel
//     (2y+a, 2x+b) of its low-resolution pixel), so every transposed read is a stride-1 read of the layout the 3x3 kernels use (no
//     bank conflicts) and a tap is a constant LDS offset;
//   * a workgroup owns 128 ci x 64 co x 4 taps; a wave 64 ci x 32 co x 4 taps = 32 accumulator tiles: 32 MFMAs per 32-pixel k-step
//     behind 24 transposed reads (12 fragments), the ratio of igemm_wgpp;
//   * ping-pong over the k-steps: a tile is 64 low-resolution pixels = two k-steps; wave group G0 (waves 0-3) reduces k-step 0 of
//     every tile, G1 k-step 1, into accumulators of their own; one group reads and stages (R interval) while the other multiplies (M
//     interval), one workgroup barrier per interval; the groups' sums meet once, through LDS, behind the last tile (fixed order);
//   * tiles are flat runs of the pixel index m (no padding of the small levels); every piece a wave stages holds the SAME eight
//     pixels (piece w of each of the two F planes and the four dy phases), so a lane decodes one pixel per tile; three ring slots
//     of 48 KiB, the pieces of tile t+2 issued in the R interval of tile t; each wave waits with ONE counted s_waitcnt vmcnt(6) per
//     tile (its own six pieces of tile t+2 may stay in flight), the stream never ends (behind the last tile it re-fetches it into the
//     slot nobody reads).
// Slabs, the bias row behind the taps and the reduction are those of igemm_wgrad (k_reduce_slabs); a single split writes in place.
#include "igemm_wgrad_body.h"

struct IgWgtParams {
    const bf16_t* x;      // [N][H][W][Cin]
    const bf16_t* dy;     // [N][2H][2W][Cout]
    float* slab;          // [nsplit] x ([4][Cout][Cin] + bias row)
    float* sbslab;        // per-split column sums of dy, or null
    long slab_stride;
    int N, H, W, Cin, Cout;
    int nsplit, ntiles;   // pixel splits (grid.z) and 64-pixel tiles in all
    unsigned w_magic;     // floor(2^32 / W)
};

namespace {
constexpr int WGT_PLANE = 64 * 128;            // 64 pixels x 64 channels
constexpr int WGT_SLOT = 6 * WGT_PLANE;        // two x planes + four dy phases
constexpr int WGT_NSLOT = 3;
}  // namespace

__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) igemm_wgt_kernel(const IgWgtParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __attribute__((address_space(3))) char* lds = (__attribute__((address_space(3))) char*)smem;
    const int lid = xcd_contiguous_id(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z), gridDim.x * gridDim.y * gridDim.z);
    const int cfb = lid % gridDim.x, csb = (lid / gridDim.x) % gridDim.y, z = lid / (gridDim.x * gridDim.y);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int grp = wave >> 2, w4 = wave & 3, wcf = w4 >> 1, wcs = w4 & 1;
    const int g4 = lane >> 4, l15 = lane & 15, q4 = l15 >> 2, p4 = lane & 3;
    const int P = p.N * p.H * p.W, W2 = 2 * p.W;
    const int n_mine = z < p.ntiles ? (p.ntiles - z + p.nsplit - 1) / p.nsplit : 0;   // tiles z, z + nsplit, ...

    f32x4 acc[4][4][2];   // [tap][ci tile][co tile]
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) acc[t][a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    const bool do_sbias = p.sbslab != nullptr && cfb == 0 && wcf == 0;   // (wave-uniform) column sums of dy: ones x S, in the waves of ci block 0
    f32x4 accs[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};

    // ---- transposed reads: this lane's pixel of the group's k-step (read rd covers pixels rd*16 .. rd*16+15 of it) inside a plane;
    // tile ct of the 64 channels sits in 32-byte block ct ^ ((ml >> 1) & 3) of the pixel's 128 bytes (the layout of igemm_wgrad's F tile)
    // (per lane: 8 offsets for the four ci tiles of the wave's x plane, 4 for its two co tiles inside ANY dy phase plane; the ring slot
    // is a scalar added per interval, the phase plane an immediate offset of the read)
    int fx[2][4], sx[2][2];
#pragma unroll
    for (int rd = 0; rd < 2; ++rd) {
        const int ml = grp * 32 + rd * 16 + 4 * g4 + q4;
        const int rbase = ml * 128 + (((ml >> 1) & 3) << 5) + p4 * 8;
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) fx[rd][ct] = (wcf * WGT_PLANE + rbase) ^ (ct << 5);
#pragma unroll
        for (int st = 0; st < 2; ++st) sx[rd][st] = rbase ^ ((wcs * 2 + st) << 5);
    }
    // ---- staging: piece `wave` of every plane = pixels wave*8 .. wave*8+7 of the tile, lane = (pixel, 16-byte position)
    const int dpix = wave * 8 + (lane >> 3);
    const int cch = (lane & 7) ^ (((dpix >> 1) & 3) << 1);   // source chunk (8 channels) this position holds
    const bool fok0 = cfb * 128 + cch * 8 < p.Cin, fok1 = cfb * 128 + 64 + cch * 8 < p.Cin, sok = csb * 64 + cch * 8 < p.Cout;
    const __amdgpu_buffer_rsrc_t rF = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rS = __builtin_amdgcn_make_buffer_rsrc((void*)p.dy, 0, 0x7fffffff, 0x00020000);
    auto issue = [&](int i, int slot) {   // the six pieces of this wave for its split's tile i
        int dp = dpix;
        asm volatile("" : "+v"(dp));      // (keeps the pixel arithmetic inside the interval instead of in registers across the loop)
        const int m = (z + i * p.nsplit) * 64 + dp;
        const bool ok = m < P;
        const int mc = ok ? m : 0;
        unsigned q = __umulhi((unsigned)mc, p.w_magic);   // q = n * H + y (floor(2^32 / W) may fall one short)
        int xx = mc - (int)q * p.W;
        if (xx >= p.W) { ++q; xx -= p.W; }
        const unsigned fo = (unsigned)((mc * p.Cin + cfb * 128 + cch * 8) * 2);
        const unsigned so = (unsigned)((((2 * (int)q) * W2 + 2 * xx) * p.Cout + csb * 64 + cch * 8) * 2);
        __attribute__((address_space(3))) char* dst = lds + slot * WGT_SLOT + wave * 1024;
        bdma16(rF, (ok && fok0) ? fo : RSU_SENT, 0u, (void*)dst);
        bdma16(rF, (ok && fok1) ? fo + 128u : RSU_SENT, 0u, (void*)(dst + WGT_PLANE));
        const unsigned sv_ = (ok && sok) ? so : RSU_SENT;
#pragma unroll
        for (int t = 0; t < 4; ++t)
            bdma16(rS, sv_, (unsigned)((((t >> 1) * W2 + (t & 1)) * p.Cout) * 2), (void*)(dst + (2 + t) * WGT_PLANE));
    };
    auto bar = [&]() {
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };
    if (n_mine == 0) return;   // (workgroup-uniform: ntiles >= nsplit by construction, kept for safety)

    // ---- prologue: tiles 0 and 1 of the split, complete before anybody reads
    issue(0, 0);
    issue(n_mine > 1 ? 1 : 0, 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    bar();
    if (grp) bar();   // G1 sits out interval 0

    for (int i = 0; i < n_mine; ++i) {
        const int slot = i % WGT_NSLOT;
        // ================= R interval: this group's k-step of tile i into registers, then the pieces of tile i + 2
        bf16x8 fa[4], sv[4][2];
        {
            __attribute__((address_space(3))) char* sl = lds + slot * WGT_SLOT;   // (wave-uniform)
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) {
                const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bf16x4*)(sl + fx[0][ct]));
                const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bf16x4*)(sl + fx[1][ct]));
                fa[ct] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
            }
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                __attribute__((address_space(3))) char* s0 = sl + sx[0][st];
                __attribute__((address_space(3))) char* s1 = sl + sx[1][st];
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bf16x4*)(s0 + (2 + t) * WGT_PLANE));
                    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bf16x4*)(s1 + (2 + t) * WGT_PLANE));
                    sv[t][st] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                }
            }
        }
        {
            const int nx = i + 2 < n_mine ? i + 2 : n_mine - 1;   // behind the last tile: that tile again, into the slot nobody reads
            issue(nx, (i + 2) % WGT_NSLOT);
        }
        if (grp) RSU_WG_WAIT(6);   // G1's R intervals are the odd ones: everything but the six pieces just issued has landed
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        bar();
        // ================= M interval: 32 MFMAs (+ 8 for the bias sums in the waves that carry them)
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int ct = 0; ct < 4; ++ct)
#pragma unroll
                for (int st = 0; st < 2; ++st) mfma_bf16_inplace(acc[t][ct][st], fa[ct], sv[t][st]);
        if (do_sbias) {
            unsigned o1 = 0x3f803f80u;
            asm volatile("" : "+v"(o1));
            const u32x4 o4 = {o1, o1, o1, o1};
            bf16x8 ones = __builtin_bit_cast(bf16x8, o4);
            asm volatile("s_nop 3" : "+v"(ones));   // VALU-written operand -> (asm) MFMA read
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int st = 0; st < 2; ++st) mfma_bf16_inplace(accs[st], ones, sv[t][st]);
        }
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        if (!grp) RSU_WG_WAIT(6);  // G0's M intervals are the odd ones
        bar();
    }
    mfma_results_fence();
    if (!grp) bar();   // G0 sits out G1's last M interval
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // nothing may land in LDS after this point
    __syncthreads();

    // ---- G1 hands its sums to G0 through LDS (the ring is dead), fixed order G0 + G1
    {
        f32x4 __attribute__((address_space(3)))* red = (f32x4 __attribute__((address_space(3)))*)lds;
        const int slot = w4 * 64 + lane;
        if (grp) {
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int b = 0; b < 2; ++b) red[((t * 4 + a) * 2 + b) * 256 + slot] = acc[t][a][b];
#pragma unroll
            for (int b = 0; b < 2; ++b) red[(32 + b) * 256 + slot] = accs[b];
        }
        __syncthreads();
        if (grp) return;
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) acc[t][a][b] += red[((t * 4 + a) * 2 + b) * 256 + slot];
#pragma unroll
        for (int b = 0; b < 2; ++b) accs[b] += red[(32 + b) * 256 + slot];
    }
    if (do_sbias && g4 == 0) {   // every row of accs holds the same sums: row 0 writes them
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            const int cs = csb * 64 + (wcs * 2 + st) * 16 + l15;
            if (cs < p.Cout) p.sbslab[(long)z * p.slab_stride + cs] = accs[st][0];
        }
    }
    // ---- this split's slab: rows = ci (4 consecutive per lane), cols = co
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            const int cs = csb * 64 + (wcs * 2 + st) * 16 + l15;
            if (cs >= p.Cout) continue;
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) {
                const int cf = cfb * 128 + (wcf * 4 + ct) * 16 + 4 * g4;
                if (cf >= p.Cin) continue;
                *(f32x4*)(p.slab + (long)z * p.slab_stride + (((long)t * p.Cout + cs) * p.Cin + cf)) = acc[t][ct][st];
            }
        }
}

bool igemm_wgt_supports(int N, int H, int W, int Cin, int Cout) {
    return Cin % 8 == 0 && Cout % 8 == 0 && Cin >= 64 && Cout >= 32 && (long)N * H * W < (1l << 28) && (long)N * 4 * H * W * Cout * 2 < 0x7ffffff0L &&
           (long)N * H * W * Cin * 2 < 0x7ffffff0L;
}
// workgroups per pixel split: ceil(Cin / 128) x ceil(Cout / 64)
int igemm_wgt_blocks(int Cin, int Cout) { return ((Cin + 127) / 128) * ((Cout + 63) / 64); }
int igemm_wgt_tiles(int N, int H, int W) { return (N * H * W + 63) / 64; }
hipError_t igemm_wgt_launch(const void* x, const void* dy, float* slab, float* sbslab, long slab_stride, int N, int H, int W, int Cin, int Cout,
                            int nsplit, hipStream_t st) {
    IgWgtParams p;
    p.x = (const bf16_t*)x; p.dy = (const bf16_t*)dy; p.slab = slab; p.sbslab = sbslab; p.slab_stride = slab_stride;
    p.N = N; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout;
    p.ntiles = igemm_wgt_tiles(N, H, W);
    p.nsplit = nsplit < 1 ? 1 : (nsplit > p.ntiles ? p.ntiles : nsplit);
    p.w_magic = W <= 1 ? 0xffffffffu : (unsigned)(0x100000000ull / (unsigned)W);
    const size_t lds = (size_t)WGT_NSLOT * WGT_SLOT;   // 144 KiB; the hand-off needs 34 x 4 KiB of it
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)igemm_wgt_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    hipLaunchKernelGGL(igemm_wgt_kernel, dim3((Cin + 127) / 128, (Cout + 63) / 64, p.nsplit), dim3(512), lds, st, p);
    return hipGetLastError();
}
