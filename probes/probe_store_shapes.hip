// Probe (developer tool, GPU box): what a tile epilogue's 16-byte-per-lane stores cost by the SHAPE of the wave instruction.
// 256 workgroups x 8 waves, every wave stores its 64 pixels x 128 bytes of a 128 x 256 tile (NHWC, Cout = 128: pixel stride 256 B) as
// 8 buffer_store_dwordx4 per tile, tile after tile, into a buffer of `tiles_in_buf` tiles per workgroup (small: the stores stay in L2;
// large: they stream to memory). Shapes:
//   0  MFMA layout as it stands: lane (g4, l15) -> pixel l15 of fragment pt, 16 bytes g4 of the 64-byte channel group pp
//   1  quad = one pixel's 64 bytes: lane L -> pixel L >> 2, bytes 16 * (L & 3)
//   2  8 lanes = one pixel's 128 bytes: lane L -> pixel L >> 3 (8 pixels per instruction), bytes 16 * (L & 7)
//   3  1 KiB contiguous per instruction (no pixel structure: the ceiling)
// build: hipcc -O3 --offload-arch=gfx950 probes/probe_store_shapes.hip -o probes/probe_store_shapes
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
template <int SHAPE>
__global__ __launch_bounds__(512) void k_store(char* out, int ntiles, int tiles_in_buf, int burst_gap) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wco = wave >> 2, wpx = wave & 3;   // 2 channel halves x 4 pixel quarters (64 pixels each)
    const int g4 = lane >> 4, l15 = lane & 15;
    u32x4 v = {(unsigned)lane, 1u, 2u, 3u};
    const long wg_bytes = (long)tiles_in_buf * 65536;
    char* base = out + (long)blockIdx.x * wg_bytes;
    for (int t = 0; t < ntiles; ++t) {
        char* tb = base + (long)(t % tiles_in_buf) * 65536;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int pt = e >> 1, pp = e & 1;
            long off;
            if (SHAPE == 0) off = (long)((wpx * 4 + pt) * 16 + l15) * 256 + wco * 128 + pp * 64 + g4 * 16;
            else if (SHAPE == 1) off = (long)((wpx * 4 + pt) * 16 + (lane >> 2)) * 256 + wco * 128 + pp * 64 + (lane & 3) * 16;
            else if (SHAPE == 2) off = (long)((wpx * 4 + pt) * 16 + pp * 8 + (lane >> 3)) * 256 + wco * 128 + (lane & 7) * 16;
            else off = (long)(wave * 8 + e) * 1024 + lane * 16;
            *(u32x4*)(tb + off) = v;
        }
        v.y += 1;
        if (burst_gap) {   // idle time between the bursts (a tile's main loop), so that the stores of a tile arrive together as in the kernel
            __builtin_amdgcn_s_barrier();
            for (int i = 0; i < burst_gap; ++i) __builtin_amdgcn_s_sleep(127);
        }
    }
}
int main(int argc, char** argv) {
    const int ntiles = 400;
    char* buf;
    const int tib_max = 64;
    hipMalloc(&buf, (size_t)256 * tib_max * 65536);
    hipMemset(buf, 0, (size_t)256 * tib_max * 65536);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int gap : {0, 2}) for (int tib : {1, 64}) for (int shape = 0; shape < 4; ++shape) {
        float best = 1e9f;
        for (int rep = 0; rep < 4; ++rep) {
            hipEventRecord(a);
            switch (shape) {
                case 0: k_store<0><<<256, 512>>>(buf, ntiles, tib, gap); break;
                case 1: k_store<1><<<256, 512>>>(buf, ntiles, tib, gap); break;
                case 2: k_store<2><<<256, 512>>>(buf, ntiles, tib, gap); break;
                default: k_store<3><<<256, 512>>>(buf, ntiles, tib, gap); break;
            }
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
        }
        const double bytes = 256.0 * ntiles * 65536;
        printf("gap %d  buffer %2d tiles/workgroup (%4d MiB)  shape %d: %8.3f ms  %6.2f TB/s  %7.0f ns per tile per CU\n", gap, tib, 256 * tib / 16, shape, best,
               bytes / best / 1e9, best * 1e6 / ntiles);
    }
    return 0;
}
