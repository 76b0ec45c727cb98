// Hardware probe (developer tool, not part of the product path): pins the gfx950
// lane<->element maps the kernels in road_segmentation_unet_amd/csrc rely on.
//   1. v_mfma_f32_16x16x32_bf16 A/B/C maps (integer data, asymmetric operands)
//   2. ds_read_b64_tr_b16 (LDS transposed read) lane map
// Build: hipcc --offload-arch=gfx950 -O2 probes/probe_layouts.hip -o probes/probe_layouts
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <cstdint>

typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;

static inline uint16_t f2bf(float f) {
    uint32_t u; memcpy(&u, &f, 4);
    u += 0x7FFF + ((u >> 16) & 1);
    return (uint16_t)(u >> 16);
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2);} } while (0)

// A is [16][32] row-major bf16, B is [32][16] row-major bf16 (k rows). D is [16][16] f32.
// Assumed maps: lane l holds A[l&15][8*(l>>4)+j], B[8*(l>>4)+j][l&15]; D: col=l&15,row=4*(l>>4)+r
__global__ void k_mfma16(const uint16_t* A, const uint16_t* B, float* D) {
    int l = threadIdx.x;
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) {
        a[j] = (short)A[(l & 15) * 32 + 8 * (l >> 4) + j];
        b[j] = (short)B[(8 * (l >> 4) + j) * 16 + (l & 15)];
    }
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D[(4 * (l >> 4) + r) * 16 + (l & 15)] = c[r];
}

// LDS image: rows of `pitch` bytes, element (row, col) 16-bit = row*256+col as an integer tag.
// Each lane supplies address of row (k0 + 4*(l>>4) + ((l&15)>>2)), col 4*(l&3) and receives 4 shorts.
__global__ void k_trread(uint16_t* out, int pitch) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int l = threadIdx.x;
    for (int i = l; i < 64 * 64; i += 64) {
        int row = i / 64, col = i % 64;
        *(uint16_t*)(smem + row * pitch + col * 2) = (uint16_t)(row * 256 + col);
    }
    __syncthreads();
    int g = l >> 4, q = (l & 15) >> 2, p = l & 3;
    int row = 4 * g + q;
    int c0 = 16;  // block's first column
    __attribute__((address_space(3))) s16x4* ptr =
        (__attribute__((address_space(3))) s16x4*)(smem + row * pitch + (c0 + 4 * p) * 2);
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(ptr);
    for (int j = 0; j < 4; ++j) out[l * 4 + j] = (uint16_t)v[j];
}

int main() {
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    printf("device: %s arch=%s CUs=%d sharedPerBlock=%zu maxSharedOptin=%zu clock=%d kHz L2=%d\n", prop.name, prop.gcnArchName,
           prop.multiProcessorCount, prop.sharedMemPerBlock, prop.sharedMemPerBlockOptin, prop.clockRate, prop.l2CacheSize);
    // ---- MFMA probe
    std::vector<uint16_t> A(16 * 32), B(32 * 16);
    std::vector<float> Af(16 * 32), Bf(32 * 16);
    for (int i = 0; i < 16; ++i) for (int k = 0; k < 32; ++k) { float v = (float)((i * 3 + k * 5) % 7 - 3); Af[i * 32 + k] = v; A[i * 32 + k] = f2bf(v); }
    for (int k = 0; k < 32; ++k) for (int j = 0; j < 16; ++j) { float v = (float)((k * 2 + j * 7) % 5 - 2); Bf[k * 16 + j] = v; B[k * 16 + j] = f2bf(v); }
    uint16_t *dA, *dB; float* dD;
    CK(hipMalloc(&dA, A.size() * 2)); CK(hipMalloc(&dB, B.size() * 2)); CK(hipMalloc(&dD, 256 * 4));
    CK(hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dB, B.data(), B.size() * 2, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_mfma16, dim3(1), dim3(64), 0, 0, dA, dB, dD);
    CK(hipDeviceSynchronize());
    std::vector<float> D(256);
    CK(hipMemcpy(D.data(), dD, 256 * 4, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
        float ref = 0; for (int k = 0; k < 32; ++k) ref += Af[i * 32 + k] * Bf[k * 16 + j];
        if (ref != D[i * 16 + j]) { if (bad < 5) printf("  mfma mismatch D[%d][%d]=%g ref %g\n", i, j, D[i * 16 + j], ref); ++bad; }
    }
    printf("MFMA16x16x32 map check: %s (%d mismatches)\n", bad ? "FAIL" : "OK", bad);

    // ---- tr read probe
    uint16_t* dO; CK(hipMalloc(&dO, 256 * 2));
    for (int pitch : {128, 160}) {
        hipLaunchKernelGGL(k_trread, dim3(1), dim3(64), 64 * 256, 0, dO, pitch);
        CK(hipDeviceSynchronize());
        std::vector<uint16_t> O(256);
        CK(hipMemcpy(O.data(), dO, 512, hipMemcpyDeviceToHost));
        // expectation: lane l (group g, i=l&15) gets column c0+i, rows 4g+0..3
        int badt = 0;
        for (int l = 0; l < 64; ++l) for (int j = 0; j < 4; ++j) {
            int g = l >> 4, i = l & 15;
            int exp = (4 * g + j) * 256 + 16 + i;
            if (O[l * 4 + j] != exp) { if (badt < 8) printf("  tr mismatch lane %d j %d got row %d col %d, expected row %d col %d\n", l, j, O[l * 4 + j] >> 8, O[l * 4 + j] & 255, exp >> 8, exp & 255); ++badt; }
        }
        printf("ds_read_tr16_b64 map check (pitch %d): %s (%d mismatches)\n", pitch, badt ? "FAIL" : "OK", badt);
        if (badt) { for (int l = 0; l < 20; ++l) printf("   lane %2d: (%d,%d) (%d,%d) (%d,%d) (%d,%d)\n", l, O[l*4]>>8, O[l*4]&255, O[l*4+1]>>8, O[l*4+1]&255, O[l*4+2]>>8, O[l*4+2]&255, O[l*4+3]>>8, O[l*4+3]&255); }
    }
    return 0;
}
