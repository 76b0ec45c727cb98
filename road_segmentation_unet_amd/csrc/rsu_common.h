// Shared device/host helpers for the gfx950 kernels of librsu_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef uint16_t bf16_t;
typedef __attribute__((ext_vector_type(8))) short bf16x8;  // one MFMA A/B fragment (8 bf16, 4 VGPRs)
typedef __attribute__((ext_vector_type(4))) short bf16x4;  // one ds_read_b64_tr_b16 result
typedef __attribute__((ext_vector_type(4))) float f32x4;   // one 16x16 accumulator tile slice
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;  // one 32x32 accumulator tile slice
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

#define RSU_LDS_PTR(T, p) ((__attribute__((address_space(3))) T*)(p))

// ---- bf16 <-> f32 -------------------------------------------------------------------------
__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
__device__ __forceinline__ float bf_lo(uint32_t u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf_hi(uint32_t u) { return __uint_as_float(u & 0xffff0000u); }
// round-to-nearest-even (plain cast: hipcc emits v_cvt_pk_bf16_f32 on gfx950, NaN stays NaN)
__device__ __forceinline__ uint32_t pack_bf2(float lo, float hi) {
    typedef __attribute__((ext_vector_type(2))) float f2;
    typedef __attribute__((ext_vector_type(2))) __bf16 b2;
    f2 v = {lo, hi};
    b2 r = __builtin_convertvector(v, b2);
    return __builtin_bit_cast(uint32_t, r);
}
__device__ __forceinline__ bf16_t f2bf(float f) {
    __bf16 r = (__bf16)f;
    return __builtin_bit_cast(bf16_t, r);
}

// ---- packed ReLU mask: 0xffff in each half of the result where the bf16 half of m is > 0 (sign clear, not zero), else 0:
// 0 - m with saturation (so that m = 0x8000, minus zero, becomes +32767), then an arithmetic shift by 15 spreads the sign: negative
// exactly where m was positive (bf16 sign bit == int16 sign bit). Two VALU instructions per two elements; as plain vector code hipcc
// turns the same test into two compares, two selects and a permute. A positive NaN counts as > 0. (`ones_pk` is no longer used.)
__device__ __forceinline__ unsigned pos_mask_pk_bf16(unsigned m, unsigned /*ones_pk*/ = 0) {
    unsigned t;
    asm("v_pk_sub_i16 %0, 0, %1 clamp\n\tv_pk_ashrrev_i16 %0, 15, %0 op_sel_hi:[0,1]" : "=&v"(t) : "v"(m));   // (op_sel_hi: both halves shift by the constant's LOW half)
    return t;
}

// ---- in-place MFMA accumulate: acc += A x B with vDst == SrcC guaranteed (inline asm, tied operand): no register is freed by
// an MFMA, so the compiler cannot rename an accumulator and re-use its old registers while the matrix pipe still reads them.
// The asm is opaque to hipcc's hazard recogniser, so the CALLER owns the wait states around it (DESIGN.md section 4;
// tools/check_mfma_hazards.py checks them in the compiled ISA): dependent MFMAs on one accumulator at least 4 MFMAs apart
// (they are 16 apart here), mfma_results_fence() straight behind the last MFMA before anything else reads the results, and an
// operand written by a VALU instruction followed by `s_nop 3` before the MFMA that reads it.
__device__ __forceinline__ void mfma_bf16_inplace(f32x4& acc, const bf16x8& a, const bf16x8& b) {
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
}
// XCD-aware workgroup numbering for one-workgroup-per-CU launches: the dispatcher deals consecutive workgroup ids round-robin over the 8
// XCDs (id & 7), each with its own L2. This turns the hardware id into a logical id such that each XCD owns a CONTIGUOUS range of logical
// ids: workgroups that share operand tiles (neighbouring logical ids) then fetch them through one L2 instead of eight.
__device__ __forceinline__ int xcd_contiguous_id(int hw_id, int total) {
    const int q = total >> 3, r = total & 7, x = hw_id & 7;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (hw_id >> 3);
}
__device__ __forceinline__ void mfma_results_fence() { asm volatile("s_nop 15\n\ts_nop 15" ::: "memory"); }

// ---- async global -> LDS copy, 16 bytes per lane (LDS destination = wave-uniform base + lane*16)
__device__ __forceinline__ void dma16(const void* gsrc, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// ---- exact division of a non-negative int by a runtime constant d >= 2: q = umulhi(n, ceil(2^32/d)),
// exact while n * d < 2^32 (host side: magic32() in rsu_api.hip)
__device__ __forceinline__ int div_magic(int n, unsigned magic) { return (int)__umulhi((unsigned)n, magic); }

// Tile geometry shared by the implicit-GEMM kernels. Output pixels of one image are cut into vertical
// strips of width SW; inside a strip pixels are flattened row-major (m = y*SW + tx) and a workgroup
// takes TM consecutive m. The input rows/cols such a tile touches are staged in LDS as a dense
// [R][CW] pixel image ("halo tile").
struct TileGeo {
    int SW;               // strip width (output pixels)
    int nstrips;          // ceil(Wo / SW)
    int tiles_per_strip;  // ceil(Ho*SW / TM)
    int CW;               // halo tile row pitch in pixels: (SW-1)*stride + (KW-1)*dil + 1, rounded up to 8
    int npix_max;         // LDS pixels reserved per buffer (multiple of 16)
    unsigned inv_SW, inv_CW;  // div_magic constants
};

#define HIP_CHECK_RET(x)                     \
    do {                                     \
        hipError_t e__ = (x);                \
        if (e__ != hipSuccess) {             \
            rsu_set_hip_error((int)e__);     \
            return RSU_EHIP;                 \
        }                                    \
    } while (0)
