// Probe: semantics of raw buffer load -> LDS (16 B/lane) on gfx950: out-of-range lanes, soffset and the range check.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdint>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(2);} } while (0)

__global__ void k(const uint32_t* src, uint32_t* out, unsigned nrec, unsigned soff, int mode) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int l = threadIdx.x;
    for (int i = l; i < 512; i += 64) ((uint32_t*)smem)[i] = 0xDEAD0000u + i;
    __syncthreads();
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, nrec, 0x00020000);
    unsigned voff = l * 16;
    if (mode == 1 && (l & 1)) voff = 0x80000000u;          // sentinel lanes
    if (mode == 2) voff = l * 16 + 4096;                    // beyond nrec via voffset
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)smem, 16, voff, soff, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = l; i < 256; i += 64) out[i] = ((uint32_t*)smem)[i];
}
int main() {
    std::vector<uint32_t> h(4096);
    for (int i = 0; i < 4096; ++i) h[i] = i;
    uint32_t *d, *o; CK(hipMalloc(&d, 16384)); CK(hipMalloc(&o, 1024));
    CK(hipMemcpy(d, h.data(), 16384, hipMemcpyHostToDevice));
    struct T { unsigned nrec, soff; int mode; const char* name; } tests[] = {
        {16384, 0, 0, "plain"}, {16384, 2048, 0, "soffset 2048"}, {0x7fffffff, 0, 1, "odd lanes sentinel 0x80000000"},
        {2048, 0, 2, "voffset beyond nrec(2048)"}, {2048, 4096, 0, "soffset beyond nrec(2048), voffset in range"},
        {512, 0, 0, "nrec 512: lanes >= 32 out of range"}};
    for (auto& t : tests) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 2048, 0, d, o, t.nrec, t.soff, t.mode);
        CK(hipDeviceSynchronize());
        std::vector<uint32_t> r(256); CK(hipMemcpy(r.data(), o, 1024, hipMemcpyDeviceToHost));
        printf("%-45s lane0: %08x %08x | lane1: %08x | lane2: %08x | lane33: %08x | lane63: %08x\n", t.name, r[0], r[1], r[4], r[8], r[132], r[252]);
    }
    return 0;
}
