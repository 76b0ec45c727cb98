// igemm_wgrad.hip -- launch side of the generic weight-gradient kernel; the kernel body lives in igemm_wgrad_body.h
#include "igemm_wgrad_body.h"

template <int WCF, int WCS, int CFT, int CST, int NTAP, int KW, int TMK, int KG>
__global__ void __launch_bounds__(WCF* WCS * KG * 64) __attribute__((amdgpu_waves_per_eu(WCF* WCS* KG / 4, WCF* WCS* KG / 4)))
igemm_wgrad_kernel(const IgWgradParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __attribute__((address_space(3))) char* lds = (__attribute__((address_space(3))) char*)smem;
    // (cfb, csb, z) from an XCD-contiguous numbering: the gx * gy workgroups of one pixel split z read the same F and S pixel tiles
    // (each F tile gy times, each S tile gx times) -- on one XCD all but the first of those reads are L2 hits
    const int lid = xcd_contiguous_id(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z), gridDim.x * gridDim.y * gridDim.z);
    const int cfb = lid % gridDim.x, csb = (lid / gridDim.x) % gridDim.y, z = lid / (gridDim.x * gridDim.y);
    igemm_wgrad_body<WCF, WCS, CFT, CST, NTAP, KW, TMK, KG>(p, lds, cfb, csb, z, threadIdx.x);
}

int igemm_wgrad_kgroups(int) { return 1; }  // wave groups are reduced inside the workgroup: one slab per grid.z slice
int igemm_wgrad_tmk(int) { return 128; }
int igemm_wgrad_cfb(int cfg) { return cfg == IGW_CFG_128x64 ? 128 : 64; }
int igemm_wgrad_csb(int cfg) { return cfg == IGW_CFG_64x16 ? 16 : 64; }

int igemm_wgrad_nsw(int cfg, int npix_max) {  // S pieces (1 KiB) per wave per tile, 8 waves
    const int csb = igemm_wgrad_csb(cfg);
    const int ppp = 64 / (csb / 8);
    return ((npix_max + ppp - 1) / ppp + 7) / 8;
}
size_t igemm_wgrad_lds_bytes(int cfg, int npix_max, int nbuf) {
    const int csb = igemm_wgrad_csb(cfg);
    const size_t staging = (size_t)nbuf * igemm_wgrad_tmk(cfg) * 2 * igemm_wgrad_cfb(cfg) +
                           (size_t)nbuf * igemm_wgrad_nsw(cfg, npix_max) * 8 * 1024;
    const size_t reduce = cfg == IGW_CFG_128x64 ? 0 : (size_t)(9 * 64 * csb + 64 * csb) * 4;  // wave-group hand-off (+ bias sums)
    return staging > reduce ? staging : reduce;
}

template <int CFG, int NTAP, int KW>
static hipError_t wlaunch_one(const IgWgradParams& p, int gx, int gy, int gz, hipStream_t st) {
    using C = WgCfg<CFG>;
    auto kern = igemm_wgrad_kernel<C::WCF, C::WCS, C::CFT, C::CST, NTAP, KW, C::TMK, C::KG>;
    const size_t lds = igemm_wgrad_lds_bytes(CFG, p.g.npix_max, p.nbuf);
    static size_t lds_set = 0;
    if (lds > lds_set) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        lds_set = lds;
    }
    hipLaunchKernelGGL(kern, dim3(gx, gy, gz), dim3(C::WCF * C::WCS * C::KG * 64), lds, st, p);
    return hipGetLastError();
}

hipError_t igemm_wgrad_launch(int cfg, int ntap, const IgWgradParams& p, int gx, int gy, int gz, hipStream_t st) {
    if (cfg == IGW_CFG_64x64) {
        if (ntap == 9) return wlaunch_one<IGW_CFG_64x64, 9, 3>(p, gx, gy, gz, st);
        if (ntap == 4) return wlaunch_one<IGW_CFG_64x64, 4, 2>(p, gx, gy, gz, st);
    } else if (cfg == IGW_CFG_64x16) {
        if (ntap == 9) return wlaunch_one<IGW_CFG_64x16, 9, 3>(p, gx, gy, gz, st);
    } else if (cfg == IGW_CFG_128x64) {
        if (ntap == 9) return wlaunch_one<IGW_CFG_128x64, 9, 3>(p, gx, gy, gz, st);
        if (ntap == 4) return wlaunch_one<IGW_CFG_128x64, 4, 2>(p, gx, gy, gz, st);
    }
    return hipErrorInvalidValue;
}
