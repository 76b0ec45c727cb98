// Parameter blocks + launch prototypes of the two MFMA implicit-GEMM kernels.
#pragma once
#include "rsu_common.h"

struct IgSrc {
    const bf16_t* ptr;
    int H, W, C;  // tensor dims
    int oy, ox;   // window origin
};

// ---------------------------------------------------------------------------------------------
// igemm_fwd: out[n][o][co] = epi( sum_tap sum_k  in[n][o*stride + tap*dil - pad][k] * A[tap][co][k] )
//   conv3x3 forward      : 9 taps, stride 1, pad 0
//   conv3x3 backward-data: 9 taps (flipped in the packing), stride 1, pad 2*dil, input = dz
//   convT 2x2 forward    : 1 tap, blockIdx.y = a*2+b selects the matrix and the output phase
//   convT 2x2 bwd-data   : 4 taps (2x2), stride 2, dil 1, input = dy
// ---------------------------------------------------------------------------------------------
struct IgFwdParams {
    IgSrc src[3];
    int nsrc;
    int nchunk[3];        // 32-channel chunks per source (ceil(C/32))
    const bf16_t* wp;     // packed A: [chunk][tap][tile16][lane][8]
    long wp_y_stride;     // elements between blockIdx.y slices of wp (convT forward), else 0
    int ntiles_w;         // 16-row tiles per (chunk, tap) in wp
    int tile_off;         // first tile of this launch's output channel 0
    const float* bias;    // [Cout] or null
    bf16_t* out;          // [N][oH][oW][outC]
    const bf16_t* mask_src;  // same geometry as out, or null
    bf16_t* pool_out;        // forward conv2 of an encoder level: the 2x2 max-pool of out, [N][oH/2][oW/2][outC] (igemm_pp, strip width 16 / 32), or null
    unsigned char* pool_code;  // ... and its code bytes [N][oH/2][oW/2][outC] (k_maxpool_fwd's), or null
    const void* zero_page;   // >= 64 zero bytes
    int N, Hin, Win;      // logical input window
    int Ho, Wo;           // output pixel grid of the GEMM (before output scatter)
    int Cout;             // channels produced (<= ncob*TN), multiple of 8
    int outC;             // channel pitch of out
    int dil, stride, pad;
    int oH, oW, ostride;  // out tensor geometry; out pixel = o*ostride + (blockIdx.y phase)
    int relu, accumulate;
    int ncob;
    int lsw;              // log2(g.SW) for the aligned-tile kernels (igemm_fwd2)
    // igemm_pp only. ksplit > 1: the reduction (the 32-channel chunks of all sources) is cut into ksplit contiguous slices, one workgroup
    // per (pixel tile, channel block, slice); a workgroup writes its fp32 partial sums to kslab (slice z at kslab + z * kslab_stride
    // floats, tile slot (tile * ncob + cob) * 2 * NST * 8 KiB inside it, in register order) and igemm_pp_finish_launch sums the slices in
    // order, applies bias (inside slice 0), ReLU / ReLU mask and stores bf16. cob_group (a divisor of ncob * ksplit, 0 = all): how many
    // (channel block, slice) units of one pixel tile sit on neighbouring workgroup ids (= on one XCD); the rest of an XCD's run walks the pixel
    // tiles -- weight-heavy layers want few units per XCD (each weight slice through ONE L2), halo-heavy layers all of them
    int ksplit, cob_group;
    float* kslab;
    long kslab_stride;
    int dbg;              // developer A/B switch (RSU_FWD_DBG): bit 0 = skip weight staging, bit 1 = skip halo staging (timing only)
    unsigned* stamps;     // diagnostic time stamps of igemm_pp (RSU_FWD_DBG bit 7 + RSU_STAMP_PTR), else null
    TileGeo g;
};

struct IgFwdCfgInfo { int TN, TM, threads; };
// persistent, counted-vmcnt pipeline (igemm_fwd2.hip)
enum { IGF2_CFG_128x256 = 0, IGF2_CFG_64x512 = 1, IGF2_CFG_128x128 = 2, IGF2_CFG_64x256 = 3, IGF2_CFG_128x192 = 4, IGF2_CFG_64x384 = 5,
       IGF2_CFG_128x320 = 6, IGF2_CFG_64x640 = 7, IGF2_NCFG = 8 };
IgFwdCfgInfo igemm_fwd2_cfg_info(int cfg);
int igemm_fwd2_max_pieces(int cfg, int ntap);
size_t igemm_fwd2_lds_bytes(int cfg, int ntap, int npix_max);
hipError_t igemm_fwd2_launch(int cfg, int ntap, const IgFwdParams& p, int grid_x, int grid_y, hipStream_t st);
// third generation (ping-pong wave groups; igemm_pp.hip): 3x3 taps, stride 1; tile shapes, LDS budget and results as igemm_fwd2
hipError_t igemm_pp_launch(int cfg, const IgFwdParams& p, int grid_x, hipStream_t st);
hipError_t igemm_pp_d2_launch(int cfg, const IgFwdParams& p, int grid_x, hipStream_t st);   // the same for dilation 2 (igemm_pp_d2.hip)
bool igemm_pp_d2_supports(int cfg, const IgFwdParams& p);
bool igemm_pp_has(int cfg);  // tile shapes the ping-pong kernel is built for
bool igemm_pp_has_ksplit(int cfg);   // ... and its split-K instantiations
bool igemm_pp_supports(int cfg, const IgFwdParams& p);  // ... and this planned launch is one of its instantiations
int igemm_pp_pool_lsw_mask(int cfg);  // strip widths (bit lsw) at which the shape can fold the 2x2 max-pool into its epilogue (0: none)
size_t igemm_pp_slab_floats(int cfg, const IgFwdParams& p);   // floats of ONE slice of the split-K slab of this planned launch
hipError_t igemm_pp_finish_launch(int cfg, const IgFwdParams& p, hipStream_t st);   // sums the slices of a split-K launch into p.out

// ---------------------------------------------------------------------------------------------
// igemm_ct (igemm_ct.hip): the 2x2 stride-2 transposed convolution as a ping-pong GEMM over the low-resolution pixels m = (n, y, x)
//   mode 0, forward      : a = x [N][H][W][Ca],   out = y [N][2H][2W][outC], Cn = Cout; packed weights = rsu_pack_convT_fwd's 4 phases
//   mode 1, backward-data: a = dy [N][2H][2W][Ca], out = dx [N][H][W][outC],  Cn = Cin;  packed weights = rsu_pack_convT_bwd's 4 taps
struct IgCtParams {
    const bf16_t* a;
    int Ca;               // channels per pixel of a (the reduction length per tap), multiple of 8
    int nchunk;           // ceil(Ca / 32)
    int N, H, W;          // low-resolution grid; N * H * W < 2^24
    unsigned magic_hw, magic_w;   // floor(2^32 / (H * W)), floor(2^32 / W) (0xffffffff for a divisor of 1): one multiply-high + one fix-up per division
    const bf16_t* wp;
    long wp_phase_stride; // elements between the packed matrices of two output phases (mode 0)
    int ntiles_w;         // 16-row tiles per (chunk, tap) block of wp
    const float* bias;    // [Cn] or null (mode 0)
    bf16_t* out;
    int outC;             // channel pitch of out
    int Cn;               // channels produced (mode 0: per output phase; a workgroup column = b * Cn + co)
    const bf16_t* mask_src;  // mode 1: ReLU mask source with the geometry of out, or null
    int ncob, nnb;        // column blocks of 128 per launch; mode 0: ncob = 2 row phases x nnb
};
bool igemm_ct_supports(int mode, int N, int H, int W, int Ca, int Cn);
hipError_t igemm_ct_launch(int mode, const IgCtParams& p, int grid_x, hipStream_t st);
// conv_first.hip: forward of the 16-channel first convolution, weights in registers, pixels straight from global memory
hipError_t conv_first_fwd_launch(const void* in16, const void* wp, int ntiles_w, const float* bias, void* y, int N, int H, int W, int Cout, int dil,
                                 int relu, int ncu, hipStream_t st, const float* x = nullptr, const float* cw = nullptr, const float* cb = nullptr);

// ---------------------------------------------------------------------------------------------
// igemm_wgrad: slab[z][tap][cs_off+cs][cf] = sum_{pix in split z} S[n][pix*stride + tap*dil][cs] * F[n][pix][cf]
//   conv3x3 backward-weight : F = dz (cf = co), S = layer input (cs = ci), 9 taps
//   convT 2x2 backward-weight: F = x (cf = ci), S = dy (cs = co), 4 taps, stride 2
// ---------------------------------------------------------------------------------------------
struct IgWgradParams {
    const bf16_t* F;
    int Hf, Wf, Cf;     // F tensor [N][Hf][Wf][Cf]; the pixel grid of the reduction
    IgSrc S;
    float* slab;        // [nsplit][ntap][CsOut][CfOut]
    float* bslab;       // per-split column sums of F (BiasAddGrad when F = dz), or null: split z at bslab + z*slab_stride
    float* sbslab;      // per-split column sums of S over all taps (bias gradient of the transposed conv, where S = dy), or null
    long slab_stride;   // floats between the slabs of consecutive splits (>= ntap*CsOut*CfOut; the bias row may sit behind the taps)
    int CsOut, CfOut, cs_off;
    const void* zero_page;
    int N, dil, stride;
    int nsplit, ntiles_total;
    int lsw;            // log2(g.SW): tiles are aligned (SW divides the pixel tile)
    int nbuf, nsw;      // staging ring depth (2 or 3) and S pieces per wave per tile (igemm_wgrad_nsw)
    int dbg;            // developer A/B switch (RSU_WG_DBG): 1 = skip the staging loads of all but the first tile (timing only)
    TileGeo g;          // TM of this geometry = pixels per reduction tile
};
enum { IGW_CFG_64x64 = 0, IGW_CFG_64x16 = 1, IGW_CFG_128x64 = 2, IGW_NCFG = 3 };
size_t igemm_wgrad_lds_bytes(int cfg, int npix_max, int nbuf);
int igemm_wgrad_nsw(int cfg, int npix_max);
int igemm_wgrad_tmk(int cfg);
int igemm_wgrad_cfb(int cfg);  // F channels per workgroup (grid.x block)
int igemm_wgrad_csb(int cfg);  // S channels per workgroup (grid.y block)
int igemm_wgrad_kgroups(int cfg);  // slabs written per grid.z slice
hipError_t igemm_wgrad_launch(int cfg, int ntap, const IgWgradParams& p, int grid_x, int grid_y, int grid_z,
                              hipStream_t st);
// igemm_wgt.hip: weight gradient + bias gradient of the 2x2 stride-2 transposed conv as a ping-pong kernel (slabs as igemm_wgrad's:
// [4][Cout][Cin] + a row of rup(Cout, 4) bias sums per pixel split; one split writes dK / db in place)
bool igemm_wgt_supports(int N, int H, int W, int Cin, int Cout);
int igemm_wgt_blocks(int Cin, int Cout);   // workgroups per pixel split
int igemm_wgt_tiles(int N, int H, int W);  // 64-pixel tiles of the reduction
hipError_t igemm_wgt_launch(const void* x, const void* dy, float* slab, float* sbslab, long slab_stride, int N, int H, int W, int Cin, int Cout,
                            int nsplit, hipStream_t st);
// igemm_wg1.hip: weight gradient + bias gradient of the first 3x3 conv (16-channel input) as a ping-pong kernel over phase images of in16
// (slabs as igemm_wgrad's 64x16 launch: [9][16][Cout] + a bias row of Cout floats per pixel split)
bool igemm_wg1_supports(int N, int H, int W, int Cout, int dil);
int igemm_wg1_blocks(int Cout);
int igemm_wg1_tiles(int N, int Ho, int Wo);
hipError_t igemm_wg1_launch(const void* in16, const void* dz, float* slab, float* bslab, long slab_stride, int N, int H, int W, int Cout, int dil,
                            int nsplit, hipStream_t st);
// the grouped launch (igemm_wgpp.hip, igemm_wg_group_kernel): up to IGW_GROUP_MAX layers' weight gradients in one launch
enum { IGW_FAM_WGPP3 = 0, IGW_FAM_WGPP4 = 1, IGW_FAM_WGPP5 = 2, IGW_FAM_WGPP6 = 3, IGW_FAM_WGP64_4 = 4, IGW_FAM_WGP64_5 = 5,
       IGW_FAM_GENERIC = 8 /* + 2 * cfg + (ntap == 4) */ };
#define IGW_GROUP_MAX 16
#define IGW_UNITS_MAX 2048
struct IgWgJob {
    IgWgradParams p;
    int gx, gy, gz;   // units of the job: F channel blocks x S channel blocks x pixel splits (gz == p.nsplit)
    int family;       // IGW_FAM_*
    int pad_;
};
struct IgWgGroupParams {
    int njobs, nwg;
    IgWgJob job[IGW_GROUP_MAX];
    int wg_first[257];              // workgroup b runs unit[wg_first[b] .. wg_first[b+1])
    unsigned unit[IGW_UNITS_MAX];   // (job << 24) | unit index inside the job
};
int igemm_wg_group_family(int cfg, int ntap, const IgWgradParams& p);
size_t igemm_wg_group_lds_bytes(int family, const IgWgradParams& p);
#define IGW_GROUP_LDS_BYTES (160 * 1024)   // the grouped launch always takes the whole LDS: its last 256 bytes hold the loop state, the rest is the unit's
hipError_t igemm_wg_group_launch(const IgWgGroupParams* dev_table, int nwg_total, hipStream_t st);
// ping-pong wave groups (igemm_wgpp.hip): the 128x64 shape of the 3x3 stride-1 weight gradient, same slabs and bits as igemm_wgrad
bool igemm_wgpp_supports(int cfg, int ntap, const IgWgradParams& p);
bool igemm_wgp64_supports(int cfg, int ntap, const IgWgradParams& p);
hipError_t igemm_wgp64_launch(const IgWgradParams& p, int gx, int gy, int gz, hipStream_t st);
hipError_t igemm_wgpp_launch(const IgWgradParams& p, int grid_x, int grid_y, int grid_z, hipStream_t st);
