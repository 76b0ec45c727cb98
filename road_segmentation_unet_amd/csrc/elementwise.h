// Launch prototypes of the bandwidth-bound kernels (elementwise.hip).
#pragma once
#include "rsu_common.h"

struct PackParams {
    int nchunks, ntap, ntiles;  // packed dims: [nchunks][ntap][ntiles][64][8]
    int rows;                   // real output rows (channels)
    int nseg, seg_c[3];         // K segments (concat sources), real channel counts
    long s_tap, s_row, s_k;     // source strides (elements)
    int flip;                   // read source tap (ntap-1-tap)
};

hipError_t ew_color_adjust(const float* x, const float* w, const float* b, void* out16, long npix, float keep, unsigned key, hipStream_t st);
hipError_t ew_scale_bf16(void* x, long n, float s, hipStream_t st);
hipError_t ew_dropout(const void* x, void* y, long n, float keep, unsigned key, hipStream_t st);
hipError_t ew_scatter_first_grads(const float* tmp, float* dw1, float* gxc, int Cout, hipStream_t st);
hipError_t ew_maxpool_fwd(const void* x, void* y, void* code, int N, int H, int W, int C, float keep, unsigned key, hipStream_t st);
hipError_t ew_pool_skip_relu_bwd(const void* yact, const void* code, const void* dpool, const void* dskip, void* dz, int N, int H, int W, int C, int Hs, int Ws,
                                 float keep, unsigned key, hipStream_t st);
int ew_colsum_blocks(long npix, int C);
hipError_t ew_colsum(const void* dz, float* db, float* ws, long npix, int C, hipStream_t st);
// out2 (optional): n2 more float4 items behind the taps of every slab, reduced into out2 by the same launch
hipError_t ew_reduce_slabs(const float* slab, float* out, float* out2, int n2, int nsplit, long slab_elems, int ntap, int CsOut, int cs_off, int cs_cnt,
                           int CfOut, hipStream_t st);
// grouped form: one launch reduces the slabs of several weight-gradient jobs (device-resident job table)
struct ReduceJob {
    const float* slab; float* out; float* out2;
    long slab_elems;
    int n2, nsplit, ntap, CsOut, cs_off, cs_cnt, CfOut;
    int block_begin, wide, pad_;
};
int ew_reduce_job_blocks(ReduceJob& j);   // sets j.wide, returns the workgroups the job needs
hipError_t ew_reduce_slabs_many(const ReduceJob* jobs_dev, int njobs, int total_blocks, hipStream_t st);
int ew_head_blocks(long npix, int C);
hipError_t ew_head(bool train, const void* act, const float* w, const float* b, const int64_t* labels, float* prob, float* logits, void* dact, float* dw,
                   float* db, float* loss_sum, float* ws, long npix, int C, float inv_count, hipStream_t st);
hipError_t ew_color_adjust_bwd(const float* gx, const float* w1, float* dW0, float* db0, int Cout, float scale, int accumulate, hipStream_t st);
hipError_t ew_momentum(float* w, float* acc, const float* g, float lr, float mu, float gscale, long n, hipStream_t st);
hipError_t ew_pack(const float* src, void* dst, const PackParams& pp, hipStream_t st);
struct PackJob { PackParams pp; const float* src; bf16_t* dst; int block_start; int pad_; };
hipError_t ew_pack_many(const PackJob* jobs_dev, int njobs, int total_blocks, hipStream_t st);
int ew_pack_blocks(const PackParams& pp);
// Momentum + re-pack in one pass (k_update_pack_many): one destination of a tensor's packed copies
struct UpDest {
    bf16_t* base[3];        // orientation B with tapmode 0/1: one buffer per segment of R1 (concat source); else base[0]
    long tap_buf_stride;    // tapmode 2 (one single-tap matrix per source tap): elements between the matrices
    int orient;             // 0: rows from R2, k from R1 (segmented, 32-padded); 1: rows from R1 (of one segment), k from R2
    int ntap, tapmode;      // taps of the packed layout; 0 same tap, 1 flipped, 2 tap 0 of the tap's own matrix
    int ntiles[3];          // 16-row tiles per (chunk, tap) of the packed layout (orientation 1: of each segment's buffer)
    int chunk0[3];          // orientation 0: first 32-k chunk of each R1 segment
};
struct UpJob {
    float* w; float* acc; const float* g;
    long n;                 // kind 0: floats of the range
    int kind;               // 0 plain Momentum range, 1 packed tensor
    int ntap, R1, R2;       // source [ntap][R1][R2], R2 contiguous
    int nseg, seg_c[3], seg_r0[3], seg_blk0[3];   // R1 segments: real rows, first row, first 32-row block
    int nrb, ncb;           // 32-blocks along R1 (all segments) and along R2
    int ndest;
    UpDest d[2];
    int block_start, pad_;
};
int ew_update_job_blocks(const UpJob& j);
hipError_t ew_update_pack_many(const UpJob* jobs_dev, int njobs, int total_blocks, float lr, float mu, float gscale, hipStream_t st);
// the same pass over ONE R1 segment of one tensor, its gradient = the ordered sum of `nsplit` weight-gradient slabs (k_update_pack_seg)
hipError_t ew_update_pack_seg(const UpJob& J, int seg, const float* slab, long stride, int nsplit, float* gout, float* out2, int n2, float lr, float mu,
                              float gscale, hipStream_t st);
hipError_t ew_extract_tiles(const float* imgs, float* tiles, int H, int S, int P, int stride, int pps, long t0, long ntiles, hipStream_t st);
hipError_t ew_overlap_add(const float* prob, float* acc, float* hits, int nimg, int H, int P, int stride, int pps, long t0, long ntiles, hipStream_t st);
hipError_t ew_overlap_finish(const float* acc, const float* hits, float* out, long n, hipStream_t st);
hipError_t ew_block_label(const float* mask, float* out, int64_t* labels, int nimg, int S, int ps, float thr, int mode, hipStream_t st);
hipError_t ew_confusion(const int64_t* pred, const int64_t* truth, long n, unsigned long long* counts, hipStream_t st);
