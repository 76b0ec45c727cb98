/*
 * rsu.h -- C ABI of the MI355X-native U-Net hot path (librsu_hip.so).
 *
 * The reference (aschneuw/road-segmentation-unet) is pure Python over TensorFlow 1.4; it has no
 * FFI of its own. The operator boundary this library replaces is the set of TF op kernels that
 * /root/reference/src/unet.py:12-97 and src/tf_aerial_images.py:103-122,147-149 instantiate, plus
 * the numpy tiler of src/images.py. Every entry point below cites the reference call site it
 * stands in for. INTEGRATION.md shows the ctypes binding a maintainer of the reference would add.
 *
 * Conventions
 *   - plain C: device pointers + sizes, no torch / C++ types. `stream` is a hipStream_t passed
 *     as void* (NULL = the default stream). All calls are asynchronous on that stream and
 *     re-entrant per stream. Every MFMA launch takes the compute units it may plan for as its own `ncu` argument; nothing has
 *     to be set between launches. Process-wide state (advisory, none of it changes a result): the DEFAULT budget behind ncu = 0
 *     (rsu_set_cu_budget), the tile-shape tuning mode and its table of measured choices (rsu_set_autotune: launches look shapes
 *     up and never measure unless the host switches to RSU_TUNE_MEASURE for an explicit tuning pass), one page of zeros per device.
 *   - devices: one process drives one GPU (torch.distributed, one rank per device). The HIP current device must be the device
 *     the stream and the pointers belong to (the Python host calls torch.cuda.set_device); tensors stay below 2 GiB each
 *     (RSU_E2BIG otherwise: split the batch).
 *   - activations: NHWC, bfloat16 ("bf16", the storage type of the fast path), raw uint16 bits.
 *     Channel counts of bf16 tensors must be multiples of 8 (16-byte pieces).
 *   - parameters / gradients / optimizer state: float32 in the reference's own layouts: conv
 *     kernels HWIO [kh][kw][Cin][Cout] (tf.layers.conv2d), transposed-conv kernels
 *     [kh][kw][Cout][Cin] (tf.layers.conv2d_transpose).
 *   - MFMA kernels read bf16 copies of the weights in "fragment order" produced by the
 *     rsu_pack_* calls (re-pack after every optimizer step).
 *   - return value: 0 on success, negative errno-style code otherwise (never throws).
 */
#ifndef RSU_H_
#define RSU_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RSU_OK 0
#define RSU_EINVAL (-22)  /* bad geometry / unsupported shape (the reference raises AssertionError) */
#define RSU_ENOMEM (-12)  /* workspace too small */
#define RSU_EHIP (-5)     /* a HIP runtime call failed; see rsu_last_hip_error() */
#define RSU_E2BIG (-7)    /* a tensor of this call reaches 2 GiB (32-bit byte offsets inside the kernels): use a smaller batch */

typedef void* rsu_stream_t; /* hipStream_t */

/* One input of a (virtually concatenated) convolution: a window of an NHWC bf16 tensor.
 * unet.py:70-85 crops the skip tensor(s) to the up-conv size and concatenates [skip,(dil skip),up]
 * on the channel axis; this library never materialises that tensor: the conv reads up to three
 * sources, each with its own crop offset (oy, ox) = ((H - h) / 2, (W - w) / 2). */
typedef struct {
    const void* ptr; /* bf16 [N][H][W][C] */
    int H, W, C;
    int oy, ox; /* origin of the window inside the tensor */
} rsu_src_t;

/* ---- library / device ------------------------------------------------------------------- */
const char* rsu_version(void);
int rsu_last_hip_error(void);
/* Compute units a persistent MFMA launch plans for (one workgroup each). New here (the reference is single-device). Every
 * MFMA entry point below takes it as its `ncu` argument, 32..256 (256 = the whole MI355X); ncu = 0 selects the process default
 * set here (256 unless changed). A data-parallel host leaves some CUs to the RCCL channel workgroups of an overlapped gradient
 * all-reduce; the Python host (unet.py, RSU_SPLIT_CHIP) gives the backward-data launches of its main stream and the
 * weight-gradient launches of its side stream disjoint shares of the chip, so that one kernel of each kind is resident at a
 * time. For the weight gradients ncu sets the number of partial sums per output tile: it changes their summation order
 * (results agree to fp32 rounding); for every other launch it only changes speed. */
int rsu_set_cu_budget(int ncu);
int rsu_get_cu_budget(void);
/* Tile shapes of the conv launches by measurement. All shapes give bit-identical results; the mode only moves time.
 *   RSU_TUNE_OFF      the cost model decides (RSU_AUTOTUNE=0 in the environment forces this);
 *   RSU_TUNE_LOOKUP   (default) a launch uses the measured shape of its (geometry, flags, ncu) if the table holds one, else the
 *                     model's; it never measures, so the launch entry points never synchronise the device;
 *   RSU_TUNE_MEASURE  a launch whose key is missing times every admissible shape on an idle device (synchronises it) and keeps
 *                     the fastest if it beats the model's choice by 3 %. The host switches this on around ONE explicit, untimed
 *                     pass over its network (UNet.tune) -- under data parallelism before any collective is in flight.
 * rsu_autotune_entries: geometries measured so far. */
#define RSU_TUNE_OFF 0
#define RSU_TUNE_LOOKUP 1
#define RSU_TUNE_MEASURE 2
int rsu_set_autotune(int mode);
int rsu_get_autotune(void);
int rsu_autotune_entries(void);
/* The table of measured choices as rows of 17 ints (opaque key words + choice): export after a run, import before another (a
 * profiling run then issues no timing launches). export returns the number of entries (rows may be NULL to count); import
 * returns the number of rows it took: rows naming a tile shape or kernel generation this build does not have are skipped (the
 * host also stamps its file with the library's hash, bench.py). */
int rsu_autotune_export(int* rows, int capacity);
int rsu_autotune_import(const int* rows, int nrows);
/* unet.py:100-115 input_size_needed(output_size, num_layers). RSU_EINVAL where the reference asserts. */
int rsu_input_size_needed(int output_size, int num_layers, int* input_size);

/* ---- weight packing (float32 reference layout -> bf16 MFMA fragment order) -------------- */
/* Bytes of a packed buffer: taps * roundup(sum_i roundup(seg_c[i],32)) * roundup(rows,64) * 2 */
size_t rsu_packed_bytes(int taps, int rows, const int* seg_c, int nseg);
/* conv forward (unet.py:34-45,88-91): A[tap][co][ci]; seg_c = channel count of each concat source */
int rsu_pack_conv_fwd(const float* w_hwio, void* packed, int k, int Cin, int Cout, const int* seg_c, int nseg,
                      rsu_stream_t stream);
/* conv backward-data (Conv2DBackpropInput): A[flipped tap][ci][co] for the input channels
 * [ci_off, ci_off+ci_cnt) of a kernel with Cin_total inputs (one pack per concat source) */
int rsu_pack_conv_bwd(const float* w_hwio, void* packed, int k, int Cin_total, int ci_off, int ci_cnt, int Cout,
                      rsu_stream_t stream);
/* transposed conv forward (unet.py:67-68): four 1x1 matrices A[a*2+b][co][ci] */
int rsu_pack_convT_fwd(const float* k_hwoi, void* packed, int Cin, int Cout, rsu_stream_t stream);
/* transposed conv backward-data: A[a*2+b][ci][co] */
int rsu_pack_convT_bwd(const float* k_hwoi, void* packed, int Cin, int Cout, rsu_stream_t stream);

/* Batched packing: every weight tensor of the network re-packed by ONE kernel launch (the per-tensor calls above cost a
 * launch each, ~64 per optimizer step). Fill a host table with rsu_pack_table_add (same arguments as the per-tensor
 * calls, selected by `kind`), copy it to the device once (the pointers are static), then call rsu_pack_table_run after
 * every optimizer step. Returns the number of table entries used by the tensor (4 for RSU_PACK_CONVT_FWD, else 1). */
#define RSU_PACK_CONV_FWD 0
#define RSU_PACK_CONV_BWD 1
#define RSU_PACK_CONVT_FWD 2
#define RSU_PACK_CONVT_BWD 3
#define RSU_PACK_CONV_FIRST 4
size_t rsu_pack_table_entry_bytes(void);
int rsu_pack_table_add(void* host_table, int index, int kind, const float* w, void* packed, int k, int Cin_total, int ci_off,
                       int ci_cnt, int Cout, const int* seg_c, int nseg);
int rsu_pack_table_finish(void* host_table, int nentries, int* total_blocks);
int rsu_pack_table_run(const void* dev_table, int nentries, int total_blocks, rsu_stream_t stream);

/* ---- network head / tail (VALU kernels) -------------------------------------------------- */
/* Dropout (unet.py:29-30, 64-65: tf.nn.dropout(net, keep) = net / keep * floor(keep + U[0,1))) is fused into the producer of
 * every tensor it applies to. `keep` in (0, 1]; 1.0 = identity (inference, --dropout=1.0). U is a counter-based hash of
 * (`key`, NHWC element index) -- TensorFlow's Philox stream cannot be reproduced; the parity tests restate the hash bit
 * for bit on the CPU. The host picks one key per dropout site and step. The backward kernels take the same (keep, key). */
/* unet.py:22-23  net = conv1x1(X - 0.5) (color_space_adjust), then the level-0 dropout. x: f32 [npix][3].
 * out16: bf16 [npix][16] = {dropout(net0)[0..2], 0, m[cj]*(x-0.5)[ci] at 4+3*ci+cj, m[0..2] at 13..15} (m = 0/1 keep
 * mask, all ones without dropout) -- channels 0..2 feed the first 3x3 conv, channels 4..15 are kept for the weight and
 * bias gradients of color_space_adjust (rsu_conv_first_bwd_weight). */
int rsu_color_adjust_fwd(const float* x, const float* w, const float* b, void* out16, long npix, float keep, unsigned key,
                         rsu_stream_t stream);
/* unet.py:34-35,42-43 first 3x3 conv of level 0 (Cin = 3) + bias + ReLU, dil = 1 or 2 (dilated branch).
 * in16 as above [N][H][W][16]; packed = rsu_pack_conv_first(w f32 HWIO [3][3][3][Cout]) (rows for channels 3..15 are
 * zero, so the (x-0.5) copy in channels 4..6 does not contribute); y bf16 [N][H-2d][W-2d][Cout]. */
size_t rsu_packed_first_bytes(int Cout);
int rsu_pack_conv_first(const float* w_hwio, void* packed, int Cout, rsu_stream_t stream);
int rsu_conv_first_fwd(const void* in16, const void* packed, const float* b, void* y, int N, int H, int W, int Cout,
                       int dil, int ncu, rsu_stream_t stream);
/* unet.py:22-23 AND unet.py:34-35,42-43 in one launch (no dropout: keep == 1): y = relu(conv3x3(conv1x1(x - 0.5, w0) + b0, W1, dilation) + b) straight from
 * the f32 input x [N][H][W][3]; w0 f32 [3][3] ([ci][cj]), b0 f32 [3] (device pointers: color_space_adjust/kernel, /bias); packed, b, y as
 * rsu_conv_first_fwd. The 16-channel tensor of rsu_color_adjust_fwd is neither written nor read (-58 MB and one launch per forward pass at
 * B = 4, 572 px); the results are bit-identical to rsu_color_adjust_fwd(keep = 1) + rsu_conv_first_fwd. A training step still calls
 * rsu_color_adjust_fwd (rsu_conv_first_bwd_weight reads its output), but may do so on another stream, off the forward pass's critical path. */
int rsu_color_conv_first_fwd(const float* x, const float* w0, const float* b0, const void* packed, const float* b, void* y, int N, int H,
                             int W, int Cout, int dil, int ncu, rsu_stream_t stream);
/* weight/bias gradients of that conv and, through it, of color_space_adjust (no input-gradient pass is needed):
 * dw1 [3][3][3][Cout]; gx [9][12][Cout]: rows 0..8 gxc[t][3*ci+cj][co] = sum_pix m[pix+t][cj] (x-0.5)[pix+t][ci] dz[pix][co],
 * rows 9..11 gm[t][cj][co] = sum_pix m[pix+t][cj] dz[pix][co]. With W1 = this conv's kernel:
 *   d(color_space_adjust/kernel)[ci][cj] = 1/keep * sum_{t,co} W1[t][cj][co] gxc[t][3*ci+cj][co]
 *   d(color_space_adjust/bias)[cj]       = 1/keep * sum_{t,co} W1[t][cj][co] gm[t][cj][co]
 * ws: float workspace of rsu_conv_first_bwd_ws_floats() floats. */
size_t rsu_conv_first_bwd_ws_floats(int Cout);
/* db (optional): BiasAddGrad of this conv, computed by the same launch */
int rsu_conv_first_bwd_weight(const void* in16, const void* dz, float* dw1, float* gx, float* db, float* ws, int N,
                              int H, int W, int Cout, int dil, int ncu, rsu_stream_t stream);
/* The two sums above in one launch: dW0 f32 [3][3] ([ci][cj]) and db0 f32 [3] from gx [9][12][Cout] and W1 f32 HWIO
 * [3][3][3][Cout]; scale = 1/keep. accumulate != 0 adds to dW0/db0 (the dilated twin conv_dilut_0 shares color_space_adjust). */
int rsu_color_adjust_bwd(const float* gx, const float* w1, float* dW0, float* db0, int Cout, float scale, int accumulate,
                         rsu_stream_t stream);
/* unet.py:95 weight_output 1x1 conv (C -> 2) fused with tf_aerial_images.py:147-148 softmax[...,1].
 * act bf16 [npix][C]; w f32 [C][2]; prob f32 [npix]; logits f32 [npix][2] or NULL (unet.forward's return value). */
int rsu_head_fwd(const void* act, const float* w, const float* b, float* prob, float* logits, long npix, int C,
                 rsu_stream_t stream);
/* Same plus tf_aerial_images.py:103-110 mean sparse softmax cross-entropy and its backward:
 * labels int64 [npix] in {0,1}; loss_sum f32[1] (+= sum of per-pixel losses; caller zeroes it);
 * dact bf16 [npix][C] = gradient wrt the PRE-activation of the last conv2 (ReLU mask applied),
 * scaled by inv_count (1 / global pixel count); dw f32 [C][2], db f32 [2] (overwritten).
 * ws: rsu_head_ws_floats(npix, C) floats. */
size_t rsu_head_ws_floats(long npix, int C);
int rsu_head_fwd_bwd(const void* act, const float* w, const float* b, const int64_t* labels, float* prob,
                     float* loss_sum, void* dact, float* dw, float* db, float* ws, long npix, int C,
                     float inv_count, rsu_stream_t stream);

/* ---- 3x3 convolution, MFMA implicit GEMM -------------------------------------------------- */
/* unet.py:34-39,42-45,88-91: y = relu(conv3x3_valid(concat(srcs), W, dilation) + b).
 * All sources share the window size (Hin, Win); y is bf16 [N][Hin-2d][Win-2d][Cout]. */
int rsu_conv2d_fwd(const rsu_src_t* srcs, int nsrc, const void* packed_fwd, const float* bias, void* y, int N,
                   int Hin, int Win, int Cout, int dil, int relu, int ncu, rsu_stream_t stream);
/* The same three launches (rsu_conv2d_fwd, rsu_conv2d_fwd_pool, rsu_conv2d_bwd_data) with a caller-owned workspace `kws` of `kws_floats`
 * floats (rsu_conv_splitk_ws_floats() is always enough; NULL / 0 = the plain entry points). A layer with far fewer (pixel tile, channel
 * block) pairs than CUs -- the deep levels of the U-Net at small batches: 18x18 .. 66x66 pixels, 512 .. 2048 channels -- may then cut its
 * reduction (taps x input channels) into up to 16 slices, one workgroup per (tile, block, slice): the slices' fp32 partial sums go to
 * `kws` and a second launch sums them IN SLICE ORDER (deterministic), adds nothing else (the bias rides in slice 0), applies ReLU /
 * the ReLU mask and stores bf16. Results equal the unsplit launch up to the association of the fp32 sum over the slices. Whether a launch
 * splits, and into how many slices, is a pure function of its geometry (N, output size, channel counts, dilation) -- never of the CU
 * budget, the tuning table or a measurement: the same launch sums in the same order in every schedule and on every box (RSU_KSPLIT=0:
 * never split). The geometry includes the batch: a layer splits into more slices at N = 1 than at N = 4, so the deep layers of one image
 * associate their fp32 sums differently in batches of different size (the distance is pinned per layer in tests/test_gpu_ops.py; making
 * the rule batch-independent cost the N = 4 step 12-14 %, profiles/r05/abenv_perimg_*.txt). The workspace must not be shared with a launch
 * running concurrently on another stream: the host keeps one per stream that issues conv launches. */
size_t rsu_conv_splitk_ws_floats(void);
int rsu_conv2d_fwd_k(const rsu_src_t* srcs, int nsrc, const void* packed_fwd, const float* bias, void* y, int N,
                     int Hin, int Win, int Cout, int dil, int relu, int ncu, float* kws, size_t kws_floats, rsu_stream_t stream);
int rsu_conv2d_fwd_pool_k(const rsu_src_t* srcs, int nsrc, const void* packed_fwd, const float* bias, void* y, void* pooled, void* code,
                          int N, int Hin, int Win, int Cout, float keep, unsigned key, int ncu, float* kws, size_t kws_floats, rsu_stream_t stream);
int rsu_conv2d_bwd_data_k(const void* dz, const void* packed_bwd, void* dx, const void* relu_src, int accumulate, int N, int H, int W,
                          int Cin_total, int ci_off, int ci_cnt, int Cout, int dil, int ncu, float* kws, size_t kws_floats, rsu_stream_t stream);
/* unet.py:44-52 in one launch: y = relu(conv3x3_valid(concat(srcs), W) + b) (dilation 1), pooled = max_pooling2d(y, 2, 2) followed by
 * the next level's dropout (keep, key as rsu_maxpool2x2_fwd), and -- code != NULL -- the code bytes of rsu_maxpool2x2_fwd_code.
 * With code != NULL the conv's output size must be even (RSU_EINVAL before anything is launched otherwise); without code bytes odd
 * sizes pool with floor semantics, as max_pooling2d does. Where a tile shape with whole 2x2 windows per wavefront fits the layer, the
 * size is even and keep == 1 the pool is part of the conv kernel's epilogue (lane shuffles on the packed results; the activation is not
 * read back from HBM); otherwise the call issues the two launches itself. Same bits either way. */
int rsu_conv2d_fwd_pool(const rsu_src_t* srcs, int nsrc, const void* packed_fwd, const float* bias, void* y, void* pooled, void* code,
                        int N, int Hin, int Win, int Cout, float keep, unsigned key, int ncu, rsu_stream_t stream);
/* Conv2DBackpropInput for input channels [ci_off, ci_off+ci_cnt) of a conv with Cin_total inputs:
 * dx bf16 [N][H][W][ci_cnt] (H, W = conv input size), dz bf16 [N][H-2d][W-2d][Cout].
 * relu_src (optional, same shape as dx): dx *= (relu_src > 0)  -- ReluGrad of the producing layer.
 * accumulate != 0: dx += result (two consumers of one tensor, unet.py:32-45). */
int rsu_conv2d_bwd_data(const void* dz, const void* packed_bwd, void* dx, const void* relu_src, int accumulate,
                        int N, int H, int W, int Cin_total, int ci_off, int ci_cnt, int Cout, int dil, int ncu,
                        rsu_stream_t stream);
/* Conv2DBackpropFilter for the input channels held by `src` (rows [ci_off, ci_off+src.C) of dw):
 * dw f32 HWIO [3][3][Cin_total][Cout] (only those rows are written). (Ho, Wo) = size of dz.
 * ws: rsu_conv2d_bwd_weight_ws_floats() floats of scratch (split-K slabs). */
size_t rsu_conv2d_bwd_weight_ws_floats(int Cin_total, int src_C, int Cout);
/* db (optional, f32 [Cout]): BiasAddGrad = sum over pixels of dz, computed by the same launch (one extra MFMA per
 * 32-pixel step); pass it with ONE of the sources of a concatenated input, NULL with the others. */
int rsu_conv2d_bwd_weight(const rsu_src_t* src, const void* dz, float* dw, float* db, float* ws, int N, int Ho, int Wo,
                          int Cin_total, int ci_off, int Cout, int dil, int ncu, rsu_stream_t stream);
/* Conv2DBackpropFilter of one concat source AND tf_aerial_images.py:120-121 ApplyMomentum + the re-pack of the MFMA copies for the kernel
 * rows that source owns, in the launch that sums the weight-gradient slabs (new; single-device training only: under data parallelism the
 * gradient all-reduce sits between the two). Arguments as rsu_conv2d_bwd_weight, then: update_entry = HOST pointer to the table entry
 * rsu_update_table_add made for this kernel (kind RSU_PACK_CONV_FWD, g = dw; its packed_bwd buffers must NOT be the ones a backward-data
 * launch of the same step may still be reading on another stream: the host keeps two sets and swaps them per step), seg = index of
 * `src` among the kernel's concat sources, (lr, mu, gscale) as rsu_update_table_run. The slabs are summed in the order of the plain entry
 * point and every element sees the arithmetic of rsu_update_table_run: w, acc and the packed copies equal
 * rsu_conv2d_bwd_weight -> rsu_update_table_run bit for bit. keep_grad != 0: dw is written as well (it always is when the launch needs no
 * slabs); db (optional) is written as by rsu_conv2d_bwd_weight -- biases are updated by the host's table of the remaining variables. */
int rsu_conv2d_bwd_weight_update(const rsu_src_t* src, const void* dz, float* dw, float* db, float* ws, int N, int Ho, int Wo,
                                 int Cin_total, int ci_off, int Cout, int dil, int ncu, const void* update_entry, int seg, float lr,
                                 float mu, float gscale, int keep_grad, rsu_stream_t stream);
/* Grouped weight gradients (new; TensorFlow's executor runs independent Conv2DBackpropFilter ops side by side, tf_aerial_images.py:120-121).
 * A weight-gradient launch that has the chip to itself writes one fp32 partial result per workgroup -- 75 MB of slabs per layer on
 * 256 CUs, read back by a reduce launch -- and a layer with few pixels cannot fill the chip. A GROUP runs up to
 * RSU_WGRAD_GROUP_MAX layers in ONE launch: every layer gets a share of the chip in proportion to its work, the shallow layers are
 * split over few workgroups (the slabs of the whole group are about one per CU), the deep ones not at all (their workgroups write the
 * gradient in place), and ONE reduce launch finishes all splits. Results equal those of the single launches up to the summation
 * order of the pixel splits (fp32 rounding), and are deterministic for a given (jobs, N, ncu).
 * Usage (the pattern of rsu_pack_table_*): fill a host table with rsu_wgrad_group_plan (the pointers are static for a network, so this
 * happens once per group and CU budget), copy its rsu_wgrad_group_table_bytes() bytes to the device once, then call
 * rsu_wgrad_group_run after every backward pass. job.kind selects the fields' meaning:
 *   RSU_WGRAD_CONV3X3  as rsu_conv2d_bwd_weight: src = the input window, dz [N][Ho][Wo][Cout], dw HWIO rows [ci_off, ci_off+src.C), db optional
 *   RSU_WGRAD_CONVT2X2 as rsu_convT2x2_bwd_weight: src = {x, H, W, Cin, 0, 0}, dz = dy [N][2H][2W][Cout], dw = dK, db optional
 * ws: rsu_wgrad_group_ws_floats() floats of scratch, owned by the group between a run and the end of its stream work. */
#define RSU_WGRAD_CONV3X3 0
#define RSU_WGRAD_CONVT2X2 1
#define RSU_WGRAD_GROUP_MAX 16
typedef struct {
    int kind;
    rsu_src_t src;
    const void* dz;
    float* dw;
    float* db;
    int Ho, Wo;                         /* RSU_WGRAD_CONV3X3: size of dz */
    int Cin_total, ci_off, Cout, dil;   /* RSU_WGRAD_CONVT2X2 reads Cout only */
} rsu_wgrad_job_t;
size_t rsu_wgrad_group_table_bytes(void);
size_t rsu_wgrad_group_ws_floats(void);
int rsu_wgrad_group_plan(const rsu_wgrad_job_t* jobs, int njobs, float* ws, int N, int ncu, void* host_table);
int rsu_wgrad_group_run(const void* host_table, const void* dev_table, rsu_stream_t stream);
/* BiasAddGrad: db[c] = sum over npix of dz[pix][c]. ws: rsu_bias_grad_ws_floats(npix, C) floats. */
size_t rsu_bias_grad_ws_floats(long npix, int C);
int rsu_bias_grad(const void* dz, float* db, float* ws, long npix, int C, rsu_stream_t stream);

/* ---- 2x2 max pool -------------------------------------------------------------------------- */
/* unet.py:52 max_pooling2d (2,2)/(2,2) VALID, then the next level's dropout (unet.py:29-30; keep = 1: none).
 * x bf16 [N][H][W][C] -> y [N][H/2][W/2][C] */
int rsu_maxpool2x2_fwd(const void* x, void* y, int N, int H, int W, int C, float keep, unsigned key,
                       rsu_stream_t stream);
/* ... and, for training, a code tensor for the gradient junction below (H, W even; code [N][H/2][W/2][C] bytes, may be NULL):
 * per pooled element bits 0-3 = (window element 2*dy+dx > 0), bits 4-5 = the window's first maximum in row-major order. */
int rsu_maxpool2x2_fwd_code(const void* x, void* y, void* code, int N, int H, int W, int C, float keep, unsigned key,
                            rsu_stream_t stream);
/* Gradient junction at an encoder output y_act (ReLU output, bf16 [N][H][W][C]):
 *   g = MaxPoolGrad(y_act, dropout_grad(dpool))  (dpool bf16 [N][H/2][W/2][C], may be NULL; (keep, key) as in
 *                                             rsu_maxpool2x2_fwd: dpool is the gradient of the DROPPED pooled tensor)
 *     + zero-pad(dskip)                      (dskip bf16 [N][Hs][Ws][C] centred, may be NULL;
 *                                             adjoint of the centre crop of unet.py:70-83)
 *   dz = g * (y_act > 0)                     (ReluGrad) -> bf16 [N][H][W][C]
 * MaxPoolGrad routes to the first maximum of the window in row-major order. */
int rsu_pool_skip_relu_bwd(const void* y_act, const void* dpool, const void* dskip, void* dz, int N, int H, int W,
                           int C, int Hs, int Ws, float keep, unsigned key, rsu_stream_t stream);
/* The same junction from the code tensor of rsu_maxpool2x2_fwd_code instead of the activation (y_act may then be NULL): one
 * byte per pooled element instead of four bf16 activations, the same bits. */
int rsu_pool_skip_relu_bwd_code(const void* y_act, const void* code, const void* dpool, const void* dskip, void* dz, int N,
                                int H, int W, int C, int Hs, int Ws, float keep, unsigned key, rsu_stream_t stream);
/* unet.py:64-65 dropout in front of a transposed conv: y = x / keep * floor(keep + U), bf16 [n] -> bf16 [n] (n % 8 == 0).
 * Its backward is fused into rsu_convT2x2_bwd_data: pass y as relu_src (y > 0 <=> x > 0 and kept) and out_scale = 1/keep. */
int rsu_dropout_fwd(const void* x, void* y, long n, float keep, unsigned key, rsu_stream_t stream);

/* ---- 2x2 stride-2 transposed convolution (unet.py:67-68) ---------------------------------- */
int rsu_convT2x2_fwd(const void* x, const void* packed_fwd, const float* bias, void* y, int N, int H, int W,
                     int Cin, int Cout, int ncu, rsu_stream_t stream);
/* dx bf16 [N][H][W][Cin] = out_scale * sum_{a,b,co} dy[2i+a][2j+b][co] K[a][b][co][ci], times (relu_src > 0) if given */
int rsu_convT2x2_bwd_data(const void* dy, const void* packed_bwd, void* dx, const void* relu_src, float out_scale,
                          int N, int H, int W, int Cin, int Cout, int ncu, rsu_stream_t stream);
/* dK f32 [2][2][Cout][Cin] and, when db != NULL, db f32 [Cout] = sum over all pixels of dy (BiasAddGrad of the transposed
 * conv, unet.py:72) from the same launch; ws: rsu_convT2x2_bwd_weight_ws_floats() floats */
size_t rsu_convT2x2_bwd_weight_ws_floats(int Cin, int Cout);
int rsu_convT2x2_bwd_weight(const void* x, const void* dy, float* dK, float* db, float* ws, int N, int H, int W,
                            int Cin, int Cout, int ncu, rsu_stream_t stream);

/* ---- optimizer (tf_aerial_images.py:116-121, MomentumOptimizer, use_nesterov=False) ------- */
/* acc = mu*acc + gscale*g ; w -= lr*acc. gscale folds the 1/world_size of data-parallel averaging. */
int rsu_momentum_step(float* w, float* acc, const float* g, float lr, float mu, float gscale, long n,
                      rsu_stream_t stream);

/* The Momentum step of EVERY live variable and the re-pack of the MFMA copies in ONE launch (new): rsu_momentum_step moves 20 B per
 * parameter and the batched re-pack reads every conv kernel twice more; this pass reads w, acc, g once, writes w, acc and both packed
 * layouts: 24 B per weight. Same arithmetic per element as rsu_momentum_step, same packed bits as rsu_pack_*. Usage (the pattern of
 * rsu_pack_table_*): one table entry per variable -- rsu_update_table_add for a conv / transposed-conv kernel with its packed buffers
 * (kind RSU_PACK_CONV_FWD: packed_fwd + one backward-data pack per concat source in packed_bwd[0..nseg), or packed_bwd = NULL for a
 * forward-only net; RSU_PACK_CONVT_FWD: packed_fwd (four matrices) + packed_bwd[0]; RSU_PACK_CONV_FIRST: packed_fwd only),
 * rsu_update_table_add_plain for everything no MFMA kernel reads (biases, colour adjust, the 1x1 head; w / acc / g 16-byte aligned) --
 * then rsu_update_table_finish, copy the table to the device once, and rsu_update_table_run after every backward pass. */
size_t rsu_update_table_entry_bytes(void);
int rsu_update_table_add_plain(void* host_table, int index, float* w, float* acc, const float* g, long n);
int rsu_update_table_add(void* host_table, int index, int kind, float* w, float* acc, const float* g, void* packed_fwd,
                         void* const* packed_bwd, int Cin_total, int Cout, const int* seg_c, int nseg);
int rsu_update_table_finish(void* host_table, int nentries, int* total_blocks);
int rsu_update_table_run(const void* dev_table, int nentries, int total_blocks, float lr, float mu, float gscale,
                         rsu_stream_t stream);

/* ---- patch / stride tiler (src/images.py) -------------------------------------------------- */
/* images.py:269-281 mirror_border + :35-85 extract_patches fused, on device: tile t (x-outer,
 * y-inner order, images.py:76-77) of image n is the [S][S] window of the symmetric-padded image at
 * origin (x0, y0) = ((t / pps) * stride, (t % pps) * stride). imgs f32 [nimg][H][H][3];
 * tiles f32 [ntiles][S][S][3] for tile indices [t0, t0+ntiles) of the flattened (img, t) list. */
int rsu_extract_tiles(const float* imgs, float* tiles, int nimg, int H, int S, int P, int stride, long t0,
                      long ntiles, rsu_stream_t stream);
/* images.py:131-164 images_from_patches, accumulation half: acc[n][y0+i][x0+j] += prob[t][i][j],
 * hits += 1 (f32 accumulators [nimg][H][H]; the division is rsu_overlap_finish). */
int rsu_overlap_add(const float* prob, float* acc, float* hits, int nimg, int H, int P, int stride, long t0,
                    long ntiles, rsu_stream_t stream);
int rsu_overlap_finish(const float* acc, const float* hits, float* out, long n, rsu_stream_t stream);

/* ---- post-processing wire format (src/images.py) and metric counters (src/summary.py) ------ */
/* images.py:256-266 quantize_mask: per patch_size block of masks f32 [nimg][S][S] (channel axis squeezed), label =
 * mean(mask >= 0.5) > threshold, written over the block of `out` (out == masks is allowed: a block is read completely before it
 * is written). patch_size <= 64. */
int rsu_quantize_mask(const float* masks, float* out, int nimg, int S, int patch_size, float threshold, rsu_stream_t stream);
/* images.py:88-99 labels_for_patches over images.py:35-85 extract_patches(masks, patch_size): labels int64
 * [nimg][S/ps][S/ps] in the reference's patch order (x outer, y inner), label = mean(patch) > threshold. */
int rsu_labels_for_patches(const float* masks, int64_t* labels, int nimg, int S, int patch_size, float threshold,
                           rsu_stream_t stream);
/* summary.py:141-147 tf.metrics.accuracy / recall / precision keep running counts: counts[0..3] += TP, FP, FN, TN of n
 * int64 {0,1} labels (caller zeroes `counts` where the reference runs tf.local_variables_initializer()). */
int rsu_confusion_counts(const int64_t* predictions, const int64_t* labels, long n, unsigned long long* counts,
                         rsu_stream_t stream);

/* ---- static shape table (tf_aerial_images.py:133-145 build_graph fixes every shape up front) ---------- */
#define RSU_OP_COLOR_ADJUST 0 /* unet.py:22-23 */
#define RSU_OP_CONV3X3 1      /* unet.py:34-45, 88-91 (dilation 1 or 2; nsrc > 1: virtually concatenated crops) */
#define RSU_OP_MAXPOOL 2      /* unet.py:52 */
#define RSU_OP_CONVT2X2 3     /* unet.py:67-68 */
#define RSU_OP_HEAD 4         /* unet.py:95 + tf_aerial_images.py:147-148 */
typedef struct {
    int kind, level;          /* RSU_OP_*; block index as in the reference's scope names (conv_<level>) */
    int Hin, Win, Cin;        /* input window (of every source) and total input channels */
    int Hout, Wout, Cout;
    int dilation, nsrc;
} rsu_plan_row_t;
typedef struct {
    int input_size;           /* unet.input_size_needed(patch_size, num_layers) */
    long num_params;          /* every variable unet.forward creates (the dead level L-1 dilated pair included) */
    long activation_elems;    /* bf16 elements of all op outputs for `batch` patches */
    size_t workspace_floats;  /* largest scratch any backward entry point of this network asks for */
} rsu_plan_totals_t;
/* Ops of unet.forward in execution order for a (num_layers, root_size, patch_size, dilated) network: a binder needs no shape
 * arithmetic of its own. rows may be NULL (count only); RSU_ENOMEM when capacity is too small (nrows still set). */
int rsu_plan(int num_layers, int root_size, int patch_size, int dilated, int batch, rsu_plan_row_t* rows, int capacity,
             int* nrows, rsu_plan_totals_t* totals);

#ifdef __cplusplus
}
#endif
#endif /* RSU_H_ */
