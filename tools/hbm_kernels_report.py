#!/usr/bin/env python3
"""Achieved HBM GB/s of the bandwidth-bound kernels of the config-2 step: algorithmic bytes per step (each tensor counted once,
SURVEY section 8d) / the kernel's time per step from a rocprofv3 kernel-stats csv (tools/gpu_bench_profile.sh).
usage: python tools/hbm_kernels_report.py profiles/rNN/kernel_stats.csv [steps_in_profile] > profiles/rNN/hbm_kernels.md"""
import csv
import sys

L, ROOT, P, B = 5, 64, 388, 4
S = P + 12 * 2 ** (L - 1) - 8
path = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 15

# encoder geometry: conv2 output of level i (the pool input) and the decoder stage sizes
pool_in, h, nf = [], S, ROOT
for i in range(L - 1):
    pool_in.append(B * (h - 4) ** 2 * nf)
    h, nf = (h - 4) // 2, nf * 2
n_pool_in = sum(pool_in)
# levels whose pool still runs as a kernel of its own (round 3: levels 1 and 3 fold it into conv2's epilogue, rsu_conv2d_fwd_pool);
# overridable: POOL_LEVELS="0,1,2,3"
import os
pool_levels = [int(v) for v in os.environ.get("POOL_LEVELS", "0,2").split(",")]
n_pool_kernel = sum(pool_in[i] for i in pool_levels)
n_param = 31031822
npix_out, npix_in = B * P * P, B * S * S
alg = {  # kernel-name prefix -> (what, bytes per step)
    "_Z13k_maxpool_fwd": ("2x2 max-pool fwd (+dropout) + code byte, levels %s: 2 B read + 0.5 B + 0.25 B written per input element" % pool_levels, 2.75 * n_pool_kernel),
    "_Z16k_conv_first_fwd": ("first 3x3 conv forward (16-channel input straight from global memory, weights in registers): 32 B read per input "
                             "pixel, 128 B written per output pixel", 32.0 * npix_in + 128.0 * B * (S - 2) ** 2),
    "_Z25k_pool_skip_relu_bwd_code": ("pool bwd + skip-gradient add + ReLU mask from the code bytes: 0.25 B code + 0.5 B dpool + ~1 B dskip (the "
                                 "cropped window) read, 2 B dz written per input element", 3.8 * n_pool_in),
    "_Z18k_update_pack_many": ("Momentum step + re-pack in one pass: w, a, g read, w, a written, both bf16 packed layouts written (24 B per weight)", 24.0 * n_param),
    "_Z6k_head": ("1x1 head + softmax + CE + gradients: 128 B read + 128 B written per output pixel", 256.0 * npix_out),
    "_Z14k_color_adjust": ("centre + colour adjust: 12 B read + 32 B written per input pixel", 44.0 * npix_in),
}
rows = list(csv.DictReader(open(path)))
print("| kernel | algorithmic traffic | MB / step | us / step | achieved GB/s | of 8 TB/s |")
print("|---|---|---|---|---|---|")
for pre, (what, nbytes) in alg.items():
    t = sum(int(r["TotalDurationNs"]) for r in rows if r["Name"].startswith(pre)) / steps
    if t == 0:
        continue
    gbs = nbytes / t
    print("| `%s` | %s | %.0f | %.1f | %.0f | %.0f %% |" % (pre[pre.index("k_"):], what, nbytes / 1e6, t / 1e3, gbs, gbs / 80.0))
