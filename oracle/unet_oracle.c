/*
 * oracle/unet_oracle.c -- CPU restatement of the reference U-Net hot path.
 *
 * TEST INFRASTRUCTURE ONLY. Nothing in the product path (road_segmentation_unet_amd/) may
 * import, link or call this file; only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg use it, and only as the checker / the reported CPU baseline.
 *
 * PARITY STATUS: "parity unpinned" at the TensorFlow boundary. The reference's arithmetic for
 * this path lives in the un-vendored third-party wheel tensorflow==1.4.0
 * (/root/reference/requirements.txt:18-19), which is not installable in this image, and the
 * reference holds no golden vectors for the network (src/test_images.py never imports unet).
 * The functions below restate the published TF-1.4 op semantics that the reference's call
 * sites select; each cites its call site. They are cross-checked against stock PyTorch-CPU
 * float64 ops in tests/test_oracle_vs_torch.py (an independent second opinion, not the
 * reference itself).
 *
 * Conventions: activations NHWC float32, conv kernels HWIO [kh][kw][Cin][Cout]
 * (tf.layers.conv2d), transposed-conv kernels [kh][kw][Cout][Cin] (tf.layers.conv2d_transpose),
 * VALID padding, cross-correlation (no kernel flip). Every output element is accumulated in
 * double and rounded once to float.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define IDX4(n, y, x, c, H, W, C) ((((size_t)(n) * (H) + (y)) * (W) + (x)) * (C) + (c))

/* round-to-nearest-even float -> bfloat16 -> float; used by the bf16-storage emulation mode of
 * the Python composition (oracle/unet_oracle.py) so that bf16 HIP kernels can be checked tightly. */
static inline float bf16_round(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return f; /* NaN stays NaN */
    u += 0x7fffu + ((u >> 16) & 1u);
    u &= 0xffff0000u;
    memcpy(&f, &u, 4);
    return f;
}

void orc_round_bf16(float* x, long n) {
#pragma omp parallel for schedule(static)
    for (long i = 0; i < n; ++i) x[i] = bf16_round(x[i]);
}

/* unet.py:22-23  net = X - 0.5 ; conv2d(net, 3, (1,1), name="color_space_adjust")
 * w is [1][1][Cin][Cout] HWIO; generic in Cin/Cout so weight_output (unet.py:95) reuses it with sub=0. */
void orc_conv1x1_fwd(const float* x, const float* w, const float* b, float* y, long npix, int Cin, int Cout,
                     float sub) {
#pragma omp parallel for schedule(static)
    for (long p = 0; p < npix; ++p) {
        for (int co = 0; co < Cout; ++co) {
            double acc = b ? (double)b[co] : 0.0;
            for (int ci = 0; ci < Cin; ++ci) acc += (double)(x[p * Cin + ci] - sub) * (double)w[ci * Cout + co];
            y[p * Cout + co] = (float)acc;
        }
    }
}

/* gradients of the 1x1 conv: dx[p][ci] = sum_co dy[p][co] w[ci][co]; dw[ci][co] = sum_p (x-sub)[p][ci] dy[p][co] */
void orc_conv1x1_bwd(const float* x, const float* w, const float* dy, float* dx, float* dw, float* db, long npix,
                     int Cin, int Cout, float sub) {
    if (dx) {
#pragma omp parallel for schedule(static)
        for (long p = 0; p < npix; ++p)
            for (int ci = 0; ci < Cin; ++ci) {
                double acc = 0.0;
                for (int co = 0; co < Cout; ++co) acc += (double)dy[p * Cout + co] * (double)w[ci * Cout + co];
                dx[p * Cin + ci] = (float)acc;
            }
    }
    if (dw) {
        for (int ci = 0; ci < Cin; ++ci)
            for (int co = 0; co < Cout; ++co) {
                double acc = 0.0;
                for (long p = 0; p < npix; ++p) acc += (double)(x[p * Cin + ci] - sub) * (double)dy[p * Cout + co];
                dw[ci * Cout + co] = (float)acc;
            }
    }
    if (db) {
        for (int co = 0; co < Cout; ++co) {
            double acc = 0.0;
            for (long p = 0; p < npix; ++p) acc += (double)dy[p * Cout + co];
            db[co] = (float)acc;
        }
    }
}

/* unet.py:34-39 (dilation_rate=(2,2)), :42-45, :88-91: tf.layers.conv2d(kxk, padding='valid') + tf.nn.relu.
 * H,W are INPUT sizes; output is Ho = H - dil*(k-1). */
void orc_conv2d_fwd(const float* x, const float* w, const float* b, float* y, int N, int H, int W, int Cin,
                    int Cout, int k, int dil, int relu) {
    const int Ho = H - dil * (k - 1), Wo = W - dil * (k - 1);
#pragma omp parallel for collapse(2) schedule(static)
    for (int n = 0; n < N; ++n)
        for (int oy = 0; oy < Ho; ++oy) {
            double* acc = (double*)malloc(sizeof(double) * (size_t)Cout);
            for (int ox = 0; ox < Wo; ++ox) {
                for (int co = 0; co < Cout; ++co) acc[co] = b ? (double)b[co] : 0.0;
                for (int ky = 0; ky < k; ++ky)
                    for (int kx = 0; kx < k; ++kx) {
                        const float* xp = x + IDX4(n, oy + ky * dil, ox + kx * dil, 0, H, W, Cin);
                        const float* wp = w + (size_t)(ky * k + kx) * Cin * Cout;
                        for (int ci = 0; ci < Cin; ++ci) {
                            const double xv = (double)xp[ci];
                            const float* wr = wp + (size_t)ci * Cout;
                            for (int co = 0; co < Cout; ++co) acc[co] += xv * (double)wr[co];
                        }
                    }
                float* yp = y + IDX4(n, oy, ox, 0, Ho, Wo, Cout);
                for (int co = 0; co < Cout; ++co) {
                    float v = (float)acc[co];
                    yp[co] = (relu && v < 0.f) ? 0.f : v;
                }
            }
            free(acc);
        }
}

/* Conv2DBackpropInput of the op above: dx[n,iy,ix,ci] = sum_{ky,kx,co} dy[n,iy-ky*dil,ix-kx*dil,co] w[ky,kx,ci,co] */
void orc_conv2d_bwd_data(const float* dy, const float* w, float* dx, int N, int H, int W, int Cin, int Cout, int k,
                         int dil) {
    const int Ho = H - dil * (k - 1), Wo = W - dil * (k - 1);
#pragma omp parallel for collapse(2) schedule(static)
    for (int n = 0; n < N; ++n)
        for (int iy = 0; iy < H; ++iy) {
            double* acc = (double*)malloc(sizeof(double) * (size_t)Cin);
            for (int ix = 0; ix < W; ++ix) {
                for (int ci = 0; ci < Cin; ++ci) acc[ci] = 0.0;
                for (int ky = 0; ky < k; ++ky) {
                    const int oy = iy - ky * dil;
                    if (oy < 0 || oy >= Ho) continue;
                    for (int kx = 0; kx < k; ++kx) {
                        const int ox = ix - kx * dil;
                        if (ox < 0 || ox >= Wo) continue;
                        const float* dp = dy + IDX4(n, oy, ox, 0, Ho, Wo, Cout);
                        const float* wp = w + (size_t)(ky * k + kx) * Cin * Cout;
                        for (int ci = 0; ci < Cin; ++ci) {
                            const float* wr = wp + (size_t)ci * Cout;
                            double s = 0.0;
                            for (int co = 0; co < Cout; ++co) s += (double)dp[co] * (double)wr[co];
                            acc[ci] += s;
                        }
                    }
                }
                float* xp = dx + IDX4(n, iy, ix, 0, H, W, Cin);
                for (int ci = 0; ci < Cin; ++ci) xp[ci] = (float)acc[ci];
            }
            free(acc);
        }
}

/* Conv2DBackpropFilter + BiasAddGrad: dw[ky,kx,ci,co] = sum_{n,oy,ox} x[n,oy+ky*dil,ox+kx*dil,ci] dy[n,oy,ox,co] */
void orc_conv2d_bwd_weight(const float* x, const float* dy, float* dw, float* db, int N, int H, int W, int Cin,
                           int Cout, int k, int dil) {
    const int Ho = H - dil * (k - 1), Wo = W - dil * (k - 1);
#pragma omp parallel for collapse(2) schedule(dynamic)
    for (int t = 0; t < k * k; ++t)
        for (int ci = 0; ci < Cin; ++ci) {
            const int ky = t / k, kx = t % k;
            double* acc = (double*)calloc((size_t)Cout, sizeof(double));
            for (int n = 0; n < N; ++n)
                for (int oy = 0; oy < Ho; ++oy)
                    for (int ox = 0; ox < Wo; ++ox) {
                        const double xv = (double)x[IDX4(n, oy + ky * dil, ox + kx * dil, ci, H, W, Cin)];
                        const float* dp = dy + IDX4(n, oy, ox, 0, Ho, Wo, Cout);
                        for (int co = 0; co < Cout; ++co) acc[co] += xv * (double)dp[co];
                    }
            float* wr = dw + ((size_t)t * Cin + ci) * Cout;
            for (int co = 0; co < Cout; ++co) wr[co] = (float)acc[co];
            free(acc);
        }
    if (db) {
        const size_t np = (size_t)N * Ho * Wo;
        for (int co = 0; co < Cout; ++co) {
            double s = 0.0;
            for (size_t p = 0; p < np; ++p) s += (double)dy[p * Cout + co];
            db[co] = (float)s;
        }
    }
}

/* ReluGrad: dz = dy * (y > 0) where y is the ReLU OUTPUT (unet.py:36,39,43,45,89,91) */
void orc_relu_bwd(const float* y, const float* dy, float* dz, long n) {
#pragma omp parallel for schedule(static)
    for (long i = 0; i < n; ++i) dz[i] = y[i] > 0.f ? dy[i] : 0.f;
}

/* unet.py:52 tf.layers.max_pooling2d(net, (2,2), strides=(2,2)) -- VALID, out = floor(in/2) */
void orc_maxpool2x2_fwd(const float* x, float* y, int N, int H, int W, int C) {
    const int Ho = H / 2, Wo = W / 2;
#pragma omp parallel for collapse(2) schedule(static)
    for (int n = 0; n < N; ++n)
        for (int oy = 0; oy < Ho; ++oy)
            for (int ox = 0; ox < Wo; ++ox)
                for (int c = 0; c < C; ++c) {
                    float m = x[IDX4(n, 2 * oy, 2 * ox, c, H, W, C)];
                    float v;
                    v = x[IDX4(n, 2 * oy, 2 * ox + 1, c, H, W, C)]; if (v > m) m = v;
                    v = x[IDX4(n, 2 * oy + 1, 2 * ox, c, H, W, C)]; if (v > m) m = v;
                    v = x[IDX4(n, 2 * oy + 1, 2 * ox + 1, c, H, W, C)]; if (v > m) m = v;
                    y[IDX4(n, oy, ox, c, Ho, Wo, C)] = m;
                }
}

/* MaxPoolGrad: the gradient goes to the FIRST maximum of the window in row-major window order
 * (0,0),(0,1),(1,0),(1,1). Ties only matter in the bf16-storage emulation; see DESIGN.md. */
void orc_maxpool2x2_bwd(const float* x, const float* dy, float* dx, int N, int H, int W, int C) {
    const int Ho = H / 2, Wo = W / 2;
    memset(dx, 0, sizeof(float) * (size_t)N * H * W * C);
#pragma omp parallel for collapse(2) schedule(static)
    for (int n = 0; n < N; ++n)
        for (int oy = 0; oy < Ho; ++oy)
            for (int ox = 0; ox < Wo; ++ox)
                for (int c = 0; c < C; ++c) {
                    int by = 0, bx = 0;
                    float m = x[IDX4(n, 2 * oy, 2 * ox, c, H, W, C)];
                    for (int a = 0; a < 2; ++a)
                        for (int b = 0; b < 2; ++b) {
                            float v = x[IDX4(n, 2 * oy + a, 2 * ox + b, c, H, W, C)];
                            if (v > m) { m = v; by = a; bx = b; }
                        }
                    dx[IDX4(n, 2 * oy + by, 2 * ox + bx, c, H, W, C)] = dy[IDX4(n, oy, ox, c, Ho, Wo, C)];
                }
}

/* unet.py:67-68 tf.layers.conv2d_transpose(net, nf, kernel_size=(2,2), strides=(2,2)) (VALID, no activation)
 * y[n,2i+a,2j+b,co] = b[co] + sum_ci x[n,i,j,ci] K[a,b,co,ci];  K layout [2][2][Cout][Cin] */
void orc_convT2x2s2_fwd(const float* x, const float* K, const float* b, float* y, int N, int H, int W, int Cin,
                        int Cout) {
    const int Ho = 2 * H, Wo = 2 * W;
#pragma omp parallel for collapse(2) schedule(static)
    for (int n = 0; n < N; ++n)
        for (int i = 0; i < H; ++i)
            for (int j = 0; j < W; ++j) {
                const float* xp = x + IDX4(n, i, j, 0, H, W, Cin);
                for (int a = 0; a < 2; ++a)
                    for (int bb = 0; bb < 2; ++bb) {
                        float* yp = y + IDX4(n, 2 * i + a, 2 * j + bb, 0, Ho, Wo, Cout);
                        const float* kp = K + (size_t)(a * 2 + bb) * Cout * Cin;
                        for (int co = 0; co < Cout; ++co) {
                            double acc = b ? (double)b[co] : 0.0;
                            const float* kr = kp + (size_t)co * Cin;
                            for (int ci = 0; ci < Cin; ++ci) acc += (double)xp[ci] * (double)kr[ci];
                            yp[co] = (float)acc;
                        }
                    }
            }
}

/* gradients of the transposed conv. dx[n,i,j,ci] = sum_{a,b,co} dy[n,2i+a,2j+b,co] K[a,b,co,ci]
 * dK[a,b,co,ci] = sum_{n,i,j} dy[n,2i+a,2j+b,co] x[n,i,j,ci];  db[co] = sum dy */
void orc_convT2x2s2_bwd(const float* x, const float* K, const float* dy, float* dx, float* dK, float* db, int N,
                        int H, int W, int Cin, int Cout) {
    const int Ho = 2 * H, Wo = 2 * W;
    if (dx) {
#pragma omp parallel for collapse(2) schedule(static)
        for (int n = 0; n < N; ++n)
            for (int i = 0; i < H; ++i)
                for (int j = 0; j < W; ++j) {
                    float* xp = dx + IDX4(n, i, j, 0, H, W, Cin);
                    for (int ci = 0; ci < Cin; ++ci) {
                        double acc = 0.0;
                        for (int a = 0; a < 2; ++a)
                            for (int bb = 0; bb < 2; ++bb) {
                                const float* dp = dy + IDX4(n, 2 * i + a, 2 * j + bb, 0, Ho, Wo, Cout);
                                const float* kp = K + (size_t)(a * 2 + bb) * Cout * Cin + ci;
                                for (int co = 0; co < Cout; ++co) acc += (double)dp[co] * (double)kp[(size_t)co * Cin];
                            }
                        xp[ci] = (float)acc;
                    }
                }
    }
    if (dK) {
#pragma omp parallel for collapse(2) schedule(dynamic)
        for (int t = 0; t < 4; ++t)
            for (int co = 0; co < Cout; ++co) {
                const int a = t / 2, bb = t % 2;
                double* acc = (double*)calloc((size_t)Cin, sizeof(double));
                for (int n = 0; n < N; ++n)
                    for (int i = 0; i < H; ++i)
                        for (int j = 0; j < W; ++j) {
                            const double dv = (double)dy[IDX4(n, 2 * i + a, 2 * j + bb, co, Ho, Wo, Cout)];
                            const float* xp = x + IDX4(n, i, j, 0, H, W, Cin);
                            for (int ci = 0; ci < Cin; ++ci) acc[ci] += dv * (double)xp[ci];
                        }
                float* kr = dK + ((size_t)t * Cout + co) * Cin;
                for (int ci = 0; ci < Cin; ++ci) kr[ci] = (float)acc[ci];
                free(acc);
            }
    }
    if (db) {
        const size_t np = (size_t)N * Ho * Wo;
        for (int co = 0; co < Cout; ++co) {
            double s = 0.0;
            for (size_t p = 0; p < np; ++p) s += (double)dy[p * Cout + co];
            db[co] = (float)s;
        }
    }
}

/* tf_aerial_images.py:147-149 and :103-110
 *   predictions = softmax(logits, dim=3)[:, :, :, 1]
 *   loss = reduce_mean(sparse_softmax_cross_entropy_with_logits(labels, logits))
 * dlogits is d(loss)/d(logits) = (softmax - onehot) / npix  (the fused TF gradient). */
void orc_softmax_ce(const float* logits, const int64_t* labels, float* prob1, double* loss_out, float* dlogits,
                    long npix) {
    double loss = 0.0;
    for (long p = 0; p < npix; ++p) {
        const double l0 = logits[2 * p], l1 = logits[2 * p + 1];
        const double m = l0 > l1 ? l0 : l1;
        const double e0 = exp(l0 - m), e1 = exp(l1 - m);
        const double s = e0 + e1;
        const double p0 = e0 / s, p1 = e1 / s;
        if (prob1) prob1[p] = (float)p1;
        if (labels) {
            const int lab = (int)labels[p];
            loss += -((lab ? l1 : l0) - m - log(s));
            if (dlogits) {
                dlogits[2 * p] = (float)((p0 - (lab == 0 ? 1.0 : 0.0)) / (double)npix);
                dlogits[2 * p + 1] = (float)((p1 - (lab == 1 ? 1.0 : 0.0)) / (double)npix);
            }
        }
    }
    if (loss_out) *loss_out = loss / (double)npix;
}

/* tf_aerial_images.py:116-121 tf.train.MomentumOptimizer(lr, momentum) (use_nesterov=False):
 *   accum = momentum * accum + grad ;  var -= lr * accum */
void orc_momentum_step(float* w, float* acc, const float* g, float lr, float mu, long n) {
#pragma omp parallel for schedule(static)
    for (long i = 0; i < n; ++i) {
        const float a = mu * acc[i] + g[i];
        acc[i] = a;
        w[i] -= lr * a;
    }
}
