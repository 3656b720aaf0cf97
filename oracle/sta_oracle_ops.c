/*
 * sta_oracle_ops.c - CPU restatement (plain C, fp32) of the arithmetic on the STA hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Imported/linked only by tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py, as the checker - never by the product path (which has no CPU
 * fallback).  Each function cites the reference code it restates (paths under
 * vista_slam/sta_model/).  The composition of these ops into the model lives in
 * oracle/sta_oracle.py.  Pinned against reference-generated golden vectors by
 * tests/test_oracle_golden.py (tests/golden/*.npz, produced by oracle/gen_golden.py).
 *
 * Build: gcc -O3 -march=native -fopenmp -shared -fPIC (oracle/build_oracle.py).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* y[M,N] = x[M,K] w[N,K]^T + b[N]      nn.Linear (blocks/sta_blocks.py:74,77,132,146,193-195,207) */
void o_linear(const float* x, const float* w, const float* b, float* y, int M, int N, int K) {
#pragma omp parallel for schedule(static)
    for (int m = 0; m < M; ++m) {
        const float* xr = x + (size_t)m * K;
        int n = 0;
        for (; n + 4 <= N; n += 4) {
            const float *w0 = w + (size_t)n * K, *w1 = w0 + K, *w2 = w1 + K, *w3 = w2 + K;
            float s0 = 0, s1 = 0, s2 = 0, s3 = 0;
#pragma omp simd reduction(+ : s0, s1, s2, s3)
            for (int k = 0; k < K; ++k) {
                float xv = xr[k];
                s0 += xv * w0[k]; s1 += xv * w1[k]; s2 += xv * w2[k]; s3 += xv * w3[k];
            }
            float* yr = y + (size_t)m * N + n;
            yr[0] = s0 + (b ? b[n] : 0.f); yr[1] = s1 + (b ? b[n + 1] : 0.f);
            yr[2] = s2 + (b ? b[n + 2] : 0.f); yr[3] = s3 + (b ? b[n + 3] : 0.f);
        }
        for (; n < N; ++n) {
            const float* w0 = w + (size_t)n * K;
            float s0 = 0;
#pragma omp simd reduction(+ : s0)
            for (int k = 0; k < K; ++k) s0 += xr[k] * w0[k];
            y[(size_t)m * N + n] = s0 + (b ? b[n] : 0.f);
        }
    }
}

/* nn.LayerNorm(C, eps) rows, affine (sta_model.py:43 eps=1e-6; blocks/sta_blocks.py:166-169,226-231) */
void o_layernorm(const float* x, const float* g, const float* b, float* y, int M, int C, float eps) {
#pragma omp parallel for schedule(static)
    for (int m = 0; m < M; ++m) {
        const float* xr = x + (size_t)m * C;
        float* yr = y + (size_t)m * C;
        double s = 0;
        for (int c = 0; c < C; ++c) s += xr[c];
        float mean = (float)(s / C);
        double v = 0;
        for (int c = 0; c < C; ++c) { double d = xr[c] - mean; v += d * d; }
        float rstd = 1.0f / sqrtf((float)(v / C) + eps);
        for (int c = 0; c < C; ++c) yr[c] = (xr[c] - mean) * rstd * g[c] + b[c];
    }
}

/* nn.GELU() exact erf form (blocks/sta_blocks.py:60,75) */
void o_gelu(float* x, int64_t n) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) x[i] = 0.5f * x[i] * (1.0f + erff(x[i] * 0.70710678118654752440f));
}

/* RoPE2D.forward (pos_embed/pos_embed.py:169-185; == rope_2d_cpu, pos_embed/curope/curope.cpp:11-47):
 * tokens (B,H,N,D) in place; per head dim D: [0,D/4),[D/4,D/2) rotated by pos_y * base^(-d/(D/4)),
 * [D/2,3D/4),[3D/4,D) by pos_x. */
void o_rope2d(float* tok, const int64_t* pos, int B, int H, int N, int D, float base) {
    const int Q = D / 4;
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < B; ++b)
        for (int h = 0; h < H; ++h)
            for (int n = 0; n < N; ++n) {
                float* t = tok + (((size_t)b * H + h) * N + n) * D;
                for (int xy = 0; xy < 2; ++xy) {
                    const float p = (float)pos[((size_t)b * N + n) * 2 + xy];
                    for (int d = 0; d < Q; ++d) {
                        const float inv_freq = 1.0f / powf(base, (float)d / (float)Q);
                        const float f = p * inv_freq;
                        const float c = cosf(f), s = sinf(f);
                        const float u = t[xy * 2 * Q + d], v = t[xy * 2 * Q + Q + d];
                        t[xy * 2 * Q + d] = u * c - v * s;
                        t[xy * 2 * Q + Q + d] = v * c + u * s;
                    }
                }
            }
}

/* softmax(q k^T * scale) v  (blocks/sta_blocks.py:143 xformers FMHA == :201-205 naive form)
 * q (B,H,Nq,D), k,v (B,H,Nk,D) -> out (B,Nq,H*D) */
void o_attention(const float* q, const float* k, const float* v, float* out, int B, int H, int Nq, int Nk, int D, float scale) {
#pragma omp parallel
    {
        float* sc = (float*)malloc(sizeof(float) * (size_t)Nk);
#pragma omp for collapse(3) schedule(static)
        for (int b = 0; b < B; ++b)
            for (int h = 0; h < H; ++h)
                for (int i = 0; i < Nq; ++i) {
                    const float* qi = q + (((size_t)b * H + h) * Nq + i) * D;
                    const float* kb = k + ((size_t)b * H + h) * Nk * D;
                    const float* vb = v + ((size_t)b * H + h) * Nk * D;
                    float mx = -INFINITY;
                    for (int j = 0; j < Nk; ++j) {
                        float s = 0;
                        for (int d = 0; d < D; ++d) s += qi[d] * kb[(size_t)j * D + d];
                        s *= scale; sc[j] = s; if (s > mx) mx = s;
                    }
                    float sum = 0;
                    for (int j = 0; j < Nk; ++j) { sc[j] = expf(sc[j] - mx); sum += sc[j]; }
                    float inv = 1.0f / sum;
                    float* o = out + ((size_t)b * Nq + i) * H * D + (size_t)h * D;
                    for (int d = 0; d < D; ++d) o[d] = 0;
                    for (int j = 0; j < Nk; ++j) {
                        const float pj = sc[j] * inv;
                        for (int d = 0; d < D; ++d) o[d] += pj * vb[(size_t)j * D + d];
                    }
                }
        free(sc);
    }
}

/* nn.Conv2d NCHW, square kernel k, stride, zero pad (blocks/sta_blocks.py:262; heads/dpt_block.py:20-77,
 * 94-112,178-186,316-324,356-410).  w [Co,Ci,k,k], b may be NULL. */
void o_conv2d(const float* x, const float* w, const float* b, float* y, int B, int Ci, int H, int W, int Co, int k, int stride, int pad) {
    const int Ho = (H + 2 * pad - k) / stride + 1, Wo = (W + 2 * pad - k) / stride + 1;
#pragma omp parallel for collapse(2) schedule(static)
    for (int bb = 0; bb < B; ++bb)
        for (int co = 0; co < Co; ++co) {
            float* yo = y + ((size_t)bb * Co + co) * Ho * Wo;
            for (int i = 0; i < Ho * Wo; ++i) yo[i] = b ? b[co] : 0.f;
            for (int ci = 0; ci < Ci; ++ci) {
                const float* xi = x + ((size_t)bb * Ci + ci) * H * W;
                const float* wk = w + ((size_t)co * Ci + ci) * k * k;
                for (int ky = 0; ky < k; ++ky)
                    for (int kx = 0; kx < k; ++kx) {
                        const float wv = wk[ky * k + kx];
                        for (int yy = 0; yy < Ho; ++yy) {
                            const int iy = yy * stride + ky - pad;
                            if (iy < 0 || iy >= H) continue;
                            float* yrow = yo + (size_t)yy * Wo;
                            const float* xrow = xi + (size_t)iy * W;
                            int x0 = 0, x1 = Wo;
                            while (x0 < Wo && x0 * stride + kx - pad < 0) ++x0;
                            while (x1 > x0 && (x1 - 1) * stride + kx - pad >= W) --x1;
                            for (int xx = x0; xx < x1; ++xx) yrow[xx] += wv * xrow[xx * stride + kx - pad];
                        }
                    }
            }
        }
}

/* nn.ConvTranspose2d with kernel == stride == k, no padding (heads/dpt_block.py:369-374,383-388).
 * x [B,Ci,H,W], w [Ci,Co,k,k] -> y [B,Co,H*k,W*k] */
void o_conv_transpose2d(const float* x, const float* w, const float* b, float* y, int B, int Ci, int H, int W, int Co, int k) {
    const int Ho = H * k, Wo = W * k;
#pragma omp parallel for collapse(2) schedule(static)
    for (int bb = 0; bb < B; ++bb)
        for (int co = 0; co < Co; ++co) {
            float* yo = y + ((size_t)bb * Co + co) * Ho * Wo;
            for (int i = 0; i < Ho * Wo; ++i) yo[i] = b ? b[co] : 0.f;
            for (int ci = 0; ci < Ci; ++ci) {
                const float* xi = x + ((size_t)bb * Ci + ci) * H * W;
                const float* wk = w + ((size_t)ci * Co + co) * k * k;
                for (int yy = 0; yy < H; ++yy)
                    for (int xx = 0; xx < W; ++xx) {
                        const float xv = xi[(size_t)yy * W + xx];
                        for (int dy = 0; dy < k; ++dy)
                            for (int dx = 0; dx < k; ++dx) yo[(size_t)(yy * k + dy) * Wo + xx * k + dx] += xv * wk[dy * k + dx];
                    }
            }
        }
}

/* F.interpolate(scale_factor=2, mode='bilinear', align_corners=True) (heads/dpt_block.py:215-216,320) */
void o_bilinear_up2(const float* x, float* y, int BC, int H, int W) {
    const int Ho = 2 * H, Wo = 2 * W;
    const float ry = Ho > 1 ? (float)(H - 1) / (float)(Ho - 1) : 0.f;
    const float rx = Wo > 1 ? (float)(W - 1) / (float)(Wo - 1) : 0.f;
#pragma omp parallel for schedule(static)
    for (int p = 0; p < BC; ++p) {
        const float* xi = x + (size_t)p * H * W;
        float* yo = y + (size_t)p * Ho * Wo;
        for (int yy = 0; yy < Ho; ++yy) {
            const float sy = ry * yy; const int y0 = (int)sy; const int y1 = y0 + (y0 < H - 1); const float fy = sy - y0;
            for (int xx = 0; xx < Wo; ++xx) {
                const float sx = rx * xx; const int x0 = (int)sx; const int x1 = x0 + (x0 < W - 1); const float fx = sx - x0;
                yo[(size_t)yy * Wo + xx] = (1.f - fy) * ((1.f - fx) * xi[(size_t)y0 * W + x0] + fx * xi[(size_t)y0 * W + x1]) +
                                           fy * ((1.f - fx) * xi[(size_t)y1 * W + x0] + fx * xi[(size_t)y1 * W + x1]);
            }
        }
    }
}

/* ---- PoseHead_small.svd_orthogonalize (heads/pose_head.py:38-57), followed literally:
 *   mt = transpose(normalize(m, dim=-1));  u,s,v = svd(mt);  det = det(v u^T)
 *   r  = [v0, v1, v2*det] u^T
 * SVD by one-sided Jacobi in double (torch.svd returns singular values in descending order). */
static void svd3(const double A[3][3], double U[3][3], double S[3], double V[3][3]) {
    double G[3][3];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { G[i][j] = A[i][j]; V[i][j] = (i == j); }
    for (int sweep = 0; sweep < 60; ++sweep) {
        double off = 0;
        for (int p = 0; p < 2; ++p)
            for (int q = p + 1; q < 3; ++q) {
                double a = 0, b = 0, c = 0;
                for (int i = 0; i < 3; ++i) { a += G[i][p] * G[i][p]; b += G[i][q] * G[i][q]; c += G[i][p] * G[i][q]; }
                off += c * c;
                if (fabs(c) < 1e-300) continue;
                double zeta = (b - a) / (2.0 * c);
                double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                double cs = 1.0 / sqrt(1.0 + t * t), sn = cs * t;
                for (int i = 0; i < 3; ++i) {
                    double gp = G[i][p], gq = G[i][q]; G[i][p] = cs * gp - sn * gq; G[i][q] = sn * gp + cs * gq;
                    double vp = V[i][p], vq = V[i][q]; V[i][p] = cs * vp - sn * vq; V[i][q] = sn * vp + cs * vq;
                }
            }
        if (off < 1e-60) break;
    }
    int idx[3] = {0, 1, 2};
    double nrm[3];
    for (int j = 0; j < 3; ++j) nrm[j] = sqrt(G[0][j] * G[0][j] + G[1][j] * G[1][j] + G[2][j] * G[2][j]);
    for (int a = 0; a < 2; ++a) for (int b = a + 1; b < 3; ++b) if (nrm[idx[b]] > nrm[idx[a]]) { int t = idx[a]; idx[a] = idx[b]; idx[b] = t; }
    double Vs[3][3];
    for (int j = 0; j < 3; ++j) {
        S[j] = nrm[idx[j]];
        for (int i = 0; i < 3; ++i) { U[i][j] = nrm[idx[j]] > 1e-300 ? G[i][idx[j]] / nrm[idx[j]] : 0.0; Vs[i][j] = V[i][idx[j]]; }
    }
    /* complete a vanishing last left vector so that U stays orthogonal */
    if (S[2] < 1e-12 * S[0]) {
        U[0][2] = U[1][0] * U[2][1] - U[2][0] * U[1][1];
        U[1][2] = U[2][0] * U[0][1] - U[0][0] * U[2][1];
        U[2][2] = U[0][0] * U[1][1] - U[1][0] * U[0][1];
    }
    memcpy(V, Vs, sizeof(Vs));
}
static double det3(const double M[3][3]) {
    return M[0][0] * (M[1][1] * M[2][2] - M[1][2] * M[2][1]) - M[0][1] * (M[1][0] * M[2][2] - M[1][2] * M[2][0]) +
           M[0][2] * (M[1][0] * M[2][1] - M[1][1] * M[2][0]);
}
void o_svd_orthogonalize(const float* m, float* r, int B) {
    for (int b = 0; b < B; ++b) {
        double mt[3][3], U[3][3], S[3], V[3][3], VUt[3][3], R[3][3];
        for (int i = 0; i < 3; ++i) {
            const float* row = m + b * 9 + i * 3;
            float n = sqrtf(row[0] * row[0] + row[1] * row[1] + row[2] * row[2]);
            if (n < 1e-12f) n = 1e-12f;
            for (int j = 0; j < 3; ++j) mt[j][i] = row[j] / n;         /* transpose of the normalised rows */
        }
        svd3(mt, U, S, V);
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { double s = 0; for (int k = 0; k < 3; ++k) s += V[i][k] * U[j][k]; VUt[i][j] = s; }
        const double det = det3(VUt);
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) {
            double s = 0;
            for (int k = 0; k < 3; ++k) s += V[i][k] * (k == 2 ? det : 1.0) * U[j][k];
            R[i][j] = s;
        }
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) r[b * 9 + i * 3 + j] = (float)R[i][j];
    }
}

/* postprocess (heads/postprocess.py:10-62), modes depth ('exp',-inf,inf), conf ('exp',1,inf):
 * out [B,4,H,W] -> pts [B,H,W,3] = xyz/clip(|xyz|,1e-8)*expm1(|xyz|), conf [B,H,W] = 1 + exp(c) */
void o_postprocess(const float* out, float* pts, float* conf, int B, int H, int W) {
    const size_t hw = (size_t)H * W;
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < (int64_t)B * (int64_t)hw; ++i) {
        const size_t b = (size_t)i / hw, p = (size_t)i % hw;
        const float* o = out + b * 4 * hw + p;
        const float x = o[0], y = o[hw], z = o[2 * hw], c = o[3 * hw];
        const float d = sqrtf(x * x + y * y + z * z);
        const float dd = d < 1e-8f ? 1e-8f : d;
        const float e = expm1f(d);
        pts[i * 3 + 0] = x / dd * e; pts[i * 3 + 1] = y / dd * e; pts[i * 3 + 2] = z / dd * e;
        conf[i] = 1.0f + expf(c);
    }
}
