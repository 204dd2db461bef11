/* Plain-C restatement of the Caffe layer arithmetic the reference reaches through pycaffe.
 * TEST INFRASTRUCTURE ONLY (see oracle/__init__.py) -- never linked into the product.
 *
 * Reference call sites:  worker.py:84-86 (net.forward), worker.py:100-106 (net.backward),
 * models/vgg19.prototxt (Convolution kernel 3 pad 1 stride 1; in-place ReLU; MAX Pooling 2/2).
 * The algorithm is BVLC Caffe's (not under /root/reference, un-pinned): here written as direct
 * loops so that it is an implementation independent of oracle/caffe_net.py's im2col+SGEMM.
 *
 * All tensors are (C, H, W) float32, contiguous.  Build: see oracle/Makefile.
 */
#include <float.h>
#include <stddef.h>
#include <string.h>

/* y[co] = relu?(b[co] + sum_{ci,ky,kx} w[co][ci][ky][kx] * x[ci][r+ky-1][c+kx-1]) */
void ref_conv3x3_forward(const float *x, const float *w, const float *b, float *y,
                         int cin, int cout, int h, int wd, int relu)
{
    const size_t plane = (size_t)h * wd;
#pragma omp parallel for schedule(dynamic, 1)
    for (int co = 0; co < cout; ++co) {
        float *yo = y + co * plane;
        for (size_t i = 0; i < plane; ++i) yo[i] = b ? b[co] : 0.0f;
        for (int ci = 0; ci < cin; ++ci) {
            const float *xi = x + ci * plane;
            const float *wk = w + ((size_t)co * cin + ci) * 9;
            for (int ky = 0; ky < 3; ++ky) {
                for (int kx = 0; kx < 3; ++kx) {
                    const float wv = wk[ky * 3 + kx];
                    const int c0 = kx == 0 ? 1 : 0, c1 = kx == 2 ? wd - 1 : wd;
                    for (int r = 0; r < h; ++r) {
                        const int rs = r + ky - 1;
                        if (rs < 0 || rs >= h) continue;
                        const float *xr = xi + (size_t)rs * wd + (kx - 1);
                        float *yr = yo + (size_t)r * wd;
                        for (int c = c0; c < c1; ++c) yr[c] += wv * xr[c];
                    }
                }
            }
        }
        if (relu)
            for (size_t i = 0; i < plane; ++i) yo[i] = yo[i] > 0.0f ? yo[i] : 0.0f;
    }
}

/* dx[ci][r][c] = sum_{co,ky,kx} w[co][ci][ky][kx] * dy[co][r-ky+1][c-kx+1] */
void ref_conv3x3_backward_data(const float *dy, const float *w, float *dx,
                               int cin, int cout, int h, int wd)
{
    const size_t plane = (size_t)h * wd;
#pragma omp parallel for schedule(dynamic, 1)
    for (int ci = 0; ci < cin; ++ci) {
        float *xo = dx + ci * plane;
        memset(xo, 0, plane * sizeof(float));
        for (int co = 0; co < cout; ++co) {
            const float *yo = dy + co * plane;
            const float *wk = w + ((size_t)co * cin + ci) * 9;
            for (int ky = 0; ky < 3; ++ky) {
                for (int kx = 0; kx < 3; ++kx) {
                    const float wv = wk[ky * 3 + kx];
                    /* source column c - kx + 1 must lie in [0, wd) */
                    const int c0 = kx == 2 ? 1 : 0, c1 = kx == 0 ? wd - 1 : wd;
                    for (int r = 0; r < h; ++r) {
                        const int rs = r - ky + 1;
                        if (rs < 0 || rs >= h) continue;
                        const float *yr = yo + (size_t)rs * wd + (1 - kx);
                        float *xr = xo + (size_t)r * wd;
                        for (int c = c0; c < c1; ++c) xr[c] += wv * yr[c];
                    }
                }
            }
        }
    }
}

/* ReLU backward on an in-place blob: g *= (data > 0) */
void ref_relu_mask(float *g, const float *data, size_t n)
{
#pragma omp parallel for
    for (size_t i = 0; i < n; ++i) g[i] = data[i] > 0.0f ? g[i] : 0.0f;
}

static int pooled(int n) { int q = (n - 2 + 1) / 2; if (n - 2 < 0) q = 0; return q + 1; }

int ref_pooled_size(int n) { return pooled(n); }

/* Caffe MAX pooling 2x2 stride 2, ceil mode, clipped windows, first strictly-greater wins.
 * argmax receives the flat index (r * wd + c) inside the input plane. */
void ref_maxpool_forward(const float *x, float *y, int *argmax, int ch, int h, int wd)
{
    const int ho = pooled(h), wo = pooled(wd);
#pragma omp parallel for
    for (int c = 0; c < ch; ++c) {
        const float *xi = x + (size_t)c * h * wd;
        for (int pr = 0; pr < ho; ++pr) {
            for (int pc = 0; pc < wo; ++pc) {
                const int r0 = pr * 2, c0 = pc * 2;
                const int r1 = r0 + 2 < h ? r0 + 2 : h, c1 = c0 + 2 < wd ? c0 + 2 : wd;
                float best = -FLT_MAX;
                int where = -1;
                for (int r = r0; r < r1; ++r)
                    for (int cc = c0; cc < c1; ++cc)
                        if (xi[r * wd + cc] > best) { best = xi[r * wd + cc]; where = r * wd + cc; }
                const size_t o = ((size_t)c * ho + pr) * wo + pc;
                y[o] = best;
                argmax[o] = where;
            }
        }
    }
}

void ref_maxpool_backward(const float *dy, const int *argmax, float *dx, int ch, int h, int wd)
{
    const int ho = pooled(h), wo = pooled(wd);
    memset(dx, 0, (size_t)ch * h * wd * sizeof(float));
#pragma omp parallel for
    for (int c = 0; c < ch; ++c)
        for (int o = 0; o < ho * wo; ++o) {
            const int a = argmax[(size_t)c * ho * wo + o];
            if (a >= 0) dx[(size_t)c * h * wd + a] += dy[(size_t)c * ho * wo + o];
        }
}
