// HBM-bound passes of the style-transfer inner loop (everything that is not a GEMM).
// Built with -ffp-contract=off so that the elementwise fp32 arithmetic rounds like the
// reference's NumPy expressions (one IEEE operation per NumPy operation, same order).
//
//   max pool fwd/bwd            models/vgg19.prototxt:45-55 ...  (Caffe ceil mode, first max)
//   content / deep-dream terms  worker.py:249-256, 271-277
//   TV + p-norm + combine + Adam  utils.py:285-304, worker.py:279-297, optimizers.py:20-27
//   pre / deprocess             worker.py:63-71
//   trace scalars               worker.py:236-301, utils.py:257-282
//   BLAS-1 for L-BFGS           utils.py:29-46, optimizers.py:62-108
#include "st2_kernels.h"
#include "wave_reduce.h"
#include <float.h>
#include <math.h>

namespace st2 {

int pooled_size(int n)
{
    int q = (n - 2 + 1) / 2;
    if (n - 2 < 0) q = 0;
    return q + 1;
}

// ------------------------------------------------------------------------------------------ pool
__global__ __launch_bounds__(256) void maxpool_fwd_k(const float* __restrict__ in, float* __restrict__ out,
                                                     int C, int H, int W, int Ho, int Wo)
{
    const size_t total = (size_t)C * Ho * Wo;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const int px = (int)(idx % Wo);
        const int py = (int)((idx / Wo) % Ho);
        const int c = (int)(idx / ((size_t)Wo * Ho));
        const float* p = in + (size_t)c * H * W;
        const int r0 = 2 * py, c0 = 2 * px;
        float best = -FLT_MAX;
#pragma unroll
        for (int dr = 0; dr < 2; ++dr)
#pragma unroll
            for (int dc = 0; dc < 2; ++dc) {
                const int r = r0 + dr, cc = c0 + dc;
                if (r < H && cc < W) {
                    const float v = p[(size_t)r * W + cc];
                    if (v > best) best = v;
                }
            }
        out[idx] = best;
    }
}

// Fast paths for H even, W % 4 == 0 (every window full): one thread = two adjacent windows = a 2 x 4 patch,
// 16-byte loads and stores.
__device__ __forceinline__ int first_max4(float a, float b, float c, float d, float* best)
{
    // Caffe scan order (r0,c0) (r0,c1) (r1,c0) (r1,c1); strictly greater wins -> first maximum
    int arg = 0; float m = a;
    if (b > m) { m = b; arg = 1; }
    if (c > m) { m = c; arg = 2; }
    if (d > m) { m = d; arg = 3; }
    *best = m;
    return arg;
}

__global__ __launch_bounds__(256) void maxpool_fwd_v4_k(const float* __restrict__ in, float* __restrict__ out,
                                                        size_t n_patches, int W4, int Ho)
{
    // patch id -> (plane row pair, quad column); in: rows of W = 4*W4 floats
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < n_patches; idx += (size_t)gridDim.x * 256) {
        const size_t q = idx % W4, pr = idx / W4;                 // pr = c * Ho + py
        const float4 r0 = *reinterpret_cast<const float4*>(in + (pr * 2) * (size_t)W4 * 4 + q * 4);
        const float4 r1 = *reinterpret_cast<const float4*>(in + (pr * 2 + 1) * (size_t)W4 * 4 + q * 4);
        float2 o;
        first_max4(r0.x, r0.y, r1.x, r1.y, &o.x);
        first_max4(r0.z, r0.w, r1.z, r1.w, &o.y);
        *reinterpret_cast<float2*>(out + pr * (size_t)W4 * 2 + q * 2) = o;
    }
}

__global__ __launch_bounds__(256) void maxpool_bwd_v4_k(const float* __restrict__ dy, const float* __restrict__ x,
                                                        float* __restrict__ dx, const float* __restrict__ inject,
                                                        int apply_mask, size_t n_patches, int W4)
{
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < n_patches; idx += (size_t)gridDim.x * 256) {
        const size_t q = idx % W4, pr = idx / W4;
        const size_t o0 = (pr * 2) * (size_t)W4 * 4 + q * 4, o1 = o0 + (size_t)W4 * 4;
        const float4 r0 = *reinterpret_cast<const float4*>(x + o0);
        const float4 r1 = *reinterpret_cast<const float4*>(x + o1);
        const float2 g = *reinterpret_cast<const float2*>(dy + pr * (size_t)W4 * 2 + q * 2);
        float best;
        const int a0 = first_max4(r0.x, r0.y, r1.x, r1.y, &best);
        const int a1 = first_max4(r0.z, r0.w, r1.z, r1.w, &best);
        float4 d0 = make_float4(a0 == 0 ? g.x : 0.f, a0 == 1 ? g.x : 0.f, a1 == 0 ? g.y : 0.f, a1 == 1 ? g.y : 0.f);
        float4 d1 = make_float4(a0 == 2 ? g.x : 0.f, a0 == 3 ? g.x : 0.f, a1 == 2 ? g.y : 0.f, a1 == 3 ? g.y : 0.f);
        if (apply_mask) {
            d0.x = r0.x > 0.f ? d0.x : 0.f; d0.y = r0.y > 0.f ? d0.y : 0.f; d0.z = r0.z > 0.f ? d0.z : 0.f; d0.w = r0.w > 0.f ? d0.w : 0.f;
            d1.x = r1.x > 0.f ? d1.x : 0.f; d1.y = r1.y > 0.f ? d1.y : 0.f; d1.z = r1.z > 0.f ? d1.z : 0.f; d1.w = r1.w > 0.f ? d1.w : 0.f;
        }
        if (inject) {
            const float4 i0 = *reinterpret_cast<const float4*>(inject + o0);
            const float4 i1 = *reinterpret_cast<const float4*>(inject + o1);
            d0.x += i0.x; d0.y += i0.y; d0.z += i0.z; d0.w += i0.w;
            d1.x += i1.x; d1.y += i1.y; d1.z += i1.z; d1.w += i1.w;
        }
        *reinterpret_cast<float4*>(dx + o0) = d0;
        *reinterpret_cast<float4*>(dx + o1) = d1;
    }
}

// The same routing from the one-byte arg-max map written by the producing conv's epilogue (bits 0-1 slot, bit 2 maximum > 0):
// a thread takes two pooled pixels = a 2 x 4 patch of dx.
__global__ __launch_bounds__(256) void maxpool_bwd_amap_k(const float* __restrict__ dy, const unsigned char* __restrict__ amap,
                                                          float* __restrict__ dx, size_t n_patches, int W4)
{
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < n_patches; idx += (size_t)gridDim.x * 256) {
        const size_t q = idx % W4, pr = idx / W4;
        const size_t o0 = (pr * 2) * (size_t)W4 * 4 + q * 4, o1 = o0 + (size_t)W4 * 4;
        const size_t po = pr * (size_t)W4 * 2 + q * 2;
        const float2 g = *reinterpret_cast<const float2*>(dy + po);
        const unsigned short mm = *reinterpret_cast<const unsigned short*>(amap + po);
        const unsigned m0 = mm & 0xffu, m1 = mm >> 8;
        const float g0 = (m0 & 4u) ? g.x : 0.f, g1 = (m1 & 4u) ? g.y : 0.f;
        const unsigned a0 = m0 & 3u, a1 = m1 & 3u;
        *reinterpret_cast<float4*>(dx + o0) = make_float4(a0 == 0 ? g0 : 0.f, a0 == 1 ? g0 : 0.f, a1 == 0 ? g1 : 0.f, a1 == 1 ? g1 : 0.f);
        *reinterpret_cast<float4*>(dx + o1) = make_float4(a0 == 2 ? g0 : 0.f, a0 == 3 ? g0 : 0.f, a1 == 2 ? g1 : 0.f, a1 == 3 ? g1 : 0.f);
    }
}

hipError_t launch_maxpool_bwd_amap(const float* dy, const unsigned char* amap, float* dx, int C, int H, int W, hipStream_t s)
{
    if (H % 2 != 0 || W % 4 != 0 || (reinterpret_cast<uintptr_t>(amap) & 1) != 0) return hipErrorInvalidValue;
    const size_t n_patches = (size_t)C * (H / 2) * (W / 4);
    maxpool_bwd_amap_k<<<reduce_grid(n_patches, 256, 65536), 256, 0, s>>>(dy, amap, dx, n_patches, W / 4);
    return hipGetLastError();
}

hipError_t launch_maxpool_fwd(const float* in, float* out, int C, int H, int W, hipStream_t s)
{
    const int Ho = pooled_size(H), Wo = pooled_size(W);
    const size_t total = (size_t)C * Ho * Wo;
    if (H % 2 == 0 && W % 4 == 0) {
        const size_t n_patches = (size_t)C * Ho * (W / 4);
        maxpool_fwd_v4_k<<<reduce_grid(n_patches, 256, 65536), 256, 0, s>>>(in, out, n_patches, W / 4, Ho);
        return hipGetLastError();
    }
    maxpool_fwd_k<<<reduce_grid(total, 256, 65536), 256, 0, s>>>(in, out, C, H, W, Ho, Wo);
    return hipGetLastError();
}

// One thread per 2x2 window: recompute the (first) arg-max from the input blob and route dy to it.
__global__ __launch_bounds__(256) void maxpool_bwd_k(const float* __restrict__ dy, const float* __restrict__ x,
                                                     float* __restrict__ dx, const float* __restrict__ inject,
                                                     int apply_mask, int C, int H, int W, int Ho, int Wo)
{
    const size_t total = (size_t)C * Ho * Wo;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const int px = (int)(idx % Wo);
        const int py = (int)((idx / Wo) % Ho);
        const int c = (int)(idx / ((size_t)Wo * Ho));
        const size_t base = (size_t)c * H * W;
        const int r0 = 2 * py, c0 = 2 * px;
        float best = -FLT_MAX;
        int arg = -1;
        float vals[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int r = r0 + (k >> 1), cc = c0 + (k & 1);
            vals[k] = 0.f;
            if (r < H && cc < W) {
                vals[k] = x[base + (size_t)r * W + cc];
                if (vals[k] > best) { best = vals[k]; arg = k; }
            }
        }
        const float g = dy[idx];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int r = r0 + (k >> 1), cc = c0 + (k & 1);
            if (r < H && cc < W) {
                const size_t o = base + (size_t)r * W + cc;
                float v = (k == arg) ? g : 0.f;
                if (apply_mask) v = vals[k] > 0.f ? v : 0.f;
                if (inject) v += inject[o];
                dx[o] = v;
            }
        }
    }
}

hipError_t launch_maxpool_bwd(const float* dy, const float* x, float* dx, const float* inject,
                              int apply_mask, int C, int H, int W, hipStream_t s)
{
    const int Ho = pooled_size(H), Wo = pooled_size(W);
    const size_t total = (size_t)C * Ho * Wo;
    if (H % 2 == 0 && W % 4 == 0) {
        const size_t n_patches = (size_t)C * Ho * (W / 4);
        maxpool_bwd_v4_k<<<reduce_grid(n_patches, 256, 65536), 256, 0, s>>>(dy, x, dx, inject, apply_mask, n_patches, W / 4);
        return hipGetLastError();
    }
    maxpool_bwd_k<<<reduce_grid(total, 256, 65536), 256, 0, s>>>(dy, x, dx, inject, apply_mask, C, H, W, Ho, Wo);
    return hipGetLastError();
}

// ------------------------------------------------------------------------- content / deep-dream
__global__ __launch_bounds__(256) void layer_elem_k(const LayerElemArgs a)
{
    __shared__ float scratch[16];
    float acc[4] = {0.f, 0.f, 0.f, 0.f};   // d2, gc2, f2, gd2
    float inv_c = 0.f, inv_d = 0.f;
    if (a.write) {
        if (a.content) inv_c = *a.norm_c;
        if (a.deepdream) inv_d = *a.norm_d;
    }
    // whole blobs (not tile-sharded) with n % 4 == 0: 16 bytes per lane and access -- the scalar loop below keeps too few bytes
    // in flight to reach the HBM rate (measured 2.3 TB/s at 33 M elements)
    const bool vec = !a.w && (a.n & 3) == 0 && ((reinterpret_cast<uintptr_t>(a.feat) | reinterpret_cast<uintptr_t>(a.inject) |
                                                 reinterpret_cast<uintptr_t>(a.target)) & 15) == 0;
    if (vec) {
        const size_t n4 = a.n >> 2;
        for (size_t q = (size_t)blockIdx.x * 256 + threadIdx.x; q < n4; q += (size_t)gridDim.x * 256) {
            const float4 f4 = reinterpret_cast<const float4*>(a.feat)[q];
            float4 t4 = make_float4(0.f, 0.f, 0.f, 0.f);
            if (a.content) t4 = reinterpret_cast<const float4*>(a.target)[q];
            const float fv[4] = {f4.x, f4.y, f4.z, f4.w}, tv[4] = {t4.x, t4.y, t4.z, t4.w};
            float ov[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {                 // same per-element arithmetic as the scalar loop
                const float f = fv[j];
                float out = 0.f;
                if (a.content) {
                    const float d = f - tv[j];
                    const float gc = a.cn_coef * d;
                    acc[0] += d * d;
                    acc[1] += gc * gc;
                    if (a.write) out += (a.cw * gc) / inv_c;
                }
                if (a.deepdream) {
                    const float gd = a.dn_coef * f;
                    acc[2] += f * f;
                    acc[3] += gd * gd;
                    if (a.write) out += (a.dw * gd) / inv_d;
                }
                ov[j] = out;
            }
            if (a.write) reinterpret_cast<float4*>(a.inject)[q] = make_float4(ov[0], ov[1], ov[2], ov[3]);
        }
    } else
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < a.n; i += (size_t)gridDim.x * 256) {
        if (a.w) {                                        // tile-sharded: only the region of interest counts
            const int x = (int)(i % a.w), y = (int)((i / a.w) % a.h);
            if (y < a.ry0 || y >= a.ry1 || x < a.rx0 || x >= a.rx1) {
                if (a.write) a.inject[i] = 0.f;
                continue;
            }
        }
        const float f = a.feat[i];
        float out = 0.f;
        if (a.content) {
            const float d = f - a.target[i];
            const float gc = a.cn_coef * d;               // (2 / n) * c_diff
            acc[0] += d * d;
            acc[1] += gc * gc;
            if (a.write) out += (a.cw * gc) / inv_c;      // diffs += cw * c_grad / cn
        }
        if (a.deepdream) {
            const float gd = a.dn_coef * f;               // (-2 / n) * F
            acc[2] += f * f;
            acc[3] += gd * gd;
            if (a.write) out += (a.dw * gd) / inv_d;
        }
        if (a.write) a.inject[i] = out;
    }
    block_sum(acc, scratch);
    if (threadIdx.x == 0) {
        if (a.part_d2) a.part_d2[blockIdx.x] = acc[0];
        if (a.part_gc2) a.part_gc2[blockIdx.x] = acc[1];
        if (a.part_f2) a.part_f2[blockIdx.x] = acc[2];
        if (a.part_gd2) a.part_gd2[blockIdx.x] = acc[3];
    }
}

hipError_t launch_layer_elem(const LayerElemArgs& a, int* n_partial, hipStream_t s)
{
    const int grid = reduce_grid(a.n, 256 * 8, kMaxPartials);
    *n_partial = grid;
    layer_elem_k<<<grid, 256, 0, s>>>(a);
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void finalize_norm_k(const float* part, int n_part, double n, float* norm)
{
    __shared__ double scratch[256];
    const double sum = sum_partials(part, n_part, scratch);
    if (threadIdx.x == 0) *norm = sqrtf((float)(sum / n));
}

hipError_t launch_finalize_norm(const float* partial, int n_partial, double n, float* norm, hipStream_t s)
{
    finalize_norm_k<<<1, 256, 0, s>>>(partial, n_partial, n, norm);
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void scaled_accumulate_k(const float* __restrict__ S, float* __restrict__ inject,
                                                           float sw, const float* __restrict__ norm,
                                                           int accumulate, size_t n)
{
    const float coef = sw / *norm;                         // sw / sn[layer]
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float base = accumulate ? inject[i] : 0.f;
        inject[i] = coef * S[i] + base;                    // saxpy
    }
}

hipError_t launch_scaled_accumulate(const float* S, float* inject, float sw, const float* norm,
                                    int accumulate, size_t n, hipStream_t s)
{
    scaled_accumulate_k<<<reduce_grid(n, 256 * 4, 8192), 256, 0, s>>>(S, inject, sw, norm, accumulate, n);
    return hipGetLastError();
}

// ----------------------------------------------------------------- TV + p-norm + combine (+ Adam)
__device__ __forceinline__ float tv_k(float q, float half_beta, int beta_is_2)
{
    // (beta/2) * q^(beta/2 - 1);  q^0 == 1 exactly for beta == 2
    return beta_is_2 ? 1.0f : half_beta * powf(q, half_beta - 1.0f);
}

// One workgroup walks tiles of IP_TY rows x IP_TX columns of one channel.  The tile and its 1-pixel circular neighbourhood are
// staged once in LDS as u = x / 255 (one IEEE division per staged value instead of seven per output, all index arithmetic per
// tile instead of per element); a thread owns one column of the tile.  Per-element arithmetic is the reference's, operation by
// operation (utils.py / worker.py:283-301); only the order of the partial sums differs from a flat sweep.
constexpr int IP_TY = 4, IP_TX = 256;
constexpr int IP_LW = IP_TX + 2;                         // staged row length

__global__ __launch_bounds__(256) void image_pass_k(const ImagePassArgs a)
{
    __shared__ float scratch[32];
    __shared__ float u_s[(IP_TY + 2) * IP_LW];
    const int H = a.H, W = a.W, tid = threadIdx.x;
    const size_t plane = (size_t)H * W;
    const float half_beta = a.tv_beta * 0.5f;
    const int beta_is_2 = a.tv_beta == 2.0f;
    // per-step Adam scalars: by value, or (graph replay: the launch arguments are frozen) from device memory
    const float corr1 = a.dyn ? a.dyn[0] : a.corr1, corr2 = a.dyn ? a.dyn[1] : a.corr2, step = a.dyn ? a.dyn[2] : a.step;
    const int p_round = (int)a.p_pow;
    const int p_int = ((float)p_round == a.p_pow && p_round >= 1 && p_round <= 16) ? p_round : 0;      // uniform
    const int tiles_x = (W + IP_TX - 1) / IP_TX, tiles_y = (H + IP_TY - 1) / IP_TY;
    const int n_tiles = a.C * tiles_y * tiles_x;
    float acc[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int tx = tile % tiles_x, rest = tile / tiles_x;
        const int ty = rest % tiles_y, ch = rest / tiles_y;
        const int x0 = tx * IP_TX, y0 = ty * IP_TY;
        const float* p = a.x + (size_t)ch * plane;
        __syncthreads();                                 // the previous tile is consumed
        for (int e = tid; e < (IP_TY + 2) * IP_LW; e += 256) {
            const int r = e / IP_LW, c = e - r * IP_LW;
            int gy = y0 - 1 + r, gx = x0 - 1 + c;        // circular neighbourhood (np.roll)
            gy = gy < 0 ? H - 1 : (gy >= H ? gy - H : gy);
            gx = gx < 0 ? W - 1 : (gx >= W ? gx - W : gx);
            u_s[e] = (gy < H && gx < W) ? p[(size_t)gy * W + gx] / 255.0f : 0.f;      // (beyond the wrap: slots no output reads)
        }
        __syncthreads();
        const int x = x0 + tid;
        if (x < W) {
#pragma unroll
            for (int ry = 0; ry < IP_TY; ++ry) {
                const int y = y0 + ry;
                if (y >= H) break;
                const float* us = u_s + (ry + 1) * IP_LW + tid + 1;      // (y, x)
                const size_t idx = (size_t)ch * plane + (size_t)y * W + x;
                const float xv = a.x[idx];
                const float u = us[0];
                const float u_r = us[1], u_l = us[-1];
                const float u_d = us[IP_LW], u_u = us[-IP_LW];
                const float u_dl = us[IP_LW - 1], u_ur = us[-IP_LW + 1];
                // this pixel
                const float a0 = u - u_r, b0 = u - u_d;
                const float q0 = (a0 * a0 + b0 * b0) + 1e-8f;
                const float k0 = tv_k(q0, half_beta, beta_is_2);
                const float da0 = (2.0f * a0) * k0, db0 = (2.0f * b0) * k0;
                // left neighbour's x-difference, upper neighbour's y-difference
                const float aL = u_l - u, bL = u_l - u_dl;
                const float qL = (aL * aL + bL * bL) + 1e-8f;
                const float daL = (2.0f * aL) * tv_k(qL, half_beta, beta_is_2);
                const float aU = u_u - u_ur, bU = u_u - u;
                const float qU = (aU * aU + bU * bU) + 1e-8f;
                const float dbU = (2.0f * bU) * tv_k(qU, half_beta, beta_is_2);
                float g_tv = da0 + db0;
                g_tv -= daL;
                g_tv -= dbU;
                acc[0] += beta_is_2 ? q0 : powf(q0, half_beta);
                // p-norm
                const float mag = fabsf(u);
                const float sgn = u > 0.f ? 1.f : (u < 0.f ? -1.f : 0.f);
                float pw, pw1;                           // |u|^p, |u|^(p-1)
                if (p_int) {
                    // integral exponent (the default p = 6): repeated multiplication, within 3 ulp of powf -- two powf calls per
                    // element were 60 % of this kernel's time (it is VALU-bound, not HBM-bound, with them)
                    pw1 = p_int > 1 ? mag : 1.0f;
                    for (int k = 2; k < p_int; ++k) pw1 *= mag;
                    pw = pw1 * mag;
                } else {
                    pw = powf(mag, a.p_pow);
                    pw1 = powf(mag, a.p_pow - 1.0f);
                }
                acc[1] += pw;
                const float g_p = sgn * pw1;
                // combine
                const float scd = a.scd ? a.scd[idx] : 0.f;
                const float tg = a.tv_w * g_tv;
                const float pg = a.p_w * g_p;
                float g = scd + tg;
                g += pg;
                acc[2] += scd * scd;
                acc[3] += tg * tg;
                acc[4] += pg * pg;
                acc[5] += g * g;
                if (a.grad) a.grad[idx] = g;
                if (a.x_out) {
                    const float m_old = a.m_is_zero ? 0.f : a.m[idx];
                    const float v_old = a.v_is_zero ? 0.f : a.v[idx];
                    const float m_new = a.d1 * m_old + a.c1 * g;
                    const float v_new = a.d2 * v_old + a.c2 * (g * g);
                    a.m[idx] = m_new;
                    a.v[idx] = v_new;
                    const float m_hat = m_new / corr1;
                    const float v_hat = v_new / corr2;
                    a.x_out[idx] = xv - (step * m_hat) / (sqrtf(v_hat) + 1e-8f);
                }
            }
        }
    }
    block_sum(acc, scratch);
    if (threadIdx.x == 0) {
#pragma unroll
        for (int i = 0; i < 6; ++i) a.partial[i * kMaxPartials + blockIdx.x] = acc[i];
    }
}

__global__ void set_scalars3_k(float* dst, float a, float b, float c) { dst[0] = a; dst[1] = b; dst[2] = c; }
hipError_t launch_set_scalars3(float* dst, float a, float b, float c, hipStream_t s)
{
    set_scalars3_k<<<1, 1, 0, s>>>(dst, a, b, c);
    return hipGetLastError();
}

hipError_t launch_image_pass(const ImagePassArgs& a, int* n_partial, hipStream_t s)
{
    const long long n_tiles = (long long)a.C * ((a.H + IP_TY - 1) / IP_TY) * ((a.W + IP_TX - 1) / IP_TX);
    const int grid = (int)(n_tiles < kMaxPartials ? (n_tiles < 1 ? 1 : n_tiles) : kMaxPartials);
    *n_partial = grid;
    image_pass_k<<<grid, 256, 0, s>>>(a);
    return hipGetLastError();
}

// Tile-sharded image pass: same arithmetic, neighbours outside the tile come from the gathered ring.
__global__ __launch_bounds__(256) void image_pass_tile_k(const ImageTileArgs t)
{
    __shared__ float scratch[32];
    const ImagePassArgs& a = t.base;
    const int th = t.th, tw = t.tw, ww = a.W;
    const size_t wplane = (size_t)a.H * a.W, tplane = (size_t)th * tw, rplane = (size_t)(th + 2) * (tw + 2);
    const size_t total = tplane * 3;
    const float half_beta = a.tv_beta * 0.5f;
    const int beta_is_2 = a.tv_beta == 2.0f;
    float acc[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const int x = (int)(idx % tw);
        const int y = (int)((idx / tw) % th);
        const int c = (int)(idx / tplane);
        const float* p = a.x + c * wplane;
        const float* rg = t.ring + c * rplane;
        auto U = [&](int yy, int xx) -> float {            // yy in [-1, th], xx in [-1, tw]
            const bool inside = yy >= 0 && yy < th && xx >= 0 && xx < tw;
            const float v = inside ? p[(size_t)(t.ty + yy) * ww + t.tx + xx] : rg[(size_t)(yy + 1) * (tw + 2) + xx + 1];
            return v / 255.0f;
        };
        const size_t widx = c * wplane + (size_t)(t.ty + y) * ww + t.tx + x;
        const float xv = p[(size_t)(t.ty + y) * ww + t.tx + x];
        const float u = xv / 255.0f;
        const float u_r = U(y, x + 1), u_d = U(y + 1, x), u_l = U(y, x - 1), u_u = U(y - 1, x);
        const float u_dl = U(y + 1, x - 1), u_ur = U(y - 1, x + 1);
        const float a0 = u - u_r, b0 = u - u_d;
        const float q0 = (a0 * a0 + b0 * b0) + 1e-8f;
        const float k0 = tv_k(q0, half_beta, beta_is_2);
        const float da0 = (2.0f * a0) * k0, db0 = (2.0f * b0) * k0;
        const float aL = u_l - u, bL = u_l - u_dl;
        const float qL = (aL * aL + bL * bL) + 1e-8f;
        const float daL = (2.0f * aL) * tv_k(qL, half_beta, beta_is_2);
        const float aU = u_u - u_ur, bU = u_u - u;
        const float qU = (aU * aU + bU * bU) + 1e-8f;
        const float dbU = (2.0f * bU) * tv_k(qU, half_beta, beta_is_2);
        float g_tv = da0 + db0;
        g_tv -= daL;
        g_tv -= dbU;
        acc[0] += beta_is_2 ? q0 : powf(q0, half_beta);
        const float mag = fabsf(u);
        const float sgn = u > 0.f ? 1.f : (u < 0.f ? -1.f : 0.f);
        acc[1] += powf(mag, a.p_pow);
        const float g_p = sgn * powf(mag, a.p_pow - 1.0f);
        const float scd = a.scd ? a.scd[widx] : 0.f;
        const float tg = a.tv_w * g_tv;
        const float pg = a.p_w * g_p;
        float g = scd + tg;
        g += pg;
        acc[2] += scd * scd;
        acc[3] += tg * tg;
        acc[4] += pg * pg;
        acc[5] += g * g;
        if (a.grad) a.grad[widx] = g;
        if (a.x_out) {
            const float m_old = a.m_is_zero ? 0.f : a.m[widx];
            const float v_old = a.v_is_zero ? 0.f : a.v[widx];
            const float m_new = a.d1 * m_old + a.c1 * g;
            const float v_new = a.d2 * v_old + a.c2 * (g * g);
            a.m[widx] = m_new;
            a.v[widx] = v_new;
            const float m_hat = m_new / a.corr1;
            const float v_hat = v_new / a.corr2;
            a.x_out[widx] = xv - (a.step * m_hat) / (sqrtf(v_hat) + 1e-8f);
        }
    }
    block_sum(acc, scratch);
    if (threadIdx.x == 0) {
#pragma unroll
        for (int i = 0; i < 6; ++i) a.partial[i * kMaxPartials + blockIdx.x] = acc[i];
    }
}

hipError_t launch_image_pass_tile(const ImageTileArgs& a, int* n_partial, hipStream_t s)
{
    const size_t total = (size_t)3 * a.th * a.tw;
    const int grid = reduce_grid(total, 256 * 4, kMaxPartials);
    *n_partial = grid;
    image_pass_tile_k<<<grid, 256, 0, s>>>(a);
    return hipGetLastError();
}

// ---------------------------------------------------------------------- Pillow-exact plane resampling
// Pillow's ImagingResample for 32-bit float planes (reference utils.py:130-160 -> Image.resize): separable,
// horizontal pass then vertical pass, per output coordinate a window [xmin, xmin + ksize) of normalised double
// coefficients (computed on the host exactly as Pillow's precompute_coeffs), double accumulation in window order,
// float32 intermediate.  This file is built with -ffp-contract=off: mul and add round separately, as in Pillow.
__global__ __launch_bounds__(256) void resample_h_k(const float* __restrict__ src, float* __restrict__ dst, size_t rows,
                                                    int w_in, int w_out, ResampleTable t, int clamp0)
{
    const size_t total = rows * w_out;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const int xx = (int)(idx % w_out);
        const size_t row = idx / w_out;
        const float* p = src + row * w_in + t.lo[xx];
        const double* k = t.k + (size_t)xx * t.kmax;
        double ss = 0.0;
        for (int i = 0; i < t.n[xx]; ++i) ss += (double)p[i] * k[i];
        float v = (float)ss;
        if (clamp0) v = v > 0.f ? v : 0.f;
        dst[idx] = v;
    }
}

__global__ __launch_bounds__(256) void resample_v_k(const float* __restrict__ src, float* __restrict__ dst, int planes,
                                                    int h_in, int h_out, int w, ResampleTable t, int clamp0)
{
    const size_t total = (size_t)planes * h_out * w;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const int x = (int)(idx % w);
        const int yy = (int)((idx / w) % h_out);
        const size_t pl = idx / ((size_t)w * h_out);
        const float* p = src + (pl * h_in + t.lo[yy]) * w + x;
        const double* k = t.k + (size_t)yy * t.kmax;
        double ss = 0.0;
        for (int i = 0; i < t.n[yy]; ++i) ss += (double)p[(size_t)i * w] * k[i];
        float v = (float)ss;
        if (clamp0) v = v > 0.f ? v : 0.f;
        dst[idx] = v;
    }
}

__global__ __launch_bounds__(256) void clamp0_copy_k(const float* src, float* dst, size_t n, int clamp0)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        dst[i] = clamp0 ? (src[i] > 0.f ? src[i] : 0.f) : src[i];
}

hipError_t launch_resample(const float* src, float* tmp, float* dst, int planes, int h_in, int w_in, int h_out, int w_out,
                           const ResampleTable& tx, const ResampleTable& ty, int clamp0, hipStream_t s)
{
    // Pillow skips a pass whose size does not change
    const bool need_h = w_in != w_out, need_v = h_in != h_out;
    const float* cur = src;
    if (need_h) {
        float* out = need_v ? tmp : dst;
        const size_t rows = (size_t)planes * h_in;
        resample_h_k<<<reduce_grid(rows * w_out, 256, 16384), 256, 0, s>>>(cur, out, rows, w_in, w_out, tx, need_v ? 0 : clamp0);
        cur = out;
    }
    if (need_v) resample_v_k<<<reduce_grid((size_t)planes * h_out * w_out, 256, 16384), 256, 0, s>>>(cur, dst, planes, h_in, h_out, w_out, ty, clamp0);
    if (!need_h && !need_v) clamp0_copy_k<<<reduce_grid((size_t)planes * h_in * w_in, 1024, 8192), 256, 0, s>>>(src, dst, (size_t)planes * h_in * w_in, clamp0);
    return hipGetLastError();
}

// -------------------------------------------------------------------------------- pre / deprocess
__constant__ float kMean[3] = {123.68f, 116.779f, 103.939f};   // worker.py:34

template <typename T>
__global__ __launch_bounds__(256) void preprocess_k(const T* __restrict__ hwc, float* __restrict__ nchw, int H, int W)
{
    const size_t plane = (size_t)H * W;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < plane * 3; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i / plane);
        const size_t p = i % plane;
        nchw[i] = (float)hwc[p * 3 + c] - kMean[c];
    }
}

__global__ __launch_bounds__(256) void deprocess_k(const float* __restrict__ nchw, float* __restrict__ hwc, int H, int W)
{
    const size_t plane = (size_t)H * W;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < plane * 3; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i % 3);
        const size_t p = i / 3;
        hwc[i] = nchw[(size_t)c * plane + p] + kMean[c];
    }
}

hipError_t launch_preprocess_u8(const uint8_t* hwc, float* nchw, int H, int W, hipStream_t s)
{
    preprocess_k<uint8_t><<<reduce_grid((size_t)H * W * 3, 1024, 8192), 256, 0, s>>>(hwc, nchw, H, W);
    return hipGetLastError();
}
hipError_t launch_preprocess_f32(const float* hwc, float* nchw, int H, int W, hipStream_t s)
{
    preprocess_k<float><<<reduce_grid((size_t)H * W * 3, 1024, 8192), 256, 0, s>>>(hwc, nchw, H, W);
    return hipGetLastError();
}
hipError_t launch_deprocess(const float* nchw, float* hwc, int H, int W, hipStream_t s)
{
    deprocess_k<<<reduce_grid((size_t)H * W * 3, 1024, 8192), 256, 0, s>>>(nchw, hwc, H, W);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------- trace scalars
// Stage 1: one workgroup per partial-sum slot (layer x 6 + 6 image slots) -> a.sums[slot] (double).
__global__ __launch_bounds__(256) void trace_sums_k(const TraceArgs a)
{
    __shared__ double scratch[256];
    const int slot = blockIdx.x;
    const int nl = a.n_layers * kLayerSlots;
    double v = 0.0;
    if (slot < nl) {
        const TraceLayer& L = a.layer[slot / kLayerSlots];
        const int k = slot % kLayerSlots;
        if (L.part[k] && L.count[k] > 0) v = sum_partials(L.part[k], L.count[k], scratch);
    } else {
        v = sum_partials(a.image_part + (slot - nl) * kMaxPartials, a.image_count, scratch);
    }
    if (threadIdx.x == 0) a.sums[slot] = v;
}

// Stage 2: the scalar arithmetic of the trace, in the reference's fp32 order (worker.py:249-301).
__global__ void finalize_trace_k(const TraceArgs a)
{
    if (threadIdx.x != 0) return;
    const double* sums = a.sums;
    float loss = 0.f;
    for (int l = 0; l < a.n_layers; ++l) {
        const TraceLayer& L = a.layer[l];
        const double* sm = sums + l * kLayerSlots;
        float* o = a.out + l * 6;
        for (int k = 0; k < 6; ++k) o[k] = 0.f;
        if (L.content) {
            const float nc = L.norm[0];
            const float c_loss = (L.cw * (float)(sm[0] / L.n)) / nc;      // cw * mean(d^2) / cn
            loss += c_loss;
            o[0] = c_loss;
            o[1] = (fabsf(L.cw) * sqrtf((float)(sm[1] / L.n))) / nc;      // rms(cw * c_grad / cn)
        }
        if (L.style) {
            const float ns = L.norm[1];
            const float s_loss = (L.sw * (float)(sm[4] / L.gram_n)) / ns; // sw * mean(D^2) / sn
            loss += s_loss;
            o[2] = s_loss;
            o[3] = fabsf(L.sw / ns) * sqrtf((float)(sm[5] / L.n));        // rms(sw / sn * s_grad)
        }
        if (L.deepdream) {
            const float nd = L.norm[2];
            const float d_loss = (-L.dw * (float)(sm[2] / L.n)) / nd;     // -dw * mean(F^2) / dn
            loss += d_loss;
            o[4] = d_loss;
            o[5] = (fabsf(L.dw) * sqrtf((float)(sm[3] / L.n))) / nd;
        }
    }
    const double* im = sums + a.n_layers * kLayerSlots;
    float* g = a.out + a.n_layers * 6;
    g[0] = loss;                                               // scd_loss
    const float t_loss = a.tv_w * (float)im[0];
    loss += t_loss;
    const float p_loss = a.p_w * ((float)im[1] / a.p_pow);
    loss += p_loss;
    g[1] = t_loss;
    g[2] = p_loss;
    g[3] = a.have_grad ? sqrtf((float)(im[2] / a.image_n)) : 0.f;
    g[4] = a.have_grad ? sqrtf((float)(im[3] / a.image_n)) : 0.f;
    g[5] = a.have_grad ? sqrtf((float)(im[4] / a.image_n)) : 0.f;
    g[6] = loss;
    g[7] = a.have_grad ? sqrtf((float)(im[5] / a.image_n)) : 0.f;
}

hipError_t launch_finalize_trace(const TraceArgs& a, hipStream_t s)
{
    trace_sums_k<<<a.n_layers * kLayerSlots + kImageSlots, 256, 0, s>>>(a);
    finalize_trace_k<<<1, 64, 0, s>>>(a);
    return hipGetLastError();
}

// ------------------------------------------------------------------ tile-sharded mode: strip pack / unpack
// All rectangles one neighbour receives travel as ONE contiguous buffer: rect r of a (C, wh, ww) window tensor is stored as
// (C, h_r, w_r) at offset off_r.  mode 0: pack (tensor -> buffer), 1: unpack (buffer -> tensor), 2: unpack and ADD.
__global__ __launch_bounds__(256) void strip_copy_k(float* __restrict__ tensor, float* __restrict__ buf, const StripTable t, int C, int wh, int ww, int mode)
{
    const size_t plane = (size_t)wh * ww;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < (size_t)t.total; idx += (size_t)gridDim.x * 256) {
        int r = 0;
#pragma unroll
        for (int k = 1; k < kMaxStripRects; ++k) r += (k < t.n && idx >= (size_t)t.off[k]) ? 1 : 0;     // rects are stored in offset order
        const size_t e = idx - t.off[r];
        const int w = t.w[r], h = t.h[r];
        const int x = (int)(e % w), y = (int)((e / w) % h), c = (int)(e / ((size_t)w * h));
        float* tp = tensor + (size_t)c * plane + (size_t)(t.y0[r] + y) * ww + t.x0[r] + x;
        if (mode == 0) buf[idx] = *tp;
        else if (mode == 1) *tp = buf[idx];
        else *tp += buf[idx];
    }
}

hipError_t launch_strip_copy(float* tensor, float* buf, const StripTable& t, int C, int wh, int ww, int mode, hipStream_t s)
{
    if (t.n <= 0 || t.total <= 0) return hipSuccess;
    const int grid = reduce_grid((size_t)t.total, 256 * 4, 2048);
    strip_copy_k<<<grid, 256, 0, s>>>(tensor, buf, t, C, wh, ww, mode);
    return hipGetLastError();
}

// ---------------------------------------------------------------------- deterministic final sums
__global__ __launch_bounds__(256) void dot_final_k(const float* part, int n_part, float* out)
{
    __shared__ double scratch[256];
    const double sum = sum_partials(part, n_part, scratch);
    if (threadIdx.x == 0) *out = (float)sum;
}

hipError_t launch_sum_partials(const float* part, int n, float* out, hipStream_t s)
{
    dot_final_k<<<1, 256, 0, s>>>(part, n, out);
    return hipGetLastError();
}

// ---------------------------------------------------------------------- BLAS-1 pieces of the tile-sharded L-BFGS
// (optimizers.py:89-108 with utils.dot / utils.axpy = sdot / saxpy): every rank holds its tile of each vector, the dot products
// are per-rank partial sums that the caller all-reduces.  HBM-bound, 4 or 8 bytes read per element.
__global__ __launch_bounds__(256) void vec_dot_k(const float* __restrict__ a, const float* __restrict__ b, size_t n, float* __restrict__ part)
{
    __shared__ float scratch[8];
    float acc[1] = {0.f};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) acc[0] += a[i] * b[i];
    block_sum(acc, scratch);
    if (threadIdx.x == 0) part[blockIdx.x] = acc[0];
}

hipError_t launch_vec_dot(const float* a, const float* b, size_t n, float* part, float* out, hipStream_t s)
{
    const int grid = reduce_grid(n, 256 * 8, kMaxPartials);
    vec_dot_k<<<grid, 256, 0, s>>>(a, b, n, part);
    dot_final_k<<<1, 256, 0, s>>>(part, grid, out);
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void vec_axpy_k(float alpha, const float* __restrict__ x, float* __restrict__ y, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) y[i] = alpha * x[i] + y[i];
}

hipError_t launch_vec_axpy(float alpha, const float* x, float* y, size_t n, hipStream_t s)
{
    vec_axpy_k<<<reduce_grid(n, 256 * 4, 8192), 256, 0, s>>>(alpha, x, y, n);
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void vec_div_k(double divisor, float* __restrict__ y, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) y[i] = (float)((double)y[i] / divisor);
}

hipError_t launch_vec_div(double divisor, float* y, size_t n, hipStream_t s)
{
    vec_div_k<<<reduce_grid(n, 256 * 4, 8192), 256, 0, s>>>(divisor, y, n);
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void lincomb_k(float a, const float* x, float b, const float* y, float* z, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        float v = a * x[i];
        if (y) v += b * y[i];
        z[i] = v;
    }
}

hipError_t launch_lincomb(float a, const float* x, float b, const float* y, float* z, size_t n, hipStream_t s)
{
    lincomb_k<<<reduce_grid(n, 256 * 4, 8192), 256, 0, s>>>(a, x, b, y, z, n);
    return hipGetLastError();
}

}  // namespace st2
