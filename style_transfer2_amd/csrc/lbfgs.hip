// Fixed-step L-BFGS (reference optimizers.py:49-125) as a device-resident state machine on gfx950.
//
// The two-loop recursion of optimizers.py:89-108 is a chain  dot -> scalar -> axpy -> dot -> ...  over vectors of
// N3 = 3 H W floats (50 MB at 2048^2).  Here every link of the chain is ONE streaming kernel that
//   * finishes the previous link's dot product in its prologue (every workgroup sums the <= 1024 per-block partials
//     in double, in a fixed order: bitwise reproducible) and forms the scalar (alpha, beta, the H0 scale) exactly as
//     the reference does -- Python doubles, rounded to fp32 where they meet the fp32 arrays (utils.py:29-46);
//   * applies this link's axpy and, in the same pass over the data, accumulates the NEXT link's dot product
//     ("axpy_i (+) dot_{i+1}"): 3 reads + 1 write per link instead of 5 streams in 3 launches.
// The last link also forms s = -step * p and x += s (optimizers.py:68-69).
// The history ring (<= 10 pairs + 1 scratch slot), the pair count, the s^T y > 1e-10 gate (optimizers.py:82) and the
// eviction of the oldest pair (:84-85) live in device memory (LbfgsDev) and are updated by lbfgs_commit_k, so a step
// needs NO host read-back: the host enqueues the same 24 launches every step and the links beyond the current pair
// count return at once.  HBM-bound: (8 k + 3) N3 x 4 bytes per two-loop with k pairs; 4 N3 x 4 for the pair update.
// Built with -ffp-contract=off (one rounding per operation, like the NumPy / BLAS-1 reference).
#include <hip/hip_runtime.h>
#include "wave_reduce.h"
#include "st2_kernels.h"

namespace st2 {

namespace {

constexpr int kLbGrid = kMaxPartials;        // workgroups (= partial-sum slots) of every link

struct VecSpan { size_t n4, n; };            // float4 part and total length

__device__ __forceinline__ float4 ld4(const float* p, size_t i) { return reinterpret_cast<const float4*>(p)[i]; }
__device__ __forceinline__ void st4(float* p, size_t i, const float4& v) { reinterpret_cast<float4*>(p)[i] = v; }

// sum of the previous link's per-block partials, identical in every thread of every workgroup
__device__ __forceinline__ double link_sum(const float* part)
{
    __shared__ double scratch[256];
    return sum_partials(part, kLbGrid, scratch);
}

__device__ __forceinline__ void store_partial(float acc, float* part)
{
    __shared__ float red[4];
    float v[1] = {acc};
    block_sum(v, red);
    if (threadIdx.x == 0) part[blockIdx.x] = v[0];
}

}  // namespace

// ---- link 0: the first dot product of the recursion: s_newest . g   (or g . g when the history is empty)
__global__ __launch_bounds__(256) void lbfgs_first_dot_k(const LbfgsArgs a)
{
    const LbfgsDev* st = a.st;
    const int count = st->count;
    const float* u = count > 0 ? a.v.s[st->order[count - 1]] : a.g;
    const size_t n4 = a.n / 4;
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const float4 x = ld4(u, i), y = ld4(a.g, i);
        acc += x.x * y.x; acc += x.y * y.y; acc += x.z * y.z; acc += x.w * y.w;
    }
    if (blockIdx.x == 0 && threadIdx.x < (a.n & 3)) { const size_t i = n4 * 4 + threadIdx.x; acc += u[i] * a.g[i]; }
    store_partial(acc, a.part);
}

// ---- first loop, position j (0 = newest pair): alpha = (s.p) / sy ; p -= alpha y ; [last: p *= sy_newest / (y.y)_newest]
//      fused: the next link's dot product (s of the next-older pair, or y of the oldest pair for the second loop)
__global__ __launch_bounds__(256) void lbfgs_loop1_k(const LbfgsArgs a, const int j)
{
    LbfgsDev* st = a.st;
    const int count = st->count;
    if (j >= count) return;
    const int slot = st->order[count - 1 - j];
    const double alpha = link_sum(a.part + (j & 1) * kMaxPartials) / st->sy[slot];          // sdot(s, p) / sy
    if (blockIdx.x == 0 && threadIdx.x == 0) st->alpha[slot] = alpha;
    const float na = (float)(-alpha);                                                       // saxpy(-alpha, y, p)
    const bool last = j == count - 1;
    const int newest = st->order[count - 1];
    const float scale = last ? (float)(st->sy[newest] / st->yy[newest]) : 1.f;               // p *= sy / sdot(y, y)
    const float* src = j == 0 ? a.g : a.p;
    const float* y = a.v.y[slot];
    const float* nxt = last ? a.v.y[st->order[0]] : a.v.s[st->order[count - 2 - j]];
    const size_t n4 = a.n / 4;
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const float4 p = ld4(src, i), yv = ld4(y, i), q = ld4(nxt, i);
        float4 t;
        t.x = na * yv.x + p.x; t.y = na * yv.y + p.y; t.z = na * yv.z + p.z; t.w = na * yv.w + p.w;
        if (last) { t.x *= scale; t.y *= scale; t.z *= scale; t.w *= scale; }
        st4(a.p, i, t);
        acc += q.x * t.x; acc += q.y * t.y; acc += q.z * t.z; acc += q.w * t.w;
    }
    if (blockIdx.x == 0 && threadIdx.x < (a.n & 3)) {
        const size_t i = n4 * 4 + threadIdx.x;
        float t = na * y[i] + src[i];
        if (last) t *= scale;
        a.p[i] = t;
        acc += nxt[i] * t;
    }
    store_partial(acc, a.part + ((j + 1) & 1) * kMaxPartials);
}

// ---- second loop, position i (0 = oldest pair): beta = (y.p) / sy ; p += (alpha - beta) s
//      fused: the next link's dot product, or -- on the last link -- s_new = -step p ; x += s_new
__global__ __launch_bounds__(256) void lbfgs_loop2_k(const LbfgsArgs a, const int i)
{
    const LbfgsDev* st = a.st;
    const int count = st->count;
    if (i >= count) return;
    const int slot = st->order[i];
    const double beta = link_sum(a.part + ((count + i) & 1) * kMaxPartials) / st->sy[slot];
    const float coef = (float)(st->alpha[slot] - beta);                                      // saxpy(alpha - beta, s, p)
    const bool last = i == count - 1;
    const float* s = a.v.s[slot];
    const float* nxt = last ? nullptr : a.v.y[st->order[i + 1]];
    float* s_new = a.v.s[st->free_slot];
    const float nstep = -a.step;
    const bool apply = last && a.apply;
    const size_t n4 = a.n / 4;
    float acc = 0.f;
    for (size_t k = (size_t)blockIdx.x * 256 + threadIdx.x; k < n4; k += (size_t)gridDim.x * 256) {
        const float4 p = ld4(a.p, k), sv = ld4(s, k);
        float4 t;
        t.x = coef * sv.x + p.x; t.y = coef * sv.y + p.y; t.z = coef * sv.z + p.z; t.w = coef * sv.w + p.w;
        if (!apply) st4(a.p, k, t);
        if (!last) {
            const float4 q = ld4(nxt, k);
            acc += q.x * t.x; acc += q.y * t.y; acc += q.z * t.z; acc += q.w * t.w;
        }
        if (apply) {                                                                        // s = -step * inv_hv(g) ; x += s
            float4 sn, xv = ld4(a.x, k);
            sn.x = nstep * t.x; sn.y = nstep * t.y; sn.z = nstep * t.z; sn.w = nstep * t.w;
            xv.x += sn.x; xv.y += sn.y; xv.z += sn.z; xv.w += sn.w;
            st4(s_new, k, sn);
            st4(a.x, k, xv);
        }
    }
    if (blockIdx.x == 0 && threadIdx.x < (a.n & 3)) {
        const size_t k = n4 * 4 + threadIdx.x;
        const float t = coef * s[k] + a.p[k];
        if (!apply) a.p[k] = t;
        if (!last) acc += nxt[k] * t;
        if (apply) { const float sn = nstep * t; s_new[k] = sn; a.x[k] += sn; }
    }
    if (!last) store_partial(acc, a.part + ((count + i + 1) & 1) * kMaxPartials);
}

// ---- empty history: p = g / sqrt(g.g / N3)  (unit-RMS direction, optimizers.py:98-99), then the update
__global__ __launch_bounds__(256) void lbfgs_apply0_k(const LbfgsArgs a)
{
    const LbfgsDev* st = a.st;
    if (st->count != 0) return;
    const double r = sqrt(link_sum(a.part) / (double)a.n);            // np.sqrt(sdot(p, p) / p.size): a float64 scalar
    float* s_new = a.v.s[st->free_slot];
    const float nstep = -a.step;
    for (size_t k = (size_t)blockIdx.x * 256 + threadIdx.x; k < a.n; k += (size_t)gridDim.x * 256) {
        const float t = (float)((double)a.g[k] / r);
        if (a.apply) { const float sn = nstep * t; s_new[k] = sn; a.x[k] += sn; }
        else a.p[k] = t;
    }
}

// ---- the new pair: y = g_new - g_old (mode 0; mode 1: y is already in place), partials of s.y and y.y
__global__ __launch_bounds__(256) void lbfgs_pair_k(const LbfgsArgs a, const float* g_new, const int mode)
{
    const LbfgsDev* st = a.st;
    const int slot = st->free_slot;
    const float* s = a.v.s[slot];
    float* y = a.v.y[slot];
    float sy = 0.f, yy = 0.f;
    const size_t n4 = a.n / 4;
    for (size_t k = (size_t)blockIdx.x * 256 + threadIdx.x; k < n4; k += (size_t)gridDim.x * 256) {
        float4 v;
        if (mode == 0) {
            const float4 gn = ld4(g_new, k), go = ld4(a.g, k);
            v.x = gn.x - go.x; v.y = gn.y - go.y; v.z = gn.z - go.z; v.w = gn.w - go.w;
            st4(y, k, v);
        } else v = ld4(y, k);
        const float4 sv = ld4(s, k);
        sy += sv.x * v.x; sy += sv.y * v.y; sy += sv.z * v.z; sy += sv.w * v.w;
        yy += v.x * v.x; yy += v.y * v.y; yy += v.z * v.z; yy += v.w * v.w;
    }
    if (blockIdx.x == 0 && threadIdx.x < (a.n & 3)) {
        const size_t k = n4 * 4 + threadIdx.x;
        float v;
        if (mode == 0) { v = g_new[k] - a.g[k]; y[k] = v; } else v = y[k];
        sy += s[k] * v;
        yy += v * v;
    }
    __shared__ float red[8];
    float v2[2] = {sy, yy};
    block_sum(v2, red);
    if (threadIdx.x == 0) { a.part2[blockIdx.x] = v2[0]; a.part2[kMaxPartials + blockIdx.x] = v2[1]; }
}

// ---- store_curvature_pair (optimizers.py:79-87): keep the pair iff s.y > 1e-10, drop the oldest beyond n_corr
__global__ __launch_bounds__(256) void lbfgs_commit_k(const LbfgsArgs a)
{
    __shared__ double scratch[256];
    LbfgsDev* st = a.st;
    const double sy = (double)(float)sum_partials(a.part2, kLbGrid, scratch);      // sdot returns an fp32 value
    const double yy = (double)(float)sum_partials(a.part2 + kMaxPartials, kLbGrid, scratch);
    if (threadIdx.x != 0) return;
    int count = st->count;
    const int slot = st->free_slot;
    if (sy > 1e-10) {
        st->sy[slot] = sy;
        st->yy[slot] = yy;
        st->order[count++] = slot;
        if (count > kLbfgsCorr) {
            for (int k = 0; k + 1 < count; ++k) st->order[k] = st->order[k + 1];
            --count;
        }
        st->count = count;
        for (int cand = 0; cand < kLbfgsSlots; ++cand) {                // the slot no live pair uses is the next scratch slot
            bool used = false;
            for (int k = 0; k < count; ++k) used = used || st->order[k] == cand;
            if (!used) { st->free_slot = cand; break; }
        }
    }
    st->last_sy = sy;
}

hipError_t launch_lbfgs_two_loop(const LbfgsArgs& a, hipStream_t s)
{
    lbfgs_first_dot_k<<<kLbGrid, 256, 0, s>>>(a);
    for (int j = 0; j < kLbfgsCorr; ++j) lbfgs_loop1_k<<<kLbGrid, 256, 0, s>>>(a, j);
    for (int i = 0; i < kLbfgsCorr; ++i) lbfgs_loop2_k<<<kLbGrid, 256, 0, s>>>(a, i);
    lbfgs_apply0_k<<<kLbGrid, 256, 0, s>>>(a);
    return hipGetLastError();
}

hipError_t launch_lbfgs_pair(const LbfgsArgs& a, const float* g_new, int mode, hipStream_t s)
{
    lbfgs_pair_k<<<kLbGrid, 256, 0, s>>>(a, g_new, mode);
    lbfgs_commit_k<<<1, 256, 0, s>>>(a);
    return hipGetLastError();
}

}  // namespace st2
