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

// ================================================================================================ Gram form
// inv_hv(g) of optimizers.py:89-108 lies in span{s_i, y_i, g}: with B = the matrix of inner products of those 2 k + 1 vectors
// the two loops run on coefficient vectors (every "sdot(s_i, p)" is sum_l delta_l B[l][s_i], every "saxpy" moves one
// coefficient), in double, by one thread -- and the N3-long vectors are touched twice per step: one pass that forms the candidate
// y and takes the 2 (2 k + 3) inner products the new pair and the new gradient add to B, one pass that forms the direction as a
// linear combination, s = -step p and x += s.  (8 k + 7) N3 x 4 bytes per step become (4 k + 10) N3 x 4: 0.73 -> 0.37 ms at
// 2048^2 with ten pairs.  Same mathematics as the chain above, different rounding (the chain rounds p to fp32 after every one of
// its 2 k axpys; here p is one fp32 linear combination with double coefficients), so this form is the default only where the
// objective itself is approximate (the bf16 feature path); ST2_LBFGS_FORM=chain|gram overrides.  The inner products of the
// candidate s are not streamed: s = -step sum_l delta_l b_l, so s.b_id = -step sum_l delta_l B[l][id] -- except s.y, which decides
// the gate and the H0 scale and is taken directly, like y.y (both are differences of nearly equal quantities otherwise).

namespace {

constexpr int kIdY = kLbfgsSlots, kIdG = 2 * kLbfgsSlots;     // id of y slot 0, id of the gradient

// bit id set <=> basis vector id is live: s and y of every kept pair (+ the candidate's s when with_candidate)
__device__ __forceinline__ unsigned live_mask(const LbfgsDev* st, bool with_candidate)
{
    unsigned m = 0;
    const int count = st->count;
    for (int k = 0; k < count; ++k) { const int slot = st->order[k]; m |= (1u << slot) | (1u << (kIdY + slot)); }
    if (with_candidate) m |= 1u << st->free_slot;
    return m;
}

__device__ __forceinline__ const float* basis_ptr(const LbfgsVecs& v, int id) { return id < kIdY ? v.s[id] : v.y[id - kIdY]; }

__device__ __forceinline__ void dot4(float& acc, const float4& a, const float4& b)
{
    acc += a.x * b.x; acc += a.y * b.y; acc += a.z * b.z; acc += a.w * b.w;
}

}  // namespace

// The live vectors, compacted: position pos < n_live streams basis id ids[pos]; the positions beyond stream `filler` again (a vector
// the sweep has just read: cache hits), so that every load of a sweep is unconditional and all of them are in flight together --
// guarded loads (one uniform branch per id) serialise to one outstanding load per wave.
struct LiveList { const float* ptr[kIdG]; int ids[kIdG]; int n; };

__device__ __forceinline__ const float* uniform_ptr(const float* p)            // the same value in every lane: keep it in scalar registers
{
    const unsigned long long u = reinterpret_cast<unsigned long long>(p);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u), hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
    return reinterpret_cast<const float*>(((unsigned long long)hi << 32) | lo);
}

__device__ __forceinline__ void build_live_list(LiveList* ll, const LbfgsArgs& a, unsigned mask, const float* filler)
{
    if (threadIdx.x == 0) {
        int n = 0;
        for (int id = 0; id < kIdG; ++id)
            if ((mask >> id) & 1u) { ll->ptr[n] = basis_ptr(a.v, id); ll->ids[n] = id; ++n; }
        ll->n = n;
        for (int pos = n; pos < kIdG; ++pos) { ll->ptr[pos] = filler; ll->ids[pos] = -1; }
    }
    __syncthreads();
}

// ---- pass 1: y = g_new - g (mode 1), per-block partial sums of g'.b_id and y.b_id for every live id (g' = g_new, or a.g in mode 0)
__global__ __launch_bounds__(256) void lbfgs_gram_pass_k(const LbfgsArgs a, const float* __restrict__ g_new, const int mode)
{
    constexpr int kAcc = 2 * kIdG + 4;                              // g'.b and y.b by position, then g'.g', y.y, y.g' (+ pad)
    __shared__ float red[kAcc * 4];
    __shared__ LiveList ll;
    const LbfgsDev* st = a.st;
    const int fs = st->free_slot;
    float* ynew = a.v.y[fs];
    const float* gp = mode == 1 ? g_new : a.g;
    build_live_list(&ll, a, live_mask(st, mode == 1), gp);
    const float* ptr[kIdG];
#pragma unroll
    for (int pos = 0; pos < kIdG; ++pos) ptr[pos] = uniform_ptr(ll.ptr[pos]);
    float accg[kIdG], accy[kIdG];                                   // by position
#pragma unroll
    for (int pos = 0; pos < kIdG; ++pos) accg[pos] = accy[pos] = 0.f;
    float acc_gg = 0.f, acc_yy = 0.f, acc_yg = 0.f;
    const size_t n4 = a.n / 4;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const float4 gn = ld4(gp, i);
        float4 y = make_float4(0.f, 0.f, 0.f, 0.f);
        if (mode == 1) {
            const float4 go = ld4(a.g, i);
            y.x = gn.x - go.x; y.y = gn.y - go.y; y.z = gn.z - go.z; y.w = gn.w - go.w;
            st4(ynew, i, y);
            dot4(acc_yy, y, y);
            dot4(acc_yg, y, gn);
        }
        dot4(acc_gg, gn, gn);
        float4 v[kIdG];
#pragma unroll
        for (int pos = 0; pos < kIdG; ++pos) v[pos] = ld4(ptr[pos], i);
#pragma unroll
        for (int pos = 0; pos < kIdG; ++pos) {
            dot4(accg[pos], v[pos], gn);
            if (mode == 1) dot4(accy[pos], v[pos], y);
        }
    }
    if (blockIdx.x == 0 && threadIdx.x < (a.n & 3)) {
        const size_t i = n4 * 4 + threadIdx.x;
        const float gn = gp[i];
        float y = 0.f;
        if (mode == 1) { y = gn - a.g[i]; ynew[i] = y; acc_yy += y * y; acc_yg += y * gn; }
        acc_gg += gn * gn;
#pragma unroll
        for (int pos = 0; pos < kIdG; ++pos) {
            const float v = ptr[pos][i];
            accg[pos] += v * gn;
            if (mode == 1) accy[pos] += v * y;
        }
    }
    float acc[kAcc];
#pragma unroll
    for (int pos = 0; pos < kIdG; ++pos) { acc[pos] = accg[pos]; acc[kIdG + pos] = accy[pos]; }
    acc[2 * kIdG] = acc_gg; acc[2 * kIdG + 1] = acc_yy; acc[2 * kIdG + 2] = acc_yg; acc[2 * kIdG + 3] = 0.f;
    block_sum(acc, red);
    if (threadIdx.x == 0) {                                         // position -> row of its id; rows of dead ids stay stale (never read)
#pragma unroll
        for (int pos = 0; pos < kIdG; ++pos) {
            const int id = ll.ids[pos];
            if (id < 0) continue;
            a.gpart[(size_t)id * kMaxPartials + blockIdx.x] = acc[pos];
            a.gpart[(size_t)(kLbNB + id) * kMaxPartials + blockIdx.x] = acc[kIdG + pos];
        }
        a.gpart[(size_t)kIdG * kMaxPartials + blockIdx.x] = acc[2 * kIdG];                          // g'.g'
        a.gpart[(size_t)(kLbNB + kIdY + fs) * kMaxPartials + blockIdx.x] = acc[2 * kIdG + 1];          // y.y
        a.gpart[(size_t)(kLbNB + kIdG) * kMaxPartials + blockIdx.x] = acc[2 * kIdG + 2];               // y.g'
    }
}

// One workgroup's copy of the small state in LDS: a dependent chain of global-memory reads costs about a microsecond per link, and the
// bookkeeping below is a few hundred of them.
struct GramShared {
    double B[kLbNB][kLbNB + 1];
    double delta[kLbNB], v[kLbNB], sums[kLbGramRows];
    double sy[kLbfgsSlots], yy[kLbfgsSlots], alpha[kLbfgsSlots];
    double r0, last_sy;
    int order[kLbfgsSlots], count, free_slot;
};

__device__ __forceinline__ unsigned live_mask_sh(const GramShared& g)
{
    unsigned m = 0;
    for (int k = 0; k < g.count; ++k) m |= (1u << g.order[k]) | (1u << (kIdY + g.order[k]));
    return m;
}

__device__ __forceinline__ double wave_allsum_f64(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// the coefficient recursion (optimizers.py:89-108 on coefficients) by ONE WAVE on the LDS copy: lane l keeps delta[l], every
// "sdot" is one product per lane and a butterfly sum (the same value in every lane), every "saxpy" touches one lane
__device__ void lbfgs_gram_recursion(GramShared& g, size_t n, int lane)
{
    const int count = g.count;
    if (count == 0) {                                               // p = g / sqrt(g.g / N3)
        const double r0 = sqrt(g.B[kIdG][kIdG] / (double)n);
        if (lane == 0) g.r0 = r0;
        if (lane < kLbNB) g.delta[lane] = lane == kIdG ? 1.0 / r0 : 0.0;   // (only for the inner products of the next candidate s)
        return;
    }
    const unsigned mask = live_mask_sh(g) | (1u << kIdG);
    const bool live = lane < kLbNB && ((mask >> lane) & 1u);
    const int row = lane < kLbNB ? lane : 0;
    double d = lane == kIdG ? 1.0 : 0.0, my_alpha = 0.0;
    for (int j = count - 1; j >= 0; --j) {                          // first loop, newest pair first
        const int slot = g.order[j];
        const double alpha = wave_allsum_f64(live ? d * g.B[row][slot] : 0.0) / g.sy[slot];
        if (lane == slot) my_alpha = alpha;
        if (lane == kIdY + slot) d -= alpha;
    }
    const int newest = g.order[count - 1];
    d *= g.sy[newest] / g.yy[newest];
    for (int i = 0; i < count; ++i) {                               // second loop, oldest pair first
        const int slot = g.order[i];
        const double beta = wave_allsum_f64(live ? d * g.B[row][kIdY + slot] : 0.0) / g.sy[slot];
        if (lane == slot) d += my_alpha - beta;
    }
    if (lane < kLbNB) g.delta[lane] = d;
    if (lane < kLbfgsSlots) g.alpha[lane] = my_alpha;
    if (lane == 0) g.r0 = 0.0;
}

// ---- bookkeeping after pass 1 (one workgroup): B rows of the candidate pair and of the new gradient, gate / commit / eviction
//      (optimizers.py:79-87, as lbfgs_commit_k), then the coefficients of the next direction
__global__ __launch_bounds__(256) void lbfgs_gram_commit_k(const LbfgsArgs a, const int mode_arg)
{
    __shared__ GramShared g;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    LbfgsDev* st = a.st;
    LbfgsGram* gm = a.gm;
    static_assert(kLbGrid == 1024, "a lane takes four 16-byte pieces of a row of partial sums");
    // mode 3 (tile-sharded): this rank's sums only -> a.gsums, no bookkeeping; modes 4 / 5: modes 0 / 1 on sums the caller has
    // all-reduced into a.gsums
    const bool from_sums = mode_arg >= 4;
    const bool sums_only = mode_arg == 3;
    const int mode = from_sums ? mode_arg - 4 : sums_only ? 0 : mode_arg;
    if (from_sums) {
        for (int r = tid; r < kLbGramRows; r += 256) g.sums[r] = (double)a.gsums[r];
    } else if (mode != 2) {                     // partial sums: wave w takes rows w, w + 4, ...; every load first, then the (fixed-order) sums
        constexpr int kRowsPerWave = (kLbGramRows + 3) / 4;
        double racc[kRowsPerWave];
#pragma unroll
        for (int k = 0; k < kRowsPerWave; ++k) {
            const int r = wave + 4 * k;
            racc[k] = 0.0;
            if (r < kLbGramRows) {
                const float4* part = reinterpret_cast<const float4*>(a.gpart + (size_t)r * kMaxPartials);
                const float4 q0 = part[lane], q1 = part[64 + lane], q2 = part[128 + lane], q3 = part[192 + lane];
                racc[k] = ((((double)q0.x + (double)q0.y) + ((double)q0.z + (double)q0.w)) + (((double)q1.x + (double)q1.y) + ((double)q1.z + (double)q1.w))) +
                          ((((double)q2.x + (double)q2.y) + ((double)q2.z + (double)q2.w)) + (((double)q3.x + (double)q3.y) + ((double)q3.z + (double)q3.w)));
            }
        }
#pragma unroll
        for (int k = 0; k < kRowsPerWave; ++k) {
            const int r = wave + 4 * k;
            double acc = racc[k];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
            if (lane == 0 && r < kLbGramRows) g.sums[r] = acc;
        }
    }
    if (sums_only) {                            // (uniform) this rank's share of every inner product, rounded to fp32 as sdot returns it
        __syncthreads();
        for (int r = tid; r < kLbGramRows; r += 256) a.gsums[r] = (float)g.sums[r];
        return;
    }
    for (int i = tid; i < kLbNB * kLbNB; i += 256) g.B[i / kLbNB][i % kLbNB] = gm->B[i / kLbNB][i % kLbNB];
    if (tid < kLbNB) g.delta[tid] = gm->delta[tid];
    if (tid < kLbfgsSlots) { g.sy[tid] = st->sy[tid]; g.yy[tid] = st->yy[tid]; g.order[tid] = st->order[tid]; g.alpha[tid] = 0.0; }
    if (tid == 0) { g.count = st->count; g.free_slot = st->free_slot; g.r0 = gm->r0; g.last_sy = st->last_sy; }
    __syncthreads();
    const unsigned kept = live_mask_sh(g);
    const unsigned old_live = kept | (1u << kIdG);
    const int fs = g.free_slot, ids = fs, idy = kIdY + fs;
    const double* G = g.sums;                   // g'.b_id
    const double* Y = g.sums + kLbNB;           // y.b_id
    const double nstep = -(double)a.step;
    double ss = 0.0;
    if (mode == 1 && wave == 0) {               // v[id] = sum_l delta_l B[l][id]: the candidate s = -step * sum_l delta_l b_l against b_id
        double v = 0.0;                         // (delta and row g of B still belong to the OLD gradient here)
        const bool mine = lane < kLbNB && ((old_live >> lane) & 1u);
        if (mine)
            for (int l = 0; l < kLbNB; ++l) if ((old_live >> l) & 1u) v += g.delta[l] * g.B[l][lane];
        if (lane < kLbNB) g.v[lane] = v;
        ss = wave_allsum_f64(mine ? g.delta[lane] * v : 0.0);
    }
    __syncthreads();
    if (tid < kIdG && ((kept >> tid) & 1u)) {   // the rows against the kept vectors, one lane per id
        if (mode == 1) {
            g.B[ids][tid] = g.B[tid][ids] = nstep * g.v[tid];
            g.B[idy][tid] = g.B[tid][idy] = Y[tid];
        }
        if (mode != 2) g.B[kIdG][tid] = g.B[tid][kIdG] = G[tid];
    }
    if (tid == 0) {
        if (mode == 1) {
            g.B[ids][ids] = nstep * nstep * ss;
            const double sy = (double)(float)Y[ids], yy = (double)(float)Y[idy];       // sdot returns an fp32 value
            g.B[idy][ids] = g.B[ids][idy] = sy;
            g.B[idy][idy] = yy;
            g.B[kIdG][ids] = g.B[ids][kIdG] = G[ids];
            g.B[kIdG][idy] = g.B[idy][kIdG] = Y[kIdG];
            int count = g.count;
            if (sy > 1e-10) {                   // store_curvature_pair, optimizers.py:79-87
                g.sy[fs] = sy;
                g.yy[fs] = yy;
                g.order[count++] = fs;
                if (count > kLbfgsCorr) {
                    for (int k = 0; k + 1 < count; ++k) g.order[k] = g.order[k + 1];
                    --count;
                }
                g.count = count;
                unsigned used = 0;
                for (int k = 0; k < count; ++k) used |= 1u << g.order[k];
                g.free_slot = __builtin_ctz(~used);                 // the slot no live pair uses is the next scratch slot
            }
            g.last_sy = sy;
        }
        if (mode != 2) g.B[kIdG][kIdG] = G[kIdG];
    }                                           // mode 2: B was loaded whole (lbfgs_gram_load_k)
    __syncthreads();
    if (wave == 0) lbfgs_gram_recursion(g, a.n_global ? a.n_global : a.n, lane);
    __syncthreads();
    for (int i = tid; i < kLbNB * kLbNB; i += 256) gm->B[i / kLbNB][i % kLbNB] = g.B[i / kLbNB][i % kLbNB];
    if (tid < kLbNB) gm->delta[tid] = g.delta[tid];
    if (tid < kLbfgsSlots) { st->sy[tid] = g.sy[tid]; st->yy[tid] = g.yy[tid]; st->order[tid] = g.order[tid]; st->alpha[tid] = g.alpha[tid]; }
    if (tid == 0) { st->count = g.count; st->free_slot = g.free_slot; st->last_sy = g.last_sy; gm->r0 = g.r0; }
}

// ---- pass 2: p = sum delta_id b_id (the gradient first, then ascending ids); a.apply: s = -step p into the free slot, x += s
__global__ __launch_bounds__(256) void lbfgs_gram_apply_k(const LbfgsArgs a)
{
    __shared__ LiveList ll;
    __shared__ float coef_s[kIdG];
    const LbfgsDev* st = a.st;
    const LbfgsGram* gm = a.gm;
    const int count = st->count;
    build_live_list(&ll, a, live_mask(st, false), a.g);
    if (threadIdx.x < kIdG) coef_s[threadIdx.x] = ll.ids[threadIdx.x] >= 0 ? (float)gm->delta[ll.ids[threadIdx.x]] : 0.f;
    __syncthreads();
    const float* ptr[kIdG];
    float coef[kIdG];
#pragma unroll
    for (int pos = 0; pos < kIdG; ++pos) { ptr[pos] = uniform_ptr(ll.ptr[pos]); coef[pos] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, coef_s[pos]))); }
    const float coef_g = (float)gm->delta[kIdG];
    float* s_new = a.v.s[st->free_slot];
    const float nstep = -a.step;
    const double r0 = gm->r0;
    const size_t n4 = a.n / 4;
    for (size_t k = (size_t)blockIdx.x * 256 + threadIdx.x; k < n4; k += (size_t)gridDim.x * 256) {
        const float4 g = ld4(a.g, k);
        float4 t;
        if (count == 0) {
            t.x = (float)((double)g.x / r0); t.y = (float)((double)g.y / r0); t.z = (float)((double)g.z / r0); t.w = (float)((double)g.w / r0);
        } else {
            float4 v[kIdG];
#pragma unroll
            for (int pos = 0; pos < kIdG; ++pos) v[pos] = ld4(ptr[pos], k);
            t.x = coef_g * g.x; t.y = coef_g * g.y; t.z = coef_g * g.z; t.w = coef_g * g.w;
#pragma unroll
            for (int pos = 0; pos < kIdG; ++pos) {
                t.x = coef[pos] * v[pos].x + t.x; t.y = coef[pos] * v[pos].y + t.y;
                t.z = coef[pos] * v[pos].z + t.z; t.w = coef[pos] * v[pos].w + t.w;
            }
        }
        if (a.apply) {
            float4 sn, xv = ld4(a.x, k);
            sn.x = nstep * t.x; sn.y = nstep * t.y; sn.z = nstep * t.z; sn.w = nstep * t.w;
            xv.x += sn.x; xv.y += sn.y; xv.z += sn.z; xv.w += sn.w;
            st4(s_new, k, sn);
            st4(a.x, k, xv);
        } else st4(a.p, k, t);
    }
    if (blockIdx.x == 0 && threadIdx.x < (a.n & 3)) {
        const size_t k = n4 * 4 + threadIdx.x;
        float t;
        if (count == 0) t = (float)((double)a.g[k] / r0);
        else {
            t = coef_g * a.g[k];
#pragma unroll
            for (int pos = 0; pos < kIdG; ++pos) t = coef[pos] * ptr[pos][k] + t;
        }
        if (a.apply) { const float sn = nstep * t; s_new[k] = sn; a.x[k] += sn; }
        else a.p[k] = t;
    }
}

// test hook: all pairwise inner products were taken one by one (launch_vec_dot); B = that table
__global__ void lbfgs_gram_load_k(const LbfgsArgs a, const float* __restrict__ dots)
{
    for (int i = threadIdx.x; i < kLbNB * kLbNB; i += blockDim.x) a.gm->B[i / kLbNB][i % kLbNB] = (double)dots[i];
}

hipError_t launch_lbfgs_gram_pass(const LbfgsArgs& a, const float* g_new, int mode, hipStream_t s)
{
    lbfgs_gram_pass_k<<<kLbGrid, 256, 0, s>>>(a, g_new, mode);
    lbfgs_gram_commit_k<<<1, 256, 0, s>>>(a, mode);
    return hipGetLastError();
}

hipError_t launch_lbfgs_gram_pass_local(const LbfgsArgs& a, const float* g_new, int mode, hipStream_t s)
{
    lbfgs_gram_pass_k<<<kLbGrid, 256, 0, s>>>(a, g_new, mode);
    lbfgs_gram_commit_k<<<1, 256, 0, s>>>(a, 3);
    return hipGetLastError();
}

hipError_t launch_lbfgs_gram_commit_global(const LbfgsArgs& a, int mode, hipStream_t s)
{
    lbfgs_gram_commit_k<<<1, 256, 0, s>>>(a, 4 + mode);
    return hipGetLastError();
}

int lbfgs_gram_rows() { return kLbGramRows; }

hipError_t launch_lbfgs_gram_apply(const LbfgsArgs& a, hipStream_t s)
{
    lbfgs_gram_apply_k<<<kLbGrid, 256, 0, s>>>(a);
    return hipGetLastError();
}

hipError_t launch_lbfgs_gram_load(const LbfgsArgs& a, const float* dots, hipStream_t s)
{
    lbfgs_gram_load_k<<<1, 256, 0, s>>>(a, dots);
    // (no partial sums to take: fill the rows the commit kernel sums with zeros is not needed in mode 2 -- it ignores them)
    lbfgs_gram_commit_k<<<1, 256, 0, s>>>(a, 2);
    return hipGetLastError();
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
