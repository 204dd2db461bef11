// Style statistics on the matrix cores (v_mfma_f32_32x32x2_f32, exact fp32).
//
//   gram_partial : G_s = F[:, slab_s] F[:, slab_s]^T        worker.py:109-114 (np.dot(x, x.T))
//   gram_reduce  : D = sum_s G_s / n - G_style, sum D^2      worker.py:114, 261, 267
//   style_grad   : S = c2 * (D @ F) [+ fused saxpy]          worker.py:262-269
//
// F is a blob, [C][hw] fp32 row-major (NCHW with N = 1).  The Gram GEMM contracts over hw
// (up to 2^20), so it is split along K into slabs -> [splits][C][C] fp32 partials that a second
// kernel sums in a fixed order (bitwise reproducible, no float atomics).
#include "st2_kernels.h"
#include "reduce.cuh"

namespace st2 {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// ---------------------------------------------------------------------------------- Gram partials
constexpr int GKT = 32;          // K elements staged per step
constexpr int GLD = GKT + 1;     // padded leading dimension: lane -> row reads hit 32 distinct banks

GramPlan gram_plan(int C, int hw)
{
    GramPlan p;
    p.bt = C > 64 ? 128 : 64;
    const int t = (C + p.bt - 1) / p.bt;
    p.tiles = t * t;
    int want = (1024 + p.tiles - 1) / p.tiles;             // ~4 workgroups per CU
    const int max_splits = (hw + 4 * GKT - 1) / (4 * GKT); // at least 128 K per slab
    if (want > max_splits) want = max_splits;
    if (want < 1) want = 1;
    int kslab = (hw + want - 1) / want;
    kslab = (kslab + GKT - 1) / GKT * GKT;
    p.kslab = kslab;
    p.splits = (hw + kslab - 1) / kslab;
    p.slab_floats = (size_t)p.splits * C * C;
    return p;
}

template <int BT>
__global__ __launch_bounds__(256) void gram_partial_k(const float* __restrict__ F, float* __restrict__ slabs,
                                                      int C, int hw, int tiles_1d, int kslab)
{
    constexpr int T = BT / 64;                       // 32x32 MFMA tiles per wave per dimension
    constexpr int ROWS_PER_T = BT / 8;               // rows each thread stages (256 threads = 8 rows x 32 k)
    __shared__ float As[BT * GLD];
    __shared__ float Bs[BT * GLD];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int tile = blockIdx.x % (tiles_1d * tiles_1d);
    const int split = blockIdx.x / (tiles_1d * tiles_1d);
    const int ti = tile / tiles_1d, tj = tile % tiles_1d;
    const int i0 = ti * BT, j0 = tj * BT;
    const bool diag = ti == tj;
    const int kbeg = split * kslab;
    const int kend = min(hw, kbeg + kslab);

    const int sk = tid & 31, sr = tid >> 5;          // staging coordinates
    float ra[ROWS_PER_T], rb[ROWS_PER_T];

    auto load = [&](int k0) {
        const int k = k0 + sk;
#pragma unroll
        for (int it = 0; it < ROWS_PER_T; ++it) {
            const int r = sr + it * 8;
            ra[it] = (i0 + r < C && k < kend) ? F[(size_t)(i0 + r) * hw + k] : 0.f;
            if (!diag) rb[it] = (j0 + r < C && k < kend) ? F[(size_t)(j0 + r) * hw + k] : 0.f;
        }
    };
    auto store = [&]() {
#pragma unroll
        for (int it = 0; it < ROWS_PER_T; ++it) {
            const int r = sr + it * 8;
            As[r * GLD + sk] = ra[it];
            if (!diag) Bs[r * GLD + sk] = rb[it];
        }
    };

    f32x16 acc[T][T];
#pragma unroll
    for (int i = 0; i < T; ++i)
#pragma unroll
        for (int j = 0; j < T; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const float* bsrc = diag ? As : Bs;
    const int khalf = lane >> 5, l31 = lane & 31;
    const float* a_base = As + (wm * (T * 32) + l31) * GLD + khalf;
    const float* b_base = bsrc + (wn * (T * 32) + l31) * GLD + khalf;

    load(kbeg);
    for (int k0 = kbeg; k0 < kend; k0 += GKT) {
        __syncthreads();
        store();
        __syncthreads();
        if (k0 + GKT < kend) load(k0 + GKT);
#pragma unroll
        for (int ks = 0; ks < GKT / 2; ++ks) {
            float av[T], bv[T];
#pragma unroll
            for (int i = 0; i < T; ++i) av[i] = a_base[i * 32 * GLD + 2 * ks];
#pragma unroll
            for (int j = 0; j < T; ++j) bv[j] = b_base[j * 32 * GLD + 2 * ks];
#pragma unroll
            for (int i = 0; i < T; ++i)
#pragma unroll
                for (int j = 0; j < T; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], acc[i][j], 0, 0, 0);
        }
    }

    float* dst = slabs + (size_t)split * C * C;
#pragma unroll
    for (int i = 0; i < T; ++i)
#pragma unroll
        for (int j = 0; j < T; ++j) {
            const int col = j0 + wn * (T * 32) + j * 32 + l31;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = i0 + wm * (T * 32) + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * khalf;
                if (row < C && col < C) dst[(size_t)row * C + col] = acc[i][j][e];
            }
        }
}

hipError_t launch_gram_partial(const float* F, float* slabs, int C, int hw, const GramPlan& pl, hipStream_t s)
{
    const int t1 = (C + pl.bt - 1) / pl.bt;
    const unsigned grid = (unsigned)(pl.tiles * pl.splits);
    if (pl.bt == 128) gram_partial_k<128><<<grid, 256, 0, s>>>(F, slabs, C, hw, t1, pl.kslab);
    else gram_partial_k<64><<<grid, 256, 0, s>>>(F, slabs, C, hw, t1, pl.kslab);
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void gram_reduce_k(const float* __restrict__ slabs, const float* __restrict__ target,
                                                     float* __restrict__ out, float* __restrict__ partial,
                                                     int cc, int splits, float n)
{
    __shared__ float scratch[4];
    float acc[1] = {0.f};
    for (int i = blockIdx.x * 256 + threadIdx.x; i < cc; i += gridDim.x * 256) {
        float sum = 0.f;
        for (int s = 0; s < splits; ++s) sum += slabs[(size_t)s * cc + i];
        float v = sum / n;                           // np.dot(x, x.T) / np.float32(x.size)
        if (target) v -= target[i];                  // gram_matrix(F) - grams[layer]
        out[i] = v;
        acc[0] += v * v;
    }
    block_sum(acc, scratch);
    if (threadIdx.x == 0 && partial) partial[blockIdx.x] = acc[0];
}

hipError_t launch_gram_reduce(const float* slabs, const float* target, float* out, float* partial,
                              int* n_partial, int C, int hw, const GramPlan& pl, hipStream_t s)
{
    const int cc = C * C;
    const int grid = reduce_grid((size_t)cc, 256, kMaxPartials);
    if (n_partial) *n_partial = grid;
    gram_reduce_k<<<grid, 256, 0, s>>>(slabs, target, out, partial, cc, pl.splits, (float)((double)C * hw));
    return hipGetLastError();
}

// ------------------------------------------------------------------------------- style gradient
// S[m][p] = c2 * sum_k D[m][k] F[k][p].  D is symmetric (difference of two Gram matrices), so the A
// operand is staged from rows of D with m contiguous: d_s[k][m] = D[k][m].
constexpr int SKC = 16;          // K (channels) per staged chunk
constexpr int SBN = 256;         // pixels per workgroup

template <int BM, int WAVES_M, int WAVES_N>
__global__ __launch_bounds__(256) void style_grad_k(const float* __restrict__ D, const float* __restrict__ F,
                                                    float* __restrict__ dst, float c2, int mode, float sw,
                                                    const float* __restrict__ norm, int accumulate,
                                                    float* __restrict__ partial, int C, int hw, int n_mtiles)
{
    constexpr int TM = BM / WAVES_M / 32;
    constexpr int TN = SBN / WAVES_N / 32;
    constexpr int ND = SKC * BM, NF = SKC * SBN;
    constexpr int D_PER_T = ND / 256, F_PER_T = NF / 256;
    __shared__ float d_s[ND];
    __shared__ float f_s[NF];
    __shared__ float scratch[4];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wave_m = wave / WAVES_N, wave_n = wave % WAVES_N;
    const int mt = blockIdx.x % n_mtiles;
    const int ptile = blockIdx.x / n_mtiles;
    const int m0 = mt * BM;
    const size_t p0 = (size_t)ptile * SBN;

    float dreg[D_PER_T], freg[F_PER_T];
    auto load = [&](int k0) {
#pragma unroll
        for (int i = 0; i < D_PER_T; ++i) {
            const int e = tid + i * 256;
            const int k = e / BM, m = e % BM;
            dreg[i] = (k0 + k < C && m0 + m < C) ? D[(size_t)(k0 + k) * C + m0 + m] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < F_PER_T; ++i) {
            const int e = tid + i * 256;
            const int k = e / SBN, p = e % SBN;
            freg[i] = (k0 + k < C && p0 + p < (size_t)hw) ? F[(size_t)(k0 + k) * hw + p0 + p] : 0.f;
        }
    };
    auto store = [&]() {
#pragma unroll
        for (int i = 0; i < D_PER_T; ++i) d_s[tid + i * 256] = dreg[i];
#pragma unroll
        for (int i = 0; i < F_PER_T; ++i) f_s[tid + i * 256] = freg[i];
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int khalf = lane >> 5, l31 = lane & 31;
    const float* a_base = d_s + khalf * BM + wave_m * (TM * 32) + l31;
    const float* b_base = f_s + khalf * SBN + wave_n * (TN * 32) + l31;

    load(0);
    for (int k0 = 0; k0 < C; k0 += SKC) {
        __syncthreads();
        store();
        __syncthreads();
        if (k0 + SKC < C) load(k0 + SKC);
#pragma unroll
        for (int kk = 0; kk < SKC / 2; ++kk) {
            float av[TM], bv[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) av[i] = a_base[2 * kk * BM + i * 32];
#pragma unroll
            for (int j = 0; j < TN; ++j) bv[j] = b_base[2 * kk * SBN + j * 32];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], acc[i][j], 0, 0, 0);
        }
    }

    const float coef = mode ? sw / *norm : 0.f;        // sw / sn[layer]
    float ss[1] = {0.f};
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const size_t p = p0 + wave_n * (TN * 32) + j * 32 + l31;
        if (p >= (size_t)hw) continue;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = m0 + wave_m * (TM * 32) + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * khalf;
                if (m >= C) continue;
                const size_t idx = (size_t)m * hw + p;
                const float v = acc[i][j][e] * c2;     // s_grad *= 2 / (gram_diff.size * feat.size)
                ss[0] += v * v;
                if (mode == 0) dst[idx] = v;
                else dst[idx] = coef * v + (accumulate ? dst[idx] : 0.f);
            }
    }
    block_sum(ss, scratch);
    if (tid == 0) partial[blockIdx.x] = ss[0];
}

hipError_t launch_style_grad(const float* D, const float* F, float* dst, float c2, int mode,
                             float sw, const float* norm, int accumulate, float* partial,
                             int* n_partial, int C, int hw, hipStream_t s)
{
    const int ptiles = (hw + SBN - 1) / SBN;
    const bool big = C > 64;
    const int bm = big ? 128 : 64;
    const int n_mtiles = (C + bm - 1) / bm;
    const long long grid = (long long)ptiles * n_mtiles;
    if (grid > 0x7fffffffLL) return hipErrorInvalidValue;
    *n_partial = (int)grid;   // NOTE: may exceed kMaxPartials; the engine sizes this slot separately
    if (big) style_grad_k<128, 2, 2><<<(unsigned)grid, 256, 0, s>>>(D, F, dst, c2, mode, sw, norm, accumulate, partial, C, hw, n_mtiles);
    else style_grad_k<64, 1, 4><<<(unsigned)grid, 256, 0, s>>>(D, F, dst, c2, mode, sw, norm, accumulate, partial, C, hw, n_mtiles);
    return hipGetLastError();
}

}  // namespace st2
