// Style statistics on the matrix cores (v_mfma_f32_32x32x2_f32, exact fp32).
//
//   gram_partial : G_s = F[:, slab_s] F[:, slab_s]^T        worker.py:109-114 (np.dot(x, x.T))
//   gram_reduce  : D = sum_s G_s / n - G_style, sum D^2      worker.py:114, 261, 267
//   (the style gradient S = c2 * (D @ F) runs on the conv pipeline: conv3x3_mfma.hip, TAPS = 1)
//
// F is a blob, [C][hw] fp32 row-major (NCHW with N = 1).  The Gram GEMM contracts over hw
// (up to 2^20), so it is split along K into slabs -> [splits][C][C] fp32 partials that a second
// kernel sums in a fixed order (bitwise reproducible, no float atomics).
#include "st2_kernels.h"
#include "wave_reduce.h"
#include <stdint.h>
#include <stdlib.h>

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
    p.tiles = t * (t + 1) / 2;                             // G is symmetric: upper-triangular tiles only, mirrored on store
    // 128-row tiles (C > 64) are matrix-core bound and two workgroups fill a CU (64 KiB of LDS each): the launch must be ONE round of
    // at most 512 workgroups -- rounding the split count UP put 513 (C = 256) and 520 (C = 512) of them on 512 slots, and the one
    // left over ran a second round alone on an idle chip (conv3_1: 72 us for 41 us of matrix work).  64-row tiles (C <= 64) are
    // HBM-bound (conv1_1: 268 MB for 8.6 GFLOP) and small (32 KiB of LDS): more workgroups per CU keep more bytes in flight.
    // ST2_GRAM_BLOCKS / ST2_GRAM_BLOCKS64 override the two targets (read per call: A/B runs in one process).
    const char* e128 = getenv("ST2_GRAM_BLOCKS");
    const char* e64 = getenv("ST2_GRAM_BLOCKS64");
    const int target128 = e128 && *e128 ? atoi(e128) : 512, target64 = e64 && *e64 ? atoi(e64) : 1024;
    int want = p.bt == 128 ? target128 / p.tiles : (target64 + p.tiles - 1) / p.tiles;
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

// the same plan with slabs of whole 64-pixel steps (gram16.hip stages 64 pixels per step)
GramPlan gram_plan16(int C, int hw)
{
    GramPlan p = gram_plan(C, hw);
    p.kslab = (p.kslab + 63) / 64 * 64;
    p.splits = (hw + p.kslab - 1) / p.kslab;
    p.slab_floats = (size_t)p.splits * C * C;
    return p;
}

template <int BT>
__global__ __launch_bounds__(256) void gram_partial_k(const float* __restrict__ F, float* __restrict__ slabs,
                                                      int C, int hw, int tiles_1d, int kslab, GramRoi roi)
{
    constexpr int T = BT / 64;                       // 32x32 MFMA tiles per wave per dimension
    constexpr int ROWS_PER_T = BT / 8;               // rows each thread stages (256 threads = 8 rows x 32 k)
    __shared__ float As[BT * GLD];
    __shared__ float Bs[BT * GLD];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int n_ut = tiles_1d * (tiles_1d + 1) / 2;
    int tile = blockIdx.x % n_ut;
    const int split = blockIdx.x / n_ut;
    int ti = 0;
    while (tile >= tiles_1d - ti) { tile -= tiles_1d - ti; ++ti; }      // row ti of the upper triangle holds tiles_1d - ti tiles
    const int tj = ti + tile;
    const int i0 = ti * BT, j0 = tj * BT;
    const bool diag = ti == tj;
    const int kbeg = split * kslab;
    const int kend = min(hw, kbeg + kslab);

    const int sk = tid & 31, sr = tid >> 5;          // staging coordinates
    float ra[ROWS_PER_T], rb[ROWS_PER_T];

    auto load = [&](int k0) {
        const int k = k0 + sk;
        // k runs over the pixels of the region of interest (the whole blob unless tile-sharded)
        const int ky = k / roi.rw, kx = k - ky * roi.rw;
        const size_t pix = (size_t)(roi.y0 + ky) * roi.pitch + roi.x0 + kx;
#pragma unroll
        for (int it = 0; it < ROWS_PER_T; ++it) {
            const int r = sr + it * 8;
            ra[it] = (i0 + r < C && k < kend) ? F[(size_t)(i0 + r) * roi.plane + pix] : 0.f;
            if (!diag) rb[it] = (j0 + r < C && k < kend) ? F[(size_t)(j0 + r) * roi.plane + pix] : 0.f;
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
                if (row < C && col < C) dst[(size_t)row * C + col] = acc[i][j][e];     // upper-triangular tiles only: the reduction mirrors
            }
        }
}

// ------------------------------------------------------------------ Gram partials, LDS-DMA pipeline
// Same contract as gram_partial_k for a whole blob with hw % 32 == 0 (every VGG blob of a power-of-two image): the
// operand rows are staged by LDS-DMA (no VGPR round trip), double-buffered, 32 pixels of K per step.
// An LDS-DMA piece lands as 64 consecutive quads, so padding is impossible; instead the QUAD SLOT of a row is
// swizzled, slot = quad ^ ((row >> 1) & 7), by choosing which global quad each lane fetches.  The MFMA operands are
// then read as one ds_read_b128 per row and four K (two k-pairs; the lane half selects .x/.z or .y/.w): the 16 lanes
// of a b128 phase (16 consecutive rows) hit 16 distinct bank quads -- conflict-free, and 4x fewer LDS instructions
// than the dword reads of the register-staged kernel.
typedef __attribute__((address_space(3))) void* gram_lptr_t;
typedef float gf32x4 __attribute__((ext_vector_type(4)));

template <int BT>
__device__ __forceinline__ void gram_partial_dma_body(const float* __restrict__ F, unsigned f_bytes, float* __restrict__ slabs,
                                                      int C, int hw, int tiles_1d, int kslab)
{
    constexpr int T = BT / 64;                       // 32x32 MFMA tiles per wave per dimension
    constexpr int IMG = BT * 32;                     // floats per operand image (BT rows x 32 K)
    constexpr int PIECES = IMG / 256;                // wave-DMAs (64 quads) per operand image
    constexpr int PPW = PIECES / 4;                  // per wave
    __shared__ __attribute__((aligned(16))) float smem[2][2][IMG];       // [stage][A/B]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int n_ut = tiles_1d * (tiles_1d + 1) / 2;
    int tile = blockIdx.x % n_ut;
    const int split = blockIdx.x / n_ut;
    int ti = 0;
    while (tile >= tiles_1d - ti) { tile -= tiles_1d - ti; ++ti; }
    const int tj = ti + tile;
    const int i0 = ti * BT, j0 = tj * BT;
    const bool diag = ti == tj;
    const int kbeg = split * kslab;
    const int kend = min(hw, kbeg + kslab);
    const int nsteps = (kend - kbeg) / 32;

    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)F, 0, f_bytes, 0x00020000);
    unsigned aoff[PPW], boff[PPW];
#pragma unroll
    for (int t = 0; t < PPW; ++t) {
        const int slot = (wave + 4 * t) * 64 + lane;          // LDS quad slot: row = slot / 8, swizzled quad = slot % 8
        const int r = slot >> 3, q = (slot & 7) ^ ((r >> 1) & 7);
        aoff[t] = i0 + r < C ? ((unsigned)(i0 + r) * (unsigned)hw + (unsigned)kbeg + 4u * q) * 4u : 0xffffffffu;
        boff[t] = j0 + r < C ? ((unsigned)(j0 + r) * (unsigned)hw + (unsigned)kbeg + 4u * q) * 4u : 0xffffffffu;
    }
    auto dma = [&](int step, int stage) {
        const unsigned so = (unsigned)step * 128u;             // 32 floats further along every row
#pragma unroll
        for (int t = 0; t < PPW; ++t) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (gram_lptr_t)(smem[stage][0] + (wave + 4 * t) * 256), 16,
                                                     aoff[t] == 0xffffffffu ? aoff[t] : aoff[t] + so, 0, 0, 0);
            if (!diag)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (gram_lptr_t)(smem[stage][1] + (wave + 4 * t) * 256), 16,
                                                         boff[t] == 0xffffffffu ? boff[t] : boff[t] + so, 0, 0, 0);
        }
    };

    f32x16 acc[T][T];
#pragma unroll
    for (int i = 0; i < T; ++i)
#pragma unroll
        for (int j = 0; j < T; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int l31 = lane & 31;
    const bool hi = lane >= 32;
    int arow[T], brow[T];                            // float offset of this lane's row in an operand image, and its swizzle key
#pragma unroll
    for (int i = 0; i < T; ++i) { arow[i] = wm * (T * 32) + i * 32 + l31; brow[i] = wn * (T * 32) + i * 32 + l31; }

    if (nsteps > 0) dma(0, 0);
    for (int st = 0; st < nsteps; ++st) {
        const int cur = st & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                              // step st has landed for every wave; stage cur^1 is free again
        if (st + 1 < nsteps) dma(st + 1, cur ^ 1);
        const float* As = smem[cur][0];
        const float* Bs = diag ? smem[cur][0] : smem[cur][1];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            float a0[T], a1[T], b0[T], b1[T];
#pragma unroll
            for (int i = 0; i < T; ++i) {
                const gf32x4 v = *reinterpret_cast<const gf32x4*>(As + (arow[i] * 8 + (q ^ ((arow[i] >> 1) & 7))) * 4);
                a0[i] = hi ? v.y : v.x; a1[i] = hi ? v.w : v.z;
            }
#pragma unroll
            for (int j = 0; j < T; ++j) {
                const gf32x4 v = *reinterpret_cast<const gf32x4*>(Bs + (brow[j] * 8 + (q ^ ((brow[j] >> 1) & 7))) * 4);
                b0[j] = hi ? v.y : v.x; b1[j] = hi ? v.w : v.z;
            }
#pragma unroll
            for (int i = 0; i < T; ++i)
#pragma unroll
                for (int j = 0; j < T; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[i], b0[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[i], b1[j], acc[i][j], 0, 0, 0);
                }
        }
    }

    const int khalf = lane >> 5;
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

// non-template entry points (a template kernel with a waves-per-SIMD launch bound loses its host stub)
__global__ __launch_bounds__(256, 2) void gram_partial_dma_128(const float* F, unsigned f_bytes, float* slabs, int C, int hw, int tiles_1d, int kslab)
{ gram_partial_dma_body<128>(F, f_bytes, slabs, C, hw, tiles_1d, kslab); }
__global__ __launch_bounds__(256, 2) void gram_partial_dma_64(const float* F, unsigned f_bytes, float* slabs, int C, int hw, int tiles_1d, int kslab)
{ gram_partial_dma_body<64>(F, f_bytes, slabs, C, hw, tiles_1d, kslab); }

hipError_t launch_gram_partial(const float* F, float* slabs, int C, int hw, const GramPlan& pl, hipStream_t s,
                               const GramRoi* roi_in)
{
    const int t1 = (C + pl.bt - 1) / pl.bt;
    const unsigned grid = (unsigned)(pl.tiles * pl.splits);
    GramRoi roi = roi_in ? *roi_in : GramRoi{0, 0, hw, hw, (size_t)hw};   // default: one "row" of hw pixels
    // whole blob, 32-pixel steps, 16-byte aligned rows, 32-bit buffer offsets: the LDS-DMA pipeline
    static const bool no_dma = [] { const char* e = getenv("ST2_GRAM_DMA"); return e && *e == '0'; }();
    if (!roi_in && !no_dma && hw % 32 == 0 && pl.kslab % 32 == 0 && (reinterpret_cast<uintptr_t>(F) & 15) == 0 &&
        4ull * C * hw < 0xfffffff0ull) {
        const unsigned fb = (unsigned)(4ull * C * hw);
        if (pl.bt == 128) gram_partial_dma_128<<<grid, 256, 0, s>>>(F, fb, slabs, C, hw, t1, pl.kslab);
        else gram_partial_dma_64<<<grid, 256, 0, s>>>(F, fb, slabs, C, hw, t1, pl.kslab);
        return hipGetLastError();
    }
    if (pl.bt == 128) gram_partial_k<128><<<grid, 256, 0, s>>>(F, slabs, C, hw, t1, pl.kslab, roi);
    else gram_partial_k<64><<<grid, 256, 0, s>>>(F, slabs, C, hw, t1, pl.kslab, roi);
    return hipGetLastError();
}

// `out` is written with leading dimension out_ld (>= C): the style-gradient GEMM wants D as [C][MPad].
// The slabs hold the upper-triangular bt x bt tiles only (G is symmetric); an element of such a tile is written to
// (r, c) and, for an off-diagonal tile, to (c, r) as well -- same value, so the result is exactly symmetric.
__global__ __launch_bounds__(256) void gram_reduce_k(const float* __restrict__ slabs, const float* __restrict__ target,
                                                     float* __restrict__ out, float* __restrict__ partial,
                                                     int cc, int splits, float n, int C, int out_ld, int bt)
{
    __shared__ float scratch[4];
    float acc[1] = {0.f};
    for (int i = blockIdx.x * 256 + threadIdx.x; i < cc; i += gridDim.x * 256) {
        const int r = i / C, c = i - r * C;
        const int tr = r / bt, tc = c / bt;
        if (tr > tc) continue;                       // produced by the mirror of (c, r)
        float sum = 0.f;
        for (int s = 0; s < splits; ++s) sum += slabs[(size_t)s * cc + i];
        float v = sum / n;                           // np.dot(x, x.T) / np.float32(x.size)
        if (target) v -= target[i];                  // gram_matrix(F) - grams[layer]  (the target is symmetric too)
        out[r * out_ld + c] = v;
        acc[0] += v * v;
        if (tr < tc) { out[c * out_ld + r] = v; acc[0] += v * v; }
    }
    block_sum(acc, scratch);
    if (threadIdx.x == 0 && partial) partial[blockIdx.x] = acc[0];
}

// First stage for many splits: fold `splits` slabs into `groups` slabs (each block: 256 elements x
// one group of consecutive splits, summed in order) so the final pass is short and still deterministic.
__global__ __launch_bounds__(256) void gram_fold_k(const float* __restrict__ slabs, float* __restrict__ folded,
                                                   int cc, int splits, int per_group, int C, int bt)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int g = blockIdx.y;
    if (i >= cc) return;
    if ((i / C) / bt > (i % C) / bt) return;         // lower-triangular tiles are never written nor read
    const int s0 = g * per_group, s1 = min(splits, s0 + per_group);
    float sum = 0.f;
    for (int s = s0; s < s1; ++s) sum += slabs[(size_t)s * cc + i];
    folded[(size_t)g * cc + i] = sum;
}

// Both stages in one launch for many splits: 16 lanes share an element, lane j sums splits j, j + 16, ... (16 loads in flight
// per lane pair of rounds), the 16 partial sums meet in LDS and are added in lane order -- a fixed order again, and a tenth of
// the launches' latency (two ~10 us kernels per style layer become one).
template <int GR_LANES>
__global__ __launch_bounds__(256) void gram_reduce_wide_k(const float* __restrict__ slabs, const float* __restrict__ target,
                                                          float* __restrict__ out, float* __restrict__ partial,
                                                          int cc, int splits, float n, int C, int out_ld, int bt)
{
    // one thread = four consecutive elements of a row (C % 4 == 0: same row, same tile), a row of threads = contiguous bytes of
    // one slab per load instruction; GR_LANES such rows of threads take the splits round-robin (8 rows of 32 threads for the
    // large matrices; 32 rows of 8 for C <= 128, whose few elements would otherwise leave most of the chip idle)
    constexpr int GR_QUADS = 256 / GR_LANES, GR_ELEMS = 4 * GR_QUADS;
    __shared__ float4 part_s[GR_LANES][GR_QUADS];
    __shared__ float scratch[4];
    const int q = threadIdx.x % GR_QUADS, ln = threadIdx.x / GR_QUADS;
    float acc[1] = {0.f};
    for (int base = blockIdx.x * GR_ELEMS; base < cc; base += gridDim.x * GR_ELEMS) {
        const int i = base + 4 * q;
        const int r = i < cc ? i / C : 0, c = i < cc ? i - r * C : 0;
        const bool live = i < cc && r / bt <= c / bt;        // lower-triangular tiles are produced by the mirror of (c, r)
        float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
        if (live)
            for (int s = ln; s < splits; s += GR_LANES) {
                const float4 v = *reinterpret_cast<const float4*>(slabs + (size_t)s * cc + i);
                sum.x += v.x; sum.y += v.y; sum.z += v.z; sum.w += v.w;
            }
        __syncthreads();                                     // the previous group's partial sums are consumed
        part_s[ln][q] = sum;
        __syncthreads();
        if (ln == 0 && live) {
            float4 tot = part_s[0][q];
#pragma unroll
            for (int j = 1; j < GR_LANES; ++j) { const float4 p = part_s[j][q]; tot.x += p.x; tot.y += p.y; tot.z += p.z; tot.w += p.w; }
            float v[4] = {tot.x / n, tot.y / n, tot.z / n, tot.w / n};       // np.dot(x, x.T) / np.float32(x.size)
            if (target) { const float4 t = *reinterpret_cast<const float4*>(target + i); v[0] -= t.x; v[1] -= t.y; v[2] -= t.z; v[3] -= t.w; }
            const bool mirror = r / bt < c / bt;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                out[r * out_ld + c + k] = v[k];
                acc[0] += v[k] * v[k];
                if (mirror) { out[(c + k) * out_ld + r] = v[k]; acc[0] += v[k] * v[k]; }
            }
        }
    }
    block_sum(acc, scratch);
    if (threadIdx.x == 0 && partial) partial[blockIdx.x] = acc[0];
}

int gram_fold_groups(const GramPlan& pl) { return pl.splits > 32 ? 32 : 0; }

hipError_t launch_gram_reduce(const float* slabs, float* folded, const float* target, float* out, int out_ld, float* partial,
                              int* n_partial, int C, double divisor, const GramPlan& pl, hipStream_t s)
{
    const int cc = C * C;
    int splits = pl.splits;
    // ST2_GRAM_REDUCE=2 keeps the two-stage reduction (read per launch)
    const char* two = getenv("ST2_GRAM_REDUCE");
    if (splits > 32 && C % 4 == 0 && pl.bt % 4 == 0 && !(two && *two == '2')) {
        const bool small = cc <= 128 * 128;
        const int grid = reduce_grid((size_t)cc, small ? 32 : 128, kMaxPartials);
        if (n_partial) *n_partial = grid;
        if (small) gram_reduce_wide_k<32><<<grid, 256, 0, s>>>(slabs, target, out, partial, cc, splits, (float)divisor, C, out_ld, pl.bt);
        else gram_reduce_wide_k<8><<<grid, 256, 0, s>>>(slabs, target, out, partial, cc, splits, (float)divisor, C, out_ld, pl.bt);
        return hipGetLastError();
    }
    const int groups = gram_fold_groups(pl);
    if (groups) {
        const int per = (pl.splits + groups - 1) / groups;
        gram_fold_k<<<dim3((cc + 255) / 256, groups), 256, 0, s>>>(slabs, folded, cc, pl.splits, per, C, pl.bt);
        slabs = folded;
        splits = (pl.splits + per - 1) / per;
    }
    const int grid = reduce_grid((size_t)cc, 256, kMaxPartials);
    if (n_partial) *n_partial = grid;
    gram_reduce_k<<<grid, 256, 0, s>>>(slabs, target, out, partial, cc, splits, (float)divisor, C, out_ld, pl.bt);
    return hipGetLastError();
}

}  // namespace st2
