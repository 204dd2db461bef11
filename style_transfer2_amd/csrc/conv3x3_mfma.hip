// conv3x3 (pad 1, stride 1), NCHW fp32, as an implicit GEMM on the CDNA4 matrix cores.
//
// Replaces what the reference reaches through pycaffe: Convolution+ReLU forward (worker.py:84-86,
// models/vgg19.prototxt:11-27 ...) and the bottom-diff half of Convolution backward plus the ReLU
// backward below it (worker.py:100-106).  Weight gradients are never needed (SURVEY 8a A4 iii).
//
// GEMM view: D[m][p] = sum_{k,tap} Wp[tap][k][m] * In[k][p + tap],  m = output channel,
// p = pixel, k = input channel.  One workgroup (4 waves) owns BM output channels x (ROWS x 32)
// pixels.  Per chunk of CC input channels it stages in LDS
//     in_s [CC][ROWS+2][34]   the activation tile with its 1-pixel halo (zero outside the image)
//     w_s  [9][CC][BM]        the weight slab, m contiguous
// and issues v_mfma_f32_32x32x2_f32 with A = weights (lane&31 -> m, lane>>5 -> k) and
// B = activations (lane&31 -> 32 consecutive pixels of one row, lane>>5 -> k): every LDS read
// is a conflict-free ds_read_b32 with an immediate offset, all 9 taps reuse the one staged tile.
// The next chunk's global loads are issued before the MFMA block and parked in registers.
// fp32 MFMA == a k-ordered fmaf chain, so results are plain IEEE fp32.
#include "st2_kernels.h"
#include <hip/hip_runtime.h>
#include <string.h>

namespace st2 {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int CC = kConvCC;
constexpr int IN_W = 34;
constexpr int NTHREADS = 256;

int conv_mpad(int M) { return (M + kCoutQuantum - 1) / kCoutQuantum * kCoutQuantum; }

size_t conv_pack_floats(int K, int M)
{
    const int nch = (K + CC - 1) / CC;
    return (size_t)nch * 9 * CC * conv_mpad(M);
}

// packed[ch][tap][c][m] = w[m][ch*CC + c][tap]       (forward: m = Cout, k = Cin)
void pack_conv_weights_fwd(const float* w, int Cout, int Cin, float* dst)
{
    const int mpad = conv_mpad(Cout);
    memset(dst, 0, conv_pack_floats(Cin, Cout) * sizeof(float));
    for (int m = 0; m < Cout; ++m)
        for (int k = 0; k < Cin; ++k)
            for (int tap = 0; tap < 9; ++tap)
                dst[(((size_t)(k / CC) * 9 + tap) * CC + (k % CC)) * mpad + m] =
                    w[((size_t)m * Cin + k) * 9 + tap];
}

// dgrad: m = Cin, k = Cout, taps flipped:  packed[ch][tap][c][m] = w[ch*CC + c][m][8 - tap]
void pack_conv_weights_dgrad(const float* w, int Cout, int Cin, float* dst)
{
    const int mpad = conv_mpad(Cin);
    memset(dst, 0, conv_pack_floats(Cout, Cin) * sizeof(float));
    for (int k = 0; k < Cout; ++k)
        for (int m = 0; m < Cin; ++m)
            for (int tap = 0; tap < 9; ++tap)
                dst[(((size_t)(k / CC) * 9 + tap) * CC + (k % CC)) * mpad + m] =
                    w[((size_t)k * Cin + m) * 9 + (8 - tap)];
}

struct ConvKArgs {
    const float* in; const float* wpack; const float* bias; float* out;
    const float* mask_src; const float* inject;
    int K, M, MPad, H, W, nch, tiles_x, tiles_y, n_mtiles, relu;
};

template <int BM, int ROWS, int WAVES_M, int WAVES_N>
__global__ __launch_bounds__(NTHREADS) void conv3x3_mfma_f32(const ConvKArgs a)
{
    constexpr int TM = BM / WAVES_M / 32;        // 32-row MFMA tiles per wave along M
    constexpr int TN = ROWS / WAVES_N;           // image rows (32-pixel MFMA tiles) per wave
    static_assert(WAVES_M * WAVES_N == 4, "4 waves per workgroup");
    static_assert(TM >= 1 && TN >= 1, "tile");
    constexpr int IN_ROWS = ROWS + 2;
    constexpr int IN_PLANE = IN_ROWS * IN_W;
    constexpr int N_IN = CC * IN_PLANE;                       // floats in the activation tile
    constexpr int N_W4 = 9 * CC * BM / 4;                     // float4s in the weight slab
    constexpr int IN_PER_T = (N_IN + NTHREADS - 1) / NTHREADS;
    constexpr int W4_PER_T = (N_W4 + NTHREADS - 1) / NTHREADS;

    __shared__ __attribute__((aligned(16))) float smem[9 * CC * BM + N_IN];
    float* w_s = smem;
    float* in_s = smem + 9 * CC * BM;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wave_m = wave / WAVES_N;
    const int wave_n = wave % WAVES_N;

    // XCD-aware, bijective block -> tile map: the blocks that share an XCD (same blockIdx % 8)
    // get a contiguous run of logical ids, m-tile fastest, so the co-resident blocks of one L2
    // read the same activation tile and neighbouring halos.
    const int nwg = gridDim.x, orig = blockIdx.x;
    const int xcd = orig & 7, q = nwg >> 3, r = nwg & 7;
    const int logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
    const int mt = logical % a.n_mtiles;
    const int pt = logical / a.n_mtiles;
    const int tx = pt % a.tiles_x;
    const int ty = pt / a.tiles_x;
    const int m0 = mt * BM;
    const int y0 = ty * ROWS;
    const int x0 = tx * 32;

    const size_t plane = (size_t)a.H * a.W;

    // ---- staging registers ----
    float4 wreg[W4_PER_T];
    float ireg[IN_PER_T];

    auto load_chunk = [&](int ch) {
        const float* wsrc = a.wpack + (size_t)ch * 9 * CC * a.MPad + m0;
#pragma unroll
        for (int i = 0; i < W4_PER_T; ++i) {
            const int f = tid + i * NTHREADS;
            if (N_W4 % NTHREADS == 0 || f < N_W4) {
                const int row = f / (BM / 4), qq = f % (BM / 4);
                wreg[i] = *reinterpret_cast<const float4*>(wsrc + (size_t)row * a.MPad + qq * 4);
            }
        }
        const int k0 = ch * CC;
#pragma unroll
        for (int i = 0; i < IN_PER_T; ++i) {
            const int e = tid + i * NTHREADS;
            float v = 0.0f;
            if (N_IN % NTHREADS == 0 || e < N_IN) {
                const int c = e / IN_PLANE;
                const int rem = e - c * IN_PLANE;
                const int rr = rem / IN_W;
                const int col = rem - rr * IN_W;
                const int gy = y0 - 1 + rr, gx = x0 - 1 + col, gk = k0 + c;
                if (gk < a.K && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W)
                    v = a.in[(size_t)gk * plane + (size_t)gy * a.W + gx];
            }
            ireg[i] = v;
        }
    };
    auto store_chunk = [&]() {
#pragma unroll
        for (int i = 0; i < W4_PER_T; ++i) {
            const int f = tid + i * NTHREADS;
            if (N_W4 % NTHREADS == 0 || f < N_W4)
                *reinterpret_cast<float4*>(w_s + f * 4) = wreg[i];
        }
#pragma unroll
        for (int i = 0; i < IN_PER_T; ++i) {
            const int e = tid + i * NTHREADS;
            if (N_IN % NTHREADS == 0 || e < N_IN) in_s[e] = ireg[i];
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

    const int khalf = lane >> 5;             // which of the 2 k's of an MFMA this lane feeds
    const int l31 = lane & 31;
    const float* a_base = w_s + khalf * BM + wave_m * (TM * 32) + l31;
    const float* b_base = in_s + khalf * IN_PLANE + (wave_n * TN) * IN_W + l31;

    load_chunk(0);
    for (int ch = 0; ch < a.nch; ++ch) {
        __syncthreads();                     // everyone is done reading the previous tile
        store_chunk();
        __syncthreads();
        if (ch + 1 < a.nch) load_chunk(ch + 1);   // in flight while the MFMAs below run
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int dy = tap / 3, dx = tap % 3;
#pragma unroll
            for (int kk = 0; kk < CC / 2; ++kk) {
                float av[TM], bv[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) av[i] = a_base[(tap * CC + 2 * kk) * BM + i * 32];
#pragma unroll
                for (int j = 0; j < TN; ++j) bv[j] = b_base[2 * kk * IN_PLANE + (j + dy) * IN_W + dx];
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], acc[i][j], 0, 0, 0);
            }
        }
    }

    // ---- epilogue: C/D map of the 32x32 MFMA: col = lane&31 (pixel), row = (e&3)+8*(e>>2)+4*(lane>>5) ----
    const int gx = x0 + l31;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int gy = y0 + wave_n * TN + j;
        if (gy >= a.H || gx >= a.W) continue;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = m0 + wave_m * (TM * 32) + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * khalf;
                if (m >= a.M) continue;
                const size_t idx = (size_t)m * plane + (size_t)gy * a.W + gx;
                float v = acc[i][j][e];
                if (a.bias) v += a.bias[m];
                if (a.relu) v = v > 0.0f ? v : 0.0f;
                if (a.mask_src) v = a.mask_src[idx] > 0.0f ? v : 0.0f;
                if (a.inject) v += a.inject[idx];
                a.out[idx] = v;
            }
        }
    }
}

template <int BM, int ROWS, int WAVES_M, int WAVES_N>
static hipError_t run(const ConvProblem& p, hipStream_t s)
{
    ConvKArgs k;
    k.in = p.in; k.wpack = p.wpack; k.bias = p.bias; k.out = p.out;
    k.mask_src = p.mask_src; k.inject = p.inject;
    k.K = p.K; k.M = p.M; k.MPad = p.MPad; k.H = p.H; k.W = p.W;
    k.nch = (p.K + CC - 1) / CC;
    k.tiles_x = (p.W + 31) / 32;
    k.tiles_y = (p.H + ROWS - 1) / ROWS;
    k.n_mtiles = p.MPad / BM;
    k.relu = p.relu;
    const long long nblk = (long long)k.tiles_x * k.tiles_y * k.n_mtiles;
    if (nblk <= 0 || nblk > 0x7fffffffLL) return hipErrorInvalidValue;
    conv3x3_mfma_f32<BM, ROWS, WAVES_M, WAVES_N><<<dim3((unsigned)nblk), dim3(NTHREADS), 0, s>>>(k);
    return hipGetLastError();
}

hipError_t launch_conv3x3(const ConvProblem& p, hipStream_t s)
{
    if (p.MPad % kCoutQuantum != 0 || p.MPad < p.M) return hipErrorInvalidValue;
    const long long px8 = (long long)((p.W + 31) / 32) * ((p.H + 7) / 8);
    if (p.MPad % 128 == 0) {
        // 128 x 256-pixel tiles while they still give every CU a few blocks, else 128 x 128
        if (px8 * (p.MPad / 128) >= 1024) return run<128, 8, 2, 2>(p, s);
        return run<128, 4, 2, 2>(p, s);
    }
    if (px8 * (p.MPad / 64) >= 1024) return run<64, 8, 1, 4>(p, s);
    return run<64, 4, 1, 4>(p, s);
}

// ------------------------------------------------------------------------------------------
// dgrad with a tiny output-channel count (conv1_1 -> image, M = 3): not GEMM shaped, so a direct
// VALU kernel.  One thread per output pixel, all M (<= 4) channels; weights (Cout*M*9 floats) in LDS.
// dx[m][y][x] = sum_{co,ky,kx} w[co][m][ky][kx] * dy[co][y-ky+1][x-kx+1]  (+ inject)
// ------------------------------------------------------------------------------------------
constexpr int SM_MAXM = 4;
constexpr int SM_TX = 32, SM_TY = 8;

__global__ __launch_bounds__(256) void conv3x3_dgrad_smallM(const float* __restrict__ dy, const float* __restrict__ w,
                                                            float* __restrict__ dx, const float* __restrict__ inject,
                                                            int Cout, int M, int H, int W)
{
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* w_s = sm;                                   // [Cout][M][9]
    float* t_s = sm + Cout * M * 9;                    // [8 channels][SM_TY+2][SM_TX+2]
    constexpr int TW = SM_TX + 2, TH = SM_TY + 2, CH = 8;
    const int tid = threadIdx.x;
    for (int i = tid; i < Cout * M * 9; i += 256) w_s[i] = w[i];
    const int x0 = blockIdx.x * SM_TX, y0 = blockIdx.y * SM_TY;
    const int lx = tid % SM_TX, ly = tid / SM_TX;
    const size_t plane = (size_t)H * W;
    float acc[SM_MAXM] = {0.f, 0.f, 0.f, 0.f};
    for (int c0 = 0; c0 < Cout; c0 += CH) {
        __syncthreads();
        for (int e = tid; e < CH * TH * TW; e += 256) {
            const int c = e / (TH * TW), rem = e % (TH * TW), rr = rem / TW, col = rem % TW;
            const int gy = y0 - 1 + rr, gx = x0 - 1 + col;
            float v = 0.f;
            if (c0 + c < Cout && gy >= 0 && gy < H && gx >= 0 && gx < W)
                v = dy[(size_t)(c0 + c) * plane + (size_t)gy * W + gx];
            t_s[e] = v;
        }
        __syncthreads();
        for (int c = 0; c < CH && c0 + c < Cout; ++c) {
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    // source pixel (y - ky + 1, x - kx + 1) -> tile coords (ly + 2 - ky, lx + 2 - kx)
                    const float g = t_s[(c * TH + (ly + 2 - ky)) * TW + (lx + 2 - kx)];
#pragma unroll
                    for (int m = 0; m < SM_MAXM; ++m)
                        if (m < M) acc[m] += w_s[((c0 + c) * M + m) * 9 + ky * 3 + kx] * g;
                }
        }
    }
    const int gx = x0 + lx, gy = y0 + ly;
    if (gx < W && gy < H)
        for (int m = 0; m < M; ++m) {
            const size_t idx = (size_t)m * plane + (size_t)gy * W + gx;
            dx[idx] = acc[m] + (inject ? inject[idx] : 0.f);
        }
}

static size_t smallM_lds(int Cout, int Cin)
{
    return ((size_t)Cout * Cin * 9 + 8 * (SM_TY + 2) * (SM_TX + 2)) * sizeof(float);
}

bool conv_dgrad_smallM_ok(int Cout, int Cin) { return Cin <= SM_MAXM && smallM_lds(Cout, Cin) <= 64 * 1024; }

hipError_t launch_conv3x3_dgrad_smallM(const float* dy, const float* w, float* dx, const float* inject,
                                       int Cout, int Cin, int H, int W, hipStream_t s)
{
    if (!conv_dgrad_smallM_ok(Cout, Cin)) return hipErrorInvalidValue;
    const size_t lds = smallM_lds(Cout, Cin);
    dim3 grid((W + SM_TX - 1) / SM_TX, (H + SM_TY - 1) / SM_TY);
    conv3x3_dgrad_smallM<<<grid, dim3(256), lds, s>>>(dy, w, dx, inject, Cout, Cin, H, W);
    return hipGetLastError();
}

}  // namespace st2
