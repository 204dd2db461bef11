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

constexpr int CC = kConvCC;          // K granularity of the packed weights
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

// One zero word every out-of-image / out-of-range lane of an LDS-DMA points at.
__device__ float g_zero_page[64];

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// Staging is done entirely by LDS-DMA (global_load_lds: no VGPR round trip, asynchronous), into a
// double-buffered LDS image; one barrier per chunk:
//     DMA(chunk c+1 -> buf[~c])  ||  MFMA(chunk c from buf[c])  ;  vmcnt(0) ; barrier
template <int BM, int ROWS, int WAVES_M, int WAVES_N, int CCK>
__global__ __launch_bounds__(NTHREADS) void conv3x3_mfma_f32(const ConvKArgs a)
{
    constexpr int TM = BM / WAVES_M / 32;        // 32-row MFMA tiles per wave along M
    constexpr int TN = ROWS / WAVES_N;           // image rows (32-pixel MFMA tiles) per wave
    static_assert(WAVES_M * WAVES_N == 4, "4 waves per workgroup");
    static_assert(TM >= 1 && TN >= 1 && CCK % CC == 0, "tile");
    constexpr int NSUB = CCK / CC;               // packed sub-slabs per staged chunk
    constexpr int IN_ROWS = ROWS + 2;
    constexpr int IN_PLANE = IN_ROWS * IN_W;
    constexpr int N_IN = CCK * IN_PLANE;                      // floats in the activation tile
    constexpr int N_IN_PAD = (N_IN + 63) / 64 * 64;
    constexpr int W_FLOATS = 9 * CCK * BM;                    // floats in the weight slab
    constexpr int BUF = W_FLOATS + N_IN_PAD;
    constexpr int W_INSTR = W_FLOATS / 4 / 64;                // dwordx4 wave-DMAs per slab
    constexpr int I_INSTR = N_IN_PAD / 64;                    // dword wave-DMAs per activation tile
    constexpr int W_PER_WAVE = (W_INSTR + 3) / 4;
    constexpr int I_PER_WAVE = (I_INSTR + 3) / 4;
    static_assert(W_FLOATS % 256 == 0, "weight slab is a whole number of 1-KiB DMA pieces");

    __shared__ __attribute__((aligned(16))) float smem[2 * BUF];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
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

    // ---- per-lane DMA sources that do not change from chunk to chunk ----
    int soff[I_PER_WAVE], scc[I_PER_WAVE];       // spatial offset (or -1) and channel-in-chunk
#pragma unroll
    for (int t = 0; t < I_PER_WAVE; ++t) {
        const int e = (wave + 4 * t) * 64 + lane;
        const int c = e / IN_PLANE;
        const int rem = e - c * IN_PLANE;
        const int rr = rem / IN_W;
        const int col = rem - rr * IN_W;
        const int gy = y0 - 1 + rr, gx = x0 - 1 + col;
        const bool ok = e < N_IN && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
        soff[t] = ok ? gy * a.W + gx : -1;
        scc[t] = c;
    }
    int woff[W_PER_WAVE];                        // float offset of this lane's float4 inside a slab
#pragma unroll
    for (int t = 0; t < W_PER_WAVE; ++t) {
        const int f = (wave + 4 * t) * 64 + lane;
        const int sub = f / (9 * CC * BM / 4);
        const int rem = f - sub * (9 * CC * BM / 4);
        const int row = rem / (BM / 4), qq = rem % (BM / 4);
        woff[t] = (sub * 9 * CC + row) * a.MPad + qq * 4;
    }

    auto dma_chunk = [&](int ch, int buf) {
        float* dst = smem + buf * BUF;
        const float* wsrc = a.wpack + (size_t)ch * NSUB * 9 * CC * a.MPad + m0;
#pragma unroll
        for (int t = 0; t < W_PER_WAVE; ++t) {
            const int i = wave + 4 * t;
            if (W_INSTR % 4 == 0 || i < W_INSTR)
                __builtin_amdgcn_global_load_lds((gptr_t)(wsrc + woff[t]), (lptr_t)(dst + i * 256), 16, 0, 0);
        }
        const int k0 = ch * CCK;
#pragma unroll
        for (int t = 0; t < I_PER_WAVE; ++t) {
            const int j = wave + 4 * t;
            if (I_INSTR % 4 == 0 || j < I_INSTR) {
                const int gk = k0 + scc[t];
                const float* src = (soff[t] >= 0 && gk < a.K) ? a.in + (size_t)gk * plane + soff[t] : g_zero_page;
                __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(dst + W_FLOATS + j * 64), 4, 0, 0);
            }
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
    const int a_off = khalf * BM + wave_m * (TM * 32) + l31;
    const int b_off = W_FLOATS + khalf * IN_PLANE + (wave_n * TN) * IN_W + l31;

    dma_chunk(0, 0);
    __syncthreads();                         // vmcnt(0) + barrier: chunk 0 has landed for every wave
    for (int ch = 0; ch < a.nch; ++ch) {
        const int cur = ch & 1;
        if (ch + 1 < a.nch) dma_chunk(ch + 1, cur ^ 1);     // lands while the MFMAs below run
        const float* a_base = smem + cur * BUF + a_off;
        const float* b_base = smem + cur * BUF + b_off;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int dy = tap / 3, dx = tap % 3;
#pragma unroll
            for (int kk = 0; kk < CCK / 2; ++kk) {
                constexpr int dummy = 0; (void)dummy;
                const int c = 2 * kk;                       // + khalf (folded into the bases)
                const int sub = c / CC, cc = c % CC;
                float av[TM], bv[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) av[i] = a_base[(sub * 9 * CC + tap * CC + cc) * BM + i * 32];
#pragma unroll
                for (int j = 0; j < TN; ++j) bv[j] = b_base[c * IN_PLANE + (j + dy) * IN_W + dx];
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], acc[i][j], 0, 0, 0);
            }
        }
        __syncthreads();                     // own DMAs done (vmcnt 0), everyone done reading buf[cur]
    }

    // ---- epilogue: C/D map of the 32x32 MFMA: col = lane&31 (pixel), row = (e&3)+8*(e>>2)+4*(lane>>5).
    // Loads of one 32x32 tile (bias / ReLU-mask source / injected diff) are issued as a batch of 16
    // independent, unconditional loads (rows beyond M are clamped, their stores skipped).
    const int gx = x0 + l31;
    const bool has_bias = a.bias != nullptr, has_mask = a.mask_src != nullptr, has_inj = a.inject != nullptr;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int gy = y0 + wave_n * TN + j;
        if (gy >= a.H || gx >= a.W) continue;
        const size_t pix = (size_t)gy * a.W + gx;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int mbase = m0 + wave_m * (TM * 32) + i * 32 + 4 * khalf;
            float v[16];
            size_t idx[16];
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = mbase + (e & 3) + 8 * (e >> 2);
                idx[e] = (size_t)(m < a.M ? m : a.M - 1) * plane + pix;
                v[e] = acc[i][j][e];
            }
            if (has_bias) {
#pragma unroll
                for (int e = 0; e < 16; ++e) v[e] += a.bias[mbase + (e & 3) + 8 * (e >> 2)];   // bias is MPad long
            }
            if (a.relu) {
#pragma unroll
                for (int e = 0; e < 16; ++e) v[e] = v[e] > 0.0f ? v[e] : 0.0f;
            }
            if (has_mask) {
                float mk[16];
#pragma unroll
                for (int e = 0; e < 16; ++e) mk[e] = a.mask_src[idx[e]];
#pragma unroll
                for (int e = 0; e < 16; ++e) v[e] = mk[e] > 0.0f ? v[e] : 0.0f;
            }
            if (has_inj) {
                float ij[16];
#pragma unroll
                for (int e = 0; e < 16; ++e) ij[e] = a.inject[idx[e]];
#pragma unroll
                for (int e = 0; e < 16; ++e) v[e] += ij[e];
            }
#pragma unroll
            for (int e = 0; e < 16; ++e)
                if (mbase + (e & 3) + 8 * (e >> 2) < a.M) a.out[idx[e]] = v[e];
        }
    }
}

template <int BM, int ROWS, int WAVES_M, int WAVES_N, int CCK>
static hipError_t run(const ConvProblem& p, hipStream_t s)
{
    ConvKArgs k;
    k.in = p.in; k.wpack = p.wpack; k.bias = p.bias; k.out = p.out;
    k.mask_src = p.mask_src; k.inject = p.inject;
    k.K = p.K; k.M = p.M; k.MPad = p.MPad; k.H = p.H; k.W = p.W;
    k.nch = (p.K + CCK - 1) / CCK;
    k.tiles_x = (p.W + 31) / 32;
    k.tiles_y = (p.H + ROWS - 1) / ROWS;
    k.n_mtiles = p.MPad / BM;
    k.relu = p.relu;
    const long long nblk = (long long)k.tiles_x * k.tiles_y * k.n_mtiles;
    if (nblk <= 0 || nblk > 0x7fffffffLL) return hipErrorInvalidValue;
    conv3x3_mfma_f32<BM, ROWS, WAVES_M, WAVES_N, CCK><<<dim3((unsigned)nblk), dim3(NTHREADS), 0, s>>>(k);
    return hipGetLastError();
}

int conv_num_configs() { return 6; }

const char* conv_config_name(int cfg)
{
    static const char* names[] = {"128x128px cc4", "128x256px cc4", "64x256px cc4", "64x128px cc4", "64x256px cc8", "128x128px cc8"};
    return cfg >= 0 && cfg < 6 ? names[cfg] : "?";
}

int conv_pick_config(const ConvProblem& p)
{
    const long long px8 = (long long)((p.W + 31) / 32) * ((p.H + 7) / 8);
    const long long px4 = (long long)((p.W + 31) / 32) * ((p.H + 3) / 4);
    if (p.MPad % 128 == 0) {
        // 128 x 256-pixel tiles while they still give every CU a few blocks, else 128 x 128,
        // else (few blocks: conv5_1) 64 x 128
        if (px8 * (p.MPad / 128) >= 1024) return 1;
        if (px4 * (p.MPad / 128) >= 512) return 0;
        return 3;
    }
    if (px8 * (p.MPad / 64) >= 1024) return 2;
    return 3;
}

hipError_t launch_conv3x3_cfg(const ConvProblem& p, int cfg, hipStream_t s)
{
    if (p.MPad % kCoutQuantum != 0 || p.MPad < p.M) return hipErrorInvalidValue;
    if (cfg < 0) cfg = conv_pick_config(p);
    if ((cfg == 0 || cfg == 1 || cfg == 5) && p.MPad % 128 != 0) return hipErrorInvalidValue;
    switch (cfg) {
    case 0: return run<128, 4, 2, 2, 4>(p, s);
    case 1: return run<128, 8, 2, 2, 4>(p, s);
    case 2: return run<64, 8, 1, 4, 4>(p, s);
    case 3: return run<64, 4, 1, 4, 4>(p, s);
    case 4: return run<64, 8, 1, 4, 8>(p, s);
    case 5: return run<128, 4, 2, 2, 8>(p, s);
    }
    return hipErrorInvalidValue;
}

hipError_t launch_conv3x3(const ConvProblem& p, hipStream_t s) { return launch_conv3x3_cfg(p, -1, s); }

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
