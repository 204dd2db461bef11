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
#include <type_traits>
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <string.h>

namespace st2 {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int CC = kConvCC;          // K granularity of the packed weights
constexpr int IN_W = 40;            // LDS activation row: pixels x0-4 .. x0+35 (10 aligned quads)
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
    unsigned short* out16;        // optional bf16 copy of the output, channel-blocked [M/8][H][W][8] (M % 8 == 0)
    int K, M, MPad, H, W, nch, tiles_x, tiles_y, n_mtiles, relu;
    unsigned in_bytes, w_bytes;   // extents of `in` and `wpack` for the buffer descriptors
    unsigned long long* stamps;   // DIAG builds only: per block {shader cycles, 100 MHz ticks} of the main loop
    // style-gradient epilogue (EPI_STYLE): out = c2*acc [raw]  or  (sw / *norm) * c2*acc + (accumulate ? out : 0)
    float c2, sw; const float* norm; int fused, accumulate; float* partial;
    int ry0, rx0, ry1, rx1;       // region of interest of the style epilogue (tile-sharded mode); whole blob otherwise
};

enum { EPI_CONV = 0, EPI_STYLE = 1 };

typedef __attribute__((address_space(3))) void* lptr_t;
constexpr unsigned kOOB = 0xffffffffu;       // a buffer voffset beyond num_records: the DMA writes zeros

// Staging is done entirely by LDS-DMA (buffer_load ... lds: no VGPR round trip, asynchronous,
// hardware zero-fill for out-of-range lanes = the conv's zero padding) into a double-buffered LDS
// image.  Every per-lane offset is computed ONCE; per chunk only a scalar offset advances.
//   ALIGNED (W % 4 == 0): an activation row is 10 aligned quads (x0-4 .. x0+35) -> dwordx4 DMA
//   else                 : the same 40-float row image, one dword per lane
// DIAG = 1 adds s_memtime/s_memrealtime stamps around the main loop (measurement builds only).
// TAPS = 9: the 3x3 conv.  TAPS = 1: a 1x1 "conv" (no halo) = the style-gradient GEMM S = D @ F with the
// (symmetric) Gram difference D as the weight matrix, on the same pipeline.
template <int BM, int ROWS, int WAVES_M, int WAVES_N, int CCK, bool ALIGNED, int DIAG, int TAPS = 9, int EPI = EPI_CONV>
__device__ __forceinline__ void conv3x3_body(const ConvKArgs& a)
{
    constexpr int HALO = TAPS == 9 ? 1 : 0;
    constexpr int IW = HALO ? IN_W : 32;         // floats per staged activation row
    constexpr int TM = BM / WAVES_M / 32;        // 32-row MFMA tiles per wave along M
    constexpr int TN = ROWS / WAVES_N;           // image rows (32-pixel MFMA tiles) per wave
    static_assert(WAVES_M * WAVES_N == 4, "4 waves per workgroup");
    static_assert(TM >= 1 && TN >= 1 && CCK % CC == 0, "tile");
    constexpr int NSUB = CCK / CC;               // packed sub-slabs per staged chunk
    constexpr int IN_ROWS = ROWS + 2 * HALO;
    constexpr int IN_PLANE = IN_ROWS * IW;
    constexpr int N_IN = CCK * IN_PLANE;                      // floats in the activation tile
    constexpr int I_LANE = ALIGNED ? 4 : 1;                   // floats per lane per DMA
    constexpr int I_INSTR = (N_IN / I_LANE + 63) / 64;        // wave-DMAs per activation tile
    constexpr int N_IN_PAD = I_INSTR * 64 * I_LANE;
    constexpr int W_FLOATS = TAPS * CCK * BM;                 // floats in the weight slab
    constexpr int BUF = W_FLOATS + N_IN_PAD;
    constexpr int W_INSTR = W_FLOATS / 4 / 64;                // dwordx4 wave-DMAs per slab
    constexpr int W_PER_WAVE = (W_INSTR + 3) / 4;
    constexpr int I_PER_WAVE = (I_INSTR + 3) / 4;
    constexpr int NPIECE = W_PER_WAVE + I_PER_WAVE;
    constexpr int NSTEP = TAPS * (CCK / 2);
    static_assert(NSTEP % 2 == 0, "two operand register sets alternate per step");
    static_assert(W_FLOATS % 256 == 0, "weight slab is a whole number of 1-KiB DMA pieces");
    constexpr int PPS = (NPIECE + NSTEP - 2) / (NSTEP - 1);   // DMA pieces per k-step (the last step carries the barrier instead)
    static_assert(PPS * (NSTEP - 1) >= NPIECE, "every DMA piece has a k-step to ride on");

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

    const unsigned plane = (unsigned)a.H * a.W;

    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)a.wpack, 0, a.w_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_i = __builtin_amdgcn_make_buffer_rsrc((void*)a.in, 0, a.in_bytes, 0x00020000);

    // ---- per-lane DMA byte offsets, constant over the whole K loop ----
    unsigned ioff[I_PER_WAVE];
#pragma unroll
    for (int t = 0; t < I_PER_WAVE; ++t) {
        const int e = ((wave + 4 * t) * 64 + lane) * I_LANE;      // first float of this lane in the tile image
        const int c = e / IN_PLANE;
        const int rem = e - c * IN_PLANE;
        const int rr = rem / IW;
        const int col = rem - rr * IW;
        const int gy = y0 - HALO + rr, gx = x0 - 4 * HALO + col;
        const bool ok = e < N_IN && gy >= 0 && gy < a.H && gx >= 0 && gx + (I_LANE - 1) < a.W;
        ioff[t] = ok ? ((unsigned)c * plane + (unsigned)gy * a.W + gx) * 4u : kOOB;
    }
    unsigned woff[W_PER_WAVE];
#pragma unroll
    for (int t = 0; t < W_PER_WAVE; ++t) {
        const int f = (wave + 4 * t) * 64 + lane;
        const int sub = f / (TAPS * CC * BM / 4);
        const int rem = f - sub * (TAPS * CC * BM / 4);
        const int row = rem / (BM / 4), qq = rem % (BM / 4);
        woff[t] = (unsigned)((sub * TAPS * CC + row) * a.MPad + qq * 4) * 4u;
    }

    // One DMA "piece" = one wave-instruction (1 KiB of weights; 64 quads/words of activations).
    // The chunk offset is added to the per-lane offset (not passed as soffset) so that the buffer
    // range check covers it: rows / channels past the end of a ragged last chunk arrive as zeros.
    auto dma_piece = [&](int t, int ch, int buf) {
        float* dst = smem + buf * BUF;
        if (t < W_PER_WAVE) {
            const int i = wave + 4 * t;
            if (W_INSTR % 4 == 0 || i < W_INSTR) {
                const unsigned soff = ((unsigned)ch * NSUB * TAPS * CC * a.MPad + m0) * 4u;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lptr_t)(dst + i * 256), 16, woff[t] + soff, 0, 0, 0);
            }
        } else {
            const int u = t - W_PER_WAVE;
            const int j = wave + 4 * u;
            if (I_INSTR % 4 == 0 || j < I_INSTR) {
                const unsigned coff = (unsigned)ch * CCK * plane * 4u;
                const unsigned vo = ioff[u] == kOOB ? kOOB : ioff[u] + coff;
                const unsigned soff = 0;
                if (ALIGNED) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_i, (lptr_t)(dst + W_FLOATS + j * 256), 16, vo, soff, 0, 0);
                else __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_i, (lptr_t)(dst + W_FLOATS + j * 64), 4, vo, soff, 0, 0);
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
    const int b_off = W_FLOATS + khalf * IN_PLANE + (wave_n * TN) * IW + l31 + 3 * HALO;   // halo image: col 3 = pixel x0-1

    // k-steps (tap-major inside a chunk) form one continuous stream across chunks.  Per step:
    //   MFMA #0 | LDS reads of the NEXT step (second register set) | one DMA piece | MFMA #1..
    // so LDS latency hides under >= 3 queued MFMAs and the DMA of chunk c+1 is spread over the first
    // steps of chunk c.  The chunk boundary (own DMAs landed: vmcnt(0); all waves done with the
    // buffer: barrier) sits inside the last step, between MFMA #0 and the first reads of the next chunk.
    float av[2][TM], bv[2][TN];
    auto fetch = [&](const float* a_base, const float* b_base, int s2, float (&ao)[TM], float (&bo)[TN]) {
        const int tap = s2 / (CCK / 2), kk = s2 % (CCK / 2);
        const int dy = HALO ? tap / 3 : 0, dx = HALO ? tap % 3 : 0;
        const int c = 2 * kk;                           // + khalf (folded into the bases)
        const int sub = c / CC, cc = c % CC;
#pragma unroll
        for (int i = 0; i < TM; ++i) ao[i] = a_base[(sub * TAPS * CC + tap * CC + cc) * BM + i * 32];
#pragma unroll
        for (int j = 0; j < TN; ++j) bo[j] = b_base[c * IN_PLANE + (j + dy) * IW + dx];
    };

#pragma unroll
    for (int t = 0; t < NPIECE; ++t) dma_piece(t, 0, 0);
    __syncthreads();                         // vmcnt(0) + barrier: chunk 0 has landed for every wave
    unsigned long long t0 = 0, r0 = 0;
    if (DIAG) { t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }
    fetch(smem + a_off, smem + b_off, 0, av[0], bv[0]);
    for (int ch = 0; ch < a.nch; ++ch) {
        const int cur = ch & 1;
        const bool more = ch + 1 < a.nch;
        const float* a_base = smem + cur * BUF + a_off;
        const float* b_base = smem + cur * BUF + b_off;
        const float* a_next = smem + (cur ^ 1) * BUF + a_off;
        const float* b_next = smem + (cur ^ 1) * BUF + b_off;
#pragma unroll
        for (int s2 = 0; s2 < NSTEP; ++s2) {
            // NSTEP is even, so the register set of a step is (s2 & 1) in every chunk
#pragma unroll
            for (int ij = 0; ij < TM * TN; ++ij) {
                const int i = ij / TN, j = ij % TN;
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s2 & 1][i], bv[s2 & 1][j], acc[i][j], 0, 0, 0);
                if (ij == 0) {
                    __builtin_amdgcn_sched_barrier(0);
                    if (s2 + 1 < NSTEP) {
                        fetch(a_base, b_base, s2 + 1, av[(s2 + 1) & 1], bv[(s2 + 1) & 1]);
                        if (more) {
#pragma unroll
                            for (int pp = 0; pp < PPS; ++pp)
                                if (s2 * PPS + pp < NPIECE) dma_piece(s2 * PPS + pp, ch + 1, cur ^ 1);
                        }
                    } else if (more) {
                        __syncthreads();
                        fetch(a_next, b_next, 0, av[0], bv[0]);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }

    if (DIAG) {
        const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
        if (tid == 0 && a.stamps) { a.stamps[2 * blockIdx.x] = t1 - t0; a.stamps[2 * blockIdx.x + 1] = r1 - r0; }
    }
    // ---- epilogue: C/D map of the 32x32 MFMA: col = lane&31 (pixel), row = (e&3)+8*(e>>2)+4*(lane>>5).
    // Loads of one 32x32 tile (bias / ReLU-mask source / injected diff) are issued as a batch of 16
    // independent, unconditional loads (rows beyond M are clamped, their stores skipped).
    // 32-bit element offsets from the (uniform) tensor bases: the tensors are < 4 GiB (checked at launch).
    const int gx = x0 + l31;
    if (EPI == EPI_STYLE) {
        // S = c2 * (D @ F); partial[block] = sum S^2; raw: out = S; fused: out = (sw / norm) * S (+ out)
        const float coef = a.fused ? a.sw / *a.norm : 0.f;
        float ss = 0.f;
        const bool full_m = m0 + BM <= a.M;                  // uniform: no per-channel bounds tests on a tile inside M
        auto style_out = [&](auto full_t) {
        constexpr bool FULL = decltype(full_t)::value;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int gy = y0 + wave_n * TN + j;
            if (gy >= a.H || gx >= a.W) continue;
            const unsigned pix = (unsigned)gy * a.W + gx;
            const bool in_roi = gy >= a.ry0 && gy < a.ry1 && gx >= a.rx0 && gx < a.rx1;
            if (!in_roi && a.fused && a.accumulate) continue;      // nothing to add outside the region
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int mbase = m0 + wave_m * (TM * 32) + i * 32 + 4 * khalf + 16 * h;
                    float v[8], old[8];
                    unsigned off[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const int m = mbase + (e & 3) + 8 * (e >> 2);
                        off[e] = (unsigned)((FULL || m < a.M) ? m : a.M - 1) * plane + pix;
                        v[e] = in_roi ? acc[i][j][8 * h + e] * a.c2 : 0.f;
                        if (FULL || m < a.M) ss += v[e] * v[e];
                    }
                    if (a.fused && a.accumulate) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) old[e] = a.out[off[e]];
                    }
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float r2 = a.fused ? coef * v[e] + (a.accumulate ? old[e] : 0.f) : v[e];
                        if (FULL || mbase + (e & 3) + 8 * (e >> 2) < a.M) a.out[off[e]] = r2;
                    }
                }
        }
        };
        if (full_m) style_out(std::true_type{}); else style_out(std::false_type{});
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) ss += __shfl_down(ss, o, 64);
        __syncthreads();                     // every wave is done with the staged tiles
        if (lane == 0) smem[wave] = ss;
        __syncthreads();
        if (tid == 0) a.partial[blockIdx.x] = (smem[0] + smem[1]) + (smem[2] + smem[3]);
        return;
    }
    const bool has_bias = a.bias != nullptr, has_mask = a.mask_src != nullptr, has_inj = a.inject != nullptr;
    const bool full_m = m0 + BM <= a.M;                      // uniform: no per-channel bounds tests on a tile inside M
    auto conv_out = [&](auto full_t) {
    constexpr bool FULL = decltype(full_t)::value;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int gy = y0 + wave_n * TN + j;
        if (gy >= a.H || gx >= a.W) continue;
        const unsigned pix = (unsigned)gy * a.W + gx;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {            // two batches of 8 accumulator rows: fewer live registers
                const int mbase = m0 + wave_m * (TM * 32) + i * 32 + 4 * khalf + 16 * h;
                float v[8];
                unsigned off[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int m = mbase + (e & 3) + 8 * (e >> 2);
                    off[e] = (unsigned)((FULL || m < a.M) ? m : a.M - 1) * plane + pix;
                    v[e] = acc[i][j][8 * h + e];
                }
                if (has_bias) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] += a.bias[mbase + (e & 3) + 8 * (e >> 2)];   // bias is MPad long
                }
                if (a.relu) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = v[e] > 0.0f ? v[e] : 0.0f;
                }
                if (has_mask) {
                    float mk[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) mk[e] = a.mask_src[off[e]];
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = mk[e] > 0.0f ? v[e] : 0.0f;
                }
                if (has_inj) {
                    float ij[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) ij[e] = a.inject[off[e]];
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] += ij[e];
                }
                if (a.out) {                         // optional: the bf16 feature path may want the bf16 copy only
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        if (FULL || mbase + (e & 3) + 8 * (e >> 2) < a.M) a.out[off[e]] = v[e];
                }
                if (a.out16) {
                    // rows mbase..+3 and mbase+8..+11: this lane holds HALF (4 channels) of two 8-channel quads of its pixel, lane ^ 32 the
                    // other halves.  v_permlane32_swap exchanges them so that lanes 0-31 own the first quad and lanes 32-63 the second:
                    // one 16-byte store per lane (512 contiguous bytes per half wave) instead of two interleaved 8-byte ones.
                    typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
                    bf16x4 pk0, pk1;
#pragma unroll
                    for (int e = 0; e < 4; ++e) { pk0[e] = (__bf16)v[e]; pk1[e] = (__bf16)v[4 + e]; }
                    const uint2 u0 = __builtin_bit_cast(uint2, pk0), u1 = __builtin_bit_cast(uint2, pk1);
                    const auto sx = __builtin_amdgcn_permlane32_swap(u0.x, u1.x, false, false);    // [0]: {u0 of lanes 0-31 | u1 of lanes 0-31}
                    const auto sy = __builtin_amdgcn_permlane32_swap(u0.y, u1.y, false, false);    // [1]: {u0 of lanes 32-63 | u1 of lanes 32-63}
                    const int mq = mbase - 4 * khalf + 8 * khalf;                                  // first channel of this lane's quad
                    if (FULL || mq < a.M)       // M is a multiple of 8 on this path (checked at launch)
                        *reinterpret_cast<uint4*>(a.out16 + ((size_t)(mq >> 3) * plane + pix) * 8) = make_uint4(sx[0], sy[0], sx[1], sy[1]);
                }
            }
        }
    }
    };
    if (full_m) conv_out(std::true_type{}); else conv_out(std::false_type{});
}

// Non-template kernel entry points (the waves-per-SIMD launch bound must be a literal).
#define ST2_CONV_KERNEL(NAME, BM, ROWS, WM, WN, CCK, DIAG, WPE)                                          \
    __global__ __launch_bounds__(NTHREADS, WPE) void NAME##_q(const ConvKArgs a) { conv3x3_body<BM, ROWS, WM, WN, CCK, true, DIAG>(a); } \
    __global__ __launch_bounds__(NTHREADS, WPE) void NAME##_w(const ConvKArgs a) { conv3x3_body<BM, ROWS, WM, WN, CCK, false, DIAG>(a); }
ST2_CONV_KERNEL(conv3x3_mfma_f32_128x128_cc4, 128, 4, 2, 2, 4, 0, 3)
ST2_CONV_KERNEL(conv3x3_mfma_f32_128x256_cc4, 128, 8, 2, 2, 4, 0, 2)
ST2_CONV_KERNEL(conv3x3_mfma_f32_64x256_cc4, 64, 8, 1, 4, 4, 0, 3)
ST2_CONV_KERNEL(conv3x3_mfma_f32_64x128_cc4, 64, 4, 1, 4, 4, 0, 3)
ST2_CONV_KERNEL(conv3x3_mfma_f32_64x256_cc8, 64, 8, 1, 4, 8, 0, 2)
ST2_CONV_KERNEL(conv3x3_mfma_f32_128x128_cc8, 128, 4, 2, 2, 8, 0, 1)
ST2_CONV_KERNEL(conv3x3_mfma_f32_128x128_cc4_stamped, 128, 4, 2, 2, 4, 1, 3)
ST2_CONV_KERNEL(conv3x3_mfma_f32_64x64_cc4, 64, 2, 2, 2, 4, 0, 4)
#define ST2_STYLE_KERNEL(NAME, BM, ROWS, WM, WN, CCK, WPE)                                                \
    __global__ __launch_bounds__(NTHREADS, WPE) void NAME##_q(const ConvKArgs a) { conv3x3_body<BM, ROWS, WM, WN, CCK, true, 0, 1, EPI_STYLE>(a); } \
    __global__ __launch_bounds__(NTHREADS, WPE) void NAME##_w(const ConvKArgs a) { conv3x3_body<BM, ROWS, WM, WN, CCK, false, 0, 1, EPI_STYLE>(a); }
ST2_STYLE_KERNEL(style_grad_mfma_f32_128x128_cc32, 128, 4, 2, 2, 32, 2)
ST2_STYLE_KERNEL(style_grad_mfma_f32_64x256_cc16, 64, 8, 1, 4, 16, 3)
ST2_STYLE_KERNEL(style_grad_mfma_f32_64x128_cc32, 64, 4, 1, 4, 32, 3)

typedef void (*conv_kernel_t)(const ConvKArgs);

static hipError_t run(const ConvProblem& p, int BM, int ROWS, int CCK, conv_kernel_t k_quad, conv_kernel_t k_word, hipStream_t s,
                      const ConvKArgs* style = nullptr, int* n_blocks = nullptr)
{
    ConvKArgs k{};
    if (style) k = *style;
    k.in = p.in; k.wpack = p.wpack; k.bias = p.bias; k.out = p.out;
    k.mask_src = p.mask_src; k.inject = p.inject;
    if (p.out16 && p.M % 8 != 0) return hipErrorInvalidValue;
    if (!p.out && (style || !p.out16)) return hipErrorInvalidValue;       // nothing to write
    k.out16 = style ? nullptr : p.out16;
    k.K = p.K; k.M = p.M; k.MPad = p.MPad; k.H = p.H; k.W = p.W;
    k.nch = (p.K + CCK - 1) / CCK;
    k.tiles_x = (p.W + 31) / 32;
    k.tiles_y = (p.H + ROWS - 1) / ROWS;
    k.n_mtiles = p.MPad / BM;
    k.relu = p.relu;
    k.stamps = p.stamps;
    // 32-bit buffer addressing: activations, weight pack and output must each be < 4 GiB
    const unsigned long long in_bytes = 4ull * p.K * p.H * p.W;
    const unsigned long long w_bytes = style ? 4ull * p.K * p.MPad : 4ull * conv_pack_floats(p.K, p.M);
    const unsigned long long out_bytes = 4ull * p.M * p.H * p.W;
    // (the input tile is staged through 32-bit BYTE offsets; the output, ReLU mask and injected diff by 32-bit ELEMENT offsets:
    //  conv1_1's 17 GB blob of an 8192 x 8192 image -- 2^32 elements -- is addressable, a 4 GiB input is not)
    (void)out_bytes;
    if (in_bytes >= 0xfffffff0ull || w_bytes >= 0xfffffff0ull || (unsigned long long)p.M * p.H * p.W > 0x100000000ull) return hipErrorInvalidValue;
    if (p.out16 && out_bytes >= 0xfffffff0ull) return hipErrorInvalidValue;
    k.in_bytes = (unsigned)in_bytes; k.w_bytes = (unsigned)w_bytes;
    const long long nblk = (long long)k.tiles_x * k.tiles_y * k.n_mtiles;
    if (nblk <= 0 || nblk > 0x7fffffffLL) return hipErrorInvalidValue;
    if (n_blocks) *n_blocks = (int)nblk;
    const bool aligned = p.W % 4 == 0 && (reinterpret_cast<uintptr_t>(p.in) & 15) == 0;
    (aligned ? k_quad : k_word)<<<dim3((unsigned)nblk), dim3(NTHREADS), 0, s>>>(k);
    return hipGetLastError();
}

int conv_num_configs() { return 8; }

const char* conv_config_name(int cfg)
{
    static const char* names[] = {"128x128px cc4", "128x256px cc4", "64x256px cc4", "64x128px cc4", "64x256px cc8", "128x128px cc8",
                                  "STAMPED 128x128px cc4", "64x64px cc4"};
    return cfg >= 0 && cfg < 8 ? names[cfg] : "?";
}

// Tile choice by a small occupancy model.  A launch is a number of equal blocks; each CU receives
// n = ceil(blocks / 256) of them and runs up to `occ` at a time.  The matrix pipe is shared, so a CU's
// time is (work of the blocks it runs) / (pipe efficiency at that many co-resident waves per SIMD):
//   eff(1) .. eff(4+) measured on MI355X with this kernel (sweeps under profiles/): one wave per SIMD
//   cannot hide the chunk-boundary bubbles, three or four nearly can.
// Full rounds of `occ` blocks run at eff(occ); the ragged tail of m < occ blocks runs at eff(m).
int conv_pick_config(const ConvProblem& p)
{
    struct Cand { int cfg, bm, rows, occ; };
    static const Cand cands[] = {{1, 128, 8, 2}, {0, 128, 4, 3}, {2, 64, 8, 4}, {3, 64, 4, 5}, {7, 64, 2, 6}};
    static const double eff[7] = {0.0, 0.66, 0.88, 0.93, 0.95, 0.96, 0.96};
    const long long tx = (p.W + 31) / 32;
    int best = 3;
    double best_t = 1e300;
    for (const Cand& c : cands) {
        if (p.MPad % c.bm != 0) continue;
        const long long blocks = tx * ((p.H + c.rows - 1) / c.rows) * (p.MPad / c.bm);
        const double work = (double)c.bm * c.rows;                 // per-block MFMA work (x K, common)
        const long long n = (blocks + 255) / 256;
        const long long rounds = n / c.occ, tail = n % c.occ;
        double t = rounds * c.occ * work / eff[c.occ];
        if (tail) t += tail * work / eff[tail];
        t *= 1.0 + 0.02 * (c.bm == 64) + 0.04 * (c.rows == 2);       // small tiles re-read activations / weights more often
        if (t < best_t) { best_t = t; best = c.cfg; }
    }
    return best;
}

hipError_t launch_conv3x3_cfg(const ConvProblem& p, int cfg, hipStream_t s)
{
    if (p.MPad % kCoutQuantum != 0 || p.MPad < p.M) return hipErrorInvalidValue;
    if (cfg < 0) cfg = conv_pick_config(p);
    if ((cfg == 0 || cfg == 1 || cfg == 5 || cfg == 6) && p.MPad % 128 != 0) return hipErrorInvalidValue;
#define ST2_RUN(NAME, BM, ROWS, CCK) return run(p, BM, ROWS, CCK, NAME##_q, NAME##_w, s)
    switch (cfg) {
    case 0: ST2_RUN(conv3x3_mfma_f32_128x128_cc4, 128, 4, 4);
    case 1: ST2_RUN(conv3x3_mfma_f32_128x256_cc4, 128, 8, 4);
    case 2: ST2_RUN(conv3x3_mfma_f32_64x256_cc4, 64, 8, 4);
    case 3: ST2_RUN(conv3x3_mfma_f32_64x128_cc4, 64, 4, 4);
    case 4: ST2_RUN(conv3x3_mfma_f32_64x256_cc8, 64, 8, 8);
    case 5: ST2_RUN(conv3x3_mfma_f32_128x128_cc8, 128, 4, 8);
    case 6: ST2_RUN(conv3x3_mfma_f32_128x128_cc4_stamped, 128, 4, 4);
    case 7: ST2_RUN(conv3x3_mfma_f32_64x64_cc4, 64, 2, 4);
    }
#undef ST2_RUN
    return hipErrorInvalidValue;
}

hipError_t launch_conv3x3(const ConvProblem& p, hipStream_t s)
{
    // ST2_CONV_CFG=<id> forces one tile configuration (debugging / tests of every configuration)
    static const int forced = [] { const char* e = getenv("ST2_CONV_CFG"); return e ? atoi(e) : -1; }();
    if (forced >= 0 && !((forced == 0 || forced == 1 || forced == 5 || forced == 6) && p.MPad % 128 != 0))
        return launch_conv3x3_cfg(p, forced, s);
    return launch_conv3x3_cfg(p, -1, s);
}

// Style gradient S = c2 * (D @ F) on the conv pipeline (TAPS = 1).  Dp is D laid out [C][MPad].
// Tile of the style-gradient launch: 128 channels x 128 pixels for C > 64 -- unless that leaves fewer than one workgroup per CU
// (conv5_1 at 1024^2: 64 x 64 pixels, 4 channel tiles x 32 pixel tiles = 128 workgroups on 256 CUs), then 64 channels x 128 pixels;
// 64 x 256 pixels for C <= 64 (one channel tile).  ST2_STYLE_SMALL=0 keeps the large tile everywhere (read per launch: A/B runs).
constexpr int SB_M = 64, SB_N = 128, SB_K = 32, SB_LDA = SB_K + 1, SB_LDB = SB_N + 4, SB_GRID = 2048;      // style_grad_big_k (below)
static bool style_big(int C, int H, int W);
static void style_tile(int C, int H, int W, int* bm, int* rows)
{
    if (C <= 64) { *bm = 64; *rows = 8; return; }
    *bm = 128; *rows = 4;
    const long long blocks = (long long)((W + 31) / 32) * ((H + 3) / 4) * (conv_mpad(C) / 128);
    const char* e = getenv("ST2_STYLE_SMALL");
    if (blocks < 256 && conv_mpad(C) % 64 == 0 && !(e && *e == '0')) *bm = 64;
}

int style_grad_blocks(int C, int H, int W)
{
    if (style_big(C, H, W)) return SB_GRID;                 // style_grad_big_k: one partial sum per workgroup of its fixed grid
    int bm, rows;
    style_tile(C, H, W, &bm, &rows);
    return ((W + 31) / 32) * ((H + rows - 1) / rows) * (conv_mpad(C) / bm);
}

// The style gradient for a blob of 4 GiB or more (conv1_1 .. conv3_1 of an 8192 x 8192 image in ONE engine): the kernels above stage F
// through a buffer resource (32-bit byte offsets).  This one addresses with 64-bit pointers: register-staged tiles of 64 channels x 128
// pixels, 32 channels of K per step, a grid-stride loop over the tiles (one partial sum per workgroup).  Same contract as
// launch_style_grad (whole blob, hw % 4 == 0); slower per FLOP than the LDS-DMA kernels -- it exists so that the large image runs at all.
__global__ __launch_bounds__(256) void style_grad_big_k(const float* __restrict__ Dp, int ld, const float* __restrict__ F, float* __restrict__ dst,
                                                        float c2, int fused, float sw, const float* __restrict__ norm, int accumulate,
                                                        float* __restrict__ partial, int C, size_t hw)
{
    __shared__ float As[SB_M * SB_LDA];
    __shared__ float Bs[SB_K * SB_LDB];
    __shared__ float red[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1, l31 = lane & 31, khalf = lane >> 5;
    const size_t n_ptiles = (hw + SB_N - 1) / SB_N;
    const size_t n_tiles = n_ptiles * ((C + SB_M - 1) / SB_M);
    const float coef = fused ? sw / *norm : 0.f;
    float ss = 0.f;
    for (size_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int m0 = (int)(tile / n_ptiles) * SB_M;
        const size_t p0 = (tile % n_ptiles) * SB_N;
        f32x16 acc[2];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
        for (int k0 = 0; k0 < C; k0 += SB_K) {
            __syncthreads();                     // the previous step's operands are consumed
#pragma unroll
            for (int i = 0; i < SB_M * SB_K / 256; ++i) {            // D tile: As[m][k]
                const int e = tid + i * 256, m = e / SB_K, k = e - m * SB_K;
                As[m * SB_LDA + k] = (m0 + m < C && k0 + k < C) ? Dp[(size_t)(m0 + m) * ld + k0 + k] : 0.f;
            }
#pragma unroll
            for (int i = 0; i < SB_K * SB_N / 4 / 256; ++i) {        // F tile: Bs[k][p], 16 bytes per load (hw % 4 == 0)
                const int q = tid + i * 256, k = q / (SB_N / 4), pq = q - k * (SB_N / 4);
                const size_t p = p0 + 4 * (size_t)pq;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (k0 + k < C && p < hw) v = *reinterpret_cast<const float4*>(F + (size_t)(k0 + k) * hw + p);
                *reinterpret_cast<float4*>(Bs + k * SB_LDB + 4 * pq) = v;
            }
            __syncthreads();
#pragma unroll
            for (int ks = 0; ks < SB_K / 2; ++ks) {
                const float av = As[(wm * 32 + l31) * SB_LDA + 2 * ks + khalf];
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const float bv = Bs[(2 * ks + khalf) * SB_LDB + wn * 64 + j * 32 + l31];
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[j], 0, 0, 0);
                }
            }
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const size_t p = p0 + wn * 64 + j * 32 + l31;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = m0 + wm * 32 + (e & 3) + 8 * (e >> 2) + 4 * khalf;
                if (m >= C || p >= hw) continue;
                const float v = acc[j][e] * c2;
                ss += v * v;
                const size_t o = (size_t)m * hw + p;
                dst[o] = fused ? coef * v + (accumulate ? dst[o] : 0.f) : v;
            }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) ss += __shfl_down(ss, o, 64);
    if (lane == 0) red[wave] = ss;
    __syncthreads();
    if (tid == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
static bool style_big(int C, int H, int W)
{
    const char* fb = getenv("ST2_STYLE_FORCE_BIG");         // test hook: =1 runs the 64-bit-addressed kernel on blobs of any size
    return 4ull * C * H * W >= 0xfffffff0ull || (fb && *fb == '1' && ((size_t)H * W) % 4 == 0);
}

hipError_t launch_style_grad(const float* Dp, const float* F, float* dst, float c2, int fused, float sw, const float* norm,
                             int accumulate, float* partial, int* n_partial, int C, int H, int W, hipStream_t s,
                             const PixRoi* roi)
{
    ConvProblem p{};
    p.in = F; p.wpack = Dp; p.out = dst; p.K = C; p.M = C; p.MPad = conv_mpad(C); p.H = H; p.W = W;
    ConvKArgs st{};
    st.c2 = c2; st.sw = sw; st.norm = norm; st.fused = fused; st.accumulate = accumulate; st.partial = partial;
    st.ry0 = roi ? roi->y0 : 0; st.rx0 = roi ? roi->x0 : 0; st.ry1 = roi ? roi->y1 : H; st.rx1 = roi ? roi->x1 : W;
    if (style_big(C, H, W)) {
        const size_t hw = (size_t)H * W;
        if (roi || hw % 4 != 0 || (reinterpret_cast<uintptr_t>(F) & 15) != 0) return hipErrorInvalidValue;
        style_grad_big_k<<<SB_GRID, 256, 0, s>>>(Dp, conv_mpad(C), F, dst, c2, fused, sw, norm, accumulate, partial, C, hw);
        if (n_partial) *n_partial = SB_GRID;
        return hipGetLastError();
    }
    int bm, rows;
    style_tile(C, H, W, &bm, &rows);
    if (bm == 128) return run(p, 128, 4, 32, style_grad_mfma_f32_128x128_cc32_q, style_grad_mfma_f32_128x128_cc32_w, s, &st, n_partial);
    if (rows == 4) return run(p, 64, 4, 32, style_grad_mfma_f32_64x128_cc32_q, style_grad_mfma_f32_64x128_cc32_w, s, &st, n_partial);
    return run(p, 64, 8, 16, style_grad_mfma_f32_64x256_cc16_q, style_grad_mfma_f32_64x256_cc16_w, s, &st, n_partial);
}

// ------------------------------------------------------------------------------------------
// dgrad with a tiny output-channel count (conv1_1 -> image, M = 3): not GEMM shaped, so a direct
// VALU kernel.  One thread per 4 consecutive output pixels, all M (<= 4) channels; weights via scalar loads.
// dx[m][y][x] = sum_{co,ky,kx} w[co][m][ky][kx] * dy[co][y-ky+1][x-kx+1]  (+ inject)
// ------------------------------------------------------------------------------------------
constexpr int SM_MAXM = 4;
constexpr int SM_PX = 4;                              // consecutive pixels per thread (register blocking along x)
constexpr int SM_TX = 32 * SM_PX, SM_TY = 8, SM_CH = 4;

// Weights are wave-uniform: they are read with scalar loads straight from the original
// (Cout, M, 3, 3) layout (27 consecutive floats per input channel for M = 3), so the LDS only
// serves the activation tile.  A thread owns 4 consecutive pixels of a row: per channel and tile row it reads
// 6 floats for 4 pixels x 3 taps (18 ds_read per 4 pixels instead of 36), then 9*M v_fmac (SGPR operand) per pixel.
template <int M>
__global__ __launch_bounds__(256) void conv3x3_dgrad_smallM(const float* __restrict__ dy, const float* __restrict__ w,
                                                            float* __restrict__ dx, const float* __restrict__ inject,
                                                            int Cout, int H, int W)
{
    constexpr int TW = SM_TX + 2, TH = SM_TY + 2;
    __shared__ float t_s[2][SM_CH * TH * TW];
    const int tid = threadIdx.x;
    const int x0 = blockIdx.x * SM_TX, y0 = blockIdx.y * SM_TY;
    const int lx = (tid & 31) * SM_PX, ly = tid >> 5;
    const size_t plane = (size_t)H * W;
    float acc[SM_PX][M];
#pragma unroll
    for (int q = 0; q < SM_PX; ++q)
#pragma unroll
        for (int m = 0; m < M; ++m) acc[q][m] = 0.f;

    constexpr int NE = SM_CH * TH * TW, PER_T = (NE + 255) / 256;
    int toff[PER_T];                          // tile element -> offset inside one channel plane (or -1)
#pragma unroll
    for (int i = 0; i < PER_T; ++i) {
        const int e = tid + i * 256;
        const int rem = e % (TH * TW), rr = rem / TW, col = rem % TW;
        const int gy = y0 - 1 + rr, gx = x0 - 1 + col;
        toff[i] = (e < NE && gy >= 0 && gy < H && gx >= 0 && gx < W) ? gy * W + gx : -1;
    }
    float stage[PER_T];
    auto load = [&](int c0) {
#pragma unroll
        for (int i = 0; i < PER_T; ++i) {
            const int c = (tid + i * 256) / (TH * TW);
            stage[i] = (toff[i] >= 0 && c0 + c < Cout) ? dy[(size_t)(c0 + c) * plane + toff[i]] : 0.f;
        }
    };
    auto store = [&](int buf) {
#pragma unroll
        for (int i = 0; i < PER_T; ++i)
            if (tid + i * 256 < NE) t_s[buf][tid + i * 256] = stage[i];
    };

    load(0);
    store(0);
    __syncthreads();
    int buf = 0;
    for (int c0 = 0; c0 < Cout; c0 += SM_CH) {
        const bool more = c0 + SM_CH < Cout;
        if (more) load(c0 + SM_CH);              // in flight while this chunk is consumed
        const float* tile = t_s[buf];
#pragma unroll
        for (int c = 0; c < SM_CH; ++c) {
            if (c0 + c < Cout) {
                const float* wc = w + (size_t)(c0 + c) * M * 9;      // uniform -> s_load
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) {
                    // source pixel (y - ky + 1, x - kx + 1) -> tile coords (ly + 2 - ky, lx + q + 2 - kx)
                    float g[SM_PX + 2];
#pragma unroll
                    for (int j = 0; j < SM_PX + 2; ++j) g[j] = tile[(c * TH + (ly + 2 - ky)) * TW + lx + j];
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                        for (int q = 0; q < SM_PX; ++q)
#pragma unroll
                            for (int m = 0; m < M; ++m) acc[q][m] += wc[m * 9 + ky * 3 + kx] * g[q + 2 - kx];
                }
            }
        }
        if (more) {
            store(buf ^ 1);
            __syncthreads();
            buf ^= 1;
        }
    }
    const int gy = y0 + ly;
    if (gy < H)
#pragma unroll
        for (int q = 0; q < SM_PX; ++q) {
            const int gx = x0 + lx + q;
            if (gx >= W) continue;
#pragma unroll
            for (int m = 0; m < M; ++m) {
                const size_t idx = (size_t)m * plane + (size_t)gy * W + gx;
                dx[idx] = acc[q][m] + (inject ? inject[idx] : 0.f);
            }
        }
}

// bf16 feature path: the same data gradient with dy read from the bf16 channel-blocked copy [Cout/8][H][W][8] the bf16
// dgrad above wrote (no fp32 diff exists then); w holds bf16-representable values (rounded on the host); products and sums
// stay fp32.  A chunk is one channel block: every thread fetches whole 16-byte quads (8 channels of one pixel) into
// registers while the previous chunk is consumed from LDS, then unpacks them into the fp32 tile [8][TH][TW].
template <int M>
__global__ __launch_bounds__(256) void conv3x3_dgrad_smallM16(const uint4* __restrict__ dy16, const float* __restrict__ w,
                                                              float* __restrict__ dx, const float* __restrict__ inject,
                                                              int Cout, int H, int W)
{
    constexpr int TW = SM_TX + 2, TH = SM_TY + 2, CH = 8;
    constexpr int NPIX = TH * TW, PER_T = (NPIX + 255) / 256;
    __shared__ float t_s[CH * NPIX];
    const int tid = threadIdx.x;
    const int x0 = blockIdx.x * SM_TX, y0 = blockIdx.y * SM_TY;
    const int lx = (tid & 31) * SM_PX, ly = tid >> 5;
    const size_t plane = (size_t)H * W;
    float acc[SM_PX][M];
#pragma unroll
    for (int q = 0; q < SM_PX; ++q)
#pragma unroll
        for (int m = 0; m < M; ++m) acc[q][m] = 0.f;
    int toff[PER_T];                          // halo-tile pixel -> offset inside one channel-block plane (or -1)
#pragma unroll
    for (int i = 0; i < PER_T; ++i) {
        const int e = tid + i * 256;
        const int rr = e / TW, col = e % TW;
        const int gy = y0 - 1 + rr, gx = x0 - 1 + col;
        toff[i] = (e < NPIX && gy >= 0 && gy < H && gx >= 0 && gx < W) ? gy * W + gx : -1;
    }
    uint4 stage[PER_T];
    auto load = [&](int cb) {
#pragma unroll
        for (int i = 0; i < PER_T; ++i) stage[i] = toff[i] >= 0 ? dy16[(size_t)cb * plane + toff[i]] : make_uint4(0, 0, 0, 0);
    };
    auto store = [&]() {
#pragma unroll
        for (int i = 0; i < PER_T; ++i) {
            const int e = tid + i * 256;
            if (e < NPIX) {
                const unsigned wv[4] = {stage[i].x, stage[i].y, stage[i].z, stage[i].w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    t_s[(2 * j) * NPIX + e] = __builtin_bit_cast(float, wv[j] << 16);
                    t_s[(2 * j + 1) * NPIX + e] = __builtin_bit_cast(float, wv[j] & 0xffff0000u);
                }
            }
        }
    };
    const int ncb = Cout / CH;
    load(0);
    for (int cb = 0; cb < ncb; ++cb) {
        if (cb) __syncthreads();                 // the previous chunk is consumed
        store();
        __syncthreads();
        if (cb + 1 < ncb) load(cb + 1);          // in flight while this chunk is consumed
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const float* wc = w + (size_t)(cb * CH + c) * M * 9;      // uniform -> s_load
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                float g[SM_PX + 2];
#pragma unroll
                for (int j = 0; j < SM_PX + 2; ++j) g[j] = t_s[c * NPIX + (ly + 2 - ky) * TW + lx + j];
#pragma unroll
                for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                    for (int q = 0; q < SM_PX; ++q)
#pragma unroll
                        for (int m = 0; m < M; ++m) acc[q][m] += wc[m * 9 + ky * 3 + kx] * g[q + 2 - kx];
            }
        }
    }
    const int gy = y0 + ly;
    if (gy < H)
#pragma unroll
        for (int q = 0; q < SM_PX; ++q) {
            const int gx = x0 + lx + q;
            if (gx >= W) continue;
#pragma unroll
            for (int m = 0; m < M; ++m) {
                const size_t idx = (size_t)m * plane + (size_t)gy * W + gx;
                dx[idx] = acc[q][m] + (inject ? inject[idx] : 0.f);
            }
        }
}

bool conv_dgrad_smallM_ok(int Cout, int Cin) { (void)Cout; return Cin >= 1 && Cin <= SM_MAXM; }

hipError_t launch_conv3x3_dgrad_smallM16(const unsigned short* dy16, const float* w_rounded, float* dx, const float* inject,
                                         int Cout, int Cin, int H, int W, hipStream_t s)
{
    if (!conv_dgrad_smallM_ok(Cout, Cin) || Cout % 8 != 0) return hipErrorInvalidValue;
    if (conv_dgrad_first_ok(Cout, Cin, H, W, true)) return launch_conv3x3_dgrad_first16(dy16, w_rounded, dx, inject, Cout, Cin, H, W, s);   // matrix cores
    dim3 grid((W + SM_TX - 1) / SM_TX, (H + SM_TY - 1) / SM_TY);
    const uint4* q16 = reinterpret_cast<const uint4*>(dy16);
    switch (Cin) {
    case 1: conv3x3_dgrad_smallM16<1><<<grid, dim3(256), 0, s>>>(q16, w_rounded, dx, inject, Cout, H, W); break;
    case 2: conv3x3_dgrad_smallM16<2><<<grid, dim3(256), 0, s>>>(q16, w_rounded, dx, inject, Cout, H, W); break;
    case 3: conv3x3_dgrad_smallM16<3><<<grid, dim3(256), 0, s>>>(q16, w_rounded, dx, inject, Cout, H, W); break;
    default: conv3x3_dgrad_smallM16<4><<<grid, dim3(256), 0, s>>>(q16, w_rounded, dx, inject, Cout, H, W); break;
    }
    return hipGetLastError();
}

// The same sum with the activation tile staged by LDS-DMA (round 4).  The kernel above fetches its 4-channel tile element by element
// (21 four-byte global loads per thread and chunk, an index division each) into registers, then stores them to LDS between two
// barriers: at 1024^2 it read the 268 MB diff at 2.2 TB/s (124 us), bound by those loads, not by its 3.6 GFLOP.  Here a chunk's tile is
// 4 channels x 10 rows x 34 ALIGNED quads (columns x0 - 4 .. x0 + 131), copied global -> LDS as 16-byte pieces of 64 lanes
// (buffer_load ... lds; quads outside the image get an out-of-range offset = hardware zero fill = the padding) through a ring of three
// buffers -- two chunks in flight while one is consumed, a counted vmcnt and one barrier per chunk; a thread reads its 6 columns of a
// row as b32 + b128 + b32.  Same sum in the same order as the kernel above (the compiler contracts the two differently: equal to
// an ulp, tests/test_gpu_parity.py).  Measured at 1024^2: 124 -> 100 us (double-buffered: 105); a v_pk_fma_f32 form of the inner
// product (two pixels per instruction) was slower (112 us) and is not kept: the bound is neither the loads nor the multiply-adds.  Needs W % 4 == 0, a 16-byte aligned diff < 4 GiB.
constexpr int SD_QROW = SM_TX / 4 + 2;                  // 34 quads per staged row
constexpr int SD_ROW = 4 * SD_QROW;                     // 136 floats
constexpr int SD_TH = SM_TY + 2;                        // 10 rows
constexpr int SD_NQ = SM_CH * SD_TH * SD_QROW;          // 1360 quads per chunk
constexpr int SD_PIECES = (SD_NQ + 63) / 64;            // 22 wave-DMAs
constexpr int SD_PPW = (SD_PIECES + 3) / 4;             // per wave
constexpr int SD_BUF = SD_PIECES * 256;                 // floats per buffer
constexpr int SD_NBUF = 3;                              // ring: two chunks in flight while one is consumed (67.5 KiB: two workgroups per CU)

template <int M>
__global__ __launch_bounds__(256) void conv3x3_dgrad_smallM_dma(const float* __restrict__ dy, unsigned dy_bytes, const float* __restrict__ w,
                                                                float* __restrict__ dx, const float* __restrict__ inject,
                                                                int Cout, int H, int W)
{
    typedef __attribute__((address_space(3))) void* sd_lptr_t;
    __shared__ __attribute__((aligned(16))) float t_s[SD_NBUF][SD_BUF];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int x0 = blockIdx.x * SM_TX, y0 = blockIdx.y * SM_TY;
    const int lx = (tid & 31) * SM_PX, ly = tid >> 5;
    const unsigned plane = (unsigned)H * (unsigned)W;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)dy, 0, dy_bytes, 0x00020000);
    unsigned qoff[SD_PPW];                   // byte offset of this lane's quad inside chunk 0 (channel c of the chunk included), or out of range
    int qch[SD_PPW];
#pragma unroll
    for (int t = 0; t < SD_PPW; ++t) {
        const int qi = (wave + 4 * t) * 64 + lane;
        const int c = qi / (SD_TH * SD_QROW), rem = qi - c * (SD_TH * SD_QROW), rr = rem / SD_QROW, qc = rem - rr * SD_QROW;
        const int gy = y0 - 1 + rr, gx = x0 - 4 + 4 * qc;
        const bool ok = wave + 4 * t < SD_PIECES && qi < SD_NQ && gy >= 0 && gy < H && gx >= 0 && gx + 3 < W;
        qoff[t] = ok ? ((unsigned)c * plane + (unsigned)gy * W + gx) * 4u : 0xffffffffu;
        qch[t] = c;
    }
    auto dma = [&](int c0, int buf) {
        const unsigned coff = (unsigned)c0 * plane * 4u;
#pragma unroll
        for (int t = 0; t < SD_PPW; ++t) {
            if (wave + 4 * t >= SD_PIECES) continue;                   // wave-uniform
            const unsigned vo = (qoff[t] == 0xffffffffu || c0 + qch[t] >= Cout) ? 0xffffffffu : qoff[t] + coff;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (sd_lptr_t)(t_s[buf] + (wave + 4 * t) * 256), 16, vo, 0, 0, 0);
        }
    };
    float acc[SM_PX][M];
#pragma unroll
    for (int q = 0; q < SM_PX; ++q)
#pragma unroll
        for (int m = 0; m < M; ++m) acc[q][m] = 0.f;

    dma(0, 0);
    if (SM_CH < Cout) dma(SM_CH, 1);
    int buf = 0;
    for (int c0 = 0; c0 < Cout; c0 += SM_CH) {
        // this chunk has landed (the NEXT chunk's pieces of this wave -- 6 for waves 0 and 1, 5 for waves 2 and 3 -- may still be in flight)
        if (c0 + SM_CH < Cout) {
            if (wave < (SD_PIECES & 3)) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(SD_PPW) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" :: "n"(SD_PPW - 1) : "memory");
        } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                         // ... for every wave; the buffer consumed in the previous iteration is free again
        if (c0 + 2 * SM_CH < Cout) dma(c0 + 2 * SM_CH, buf >= 1 ? buf - 1 : SD_NBUF - 1);
        const float* tile = t_s[buf];
#pragma unroll
        for (int c = 0; c < SM_CH; ++c) {
            if (c0 + c < Cout) {
                const float* wc = w + (size_t)(c0 + c) * M * 9;      // uniform -> s_load
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) {
                    // source pixel (y - ky + 1, x - kx + 1) -> staged row ly + 2 - ky, staged column lx + q + 5 - kx (column 0 = pixel x0 - 4)
                    const float* row = tile + (c * SD_TH + (ly + 2 - ky)) * SD_ROW + lx;
                    float g[SM_PX + 2];
                    const float4 mid = *reinterpret_cast<const float4*>(row + 4);
                    g[0] = row[3]; g[1] = mid.x; g[2] = mid.y; g[3] = mid.z; g[4] = mid.w; g[5] = row[8];
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                        for (int q = 0; q < SM_PX; ++q)
#pragma unroll
                            for (int m = 0; m < M; ++m) acc[q][m] += wc[m * 9 + ky * 3 + kx] * g[q + 2 - kx];
                }
            }
        }
        buf = buf + 1 == SD_NBUF ? 0 : buf + 1;
    }
    const int gy = y0 + ly, gx = x0 + lx;
    if (gy < H && gx < W) {                      // W % 4 == 0: the thread's four pixels are inside together
#pragma unroll
        for (int m = 0; m < M; ++m) {
            const size_t idx = (size_t)m * plane + (size_t)gy * W + gx;
            float4 o = make_float4(acc[0][m], acc[1][m], acc[2][m], acc[3][m]);
            if (inject) { const float4 ij = *reinterpret_cast<const float4*>(inject + idx); o.x += ij.x; o.y += ij.y; o.z += ij.z; o.w += ij.w; }
            *reinterpret_cast<float4*>(dx + idx) = o;
        }
    }
}

hipError_t launch_conv3x3_dgrad_smallM(const float* dy, const float* w, float* dx, const float* inject,
                                       int Cout, int Cin, int H, int W, hipStream_t s)
{
    if (!conv_dgrad_smallM_ok(Cout, Cin)) return hipErrorInvalidValue;
    {   // ST2_DGRAD_FIRST selects among the older kernels when it is set; unset, the strip walker takes the layer it is built for
        const char* e = getenv("ST2_DGRAD_FIRST");
        if (!(e && *e) && conv_dgrad_first_strip_ok(Cout, Cin, H, W)) return launch_conv3x3_dgrad_first_strip(dy, w, dx, inject, Cout, Cin, H, W, s);
    }
    if (conv_dgrad_first_quad_ok(Cout, Cin, H, W, dy)) return launch_conv3x3_dgrad_first_quad(dy, w, dx, inject, Cout, Cin, H, W, s);        // matrix cores, quads
    if (conv_dgrad_first_ok(Cout, Cin, H, W, false)) return launch_conv3x3_dgrad_first(dy, w, dx, inject, Cout, Cin, H, W, s);              // matrix cores
    dim3 grid((W + SM_TX - 1) / SM_TX, (H + SM_TY - 1) / SM_TY);
    {   // LDS-DMA staging where the layout allows it (ST2_DGRAD_SMALLM_DMA=0: the register-staged kernel; read per launch: A/B runs)
        const char* e = getenv("ST2_DGRAD_SMALLM_DMA");
        const unsigned long long bytes = 4ull * Cout * H * W;
        if (!(e && *e == '0') && W % 4 == 0 && bytes < 0xfffffff0ull && (reinterpret_cast<uintptr_t>(dy) & 15) == 0 &&
            (reinterpret_cast<uintptr_t>(dx) & 15) == 0 && (!inject || (reinterpret_cast<uintptr_t>(inject) & 15) == 0)) {
            switch (Cin) {
            case 1: conv3x3_dgrad_smallM_dma<1><<<grid, dim3(256), 0, s>>>(dy, (unsigned)bytes, w, dx, inject, Cout, H, W); break;
            case 2: conv3x3_dgrad_smallM_dma<2><<<grid, dim3(256), 0, s>>>(dy, (unsigned)bytes, w, dx, inject, Cout, H, W); break;
            case 3: conv3x3_dgrad_smallM_dma<3><<<grid, dim3(256), 0, s>>>(dy, (unsigned)bytes, w, dx, inject, Cout, H, W); break;
            default: conv3x3_dgrad_smallM_dma<4><<<grid, dim3(256), 0, s>>>(dy, (unsigned)bytes, w, dx, inject, Cout, H, W); break;
            }
            return hipGetLastError();
        }
    }
    switch (Cin) {
    case 1: conv3x3_dgrad_smallM<1><<<grid, dim3(256), 0, s>>>(dy, w, dx, inject, Cout, H, W); break;
    case 2: conv3x3_dgrad_smallM<2><<<grid, dim3(256), 0, s>>>(dy, w, dx, inject, Cout, H, W); break;
    case 3: conv3x3_dgrad_smallM<3><<<grid, dim3(256), 0, s>>>(dy, w, dx, inject, Cout, H, W); break;
    default: conv3x3_dgrad_smallM<4><<<grid, dim3(256), 0, s>>>(dy, w, dx, inject, Cout, H, W); break;
    }
    return hipGetLastError();
}

}  // namespace st2
