// Data gradient of the FIRST conv (conv1_1: 64 -> 3 channels, pycaffe Convolution backward behind worker.py:100-106) on the
// matrix cores.  With M <= 3 output channels the implicit GEMM of the other layers has nothing to tile, and a direct VALU kernel
// (conv3x3_dgrad_smallM*) spends 27 M FMAs per pixel and input channel.  The sum factorises instead:
//     dx[m][y][x] = sum_c sum_tap w[c][m][tap] dy[c][(y, x) - off(tap)]
//                 = sum_tap Z[m, tap][(y, x) - off(tap)],        Z[r][p] = sum_c A[r][c] dy[c][p],  r = 9 m + tap, A[r][c] = w[c][m][tap]
// Z is a 1x1 convolution from Cout channels to 9 M <= 27 (padded to 32) rows -- exactly one 32-row MFMA tile, K = Cout -- and the
// second step is 9 M shifted adds per pixel out of LDS.  One workgroup = 8 x 32 output pixels: it takes Z on the 10 x 34 halo tile
// (11 groups of 32 pixels spread over the 4 waves, operands straight from global memory into the MFMA's B registers: the diff is
// read once, coalesced along x), parks the 27 rows in LDS and adds.  Bound: HBM (the diff of conv1_1's blob: 4 or 2 bytes per
// element, 1.33x for the halo -- served by L2).  Every product and sum is fp32 (bf16 variant: exact bf16 products, fp32 sums).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "st2_kernels.h"

namespace st2 {

typedef float df_f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 df_bf16x8 __attribute__((ext_vector_type(8)));

constexpr int DF_TH = 8, DF_TW = 32;                     // output pixels per workgroup
constexpr int DF_HH = DF_TH + 2, DF_HW = DF_TW + 2;      // halo tile
constexpr int DF_NPIX = DF_HH * DF_HW;                   // 340
constexpr int DF_NG = (DF_NPIX + 31) / 32;               // 11 groups of 32 halo pixels
constexpr int DF_ZS = DF_NG * 32 + 1;                    // padded row stride of Z in LDS
constexpr int DF_ROWS = 27;

// second step, shared by both variants: thread = one output pixel, 9 M shifted reads of Z
template <int M>
__device__ __forceinline__ void dgrad_first_gather(const float* z_s, float* __restrict__ dx, const float* __restrict__ inject,
                                                   int x0, int y0, int H, int W)
{
    const int tid = threadIdx.x, px = tid & 31, py = tid >> 5;
    const int gx = x0 + px, gy = y0 + py;
    if (gx >= W || gy >= H) return;
    const size_t plane = (size_t)H * W;
#pragma unroll
    for (int m = 0; m < M; ++m) {
        float acc = 0.f;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx)      // source pixel (y - ky + 1, x - kx + 1) = halo (py + 2 - ky, px + 2 - kx)
                acc += z_s[(m * 9 + ky * 3 + kx) * DF_ZS + (py + 2 - ky) * DF_HW + (px + 2 - kx)];
        const size_t idx = (size_t)m * plane + (size_t)gy * W + gx;
        dx[idx] = acc + (inject ? inject[idx] : 0.f);
    }
}

// rows 0 .. 26 of one 32 x 32 accumulator tile (C/D map: column = lane & 31, row = (e & 3) + 8 (e >> 2) + 4 (lane >> 5)) -> LDS
__device__ __forceinline__ void dgrad_first_park(float* z_s, const df_f32x16& acc, int group, int lane)
{
    const int l31 = lane & 31, khalf = lane >> 5;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int row = (e & 3) + 8 * (e >> 2) + 4 * khalf;
        if (row < DF_ROWS) z_s[row * DF_ZS + group * 32 + l31] = acc[e];
    }
}

// fp32: dy [Cout][H][W], w (Cout, M, 3, 3); v_mfma_f32_32x32x2_f32, one k-pair = two channels
template <int M>
__global__ __launch_bounds__(256) void conv3x3_dgrad_first_f32(const float* __restrict__ dy, const float* __restrict__ w,
                                                               float* __restrict__ dx, const float* __restrict__ inject,
                                                               int Cout, int H, int W)
{
    __shared__ float z_s[DF_ROWS * DF_ZS];
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, khalf = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int x0 = blockIdx.x * DF_TW, y0 = blockIdx.y * DF_TH;
    const size_t plane = (size_t)H * W;
    constexpr int KC = 32;                                   // k-pairs per chunk (64 channels)
    // this wave's groups: wave, wave + 4, wave + 8
    df_f32x16 acc[3];
    int off[3];
#pragma unroll
    for (int t = 0; t < 3; ++t) {
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
        const int hp = (wave + 4 * t) * 32 + l31;
        const int hy = hp / DF_HW, hx = hp - hy * DF_HW;
        const int gy = y0 - 1 + hy, gx = x0 - 1 + hx;
        off[t] = (wave + 4 * t < DF_NG && hp < DF_NPIX && gy >= 0 && gy < H && gx >= 0 && gx < W) ? gy * W + gx : -1;
    }
    const int r = l31, m_r = r / 9, tap_r = r - 9 * m_r;     // this lane's row of A
    for (int c0 = 0; c0 < Cout; c0 += 2 * KC) {
        // every B operand of this pass first (96 loads in flight per lane): their latency covers the staging of A
        float b[3][KC];
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int kp = 0; kp < KC; ++kp) {
                const int c = c0 + 2 * kp + khalf;
                b[t][kp] = (off[t] >= 0 && c < Cout) ? dy[(size_t)c * plane + off[t]] : 0.f;
            }
        // A: the weights of these 64 channels, coalesced into LDS (the Z rows are not there yet), then one gather per lane
        if (c0) __syncthreads();
        const int nw = min(2 * KC, Cout - c0) * M * 9;
        for (int i = tid; i < nw; i += 256) z_s[i] = w[(size_t)c0 * M * 9 + i];
        __syncthreads();
        float a[KC];
#pragma unroll
        for (int kp = 0; kp < KC; ++kp) {
            const int cl = 2 * kp + khalf;
            a[kp] = (r < 9 * M && c0 + cl < Cout) ? z_s[(cl * M + m_r) * 9 + tap_r] : 0.f;
        }
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            if (wave + 4 * t >= DF_NG) continue;             // wave-uniform
#pragma unroll
            for (int kp = 0; kp < KC; ++kp) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kp], b[t][kp], acc[t], 0, 0, 0);
        }
    }
    __syncthreads();                                         // every wave has taken its A operands out of LDS
#pragma unroll
    for (int t = 0; t < 3; ++t)
        if (wave + 4 * t < DF_NG) dgrad_first_park(z_s, acc[t], wave + 4 * t, lane);
    __syncthreads();
    dgrad_first_gather<M>(z_s, dx, inject, x0, y0, H, W);
}

// bf16 feature path: dy16 = the channel-blocked bf16 copy [Cout/8][H][W][8] (a quad IS the B fragment), w holds
// bf16-representable values (rounded on the host); v_mfma_f32_32x32x16_bf16, one k-step = 16 channels
template <int M>
__global__ __launch_bounds__(256) void conv3x3_dgrad_first_bf16(const uint4* __restrict__ dy16, const float* __restrict__ w,
                                                                float* __restrict__ dx, const float* __restrict__ inject,
                                                                int Cout, int H, int W)
{
    __shared__ float z_s[DF_ROWS * DF_ZS];
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, khalf = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int x0 = blockIdx.x * DF_TW, y0 = blockIdx.y * DF_TH;
    const size_t plane = (size_t)H * W;
    constexpr int KC = 4;                                    // k-steps per chunk (64 channels)
    df_f32x16 acc[3];
    int off[3];
#pragma unroll
    for (int t = 0; t < 3; ++t) {
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
        const int hp = (wave + 4 * t) * 32 + l31;
        const int hy = hp / DF_HW, hx = hp - hy * DF_HW;
        const int gy = y0 - 1 + hy, gx = x0 - 1 + hx;
        off[t] = (wave + 4 * t < DF_NG && hp < DF_NPIX && gy >= 0 && gy < H && gx >= 0 && gx < W) ? gy * W + gx : -1;
    }
    const int r = l31, m_r = r / 9, tap_r = r - 9 * m_r;
    const int nks = Cout / 16;
    for (int k0 = 0; k0 < nks; k0 += KC) {
        uint4 b[3][KC];                                      // every B operand of this pass first
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int ks = 0; ks < KC; ++ks)
                b[t][ks] = (off[t] >= 0 && k0 + ks < nks) ? dy16[(size_t)(2 * (k0 + ks) + khalf) * plane + off[t]] : make_uint4(0, 0, 0, 0);
        if (k0) __syncthreads();
        const int nw = min(16 * KC, Cout - 16 * k0) * M * 9;
        for (int i = tid; i < nw; i += 256) z_s[i] = w[(size_t)16 * k0 * M * 9 + i];
        __syncthreads();
        df_bf16x8 a[KC];
#pragma unroll
        for (int ks = 0; ks < KC; ++ks)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int cl = 16 * ks + 8 * khalf + j;
                a[ks][j] = (__bf16)((r < 9 * M && k0 + ks < nks) ? z_s[(cl * M + m_r) * 9 + tap_r] : 0.f);     // exact: w is bf16-representable
            }
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            if (wave + 4 * t >= DF_NG) continue;
#pragma unroll
            for (int ks = 0; ks < KC; ++ks)
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ks], __builtin_bit_cast(df_bf16x8, b[t][ks]), acc[t], 0, 0, 0);
        }
    }
    __syncthreads();                                         // every wave has taken its A operands out of LDS
#pragma unroll
    for (int t = 0; t < 3; ++t)
        if (wave + 4 * t < DF_NG) dgrad_first_park(z_s, acc[t], wave + 4 * t, lane);
    __syncthreads();
    dgrad_first_gather<M>(z_s, dx, inject, x0, y0, H, W);
}

// bf16, Cout = 64 (round 4): the same sums in the same order (bit-identical to conv3x3_dgrad_first_bf16), laid out for the memory
// system.  The tile kernel above reads its halo tile (1.33x the diff), computes, stores and exits -- 16384 short-lived workgroups
// at 2048^2, 2.5 TB/s.  Here a workgroup owns a COLUMN STRIP, 126 output pixels wide (128 with the halo: one 32-pixel MFMA column
// group per wave), and walks down it row by row: a row of the diff is read once (no vertical halo: Z rows stay in a four-slot LDS
// ring until the three output rows that need them are done), its 16-byte quads ARE the B fragments (4 per lane and row, the rows
// r + 1 .. r + 4 under way while row r is multiplied), four MFMAs make the row's Z[27][128], one barrier, and 252 threads add the
// 27 shifted terms of output row r - 1.  Per row and workgroup: 16 KB of diff in, 1.5 KB out; the strips x segments grid is sized
// to be resident at once (two workgroups per CU, 128 KB of loads in flight per CU).
constexpr int DS_OW = 126, DS_HW = 128, DS_ZS = DS_HW + 1, DS_SLOTS = 4, DS_DEPTH = 4;

template <int M>
__global__ __launch_bounds__(256, 2) void conv3x3_dgrad_first_bf16_strip(const uint4* __restrict__ dy16, const float* __restrict__ w,
                                                                         float* __restrict__ dx, const float* __restrict__ inject,
                                                                         int H, int W, int seg, unsigned dy_bytes)
{
    __shared__ float z_s[DS_SLOTS][DF_ROWS][DS_ZS];
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, khalf = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int x0 = blockIdx.x * DS_OW, y0 = blockIdx.y * seg;
    const int nq = min(seg, H - y0) + 2;                     // halo rows of this segment: gy = y0 - 1 + q
    const unsigned plane = (unsigned)H * W;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)dy16, 0, dy_bytes, 0x00020000);
    const int gx = x0 - 1 + 32 * wave + l31;
    const bool colok = gx >= 0 && gx < W;
    // A fragments (rows = the 9 M (m, tap) pairs, K = 16 channels per step): registers for the life of the workgroup
    const int r = l31, m_r = r / 9, tap_r = r - 9 * m_r;
    df_bf16x8 a[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int cl = 16 * ks + 8 * khalf + j;
            a[ks][j] = (__bf16)(r < 9 * M ? w[(cl * M + m_r) * 9 + tap_r] : 0.f);          // exact: w is bf16-representable
        }
    auto load_row = [&](int q, uint4 (&b)[4]) {              // (an offset beyond the tensor reads zeros: rows / columns outside the image)
        const int gy = y0 - 1 + q;
        const bool ok = colok && gy >= 0 && gy < H && q < nq;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const unsigned off = ok ? ((unsigned)(2 * ks + khalf) * plane + (unsigned)gy * W + gx) * 16u : 0xfffffff0u;
            b[ks] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0));
        }
    };
    uint4 b[DS_DEPTH][4];
#pragma unroll
    for (int u = 0; u < DS_DEPTH; ++u) load_row(u, b[u]);
    const int col = tid & 127, half = tid >> 7;              // the adds: thread = one output column; half 0 takes m = 0 (and 1 of 3), half 1 the last m
    const int ox = x0 + col;
    const bool out_ok = col < DS_OW && ox < W;
    for (int q0 = 0; q0 < nq; q0 += DS_DEPTH) {
#pragma unroll
        for (int u = 0; u < DS_DEPTH; ++u) {
            const int q = q0 + u;
            if (q >= nq) break;                              // uniform
            df_f32x16 acc;
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ks], __builtin_bit_cast(df_bf16x8, b[u][ks]), acc, 0, 0, 0);
            load_row(q + DS_DEPTH, b[u]);
            float* zq = &z_s[q & (DS_SLOTS - 1)][0][0];
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = (e & 3) + 8 * (e >> 2) + 4 * khalf;
                if (row < DF_ROWS) zq[row * DS_ZS + 32 * wave + l31] = acc[e];
            }
            // one barrier per row: the slot written next (q + 1) was last read for output row q - 3 + 1, two barriers ago
            __syncthreads();
            if (q >= 2 && out_ok) {
                const int oy = y0 + q - 2;
                auto out_m = [&](int m) {
                    float v = 0.f;
#pragma unroll
                    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                        for (int kx = 0; kx < 3; ++kx)      // source pixel (y - ky + 1, x - kx + 1) = halo row q - ky, halo column col + 2 - kx
                            v += z_s[(q - ky) & (DS_SLOTS - 1)][m * 9 + ky * 3 + kx][col + 2 - kx];
                    const size_t idx = (size_t)m * plane + (size_t)oy * W + ox;
                    dx[idx] = v + (inject ? inject[idx] : 0.f);
                };
                if (half == 0) { out_m(0); if (M == 3) out_m(1); }
                else if (M >= 2) out_m(M - 1);
            }
        }
    }
}

// fp32, Cout = 64: the strip walker on v_mfma_f32_32x32x2_f32 (the sums of conv3x3_dgrad_first_f32 in the same order, bit-identical).
// A row of the strip = 32 k-pairs = 32 four-byte loads per lane (a wave's instruction reads 128 contiguous bytes of two channel
// planes), rows r + 1 .. r + DSF_DEPTH in flight; everything else as in the bf16 walker.
constexpr int DSF_DEPTH = 3;

template <int M>
__global__ __launch_bounds__(256, 2) void conv3x3_dgrad_first_f32_strip(const float* __restrict__ dy, const float* __restrict__ w,
                                                                        float* __restrict__ dx, const float* __restrict__ inject,
                                                                        int H, int W, int seg, unsigned dy_bytes)
{
    __shared__ float z_s[DS_SLOTS][DF_ROWS][DS_ZS];
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, khalf = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int x0 = blockIdx.x * DS_OW, y0 = blockIdx.y * seg;
    const int nq = min(seg, H - y0) + 2;                     // halo rows of this segment: gy = y0 - 1 + q
    const unsigned plane = (unsigned)H * W;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)dy, 0, dy_bytes, 0x00020000);
    const int gx = x0 - 1 + 32 * wave + l31;
    const bool colok = gx >= 0 && gx < W;
    const int r = l31, m_r = r / 9, tap_r = r - 9 * m_r;
    float a[32];                                             // A: row r = (m, tap), k-pair kp = channels 2 kp + khalf
#pragma unroll
    for (int kp = 0; kp < 32; ++kp) a[kp] = r < 9 * M ? w[((2 * kp + khalf) * M + m_r) * 9 + tap_r] : 0.f;
    auto load_row = [&](int q, float (&b)[32]) {             // (an offset beyond the tensor reads zeros: rows / columns outside the image)
        const int gy = y0 - 1 + q;
        const bool ok = colok && gy >= 0 && gy < H && q < nq;
        // the k-pair's planes ride in the SCALAR offset, which the hardware's range check leaves out (raw buffers check the vector
        // offset + the immediate): a lane outside the image stays out of range whatever is added, and one register addresses the row
        const unsigned base = ok ? ((unsigned)khalf * plane + (unsigned)gy * W + gx) * 4u : 0xfffffff0u;
#pragma unroll
        for (int kp = 0; kp < 32; ++kp)
            b[kp] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, base, (unsigned)(2 * kp) * plane * 4u, 0));
    };
    float b[DSF_DEPTH][32];
#pragma unroll
    for (int u = 0; u < DSF_DEPTH; ++u) load_row(u, b[u]);
    const int col = tid & 127, half = tid >> 7;
    const int ox = x0 + col;
    const bool out_ok = col < DS_OW && ox < W;
    for (int q0 = 0; q0 < nq; q0 += DSF_DEPTH) {
#pragma unroll
        for (int u = 0; u < DSF_DEPTH; ++u) {
            const int q = q0 + u;
            if (q >= nq) break;                              // uniform
            df_f32x16 acc;
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
            for (int kp = 0; kp < 32; ++kp) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kp], b[u][kp], acc, 0, 0, 0);
            load_row(q + DSF_DEPTH, b[u]);
            float* zq = &z_s[q & (DS_SLOTS - 1)][0][0];
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = (e & 3) + 8 * (e >> 2) + 4 * khalf;
                if (row < DF_ROWS) zq[row * DS_ZS + 32 * wave + l31] = acc[e];
            }
            __syncthreads();                                 // (one per row: see the bf16 walker)
            if (q >= 2 && out_ok) {
                const int oy = y0 + q - 2;
                auto out_m = [&](int m) {
                    float v = 0.f;
#pragma unroll
                    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                        for (int kx = 0; kx < 3; ++kx)
                            v += z_s[(q - ky) & (DS_SLOTS - 1)][m * 9 + ky * 3 + kx][col + 2 - kx];
                    const size_t idx = (size_t)m * plane + (size_t)oy * W + ox;
                    dx[idx] = v + (inject ? inject[idx] : 0.f);
                };
                if (half == 0) { out_m(0); if (M == 3) out_m(1); }
                else if (M >= 2) out_m(M - 1);
            }
        }
    }
}

// fp32, aligned widths (W % 4 == 0): every lane fetches 16-byte QUADS of the diff (four pixels of one channel) and feeds them to
// FOUR accumulator tiles, one per pixel of the quad -- 32 loads in flight per lane instead of 96 four-byte ones.  One workgroup =
// 5 x 64 output pixels; its halo tile is 7 rows x 72 columns (x0 - 4 .. x0 + 67) = 126 quads = one 32-lane group per wave.
constexpr int DQ_TH = 5, DQ_TW = 64;
constexpr int DQ_HH = DQ_TH + 2, DQ_HQ = (DQ_TW + 8) / 4;          // 7 rows x 18 quads
constexpr int DQ_NQ = DQ_HH * DQ_HQ;                               // 126
constexpr int DQ_ZS = DQ_HH * DQ_HQ * 4 + 1;                       // padded row stride of Z (505 floats)
static_assert(DQ_NQ <= 128, "one quad per lane of the four waves");

template <int M>
__global__ __launch_bounds__(256, 2) void conv3x3_dgrad_first_f32q(const float* __restrict__ dy, const float* __restrict__ w,
                                                                   float* __restrict__ dx, const float* __restrict__ inject,
                                                                   int Cout, int H, int W)
{
    __shared__ float z_s[DF_ROWS * DQ_ZS];
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, khalf = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int x0 = blockIdx.x * DQ_TW, y0 = blockIdx.y * DQ_TH;
    const size_t plane = (size_t)H * W;
    constexpr int KC = 32;                                   // k-pairs per pass (64 channels)
    const int qi = wave * 32 + l31;                          // this lane's quad of the halo tile
    const int hy = qi / DQ_HQ, hq = qi - hy * DQ_HQ;
    const int gy = y0 - 1 + hy, gx = x0 - 4 + 4 * hq;        // W % 4 == 0: a quad is inside the row or outside it
    const int off = (qi < DQ_NQ && gy >= 0 && gy < H && gx >= 0 && gx < W) ? gy * W + gx : -1;
    df_f32x16 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
    const int r = l31, m_r = r / 9, tap_r = r - 9 * m_r;
    for (int c0 = 0; c0 < Cout; c0 += 2 * KC) {
        float4 b[KC];                                        // every operand of this pass first: 32 x 16 bytes in flight per lane
#pragma unroll
        for (int kp = 0; kp < KC; ++kp) {
            const int c = c0 + 2 * kp + khalf;
            b[kp] = (off >= 0 && c < Cout) ? *reinterpret_cast<const float4*>(dy + (size_t)c * plane + off) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        if (c0) __syncthreads();
        const int nw = min(2 * KC, Cout - c0) * M * 9;
        for (int i = tid; i < nw; i += 256) z_s[i] = w[(size_t)c0 * M * 9 + i];
        __syncthreads();
        const bool a_row = r < 9 * M;
        const float* a_src = z_s + (khalf * M + m_r) * 9 + tap_r;            // + 2 kp * M * 9: this lane's row, channel 2 kp + khalf
#pragma unroll
        for (int kp = 0; kp < KC; ++kp) {
            // (A straight from LDS, one value per four MFMAs: the 128 operand registers leave no room for a register copy of it)
            const float a = (a_row && c0 + 2 * kp + khalf < Cout) ? a_src[2 * kp * M * 9] : 0.f;
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b[kp].x, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b[kp].y, acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b[kp].z, acc[2], 0, 0, 0);
            acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b[kp].w, acc[3], 0, 0, 0);
        }
    }
    __syncthreads();                                         // every wave has taken its A operands out of LDS
    // park rows 0 .. 26: column l31 of accumulator j is pixel j of quad qi = halo pixel (hy, 4 hq + j)
    if (qi < DQ_NQ) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = (e & 3) + 8 * (e >> 2) + 4 * khalf;
                if (row < DF_ROWS) z_s[row * DQ_ZS + 4 * qi + j] = acc[j][e];
            }
    }
    __syncthreads();
    for (int o = tid; o < DQ_TH * DQ_TW; o += 256) {
        const int py = o / DQ_TW, px = o - py * DQ_TW;
        const int oy = y0 + py, ox = x0 + px;
        if (oy >= H || ox >= W) continue;
#pragma unroll
        for (int m = 0; m < M; ++m) {
            float v = 0.f;
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx)      // source pixel (y - ky + 1, x - kx + 1) = halo (py + 2 - ky, px + 5 - kx)
                    v += z_s[(m * 9 + ky * 3 + kx) * DQ_ZS + (py + 2 - ky) * (DQ_HQ * 4) + (px + 5 - kx)];
            const size_t idx = (size_t)m * plane + (size_t)oy * W + ox;
            dx[idx] = v + (inject ? inject[idx] : 0.f);
        }
    }
}

// Measured (profiles/r02_zb_dgrad_first_ab.txt): the bf16 variant beats the VALU kernel (2048^2: 303 -> 265 us), the fp32 variant does not (1024^2:
// 122 -> 200 us: 96 four-byte operand loads per lane against 12 sixteen-byte ones) -- so by default only the bf16 path uses this file.
// ST2_DGRAD_FIRST=1 routes fp32 here too, =0 neither; read per launch (the tests run every combination).
bool conv_dgrad_first_ok(int Cout, int Cin, int H, int W, bool bf16)
{
    const char* e = getenv("ST2_DGRAD_FIRST");
    const bool enabled = e && *e ? (*e == '1') : bf16;
    return enabled && Cin >= 1 && Cin <= 3 && Cout >= 2 && Cout % (bf16 ? 16 : 2) == 0 && H >= 1 && W >= 1 &&
           (unsigned long long)H * W < 0x7fffffffull;
}

// the quad variant: aligned rows.  Measured at 1024^2: 126 us against 125 us for the VALU kernel (both read the 268 MB diff at
// 2.2 TB/s: 64 channel planes x a few short row segments per workgroup is what bounds them, not the arithmetic) -- so it is OFF
// unless ST2_DGRAD_FIRST_Q=1 (read per launch; the tests run it)
bool conv_dgrad_first_quad_ok(int Cout, int Cin, int H, int W, const float* dy)
{
    const char* e = getenv("ST2_DGRAD_FIRST_Q");
    return (e && *e == '1') && Cin >= 1 && Cin <= 3 && Cout >= 2 && Cout % 2 == 0 && W % 4 == 0 && H >= 1 &&
           (unsigned long long)H * W < 0x7fffffffull && (reinterpret_cast<uintptr_t>(dy) & 15) == 0;
}

hipError_t launch_conv3x3_dgrad_first_quad(const float* dy, const float* w, float* dx, const float* inject, int Cout, int Cin, int H, int W, hipStream_t s)
{
    if (!conv_dgrad_first_quad_ok(Cout, Cin, H, W, dy)) return hipErrorInvalidValue;
    const dim3 grid((W + DQ_TW - 1) / DQ_TW, (H + DQ_TH - 1) / DQ_TH);
    switch (Cin) {
    case 1: conv3x3_dgrad_first_f32q<1><<<grid, 256, 0, s>>>(dy, w, dx, inject, Cout, H, W); break;
    case 2: conv3x3_dgrad_first_f32q<2><<<grid, 256, 0, s>>>(dy, w, dx, inject, Cout, H, W); break;
    default: conv3x3_dgrad_first_f32q<3><<<grid, 256, 0, s>>>(dy, w, dx, inject, Cout, H, W); break;
    }
    return hipGetLastError();
}

// strips x segments of the walkers: as many segments as keep the grid within the resident capacity (2 workgroups per CU), at least 8 rows each
static dim3 dgrad_strip_grid(int H, int W, int* seg_out)
{
    static int capacity = 0;
    if (!capacity) {
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
        capacity = 2 * cus;
    }
    const int strips = (W + DS_OW - 1) / DS_OW;
    const int max_segs = capacity / strips > 0 ? capacity / strips : 1;
    int seg = (H + max_segs - 1) / max_segs;
    if (seg < 8) seg = 8;
    *seg_out = seg;
    return dim3(strips, (H + seg - 1) / seg);
}

// fp32 strip walker: Cout = 64, tensor below 4 GiB (32-bit buffer offsets); ST2_DGRAD_FIRST_STRIP=0 switches it off (read per launch)
bool conv_dgrad_first_strip_ok(int Cout, int Cin, int H, int W)
{
    const char* e = getenv("ST2_DGRAD_FIRST_STRIP");
    return !(e && *e == '0') && Cout == 64 && Cin >= 1 && Cin <= 3 && H >= 1 && W >= 1 && 4ull * Cout * H * W < 0xfffffff0ull;
}

hipError_t launch_conv3x3_dgrad_first_strip(const float* dy, const float* w, float* dx, const float* inject, int Cout, int Cin, int H, int W, hipStream_t s)
{
    if (!conv_dgrad_first_strip_ok(Cout, Cin, H, W)) return hipErrorInvalidValue;
    int seg = 0;
    const dim3 grid = dgrad_strip_grid(H, W, &seg);
    const unsigned bytes = (unsigned)(4ull * Cout * H * W);
    switch (Cin) {
    case 1: conv3x3_dgrad_first_f32_strip<1><<<grid, 256, 0, s>>>(dy, w, dx, inject, H, W, seg, bytes); break;
    case 2: conv3x3_dgrad_first_f32_strip<2><<<grid, 256, 0, s>>>(dy, w, dx, inject, H, W, seg, bytes); break;
    default: conv3x3_dgrad_first_f32_strip<3><<<grid, 256, 0, s>>>(dy, w, dx, inject, H, W, seg, bytes); break;
    }
    return hipGetLastError();
}

hipError_t launch_conv3x3_dgrad_first(const float* dy, const float* w, float* dx, const float* inject, int Cout, int Cin, int H, int W, hipStream_t s)
{
    if (!conv_dgrad_first_ok(Cout, Cin, H, W, false)) return hipErrorInvalidValue;
    const dim3 grid((W + DF_TW - 1) / DF_TW, (H + DF_TH - 1) / DF_TH);
    switch (Cin) {
    case 1: conv3x3_dgrad_first_f32<1><<<grid, 256, 0, s>>>(dy, w, dx, inject, Cout, H, W); break;
    case 2: conv3x3_dgrad_first_f32<2><<<grid, 256, 0, s>>>(dy, w, dx, inject, Cout, H, W); break;
    default: conv3x3_dgrad_first_f32<3><<<grid, 256, 0, s>>>(dy, w, dx, inject, Cout, H, W); break;
    }
    return hipGetLastError();
}

hipError_t launch_conv3x3_dgrad_first16(const unsigned short* dy16, const float* w_rounded, float* dx, const float* inject, int Cout, int Cin,
                                        int H, int W, hipStream_t s)
{
    if (!conv_dgrad_first_ok(Cout, Cin, H, W, true) || (reinterpret_cast<uintptr_t>(dy16) & 15) != 0) return hipErrorInvalidValue;
    const uint4* q16 = reinterpret_cast<const uint4*>(dy16);
    {   // Cout = 64: the strip walker (ST2_DGRAD_FIRST_STRIP=0: the tile kernel; read per launch, the tests compare both)
        const char* e = getenv("ST2_DGRAD_FIRST_STRIP");
        const unsigned long long bytes = 16ull * (Cout / 8) * H * W;
        if (!(e && *e == '0') && Cout == 64 && bytes < 0xfffffff0ull) {
            int seg = 0;
            const dim3 grid = dgrad_strip_grid(H, W, &seg);
            switch (Cin) {
            case 1: conv3x3_dgrad_first_bf16_strip<1><<<grid, 256, 0, s>>>(q16, w_rounded, dx, inject, H, W, seg, (unsigned)bytes); break;
            case 2: conv3x3_dgrad_first_bf16_strip<2><<<grid, 256, 0, s>>>(q16, w_rounded, dx, inject, H, W, seg, (unsigned)bytes); break;
            default: conv3x3_dgrad_first_bf16_strip<3><<<grid, 256, 0, s>>>(q16, w_rounded, dx, inject, H, W, seg, (unsigned)bytes); break;
            }
            return hipGetLastError();
        }
    }
    const dim3 grid((W + DF_TW - 1) / DF_TW, (H + DF_TH - 1) / DF_TH);
    switch (Cin) {
    case 1: conv3x3_dgrad_first_bf16<1><<<grid, 256, 0, s>>>(q16, w_rounded, dx, inject, Cout, H, W); break;
    case 2: conv3x3_dgrad_first_bf16<2><<<grid, 256, 0, s>>>(q16, w_rounded, dx, inject, Cout, H, W); break;
    default: conv3x3_dgrad_first_bf16<3><<<grid, 256, 0, s>>>(q16, w_rounded, dx, inject, Cout, H, W); break;
    }
    return hipGetLastError();
}

}  // namespace st2
