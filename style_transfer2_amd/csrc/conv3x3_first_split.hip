// Forward of the FIRST conv (conv1_1: 3 -> 64 channels; pycaffe Convolution forward behind worker.py:77-86) for the bf16 feature path.
//
// The image must not be rounded to bf16 (an update of 0.1 grey levels on a pixel near 150 would vanish), so until round 3 this layer ran on
// the fp32 matrix cores: K = 27 -> 18 v_mfma_f32_32x32x2 per 32 x 32 output tile = 0.14 ms of matrix-pipe time at 2048^2 for a launch whose
// HBM traffic (a 48 MB image in, a 0.54 GB bf16 blob out) is worth 0.09 ms -- measured 0.254 ms.  Here every fp32 operand is split into three
// bf16 terms (x = x1 + x2 + x3 exactly: 8 + 8 + 8 significand bits) and the product is taken as the six bf16 x bf16 partial products of
// weight <= 2 in the expansion,
//     x w = x1 w1 + (x1 w2 + x2 w1) + (x1 w3 + x3 w1 + x2 w2)  [+ terms below 2^-24 |x w|],
// each exact in fp32, accumulated in fp32 by v_mfma_f32_32x32x16_bf16 (smallest terms first): fp32-grade results (the test bar against the
// fp32-operand oracle stays 1e-6) at 12 bf16 MFMAs = 384 cycles per tile instead of 18 fp32 ones = 1152.  The launch becomes HBM-bound.
// GEMM: rows = 32 output channels (A = split weights, packed on the host once), columns = 32 pixels of one image row (B = the 27 taps of
// each pixel, gathered from an LDS copy of the halo tile and split in registers), K = 32 per partial product (27 taps + 5 zeros).
// One workgroup = 4 rows x 128 pixels x all channels; a wave takes one row (four 32-pixel groups); split weights and bias sit in LDS.  Epilogue: bias, ReLU, the bf16 channel-blocked copy
// [M/8][H][W][8] in 16-byte stores (lane-half exchange as in conv3x3_mfma.hip) and, if asked for, the fp32 blob.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "st2_kernels.h"

namespace st2 {

typedef float fs_f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 fs_bf16x8 __attribute__((ext_vector_type(8)));

constexpr int FS_TW = 128;                     // output tile of a workgroup: FS_TH rows x 128 pixels; its 32-pixel groups are dealt to the four waves in row-major runs
constexpr int FS_TPR = FS_TW / 32;             // 32-pixel MFMA column groups per tile row
constexpr int FS_HW = FS_TW + 2;
constexpr int FS_MAXC = 3;                     // input channels: RGB (27 taps + the bias tap fit the 32-wide K block)
constexpr int FS_MAXM = 128;                   // output channels (the split weights of all of them sit in LDS: 6 KB per 32)

static unsigned short fs_f2bf(float f)           // round-to-nearest-even, host side
{
    unsigned u;
    memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);
    return (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}
static float fs_bf2f(unsigned short h) { unsigned u = (unsigned)h << 16; float f; memcpy(&f, &u, 4); return f; }

size_t conv_first_split_pack_elems(int Cout) { return (size_t)(Cout / 32) * 3 * 2 * 64 * 8; }

// dst[(((mt * 3 + split) * 2 + kstep) * 64 + lane) * 8 + j] = split `split` of w[mt * 32 + lane % 32][kk], kk = 16 kstep + 8 (lane / 32) + j
// (kk = c * 9 + ky * 3 + kx; zero for kk >= 9 Cin): the A fragment of v_mfma_f32_32x32x16_bf16, one 16-byte load per lane
// Tap 27 carries the bias: its B operand is the constant 1 (split: 1, 0, 0).
void pack_conv_first_split(const float* w /*Cout, Cin, 3, 3*/, const float* bias /*Cout or nullptr*/, int Cout, int Cin, unsigned short* dst)
{
    memset(dst, 0, conv_first_split_pack_elems(Cout) * sizeof(unsigned short));
    for (int m = 0; m < Cout; ++m)
        for (int kk = 0; kk < 28; ++kk) {
            if (kk >= 9 * Cin && kk != 27) continue;
            const float v = kk == 27 ? (bias ? bias[m] : 0.f) : w[(size_t)m * Cin * 9 + kk];
            const unsigned short h1 = fs_f2bf(v);
            const float r1 = v - fs_bf2f(h1);
            const unsigned short h2 = fs_f2bf(r1);
            const unsigned short h3 = fs_f2bf(r1 - fs_bf2f(h2));
            const unsigned short hs[3] = {h1, h2, h3};
            const int mt = m / 32, lane = (m % 32) + 32 * ((kk % 16) / 8), kstep = kk / 16, j = kk % 8;
            for (int s = 0; s < 3; ++s) dst[((((size_t)mt * 3 + s) * 2 + kstep) * 64 + lane) * 8 + j] = hs[s];
        }
}

struct FirstSplitArgs {
    const float* x; const uint4* wpk; float* out; unsigned short* out16;
    int M, H, W, relu;
    unsigned short* bits;                          // optional: one bit per output element, "the bf16 copy is non-zero" (Conv16Problem::bits_out's layout)
};

// two consecutive taps' 16-bit fields -> one dword of a B fragment
__device__ __forceinline__ unsigned lo16_pair(unsigned a, unsigned b) { return __builtin_amdgcn_perm(b, a, 0x05040100u); }   // {b.lo : a.lo}
__device__ __forceinline__ unsigned hi16_pair(unsigned a, unsigned b) { return __builtin_amdgcn_perm(b, a, 0x07060302u); }   // {b.hi : a.hi}

// bit e of the result: 16-bit field e of (u0.x, u0.y, u1.x, u1.y) is non-zero (eight bf16 values -> one byte of a sign map)
__device__ __forceinline__ unsigned nonzero_halves(uint2 u0, uint2 u1)
{
    auto two = [](unsigned w) { return ((w & 0xffffu) ? 1u : 0u) | ((w >> 16) ? 2u : 0u); };
    return two(u0.x) | (two(u0.y) << 2) | (two(u1.x) << 4) | (two(u1.y) << 6);
}

// MT = the most 32-channel groups a launch may have (sizes the weight image in LDS: 6 KB each; 64 channels -> 31 KB with the
// halo tile, four workgroups per CU instead of three)
template <int FS_TH, int MT>
__global__ __launch_bounds__(256, MT <= 2 ? 4 : 3) void conv3x3_first_split_k(const FirstSplitArgs a)
{
    constexpr int FS_HH = FS_TH + 2;
    // the halo tile, every value already split: .x = x1 | x2 << 16, .y = x3 (bf16 bit patterns) -- a pixel is split once, not once per tap
    __shared__ uint2 x_s[FS_MAXC * FS_HH * FS_HW];
    __shared__ uint4 w_s[MT * 6 * 64];                              // A fragments: [mt][split][k-step][lane]
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, khalf = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int x0 = blockIdx.x * FS_TW, y0 = blockIdx.y * FS_TH;
    const int H = a.H, W = a.W;
    const size_t plane = (size_t)H * W;
    const int mtiles = a.M / 32;
    // split weights (+ the bias as tap 27) once per workgroup: a global load per use costs its latency sixteen times per wave
    for (int e = tid; e < mtiles * 6 * 64; e += 256) w_s[e] = a.wpk[e];
    for (int e = tid; e < FS_MAXC * FS_HH * FS_HW; e += 256) {       // zero outside the image (pad = 1)
        const int c = e / (FS_HH * FS_HW), r = e - c * (FS_HH * FS_HW);
        const int hy = r / FS_HW, hx = r - hy * FS_HW;
        const int gy = y0 - 1 + hy, gx = x0 - 1 + hx;
        const float v = (gy >= 0 && gy < H && gx >= 0 && gx < W) ? a.x[(size_t)c * plane + (size_t)gy * W + gx] : 0.f;
        const __bf16 h1 = (__bf16)v;
        const float r1 = v - (float)h1;
        const __bf16 h2 = (__bf16)r1;
        const __bf16 h3 = (__bf16)(r1 - (float)h2);
        x_s[e] = make_uint2((unsigned)__builtin_bit_cast(unsigned short, h1) | ((unsigned)__builtin_bit_cast(unsigned short, h2) << 16),
                            (unsigned)__builtin_bit_cast(unsigned short, h3));
    }
    // this lane's 16 taps: kk = 8 khalf + j (k-step 0) and 16 + 8 khalf + j (k-step 1), kk = c * 9 + ky * 3 + kx -> offset into the halo
    // tile relative to its pixel.  kk = 27 is the constant 1 that carries the bias, 28 .. 31 are zeros (lanes with khalf = 1, k-step 1).
    int off[2][8];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int kk = 16 * ks + 8 * khalf + j;
            const int c = kk / 9, rem = kk - 9 * c, ky = rem / 3, kx = rem - 3 * ky;
            off[ks][j] = kk < 27 ? (c * FS_HH + ky) * FS_HW + kx : 0;
        }
    __syncthreads();
    constexpr int GPW = FS_TH * FS_TPR / 4;                          // groups per wave
    static_assert(FS_TH * FS_TPR % 4 == 0, "whole groups per wave");
#pragma unroll 1
    for (int g = 0; g < GPW; ++g) {
        const int gi = wave * GPW + g, row = gi / FS_TPR, grp = gi % FS_TPR, gy = y0 + row;
        if (gy >= H || x0 + grp * 32 >= W) continue;                 // wave-uniform
        const int col = grp * 32 + l31, gx = x0 + col;
        const int base = row * FS_HW + col;
        // B fragments: the three splits of the 16 taps of pixel (row, col)
        uint2 t[2][8];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int j = 0; j < 8; ++j) t[ks][j] = x_s[off[ks][j] + base];
        if (khalf) {                                                 // taps 27 .. 31 of this lane
            t[1][3] = make_uint2(0x3f80u, 0u);                       // 1.0 = 0x3f80 | 0 << 16, x3 = 0
#pragma unroll
            for (int j = 4; j < 8; ++j) t[1][j] = make_uint2(0u, 0u);
        }
        fs_bf16x8 b1[2], b2[2], b3[2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            uint4 q1, q2, q3;
            q1.x = lo16_pair(t[ks][0].x, t[ks][1].x); q1.y = lo16_pair(t[ks][2].x, t[ks][3].x);
            q1.z = lo16_pair(t[ks][4].x, t[ks][5].x); q1.w = lo16_pair(t[ks][6].x, t[ks][7].x);
            q2.x = hi16_pair(t[ks][0].x, t[ks][1].x); q2.y = hi16_pair(t[ks][2].x, t[ks][3].x);
            q2.z = hi16_pair(t[ks][4].x, t[ks][5].x); q2.w = hi16_pair(t[ks][6].x, t[ks][7].x);
            q3.x = lo16_pair(t[ks][0].y, t[ks][1].y); q3.y = lo16_pair(t[ks][2].y, t[ks][3].y);
            q3.z = lo16_pair(t[ks][4].y, t[ks][5].y); q3.w = lo16_pair(t[ks][6].y, t[ks][7].y);
            b1[ks] = __builtin_bit_cast(fs_bf16x8, q1); b2[ks] = __builtin_bit_cast(fs_bf16x8, q2); b3[ks] = __builtin_bit_cast(fs_bf16x8, q3);
        }
        const bool inside = gx < W;
        const size_t pix = (size_t)gy * W + (gx < W ? gx : W - 1);
        for (int mt = 0; mt < mtiles; ++mt) {
            fs_bf16x8 w1[2], w2[2], w3[2];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                w1[ks] = __builtin_bit_cast(fs_bf16x8, w_s[((mt * 3 + 0) * 2 + ks) * 64 + lane]);
                w2[ks] = __builtin_bit_cast(fs_bf16x8, w_s[((mt * 3 + 1) * 2 + ks) * 64 + lane]);
                w3[ks] = __builtin_bit_cast(fs_bf16x8, w_s[((mt * 3 + 2) * 2 + ks) * 64 + lane]);
            }
            fs_f32x16 acc;
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {                          // smallest partial products first
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w3[ks], b1[ks], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1[ks], b3[ks], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w2[ks], b2[ks], acc, 0, 0, 0);
            }
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w2[ks], b1[ks], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1[ks], b2[ks], acc, 0, 0, 0);
            }
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1[ks], b1[ks], acc, 0, 0, 0);
            // C/D map: column = lane & 31 (the pixel), row = (e & 3) + 8 (e >> 2) + 4 (lane >> 5) (the channel within the tile)
            unsigned bits = 0;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int mbase = mt * 32 + 4 * khalf + 16 * h;
                float v[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    v[e] = acc[8 * h + e];
                    if (a.relu) v[e] = v[e] > 0.f ? v[e] : 0.f;
                }
                if (a.out && inside) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) a.out[(size_t)(mbase + (e & 3) + 8 * (e >> 2)) * plane + pix] = v[e];
                }
                if (a.out16) {
                    // this lane holds half (4 channels) of two 8-channel quads of its pixel, lane ^ 32 the other halves: exchange, then one
                    // 16-byte store per lane (conv3x3_mfma.hip's epilogue)
                    typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
                    bf16x4 pk0, pk1;
#pragma unroll
                    for (int e = 0; e < 4; ++e) { pk0[e] = (__bf16)v[e]; pk1[e] = (__bf16)v[4 + e]; }
                    const uint2 u0 = __builtin_bit_cast(uint2, pk0), u1 = __builtin_bit_cast(uint2, pk1);
                    bits |= nonzero_halves(u0, u1) << (8 * h);
                    const auto sx = __builtin_amdgcn_permlane32_swap(u0.x, u1.x, false, false);
                    const auto sy = __builtin_amdgcn_permlane32_swap(u0.y, u1.y, false, false);
                    const int mq = mt * 32 + 16 * h + 8 * khalf;                       // first channel of this lane's quad
                    if (inside)
                        *reinterpret_cast<uint4*>(a.out16 + ((size_t)(mq >> 3) * plane + pix) * 8) = make_uint4(sx[0], sy[0], sx[1], sy[1]);
                }
            }
            if (a.bits) {                                            // lane ^ 32 holds the other 16-bit word of the same pixel: one dword store per pixel
                const auto sw = __builtin_amdgcn_permlane32_swap(bits, bits, false, false);
                if (inside && !khalf) reinterpret_cast<unsigned*>(a.bits)[(size_t)mt * plane + pix] = bits | (sw[1] << 16);
            }
        }
    }
}

// ST2_FIRST_SPLIT=0: the fp32-matrix-core kernel for this layer (read per launch; the tests compare both)
bool conv_first_split_ok(int Cin, int Cout, int H, int W)
{
    const char* e = getenv("ST2_FIRST_SPLIT");
    if (e && *e == '0') return false;
    return Cin == FS_MAXC && Cout >= 32 && Cout % 32 == 0 && Cout <= FS_MAXM && H >= 1 && W >= 1 && (unsigned long long)H * W * Cout < 0x7fffffffull * 2;
}

hipError_t launch_conv3x3_first_split(const float* x, const unsigned short* wpk, float* out, unsigned short* out16,
                                      int Cin, int Cout, int H, int W, int relu, hipStream_t s, unsigned short* bits_out)
{
    if (!conv_first_split_ok(Cin, Cout, H, W) || (!out && !out16) || (bits_out && (!out16 || (reinterpret_cast<uintptr_t>(bits_out) & 3) != 0)) || (reinterpret_cast<uintptr_t>(wpk) & 15) != 0 ||
        (out16 && (reinterpret_cast<uintptr_t>(out16) & 15) != 0))
        return hipErrorInvalidValue;
    FirstSplitArgs a{x, reinterpret_cast<const uint4*>(wpk), out, out16, Cout, H, W, relu, bits_out};
    // four rows per workgroup: measured at 2048^2 against 1 / 2 / 8 (0.267 / 0.217 / 0.233 ms against 0.211)
    const dim3 grid((W + FS_TW - 1) / FS_TW, (H + 3) / 4);
    if (Cout <= 64) conv3x3_first_split_k<4, 2><<<grid, 256, 0, s>>>(a);
    else conv3x3_first_split_k<4, FS_MAXM / 32><<<grid, 256, 0, s>>>(a);
    return hipGetLastError();
}

}  // namespace st2
