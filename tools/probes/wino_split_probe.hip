// tools/probes/wino_split_probe.hip -- go / no-go probe (round 5) for the "split-operand Winograd": the transform-domain products of
// F(2x2,3x3) taken as SIX EXACT bf16 partial products of three-way split fp32 operands on v_mfma_f32_32x32x16_bf16 (fp32 accumulate)
// instead of one v_mfma_f32_32x32x2_f32 product: 12 matrix-pipe cycles per k instead of 32.  Not product code; one forward layer
// (bias + ReLU), aligned shapes only (H % 8 == 0, W % 32 == 0, K % 16 == 0, M % 64 == 0), checked against a CPU loop nest.
//
// Workgroup = 4 waves = 64 output channels x (8 rows x 32 columns) = 64 tiles of 2x2 outputs.  Wave i owns ROW i of the 4x4
// transform-domain matrix (positions 4 i .. 4 i + 3) for all of it: 4 positions x 2 channel groups x 2 tile groups = 16 accumulators
// of 32x32 (256 AGPRs).  What that buys:
//   * every U fragment is loaded by exactly one wave (L2 -> VGPR, 16 bytes per lane = one A operand), 96 KiB per 16-channel chunk per CU;
//   * row i of B^T d B needs two raw rows only, and the lane that builds it is the lane that feeds it to the matrix core: lane
//     (tile, k half) reads 8 channels of its tile from the raw LDS image, transforms, splits three ways and packs -- the packed
//     registers ARE the B operands.  V never touches LDS; the waves share the raw image only (one barrier per 96 MFMAs);
//   * the price is the epilogue: the output transform needs all four rows, exchanged through LDS.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <algorithm>
#include <type_traits>
#include <vector>
#include "st2_probes.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void* lptr_t;

constexpr int WS_CH = 16;                 // input channels per chunk = the k of one MFMA
constexpr int WS_IW = 40;                 // staged columns x0-4 .. x0+35
constexpr int WS_ROWS = 10;               // staged rows y0-1 .. y0+8
constexpr int WS_PLANE = WS_ROWS * WS_IW; // 400 floats per channel
constexpr int WS_RAW = WS_CH * WS_PLANE;  // 6400 floats = 25 wave-DMAs of 1 KiB
constexpr int WS_PIECES = WS_RAW / 256;
constexpr unsigned kOOB = 0xffffffffu;
constexpr int RAW_SHIFT = 1;             // the staged image starts one float into its buffer: pixel x0 - 1 + 2 t sits at an even index (ds_read_b64)

struct WsArgs {
    const float* in; const uint4* upack; const float* bias; float* out;
    int K, M, H, W, nch, tiles_x, tiles_y, relu;
    unsigned in_bytes, u_bytes;
    unsigned long long* stamps;           // per block: {shader cycles of the main loop, 100 MHz ticks of it, cycles prologue, cycles epilogue}
};

__device__ __forceinline__ bf16x8 as_bf(const uint4& u) { return __builtin_bit_cast(bf16x8, u); }
__device__ __forceinline__ unsigned pk_bf16(f32x2 v) { return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2)); }

// a - b on a register pair in ONE instruction (the compiler splits half of these into two v_add_f32)
__device__ __forceinline__ f32x2 pk_sub(f32x2 a, f32x2 b)
{
    f32x2 r;
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

__device__ __forceinline__ f32x2 pk_add(f32x2 a, f32x2 b)
{
    f32x2 r;
    asm("v_pk_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// {a, b} rounded to bf16 (nearest even) in one dword: a in the low half
__device__ __forceinline__ unsigned cvt2(float a, float b)
{
    f32x2 v; v.x = a; v.y = b;
    return pk_bf16(v);
}
// the fp32 values of the LOW (first channel) / HIGH (second channel) bf16 halves of two packed dwords, as a register pair
__device__ __forceinline__ f32x2 lo_pair(unsigned h0, unsigned h1)
{
    f32x2 t; t.x = __builtin_bit_cast(float, h0 << 16); t.y = __builtin_bit_cast(float, h1 << 16);
    return t;
}
__device__ __forceinline__ f32x2 hi_pair(unsigned h0, unsigned h1)
{
    f32x2 t; t.x = __builtin_bit_cast(float, h0 & 0xffff0000u); t.y = __builtin_bit_cast(float, h1 & 0xffff0000u);
    return t;
}

typedef int i32x4 __attribute__((ext_vector_type(4)));
// LDS-DMA by hand (the compiler does not see it: it would otherwise drain EVERY outstanding DMA -- s_waitcnt vmcnt(0) -- before any LDS
// read that may alias one of them; the waits are placed by hand instead): 64 lanes x 16 bytes -> LDS [lds_addr, lds_addr + 1 KiB)
__device__ __forceinline__ void dma16(const i32x4& rsrc, unsigned lds_addr, unsigned voff, unsigned soff)
{
    asm volatile("s_mov_b32 m0, %0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" :: "s"(lds_addr), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
}
__device__ __forceinline__ i32x4 make_rsrc(const void* p, unsigned bytes)
{
    const unsigned long long a = (unsigned long long)p;
    i32x4 r;
    r.x = (int)(unsigned)a; r.y = (int)(unsigned)((a >> 32) & 0xffffu); r.z = (int)bytes; r.w = 0x00020000;
    return r;
}
template <int I> __device__ __forceinline__ void set_comp(uint4& v, unsigned x)
{
    if constexpr (I == 0) v.x = x; else if constexpr (I == 1) v.y = x; else if constexpr (I == 2) v.z = x; else v.w = x;
}
#define WS_INL __attribute__((always_inline))

// VAR (what-bounds-the-loop experiments, results wrong): 1 = no U refills in the loop, 2 = no B builds, 4 = no raw reads / row transform
template <int STAMP, int VAR = 0>
__global__ __launch_bounds__(256, 1) void wino_split_probe_k(const WsArgs a)
{
    // raw[2][6400] floats during the main loop; the row exchange of the epilogue afterwards ([src wave][combo][e 8][lane] float2 = 64 KiB)
    __shared__ __attribute__((aligned(16))) float lds[16384];
    // U ring: slot j holds the A fragments of position j of a chunk, [slot][wave][m group * 3 + split][lane]; wave-private (each wave DMAs
    // and reads its own 6 KiB per slot): no barrier ever concerns it
    __shared__ uint4 u_s[4 * 4 * 6 * 64];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int t31 = lane & 31, kq = lane >> 5;

    const int nwg = gridDim.x;
    const int orig = blockIdx.x;
    const int xcd = orig & 7, q8 = nwg >> 3, r8 = nwg & 7;
    const int logical = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (orig >> 3);
    const int n_pt = a.tiles_x * a.tiles_y;
    const int mt = logical / n_pt;
    const int pt = logical - mt * n_pt;
    const int tx = pt % a.tiles_x, ty = pt / a.tiles_x;
    const int y0 = ty * 8, x0 = tx * 32;
    const unsigned plane = (unsigned)a.H * a.W;
    const int nch = a.nch;

    const unsigned long long t_begin = STAMP ? __builtin_amdgcn_s_memtime() : 0ull;

    const i32x4 rs_i = make_rsrc(a.in, a.in_bytes), rs_u = make_rsrc(a.upack, a.u_bytes);
    const unsigned lds_raw = (unsigned)(size_t)(lptr_t)lds, lds_u = (unsigned)(size_t)(lptr_t)u_s;
    unsigned ioff[7];
#pragma unroll
    for (int t = 0; t < 7; ++t) {
        const int piece = wave + 4 * t;
        const int e = (piece * 64 + lane) * 4;
        const int c = e / WS_PLANE;
        const int rem = e - c * WS_PLANE;
        const int rr = rem / WS_IW;
        const int col = rem - rr * WS_IW;
        const int gy = y0 - 1 + rr, gx = x0 - 4 + col;
        const bool ok = piece < WS_PIECES && gy >= 0 && gy < a.H && gx >= 0 && gx + 3 < a.W;
        ioff[t] = ok ? ((unsigned)c * plane + (unsigned)gy * a.W + gx) * 4u : kOOB;       // out of range -> the hardware writes zeros: the padding
    }
    auto dma_raw = [&](int ch, int buf) WS_INL {
        const unsigned coff = (unsigned)ch * WS_CH * plane * 4u;                          // scalar offset (outside the range check)
#pragma unroll
        for (int t = 0; t < 7; ++t) {
            const int piece = wave + 4 * t;                     // wave-uniform
            if (piece < WS_PIECES) dma16(rs_i, lds_raw + (unsigned)(buf * WS_RAW + piece * 256 + RAW_SHIFT) * 4u, ioff[t], coff);
        }
    };

    // row i of B^T d:  i = 0: d0 - d2   1: d1 + d2   2: d2 - d1   3: d1 - d3      (X = first, Y = second row; staged row 0 = y0 - 1)
    const int row_x = wave == 0 ? 0 : wave == 2 ? 2 : 1;
    const int row_y = wave == 0 ? 2 : wave == 1 ? 2 : wave == 2 ? 1 : 3;
    const int x_base = (8 * kq) * WS_PLANE + (2 * (t31 >> 4)) * WS_IW + 2 * (t31 & 15) + 3 + RAW_SHIFT;     // column 3 = pixel x0 - 1
    const float* const pX = lds + x_base + row_x * WS_IW;
    const float* const pY = lds + x_base + row_y * WS_IW;

    // A fragments: [m tile][chunk][pos 16][m group 2][split 3][lane 64] uint4 = 6 KiB per (chunk, position) and wave
    const unsigned u_lane = (unsigned)lane * 16u;
    const unsigned u_base = (unsigned)((mt * nch * 16 + 4 * wave) * 6 * 64 * 16);      // wave-uniform
    auto dma_u = [&](int c, int j) WS_INL {   // position j of chunk c -> slot j
        const unsigned so = u_base + (unsigned)((c * 16 + j) * 6 * 64 * 16);
        const unsigned dst = lds_u + (unsigned)((j * 4 + wave) * 6 * 64 * 16);
#pragma unroll
        for (int q = 0; q < 6; ++q) dma16(rs_u, dst + q * 1024, u_lane, so + q * 1024);
    };
    auto dma_u_one = [&](int c, int j, int q) WS_INL {
        dma16(rs_u, lds_u + (unsigned)((j * 4 + wave) * 6 * 64 * 16) + q * 1024, u_lane, u_base + (unsigned)((c * 16 + j) * 6 * 64 * 16) + q * 1024);
    };
    uint4 aop[2][6];                     // A operands of the current / next position: [m group * 3 + split]
    auto a_fetch = [&](int j, int set) WS_INL {
        const uint4* p = u_s + (j * 4 + wave) * 6 * 64 + lane;
#pragma unroll
        for (int q = 0; q < 6; ++q) aop[set][q] = p[q * 64];
    };

    auto a_fetch_one = [&](int j, int set, int q) WS_INL { aop[set][q] = u_s[(j * 4 + wave) * 6 * 64 + q * 64 + lane]; };
    // VALU budget: a bf16 MFMA occupies the matrix pipe for 32 cycles; plain VALU instructions (~5 cycles each) issue underneath it,
    // PACKED fp32 instructions do not (measured, tools/probes/valu_rate.py: v_pk_add_f32 between MFMAs costs its own time plus ~12 cycles
    // per switch -- they run on the matrix pipe's lanes).  So: scalar v_sub / v_fma only, and the split's "o - float(bf16 half)" as ONE
    // v_dot2c_f32_bf16 against the constant (-1, 0) / (0, -1) instead of shift / and + subtract.
    float wv[8][4][2];                   // row i of B^T d of this lane's two tiles: [channel][column][tile group]
    const float sgn = wave == 1 ? 1.f : -1.f;
    f32x2 tx_[2][4], ty_[2][4];          // raw values of two channels in flight: [channel parity][tile group * 2 + column pair]
    // the lane's four columns of a staged row start at an EVEN float index (RAW_SHIFT): one ds_read2_b64 per row and tile group
    auto w_read = [&](int ch, int buf, auto part_t) WS_INL {
        constexpr int part = decltype(part_t)::value;               // 0: X rows, 1: Y rows
        const float* src = (part == 0 ? pX : pY) + buf * WS_RAW + ch * WS_PLANE;
#pragma unroll
        for (int tg = 0; tg < 2; ++tg) {
            const f32x2* q = reinterpret_cast<const f32x2*>(__builtin_assume_aligned(src + tg * 4 * WS_IW, 8));
            if (part == 0) { tx_[ch & 1][tg * 2] = q[0]; tx_[ch & 1][tg * 2 + 1] = q[1]; }
            else { ty_[ch & 1][tg * 2] = q[0]; ty_[ch & 1][tg * 2 + 1] = q[1]; }
        }
    };
    auto w_fma = [&](int ch, int half) WS_INL {                      // half = column pair (a constant after inlining)
#pragma unroll
        for (int tg = 0; tg < 2; ++tg) {
            asm("v_fma_f32 %0, %1, %2, %3" : "=v"(wv[ch][2 * half][tg]) : "v"(ty_[ch & 1][tg * 2 + half].x), "v"(sgn), "v"(tx_[ch & 1][tg * 2 + half].x));
            asm("v_fma_f32 %0, %1, %2, %3" : "=v"(wv[ch][2 * half + 1][tg]) : "v"(ty_[ch & 1][tg * 2 + half].y), "v"(sgn), "v"(tx_[ch & 1][tg * 2 + half].y));
        }
    };
    uint4 bop[2][6];                     // B operands of the current / next position: [tile group * 3 + split]
    // the three-way split of one channel pair (2 cp, 2 cp + 1) of position j for both tile groups, in six parts of 2 .. 4 instructions
    float oa[2], ob[2], ta[2], tb[2];    // [tile group]: channel 2 cp / 2 cp + 1 and the fp32 value of their leading bf16 term
    unsigned h[2];
    // first term rounded to nearest (its residual is then zero-mean: so are the three dropped products), second and third by TRUNCATION
    // (v_perm of the high halves: the residual after two 8-bit terms has at most 8 significant bits left, so the third term is exact
    // and the three terms sum to the fp32 value exactly)
    auto build_part = [&](auto j_t, auto cp_t, auto part_t, int set) WS_INL {
        constexpr int j = decltype(j_t)::value, cp = decltype(cp_t)::value, part = decltype(part_t)::value;
        constexpr int ca = 2 * cp, cb = 2 * cp + 1;
        if constexpr (part == 0) {
#pragma unroll
            for (int tg = 0; tg < 2; ++tg) {
                // (asm: the compiler would pair some of these into v_pk_add_f32, which stalls behind the MFMA in flight)
                constexpr int c1 = j == 0 ? 0 : j == 1 ? 1 : j == 2 ? 2 : 1, c2 = j == 0 ? 2 : j == 1 ? 2 : j == 2 ? 1 : 3;
                if constexpr (j == 1) {
                    asm("v_add_f32 %0, %1, %2" : "=v"(oa[tg]) : "v"(wv[ca][c1][tg]), "v"(wv[ca][c2][tg]));
                    asm("v_add_f32 %0, %1, %2" : "=v"(ob[tg]) : "v"(wv[cb][c1][tg]), "v"(wv[cb][c2][tg]));
                } else {
                    asm("v_sub_f32 %0, %1, %2" : "=v"(oa[tg]) : "v"(wv[ca][c1][tg]), "v"(wv[ca][c2][tg]));
                    asm("v_sub_f32 %0, %1, %2" : "=v"(ob[tg]) : "v"(wv[cb][c1][tg]), "v"(wv[cb][c2][tg]));
                }
            }
        } else if constexpr (part == 1) {
            h[0] = cvt2(oa[0], ob[0]); h[1] = cvt2(oa[1], ob[1]);
            set_comp<cp>(bop[set][0], h[0]); set_comp<cp>(bop[set][3], h[1]);
        } else if constexpr (part == 2) {
#pragma unroll
            for (int tg = 0; tg < 2; ++tg) { ta[tg] = __builtin_bit_cast(float, h[tg] << 16); tb[tg] = __builtin_bit_cast(float, h[tg] & 0xffff0000u); }
        } else if constexpr (part == 3) {
#pragma unroll
            for (int tg = 0; tg < 2; ++tg) {
                asm("v_sub_f32 %0, %0, %1" : "+v"(oa[tg]) : "v"(ta[tg]));
                asm("v_sub_f32 %0, %0, %1" : "+v"(ob[tg]) : "v"(tb[tg]));
            }
        } else if constexpr (part == 4) {
#pragma unroll
            for (int tg = 0; tg < 2; ++tg) {
                set_comp<cp>(bop[set][3 * tg + 1], __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, ob[tg]), __builtin_bit_cast(unsigned, oa[tg]), 0x07060302u));
                ta[tg] = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, oa[tg]) & 0xffff0000u);
                tb[tg] = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, ob[tg]) & 0xffff0000u);
            }
        } else {
#pragma unroll
            for (int tg = 0; tg < 2; ++tg) {
                asm("v_sub_f32 %0, %0, %1" : "+v"(oa[tg]) : "v"(ta[tg]));
                asm("v_sub_f32 %0, %0, %1" : "+v"(ob[tg]) : "v"(tb[tg]));
                set_comp<cp>(bop[set][3 * tg + 2], __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, ob[tg]), __builtin_bit_cast(unsigned, oa[tg]), 0x07060302u));
            }
        }
    };
    auto build_all = [&](auto j_t, int set) WS_INL {       // prologue only
        using I0_ = std::integral_constant<int, 0>; using I1_ = std::integral_constant<int, 1>; using I2_ = std::integral_constant<int, 2>;
        using I3_ = std::integral_constant<int, 3>; using I4_ = std::integral_constant<int, 4>; using I5_ = std::integral_constant<int, 5>;
        auto one = [&](auto cp_t) WS_INL { build_part(j_t, cp_t, I0_{}, set); build_part(j_t, cp_t, I1_{}, set); build_part(j_t, cp_t, I2_{}, set);
                                           build_part(j_t, cp_t, I3_{}, set); build_part(j_t, cp_t, I4_{}, set); build_part(j_t, cp_t, I5_{}, set); };
        one(I0_{}); one(I1_{}); one(I2_{}); one(I3_{});
    };

    f32x16 acc[16];                      // [position j][m group][tile group]
#pragma unroll
    for (int p = 0; p < 16; ++p)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[p][e] = 0.f;

    // ---- prologue ----
    using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;
    dma_raw(0, 0);
    if (nch > 1) dma_raw(1, 1);
    dma_u(0, 0); dma_u(0, 1); dma_u(0, 2); dma_u(0, 3);
    __builtin_amdgcn_s_waitcnt(0);       // everything landed (prologue only)
    __syncthreads();
    a_fetch(0, 0);
#pragma unroll
    for (int ch = 0; ch < 8; ++ch) { w_read(ch, 0, I0{}); w_read(ch, 0, I1{}); w_fma(ch, 0); w_fma(ch, 1); }
    build_all(I0{}, 0);

    unsigned long long t0 = 0, r0 = 0;
    if (STAMP) { t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }

    // six partial products, smallest first: (U split, V split)
    constexpr int PA[6] = {2, 0, 1, 1, 0, 0};
    constexpr int PB[6] = {0, 2, 1, 0, 1, 0};

    // One step = one position j of a chunk = 24 MFMA slots on four accumulators (a dependent MFMA is four instructions away: with two
    // accumulators 48 cycles per MFMA were measured instead of 32).  Everything else is dealt to the slots by hand and pinned there
    // (sched_barrier): a bf16 MFMA occupies the matrix pipe for 32 cycles = 8 issue slots, the auxiliary work is <= 7 instructions per slot:
    //   slot 0          this step's ring slot is refilled (position j of the NEXT chunk), the next step's A operands are read
    //   slots 6 g + p   part p of the three-way split of channel pair g of the NEXT position's B operands
    //   steps 2 and 3   the next chunk's row transform, channel by channel, right behind the last use of the old values: reads two
    //                   slots ahead of the v_fma that consume them
    auto step = [&](const int c, auto j_t, auto more_t) WS_INL {
        constexpr int j = decltype(j_t)::value;
        constexpr bool MORE = decltype(more_t)::value;
        constexpr int set = j & 1;
        using JN = std::integral_constant<int, (j + 1) & 3>;
        const int nbuf = (c + 1) & 1;
        auto aux = [&](auto k_t) WS_INL {
            constexpr int k = decltype(k_t)::value;
            if (!(VAR & 1)) {
                // slots 0 .. 5: this step's ring slot is refilled (its reads, a step ago, have returned); slot 6: wait for the next step's
                // slot -- its DMA was issued three steps ago, 6 per step since (the raw pieces in between only make the wait stricter);
                // slots 6 .. 11: the next step's A operands
                if constexpr (MORE && k < 6) {
                    if constexpr (k == 0) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    dma_u_one(c + 1, j, k);
                }
                if constexpr (k == 6) {
                    if (MORE) asm volatile("s_waitcnt vmcnt(18)" ::: "memory");
                    else if (j == 0) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
                    else if (j == 1) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
                    else if (j == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                if constexpr (k >= 6 && k < 12 && (MORE || j < 3)) a_fetch_one((j + 1) & 3, set ^ 1, k - 6);
            }
            if constexpr (MORE && (j == 2 || j == 3) && !(VAR & 4)) {
                // the next chunk's row transform: step 2 channels 0 .. 3, step 3 channels 4 .. 7; channel c0 + n reads X in slot 4 n + 1, Y in
                // 4 n + 2 (two ds_read2_b64 each), v_fma in slots 4 n + 4 and 4 n + 5; the old values of channels 2 g, 2 g + 1 were last used in
                // slot 6 g of step 2, the new ones are first used in slot 6 g of step 3 (channel pair g)
                constexpr int c0 = j == 2 ? 0 : 4;
                if constexpr (k >= 1 && k <= 17) {
                    constexpr int n = (k - 1) / 4, ph = (k - 1) % 4;
                    if constexpr (ph == 0) w_read(c0 + n, nbuf, I0{});
                    if constexpr (ph == 1) w_read(c0 + n, nbuf, I1{});
                    if constexpr (ph == 3) w_fma(c0 + n, 0);
                    if constexpr (ph == 0 && n > 0) w_fma(c0 + n - 1, 1);
                    if constexpr (k == 17) w_fma(c0 + 3, 1);
                }
            }
            if (!(VAR & 2) && (MORE || j < 3))
                build_part(JN{}, std::integral_constant<int, k / 6>{}, std::integral_constant<int, k % 6>{}, set ^ 1);
        };
        auto slots = [&](auto k_t, auto&& self) WS_INL {
            constexpr int k = decltype(k_t)::value;
            if constexpr (k < 24) {
                constexpr int p = k >> 2, mg = (k >> 1) & 1, tg = k & 1;
                acc[j * 4 + mg * 2 + tg] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf(aop[(VAR & 1) ? 0 : set][mg * 3 + PA[p]]), as_bf(bop[set][tg * 3 + PB[p]]),
                                                                                 acc[j * 4 + mg * 2 + tg], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                aux(k_t);
                __builtin_amdgcn_sched_barrier(0);
                self(std::integral_constant<int, k + 1>{}, self);
            }
        };
        slots(I0{}, slots);
    };
    auto chunk = [&](const int c, auto more_t, auto more2_t) WS_INL {
        constexpr bool MORE = decltype(more_t)::value, MORE2 = decltype(more2_t)::value;
        // raw(c + 1) has landed in every wave (its DMA is older than the 18 newest of the last step's wait), and every wave is done
        // reading raw(c) (steps 2 and 3 of the previous chunk): buffer c & 1 is free for raw(c + 2)
        if (MORE) asm volatile("s_barrier" ::: "memory");
        if (MORE2) dma_raw(c + 2, c & 1);
        step(c, I0{}, more_t); step(c, I1{}, more_t); step(c, I2{}, more_t); step(c, I3{}, more_t);
    };
    {
        using T = std::true_type; using F = std::false_type;
        int c = 0;
        for (; c + 2 < nch; ++c) chunk(c, T{}, T{});
        if (c + 1 < nch) { chunk(c, T{}, F{}); ++c; }
        chunk(c, F{}, F{});
    }

    unsigned long long t1 = 0, r1 = 0;
    if (STAMP) { t1 = __builtin_amdgcn_s_memtime(); r1 = __builtin_amdgcn_s_memrealtime(); }

    // ---- epilogue: in-wave column transform t = M A (2 values per row), rows combined through LDS, combo q finished by wave q ----
    float2* const xch = reinterpret_cast<float2*>(lds);             // [src wave 4][combo 4][e 8][lane 64]
    const int mg_f = wave >> 1, tg_f = wave & 1;                    // the combo this wave finishes
    const int gy0 = y0 + 4 * tg_f + 2 * (t31 >> 4), gx = x0 + 2 * (t31 & 15);
    const unsigned pix0 = (unsigned)gy0 * a.W + gx;
    const int m_base = mt * 64 + mg_f * 32 + 4 * kq;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        __syncthreads();                                            // raw images / the previous half's exchange are dead
#pragma unroll
        for (int qq = 0; qq < 4; ++qq)
#pragma unroll
            for (int el = 0; el < 8; ++el) {
                const int e = 8 * half + el;
                const float m0 = acc[0 * 4 + qq][e], m1 = acc[1 * 4 + qq][e], m2 = acc[2 * 4 + qq][e], m3 = acc[3 * 4 + qq][e];
                xch[((wave * 4 + qq) * 8 + el) * 64 + lane] = make_float2(m0 + m1 + m2, m1 - m2 - m3);
            }
        __syncthreads();
#pragma unroll
        for (int el = 0; el < 8; ++el) {
            const int e = 8 * half + el;
            const float2 s0 = xch[((0 * 4 + wave) * 8 + el) * 64 + lane], s1 = xch[((1 * 4 + wave) * 8 + el) * 64 + lane];
            const float2 s2 = xch[((2 * 4 + wave) * 8 + el) * 64 + lane], s3 = xch[((3 * 4 + wave) * 8 + el) * 64 + lane];
            float y00 = s0.x + s1.x + s2.x, y01 = s0.y + s1.y + s2.y;
            float y10 = s1.x - s2.x - s3.x, y11 = s1.y - s2.y - s3.y;
            const int m = m_base + 8 * (e >> 2) + (e & 3);
            const float b = a.bias ? a.bias[m] : 0.f;
            y00 += b; y01 += b; y10 += b; y11 += b;
            if (a.relu) { y00 = fmaxf(y00, 0.f); y01 = fmaxf(y01, 0.f); y10 = fmaxf(y10, 0.f); y11 = fmaxf(y11, 0.f); }
            float* dst = a.out + (size_t)m * plane + pix0;
            *reinterpret_cast<float2*>(dst) = make_float2(y00, y01);
            *reinterpret_cast<float2*>(dst + a.W) = make_float2(y10, y11);
        }
    }
    if (STAMP && tid == 0 && a.stamps) {
        const unsigned long long t2 = __builtin_amdgcn_s_memtime();
        a.stamps[4 * blockIdx.x] = t1 - t0; a.stamps[4 * blockIdx.x + 1] = r1 - r0;
        a.stamps[4 * blockIdx.x + 2] = t0 - t_begin; a.stamps[4 * blockIdx.x + 3] = t2 - t1;
    }
}

// ---- host side ----
unsigned short f2bf(float f)
{
    unsigned u;
    memcpy(&u, &f, 4);
    return (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}
float bf2f(unsigned short h) { unsigned u = (unsigned)h << 16; float f; memcpy(&f, &u, 4); return f; }

size_t ws_pack_elems(int K, int M) { return (size_t)(M / 64) * (K / 16) * 16 * 2 * 3 * 64 * 8; }

void ws_pack(const float* w /*M, K, 3, 3*/, int M, int K, unsigned short* dst)
{
    static const double G[4][3] = {{1, 0, 0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0, 0, 1}};
    const int nch = K / 16;
    for (int m = 0; m < M; ++m)
        for (int k = 0; k < K; ++k) {
            double g[3][3], t[4][3];
            for (int tap = 0; tap < 9; ++tap) g[tap / 3][tap % 3] = w[((size_t)m * K + k) * 9 + tap];
            for (int i = 0; i < 4; ++i)
                for (int j = 0; j < 3; ++j) t[i][j] = G[i][0] * g[0][j] + G[i][1] * g[1][j] + G[i][2] * g[2][j];
            for (int i = 0; i < 4; ++i)
                for (int j = 0; j < 4; ++j) {
                    const float u = (float)(t[i][0] * G[j][0] + t[i][1] * G[j][1] + t[i][2] * G[j][2]);
                    const unsigned short h1 = f2bf(u);
                    const float r1 = u - bf2f(h1);
                    const unsigned short h2 = f2bf(r1);
                    const unsigned short h3 = f2bf(r1 - bf2f(h2));
                    const unsigned short hs[3] = {h1, h2, h3};
                    const int mtile = m / 64, mg = (m % 64) / 32, lane = (m % 32) + 32 * ((k % 16) / 8), jj = k % 8, pos = 4 * i + j;
                    for (int s = 0; s < 3; ++s)
                        dst[((((((size_t)mtile * nch + k / 16) * 16 + pos) * 2 + mg) * 3 + s) * 64 + lane) * 8 + jj] = hs[s];
                }
        }
}

thread_local char g_err[512] = "";

}  // namespace

#define WS_TRY(expr)                                                                                        \
    do {                                                                                                    \
        hipError_t e_ = (expr);                                                                             \
        if (e_ != hipSuccess) { snprintf(g_err, sizeof g_err, "%s: %s", #expr, hipGetErrorString(e_)); return 3; } \
    } while (0)

extern "C" const char* st_probe_wino_split_error(void) { return g_err; }

// One forward layer K -> M at H x W (bias + ReLU).  check != 0: compare with a CPU loop nest (double accumulation) on the whole output
// (keep the shape small).  Returns the average launch time, the relative L2 error, and the median block's main-loop cycles / clock.
extern "C" int st_probe_wino_split(int device_id, int K, int M, int H, int W, int iters, int check_or_variant, double* avg_ms, double* rel_l2,
                                   double* loop_cycles, double* clock_mhz, double* pro_cycles, double* epi_cycles)
{
    if (K <= 0 || K % 16 || M <= 0 || M % 64 || H <= 0 || H % 8 || W <= 0 || W % 32 || iters <= 0) { snprintf(g_err, sizeof g_err, "shape not supported by the probe"); return 1; }
    WS_TRY(hipSetDevice(device_id));
    const int check = check_or_variant == 1, var = check_or_variant >= 100 ? check_or_variant - 100 : 0;     // 100 + VAR: a loop experiment
    const size_t n_in = (size_t)K * H * W, n_out = (size_t)M * H * W;
    std::vector<float> w((size_t)M * K * 9), hin(n_in), hb(M);
    uint32_t st = 12345u;
    auto rnd = [&]() { st = st * 1664525u + 1013904223u; return ((st >> 8) & 0xffff) / 32768.0f - 1.0f; };
    for (auto& x : w) x = rnd() * 0.05f;
    for (auto& x : hin) x = rnd();
    for (auto& x : hb) x = rnd() * 0.1f;
    std::vector<unsigned short> pk(ws_pack_elems(K, M), 0);
    ws_pack(w.data(), M, K, pk.data());
    float *din = nullptr, *db = nullptr, *dout = nullptr;
    void* dpk = nullptr;
    unsigned long long* dst = nullptr;
    WS_TRY(hipMalloc((void**)&din, n_in * 4)); WS_TRY(hipMalloc((void**)&db, M * 4)); WS_TRY(hipMalloc((void**)&dout, n_out * 4));
    WS_TRY(hipMalloc(&dpk, pk.size() * 2));
    WS_TRY(hipMemcpy(din, hin.data(), n_in * 4, hipMemcpyHostToDevice));
    WS_TRY(hipMemcpy(db, hb.data(), M * 4, hipMemcpyHostToDevice));
    WS_TRY(hipMemcpy(dpk, pk.data(), pk.size() * 2, hipMemcpyHostToDevice));
    WsArgs a{};
    a.in = din; a.upack = (const uint4*)dpk; a.bias = db; a.out = dout;
    a.K = K; a.M = M; a.H = H; a.W = W; a.nch = K / 16; a.tiles_x = W / 32; a.tiles_y = H / 8; a.relu = 1;
    a.in_bytes = (unsigned)(n_in * 4); a.u_bytes = (unsigned)(pk.size() * 2);
    const int blocks = a.tiles_x * a.tiles_y * (M / 64);
    WS_TRY(hipMalloc((void**)&dst, (size_t)blocks * 32));
    WS_TRY(hipMemset(dst, 0, (size_t)blocks * 32));
    a.stamps = dst;
    hipStream_t s;
    WS_TRY(hipStreamCreate(&s));
    hipEvent_t e0, e1;
    WS_TRY(hipEventCreate(&e0)); WS_TRY(hipEventCreate(&e1));
    auto launch = [&](bool stamp) {
        switch (var * 2 + (stamp ? 1 : 0)) {
#define WS_CASE(V) case V * 2: wino_split_probe_k<0, V><<<blocks, 256, 0, s>>>(a); break; case V * 2 + 1: wino_split_probe_k<1, V><<<blocks, 256, 0, s>>>(a); break;
        WS_CASE(0) WS_CASE(1) WS_CASE(2) WS_CASE(3) WS_CASE(4) WS_CASE(6) WS_CASE(7)
#undef WS_CASE
        default: break;
        }
    };
    for (int i = 0; i < 2; ++i) launch(false);
    WS_TRY(hipGetLastError());
    WS_TRY(hipStreamSynchronize(s));
    WS_TRY(hipEventRecord(e0, s));
    for (int i = 0; i < iters; ++i) launch(false);
    WS_TRY(hipEventRecord(e1, s));
    WS_TRY(hipStreamSynchronize(s));
    float ms = 0.f;
    WS_TRY(hipEventElapsedTime(&ms, e0, e1));
    if (avg_ms) *avg_ms = ms / iters;
    // stamped launches: main-loop cycles and the shader clock inside it (median block)
    for (int i = 0; i < 3; ++i) launch(true);
    WS_TRY(hipStreamSynchronize(s));
    {
        std::vector<unsigned long long> h((size_t)blocks * 4);
        WS_TRY(hipMemcpy(h.data(), dst, (size_t)blocks * 32, hipMemcpyDeviceToHost));
        std::vector<double> cyc, clk, pro, epi;
        for (int b = 0; b < blocks; ++b)
            if (h[4 * b + 1]) { cyc.push_back((double)h[4 * b]); clk.push_back((double)h[4 * b] / (double)h[4 * b + 1] * 100.0); pro.push_back((double)h[4 * b + 2]); epi.push_back((double)h[4 * b + 3]); }
        auto med = [](std::vector<double>& v) { if (v.empty()) return 0.0; std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
        if (loop_cycles) *loop_cycles = med(cyc);
        if (clock_mhz) *clock_mhz = med(clk);
        if (pro_cycles) *pro_cycles = med(pro);
        if (epi_cycles) *epi_cycles = med(epi);
    }
    if (rel_l2) *rel_l2 = -1.0;
    if (check) {
        std::vector<float> hout(n_out);
        WS_TRY(hipMemcpy(hout.data(), dout, n_out * 4, hipMemcpyDeviceToHost));
        double num = 0, den = 0;
        for (int m = 0; m < M; ++m)
            for (int y = 0; y < H; ++y)
                for (int x = 0; x < W; ++x) {
                    double s2 = hb[m];
                    for (int k = 0; k < K; ++k)
                        for (int dy = 0; dy < 3; ++dy) {
                            const int yy = y + dy - 1;
                            if (yy < 0 || yy >= H) continue;
                            for (int dx = 0; dx < 3; ++dx) {
                                const int xx = x + dx - 1;
                                if (xx < 0 || xx >= W) continue;
                                s2 += (double)w[((size_t)m * K + k) * 9 + dy * 3 + dx] * hin[((size_t)k * H + yy) * W + xx];
                            }
                        }
                    if (s2 < 0) s2 = 0;
                    const double d = (double)hout[((size_t)m * H + y) * W + x] - s2;
                    num += d * d; den += s2 * s2;
                }
        if (rel_l2) *rel_l2 = sqrt(num / std::max(den, 1e-300));
    }
    (void)hipFree(din); (void)hipFree(db); (void)hipFree(dout); (void)hipFree(dpk); (void)hipFree(dst);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); (void)hipStreamDestroy(s);
    return 0;
}

// ---- instruction-rate probe: shader cycles per VALU instruction of one kind (eight independent chains, one wave per SIMD), alone and
// with NV of them between consecutive v_mfma_f32_32x32x16_bf16 (four accumulators in rotation) ----
namespace {
template <int KIND>
__device__ __forceinline__ void valu_op(f32x2& x, unsigned& u, float s)
{
    if constexpr (KIND == 0) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x.x) : "v"(s));
    else if constexpr (KIND == 1) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(x) : "v"(x));
    else if constexpr (KIND == 2) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(u) : "v"(x.x), "v"(x.y));
    else if constexpr (KIND == 3) asm volatile("v_lshlrev_b32 %0, 16, %0" : "+v"(u));
    else if constexpr (KIND == 4) asm volatile("v_and_b32 %0, 0xffff0000, %0" : "+v"(u));
    else if constexpr (KIND == 5) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(x.x) : "v"(s));
    else if constexpr (KIND == 6) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(u) : "v"(x.x), "v"(0x07060302u));
    else if constexpr (KIND == 7) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(x) : "v"(x));
    else if constexpr (KIND == 8) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(x) : "v"(x));
    else if constexpr (KIND == 9) asm volatile("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(x.x) : "v"(u), "v"(0xbf80u));
    else if constexpr (KIND == 10) asm volatile("ds_read_b32 %0, %1" : "=v"(x.x) : "v"(u & 0x3ffcu));
    else if constexpr (KIND == 11) asm volatile("ds_read_b64 %0, %1" : "=v"(x) : "v"(u & 0x3ff8u));
    else if constexpr (KIND == 12) asm volatile("ds_read2_b32 %0, %1 offset0:3 offset1:4" : "=v"(x) : "v"(u & 0x3ffcu));
    else if constexpr (KIND == 13) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(x.x) : "v"(x.y));
    else if constexpr (KIND == 14) asm volatile("v_bfe_u32 %0, %0, 16, 1" : "+v"(u));
}
template <int KIND, int NV, bool MFMA>
__global__ __launch_bounds__(256, 1) void rate_probe_k(float* out, unsigned long long* cyc, int iters, float seed)
{
    f32x16 acc[4];
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[p][e] = 0.f;
    f32x2 x[8];
    unsigned u[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { x[i].x = seed * (i + 1); x[i].y = seed + i; u[i] = threadIdx.x * 2654435761u + i; }
    uint4 av = make_uint4(0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u), bv = av;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            if constexpr (MFMA) {
                acc[k & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf(av), as_bf(bv), acc[k & 3], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int v = 0; v < NV; ++v) valu_op<KIND>(x[(k * NV + v) & 7], u[(k * NV + v) & 7], seed);
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (KIND >= 10 && KIND <= 12) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float r = 0.f;
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int e = 0; e < 16; ++e) r += acc[p][e];
#pragma unroll
    for (int i = 0; i < 8; ++i) r += x[i].x + x[i].y + __builtin_bit_cast(float, u[i] & 0x3fffffffu);
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = r;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
}  // namespace

// kind: 0 v_add_f32, 1 v_pk_add_f32, 2 v_cvt_pk_bf16_f32, 3 v_lshlrev_b32, 4 v_and_b32, 5 v_fma_f32, 6 v_perm_b32, 7 v_pk_fma_f32, 8 v_pk_mul_f32.
// nv: instructions per group (4 or 8), with_mfma: a bf16 MFMA in front of every group.  Returns shader cycles per GROUP (median block).
extern "C" int st_probe_valu_rate(int device_id, int kind, int nv, int with_mfma, double* cycles_per_group)
{
    WS_TRY(hipSetDevice(device_id));
    const int blocks = 256, iters = 2000;
    float* out = nullptr;
    unsigned long long* cyc = nullptr;
    WS_TRY(hipMalloc((void**)&out, (size_t)blocks * 256 * 4));
    WS_TRY(hipMalloc((void**)&cyc, blocks * 8));
    bool ok = true;
#define RP(K, N, M) if (kind == K && nv == N && with_mfma == M) rate_probe_k<K, N, (M != 0)><<<blocks, 256>>>(out, cyc, iters, 0.37f); else
#define RPK(K) RP(K, 4, 0) RP(K, 8, 0) RP(K, 0, 1) RP(K, 2, 1) RP(K, 4, 1) RP(K, 6, 1) RP(K, 8, 1)
    for (int rep = 0; rep < 2; ++rep) {
        RPK(0) RPK(1) RPK(2) RPK(3) RPK(4) RPK(5) RPK(6) RPK(7) RPK(8) RPK(9) RPK(10) RPK(11) RPK(12) RPK(13) RPK(14) { ok = false; }
    }
#undef RPK
#undef RP
    if (!ok) { snprintf(g_err, sizeof g_err, "no such rate probe variant"); return 1; }
    WS_TRY(hipDeviceSynchronize());
    std::vector<unsigned long long> h(blocks);
    WS_TRY(hipMemcpy(h.data(), cyc, blocks * 8, hipMemcpyDeviceToHost));
    std::sort(h.begin(), h.end());
    if (cycles_per_group) *cycles_per_group = (double)h[blocks / 2] / ((double)iters * 16);
    (void)hipFree(out); (void)hipFree(cyc);
    return 0;
}

// ---- the PRODUCT kernel (style_transfer2_amd/csrc/conv3x3_wino_split.hip) through its launcher: timing, stamps, check of the forward ----
#include "../../style_transfer2_amd/csrc/st2_kernels.h"
// mode: 0 forward (bias + ReLU), 1 data gradient (ReLU mask + injected diff), 2 forward + fused pool + arg-max map, 3 = 2 without the blob
extern "C" int st_probe_wino_split_product(int device_id, int K, int M, int H, int W, int iters, int mode, int check, double* avg_ms, double* rel_l2,
                                           double* loop_cycles, double* clock_mhz, double* pro_cycles, double* epi_cycles)
{
    using namespace st2;
    if (!conv_wino_split_ok(K, M, H, W) || iters <= 0) { snprintf(g_err, sizeof g_err, "shape not supported"); return 1; }
    WS_TRY(hipSetDevice(device_id));
    const size_t n_in = (size_t)K * H * W, n_out = (size_t)M * H * W;
    std::vector<float> w((size_t)M * K * 9), hin(n_in), hb(M);
    uint32_t st = 12345u;
    auto rnd = [&]() { st = st * 1664525u + 1013904223u; return ((st >> 8) & 0xffff) / 32768.0f - 1.0f; };
    for (auto& x : w) x = rnd() * 0.05f;
    for (auto& x : hin) x = rnd();
    for (auto& x : hb) x = rnd() * 0.1f;
    std::vector<unsigned short> pk(wino_split_pack_elems(K, M), 0);
    const int want = mode >= 4 ? mode - 4 : 3;      // modes 4 .. 7: data gradient with bit 0 = mask, bit 1 = inject
    if (mode >= 4) mode = 1;
    if (mode == 1) pack_wino_split_weights_dgrad(w.data(), K, M, pk.data()); else pack_wino_split_weights_fwd(w.data(), M, K, pk.data());
    float *din = nullptr, *db = nullptr, *dout = nullptr, *dmask = nullptr, *dinj = nullptr, *dpool = nullptr;
    unsigned char* damap = nullptr;
    void* dpk = nullptr;
    unsigned long long* dst = nullptr;
    WS_TRY(hipMalloc((void**)&din, n_in * 4)); WS_TRY(hipMalloc((void**)&db, M * 4)); WS_TRY(hipMalloc((void**)&dout, n_out * 4));
    WS_TRY(hipMalloc(&dpk, pk.size() * 2));
    WS_TRY(hipMemcpy(din, hin.data(), n_in * 4, hipMemcpyHostToDevice));
    WS_TRY(hipMemcpy(db, hb.data(), M * 4, hipMemcpyHostToDevice));
    WS_TRY(hipMemcpy(dpk, pk.data(), pk.size() * 2, hipMemcpyHostToDevice));
    ConvProblem p{};
    p.in = din; p.wpack = (const float*)dpk; p.bias = mode == 1 ? nullptr : db; p.out = dout; p.K = K; p.M = M; p.H = H; p.W = W; p.relu = mode == 1 ? 0 : 1;
    if (mode == 1) {
        WS_TRY(hipMalloc((void**)&dmask, n_out * 4)); WS_TRY(hipMalloc((void**)&dinj, n_out * 4));
        for (size_t off = 0; off < n_out; off += n_in) {
            const size_t n = std::min(n_in, n_out - off);
            WS_TRY(hipMemcpy(dmask + off, din, n * 4, hipMemcpyDeviceToDevice)); WS_TRY(hipMemcpy(dinj + off, din, n * 4, hipMemcpyDeviceToDevice));
        }
        p.mask_src = (want & 1) ? dmask : nullptr; p.inject = (want & 2) ? dinj : nullptr;
    }
    if (mode >= 2) {
        const size_t np = (size_t)M * ((H + 1) / 2) * ((W + 1) / 2);
        WS_TRY(hipMalloc((void**)&dpool, np * 4)); WS_TRY(hipMalloc((void**)&damap, np));
        p.pool_out = dpool; p.pool_amap = damap;
        if (mode == 3) p.out = nullptr;
    }
    const int blocks = ((W + 31) / 32) * ((H + 7) / 8) * (M / 64);
    hipStream_t s;
    WS_TRY(hipStreamCreate(&s));
    hipEvent_t e0, e1;
    WS_TRY(hipEventCreate(&e0)); WS_TRY(hipEventCreate(&e1));
    for (int i = 0; i < 2; ++i) WS_TRY(launch_conv3x3_wino_split(p, s));
    WS_TRY(hipStreamSynchronize(s));
    WS_TRY(hipEventRecord(e0, s));
    for (int i = 0; i < iters; ++i) (void)launch_conv3x3_wino_split(p, s);
    WS_TRY(hipEventRecord(e1, s));
    WS_TRY(hipStreamSynchronize(s));
    float ms = 0.f;
    WS_TRY(hipEventElapsedTime(&ms, e0, e1));
    if (avg_ms) *avg_ms = ms / iters;
    if (rel_l2) *rel_l2 = -1.0;
    if (check && mode == 0) {
        std::vector<float> hout(n_out);
        WS_TRY(hipMemcpy(hout.data(), dout, n_out * 4, hipMemcpyDeviceToHost));
        double num = 0, den = 0;
        for (int m = 0; m < M; ++m)
            for (int y = 0; y < H; ++y)
                for (int x = 0; x < W; ++x) {
                    double s2 = hb[m];
                    for (int k = 0; k < K; ++k)
                        for (int dy = 0; dy < 3; ++dy) {
                            const int yy = y + dy - 1;
                            if (yy < 0 || yy >= H) continue;
                            for (int dx = 0; dx < 3; ++dx) {
                                const int xx = x + dx - 1;
                                if (xx < 0 || xx >= W) continue;
                                s2 += (double)w[((size_t)m * K + k) * 9 + dy * 3 + dx] * hin[((size_t)k * H + yy) * W + xx];
                            }
                        }
                    if (s2 < 0) s2 = 0;
                    const double d = (double)hout[((size_t)m * H + y) * W + x] - s2;
                    num += d * d; den += s2 * s2;
                }
        if (rel_l2) *rel_l2 = sqrt(num / std::max(den, 1e-300));
    }
    if (check && mode == 1) {      // data gradient: out[m] = (mask > 0 ? sum_k sum_taps w[k][m][8 - tap] in[k] : 0) + inject   (K = Cout, M = Cin of w)
        std::vector<float> hout(n_out), hm(n_out);
        WS_TRY(hipMemcpy(hout.data(), dout, n_out * 4, hipMemcpyDeviceToHost));
        WS_TRY(hipMemcpy(hm.data(), dmask, n_out * 4, hipMemcpyDeviceToHost));
        double num = 0, den = 0;
        for (int m = 0; m < M; ++m)
            for (int y = 0; y < H; ++y)
                for (int x = 0; x < W; ++x) {
                    double s2 = 0;
                    for (int k = 0; k < K; ++k)
                        for (int dy = 0; dy < 3; ++dy) {
                            const int yy = y + dy - 1;
                            if (yy < 0 || yy >= H) continue;
                            for (int dx = 0; dx < 3; ++dx) {
                                const int xx = x + dx - 1;
                                if (xx < 0 || xx >= W) continue;
                                s2 += (double)w[((size_t)k * M + m) * 9 + (8 - (dy * 3 + dx))] * hin[((size_t)k * H + yy) * W + xx];
                            }
                        }
                    const size_t o = ((size_t)m * H + y) * W + x;
                    if ((want & 1) && !(hm[o] > 0)) s2 = 0;
                    if (want & 2) s2 += hm[o];                      // inject = the same data as the mask
                    const double d = (double)hout[o] - s2;
                    num += d * d; den += s2 * s2;
                }
        if (rel_l2) *rel_l2 = sqrt(num / std::max(den, 1e-300));
    }
    if (mode <= 1) {       // stamped launches (one pass, blob written)
        WS_TRY(hipMalloc((void**)&dst, (size_t)blocks * 64));
        WS_TRY(hipMemset(dst, 0, (size_t)blocks * 64));
        p.stamps = dst;
        for (int i = 0; i < 3; ++i) WS_TRY(launch_conv3x3_wino_split(p, s));
        WS_TRY(hipStreamSynchronize(s));
        std::vector<unsigned long long> h((size_t)blocks * 8);
        WS_TRY(hipMemcpy(h.data(), dst, (size_t)blocks * 64, hipMemcpyDeviceToHost));
        std::vector<double> cyc, clk, pro, epi;
        for (int b = 0; b < blocks; ++b) {
            const unsigned long long* q = &h[8 * b];
            if (q[4] > q[2]) { cyc.push_back((double)(q[3] - q[1])); clk.push_back((double)(q[3] - q[1]) / (double)(q[4] - q[2]) * 100.0); pro.push_back((double)(q[1] - q[0])); epi.push_back((double)(q[5] - q[3])); }
        }
        auto med = [](std::vector<double>& v) { if (v.empty()) return 0.0; std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
        if (loop_cycles) *loop_cycles = med(cyc);
        if (clock_mhz) *clock_mhz = med(clk);
        if (pro_cycles) *pro_cycles = med(pro);
        if (epi_cycles) *epi_cycles = med(epi);
    }
    (void)hipFree(din); (void)hipFree(db); (void)hipFree(dout); (void)hipFree(dpk); (void)hipFree(dst); (void)hipFree(dmask); (void)hipFree(dinj); (void)hipFree(dpool); (void)hipFree(damap);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); (void)hipStreamDestroy(s);
    return 0;
}
