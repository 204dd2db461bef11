// Split-operand Winograd F(2x2,3x3) conv3x3 on the bf16 matrix cores (gfx950): forward and data gradient of the VGG convs with fp32
// results (reference: pycaffe Convolution forward / backward behind worker.py:84-86 and :100-106), opt-in: st_set_conv_algo(ctx, 2).
//
//   Y = A^T [ sum_k (G g_k G^T) (.) (B^T d_k B) ] A        as conv3x3_winograd.hip, but every transform-domain product u v of two fp32
// numbers is taken as the six bf16 x bf16 partial products of weight <= 2 of three-way split operands (u = u1 + u2 + u3, v = v1 + v2 + v3
// exactly, 8 + 8 + 8 significand bits):
//   u v = u1 v1 + (u1 v2 + u2 v1) + (u1 v3 + u3 v1 + u2 v2)   [+ three terms below 2^-24 |u v|],
// each product exact in fp32, accumulated in fp32 by v_mfma_f32_32x32x16_bf16, smallest first: 12 matrix-pipe cycles per k instead of the
// 32 of v_mfma_f32_32x32x2_f32.  What conv3x3_first_split.hip does for conv1_1, in the Winograd domain; measured against a double
// loop nest the result is CLOSER than the IEEE-fp32 Winograd kernel's (1e-7 .. 2e-7 against 3e-7 .. 6e-7: the fp32 kernel rounds every
// product to 24 bits before it adds it, this one adds exact products).  U is split on the host (3x the bf16 image = 1.5x the fp32 one), V in
// the input transform.  Go / no-go record of the design: tools/probes/wino_split_probe.hip, profiles/r05_*_wino_split_probe*.txt.
//
// Workgroup = 4 waves (one per SIMD) = 64 output channels x (8 rows x 32 columns) = 64 tiles of 2x2 outputs.  Wave i owns ROW i of the
// 4x4 transform-domain matrix (positions 4 i .. 4 i + 3) for the whole block: 4 positions x 2 channel groups x 2 tile groups = 16
// accumulators of 32x32 (256 AGPRs).  Per 16 input channels (= the k of one MFMA; a "chunk") a wave issues 4 "steps" (one per position)
// of 24 MFMAs (6 partial products x 4 accumulators: a dependent MFMA is four instructions away).
//   * U: [m tile][chunk][pos 16][m group 2][split 3][lane 64] x 16 bytes = the A fragments as they sit in the registers; every fragment is
//     needed by exactly one wave.  L2 -> LDS by LDS-DMA into a wave-private ring of four slots (one per position, 6 KiB each), refilled four
//     steps ahead, read as ds_read_b128 a step ahead: no barrier ever concerns it (~26 B/clk/CU from the XCD's L2 when the loop is
//     MFMA-bound: the blocks of one XCD walk the same channel slice, XCD-aware block map as in the fp32 kernel).
//   * V never touches LDS.  Row i of B^T d B needs two raw rows only, and the lane that builds it is the lane that feeds it to the matrix
//     core: lane (tile t, k half) holds, for its tile in both tile groups, row i of B^T d of its 8 channels (64 registers, computed once per
//     chunk from the raw LDS image: one v_fma each); per step it forms the position's column combination, splits it three ways and packs
//     channel pairs: the packed registers ARE the B operands.  The waves share only the raw image (one s_barrier per 96 MFMAs).
//   * VALU budget: plain VALU instructions (~5 cycles) issue underneath a bf16 MFMA (32 cycles), PACKED fp32 and dot2 instructions do not
//     (tools/probes/valu_rate.py: they run on the matrix pipe's lanes and cost their own time plus ~12 cycles per switch) -- hence scalar
//     v_sub / v_fma through inline asm where the compiler would pair them.  First term rounded to nearest (its residual is zero-mean, so
//     are the dropped products), second and third by truncation (v_perm of the high halves; the third is exact: 24 - 16 bits are left).
//   * everything but the MFMAs is dealt to the 24 slots of a step by hand and pinned there (sched_barrier).
//   * raw activations: LDS-DMA, 16 channels x 10 rows x 40 floats per chunk as aligned quads, hardware zero fill outside the image = the
//     padding, double-buffered two chunks ahead, image [row][channel][column] one float into its buffer so that a lane's four columns are
//     one ds_read2_b64 at immediate offsets.  The DMA is issued through inline asm: seen by the compiler, every outstanding one is drained
//     (s_waitcnt vmcnt(0)) before any LDS read that may alias it; the waits are counted by hand instead.
//   * epilogue: the output transform needs all four rows: each wave reduces its row to the two column sums of A^T M A's inner product,
//     the waves exchange them through LDS (the dead raw / U images) and wave q finishes (channel group, tile group) q: bias / ReLU (forward),
//     ReLU mask + injected diff (data gradient), optionally the fused 2x2 max-pool with its arg-max byte, 16-byte stores after a DPP row swap
//     -- the fp32 kernel's epilogue.
// Requirements (else the caller takes conv3x3_winograd.hip): K % 16 == 0, M % 64 == 0, W % 4 == 0, tensors below 4 GiB.  No unpooling
// input transform (the expansion costs 16 VALU per tile and channel in a loop that is VALU-bound already): the pool's backward stays
// maxpool_bwd_amap_k for these launches.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <type_traits>
#include "st2_kernels.h"

namespace st2 {

namespace {

typedef float ws_f32x16 __attribute__((ext_vector_type(16)));
typedef float ws_f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 ws_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 ws_bf16x2 __attribute__((ext_vector_type(2)));
typedef int ws_i32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* ws_lptr_t;

constexpr int WS_CH = 16;                 // input channels per chunk = the k of one MFMA
constexpr int WS_IW = 40;                 // staged columns x0-4 .. x0+35
constexpr int WS_ROWS = 10;               // staged rows y0-1 .. y0+8
constexpr int WS_ROW = WS_CH * WS_IW;     // 640 floats per staged row: [row][channel][column]
constexpr int WS_RAW = WS_ROWS * WS_ROW;  // 6400 floats = 25 wave-DMAs of 1 KiB
constexpr int WS_PIECES = WS_RAW / 256;
constexpr int WS_SHIFT = 1;               // the image starts one float into its buffer: pixel x0 - 1 + 2 t sits at an even index
constexpr int WS_RAWBUF = 6656;           // floats per raw buffer (26 KiB)
constexpr int WS_USLOT = 4 * 6 * 64;      // uint4 per ring slot: [wave][m group * 3 + split][lane]
constexpr unsigned kWsOOB = 0xffffffffu;

struct WsKArgs {
    const float* in; const uint4* upack; const float* bias; float* out;
    const float* mask_src; const float* inject;
    int K, M, H, W, nch, tiles_x, tiles_y, relu;
    unsigned in_bytes, u_bytes;
    float* pool_out; int pool_h, pool_w;   // optional fused 2x2/2 max-pool of the (post-ReLU) output: [M][pool_h][pool_w]
    unsigned char* pool_amap;              // optional, with pool_out: bits 0-1 = slot of the FIRST maximum, bit 2 = maximum > 0 after bias
    int splits; float* scratch;            // split-K: split s accumulates chunks [s, s+1) * nch / splits into scratch[s] (raw partial sums)
    unsigned long long* stamps;            // DIAG builds: per block 8 values {t begin, t loop, ticks loop, t epilogue, ticks epilogue, t end} (s_memtime / s_memrealtime; stored as taken)
};

__device__ __forceinline__ ws_bf16x8 ws_bf(const uint4& u) { return __builtin_bit_cast(ws_bf16x8, u); }
// {a, b} rounded to bf16 (nearest even) in one dword: a in the low half
__device__ __forceinline__ unsigned ws_cvt2(float a, float b)
{
    ws_f32x2 v; v.x = a; v.y = b;
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, ws_bf16x2));
}
// the high halves of {a, b} in one dword (truncation to bf16): a in the low half
__device__ __forceinline__ unsigned ws_hi2(float a, float b)
{
    return __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, b), __builtin_bit_cast(unsigned, a), 0x07060302u);
}
__device__ __forceinline__ float ws_hi(float a) { return __builtin_bit_cast(float, __builtin_bit_cast(unsigned, a) & 0xffff0000u); }
// 64 lanes x 16 bytes -> LDS [lds_addr, lds_addr + 1 KiB), by hand (see the header)
__device__ __forceinline__ void ws_dma16(const ws_i32x4& rsrc, unsigned lds_addr, unsigned voff, unsigned soff)
{
    asm volatile("s_mov_b32 m0, %0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" :: "s"(lds_addr), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
}
// ... with an instruction offset, which the hardware adds to the memory address AND to the LDS address: the pieces of one ring slot
// differ by 1 KiB on both sides, so one M0 / scalar-offset pair serves four of them
template <int OFF>
__device__ __forceinline__ void ws_dma16_at(const ws_i32x4& rsrc, unsigned lds_addr, unsigned voff, unsigned soff)
{
    static_assert(OFF >= 0 && OFF < 4096, "12-bit instruction offset");
    asm volatile("s_mov_b32 m0, %0\n\tbuffer_load_dwordx4 %1, %2, %3 offen offset:%4 lds" :: "s"(lds_addr), "v"(voff), "s"(rsrc), "s"(soff), "n"(OFF) : "memory");
}
__device__ __forceinline__ ws_i32x4 ws_rsrc(const void* p, unsigned bytes)
{
    const unsigned long long a = (unsigned long long)p;
    ws_i32x4 r;
    r.x = (int)(unsigned)a; r.y = (int)(unsigned)((a >> 32) & 0xffffu); r.z = (int)bytes; r.w = 0x00020000;
    return r;
}
template <int I> __device__ __forceinline__ void ws_set(uint4& v, unsigned x)
{
    if constexpr (I == 0) v.x = x; else if constexpr (I == 1) v.y = x; else if constexpr (I == 2) v.z = x; else v.w = x;
}
#define WS_INL __attribute__((always_inline))
// what-bounds-the-loop experiments (results wrong; never in a committed build): 1 = no U refills / A fetches in the loop, 2 = no B builds,
// 4 = no raw reads / row transform.  Built by hand: hipcc -DWS_VAR=n (tools/probes/wino_split_var.sh)
#ifndef WS_VAR
#define WS_VAR 0
#endif
template <int N> using WsI = std::integral_constant<int, N>;

// POOL: forward launches that also write the 2x2 max-pool of their output (and its arg-max map, when asked for);
// NOOUT: POOL launches with the map that do not write the full-resolution blob (conv3x3_winograd.hip);
// DG: data-gradient launches (1: injected diff at most, 2: ReLU mask of the blob below as well; no bias, ReLU or pool).
// One epilogue per kind: with the options as run-time flags the epilogue was 135 instructions per accumulator row, most of them
// control flow and 64-bit address arithmetic around stores that a launch does not have.
template <bool NOOUT, int DG, bool POOL, int DIAG>
__device__ __forceinline__ void conv3x3_wino_split_body(const WsKArgs& a)
{
    // raw[2] during the main loop + the U ring; the row exchange of the epilogue afterwards ([src wave][combo][e 8][lane] float2 = 64 KiB)
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * WS_RAWBUF * 4 + 4 * WS_USLOT * 16];      // 148 KiB: one workgroup per CU
    static_assert(sizeof(smem) <= 160 * 1024 && sizeof(smem) >= 64 * 1024, "LDS budget; the exchange fits");
    float* const lds = reinterpret_cast<float*>(smem);
    uint4* const u_s = reinterpret_cast<uint4*>(smem + 2 * WS_RAWBUF * 4);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int t31_l = lane & 31, kq_l = lane >> 5;                  // (main loop; the epilogue re-derives its own)

    // XCD-aware bijective block -> tile map, pixel tile fastest (the blocks of one XCD share a channel slice of U)
    const int nwg = gridDim.x / a.splits;
    const int split = blockIdx.x / nwg, orig = blockIdx.x - split * nwg;
    const int nch = a.nch / a.splits, c_first = split * nch;
    const int xcd = orig & 7, q8 = nwg >> 3, r8 = nwg & 7;
    const int logical = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (orig >> 3);
    const int n_pt = a.tiles_x * a.tiles_y;
    const int mt = logical / n_pt;
    const int pt = logical - mt * n_pt;
    const int tx = pt % a.tiles_x, ty = pt / a.tiles_x;
    const int y0 = ty * 8, x0 = tx * 32;
    const unsigned plane = (unsigned)a.H * a.W;

    if (DIAG && tid == 0 && a.stamps) a.stamps[8 * blockIdx.x] = __builtin_amdgcn_s_memtime();

    const ws_i32x4 rs_i = ws_rsrc(a.in, a.in_bytes), rs_u = ws_rsrc(a.upack, a.u_bytes);
    const unsigned lds_raw = (unsigned)(size_t)(ws_lptr_t)lds, lds_u = (unsigned)(size_t)(ws_lptr_t)u_s;
    unsigned ioff[7];
#pragma unroll
    for (int t = 0; t < 7; ++t) {
        const int piece = wave + 4 * t;
        const int e = (piece * 64 + lane) * 4;
        const int rr = e / WS_ROW;
        const int rem = e - rr * WS_ROW;
        const int c = rem / WS_IW;
        const int col = rem - c * WS_IW;
        const int gy = y0 - 1 + rr, gx = x0 - 4 + col;
        const bool ok = piece < WS_PIECES && gy >= 0 && gy < a.H && gx >= 0 && gx + 3 < a.W;
        ioff[t] = ok ? ((unsigned)c * plane + (unsigned)gy * a.W + gx) * 4u : kWsOOB;      // out of range -> zeros: the padding
    }
    auto dma_raw = [&](int ch, int buf) WS_INL {
        const unsigned coff = (unsigned)(c_first + ch) * WS_CH * plane * 4u;               // scalar offset (outside the range check)
#pragma unroll
        for (int t = 0; t < 7; ++t) {
            const int piece = wave + 4 * t;                     // wave-uniform
            if (piece < WS_PIECES) ws_dma16(rs_i, lds_raw + (unsigned)(buf * WS_RAWBUF + piece * 256 + WS_SHIFT) * 4u, ioff[t], coff);
        }
    };

    // row i of B^T d:  i = 0: d0 - d2   1: d1 + d2   2: d2 - d1   3: d1 - d3      (X = first, Y = second row; staged row 0 = y0 - 1)
    const int row_x = wave == 0 ? 0 : wave == 2 ? 2 : 1;
    const int row_y = wave == 0 ? 2 : wave == 1 ? 2 : wave == 2 ? 1 : 3;
    const int x_base = (2 * (t31_l >> 4)) * WS_ROW + (8 * kq_l) * WS_IW + 2 * (t31_l & 15) + 3 + WS_SHIFT;      // column 3 = pixel x0 - 1
    const float* const pX = lds + x_base + row_x * WS_ROW;
    const float* const pY = lds + x_base + row_y * WS_ROW;

    const unsigned u_lane = (unsigned)lane * 16u;
    const unsigned u_base = (unsigned)(((mt * a.nch + c_first) * 16 + 4 * wave) * 6 * 64 * 16);      // wave-uniform
    // piece q of position j of chunk c -> slot j: M0 = the slot's LDS address (+ 4 KiB for pieces 4, 5), scalar offset = the fragment's
    // place in the pack, both wave-uniform; the piece index travels in the instruction offset
    auto dma_u_one = [&](int c, int j, auto q_t) WS_INL {
        constexpr int q = decltype(q_t)::value;
        const unsigned dst = lds_u + (unsigned)((j * 4 + wave) * 6 * 1024), so = u_base + (unsigned)((c * 16 + j) * 6 * 1024);
        ws_dma16_at<(q & 3) * 1024>(rs_u, dst + (q >> 2) * 4096, u_lane, so + (q >> 2) * 4096);
    };
    uint4 aop[2][6];                     // A operands of the current / next position: [m group * 3 + split]
    // (one LDS base per ring slot, the piece in the instruction's offset: no address arithmetic in the loop)
    const uint4* ubase[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) ubase[j] = u_s + (j * 4 + wave) * 6 * 64 + lane;
    auto a_fetch_one = [&](int j, int set, int q) WS_INL { aop[set][q] = ubase[j][q * 64]; };

    float wv[8][4][2];                   // row i of B^T d of this lane's two tiles: [channel][column][tile group]
    const float sgn = wave == 1 ? 1.f : -1.f;
    ws_f32x2 tx_[2][4], ty_[2][4];       // raw values of two channels in flight: [channel parity][tile group * 2 + column pair]
    // LDS byte addresses of the X / Y rows of tile group 0 / 1 in the raw buffer being transformed (set once per chunk and made opaque:
    // the compiler otherwise keeps two of the four and re-derives the others with a v_add per channel; the channel is an immediate)
    typedef const __attribute__((address_space(3))) ws_f32x2* ws_l2ptr_t;
    unsigned wb[4];
    const unsigned lds_x = (unsigned)(size_t)(ws_lptr_t)pX, lds_y = (unsigned)(size_t)(ws_lptr_t)pY;
    auto w_bases = [&](int buf) WS_INL {
#pragma unroll
        for (int tg = 0; tg < 2; ++tg) {
            wb[tg] = lds_x + (unsigned)(buf * WS_RAWBUF + tg * 4 * WS_ROW) * 4u;
            wb[2 + tg] = lds_y + (unsigned)(buf * WS_RAWBUF + tg * 4 * WS_ROW) * 4u;
        }
        asm volatile("" : "+v"(wb[0]), "+v"(wb[1]), "+v"(wb[2]), "+v"(wb[3]));
    };
    auto w_read = [&](int ch, int part) WS_INL {                  // part 0: the X rows, 1: the Y rows (one ds_read2_b64 per tile group)
#pragma unroll
        for (int tg = 0; tg < 2; ++tg) {
            const ws_l2ptr_t q = (ws_l2ptr_t)(size_t)(wb[2 * part + tg] + (unsigned)(ch * WS_IW * 4));
            if (part == 0) { tx_[ch & 1][tg * 2] = q[0]; tx_[ch & 1][tg * 2 + 1] = q[1]; }
            else { ty_[ch & 1][tg * 2] = q[0]; ty_[ch & 1][tg * 2 + 1] = q[1]; }
        }
    };
    auto w_fma = [&](int ch, int col) WS_INL {                    // column col of both tile groups
#pragma unroll
        for (int tg = 0; tg < 2; ++tg) {
            const float yv = (col & 1) ? ty_[ch & 1][tg * 2 + (col >> 1)].y : ty_[ch & 1][tg * 2 + (col >> 1)].x;
            const float xv = (col & 1) ? tx_[ch & 1][tg * 2 + (col >> 1)].y : tx_[ch & 1][tg * 2 + (col >> 1)].x;
            asm("v_fma_f32 %0, %1, %2, %3" : "=v"(wv[ch][col][tg]) : "v"(yv), "v"(sgn), "v"(xv));
        }
    };
    uint4 bop[2][6];                     // B operands of the current / next position: [tile group * 3 + split]
    // the three-way split of one channel pair (2 cp, 2 cp + 1) of position j for both tile groups: 26 instructions in six parts
    float oa[2], ob[2], ta[2], tb[2];    // [tile group]: channel 2 cp / 2 cp + 1, and the fp32 value of their current bf16 term
    unsigned hh[2];
    auto build_part = [&](auto j_t, auto cp_t, auto part_t, int set) WS_INL {
        constexpr int j = decltype(j_t)::value, cp = decltype(cp_t)::value, part = decltype(part_t)::value;
        constexpr int ca = 2 * cp, cb = 2 * cp + 1;
        constexpr int c1 = j == 0 ? 0 : j == 1 ? 1 : j == 2 ? 2 : 1, c2 = j == 0 ? 2 : j == 1 ? 2 : j == 2 ? 1 : 3;
        if constexpr (part == 0) {           // the position's column combination (asm: the compiler would pair these into v_pk_add_f32)
#pragma unroll
            for (int tg = 0; tg < 2; ++tg) {
                if constexpr (j == 1) {
                    asm("v_add_f32 %0, %1, %2" : "=v"(oa[tg]) : "v"(wv[ca][c1][tg]), "v"(wv[ca][c2][tg]));
                    asm("v_add_f32 %0, %1, %2" : "=v"(ob[tg]) : "v"(wv[cb][c1][tg]), "v"(wv[cb][c2][tg]));
                } else {
                    asm("v_sub_f32 %0, %1, %2" : "=v"(oa[tg]) : "v"(wv[ca][c1][tg]), "v"(wv[ca][c2][tg]));
                    asm("v_sub_f32 %0, %1, %2" : "=v"(ob[tg]) : "v"(wv[cb][c1][tg]), "v"(wv[cb][c2][tg]));
                }
            }
        } else if constexpr (part == 1) {    // first term, to nearest; the fp32 values of tile group 0's
            hh[0] = ws_cvt2(oa[0], ob[0]); hh[1] = ws_cvt2(oa[1], ob[1]);
            ws_set<cp>(bop[set][0], hh[0]); ws_set<cp>(bop[set][3], hh[1]);
            ta[0] = __builtin_bit_cast(float, hh[0] << 16); tb[0] = __builtin_bit_cast(float, hh[0] & 0xffff0000u);
        } else if constexpr (part == 2) {
            ta[1] = __builtin_bit_cast(float, hh[1] << 16); tb[1] = __builtin_bit_cast(float, hh[1] & 0xffff0000u);
            asm("v_sub_f32 %0, %0, %1" : "+v"(oa[0]) : "v"(ta[0]));
            asm("v_sub_f32 %0, %0, %1" : "+v"(ob[0]) : "v"(tb[0]));
        } else if constexpr (part == 3) {    // second term of tile group 0, by truncation
            asm("v_sub_f32 %0, %0, %1" : "+v"(oa[1]) : "v"(ta[1]));
            asm("v_sub_f32 %0, %0, %1" : "+v"(ob[1]) : "v"(tb[1]));
            ws_set<cp>(bop[set][1], ws_hi2(oa[0], ob[0]));
            ta[0] = ws_hi(oa[0]); tb[0] = ws_hi(ob[0]);
        } else if constexpr (part == 4) {
            ws_set<cp>(bop[set][4], ws_hi2(oa[1], ob[1]));
            ta[1] = ws_hi(oa[1]); tb[1] = ws_hi(ob[1]);
            asm("v_sub_f32 %0, %0, %1" : "+v"(oa[0]) : "v"(ta[0]));
            asm("v_sub_f32 %0, %0, %1" : "+v"(ob[0]) : "v"(tb[0]));
        } else {                             // third term: what is left has at most 8 significant bits
            asm("v_sub_f32 %0, %0, %1" : "+v"(oa[1]) : "v"(ta[1]));
            asm("v_sub_f32 %0, %0, %1" : "+v"(ob[1]) : "v"(tb[1]));
            ws_set<cp>(bop[set][2], ws_hi2(oa[0], ob[0]));
            ws_set<cp>(bop[set][5], ws_hi2(oa[1], ob[1]));
        }
    };

    ws_f32x16 acc[16];                   // [position j][m group][tile group]

    // ---- prologue: raw chunks 0 and 1, the four ring slots of chunk 0, the row transform and position 0 of chunk 0 ----
    dma_raw(0, 0);
    if (nch > 1) dma_raw(1, 1);
#pragma unroll
    for (int j = 0; j < 4; ++j) { dma_u_one(0, j, WsI<0>{}); dma_u_one(0, j, WsI<1>{}); dma_u_one(0, j, WsI<2>{}); dma_u_one(0, j, WsI<3>{}); dma_u_one(0, j, WsI<4>{}); dma_u_one(0, j, WsI<5>{}); }
    __builtin_amdgcn_sched_barrier(0);       // (the 256 accumulator writes below run under the DMAs' latency, not in front of their issue)
#pragma unroll
    for (int p = 0; p < 16; ++p)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[p][e] = 0.f;
    __builtin_amdgcn_sched_barrier(0);
    // (the ring slots of positions 1 .. 3 -- the 18 newest DMAs -- are not waited for here: the steps' own counted waits cover them)
    asm volatile("s_waitcnt vmcnt(18)" ::: "memory");
    asm volatile("s_barrier" ::: "memory");       // both raw images and slot 0 have landed in every wave
#pragma unroll
    for (int q = 0; q < 6; ++q) a_fetch_one(0, 0, q);
#pragma unroll
    for (int ch = 0; ch < 8; ++ch) {           // (the next channel's reads are in flight while this one's columns are combined)
        if (ch == 0) { w_bases(0); w_read(0, 0); w_read(0, 1); }
        if (ch + 1 < 8) { w_read(ch + 1, 0); w_read(ch + 1, 1); }
#pragma unroll
        for (int col = 0; col < 4; ++col) w_fma(ch, col);
    }
    {
        auto one = [&](auto cp_t) WS_INL { build_part(WsI<0>{}, cp_t, WsI<0>{}, 0); build_part(WsI<0>{}, cp_t, WsI<1>{}, 0); build_part(WsI<0>{}, cp_t, WsI<2>{}, 0);
                                           build_part(WsI<0>{}, cp_t, WsI<3>{}, 0); build_part(WsI<0>{}, cp_t, WsI<4>{}, 0); build_part(WsI<0>{}, cp_t, WsI<5>{}, 0); };
        one(WsI<0>{}); one(WsI<1>{}); one(WsI<2>{}); one(WsI<3>{});
    }

    if (DIAG && tid == 0 && a.stamps) { a.stamps[8 * blockIdx.x + 1] = __builtin_amdgcn_s_memtime(); a.stamps[8 * blockIdx.x + 2] = __builtin_amdgcn_s_memrealtime(); }

    // six partial products, smallest first: (U split, V split)
    constexpr int PA[6] = {2, 0, 1, 1, 0, 0};
    constexpr int PB[6] = {0, 2, 1, 0, 1, 0};

    // One step = one position j of a chunk = 24 MFMA slots.  Dealt to the slots:
    //   slots 0 .. 5    the ring slot this step's A operands came from (read a step ago) is refilled: position j of the NEXT chunk
    //   slot 6          wait for the NEXT step's slot: its DMA was issued three steps ago, 6 per step since (the raw pieces in between only
    //                   make the wait stricter);   slots 6 .. 11: the next step's A operands
    //   slots 6 g + p   part p of the three-way split of channel pair g of the NEXT position's B operands
    //   steps 2 and 3   the next chunk's row transform, channel by channel, behind the last use of the old values (channel pair g of the old
    //                   chunk is last read in slot 6 g of step 2, of the new one first in slot 6 g of step 3): step 2 channels 0 .. 4
    //                   (starting in slots 1, 5, 9, 13, 17), step 3 channels 5 .. 7 (slots 0, 4, 8); a channel's X rows are read in its
    //                   first slot, its Y rows in the second, its four columns are combined in slots +4 .. +7
    auto step = [&](const int c, auto j_t, auto more_t) WS_INL {
        constexpr int j = decltype(j_t)::value;
        constexpr bool MORE = decltype(more_t)::value;
        constexpr int set = j & 1;
        using JN = WsI<(j + 1) & 3>;
        const int nbuf = (c + 1) & 1;
        auto aux = [&](auto k_t) WS_INL {
            constexpr int k = decltype(k_t)::value;
            if constexpr (MORE && k < 6 && !(WS_VAR & 1)) {
                if constexpr (k == 0) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                dma_u_one(c + 1, j, k_t);
            }
            if constexpr (k == 6 && !(WS_VAR & 1)) {
                if (MORE) asm volatile("s_waitcnt vmcnt(18)" ::: "memory");
                else if (j == 0) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
                else if (j == 1) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
                else if (j == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            if constexpr (k >= 6 && k < 12 && (MORE || j < 3) && !(WS_VAR & 1)) a_fetch_one((j + 1) & 3, set ^ 1, k - 6);
            if constexpr (MORE && (j == 2 || j == 3) && !(WS_VAR & 4)) {
                constexpr int first = j == 2 ? 0 : 5, count = j == 2 ? 5 : 3, s0 = j == 2 ? 1 : 0;
                if constexpr (j == 2 && k == 0) w_bases(nbuf);
#pragma unroll
                for (int n = 0; n < count; ++n) {
                    const int s = s0 + 4 * n;
                    if (k == s) w_read(first + n, 0);
                    if (k == s + 1) w_read(first + n, 1);
#pragma unroll
                    for (int col = 0; col < 4; ++col)
                        if (k == (s + 4 + col < 23 ? s + 4 + col : 23)) w_fma(first + n, col);      // three slots behind the reads and more
                }
            }
            if constexpr ((MORE || j < 3) && !(WS_VAR & 2)) build_part(JN{}, WsI<k / 6>{}, WsI<k % 6>{}, set ^ 1);
        };
        auto slots = [&](auto k_t, auto&& self) WS_INL {
            constexpr int k = decltype(k_t)::value;
            if constexpr (k < 24) {
                constexpr int p = k >> 2, mg = (k >> 1) & 1, tg = k & 1;
                acc[j * 4 + mg * 2 + tg] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ws_bf(aop[(WS_VAR & 1) ? 0 : set][mg * 3 + PA[p]]), ws_bf(bop[(WS_VAR & 2) ? 0 : set][tg * 3 + PB[p]]),
                                                                                 acc[j * 4 + mg * 2 + tg], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                aux(k_t);
                __builtin_amdgcn_sched_barrier(0);
                self(WsI<k + 1>{}, self);
            }
        };
        slots(WsI<0>{}, slots);
    };
    auto chunk = [&](const int c, auto more_t, auto more2_t) WS_INL {
        constexpr bool MORE = decltype(more_t)::value, MORE2 = decltype(more2_t)::value;
        // raw(c + 1) has landed in every wave (its DMA is older than the 18 newest of the last step's wait), and every wave is done
        // reading raw(c) (steps 2 and 3 of the previous chunk): buffer c & 1 is free for raw(c + 2)
        if (MORE) asm volatile("s_barrier" ::: "memory");
        if (MORE2) dma_raw(c + 2, c & 1);
        step(c, WsI<0>{}, more_t); step(c, WsI<1>{}, more_t); step(c, WsI<2>{}, more_t); step(c, WsI<3>{}, more_t);
    };
    {
        using T = std::true_type; using F = std::false_type;
        int c = 0;
        for (; c + 2 < nch; ++c) chunk(c, T{}, T{});
        if (c + 1 < nch) { chunk(c, T{}, F{}); ++c; }
        chunk(c, F{}, F{});
    }

    if (DIAG && threadIdx.x == 0 && a.stamps) { a.stamps[8 * blockIdx.x + 3] = __builtin_amdgcn_s_memtime(); a.stamps[8 * blockIdx.x + 4] = __builtin_amdgcn_s_memrealtime(); }

    // ---- epilogue.  Output transform of accumulator element e: y = A^T M A with M = the 4x4 transform-domain tile, row i in wave i:
    // each wave reduces its row to t = (M0 + M1 + M2, M1 - M2 - M3), the waves exchange the t of the three combos they do not finish
    // through LDS, and wave q finishes combo q = (m group, tile group): y0. = t(0) + t(1) + t(2), y1. = t(1) - t(2) - t(3).
    // Every global access is a buffer access: "no mask" / "no inject" / "no bias" are zero-size resources, "lane outside the image" an
    // out-of-range offset -- no branch, no 64-bit address.
    float2* const xch = reinterpret_cast<float2*>(lds);             // [src wave 4][combo 4][e 8][lane 64] (the raw images and 12 KiB of the ring are dead)
    const int mg_f = wave >> 1, tg_f = wave & 1;
    // (the lane-dependent values of the epilogue are derived from a lane index re-derived HERE: computed from `lane` they are hoisted
    // above the main loop and, at 250 registers in there, spilled across it)
    int lane_e = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));      // = lane, from the exec mask: nothing lives across the loop
    asm volatile("" : "+v"(lane_e));
    const int t31 = lane_e & 31, kq = lane_e >> 5;
    const int odd = lane_e & 1;
    const int gy = y0 + 4 * tg_f + 2 * (t31 >> 4) + odd;            // this lane's row after the swap
    const int gx4 = x0 + 2 * (t31 & 14);                            // first of the lane pair's 4 pixels (16-byte aligned)
    const bool part = a.splits > 1;
    const bool has_bias = !part && a.bias != nullptr, has_mask = DG == 2 && !part && a.mask_src != nullptr, has_inj = !part && a.inject != nullptr;
    const float floor_v = (!part && a.relu) ? 0.f : -__builtin_inff();      // ReLU as max(., floor)
    float* const outp = part ? a.scratch + (size_t)split * a.M * plane : a.out;
    const bool live = gx4 < a.W && gy < a.H;
    const int mw = mt * 64 + mg_f * 32 + 4 * kq;
    typedef unsigned ws_u32x4 __attribute__((ext_vector_type(4)));
    const unsigned out_bytes = (unsigned)a.M * plane * 4u, plane4 = plane * 4u;
    const unsigned off0 = live ? ((unsigned)mw * plane + (unsigned)gy * a.W + gx4) * 4u : kWsOOB;
    auto off_of = [&](int e) WS_INL { return live ? off0 + (unsigned)(8 * (e >> 2) + (e & 3)) * plane4 : kWsOOB; };
    const __amdgpu_buffer_rsrc_t rs_mk = __builtin_amdgcn_make_buffer_rsrc((void*)a.mask_src, 0, has_mask ? out_bytes : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_ij = __builtin_amdgcn_make_buffer_rsrc((void*)a.inject, 0, has_inj ? out_bytes : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_bs = __builtin_amdgcn_make_buffer_rsrc((void*)a.bias, 0, has_bias ? (unsigned)a.M * 4u : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_o = __builtin_amdgcn_make_buffer_rsrc((void*)outp, 0, NOOUT ? 0u : out_bytes, 0x00020000);
    // the fused pool (Caffe MAX 2x2/2, ceil mode): the lane's 2x2 tile IS one pooling window (tile origins are even); bias and ReLU
    // commute with max.  A window clipped by the bottom edge keeps its first row only.
    const int ty2 = y0 + 4 * tg_f + 2 * (t31 >> 4), tx2 = x0 + 2 * (t31 & 15);
    const bool plive = POOL && tx2 < a.W && ty2 < a.H, prow1 = ty2 + 1 < a.H;
    const unsigned pplane = POOL ? (unsigned)a.pool_h * a.pool_w : 0u;
    const unsigned po0 = plive ? (unsigned)mw * pplane + (unsigned)(ty2 >> 1) * a.pool_w + (tx2 >> 1) : 0u;
    const __amdgpu_buffer_rsrc_t rs_po = __builtin_amdgcn_make_buffer_rsrc((void*)a.pool_out, 0, POOL ? (unsigned)a.M * pplane * 4u : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_pa = __builtin_amdgcn_make_buffer_rsrc((void*)a.pool_amap, 0, (POOL && a.pool_amap) ? (unsigned)a.M * pplane : 0u, 0x00020000);
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        // the bias / ReLU-mask / injected-diff values of this half's 8 accumulator rows are requested before its exchange round (the
        // loop's registers are dead): their latency runs under the LDS traffic and the barriers
        float mk[DG == 2 ? 8 : 1][4], ij[DG ? 8 : 1][4], bs[DG ? 1 : 8];
        if constexpr (DG == 2) {
#pragma unroll
            for (int el = 0; el < 8; ++el) {
                const uint4 v = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rs_mk, off_of(8 * half + el), 0, 0));
                mk[el][0] = __builtin_bit_cast(float, v.x); mk[el][1] = __builtin_bit_cast(float, v.y); mk[el][2] = __builtin_bit_cast(float, v.z); mk[el][3] = __builtin_bit_cast(float, v.w);
            }
        }
        if constexpr (DG != 0) {
#pragma unroll
            for (int el = 0; el < 8; ++el) {
                const uint4 v = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rs_ij, off_of(8 * half + el), 0, 0));
                ij[el][0] = __builtin_bit_cast(float, v.x); ij[el][1] = __builtin_bit_cast(float, v.y); ij[el][2] = __builtin_bit_cast(float, v.z); ij[el][3] = __builtin_bit_cast(float, v.w);
            }
        } else {
#pragma unroll
            for (int el = 0; el < 8; ++el) {
                const int e = 8 * half + el;
                bs[el] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_bs, (mw + 8 * (e >> 2) + (e & 3)) * 4, 0, 0));
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();                                            // the images / the previous half's exchange are dead
#pragma unroll
        for (int qq = 0; qq < 4; ++qq)
#pragma unroll
            for (int el = 0; el < 8; ++el) {
                const int e = 8 * half + el;
                const float m0 = acc[0 * 4 + qq][e], m1 = acc[1 * 4 + qq][e], m2 = acc[2 * 4 + qq][e], m3 = acc[3 * 4 + qq][e];
                xch[((wave * 4 + qq) * 8 + el) * 64 + lane_e] = make_float2(m0 + m1 + m2, m1 - m2 - m3);
                if (el == 3 || el == 7) __builtin_amdgcn_sched_barrier(0);      // (keeps the accumulator reads from being hoisted en bloc: they would spill the values above)
            }
        __syncthreads();
#pragma unroll
        for (int el = 0; el < 8; ++el) {
            const int e = 8 * half + el;
            const float2 s0 = xch[((0 * 4 + wave) * 8 + el) * 64 + lane_e], s1 = xch[((1 * 4 + wave) * 8 + el) * 64 + lane_e];
            const float2 s2 = xch[((2 * 4 + wave) * 8 + el) * 64 + lane_e], s3 = xch[((3 * 4 + wave) * 8 + el) * 64 + lane_e];
            const float y00 = s0.x + s1.x + s2.x, y01 = s0.y + s1.y + s2.y;
            const float y10 = s1.x - s2.x - s3.x, y11 = s1.y - s2.y - s3.y;
            if constexpr (POOL) {
                const float b = bs[DG ? 0 : el];
                const float p0 = y00 > y01 ? y00 : y01, p1 = y10 > y11 ? y10 : y11;
                const float pm = prow1 ? (p0 > p1 ? p0 : p1) : p0;
                // WHERE the maximum is (first one of a row-major scan) and whether it is positive after the bias -- compared AFTER the
                // bias, as the stored blob the reference's pooling layer scans is (conv3x3_winograd.hip)
                const float qb = pm + b;
                const unsigned slot = y00 + b == qb ? 0u : (y01 + b == qb ? 1u : (y10 + b == qb ? 2u : 3u));
                const unsigned po = plive ? po0 + (unsigned)(8 * (e >> 2) + (e & 3)) * pplane : kWsOOB;       // (kWsOOB * 4 wraps to an offset that is out of range too)
                __builtin_amdgcn_raw_buffer_store_b8((unsigned char)(slot | (qb > 0.f ? 4u : 0u)), rs_pa, po, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, __builtin_fmaxf(qb, floor_v)), rs_po, plive ? po * 4u : kWsOOB, 0, 0);
            }
            if constexpr (NOOUT) continue;
            // give away the row this lane does not keep, receive the partner's part of the row it keeps (quad_perm 1,0,3,2)
            const float g0 = odd ? y00 : y10, g1 = odd ? y01 : y11;
            const float q0 = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, g0), 0xB1, 0xf, 0xf, true));
            const float q1 = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, g1), 0xB1, 0xf, 0xf, true));
            float o[4];
            o[0] = odd ? q0 : y00; o[1] = odd ? q1 : y01; o[2] = odd ? y10 : q0; o[3] = odd ? y11 : q1;
            if constexpr (DG) {
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    if constexpr (DG == 2) o[jj] = mk[el][jj] > 0.f ? o[jj] : 0.f;
                    o[jj] += ij[el][jj];
                }
            } else {
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) o[jj] = __builtin_fmaxf(o[jj] + bs[DG ? 0 : el], floor_v);
            }
            ws_u32x4 ov;
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) ov[jj] = __builtin_bit_cast(unsigned, o[jj]);
            __builtin_amdgcn_raw_buffer_store_b128(ov, rs_o, off_of(e), 0, 0);
        }
    }
    if (DIAG && threadIdx.x == 0 && a.stamps) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); a.stamps[8 * blockIdx.x + 5] = __builtin_amdgcn_s_memtime(); }
}

}  // namespace

__global__ __launch_bounds__(256, 1) void conv3x3_wino_split_64x256(const WsKArgs a) { conv3x3_wino_split_body<false, 0, false, 0>(a); }
__global__ __launch_bounds__(256, 1) void conv3x3_wino_split_64x256_pool(const WsKArgs a) { conv3x3_wino_split_body<false, 0, true, 0>(a); }
__global__ __launch_bounds__(256, 1) void conv3x3_wino_split_64x256_noout(const WsKArgs a) { conv3x3_wino_split_body<true, 0, true, 0>(a); }
__global__ __launch_bounds__(256, 1) void conv3x3_wino_split_64x256_dgrad(const WsKArgs a) { conv3x3_wino_split_body<false, 1, false, 0>(a); }
__global__ __launch_bounds__(256, 1) void conv3x3_wino_split_64x256_dgrad_masked(const WsKArgs a) { conv3x3_wino_split_body<false, 2, false, 0>(a); }
__global__ __launch_bounds__(256, 1) void conv3x3_wino_split_64x256_stamped(const WsKArgs a) { conv3x3_wino_split_body<false, 0, false, 1>(a); }
__global__ __launch_bounds__(256, 1) void conv3x3_wino_split_64x256_dgrad_stamped(const WsKArgs a) { conv3x3_wino_split_body<false, 2, false, 1>(a); }

// ---- host side ----
static unsigned short ws_f2bf(float f)             // round to nearest even (finite input)
{
    unsigned u;
    memcpy(&u, &f, 4);
    return (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}
static float ws_bf2f(unsigned short h) { unsigned u = (unsigned)h << 16; float f; memcpy(&f, &u, 4); return f; }

size_t wino_split_pack_elems(int K, int M) { return (size_t)(M / 64) * (K / 16) * 16 * 2 * 3 * 64 * 8; }

// dst[((((((m / 64) * K/16 + k / 16) * 16 + pos) * 2 + (m % 64) / 32) * 3 + split) * 64 + lane) * 8 + k % 8], lane = m % 32 + 32 * ((k % 16) / 8):
// the three bf16 terms of (G g G^T)[pos] in the A-fragment order of v_mfma_f32_32x32x16_bf16 (one 16-byte load per lane)
static void ws_pack(const float* w, int Cout, int Cin, bool dgrad, unsigned short* dst)
{
    const int K = dgrad ? Cout : Cin, M = dgrad ? Cin : Cout, nch = K / 16;
    static const double G[4][3] = {{1, 0, 0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0, 0, 1}};
    for (int m = 0; m < M; ++m)
        for (int k = 0; k < K; ++k) {
            double g[3][3], t[4][3];
            for (int tap = 0; tap < 9; ++tap)
                g[tap / 3][tap % 3] = dgrad ? w[((size_t)k * Cin + m) * 9 + (8 - tap)] : w[((size_t)m * Cin + k) * 9 + tap];
            for (int i = 0; i < 4; ++i)
                for (int j = 0; j < 3; ++j) t[i][j] = G[i][0] * g[0][j] + G[i][1] * g[1][j] + G[i][2] * g[2][j];
            for (int i = 0; i < 4; ++i)
                for (int j = 0; j < 4; ++j) {
                    const float u = (float)(t[i][0] * G[j][0] + t[i][1] * G[j][1] + t[i][2] * G[j][2]);
                    const unsigned short h1 = ws_f2bf(u);
                    const float r1 = u - ws_bf2f(h1);
                    const unsigned short h2 = ws_f2bf(r1);
                    const unsigned short h3 = ws_f2bf(r1 - ws_bf2f(h2));
                    const unsigned short hs[3] = {h1, h2, h3};
                    const int lane = (m % 32) + 32 * ((k % 16) / 8), pos = 4 * i + j;
                    for (int s = 0; s < 3; ++s)
                        dst[((((((size_t)(m / 64) * nch + k / 16) * 16 + pos) * 2 + (m % 64) / 32) * 3 + s) * 64 + lane) * 8 + k % 8] = hs[s];
                }
        }
}
void pack_wino_split_weights_fwd(const float* w, int Cout, int Cin, unsigned short* dst) { ws_pack(w, Cout, Cin, false, dst); }
void pack_wino_split_weights_dgrad(const float* w, int Cout, int Cin, unsigned short* dst) { ws_pack(w, Cout, Cin, true, dst); }

bool conv_wino_split_ok(int K, int M, int H, int W)
{
    if (!(K >= 16 && K % 16 == 0 && M >= 64 && M % 64 == 0 && W >= 4 && W % 4 == 0 && H >= 1)) return false;
    return 4ull * K * H * W < 0xfffffff0ull && 4ull * M * H * W < 0xfffffff0ull && 2ull * wino_split_pack_elems(K, M) < 0xfffffff0ull;
}

// Split-K factor for a launch that would leave most CUs idle (conv5_1 at 1024^2: 128 workgroups on 256 CUs)
int conv_wino_split_splits(int K, int M, int H, int W)
{
    static const bool off = [] { const char* e = getenv("ST2_WINO_SPLITK"); return e && *e == '0'; }();
    if (off || !conv_wino_split_ok(K, M, H, W)) return 1;
    const long long nblk = (long long)((W + 31) / 32) * ((H + 7) / 8) * (M / 64);
    const int nch = K / WS_CH;
    int sp = 1;
    while (nblk * sp * 2 <= 256 && nch % (sp * 2) == 0 && nch / (sp * 2) >= 4 && sp < 16) sp *= 2;
    return sp;
}
bool conv_wino_split_can_pool(int K, int M, int H, int W)
{
    static const bool off = [] { const char* e = getenv("ST2_WINO_POOL"); return e && *e == '0'; }();
    return !off && conv_wino_split_ok(K, M, H, W) && conv_wino_split_splits(K, M, H, W) == 1;
}
bool conv_wino_split_pool_amap_ok(int K, int M, int H, int W) { return conv_wino_split_can_pool(K, M, H, W) && H % 2 == 0; }
bool conv_wino_split_can_skip_out(int K, int M, int H, int W) { return conv_wino_split_pool_amap_ok(K, M, H, W); }

// p.wpack = the split pack (pack_wino_split_weights_*, passed as const float*); the ConvProblem fields of launch_conv3x3_wino but unpool_amap
hipError_t launch_conv3x3_wino_split(const ConvProblem& p, hipStream_t s)
{
    if (!conv_wino_split_ok(p.K, p.M, p.H, p.W) || (reinterpret_cast<uintptr_t>(p.in) & 15) != 0 || p.unpool_amap) return hipErrorInvalidValue;
    WsKArgs k{};
    k.in = p.in; k.upack = reinterpret_cast<const uint4*>(p.wpack); k.bias = p.bias; k.out = p.out;
    k.mask_src = p.mask_src; k.inject = p.inject;
    k.K = p.K; k.M = p.M; k.H = p.H; k.W = p.W; k.nch = p.K / WS_CH;
    k.tiles_x = (p.W + 31) / 32; k.tiles_y = (p.H + 7) / 8; k.relu = p.relu;
    k.in_bytes = (unsigned)(4ull * p.K * p.H * p.W);
    k.u_bytes = (unsigned)(2ull * wino_split_pack_elems(p.K, p.M));
    k.stamps = p.stamps;
    const long long nblk = (long long)k.tiles_x * k.tiles_y * (p.M / 64);
    if (nblk <= 0 || nblk > 0x7fffffffLL) return hipErrorInvalidValue;
    k.splits = 1; k.scratch = nullptr;
    k.pool_out = p.pool_out; k.pool_h = (p.H + 1) / 2; k.pool_w = (p.W + 1) / 2;
    k.pool_amap = p.pool_out ? p.pool_amap : nullptr;
    if (p.scratch) {
        const int sp = conv_wino_split_splits(p.K, p.M, p.H, p.W);
        if (sp > 1 && p.scratch_floats >= (size_t)sp * p.M * p.H * p.W) { k.splits = sp; k.scratch = p.scratch; }
    }
    if (k.splits > 1 && p.pool_out) return hipErrorInvalidValue;              // the caller asks conv_wino_split_can_pool() first
    if (p.pool_amap && p.pool_out && p.H % 2 != 0) return hipErrorInvalidValue;
    // out == nullptr: only with the fused pool AND its arg-max map, one pass, no mask / inject
    const bool noout = !p.out;
    if (noout && (!k.pool_out || !k.pool_amap || k.splits > 1 || p.mask_src || p.inject)) return hipErrorInvalidValue;
    const dim3 g((unsigned)(nblk * k.splits)), b(256);
    // data-gradient epilogue (mask / inject), forward epilogue (bias / ReLU) with or without the pool; a split-K launch writes raw partial
    // sums: the plain forward build
    const bool dg = k.splits == 1 && (p.mask_src || p.inject);
    if (dg && (p.bias || p.relu || p.pool_out)) return hipErrorInvalidValue;
    if (p.stamps) {
        if (noout || k.pool_out) return hipErrorInvalidValue;
        if (dg) conv3x3_wino_split_64x256_dgrad_stamped<<<g, b, 0, s>>>(k); else conv3x3_wino_split_64x256_stamped<<<g, b, 0, s>>>(k);
    }
    else if (noout) conv3x3_wino_split_64x256_noout<<<g, b, 0, s>>>(k);
    else if (k.pool_out) conv3x3_wino_split_64x256_pool<<<g, b, 0, s>>>(k);
    else if (dg && p.mask_src) conv3x3_wino_split_64x256_dgrad_masked<<<g, b, 0, s>>>(k);
    else if (dg) conv3x3_wino_split_64x256_dgrad<<<g, b, 0, s>>>(k);
    else conv3x3_wino_split_64x256<<<g, b, 0, s>>>(k);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess || k.splits == 1) return e;
    return launch_wino_combine(k.scratch, k.splits, p.bias, p.relu, p.mask_src, p.inject, p.out, p.M, p.H, p.W, s);
}

}  // namespace st2
