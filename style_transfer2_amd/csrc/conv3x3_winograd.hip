// Winograd F(2x2,3x3) conv3x3 on the fp32 matrix cores (gfx950): forward and data-gradient of the VGG convs
// (reference: pycaffe Convolution forward / backward behind worker.py:84-86 and :100-106): the kernel
// (conv3x3_wino_body) and its host side (packing, launch, split-K).  The probes that sized the design (operand feed,
// issue rate, LDS-staged feed) live in tools/probes/, outside the product library.
//
// The 16 transform-domain GEMMs of one block keep 16 x (32 m x 32 tiles) fp32 accumulators per wave (256 VGPRs),
// so the block tile is small (128 m x 32 tiles) and the transformed weights U have to stream at ~16 B/clk/CU.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <type_traits>
#include "st2_kernels.h"

namespace st2 {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// ===========================================================================================================
// Winograd F(2x2,3x3) conv3x3 (pad 1, stride 1), NCHW fp32, on v_mfma_f32_32x32x2_f32.
//
//   Y = A^T [ sum_k (G g_k G^T) (.) (B^T d_k B) ] A        per 2x2 output tile, 4x4 input tile d, 3x3 filter g
//
// 16 transform-domain positions = 16 independent GEMMs  Acc_pos[m][tile] += U_pos[m][k] * V_pos[k][tile]:
// 4 multiplies per output instead of 9, so the matrix pipe executes 2.25x fewer flops than the direct kernel
// (conv3x3_mfma.hip) for the same result up to fp32 reassociation (every product and sum is still IEEE fp32).
//
// One workgroup = 4 waves = 128 output channels x (4 rows x 32 columns) pixels = 32 tiles (2 tile rows x 16), or
// 64 channels x (8 rows x 32 columns) for layers with <= 64 output channels.
// A wave owns 32 channels x 32 tiles: 16 positions x one 32x32 accumulator = 256 AGPRs, which is why the kernel
// runs one wave per SIMD and everything below is software-pipelined by hand.
//   U (host-transformed weights) never touches LDS: it is packed in MFMA A-operand order
//       [m/32][k/2][pos/4][lane][pos%4]   (lane&31 -> m, lane>>5 -> k parity)
//     and streamed L2 -> VGPR with global_load_dwordx4 three k-pairs ahead through a 4-set register ring (16 KiB per
//     k-pair per CU: the CU's vector-memory ingest rate, ~13.8 B/clk, is what bounds the main loop; the workgroups
//     that share an XCD walk the same channel slice in step, so the stream is served by that XCD's L2);
//   raw activations: LDS-DMA, 8 channels x 6 rows x 40 floats per chunk as aligned quads (any-width build: one
//     64-lane dword piece per row), zero fill outside the image = the padding, double-buffered, two chunks ahead;
//   V: each thread transforms one (tile, channel) pair per chunk (32 adds) one chunk ahead and writes the four rows
//     of B^T d B as ds_write_b128 into the B-operand image [k-pair][pos/4][k parity * 32 + tile][pos%4];
//   per k-pair: 16 MFMAs; the auxiliary work (B operands of the next k-pair = 4 ds_read_b128, U of k-pair +3,
//     a third of the input transform, DMA issue) sits in ONE clump after the first MFMA, the other 15 MFMAs run
//     back to back (one wave per SIMD: nothing a wave issues overlaps its own MFMAs); one s_barrier per 64 MFMAs
//     with `s_waitcnt vmcnt(8)` so that the U ring stays in flight across it.
// Epilogue: the output transform is in-lane (a lane holds all 16 positions of its (m, tile) pairs); lane pairs swap
// one row (DPP) so that every lane loads / stores 16 bytes; bias / ReLU / ReLU-mask / injected diff as in the direct
// kernel; optionally the 2x2 max-pool of the output (a tile is one pooling window).  Launches with few workgroups
// split K (wino_combine_k finishes them).
// Requirements (else the caller uses the direct kernel): K % 8 == 0, >= 48 output channels; tensors of 4 GiB and more: the BIG builds.
// ===========================================================================================================

constexpr int WN_CH = 8;                         // input channels per chunk (4 k-pairs)
constexpr int WN_IW = 40;                        // staged columns x0-4 .. x0+35 (10 aligned quads)
constexpr int WN_V = 4 * 16 * 64;                // floats per V image (one 32-tile group, one chunk)

size_t wino_pack_floats(int K, int M) { return (size_t)((M + 127) / 128 * 4) * (K / 2) * 1024; }

static void wino_pack(const float* w, int Cout, int Cin, bool dgrad, float* dst)
{
    const int K = dgrad ? Cout : Cin, M = dgrad ? Cin : Cout, nkp = K / 2;
    memset(dst, 0, wino_pack_floats(K, M) * sizeof(float));
    static const double G[4][3] = {{1, 0, 0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0, 0, 1}};
    for (int m = 0; m < M; ++m)
        for (int k = 0; k < K; ++k) {
            double g[3][3], t[4][3], u[4][4];
            for (int tap = 0; tap < 9; ++tap)
                g[tap / 3][tap % 3] = dgrad ? w[((size_t)k * Cin + m) * 9 + (8 - tap)] : w[((size_t)m * Cin + k) * 9 + tap];
            for (int i = 0; i < 4; ++i)
                for (int j = 0; j < 3; ++j) t[i][j] = G[i][0] * g[0][j] + G[i][1] * g[1][j] + G[i][2] * g[2][j];
            for (int i = 0; i < 4; ++i)
                for (int j = 0; j < 4; ++j) u[i][j] = t[i][0] * G[j][0] + t[i][1] * G[j][1] + t[i][2] * G[j][2];
            const size_t base = ((size_t)(m / 32) * nkp + k / 2) * 1024;
            const int lane = (k & 1) * 32 + (m & 31);
            for (int pos = 0; pos < 16; ++pos)
                dst[base + ((size_t)(pos >> 2) * 64 + lane) * 4 + (pos & 3)] = (float)u[pos >> 2][pos & 3];
        }
}
void pack_wino_weights_fwd(const float* w, int Cout, int Cin, float* dst) { wino_pack(w, Cout, Cin, false, dst); }
void pack_wino_weights_dgrad(const float* w, int Cout, int Cin, float* dst) { wino_pack(w, Cout, Cin, true, dst); }

struct WinoKArgs {
    const float* in; const float4* upack; const float* bias; float* out;
    const float* mask_src; const float* inject;
    int K, M, H, W, nch, tiles_x, tiles_y, n_mtiles, relu;
    unsigned in_bytes, u_bytes;
    float* pool_out; int pool_h, pool_w;   // optional fused 2x2/2 max-pool of the (post-ReLU) output: [M][pool_h][pool_w]
    unsigned char* pool_amap;              // optional, with pool_out: [M][pool_h][pool_w] bytes, bits 0-1 = slot of the FIRST maximum
                                           // (row-major in the window), bit 2 = maximum > 0 after bias: all the pool backward needs
    // optional (data-gradient launches below a pool, UNPOOL builds): `in` is the POOLED diff [K][ph][pw] and `unpool_amap` the arg-max map
    // of that pool ([K][ph][pw] bytes: slot | positive << 2); the input transform expands them (conv_wino_can_unpool)
    const unsigned char* unpool_amap; unsigned amap_bytes; int ph, pw;
    int splits; float* scratch;   // split-K (few workgroups, deep K): split s accumulates chunks [s, s+1) * nch / splits into scratch[s]
    unsigned long long* stamps;   // DIAG builds only: per block {shader cycles, 100 MHz ticks} of the main loop
};

typedef __attribute__((address_space(3))) void* lptr_t;
constexpr unsigned kOOB = 0xffffffffu;

__device__ __forceinline__ float f4c(const float4& v, int i) { return i == 0 ? v.x : i == 1 ? v.y : i == 2 ? v.z : v.w; }

// WM waves along m (32 channels each) x TG tile groups (32 tiles = 4 rows x 32 columns each), WM * TG = 4:
//   <4,1>: 128 channels x  4 rows x 32 columns      <2,2>: 64 channels x 8 rows x 32 columns
// With TG = 2 the two waves that share a channel slice load the same U lines together (one L2 fetch), and every
// thread transforms two (tile, channel) pairs per chunk.
// PS = true (position split, WM = TG = 2): the two waves that share a channel slice split the 16 transform-domain
//                POSITIONS instead of the tile groups: wave (m, ph) accumulates positions 8 ph .. 8 ph + 7 of BOTH tile
//                groups (still 16 accumulators).  Every U quad is then loaded by exactly one wave and feeds two MFMAs:
//                8 KiB of U per k-pair per CU instead of 16 -- the vector-memory ingest of the CU (~12.8 B/clk) is what
//                bounds the main loop of the other two variants.  Price: the output transform needs all 16 positions of a
//                tile in one lane, so the epilogue exchanges PARTIAL 2x2 outputs (the transform is linear: 4 floats per
//                (channel, tile) instead of 8 positions) through LDS: wave ph finishes tile group ph.
// W8 = true (<4,1> only): EIGHT waves, two per SIMD.  A pair of SIMD partners (waves w and w + 4) shares one channel
//                slice and splits the 16 positions: 128 accumulator registers per wave instead of 256, which is what lets
//                two waves live on one SIMD.  With one wave per SIMD nothing a wave issues overlaps its own fp32 MFMAs
//                (profiles/r01_f_issue_probe: 64 cycles per MFMA bare, +13 for the first two instructions in between), so
//                every operand fetch / transform / DMA instruction is paid in full; with two waves the partner's MFMAs
//                run underneath them.  U bytes per CU are unchanged (each wave loads the two quads of its position half),
//                the input transform of a chunk is done by ONE of the two halves (alternating by chunk), and the
//                epilogue exchanges partial 2x2 outputs between the partners through LDS (as PS does): the wave with
//                position half ph finishes accumulator rows 8 ph .. 8 ph + 7.
// QUAD = true : W % 4 == 0 -- activation rows staged as aligned quads (40 floats, x0-4 .. x0+35), 16-byte epilogue accesses.
// QUAD = false: any W     -- one staged row = one dword LDS-DMA piece of 64 lanes (x0-1 .. x0+62, 34 used; the row part of
//                the address is scalar, the lane part is computed once), 8-byte (W even) or 4-byte epilogue accesses.
// H4 = true (WM = 2, TG = 1): FOUR waves, 64 channels x 4 rows x 32 columns: wave (m, ph) keeps positions 8 ph .. 8 ph + 7 of
//                channel slice m -- 128 accumulator registers, so TWO workgroups share a CU (two waves per SIMD that belong to
//                DIFFERENT workgroups: no barrier couples them, and one workgroup's prologue / epilogue runs under the other's
//                MFMAs).  For layers whose K loop is short (conv1_2: 8 chunks) the epilogue of the one-wave-per-SIMD kernels
//                is ~40 % of a workgroup's time and nothing overlaps it.  Epilogue exchange as W8 (partner = wave ^ 2), the
//                exchange buffer is the dead V images.
// UNPOOL = true (QUAD, one tile group; data-gradient launches directly below a max-pool): the launch reads the POOLED diff and the
//                pool's one-byte arg-max map instead of the full-resolution diff the pool backward would have written: per chunk
//                8 channels x 4 pooled rows x 24 columns of floats and of bytes are staged (3 + 0.75 KiB instead of 7.5), and the input
//                transform of a 4x4 tile reads its 3x3 pooling windows: d[i][j] = (slot of window == position of (i, j) in it, and the
//                maximum was positive) ? pooled diff : 0 -- exactly the values maxpool_bwd_amap_k stores, so the result is the same bit
//                for bit, and that kernel, its 4x larger output and this launch's read of it are gone.
// BIG = true (QUAD, one tile group): tensors of 4 GiB and more (one engine on an 8192 x 8192 image: conv1 blobs of 17 GB).  Buffer addressing is
//                a 32-bit byte offset from the resource's base, so the base moves instead: the activation resource is rebuilt per chunk at
//                channel (c_first + ch) * 8 (a 64-bit scalar add; a chunk's 8 planes must stay below 4 GiB: H * W < 2^27) and the lane offsets are
//                relative to the chunk.  The epilogue's 32-bit ELEMENT offsets already reach 2^32 elements (64 x 8192 x 8192 exactly).
// NOOUT = true (forward launches with the fused pool AND its arg-max map): the full-resolution blob is not written -- nothing reads it when
//                the layer is pooled and carries no weight (the next conv reads the pooled blob, the pool's backward the map with the ReLU
//                sign in it): the row exchange, bias / ReLU of the four pixels and the 16-byte store of every accumulator row are gone.
//                A build of its own because the 512-register kernels have no register for a run-time flag (it spilled).
// EPI != 0 (round 5; QUAD, one pass, tensors below 4 GiB, the 128-channel and the half-tile builds): an epilogue per launch KIND instead of
//                one epilogue with every option as a run-time flag -- 1 forward (bias + ReLU), 2 forward + fused pool (+ arg-max map; with
//                NOOUT: the pooled blob only), 3 data gradient with the ReLU mask of the blob below, 4 data gradient without one (injected
//                diff optional in both).  Every global access is a buffer access ("absent" = a zero-size resource, "outside the image or
//                the channel range" = an out-of-range offset): no branch around a load or store, no 64-bit address arithmetic.  The
//                arithmetic is the generic epilogue's, operation for operation: results are the same bits.  Measured on the split-operand
//                kernel first (conv3x3_wino_split.hip: 135 -> ~50 instructions per accumulator row, 12.9 k -> 6.9 k cycles); here the
//                generic epilogue was ~3 200 instructions per wave after the last MFMA for ~900 of essential work.
template <int WM, int TG, int DIAG = 0, bool QUAD = true, bool PS = false, bool W8 = false, bool H4 = false, bool UNPOOL = false, bool BIG = false,
          bool NOOUT = false, int EPI = 0>
__device__ __forceinline__ void conv3x3_wino_body(const WinoKArgs& a)
{
    static_assert(EPI == 0 || (QUAD && TG == 1 && !PS && !W8 && !BIG && DIAG == 0), "specialised epilogues: aligned widths, one tile group, four waves, below 4 GiB");
    static_assert(!NOOUT || EPI == 0 || EPI == 2, "no-output builds pool");
    static_assert(!NOOUT || (QUAD && TG == 1 && !PS && !W8 && !UNPOOL && !BIG && DIAG == 0), "no-output builds: the builds that write the arg-max map");
    static_assert(!BIG || (QUAD && TG == 1 && !PS && !W8 && DIAG == 0), "big tensors: aligned widths, one tile group, four waves");
    static_assert(!UNPOOL || (QUAD && TG == 1 && !PS && !W8), "unpool: aligned widths, one tile group, four waves");
    static_assert(H4 || (WM * TG == 4 && (TG == 1 || TG == 2)), "4 waves");
    static_assert(!PS || (WM == 2 && TG == 2), "position split: 2 channel slices x 2 position halves");
    static_assert(!W8 || (WM == 4 && TG == 1 && QUAD && !PS), "eight waves: 4 channel slices x 2 position halves, aligned widths");
    static_assert(!H4 || (WM == 2 && TG == 1 && QUAD && !PS && !W8), "half tile: 2 channel slices x 2 position halves, aligned widths");
    constexpr bool RS = W8 || H4;                        // accumulator ROWS of a channel slice are finished by two partner waves
    constexpr bool AMAP = QUAD && TG == 1;               // builds whose fused pool also writes the arg-max map (register budget)
    constexpr int NW = W8 ? 8 : 4;                       // waves per workgroup
    constexpr int PARTNER = W8 ? 4 : 2;                  // RS: wave ^ PARTNER holds the other position half of the same channel slice
    constexpr bool SPLIT = PS || RS;                     // the 16 positions are split over two waves
    constexpr int BM = 32 * WM;
    constexpr int PROWS = 4 * TG;                        // pixel rows per block
    constexpr int IN_ROWS = PROWS + 2;
    constexpr int IW = QUAD ? WN_IW : 64;                // floats per staged row
    constexpr int COL0 = QUAD ? 3 : 0;                   // staged column of pixel x0 - 1
    constexpr int PLANE = IN_ROWS * IW;
    constexpr int UP_ROWS = PROWS / 2 + 2, UP_COLS = 24;  // UNPOOL: staged pooled rows y0/2 - 1 .., columns x0/2 - 4 .. x0/2 + 19
    constexpr int UP_PLANE = UP_ROWS * UP_COLS;          // per channel: floats of the pooled diff = bytes of the arg-max map
    constexpr int UP_F = WN_CH * UP_PLANE;               // 768 floats, then the same geometry in bytes
    constexpr int UP_FQ = UP_F / 256, UP_AQ = UP_F / 256; // wave-DMAs: 16-byte pieces of floats, 4-byte pieces of map bytes (64 lanes each)
    static_assert(!UNPOOL || (UP_F % 256 == 0 && NW == 4), "unpool staging is a whole number of wave-DMAs");
    constexpr int N_RAW = UNPOOL ? UP_F + UP_F / 4 : WN_CH * PLANE;                 // floats staged per chunk
    constexpr int I_PER_WAVE = UNPOOL ? (UP_FQ + UP_AQ + NW - 1) / NW
                             : QUAD ? (N_RAW / 4 + 64 * NW - 1) / (64 * NW) : WN_CH * IN_ROWS / 4;   // wave-DMAs per wave (64 quads / 64 dwords each)
    constexpr int RAW = UNPOOL ? 1024 : QUAD ? I_PER_WAVE * NW * 256 : N_RAW;          // floats per raw buffer
    static_assert(QUAD || (WN_CH * IN_ROWS) % 4 == 0, "rows divide over the four waves");

    __shared__ __attribute__((aligned(16))) float raw_s[2][RAW];
    __shared__ __attribute__((aligned(16))) float v_s[2][TG][WN_V];
    __shared__ __attribute__((aligned(16))) float x8_own[W8 ? 8 * 8 * 64 * 4 : 4];     // W8: partial outputs for the partner, [wave][row][lane][4]
    static_assert(!H4 || 4 * 8 * 64 * 4 <= 2 * TG * WN_V, "H4: the exchange buffer fits the V images");
    float* const x8_s = H4 ? &v_s[0][0][0] : x8_own;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // PS: wave_g is the position half in the main loop and the tile group the wave finishes in the epilogue
    const int wave_m = W8 ? (wave & 3) : H4 ? (wave & 1) : wave / TG, wave_g = RS ? 0 : wave % TG;
    const int ph = W8 ? wave >> 2 : H4 ? wave >> 1 : wave_g;              // position half (SPLIT builds)

    // XCD-aware bijective block -> tile map, pixel tile fastest: the co-resident blocks of one XCD work on the
    // same channel slice of U (the dominant stream) and on neighbouring pixel tiles.
    const int nwg = gridDim.x / a.splits;
    const int split = blockIdx.x / nwg, orig = blockIdx.x - split * nwg;
    const int nch = a.nch / a.splits, c_first = split * nch;       // this workgroup's chunks of the K loop
    const int xcd = orig & 7, q = nwg >> 3, r = nwg & 7;
    const int logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
    const int n_pt = a.tiles_x * a.tiles_y;
    const int mt = logical / n_pt;
    const int pt = logical - mt * n_pt;
    const int tx = pt % a.tiles_x;
    const int ty = pt / a.tiles_x;
    const int y0 = ty * PROWS, x0 = tx * 32;
    const unsigned plane = (unsigned)a.H * a.W;
    const int nkp = a.K >> 1;

    const unsigned long long t_begin = DIAG ? __builtin_amdgcn_s_memtime() : 0ull;

    const __amdgpu_buffer_rsrc_t rs_i = __builtin_amdgcn_make_buffer_rsrc((void*)a.in, 0, a.in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc((void*)(UNPOOL ? a.unpool_amap : nullptr), 0, UNPOOL ? a.amap_bytes : 0u, 0x00020000);
    const unsigned pplane = UNPOOL ? (unsigned)a.ph * a.pw : 0u;
    unsigned ioff[QUAD ? I_PER_WAVE : 1];
    if (UNPOOL) {
        // pieces 0 .. UP_FQ - 1: quads of pooled floats; pieces UP_FQ .. UP_FQ + UP_AQ - 1: dwords of map bytes -- the same
        // (channel, row, column) geometry, 4 columns per lane either way
#pragma unroll
        for (int t = 0; t < I_PER_WAVE; ++t) {
            const int piece = wave + NW * t;
            const bool bytes = piece >= UP_FQ;
            const int e = ((bytes ? piece - UP_FQ : piece) * 64 + lane) * 4;          // element (float / byte) index in the staged image
            const int c = e / UP_PLANE;
            const int rem = e - c * UP_PLANE;
            const int rr = rem / UP_COLS;
            const int col = rem - rr * UP_COLS;
            const int gy = (y0 >> 1) - 1 + rr, gx = (x0 >> 1) - 4 + col;
            const bool ok = piece < UP_FQ + UP_AQ && gy >= 0 && gy < a.ph && gx >= 0 && gx + 3 < a.pw;
            // (BIG: the floats are addressed from the chunk's own base; the map bytes -- a quarter of the size -- keep the tensor's)
            ioff[t] = ok ? ((unsigned)((BIG && !bytes ? 0 : c_first * WN_CH) + c) * pplane + (unsigned)gy * a.pw + gx) * (bytes ? 1u : 4u) : kOOB;
        }
    } else if (QUAD) {
#pragma unroll
        for (int t = 0; t < I_PER_WAVE; ++t) {
            const int e = ((wave + NW * t) * 64 + lane) * 4;
            const int c = e / PLANE;
            const int rem = e - c * PLANE;
            const int rr = rem / IW;
            const int col = rem - rr * IW;
            const int gy = y0 - 1 + rr, gx = x0 - 4 + col;
            const bool ok = e < N_RAW && gy >= 0 && gy < a.H && gx >= 0 && gx + 3 < a.W;
            ioff[t] = ok ? ((unsigned)((BIG ? 0 : c_first * WN_CH) + c) * plane + (unsigned)gy * a.W + gx) * 4u : kOOB;
        }
    } else {
        const int gx = x0 - 1 + lane;                                  // lane = staged column
        ioff[0] = (lane < 34 && gx >= 0 && gx < a.W) ? (unsigned)gx * 4u : kOOB;
    }
    auto dma_raw = [&](int ch, int buf) {
        // BIG: this chunk's 8 planes behind a resource of their own (a.in_bytes = the bytes of ONE chunk)
        const __amdgpu_buffer_rsrc_t rs_c = BIG ? __builtin_amdgcn_make_buffer_rsrc(
            (void*)(a.in + (size_t)(c_first + ch) * WN_CH * (UNPOOL ? pplane : plane)), 0, a.in_bytes, 0x00020000) : rs_i;
        if (UNPOOL) {
#pragma unroll
            for (int t = 0; t < I_PER_WAVE; ++t) {
                const int piece = wave + NW * t;                           // wave-uniform
                if (piece < UP_FQ) {
                    const unsigned vo = (ioff[t] == kOOB || BIG) ? ioff[t] : ioff[t] + (unsigned)ch * WN_CH * pplane * 4u;
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_c, (lptr_t)(raw_s[buf] + piece * 256), 16, vo, 0, 0, 0);
                } else if (piece < UP_FQ + UP_AQ) {
                    const unsigned vo = ioff[t] == kOOB ? kOOB : ioff[t] + (unsigned)ch * WN_CH * pplane;
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a, (lptr_t)(raw_s[buf] + UP_F + (piece - UP_FQ) * 64), 4, vo, 0, 0, 0);
                }
            }
        } else if (QUAD) {
            const unsigned coff = BIG ? 0u : (unsigned)ch * WN_CH * plane * 4u;
#pragma unroll
            for (int t = 0; t < I_PER_WAVE; ++t) {
                const unsigned vo = ioff[t] == kOOB ? kOOB : ioff[t] + coff;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_c, (lptr_t)(raw_s[buf] + (wave + NW * t) * 256), 16, vo, 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int t = 0; t < I_PER_WAVE; ++t) {
                const int row = wave + 4 * t;                          // staged row: channel row / IN_ROWS, image row y0 - 1 + row % IN_ROWS
                const int c = row / IN_ROWS, gy = y0 - 1 + (row - c * IN_ROWS);
                const bool row_ok = gy >= 0 && gy < a.H;               // wave-uniform
                const unsigned base = ((unsigned)((c_first + ch) * WN_CH + c) * plane + (unsigned)(row_ok ? gy : 0) * a.W) * 4u;
                const unsigned vo = (row_ok && ioff[0] != kOOB) ? ioff[0] + base : kOOB;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_i, (lptr_t)(raw_s[buf] + row * 64), 4, vo, 0, 0, 0);
            }
        }
    };

    // input transform: this thread owns tile xt of channel xch, in every tile group, of every chunk.
    // V image of one (chunk, tile group): [k-pair 4][pos/4][k parity * 32 + tile][pos%4] -- the B operands of four
    // positions are one ds_read_b128, a transformed row is one ds_write_b128.
    const int xt = tid & 31, xch = (tid >> 5) & 7;       // W8: both position halves map onto the same 256 pairs; one of them works per chunk
    // UNPOOL: first of the tile's 3x3 pooling windows: staged row xt >> 4 (= pooled row y0/2 - 1 + ...), staged column (xt & 15) + 3
    const int x_raw = UNPOOL ? xch * UP_PLANE + (xt >> 4) * UP_COLS + (xt & 15) + 3
                             : xch * PLANE + (2 * (xt >> 4)) * IW + 2 * (xt & 15) + COL0;          // column COL0 = pixel x0 - 1
    const int x_v = (((xch >> 1) * 4) * 64 + (xch & 1) * 32 + xt) * 4;
    float d[TG][16];
    // Empty asm with the 16 values as in/out operands: arithmetic on them cannot be scheduled across it, which is
    // what keeps the transform inside its clump (sched_barrier alone orders only instructions with side effects).
    auto pin = [&]() {
#pragma unroll
        for (int g = 0; g < TG; ++g) {
            asm volatile("" : "+v"(d[g][0]), "+v"(d[g][1]), "+v"(d[g][2]), "+v"(d[g][3]), "+v"(d[g][4]), "+v"(d[g][5]), "+v"(d[g][6]), "+v"(d[g][7]));
            asm volatile("" : "+v"(d[g][8]), "+v"(d[g][9]), "+v"(d[g][10]), "+v"(d[g][11]), "+v"(d[g][12]), "+v"(d[g][13]), "+v"(d[g][14]), "+v"(d[g][15]));
        }
    };
    float up_v[UNPOOL ? 9 : 1];
    unsigned up_m[UNPOOL ? 9 : 1];
    // UNPOOL: window (r, c) of the 3x3 covers tile rows {0}, {1, 2}, {3} for r = 0, 1, 2 (the row's parity inside the window is 1, 0,
    // 1, 0) and the same along x; an element keeps the pooled diff iff the map byte is 4 | 2 * parity_y + parity_x
    auto up_expand = [&]() {
        if constexpr (UNPOOL) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int r = (i + 1) >> 1, cc = (j + 1) >> 1;
                    const unsigned want = 4u | (2u * ((i + 1) & 1)) | ((j + 1) & 1);
                    d[0][4 * i + j] = up_m[3 * r + cc] == want ? up_v[3 * r + cc] : 0.f;
                }
        }
    };
    auto xf_read = [&](const float* rp) {
        if constexpr (UNPOOL) {
            const unsigned char* bp = reinterpret_cast<const unsigned char*>(rp - x_raw + UP_F) + x_raw;      // same geometry, in bytes
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int cc = 0; cc < 3; ++cc) {
                    up_v[3 * r + cc] = rp[r * UP_COLS + cc];
                    up_m[3 * r + cc] = bp[r * UP_COLS + cc];
                }
            return;
        }
#pragma unroll
        for (int g = 0; g < TG; ++g)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) d[g][4 * i + j] = rp[(4 * g + i) * IW + j];
    };
    auto xf_math = [&]() {
#pragma unroll
        for (int g = 0; g < TG; ++g) {
            float wv[16];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                wv[j] = d[g][j] - d[g][8 + j]; wv[4 + j] = d[g][4 + j] + d[g][8 + j];
                wv[8 + j] = d[g][8 + j] - d[g][4 + j]; wv[12 + j] = d[g][4 + j] - d[g][12 + j];
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                d[g][4 * i] = wv[4 * i] - wv[4 * i + 2]; d[g][4 * i + 1] = wv[4 * i + 1] + wv[4 * i + 2];
                d[g][4 * i + 2] = wv[4 * i + 2] - wv[4 * i + 1]; d[g][4 * i + 3] = wv[4 * i + 1] - wv[4 * i + 3];
            }
        }
    };
    auto xf_write = [&](float* vp) {
#pragma unroll
        for (int g = 0; g < TG; ++g)
#pragma unroll
            for (int i = 0; i < 4; ++i)
                *reinterpret_cast<float4*>(vp + g * WN_V + i * 256) = make_float4(d[g][4 * i], d[g][4 * i + 1], d[g][4 * i + 2], d[g][4 * i + 3]);
    };

    const f32x4* up = reinterpret_cast<const f32x4*>(a.upack) + ((size_t)(mt * WM + wave_m) * nkp + 4 * c_first) * 256 + lane;
    constexpr int UQ = SPLIT ? 2 : 4;      // U quads (4 positions each) a wave needs per k-pair
    f32x4 ua[4][UQ];                    // U ring: set kp % 4 holds k-pair kp, refilled three k-pairs ahead
    auto u_fill = [&](int kp) {
#pragma unroll
        for (int g = 0; g < UQ; ++g) ua[kp & 3][g] = up[((size_t)kp * 4 + (SPLIT ? 2 * ph + g : g)) * 64];
        asm volatile("" ::: "memory");   // keeps the loads here (no folding into "load at use")
    };
    // makes the compiler wait for a whole U set at one place (inside a clump) instead of before each group of four MFMAs
    auto u_pin = [&](int set) {
        if constexpr (SPLIT) asm volatile("" : "+v"(ua[set][0]), "+v"(ua[set][1]));
        else asm volatile("" : "+v"(ua[set][0]), "+v"(ua[set][1]), "+v"(ua[set][2]), "+v"(ua[set][3]));
    };
    constexpr int NACC = RS ? 8 : 16;   // 32x32 accumulators per wave
    constexpr int NBQ = NACC / 4;
    f32x4 bq[2][NBQ];                   // B operands of two k-pairs, four positions per quad (PS: [tile group][quad of the half])
    auto b_fetch = [&](int buf, int kpl, int set) {
#pragma unroll
        for (int pg = 0; pg < NBQ; ++pg) {
            const float* vimg = PS ? v_s[buf][pg >> 1] : v_s[buf][wave_g];
            const int quad = PS ? 2 * ph + (pg & 1) : (RS ? 2 * ph + pg : pg);
            bq[set][pg] = *reinterpret_cast<const f32x4*>(vimg + ((kpl * 4 + quad) * 64 + lane) * 4);
        }
    };
    // one wait for the rest of a B set (its first quad is awaited by the k-pair's first MFMA) instead of one per quad
    auto b_pin = [&](int set) {
        if constexpr (RS) asm volatile("" : "+v"(bq[set][1]));
        else asm volatile("" : "+v"(bq[set][1]), "+v"(bq[set][2]), "+v"(bq[set][3]));
    };

    f32x16 acc[NACC];
#pragma unroll
    for (int p = 0; p < NACC; ++p)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[p][e] = 0.f;

    // ---- prologue: raw chunks 0 and 1, U of k-pairs 0..2, V of chunk 0 ----
    dma_raw(0, 0);
    if (nch > 1) dma_raw(1, 1);
    u_fill(0); u_fill(1); u_fill(2);
    __syncthreads();
    if (!W8 || ph == 0) {
        xf_read(raw_s[0] + x_raw);
        up_expand();
        xf_math();
        xf_write(&v_s[0][0][0] + x_v);
    }
    __syncthreads();
    b_fetch(0, 0, 0);

    unsigned long long t0 = 0, r0 = 0;
    if (DIAG) { t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }

    // One chunk = 4 k-pairs x 16 MFMAs.  With one wave per SIMD nothing this wave issues overlaps its own MFMAs
    // (measured: a bare MFMA stream runs at 64.0 cycles each, every instruction in between adds ~3 cycles plus ~9 per
    // interrupted MFMA pair), so the auxiliary work is kept minimal and CLUMPED: one clump per k-pair, right after
    // its first MFMA, then 15 MFMAs back to back.
    //   every clump : B operands of the next k-pair (4 ds_read_b128), U of k-pair +3 (4 global_load_dwordx4)
    //   k-pair 0    : + raw reads of chunk c+1 (issued, consumed one clump later), raw DMA of chunk c+2
    //   k-pair 1    : + transform arithmetic        k-pair 2: + V writes
    //   k-pair 3    : wait own V writes / raw DMA, s_barrier, then the fetch from the next chunk's V image
    auto chunk = [&](const int c, auto more_t, auto more2_t) {
        constexpr bool MORE = decltype(more_t)::value, MORE2 = decltype(more2_t)::value;
        const int cur = c & 1;
        const bool do_xf = !W8 || ph == ((c + 1) & 1);       // W8: the two position halves take turns with the input transform
#pragma unroll
        for (int kpl = 0; kpl < 4; ++kpl) {
            const int kp = 4 * c + kpl;
            const int set = kpl & 1;
#pragma unroll
            for (int p = 0; p < NACC; ++p) {
                // PS: accumulator p = tile group p / 8, position 8 * half + p % 8 (U quad (p / 4) % 2 of this wave's two)
                acc[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(ua[kpl][SPLIT ? (p >> 2) & 1 : p >> 2][p & 3], bq[set][p >> 2][p & 3], acc[p], 0, 0, 0);
                if (p == 0) {
                    __builtin_amdgcn_sched_barrier(0);
                    b_pin(set);
                    if (kpl < 3) b_fetch(cur, kpl + 1, set ^ 1);
                    else if (MORE) {
                        // own V writes done (lgkmcnt), own raw DMA landed (only the U loads of k-pairs 1 and 2 came after it)
                        if constexpr (SPLIT) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)\n\ts_barrier" ::: "memory");
                        else asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)\n\ts_barrier" ::: "memory");
                        b_fetch(cur ^ 1, 0, 0);
                    }
                    if (MORE || kpl == 0) u_fill(kp + 3);
                    if (MORE && do_xf) {
                        if (kpl == 0) xf_read(raw_s[cur ^ 1] + x_raw);
                        if (kpl == 1) { up_expand(); pin(); xf_math(); pin(); }
                        if (kpl == 2) xf_write(&v_s[cur ^ 1][0][0] + x_v);
                    }
                    if (MORE2 && kpl == 0) dma_raw(c + 2, cur);
                    if (MORE || kpl < 3) u_pin((kpl + 1) & 3);        // the next k-pair's U (loaded three k-pairs ago)
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    {
        using T = std::true_type; using F = std::false_type;
        int c = 0;
        for (; c + 2 < nch; ++c) chunk(c, T{}, T{});
        if (c + 1 < nch) { chunk(c, T{}, F{}); ++c; }
        chunk(c, F{}, F{});
    }

    unsigned long long t_loop_end = 0;
    if (DIAG) {
        const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
        t_loop_end = t1;
        if (tid == 0 && a.stamps) { a.stamps[2 * blockIdx.x] = t1 - t0; a.stamps[2 * blockIdx.x + 1] = r1 - r0; }
    }
    // Output transform of accumulator element e: the lane's 2x2 outputs y = A^T M A (M = the 4x4 transform-domain tile).
    // PS: this wave holds two rows of M (positions 8 ph .. 8 ph + 7) of BOTH tile groups; the transform is linear, so each
    // wave reduces its rows to a partial 2x2 output, hands the partial of the other tile group to its partner through LDS
    // (the V images are dead by now) and finishes its own group: y = partial(rows 0,1) + partial(rows 2,3).
    float* const xbuf = &v_s[0][0][0];                         // PS: [wave][e][lane][4] floats = 4 x 16 KiB
    // (the tile group is a compile-time constant: a run-time index into the accumulators would put them in scratch)
    // W8: same exchange between SIMD partners (waves w, w + 4), split by accumulator ROW instead of tile group: the wave
    // with position half ph finishes rows e = 8 ph .. 8 ph + 7; `e0_t` makes the row index a compile-time constant too.
    auto partial_out = [&](auto grp_t, auto e0_t, int el, float (&y)[4]) {    // SPLIT builds: this wave's two rows of M
        constexpr int grp = decltype(grp_t)::value, e0 = decltype(e0_t)::value;
        float t0[4], t1[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float ra = acc[(grp * 8 + j) % NACC][e0 + el], rb = acc[(grp * 8 + 4 + j) % NACC][e0 + el];      // rows 2 ph and 2 ph + 1 of M
            if (ph == 0) { t0[j] = ra + rb; t1[j] = rb; }                             // rows 0, 1:  tt0 = M0 + M1, tt1 = M1
            else { t0[j] = ra; t1[j] = -ra - rb; }                                    // rows 2, 3:  tt0 = M2, tt1 = -M2 - M3
        }
        y[0] = t0[0] + t0[1] + t0[2]; y[1] = t0[1] - t0[2] - t0[3];
        y[2] = t1[0] + t1[1] + t1[2]; y[3] = t1[1] - t1[2] - t1[3];
    };
    if constexpr (PS) {
        __syncthreads();                                       // every wave is done with the V images
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            float y[4];
            if (wave_g == 0) partial_out(std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{}, e, y);      // the group the partner finishes
            else partial_out(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, e, y);
            *reinterpret_cast<float4*>(xbuf + ((wave * 16 + e) * 64 + lane) * 4) = make_float4(y[0], y[1], y[2], y[3]);
        }
        __syncthreads();
    }
    if constexpr (RS) {
        if constexpr (H4) __syncthreads();                     // every wave is done with the V images (the exchange buffer)
#pragma unroll
        for (int el = 0; el < 8; ++el) {                       // the rows the partner finishes
            float y[4];
            if (ph == 0) partial_out(std::integral_constant<int, 0>{}, std::integral_constant<int, 8>{}, el, y);
            else partial_out(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, el, y);
            *reinterpret_cast<float4*>(x8_s + ((wave * 8 + el) * 64 + lane) * 4) = make_float4(y[0], y[1], y[2], y[3]);
        }
        __syncthreads();
    }
    auto out_xf = [&](int e, float& y00, float& y01, float& y10, float& y11) {
        if constexpr (RS) {                                    // e = row index WITHIN this wave's half (0..7)
            float y[4];
            if (ph == 0) partial_out(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, e, y);
            else partial_out(std::integral_constant<int, 0>{}, std::integral_constant<int, 8>{}, e, y);
            const float4 o = *reinterpret_cast<const float4*>(x8_s + (((wave ^ PARTNER) * 8 + e) * 64 + lane) * 4);
            y00 = y[0] + o.x; y01 = y[1] + o.y; y10 = y[2] + o.z; y11 = y[3] + o.w;
        } else if constexpr (PS) {
            float y[4];
            if (wave_g == 0) partial_out(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, e, y);
            else partial_out(std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{}, e, y);
            const float4 o = *reinterpret_cast<const float4*>(xbuf + (((wave ^ 1) * 16 + e) * 64 + lane) * 4);
            // rows 0,1 + rows 2,3, whichever wave adds them
            y00 = y[0] + o.x; y01 = y[1] + o.y; y10 = y[2] + o.z; y11 = y[3] + o.w;
        } else {
            float tt[2][4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                tt[0][j] = acc[j][e] + acc[4 + j][e] + acc[(8 + j) % NACC][e];
                tt[1][j] = acc[4 + j][e] - acc[(8 + j) % NACC][e] - acc[(12 + j) % NACC][e];
            }
            y00 = tt[0][0] + tt[0][1] + tt[0][2]; y01 = tt[0][1] - tt[0][2] - tt[0][3];
            y10 = tt[1][0] + tt[1][1] + tt[1][2]; y11 = tt[1][1] - tt[1][2] - tt[1][3];
        }
    };
  if (!QUAD) {
    // ---- epilogue for any width: the lane stores its own 2x2 tile, 8-byte accesses when W is even, 4-byte otherwise
    const int t31 = lane & 31, khalf = lane >> 5;
    const int gy0 = y0 + 4 * wave_g + 2 * (t31 >> 4), gx = x0 + 2 * (t31 & 15);
    const bool part = a.splits > 1;
    const bool has_bias = !part && a.bias != nullptr, has_mask = !part && a.mask_src != nullptr, has_inj = !part && a.inject != nullptr;
    const bool relu = !part && a.relu;
    float* const outp = part ? a.scratch + (size_t)split * a.M * plane : a.out;
    const bool live = gx < a.W && gy0 < a.H;
    const bool row1 = gy0 + 1 < a.H, col1 = gx + 1 < a.W, wide = (a.W & 1) == 0;     // W even: gx + 1 < W and 8-byte alignment
    const int mw = mt * BM + wave_m * 32 + 4 * khalf;
    const unsigned pix0 = live ? (unsigned)gy0 * a.W + gx : 0u, pix1 = (live && row1) ? pix0 + a.W : pix0;
    auto ld2 = [&](const float* base, unsigned off) -> float2 {
        if (wide) return *reinterpret_cast<const float2*>(base + off);
        return make_float2(base[off], col1 ? base[off + 1] : 0.f);
    };
#pragma unroll
    for (int eb = 0; eb < 4; ++eb) {
        const int mb = mw + 8 * eb;
        unsigned off[4];
        float2 mk[4][2], ij[4][2];
        float bs[4];
#pragma unroll
        for (int ee = 0; ee < 4; ++ee) {
            const int m = mb + ee < a.M ? mb + ee : a.M - 1;
            off[ee] = (unsigned)m * plane;
            bs[ee] = has_bias ? a.bias[m] : 0.f;
        }
        if (has_mask) {
#pragma unroll
            for (int ee = 0; ee < 4; ++ee) { mk[ee][0] = ld2(a.mask_src, off[ee] + pix0); mk[ee][1] = ld2(a.mask_src, off[ee] + pix1); }
        }
        if (has_inj) {
#pragma unroll
            for (int ee = 0; ee < 4; ++ee) { ij[ee][0] = ld2(a.inject, off[ee] + pix0); ij[ee][1] = ld2(a.inject, off[ee] + pix1); }
        }
#pragma unroll
        for (int ee = 0; ee < 4; ++ee) {
            const int e = 4 * eb + ee;
            float o[2][2];
            out_xf(e, o[0][0], o[0][1], o[1][0], o[1][1]);
            if (a.pool_out && live && mb + ee < a.M) {         // fused max-pool: windows clipped at the right / bottom edge
                float pm = o[0][0];
                if (col1) pm = pm > o[0][1] ? pm : o[0][1];
                if (row1) { pm = pm > o[1][0] ? pm : o[1][0]; if (col1) pm = pm > o[1][1] ? pm : o[1][1]; }
                pm += bs[ee];
                if (relu) pm = pm > 0.f ? pm : 0.f;
                a.pool_out[((size_t)(mb + ee) * a.pool_h + (gy0 >> 1)) * a.pool_w + (gx >> 1)] = pm;
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                float o0 = o[i][0] + bs[ee], o1 = o[i][1] + bs[ee];
                if (relu) { o0 = o0 > 0.f ? o0 : 0.f; o1 = o1 > 0.f ? o1 : 0.f; }
                if (has_mask) { o0 = mk[ee][i].x > 0.f ? o0 : 0.f; o1 = mk[ee][i].y > 0.f ? o1 : 0.f; }
                if (has_inj) { o0 += ij[ee][i].x; o1 += ij[ee][i].y; }
                if (live && mb + ee < a.M && (i == 0 || row1)) {
                    float* dst = outp + off[ee] + (i ? pix1 : pix0);
                    if (wide) *reinterpret_cast<float2*>(dst) = make_float2(o0, o1);
                    else { dst[0] = o0; if (col1) dst[1] = o1; }
                }
            }
        }
    }
  } else if constexpr (EPI != 0) {
    // ---- specialised epilogues (see EPI above): the generic one below, kind by kind, through buffer accesses
    constexpr bool E_FWD = EPI == 1 || EPI == 2, E_POOL = EPI == 2, E_MASK = EPI == 3;
    typedef unsigned wn_u32x4 __attribute__((ext_vector_type(4)));
    const int t31 = lane & 31, khalf = lane >> 5;
    const int odd = lane & 1;
    const int gy = y0 + 4 * wave_g + 2 * (t31 >> 4) + odd;          // this lane's row after the swap
    const int gx4 = x0 + 2 * (t31 & 14);                            // first of the pair's 4 pixels (16-byte aligned)
    const bool live = gx4 < a.W && gy < a.H;
    const int mw = mt * BM + wave_m * 32 + 4 * khalf;
    const unsigned pix = live ? (unsigned)gy * a.W + gx4 : 0u;
    const unsigned out_bytes = (unsigned)a.M * plane * 4u;
    const bool has_inj = a.inject != nullptr;
    const __amdgpu_buffer_rsrc_t rs_o = __builtin_amdgcn_make_buffer_rsrc((void*)a.out, 0, NOOUT ? 0u : out_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_mk = __builtin_amdgcn_make_buffer_rsrc((void*)a.mask_src, 0, E_MASK ? out_bytes : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_ij = __builtin_amdgcn_make_buffer_rsrc((void*)a.inject, 0, (!E_FWD && has_inj) ? out_bytes : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_bs = __builtin_amdgcn_make_buffer_rsrc((void*)a.bias, 0, E_FWD ? (unsigned)a.M * 4u : 0u, 0x00020000);
    // the fused pool: the lane's 2x2 tile IS one pooling window; a window clipped by the bottom edge keeps its first row only
    const int ty2 = y0 + 4 * wave_g + 2 * (t31 >> 4), tx2 = x0 + 2 * (t31 & 15);
    const bool plive = E_POOL && tx2 < a.W && ty2 < a.H, prow1 = ty2 + 1 < a.H;
    const unsigned pplane = E_POOL ? (unsigned)a.pool_h * a.pool_w : 0u;
    const unsigned ppix = plive ? (unsigned)(ty2 >> 1) * a.pool_w + (tx2 >> 1) : 0u;
    const __amdgpu_buffer_rsrc_t rs_po = __builtin_amdgcn_make_buffer_rsrc((void*)a.pool_out, 0, E_POOL ? (unsigned)a.M * pplane * 4u : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_pa = __builtin_amdgcn_make_buffer_rsrc((void*)a.pool_amap, 0, (E_POOL && a.pool_amap) ? (unsigned)a.M * pplane : 0u, 0x00020000);
    constexpr int NEB = RS ? 2 : 4;                         // H4: this wave finishes accumulator rows 8 ph .. 8 ph + 7
#pragma unroll
    for (int ebl = 0; ebl < NEB; ++ebl) {
        const int eb = RS ? 2 * ph + ebl : ebl;
        const int mb = mw + 8 * eb;                         // rows mb .. mb+3 (e = 4 eb + 0..3)
        unsigned off[4];
        float mk[E_MASK ? 4 : 1][4], ij[E_FWD ? 1 : 4][4], bs[E_FWD ? 4 : 1];
#pragma unroll
        for (int ee = 0; ee < 4; ++ee) {
            off[ee] = (live && mb + ee < a.M) ? ((unsigned)(mb + ee) * plane + pix) * 4u : kOOB;
            if constexpr (E_FWD) bs[ee] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_bs, (mb + ee) * 4, 0, 0));      // (beyond M: out of range, 0)
        }
        if constexpr (E_MASK) {
#pragma unroll
            for (int ee = 0; ee < 4; ++ee) {
                const uint4 v = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rs_mk, off[ee], 0, 0));
                mk[ee][0] = __builtin_bit_cast(float, v.x); mk[ee][1] = __builtin_bit_cast(float, v.y); mk[ee][2] = __builtin_bit_cast(float, v.z); mk[ee][3] = __builtin_bit_cast(float, v.w);
            }
        }
        if constexpr (!E_FWD) {
#pragma unroll
            for (int ee = 0; ee < 4; ++ee) {
                const uint4 v = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rs_ij, off[ee], 0, 0));
                ij[ee][0] = __builtin_bit_cast(float, v.x); ij[ee][1] = __builtin_bit_cast(float, v.y); ij[ee][2] = __builtin_bit_cast(float, v.z); ij[ee][3] = __builtin_bit_cast(float, v.w);
            }
        }
#pragma unroll
        for (int ee = 0; ee < 4; ++ee) {
            const int e = 4 * ebl + ee;                     // compile-time (H4: row within this wave's half)
            float y00, y01, y10, y11;
            out_xf(e, y00, y01, y10, y11);
            if constexpr (E_POOL) {
                const float b = bs[E_FWD ? ee : 0];
                float pm = y00 > y01 ? y00 : y01;
                const float p1 = y10 > y11 ? y10 : y11, pm2 = pm > p1 ? pm : p1;
                pm = prow1 ? pm2 : pm;
                const float qb = pm + b;
                const unsigned slot = y00 + b == qb ? 0u : (y01 + b == qb ? 1u : (y10 + b == qb ? 2u : 3u));
                const unsigned po = (plive && mb + ee < a.M) ? (unsigned)(mb + ee) * pplane + ppix : kOOB;
                __builtin_amdgcn_raw_buffer_store_b8((unsigned char)(slot | (qb > 0.f ? 4u : 0u)), rs_pa, po, 0, 0);
                pm += b;
                pm = pm > 0.f ? pm : 0.f;
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, pm), rs_po, po == kOOB ? kOOB : po * 4u, 0, 0);
            }
            if constexpr (NOOUT) continue;
            // give away the row this lane does not keep, receive the partner's part of the row it keeps (quad_perm 1,0,3,2)
            const float s0 = odd ? y00 : y10, s1 = odd ? y01 : y11;
            const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s0), 0xB1, 0xf, 0xf, true));
            const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s1), 0xB1, 0xf, 0xf, true));
            float o[4];
            o[0] = odd ? r0 : y00; o[1] = odd ? r1 : y01; o[2] = odd ? y10 : r0; o[3] = odd ? y11 : r1;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if constexpr (E_FWD) {
                    o[j] += bs[E_FWD ? ee : 0];
                    o[j] = o[j] > 0.f ? o[j] : 0.f;
                } else {
                    if constexpr (E_MASK) o[j] = mk[E_MASK ? ee : 0][j] > 0.f ? o[j] : 0.f;
                    const float oi = o[j] + ij[E_FWD ? 0 : ee][j];
                    o[j] = has_inj ? oi : o[j];
                }
            }
            wn_u32x4 ov;
#pragma unroll
            for (int j = 0; j < 4; ++j) ov[j] = __builtin_bit_cast(unsigned, o[j]);
            __builtin_amdgcn_raw_buffer_store_b128(ov, rs_o, off[ee], 0, 0);
        }
    }
  } else {
    // ---- epilogue: output transform (in-lane), then bias / ReLU / mask / inject.
    // A lane holds the 2x2 outputs of one tile; tiles of neighbouring lanes are neighbours in x.  Each lane PAIR swaps
    // one row (two DPP moves per accumulator element) so that the even lane owns row 0 and the odd lane row 1 of the
    // pair's 4 consecutive pixels: one 16-byte load / store per lane instead of two 8-byte ones -- the epilogue is
    // bound by vector-memory instruction issue (64 KiB per workgroup), not by arithmetic.
    // Four accumulator rows at a time: their mask / inject loads are issued together, ahead of the arithmetic.
    const int t31 = lane & 31, khalf = lane >> 5;
    const int odd = lane & 1;
    const int gy = y0 + 4 * wave_g + 2 * (t31 >> 4) + odd;          // this lane's row after the swap
    const int gx4 = x0 + 2 * (t31 & 14);                            // first of the pair's 4 pixels (16-byte aligned)
    // split-K: raw partial sums go to scratch[split]; bias / ReLU / mask / inject are applied by wino_combine_k
    const bool part = a.splits > 1;
    const bool has_bias = !part && a.bias != nullptr, has_mask = !part && a.mask_src != nullptr, has_inj = !part && a.inject != nullptr;
    const bool relu = !part && a.relu;
    float* const outp = part ? a.scratch + (size_t)split * a.M * plane : a.out;
    const bool live = gx4 < a.W && gy < a.H;
    const int mw = mt * BM + wave_m * 32 + 4 * khalf;
    const unsigned pix = live ? (unsigned)gy * a.W + gx4 : 0u;
    constexpr int NEB = RS ? 2 : 4;                         // W8 / H4: this wave finishes accumulator rows 8 ph .. 8 ph + 7
#pragma unroll
    for (int ebl = 0; ebl < NEB; ++ebl) {
        const int eb = RS ? 2 * ph + ebl : ebl;
        const int mb = mw + 8 * eb;                         // rows mb .. mb+3 (e = 4 eb + 0..3)
        unsigned off[4];
        float4 mk[4], ij[4];
        float bs[4];
#pragma unroll
        for (int ee = 0; ee < 4; ++ee) {
            const int m = mb + ee < a.M ? mb + ee : a.M - 1;
            off[ee] = (unsigned)m * plane + pix;
            bs[ee] = has_bias ? a.bias[m] : 0.f;
        }
        if (has_mask) {
#pragma unroll
            for (int ee = 0; ee < 4; ++ee) mk[ee] = *reinterpret_cast<const float4*>(a.mask_src + off[ee]);
        }
        if (has_inj) {
#pragma unroll
            for (int ee = 0; ee < 4; ++ee) ij[ee] = *reinterpret_cast<const float4*>(a.inject + off[ee]);
        }
#pragma unroll
        for (int ee = 0; ee < 4; ++ee) {
            const int e = 4 * ebl + ee;                     // compile-time (W8: row within this wave's half)
            float y00, y01, y10, y11;
            out_xf(e, y00, y01, y10, y11);
            if (a.pool_out) {
                // fused max-pool (Caffe MAX 2x2/2, ceil mode): the lane's 2x2 tile IS one pooling window (tile origins are
                // even); bias and ReLU commute with max.  A window clipped by the bottom edge keeps its first row only.
                const int ty2 = y0 + 4 * wave_g + 2 * (t31 >> 4), tx2 = x0 + 2 * (t31 & 15);
                if (tx2 < a.W && ty2 < a.H && mb + ee < a.M) {
                    const size_t po = ((size_t)(mb + ee) * a.pool_h + (ty2 >> 1)) * a.pool_w + (tx2 >> 1);
                    float pm = y00 > y01 ? y00 : y01;
                    if (ty2 + 1 < a.H) { const float p1 = y10 > y11 ? y10 : y11; pm = pm > p1 ? pm : p1; }
                    if constexpr (AMAP) {
                        // also record WHERE the maximum is (first one of a row-major scan; W % 4 == 0: the window has both columns) and
                        // whether it is positive after the bias: the pool's backward then reads this byte instead of both blobs
                        // (compared AFTER the bias, as the stored blob the reference's pooling layer scans is: two sums that round to the
                        // same float tie there, and the first one wins)
                        if (a.pool_amap) {
                            const float qb = pm + bs[ee];
                            const unsigned slot = y00 + bs[ee] == qb ? 0u : (y01 + bs[ee] == qb ? 1u : (y10 + bs[ee] == qb ? 2u : 3u));
                            a.pool_amap[po] = (unsigned char)(slot | (qb > 0.f ? 4u : 0u));
                        }
                    }
                    pm += bs[ee];
                    if (relu) pm = pm > 0.f ? pm : 0.f;
                    a.pool_out[po] = pm;
                }
            }
            if constexpr (NOOUT) continue;         // only the pooled blob and its arg-max map are wanted: nothing reads this one
            // give away the row this lane does not keep, receive the partner's part of the row it keeps (quad_perm 1,0,3,2)
            const float s0 = odd ? y00 : y10, s1 = odd ? y01 : y11;
            const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s0), 0xB1, 0xf, 0xf, true));
            const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s1), 0xB1, 0xf, 0xf, true));
            float o[4];
            o[0] = odd ? r0 : y00; o[1] = odd ? r1 : y01; o[2] = odd ? y10 : r0; o[3] = odd ? y11 : r1;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                o[j] += bs[ee];
                if (relu) o[j] = o[j] > 0.f ? o[j] : 0.f;
            }
            if (has_mask) {
                o[0] = mk[ee].x > 0.f ? o[0] : 0.f; o[1] = mk[ee].y > 0.f ? o[1] : 0.f;
                o[2] = mk[ee].z > 0.f ? o[2] : 0.f; o[3] = mk[ee].w > 0.f ? o[3] : 0.f;
            }
            if (has_inj) { o[0] += ij[ee].x; o[1] += ij[ee].y; o[2] += ij[ee].z; o[3] += ij[ee].w; }
            if (live && mb + ee < a.M) *reinterpret_cast<float4*>(outp + off[ee]) = make_float4(o[0], o[1], o[2], o[3]);
        }
    }
  }
    if (DIAG) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long t2 = __builtin_amdgcn_s_memtime();
        if (tid == 0 && a.stamps && blockIdx.x == 300)
            printf("[wino stamps] block 300: prologue %llu, loop %llu, epilogue (stores drained) %llu cycles\n", t0 - t_begin, t_loop_end - t0, t2 - t_loop_end);
    }
}

// (A two-waves-per-SIMD variant -- 8 waves, each keeping 8 of the 16 positions, U staged once in LDS, one barrier per
// k-pair -- was built and measured in round 1: 5 531 cycles per chunk against 4 888-5 324 for this kernel, because the
// per-k-pair barrier re-aligns the two waves of a SIMD and their auxiliary clumps then collide.  Removed)

__global__ __launch_bounds__(256, 1) void conv3x3_wino_f32_128x128(const WinoKArgs a) { conv3x3_wino_body<4, 1>(a); }
__global__ __launch_bounds__(256, 1) void conv3x3_wino_f32_64x256(const WinoKArgs a) { conv3x3_wino_body<2, 2>(a); }
__global__ __launch_bounds__(256, 1) void conv3x3_wino_f32_128x128_anyw(const WinoKArgs a) { conv3x3_wino_body<4, 1, 0, false>(a); }
__global__ __launch_bounds__(256, 1) void conv3x3_wino_f32_64x256_anyw(const WinoKArgs a) { conv3x3_wino_body<2, 2, 0, false>(a); }
__global__ __launch_bounds__(256, 1) void conv3x3_wino_f32_ps64x256(const WinoKArgs a) { conv3x3_wino_body<2, 2, 0, true, true>(a); }
__global__ __launch_bounds__(256, 1) void conv3x3_wino_f32_ps64x256_stamped(const WinoKArgs a) { conv3x3_wino_body<2, 2, 1, true, true>(a); }
__global__ __launch_bounds__(512, 2) void conv3x3_wino_f32_w8_128x128(const WinoKArgs a) { conv3x3_wino_body<4, 1, 0, true, false, true>(a); }
__global__ __launch_bounds__(512, 2) void conv3x3_wino_f32_w8_128x128_stamped(const WinoKArgs a) { conv3x3_wino_body<4, 1, 1, true, false, true>(a); }
__global__ __launch_bounds__(256, 2) void conv3x3_wino_f32_h4_64x128(const WinoKArgs a) { conv3x3_wino_body<2, 1, 0, true, false, false, true>(a); }
__global__ __launch_bounds__(256, 2) void conv3x3_wino_f32_h4_64x128_stamped(const WinoKArgs a) { conv3x3_wino_body<2, 1, 1, true, false, false, true>(a); }
__global__ __launch_bounds__(256, 1) void conv3x3_wino_f32_128x128_unpool(const WinoKArgs a) { conv3x3_wino_body<4, 1, 0, true, false, false, false, true>(a); }
__global__ __launch_bounds__(256, 2) void conv3x3_wino_f32_h4_64x128_unpool(const WinoKArgs a) { conv3x3_wino_body<2, 1, 0, true, false, false, true, true>(a); }
// forward launches that write only the pooled blob and its arg-max map
__global__ __launch_bounds__(256, 1) void conv3x3_wino_f32_128x128_noout(const WinoKArgs a) { conv3x3_wino_body<4, 1, 0, true, false, false, false, false, false, true>(a); }
__global__ __launch_bounds__(256, 2) void conv3x3_wino_f32_h4_64x128_noout(const WinoKArgs a) { conv3x3_wino_body<2, 1, 0, true, false, false, true, false, false, true>(a); }
// specialised epilogues of the six hot builds (EPI: 1 forward, 2 forward + pool, 3 masked data gradient, 4 unmasked data gradient)
#define ST2_WINO_EPI(NAME, BOUNDS, ...) __global__ __launch_bounds__(256, BOUNDS) void NAME(const WinoKArgs a) { conv3x3_wino_body<__VA_ARGS__>(a); }
ST2_WINO_EPI(conv3x3_wino_f32_128x128_fwd, 1, 4, 1, 0, true, false, false, false, false, false, false, 1)
ST2_WINO_EPI(conv3x3_wino_f32_128x128_pool, 1, 4, 1, 0, true, false, false, false, false, false, false, 2)
ST2_WINO_EPI(conv3x3_wino_f32_128x128_poolonly, 1, 4, 1, 0, true, false, false, false, false, false, true, 2)
ST2_WINO_EPI(conv3x3_wino_f32_128x128_dgm, 1, 4, 1, 0, true, false, false, false, false, false, false, 3)
ST2_WINO_EPI(conv3x3_wino_f32_128x128_dg, 1, 4, 1, 0, true, false, false, false, false, false, false, 4)
ST2_WINO_EPI(conv3x3_wino_f32_128x128_unpool_dgm, 1, 4, 1, 0, true, false, false, false, true, false, false, 3)
ST2_WINO_EPI(conv3x3_wino_f32_h4_64x128_fwd, 2, 2, 1, 0, true, false, false, true, false, false, false, 1)
ST2_WINO_EPI(conv3x3_wino_f32_h4_64x128_pool, 2, 2, 1, 0, true, false, false, true, false, false, false, 2)
ST2_WINO_EPI(conv3x3_wino_f32_h4_64x128_poolonly, 2, 2, 1, 0, true, false, false, true, false, false, true, 2)
ST2_WINO_EPI(conv3x3_wino_f32_h4_64x128_dgm, 2, 2, 1, 0, true, false, false, true, false, false, false, 3)
ST2_WINO_EPI(conv3x3_wino_f32_h4_64x128_dg, 2, 2, 1, 0, true, false, false, true, false, false, false, 4)
ST2_WINO_EPI(conv3x3_wino_f32_h4_64x128_unpool_dgm, 2, 2, 1, 0, true, false, false, true, true, false, false, 3)
#undef ST2_WINO_EPI
// tensors of 4 GiB and more (per-chunk buffer resources): the 128-channel and the half-tile kernel, plain and unpooling
__global__ __launch_bounds__(256, 1) void conv3x3_wino_f32_128x128_big(const WinoKArgs a) { conv3x3_wino_body<4, 1, 0, true, false, false, false, false, true>(a); }
__global__ __launch_bounds__(256, 2) void conv3x3_wino_f32_h4_64x128_big(const WinoKArgs a) { conv3x3_wino_body<2, 1, 0, true, false, false, true, false, true>(a); }
__global__ __launch_bounds__(256, 1) void conv3x3_wino_f32_128x128_unpool_big(const WinoKArgs a) { conv3x3_wino_body<4, 1, 0, true, false, false, false, true, true>(a); }
__global__ __launch_bounds__(256, 2) void conv3x3_wino_f32_h4_64x128_unpool_big(const WinoKArgs a) { conv3x3_wino_body<2, 1, 0, true, false, false, true, true, true>(a); }
__global__ __launch_bounds__(256, 1) void conv3x3_wino_f32_128x128_stamped(const WinoKArgs a) { conv3x3_wino_body<4, 1, 1>(a); }
__global__ __launch_bounds__(256, 1) void conv3x3_wino_f32_64x256_stamped(const WinoKArgs a) { conv3x3_wino_body<2, 2, 1>(a); }

// out = sum_s scratch[s] (+ bias) -> ReLU -> mask -> + inject, float4 per thread (plane % 4 == 0 because W % 4 == 0)
__global__ __launch_bounds__(256) void wino_combine_k(const float* __restrict__ scratch, int splits, const float* __restrict__ bias, int relu,
                                                      const float* __restrict__ mask_src, const float* __restrict__ inject,
                                                      float* __restrict__ out, int M, unsigned plane)
{
    const size_t n4 = (size_t)M * plane / 4, per = (size_t)M * plane;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        float4 v = *reinterpret_cast<const float4*>(scratch + 4 * i);
        for (int sp = 1; sp < splits; ++sp) {
            const float4 w = *reinterpret_cast<const float4*>(scratch + sp * per + 4 * i);
            v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w;
        }
        if (bias) { const float b = bias[(4 * i) / plane]; v.x += b; v.y += b; v.z += b; v.w += b; }
        if (relu) { v.x = v.x > 0.f ? v.x : 0.f; v.y = v.y > 0.f ? v.y : 0.f; v.z = v.z > 0.f ? v.z : 0.f; v.w = v.w > 0.f ? v.w : 0.f; }
        if (mask_src) {
            const float4 m = *reinterpret_cast<const float4*>(mask_src + 4 * i);
            v.x = m.x > 0.f ? v.x : 0.f; v.y = m.y > 0.f ? v.y : 0.f; v.z = m.z > 0.f ? v.z : 0.f; v.w = m.w > 0.f ? v.w : 0.f;
        }
        if (inject) { const float4 j = *reinterpret_cast<const float4*>(inject + 4 * i); v.x += j.x; v.y += j.y; v.z += j.z; v.w += j.w; }
        *reinterpret_cast<float4*>(out + 4 * i) = v;
    }
}

// the split-K combine pass, also for conv3x3_wino_split.hip (plane % 4 == 0)
hipError_t launch_wino_combine(const float* scratch, int splits, const float* bias, int relu, const float* mask_src, const float* inject,
                               float* out, int M, int H, int W, hipStream_t s)
{
    if (((size_t)H * W) % 4 != 0) return hipErrorInvalidValue;
    const size_t n4 = (size_t)M * H * W / 4;
    const unsigned grid = (unsigned)((n4 + 255) / 256 < 2048 ? (n4 + 255) / 256 : 2048);
    wino_combine_k<<<grid, 256, 0, s>>>(scratch, splits, bias, relu, mask_src, inject, out, M, (unsigned)(H * W));
    return hipGetLastError();
}

static int wino_default_variant(int M, int W);
static bool wino_variant_small(int variant);
static bool wino_needs_big(int K, int M, int H, int W);
bool conv_wino_ok(int K, int M, int H, int W);

// Split-K factor for a launch that would leave most CUs idle (conv5_1 at 1024^2: 128 workgroups on 256 CUs)
int conv_wino_splits(int K, int M, int H, int W)
{
    static const bool off = [] { const char* e = getenv("ST2_WINO_SPLITK"); return e && *e == '0'; }();
    if (off || !conv_wino_ok(K, M, H, W) || ((size_t)H * W) % 4 != 0) return 1;      // the combine pass works on float4
    int v = wino_default_variant(M, W);
    if (v == 3 && W % 4 != 0) v = 1;
    if (v == 6 && W % 4 != 0) v = 0;
    if ((v == 8 || v == 9) && W % 4 != 0) v = 1;
    const bool half = v == 8 || v == 9;
    const int bm = (wino_variant_small(v) || half) ? 64 : 128, prows = wino_variant_small(v) ? 8 : 4;
    const long long nblk = (long long)((W + 31) / 32) * ((H + prows - 1) / prows) * ((M + bm - 1) / bm);
    const int nch = K / WN_CH;
    int sp = 1;
    while (nblk * sp * 2 <= 256 && nch % (sp * 2) == 0 && nch / (sp * 2) >= 4 && sp < 16) sp *= 2;
    return sp;
}

// may this launch also write the max-pooled blob?  (forward epilogue only; not when K is split across workgroups)
bool conv_wino_can_pool(int K, int M, int H, int W)
{
    static const bool off = [] { const char* e = getenv("ST2_WINO_POOL"); return e && *e == '0'; }();
    return !off && conv_wino_ok(K, M, H, W) && conv_wino_splits(K, M, H, W) == 1;
}

// does a fused-pool launch of this shape also write the arg-max map (ConvProblem::pool_amap)?  The one-tile-group builds of aligned widths do.
bool conv_wino_pool_amap_ok(int K, int M, int H, int W)
{
    if (!conv_wino_can_pool(K, M, H, W) || W % 4 != 0 || H % 2 != 0) return false;
    const int v = wino_default_variant(M, W);
    return v == 0 || v == 6 || v == 8;
}

// may a forward launch of this shape with the fused pool skip its full-resolution blob (ConvProblem::out == nullptr)?
bool conv_wino_can_skip_out(int K, int M, int H, int W)
{
    if (!conv_wino_pool_amap_ok(K, M, H, W) || wino_needs_big(K, M, H, W)) return false;
    const int v = wino_default_variant(M, W);
    return v == 0 || v == 8;
}

// may a data-gradient launch of this shape read the pooled diff + the arg-max map of the pool above its input (ConvProblem::unpool_amap)?
// The 128-channel and the half-tile build have the variant; the staged pooled rows need W % 32 == 0 and H % 2 == 0.
bool conv_wino_can_unpool(int K, int M, int H, int W)
{
    const char* e = getenv("ST2_WINO_UNPOOL");               // =0: keep maxpool_bwd_amap_k (read per launch: the tests compare both)
    if ((e && *e == '0') || !conv_wino_ok(K, M, H, W) || W % 32 != 0 || H % 2 != 0) return false;
    const int v = wino_default_variant(M, W);
    return v == 0 || v == 8;                                 // (a pooled diff of 4 GiB or more: the BIG builds, conv_wino_ok has checked the sizes)
}

// a tensor of 4 GiB or more on either side: the BIG builds (aligned widths, the 128-channel or the half-tile variant) take it while one
// chunk of 8 planes stays below 4 GiB (32-bit byte offsets from the chunk's base) and the output below 2^32 elements (32-bit element offsets)
static bool wino_needs_big(int K, int M, int H, int W) { return 4ull * K * H * W >= 0xfffffff0ull || 4ull * M * H * W >= 0xfffffff0ull; }
static bool wino_big_ok(int K, int M, int H, int W)
{
    const char* e = getenv("ST2_WINO_BIG");                  // =0: refuse tensors >= 4 GiB as before (read per call)
    if (e && *e == '0') return false;
    const int v = wino_default_variant(M, W);
    (void)K;                                                 // (the input is addressed chunk by chunk: any depth)
    return W % 4 == 0 && (v == 0 || v == 8) && 32ull * H * W < 0xfffffff0ull && (unsigned long long)M * H * W <= 0x100000000ull;
}

bool conv_wino_ok(int K, int M, int H, int W)
{
    static const bool anyw = [] { const char* e = getenv("ST2_WINO_ANYW"); return !(e && *e == '0'); }();
    if (!(K >= 8 && K % 8 == 0 && (W % 4 == 0 || anyw) && W >= 1 && M >= 48 && H >= 1)) return false;
    return !wino_needs_big(K, M, H, W) || wino_big_ok(K, M, H, W);
}

// variant: 0 = 128 channels x 4x32 pixels, 1 = 64 channels x 8x32 pixels (tile groups split over the waves),
// 3 = 64 channels x 8x32 pixels with the POSITIONS split over the waves (half the U stream), -1 = choose;
// 6 = 128 channels x 4x32 pixels with EIGHT waves (two per SIMD, positions split between the partners);
// 8 = 64 channels x 4x32 pixels, positions split between two waves, two workgroups per CU;
// 2 / 5 / 4 / 7 / 9 = 0 / 1 / 3 / 6 / 8 with cycle stamps (diagnostic builds).
// p.wpack = the Winograd pack (pack_wino_weights_*); p.bias may be any length >= M
static int wino_default_variant(int M, int W)
{
    const char* fe = getenv("ST2_WINO_CFG");               // read per launch: the tests force every variant on every shape
    if (fe && *fe) return atoi(fe);
    const char* pe = getenv("ST2_WINO_PS");
    const bool ps = pe && *pe == '1';                      // measured (profiles/r02_c_*): no faster than variant 1 -- off unless asked for
    const int pad128 = (M + 127) / 128 * 128, pad64 = (M + 63) / 64 * 64;
    if (ps && W % 4 == 0) return 3;       // the any-width build of the position split would spill (it is not built)
    const char* w8 = getenv("ST2_WINO_W8");                  // measured (profiles/r02_d_*): main loop 2 % slower, epilogue 25 % faster,
    if (w8 && *w8 == '1' && W % 4 == 0 && pad128 <= pad64) return 6;   // layer times within 1 % of variant 0 -- off unless asked for
    if (pad64 < pad128) {
        // <= 64 useful channels per 128: the half tile with two workgroups per CU where the width allows it (measured,
        // profiles/r02_v_*: conv1_2 -10 %, conv2_1 dgrad -8 % against variant 1; within 3 % of variant 0 elsewhere)
        const char* h4 = getenv("ST2_WINO_H4");
        return (W % 4 == 0 && !(h4 && *h4 == '0')) ? 8 : 1;
    }
    return 0;
}
static bool wino_variant_small(int variant) { return variant == 1 || variant == 3 || variant == 4 || variant == 5; }

hipError_t launch_conv3x3_wino_cfg(const ConvProblem& p, int variant, hipStream_t s)
{
    if (!conv_wino_ok(p.K, p.M, p.H, p.W) || (reinterpret_cast<uintptr_t>(p.in) & 15) != 0) return hipErrorInvalidValue;
    const bool quad = p.W % 4 == 0;                  // else the any-width kernels (dword staging, narrower epilogue accesses)
    const bool forced_auto = variant < 0;           // split-K only on the automatic path
    if (variant < 0) variant = wino_default_variant(p.M, p.W);
    if (variant == 3 && !quad) variant = 1;
    if (variant == 6 && !quad) variant = 0;
    if ((variant == 8 || variant == 9) && !quad) variant = 1;
    if (variant > 9) return hipErrorInvalidValue;
    const bool small = wino_variant_small(variant), half = variant == 8 || variant == 9;
    const int bm = (small || half) ? 64 : 128, prows = small ? 8 : 4;
    WinoKArgs k{};
    k.in = p.in; k.upack = reinterpret_cast<const float4*>(p.wpack); k.bias = p.bias; k.out = p.out;
    k.mask_src = p.mask_src; k.inject = p.inject;
    k.K = p.K; k.M = p.M; k.H = p.H; k.W = p.W; k.nch = p.K / WN_CH;
    k.tiles_x = (p.W + 31) / 32; k.tiles_y = (p.H + prows - 1) / prows; k.n_mtiles = (p.M + bm - 1) / bm; k.relu = p.relu;
    bool big = wino_needs_big(p.K, p.M, p.H, p.W);
    if (big && (!wino_big_ok(p.K, p.M, p.H, p.W) || !(variant == 0 || variant == 8) || !quad)) return hipErrorInvalidValue;
    {   // test hook: ST2_WINO_FORCE_BIG=1 runs the BIG builds on tensors of any size (tests/test_gpu_winograd.py compares them bit for bit)
        const char* fb = getenv("ST2_WINO_FORCE_BIG");
        if (fb && *fb == '1' && quad && (variant == 0 || variant == 8) && wino_big_ok(p.K, p.M, p.H, p.W)) big = true;
    }
    k.in_bytes = big ? (unsigned)(4ull * WN_CH * p.H * p.W) : (unsigned)(4ull * p.K * p.H * p.W);      // BIG: the bytes of one chunk
    const bool unpool = p.unpool_amap != nullptr;
    if (unpool) {
        if (!(variant == 0 || variant == 8) || !quad || p.W % 32 != 0 || p.H % 2 != 0 || p.pool_out) return hipErrorInvalidValue;
        k.unpool_amap = p.unpool_amap; k.ph = p.H / 2; k.pw = p.W / 2;
        k.in_bytes = big ? (unsigned)(4ull * WN_CH * k.ph * k.pw) : (unsigned)(4ull * p.K * k.ph * k.pw);
        if ((unsigned long long)p.K * k.ph * k.pw >= 0xfffffff0ull) return hipErrorInvalidValue;
        k.amap_bytes = (unsigned)((unsigned long long)p.K * k.ph * k.pw);
    }
    k.u_bytes = (unsigned)(4ull * wino_pack_floats(p.K, p.M));
    k.stamps = p.stamps;
    const long long nblk = (long long)k.tiles_x * k.tiles_y * k.n_mtiles;
    if (nblk <= 0 || nblk > 0x7fffffffLL) return hipErrorInvalidValue;
    k.splits = 1; k.scratch = nullptr;
    k.pool_out = p.pool_out; k.pool_h = (p.H + 1) / 2; k.pool_w = (p.W + 1) / 2;
    k.pool_amap = (p.pool_out && quad && (variant == 0 || variant == 6 || variant == 8)) ? p.pool_amap : nullptr;
    if (forced_auto && p.scratch) {
        const int sp = conv_wino_splits(p.K, p.M, p.H, p.W);
        if (sp > 1 && p.scratch_floats >= (size_t)sp * p.M * p.H * p.W) { k.splits = sp; k.scratch = p.scratch; }
    }
    const bool stamped = variant == 2 || variant == 4 || variant == 5 || variant == 7 || variant == 9;
    if (stamped && (!quad || k.splits > 1)) return hipErrorInvalidValue;      // the stamped builds are quad-only, one pass
    if (k.splits > 1 && p.pool_out) return hipErrorInvalidValue;              // the caller asks conv_wino_can_pool() first
    // out == nullptr: only with the fused pool AND its arg-max map (variants 0 and 8 on aligned widths), one pass, no mask / inject
    const bool noout = !p.out;
    if (noout && (!k.pool_out || !k.pool_amap || k.splits > 1 || !quad || p.mask_src || p.inject || big || unpool || !(variant == 0 || variant == 8)))
        return hipErrorInvalidValue;
    const dim3 g((unsigned)(nblk * k.splits)), b(256);
    {   // the epilogue specialised for this launch's kind, where there is one (ST2_WINO_EPI=0: the generic epilogue; read per launch: the
        // tests compare both bit for bit)
        const char* ee = getenv("ST2_WINO_EPI");
        const bool fwd = p.bias && p.relu && !p.mask_src && !p.inject, dgr = !p.bias && !p.relu && !p.pool_out;
        if (!(ee && *ee == '0') && (variant == 0 || variant == 8) && quad && !big && k.splits == 1 && (fwd || dgr) && (!k.pool_out || k.pool_amap || !p.pool_amap)) {
            const bool v0 = variant == 0;
            bool done = true;
            if (noout) { if (v0) conv3x3_wino_f32_128x128_poolonly<<<g, b, 0, s>>>(k); else conv3x3_wino_f32_h4_64x128_poolonly<<<g, b, 0, s>>>(k); }
            else if (unpool) {
                if (dgr && p.mask_src) { if (v0) conv3x3_wino_f32_128x128_unpool_dgm<<<g, b, 0, s>>>(k); else conv3x3_wino_f32_h4_64x128_unpool_dgm<<<g, b, 0, s>>>(k); }
                else done = false;
            }
            else if (fwd && k.pool_out) { if (v0) conv3x3_wino_f32_128x128_pool<<<g, b, 0, s>>>(k); else conv3x3_wino_f32_h4_64x128_pool<<<g, b, 0, s>>>(k); }
            else if (fwd) { if (v0) conv3x3_wino_f32_128x128_fwd<<<g, b, 0, s>>>(k); else conv3x3_wino_f32_h4_64x128_fwd<<<g, b, 0, s>>>(k); }
            else if (p.mask_src) { if (v0) conv3x3_wino_f32_128x128_dgm<<<g, b, 0, s>>>(k); else conv3x3_wino_f32_h4_64x128_dgm<<<g, b, 0, s>>>(k); }
            else { if (v0) conv3x3_wino_f32_128x128_dg<<<g, b, 0, s>>>(k); else conv3x3_wino_f32_h4_64x128_dg<<<g, b, 0, s>>>(k); }
            if (done) return hipGetLastError();
        }
    }
    switch (variant) {
    case 0: if (noout) conv3x3_wino_f32_128x128_noout<<<g, b, 0, s>>>(k);
            else if (big) { if (unpool) conv3x3_wino_f32_128x128_unpool_big<<<g, b, 0, s>>>(k); else conv3x3_wino_f32_128x128_big<<<g, b, 0, s>>>(k); }
            else if (unpool) conv3x3_wino_f32_128x128_unpool<<<g, b, 0, s>>>(k);
            else if (quad) conv3x3_wino_f32_128x128<<<g, b, 0, s>>>(k); else conv3x3_wino_f32_128x128_anyw<<<g, b, 0, s>>>(k); break;
    case 1: if (quad) conv3x3_wino_f32_64x256<<<g, b, 0, s>>>(k); else conv3x3_wino_f32_64x256_anyw<<<g, b, 0, s>>>(k); break;
    case 3: conv3x3_wino_f32_ps64x256<<<g, b, 0, s>>>(k); break;
    case 2: conv3x3_wino_f32_128x128_stamped<<<g, b, 0, s>>>(k); break;
    case 4: conv3x3_wino_f32_ps64x256_stamped<<<g, b, 0, s>>>(k); break;
    case 6: conv3x3_wino_f32_w8_128x128<<<g, dim3(512), 0, s>>>(k); break;
    case 7: conv3x3_wino_f32_w8_128x128_stamped<<<g, dim3(512), 0, s>>>(k); break;
    case 8: if (noout) conv3x3_wino_f32_h4_64x128_noout<<<g, b, 0, s>>>(k);
            else if (big) { if (unpool) conv3x3_wino_f32_h4_64x128_unpool_big<<<g, b, 0, s>>>(k); else conv3x3_wino_f32_h4_64x128_big<<<g, b, 0, s>>>(k); }
            else if (unpool) conv3x3_wino_f32_h4_64x128_unpool<<<g, b, 0, s>>>(k); else conv3x3_wino_f32_h4_64x128<<<g, b, 0, s>>>(k);
            break;
    case 9: conv3x3_wino_f32_h4_64x128_stamped<<<g, b, 0, s>>>(k); break;
    default: conv3x3_wino_f32_64x256_stamped<<<g, b, 0, s>>>(k); break;
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess || k.splits == 1) return e;
    return launch_wino_combine(k.scratch, k.splits, p.bias, p.relu, p.mask_src, p.inject, p.out, p.M, p.H, p.W, s);      // (conv_wino_splits() declines planes that are no multiple of 4)
}

hipError_t launch_conv3x3_wino(const ConvProblem& p, hipStream_t s) { return launch_conv3x3_wino_cfg(p, -1, s); }

}  // namespace st2
