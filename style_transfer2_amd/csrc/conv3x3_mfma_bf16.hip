// conv3x3 (pad 1, stride 1) on the bf16 matrix cores: v_mfma_f32_32x32x16_bf16, fp32 accumulate.
// BASELINE config 3 ("bf16 features / fp32 Gram"): the operands of every conv (activations, weights, and the
// diffs of the backward pass) are bf16; accumulation, bias/ReLU, ReLU-mask, injected diffs and everything outside
// the convs (Gram, losses, TV, optimizer) stay fp32.  Each launch reads a bf16 channel-blocked copy of its input
//      act16 [C/8][H][W][8]            (8 consecutive channels of one pixel = one 16-byte quad)
// and writes the fp32 NCHW blob the rest of the engine uses plus, optionally, the bf16 copy for the next conv.
// "Lean" launches (engine.cpp, bf16 objective evaluation) cut the HBM traffic that bounds the 64- and 128-channel layers:
// the fp32 blob / fp32 diff is not written when its only consumer is another bf16 conv (out == nullptr), the ReLU mask
// of the backward pass is read from the bf16 copy (mask16: 2 bytes instead of 4), and the forward epilogue can pool its
// own output (Caffe MAX 2x2/2, ceil mode, first-max) into the bf16 copy the next conv reads plus a one-byte-per-element
// arg-max map (bits 0-1 = window slot, bit 2 = the maximum is positive) that replaces the blob in the pool backward.
// Round 4: forward and data-gradient epilogues are separate builds (conv16_body<..., DG, MB>); forward launches also write a 1-bit
// sign map of their post-ReLU output (bits_out) that the data gradient above masks with (mask_bits); the data gradient below a
// pool can take the POOLED diff and expand it in its staged tile (UNPOOL); K <= 64 launches use one staging buffer (SB).
//
// Same pipeline as conv3x3_mfma.hip (LDS-DMA with buffer descriptors, double-buffered LDS, one barrier per chunk,
// pinned issue order); what changes is the operand shape: K = 16 channels per MFMA, lane (l&31, l>>5) holds 8
// consecutive channels (block l>>5 of the chunk) of output row / pixel l&31, so every fragment is ONE ds_read_b128:
//      w_s  [9 taps][2 blocks][BM][8 bf16]      in_s [2 blocks][ROWS+2][34 px][8 bf16]
// A chunk is 16 input channels x 9 taps = 9 k-steps; three operand register sets rotate (9 is odd).
#include "st2_kernels.h"
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <string.h>
#include <type_traits>

#ifndef CONV16_BR
#define CONV16_BR 1          // 0: every tile on the tap-by-tap loop (development A/B: build a second library, ST2_HIP_LIB)
#endif

namespace st2 {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lptr_t;

constexpr int NT = 256;
constexpr int PXW = 34;                       // staged pixels per row: x0-1 .. x0+32
constexpr unsigned kOOB16 = 0xffffffffu;
// A 16-byte buffer store whose soffset is an SGPR must not be followed by a VALU write of its data registers: measured on gfx950 (round 5),
// buffer_store_dwordx4 v[80:83], v116, s[20:23], s35 offen  followed by  v_pk_add_f32 v[80:81], ...  stored the NEW v81 in lanes 12 .. 15 of
// every 16 -- the hazard of the ISA manual's wait-state table (store data of more than 64 bits), which hipcc (ROCm 7.2) guards only when
// soffset is NOT a register.  So the per-launch-kind epilogues add the wave-uniform part of the offset into the vector offset and pass
// soffset = 0 (the guarded form; one v_add per store).  The "do not write" offset of such a store is 2 GiB: above every bf16 buffer of a
// launch (their fp32 twin is checked to be < 4 GiB) and, unlike kOOB16, still out of range after the uniform part is added.
constexpr unsigned kOOBStore = 0x80000000u;

static unsigned short f2bf(float f)           // round-to-nearest-even, host side
{
    unsigned u;
    memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);   // NaN stays NaN
    return (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}

size_t conv16_pack_elems(int K, int M) { return (size_t)((K + 15) / 16) * 9 * 2 * conv_mpad(M) * 8; }

// packed[ch][tap][h][m][j] = w[m][ch*16 + h*8 + j][tap]
void pack_conv_weights16_fwd(const float* w, int Cout, int Cin, unsigned short* dst)
{
    const int mpad = conv_mpad(Cout);
    memset(dst, 0, conv16_pack_elems(Cin, Cout) * sizeof(unsigned short));
    for (int m = 0; m < Cout; ++m)
        for (int k = 0; k < Cin; ++k)
            for (int tap = 0; tap < 9; ++tap)
                dst[((((size_t)(k / 16) * 9 + tap) * 2 + (k % 16) / 8) * mpad + m) * 8 + (k % 8)] = f2bf(w[((size_t)m * Cin + k) * 9 + tap]);
}

// data gradient: m = Cin, k = Cout, taps flipped
void pack_conv_weights16_dgrad(const float* w, int Cout, int Cin, unsigned short* dst)
{
    const int mpad = conv_mpad(Cin);
    memset(dst, 0, conv16_pack_elems(Cout, Cin) * sizeof(unsigned short));
    for (int k = 0; k < Cout; ++k)
        for (int m = 0; m < Cin; ++m)
            for (int tap = 0; tap < 9; ++tap)
                dst[((((size_t)(k / 16) * 9 + tap) * 2 + (k % 16) / 8) * mpad + m) * 8 + (k % 8)] = f2bf(w[((size_t)k * Cin + m) * 9 + (8 - tap)]);
}

struct Conv16KArgs {
    const unsigned short* in16; const unsigned short* wpack; const float* bias; float* out; unsigned short* out16;
    const float* mask_src; const float* inject; const unsigned short* mask16;
    unsigned short* pool16; float* pool32; unsigned char* amap; int pool_h, pool_w;
    int K, M, MPad, H, W, nch, tiles_x, tiles_y, n_mtiles, relu;
    unsigned in_bytes, w_bytes;
    // fused style term (data-gradient launches): out = mask(conv) + D' @ F, F = the bf16 copy of the blob this launch differentiates
    const unsigned short* s_in16; const unsigned short* s_wpack; int s_nch; unsigned s_in_bytes, s_w_bytes;
    // UNPOOL builds (data-gradient launches directly below a max-pool): in16 is the POOLED diff [K/8][up_h][up_w][8] and up_amap the
    // pool's arg-max map ([K/8][up_h][up_w][8] bytes: slot | positive << 2); the staged activation tile is expanded in LDS
    const unsigned char* up_amap; int up_h, up_w;
    // sign maps (Conv16Problem::bits_out / mask_bits): [M / 32][H * W][2] 16-bit words in the accumulator layout
    unsigned short* bits_out; const unsigned short* mask_bits;
    int diag_stagger;                    // DIAG builds only: n > 0 = the first-generation workgroups in odd wave slots sleep n x 3.4 us before starting
    int diag_nodma;                      // DIAG builds only: 1 = the main loop issues no staging loads (multiplies stale LDS: the loop's own pace)
    unsigned long long* stamps;          // DIAG builds only (tools/probes): per workgroup {start, first chunk landed, main loop done, end} in 100 MHz ticks + the shader-clock counter at the two middle points
};

// v if bit `bit` of `word` is set, else +0: the one-bit field sign-extended (v_bfe_i32: 0 or ~0) and-ed onto the value -- two VALU
// operations and no condition register (128 selects in a row otherwise queue up on SGPR pairs and spill)
__device__ __forceinline__ float keep_if_bit(float v, unsigned word, int bit)
{
    const int m = __builtin_amdgcn_sbfe((int)word, bit, 1);
    return __builtin_bit_cast(float, __builtin_bit_cast(int, v) & m);
}

// bit e of the result: 16-bit field e of (u0.x, u0.y, u1.x, u1.y) is non-zero (eight bf16 values -> one byte of a sign map)
__device__ __forceinline__ unsigned nonzero_halves16(uint2 u0, uint2 u1)
{
    auto two = [](unsigned w) { return ((w & 0xffffu) ? 1u : 0u) | ((w >> 16) ? 2u : 0u); };
    return two(u0.x) | (two(u0.y) << 2) | (two(u1.x) << 4) | (two(u1.y) << 6);
}

// SB = true: ONE staging buffer instead of two.  A short reduction (K <= 64: two to four chunks) never reaches the steady state
// the double buffer is built for -- the workgroup waits for its first chunk at HBM latency with nothing to overlap -- so the
// short-K launches trade the second buffer for occupancy: 29 KiB of LDS and 120 registers per workgroup (forward and sign-map
// data-gradient builds), FOUR workgroups per CU, and one workgroup's DMA wait runs under the other three's MFMAs.
// UNPOOL = true (round 4; the double-buffered pipeline only): the launch differentiates the conv directly below a max-pool and reads
// the POOLED diff instead of the full-resolution one maxpool_bwd_idx16_k would have written.  Every staged quad (8 channels of one
// full-resolution pixel) is fetched from its pooling window's quad of the pooled diff -- the four pixels of a window fetch the same
// 16 bytes, served by the caches -- and, once a chunk has landed, each lane passes the quads it fetched through the window's eight
// arg-max bytes: a channel keeps its value iff its byte says "this position, maximum positive" (exactly maxpool_bwd_idx16_k's rule),
// in LDS, before any wave reads operands.  Gone: that kernel, its full-resolution output and this launch's read of it.
// DG: which epilogue the build carries (round 4: one epilogue with every option of both directions ran out of registers).
//   DG = false, "forward":        bias, ReLU, fp32 / bf16 outputs, the fused max-pool (pool16 / pool32 / amap), the sign map (bits_out)
//   DG = true,  "data gradient":  ReLU mask of the blob below (mask_src / mask16 / mask_bits), inject, fp32 / bf16 outputs, the fused style term
// (a launch with none of these options runs on the forward build)
// MB (data-gradient builds): the ReLU mask comes as a sign map (mask_bits) -- the other two forms have their own build, for the same reason
// EPI (round 5): 0 = the epilogue with every option of its direction behind run-time flags; 1 .. 3 = the epilogue of ONE launch kind of
// the lean flow (see "one epilogue per launch kind" below) -- at K = 64 the general epilogue ran as long as the main loop
template <int BM, int ROWS, int WAVES_M, int WAVES_N, bool SB = false, bool UNPOOL = false, bool DG = UNPOOL, bool MB = false, bool DIAG = false, int EPI = 0>
__device__ __forceinline__ void conv16_body(const Conv16KArgs& a)
{
    // BR (round 5): the four-rows-per-wave tile walks a chunk column by column of the 3x3 stencil and keeps the activation fragments of its
    // six input rows in registers -- see "activation fragments reused" at the main loop
    constexpr bool BR = CONV16_BR && !SB && !UNPOOL && ROWS / WAVES_N == 4 && BM / WAVES_M / 32 == 2;
    unsigned long long t_start = 0, t_first = 0, t_loop = 0, c_first = 0, c_loop = 0;      // 100 MHz ticks / shader cycles
    if constexpr (DIAG) {
        if (a.diag_stagger > 0 && blockIdx.x < 2 * 256) {       // does a convoy of co-resident workgroups cost anything?  offset their phases once
            const unsigned slot = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 4);      // HW_ID.wave_id
            if (slot & 1) for (int i = 0; i < a.diag_stagger; ++i) __builtin_amdgcn_s_sleep(127);
        }
        t_start = __builtin_amdgcn_s_memrealtime();
    }
    static_assert(DG || !MB, "sign-map masks are a data-gradient option");
    static_assert(!(SB && UNPOOL), "the unpooling build uses the double-buffered pipeline");
    static_assert(DG || !UNPOOL, "unpooling is a data-gradient option");
    constexpr int TM = BM / WAVES_M / 32;
    constexpr int TN = ROWS / WAVES_N;
    static_assert(WAVES_M * WAVES_N == 4 && TM >= 1 && TN >= 1, "tile");
    constexpr int IN_ROWS = ROWS + 2;
    constexpr int W_QUADS = 9 * 2 * BM;                          // 16-byte units of the weight slab
    constexpr int I_QUADS = 2 * IN_ROWS * PXW;
    constexpr int W_INSTR = W_QUADS / 64;
    constexpr int I_INSTR = (I_QUADS + 63) / 64;
    constexpr int I_QUADS_PAD = I_INSTR * 64;
    constexpr int BUF_Q = W_QUADS + I_QUADS_PAD;                 // quads per LDS buffer
    constexpr int W_PER_WAVE = (W_INSTR + 3) / 4;
    constexpr int I_PER_WAVE = (I_INSTR + 3) / 4;
    constexpr int NPIECE = W_PER_WAVE + I_PER_WAVE;
    constexpr int NSTEP = 9;
    constexpr int PPS = (NPIECE + NSTEP - 2) / (NSTEP - 1);
    static_assert(W_QUADS % 64 == 0, "weight slab is a whole number of 1-KiB DMA pieces");

    __shared__ __attribute__((aligned(16))) uint4 smem[(SB ? 1 : 2) * BUF_Q];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave_m = wave / WAVES_N, wave_n = wave % WAVES_N;

    const int nwg = gridDim.x, orig = blockIdx.x;
    const int xcd = orig & 7, q = nwg >> 3, r = nwg & 7;
    const int logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
    const int mt = logical % a.n_mtiles;
    const int pt = logical / a.n_mtiles;
    const int tx = pt % a.tiles_x, ty = pt / a.tiles_x;
    const int m0 = mt * BM, y0 = ty * ROWS, x0 = tx * 32;
    const unsigned plane = (unsigned)a.H * a.W;

    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)a.wpack, 0, a.w_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_i = __builtin_amdgcn_make_buffer_rsrc((void*)a.in16, 0, a.in_bytes, 0x00020000);

    const unsigned pplane = UNPOOL ? (unsigned)a.up_h * a.up_w : 0u;      // quads per channel block of the pooled diff
    auto full_res_offset = [&](int t, int ln) -> unsigned {      // byte offset of staged quad (wave + 4 t) * 64 + ln in a [blocks][H][W] quad tensor
        const int f = (wave + 4 * t) * 64 + ln;                  // quad index in the activation image
        const int h = f / (IN_ROWS * PXW);
        const int rem = f - h * (IN_ROWS * PXW);
        const int rr = rem / PXW, col = rem - rr * PXW;
        const int gy = y0 - 1 + rr, gx = x0 - 1 + col;
        const bool ok = f < I_QUADS && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
        return ok ? ((unsigned)h * plane + (unsigned)gy * a.W + gx) * 16u : kOOB16;
    };
    unsigned ioff[I_PER_WAVE];
    unsigned up_here = 0;                                        // UNPOOL: 3 bits per staged quad: 4 | 2 (y & 1) | (x & 1) -- what the arg-max byte must say
#pragma unroll
    for (int t = 0; t < I_PER_WAVE; ++t) {
        if constexpr (UNPOOL) {
            const int f = (wave + 4 * t) * 64 + lane;
            const int h = f / (IN_ROWS * PXW);
            const int rem = f - h * (IN_ROWS * PXW);
            const int rr = rem / PXW, col = rem - rr * PXW;
            const int gy = y0 - 1 + rr, gx = x0 - 1 + col;
            const bool ok = f < I_QUADS && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
            ioff[t] = ok ? ((unsigned)h * pplane + (unsigned)(gy >> 1) * a.up_w + (gx >> 1)) * 16u : kOOB16;
            up_here |= (4u | (2u * (gy & 1)) | (gx & 1)) << (3 * t);
        } else ioff[t] = full_res_offset(t, lane);
    }
    static_assert(!UNPOOL || 3 * I_PER_WAVE <= 32, "the window positions of a lane's quads fit one register");
    unsigned woff[W_PER_WAVE];
#pragma unroll
    for (int t = 0; t < W_PER_WAVE; ++t) {
        const int f = (wave + 4 * t) * 64 + lane;                // quad index in the weight slab: [tap][h][m]
        const int th = f / BM, m = f - th * BM;
        woff[t] = ((unsigned)th * a.MPad + m) * 16u;
    }

    auto dma_piece = [&](int t, int ch, int buf) {
        uint4* dst = smem + buf * BUF_Q;
        if (t < W_PER_WAVE) {
            const int i = wave + 4 * t;
            if (W_INSTR % 4 == 0 || i < W_INSTR) {
                const unsigned coff = ((unsigned)ch * 18u * a.MPad + m0) * 16u;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lptr_t)(dst + i * 64), 16, woff[t] + coff, 0, 0, 0);
            }
        } else {
            const int u = t - W_PER_WAVE;
            const int j = wave + 4 * u;
            if (I_INSTR % 4 == 0 || j < I_INSTR) {
                const unsigned coff = (unsigned)ch * 2u * (UNPOOL ? pplane : plane) * 16u;
                const unsigned vo = ioff[u] == kOOB16 ? kOOB16 : ioff[u] + coff;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_i, (lptr_t)(dst + W_QUADS + j * 64), 16, vo, 0, 0, 0);
            }
        }
    };
    // UNPOOL: the expansion of a staged chunk, by the lane that fetched each quad -- so it needs that lane's own loads back, not a
    // barrier.  The arg-max bytes (8 per quad, from the caches: the four pixels of a window share them) are requested two MFMA steps
    // ahead (unpool_prefetch); at the last step, the wave's pieces having landed (unpool_landed), each lane reads its quads back from
    // LDS, keeps the channels whose byte says "this position, maximum positive" and writes them in place (unpool_quad); the chunk's
    // barrier follows as in the plain build.  (Measured, 2048^2 bf16: the four launches cost 0.16 ms more and maxpool_bwd_idx16_k's
    // 0.28 ms are gone; spreading the quads over the MFMAs of steps 6 and 7 instead was slower, +0.22 .. 0.33 ms.)
    // Per 4 channels: t = (bytes ^ here) & 7 per byte is 0 iff kept; (t + 7) bit 3 = "not kept"; bytes 0/1 -> halves of two words.
    uint2 mm[UNPOOL ? I_PER_WAVE : 1];
    auto unpool_prefetch = [&](int ch) {
        if constexpr (UNPOOL) {
            const unsigned coff = (unsigned)ch * 2u * pplane * 8u;
#pragma unroll
            for (int u = 0; u < I_PER_WAVE; ++u) {
                const bool live = (I_INSTR % 4 == 0 || wave + 4 * u < I_INSTR) && ioff[u] != kOOB16;
                mm[u] = live ? *reinterpret_cast<const uint2*>(a.up_amap + (size_t)(ioff[u] >> 1) + coff) : make_uint2(0u, 0u);
            }
        }
    };
    auto unpool_landed = [&]() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); };      // this lane's DMA pieces of the chunk and its arg-max bytes
    auto unpool_quad = [&](int u, int buf) {
        if constexpr (UNPOOL) {
            uint4* img = smem + buf * BUF_Q + W_QUADS;
            const int j = wave + 4 * u;
            if ((I_INSTR % 4 == 0 || j < I_INSTR) && ioff[u] != kOOB16) {          // (quads outside the image are zero already)
                const unsigned here = ((up_here >> (3 * u)) & 7u) * 0x01010101u;
                const uint4 d = img[j * 64 + lane];
                auto keep = [&](unsigned bytes) {                // 0xff per byte whose low three bits equal `here`
                    const unsigned t = (bytes ^ here) & 0x07070707u;
                    return ((((t + 0x07070707u) >> 3) & 0x01010101u) ^ 0x01010101u) * 0xffu;
                };
                const unsigned k0 = keep(mm[u].x), k1 = keep(mm[u].y);
                uint4 o;
                o.x = d.x & __builtin_amdgcn_perm(0u, k0, 0x01010000u);
                o.y = d.y & __builtin_amdgcn_perm(0u, k0, 0x03030202u);
                o.z = d.z & __builtin_amdgcn_perm(0u, k1, 0x01010000u);
                o.w = d.w & __builtin_amdgcn_perm(0u, k1, 0x03030202u);
                img[j * 64 + lane] = o;
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

    const int khalf = lane >> 5, l31 = lane & 31;
    const int a_off = khalf * BM + wave_m * (TM * 32) + l31;                                   // + tap*2*BM + i*32
    const int b_off = W_QUADS + khalf * IN_ROWS * PXW + (wave_n * TN) * PXW + l31;             // + (j+dy)*PXW + dx

    // Every build adds the nine taps of a chunk in the SAME order -- column by column of the stencil (dx = 0: dy = 0, 1, 2; dx = 1: ...) --
    // so that the tile configurations stay bit-identical to each other (the tests compare them): step s of a chunk is tap 3 (s % 3) + s / 3
    auto tap_of = [](int s2) { return (s2 % 3) * 3 + s2 / 3; };
    bf16x8 av[BR ? 1 : 3][TM], bv[BR ? 1 : 3][TN];
    auto fetch = [&](const uint4* base, int tap, bf16x8 (&ao)[TM], bf16x8 (&bo)[TN]) {
        const int dy = tap / 3, dx = tap % 3;
#pragma unroll
        for (int i = 0; i < TM; ++i) ao[i] = __builtin_bit_cast(bf16x8, base[a_off + tap * 2 * BM + i * 32]);
#pragma unroll
        for (int j = 0; j < TN; ++j) bo[j] = __builtin_bit_cast(bf16x8, base[b_off + (j + dy) * PXW + dx]);
    };

#pragma unroll
    for (int t = 0; t < NPIECE; ++t) dma_piece(t, 0, 0);
  if constexpr (SB) {
    for (int ch = 0; ch < a.nch; ++ch) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                         // chunk ch has landed
        if constexpr (DIAG) { if (ch == 0) { t_first = __builtin_amdgcn_s_memrealtime(); c_first = __builtin_amdgcn_s_memtime(); } }
        fetch(smem, tap_of(0), av[0], bv[0]);
#pragma unroll
        for (int s2 = 0; s2 < NSTEP; ++s2) {
#pragma unroll
            for (int ij = 0; ij < TM * TN; ++ij) {
                const int i = ij / TN, j = ij % TN;
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[s2 % 3][i], bv[s2 % 3][j], acc[i][j], 0, 0, 0);
                if (ij == 0 && s2 + 1 < NSTEP) {
                    __builtin_amdgcn_sched_barrier(0);
                    fetch(smem, tap_of(s2 + 1), av[(s2 + 1) % 3], bv[(s2 + 1) % 3]);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (ch + 1 < a.nch) {
            __syncthreads();                                     // every wave has read its last operands of chunk ch
#pragma unroll
            for (int t = 0; t < NPIECE; ++t) if (!(DIAG && a.diag_nodma)) dma_piece(t, ch + 1, 0);
        }
    }
  } else if constexpr (BR) {
    // ---- activation fragments reused (round 5).  The tap-by-tap loop reads TM + TN fragments from LDS per TM * TN MFMAs -- 0.75 ds_read_b128
    // per MFMA on this tile, and a wave's ds_read_b128 holds the CU's LDS pipe for 8 clocks: four waves keep it busy 3/4 of the time their
    // MFMAs take, which is what held the deep layers at ~0.57 of the matrix-core peak.  But tap (dy, dx) of output row j reads input row
    // j + dy: a wave's four output rows touch six input rows, 18 (row, dx) fragments per chunk instead of 36 (tap, row).  So a chunk runs
    // as three phases, one per stencil COLUMN dx: the six row fragments of that column live in registers (bfr), group r of a phase issues
    // every MFMA that reads row r -- (dy, j = r - dy) for the dy that give a valid j, both 32-channel groups -- and then refills bfr[r]
    // with the next phase's fragment (its next use is a whole phase away).  The weight fragments of a phase (3 dy x TM) are loaded one
    // per group during the phase before, into the next of three register sets (three phases per chunk: the set index is static).
    // 12 reads per 24 MFMAs: 0.5.  Per accumulator the taps arrive as (dx, dy ascending) -- tap_of()'s order.  The chunk barrier sits
    // after group 0 of the last phase; every read of the next chunk's buffer comes behind it.
    static_assert(TN + 2 == 3 * TM, "one weight fragment and one activation fragment per group");
    bf16x8 afr[3][3][TM], bfr[TN + 2];
    auto load_a = [&](const uint4* base, int dx, int idx, bf16x8 (&ao)[3][TM]) __attribute__((always_inline)) {
        const int dy = idx / TM, i = idx % TM;
        ao[dy][i] = __builtin_bit_cast(bf16x8, base[a_off + (dy * 3 + dx) * 2 * BM + i * 32]);
    };
    auto load_b = [&](const uint4* base, int r, int dx) __attribute__((always_inline)) { bfr[r] = __builtin_bit_cast(bf16x8, base[b_off + r * PXW + dx]); };
    __syncthreads();
    if constexpr (DIAG) { t_first = __builtin_amdgcn_s_memrealtime(); c_first = __builtin_amdgcn_s_memtime(); }
#pragma unroll
    for (int r = 0; r < TN + 2; ++r) { load_a(smem, 0, r, afr[0]); load_b(smem, r, 0); }
    for (int ch = 0; ch < a.nch; ++ch) {
        const int cur = ch & 1;
        const bool more = ch + 1 < a.nch;
        const uint4* base = smem + cur * BUF_Q;
        const uint4* next = smem + (cur ^ 1) * BUF_Q;
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
#pragma unroll
            for (int r = 0; r < TN + 2; ++r) {
#pragma unroll
                for (int dy = 0; dy < 3; ++dy) {
                    const int j = r - dy;
                    if (j < 0 || j >= TN) continue;
#pragma unroll
                    for (int i = 0; i < TM; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr[dx][dy][i], bfr[r], acc[i][j], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                if (dx < 2) {
                    load_b(base, r, dx + 1);
                    load_a(base, dx + 1, r, afr[dx + 1]);
                    if (more && !(DIAG && a.diag_nodma) && dx * (TN + 2) + r < NPIECE) dma_piece(dx * (TN + 2) + r, ch + 1, cur ^ 1);
                } else if (more) {
                    if (r == 0) __syncthreads();                 // chunk ch + 1 has landed; every wave has issued its last reads of chunk ch
                    else { load_b(next, r - 1, 0); load_a(next, 0, r - 1, afr[0]); }
                    if (r == TN + 1) { load_b(next, r, 0); load_a(next, 0, r, afr[0]); }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
  } else {
    if constexpr (UNPOOL) {
        unpool_prefetch(0); unpool_landed();
#pragma unroll
        for (int u = 0; u < I_PER_WAVE; ++u) unpool_quad(u, 0);
    }
    __syncthreads();
    if constexpr (DIAG) { t_first = __builtin_amdgcn_s_memrealtime(); c_first = __builtin_amdgcn_s_memtime(); }
    fetch(smem, tap_of(0), av[0], bv[0]);
    for (int ch = 0; ch < a.nch; ++ch) {
        const int cur = ch & 1;
        const bool more = ch + 1 < a.nch;
        const uint4* base = smem + cur * BUF_Q;
        const uint4* next = smem + (cur ^ 1) * BUF_Q;
#pragma unroll
        for (int s2 = 0; s2 < NSTEP; ++s2) {
#pragma unroll
            for (int ij = 0; ij < TM * TN; ++ij) {
                const int i = ij / TN, j = ij % TN;
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[s2 % 3][i], bv[s2 % 3][j], acc[i][j], 0, 0, 0);
                if (ij == 0) {
                    __builtin_amdgcn_sched_barrier(0);
                    if (s2 + 1 < NSTEP) {
                        fetch(base, tap_of(s2 + 1), av[(s2 + 1) % 3], bv[(s2 + 1) % 3]);
                        if (more && !(DIAG && a.diag_nodma)) {
#pragma unroll
                            for (int pp = 0; pp < PPS; ++pp)
                                if (s2 * PPS + pp < NPIECE) dma_piece(s2 * PPS + pp, ch + 1, cur ^ 1);
                            if (UNPOOL && s2 == NSTEP - 3) unpool_prefetch(ch + 1);
                        }
                    } else if (more) {
                        if constexpr (UNPOOL) {
                            unpool_landed();
#pragma unroll
                            for (int u = 0; u < I_PER_WAVE; ++u) unpool_quad(u, cur ^ 1);
                        }
                        __syncthreads();
                        fetch(next, tap_of(0), av[0], bv[0]);    // 9 % 3 == 0: the next chunk starts on set 0 again
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
  }

    if constexpr (DIAG) { t_loop = __builtin_amdgcn_s_memrealtime(); c_loop = __builtin_amdgcn_s_memtime(); }
    // ---- epilogue: fp32 blob (same as the fp32 kernel, optional) + bf16 channel-blocked copy (optional) + fused pool (optional)
    // Written for the memory system, not for brevity: every load the epilogue needs (the bf16 ReLU masks of the whole tile,
    // the bias) is issued up front and unconditionally (clamped addresses instead of branches: a per-element "load or zero"
    // makes hipcc branch around each load and wait for it alone), and the per-channel bounds tests of the stores vanish on
    // the (wave-uniform) fast path of a tile that lies inside M.
    const int gx = x0 + l31;
    const bool colv = gx < a.W;
    const bool fuse_style = DG && a.s_nch > 0;                   // the ReLU mask is then applied in registers, before the style chunks
    constexpr bool has_bits = MB;                                // the ReLU mask as a sign map: 16 elements per 2-byte load (instead of mask16)
    const bool has_bias = !DG && a.bias != nullptr, relu = !DG && a.relu, has_mask = DG && !MB && a.mask_src != nullptr, has_inj = DG && a.inject != nullptr,
               has_mask16 = DG && !MB && a.mask16 != nullptr && !fuse_style;
    const bool pooling = !DG && TN % 2 == 0 && (a.pool16 || a.pool32 || a.amap);
    const bool bits_out = !DG && a.bits_out != nullptr;
    const bool full_m = m0 + BM <= a.M;                          // uniform: no channel of this tile is padding
    unsigned pixj[TN];
    bool livej[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int gy = y0 + wave_n * TN + j;
        livej[j] = colv && gy < a.H;
        pixj[j] = livej[j] ? (unsigned)gy * a.W + gx : 0u;
    }
    // masks of the whole tile up front when that fits the register file, otherwise per 32-channel group
    constexpr bool MK_ALL = TM * TN <= 4;
    constexpr int MK_I = MK_ALL ? TM : 1;
    uint2 mk16[MK_I][2][TN][2];
    auto load_masks = [&](int i, uint2 (&mk)[2][TN][2]) {
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int g = 0; g < 2; ++g) {
                    const int mg0 = m0 + wave_m * (TM * 32) + i * 32 + 4 * khalf + 16 * h + 8 * g;
                    const int mg = mg0 < a.M ? mg0 : 0;
                    mk[h][j][g] = *reinterpret_cast<const uint2*>(a.mask16 + ((size_t)(mg >> 3) * plane + pixj[j]) * 8 + (mg & 7));
                }
    };
    // sign maps go through buffer resources: one offset register per access instead of a 64-bit address (this epilogue has none to spare);
    // 4 bytes per (32-channel group, pixel) -- below 2^32 with the output's own byte count
    const unsigned bits_bytes = (unsigned)(a.M >> 5) * plane * 4u;
    const __amdgpu_buffer_rsrc_t rs_mb = __builtin_amdgcn_make_buffer_rsrc((void*)a.mask_bits, 0, has_bits ? bits_bytes : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_ob = __builtin_amdgcn_make_buffer_rsrc((void*)a.bits_out, 0, bits_out ? bits_bytes : 0u, 0x00020000);
    // (the group offset travels in the SCALAR offset, which the hardware's range check leaves out: a padding group -- M % 64 == 32 --
    // must be kept off the buffer explicitly, or its accesses land `plane * 4` bytes past the map; wave-uniform test)
    const unsigned n_groups = (unsigned)a.M >> 5;
    auto load_bits = [&](int i, unsigned (&mb)[TN]) {            // word j: bit 8 h + e keeps acc[i][j][8 h + e]  (a padding group: 0)
        const unsigned blk = (unsigned)(m0 + wave_m * (TM * 32) + i * 32) >> 5;
#pragma unroll
        for (int j = 0; j < TN; ++j) mb[j] = blk < n_groups ? __builtin_amdgcn_raw_buffer_load_b16(rs_mb, pixj[j] * 4u + 2u * khalf, blk * plane * 4u, 0) : 0u;
    };
    if (fuse_style) {
        // ---- fused style gradient (worker.py:262-269 behind the ranged backward of :100-106): this launch's output is the diff of a
        // style layer's blob, diff = mask(dgrad) + sw / norm * c2 * (D @ F).  The conv sum is masked in registers, then C / 16 more
        // chunks accumulate D' @ F on top: A = D' = D scaled, as hi + lo bf16 terms (slab [hl][k half][BM] per 16 channels, packed by
        // style16_pack_d_scaled_k), B = the blob's own bf16 copy, staged as the conv's activation tile (its centre tap is the pixel).
        // Saves the separate style-gradient kernel, its fp32 output and this epilogue's read of it.
        // Round 5: a style chunk is 2 MFMAs per accumulator behind a full staging step, so the chunks ran at one DMA latency each (conv2_2's
        // data gradient at 2048^2: 14 of a workgroup's 44 us).  The double-buffered builds now stage a style chunk COMPACTLY -- the D' slab +
        // the blob's 16 channels of exactly this tile's pixels, [k half][row][32 px], no halo: 20 KiB instead of 38 -- into a ring of three
        // slots over the main loop's two buffers, two chunks in flight behind the one being multiplied (waits counted by hand), the first two
        // issued before the mask is applied to the accumulators.
        constexpr int SW_QUADS = 4 * BM;                         // [hl][k half][BM]
        constexpr int SW_INSTR = SW_QUADS / 64;
        static_assert(SW_QUADS <= W_QUADS, "the style slab fits the weight region");
        constexpr int S_ACT = 2 * ROWS * 32, S_SLOT = SW_QUADS + S_ACT, SA_INSTR = S_ACT / 64;
        constexpr int S_PW = SW_INSTR / 4 + SA_INSTR / 4;        // LDS-DMA instructions per wave and chunk
        static_assert(SB || (3 * S_SLOT <= 2 * BUF_Q && SW_INSTR % 4 == 0 && SA_INSTR % 4 == 0), "three compact style chunks fit the two staging buffers");
        const __amdgpu_buffer_rsrc_t rs_d = __builtin_amdgcn_make_buffer_rsrc((void*)a.s_wpack, 0, a.s_w_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_s = __builtin_amdgcn_make_buffer_rsrc((void*)a.s_in16, 0, a.s_in_bytes, 0x00020000);
        int ln = lane;
        if constexpr (UNPOOL) asm volatile("" : "+v"(ln));      // computed HERE: hoisted above the main loop they would live through it in scratch
        unsigned soff2[SB ? 1 : SA_INSTR / 4];
        if constexpr (!SB) {
#pragma unroll
            for (int t = 0; t < SA_INSTR / 4; ++t) {
                const int q = (wave + 4 * t) * 64 + ln;          // quad of the compact tile: [k half][row][px]
                const int h = q / (ROWS * 32), rem = q - h * (ROWS * 32);
                const int gy = y0 + (rem >> 5), gx2 = x0 + (rem & 31);
                soff2[t] = (gy < a.H && gx2 < a.W) ? ((unsigned)h * plane + (unsigned)gy * a.W + gx2) * 16u : kOOB16;
            }
        }
        auto s_dma3 = [&](int ch, int slot) __attribute__((always_inline)) {
            uint4* dst = smem + slot * S_SLOT;
#pragma unroll
            for (int t = 0; t < SW_INSTR / 4; ++t) {
                const int i = wave + 4 * t;
                const int q = i * 64 + ln, r = q / BM, m = q - r * BM;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_d, (lptr_t)(dst + i * 64), 16, ((unsigned)(ch * 4 + r) * a.MPad + m0 + m) * 16u, 0, 0, 0);
            }
#pragma unroll
            for (int t = 0; t < SA_INSTR / 4; ++t) {
                const unsigned coff = (unsigned)ch * 2u * plane * 16u;
                const unsigned vo = soff2[t] == kOOB16 ? kOOB16 : soff2[t] + coff;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_s, (lptr_t)(dst + SW_QUADS + (wave + 4 * t) * 64), 16, vo, 0, 0, 0);
            }
        };
        if constexpr (!SB) {
            __syncthreads();                                     // every wave has left the main loop's buffers
            s_dma3(0, 0);
            if (a.s_nch > 1) s_dma3(1, 1);
        }
        if (has_bits) {
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                unsigned mb[TN];
                load_bits(i, mb);
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[i][j][e] = keep_if_bit(acc[i][j][e], mb[j], e);
            }
        } else if (!MB && a.mask16) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int h = 0; h < 2; ++h) {                    // (one half of the 32-channel group at a time: 16 registers of masks, not 32)
                    uint2 mk[TN][2];
#pragma unroll
                    for (int j = 0; j < TN; ++j)
#pragma unroll
                        for (int g = 0; g < 2; ++g) {
                            const int mg0 = m0 + wave_m * (TM * 32) + i * 32 + 4 * khalf + 16 * h + 8 * g;
                            const int mg = mg0 < a.M ? mg0 : 0;
                            mk[j][g] = *reinterpret_cast<const uint2*>(a.mask16 + ((size_t)(mg >> 3) * plane + pixj[j]) * 8 + (mg & 7));
                        }
#pragma unroll
                    for (int j = 0; j < TN; ++j)
#pragma unroll
                        for (int g = 0; g < 2; ++g) {
                            const uint2 m2 = mk[j][g];
                            acc[i][j][8 * h + 4 * g + 0] = (m2.x & 0xffffu) ? acc[i][j][8 * h + 4 * g + 0] : 0.0f;
                            acc[i][j][8 * h + 4 * g + 1] = (m2.x >> 16) ? acc[i][j][8 * h + 4 * g + 1] : 0.0f;
                            acc[i][j][8 * h + 4 * g + 2] = (m2.y & 0xffffu) ? acc[i][j][8 * h + 4 * g + 2] : 0.0f;
                            acc[i][j][8 * h + 4 * g + 3] = (m2.y >> 16) ? acc[i][j][8 * h + 4 * g + 3] : 0.0f;
                        }
                }
        }
      if constexpr (SB) {                                      // single staging buffer: one chunk at a time, staged as the conv's own tile
        unsigned soff[I_PER_WAVE];
#pragma unroll
        for (int u = 0; u < I_PER_WAVE; ++u) soff[u] = ioff[u];
        auto s_dma = [&](int ch) {
            uint4* dst = smem;
#pragma unroll
            for (int t = 0; t < (SW_INSTR + 3) / 4; ++t) {
                const int i = wave + 4 * t;
                if (SW_INSTR % 4 == 0 || i < SW_INSTR) {
                    const int q = i * 64 + ln, r = q / BM, m = q - r * BM;
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_d, (lptr_t)(dst + i * 64), 16, ((unsigned)(ch * 4 + r) * a.MPad + m0 + m) * 16u, 0, 0, 0);
                }
            }
#pragma unroll
            for (int u = 0; u < I_PER_WAVE; ++u) {
                const int j = wave + 4 * u;
                if (I_INSTR % 4 == 0 || j < I_INSTR) {
                    const unsigned coff = (unsigned)ch * 2u * plane * 16u;
                    const unsigned vo = soff[u] == kOOB16 ? kOOB16 : soff[u] + coff;
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_s, (lptr_t)(dst + W_QUADS + j * 64), 16, vo, 0, 0, 0);
                }
            }
        };
        __syncthreads();                                         // every wave has left the main loop's buffer
        s_dma(0);
        for (int ch = 0; ch < a.s_nch; ++ch) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();                                     // chunk ch has landed
            bf16x8 ahi[TM], alo[TM], bq[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                ahi[i] = __builtin_bit_cast(bf16x8, smem[khalf * BM + wave_m * (TM * 32) + i * 32 + l31]);
                alo[i] = __builtin_bit_cast(bf16x8, smem[(2 + khalf) * BM + wave_m * (TM * 32) + i * 32 + l31]);
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) bq[j] = __builtin_bit_cast(bf16x8, smem[b_off + (j + 1) * PXW + 1]);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ahi[i], bq[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(alo[i], bq[j], acc[i][j], 0, 0, 0);
                }
            if (ch + 1 < a.s_nch) {
                __syncthreads();                                 // the operands of chunk ch are in registers everywhere
                s_dma(ch + 1);
            }
        }
      } else {
        int slot = 0;
        for (int ch = 0; ch < a.s_nch; ++ch) {
            // chunk ch has landed when all but the newer chunk's loads are back (this wave's; the barrier makes it every wave's)
            if (ch + 1 < a.s_nch) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(S_PW) : "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                        // ... and every wave has read chunk ch - 1: its slot takes chunk ch + 2
            const int slot2 = slot == 0 ? 2 : slot - 1;
            if (ch + 2 < a.s_nch) s_dma3(ch + 2, slot2);
            const uint4* base = smem + slot * S_SLOT;
            bf16x8 ahi[TM], alo[TM], bq[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                ahi[i] = __builtin_bit_cast(bf16x8, base[khalf * BM + wave_m * (TM * 32) + i * 32 + l31]);
                alo[i] = __builtin_bit_cast(bf16x8, base[(2 + khalf) * BM + wave_m * (TM * 32) + i * 32 + l31]);
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) bq[j] = __builtin_bit_cast(bf16x8, base[SW_QUADS + khalf * (ROWS * 32) + (wave_n * TN + j) * 32 + l31]);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ahi[i], bq[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(alo[i], bq[j], acc[i][j], 0, 0, 0);
                }
            slot = slot == 2 ? 0 : slot + 1;
        }
      }
    }
    if (MK_ALL && has_mask16) {
#pragma unroll
        for (int i = 0; i < TM; ++i) load_masks(i, mk16[i]);
    }
    auto tile_out = [&](auto full_t) {
        constexpr bool FULL = decltype(full_t)::value;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            if (!MK_ALL && has_mask16) load_masks(i, mk16[0]);
            unsigned mbits[TN];
            if (has_bits && !fuse_style) load_bits(i, mbits);
            unsigned obits[TN];                                  // bits_out: word (i, j) of this lane, filled by both h
#pragma unroll
            for (int j = 0; j < TN; ++j) obits[j] = 0u;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int mbase = m0 + wave_m * (TM * 32) + i * 32 + 4 * khalf + 16 * h;
                float v[TN][8];
                float bs[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) bs[e] = 0.0f;
                if (has_bias) {                                   // the bias array is MPad long: no bounds test
#pragma unroll
                    for (int e = 0; e < 8; ++e) bs[e] = a.bias[mbase + (e & 3) + 8 * (e >> 2)];
                }
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const bool live = livej[j];
                    const unsigned pix = pixj[j];
                    unsigned off[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const int m = mbase + (e & 3) + 8 * (e >> 2);
                        off[e] = (unsigned)((FULL || m < a.M) ? m : a.M - 1) * plane + pix;
                        v[j][e] = acc[i][j][8 * h + e] + bs[e];
                    }
                    if (relu) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[j][e] = v[j][e] > 0.0f ? v[j][e] : 0.0f;
                    }
                    if (has_bits && !fuse_style) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[j][e] = keep_if_bit(v[j][e], mbits[j], 8 * h + e);
                    } else if (has_mask16) {
                        // the blob below is post-ReLU (>= 0): its bf16 copy is non-zero exactly where it is positive
#pragma unroll
                        for (int g = 0; g < 2; ++g) {
                            const uint2 mk = mk16[MK_ALL ? i : 0][h][j][g];
                            v[j][4 * g + 0] = (mk.x & 0xffffu) ? v[j][4 * g + 0] : 0.0f;
                            v[j][4 * g + 1] = (mk.x >> 16) ? v[j][4 * g + 1] : 0.0f;
                            v[j][4 * g + 2] = (mk.y & 0xffffu) ? v[j][4 * g + 2] : 0.0f;
                            v[j][4 * g + 3] = (mk.y >> 16) ? v[j][4 * g + 3] : 0.0f;
                        }
                    } else if (has_mask) {
                        float mk[8];
#pragma unroll
                        for (int e = 0; e < 8; ++e) mk[e] = a.mask_src[off[e]];
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[j][e] = mk[e] > 0.0f ? v[j][e] : 0.0f;
                    }
                    if (has_inj) {
                        float ij[8];
#pragma unroll
                        for (int e = 0; e < 8; ++e) ij[e] = a.inject[off[e]];
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[j][e] += ij[e];
                    }
                    if (a.out && live) {
#pragma unroll
                        for (int e = 0; e < 8; ++e)
                            if (FULL || mbase + (e & 3) + 8 * (e >> 2) < a.M) a.out[off[e]] = v[j][e];
                    }
                    if (a.out16 && live) {
                        // rows mbase..+3 and mbase+8..+11: this lane holds HALF (4 channels) of two 8-channel quads of its pixel, lane ^ 32
                        // the other halves (same pixel, so both are live together).  v_permlane32_swap exchanges them: lanes 0-31 own the
                        // first quad, lanes 32-63 the second -- one 16-byte store per lane instead of two interleaved 8-byte ones.
                        bf16x4 pk0, pk1;
#pragma unroll
                        for (int e = 0; e < 4; ++e) { pk0[e] = (__bf16)v[j][e]; pk1[e] = (__bf16)v[j][4 + e]; }
                        const uint2 u0 = __builtin_bit_cast(uint2, pk0), u1 = __builtin_bit_cast(uint2, pk1);
                        if (bits_out) obits[j] |= nonzero_halves16(u0, u1) << (8 * h);
                        const auto sx = __builtin_amdgcn_permlane32_swap(u0.x, u1.x, false, false);
                        const auto sy = __builtin_amdgcn_permlane32_swap(u0.y, u1.y, false, false);
                        const int mq = mbase - 4 * khalf + 8 * khalf;                  // first channel of this lane's quad
                        if (FULL || mq < a.M)           // M is a multiple of 8 on this path (checked at launch)
                            *reinterpret_cast<uint4*>(a.out16 + ((size_t)(mq >> 3) * plane + pix) * 8) = make_uint4(sx[0], sy[0], sx[1], sy[1]);
                    }
                }
                if constexpr (TN % 2 == 0) {
                    if (pooling) {
                        // Caffe MAX 2x2/2, ceil mode: each pair of this wave's rows is one row of pooling windows (row origins are
                        // even), lane pairs (even gx, gx + 1) are the two columns.  First maximum of a row-major scan, strictly
                        // greater (oracle.caffe_net.maxpool_forward); windows are clipped at the right / bottom edge.
#pragma unroll
                        for (int jp = 0; jp < TN; jp += 2) {
                            const int gy0 = y0 + wave_n * TN + jp;
                            const bool row1 = gy0 + 1 < a.H, col1 = gx + 1 < a.W;
                            const bool writer = !(l31 & 1) && colv && gy0 < a.H;
                            const size_t pplane = (size_t)a.pool_h * a.pool_w;
                            const size_t ppix = writer ? (size_t)(gy0 >> 1) * a.pool_w + (gx >> 1) : 0;
                            float best[8];
                            unsigned code[8];
#pragma unroll
                            for (int e = 0; e < 8; ++e) {
                                const float p0 = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v[jp][e]), 0xB1, 0xf, 0xf, true));
                                const float p1 = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v[jp + 1][e]), 0xB1, 0xf, 0xf, true));
                                float bb = v[jp][e];
                                unsigned sl = 0;
                                if (col1 && p0 > bb) { bb = p0; sl = 1; }
                                if (row1 && v[jp + 1][e] > bb) { bb = v[jp + 1][e]; sl = 2; }
                                if (row1 && col1 && p1 > bb) { bb = p1; sl = 3; }
                                best[e] = bb;
                                code[e] = sl | (bb > 0.0f ? 4u : 0u);
                            }
                            if (writer) {
#pragma unroll
                                for (int g = 0; g < 2; ++g) {
                                    const int mg = mbase + 8 * g;
                                    if (!FULL && mg >= a.M) continue;
                                    const size_t q8 = ((size_t)(mg >> 3) * pplane + ppix) * 8 + (mg & 7);
                                    if (a.pool16) {
                                        bf16x4 pk;
#pragma unroll
                                        for (int e = 0; e < 4; ++e) pk[e] = (__bf16)best[4 * g + e];
                                        *reinterpret_cast<bf16x4*>(a.pool16 + q8) = pk;
                                    }
                                    if (a.amap)
                                        *reinterpret_cast<unsigned*>(a.amap + q8) = code[4 * g] | (code[4 * g + 1] << 8) | (code[4 * g + 2] << 16) | (code[4 * g + 3] << 24);
                                    if (a.pool32) {
#pragma unroll
                                        for (int e = 0; e < 4; ++e)
                                            if (FULL || mg + e < a.M) a.pool32[(size_t)(mg + e) * pplane + ppix] = best[4 * g + e];
                                    }
                                }
                            }
                        }
                    }
                }
            }
            const unsigned blk_o = (unsigned)(m0 + wave_m * (TM * 32) + i * 32) >> 5;
            if (bits_out && blk_o < n_groups) {                   // (M % 32 == 0 on this path: the 32-channel group is whole, or padding and skipped)
                const unsigned blk = blk_o;
#pragma unroll
                for (int j = 0; j < TN; ++j) {                   // lane ^ 32 holds the other 16-bit word of the same pixel: one dword store per pixel
                    const auto sw = __builtin_amdgcn_permlane32_swap(obits[j], obits[j], false, false);
                    if (livej[j] && !khalf) __builtin_amdgcn_raw_buffer_store_b32(obits[j] | (sw[1] << 16), rs_ob, pixj[j] * 4u, blk * plane * 4u, 0);
                }
            }
        }
    };
    if constexpr (EPI != 0) {
        // ---- one epilogue per launch kind (round 5).  The lean flow launches three kinds over and over, none of which touches fp32:
        //   EPI 1  forward build:        (bias, ReLU) -> bf16 copy (+ sign map)         -- also the data gradients with no option at all
        //   EPI 2  forward build:        (bias, ReLU) -> 2x2 max-pool -> pooled bf16 copy + arg-max map (whole windows: H, W even)
        //   EPI 3  data-gradient build:  sign-map ReLU mask (here, or before the fused style chunks) -> bf16 copy
        // and two more once per step, at the blob that carries a content / deep-dream term:
        //   EPI 4  = 1 + the fp32 blob (the loss reads fp32)        EPI 5  = 3 + the injected fp32 diff of that term, added after the mask
        // every tile whole in M (M % BM == 0, checked at launch).  Same arithmetic on every value as the general epilogue below (the
        // tests compare bit for bit, ST2_CONV16_EPI=0 selects the general one), but every global access is a buffer access -- a
        // wave-uniform scalar offset per 16-channel row group, one vector offset per pixel row, out-of-range for what must not be
        // written -- so the per-element 64-bit addresses, bounds tests and run-time option branches are gone: ~40 instructions per
        // (16 channels x 32 pixels) instead of ~300.  ReLU is an integer maximum of the bit pattern with 0 (off: with INT_MIN): v > 0 ? v : +0
        // for every v that is not a NaN.
        constexpr bool E_F = EPI == 1 || EPI == 4, E_D = EPI == 3 || EPI == 5;
        static_assert(!DIAG && E_D == (DG && MB) && (EPI != 2 || TN % 2 == 0), "launch kinds");
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
        const unsigned mrow0 = (unsigned)(m0 + wave_m * (TM * 32));
        const int floor_i = (!DG && a.relu) ? 0 : (int)0x80000000;
        const __amdgpu_buffer_rsrc_t rs_bias = __builtin_amdgcn_make_buffer_rsrc((void*)a.bias, 0, (!DG && a.bias) ? (unsigned)a.MPad * 4u : 0u, 0x00020000);
        auto finish8 = [&](int i, int j, int h, const float (&bs)[8], float (&v)[8]) __attribute__((always_inline)) {      // bias + ReLU of 8 accumulators
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float s = acc[i][j][8 * h + e] + bs[e];
                const int b = __builtin_bit_cast(int, s);
                v[e] = __builtin_bit_cast(float, __builtin_elementwise_max(b, floor_i));
            }
        };
        auto load_bias = [&](int i, int h, float (&bs)[8]) __attribute__((always_inline)) {      // channels mbase .. +3 and mbase + 8 .. + 11 (zeros without a bias)
            const unsigned so = (mrow0 + i * 32 + 16 * h) * 4u;
            // (through HIP's uint4: the builtin's result assigned to an ext_vector_type is narrowed to ONE dword by this compiler)
            const uint4 b0 = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rs_bias, 16u * khalf, so, 0));
            const uint4 b1 = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rs_bias, 16u * khalf + 32u, so, 0));
            bs[0] = __builtin_bit_cast(float, b0.x); bs[1] = __builtin_bit_cast(float, b0.y); bs[2] = __builtin_bit_cast(float, b0.z); bs[3] = __builtin_bit_cast(float, b0.w);
            bs[4] = __builtin_bit_cast(float, b1.x); bs[5] = __builtin_bit_cast(float, b1.y); bs[6] = __builtin_bit_cast(float, b1.z); bs[7] = __builtin_bit_cast(float, b1.w);
        };
        auto pack8 = [&](const float (&v)[8], uint2& u0, uint2& u1) __attribute__((always_inline)) {
            bf16x4 pk0, pk1;
#pragma unroll
            for (int e = 0; e < 4; ++e) { pk0[e] = (__bf16)v[e]; pk1[e] = (__bf16)v[4 + e]; }
            u0 = __builtin_bit_cast(uint2, pk0); u1 = __builtin_bit_cast(uint2, pk1);
        };
        if constexpr (E_F || E_D) {
            const __amdgpu_buffer_rsrc_t rs_o = __builtin_amdgcn_make_buffer_rsrc((void*)a.out16, 0, (unsigned)a.M * plane * 2u, 0x00020000);
            unsigned vo[TN];                                     // this lane's quad (lanes 32 .. 63: the second quad of the 16 channels) of pixel j
#pragma unroll
            for (int j = 0; j < TN; ++j) vo[j] = livej[j] ? ((unsigned)khalf * plane + pixj[j]) * 16u : kOOBStore;
            // fp32 side of kinds 4 / 5 (NCHW): element e of a lane is channel mbase + (e & 3) + 8 (e >> 2), mbase = row group + 4 k half + 16 h --
            // the lane's part (pixel, k half) in the vector offset, the rest in the scalar offset (4-byte accesses: no store-data hazard)
            const __amdgpu_buffer_rsrc_t rs_f32 = __builtin_amdgcn_make_buffer_rsrc((void*)(EPI == 4 ? (const float*)a.out : a.inject), 0, (unsigned)a.M * plane * 4u, 0x00020000);
            unsigned vf[(EPI == 4 || EPI == 5) ? TN : 1];
            if constexpr (EPI == 4 || EPI == 5) {
#pragma unroll
                for (int j = 0; j < TN; ++j) vf[j] = livej[j] ? (4u * khalf * plane + pixj[j]) * 4u : kOOB16;
            }
            // eight bf16 values in four words -> one byte, bit e = value e is non-zero (nonzero_halves16 without a condition register)
            auto nz8 = [&](uint2 u0, uint2 u1) __attribute__((always_inline)) -> unsigned {
                // min(half, 1) per 16-bit half.  The 1s pass through an EMPTY asm so that the compiler cannot turn the minimum back into two
                // compares + selects per word (it does when it sees the constant); the instruction itself stays its own to schedule
                typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
                unsigned ones = 0x00010001u;
                asm("" : "+v"(ones));
                auto one = [&](unsigned w) __attribute__((always_inline)) {
                    return __builtin_bit_cast(unsigned, __builtin_elementwise_min(__builtin_bit_cast(u16x2, w), __builtin_bit_cast(u16x2, ones)));
                };
                const unsigned x = one(u0.x) | (one(u0.y) << 2), y = one(u1.x) | (one(u1.y) << 2);
                const unsigned z = x | (y << 4);                 // low halves at bits 0 2 4 6, high halves at 16 18 20 22
                return (z & 0x55u) | ((z >> 15) & 0xaau);
            };
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                unsigned mbits[TN];
                if constexpr (E_D) { if (!fuse_style) load_bits(i, mbits); }
                unsigned obits[TN];
#pragma unroll
                for (int j = 0; j < TN; ++j) obits[j] = 0u;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    float bs[8];
                    if constexpr (E_F) load_bias(i, h, bs);
                    const unsigned so = ((mrow0 + i * 32 + 16 * h) >> 3) * plane * 16u;
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        float v[8];
                        if constexpr (E_F) finish8(i, j, h, bs, v);
                        else {
                            float ij[8];
                            if constexpr (EPI == 5) {
#pragma unroll
                                for (int e = 0; e < 8; ++e)
                                    ij[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_f32, vf[j], (mrow0 + i * 32 + 16 * h + (e & 3) + 8 * (e >> 2)) * plane * 4u, 0));
                            }
#pragma unroll
                            for (int e = 0; e < 8; ++e) v[e] = acc[i][j][8 * h + e];
                            if (!fuse_style) {
#pragma unroll
                                for (int e = 0; e < 8; ++e) v[e] = keep_if_bit(v[e], mbits[j], 8 * h + e);
                            }
                            if constexpr (EPI == 5) {
#pragma unroll
                                for (int e = 0; e < 8; ++e) v[e] += ij[e];
                            }
                        }
                        if constexpr (EPI == 4) {
#pragma unroll
                            for (int e = 0; e < 8; ++e)
                                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v[e]), rs_f32, vf[j], (mrow0 + i * 32 + 16 * h + (e & 3) + 8 * (e >> 2)) * plane * 4u, 0);
                        }
                        uint2 u0, u1;
                        pack8(v, u0, u1);
                        if (E_F && bits_out) obits[j] |= nz8(u0, u1) << (8 * h);
                        const auto sx = __builtin_amdgcn_permlane32_swap(u0.x, u1.x, false, false);
                        const auto sy = __builtin_amdgcn_permlane32_swap(u0.y, u1.y, false, false);
                        __builtin_amdgcn_raw_buffer_store_b128((u32x4){sx[0], sy[0], sx[1], sy[1]}, rs_o, vo[j] + so, 0, 0);      // (no SGPR offset: see kOOBStore)
                    }
                }
                if (E_F && bits_out) {                          // (M % BM == 0: no padding group on this path)
                    const unsigned blk = (mrow0 + i * 32) >> 5;
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        const auto sw = __builtin_amdgcn_permlane32_swap(obits[j], obits[j], false, false);
                        __builtin_amdgcn_raw_buffer_store_b32(obits[j] | (sw[1] << 16), rs_ob, (livej[j] && !khalf) ? pixj[j] * 4u : kOOB16, blk * plane * 4u, 0);
                    }
                }
            }
        } else {
            // Caffe MAX 2x2/2 on whole windows: a pair of this wave's rows is one row of windows, lane pairs (even gx, gx + 1) its columns;
            // first maximum of the row-major scan, strictly greater (oracle.caffe_net.maxpool_forward) -- the general epilogue's scan
            const unsigned pplane = (unsigned)a.pool_h * a.pool_w;
            const __amdgpu_buffer_rsrc_t rs_p = __builtin_amdgcn_make_buffer_rsrc((void*)a.pool16, 0, (unsigned)a.M * pplane * 2u, 0x00020000);
            const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc((void*)a.amap, 0, (unsigned)a.M * pplane, 0x00020000);
            unsigned vq[TN / 2];                                 // quad index of this lane's window of row pair jp (writers: even columns inside the image)
#pragma unroll
            for (int jp = 0; jp < TN; jp += 2) {
                const int gy0 = y0 + wave_n * TN + jp;
                const bool writer = !(l31 & 1) && colv && gy0 < a.H;
                vq[jp / 2] = writer ? (unsigned)khalf * pplane + (unsigned)(gy0 >> 1) * a.pool_w + (gx >> 1) : (kOOBStore >> 4);
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    float bs[8];
                    load_bias(i, h, bs);
                    const unsigned sq = ((mrow0 + i * 32 + 16 * h) >> 3) * pplane;
                    float v[TN][8];
#pragma unroll
                    for (int j = 0; j < TN; ++j) finish8(i, j, h, bs, v[j]);
#pragma unroll
                    for (int jp = 0; jp < TN; jp += 2) {
                        float best[8];
                        unsigned code[8];
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const float p0 = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v[jp][e]), 0xB1, 0xf, 0xf, true));
                            const float p1 = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v[jp + 1][e]), 0xB1, 0xf, 0xf, true));
                            float bb = v[jp][e];
                            unsigned sl = 0;
                            if (p0 > bb) { bb = p0; sl = 1; }
                            if (v[jp + 1][e] > bb) { bb = v[jp + 1][e]; sl = 2; }
                            if (p1 > bb) { bb = p1; sl = 3; }
                            best[e] = bb;
                            code[e] = sl | (bb > 0.0f ? 4u : 0u);
                        }
                        uint2 u0, u1;
                        pack8(best, u0, u1);
                        const unsigned c0 = code[0] | (code[1] << 8) | (code[2] << 16) | (code[3] << 24);
                        const unsigned c1 = code[4] | (code[5] << 8) | (code[6] << 16) | (code[7] << 24);
                        const auto sx = __builtin_amdgcn_permlane32_swap(u0.x, u1.x, false, false);
                        const auto sy = __builtin_amdgcn_permlane32_swap(u0.y, u1.y, false, false);
                        const auto sc = __builtin_amdgcn_permlane32_swap(c0, c1, false, false);
                        __builtin_amdgcn_raw_buffer_store_b128((u32x4){sx[0], sy[0], sx[1], sy[1]}, rs_p, (vq[jp / 2] + sq) * 16u, 0, 0);      // (no SGPR offset: see kOOBStore)
                        __builtin_amdgcn_raw_buffer_store_b64((u32x2){sc[0], sc[1]}, rs_a, vq[jp / 2] * 8u, sq * 8u, 0);
                    }
                }
        }
    } else {
        if (full_m) tile_out(std::true_type{}); else tile_out(std::false_type{});
    }
    if constexpr (DIAG) {
        if (a.stamps && threadIdx.x == 0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the stores are out of the wave's queue
            unsigned long long* st = a.stamps + 6 * (size_t)blockIdx.x;
            st[0] = t_start; st[1] = t_first; st[2] = t_loop; st[3] = __builtin_amdgcn_s_memrealtime(); st[4] = c_first; st[5] = c_loop;
        }
    }
}

#define ST2_CONV16_KERNEL(NAME, BM, ROWS, WM, WN, WPE) \
    __global__ __launch_bounds__(NT, WPE) void NAME(const Conv16KArgs a) { conv16_body<BM, ROWS, WM, WN, false, false, false>(a); } \
    __global__ __launch_bounds__(NT, WPE) void NAME##_dg(const Conv16KArgs a) { conv16_body<BM, ROWS, WM, WN, false, false, true>(a); } \
    __global__ __launch_bounds__(NT, WPE) void NAME##_dgb(const Conv16KArgs a) { conv16_body<BM, ROWS, WM, WN, false, false, true, true>(a); }
ST2_CONV16_KERNEL(conv3x3_mfma_bf16_64x256, 64, 8, 1, 4, 2)
ST2_CONV16_KERNEL(conv3x3_mfma_bf16_128x128, 128, 4, 2, 2, 1)
ST2_CONV16_KERNEL(conv3x3_mfma_bf16_64x128, 64, 4, 1, 4, 3)
ST2_CONV16_KERNEL(conv3x3_mfma_bf16_64x512, 64, 16, 1, 4, 2)     // 4 rows per wave: twice the MFMA work per staged weight slab
// data-gradient launches directly below a max-pool: the pooled diff + the arg-max map, expanded in the staged tile (conv16_body, UNPOOL)
__global__ __launch_bounds__(NT, 2) void conv3x3_mfma_bf16_64x512_unpool(const Conv16KArgs a) { conv16_body<64, 16, 1, 4, false, true, true>(a); }
__global__ __launch_bounds__(NT, 2) void conv3x3_mfma_bf16_64x256_unpool(const Conv16KArgs a) { conv16_body<64, 8, 1, 4, false, true, true>(a); }
__global__ __launch_bounds__(NT, 2) void conv3x3_mfma_bf16_64x512_unpool_b(const Conv16KArgs a) { conv16_body<64, 16, 1, 4, false, true, true, true>(a); }
__global__ __launch_bounds__(NT, 2) void conv3x3_mfma_bf16_64x256_unpool_b(const Conv16KArgs a) { conv16_body<64, 8, 1, 4, false, true, true, true>(a); }
#ifndef SB_WPE
#define SB_WPE 4
#endif
// single staging buffer, four workgroups per CU (three for the legacy-mask data gradient: 150 registers): the short-K launches (conv16_body, SB)
__global__ __launch_bounds__(NT, SB_WPE) void conv3x3_mfma_bf16_64x256_sb(const Conv16KArgs a) { conv16_body<64, 8, 1, 4, true, false, false>(a); }
__global__ __launch_bounds__(NT, 3) void conv3x3_mfma_bf16_64x256_sb_dg(const Conv16KArgs a) { conv16_body<64, 8, 1, 4, true, false, true>(a); }
// DIAG builds (tools/probes only: Conv16Problem::stamps)
__global__ __launch_bounds__(NT, 2) void conv3x3_mfma_bf16_64x512_diag(const Conv16KArgs a) { conv16_body<64, 16, 1, 4, false, false, false, false, true>(a); }
__global__ __launch_bounds__(NT, 3) void conv3x3_mfma_bf16_64x256_sb_diag(const Conv16KArgs a) { conv16_body<64, 8, 1, 4, true, false, false, false, true>(a); }
__global__ __launch_bounds__(NT, SB_WPE) void conv3x3_mfma_bf16_64x256_sb_dgb(const Conv16KArgs a) { conv16_body<64, 8, 1, 4, true, false, true, true>(a); }

// one epilogue per launch kind (conv16_body, EPI): the tiles and kinds the lean 2048^2 flow launches
__global__ __launch_bounds__(NT, SB_WPE) void conv3x3_mfma_bf16_64x256_sb_f16(const Conv16KArgs a) { conv16_body<64, 8, 1, 4, true, false, false, false, false, 1>(a); }
__global__ __launch_bounds__(NT, SB_WPE) void conv3x3_mfma_bf16_64x256_sb_pool(const Conv16KArgs a) { conv16_body<64, 8, 1, 4, true, false, false, false, false, 2>(a); }
__global__ __launch_bounds__(NT, 2) void conv3x3_mfma_bf16_64x512_f16(const Conv16KArgs a) { conv16_body<64, 16, 1, 4, false, false, false, false, false, 1>(a); }
__global__ __launch_bounds__(NT, 2) void conv3x3_mfma_bf16_64x512_pool(const Conv16KArgs a) { conv16_body<64, 16, 1, 4, false, false, false, false, false, 2>(a); }
__global__ __launch_bounds__(NT, 2) void conv3x3_mfma_bf16_64x512_dgb16(const Conv16KArgs a) { conv16_body<64, 16, 1, 4, false, false, true, true, false, 3>(a); }
__global__ __launch_bounds__(NT, 2) void conv3x3_mfma_bf16_64x512_unpool_b16(const Conv16KArgs a) { conv16_body<64, 16, 1, 4, false, true, true, true, false, 3>(a); }
__global__ __launch_bounds__(NT, 2) void conv3x3_mfma_bf16_64x512_f16o(const Conv16KArgs a) { conv16_body<64, 16, 1, 4, false, false, false, false, false, 4>(a); }
__global__ __launch_bounds__(NT, 2) void conv3x3_mfma_bf16_64x512_dgb16i(const Conv16KArgs a) { conv16_body<64, 16, 1, 4, false, false, true, true, false, 5>(a); }
__global__ __launch_bounds__(NT, 2) void conv3x3_mfma_bf16_64x256_f16(const Conv16KArgs a) { conv16_body<64, 8, 1, 4, false, false, false, false, false, 1>(a); }
__global__ __launch_bounds__(NT, 2) void conv3x3_mfma_bf16_64x256_dgb16(const Conv16KArgs a) { conv16_body<64, 8, 1, 4, false, false, true, true, false, 3>(a); }

static int conv16_pick_cfg(const Conv16Problem& p)
{
    const char* env = getenv("ST2_CONV16_CFG");             // forces one tile configuration (tests of every configuration)
    const int forced = env && *env ? atoi(env) : -1;
    const long long tx = (p.W + 31) / 32;
    const char* big = getenv("ST2_CONV16_BIG_MIN");         // tuning knob: least number of 64x512 workgroups that selects that tile
    const long long big_min = big && *big ? atoll(big) : 512;
    int cfg;                                               // 0: 64x256px, 1: 128x128px, 2: 64x128px, 3: 64x512px
    if (forced >= 0) cfg = forced;
    else if (tx * ((p.H + 15) / 16) * (p.MPad / 64) >= big_min) cfg = 3;    // enough workgroups even at 512 pixels each
    else cfg = (tx * ((p.H + 7) / 8) * (p.MPad / 64) >= 512) ? 0 : 2;
    if (cfg == 1 && p.MPad % 128 != 0) cfg = 0;
    if (cfg > 3) cfg = 0;
    return cfg;
}

// may this launch pool its own output (Conv16Problem::pool16 / pool32 / amap)?  Needs a tile configuration whose waves
// hold two rows (0 and 1) and M % 8 == 0 (channel-blocked outputs).
bool conv16_can_pool(const Conv16Problem& p) { return conv16_pick_cfg(p) != 2 && p.M % 8 == 0; }

// may a data-gradient launch of this shape take the pooled diff + the pool's arg-max map (Conv16Problem::unpool_amap)?  The 64 x 512
// and 64 x 256 pixel tiles have the build; even H and W (every window whole).  ST2_CONV16_UNPOOL=0: keep maxpool_bwd_idx16_k (read per call).
bool conv16_can_unpool(const Conv16Problem& p)
{
    const char* e = getenv("ST2_CONV16_UNPOOL");
    if (e && *e == '0') return false;
    // ... and K <= 128 (ST2_CONV16_UNPOOL_MAXK): the expansion costs the launch about as much per pooled element as the separate kernel
    // did, which pays where that kernel's full-resolution output was the expense -- measured at 2048^2: conv1_2 -145 + 0 us,
    // conv2_2 -73 + 12, conv3_4 -39 + 32, conv4_4 -24 + 33
    const char* mk = getenv("ST2_CONV16_UNPOOL_MAXK");
    if (p.K > (mk && *mk ? atoi(mk) : 128)) return false;
    const int cfg = conv16_pick_cfg(p);
    return (cfg == 3 || cfg == 0) && p.H % 2 == 0 && p.W % 2 == 0 && p.K % 16 == 0;      // whole windows, whole 16-channel chunks
}

hipError_t launch_conv3x3_bf16(const Conv16Problem& p, hipStream_t s)
{
    if (p.MPad % kCoutQuantum != 0 || p.MPad < p.M) return hipErrorInvalidValue;
    if ((p.out16 || p.mask16) && p.M % 8 != 0) return hipErrorInvalidValue;
    const bool pools = p.pool16 || p.pool32 || p.amap;
    if (pools && !conv16_can_pool(p)) return hipErrorInvalidValue;
    if (!p.out && !p.out16 && !pools) return hipErrorInvalidValue;          // nothing to write
    const long long tx = (p.W + 31) / 32;
    int cfg = conv16_pick_cfg(p);
    // short reductions (K <= ST2_CONV16_SB_MAXK, default 64): the 64x256 tile with ONE staging buffer, four workgroups per CU (120
    // registers, 29 KiB of LDS each since the forward / data-gradient epilogues became builds of their own).  Measured in isolation
    // (tools/probes/conv16_shallow.sh, 2048^2 job): conv1_2 forward with its pool 386 -> 344 us, conv2_1 forward 186 -> 175, K = 128: 279 -> 268;
    // in the 2048^2 bf16 job: forward class 2.507 -> 2.466 ms with 64, the same with 128 but the data gradients +0.01 (profiles/r04_k_bf16_sb_ab.txt)
    const char* sbe = getenv("ST2_CONV16_SB_MAXK");          // read per launch: the tests compare both pipelines
    const int sb_maxk = sbe && *sbe ? atoi(sbe) : 64;
    const bool sb = p.K <= sb_maxk && (cfg == 3 || cfg == 0) && !p.unpool_amap;       // (the unpooling builds are double-buffered)
    if (sb) cfg = 0;
    const int BM = cfg == 1 ? 128 : 64, ROWS = cfg == 0 ? 8 : cfg == 3 ? 16 : 4;
    Conv16KArgs k{};
    k.in16 = p.in16; k.wpack = p.wpack16; k.bias = p.bias; k.out = p.out; k.out16 = p.out16;
    k.mask_src = p.mask_src; k.inject = p.inject; k.mask16 = p.mask16;
    k.pool16 = p.pool16; k.pool32 = p.pool32; k.amap = p.amap; k.pool_h = (p.H + 1) / 2; k.pool_w = (p.W + 1) / 2;
    k.K = p.K; k.M = p.M; k.MPad = p.MPad; k.H = p.H; k.W = p.W;
    k.nch = (p.K + 15) / 16;
    k.tiles_x = (int)tx; k.tiles_y = (p.H + ROWS - 1) / ROWS; k.n_mtiles = p.MPad / BM; k.relu = p.relu;
    if ((p.bits_out || p.mask_bits) && p.M % 32 != 0) return hipErrorInvalidValue;
    if (p.bits_out && (!p.out16 || (reinterpret_cast<uintptr_t>(p.bits_out) & 3) != 0)) return hipErrorInvalidValue;
    k.bits_out = p.bits_out; k.mask_bits = p.mask_bits;
    const bool unpool = p.unpool_amap != nullptr;
    if (unpool && (!conv16_can_unpool(p) || pools)) return hipErrorInvalidValue;
    k.up_amap = p.unpool_amap; k.up_h = p.H / 2; k.up_w = p.W / 2;
    const unsigned long long in_bytes = 16ull * ((p.K + 7) / 8) * (unpool ? (unsigned long long)k.up_h * k.up_w : (unsigned long long)p.H * p.W),
                             w_bytes = 2ull * conv16_pack_elems(p.K, p.M);
    const unsigned long long out_bytes = 4ull * p.M * p.H * p.W;
    if (in_bytes >= 0xfffffff0ull || w_bytes >= 0xfffffff0ull || out_bytes >= 0xfffffff0ull) return hipErrorInvalidValue;
    if (p.s_in16 || p.s_wpack16) {          // fused style term: a data-gradient launch whose output blob has M % 16 == 0 channels
        const unsigned long long s_in = 16ull * (p.M / 8) * p.H * p.W, s_w = 2ull * style_fuse_pack_elems(p.M, p.MPad);
        if (!p.s_in16 || !p.s_wpack16 || p.M % 16 != 0 || p.relu || p.bias || p.mask_src || pools || s_in >= 0xfffffff0ull || s_w >= 0xfffffff0ull) return hipErrorInvalidValue;
        if ((reinterpret_cast<uintptr_t>(p.s_in16) & 15) != 0 || (reinterpret_cast<uintptr_t>(p.s_wpack16) & 15) != 0) return hipErrorInvalidValue;
        k.s_in16 = p.s_in16; k.s_wpack = p.s_wpack16; k.s_nch = p.M / 16; k.s_in_bytes = (unsigned)s_in; k.s_w_bytes = (unsigned)s_w;
    }
    k.in_bytes = (unsigned)in_bytes; k.w_bytes = (unsigned)w_bytes;
    const long long nblk = (long long)k.tiles_x * k.tiles_y * k.n_mtiles;
    if (nblk <= 0 || nblk > 0x7fffffffLL) return hipErrorInvalidValue;
    // which epilogue: the data-gradient build iff the launch uses one of its options; both directions' options together have no build
    const bool dg = p.mask_src || p.mask16 || p.mask_bits || p.inject || p.s_in16 || unpool;
    if (dg && (p.bias || p.relu || pools || p.bits_out)) return hipErrorInvalidValue;
    const dim3 grid((unsigned)nblk), block(NT);
    if (p.stamps) {                                                              // measurement builds: forward launches of two tiles
        if (dg || !(cfg == 3 || (cfg == 0 && sb))) return hipErrorInvalidValue;
        k.stamps = p.stamps;
        { const char* nd = getenv("ST2_BENCH_NODMA"); k.diag_nodma = nd && *nd == '1'; }
        { const char* sg = getenv("ST2_BENCH_STAGGER"); k.diag_stagger = sg && *sg ? atoi(sg) : 0; }
        if (sb) conv3x3_mfma_bf16_64x256_sb_diag<<<grid, block, 0, s>>>(k); else conv3x3_mfma_bf16_64x512_diag<<<grid, block, 0, s>>>(k);
        return hipGetLastError();
    }
    const bool mb = p.mask_bits != nullptr;
    if (mb && (p.mask_src || p.mask16)) return hipErrorInvalidValue;            // one form of the mask per launch
    // one epilogue per launch kind (conv16_body, EPI) where the launch is one of those kinds; ST2_CONV16_EPI=0: the general epilogue (read per launch: tests)
    int epi = 0;
    {
        const char* ee = getenv("ST2_CONV16_EPI");
        const bool kinds = !(ee && *ee == '0') && p.M % BM == 0 && !p.mask_src && !p.mask16 && !p.pool32;
        const bool lean16 = kinds && !p.out && !p.inject;
        if (lean16 && !dg && p.out16 && !pools) epi = 1;
        else if (lean16 && !dg && pools && p.pool16 && p.amap && !p.out16 && !p.bits_out && p.H % 2 == 0 && p.W % 2 == 0) epi = 2;
        else if (lean16 && dg && mb && p.out16) epi = 3;
        else if (kinds && !dg && p.out && p.out16 && !pools && !p.inject && cfg == 3 && !sb) epi = 4;
        else if (kinds && dg && mb && p.out16 && !p.out && p.inject && cfg == 3 && !unpool) epi = 5;
        const char* ek = getenv("ST2_CONV16_EPI_KINDS");     // bit (kind - 1): that kind may run (default: all five)
        if (epi && ek && *ek && !((atoi(ek) >> (epi - 1)) & 1)) epi = 0;
    }
    if (epi == 4) { conv3x3_mfma_bf16_64x512_f16o<<<grid, block, 0, s>>>(k); return hipGetLastError(); }
    if (epi == 5) { conv3x3_mfma_bf16_64x512_dgb16i<<<grid, block, 0, s>>>(k); return hipGetLastError(); }
    if (epi && cfg == 0 && sb && !dg) { if (epi == 1) conv3x3_mfma_bf16_64x256_sb_f16<<<grid, block, 0, s>>>(k); else conv3x3_mfma_bf16_64x256_sb_pool<<<grid, block, 0, s>>>(k); return hipGetLastError(); }
    if (epi && cfg == 3 && !dg) { if (epi == 1) conv3x3_mfma_bf16_64x512_f16<<<grid, block, 0, s>>>(k); else conv3x3_mfma_bf16_64x512_pool<<<grid, block, 0, s>>>(k); return hipGetLastError(); }
    if (epi == 3 && cfg == 3) { if (unpool) conv3x3_mfma_bf16_64x512_unpool_b16<<<grid, block, 0, s>>>(k); else conv3x3_mfma_bf16_64x512_dgb16<<<grid, block, 0, s>>>(k); return hipGetLastError(); }
    if (epi == 1 && cfg == 0 && !sb) { conv3x3_mfma_bf16_64x256_f16<<<grid, block, 0, s>>>(k); return hipGetLastError(); }
    if (epi == 3 && cfg == 0 && !sb && !unpool) { conv3x3_mfma_bf16_64x256_dgb16<<<grid, block, 0, s>>>(k); return hipGetLastError(); }
#define ST2_CONV16_LAUNCH3(NAME) do { if (mb) NAME##_dgb<<<grid, block, 0, s>>>(k); else if (dg) NAME##_dg<<<grid, block, 0, s>>>(k); else NAME<<<grid, block, 0, s>>>(k); } while (0)
    if (cfg == 0 && sb) ST2_CONV16_LAUNCH3(conv3x3_mfma_bf16_64x256_sb);
    else if (cfg == 3 && unpool) { if (mb) conv3x3_mfma_bf16_64x512_unpool_b<<<grid, block, 0, s>>>(k); else conv3x3_mfma_bf16_64x512_unpool<<<grid, block, 0, s>>>(k); }
    else if (cfg == 0 && unpool) { if (mb) conv3x3_mfma_bf16_64x256_unpool_b<<<grid, block, 0, s>>>(k); else conv3x3_mfma_bf16_64x256_unpool<<<grid, block, 0, s>>>(k); }
    else if (cfg == 3) ST2_CONV16_LAUNCH3(conv3x3_mfma_bf16_64x512);
    else if (cfg == 0) ST2_CONV16_LAUNCH3(conv3x3_mfma_bf16_64x256);
    else if (cfg == 1) ST2_CONV16_LAUNCH3(conv3x3_mfma_bf16_128x128);
    else ST2_CONV16_LAUNCH3(conv3x3_mfma_bf16_64x128);
#undef ST2_CONV16_LAUNCH3
    return hipGetLastError();
}

// Max-pool backward from the arg-max map of the fused forward pool, all operands channel-blocked:
//   dx16[cb][y][x][8] = (slot(amap[cb][y/2][x/2][j]) == 2 (y & 1) + (x & 1) and the maximum was positive) ? dy16[cb][y/2][x/2][j] : 0
// The "positive" bit is the ReLU mask of the conv blob the pool reads (in-place ReLU: a window whose maximum is 0 passes
// nothing on).  Routing only places values, so rounding dy to bf16 before or after it is the same.
__global__ __launch_bounds__(256) void maxpool_bwd_idx16_k(const uint4* __restrict__ dy16, const uint2* __restrict__ amap,
                                                           uint4* __restrict__ dx16, int CB, int H, int W)
{
    const int ph = (H + 1) / 2, pw = (W + 1) / 2;
    const size_t total = (size_t)CB * H * W;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const int x = (int)(idx % W);
        const size_t r = idx / W;
        const int y = (int)(r % H), cb = (int)(r / H);
        const size_t pi = ((size_t)cb * ph + (y >> 1)) * pw + (x >> 1);
        const uint4 d = dy16[pi];
        const uint2 m = amap[pi];
        const unsigned here = 2u * (y & 1) + (x & 1) + 4u;              // slot | positive
        auto pick = [&](unsigned word, unsigned c0, unsigned c1) {
            return ((c0 & 7u) == here ? (word & 0xffffu) : 0u) | ((c1 & 7u) == here ? (word & 0xffff0000u) : 0u);
        };
        uint4 o;
        o.x = pick(d.x, m.x, m.x >> 8); o.y = pick(d.y, m.x >> 16, m.x >> 24);
        o.z = pick(d.z, m.y, m.y >> 8); o.w = pick(d.w, m.y >> 16, m.y >> 24);
        dx16[idx] = o;
    }
}

hipError_t launch_maxpool_bwd_idx16(const unsigned short* dy16, const unsigned char* amap, unsigned short* dx16, int C, int H, int W, hipStream_t s)
{
    const int CB = (C + 7) / 8;
    const size_t total = (size_t)CB * H * W;
    size_t grid = (total + 255) / 256;
    if (grid > 16384) grid = 16384;
    maxpool_bwd_idx16_k<<<(unsigned)grid, 256, 0, s>>>(reinterpret_cast<const uint4*>(dy16), reinterpret_cast<const uint2*>(amap),
                                                        reinterpret_cast<uint4*>(dx16), CB, H, W);
    return hipGetLastError();
}

// fp32 [C][HW] -> bf16 channel-blocked [C/8][HW][8]; channels beyond C are written as zero
__global__ __launch_bounds__(256) void pack_act16_k(const float* __restrict__ src, unsigned short* __restrict__ dst, int C, size_t hw)
{
    const size_t total = (size_t)((C + 7) / 8) * hw;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const size_t p = idx % hw, cb = idx / hw;
        bf16x8 v;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int c = (int)cb * 8 + j;
            v[j] = (__bf16)(c < C ? src[(size_t)c * hw + p] : 0.0f);
        }
        *reinterpret_cast<bf16x8*>(dst + idx * 8) = v;
    }
}

hipError_t launch_pack_act16(const float* src, unsigned short* dst, int C, size_t hw, hipStream_t s)
{
    const size_t total = (size_t)((C + 7) / 8) * hw;
    size_t grid = (total + 255) / 256;
    if (grid > 65536) grid = 65536;
    pack_act16_k<<<(unsigned)grid, 256, 0, s>>>(src, dst, C, hw);
    return hipGetLastError();
}

}  // namespace st2
