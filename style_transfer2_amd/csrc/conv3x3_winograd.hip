// Winograd F(2x2,3x3) conv3x3 on the fp32 matrix cores (gfx950) -- feasibility probe of the operand feed.
//
// The 16 transform-domain GEMMs of one block keep 16 x (32 m x 32 tiles) fp32 accumulators per wave (256 VGPRs),
// so the block tile is small (128 m x 32 tiles) and the transformed weights U have to stream at ~16 B/clk/CU.
// The probe measures whether that stream can come straight from L2 into MFMA A-operand registers
// (global_load_dwordx4, packed in operand order, no LDS) while the matrix pipe stays busy.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "st2_kernels.h"

namespace st2 {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// U layout: [m32][kpair][posgroup 4][lane 64][4 floats]; one wave-load (dwordx4) = 1 KB contiguous
template <int DEPTH, int ROT>
__global__ __launch_bounds__(256, 1) void wino_probe_k(const float4* __restrict__ U, float* out, int nkp, int n_mt)
{
    __shared__ float vs[2][16 * 64];
    for (int i = threadIdx.x; i < 2 * 16 * 64; i += 256) {
        unsigned h = (i + 1) * 2654435761u + blockIdx.x * 40503u;
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        (&vs[0][0])[i] = (h & 0xffffff) / 8388608.0f - 1.0f;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int xcd = blockIdx.x & 7;
    const int mt = (xcd * 2 + ((blockIdx.x >> 3) & 1)) % n_mt;          // the blocks of one XCD share two 128-m slices
    const float4* up = U + (size_t)(mt * 4 + wave) * nkp * 256 + lane;
    const int rot = ROT ? (int)((blockIdx.x >> 4) & 7) * (nkp / 8) : 0;      // staggered start of the k walk
    f32x16 acc[16];
#pragma unroll
    for (int p = 0; p < 16; ++p)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[p][e] = 0.f;
    // register prefetch ring: slot d holds k-pair (kp + d); the loads of k-pair kp + DEPTH are issued before the
    // MFMAs of k-pair kp.  The empty asm keeps InstCombine from folding the loop-carried loads into "load at use".
    float4 ua[DEPTH][4];
#pragma unroll
    for (int d = 0; d < DEPTH; ++d)
#pragma unroll
        for (int g = 0; g < 4; ++g) ua[d][g] = up[(size_t)(((d + rot) % nkp) * 4 + g) * 64];
    asm volatile("" ::: "memory");
    for (int kp = 0; kp < nkp; kp += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            float4 cur[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) cur[g] = ua[d][g];
            const int nxt = ((kp + d + DEPTH < nkp ? kp + d + DEPTH : kp + d) + rot) % nkp;     // tail: reload (harmless)
#pragma unroll
            for (int g = 0; g < 4; ++g) ua[d][g] = up[((size_t)nxt * 4 + g) * 64];
            asm volatile("" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            const float* vb = &vs[(kp + d) & 1][lane];
            float b[16];
#pragma unroll
            for (int p = 0; p < 16; ++p) b[p] = vb[p * 64];
#pragma unroll
            for (int p = 0; p < 16; ++p) {
                const float a = (p & 3) == 0 ? cur[p >> 2].x : (p & 3) == 1 ? cur[p >> 2].y : (p & 3) == 2 ? cur[p >> 2].z : cur[p >> 2].w;
                acc[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b[p], acc[p], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float r = 0.f;
#pragma unroll
    for (int p = 0; p < 16; ++p)
#pragma unroll
        for (int e = 0; e < 16; ++e) r += acc[p][e];
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = r;
}

hipError_t launch_wino_probe(const float* U, float* out, int blocks, int nkp, int n_mt, int depth, hipStream_t s)
{
    if (nkp <= 0 || nkp % 4 != 0) return hipErrorInvalidValue;
    const float4* u4 = reinterpret_cast<const float4*>(U);
    switch (depth) {
    case 1: wino_probe_k<1, 0><<<blocks, 256, 0, s>>>(u4, out, nkp, n_mt); break;
    case 2: wino_probe_k<2, 0><<<blocks, 256, 0, s>>>(u4, out, nkp, n_mt); break;
    case 4: wino_probe_k<4, 0><<<blocks, 256, 0, s>>>(u4, out, nkp, n_mt); break;
    case 12: wino_probe_k<2, 1><<<blocks, 256, 0, s>>>(u4, out, nkp, n_mt); break;     // depth 2, staggered k walk
    default: return hipErrorInvalidValue;      // nkp must be a multiple of depth (the ring is unrolled by it)
    }
    return hipGetLastError();
}


// ===========================================================================================================
// Winograd F(2x2,3x3) conv3x3 (pad 1, stride 1), NCHW fp32, on v_mfma_f32_32x32x2_f32.
//
//   Y = A^T [ sum_k (G g_k G^T) (.) (B^T d_k B) ] A        per 2x2 output tile, 4x4 input tile d, 3x3 filter g
//
// 16 transform-domain positions = 16 independent GEMMs  Acc_pos[m][tile] += U_pos[m][k] * V_pos[k][tile]:
// 4 multiplies per output instead of 9, so the matrix pipe executes 2.25x fewer flops than the direct kernel
// (conv3x3_mfma.hip) for the same result up to fp32 reassociation (every product and sum is still IEEE fp32).
//
// One workgroup = 4 waves = 128 output channels x (4 rows x 32 columns) pixels = 32 tiles (2 tile rows x 16).
// Wave w owns channels [32w, 32w+32) and all 32 tiles: 16 positions x one 32x32 accumulator = 256 AGPRs,
// which is why the kernel runs one wave per SIMD and everything below is software-pipelined by hand.
//   U (host-transformed weights) never touches LDS: it is packed in MFMA A-operand order
//       [m/32][k/2][pos/4][lane][pos%4]   (lane&31 -> m, lane>>5 -> k parity)
//     and streamed L2 -> VGPR with global_load_dwordx4, two k-pairs ahead (16 B/clk/CU; the blocks that share
//     an XCD walk the same 128-channel slice in step, so the stream is served by that XCD's L2);
//   raw activations: LDS-DMA, 8 channels x 6 rows x 40 floats per chunk (zero fill outside the image = padding),
//     double-buffered, issued two chunks ahead;
//   V: each thread transforms one (tile, channel) pair per chunk (32 adds) and writes the 16 positions to the
//     B-operand image [k-pair][pos][k parity * 32 + tile], double-buffered, one chunk ahead;
//   per k-pair: 16 MFMAs, each followed by a pinned slice of the auxiliary work (operand fetch for the next
//     k-pair, U loads, a quarter of the input transform, DMA issue), one s_barrier per 64 MFMAs.
// Epilogue: the output transform is in-lane (a lane holds all 16 positions of its (m, tile) pairs), then the same
// bias / ReLU / ReLU-mask / injected-diff epilogue as the direct kernel, float2 stores.
// Requirements (else the caller uses the direct kernel): K % 8 == 0, W % 4 == 0, tensors < 4 GiB.
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
    unsigned in_bytes;
};

typedef __attribute__((address_space(3))) void* lptr_t;
constexpr unsigned kOOB = 0xffffffffu;

__device__ __forceinline__ float f4c(const float4& v, int i) { return i == 0 ? v.x : i == 1 ? v.y : i == 2 ? v.z : v.w; }

// WM waves along m (32 channels each) x TG tile groups (32 tiles = 4 rows x 32 columns each), WM * TG = 4:
//   <4,1>: 128 channels x  4 rows x 32 columns      <2,2>: 64 channels x 8 rows x 32 columns
// With TG = 2 the two waves that share a channel slice load the same U lines together (one L2 fetch), and every
// thread transforms two (tile, channel) pairs per chunk.
template <int WM, int TG>
__device__ __forceinline__ void conv3x3_wino_body(const WinoKArgs& a)
{
    static_assert(WM * TG == 4 && (TG == 1 || TG == 2), "4 waves");
    constexpr int BM = 32 * WM;
    constexpr int PROWS = 4 * TG;                        // pixel rows per block
    constexpr int IN_ROWS = PROWS + 2;
    constexpr int PLANE = IN_ROWS * WN_IW;
    constexpr int N_RAW = WN_CH * PLANE;                 // floats staged per chunk
    constexpr int I_PER_WAVE = (N_RAW / 4 + 255) / 256;  // wave-DMAs (64 quads) per wave
    constexpr int RAW = I_PER_WAVE * 1024;               // floats per raw buffer
    constexpr int NU = 4 * TG;                           // transform units of 4 LDS ops (reads, writes); 2 * NU of 4 VALU ops

    __shared__ __attribute__((aligned(16))) float raw_s[2][RAW];
    __shared__ __attribute__((aligned(16))) float v_s[2][TG][WN_V];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave_m = wave / TG, wave_g = wave % TG;

    // XCD-aware bijective block -> tile map, pixel tile fastest: the co-resident blocks of one XCD work on the
    // same channel slice of U (the dominant stream) and on neighbouring pixel tiles.
    const int nwg = gridDim.x, orig = blockIdx.x;
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

    const __amdgpu_buffer_rsrc_t rs_i = __builtin_amdgcn_make_buffer_rsrc((void*)a.in, 0, a.in_bytes, 0x00020000);
    unsigned ioff[I_PER_WAVE];
#pragma unroll
    for (int t = 0; t < I_PER_WAVE; ++t) {
        const int e = ((wave + 4 * t) * 64 + lane) * 4;
        const int c = e / PLANE;
        const int rem = e - c * PLANE;
        const int rr = rem / WN_IW;
        const int col = rem - rr * WN_IW;
        const int gy = y0 - 1 + rr, gx = x0 - 4 + col;
        const bool ok = e < N_RAW && gy >= 0 && gy < a.H && gx >= 0 && gx + 3 < a.W;
        ioff[t] = ok ? ((unsigned)c * plane + (unsigned)gy * a.W + gx) * 4u : kOOB;
    }
    auto dma_raw = [&](int ch, int buf) {
        const unsigned coff = (unsigned)ch * WN_CH * plane * 4u;
#pragma unroll
        for (int t = 0; t < I_PER_WAVE; ++t) {
            const unsigned vo = ioff[t] == kOOB ? kOOB : ioff[t] + coff;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_i, (lptr_t)(raw_s[buf] + (wave + 4 * t) * 256), 16, vo, 0, 0, 0);
        }
    };

    // input transform: this thread owns tile xt of channel xch, in every tile group, of every chunk
    const int xt = tid & 31, xch = tid >> 5;
    const int x_raw = xch * PLANE + (2 * (xt >> 4)) * WN_IW + 2 * (xt & 15) + 3;          // column 3 = pixel x0 - 1
    const int x_v = ((xch >> 1) * 16) * 64 + (xch & 1) * 32 + xt;
    float d[TG][16], wv[TG][16];
    // unit u of the three transform phases (each unit = 4 instructions), u in [0, NU) / [0, 2 NU) / [0, NU)
    auto xf_read = [&](const float* rp, int u) {
        const int g = u >> 2, i = u & 3;
#pragma unroll
        for (int j = 0; j < 4; ++j) d[g][4 * i + j] = rp[(4 * g + i) * WN_IW + j];
    };
    auto xf_math = [&](int u) {
        const int g = u >> 3, h = u & 7;
        if (h < 4) {
            const int j = h;
            wv[g][j] = d[g][j] - d[g][8 + j]; wv[g][4 + j] = d[g][4 + j] + d[g][8 + j];
            wv[g][8 + j] = d[g][8 + j] - d[g][4 + j]; wv[g][12 + j] = d[g][4 + j] - d[g][12 + j];
        } else {
            const int i = h - 4;
            const float v0 = wv[g][4 * i] - wv[g][4 * i + 2], v1 = wv[g][4 * i + 1] + wv[g][4 * i + 2];
            const float v2 = wv[g][4 * i + 2] - wv[g][4 * i + 1], v3 = wv[g][4 * i + 1] - wv[g][4 * i + 3];
            d[g][4 * i] = v0; d[g][4 * i + 1] = v1; d[g][4 * i + 2] = v2; d[g][4 * i + 3] = v3;
        }
    };
    auto xf_write = [&](float* vp, int u) {
        const int g = u >> 2, i = u & 3;
#pragma unroll
        for (int j = 0; j < 4; ++j) vp[g * WN_V + (4 * i + j) * 64] = d[g][4 * i + j];
    };

    const float4* up = a.upack + ((size_t)(mt * WM + wave_m) * nkp) * 256 + lane;
    auto u_load = [&](int kp, int g) -> float4 {
        const int kk = kp < nkp ? kp : nkp - 1;                 // tail: a harmless reload
        return up[((size_t)kk * 4 + g) * 64];
    };

    f32x16 acc[16];
#pragma unroll
    for (int p = 0; p < 16; ++p)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[p][e] = 0.f;

    // ---- prologue: raw chunks 0 and 1, U of k-pairs 0 and 1, V of chunk 0 ----
    dma_raw(0, 0);
    if (a.nch > 1) dma_raw(1, 1);
    float4 ua[2][4];
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
        for (int g = 0; g < 4; ++g) ua[s2][g] = u_load(s2, g);
    asm volatile("" ::: "memory");
    __syncthreads();
#pragma unroll
    for (int u = 0; u < NU; ++u) xf_read(raw_s[0] + x_raw, u);
#pragma unroll
    for (int u = 0; u < 2 * NU; ++u) xf_math(u);
#pragma unroll
    for (int u = 0; u < NU; ++u) xf_write(&v_s[0][0][0] + x_v, u);
    __syncthreads();
    float bv[2][16];
#pragma unroll
    for (int p = 0; p < 16; ++p) bv[0][p] = v_s[0][wave_g][p * 64 + lane];

    for (int c = 0; c < a.nch; ++c) {
        const int cur = c & 1;
        const bool more = c + 1 < a.nch, more2 = c + 2 < a.nch;
        const float* rp = raw_s[cur ^ 1] + x_raw;
        float* vp = &v_s[cur ^ 1][0][0] + x_v;
        const float* bcur = v_s[cur][wave_g] + lane;
        const float* bnxt = v_s[cur ^ 1][wave_g] + lane;
#pragma unroll
        for (int kpl = 0; kpl < 4; ++kpl) {
            const int kp = 4 * c + kpl;
            const int set = kpl & 1;
#pragma unroll
            for (int p = 0; p < 16; ++p) {
                acc[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(f4c(ua[set][p >> 2], p & 3), bv[set][p], acc[p], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                // --- U stream: group g of k-pair kp+2 replaces the group the last four MFMAs consumed ---
                if ((p & 3) == 3) {
                    ua[set][p >> 2] = u_load(kp + 2, p >> 2);
                    asm volatile("" ::: "memory");
                }
                // --- B operands of the next k-pair (after the chunk barrier when it is the next chunk's first) ---
                if (kpl < 3) {
                    if (p >= 1 && p <= 4) {
#pragma unroll
                        for (int jj = 0; jj < 4; ++jj) bv[set ^ 1][4 * (p - 1) + jj] = bcur[((kpl + 1) * 16 + 4 * (p - 1) + jj) * 64];
                    }
                } else if (more) {
                    if (p == 0) {
                        // own V writes done (lgkmcnt), own raw DMA landed (>= 4 U loads were issued after it), all waves here
                        asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)\n\ts_barrier" ::: "memory");
                    }
                    if (p >= 1 && p <= 4) {
#pragma unroll
                        for (int jj = 0; jj < 4; ++jj) bv[0][4 * (p - 1) + jj] = bnxt[(4 * (p - 1) + jj) * 64];
                    }
                }
                // --- input transform of chunk c+1: reads in k-pair 0, arithmetic in 1, writes in 2; raw DMA of chunk c+2 ---
                if (more) {
                    if (kpl == 0 && p >= 5 && p < 5 + NU) xf_read(rp, p - 5);
                    if (kpl == 1 && TG == 1 && p >= 5 && p < 13) xf_math(p - 5);
                    if (kpl == 1 && TG == 2) xf_math(p);
                    if (kpl == 2 && p >= 5 && p < 5 + NU) xf_write(vp, p - 5);
                }
                if (more2 && kpl == 1 && p == 13) dma_raw(c + 2, cur);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }

    // ---- epilogue: output transform (in-lane), then bias / ReLU / mask / inject, float2 stores.
    // Four accumulator rows at a time: their mask / inject loads are issued together, ahead of the arithmetic.
    const int t31 = lane & 31, khalf = lane >> 5;
    const int gy0 = y0 + 4 * wave_g + 2 * (t31 >> 4), gx = x0 + 2 * (t31 & 15);
    const bool has_bias = a.bias != nullptr, has_mask = a.mask_src != nullptr, has_inj = a.inject != nullptr;
    if (gx >= a.W || gy0 >= a.H) return;
    const bool row1 = gy0 + 1 < a.H;
    const int mw = mt * BM + wave_m * 32 + 4 * khalf;
    const unsigned pix0 = (unsigned)gy0 * a.W + gx, pix1 = row1 ? pix0 + a.W : pix0;
#pragma unroll
    for (int eb = 0; eb < 4; ++eb) {
        const int mb = mw + 8 * eb;                         // rows mb .. mb+3 (e = 4 eb + 0..3)
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
            for (int ee = 0; ee < 4; ++ee) {
                mk[ee][0] = *reinterpret_cast<const float2*>(a.mask_src + off[ee] + pix0);
                mk[ee][1] = *reinterpret_cast<const float2*>(a.mask_src + off[ee] + pix1);
            }
        }
        if (has_inj) {
#pragma unroll
            for (int ee = 0; ee < 4; ++ee) {
                ij[ee][0] = *reinterpret_cast<const float2*>(a.inject + off[ee] + pix0);
                ij[ee][1] = *reinterpret_cast<const float2*>(a.inject + off[ee] + pix1);
            }
        }
#pragma unroll
        for (int ee = 0; ee < 4; ++ee) {
            const int e = 4 * eb + ee;
            float tt[2][4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                tt[0][j] = acc[j][e] + acc[4 + j][e] + acc[8 + j][e];
                tt[1][j] = acc[4 + j][e] - acc[8 + j][e] - acc[12 + j][e];
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                float o0 = tt[i][0] + tt[i][1] + tt[i][2] + bs[ee];
                float o1 = tt[i][1] - tt[i][2] - tt[i][3] + bs[ee];
                if (a.relu) { o0 = o0 > 0.f ? o0 : 0.f; o1 = o1 > 0.f ? o1 : 0.f; }
                if (has_mask) { o0 = mk[ee][i].x > 0.f ? o0 : 0.f; o1 = mk[ee][i].y > 0.f ? o1 : 0.f; }
                if (has_inj) { o0 += ij[ee][i].x; o1 += ij[ee][i].y; }
                if (mb + ee < a.M && (i == 0 || row1))
                    *reinterpret_cast<float2*>(a.out + off[ee] + (i ? pix1 : pix0)) = make_float2(o0, o1);
            }
        }
    }
}

__global__ __launch_bounds__(256, 1) void conv3x3_wino_f32_128x128(const WinoKArgs a) { conv3x3_wino_body<4, 1>(a); }
__global__ __launch_bounds__(256, 1) void conv3x3_wino_f32_64x256(const WinoKArgs a) { conv3x3_wino_body<2, 2>(a); }

bool conv_wino_ok(int K, int M, int H, int W)
{
    return K >= 8 && K % 8 == 0 && W % 4 == 0 && M >= 48 && H >= 1 && 4ull * K * H * W < 0xfffffff0ull && 4ull * M * H * W < 0xfffffff0ull;
}

// variant: 0 = 128 channels x 4x32 pixels, 1 = 64 channels x 8x32 pixels, -1 = choose
// p.wpack = the Winograd pack (pack_wino_weights_*); p.bias may be any length >= M
hipError_t launch_conv3x3_wino_cfg(const ConvProblem& p, int variant, hipStream_t s)
{
    if (!conv_wino_ok(p.K, p.M, p.H, p.W) || (reinterpret_cast<uintptr_t>(p.in) & 15) != 0) return hipErrorInvalidValue;
    if (variant < 0) {
        const char* env = getenv("ST2_WINO_CFG");            // forces a variant (tests of both variants on every shape)
        const int forced = env && *env ? atoi(env) : -1;
        const int pad128 = (p.M + 127) / 128 * 128, pad64 = (p.M + 63) / 64 * 64;
        variant = forced >= 0 ? forced : (pad64 < pad128 ? 1 : 0);
    }
    const int bm = variant == 1 ? 64 : 128, prows = variant == 1 ? 8 : 4;
    WinoKArgs k{};
    k.in = p.in; k.upack = reinterpret_cast<const float4*>(p.wpack); k.bias = p.bias; k.out = p.out;
    k.mask_src = p.mask_src; k.inject = p.inject;
    k.K = p.K; k.M = p.M; k.H = p.H; k.W = p.W; k.nch = p.K / WN_CH;
    k.tiles_x = (p.W + 31) / 32; k.tiles_y = (p.H + prows - 1) / prows; k.n_mtiles = (p.M + bm - 1) / bm; k.relu = p.relu;
    k.in_bytes = (unsigned)(4ull * p.K * p.H * p.W);
    const long long nblk = (long long)k.tiles_x * k.tiles_y * k.n_mtiles;
    if (nblk <= 0 || nblk > 0x7fffffffLL) return hipErrorInvalidValue;
    if (variant == 1) conv3x3_wino_f32_64x256<<<dim3((unsigned)nblk), dim3(256), 0, s>>>(k);
    else conv3x3_wino_f32_128x128<<<dim3((unsigned)nblk), dim3(256), 0, s>>>(k);
    return hipGetLastError();
}

hipError_t launch_conv3x3_wino(const ConvProblem& p, hipStream_t s) { return launch_conv3x3_wino_cfg(p, -1, s); }

}  // namespace st2
