// Winograd F(2x2,3x3) conv3x3 on the fp32 matrix cores (gfx950) -- feasibility probe of the operand feed.
//
// The 16 transform-domain GEMMs of one block keep 16 x (32 m x 32 tiles) fp32 accumulators per wave (256 VGPRs),
// so the block tile is small (128 m x 32 tiles) and the transformed weights U have to stream at ~16 B/clk/CU.
// The probe measures whether that stream can come straight from L2 into MFMA A-operand registers
// (global_load_dwordx4, packed in operand order, no LDS) while the matrix pipe stays busy.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include "st2_kernels.h"

namespace st2 {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// U layout: [m32][kpair][posgroup 4][lane 64][4 floats]; one wave-load (dwordx4) = 1 KB contiguous
template <int DEPTH>
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
        for (int g = 0; g < 4; ++g) ua[d][g] = up[(size_t)(d * 4 + g) * 64];
    asm volatile("" ::: "memory");
    for (int kp = 0; kp < nkp; kp += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            float4 cur[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) cur[g] = ua[d][g];
            const int nxt = kp + d + DEPTH < nkp ? kp + d + DEPTH : kp + d;     // tail: reload (harmless)
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
    case 1: wino_probe_k<1><<<blocks, 256, 0, s>>>(u4, out, nkp, n_mt); break;
    case 2: wino_probe_k<2><<<blocks, 256, 0, s>>>(u4, out, nkp, n_mt); break;
    case 4: wino_probe_k<4><<<blocks, 256, 0, s>>>(u4, out, nkp, n_mt); break;
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
constexpr int WN_ROWS = 6, WN_IW = 40;           // staged rows y0-1 .. y0+4, columns x0-4 .. x0+35
constexpr int WN_PLANE = WN_ROWS * WN_IW;        // 240
constexpr int WN_RAW = 2048;                     // floats per raw buffer (8 wave-DMAs of 64 quads; 1920 used)
constexpr int WN_V = 4 * 16 * 64;                // floats per V buffer

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

__global__ __launch_bounds__(256, 1) void conv3x3_wino_f32_128x128(const WinoKArgs a)
{
    __shared__ __attribute__((aligned(16))) float raw_s[2][WN_RAW];
    __shared__ __attribute__((aligned(16))) float v_s[2][WN_V];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    // XCD-aware bijective block -> tile map, pixel tile fastest: the co-resident blocks of one XCD work on the
    // same 128-channel slice of U (the dominant stream) and on neighbouring pixel tiles.
    const int nwg = gridDim.x, orig = blockIdx.x;
    const int xcd = orig & 7, q = nwg >> 3, r = nwg & 7;
    const int logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
    const int n_pt = a.tiles_x * a.tiles_y;
    const int mt = logical / n_pt;
    const int pt = logical - mt * n_pt;
    const int tx = pt % a.tiles_x;
    const int ty = pt / a.tiles_x;
    const int y0 = ty * 4, x0 = tx * 32;
    const unsigned plane = (unsigned)a.H * a.W;
    const int nkp = a.K >> 1;

    const __amdgpu_buffer_rsrc_t rs_i = __builtin_amdgcn_make_buffer_rsrc((void*)a.in, 0, a.in_bytes, 0x00020000);
    unsigned ioff[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int e = ((wave + 4 * t) * 64 + lane) * 4;
        const int c = e / WN_PLANE;
        const int rem = e - c * WN_PLANE;
        const int rr = rem / WN_IW;
        const int col = rem - rr * WN_IW;
        const int gy = y0 - 1 + rr, gx = x0 - 4 + col;
        const bool ok = e < WN_CH * WN_PLANE && gy >= 0 && gy < a.H && gx >= 0 && gx + 3 < a.W;
        ioff[t] = ok ? ((unsigned)c * plane + (unsigned)gy * a.W + gx) * 4u : kOOB;
    }
    auto dma_raw = [&](int ch, int buf) {
        const unsigned coff = (unsigned)ch * WN_CH * plane * 4u;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const unsigned vo = ioff[t] == kOOB ? kOOB : ioff[t] + coff;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_i, (lptr_t)(raw_s[buf] + (wave + 4 * t) * 256), 16, vo, 0, 0, 0);
        }
    };

    // input transform: this thread owns tile xt of channel xch of every chunk
    const int xt = tid & 31, xch = tid >> 5;
    const int x_raw = xch * WN_PLANE + (2 * (xt >> 4)) * WN_IW + 2 * (xt & 15) + 3;       // column 3 = pixel x0 - 1
    const int x_v = ((xch >> 1) * 16) * 64 + (xch & 1) * 32 + xt;
    float d[16], wv[16];

    const float4* up = a.upack + ((size_t)(mt * 4 + wave) * nkp) * 256 + lane;
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
    {
        const float* rp = raw_s[0] + x_raw;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) d[4 * i + j] = rp[i * WN_IW + j];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            wv[j] = d[j] - d[8 + j]; wv[4 + j] = d[4 + j] + d[8 + j]; wv[8 + j] = d[8 + j] - d[4 + j]; wv[12 + j] = d[4 + j] - d[12 + j];
        }
        float* vp = v_s[0] + x_v;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            vp[(4 * i + 0) * 64] = wv[4 * i] - wv[4 * i + 2];
            vp[(4 * i + 1) * 64] = wv[4 * i + 1] + wv[4 * i + 2];
            vp[(4 * i + 2) * 64] = wv[4 * i + 2] - wv[4 * i + 1];
            vp[(4 * i + 3) * 64] = wv[4 * i + 1] - wv[4 * i + 3];
        }
    }
    __syncthreads();
    float bv[2][16];
#pragma unroll
    for (int p = 0; p < 16; ++p) bv[0][p] = v_s[0][p * 64 + lane];

    for (int c = 0; c < a.nch; ++c) {
        const int cur = c & 1;
        const bool more = c + 1 < a.nch, more2 = c + 2 < a.nch;
        const float* rp = raw_s[cur ^ 1] + x_raw;
        float* vp = v_s[cur ^ 1] + x_v;
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
                        for (int jj = 0; jj < 4; ++jj) bv[set ^ 1][4 * (p - 1) + jj] = v_s[cur][((kpl + 1) * 16 + 4 * (p - 1) + jj) * 64 + lane];
                    }
                } else if (more) {
                    if (p == 0) {
                        // own V writes done (lgkmcnt), own raw DMA landed (>= 4 U loads were issued after it), all waves here
                        asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)\n\ts_barrier" ::: "memory");
                    }
                    if (p >= 1 && p <= 4) {
#pragma unroll
                        for (int jj = 0; jj < 4; ++jj) bv[0][4 * (p - 1) + jj] = v_s[cur ^ 1][(4 * (p - 1) + jj) * 64 + lane];
                    }
                }
                // --- input transform of chunk c+1, spread over k-pairs 0..2; raw DMA of chunk c+2 ---
                if (more) {
                    if (kpl == 0 && p >= 5 && p <= 8) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) d[4 * (p - 5) + j] = rp[(p - 5) * WN_IW + j];
                    }
                    if (kpl == 1 && p >= 5 && p <= 8) {
                        const int j = p - 5;
                        wv[j] = d[j] - d[8 + j]; wv[4 + j] = d[4 + j] + d[8 + j]; wv[8 + j] = d[8 + j] - d[4 + j]; wv[12 + j] = d[4 + j] - d[12 + j];
                    }
                    if (kpl == 1 && p >= 9 && p <= 12) {
                        const int i = p - 9;
                        const float v0 = wv[4 * i] - wv[4 * i + 2], v1 = wv[4 * i + 1] + wv[4 * i + 2];
                        const float v2 = wv[4 * i + 2] - wv[4 * i + 1], v3 = wv[4 * i + 1] - wv[4 * i + 3];
                        d[4 * i] = v0; d[4 * i + 1] = v1; d[4 * i + 2] = v2; d[4 * i + 3] = v3;
                    }
                    if (kpl == 2 && p >= 5 && p <= 8) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) vp[(4 * (p - 5) + j) * 64] = d[4 * (p - 5) + j];
                    }
                }
                if (more2 && kpl == 1 && p == 13) dma_raw(c + 2, cur);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }

    // ---- epilogue: output transform (in-lane), then bias / ReLU / mask / inject, float2 stores ----
    const int t31 = lane & 31, khalf = lane >> 5;
    const int gy0 = y0 + 2 * (t31 >> 4), gx = x0 + 2 * (t31 & 15);
    const bool has_bias = a.bias != nullptr, has_mask = a.mask_src != nullptr, has_inj = a.inject != nullptr;
    if (gx >= a.W) return;
    const int mw = mt * 128 + wave * 32 + 4 * khalf;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int m = mw + (e & 3) + 8 * (e >> 2);
        if (m >= a.M) continue;
        float tt[2][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            tt[0][j] = acc[j][e] + acc[4 + j][e] + acc[8 + j][e];
            tt[1][j] = acc[4 + j][e] - acc[8 + j][e] - acc[12 + j][e];
        }
        const float bias = has_bias ? a.bias[m] : 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int gy = gy0 + i;
            if (gy >= a.H) continue;
            float o0 = tt[i][0] + tt[i][1] + tt[i][2] + bias;
            float o1 = tt[i][1] - tt[i][2] - tt[i][3] + bias;
            if (a.relu) { o0 = o0 > 0.f ? o0 : 0.f; o1 = o1 > 0.f ? o1 : 0.f; }
            const size_t off = (size_t)m * plane + (size_t)gy * a.W + gx;
            if (has_mask) {
                const float2 mk = *reinterpret_cast<const float2*>(a.mask_src + off);
                o0 = mk.x > 0.f ? o0 : 0.f; o1 = mk.y > 0.f ? o1 : 0.f;
            }
            if (has_inj) {
                const float2 ij = *reinterpret_cast<const float2*>(a.inject + off);
                o0 += ij.x; o1 += ij.y;
            }
            *reinterpret_cast<float2*>(a.out + off) = make_float2(o0, o1);
        }
    }
}

bool conv_wino_ok(int K, int M, int H, int W)
{
    return K >= 8 && K % 8 == 0 && W % 4 == 0 && M >= 96 && H >= 1 && 4ull * K * H * W < 0xfffffff0ull && 4ull * M * H * W < 0xfffffff0ull;
}

// p.wpack = the Winograd pack (pack_wino_weights_*); p.bias may be any length >= M
hipError_t launch_conv3x3_wino(const ConvProblem& p, hipStream_t s)
{
    if (!conv_wino_ok(p.K, p.M, p.H, p.W) || (reinterpret_cast<uintptr_t>(p.in) & 15) != 0) return hipErrorInvalidValue;
    WinoKArgs k{};
    k.in = p.in; k.upack = reinterpret_cast<const float4*>(p.wpack); k.bias = p.bias; k.out = p.out;
    k.mask_src = p.mask_src; k.inject = p.inject;
    k.K = p.K; k.M = p.M; k.H = p.H; k.W = p.W; k.nch = p.K / WN_CH;
    k.tiles_x = (p.W + 31) / 32; k.tiles_y = (p.H + 3) / 4; k.n_mtiles = (p.M + 127) / 128; k.relu = p.relu;
    k.in_bytes = (unsigned)(4ull * p.K * p.H * p.W);
    const long long nblk = (long long)k.tiles_x * k.tiles_y * k.n_mtiles;
    if (nblk <= 0 || nblk > 0x7fffffffLL) return hipErrorInvalidValue;
    conv3x3_wino_f32_128x128<<<dim3((unsigned)nblk), dim3(256), 0, s>>>(k);
    return hipGetLastError();
}

}  // namespace st2
