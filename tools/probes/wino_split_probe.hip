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

struct WsArgs {
    const float* in; const uint4* upack; const float* bias; float* out;
    int K, M, H, W, nch, tiles_x, tiles_y, relu;
    unsigned in_bytes;
    unsigned long long* stamps;           // per block: {shader cycles of the main loop, 100 MHz ticks of it, cycles prologue, cycles epilogue}
};

__device__ __forceinline__ bf16x8 as_bf(const uint4& u) { return __builtin_bit_cast(bf16x8, u); }
__device__ __forceinline__ unsigned pk_bf16(f32x2 v) { return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2)); }
__device__ __forceinline__ f32x2 bf_pair_as_f32(unsigned h)
{
    f32x2 t;
    t.x = __builtin_bit_cast(float, h << 16);
    t.y = __builtin_bit_cast(float, h & 0xffff0000u);
    return t;
}

template <int STAMP>
__global__ __launch_bounds__(256, 1) void wino_split_probe_k(const WsArgs a)
{
    // raw[2][6400] floats during the main loop; the row exchange of the epilogue afterwards ([src wave][combo][e 8][lane] float2 = 64 KiB)
    __shared__ __attribute__((aligned(16))) float lds[16384];

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

    const __amdgpu_buffer_rsrc_t rs_i = __builtin_amdgcn_make_buffer_rsrc((void*)a.in, 0, a.in_bytes, 0x00020000);
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
        ioff[t] = ok ? ((unsigned)c * plane + (unsigned)gy * a.W + gx) * 4u : kOOB;
    }
    auto dma_raw = [&](int ch, int buf) {
        const unsigned coff = (unsigned)ch * WS_CH * plane * 4u;
#pragma unroll
        for (int t = 0; t < 7; ++t) {
            const int piece = wave + 4 * t;                     // wave-uniform
            if (piece < WS_PIECES) {
                const unsigned vo = ioff[t] == kOOB ? kOOB : ioff[t] + coff;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_i, (lptr_t)(lds + buf * WS_RAW + piece * 256), 16, vo, 0, 0, 0);
            }
        }
    };

    // row i of B^T d:  i = 0: d0 - d2   1: d1 + d2   2: d2 - d1   3: d1 - d3      (X = first, Y = second row; staged row 0 = y0 - 1)
    const int row_x = wave == 0 ? 0 : wave == 2 ? 2 : 1;
    const int row_y = wave == 0 ? 2 : wave == 1 ? 2 : wave == 2 ? 1 : 3;
    const int x_base = (8 * kq) * WS_PLANE + (2 * (t31 >> 4)) * WS_IW + 2 * (t31 & 15) + 3;     // column 3 = pixel x0 - 1
    const float* const pX = lds + x_base + row_x * WS_IW;
    const float* const pY = lds + x_base + row_y * WS_IW;

    // A fragments: [m tile][chunk][pos 16][m group 2][split 3][lane 64] uint4; half-step hs = 2 j + mg of a chunk reads 3 of them
    const uint4* up = a.upack + ((size_t)mt * nch * 16 + 4 * wave) * 6 * 64 + lane;
    uint4 ua[4][3];                      // ring of four half-steps, filled three half-steps ahead
    auto u_fill = [&](int hs_global) {   // hs_global = 8 * chunk + 2 * j + mg
        const int c = hs_global >> 3, j = (hs_global >> 1) & 3, mg = hs_global & 1;
        const uint4* p = up + ((size_t)(c * 16 + j) * 6 + mg * 3) * 64;
#pragma unroll
        for (int s = 0; s < 3; ++s) ua[hs_global & 3][s] = p[s * 64];
        asm volatile("" ::: "memory");
    };

    f32x2 wp[2][4][4];                   // row i of B^T d for this lane's tile: [tile group][channel pair][column]
    auto compute_w = [&](int buf) {
        const float* x = pX + buf * WS_RAW;
        const float* y = pY + buf * WS_RAW;
#pragma unroll
        for (int tg = 0; tg < 2; ++tg)
#pragma unroll
            for (int cp = 0; cp < 4; ++cp)
#pragma unroll
                for (int col = 0; col < 4; ++col) {
                    const int o0 = (2 * cp) * WS_PLANE + tg * 4 * WS_IW + col, o1 = o0 + WS_PLANE;
                    f32x2 X, Y;
                    X.x = x[o0]; X.y = x[o1]; Y.x = y[o0]; Y.y = y[o1];
                    wp[tg][cp][col] = wave == 1 ? X + Y : X - Y;
                }
    };
    uint4 bop[2][6];                     // B operands of the current / next position: [tile group * 3 + split]
    auto build_b = [&](auto j_t, int set) {
        constexpr int j = decltype(j_t)::value;
#pragma unroll
        for (int tg = 0; tg < 2; ++tg) {
            unsigned H[4], Mi[4], L[4];
#pragma unroll
            for (int cp = 0; cp < 4; ++cp) {
                const f32x2 o = j == 0 ? wp[tg][cp][0] - wp[tg][cp][2] : j == 1 ? wp[tg][cp][1] + wp[tg][cp][2]
                              : j == 2 ? wp[tg][cp][2] - wp[tg][cp][1] : wp[tg][cp][1] - wp[tg][cp][3];
                H[cp] = pk_bf16(o);
                const f32x2 r1 = o - bf_pair_as_f32(H[cp]);
                Mi[cp] = pk_bf16(r1);
                const f32x2 r2 = r1 - bf_pair_as_f32(Mi[cp]);
                L[cp] = pk_bf16(r2);
            }
            bop[set][tg * 3 + 0] = make_uint4(H[0], H[1], H[2], H[3]);
            bop[set][tg * 3 + 1] = make_uint4(Mi[0], Mi[1], Mi[2], Mi[3]);
            bop[set][tg * 3 + 2] = make_uint4(L[0], L[1], L[2], L[3]);
        }
    };

    f32x16 acc[16];                      // [position j][m group][tile group]
#pragma unroll
    for (int p = 0; p < 16; ++p)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[p][e] = 0.f;

    // ---- prologue ----
    dma_raw(0, 0);
    if (nch > 1) dma_raw(1, 1);
    u_fill(0); u_fill(1); u_fill(2);
    __builtin_amdgcn_s_waitcnt(0);       // everything landed (prologue only)
    __syncthreads();
    compute_w(0);
    build_b(std::integral_constant<int, 0>{}, 0);

    unsigned long long t0 = 0, r0 = 0;
    if (STAMP) { t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }

    // six partial products, smallest first: (U split, V split)
    constexpr int PA[6] = {2, 0, 1, 1, 0, 0};
    constexpr int PB[6] = {0, 2, 1, 0, 1, 0};

    auto chunk = [&](const int c, auto more_t, auto more2_t) {
        constexpr bool MORE = decltype(more_t)::value, MORE2 = decltype(more2_t)::value;
        // raw(c + 1) has landed in every wave (its DMA was issued a chunk ago, before U loads that have been consumed since), and every
        // wave is done reading raw(c): buffer c & 1 is free for raw(c + 2)
        if (MORE) asm volatile("s_waitcnt vmcnt(9)\n\ts_barrier" ::: "memory");
        if (MORE2) dma_raw(c + 2, c & 1);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int set = j & 1;
#pragma unroll
            for (int mg = 0; mg < 2; ++mg) {
                const int hs = 2 * j + mg;
                if (MORE || hs + 3 < 8) u_fill(8 * c + hs + 3);
#pragma unroll
                for (int p = 0; p < 6; ++p)
#pragma unroll
                    for (int tg = 0; tg < 2; ++tg)
                        acc[j * 4 + mg * 2 + tg] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf(ua[hs & 3][PA[p]]), as_bf(bop[set][tg * 3 + PB[p]]),
                                                                                         acc[j * 4 + mg * 2 + tg], 0, 0, 0);
                if (mg == 0) {
                    if (j == 0) build_b(std::integral_constant<int, 1>{}, set ^ 1);
                    if (j == 1) build_b(std::integral_constant<int, 2>{}, set ^ 1);
                    if (j == 2) build_b(std::integral_constant<int, 3>{}, set ^ 1);
                    if (j == 3 && MORE) compute_w((c + 1) & 1);
                } else if (j == 3 && MORE) build_b(std::integral_constant<int, 0>{}, set ^ 1);
            }
        }
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
extern "C" int st_probe_wino_split(int device_id, int K, int M, int H, int W, int iters, int check, double* avg_ms, double* rel_l2,
                                   double* loop_cycles, double* clock_mhz, double* pro_cycles, double* epi_cycles)
{
    if (K <= 0 || K % 16 || M <= 0 || M % 64 || H <= 0 || H % 8 || W <= 0 || W % 32 || iters <= 0) { snprintf(g_err, sizeof g_err, "shape not supported by the probe"); return 1; }
    WS_TRY(hipSetDevice(device_id));
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
    a.in_bytes = (unsigned)(n_in * 4);
    const int blocks = a.tiles_x * a.tiles_y * (M / 64);
    WS_TRY(hipMalloc((void**)&dst, (size_t)blocks * 32));
    WS_TRY(hipMemset(dst, 0, (size_t)blocks * 32));
    a.stamps = dst;
    hipStream_t s;
    WS_TRY(hipStreamCreate(&s));
    hipEvent_t e0, e1;
    WS_TRY(hipEventCreate(&e0)); WS_TRY(hipEventCreate(&e1));
    for (int i = 0; i < 2; ++i) wino_split_probe_k<0><<<blocks, 256, 0, s>>>(a);
    WS_TRY(hipGetLastError());
    WS_TRY(hipStreamSynchronize(s));
    WS_TRY(hipEventRecord(e0, s));
    for (int i = 0; i < iters; ++i) wino_split_probe_k<0><<<blocks, 256, 0, s>>>(a);
    WS_TRY(hipEventRecord(e1, s));
    WS_TRY(hipStreamSynchronize(s));
    float ms = 0.f;
    WS_TRY(hipEventElapsedTime(&ms, e0, e1));
    if (avg_ms) *avg_ms = ms / iters;
    // stamped launches: main-loop cycles and the shader clock inside it (median block)
    for (int i = 0; i < 3; ++i) wino_split_probe_k<1><<<blocks, 256, 0, s>>>(a);
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
