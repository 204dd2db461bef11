// Style gradient of the bf16 feature path (BASELINE config 3) on the bf16 matrix cores:
//     S = c2 * (D @ F)         D = G - G_style (C x C, fp32, symmetric), F = the blob [C][hw]        worker.py:262-269
// F is read from the bf16 channel-blocked copy [C/8][hw][8] the forward pass already wrote for the next conv: the
// contraction runs over CHANNELS, and a 16-byte quad of that copy is exactly the B fragment of v_mfma_f32_32x32x16_bf16
// (8 consecutive k for one pixel).  D keeps fp32 accuracy: it is split on the device into D = hi + lo (two bf16 terms,
// residual 2^-17) and both halves are multiplied with the same B fragment into one accumulator.  Output, scaling, the
// fused saxpy into the layer diff and the per-block sum of S^2 are fp32, as in style_grad_mfma_f32_* (conv3x3_mfma.hip).
// Bound: HBM for C <= 128 (2 + 4 bytes per element), the CU's vector-memory ingest for C >= 256 (F is re-read once per
// 128-channel output tile, D once per 128-pixel tile, both from L2).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "wave_reduce.h"
#include "st2_kernels.h"

namespace st2 {

typedef float s16_f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 s16_bf16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void* s16_lptr_t;

constexpr int S16_PX = 128;          // pixels per workgroup (4 waves x 32)
constexpr int S16_KC = 64;           // channels per staged chunk (4 MFMA k-steps)

// A operand image of D for the whole layer: quad[((ks * 2 + hl) * 2 + half) * Mp + m] = 8 bf16 of row m, channels
// 16 ks + 8 half .. + 7; hl = 0: hi = bf16(D), hl = 1: lo = bf16(D - hi).  Mp = C rounded up to 64.
__global__ __launch_bounds__(256) void style16_pack_d_k(const float* __restrict__ D, int ld, int C, int Mp, unsigned short* __restrict__ A16)
{
    const int nq = (C / 16) * 2 * 2 * Mp;
    for (int q = blockIdx.x * 256 + threadIdx.x; q < nq; q += gridDim.x * 256) {
        const int m = q % Mp;
        const int r = q / Mp;
        const int half = r & 1, hl = (r >> 1) & 1, ks = r >> 2;
        s16_bf16x8 v;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = 16 * ks + 8 * half + j;
            const float d = m < C ? D[(size_t)m * ld + k] : 0.0f;
            const __bf16 hi = (__bf16)d;
            v[j] = hl ? (__bf16)(d - (float)hi) : hi;
        }
        *reinterpret_cast<s16_bf16x8*>(A16 + (size_t)q * 8) = v;
    }
}

// The same image for the style term fused into the data-gradient conv (conv3x3_mfma_bf16.hip): one k-step = one chunk of that
// kernel, Mp = the conv's padded channel count, and D is scaled by sw / norm * c2 BEFORE the split (the conv epilogue has no
// per-term scaling): quad[((ks * 2 + hl) * 2 + half) * Mp + m].
__global__ __launch_bounds__(256) void style16_pack_d_scaled_k(const float* __restrict__ D, int ld, int C, int Mp, float c2, float sw,
                                                                const float* __restrict__ norm, unsigned short* __restrict__ A16)
{
    const int nq = (C / 16) * 2 * 2 * Mp;
    const float coef = sw / *norm;
    for (int q = blockIdx.x * 256 + threadIdx.x; q < nq; q += gridDim.x * 256) {
        const int m = q % Mp;
        const int r = q / Mp;
        const int half = r & 1, hl = (r >> 1) & 1, ks = r >> 2;
        s16_bf16x8 v;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = 16 * ks + 8 * half + j;
            const float d = m < C ? coef * (D[(size_t)m * ld + k] * c2) : 0.0f;      // same association as style_grad16: coef * (c2 * .)
            const __bf16 hi = (__bf16)d;
            v[j] = hl ? (__bf16)(d - (float)hi) : hi;
        }
        *reinterpret_cast<s16_bf16x8*>(A16 + (size_t)q * 8) = v;
    }
}

size_t style_fuse_pack_elems(int C, int MPad) { return (size_t)(C / 16) * 4 * MPad * 8; }

hipError_t launch_style_fuse_pack(const float* D, int ld, int C, int MPad, float c2, float sw, const float* norm, unsigned short* A16, hipStream_t s)
{
    if (C % 16 != 0 || MPad < C) return hipErrorInvalidValue;
    const int nq = (C / 16) * 4 * MPad;
    style16_pack_d_scaled_k<<<(nq + 255) / 256, 256, 0, s>>>(D, ld, C, MPad, c2, sw, norm, A16);
    return hipGetLastError();
}

// sum_ij (D (D + A))_ij D_ij, one workgroup per 32 x 32 tile of the product, on the fp32 matrix cores.  D and A are symmetric, so both
// operands are read along rows: lane l (0..31) of k-half h holds D[i0 + l][2 kp + h] and (D + A)[j0 + l][2 kp + h].
// The K loop is a chain of load -> 16 MFMAs rounds, each a memory latency long (C = 512: 16 rounds, 32 us for 0.27 GFLOP), and the
// result is LINEAR in the product: the four waves of a workgroup take a quarter of K each and their sums are added (round 4: 62 ->
// ~20 us per step over the four fused style layers at 2048^2).
__global__ __launch_bounds__(256) void style_s2_trace_k(const float* __restrict__ D, int ld, const float* __restrict__ A, int C, float scale,
                                                        float* __restrict__ partial)
{
    __shared__ float red[4];
    const int t1 = C / 32;
    const int ti = blockIdx.x / t1, tj = blockIdx.x - ti * t1;
    const int lane = threadIdx.x & 63, l31 = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const float* drow = D + (size_t)(32 * ti + l31) * ld;
    const float* grow_d = D + (size_t)(32 * tj + l31) * ld;
    const float* grow_a = A + (size_t)(32 * tj + l31) * C;
    s16_f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.0f;
    const int rounds = C / 32, per = (rounds + 3) / 4;
    const int r_end = min(rounds, (wave + 1) * per);
    for (int r = wave * per; r < r_end; ++r) {          // C % 32 == 0; 24 loads in flight, then 16 MFMAs
        const int k0 = 32 * r;
        float4 a4[8], d4[8], t4[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            a4[u] = *reinterpret_cast<const float4*>(drow + k0 + 4 * u);
            d4[u] = *reinterpret_cast<const float4*>(grow_d + k0 + 4 * u);
            t4[u] = *reinterpret_cast<const float4*>(grow_a + k0 + 4 * u);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const float b0 = h ? d4[u].y + t4[u].y : d4[u].x + t4[u].x, b1 = h ? d4[u].w + t4[u].w : d4[u].z + t4[u].z;
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(h ? a4[u].y : a4[u].x, b0, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(h ? a4[u].w : a4[u].z, b1, acc, 0, 0, 0);
        }
    }
    float ss = 0.0f;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int row = 32 * ti + (e & 3) + 8 * (e >> 2) + 4 * h, col = 32 * tj + l31;
        ss += acc[e] * D[(size_t)row * ld + col];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o);
    if (lane == 0) red[wave] = ss;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = ((red[0] + red[1]) + (red[2] + red[3])) * scale;
}

int style_s2_trace_blocks(int C) { return (C / 32) * (C / 32); }

hipError_t launch_style_s2_trace(const float* D, int ld, const float* A, int C, double n, float c2, float* partial, int* n_partial, hipStream_t s)
{
    if (C % 32 != 0 || ld % 4 != 0 || (reinterpret_cast<uintptr_t>(D) & 15) != 0 || (reinterpret_cast<uintptr_t>(A) & 15) != 0) return hipErrorInvalidValue;
    const int nb = style_s2_trace_blocks(C);
    style_s2_trace_k<<<nb, 256, 0, s>>>(D, ld, A, C, (float)((double)c2 * (double)c2 * n), partial);
    if (n_partial) *n_partial = nb;
    return hipGetLastError();
}

struct Style16Args {
    const unsigned short* A16; const unsigned short* F16; float* out; const float* norm; float* partial;
    float c2, sw; int fused, accumulate;
    int C, Mp, n_mtiles; unsigned hw, a_bytes, f_bytes;
    // region of interest (tile-sharded mode): hw counts the pixels of a rectangle of the blob (first pixel (ry0, rx0), rrw wide, row
    // pitch rpitch, rplane pixels per channel); only those pixels are read and written.  rrw == 0: the whole blob.
    int ry0, rx0, rrw, rpitch; unsigned rplane;
};

template <int BM>
__device__ __forceinline__ void style_grad16_body(const Style16Args& a)
{
    constexpr int TM = BM / 32;
    constexpr int AQ = 4 * 2 * 2 * BM;               // quads of the A slab of one chunk: [ks 4][hl 2][half 2][BM]
    constexpr int BQ = 4 * 2 * S16_PX;               // quads of the B tile of one chunk: [ks 4][half 2][128 px]
    constexpr int A_PW = AQ / 256, B_PW = BQ / 256;  // 1-KiB DMA pieces per wave
    // ONE chunk buffer (48 / 32 KiB): the K loop is 1-8 chunks long, so the overlap of staging and MFMA comes from the
    // 3-5 workgroups a CU holds, not from double-buffering inside one; the same memory stages the output rows afterwards
    __shared__ __attribute__((aligned(16))) uint4 smem[AQ + BQ];
    __shared__ float red[4];

    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, khalf = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int mt = blockIdx.x % a.n_mtiles, pt = blockIdx.x / a.n_mtiles;        // the M tiles of one pixel tile run together (L2 reuse of F)
    const int m0 = mt * BM;
    const unsigned p0 = (unsigned)pt * S16_PX;
    const int nch = a.C / S16_KC;
    const unsigned plane = a.rrw ? a.rplane : a.hw;                  // pixels per channel (block) of the blob itself
    auto pixel = [&](unsigned q) -> unsigned {                       // q-th pixel of the (region of the) blob -> offset inside a channel
        if (!a.rrw) return q;
        const unsigned qy = q / (unsigned)a.rrw;
        return (a.ry0 + qy) * (unsigned)a.rpitch + a.rx0 + (q - qy * (unsigned)a.rrw);
    };

    const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc((void*)a.A16, 0, a.a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_f = __builtin_amdgcn_make_buffer_rsrc((void*)a.F16, 0, a.f_bytes, 0x00020000);
    unsigned aoff[A_PW], boff[B_PW];
#pragma unroll
    for (int t = 0; t < A_PW; ++t) {
        const int q = (wave + 4 * t) * 64 + lane;                   // [ks][hl][half][m]
        const int m = q % BM, r = q / BM;                           // r = (ks * 2 + hl) * 2 + half
        aoff[t] = ((unsigned)r * a.Mp + m0 + m) * 16u;
    }
#pragma unroll
    for (int t = 0; t < B_PW; ++t) {
        const int q = (wave + 4 * t) * 64 + lane;                   // [ks][half][px]
        const int px = q % S16_PX, r = q / S16_PX;                  // r = ks * 2 + half = channel block within the chunk
        boff[t] = p0 + px < a.hw ? ((unsigned)r * plane + pixel(p0 + px)) * 16u : 0xffffffffu;
    }

    s16_f32x16 acc[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.0f;

    for (int ch = 0; ch < nch; ++ch) {
        if (ch) __syncthreads();                                    // every wave is done with the previous chunk
        const unsigned ca = (unsigned)ch * 16u * a.Mp * 16u;        // 4 k-steps x 2 x 2 rows of Mp quads
        const unsigned cb = (unsigned)ch * 8u * plane * 16u;        // 8 channel blocks
#pragma unroll
        for (int t = 0; t < A_PW; ++t)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a, (s16_lptr_t)(smem + (wave + 4 * t) * 64), 16, aoff[t] + ca, 0, 0, 0);
#pragma unroll
        for (int t = 0; t < B_PW; ++t)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_f, (s16_lptr_t)(smem + AQ + (wave + 4 * t) * 64), 16,
                                                     boff[t] == 0xffffffffu ? boff[t] : boff[t] + cb, 0, 0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const uint4* A = smem;
        const uint4* B = smem + AQ;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const s16_bf16x8 b = __builtin_bit_cast(s16_bf16x8, B[(ks * 2 + khalf) * S16_PX + wave * 32 + l31]);
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const s16_bf16x8 hi = __builtin_bit_cast(s16_bf16x8, A[((ks * 2 + 0) * 2 + khalf) * BM + i * 32 + l31]);
                const s16_bf16x8 lo = __builtin_bit_cast(s16_bf16x8, A[((ks * 2 + 1) * 2 + khalf) * BM + i * 32 + l31]);
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(hi, b, acc[i], 0, 0, 0);
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(lo, b, acc[i], 0, 0, 0);
            }
        }
    }

    // ---- epilogue (C/D map: column = lane & 31 = pixel, row = (e & 3) + 8 (e >> 2) + 4 (lane >> 5))
    const unsigned p = p0 + wave * 32 + l31;
    const bool live = p < a.hw;
    const float coef = a.fused ? a.sw / *a.norm : 0.0f;
    float ss = 0.0f;
    // (a region of interest takes the 16-byte store path when its rows start and end on 16-byte boundaries of the blob)
    if ((a.hw & 3u) == 0 && (!a.rrw || ((a.rrw | a.rx0 | a.rpitch) & 3) == 0)) {
        // Rows of 128 pixels through LDS: a lane-per-pixel store is 4 bytes per lane and channel (64 store instructions per
        // wave, issue-bound); staged, every lane stores 16 bytes of one channel row (4x fewer instructions, whole lines).
        float* stage = reinterpret_cast<float*>(smem);              // [64 channels][128 pixels]
#pragma unroll
        for (int half = 0; half < BM / 64; ++half) {
            __syncthreads();                                        // the operand chunk / the previous half is consumed
#pragma unroll
            for (int ii = 0; ii < 2; ++ii) {
                const int i = 2 * half + ii;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int ml = ii * 32 + (e & 3) + 8 * ((e >> 2) & 1) + 4 * khalf + 16 * (e >> 3);     // channel within this half
                    const float v = live && m0 + 64 * half + ml < a.C ? acc[i][e] * a.c2 : 0.0f;
                    ss += v * v;
                    stage[ml * S16_PX + wave * 32 + l31] = a.fused ? coef * v : v;
                }
            }
            __syncthreads();
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const int idx = t * 256 + tid, row = idx >> 5, c4 = (idx & 31) * 4;
                const int m = m0 + 64 * half + row;
                if (m < a.C && p0 + c4 < a.hw) {
                    float4 v = *reinterpret_cast<const float4*>(stage + row * S16_PX + c4);
                    float* dst = a.out + (size_t)m * plane + pixel(p0 + c4);
                    if (a.fused && a.accumulate) { const float4 o = *reinterpret_cast<const float4*>(dst); v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
                    *reinterpret_cast<float4*>(dst) = v;
                }
            }
        }
    } else {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int mbase = m0 + i * 32 + 4 * khalf + 16 * h;
                float v[8], old[8];
                unsigned off[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int m = mbase + (e & 3) + 8 * (e >> 2);
                    off[e] = (unsigned)(m < a.C ? m : a.C - 1) * plane + (live ? pixel(p) : 0u);
                    v[e] = live && m < a.C ? acc[i][8 * h + e] * a.c2 : 0.0f;
                    ss += v[e] * v[e];
                }
                if (a.fused && a.accumulate) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) old[e] = a.out[off[e]];
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float r2 = a.fused ? coef * v[e] + (a.accumulate ? old[e] : 0.0f) : v[e];
                    if (live && mbase + (e & 3) + 8 * (e >> 2) < a.C) a.out[off[e]] = r2;
                }
            }
    }
    float sv[1] = {ss};
    block_sum(sv, red);
    if (tid == 0) a.partial[blockIdx.x] = sv[0];
}

// non-template entry points (a template kernel with a launch bound loses its host stub with this toolchain)
__global__ __launch_bounds__(256) void style_grad16_128(const Style16Args a) { style_grad16_body<128>(a); }
__global__ __launch_bounds__(256) void style_grad16_64(const Style16Args a) { style_grad16_body<64>(a); }

bool style_grad16_ok(int C, size_t hw)
{
    return C >= 64 && C % 64 == 0 && hw > 0 && 16ull * (C / 8) * hw < 0xfffffff0ull && 4ull * C * hw < 0xfffffff0ull;
}
static int style16_bm(int C) { return C >= 128 ? 128 : 64; }
size_t style_grad16_pack_elems(int C) { return (size_t)(C / 16) * 4 * ((C + 127) / 128 * 128) * 8; }
int style_grad16_blocks(int C, size_t hw) { return (int)((hw + S16_PX - 1) / S16_PX) * ((C + style16_bm(C) - 1) / style16_bm(C)); }

hipError_t launch_style_grad16(const float* Dp, int ld, unsigned short* A16, const unsigned short* F16, float* dst, float c2, int fused,
                               float sw, const float* norm, int accumulate, float* partial, int* n_partial, int C, size_t hw, hipStream_t s,
                               const GramRoi* roi)
{
    // roi: hw = the region's pixel count; the blob itself has roi->plane pixels per channel
    if (!style_grad16_ok(C, roi ? roi->plane : hw) || hw == 0 || (roi && (roi->rw <= 0 || hw > roi->plane))) return hipErrorInvalidValue;
    const int bm = style16_bm(C), Mp = (C + 127) / 128 * 128;
    const int nq = (C / 16) * 4 * Mp;
    style16_pack_d_k<<<(nq + 255) / 256, 256, 0, s>>>(Dp, ld, C, Mp, A16);
    Style16Args a{};
    a.A16 = A16; a.F16 = F16; a.out = dst; a.norm = norm; a.partial = partial;
    a.c2 = c2; a.sw = sw; a.fused = fused; a.accumulate = accumulate;
    a.C = C; a.Mp = Mp; a.n_mtiles = (C + bm - 1) / bm; a.hw = (unsigned)hw;
    a.a_bytes = (unsigned)(style_grad16_pack_elems(C) * 2); a.f_bytes = (unsigned)(16ull * (C / 8) * (roi ? roi->plane : hw));
    if (roi) { a.ry0 = roi->y0; a.rx0 = roi->x0; a.rrw = roi->rw; a.rpitch = roi->pitch; a.rplane = (unsigned)roi->plane; }
    const int grid = style_grad16_blocks(C, hw);
    if (n_partial) *n_partial = grid;
    if (bm == 128) style_grad16_128<<<grid, 256, 0, s>>>(a);
    else style_grad16_64<<<grid, 256, 0, s>>>(a);
    return hipGetLastError();
}

}  // namespace st2
