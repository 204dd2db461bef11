// Gram partials of the bf16 feature path (BASELINE config 3: "bf16 features / fp32 Gram") on the bf16 matrix cores:
//     G_s = F[:, slab_s] F[:, slab_s]^T       fp32 accumulation, F = bf16 features                 worker.py:109-114
// F is read from the bf16 channel-blocked copy [C/8][hw][8] that the forward pass wrote for the next conv.  The Gram
// GEMM contracts over PIXELS, but a 16-byte quad of that copy holds 8 CHANNELS of one pixel -- the transpose of what an
// MFMA fragment wants (8 consecutive k of one row).  ds_read_b64_tr_b16 does the transpose on the way out of LDS: a
// group of 16 lanes reads a 4 (pixels) x 16 (channels) block, lane 4q + p supplying the address of pixel q, channels
// 4p .. 4p+3 (8 bytes), and lane i receiving channel i of the 4 pixels -- two such reads are one operand fragment.
// LDS image of an operand: one row of 64 pixel-quads per channel block (exactly one 1-KiB LDS-DMA piece), rows 1088
// bytes apart (the 64-byte pad puts the four channel blocks a 32-lane half touches on different banks).
// Same slab format, tile order and split-K plan as gram_partial_dma_* (gram.hip): gram_fold_k / gram_reduce_k finish it.
// Bound: the CU's vector-memory ingest / HBM (2 bytes per feature element, once per output tile row).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "st2_kernels.h"

namespace st2 {

typedef float g16_f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 g16_bf16x8 __attribute__((ext_vector_type(8)));
typedef short g16_s4 __attribute__((ext_vector_type(4)));
typedef short g16_s8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void* g16_lptr_t;

constexpr int G16_STEP = 64;                 // pixels staged per step (4 MFMA k-steps)
constexpr int G16_ROWQ = 68;                 // quads per channel-block row in LDS: 64 + 4 of padding (1088 bytes)

// roi (tile-sharded mode): the contraction runs over the hw = rw x rows pixels of a rectangle of the blob (first pixel (y0, x0), row
// pitch `pitch`, `plane` pixels per channel-block row); rw == 0: the whole blob, pixel k at offset k.
template <int BT>
__device__ __forceinline__ void gram16_body(const unsigned short* __restrict__ F16, unsigned f_bytes, float* __restrict__ slabs,
                                            int C, int hw, int tiles_1d, int kslab, GramRoi roi)
{
    constexpr int T = BT / 64;                       // 32x32 MFMA tiles per wave per dimension
    constexpr int CBR = BT / 8;                      // channel-block rows per operand image
    constexpr int IMG = CBR * G16_ROWQ;              // quads per operand image
    constexpr int PPW = CBR / 4;                     // 1-KiB DMA pieces per wave and operand
    __shared__ __attribute__((aligned(16))) uint4 smem[2][2][IMG];       // [stage][A/B]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int n_ut = tiles_1d * (tiles_1d + 1) / 2;
    int tile = blockIdx.x % n_ut;
    const int split = blockIdx.x / n_ut;
    int ti = 0;
    while (tile >= tiles_1d - ti) { tile -= tiles_1d - ti; ++ti; }
    const int tj = ti + tile;
    const int i0 = ti * BT, j0 = tj * BT;
    const bool diag = ti == tj;
    const int kbeg = split * kslab;
    const int kend = min(hw, kbeg + kslab);
    const int nsteps = (kend - kbeg + G16_STEP - 1) / G16_STEP;      // a ragged last step (regions of interest) reads zeros past kend
    const unsigned plane = roi.rw ? (unsigned)roi.plane : (unsigned)hw;

    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)F16, 0, f_bytes, 0x00020000);
    unsigned aoff[PPW], boff[PPW];
#pragma unroll
    for (int t = 0; t < PPW; ++t) {
        const int cbr = wave + 4 * t;                // channel-block row of this piece; lane = pixel within the step
        const int ca = i0 / 8 + cbr, cb = j0 / 8 + cbr;
        aoff[t] = ca * 8 < C ? (unsigned)ca * plane * 16u : 0xffffffffu;
        boff[t] = cb * 8 < C ? (unsigned)cb * plane * 16u : 0xffffffffu;
    }
    auto dma = [&](int step, int stage) {
        const int k = kbeg + step * G16_STEP + lane;                 // this lane's pixel of the contraction
        unsigned pix = (unsigned)k;
        if (roi.rw) { const int ky = k / roi.rw; pix = (unsigned)(roi.y0 + ky) * roi.pitch + roi.x0 + (k - ky * roi.rw); }
        const unsigned so = k < kend ? pix * 16u : 0xffffffffu;
#pragma unroll
        for (int t = 0; t < PPW; ++t) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (g16_lptr_t)(smem[stage][0] + (wave + 4 * t) * G16_ROWQ), 16,
                                                     (aoff[t] == 0xffffffffu || so == 0xffffffffu) ? 0xffffffffu : aoff[t] + so, 0, 0, 0);
            if (!diag)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (g16_lptr_t)(smem[stage][1] + (wave + 4 * t) * G16_ROWQ), 16,
                                                         (boff[t] == 0xffffffffu || so == 0xffffffffu) ? 0xffffffffu : boff[t] + so, 0, 0, 0);
        }
    };

    g16_f32x16 acc[T][T];
#pragma unroll
    for (int i = 0; i < T; ++i)
#pragma unroll
        for (int j = 0; j < T; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

    // transposed-read addressing of this lane (bytes inside an operand image, without the k-step / fragment offsets):
    // group g = lane / 16: rows 16 (g & 1) .. + 15 of the 32-row fragment, k half g >> 1; lane 4 q + p of the group
    // addresses pixel q, channels 4 p .. 4 p + 3 of that 16-channel slice.
    const int g = lane >> 4, i16 = lane & 15, q = i16 >> 2, p = i16 & 3, kh = g >> 1;
    const int lane_byte = ((2 * (g & 1) + (p >> 1)) * G16_ROWQ + 8 * kh + q) * 16 + (p & 1) * 8;
    auto frag = [&](const uint4* img, int row32, int ks) -> g16_bf16x8 {
        // row32: first of the fragment's 32 channels inside the operand image; ks: 16-pixel k-step inside the staged step
        const char* base = reinterpret_cast<const char*>(img) + lane_byte + ((row32 / 8) * G16_ROWQ + ks * 16) * 16;
        const g16_s4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((g16_s4 __attribute__((address_space(3)))*)(base));
        const g16_s4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((g16_s4 __attribute__((address_space(3)))*)(base + 4 * 16));
        g16_s8 v;
        v[0] = lo[0]; v[1] = lo[1]; v[2] = lo[2]; v[3] = lo[3]; v[4] = hi[0]; v[5] = hi[1]; v[6] = hi[2]; v[7] = hi[3];
        return __builtin_bit_cast(g16_bf16x8, v);
    };

    if (nsteps > 0) dma(0, 0);
    for (int st = 0; st < nsteps; ++st) {
        const int cur = st & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                              // step st has landed for every wave; stage cur^1 is free again
        if (st + 1 < nsteps) dma(st + 1, cur ^ 1);
        const uint4* As = smem[cur][0];
        const uint4* Bs = diag ? smem[cur][0] : smem[cur][1];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            g16_bf16x8 a[T], b[T];
#pragma unroll
            for (int i = 0; i < T; ++i) a[i] = frag(As, wm * (T * 32) + i * 32, ks);
#pragma unroll
            for (int j = 0; j < T; ++j) b[j] = frag(Bs, wn * (T * 32) + j * 32, ks);
#pragma unroll
            for (int i = 0; i < T; ++i)
#pragma unroll
                for (int j = 0; j < T; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
    }

    const int khalf = lane >> 5, l31 = lane & 31;
    float* dst = slabs + (size_t)split * C * C;
#pragma unroll
    for (int i = 0; i < T; ++i)
#pragma unroll
        for (int j = 0; j < T; ++j) {
            const int col = j0 + wn * (T * 32) + j * 32 + l31;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = i0 + wm * (T * 32) + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * khalf;
                if (row < C && col < C) dst[(size_t)row * C + col] = acc[i][j][e];     // upper-triangular tiles only: the reduction mirrors
            }
        }
}

__global__ __launch_bounds__(256, 2) void gram16_partial_128(const unsigned short* F16, unsigned f_bytes, float* slabs, int C, int hw, int tiles_1d, int kslab, GramRoi roi)
{ gram16_body<128>(F16, f_bytes, slabs, C, hw, tiles_1d, kslab, roi); }
__global__ __launch_bounds__(256, 2) void gram16_partial_64(const unsigned short* F16, unsigned f_bytes, float* slabs, int C, int hw, int tiles_1d, int kslab, GramRoi roi)
{ gram16_body<64>(F16, f_bytes, slabs, C, hw, tiles_1d, kslab, roi); }

// whole blobs with hw % 64 == 0, C % 8 == 0 and a plan whose slabs are whole 64-pixel steps
bool gram16_ok(int C, int hw, const GramPlan& pl)
{
    return C % 8 == 0 && hw % G16_STEP == 0 && pl.kslab % G16_STEP == 0 && 16ull * (C / 8) * hw < 0xfffffff0ull;
}
// a region of interest of hw pixels inside a blob of `plane` pixels per channel: any hw (the last step is zero-padded)
bool gram16_roi_ok(int C, int hw, size_t plane, const GramPlan& pl)
{
    return C % 8 == 0 && hw > 0 && pl.kslab % G16_STEP == 0 && 16ull * (C / 8) * plane < 0xfffffff0ull;
}

hipError_t launch_gram16_partial(const unsigned short* F16, float* slabs, int C, int hw, const GramPlan& pl, hipStream_t s, const GramRoi* roi)
{
    if ((reinterpret_cast<uintptr_t>(F16) & 15) != 0) return hipErrorInvalidValue;
    if (roi ? !gram16_roi_ok(C, hw, roi->plane, pl) || roi->rw <= 0 : !gram16_ok(C, hw, pl)) return hipErrorInvalidValue;
    const int t1 = (C + pl.bt - 1) / pl.bt;
    const unsigned grid = (unsigned)(pl.tiles * pl.splits);
    const unsigned fb = (unsigned)(16ull * (C / 8) * (roi ? roi->plane : (size_t)hw));
    const GramRoi r = roi ? *roi : GramRoi{0, 0, 0, 0, 0};
    if (pl.bt == 128) gram16_partial_128<<<grid, 256, 0, s>>>(F16, fb, slabs, C, hw, t1, pl.kslab, r);
    else gram16_partial_64<<<grid, 256, 0, s>>>(F16, fb, slabs, C, hw, t1, pl.kslab, r);
    return hipGetLastError();
}

}  // namespace st2
