// tools/probes/probes.hip -- development probes and the isolated conv timing hook (tools/probes/st2_probes.h).
// Built into tools/probes/libst2_probes.so by style_transfer2_amd/build.py:build_probes(); links against
// libst2_hip.so for the product's conv launchers (st2::launch_conv3x3_cfg, ...).  Not product code.
//
// Record (round 1, gpurun_out/wino_probe.txt): st_bench_wino_probe(depth = 4) ended in "Memory access fault by GPU"
// on an MI355X.  That instantiation (wino_probe_k<4,0>) was the ONLY kernel of the library that needed a private
// segment: 256 accumulators + a 64-register U ring + operands exhaust the 512-register file, and hipcc spills through
// scratch (.private_segment_fixed_size 12, .vgpr_spill_count > 0; its indexing is the same bounded `% nkp` walk as the
// depth-1/2 instantiations, which run clean).  The instantiation is deleted and depth 4 is rejected; no product kernel
// uses scratch, which tests/test_boundary.py::test_no_product_kernel_uses_scratch now enforces on the built library.
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>
#include "../../style_transfer2_amd/csrc/st2_kernels.h"
#include "st2_probes.h"

namespace st2 {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------------------------------
// Ceiling probes (measurement only): what the matrix pipe sustains on this chip for the same
// instruction with (variant 0) register operands only, (variant 1) + the conv kernel's LDS operand
// reads, (variant 2) variant 1 at 1 wave per SIMD.
// ------------------------------------------------------------------------------------------
// variant: 0 registers only (smooth data) | 1 LDS reads, smooth data | 2 LDS reads, random data |
//          3 = 2 with the conv kernel's pinned issue order | 4 = random register operands, no LDS
template <int VARIANT>
__global__ __launch_bounds__(256) void mfma_probe_k(float* out, int iters, float seed)
{
    __shared__ float lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 256) {
        unsigned h = (i + 1) * 2654435761u + blockIdx.x * 40503u;
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        lds[i] = (VARIANT >= 2) ? ((h & 0xffffff) / 8388608.0f - 1.0f) : seed + i * 1e-4f;
    }
    __syncthreads();
    f32x16 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    float a0 = seed + threadIdx.x * 1e-3f, a1 = a0 * 0.5f, b0 = 1.f - a0, b1 = b0 * 0.25f;
    if (VARIANT == 4) { a0 = lds[threadIdx.x]; a1 = lds[threadIdx.x + 256]; b0 = lds[threadIdx.x + 512]; b1 = lds[threadIdx.x + 768]; }
    const float* base = lds + (threadIdx.x & 63);
    for (int it = 0; it < iters; ++it) {
        if (VARIANT == 3) {
            float av[2][2], bv[2][2];
            av[0][0] = base[0]; av[0][1] = base[32]; bv[0][0] = base[4096]; bv[0][1] = base[4096 + 34];
#pragma unroll
            for (int s2 = 0; s2 < 18; ++s2) {
#pragma unroll
                for (int ij = 0; ij < 4; ++ij) {
                    acc[ij] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s2 & 1][ij >> 1], bv[s2 & 1][ij & 1], acc[ij], 0, 0, 0);
                    if (ij == 0) {
                        __builtin_amdgcn_sched_barrier(0);
                        if (s2 + 1 < 18) {
                            av[(s2 + 1) & 1][0] = base[(s2 + 1) * 128]; av[(s2 + 1) & 1][1] = base[(s2 + 1) * 128 + 32];
                            bv[(s2 + 1) & 1][0] = base[4096 + (s2 + 1) * 70]; bv[(s2 + 1) & 1][1] = base[4096 + (s2 + 1) * 70 + 34];
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
#pragma unroll
            for (int s2 = 0; s2 < 18; ++s2) {
                if (VARIANT >= 1 && VARIANT <= 2) {
                    a0 = base[s2 * 128]; a1 = base[s2 * 128 + 32];
                    b0 = base[4096 + s2 * 70]; b1 = base[4096 + s2 * 70 + 34];
                }
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[1], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[2], 0, 0, 0);
                acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[3], 0, 0, 0);
            }
        }
    }
    float r = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) r += acc[i][e];
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = r;
}

hipError_t launch_mfma_probe(int variant, float* out, int blocks, int iters, hipStream_t s)
{
    switch (variant) {
    case 0: mfma_probe_k<0><<<blocks, 256, 0, s>>>(out, iters, 0.37f); break;
    case 1: mfma_probe_k<1><<<blocks, 256, 0, s>>>(out, iters, 0.37f); break;
    case 2: mfma_probe_k<2><<<blocks, 256, 0, s>>>(out, iters, 0.37f); break;
    case 3: mfma_probe_k<3><<<blocks, 256, 0, s>>>(out, iters, 0.37f); break;
    default: mfma_probe_k<4><<<blocks, 256, 0, s>>>(out, iters, 0.37f); break;
    }
    return hipGetLastError();
}


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

// Issue-rate probe: one wave per SIMD (256 accumulators), 16 independent MFMAs per iteration with register operands,
// NAUX independent VALU / NLDS ds_read instructions pinned after each; reports shader cycles per MFMA.
template <int NAUX, int NLDS>
__global__ __launch_bounds__(256, 1) void wino_issue_probe_k(float* out, unsigned long long* cycles, int iters, float seed)
{
    __shared__ float lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = seed + i;
    __syncthreads();
    f32x16 acc[16];
#pragma unroll
    for (int p = 0; p < 16; ++p)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[p][e] = 0.f;
    float a0 = seed + threadIdx.x * 1e-3f, b0 = 1.f - a0;
    float x[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) x[j] = seed * (j + 1);
    float ld[4] = {0.f, 0.f, 0.f, 0.f};
    const float* lp = lds + (threadIdx.x & 63);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int p = 0; p < 16; ++p) {
            acc[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[p], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < NAUX; ++j) x[j & 7] = x[j & 7] * 1.0001f + seed;
#pragma unroll
            for (int j = 0; j < NLDS; ++j) ld[j & 3] += lp[(p * 4 + j) * 64];
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float r = 0.f;
#pragma unroll
    for (int p = 0; p < 16; ++p)
#pragma unroll
        for (int e = 0; e < 16; ++e) r += acc[p][e];
#pragma unroll
    for (int j = 0; j < 8; ++j) r += x[j];
    r += ld[0] + ld[1] + ld[2] + ld[3];
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = r;
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

// Feed probe for the LDS-staged U design: 64 channels x 64 tiles per workgroup; per k-pair every wave issues 2 LDS-DMA
// pieces (its share of the 8-KiB U slab) + EXTRA_DMA more (standing in for the raw activation tile), one barrier,
// 8 ds_read_b128 (A and B operands of the next k-pair), 16 MFMAs.  Reports shader cycles per k-pair.
typedef __attribute__((address_space(3))) void* lptr_probe_t;
template <int EXTRA_DMA>
__global__ __launch_bounds__(256, 1) void wino_lds_probe_k(const float* __restrict__ U, unsigned u_bytes, float* out,
                                                           unsigned long long* cycles, int nkp)
{
    __shared__ __attribute__((aligned(16))) float u_s[4][2048];
    __shared__ __attribute__((aligned(16))) float v_s[2][1024];
    __shared__ __attribute__((aligned(16))) float junk[4][256];
    for (int i = threadIdx.x; i < 2048; i += 256) (&v_s[0][0])[i] = 0.001f * i;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wave_m = wave >> 1;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)U, 0, u_bytes, 0x00020000);
    const unsigned mt = (blockIdx.x & 7);                      // one 64-channel slab per XCD
    const unsigned lane_off = (unsigned)lane * 16u;
    auto dma_u = [&](int kp) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int piece = wave + 4 * t;                    // 8 pieces of 1 KiB: slice = piece / 4, quarter = piece % 4
            const unsigned src = ((mt * 2 + piece / 4) * (unsigned)nkp + (unsigned)kp) * 4096u + (piece % 4) * 1024u;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lptr_probe_t)(u_s[kp & 3] + piece * 256), 16, lane_off, src, 0, 0);
        }
#pragma unroll
        for (int t = 0; t < EXTRA_DMA; ++t)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lptr_probe_t)(junk[wave]), 16, lane_off, (unsigned)(kp * 4 + t) * 1024u, 0, 0);
    };
    f32x16 acc[16];
#pragma unroll
    for (int p = 0; p < 16; ++p)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[p][e] = 0.f;
    f32x4 aq[2][4], bq[2][4];
    dma_u(0); dma_u(1);
    __syncthreads();
#pragma unroll
    for (int pg = 0; pg < 4; ++pg) {
        aq[0][pg] = *reinterpret_cast<const f32x4*>(&u_s[0][wave_m * 1024 + (pg * 64 + lane) * 4]);
        bq[0][pg] = *reinterpret_cast<const f32x4*>(&v_s[0][(pg * 64 + lane) * 4]);
    }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int kp0 = 0; kp0 < nkp; kp0 += 4) {
#pragma unroll
        for (int kpl = 0; kpl < 4; ++kpl) {
            const int kp = kp0 + kpl, set = kpl & 1;
#pragma unroll
            for (int p = 0; p < 16; ++p) {
                acc[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(aq[set][p >> 2][p & 3], bq[set][p >> 2][p & 3], acc[p], 0, 0, 0);
                if (p == 0) {
                    __builtin_amdgcn_sched_barrier(0);
                    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll
                    for (int pg = 0; pg < 4; ++pg) {
                        aq[set ^ 1][pg] = *reinterpret_cast<const f32x4*>(&u_s[(kpl + 1) & 3][wave_m * 1024 + (pg * 64 + lane) * 4]);
                        bq[set ^ 1][pg] = *reinterpret_cast<const f32x4*>(&v_s[set ^ 1][(pg * 64 + lane) * 4]);
                    }
                    dma_u(kp + 2 < nkp ? kp + 2 : kp);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float r = 0.f;
#pragma unroll
    for (int p = 0; p < 16; ++p)
#pragma unroll
        for (int e = 0; e < 16; ++e) r += acc[p][e];
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = r + junk[0][threadIdx.x & 255];
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

hipError_t launch_wino_lds_probe(int extra_dma, const float* U, unsigned u_bytes, float* out, unsigned long long* cycles,
                                 int blocks, int nkp, hipStream_t s)
{
    if (nkp < 4 || nkp % 4) return hipErrorInvalidValue;
    switch (extra_dma) {
    case 0: wino_lds_probe_k<0><<<blocks, 256, 0, s>>>(U, u_bytes, out, cycles, nkp); break;
    case 1: wino_lds_probe_k<1><<<blocks, 256, 0, s>>>(U, u_bytes, out, cycles, nkp); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_wino_issue_probe(int naux, int nlds, float* out, unsigned long long* cycles, int blocks, int iters, hipStream_t s)
{
#define ST2_IP(A, L) if (naux == A && nlds == L) { wino_issue_probe_k<A, L><<<blocks, 256, 0, s>>>(out, cycles, iters, 0.37f); return hipGetLastError(); }
    ST2_IP(0, 0) ST2_IP(2, 0) ST2_IP(4, 0) ST2_IP(8, 0) ST2_IP(12, 0) ST2_IP(0, 2) ST2_IP(0, 4) ST2_IP(4, 2) ST2_IP(4, 4)
#undef ST2_IP
    return hipErrorInvalidValue;
}

hipError_t launch_wino_probe(const float* U, float* out, int blocks, int nkp, int n_mt, int depth, hipStream_t s)
{
    if (nkp <= 0 || nkp % 4 != 0) return hipErrorInvalidValue;
    const float4* u4 = reinterpret_cast<const float4*>(U);
    switch (depth) {
    case 1: wino_probe_k<1, 0><<<blocks, 256, 0, s>>>(u4, out, nkp, n_mt); break;
    case 2: wino_probe_k<2, 0><<<blocks, 256, 0, s>>>(u4, out, nkp, n_mt); break;
    case 12: wino_probe_k<2, 1><<<blocks, 256, 0, s>>>(u4, out, nkp, n_mt); break;     // depth 2, staggered k walk
    default: return hipErrorInvalidValue;      // nkp must be a multiple of depth (the ring is unrolled by it)
    }
    return hipGetLastError();
}



}  // namespace st2

using namespace st2;

enum { ST_OK = 0, ST_ERR_ARG = 1, ST_ERR_STATE = 2, ST_ERR_HIP = 3 };
static thread_local char g_err[1024] = "";
static int fail(int code, const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}
#define HIP_TRY(expr)                                                                               \
    do {                                                                                            \
        hipError_t e_ = (expr);                                                                     \
        if (e_ != hipSuccess)                                                                       \
            return fail(ST_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)
#define ST_TRY(expr)                    \
    do {                                \
        int r_ = (expr);                \
        if (r_ != ST_OK) return r_;     \
    } while (0)
static int dmalloc(float** p, size_t nfloats)
{
    void* q = nullptr;
    hipError_t e = hipMalloc(&q, std::max<size_t>(nfloats, 1) * sizeof(float));
    if (e != hipSuccess) return fail(ST_ERR_HIP, "hipMalloc(%zu floats): %s", nfloats, hipGetErrorString(e));
    *p = (float*)q;
    return ST_OK;
}
static void dfree(float*& p)
{
    if (p && hipFree(p) != hipSuccess) (void)hipGetLastError();
    p = nullptr;
}

extern "C" {

const char* st_probe_last_error(void) { return g_err; }

int st_bench_mfma(int device_id, int variant, int blocks_per_cu, double* tflops)
{
    if (!tflops || blocks_per_cu <= 0) return fail(ST_ERR_ARG, "bad argument");
    HIP_TRY(hipSetDevice(device_id));
    const int blocks = 256 * blocks_per_cu, iters = 2000;
    float* out = nullptr;
    ST_TRY(dmalloc(&out, (size_t)blocks * 256));
    hipStream_t s;
    HIP_TRY(hipStreamCreate(&s));
    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreate(&e0)); HIP_TRY(hipEventCreate(&e1));
    (void)launch_mfma_probe(variant, out, blocks, iters, s);
    (void)hipEventRecord(e0, s);
    const int reps = 5;
    for (int i = 0; i < reps; ++i) (void)launch_mfma_probe(variant, out, blocks, iters, s);
    (void)hipEventRecord(e1, s);
    HIP_TRY(hipStreamSynchronize(s));
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)reps * blocks * 4 /*waves*/ * iters * 72.0 * 4096.0;
    *tflops = flops / (ms * 1e-3) / 1e12;
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); (void)hipStreamDestroy(s);
    dfree(out);
    return ST_OK;
}

int st_bench_lds_feed_probe(int device_id, int extra_dma, int K, int blocks, double* cycles_per_kpair)
{
    if (!cycles_per_kpair || K < 8 || K % 8 || blocks <= 0) return fail(ST_ERR_ARG, "bad argument");
    HIP_TRY(hipSetDevice(device_id));
    const int nkp = K / 2;
    const size_t n_u = (size_t)16 * nkp * 1024;             // 8 slabs of 2 x 32 channels
    float *out = nullptr, *U = nullptr;
    unsigned long long* cyc = nullptr;
    ST_TRY(dmalloc(&out, (size_t)blocks * 256)); ST_TRY(dmalloc(&U, n_u));
    HIP_TRY(hipMemset(U, 0x3c, n_u * 4));
    HIP_TRY(hipMalloc((void**)&cyc, blocks * sizeof(unsigned long long)));
    hipStream_t s;
    HIP_TRY(hipStreamCreate(&s));
    int rc = ST_OK;
    for (int i = 0; i < 20 && rc == ST_OK; ++i)
        if (launch_wino_lds_probe(extra_dma, U, (unsigned)(n_u * 4), out, cyc, blocks, nkp, s) != hipSuccess) rc = fail(ST_ERR_ARG, "no such probe variant");
    if (rc == ST_OK) {
        HIP_TRY(hipStreamSynchronize(s));
        std::vector<unsigned long long> h(blocks);
        HIP_TRY(hipMemcpy(h.data(), cyc, blocks * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        std::sort(h.begin(), h.end());
        *cycles_per_kpair = (double)h[blocks / 2] / nkp;
    }
    (void)hipStreamDestroy(s);
    dfree(out); dfree(U); (void)hipFree(cyc);
    return rc;
}

int st_bench_issue_probe(int device_id, int naux, int nlds, double* cycles_per_mfma)
{
    if (!cycles_per_mfma) return fail(ST_ERR_ARG, "bad argument");
    HIP_TRY(hipSetDevice(device_id));
    const int blocks = 256, iters = 20000;
    float* out = nullptr;
    unsigned long long* cyc = nullptr;
    ST_TRY(dmalloc(&out, (size_t)blocks * 256));
    HIP_TRY(hipMalloc((void**)&cyc, blocks * sizeof(unsigned long long)));
    hipStream_t s;
    HIP_TRY(hipStreamCreate(&s));
    int rc = ST_OK;
    for (int i = 0; i < 2 && rc == ST_OK; ++i)
        if (launch_wino_issue_probe(naux, nlds, out, cyc, blocks, iters, s) != hipSuccess) rc = fail(ST_ERR_ARG, "no such probe variant");
    if (rc == ST_OK) {
        HIP_TRY(hipStreamSynchronize(s));
        std::vector<unsigned long long> h(blocks);
        HIP_TRY(hipMemcpy(h.data(), cyc, blocks * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        std::sort(h.begin(), h.end());
        *cycles_per_mfma = (double)h[blocks / 2] / ((double)iters * 16.0);
    }
    (void)hipStreamDestroy(s);
    dfree(out); (void)hipFree(cyc);
    return rc;
}

int st_bench_wino_probe(int device_id, int blocks_per_cu, int K, int M, int depth, double* tflops)
{
    if (!tflops || blocks_per_cu == 0 || K < 8 || K % 8 || M < 128 || M % 128 || (depth != 1 && depth != 2 && depth != 12))
        return fail(ST_ERR_ARG, "bad argument");
    HIP_TRY(hipSetDevice(device_id));
    const int blocks = blocks_per_cu > 0 ? 256 * blocks_per_cu : -blocks_per_cu, nkp = K / 2, n_mt = M / 128;   // < 0: absolute block count
    const size_t n_u = (size_t)(M / 32) * nkp * 256 * 4;
    float *out = nullptr, *U = nullptr;
    ST_TRY(dmalloc(&out, (size_t)blocks * 256)); ST_TRY(dmalloc(&U, n_u));
    HIP_TRY(hipMemset(U, 0x3c, n_u * 4));
    hipStream_t s;
    HIP_TRY(hipStreamCreate(&s));
    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreate(&e0)); HIP_TRY(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) (void)launch_wino_probe(U, out, blocks, nkp, n_mt, depth, s);
    (void)hipEventRecord(e0, s);
    const int reps = 10;
    for (int i = 0; i < reps; ++i) (void)launch_wino_probe(U, out, blocks, nkp, n_mt, depth, s);
    (void)hipEventRecord(e1, s);
    HIP_TRY(hipStreamSynchronize(s));
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    *tflops = (double)reps * blocks * 4 * nkp * 16.0 * 4096.0 / (ms * 1e-3) / 1e12;
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); (void)hipStreamDestroy(s);
    dfree(out); dfree(U);
    return ST_OK;
}

int st_conv_num_configs(void) { return conv_num_configs(); }
const char* st_conv_config_name(int cfg) { return conv_config_name(cfg); }

int st_bench_conv(int device_id, int K, int M, int H, int W, int cfg, int dgrad_epilogue, int iters,
                  double* avg_ms, int* cfg_used)
{
    if (K <= 0 || M <= 0 || H <= 0 || W <= 0 || iters <= 0 || !avg_ms) return fail(ST_ERR_ARG, "bad argument");
    HIP_TRY(hipSetDevice(device_id));
    const size_t n_in = (size_t)K * H * W, n_out = (size_t)M * H * W;
    std::vector<float> w((size_t)M * K * 9), pk(conv_pack_floats(K, M)), hin(n_in), hb(conv_mpad(M), 0.1f);
    uint32_t st = 12345u;
    auto rnd = [&]() { st = st * 1664525u + 1013904223u; return ((st >> 8) & 0xffff) / 32768.0f - 1.0f; };
    for (auto& x : w) x = rnd() * 0.05f;
    // ST2_BENCH_ZERO=1: all-zero activations (same instruction stream, far less switching energy): tells a power-held clock
    // from a cycle-bound loop (MI355X_MICROARCH.md, DVFS give-back)
    const bool zero_in = getenv("ST2_BENCH_ZERO") && *getenv("ST2_BENCH_ZERO") == '1';
    for (auto& x : hin) x = zero_in ? 0.f : rnd();
    const bool wino = cfg >= 100;      // 100: choose, 101: 128 ch x 4x32 px, 102: 64 ch x 8x32 px, 104: 64 ch x 8x32 px position-split, 107: 128 ch x 4x32 px with 8 waves, 109: 64 ch x 4x32 px half tile (two workgroups per CU); 103/106/105/108/110: those stamped
    if (wino) {
        if (!conv_wino_ok(K, M, H, W)) return fail(ST_ERR_ARG, "shape not eligible for the Winograd kernel");
        pk.assign(wino_pack_floats(K, M), 0.f);
        pack_wino_weights_fwd(w.data(), M, K, pk.data());
    } else
    pack_conv_weights_fwd(w.data(), M, K, pk.data());
    float *din = nullptr, *dw = nullptr, *db = nullptr, *dout = nullptr, *dmask = nullptr, *dinj = nullptr;
    ST_TRY(dmalloc(&din, n_in)); ST_TRY(dmalloc(&dw, pk.size())); ST_TRY(dmalloc(&db, hb.size())); ST_TRY(dmalloc(&dout, n_out));
    HIP_TRY(hipMemcpy(din, hin.data(), n_in * 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(dw, pk.data(), pk.size() * 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(db, hb.data(), hb.size() * 4, hipMemcpyHostToDevice));
    // dgrad_epilogue: 1 = ReLU mask + injected diff, 2 = mask only, 3 = injected diff only, 4 = neither (a data gradient's bare store)
    const bool want_mask = dgrad_epilogue == 1 || dgrad_epilogue == 2, want_inj = dgrad_epilogue == 1 || dgrad_epilogue == 3;
    if (want_mask) ST_TRY(dmalloc(&dmask, n_out));
    if (want_inj) ST_TRY(dmalloc(&dinj, n_out));
    for (size_t off = 0; off < n_out && dgrad_epilogue; off += n_in) {      // reuse the random input as mask / inject data
        const size_t n = std::min(n_in, n_out - off);
        if (want_mask) HIP_TRY(hipMemcpy(dmask + off, din, n * 4, hipMemcpyDeviceToDevice));
        if (want_inj) HIP_TRY(hipMemcpy(dinj + off, din, n * 4, hipMemcpyDeviceToDevice));
    }
    ConvProblem p{};
    p.in = din; p.wpack = dw; p.bias = dgrad_epilogue ? nullptr : db; p.out = dout; p.mask_src = dmask; p.inject = dinj;
    p.K = K; p.M = M; p.MPad = conv_mpad(M); p.H = H; p.W = W; p.relu = dgrad_epilogue ? 0 : 1;
    if (cfg < 0) cfg = conv_pick_config(p);
    if (cfg_used) *cfg_used = cfg;
    float* dscr = nullptr;
    if (cfg == 100 && conv_wino_splits(K, M, H, W) > 1) {      // the automatic Winograd path may split K
        p.scratch_floats = (size_t)conv_wino_splits(K, M, H, W) * n_out;
        ST_TRY(dmalloc(&dscr, p.scratch_floats));
        p.scratch = dscr;
    }
    unsigned long long* dstamps = nullptr;
    const size_t max_blocks = (size_t)((W + 31) / 32) * ((H + 3) / 4) * (p.MPad / 64);
    if (cfg == 6 || cfg == 103 || cfg == 105 || cfg == 106 || cfg == 108 || cfg == 110) { HIP_TRY(hipMalloc((void**)&dstamps, max_blocks * 16)); HIP_TRY(hipMemset(dstamps, 0, max_blocks * 16)); p.stamps = dstamps; }
    hipStream_t s;
    HIP_TRY(hipStreamCreate(&s));
    auto launch = [&]() { return wino ? launch_conv3x3_wino_cfg(p, cfg - 101, s) : launch_conv3x3_cfg(p, cfg, s); };
    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreate(&e0)); HIP_TRY(hipEventCreate(&e1));
    int rc = ST_OK;
    for (int i = 0; i < 2 && rc == ST_OK; ++i) if (launch() != hipSuccess) rc = fail(ST_ERR_HIP, "conv launch failed (cfg %d)", cfg);
    if (rc == ST_OK) {
        (void)hipEventRecord(e0, s);
        for (int i = 0; i < iters; ++i) (void)launch();
        (void)hipEventRecord(e1, s);
        if (hipStreamSynchronize(s) != hipSuccess) rc = fail(ST_ERR_HIP, "conv bench sync failed");
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, e0, e1);
        *avg_ms = ms / iters;
        if (dstamps) {     // in-kernel clock and cycles of the main loop, median block
            std::vector<unsigned long long> h(max_blocks * 2);
            (void)hipMemcpy(h.data(), dstamps, max_blocks * 16, hipMemcpyDeviceToHost);
            std::vector<double> cyc, clk;
            for (size_t b = 0; b < max_blocks; ++b) if (h[2 * b + 1]) { cyc.push_back((double)h[2 * b]); clk.push_back((double)h[2 * b] / (double)h[2 * b + 1] * 0.1); }
            if (!cyc.empty()) {
                std::sort(cyc.begin(), cyc.end()); std::sort(clk.begin(), clk.end());
                fprintf(stderr, "[stamps] blocks=%zu loop cycles median=%.0f (min %.0f max %.0f) in-kernel clock median=%.3f GHz; chunks=%d -> %.0f cycles/chunk\n",
                        cyc.size(), cyc[cyc.size() / 2], cyc.front(), cyc.back(), clk[clk.size() / 2], wino ? K / 8 : (K + 3) / 4, cyc[cyc.size() / 2] / (wino ? K / 8 : (K + 3) / 4));
            }
        }
    }
    if (dstamps) (void)hipFree(dstamps);
    dfree(dscr);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); (void)hipStreamDestroy(s);
    dfree(din); dfree(dw); dfree(db); dfree(dout); dfree(dmask); dfree(dinj);
    return rc;
}

// Isolated timing of the bf16 conv launcher (conv3x3_mfma_bf16.hip) on one layer shape with the options one launch of the lean
// flow carries.  mode: sum of  1 out16, 2 fp32 out, 4 fused pool (pool16 + amap), 8 bits_out   (forward: bias + ReLU)
//                               16 data gradient: mask_bits, 32 mask16, 64 unpool (pooled diff + arg-max map), 128 fused style term
// The tile configuration is the launcher's own (ST2_CONV16_CFG / ST2_CONV16_SB_MAXK / ST2_CONV16_BIG_MIN apply).  Random data.
int st_bench_conv16(int device_id, int K, int M, int H, int W, int mode, int iters, double* avg_ms)
{
    if (K <= 0 || M <= 0 || H <= 0 || W <= 0 || iters <= 0 || !avg_ms || K % 8 || M % 8) return fail(ST_ERR_ARG, "bad argument");
    HIP_TRY(hipSetDevice(device_id));
    const size_t hw = (size_t)H * W, phw = (size_t)((H + 1) / 2) * ((W + 1) / 2);
    const bool dg = (mode & (16 | 32 | 64 | 128)) != 0;
    std::vector<float> w((size_t)M * K * 9);
    uint32_t st = 777u;
    auto rnd = [&]() { st = st * 1664525u + 1013904223u; return ((st >> 8) & 0xffff) / 32768.0f - 1.0f; };
    for (auto& x : w) x = rnd() * 0.05f;
    std::vector<unsigned short> pk(conv16_pack_elems(K, M));
    if (dg) pack_conv_weights16_dgrad(w.data(), K, M, pk.data()); else pack_conv_weights16_fwd(w.data(), M, K, pk.data());
    auto dalloc = [&](void** p, size_t bytes, int fill) -> int {
        if (hipMalloc(p, std::max<size_t>(bytes, 16)) != hipSuccess) return fail(ST_ERR_HIP, "hipMalloc(%zu)", bytes);
        // bf16 pattern 0x3c00.. (small positive numbers) / arg-max bytes 4..7 / all-ones bits
        return hipMemset(*p, fill, std::max<size_t>(bytes, 16)) == hipSuccess ? ST_OK : fail(ST_ERR_HIP, "hipMemset");
    };
    void *din = nullptr, *dw = nullptr, *dbias = nullptr, *dout16 = nullptr, *dout = nullptr, *dpool = nullptr, *damap = nullptr, *dbits = nullptr,
         *dmask16 = nullptr, *dF = nullptr, *dD = nullptr, *dA = nullptr, *dnorm = nullptr;
    const size_t in_elems = (size_t)(K / 8) * ((mode & 64) ? phw : hw) * 8;
    ST_TRY(dalloc(&din, in_elems * 2, 0x3c)); ST_TRY(dalloc(&dw, pk.size() * 2, 0)); ST_TRY(dalloc(&dbias, (size_t)conv_mpad(M) * 4, 0));
    HIP_TRY(hipMemcpy(dw, pk.data(), pk.size() * 2, hipMemcpyHostToDevice));
    ST_TRY(dalloc(&dout16, (size_t)(M / 8) * hw * 16, 0));
    Conv16Problem p{};
    p.in16 = (const unsigned short*)din; p.wpack16 = (const unsigned short*)dw; p.K = K; p.M = M; p.MPad = conv_mpad(M); p.H = H; p.W = W;
    if (mode & 1) p.out16 = (unsigned short*)dout16;
    if (mode & 2) { ST_TRY(dalloc(&dout, (size_t)M * hw * 4, 0)); p.out = (float*)dout; }
    if (!dg) {
        p.bias = (const float*)dbias; p.relu = 1;
        if (mode & 4) { ST_TRY(dalloc(&dpool, (size_t)(M / 8) * phw * 16, 0)); ST_TRY(dalloc(&damap, (size_t)(M / 8) * phw * 8, 0)); p.pool16 = (unsigned short*)dpool; p.amap = (unsigned char*)damap; }
        if (mode & 8) { ST_TRY(dalloc(&dbits, conv16_bits_elems(M, hw) * 2, 0)); p.bits_out = (unsigned short*)dbits; }
    } else {
        if (mode & 16) { ST_TRY(dalloc(&dbits, conv16_bits_elems(M, hw) * 2, 0xff)); p.mask_bits = (const unsigned short*)dbits; }
        if (mode & 32) { ST_TRY(dalloc(&dmask16, (size_t)(M / 8) * hw * 16, 0x3c)); p.mask16 = (const unsigned short*)dmask16; }
        if (mode & 64) { ST_TRY(dalloc(&damap, (size_t)(K / 8) * phw * 8, 0x05)); p.unpool_amap = (const unsigned char*)damap; }
        if (mode & 128) {
            ST_TRY(dalloc(&dF, (size_t)(M / 8) * hw * 16, 0x3c)); ST_TRY(dalloc(&dD, (size_t)M * M * 4, 0)); ST_TRY(dalloc(&dnorm, 16, 0));
            ST_TRY(dalloc(&dA, style_fuse_pack_elems(M, p.MPad) * 2, 0));
            const float one = 1.0f;
            HIP_TRY(hipMemcpy(dnorm, &one, 4, hipMemcpyHostToDevice));
            HIP_TRY(launch_style_fuse_pack((const float*)dD, M, M, p.MPad, 1.0f, 1.0f, (const float*)dnorm, (unsigned short*)dA, 0));
            p.s_in16 = (const unsigned short*)dF; p.s_wpack16 = (const unsigned short*)dA;
        }
    }
    // ST2_BENCH_STAMPS=1 (forward modes, 64x512 / SB tiles): in-kernel phase times of every workgroup
    void* dst = nullptr;
    const size_t max_wg = (size_t)((W + 31) / 32) * ((H + 3) / 4) * (p.MPad / 64);
    const bool stamps = !dg && getenv("ST2_BENCH_STAMPS") && *getenv("ST2_BENCH_STAMPS") == '1';
    if (stamps) { ST_TRY(dalloc(&dst, max_wg * 48, 0)); p.stamps = (unsigned long long*)dst; }
    hipStream_t s;
    HIP_TRY(hipStreamCreate(&s));
    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreate(&e0)); HIP_TRY(hipEventCreate(&e1));
    int rc = ST_OK;
    HIP_TRY(hipDeviceSynchronize());
    for (int i = 0; i < 2 && rc == ST_OK; ++i) { hipError_t e = launch_conv3x3_bf16(p, s); if (e != hipSuccess) rc = fail(ST_ERR_HIP, "bf16 conv launch failed: %s", hipGetErrorString(e)); }
    if (rc == ST_OK) {
        (void)hipEventRecord(e0, s);
        for (int i = 0; i < iters; ++i) (void)launch_conv3x3_bf16(p, s);
        (void)hipEventRecord(e1, s);
        if (hipStreamSynchronize(s) != hipSuccess) rc = fail(ST_ERR_HIP, "conv16 bench sync failed");
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, e0, e1);
        *avg_ms = ms / iters;
        if (stamps) {          // the last launch's stamps: medians of the three phases and of the whole lifetime, and the launch's span
            std::vector<unsigned long long> h(max_wg * 6);
            (void)hipMemcpy(h.data(), dst, max_wg * 48, hipMemcpyDeviceToHost);
            std::vector<double> pro, loop, epi, life, clk;
            unsigned long long tmin = ~0ull, tmax = 0;
            for (size_t b = 0; b < max_wg; ++b) {
                const unsigned long long* t = &h[6 * b];
                if (!t[3]) continue;
                pro.push_back((t[1] - t[0]) * 0.01); loop.push_back((t[2] - t[1]) * 0.01); epi.push_back((t[3] - t[2]) * 0.01); life.push_back((t[3] - t[0]) * 0.01);
                tmin = std::min(tmin, t[0]); tmax = std::max(tmax, t[3]);
                if (t[2] > t[1]) clk.push_back((double)(t[5] - t[4]) / (double)(t[2] - t[1]) * 0.1);      // shader cycles per 10 ns -> GHz
            }
            auto med = [](std::vector<double>& v) { std::sort(v.begin(), v.end()); return v.empty() ? 0.0 : v[v.size() / 2]; };
            auto p90 = [](std::vector<double>& v) { return v.empty() ? 0.0 : v[v.size() * 9 / 10]; };
            const double m0 = med(pro), m1 = med(loop), m2 = med(epi), m3 = med(life);
            fprintf(stderr, "[stamps] workgroups=%zu  first chunk %.2f us (p90 %.2f)  main loop %.2f us (p90 %.2f)  epilogue %.2f us (p90 %.2f)  lifetime %.2f us; launch span %.1f us; shader clock in the main loop %.3f GHz\n",
                    life.size(), m0, p90(pro), m1, p90(loop), m2, p90(epi), m3, (tmax - tmin) * 0.01, med(clk));
        }
    }
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); (void)hipStreamDestroy(s);
    for (void* q : {din, dw, dbias, dout16, dout, dpool, damap, dbits, dmask16, dF, dD, dA, dnorm, dst}) if (q) (void)hipFree(q);
    return rc;
}

}  // extern "C"
