// The device-resident style-transfer engine behind include/st2.h.
//
// One st_ctx == one reference worker's model + StyleTransfer + optimizer (worker.py:32-315,
// optimizers.py:7-125) with all tensors living in HBM.  The host only sequences launches on one
// HIP stream; nothing crosses PCIe inside an iteration unless the caller asks for the iterate.
#include "../../include/st2.h"
#include "st2_kernels.h"

#include <hip/hip_runtime.h>
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <memory>
#include <vector>

using namespace st2;

// ------------------------------------------------------------------------------------------ errors
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

extern "C" const char* st_last_error(void) { return g_err; }

// --------------------------------------------------------------------------------------- profiling
enum ProfClass { P_CONV_FWD, P_CONV_DGRAD, P_POOL_FWD, P_POOL_BWD, P_GRAM, P_GRAM_REDUCE, P_STYLE_GRAD,
                 P_LAYER_ELEM, P_IMAGE_PASS, P_FINALIZE, P_VECTOR, P_MISC, P_CONV_FWD_WINO, P_CONV_DGRAD_WINO, P_COUNT };
static const char* kProfNames[P_COUNT] = {"conv3x3_fwd_mfma_f32", "conv3x3_dgrad_mfma_f32", "maxpool_fwd", "maxpool_bwd",
                                          "gram_partial_mfma_f32", "gram_reduce", "style_grad_mfma_f32", "layer_elem",
                                          "image_pass", "finalize", "vector_ops", "misc", "conv3x3_fwd_wino_f32", "conv3x3_dgrad_wino_f32"};
struct ProfRec { int cls; hipEvent_t a, b; double flops, bytes; };

// ------------------------------------------------------------------------------------------- types
struct Layer {
    bool is_conv = false;
    std::string name;
    int cin = 0, cout = 0;
    float *w_fwd = nullptr, *w_bwd = nullptr, *w_raw = nullptr, *w_raw_r = nullptr, *bias = nullptr;   // w_raw_r: w_raw rounded to bf16 values
    unsigned short *w16_fwd = nullptr, *w16_bwd = nullptr;       // bf16 packs (bf16 feature path)
    float *u_fwd = nullptr, *u_bwd = nullptr;                    // Winograd F(2x2,3x3) packs (null: not eligible)
    bool loaded = false;
};

struct ActSet {                    // activations of one forward geometry
    int H = 0, W = 0;
    std::vector<int> C, h, w;
    std::vector<float*> data;      // data[0] is borrowed (the image itself)
    std::vector<unsigned short*> data16;   // bf16 channel-blocked copies of the blobs that feed a bf16 conv
    std::vector<unsigned char*> amap;      // lean bf16 path: arg-max maps of the pools fused into the producing conv
    std::vector<char> has32, amap_ok;      // per blob: fp32 copy / arg-max map written by the last forward
    int valid_to = -1;
};

struct ActiveLayer { int blob; float cw, sw, dw; bool c, s, d; };

static const struct { int kind; const char* name; int cin, cout; } kVgg19[] = {
    {0, "conv1_1", 3, 64}, {0, "conv1_2", 64, 64}, {1, "pool1", 0, 0},
    {0, "conv2_1", 64, 128}, {0, "conv2_2", 128, 128}, {1, "pool2", 0, 0},
    {0, "conv3_1", 128, 256}, {0, "conv3_2", 256, 256}, {0, "conv3_3", 256, 256}, {0, "conv3_4", 256, 256}, {1, "pool3", 0, 0},
    {0, "conv4_1", 256, 512}, {0, "conv4_2", 512, 512}, {0, "conv4_3", 512, 512}, {0, "conv4_4", 512, 512}, {1, "pool4", 0, 0},
    {0, "conv5_1", 512, 512}, {0, "conv5_2", 512, 512}, {0, "conv5_3", 512, 512}, {0, "conv5_4", 512, 512}, {1, "pool5", 0, 0},
};

static bool nonzero(float w) { return fabsf(w) > 1e-15f; }   // NaN compares false: worker.py:234

struct st_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool bf16 = false;                             // conv operands in bf16 (BASELINE config 3)
    bool lean = false;                             // bf16 objective evaluations skip the fp32 tensors only bf16 convs would read
    bool wino = true;                              // Winograd F(2x2,3x3) for the eligible fp32 convs (ST2_WINO=0 disables)
    unsigned short *diff16A = nullptr, *diff16B = nullptr;
    std::vector<Layer> topo;
    std::vector<std::string> blob_names;
    int nb = 0;                                    // number of blobs (= layers + 1)

    ActSet act;                                    // geometry of input/content
    // image state
    int H = 0, W = 0;                              // input geometry (0 = no input)
    float* x[2] = {nullptr, nullptr};
    int cur = 0;
    float* fwd_x = nullptr; size_t fwd_x_cap = 0;   // image of the st_forward test hook (never the job's iterate)
    float* grad = nullptr;                         // combined gradient (opfunc / L-BFGS)
    // content / style
    int cH = 0, cW = 0;
    std::vector<float*> content_feat;              // per blob
    float* content_x = nullptr;                    // preprocessed content image (for resample_content)
    std::vector<float*> style_gram;                // per blob, C*C
    bool have_content = false, have_style = false;
    // objective
    std::vector<ActiveLayer> rows;                 // every row of the weights table, in order
    std::vector<ActiveLayer> active;               // rows with any non-zero weight
    float tv_w = 1, tv_pow = 1, p_w = 1, p_pow = 1;   // worker.py:133 defaults
    float* norms = nullptr;                        // [nb][3] on device
    std::vector<char> norm_valid;                  // [nb*3]
    // work buffers (input geometry)
    std::vector<float*> inject;
    float *diffA = nullptr, *diffB = nullptr, *stmp = nullptr;
    size_t max_blob = 0;
    float *gram_slabs = nullptr, *gram_fold = nullptr, *dbuf = nullptr;
    unsigned short* d16 = nullptr; size_t d16_cap = 0;            // bf16 path: hi/lo operand image of D (style16.hip)
    // bf16 path, style term fused into the data-gradient conv above the style blob: per blob the scaled hi/lo image of D (kept until
    // that conv has run) and, during one objective evaluation, the operands handed to backward_chain
    std::vector<unsigned short*> sfuse_w; std::vector<size_t> sfuse_cap;
    std::vector<const unsigned short*> sf_in, sf_w;
    float* conv_scratch = nullptr; size_t conv_scratch_cap = 0;       // split-K partial sums of Winograd launches
    // hipGraph replay of the steady-state Adam step (launch-bound regime: small images)
    unsigned long long epoch = 0;                  // bumped by every API call that can change what a step launches
    hipGraphExec_t gexec[2] = {nullptr, nullptr};  // one per parity of the x ping-pong
    unsigned long long gepoch[2] = {0, 0};
    float* adam_dyn = nullptr;                     // device {corr1, corr2, step}: the only per-step arguments
    bool capturing = false, graphs = false;        // opt-in (ST2_GRAPH=1): measured, no gain -- see step_graph_ok()
    int plain_steps = 0;                           // normal steps since the last epoch change (buffers are allocated lazily)
    unsigned long long plain_epoch = ~0ull;
    size_t graph_max_px = 768 * 768;
    long long graph_replays = 0;
    size_t gram_slab_cap = 0, gram_fold_cap = 0;
    std::vector<float*> layer_part;                // per blob: 5 * kMaxPartials
    std::vector<float*> s2_part;                   // per blob: style-grad partial sums
    std::vector<int> s2_cap;
    std::vector<int> cnt;                          // per blob * 6 partial counts
    float* image_part = nullptr;                   // 6 * kMaxPartials
    int image_cnt = 0;
    float* trace_dev = nullptr;
    double* trace_sums = nullptr;                  // device scratch of the trace finalisation
    float* trace_host = nullptr;                   // pinned
    int trace_len_last = 8;
    float* hwc_dev = nullptr;
    // pipelined iterations (st_step_begin / st_step_end): up to two in flight; the iterate of step k travels to pinned host memory
    // on its own stream while step k + 1 computes
    struct Pipe {
        // kSlots buffers although only two iterations are ever in flight: an iterate handed out by st_step_end stays valid for
        // kSlots - 1 further begins, which is what lets the worker's sender thread pickle it without a host-side copy
        static constexpr int kSlots = 6;
        hipStream_t copy = nullptr;
        float* hwc[kSlots] = {}; float* img_pin[kSlots] = {}; float* trace_pin[kSlots] = {};
        hipEvent_t ready[kSlots] = {}, done[kSlots] = {};
        size_t cap = 0; long long head = 0; int count = 0, tlen[kSlots] = {}, H[kSlots] = {}, W[kSlots] = {};
    } pipe;
    void* stage_dev = nullptr; size_t stage_cap = 0;
    // optimizer
    int opt_kind = ST_OPT_NONE;
    double step_size = 1.0;
    float *m = nullptr, *v = nullptr;
    int items1 = 0, items2 = 0;
    bool m_zero = true, v_zero = true;
    // L-BFGS
    static const int kCorr = kLbfgsCorr;
    float* hs[kLbfgsSlots] = {nullptr};            // ring of s vectors (10 pairs + the one being formed)
    float* hy[kLbfgsSlots] = {nullptr};
    LbfgsDev* lb_dev = nullptr;                    // history bookkeeping (pair count, ring order, s.y, y.y): device-resident
    bool lb_clear = true;                          // history to be emptied before the next step (reset / objective_changed)
    float* lb_part = nullptr;                      // [4][kMaxPartials] partial sums of the chained dot products
    float* g_cur = nullptr; float* pvec = nullptr;
    bool have_cur = false;
    float last_loss = 0.f;
    // tile-sharded mode (BASELINE config 5): this context holds ONE window of a larger image
    struct Tile {
        bool on = false;
        int gH = 0, gW = 0, wy0 = 0, wx0 = 0, ty0 = 0, tx0 = 0, ty1 = 0, tx1 = 0;
        float *p1 = nullptr, *p2 = nullptr, *p3 = nullptr, *pd = nullptr;    // reduce buffers (device)
        size_t p1_n = 0, p2_n = 0, p3_n = 0, pd_n = 0;
        bool s2_in_p2 = false;
        float* wgrad = nullptr;                    // window gradient (3, wh, ww)
    } tile;
    // profiling
    bool prof_on = false;
    std::vector<ProfRec> prof;
    std::vector<hipEvent_t> ev_pool;
    size_t ev_used = 0;

    int blob_c(int i) const { return act.C[i]; }
};

// ---------------------------------------------------------------------------------------- helpers
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
    if (p && hipFree(p) != hipSuccess) (void)hipGetLastError();   // never leave a sticky error behind
    p = nullptr;
}

// room for the split-K partial sums of a Winograd launch that would otherwise leave most CUs idle
static int wino_scratch(st_ctx* c, ConvProblem& p);

static int dmalloc16(unsigned short** p, size_t n)
{
    void* q = nullptr;
    hipError_t e = hipMalloc(&q, std::max<size_t>(n, 8) * sizeof(unsigned short));
    if (e != hipSuccess) return fail(ST_ERR_HIP, "hipMalloc(%zu bf16): %s", n, hipGetErrorString(e));
    *p = (unsigned short*)q;
    return ST_OK;
}
static void dfree16(unsigned short*& p)
{
    if (p && hipFree(p) != hipSuccess) (void)hipGetLastError();
    p = nullptr;
}
static size_t act16_elems(int C, size_t hw) { return (size_t)((C + 7) / 8) * hw * 8; }
static bool conv16_ok(const st_ctx* c, int K) ;

struct ProfScope {
    st_ctx* c; int idx = -1;
    ProfScope(st_ctx* ctx, int cls, double flops, double bytes) : c(ctx)
    {
        if (!c->prof_on) return;
        auto get = [&]() {
            if (c->ev_used == c->ev_pool.size()) {
                hipEvent_t e;
                (void)hipEventCreate(&e);
                c->ev_pool.push_back(e);
            }
            return c->ev_pool[c->ev_used++];
        };
        ProfRec r{cls, get(), get(), flops, bytes};
        (void)hipEventRecord(r.a, c->stream);
        c->prof.push_back(r);
        idx = (int)c->prof.size() - 1;
    }
    ~ProfScope()
    {
        if (idx >= 0) (void)hipEventRecord(c->prof[idx].b, c->stream);
    }
};

static bool conv16_ok(const st_ctx* c, int K) { (void)c; return K >= 8 && K % 8 == 0; }

static int wino_scratch(st_ctx* c, ConvProblem& p)
{
    const int sp = conv_wino_splits(p.K, p.M, p.H, p.W);
    if (sp <= 1) return ST_OK;
    const size_t need = (size_t)sp * p.M * p.H * p.W;
    if (need > c->conv_scratch_cap) {
        HIP_TRY(hipStreamSynchronize(c->stream));
        dfree(c->conv_scratch); c->conv_scratch_cap = 0;
        ST_TRY(dmalloc(&c->conv_scratch, need));
        c->conv_scratch_cap = need;
    }
    p.scratch = c->conv_scratch; p.scratch_floats = c->conv_scratch_cap;
    return ST_OK;
}

static void shapes_for(const st_ctx* c, int H, int W, std::vector<int>& C, std::vector<int>& h, std::vector<int>& w)
{
    C.assign(c->nb, 0); h.assign(c->nb, 0); w.assign(c->nb, 0);
    C[0] = 3; h[0] = H; w[0] = W;
    for (int i = 1; i < c->nb; ++i) {
        const Layer& L = c->topo[i - 1];
        if (L.is_conv) { C[i] = L.cout; h[i] = h[i - 1]; w[i] = w[i - 1]; }
        else { C[i] = C[i - 1]; h[i] = pooled_size(h[i - 1]); w[i] = pooled_size(w[i - 1]); }
    }
}

static void act_free(ActSet& a)
{
    for (size_t i = 1; i < a.data.size(); ++i) dfree(a.data[i]);
    for (size_t i = 0; i < a.data16.size(); ++i) dfree16(a.data16[i]);
    for (size_t i = 0; i < a.amap.size(); ++i) if (a.amap[i]) { (void)hipFree(a.amap[i]); a.amap[i] = nullptr; }
    a.data.clear();
    a.data16.clear();
    a.amap.clear();
    a.H = a.W = 0;
    a.valid_to = -1;
}

static int act_ensure(st_ctx* c, ActSet& a, int H, int W)
{
    if (a.H == H && a.W == W && !a.data.empty()) return ST_OK;
    act_free(a);
    shapes_for(c, H, W, a.C, a.h, a.w);
    a.data.assign(c->nb, nullptr);
    a.data16.assign(c->nb, nullptr);
    a.amap.assign(c->nb, nullptr);
    a.has32.assign(c->nb, 0);
    a.amap_ok.assign(c->nb, 0);
    for (int i = 1; i < c->nb; ++i) ST_TRY(dmalloc(&a.data[i], (size_t)a.C[i] * a.h[i] * a.w[i]));
    a.H = H; a.W = W;
    return ST_OK;
}

static bool blob_active(const st_ctx* c, int b)
{
    for (const ActiveLayer& al : c->active) if (al.blob == b) return true;
    return false;
}

// Will the style term of blob b (a style layer) run entirely on the blob's bf16 copy (gram16.hip + style16.hip)?  Decided from
// shapes only, so that the forward (which may then skip the fp32 blob) and the objective agree.
static bool style_runs16(const st_ctx* c, const ActSet& a, int b)
{
    if (!c->bf16 || c->tile.on || b < 1 || !c->topo[b - 1].is_conv) return false;
    const int C = a.C[b], hw = a.h[b] * a.w[b];
    return conv16_ok(c, C) && style_grad16_ok(C, (size_t)hw) && C % 8 == 0 && hw % 64 == 0 && gram16_ok(C, hw, gram_plan16(C, hw));
}

// lean evaluation: does anything read blob b in fp32?  Content / deep-dream terms do (layer_elem_k); a style term only when
// its Gram / gradient cannot run on the bf16 copy.
static bool blob_needs32(const st_ctx* c, const ActSet& a, int b)
{
    for (const ActiveLayer& al : c->active)
        if (al.blob == b && (al.c || al.d || (al.s && !style_runs16(c, a, b)))) return true;
    return false;
}

// May the style gradient of blob b ride on the data-gradient conv of the layer above it (conv3x3_mfma_bf16.hip, fused style term)?
static bool style_fuse_ok(const st_ctx* c, const ActSet& a, int b, int last)
{
    const char* e = getenv("ST2_STYLE_FUSE");           // read per evaluation: the tests compare both flows in one process
    if ((e && *e == '0') || !style_runs16(c, a, b) || b + 1 > last || a.C[b] % 32 != 0) return false;
    const Layer& up = c->topo[b];                       // layer b + 1: consumes blob b
    return up.is_conv && up.loaded && conv16_ok(c, up.cout) && up.cin == a.C[b];
}

// `lean` (bf16 objective evaluations only): a conv blob whose only consumers are bf16 convs / a fused pool is not written
// in fp32 at all, and a pool that follows such a conv is computed in that conv's epilogue (bf16 pooled copy + arg-max map).
static int forward_range(st_ctx* c, ActSet& a, const float* x, int last, bool lean = false)
{
    a.data[0] = const_cast<float*>(x);
    a.has32.assign(c->nb, 0); a.amap_ok.assign(c->nb, 0);
    a.has32[0] = 1;
    int pooled_by_conv = -1;
    for (int i = 1; i <= last; ++i) {
        const Layer& L = c->topo[i - 1];
        if (L.is_conv) {
            if (!L.loaded) return fail(ST_ERR_STATE, "weights of %s were never loaded", L.name.c_str());
            const double px = (double)a.h[i] * a.w[i];
            // does the layer that consumes blob i run on the bf16 matrix cores?
            bool next16 = c->bf16 && i < last && c->topo[i].is_conv && conv16_ok(c, L.cout);
            const bool conv_next16 = next16;
            // ... or does the style gradient of this blob (bf16 path: style16.hip reads the bf16 copy)?
            if (c->bf16 && !c->tile.on && conv16_ok(c, L.cout) && style_grad16_ok(L.cout, (size_t)a.h[i] * a.w[i]))
                for (const ActiveLayer& al : c->active) if (al.blob == i && al.s) next16 = true;
            if (next16 && !a.data16[i]) ST_TRY(dmalloc16(&a.data16[i], act16_elems(a.C[i], (size_t)a.h[i] * a.w[i])));
            a.has32[i] = 1;
            if (c->bf16 && conv16_ok(c, L.cin) && a.data16[i - 1]) {
                Conv16Problem p{};
                p.in16 = a.data16[i - 1]; p.wpack16 = L.w16_fwd; p.bias = L.bias; p.out = a.data[i];
                p.out16 = next16 ? a.data16[i] : nullptr;
                p.K = L.cin; p.M = L.cout; p.MPad = conv_mpad(L.cout); p.H = a.h[i]; p.W = a.w[i]; p.relu = 1;
                double bytes = px * (2.0 * L.cin + 4.0 * L.cout + (next16 ? 2.0 * L.cout : 0.0));
                if (lean && !blob_needs32(c, a, i) && i < last) {
                    const bool next_pool = !c->topo[i].is_conv;
                    if (next_pool && !blob_active(c, i) && conv16_can_pool(p)) {       // (a weighted blob gets an injected diff: classic pool backward)
                        // the pool rides on this launch: pooled bf16 copy for the conv after it, arg-max map for the backward
                        const int pb = i + 1, pc = a.C[pb];
                        const size_t phw = (size_t)a.h[pb] * a.w[pb];
                        const bool pool_feeds16 = pb < last && c->topo[pb].is_conv && conv16_ok(c, pc);
                        if (pool_feeds16 && !a.data16[pb]) ST_TRY(dmalloc16(&a.data16[pb], act16_elems(pc, phw)));
                        if (!a.amap[pb]) HIP_TRY(hipMalloc((void**)&a.amap[pb], act16_elems(pc, phw)));
                        p.pool16 = pool_feeds16 ? a.data16[pb] : nullptr;
                        p.pool32 = (!pool_feeds16 || blob_active(c, pb) || pb == last) ? a.data[pb] : nullptr;
                        p.amap = a.amap[pb];
                        p.out = nullptr;
                        a.has32[i] = 0; a.has32[pb] = p.pool32 != nullptr; a.amap_ok[pb] = 1;
                        pooled_by_conv = pb;
                        bytes = px * (2.0 * L.cin + 0.25 * L.cout * (1.0 + (pool_feeds16 ? 2.0 : 0.0) + (p.pool32 ? 4.0 : 0.0)));
                    } else if (conv_next16) {
                        p.out = nullptr;                   // the next conv reads the bf16 copy; the backward masks with it too
                        a.has32[i] = 0;
                        bytes = px * (2.0 * L.cin + 2.0 * L.cout);
                    }
                }
                ProfScope ps(c, P_CONV_FWD, 2.0 * 9 * L.cin * L.cout * px, bytes);
                HIP_TRY(launch_conv3x3_bf16(p, c->stream));
            } else {
                ConvProblem p{};
                bool packed = false;
                p.in = a.data[i - 1]; p.wpack = L.w_fwd; p.bias = L.bias; p.out = a.data[i];
                p.K = L.cin; p.M = L.cout; p.MPad = conv_mpad(L.cout); p.H = a.h[i]; p.W = a.w[i]; p.relu = 1;
                { const bool wino = c->wino && L.u_fwd && conv_wino_ok(p.K, p.M, p.H, p.W);
                  // flops are the ALGORITHMIC (direct-convolution) count in both classes; Winograd executes 4/9 of them
                  ProfScope ps(c, wino ? P_CONV_FWD_WINO : P_CONV_FWD, 2.0 * 9 * L.cin * L.cout * px, 4.0 * px * (L.cin + L.cout));
                  if (wino) {
                      p.wpack = L.u_fwd; ST_TRY(wino_scratch(c, p));
                      // the max-pool that follows rides on this launch's epilogue (the pooled blob is written beside the conv blob)
                      if (i < last && !c->topo[i].is_conv && !c->bf16 && conv_wino_can_pool(p.K, p.M, p.H, p.W)) {
                          p.pool_out = a.data[i + 1]; pooled_by_conv = i + 1; a.has32[i + 1] = 1;
                          // ... and a one-byte arg-max map for the pool's backward (maxpool_bwd_amap_k: neither blob is read again)
                          const char* ae = getenv("ST2_POOL_AMAP");          // =0: the classic pool backward (read per forward: the tests compare both)
                          if (!(ae && *ae == '0') && conv_wino_pool_amap_ok(p.K, p.M, p.H, p.W)) {
                              // (sized like the bf16 path's map of the same blob: the buffer is shared when the precision is switched)
                              const size_t pn = act16_elems(a.C[i + 1], (size_t)a.h[i + 1] * a.w[i + 1]);
                              if (!a.amap[i + 1]) HIP_TRY(hipMalloc((void**)&a.amap[i + 1], pn));
                              p.pool_amap = a.amap[i + 1]; a.amap_ok[i + 1] = 2;      // 2: the fp32 layout [C][ph][pw]
                          }
                      }
                      HIP_TRY(launch_conv3x3_wino(p, c->stream));
                  }
                  else { if (next16 && L.cout % 8 == 0) { p.out16 = a.data16[i]; packed = true; }      // the epilogue writes the bf16 copy too
                         // lean: conv1_1's fp32 blob is written only if something reads it (conv1_2, the ReLU mask and a style term take the copy)
                         if (lean && packed && conv_next16 && i < last && !blob_needs32(c, a, i)) { p.out = nullptr; a.has32[i] = 0; }
                         HIP_TRY(launch_conv3x3(p, c->stream)); } }
                if (next16 && !packed) { ProfScope ps(c, P_MISC, 0, px * 6.0 * L.cout); HIP_TRY(launch_pack_act16(a.data[i], a.data16[i], a.C[i], (size_t)a.h[i] * a.w[i], c->stream)); }
            }
        } else if (i == pooled_by_conv) {
            // written by the producing conv's epilogue
        } else {
            const double n_in = (double)a.C[i - 1] * a.h[i - 1] * a.w[i - 1];
            a.has32[i] = 1;
            { ProfScope ps(c, P_POOL_FWD, 0, 4.0 * n_in * 1.25);
              HIP_TRY(launch_maxpool_fwd(a.data[i - 1], a.data[i], a.C[i - 1], a.h[i - 1], a.w[i - 1], c->stream)); }
            if (c->bf16 && i < last && c->topo[i].is_conv && conv16_ok(c, a.C[i])) {
                const size_t hw = (size_t)a.h[i] * a.w[i];
                if (!a.data16[i]) ST_TRY(dmalloc16(&a.data16[i], act16_elems(a.C[i], hw)));
                ProfScope ps(c, P_MISC, 0, hw * 6.0 * a.C[i]);
                HIP_TRY(launch_pack_act16(a.data[i], a.data16[i], a.C[i], hw, c->stream));
            }
        }
    }
    a.valid_to = last;
    return ST_OK;
}

static int ensure_gram_bufs(st_ctx* c, int C, int hw, GramPlan& pl, bool plan16 = false)
{
    pl = plan16 ? gram_plan16(C, hw) : gram_plan(C, hw);
    if (pl.slab_floats > c->gram_slab_cap) {
        dfree(c->gram_slabs);
        ST_TRY(dmalloc(&c->gram_slabs, pl.slab_floats));
        c->gram_slab_cap = pl.slab_floats;
    }
    const size_t fold = (size_t)gram_fold_groups(pl) * C * C;
    if (fold > c->gram_fold_cap) {
        dfree(c->gram_fold);
        ST_TRY(dmalloc(&c->gram_fold, fold));
        c->gram_fold_cap = fold;
    }
    return ST_OK;
}

// G (or G - target) of blob data F -> out (C*C); optional sum-of-squares partials
// F16 (optional): the bf16 channel-blocked copy of the blob -- the bf16 feature path then takes the partials on the bf16 matrix cores
static int gram_into(st_ctx* c, const float* F, int C, int hw, const float* target, float* out, int out_ld, float* partial, int* n_partial,
                     const unsigned short* F16 = nullptr)
{
    GramPlan pl;
    const bool use16 = F16 && C % 8 == 0 && hw % 64 == 0 && gram16_ok(C, hw, gram_plan16(C, hw));
    ST_TRY(ensure_gram_bufs(c, C, hw, pl, use16));
    {
        ProfScope ps(c, P_GRAM, 2.0 * C * C * (double)hw, (use16 ? 2.0 : 4.0) * C * (double)hw);
        if (use16) HIP_TRY(launch_gram16_partial(F16, c->gram_slabs, C, hw, pl, c->stream));
        else HIP_TRY(launch_gram_partial(F, c->gram_slabs, C, hw, pl, c->stream));
    }
    {
        ProfScope ps(c, P_GRAM_REDUCE, 0, 4.0 * (double)pl.slab_floats);
        HIP_TRY(launch_gram_reduce(c->gram_slabs, c->gram_fold, target, out, out_ld, partial, n_partial, C, (double)C * hw, pl, c->stream));
    }
    return ST_OK;
}

// backward chain from blob `top` whose diff is `cur` down to data; returns pointer in *out.
// `lean` must be what the forward that filled c->act ran with: the fp32 diff of a layer is then written only when its
// consumer needs fp32 (a pool without arg-max map, the 3-channel conv1_1 kernel, a non-bf16 conv), ReLU masks come from the
// bf16 copies, and pools fused into their producing conv are back-propagated through their arg-max maps in bf16.
static int backward_chain(st_ctx* c, int top, const float* top_diff, const std::vector<const float*>& inj, const float** out, bool lean = false)
{
    const ActSet& a = c->act;
    const float* cur = top_diff;
    const unsigned short* cur16 = nullptr;             // bf16 copy of the running diff, when a producer already made it
    if (c->bf16 && !c->diff16A) {
        const size_t cap = c->max_blob + 8 * (size_t)a.h[0] * a.w[0];
        ST_TRY(dmalloc16(&c->diff16A, cap)); ST_TRY(dmalloc16(&c->diff16B, cap));
    }
    auto conv_takes16 = [&](int layer_index) {         // does conv layer `layer_index` (1-based blob index) read its diff as bf16?
        const Layer& P = c->topo[layer_index - 1];
        if (!P.is_conv || !conv16_ok(c, P.cout)) return false;
        const bool first_small = conv_dgrad_smallM_ok(P.cout, P.cin) && !(layer_index - 1 >= 1 && c->topo[layer_index - 2].is_conv);
        return !conv_dgrad_smallM_ok(P.cout, P.cin) || (first_small && P.w_raw_r != nullptr);
    };
    for (int i = top; i >= 1; --i) {
        const Layer& L = c->topo[i - 1];
        const int below = i - 1;
        float* dst = (cur == c->diffA) ? c->diffB : c->diffA;
        unsigned short* dst16 = (cur16 == c->diff16A) ? c->diff16B : c->diff16A;
        const bool below_is_conv = below >= 1 && c->topo[below - 1].is_conv;
        const float* mask_src = below_is_conv ? a.data[below] : nullptr;
        const float* inject = inj[below];
        if (L.is_conv) {
            const double px = (double)a.h[i] * a.w[i];
            const bool small_m = !mask_src && conv_dgrad_smallM_ok(L.cout, L.cin);
            const bool wino_bwd = !small_m && !(c->bf16 && conv16_ok(c, L.cout)) && c->wino && L.u_bwd && conv_wino_ok(L.cout, L.cin, a.h[i], a.w[i]);
            if (small_m && c->bf16 && L.w_raw_r) {
                // bf16 feature path: this conv's operands are bf16 too -- the diff arrives as (or is packed into) a bf16 copy
                const size_t hw = (size_t)a.h[i] * a.w[i];
                if (!cur16) {
                    if (!cur) return fail(ST_ERR_STATE, "internal: no diff above %s", L.name.c_str());
                    ProfScope ps(c, P_MISC, 0, hw * 6.0 * L.cout);
                    HIP_TRY(launch_pack_act16(cur, dst16, L.cout, hw, c->stream));
                    cur16 = dst16;
                }
                ProfScope ps(c, P_CONV_DGRAD, 2.0 * 9 * L.cin * L.cout * px, px * (2.0 * L.cout + 4.0 * L.cin));
                HIP_TRY(launch_conv3x3_dgrad_smallM16(cur16, L.w_raw_r, dst, inject, L.cout, L.cin, a.h[i], a.w[i], c->stream));
                cur16 = nullptr; cur = dst;
            } else if (small_m) {
                if (!cur) return fail(ST_ERR_STATE, "internal: fp32 diff missing above %s", L.name.c_str());
                ProfScope ps(c, P_CONV_DGRAD, 2.0 * 9 * L.cin * L.cout * px, 4.0 * px * (L.cin + L.cout));
                HIP_TRY(launch_conv3x3_dgrad_smallM(cur, L.w_raw, dst, inject, L.cout, L.cin, a.h[i], a.w[i], c->stream));
                cur16 = nullptr; cur = dst;
            } else if (c->bf16 && conv16_ok(c, L.cout)) {
                const size_t hw = (size_t)a.h[i] * a.w[i];
                if (!cur16) {                                      // top diff / classic pool-backward output: make the bf16 copy
                    if (!cur) return fail(ST_ERR_STATE, "internal: no diff above %s", L.name.c_str());
                    unsigned short* tmp16 = dst16;
                    ProfScope ps(c, P_MISC, 0, hw * 6.0 * L.cout);
                    HIP_TRY(launch_pack_act16(cur, tmp16, L.cout, hw, c->stream));
                    cur16 = tmp16;
                    dst16 = (cur16 == c->diff16A) ? c->diff16B : c->diff16A;
                }
                // the consumer of this launch's output takes bf16 iff it is a bf16 dgrad conv, or (lean) a pool with an arg-max map
                bool below16 = below >= 1 && conv_takes16(below);
                if (lean && below >= 1 && !c->topo[below - 1].is_conv && a.amap_ok[below] && L.cin % 8 == 0) below16 = true;
                Conv16Problem p{};
                p.in16 = cur16; p.wpack16 = L.w16_bwd; p.bias = nullptr; p.out = dst; p.out16 = below16 ? dst16 : nullptr;
                p.mask_src = mask_src; p.inject = inject;
                if (lean && mask_src && a.data16[below]) { p.mask16 = a.data16[below]; p.mask_src = nullptr; }
                const bool fused = below >= 1 && (size_t)below < c->sf_w.size() && c->sf_w[below] != nullptr;
                if (fused) {                // the style gradient of blob `below` rides on this launch: out = mask(conv) + D' @ F (+ inject)
                    p.s_in16 = c->sf_in[below]; p.s_wpack16 = c->sf_w[below];
                    if (mask_src) { p.mask16 = a.data16[below]; p.mask_src = nullptr; }      // the mask is applied in registers, from the bf16 copy
                }
                if (p.mask_src && !a.has32[below]) return fail(ST_ERR_STATE, "internal: mask blob %d missing", below);
                if (lean && below16) p.out = nullptr;
                p.K = L.cout; p.M = L.cin; p.MPad = conv_mpad(L.cin); p.H = a.h[i]; p.W = a.w[i]; p.relu = 0;
                ProfScope ps(c, P_CONV_DGRAD, 2.0 * 9 * L.cin * L.cout * px + (fused ? 2.0 * L.cin * L.cin * px : 0.0),
                             px * (2.0 * L.cout + (fused ? 2.0 : 0.0) * L.cin + (p.out ? 4.0 : 0.0) * L.cin + (p.out16 ? 2.0 : 0.0) * L.cin + (mask_src ? (p.mask16 ? 2.0 : 4.0) : 0.0) * L.cin));
                HIP_TRY(launch_conv3x3_bf16(p, c->stream));
                cur16 = below16 ? dst16 : nullptr;
                cur = p.out ? dst : nullptr;
            } else {
                if (!cur) return fail(ST_ERR_STATE, "internal: fp32 diff missing above %s", L.name.c_str());
                cur16 = nullptr;
                ConvProblem p{};
                p.in = cur; p.wpack = L.w_bwd; p.bias = nullptr; p.out = dst;
                p.mask_src = mask_src; p.inject = inject;
                p.K = L.cout; p.M = L.cin; p.MPad = conv_mpad(L.cin); p.H = a.h[i]; p.W = a.w[i]; p.relu = 0;
                ProfScope ps(c, wino_bwd ? P_CONV_DGRAD_WINO : P_CONV_DGRAD, 2.0 * 9 * L.cin * L.cout * px, 4.0 * px * (L.cin + L.cout));
                if (wino_bwd) { p.wpack = L.u_bwd; ST_TRY(wino_scratch(c, p)); HIP_TRY(launch_conv3x3_wino(p, c->stream)); }
                else HIP_TRY(launch_conv3x3(p, c->stream));
                cur = dst;
            }
        } else if (lean && a.amap_ok[i] && !inject && below >= 1 && conv_takes16(below)) {
            // pool fused into its producing conv: route the bf16 diff through the arg-max map (ReLU mask of the conv blob included)
            const int C = a.C[below];
            const size_t hw_top = (size_t)a.h[i] * a.w[i], hw = (size_t)a.h[below] * a.w[below];
            if (!cur16) {
                if (!cur) return fail(ST_ERR_STATE, "internal: no diff above %s", L.name.c_str());
                unsigned short* tmp16 = dst16;
                ProfScope ps(c, P_MISC, 0, hw_top * 6.0 * C);
                HIP_TRY(launch_pack_act16(cur, tmp16, C, hw_top, c->stream));
                cur16 = tmp16;
                dst16 = (cur16 == c->diff16A) ? c->diff16B : c->diff16A;
            }
            ProfScope ps(c, P_POOL_BWD, 0, (double)C * (hw_top * 3.0 + hw * 2.0));
            HIP_TRY(launch_maxpool_bwd_idx16(cur16, a.amap[i], dst16, C, a.h[below], a.w[below], c->stream));
            cur16 = dst16; cur = nullptr;
        } else if (a.amap_ok[i] == 2 && !inject && mask_src && cur) {
            // pool fused into its producing Winograd conv (fp32): route the diff through the arg-max map, ReLU mask included
            ProfScope ps(c, P_POOL_BWD, 0, (double)a.C[below] * ((double)a.h[i] * a.w[i] * 5.0 + (double)a.h[below] * a.w[below] * 4.0));
            HIP_TRY(launch_maxpool_bwd_amap(cur, a.amap[i], dst, a.C[below], a.h[below], a.w[below], c->stream));
            cur16 = nullptr; cur = dst;
        } else {
            if (!cur) return fail(ST_ERR_STATE, "internal: fp32 diff missing above %s", L.name.c_str());
            if (!a.has32[below]) return fail(ST_ERR_STATE, "internal: pool input blob %d missing", below);
            const double n_in = (double)a.C[below] * a.h[below] * a.w[below];
            ProfScope ps(c, P_POOL_BWD, 0, 4.0 * n_in * 2.25);
            HIP_TRY(launch_maxpool_bwd(cur, a.data[below], dst, inject, mask_src != nullptr, a.C[below], a.h[below], a.w[below], c->stream));
            cur16 = nullptr; cur = dst;
        }
    }
    if (!cur) return fail(ST_ERR_STATE, "internal: the backward chain ended without an fp32 image gradient");
    *out = cur;
    return ST_OK;
}

static int ensure_input_buffers(st_ctx* c, int H, int W)
{
    if (c->H == H && c->W == W && c->x[0]) return ST_OK;
    const size_t n3 = (size_t)3 * H * W;
    for (int i = 0; i < 2; ++i) { dfree(c->x[i]); ST_TRY(dmalloc(&c->x[i], n3)); }
    dfree(c->grad); ST_TRY(dmalloc(&c->grad, n3));
    dfree(c->m); dfree(c->v);
    ST_TRY(dmalloc(&c->m, n3)); ST_TRY(dmalloc(&c->v, n3));
    dfree(c->g_cur); dfree(c->pvec);
    for (int i = 0; i <= st_ctx::kCorr; ++i) { dfree(c->hs[i]); dfree(c->hy[i]); }
    dfree(c->hwc_dev); ST_TRY(dmalloc(&c->hwc_dev, n3));
    c->H = H; c->W = W; c->cur = 0;
    // work buffers that follow the input geometry
    for (auto& p : c->inject) dfree(p);
    dfree(c->diffA); dfree(c->diffB); dfree(c->stmp); dfree16(c->diff16A); dfree16(c->diff16B);
    std::vector<int> C, h, w;
    shapes_for(c, H, W, C, h, w);
    c->max_blob = 0;
    for (int i = 0; i < c->nb; ++i) c->max_blob = std::max(c->max_blob, (size_t)C[i] * h[i] * w[i]);
    return ST_OK;
}

static int stage_upload(st_ctx* c, const void* host, size_t bytes)
{
    if (bytes > c->stage_cap) {
        if (c->stage_dev) (void)hipFree(c->stage_dev);
        c->stage_dev = nullptr;
        HIP_TRY(hipMalloc(&c->stage_dev, bytes));
        c->stage_cap = bytes;
    }
    HIP_TRY(hipMemcpyAsync(c->stage_dev, host, bytes, hipMemcpyHostToDevice, c->stream));
    return ST_OK;
}

static int preprocess_into(st_ctx* c, const void* hwc, int H, int W, int is_u8, float* dst)
{
    if (!hwc || H <= 0 || W <= 0) return fail(ST_ERR_ARG, "bad image (%p, %d x %d)", hwc, H, W);
    const size_t n = (size_t)H * W * 3;
    ST_TRY(stage_upload(c, hwc, n * (is_u8 ? 1 : 4)));
    ProfScope ps(c, P_MISC, 0, 0);
    if (is_u8) HIP_TRY(launch_preprocess_u8((const uint8_t*)c->stage_dev, dst, H, W, c->stream));
    else HIP_TRY(launch_preprocess_f32((const float*)c->stage_dev, dst, H, W, c->stream));
    return ST_OK;
}

// an input of a new geometry: every size-dependent optimizer tensor starts from zero
static int set_input_common(st_ctx* c, int H, int W)
{
    const bool reshaped = !(c->H == H && c->W == W && c->x[0]);
    ST_TRY(ensure_input_buffers(c, H, W));
    if (reshaped) {            // every size-dependent optimizer tensor starts from zero
        c->m_zero = c->v_zero = true;
        c->lb_clear = true;
        c->have_cur = false;
    }
    return ST_OK;
}

// ------------------------------------------------------------------------------------ the objective
static int eval_objective(st_ctx* c, const float* x, bool want_grad, float* grad_out, bool adam, float* x_next)
{
    if (!c->x[0]) return fail(ST_ERR_STATE, "no input image");
    ST_TRY(act_ensure(c, c->act, c->H, c->W));
    ActSet& a = c->act;
    int last = 0;
    for (const ActiveLayer& al : c->active) last = std::max(last, al.blob);
    for (const ActiveLayer& al : c->active) {
        if (al.c && (!c->have_content || c->cH != c->H || c->cW != c->W))
            return fail(ST_ERR_STATE, "content features missing or of a different size than the input");
        if (al.s && !c->have_style) return fail(ST_ERR_STATE, "style Gram matrices missing");
    }
    const bool lean = c->bf16 && c->lean && !c->tile.on;
    ST_TRY(forward_range(c, a, x, last, lean));

    std::vector<const float*> inj(c->nb, nullptr);
    std::fill(c->cnt.begin(), c->cnt.end(), 0);
    c->sf_in.assign(c->nb, nullptr); c->sf_w.assign(c->nb, nullptr);
    for (const ActiveLayer& al : c->active) {
        const int b = al.blob;
        const int C = a.C[b], hw = a.h[b] * a.w[b];
        const size_t n = (size_t)C * hw;
        if (!c->inject[b]) ST_TRY(dmalloc(&c->inject[b], n));
        if (!c->layer_part[b]) ST_TRY(dmalloc(&c->layer_part[b], 5 * kMaxPartials));
        float* part = c->layer_part[b];
        float* nrm = c->norms + b * 3;
        int* cnt = &c->cnt[b * 6];
        bool wrote = false;
        if (al.c || al.d) {
            LayerElemArgs e{};
            e.feat = a.data[b]; e.target = al.c ? c->content_feat[b] : nullptr; e.inject = c->inject[b];
            e.n = n; e.cn_coef = (float)(2.0 / (double)n); e.dn_coef = (float)(-2.0 / (double)n);
            e.cw = al.cw; e.dw = al.dw; e.content = al.c; e.deepdream = al.d;
            e.norm_c = nrm + 0; e.norm_d = nrm + 2;
            e.part_d2 = part; e.part_gc2 = part + kMaxPartials; e.part_f2 = part + 2 * kMaxPartials; e.part_gd2 = part + 3 * kMaxPartials;
            const bool need_norm = (al.c && !c->norm_valid[b * 3 + 0]) || (al.d && !c->norm_valid[b * 3 + 2]);
            int np = 0;
            if (need_norm) {      // first evaluation after reset(): norms are captured (worker.py:253-254,274-275)
                e.write = 0;
                { ProfScope ps(c, P_LAYER_ELEM, 0, 4.0 * n * (al.c ? 2 : 1)); HIP_TRY(launch_layer_elem(e, &np, c->stream)); }
                ProfScope ps(c, P_FINALIZE, 0, 0);
                if (al.c && !c->norm_valid[b * 3 + 0]) { HIP_TRY(launch_finalize_norm(e.part_gc2, np, (double)n, nrm + 0, c->stream)); c->norm_valid[b * 3 + 0] = 1; }
                if (al.d && !c->norm_valid[b * 3 + 2]) { HIP_TRY(launch_finalize_norm(e.part_gd2, np, (double)n, nrm + 2, c->stream)); c->norm_valid[b * 3 + 2] = 1; }
            }
            e.write = 1;
            { ProfScope ps(c, P_LAYER_ELEM, 0, 4.0 * n * (al.c ? 3 : 2)); HIP_TRY(launch_layer_elem(e, &np, c->stream)); }
            cnt[0] = cnt[1] = cnt[2] = cnt[3] = np;
            wrote = true;
        }
        if (al.s) {
            if (!c->dbuf) {          // [C][MPad] scratch for D = G - G_style, sized for the widest blob
                size_t cc = 1;
                for (int i = 0; i < c->nb; ++i) cc = std::max(cc, (size_t)a.C[i] * conv_mpad(a.C[i]));
                ST_TRY(dmalloc(&c->dbuf, cc));
                HIP_TRY(hipMemsetAsync(c->dbuf, 0, cc * sizeof(float), c->stream));
            }
            // bf16 path: the Gram of the CURRENT features is taken from their bf16 copy (the style targets stay fp32 Grams)
            const bool f16_fresh = c->bf16 && !c->tile.on && a.data16[b] && b >= 1 && c->topo[b - 1].is_conv && style_grad16_ok(C, (size_t)hw);
            if (!a.has32[b] && !(a.data16[b] && style_runs16(c, a, b))) return fail(ST_ERR_STATE, "internal: style blob %d has neither an fp32 nor a usable bf16 copy", b);
            ST_TRY(gram_into(c, a.data[b], C, hw, c->style_gram[b], c->dbuf, conv_mpad(C), part + 4 * kMaxPartials, &cnt[4], f16_fresh ? a.data16[b] : nullptr));
            const float c2 = (float)(2.0 / ((double)C * C * (double)n));
            // bf16 path: F from its bf16 copy on the bf16 matrix cores (written by this forward: b <= last, a style layer)
            const bool s16 = c->bf16 && !c->tile.on && a.data16[b] && b >= 1 && c->topo[b - 1].is_conv && style_grad16_ok(C, (size_t)hw);
            // bf16 path, norm known: the gradient rides on the data-gradient conv above this blob; only its trace value is taken here
            const bool fuse = want_grad && s16 && c->norm_valid[b * 3 + 1] && style_fuse_ok(c, a, b, last);
            const int need = fuse ? style_s2_trace_blocks(C) : s16 ? style_grad16_blocks(C, (size_t)hw) : style_grad_blocks(C, a.h[b], a.w[b]);
            if (c->s2_cap[b] < need) { dfree(c->s2_part[b]); ST_TRY(dmalloc(&c->s2_part[b], need)); c->s2_cap[b] = need; }
            if (s16 && style_grad16_pack_elems(C) > c->d16_cap) {
                dfree16(c->d16); c->d16_cap = 0;
                ST_TRY(dmalloc16(&c->d16, style_grad16_pack_elems(C)));
                c->d16_cap = style_grad16_pack_elems(C);
            }
            const double fl = 2.0 * C * C * (double)hw;
            auto style_launch = [&](float* dst, int fused, int accumulate) -> int {
                ProfScope ps(c, P_STYLE_GRAD, fl, n * (s16 ? 6.0 : 8.0));
                if (s16) HIP_TRY(launch_style_grad16(c->dbuf, conv_mpad(C), c->d16, a.data16[b], dst, c2, fused, al.sw, nrm + 1, accumulate, c->s2_part[b], &cnt[5], C, (size_t)hw, c->stream));
                else HIP_TRY(launch_style_grad(c->dbuf, a.data[b], dst, c2, fused, al.sw, nrm + 1, accumulate, c->s2_part[b], &cnt[5], C, a.h[b], a.w[b], c->stream));
                return ST_OK;
            };
            if (fuse) {
                const size_t pe = style_fuse_pack_elems(C, conv_mpad(C));
                if (c->sfuse_cap[b] < pe) { dfree16(c->sfuse_w[b]); c->sfuse_cap[b] = 0; ST_TRY(dmalloc16(&c->sfuse_w[b], pe)); c->sfuse_cap[b] = pe; }
                { ProfScope ps(c, P_MISC, 0, 4.0 * C * C + 2.0 * pe);
                  HIP_TRY(launch_style_fuse_pack(c->dbuf, conv_mpad(C), C, conv_mpad(C), c2, al.sw, nrm + 1, c->sfuse_w[b], c->stream)); }
                { ProfScope ps(c, P_STYLE_GRAD, 2.0 * C * C * (double)C, 12.0 * C * C);
                  HIP_TRY(launch_style_s2_trace(c->dbuf, conv_mpad(C), c->style_gram[b], C, (double)C * hw, c2, c->s2_part[b], &cnt[5], c->stream)); }
                c->sf_in[b] = a.data16[b]; c->sf_w[b] = c->sfuse_w[b];
            } else if (c->norm_valid[b * 3 + 1]) {
                ST_TRY(style_launch(c->inject[b], 1, wrote));
            } else {              // first evaluation: S unscaled -> norm -> saxpy (worker.py:265-269)
                if (!c->stmp) ST_TRY(dmalloc(&c->stmp, c->max_blob));
                ST_TRY(style_launch(c->stmp, 0, 0));
                { ProfScope ps(c, P_FINALIZE, 0, 0);
                  HIP_TRY(launch_finalize_norm(c->s2_part[b], cnt[5], (double)n, nrm + 1, c->stream)); }
                c->norm_valid[b * 3 + 1] = 1;
                ProfScope ps(c, P_VECTOR, 0, 4.0 * n * 3);
                HIP_TRY(launch_scaled_accumulate(c->stmp, c->inject[b], al.sw, nrm + 1, wrote, n, c->stream));
            }
        }
        inj[b] = (c->sf_w[b] && !wrote) ? nullptr : c->inject[b];          // (a fused style term writes nothing into the inject buffer)
    }

    const float* scd = nullptr;
    if (want_grad && !c->active.empty()) {
        if (!c->diffA) { ST_TRY(dmalloc(&c->diffA, c->max_blob)); ST_TRY(dmalloc(&c->diffB, c->max_blob)); }
        if (last == 0) scd = inj[0];
        else {
            std::vector<const float*> below = inj;
            const int rc = backward_chain(c, last, inj[last], below, &scd, lean);
            c->sf_in.assign(c->nb, nullptr); c->sf_w.assign(c->nb, nullptr);      // (the ranged-backward entry points never fuse)
            ST_TRY(rc);
        }
    }

    {
        ImagePassArgs ip{};
        ip.x = x; ip.scd = scd; ip.grad = want_grad ? grad_out : nullptr;
        ip.C = 3; ip.H = c->H; ip.W = c->W;
        ip.tv_w = c->tv_w; ip.tv_beta = c->tv_pow; ip.p_w = c->p_w; ip.p_pow = c->p_pow;
        ip.partial = c->image_part;
        if (adam) {
            // utils.py:58-64: python doubles are rounded to fp32 when they meet the fp32 arrays
            ip.x_out = x_next; ip.m = c->m; ip.v = c->v;
            ip.d1 = (float)0.9; ip.c1 = (float)(1 - 0.9); ip.d2 = (float)0.999; ip.c2 = (float)(1 - 0.999);
            ip.corr1 = (float)(1 - pow(0.9, c->items1)); ip.corr2 = (float)(1 - pow(0.999, c->items2));
            ip.step = (float)c->step_size;
            ip.m_is_zero = c->m_zero; ip.v_is_zero = c->v_zero;
            if (c->capturing) ip.dyn = c->adam_dyn;
        }
        const double n3 = 3.0 * c->H * c->W;
        ProfScope ps(c, P_IMAGE_PASS, 0, 4.0 * n3 * (adam ? 7 : 3));
        HIP_TRY(launch_image_pass(ip, &c->image_cnt, c->stream));
    }

    {
        TraceArgs t{};
        t.n_layers = (int)c->active.size();
        for (int l = 0; l < t.n_layers; ++l) {
            const ActiveLayer& al = c->active[l];
            const int b = al.blob;
            TraceLayer& L = t.layer[l];
            L.content = al.c; L.style = al.s; L.deepdream = al.d;
            L.cw = al.cw; L.sw = al.sw; L.dw = al.dw;
            L.n = (double)a.C[b] * a.h[b] * a.w[b];
            L.gram_n = (double)a.C[b] * a.C[b];
            for (int k = 0; k < 5; ++k) { L.part[k] = c->layer_part[b] + k * kMaxPartials; L.count[k] = c->cnt[b * 6 + k]; }
            L.part[5] = c->s2_part[b]; L.count[5] = c->cnt[b * 6 + 5];
            L.norm = c->norms + b * 3;
        }
        t.image_part = c->image_part; t.image_count = c->image_cnt; t.image_n = 3.0 * c->H * c->W;
        t.tv_w = c->tv_w; t.p_w = c->p_w; t.p_pow = c->p_pow; t.have_grad = want_grad;
        t.out = c->trace_dev;
        t.sums = c->trace_sums;
        c->trace_len_last = t.n_layers * 6 + 8;
        ProfScope ps(c, P_FINALIZE, 0, 0);
        HIP_TRY(launch_finalize_trace(t, c->stream));
    }
    return ST_OK;
}

static int read_trace(st_ctx* c, double* trace, float* loss)
{
    const int n = c->trace_len_last;
    HIP_TRY(hipMemcpyAsync(c->trace_host, c->trace_dev, n * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (trace) for (int i = 0; i < n; ++i) trace[i] = c->trace_host[i];
    c->last_loss = c->trace_host[n - 2];
    if (loss) *loss = c->last_loss;
    return ST_OK;
}

// ------------------------------------------------------------------------------------------ L-BFGS
static int lbfgs_alloc(st_ctx* c)
{
    const size_t n3 = (size_t)3 * c->H * c->W;
    if (!c->g_cur) ST_TRY(dmalloc(&c->g_cur, n3));
    if (!c->pvec) ST_TRY(dmalloc(&c->pvec, n3));
    for (int i = 0; i <= st_ctx::kCorr; ++i) {
        if (!c->hs[i]) ST_TRY(dmalloc(&c->hs[i], n3));
        if (!c->hy[i]) ST_TRY(dmalloc(&c->hy[i], n3));
    }
    return ST_OK;
}

static LbfgsArgs lbfgs_args(st_ctx* c, int apply)
{
    LbfgsArgs a{};
    for (int i = 0; i < kLbfgsSlots; ++i) { a.v.s[i] = c->hs[i]; a.v.y[i] = c->hy[i]; }
    a.st = c->lb_dev; a.part = c->lb_part; a.part2 = c->lb_part + 2 * kMaxPartials;
    a.g = c->g_cur; a.p = c->pvec; a.x = c->x[c->cur];
    a.n = (size_t)3 * c->H * c->W; a.step = (float)c->step_size; a.apply = apply;
    return a;
}

// One LBFGSOptimizer.step (optimizers.py:62-77).  Nothing is read back: the pair count, the ring order and the
// s.y > 1e-10 decision live on the device (lbfgs.hip), so consecutive steps queue up like Adam steps do.
static int lbfgs_step(st_ctx* c)
{
    ST_TRY(lbfgs_alloc(c));
    const size_t n = (size_t)3 * c->H * c->W;
    float* x = c->x[c->cur];
    hipStream_t s = c->stream;
    if (c->lb_clear) {              // objective_changed / a new optimizer: sy = [], ss = [], ys = [] (optimizers.py:121-125)
        HIP_TRY(hipMemsetAsync(c->lb_dev, 0, sizeof(LbfgsDev), s));
        c->lb_clear = false;
    }
    if (!c->have_cur) {             // optimizers.py:64-65
        ST_TRY(eval_objective(c, x, true, c->g_cur, false, nullptr));
        c->have_cur = true;
    }
    {   // s = -step * inv_hv(grad) ; x += s          (optimizers.py:68-69, 89-108)
        ProfScope ps(c, P_VECTOR, 0, 4.0 * n * (8.0 * kLbfgsCorr + 3.0));
        HIP_TRY(launch_lbfgs_two_loop(lbfgs_args(c, 1), s));
    }
    ST_TRY(eval_objective(c, x, true, c->grad, false, nullptr));       // new loss / grad (optimizers.py:72)
    {   // y = grad - self.grad ; store_curvature_pair(s, y)            (optimizers.py:73-87)
        ProfScope ps(c, P_VECTOR, 0, 4.0 * n * 4.0);
        HIP_TRY(launch_lbfgs_pair(lbfgs_args(c, 1), c->grad, 0, s));
    }
    std::swap(c->g_cur, c->grad);
    return ST_OK;
}

// ---- hipGraph replay of the steady-state Adam step ---------------------------------------------------------------
// At small image sizes a step is ~60 dependent launches of a few microseconds each.  Measured on MI355X (round 1): the
// replay is bit-identical and exactly as fast as plain launches (128 px: 0.92 vs 0.91 ms, 256 px: 1.14 vs 1.13 ms) -- the
// step is bound by the execution latency of the dependent kernel chain, not by launch overhead -- so it is OFF unless
// ST2_GRAPH=1.  In steady state
// (norms frozen, Adam moments live, nothing reconfigured) the launch sequence and every argument except the two Adam
// bias corrections and the step size are constant per parity of the x ping-pong, so the step is captured once per
// parity and replayed; those three scalars travel through a 12-byte device buffer written by a 1-thread kernel.
static bool step_graph_ok(const st_ctx* c)
{
    if (!c->graphs || c->prof_on || c->tile.on || c->m_zero || c->v_zero || c->active.empty()) return false;
    if ((size_t)c->H * c->W > c->graph_max_px) return false;
    if (c->plain_epoch != c->epoch || c->plain_steps < 1) return false;       // one plain step first: lazy allocations, norm capture
    for (const ActiveLayer& al : c->active) {
        if (al.c && !c->norm_valid[al.blob * 3 + 0]) return false;
        if (al.s && !c->norm_valid[al.blob * 3 + 1]) return false;
        if (al.d && !c->norm_valid[al.blob * 3 + 2]) return false;
    }
    return true;
}

static int step_graph_capture(st_ctx* c, int par)
{
    if (c->gexec[par]) { (void)hipGraphExecDestroy(c->gexec[par]); c->gexec[par] = nullptr; }
    if (!c->adam_dyn) ST_TRY(dmalloc(&c->adam_dyn, 4));
    hipGraph_t g = nullptr;
    HIP_TRY(hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal));
    c->capturing = true;
    const int rc = eval_objective(c, c->x[par], true, nullptr, true, c->x[par ^ 1]);
    c->capturing = false;
    const hipError_t e = hipStreamEndCapture(c->stream, &g);
    if (rc != ST_OK || e != hipSuccess || !g) {      // something in the step is not capturable here: plain launches from now on
        if (g) (void)hipGraphDestroy(g);
        (void)hipGetLastError();
        c->graphs = false;
        return ST_OK;
    }
    const hipError_t ei = hipGraphInstantiate(&c->gexec[par], g, nullptr, nullptr, 0);
    (void)hipGraphDestroy(g);
    if (ei != hipSuccess) { c->gexec[par] = nullptr; (void)hipGetLastError(); c->graphs = false; return ST_OK; }
    c->gepoch[par] = c->epoch;
    return ST_OK;
}

// ------------------------------------------------------------------------------------------- C ABI
extern "C" {

int st_create(st_ctx** out, int device_id, const st_layer_desc* layers, int n_layers)
{
    if (!out) return fail(ST_ERR_ARG, "out is NULL");
    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if (ndev <= 0) return fail(ST_ERR_HIP, "no HIP device visible");
    if (device_id < 0 || device_id >= ndev) return fail(ST_ERR_ARG, "device %d out of range (%d visible)", device_id, ndev);
    HIP_TRY(hipSetDevice(device_id));
    st_ctx* c = new st_ctx();
    c->device = device_id;
    { const char* e = getenv("ST2_WINO"); if (e && *e) c->wino = atoi(e) != 0; }
    { const char* e = getenv("ST2_GRAPH"); if (e && *e) c->graphs = atoi(e) != 0; }
    { const char* e = getenv("ST2_GRAPH_MAX_PX"); if (e && *e) c->graph_max_px = (size_t)atoll(e) * (size_t)atoll(e); }
    if (n_layers <= 0) {
        for (const auto& l : kVgg19) {
            Layer L; L.is_conv = l.kind == 0; L.name = l.name; L.cin = l.cin; L.cout = l.cout;
            c->topo.push_back(L);
        }
    } else {
        int cprev = 3;
        for (int i = 0; i < n_layers; ++i) {
            Layer L; L.is_conv = layers[i].kind == ST_LAYER_CONV; L.name = layers[i].name ? layers[i].name : "";
            if (L.is_conv) {
                L.cin = layers[i].cin; L.cout = layers[i].cout;
                if (L.cin != cprev || L.cout <= 0) { delete c; return fail(ST_ERR_ARG, "layer %s: cin %d does not follow %d", L.name.c_str(), L.cin, cprev); }
                cprev = L.cout;
            }
            c->topo.push_back(L);
        }
    }
    if ((int)c->topo.size() + 1 > kMaxTraceLayers) { delete c; return fail(ST_ERR_ARG, "too many layers"); }
    c->blob_names.push_back("data");
    for (const Layer& L : c->topo) c->blob_names.push_back(L.name);
    c->nb = (int)c->blob_names.size();
    HIP_TRY(hipStreamCreate(&c->stream));
    c->content_feat.assign(c->nb, nullptr);
    c->style_gram.assign(c->nb, nullptr);
    c->inject.assign(c->nb, nullptr);
    c->layer_part.assign(c->nb, nullptr);
    c->s2_part.assign(c->nb, nullptr);
    c->sfuse_w.assign(c->nb, nullptr); c->sfuse_cap.assign(c->nb, 0);
    c->s2_cap.assign(c->nb, 0);
    c->cnt.assign(c->nb * 6, 0);
    c->norm_valid.assign(c->nb * 3, 0);
    ST_TRY(dmalloc(&c->norms, c->nb * 3));
    ST_TRY(dmalloc(&c->image_part, 6 * kMaxPartials));
    ST_TRY(dmalloc(&c->trace_dev, kMaxTraceLayers * 6 + 8));
    HIP_TRY(hipMalloc((void**)&c->trace_sums, (kMaxTraceLayers * kLayerSlots + kImageSlots) * sizeof(double)));
    HIP_TRY(hipMalloc((void**)&c->lb_dev, sizeof(LbfgsDev)));
    HIP_TRY(hipMemset(c->lb_dev, 0, sizeof(LbfgsDev)));
    ST_TRY(dmalloc(&c->lb_part, 4 * kMaxPartials));
    HIP_TRY(hipHostMalloc((void**)&c->trace_host, (kMaxTraceLayers * 6 + 8) * sizeof(float), 0));
    // worker.py:129-133: all-ones weights over every blob until SetWeights arrives
    for (int b = 0; b < c->nb; ++b) c->rows.push_back(ActiveLayer{b, 1.f, 1.f, 1.f, true, true, true});
    c->active = c->rows;
    *out = c;
    return ST_OK;
}

int st_destroy(st_ctx* c)
{
    if (!c) return ST_OK;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    for (int i = 0; i < 2; ++i) if (c->gexec[i]) { (void)hipGraphExecDestroy(c->gexec[i]); c->gexec[i] = nullptr; }
    dfree(c->adam_dyn);
    for (Layer& L : c->topo) { dfree(L.w_fwd); dfree(L.w_bwd); dfree(L.w_raw); dfree(L.w_raw_r); dfree(L.bias); dfree16(L.w16_fwd); dfree16(L.w16_bwd); dfree(L.u_fwd); dfree(L.u_bwd); }
    dfree16(c->diff16A); dfree16(c->diff16B);
    act_free(c->act);
    for (int i = 0; i < 2; ++i) dfree(c->x[i]);
    dfree(c->fwd_x);
    dfree(c->grad); dfree(c->m); dfree(c->v); dfree(c->g_cur); dfree(c->pvec);
    for (int i = 0; i <= st_ctx::kCorr; ++i) { dfree(c->hs[i]); dfree(c->hy[i]); }
    for (auto& p : c->content_feat) dfree(p);
    dfree(c->content_x);
    for (auto& p : c->style_gram) dfree(p);
    for (auto& p : c->inject) dfree(p);
    for (auto& p : c->layer_part) dfree(p);
    for (auto& p : c->s2_part) dfree(p);
    for (auto& p : c->sfuse_w) dfree16(p);
    dfree(c->diffA); dfree(c->diffB); dfree(c->stmp); dfree(c->gram_slabs); dfree(c->gram_fold); dfree(c->dbuf); dfree16(c->d16); dfree(c->conv_scratch);
    dfree(c->tile.p1); dfree(c->tile.p2); dfree(c->tile.p3); dfree(c->tile.pd); dfree(c->tile.wgrad);
    dfree(c->norms); dfree(c->image_part); dfree(c->trace_dev); dfree(c->lb_part); dfree(c->hwc_dev);
    if (c->lb_dev) (void)hipFree(c->lb_dev);
    if (c->stage_dev) (void)hipFree(c->stage_dev);
    if (c->trace_sums) (void)hipFree(c->trace_sums);
    if (c->trace_host) (void)hipHostFree(c->trace_host);
    if (c->pipe.copy) {
        (void)hipStreamSynchronize(c->pipe.copy);
        for (int i = 0; i < st_ctx::Pipe::kSlots; ++i) {
            dfree(c->pipe.hwc[i]);
            if (c->pipe.img_pin[i]) (void)hipHostFree(c->pipe.img_pin[i]);
            if (c->pipe.trace_pin[i]) (void)hipHostFree(c->pipe.trace_pin[i]);
            if (c->pipe.ready[i]) (void)hipEventDestroy(c->pipe.ready[i]);
            if (c->pipe.done[i]) (void)hipEventDestroy(c->pipe.done[i]);
        }
        (void)hipStreamDestroy(c->pipe.copy);
    }
    for (hipEvent_t e : c->ev_pool) (void)hipEventDestroy(e);
    (void)hipStreamDestroy(c->stream);
    delete c;
    return ST_OK;
}

int st_load_conv_weights(st_ctx* c, const char* layer, const float* w, const float* bias)
{
    if (c) c->epoch++;       // anything but st_step may change what a step launches: captured step graphs are stale
    if (!c || !layer || !w) return fail(ST_ERR_ARG, "NULL argument");
    HIP_TRY(hipSetDevice(c->device));
    for (Layer& L : c->topo) {
        if (!L.is_conv || L.name != layer) continue;
        const size_t nf = conv_pack_floats(L.cin, L.cout), nb = conv_pack_floats(L.cout, L.cin);
        std::vector<float> pf(nf), pb(nb), bp(conv_mpad(L.cout), 0.f);
        pack_conv_weights_fwd(w, L.cout, L.cin, pf.data());
        pack_conv_weights_dgrad(w, L.cout, L.cin, pb.data());
        if (bias) memcpy(bp.data(), bias, L.cout * sizeof(float));
        dfree(L.w_fwd); dfree(L.w_bwd); dfree(L.w_raw); dfree(L.w_raw_r); dfree(L.bias); dfree16(L.w16_fwd); dfree16(L.w16_bwd); dfree(L.u_fwd); dfree(L.u_bwd);
        for (int dir = 0; dir < 2; ++dir) {   // Winograd packs for the directions the Winograd kernel can take (any image size)
            const int K = dir ? L.cout : L.cin, M = dir ? L.cin : L.cout;
            if (!conv_wino_ok(K, M, 4, 4)) continue;
            std::vector<float> hu(wino_pack_floats(K, M));
            if (dir) pack_wino_weights_dgrad(w, L.cout, L.cin, hu.data()); else pack_wino_weights_fwd(w, L.cout, L.cin, hu.data());
            float** dst = dir ? &L.u_bwd : &L.u_fwd;
            ST_TRY(dmalloc(dst, hu.size()));
            HIP_TRY(hipMemcpy(*dst, hu.data(), hu.size() * sizeof(float), hipMemcpyHostToDevice));
        }
        {   // bf16 packs for the bf16 feature path
            const size_t n16f = conv16_pack_elems(L.cin, L.cout), n16b = conv16_pack_elems(L.cout, L.cin);
            std::vector<unsigned short> hf(n16f), hb(n16b);
            pack_conv_weights16_fwd(w, L.cout, L.cin, hf.data());
            pack_conv_weights16_dgrad(w, L.cout, L.cin, hb.data());
            ST_TRY(dmalloc16(&L.w16_fwd, n16f)); ST_TRY(dmalloc16(&L.w16_bwd, n16b));
            HIP_TRY(hipMemcpy(L.w16_fwd, hf.data(), n16f * 2, hipMemcpyHostToDevice));
            HIP_TRY(hipMemcpy(L.w16_bwd, hb.data(), n16b * 2, hipMemcpyHostToDevice));
        }
        ST_TRY(dmalloc(&L.w_fwd, nf)); ST_TRY(dmalloc(&L.w_bwd, nb));
        ST_TRY(dmalloc(&L.w_raw, (size_t)L.cout * L.cin * 9)); ST_TRY(dmalloc(&L.bias, bp.size()));
        HIP_TRY(hipMemcpy(L.w_fwd, pf.data(), nf * sizeof(float), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(L.w_bwd, pb.data(), nb * sizeof(float), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(L.w_raw, w, (size_t)L.cout * L.cin * 9 * sizeof(float), hipMemcpyHostToDevice));
        if (conv_dgrad_smallM_ok(L.cout, L.cin) && L.cout % 8 == 0) {       // bf16 path: the first layer's dgrad multiplies bf16 operands
            std::vector<float> wr((size_t)L.cout * L.cin * 9);
            for (size_t k = 0; k < wr.size(); ++k) {
                unsigned u; memcpy(&u, &w[k], 4);
                u = (u + 0x7fffu + ((u >> 16) & 1u)) & 0xffff0000u;          // round to nearest even (weights are finite)
                memcpy(&wr[k], &u, 4);
            }
            ST_TRY(dmalloc(&L.w_raw_r, wr.size()));
            HIP_TRY(hipMemcpy(L.w_raw_r, wr.data(), wr.size() * sizeof(float), hipMemcpyHostToDevice));
        }
        HIP_TRY(hipMemcpy(L.bias, bp.data(), bp.size() * sizeof(float), hipMemcpyHostToDevice));
        L.loaded = true;
        return ST_OK;
    }
    return fail(ST_ERR_ARG, "no conv layer named %s", layer);
}

int st_set_conv_algo(st_ctx* c, int winograd)
{
    if (c) c->epoch++;       // anything but st_step may change what a step launches: captured step graphs are stale
    if (!c) return fail(ST_ERR_ARG, "ctx is NULL");
    c->wino = winograd != 0;
    return ST_OK;
}

int st_set_precision(st_ctx* c, int bf16_features)
{
    if (c) c->epoch++;       // anything but st_step may change what a step launches: captured step graphs are stale
    if (!c) return fail(ST_ERR_ARG, "ctx is NULL");
    c->bf16 = bf16_features != 0;
    // 1: the lean data flow (objective evaluations write fp32 only where something reads fp32); 2: every fp32 blob and diff
    // as in round 1 (A/B reference of the tests); environment ST2_BF16_LEAN=0 forces 2
    const char* e = getenv("ST2_BF16_LEAN");
    c->lean = bf16_features == 1 && !(e && *e == '0');
    return ST_OK;
}

int st_num_blobs(st_ctx* c) { return c ? c->nb : 0; }
const char* st_blob_name(st_ctx* c, int i) { return (c && i >= 0 && i < c->nb) ? c->blob_names[i].c_str() : nullptr; }

int st_blob_shape(st_ctx* c, int index, int H, int W, int* oc, int* oh, int* ow)
{
    if (!c || index < 0 || index >= c->nb) return fail(ST_ERR_ARG, "bad blob index %d", index);
    std::vector<int> C, h, w;
    shapes_for(c, H, W, C, h, w);
    if (oc) *oc = C[index];
    if (oh) *oh = h[index];
    if (ow) *ow = w[index];
    return ST_OK;
}

// ---- model test hooks
int st_forward(st_ctx* c, const float* x_nchw, int H, int W, int last_blob)
{
    if (c) c->epoch++;       // anything but st_step may change what a step launches: captured step graphs are stale
    if (!c || !x_nchw || H <= 0 || W <= 0) return fail(ST_ERR_ARG, "bad argument");
    HIP_TRY(hipSetDevice(c->device));
    // The probe image lives in a buffer of its own: the job's iterate (x) is never touched.  A forward at ANOTHER
    // geometry re-creates the size-dependent buffers exactly like st_set_input at a new size does (optimizer state
    // starts from zero again); st_backward works on this geometry.
    ST_TRY(set_input_common(c, H, W));
    ST_TRY(act_ensure(c, c->act, H, W));
    const size_t n3 = (size_t)3 * H * W;
    if (n3 > c->fwd_x_cap) { dfree(c->fwd_x); c->fwd_x_cap = 0; ST_TRY(dmalloc(&c->fwd_x, n3)); c->fwd_x_cap = n3; }
    HIP_TRY(hipMemcpyAsync(c->fwd_x, x_nchw, n3 * sizeof(float), hipMemcpyHostToDevice, c->stream));
    if (last_blob < 0 || last_blob >= c->nb) last_blob = c->nb - 1;
    ST_TRY(forward_range(c, c->act, c->fwd_x, last_blob));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return ST_OK;
}

int st_get_blob(st_ctx* c, int index, float* out)
{
    if (!c || index < 0 || index >= c->nb || !out) return fail(ST_ERR_ARG, "bad argument");
    if (index > c->act.valid_to) return fail(ST_ERR_STATE, "blob %d was not computed by the last forward", index);
    if (!c->act.has32[index]) return fail(ST_ERR_STATE, "blob %d (%s) is not materialised in fp32 by the lean bf16 evaluation (st_set_precision(ctx, 2) keeps every blob)", index, c->blob_names[index].c_str());
    const size_t n = (size_t)c->act.C[index] * c->act.h[index] * c->act.w[index];
    HIP_TRY(hipMemcpy(out, c->act.data[index], n * sizeof(float), hipMemcpyDeviceToHost));
    return ST_OK;
}

int st_backward(st_ctx* c, int n, const int* blob_index, const float* const* diffs, float* out_grad)
{
    if (c) c->epoch++;       // anything but st_step may change what a step launches: captured step graphs are stale
    if (!c || !out_grad || n < 0) return fail(ST_ERR_ARG, "bad argument");
    if (c->act.valid_to < 0) return fail(ST_ERR_STATE, "st_forward first");
    HIP_TRY(hipSetDevice(c->device));
    const size_t n3 = (size_t)3 * c->H * c->W;
    std::vector<const float*> inj(c->nb, nullptr);
    int top = -1;
    for (int i = 0; i < n; ++i) {
        const int b = blob_index[i];
        if (b < 0 || b > c->act.valid_to) return fail(ST_ERR_ARG, "diff for blob %d which the last forward did not reach", b);
        const size_t nb = (size_t)c->act.C[b] * c->act.h[b] * c->act.w[b];
        if (!c->inject[b]) ST_TRY(dmalloc(&c->inject[b], nb));
        HIP_TRY(hipMemcpyAsync(c->inject[b], diffs[i], nb * sizeof(float), hipMemcpyHostToDevice, c->stream));
        inj[b] = c->inject[b];
        top = std::max(top, b);
    }
    if (top < 0) { memset(out_grad, 0, n3 * sizeof(float)); return ST_OK; }
    if (!c->diffA) { ST_TRY(dmalloc(&c->diffA, c->max_blob)); ST_TRY(dmalloc(&c->diffB, c->max_blob)); }
    const float* g = inj[0];
    if (top > 0) ST_TRY(backward_chain(c, top, inj[top], inj, &g));
    HIP_TRY(hipMemcpyAsync(out_grad, g, n3 * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return ST_OK;
}

int st_gram(st_ctx* c, int index, float* out)
{
    if (c) c->epoch++;       // anything but st_step may change what a step launches: captured step graphs are stale
    if (!c || index < 0 || index >= c->nb || !out) return fail(ST_ERR_ARG, "bad argument");
    if (index > c->act.valid_to) return fail(ST_ERR_STATE, "blob %d was not computed by the last forward", index);
    HIP_TRY(hipSetDevice(c->device));
    const int C = c->act.C[index], hw = c->act.h[index] * c->act.w[index];
    float* g = nullptr;
    ST_TRY(dmalloc(&g, (size_t)C * C));
    int r = gram_into(c, c->act.data[index], C, hw, nullptr, g, C, nullptr, nullptr);
    if (r == ST_OK) {
        hipError_t e = hipMemcpyAsync(out, g, (size_t)C * C * sizeof(float), hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) r = fail(ST_ERR_HIP, "gram copy: %s", hipGetErrorString(e));
    }
    dfree(g);
    return r;
}

// ---- image slots
int st_set_input(st_ctx* c, const void* hwc, int H, int W, int is_u8)
{
    if (c) c->epoch++;       // anything but st_step may change what a step launches: captured step graphs are stale
    if (!c) return fail(ST_ERR_ARG, "ctx is NULL");
    HIP_TRY(hipSetDevice(c->device));
    ST_TRY(set_input_common(c, H, W));
    ST_TRY(preprocess_into(c, hwc, H, W, is_u8, c->x[c->cur]));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return ST_OK;
}

int st_set_input_nchw(st_ctx* c, const float* x, int H, int W)
{
    if (c) c->epoch++;       // anything but st_step may change what a step launches: captured step graphs are stale
    if (!c || !x || H <= 0 || W <= 0) return fail(ST_ERR_ARG, "bad argument");
    HIP_TRY(hipSetDevice(c->device));
    ST_TRY(set_input_common(c, H, W));
    HIP_TRY(hipMemcpy(c->x[c->cur], x, (size_t)3 * H * W * sizeof(float), hipMemcpyHostToDevice));
    return ST_OK;
}

int st_get_input_nchw(st_ctx* c, float* out)
{
    if (!c || !out || !c->x[0]) return fail(ST_ERR_STATE, "no input image");
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipMemcpy(out, c->x[c->cur], (size_t)3 * c->H * c->W * sizeof(float), hipMemcpyDeviceToHost));
    return ST_OK;
}

int st_input_shape(st_ctx* c, int* H, int* W)
{
    if (!c) return fail(ST_ERR_ARG, "ctx is NULL");
    if (H) *H = c->H;
    if (W) *W = c->W;
    return ST_OK;
}

static int content_from_device(st_ctx* c, const float* xdev, int H, int W)
{
    ST_TRY(act_ensure(c, c->act, H, W));
    ST_TRY(forward_range(c, c->act, xdev, c->nb - 1));
    for (int i = 0; i < c->nb; ++i) {
        const size_t n = (size_t)c->act.C[i] * c->act.h[i] * c->act.w[i];
        if (c->cH != H || c->cW != W || !c->content_feat[i]) { dfree(c->content_feat[i]); ST_TRY(dmalloc(&c->content_feat[i], n)); }
        HIP_TRY(hipMemcpyAsync(c->content_feat[i], c->act.data[i], n * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
    }
    if (xdev != c->content_x) {                 // keep the preprocessed image itself (== blob "data")
        if (c->cH != H || c->cW != W || !c->content_x) { dfree(c->content_x); ST_TRY(dmalloc(&c->content_x, (size_t)3 * H * W)); }
        HIP_TRY(hipMemcpyAsync(c->content_x, xdev, (size_t)3 * H * W * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
    }
    c->cH = H; c->cW = W;
    c->have_content = true;
    HIP_TRY(hipStreamSynchronize(c->stream));
    return ST_OK;
}

int st_set_content(st_ctx* c, const void* hwc, int H, int W, int is_u8)
{
    if (c) c->epoch++;       // anything but st_step may change what a step launches: captured step graphs are stale
    if (!c) return fail(ST_ERR_ARG, "ctx is NULL");
    HIP_TRY(hipSetDevice(c->device));
    float* tmp = nullptr;
    ST_TRY(dmalloc(&tmp, (size_t)3 * H * W));
    int r = preprocess_into(c, hwc, H, W, is_u8, tmp);
    if (r == ST_OK) r = content_from_device(c, tmp, H, W);
    (void)hipStreamSynchronize(c->stream);
    dfree(tmp);
    return r;
}

int st_set_content_nchw(st_ctx* c, const float* x, int H, int W)
{
    if (c) c->epoch++;       // anything but st_step may change what a step launches: captured step graphs are stale
    if (!c || !x || H <= 0 || W <= 0) return fail(ST_ERR_ARG, "bad argument");
    HIP_TRY(hipSetDevice(c->device));
    float* tmp = nullptr;
    ST_TRY(dmalloc(&tmp, (size_t)3 * H * W));
    int r = ST_OK;
    if (hipMemcpy(tmp, x, (size_t)3 * H * W * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) r = fail(ST_ERR_HIP, "content upload failed");
    if (r == ST_OK) r = content_from_device(c, tmp, H, W);
    (void)hipStreamSynchronize(c->stream);
    dfree(tmp);
    return r;
}

int st_set_style(st_ctx* c, const void* hwc, int H, int W, int is_u8)
{
    if (c) c->epoch++;       // anything but st_step may change what a step launches: captured step graphs are stale
    if (!c) return fail(ST_ERR_ARG, "ctx is NULL");
    HIP_TRY(hipSetDevice(c->device));
    float* tmp = nullptr;
    ST_TRY(dmalloc(&tmp, (size_t)3 * H * W));
    ActSet aux;
    ActSet* a = &aux;
    const bool same = c->act.H == H && c->act.W == W && !c->act.data.empty();
    if (same) a = &c->act;
    int r = preprocess_into(c, hwc, H, W, is_u8, tmp);
    if (r == ST_OK) r = act_ensure(c, *a, H, W);
    if (r == ST_OK) r = forward_range(c, *a, tmp, c->nb - 1);
    for (int i = 0; i < c->nb && r == ST_OK; ++i) {
        const int C = a->C[i], hw = a->h[i] * a->w[i];
        if (!c->style_gram[i]) r = dmalloc(&c->style_gram[i], (size_t)C * C);
        if (r == ST_OK) r = gram_into(c, a->data[i], C, hw, nullptr, c->style_gram[i], C, nullptr, nullptr);
    }
    (void)hipStreamSynchronize(c->stream);
    if (!same) act_free(aux);
    else c->act.valid_to = -1;
    dfree(tmp);
    if (r == ST_OK) c->have_style = true;
    return r;
}

// ---- objective
int st_set_weights(st_ctx* c, int n_rows, const int* blob_index, const float* content, const float* style,
                   const float* deepdream, const double params[4])
{
    if (c) c->epoch++;       // anything but st_step may change what a step launches: captured step graphs are stale
    if (!c || n_rows < 0 || (n_rows && (!blob_index || !content || !style || !deepdream)) || !params)
        return fail(ST_ERR_ARG, "bad argument");
    std::vector<ActiveLayer> rows;
    for (int i = 0; i < n_rows; ++i) {
        const int b = blob_index[i];
        if (b < 0 || b >= c->nb) return fail(ST_ERR_ARG, "row %d names blob %d", i, b);
        rows.push_back(ActiveLayer{b, content[i], style[i], deepdream[i], nonzero(content[i]), nonzero(style[i]), nonzero(deepdream[i])});
    }
    c->rows = rows;
    c->active.clear();
    for (const ActiveLayer& r : rows) if (r.c || r.s || r.d) c->active.push_back(r);
    c->tv_w = (float)params[0]; c->tv_pow = (float)params[1]; c->p_w = (float)params[2]; c->p_pow = (float)params[3];
    return ST_OK;
}

int st_clear_norms(st_ctx* c)
{
    if (c) c->epoch++;       // anything but st_step may change what a step launches: captured step graphs are stale
    if (!c) return fail(ST_ERR_ARG, "ctx is NULL");
    std::fill(c->norm_valid.begin(), c->norm_valid.end(), 0);
    return ST_OK;
}

int st_trace_len(st_ctx* c) { return c ? (int)c->active.size() * 6 + 8 : 0; }

int st_opfunc(st_ctx* c, float* out_loss, float* out_grad, double* trace)
{
    if (c) c->epoch++;       // anything but st_step may change what a step launches: captured step graphs are stale
    if (!c) return fail(ST_ERR_ARG, "ctx is NULL");
    HIP_TRY(hipSetDevice(c->device));
    ST_TRY(eval_objective(c, c->x[c->cur], out_grad != nullptr, c->grad, false, nullptr));
    if (out_grad) HIP_TRY(hipMemcpyAsync(out_grad, c->grad, (size_t)3 * c->H * c->W * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    return read_trace(c, trace, out_loss);
}

// ---- optimizers
int st_optimizer_reset(st_ctx* c, int kind, double step_size)
{
    if (c) c->epoch++;       // anything but st_step may change what a step launches: captured step graphs are stale
    if (!c || (kind != ST_OPT_ADAM && kind != ST_OPT_LBFGS)) return fail(ST_ERR_ARG, "bad optimizer kind %d", kind);
    c->opt_kind = kind;
    c->step_size = step_size;
    c->items1 = c->items2 = 0;
    c->m_zero = c->v_zero = true;
    c->lb_clear = true;
    c->have_cur = false;
    return ST_OK;
}

int st_optimizer_set_step(st_ctx* c, double step_size)
{
    if (c) c->epoch++;       // anything but st_step may change what a step launches: captured step graphs are stale
    if (!c) return fail(ST_ERR_ARG, "ctx is NULL");
    c->step_size = step_size;
    return ST_OK;
}

int st_optimizer_kind(st_ctx* c) { return c ? c->opt_kind : ST_OPT_NONE; }

int st_objective_changed(st_ctx* c)
{
    if (c) c->epoch++;       // anything but st_step may change what a step launches: captured step graphs are stale
    if (!c) return fail(ST_ERR_ARG, "ctx is NULL");
    if (c->opt_kind == ST_OPT_ADAM) {            // optimizers.py:42-46: t = 0, g1.clear(); g2 persists
        c->items1 = 0;
        c->m_zero = true;
    } else if (c->opt_kind == ST_OPT_LBFGS) {    // optimizers.py:121-125
        c->lb_clear = true;
        c->have_cur = false;
    }
    return ST_OK;
}

int st_adam_get_state(st_ctx* c, float* m, float* v, int* items1, int* items2)
{
    if (!c || !c->x[0]) return fail(ST_ERR_STATE, "no input image");
    const size_t bytes = (size_t)3 * c->H * c->W * sizeof(float);
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (m) { if (c->m_zero) memset(m, 0, bytes); else HIP_TRY(hipMemcpy(m, c->m, bytes, hipMemcpyDeviceToHost)); }
    if (v) { if (c->v_zero) memset(v, 0, bytes); else HIP_TRY(hipMemcpy(v, c->v, bytes, hipMemcpyDeviceToHost)); }
    if (items1) *items1 = c->items1;
    if (items2) *items2 = c->items2;
    return ST_OK;
}

int st_adam_set_state(st_ctx* c, const float* m, const float* v, int items1, int items2)
{
    if (c) c->epoch++;       // anything but st_step may change what a step launches: captured step graphs are stale
    if (!c || !c->x[0]) return fail(ST_ERR_STATE, "no input image");
    const size_t bytes = (size_t)3 * c->H * c->W * sizeof(float);
    if (m) { HIP_TRY(hipMemcpy(c->m, m, bytes, hipMemcpyHostToDevice)); c->m_zero = false; } else c->m_zero = true;
    if (v) { HIP_TRY(hipMemcpy(c->v, v, bytes, hipMemcpyHostToDevice)); c->v_zero = false; } else c->v_zero = true;
    c->items1 = items1; c->items2 = items2;
    return ST_OK;
}

// the optimizer step itself: everything st_step launches before the iterate is read back
static int step_enqueue(st_ctx* c)
{
    if (c->opt_kind == ST_OPT_ADAM) {
        c->items1 += 1; c->items2 += 1;          // DecayingMean.__call__(item), utils.py:58-61
        bool replayed = false;
        if (step_graph_ok(c)) {
            const int par = c->cur;
            if (!c->gexec[par] || c->gepoch[par] != c->epoch) ST_TRY(step_graph_capture(c, par));
            if (c->gexec[par]) {
                // the only per-step arguments (utils.py:58-64: python doubles rounded to fp32 where they meet the arrays)
                HIP_TRY(launch_set_scalars3(c->adam_dyn, (float)(1 - pow(0.9, c->items1)), (float)(1 - pow(0.999, c->items2)),
                                            (float)c->step_size, c->stream));
                HIP_TRY(hipGraphLaunch(c->gexec[par], c->stream));
                replayed = true;
                c->graph_replays += 1;
            }
        }
        if (!replayed) {
            ST_TRY(eval_objective(c, c->x[c->cur], true, nullptr, true, c->x[c->cur ^ 1]));
            if (c->plain_epoch != c->epoch) { c->plain_epoch = c->epoch; c->plain_steps = 0; }
            c->plain_steps += 1;
        }
        c->cur ^= 1;
        c->m_zero = c->v_zero = false;
    } else if (c->opt_kind == ST_OPT_LBFGS) {
        ST_TRY(lbfgs_step(c));
    } else {
        return fail(ST_ERR_STATE, "no optimizer: call st_optimizer_reset first");
    }
    return ST_OK;
}

int st_step(st_ctx* c, float* out_hwc, double* trace, float* out_loss)
{
    if (!c) return fail(ST_ERR_ARG, "ctx is NULL");
    if (!c->x[0]) return fail(ST_ERR_STATE, "no input image");
    if (c->pipe.count) return fail(ST_ERR_STATE, "%d pipelined iteration(s) in flight: st_step_end first", c->pipe.count);
    HIP_TRY(hipSetDevice(c->device));
    ST_TRY(step_enqueue(c));
    if (out_hwc) {
        { ProfScope ps(c, P_MISC, 0, 0); HIP_TRY(launch_deprocess(c->x[c->cur], c->hwc_dev, c->H, c->W, c->stream)); }
        HIP_TRY(hipMemcpyAsync(out_hwc, c->hwc_dev, (size_t)3 * c->H * c->W * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    }
    if (out_hwc || trace || out_loss) return read_trace(c, trace, out_loss);
    return ST_OK;
}

// Pipelined form of st_step for the worker loop (worker.py:380-395: step, send Iterate, poll, step ...): begin() queues the
// iteration and the asynchronous copy of its iterate / trace, end() hands the OLDEST queued iteration's results over.  With one
// begin() ahead of every end() the GPU starts iteration k + 1 while iterate k crosses PCIe and is pickled.
int st_step_begin(st_ctx* c)
{
    if (!c) return fail(ST_ERR_ARG, "ctx is NULL");
    if (!c->x[0]) return fail(ST_ERR_STATE, "no input image");
    st_ctx::Pipe& p = c->pipe;
    if (p.count >= 2) return fail(ST_ERR_STATE, "two iterations are already in flight: st_step_end first");
    HIP_TRY(hipSetDevice(c->device));
    const size_t n3 = (size_t)3 * c->H * c->W;
    if (!p.copy) {
        HIP_TRY(hipStreamCreateWithFlags(&p.copy, hipStreamNonBlocking));
        for (int i = 0; i < st_ctx::Pipe::kSlots; ++i) {
            HIP_TRY(hipEventCreateWithFlags(&p.ready[i], hipEventDisableTiming));
            HIP_TRY(hipEventCreateWithFlags(&p.done[i], hipEventDisableTiming));
            HIP_TRY(hipHostMalloc((void**)&p.trace_pin[i], (kMaxTraceLayers * 6 + 8) * sizeof(float), 0));
        }
    }
    if (n3 > p.cap) {
        if (p.count) return fail(ST_ERR_STATE, "the input grew while an iteration is in flight: st_step_end first");
        for (int i = 0; i < st_ctx::Pipe::kSlots; ++i) {
            dfree(p.hwc[i]);
            if (p.img_pin[i]) { (void)hipHostFree(p.img_pin[i]); p.img_pin[i] = nullptr; }
            ST_TRY(dmalloc(&p.hwc[i], n3));
            HIP_TRY(hipHostMalloc((void**)&p.img_pin[i], n3 * sizeof(float), 0));
        }
        p.cap = n3;
    }
    ST_TRY(step_enqueue(c));
    const int slot = (int)((p.head + p.count) % st_ctx::Pipe::kSlots);
    { ProfScope ps(c, P_MISC, 0, 0); HIP_TRY(launch_deprocess(c->x[c->cur], p.hwc[slot], c->H, c->W, c->stream)); }
    p.tlen[slot] = c->trace_len_last; p.H[slot] = c->H; p.W[slot] = c->W;
    HIP_TRY(hipMemcpyAsync(p.trace_pin[slot], c->trace_dev, p.tlen[slot] * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipEventRecord(p.ready[slot], c->stream));
    HIP_TRY(hipStreamWaitEvent(p.copy, p.ready[slot], 0));
    HIP_TRY(hipMemcpyAsync(p.img_pin[slot], p.hwc[slot], n3 * sizeof(float), hipMemcpyDeviceToHost, p.copy));
    HIP_TRY(hipEventRecord(p.done[slot], p.copy));
    p.count += 1;
    return ST_OK;
}

int st_step_pending(st_ctx* c) { return c ? c->pipe.count : 0; }

int st_step_end(st_ctx* c, const float** out_hwc, int* out_h, int* out_w, double* trace, float* out_loss)
{
    if (!c) return fail(ST_ERR_ARG, "ctx is NULL");
    st_ctx::Pipe& p = c->pipe;
    if (!p.count) return fail(ST_ERR_STATE, "no iteration in flight: st_step_begin first");
    HIP_TRY(hipSetDevice(c->device));
    const int slot = (int)(p.head % st_ctx::Pipe::kSlots);
    HIP_TRY(hipEventSynchronize(p.done[slot]));          // (the trace copy precedes `ready`, which precedes `done`)
    const int n = p.tlen[slot];
    if (trace) for (int i = 0; i < n; ++i) trace[i] = p.trace_pin[slot][i];
    c->last_loss = p.trace_pin[slot][n - 2];
    if (out_loss) *out_loss = c->last_loss;
    if (out_hwc) *out_hwc = p.img_pin[slot];             // valid for the next kSlots - 1 calls of st_step_begin
    if (out_h) *out_h = p.H[slot];
    if (out_w) *out_w = p.W[slot];
    p.head += 1; p.count -= 1;
    return ST_OK;
}

// Test hook: p = inv_hv(g) of optimizers.py:89-108 for a GIVEN history, run by the same device two-loop the optimizer
// uses (lbfgs.hip), without applying the update.  pairs are oldest first; each must pass the s.y > 1e-10 gate.
// The optimizer's own history is replaced: the next L-BFGS step starts from an empty one.
int st_lbfgs_inv_hv(st_ctx* c, int n_pairs, const float* const* s_vecs, const float* const* y_vecs, const float* g, float* out_p)
{
    if (c) c->epoch++;
    if (!c || !g || !out_p || n_pairs < 0 || n_pairs > kLbfgsCorr || (n_pairs && (!s_vecs || !y_vecs))) return fail(ST_ERR_ARG, "bad argument");
    if (!c->x[0]) return fail(ST_ERR_STATE, "no input image (it fixes the vector length)");
    HIP_TRY(hipSetDevice(c->device));
    ST_TRY(lbfgs_alloc(c));
    const size_t bytes = (size_t)3 * c->H * c->W * sizeof(float);
    hipStream_t st = c->stream;
    HIP_TRY(hipMemsetAsync(c->lb_dev, 0, sizeof(LbfgsDev), st));
    c->lb_clear = true; c->have_cur = false;
    for (int k = 0; k < n_pairs; ++k) {        // an empty ring hands out slots 0, 1, 2, ... while every pair is kept
        HIP_TRY(hipMemcpyAsync(c->hs[k], s_vecs[k], bytes, hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync(c->hy[k], y_vecs[k], bytes, hipMemcpyHostToDevice, st));
        HIP_TRY(launch_lbfgs_pair(lbfgs_args(c, 0), nullptr, 1, st));
    }
    HIP_TRY(hipMemcpyAsync(c->g_cur, g, bytes, hipMemcpyHostToDevice, st));
    LbfgsDev host{};
    HIP_TRY(hipMemcpyAsync(&host, c->lb_dev, sizeof host, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (host.count != n_pairs) return fail(ST_ERR_ARG, "%d of %d pairs failed the s.y > 1e-10 gate", n_pairs - host.count, n_pairs);
    HIP_TRY(launch_lbfgs_two_loop(lbfgs_args(c, 0), st));
    HIP_TRY(hipMemcpyAsync(out_p, c->pvec, bytes, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return ST_OK;
}

int st_graph_replays(st_ctx* c, long long* n)
{
    if (!c || !n) return fail(ST_ERR_ARG, "bad argument");
    *n = c->graph_replays;
    return ST_OK;
}

int st_sync(st_ctx* c)
{
    if (!c) return fail(ST_ERR_ARG, "ctx is NULL");
    HIP_TRY(hipStreamSynchronize(c->stream));
    return ST_OK;
}

// ---- device-resident resampling (Pillow-exact; utils.py:130-160) -----------------------------------------------
struct DevTable { int* lo = nullptr; int* n = nullptr; double* k = nullptr; ResampleTable t{}; int out = 0; };

static int table_upload(const st_resample_table* h, DevTable* d)
{
    if (!h || !h->lo || !h->n || !h->k || h->kmax <= 0 || h->out_size <= 0) return fail(ST_ERR_ARG, "bad resample table");
    const size_t no = (size_t)h->out_size;
    HIP_TRY(hipMalloc((void**)&d->lo, no * sizeof(int)));
    HIP_TRY(hipMalloc((void**)&d->n, no * sizeof(int)));
    HIP_TRY(hipMalloc((void**)&d->k, no * h->kmax * sizeof(double)));
    HIP_TRY(hipMemcpy(d->lo, h->lo, no * sizeof(int), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(d->n, h->n, no * sizeof(int), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(d->k, h->k, no * h->kmax * sizeof(double), hipMemcpyHostToDevice));
    d->t = ResampleTable{d->lo, d->n, d->k, h->kmax};
    d->out = h->out_size;
    return ST_OK;
}
static void table_free(DevTable* d)
{
    if (d->lo) (void)hipFree(d->lo);
    if (d->n) (void)hipFree(d->n);
    if (d->k) (void)hipFree(d->k);
    *d = DevTable{};
}
static int tables_valid_for(const st_resample_table* x, const st_resample_table* y, int H, int W)
{
    for (int i = 0; x && i < x->out_size; ++i) if (x->lo[i] < 0 || x->n[i] < 0 || x->lo[i] + x->n[i] > W || x->n[i] > x->kmax) return 0;
    for (int i = 0; y && i < y->out_size; ++i) if (y->lo[i] < 0 || y->n[i] < 0 || y->lo[i] + y->n[i] > H || y->n[i] > y->kmax) return 0;
    return 1;
}

int st_resample_state(st_ctx* c, const st_resample_table* lan_x, const st_resample_table* lan_y,
                      const st_resample_table* bil_x, const st_resample_table* bil_y, const float* new_x_nchw)
{
    if (c) c->epoch++;       // anything but st_step may change what a step launches: captured step graphs are stale
    if (!c || !c->x[0]) return fail(ST_ERR_STATE, "no input image");
    if (!lan_x || !lan_y) return fail(ST_ERR_ARG, "Lanczos tables are required");
    HIP_TRY(hipSetDevice(c->device));
    const int H = c->H, W = c->W, H2 = lan_y->out_size, W2 = lan_x->out_size;
    const bool adam = c->opt_kind == ST_OPT_ADAM;
    if (adam && (!bil_x || !bil_y || bil_x->out_size != W2 || bil_y->out_size != H2)) return fail(ST_ERR_ARG, "Adam needs bilinear tables of the same output size");
    if (!tables_valid_for(lan_x, lan_y, H, W) || (adam && !tables_valid_for(bil_x, bil_y, H, W))) return fail(ST_ERR_ARG, "resample table does not fit the %dx%d state", H, W);
    HIP_TRY(hipStreamSynchronize(c->stream));
    DevTable lx, ly, bx, by;
    int rc = table_upload(lan_x, &lx);
    if (rc == ST_OK) rc = table_upload(lan_y, &ly);
    if (rc == ST_OK && adam) rc = table_upload(bil_x, &bx);
    if (rc == ST_OK && adam) rc = table_upload(bil_y, &by);
    const size_t n2 = (size_t)3 * H2 * W2, ntmp = (size_t)3 * H * W2;
    float *tx = nullptr, *tm = nullptr, *tv = nullptr, *tmp = nullptr;
    if (rc == ST_OK) rc = dmalloc(&tx, n2);
    if (rc == ST_OK) rc = dmalloc(&tmp, ntmp);
    const bool keep_m = adam && !c->m_zero, keep_v = adam && !c->v_zero;
    if (rc == ST_OK && keep_m) rc = dmalloc(&tm, n2);
    if (rc == ST_OK && keep_v) rc = dmalloc(&tv, n2);
    auto hip_ok = [&](hipError_t e, const char* what) { if (e != hipSuccess && rc == ST_OK) rc = fail(ST_ERR_HIP, "%s: %s", what, hipGetErrorString(e)); };
    if (rc == ST_OK) {
        if (new_x_nchw) hip_ok(hipMemcpyAsync(tx, new_x_nchw, n2 * sizeof(float), hipMemcpyHostToDevice, c->stream), "new x upload");
        else hip_ok(launch_resample(c->x[c->cur], tmp, tx, 3, H, W, H2, W2, lx.t, ly.t, 0, c->stream), "resample x");
        if (keep_m) hip_ok(launch_resample(c->m, tmp, tm, 3, H, W, H2, W2, lx.t, ly.t, 0, c->stream), "resample m");
        if (keep_v) hip_ok(launch_resample(c->v, tmp, tv, 3, H, W, H2, W2, bx.t, by.t, 1, c->stream), "resample v");   // np.maximum(0, .)
        hip_ok(hipStreamSynchronize(c->stream), "resample sync");
    }
    const bool mz = c->m_zero, vz = c->v_zero;
    const int i1 = c->items1, i2 = c->items2;
    if (rc == ST_OK) rc = ensure_input_buffers(c, H2, W2);            // frees and re-creates x, m, v, L-BFGS vectors
    if (rc == ST_OK) {
        c->lb_clear = true; c->have_cur = false;
        hip_ok(hipMemcpyAsync(c->x[c->cur], tx, n2 * sizeof(float), hipMemcpyDeviceToDevice, c->stream), "x copy");
        if (keep_m) hip_ok(hipMemcpyAsync(c->m, tm, n2 * sizeof(float), hipMemcpyDeviceToDevice, c->stream), "m copy");
        if (keep_v) hip_ok(hipMemcpyAsync(c->v, tv, n2 * sizeof(float), hipMemcpyDeviceToDevice, c->stream), "v copy");
        hip_ok(hipStreamSynchronize(c->stream), "copy sync");
        c->m_zero = mz; c->v_zero = vz; c->items1 = i1; c->items2 = i2;
    }
    dfree(tx); dfree(tm); dfree(tv); dfree(tmp);
    table_free(&lx); table_free(&ly); table_free(&bx); table_free(&by);
    return rc;
}

int st_resample_content(st_ctx* c, const st_resample_table* lan_x, const st_resample_table* lan_y)
{
    if (c) c->epoch++;       // anything but st_step may change what a step launches: captured step graphs are stale
    if (!c || !c->have_content || !c->content_x) return fail(ST_ERR_STATE, "no content image");
    if (!lan_x || !lan_y) return fail(ST_ERR_ARG, "Lanczos tables are required");
    HIP_TRY(hipSetDevice(c->device));
    const int H = c->cH, W = c->cW, H2 = lan_y->out_size, W2 = lan_x->out_size;
    if (!tables_valid_for(lan_x, lan_y, H, W)) return fail(ST_ERR_ARG, "resample table does not fit the %dx%d content", H, W);
    DevTable lx, ly;
    int rc = table_upload(lan_x, &lx);
    if (rc == ST_OK) rc = table_upload(lan_y, &ly);
    float *tx = nullptr, *tmp = nullptr;
    if (rc == ST_OK) rc = dmalloc(&tx, (size_t)3 * H2 * W2);
    if (rc == ST_OK) rc = dmalloc(&tmp, (size_t)3 * H * W2);
    if (rc == ST_OK && launch_resample(c->content_x, tmp, tx, 3, H, W, H2, W2, lx.t, ly.t, 0, c->stream) != hipSuccess) rc = fail(ST_ERR_HIP, "content resample failed");
    if (rc == ST_OK && hipStreamSynchronize(c->stream) != hipSuccess) rc = fail(ST_ERR_HIP, "content resample sync failed");
    if (rc == ST_OK) rc = content_from_device(c, tx, H2, W2);
    (void)hipStreamSynchronize(c->stream);
    dfree(tx); dfree(tmp);
    table_free(&lx); table_free(&ly);
    return rc;
}

int st_get_content_nchw(st_ctx* c, float* out, int* H, int* W)
{
    if (!c || !c->have_content || !c->content_x) return fail(ST_ERR_STATE, "no content image");
    if (H) *H = c->cH;
    if (W) *W = c->cW;
    if (out) {
        HIP_TRY(hipStreamSynchronize(c->stream));
        HIP_TRY(hipMemcpy(out, c->content_x, (size_t)3 * c->cH * c->cW * sizeof(float), hipMemcpyDeviceToHost));
    }
    return ST_OK;
}

// ---- tile-sharded single image (style_transfer2_amd/tiling.py has the design) ----------------------------------
// This context runs the window (tile + apron) of one rank.  Every reduction is restricted to the tile's region
// of each blob and left UN-normalised in a flat device buffer that the caller all-reduces (RCCL) between phases.
struct BlobRoi { int y0, x0, y1, x1; double n_global; };

static BlobRoi tile_roi(const st_ctx* c, int b)
{
    const st_ctx::Tile& t = c->tile;
    int s = 1, gh = t.gH, gw = t.gW;
    for (int i = 1; i <= b; ++i)
        if (!c->topo[i - 1].is_conv) { s *= 2; gh = pooled_size(gh); gw = pooled_size(gw); }
    const ActSet& a = c->act;
    BlobRoi r;
    r.y0 = (t.ty0 - t.wy0) / s; r.x0 = (t.tx0 - t.wx0) / s;
    r.y1 = t.ty1 == t.gH ? a.h[b] : (t.ty1 - t.wy0) / s;
    r.x1 = t.tx1 == t.gW ? a.w[b] : (t.tx1 - t.wx0) / s;
    r.n_global = (double)a.C[b] * gh * gw;
    return r;
}

int st_tile_configure(st_ctx* c, int gH, int gW, int wy0, int wx0, int ty0, int tx0, int ty1, int tx1)
{
    if (c) c->epoch++;       // anything but st_step may change what a step launches: captured step graphs are stale
    if (!c || !c->x[0]) return fail(ST_ERR_STATE, "set the window image first (st_set_input)");
    if (wy0 < 0 || wx0 < 0 || wy0 + c->H > gH || wx0 + c->W > gW || ty0 < wy0 || tx0 < wx0 ||
        ty1 > wy0 + c->H || tx1 > wx0 + c->W || ty1 <= ty0 || tx1 <= tx0)
        return fail(ST_ERR_ARG, "tile/window geometry is inconsistent with the %dx%d window", c->H, c->W);
    st_ctx::Tile& t = c->tile;
    t.on = true; t.gH = gH; t.gW = gW; t.wy0 = wy0; t.wx0 = wx0; t.ty0 = ty0; t.tx0 = tx0; t.ty1 = ty1; t.tx1 = tx1;
    return ST_OK;
}

static int tile_ensure(float** p, size_t* cap, size_t n)
{
    if (n > *cap) { dfree(*p); ST_TRY(dmalloc(p, n)); *cap = n; }
    return ST_OK;
}

// phase 1: forward on the window, region sums [d2, gc2, F2, gd2] per active layer and the RAW Gram sums
int st_tile_forward(st_ctx* c, float** dev_ptr, int* n_floats)
{
    if (c) c->epoch++;       // anything but st_step may change what a step launches: captured step graphs are stale
    if (!c || !c->tile.on) return fail(ST_ERR_STATE, "st_tile_configure first");
    HIP_TRY(hipSetDevice(c->device));
    ST_TRY(act_ensure(c, c->act, c->H, c->W));
    ActSet& a = c->act;
    int last = 0;
    size_t n1 = 0;
    for (const ActiveLayer& al : c->active) {
        last = std::max(last, al.blob);
        n1 += 4 + (al.s ? (size_t)a.C[al.blob] * a.C[al.blob] : 0);
        if (al.c && (!c->have_content || c->cH != c->H || c->cW != c->W)) return fail(ST_ERR_STATE, "content features missing");
        if (al.s && !c->have_style) return fail(ST_ERR_STATE, "style Gram matrices missing");
    }
    ST_TRY(tile_ensure(&c->tile.p1, &c->tile.p1_n, std::max<size_t>(n1, 1)));
    HIP_TRY(hipMemsetAsync(c->tile.p1, 0, std::max<size_t>(n1, 1) * sizeof(float), c->stream));
    ST_TRY(forward_range(c, a, c->x[c->cur], last));
    size_t pos = 0;
    for (const ActiveLayer& al : c->active) {
        const int b = al.blob, C = a.C[b];
        const BlobRoi r = tile_roi(c, b);
        const size_t n = (size_t)C * a.h[b] * a.w[b];
        if (!c->layer_part[b]) ST_TRY(dmalloc(&c->layer_part[b], 5 * kMaxPartials));
        float* part = c->layer_part[b];
        if (al.c || al.d) {
            LayerElemArgs e{};
            e.feat = a.data[b]; e.target = al.c ? c->content_feat[b] : nullptr; e.n = n;
            e.cn_coef = (float)(2.0 / r.n_global); e.dn_coef = (float)(-2.0 / r.n_global);
            e.content = al.c; e.deepdream = al.d; e.write = 0;
            e.part_d2 = part; e.part_gc2 = part + kMaxPartials; e.part_f2 = part + 2 * kMaxPartials; e.part_gd2 = part + 3 * kMaxPartials;
            e.h = a.h[b]; e.w = a.w[b]; e.ry0 = r.y0; e.rx0 = r.x0; e.ry1 = r.y1; e.rx1 = r.x1;
            int np = 0;
            HIP_TRY(launch_layer_elem(e, &np, c->stream));
            for (int k = 0; k < 4; ++k) HIP_TRY(launch_sum_partials(part + k * kMaxPartials, np, c->tile.p1 + pos + k, c->stream));
        }
        pos += 4;
        if (al.s) {
            const int rw = r.x1 - r.x0, rh = r.y1 - r.y0, hw = rw * rh;
            GramPlan pl;
            ST_TRY(ensure_gram_bufs(c, C, hw, pl));
            GramRoi roi{r.y0, r.x0, rw, a.w[b], (size_t)a.h[b] * a.w[b]};
            HIP_TRY(launch_gram_partial(a.data[b], c->gram_slabs, C, hw, pl, c->stream, &roi));
            // raw sum over this rank's region (divisor 1, no target), contiguous C x C
            HIP_TRY(launch_gram_reduce(c->gram_slabs, c->gram_fold, nullptr, c->tile.p1 + pos, C, nullptr, nullptr, C, 1.0, pl, c->stream));
            pos += (size_t)C * C;
        }
    }
    if (dev_ptr) *dev_ptr = c->tile.p1;
    if (n_floats) *n_floats = (int)n1;
    return ST_OK;
}

// phase 2a: after the all-reduce of p1.  Captures missing content / deep-dream norms.  If a style norm is still
// missing (first evaluation after reset) it returns the p2 buffer (n > 0): the caller then runs st_tile_style_raw,
// all-reduces p2 and only then calls st_tile_losses_finish.
int st_tile_losses(st_ctx* c, float** dev_ptr, int* n_floats)
{
    if (c) c->epoch++;       // anything but st_step may change what a step launches: captured step graphs are stale
    if (!c || !c->tile.on) return fail(ST_ERR_STATE, "st_tile_configure first");
    HIP_TRY(hipSetDevice(c->device));
    ActSet& a = c->act;
    int n_style = 0;
    bool missing = false;
    for (const ActiveLayer& al : c->active) if (al.s) { ++n_style; missing = missing || !c->norm_valid[al.blob * 3 + 1]; }
    ST_TRY(tile_ensure(&c->tile.p2, &c->tile.p2_n, std::max(n_style, 1)));
    ST_TRY(tile_ensure(&c->tile.pd, &c->tile.pd_n, std::max(n_style, 1)));
    c->tile.s2_in_p2 = missing;
    if (!c->dbuf) {
        size_t cc = 1;
        for (int i = 0; i < c->nb; ++i) cc = std::max(cc, (size_t)a.C[i] * conv_mpad(a.C[i]));
        ST_TRY(dmalloc(&c->dbuf, cc));
        HIP_TRY(hipMemsetAsync(c->dbuf, 0, cc * sizeof(float), c->stream));
    }
    size_t pos = 0;
    for (const ActiveLayer& al : c->active) {
        const int b = al.blob, C = a.C[b];
        const BlobRoi r = tile_roi(c, b);
        float* nrm = c->norms + b * 3;
        if (al.c && !c->norm_valid[b * 3 + 0]) { HIP_TRY(launch_finalize_norm(c->tile.p1 + pos + 1, 1, r.n_global, nrm + 0, c->stream)); c->norm_valid[b * 3 + 0] = 1; }
        if (al.d && !c->norm_valid[b * 3 + 2]) { HIP_TRY(launch_finalize_norm(c->tile.p1 + pos + 3, 1, r.n_global, nrm + 2, c->stream)); c->norm_valid[b * 3 + 2] = 1; }
        pos += 4;
        if (al.s) pos += (size_t)C * C;
    }
    if (dev_ptr) *dev_ptr = missing ? c->tile.p2 : nullptr;
    if (n_floats) *n_floats = missing ? n_style : 0;
    return ST_OK;
}

}  // extern "C" (tile helpers continue below)

// D = Graw / n_global - G_style into dbuf ([C][MPad]); sum D^2 -> pd[k]
static int tile_style_D(st_ctx* c, int b, const float* graw, double n_global, float* pd_slot)
{
    const ActSet& a = c->act;
    const int C = a.C[b];
    if (!c->layer_part[b]) ST_TRY(dmalloc(&c->layer_part[b], 5 * kMaxPartials));
    float* part = c->layer_part[b] + 4 * kMaxPartials;
    int np = 0;
    GramPlan one{}; one.splits = 1;
    HIP_TRY(launch_gram_reduce(graw, nullptr, c->style_gram[b], c->dbuf, conv_mpad(C), part, &np, C, n_global, one, c->stream));
    HIP_TRY(launch_sum_partials(part, np, pd_slot, c->stream));
    return ST_OK;
}

extern "C" {

// phase 2b: injected diffs of every active layer (region only, zero elsewhere)
int st_tile_losses_finish(st_ctx* c)
{
    if (c) c->epoch++;       // anything but st_step may change what a step launches: captured step graphs are stale
    if (!c || !c->tile.on) return fail(ST_ERR_STATE, "st_tile_configure first");
    HIP_TRY(hipSetDevice(c->device));
    ActSet& a = c->act;
    size_t pos = 0;
    int k = 0;
    const bool two_step = c->tile.s2_in_p2;
    for (const ActiveLayer& al : c->active) {
        const int b = al.blob, C = a.C[b];
        const BlobRoi r = tile_roi(c, b);
        const size_t n = (size_t)C * a.h[b] * a.w[b];
        if (!c->inject[b]) ST_TRY(dmalloc(&c->inject[b], n));
        float* nrm = c->norms + b * 3;
        bool wrote = false;
        if (al.c || al.d) {
            LayerElemArgs e{};
            e.feat = a.data[b]; e.target = al.c ? c->content_feat[b] : nullptr; e.inject = c->inject[b]; e.n = n;
            e.cn_coef = (float)(2.0 / r.n_global); e.dn_coef = (float)(-2.0 / r.n_global);
            e.cw = al.cw; e.dw = al.dw; e.content = al.c; e.deepdream = al.d; e.write = 1;
            e.norm_c = nrm + 0; e.norm_d = nrm + 2;
            e.h = a.h[b]; e.w = a.w[b]; e.ry0 = r.y0; e.rx0 = r.x0; e.ry1 = r.y1; e.rx1 = r.x1;
            int np = 0;
            HIP_TRY(launch_layer_elem(e, &np, c->stream));
            wrote = true;
        }
        pos += 4;
        if (al.s) {
            ST_TRY(tile_style_D(c, b, c->tile.p1 + pos, r.n_global, c->tile.pd + k));
            const float c2 = (float)(2.0 / ((double)C * C * r.n_global));
            const int need = style_grad_blocks(C, a.h[b], a.w[b]);
            if (c->s2_cap[b] < need) { dfree(c->s2_part[b]); ST_TRY(dmalloc(&c->s2_part[b], need)); c->s2_cap[b] = need; }
            PixRoi pr{r.y0, r.x0, r.y1, r.x1};
            int np = 0;
            if (two_step) {
                // norm from the all-reduced sum S^2 of the first pass (st_tile_style_raw), then saxpy
                if (!c->norm_valid[b * 3 + 1]) {
                    HIP_TRY(launch_finalize_norm(c->tile.p2 + k, 1, r.n_global, nrm + 1, c->stream));
                    c->norm_valid[b * 3 + 1] = 1;
                }
                HIP_TRY(launch_style_grad(c->dbuf, a.data[b], c->inject[b], c2, 1, al.sw, nrm + 1, wrote, c->s2_part[b], &np, C, a.h[b], a.w[b], c->stream, &pr));
            } else {
                HIP_TRY(launch_style_grad(c->dbuf, a.data[b], c->inject[b], c2, 1, al.sw, nrm + 1, wrote, c->s2_part[b], &np, C, a.h[b], a.w[b], c->stream, &pr));
                // sum S^2 of this rank's region -> p3 tail (all-reduced with the image sums)
                ST_TRY(tile_ensure(&c->tile.p3, &c->tile.p3_n, 6 + kMaxTraceLayers));
                HIP_TRY(launch_sum_partials(c->s2_part[b], np, c->tile.p3 + 6 + k, c->stream));
            }
            pos += (size_t)C * C;
            ++k;
        }
    }
    return ST_OK;
}

// first evaluation only: unscaled style gradients, sum S^2 per style layer -> p2 (to be all-reduced)
int st_tile_style_raw(st_ctx* c)
{
    if (c) c->epoch++;       // anything but st_step may change what a step launches: captured step graphs are stale
    if (!c || !c->tile.on) return fail(ST_ERR_STATE, "st_tile_configure first");
    HIP_TRY(hipSetDevice(c->device));
    ActSet& a = c->act;
    size_t pos = 0;
    int k = 0;
    for (const ActiveLayer& al : c->active) {
        const int b = al.blob, C = a.C[b];
        pos += 4;
        if (!al.s) continue;
        const BlobRoi r = tile_roi(c, b);
        ST_TRY(tile_style_D(c, b, c->tile.p1 + pos, r.n_global, c->tile.pd + k));
        const float c2 = (float)(2.0 / ((double)C * C * r.n_global));
        const int need = style_grad_blocks(C, a.h[b], a.w[b]);
        if (c->s2_cap[b] < need) { dfree(c->s2_part[b]); ST_TRY(dmalloc(&c->s2_part[b], need)); c->s2_cap[b] = need; }
        if (!c->stmp) ST_TRY(dmalloc(&c->stmp, c->max_blob));
        PixRoi pr{r.y0, r.x0, r.y1, r.x1};
        int np = 0;
        HIP_TRY(launch_style_grad(c->dbuf, a.data[b], c->stmp, c2, 0, al.sw, c->norms + b * 3 + 1, 0, c->s2_part[b], &np, C, a.h[b], a.w[b], c->stream, &pr));
        HIP_TRY(launch_sum_partials(c->s2_part[b], np, c->tile.p2 + k, c->stream));
        pos += (size_t)C * C;
        ++k;
    }
    return ST_OK;
}

// phase 3: ranged backward on the window; returns the device pointer of the (3, wh, ww) gradient
int st_tile_backward(st_ctx* c, float** dev_grad)
{
    if (c) c->epoch++;       // anything but st_step may change what a step launches: captured step graphs are stale
    if (!c || !c->tile.on) return fail(ST_ERR_STATE, "st_tile_configure first");
    HIP_TRY(hipSetDevice(c->device));
    const size_t n3 = (size_t)3 * c->H * c->W;
    if (!c->tile.wgrad) ST_TRY(dmalloc(&c->tile.wgrad, n3));
    std::vector<const float*> inj(c->nb, nullptr);
    int last = -1;
    for (const ActiveLayer& al : c->active) { inj[al.blob] = c->inject[al.blob]; last = std::max(last, al.blob); }
    if (last < 0) HIP_TRY(hipMemsetAsync(c->tile.wgrad, 0, n3 * sizeof(float), c->stream));
    else {
        if (!c->diffA) { ST_TRY(dmalloc(&c->diffA, c->max_blob)); ST_TRY(dmalloc(&c->diffB, c->max_blob)); }
        const float* g = inj[0];
        if (last > 0) ST_TRY(backward_chain(c, last, inj[last], inj, &g));
        HIP_TRY(hipMemcpyAsync(c->tile.wgrad, g, n3 * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
    }
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (dev_grad) *dev_grad = c->tile.wgrad;
    return ST_OK;
}

// phase 4: TV + p-norm + combine + Adam on the tile; ring = (3, th+2, tw+2) device tensor.  The 6 image sums
// go to p3[0..6) (p3[6..) already holds this rank's sum S^2 per style layer in steady state).
int st_tile_update(st_ctx* c, const float* ring_dev, float** dev_ptr, int* n_floats)
{
    if (c) c->epoch++;       // anything but st_step may change what a step launches: captured step graphs are stale
    if (!c || !c->tile.on || !ring_dev) return fail(ST_ERR_STATE, "st_tile_configure first");
    if (c->opt_kind != ST_OPT_ADAM) return fail(ST_ERR_STATE, "the tile-sharded mode implements Adam");
    HIP_TRY(hipSetDevice(c->device));
    const st_ctx::Tile& t = c->tile;
    ST_TRY(tile_ensure(&c->tile.p3, &c->tile.p3_n, 6 + kMaxTraceLayers));
    c->items1 += 1; c->items2 += 1;
    ImageTileArgs ta{};
    ImagePassArgs& ip = ta.base;
    ip.x = c->x[c->cur]; ip.scd = c->tile.wgrad; ip.grad = nullptr; ip.C = 3; ip.H = c->H; ip.W = c->W;
    ip.tv_w = c->tv_w; ip.tv_beta = c->tv_pow; ip.p_w = c->p_w; ip.p_pow = c->p_pow; ip.partial = c->image_part;
    ip.x_out = c->x[c->cur ^ 1]; ip.m = c->m; ip.v = c->v;
    ip.d1 = (float)0.9; ip.c1 = (float)(1 - 0.9); ip.d2 = (float)0.999; ip.c2 = (float)(1 - 0.999);
    ip.corr1 = (float)(1 - pow(0.9, c->items1)); ip.corr2 = (float)(1 - pow(0.999, c->items2));
    ip.step = (float)c->step_size; ip.m_is_zero = c->m_zero; ip.v_is_zero = c->v_zero;
    ta.ring = ring_dev; ta.ty = t.ty0 - t.wy0; ta.tx = t.tx0 - t.wx0; ta.th = t.ty1 - t.ty0; ta.tw = t.tx1 - t.tx0;
    // the untouched apron of x_next is refreshed by the caller; start it from the current values
    HIP_TRY(hipMemcpyAsync(c->x[c->cur ^ 1], c->x[c->cur], (size_t)3 * c->H * c->W * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
    int np = 0;
    HIP_TRY(launch_image_pass_tile(ta, &np, c->stream));
    for (int k = 0; k < 6; ++k) HIP_TRY(launch_sum_partials(c->image_part + k * kMaxPartials, np, c->tile.p3 + k, c->stream));
    c->m_zero = c->v_zero = false;
    int n_style = 0;
    for (const ActiveLayer& al : c->active) n_style += al.s;
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (dev_ptr) *dev_ptr = c->tile.p3;
    if (n_floats) *n_floats = 6 + (c->tile.s2_in_p2 ? 0 : n_style);
    return ST_OK;
}

// The same image-space pass without an optimizer update: the combined gradient of the tile's pixels goes to the window-sized
// gradient buffer (st_tile_buffer 4), the partial sums come back as from st_tile_update.  The tile-sharded L-BFGS
// (style_transfer2_amd/tiled.py) evaluates the objective with it and owns the update itself.
int st_tile_gradient(st_ctx* c, const float* ring_dev, float** dev_ptr, int* n_floats)
{
    if (c) c->epoch++;
    if (!c || !c->tile.on || !ring_dev) return fail(ST_ERR_STATE, "st_tile_configure first");
    HIP_TRY(hipSetDevice(c->device));
    const st_ctx::Tile& t = c->tile;
    ST_TRY(tile_ensure(&c->tile.p3, &c->tile.p3_n, 6 + kMaxTraceLayers));
    ImageTileArgs ta{};
    ImagePassArgs& ip = ta.base;
    ip.x = c->x[c->cur]; ip.scd = c->tile.wgrad; ip.grad = c->grad; ip.C = 3; ip.H = c->H; ip.W = c->W;
    ip.tv_w = c->tv_w; ip.tv_beta = c->tv_pow; ip.p_w = c->p_w; ip.p_pow = c->p_pow; ip.partial = c->image_part;
    ip.x_out = nullptr;
    ta.ring = ring_dev; ta.ty = t.ty0 - t.wy0; ta.tx = t.tx0 - t.wx0; ta.th = t.ty1 - t.ty0; ta.tw = t.tx1 - t.tx0;
    int np = 0;
    HIP_TRY(launch_image_pass_tile(ta, &np, c->stream));
    for (int k = 0; k < 6; ++k) HIP_TRY(launch_sum_partials(c->image_part + k * kMaxPartials, np, c->tile.p3 + k, c->stream));
    int n_style = 0;
    for (const ActiveLayer& al : c->active) n_style += al.s;
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (dev_ptr) *dev_ptr = c->tile.p3;
    if (n_floats) *n_floats = 6 + (c->tile.s2_in_p2 ? 0 : n_style);
    return ST_OK;
}

// BLAS-1 pieces for a caller that keeps its own vectors on this device (the tile-sharded L-BFGS): out_dev[0] = sum a b over n
// elements (this rank's partial sum; fixed summation order), y = alpha x + y.  Synchronous with respect to the host.
int st_vec_dot(st_ctx* c, const float* a_dev, const float* b_dev, long long n, float* out_dev)
{
    if (!c || !a_dev || !b_dev || !out_dev || n < 0) return fail(ST_ERR_ARG, "bad argument");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(launch_vec_dot(a_dev, b_dev, (size_t)n, c->image_part, out_dev, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return ST_OK;
}

int st_vec_axpy(st_ctx* c, float alpha, const float* x_dev, float* y_dev, long long n)
{
    if (!c || !x_dev || !y_dev || n < 0) return fail(ST_ERR_ARG, "bad argument");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(launch_vec_axpy(alpha, x_dev, y_dev, (size_t)n, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return ST_OK;
}

// device pointers of the buffers the caller exchanges: which = 0 current x, 1 next x, 2 local sum D^2 per style layer
int st_tile_buffer(st_ctx* c, int which, float** dev_ptr)
{
    if (c) c->epoch++;       // anything but st_step may change what a step launches: captured step graphs are stale
    if (!c || !dev_ptr) return fail(ST_ERR_ARG, "bad argument");
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (which == 0) *dev_ptr = c->x[c->cur];
    else if (which == 1) *dev_ptr = c->x[c->cur ^ 1];
    else if (which == 2) *dev_ptr = c->tile.pd;
    else if (which == 3) *dev_ptr = c->norms;
    else if (which == 4) *dev_ptr = c->grad;
    else return fail(ST_ERR_ARG, "unknown buffer %d", which);
    return ST_OK;
}

// Pack (mode 0) the rectangles `rects` ([n][4] = y0, x0, h, w in window coordinates) of the (C, wh, ww) device tensor into
// the contiguous device buffer `buf`, or unpack them from it (mode 1 assign, 2 add): one launch per neighbour and phase.
int st_tile_strips(st_ctx* c, void* tensor_dev, int C, int wh, int ww, int n, const int* rects, void* buf_dev, int mode)
{
    if (!c || !tensor_dev || !buf_dev || n < 0 || n > kMaxStripRects || (n && !rects) || mode < 0 || mode > 2) return fail(ST_ERR_ARG, "bad argument");
    HIP_TRY(hipSetDevice(c->device));
    StripTable t{};
    t.n = n;
    int total = 0;
    for (int r = 0; r < n; ++r) {
        t.y0[r] = rects[4 * r]; t.x0[r] = rects[4 * r + 1]; t.h[r] = rects[4 * r + 2]; t.w[r] = rects[4 * r + 3];
        if (t.y0[r] < 0 || t.x0[r] < 0 || t.h[r] <= 0 || t.w[r] <= 0 || t.y0[r] + t.h[r] > wh || t.x0[r] + t.w[r] > ww)
            return fail(ST_ERR_ARG, "strip %d lies outside the %dx%d window", r, wh, ww);
        t.off[r] = total;
        total += C * t.h[r] * t.w[r];
    }
    t.total = total;
    HIP_TRY(launch_strip_copy((float*)tensor_dev, (float*)buf_dev, t, C, wh, ww, mode, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return ST_OK;
}

int st_tile_swap(st_ctx* c)
{
    if (c) c->epoch++;       // anything but st_step may change what a step launches: captured step graphs are stale
    if (!c) return fail(ST_ERR_ARG, "ctx is NULL");
    c->cur ^= 1;
    return ST_OK;
}

// ---- measurement
int st_profile_enable(st_ctx* c, int on)
{
    if (c) c->epoch++;       // anything but st_step may change what a step launches: captured step graphs are stale
    if (!c) return fail(ST_ERR_ARG, "ctx is NULL");
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->prof_on = on != 0;
    c->prof.clear();
    c->ev_used = 0;
    return ST_OK;
}

int st_profile_num_classes(void) { return P_COUNT; }
const char* st_profile_class_name(int cls) { return (cls >= 0 && cls < P_COUNT) ? kProfNames[cls] : nullptr; }

int st_profile_read(st_ctx* c, long long* launches, double* ms, double* flops, double* bytes)
{
    if (!c) return fail(ST_ERR_ARG, "ctx is NULL");
    HIP_TRY(hipStreamSynchronize(c->stream));
    for (int i = 0; i < P_COUNT; ++i) {
        if (launches) launches[i] = 0;
        if (ms) ms[i] = 0;
        if (flops) flops[i] = 0;
        if (bytes) bytes[i] = 0;
    }
    for (const ProfRec& r : c->prof) {
        float t = 0.f;
        HIP_TRY(hipEventElapsedTime(&t, r.a, r.b));
        if (launches) launches[r.cls] += 1;
        if (ms) ms[r.cls] += t;
        if (flops) flops[r.cls] += r.flops;
        if (bytes) bytes[r.cls] += r.bytes;
    }
    c->prof.clear();
    c->ev_used = 0;
    return ST_OK;
}

}  // extern "C"
