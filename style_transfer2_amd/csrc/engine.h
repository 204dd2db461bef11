// Shared declarations of the engine translation units (engine*.cpp): the context behind include/st2.h, its helpers and the
// launch sequences (forward / ranged backward / objective / optimizer step) the C ABI entry points are built from.
#pragma once
#include "../../include/st2.h"
#include "st2_kernels.h"

#include <hip/hip_runtime.h>
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <memory>
#include <string>
#include <vector>

namespace st2e {
using namespace st2;

// ------------------------------------------------------------------------------------------ errors
int fail(int code, const char* fmt, ...);
#define HIP_TRY(expr)                                                                               \
    do {                                                                                            \
        hipError_t e_ = (expr);                                                                     \
        if (e_ != hipSuccess)                                                                       \
            return st2e::fail(ST_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)
#define ST_TRY(expr)                    \
    do {                                \
        int r_ = (expr);                \
        if (r_ != ST_OK) return r_;     \
    } while (0)

// --------------------------------------------------------------------------------------- profiling
enum ProfClass { P_CONV_FWD, P_CONV_DGRAD, P_POOL_FWD, P_POOL_BWD, P_GRAM, P_GRAM_REDUCE, P_STYLE_GRAD,
                 P_LAYER_ELEM, P_IMAGE_PASS, P_FINALIZE, P_VECTOR, P_MISC, P_CONV_FWD_WINO, P_CONV_DGRAD_WINO,
                 P_CONV_FWD_BF16, P_CONV_DGRAD_BF16, P_COMM, P_GRAM_BF16, P_STYLE_GRAD_BF16, P_CONV_FWD_WSPLIT, P_CONV_DGRAD_WSPLIT, P_STYLE_FUSED_BF16, P_COUNT };
extern const char* const kProfNames[P_COUNT];
struct ProfRec { int cls; hipEvent_t a, b; double flops, bytes; };

// ------------------------------------------------------------------------------------------- types
struct Layer {
    bool is_conv = false;
    std::string name;
    int cin = 0, cout = 0;
    float *w_fwd = nullptr, *w_bwd = nullptr, *w_raw = nullptr, *w_raw_r = nullptr, *bias = nullptr;   // w_raw_r: w_raw rounded to bf16 values
    unsigned short *w16_fwd = nullptr, *w16_bwd = nullptr;       // bf16 packs (bf16 feature path)
    unsigned short* w_split = nullptr;                           // first layer, bf16 path: three-way bf16 split of the weights (conv3x3_first_split.hip)
    float *u_fwd = nullptr, *u_bwd = nullptr;                    // Winograd F(2x2,3x3) packs (null: not eligible)
    unsigned short *us_fwd = nullptr, *us_bwd = nullptr;         // split-operand Winograd packs (conv3x3_wino_split.hip; made when st_set_conv_algo(ctx, 2) asks for them)
    bool loaded = false;
};

struct ActSet {                    // activations of one forward geometry
    int H = 0, W = 0;
    std::vector<int> C, h, w;
    std::vector<float*> data;      // data[0] is borrowed (the image itself)
    std::vector<unsigned short*> data16;   // bf16 channel-blocked copies of the blobs that feed a bf16 conv
    std::vector<unsigned char*> amap;      // lean bf16 path: arg-max maps of the pools fused into the producing conv
    std::vector<char> has32, amap_ok;      // per blob: fp32 copy / arg-max map written by the last forward
    std::vector<unsigned short*> bits;     // bf16 lean flow: sign map of a conv blob that feeds a bf16 conv (Conv16Problem::bits_out) ...
    std::vector<char> bits_ok;             // ... written by the last forward: the data gradient above masks with it instead of the bf16 copy
    int valid_to = -1;
};

struct ActiveLayer { int blob; float cw, sw, dw; bool c, s, d; };

inline bool nonzero(float w) { return fabsf(w) > 1e-15f; }   // NaN compares false: worker.py:234
}  // namespace st2e

using namespace st2e;

struct st_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool bf16 = false;                             // conv operands in bf16 (BASELINE config 3)
    bool lean = false;                             // bf16 objective evaluations skip the fp32 tensors only bf16 convs would read
    bool in_step = false;                          // inside step_enqueue: objective evaluations may skip fp32 blobs nothing reads (lean fp32)
    bool wino = true;                              // Winograd F(2x2,3x3) for the eligible fp32 convs (ST2_WINO=0 disables)
    bool wino_split = false;                       // ... on the bf16 matrix cores with three-way split operands where the shape allows (st_set_conv_algo(ctx, 2))
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
    std::vector<char> inject_roi_zero;             // per blob: the inject buffer is zero outside the tile's region of interest (tile-sharded bf16 style term)
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
        char* pin_base[kSlots] = {};               // the pinned allocation of a slot: [head room | image | tail room]; img_pin points at the image
        hipEvent_t ready[kSlots] = {}, done[kSlots] = {};
        size_t cap = 0; long long head = 0; int count = 0, tlen[kSlots] = {}, H[kSlots] = {}, W[kSlots] = {};
        // bytes the caller may write in front of / behind an iterate handed out by st_step_end (a message frame around the image,
        // st_step_frame_room): requested, and what the current buffers were allocated with
        size_t want_head = 0, want_tail = 0, have_head = 0, have_tail = 0;
        // Buffers replaced by a re-allocation (the input grew, the frame room changed) stay alive until kSlots further begins have
        // passed: views handed out before it keep the documented lifetime.
        struct Retired { void* p; long long at; };
        std::vector<Retired> retired;
        long long begins = 0;
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
    LbfgsGram* lb_gram = nullptr;                  // Gram form (lbfgs.hip, second half): inner-product matrix + coefficients
    float* lb_gpart = nullptr;                     // [kLbGramRows][kMaxPartials] partial sums of the inner-product pass
    float* lb_dots = nullptr;                      // test hook: [kLbNB][kLbNB] pairwise inner products
    bool lb_gram_form = false;                     // form of the current history (decided while it is empty)
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
        bool fused = false;                        // inside st_tile_step: the phases do not synchronise the stream on their own
        // fused L-BFGS over the sharded image (engine_comm.cpp): this rank's tile of x as a compact (3, th, tw) vector, the sums of
        // one inner-product pass (all-reduced), and the global image size the unit-RMS first direction divides by
        float* lb_x = nullptr; float* lb_sums = nullptr; size_t lb_n = 0;
    } tile;
    // communicator of the tile-sharded mode (engine_comm.cpp): RCCL over xGMI, or caller-supplied transport functions (tests)
    struct Comm {
        void* lib = nullptr;                       // librccl.so, loaded on first use
        void* comm = nullptr;                      // ncclComm_t
        int rank = 0, world = 1;
        st_allreduce_fn ar = nullptr; st_exchange_fn ex = nullptr; void* user = nullptr;
        struct Peer { int peer = 0; std::vector<int> send, recv; float *sbuf = nullptr, *rbuf = nullptr; size_t sn = 0, rn = 0; };
        std::vector<Peer> plan[3];                 // per phase (ST_TILE_PLAN_*): the peers this rank exchanges strips with
        bool planned[3] = {false, false, false};
        bool self_via_rccl = false;                // test hook (ST2_COMM_SELF_VIA_RCCL=1): copies to the own rank travel through RCCL too
        float* ring = nullptr; size_t ring_cap = 0;
        long long steps = 0;
        std::vector<float> h1, h2, h3, hd, hn;     // host side of a traced step (sized once: no allocation per step)
        float *tile_chw = nullptr, *tile_hwc = nullptr; size_t tile_cap = 0;      // st_tile_get_tile's staging (kept across calls)
    } comm;
    // profiling
    bool prof_on = false;
    std::vector<ProfRec> prof;
    std::vector<hipEvent_t> ev_pool;
    size_t ev_used = 0;

    int blob_c(int i) const { return act.C[i]; }
};

namespace st2e {
// ---------------------------------------------------------------------------------------- helpers (engine.cpp)
int dmalloc(float** p, size_t nfloats);
void dfree(float*& p);
int dmalloc16(unsigned short** p, size_t n);
void dfree16(unsigned short*& p);
inline size_t act16_elems(int C, size_t hw) { return (size_t)((C + 7) / 8) * hw * 8; }
inline bool conv16_ok(const st_ctx* c, int K) { (void)c; return K >= 8 && K % 8 == 0; }
int wino_scratch(st_ctx* c, ConvProblem& p, bool split_kernel = false);      // room for the split-K partial sums of a Winograd launch that would otherwise leave most CUs idle

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

// work that rides on another class's launch (the style-gradient chunks fused into a bf16 data-gradient conv): flops booked to a class of
// their own with no time, so that the carrying class's flops stay the ones SURVEY 8(d) counts for it
inline void prof_note(st_ctx* c, int cls, double flops)
{
    if (!c->prof_on) return;
    if (c->ev_used == c->ev_pool.size()) { hipEvent_t e; (void)hipEventCreate(&e); c->ev_pool.push_back(e); }
    hipEvent_t e = c->ev_pool[c->ev_used++];
    (void)hipEventRecord(e, c->stream);
    c->prof.push_back(ProfRec{cls, e, e, flops, 0.0});
}

void shapes_for(const st_ctx* c, int H, int W, std::vector<int>& C, std::vector<int>& h, std::vector<int>& w);
void act_free(ActSet& a);
int act_ensure(st_ctx* c, ActSet& a, int H, int W);
bool blob_active(const st_ctx* c, int b);
bool style_runs16(const st_ctx* c, const ActSet& a, int b);
bool blob_needs32(const st_ctx* c, const ActSet& a, int b);
bool style_fuse_ok(const st_ctx* c, const ActSet& a, int b, int last);
int forward_range(st_ctx* c, ActSet& a, const float* x, int last, bool lean = false);
int ensure_gram_bufs(st_ctx* c, int C, int hw, GramPlan& pl, bool plan16 = false);
int gram_into(st_ctx* c, const float* F, int C, int hw, const float* target, float* out, int out_ld, float* partial, int* n_partial,
              const unsigned short* F16 = nullptr);
int backward_chain(st_ctx* c, int top, const float* top_diff, const std::vector<const float*>& inj, const float** out, bool lean = false);
int ensure_input_buffers(st_ctx* c, int H, int W);
int stage_upload(st_ctx* c, const void* host, size_t bytes);
int preprocess_into(st_ctx* c, const void* hwc, int H, int W, int is_u8, float* dst);
int set_input_common(st_ctx* c, int H, int W);
int content_from_device(st_ctx* c, const float* xdev, int H, int W);
int ensure_content_features(st_ctx* c);           // features of content-weighted blobs dropped by st_set_weights: recompute from the kept image
// ---------------------------------------------------------------------------------------- engine_objective.cpp
int eval_objective(st_ctx* c, const float* x, bool want_grad, float* grad_out, bool adam, float* x_next);
int read_trace(st_ctx* c, double* trace, float* loss);
// ---------------------------------------------------------------------------------------- engine_step.cpp
int lbfgs_alloc(st_ctx* c);
LbfgsArgs lbfgs_args(st_ctx* c, int apply);
// ---------------------------------------------------------------------------------------- engine_comm.cpp
void comm_free(st_ctx* c);
}  // namespace st2e
