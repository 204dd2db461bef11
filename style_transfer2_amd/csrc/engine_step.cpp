// Optimizers and the iteration: optimizers.py:7-125, StyleTransfer.step (worker.py:303-310), pipelined form, hipGraph replay.
#include "engine.h"

namespace st2e {
// ------------------------------------------------------------------------------------------ L-BFGS
int lbfgs_alloc(st_ctx* c)
{
    const size_t n3 = (size_t)3 * c->H * c->W;
    if (!c->g_cur) ST_TRY(dmalloc(&c->g_cur, n3));
    if (!c->pvec) ST_TRY(dmalloc(&c->pvec, n3));
    for (int i = 0; i <= st_ctx::kCorr; ++i) {
        if (!c->hs[i]) ST_TRY(dmalloc(&c->hs[i], n3));
        if (!c->hy[i]) ST_TRY(dmalloc(&c->hy[i], n3));
    }
    if (!c->lb_gram) {
        HIP_TRY(hipMalloc((void**)&c->lb_gram, sizeof(LbfgsGram)));
        HIP_TRY(hipMemset(c->lb_gram, 0, sizeof(LbfgsGram)));
    }
    if (!c->lb_gpart) ST_TRY(dmalloc(&c->lb_gpart, (size_t)kLbGramRows * kMaxPartials));
    return ST_OK;
}

// Gram form (two passes over the history per step) where the objective is approximate anyway -- the bf16 feature path --, the
// chain (the reference's operation order, fp32 axpy by axpy) otherwise; ST2_LBFGS_FORM=chain|gram overrides.  Decided while the
// history is empty: the two forms keep different state.
static bool lbfgs_wants_gram(const st_ctx* c)
{
    const char* e = getenv("ST2_LBFGS_FORM");
    if (e && *e) return e[0] == 'g' || e[0] == 'G';
    return c->bf16;
}

LbfgsArgs lbfgs_args(st_ctx* c, int apply)
{
    LbfgsArgs a{};
    for (int i = 0; i < kLbfgsSlots; ++i) { a.v.s[i] = c->hs[i]; a.v.y[i] = c->hy[i]; }
    a.st = c->lb_dev; a.part = c->lb_part; a.part2 = c->lb_part + 2 * kMaxPartials;
    a.gm = c->lb_gram; a.gpart = c->lb_gpart;
    a.g = c->g_cur; a.p = c->pvec; a.x = c->x[c->cur];
    a.n = (size_t)3 * c->H * c->W; a.step = (float)c->step_size; a.apply = apply;
    a.gsums = nullptr; a.n_global = 0;
    return a;
}

// One LBFGSOptimizer.step (optimizers.py:62-77).  Nothing is read back: the pair count, the ring order and the
// s.y > 1e-10 decision live on the device (lbfgs.hip), so consecutive steps queue up like Adam steps do.
int lbfgs_step(st_ctx* c)
{
    ST_TRY(lbfgs_alloc(c));
    const size_t n = (size_t)3 * c->H * c->W;
    float* x = c->x[c->cur];
    hipStream_t s = c->stream;
    if (c->lb_clear) {              // objective_changed / a new optimizer: sy = [], ss = [], ys = [] (optimizers.py:121-125)
        HIP_TRY(hipMemsetAsync(c->lb_dev, 0, sizeof(LbfgsDev), s));
        HIP_TRY(hipMemsetAsync(c->lb_gram, 0, sizeof(LbfgsGram), s));
        c->lb_clear = false;
        c->lb_gram_form = lbfgs_wants_gram(c);
        if (c->lb_gram_form) c->have_cur = false;      // (the gradient's inner products belong to the history that was just dropped)
    }
    const bool gram = c->lb_gram_form;
    if (!c->have_cur) {             // optimizers.py:64-65
        ST_TRY(eval_objective(c, x, true, c->g_cur, false, nullptr));
        c->have_cur = true;
        if (gram) {
            ProfScope ps(c, P_VECTOR, 0, 4.0 * n);
            HIP_TRY(launch_lbfgs_gram_pass(lbfgs_args(c, 1), nullptr, 0, s));
        }
    }
    {   // s = -step * inv_hv(grad) ; x += s          (optimizers.py:68-69, 89-108)
        ProfScope ps(c, P_VECTOR, 0, 4.0 * n * (gram ? 2.0 * kLbfgsCorr + 4.0 : 8.0 * kLbfgsCorr + 3.0));
        if (gram) HIP_TRY(launch_lbfgs_gram_apply(lbfgs_args(c, 1), s));
        else HIP_TRY(launch_lbfgs_two_loop(lbfgs_args(c, 1), s));
    }
    ST_TRY(eval_objective(c, x, true, c->grad, false, nullptr));       // new loss / grad (optimizers.py:72)
    {   // y = grad - self.grad ; store_curvature_pair(s, y)            (optimizers.py:73-87)
        ProfScope ps(c, P_VECTOR, 0, 4.0 * n * (gram ? 2.0 * kLbfgsCorr + 4.0 : 4.0));
        if (gram) HIP_TRY(launch_lbfgs_gram_pass(lbfgs_args(c, 1), c->grad, 1, s));
        else HIP_TRY(launch_lbfgs_pair(lbfgs_args(c, 1), c->grad, 0, s));
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
bool step_graph_ok(const st_ctx* c)
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

int step_graph_capture(st_ctx* c, int par)
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
// the optimizer step itself: everything st_step launches before the iterate is read back
int step_enqueue(st_ctx* c)
{
    struct InStep { st_ctx* c; InStep(st_ctx* x) : c(x) { c->in_step = true; } ~InStep() { c->in_step = false; } } in_step(c);
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
}  // namespace st2e

extern "C" {

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
    p.begins += 1;
    // buffers a re-allocation replaced: free them once no view handed out before it can still be in use (st2.h: five further begins)
    for (size_t i = 0; i < p.retired.size();) {
        if (p.begins - p.retired[i].at >= st_ctx::Pipe::kSlots) { (void)hipHostFree(p.retired[i].p); p.retired.erase(p.retired.begin() + i); }
        else ++i;
    }
    if (n3 > p.cap || p.want_head != p.have_head || p.want_tail != p.have_tail) {
        if (p.count) return fail(ST_ERR_STATE, "the input grew (or the frame room changed) while an iteration is in flight: st_step_end first");
        const size_t cap = std::max(n3, p.cap);
        for (int i = 0; i < st_ctx::Pipe::kSlots; ++i) {
            // the iterates already handed out are views of these buffers: retire them instead of freeing them
            if (p.pin_base[i]) { p.retired.push_back({p.pin_base[i], p.begins}); p.pin_base[i] = nullptr; p.img_pin[i] = nullptr; }
            if (cap > p.cap) { dfree(p.hwc[i]); ST_TRY(dmalloc(&p.hwc[i], cap)); }
            HIP_TRY(hipHostMalloc((void**)&p.pin_base[i], p.want_head + cap * sizeof(float) + p.want_tail, 0));
            p.img_pin[i] = (float*)(p.pin_base[i] + p.want_head);
        }
        p.cap = cap; p.have_head = p.want_head; p.have_tail = p.want_tail;
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

int st_step_frame_room(st_ctx* c, size_t head_bytes, size_t tail_bytes)
{
    if (!c) return fail(ST_ERR_ARG, "ctx is NULL");
    if (head_bytes > (1u << 20) || tail_bytes > (1u << 20)) return fail(ST_ERR_ARG, "frame room is limited to 1 MiB on either side");
    // An iteration begun before this call was given a slot WITHOUT the room; collected after it, the caller would assemble its frame
    // outside that slot's pinned allocation.  Refuse: the caller collects everything in flight first.
    if (c->pipe.count) return fail(ST_ERR_STATE, "%d iteration(s) in flight: collect them with st_step_end before changing the frame room", c->pipe.count);
    // the image stays page-aligned inside the pinned allocation
    c->pipe.want_head = (head_bytes + 4095) / 4096 * 4096;
    c->pipe.want_tail = tail_bytes;
    return ST_OK;
}

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
    if (lbfgs_wants_gram(c)) {      // the Gram form of the same recursion: its matrix from pairwise inner products taken one by one
        const size_t n3 = (size_t)3 * c->H * c->W;
        if (!c->lb_dots) ST_TRY(dmalloc(&c->lb_dots, (size_t)kLbNB * kLbNB));
        HIP_TRY(hipMemsetAsync(c->lb_dots, 0, sizeof(float) * kLbNB * kLbNB, st));
        std::vector<std::pair<int, const float*>> basis;
        for (int k = 0; k < n_pairs; ++k) { basis.push_back({k, c->hs[k]}); basis.push_back({kLbfgsSlots + k, c->hy[k]}); }
        basis.push_back({2 * kLbfgsSlots, c->g_cur});
        for (size_t i = 0; i < basis.size(); ++i)
            for (size_t j = i; j < basis.size(); ++j) {
                HIP_TRY(launch_vec_dot(basis[i].second, basis[j].second, n3, c->lb_part, c->lb_dots + basis[i].first * kLbNB + basis[j].first, st));
                if (i != j) HIP_TRY(hipMemcpyAsync(c->lb_dots + basis[j].first * kLbNB + basis[i].first, c->lb_dots + basis[i].first * kLbNB + basis[j].first,
                                                  sizeof(float), hipMemcpyDeviceToDevice, st));
            }
        HIP_TRY(launch_lbfgs_gram_load(lbfgs_args(c, 0), c->lb_dots, st));
        HIP_TRY(launch_lbfgs_gram_apply(lbfgs_args(c, 0), st));
    } else
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

}  // extern "C"
