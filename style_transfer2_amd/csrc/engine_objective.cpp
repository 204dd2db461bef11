// The objective: StyleTransfer.opfunc (worker.py:231-301) as one launch sequence, its trace, and the C ABI around it.
#include "engine.h"

namespace st2e {
// ------------------------------------------------------------------------------------ the objective
int eval_objective(st_ctx* c, const float* x, bool want_grad, float* grad_out, bool adam, float* x_next)
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
    // lean data flow: tensors nothing reads are not written.  bf16: the whole evaluation (st_set_precision(ctx, 1)); fp32: inside an
    // iteration (st_step / st_step_begin), where no caller can ask for a blob afterwards -- st_opfunc keeps every blob for the test hooks.
    // Identical values either way.  ST2_LEAN32=0 switches the fp32 part off (read per evaluation: A/B runs).
    const bool lean = c->bf16 && c->lean && !c->tile.on;         // the bf16 data flow (forward AND backward)
    bool lean32 = false;                                          // fp32: the forward only (dead pooled-layer blobs)
    if (!c->bf16 && c->in_step && !c->tile.on) { const char* e = getenv("ST2_LEAN32"); lean32 = !(e && *e == '0'); }
    ST_TRY(ensure_content_features(c));
    ST_TRY(forward_range(c, a, x, last, lean || lean32));

    std::vector<const float*> inj(c->nb, nullptr);
    std::fill(c->cnt.begin(), c->cnt.end(), 0);
    c->sf_in.assign(c->nb, nullptr); c->sf_w.assign(c->nb, nullptr);
    for (const ActiveLayer& al : c->active) {
        const int b = al.blob;
        const int C = a.C[b], hw = a.h[b] * a.w[b];
        const size_t n = (size_t)C * hw;
        if (!c->inject[b]) ST_TRY(dmalloc(&c->inject[b], n));
        c->inject_roi_zero[b] = 0;
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
                ProfScope ps(c, s16 ? P_STYLE_GRAD_BF16 : P_STYLE_GRAD, fl, n * (s16 ? 6.0 : 8.0));
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

int read_trace(st_ctx* c, double* trace, float* loss)
{
    const int n = c->trace_len_last;
    HIP_TRY(hipMemcpyAsync(c->trace_host, c->trace_dev, n * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (trace) for (int i = 0; i < n; ++i) trace[i] = c->trace_host[i];
    c->last_loss = c->trace_host[n - 2];
    if (loss) *loss = c->last_loss;
    return ST_OK;
}
}  // namespace st2e

extern "C" {

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
    // content features are kept only where a content weight reads them (ensure_content_features brings back what a later table needs)
    if (c->have_content) {
        std::vector<char> used(c->nb, 0);
        for (const ActiveLayer& al : c->active) if (al.c) used[al.blob] = 1;
        HIP_TRY(hipSetDevice(c->device));
        HIP_TRY(hipStreamSynchronize(c->stream));
        for (int b = 0; b < c->nb; ++b) if (!used[b]) dfree(c->content_feat[b]);
    }
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

}  // extern "C"
