// Tile-sharded single image (BASELINE config 5): the phases of one iteration on one rank's window.
#include "engine.h"

// ---- tile-sharded single image (style_transfer2_amd/tiling.py has the design) ----------------------------------
// This context runs the window (tile + apron) of one rank.  Every reduction is restricted to the tile's region
// of each blob and left UN-normalised in a flat device buffer that the caller all-reduces (RCCL) between phases.
namespace st2e {
struct BlobRoi { int y0, x0, y1, x1; double n_global; };

BlobRoi tile_roi(const st_ctx* c, int b)
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
int tile_ensure(float** p, size_t* cap, size_t n)
{
    if (n > *cap) { dfree(*p); ST_TRY(dmalloc(p, n)); *cap = n; }
    return ST_OK;
}
// bf16 operands: does the style term of blob b run on its bf16 copy here (region-of-interest forms of gram16.hip / style16.hip)?
bool tile_style16(const st_ctx* c, int b)
{
    return c->bf16 && c->act.data16[b] && style_runs16(c, c->act, b);
}

// the style gradient of blob b over the tile's region (worker.py:262-269): fp32 kernel on the fp32 blob, or -- bf16 operands -- the
// bf16 kernel on the blob's bf16 copy, which touches the region's pixels only (the inject buffer is zeroed outside it once)
int tile_style_grad(st_ctx* c, int b, const BlobRoi& r, float* dst, bool is_inject, float c2, int fused, float sw, int accumulate, int* np)
{
    const ActSet& a = c->act;
    const int C = a.C[b];
    if (!tile_style16(c, b)) {
        if (!a.has32[b]) return fail(ST_ERR_STATE, "internal: style blob %d has no fp32 copy", b);
        const int need = style_grad_blocks(C, a.h[b], a.w[b]);
        if (c->s2_cap[b] < need) { dfree(c->s2_part[b]); ST_TRY(dmalloc(&c->s2_part[b], need)); c->s2_cap[b] = need; }
        PixRoi pr{r.y0, r.x0, r.y1, r.x1};
        HIP_TRY(launch_style_grad(c->dbuf, a.data[b], dst, c2, fused, sw, c->norms + b * 3 + 1, accumulate, c->s2_part[b], np, C, a.h[b], a.w[b], c->stream, &pr));
        return ST_OK;
    }
    const int rw = r.x1 - r.x0, rh = r.y1 - r.y0;
    const size_t hw = (size_t)rw * rh, plane = (size_t)a.h[b] * a.w[b];
    const int need = style_grad16_blocks(C, hw);
    if (c->s2_cap[b] < need) { dfree(c->s2_part[b]); ST_TRY(dmalloc(&c->s2_part[b], need)); c->s2_cap[b] = need; }
    if (style_grad16_pack_elems(C) > c->d16_cap) {
        dfree16(c->d16); c->d16_cap = 0;
        ST_TRY(dmalloc16(&c->d16, style_grad16_pack_elems(C)));
        c->d16_cap = style_grad16_pack_elems(C);
    }
    if (is_inject && !accumulate && !c->inject_roi_zero[b]) {
        HIP_TRY(hipMemsetAsync(dst, 0, (size_t)C * plane * sizeof(float), c->stream));
        c->inject_roi_zero[b] = 1;
    }
    GramRoi roi{r.y0, r.x0, rw, a.w[b], plane};
    HIP_TRY(launch_style_grad16(c->dbuf, conv_mpad(C), c->d16, a.data16[b], dst, c2, fused, sw, c->norms + b * 3 + 1, accumulate, c->s2_part[b], np,
                                C, hw, c->stream, &roi));
    return ST_OK;
}

// D = Graw / n_global - G_style into dbuf ([C][MPad]); sum D^2 -> pd[k]
int tile_style_D(st_ctx* c, int b, const float* graw, double n_global, float* pd_slot)
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
}  // namespace st2e

extern "C" {

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
    ST_TRY(ensure_content_features(c));
    ST_TRY(tile_ensure(&c->tile.p1, &c->tile.p1_n, std::max<size_t>(n1, 1)));
    HIP_TRY(hipMemsetAsync(c->tile.p1, 0, std::max<size_t>(n1, 1) * sizeof(float), c->stream));
    // bf16 operands: the lean data flow here too -- an fp32 blob / diff is written only where something reads fp32 (the weighted
    // blobs: the region-of-interest loss kernels are fp32), pools ride on their producing conv, the backward masks from the bf16 copies
    // fp32, inside the fused iteration (st_tile_step): the full-resolution blobs of pooled, un-weighted layers are not written either
    bool lean32 = false;
    if (!c->bf16 && c->tile.fused) { const char* e = getenv("ST2_LEAN32"); lean32 = !(e && *e == '0'); }
    ST_TRY(forward_range(c, a, c->x[c->cur], last, (c->bf16 && c->lean) || lean32));
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
            GramRoi roi{r.y0, r.x0, rw, a.w[b], (size_t)a.h[b] * a.w[b]};
            if (tile_style16(c, b)) {       // bf16 operands: the partials on the bf16 matrix cores, from the blob's bf16 copy (gram16.hip)
                ST_TRY(ensure_gram_bufs(c, C, hw, pl, true));
                HIP_TRY(launch_gram16_partial(a.data16[b], c->gram_slabs, C, hw, pl, c->stream, &roi));
            } else {
                if (!a.has32[b]) return fail(ST_ERR_STATE, "internal: style blob %d has no fp32 copy", b);
                ST_TRY(ensure_gram_bufs(c, C, hw, pl));
                HIP_TRY(launch_gram_partial(a.data[b], c->gram_slabs, C, hw, pl, c->stream, &roi));
            }
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
            int np = 0;
            if (two_step) {
                // norm from the all-reduced sum S^2 of the first pass (st_tile_style_raw), then saxpy
                if (!c->norm_valid[b * 3 + 1]) {
                    HIP_TRY(launch_finalize_norm(c->tile.p2 + k, 1, r.n_global, nrm + 1, c->stream));
                    c->norm_valid[b * 3 + 1] = 1;
                }
                ST_TRY(tile_style_grad(c, b, r, c->inject[b], true, c2, 1, al.sw, wrote, &np));
            } else {
                ST_TRY(tile_style_grad(c, b, r, c->inject[b], true, c2, 1, al.sw, wrote, &np));
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
        if (!c->stmp) ST_TRY(dmalloc(&c->stmp, c->max_blob));
        int np = 0;
        ST_TRY(tile_style_grad(c, b, r, c->stmp, false, c2, 0, al.sw, 0, &np));
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
        if (last > 0) ST_TRY(backward_chain(c, last, inj[last], inj, &g, c->bf16 && c->lean));
        HIP_TRY(hipMemcpyAsync(c->tile.wgrad, g, n3 * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
    }
    if (!c->tile.fused) HIP_TRY(hipStreamSynchronize(c->stream));
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
    if (!c->tile.fused) HIP_TRY(hipStreamSynchronize(c->stream));
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
    if (!c->tile.fused) HIP_TRY(hipStreamSynchronize(c->stream));
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
    if (!c->tile.fused) HIP_TRY(hipStreamSynchronize(c->stream));
    return ST_OK;
}

int st_vec_axpy(st_ctx* c, float alpha, const float* x_dev, float* y_dev, long long n)
{
    if (!c || !x_dev || !y_dev || n < 0) return fail(ST_ERR_ARG, "bad argument");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(launch_vec_axpy(alpha, x_dev, y_dev, (size_t)n, c->stream));
    if (!c->tile.fused) HIP_TRY(hipStreamSynchronize(c->stream));
    return ST_OK;
}

int st_vec_div(st_ctx* c, double divisor, float* y_dev, long long n)
{
    if (!c || !y_dev || n < 0 || !(divisor != 0.0)) return fail(ST_ERR_ARG, "bad argument");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(launch_vec_div(divisor, y_dev, (size_t)n, c->stream));
    if (!c->tile.fused) HIP_TRY(hipStreamSynchronize(c->stream));
    return ST_OK;
}

// device pointers of the buffers the caller exchanges: which = 0 current x, 1 next x, 2 local sum D^2 per style layer
int st_tile_buffer(st_ctx* c, int which, float** dev_ptr)
{
    if (c) c->epoch++;       // anything but st_step may change what a step launches: captured step graphs are stale
    if (!c || !dev_ptr) return fail(ST_ERR_ARG, "bad argument");
    if (!c->tile.fused) HIP_TRY(hipStreamSynchronize(c->stream));
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
    if (!c->tile.fused) HIP_TRY(hipStreamSynchronize(c->stream));
    return ST_OK;
}

int st_tile_swap(st_ctx* c)
{
    if (c) c->epoch++;       // anything but st_step may change what a step launches: captured step graphs are stale
    if (!c) return fail(ST_ERR_ARG, "ctx is NULL");
    c->cur ^= 1;
    return ST_OK;
}

}  // extern "C"
