// Device-resident resampling (Pillow-exact; utils.py:130-160, optimizers.py:29-40,110-119, worker.py:154-170).
#include "engine.h"

namespace st2e {
struct DevTable { int* lo = nullptr; int* n = nullptr; double* k = nullptr; ResampleTable t{}; int out = 0; };

int table_upload(const st_resample_table* h, DevTable* d)
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
void table_free(DevTable* d)
{
    if (d->lo) (void)hipFree(d->lo);
    if (d->n) (void)hipFree(d->n);
    if (d->k) (void)hipFree(d->k);
    *d = DevTable{};
}
int tables_valid_for(const st_resample_table* x, const st_resample_table* y, int H, int W)
{
    for (int i = 0; x && i < x->out_size; ++i) if (x->lo[i] < 0 || x->n[i] < 0 || x->lo[i] + x->n[i] > W || x->n[i] > x->kmax) return 0;
    for (int i = 0; y && i < y->out_size; ++i) if (y->lo[i] < 0 || y->n[i] < 0 || y->lo[i] + y->n[i] > H || y->n[i] > y->kmax) return 0;
    return 1;
}
}  // namespace st2e

extern "C" {

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

}  // extern "C"
