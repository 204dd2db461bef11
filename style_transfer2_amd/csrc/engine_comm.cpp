// The tile-sharded iteration with its communication inside the engine (BASELINE config 5: one image over several GPUs).
//
// One RCCL communicator per context (librccl is loaded on first use, so a single-GPU worker never touches it).  The collectives
// of an Adam iteration -- two all-reduces of the phase buffers and three strip exchanges with the grid neighbours -- are enqueued
// on the engine's own stream between the compute phases of engine_tile.cpp; the host synchronises once per iteration to read the
// trace.  xGMI is point to point: every neighbour gets ONE message per phase (its rectangles packed by one kernel), all sends and
// receives of a phase sit in one ncclGroup.  A caller-supplied transport can stand in for RCCL (tests: several ranks on one GPU).
#include "engine.h"

#include <dlfcn.h>

namespace st2e {

// ---- the few RCCL entry points used, resolved from librccl at run time ------------------------------------------------------------
struct RcclId { char internal[ST_COMM_ID_BYTES]; };
typedef int (*fn_get_id)(RcclId*);
typedef int (*fn_init_rank)(void**, int, RcclId, int);
typedef int (*fn_destroy)(void*);
typedef const char* (*fn_errstr)(int);
typedef int (*fn_allreduce)(const void*, void*, size_t, int, int, void*, hipStream_t);
typedef int (*fn_sendrecv)(void*, size_t, int, int, void*, hipStream_t);
typedef int (*fn_group)(void);
typedef int (*fn_version)(int*);
static struct Rccl {
    void* lib = nullptr;
    fn_get_id get_id = nullptr; fn_init_rank init_rank = nullptr; fn_destroy destroy = nullptr; fn_errstr errstr = nullptr;
    fn_allreduce allreduce = nullptr; fn_sendrecv send = nullptr, recv = nullptr; fn_group group_start = nullptr, group_end = nullptr;
    fn_version version = nullptr;
} g_rccl;
constexpr int kNcclFloat = 7, kNcclSum = 0;        // ncclFloat32, ncclSum (rccl.h)

static int rccl_load()
{
    if (g_rccl.lib) return ST_OK;
    void* lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!lib) lib = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
    if (!lib) lib = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!lib) return fail(ST_ERR_HIP, "cannot load librccl.so: %s", dlerror());
    Rccl r;
    r.lib = lib;
    r.get_id = (fn_get_id)dlsym(lib, "ncclGetUniqueId"); r.init_rank = (fn_init_rank)dlsym(lib, "ncclCommInitRank");
    r.destroy = (fn_destroy)dlsym(lib, "ncclCommDestroy"); r.errstr = (fn_errstr)dlsym(lib, "ncclGetErrorString");
    r.allreduce = (fn_allreduce)dlsym(lib, "ncclAllReduce"); r.send = (fn_sendrecv)dlsym(lib, "ncclSend"); r.recv = (fn_sendrecv)dlsym(lib, "ncclRecv");
    r.group_start = (fn_group)dlsym(lib, "ncclGroupStart"); r.group_end = (fn_group)dlsym(lib, "ncclGroupEnd");
    r.version = (fn_version)dlsym(lib, "ncclGetVersion");
    if (!r.get_id || !r.init_rank || !r.destroy || !r.errstr || !r.allreduce || !r.send || !r.recv || !r.group_start || !r.group_end) {
        dlclose(lib);
        return fail(ST_ERR_HIP, "librccl.so lacks an entry point this engine needs");
    }
    g_rccl = r;
    return ST_OK;
}
#define RCCL_TRY(expr)                                                                                         \
    do {                                                                                                       \
        int r_ = (expr);                                                                                       \
        if (r_ != 0) return fail(ST_ERR_HIP, "%s failed: %s (%s:%d)", #expr, g_rccl.errstr(r_), __FILE__, __LINE__); \
    } while (0)

void comm_free(st_ctx* c)
{
    st_ctx::Comm& m = c->comm;
    for (auto& plan : m.plan) {
        for (auto& p : plan) { dfree(p.sbuf); dfree(p.rbuf); }
        plan.clear();
    }
    dfree(m.ring); m.ring_cap = 0;
    dfree(m.tile_chw); dfree(m.tile_hwc); m.tile_cap = 0;
    if (m.comm && g_rccl.destroy) (void)g_rccl.destroy(m.comm);
    m.comm = nullptr;
    m.ar = nullptr; m.ex = nullptr; m.user = nullptr;
    m.world = 1; m.rank = 0;
}

static bool comm_ready(const st_ctx* c) { return c->comm.world == 1 || c->comm.comm || (c->comm.ar && c->comm.ex); }

// in-place sum of n floats over the ranks, ordered on the engine's stream
static int comm_allreduce(st_ctx* c, float* buf, int n)
{
    st_ctx::Comm& m = c->comm;
    if ((m.world == 1 && !m.comm) || n <= 0 || !buf) return ST_OK;         // (a one-rank RCCL communicator still runs the collective)
    ProfScope ps(c, P_COMM, 0, 4.0 * n);
    if (m.ar && m.world > 1) {
        HIP_TRY(hipStreamSynchronize(c->stream));
        if (m.ar(m.user, buf, n) != 0) return fail(ST_ERR_HIP, "the caller's all-reduce failed");
        return ST_OK;
    }
    if (!m.comm) return ST_OK;
    RCCL_TRY(g_rccl.allreduce(buf, buf, (size_t)n, kNcclFloat, kNcclSum, m.comm, c->stream));
    return ST_OK;
}

static int strips(st_ctx* c, float* tensor, int C, int h, int w, const std::vector<int>& rects, float* buf, int mode)
{
    const int n = (int)rects.size() / 4;
    size_t off = 0;
    for (int r0 = 0; r0 < n; r0 += kMaxStripRects) {        // at most kMaxStripRects rectangles per launch
        StripTable t{};
        t.n = std::min(kMaxStripRects, n - r0);
        int total = 0;
        for (int r = 0; r < t.n; ++r) {
            const int* q = &rects[4 * (r0 + r)];
            t.y0[r] = q[0]; t.x0[r] = q[1]; t.h[r] = q[2]; t.w[r] = q[3];
            if (q[0] < 0 || q[1] < 0 || q[2] <= 0 || q[3] <= 0 || q[0] + q[2] > h || q[1] + q[3] > w)
                return fail(ST_ERR_ARG, "planned strip %d lies outside the %dx%d tensor", r0 + r, h, w);
            t.off[r] = total;
            total += C * q[2] * q[3];
        }
        t.total = total;
        HIP_TRY(launch_strip_copy(tensor, buf + off, t, C, h, w, mode, c->stream));
        off += (size_t)total;
    }
    return ST_OK;
}

// One exchange phase: pack what each neighbour gets out of `src`, one message per neighbour, unpack (assign / add) into `dst`.
static int comm_exchange(st_ctx* c, int phase, float* src, int sh, int sw, float* dst, int dh, int dw, bool add)
{
    st_ctx::Comm& m = c->comm;
    if (!m.planned[phase]) return fail(ST_ERR_STATE, "st_tile_plan(%d) first", phase);
    auto& plan = m.plan[phase];
    ProfScope ps(c, P_COMM, 0, 0);
    // ST2_COMM_SELF_VIA_RCCL=1 (test hook): a rank's copies to itself travel as ncclSend / ncclRecv to its own rank too
    const bool self_rccl = m.self_via_rccl && m.comm && !m.ex;
    auto remote = [&](const st_ctx::Comm::Peer& p) { return p.peer != m.rank || (self_rccl && p.rbuf); };
    for (auto& p : plan) {
        if (p.sn) ST_TRY(strips(c, src, 3, sh, sw, p.send, p.sbuf, 0));
        if (!remote(p)) {                   // the periodic wrap lands on this rank's own tile: a local copy through the send buffer
            if (p.rn) ST_TRY(strips(c, dst, 3, dh, dw, p.recv, p.sbuf, add ? 2 : 1));
        }
    }
    if (m.world > 1 || self_rccl) {
        if (m.ex) {
            std::vector<int> sp, sc, rp, rc;
            std::vector<float*> sb, rb;
            for (auto& p : plan) {
                if (!remote(p)) continue;
                if (p.sn) { sp.push_back(p.peer); sb.push_back(p.sbuf); sc.push_back((int)p.sn); }
                if (p.rn) { rp.push_back(p.peer); rb.push_back(p.rbuf); rc.push_back((int)p.rn); }
            }
            HIP_TRY(hipStreamSynchronize(c->stream));
            if (m.ex(m.user, (int)sp.size(), sp.data(), sb.data(), sc.data(), (int)rp.size(), rp.data(), rb.data(), rc.data()) != 0)
                return fail(ST_ERR_HIP, "the caller's strip exchange failed");
        } else {
            // every exit path closes the group: a send / recv that fails inside an open group would leave it open on this thread,
            // later RCCL calls would never be issued and the peers would wait for ever instead of seeing this rank's error
            RCCL_TRY(g_rccl.group_start());
            int first = 0;
            const char* what = "";
            for (auto& p : plan) {
                if (!remote(p) || first) continue;
                if (p.sn && (first = g_rccl.send(p.sbuf, p.sn, kNcclFloat, p.peer, m.comm, c->stream)) != 0) { what = "ncclSend"; continue; }
                if (p.rn && (first = g_rccl.recv(p.rbuf, p.rn, kNcclFloat, p.peer, m.comm, c->stream)) != 0) what = "ncclRecv";
            }
            const int ended = g_rccl.group_end();
            if (first) return fail(ST_ERR_HIP, "%s failed inside the strip exchange of phase %d: %s", what, phase, g_rccl.errstr(first));
            if (ended) return fail(ST_ERR_HIP, "ncclGroupEnd failed in the strip exchange of phase %d: %s", phase, g_rccl.errstr(ended));
        }
    }
    for (auto& p : plan)                    // ascending peer (the plan's order), plan order inside: deterministic sums
        if (remote(p) && p.rn) ST_TRY(strips(c, dst, 3, dh, dw, p.recv, p.rbuf, add ? 2 : 1));
    return ST_OK;
}

// Trace scalars from the reduced sums, in the reference's fp32 order (worker.py:249-301); the layout of the single-GPU engine's
// trace: 6 per active layer + 8.  p1 / p2 / p3 are the all-reduced phase buffers, pd the sum D^2 per style layer.
static void tile_trace(const st_ctx* c, const float* p1, const float* s2, const float* p3, const float* pd, const float* norms, double* out)
{
    const ActSet& a = c->act;
    float loss = 0.f;
    size_t pos = 0;
    int k = 0, o = 0;
    for (const ActiveLayer& al : c->active) {
        const int b = al.blob, C = a.C[b];
        int gh = c->tile.gH, gw = c->tile.gW;
        for (int i = 1; i <= b; ++i) if (!c->topo[i - 1].is_conv) { gh = pooled_size(gh); gw = pooled_size(gw); }
        const float n = (float)((double)C * gh * gw);
        const float* sums = p1 + pos;
        pos += 4;
        float v[6] = {0, 0, 0, 0, 0, 0};
        if (al.c) {
            const float cn = norms[b * 3 + 0];
            v[0] = al.cw * (sums[0] / n) / cn;
            v[1] = fabsf(al.cw) * sqrtf(sums[1] / n) / cn;
            loss += v[0];
        }
        if (al.s) {
            const float sn = norms[b * 3 + 1];
            v[2] = al.sw * (pd[k] / (float)(C * C)) / sn;
            v[3] = fabsf(al.sw / sn) * sqrtf(s2[k] / n);
            loss += v[2];
            pos += (size_t)C * C;
            ++k;
        }
        if (al.d) {
            const float dn = norms[b * 3 + 2];
            v[4] = -al.dw * (sums[2] / n) / dn;
            v[5] = fabsf(al.dw) * sqrtf(sums[3] / n) / dn;
            loss += v[4];
        }
        for (int j = 0; j < 6; ++j) out[o++] = v[j];
    }
    const float n3 = (float)(3.0 * c->tile.gH * c->tile.gW);
    const float scd = loss, tl = c->tv_w * p3[0], pl = c->p_w * (p3[1] / c->p_pow), total = scd + tl + pl;
    const float tail[8] = {scd, tl, pl, sqrtf(p3[2] / n3), sqrtf(p3[3] / n3), sqrtf(p3[4] / n3), total, sqrtf(p3[5] / n3)};
    for (int j = 0; j < 8; ++j) out[o++] = tail[j];
}

}  // namespace st2e

extern "C" {

int st_comm_unique_id(char out_id[ST_COMM_ID_BYTES])
{
    if (!out_id) return fail(ST_ERR_ARG, "out_id is NULL");
    ST_TRY(rccl_load());
    RcclId id;
    RCCL_TRY(g_rccl.get_id(&id));
    memcpy(out_id, id.internal, ST_COMM_ID_BYTES);
    return ST_OK;
}

int st_comm_init(st_ctx* c, const char id[ST_COMM_ID_BYTES], int rank, int world)
{
    if (!c || !id || world < 1 || rank < 0 || rank >= world) return fail(ST_ERR_ARG, "bad argument");
    HIP_TRY(hipSetDevice(c->device));
    ST_TRY(rccl_load());
    // Preflight: everything that would otherwise show up as a hang inside the first collective is checked here and reported.
    if (g_rccl.version) {                        // grouped ncclSend / ncclRecv (the strip exchanges) exist since 2.7
        int v = 0;
        if (g_rccl.version(&v) == 0 && v < 2700) return fail(ST_ERR_HIP, "librccl reports version %d: the strip exchanges need ncclSend / ncclRecv (>= 2.7)", v);
    }
    if (c->tile.on && c->comm.planned[0]) {      // plans handed over before the communicator: their peers must exist in this world
        for (int ph = 0; ph < 3; ++ph)
            for (const auto& p : c->comm.plan[ph])
                if (p.peer >= world) return fail(ST_ERR_ARG, "the exchange plan of phase %d names rank %d but the communicator has %d ranks (world must equal rows x cols of the tile grid)", ph, p.peer, world);
    }
    {   // xGMI is point to point: a peer this device cannot reach directly makes RCCL stage through the host -- legal, slow, worth a line
        int ndev = 0;
        if (hipGetDeviceCount(&ndev) == hipSuccess && ndev > 1) {
            int unreachable = 0;
            for (int d = 0; d < ndev; ++d) {
                int ok = 1;
                if (d != c->device && hipDeviceCanAccessPeer(&ok, c->device, d) == hipSuccess && !ok) ++unreachable;
            }
            if (unreachable) {
                // ST2_REQUIRE_PEER_ACCESS=1: a job that must not run through host staging fails here, on every rank that sees it, before the
                // first collective (default: a line on stderr -- the job is correct either way)
                const char* req = getenv("ST2_REQUIRE_PEER_ACCESS");
                if (req && *req == '1') { (void)hipGetLastError(); return fail(ST_ERR_STATE, "device %d has no direct peer access to %d of the %d visible devices and ST2_REQUIRE_PEER_ACCESS=1", c->device, unreachable, ndev - 1); }
                fprintf(stderr, "st_comm_init: device %d has no direct peer access to %d of the %d visible devices (RCCL will stage through host memory)\n", c->device, unreachable, ndev - 1);
            }
        }
        (void)hipGetLastError();
    }
    if (c->comm.comm) { (void)g_rccl.destroy(c->comm.comm); c->comm.comm = nullptr; }
    RcclId uid;
    memcpy(uid.internal, id, ST_COMM_ID_BYTES);
    RCCL_TRY(g_rccl.init_rank(&c->comm.comm, world, uid, rank));
    c->comm.rank = rank; c->comm.world = world;
    c->comm.ar = nullptr; c->comm.ex = nullptr; c->comm.user = nullptr;
    { const char* e = getenv("ST2_COMM_SELF_VIA_RCCL"); c->comm.self_via_rccl = e && *e == '1'; }
    return ST_OK;
}

int st_comm_callbacks(st_ctx* c, int rank, int world, st_allreduce_fn allreduce, st_exchange_fn exchange, void* user)
{
    if (!c || world < 1 || rank < 0 || rank >= world || (world > 1 && (!allreduce || !exchange))) return fail(ST_ERR_ARG, "bad argument");
    if (c->comm.comm && g_rccl.destroy) { (void)g_rccl.destroy(c->comm.comm); c->comm.comm = nullptr; }
    c->comm.rank = rank; c->comm.world = world;
    c->comm.ar = allreduce; c->comm.ex = exchange; c->comm.user = user;
    return ST_OK;
}

int st_comm_destroy(st_ctx* c)
{
    if (!c) return fail(ST_ERR_ARG, "ctx is NULL");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    comm_free(c);
    return ST_OK;
}

int st_comm_barrier(st_ctx* c)
{
    if (!c) return fail(ST_ERR_ARG, "ctx is NULL");
    if (!comm_ready(c)) return fail(ST_ERR_STATE, "st_comm_init first");
    HIP_TRY(hipSetDevice(c->device));
    if (c->comm.world > 1) {
        if (c->comm.ring_cap < 4) { dfree(c->comm.ring); ST_TRY(dmalloc(&c->comm.ring, 4)); c->comm.ring_cap = 4; }
        HIP_TRY(hipMemsetAsync(c->comm.ring, 0, sizeof(float), c->stream));
        ST_TRY(comm_allreduce(c, c->comm.ring, 1));
    }
    HIP_TRY(hipStreamSynchronize(c->stream));
    return ST_OK;
}

int st_tile_plan(st_ctx* c, int phase, int n_peers, const st_tile_peer* peers)
{
    if (c) c->epoch++;
    if (!c || phase < 0 || phase > 2 || n_peers < 0 || (n_peers && !peers)) return fail(ST_ERR_ARG, "bad argument");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    auto& plan = c->comm.plan[phase];
    for (auto& p : plan) { dfree(p.sbuf); dfree(p.rbuf); }
    plan.clear();
    for (int i = 0; i < n_peers; ++i) {
        const st_tile_peer& q = peers[i];
        if (q.n_send < 0 || q.n_recv < 0 || (q.n_send && !q.send_rects) || (q.n_recv && !q.recv_rects)) return fail(ST_ERR_ARG, "peer %d: bad rectangle lists", i);
        if (i && q.peer <= peers[i - 1].peer) return fail(ST_ERR_ARG, "peers must be listed in ascending order, once each");
        if (q.peer < 0 || (comm_ready(c) && c->comm.world > 1 && q.peer >= c->comm.world))
            return fail(ST_ERR_ARG, "peer %d does not exist in a communicator of %d ranks (world must equal rows x cols of the tile grid)", q.peer, c->comm.world);
        st_ctx::Comm::Peer p;
        p.peer = q.peer;
        p.send.assign(q.send_rects, q.send_rects + 4 * q.n_send);
        p.recv.assign(q.recv_rects, q.recv_rects + 4 * q.n_recv);
        for (int r = 0; r < q.n_send; ++r) p.sn += (size_t)3 * p.send[4 * r + 2] * p.send[4 * r + 3];
        for (int r = 0; r < q.n_recv; ++r) p.rn += (size_t)3 * p.recv[4 * r + 2] * p.recv[4 * r + 3];
        if (q.peer == c->comm.rank && p.sn != p.rn) return fail(ST_ERR_ARG, "a local copy must send and receive the same number of pixels");
        if (p.sn > 0x7fffffffu || p.rn > 0x7fffffffu) return fail(ST_ERR_ARG, "a strip message is limited to 2^31 floats");
        if (p.sn) ST_TRY(dmalloc(&p.sbuf, p.sn));
        if (p.rn && (q.peer != c->comm.rank || c->comm.self_via_rccl)) {
            ST_TRY(dmalloc(&p.rbuf, p.rn));
            // a transport that delivers nothing (the solo-rank timing hook) must leave zeros, not whatever the allocation held
            HIP_TRY(hipMemset(p.rbuf, 0, p.rn * sizeof(float)));
        }
        plan.push_back(std::move(p));
    }
    c->comm.planned[phase] = true;
    return ST_OK;
}

// One objective evaluation of the sharded image at x[cur] (worker.py:231-301 over all ranks): forward -> all-reduce of the phase-1 sums
// -> losses (first evaluation: raw style gradients -> all-reduce) -> ranged backward -> overlap-add of the window gradients -> the
// torus ring of the TV stencil -> image-space pass.  adam: that pass is the fused TV / p-norm / Adam update into x[cur ^ 1]; else it
// leaves the combined gradient of the tile's pixels in c->grad (window layout).  The phase-3 sums are all-reduced either way.
struct TileEval { float *p1 = nullptr, *p2 = nullptr, *p3 = nullptr; int n1 = 0, n2 = 0, n3 = 0; };
static int tile_evaluate(st_ctx* c, bool adam, TileEval& e, bool want_trace)
{
    st_ctx::Tile& t = c->tile;
    const int wh = c->H, ww = c->W, th = t.ty1 - t.ty0, tw = t.tx1 - t.tx0;
    float* wgrad = nullptr;
    ST_TRY(st_tile_forward(c, &e.p1, &e.n1));                                   // phase 1: forward, region sums + raw Gram sums
    ST_TRY(comm_allreduce(c, e.p1, e.n1));
    ST_TRY(st_tile_losses(c, &e.p2, &e.n2));                                    // phase 2: norms (first evaluation: raw style gradients)
    if (e.n2) {
        ST_TRY(st_tile_style_raw(c));
        ST_TRY(comm_allreduce(c, e.p2, e.n2));
    }
    ST_TRY(st_tile_losses_finish(c));
    ST_TRY(st_tile_backward(c, &wgrad));                                        // phase 3: ranged backward on the window
    ST_TRY(comm_exchange(c, ST_TILE_PLAN_OVERLAP, wgrad, wh, ww, wgrad, wh, ww, true));
    const size_t ring_n = (size_t)3 * (th + 2) * (tw + 2);
    if (ring_n > c->comm.ring_cap) { HIP_TRY(hipStreamSynchronize(c->stream)); dfree(c->comm.ring); c->comm.ring_cap = 0; ST_TRY(dmalloc(&c->comm.ring, ring_n)); c->comm.ring_cap = ring_n; }
    HIP_TRY(hipMemsetAsync(c->comm.ring, 0, ring_n * sizeof(float), c->stream));
    ST_TRY(comm_exchange(c, ST_TILE_PLAN_RING, c->x[c->cur], wh, ww, c->comm.ring, th + 2, tw + 2, false));
    if (adam) ST_TRY(st_tile_update(c, c->comm.ring, &e.p3, &e.n3));            // phase 4: TV / p-norm / Adam on the tile
    else ST_TRY(st_tile_gradient(c, c->comm.ring, &e.p3, &e.n3));               //          or the combined gradient only
    // the image-space sums (and, in steady state, the style-gradient sums) feed the TRACE only.  They are reduced on EVERY call all the
    // same (a few KB; until round 4 a caller that passed trace == NULL skipped this collective: ranks that disagreed on NULL for one
    // iteration then desynchronised the collective sequence and RCCL hung instead of failing) -- trace == NULL now only means that
    // nothing is read back and nothing waits for the GPU
    (void)want_trace;
    ST_TRY(comm_allreduce(c, e.p3, e.n3));
    return ST_OK;
}

// the trace of the evaluation just enqueued: the reduced sums cross PCIe (a few KB + the Gram sums), the scalars are finished on the host
static int tile_read_trace(st_ctx* c, const TileEval& e, double* trace)
{
    int n_style = 0;
    for (const ActiveLayer& al : c->active) n_style += al.s;
    std::vector<float> &h1 = c->comm.h1, &h2 = c->comm.h2, &h3 = c->comm.h3, &hd = c->comm.hd, &hn = c->comm.hn;      // (kept in the context: grown, never shrunk)
    auto room = [](std::vector<float>& v, size_t n) { if (v.size() < n) v.resize(n); };
    room(h1, (size_t)std::max(e.n1, 1)); room(h2, (size_t)std::max(std::max(n_style, e.n2), 1)); room(h3, (size_t)std::max(6 + kMaxTraceLayers, e.n3));
    room(hd, (size_t)std::max(n_style, 1)); room(hn, (size_t)c->nb * 3);
    if (e.n1) HIP_TRY(hipMemcpyAsync(h1.data(), e.p1, (size_t)e.n1 * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    if (e.n2) HIP_TRY(hipMemcpyAsync(h2.data(), e.p2, (size_t)e.n2 * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpyAsync(h3.data(), e.p3, (size_t)e.n3 * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    if (n_style) HIP_TRY(hipMemcpyAsync(hd.data(), c->tile.pd, (size_t)n_style * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpyAsync(hn.data(), c->norms, hn.size() * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    tile_trace(c, h1.data(), e.n2 ? h2.data() : h3.data() + 6, h3.data(), hd.data(), hn.data(), trace);
    return ST_OK;
}

// LBFGSOptimizer.step (optimizers.py:62-108) over the sharded image, fused: every rank keeps its TILE of x, of the gradient and of
// the <= 10 curvature pairs as compact (3, th, tw) vectors and runs the Gram form of the recursion (lbfgs.hip): the direction is a
// linear combination of {s_i, y_i, g} whose coefficients follow from the matrix of their inner products.  One pass over the tile's
// vectors yields this rank's share of the 2 (2k + 3) inner products a new pair and a new gradient add; ONE all-reduce of that
// 2 kLbNB-vector replaces the <= 2k + 3 scalar all-reduces of the chain form (tiled.TiledTransfer), after which every rank runs the same
// bookkeeping (the s.y > 1e-10 gate, eviction) and coefficient recursion on the same numbers.  Per step: 2 all-reduces + 3 exchanges
// for the evaluation, 1 all-reduce for the pair, 1 apron refresh.
static int tile_lbfgs_step(st_ctx* c, TileEval& e, bool want_trace)
{
    st_ctx::Tile& t = c->tile;
    const int wh = c->H, ww = c->W, th = t.ty1 - t.ty0, tw = t.tx1 - t.tx0;
    const size_t n_t = (size_t)3 * th * tw;
    ST_TRY(lbfgs_alloc(c));                                                     // (window-sized vectors: the tile's fit)
    if (t.lb_n != n_t) {
        HIP_TRY(hipStreamSynchronize(c->stream));
        dfree(t.lb_x); dfree(t.lb_sums);
        ST_TRY(dmalloc(&t.lb_x, n_t)); ST_TRY(dmalloc(&t.lb_sums, (size_t)lbfgs_gram_rows()));
        HIP_TRY(hipMemset(c->lb_gpart, 0, (size_t)lbfgs_gram_rows() * kMaxPartials * sizeof(float)));      // rows of dead ids are summed too (never used)
        t.lb_n = n_t;
    }
    hipStream_t s = c->stream;
    const std::vector<int> tile_rect = {t.ty0 - t.wy0, t.tx0 - t.wx0, th, tw};
    auto args = [&](int apply) {
        LbfgsArgs a = lbfgs_args(c, apply);
        a.n = n_t; a.n_global = (size_t)3 * t.gH * t.gW; a.x = t.lb_x; a.gsums = t.lb_sums;
        return a;
    };
    if (c->lb_clear) {              // objective_changed / a new optimizer: sy = [], ss = [], ys = [] (optimizers.py:121-125)
        HIP_TRY(hipMemsetAsync(c->lb_dev, 0, sizeof(LbfgsDev), s));
        HIP_TRY(hipMemsetAsync(c->lb_gram, 0, sizeof(LbfgsGram), s));
        c->lb_clear = false;
        c->lb_gram_form = true;
        c->have_cur = false;
    }
    if (!c->lb_gram_form) return fail(ST_ERR_STATE, "the history was built by the chain form: reset the optimizer before the fused tile-sharded L-BFGS");
    if (!c->have_cur) {             // optimizers.py:64-65: loss, grad at the starting point
        ST_TRY(tile_evaluate(c, false, e, false));
        ST_TRY(strips(c, c->grad, 3, wh, ww, tile_rect, c->g_cur, 0));
        { ProfScope ps(c, P_VECTOR, 0, 4.0 * n_t);
          HIP_TRY(launch_lbfgs_gram_pass_local(args(1), nullptr, 0, s)); }
        ST_TRY(comm_allreduce(c, t.lb_sums, lbfgs_gram_rows()));
        HIP_TRY(launch_lbfgs_gram_commit_global(args(1), 0, s));
        c->have_cur = true;
    }
    {   // s = -step * inv_hv(grad); x += s on the tile (optimizers.py:68-69, 89-108), then the neighbours' aprons follow
        ST_TRY(strips(c, c->x[c->cur], 3, wh, ww, tile_rect, t.lb_x, 0));
        { ProfScope ps(c, P_VECTOR, 0, 4.0 * n_t * (2.0 * kLbfgsCorr + 4.0));
          HIP_TRY(launch_lbfgs_gram_apply(args(1), s)); }
        ST_TRY(strips(c, c->x[c->cur], 3, wh, ww, tile_rect, t.lb_x, 1));
        ST_TRY(comm_exchange(c, ST_TILE_PLAN_REFRESH, c->x[c->cur], wh, ww, c->x[c->cur], wh, ww, false));
    }
    ST_TRY(tile_evaluate(c, false, e, want_trace));                             // loss, grad = opfunc(x)           (optimizers.py:72)
    ST_TRY(strips(c, c->grad, 3, wh, ww, tile_rect, c->pvec, 0));               // this rank's tile of the new gradient
    {   // y = grad - self.grad; store_curvature_pair(s, y)                                                         (optimizers.py:73-87)
        { ProfScope ps(c, P_VECTOR, 0, 4.0 * n_t * (2.0 * kLbfgsCorr + 4.0));
          HIP_TRY(launch_lbfgs_gram_pass_local(args(1), c->pvec, 1, s)); }
        ST_TRY(comm_allreduce(c, t.lb_sums, lbfgs_gram_rows()));
        HIP_TRY(launch_lbfgs_gram_commit_global(args(1), 1, s));
    }
    std::swap(c->g_cur, c->pvec);
    return ST_OK;
}

int st_tile_step(st_ctx* c, double* trace)
{
    if (c) c->epoch++;
    if (!c || !c->tile.on) return fail(ST_ERR_STATE, "st_tile_configure first");
    if (!comm_ready(c)) return fail(ST_ERR_STATE, "st_comm_init (or st_comm_callbacks) first");
    if (c->opt_kind != ST_OPT_ADAM && c->opt_kind != ST_OPT_LBFGS) return fail(ST_ERR_STATE, "no optimizer: call st_optimizer_reset first");
    HIP_TRY(hipSetDevice(c->device));
    const int wh = c->H, ww = c->W;
    struct Unfuse { st_ctx* c; ~Unfuse() { c->tile.fused = false; } } unfuse{c};
    c->tile.fused = true;
    TileEval e;
    if (c->opt_kind == ST_OPT_LBFGS) {
        ST_TRY(tile_lbfgs_step(c, e, trace != nullptr));
        if (trace) ST_TRY(tile_read_trace(c, e, trace));                        // (x was updated in place: no swap)
    } else {
        ST_TRY(tile_evaluate(c, true, e, trace != nullptr));
        ST_TRY(comm_exchange(c, ST_TILE_PLAN_REFRESH, c->x[c->cur ^ 1], wh, ww, c->x[c->cur ^ 1], wh, ww, false));
        if (trace) ST_TRY(tile_read_trace(c, e, trace));
        c->cur ^= 1;                                                            // st_tile_swap
    }
    c->comm.steps += 1;
    return ST_OK;
}

int st_tile_get_tile(st_ctx* c, float* out_hwc)
{
    if (!c || !c->tile.on || !out_hwc) return fail(ST_ERR_STATE, "st_tile_configure first");
    HIP_TRY(hipSetDevice(c->device));
    const st_ctx::Tile& t = c->tile;
    const int th = t.ty1 - t.ty0, tw = t.tx1 - t.tx0;
    const size_t n = (size_t)3 * th * tw;
    if (n > c->comm.tile_cap) {          // staging kept in the context (two device allocations per call until round 4)
        HIP_TRY(hipStreamSynchronize(c->stream));
        dfree(c->comm.tile_chw); dfree(c->comm.tile_hwc); c->comm.tile_cap = 0;
        ST_TRY(dmalloc(&c->comm.tile_chw, n)); ST_TRY(dmalloc(&c->comm.tile_hwc, n));
        c->comm.tile_cap = n;
    }
    float *chw = c->comm.tile_chw, *hwc = c->comm.tile_hwc;
    int rc = ST_OK;
    {
        std::vector<int> rect = {t.ty0 - t.wy0, t.tx0 - t.wx0, th, tw};
        rc = strips(c, c->x[c->cur], 3, c->H, c->W, rect, chw, 0);
    }
    if (rc == ST_OK && launch_deprocess(chw, hwc, th, tw, c->stream) != hipSuccess) rc = fail(ST_ERR_HIP, "deprocess launch failed");
    if (rc == ST_OK && hipMemcpyAsync(out_hwc, hwc, n * sizeof(float), hipMemcpyDeviceToHost, c->stream) != hipSuccess) rc = fail(ST_ERR_HIP, "tile copy failed");
    if (hipStreamSynchronize(c->stream) != hipSuccess && rc == ST_OK) rc = fail(ST_ERR_HIP, "tile copy failed");
    return rc;
}

}  // extern "C"
