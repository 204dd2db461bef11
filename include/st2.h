/* st2.h -- C ABI of libst2_hip.so: the MI355X-resident style-transfer inner loop.
 *
 * The reference (crowsonkb/style_transfer2) has no FFI of its own: the hot path sits behind two
 * Python duck types inside worker.py -- the model (``CaffeModel``, worker.py:32-106) and the
 * objective/optimizer pair (``StyleTransfer`` worker.py:117-315, optimizers.py:7-125).  Each entry
 * point below names the reference interface it replaces.  The reference-side binding (a ctypes
 * stub that drops into worker.py) is shown in INTEGRATION.md; this repo's own host mirror is
 * style_transfer2_amd/engine.py.
 *
 * Conventions: plain pointers and sizes only; every function returns 0 on success and a non-zero
 * code on failure with a message available from st_last_error(); host buffers are caller-owned
 * and are fully consumed/produced before the call returns; all device state is owned by the
 * handle; one handle is driven by one host thread.  Images cross the boundary as HxWx3 RGB
 * (uint8 or float32) exactly as in messages.py:89-110; tensors as contiguous NCHW float32.
 */
#ifndef ST2_H
#define ST2_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct st_ctx st_ctx;

enum { ST_OK = 0, ST_ERR_ARG = 1, ST_ERR_STATE = 2, ST_ERR_HIP = 3 };
enum { ST_OPT_NONE = 0, ST_OPT_ADAM = 1, ST_OPT_LBFGS = 2 };
enum { ST_LAYER_CONV = 0, ST_LAYER_POOL = 1 };

/* One layer of a VGG-shaped topology: Conv3x3(pad 1)+ReLU, or MaxPool 2x2/2 (Caffe ceil mode). */
typedef struct st_layer_desc {
    int kind;            /* ST_LAYER_CONV | ST_LAYER_POOL */
    const char* name;    /* blob/layer name, e.g. "conv1_1" */
    int cin, cout;       /* conv only */
} st_layer_desc;

const char* st_last_error(void);

/* ---- model: replaces CaffeModel.__init__/reload_net (worker.py:36-61) ------------------------- */
/* n_layers == 0 selects models/vgg19.prototxt (16 conv + 5 pool). device_id follows the `gpu`
 * config key (worker.py:328): it is passed to hipSetDevice. */
int st_create(st_ctx** out, int device_id, const st_layer_desc* layers, int n_layers);
int st_destroy(st_ctx* ctx);
/* weights of one conv layer, Caffe layout (Cout, Cin, 3, 3) + bias (Cout)  [caffe.Net(weights=)] */
int st_load_conv_weights(st_ctx* ctx, const char* layer, const float* w, const float* bias);
/* 1 (default; the environment variable ST2_WINO=0 also clears it): fp32 convs whose shape allows it (reduction
 * depth % 8 == 0, >= 48 output channels; any width) run as Winograd F(2x2,3x3) on the fp32 matrix cores --
 * 2.25x fewer multiplies, same IEEE fp32 products and sums in a different association.  0: direct kernel only.
 * 2 (opt-in, round 5): as 1, and where the shape also allows it (reduction depth % 16 == 0, output channels % 64 == 0,
 * width % 4 == 0, tensors below 4 GiB) the transform-domain products are taken on the bf16 matrix cores as the six
 * exact bf16 x bf16 partial products of three-way split fp32 operands, accumulated in fp32 (conv3x3_wino_split.hip):
 * fp32-grade results (closer to a double-precision convolution than algorithm 1's), not bit-identical to them. */
int st_set_conv_algo(st_ctx* ctx, int winograd);
/* 0 (default): fp32 throughout.  1: bf16 feature path (BASELINE config 3) -- conv operands (activations, weights,
 * backward diffs) in bf16 on v_mfma_f32_32x32x16_bf16, fp32 accumulate; Gram, losses, optimizer stay fp32.  Objective
 * evaluations then write an fp32 blob / diff only where something reads fp32 (weighted layers, pools without a fused
 * producer, the 3-channel first layer): st_get_blob reports the others as not materialised.  2: as 1 with every fp32 blob
 * and diff written (same results bit for bit; the A/B reference of the tests; environment ST2_BF16_LEAN=0 forces it). */
int st_set_precision(st_ctx* ctx, int bf16_features);
/* CaffeModel.layers (worker.py:73-75): blob names in network order, "data" first */
int st_num_blobs(st_ctx* ctx);
const char* st_blob_name(st_ctx* ctx, int index);
int st_blob_shape(st_ctx* ctx, int index, int H, int W, int* c, int* h, int* w);

/* ---- model test hooks: CaffeModel.forward / backward (worker.py:77-106) ----------------------- */
/* runs the net on a preprocessed (1,3,H,W) image up to blob `last_blob` (-1: whole net) */
int st_forward(st_ctx* ctx, const float* x_nchw, int H, int W, int last_blob);
int st_get_blob(st_ctx* ctx, int index, float* out_chw);
/* ranged backward with per-blob diff injection after st_forward; diffs[i] is (C,h,w) of blob
 * blob_index[i]; writes d/d(data) as (3,H,W) */
int st_backward(st_ctx* ctx, int n, const int* blob_index, const float* const* diffs, float* out_grad);
/* gram_matrix (worker.py:109-114) of a blob of the last st_forward: out is C*C */
int st_gram(st_ctx* ctx, int index, float* out);

/* ---- image slots: StyleTransfer.set_input / set_content / set_style (worker.py:191-218) ------- */
/* preprocess (worker.py:63-66) + upload.  is_u8: 1 = uint8 HWC, 0 = float32 HWC. */
int st_set_input(st_ctx* ctx, const void* hwc, int H, int W, int is_u8);
/* st_set_content: + forward; keeps the preprocessed image and the features of the blobs that carry a content weight at the time
 * (every blob until the first st_set_weights, like worker.py:204-209).  st_set_weights drops the features no content weight reads;
 * an evaluation whose weight table names a blob without features takes them again from the kept image first (same kernels). */
int st_set_content(st_ctx* ctx, const void* hwc, int H, int W, int is_u8);
int st_set_style(st_ctx* ctx, const void* hwc, int H, int W, int is_u8);     /* + full forward, Gram of every blob */
/* already-preprocessed NCHW variants (resample paths, worker.py:154-170) */
int st_set_input_nchw(st_ctx* ctx, const float* x, int H, int W);
int st_set_content_nchw(st_ctx* ctx, const float* x, int H, int W);
int st_get_input_nchw(st_ctx* ctx, float* out);          /* current x, (3,H,W) */
int st_input_shape(st_ctx* ctx, int* H, int* W);

/* ---- objective: StyleTransfer.set_weights / reset / opfunc (worker.py:172-175,226-301) -------- */
/* rows in DataFrame order; NaN cells allowed (treated as zero, worker.py:234). params = tv,
 * tv_power, p, p_power (messages.py:147). */
int st_set_weights(st_ctx* ctx, int n_rows, const int* blob_index, const float* content,
                   const float* style, const float* deepdream, const double params[4]);
int st_clear_norms(st_ctx* ctx);                          /* the `norms` half of reset() */
/* number of trace scalars an evaluation produces: 6 per active layer + 8 */
int st_trace_len(st_ctx* ctx);
/* evaluates opfunc at the current input; out_grad (3,H,W) may be NULL (return_grad=False) */
int st_opfunc(st_ctx* ctx, float* out_loss, float* out_grad, double* trace);

/* ---- optimizers: optimizers.py:7-125 ------------------------------------------------------------ */
int st_optimizer_reset(st_ctx* ctx, int kind, double step_size);   /* new optimizer instance */
int st_optimizer_set_step(st_ctx* ctx, double step_size);
int st_optimizer_kind(st_ctx* ctx);
int st_objective_changed(st_ctx* ctx);
/* Adam state for the host-side resample path (optimizers.py:29-40); arrays are (3,H,W) */
int st_adam_get_state(st_ctx* ctx, float* m, float* v, int* items1, int* items2);
int st_adam_set_state(st_ctx* ctx, const float* m, const float* v, int items1, int items2);
/* one StyleTransfer.step (worker.py:303-310): optimizer step + deprocess.
 * out_hwc (H,W,3 float32) and trace may be NULL: then nothing is copied back and the call does not
 * synchronise (device-resident loop). */
int st_step(st_ctx* ctx, float* out_hwc, double* trace, float* out_loss);
/* The same iteration in two halves, for the worker loop (worker.py:380-395: step, send Iterate, poll the socket, step ...).
 * st_step_begin queues one StyleTransfer.step and the asynchronous copy of its iterate and trace into pinned host memory (own
 * stream) and returns at once; st_step_end waits for the OLDEST queued iteration and hands it over: *out_hwc points at a
 * (H,W,3) float32 array owned by the context, valid for the next FIVE calls of st_step_begin (six buffers rotate; a sender that
 * is at most four iterates behind can serialise it in place).  At most two iterations may
 * be in flight; while any is, st_step refuses.  Calling begin(k+1) before end(k) lets the GPU compute iteration k+1 while
 * iterate k crosses PCIe and is pickled -- the results are those of st_step, bit for bit. */
int st_step_begin(st_ctx* ctx);
int st_step_end(st_ctx* ctx, const float** out_hwc, int* out_h, int* out_w, double* trace, float* out_loss);
int st_step_pending(st_ctx* ctx);      /* iterations begun and not yet ended (0..2) */
/* Room for a message frame around the iterate (worker.py:351-353 sends messages.Iterate(image, i, trace) as ONE pickle, messages.py:64-74):
 * from the next st_step_begin on, the caller may write `head_bytes` in front of and `tail_bytes` behind the image st_step_end hands out
 * (same pinned allocation, same lifetime), so that the pickle's fixed header, the image the GPU copied in and the variable tail form one
 * contiguous buffer the transport sends without a host-side copy.  At most 1 MiB each; the image stays page-aligned.  Buffers replaced
 * by a re-allocation (a larger input, other rooms) stay valid for the views already handed out, for the same five further begins.
 * Refused (ST_ERR_STATE) while an iteration is in flight: its slot was sized without the room -- collect with st_step_end first. */
int st_step_frame_room(st_ctx* ctx, size_t head_bytes, size_t tail_bytes);
/* Steps so far that ran as a hipGraph replay.  Opt-in (environment ST2_GRAPH=1; ST2_GRAPH_MAX_PX=<edge>, default 768):
 * steady-state Adam steps are captured once per ping-pong parity and replayed, bit-identical to plain launches.  Measured
 * on MI355X it is no faster (the step is bound by the dependent kernels' execution latency), hence off by default. */
int st_graph_replays(st_ctx* ctx, long long* n);
/* Test hook for LBFGSOptimizer.inv_hv (optimizers.py:89-108): p = H g by the device two-loop recursion for a given
 * history of n_pairs <= 10 (s, y) pairs, oldest first, each a (3,H,W) array of the current input geometry with
 * s.y > 1e-10; nothing is applied to the iterate.  Replaces the optimizer's history (the next step starts empty). */
int st_lbfgs_inv_hv(st_ctx* ctx, int n_pairs, const float* const* s, const float* const* y, const float* g, float* out_p);
int st_sync(st_ctx* ctx);

/* ---- measurement ----------------------------------------------------------------------------------- */
/* per-kernel-class HIP-event timing on the engine's own stream */
int st_profile_enable(st_ctx* ctx, int on);
int st_profile_num_classes(void);
const char* st_profile_class_name(int cls);
/* sums since the last call: launches, milliseconds, algorithmic FLOPs and bytes per class */
int st_profile_read(st_ctx* ctx, long long* launches, double* ms, double* flops, double* bytes);

/* ---- device-resident resampling: optimizer.resample / StyleTransfer.resample_content ------------------------
 * (optimizers.py:29-40,110-119; worker.py:154-170; utils.py:130-160 = Pillow Image.resize on float planes).
 * A table describes one axis exactly as Pillow's precompute_coeffs does (host-computed): for output index i the
 * window starts at lo[i], has n[i] taps, coefficients k[i*kmax .. ] (double, normalised). */
typedef struct st_resample_table { const int* lo; const int* n; const double* k; int kmax; int out_size; } st_resample_table;
/* Resamples the optimizer state to (lan_y.out_size x lan_x.out_size): x by Lanczos (or replaced by new_x_nchw when not
 * NULL), Adam's m by Lanczos, Adam's v by bilinear then max(0, .); L-BFGS: x only (the caller then signals
 * objective_changed).  bil_* may be NULL for L-BFGS. */
int st_resample_state(st_ctx* ctx, const st_resample_table* lan_x, const st_resample_table* lan_y,
                      const st_resample_table* bil_x, const st_resample_table* bil_y, const float* new_x_nchw);
/* Resamples the preprocessed content image by Lanczos and recomputes the content features. */
int st_resample_content(st_ctx* ctx, const st_resample_table* lan_x, const st_resample_table* lan_y);
int st_get_content_nchw(st_ctx* ctx, float* out, int* H, int* W);   /* out may be NULL to query the shape */

/* ---- tile-sharded single image (BASELINE config 5; style_transfer2_amd/tiling.py) ------------------------------------
 * One context = one rank's window (tile + apron) of a gH x gW image.  No reference counterpart: the reference
 * runs one image per worker; its numerics (global Gram normalisation worker.py:114, global RMS norms :254,266,275,
 * periodic TV utils.py:232-254) fix what must be reduced/exchanged.  Phases of one Adam iteration; the caller
 * all-reduces the returned device buffers (RCCL) and exchanges gradient / image strips between them:
 *   st_tile_forward -> [AR p1] -> st_tile_losses -> (first eval: st_tile_style_raw -> [AR p2]) -> st_tile_losses_finish
 *   -> st_tile_backward -> [overlap-add window gradient] -> st_tile_update(ring) -> [AR p3] -> [apron refresh] -> st_tile_swap
 * p1 = per active layer {sum d^2, sum gc^2, sum F^2, sum gd^2} (+ C*C raw Gram sums for style layers), p2/p3 tail =
 * sum S^2 per style layer, p3 head = {tv, p, scd^2, tvg^2, pg^2, grad^2} sums over the tile. */
int st_tile_configure(st_ctx* ctx, int gH, int gW, int wy0, int wx0, int ty0, int tx0, int ty1, int tx1);
int st_tile_forward(st_ctx* ctx, float** dev_ptr, int* n_floats);
int st_tile_losses(st_ctx* ctx, float** dev_ptr, int* n_floats);
int st_tile_style_raw(st_ctx* ctx);
int st_tile_losses_finish(st_ctx* ctx);
int st_tile_backward(st_ctx* ctx, float** dev_grad);
int st_tile_update(st_ctx* ctx, const float* ring_dev, float** dev_ptr, int* n_floats);
/* st_tile_update without the optimizer: the combined gradient (network + TV + p-norm) of the tile's pixels is written to the
 * window-sized gradient buffer (st_tile_buffer 4), the six partial sums (+ style-gradient sums) come back as from st_tile_update.
 * The tile-sharded L-BFGS (optimizers.py:62-108 with all-reduced dot products) evaluates its objective through this. */
int st_tile_gradient(st_ctx* ctx, const float* ring_dev, float** dev_ptr, int* n_floats);
/* utils.dot / utils.axpy (utils.py:29-47) on vectors the caller keeps on this device: out_dev[0] = sum a b (this rank's partial
 * sum, to be all-reduced); y = alpha x + y.  Both return when the result is in place. */
int st_vec_dot(st_ctx* ctx, const float* a_dev, const float* b_dev, long long n, float* out_dev);
int st_vec_axpy(st_ctx* ctx, float alpha, const float* x_dev, float* y_dev, long long n);
/* y = float(double(y) / divisor): `p /= np.sqrt(dot(p, p) / p.size)` of optimizers.py:99 divides by a float64 scalar */
int st_vec_div(st_ctx* ctx, double divisor, float* y_dev, long long n);
/* which: 0 current x (3,wh,ww), 1 next x, 2 local sum D^2 per style layer, 3 norms [blob][c,s,d], 4 gradient (3,wh,ww) */
int st_tile_buffer(st_ctx* ctx, int which, float** dev_ptr);
int st_tile_swap(st_ctx* ctx);
/* strips exchanged with ONE neighbour: rects [n][4] = {y0, x0, h, w} (window coordinates, n <= 12) of the (C, wh, ww) device
 * tensor <-> one contiguous device buffer (rect r stored as (C, h_r, w_r), in order).  mode 0 pack, 1 unpack, 2 unpack-add. */
int st_tile_strips(st_ctx* ctx, void* tensor_dev, int C, int wh, int ww, int n, const int* rects, void* buf_dev, int mode);


/* ---- the tile-sharded iteration with its communication inside the engine (BASELINE config 5: "RCCL halo exchange over xGMI") -----
 * The phases above leave the collectives to the caller.  Here the engine owns them: one RCCL communicator per context, the two
 * all-reduces and the three strip exchanges of an Adam iteration are enqueued on the engine's own stream between the compute phases
 * (ncclAllReduce in place on the phase buffers; one grouped ncclSend / ncclRecv per neighbour and phase on packed strip buffers), and
 * the host synchronises once per iteration, to read the trace.  No reference counterpart (the reference runs one image per worker);
 * what must cross ranks is fixed by worker.py:114 (global Gram normalisation), :254,266,275 (global RMS norms) and utils.py:232-254
 * (periodic TV).  The geometry and the exchange plan come from the caller (style_transfer2_amd/tiling.py). */
#define ST_COMM_ID_BYTES 128
int st_comm_unique_id(char out_id[ST_COMM_ID_BYTES]);                       /* ncclGetUniqueId: on rank 0, then handed to every rank */
/* ncclCommInitRank on the context's device.  Preflight first, so that a mis-launch is an error here and not a hang in the first
 * collective: RCCL >= 2.7 (grouped send / recv), every peer of a plan already handed over < world (world = rows x cols of the grid);
 * devices without direct peer access are reported on stderr. */
int st_comm_init(st_ctx* ctx, const char id[ST_COMM_ID_BYTES], int rank, int world);
/* A caller-supplied transport in place of RCCL (the tests run several ranks on ONE GPU, which RCCL refuses): allreduce sums `n` floats
 * in place over the ranks; exchange sends send_buf[i] (send_count[i] floats) to send_peer[i] and fills recv_buf[j] from recv_peer[j].
 * All buffers are device memory; the engine has synchronised its stream before the call and the data must be in place on return. */
typedef int (*st_allreduce_fn)(void* user, float* dev_buf, int n);
typedef int (*st_exchange_fn)(void* user, int n_send, const int* send_peer, float* const* send_buf, const int* send_count,
                              int n_recv, const int* recv_peer, float* const* recv_buf, const int* recv_count);
int st_comm_callbacks(st_ctx* ctx, int rank, int world, st_allreduce_fn allreduce, st_exchange_fn exchange, void* user);
int st_comm_destroy(st_ctx* ctx);
int st_comm_barrier(st_ctx* ctx);                                            /* all-reduce of one float + stream synchronisation */
/* The strips this rank exchanges in one phase.  rects are [n][4] = {y0, x0, h, w}: send rects in the coordinates of the tensor packed
 * (the window), recv rects in those of the tensor unpacked into (the window; the (3, th + 2, tw + 2) ring for ST_TILE_PLAN_RING).
 * A peer equal to the own rank (ring phase only: the image's periodic wrap may land on this rank's own tile) is a local copy;
 * its send and recv rects pair up one to one. */
enum { ST_TILE_PLAN_OVERLAP = 0, ST_TILE_PLAN_RING = 1, ST_TILE_PLAN_REFRESH = 2 };
typedef struct st_tile_peer { int peer; int n_send; const int* send_rects; int n_recv; const int* recv_rects; } st_tile_peer;
int st_tile_plan(st_ctx* ctx, int phase, int n_peers, const st_tile_peer* peers);
/* One iteration of the tile-sharded image with the optimizer st_optimizer_reset chose (after st_tile_configure, st_comm_init /
 * st_comm_callbacks and the three st_tile_plan calls).
 * Adam (optimizers.py:20-27): forward -> all-reduce -> losses (first evaluation: style gradients raw -> all-reduce) -> backward ->
 * overlap-add exchange -> ring exchange -> TV / p-norm / Adam on the tile -> all-reduce -> apron refresh exchange -> swap.
 * L-BFGS (optimizers.py:62-108, the reference's default: worker.py:135-136): every rank keeps its tile of x, of the gradient and of the
 * <= 10 curvature pairs; the direction is formed in the Gram form (coefficients from the matrix of inner products of {s_i, y_i, g}):
 * apply s = -step * H g to the tile -> apron refresh exchange -> the evaluation above with the combined gradient in place of the Adam
 * update -> ONE all-reduce of this step's 2 (2 k + 3) new inner products -> the s.y > 1e-10 gate, eviction and the next coefficients,
 * identically on every rank.  The first step after a reset evaluates twice (optimizers.py:64-65).
 * trace: st_trace_len(ctx) values, the layout of st_step's; NULL: nothing is read back and the call does not wait for the GPU.  The
 * collective sequence is the same either way (round 5: the few KB of image-space sums that only the trace reads are reduced on every
 * call), so ranks may choose NULL independently of each other. */
int st_tile_step(st_ctx* ctx, double* trace);
int st_tile_get_tile(st_ctx* ctx, float* out_hwc);                           /* this rank's tile of the current iterate, (th, tw, 3) deprocessed */

#ifdef __cplusplus
}
#endif
#endif /* ST2_H */
