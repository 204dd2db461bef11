/* st2_probes.h -- C entry points of tools/probes/libst2_probes.so: micro-benchmarks that sized the kernels of
 * libst2_hip.so (matrix-pipe ceiling, operand-feed and issue-rate probes) and an isolated timing hook for the conv
 * kernels.  Development tools: NOT part of the product library or of its C ABI (include/st2.h), never loaded by
 * worker.py / bench.py / the tests of the product path.  Every function returns 0 on success; st_probe_last_error()
 * has the message otherwise. */
#ifndef ST2_PROBES_H
#define ST2_PROBES_H
#ifdef __cplusplus
extern "C" {
#endif
const char* st_probe_last_error(void);
/* isolated timing of the conv3x3 MFMA kernel on one layer shape (K input channels, M output channels, HxW, random
 * data).  cfg < 0: the engine's own tile choice (returned in *cfg_used); cfg >= 100: the Winograd kernel (100: its own
 * choice incl. split-K, 101: 128 ch x 4x32 px, 102: 64 ch x 8x32 px, 104: 64 ch x 8x32 px position-split,
 * 103 / 106 / 105: those with cycle stamps).
 * dgrad_epilogue: 0 = forward (bias + ReLU); 1 = the backward pass's ReLU mask + injected diff, 2 = mask only, 3 = injected diff
 * only, 4 = neither -- what the mask / inject loads of an epilogue cost, as an upper bound on what any cheaper mask can give back. */
int st_bench_conv(int device_id, int K, int M, int H, int W, int cfg, int dgrad_epilogue, int iters,
                  double* avg_ms, int* cfg_used);
int st_conv_num_configs(void);
const char* st_conv_config_name(int cfg);
/* matrix-pipe ceiling probe: variant 0 = register operands, 1 = + LDS operand reads; blocks_per_cu 256-thread
 * workgroups per CU; returns sustained TFLOP/s of v_mfma_f32_32x32x2_f32 */
int st_bench_mfma(int device_id, int variant, int blocks_per_cu, double* tflops);
/* issue-rate probe: shader cycles per v_mfma_f32_32x32x2_f32 at one wave per SIMD with naux VALU + nlds ds_read between MFMAs */
int st_bench_issue_probe(int device_id, int naux, int nlds, double* cycles_per_mfma);
/* feed probe of the LDS-staged-U Winograd design: shader cycles per k-pair (16 MFMAs = 1024 cycles ideal) */
int st_bench_lds_feed_probe(int device_id, int extra_dma, int K, int blocks, double* cycles_per_kpair);
/* Winograd operand-feed probe: executed MFMA TFLOP/s with the A operands streamed L2 -> VGPR; depth = k-pairs in
 * flight: 1, 2, or 12 (= 2 with a staggered k walk).  Depth 4 is rejected: see probes.hip. */
int st_bench_wino_probe(int device_id, int blocks_per_cu, int K, int M, int depth, double* tflops);
/* isolated timing of the bf16 conv launcher with the options of one lean-flow launch.  mode = sum of: 1 out16, 2 fp32 out,
 * 4 fused pool (pool16 + amap), 8 bits_out (forward: bias + ReLU); 16 mask_bits, 32 mask16, 64 unpool, 128 fused style term
 * (any of these: a data-gradient launch) */
int st_bench_conv16(int device_id, int K, int M, int H, int W, int mode, int iters, double* avg_ms);
/* round 5: go / no-go probe of the split-operand Winograd (tools/probes/wino_split_probe.hip): one forward layer K -> M at H x W with the
 * transform-domain products as six bf16 partial products of three-way split fp32 operands; check != 0 compares with a CPU loop nest */
int st_probe_wino_split(int device_id, int K, int M, int H, int W, int iters, int check, double* avg_ms, double* rel_l2,
                        double* loop_cycles, double* clock_mhz, double* pro_cycles, double* epi_cycles);
const char* st_probe_wino_split_error(void);
/* the product kernel (conv3x3_wino_split.hip) through launch_conv3x3_wino_split: mode 0 forward, 1 data gradient, 2 forward + fused pool + map,
 * 3 = 2 without the blob; check: forward against a CPU loop nest */
int st_probe_wino_split_product(int device_id, int K, int M, int H, int W, int iters, int mode, int check, double* avg_ms, double* rel_l2,
                                double* loop_cycles, double* clock_mhz, double* pro_cycles, double* epi_cycles);
/* shader cycles per group of nv VALU instructions of one kind (see wino_split_probe.hip), optionally behind a bf16 MFMA each */
int st_probe_valu_rate(int device_id, int kind, int nv, int with_mfma, double* cycles_per_group);
#ifdef __cplusplus
}
#endif
#endif
