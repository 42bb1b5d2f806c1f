/*
 * pcfa_hip.h -- C-ABI of libpcfa_hip.so, the MI355X (gfx950) kernels behind the
 * PCFA perturbation-optimisation hot path.
 *
 * Everything here is `extern "C"`, takes raw DEVICE pointers + sizes + a
 * hipStream_t (passed as void*), allocates nothing, never synchronises and is
 * re-entrant per stream, so a caller may capture the calls into a hipGraph.
 * All tensors are fp32, contiguous unless strides are passed explicitly.
 * Return value: 0 = PCFA_OK, <0 = argument error (PCFA_ERR_*), >0 = hipError_t
 * of the failed launch.
 *
 * Each entry point names the reference interface (cv-stuttgart/PCFA, paths
 * relative to the reference root) that it replaces.
 */
#ifndef PCFA_HIP_H
#define PCFA_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PCFA_OK 0
#define PCFA_ERR_INVALID_ARG (-1)
#define PCFA_ERR_UNSUPPORTED (-2)
#define PCFA_ERR_WORKSPACE (-3)

/* Exported from libpcfa_hip.so (the library is built with -fvisibility=hidden). */
#if defined(__GNUC__)
#define PCFA_API __attribute__((visibility("default")))
#else
#define PCFA_API
#endif

#define PCFA_MAX_LEVELS 8
#define PCFA_ABI_VERSION 2

/* f_type of the similarity term: helper_functions/losses.py:145-174 */
#define PCFA_LOSS_AEE 0
#define PCFA_LOSS_MSE 1
#define PCFA_LOSS_COSIM 2

/* box constraint: attack_PCFA.py:20-29, helper_functions/own_models.py:72-80 */
#define PCFA_BOX_CLIPPING 0
#define PCFA_BOX_CHANGE_OF_VARIABLES 1

PCFA_API int pcfa_abi_version(void);
/* Measurement hook.  Queues two caller-owned hipEvent_t for the nth kernel (0 = next) that the CALLING THREAD
 * launches through this library after the first arm of a batch (up to 8 pairs may be queued).  That kernel is
 * dispatched with hipExtLaunchKernel, which attaches the events to its dispatch packet: after completion
 * hipEventElapsedTime(start, stop) is the kernel's own duration -- the timestamps rocprofv3's kernel trace
 * reads -- without the two barrier packets an event bracket around a launch adds (about 7 us on MI355X).
 * Entry points that launch several kernels: pcfa_corr_pyramid_bwd = {GEMM dfmap1, reduce, GEMM df2ext, reduce,
 * pooling adjoint}, pcfa_flow_loss_fwd = {partials, final}, pcfa_spatial_corr_bwd = {grad_in1, grad_in2},
 * pcfa_flownet_corr_bwd = {grad_in1, grad_in2}, pcfa_resample2d_bwd = {clear grad_in1, scatter + flow gradient},
 * pcfa_instnorm_fwd / _bwd = {partial sums, apply}, pcfa_pwc_warp_bwd = {clear, scatter + flow gradient}.
 * nth < 0 drops every pair still queued (events of kernels that were never launched stay unrecorded). */
PCFA_API int pcfa_timing_arm(void* start_event, void* stop_event, int nth);

/* Launches an empty kernel on `stream`: lets a caller calibrate the fixed cost of bracketing one launch with
 * HIP events (bench.py reports it next to the per-launch timings). */
PCFA_API int pcfa_null_launch(void* stream);
/* Test aid: fills the LDS of every CU with `pattern` (2048 workgroups that each own a CU's 160 KB for a moment).  LDS is
 * not cleared between workgroups, so a kernel that reads LDS it never wrote sees whatever the previous tenant left: its
 * own earlier workgroups when it runs alone (run-to-run identical), another pair's kernels with two pairs in flight.
 * tests/test_gpu_parity.py launches this before every kernel of a closure with two different patterns and demands
 * identical bits (no entry point of the attack path calls it). */
PCFA_API int pcfa_poison_lds(unsigned pattern, void* stream);
/* Its control: n workgroups each store one word of an LDS array they never wrote into out[0..n) -- after pcfa_poison_lds
 * they must report the pattern. */
PCFA_API int pcfa_peek_lds(unsigned* out, int n, void* stream);
/* Calibration of the two roofs on the box the benchmark runs on (bench.py `calibration`; nothing in the attack path
 * calls them).  pcfa_calib_mfma_f32: `blocks` workgroups of four waves each issue `iters` x 4 independent
 * v_mfma_f32_32x32x2_f32 from registers (no memory, no LDS); returns the flop issued (< 0 on error) -- divide by the
 * launch's duration.  pcfa_calib_copy: dst = src as a float4 grid-stride stream (2 x 4 x n_floats bytes of traffic). */
PCFA_API long long pcfa_calib_mfma_f32(float* scratch, int blocks, int iters, void* stream);
PCFA_API int pcfa_calib_copy(const float* src, float* dst, long long n_floats, void* stream);
/* Human-readable text for a return code (static storage). */
PCFA_API const char* pcfa_status_string(int status);

/* ------------------------------------------------------------------------- *
 * RAFT / GMA all-pairs correlation pyramid.
 * Replaces CorrBlock.__init__ + CorrBlock.corr  (models/raft/corr.py:12-27,52-60;
 * identical models/gma/corr.py:15-30,55-63).
 *
 * HBM layout ("slab per query"): the 4D volume and its pooled levels are kept
 * as ONE matrix  pyr[B*Q][slab]  (Q = H*W), row q holding level 0, level 1
 * ((H/2)x(W/2)), ... back to back.  Inside a row every level is stored as
 * 4x4-texel tiles (64 B), tiles row-major, texels row-major inside a tile,
 * levels padded to whole tiles (pad texels are 0):
 *   index(l, y, x) = off_l + ((y>>2)*ceil(W_l/4) + (x>>2))*16 + (y&3)*4 + (x&3)
 * pcfa_corr_tiled_index() evaluates it.  Element (q, index(l,y,x)) is the
 * reference's corr_pyramid[l][q, 0, y, x].
 * ------------------------------------------------------------------------- */

/* floats per query row (multiple of 16) */
PCFA_API long long pcfa_corr_slab_floats(int H, int W, int num_levels);
/* offset (floats) of level l inside a row; *h_l,*w_l receive its extent */
PCFA_API long long pcfa_corr_level_offset(int H, int W, int num_levels, int level, int* h_l, int* w_l);

/* index of texel (y, x) of level l inside a query row, or -1 */
PCFA_API long long pcfa_corr_tiled_index(int H, int W, int num_levels, int level, int y, int x);

/* f2ext[B][D][slab] = fmap2 and its successively 2x2-average-pooled copies
 * (floor semantics of F.avg_pool2d(.,2,stride=2)), columns in the tiled order
 * above; tile-padding columns are 0. */
PCFA_API int pcfa_corr_f2ext_fwd(const float* fmap2, float* f2ext, int B, int D, int H, int W,
                        int num_levels, void* stream);

/* pyr[b*Q+q][n] = sum_d fmap1[b][d][q] * f2ext[b][d][n] / sqrt(D)
 * (fp32-input MFMA, k-ordered fma chain).  fmap1: [B][D][Q]. */
PCFA_API int pcfa_corr_pyramid_fwd(const float* fmap1, const float* f2ext, float* pyr, int B, int D,
                          int H, int W, int num_levels, void* stream);

/* Backward of the two calls above.  dpyr: [B*Q][slab] gradient w.r.t. pyr (as
 * accumulated by pcfa_corr_lookup_bwd).  Writes dfmap1,dfmap2 [B][D][Q].
 * workspace: device scratch of >= pcfa_corr_pyramid_bwd_workspace_bytes(). */
PCFA_API size_t pcfa_corr_pyramid_bwd_workspace_bytes(int B, int D, int H, int W, int num_levels);
PCFA_API int pcfa_corr_pyramid_bwd(const float* dpyr, const float* fmap1, const float* f2ext,
                          float* dfmap1, float* dfmap2, void* workspace, size_t workspace_bytes,
                          int B, int D, int H, int W, int num_levels, void* stream);
/* The same backward with the lookups' coordinates: dpyr is zero outside the (2r+2)^2 windows its lookups touched
 * (CorrBlock.__call__, models/raft/corr.py:29-50 -- 12 lookups per RAFT forward, all near the diagonal of the 4D
 * volume), so both products skip the slab-column / query-row ranges no window reaches (per 128-wide block, bounding
 * rows of the windows of ALL given lookups, one texel of slack: conservative for any coordinates).  `coords`: host
 * array of n_coords (<= 32) device pointers to [B][2][H][W] coordinate tensors -- every lookup whose backward
 * accumulated into dpyr; n_coords = 0 (or misaligned operands) computes the dense products of pcfa_corr_pyramid_bwd.
 * Results are those of the dense form up to the summation order of the split-K partials (the skipped terms are exact
 * zeros); deterministic.  workspace >= pcfa_corr_pyramid_bwd_windows_workspace_bytes(). */
PCFA_API size_t pcfa_corr_pyramid_bwd_windows_workspace_bytes(int B, int D, int H, int W, int num_levels);
PCFA_API int pcfa_corr_pyramid_bwd_windows(const float* dpyr, const float* fmap1, const float* f2ext, float* dfmap1,
                                  float* dfmap2, void* workspace, size_t workspace_bytes,
                                  const float* const* coords, int n_coords, int radius, int B, int D, int H, int W,
                                  int num_levels, void* stream);

/* CorrBlock.__call__ (models/raft/corr.py:29-50) + bilinear_sampler
 * (models/raft/utils/utils.py:57-71): coords [B][2][H][W] (x then y, pixel
 * units of level 0), out [B][L*(2r+1)^2][H][W]; channel l*(2r+1)^2 + a*(2r+1) + b
 * samples level l at (x = cx/2^l + a - r, y = cy/2^l + b - r), bilinear,
 * zeros outside (the reference's x-major window order). */
PCFA_API int pcfa_corr_lookup_fwd(const float* pyr, const float* coords, float* out, int B, int H, int W,
                         int num_levels, int radius, void* stream);

/* dpyr += d out / d pyr ^T * grad_out   (no gradient w.r.t. coords: the
 * reference detaches them, models/raft/raft.py:122-123).  Deterministic: every
 * (query, level) window is owned by one workgroup, no atomics.  dpyr must have
 * been zeroed by the caller before the first accumulation of a backward pass. */
PCFA_API int pcfa_corr_lookup_bwd(float* dpyr, const float* coords, const float* grad_out, int B, int H,
                         int W, int num_levels, int radius, void* stream);

/* GMA attention products and softmax (models/gma/gma.py:34-77 Attention, :79-115 Aggregate).
 * pcfa_gemm_f32: C[b][m][n] = alpha * sum_k A(m,k) B(k,n) on the fp32 matrix cores (the pyramid's GEMM core, exact
 * fp32 products).  a_kmajor 0: A stored [M][K] (lda = row stride), 1: stored [K][M]; b_kmajor 0: B stored [N][K],
 * 1: stored [K][N]; (a_kmajor, b_kmajor) = (1, 0) is PCFA_ERR_UNSUPPORTED.  splits > 1: split-K through `workspace`
 * (pcfa_gemm_f32_workspace_bytes) with an ordered reduction; C must then be dense (ldc = N).
 *   sim = alpha * q k^T           (0,0)      out = attn v              (0,1)
 *   dv  = attn^T g                (1,1)      d_attn = [g_1|..|g_6] [v_1|..|v_6]^T   (0,0), ONE product for all iterations
 * pcfa_softmax_rows_fwd / _bwd: y = softmax(x) along the last axis, gx = y * (gy - sum_j gy_j y_j); a row stays in
 * registers: one read and one write of the [rows][cols] matrix per direction; x / y and gy / gx may alias. */
PCFA_API size_t pcfa_gemm_f32_workspace_bytes(int M, int N, int batch, int splits);
PCFA_API int pcfa_gemm_f32(const float* A, const float* B, float* C, int M, int N, int K, long long lda, long long ldb,
                  long long ldc, int a_kmajor, int b_kmajor, int batch, long long bsA, long long bsB, long long bsC,
                  float alpha, int splits, void* workspace, size_t workspace_bytes, void* stream);
PCFA_API int pcfa_softmax_rows_fwd(const float* x, float* y, long long rows, int cols, void* stream);
PCFA_API int pcfa_softmax_rows_bwd(const float* y, const float* grad_y, float* grad_x, long long rows, int cols,
                          void* stream);

/* Lookup fused with the motion encoder's first layer:
 *   out = relu?(convc1(CorrBlock.__call__(coords)))   models/raft/corr.py:29-50 feeding models/raft/update.py:79-93
 *   (convc1 = Conv2d(4*81, 256, 1)); identical in models/gma.
 * The [324][Q] lookup result never reaches memory: taps are blended into LDS and multiplied with the weights on the
 * fp32 matrix cores in the same workgroup (32 queries x 4 levels x 256 channels); the backward multiplies
 * grad_out * [out > 0] with the transposed weights and scatters straight into dpyr (same ownership as
 * pcfa_corr_lookup_bwd: no atomics).  `packed` = pcfa_lookup_convc1_pack_weights(weight[256][324]) (both operand
 * orders, pcfa_lookup_convc1_packed_floats(256) floats).  num_levels = 4, radius = 4, Cout = 256 only
 * (PCFA_ERR_UNSUPPORTED otherwise: compose pcfa_corr_lookup_* with a convolution).
 * pyr / dpyr / coords as for pcfa_corr_lookup_fwd / _bwd; out, grad_out: [B][256][H][W]. */
PCFA_API long long pcfa_lookup_convc1_packed_floats(int Cout);
PCFA_API int pcfa_lookup_convc1_pack_weights(const float* weight, float* packed, int Cout, int Cin, void* stream);
PCFA_API int pcfa_lookup_convc1_fwd(const float* pyr, const float* coords, const float* packed, const float* bias,
                           float* out, int B, int H, int W, int num_levels, int radius, int Cout, int relu,
                           void* stream);
PCFA_API int pcfa_lookup_convc1_bwd(float* dpyr, const float* coords, const float* packed, const float* out,
                           const float* grad_out, int B, int H, int W, int num_levels, int radius, int Cout,
                           int relu, void* stream);

/* ------------------------------------------------------------------------- *
 * PWC-Net cost volume = spatial_correlation_sampler_backend.forward/backward
 * (models/PWCNet/cpu_spatial_correlation_sampler-0.3.0/Correlation_Module/
 *  correlation_sampler.cpp:58-81,83-112 ; CPU math correlation.cpp:9-37,39-73,
 *  75-124,126-178).  Same 12 integers in the same order.
 * in1,in2: [B][C][iH][iW];  out / grad_out: [B][patchH][patchW][oH][oW] with
 * oH = (iH + 2*padH - ((kH-1)*dilH+1))/dH + 1 (same for W).
 * The caller allocates out / grad_in1 / grad_in2 (the kernels fully overwrite
 * them; no pre-zeroing needed).
 * ------------------------------------------------------------------------- */
PCFA_API int pcfa_spatial_corr_out_size(int iH, int iW, int kH, int kW, int padH, int padW, int dilH,
                               int dilW, int dH, int dW, int* oH, int* oW);
PCFA_API int pcfa_spatial_corr_fwd(const float* in1, const float* in2, float* out, int B, int C, int iH,
                          int iW, int kH, int kW, int patchH, int patchW, int padH, int padW,
                          int dilH, int dilW, int dil_patchH, int dil_patchW, int dH, int dW,
                          void* stream);
PCFA_API int pcfa_spatial_corr_bwd(const float* in1, const float* in2, const float* grad_out,
                          float* grad_in1, float* grad_in2, int B, int C, int iH, int iW, int kH,
                          int kW, int patchH, int patchW, int padH, int padW, int dilH, int dilW,
                          int dil_patchH, int dil_patchW, int dH, int dW, void* stream);

/* PWC-Net's use of the sampler with its two elementwise followers fused in:
 *   out = leaky_relu(correlate(in1, in2), slope),  correlate = 9x9 cost volume * scale (scale = 1/C)
 * (models/PWCNet/PWCNet.py:45-58 `correlate`, then self.leakyRELU at :249,264,278,292,308).  k = 1, patch 9,
 * stride 1, pad 0; out / grad_out: [B][81][iH][iW].  pcfa_cost_volume9_bwd reads the forward's output for the
 * LeakyReLU mask and applies mask * scale to the gradient taps while staging them: ONE launch for both gradients.
 * Needs iW % 4 == 0 and 16-B aligned pointers (PCFA_ERR_UNSUPPORTED otherwise: the caller composes
 * pcfa_spatial_corr_* with elementwise kernels instead). */
PCFA_API int pcfa_cost_volume9_fwd(const float* in1, const float* in2, float* out, int B, int C, int iH, int iW,
                          float scale, float slope, void* stream);
PCFA_API int pcfa_cost_volume9_bwd(const float* in1, const float* in2, const float* fwd_out, const float* grad_out,
                          float* grad_in1, float* grad_in2, int B, int C, int iH, int iW, float scale,
                          float slope, void* stream);

/* ------------------------------------------------------------------------- *
 * FlowNet2's native operators (CUDA-only extensions in the
 * reference, no CPU build exists there).
 *
 * Correlation -- replaces correlation_cuda.forward / .backward
 * (models/FlowNet/correlation_package/correlation_cuda.cc:10-87,89-167; kernels
 * correlation_cuda_kernel.cu:74-147,150-333; Python side correlation.py:10-67).
 * Same five integers in the same order (corr_multiply is ignored by the reference
 * kernels and has no counterpart).  in1,in2: [B][C][H][W];
 * out / grad_out: [B][D*D][oH][oW], D = 2*(max_displacement/stride2)+1,
 * oH = ceil((H + 2*pad_size - 2*((kernel_size-1)/2 + max_displacement)) / stride1):
 *   out[b][tj*D+ti][y][x] = 1/(k*k*C) * sum_{j,i,c} P1[c][y*s1+md+j][x*s1+md+i]
 *                                                 * P2[c][y*s1+md+(tj-D/2)*s2+j][x*s1+md+(ti-D/2)*s2+i]
 * with P = input zero-padded by pad_size.  The reference materialises P1/P2 as
 * channels-last copies (rbot1/rbot2, caller-provided and resize_d); here padding is
 * a load predicate and the two scratch tensors do not exist.  Outputs are fully
 * overwritten (no pre-zeroing).  Backward: stride1 == 1 only (PCFA_ERR_UNSUPPORTED
 * otherwise; FlowNetC.py:31-35 is the only call site and uses 20,1,20,1,2).
 *
 * Resample2d -- replaces resample2d_cuda.forward / .backward
 * (resample2d_package/resample2d_cuda.cc:6-24, resample2d_kernel.cu:16-72,75-201).
 * in1: [B][C][iH][iW]; flow: [B][2][H][W] (x, y displacement in pixels);
 * out / grad_out: [B][C][H][W].  kernel_size must be 1 (the value FlowNet2 uses;
 * larger values read past the border in the reference) and H <= iH, W <= iW.
 * pcfa_resample2d_bwd zeroes grad_in1 itself (a fill kernel on `stream`, then the
 * scatter kernel) and scatters with hardware fp32 atomics like the reference's atomicAdd, so the last
 * bits of grad_in1 depend on the order of arrival; grad_flow is a gather.
 *
 * ChannelNorm -- replaces channelnorm_cuda.forward / .backward
 * (channelnorm_package/channelnorm_cuda.cc:6-25, channelnorm_kernel.cu:18-60,63-96).
 * in: [B][C][plane]; out: [B][1][plane] = sqrt(sum_c in^2);
 * grad_in = grad_out * in / (out + 1e-9).  norm_deg must be 2 (the reference ignores it).
 * ------------------------------------------------------------------------- */
PCFA_API int pcfa_flownet_corr_out_size(int H, int W, int pad_size, int kernel_size, int max_displacement,
                                        int stride1, int stride2, int* out_channels, int* oH, int* oW);
PCFA_API int pcfa_flownet_corr_fwd(const float* in1, const float* in2, float* out, int B, int C, int H, int W,
                                   int pad_size, int kernel_size, int max_displacement, int stride1, int stride2,
                                   void* stream);
PCFA_API int pcfa_flownet_corr_bwd(const float* in1, const float* in2, const float* grad_out, float* grad_in1,
                                   float* grad_in2, int B, int C, int H, int W, int pad_size, int kernel_size,
                                   int max_displacement, int stride1, int stride2, void* stream);
PCFA_API int pcfa_resample2d_fwd(const float* in1, const float* flow, float* out, int B, int C, int iH, int iW,
                                 int H, int W, int kernel_size, int bilinear, void* stream);
PCFA_API int pcfa_resample2d_bwd(const float* in1, const float* flow, const float* grad_out, float* grad_in1,
                                 float* grad_flow, int B, int C, int iH, int iW, int H, int W, int kernel_size,
                                 int bilinear, void* stream);
PCFA_API int pcfa_channelnorm_fwd(const float* in, float* out, int B, int C, long long plane, int norm_deg,
                                  void* stream);
PCFA_API int pcfa_channelnorm_bwd(const float* in, const float* out, const float* grad_out, float* grad_in, int B,
                                  int C, long long plane, int norm_deg, void* stream);

/* ------------------------------------------------------------------------- *
 * Attack math (fused elementwise + reductions).
 * ------------------------------------------------------------------------- */

/* ScaledInputModel.forward prologue (helper_functions/own_models.py:62-85):
 *   x = image (+ delta broadcast over batch, delta may be NULL)
 *   if cov:  x = 0.5/(1-eps) * (tanh(x) + (1-eps))
 *   x = clamp(x, 0, 1);  x *= scale  (scale = 255 if make_unit_input else 1)
 * image,out: [B][chw]; delta: [chw] or NULL.  `n_per_item` = C*H*W. */
PCFA_API int pcfa_box_transform_fwd(const float* image, const float* delta, float* out, int B,
                           long long n_per_item, int change_of_variables, double eps_box,
                           float scale, void* stream);
/* grad_image [B][chw] (may be NULL) and grad_delta [chw] (may be NULL; summed
 * over the batch in a fixed order) from grad_out. */
PCFA_API int pcfa_box_transform_bwd(const float* image, const float* delta, const float* grad_out,
                           float* grad_image, float* grad_delta, int B, long long n_per_item,
                           int change_of_variables, double eps_box, float scale, void* stream);

/* RAFT.forward's input normalisation `2 * (image / 255.0) - 1.0` of both images (models/raft/raft.py:88-89,
 * models/gma/network.py:79-80) in one launch per direction: pair = [n(image1); n(image2)] ([2B][n], the feature encoder's
 * batch), ctx = n(image1) ([B][n], the context encoder's input); backward: grad_image1 = ((grad_pair[:B] + grad_ctx) * 2) / 255,
 * grad_image2 = (grad_pair[B:] * 2) / 255 (grad_ctx may be NULL).  Bit-identical to the torch expression on the GPU. */
PCFA_API int pcfa_pm1_pair_fwd(const float* image1, const float* image2, float* pair, float* ctx, int B, long long n,
                               void* stream);
PCFA_API int pcfa_pm1_pair_bwd(const float* grad_pair, const float* grad_ctx, float* grad_image1, float* grad_image2, int B,
                               long long n, void* stream);

/* extract_deltas (attack_PCFA.py:20-29): delta = box(nw_input) - image. */
PCFA_API int pcfa_extract_deltas_fwd(const float* nw_input, const float* image, float* delta,
                            long long n, int change_of_variables, double eps_box, void* stream);
PCFA_API int pcfa_extract_deltas_bwd(const float* nw_input, const float* grad_delta, float* grad_nw_input,
                            long long n, int change_of_variables, double eps_box, void* stream);

/* extract_deltas_joint (attack_PCFA.py:32-37):
 *   up = clamp(nw_delta + imax, 0, 1) - imax ; delta = clamp(up + imin, 0, 1) - imin */
PCFA_API int pcfa_extract_deltas_joint_fwd(const float* nw_delta, const float* images_max,
                                  const float* images_min, float* delta, long long n,
                                  void* stream);
PCFA_API int pcfa_extract_deltas_joint_bwd(const float* nw_delta, const float* images_max,
                                  const float* images_min, const float* grad_delta,
                                  float* grad_nw_delta, long long n, void* stream);

/* loss_delta_constraint (helper_functions/losses.py:200-230) =
 *   get_loss(f_type, pred, target) + mu * relu((|d1|^2+|d2|^2)/(n1+n2) - bound^2)
 * pred/target: [B][2][H][W] views given by element strides (sb,sc,sh,sw) so the
 * un-padded crop of the network output is read in place (no copy).
 * out_scalars (8 floats): [0]=loss, [1]=similarity term, [2]=mean squared delta,
 * [3..5] = cosim sums (p.t, p.p, t.t), [6] = mean squared delta - bound^2, [7] unused.
 * workspace >= pcfa_flow_loss_workspace_bytes().
 * delta2 may alias delta1 (joint perturbation; then both count, as in the reference). */
PCFA_API size_t pcfa_flow_loss_workspace_bytes(void);
PCFA_API int pcfa_flow_loss_fwd(const float* pred, const long long pred_strides[4], const float* target,
                       const long long target_strides[4], int B, int H, int W,
                       const float* delta1, long long n1, const float* delta2, long long n2,
                       float delta_bound, float mu, int f_type, float* out_scalars,
                       void* workspace, void* stream);
/* grad_pred is written DENSE [B][2][H][W]; grad_delta1/2 [n1]/[n2] may be NULL.
 * If delta2 aliases delta1 pass grad_delta2 = NULL and joint=1 (gradient doubled). */
PCFA_API int pcfa_flow_loss_bwd(const float* pred, const long long pred_strides[4], const float* target,
                       const long long target_strides[4], int B, int H, int W,
                       const float* delta1, long long n1, const float* delta2, long long n2,
                       float mu, int f_type, int joint, const float* fwd_scalars,
                       const float* grad_loss, float* grad_pred, float* grad_delta1,
                       float* grad_delta2, void* stream);

/* ------------------------------------------------------------------------- *
 * SepConvGRU gate arithmetic (models/raft/update.py:45-60 == models/gma/update.py:51-66), fused:
 *   gates : z = sigmoid(zc), r = sigmoid(rc), rh = r * h            (zc, rc = convz(hx), convr(hx))
 *   update: q = tanh(qc), hnew = (1 - z) * h + z * q                (qc = convq(cat[rh, x]))
 * and their backward passes.  All arrays have n floats (NCHW, plane = H*W, `channels` channels) ; 16-byte aligned
 * arrays take the vectorised path, anything else a scalar loop.  zc/rc/qc are the convolution outputs WITHOUT bias; bias_z/r/q ([channels], may be NULL) are
 * added here, which saves the separate bias-add launch torch would issue per convolution.  add_z/r/q (n floats,
 * may be NULL) are added to the pre-activations as well: the gate convolutions are linear in their input
 * [h | inp | motion], and `inp` does not change over the refinement iterations, so its contribution
 * conv(inp, W[:, inp-slice]) is computed once per forward and enters every iteration through these pointers.
 * ------------------------------------------------------------------------- */
PCFA_API int pcfa_gru_gates_fwd(const float* zc, const float* rc, const float* h, const float* bias_z,
                       const float* bias_r, const float* add_z, const float* add_r, float* z, float* r,
                       float* rh, long long n, int plane, int channels, void* stream);
PCFA_API int pcfa_gru_gates_bwd(const float* z, const float* r, const float* h, const float* dz, const float* drh,
                       float* dzc, float* drc, float* dh, long long n, void* stream);
/* pcfa_gru_gates_bwd with dh = dh_in + drh * r: dh_in (nullable) is the gradient that reached h on another path
 * (the (1 - z) * g term of pcfa_gru_update_bwd), folded in here instead of by a separate add kernel. */
PCFA_API int pcfa_gru_gates_bwd_acc(const float* z, const float* r, const float* h, const float* dz, const float* drh,
                           const float* dh_in, float* dzc, float* drc, float* dh, long long n, void* stream);
PCFA_API int pcfa_gru_update_fwd(const float* z, const float* qc, const float* h, const float* bias_q,
                        const float* add_q, float* q, float* hnew, long long n, int plane, int channels,
                        void* stream);
PCFA_API int pcfa_gru_update_bwd(const float* z, const float* q, const float* h, const float* g, float* dz,
                        float* dqc, float* dh, long long n, void* stream);

/* The 5-tap gate convolutions of SepConvGRU -- Conv2d(c, 128, (1,5), padding=(0,2)) and ((5,1), padding=(2,0)),
 * reference models/raft/update.py:36-42 (models/gma/update.py:36-42), applied at :45-60 -- without bias, as an
 * implicit fp32-MFMA GEMM:
 *     out[b,co,y,x] = sum_{t<5} sum_{ci<Ca+Cb} w_packed[t][ci][co] * in[b,ci, y+dy, x+dx],   zero outside the image,
 *     (dy,dx) = (0,t-2) if vertical == 0 (1x5), (t-2,0) if vertical != 0 (5x1),
 * where in = [in_a | in_b] concatenated along channels and read in place (in_b may be NULL with Cb = 0): the
 * reference's torch.cat([h, x]) is never materialised.  in_a [B,Ca,H,W], in_b [B,Cb,H,W], out [B,Cout,H,W],
 * all contiguous fp32.  w_packed [5][Ca+Cb][Cout] comes from pcfa_sepconv5_pack_weights.
 *
 * The data gradient of the operator is the operator itself on grad_out with the `bwd` packing:
 *     grad_in = pcfa_sepconv5_fwd(grad_out, Cout, NULL, 0, bwd_packed, ..., Cout := Ca+Cb, ...),
 * channels [0,Ca) of the result belonging to in_a and the rest to in_b.  (No weight gradient: the attack
 * freezes the network, attack_PCFA.py:573-574.)
 *
 * pcfa_sepconv5_pack_weights: w [Cout][Cin][5] (the Conv2d weight, either orientation, flattened) ->
 * fwd_packed [5][Cin][Cout] and/or bwd_packed [5][Cout][Cin] (either may be NULL). */
/* Floats of one packing: the direct operand order [5][Cin][Cout] followed, where the shape is eligible (Cin % 8 == 0),
 * by the Winograd-domain weights of the F(2,5) kernel (csrc/sepconv5_wino.hip: 6 points, MFMA operand order).
 * fwd_packed must hold pcfa_sepconv5_packed_floats(Cout, Cin) floats, bwd_packed (the operator with the channel roles
 * swapped) pcfa_sepconv5_packed_floats(Cin, Cout); every pcfa_sepconv5_* launch reads buffers of that length. */
PCFA_API long long pcfa_sepconv5_packed_floats(int Cout, int Cin);
/* Process-wide algorithm switch of every pcfa_sepconv5_* launch: 1 = 1-D Winograd F(2,5) where the shape is eligible
 * (default; PCFA_SEPCONV_WINO=0 in the environment starts with 0), 0 = the direct implicit GEMM everywhere, negative =
 * query only.  Returns the previous setting.  Eligible: Cin % 32 == 0 (% 64 for narrow grids), Cout % 32 == 0,
 * W % 128 == 0 (1x5) / W % 64 == 0 (5x1), 16-B aligned tensors. */
PCFA_API int pcfa_sepconv5_algo(int use_winograd);
/* 1 when a launch of this shape takes the Winograd kernel under the current setting (aligned tensors assumed). */
PCFA_API int pcfa_sepconv5_uses_winograd(int B, int Ca, int Cb, int Cout, int H, int W, int vertical);
PCFA_API int pcfa_sepconv5_pack_weights(const float* w, float* fwd_packed, float* bwd_packed, int Cout, int Cin,
                               void* stream);
PCFA_API int pcfa_sepconv5_fwd(const float* in_a, int Ca, const float* in_b, int Cb, const float* w_packed,
                      float* out, int B, int Cout, int H, int W, int vertical, void* stream);

/* pcfa_sepconv5_fwd with the output channels split over two tensors -- channels [0, Cout_a) to out_a
 * [B,Cout_a,H,W], the rest to out_b [B,Cout-Cout_a,H,W] -- and optional accumulation (out += result) per tensor.
 * Used for the data gradient: grad of [h | x] lands directly in grad_h and grad_x, summed in place with the
 * gradients those tensors receive from their other consumers (SepConvGRU uses h three times and x four times per
 * step, models/raft/update.py:45-60). */
PCFA_API int pcfa_sepconv5_fwd_split(const float* in_a, int Ca, const float* in_b, int Cb, const float* w_packed,
                            float* out_a, int Cout_a, int accumulate_a, float* out_b, int accumulate_b, int B,
                            int Cout, int H, int W, int vertical, void* stream);
/* The same, and the b part of the result (channels >= Cout_a) additionally passes through a deferred ReLU backward:
 * after the (accumulating) write, channels < mask_channels of out_b are zeroed where mask_b <= 0 (mask_b: the tensor
 * whose ReLU it is, shape of out_b).  Lets the motion encoder's last ReLU (models/raft/update.py:101) be differentiated
 * by the GRU's backward, which produces that gradient anyway, instead of by a launch of its own.  Must be the LAST
 * write of out_b. */
PCFA_API int pcfa_sepconv5_fwd_split_masked(const float* in_a, int Ca, const float* in_b, int Cb, const float* w_packed,
                                   float* out_a, int Cout_a, int accumulate_a, float* out_b, int accumulate_b,
                                   const float* mask_b, int mask_channels, int B, int Cout, int H, int W,
                                   int vertical, void* stream);

/* SepConvGRU half-step with the gate arithmetic in the convolution's epilogue (models/raft/update.py:45-60; replaces
 * pcfa_sepconv5_fwd + pcfa_gru_gates_fwd resp. + pcfa_gru_update_fwd; C % 32 == 0):
 *   gates : z = sigmoid(conv_z([h | rest]) + add_zr[:, :C]),  r = sigmoid(conv_r(..) + add_zr[:, C:]),  rh = r * h
 *           with the stacked weight w_packed [5][C + Cr][2C];
 *   update: q = tanh(conv_q([rh | rest]) + add_q),  hnew = (1 - z) * h + z * q,   w_packed [5][C + Cr][C].
 * add_* = the pre-activation contribution of the constant context features, bias included ([B][2C or C][H][W]). */
PCFA_API int pcfa_sepconv5_gru_gates_fwd(const float* h, int C, const float* rest, int Cr, const float* w_packed,
                                const float* add_zr, float* z, float* r, float* rh, int B, int H, int W,
                                int vertical, void* stream);
PCFA_API int pcfa_sepconv5_gru_update_fwd(const float* rh, int C, const float* rest, int Cr, const float* w_packed,
                                 const float* add_q, const float* z, const float* h, float* q, float* hnew, int B,
                                 int H, int W, int vertical, void* stream);
/* The backward counterparts (data gradients through the same convolutions with the elementwise backward in the
 * epilogue; w_packed_bwd = the second packing of pcfa_sepconv5_pack_weights; C % 32 == 0):
 *   gates_bwd : [drh | d_rest] = conv_q'(dqc); drh is consumed at once by pcfa_gru_gates_bwd_acc's arithmetic:
 *               dzr[:, :C] = dz (1-z) z,  dzr[:, C:] = (drh h)(1-r) r,  dh = dh_in + drh r  (dh_in may be NULL or dh);
 *               d_rest is written (accumulate_rest = 0) or accumulated.
 *   update_bwd: [ddh | d_rest] = conv_zr'(dzr); g = dh_acc + ddh is the gradient of the PREVIOUS half-step's output and
 *               goes straight through pcfa_gru_update_bwd's arithmetic with that half-step's (z, q, h):
 *               dz_prev = g q - g h,  dqc_prev = (g z)(1 - q q),  dh_prev = g (1 - z);  d_rest accumulates. */
PCFA_API int pcfa_sepconv5_gru_gates_bwd(const float* dqc, int C, int Cr, const float* w_packed_bwd, const float* z,
                                const float* r, const float* h, const float* dz, const float* dh_in, float* dzr,
                                float* dh, float* d_rest, int accumulate_rest, int B, int H, int W, int vertical,
                                void* stream);
PCFA_API int pcfa_sepconv5_gru_update_bwd(const float* dzr, int C, int Cr, const float* w_packed_bwd, const float* dh_acc,
                                 const float* z_prev, const float* q_prev, const float* h_prev, float* dz_prev,
                                 float* dqc_prev, float* dh_prev, float* d_rest, int B, int H, int W, int vertical,
                                 void* stream);

/* 3x3 / stride 1 / pad 1 convolution (the update-block convolutions, models/raft/update.py:6-16,79-101) as Winograd
 * F(2x2,3x3) on the fp32 matrix cores, bias and ReLU fused:  out[b,n] = act(bias[n] + sum_k w[n,k] (*) x[b,k]).
 * pcfa_conv3x3_pack_weights: w [Cout][Cin][3][3] -> fwd_packed [16][Cin][pad64(Cout)] = G w G^T and/or
 * bwd_packed [16][Cout][pad64(Cin)] (flipped, transposed: the data gradient is pcfa_conv3x3_fwd(grad_out, bwd_packed,
 * NULL, grad_in, B, K = Cout, N = Cin, ...)); each needs pcfa_conv3x3_packed_floats(K, N) floats, 16-B aligned.
 * pcfa_conv3x3_fwd: x [B][K][H][W], out [B][N][H][W], bias [N] or NULL, relu != 0 applies max(., 0).
 * No weight gradient (the attack freezes the network). */
PCFA_API long long pcfa_conv3x3_packed_floats(int K, int N);
PCFA_API int pcfa_conv3x3_pack_weights(const float* w, float* fwd_packed, float* bwd_packed, int Cout, int Cin,
                              void* stream);
PCFA_API int pcfa_conv3x3_fwd(const float* x, const float* packed, const float* bias, float* out, int B, int K, int N,
                     int H, int W, int relu, void* stream);
/* The same with a choice of activation: act 0 = none, 1 = ReLU, 2 = LeakyReLU(slope) (PWC-Net's conv(),
 * models/PWCNet/PWCNet.py:29-35); pcfa_leaky_relu_bwd: grad_x = grad_out * (out > 0 ? 1 : slope). */
PCFA_API int pcfa_conv3x3_act_fwd(const float* x, const float* packed, const float* bias, float* out, int B, int K,
                         int N, int H, int W, int act, float slope, void* stream);
PCFA_API int pcfa_leaky_relu_bwd(const float* out, const float* grad_out, float* grad_x, float slope, long long n,
                        void* stream);
/* Two independent pcfa_conv3x3_act_fwd problems over the same H x W (batch 1) in ONE launch: the motion encoder's
 * convc2(cor1) and convf2(flo1) (models/raft/update.py:92,94).  Each alone leaves the last round of workgroups partly
 * empty on the 256 CUs; together the second problem's workgroups fill the first one's tail.  K % 8 == 0 must agree
 * between the two (one kernel instance). */
PCFA_API int pcfa_conv3x3_act_fwd_pair(const float* x, const float* packed, const float* bias, float* out, int K, int N,
                              const float* x2, const float* packed2, const float* bias2, float* out2, int K2, int N2,
                              int H, int W, int act, float slope, void* stream);
/* pcfa_conv3x3_fwd without bias / activation whose result is zeroed where mask <= 0 (mask: shape of out).  As a data
 * gradient (packed = bwd_packed) with mask = the convolution's own input this is conv'(g) * [x > 0]: the ReLU backward
 * of the layer that produced x (models/raft/update.py:92,94 convc2 / convf2 feeding conv), fused into the epilogue. */
/* The general data-gradient form: grad_in = conv'(g) [zeroed where mask <= 0] [+ addend]; mask and addend are optional
 * (NULL) and have grad_in's shape.  addend = the gradient the same tensor receives from its other consumer -- the
 * residual path of ResidualBlock (models/raft/extractor.py:50-58) -- summed in the epilogue instead of by autograd. */
PCFA_API int pcfa_conv3x3_fused_bwd(const float* g, const float* packed_bwd, const float* mask, const float* addend,
                           float* grad_in, int B, int K, int N, int H, int W, void* stream);
PCFA_API int pcfa_conv3x3_masked_fwd(const float* x, const float* packed, const float* mask, float* out, int B, int K,
                            int N, int H, int W, void* stream);
/* The general form of all of the above, with a choice of transform (pcfa_amd/csrc/conv3x3_f43.hip):
 *   out = act(bias + conv(x))  [x slope where mask <= 0]  [+ addend]        (bias, mask, addend optional)
 * `packed` holds both transforms of the weights (pcfa_conv3x3_pack_weights: [F(2x2,3x3) | F(4x4,3x3)]).  Shapes where
 * it was measured faster (pcfa_conv3x3_algo: W % 4 == 0, 16-B aligned tensors, maps of >= 100k pixels, or >= 24 x 64
 * pixels with K * N >= 192 * 256) run as Winograd F(4x4,3x3) -- 36 products per 16 outputs, 1.78x fewer matrix
 * instructions than F(2x2,3x3); fp32 rounding error ~2e-6 relative per layer instead of ~3.5e-7 -- everything else as
 * F(2x2,3x3).  Where one launch would leave CUs idle (RAFT's 55 x 128 maps: 56-112
 * workgroups) the input channels are split over several workgroups that write partial outputs into `workspace`
 * (pcfa_conv3x3_workspace_bytes, caller-owned, 16-B aligned, may be NULL when that returns 0) and a streaming kernel
 * adds them in index order: deterministic, no atomics.  Small maps of one or two images (PWC-Net's 6 x 20 .. 24 x 80
 * levels: 3-48 workgroups that would walk 16-79 channel chunks one after the other) run F(2x2,3x3) with the input
 * channels sliced over workgroups the same way (partial outputs in `workspace`, the same finish pass). */
PCFA_API int pcfa_conv3x3_algo(int B, int K, int N, int H, int W);   /* 23 or 43: the transform pcfa_conv3x3_run picks */
PCFA_API size_t pcfa_conv3x3_workspace_bytes(int B, int K, int N, int H, int W);
/* mask (act = 0 only; same shape as out) is the OUTPUT of the (Leaky)ReLU layer that produced the tensor this data
 * gradient belongs to: the result is multiplied by 1 where mask > 0 and by `slope` elsewhere (0: ReLU, an exact zero) --
 * that layer's activation backward, deferred into this epilogue.  mask_channels = 0: every channel, BEFORE the addend;
 * mask_channels = m > 0: channels [0, m) only, AFTER the addend (PWC-Net's dense decoder blocks, PWCNet.py:234-323: the
 * gradient of a block layer's output is complete once its last consumer -- the next layer's data gradient, which adds
 * into the running gradient of the block buffer -- has run). */
PCFA_API int pcfa_conv3x3_run(const float* x, const float* packed, const float* bias, const float* mask,
                     const float* addend, float* out, int B, int K, int N, int H, int W, int act, float slope,
                     int mask_channels, void* workspace, size_t workspace_bytes, void* stream);

/* out = relu(x + bias[c]) and its backward gx = grad_out * (out > 0): the "conv -> +bias -> ReLU" tail of the
 * motion encoder / flow head convolutions (models/raft/update.py:12-16,91-101) in one pass. */
PCFA_API int pcfa_bias_relu_fwd(const float* x, const float* bias, float* out, long long n, int plane, int channels,
                       void* stream);
PCFA_API int pcfa_relu_bwd(const float* out, const float* grad_out, float* grad_x, long long n, void* stream);
/* grad_x = grad_out * [out > 0] and grad_x2 = grad_x * [out2 > 0] in one pass: the backward of relu(a + relu(c)) towards
 * a and towards c's pre-activation (ResidualBlock, models/raft/extractor.py:50-58, with out2 = relu(c)). */
PCFA_API int pcfa_relu_bwd2(const float* out, const float* out2, const float* grad_out, float* grad_x, float* grad_x2,
                   long long n, void* stream);

/* act(conv2d(x, w, bias, stride=1, padding=ksize/2)) for Cin <= 4 input channels, forward only: convf1 of the
 * motion encoder (models/raft/update.py:79-101, Conv2d(2, 128, 7, padding=3) + ReLU on the detached flow).
 * x: [B][Cin][H][W]; w: [N][Cin][k][k]; out: [B][N][H][W].  Supported (Cin, ksize): (2,7) (1,7) (2,5) (2,3);
 * anything else returns PCFA_ERR_UNSUPPORTED (the caller keeps the library convolution). */
PCFA_API int pcfa_conv_fewin_fwd(const float* x, const float* w, const float* bias, float* out, int B, int Cin, int N,
                                 int H, int W, int ksize, int relu, void* stream);
/* The same layer with the frozen weight packed once in MFMA operand order (pcfa_conv_fewin_packed_floats(Cin, N, ksize)
 * floats, 16-B aligned): operands go from L2 straight into registers, only the flat input range is staged in LDS,
 * 32-pixel tiles.  Same results as pcfa_conv_fewin_fwd (same products, same k order). */
PCFA_API long long pcfa_conv_fewin_packed_floats(int Cin, int N, int ksize);
PCFA_API int pcfa_conv_fewin_pack(const float* w, float* packed, int Cin, int N, int ksize, void* stream);
PCFA_API int pcfa_conv_fewin_packed_fwd(const float* x, const float* packed, const float* bias, float* out, int B, int Cin,
                               int N, int H, int W, int ksize, int relu, void* stream);

/* act(conv2d(x, w, bias, stride=2, padding=ksize/2)) for a frozen weight on the fp32 matrix cores: the encoders' stem
 * Conv2d(3, 64, 7, stride=2, padding=3) (models/raft/extractor.py:118, models/gma/extractor.py:118) and the 3x3 / stride-2
 * first convolution of the down-sampling residual blocks (extractor.py:23-58 with stride=2).  x: [B][Cin][H][W],
 * w: [N][Cin][k][k], out: [B][N][Ho][Wo] with Ho = (H + 2 (k/2) - k) / 2 + 1.  pcfa_conv_s2_supported: (Cin, k) = (3, 7) or
 * k = 3, W % 4 == 0, 16-B aligned x -- anything else is PCFA_ERR_UNSUPPORTED (the caller keeps the library convolution).
 * packed: pcfa_conv_s2_packed_floats(Cin, N, ksize) floats written once by pcfa_conv_s2_pack (MFMA operand order).
 * act: 0 none, 1 ReLU, 2 LeakyReLU(slope). */
PCFA_API int pcfa_conv_s2_supported(int Cin, int N, int ksize, int H, int W);
PCFA_API long long pcfa_conv_s2_packed_floats(int Cin, int N, int ksize);
PCFA_API int pcfa_conv_s2_pack(const float* w, float* packed, int Cin, int N, int ksize, void* stream);
PCFA_API int pcfa_conv_s2_fwd(const float* x, const float* packed, const float* bias, float* out, int B, int Cin, int N,
                              int H, int W, int ksize, int act, float slope, void* stream);
/* Data gradient of the stride-2 convolutions above: grad_x[B][Cin][H][W] (overwritten) from grad_out[B][N][Ho][Wo].
 * ksize = 3: four stride-1 convolutions of grad_out by output parity (one MFMA per (p, q) and channel pair, nothing
 * scattered, no layout transposes).  (Cin, ksize) = (3, 7), the stem: a 4x4-tap convolution of grad_out onto the 12
 * (channel, parity) rows on the 16x16x4 fp32 MFMA (the library: GEMM into a 147 x pixels matrix + col2im).
 * W % 8 == 0 (PCFA_ERR_UNSUPPORTED otherwise: the caller keeps the library gradient); packed:
 * pcfa_conv_s2_bwd_packed_floats() floats written once by pcfa_conv_s2_bwd_pack from w[N][Cin][k][k]. */
PCFA_API int pcfa_conv_s2_bwd_supported(int Cin, int N, int ksize, int H, int W);
PCFA_API long long pcfa_conv_s2_bwd_packed_floats(int Cin, int N, int ksize);
PCFA_API int pcfa_conv_s2_bwd_pack(const float* w, float* packed, int Cin, int N, int ksize, void* stream);
PCFA_API int pcfa_conv_s2_bwd(const float* grad_out, const float* packed, float* grad_x, int B, int Cin, int N, int H, int W,
                              int ksize, void* stream);
/* The entry of a down-sampling residual block (models/raft/extractor.py:23-58 with stride = 2) in one launch per direction:
 *   out   = act(conv2d(x, w,  bias,   stride=2, padding=1))     w:  [N][Cin][3][3]   (conv1)
 *   out_d =     conv2d(x, wd, bias_d, stride=2)                 wd: [N][Cin][1][1]   (downsample[0]; its norm follows)
 * The 1x1 convolution reads the centre tap of the 3x3 window: two more MFMAs per four input channels on operands that are
 * already staged.  Backward: grad_x = conv1^T(grad_out) + downsample^T(grad_out_d) in the (even row, even column) parity
 * class of pcfa_conv_s2_bwd -- the separate gradient add over x disappears.  Support as pcfa_conv_s2_supported(Cin, N, 3,
 * H, W) / pcfa_conv_s2_bwd_supported; packed buffers from the _ds_pack / _ds_bwd_pack entry points. */
PCFA_API long long pcfa_conv_s2_ds_packed_floats(int Cin, int N);
PCFA_API int pcfa_conv_s2_ds_pack(const float* w, const float* wd, float* packed, int Cin, int N, void* stream);
PCFA_API int pcfa_conv_s2_ds_fwd(const float* x, const float* packed, const float* bias, float* out, const float* bias_d,
                                 float* out_d, int B, int Cin, int N, int H, int W, int act, float slope, void* stream);
PCFA_API long long pcfa_conv_s2_ds_bwd_packed_floats(int Cin, int N);
PCFA_API int pcfa_conv_s2_ds_bwd_pack(const float* w, const float* wd, float* packed, int Cin, int N, void* stream);
PCFA_API int pcfa_conv_s2_ds_bwd(const float* grad_out, const float* grad_out_d, const float* packed, float* grad_x, int B,
                                 int Cin, int N, int H, int W, void* stream);

/* PWC-Net's backward warp (models/PWCNet/PWCNet.py:166-206) as one pass per direction:
 *   out = grid_sample(x, normalise(meshgrid + flo)) * (grid_sample(ones, ...) >= mask_threshold)
 * bilinear, zero padding, align_corners = False, the reference's fp32 coordinate arithmetic (normalise by W-1, then
 * grid_sample's un-normalisation by W).  x, out, grad_*: [B][C][H][W]; flo, grad_flo: [B][2][H][W].
 * pcfa_pwc_warp_bwd = {clear grad_x and grad_flo, scatter (hardware fp32 atomics, like grid_sampler_2d_backward) +
 * flow gradient}.  flow_scale: the warp is taken along flow_scale * flo (PWCNet.py:262,276,290,306 pass `up_flow * 0.625`
 * ... `* 5.0`): the product is rounded to fp32 before use and grad_flo = flow_scale * (gradient of the scaled flow), i.e.
 * bit for bit what the two element-wise launches of `warp(x, up_flow * s)` and its backward give; 1.0f = plain flo. */
PCFA_API int pcfa_pwc_warp_fwd(const float* x, const float* flo, float* out, int B, int C, int H, int W,
                               float mask_threshold, float flow_scale, void* stream);
PCFA_API int pcfa_pwc_warp_bwd(const float* x, const float* flo, const float* grad_out, float* grad_x,
                               float* grad_flo, int B, int C, int H, int W, float mask_threshold, float flow_scale,
                               void* stream);
/* The same backward, bit-reproducible: the scatter adds fixed-point int64 values (integer adds commute, so the
 * order in which the atomics land does not matter) and the flow gradient's channel groups are summed in index order;
 * grid_sampler_2d_backward (and pcfa_pwc_warp_bwd) add fp32 values in whatever order the hardware serves them, which
 * made two 20-step PWC-Net attacks on the same pair end 3 % apart (profiles/r03_schedule_parity_pwcnet_20steps.json).
 * The scatter goes through a per-workgroup LDS window (16 x 16 pixel tiles, 32 x 32 texels, 64-bit LDS adds) that is
 * flushed with one global atomic per non-zero texel; taps outside the window use global atomics directly.
 * The fixed point is scaled per call to max|grad_out| (40 bits below its leading power of two, room for 2^22 addends
 * of maximal size).  workspace >= pcfa_pwc_warp_bwd_det_workspace_bytes(), 8-B aligned.  {clear + max, scatter, finish} */
PCFA_API size_t pcfa_pwc_warp_bwd_det_workspace_bytes(int B, int C, int H, int W);
PCFA_API int pcfa_pwc_warp_bwd_det(const float* x, const float* flo, const float* grad_out, float* grad_x,
                          float* grad_flo, void* workspace, size_t workspace_bytes, int B, int C, int H, int W,
                          float mask_threshold, float flow_scale, void* stream);

/* 3x3 / stride 1 / pad 1 convolution with N <= 4 output channels and its data gradient (frozen weights): the
 * flow-prediction layers -- FlowHead.conv2 of RAFT / GMA (models/raft/update.py:6-14), predict_flow of PWC-Net
 * (models/PWCNet/PWCNet.py:37-38) and FlowNet2 (models/FlowNet/submodules.py:33-34).  A stream over the input
 * (HBM-bound), not matrix-core work.  x / grad_x: [B][K][H][W]; w: [N][K][3][3] as nn.Conv2d stores it;
 * bias: [N] or NULL; out / grad_out: [B][N][H][W]; workspace: pcfa_conv3x3_fewout_workspace_bytes() bytes (0 for
 * large planes; small planes split the channel sum over workgroups: {partial sums, reduce}).  Fixed summation
 * order (bitwise reproducible). */
PCFA_API size_t pcfa_conv3x3_fewout_workspace_bytes(int B, int K, int N, int H, int W);
PCFA_API int pcfa_conv3x3_fewout_fwd(const float* x, const float* w, const float* bias, float* out, void* workspace,
                                     int B, int K, int N, int H, int W, void* stream);
/* addend ([B][K][H][W] or NULL): grad_x = data gradient + addend -- the gradient another consumer of x produced (PWC-Net:
 * upfeat reads the same decoder output as predict_flow, PWCNet.py:256-257), summed here instead of by a separate pass. */
PCFA_API int pcfa_conv3x3_fewout_bwd(const float* grad_out, const float* w, const float* addend, float* grad_x, int B,
                                     int K, int N, int H, int W, void* stream);

/* ConvTranspose2d(K, N, kernel_size=4, stride=2, padding=1) with N <= 4 output channels and its data gradient (frozen
 * weights): PWC-Net's `deconv` layers -- deconv6..2 (2 -> 2 channels) and upfeat6..3 (529..661 -> 2 channels)
 * (models/PWCNet/PWCNet.py:42-43, :107-147, used :259-304).  Same stream over the input as pcfa_conv3x3_fewout_*: the
 * 2x2 output block of an input pixel reads the pixel's 3x3 neighbourhood.  x / grad_x: [B][K][H][W]; w: [K][N][4][4]
 * as nn.ConvTranspose2d stores it; bias: [N] or NULL; out / grad_out: [B][N][2H][2W]; workspace:
 * pcfa_deconv4s2_fewout_workspace_bytes() bytes (0 unless the plane is small and K large: {partial sums, reduce}).
 * Fixed summation order (bitwise reproducible; the library chose its algorithm per process). */
PCFA_API size_t pcfa_deconv4s2_fewout_workspace_bytes(int B, int K, int N, int H, int W);
PCFA_API int pcfa_deconv4s2_fewout_fwd(const float* x, const float* w, const float* bias, float* out, void* workspace,
                                       int B, int K, int N, int H, int W, void* stream);
PCFA_API int pcfa_deconv4s2_fewout_bwd(const float* grad_out, const float* w, float* grad_x, int B, int K, int N, int H,
                                       int W, void* stream);

/* out = mul * nn.Upsample(scale_factor=factor, mode='bilinear')(in) (align_corners=False) and its backward:
 * `20 * self.upsample(flow2)` of PWC-Net (models/PWCNet/PWCNet.py:73,321).  in / grad_in: [planes][H][W];
 * out / grad_out: [planes][factor*H][factor*W].  The backward is a gather (ATen's scatters with fp32 atomics):
 * bitwise reproducible. */
PCFA_API int pcfa_upsample_bilinear_fwd(const float* in, float* out, int planes, int H, int W, int factor, float mul,
                                        void* stream);
PCFA_API int pcfa_upsample_bilinear_bwd(const float* grad_out, float* grad_in, int planes, int H, int W, int factor,
                                        float mul, void* stream);

/* InstanceNorm2d without affine parameters on batch statistics, fused with the ReLU that follows it: every
 * `relu(norm(conv(x)))` / `norm(conv(x))` of the feature encoder (models/raft/extractor.py:23-58 with
 * norm_fn='instance', :118-157; nn.InstanceNorm2d defaults: eps 1e-5, biased variance).
 *   x, y, grad_out, grad_x: [planes][plane] (planes = B*C, plane = H*W);  mean_rstd: [planes][2] written by the
 *   forward, read by the backward;  workspace: pcfa_instnorm_workspace_bytes(planes, plane) bytes, 16-B aligned.
 *   forward : y = relu?((x - mean) * rstd), rstd = 1/sqrt(var + eps)
 *   backward: grad_x = rstd * (g - mean(g) - xhat * mean(g * xhat)),  g = grad_out * (xhat > 0) if relu else grad_out
 * Two launches each {partial sums, apply}; sums are accumulated in fp64 and combined in index order (bitwise
 * reproducible).  pcfa_add_relu_fwd: out = relu(a + b), the block output of ResidualBlock.forward
 * (extractor.py:50-58); its backward for both operands is pcfa_relu_bwd(out, grad_out, ...). */
PCFA_API size_t pcfa_instnorm_workspace_bytes(int planes, long long plane);
PCFA_API int pcfa_instnorm_fwd(const float* x, float* y, float* mean_rstd, void* workspace, int planes,
                               long long plane, float eps, int relu, void* stream);
PCFA_API int pcfa_instnorm_bwd(const float* x, const float* mean_rstd, const float* grad_out, float* grad_x,
                               void* workspace, int planes, long long plane, int relu, void* stream);
PCFA_API int pcfa_add_relu_fwd(const float* a, const float* b, float* out, long long n, void* stream);
/* RAFT / GMA convex upsampling (models/raft/raft.py:72-83 upsample_flow; models/gma/network.py likewise):
 *   out[n, c, 8h + i, 8w + j] = sum_k softmax_k(mask[n, k*64 + i*8 + j, h, w]) * 8 * flow[n, c, h + k/3 - 1, w + k%3 - 1]
 * flow [N][2][H][W], mask [N][576][H][W] (the 0.25-scaled logits), out [N][2][8H][8W]; zero padding outside the map.
 * One streaming launch forward; backward = one launch for grad_mask + a 9-tap gather for grad_flow through a
 * workspace of pcfa_convex_upsample_workspace_floats(N, H, W) floats (no atomics: bitwise reproducible). */
PCFA_API int pcfa_convex_upsample_fwd(const float* flow, const float* mask, float* out, int N, int H, int W, void* stream);
PCFA_API long long pcfa_convex_upsample_workspace_floats(int N, int H, int W);
PCFA_API int pcfa_convex_upsample_bwd(const float* flow, const float* mask, const float* grad_out, float* grad_flow,
                             float* grad_mask, float* workspace, int N, int H, int W, void* stream);

/* One refinement iteration's coordinate bookkeeping (models/raft/raft.py:122-137, models/gma/network.py likewise):
 * coords1_new = coords1 + delta,  flow_new = coords1_new - coords0;  n floats each. */
PCFA_API int pcfa_flow_step(const float* coords1, const float* delta, const float* coords0, float* coords1_new,
                   float* flow_new, long long n, void* stream);

/* out = srcs[0] + ... + srcs[n-1], n <= 16 device pointers in a HOST array, summed in index order by one launch.
 * Backward of a tensor read by every refinement iteration (the hoisted gate pre-activations of SepConvGRU,
 * models/raft/update.py:45-60: autograd would add the twelve contributions pairwise, eleven launches per tensor). */
PCFA_API int pcfa_sum_n(const float* const* srcs, int n, float* out, long long numel, void* stream);

/* Metric helpers (helper_functions/losses.py:3-30,129-142): out[0] = sum over
 * pixels of sqrt(du^2+dv^2) / (B*H*W);  pcfa_sum_squares: out[0] = sum x^2. */
PCFA_API int pcfa_avg_epe(const float* flow1, const long long strides1[4], const float* flow2,
                 const long long strides2[4], int B, int H, int W, float* out, void* workspace,
                 void* stream);
PCFA_API int pcfa_sum_squares(const float* x, long long n, float* out, void* workspace, void* stream);

/* ------------------------------------------------------------------------- *
 * L-BFGS vector math of the attack loop: the memory update and the two-loop
 * recursion inside torch.optim.LBFGS.step (torch==1.7.1 pinned by the reference,
 * scripts/requirements.txt:2; call sites attack_PCFA.py:97,114,382,388 with
 * max_iter=10 and no line search).  Same operation sequence as the optimiser's
 * Python loop, 2m+1 launches instead of 4m+3, no host round trip inside.
 *   pair:      y = g - g_prev ; s = t*d ; scal4 = { y.s, y.y, 1/(y.s), (y.s)/(y.y) } ;
 *              g_prev = g when update_prev != 0.  y_out / s_out are the candidate history
 *              slot (the caller keeps it only if y.s > 1e-10, as the optimiser does).
 *   direction: d = -H_k g by the two-loop recursion over the `count` newest pairs of the ring
 *              S, Y [capacity][ld] (pair k-th oldest lives in row (first + k) % capacity),
 *              ro[capacity] = 1/(y.s) per row, H = device scalar (y.s)/(y.y) of the newest
 *              pair, al[count] scratch.  workspace: pcfa_lbfgs_workspace_floats() floats.
 * All vectors 16-B aligned, ld % 4 == 0. */
PCFA_API size_t pcfa_lbfgs_workspace_floats(void);
PCFA_API int pcfa_lbfgs_pair(const float* g, float* g_prev, const float* d, float t, float* y_out, float* s_out,
                             float* scal4, float* workspace, int update_prev, long long n, void* stream);
PCFA_API int pcfa_lbfgs_direction(const float* g, const float* S, const float* Y, const float* ro, const float* H,
                                  float* al, float* d, float* workspace, int first, int count, int capacity,
                                  long long ld, long long n, void* stream);

/* The same optimiser step in the coefficient space of the history ("Gram form", pcfa_amd/csrc/lbfgs_gram.hip): the
 * two-loop recursion of torch.optim.LBFGS.step (attack_PCFA.py:97,114,382,388) is two triangular substitutions on the
 * inner products s_i.y_j, y_i.y_j, s_i.g, y_i.g, so an iteration reads the history twice instead of 2m+1 times:
 *   update:    y = g - g_prev, s = t*d into the candidate row (first + count) % (capacity + 1), g_prev = g; inner
 *              products of g and of the new pair with every stored vector (one sweep); the curvature test
 *              y.s > 1e-10 of the optimiser is taken ON THE DEVICE: an accepted pair is committed to the ring
 *              (the oldest one dropped at `capacity` pairs), H = (y.s)/(y.y); then the coefficients of
 *              d = cg*g + sum_r (cS[r]*S_r + cY[r]*Y_r) are solved in fp64 by one workgroup.
 *   direction: forms d with that combination (one sweep), out_gtd_dmax = { g.d, max|d| }.
 * `state` (pcfa_lbfgs_gram_state_bytes, device memory, 16-B aligned) holds the ring position, H, the inner-product
 * matrices and the coefficients; it begins with { int first, count, accepted, rows; float H, cg, ys, yy, gtd, dmax }
 * which the caller may read back.  pcfa_lbfgs_gram_reset empties the history (first = count = 0, H = 1).
 * S, Y: [capacity + 1][ld] rings; every vector has ld floats (ld % 4 == 0, pad elements zero), 16-B aligned.
 * capacity <= 128 (PCFA_ERR_UNSUPPORTED above: use the two-loop entry points).  Deterministic: no atomics. */
PCFA_API size_t pcfa_lbfgs_gram_state_bytes(int capacity);
PCFA_API size_t pcfa_lbfgs_gram_workspace_bytes(int capacity, long long ld);
PCFA_API int pcfa_lbfgs_gram_reset(void* state, int capacity, void* stream);
PCFA_API int pcfa_lbfgs_gram_update(const float* g, float* g_prev, const float* d, float t, float* S, float* Y,
                                    void* state, void* workspace, int capacity, long long ld, void* stream);
PCFA_API int pcfa_lbfgs_gram_direction(const float* g, const float* S, const float* Y, void* state, float* d,
                                       float* out_gtd_dmax, void* workspace, int capacity, long long ld, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* PCFA_HIP_H */
