/* wsi_hip.h - C ABI of libwsi_hip.so: the gfx950 (MI355X) replacement for the per-patch CNN
 * inference hot path of acproject/wsi-segmentation-pipeline.
 *
 * The reference is pure Python and has no FFI of its own; these entry
 * points are what a reference-side binding for this path would bind (ctypes stub: INTEGRATION.md).
 * Each one cites the reference code it replaces (paths relative to the reference root).
 *
 * Conventions
 *   - plain pointers and sizes only; device pointers are raw HIP device addresses
 *   - every device function takes a hipStream_t (passed as void*), enqueues asynchronously and
 *     returns 0 or a negative errno (-22 EINVAL bad shape/argument, -14 EFAULT launch failure);
 *     nothing throws, nothing allocates: workspaces are caller-owned
 *   - planes = 2 ("parity"): fp16 hi + fp16 lo operand pair (r05; r01-r04: a bf16 pair), three MFMA passes, fp32 accumulate: 22
 *                  significand bits per operand where both halves are normal fp16 numbers, 2^-25 absolute where lo is subnormal;
 *                  packed weights carry one power-of-two scale per output channel (largest |weight| of a channel in [2^13, 2^14)), so
 *                  weights of any magnitude fit.  Max |logit - reference| on the five reference-generated margin families
 *                  (tests/golden/margin_*.npz, tests/test_gpu_margin.py): 2.0e-5 (bf16 pair: 4.0e-4); 5e-6 on the bench tiles; dense
 *                  per-pixel U-Net logits at |logit| 16: 1.2e-4 (bf16 pair: 1.01e-3, over the 1e-3 contract).  Activations are
 *                  clamped to the fp16 range (+-65504) when a conv writes them, as in mode 3.
 *     planes = 3 ("mx", the default): fp16 main pass + MX-fp6 (e2m3) block-scaled cross terms, one E8M0 scale per 32 channels
 *                  and plane: three MFMA instructions per step instead of six.  4.6e-4 on the same five families (|logit| up
 *                  to 16), 1e-4 on the default fixtures: inside the 1e-3 contract everywhere it was measured (profiles/r05_margin_families.json);
 *                  2.2e-3 on the dense per-pixel path: NOT a contract mode there.  (r01-r02 shipped fp4 cross terms in the same line:
 *                  1.9e-3 on two of the families.)
 *     planes = 1 ("speed"): single-pass bf16, logit error ~2e-2: roofline studies only, never a
 *                  contract mode
 *   - "PF" = padded-flat activation layout, see wsi_pf_* below and DESIGN.md
 */
#ifndef WSI_HIP_H
#define WSI_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define WSI_HIP_ABI_VERSION 6            /* r05: planes 2 = fp16 pair, its packed conv weights end in cout inverse channel scales; wsi_unet_decoder_weights.tail_w */
int wsi_hip_abi_version(void);

/* ---- padded-flat layout helpers (host) -------------------------------------------------------
 * pixel (n,y,x) lives at pixel index (W+2) + n*(H+1)*(W+1) + y*(W+1) + x; each pixel holds
 * C channels at 2 (planes 1) or 4 (planes 2, 3) bytes per channel, in 128-byte lines.  wsi_pf_bytes = allocation size; a PF buffer must be zero-filled once before
 * first use (kernels never write the pad positions). */
size_t wsi_pf_bytes(int n, int h, int w, int c, int planes);
long long wsi_pf_pixel_index(int n, int y, int x, int h, int w);

/* ---- weight prepack (host, CPU memory in and out) -------------------------------------------
 * Folds eval-mode BatchNorm (resnets_shift.py:42,45,124; eps 1e-5) into the conv weights and
 * emits them in per-lane MFMA operand order.  bn_* may be NULL (no BN: scale 1, bias 0).
 *   conv: w OIHW fp32 [cout][cin][k][k], k in {1,3}, cout % 32 == 0, cin a whole number of 128-byte lines (cin % 64 == 0 for
 *         planes 1, cin % 32 == 0 for planes 2 / 3; 32-channel tensors are served by the stride-1 3x3 convolution only)
 *         wpk_out: wsi_prepack_conv_bytes() bytes (planes 2: the fragment blocks, then cout floats = the inverse of the power-of-two
 *         scale each output channel's weights were multiplied by; the kernels apply it before the bias); bias_out: cout floats
 *   stem: w [64][3][7][7] (resnets_shift.py:122)  */
size_t wsi_prepack_conv_bytes(int cout, int cin, int k, int planes);
int wsi_prepack_conv(const float* w, const float* bn_weight, const float* bn_bias, const float* bn_mean,
                     const float* bn_var, float eps, int cout, int cin, int k, int planes, void* wpk_out,
                     float* bias_out);
size_t wsi_prepack_stem_bytes(int planes);
int wsi_prepack_stem(const float* w, const float* bn_weight, const float* bn_bias, const float* bn_mean,
                     const float* bn_var, float eps, int planes, void* wpk_out, float* bias_out);
/* Stem weights for u8 slide input (planes >= 2): ToTensor + Normalize (utils/preprocessing.py:209-212) and BN are folded into
 * fixed-point weights (three balanced base-256 digits = 24 bits; one scale per output channel), so
 * the kernel runs the 7x7 convolution in INTEGER arithmetic (v_mfma_i32_32x32x32_i8) on the bytes x - 128 plus an "inside"
 * byte that makes zero padding exact: no per-pixel table look-up, exact i32 accumulation, one MFMA pass per digit.
 * wpk_out: wsi_prepack_stem_bytes(2) bytes (opaque); bias_out: 64 floats. */
int wsi_prepack_stem_u8(const float* w, const float* bn_weight, const float* bn_bias, const float* bn_mean,
                        const float* bn_var, float eps, const float mean[3], const float std_[3], int planes,
                        void* wpk_out, float* bias_out);
/* 3x256 table of (u8/255 - mean[c]) / std[c] evaluated in fp32 exactly like torchvision
 * ToTensor + Normalize (utils/preprocessing.py:209-212, myargs.py:127-130). */
int wsi_normalize_u8_lut(const float mean[3], const float std_[3], float* lut_out /* [3][256] */);

/* ---- stem (resnets_shift.py:196-199 conv1+bn1+relu+maxpool) ---------------------------------
 * Input either f32 NCHW [n][3][h][w] (in_f32 != NULL) or a u8 RGB slide + tile corners + LUT
 * (fused utils/dataset.py:174-178 read + transform).  h % 16 == 0, w % 4 == 0.
 * scratch: n*(h/2)*(w/2)*64 floats.  out_pf: PF (h/4, w/4, 64). */
int wsi_stem_conv7x7_bn_relu_maxpool(const float* in_f32, const uint8_t* slide, long long slide_pitch_bytes,
                                     int slide_h, int slide_w, const int* tile_xy, const float* lut,
                                     const void* stem_wpk, const float* stem_bias,
                                     const void* stem_wpk_u8 /* wsi_prepack_stem_u8 output or NULL */,
                                     const float* stem_bias_u8, const float* norm_mean_std /* HOST: mean[3], std[3] */,
                                     int n, int h, int w, float* scratch, void* out_pf, int planes, void* stream);

/* tuning / A-B hook: fused = 1 (default) runs the single fused stem+maxpool kernel (no fp32
 * intermediate; scratch unused), fused = 0 the two-kernel form, fused = 2 the fused kernel with the table
 * look-up arithmetic even when u8 weights are supplied, fused = 3 the integer stem in its one-strip launch form (digit
 * planes in registers instead of shared in LDS; bit-identical); rows_per_seg = pooled rows per workgroup of the
 * fused kernel (default 32).  Process-wide. */
int wsi_stem_set_mode(int fused, int rows_per_seg);

/* ---- conv + folded BN (+ residual) (+ ReLU) (resnets_shift.py:49-65, 19-27) -------------------
 * in_pf: PF (h_in, w_in, cin); out_pf / resid_pf: PF (h_in/stride, w_in/stride, cout).
 * resid_pf may be NULL.  in_pf must not alias out_pf. */
int wsi_conv3x3_bn_act(const void* in_pf, void* out_pf, const void* resid_pf, const void* wpk, const float* bias,
                       int n, int h_in, int w_in, int cin, int cout, int stride, int relu, int planes,
                       void* stream);
int wsi_conv1x1_bn(const void* in_pf, void* out_pf, const void* wpk, const float* bias, int n, int h_in, int w_in,
                   int cin, int cout, int stride, int planes, void* stream);
/* The two stride-2 convs of a downsampling BasicBlock in one pass over the input
 * (resnets_shift.py:41 conv1 with stride 2 + ReLU, and :173-177 the 1x1 stride-2 downsample, no ReLU):
 * out_conv_pf = relu(bn1(conv3x3_s2(x))), out_ds_pf = bn_d(conv1x1_s2(x)).  cout % 128 == 0. */
/* The first conv of a U-Net decoder block with its input assembly fused in (r04): out = act(bn(conv3x3(cat(up2(up_pf), skip_pf)))),
 * i.e. segmentation_models_pytorch's DecoderBlock `x = F.interpolate(x, scale_factor=2, mode='nearest'); x = torch.cat([x, skip], 1);
 * x = conv(x)` (third-party, absent; called at /root/reference/utils/eval.py:51,199-200 through `model.decoder(...)`;
 * /root/reference/eval_tumorbed.py:21-28 builds it).  up_pf: PF tensor (n, h/2, w/2, c_up); skip_pf: PF tensor (n, h, w, c_skip) or
 * NULL with c_skip = 0; wpk: wsi_prepack_conv of the (cout, c_up + c_skip, 3, 3) weights in cat order; out_pf: (n, h, w, cout).
 * -EINVAL when the shape's kernel has no fused form (the caller then materialises the concatenated tensor). */
int wsi_conv3x3_up_concat_bn_act(const void* up_pf, const void* skip_pf, void* out_pf, const void* wpk, const float* bias, int n, int h, int w,
                                 int c_up, int c_skip, int cout, int relu, int planes, void* stream);
int wsi_conv3x3s2_ds_fused(const void* in_pf, void* out_conv_pf, void* out_ds_pf, const void* wpk3, const float* bias3,
                           const void* wpk1, const float* bias1, int n, int h_in, int w_in, int cin, int cout, int planes,
                           void* stream);
/* Phase-split tensors: the four phase images (y&1, x&1) of an (n,h,w,c) tensor, each a PF tensor of geometry
 * (n,h/2,w/2,c), concatenated (wsi_pf_split_bytes = 4 * wsi_pf_bytes(n,h/2,w/2,c,planes); zero-fill once like a PF
 * buffer).  A stride-1 conv can WRITE its output in this form (wsi_conv3x3_bn_act_split; h, w even; planes >= 2) and
 * the stride-2 block entry can READ it (wsi_conv3x3s2_ds_fused_split: the "wide" stride-2 kernel, 8 waves sharing
 * one weight stage per tap; output maps up to 33 wide, cout % 128 == 0, planes >= 2, otherwise -EINVAL).  Used by
 * wsi_trunk_forward between the residual stages; resnets_shift.py:41,173-177 as above. */
size_t wsi_pf_split_bytes(int n, int h, int w, int c, int planes);
int wsi_conv3x3_bn_act_split(const void* in_pf, void* out_split, const void* resid_pf, const void* wpk, const float* bias,
                             int n, int h, int w, int cin, int cout, int relu, int planes, void* stream);
int wsi_conv3x3s2_ds_fused_split(const void* in_split, void* out_conv_pf, void* out_ds_pf, const void* wpk3,
                                 const float* bias3, const void* wpk1, const float* bias1, int n, int h_in, int w_in,
                                 int cin, int cout, int planes, void* stream);
/* A-B hook: s2_slab = 1 (default) routes stride-2 3x3 convs to the phase-slab kernel (64-pixel tiles) and lets
 * the trunk fuse the downsample branch; 3 = the same with 128-pixel tiles; 0 = per-tap gather kernel + separate
 * 1x1 launch.  Flags added to the value: +8 XCD-aware workgroup order, +16 / +32 use the wide stride-1 kernel only
 * from 256 channels / never (default: from 128), +128 the trunk keeps ordinary PF between stages (no phase-split
 * hand-over to the wide stride-2 kernel), +256 / +512 XCD-contiguous pixel-tile ranges off / 64-channel layer only
 * (default: every stride-1 layer), +2048 the strided blocks' 1x1 downsample as its own tensor + residual instead of an extra K
 * segment of the block's second conv (mode 3), +4096 layer-1 kernel without paired-tile LDS addressing, +16384 the trunk keeps
 * 128-byte lines for the stem output and the layer-1 tensors (mode 3 default: 96-byte lines there - the hi6 plane is rebuilt
 * in LDS by the layer-1 kernel; bit-identical results), +32768 the wide stride-2 kernel with 128 instead of 256 output channels
 * per workgroup on the layer-3 / layer-4 entries (mode 3; bit-identical), +1024 the 64-channel layer 1 on the r03 slab3 kernel instead of
 * the row-stacked kernel (mode 3, 64-wide maps; results equal to a few ulps of the fp32 sums: another summation order), +65536 the
 * U-Net decoder blocks write the upsampled + concatenated tensor before their first conv instead of reading both sources in it
 * (bit-identical), +131072 the wide stride-1 kernel keeps the 9-pixel slab pitch on 8 x 8 maps (r05 default: 8-pixel slab rows, no LDS
 * bank conflicts; bit-identical), +1048576 the 64-channel layer 1 on the persistent producer-fed kernel (r05 study route: bit-identical,
 * measured 30-45 % slower than the row-stacked kernel), +2097152 the U-Net decoder's last block and head as three launches even when
 * the fused-tail weights are present (planes 2; results equal to fp32 rounding of the summed polyphase weights, not bit-identical).
 * Process-wide. */
int wsi_conv_set_mode(int s2_slab);
/* tuning hook: same as wsi_conv3x3_bn_act with an explicit tile configuration for the stride-1
 * kernel (cfg index into the table in csrc/conv.hip; -1 = tuned default; -22 if not applicable) */
int wsi_conv3x3_bn_act_cfg(const void* in_pf, void* out_pf, const void* resid_pf, const void* wpk, const float* bias,
                           int n, int h_in, int w_in, int cin, int cout, int stride, int relu, int planes, int cfg,
                           void* stream);

/* ---- heads -----------------------------------------------------------------------------------
 * avgpool_fc: AdaptiveAvgPool2d(1) + flatten (+ Linear(c -> k)) (resnets_shift.py:206-208,
 *   models/models.py:32-38).  feat_out [n][c] and logits_out [n][k] may each be NULL.
 * linear: y[b][j] = act(x[b] . w[j] + bias[j]) in fp32 (resnets_shift.py:135-139, models.py:46-50) */
int wsi_avgpool_fc(const void* in_pf, int n, int h, int w, int c, const float* fc_w, const float* fc_b, int k,
                   float* feat_out, float* logits_out, int planes, void* stream);
int wsi_linear(const float* x, const float* w, const float* bias, float* y, int b, int k, int j, int relu,
               void* stream);

/* ---- layout converters (API boundary + tests) ----------------------------------------------- */
int wsi_pf_pack(const float* in_nchw, void* out_pf, int n, int c, int h, int w, int planes, void* stream);
int wsi_pf_unpack(const void* in_pf, float* out_nchw, int n, int c, int h, int w, int planes, void* stream);

/* ---- whole trunk: stem + layer1..4 (+ avgpool + Linear) for n patches -------------------------
 * The per-batch compute of ResNet.forward (resnets_shift.py:194-212) and of
 * predict_tumorbed(mode='cls') (utils/eval.py:196-198).  All pointers device memory. */
typedef struct {
    const void* stem_w;   const float* stem_b;
    const void* stem_w_u8; const float* stem_b_u8;     /* wsi_prepack_stem_u8 (u8 slide input) or NULL */
    float norm[6];                                     /* mean[3], std[3] of the transform folded into stem_w_u8 */
    const void* conv_w[16]; const float* conv_b[16];   /* layerL.B.convK at index (L-1)*4 + B*2 + (K-1) */
    const void* down_w[3];  const float* down_b[3];    /* layer2..4 .0.downsample */
    const float* head_w;  const float* head_b;  int head_k;  /* Linear(512 -> head_k) or NULL */
    int planes;
} wsi_trunk_weights;

size_t wsi_trunk_workspace_bytes(int n, int h, int w, int planes);
/* zero-fills the workspace for the (n,h,w,planes) plan; call once before the first forward */
int wsi_trunk_workspace_init(void* workspace, int n, int h, int w, int planes, void* stream);
/* call before freeing a workspace: the library forgets what it remembered about that address (the line layout its stage-0
 * buffers last held), so a later allocation at the same address starts clean.  The reference has no counterpart: its
 * activations are torch tensors freed by the allocator (/root/reference/utils/eval.py:190-215 allocates per batch). */
int wsi_trunk_workspace_release(void* workspace);
/* workspace_n: the image count the workspace was sized and initialised for (>= n; 0 means n): one workspace planned
 * for the largest batch serves every smaller one (ragged last batches, variable bag counts) without re-initialisation. */
int wsi_trunk_forward(const wsi_trunk_weights* wt, const float* in_f32, const uint8_t* slide,
                      long long slide_pitch_bytes, int slide_h, int slide_w, const int* tile_xy, const float* lut,
                      int n, int h, int w, void* workspace, int workspace_n, float* feat_out /* [n][512] or NULL */,
                      float* logits_out /* [n][head_k] or NULL */, float* fmap_out /* f32 NCHW [n][512][h/32][w/32] or NULL */,
                      void* stream);
/* Sub-batching of the early (large-map) stages so their tensors stay in the 256 MiB Infinity Cache:
 * stem+maxpool run `stem_chunk` images at a time, layer1 `layer1_chunk` (a multiple of stem_chunk);
 * 0 = whole batch (default; measured on MI355X, r03: every chunk size from 32 to 768 is slower than the whole batch at
 * 6 162 tiles - DESIGN.md section 4).  Results are bit-identical to the unchunked run for every setting.  Process-wide. */
int wsi_trunk_set_chunks(int stem_chunk, int layer1_chunk);
/* debug / parity taps: run the trunk up to stage `stop_after` (0 = stem+maxpool output, 1..8 =
 * layer1.0, layer1.1, ..., layer4.1) and unpack that tensor to f32 NCHW. */
int wsi_trunk_forward_tap(const wsi_trunk_weights* wt, const float* in_f32, const uint8_t* slide,
                          long long slide_pitch_bytes, int slide_h, int slide_w, const int* tile_xy, const float* lut,
                          int n, int h, int w, void* workspace, int workspace_n, int stop_after, float* tap_out_nchw,
                          void* stream);

/* ---- measurement hook -------------------------------------------------------------------------
 * wsi_prof_begin arms HIP-event timing (on the launch stream) of every conv / stem launch made by
 * wsi_trunk_forward; wsi_prof_end disarms, waits for the events and returns the number of records
 * copied: ms, kind (1 = 3x3 stride 1 of layers 2-4, 5 = 3x3 stride 1 of layer 1, 2 = 3x3 stride 2, 3 = 1x1 downsample,
 * 4 = stem+maxpool)
 * and algorithmic FLOPs (2*M*N*K over real output pixels) per launch. */
int wsi_prof_begin(int max_records);
int wsi_prof_end(float* ms_out, int* kind_out, double* flops_out, int cap);

/* ---- slide-side ops ---------------------------------------------------------------------------
 * tile_gather: utils/dataset.py:171-185 (+ transform): normalised f32 NCHW [n][3][ph][pw].
 * stitch_add:  utils/eval.py:213-215: pred[c][ty+.. , tx+..] += logits[t][c] over dy x dx (f64).
 * softmax_threshold_argmax: utils/preprocessing.py:156-172 (+ heat map utils/eval.py:219-228):
 *   heat_mode 0 = probs[1] ('cls'), 1 = probs[2]+probs[3] ('seg'); probs/classes/heat may be NULL */
int wsi_tile_gather(const uint8_t* slide, long long slide_pitch_bytes, int slide_h, int slide_w, const int* tile_xy,
                    const float* lut, float* out_nchw, int n, int ph, int pw, void* stream);
int wsi_stitch_add(const float* tile_logits, const int* map_xy, int t, int c, int dy, int dx, double* pred, int map_h,
                   int map_w, void* stream);
/* dense form (utils/eval.py:58-60, predict_wsis): tile_pred [t][c][ph][pw] fp32 added at map_xy */
int wsi_stitch_add_dense(const float* tile_pred, const int* map_xy, int t, int c, int ph, int pw, double* pred, int map_h,
                         int map_w, void* stream);
int wsi_softmax_threshold_argmax(const double* pred, int c, long long hw, const double* class_thresh, double* probs,
                                 uint8_t* classes, const uint8_t* mask, int heat_mode, uint8_t* heat, void* stream);
/* ---- slide ingestion in front of the tile producer (utils/dataset.py:171-185: read_region(...).convert('RGB') [+ image.resize]) ----
 *   wsi_ring_*        `slots` pinned-host buffers of slot_bytes (multiple of 4) + device staging + a copy stream of the ring's own.
 *                     Producer protocol per band: wsi_ring_wait_slot (blocks until that slot's previous copy has landed) ->
 *                     decode into wsi_ring_host_slot (any thread) -> wsi_ring_submit (one thread at a time): async H2D + unpack of
 *                     `rows` x `width` pixels of `channels` (3, or 4 with the alpha byte dropped like PIL convert('RGB')) bytes,
 *                     row pitch src_pitch, into level_rows (device pointer to the first destination row, pitch level_pitch).
 *                     wsi_ring_acquire (before the first submit into memory allocated on compute_stream) makes the copy stream wait
 *                     for everything enqueued on compute_stream so far - the destination may be a recycled block that queued
 *                     kernels still read; wsi_ring_fence makes compute_stream wait for everything submitted so far (neither blocks
 *                     the host); wsi_ring_drain blocks the host.  Nothing else touches the compute stream: decode, copies and the
 *                     trunk overlap.  A ring belongs to the device current at its creation (wsi_ring_device).
 *   wsi_resample_*    the scan_resize != 1 branch (utils/dataset.py:180-181 `image.resize((tile_w, tile_h))`, Pillow's default
 *                     BICUBIC): n tiles of (in_h, in_w) read from the u8 slide at tile_xy (out-of-slide pixels 0) -> out
 *                     (n, out_h, out_w, 3) u8, bit-exact with Pillow (22-bit fixed-point taps, horizontal then vertical pass, u8
 *                     between); scratch: wsi_resample_scratch_bytes(plan, n).  A plan holds the tap tables on the current device
 *                     (creation synchronises; reuse it). */
typedef struct wsi_ring wsi_ring;
typedef struct wsi_resample_plan wsi_resample_plan;
int wsi_ring_create(wsi_ring** out, int slots, size_t slot_bytes);
void* wsi_ring_host_slot(wsi_ring* ring, int slot);
int wsi_ring_wait_slot(wsi_ring* ring, int slot);
int wsi_ring_submit(wsi_ring* ring, int slot, int rows, int width, int channels, long long src_pitch, uint8_t* level_rows, long long level_pitch);
int wsi_ring_acquire(wsi_ring* ring, void* compute_stream);
int wsi_ring_fence(wsi_ring* ring, void* compute_stream);
int wsi_ring_device(const wsi_ring* ring);
int wsi_ring_drain(wsi_ring* ring);
void wsi_ring_destroy(wsi_ring* ring);
int wsi_resample_plan_create(wsi_resample_plan** out, int in_h, int in_w, int out_h, int out_w);
void wsi_resample_plan_destroy(wsi_resample_plan* plan);
size_t wsi_resample_scratch_bytes(const wsi_resample_plan* plan, int n);
int wsi_resample_tiles(const wsi_resample_plan* plan, const uint8_t* slide, long long pitch, int sh, int sw, const int* tile_xy, int n,
                       uint8_t* out, void* scratch, void* stream);
/* ---- region-proposal generation (the step in front of the bag path; bit-exact against oracle/proposals_oracle.py) ----
 *   wsi_find_nuclei_hsv        utils/preprocessing.py:94-98 (mode 'hsv'): skimage rgb2hsv saturation > mu_percent in float64 on
 *                              packed u8 pixels (pixel_stride bytes apart, R G B first) -> 0/1 mask
 *   wsi_connected_components   scannet.py:55 cv2.connectedComponentsWithStats((mask > 0)): 8-connected, int32 labels, 0 =
 *                              background, 1.. in raster order of first pixels; count_out (device int) = number of components;
 *                              scratch: wsi_connected_components_scratch_bytes.  Synchronises the stream (convergence flag).
 *   wsi_kmeans_points          utils/regiontools.py:89 key points: Lloyd iterations from the caller's initial centres on integer
 *                              (x, y) points, float64 distances, ties to the lower index, exact integer sums, empty clusters keep
 *                              their centre, stops when no label changes (the reference's sklearn MiniBatchKMeans is RNG / version
 *                              dependent: own deterministic spec); scratch: (3 k + 1) * 8 bytes.  Synchronises the stream.  On
 *                              return the scratch holds the last assignment's exact per-cluster sums: {sum x, sum y, count} as int64
 *                              triples - what a caller needs to compare two partitions exactly.
 *   wsi_kmeans_seed_farthest   r04: farthest-point initial centres for wsi_kmeans_points (the point farthest from the floor-mean,
 *                              then k - 1 times the point farthest from its nearest seed; exact integers, ties to the lower index);
 *                              scratch: n * 8 bytes.  The key points are the better of two Lloyd runs (raster-stratified and
 *                              farthest-point seeds; smaller within-cluster sum of squares, compared exactly). */
int wsi_find_nuclei_hsv(const uint8_t* rgb, long long npix, int pixel_stride, double mu_percent, uint8_t* mask_out, void* stream);
size_t wsi_connected_components_scratch_bytes(int h, int w);
int wsi_connected_components(const uint8_t* mask, int h, int w, int* labels_out, int* count_out, void* scratch, void* stream);
int wsi_kmeans_points(const int* points_xy, int n, double* centres_xy, int k, int max_iters, int* labels_out, void* scratch, void* stream);
int wsi_kmeans_seed_farthest(const int* points_xy, int n, int k, double* centres_xy_out, void* scratch, void* stream);
/* utils/preprocessing.py:88-92 find_nuclei(mode='lab'): mask = a > (1 + mu_percent) * mean(a), a = the second channel of skimage's
 * rgb2lab (own deterministic spec: a in 2^-20 fixed point, exact mean; parity unpinned).  scratch: 16 + 4 * npix bytes.
 * utils/preprocessing.py:101-106 fill_mask: wsi_fill_holes = scipy.ndimage.binary_fill_holes (background components, 4-connected,
 * that touch no border are filled; synchronises the stream like wsi_connected_components); the 10x10 close that follows is two
 * wsi_morph_rect calls (dilate, erode). */
int wsi_find_nuclei_lab(const uint8_t* rgb, long long npix, int pixel_stride, double mu_percent, uint8_t* mask_out, void* scratch, void* stream);
size_t wsi_fill_holes_scratch_bytes(int h, int w);
int wsi_fill_holes(const uint8_t* mask, int h, int w, uint8_t* out, void* scratch, void* stream);
/* slic.py:43 skimage.segmentation.slic(img_as_float(rgb), n_segments, compactness, sigma, enforce_connectivity=False) on a 2-D RGB
 * thumbnail, as an own deterministic specification of the published algorithm (skimage is absent: parity unpinned;
 * oracle/proposals_oracle.py slic_labels): Gaussian filter with the caller's 2 radius + 1 float64 weights (scipy.ndimage order,
 * reflect; radius 0 = none) -> rgb2lab / compactness in 2^-20 fixed point -> `iters` rounds of windowed nearest-centre assignment
 * (ties to the lower centre) + exact-integer mean update.  segments: k x {cy, cx, c0, c1, c2, alive} float64, in: skimage's
 * regular grid (colour 0, alive 1), out: the final centres; step_y / step_x: the grid steps, step = their maximum.
 * labels_out (h, w) int32 in [0, k).  scratch: wsi_slic_scratch_bytes(h, w, k); k <= 2048. */
size_t wsi_slic_scratch_bytes(int h, int w, int k);
int wsi_slic(const uint8_t* rgb, int h, int w, const double* gauss_weights, int radius, double* segments, int k, int step_y, int step_x,
             double step, double compactness, int iters, int* labels_out, void* scratch, void* stream);
/* tile list of the sliding-window path on the device (utils/dataset.py:143-166): candidates in the reference's order (interior
 * raster from (1,1) with strides (sw, sh), then the right-edge column x = iw-1-pw, then the bottom-edge row y = ih-1-ph), kept iff
 * the mask window mask[int(y*m) : +int(ph*m), int(x*m) : +int(pw*m)] (clipped like a numpy slice; mask == NULL keeps all) has a
 * nonzero fraction >= thresh.  tile_xy_out: room for wsi_tile_grid_candidates pairs, filled compacted in order; count_out: device int;
 * scratch: wsi_tile_grid_scratch_bytes(candidates). */
long long wsi_tile_grid_candidates(int iw, int ih, int ph, int pw, int sh, int sw);
size_t wsi_tile_grid_scratch_bytes(long long candidates);
int wsi_tile_grid(int iw, int ih, int ph, int pw, int sh, int sw, const uint8_t* mask, int mask_h, int mask_w, double m, double thresh,
                  int* tile_xy_out, int* count_out, void* scratch, void* stream);
/* guard of the float64 stitch (wsi_stitch_add / _dense use float64 atomics): out2 (device, 2 ints) = {smallest, largest} biased
 * exponent of the nonzero finite values.  While (largest - smallest) + log2(addends per map pixel) <= 29 every float64 sum of
 * these fp32 values is exact, so the accumulate is order-independent and bit-stable (like the reference's, whose DataLoader
 * shuffles: utils/dataset.py:192) */
int wsi_exponent_span(const float* values, long long n, int* out2, void* stream);
/* region paint (scannet.py:154-155, slic.py:98-99): label[pixel_idx[e]] = region_class[region_of[e]] for every entry e, regions
 * numbered in paint order - where regions overlap the last one wins, as in the reference's loop; label is int64 (np.zeros of
 * the reference), untouched elsewhere; winner_scratch: npix ints */
int wsi_paint_regions(const long long* pixel_idx, const int* region_of, long long n, const uint8_t* region_class, int* winner_scratch,
                      long long* label, long long npix, void* stream);

/* ---- tumour-bed post-process of the stitched map (all device memory; byte / integer / float64 work) ----------
 * Bit-exact against oracle/postprocess_oracle.py, which restates the published algorithms of the third-party calls
 * the reference makes here (OpenCV, scikit-image, mahotas: absent and un-pinned, so parity is unpinned by the
 * reference itself).
 *   wsi_resize_bilinear_f64  utils/eval.py:66-71   cv2.resize(pred[c], level_dimensions[2]) (INTER_LINEAR, float64)
 *   wsi_argmax_classes       utils/eval.py:82      np.argmax(pred, 0) -> u8
 *   wsi_morph_rect           utils/eval.py:91,95   cv2.erode (op 0) / cv2.dilate (op 1) / MORPH_OPEN (op 2), k x k ones,
 *                                                  default anchor and border; tmp: h*w bytes, no aliasing
 *   wsi_convex_hull_image    utils/eval.py:92      skimage convex_hull_image (offset_coordinates=True), exact predicate
 *   wsi_bwperim              utils/eval.py:94      mahotas.bwperim(n=4)
 *   wsi_tumor_bed            utils/eval.py:90-96 and paper_tools/overlay_tb_wsi.py:46-64 in one call:
 *       (codes >= min_code) -> open k x k -> hull image (tb_pred_out) -> perimeter -> dilate (outline_out);
 *       codes = class map with min_code 2, or u8 heat map with min_code ceil(0.9 * 255) = 230; opened_out may be NULL
 *   wsi_hull_polygon         the hull of the last wsi_tumor_bed / wsi_convex_hull_image on this workspace as a closed
 *                            (x, y) float64 contour (the ordered input of wsi_esp); count_out: device int
 *   wsi_mask_iou_counts      utils/eval.py:104     out2 = {sum(a & b), sum(a | b)} (a, b compared != 0)
 *   wsi_score_counts         utils/eval.py:107-121 out6 = {#(gt>0), #(p==gt & gt>0), sum|p-gt|, the reference's weight sum,
 *                                                  #(p>0 & gt>0), #(p>0 | gt>0)}; p is multiplied by mask when given
 *   wsi_esp                  contour_ordering.py:33-60  evenly_spaced_points_on_a_contour; scratch: n doubles */
int wsi_resize_bilinear_f64(const double* src, int c, int hs, int ws, double* dst, int hd, int wd, void* stream);
int wsi_argmax_classes(const double* pred, int c, long long hw, uint8_t* classes, void* stream);
int wsi_morph_rect(const uint8_t* src, uint8_t* dst, uint8_t* tmp, int h, int w, int k, int op, void* stream);
int wsi_bwperim(const uint8_t* src, uint8_t* dst, int h, int w, void* stream);
size_t wsi_tumor_bed_workspace_bytes(int h, int w);
int wsi_convex_hull_image(const uint8_t* src, uint8_t* dst, int h, int w, void* workspace, void* stream);
int wsi_tumor_bed(const uint8_t* codes, int h, int w, int min_code, int open_k, int dilate_k, uint8_t* opened_out,
                  uint8_t* tb_pred_out, uint8_t* outline_out, void* workspace, void* stream);
int wsi_hull_polygon(void* workspace, int h, int w, double* out_xy, int cap, int* count_out, void* stream);
int wsi_mask_iou_counts(const uint8_t* a, const uint8_t* b, long long n, unsigned long long* out2, void* stream);
int wsi_score_counts(const uint8_t* p, const uint8_t* gt, const uint8_t* mask, long long n, unsigned long long* out6,
                     void* stream);
int wsi_esp(const double* pts_xy, int n, int num_pts, double* out_xy, double* scratch, void* stream);

/* ---- U-Net decoder: the dense 'seg' path (utils/eval.py:51 `model(batch_image)`, :196-200 `model.decoder(model.encoder(x))`) ----
 * The reference drives segmentation_models_pytorch's Unet('resnet18') here (eval_tumorbed.py:21-28): third-party, absent and
 * un-pinned, so the architecture is restated from the published 0.0.x source (parity unpinned):
 * encoder maps [x4 512 ch /32, x3 256 /16, x2 128 /8, x1 64 /4, x0 64 /2 (conv1+bn1+relu before the max pool)]; five decoder
 * blocks L = 1..5: nearest x2 upsample, concat the next skip, 2 x (conv3x3 + BN + ReLU) to 256/128/64/32/16 channels;
 * final_conv 1x1 to `classes`.  Decoder convs are prepacked with wsi_prepack_conv (k = 3) on channel counts padded to
 * multiples of 64 (zero weights in the padding): conv index 2(L-1)+J, cin = {768,256, 384,128, 192,64, 128,64, 64,64},
 * cout = {256,256, 128,128, 64,64, 64,64, 64,64}; head_w [classes][head_cin] fp32 with head_cin = 16.
 * wsi_unet_forward: stem + trunk + decoder for n patches (inputs as wsi_trunk_forward); logits_out [n][classes][h][w] fp32 and /
 *   or enc_out[5] = the five encoder maps as fp32 NCHW (either may be NULL).  Workspace: wsi_unet_workspace_bytes, zero-filled
 *   once by wsi_unet_workspace_init; workspace_n as in wsi_trunk_forward.
 * wsi_unet_decoder: the decoder alone on caller-held fp32 NCHW encoder maps (same workspace). */
typedef struct {
    const void* conv_w[10]; const float* conv_b[10];
    int cin[10], cout[10];
    const float* head_w; const float* head_b;
    int head_cin, classes;
    const void* tail_w;          /* device copy of wsi_unet_tail_prepack's blob, or NULL: the fused last block + head (planes 2) */
} wsi_unet_decoder_weights;
/* r05, planes 2: the LAST decoder block (upsample, two 3x3 conv + BN + ReLU at full resolution) and the 1x1 head as ONE kernel
 * (csrc/tail.hip: the conv on the upsampled map as a polyphase filter on the low-resolution rows, both intermediate tensors kept in
 * LDS).  w1 [cmid][cin][3][3], w2 [cmid][cmid][3][3] fp32 with their BatchNorm vectors, head_w [classes][cmid], head_b [classes];
 * cin = 32, cmid <= 16, classes <= 4.  `out` = wsi_unet_tail_prepack_bytes() bytes of host memory; copy it to the device and store the
 * pointer in wsi_unet_decoder_weights.tail_w.  The three-launch path stays in force when tail_w is NULL, in the other precision
 * modes and for maps wider than 256. */
size_t wsi_unet_tail_prepack_bytes(void);
int wsi_unet_tail_prepack(const float* w1, const float* bn1_weight, const float* bn1_bias, const float* bn1_mean, const float* bn1_var,
                          const float* w2, const float* bn2_weight, const float* bn2_bias, const float* bn2_mean, const float* bn2_var,
                          float eps, const float* head_w, const float* head_b, int cin, int cmid, int classes, void* out);
size_t wsi_unet_workspace_bytes(const wsi_unet_decoder_weights* dw, int n, int h, int w, int planes);
int wsi_unet_workspace_init(const wsi_unet_decoder_weights* dw, void* workspace, int n, int h, int w, int planes, void* stream);
int wsi_unet_forward(const wsi_trunk_weights* wt, const wsi_unet_decoder_weights* dw, const float* in_f32, const uint8_t* slide,
                     long long slide_pitch_bytes, int slide_h, int slide_w, const int* tile_xy, const float* lut, int n, int h,
                     int w, void* workspace, int workspace_n, float* logits_out, float* enc_out[5], void* stream);
int wsi_unet_decoder(const wsi_unet_decoder_weights* dw, const float* const enc_nchw[5], int n, int h, int w, int planes,
                     void* workspace, int workspace_n, float* logits_out, void* stream);
/* F.interpolate(pred_src, (tile_h * r, tile_w * r)) of utils/eval.py:202-206 (default mode 'nearest') on planes_n
 * contiguous fp32 (hs, ws) planes */
int wsi_resize_nearest_f32(const float* src, long long planes_n, int hs, int ws, float* dst, int hd, int wd, void* stream);

#ifdef __cplusplus
}
#endif
#endif
