/* yolo_mi355x.h — C ABI of libyolo_mi355x.so (MI355X / gfx950 YOLOv3 hot path).
 *
 * The reference (GabeTsai/YOLO-For-Turbines) has no FFI/plugin layer: its boundary is the
 * Python call surface of code/model.py and code/utils.py. Every entry point
 * below states which reference routine's arithmetic it replaces. The host-side mirror of the
 * reference interface (same class / function names) lives in yolo_for_turbines_amd/ and calls
 * these through ctypes; INTEGRATION.md shows the binding.
 *
 * Conventions
 *  - all data pointers are DEVICE pointers (tensor.data_ptr()); the caller owns every buffer,
 *    including workspaces; the library allocates nothing and keeps no mutable global state
 *    except a thread-local error string;
 *  - `stream` is a hipStream_t passed as void*; every function only enqueues work on it and
 *    never synchronises the device;
 *  - return value 0 = ok, negative = error (message: yolo_last_error()).
 *  - activations are NHWC ("pixel-major"): element (n,h,w,c) of a tensor with channel stride
 *    `ld` and channel offset `off` lives at ((n*H + h)*W + w)*ld + off + c. A concat buffer is
 *    simply one allocation with a larger `ld` that several producers write slices of.
 */
#ifndef YOLO_MI355X_H
#define YOLO_MI355X_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define YOLO_OK 0
#define YOLO_ERR_ARG (-1)
#define YOLO_ERR_UNSUPPORTED (-2)
#define YOLO_ERR_LAUNCH (-3)
#define YOLO_ERR_WORKSPACE (-4)

enum { YOLO_F32 = 0, YOLO_F16 = 1, YOLO_BF16 = 2 };
enum { YOLO_ACT_NONE = 0, YOLO_ACT_LEAKY = 1, YOLO_ACT_MISH = 2 };
/* where the epilogue writes: plain NHWC; NHWC with nearest 2x upsample (each value goes to its
 * 2x2 destination pixels, replaces nn.Upsample + the first half of torch.cat, model.py:189-191,
 * 222); or the detection-head layout (B,3,g,g,5+nc) contiguous (replaces the reshape + permute
 * of ScalePredictionBlock.forward, model.py:145-148; conv channel = a*(5+nc)+k). */
enum { YOLO_OUT_NHWC = 0, YOLO_OUT_UPSAMPLE2X = 1, YOLO_OUT_HEAD = 2 };
enum { YOLO_FLAG_RESIDUAL = 1, YOLO_FLAG_NANCHECK = 2 };

/* One fused block: y = [residual +] act(scale[c] * conv(x, w)[c] + shift[c]).
 * Replaces CNNBlock.forward (model.py:80-86: Conv2d -> BatchNorm2d(eval) -> LeakyReLU/Mish, or
 * bare Conv2d + bias) and the `x + layer(x)` of ResidualBlock.forward (model.py:115-121).
 * BN is folded by yolo_bn_fold(); a bare conv passes scale = 1, shift = bias. */
typedef struct yolo_conv_desc {
    int32_t n, h, w;        /* input batch, height, width                                  */
    int32_t cin, cout;      /* logical channel counts                                      */
    int32_t ksize, stride;  /* 1 or 3 (padding = ksize/2, model.py:201); 1 or 2            */
    int32_t x_ld, x_off;    /* input channel stride / offset (elements)                    */
    int32_t y_ld, y_off;    /* output  "   (ignored for YOLO_OUT_HEAD)                     */
    int32_t r_ld, r_off;    /* residual "  (YOLO_FLAG_RESIDUAL), same spatial size as y    */
    int32_t act;            /* YOLO_ACT_*                                                  */
    int32_t out_mode;       /* YOLO_OUT_*                                                  */
    int32_t dtype;          /* YOLO_F32, YOLO_F16 or YOLO_BF16: activations (x, residual, y) and   */
                            /* packed weights; accumulation, scale/shift and heads stay fp32       */
    int32_t flags;          /* YOLO_FLAG_*                                                 */
    int32_t tile;           /* 0 = library heuristic; else forced tile id (tuning/tests)   */
} yolo_conv_desc;

/* One queued launch for yolo_conv_fwd_batch (pointers as 64-bit integers so the table can be
 * built once per input shape on the host and replayed every forward). */
typedef struct yolo_conv_op {
    yolo_conv_desc d;
    uint64_t x, w_packed, scale, shift, residual, y;
    uint64_t workspace, workspace_bytes;   /* yolo_conv_workspace_bytes(&d); 0 / 0 = none (direct kernels only) */
} yolo_conv_op;

const char* yolo_last_error(void);
int yolo_version(void);

/* ---- weights (replaces nothing arithmetic: layout change of nn.Conv2d.weight, OIHW fp32,
 *      as filled by the Darknet loader model.py:293-305) ---------------------------------- */
/* elements of the packed buffer. It holds (1) the row-major matrix [cout_pad128][K_pad32],
 * K index = (kh*k + kw)*cin_pad + ci (channels innermost, matching NHWC gathers), used by the
 * register-staged kernel, and, when cin % 32 == 0, (2) a copy in MFMA-fragment order
 * [cout_pad128/32][cin/32][k*k][4][64 lanes][4] streamed straight into registers by the
 * stride-1 patch kernel, and, when ksize == 3 and cin % 4 == 0, (3) the Winograd-domain filters
 * G g G^T as [16][cin/4][cout_pad64][4] (yolo_conv_fwd_ws). */
size_t yolo_packed_weight_elems(int cout, int cin, int ksize);
/* bytes of the packed buffer for a dtype. YOLO_F16 / YOLO_BF16: fragment order for
 * v_mfma_f32_32x32x16_{f16,bf16}: [cout_pad128/32][cin/32][k*k][2][64 lanes][8 halfs] (cin % 32 == 0). */
size_t yolo_packed_weight_bytes(int cout, int cin, int ksize, int dtype);
int yolo_pack_weights(const float* w_oihw, void* w_packed, int cout, int cin, int ksize, int dtype, void* stream);
/* Many layers at once (every weight tensor changes at `optimizer.step()`, train.py:68): items is a HOST array. 16-bit dtypes:
 * one launch per 48 items. dgrad = 0: the layout of yolo_pack_weights; dgrad = 1: of yolo_pack_weights_dgrad(flip = 1)
 * (stride-1 layers only). fp32: the per-item functions are called in turn. */
typedef struct yolo_pack_item { const float* w_oihw; void* w_packed; int cout, cin, ksize, reserved; } yolo_pack_item;
int yolo_pack_weights_batch(const yolo_pack_item* items, int n, int dgrad, int dtype, void* stream);
/* inverse (for gradients / checkpoint export): packed -> OIHW */
int yolo_unpack_weights(const void* w_packed, float* w_oihw, int cout, int cin, int ksize, int dtype, void* stream);
/* scale = gamma / sqrt(var + eps), shift = beta - mean * scale (nn.BatchNorm2d eval,
 * model.py:61,84). gamma == NULL: scale = 1, shift = beta (bias of a bare conv, model.py:86). */
int yolo_bn_fold(const float* gamma, const float* beta, const float* mean, const float* var, float eps,
                 float* scale, float* shift, int c, void* stream);

/* ---- layout boundary ------------------------------------------------------------------- */
/* (N,C,H,W) fp32 -> NHWC with c_pad >= C channels (extra channels zero). Sets *nan_flag |= 1 if
 * any input element is NaN (the `assert torch.sum(torch.isnan(x)) == 0` of model.py:175). */
int yolo_nchw_to_nhwc(const float* x, void* y, int n, int c, int h, int w, int c_pad, int dtype,
                      int32_t* nan_flag, void* stream);
int yolo_nhwc_to_nchw(const void* x, float* y, int n, int c, int h, int w, int x_ld, int x_off, int dtype, void* stream);

/* ---- stem: layers[0] (3 -> 32, 3x3, stride 1) straight from the NCHW input ---------------- */
/* Replaces CNNBlock.forward for the first block (model.py:20-21,80-86) together with the
 * NCHW -> NHWC conversion and the NaN-input guard (model.py:175): x_nchw (N,3,H,W) fp32 ->
 * y NHWC in `dtype` (fp32, or fp16/bf16 for the 16-bit path; ld/off like yolo_conv_desc). Direct VALU
 * convolution in fp32: the layer is bound by its output
 * bytes, not by FLOPs. Weights as [27][cout] (yolo_stem_pack from OIHW). *nan_flag |= 1 for a NaN
 * input element, |= 2 for a NaN output. */
int yolo_stem_supported(int cin, int cout, int ksize, int stride);
int yolo_stem_pack(const float* w_oihw, float* w_k_major, int cout, void* stream);
int yolo_stem_fwd(const float* x_nchw, const float* w_k_major, const float* scale, const float* shift, void* y,
                  int n, int h, int w, int cout, int y_ld, int y_off, int act, int dtype, int32_t* nan_flag, void* stream);

/* ---- convolution blocks ---------------------------------------------------------------- */
/* nan_flag: *nan_flag |= 2 when YOLO_FLAG_NANCHECK is set and an output element is NaN
 * (the per-layer `raise ValueError("Nan in layer")` of model.py:183-184, checked once by host). */
int yolo_conv_fwd(const yolo_conv_desc* d, const void* x, const void* w_packed, const float* scale,
                  const float* shift, const void* residual, void* y, int32_t* nan_flag, void* stream);
int yolo_conv_fwd_batch(const yolo_conv_op* ops, int n_ops, int32_t* nan_flag, void* stream);
/* The same block with a caller-owned workspace. The fp32 3x3 stride-1 blocks with >= 64 input channels (the second convolution
 * of the residual units of model.py:115-121 at 104x104 ... 13x13, the 3x3 layers of the neck and of ScalePredictionBlock
 * model.py:140-143) then run as Winograd F(2x2, 3x3) - the algorithm PyTorch's backend itself picks for them: an input
 * transform pass into the workspace (16 planes of the 4x4 tiles, [xi][cin/4][tile][4]), 16 matrix products on the
 * transformed filters (third section of the packed fp32 buffer: [xi][cin/4][cout_pad64][4]) and the output transform in the
 * epilogue; 1 / 2.25 of the direct convolution's multiplications, same result within a few ulp of the transforms'
 * additions. yolo_conv_workspace_bytes: what descriptor d needs (0 = the launch takes no workspace); a NULL or smaller
 * workspace makes tile 0 fall back to the direct kernels of yolo_conv_fwd, tile 13 (= Winograd, forced) fail.
 * Streams that run concurrently need a workspace each. */
size_t yolo_conv_workspace_bytes(const yolo_conv_desc* d);
int yolo_conv_fwd_ws(const yolo_conv_desc* d, const void* x, const void* w_packed, const float* scale, const float* shift,
                     const void* residual, void* y, void* workspace, size_t workspace_bytes, int32_t* nan_flag, void* stream);
/* tile id the heuristic would pick (exposed for tests / tuning) and number of tile ids */
int yolo_conv_pick_tile(const yolo_conv_desc* d);
int yolo_conv_num_tiles(void);

/* ---- training: batch-statistics BatchNorm, activation, and the conv gradients ------------- */
/* All of these replace pieces of `grad_scaler.scale(loss).backward()` / the train-mode forward of
 * train.py:53-67 for the blocks of model.py:47-121 (nn.Conv2d, nn.BatchNorm2d(train), LeakyReLU /
 * Mish, skip add, nn.Upsample). Reductions are deterministic (fixed order, fp64 across threads). */
size_t yolo_bn_workspace_bytes(int m, int c);
/* Activation / gradient tensors (z, y, residual, dy, dz, x, dx, dup) are NHWC in `dtype` (YOLO_F32, or
 * YOLO_BF16 / YOLO_F16 for the autocast training path of train.py:53: 16-bit storage, fp32 arithmetic,
 * one rounding at the store); statistics, parameters and parameter gradients are always fp32.
 *
 * z: raw conv output, m = N*H*W pixels, c channels (ld/off as usual). Batch mean / biased variance ->
 * mean, invstd = 1/sqrt(var+eps), scale = gamma*invstd, shift = beta (the train kernels evaluate
 * (z - mean)*scale + shift in that order, like PyTorch); running stats are updated in place like
 * nn.BatchNorm2d (momentum, unbiased variance) unless the pointers are NULL. */
int yolo_bn_stats(const void* z, int m, int c, int ld, int off, const float* gamma, const float* beta, float momentum,
                  float eps, float* running_mean, float* running_var, float* mean, float* invstd, float* scale,
                  float* shift, int dtype, void* workspace, size_t workspace_bytes, void* stream);
/* The statistics pass fused into the producing convolution (16-bit kernels on LDS-DMA: 3x3 stride 1 with > 64 output channels,
 * 1x1 with >= 128 input and output channels): yolo_conv_fwd_stats = the raw convolution (identity epilogue: act NONE, no
 * residual, YOLO_OUT_NHWC) that also writes per-wave partial sums of z and z^2 (of the ROUNDED values it stores) as
 * stats[row][2][ld] fp32; yolo_bn_stats_from_partials = the second half of yolo_bn_stats on those rows (same outputs, fp64
 * across rows in a fixed order). yolo_conv_stats_rows: rows (and *ld) for a descriptor, 0 = no fused kernel for it. */
int yolo_conv_stats_rows(const yolo_conv_desc* d, int* ld);
int yolo_conv_fwd_stats(const yolo_conv_desc* d, const void* x, const void* w_packed, void* z, float* stats, size_t stats_bytes,
                        void* stream);
int yolo_bn_stats_from_partials(const float* partial, int rows, int ld, int m, int c, const float* gamma, const float* beta,
                                float momentum, float eps, float* running_mean, float* running_var, float* mean, float* invstd,
                                float* scale, float* shift, void* stream);
/* The BACKWARD reduction pass fused the same way. The gradient dy of a block's output is written last by the input-gradient
 * convolution of the first layer that consumes it (a stride-1 convolution with the flipped weights, identity epilogue,
 * optionally accumulating onto the running gradient); yolo_conv_dgrad_bstats is that launch (d = the descriptor yolo_conv_fwd
 * would get, act NONE, YOLO_OUT_NHWC, [YOLO_FLAG_RESIDUAL]) and ALSO writes per-wave partial sums of du and du * (z - mean),
 * du = dy * act'((z - mean) * scale + shift), over the ROUNDED dy it stores, for the block that produced the gradient's
 * tensor (its raw conv output z, its batch mean / scale / shift, act = LeakyReLU or Mish). stats[row][2][ld] fp32 as above,
 * followed by 3 * c floats of scratch. yolo_bn_act_bwd_rows = yolo_bn_act_bwd without its reduction pass, from those rows:
 * same dgamma / dbeta / dz. yolo_conv_bstats_rows: rows (and *ld) for a descriptor, 0 = no fused kernel for it. */
int yolo_conv_bstats_rows(const yolo_conv_desc* d, int* ld);
int yolo_conv_dgrad_bstats(const yolo_conv_desc* d, const void* dz, const void* w_packed, const void* residual, void* dx, const void* z,
                           int z_ld, int z_off, const float* mean, const float* scale, const float* shift, int act, float* stats,
                           size_t stats_bytes, void* stream);
int yolo_bn_act_bwd_rows(const void* dy, int dy_ld, int dy_off, const void* z, int z_ld, int z_off, const float* gamma,
                         const float* mean, const float* invstd, const float* scale, const float* shift, int m, int c, int act,
                         float* dgamma, float* dbeta, void* dz, int dz_ld, int dz_off, int dtype, float* rows, int nrows, int rows_ld,
                         void* stream);
/* y = act((z - mean)*scale + shift) [+ residual] (mean may be NULL = 0); out_mode YOLO_OUT_NHWC or
 * YOLO_OUT_UPSAMPLE2X. */
int yolo_bn_act_fwd(const void* z, int z_ld, int z_off, const float* mean, const float* scale, const float* shift, const void* residual,
                    int r_ld, int r_off, void* y, int y_ld, int y_off, int n, int h, int w, int c, int act, int out_mode,
                    int dtype, int32_t* nan_flag, void* stream);
/* Backward of act(BN(z)): dy -> dgamma, dbeta, dz (gradient of the raw conv output). gamma == NULL:
 * bare conv (model.py:86) -> only dbeta (= bias gradient) is produced and dz is dy itself. */
int yolo_bn_act_bwd(const void* dy, int dy_ld, int dy_off, const void* z, int z_ld, int z_off, const float* gamma,
                    const float* mean, const float* invstd, const float* scale, const float* shift, int m, int c, int act,
                    float* dgamma, float* dbeta, void* dz, int dz_ld, int dz_off, int dtype, void* workspace, size_t workspace_bytes,
                    void* stream);
/* gradient of nn.Upsample(scale_factor=2): dx(n,h,w,c) = sum of the 2x2 destinations in dup(n,2h,2w,.) */
int yolo_upsample2x_bwd(const void* dup, int d_ld, int d_off, void* dx, int x_ld, int x_off, int n, int h, int w, int c,
                        int dtype, void* stream);
/* dW (OIHW fp32) = sum over pixels dz (x) x; n,h,w = INPUT dims of the conv; dz has the output dims.
 * 16-bit operands with cin % 32 == 0 run on the bf16/f16 matrix cores, everything else on the f32 ones. */
size_t yolo_wgrad_workspace_bytes(int n, int h, int w, int cin, int cout, int ksize, int stride, int dtype);
int yolo_conv_wgrad(const void* dz, int dz_ld, int dz_off, const void* x, int x_ld, int x_off, float* dw_oihw, int n, int h,
                    int w, int cin, int cout, int ksize, int stride, int dtype, void* workspace, size_t workspace_bytes, void* stream);
/* Test hook for the gfx950 transposing LDS read (ds_read_b64_tr_b16) the 16-bit wgrad kernel is built on:
 * in = [64][ld] 16-bit image; out[lane][8] = what lane receives as its 32x32x16 MFMA operand, i.e.
 * in[8*(lane/32) + e][lane%32]. */
int yolo_debug_tr_probe(const void* in, void* out, int ld, void* stream);
/* Weights of the input-gradient convolution, from the fp32 OIHW master weights. flip = 1: stride-1 convs —
 * the result is a packed buffer for yolo_conv_fwd (same dtype) with (cin' = cout rounded up to 32,
 * cout' = cin, same ksize, stride 1): dx = conv(dz, W'). flip = 0: operand of yolo_conv_dgrad_s2 in the same dtype
 * (fp32: row-major W'; 16-bit: four tap-subset fragment streams, one per output parity class). */
size_t yolo_packed_dgrad_bytes(int cout, int cin, int ksize, int flip, int dtype);
int yolo_pack_weights_dgrad(const float* w_oihw, void* w_packed, int cout, int cin, int ksize, int flip, int dtype, void* stream);
/* dx (n,2ho,2wo,cin) = transposed 3x3 stride-2 conv of dz (n,ho,wo,cout) [+ residual] */
int yolo_conv_dgrad_s2(const void* dz, int dz_ld, int dz_off, const void* w_packed, const void* residual, int r_ld, int r_off,
                       void* dx, int dx_ld, int dx_off, int n, int ho, int wo, int cin, int cout, int dtype, void* stream);
/* upstream gradient in the head layout (B,3,g,g,D) fp32, any strides -> NHWC (B,g,g,ld) in dtype, channel a*D+k, pads 0 */
int yolo_head_grad_to_nhwc(const float* dp, const int64_t* strides5, void* out, int b, int g, int d, int ld, int dtype, void* stream);

/* ---- launch tables for the train-mode forward and the backward (train.py:41-82 run eagerly) -------------------------------- */
/* The reference's loop issues its step one Python statement at a time; here that is ~900 launches, and issued through an FFI
 * one by one the host is as slow as the GPU. A table of recorded calls is replayed by ONE call instead (the training
 * counterpart of yolo_conv_fwd_batch). yolo_call: fn = YOLO_FN_*, a[] = the arguments of that function in declaration order
 * WITHOUT the trailing stream: integers, size_t and pointers as 64-bit values, float arguments as the bits of a double.
 * Pointers inside a[] (descriptors, stride arrays, item tables) must stay valid while the table is in use.
 * yolo_reloc entries, sorted by call: at run time a[arg] of calls[call] is replaced by slots[slot] + offset - for the few pointers
 * that change from step to step (input batch, prediction tensors, upstream gradients). Stops at the first failing call. */
enum { YOLO_FN_FILL_ZERO = 1, YOLO_FN_COPY_D2D, YOLO_FN_NCHW_TO_NHWC, YOLO_FN_STEM_FWD, YOLO_FN_CONV_FWD, YOLO_FN_BN_STATS,
       YOLO_FN_BN_ACT_FWD, YOLO_FN_BN_ACT_BWD, YOLO_FN_UPSAMPLE2X_BWD, YOLO_FN_CONV_WGRAD, YOLO_FN_PACK_WEIGHTS_DGRAD,
       YOLO_FN_PACK_WEIGHTS_BATCH, YOLO_FN_CONV_DGRAD_S2, YOLO_FN_HEAD_GRAD_TO_NHWC, YOLO_FN_CONV_FWD_STATS,
       YOLO_FN_BN_STATS_FROM_PARTIALS, YOLO_FN_CONV_DGRAD_BSTATS, YOLO_FN_BN_ACT_BWD_ROWS, YOLO_FN_CONV_FWD_WS };
#define YOLO_CALL_MAX_ARGS 24
typedef struct yolo_call { int32_t fn; int32_t reserved; uint64_t a[YOLO_CALL_MAX_ARGS]; } yolo_call;
typedef struct yolo_reloc { int32_t call, arg, slot, reserved; int64_t offset; } yolo_reloc;
int yolo_run_calls(const yolo_call* calls, int n_calls, const yolo_reloc* relocs, int n_relocs, const uint64_t* slots, int n_slots,
                   void* stream);
/* the two halves of a fine-tune step (train.py:54 forward in train mode, :67 backward): same table format */
int yolo_train_fwd_batch(const yolo_call* calls, int n_calls, const yolo_reloc* relocs, int n_relocs, const uint64_t* slots, int n_slots,
                         void* stream);
int yolo_train_bwd_batch(const yolo_call* calls, int n_calls, const yolo_reloc* relocs, int n_relocs, const uint64_t* slots, int n_slots,
                         void* stream);
/* stream-ordered memset(0) / device-to-device copy: the two non-kernel operations of the step, as table entries */
int yolo_fill_zero(void* p, size_t bytes, void* stream);
int yolo_copy_d2d(void* dst, const void* src, size_t bytes, void* stream);

/* ---- letterbox (config.py:101-113: LongestMaxSize -> centred PadIfNeeded(0) -> /255 -> CHW); parity with cv2 UNPINNED --- */
/* img: uint8 (h, w, 3) on the device; out: fp32 (3, size, size). new_hw / pad_tl (host pointers, may be NULL) receive the
 * resized size and the top / left padding, which un-letterboxing the boxes needs (utils.py:475-501). */
int yolo_letterbox(const unsigned char* img_hwc, int h, int w, int size, float* out_chw, int* new_hw, int* pad_tl, void* stream);

/* ---- ground-truth tensors (next to the hot path: dataset.py:119-161) ------------------------ */
/* boxes (B, max_boxes, 5) fp32 [x, y, w, h, class] normalised to [0,1), counts (B) valid boxes per image (list order
 * matters), anchors (9,2) normalised and scale-major (config.ANCHORS flattened). Writes the three target tensors
 * (B,3,g,g,6), g = S/32, S/16, S/8, rows [x_cell, y_cell, w_cells, h_cells, obj in {1,0,-1}, class] — zero-filled
 * here first. Same assignment rule and quirks as the reference loop (see csrc/targets.hip). */
int yolo_build_targets(const float* boxes, const int32_t* counts, int max_boxes, const float* anchors_9x2, int b, int image_size,
                       float ignore_iou, float* t0, float* t1, float* t2, void* stream);

/* ---- evaluation: average precision per class (utils.py:193-274) ----------------------------- */
/* rows are [image, x, y, w, h, objectness, class] fp32. dets_sorted: class ascending, objectness descending (stable);
 * gts_sorted: class ascending, image ascending (stable); *_class_offsets: [num_classes + 1] row ranges.
 * assigned: n_gt int32 scratch, tp_flags: one fp32 per detection (1 = true positive), ap_per_class: trapezoid area of the
 * precision/recall curve, -1 for classes without ground truth (the reference skips them). */
int yolo_map_match(const float* dets_sorted, const int32_t* det_class_offsets, const float* gts_sorted, const int32_t* gt_class_offsets,
                   int num_classes, int n_gt, float iou_threshold, int center, int32_t* assigned, float* tp_flags, float* ap_per_class,
                   void* stream);

/* check_model_accuracy (utils.py:334-381) for one scale of one batch: counts5 += [class correct, n_obj, objectness correct
 * on object cells, objectness correct on no-object cells, n_noobj]; pred / target as for yolo_loss_fwd. */
int yolo_accuracy_counts(const float* pred, const int64_t* strides5, const float* target, int b, int g, int nc, float obj_threshold,
                         unsigned long long* counts5, void* stream);

/* ---- fused per-scale loss (optional replacement of YOLOLoss.forward, loss.py:29-81) -------- */
/* pred (B,3,g,g,5+nc) fp32 through element strides; target (B,3,g,g,6) fp32 contiguous
 * [x_cell,y_cell,w_cells,h_cells,obj in {1,0,-1},class]; anchors (3,2) in grid units.
 * losses4 = [5*box, 1*object, 0.5*noobj, 1*class] (each a mean over the selected cells of this call, as in
 * the reference), counts2 = [n_obj, n_noobj] for the backward. Deterministic (fixed-order fp64 sums), no host
 * dependence (graph-capturable), and — unlike the reference — no in-place mutation of pred / target. */
size_t yolo_loss_workspace_bytes(int b, int g);
int yolo_loss_fwd(const float* pred, const int64_t* strides5, const float* target, const float* anchors_3x2, int b, int g, int nc,
                  float* losses4, float* counts2, void* workspace, size_t workspace_bytes, void* stream);
/* dpred (B,3,g,g,5+nc) contiguous = sum_k grad_losses4[k] * d losses4[k] / d pred */
int yolo_loss_bwd(const float* pred, const int64_t* strides5, const float* target, const float* anchors_3x2, int b, int g, int nc,
                  const float* counts2, const float* grad_losses4, float* dpred, void* stream);

/* ---- optimizer step (code/train.py:171-172 torch.optim.SGD(model.parameters(), lr, momentum, weight_decay); :68 step) ---- */
/* One launch over every parameter tensor; the same bits as torch.optim.SGD's default implementation (each of its passes
 * rounds a + alpha * b once). items (device): n_items x yolo_sgd_item; an item with g == NULL is skipped; n < 0 marks a
 * momentum buffer that does not exist yet (first step: buf = g + wd * p). chunks (device): n_chunks x {item index, first
 * element}, one per yolo_sgd_chunk_elems() elements of every item. */
typedef struct yolo_sgd_item { float* p; const float* g; float* buf; long long n; } yolo_sgd_item;
int yolo_sgd_chunk_elems(void);
int yolo_sgd_step(const void* items_dev, const int32_t* chunks_dev, int n_chunks, float lr, float momentum, float dampening,
                  float weight_decay, int nesterov, int maximize, void* stream);
/* The same update with {lr, momentum, dampening, weight_decay} read from DEVICE memory (16-byte aligned float[4]) when the
 * kernel runs: a launch captured in a HIP graph then follows the LR scheduler of train.py:71-74,187-189 (stepped after
 * every batch) instead of replaying the capture-time values. nesterov needs momentum > 0 and dampening == 0 (caller's
 * responsibility here: the values are not on the host). */
int yolo_sgd_step_hp(const void* items_dev, const int32_t* chunks_dev, int n_chunks, const float* hyper4_dev, int nesterov, int maximize,
                     void* stream);

/* ---- post-processing ------------------------------------------------------------------- */
/* Replaces cells_to_boxes (utils.py:86-148) for one scale.
 * pred: (B,3,g,g,5+nc) fp32 addressed through element strides s[5] (so the reference's permuted
 * view and this library's contiguous head layout both work). is_pred != 0: pred[...,0:2] <-
 * sigmoid, pred[...,2:4] <- exp * anchors (IN PLACE, like the reference), obj = sigmoid,
 * cls = first argmax. boxes: (B, n_total, 6) fp32 rows [cx,cy,w,h,obj,cls]; this scale writes
 * rows box_offset + a*g*g + row*g + col  (n_total, box_offset let three scales share one
 * buffer in the reference's concatenation order, demo.py:44-51). is_pred == 0: 5+nc must be 6. */
int yolo_decode(void* pred, const int64_t* strides5, const float* anchors_3x2, int b, int g, int nc,
                int is_pred, float* boxes, int n_total, int box_offset, void* stream);
/* The three scales of one forward (is_pred = 1) in one launch: preds3[k] / strides15[5k..5k+4] / anchors3[k] / grids3[k] in the
 * reference's concatenation order (scale 0, 1, 2: demo.py:44-51); boxes (B, n_total, 6) with n_total = sum 3 g_k^2. */
int yolo_decode3(void* const* preds3, const int64_t* strides15, const float* const* anchors3, const int* grids3, int b, int nc,
                 float* boxes, int n_total, void* stream);
/* write_back = 0: the same boxes WITHOUT the reference's in-place sigmoid / exp write-back into the prediction tensors
 * (utils.py:106-110) - for callers that never look at the predictions again (demo.py:44-55, utils.py:300-321): the kernel's
 * writes drop to the algorithmic 24 bytes per box. yolo_decode takes the same choice as is_pred = 2. */
int yolo_decode3_ex(void* const* preds3, const int64_t* strides15, const float* const* anchors3, const int* grids3, int b, int nc,
                    int write_back, float* boxes, int n_total, void* stream);

/* Replaces non_max_suppression (utils.py:150-191) with calc_iou (utils.py:38-84) inlined,
 * batched over images. boxes: (B, n, 6) fp32. keep_idx: (B, n) int32, keep_count: (B) int32:
 * for image b the first keep_count[b] entries are indices into its n input rows, in the
 * reference's output order (objectness descending, ties in input order). Bit-exact contract:
 * same kept set and order as the reference for any fp32 input. center != 0 <=> box_format ==
 * "center"; every other string means (x1,y1,w,h) as is (utils.py:57-67). */
/* Ascending sort of n UNIQUE 64-bit keys, none equal to ~0 (2,048-key chunks in LDS + rank merge, the ordering kernels of yolo_nms):
 * the stable list sorts of calc_mAP (utils.py:206,232) as one sort of (major | minor | original index) keys. n <= 262,144. */
size_t yolo_sort_u64_workspace_bytes(int n);
int yolo_sort_u64(const uint64_t* keys, uint64_t* sorted, int n, void* workspace, size_t workspace_bytes, void* stream);
size_t yolo_nms_workspace_bytes(int b, int n);
int yolo_nms(const float* boxes, int b, int n, double iou_threshold, double obj_threshold, int center,
             int32_t* keep_idx, int32_t* keep_count, void* workspace, size_t workspace_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* YOLO_MI355X_H */
