/*
 * fosvos_hip.h - C ABI of libfosvos_hip.so: hand-written HIP (gfx950 / MI355X) kernels for the
 * OSVOS-VGG fine-tune hot path of klausondrag/FOSVOS.
 *
 * The reference has no FFI layer: its hot path runs through stock torch.nn modules
 * (src/networks/osvos_vgg.py:42-48,56,90-93 and src/layers/osvos_layers.py:17-54), i.e. the
 * arithmetic lives in third-party torch/cuDNN.  Each entry point below replaces one of those
 * call sites; the citation after "replaces:" names it (paths relative to the reference root).
 * The Python host (fosvos_amd/) binds these with ctypes; INTEGRATION.md shows the stub.
 *
 * Conventions
 *   - plain pointers and sizes only; every pointer is a DEVICE pointer unless stated otherwise;
 *   - the library allocates nothing persistent: outputs and workspaces belong to the caller,
 *     `*_workspace_bytes` tells how much scratch an op needs;
 *   - every call is asynchronous on `stream` (a hipStream_t) of `device`; no implicit sync;
 *   - return 0 on success, a negative FOSVOS_E_* code otherwise; fosvos_last_error() gives a
 *     thread-local message.  Nothing aborts or throws across the ABI;
 *   - re-entrant; the device is selected on every call (autograd runs backward on another thread);
 *   - no hidden state: the only objects that outlive a call are the inter-stream events of the multi-stream network
 *     calls, and those live in a caller-owned context (fosvos_ctx, one per model), never in the library.
 *
 * Tensor layouts
 *   frame   fp32 NCHW [N,3,H,W]                  (the reference's input layout)
 *   act     bf16 NHWC [N,H,W,C], C % 32 == 0     (internal feature maps, uint16_t storage)
 *   side    fp32 NHWC [N,h,w,16]                 (side_prep outputs)
 *   logit   fp32 [N,1,H,W]
 *   weight  fp32 OIHW [Co,Ci,3,3] masters as in the reference state_dict; bf16 packed copies
 *           [Ci/32][9][4][Co][8] for the MFMA kernels (see fosvos_pack_conv3x3_weights).
 */
#ifndef FOSVOS_HIP_H
#define FOSVOS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FOSVOS_ABI_VERSION 17

#define FOSVOS_OK 0
#define FOSVOS_E_SHAPE (-1)     /* unsupported or inconsistent shape            */
#define FOSVOS_E_ARG (-2)       /* null pointer / bad flag                       */
#define FOSVOS_E_WORKSPACE (-3) /* caller workspace too small                    */
#define FOSVOS_E_HIP (-4)       /* HIP runtime error (message has the hip text)  */

/* conv epilogue flags */
#define FOSVOS_CONV_RELU 1u    /* y = max(y, 0)                                   */
#define FOSVOS_CONV_OUT_F32 2u /* store fp32 NHWC instead of bf16 NHWC            */
#define FOSVOS_CONV_FP32_MATH 4u /* fosvos_conv7x7s2_first_fwd only: fp32 frame and weights on the vector ALU */

int fosvos_abi_version(void);

/* ---- launch profiler --------------------------------------------------------------------------------------------
 * Between _start and _stop every kernel the library launches (from any entry point, the native layer loops
 * included) is bracketed by a pair of timing hipEvents recorded on the stream the kernel is launched on, so the
 * two-stream backward is measured as it ships.  _stop synchronises the device and returns one record per kernel name
 * (the names rocprofv3 --kernel-trace prints, templates included where several instantiations run): launches, summed
 * event time in ms, summed algorithmic FLOPs (0 for the byte movers).  The events are created by _start (room for
 * max_launches launches; later ones go untimed) and destroyed by _stop: the only allocation the library ever makes,
 * and only on request.  One profile at a time per process. */
typedef struct fosvos_profile_record {
    char name[96];
    int launches;
    double ms;
    double flops;
} fosvos_profile_record;
int fosvos_profile_start(int device, int max_launches);
int fosvos_profile_stop(int device, fosvos_profile_record *out, int capacity, int *n_out);
const char *fosvos_last_error(void);
/* Name of the gfx target the code object was built for ("gfx950"). */
const char *fosvos_build_arch(void);

/* ---- execution context ------------------------------------------------------------------------------------------
 * The multi-stream network calls (fosvos_vgg_forward_streams, fosvos_vgg_backward, fosvos_resnet_forward with an
 * aux_stream) order their streams with timing-disabled hipEvents, and fosvos_vgg_backward publishes its gradient
 * buckets through such events.  They belong to a context the CALLER creates, owns and destroys - one per model (or per
 * host thread that drives a model): two models trained from two host threads on one GPU never share an event, and
 * fosvos_vgg_grad_bucket_wait refers to the last backward pass OF ITS CONTEXT, not "of the device".  A context is bound
 * to one device; calls on one context must not overlap in time (one host thread at a time).  Nothing else in the
 * library keeps mutable state between calls (immutable per-device kernel attributes aside).
 * _destroy does not synchronise: the caller makes sure the streams are done with the events (as for any hipEvent). */
typedef struct fosvos_ctx fosvos_ctx;
int fosvos_ctx_create(int device, fosvos_ctx **ctx_out);
int fosvos_ctx_destroy(fosvos_ctx *ctx);
int fosvos_ctx_device(const fosvos_ctx *ctx); /* the device the context was created on, -1 for NULL */

/* ---- layout helpers (test/debug plumbing and the model's input/feature export) -------------- */
/* fp32 NCHW -> bf16 NHWC, channels zero-padded from C to Cpad (Cpad % 8 == 0, Cpad >= C). */
int fosvos_nchw_f32_to_nhwc_bf16(const float *src, uint16_t *dst, int N, int C, int H, int W, int Cpad,
                                 int device, void *stream);
/* bf16 NHWC (first C of Cpad channels) -> fp32 NCHW. */
int fosvos_nhwc_bf16_to_nchw_f32(const uint16_t *src, float *dst, int N, int C, int H, int W, int Cpad,
                                 int device, void *stream);
/* fp32 NHWC <-> fp32 NCHW */
int fosvos_nhwc_f32_to_nchw_f32(const float *src, float *dst, int N, int C, int H, int W, int device, void *stream);
int fosvos_nchw_f32_to_nhwc_f32(const float *src, float *dst, int N, int C, int H, int W, int device, void *stream);

/* ---- weight packing ---------------------------------------------------------------------------
 * fp32 OIHW master -> the two bf16 images the MFMA kernels read.
 *   w_fwd  [ceil(Ci/32)][9][4][Co_pad][8] : element (co,ci,tap) of the forward conv
 *   w_dgrad[ceil(Co/32)][9][4][Ci_pad][8] : element (ci,co,8-tap), i.e. the transposed, 180-degree
 *                                           rotated filter, so dgrad runs as a forward conv
 * Channel counts are zero-padded to multiples of 32 on the contraction side and to 16 on the
 * output side (Co_pad = roundup(Co,16), Ci_pad = roundup(Ci,16)).  Either output may be NULL.
 * replaces: the weight operand of nn.Conv2d (src/networks/osvos_vgg.py:42,92). */
int fosvos_pack_conv3x3_weights(const float *w_oihw, int Co, int Ci, uint16_t *w_fwd, uint16_t *w_dgrad,
                                int device, void *stream);
size_t fosvos_packed_weight_elems(int out_ch, int in_ch); /* elements of one packed image */

/* Both images of several layers in ONE launch (what the shipped module does after every optimizer step; replaces
 * 2 launches per layer).  Entry i packs w [Co,Ci,3,3] into w_fwd and/or w_dgrad (either may be NULL); Ci % 32 == 0.
 * `entries` is host memory, read before the call returns. */
typedef struct fosvos_pack_entry {
    const float *w;
    uint16_t *w_fwd, *w_dgrad;
    int Co, Ci;
} fosvos_pack_entry;
int fosvos_pack_conv3x3_weights_multi(const fosvos_pack_entry *entries, int n, int device, void *stream);

/* ---- first layer: conv1_1 (Ci = 3) straight from the fp32 NCHW frame --------------------------
 * y[N,H,W,Co] bf16 = relu(conv3x3(frame, w, pad 1) + b), fp32 VALU arithmetic (K = 27).
 * replaces: stages[0][0..1] = Conv2d(3,64,3,pad=1)+ReLU (src/networks/osvos_vgg.py:92-93). */
int fosvos_conv3x3_first_fwd(const float *frame, const float *w_oihw, const float *bias, uint16_t *y, int N, int H,
                             int W, int Co, int device, void *stream);
/* The same, also writing the ReLU mask of y as one bit per element: relu_bits [N,H,W,Co/8] bytes (NULL = none), bit e of
 * byte g set where y[..., 8 g + e] > 0 - what fosvos_conv3x3_dgrad_bits reads instead of y. */
int fosvos_conv3x3_first_fwd_bits(const float *frame, const float *w_oihw, const float *bias, uint16_t *y,
                                  uint8_t *relu_bits, int N, int H, int W, int Co, int device, void *stream);
/* Launch geometry of the above: 8 x 32-pixel tiles and the (persistent) workgroups that walk them; tiles > workgroups
 * means every workgroup loops over several tiles with its double-buffered staging.  Host arithmetic only. */
int fosvos_conv3x3_first_plan(int N, int H, int W, int *tiles, int *workgroups);
/* dw[Co,3,3,3], db[Co] (fp32, overwritten) from the frame and dy[N,H,W,Co] bf16.  No dgrad: the
 * image needs no gradient.  workspace: fosvos_conv3x3_first_wgrad_workspace_bytes. */
int fosvos_conv3x3_first_wgrad(const float *frame, const uint16_t *dy, float *dw_oihw, float *dbias, int N, int H,
                               int W, int Co, void *workspace, size_t workspace_bytes, int device, void *stream);
size_t fosvos_conv3x3_first_wgrad_workspace_bytes(int N, int H, int W, int Co);

/* ---- 3x3 conv, pad 1, stride 1, as an implicit GEMM on bf16 MFMA with fp32 accumulation -------
 * y[N,H,W,Co] = epilogue( sum_{tap,ci} x[N,H+dy,W+dx,ci] * w[co,ci,tap] + bias[co] )
 *   x        bf16 NHWC with Ci_pad = roundup(Ci,32) channels
 *   w_packed the matching image from fosvos_pack_conv3x3_weights
 *   bias     fp32 [Co] or NULL
 *   y        bf16 NHWC with Co channels (Co % 16 == 0), or fp32 NHWC when FOSVOS_CONV_OUT_F32
 * replaces: Conv2d(3x3,pad=1)[+ReLU] in stages[*] and side_prep[*]
 *           (src/networks/osvos_vgg.py:42,92-93). */
int fosvos_conv3x3_fwd(const uint16_t *x, const uint16_t *w_packed, const float *bias, void *y, int N, int H, int W,
                       int Ci, int Co, unsigned flags, void *workspace, size_t workspace_bytes, int device,
                       void *stream);

/* Stride 2 (pad 1): y[N,(H-1)/2+1,(W-1)/2+1,Co] = act(conv3x3(x) + bias) at the even pixels.  Computed by the
 * stride-1 MFMA kernel with a subsampling store: 4x the arithmetic of a strided kernel, which the MFMA rate more than pays
 * for at Ci >= 32.  Co % 64 == 0 or Co = 32; one launch, the workspace is not used.
 * replaces: the stride-2 conv1 + bn1 + relu that opens ResNet stages 2-4 (src/networks/osvos_resnet.py:101-103). */
int fosvos_conv3x3_s2_fwd(const uint16_t *x, const uint16_t *w_packed, const float *bias, uint16_t *y, int N, int H, int W,
                          int Ci, int Co, unsigned flags, void *workspace, size_t workspace_bytes, int device, void *stream);
/* The residual form: y = act(conv3x3(x) + bias + addend), ReLU (FOSVOS_CONV_RELU) applied AFTER the add; addend is bf16
 * [N,H,W,Co] or NULL.  Co % 64 == 0 or Co = 32; without an addend also Co = 16, then optionally FOSVOS_CONV_OUT_F32 (the
 * side_prep form).  Always ONE launch (no split-K, the workspace is not used): this is the inference
 * form, whose chain of small dependent launches pays more for a second kernel than the split wins.
 * replaces: conv2 + bn2 + `out += residual` + relu of torchvision's BasicBlock (src/networks/osvos_resnet.py:203-214). */
int fosvos_conv3x3_fwd_add(const uint16_t *x, const uint16_t *w_packed, const float *bias, const uint16_t *addend,
                           void *y, int N, int H, int W, int Ci, int Co, unsigned flags, void *workspace,
                           size_t workspace_bytes, int device, void *stream);
/* fosvos_conv3x3_fwd plus MaxPool2d(2,2,ceil_mode=True) of its output in the same launch (the last conv of a VGG stage
 * feeds both the side branch and the pool, src/networks/osvos_vgg.py:90-93): y as above (bf16, Co % 64 == 0),
 * y_pool = [N, ceil(H/2), ceil(W/2), Co] bf16.  Saves the pool kernel's re-read of y. */
int fosvos_conv3x3_fwd_pool(const uint16_t *x, const uint16_t *w_packed, const float *bias, uint16_t *y,
                            uint16_t *y_pool, int N, int H, int W, int Ci, int Co, unsigned flags, void *workspace,
                            size_t workspace_bytes, int device, void *stream);
/* Scratch for fwd/dgrad with `in_ch` contraction and `out_ch` output channels (split-K partial
 * slabs for layers whose pixel count alone cannot fill 256 CUs; 0 when none is needed). */
size_t fosvos_conv3x3_workspace_bytes(int N, int H, int W, int in_ch, int out_ch);
/* Which kernel instantiation fosvos_conv3x3_fwd / _fwd_pool / _dgrad launch for a shape (`in_ch` contraction, `out_ch`
 * output channels): the pixel tile (tile_h x tile_w) x tile_co output channels of one workgroup, the K split count and the
 * number of workgroups.  Pure host arithmetic (no device call); the parity tests use it to assert that the cases they run
 * really select the instantiations the 480x854 step runs (k_conv3x3_igemm<Tile<tile_h, tile_w, tile_co, ..>>). */
typedef struct fosvos_conv3x3_plan_info {
    int tile_h, tile_w, tile_co;
    int k_splits;
    int workgroups; /* per K split */
    int persistent; /* 1: the persistent eight-wave forward kernel k_conv3x3_pp (conv_pp.hip): `workgroups` of them, one per
                     * CU, walk the 8 x 32-pixel x 64-channel tiles in two groups of four waves that alternate between the
                     * MFMAs of a K chunk and the memory work of the other tile */
} fosvos_conv3x3_plan_info;
int fosvos_conv3x3_plan(int N, int H, int W, int in_ch, int out_ch, fosvos_conv3x3_plan_info *out);
/* The same question for a FORWARD launch (fosvos_conv3x3_fwd / _fwd_pool with these flags): bf16-output launches whose
 * tiles fill the chip run the persistent kernel; every other form (data gradient, residual add, stride 2, fp32 output) is
 * what fosvos_conv3x3_plan reports. */
int fosvos_conv3x3_fwd_plan(int N, int H, int W, int in_ch, int out_ch, unsigned flags, fosvos_conv3x3_plan_info *out);
/* dx = mask( dgrad(dy) ) + addend:  the same kernel run on the rotated/transposed filter image.
 *   dy       bf16 NHWC, Co_pad = roundup(Co,32) channels (Co = the forward op's OUTPUT channels)
 *   relu_src bf16 NHWC [N,H,W,Ci] or NULL: where relu_src <= 0 the computed gradient is zeroed
 *            (ReLU backward of the layer that produced x, fused)
 *   addend   bf16 NHWC [N,H,W,Ci] or NULL: added after masking (gradient arriving at the same
 *            tensor through another consumer); may alias dx
 * replaces: autograd's conv backward-data for the same modules. */
int fosvos_conv3x3_dgrad(const uint16_t *dy, const uint16_t *w_dgrad_packed, const uint16_t *relu_src,
                         const uint16_t *addend, uint16_t *dx, int N, int H, int W, int Ci, int Co, void *workspace,
                         size_t workspace_bytes, int device, void *stream);
/* The same with the ReLU mask given as ONE BIT per element: relu_bits [N,H,W,Ci/8] bytes, bit e of byte g set where channel
 * 8 g + e of the producing layer's output is > 0 (fosvos_conv3x3_first_fwd_bits writes it for conv1_1).  Bit for bit the
 * result of fosvos_conv3x3_dgrad on the bf16 image the bits were taken from; 1/16 of the mask bytes - conv1_2's data gradient
 * at 480x854 is bound by HBM traffic. */
int fosvos_conv3x3_dgrad_bits(const uint16_t *dy, const uint16_t *w_dgrad_packed, const uint8_t *relu_bits,
                              const uint16_t *addend, uint16_t *dx, int N, int H, int W, int Ci, int Co, void *workspace,
                              size_t workspace_bytes, int device, void *stream);
/* Data gradient of a conv whose INPUT x also feeds a MaxPool2d(2, 2, ceil_mode=True) - side_prep[i] on a stage output
 * (src/networks/osvos_vgg.py:61-83: `side_prep[i](x)` beside `stages[i+1](x)`) - with the pool's backward in the same pass:
 *   dx = [x > 0] * dgrad(dy) + maxpool2x2_ceil_bwd(x, d_pooled)         (fosvos_maxpool2x2_ceil_bwd with relu_mask = 1)
 *   x        bf16 NHWC [N,H,W,Ci]: the post-ReLU stage output (ReLU mask and the pool's arg-max source)
 *   d_pooled bf16 NHWC [N,ceil(H/2),ceil(W/2),Ci]: gradient wrt the pooled map
 * Bit for bit the result of the pool backward followed by fosvos_conv3x3_dgrad(dy, w, x, addend = that, dx), which is also
 * what runs for shapes whose plan has no fused form (split-K launches, Ci not a multiple of 64).  dx must not alias x.
 * replaces: autograd's backward of the two consumers of a stage output. */
int fosvos_conv3x3_dgrad_unpool(const uint16_t *dy, const uint16_t *w_dgrad_packed, const uint16_t *x,
                                const uint16_t *d_pooled, uint16_t *dx, int N, int H, int W, int Ci, int Co,
                                void *workspace, size_t workspace_bytes, int device, void *stream);
/* dw[Co,Ci,3,3] (fp32 OIHW) and db[Co] (may be NULL) from x[N,H,W,Ci] and dy[N,H,W,Co_pad] (bf16).
 * Deterministic: split partial sums are written as slabs to the workspace and reduced in a fixed
 * order.  accumulate != 0 adds into dw/db instead of overwriting.
 * replaces: autograd's conv backward-weight/bias. */
int fosvos_conv3x3_wgrad(const uint16_t *x, const uint16_t *dy, float *dw_oihw, float *dbias, int N, int H, int W,
                         int Ci, int Co, int accumulate, void *workspace, size_t workspace_bytes, int device,
                         void *stream);
size_t fosvos_conv3x3_wgrad_workspace_bytes(int N, int H, int W, int Ci, int Co);
/* The two halves of fosvos_conv3x3_wgrad, for callers that batch or time them apart (the network call queues every
 * layer's reduction and runs them together): _slabs runs the MFMA kernel, leaving per-split partial sums (and, with
 * with_bias != 0, bias partials) in the workspace; _reduce turns that workspace into dw / db (db NULL = no bias). */
int fosvos_conv3x3_wgrad_slabs(const uint16_t *x, const uint16_t *dy, int with_bias, int N, int H, int W, int Ci, int Co,
                               void *workspace, size_t workspace_bytes, int device, void *stream);
int fosvos_conv3x3_wgrad_reduce(float *dw_oihw, float *dbias, int N, int H, int W, int Ci, int Co, int accumulate,
                                void *workspace, size_t workspace_bytes, int device, void *stream);

/* ---- 2x2 stride-2 ceil-mode max pool on bf16 NHWC ---------------------------------------------
 * y[N,ceil(H/2),ceil(W/2),C]; ragged last row/column windows hold 2 or 1 elements.
 * replaces: MaxPool2d(2,2,ceil_mode=True) (src/networks/osvos_vgg.py:90). */
int fosvos_maxpool2x2_ceil_fwd(const uint16_t *x, uint16_t *y, int N, int H, int W, int C, int device, void *stream);
/* dx[N,H,W,C]: dy routed to the first maximum of each window in (row, column) scan order, zero
 * elsewhere; relu_mask != 0 also zeroes it where x <= 0 (ReLU backward of the producer, fused). */
int fosvos_maxpool2x2_ceil_bwd(const uint16_t *x, const uint16_t *dy, uint16_t *dx, int N, int H, int W, int C,
                               int relu_mask, int device, void *stream);

/* ---- fused side-output head -------------------------------------------------------------------
 * For the four scales s (stride f = 2,4,8,16; deconv kernel k = 2f) with side[s] fp32 NHWC
 * [N,hs[s],ws[s],16]:
 *   fused[n,0,Y,X]   = fuse_b + sum_s sum_c fuse_w[16 s + c] * crop(up_s(side[s]))[c,Y,X]
 *   side_out[s][..]  = crop(up1_s(dsn_b[s] + sum_c dsn_w[s][c] * side[s][c]))        (optional)
 * up_s is the transposed conv with the DIAGONAL of upscale[s].weight, passed channel-fastest as
 * filt[s] = [k][k][16] fp32 (one k x k filter per channel); up1_s uses filt1[s] = [k][k].  crop is the
 * reference's centre crop (floor(d/2) leading pixels removed).
 * filt_uniform: bit s set = the caller PROMISES that the 16 channel filters of filt[s] are identical, element for element
 * (what interp_surgery writes - src/layers/osvos_layers.py:70-81 - and the optimizers keep: lr 0,
 * src/util/network_provider.py:110-111,154-155).  The kernels then contract the channels once per low-resolution pixel
 * and apply ONE k x k filter (same sums in another order; 16x fewer multiply-adds and LDS reads).  0 = the general form.
 * A set bit with filters that differ between channels evaluates channel 0's filter for all: the check is the caller's.
 * replaces: upscale[s], upscale_[s], score_dsn[s], center_crop, torch.cat and fuse
 *           (src/networks/osvos_vgg.py:69-82, src/layers/osvos_layers.py:47-54). */
int fosvos_head_fwd(const float *const side[4], const int hs[4], const int ws[4], const float *const filt[4],
                    const float *const filt1[4], const float *dsn_w /*[4][16]*/, const float *dsn_b /*[4]*/,
                    const float *fuse_w /*[64]*/, const float *fuse_b /*[1]*/, float *fused,
                    float *const side_out[4] /* all NULL or all set */, int N, int H, int W, int filt_uniform, int device,
                    void *stream);
/* Backward of the above.
 *   d_fused [N,1,H,W] fp32 or NULL;  d_side_out[s] [N,1,H,W] fp32 or NULL (all or none)
 *   d_side[s]   bf16 NHWC [N,hs,ws,32]: channels 0..15 = gradient wrt side[s], 16..31 = 0 (the
 *               padded image the side_prep dgrad/wgrad MFMA kernels read)
 *   d_fuse_w[64], d_fuse_b[1], d_dsn_w[4*16], d_dsn_b[4]: fp32, overwritten (d_dsn_* may be NULL
 *               when d_side_out is NULL).
 * workspace: fosvos_head_bwd_workspace_bytes. */
int fosvos_head_bwd(const float *const side[4], const int hs[4], const int ws[4], const float *const filt[4],
                    const float *const filt1[4], const float *dsn_w, const float *fuse_w, const float *d_fused,
                    const float *const d_side_out[4], uint16_t *const d_side[4], float *d_fuse_w, float *d_fuse_b,
                    float *d_dsn_w, float *d_dsn_b, int N, int H, int W, int filt_uniform /* as fosvos_head_fwd */,
                    void *workspace, size_t workspace_bytes, int device, void *stream);
size_t fosvos_head_bwd_workspace_bytes(int N, int H, int W);

/* ---- class-balanced BCE-with-logits, loss and gradient in one call ----------------------------
 * y_i = label_i >= 0.5; Np = #y, Nn = numel - Np;
 * loss = (Nn/numel) * sum_{y=1} l_i + (Np/numel) * sum_{y=0} l_i,  l_i = max(x,0) - x y + log(1+exp(-|x|))
 * [divided by numel when size_average]; grad_i = scale * w_i * (sigmoid(x_i) - y_i) with the same
 * weights (and the 1/numel factor when size_average).  Reductions run over the whole batch tensor
 * and are deterministic (fixed-order fp64 partials).  loss_out: one fp32 on the device.
 * grad may be NULL.  workspace: fosvos_cbce_workspace_bytes.
 * replaces: class_balanced_cross_entropy_loss + its autograd (src/layers/osvos_layers.py:17-44). */
int fosvos_cbce_loss(const float *logits, const float *label, int64_t numel, int size_average, float grad_scale,
                     float *loss_out, float *grad, void *workspace, size_t workspace_bytes, int device,
                     void *stream);
size_t fosvos_cbce_workspace_bytes(int64_t numel);
/* The loss of EVERY FRAME of a batch separately, in one set of launches: logits / label / grad are [n_frames][frame_numel],
 * loss_out gets n_frames values, each what fosvos_cbce_loss returns for that frame alone (own class counts).  The online
 * loop uses it when it runs several micro-batches of an accumulation cycle as one batched pass: the reference computes
 * the loss per frame (src/train_online.py:80 on a [1,1,H,W] tensor).  frame_numel must be a multiple of 4 when
 * n_frames > 1; workspace: n_frames x fosvos_cbce_workspace_bytes. */
int fosvos_cbce_loss_frames(const float *logits, const float *label, int64_t frame_numel, int n_frames, int size_average,
                            float grad_scale, float *loss_out, float *grad, void *workspace, size_t workspace_bytes,
                            int device, void *stream);
/* fosvos_cbce_loss_frames in its three launches, for a caller that has other work to put between them (the online loop):
 * FOSVOS_CBCE_COUNT needs only the labels (it can run before the forward pass that produces the logits), FOSVOS_CBCE_LOSS
 * writes grad and the loss partials (the backward pass needs nothing more), FOSVOS_CBCE_FINISH writes loss_out (it can run
 * behind the backward pass).  `parts` = the stages of this call, run in that order; arguments a stage does not read may
 * be NULL (COUNT: logits, loss_out, grad; FINISH: logits, label, grad).  The stages of one loss share `workspace`: same
 * pointer, untouched by anything else in between.  All three in one call = fosvos_cbce_loss_frames, bit for bit. */
#define FOSVOS_CBCE_COUNT 1
#define FOSVOS_CBCE_LOSS 2
#define FOSVOS_CBCE_FINISH 4
int fosvos_cbce_loss_frames_parts(const float *logits, const float *label, int64_t frame_numel, int n_frames,
                                  int size_average, float grad_scale, float *loss_out, float *grad, void *workspace,
                                  size_t workspace_bytes, int parts, int device, void *stream);
/* The same loss on ONE SHARD of a batch that is split over data-parallel ranks.  The reference counts positives /
 * negatives over the whole batch tensor (src/layers/osvos_layers.py:28-39), so the class weights of a shard must
 * come from the whole batch: batch_counts = DEVICE double[2] {positives, pixels} summed over all shards (the caller
 * all-reduces them).  loss_out is this shard's part of the batch loss (the parts add up to it); with size_average the
 * division is by the batch's pixel count.  grad is what the single-process batch would hold for these pixels. */
int fosvos_cbce_loss_batch_counts(const float *logits, const float *label, int64_t numel, int size_average,
                                  float grad_scale, const double *batch_counts, float *loss_out, float *grad,
                                  void *workspace, size_t workspace_bytes, int device, void *stream);

/* ---- SGD with momentum, torch.optim.SGD semantics, many tensors per launch --------------------
 * For tensor t with n[t] elements: g = grad + wd[t]*p; buf = first_step ? g : momentum*buf + g;
 * p -= lr[t]*buf.  `table` is a DEVICE array of n_tensors fosvos_sgd_entry records.
 * `first_step` is a flag word: bit 0 = first step (buf = g), bit 1 (FOSVOS_SGD_ZERO_GRAD) = every gradient is overwritten
 * with zeros once it has been read - optimizer.step() + optimizer.zero_grad() (src/train_online.py:100-104) in one pass over
 * the gradients instead of a step and a memset.
 * replaces: optim.SGD.step for the groups of src/util/network_provider.py:144-159 / 98-125. */
#define FOSVOS_SGD_FIRST_STEP 1
#define FOSVOS_SGD_ZERO_GRAD 2
typedef struct fosvos_sgd_entry {
    float *param;
    const float *grad; /* written (zeroed) only under FOSVOS_SGD_ZERO_GRAD */
    float *momentum_buf;
    int64_t numel;
    float lr;
    float weight_decay;
} fosvos_sgd_entry;
int fosvos_sgd_momentum_step(const fosvos_sgd_entry *table, int n_tensors, int64_t max_numel, float momentum,
                             int first_step, int device, void *stream);

/* ---- thin-channel ResNet inference path (OSVOS_RESNET and the nets prune.py derives from it) ------
 * Activations: bf16 NHWC [N,H,W,Cp] with Cp = channels rounded up to a multiple of 8, padded channels zero.
 * Eval-mode BatchNorm is folded into the conv in front of it when the weights are packed:
 *   s[o] = bn_weight[o] / sqrt(bn_var[o] + eps);  w'[o] = w[o] * s[o];  b'[o] = bn_bias[o] - bn_mean[o] * s[o]
 *   (+ conv_bias[o] * s[o]; bn_* all NULL: no BatchNorm, b' = conv_bias or 0).
 * Packed image: uint32 [Cip/8][k*k][4][Cop] (two bf16 input channels per word), fosvos_conv2d_packed_dwords words;
 * folded bias: fosvos_conv2d_bias_elems floats (zero beyond Co).  The contraction runs on the vector ALU
 * (v_dot2c_f32_bf16, fp32 accumulate) with any channel counts - these layers are too thin for the MFMA path.
 * replaces: nn.Conv2d + nn.BatchNorm2d (+ residual add) + nn.ReLU of torchvision's BasicBlock / Bottleneck as wired by
 *           src/networks/osvos_resnet.py:91-121,187-216, and the side_prep convs of :135. */
/* The fold alone (fp32 OIHW in, fp32 OIHW out, bias_out[Co]): for 3x3 stride-1 layers whose channel counts fit the MFMA
 * path (Ci % 32 == 0, Co % 64 == 0), which then go through fosvos_pack_conv3x3_weights and fosvos_conv3x3_fwd[_add]. */
int fosvos_fold_conv_bn(const float *w_oihw, int Co, int Ci, int k, const float *conv_bias, const float *bn_weight,
                        const float *bn_bias, const float *bn_mean, const float *bn_var, float eps, float *w_folded,
                        float *bias_out, int device, void *stream);
size_t fosvos_conv2d_packed_dwords(int out_ch, int in_ch, int k);
size_t fosvos_conv2d_bias_elems(int out_ch);
int fosvos_pack_conv2d_bn(const float *w_oihw, int Co, int Ci, int k /* 1 or 3 */, const float *conv_bias,
                          const float *bn_weight, const float *bn_bias, const float *bn_mean, const float *bn_var,
                          float eps, uint32_t *w_packed, float *bias_out, int device, void *stream);
/* y[N,Ho,Wo,Cop] = act(conv_k,stride,pad=k/2(x) + bias + addend); Ho = (H + 2 (k/2) - k) / stride + 1.
 * addend: bf16 [N,Ho,Wo,Cop] or NULL (the residual branch).  flags: FOSVOS_CONV_RELU, FOSVOS_CONV_OUT_F32 (3x3
 * stride 1 without addend only: the fp32 side maps the head reads). */
int fosvos_conv2d_fwd(const uint16_t *x, const uint32_t *w_packed, const float *bias, const uint16_t *addend, void *y,
                      int N, int H, int W, int Ci, int Co, int k, int stride, unsigned flags, int device, void *stream);
/* First layer: 7x7 stride 2 pad 3 on the fp32 NCHW frame (3 channels) + folded BatchNorm (+ ReLU) -> bf16 NHWC
 * [N,(H-1)/2+1,(W-1)/2+1,Cop].  Packed image (fosvos_conv7x7_packed_elems floats): fp32 [49*3][Cop], then the same
 * filter as bf16 MFMA fragments.  By default (Cop <= 64) the layer runs on the matrix cores with the frame rounded to
 * bf16 like every other activation of the path; FOSVOS_CONV_FP32_MATH keeps frame and weights fp32 on the vector ALU.
 * replaces: layer_base conv1 + bn1 + relu (src/networks/osvos_resnet.py:92-94). */
size_t fosvos_conv7x7_packed_elems(int out_ch);
int fosvos_pack_conv7x7_bn(const float *w_oihw, int Co, const float *bn_weight, const float *bn_bias,
                           const float *bn_mean, const float *bn_var, float eps, float *w_packed, float *bias_out,
                           int device, void *stream);
int fosvos_conv7x7s2_first_fwd(const float *frame, const float *w_packed, const float *bias, uint16_t *y, int N, int H,
                               int W, int Co, unsigned flags, int device, void *stream);
/* layer_base in ONE launch: conv 7x7/2 + folded BatchNorm + ReLU + MaxPool2d(3, 2, 1) -> bf16 NHWC
 * [N,Hp,Wp,Cop], Hp = ((H-1)/2+1 - 1)/2 + 1; the conv map is never written.  MFMA form only (Cop <= 64); same bits as
 * fosvos_conv7x7s2_first_fwd followed by fosvos_maxpool3x3s2_fwd.
 * replaces: layer_base (src/networks/osvos_resnet.py:91-96). */
int fosvos_conv7x7s2_pool_first_fwd(const float *frame, const float *w_packed, const float *bias, uint16_t *y_pooled, int N,
                                    int H, int W, int Co, int device, void *stream);
/* MaxPool2d(kernel 3, stride 2, padding 1) on bf16 NHWC, C % 8 == 0: [N,H,W,C] -> [N,(H-1)/2+1,(W-1)/2+1,C].
 * replaces: layer_base maxpool (src/networks/osvos_resnet.py:95). */
int fosvos_maxpool3x3s2_fwd(const uint16_t *x, uint16_t *y, int N, int H, int W, int C, int device, void *stream);
/* Side-output head with a free stride per scale (OSVOS_RESNET: 4, 8, 16, 32), forward only:
 *   fused[n,0,Y,X]  = fuse_b + sum_s sum_c sum_taps filt[s][ky][kx][c] * side[s][n,i,j,c]
 *   side_out[s][..] = sum_taps filt1[s][ky][kx] * (dsn_b[s] + sum_c dsn_w[s][c] * side[s][n,i,j,c])   (optional)
 * over the taps of a transposed conv with kernel 2 f_s and stride f_s, centre-cropped to H x W as the reference
 * does.  filt[s] = [2f][2f][16] is the upscale_side_prep[s] weight CONTRACTED with the fuse weights of scale s
 * (G[ky][kx][ci] = sum_co fuse_w[16 s + co] * W[ci][co][ky][kx]; both maps are linear), so the full 16 -> 16
 * transposed conv, the concat and the 1x1 fuse cost one filter.  filt1[s] = [2f][2f].
 * replaces: upscale_side_prep, score_dsn, upscale_score_dsn, center_crop, torch.cat, layer_fuse
 *           (src/networks/osvos_resnet.py:53-66). */
int fosvos_deconv_head_fwd(const float *const side[4], const int hs[4], const int ws[4], const int stride[4],
                           const float *const filt[4], const float *const filt1[4], const float *dsn_w /*[4][16]*/,
                           const float *dsn_b /*[4]*/, const float *fuse_b /*[1]*/, float *fused,
                           float *const side_out[4] /* all NULL or all set */, int N, int H, int W, int device,
                           void *stream);

/* Whole-network forward of the ResNet family: ONE call issues every kernel of OSVOS_RESNET.forward (first conv, pool,
 * every block with its residual branch, the four side_prep convs, the head) over a caller-provided arena.  The
 * structs live in HOST memory for the duration of the call and hold device pointers to images made by the pack
 * calls above; `blocks` is a host array listing the blocks stage after stage.
 * replaces: OSVOS_RESNET.forward (src/networks/osvos_resnet.py:42-68) as driven from Python. */
typedef struct fosvos_conv2d_desc {
    const void *w_packed;       /* kind 0: fosvos_pack_conv2d_bn image; kind 1: fosvos_pack_conv3x3_weights forward image */
    const float *bias;          /* kind 0: fosvos_conv2d_bias_elems floats; kind 1: Co floats */
    int Ci, Co, k, stride;
    int kind;                   /* 0 = vector-ALU direct conv, 1 = MFMA implicit GEMM (3x3, Ci % 32 == 0, Co % 64 == 0 or Co = 32; Co = 16
                                 * for side_prep; stride 2 through fosvos_conv3x3_s2_fwd) */
} fosvos_conv2d_desc;
typedef struct fosvos_resnet_block {
    fosvos_conv2d_desc conv[3]; /* conv+bn(+relu) chain; the last one adds the residual before its ReLU */
    int n_convs;                /* 2 = BasicBlock, 3 = Bottleneck */
    int has_down;
    fosvos_conv2d_desc down;    /* 1x1 downsample conv + bn of the residual branch */
} fosvos_resnet_block;
typedef struct fosvos_resnet_net {
    const float *first_w;       /* fosvos_pack_conv7x7_bn image */
    const float *first_b;
    int first_co;
    int first_fp32_math;        /* != 0: FOSVOS_CONV_FP32_MATH for the first layer */
    int first_unfused;          /* != 0: first layer and max pool as two launches (A/B runs) */
    int blocks_per_stage[4];
    const fosvos_resnet_block *blocks;
    fosvos_conv2d_desc side[4]; /* side_prep: 3x3 stride 1, Co = 16 */
    const float *filt[4];       /* as fosvos_deconv_head_fwd */
    const float *filt1[4];
    int stride[4];
    const float *dsn_w, *dsn_b, *fuse_b;
} fosvos_resnet_net;
size_t fosvos_resnet_arena_bytes(const fosvos_resnet_net *net, int N, int H, int W);
/* fused: [N,1,H,W] fp32; side_out: four [N,1,H,W] fp32 buffers or NULL.
 * aux_stream (a second hipStream_t of the same device, or NULL): when given, the kernels the trunk does not wait for
 * right away - each stage's side_prep conv and the 1x1 downsample convs - are issued on it beside the trunk's chain
 * (thin layers leave most of the chip idle), ordered by events of `ctx` (required with an aux_stream, else it may be
 * NULL); the caller sees ordinary single-stream semantics on `stream`.  Measured on MI355X at 1920x1080 this
 * costs 0.1 ms per frame more than it saves (cross-stream waits), so the shipped host passes NULL. */
int fosvos_resnet_forward(const fosvos_resnet_net *net, const float *frame, int N, int H, int W, void *arena,
                          size_t arena_bytes, float *fused, float *const side_out[4], int device, void *stream,
                          fosvos_ctx *ctx, void *aux_stream);

/* ---- whole-network entry points ------------------------------------------------------------------
 * The reference drives ~60 torch.nn calls per forward from Python (src/networks/osvos_vgg.py:61-83) and
 * autograd replays them backward.  Here the layer loop itself is native: ONE call issues every kernel of
 * OSVOS_VGG.forward, ONE call every kernel of its backward, over a caller-provided arena (activations,
 * activation gradients, op workspaces; layout private to the library, size from fosvos_vgg_arena_bytes).
 * The structs hold device pointers only and live in HOST memory for the duration of the call.
 * Conv index c = 0..12 runs conv1_1 .. conv5_3; side index i = 0..3 hangs off stages 2..5.
 * replaces: OSVOS_VGG.forward and its autograd graph. */
typedef struct fosvos_vgg_weights {
    const float *conv_w[13];      /* fp32 OIHW masters (only conv_w[0] is read by a kernel; the rest via images) */
    const float *conv_b[13];      /* fp32 [Co] */
    const uint16_t *conv_wf[13];  /* packed forward images ([0] unused) */
    const uint16_t *conv_wd[13];  /* packed dgrad images  ([0] unused) */
    const float *side_b[4];
    const uint16_t *side_wf[4];
    const uint16_t *side_wd[4];
    const float *filt[4];         /* [k][k][16] per-channel deconv filters */
    const float *filt1[4];        /* [k][k] */
    const float *dsn_w;           /* [4][16] */
    const float *dsn_b;           /* [4] */
    const float *fuse_w;          /* [64] */
    const float *fuse_b;          /* [1] */
    int filt_uniform;             /* bit s: filt[s]'s 16 channel filters are identical (see fosvos_head_fwd) */
} fosvos_vgg_weights;

typedef struct fosvos_vgg_grads {
    float *conv_w[13];            /* fp32 OIHW */
    float *conv_b[13];
    float *side_w[4];             /* [16,C,3,3] */
    float *side_b[4];
    float *dsn_w;                 /* [4][16] or NULL when no side-output gradient flows */
    float *dsn_b;                 /* [4] or NULL */
    float *fuse_w;                /* [64] */
    float *fuse_b;                /* [1] */
    int accumulate;               /* != 0: add into the buffers (gradient accumulation), else overwrite */
    int defer_join;               /* != 0 with an aux_stream: do NOT make `stream` wait for the weight gradients at the
                                   * end of the call; the caller joins (stream waits on aux_stream) before it reads
                                   * them, and must not reuse this arena before that.  Lets the next forward pass
                                   * (other arena, same weights) overlap the tail of the weight-gradient kernels. */
    int bucket_events;            /* != 0: publish the gradients in four pieces as the pass finishes them (stage 5,
                                   * stage 4, stage 3, the rest) so that a data-parallel caller can start each piece's
                                   * all-reduce early: see fosvos_vgg_grad_bucket_wait. */
    int last_pass_of_cycle;       /* != 0: a hint - no forward pass follows this backward pass before the optimizer step
                                   * (the last one of an accumulation cycle), so the weight-gradient kernels of the first
                                   * two stages, which run after the data-gradient chain has ended, may take the whole
                                   * chip (256 pixel splits instead of the shared-chip count).  Results differ only in
                                   * the order of the fp32 sums over splits. */
} fosvos_vgg_grads;

size_t fosvos_vgg_arena_bytes(int N, int H, int W);
/* fused: [N,1,H,W] fp32; side_out: four [N,1,H,W] fp32 buffers or NULL.  The arena keeps what backward needs.
 * A batch (N >= 2) runs as two chains of ceil(N/2) and floor(N/2) frames - each layer is launched once per chain - so
 * that a frame's arithmetic (tile plan, K split) is the same whether or not a second stream is there to run the chains
 * side by side (fosvos_vgg_forward_streams). */
int fosvos_vgg_forward(const fosvos_vgg_weights *w, const float *frame, int N, int H, int W, void *arena,
                       size_t arena_bytes, float *fused, float *const side_out[4], int device, void *stream);
/* The same pass with a second stream of the same device (or NULL = fosvos_vgg_forward): the second chain of frames runs on
 * aux_stream beside the first (one frame: only its four side_prep convs, beside the next stage's backbone convs); `stream`
 * waits for it (events of `ctx`, whose device the call runs on) in front of the head, so the caller sees single-stream
 * semantics on `stream`.  Results are bit-identical to fosvos_vgg_forward. */
int fosvos_vgg_forward_streams(fosvos_ctx *ctx, const fosvos_vgg_weights *w, const float *frame, int N, int H, int W,
                               void *arena, size_t arena_bytes, float *fused, float *const side_out[4], void *stream,
                               void *aux_stream);
/* d_fused / d_side_out: upstream gradients ([N,1,H,W] fp32; d_fused or all four d_side_out may be NULL).
 * Must follow a fosvos_vgg_forward on the same arena, frame and shape; runs on ctx's device.
 * aux_stream (a second hipStream_t of the same device, or NULL): when given, every weight-gradient kernel is
 * issued on it while the data-gradient chain stays on `stream`; the two are ordered by events (a layer's wgrad
 * waits for that layer's output gradient; `stream` waits for the last wgrad before the call's work is complete
 * in stream order), so the caller sees ordinary single-stream semantics on `stream`.  The events are the context's:
 * the library itself keeps no state between calls. */
int fosvos_vgg_backward(fosvos_ctx *ctx, const fosvos_vgg_weights *w, const fosvos_vgg_grads *g, const float *frame, int N,
                        int H, int W, void *arena, size_t arena_bytes, const float *d_fused,
                        const float *const d_side_out[4], void *stream, void *aux_stream);
/* Make `stream` wait until gradient bucket `bucket` of the LAST fosvos_vgg_backward ON THIS CONTEXT (called with
 * grads.bucket_events != 0) is complete in its buffers: 0 = conv5_1..conv5_3 (stages.4), 1 = conv4_1..conv4_3
 * (stages.3), 2 = conv3_1..conv3_3 (stages.2) - 97 % of the gradient bytes, each published as the pass finishes it - then
 * 3 = conv1_1..conv2_2, 4 = side_prep / score_dsn / fuse (3 and 4 complete together, at the end of the pass, on both of
 * its streams).  Nothing blocks on the host.  The reference has no collective; this is the hook the
 * RCCL gradient all-reduce of BASELINE.json's north_star overlaps the backward pass with. */
int fosvos_vgg_grad_bucket_wait(fosvos_ctx *ctx, int bucket, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* FOSVOS_HIP_H */
