// Native layer loop of OSVOS-VGG: one C-ABI call issues every kernel of the forward pass, one call every
// kernel of the backward pass (reference: OSVOS_VGG.forward, src/networks/osvos_vgg.py:61-83, and the autograd
// graph torch builds from it).  Host-side only: this file launches the kernels of the other translation units
// through their exported entry points, over an arena whose layout is computed here.
//
// Why native: driven from Python, each of the ~100 ops of a training step costs ~19 us of interpreter, ctypes
// marshalling and allocator time (1.8 ms/step - as much as the GPU needs); from C++ a launch costs ~3 us.
#include <stdlib.h>

#include "common.hpp"

using namespace fosvos;

namespace {
constexpr int kNConv = 13;
constexpr int kStageOf[kNConv] = {0, 0, 1, 1, 2, 2, 2, 3, 3, 3, 4, 4, 4};
constexpr int kCin[kNConv] = {3, 64, 64, 128, 128, 256, 256, 256, 512, 512, 512, 512, 512};
constexpr int kCout[kNConv] = {64, 64, 128, 128, 256, 256, 256, 512, 512, 512, 512, 512, 512};
constexpr int kFirstOfStage[5] = {0, 2, 4, 7, 10};
constexpr int kLastOfStage[5] = {1, 3, 6, 9, 12};
constexpr int kStageCh[5] = {64, 128, 256, 512, 512};

inline size_t up256(size_t v) { return (v + 255) / 256 * 256; }

struct Arena {
    int sh[5], sw[5];        // stage resolutions
    size_t act[kNConv];      // conv outputs, bf16 NHWC
    size_t gact[kNConv];     // gradients wrt conv outputs (ReLU-masked), bf16 NHWC
    size_t pooled[4];        // pool outputs = inputs of stages 1..4
    size_t gpooled[4];       // gradients wrt them
    size_t side[4];          // fp32 NHWC [N,h,w,16]
    size_t dside[4];         // bf16 NHWC [N,h,w,32]
    size_t bits0;            // the ReLU mask of conv1_1's output as bits, [N,H,W,8] bytes (conv1_2's data gradient reads it)
    size_t ws, ws_bytes;     // op workspace of the main stream (split-K slabs)
    size_t hws, hws_bytes;   // head backward slabs (written from both streams, so never shared with `ws`)
    // per-layer slab workspaces of the wgrad (auxiliary) stream: every layer's slabs stay live until the batched
    // reduction at the end of the backward pass
    size_t wsa_conv[kNConv], wsa_conv_bytes[kNConv], wsa_side[4], wsa_side_bytes[4];
    size_t total;
};

Arena make_arena(int N, int H, int W) {
    Arena a{};
    a.sh[0] = H; a.sw[0] = W;
    for (int s = 1; s < 5; ++s) { a.sh[s] = (a.sh[s - 1] + 1) / 2; a.sw[s] = (a.sw[s - 1] + 1) / 2; }
    size_t off = 0;
    auto take = [&](size_t bytes) { const size_t o = off; off += up256(bytes); return o; };
    for (int c = 0; c < kNConv; ++c) {
        const int s = kStageOf[c];
        const size_t bytes = (size_t)N * a.sh[s] * a.sw[s] * kCout[c] * 2;
        a.act[c] = take(bytes);
        a.gact[c] = take(bytes);
    }
    for (int s = 1; s < 5; ++s) {
        const size_t bytes = (size_t)N * a.sh[s] * a.sw[s] * kStageCh[s - 1] * 2;
        a.pooled[s - 1] = take(bytes);
        a.gpooled[s - 1] = take(bytes);
        a.side[s - 1] = take((size_t)N * a.sh[s] * a.sw[s] * 16 * 4);
        a.dside[s - 1] = take((size_t)N * a.sh[s] * a.sw[s] * 32 * 2);
    }
    a.bits0 = take((size_t)N * H * W * (kCout[0] / 8));
    a.hws_bytes = up256(fosvos_head_bwd_workspace_bytes(N, H, W));
    a.hws = take(a.hws_bytes);
    size_t ws = 256;
    a.wsa_conv_bytes[0] = up256(fosvos_conv3x3_first_wgrad_workspace_bytes(N, H, W, 64));
    for (int c = 1; c < kNConv; ++c) {
        const int s = kStageOf[c];
        a.wsa_conv_bytes[c] = fosvos_conv3x3_wgrad_workspace_bytes(N, a.sh[s], a.sw[s], kCin[c], kCout[c]);
        // (in the forward pass the second chain of a split batch uses it as its split-K workspace)
        if (N >= 2)
            a.wsa_conv_bytes[c] = std::max(a.wsa_conv_bytes[c],
                                           fosvos_conv3x3_workspace_bytes(N - (N + 1) / 2, a.sh[s], a.sw[s], kCin[c], kCout[c]));
        a.wsa_conv_bytes[c] = up256(a.wsa_conv_bytes[c]);
        ws = std::max(ws, fosvos_conv3x3_workspace_bytes(N, a.sh[s], a.sw[s], kCin[c], kCout[c]));   // fwd split-K
        if (N >= 2)  // ... of the first chain of a split batch (fewer frames can mean MORE splits)
            ws = std::max(ws, fosvos_conv3x3_workspace_bytes((N + 1) / 2, a.sh[s], a.sw[s], kCin[c], kCout[c]));
        ws = std::max(ws, fosvos_conv3x3_workspace_bytes(N, a.sh[s], a.sw[s], kCout[c], kCin[c]));   // dgrad split-K
    }
    for (int s = 1; s < 5; ++s) {
        // (also the split-K workspace of the side layer's FORWARD conv when that runs on the auxiliary stream)
        a.wsa_side_bytes[s - 1] = up256(std::max({fosvos_conv3x3_wgrad_workspace_bytes(N, a.sh[s], a.sw[s], kStageCh[s], 16),
                                                  fosvos_conv3x3_workspace_bytes(N, a.sh[s], a.sw[s], kStageCh[s], 16),
                                                  N >= 2 ? fosvos_conv3x3_workspace_bytes(N - (N + 1) / 2, a.sh[s], a.sw[s], kStageCh[s], 16) : (size_t)0}));
        ws = std::max(ws, fosvos_conv3x3_workspace_bytes(N, a.sh[s], a.sw[s], kStageCh[s], 16));
        if (N >= 2) ws = std::max(ws, fosvos_conv3x3_workspace_bytes((N + 1) / 2, a.sh[s], a.sw[s], kStageCh[s], 16));
        ws = std::max(ws, fosvos_conv3x3_workspace_bytes(N, a.sh[s], a.sw[s], 16, kStageCh[s]));
    }
    a.ws_bytes = up256(ws);
    a.ws = take(a.ws_bytes);
    for (int c = 0; c < kNConv; ++c) a.wsa_conv[c] = take(a.wsa_conv_bytes[c]);
    for (int i = 0; i < 4; ++i) a.wsa_side[i] = take(a.wsa_side_bytes[i]);
    a.total = off;
    return a;
}

// ---- events of the caller's context (fosvos_ctx::vgg_ev): timing-disabled, for the two-stream passes
// 13 conv-output gradients + head + fork + join + d_side[0..2] (aux -> main) + gradient-bucket events: stage 5, 4, 3 (19..21)
// and "everything" (22) + "stage 2's weight gradients are queued" (23) + "the main stream's share of the tail is done" (24)
static_assert(kFosvosVggEvents == 25, "event slots below");
constexpr int kBucketFinalEvent = 22;  // every gradient the wgrad stream produces is final
constexpr int kStage2WgradEvent = 23;  // stage 2's weight-gradient kernels are queued on the wgrad stream
constexpr int kMainTailEvent = 24;     // the reductions offloaded to the main stream's end are done
constexpr int kBucketEvent0 = 19;

#define FOSVOS_TRY(expr)          \
    do {                          \
        const int rc_ = (expr);   \
        if (rc_ != FOSVOS_OK) return rc_; \
    } while (0)

int check_net(const void *w, const void *frame, const void *arena, int N, int H, int W, size_t arena_bytes,
              const Arena &a, const char *who) {
    FOSVOS_REQUIRE(w && frame && arena, FOSVOS_E_ARG, "%s: null pointer", who);
    FOSVOS_REQUIRE(N > 0 && H > 0 && W > 0, FOSVOS_E_SHAPE, "%s: bad shape N=%d H=%d W=%d", who, N, H, W);
    FOSVOS_REQUIRE(arena_bytes >= a.total, FOSVOS_E_WORKSPACE, "%s: arena %zu < %zu bytes", who, arena_bytes, a.total);
    FOSVOS_REQUIRE(((uintptr_t)arena & 255) == 0, FOSVOS_E_ARG, "%s: arena must be 256-byte aligned", who);
    return FOSVOS_OK;
}
}  // namespace

extern "C" size_t fosvos_vgg_arena_bytes(int N, int H, int W) {
    if (N <= 0 || H <= 0 || W <= 0) return 0;
    return make_arena(N, H, W).total;
}

// aux_stream (optional): the four side_prep convs (16 output channels: memory-bound, 120 us per five 480x854 frames) run on
// it, beside the backbone's MFMA-bound convs of the NEXT stage, instead of between them; `stream` waits for them in front of
// the head.  The caller sees single-stream semantics on `stream`.
namespace {
int forward_impl(hipEvent_t *ev, const fosvos_vgg_weights *w, const float *frame, int N, int H, int W, void *arena,
                 size_t arena_bytes, float *fused, float *const side_out[4], int device, void *stream, void *aux_stream) {
    const Arena a = make_arena(N, H, W);
    FOSVOS_TRY(check_net(w, frame, arena, N, H, W, arena_bytes, a, "vgg_forward"));
    FOSVOS_REQUIRE(fused, FOSVOS_E_ARG, "vgg_forward: null output");
    char *base = reinterpret_cast<char *>(arena);
    auto act = [&](int c) { return reinterpret_cast<uint16_t *>(base + a.act[c]); };
    void *ws = base + a.ws;
    hipStream_t sm = (hipStream_t)stream;
    const bool par = ev != nullptr && aux_stream != nullptr && aux_stream != stream;
    hipStream_t sa = par ? (hipStream_t)aux_stream : sm;
    if (par) FOSVOS_ENTER(device);
    // The frames of a batched pass run as TWO independent chains, the first ceil(N/2) on `stream`, the rest on the auxiliary
    // stream: the launches of one chain fill the partial last round of the other's (five 480x854 frames: conv3 is 4.1 rounds
    // of 512 workgroups, conv4 2.2) and what one chain's workgroups spend outside their chunk loops.  +1.0 % on the step.
    // Without an auxiliary stream the same two chains run one after the other (the arithmetic of a pass - tile plans, split-K -
    // does not depend on how it is scheduled).  FOSVOS_FWD_SPLIT=0 (lab switch): one chain of N frames.
    static const bool split_on = lab_env_int("FOSVOS_FWD_SPLIT", 1) != 0;
    if (split_on && N >= 2) {
        const int na = (N + 1) / 2;
        if (par) {
            FOSVOS_HIP_CHECK(hipEventRecord(ev[14], sm));
            FOSVOS_HIP_CHECK(hipStreamWaitEvent(sa, ev[14], 0));
        }
        for (int half = 0; half < 2; ++half) {
            const int f0 = half ? na : 0, nf = half ? N - na : na;
            hipStream_t st = half ? sa : sm;
            const uint16_t *xh = nullptr;
            for (int c = 0; c < kNConv; ++c) {
                const int s = kStageOf[c];
                const size_t px = (size_t)a.sh[s] * a.sw[s];
                uint16_t *yc = act(c) + (size_t)f0 * px * kCout[c];
                // the second chain's split-K workspace: the layer's own weight-gradient workspace (idle in the forward pass)
                void *wsc = half ? base + a.wsa_conv[c] : ws;
                const size_t wsn = half ? a.wsa_conv_bytes[c] : a.ws_bytes;
                if (c == 0) {
                    FOSVOS_TRY(fosvos_conv3x3_first_fwd_bits(frame + (size_t)f0 * 3 * H * W, w->conv_w[0], w->conv_b[0], yc,
                                                             reinterpret_cast<uint8_t *>(base + a.bits0) + (size_t)f0 * px * (kCout[0] / 8),
                                                             nf, H, W, kCout[0], device, st));
                } else {
                    if (c == kFirstOfStage[s]) {
                        const size_t ppx = (size_t)a.sh[s] * a.sw[s];
                        xh = reinterpret_cast<uint16_t *>(base + a.pooled[s - 1]) + (size_t)f0 * ppx * kCin[c];
                    }
                    FOSVOS_REQUIRE(wsn >= fosvos_conv3x3_workspace_bytes(nf, a.sh[s], a.sw[s], kCin[c], kCout[c]),
                                   FOSVOS_E_WORKSPACE, "vgg_forward: split-K workspace of the second chain");
                    if (c == kLastOfStage[s] && s < 4) {
                        const size_t qpx = (size_t)a.sh[s + 1] * a.sw[s + 1];
                        FOSVOS_TRY(fosvos_conv3x3_fwd_pool(xh, w->conv_wf[c], w->conv_b[c], yc,
                                                           reinterpret_cast<uint16_t *>(base + a.pooled[s]) + (size_t)f0 * qpx * kCout[c],
                                                           nf, a.sh[s], a.sw[s], kCin[c], kCout[c], FOSVOS_CONV_RELU, wsc, wsn,
                                                           device, st));
                    } else {
                        FOSVOS_TRY(fosvos_conv3x3_fwd(xh, w->conv_wf[c], w->conv_b[c], yc, nf, a.sh[s], a.sw[s], kCin[c],
                                                      kCout[c], FOSVOS_CONV_RELU, wsc, wsn, device, st));
                    }
                }
                xh = yc;
                if (s > 0 && c == kLastOfStage[s]) {
                    float *so = reinterpret_cast<float *>(base + a.side[s - 1]) + (size_t)f0 * px * 16;
                    FOSVOS_TRY(fosvos_conv3x3_fwd(xh, w->side_wf[s - 1], w->side_b[s - 1], so, nf, a.sh[s], a.sw[s], kStageCh[s],
                                                  16, FOSVOS_CONV_OUT_F32, half ? base + a.wsa_side[s - 1] : ws,
                                                  half ? a.wsa_side_bytes[s - 1] : a.ws_bytes, device, st));
                }
            }
        }
        // each chain runs the head for its own frames (the head is per frame), so the first chain's head runs beside the
        // second chain's last convs instead of behind them
        static const bool head_per_chain = lab_env_int("FOSVOS_HEAD_PER_CHAIN", 1) != 0;
        if (head_per_chain) {
            for (int half = 0; half < 2; ++half) {
                const int f0 = half ? na : 0, nf = half ? N - na : na;
                const float *sideh[4];
                float *soh[4] = {nullptr, nullptr, nullptr, nullptr};
                int hsh[4], wsh[4];
                for (int i = 0; i < 4; ++i) {
                    hsh[i] = a.sh[i + 1];
                    wsh[i] = a.sw[i + 1];
                    sideh[i] = reinterpret_cast<const float *>(base + a.side[i]) + (size_t)f0 * hsh[i] * wsh[i] * 16;
                    if (side_out && side_out[i]) soh[i] = side_out[i] + (size_t)f0 * H * W;
                }
                FOSVOS_TRY(fosvos_head_fwd(sideh, hsh, wsh, w->filt, w->filt1, w->dsn_w, w->dsn_b, w->fuse_w, w->fuse_b,
                                           fused + (size_t)f0 * H * W, (side_out && side_out[0]) ? soh : nullptr, nf, H, W,
                                           w->filt_uniform, device, half ? sa : sm));
            }
        }
        if (par) {
            FOSVOS_HIP_CHECK(hipEventRecord(ev[13], sa));
            FOSVOS_HIP_CHECK(hipStreamWaitEvent(sm, ev[13], 0));
        }
        if (head_per_chain) return FOSVOS_OK;
    } else {
    const uint16_t *x = nullptr;
    for (int c = 0; c < kNConv; ++c) {
        const int s = kStageOf[c];
        if (c == 0) {
            FOSVOS_TRY(fosvos_conv3x3_first_fwd_bits(frame, w->conv_w[0], w->conv_b[0], act(0),
                                                     reinterpret_cast<uint8_t *>(base + a.bits0), N, H, W, kCout[0], device, stream));
        } else {
            if (c == kFirstOfStage[s])  // stage entry: the pooled map the previous stage's last conv wrote
                x = reinterpret_cast<uint16_t *>(base + a.pooled[s - 1]);
            if (c == kLastOfStage[s] && s < 4) {  // the pool that feeds the next stage rides in this conv's epilogue
                FOSVOS_TRY(fosvos_conv3x3_fwd_pool(x, w->conv_wf[c], w->conv_b[c], act(c),
                                                   reinterpret_cast<uint16_t *>(base + a.pooled[s]), N, a.sh[s], a.sw[s],
                                                   kCin[c], kCout[c], FOSVOS_CONV_RELU, ws, a.ws_bytes, device, stream));
            } else {
                FOSVOS_TRY(fosvos_conv3x3_fwd(x, w->conv_wf[c], w->conv_b[c], act(c), N, a.sh[s], a.sw[s], kCin[c], kCout[c],
                                              FOSVOS_CONV_RELU, ws, a.ws_bytes, device, stream));
            }
        }
        x = act(c);
        if (s > 0 && c == kLastOfStage[s]) {
            // the last stage's side conv has nothing left to run beside: it stays on `stream` (the round trip through the
            // other stream cost 25 us in front of the head)
            const bool on_aux = par && s < 4;
            if (on_aux) {  // the stage output is complete on `stream`: the side conv may read it on the other one
                FOSVOS_HIP_CHECK(hipEventRecord(ev[c], sm));
                FOSVOS_HIP_CHECK(hipStreamWaitEvent(sa, ev[c], 0));
            }
            // (on the auxiliary stream the layer's own weight-gradient workspace doubles as its split-K workspace: `ws`
            // is in use by the backbone convs running beside it)
            FOSVOS_TRY(fosvos_conv3x3_fwd(x, w->side_wf[s - 1], w->side_b[s - 1], base + a.side[s - 1], N, a.sh[s], a.sw[s],
                                          kStageCh[s], 16, FOSVOS_CONV_OUT_F32, on_aux ? base + a.wsa_side[s - 1] : ws,
                                          on_aux ? a.wsa_side_bytes[s - 1] : a.ws_bytes, device, on_aux ? sa : sm));
        }
    }
    if (par) {  // join: the head reads the four side maps
        FOSVOS_HIP_CHECK(hipEventRecord(ev[13], sa));
        FOSVOS_HIP_CHECK(hipStreamWaitEvent(sm, ev[13], 0));
    }
    }
    const float *side[4];
    int hs[4], wsz[4];
    for (int i = 0; i < 4; ++i) {
        side[i] = reinterpret_cast<const float *>(base + a.side[i]);
        hs[i] = a.sh[i + 1];
        wsz[i] = a.sw[i + 1];
    }
    return fosvos_head_fwd(side, hs, wsz, w->filt, w->filt1, w->dsn_w, w->dsn_b, w->fuse_w, w->fuse_b, fused, side_out, N, H,
                           W, w->filt_uniform, device, stream);
}
}  // namespace

extern "C" int fosvos_vgg_forward_streams(fosvos_ctx *ctx, const fosvos_vgg_weights *w, const float *frame, int N, int H,
                                          int W, void *arena, size_t arena_bytes, float *fused, float *const side_out[4],
                                          void *stream, void *aux_stream) {
    FOSVOS_TRY(ctx_check(ctx, "vgg_forward_streams"));
    return forward_impl(ctx->vgg_ev, w, frame, N, H, W, arena, arena_bytes, fused, side_out, ctx->device, stream, aux_stream);
}

extern "C" int fosvos_vgg_forward(const fosvos_vgg_weights *w, const float *frame, int N, int H, int W, void *arena,
                                  size_t arena_bytes, float *fused, float *const side_out[4], int device,
                                  void *stream) {
    return forward_impl(nullptr, w, frame, N, H, W, arena, arena_bytes, fused, side_out, device, stream, nullptr);
}

extern "C" int fosvos_vgg_backward(fosvos_ctx *ctx, const fosvos_vgg_weights *w, const fosvos_vgg_grads *g,
                                   const float *frame, int N, int H, int W, void *arena, size_t arena_bytes,
                                   const float *d_fused, const float *const d_side_out[4], void *stream,
                                   void *aux_stream) {
    FOSVOS_TRY(ctx_check(ctx, "vgg_backward"));
    const int device = ctx->device;
    const Arena a = make_arena(N, H, W);
    FOSVOS_TRY(check_net(w, frame, arena, N, H, W, arena_bytes, a, "vgg_backward"));
    FOSVOS_REQUIRE(g, FOSVOS_E_ARG, "vgg_backward: null gradient table");
    const bool with_so = d_side_out && d_side_out[0];
    FOSVOS_REQUIRE(d_fused || with_so, FOSVOS_E_ARG, "vgg_backward: no upstream gradient given");
    char *base = reinterpret_cast<char *>(arena);
    auto act = [&](int c) { return reinterpret_cast<uint16_t *>(base + a.act[c]); };
    auto gact = [&](int c) { return reinterpret_cast<uint16_t *>(base + a.gact[c]); };
    void *ws = base + a.ws;
    const int acc = g->accumulate ? 1 : 0;
    WgradReduceTable reduce;  // every layer queues its slab reduction; two launches at the end run them all
    reduce.n = 0;
    WgradReduceTable reduce_m;  // ... the reductions the MAIN stream runs behind its last data-gradient kernel (see `offload`)
    reduce_m.n = 0;

    // ---- two streams: data-gradient chain on `stream`, weight gradients on `aux_stream`
    hipStream_t sm = (hipStream_t)stream;
    const bool par = aux_stream != nullptr && aux_stream != stream;
    hipStream_t sa = par ? (hipStream_t)aux_stream : sm;
    hipEvent_t *ev = ctx->vgg_ev;
    const bool buckets = g->bucket_events != 0;
    // the weight gradients of stages 1-2 come last; after the cycle's LAST backward pass nothing runs beside them
    const bool tail = g->last_pass_of_cycle != 0;
    static const int tail_stages = lab_env_int("FOSVOS_TAIL_STAGES", 1);  // lab switch
    // In the cycle's last pass the data-gradient stream runs dry while the weight-gradient stream still owes stages 2-1
    // (timeline: ~300 us with the chip a quarter full).  Two pieces of that tail need nothing the wgrad stream has not
    // long finished, and move to the main stream's end: conv1_1's weight gradient (its operands are the main stream's own
    // last outputs) and the slab reduction of stage 2 (behind an event the wgrad stream records after stage 2's kernels).
    static const bool offload_on = lab_env_int("FOSVOS_TAIL_OFFLOAD", 1) != 0;
    const bool offload = tail && aux_stream != nullptr && aux_stream != stream && offload_on;
    FOSVOS_ENTER(device);
    ctx->buckets_recorded = false;
    if (par) {
        // fork: the wgrad stream may not run ahead of what `stream` has queued (the forward pass, the loss)
        FOSVOS_HIP_CHECK(hipEventRecord(ev[14], sm));
        FOSVOS_HIP_CHECK(hipStreamWaitEvent(sa, ev[14], 0));
    }
    // "tensor e is complete on the main stream": lets the wgrad stream consume it
    auto publish = [&](int e) -> int {
        if (!par) return FOSVOS_OK;
        FOSVOS_HIP_CHECK(hipEventRecord(ev[e], sm));
        FOSVOS_HIP_CHECK(hipStreamWaitEvent(sa, ev[e], 0));
        return FOSVOS_OK;
    };

    // ---- head: d_side[i], fuse / score_dsn gradients
    const float *side[4];
    uint16_t *dside[4];
    int hs[4], wsz[4];
    for (int i = 0; i < 4; ++i) {
        side[i] = reinterpret_cast<const float *>(base + a.side[i]);
        dside[i] = reinterpret_cast<uint16_t *>(base + a.dside[i]);
        hs[i] = a.sh[i + 1];
        wsz[i] = a.sw[i + 1];
    }
    const HeadBwdArgs ha{side, hs, wsz, w->filt, with_so ? w->filt1 : nullptr, with_so ? w->dsn_w : nullptr, w->fuse_w,
                         d_fused, with_so ? d_side_out : nullptr, dside, g->fuse_w, g->fuse_b,
                         with_so ? g->dsn_w : nullptr, with_so ? g->dsn_b : nullptr, N, H, W, acc, base + a.hws,
                         a.hws_bytes, device, w->filt_uniform & 15};
    FOSVOS_TRY(head_bwd_check(ha));
    // Only d_side[3] is needed right away (stage 5 runs first): it stays on the data-gradient stream.  The three
    // larger scales and the fuse / score_dsn weight gradients go to the wgrad stream; the data-gradient stream picks
    // d_side[s] up (event 16+s) when it reaches side_prep[s].
    FOSVOS_TRY(head_bwd_scale(ha, 3, sm));
    for (int i = 2; i >= 0; --i) {
        FOSVOS_TRY(head_bwd_scale(ha, i, sa));
        if (par) FOSVOS_HIP_CHECK(hipEventRecord(ev[16 + i], sa));
    }
    FOSVOS_TRY(publish(13));  // d_side[3] and its slabs are complete: the finish pass and side_prep[3]'s wgrad may run
    // (the finish pass - five blocks, 50-130 us - stays here, at the head of the weight-gradient stream, where the small
    // kernels between the passes cover it: moved to the end of the main stream it delays the next cycle's first kernel,
    // -2.3 % on the step, profiles/r04_lab_step_ab_head_loss.txt)
    FOSVOS_TRY(head_bwd_finish(ha, sa));

    // ---- stages 4..0
    static const bool unpool_fused = lab_env_int("FOSVOS_UNPOOL", 1) != 0;  // lab switch: 0 = pool backward as its own pass
    for (int s = 4; s >= 0; --s) {
        const int hh = a.sh[s], ww = a.sw[s], last = kLastOfStage[s], first = kFirstOfStage[s];
        if (s > 0) {
            // side_prep[s-1]: wgrad from (stage output, d_side) on the wgrad stream; its dgrad into the stage-output
            // gradient: ReLU-masked and added to what came back through the next stage's pool (already in gact[last])
            FOSVOS_TRY(wgrad_impl(act(last), dside[s - 1], g->side_w[s - 1], g->side_b[s - 1], N, hh, ww, kStageCh[s], 16,
                                  acc, base + a.wsa_side[s - 1], a.wsa_side_bytes[s - 1], device, sa,
                                  (offload && s == 1) ? &reduce_m : &reduce, tail && s <= tail_stages));
            const uint16_t *addend = (s < 4) ? gact(last) : nullptr;
            if (par && s < 4) FOSVOS_HIP_CHECK(hipStreamWaitEvent(sm, ev[16 + s - 1], 0));  // d_side[s-1] from the wgrad stream
            if (s < 4 && unpool_fused)  // ... the pool backward in the same pass (no pool pass ran for this stage: see below)
                FOSVOS_TRY(fosvos_conv3x3_dgrad_unpool(dside[s - 1], w->side_wd[s - 1], act(last),
                                                       reinterpret_cast<const uint16_t *>(base + a.gpooled[s]), gact(last), N, hh,
                                                       ww, kStageCh[s], 16, ws, a.ws_bytes, device, sm));
            else
                FOSVOS_TRY(fosvos_conv3x3_dgrad(dside[s - 1], w->side_wd[s - 1], act(last), addend, gact(last), N, hh, ww,
                                                kStageCh[s], 16, ws, a.ws_bytes, device, sm));
        }
        FOSVOS_TRY(publish(last));  // gradient wrt the stage output is complete
        for (int c = last; c >= first; --c) {
            if (c == 0) {
                if (offload)  // gact(0) is the main stream's own last output: no event, and it runs beside conv1_2's
                    FOSVOS_TRY(first_wgrad_impl(frame, gact(0), g->conv_w[0], g->conv_b[0], N, H, W, kCout[0], acc,
                                                base + a.wsa_conv[0], a.wsa_conv_bytes[0], device, sm, &reduce_m));
                else
                    FOSVOS_TRY(first_wgrad_impl(frame, gact(0), g->conv_w[0], g->conv_b[0], N, H, W, kCout[0], acc,
                                                base + a.wsa_conv[0], a.wsa_conv_bytes[0], device, sa, &reduce));
                break;
            }
            const bool from_pool = (c == first) && s > 0;  // its input is the pool output; conv 1 reads conv 0
            const uint16_t *xin = from_pool ? reinterpret_cast<const uint16_t *>(base + a.pooled[s - 1]) : act(c - 1);
            // dgrad: into the previous conv's output gradient (masked by its ReLU), or into the pool-output gradient
            // (unmasked: the pool backward applies the producer's mask)
            uint16_t *dx = from_pool ? reinterpret_cast<uint16_t *>(base + a.gpooled[s - 1]) : gact(c - 1);
            const uint16_t *mask = from_pool ? nullptr : act(c - 1);
            // In a 3-conv stage the middle conv's gradient gets no event of its own (an event costs ~3 us of main-stream
            // time): its dgrad is issued first and the event behind it covers both gact(c) and gact(c-1).
            const bool middle = (last - first == 2) && c == last - 1;
            if (!middle)
                FOSVOS_TRY(wgrad_impl(xin, gact(c), g->conv_w[c], g->conv_b[c], N, hh, ww, kCin[c], kCout[c], acc,
                                      base + a.wsa_conv[c], a.wsa_conv_bytes[c], device, sa,
                                      (offload && s == 1) ? &reduce_m : &reduce, tail && s <= tail_stages));
            static const bool bits_on = lab_env_int("FOSVOS_RELU_BITS", 1) != 0;  // lab switch: 0 = read conv1_1's output itself
            if (c == 1 && bits_on)  // conv1_2: its input's ReLU mask as the bits conv1_1's forward wrote (8 B instead of 128 B a pixel)
                FOSVOS_TRY(fosvos_conv3x3_dgrad_bits(gact(c), w->conv_wd[c], reinterpret_cast<const uint8_t *>(base + a.bits0), nullptr,
                                                     dx, N, hh, ww, kCin[c], kCout[c], ws, a.ws_bytes, device, sm));
            else
                FOSVOS_TRY(fosvos_conv3x3_dgrad(gact(c), w->conv_wd[c], mask, nullptr, dx, N, hh, ww, kCin[c], kCout[c], ws,
                                                a.ws_bytes, device, sm));
            const bool skip_event = (last - first == 2) && c == last;  // gact(last-1): covered by the middle conv's event
            if (!from_pool && !skip_event) FOSVOS_TRY(publish(c - 1));
            if (middle)
                FOSVOS_TRY(wgrad_impl(xin, gact(c), g->conv_w[c], g->conv_b[c], N, hh, ww, kCin[c], kCout[c], acc,
                                      base + a.wsa_conv[c], a.wsa_conv_bytes[c], device, sa,
                                      (offload && s == 1) ? &reduce_m : &reduce, tail && s <= tail_stages));
        }
        // pool backward into the previous stage's output gradient, its ReLU mask fused - for stage 1 only (it has no
        // side_prep): the outputs of stages 2-4 get theirs inside side_prep's data gradient, next iteration
        if (s == 1 || (s > 0 && !unpool_fused))
            FOSVOS_TRY(fosvos_maxpool2x2_ceil_bwd(act(kLastOfStage[s - 1]), reinterpret_cast<const uint16_t *>(base + a.gpooled[s - 1]),
                                                  gact(kLastOfStage[s - 1]), N, a.sh[s - 1], a.sw[s - 1], kStageCh[s - 1], 1,
                                                  device, sm));
        if (s >= 1) {
            // The slab reductions queued so far run now (two launches) instead of all at the end: the weight-gradient
            // stream has slack, and what stands between the last data-gradient kernel and the optimizer step shrinks to
            // stage 1's own reduction.  Data-parallel step: stages 5, 4, 3 (buckets 0-2) - 97 % of the gradient
            // bytes - are published here, so their all-reduce runs under the rest of the backward pass.
            FOSVOS_TRY(wgrad_reduce_all(&reduce, device, sa));
            if (buckets && s >= 2) FOSVOS_HIP_CHECK(hipEventRecord(ev[kBucketEvent0 + (4 - s)], sa));
            if (offload && s == 1) FOSVOS_HIP_CHECK(hipEventRecord(ev[kStage2WgradEvent], sa));
        }
    }
    FOSVOS_TRY(wgrad_reduce_all(&reduce, device, sa));  // slabs -> dw / db for all (remaining) layers
    if (offload) {  // stage 2's and conv1_1's reductions: on the main stream, beside conv1_2's weight-gradient kernel
        FOSVOS_HIP_CHECK(hipStreamWaitEvent(sm, ev[kStage2WgradEvent], 0));
        FOSVOS_TRY(wgrad_reduce_all(&reduce_m, device, sm));
    }
    if (buckets) FOSVOS_HIP_CHECK(hipEventRecord(ev[kMainTailEvent], sm));  // (also covers the head's main-stream share)
    if (buckets) {
        FOSVOS_HIP_CHECK(hipEventRecord(ev[kBucketFinalEvent], sa));  // buckets 3 and 4: everything the wgrad stream owes
        ctx->buckets_recorded = true;
    }
    if (par && !g->defer_join) {  // join: everything after this call on `stream` sees the weight gradients
        FOSVOS_HIP_CHECK(hipEventRecord(ev[15], sa));
        FOSVOS_HIP_CHECK(hipStreamWaitEvent(sm, ev[15], 0));
    }
    return FOSVOS_OK;
}

extern "C" int fosvos_vgg_grad_bucket_wait(fosvos_ctx *ctx, int bucket, void *stream) {
    FOSVOS_TRY(ctx_check(ctx, "vgg_grad_bucket_wait"));
    FOSVOS_REQUIRE(bucket >= 0 && bucket < 5, FOSVOS_E_ARG, "vgg_grad_bucket_wait: bucket %d not in 0..4", bucket);
    FOSVOS_ENTER(ctx->device);
    FOSVOS_REQUIRE(ctx->buckets_recorded, FOSVOS_E_ARG,
                   "vgg_grad_bucket_wait: the last fosvos_vgg_backward on this context did not set grads.bucket_events");
    hipEvent_t *ev = ctx->vgg_ev;
    if (bucket < 3) {
        FOSVOS_HIP_CHECK(hipStreamWaitEvent((hipStream_t)stream, ev[kBucketEvent0 + bucket], 0));
    } else {  // the tail buckets: the wgrad stream's last reduction AND what the pass left at the end of its main stream
        FOSVOS_HIP_CHECK(hipStreamWaitEvent((hipStream_t)stream, ev[kBucketFinalEvent], 0));
        FOSVOS_HIP_CHECK(hipStreamWaitEvent((hipStream_t)stream, ev[kMainTailEvent], 0));
    }
    return FOSVOS_OK;
}
