/*
 * rdo_ptq_hip.h -- C ABI of librdoptq_hip.so: the MI355X (gfx950) kernels behind the RDO-PTQ calibration hot path.
 *
 * The reference (Eric-qi/RDO-PTQ, /root/reference/task-oriented-PTQ) has no FFI: its hot path is Python calling ATen /
 * CompressAI ops.  Each entry point below names the reference call site it replaces.  All pointers are BORROWED device
 * pointers (hipMalloc'ed by the caller, e.g. torch tensors), `stream` is a hipStream_t passed as void*, nothing is
 * allocated or synchronised inside, no exceptions cross the boundary: every function returns 0 on success or a negative
 * RDO_E* code, with a human-readable message available from rdo_last_error() (thread-local).
 *
 * Layouts (fp32 everywhere):
 *   activations  NHWC  x[b][h][w][c]              (== torch channels_last of the reference's NCHW tensors)
 *   conv weights OHWI  w[co][kh][kw][ci]          (== torch channels_last of the reference's OIHW weights)
 *   dgrad weights      wd[ci][kh'][kw'][co] with taps flipped (kh' = KH-1-kh), produced by rdo_adaround_step / rdo_weight_dgrad_layout
 *   GDN gamma    [i][j] row-major (i = output channel), beta [i]
 */
#ifndef RDO_PTQ_HIP_H
#define RDO_PTQ_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RDO_OK 0
#define RDO_EINVAL (-22)   /* bad argument / unsupported shape */
#define RDO_EHIP (-5)      /* a HIP runtime call failed */
#define RDO_ENOMEM (-12)   /* workspace too small */

/* Per-iteration loss logs are [iters][RDO_LOG_SLOTS] floats: kernels spread their atomicAdd partials over the slots of the
 * current iteration (contended same-address float atomics serialise); the value of an iteration is the sum of its slots. */
#define RDO_LOG_SLOTS 32

/* epilogue / prologue selectors for the conv kernels */
enum {
    RDO_EPI_NONE = 0,       /* out = acc + bias */
    RDO_EPI_LRELU = 1,      /* out = leaky_relu(acc + bias, 0.01)                         quant_block.py:238,273,301,305 */
    RDO_EPI_LRELU_BWD = 2,  /* out = acc * (aux > 0 ? 1 : 0.01)        (dgrad through a LeakyReLU whose OUTPUT is aux)    */
    RDO_EPI_GDN = 3,        /* out = aux * rsqrt(acc + bias)                                      quant_layer.py:147-153 */
    RDO_EPI_IGDN = 4,       /* out = aux * sqrt(acc + bias)                                       quant_layer.py:147-153 */
    RDO_EPI_RELU = 5,       /* out = max(acc + bias, 0): nn.ReLU fused into the QuantModule          quant_model.py:51-54 */
    RDO_EPI_RELU_BWD = 6,   /* out = aux > 0 ? acc : 0                    (dgrad through a ReLU whose OUTPUT is aux)        */
    RDO_EPI_GELU = 7,       /* out = gelu(acc + bias), exact erf form: Mlp.act fused into fc1            models/layers.py:37-47 */
    RDO_EPI_GELU_BWD = 8    /* out = acc * gelu'(aux)       (dgrad through a GELU whose INPUT -- fc1's `pre` output -- is aux)    */
};

typedef struct rdo_conv_desc {
    int32_t B, H, W, Cin;        /* input  [B][H][W][Cin]   */
    int32_t Ho, Wo, Cout;        /* output [B][Ho][Wo][Cout] */
    int32_t KH, KW, stride, pad; /* square stride / symmetric zero padding, dilation 1, groups 1 */
    int32_t epilogue;            /* RDO_EPI_*  */
    int32_t square_input;        /* 1: the kernel squares x on load (GDN norm pool, quant_layer.py:147) */
    int32_t add_residual;        /* 1: out += residual (after the activation)        quant_block.py:245,279,310 */
} rdo_conv_desc;

const char* rdo_version(void);
const char* rdo_last_error(void);
/* Kernel-variant switches for A/B measurements and parity tests (process-wide; the defaults are what ships).  Keys:
 *   "wgrad_x6_w8"  1 (default): eight-wave bf16x6 weight-gradient kernel, 0: the four-wave one
 *   "conv_x6"      1 (default): large convolutions on the split-bf16 MFMA path, 0: everything on the fp32 MFMA
 *   "fwd_x6_ver"   forward bf16x6 kernel generation (default: newest)
 *   "xcd"          1 (default): XCD-aware tile numbering in the bf16x6 kernels
 *   "graph_unroll" iterations per replayed graph in rdo_plan_run for long runs (default 8; 1: one graph launch per iteration)
 *   "x6p_halo"     1 (default): 3x3 stride-1 pad-1 plane-input convs with H, W multiples of 16 run the halo-tile kernel (256 x 192 tiles,
 *                  or 256 x 64 tiles where the wide ones would need a K split); 2: wide tiles only; 0: the per-tap kernel
 *   "wgrad_p3_row" 1 (default): 3x3 stride-1 plane-input weight gradients share one input row image between the three kw taps
 *   "thin_mfma"    1 (default): weight gradients with <= 4 input channels and a patch of 5..32 values run the gather-operand MFMA kernel
 *   "h2_stagger"   1 (default): in the halo-tile plane-input conv waves 4-7 run two thirds of a K stage behind waves 0-3 (same results)
 *   "ada_w1_min"   batched AdaRound step: tensors with at least this many gradient slabs are walked one element per thread (4x the
 *                  threads on the slab chain) instead of four
 *   "tail_grid"    most workgroups of a fused loss kernel (each ends with one atomic add into the 32-slot loss log)
 *   "x6p_ablate"   ONLY in a diagnostic build (`make DIAG=1`, -DRDO_DIAG; the shipped library rejects a non-zero value and ignores
 *                  RDO_X6P_ABLATE): bit mask for the plane-input conv (results are WRONG when non-zero): 1 no A DMA, 2 no B DMA, 4 no MFMA,
 *                  8 no fragment reads; halo kernel only: 16 no epilogue, 32 the second weight plane is not read (30 % fewer
 *                  LDS->register bytes).  The same build exports the stamp reader used by tools/h2_stamps.py: shader-clock
 *                  cycles and 100 MHz wall ticks of the halo kernel's K loop per workgroup
 * Returns RDO_EINVAL for an unknown key.  rdo_get_tuning returns the current value (or -1). */
int rdo_set_tuning(const char* key, int32_t value);
int rdo_get_tuning(const char* key);

/* ---- K1/K2/K3: convolution as implicit GEMM on fp32 MFMA -------- replaces F.conv2d at quant_layer.py:123 (and the
 * 1x1 `F.conv2d(x**2, gamma, beta)` of f_gdn, quant_layer.py:147).  `pre` (nullable) receives acc+bias before the
 * epilogue (needed by the GDN backward).  dgrad of a stride-1 conv is this same entry point run on dY with `wd`. */
int rdo_conv2d_fwd(const rdo_conv_desc* d, const float* x, const float* w, const float* bias, const float* aux,
                   const float* residual, float* out, float* pre, float* workspace, int64_t workspace_floats,
                   const void* wplanes /* nullable: 3 x Cout*KH*KW*Cin bf16 from rdo_split_bf16x3_conv(w) */, void* stream);
/* Split factor rdo_conv2d_fwd uses for this shape (1: no K split); `has_planes`: the caller passes bf16 weight planes. */
int rdo_conv2d_fwd_ksplit(const rdo_conv_desc* d, int has_planes, int64_t workspace_floats);
/* Only the K-split accumulation of rdo_conv2d_fwd: partial sums [ksplit][B*Ho*Wo][Cout] stay in `workspace` (ksplit =
 * rdo_conv2d_fwd_ksplit(...) must be >= 2) for rdo_loss_act_bwd_splitk.  No bias, no epilogue. */
int rdo_conv2d_fwd_partials(const rdo_conv_desc* d, const float* x, const float* w, const void* wplanes /* nullable */,
                            float* workspace, int64_t workspace_floats, void* stream);
/* Large problems can run on the bf16 MFMA at fp32-level accuracy: every fp32 operand is split EXACTLY into three bf16
 * planes (x = x1 + x2 + x3) and the six significant cross products are accumulated in fp32 (dropped terms <= 3*2^-24 |x w|).
 * Activations are split in the kernel's loader; the caller supplies the weight planes.  rdo_conv2d_fwd takes this path when
 * `wplanes` is given and rdo_conv2d_fwd_uses_bf16x6(d) is 1; rdo_conv2d_fwd_bf16x6 forces it (Cin % 16 == 0). */
int rdo_conv2d_fwd_uses_bf16x6(const rdo_conv_desc* d);
/* Weight planes of the split-precision path, "fragment order": plane p holds element (co, kh, kw, ci) of w [Cout][KH][KW][Cin] at
 *   p * numel + ((((ci / 16) * KH + kh) * KW + kw) * Cout + co) * 16 + ci % 16        (Cin % 16 == 0; otherwise element order)
 * so that the Cout x 16 weight tile of one K stage is one contiguous run for the kernels' LDS-DMA (whole cache lines).
 * rdo_split_bf16x3 keeps the element order (generic helper, not a conv operand). */
int rdo_split_bf16x3_conv(const float* w, int32_t Cout, int32_t KH, int32_t KW, int32_t Cin, void* planes /* 3*numel bf16 */,
                          void* stream);
int rdo_split_bf16x3(const float* w, int64_t n, void* planes /* 3*n bf16 */, void* stream);
int rdo_conv2d_fwd_bf16x6(const rdo_conv_desc* d, const float* x, const void* wplanes, const float* bias, const float* aux,
                          const float* residual, float* out, float* pre, float* workspace, int64_t workspace_floats, void* stream);
int rdo_conv2d_fwd_bf16x6_ksplit(const rdo_conv_desc* d);   /* K split this path uses for a shape (0: shape too small) */
/* Small problems (few output tiles) are split over K into `workspace` (deterministic two-pass reduction).  Returns the
 * number of floats the kernel would like for this shape (0 = no split); with a smaller / NULL workspace it does not split. */
int64_t rdo_conv2d_fwd_workspace(const rdo_conv_desc* d);

/* weight gradient: dw[co][kh][kw][ci] = sum_m dy[m][co] * x[pix(m,kh,kw)][ci]  (autograd of quant_layer.py:123).
 * Split over `nsplit` pixel chunks into `slabs[nsplit][Cout*KH*KW*Cin]` (deterministic; reduced by rdo_adaround_step or
 * rdo_reduce_slabs).  rdo_conv2d_wgrad_nsplit() returns the split count the kernel wants for a shape. */
int rdo_conv2d_wgrad_nsplit(const rdo_conv_desc* d);
int rdo_conv2d_wgrad(const rdo_conv_desc* d, const float* x, const float* dy, float* slabs, int nsplit, void* stream);
int rdo_conv2d_wgrad_uses_bf16x6(const rdo_conv_desc* d);   /* 1: this shape runs on the split-bf16 MFMA path (fp32 accuracy) */
int rdo_reduce_slabs(const float* slabs, int nsplit, int64_t numel, float* out, void* stream);

/* ---- K4/K8/K10: AdaRound ---------------------------------------- quantizer.py:427-452, layer_opt.py:159-165,254,307 */
typedef struct rdo_ada_desc {
    int64_t numel;      /* weight elements, stored [rows][inner] with delta/zp per row (channel-wise dim 0) */
    int32_t rows;       /* Cout (conv) or C (GDN gamma rows); 1 for layer-wise scales */
    int32_t n_levels;   /* 2^bits */
    int32_t reparam;    /* 1: weight is a GDN gamma -> emit gamma' = max(wq, bound)^2 - pedestal (and chain the gradient) */
    float reparam_bound, reparam_pedestal;
    int32_t KH, KW, Cin; /* for the dgrad layout of conv weights (0,0,0: do not emit wd; for gamma wd = transpose) */
} rdo_ada_desc;

/* alpha0 = -log((zeta-gamma)/(frac(w/delta)-gamma) - 1)                                    quantizer.py:454-462 */
int rdo_adaround_init_alpha(const rdo_ada_desc* d, const float* w, const float* delta, float* alpha, void* stream);

/* soft (soft!=0) or hard rounding forward into the kernel layouts wq (same layout as w) and wd (nullable). */
int rdo_adaround_fwd(const rdo_ada_desc* d, const float* w, const float* alpha, const float* delta, const float* zp,
                     int soft, float* wq, float* wd, void* stream);

/* per-iteration schedule row (host-precomputed, device-resident): the unit executor indexes it with *iter_ptr */
typedef struct rdo_sched_row {
    float b;            /* temperature, LinearTempDecay utils.py:37-54 */
    float round_on;     /* 0 while count < warmup*iters (layer_opt.py:160-161), else 1 */
    float step_size;    /* lr / (1 - beta1^t)          torch.optim.Adam */
    float bc2_sqrt;     /* sqrt(1 - beta2^t) */
} rdo_sched_row;

/* Fused: reduce `nsplit` wgrad slabs (+ optional pre-reduced grad) -> dL/dwq -> chain through the soft quantiser (and the
 * GDN re-parametrisation) -> + round-loss gradient -> Adam(beta1 .9, beta2 .999, eps 1e-8) on alpha -> next soft wq / wd.
 * Accumulates weight*sum(1-|2h-1|^b) of the CURRENT alpha (before the update) into round_loss_out[*iter_ptr][slot] (atomicAdd).
 * `grad_scale` multiplies the data gradient (1/world_size after a sum all-reduce). */
int rdo_adaround_step(const rdo_ada_desc* d, const float* w, const float* delta, const float* zp, const float* slabs,
                      int nsplit, float grad_scale, float round_weight, const rdo_sched_row* sched, const int32_t* iter_ptr,
                      float* alpha, float* adam_m, float* adam_v, float* wq, float* wd, float* round_loss_out,
                      void* wq_planes, void* wd_planes, float wq_plane_scale, float wd_plane_scale, void* stream);
/* wq_planes / wd_planes (nullable): the planes of the new wq / wd in fragment order, written in the same pass so that the
 * split-precision conv paths need no separate refresh launch.  plane scale == 0: 3 x numel bf16, the exact three-way split (what
 * rdo_split_bf16x3_conv would produce for wq [rows][KH][KW][Cin] and for wd [Cin][KH][KW][rows]; operand of the fp32-input kernels
 * rdo_conv2d_fwd / rdo_conv2d_fwd_bf16x6); plane scale > 0 (a power of two): 2 x numel fp16, the two-way split of w * scale
 * (rdo_split_h2_conv; operand of rdo_conv2d_fwd_h2). */

/* rdo_adaround_step (mode 0), rdo_adaround_grad (mode 1: slabs -> dalpha) or rdo_adaround_apply (mode 2: dalpha -> update) for all
 * (<= 8) weight tensors of a unit in one launch (plus one launch for their dgrad layouts), numel % 4 == 0.
 * advance_iter (nullable): device iteration counter to increment once the step is done -- replaces a trailing rdo_iter_advance. */
typedef struct rdo_ada_step_item {
    rdo_ada_desc d;
    const float *w, *delta, *zp, *slabs /* modes 0, 1 */;
    int32_t nsplit;
    float *alpha, *adam_m, *adam_v, *wq, *wd /* nullable */;
    void *wq_planes /* nullable */, *wd_planes /* nullable */;
    float* dalpha;      /* modes 1 (out) and 2 (in): this tensor's slice of the data-parallel gradient bucket */
    float wq_plane_scale, wd_plane_scale;   /* 0: bf16 three-way planes, > 0: fp16 two-way planes of w * scale (see rdo_adaround_step) */
    /* nullable: fp16 two-way planes of the new soft weight [rows][inner] * lin_plane_scale in the fragment order of rdo_linear_h2
     * (what rdo_split_h2_linear would make of wq), and of its transpose (of wd: the input-gradient Linear) -- a Linear / GDN gamma
     * whose token matrices run on rdo_linear_h2 gets them from the step itself instead of two more launches.  rows, inner % 32 == 0. */
    void *lin_fwd_planes, *lin_bwd_planes;
    float lin_plane_scale;
} rdo_ada_step_item;
int rdo_adaround_step_batch(const rdo_ada_step_item* items, int32_t n, int32_t mode, float grad_scale, float round_weight,
                            const rdo_sched_row* sched, const int32_t* iter_ptr, float* round_loss_out, int32_t* advance_iter,
                            int32_t* iter_shadow /* nullable alternative to advance_iter: receives *iter_ptr + 1 (hand-over above) */,
                            void* stream);

/* Round 6: the step launch of iteration i also ASSEMBLES THE MINI-BATCH OF ITERATION i + 1 (gather + QDrop, rdo_gather_qdrop /
 * rdo_gather_qdrop_h2 below: same index arithmetic, counter RNG and plane split -- csrc/gather_body.h) in extra workgroups behind the
 * step's own: the step is latency-bound, the gather a plain stream, both run behind every reader of the current mini-batch, and the
 * iteration loses a launch (layer_opt.py:289-292 moved behind :307 of the previous pass; iteration 0's mini-batch is assembled by
 * a stand-alone gather before the loop).  The gather reads row *iter_ptr + 1 of idx_table and is skipped when that row does not exist
 * (n_iters).  Counter hand-over with this entry: the loss / tail launch of an iteration reads the real counter and publishes it into
 * a second word (rdo_iter_bind_publish); this launch is given iter_ptr = that word and iter_shadow = the real counter, into which its
 * first thread stores *iter_ptr + 1 -- no thread of the launch reads the word another thread of it writes. */
typedef struct rdo_gather_desc {
    const float *cache_q, *cache_fp;
    const int32_t* idx_table;           /* [n_iters][B] */
    int32_t n_iters, B, batch_offset;
    int64_t per_image;
    int32_t C;                          /* plane form only */
    float prob;
    uint32_t seed;
    float* out;                         /* fp32 mini-batch; nullable when out_planes is given */
    void* out_planes;                   /* nullable: H2 planes [2][B * per_image] of out * out_scale (rdo_gather_qdrop_h2) */
    float out_scale;
    int32_t* overflow_flag;             /* nullable: the overflow word pair of out_planes (NULL: the word bound by rdo_h2_bind_flag / the default) */
} rdo_gather_desc;
int rdo_adaround_step_batch_gather(const rdo_ada_step_item* items, int32_t n, int32_t mode /* 0 or 2 */, float grad_scale, float round_weight,
                                   const rdo_sched_row* sched, const int32_t* iter_ptr, float* round_loss_out, int32_t* iter_shadow,
                                   const rdo_gather_desc* next, void* stream);
/* The NEXT loss / tail launch issued (or recorded) on this thread -- rdo_lp2_loss_grad, rdo_lp_loss_grad, rdo_loss_act_bwd(_splitk),
 * rdo_loss_gdn_bwd, rdo_conv2d_fwd_h2_tail -- stores the iteration number it read to *publish (its first thread), then the binding is
 * gone.  Returns 1 when an earlier binding was still pending (never consumed), else 0; publish = NULL only clears. */
int rdo_iter_bind_publish(int32_t* publish);

/* ---- the data path of a 1 x 1 LAYER unit's iteration in ONE launch (round 6): pre = x W~^T + b (quant_layer.py:113-123), out = act(pre),
 * loss += coef * lp_loss(out, target[idx]) (layer_opt.py:133,150), dpre = act'(pre) dL/dout, and the weight-gradient slabs dpre^T x --
 * rdo_conv2d_fwd + rdo_loss_act_bwd + rdo_conv2d_wgrad for the 1x1 192 <-> 96 convs of Cheng2020-attn's attention blocks.  Two forms:
 * split-fp16 MFMA with scales taken on the fly (per output channel, per token, per tile of dL/dpre: nothing probed, nothing that can
 * overflow) where Cout is a multiple of 96, and exact fp32 arithmetic (fp32 MFMA) for every other shape or after rdo_unit1x1_form(0).
 * x [M][K], w [N][K]; tokens % 32 == 0, 32 <= K <= 192, K % 32 == 0, N % 32 == 0; slabs [nslab][N][K] with
 * nslab = rdo_unit1x1_nslab(M, N) (<= 256).  A loss / tail launch in the sense of rdo_iter_bind_publish. */
int rdo_unit1x1_supported(int64_t M, int32_t K, int32_t N);
int rdo_unit1x1_nslab(int64_t M, int32_t N);
/* 1 (default): the split-fp16 form where the shape allows it; 0: always the exact form.  Returns the previous setting.  Process-wide, and
 * rdo_unit1x1_nslab follows it: switch before a unit's slabs are sized. */
int rdo_unit1x1_form(int32_t form);
int rdo_unit1x1(const float* x, int64_t M, int32_t K, int32_t N, const float* w, const float* bias, const float* tgt_cache,
                const int32_t* idx_table, const int32_t* iter_ptr, int32_t B, float coef, int32_t act /* 0 none, 1 LeakyReLU(0.01), 2 ReLU */,
                float* slabs, int32_t nslab, float* loss_out, void* stream);

/* only the data-gradient half (slab reduce + chain rule -> dalpha_data), for the all-reduce bucket of the DP path */
int rdo_adaround_grad(const rdo_ada_desc* d, const float* w, const float* alpha, const float* delta, const float* zp,
                      const float* slabs, int nsplit, float* dalpha, void* stream);
/* ... and the second half applied to an (all-reduced) dalpha bucket */
int rdo_adaround_apply(const rdo_ada_desc* d, const float* w, const float* delta, const float* zp, const float* dalpha,
                       float grad_scale, float round_weight, const rdo_sched_row* sched, const int32_t* iter_ptr,
                       float* alpha, float* adam_m, float* adam_v, float* wq, float* wd, float* round_loss_out,
                       void* wq_planes, void* wd_planes, float wq_plane_scale, float wd_plane_scale, void* stream);

/* ---- K5: nearest fake-quant of a weight (UniformAffineQuantizer.forward)                  quantizer.py:175-177 */
int rdo_uaq_fakequant(const rdo_ada_desc* d, const float* w, const float* delta, const float* zp, float* wq, float* wd,
                      void* stream);
/* channel-wise 'max' init: delta[r] = max((max(x,0)-min(x,0))/(L-1), 1e-8), zp[r] = round(-min/delta) quantizer.py:281-298 */
int rdo_uaq_init_minmax(const float* w, int32_t rows, int64_t inner, int32_t n_levels, float* delta, float* zp,
                        void* stream);

/* ---- K6: dynamic per-channel activation quant-dequant (ActQuant)                          quantizer.py:81-117
 * zp = min_c, r = max(max_c - zp, 1e-6), out = round(clamp((x - zp)/r, -1, 1) * (2^n_bits - 1)) / (2^n_bits - 1) * r + zp.
 * The reference hard-wires n_bits = 8 (`b_w=8`, quantizer.py:81); other widths (BASELINE config "W10A10") are an extension. */
int rdo_actquant_perchannel(const float* x, int64_t npix, int32_t C, int32_t n_bits, float* out,
                            float* ws_minmax /* rdo_actquant_workspace(C) floats of scratch, no initial state needed */, void* stream);
int64_t rdo_actquant_workspace(int32_t C);   /* floats: the 2 C results + the per-workgroup partial minima / maxima */

/* ---- K7: mini-batch assembly: out[b] = keep ? cache_q[idx[b]] : cache_fp[idx[b]], keep ~ counter RNG(seed, iter, i)
 * replaces cached_inps[..][idx] + torch.where(torch.rand_like(x) < p, x_q, x_fp)                layer_opt.py:289-292
 * i = element index in the GLOBAL mini-batch: (batch_offset + b) * per_image + offset.  A data-parallel rank that holds rows
 * [batch_offset, batch_offset + B) of the global mini-batch draws exactly the mask slice a single process would (same seed). */
int rdo_gather_qdrop(const float* cache_q, const float* cache_fp, const int32_t* idx_table, const int32_t* iter_ptr,
                     int32_t B, int32_t batch_offset, int64_t per_image, float prob, uint32_t seed, float* out,
                     int32_t* iter_publish /* nullable, see "iteration-counter hand-over" */, void* stream);
/* Iteration-counter hand-over (saves the one-thread "counter += 1" launch per iteration): every kernel of an iteration reads the
 * device counter *iter_ptr; instead of incrementing it after the last kernel, the AdaRound step (the LAST kernel, which only reads
 * *iter_ptr) leaves *iter_ptr + 1 in a second word `iter_shadow`, and the gather (the FIRST kernel of the next iteration) is handed
 * iter_ptr = that shadow word and iter_publish = the real counter, which its first thread stores while every thread uses the value
 * read from the shadow.  No kernel reads a word that another thread of the same launch writes.  Both words start equal. */

/* ---- K8: lp_loss(pred, tgt[idx]) forward + gradient, p = 2: loss = sum((pred-tgt)^2)/(npix), sum over channels
 * grad = coef * 2 (pred - tgt) / npix ; `coef` = 2 reproduces rec_loss + (degenerate) task_loss of SURVEY 3.4.
 * Adds `coef * loss` into loss_out[*iter_ptr][slot]  (atomicAdd, RDO_LOG_SLOTS slots per iteration).                          quantizer.py:71-79, layer_opt.py:133,150 */
int rdo_lp2_loss_grad(const float* pred, const float* tgt_cache, const int32_t* idx_table, const int32_t* iter_ptr,
                      int32_t B, int64_t per_image, int32_t C, float coef, float* grad, float* loss_out, void* stream);

/* general task exponent (main2.py:52 --task_loss -> LossFunction.metric, layer_opt.py:150,274): adds
 *   coef2 * sum d^2 / npix + coefp * sum |d|^p / npix   to the loss slot and writes its gradient; p >= 1.
 * (coef2, coefp) = (1, 1) is rec_loss + task_loss on the same tensors, (0, 1) the task term alone.  `loss_out_p` (nullable):
 * when given, the |d|^p term is logged there and only the p = 2 term goes to loss_out (the reference's 500-iteration log
 * prints task and rec apart, layer_opt.py:168-170).                                                                           */
int rdo_lp_loss_grad(const float* pred, const float* tgt_cache, const int32_t* idx_table, const int32_t* iter_ptr,
                     int32_t B, int64_t per_image, int32_t C, float coef2, float coefp, float p, float* grad,
                     float* loss_out, float* loss_out_p, void* stream);

/* ---- K9: element-wise helpers on NHWC tensors */
int rdo_lrelu_fwd(const float* x, int64_t n, float* out, void* stream);                         /* nn.LeakyReLU(0.01) */
int rdo_lrelu_bwd(const float* g, const float* y, int64_t n, float* out, void* stream);          /* out = g*(y>0?1:.01) */
int rdo_relu_fwd(const float* x, int64_t n, float* out, void* stream);                          /* nn.ReLU             */
int rdo_relu_bwd(const float* g, const float* y, int64_t n, float* out, void* stream);          /* out = y>0 ? g : 0   */
int rdo_pixel_shuffle(const float* x, int32_t B, int32_t H, int32_t W, int32_t C, int32_t r, int32_t inverse, float* out,
                      void* stream);                                                     /* F.pixel_shuffle quant_layer.py:109 */
int rdo_add(const float* a, const float* b, int64_t n, float* out, void* stream);
/* GDN backward, element-wise halves (the two 1x1 GEMMs go through rdo_conv2d_fwd / rdo_conv2d_wgrad):
 *   t   = -1/2 g x n^-3/2 (GDN)  |  1/2 g x n^-1/2 (IGDN)
 *   dx  = g n^-1/2 + 2 x acc (GDN)  |  g n^1/2 + 2 x acc (IGDN)    with acc = t . gamma'                         */
int rdo_gdn_bwd_t(const float* g, const float* x, const float* norm, int64_t n, int32_t inverse, float* t, void* stream);
int rdo_gdn_bwd_dx(const float* g, const float* x, const float* norm, const float* acc, int64_t n, int32_t inverse,
                   float* dx, void* stream);
int rdo_nchw_to_nhwc(const float* x, int32_t B, int32_t C, int32_t H, int32_t W, int32_t inverse, float* out, void* stream);
int rdo_iter_advance(int32_t* iter_ptr, void* stream);
/* F.conv_transpose2d (quant_layer.py:30-37,123) = zero-insertion + padding of the input, then rdo_conv2d_fwd (stride 1, pad 0)
 * with the weight re-laid out [co][K-1-kh][K-1-kw][ci]:  out[b][pt + h*s][pl + w*s][c] = x[b][h][w][c], zeros elsewhere */
int rdo_zero_insert(const float* x, int32_t B, int32_t H, int32_t W, int32_t C, int32_t stride, int32_t pad_top, int32_t pad_left,
                    int32_t Ho, int32_t Wo, float* out, void* stream);
/* Transposed conv WITHOUT zero insertion (round 3): ConvTranspose2d(k, stride s, padding p) whose output is s x the input size
 * (output_padding = s + 2p - k) equals a stride-1 conv with s^2 * Cout output channels and a K' x K' window (padding K' / 2) followed
 * by F.pixel_shuffle(s): output pixel (s i + a, s j + b) meets only the taps kh = (a + p) mod s + s t, reading input rows
 * i + (a + p) / s - t.  rdo_tconv_expand builds that phase weight [Cout * s^2][K'][K'][Cin] from the kernel-layout weight
 * [Cout][taps][Cin] and a host-made `map` [s^2][K'^2] (tap index or -1 for a structural zero); rdo_tconv_fold gathers the gradient
 * slabs of the phase weight back into slabs of the kernel weight (`inv` [taps] = window entry of each tap).  k = 5, s = 2 (the deconv
 * of models/utils.py:107-116): 36 window entries for 25 taps, against 100 multiply-adds per input pixel through rdo_zero_insert. */
int rdo_tconv_expand(const float* w, const int32_t* map, int32_t Cout, int32_t taps, int32_t Cin, int32_t S2, int32_t KK2, float* wphase,
                     void* stream);
int rdo_tconv_fold(const float* slabs_phase, const int32_t* inv, int32_t nsplit, int32_t Cout, int32_t taps, int32_t Cin, int32_t S2,
                   int32_t KK2, float* slabs, void* stream);
/* F.layer_norm over the last dimension (quant_layer.py:44-49,121): rows x C, biased variance, eps inside the sqrt */
int rdo_layer_norm(const float* x, const float* weight, const float* bias, int64_t rows, int32_t C, float eps, float* out,
                   void* stream);

/* ---- Lu2022 transformer path: window attention core, LayerNorm backward, GELU ------------------------------
 * Replaces the tensor ops of WindowAttention / SwinTransformerBlock (models/layers.py:138-170, 260-303) and of their Quant*
 * wrappers (quantization/quant_block.py:384-417, 497-547).  Tensors are [B, H, W, *] fp32 in natural pixel order; the cyclic
 * shift and the window partition are address arithmetic inside the kernels.  qkv channels: which*C + head*(C/heads) + d. */
typedef struct rdo_attn_desc {
    int32_t B, H, W, C;          /* tokens B*H*W, model width C */
    int32_t heads;               /* C % heads == 0 */
    int32_t window;              /* window side; H % window == W % window == 0; window*window <= 64 */
    int32_t shift;               /* cyclic shift (0 or window/2): also switches the -100 region mask on (layers.py:236-257) */
    float scale;                 /* q multiplier, head_dim^-0.5 unless overridden (layers.py:107) */
} rdo_attn_desc;
/* bias: [heads][N][N] = relative_position_bias_table[relative_position_index] (layers.py:152-155), N = window^2.
 * out [B,H,W,C] = softmax(scale*q k^T + bias + mask) v.  probs (nullable): the probabilities as [windows][N][N][heads] for the
 * activation-quantised evaluation path (quant_block.py:410-411); out == NULL stops after writing probs. */
int rdo_window_attention_fwd(const rdo_attn_desc* d, const float* qkv, const float* bias, float* out, float* probs, void* stream);
/* out = probs @ v for probabilities produced (and possibly fake-quantised) outside */
int rdo_window_attention_pv(const rdo_attn_desc* d, const float* qkv, const float* probs, float* out, void* stream);
/* dqkv [B,H,W,3C] from dout [B,H,W,C]; probabilities are recomputed */
int rdo_window_attention_bwd(const rdo_attn_desc* d, const float* qkv, const float* bias, const float* dout, float* dqkv, void* stream);
/* LayerNorm backward over the last dimension (C <= 512).  dx (nullable) and, when dgamma_slabs != NULL, nslabs partial sums
 * [nslabs][C] of dy * xhat (consumed by rdo_adaround_step like the wgrad slabs).  gamma == NULL means weight 1. */
int rdo_layer_norm_bwd(const float* x, const float* gamma, const float* dy, int64_t rows, int32_t C, float eps, float* dx,
                       float* dgamma_slabs, int32_t nslabs, void* stream);
/* Residual add + LayerNorm in one pass (SwinTransformerBlock.forward, models/layers.py:260-300: `x = shortcut + attn(...)` followed by
 * `norm2(x)`, and the next block's `norm1` of `x + mlp(...)`): s = a (+ b, nullable) -> sum_out (nullable; needs b), out = LayerNorm(s).
 * C a multiple of 4, <= 512; pointers 16-byte aligned. */
int rdo_add_layer_norm(const float* a, const float* b, const float* weight, const float* bias, int64_t rows, int32_t C, float eps,
                       float* sum_out, float* out, void* stream);
/* rdo_layer_norm_bwd with the gradient that reaches x along the residual path folded in: dx = (add1) (+ add2) + LayerNorm-backward(dy)
 * (both addends nullable; add2 needs add1).  Replaces autograd's AddBackward accumulation behind every LayerNorm of a Swin block. */
int rdo_layer_norm_bwd_add(const float* x, const float* gamma, const float* dy, const float* add1, const float* add2, int64_t rows,
                           int32_t C, float eps, float* dx, float* dgamma_slabs, int32_t nslabs, void* stream);
int rdo_add3(const float* a, const float* b, const float* c, int64_t n, float* out, void* stream);   /* (a + b) + c: RSTB output, layers.py:433 */
/* Token-matrix Linear  out[M][N] = x[M][K] W^T + bias  (F.linear of the Swin blocks, models/layers.py:44-47,147,163; quant_layer.py:119;
 * its input gradient is the same entry on the planes of W^T) on fp16 two-way-split MFMA with a PER-TOKEN dynamic power-of-two scale
 * taken inside the kernel: no probed scale, no overflow flag (csrc/linear_h2.hip).  `wplanes`: rdo_split_h2_linear of W [N][K] * wscale
 * (2 * N * K halfs, fragment order).  rdo_linear_h2_supported: M % 64 == 0; K and N each a multiple of 192, or 96 (the 192 <-> 96 1x1 convs
 * of Cheng2020-attn's attention blocks as token matrices). */
int rdo_linear_h2_supported(int64_t M, int32_t K, int32_t N);
int rdo_split_h2_linear(const float* w, int32_t N, int32_t K, float scale, void* planes, void* stream);
int rdo_linear_h2(const float* x, int64_t M, int32_t K, int32_t N, const void* wplanes, float wscale, const float* bias,
                  int32_t square_input /* x enters squared: the GDN norm pool beta' + gamma' . x^2, quant_layer.py:147 */, float* out, void* stream);
/* ... with an epilogue: RDO_EPI_GELU (out = gelu(y), `pre` = y: Mlp.fc1 + nn.GELU(), models/layers.py:44-47) or RDO_EPI_GELU_BWD
 * (out = y * gelu'(aux): the input gradient of fc2 taken through that GELU, aux = fc1's `pre`). */
int rdo_linear_h2_epi(const float* x, int64_t M, int32_t K, int32_t N, const void* wplanes, float wscale, const float* bias,
                      int32_t square_input, int32_t epilogue, float* pre, const float* aux, float* out, void* stream);
int rdo_gelu_fwd(const float* x, int64_t n, float* out, void* stream);                      /* nn.GELU(): exact erf form */
int rdo_gelu_bwd(const float* dy, const float* x, int64_t n, float* dx, void* stream);
int rdo_round(const float* x, int64_t n, float* out, void* stream);                         /* torch.round (half to even): round_ste forward */

/* ---- K12: entropy-model likelihoods (eval rounding) and the rate / distortion sums -------------------------------------
 * Element-wise on [n] fp32 (NHWC: channel = i % C).  `params` = per channel 58 floats [33 softplus(matrix) | 13 bias |
 * 12 tanh(factor)] of CompressAI's EntropyBottleneck(filters=(3,3,3,3)); medians[C].       models/nic_cvt.py:297,300-308;
 * losses/losses.py:15-35; test_datasets.py:21-33 */
int rdo_factorized_likelihood_fwd(const float* z, const float* params, const float* medians, int64_t n, int32_t C, float* zhat,
                                  float* lik, void* stream);
/* gradient of grad_scale * sum(-log2 p) w.r.t. z^ (handed to z by the straight-through rounding): the R term of the opt-in
 * R + lambda*D task loss (losses/losses.py:15-35) through the factorised prior */
int rdo_factorized_likelihood_bwd(const float* zhat, const float* params, int64_t n, int32_t C, float grad_scale, float* dz, void* stream);
int rdo_gaussian_likelihood_fwd(const float* y, const float* scales, const float* means /* nullable */, int64_t n,
                                float scale_bound, float* yhat /* nullable */, float* lik, void* stream);
/* gradients of grad_scale * sum(-log2 p) w.r.t. scales / means (y^ constant: straight-through rounding) */
int rdo_gaussian_likelihood_bwd(const float* yhat, const float* scales, const float* means, int64_t n, float scale_bound,
                                float grad_scale, float* dscales, float* dmeans, void* stream);
int rdo_neg_log2_sum(const float* lik, int64_t n, float scale, float* out /* += */, void* stream);   /* bpp numerator */
int rdo_sq_diff_sum(const float* a, const float* b, int64_t n, float scale, int32_t clamp01_a, float* out /* += */,
                    void* stream);                                                                    /* MSE numerator */

/* ---- "H2" tensors and fused unit tails -------------------------------------------------------------------------------------------
 * An H2 tensor is an fp32 NHWC activation [M pixels][C channels] (C % 16 == 0) stored as the exact two-way fp16 split of its
 * power-of-two-scaled self:  x * s = h1 + h2 (+ <= 2^-24 |x s|),  h1 = fp16(x s), h2 = fp16(x s - h1), in "slice-major" planes
 * [2][C/16][M][16]:
 *     element (m, c) of plane p  at   p * M * C + ((c / 16) * M + m) * 16 + c % 16
 * The split-precision GEMM kernels read it by LDS-DMA -- a K stage is the 16 channels of one slice for a run of pixels, i.e.
 * contiguous 32-byte records -- so the conversion work is done once per element by the PRODUCER instead of once per use inside the K
 * loops, and a product of two such operands takes THREE fp16 MFMA products (h1 g1 + h1 g2 + h2 g1) at the accuracy of an fp32 fma
 * chain (tools/f16_probe.hip) where the exact three-way bf16 split of rounds 1-2 took six.  `s` is chosen per tensor by the caller (a
 * power of two; the engine takes it from a probe iteration so that max |x s| ~ 2^7; anything in [2^-2, 2^15] keeps fp32 accuracy, and
 * fp16 denormals are honoured by the MFMA).  A value with |x s| > 65504 cannot be stored: producers raise the sticky device flag read
 * by rdo_h2_overflow() -- the engine checks it after every run and fails loudly.
 * Producers: rdo_conv2d_fwd_h2 (epilogue), rdo_gather_qdrop_h2, rdo_loss_act_bwd, rdo_loss_gdn_bwd, rdo_gdn_bwd_dx_h2,
 * rdo_pixel_shuffle_h2, rdo_pixel_unshuffle2, rdo_split_h2 (from an fp32 tensor); weights: rdo_adaround_step* (plane scale > 0),
 * rdo_split_h2_conv. */
int rdo_h2_overflow(int reset);   /* 1: some producer met a value outside fp16 range since the last reset (synchronises); -1: error */
/* The flag word the H2 producers LAUNCHED (or recorded into a plan) BY THIS THREAD from now on raise: a caller-owned, zero-initialised
 * PAIR of int32 in device memory -- flag[0] holds the fp32 bit pattern of the largest FINITE |x s| that did not fit (atomic max), i.e. by
 * how much the scale was outgrown; flag[1] is set when an inf / NaN was met (everything downstream of a first overflow) --, so that every calibration unit (and every plane tensor of it) watches its own flag (one engine's overflow is not consumed by another, and a
 * data-parallel group can MAX-reduce the words before deciding); NULL re-binds the per-device default read by rdo_h2_overflow().  The
 * pointer is a kernel argument fixed at launch / record time: replays of a recorded plan keep raising the flag bound when it was
 * recorded.  The engine polls its word during long runs and restarts the unit with re-probed scales (or on fp32 activations) instead
 * of losing the run -- the reference's fp32 arithmetic (quant_layer.py:123) cannot overflow at all. */
int rdo_h2_bind_flag(int32_t* flag);
int rdo_split_h2(const float* x, int64_t npix, int32_t C, float scale, void* planes /* 2*npix*C fp16 */, void* stream);
/* conv weight [Cout][KH][KW][Cin] -> two fp16 planes of w * scale in fragment order (see rdo_split_bf16x3_conv for the order) */
int rdo_split_h2_conv(const float* w, int32_t Cout, int32_t KH, int32_t KW, int32_t Cin, float scale, void* planes /* 2*numel fp16 */,
                      void* stream);
/* 1: rdo_conv2d_fwd_h2 accepts this shape (a large problem of rdo_conv2d_fwd_uses_bf16x6 with Cin % 16 == 0, Cout % 16 == 0, no
 * square_input) */
int rdo_conv2d_fwd_h2_supported(const rdo_conv_desc* d);
int64_t rdo_conv2d_fwd_h2_workspace(const rdo_conv_desc* d);   /* floats of split-K scratch rdo_conv2d_fwd_h2 wants (0: none) */
/* rdo_conv2d_fwd (same epilogues, same results to fp32 accumulation order) with the activation given as H2 planes (scale x_scale) and
 * the weight as fragment-ordered fp16 planes (scale w_scale; rdo_split_h2_conv / rdo_adaround_step).  Any of out / pre / out_planes
 * may be NULL (at least one is required); out_planes receives the H2 form of `out` with scale out_scale. */
int rdo_conv2d_fwd_h2(const rdo_conv_desc* d, const void* x_planes, float x_scale, const void* wplanes, float w_scale, const float* bias,
                      const float* aux,
                      const void* aux_planes /* nullable: H2 form of aux, enough for RDO_EPI_LRELU_BWD / _RELU_BWD (sign only) */,
                      const float* residual, float* out, float* pre, void* out_planes, float out_scale, float* workspace,
                      int64_t workspace_floats, void* stream);
/* Last conv of a unit + its tail in ONE launch: rdo_conv2d_fwd_h2 followed by rdo_loss_act_bwd (residual as planes, dL/dpre as planes)
 * with the pre-activation never written: the halo kernel's epilogue forms out = act(conv + bias) + residual, the loss against
 * tgt_cache[idx] and dL/dpre.  3x3 / stride 1 / pad 1 shapes of the halo kernel only (rdo_conv2d_fwd_h2_tail_supported);
 * same bits as the two-call form (tests/test_gpu_h2.py). */
int rdo_conv2d_fwd_h2_tail_supported(const rdo_conv_desc* d);
int rdo_conv2d_fwd_h2_tail(const rdo_conv_desc* d, const void* x_planes, float x_scale, const void* wplanes, float w_scale,
                           const float* bias, const void* residual_planes /* nullable */, float residual_scale, const float* tgt_cache,
                           const int32_t* idx_table, const int32_t* iter_ptr, int32_t B, float coef, int32_t act, void* dpre_planes,
                           float dpre_scale, float* loss_out, void* stream);
/* rdo_conv2d_wgrad with both operands as H2 planes (x: [B*H*W][Cin], dy: [B*Ho*Wo][Cout]); same slabs, same nsplit rule
 * (rdo_conv2d_wgrad_nsplit).  Supported for the shapes of rdo_conv2d_wgrad_uses_bf16x6 with Cin % 16 == Cout % 16 == 0, no square_input. */
int rdo_conv2d_wgrad_h2_supported(const rdo_conv_desc* d);
/* ... and additionally k x k (k > 1) shapes with 64 <= Cin, Cout (multiples of 16) over >= 8192 output pixels -- the weight gradient of a
 * layer unit on planes (quant_layer.py:123 under autograd; the 3 x 3 96 -> 96 convs of Cheng2020-attn's attention blocks): the general
 * plane kernel with its 192 x 192 tile masked.  rdo_conv2d_wgrad_h2 accepts every shape one of the two predicates accepts. */
int rdo_conv2d_wgrad_h2_layer_supported(const rdo_conv_desc* d);
int rdo_conv2d_wgrad_h2(const rdo_conv_desc* d, const void* x_planes, float x_scale, const void* dy_planes, float dy_scale, float* slabs,
                        int nsplit, void* stream);
/* rdo_gather_qdrop writing the mini-batch as H2 planes (and as fp32 when `out` != NULL) */
int rdo_gather_qdrop_h2(const float* cache_q, const float* cache_fp, const int32_t* idx_table, const int32_t* iter_ptr, int32_t B,
                        int32_t batch_offset, int64_t per_image, int32_t C, float prob, uint32_t seed, float* out, void* out_planes,
                        float out_scale, int32_t* iter_publish, void* stream);
/* Tail of a unit whose last op is a conv (+ activation) (+ residual):   out = act(pre) + residual ; d = out - tgt[idx]
 *   loss_out[*iter][slot] += coef * sum d^2 / npix ; grad_out = coef * 2 d / npix ; dpre = grad_out * act'(pre)
 * i.e. the activation epilogue of the conv, rdo_lp2_loss_grad and rdo_lrelu_bwd / rdo_relu_bwd in one pass (layer_opt.py:133,150,
 * 303-306).  act: 0 none, 1 LeakyReLU(0.01), 2 ReLU.  out / grad_out / dpre / dpre_planes are optional outputs. */
int rdo_loss_act_bwd(const float* pre, const float* residual, const void* residual_planes /* nullable: the residual as H2 planes */,
                     float residual_scale, const float* tgt_cache, const int32_t* idx_table, const int32_t* iter_ptr, int32_t B,
                     int64_t per_image, int32_t C, float coef, int32_t act, float* out, float* grad_out, float* dpre, void* dpre_planes,
                     float dpre_scale, float* loss_out, void* stream);
/* The same tail taking the conv's output as the K-split partial sums rdo_conv2d_fwd_partials left in the workspace
 * ([ksplit][B*per_image]): pre = sum of the slabs in slab order + bias -- the conv's own second pass folded into the first load, so
 * conv epilogue, loss and activation backward are ONE pass and `pre` never exists in memory.  Bit-identical to rdo_conv2d_fwd
 * followed by rdo_loss_act_bwd. */
int rdo_loss_act_bwd_splitk(const float* partial, int32_t ksplit, const float* bias /* nullable */, const float* residual,
                            const float* tgt_cache, const int32_t* idx_table, const int32_t* iter_ptr, int32_t B, int64_t per_image,
                            int32_t C, float coef, int32_t act, float* out, float* grad_out, float* dpre, float* loss_out, void* stream);
/* Tail of a unit that ends in GDN / IGDN (+ residual):  out = x * norm^(-1/2 | +1/2) + residual ; loss and grad_out as above ;
 *   t = dL/dnorm = -1/2 g x norm^-3/2 (GDN) | 1/2 g x norm^-1/2 (IGDN)      = GDN epilogue + rdo_lp2_loss_grad + rdo_gdn_bwd_t */
int rdo_loss_gdn_bwd(const float* x, const float* norm, const float* residual, const float* tgt_cache, const int32_t* idx_table,
                     const int32_t* iter_ptr, int32_t B, int64_t per_image, int32_t C, float coef, int32_t inverse, float* out,
                     float* grad_out, float* t, void* t_planes, float t_scale, float* loss_out, void* stream);
int rdo_gdn_bwd_dx_h2(const float* g, const float* x, const float* norm, const float* acc, int64_t n, int32_t C, int32_t inverse,
                      float* dx, void* dx_planes, float dx_scale, void* stream);   /* rdo_gdn_bwd_dx with fp32 and / or H2 output */
/* F.pixel_shuffle(x, 2) on NHWC: [B,H,W,4C] -> [B,2H,2W,C] as fp32 and / or H2 planes */
int rdo_pixel_shuffle_h2(const float* x, int32_t B, int32_t H, int32_t W, int32_t C, float* out, void* out_planes, float out_scale,
                         void* stream);
/* its gradient: [B,2H,2W,C] -> [B,H,W,4C] (= rdo_pixel_shuffle(..., inverse = 1) for r = 2, 16-byte accesses on both sides), as fp32
 * and / or H2 planes */
int rdo_pixel_unshuffle2(const float* x, int32_t B, int32_t H, int32_t W, int32_t C, float* out, void* out_planes, float out_scale,
                         void* stream);

/* ---- unit executor: a recorded sequence of the calls above, replayed per calibration iteration with no host work.
 * Python records the per-iteration op list once per unit (layer_reconstruction / block_reconstruction, layer_opt.py:287-309);
 * rdo_plan_run() enqueues `n_iters` iterations, through a captured hipGraph when `use_graph` != 0. */
typedef struct rdo_plan rdo_plan;
rdo_plan* rdo_plan_create(void);
void rdo_plan_destroy(rdo_plan* p);
int rdo_plan_begin_record(rdo_plan* p);      /* subsequent rdo_* calls on this thread are recorded instead of launched */
int rdo_plan_end_record(rdo_plan* p);
/* suspend (on != 0) / resume recording on this thread: calls in between are launched, not recorded (one-time preparation of constants
 * found while a plan is being recorded); returns 1 when recording was already suspended, else 0 */
int rdo_plan_suspend_record(int on);
int rdo_plan_num_ops(const rdo_plan* p);
int rdo_plan_run(rdo_plan* p, int n_iters, int use_graph, void* stream);
/* capture + instantiate the graph(s) a later rdo_plan_run(p, n_iters, 1, .) replays, launching nothing (set-up work, like recording) */
int rdo_plan_prepare(rdo_plan* p, int n_iters);
/* one iteration of `p` then one of `q` as ONE graph launch (cached per partner): "apply of iteration i + forward/backward of iteration
 * i + 1" between two collectives of the data-parallel host loop */
int rdo_plan_run_then(rdo_plan* p, rdo_plan* q, int use_graph, void* stream);
/* measurement: kernel family tag + algorithmic FLOPs / bytes of op i; one eager iteration timed op by op with hipEvents
 * on `stream` (ms[num_ops]; synchronises the stream). */
int rdo_plan_op_info(const rdo_plan* p, int i, const char** tag, double* flops, double* bytes);
int rdo_plan_profile(rdo_plan* p, float* ms, void* stream);

/* ---- MS-SSIM pieces of the evaluation path (pytorch_msssim.ms_ssim as used by test_datasets.py:25-27, losses/losses.py:27,54).
 * Planes [planes][H][W] fp32 (planes = B*C of an NCHW image).  rdo_ssim_level: spatial means of the SSIM and contrast-structure
 * maps of one scale (11-tap separable window `window11`: HOST pointer to 11 floats; valid padding) -> ssim_mean[planes],
 * cs_mean[planes].  rdo_avg_pool2: F.avg_pool2d(x, 2, padding=(H%2, W%2)) -> [planes][(H+2(H%2)-2)/2+1][...]. */
int rdo_ssim_level(const float* x, const float* y, int32_t planes, int32_t H, int32_t W, const float* window11, float c1, float c2,
                   float* ssim_mean, float* cs_mean, void* stream);
int rdo_avg_pool2(const float* x, int32_t planes, int32_t H, int32_t W, float* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* RDO_PTQ_HIP_H */
