/* groove_hip.h -- C ABI of libgroove_hip.so: the MI355X (gfx950) hot path of the GrooveTransformer
 * train / predict step.
 *
 * The reference has no FFI: its hot path sits behind a *Python module API* supplied by the
 * un-vendored submodule `BaseGrooveTransformers` (ref:train.py:12 `initialize_model, calculate_loss,
 * train_loop`; ref:evaluator.py:173 `model.predict`).  Each entry point below names the reference
 * interface it replaces; INTEGRATION.md shows the ctypes stub a maintainer of the reference adds.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller (torch caching allocator); the library
 *     never allocates, frees or synchronises; every call enqueues on the given hipStream_t and is
 *     hipGraph-capturable (no host reads of device data, no host-dependent launch parameters
 *     except the gt_config);
 *   - return value 0 = ok, <0 = error; gt_last_error() gives the message (thread-local);
 *   - activations are row-major (M, features), M = batch*32 rows, row m = b*32 + t -- the layout
 *     the reference's DataLoader hands over ((B,32,S) / (B,32,27) contiguous fp32,
 *     ref:dataset.py:263-264,355-356);
 *   - HVO tensors are (M,27) = [hits(9) | velocities(9) | offsets(9)] (ref:utils.py:38-47,
 *     ref:evaluator.py:173-177);
 *   - parameters live in ONE flat fp32 buffer in state-dict order (names = the demo checkpoint's keys,
 *     ref:demo/transformer_run_171tyqit_Epoch_1.Model); gt_param_layout() gives each tensor's offset.
 *     Gradients use the same layout in a second flat buffer (one RCCL all-reduce, one fused update).
 */
#ifndef GROOVE_HIP_H
#define GROOVE_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* gt_stream_t;   /* a hipStream_t (ihipStream_t*), passed opaquely so C callers need no HIP headers */

#define GT_T 32          /* max_len, hard-coded by the reference: ref:train.py:128 */
#define GT_TGT 27        /* embedding_size_tgt: ref:train.py:132 */
#define GT_VOICES 9
#define GT_MAX_D 512     /* largest d_model in the reference's sweeps: ref:configs/InfillingClosedHH_sweep.yaml */

/* params["model"] of ref:train.py:115-134 */
typedef struct gt_config {
  int32_t batch;          /* sequences in this call (B) */
  int32_t src_dim;        /* embedding_size_src: 16 (MSO) or 27 (symbolic), ref:train.py:129-131 */
  int32_t d_model;
  int32_t n_heads;
  int32_t dim_ff;
  int32_t n_enc_layers;
  int32_t n_dec_layers;   /* 0 = encoder_only (ref:train.py:125-127) */
  float dropout;
  int32_t precision;      /* 0 = fp32 everywhere (the parity path).  1 = BASELINE configs[4]: the operands of every Linear's
                           * forward / dgrad / wgrad GEMM are rounded to bf16 (round-to-nearest-even) on their way into the
                           * matrix cores (v_mfma_f32_16x16x32_bf16, fp32 accumulate); master weights, activations in HBM,
                           * attention core, LayerNorm, softmax, loss and optimizer stay fp32.  Bias gradients are column sums
                           * of the bf16-rounded output gradient.
                           * 2 = bf16 where the bytes are (round 5): on top of 1, the Linear OUTPUTS of the encoder layers that
                           * torch.autocast(bfloat16) hands on as bf16 are stored in bf16 alone -- qkv (the attention kernels then do
                           * fp32 arithmetic on bf16-stored q / k / v), the out-proj / linear2 outputs ahead of their LayerNorm, the
                           * dgrad outputs ahead of a LayerNorm backward, dctx.  Residual stream, LayerNorm statistics, softmax, loss,
                           * master weights and optimizer stay fp32.  In force where the bf16 operand shadows apply at level 2 and
                           * the heads are 64 / 128 wide (gt_precision_in_force); elsewhere it runs as precision 1. */
  int32_t flags;          /* per-CALLER schedule switches (round 6; 0 = the library's defaults).  They select among schedules that give the same
                           * numbers, never change the workspace layout, and -- unlike gt_set_seq_quad / gt_set_ln_exchange, which are
                           * process-wide -- bind only the calls made with this gt_config: two engines in one process can differ.
                           * GT_CFG_NO_QUAD: no four-workgroups-per-sequence schedule (no in-launch pair exchange);
                           * GT_CFG_NO_LN_XCHG: no LayerNorm inside the Linears (no in-launch row exchange). */
} gt_config;
#define GT_CFG_NO_QUAD 1
#define GT_CFG_NO_LN_XCHG 2

/* device-resident per-step state, so that a captured hipGraph replays with fresh dropout masks,
 * Adam bias correction and learning rate without host-side kernel-argument changes */
typedef struct gt_step_state {
  uint32_t seed_lo, seed_hi;   /* dropout RNG seed */
  uint32_t step;               /* dropout stream id; incremented by gt_optimizer_step (a host may overwrite it) */
  uint32_t opt_step;           /* optimizer updates applied so far (Adam t-1); incremented by gt_optimizer_step */
  float lr;                    /* ref:train.py:136 learning_rate */
  float grad_scale;            /* 1/world_size for data-parallel averaging, else 1 */
  float beta1, beta2, eps;     /* Adam (torch defaults 0.9 / 0.999 / 1e-8) */
  float pad2[3];               /* [0],[1]: ticket words of the loss / optimizer reductions (must start as 0) */
} gt_step_state;

/* ---- dropout RNG (shared with oracle/numpy_groove.py) ------------------------------------------
 * fmix32 = murmur3 finaliser.  key(site) = fmix32((fmix32((seed_lo ^ fmix32(step)) ^ site*0x9E3779B9)
 * ^ seed_hi) + 0x7F4A7C15);  r(idx) = fmix32((idx*0x9E3779B1) ^ key);  keep iff (r>>8) >= p*2^24;
 * kept values are scaled by 1/(1-p).  idx = flat element index of the dropped tensor
 * ((M,d) / (M,F): m*cols+c; attention probabilities: ((b*H+h)*32+i)*32+j).  Sites: */
#define GT_SITE_PE_ENC 0
#define GT_SITE_PE_DEC 1
#define GT_SITE_LAYER0 16   /* site = 16 + 8*global_layer + kind; decoder layers follow the encoder's */
#define GT_SITE_ATTN 0      /* self-attention probabilities */
#define GT_SITE_DROP1 1     /* dropout1 after self-attn out-proj */
#define GT_SITE_FFN 2       /* dropout inside the FFN */
#define GT_SITE_DROPF 3     /* dropout on the FFN output (encoder dropout2 / decoder dropout3) */
#define GT_SITE_XATTN 4     /* decoder cross-attention probabilities */
#define GT_SITE_DROP2 5     /* decoder dropout2 after cross-attn out-proj */

const char* gt_last_error(void);
int gt_version(void);

/* Parameter layout in the flat buffer.  n_tensors / n_floats may be NULL.
 * Replaces: model.state_dict() ordering of the reference's nn.Module (ckpt key order). */
int gt_param_count(const gt_config* cfg, int64_t* n_tensors, int64_t* n_floats);
/* offsets[i], sizes[i] in floats; rows[i], cols[i] the 2-D shape (cols = 0 for vectors). */
int gt_param_layout(const gt_config* cfg, int64_t* offsets, int64_t* sizes, int32_t* rows, int32_t* cols);

/* Bytes of scratch the forward/backward of this config needs (saved activations + temporaries). */
size_t gt_workspace_bytes(const gt_config* cfg);
/* Call ONCE on a freshly allocated workspace, before its first use (stream-ordered; the library itself never allocates): zeroes the
 * pair-exchange region of the four-workgroups-per-sequence forward (gt_set_seq_quad), whose granules are zero between launches by
 * protocol.  A no-op for configs without such a region.  No counterpart in the reference (torch owns its activations there). */
int gt_workspace_init(const gt_config* cfg, float* ws, gt_stream_t stream);
/* Test/debug: locate a named saved activation inside the workspace (offset & count in floats).
 * names: "x0","a0","qkv","P","ctx","xhat1","rstd1","x1","hact","xhat2","rstd2","x2","memory",
 * "enc_xhat","dlogits" ... ; layer = global layer index (decoder layers follow the encoder's). */
int gt_ws_find(const gt_config* cfg, const char* name, int layer, int64_t* offset, int64_t* count);

/* Replaces GrooveTransformer(Encoder).forward(src[, tgt]) (ref:train.py:195-215 via train_loop;
 * module tree pinned by the demo checkpoint).  x (M,src_dim); tgt_in (M,27) teacher-forcing input
 * or NULL when n_dec_layers==0; pe (32,d_model) the registered positional-encoding buffer;
 * hvo_out (M,27) = [h logits | sigmoid v | 0.5 tanh o].  state==NULL or train==0 -> eval mode
 * (no dropout).  Saved activations for gt_backward are left in ws. */
int gt_forward(const gt_config* cfg, const float* params, const float* pe, const float* x,
               const float* tgt_in, float* hvo_out, float* ws, const gt_step_state* state, int train,
               gt_stream_t stream);

/* Replaces calculate_loss(pred, y, bce_fn, mse_fn, hit_loss_penalty) (ref:train.py:201-203,213;
 * bce/mse: ref:train.py:176-179; penalty: ref:train.py:55-58).  stats (8 floats, zeroed by the
 * call): [0] loss, [1] hit accuracy, [2] unused (perplexity = exp(stats[3]) on the host), [3] bce,
 * [4] mse_v, [5] mse_o.  d_hvo (M,27) or NULL: d loss / d (h,v,o). */
int gt_loss(const gt_config* cfg, const float* hvo, const float* y, float hit_loss_penalty,
            float* stats, float* d_hvo, gt_stream_t stream);

/* Replaces loss.backward() for the same module.  d_hvo (M,27) = grad w.r.t. forward's outputs.
 * grads: flat buffer, same layout as params; zeroed first unless accumulate != 0. */
int gt_backward(const gt_config* cfg, const float* params, float* grads, const float* x,
                const float* tgt_in, const float* hvo, const float* d_hvo, float* ws,
                const gt_step_state* state, int train, int accumulate, gt_stream_t stream);

/* Replaces optimizer.step() of torch.optim.SGD(lr, momentum=0) (ckpt: optimizer param_groups) /
 * torch.optim.Adam(lr) (ref:train.py:40-42).  algo 0 = sgd, 1 = adam (m, v: flat moment buffers).
 * Reads lr/grad_scale/betas from *state and increments state->step / opt_step.  zero_grads != 0 also
 * zeroes the consumed gradient buffer (what opt.zero_grad() does before the next batch).
 * PLAIN: every one of the n elements is updated, for any n (a sub-range, an unpadded buffer); no error word and no guard
 * element are looked at -- the fail-safe of the in-launch exchanges belongs to gt_train_step / gt_optimizer_step_ws, which
 * know the configuration and its workspace. */
int gt_optimizer_step(int algo, float* params, float* grads, float* m, float* v, int64_t n,
                      gt_step_state* state, int zero_grads, gt_stream_t stream);

/* gt_optimizer_step for the data-parallel step sequence (gt_train_step(skip_update = 1..3), all-reduce, update): with the
 * configuration and its workspace at hand the update on the sequence-resident path also writes the next step's fragment-ordered
 * weight copies (see GT_STEP_PACKS_CURRENT below).  The buffers are the WHOLE flat buffers of the configuration (gt_param_count
 * n_floats elements).  This entry point -- like the update inside gt_train_step -- carries the fail-safe of the in-launch exchanges:
 * with the workspace's error word set ("xchg_err"), or with a non-zero GUARD element grads[n_floats - 1] (padding behind the 27-float
 * output bias; gt_dp_guard writes it, the gradient all-reduce sums it), NOTHING is applied: parameters and moments stay, consumed
 * gradients are cleared (zero_grads), state->step advances (the batch was consumed: fresh dropout masks) but state->opt_step (Adam's t)
 * does not, and word 1 of the "xchg_err" header counts the skipped update.  The guard element itself is neither updated nor cleared. */
int gt_optimizer_step_ws(const gt_config* cfg, int algo, float* params, float* grads, float* m, float* v, float* ws,
                         gt_step_state* state, int zero_grads, gt_stream_t stream);

/* Gradient buckets for overlapping the data-parallel all-reduce with backward (no reference counterpart: the
 * reference is single-device).  Writes up to two [offset, offset+count) ranges of the flat gradient buffer in the order
 * backward completes them and returns their number: 2 when the model splits (encoder-decoder: decoder half first;
 * encoder-only with >= 2 layers: upper half of the layers first; sequence-resident path: only while the weight gradients ride in
 * the SPLIT backward phases -- gt_set_seq_ride -- where everything from encoder layer L - p + 1 on is final after phase p), else 1
 * (the whole buffer). */
int gt_grad_buckets(const gt_config* cfg, int64_t* offsets, int64_t* counts);

/* One whole train step of train_loop's batch body (ref:train.py:195-215): [shift y for the decoder]
 * forward, loss, backward, optimizer update.  tgt_scratch (M,27) is only used when n_dec_layers>0.
 * skip_update: 0 = the whole step; 1 = no optimizer update (data-parallel: all-reduce grads, then gt_optimizer_step);
 * 2 = as 1 but backward stops as soon as bucket 0 of gt_grad_buckets() is final; 3 = ONLY the rest of that backward
 * (same buffers, directly after a skip_update=2 call) -- the caller all-reduces bucket 0 while 3 runs.
 * PRECONDITION: grads is all zeros on entry (zero it once after allocation; the update this call -- or the
 * caller's gt_optimizer_step(zero_grads=1) -- leaves it zeroed again, so no per-step memset is enqueued).
 * skip_update | GT_STEP_PACKS_CURRENT: on the sequence-resident path (gt_step_launches() > 0) a whole step (skip_update 0) ends with an
 * update that also writes the NEXT step's fragment-ordered weight copies into ws; a caller that knows the previous call on this ws was
 * such a step and that nobody has written params since may set this bit, and the packing launch at the head of the step is skipped.
 * Ignored on the other paths. */
#define GT_STEP_PACKS_CURRENT 4
int gt_train_step(const gt_config* cfg, int algo, float* params, float* grads, float* m, float* v,
                  const float* pe, const float* x, const float* y, float hit_loss_penalty,
                  float* hvo_out, float* stats, float* tgt_scratch, float* ws, gt_step_state* state,
                  int skip_update, gt_stream_t stream);

/* Replaces model.predict(src, use_thres=True, thres=0.5) (ref:evaluator.py:173-177): eval forward,
 * h = sigmoid(logit) > thres ? 1 : 0 (or the probability when use_thres == 0); the encoder-decoder
 * runs the 32-step greedy decode.  hvo_out (M,27) is the concatenated HVO the evaluator builds. */
int gt_predict(const gt_config* cfg, const float* params, const float* pe, const float* x,
               float* hvo_out, float thres, int use_thres, float* tgt_scratch, float* ws,
               gt_stream_t stream);
/* model.predict(src, use_pd=True): as gt_predict, but the hits are SAMPLED from the predicted probabilities: h = 1 iff p > u with
 * u = (fmix32((idx * 0x9E3779B1) ^ seed) >> 8) * 2^-24, idx = (row * 27 + voice column) of hvo_out (the hash of the dropout masks,
 * shared with oracle/numpy_groove.py).  The encoder-decoder's greedy decode feeds the sampled hits back. */
int gt_predict_pd(const gt_config* cfg, const float* params, const float* pe, const float* x, float* hvo_out, uint32_t seed,
                  float* tgt_scratch, float* ws, gt_stream_t stream);
/* The same for a CHUNK of a larger evaluation set (ref:evaluator.py:173 hands the whole set over at once; the host walks it in
 * workspace-sized chunks): first_seq = index of x's first sequence inside the set; idx above counts from the start of the SET, so the
 * samples do not depend on the chunk size.  gt_predict_pd == gt_predict_pd_at(first_seq = 0). */
int gt_predict_pd_at(const gt_config* cfg, const float* params, const float* pe, const float* x, float* hvo_out, uint32_t seed,
                     int64_t first_seq, float* tgt_scratch, float* ws, gt_stream_t stream);

/* Replaces, for the device side, what the reference's evaluator computes from model.predict's output per epoch
 * (ref:evaluator.py:522-525: get_hits_accuracies / get_velocity_errors / get_micro_timing_errors over the 9 voices of
 * ROLAND_REDUCED_MAPPING): hvo_pred / hvo_gt are (n_rows,27) HVO tensors (n_rows = sequences * 32).  out30:
 * [0] hit accuracy over all voices, [1..9] per voice; [10] velocity MSE, [11..19] per voice; [20] offset MSE, [21..29] per
 * voice.  scratch: gt_voice_metrics_scratch_floats(n_rows) floats.  Fixed summation order: bitwise reproducible. */
int64_t gt_voice_metrics_scratch_floats(int64_t n_rows);
int gt_voice_metrics(const float* hvo_pred, const float* hvo_gt, int64_t n_rows, float* out30, float* scratch,
                     gt_stream_t stream);

/* Replaces GrooveMidiDatasetInfilling.__getitem__ + the DataLoader's collate for a dataset resident in HBM
 * (ref:dataset.py:263-264,355-356; ref:train.py:156-158): x[b] = xs[idx[b]], y[b] = ys[idx[b]] for b < batch, both step
 * inputs in ONE launch.  xs (n_seq,32,src_dim), ys (n_seq,32,27) fp32; idx int64 on the device (out-of-range entries are
 * clamped into the dataset). */
int gt_gather_batch(const float* xs, const float* ys, const int64_t* idx, int64_t n_seq, int32_t batch, int32_t src_dim,
                    float* x, float* y, gt_stream_t stream);

/* Measurement aid (no reference counterpart): with profiling on, every kernel launch is bracketed by
 * HIP events on its own stream.  gt_profile_report synchronises and writes one text row per kernel
 * class: "label launches total_ms total_flops total_bytes".  Not graph-capturable while on. */
int gt_profile_enable(int on);
/* on: grouped weight-gradient dispatches run on an internal side stream (created once, on first use) and may overlap
 * the backward chain; recorded as fork/join edges when the call is being captured into a hipGraph.  off (default;
 * env GT_OVERLAP=1 switches the default): a single stream -- measured faster on ROCm 7.2, see DESIGN.md. */
int gt_set_overlap(int on);
/* Sequence-resident kernels (csrc/gt_seq.h): for encoder-only fp32 models with d_model <= 128 (% 16), dim_feedforward <= 512
 * (% 16), src_dim <= 32 and head_dim 16 / 32 / 64 or below 16, ONE workgroup per sequence runs the whole forward (and one the
 * whole backward) in a single launch -- or, in the SPLIT mode below, two workgroups per sequence and one launch per layer and
 * direction; gt_step_launches() gives the launch count of a train step.  Default (neither this call nor env GT_SEQ): d_model <= 64
 * and d_model == 128 always, the other widths of the 128 class from batch 64 up.  on = 1 forces them wherever supported, on = 0
 * switches them off (env GT_SEQ=1 / GT_SEQ=0 do the same).  Same results as the one-kernel-per-op path to fp32 rounding. */
int gt_set_seq(int on);
/* Two workgroups per sequence (16 token rows each) and one launch per layer and direction for the sequence-resident kernels at
 * d_model 128 or 32: -1 = default (when 2 x batch workgroups fit the CUs once; d_model 32 only with dim_feedforward >= 256), 0 = off,
 * 1 = on (env GT_SEQ_SPLIT=0/1 does the same).  Same results bit for bit: the split is over token rows. */
int gt_set_seq_split(int on);
/* FOUR workgroups per sequence in the forward of the SPLIT mode at d_model 128 (csrc/gt_seq.h, QUAD): the two workgroups of a 16-row
 * half are column partners that each compute half of dim_feedforward and swap their partial FFN2 results through one in-launch pair
 * exchange per layer (8-byte tagged granules, agent-scope stores / loads) -- the forward then fills 4 x batch CUs instead of 2 x batch.
 * -1 = default (whenever 4 x batch workgroups fit the CUs at once and dim_feedforward % 32 == 0), 0 = off, 1 = on under the same
 * condition (env GT_SEQ_QUAD=0/1).  Results agree with the SPLIT mode to fp32 rounding (the FFN2 contraction is summed as two halves);
 * bitwise repeatable run to run.  Needs gt_workspace_init on the workspace. */
int gt_set_seq_quad(int on);
/* The pair exchange polls with a bound: a partner workgroup that never arrives (the two were not resident at the same time: another
 * stream's kernel, a CU mask, another process holds CUs) makes the waiting workgroup give up, raise the error word at the head of the
 * "seq_xchg" workspace region (gt_ws_find) and go on with garbage partials.  Fail-safe (round 5): with the word set -- it stays set until
 * the host zeroes the region with gt_workspace_init -- the fused update (gt_train_step, gt_optimizer_step_ws) leaves parameters and
 * optimizer moments untouched and only clears the consumed gradients; the same when gradient element n_floats - 1 (padding behind the
 * 27-float output bias; a data-parallel host writes its error flag there before the all-reduce) is non-zero, so every rank skips
 * together.  gt_set_xchg_spin_max: polls before giving up (<= 0: the default, 2^22 = seconds; tests lower it). */
int gt_set_xchg_spin_max(int polls);
/* "xchg_err" (gt_ws_find) = two 32-bit words at the head of whichever exchange region a shape has: [0] the error word, [1] the number of
 * updates skipped because of it (or of a data-parallel guard) since the last gt_workspace_init.
 * Data-parallel hosts: grads[n_floats - 1] = 1 if this workspace's exchange error word is set, else 0 -- enqueue between the backward
 * (gt_train_step(skip_update = 1 / 2)) and the gradient all-reduce; a no-op for shapes without an exchange region. */
int gt_dp_guard(const gt_config* cfg, float* grads, const float* ws, gt_stream_t stream);
/* LayerNorm inside the producing Linear / dgrad (csrc/gt_gemm64.h, round 5): at d_model 256 / 512, where the 64 x 64-tile kernels apply and
 * the whole grid is resident at once (a GPU's share of a data-parallel batch: 2048 tokens), the N / 64 workgroups of a row block exchange
 * their row partials inside the launch (tagged 8-byte granules, agent-scope stores / polling loads; "rowx" workspace region, zeroed once
 * by gt_workspace_init) and each normalises its own tile -- no row pass of its own.  -1 = default (on where it applies: 3/4 .. 2 tiles of 64 x 64 per CU), 0 = off (env GT_LN_XCHG=0): the norm as a
 * row pass of its own; 1-1.5 % of the step at 2048 tokens (csrc/groove_hip.hip, ln_xchg).  Same
 * bounded-spin / error-word / skipped-update contract as the pair exchange above ("xchg_err" names the word of whichever a shape has). */
int gt_set_ln_exchange(int on);
/* Test aid: nblocks workgroups that each pin 96 KB of LDS (no sequence workgroup fits beside one) for `usec` microseconds -- a second
 * stream holding CUs while a four-workgroups-per-sequence launch is in flight. */
int gt_debug_occupy_cus(int nblocks, int usec, gt_stream_t stream);
/* gt_config.precision = 1 (BASELINE configs[4]) at d_model 256 / 512 with every Linear of a layer on the big-tile kernel: bf16 copies of
 * the GEMM operands.  The producers of every activation / gradient a Linear, dgrad or weight gradient of an encoder layer consumes write
 * it in bf16 (8 tensors per layer), a per-step kernel writes bf16 copies of the layers' weights and of their transposes, and the GEMMs
 * stage those (csrc/gt_gemm32.h gemm32h_kernel: half the bytes per flop).  level 1: the copies sit BESIDE the fp32 tensors; level 2 (the
 * default): the tensors nothing but a GEMM reads -- ctx, hact, dhid, dqkv, the dropout-masked dz copies -- are stored in bf16 ALONE (the
 * reference's torch autograd keeps the same intermediates, in fp32: nothing of its interface sees them).  0: off.  -1: the environment
 * (GT_BF16_SHADOWS=0/1/2) or the default.  Outputs, losses and gradients are bit-identical at every level.  Changes gt_workspace_bytes and
 * which gt_ws_find names are live: set it before sizing a workspace.  gt_operand_shadow_level: the level in force for a configuration. */
int gt_set_operand_shadows(int level);
/* Bumped by every call that changes a switch the workspace layout depends on (today: gt_set_operand_shadows).  A host that keeps workspaces
 * across calls compares it with the value it saw when it sized them and re-makes them on a mismatch (StepEngine.slot does). */
int gt_layout_epoch(void);
int gt_operand_shadow_level(const gt_config* cfg);
/* The precision a configuration really runs at: 2 only where gt_config.precision = 2 applies (see there), else 1 / 0. */
int gt_precision_in_force(const gt_config* cfg);
/* Weight gradients as RIDER workgroups (csrc/gt_seq_wg.h): in the SPLIT mode at d_model 128 the backward phases' launches carry, on
 * the CUs their 2 x batch sequence workgroups leave idle, the weight gradients whose operands the earlier phases completed; one
 * workgroup owns a 32 x 64 gradient tile over ALL tokens (no atomics: bitwise reproducible), and a tail launch does what cannot ride
 * (layer 0's in-proj, the input layer), the LayerNorm parameter gradients and the step-counter bump.  -1 = default (when at least 64
 * CUs are idle beside the sequence workgroups), 0 = off (the grouped dispatch at the end of backward), 1 = on wherever supported
 * (env GT_SEQ_RIDE=0/1 does the same).  Same results to fp32 rounding (another summation order over the tokens). */
int gt_set_seq_ride(int on);
/* Measurement aid (tools/wg_unit_bench.py): the rider units of backward phase `phase` as a launch of their own (token range split
 * `ksplit` ways), ADDING into grads; operands = what the last backward on this workspace left there. */
int gt_debug_seq_wg_phase(const gt_config* cfg, const float* params, float* grads, const float* x, float* ws, int phase, int ksplit,
                          gt_stream_t stream);
/* Bitwise-reproducible weight gradients: each output tile of a weight gradient is owned by ONE workgroup that walks all tokens
 * (no split over the token dimension), so the fp32 atomic adds have a single contributor per element.  Everything else of the
 * step is reproducible already (fixed-order reductions).  Off by default: the small shapes lose their token parallelism
 * (env GT_DETERMINISTIC=1 does the same). */
int gt_set_deterministic(int on);
/* Kernel launches of one gt_train_step for this configuration when it runs on the sequence-resident path (7 whole-sequence; SPLIT: 2 L + 4 with rider weight gradients, 2 L + 6
 * without; one less with GT_STEP_PACKS_CURRENT), 0 when
 * it runs one kernel per operation (dozens to hundreds).  A host that replays the step as a captured hipGraph can use it to
 * decide: measured on MI355X / ROCm 7.2 a 12-launch step is 2 % FASTER enqueued directly (0.241 vs 0.246 ms) -- the graph costs
 * ~0.4 us per node, and at this count the host stays ahead of the GPU by itself. */
int gt_step_launches(const gt_config* cfg);
int gt_profile_report(char* buf, size_t buf_len, int max_rows);

#ifdef __cplusplus
}
#endif
#endif /* GROOVE_HIP_H */
