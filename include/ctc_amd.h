/* ctc_amd.h -- C ABI of the MI355X-native CTC loss engine (libctc_amd.so).
 *
 * The drop-in boundary for the reference's loss path.  The reference has no FFI:
 * its loss is two Python nn.Modules (NoBlankCTC.py:22-141, NoBlankBinaryCTC.py:22-151)
 * called as  loss = ctc_loss(v_output, v_target, input_length, v_target_length)
 * (train.py:427, 576) followed by  loss.backward()  (train.py:444), plus
 * torch.nn.CTCLoss(blank=0) at models/layers/AsyncTFCriterion.py:198,319-321.
 * Each entry point below replaces one of those call sites' device work; the Python
 * host layer (ctc_amd/) binds them with ctypes and mirrors the modules' signatures.
 *
 * Conventions
 *  - plain pointers and sizes only; every pointer is a DEVICE pointer unless noted;
 *  - `stream` is a hipStream_t passed as void* (NULL = the null stream); calls only
 *    enqueue work: no allocation, no host synchronisation, graph-capture safe;
 *  - inputs are read-only; outputs are fully overwritten (grad rows t >= T_b get 0);
 *  - the caller owns all memory.  `workspace` must hold ctc_amd_workspace_bytes(...)
 *    bytes, be zero-filled ONCE when allocated, and not be shared by launches that
 *    may run concurrently (one workspace per stream); bytes [8,12) are the status word
 *    (ctc_amd_workspace_status), bytes [44,48) the gate word (ctc_amd_collective_gate);
 *  - return value: 0 on success, a hipError_t (> 0) from the launch, or one of the
 *    negative CTC_AMD_ERR_* codes; ctc_amd_error_string() describes any of them.
 */
#ifndef CTC_AMD_H
#define CTC_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CTC_AMD_ABI_VERSION 2

#define CTC_AMD_ERR_BAD_ARGUMENT      (-1)  /* null pointer, non-positive size ... */
#define CTC_AMD_ERR_UNSUPPORTED_SHAPE (-2)  /* S > 256 (255 for blank-CTC); binary: T*S beyond LDS */
#define CTC_AMD_ERR_CODE_OVERFLOW     (-3)  /* target dedup in the reference's int32 row codes with C > 64: the
                                             * reference raises OverflowError there (2**o at o >= 64) */

/* variants for ctc_amd_workspace_bytes */
#define CTC_AMD_NOBLANK 0
#define CTC_AMD_BINARY  1
#define CTC_AMD_BLANK   2

int ctc_amd_abi_version(void);
const char *ctc_amd_error_string(int code);

/* Bytes of device workspace a call of this shape needs (>= 256: a header with the in-launch batch
 * reduction's words and the status word, then per-variant areas).  No-blank lattices that do not fit
 * in LDS (long sequences) and the blank-CTC lattice live in this workspace. */
size_t ctc_amd_workspace_bytes(int variant, int T, int B, int C, int S);

/* NoBlankCTC.forward (NoBlankCTC.py:129-141) + the gradient autograd would produce
 * for it (train.py:444), one fused launch.
 *   x        [T,B,C] fp32 raw logits, element strides stride_t / stride_b, unit
 *            stride over C (the module applies LogSoftmax(dim=2) itself, :136)
 *   labels   [B,S] class indices, int32 (labels_i64 = 0) or int64 (= 1); entries at
 *            l >= tgt_len[b] are never dereferenced (the dataset pads with -1)
 *   in_len   [B] int64, 1 <= T_b <= T        tgt_len [B] int64, 1 <= L_b <= S
 *   nll      [B]  out: -alpha[T_b-1, L_b-1]  (1e13 when no alignment exists)
 *   loss     [1]  out: loss_scale * sum_b nll[b]   (loss_scale = 1/B for the
 *            reference's batch mean :139-140; 1/B_global on a batch shard).  The sum is taken
 *            in-launch in FIXED POINT (order-independent, bitwise reproducible): every nll is
 *            rounded to a multiple of 2^-F, F = 27 - ceil(log2 B) fractional bits (B = 256: 2^-19,
 *            i.e. the loss is within 2^-20 of the exactly rounded mean; the reference's
 *            torch.mean is an fp32 sum in batch order).  nll[] itself and the gradient are
 *            not quantised.  Values that do not fit (>= 8192, the 1e13 sentinel, NaN) are
 *            added from memory in double.
 *   grad     [T,B,C] contiguous out, or NULL for a forward-only call:
 *            grad_scale * (softmax(x)[t,b,c] - sum_{l<L_b, lab[b,l]=c} gamma_t(l)),
 *            exactly 0 for t >= T_b and for samples with no alignment
 */
int ctc_amd_noblank_loss_grad(const float *x, int64_t stride_t, int64_t stride_b,
                              const void *labels, int labels_i64,
                              const int64_t *in_len, const int64_t *tgt_len,
                              int T, int B, int C, int S,
                              float loss_scale, float grad_scale,
                              float *nll, float *loss, float *grad,
                              void *workspace, void *stream);

/* The label-smoothing variant sketched in comments at NoBlankCTC.py:100-107 ("the true class times lambda,
 * the others times (1 - lambda) / n_unit") and CrossEntropy.py: the emission of label position l becomes
 *   e[t,b,l] = lambda * lp[t,b,c_l] + (1 - lambda)/C * sum_{n != c_l} lp[t,b,n],   lp = LogSoftmax(x),
 * everything else as ctc_amd_noblank_loss_grad; grad = grad_scale * ((1 - b) softmax - a occupancy - b) with
 * b = (1 - lambda)/C, a = lambda - b.  The reference never runs this code (parity unpinned: checked against
 * the numpy restatement and finite differences).  Shapes: C even <= 256, S <= 31, T <= 168, 8-byte aligned
 * rows (the four-rows-per-wave kernel); others return CTC_AMD_ERR_UNSUPPORTED_SHAPE.  lambda in [0, 1]. */
int ctc_amd_noblank_smoothed_loss_grad(const float *x, int64_t stride_t, int64_t stride_b,
                                       const void *labels, int labels_i64,
                                       const int64_t *in_len, const int64_t *tgt_len,
                                       int T, int B, int C, int S, float label_smoothing,
                                       float loss_scale, float grad_scale,
                                       float *nll, float *loss, float *grad,
                                       void *workspace, void *stream);

/* NoBlankBinaryCTC.forward (NoBlankBinaryCTC.py:139-151) + gradient.  Same as above
 * except   y [B,S,C] fp32 multi-hot / soft targets in [0,1], contiguous;
 * emission = -BCELoss(sigmoid(x[t,b,:]), y[b,l,:]) (:112,:88, logs clamped at -100);
 * grad = grad_scale/C * (sigmoid(x) - sum_l gamma_t(l) y[b,l,c]).
 */
int ctc_amd_binary_loss_grad(const float *x, int64_t stride_t, int64_t stride_b,
                             const float *y,
                             const int64_t *in_len, const int64_t *tgt_len,
                             int T, int B, int C, int S,
                             float loss_scale, float grad_scale,
                             float *nll, float *loss, float *grad,
                             void *workspace, void *stream);

/* torch.nn.CTCLoss(blank, reduction='mean', zero_infinity=False) as used at
 * models/layers/AsyncTFCriterion.py:198,319-321 (padded [B,S] targets).
 *   log_probs [T,B,C] fp32, already normalised; targets [B,S] int32/int64
 *   nll   [B]  out: un-normalised negative log-likelihood (+inf if infeasible)
 *   loss  [1]  out: loss_scale * sum_b nll[b] / max(L_b,1)
 *   grad  [T,B,C] out or NULL: (exp(lp) - occupancy) * grad_scale / max(L_b,1)
 * Long sequences (T >= 256) on batches of #CUs/11..#CUs/2 samples with >= 4 lattice states per lane and
 * B*C >= 16384 (BASELINE config 5 and its neighbourhood) run as ONE persistent
 * launch of at most one workgroup per CU in which workgroups wait for each other (bounded: a wait
 * that runs out poisons nll / grad with NaN instead of hanging).  Kernels of other streams on the
 * same device can only delay it; the workspace belongs to one call in flight at a time, as for every
 * entry point.  ctc_amd_blank_set_schedule() forces / forbids that schedule.
 * Accuracy: the lattice scans are fp32 in the log2 domain, like torch's own fp32 kernels, and lose resolution with T
 * (alpha reaches -2e4 at T = 2000, where one ulp is 2e-3): the gradient of the BATCH-MEAN loss is within 1e-4 of float64
 * at every tested shape (8e-6 at BASELINE config 5, where torch's fp32 CPU kernel is 1.9e-4 off), but the un-normalised
 * per-sample occupancies behind it -- grad * max(L_b,1) / grad_scale -- are only good to ~1e-2 at T = 2000 (7e-3
 * measured).  A caller that rescales the gradient per sample by factors >> 1 inherits that.
 */
int ctc_amd_blank_loss_grad(const float *log_probs, int64_t stride_t, int64_t stride_b,
                            const void *targets, int targets_i64,
                            const int64_t *in_len, const int64_t *tgt_len,
                            int T, int B, int C, int S, int blank,
                            float loss_scale, float grad_scale,
                            float *nll, float *loss, float *grad,
                            void *workspace, void *stream);

/* Schedule of the long-sequence blank-CTC path: -1 = the library's own choice (default), 1 / 0 = force /
 * forbid the single persistent launch, 2 = force it with the worker pool gathering the emission rows (the
 * library's own choice around BASELINE config 5) (tests and measurements; process-wide, thread-safe).  The
 * environment variable CTC_AMD_BLANK_FUSED=1 / 0, read ONCE at first use, sets the initial value. */
int ctc_amd_blank_set_schedule(int mode);

/* In-launch hand-offs between waves / workgroups wait with a bound (~1 s).  A wait that runs out never
 * yields a plausible number: the outputs that could not be produced are filled with NaN (nll of the
 * sample, the loss, the sample's gradient rows) and a bit is ORed into the workspace's STATUS word, which
 * stays set until cleared here.  Reads the word (synchronising `stream`); clear != 0 resets it.
 * Bits: 1 no-blank, 2 binary, 4 blank-CTC launch starved.  Never observed outside fault-injection builds;
 * the persistent blank-CTC launch is the one place where another process's kernels could cause it. */
int ctc_amd_workspace_status(void *workspace, int clear, void *stream, unsigned *status_host);

/* Collective gate (batch-sharded use, one process per GPU): keeps a collective from taking CUs away from the
 * NEXT loss launch.  The no-blank / binary launch at B >= #CUs wants every CU (one 1024-thread workgroup, 122 KB
 * of LDS, 288 of a SIMD's 512 registers per lane); RCCL's collective kernel (256 threads, 19.7 KB of LDS, 261-280
 * registers per lane) cannot share a CU with such a workgroup, so an all-reduce that is dispatched just BEFORE
 * the launch costs it a whole second round of workgroups (measured: 15 -> 23 us per launch with one resident
 * collective workgroup; dispatched after the launch has filled the chip it costs nothing).  Enqueue this on the
 * stream the collective will be ordered behind, AFTER the previous loss launch and BEFORE the collective: a
 * one-wave kernel that returns once min(B, #CUs) workgroups of the loss launch that uses `workspace` have
 * started, or after timeout_us (bounded: never gate a collective that no loss launch follows).  Counting is
 * switched on by the first gate used on a workspace (the loss launches of a workspace that never saw a gate pay
 * nothing for it; the launch right after the first gate is not counted yet and that gate runs into its bound):
 * every no-blank / binary workgroup then counts itself at entry in sixteen sharded words of the workspace, the
 * launch's last workgroup puts them back to 0.  The blank-CTC launches do not count (a collective is short against
 * them). */
int ctc_amd_collective_gate(void *workspace, int B, int timeout_us, void *stream);

/* backward of the autograd.Function: grad[i] *= *grad_out (a device scalar, the
 * upstream gradient of the 0-dim loss).  Every workgroup reads *grad_out and exits
 * at once when it is exactly 1.0f (the loss.backward() case, train.py:444), so the
 * common case moves no data and needs no host synchronisation. */
int ctc_amd_scale_grad(float *grad, const float *grad_out, size_t n, void *stream);

/* Best-path (Viterbi) alignment on the no-blank lattice: the max-semiring twin of
 * the alpha recursion (the reference evaluates with per-step argmax and
 * DTW-like helpers, train.py:82-136,434).
 *   path  [B,T] int32 out: label position l_t of the best alignment for t < T_b,
 *         -1 for t >= T_b or when no alignment exists
 *   score [B]   out: log-probability of that alignment
 */
int ctc_amd_noblank_best_path(const float *x, int64_t stride_t, int64_t stride_b,
                              const void *labels, int labels_i64,
                              const int64_t *in_len, const int64_t *tgt_len,
                              int T, int B, int C, int S,
                              int32_t *path, float *score,
                              void *workspace, void *stream);

/* The same on the lattice of the binary variant (SURVEY 8f-1 for NoBlankBinaryCTC): y [B,S,C] float label rows;
 * cell (t, l) costs -nn.BCELoss()(sigmoid(x[t,b,:]), y[b,l,:]) (NoBlankBinaryCTC.py:112,:88,:146).  path [B,T] (label-row
 * positions, -1 behind T_b or when no alignment exists), score [B] its log-probability. */
int ctc_amd_binary_best_path(const float *x, int64_t stride_t, int64_t stride_b, const float *y,
                             const int64_t *in_len, const int64_t *tgt_len,
                             int T, int B, int C, int S,
                             int32_t *path, float *score, void *workspace, void *stream);

/* Target construction: the dedup step of the reference's dataset preparation,
 * datasets/charades_ctc_next_pred.py:646-651,663-678 (same code at :503-505,523-531) -- out[b] = the rows of
 * rows[b] whose code is new, in order of first appearance, remaining rows filled with -1 (:676-678);
 * length[b] = how many.  rows, out: [B,S,C] int32; length [B] int64.  No workspace.
 *   exact_rows = 0 (the reference's arithmetic, bit-exact at every C <= 64): rows are compared through the
 *     int32 code  sum_o row[o] * 2**o  as torch accumulates it into an IntTensor -- wrapped to 32 bits, so
 *     class 31 is the sign bit and classes 32..63 drop out (opts.py:60-61 default to 38 / 33 classes: rows
 *     that differ only there collide, rows made only of them never enter); a row enters when its code is
 *     not among the kept codes, an array that starts as zeros (code 0 never enters).  C > 64 returns
 *     CTC_AMD_ERR_CODE_OVERFLOW: the reference raises OverflowError at 2**64.
 *   exact_rows = 1: rows compared exactly over all C classes (which classes are non-zero), any C. */
int ctc_amd_dedup_multihot_targets(const int32_t *rows, int B, int S, int C, int exact_rows,
                                   int32_t *out, int64_t *length, void *stream);

/* Per-step posteriors of the no-blank lattice: gamma[b,t,l] = P(state l at step
 * t | x, targets) = exp(alpha_t(l) + beta_t(l) + nll), the quantity the loss gradient scatters
 * by class; rows sum to 1 for t < T_b and are 0 beyond T_b / L_b.  Same inputs as
 * ctc_amd_noblank_loss_grad; gamma is [B,T,S] fp32 contiguous; nll [B] is also produced. */
int ctc_amd_noblank_posteriors(const float *x, int64_t stride_t, int64_t stride_b,
                               const void *labels, int labels_i64,
                               const int64_t *in_len, const int64_t *tgt_len,
                               int T, int B, int C, int S,
                               float *nll, float *gamma,
                               void *workspace, void *stream);

/* The same for the binary lattice (NoBlankBinaryCTC): gamma[b,t,l] = P(target row l at step t | x, y) -- what the
 * binary loss gradient contracts with the target rows.  y [B,S,C] as ctc_amd_binary_loss_grad; gamma [B,T,S], nll [B].
 * Shapes of the pipelined binary kernel only (S <= 64, T <= 168, C <= 256, LDS images fit); others return
 * CTC_AMD_ERR_UNSUPPORTED_SHAPE. */
int ctc_amd_binary_posteriors(const float *x, int64_t stride_t, int64_t stride_b, const float *y,
                              const int64_t *in_len, const int64_t *tgt_len,
                              int T, int B, int C, int S,
                              float *nll, float *gamma,
                              void *workspace, void *stream);

/* The producer step of the logits: one torch.nn.LSTMCell step of the reference's LSTM_cell.forward
 * (LSTM.py:39-51: `v_hsn, v_csn = self.v_cell(v, (v_hsn, v_csn)); v_series[time] = v_hsn`), fused with the write of
 * the hidden state into the logits tensor the losses read.
 *   x [B,I], h [B,H], c [B,H]: the cell's input and state (contiguous fp32);  w_ih [4H,I], w_hh [4H,H], b_ih, b_hh [4H]:
 *   nn.LSTMCell's parameters (gate order i, f, g, o);  h_out, c_out [B,H]: the new state (may alias h / c);
 *   gates_out [B,4H] or NULL: the gate activations (what a backward pass needs);
 *   series_row or NULL: row b of v_series[time] starts at series_row + b * series_stride_b; columns [0,H) get the new
 *   hidden state, columns [H, series_cols) get pad_value (a pitch of H + 1 with pad_value = -1e30 gives an odd class
 *   count the even, 8-byte aligned rows of the fastest loss kernel; the padded class has softmax 0: no loss or gradient
 *   value changes).  gates = x W_ih^T + b_ih + h W_hh^T + b_hh; c' = sigma(f) c + sigma(i) tanh(g); h' = sigma(o) tanh(c'). */
int ctc_amd_lstm_cell_step(const float *x, const float *h, const float *c,
                           const float *w_ih, const float *w_hh, const float *b_ih, const float *b_hh,
                           int B, int I, int H,
                           float *h_out, float *c_out, float *gates_out,
                           float *series_row, int64_t series_stride_b, int series_cols, float pad_value,
                           void *stream);

/* The whole loop of LSTM_cell.forward (LSTM.py:44-51) over the T frames as ONE launch, for the reference's class counts
 * (I + H <= 80, I <= 64, H <= 64; other sizes: CTC_AMD_ERR_UNSUPPORTED_SHAPE -- step frame by frame with the call above).
 *   x [T,B,I]: the cell inputs of all frames (contiguous fp32: `self.v(feat[time])` for every time);  h0, c0 [B,H];
 *   series: row (t, b) of v_series starts at series + t * series_stride_t + b * series_stride_b (columns as above);
 *   gates_out [T,B,4H], cells_out [T+1,B,H] (c_0 .. c_T) or NULL: what a backward pass needs;  h_out, c_out [B,H] or NULL:
 *   the state after the last frame.  Same arithmetic, in the same order, as T calls of ctc_amd_lstm_cell_step. */
int ctc_amd_lstm_series(const float *x, const float *h0, const float *c0,
                        const float *w_ih, const float *w_hh, const float *b_ih, const float *b_hh,
                        int T, int B, int I, int H,
                        float *series, int64_t series_stride_t, int64_t series_stride_b, int series_cols, float pad_value,
                        float *gates_out, float *cells_out, float *h_out, float *c_out, void *stream);

/* The backward RECURRENCE of ctc_amd_lstm_series as one launch (H <= 64): from the upstream gradient of v_series
 * (row (t, b) at d_series + t * ds_stride_t + b * ds_stride_b, unit stride over the classes, columns [0,H) read) and the
 * gates_out / cells_out of the forward launch to dpre_out [T,B,4H] -- the gradient of every frame's gate pre-activations --
 * and the gradients of the initial state dh0_out, dc0_out [B,H].  What is left of the backward pass has no recurrence in
 * it: dx = dpre W_ih, dW_ih = sum_tb dpre^T x, dW_hh = sum_tb dpre^T h_{t-1}, db = sum_tb dpre (plain GEMMs). */
int ctc_amd_lstm_series_backward(const float *d_series, int64_t ds_stride_t, int64_t ds_stride_b,
                                 const float *gates, const float *cells, const float *w_hh,
                                 int T, int B, int H, float *dpre_out, float *dh0_out, float *dc0_out, void *stream);

/* The HEAD of the producer (LSTM.py:8-18 `nn.Linear(inDim, outDim) -> nn.BatchNorm1d -> nn.ReLU -> nn.Dropout`,
 * called once per frame at LSTM.py:48) for all T frames as ONE launch: out[t] = dropout(relu(batchnorm(feat[t] W^T + b))).
 *   feat: row (t, b) of K floats at feat + t * feat_stride_t + b * feat_stride_b (16-byte aligned rows, K a multiple of 16);
 *   weight [C,K], bias [C]: the Linear layer;  bn_weight, bn_bias [C]: BatchNorm1d's affine parameters;
 *   running_mean / running_var [C]: eval mode (BatchNorm on its running statistics); both NULL: train mode -- the
 *   statistics of each FRAME's batch of B rows (biased variance, eps inside the square root), as the reference's per-frame
 *   calls compute them (B <= 256: one workgroup holds a frame's rows; B >= 2);
 *   mask [T,B,C] or NULL: the dropout mask, already scaled by 1 / (1 - p) (the caller draws it: the random stream stays
 *   the framework's);  out: row (t, b) of C floats at out + t * out_stride_t + b * out_stride_b;
 *   linear_out [T,B,C], save_mean / save_var / save_invstd [T,C] or NULL: what a backward pass and the update of the
 *   running statistics need (train mode; running = (1 - m) running + m stat, frame after frame, var unbiased: the caller).
 * The product is exact fp32 on the matrix cores (an fmaf chain per output; the order of the sum over k differs from a
 * BLAS GEMM's: results agree with torch's layers to ~1e-6 relative). */
int ctc_amd_head_forward(const float *feat, int64_t feat_stride_t, int64_t feat_stride_b,
                         const float *weight, const float *bias, const float *bn_weight, const float *bn_bias,
                         const float *running_mean, const float *running_var, float eps, const float *mask,
                         int T, int B, int K, int C,
                         float *out, int64_t out_stride_t, int64_t out_stride_b,
                         float *linear_out, float *save_mean, float *save_var, float *save_invstd, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* CTC_AMD_H */
