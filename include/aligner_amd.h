/*
 * aligner_amd.h -- C ABI of libaligner_amd.so (MI355X / gfx950).
 *
 * Drop-in boundary for the TTS alignment hot path of xiaozhah/Aligner:
 *   monotonic_align.maximum_path(value, mask)      reference __init__.py:6-21
 *   monotonic_align.core.maximum_path_c(...)       reference core.pyx:38-45
 *   (per-utterance DP maximum_path_each)           reference core.pyx:7-35
 * plus the soft-attention front end that produces `value` (build-defined spec,
 * SURVEY.md 7.4; the reference snapshot only links the paper, README.md:50).
 *
 * Conventions
 *   - plain pointers and sizes only; no torch / HIP types in any signature.
 *     `stream` is a hipStream_t passed as void* (NULL = the null stream).
 *   - every *_dev pointer is device memory on the current HIP device; tensors
 *     are C-contiguous, [B, Tx, Ty] with the mel axis (Ty) contiguous, exactly
 *     the layout the reference's memoryviews demand (core.pyx:9,40).
 *   - all device entry points are asynchronous on `stream`, never allocate,
 *     never synchronise, and are hipGraph-capturable.
 *   - return 0 on success or a negative ALIGNER_E* code; aligner_last_error()
 *     returns a thread-local message for the last failure on this thread.
 *   - the library contains NO CPU implementation of the path: without a GPU the
 *     device entry points fail with ALIGNER_EHIP.
 */
#ifndef ALIGNER_AMD_H
#define ALIGNER_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 4: aligner_maxpath_workspace_bytes grew (per-utterance mask verdicts), aligner_fused_align_f32 left the library,
 *    the prepared-weights image of round 4 (old image + the GEMM form's) and the forward-sum / search workspaces of round 4
 *    are what the *_bytes functions of THIS version return: size every buffer with them, never with constants;
 *    new entry points aligner_softattn_ld / aligner_maxpath_ld (a row pitch for the pipeline's own intermediate) and the
 *    test flag ALIGNER_F_TEST_IMPATIENT_FIRST_HALF. */
#define ALIGNER_ABI_VERSION 4

/* error codes */
#define ALIGNER_OK       0
#define ALIGNER_EINVAL (-22) /* null pointer / bad shape / bad dtype / bad flag */
#define ALIGNER_EDOM   (-33) /* shape outside what the kernels support          */
#define ALIGNER_ENOSPC (-28) /* workspace too small                             */
#define ALIGNER_EHIP    (-5) /* HIP runtime failure (message has the detail)    */

/* element types for mask / path buffers */
#define ALIGNER_DT_F32  0
#define ALIGNER_DT_F16  1
#define ALIGNER_DT_BF16 2
#define ALIGNER_DT_F64  3
#define ALIGNER_DT_I32  4
#define ALIGNER_DT_U8   5   /* also torch.bool */
#define ALIGNER_DT_I64  6

/* flags for aligner_maxpath_* */
#define ALIGNER_F_STRICT_MASK   1  /* multiply value by mask element-wise first
                                      (__init__.py:11) instead of using the mask
                                      for lengths only.  The result is that of
                                      the multiply, bit for bit; the multiply
                                      itself happens only for the utterances whose
                                      mask is not ONE on every score the search
                                      reads (the rectangle [0,t_x) x [0,t_y)):
                                      that is checked on the device, per
                                      utterance -- by the workgroups that write
                                      the dense path beside an optimistic search
                                      (a second launch redoes the others), or by
                                      a pass over the mask in front of it.      */
#define ALIGNER_F_COMPAT_TXGTTY 2  /* t_x > t_y: reproduce the reference's
                                      result -- its forward band is empty, so
                                      its backtrack (core.pyx:32-35) walks up
                                      from row t_x-1 on the RAW scores --
                                      instead of an all-zero path and
                                      ALIGNER_ST_BAD_LENGTHS                   */
#define ALIGNER_F_FORCE_GENERIC 4  /* use the generic (barrier-per-frame) kernel */

#define ALIGNER_F_NO_PREV_TABLE 16 /* testing: do not keep the backtrack's second LDS table (the walk then
                                      takes the flag-and-retry steps it uses for long utterances)   */
#define ALIGNER_F_STREAM_PATH  128 /* the dense path is written with non-temporal stores: a large output that
                                      nothing on the GPU reads back soon then does not displace the score
                                      tensors of other batches in flight from the caches (three batches in
                                      flight: 39.9 -> 33.8 us per step at [64,200,1000]; one batch alone is
                                      2 us slower).  The caller's choice; off by default                */
#define ALIGNER_F_ONE_CU       256 /* never split an utterance over two workgroups.  By default text of 253..504
                                      rows over 1792 or more mel frames, in a batch of at most an eighth of the
                                      CU count (B <= 32 on MI355X), runs as two workgroups per utterance, each on its
                                      own CU: same results, the sweep bound by two CUs' vector units instead of
                                      one's (long-form [8,500,4000]: 134 -> 106 us).  The boundary row between the
                                      halves travels through the workspace, which is why the call then starts
                                      with a small fill kernel on the same stream.  Only taken when all 2B workgroups
                                      fit the chip at once; the second half's waits are bounded, and if the first half
                                      did not deliver in time (other work kept it off the GPU for ~0.3 s) the
                                      utterance comes back with an all-zero path, zero durations and
                                      ALIGNER_ST_INTERNAL in the status word -- check it (aligner_maxpath_read_status;
                                      aligner_maxpath_host_f32 does and fails with ALIGNER_EHIP) */
#define ALIGNER_F_TWO_CUS      512 /* take the two-workgroup form whenever the text has 253..504 rows, whatever
                                      the batch size and mel length (testing; a short sweep loses by it) */
#define ALIGNER_F_PATH_PREZEROED 2048 /* aligner_maxpath*: path_out_dev already holds zeros (aligner_maxpath_zero_path, on another
                                      stream or graph branch, BEFORE this call starts): the search kernel writes the
                                      path's ones itself -- core.pyx:33 on the np.zeros of __init__.py:15 -- and no
                                      expand kernel follows.  The dense path costs the step's serial chain nothing
                                      (bench.py, one batch at a time: 65.9 -> see DESIGN 5).  A path that is NOT all
                                      zero keeps its stale ones: the caller's contract                        */
#define ALIGNER_F_SEPARATE_EXPAND 4096 /* aligner_maxpath*: always write the dense path with the expand kernel behind the search.
                                      By default, where the batch leaves at least 32 CUs idle and the one-workgroup-
                                      per-utterance kernel runs, the search launch carries extra workgroups that write
                                      the path's zeros on those CUs while the others search, and every utterance's
                                      workgroup writes its ones at the end: one launch, nothing of the 4*B*Tx*Ty-byte
                                      write left on the serial chain (same bits; A/B switch for measurements)  */
#define ALIGNER_F_TEST_DROP_FIRST_HALF 1024 /* testing: in the two-workgroup form the first half of every utterance leaves
                                      without delivering, so the second gives up after its bounded wait: all-zero
                                      path, zero durations, ALIGNER_ST_INTERNAL -- the defined failure of that form */
#define ALIGNER_F_TEST_IMPATIENT_FIRST_HALF 16384 /* testing: in the two-workgroup form the first half gives up waiting for the
                                      backtrack's hand-over at once (as it would after its bounded wait on a contended
                                      chip): ALIGNER_ST_INTERNAL is set, and the second half, finding that, walks
                                      every row itself -- the answer is late, not lost */
#define ALIGNER_F_TEST_DROP_ZERO_REPORTS 8192 /* testing: the zero workgroups of the one-launch dense path (above) write their
                                      zeros but never report, so every utterance's workgroup gives up after a (here
                                      shortened) bounded wait -- the defined failure of that form: no 1 in the path
                                      (all zeros), zero durations, no token on any frame, ALIGNER_ST_INTERNAL */
#define ALIGNER_F_WRITE_Q       64 /* also overwrite the fp32 score block with the running scores Q inside
                                      the band, in place, exactly as the reference does (core.pyx:18,30:
                                      `value[x, y] = max(v_cur, v_prev) + value[x, y]`).  Takes the
                                      barrier-per-frame kernel (the fast one never materialises Q); the
                                      score pointer must be writable                                 */

/* bits of the device status word (aligner_maxpath_read_status) */
#define ALIGNER_ST_BAD_LENGTHS  1  /* some utterance had t_x < 1 or t_x > t_y
                                      (undefined behaviour in the reference,
                                      core.pyx:15,33-34); its path is all zero */
#define ALIGNER_ST_CLAMPED      2  /* some t_x > Tx or t_y > Ty was clamped     */
#define ALIGNER_ST_INTERNAL     4  /* internal consistency check failed         */

int         aligner_abi_version(void);
const char *aligner_last_error(void);

/* Number of visible HIP devices (0 when there is no GPU); never fails. */
int         aligner_device_count(void);

/* ---- monotonic alignment search ---------------------------------------- */

/* Bytes of device workspace aligner_maxpath_f32 needs for this shape. */
size_t aligner_maxpath_workspace_bytes(int B, int Tx, int Ty);

/* Replaces __init__.py:18-19: t_x[b] = sum_x mask[b,x,0], t_y[b] = sum_y mask[b,0,y]
 * (float sum truncated to int32, like ndarray.astype(np.int32)). */
int aligner_lengths_from_mask(const void *mask_dev, int mask_dtype,
                              int B, int Tx, int Ty,
                              int32_t *t_xs_dev, int32_t *t_ys_dev, void *stream);

/*
 * Replaces maximum_path_c (core.pyx:40-45) + the wrapper's marshalling
 * (__init__.py:11-21) with everything resident in HBM.
 *
 *   value_dev   [B,Tx,Ty] fp32, NOT modified (the reference mutates its private
 *               host copy; the caller's tensor is untouched there too).
 *   mask_dev    optional [B,Tx,Ty] of mask_dtype (F32, BF16, F16, U8 or I32).  Used (a) to derive
 *               lengths when t_xs_dev/t_ys_dev are NULL and (b) for the
 *               element-wise multiply when ALIGNER_F_STRICT_MASK is set.
 *   t_xs_dev,t_ys_dev  optional [B] int32 lengths; when NULL they are derived
 *               from mask_dev into the workspace.
 *   path_out_dev optional [B,Tx,Ty] of path_dtype (F32,F16,BF16,F64,I32,U8,I64),
 *               fully written (zeros and ones): the reference's return value.
 *   tok_out_dev optional [B,Ty] int32: text-token index per mel frame, -1 for
 *               frames >= t_y.
 *   dur_out_dev optional [B,Tx] int32: frames per token (= path.sum(2)).
 *   max_neg_val core.pyx:40 (default -1e9 in the reference).
 */
/* The same for scores of `value_dtype` F32, BF16 or F16: 16-bit scores are up-cast to fp32 (exactly) inside the
 * kernels' loaders -- no conversion pass, half the read traffic -- and the DP runs in fp32 like the reference
 * (__init__.py:14).  With ALIGNER_F_STRICT_MASK the mask must have the scores' dtype: the product is rounded in
 * that dtype, as torch's `value * mask` (__init__.py:11) is.  16-byte aligned pointers and Ty % 8 == 0 take the
 * fast kernels, anything else the generic one. */
int aligner_maxpath(const void *value_dev, int value_dtype,
                    const void *mask_dev, int mask_dtype,
                    const int32_t *t_xs_dev, const int32_t *t_ys_dev,
                    void *path_out_dev, int path_dtype,
                    int32_t *tok_out_dev, int32_t *dur_out_dev,
                    void *workspace_dev, size_t workspace_bytes,
                    int B, int Tx, int Ty,
                    float max_neg_val, int flags, void *stream);
/* aligner_maxpath for scores with a ROW PITCH of their own: element [b,x,y] at ((b*Tx + x)*ld_value + y), ld_value >= Ty --
 * what aligner_softattn_ld writes (rows that start on whole 128-byte lines).  Lengths are given, there is no
 * mask (ALIGNER_F_STRICT_MASK, ALIGNER_F_WRITE_Q: ALIGNER_EINVAL); path / tok / dur are the contiguous outputs of
 * aligner_maxpath.  The columns Ty .. ld_value-1 are never read into a result. */
int aligner_maxpath_ld(const void *value_dev, int value_dtype, int ld_value,
                       const int32_t *t_xs_dev, const int32_t *t_ys_dev,
                       void *path_out_dev, int path_dtype, int32_t *tok_out_dev, int32_t *dur_out_dev,
                       void *workspace_dev, size_t workspace_bytes, int B, int Tx, int Ty,
                       float max_neg_val, int flags, void *stream);
int aligner_maxpath_forward(const void *value_dev, int value_dtype,
                            const void *mask_dev, int mask_dtype,
                            const int32_t *t_xs_dev, const int32_t *t_ys_dev,
                            int32_t *tok_out_dev, int32_t *dur_out_dev,
                            void *workspace_dev, size_t workspace_bytes,
                            int B, int Tx, int Ty,
                            float max_neg_val, int flags, void *stream);

/* fp32 scores (value_dtype = ALIGNER_DT_F32). */
int aligner_maxpath_f32(const float *value_dev,
                        const void *mask_dev, int mask_dtype,
                        const int32_t *t_xs_dev, const int32_t *t_ys_dev,
                        void *path_out_dev, int path_dtype,
                        int32_t *tok_out_dev, int32_t *dur_out_dev,
                        void *workspace_dev, size_t workspace_bytes,
                        int B, int Tx, int Ty,
                        float max_neg_val, int flags, void *stream);

/* The two stages of aligner_maxpath_f32 as separate launches (profiling and
 * callers that only want durations): forward sweep + backtrack -> token starts in
 * the workspace (+ optional tok/dur), and workspace -> dense 0/1 path. */
int aligner_maxpath_forward_f32(const float *value_dev,
                                const void *mask_dev, int mask_dtype,
                                const int32_t *t_xs_dev, const int32_t *t_ys_dev,
                                int32_t *tok_out_dev, int32_t *dur_out_dev,
                                void *workspace_dev, size_t workspace_bytes,
                                int B, int Tx, int Ty,
                                float max_neg_val, int flags, void *stream);
int aligner_maxpath_expand(const void *workspace_dev, void *path_out_dev, int path_dtype,
                           int B, int Tx, int Ty, void *stream);
int aligner_maxpath_expand_ex(const void *workspace_dev, void *path_out_dev, int path_dtype,
                           int B, int Tx, int Ty, int flags /* ALIGNER_F_STREAM_PATH */, void *stream);

/* The dense path in the reference's own two steps -- zeros (__init__.py:15), then one 1 per frame (core.pyx:33) -- for
 * callers with a second stream or a graph branch: aligner_maxpath_zero_path does not depend on the search and can
 * run BESIDE aligner_maxpath_forward*; aligner_maxpath_scatter_path (after both) then writes t_y elements per
 * utterance instead of the whole tensor.  Together they equal aligner_maxpath_expand, bit for bit, for every path
 * dtype; flags: ALIGNER_F_STREAM_PATH (non-temporal zeros).  bench.py's one-batch-at-a-time step is built this way. */
int aligner_maxpath_zero_path(void *path_out_dev, int path_dtype, int B, int Tx, int Ty, int flags, void *stream);
int aligner_maxpath_scatter_path(const void *workspace_dev, void *path_out_dev, int path_dtype,
                                 int B, int Tx, int Ty, void *stream);

/* Blocking read-and-clear of the workspace's status word (ALIGNER_ST_* bits).  The
 * word is sticky: kernels only OR bits into it, so the first 256 bytes of a fresh
 * workspace must be zero (hipMemset once at allocation; no per-call memset). */
int aligner_maxpath_read_status(void *workspace_dev, int32_t *status_host, void *stream);

/* Development aid: install (or clear with NULL) a device buffer of B*16*8 uint64 that
 * the forward kernels fill with shader-clock stamps per wave (see maxpath.hip). */
void aligner_debug_set_stamps(void *stamps_dev);

/* Development switches, process-wide (0/1; defaults from the environment variables ALIGNER_FWDSUM_ONE_WAVE /
 * ALIGNER_SOFTATTN_EXACT, read once at load): "fwdsum_one_wave" forces the one-sweeping-wave forward-sum
 * kernels, "softattn_exact" the exact-product similarity kernel; "mobo_drop_segment" (a segment index, -1 = off)
 * makes that position segment of every utterance withhold its rows from the next one, which then gives up after a
 * short wait: the test of the boundary search's defined failure; "mobo_start_lag" (rows a position segment lets the
 * one before it get ahead before it starts, default 0) and "mobo_lanes" (lanes per position of the split form: 1, 2, 4; 0 = the plan's
 * choice) are the boundary search's measurement knobs (tools/mobo_time.py).  ALIGNER_EINVAL for an unknown name. */
int aligner_debug_set_option(const char *name, int value);

/*
 * Host-buffer form with exactly maximum_path_c's contract (core.pyx:40):
 * paths[B,Tx,Ty] int32 receives the path (fully overwritten), values[B,Tx,Ty]
 * fp32 holds the scores; with ALIGNER_F_WRITE_Q in `flags` it comes back as the
 * running score Q inside every utterance's band, bit for bit what the reference
 * leaves there (without the flag it is only read), t_xs/t_ys[B] int32.  Stages
 * through device memory it allocates and frees; blocking.  Returns ALIGNER_EDOM
 * if any t_x < 1 or t_x > t_y.
 */
int aligner_maxpath_host_f32(int32_t *paths, float *values,
                             const int32_t *t_xs, const int32_t *t_ys,
                             int B, int Tx, int Ty, float max_neg_val, int flags);

/* ---- soft-attention front end (build-defined spec) ------- */

#define ALIGNER_SIM_L2   0  /* logit = -temperature * sum_c (q - k)^2 */
#define ALIGNER_SIM_DOT  1  /* logit = temperature * sum_c q*k        */

/*
 * logp[b,i,j] = log_softmax_i( logit[b,i,j] ) (+ log(prior[b,i,j] + 1e-8))
 *   keys_dev    [B,C,Tx] fp32  encoded text  (channel-major, as Conv1d emits)
 *   queries_dev [B,C,Ty] fp32  encoded mel
 *   t_xs_dev    optional [B] int32: text rows >= t_x are excluded from the
 *               softmax and written as -inf; NULL = all Tx rows valid.
 *   prior_dev   optional [B,Tx,Ty] fp32
 *   logp_out_dev [B,Tx,Ty] fp32; soft_out_dev optional [B,Tx,Ty] fp32 = softmax_i(logp)
 *   workspace_dev  aligner_softattn_workspace_bytes(B,C,Tx) bytes: the text operand split
 *               to bf16 halves in MFMA fragment order, prepared once per call.
 * Arithmetic: the contraction runs on the bf16 matrix cores with every fp32 operand split in two bf16 halves
 * (hi*hi + hi*lo + lo*hi, fp32 accumulate: ~2^-16 relative per product), which keeps |logp - fp32 reference|
 * below 1e-4 for temperature <= 0.002 (L2) / 0.2 (dot) on encodings of a few units per channel.  Sharper
 * temperatures take an exact-product kernel (fp32 MFMA, about a quarter of the matrix rate) automatically.
 */
size_t aligner_softattn_workspace_bytes(int B, int C, int Tx);
/* logp_out_dev of logp_dtype F32 or BF16 (round to nearest even; the layout aligner_maxpath reads directly:
 * BASELINE config 5, "bf16 similarity + int32 path"); everything else as aligner_softattn_f32. */
int aligner_softattn(const float *keys_dev, const float *queries_dev,
                     const int32_t *t_xs_dev, const float *prior_dev,
                     void *logp_out_dev, int logp_dtype, float *soft_out_dev,
                     void *workspace_dev, size_t workspace_bytes,
                     int B, int C, int Tx, int Ty,
                     float temperature, int sim, void *stream);
int aligner_softattn_f32(const float *keys_dev, const float *queries_dev,
                         const int32_t *t_xs_dev, const float *prior_dev,
                         float *logp_out_dev, float *soft_out_dev,
                         void *workspace_dev, size_t workspace_bytes,
                         int B, int C, int Tx, int Ty,
                         float temperature, int sim, void *stream);
/* aligner_softattn with a ROW PITCH for logp_out_dev: element [b,i,j] at ((b*Tx + i)*ld_logp + j), ld_logp >= Ty, rows on
 * 16-byte boundaries.  For the pipeline's own intermediate (similarity -> alignment search: aligner_maxpath_ld reads it):
 * with ld_logp a multiple of 32 (fp32) every 32-frame run a wave stores is one whole 128-byte line -- rows of 4000 bytes
 * (T_mel = 1000) cut three quarters of those runs in two.  Only the columns < Ty are written.  The row-tile
 * form only (Tx <= 224, C = 80 or 128, Ty % 4 == 0, no prior, no soft output): ALIGNER_EDOM otherwise. */
int aligner_softattn_ld(const float *keys_dev, const float *queries_dev,
                        const int32_t *t_xs_dev, const float *prior_dev,
                        void *logp_out_dev, int logp_dtype, int ld_logp, float *soft_out_dev,
                        void *workspace_dev, size_t workspace_bytes,
                        int B, int C, int Tx, int Ty,
                        float temperature, int sim, void *stream);

/* y[b,o,t] = act( bias[o] + sum_{i,k} w[o,i,k] * x[b,i,t+k-K/2] ), zero padded
 * ("same"), K odd; the 1-D conv of the text / mel encoders. relu: 0 or 1. */
int aligner_conv1d_f32(const float *x_dev, const float *w_dev, const float *bias_dev,
                       float *y_dev, int B, int Cin, int Cout, int T, int K,
                       int relu, void *stream);

/*
 * The same convolution with the weights prepared once (split into bf16 halves, in the matrix cores'
 * fragment order): aligner_conv1d_prepare_f32 writes aligner_conv1d_prepared_bytes(Cout,Cin,K) bytes,
 * aligner_conv1d_prepared_f32 consumes them.  Products are hi*hi + hi*lo + lo*hi in fp32 accumulators
 * (~2^-16 relative per product).  This is the fast path of the encoders: prepare per weight update.
 */
size_t aligner_conv1d_prepared_bytes(int Cout, int Cin, int K);
int aligner_conv1d_prepare_f32(const float *w_dev, void *prepared_dev, size_t prepared_bytes,
                               int Cout, int Cin, int K, void *stream);
int aligner_conv1d_prepared_f32(const float *x_dev, const void *prepared_dev, const float *bias_dev,
                                float *y_dev, int B, int Cin, int Cout, int T, int K,
                                int relu, void *stream);
/*
 * ... and with a workspace, which the fast form needs (csrc/convgemm.hip): the activations are split into bf16 halves
 * and transposed to channels-last fragments by a streaming pass into the workspace
 * (aligner_conv1d_workspace_bytes(B,Cin,Cout,T,K) bytes, 16-byte aligned; 0 = this layer has no such form and takes the
 * kernel of aligner_conv1d_prepared_f32), then a GEMM-structured kernel multiplies them with the prepared weights --
 * activations global -> LDS by LDS-DMA, the weights' fragments straight into registers, nothing waited for inside the
 * loop.  Same numerics as above.
 */
size_t aligner_conv1d_workspace_bytes(int B, int Cin, int Cout, int T, int K);
int aligner_conv1d_prepared_ws_f32(const float *x_dev, const void *prepared_dev, const float *bias_dev,
                                   float *y_dev, void *workspace_dev, size_t workspace_bytes,
                                   int B, int Cin, int Cout, int T, int K, int relu, void *stream);

/*
 * A whole encoder stack y = conv_n(...act(conv_1(x))) in one call (the text / mel encoders of SURVEY.md 7.4): the first
 * layer's input is split once, every k = 1 layer reads the split image its producer's epilogue wrote (no fp32 round trip
 * between layers), the last layer writes fp32 [B, Cout_n, T].  layers[i].prepared = that layer's
 * aligner_conv1d_prepare_f32 image; workspace aligner_conv_stack_workspace_bytes(...) bytes, 16-byte aligned
 * (0: some layer has no GEMM form -- run the layers one by one through aligner_conv1d_prepared_ws_f32 instead).
 */
typedef struct aligner_conv_layer {
    const void *prepared;       /* device: aligner_conv1d_prepare_f32(w, ..., Cout, Cin, K) */
    const float *bias;          /* device, [Cout], nullable */
    int Cin, Cout, K, relu;
} aligner_conv_layer;
size_t aligner_conv_stack_workspace_bytes(const aligner_conv_layer *layers, int n_layers, int B, int T);
int aligner_conv_stack_f32(const float *x_dev, const aligner_conv_layer *layers, int n_layers, float *y_dev,
                           void *workspace_dev, size_t workspace_bytes, int B, int T, void *stream);

/* ---- the callers either side of the path (build-defined specs) ---- */

/*
 * Forward-sum alignment objective: loss[b] = -log of the summed likelihood of ALL monotonic
 * alignments of logp[b] -- the column recurrence of maximum_path_each (core.pyx:17-30) with
 * log-sum-exp in place of max -- and, when grad_out_dev is not NULL, its gradient
 * d loss / d logp = -(posterior occupancy of each cell), 0 outside [0,t_x) x [0,t_y).
 *   logp_dev [B,Tx,Ty] fp32 (Tx <= 1024), t_xs_dev/t_ys_dev [B] int32, loss_out_dev [B] fp32
 *   (+inf where t_x < 1 or t_x > t_y), grad_out_dev optional [B,Tx,Ty] fp32,
 *   workspace_dev aligner_forward_sum_workspace_bytes(B,Tx,Ty) bytes (the alpha tiles).
 * With the gradient, batches of at most half the CU count run the alpha and the beta sweep side by side in one launch
 * (beta travels through grad_out_dev) and a combining pass writes the gradient in place; larger batches run forward, then
 * backward.  Same results up to fp32 rounding of the exponent sum (tests/test_objective.py).  T_mel % 4 == 0 and 16-byte
 * aligned logp / grad pointers take the faster tile movers.
 */
size_t aligner_forward_sum_workspace_bytes(int B, int Tx, int Ty);
int aligner_forward_sum_f32(const float *logp_dev, const int32_t *t_xs_dev, const int32_t *t_ys_dev,
                            float *loss_out_dev, float *grad_out_dev,
                            void *workspace_dev, size_t workspace_bytes,
                            int B, int Tx, int Ty, void *stream);

/*
 * The CTC form of the same objective -- the published one (OTA's forward-sum loss, README.md:21-25,50): a blank column
 * of log-prob `blank_logprob` (-1 in the paper's code) is put before the text rows, every frame is renormalised over
 * blank + the utterance's t_x text rows (log_softmax), and loss[b] = CTC loss of the token sequence 1..t_x over the
 * t_y frames = torch.nn.functional.ctc_loss(log_softmax(pad(scores)), targets = 1..t_x, reduction = "none") -- which
 * is what tests/test_objective.py checks it against, in float64.  grad_out_dev (optional) = d loss / d scores
 * = softmax over blank + text of the frame - posterior occupancy of the token, 0 outside [0,t_x) x [0,t_y).
 * scores_dev [B,Tx,Ty] fp32 (Tx <= 1023); loss +inf where t_x < 1 or t_x > t_y (no labelling exists; gradient 0).
 */
size_t aligner_forward_sum_ctc_workspace_bytes(int B, int Tx, int Ty);
int aligner_forward_sum_ctc_f32(const float *scores_dev, const int32_t *t_xs_dev, const int32_t *t_ys_dev,
                                float blank_logprob, float *loss_out_dev, float *grad_out_dev,
                                void *workspace_dev, size_t workspace_bytes,
                                int B, int Tx, int Ty, void *stream);

/* prior[b,x,y] = BetaBinomial(n = t_x, a = scaling*(y+1), b = scaling*(t_y-y)).pmf(x) for x < t_x,
 * y < t_y, 0 elsewhere; [B,Tx,Ty] fp32, the `prior_dev` operand of aligner_softattn_f32. */
int aligner_beta_binomial_prior_f32(const int32_t *t_xs_dev, const int32_t *t_ys_dev, float *prior_out_dev,
                                    int B, int Tx, int Ty, float scaling, void *stream);

/* Length regulator: out[b,c,y] = h[b,c,x(y)] where x(y) is the token whose duration interval holds
 * frame y (durations as aligner_maxpath_f32 writes them, negative values count as 0); frames past
 * sum(durations[b]) are 0.  h_dev [B,C,Tx], durations_dev [B,Tx] int32, out_dev optional [B,C,Ty],
 * tok_out_dev optional [B,Ty] int32 (-1 past the end). */
int aligner_regulate_f32(const float *h_dev, const int32_t *durations_dev, float *out_dev, int32_t *tok_out_dev,
                         int B, int C, int Tx, int Ty, void *stream);

/*
 * MoBoAligner monotonic boundary search (BASELINE config 5; build-defined spec from the paper the reference
 * links at README.md:49 -- its code is on a branch that is not in the snapshot; stated in full in
 * oracle/mobo_oracle.py and aligner_amd/csrc/mobo.hip).  Boundaries 0 = b_-1 < b_0 < ... < b_{t_x-1} = t_y,
 * token durations 1..max_duration, P(b_i = j | b_{i-1} = k) = softmax over the feasible j in (k, k+max_duration]
 * of energies[b,i,j-1].
 *   energies_dev   [B,Tx,Ty] of energy_dtype F32, BF16 or F16 (the alignment layout, mel axis contiguous)
 *   boundaries_out_dev [B,Tx] int32: b_i of the most probable boundary sequence (rows >= t_x: t_y); NULL (with
 *                      durations_out_dev and map_score_out_dev NULL as well) when only log_alpha / gamma are wanted
 *                      (masked energies that leave no boundary sequence are then not reported: log_alpha of the last
 *                      token's row is -inf everywhere, gamma 0)
 *   durations_out_dev  optional [B,Tx] int32; map_score_out_dev optional [B] fp32 (its log-probability)
 *   log_alpha_out_dev  optional [B,Tx,Ty] fp32: log P(b_i = j) at [i, j-1], -inf where impossible
 *   gamma_out_dev      optional [B,Tx,Ty] fp32 soft alignment P(b_{i-1} <= y < b_i); needs log_alpha_out_dev
 *   workspace_dev      aligner_boundary_search_workspace_bytes_ex(B,Tx,Ty,max_duration) bytes (the form without
 *                      max_duration is an upper bound for every window), first 256 zeroed once (status word as for
 *                      aligner_maxpath: ALIGNER_ST_BAD_LENGTHS for an utterance without any segmentation -- not
 *                      t_x <= t_y <= t_x*max_duration, or masked (-inf) energies that leave no boundary sequence a
 *                      positive probability; its outputs are 0 / -inf).  It holds the normalisers
 *                      [B,Tx,Ty] fp32, the per-(token, position) durations [B,Tx,Ty+1] u16 and the ring through
 *                      which the position segments of an utterance hand their last max_duration entries on.
 * Only the chain that is asked for runs: the max-product one alone without log_alpha_out_dev (the MAP sequence: 0.47 ms at
 * [8,500,4000], max_duration 32), the sum-product one alone without boundaries_out_dev (log_alpha, gamma: a training
 * step's forward pass), both in one kernel otherwise (0.70 ms) -- with identical results.
 * Three launches: normalisers of every (utterance, token, position) on the whole chip; the token chain with an
 * utterance's positions cut into segments, one workgroup each (as many as fit the CUs: [8,500,4000] runs 32 per
 * utterance), a segment waiting only for the one before it -- ALIGNER_ST_INTERNAL and all-zero boundaries /
 * durations for an utterance whose segment gave up waiting (only possible when other work keeps its predecessor
 * off the GPU for seconds); the MAP backtrack.  Ty is limited to 8192 positions per segment and by LDS
 * (24 bytes per position and per window entry); ALIGNER_EDOM beyond.
 */
size_t aligner_boundary_search_workspace_bytes(int B, int Tx, int Ty);
size_t aligner_boundary_search_workspace_bytes_ex(int B, int Tx, int Ty, int max_duration);
int aligner_boundary_search(const void *energies_dev, int energy_dtype,
                            const int32_t *t_xs_dev, const int32_t *t_ys_dev, int max_duration,
                            int32_t *boundaries_out_dev, int32_t *durations_out_dev, float *map_score_out_dev,
                            float *log_alpha_out_dev, float *gamma_out_dev,
                            void *workspace_dev, size_t workspace_bytes,
                            int B, int Tx, int Ty, void *stream);

/*
 * Gradient of the boundary search: d loss / d energies for a loss that reads log_alpha and / or gamma (what a
 * MoBoAligner training step needs behind the soft alignment; build-defined like the search, pinned in
 * oracle/mobo_oracle.py::boundary_search_backward to autograd over a float64 restatement of the forward pass).
 *   energies_dev, t_xs_dev, t_ys_dev, max_duration: as given to aligner_boundary_search
 *   log_alpha_dev       [B,Tx,Ty] fp32: that call's log_alpha_out_dev
 *   grad_log_alpha_dev  optional [B,Tx,Ty] fp32 cotangent of log_alpha (entries where log_alpha is -inf are ignored)
 *   grad_gamma_dev      optional [B,Tx,Ty] fp32 cotangent of gamma; at least one of the two
 *   grad_energies_out_dev [B,Tx,Ty] fp32 (0 outside an utterance's lengths and for an utterance without a
 *                       segmentation)
 *   workspace_dev       aligner_boundary_search_backward_workspace_bytes(...) bytes, first 256 zeroed once (status
 *                       word as above).  Five [B,Tx,Ty] fp32 planes and the hand-off ring.
 * The MAP outputs (boundaries, durations, map_score) are not differentiated.  Four launches: the normalisers, the
 * per-cell operands, the chain over the token rows from the last to the first (the search's position segments,
 * handing their first max_duration entries to the left), the gradient of every cell.  Same limits and the same
 * defined failure (ALIGNER_ST_INTERNAL, all-zero gradient for that utterance) as the search.
 */
size_t aligner_boundary_search_backward_workspace_bytes(int B, int Tx, int Ty, int max_duration);
int aligner_boundary_search_backward(const void *energies_dev, int energy_dtype,
                                     const int32_t *t_xs_dev, const int32_t *t_ys_dev, int max_duration,
                                     const float *log_alpha_dev, const float *grad_log_alpha_dev,
                                     const float *grad_gamma_dev, float *grad_energies_out_dev,
                                     void *workspace_dev, size_t workspace_bytes,
                                     int B, int Tx, int Ty, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* ALIGNER_AMD_H */
