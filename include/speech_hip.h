/*
 * speech_hip.h -- C ABI of libspeech_hip.so: gfx950 (MI355X) kernels for the fernand/speech
 * ASR hot path.  This is the drop-in boundary: plain pointers and sizes, no torch types.
 *
 * Conventions (all entry points):
 *   - return 0 on success, a negative SP_E* code on error; sp_last_error() gives the message
 *     of the calling thread's last failure;
 *   - every buffer is caller-owned DEVICE memory unless a parameter says "host";
 *   - work is enqueued on `stream` (a hipStream_t passed as void*; NULL = default stream)
 *     and is asynchronous: no hidden synchronisation, no allocation (query *_workspace_bytes
 *     and pass the workspace in);
 *   - shapes are validated on the host before any launch; a shape the kernels do not
 *     support is an error, never a silent fallback.
 *
 * The reference has no FFI of its own (it is pure Python over torch/torchaudio/sru wheels);
 * each entry point below cites the reference call site whose arithmetic it replaces.
 */
#ifndef SPEECH_HIP_H
#define SPEECH_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SP_OK 0
#define SP_EINVAL (-1)      /* bad argument / unsupported shape */
#define SP_EWORKSPACE (-2)  /* workspace too small */
#define SP_ELAUNCH (-3)     /* HIP launch/runtime error */

/* element types of activation buffers */
#define SP_F32 0
#define SP_F16 1
#define SP_BF16 2

const char* sp_last_error(void);
int sp_version(void);

/* ---------------------------------------------------------------------------------------
 * Frontend.  Replaces data.py:15-24 (torchaudio MelSpectrogram(16000, n_fft=400, hop=160,
 * n_mels=80, power=1.0) + FrequencyMasking(15) + TimeMasking(35)) and the zero padding of
 * data.py:174-175 (pad_sequence -> [B,1,80,Tmax]).
 *
 * sp_melspec_plan_bytes / sp_melspec_plan: turn a HOST filterbank [201 x 80] (row-major,
 * fb[f*80+m], what torchaudio.functional.melscale_fbanks returns) into a device "plan"
 * (twiddles, window, band-limited filterbank tiles).  Done once; the plan is caller-owned.
 *
 * sp_melspec_fwd: wav = concatenated utterances (f32 in [-1,1]; or, with flags bit2, int16 PCM read
 * as sample/32768 -- what torchaudio.load(normalize=True) gives the reference for 16-bit files, at
 * half the bytes); wav_offsets[B+1] (DEVICE int64,
 * in samples) delimits them.  Utterance b has N_b = off[b+1]-off[b] samples (N_b > 200,
 * reflect padding needs it; the caller checks) and T_b = 1 + N_b/160 frames.  out is
 * [B, 80, T_max] f32 (= [B,1,80,T_max]); frames t >= T_b are written as 0.  masks: DEVICE
 * int32 [B][4] = (f0, fw, t0, tw) or NULL; rows [f0,f0+fw) and frames [t0,t0+tw) of utterance b
 * become 0.  flags (0 = the reference's magnitude mel, the parity mode): bit0 = natural log of (mel + 1e-6);
 * bit1 (value 2) = per-utterance normalisation, applied after the log stage if that is on: every mel row of
 * every utterance is shifted and scaled to zero mean and unit (biased) variance over the utterance's T_b valid
 * frames, y = (x - mean) / sqrt(var + 1e-5); the SpecAugment bands are zeroed after the statistics were taken and
 * padding frames stay 0 (a second, light launch on the same stream).  Neither stage exists in the reference
 * (SURVEY D1); both are the optional stages BASELINE's north_star names.  bit2 (value 4) = wav is int16.
 * sp_melspec_plan copies the plan from pageable host memory and waits for that copy on `stream` before it
 * returns (its host staging buffer is reused by the next call): the one entry point that synchronises.
 * --------------------------------------------------------------------------------------- */
size_t sp_melspec_plan_bytes(void);
int sp_melspec_plan(const float* fbank_host, void* plan_dev, void* stream);
int sp_melspec_fwd(const void* wav, const int64_t* wav_offsets, int B, int T_max,
                   const void* plan_dev, const int32_t* masks, int flags, float* out,
                   void* stream);

/* ---------------------------------------------------------------------------------------
 * CTC.  Replaces train.py:69-77 / :114-122: F.log_softmax(dim=2) -> transpose ->
 * torch.nn.CTCLoss(blank=0) (reduction "mean", zero_infinity False), forward and gradient.
 *
 * logits: element (t,b,c) at logits[t*stride_t + b*stride_b + c] (f32), so both the model's
 * [B,T,V] output and the reference's [T,B,V] copy can be passed without a transpose.
 * labels int64 [B][label_stride] (padded), input_lengths/label_lengths int32 [B] (device).
 * max_label_len: host upper bound on label_lengths (sizes the lattice; <= 1023).  A label_lengths[b] above it
 * is clamped to it: that utterance is scored, normalised and differentiated against its first max_label_len
 * labels (the caller, who knows the lengths on the host, is expected to pass a true bound).
 * Outputs: nll[B] (per-utterance negative log likelihood, +inf if infeasible), loss[1] =
 * mean_b nll_b / max(L_b,1), grad (may be NULL) laid out like logits =
 * grad_scale * d loss / d logits, of grad_dtype SP_F32, SP_BF16 or SP_F16 (same strides in elements; a
 * 16-bit type when the consumer is a GEMM of that type: its rounding happens in the store); frames t >=
 * input_lengths[b] get 0.
 * Range: the lattice runs on scaled f64 probabilities, not in log space.  It agrees with torch's float64 CTC to
 * 1e-7 on scores as sharp as a mean loss of 45 bits (31 nats) per frame (tests/test_gpu_kernels.py
 * test_ctc_sharp_logits_range; a randomly initialised model sits at log2 V, a trained one below 3); scores that are
 * sharper AND wrong for tens of frames on end can push the states that carry the posterior out of f64's range below
 * the row maximum: the result then degrades (percent-level error, finally +inf) instead of staying exact.
 * Lattices of up to 1024 states are spilled to the workspace as packed four-byte states (f64's exponent, 21 mantissa
 * bits): posteriors carry 2.4e-7 of relative rounding, the likelihood none (csrc/ctc.hip).
 * Three launches on `stream` and nothing else (no second stream, no events, no allocation): the dense half of the
 * gradient shares the lattice's launch, see csrc/ctc.hip.  grad rows are written twice (dense half, then the label
 * classes' entries) -- once, by the third launch, for vocabularies of <= 64 classes (the reference's characters):
 * read grad only after the call's work on `stream` is complete.
 * --------------------------------------------------------------------------------------- */
size_t sp_ctc_workspace_bytes(int T, int B, int V, int max_label_len);
int sp_ctc_fwd_bwd(const float* logits, int64_t stride_t, int64_t stride_b, const int64_t* labels,
                   int label_stride, const int32_t* input_lengths, const int32_t* label_lengths,
                   int T, int B, int V, int blank, int max_label_len, float grad_scale,
                   float* nll, float* loss, void* grad, int grad_dtype, void* workspace,
                   size_t workspace_bytes, void* stream);

/* sp_ctc_fwd_bwd_noalloc: sp_ctc_fwd_bwd without the workspace arguments, for callers that cannot
 * allocate or size device memory at the call site (code under HIP graph capture, tight serving loops,
 * harnesses that forbid hipMalloc).  It has NO counterpart in the reference.  Same kernels, same
 * results bit for bit; the workspace is a 64 MiB buffer built into the library (one copy per GPU, held
 * from the first use of the library on that GPU until the process ends; sp_ctc_noalloc_max_bytes()
 * returns its size).  A shape whose sp_ctc_workspace_bytes exceeds it is refused with SP_EWORKSPACE
 * (configs[1], T'=751 x B=32 x V=1000 x L<=60, needs 56 MB and fits).  The call allocates nothing and
 * frees nothing; its first call on a device looks the buffer's address up (make that call outside a
 * graph capture; B = 0 is enough).  All calls on one GPU share the buffer: calls that overlap on one GPU
 * (two streams, two threads) race -- serialise them or use sp_ctc_fwd_bwd with separate workspaces.
 * Nothing in the buffer has to be zero on entry and nothing of a previous call is read. */
size_t sp_ctc_noalloc_max_bytes(void);
int sp_ctc_fwd_bwd_noalloc(const float* logits, int64_t stride_t, int64_t stride_b,
                           const int64_t* labels, int label_stride, const int32_t* input_lengths,
                           const int32_t* label_lengths, int T, int B, int V, int blank,
                           int max_label_len, float grad_scale, float* nll, float* loss, void* grad,
                           int grad_dtype, void* stream);

/* ---------------------------------------------------------------------------------------
 * Greedy CTC decode.  Replaces decoder.py:28-57: argmax over V (ties -> lowest index, NaN
 * counts as maximum, as torch.argmax), drop blanks, drop a frame equal to its predecessor.
 * scores [B][T][V] contiguous, dtype SP_F32/SP_F16/SP_BF16.  frame_lengths int32 [B] or NULL
 * (NULL = all T frames, which is what the reference does).  ids int32 [B][T]: the decoded
 * ids of utterance b occupy ids[b][0 .. lengths[b]), the rest is -1.
 * --------------------------------------------------------------------------------------- */
int sp_greedy_decode(const void* scores, int dtype, int B, int T, int V, int blank,
                     const int32_t* frame_lengths, int32_t* ids, int32_t* lengths, void* stream);

/* Host-side unit-cost edit distance on int32 sequences (what Levenshtein.distance computes
 * for decoder.py:17 and :25).  Host pointers; returns the distance (>= 0). */
int sp_edit_distance(const int32_t* a, int na, const int32_t* b, int nb);

/* ---------------------------------------------------------------------------------------
 * CTC prefix beam search with an n-gram language model.  Replaces the pyctcdecode path of
 * scripts/eval_dataset.py:20-27 (build_ctcdecoder(list(text.CHARS), <kenlm scorer>, alpha=0.3, beta=1.0))
 * and :43-51 (decode_batch(pool, logits.cpu().numpy())).  Parity unpinned: pyctcdecode and kenlm are not
 * installed where this was built; the scoring rules are the published ones (see csrc/beam.hip) and the
 * tests compare with an exhaustive enumeration of tiny lattices.
 *
 * sp_beam_candidates (GPU, asynchronous on `stream`): scores [B][T][V] contiguous logits (any of
 * SP_F32/F16/BF16; log-probabilities are logits too).  Per frame: log-softmax, clip to [log 1e-15, 0], then
 * every token with log p >= token_min_logp plus the argmax, ascending by id: cand_n[B*T] (the TRUE count,
 * which may exceed kmax: the search then refuses the frame), cand_id / cand_lp [B*T][kmax].  A frame at or
 * past frame_lengths[b] (device int32 [B] or NULL) gets count 0.  kmax >= min(V, floor(exp(-token_min_logp)) + 2)
 * can never overflow.
 *
 * sp_ngram_load_arpa (host): reads an ARPA text file; alpha / beta weight log p(word | history) (natural
 * log) and the per-word bonus, unk_score_offset (log10 units, pyctcdecode's -10) is added to a word that is
 * not a unigram and, while a word is open, to a prefix that starts no unigram; score_boundary adds <s> / </s>.
 * sp_ngram_score_sentence returns the plain log10 probability of a word sequence (for tests and tools).
 *
 * sp_ctc_beam_decode (host pointers throughout, blocking, n_threads worker threads over utterances):
 * label_codepoints[V] gives one Unicode code point per token (the blank's is ignored); `space` is the word
 * separator's id (-1: none; then lm must be NULL).  frame_lengths host int32 [B] or NULL.  Output per
 * utterance the n_best hypotheses, best first: out_ids [B][n_best][max_out] (token ids, words separated by
 * single spaces, -1 padded), out_len [B][n_best] (-1 = fewer hypotheses survived), out_scores
 * [B][n_best][2] = (log-sum-exp of the CTC alignments, that + language-model score).
 * --------------------------------------------------------------------------------------- */
int sp_beam_candidates(const void* scores, int dtype, int B, int T, int V, float token_min_logp,
                       const int32_t* frame_lengths, int kmax, int32_t* cand_n, int32_t* cand_id,
                       float* cand_lp, void* stream);
int sp_ngram_load_arpa(const char* path, float alpha, float beta, float unk_score_offset,
                       int score_boundary, void** handle);
void sp_ngram_free(void* handle);
int sp_ngram_order(const void* handle);
int sp_ngram_score_sentence(const void* handle, const char* const* words, int n, int bos, int eos,
                            double* log10_total);
int sp_ctc_beam_decode(const int32_t* cand_n, const int32_t* cand_id, const float* cand_lp, int B, int T,
                       int kmax, const int32_t* frame_lengths, const int32_t* label_codepoints, int V,
                       int blank, int space, const void* lm, int beam_width, float beam_prune_logp,
                       int n_threads, int n_best, int max_out, int32_t* out_ids, int32_t* out_len,
                       float* out_scores);

/* ---------------------------------------------------------------------------------------
 * SRU recurrence.  Replaces the `sru` package's fused recurrence kernel invoked through
 * net.py:61-72 (sru.SRU(..., bidirectional=True, layer_norm=True, rescale=False,
 * highway_bias=0)).  One launch = one layer, both directions.
 *
 * Layout (internal, chosen for coalescing; the Python wrapper permutes `weight` columns):
 *   U   [T*B][k][2][d]   gate pre-activations, k = 3 or 4 planes (u0, u1=forget, u2=reset,
 *                        u3=highway input when k == 4); dir 0 runs t = 0..T-1, dir 1 runs
 *                        t = T-1..0
 *   xres [T*B][2][d]     highway input when k == 3 (the layer's un-normalised input)
 *   wc, bias [2][2][d]   (v_f | v_r), (b_f | b_r), f32
 *   mask [B][2][d] f32 or NULL   variational dropout mask on the cell state (already / (1-p))
 *   h   [T*B][2][d]      output, act dtype;   c [T*B][2][d] cell states kept for the backward, act
 *                        dtype too (the recurrence runs in f32 registers; the sru package saves c in
 *                        the activation dtype as well).  sp_sru_fwd accepts c == NULL (inference:
 *                        nothing is kept, a third of the forward's stores)
 * dtype = element type of U, xres, h, and of gh/gU/gxres in the backward.
 * Backward: gh is d loss / d h.  Writes gU (same layout as U), gxres ([T*B][2][d], k == 3
 * only, else may be NULL), and ACCUMULATES into gwc/gbias ([2][2][d] f32, caller zeroes).
 * --------------------------------------------------------------------------------------- */
int sp_sru_fwd(const void* U, const void* xres, const float* wc, const float* bias,
               const float* mask, int T, int B, int d, int k, int dtype, void* h, void* c,
               void* stream);
int sp_sru_bwd(const void* U, const void* xres, const float* wc, const float* bias,
               const float* mask, const void* c, const void* gh, int T, int B, int d, int k,
               int dtype, void* gU, void* gxres, float* gwc, float* gbias, void* stream);

/* ---------------------------------------------------------------------------------------
 * Optimizer step over a flat fp32 arena.  Replaces train.py:81-85 (clip_grad_norm_ over all
 * parameters, then bitsandbytes Adam8bit.step()).
 * sp_sumsq_f32: out[0] = sum x_i^2; out must hold SP_SUMSQ_WORDS floats (out[1..] are per-workgroup
 * partials, summed in a fixed order: the result is bit-reproducible, so data-parallel replicas that
 * hold the same reduced gradient derive the same clip coefficient and stay identical).
 * sp_adam_step_f32: bias-corrected Adam, step >= 1.  Gradient used = clip * g / grad_div where
 * clip = min(1, max_norm / (sqrt(sumsq[0]) / |grad_div| + 1e-6)) if sumsq != NULL and max_norm > 0, else
 * 1: clip_grad_norm_ applied to g / grad_div, the gradient the update uses (data parallel: g is the sum
 * over the ranks, grad_div the world size), no host sync.  decoupled != 0: p *= 1 - lr*wd after the update
 * (bitsandbytes kernels); 0: g += wd * p (L2).
 * sp_adam8_step_f32: the same update with the moments held as bitsandbytes-style block-wise 8-bit codes
 * (train.py:254 Adam8bit): state1 / state2 uint8 [n], absmax1 / absmax2 f32 [ceil(n / SP_ADAM8_BLOCK)]; a
 * moment = code_table[state] * absmax of its block; signed dynamic-exponent table for the first moment,
 * unsigned for the second; after the update every block is re-scaled to its new absolute maximum and
 * re-coded to the nearest table entry.  Fresh state: state1 = 127 (the code of zero), state2 = 0, absmax = 0.
 * n % 4 == 0.  Parity unpinned (bitsandbytes absent; the table follows its published construction).
 * --------------------------------------------------------------------------------------- */
#define SP_SUMSQ_WORDS 1025
#define SP_ADAM8_BLOCK 2048
int sp_sumsq_f32(const float* x, int64_t n, float* out, void* stream);
int sp_adam_step_f32(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1,
                     float beta2, float eps, float weight_decay, int step, int decoupled,
                     const float* sumsq, float max_norm, float grad_div, void* stream);
int sp_adam8_step_f32(float* p, const float* g, uint8_t* state1, uint8_t* state2, float* absmax1,
                      float* absmax2, int64_t n, float lr, float beta1, float beta2, float eps,
                      float weight_decay, int step, int decoupled, const float* sumsq, float max_norm,
                      float grad_div, void* stream);

/* ---------------------------------------------------------------------------------------
 * Conv / BatchNorm / ReLU residual stack.  Replaces net.py:9-42, :56-59 (cuDNN Conv2d(1,32,3,s2),
 * 3 x [Conv2d(32,32,(3,5)) -> BatchNorm2d -> ReLU -> Conv2d -> BatchNorm2d -> += x -> ReLU]).
 * Activations are channel-last bf16 [B][F][T][32]; N = B*F*T positions.
 *
 * sp_conv_pack_w: f32 weight [32][32][3][5] -> bf16 MFMA fragment image (15*2*64*8 elements);
 *   bwd_data 1 packs the flipped, channel-transposed weight of the data gradient, 2 both images
 *   (forward at wpack[0], backward-data at wpack[15*2*64*8]) in one launch.  sp_conv_pack_w_batched: the same for
 *   L <= 16 weights in ONE launch (w / wpack are HOST arrays of L device pointers).
 * sp_conv3x5: out = conv(tf(in)) (+ bias).  in_mode 0: tf = id; 1: tf = relu(c[0:32]*in + c[32:64])
 *   (BatchNorm+ReLU of the producer, applied while staging); 2: tf = c[0:32]*in + c[32:64]*in2 +
 *   c[64:96] (BatchNorm backward from (dz, y)).  Zero padding applies after tf.  epi_mode 0: none
 *   (forward, bias added); 1: out *= [e[0:32]*epi_src + e[32:64] > 0] (ReLU mask recomputed from
 *   the pre-BN tensor); 2: out += epi_src (residual gradient).  Supported pairs: (0,0) (1,0) (2,1) (2,2).
 *   bn_sums (may be NULL): the BatchNorm reduction over the OUTPUT fused into the epilogue --
 *   epi_mode 0: what sp_bn_stats(out) gives; epi_mode 1: what sp_bn_bwd_reduce(out, epi_src) gives.
 *   Every `sums` / `bn_sums` buffer holds SP_BN_SUMS_WORDS floats: eight rows of 64, then one 64-float
 *   partial per workgroup of the producing launch (at most 1024), then one word with the number of
 *   partials (an int).  The producers write the partials and the count; sp_bn_finalize_* adds the
 *   partials into the eight rows in a fixed order (and leaves the rows in the buffer: their sum is
 *   the result) before it uses them.  No float atomics: statistics, and with them every bf16
 *   activation, are bit-reproducible.
 * sp_conv3x5_wgrad: gw[32][32][3][5] += sum_pos dy[pos][co] * a[pos + tap][ci], with
 *   a = a_src (a_mode 0) or relu(BN(a_src)) (a_mode 1), dy = g_coef affine of (g, y) as in_mode 2.
 *   gwp_ws: workspace of SP_CONV_WGRAD_WS_WORDS floats (one 15*32*32 slab per persistent workgroup, at most
 *   SP_CONV_WGRAD_SLABS).  Nothing in it has to be zero: every slab that is read has been written by the same
 *   call, and the slabs are added in a fixed order (no float atomics: gw is bit-reproducible).
 * sp_conv_stem_fwd/bwd: Conv2d(1,32,3,stride 2,pad 1) on the f32 spectrogram [B][Fin][Tin] ->
 *   [B][F][T][32], F = (Fin-1)/2+1, T = (Tin-1)/2+1; bwd accumulates gw[32*9], gbias[32].
 * sp_bn_stats: (sum y | sum y^2) per channel, as per-workgroup partials in sums (see above).
 * sp_bn_bwd_reduce: sums = (sum g, sum g*y).
 * sp_bn_finalize_fwd: coef = (scale, shift), save = (mean, invstd); training != 0 uses the batch
 *   sums and updates running stats (momentum, unbiased variance), else uses running stats.
 * sp_bn_finalize_bwd: coef3 (in_mode 2 coefficients) from the backward sums; ggamma/gbeta accumulate.
 * sp_block_out: z = relu(coef-affine(y2) + x); z granule address = b*zsb + f*zsf + t*zst (elements),
 *   which lets the last block write the [T][B][F*32] layout the SRU stack reads.
 * sp_block_bwd_in: gpre = gz * [z > 0] (gz, z in that layout); sums = (sum gpre, sum gpre*y2).
 * --------------------------------------------------------------------------------------- */
#define SP_BN_SUMS_WORDS (8 * 64 + 1024 * 64 + 64)
#define SP_CONV_WGRAD_SLABS 512
#define SP_CONV_WGRAD_WS_WORDS (SP_CONV_WGRAD_SLABS * 15 * 32 * 32)
int sp_conv_pack_w(const float* w, int bwd_data, void* wpack, void* stream);
int sp_conv_pack_w_batched(const float* const* w, int L, int bwd_data, void* const* wpack, void* stream);
int sp_conv3x5(const void* in, const void* in2, const float* coef, int in_mode, const void* wpack,
               const float* bias, const void* epi_src, const float* epi_coef, int epi_mode, int B,
               int F, int T, void* out, float* bn_sums, void* stream);
int sp_conv3x5_wgrad(const void* a_src, const float* a_coef, int a_mode, const void* g,
                     const void* y, const float* g_coef, int B, int F, int T, float* gwp_ws,
                     float* gw, void* stream);
int sp_conv_stem_fwd(const float* spec, const float* w, const float* bias, int B, int Fin, int Tin,
                     void* out, void* stream);
int sp_conv_stem_bwd(const float* spec, const void* gout, int B, int Fin, int Tin, float* gw,
                     float* gbias, void* stream);
int sp_bn_stats(const void* y, int64_t npos, float* sums, void* stream);
int sp_bn_bwd_reduce(const void* g, const void* y, int64_t npos, float* sums, void* stream);
int sp_bn_finalize_fwd(float* sums, int64_t npos, const float* gamma, const float* beta,
                       float* rmean, float* rvar, float momentum, float eps, int training,
                       float* coef, float* save, void* stream);
int sp_bn_finalize_bwd(float* sums, const float* save, const float* gamma, int64_t npos,
                       float* coef3, float* ggamma, float* gbeta, void* stream);
int sp_block_out(const void* y2, const float* coef, const void* x, int B, int F, int T, void* z,
                 int64_t zsb, int64_t zsf, int64_t zst, void* stream);
int sp_block_bwd_in(const void* gz, const void* z, int64_t zsb, int64_t zsf, int64_t zst,
                    const void* y2, int B, int F, int T, void* gpre, float* sums, void* stream);

/* The conv stack in fp32, for EVALUATION outside autocast (train.py:105-135 and :162-168 run net.py:56-59 and
 * :6-42 in fp32; scripts/predict.py:40-47 on the CPU): one f32 convolution with the inference BatchNorm folded in.
 *   out(b, f, t, co) = act( scale[co] * sum_{kh, kw, ci} x(b, f*stride + kh - KH/2, t*stride + kw - KW/2, ci) * w[kh][kw][ci][co]
 *                           + shift[co] + res(b, f, t, co) )
 * x: f32, element (b, f, t, ci) at x[b*xsb + f*xsf + t*xst + ci], Cin = 1 (the spectrogram, net.py:56) or 32;
 * w: f32 [KH][KW][Cin][32] (the state_dict's [32][Cin][KH][KW] permuted by the caller); zero padding of KH/2, KW/2
 * (odd kernels); stride 1 or 2; scale / shift: f32 [32] -- (1, bias) for a plain convolution, (gamma / sqrt(var +
 * eps), (bias - mean) * that + beta) for conv + BatchNorm on running statistics; res: NULL or f32 with its own strides
 * (the residual of net.py:38-40); relu != 0: act = max(., 0).  out / res: element (b, f, t, co) at [b*osb + f*osf +
 * t*ost + co] (so the last block can write [T'][B][F*32] directly, net.py:81-83); Fout = (Fin - 1) / stride + 1.
 * f32 operands and f32 accumulation in one fixed order per output (vector ALUs, no MFMA): results do not depend on
 * the launch.  Pointers 16-byte aligned, strides multiples of 4 elements (x: only when Cin = 32). */
int sp_conv2d_f32(const float* x, int64_t xsb, int64_t xsf, int64_t xst, int Cin, const float* w, const float* scale,
                  const float* shift, const float* res, int64_t rsb, int64_t rsf, int64_t rst, int relu, int B,
                  int Fin, int Tin, int KH, int KW, int stride, float* out, int64_t osb, int64_t osf, int64_t ost,
                  void* stream);

/* Projection-weight layout conversion around the SRU projection GEMM (net.py:61-72; the sru
 * package stores weight[n_in][dir][d][k]).  sp_sru_pack_w: f32 state_dict weight -> Wk[n_in][k][dir][d]
 * in dtype (SP_F32 / SP_BF16), rows re-ordered from c*F + f to f*C + c (C = 1: unchanged; F = n_in/C)
 * -- the order the conv stack hands its features over in (net.py:81-83).  sp_sru_unpack_dw: the
 * inverse for the gradient, summing S split-K partials [S][n_in][k][dir][d] of dtype into f32
 * dweight[n_in][dir][d][k] (overwritten, or added to when accumulate != 0: the caller's gradient
 * buffer).  sp_sru_pack_wt: the same weight transposed, WkT[k][dir][d][n_in] with row pitch ldo elements
 * (>= n_in, even): the K-contiguous B operand of sp_gemm for the forward projection; with out_w
 * != NULL the same pass also writes sp_sru_pack_w's Wk there (what the data gradient's GEMM reads).
 * n_in % 64 == 0 (bf16; 32 for f32), d % 64 == 0.  sp_sru_pack_wt_batched: the same for L <= 16 layers of
 * one shape in ONE launch; weights / out_t / out_w are HOST arrays of L device pointers (out_w NULL: none;
 * weight pointers 16-byte aligned). */
int sp_sru_pack_w(const float* weight, int n_in, int d, int k, int C, int dtype, void* out, void* stream);
int sp_sru_pack_wt(const float* weight, int n_in, int d, int k, int C, int dtype, void* out_t, int64_t ldo,
                   void* out_w, void* stream);
int sp_sru_pack_wt_batched(const float* const* weights, int L, int n_in, int d, int k, int C, int dtype,
                           void* const* out_t, int64_t ldo, void* const* out_w, void* stream);
int sp_sru_unpack_dw(const void* partials, int S, int n_in, int d, int k, int C, int dtype,
                     int accumulate, float* dweight, void* stream);

/* ---------------------------------------------------------------------------------------
 * Row LayerNorm.  Replaces the layer norms of net.py:61-76 (sru.SRU(layer_norm=True) on every
 * layer input; nn.LayerNorm before the classifier).  x, y, gy, gx, gres: [M][N] of dtype
 * (SP_F32 / SP_BF16), N % 8 == 0, N <= 2048; gamma, beta, mean, rstd, dgamma, dbeta f32.
 * Backward: gx = LN'(gy) (+ gres if not NULL: the SRU highway's residual gradient);
 * dgamma / dbeta are ACCUMULATED (caller zeroes).  ldy: row pitch of y in elements (>= N, multiple
 * of 8; N for a dense output) -- a pitch that is not a multiple of 1024 bytes keeps the GEMM that
 * reads y off the L2-channel conflicts of power-of-two row strides (measured 11 % on the forward projection).
 * --------------------------------------------------------------------------------------- */
int sp_layernorm_fwd(const void* x, const float* gamma, const float* beta, int64_t M, int N, float eps,
                     int dtype, void* y, int64_t ldy, float* mean, float* rstd, void* stream);
int sp_layernorm_bwd(const void* gy, const void* x, const float* mean, const float* rstd,
                     const float* gamma, const void* gres, int64_t M, int N, int dtype, void* gx,
                     float* dgamma, float* dbeta, void* stream);

/* ---------------------------------------------------------------------------------------
 * General GEMM on the matrix cores, f32 accumulation:
 *     C[m][n] (+)= sum_k A(m,k) B(n,k),   A(m,k) = A[m*sam + k*sak],   B(n,k) = B[n*sbn + k*sbk]
 * Replaces every matmul the reference reaches through net.py:61-76: sru.SRU's `x.mm(weight)` and the
 * two gradient matmuls autograd derives from it, and the classifier's nn.Linear(1024, n_vocab+1,
 * bias=False) with its gradients.  Each operand must be contiguous along k (sak / sbk == 1) or along
 * its row index (sam / sbn == 1), which covers the NT, NN, TN and TT forms; strides in elements.
 * in_dtype SP_BF16 / SP_F16 (v_mfma_f32_32x32x16 / 16x16x32) or SP_F32 (v_mfma_f32_32x32x2_f32, exact f32);
 * out_dtype SP_F32 / SP_BF16 / SP_F16, row pitch ldc >= N.  accumulate != 0 (f32 output only): C += the
 * product.  split_k = S > 1: K is cut into S ranges and range z writes its partial product to
 * C + z * c_split_stride elements (no atomics: sum the slabs with sp_sum_slabs_f32 or sp_sru_unpack_dw).
 * Any M, N, K >= 0.  16-byte operand loads are used when the bases are 16-byte aligned and the non-unit
 * strides are multiples of 8 (16-bit) or 4 (f32) elements, element-wise loads otherwise.  Large bf16
 * problems with both operands K-contiguous (and, for the weight gradient, both row-contiguous) run on
 * the persistent LDS-DMA kernels of csrc/gemm.hip; which kernel runs is decided by the shape alone.
 *
 * sp_gemm_tn_splits: the split_k this library would choose for a weight-gradient product (A and B both
 * row-contiguous, K = M, N, K as passed to sp_gemm): enough K ranges to give every CU a 256 x 256 tile.
 *
 * sp_sum_slabs_f32: out[i] (+)= scale * sum_s partials[s * slab_stride + i], i < n (n, slab_stride
 * multiples of 4; scale_dev: device float or NULL = 1) -- the reduction of split-K slabs straight into a
 * gradient buffer, with the incoming loss gradient (a device scalar) folded in.
 *
 * sp_linear_pack: the classifier's f32 weight [V][n] (net.py:73-76, nn.Linear(1024, n_vocab+1, bias=False)) ->
 * its 16-bit GEMM operands in one launch: out_w [Vp][n] (rows V..Vp zero; Vp >= V, e.g. V rounded up to the 256
 * columns the persistent kernels tile by) and / or out_t [n][Vp] = scale * W^T (scale_dev: device float or
 * NULL = 1 -- the loss gradient that reaches the data-gradient product).  Either output may be NULL.
 * --------------------------------------------------------------------------------------- */
int sp_gemm(const void* A, const void* B, void* C, int M, int N, int K, int64_t sam, int64_t sak,
            int64_t sbn, int64_t sbk, int64_t ldc, int in_dtype, int out_dtype, int accumulate,
            int split_k, int64_t c_split_stride, void* stream);
int sp_gemm_tn_splits(int M, int N, int K, int in_dtype);
int sp_sum_slabs_f32(const float* partials, int S, int64_t n, int64_t slab_stride, float* out,
                     int accumulate, const float* scale_dev, void* stream);
int sp_linear_pack(const float* weight, int V, int n, int Vp, const float* scale_dev, int dtype, void* out_w,
                   void* out_t, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SPEECH_HIP_H */
