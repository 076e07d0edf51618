/* libcvae_hip: C ABI of the MI355X (gfx950) hot path of Coupled-VAE.
 *
 * The reference has no FFI: its hot path is Python on torch.  The entry
 * points below are what a binding for that path binds: one call per reference function on
 * the path (forward and, for training, its backward), citing the reference code replaced.
 * CV = /root/reference/Coupled-VAE, CC = /root/reference/Coupled-CVAE.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer unless named host_*; float = fp32, ids = int64.
 *  - row-major, contiguous tensors; sequences are time-major inside the library: (T, B, .).
 *  - `stream` is a hipStream_t (NULL = default stream); all work is enqueued and, in the steady state, nothing
 *    synchronises or allocates.  The exception is warm-up: the FIRST call on a (device, stream) that needs one of
 *    the library-owned buffers listed below, or a call with a larger shape than any before, hipMallocs it
 *    (hipMemset; on regrowth hipFree, which synchronises the device).  cvae_release_stream / cvae_shutdown free them.
 *  - return 0 on success, negative on error; cvae_last_error() returns the message
 *    (thread-local).  Shape / pointer preconditions are checked on the host before launch.
 *  - workspaces are passed in by the caller.  Exceptions owned by the library, one set per (device, stream),
 *    allocated at their first use and regrown only when a larger shape arrives (warm-up; steady-state calls
 *    allocate nothing): the whole-sequence recurrent kernels' sync words and partial-gradient exchange (armed with
 *    a sentinel word by a fill launch before every backward recurrence), the
 *    split-K products' partial-sum slabs, the live-row index lists of the backward passes, the weight images of the
 *    weight-streaming recurrent step kernels, the clip norm's per-block partial sums and (cvae_deterministic only)
 *    the fixed-point buffer of the embedding-gradient scatter.
 *  - cell: 0 = GRU (gate rows r,z,n), 1 = LSTM (i,f,g,o), PyTorch equations and layouts.
 *    G = 3 or 4 below.
 */
#ifndef CVAE_HIP_H
#define CVAE_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* 100 for a product build; negative for a timing-only diagnostic build (an *_EXP_* switch removed work from a kernel:
 * results are wrong, csrc/common.hpp) -- bindings refuse those. */
int cvae_version(void);
const char* cvae_last_error(void);
/* Checks that `device` is a gfx950 part and makes it current. */
int cvae_init(int device);

/* ---- a1: batch layout.  dataloaders/ptb.py:118-129 (process_raw_sent) + the truncation
 * to the batch maximum of Experiment.py:659-664.  raw_flat/raw_offsets: CSR token ids of B
 * sentences.  Sentences are cut at max_len; rows are right-padded with `pad` to `width`
 * (bare) / width+1 (go, eos).  width = max_len for the un-truncated eval layout, or
 * min(max_len, longest sentence) for the training layout. */
int cvae_pack_batch(const int64_t* raw_flat, const int64_t* raw_offsets, int B, int max_len, int width,
                    int64_t pad, int64_t go, int64_t eos, int64_t* bare, int64_t* go_out, int64_t* eos_out,
                    int64_t* len, void* stream);
/* ---- f2: the same layout for sentences sel[0..B) of a corpus whose CSR arrays are resident on the
 * device (replaces the reference's DataLoader workers + per-batch host tensors, Experiment.py:641):
 * per batch only the B indices cross PCIe.  Indices must be within the corpus (not checked). */
int cvae_pack_batch_indexed(const int64_t* corpus_flat, const int64_t* corpus_offsets, const int64_t* sel, int B,
                            int max_len, int width, int64_t pad, int64_t go, int64_t eos, int64_t* bare,
                            int64_t* go_out, int64_t* eos_out, int64_t* len, void* stream);
/* ---- a2: utils/utils.py:18-21 (sort_resort): stable descending order and its inverse. */
int cvae_sort_resort(const int64_t* len, int B, int64_t* s_idx, int64_t* res_idx, void* stream);

/* ---- generic fp32 MFMA product  C = act(op(A) op(B) + bias) (+ C);  nn.Linear / F.linear.
 * transA=0: A[M,K] (lda) ; 1: A[K,M].  transB=0: B[K,N] (ldb) ; 1: B[N,K].
 * act: 0 none, 1 tanh, 2 relu, 3 exp.  beta: 0 overwrite, 1 accumulate. */
int cvae_gemm(int transA, int transB, int M, int N, int K, const float* A, int64_t lda, const float* B,
              int64_t ldb, float* C, int64_t ldc, const float* bias, int act, int beta, void* stream);
/* y = act(x W^T + b): Gaussian heads (vae_couple.py:798-799), toInit (:429-434), PostRepr
 * (CC vae_couple.py:263).  x[M,K] W[N,K] b[N] y[M,N]. */
int cvae_linear_forward(const float* x, const float* W, const float* b, int act, int M, int N, int K, float* y,
                        void* stream);
/* Backward of the above.  dy[M,N] is upstream; y is the forward output (needed for act != 0).
 * dx[M,K] (nullable) is overwritten, dW[N,K] and db[N] are overwritten.  scratch: M*N floats. */
int cvae_linear_backward(const float* x, const float* W, const float* y, const float* dy, int act, int M, int N,
                         int K, float* dx, float* dW, float* db, float* scratch, void* stream);

/* ---- a3: Encoder.forward (Modules/encoder.py:41-70), bidirectional 1-layer GRU/LSTM over
 * packed sequences, returning only what callers consume: last[B,2H] = [h_fwd | h_bwd].
 * ids[B,L]; len[B]; emb[V,E]; mask: NULL (eval) or the dropout mask (L,B,E) with values
 * {0, 1/(1-p)}; mask_rows: NULL or int64[B], column of `mask` that batch row b uses (the
 * reference draws the mask in length-sorted order: pass res_idx of cvae_sort_resort).
 * w[8] = {W_ih, W_hh, b_ih, b_hh, W_ih_reverse, W_hh_reverse, b_ih_reverse, b_hh_reverse}.
 * Saved for backward (all caller-allocated): X[L,B,E], xproj[2][L,B,G*H], hs[2][(L+1),B,H],
 * cs[2][(L+1),B,H] (LSTM, else NULL), gates[2][L,B,4H] (NULL in eval: nothing saved). */
int cvae_encoder_forward(int cell, int B, int L, int E, int H, int64_t V, const int64_t* ids, const int64_t* len,
                         const float* emb, const float* mask, const int64_t* mask_rows, const float* const* w,
                         float* X, float* xproj, float* hs, float* cs, float* gates, float* last, void* stream);
/* dlast[B,2H] -> dw[8] (same order as w, overwritten), demb[V,E] (ACCUMULATED into).
 * scratch: floats, at least cvae_encoder_backward_scratch(cell,B,L,E,H). */
int64_t cvae_encoder_backward_scratch(int cell, int B, int L, int E, int H);
int cvae_encoder_backward(int cell, int B, int L, int E, int H, int64_t V, const int64_t* ids, const int64_t* len,
                          const float* mask, const int64_t* mask_rows, const float* const* w, const float* X,
                          const float* hs, const float* cs, const float* gates, const float* dlast,
                          float* const* dw, float* demb, float* scratch, void* stream);
/* ne (1 or 2) encoders over the SAME embedding, cell, B, L, E, H in the same launches: the dialogue
 * model's post and response encoders (CC/Modules/cvae_couple.py forward: PostEncoder then
 * RespEncoder, CC/Modules/encoder.py:41-70 each).  The caller pads the shorter batch's ids (any valid
 * id) and mask (ones) to the common L; steps past len[b] are carries, so results equal the separate
 * calls.  Every pointer argument is a host array of ne device pointers laid out as in the
 * single-encoder call (entries of mask, mask_rows, cs, gates may be NULL as there); w / dw hold
 * 8 pointers per encoder (w[8*e + i]); scratch[e] as for cvae_encoder_backward. */
int cvae_encoder_forward_n(int ne, int cell, int B, int L, int E, int H, int64_t V, const int64_t* const* ids,
                           const int64_t* const* len, const float* emb, const float* const* mask,
                           const int64_t* const* mask_rows, const float* const* w, float* const* X,
                           float* const* xproj, float* const* hs, float* const* cs, float* const* gates,
                           float* const* last, void* stream);
int cvae_encoder_backward_n(int ne, int cell, int B, int L, int E, int H, int64_t V, const int64_t* const* ids,
                            const int64_t* const* len, const float* const* mask, const int64_t* const* mask_rows,
                            const float* const* w, const float* const* X, const float* const* hs,
                            const float* const* cs, const float* const* gates, const float* const* dlast,
                            float* const* dw, float* demb, float* const* scratch, void* stream);

/* ---- a7: Decoder.forward, teacher-forcing branch (Modules/vae_couple.py:86-115,
 * Modules/decoder.py:85-107) up to final = tanh(W_c drop(h_t)) -- the vocab projection is
 * fused into the loss (cvae_projce_*).  R rows = B (training) or n_sample*B (IW eval);
 * init[R,E]; go[B,T] ids shared by the R/B row groups (row r decodes sentence r % B).
 * mask_in (T,R,2E) / mask_out (T,R,Hd): NULL in eval.  w[6] = {W_ih[G*Hd,2E], W_hh, b_ih,
 * b_hh, W_c[E,Hd], b_c[E]}.  Step 0 of the internal (T+1)-step sequence is the prefix step
 * h_0 = RNN([init,init], 0) (:96-103).
 * train != 0: saves X[(T+1),R,2E], xproj[(T+1),R,G*Hd], hs[(T+2),R,Hd], cs (LSTM), gates
 *   [(T+1),R,4Hd], hdrop[T,R,Hd]; requires R == B.
 * train == 0: exact eval dedup (SURVEY 7 "IW eval savings"): the emb(go_t) half of the input
 *   projection is computed once per sentence and the init half once per row; X is then
 *   [T,B,E] scratch, xproj [T,B,G*Hd], xadd[2,R,G*Hd], hp[R,G*Hd] scratch (R > 256),
 *   gates NULL.
 * final[T,R,E] out. */
int cvae_decoder_forward(int cell, int train, int B, int R, int T, int E, int Hd, int64_t V, const float* init,
                         const int64_t* go, const float* emb, const float* mask_in, const float* mask_out,
                         const float* const* w, float* X, float* xproj, float* xadd, float* hp, float* hs,
                         float* cs, float* gates, float* hdrop, float* final, void* stream);
/* IW-eval variant of the train == 0 branch for the importance-sampling loop
 * (Modules/vae_couple.py:526-578): rows are sentence-major, r = b*K + k for sample k of
 * sentence b, and the caller has ordered the B sentences by non-increasing target length, so
 * the rows with a non-<pad> target at step t are the prefix [0, live_sent[t]*K).  Each step
 * computes that prefix only; rows of hs/cs/final past it are left untouched (their targets are
 * <pad> and cvae_projce_forward never reads them).  Per live row the arithmetic is that of
 * cvae_decoder_forward(train=0).  live_sent: HOST int[T], non-increasing, within [0,B].
 * Requires B*K > 256 (smaller row counts take cvae_decoder_forward).  Buffers as in the
 * train == 0 branch with R = B*K; hp is required. */
int cvae_decoder_forward_sorted(int cell, int B, int K, int T, int E, int Hd, int64_t V, const float* init,
                                const int64_t* go, const float* emb, const int* live_sent, const float* const* w,
                                float* X, float* xproj, float* xadd, float* hp, float* hs, float* cs, float* final,
                                void* stream);
/* ---- f4 (sibling variant lm, Modules/lm.py:78-100): the training branch of the decoder without the
 * prefix step (h_0 = 0).  The LM's input has no conditioning half: the caller passes init = 0 and
 * W_ih = [W_ih_lm | 0] ([G*Hd, 2E]); buffers and arguments as cvae_decoder_forward(train=1) /
 * cvae_decoder_backward (step 0's slots are zero-filled and unused; dinit comes back zero). */
int cvae_decoder_forward_noprefix(int cell, int B, int T, int E, int Hd, int64_t V, const float* init,
                                  const int64_t* go, const float* emb, const float* mask_in, const float* mask_out,
                                  const float* const* w, float* X, float* xproj, float* hs, float* cs, float* gates,
                                  float* hdrop, float* final, void* stream);
int cvae_decoder_backward_noprefix(int cell, int B, int T, int E, int Hd, int64_t V, const int64_t* go,
                                   const float* mask_in, const float* mask_out, const float* const* w, const float* X,
                                   const float* hs, const float* cs, const float* gates, const float* hdrop,
                                   const float* final, const float* dfinal, float* dinit, float* const* dw,
                                   float* demb, float* scratch, void* stream);
/* ---- f3: greedy branch of Decoder.forward (Modules/decoder.py:176-201, Modules/vae_couple.py:
 * 173-201; used by sample_from_prior / decode_from / gen_interps).  T = max_len steps, no early stop
 * (as the reference): y_t = argmax_v ToVocab(tanh(W_c h_t))[v] with entry mask_id set to -inf first
 * (the reference masks <unk> = 3; pass -1 for no mask), ties to the lowest id.  first[B]: the <go>
 * ids (helper[:, 0]).  out_ids[B,T] int64.  ws: cvae_decoder_greedy_scratch(...) floats, 16-byte
 * aligned.  B <= 256.  Dropout is off (eval mode), as in every caller of this branch. */
int64_t cvae_decoder_greedy_scratch(int cell, int B, int T, int E, int Hd, int64_t V);
int cvae_decoder_greedy(int cell, int B, int T, int E, int Hd, int64_t V, const float* init, const int64_t* first,
                        const float* emb, const float* const* w, const float* W_v, const float* b_v, int mask_id,
                        float* ws, int64_t* out_ids, void* stream);
/* ---- f3: beam branch of Decoder.forward (Modules/decoder.py:114-168, Modules/vae_couple.py:122-172;
 * taken when config.beam_size > 1).  K beams per sentence, T = max_len steps, no early stop.  Each
 * step: log_softmax of the masked logits (entry mask_id at -inf), the K best tokens of every live beam
 * (torch.topk; ties to the lower id), then per sentence the K cheapest of the ns*K candidates by
 * accumulated negative log-likelihood, ties in candidate order (source beam, then rank) as Python's
 * stable sort gives (decoder.py:161).  out_ids[B,T]: beam[0].sent; out_nll[B] (may be NULL): its cost.
 * 2 <= K <= 8, K*B <= 256.  ws: cvae_decoder_beam_scratch(...) floats, 16-byte aligned.  The
 * reference's branch fails for the LSTM (state indexed as one tensor); here (h, c) are carried alike. */
int64_t cvae_decoder_beam_scratch(int cell, int B, int K, int T, int E, int Hd, int64_t V);
int cvae_decoder_beam(int cell, int B, int K, int T, int E, int Hd, int64_t V, const float* init, const int64_t* first,
                      const float* emb, const float* const* w, const float* W_v, const float* b_v, int mask_id,
                      float* ws, int64_t* out_ids, float* out_nll, void* stream);
int64_t cvae_decoder_backward_scratch(int cell, int B, int T, int E, int Hd);
/* dfinal[T,B,E] -> dinit[B,E] (overwritten), dw[6] (overwritten), demb[V,E] (ACCUMULATED). */
int cvae_decoder_backward(int cell, int B, int T, int E, int Hd, int64_t V, const int64_t* go,
                          const float* mask_in, const float* mask_out, const float* const* w, const float* X,
                          const float* hs, const float* cs, const float* gates, const float* hdrop,
                          const float* final, const float* dfinal, float* dinit, float* const* dw, float* demb,
                          float* scratch, void* stream);

/* The same for nd <= 4 decoders of identical shape run in lock-step (training only, R == B):
 * the stochastic Decoder and the deterministic DecoderCouple of VAE_COUPLE.forward
 * (Modules/vae_couple.py:733-739) are independent recurrences, so each timestep of both is
 * ONE launch.  Every per-decoder argument becomes an array of nd pointers; w and dw hold
 * 6*nd entries (decoder d at [6d, 6d+6)); scratch[d]: cvae_decoder_backward_scratch floats. */
int cvae_decoder_forward_n(int nd, int cell, int B, int T, int E, int Hd, int64_t V, const float* const* init,
                           const int64_t* go, const float* emb, const float* const* mask_in,
                           const float* const* mask_out, const float* const* w, float* const* X,
                           float* const* xproj, float* const* hs, float* const* cs, float* const* gates,
                           float* const* hdrop, float* const* final, void* stream);
int cvae_decoder_backward_n(int nd, int cell, int B, int T, int E, int Hd, int64_t V, const int64_t* go,
                            const float* const* mask_in, const float* const* mask_out, const float* const* w,
                            const float* const* X, const float* const* hs, const float* const* cs,
                            const float* const* gates, const float* const* hdrop, const float* const* final,
                            const float* const* dfinal, float* const* dinit, float* const* dw, float* demb,
                            float* const* scratch, void* stream);

/* ---- a7 tail + a8: ToVocab + SeqLoss(keep_batch=True) (Modules/vae_couple.py:119,
 * Modules/Losses/SeqLoss.py:36-44), fused: logits are never materialised, and rows whose target
 * is `pad` (loss 0, gradient 0 in the reference) are never multiplied.
 * final[nseg][T*R, E]: nseg stacked decoder outputs that share the targets (the decoder and its
 * coupled twin, vae_couple.py:371-376), rows time-major n = t*R + r inside a segment;
 * W[V,E], b[V], tgt[T*R] (time-major targets)
 * -> loss_row[nseg*T*R] (0 where tgt == pad), lse[nseg*T*R], sent[nseg*R] = sum over t.
 * ws: cvae_projce_workspace(nseg*T*R, E, V) floats. */
int64_t cvae_projce_workspace(int64_t N, int64_t E, int64_t V);

int cvae_projce_forward(const float* final, const float* W, const float* b, const int64_t* tgt, int64_t pad,
                        int nseg, int T, int R, int E, int64_t V, float* loss_row, float* lse, float* sent,
                        float* ws, void* stream);
/* gsent[nseg*R] = dL/dsent.  dfinal[nseg*T*R,E], dW[V,E], db[V] are overwritten.
 * ws: cvae_projce_backward_workspace(nseg*T*R, E, V) floats. */
int64_t cvae_projce_backward_workspace(int64_t N, int64_t E, int64_t V);
int cvae_projce_backward(const float* final, const float* W, const float* b, const int64_t* tgt, int64_t pad,
                         const float* lse, const float* gsent, int nseg, int T, int R, int E, int64_t V,
                         float* dfinal, float* dW, float* db, float* ws, void* stream);
/* Training-step variant that does not recompute the projection in backward (E <= 200):
 * cvae_projce_forward_train also stores the logits of the live (non-pad) rows into `logits`
 * (cvae_projce_logits_workspace(N, E, V) floats, 16-byte aligned; 0 = not supported for this E), and
 * cvae_projce_backward_saved turns them into d loss / d logits IN PLACE (the buffer is consumed) before
 * the same two products as cvae_projce_backward.  Results equal the plain pair's. */
int64_t cvae_projce_logits_workspace(int64_t N, int64_t E, int64_t V);
int cvae_projce_forward_train(const float* final, const float* W, const float* b, const int64_t* tgt, int64_t pad,
                              int nseg, int T, int R, int E, int64_t V, float* loss_row, float* lse, float* sent,
                              float* ws, float* logits, void* stream);
int cvae_projce_backward_saved(const float* final, const float* W, const float* b, const int64_t* tgt, int64_t pad,
                               const float* lse, const float* gsent, int nseg, int T, int R, int E, int64_t V,
                               float* dfinal, float* dW, float* db, float* ws, float* logits, void* stream);
/* The same pair for nd <= 4 decoders of identical shape over ONE set of targets -- Decoder and DecoderCouple of a
 * training step, whose losses the reference takes one after the other on the same `eos`
 * (Coupled-VAE/Experiment.py:684-702; Modules/Losses/SeqLoss.py).  Arguments are arrays of nd pointers (ws[d]:
 * cvae_projce_workspace(T*R, E, V) floats each; logits[d]: cvae_projce_logits_workspace floats each).  The live-row
 * list is built once and kept behind logits[0] for the backward call, which zero-fills every gradient in one launch
 * and runs the decoders' products of one kind in one launch where they are small enough to share the grid
 * (ws there: cvae_projce_backward_saved_n_workspace(E) floats, shared).  Results equal nd separate calls'. */
int cvae_projce_forward_train_n(int nd, const float* const* final, const float* const* W, const float* const* b,
                                const int64_t* tgt, int64_t pad, int T, int R, int E, int64_t V,
                                float* const* loss_row, float* const* lse, float* const* sent, float* const* ws,
                                float* const* logits, void* stream);
int64_t cvae_projce_backward_saved_n_workspace(int64_t E);
int cvae_projce_backward_saved_n(int nd, const float* const* final, const float* const* W, const float* const* b,
                                 const int64_t* tgt, int64_t pad, const float* const* lse, const float* const* gsent,
                                 int T, int R, int E, int64_t V, float* const* dfinal, float* const* dW,
                                 float* const* db, float* ws, float* const* logits, void* stream);
/* tgt_tm[t*R + r] = eos[(r % B)*T + t]: time-major targets for R = n*B rows. */
int cvae_targets_time_major(const int64_t* eos, int B, int T, int R, int64_t* tgt_tm, void* stream);
/* tgt_tm[t*B*K + r] = eos[(r / K)*T + t]: the same for sentence-major rows r = b*K + k. */
int cvae_targets_sentence_major(const int64_t* eos, int B, int T, int K, int64_t* tgt_tm, void* stream);
/* Materialise logits[B,T,V] (batch-major, as the reference returns them) for callers that
 * index them; not used on the hot path. */
int cvae_logits(const float* final, const float* W, const float* b, int T, int B, int E, int64_t V, float* tmp,
                float* logits, void* stream);

/* ---- a5: rsample (Modules/vae_couple.py:801): z[n,b,:] = mu[b,:] + sigma[b,:]*eps[n,b,:]. */
int cvae_reparam_forward(const float* mu, const float* sigma, const float* eps, int n, int B, int Z, float* z,
                         void* stream);
/* dz[B,Z] (n = 1) -> dmu += dz, dsigma += dz*eps (both ACCUMULATED). */
int cvae_reparam_backward(const float* dz, const float* eps, int B, int Z, float* dmu, float* dsigma,
                          void* stream);
/* ---- a9: kl_divergence(N(mu,sigma) || N(mu_p,sigma_p)).sum(1) (Experiment.py:733-735;
 * CC Experiment.py:398-399).  mu_p = sigma_p = NULL means the standard normal. -> kl[B]. */
int cvae_kl_forward(const float* mu, const float* sigma, const float* mu_p, const float* sigma_p, int B, int Z,
                    float* kl, void* stream);
/* gkl[B] upstream -> dmu, dsigma (overwritten), dmu_p, dsigma_p (overwritten, nullable). */
int cvae_kl_backward(const float* mu, const float* sigma, const float* mu_p, const float* sigma_p,
                     const float* gkl, int B, int Z, float* dmu, float* dsigma, float* dmu_p, float* dsigma_p,
                     void* stream);
/* ---- f4 (sibling variant surprising-fix*, Experiment.py:1137-1142): free bits,
 * kl[b] = sum_j max(KL(N(mu,sigma) || N(0,1))_j, floor), floor = lambda_fb / Z; the gradient passes
 * where the element is >= floor (torch.clamp). */
int cvae_kl_fb_forward(const float* mu, const float* sigma, int B, int Z, float floor, float* kl, void* stream);
int cvae_kl_fb_backward(const float* mu, const float* sigma, const float* gkl, int B, int Z, float floor,
                        float* dmu, float* dsigma, void* stream);
/* ---- a10: euclidean + compute_imq_dist_couple (Experiment.py:742-751,1892-1923).
 * euclid != 0: loss[b] = d ; else loss[b] = -sum_s C_s/(C_s+d), C_s = 2*Z*s.  d[b] also out. */
int cvae_couple_forward(const float* init, const float* init_c, int B, int E, int Z, int euclid, float* d,
                        float* loss, void* stream);
/* gloss[B] -> dinit[B,E] (init_c is detached in the reference: no gradient). */
int cvae_couple_backward(const float* init, const float* init_c, const float* d, const float* gloss, int B,
                         int E, int Z, int euclid, float* dinit, void* stream);

/* ---- a11: loss assembly (Experiment.py:753-757; CC Experiment.py:405-415): the batch means of n <= 6
 * per-sentence terms (seq, kl, seq_couple, couple; bow of Experiment.py:703-706 when lambda_BoW > 0) and tot = sum_i coef[i] * mean(term_i), one launch.
 * terms: host array of n device pointers [B]; host_coef: n host floats (1, lambda_KL, lambda_seq_couple,
 * lambda_couple).  out[n+1] (device): out[0] = tot, out[1+i] = mean(term_i). */
int cvae_loss_assemble(int n, int B, const float* const* terms, const float* host_coef, float* out, void* stream);
/* gtot: device scalar upstream -> dterms[i][b] = gtot * coef[i] / B (overwritten). */
int cvae_loss_assemble_backward(int n, int B, const float* gtot, const float* host_coef, float* const* dterms,
                                void* stream);

/* ---- a13/a14: importance_sampling tail (Modules/vae_couple.py:558-578; log_sum_exp
 * utils/utils.py:69-82).  z[K,B,Z], xent[K,B] = -log p(x|z), posterior (mu,sigma)[B,Z],
 * prior (mu_p,sigma_p)[B,Z] or NULL (standard normal) ->
 *   nll[b] = -(LSE_k(log p(z) - xent - log q(z|x)) - log K), kl[b] = mean_k(log q - log p),
 *   xent_mean[b] = mean_k xent. */
int cvae_iw_reduce(const float* z, const float* xent, const float* mu, const float* sigma, const float* mu_p,
                   const float* sigma_p, int K, int B, int Z, float* nll, float* kl, float* xent_mean,
                   void* stream);
/* ---- a15: importance_sampling_mi (Modules/vae_couple.py:471-502).  z[K,B,Z] -> mi[B].
 * The (K,B,B) pairwise tensor is never formed.  ws: 2*K*B floats. */
int cvae_mi(const float* z, const float* mu, const float* sigma, int K, int B, int Z, float* ws, float* mi,
            void* stream);

/* ---- cvae_mi for callers that cannot allocate.  No counterpart in the reference: it exists for callers that
 * cannot call hipMalloc or bring a workspace -- code under HIP graph capture, tight serving loops, harnesses
 * that forbid allocation.  Same kernels and results as cvae_mi; the two [K,B] intermediates live in a
 * file-scope device array of the library (1 MiB per GPU, held until the process ends) whose per-device
 * address cvae_init(device) looks up once.  K*B <= cvae_mi_static_max_rows() (131072), else an error;
 * cvae_init must have been called for the current device.  Calls no hipMalloc / hipFree.  All calls on one
 * GPU share that GPU's array: calls that overlap on one GPU (two streams, two threads) race. */
int64_t cvae_mi_static_max_rows(void);
int cvae_mi_static(const float* z, const float* mu, const float* sigma, int K, int B, int Z, float* mi, void* stream);

/* ---- a12: clip_grad_norm_ + Adam over a flat parameter arena
 * (Experiment.py:509-522,596-599), with the reference's duplicate-listing behaviour: the
 * segment [alias_off, alias_off+alias_n) (the shared embedding) is listed `alias_mult`
 * times: its squared norm counts that many times, its gradient is scaled by coef^mult and it
 * takes `alias_mult` Adam steps per call (step counter kept by the caller: step_other for
 * the rest, step_alias = value BEFORE this call for the alias segment).
 * sumsq: 2 floats that receive the squared norms (all gradients, alias segment); they are summed without float
 * atomics, block partials in a fixed order: the clip coefficient is bit-identical from run to run.
 * params/grads/m/v: n floats each. */
int cvae_clip_adam(float* params, float* grads, float* m, float* v, int64_t n, int64_t alias_off,
                   int64_t alias_n, int alias_mult, float max_norm, float lr, float beta1, float beta2,
                   float eps, int64_t step_other, int64_t step_alias, float* sumsq, float* norm_out,
                   void* stream);
/* The same with one more parameter group, [late_off, late_off + late_n) (outside the alias segment): parameters
 * whose gradient is None in some iterations of the reference.  torch's clip_grad_norm_ and Adam skip a parameter
 * without a gradient: it does not count in the norm, m / v / the parameter stay as they are and its per-parameter
 * step count does not advance -- the BoW head while `lambda_KL <= 1e-5` leaves the bag-of-words term out of the
 * total (CV/Experiment.py:711-713, 759-761; step at :509-522).  step_late = the number of steps THIS group has
 * taken before the call (its bias corrections use step_late + 1); step_late < 0 = the group has no gradient in this
 * iteration and is skipped as above.  late_n = 0: cvae_clip_adam. */
int cvae_clip_adam_late(float* params, float* grads, float* m, float* v, int64_t n, int64_t alias_off,
                        int64_t alias_n, int alias_mult, float max_norm, float lr, float beta1, float beta2,
                        float eps, int64_t step_other, int64_t step_alias, int64_t late_off, int64_t late_n,
                        int64_t step_late, float* sumsq, float* norm_out, void* stream);

/* ---- arithmetic of the large dense products (no reference counterpart: torch computes them in fp32).  The
 * hidden x vocab product of cvae_projce_* (E <= 200, E % 8 == 0, 16-byte rows) and the many-row recurrent gate
 * product of the IW-evaluation decoder (R > 256) run as error-compensated split-bf16 ("bf16x3") by default: every
 * fp32 operand = hi + lo in bf16, product = hi.hi + hi.lo + lo.hi on v_mfma_f32_32x32x16_bf16 with fp32
 * accumulation -- relative error ~2^-17 per product, i.e. one output is accurate to ~5e-6 of its magnitude (fp32-input
 * MFMA: ~1e-7); the x W^T projections (K a multiple of 8) and the forward recurrence use the same form.  Measured against fp64 at E = 200, V = 20000: per-sentence loss
 * 2.0e-7 relative (fp32-input MFMA kernel: 1.7e-7), gradients 9.8e-7 for both
 * (tests/test_gpu_kernels.py::test_projce_bf16x3_error).  The long-K gradient products (dX = P.W, [dW|db] = P^T.[X|1],
 * the recurrent layers' weight gradients; 16-byte aligned operands) use a THREE-term split and the six leading
 * products ("bf16x6", ~2^-24 per product): with two terms the weight gradients were 10x less accurate than the
 * fp32-input kernel's, with three they are as accurate (scripts/grad_error_modes.py: worst element 7e-7 .. 1.3e-5 of
 * the tensor's maximum vs the oracle, fp32-input kernels 1.0e-6 .. 2.4e-5).  cvae_split_bf16(0) selects the
 * fp32-input MFMA kernels (v_mfma_f32_32x32x2_f32) everywhere; returns the previous setting; env CVAE_SPLIT_BF16=0
 * sets the default. */
int cvae_split_bf16(int on);
/* Run-to-run bit-reproducible results (no counterpart in the reference; default off, env CVAE_DETERMINISTIC=1):
 * the three places that add floats with atomics take an order-independent path -- the embedding-gradient scatter of
 * cvae_encoder_backward* / cvae_decoder_backward* goes through 64-bit fixed point (2^-36 units; library-owned
 * V*E*8-byte buffer per stream), cvae_gemm's small-output products are not split over K, column sums run in one
 * block per column group.  Everything else (split-K slabs, recurrent exchanges, the clip norm) adds in a fixed
 * order in either mode.  Returns the previous setting. */
int cvae_deterministic(int on);
/* Recurrent layers whose W_hh is far larger than the chip's L2 (R <= 64 rows, H % 64 == 0, split-bf16 on; the scale-up
 * shape: 67 MB per 2048-unit LSTM) run their per-step kernels in a weight-streaming form: W_hh pre-split once per
 * sequence into a fragment-ordered bf16 hi / lo image, bf16x3 products, fused cell (forward) / K-split partial
 * products summed in a fixed order (backward).  The form is used when G*H*H*4 >= the threshold set here (default
 * 8 MB, env CVAE_STREAM_RNN_MIN_MB; 0 = every supported shape; tests use it to exercise the kernels at small
 * shapes).  bytes < 0 only queries.  Returns the previous threshold.  No reference counterpart. */
int64_t cvae_stream_rnn_min_bytes(int64_t bytes);

/* ---- measurement aid, no reference counterpart: opt-in HIP-event timing of the library's
 * own kernels on the launch stream (bench.py's roofline leg).  enable(1) resets and starts
 * recording (creates events on first use), enable(0) stops.  kernel: 0 tiled MFMA GEMM,
 * 1 skinny MFMA product / fused RNN step, 2 RNN cell forward, 3 RNN cell backward,
 * 4 projection+CE core (forward and gradient), 5 long-K split GEMM (gemm_wide), 6 / 7 whole-sequence
 * recurrent kernel forward / backward.  total_work: algorithmic FLOPs (0,1,4,5,6,7) or bytes (2,3);
 * kernels whose row count lives on the device (4, 5) count the live rows. */
int cvae_profile_enable(int on);
int cvae_profile_read(int kernel, int64_t* launches, double* total_ms, double* total_work);

/* Library-owned device buffers (split-K slabs, the recurrent kernels' exchange and sync words, live-row lists,
 * weight images, the fixed-point embedding accumulator) are kept per (device, stream), allocated at the first call
 * that needs them and regrown when a larger shape arrives -- those calls hipMalloc / hipMemset (and hipFree on
 * regrowth, which synchronises the device): warm every shape up once before timing or capturing.  A process that
 * creates and destroys streams calls cvae_release_stream(stream) before destroying one (synchronises that stream,
 * frees its buffers on the current device; a later stream that reuses the handle starts clean); cvae_shutdown()
 * does the same for every stream of the current device.  No counterpart in the reference. */
int cvae_release_stream(void* stream);
int cvae_shutdown(void);

/* small helpers used by the host mirror */
/* Synchronises the device and returns (then clears) the timeout word of the whole-sequence
 * recurrent kernels: 0 = every in-kernel wait completed; 1 = a bounded spin ran out (results of
 * that call are invalid).  No counterpart in the reference (torch's cuDNN RNN has no such state). */
int cvae_persist_status(void);
/* Data-parallel training: *dst = 1.0f if the word is set else 0.0f (export, enqueued); raise the local word if *src is
 * non-zero (import).  A caller puts the exported float into its gradient all-reduce and imports the reduced value, so
 * that a time-out on any rank makes cvae_clip_adam skip that step on every rank.  No counterpart in the reference. */
int cvae_persist_status_export(float* dst, void* stream);
int cvae_persist_status_import(const float* src, void* stream);
/* Runtime form of CVAE_PERSISTENT_RNN: 0 = recurrences run on the per-timestep kernels from the next call on (what a
 * caller switches to after a time-out, without restarting); returns the previous setting. */
int cvae_persistent_rnn(int on);
/* Many-row recurrent step (IW evaluation, LSTM / GRU, H = 256 / 512): live-row count from which the weights-in-registers
 * kernel replaces the 128-row-tile one (default 64; env CVAE_RNN_LARGE_WREG_MIN; a huge value = never; negative =
 * query only).  Returns the previous value.  A tuning / test switch with no counterpart in the reference. */
int cvae_rnn_large_wreg(int min_rows);
/* CUs the whole-sequence kernels leave free (default 0; env CVAE_PERSIST_HEADROOM): a data-parallel job sets it so
 * that RCCL's kernels find room beside them; lock-stepped sequences that no longer fit together run as two
 * launches.  cus < 0 queries; returns the previous value.  No counterpart in the reference. */
int cvae_persist_headroom(int cus);
/* The same word read without synchronising or clearing (it lives in host-mapped pinned memory and the kernel
 * that timed out writes it directly): 1 as soon as that store has landed.  Cheap enough to call every training
 * step.  cvae_clip_adam takes its own look at the word on the device and leaves the parameters untouched while
 * it is set, so a timed-out step is never applied; clear it with cvae_persist_status(). */
int cvae_persist_status_peek(void);
/* Test aid: writes the word from the host (what a timed-out kernel would do), to exercise the callers' handling. */
int cvae_persist_status_set(int value);
int cvae_sum_time(const float* x, int T, int R, float* out, void* stream);       /* out[r] = sum_t x[t,r] */
int cvae_axpy(const float* x, float a, int64_t n, float* y, void* stream);       /* y += a*x */

#ifdef __cplusplus
}
#endif
#endif
