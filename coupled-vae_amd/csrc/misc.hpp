// Launchers of misc.hip / projce.hip used by api.hip.
#pragma once
#include "common.hpp"

namespace cvae {

int launch_pack_batch(const int64_t* flat, const int64_t* off, const int64_t* sel, int B, int max_len, int width, int64_t pad,
                      int64_t go, int64_t eos, int64_t* bare, int64_t* go_out, int64_t* eos_out, int64_t* len,
                      hipStream_t s);
int launch_sort_resort(const int64_t* len, int B, int64_t* s_idx, int64_t* res_idx, hipStream_t s);
int launch_embed(int T, int R, int E, int with_cond, int prefix, const int64_t* ids, int Bids, const float* emb,
                 const float* cond, const float* mask, const int64_t* mask_rows, float* out, hipStream_t s);
// V: rows of demb (only the deterministic form needs it; 0 = unknown -> float atomics)
int launch_embed_scatter(int T, int B, int E, int W, int slot0, const int64_t* ids, const float* dX,
                         const float* mask, const int64_t* mask_rows, float* demb, hipStream_t s, long V = 0);
int launch_dec_dinit(int T, int B, int E, const float* dX, const float* mask, float* dinit, hipStream_t s);
int launch_mul(long n, const float* a, const float* b, float* o, hipStream_t s);
int launch_act_bwd(long n, int act, const float* y, const float* dy, float* o, hipStream_t s);
int launch_axpy(long n, const float* x, float a, float* y, hipStream_t s);
int launch_sum_time(const float* x, int T, int R, float* out, hipStream_t s);
int launch_bcast_time(const float* g, int T, int R, float* out, hipStream_t s);
int launch_copy_cols(int R, int C, const float* src, long lds, int soff, float* dst, long ldd, int doff,
                     hipStream_t s);
int launch_targets_tm(const int64_t* eos, int B, int T, int R, int64_t* out, hipStream_t s);
int launch_targets_sm(const int64_t* eos, int B, int T, int K, int64_t* out, hipStream_t s);
// zero several buffers with one launch: add(...) then flush() (also flushes when full)
struct ZeroList { static constexpr int MAX = 8; float* p[MAX]; long n[MAX]; unsigned bits[MAX]; int count; };
struct ZeroBatch {
  ZeroList z = {};
  hipStream_t s;
  int flush_status = 0;
  explicit ZeroBatch(hipStream_t stream) : s(stream) {}
  void add(float* p, long n);
  void add_fill(float* p, long n, unsigned bits);      // every 4-byte word = bits (add = add_fill(.., 0))
  int flush();
};
bool linear_bwd_small_supported(int M, int N, int K);
int launch_linear_bwd_small(int M, int N, int K, int act, const float* x, const float* W, const float* y,
                            const float* dy, float* dx, float* dW, float* db, hipStream_t s);
int launch_loss_assemble(int n, int B, const float* const* terms, const float* coef, float* out, hipStream_t s);
int launch_loss_assemble_bwd(int n, int B, const float* gtot, const float* coef, float* const* dterms, hipStream_t s);
int launch_logsoftmax_topk(int R, int V, int K, const float* logits, long ld, int mask_id, float* lv, int* li,
                           hipStream_t s);
int launch_beam_merge(int B, int K, int ns, int t, const float* lv, const int* li, const float* nll_prev,
                      float* nll_new, int* src, int64_t* cur, int* bp, int* tok, hipStream_t s);
int launch_beam_gather(int B, int K, int H, const int* src, const float* in, float* out, hipStream_t s);
int launch_beam_backtrack(int B, int K, int T, const int* bp, const int* tok, const float* nll, int64_t* out,
                          float* out_nll, hipStream_t s);
int launch_argmax_rows(int B, int V, const float* logits, long ld, int mask_id, int64_t* out, long out_ld,
                       long out_off, int64_t* cur, hipStream_t s);
int launch_add_i64(const int64_t* x, int64_t c, int n, int64_t* y, hipStream_t s);
int launch_live_rows_tm(int T, int B, const int64_t* za, int off, int* rows, int* count, hipStream_t s);
// several lists over the same batch of B rows in one launch; za_out (optional): za + off as an array
struct LiveRowsJob { int T; int off; const int64_t* za; int* rows; int* count; int64_t* za_out; };
struct LiveRowsBatch { static constexpr int MAX = 8; LiveRowsJob j[MAX]; int B; };
int launch_live_rows_tm_n(const LiveRowsJob* jobs, int n, int B, hipStream_t s);
int launch_last_nonzero_step_n(int n, int T, int R, int H, const float* const* x, int64_t* const* za, hipStream_t s);
int launch_last_nonzero_step(int T, int R, int H, const float* x, int64_t* za, hipStream_t s);
int launch_logits_bt(int T, int B, long V, const float* tmp, float* out, hipStream_t s);
int launch_reparam_fwd(const float* mu, const float* sigma, const float* eps, int n, int B, int Z, float* z,
                       hipStream_t s);
int launch_reparam_bwd(const float* dz, const float* eps, int B, int Z, float* dmu, float* dsigma, hipStream_t s);
int launch_kl_fwd(const float* mu, const float* sigma, const float* mu_p, const float* sigma_p, int B, int Z,
                  float* kl, hipStream_t s);
int launch_kl_bwd(const float* mu, const float* sigma, const float* mu_p, const float* sigma_p, const float* gkl,
                  int B, int Z, float* dmu, float* dsigma, float* dmu_p, float* dsigma_p, hipStream_t s);
int launch_kl_fb_fwd(const float* mu, const float* sigma, int B, int Z, float floor, float* kl, hipStream_t s);
int launch_kl_fb_bwd(const float* mu, const float* sigma, const float* gkl, int B, int Z, float floor, float* dmu,
                     float* dsigma, hipStream_t s);
int launch_couple_fwd(const float* a, const float* c, int B, int E, int Z, int euclid, float* d, float* loss,
                      hipStream_t s);
int launch_couple_bwd(const float* a, const float* c, const float* d, const float* g, int B, int E, int Z,
                      int euclid, float* da, hipStream_t s);
int launch_iw_reduce(const float* z, const float* xent, const float* mu, const float* sigma, const float* mu_p,
                     const float* sigma_p, int K, int B, int Z, float* nll, float* kl, float* xent_mean,
                     hipStream_t s);
int launch_mi(const float* z, const float* mu, const float* sigma, int K, int B, int Z, float* lqzx, float* lqz,
              float* mi, hipStream_t s);
// static-scratch variant (no allocation, no workspace argument): see misc.hip
int mi_static_init(int device);
long mi_static_max_rows();
int launch_mi_static(const float* z, const float* mu, const float* sigma, int K, int B, int Z, float* mi,
                     hipStream_t s);
int launch_status_export(float* dst, hipStream_t s);
int launch_status_import(const float* src, hipStream_t s);
int launch_clip_adam(float* p, float* g, float* m, float* v, long n, long alias_off, long alias_n, int mult,
                     float max_norm, float lr, float b1, float b2, float eps, long step_other, long step_alias,
                     long late_off, long late_n, long step_late, float* sumsq, float* norm_out, hipStream_t s);

// projce.hip
long projce_workspace_floats(int N, int V);
int launch_projce_forward(const float* X, const float* W, const float* bias, const int64_t* tgt, int64_t pad,
                          int N, int E, int V, float* loss, float* lse, float* ws, hipStream_t s);
long projce_bwd_workspace_floats(int N, int V);
int launch_projce_backward(const float* X, const float* W, const float* bias, const int64_t* tgt, int64_t pad,
                           const float* lse, const float* gloss, int N, int E, int V, float* dX, float* dW,
                           float* db, float* ws, hipStream_t s);
int launch_colsum_acc(const float* P, int M, int N, long ld, float* out, hipStream_t s);

// ce_fast.hip (E <= 200): live-row compaction + register-resident rows + 224-wide split-K GEMMs
bool ce_fast_supported(int E);
long ce_fast_workspace_floats(long N, int V);
int launch_ce_fast_forward(const float* X, const float* W, const float* bias, const int64_t* tgt, int64_t pad,
                           long N, long TR, int E, int V, float* loss, float* lse, float* ws, float* logits,
                           hipStream_t s);
long ce_fast_logits_floats(long N, int V);
long ce_fast_bwd_workspace_floats(long N, int V);
int launch_ce_fast_backward(const float* X, const float* W, const float* bias, const int64_t* tgt, int64_t pad,
                            const float* lse, const float* gsent, long N, long TR, int R, int E, int V,
                            float* dX, float* dW, float* db, float* ws, float* logits, hipStream_t s);

// the same for nd decoders over one set of targets (saved-logits training path; the row list is built once)
int launch_ce_fast_forward_n(int nd, const float* const* X, const float* const* W, const float* const* bias,
                             const int64_t* tgt, int64_t pad, long N, long TR, int E, int V, float* const* loss,
                             float* const* lse, float* const* ws, float* const* logits, hipStream_t s);
int launch_ce_fast_backward_n(int nd, const float* const* X, const float* const* W, const float* const* bias,
                              const int64_t* tgt, int64_t pad, const float* const* lse, const float* const* gsent,
                              long N, long TR, int R, int E, int V, float* const* dX, float* const* dW,
                              float* const* db, float* ws, float* const* logits, hipStream_t s);

}  // namespace cvae
