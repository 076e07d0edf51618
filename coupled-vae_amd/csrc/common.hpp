// Shared host/device helpers for libcvae_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

// Timing-only diagnostic switches (CE_EXP_*, GW_EXP_*, RB_EXP_*, WR_EXP_*: scripts/build_variant.sh) remove work from a kernel to
// see what it costs; a library built with any of them computes garbage.  They are accepted only together with
// -DCVAE_TIMING_BUILD, and every object compiled that way flips the sign of cvae_version(), so that the Python
// binding (and tests/test_abi_and_host.py) refuse such a library unless CVAE_ALLOW_TIMING_BUILD=1.
#if !defined(CVAE_TIMING_BUILD) && \
    (defined(CE_EXP_NOLOAD) || defined(CE_EXP_NOLDS) || defined(CE_EXP_NOEPI) || defined(CE_EXP_STAMPS) || \
     (defined(GW_EXP_NOSPLIT) && GW_EXP_NOSPLIT) || (defined(GW_EXP_NOMFMA) && GW_EXP_NOMFMA) || \
     (defined(GW_EXP_NOLOAD) && GW_EXP_NOLOAD) || (defined(GW_EXP_NOLDS) && GW_EXP_NOLDS) || (defined(GW_EXP_NOB) && GW_EXP_NOB) || \
     (defined(RB_EXP_NOSPLIT) && RB_EXP_NOSPLIT) || (defined(RB_EXP_NOEPI) && RB_EXP_NOEPI) || \
     (defined(RB_EXP_NOMFMA) && RB_EXP_NOMFMA) || (defined(WR_EXP_NOEPILOAD) && WR_EXP_NOEPILOAD) || \
     (defined(WR_EXP_NOSTAGELOAD) && WR_EXP_NOSTAGELOAD) || \
     (defined(WR_EXP_NOMFMA) && WR_EXP_NOMFMA) || (defined(WR_EXP_NOCELL) && WR_EXP_NOCELL) || \
     (defined(WR_EXP_NOB) && WR_EXP_NOB) || (defined(WR_EXP_STAMPS) && WR_EXP_STAMPS))
#error "*_EXP_* timing switches need -DCVAE_TIMING_BUILD (the library then reports a negative cvae_version())"
#endif
namespace cvae {
extern int g_timing_build;              // api.hip; 1 once any object of the library was compiled with CVAE_TIMING_BUILD
#ifdef CVAE_TIMING_BUILD
namespace { struct TimingBuildMark { TimingBuildMark() { g_timing_build = 1; } } g_timing_build_mark; }
#endif
}
namespace cvae {

// ---- error plumbing: every extern "C" entry returns 0 or a negative code and leaves a
// message for cvae_last_error().
void set_error(const char* fmt, ...);
int check_hip(hipError_t e, const char* what);

#define CVAE_REQUIRE(cond, ...)                         \
  do {                                                  \
    if (!(cond)) {                                      \
      ::cvae::set_error(__VA_ARGS__);                   \
      return -1;                                        \
    }                                                   \
  } while (0)

#define CVAE_LAUNCH_CHECK(name)                                        \
  do {                                                                 \
    int _rc = ::cvae::check_hip(hipGetLastError(), name);              \
    if (_rc) return _rc;                                               \
  } while (0)

#define CVAE_TRY(expr)          \
  do {                          \
    int _rc = (expr);           \
    if (_rc) return _rc;        \
  } while (0)

// ---- opt-in per-kernel timing (bench.py's roofline leg): HIP events recorded on the launch
// stream around the kernels named below.  Off by default: no events, no overhead.
enum ProfKernel { PK_GEMM_TILED = 0, PK_SKINNY = 1, PK_CELL_FWD = 2, PK_CELL_BWD = 3, PK_CE_FWD = 4, PK_GEMM_WIDE = 5,
                  PK_RNN_PERSIST = 6, PK_RNN_PERSIST_BWD = 7, PK_RNN_LARGE = 8, PK_GEMM_NT = 9,
                  PK_CLIP_ADAM = 10, PK_CE_GRAD = 11, PK_IW_REDUCE = 12, PK_COUNT = 13 };   // 10-12: HBM-bound, work = bytes
struct ProfScope {
  int slot;
  hipStream_t s;
  ProfScope(int kernel, hipStream_t stream, double work);   // work: algorithmic FLOPs (MFMA kernels) or bytes
  ~ProfScope();
};

bool prof_enabled();   // true while cvae_profile_enable(1) is in effect

// Error-compensated split-bf16 products ("bf16x3", ce_fast.hip / rnn_large.hip): on by default, env CVAE_SPLIT_BF16=0
// or cvae_split_bf16(0) selects the fp32-input MFMA kernels instead.
bool split_bf16_enabled();
int split_bf16_set(int on);      // returns the previous setting
// Run-to-run bit-reproducible results (cvae_deterministic / CVAE_DETERMINISTIC=1, default off): the three places that
// add floats with atomics take an order-independent path instead (embedding-gradient scatter through 64-bit fixed
// point, no atomic split-K in gemm_tiled, column sums in one block per column group).
bool deterministic_enabled();
int deterministic_set(int on);   // returns the previous setting

// ---- library-owned device scratch (api.hip): one buffer per (kind, device, stream), so launches on different
// streams (or host threads driving different streams) never share flag words, exchange slabs or index lists.
// Allocated at first use and regrown only when a larger shape arrives (warm-up; hipFree of the old buffer
// synchronises the device); steady-state calls allocate nothing.  Guarded by a mutex.
enum LibScratchKind { LS_PERSIST_SYNC = 0, LS_PERSIST_EXCH = 1, LS_WIDE_PART = 2, LS_ROWS = 3, LS_STREAM_IMG = 4,
                      LS_STREAM_PARTS = 5, LS_SUMSQ = 6, LS_EMB_FIX = 7, LS_H_IMAGE = 8, LS_W_IMAGE = 9, LS_KINDS = 10 };
// zero != 0: a new buffer is zero-filled (synchronously) before it is handed out.  fresh (optional): set when
// this call (re)allocated.
int lib_scratch(int kind, hipStream_t s, size_t bytes, void** out, int zero = 0, bool* fresh = nullptr);
// The recurrent kernels' timeout word: ONE int in host-mapped pinned memory (device pointer returned), written
// by a kernel whose bounded spin ran out; readable on the host without synchronising (cvae_persist_status_peek).
int persist_status_word(int** dev_ptr);
// rnn_persist.hip: size of a stream's sync buffer (LS_PERSIST_SYNC) and the spare word in it that holds the
// device-side copy of the timeout word taken at each optimizer step
size_t persist_sync_bytes();
int* persist_status_copy_word(void* sync_buffer);
int persist_status_peek();

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

enum Act { ACT_NONE = 0, ACT_TANH = 1, ACT_RELU = 2, ACT_EXP = 3, ACT_CEGRAD = 4 };
enum Cell { CELL_GRU = 0, CELL_LSTM = 1 };

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }
// tanh through exp: accurate to ~1e-7 relative over the whole range (no fast-math shortcuts)
__device__ __forceinline__ float tanhf_(float x) { return tanhf(x); }

// Two-instruction forms (v_exp_f32 + v_rcp_f32, about 1 ulp each; absolute error ~1e-7, no denormal or range
// handling needed: exp2 saturates to 0 / inf and rcp(inf) = 0).  For the many-row evaluation step, whose cell
// epilogue otherwise spends as many issue cycles on libm's expf / tanhf / IEEE division as its main loop on MFMAs.
__device__ __forceinline__ float sigmoid_fast(float x) {
  return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
}
__device__ __forceinline__ float tanh_fast(float x) {
  return fmaf(-2.0f, __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(2.8853900817779268f * x)), 1.0f);
}

__device__ __forceinline__ float apply_act(float v, int act) {
  switch (act) {
    case ACT_TANH: return tanhf_(v);
    case ACT_RELU: return v > 0.f ? v : 0.f;
    case ACT_EXP: return expf(v);
    default: return v;
  }
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// ---- internal launchers shared between translation units (all enqueue on `s`) ----------
constexpr int GEMM_MAX_BATCH = 256;
struct GemmDesc {
  int transA, transB;        // op(A) is MxK: transA=0 -> A[m*lda+k], 1 -> A[k*lda+m]
  int M, N, K;               // op(B) is KxN: transB=0 -> B[k*ldb+n], 1 -> B[n*ldb+k]
  const float* A; long lda;
  const float* B; long ldb;
  float* C; long ldc;
  const float* bias;         // [N] or null
  int act;                   // Act, applied after bias
  int beta;                  // 0: overwrite, 1: C += result
  // ACT_CEGRAD epilogue (projce backward): C = coef[m] * (exp(v - lse[m]) - (n == tgt[m]))
  const float* row_lse; const float* row_coef; const int64_t* row_tgt;
  // transA products only (C = A^T B, weight gradients): also colsum_out[m] = sum_k A[k][m], the bias
  // gradient.  Free when the product runs as gemm_wide with a spare tile column (a ones column in
  // the staged B tile); otherwise one extra pass over A.
  float* colsum_out;
  // nbatch > 0 (plain A only): nbatch products sharing B; batch i reads A + i*batchA, writes C + i*batchC
  // and has host_mbatch[i] rows (M is ignored).  One launch of the tiled kernel, blockIdx.z = batch.
  int nbatch; long batchA, batchC; const int* host_mbatch;
  // Optional live-row list (device): the product only visits rows[0 .. *nlive) of the row dimension
  // that A shares with C (bound_dim 1: C = A B, rows of A and C; the caller has zeroed C) or with
  // B (bound_dim 2, transA: C = A^T B, rows of A and B).  Rows outside the list must be exactly
  // zero in A for the result to be the full product.  Always runs as gemm_wide.
  const int* rows; const int* nlive; int bound_dim;
};
int launch_gemm(const GemmDesc& d, hipStream_t s);
int launch_gemm_n(const GemmDesc* d, int n, hipStream_t s);     // same-shaped gemm_wide products share one launch
int launch_colsum_acc(const float* P, int M, int N, long ld, float* out, hipStream_t s);
bool gemm_wide_has_spare_column(int N);

// Long-K split-K product with a few hundred output columns (gemm_wide.hip): C += A . B, B = [k][n].
struct G224Args {
  const float* A; long lda;
  const float* B; long ldb;
  float* C; long ldc;
  int M, N, K;
  const int* nlive; int r0;   // optional device-side bound: lim = max(0, *nlive - r0)
  int dev_bound;              // 0: none; 1: M <= lim, C row = rows[r0 + m]; 2: K <= lim, B row = rows[r0 + k]
  const int* rows;
  int a_ind;                  // 1: rows of A are gathered through rows[] too (A[rows[r0+m]][k] / A[rows[r0+k]][m])
  float* extra;               // optional [M]: extra[m] (+)= sum_k A[m,k] (a ones column appended to B; N <= 220)
  int beta;                   // 0: C (and extra) overwritten, 1: accumulated into
  float* part;                // partial-sum slabs (gemm_wide_part_floats); null -> library scratch
  long prof_units;            // profiling: live extent of the device-bounded dimension (0: use the host bound)
  int ntiles, tile_w, part_ld, min_chunks;   // filled by the launcher
  int vecA, vecB, vecC;
};
constexpr int G224_MAX_BATCH = 4;
struct G224Batch { G224Args a[G224_MAX_BATCH]; };
long gemm_wide_part_floats(int N, bool extra);
int launch_gemm_wide(bool a_k_contiguous, G224Args a, hipStream_t s);
int launch_gemm_wide_n(bool a_k_contiguous, const G224Args* a, int n, hipStream_t s);   // n same-shaped products, one launch
int persistent_grid_blocks();
int launch_zero(float* p, long n, hipStream_t s);
int launch_copy(float* dst, const float* src, long n, hipStream_t s);

// Few-row product with the K dimension split over the waves of a block (skinny.hip).
enum SkinnyEpi { EPI_LINEAR = 0, EPI_GRU = 1, EPI_LSTM = 2, EPI_GRU_BWD = 3, EPI_LSTM_BWD = 4 };
struct SkinnyArgs {
  const float* A; long lda;            // [M,K], k contiguous
  const float* B; long ldb; int transB;  // op(B) KxN: transB=1 -> B[n*ldb+k], 0 -> B[k*ldb+n]
  int M, N, K;                         // N = output columns per gate group
  int ngroups; long gstride;           // group g uses op(B) columns n + g*gstride
  // EPI_LINEAR: C = act(A.op(B) + bias) (+ C)
  float* C; long ldc; const float* bias; int act, beta;
  // EPI_GRU / EPI_LSTM: N = H, groups = gates (r,z,n | i,f,g,o); A = h_prev
  const float* xproj; long ldx; int xmod;  // x-side pre-activations of row r: xproj[(r % xmod)*ldx + g*H + n]
  const float* xadd;                   // optional [M, G*H], added to xproj (time-invariant part)
  const float* b_hh;                   // [G*H]
  const float* h_prev; const float* c_prev;  // [M,H]
  float* h_out; float* c_out;          // [M,H]
  float* gates; long ldg;              // optional [M, 4H]: GRU r,z,n,(W_hn h + b_hn); LSTM i,f,g,o
  const int64_t* lengths; int t;       // optional: row r is updated only while t < lengths[r]
  // EPI_*_BWD: A = dG_h of step t (K = G*H), B = W_hh^T [H, G*H] (transB = 1), N = H.
  //   dh = A.op(B) + dh_buf (+ dhs_prev) is the full gradient wrt the output of the PREVIOUS step
  //   (forward order); when has_prev the epilogue runs that step's cell backward in place:
  //   writes dgx_prev / dgh_prev, leaves the direct carry in dh_buf (and dc in dc_buf).
  //   When !has_prev it just stores dh into dh_buf (= gradient wrt the initial state).
  float* dh_buf; float* dc_buf;        // [M,H] in/out
  const float* dhs_prev;               // [M,H] or null
  const float* gates_prev;             // [M,4H]
  const float* hin_prev; const float* cin_prev; const float* cout_prev;   // state in / out of the previous step
  float* dgx_prev; float* dgh_prev;    // [M,G*H]; dgh_prev null for LSTM (same as dgx)
  int has_prev, t_prev;
};
int launch_skinny(const SkinnyArgs& a, int epi, hipStream_t s);
// n <= 4 independent problems of identical shape in ONE launch (blockIdx.z picks the problem):
// lock-step timesteps of independent recurrences (two encoder directions, two decoders).
constexpr int SKINNY_MAX_BATCH = 4;
int launch_skinny_multi(const SkinnyArgs* a, int n, int epi, hipStream_t s);

// Weight-streaming variant of the few-row product (skinny.hip), for recurrent weights far larger than the chip's
// L2 (the scale-up shape: W_hh of a 2048-unit LSTM is 67 MB): error-compensated split-bf16 (bf16x3) on
// v_mfma_f32_16x16x32_bf16, the weights read from a fragment-ordered bf16 hi / lo image that
// launch_stream_presplit writes once per sequence, M <= 64 rows, K % 32 == 0.
//   EPI_GRU / EPI_LSTM: one block = 16 hidden units x all gates over the whole K, fused cell epilogue (as skinny).
//   EPI_LINEAR        : one block = 64 output columns x one of `ksplit` K ranges; writes the partial products
//                       parts[(q*ksplit + y)*M*N + m*N + n] (problem q, K range y): the caller sums them.
bool stream_rnn_supported(int R, int H, int G);        // shape + switch (split-bf16 on, weights >= the threshold)
long stream_rnn_min_weight_bytes(long set_to);         // < 0: query; returns the previous threshold
size_t stream_image_bytes(int ncols, int K);
// column (block bx, group g, j) of the image = row `bx*cb + g*gs + j` of op(B); element (col, k) = W[col*col_stride + k*k_stride]
int launch_stream_presplit(const float* W, long col_stride, long k_stride, int ncols, int K, int ng, int cb, long gs,
                           void* img, hipStream_t s);
int launch_skinny_stream(const SkinnyArgs* a, void* const* imgs, int n, int epi, int ksplit, float* parts, hipStream_t s);
int stream_linear_ksplit(int n, int N, int K);


// One direction of a 1-layer GRU/LSTM over T steps (rnn.hip).  State slots: hs/cs hold T+1
// [R,H] slabs; forward reads slot t and writes slot t+1 (initial state in slot 0), reverse
// walks t = T-1..0, reads slot t+1 and writes slot t (initial state in slot T).
constexpr int SKINNY_MAX_ROWS = 256;
struct RnnFwd {
  int cell, T, R, H, reverse;
  const float* xproj; int xmod;     // [T, xmod, G*H] x-side pre-activations (b_ih included); row r uses r % xmod
  const float* xadd;                // [R, G*H] or null: time-invariant x-side part
  const float* W_hh; const float* b_hh;
  const float* h0; const float* c0; // [R,H] or null (zeros)
  int init_in_place;                // 1: the initial slot of hs/cs already holds the state
  const int64_t* lengths;           // [R] or null: packed-sequence masking (row live iff t < len)
  float* hs; float* cs;             // [(T+1),R,H]; cs null for GRU
  float* gates;                     // [T,R,4H] or null (saved for backward)
  float* hp_scratch;                // [R,G*H]; required when R > SKINNY_MAX_ROWS
  // Large-R path only (IW eval, sentence-major rows r = sentence*xdiv + sample):
  int xdiv;                         // > 1: row r uses x-side row (r / xdiv) % xmod
  const int* live_rows;             // host [T] or null: step t touches rows [0, live_rows[t]) only
};
int launch_rnn_forward(const RnnFwd& a, hipStream_t s);
// ns independent sequences with the same cell / T / R / H, advanced in lock-step
int launch_rnn_forward_multi(const RnnFwd* a, int ns, hipStream_t s);

struct RnnBwd {
  int cell, T, R, H, reverse;
  const float* dhs;                 // [T,R,H] grad wrt each step's output, or null
  const float* dh_last; const float* dc_last;  // [R,H] grad wrt the final state, or null
  const float* hs; const float* cs; const float* gates;
  const float* W_hh;
  const int64_t* lengths;
  const int64_t* zero_after;        // [R] or null: row r's incoming gradient is exactly zero at steps t >= zero_after[r]
  float* dgx; float* dgh;           // [T,R,G*H] pre-activation grads, x side / h side (LSTM: dgh may be null = dgx)
  float* dh0; float* dc0;           // [R,H] out: grad wrt the initial state
  float* whhT;                      // [H, G*H] scratch: W_hh transposed (needed when R <= SKINNY_MAX_ROWS)
  float* dh_tmp; float* dc_tmp;     // [R,H] scratch (large-R path only)
};
int launch_rnn_backward(const RnnBwd& a, hipStream_t s);
int launch_rnn_backward_multi(const RnnBwd* a, int ns, hipStream_t s);
// rnn_large.hip: one step for many rows, gate product + cell in one kernel (IW evaluation)
bool rnn_large_fused_supported(int cell, int H, const float* h, const float* W_hh);
// the same step with W_hh resident in registers (LSTM, H = 256 / 512, many rows): reads h_prev as a bf16 (hi, lo) image
// ([2][img_lo / H rows][H]; launch_rnn_large_split makes one from fp32 rows) and writes h_out's image next to h_out
bool rnn_large_wreg_supported(int cell, int H, int live, int xmod, int xdiv, const float* h, const float* W_hh);
size_t rnn_large_image_bytes(int R, int H);
long rnn_large_image_lo(int R, int H);          // elements from the hi half to the lo half
int launch_rnn_large_split(const float* h, int rows, int H, void* img, long img_lo, hipStream_t s);
size_t rnn_large_wimg_bytes(int H);
int launch_rnn_large_wimg(int cell, const float* W_hh, int H, void* wimg, hipStream_t s);       // W_hh -> the kernel's fragment image
int launch_rnn_large_wreg_step(int cell, int live, int H, const float* h_prev, const float* c_prev, const void* wimg,
                               const float* b_hh, const float* xproj, int xmod, int xdiv, const float* xadd, float* h_out,
                               float* c_out, const void* img_in, void* img_out, long img_lo, hipStream_t s);
int launch_rnn_large_step(int cell, int live, int H, const float* h_prev, const float* c_prev, const float* W_hh,
                          const float* b_hh, const float* xproj, int xmod, int xdiv, const float* xadd, float* h_out,
                          float* c_out, hipStream_t s);
// rnn_persist.hip: all T steps of up to four lock-stepped sequences in one launch (R <= 32)
bool rnn_persist_supported(int cell, int T, int R, int H, int ns);
bool rnn_persist_bwd_supported(int cell, int T, int R, int H, int ns);
struct ZeroBatch;
// initial slots already set or queued on zb (flushed here together with the flag words)
int launch_rnn_forward_persist(const RnnFwd* a, int ns, hipStream_t s, ZeroBatch* zb = nullptr);
int launch_rnn_backward_persist(const RnnBwd* a, int ns, hipStream_t s);
int persist_status_and_clear();
int rnn_large_wreg_set(int min_rows);  // rnn_large.hip: row count from which the weights-in-registers step kernel runs; returns the previous one
int persist_rnn_set(int on);           // runtime form of CVAE_PERSISTENT_RNN; returns the previous setting
int persist_headroom_set(int cus);     // CUs left free by the whole-sequence kernels (cus < 0: query); returns the previous value

}  // namespace cvae
