// extern "C" surface of libcvae_hip (declared in include/cvae_hip.h) and the host-side
// sequencing of the kernels: which GEMM feeds which recurrence, in what layout.
#include "../../include/cvae_hip.h"

#include <stdarg.h>
#include <string.h>

#include <map>
#include <mutex>
#include <tuple>
#include <vector>

#include "common.hpp"
#include "misc.hpp"

namespace cvae { int g_timing_build = 0; }
namespace cvae {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int check_hip(hipError_t e, const char* what) {
  if (e == hipSuccess) return 0;
  set_error("%s: %s", what, hipGetErrorString(e));
  return -2;
}

// ---------------------------------------------------------------- profiling state
struct ProfRec { int kernel; double work; hipEvent_t beg, end; };
static bool g_prof_on = false;
static std::vector<ProfRec> g_prof;
static std::vector<hipEvent_t> g_pool;
static size_t g_pool_used = 0;
constexpr size_t kProfMaxRecs = 1 << 17;

static bool prof_event(hipEvent_t* e) {
  if (g_pool_used == g_pool.size()) {
    hipEvent_t ne;
    if (hipEventCreate(&ne) != hipSuccess) return false;
    g_pool.push_back(ne);
  }
  *e = g_pool[g_pool_used++];
  return true;
}

bool prof_enabled() { return g_prof_on; }

static int g_split_bf16 = -1;                    // -1: not set yet (env CVAE_SPLIT_BF16, default on)
bool split_bf16_enabled() {
  if (g_split_bf16 < 0) { const char* e = getenv("CVAE_SPLIT_BF16"); g_split_bf16 = (e && e[0] == '0') ? 0 : 1; }
  return g_split_bf16 != 0;
}
int split_bf16_set(int on) {
  const int prev = split_bf16_enabled() ? 1 : 0;
  g_split_bf16 = on ? 1 : 0;
  return prev;
}

static int g_deterministic = -1;                 // -1: not set yet (env CVAE_DETERMINISTIC, default off)
bool deterministic_enabled() {
  if (g_deterministic < 0) { const char* e = getenv("CVAE_DETERMINISTIC"); g_deterministic = (e && e[0] == '1') ? 1 : 0; }
  return g_deterministic != 0;
}
int deterministic_set(int on) {
  const int prev = deterministic_enabled() ? 1 : 0;
  g_deterministic = on ? 1 : 0;
  return prev;
}

ProfScope::ProfScope(int kernel, hipStream_t stream, double work) : slot(-1), s(stream) {
  if (!g_prof_on || g_prof.size() >= kProfMaxRecs) return;
  ProfRec r;
  r.kernel = kernel; r.work = work;
  if (!prof_event(&r.beg) || !prof_event(&r.end)) return;
  (void)hipEventRecord(r.beg, s);
  slot = (int)g_prof.size();
  g_prof.push_back(r);
}
ProfScope::~ProfScope() {
  if (slot >= 0) (void)hipEventRecord(g_prof[slot].end, s);
}

// ---------------------------------------------------------------- library-owned scratch
struct LibBuf { void* p; size_t bytes; };
static std::mutex g_scratch_mu;
static std::map<std::tuple<int, int, hipStream_t>, LibBuf> g_scratch;

int lib_scratch(int kind, hipStream_t s, size_t bytes, void** out, int zero, bool* fresh) {
  CVAE_REQUIRE(kind >= 0 && kind < LS_KINDS && out, "lib_scratch: bad argument");
  int dev = 0;
  CVAE_TRY(check_hip(hipGetDevice(&dev), "lib_scratch: hipGetDevice"));
  std::lock_guard<std::mutex> lock(g_scratch_mu);
  LibBuf& b = g_scratch[std::make_tuple(kind, dev, s)];
  if (fresh) *fresh = false;
  if (bytes > b.bytes) {
    if (b.p) CVAE_TRY(check_hip(hipFree(b.p), "lib_scratch: hipFree"));          // implicit device sync (warm-up only)
    b.p = nullptr; b.bytes = 0;
    CVAE_TRY(check_hip(hipMalloc(&b.p, bytes), "lib_scratch: hipMalloc"));
    b.bytes = bytes;
    if (zero) CVAE_TRY(check_hip(hipMemset(b.p, 0, bytes), "lib_scratch: hipMemset"));
    if (fresh) *fresh = true;
  }
  *out = b.p;
  return 0;
}

// Releases the library-owned buffers of one stream of the current device (all streams of all devices: every = true).
// The stream (device) is synchronised first: no kernel may still be using them.
static int lib_scratch_release(hipStream_t s, bool every) {
  int dev = 0;
  CVAE_TRY(check_hip(hipGetDevice(&dev), "release: hipGetDevice"));
  if (every) CVAE_TRY(check_hip(hipDeviceSynchronize(), "release: hipDeviceSynchronize"));
  else CVAE_TRY(check_hip(hipStreamSynchronize(s), "release: hipStreamSynchronize"));
  std::lock_guard<std::mutex> lock(g_scratch_mu);
  for (auto it = g_scratch.begin(); it != g_scratch.end();) {
    const bool hit = every ? std::get<1>(it->first) == dev : (std::get<1>(it->first) == dev && std::get<2>(it->first) == s);
    if (hit) {
      if (it->second.p) CVAE_TRY(check_hip(hipFree(it->second.p), "release: hipFree"));
      it = g_scratch.erase(it);
    } else {
      ++it;
    }
  }
  return 0;
}

constexpr int kMaxDevices = 64;
static int* g_status_host = nullptr;     // host address of the mapped word (portable: every device can map it)
static int* g_status_dev[kMaxDevices] = {};      // the same word as each device sees it
int persist_status_word(int** dev_ptr) {
  int dev = 0;
  CVAE_TRY(check_hip(hipGetDevice(&dev), "persist status: hipGetDevice"));
  CVAE_REQUIRE(dev >= 0 && dev < kMaxDevices, "persist status: device %d", dev);
  std::lock_guard<std::mutex> lock(g_scratch_mu);
  if (!g_status_host) {
    void* h = nullptr;
    CVAE_TRY(check_hip(hipHostMalloc(&h, 64, hipHostMallocMapped | hipHostMallocPortable), "persist status: hipHostMalloc"));
    memset(h, 0, 64);
    g_status_host = static_cast<int*>(h);
  }
  if (!g_status_dev[dev]) {              // the device pointer belongs to the device that is current
    void* d = nullptr;
    CVAE_TRY(check_hip(hipHostGetDevicePointer(&d, g_status_host, 0), "persist status: hipHostGetDevicePointer"));
    g_status_dev[dev] = static_cast<int*>(d);
  }
  if (dev_ptr) *dev_ptr = g_status_dev[dev];
  return 0;
}
// host-side views for rnn_persist.hip
int* persist_status_host() { return g_status_host; }

static inline hipStream_t S(void* s) { return reinterpret_cast<hipStream_t>(s); }
static inline int gates_of(int cell) { return cell == CELL_GRU ? 3 : 4; }

static int linear_nt(const float* x, const float* W, long ldw, const float* b, int act, int M, int N, int K,
                     float* y, hipStream_t s) {
  GemmDesc d = {};
  d.transA = 0; d.transB = 1; d.M = M; d.N = N; d.K = K;
  d.A = x; d.lda = K; d.B = W; d.ldb = ldw; d.C = y; d.ldc = N; d.bias = b; d.act = act;
  return launch_gemm(d, s);
}

// dW[N,K] = dy[M,N]^T . x[M,K]
// Time-major live-row list of a [T, B, .] gradient (device): rows whose gradient is exactly zero (padded
// steps, decoder rows past their last target) are left out of the products that consume it.
struct LiveRows { const int* rows; const int* n; };

static bool live_rows_enabled() {
  static int on = -1;
  if (on < 0) { const char* e = getenv("CVAE_LIVE_ROWS"); on = (e && e[0] == '0') ? 0 : 1; }
  return on != 0;
}
// library-owned index scratch, per stream (lib_scratch)
static int lib_rows(long ints, int** out, hipStream_t s) {
  void* p = nullptr;
  CVAE_TRY(lib_scratch(LS_ROWS, s, (size_t)ints * sizeof(int), &p));
  *out = static_cast<int*>(p);
  return 0;
}

// db != null: also db[N] = column sums of dy (the bias gradient), see GemmDesc::colsum_out
// lr != null: only the listed rows of dy / x are visited (the others are exactly zero in dy)
static GemmDesc desc_grad_weight(const float* dy, long lddy, const float* x, long ldx, int M, int N, int K, float* dW,
                                 int beta, float* db = nullptr, const LiveRows* lr = nullptr) {
  GemmDesc d = {};
  d.transA = 1; d.transB = 0; d.M = N; d.N = K; d.K = M;
  d.A = dy; d.lda = lddy; d.B = x; d.ldb = ldx; d.C = dW; d.ldc = K; d.beta = beta; d.colsum_out = db;
  if (lr) { d.rows = lr->rows; d.nlive = lr->n; d.bound_dim = 2; }
  return d;
}
static int grad_weight(const float* dy, long lddy, const float* x, long ldx, int M, int N, int K, float* dW,
                       int beta, hipStream_t s, float* db = nullptr, const LiveRows* lr = nullptr) {
  return launch_gemm(desc_grad_weight(dy, lddy, x, ldx, M, N, K, dW, beta, db, lr), s);
}

// dx[M,K] = dy[M,N] . W[N,K]
// lr != null: only the listed rows of dy are visited and only those rows of dx written: the caller has
// zeroed dx (beta == 0) or dx already holds the term to add to (beta == 1)
static GemmDesc desc_grad_input(const float* dy, long lddy, const float* W, long ldw, int M, int N, int K, float* dx,
                                long lddx, int beta, const LiveRows* lr = nullptr) {
  GemmDesc d = {};
  d.transA = 0; d.transB = 0; d.M = M; d.N = K; d.K = N;
  d.A = dy; d.lda = lddy; d.B = W; d.ldb = ldw; d.C = dx; d.ldc = lddx; d.beta = beta;
  if (lr) { d.rows = lr->rows; d.nlive = lr->n; d.bound_dim = 1; }
  return d;
}
static int grad_input(const float* dy, long lddy, const float* W, long ldw, int M, int N, int K, float* dx,
                      long lddx, int beta, hipStream_t s, const LiveRows* lr = nullptr) {
  return launch_gemm(desc_grad_input(dy, lddy, W, ldw, M, N, K, dx, lddx, beta, lr), s);
}

static int grad_bias(const float* dy, int M, int N, long ld, float* db, hipStream_t s) {
  CVAE_TRY(launch_zero(db, (long)N, s));
  return launch_colsum_acc(dy, M, N, ld, db, s);
}

}  // namespace cvae

using namespace cvae;

extern "C" {

int cvae_version(void) { return cvae::g_timing_build ? -100 : 100; }
const char* cvae_last_error(void) { return g_err; }

int cvae_init(int device) {
  int n = 0;
  CVAE_TRY(check_hip(hipGetDeviceCount(&n), "hipGetDeviceCount"));
  CVAE_REQUIRE(device >= 0 && device < n, "cvae_init: device %d out of range (%d visible)", device, n);
  hipDeviceProp_t prop;
  CVAE_TRY(check_hip(hipGetDeviceProperties(&prop, device), "hipGetDeviceProperties"));
  CVAE_REQUIRE(strncmp(prop.gcnArchName, "gfx950", 6) == 0,
               "cvae_init: device %d is %s; this library is built for gfx950 (MI355X) only", device,
               prop.gcnArchName);
  CVAE_TRY(check_hip(hipSetDevice(device), "hipSetDevice"));
  CVAE_TRY(mi_static_init(device));        // device address of this GPU's copy of the static scratch array
  return 0;
}

int cvae_profile_enable(int on) {
  g_prof_on = on != 0;
  g_prof.clear();
  g_pool_used = 0;
  return 0;
}

int cvae_profile_read(int kernel, int64_t* launches, double* total_ms, double* total_work) {
  CVAE_REQUIRE(kernel >= 0 && kernel < PK_COUNT && launches && total_ms && total_work, "profile_read: bad argument");
  *launches = 0; *total_ms = 0.0; *total_work = 0.0;
  for (const ProfRec& r : g_prof) {
    if (r.kernel != kernel) continue;
    CVAE_TRY(check_hip(hipEventSynchronize(r.end), "profile_read: hipEventSynchronize"));
    float ms = 0.f;
    CVAE_TRY(check_hip(hipEventElapsedTime(&ms, r.beg, r.end), "profile_read: hipEventElapsedTime"));
    *launches += 1; *total_ms += ms; *total_work += r.work;
  }
  return 0;
}

int cvae_pack_batch(const int64_t* raw_flat, const int64_t* raw_offsets, int B, int max_len, int width,
                    int64_t pad, int64_t go, int64_t eos, int64_t* bare, int64_t* go_out, int64_t* eos_out,
                    int64_t* len, void* stream) {
  return launch_pack_batch(raw_flat, raw_offsets, nullptr, B, max_len, width, pad, go, eos, bare, go_out, eos_out,
                           len, S(stream));
}
int cvae_pack_batch_indexed(const int64_t* corpus_flat, const int64_t* corpus_offsets, const int64_t* sel, int B,
                            int max_len, int width, int64_t pad, int64_t go, int64_t eos, int64_t* bare,
                            int64_t* go_out, int64_t* eos_out, int64_t* len, void* stream) {
  CVAE_REQUIRE(sel || B == 0, "pack_batch_indexed: null index vector");
  return launch_pack_batch(corpus_flat, corpus_offsets, sel, B, max_len, width, pad, go, eos, bare, go_out, eos_out,
                           len, S(stream));
}

int cvae_sort_resort(const int64_t* len, int B, int64_t* s_idx, int64_t* res_idx, void* stream) {
  return launch_sort_resort(len, B, s_idx, res_idx, S(stream));
}

int cvae_gemm(int transA, int transB, int M, int N, int K, const float* A, int64_t lda, const float* B,
              int64_t ldb, float* C, int64_t ldc, const float* bias, int act, int beta, void* stream) {
  CVAE_REQUIRE(act >= ACT_NONE && act <= ACT_EXP, "cvae_gemm: bad act %d", act);
  GemmDesc d = {};
  d.transA = transA; d.transB = transB; d.M = M; d.N = N; d.K = K;
  d.A = A; d.lda = lda; d.B = B; d.ldb = ldb; d.C = C; d.ldc = ldc; d.bias = bias; d.act = act; d.beta = beta;
  return launch_gemm(d, S(stream));
}

int cvae_linear_forward(const float* x, const float* W, const float* b, int act, int M, int N, int K, float* y,
                        void* stream) {
  CVAE_REQUIRE(act >= ACT_NONE && act <= ACT_EXP, "linear_forward: bad act %d", act);
  return linear_nt(x, W, K, b, act, M, N, K, y, S(stream));
}

int cvae_linear_backward(const float* x, const float* W, const float* y, const float* dy, int act, int M, int N,
                         int K, float* dx, float* dW, float* db, float* scratch, void* stream) {
  hipStream_t s = S(stream);
  CVAE_REQUIRE(x && W && dy && dW && db, "linear_backward: null operand");
  if (linear_bwd_small_supported(M, N, K)) {           // training-batch heads: one launch for everything
    CVAE_REQUIRE(act == ACT_NONE || y, "linear_backward: act needs y");
    return launch_linear_bwd_small(M, N, K, act, x, W, y, dy, dx, dW, db, s);
  }
  const float* dpre = dy;
  if (act != ACT_NONE) {
    CVAE_REQUIRE(y && scratch, "linear_backward: act needs y and scratch");
    CVAE_TRY(launch_act_bwd((long)M * N, act, y, dy, scratch, s));
    dpre = scratch;
  }
  CVAE_TRY(grad_weight(dpre, N, x, K, M, N, K, dW, 0, s));
  CVAE_TRY(grad_bias(dpre, M, N, N, db, s));
  if (dx) CVAE_TRY(grad_input(dpre, N, W, K, M, N, K, dx, K, 0, s));
  return 0;
}

// ------------------------------------------------------------------------------ encoder
// ne encoders with the same (B, L, E, H) and cell (a text model has one; the dialogue model's post and
// response encoders are padded to a common L by the caller): all 2*ne directions advance in the same
// launches -- one whole-sequence kernel launch instead of ne.
struct EncFwd {
  const int64_t* ids; const int64_t* len; const float* mask; const int64_t* mask_rows; const float* const* w;
  float *X, *xproj, *hs, *cs, *gates, *last;
};
static int encoder_forward_impl(int ne, int cell, int B, int L, int E, int H, const float* emb, const EncFwd* ee,
                                hipStream_t s) {
  CVAE_REQUIRE(ne >= 1 && 2 * ne <= SKINNY_MAX_BATCH, "encoder_forward: %d encoders (max %d)", ne, SKINNY_MAX_BATCH / 2);
  const int G = gates_of(cell);
  RnnFwd f[SKINNY_MAX_BATCH] = {};
  for (int e = 0; e < ne; ++e) {
    const EncFwd& q = ee[e];
    CVAE_REQUIRE(q.ids && q.len && q.w && q.X && q.xproj && q.hs && q.last, "encoder_forward: null operand");
    CVAE_REQUIRE(cell == CELL_GRU || q.cs, "encoder_forward: LSTM needs cs");
    CVAE_TRY(launch_embed(L, B, E, 0, 0, q.ids, B, emb, nullptr, q.mask, q.mask_rows, q.X, s));
    for (int d = 0; d < 2; ++d) {
      const float* W_ih = q.w[4 * d + 0]; const float* W_hh = q.w[4 * d + 1];
      const float* b_ih = q.w[4 * d + 2]; const float* b_hh = q.w[4 * d + 3];
      float* xp = q.xproj + (long)d * L * B * G * H;
      CVAE_TRY(linear_nt(q.X, W_ih, E, b_ih, ACT_NONE, L * B, G * H, E, xp, s));
      RnnFwd& r = f[2 * e + d];
      r.cell = cell; r.T = L; r.R = B; r.H = H; r.reverse = d;
      r.xproj = xp; r.xmod = B; r.W_hh = W_hh; r.b_hh = b_hh; r.lengths = q.len;
      r.hs = q.hs + (long)d * (L + 1) * B * H;
      r.cs = q.cs ? q.cs + (long)d * (L + 1) * B * H : nullptr;
      r.gates = q.gates ? q.gates + (long)d * L * B * 4 * H : nullptr;
    }
  }
  CVAE_TRY(launch_rnn_forward_multi(f, 2 * ne, s));       // every direction advances in the same launches
  for (int e = 0; e < ne; ++e)
    for (int d = 0; d < 2; ++d) {
      const float* fin = f[2 * e + d].hs + (long)(d ? 0 : L) * B * H;
      CVAE_TRY(launch_copy_cols(B, H, fin, H, 0, ee[e].last, 2L * H, d * H, s));
    }
  return 0;
}

int cvae_encoder_forward(int cell, int B, int L, int E, int H, int64_t V, const int64_t* ids, const int64_t* len,
                         const float* emb, const float* mask, const int64_t* mask_rows, const float* const* w,
                         float* X, float* xproj, float* hs, float* cs, float* gates, float* last, void* stream) {
  CVAE_REQUIRE(cell == CELL_GRU || cell == CELL_LSTM, "encoder_forward: bad cell");
  CVAE_REQUIRE(B > 0 && L > 0 && E > 0 && H > 0 && V > 0, "encoder_forward: bad shape B=%d L=%d E=%d H=%d", B, L, E, H);
  CVAE_REQUIRE(B <= SKINNY_MAX_ROWS, "encoder_forward: B=%d > %d not supported", B, SKINNY_MAX_ROWS);
  CVAE_REQUIRE(emb, "encoder_forward: null operand");
  EncFwd q = {ids, len, mask, mask_rows, w, X, xproj, hs, cs, gates, last};
  return encoder_forward_impl(1, cell, B, L, E, H, emb, &q, S(stream));
}

int cvae_encoder_forward_n(int ne, int cell, int B, int L, int E, int H, int64_t V, const int64_t* const* ids,
                           const int64_t* const* len, const float* emb, const float* const* mask,
                           const int64_t* const* mask_rows, const float* const* w, float* const* X,
                           float* const* xproj, float* const* hs, float* const* cs, float* const* gates,
                           float* const* last, void* stream) {
  CVAE_REQUIRE(cell == CELL_GRU || cell == CELL_LSTM, "encoder_forward_n: bad cell");
  CVAE_REQUIRE(ne >= 1 && 2 * ne <= SKINNY_MAX_BATCH && B > 0 && B <= SKINNY_MAX_ROWS && L > 0 && E > 0 && H > 0 && V > 0,
               "encoder_forward_n: bad shape ne=%d B=%d L=%d E=%d H=%d", ne, B, L, E, H);
  CVAE_REQUIRE(ids && len && emb && mask && mask_rows && w && X && xproj && hs && cs && gates && last,
               "encoder_forward_n: null pointer array");
  EncFwd q[SKINNY_MAX_BATCH / 2];
  for (int e = 0; e < ne; ++e)
    q[e] = EncFwd{ids[e], len[e], mask[e], mask_rows[e], w + 8 * e, X[e], xproj[e], hs[e], cs[e], gates[e], last[e]};
  return encoder_forward_impl(ne, cell, B, L, E, H, emb, q, S(stream));
}

int64_t cvae_encoder_backward_scratch(int cell, int B, int L, int E, int H) {
  const int64_t G = gates_of(cell);
  //   per direction: dgx, dgh [L,B,G*H], dh_last, dh0, dh_tmp, dc0, dc_tmp [B,H], W_hh^T [H,G*H]; shared dX [L,B,E]
  return 2 * (2 * (int64_t)L * B * G * H + 5 * (int64_t)B * H + G * (int64_t)H * H) + (int64_t)L * B * E;
}

struct EncBwd {
  const int64_t* ids; const int64_t* len; const float* mask; const int64_t* mask_rows; const float* const* w;
  const float *X, *hs, *cs, *gates, *dlast;
  float* const* dw; float* scratch;
};
static int encoder_backward_impl(int ne, int cell, int B, int L, int E, int H, const EncBwd* ee, float* demb,
                                 hipStream_t s, long V) {
  CVAE_REQUIRE(ne >= 1 && 2 * ne <= SKINNY_MAX_BATCH, "encoder_backward: %d encoders (max %d)", ne, SKINNY_MAX_BATCH / 2);
  const int G = gates_of(cell);
  const long LBG = (long)L * B * G * H, BH = (long)B * H;
  const long per_dir = 2 * LBG + 5 * BH + (long)G * H * H;
  RnnBwd r[SKINNY_MAX_BATCH] = {};
  float* dgx[SKINNY_MAX_BATCH]; float* dgh[SKINNY_MAX_BATCH];
  for (int e = 0; e < ne; ++e) {
    const EncBwd& q = ee[e];
    CVAE_REQUIRE(q.ids && q.len && q.w && q.X && q.hs && q.gates && q.dlast && q.dw && q.scratch,
                 "encoder_backward: null operand");
    for (int d = 0; d < 2; ++d) {
      const int i = 2 * e + d;
      float* base = q.scratch + d * per_dir;
      dgx[i] = base; dgh[i] = base + LBG;
      float* dh_last = base + 2 * LBG;
      float* dh0 = dh_last + BH; float* dh_tmp = dh0 + BH; float* dc0 = dh_tmp + BH; float* dc_tmp = dc0 + BH;
      float* whhT = dc_tmp + BH;
      CVAE_TRY(launch_copy_cols(B, H, q.dlast, 2L * H, d * H, dh_last, H, 0, s));
      r[i].cell = cell; r[i].T = L; r[i].R = B; r[i].H = H; r[i].reverse = d;
      r[i].dh_last = dh_last; r[i].hs = q.hs + (long)d * (L + 1) * B * H;
      r[i].cs = q.cs ? q.cs + (long)d * (L + 1) * B * H : nullptr;
      r[i].gates = q.gates + (long)d * L * B * 4 * H; r[i].W_hh = q.w[4 * d + 1]; r[i].lengths = q.len;
      r[i].dgx = dgx[i]; r[i].dgh = (cell == CELL_GRU) ? dgh[i] : nullptr;
      r[i].dh0 = dh0; r[i].dc0 = dc0; r[i].dh_tmp = dh_tmp; r[i].dc_tmp = dc_tmp; r[i].whhT = whhT;
    }
  }
  CVAE_TRY(launch_rnn_backward_multi(r, 2 * ne, s));
  // padded steps (t >= len[b]) have exactly zero pre-activation gradients: leave them out
  const bool use_live = live_rows_enabled() && (long)L * B >= 1024;
  const long per_list = 16 + (long)L * B;
  int* buf = nullptr;
  if (use_live) CVAE_TRY(lib_rows(ne * per_list, &buf, s));
  LiveRows live[SKINNY_MAX_BATCH];
  const LiveRows* lr[SKINNY_MAX_BATCH];
  float* dX[SKINNY_MAX_BATCH];
  const bool dbh_sep = cell == CELL_GRU && !gemm_wide_has_spare_column(H);
  ZeroBatch zb(s);
  for (int e = 0; e < ne; ++e) {
    const EncBwd& q = ee[e];
    dX[e] = q.scratch + 2 * per_dir;
    lr[e] = nullptr;
    if (use_live) {
      int* lb = buf + e * per_list;
      CVAE_TRY(launch_live_rows_tm(L, B, q.len, 0, lb + 16, lb, s));
      live[e].rows = lb + 16; live[e].n = lb;
      lr[e] = &live[e];
      zb.add(dX[e], (long)L * B * E);
    }
    if (dbh_sep) { zb.add(q.dw[3], (long)G * H); zb.add(q.dw[7], (long)G * H); }
  }
  CVAE_TRY(zb.flush());
  // the products of one kind of both directions (and encoders) share a launch (launch_gemm_n)
  GemmDesc gd[SKINNY_MAX_BATCH];
  for (int i = 0; i < 2 * ne; ++i) {
    const int e = i >> 1, d = i & 1;
    const float* dgh_eff = (cell == CELL_GRU) ? dgh[i] : dgx[i];
    const float* hprev = r[i].hs + (d ? BH : 0);          // h_{t-1} of step t: forward -> slot t, reverse -> slot t+1
    gd[i] = desc_grad_weight(dgh_eff, (long)G * H, hprev, H, L * B, G * H, H, ee[e].dw[4 * d + 1], 0,
                             (cell == CELL_GRU && !dbh_sep) ? ee[e].dw[4 * d + 3] : nullptr, lr[e]);
  }
  CVAE_TRY(launch_gemm_n(gd, 2 * ne, s));
  for (int i = 0; i < 2 * ne; ++i) {
    const int e = i >> 1, d = i & 1;
    gd[i] = desc_grad_weight(dgx[i], (long)G * H, ee[e].X, E, L * B, G * H, E, ee[e].dw[4 * d + 0], 0, ee[e].dw[4 * d + 2], lr[e]);
  }
  CVAE_TRY(launch_gemm_n(gd, 2 * ne, s));
  for (int i = 0; i < 2 * ne; ++i) {
    const int e = i >> 1, d = i & 1;
    // LSTM: both bias vectors see the same pre-activation gradient
    if (cell != CELL_GRU) CVAE_TRY(launch_copy(ee[e].dw[4 * d + 3], ee[e].dw[4 * d + 2], (long)G * H, s));
    else if (dbh_sep) CVAE_TRY(launch_colsum_acc(dgh[i], L * B, G * H, (long)G * H, ee[e].dw[4 * d + 3], s));
  }
  for (int d = 0; d < 2; ++d) {                           // direction 1 adds to direction 0's dX
    for (int e = 0; e < ne; ++e)
      gd[e] = desc_grad_input(dgx[2 * e + d], (long)G * H, ee[e].w[4 * d + 0], E, L * B, G * H, E, dX[e], E, d, lr[e]);
    CVAE_TRY(launch_gemm_n(gd, ne, s));
  }
  for (int e = 0; e < ne; ++e)
    CVAE_TRY(launch_embed_scatter(L, B, E, E, 0, ee[e].ids, dX[e], ee[e].mask, ee[e].mask_rows, demb, s, V));
  return 0;
}

int cvae_encoder_backward(int cell, int B, int L, int E, int H, int64_t V, const int64_t* ids, const int64_t* len,
                          const float* mask, const int64_t* mask_rows, const float* const* w, const float* X,
                          const float* hs, const float* cs, const float* gates, const float* dlast,
                          float* const* dw, float* demb, float* scratch, void* stream) {
  CVAE_REQUIRE(cell == CELL_GRU || cell == CELL_LSTM, "encoder_backward: bad cell");
  CVAE_REQUIRE(B > 0 && L > 0 && E > 0 && H > 0 && V > 0, "encoder_backward: bad shape");
  CVAE_REQUIRE(demb, "encoder_backward: null operand");
  EncBwd q = {ids, len, mask, mask_rows, w, X, hs, cs, gates, dlast, dw, scratch};
  return encoder_backward_impl(1, cell, B, L, E, H, &q, demb, S(stream), (long)V);
}

int cvae_encoder_backward_n(int ne, int cell, int B, int L, int E, int H, int64_t V, const int64_t* const* ids,
                            const int64_t* const* len, const float* const* mask, const int64_t* const* mask_rows,
                            const float* const* w, const float* const* X, const float* const* hs,
                            const float* const* cs, const float* const* gates, const float* const* dlast,
                            float* const* dw, float* demb, float* const* scratch, void* stream) {
  CVAE_REQUIRE(cell == CELL_GRU || cell == CELL_LSTM, "encoder_backward_n: bad cell");
  CVAE_REQUIRE(ne >= 1 && 2 * ne <= SKINNY_MAX_BATCH && B > 0 && L > 0 && E > 0 && H > 0 && V > 0,
               "encoder_backward_n: bad shape ne=%d B=%d L=%d E=%d H=%d", ne, B, L, E, H);
  CVAE_REQUIRE(ids && len && mask && mask_rows && w && X && hs && cs && gates && dlast && dw && demb && scratch,
               "encoder_backward_n: null pointer array");
  EncBwd q[SKINNY_MAX_BATCH / 2];
  for (int e = 0; e < ne; ++e)
    q[e] = EncBwd{ids[e], len[e], mask[e], mask_rows[e], w + 8 * e, X[e], hs[e], cs[e], gates[e], dlast[e],
                  dw + 8 * e, scratch[e]};
  return encoder_backward_impl(ne, cell, B, L, E, H, q, demb, S(stream), (long)V);
}

// ------------------------------------------------------------------------------ decoder
struct DecTrainFwd {
  const float* init; const float* mask_in; const float* mask_out; const float* const* w;
  float *X, *xproj, *hs, *cs, *gates, *hdrop, *final;
};

// nd decoders with identical shapes (the stochastic and the coupled deterministic one) advance
// in the same launches: their recurrences are independent.
// skip_prefix (the LM's decoder, CV/Modules/lm.py:78-100): no prefix step, h_0 = 0.  The recurrence then
// covers internal steps 1..T starting from a zeroed slot 1; step 0's buffers are left unused.
static int decoder_train_forward(int nd, int cell, int B, int T, int E, int Hd, const int64_t* go, const float* emb,
                                 const DecTrainFwd* dd, float* hp, hipStream_t s, int skip_prefix = 0) {
  const int G = gates_of(cell);
  const long BH = (long)B * Hd, GH = (long)G * Hd;
  RnnFwd f[SKINNY_MAX_BATCH] = {};
  for (int d = 0; d < nd; ++d) {
    const DecTrainFwd& q = dd[d];
    CVAE_REQUIRE(q.init && q.w && q.X && q.xproj && q.hs && q.gates && q.final, "decoder_forward(train): null operand");
    CVAE_REQUIRE(cell == CELL_GRU || q.cs, "decoder_forward: LSTM needs cs");
    CVAE_REQUIRE(q.mask_out == nullptr || q.hdrop, "decoder_forward(train): hdrop missing");
    CVAE_TRY(launch_embed(T, B, E, 1, 1, go, B, emb, q.init, q.mask_in, nullptr, q.X, s));
    CVAE_TRY(linear_nt(q.X, q.w[0], 2L * E, q.w[2], ACT_NONE, (T + 1) * B, (int)GH, 2 * E, q.xproj, s));
    f[d].cell = cell; f[d].T = T + 1; f[d].R = B; f[d].H = Hd; f[d].xproj = q.xproj; f[d].xmod = B;
    f[d].W_hh = q.w[1]; f[d].b_hh = q.w[3]; f[d].hs = q.hs; f[d].cs = q.cs; f[d].gates = q.gates; f[d].hp_scratch = hp;
    if (skip_prefix) {
      CVAE_TRY(launch_zero(q.hs, 2 * BH, s));
      if (q.cs) CVAE_TRY(launch_zero(q.cs, 2 * BH, s));
      CVAE_TRY(launch_zero(q.gates, (long)B * 4 * Hd, s));
      f[d].T = T; f[d].xproj = q.xproj + (long)B * GH; f[d].hs = q.hs + BH; f[d].cs = q.cs ? q.cs + BH : nullptr;
      f[d].gates = q.gates + (long)B * 4 * Hd; f[d].init_in_place = 1;
    }
  }
  CVAE_TRY(launch_rnn_forward_multi(f, nd, s));
  for (int d = 0; d < nd; ++d) {
    const DecTrainFwd& q = dd[d];
    const float* hout = q.hs + 2 * BH;
    if (q.mask_out) {
      CVAE_TRY(launch_mul((long)T * BH, hout, q.mask_out, q.hdrop, s));
      hout = q.hdrop;
    }
    CVAE_TRY(linear_nt(hout, q.w[4], Hd, q.w[5], ACT_TANH, T * B, E, Hd, q.final, s));
  }
  return 0;
}

int cvae_decoder_forward(int cell, int train, int B, int R, int T, int E, int Hd, int64_t V, const float* init,
                         const int64_t* go, const float* emb, const float* mask_in, const float* mask_out,
                         const float* const* w, float* X, float* xproj, float* xadd, float* hp, float* hs,
                         float* cs, float* gates, float* hdrop, float* final, void* stream) {
  hipStream_t s = S(stream);
  CVAE_REQUIRE(cell == CELL_GRU || cell == CELL_LSTM, "decoder_forward: bad cell");
  CVAE_REQUIRE(B > 0 && R > 0 && T > 0 && E > 0 && Hd > 0 && V > 0 && R % B == 0,
               "decoder_forward: bad shape B=%d R=%d T=%d E=%d Hd=%d", B, R, T, E, Hd);
  CVAE_REQUIRE(init && go && emb && w && X && xproj && hs && final, "decoder_forward: null operand");
  CVAE_REQUIRE(cell == CELL_GRU || cs, "decoder_forward: LSTM needs cs");
  CVAE_REQUIRE(R <= SKINNY_MAX_ROWS || hp, "decoder_forward: hp scratch required for R > %d", SKINNY_MAX_ROWS);
  const int G = gates_of(cell);
  const float* W_ih = w[0]; const float* W_hh = w[1]; const float* b_ih = w[2]; const float* b_hh = w[3];
  const float* W_c = w[4]; const float* b_c = w[5];
  const long RH = (long)R * Hd, GH = (long)G * Hd;
  if (train) {
    CVAE_REQUIRE(R == B, "decoder_forward(train): R must equal B");
    DecTrainFwd q = {init, mask_in, mask_out, w, X, xproj, hs, cs, gates, hdrop, final};
    return decoder_train_forward(1, cell, B, T, E, Hd, go, emb, &q, hp, s);
  }
  CVAE_REQUIRE(!mask_in && !mask_out, "decoder_forward(eval): dropout masks given");
  CVAE_REQUIRE(xadd, "decoder_forward(eval): xadd missing");
  // x-side pre-activations, de-duplicated:
  //   steps 1..T : W_ih[:, :E] emb(go_t) + b_ih (per sentence)  +  W_ih[:, E:] init (per row, time-invariant)
  //   step 0     : W_ih[:, :E] init + b_ih (per row)            +  the same time-invariant part
  float* xinit = xadd;              // [R, G*Hd]
  float* x0 = xadd + (long)R * GH;  // [R, G*Hd]
  CVAE_TRY(launch_embed(T, B, E, 0, 0, go, B, emb, nullptr, nullptr, nullptr, X, s));
  CVAE_TRY(linear_nt(X, W_ih, 2L * E, b_ih, ACT_NONE, T * B, (int)GH, E, xproj, s));
  CVAE_TRY(linear_nt(init, W_ih + E, 2L * E, nullptr, ACT_NONE, R, (int)GH, E, xinit, s));
  CVAE_TRY(linear_nt(init, W_ih, 2L * E, b_ih, ACT_NONE, R, (int)GH, E, x0, s));
  RnnFwd f0 = {};
  f0.cell = cell; f0.T = 1; f0.R = R; f0.H = Hd; f0.xproj = x0; f0.xmod = R; f0.xadd = xinit;
  f0.W_hh = W_hh; f0.b_hh = b_hh; f0.hs = hs; f0.cs = cs; f0.hp_scratch = hp;
  CVAE_TRY(launch_rnn_forward(f0, s));
  RnnFwd f = f0;
  f.T = T; f.xproj = xproj; f.xmod = B; f.hs = hs + RH; f.cs = cs ? cs + RH : nullptr;
  f.init_in_place = 1;                               // slot 0 of this view = h_0 from the prefix step
  CVAE_TRY(launch_rnn_forward(f, s));
  return linear_nt(hs + 2 * RH, W_c, Hd, b_c, ACT_TANH, T * R, E, Hd, final, s);
}

// Importance-weighted evaluation layout: rows are sentence-major (r = b*K + k, K samples of each
// of B sentences) and the caller has ordered the sentences by non-increasing target length, so
// the rows still inside their sentence at step t are the prefix [0, live_sent[t]*K).  Each step's
// recurrent product, cell update and W_c projection run on that prefix only; rows of hs/cs/final
// past it are left untouched (their targets are <pad>, which cvae_projce_forward never reads).
// Same arithmetic per live row as cvae_decoder_forward(train=0).  live_sent is a HOST array.
int cvae_decoder_forward_sorted(int cell, int B, int K, int T, int E, int Hd, int64_t V, const float* init,
                                const int64_t* go, const float* emb, const int* live_sent, const float* const* w,
                                float* X, float* xproj, float* xadd, float* hp, float* hs, float* cs, float* final,
                                void* stream) {
  hipStream_t s = S(stream);
  CVAE_REQUIRE(cell == CELL_GRU || cell == CELL_LSTM, "decoder_forward_sorted: bad cell");
  CVAE_REQUIRE(B > 0 && K > 0 && T > 0 && E > 0 && Hd > 0 && V > 0 && (long)B * K <= INT_MAX,
               "decoder_forward_sorted: bad shape B=%d K=%d T=%d E=%d Hd=%d", B, K, T, E, Hd);
  const int R = B * K;
  CVAE_REQUIRE(R > SKINNY_MAX_ROWS, "decoder_forward_sorted: needs B*K > %d rows (use cvae_decoder_forward)",
               SKINNY_MAX_ROWS);
  CVAE_REQUIRE(init && go && emb && live_sent && w && X && xproj && xadd && hp && hs && final,
               "decoder_forward_sorted: null operand");
  CVAE_REQUIRE(cell == CELL_GRU || cs, "decoder_forward_sorted: LSTM needs cs");
  std::vector<int> live(T);
  for (int t = 0; t < T; ++t) {
    CVAE_REQUIRE(live_sent[t] >= 0 && live_sent[t] <= B && (t == 0 || live_sent[t] <= live_sent[t - 1]),
                 "decoder_forward_sorted: live_sent must be non-increasing within [0,B] (t=%d: %d)", t, live_sent[t]);
    live[t] = live_sent[t] * K;
  }
  const int G = gates_of(cell);
  const float* W_ih = w[0]; const float* W_hh = w[1]; const float* b_ih = w[2]; const float* b_hh = w[3];
  const float* W_c = w[4]; const float* b_c = w[5];
  const long RH = (long)R * Hd, GH = (long)G * Hd;
  float* xinit = xadd;              // [R, G*Hd]
  float* x0 = xadd + (long)R * GH;  // [R, G*Hd]
  CVAE_TRY(launch_embed(T, B, E, 0, 0, go, B, emb, nullptr, nullptr, nullptr, X, s));
  CVAE_TRY(linear_nt(X, W_ih, 2L * E, b_ih, ACT_NONE, T * B, (int)GH, E, xproj, s));
  CVAE_TRY(linear_nt(init, W_ih + E, 2L * E, nullptr, ACT_NONE, R, (int)GH, E, xinit, s));
  CVAE_TRY(linear_nt(init, W_ih, 2L * E, b_ih, ACT_NONE, R, (int)GH, E, x0, s));
  RnnFwd f0 = {};
  f0.cell = cell; f0.T = 1; f0.R = R; f0.H = Hd; f0.xproj = x0; f0.xmod = R; f0.xadd = xinit;
  f0.W_hh = W_hh; f0.b_hh = b_hh; f0.hs = hs; f0.cs = cs; f0.hp_scratch = hp;
  CVAE_TRY(launch_rnn_forward(f0, s));
  RnnFwd f = f0;
  f.T = T; f.xproj = xproj; f.xmod = B; f.xdiv = K; f.live_rows = live.data();
  f.hs = hs + RH; f.cs = cs ? cs + RH : nullptr;
  f.init_in_place = 1;
  CVAE_TRY(launch_rnn_forward(f, s));
  // final_t = tanh(W_c h_t) for the live rows of every step: one batched launch (blockIdx.z = t)
  if (T <= GEMM_MAX_BATCH) {
    GemmDesc d = {};
    d.transA = 0; d.transB = 1; d.N = E; d.K = Hd;
    d.A = hs + 2 * RH; d.lda = Hd; d.B = W_c; d.ldb = Hd; d.C = final; d.ldc = E; d.bias = b_c; d.act = ACT_TANH;
    d.nbatch = T; d.batchA = RH; d.batchC = (long)R * E; d.host_mbatch = live.data();
    return launch_gemm(d, s);
  }
  for (int t = 0; t < T; ++t)
    if (live[t] > 0)
      CVAE_TRY(linear_nt(hs + (t + 2) * RH, W_c, Hd, b_c, ACT_TANH, live[t], E, Hd, final + (long)t * R * E, s));
  return 0;
}

// ---- f3: greedy branch of Decoder.forward (CV/Modules/decoder.py:176-201; vae_couple.py:173-201).
// Step-wise variant of the eval decoder: h_0 from the prefix step, then T times
//   x_t = [emb(y_{t-1}), init] -> RNN step -> tanh(W_c h) -> ToVocab -> mask <unk> -> argmax -> y_t,
// with y_{-1} = first[b].  The time-invariant half of the input projection (W_ih[:, E:] init) is
// computed once.  Everything stays on the device; no host synchronisation between steps.
int64_t cvae_decoder_greedy_scratch(int cell, int B, int T, int E, int Hd, int64_t V) {
  const long G = gates_of(cell), RH = (long)B * Hd;
  // xadd[2,B,G*Hd] + X[B,E] + xp[B,G*Hd] + hs[(T+2),B,Hd] + cs[(T+2),B,Hd] + final[B,E] + logits[B,V] + cur ids (int64)
  return 3 * B * G * Hd + (long)B * E * 2 + 2 * (T + 2) * RH + (long)B * V + 2L * B + 64;
}
int cvae_decoder_greedy(int cell, int B, int T, int E, int Hd, int64_t V, const float* init, const int64_t* first,
                        const float* emb, const float* const* w, const float* W_v, const float* b_v, int mask_id,
                        float* ws, int64_t* out_ids, void* stream) {
  hipStream_t s = S(stream);
  CVAE_REQUIRE(cell == CELL_GRU || cell == CELL_LSTM, "decoder_greedy: bad cell");
  CVAE_REQUIRE(B > 0 && B <= SKINNY_MAX_ROWS && T > 0 && E > 0 && Hd > 0 && V > 0 && V <= INT_MAX,
               "decoder_greedy: bad shape B=%d T=%d E=%d Hd=%d (B <= %d)", B, T, E, Hd, SKINNY_MAX_ROWS);
  CVAE_REQUIRE(init && first && emb && w && W_v && b_v && ws && out_ids, "decoder_greedy: null operand");
  const int G = gates_of(cell);
  const float* W_ih = w[0]; const float* W_hh = w[1]; const float* b_ih = w[2]; const float* b_hh = w[3];
  const float* W_c = w[4]; const float* b_c = w[5];
  const long RH = (long)B * Hd, GH = (long)G * Hd;
  float* xinit = ws;                        // [B, G*Hd]  W_ih[:, E:] init
  float* x0 = xinit + B * GH;               // [B, G*Hd]  W_ih[:, :E] init + b_ih (prefix step)
  float* xp = x0 + B * GH;                  // [B, G*Hd]  W_ih[:, :E] emb(y_{t-1}) + b_ih
  float* X = xp + B * GH;                   // [B, E]
  float* fin = X + (long)B * E;             // [B, E]
  float* hs = fin + (long)B * E;            // [(T+2), B, Hd]
  float* cs = hs + (T + 2) * RH;            // [(T+2), B, Hd]
  float* logits = cs + (T + 2) * RH;        // [B, V]
  uintptr_t cp = reinterpret_cast<uintptr_t>(logits + (long)B * V);
  int64_t* cur = reinterpret_cast<int64_t*>((cp + 15) & ~uintptr_t(15));    // [B]
  CVAE_TRY(linear_nt(init, W_ih + E, 2L * E, nullptr, ACT_NONE, B, (int)GH, E, xinit, s));
  CVAE_TRY(linear_nt(init, W_ih, 2L * E, b_ih, ACT_NONE, B, (int)GH, E, x0, s));
  RnnFwd f = {};
  f.cell = cell; f.T = 1; f.R = B; f.H = Hd; f.xproj = x0; f.xmod = B; f.xadd = xinit;
  f.W_hh = W_hh; f.b_hh = b_hh; f.hs = hs; f.cs = cell == CELL_LSTM ? cs : nullptr;
  CVAE_TRY(launch_rnn_forward(f, s));                                      // h_0 (slot 1) from zeros (slot 0)
  f.xproj = xp; f.init_in_place = 1;
  CVAE_TRY(launch_copy(reinterpret_cast<float*>(cur), reinterpret_cast<const float*>(first), 2L * B, s));
  for (int t = 0; t < T; ++t) {
    CVAE_TRY(launch_embed(1, B, E, 0, 0, cur, B, emb, nullptr, nullptr, nullptr, X, s));
    CVAE_TRY(linear_nt(X, W_ih, 2L * E, b_ih, ACT_NONE, B, (int)GH, E, xp, s));
    f.hs = hs + (t + 1) * RH; f.cs = cell == CELL_LSTM ? cs + (t + 1) * RH : nullptr;
    CVAE_TRY(launch_rnn_forward(f, s));
    CVAE_TRY(linear_nt(hs + (t + 2) * RH, W_c, Hd, b_c, ACT_TANH, B, E, Hd, fin, s));
    CVAE_TRY(linear_nt(fin, W_v, E, b_v, ACT_NONE, B, (int)V, E, logits, s));
    CVAE_TRY(launch_argmax_rows(B, (int)V, logits, V, mask_id, out_ids, T, t, cur, s));
  }
  return 0;
}

// ---- f3: beam branch of Decoder.forward (CV/Modules/decoder.py:114-168; vae_couple.py:122-172).
// The reference keeps `beam` states of B rows each and loops over them in Python; here the K beams are
// rows k*B + b of one K*B-row batch, so a step is one pass of the same kernels as the greedy branch plus
// top-K, the per-sentence selection and a gather of the chosen states.  No host synchronisation.
// Step 0 has a single source state (B rows).  Sentences never end early (the reference decodes
// max_len tokens whatever they are).  The reference's branch only runs with the GRU (it indexes the
// state as one tensor); the LSTM here carries (h, c) the same way.
static long beam_scratch_floats(int cell, int B, int K, int T, int E, int Hd, int64_t V) {
  const long G = gates_of(cell), R = (long)K * B;
  // xinit[R,GH] x0[B,GH] xp[R,GH] X[R,E] fin[R,E] hs 2x[2,R,Hd] cs 2x[2,R,Hd] logits[R,V] lv[R,K] nll 2x[R]
  // | int: li[R,K] src[R] bp[T,R] tok[T,R] | int64: cur[R]
  return R * G * Hd * 2 + B * G * Hd + 2 * R * E + 8 * R * Hd + R * V + R * K + 2 * R + R * K + R + 2L * T * R +
         2 * R + 64;
}
int64_t cvae_decoder_beam_scratch(int cell, int B, int K, int T, int E, int Hd, int64_t V) {
  return beam_scratch_floats(cell, B, K, T, E, Hd, V);
}
int cvae_decoder_beam(int cell, int B, int K, int T, int E, int Hd, int64_t V, const float* init, const int64_t* first,
                      const float* emb, const float* const* w, const float* W_v, const float* b_v, int mask_id,
                      float* ws, int64_t* out_ids, float* out_nll, void* stream) {
  hipStream_t s = S(stream);
  CVAE_REQUIRE(cell == CELL_GRU || cell == CELL_LSTM, "decoder_beam: bad cell");
  CVAE_REQUIRE(B > 0 && K >= 2 && K <= 8 && (long)K * B <= SKINNY_MAX_ROWS && T > 0 && E > 0 && Hd > 0 && V > K &&
                   V <= INT_MAX,
               "decoder_beam: bad shape B=%d K=%d T=%d E=%d Hd=%d (2 <= K <= 8, K*B <= %d)", B, K, T, E, Hd,
               SKINNY_MAX_ROWS);
  CVAE_REQUIRE(init && first && emb && w && W_v && b_v && ws && out_ids, "decoder_beam: null operand");
  const int G = gates_of(cell), R = K * B;
  const float* W_ih = w[0]; const float* W_hh = w[1]; const float* b_ih = w[2]; const float* b_hh = w[3];
  const float* W_c = w[4]; const float* b_c = w[5];
  const long GH = (long)G * Hd, RH = (long)R * Hd;
  float* xinit = ws;                         // [K][B, G*Hd]  W_ih[:, E:] init, one copy per beam
  float* x0 = xinit + R * GH;                // [B, G*Hd]     W_ih[:, :E] init + b_ih (prefix step)
  float* xp = x0 + B * GH;                   // [R, G*Hd]
  float* X = xp + R * GH;                    // [R, E]
  float* fin = X + (long)R * E;              // [R, E]
  float* hsb[2] = {fin + (long)R * E, fin + (long)R * E + 2 * RH};       // 2 x [2, R, Hd]
  float* csb[2] = {hsb[1] + 2 * RH, hsb[1] + 4 * RH};
  float* logits = csb[1] + 2 * RH;           // [R, V]
  float* lv = logits + (long)R * V;          // [R, K]
  float* nll[2] = {lv + (long)R * K, lv + (long)R * K + R};
  int* li = reinterpret_cast<int*>(nll[1] + R);                          // [R, K]
  int* src = li + (long)R * K;               // [R]
  int* bp = src + R;                         // [T, R]
  int* tok = bp + (long)T * R;               // [T, R]
  uintptr_t cp = reinterpret_cast<uintptr_t>(tok + (long)T * R);
  int64_t* cur = reinterpret_cast<int64_t*>((cp + 15) & ~uintptr_t(15));  // [R]
  const bool lstm = cell == CELL_LSTM;

  CVAE_TRY(linear_nt(init, W_ih + E, 2L * E, nullptr, ACT_NONE, B, (int)GH, E, xinit, s));
  for (int k = 1; k < K; ++k) CVAE_TRY(launch_copy(xinit + k * B * GH, xinit, B * GH, s));
  CVAE_TRY(linear_nt(init, W_ih, 2L * E, b_ih, ACT_NONE, B, (int)GH, E, x0, s));
  RnnFwd f = {};
  f.cell = cell; f.T = 1; f.R = B; f.H = Hd; f.xproj = x0; f.xmod = B; f.xadd = xinit;
  f.W_hh = W_hh; f.b_hh = b_hh; f.hs = hsb[0]; f.cs = lstm ? csb[0] : nullptr;
  CVAE_TRY(launch_rnn_forward(f, s));        // h_0 (B rows at offset B*Hd) from zeros
  f.xproj = xp; f.init_in_place = 1;
  CVAE_TRY(launch_copy(reinterpret_cast<float*>(cur), reinterpret_cast<const float*>(first), 2L * B, s));
  CVAE_TRY(launch_zero(nll[0], 2L * R, s));
  for (int t = 0; t < T; ++t) {
    const int ns = t == 0 ? 1 : K, Rt = ns * B;
    // step 0 continues in buffer 0 behind the prefix step (input = its output slot); later steps
    // read slot 0 of buffer t & 1 (written by the previous gather) and write its slot 1
    float* h_in = t == 0 ? hsb[0] + (long)B * Hd : hsb[t & 1];
    float* c_in = t == 0 ? csb[0] + (long)B * Hd : csb[t & 1];
    float* h_out = h_in + (long)Rt * Hd;
    float* c_out = c_in + (long)Rt * Hd;
    CVAE_TRY(launch_embed(1, Rt, E, 0, 0, cur, Rt, emb, nullptr, nullptr, nullptr, X, s));
    CVAE_TRY(linear_nt(X, W_ih, 2L * E, b_ih, ACT_NONE, Rt, (int)GH, E, xp, s));
    f.R = Rt; f.xmod = Rt; f.hs = h_in; f.cs = lstm ? c_in : nullptr;
    CVAE_TRY(launch_rnn_forward(f, s));
    CVAE_TRY(linear_nt(h_out, W_c, Hd, b_c, ACT_TANH, Rt, E, Hd, fin, s));
    CVAE_TRY(linear_nt(fin, W_v, E, b_v, ACT_NONE, Rt, (int)V, E, logits, s));
    CVAE_TRY(launch_logsoftmax_topk(Rt, (int)V, K, logits, V, mask_id, lv, li, s));
    CVAE_TRY(launch_beam_merge(B, K, ns, t, lv, li, nll[t & 1], nll[(t + 1) & 1], src, cur, bp, tok, s));
    CVAE_TRY(launch_beam_gather(B, K, Hd, src, h_out, hsb[(t + 1) & 1], s));
    if (lstm) CVAE_TRY(launch_beam_gather(B, K, Hd, src, c_out, csb[(t + 1) & 1], s));
  }
  CVAE_TRY(launch_beam_backtrack(B, K, T, bp, tok, nll[T & 1], out_ids, out_nll, s));
  return 0;
}

int cvae_decoder_forward_n(int nd, int cell, int B, int T, int E, int Hd, int64_t V, const float* const* init,
                           const int64_t* go, const float* emb, const float* const* mask_in,
                           const float* const* mask_out, const float* const* w, float* const* X,
                           float* const* xproj, float* const* hs, float* const* cs, float* const* gates,
                           float* const* hdrop, float* const* final, void* stream) {
  CVAE_REQUIRE(nd >= 1 && nd <= SKINNY_MAX_BATCH, "decoder_forward_n: nd=%d (max %d)", nd, SKINNY_MAX_BATCH);
  CVAE_REQUIRE(cell == CELL_GRU || cell == CELL_LSTM, "decoder_forward_n: bad cell");
  CVAE_REQUIRE(B > 0 && B <= SKINNY_MAX_ROWS && T > 0 && E > 0 && Hd > 0 && V > 0, "decoder_forward_n: bad shape");
  CVAE_REQUIRE(init && go && emb && w && X && xproj && hs && gates && final, "decoder_forward_n: null operand");
  DecTrainFwd q[SKINNY_MAX_BATCH];
  for (int d = 0; d < nd; ++d)
    q[d] = {init[d], mask_in ? mask_in[d] : nullptr, mask_out ? mask_out[d] : nullptr, w + 6 * d, X[d], xproj[d],
            hs[d], cs ? cs[d] : nullptr, gates[d], hdrop ? hdrop[d] : nullptr, final[d]};
  return decoder_train_forward(nd, cell, B, T, E, Hd, go, emb, q, nullptr, S(stream));
}

int64_t cvae_decoder_backward_scratch(int cell, int B, int T, int E, int Hd) {
  const int64_t G = gates_of(cell), T1 = T + 1;
  // dpre [T,B,E] + dhs [(T+1),B,Hd] + dgx,dgh [(T+1),B,G*Hd] + dX [(T+1),B,2E] + 4 x [B,Hd] + W_hh^T
  return (int64_t)T * B * E + T1 * B * Hd + 2 * T1 * B * G * Hd + T1 * B * 2 * E + 4 * (int64_t)B * Hd +
         G * (int64_t)Hd * Hd;
}

struct DecBwd {
  const float* mask_in; const float* mask_out; const float* const* w;
  const float *X, *hs, *cs, *gates, *hdrop, *final, *dfinal;
  float* dinit; float* const* dw; float* scratch;
};

static int decoder_backward_impl(int nd, int cell, int B, int T, int E, int Hd, const int64_t* go,
                                 const DecBwd* dd, float* demb, hipStream_t s, long V, int skip_prefix = 0) {
  CVAE_REQUIRE(nd >= 1 && nd <= SKINNY_MAX_BATCH, "decoder_backward: %d decoders (max %d)", nd, SKINNY_MAX_BATCH);
  const int G = gates_of(cell), T1 = T + 1;
  const long BH = (long)B * Hd, GH = (long)G * Hd;
  RnnBwd r[SKINNY_MAX_BATCH] = {};
  float* dgx[SKINNY_MAX_BATCH]; float* dgh[SKINNY_MAX_BATCH]; float* dX[SKINNY_MAX_BATCH];
  float* dpre[SKINNY_MAX_BATCH]; float* dhs[SKINNY_MAX_BATCH]; const float* hout[SKINNY_MAX_BATCH];
  LiveRows live_w[SKINNY_MAX_BATCH], live_r[SKINNY_MAX_BATCH];
  const LiveRows* lrw[SKINNY_MAX_BATCH]; const LiveRows* lrr[SKINNY_MAX_BATCH];
  GemmDesc gd[SKINNY_MAX_BATCH];
  const float* nz_x[SKINNY_MAX_BATCH]; int64_t* nz_za[SKINNY_MAX_BATCH];
  LiveRowsJob jobs[2 * SKINNY_MAX_BATCH];
  const bool use_live = live_rows_enabled() && (long)T * B >= 1024 && B <= SKINNY_MAX_ROWS;
  const long per_dec = (4L * B + 16 + (long)T * B + (long)T1 * B + 3) / 4 * 4;     // ints per decoder
  int* rows_buf = nullptr;
  if (use_live) {
    CVAE_TRY(lib_rows((long)nd * per_dec, &rows_buf, s));   // shared with the encoder's list: calls are stream-ordered
  }
  // The bias gradients ride in the weight-gradient products as a ones column where the tiling has a spare column
  // (gemm_wide), else they are a pass of their own over the pre-activation gradient.
  const bool dbc_sep = !gemm_wide_has_spare_column(Hd);
  const bool dbh_sep = cell == CELL_GRU && !gemm_wide_has_spare_column(Hd);
  ZeroBatch zb(s);                                          // every buffer of this call that must start at zero
  for (int d = 0; d < nd; ++d) {
    const DecBwd& q = dd[d];
    CVAE_REQUIRE(q.w && q.X && q.hs && q.gates && q.final && q.dfinal && q.dinit && q.dw && q.scratch,
                 "decoder_backward: null operand");
    CVAE_REQUIRE(!q.mask_out || q.hdrop, "decoder_backward: hdrop missing");
    dpre[d] = q.scratch;
    dhs[d] = dpre[d] + (long)T * B * E;
    dgx[d] = dhs[d] + (long)T1 * BH;
    dgh[d] = dgx[d] + (long)T1 * B * GH;
    dX[d] = dgh[d] + (long)T1 * B * GH;
    float* dh0 = dX[d] + (long)T1 * B * 2 * E;
    float* dh_tmp = dh0 + BH; float* dc0 = dh_tmp + BH; float* dc_tmp = dc0 + BH;
    float* whhT = dc_tmp + BH;
    hout[d] = q.mask_out ? q.hdrop : q.hs + 2 * BH;
    // Rows (t, b) past sentence b's last target receive an exactly zero dfinal row (the loss ignores
    // <pad>): find each sentence's last non-zero step and leave the dead rows out of every product.
    lrw[d] = lrr[d] = nullptr;
    if (use_live) {
      int* buf = rows_buf + (long)d * per_dec;
      int64_t* za = reinterpret_cast<int64_t*>(buf);            // [B] last live target step + 1
      int64_t* za1 = za + B;                                    // [B] the same in recurrence steps (prefix step first)
      int* cnt = buf + 4 * B;
      zb.add(reinterpret_cast<float*>(za), 2L * B);
      nz_x[d] = q.dfinal; nz_za[d] = za;
      jobs[2 * d] = LiveRowsJob{T, 0, za, cnt + 16, cnt, nullptr};
      jobs[2 * d + 1] = LiveRowsJob{T1, 1, za, cnt + 16 + (long)T * B, cnt + 1, za1};
      live_w[d].rows = cnt + 16; live_w[d].n = cnt;
      live_r[d].rows = cnt + 16 + (long)T * B; live_r[d].n = cnt + 1;
      lrw[d] = &live_w[d]; lrr[d] = &live_r[d];
      r[d].zero_after = za1;
    }
    if (dbc_sep) zb.add(q.dw[5], E);
    if (dbh_sep) zb.add(q.dw[3], GH);
    zb.add(dhs[d], use_live ? (long)T1 * BH : (long)BH);
    if (lrr[d]) zb.add(dX[d], (long)T1 * B * 2 * E);
    r[d].cell = cell; r[d].T = T1; r[d].R = B; r[d].H = Hd; r[d].reverse = 0;
    r[d].dhs = dhs[d]; r[d].hs = q.hs; r[d].cs = q.cs; r[d].gates = q.gates; r[d].W_hh = q.w[1];
    r[d].dgx = dgx[d]; r[d].dgh = (cell == CELL_GRU) ? dgh[d] : nullptr;
    r[d].dh0 = dh0; r[d].dc0 = dc0; r[d].dh_tmp = dh_tmp; r[d].dc_tmp = dc_tmp; r[d].whhT = whhT;
    if (skip_prefix) {      // internal steps 1..T only; step 0 contributes nothing
      zb.add(dgx[d], (long)B * GH);
      if (cell == CELL_GRU) zb.add(dgh[d], (long)B * GH);
      r[d].T = T; r[d].dhs = dhs[d] + BH; r[d].hs = q.hs + BH; r[d].cs = q.cs ? q.cs + BH : nullptr;
      r[d].gates = q.gates + (long)B * 4 * Hd; r[d].dgx = dgx[d] + (long)B * GH;
      r[d].dgh = (cell == CELL_GRU) ? dgh[d] + (long)B * GH : nullptr;
      if (r[d].zero_after) r[d].zero_after -= B;          // za (target steps) instead of za + 1
    }
  }
  CVAE_TRY(zb.flush());
  if (use_live) {                                           // one scan and one list-building launch for all decoders
    CVAE_TRY(launch_last_nonzero_step_n(nd, T, B, E, nz_x, nz_za, s));
    CVAE_TRY(launch_live_rows_tm_n(jobs, 2 * nd, B, s));
  }
  // final = tanh(hout W_c^T + b_c)
  for (int d = 0; d < nd; ++d)
    CVAE_TRY(launch_act_bwd((long)T * B * E, ACT_TANH, dd[d].final, dd[d].dfinal, dpre[d], s));
  // the decoders' products of one kind share a launch (launch_gemm_n)
  for (int d = 0; d < nd; ++d)
    gd[d] = desc_grad_weight(dpre[d], E, hout[d], Hd, T * B, E, Hd, dd[d].dw[4], 0, dbc_sep ? nullptr : dd[d].dw[5], lrw[d]);
  CVAE_TRY(launch_gemm_n(gd, nd, s));
  if (dbc_sep)
    for (int d = 0; d < nd; ++d) CVAE_TRY(launch_colsum_acc(dpre[d], T * B, E, E, dd[d].dw[5], s));
  for (int d = 0; d < nd; ++d)
    gd[d] = desc_grad_input(dpre[d], E, dd[d].w[4], Hd, T * B, E, Hd, dhs[d] + BH, Hd, 0, lrw[d]);
  CVAE_TRY(launch_gemm_n(gd, nd, s));
  for (int d = 0; d < nd; ++d)
    if (dd[d].mask_out) CVAE_TRY(launch_mul((long)T * BH, dhs[d] + BH, dd[d].mask_out, dhs[d] + BH, s));
  CVAE_TRY(launch_rnn_backward_multi(r, nd, s));
  for (int d = 0; d < nd; ++d) {
    const float* dgh_eff = (cell == CELL_GRU) ? dgh[d] : dgx[d];
    gd[d] = desc_grad_weight(dgh_eff, GH, dd[d].hs, Hd, T1 * B, (int)GH, Hd, dd[d].dw[1], 0,
                             (cell == CELL_GRU && !dbh_sep) ? dd[d].dw[3] : nullptr, lrr[d]);
  }
  CVAE_TRY(launch_gemm_n(gd, nd, s));
  for (int d = 0; d < nd; ++d)
    gd[d] = desc_grad_weight(dgx[d], GH, dd[d].X, 2L * E, T1 * B, (int)GH, 2 * E, dd[d].dw[0], 0, dd[d].dw[2], lrr[d]);
  CVAE_TRY(launch_gemm_n(gd, nd, s));
  for (int d = 0; d < nd; ++d) {
    // LSTM: both bias vectors see the same pre-activation gradient
    if (cell != CELL_GRU) CVAE_TRY(launch_copy(dd[d].dw[3], dd[d].dw[2], GH, s));
    else if (dbh_sep) CVAE_TRY(launch_colsum_acc(dgh[d], T1 * B, (int)GH, GH, dd[d].dw[3], s));
  }
  for (int d = 0; d < nd; ++d)
    gd[d] = desc_grad_input(dgx[d], GH, dd[d].w[0], 2L * E, T1 * B, (int)GH, 2 * E, dX[d], 2L * E, 0, lrr[d]);
  CVAE_TRY(launch_gemm_n(gd, nd, s));
  for (int d = 0; d < nd; ++d) {
    CVAE_TRY(launch_dec_dinit(T, B, E, dX[d], dd[d].mask_in, dd[d].dinit, s));
    CVAE_TRY(launch_embed_scatter(T, B, E, 2 * E, 1, go, dX[d], dd[d].mask_in, nullptr, demb, s, V));
  }
  return 0;
}

int cvae_decoder_backward(int cell, int B, int T, int E, int Hd, int64_t V, const int64_t* go,
                          const float* mask_in, const float* mask_out, const float* const* w, const float* X,
                          const float* hs, const float* cs, const float* gates, const float* hdrop,
                          const float* final, const float* dfinal, float* dinit, float* const* dw, float* demb,
                          float* scratch, void* stream) {
  CVAE_REQUIRE(cell == CELL_GRU || cell == CELL_LSTM, "decoder_backward: bad cell");
  CVAE_REQUIRE(B > 0 && T > 0 && E > 0 && Hd > 0 && V > 0, "decoder_backward: bad shape");
  CVAE_REQUIRE(go && demb, "decoder_backward: null operand");
  DecBwd q = {mask_in, mask_out, w, X, hs, cs, gates, hdrop, final, dfinal, dinit, dw, scratch};
  return decoder_backward_impl(1, cell, B, T, E, Hd, go, &q, demb, S(stream), (long)V);
}

// LM decoder (CV/Modules/lm.py:78-100): the training branch without the prefix step (h_0 = 0).  The caller
// passes init = 0 and W_ih = [W_ih_lm | 0], which makes the conditioning half of the input vanish.
int cvae_decoder_forward_noprefix(int cell, int B, int T, int E, int Hd, int64_t V, const float* init,
                                  const int64_t* go, const float* emb, const float* mask_in, const float* mask_out,
                                  const float* const* w, float* X, float* xproj, float* hs, float* cs, float* gates,
                                  float* hdrop, float* final, void* stream) {
  CVAE_REQUIRE(cell == CELL_GRU || cell == CELL_LSTM, "decoder_forward_noprefix: bad cell");
  CVAE_REQUIRE(B > 0 && B <= SKINNY_MAX_ROWS && T > 0 && E > 0 && Hd > 0 && V > 0, "decoder_forward_noprefix: bad shape");
  CVAE_REQUIRE(init && go && emb && w, "decoder_forward_noprefix: null operand");
  DecTrainFwd q = {init, mask_in, mask_out, w, X, xproj, hs, cs, gates, hdrop, final};
  return decoder_train_forward(1, cell, B, T, E, Hd, go, emb, &q, nullptr, S(stream), 1);
}
int cvae_decoder_backward_noprefix(int cell, int B, int T, int E, int Hd, int64_t V, const int64_t* go,
                                   const float* mask_in, const float* mask_out, const float* const* w, const float* X,
                                   const float* hs, const float* cs, const float* gates, const float* hdrop,
                                   const float* final, const float* dfinal, float* dinit, float* const* dw,
                                   float* demb, float* scratch, void* stream) {
  CVAE_REQUIRE(cell == CELL_GRU || cell == CELL_LSTM, "decoder_backward_noprefix: bad cell");
  CVAE_REQUIRE(B > 0 && T > 0 && E > 0 && Hd > 0 && V > 0, "decoder_backward_noprefix: bad shape");
  CVAE_REQUIRE(go && demb, "decoder_backward_noprefix: null operand");
  DecBwd q = {mask_in, mask_out, w, X, hs, cs, gates, hdrop, final, dfinal, dinit, dw, scratch};
  return decoder_backward_impl(1, cell, B, T, E, Hd, go, &q, demb, S(stream), (long)V, 1);
}

int cvae_decoder_backward_n(int nd, int cell, int B, int T, int E, int Hd, int64_t V, const int64_t* go,
                            const float* const* mask_in, const float* const* mask_out, const float* const* w,
                            const float* const* X, const float* const* hs, const float* const* cs,
                            const float* const* gates, const float* const* hdrop, const float* const* final,
                            const float* const* dfinal, float* const* dinit, float* const* dw, float* demb,
                            float* const* scratch, void* stream) {
  CVAE_REQUIRE(nd >= 1 && nd <= SKINNY_MAX_BATCH, "decoder_backward_n: nd=%d (max %d)", nd, SKINNY_MAX_BATCH);
  CVAE_REQUIRE(cell == CELL_GRU || cell == CELL_LSTM, "decoder_backward_n: bad cell");
  CVAE_REQUIRE(B > 0 && B <= SKINNY_MAX_ROWS && T > 0 && E > 0 && Hd > 0 && V > 0, "decoder_backward_n: bad shape");
  CVAE_REQUIRE(go && w && X && hs && gates && final && dfinal && dinit && dw && demb && scratch,
               "decoder_backward_n: null operand");
  DecBwd q[SKINNY_MAX_BATCH];
  for (int d = 0; d < nd; ++d)
    q[d] = {mask_in ? mask_in[d] : nullptr, mask_out ? mask_out[d] : nullptr, w + 6 * d, X[d], hs[d],
            cs ? cs[d] : nullptr, gates[d], hdrop ? hdrop[d] : nullptr, final[d], dfinal[d], dinit[d], dw + 6 * d,
            scratch[d]};
  return decoder_backward_impl(nd, cell, B, T, E, Hd, go, q, demb, S(stream), (long)V);
}

// ------------------------------------------------------------------------------ projection + CE
int64_t cvae_projce_workspace(int64_t N, int64_t E, int64_t V) {
  if (ce_fast_supported((int)E)) return ce_fast_workspace_floats(N, (int)V);
  return projce_workspace_floats((int)N, (int)V);
}

int64_t cvae_projce_logits_workspace(int64_t N, int64_t E, int64_t V) {
  return ce_fast_supported((int)E) ? ce_fast_logits_floats(N, (int)V) : 0;
}

static int projce_forward_impl(const float* final, const float* W, const float* b, const int64_t* tgt, int64_t pad,
                               int nseg, int T, int R, int E, int64_t V, float* loss_row, float* lse, float* sent,
                               float* ws, float* logits, void* stream) {
  hipStream_t s = S(stream);
  CVAE_REQUIRE(nseg >= 1 && T >= 0 && R >= 0 && (int64_t)nseg * T * R < (1LL << 31) && V > 0 && V < (1LL << 31),
               "projce_forward: bad shape");
  const long TR = (long)T * R, N = TR * nseg;
  if (N == 0) return 0;
  if (ce_fast_supported(E)) {
    CVAE_TRY(launch_ce_fast_forward(final, W, b, tgt, pad, N, TR, E, (int)V, loss_row, lse, ws, logits, s));
  } else {
    for (int g = 0; g < nseg; ++g)
      CVAE_TRY(launch_projce_forward(final + g * TR * E, W, b, tgt, pad, (int)TR, E, (int)V, loss_row + g * TR,
                                     lse + g * TR, ws, s));
  }
  if (sent)
    for (int g = 0; g < nseg; ++g) CVAE_TRY(launch_sum_time(loss_row + g * TR, T, R, sent + (long)g * R, s));
  return 0;
}

int cvae_projce_forward(const float* final, const float* W, const float* b, const int64_t* tgt, int64_t pad,
                        int nseg, int T, int R, int E, int64_t V, float* loss_row, float* lse, float* sent,
                        float* ws, void* stream) {
  return projce_forward_impl(final, W, b, tgt, pad, nseg, T, R, E, V, loss_row, lse, sent, ws, nullptr, stream);
}
int cvae_projce_forward_train(const float* final, const float* W, const float* b, const int64_t* tgt, int64_t pad,
                              int nseg, int T, int R, int E, int64_t V, float* loss_row, float* lse, float* sent,
                              float* ws, float* logits, void* stream) {
  CVAE_REQUIRE(!logits || ce_fast_supported(E), "projce_forward_train: logits are only saved for E <= 200");
  return projce_forward_impl(final, W, b, tgt, pad, nseg, T, R, E, V, loss_row, lse, sent, ws, logits, stream);
}

// nd decoders over the same targets (Decoder + DecoderCouple of one training step), saved-logits path, E <= 200
int cvae_projce_forward_train_n(int nd, const float* const* final, const float* const* W, const float* const* b,
                                const int64_t* tgt, int64_t pad, int T, int R, int E, int64_t V,
                                float* const* loss_row, float* const* lse, float* const* sent, float* const* ws,
                                float* const* logits, void* stream) {
  hipStream_t s = S(stream);
  CVAE_REQUIRE(nd >= 1 && nd <= SKINNY_MAX_BATCH && T >= 0 && R >= 0 && (int64_t)T * R < (1LL << 31) && V > 0 &&
               V < (1LL << 31) && ce_fast_supported(E), "projce_forward_train_n: bad shape");
  CVAE_REQUIRE(final && W && b && tgt && loss_row && lse && sent && ws && logits, "projce_forward_train_n: null operand");
  const long N = (long)T * R;
  if (N == 0) return 0;
  CVAE_TRY(launch_ce_fast_forward_n(nd, final, W, b, tgt, pad, N, N, E, (int)V, loss_row, lse, ws, logits, s));
  for (int d = 0; d < nd; ++d) {
    CVAE_REQUIRE(sent[d], "projce_forward_train_n: null operand");
    CVAE_TRY(launch_sum_time(loss_row[d], T, R, sent[d], s));
  }
  return 0;
}
int64_t cvae_projce_backward_saved_n_workspace(int64_t E) {
  return ce_fast_supported((int)E) ? gemm_wide_part_floats(200, true) + 32 : 0;
}
int cvae_projce_backward_saved_n(int nd, const float* const* final, const float* const* W, const float* const* b,
                                 const int64_t* tgt, int64_t pad, const float* const* lse, const float* const* gsent,
                                 int T, int R, int E, int64_t V, float* const* dfinal, float* const* dW,
                                 float* const* db, float* ws, float* const* logits, void* stream) {
  CVAE_REQUIRE(nd >= 1 && nd <= SKINNY_MAX_BATCH && T >= 0 && R >= 0 && (int64_t)T * R < (1LL << 31) && V > 0 &&
               V < (1LL << 31) && ce_fast_supported(E), "projce_backward_saved_n: bad shape");
  const long N = (long)T * R;
  return launch_ce_fast_backward_n(nd, final, W, b, tgt, pad, lse, gsent, N, N, R, E, (int)V, dfinal, dW, db, ws,
                                   logits, S(stream));
}

int cvae_split_bf16(int on) { return split_bf16_set(on); }
int cvae_deterministic(int on) { return deterministic_set(on); }
int64_t cvae_stream_rnn_min_bytes(int64_t bytes) { return stream_rnn_min_weight_bytes(bytes); }

int64_t cvae_projce_backward_workspace(int64_t N, int64_t E, int64_t V) {
  if (ce_fast_supported((int)E)) return ce_fast_bwd_workspace_floats(N, (int)V);
  return projce_bwd_workspace_floats((int)N, (int)V) + N;   // + broadcast of gsent over time
}

static int projce_backward_impl(const float* final, const float* W, const float* b, const int64_t* tgt, int64_t pad,
                                const float* lse, const float* gsent, int nseg, int T, int R, int E, int64_t V,
                                float* dfinal, float* dW, float* db, float* ws, float* logits, void* stream) {
  hipStream_t s = S(stream);
  CVAE_REQUIRE(nseg >= 1 && T >= 0 && R >= 0 && (int64_t)nseg * T * R < (1LL << 31) && V > 0 && V < (1LL << 31),
               "projce_backward: bad shape");
  CVAE_REQUIRE(gsent && ws && dW && db, "projce_backward: null operand");
  const long TR = (long)T * R, N = TR * nseg;
  if (ce_fast_supported(E))
    return launch_ce_fast_backward(final, W, b, tgt, pad, lse, gsent, N, TR, R, E, (int)V, dfinal, dW, db, ws, logits, s);
  CVAE_TRY(launch_zero(dW, (long)V * E, s));
  CVAE_TRY(launch_zero(db, (long)V, s));
  for (int g = 0; g < nseg; ++g) {
    float* grow = ws;
    CVAE_TRY(launch_bcast_time(gsent + (long)g * R, T, R, grow, s));
    CVAE_TRY(launch_projce_backward(final + g * TR * E, W, b, tgt, pad, lse + g * TR, grow, (int)TR, E, (int)V,
                                    dfinal + g * TR * E, dW, db, ws + TR, s));
  }
  return 0;
}

int cvae_projce_backward(const float* final, const float* W, const float* b, const int64_t* tgt, int64_t pad,
                         const float* lse, const float* gsent, int nseg, int T, int R, int E, int64_t V,
                         float* dfinal, float* dW, float* db, float* ws, void* stream) {
  return projce_backward_impl(final, W, b, tgt, pad, lse, gsent, nseg, T, R, E, V, dfinal, dW, db, ws, nullptr, stream);
}
int cvae_projce_backward_saved(const float* final, const float* W, const float* b, const int64_t* tgt, int64_t pad,
                               const float* lse, const float* gsent, int nseg, int T, int R, int E, int64_t V,
                               float* dfinal, float* dW, float* db, float* ws, float* logits, void* stream) {
  CVAE_REQUIRE(logits && ce_fast_supported(E), "projce_backward_saved: needs the logits saved by projce_forward_train");
  return projce_backward_impl(final, W, b, tgt, pad, lse, gsent, nseg, T, R, E, V, dfinal, dW, db, ws, logits, stream);
}

int cvae_targets_time_major(const int64_t* eos, int B, int T, int R, int64_t* tgt_tm, void* stream) {
  CVAE_REQUIRE(B > 0 && R % B == 0, "targets_time_major: R must be a multiple of B");
  return launch_targets_tm(eos, B, T, R, tgt_tm, S(stream));
}

// sentence-major variant: row r = b*K + k reads sentence b
int cvae_targets_sentence_major(const int64_t* eos, int B, int T, int K, int64_t* tgt_tm, void* stream) {
  CVAE_REQUIRE(B > 0 && K > 0 && eos && tgt_tm, "targets_sentence_major: bad arguments");
  return launch_targets_sm(eos, B, T, K, tgt_tm, S(stream));
}

int cvae_logits(const float* final, const float* W, const float* b, int T, int B, int E, int64_t V, float* tmp,
                float* logits, void* stream) {
  hipStream_t s = S(stream);
  CVAE_TRY(linear_nt(final, W, E, b, ACT_NONE, T * B, (int)V, E, tmp, s));
  return launch_logits_bt(T, B, (long)V, tmp, logits, s);
}

// ------------------------------------------------------------------------------ latent space
int cvae_reparam_forward(const float* mu, const float* sigma, const float* eps, int n, int B, int Z, float* z,
                         void* stream) {
  CVAE_REQUIRE(mu && sigma && eps && z, "reparam_forward: null operand");
  return launch_reparam_fwd(mu, sigma, eps, n, B, Z, z, S(stream));
}
int cvae_reparam_backward(const float* dz, const float* eps, int B, int Z, float* dmu, float* dsigma,
                          void* stream) {
  CVAE_REQUIRE(dz && eps && dmu && dsigma, "reparam_backward: null operand");
  return launch_reparam_bwd(dz, eps, B, Z, dmu, dsigma, S(stream));
}
int cvae_kl_forward(const float* mu, const float* sigma, const float* mu_p, const float* sigma_p, int B, int Z,
                    float* kl, void* stream) {
  CVAE_REQUIRE(mu && sigma && kl, "kl_forward: null operand");
  return launch_kl_fwd(mu, sigma, mu_p, sigma_p, B, Z, kl, S(stream));
}
int cvae_kl_backward(const float* mu, const float* sigma, const float* mu_p, const float* sigma_p,
                     const float* gkl, int B, int Z, float* dmu, float* dsigma, float* dmu_p, float* dsigma_p,
                     void* stream) {
  CVAE_REQUIRE(mu && sigma && gkl && dmu && dsigma, "kl_backward: null operand");
  return launch_kl_bwd(mu, sigma, mu_p, sigma_p, gkl, B, Z, dmu, dsigma, dmu_p, dsigma_p, S(stream));
}
int cvae_loss_assemble(int n, int B, const float* const* terms, const float* host_coef, float* out, void* stream) {
  return launch_loss_assemble(n, B, terms, host_coef, out, S(stream));
}
int cvae_loss_assemble_backward(int n, int B, const float* gtot, const float* host_coef, float* const* dterms,
                                void* stream) {
  return launch_loss_assemble_bwd(n, B, gtot, host_coef, dterms, S(stream));
}
int cvae_kl_fb_forward(const float* mu, const float* sigma, int B, int Z, float floor, float* kl, void* stream) {
  CVAE_REQUIRE(mu && sigma && kl, "kl_fb_forward: null operand");
  return launch_kl_fb_fwd(mu, sigma, B, Z, floor, kl, S(stream));
}
int cvae_kl_fb_backward(const float* mu, const float* sigma, const float* gkl, int B, int Z, float floor,
                        float* dmu, float* dsigma, void* stream) {
  CVAE_REQUIRE(mu && sigma && gkl && dmu && dsigma, "kl_fb_backward: null operand");
  return launch_kl_fb_bwd(mu, sigma, gkl, B, Z, floor, dmu, dsigma, S(stream));
}
int cvae_couple_forward(const float* init, const float* init_c, int B, int E, int Z, int euclid, float* d,
                        float* loss, void* stream) {
  CVAE_REQUIRE(init && init_c && d && loss, "couple_forward: null operand");
  return launch_couple_fwd(init, init_c, B, E, Z, euclid, d, loss, S(stream));
}
int cvae_couple_backward(const float* init, const float* init_c, const float* d, const float* gloss, int B,
                         int E, int Z, int euclid, float* dinit, void* stream) {
  CVAE_REQUIRE(init && init_c && d && gloss && dinit, "couple_backward: null operand");
  return launch_couple_bwd(init, init_c, d, gloss, B, E, Z, euclid, dinit, S(stream));
}
int cvae_iw_reduce(const float* z, const float* xent, const float* mu, const float* sigma, const float* mu_p,
                   const float* sigma_p, int K, int B, int Z, float* nll, float* kl, float* xent_mean,
                   void* stream) {
  CVAE_REQUIRE(z && xent && mu && sigma && nll && kl && xent_mean, "iw_reduce: null operand");
  return launch_iw_reduce(z, xent, mu, sigma, mu_p, sigma_p, K, B, Z, nll, kl, xent_mean, S(stream));
}
int cvae_mi(const float* z, const float* mu, const float* sigma, int K, int B, int Z, float* ws, float* mi,
               void* stream) {
  CVAE_REQUIRE(z && mu && sigma && ws && mi, "mi: null operand");
  return launch_mi(z, mu, sigma, K, B, Z, ws, ws + (long)K * B, mi, S(stream));
}
int64_t cvae_mi_static_max_rows(void) { return mi_static_max_rows(); }
int cvae_mi_static(const float* z, const float* mu, const float* sigma, int K, int B, int Z, float* mi, void* stream) {
  CVAE_REQUIRE(z && mu && sigma && mi, "mi_static: null operand");
  return launch_mi_static(z, mu, sigma, K, B, Z, mi, S(stream));
}
int cvae_clip_adam(float* params, float* grads, float* m, float* v, int64_t n, int64_t alias_off,
                   int64_t alias_n, int alias_mult, float max_norm, float lr, float beta1, float beta2,
                   float eps, int64_t step_other, int64_t step_alias, float* sumsq, float* norm_out,
                   void* stream) {
  return launch_clip_adam(params, grads, m, v, n, alias_off, alias_n, alias_mult, max_norm, lr, beta1, beta2, eps,
                          step_other, step_alias, 0, 0, 0, sumsq, norm_out, S(stream));
}
int cvae_clip_adam_late(float* params, float* grads, float* m, float* v, int64_t n, int64_t alias_off,
                        int64_t alias_n, int alias_mult, float max_norm, float lr, float beta1, float beta2,
                        float eps, int64_t step_other, int64_t step_alias, int64_t late_off, int64_t late_n,
                        int64_t step_late, float* sumsq, float* norm_out, void* stream) {
  return launch_clip_adam(params, grads, m, v, n, alias_off, alias_n, alias_mult, max_norm, lr, beta1, beta2, eps,
                          step_other, step_alias, late_off, late_n, step_late, sumsq, norm_out, S(stream));
}
int cvae_release_stream(void* stream) { return lib_scratch_release(S(stream), false); }
int cvae_shutdown(void) { return lib_scratch_release(nullptr, true); }
int cvae_persist_status_export(float* dst, void* stream) { return launch_status_export(dst, S(stream)); }
int cvae_persist_status_import(const float* src, void* stream) { return launch_status_import(src, S(stream)); }
int cvae_persistent_rnn(int on) { return persist_rnn_set(on); }
int cvae_rnn_large_wreg(int min_rows) { return rnn_large_wreg_set(min_rows); }
int cvae_persist_headroom(int cus) { return persist_headroom_set(cus); }
int cvae_persist_status(void) { return persist_status_and_clear(); }
int cvae_persist_status_peek(void) { return persist_status_peek(); }
int cvae_persist_status_set(int value) {
  CVAE_TRY(persist_status_word(nullptr));
  __atomic_store_n(persist_status_host(), value ? 1 : 0, __ATOMIC_RELEASE);
  return 0;
}

int cvae_sum_time(const float* x, int T, int R, float* out, void* stream) {
  return launch_sum_time(x, T, R, out, S(stream));
}
int cvae_axpy(const float* x, float a, int64_t n, float* y, void* stream) {
  return launch_axpy(n, x, a, y, S(stream));
}

}  // extern "C"
