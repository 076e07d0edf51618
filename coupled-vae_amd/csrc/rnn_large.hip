// One recurrent step for many rows (importance-weighted evaluation, R = samples x sentences):
//   h_t = cell(h_{t-1} W_hh^T + b_hh, x-side pre-activations, h_{t-1} [, c_{t-1}])
// as ONE kernel: the gate product on the matrix cores with the GRU / LSTM cell in its epilogue.
// The unfused form (tiled GEMM into hp[R, G*H], then a pointwise kernel) writes and re-reads
// G*H floats per row and step -- 262 MB per step at R = 16000, H = 512 -- which costs as much as
// the product itself.
//
//  * block tile: 128 rows x 32 hidden units x all G gates.  W_hh rows g*H + u0 + (0..31) of the
//    G gates are staged as G groups of 32 LDS rows, so a wave's 32x32 MFMA block j IS gate j of
//    its 32 units and a lane ends up with all G pre-activations of (row, unit): the cell is
//    lane-local, no shuffles, no LDS round trip;
//  * both operands are k-contiguous ([row][k]); LDS rows of BK + 4 floats make the b128 fragment
//    reads conflict-free; lane (l31, lhi) reads k = 8q + 4*lhi + (0..3) of its row for A and B
//    alike, i.e. both sides see the same permutation of k (v_mfma_f32_32x32x2_f32, 4 steps per read);
//  * double-buffered LDS, one barrier per RL_BK-wide K chunk (16: 354 us per 16 000-row step against 375 us
//    with 32), every global load a whole float4
//    (rows past the live prefix are clamped to a valid row; their results are not stored).
#include "common.hpp"

namespace cvae {

#ifndef RL_BK_
#define RL_BK_ 16
#endif
#ifndef RL_OCC_
#define RL_OCC_ 2
#endif
constexpr int RL_BM = 128, RL_BU = 32, RL_BK = RL_BK_, RL_LD = RL_BK + 4, RL_F4 = RL_BK / 4;   // RL_LD / 4 odd: conflict-free b128

struct RnnLargeArgs {
  const float* h_prev; const float* c_prev; const float* W_hh; const float* b_hh;
  const float* xproj; const float* xadd;      // xproj row (r / xdiv) % xmod, xadd row r (or null)
  float* h_out; float* c_out;
  int live, H, xmod, xdiv;
};

template <int CELL>
__global__ __launch_bounds__(256, RL_OCC_) void rnn_large_step_kernel(RnnLargeArgs p) {
  constexpr int G = CELL == CELL_GRU ? 3 : 4;
  extern __shared__ __attribute__((aligned(16))) float rl_smem[];       // 2 x (A tile + B tile) = 72 KB (LSTM)
  float (*As)[RL_BM * RL_LD] = reinterpret_cast<float (*)[RL_BM * RL_LD]>(rl_smem);
  float (*Bs)[G * RL_BU * RL_LD] = reinterpret_cast<float (*)[G * RL_BU * RL_LD]>(rl_smem + 2 * RL_BM * RL_LD);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, lhi = lane >> 5;
  const int H = p.H, GH = G * H;
  // XCD-aware tile order: workgroups go round-robin over the 8 XCDs, each with its own L2.  All unit
  // tiles of one 128-row tile are given to one XCD (its h rows are then fetched into that L2 once).
  const int nu = H / RL_BU, L = blockIdx.x;
  const int rowtile = ((L >> 3) / nu) * 8 + (L & 7), utile = (L >> 3) % nu;
  const int u0 = utile * RL_BU, m0 = rowtile * RL_BM;
  if (m0 >= p.live) return;

  // staging: A 128 rows x RL_F4 float4, B (G*32) rows x RL_F4 float4; thread -> (row = idx / RL_F4, k4 = idx % RL_F4)
  constexpr int NA = RL_BM * (RL_BK / 4) / 256;          // 4
  constexpr int NBT = G * RL_BU * (RL_BK / 4);           // float4 pieces of the B tile: 384 (GRU) / 512 (LSTM)
  constexpr int NB = (NBT + 255) / 256;                  // 2; the GRU's second pass covers half the block
  constexpr bool NB_EXACT = NB * 256 == NBT;
  // element offsets from the kernel-argument base pointers (kept as offsets so that the loads stay
  // global_load: pointer arrays that are advanced in the loop degrade to flat loads + scratch)
  const float* __restrict__ gA = p.h_prev;
  const float* __restrict__ gB = p.W_hh;
  long oa[NA], ob[NB];
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    const int idx = tid + 256 * i;
    const int row = min(m0 + idx / RL_F4, p.live - 1);
    oa[i] = (long)row * H + 4 * (idx % RL_F4);
  }
#pragma unroll
  for (int i = 0; i < NB; ++i) {
    const int idx = tid + 256 * i;
    const int br = min(idx / RL_F4, G * RL_BU - 1);       // 0 .. G*32-1: gate br / 32, unit br % 32 (GRU: the
                                                          // surplus threads of the last pass re-load the last row)
    ob[i] = (long)((br >> 5) * H + u0 + (br & 31)) * H + 4 * (idx % RL_F4);
  }
  int kload = 0;                                          // k of the next chunk to load
  f32x4 sa[NA], sb[NB];
#define RL_STAGE_LOAD()                                                                          \
  do {                                                                                           \
    _Pragma("unroll") for (int i = 0; i < NA; ++i) {                                             \
      sa[i] = *reinterpret_cast<const f32x4*>(gA + oa[i] + kload);                               \
    }                                                                                            \
    _Pragma("unroll") for (int i = 0; i < NB; ++i) {                                             \
      sb[i] = *reinterpret_cast<const f32x4*>(gB + ob[i] + kload);                               \
    }                                                                                            \
    kload += RL_BK;                                                                              \
  } while (0)
#define RL_STAGE_STORE(buf)                                                                      \
  do {                                                                                           \
    _Pragma("unroll") for (int i = 0; i < NA; ++i) {                                             \
      const int idx = tid + 256 * i;                                                             \
      *reinterpret_cast<f32x4*>(&As[buf][(idx / RL_F4) * RL_LD + 4 * (idx % RL_F4)]) = sa[i];           \
    }                                                                                            \
    _Pragma("unroll") for (int i = 0; i < NB; ++i) {                                             \
      const int idx = tid + 256 * i;                                                             \
      if (NB_EXACT || idx < NBT)                                                                 \
        *reinterpret_cast<f32x4*>(&Bs[buf][(idx / RL_F4) * RL_LD + 4 * (idx % RL_F4)]) = sb[i];  \
    }                                                                                            \
  } while (0)

  f32x16 acc[G];
#pragma unroll
  for (int g = 0; g < G; ++g)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[g][r] = 0.f;

  const int nk = H / RL_BK;
  RL_STAGE_LOAD();
  RL_STAGE_STORE(0);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) RL_STAGE_LOAD();
    const float* ar = &As[buf][(wave * 32 + l31) * RL_LD + 4 * lhi];
    const float* br = &Bs[buf][l31 * RL_LD + 4 * lhi];
#pragma unroll
    for (int q = 0; q < RL_BK / 8; ++q) {
      const float4 a = *reinterpret_cast<const float4*>(ar + 8 * q);
      float4 b[G];
#pragma unroll
      for (int g = 0; g < G; ++g) b[g] = *reinterpret_cast<const float4*>(br + g * RL_BU * RL_LD + 8 * q);
#pragma unroll
      for (int g = 0; g < G; ++g) acc[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b[g].x, acc[g], 0, 0, 0);
#pragma unroll
      for (int g = 0; g < G; ++g) acc[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b[g].y, acc[g], 0, 0, 0);
#pragma unroll
      for (int g = 0; g < G; ++g) acc[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b[g].z, acc[g], 0, 0, 0);
#pragma unroll
      for (int g = 0; g < G; ++g) acc[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b[g].w, acc[g], 0, 0, 0);
    }
    if (kt + 1 < nk) RL_STAGE_STORE(buf ^ 1);
    __syncthreads();
  }

  // cell: lane = unit u0 + l31; acc[g][r] = gate g of row m0 + wave*32 + (r&3) + 8*(r>>2) + 4*lhi
  const int u = u0 + l31;
  float bh[G];
#pragma unroll
  for (int g = 0; g < G; ++g) bh[g] = p.b_hh[g * H + u];
  // rows in four groups of four: all of a group's loads are issued before its arithmetic (one memory
  // round trip per group instead of one per row); rows past the live prefix load a clamped row and
  // are simply not stored.
#pragma unroll
  for (int rg = 0; rg < 4; ++rg) {
    float x[4][G], hp[4], cp[4];
    int mm[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int m = m0 + wave * 32 + j + 8 * rg + 4 * lhi;
      mm[j] = m;
      const int mc = min(m, p.live - 1);
      const float* xp = p.xproj + (long)((mc / p.xdiv) % p.xmod) * GH + u;
#pragma unroll
      for (int g = 0; g < G; ++g) x[j][g] = xp[g * H];
      if (p.xadd) {
        const float* xa = p.xadd + (long)mc * GH + u;
#pragma unroll
        for (int g = 0; g < G; ++g) x[j][g] += xa[g * H];
      }
      hp[j] = p.h_prev[(long)mc * H + u];
      cp[j] = (CELL == CELL_LSTM) ? p.c_prev[(long)mc * H + u] : 0.f;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int r = 4 * rg + j;
      float sg[G];
#pragma unroll
      for (int g = 0; g < G; ++g) sg[g] = acc[g][r] + bh[g];
      const long o = (long)mm[j] * H + u;
      if constexpr (CELL == CELL_GRU) {
        const float rgate = sigmoidf_(x[j][0] + sg[0]);
        const float zg = sigmoidf_(x[j][1] + sg[1]);
        const float nn = tanhf_(x[j][2] + rgate * sg[2]);
        if (mm[j] < p.live) p.h_out[o] = (1.f - zg) * nn + zg * hp[j];
      } else {
        const float ig = sigmoidf_(x[j][0] + sg[0]);
        const float fg = sigmoidf_(x[j][1] + sg[1]);
        const float gg = tanhf_(x[j][2] + sg[2]);
        const float og = sigmoidf_(x[j][3] + sg[3]);
        const float cnew = fg * cp[j] + ig * gg;
        if (mm[j] < p.live) {
          p.h_out[o] = og * tanhf_(cnew);
          p.c_out[o] = cnew;
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------------
// The same step with the gate product on the bf16 matrix pipe, error-compensated ("bf16x3", see ce_fast.hip):
// h and W_hh are split into hi + lo bf16 while they are staged (registers -> LDS), and each 16-deep K block costs
// three v_mfma_f32_32x32x16_bf16 per gate (hi.lo, lo.hi, hi.hi) instead of eight v_mfma_f32_32x32x2_f32 at twice the
// cycles each.  Tile and roles as above (128 rows x 32 units x G gates, a wave's N-tile j is gate j: the cell stays
// lane-local); K chunk 32 (two MFMA K blocks per barrier); LDS rows are 32 bf16 = four 16-byte slots, slot index
// XOR-swizzled with (row >> 2) & 3 so that the b128 fragment reads of 16 consecutive rows are conflict-free
// without padding: 2 x (A hi|lo + B hi|lo) = 64 KB, two blocks per CU.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x8 __attribute__((ext_vector_type(8)));
constexpr int RB_BK = 32;

// RB_EXP_* : timing-only diagnostic builds (scripts/build_variant.sh), results are wrong with any of them set
#ifndef RB_EXP_NOSPLIT
#define RB_EXP_NOSPLIT 0
#endif
#ifndef RB_EXP_NOEPI
#define RB_EXP_NOEPI 0
#endif
#ifndef RB_EXP_NOMFMA
#define RB_EXP_NOMFMA 0
#endif
__device__ __forceinline__ void rb_split(const f32x4 a, const f32x4 b, bf16x8& hi, bf16x8& lo) {
  f32x8 v;
  v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3]; v[4] = b[0]; v[5] = b[1]; v[6] = b[2]; v[7] = b[3];
#if RB_EXP_NOSPLIT
  union { f32x8 f; bf16x8 h[2]; } u;
  u.f = v;
  hi = u.h[0]; lo = u.h[1];
#else
  hi = __builtin_convertvector(v, bf16x8);
  lo = __builtin_convertvector(v - __builtin_convertvector(hi, f32x8), bf16x8);
#endif
}

template <int CELL>
__global__ __launch_bounds__(256, 2) void rnn_large_step_bf3_kernel(RnnLargeArgs p) {
  constexpr int G = CELL == CELL_GRU ? 3 : 4;
  constexpr int BROWS = G * RL_BU;                      // rows of the B tile: gate-major, 32 units each
  // LDS image, in bf16x8 (16-byte) slots: [buf][A hi: 128 rows x 4][A lo][B hi: BROWS x 4][B lo]
  constexpr int A_SL = RL_BM * 4, B_SL = BROWS * 4, BUF_SL = 2 * A_SL + 2 * B_SL;
  extern __shared__ __attribute__((aligned(16))) float rl_smem[];
  bf16x8* sm = reinterpret_cast<bf16x8*>(rl_smem);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, lhi = lane >> 5;
  const int H = p.H, GH = G * H;
  const int nu = H / RL_BU, L = blockIdx.x;
  const int rowtile = ((L >> 3) / nu) * 8 + (L & 7), utile = (L >> 3) % nu;
  const int u0 = utile * RL_BU, m0 = rowtile * RL_BM;
  if (m0 >= p.live) return;

  // staging tasks: one task = 8 consecutive k of one row = one LDS slot.  A: 512 tasks, B: BROWS * 4 (384 / 512)
  constexpr int NTA = 2, NTB = (B_SL + 255) / 256;
  constexpr bool NB_EXACT = NTB * 256 == B_SL;
  const float* __restrict__ gA = p.h_prev;
  const float* __restrict__ gB = p.W_hh;
  long oa[NTA], ob[NTB];
  int sa_slot[NTA], sb_slot[NTB];
#pragma unroll
  for (int i = 0; i < NTA; ++i) {
    const int idx = tid + 256 * i, r = idx >> 2, sl = idx & 3;
    oa[i] = (long)min(m0 + r, p.live - 1) * H + 8 * sl;
    sa_slot[i] = r * 4 + (sl ^ ((r >> 2) & 3));
  }
#pragma unroll
  for (int i = 0; i < NTB; ++i) {
    const int idx = min(tid + 256 * i, B_SL - 1), r = idx >> 2, sl = idx & 3;      // r: gate r / 32, unit r % 32
    ob[i] = (long)((r >> 5) * H + u0 + (r & 31)) * H + 8 * sl;
    sb_slot[i] = r * 4 + (sl ^ ((r >> 2) & 3));
  }
  int kload = 0;
  // two register sets: the loads of chunk kt+2 are issued before chunk kt is multiplied and are split / stored to
  // LDS one iteration later (a single set left them one MFMA phase to land: the kernel ran at the latency of its
  // staging loads, not at the matrix pipe's rate)
  f32x4 ra[2][NTA][2], rb[2][NTB][2];
#define RB_STAGE_LOAD(set)                                                                       \
  do {                                                                                           \
    _Pragma("unroll") for (int i = 0; i < NTA; ++i) {                                            \
      ra[set][i][0] = *reinterpret_cast<const f32x4*>(gA + oa[i] + kload);                       \
      ra[set][i][1] = *reinterpret_cast<const f32x4*>(gA + oa[i] + kload + 4);                   \
    }                                                                                            \
    _Pragma("unroll") for (int i = 0; i < NTB; ++i) {                                            \
      rb[set][i][0] = *reinterpret_cast<const f32x4*>(gB + ob[i] + kload);                       \
      rb[set][i][1] = *reinterpret_cast<const f32x4*>(gB + ob[i] + kload + 4);                   \
    }                                                                                            \
    kload += RB_BK;                                                                              \
  } while (0)
#define RB_STAGE_STORE(set, buf)                                                                 \
  do {                                                                                           \
    bf16x8* base = sm + (buf) * BUF_SL;                                                          \
    _Pragma("unroll") for (int i = 0; i < NTA; ++i) {                                            \
      bf16x8 hi, lo;                                                                             \
      rb_split(ra[set][i][0], ra[set][i][1], hi, lo);                                            \
      base[sa_slot[i]] = hi;                                                                     \
      base[A_SL + sa_slot[i]] = lo;                                                              \
    }                                                                                            \
    _Pragma("unroll") for (int i = 0; i < NTB; ++i) {                                            \
      if (NB_EXACT || tid + 256 * i < B_SL) {                                                    \
        bf16x8 hi, lo;                                                                           \
        rb_split(rb[set][i][0], rb[set][i][1], hi, lo);                                          \
        base[2 * A_SL + sb_slot[i]] = hi;                                                        \
        base[2 * A_SL + B_SL + sb_slot[i]] = lo;                                                 \
      }                                                                                          \
    }                                                                                            \
  } while (0)

  f32x16 acc[G];
#pragma unroll
  for (int g = 0; g < G; ++g)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[g][r] = 0.f;

  // fragment slots of this lane: A row wave*32 + l31, B row g*32 + l31 (same swizzle term for all: (l31 >> 2) & 3)
  const int arow = wave * 32 + l31, sw = (l31 >> 2) & 3;
  const int nk = H / RB_BK;
  RB_STAGE_LOAD(0);
  RB_STAGE_STORE(0, 0);
  if (nk > 1) RB_STAGE_LOAD(1);
  __syncthreads();
  for (int kt0 = 0; kt0 < nk; kt0 += 2) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int kt = kt0 + u;
      if (kt < nk) {
        if (kt + 2 < nk) RB_STAGE_LOAD(u);                 // set u held chunk kt, already in LDS
        const bf16x8* base = sm + u * BUF_SL;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
          const int sl = (2 * kb + lhi) ^ sw;
          const bf16x8 ah = base[arow * 4 + sl], al = base[A_SL + arow * 4 + sl];
          bf16x8 bh[G], bl[G];
#pragma unroll
          for (int g = 0; g < G; ++g) {
            bh[g] = base[2 * A_SL + (g * RL_BU + l31) * 4 + sl];
            bl[g] = base[2 * A_SL + B_SL + (g * RL_BU + l31) * 4 + sl];
          }
#if RB_EXP_NOMFMA
#pragma unroll
          for (int g = 0; g < G; ++g) { acc[g][0] += (float)ah[0] * (float)bl[g][0] + (float)al[1] * (float)bh[g][1]; }
#else
#pragma unroll
          for (int g = 0; g < G; ++g) acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl[g], acc[g], 0, 0, 0);
#pragma unroll
          for (int g = 0; g < G; ++g) acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh[g], acc[g], 0, 0, 0);
#pragma unroll
          for (int g = 0; g < G; ++g) acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh[g], acc[g], 0, 0, 0);
#endif
        }
        if (kt + 1 < nk) RB_STAGE_STORE(u ^ 1, u ^ 1);     // chunk kt+1, loaded one iteration ago
        __syncthreads();
      }
    }
  }
#undef RB_STAGE_LOAD
#undef RB_STAGE_STORE

  // cell: identical to rnn_large_step_kernel (lane = unit u0 + l31; acc[g][r] = gate g of row
  // m0 + wave*32 + (r&3) + 8*(r>>2) + 4*lhi)
  const int u = u0 + l31;
  float bh_[G];
#pragma unroll
  for (int g = 0; g < G; ++g) bh_[g] = p.b_hh[g * H + u];
#pragma unroll
  for (int rg = 0; rg < 4; ++rg) {
    float x[4][G], hp[4], cp[4];
    int mm[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int m = m0 + wave * 32 + j + 8 * rg + 4 * lhi;
      mm[j] = m;
      const int mc = min(m, p.live - 1);
#if RB_EXP_NOEPI
#pragma unroll
      for (int g = 0; g < G; ++g) x[j][g] = 0.1f * g;
      hp[j] = 0.5f; cp[j] = 0.25f; (void)mc;
#else
      const float* xp = p.xproj + (long)((mc / p.xdiv) % p.xmod) * GH + u;
#pragma unroll
      for (int g = 0; g < G; ++g) x[j][g] = xp[g * H];
      if (p.xadd) {
        const float* xa = p.xadd + (long)mc * GH + u;
#pragma unroll
        for (int g = 0; g < G; ++g) x[j][g] += xa[g * H];
      }
      hp[j] = p.h_prev[(long)mc * H + u];
      cp[j] = (CELL == CELL_LSTM) ? p.c_prev[(long)mc * H + u] : 0.f;
#endif
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int r = 4 * rg + j;
      float sg[G];
#pragma unroll
      for (int g = 0; g < G; ++g) sg[g] = acc[g][r] + bh_[g];
      const long o = (long)mm[j] * H + u;
      if constexpr (CELL == CELL_GRU) {
        const float rgate = sigmoid_fast(x[j][0] + sg[0]);
        const float zg = sigmoid_fast(x[j][1] + sg[1]);
        const float nn = tanh_fast(x[j][2] + rgate * sg[2]);
        if (mm[j] < p.live) p.h_out[o] = (1.f - zg) * nn + zg * hp[j];
      } else {
        const float ig = sigmoid_fast(x[j][0] + sg[0]);
        const float fg = sigmoid_fast(x[j][1] + sg[1]);
        const float gg = tanh_fast(x[j][2] + sg[2]);
        const float og = sigmoid_fast(x[j][3] + sg[3]);
        const float cnew = fg * cp[j] + ig * gg;
        if (mm[j] < p.live) {
          p.h_out[o] = og * tanh_fast(cnew);
          p.c_out[o] = cnew;
        }
      }
    }
  }
}

bool rnn_large_fused_supported(int cell, int H, const float* h, const float* W_hh) {
  static int on = -1;
  if (on < 0) {
    const char* e = getenv("CVAE_FUSED_LARGE_STEP");
    on = (e && e[0] == '0') ? 0 : 1;
  }
  return on && H % RL_BK == 0 && H % RL_BU == 0 && aligned16(h) && aligned16(W_hh);
}

// rows [0, live) of one step; h_prev / h_out (and c) are [R, H] slabs
int launch_rnn_large_step(int cell, int live, int H, const float* h_prev, const float* c_prev, const float* W_hh,
                          const float* b_hh, const float* xproj, int xmod, int xdiv, const float* xadd, float* h_out,
                          float* c_out, hipStream_t s) {
  if (live <= 0) return 0;
  CVAE_REQUIRE(h_prev && W_hh && b_hh && xproj && h_out && xmod > 0 && xdiv > 0, "rnn_large_step: null operand");
  CVAE_REQUIRE(cell == CELL_GRU || (c_prev && c_out), "rnn_large_step: LSTM needs c");
  CVAE_REQUIRE(H % RL_BK == 0 && H % RL_BU == 0, "rnn_large_step: H=%d must be a multiple of 32", H);
  RnnLargeArgs a = {h_prev, c_prev, W_hh, b_hh, xproj, xadd, h_out, c_out, live, H, xmod, xdiv};
  const int G = cell == CELL_GRU ? 3 : 4;
  const dim3 grid((H / RL_BU) * ((cdiv(live, RL_BM) + 7) / 8 * 8)), block(256);
  const size_t lds = (size_t)2 * (RL_BM + G * RL_BU) * RL_LD * sizeof(float);
  static bool attr_set = false;
  if (!attr_set) {
    CVAE_TRY(check_hip(hipFuncSetAttribute(reinterpret_cast<const void*>(&rnn_large_step_kernel<CELL_GRU>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024), "rnn_large LDS attribute"));
    CVAE_TRY(check_hip(hipFuncSetAttribute(reinterpret_cast<const void*>(&rnn_large_step_kernel<CELL_LSTM>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024), "rnn_large LDS attribute"));
    attr_set = true;
  }
  ProfScope prof(PK_RNN_LARGE, s, 2.0 * live * G * H * (double)H);
  if (split_bf16_enabled() && H % RB_BK == 0) {
    const size_t lds3 = (size_t)2 * (2 * RL_BM * 4 + 2 * G * RL_BU * 4) * 16;      // 64 KB (LSTM) / 56 KB (GRU)
    static bool attr3 = false;
    if (!attr3) {
      CVAE_TRY(check_hip(hipFuncSetAttribute(reinterpret_cast<const void*>(&rnn_large_step_bf3_kernel<CELL_GRU>),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024), "rnn_large bf3 LDS attribute"));
      CVAE_TRY(check_hip(hipFuncSetAttribute(reinterpret_cast<const void*>(&rnn_large_step_bf3_kernel<CELL_LSTM>),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024), "rnn_large bf3 LDS attribute"));
      attr3 = true;
    }
    if (cell == CELL_GRU) hipLaunchKernelGGL(rnn_large_step_bf3_kernel<CELL_GRU>, grid, block, lds3, s, a);
    else hipLaunchKernelGGL(rnn_large_step_bf3_kernel<CELL_LSTM>, grid, block, lds3, s, a);
    CVAE_LAUNCH_CHECK("rnn_large_step_bf3_kernel");
    return 0;
  }
  if (cell == CELL_GRU) hipLaunchKernelGGL(rnn_large_step_kernel<CELL_GRU>, grid, block, lds, s, a);
  else hipLaunchKernelGGL(rnn_large_step_kernel<CELL_LSTM>, grid, block, lds, s, a);
  CVAE_LAUNCH_CHECK("rnn_large_step_kernel");
  return 0;
}

}  // namespace cvae
