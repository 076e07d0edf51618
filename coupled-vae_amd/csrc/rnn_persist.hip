// Whole-sequence GRU / LSTM kernels for training batch sizes (R <= 32 rows per sequence):
// ONE launch runs all T steps of up to four lock-stepped recurrences (the two encoder directions,
// the decoder and its coupled twin), instead of one launch per step.
// Reference semantics: nn.GRU / nn.LSTM as used by CV/Modules/encoder.py:27-39 and
// CV/Modules/vae_couple.py:58-70 (same cell equations as cell_fwd_kernel / cell_bwd_point).
//
// Why: at B = 32 a step is 67 MFLOP; the per-step kernels of rnn.hip / skinny.hip spend ~7.5 us in
// launch, operand fetch (W_hh is re-read from L2 every step) and drain, plus ~1.5 us of kernel
// boundary.  Here every block keeps ITS slice of W_hh resident for the whole sequence and the steps
// are separated by an in-kernel all-gather of the 32 x H state:
//   * forward: grid = (H / 4) blocks per sequence, one block per CU (<= 256 blocks in all): block b owns
//     hidden units 4b .. 4b+3, i.e. 4*G gate columns; backward: H / 8 blocks per sequence, block b owns
//     8 units (half the blocks, flags and exchange volume: rnn_bwd_persist_kernel<CELL, HU>);
//   * forward step:  gates[32 x 4G] = h_{t-1}[32 x H] . W_hh[4G rows]^T with v_mfma_f32_16x16x4_f32,
//     K split over the block's waves (W fragments live in registers), partial tiles summed through
//     LDS, then 128 threads run the cell for their (row, unit) and publish h_t;
//   * backward step: the block owns the gate rows of its units, i.e. a K-slice of dh_{t-1} = dG_t . W_hh:
//     part[32 x H] = dG_t[:, rows of U] . W_hh[rows of U, :] (v_mfma_f32_16x16x4_f32, W slice in registers),
//     reduce-scattered through a double-buffered destination-major exchange; then the cell backward of
//     step t-1 for the block's units (details above rnn_bwd_persist_kernel);
//   * hand-off between blocks (MI355X_MICROARCH.md, inter-workgroup visibility, first table row):
//     payload stored write-through (`sc1` buffer stores), every storing wave drains (vmcnt(0)),
//     workgroup barrier, ONE lane stores the block's epoch flag (sc1); consumers poll the per-block flags
//     with `sc1` loads (one wave, bounded), barrier, then read the payload with `sc1` buffer loads.
//     (Tried and dropped: one atomic counter per sequence -- 128 arrivals serialise at ~12 ns each; and
//     data-tagged {value, epoch} granules polled directly by the consumer waves: 16-byte sc1 buffer loads
//     re-reading a line that had been polled before the store landed kept returning the old copy; 8-byte
//     agent-scope loads were correct but 2x slower than flag + 16-byte payload loads.)
//     Every spin is bounded: on a timeout the block raises the status word (host-mapped pinned memory:
//     cvae_persist_status / _peek read it, cvae_clip_adam refuses to apply a step while it is set) and the
//     whole grid drains.
#include "common.hpp"
#include "misc.hpp"
#include "cell_math.hpp"

namespace cvae {

constexpr int P_HU = 4;              // hidden units per block
constexpr int P_THREADS = 1024;      // 16 waves
constexpr unsigned P_SPIN_LIMIT = 1u << 19;
constexpr int P_MAXB = 128;          // blocks per sequence (H <= 512)
constexpr int P_AUX_SC1 = 16;        // buffer load/store aux bit: sc1 (write-through / L1 bypass)

typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef unsigned u32x4v __attribute__((ext_vector_type(4)));

struct PSeqF {
  const float* xproj; const float* xadd; const float* W_hh; const float* b_hh; const int64_t* lengths;
  float* hs; float* cs; float* gates;
  int xmod, reverse;
};
struct PFwdArgs {
  PSeqF q[SKINNY_MAX_BATCH];
  int T, R, H, nw;                   // nw: waves that take part in the K split (H / nw multiple of 16)
  int poll_delay, poll_delay_cell;   // in-band form: s_sleep(1) periods before a step's first poll (other / cell waves)
  unsigned* flags;                   // [ns][P_MAXB] epoch flags, zero at launch
  int* status;
  unsigned long long* stamps;
};

struct PSeqB {
  const float* dhs; const float* dh_last; const float* dc_last;
  const float* hs; const float* cs; const float* gates; const float* W_hh; const int64_t* lengths;
  float* dgx; float* dgh;            // dgh == dgx for LSTM
  float* dh0; float* dc0;
  int reverse;
  const int64_t* zero_after;         // [R] or null: row r has an exactly zero gradient at steps t >= zero_after[r]
};
struct PBwdArgs {
  PSeqB q[SKINNY_MAX_BATCH];
  int T, R, H;
  unsigned* flags;
  int* status;
  unsigned long long* stamps;
  float* exch;                       // [ns][2][nb][nb][32][4] partial-sum exchange
  int poll_delay;                    // in-band form: s_sleep(1) periods before a step's first poll
};

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, 0x7fffffff, 0x00020000);
}
__device__ __forceinline__ f32x4v ld16_sc1(__amdgpu_buffer_rsrc_t r, long float_off) {
  const u32x4v v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)(float_off * 4), 0, P_AUX_SC1);
  return __builtin_bit_cast(f32x4v, v);
}
__device__ __forceinline__ void st16_sc1(__amdgpu_buffer_rsrc_t r, long float_off, f32x4v v) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4v, v), r, (int)(float_off * 4), 0, P_AUX_SC1);
}

// Per-block epoch flags (no atomics: 128 arrivals on one counter serialise at ~12 ns each).
// All threads call; wave 0 polls the sequence's nb flags (one sc1 load per lane per 64 blocks).
// Returns false (uniformly) on timeout.
__device__ __forceinline__ bool wait_flags(const unsigned* flags, int nb, unsigned epoch, int* status, int* s_abort) {
  if (threadIdx.x < 64) {
    unsigned spins = 0;
    int bad = 0;
    for (;;) {
      int ok = 1;
      for (int k = threadIdx.x; k < nb; k += 64)
        ok &= (__hip_atomic_load(flags + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= epoch) ? 1 : 0;
      if (__all(ok)) break;
      __builtin_amdgcn_s_sleep(1);
      if (++spins > P_SPIN_LIMIT) { bad = 1; break; }
    }
    if (threadIdx.x == 0) {
      if (bad) __hip_atomic_store(status, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);   // host-mapped word
      *s_abort = bad;
    }
  }
  __syncthreads();
  return *s_abort == 0;
}

// Every storing wave has drained; one lane raises this block's flag (sc1 store).  (Raising it from
// the last wave instead, to keep wave 0's back-edge vmcnt drain off the flag store, made the flag
// later for everyone else: +15 % per step.  Lane 0 it is.)
__device__ __forceinline__ void publish(unsigned* flags, unsigned epoch) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_store(flags + blockIdx.x, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// optional phase stamps (diagnostics, CVAE_PERSIST_STAMPS=1): block (0,0) thread 0 accumulates cycles
#define P_STAMP(k)                                                                 \
  do {                                                                             \
    if (stamp_on) {                                                                \
      const unsigned long long now_ = __builtin_amdgcn_s_memtime();                \
      sacc_[k] += (unsigned)(now_ - tprev_);                                       \
      tprev_ = now_;                                                               \
    }                                                                              \
  } while (0)

// Activations of the in-kernel cell: v_exp_f32 + v_rcp_f32 (absolute error ~3e-7, far inside the
// 1e-4 parity bar); the cell sits on the per-step critical path, libm's tanhf alone is ~60 instructions.
__device__ __forceinline__ float fast_sigmoid(float x) { return __frcp_rn(1.f + __expf(-x)); }
__device__ __forceinline__ float fast_tanh(float x) { return 1.f - 2.f * __frcp_rn(1.f + __expf(2.f * x)); }

typedef __bf16 bf16x8p __attribute__((ext_vector_type(8)));
typedef float f32x8p __attribute__((ext_vector_type(8)));
__device__ __forceinline__ void split3(float v, __bf16& h, __bf16& m, __bf16& l) {
  h = (__bf16)v;
  const float r1 = v - (float)h;
  m = (__bf16)r1;
  l = (__bf16)(r1 - (float)m);
}

// --------------------------------------------------------------------------------------- forward
// BF: the gate product on the bf16 matrix pipe, two-term split ("bf16x3", the forward products' form: ce_fast.hip).
// Every wave owns 32 of K = one v_mfma_f32_16x16x32_bf16 K block (NW = H / 32): three MFMAs of 16 cycles per row
// fragment instead of eight v_mfma_f32_16x16x4_f32 of 32; W_hh's fragments are split once, h as it arrives.
template <int CELL, bool BF>
__global__ __launch_bounds__(P_THREADS) void rnn_fwd_persist_kernel(PFwdArgs p) {
  constexpr int G = CELL == CELL_GRU ? 3 : 4;
  constexpr int NC = P_HU * G;                       // live gate columns of this block (12 / 16)
  __shared__ float red[16][32][16];
  __shared__ int s_abort;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, kq = lane >> 4;
  const PSeqF& q = p.q[blockIdx.y];
  const int H = p.H, R = p.R, T = p.T, nb = gridDim.x;
  const int u0 = blockIdx.x * P_HU;
  const int NW = p.nw, KW = H / NW, NSL = KW / 4;    // floats per lane per row fragment (4..16, multiple of 4)
  const long RH = (long)R * H;
  const int GH = G * H;
  unsigned* flg = p.flags + blockIdx.y * P_MAXB;
  const bool stamp_on = p.stamps && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0;
  unsigned long long tprev_ = stamp_on ? __builtin_amdgcn_s_memtime() : 0ull;
  unsigned sacc_[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};          // cycles fit 32 bits for one launch
  const __amdgpu_buffer_rsrc_t rs_hs = make_rsrc(q.hs);
  if (tid == 0) s_abort = 0;

  // W_hh fragments (B operand of 16x16x4: lane = column l15, k group kq), resident for all T steps.
  // Column c = u*G + g of the block is gate row g*H + u0 + u; MFMA step s of this lane uses
  // k = wave*KW + kq*NSL + s (the same permutation of K is applied to h below).
  float w[BF ? 1 : 16];
  bf16x8p wbh, wbl;                                    // BF: k = wave*32 + kq*8 .. +7 of this lane's column (NSL == 8)
#pragma unroll
  for (int s = 0; s < (BF ? 1 : 16); ++s) w[s] = 0.f;
#pragma unroll
  for (int e = 0; e < 8; ++e) { wbh[e] = (__bf16)0.f; wbl[e] = (__bf16)0.f; }
  const int kbase = wave * KW + kq * NSL;
  if (wave < NW && l15 < NC) {
    const int u = l15 / G, g = l15 % G;
    const float* wr = q.W_hh + (long)(g * H + u0 + u) * H + kbase;
    if constexpr (BF) {
      const float4 v0 = *reinterpret_cast<const float4*>(wr), v1 = *reinterpret_cast<const float4*>(wr + 4);
      const f32x8p v = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
      wbh = __builtin_convertvector(v, bf16x8p);
      wbl = __builtin_convertvector(v - __builtin_convertvector(wbh, f32x8p), bf16x8p);
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (4 * j < NSL) {
          const float4 v = *reinterpret_cast<const float4*>(wr + 4 * j);
          w[4 * j] = v.x; w[4 * j + 1] = v.y; w[4 * j + 2] = v.z; w[4 * j + 3] = v.w;
        }
    }
  }

  // cell threads: (row, unit) = (tid >> 2, tid & 3); their state lives in registers
  const bool cellt = tid < 128;
  const int crow = tid >> 2, cu = tid & 3, cn = u0 + cu;
  const bool crv = cellt && crow < R;
  float hprev = 0.f, cprev = 0.f, bh[G], xa[G];
  int64_t len = 0;
#pragma unroll
  for (int g = 0; g < G; ++g) { bh[g] = 0.f; xa[g] = 0.f; }
  const int init_slot = q.reverse ? T : 0;
  if (crv) {
    hprev = q.hs[init_slot * RH + (long)crow * H + cn];
    if (CELL == CELL_LSTM) cprev = q.cs[init_slot * RH + (long)crow * H + cn];
#pragma unroll
    for (int g = 0; g < G; ++g) {
      bh[g] = q.b_hh[g * H + cn];
      if (q.xadd) xa[g] = q.xadd[(long)crow * GH + g * H + cn];
    }
    len = q.lengths ? q.lengths[crow] : (int64_t)T;
  }
  const int xrow = crv ? crow % q.xmod : 0;
  __syncthreads();

  for (int i = 0; i < T; ++i) {
    const int t = q.reverse ? T - 1 - i : i;
    const int in_slot = q.reverse ? t + 1 : t, out_slot = q.reverse ? t : t + 1;
    // x-side pre-activations of this step: independent of the recurrence, fetched before the wait
    float x[G];
#pragma unroll
    for (int g = 0; g < G; ++g) x[g] = 0.f;
    if (crv) {
      const float* xp = q.xproj + ((long)t * q.xmod + xrow) * GH + cn;
#pragma unroll
      for (int g = 0; g < G; ++g) x[g] = xp[g * H];          // consumed after the product: no wait here
    }
    P_STAMP(0);
    if (i > 0 && !wait_flags(flg, nb, (unsigned)i, p.status, &s_abort)) return;
    P_STAMP(1);

    // h_{t-1}: A operand, two 16-row fragments, this wave's K range
    f32x4v acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    if (wave < NW) {
      f32x4v a[2][4];
#pragma unroll
      for (int f = 0; f < 2; ++f) {
        const int row = 16 * f + l15;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          a[f][j] = f32x4v{0.f, 0.f, 0.f, 0.f};
          if (4 * j < NSL && row < R) a[f][j] = ld16_sc1(rs_hs, in_slot * RH + (long)row * H + kbase + 4 * j);
        }
      }
      if constexpr (BF) {
        bf16x8p ah[2], al[2];
#pragma unroll
        for (int f = 0; f < 2; ++f) {
          const f32x8p v = {a[f][0][0], a[f][0][1], a[f][0][2], a[f][0][3], a[f][1][0], a[f][1][1], a[f][1][2], a[f][1][3]};
          ah[f] = __builtin_convertvector(v, bf16x8p);
          al[f] = __builtin_convertvector(v - __builtin_convertvector(ah[f], f32x8p), bf16x8p);
        }
#pragma unroll
        for (int f = 0; f < 2; ++f) acc[f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[f], wbh, acc[f], 0, 0, 0);
#pragma unroll
        for (int f = 0; f < 2; ++f) acc[f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[f], wbl, acc[f], 0, 0, 0);
#pragma unroll
        for (int f = 0; f < 2; ++f) acc[f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[f], wbh, acc[f], 0, 0, 0);
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (4 * j < NSL) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0][j][e], w[4 * j + e], acc[0], 0, 0, 0);
              acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[1][j][e], w[4 * j + e], acc[1], 0, 0, 0);
            }
          }
      }
    }
    P_STAMP(2);
    // C layout of 16x16x4: acc[f][r] = (row 16f + 4*kq + r, column l15)
#pragma unroll
    for (int f = 0; f < 2; ++f)
#pragma unroll
      for (int r = 0; r < 4; ++r) red[wave][16 * f + 4 * kq + r][l15] = acc[f][r];
    __syncthreads();
    P_STAMP(3);

    float sv_c = 0.f, sv_g[4] = {0.f, 0.f, 0.f, 0.f};      // this step's c and gates, stored after the flag
    if (cellt) {
      float s[G];
#pragma unroll
      for (int g = 0; g < G; ++g) s[g] = 0.f;
      for (int wv = 0; wv < NW; ++wv) {
#pragma unroll
        for (int g = 0; g < G; ++g) s[g] += red[wv][crow][cu * G + g];
      }
      float hnew = hprev, cnew = cprev, g0, g1, g2, g3;
      const bool live = (int64_t)t < len;
#pragma unroll
      for (int g = 0; g < G; ++g) { s[g] += bh[g]; x[g] += xa[g]; }
      if (crv) {
        if constexpr (CELL == CELL_GRU) {
          const float rg = fast_sigmoid(x[0] + s[0]);
          const float zg = fast_sigmoid(x[1] + s[1]);
          const float nn = fast_tanh(x[2] + rg * s[2]);
          if (live) hnew = (1.f - zg) * nn + zg * hprev;
          g0 = rg; g1 = zg; g2 = nn; g3 = s[2];
        } else {
          const float ig = fast_sigmoid(x[0] + s[0]);
          const float fg = fast_sigmoid(x[1] + s[1]);
          const float gg = fast_tanh(x[2] + s[2]);
          const float og = fast_sigmoid(x[3] + s[3]);
          const float cc = fg * cprev + ig * gg;
          if (live) { hnew = og * fast_tanh(cc); cnew = cc; }
          g0 = ig; g1 = fg; g2 = gg; g3 = og;
        }
      } else {
        g0 = g1 = g2 = g3 = 0.f;
      }
      hprev = hnew; cprev = cnew;
      // gather the 4 units of a row into the lane of unit 0: 16-byte stores.  Only h is the hand-off
      // payload: it goes out first and alone; c and the gates (saved for backward) are packed and
      // stored AFTER the flag is raised, off the step's critical path (the next publish drains them).
      f32x4v hv;
      hv[0] = hnew;
#pragma unroll
      for (int d = 1; d < 4; ++d) hv[d] = __shfl_down(hnew, d, 64);
      if (crv && cu == 0) st16_sc1(rs_hs, out_slot * RH + (long)crow * H + u0, hv);
      sv_c = cnew; sv_g[0] = g0; sv_g[1] = g1; sv_g[2] = g2; sv_g[3] = g3;
    }
    P_STAMP(4);
    publish(flg, (unsigned)(i + 1));
    P_STAMP(5);
    if (cellt) {
      f32x4v cv, gv[4];
      cv[0] = sv_c;
#pragma unroll
      for (int g = 0; g < 4; ++g) gv[g][0] = sv_g[g];
#pragma unroll
      for (int d = 1; d < 4; ++d) {
        cv[d] = __shfl_down(sv_c, d, 64);
#pragma unroll
        for (int g = 0; g < 4; ++g) gv[g][d] = __shfl_down(sv_g[g], d, 64);
      }
      if (crv && cu == 0) {
        const long o = (long)crow * H + u0;
        if (CELL == CELL_LSTM) *reinterpret_cast<f32x4v*>(q.cs + out_slot * RH + o) = cv;
        if (q.gates) {
          float* gp = q.gates + ((long)t * R + crow) * 4 * H + u0;
#pragma unroll
          for (int g = 0; g < 4; ++g) *reinterpret_cast<f32x4v*>(gp + g * H) = gv[g];
        }
      }
    }
  }
  if (stamp_on)
    for (int k = 0; k < 10; ++k) p.stamps[k] += sacc_[k];
}

// ------------------------------------------------------------------------- forward, in-band hand-off
// The same step as rnn_fwd_persist_kernel<CELL, true> (bf16x3 product) with ONE memory round trip per step instead
// of two.  The flag form serialises  payload stores -> drain -> flag store -> [flag poll] -> barrier -> payload load;
// here the payload is its own flag: before the launch every slot of hs the kernel is going to write is filled with
// the word P_SENT (0xFFFFFFFF, a NaN pattern arithmetic never produces: AMD's default NaN is 0x7FC00000), the cell
// threads store h with 16-byte write-through (sc1) stores and go on, and every consuming wave polls the very 16-byte
// pieces it multiplies (volatile sc1 buffer loads, so the loop re-reads memory) until none of their words is the
// sentinel.  A 16-byte piece is written by ONE lane's one store; a piece observed half-written would show a
// sentinel word and simply be read again.  Only pieces still missing are re-read.  No flag words, no drain, no
// publish barrier; one workgroup barrier per step (the partial sums go through a double-buffered LDS tile).
// Spins are bounded as before (status word, grid drains).
// Column c of the block's 16 MFMA columns is (unit c >> 2, gate c & 3) (a GRU leaves gate 3 empty), so a cell thread
// reads its unit's gates of one partial tile with ONE ds_read_b128.
constexpr unsigned P_SENT = 0xFFFFFFFFu;
constexpr int P_AUX_SC1_VOLATILE = (int)(0x80000000u | (unsigned)P_AUX_SC1);   // bit 31: volatile (compiler side)

__device__ __forceinline__ u32x4v ld16_sc1_poll(__amdgpu_buffer_rsrc_t r, long float_off) {
  return __builtin_amdgcn_raw_buffer_load_b128(r, (int)(float_off * 4), 0, P_AUX_SC1_VOLATILE);
}
__device__ __forceinline__ bool piece_landed(u32x4v v) {
  return v[0] != P_SENT && v[1] != P_SENT && v[2] != P_SENT && v[3] != P_SENT;
}

// Work split (HU, RG): a sequence's R <= 32 rows are cut into RG groups of RPG = 32 / RG rows, its H units into
// H / HU slices; a block owns (row group, unit slice) and exchanges h only with the H / HU blocks of its row group
// (rows never interact).  HU = 4, RG = 1 is the split of the flag kernel: every block pulls the whole 32 x H state
// (64 KB at H = 512) through its CU's memory pipe each step and every line is read by H / 4 CUs.  HU = 8 / RG = 2
// and HU = 16 / RG = 4 keep the block count and halve / quarter both (a 16-row MFMA fragment is the minimum, so
// RG = 4 multiplies 8 live rows in a 16-row fragment).
// value of lane + D of the same 4-lane quad (D = 1..3; wraps inside the quad): one DPP move, no LDS round trip
template <int D>
__device__ __forceinline__ float quad_down(float v) {
  constexpr int ctrl = ((0 + D) & 3) | (((1 + D) & 3) << 2) | (((2 + D) & 3) << 4) | (((3 + D) & 3) << 6);   // quad_perm
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, 0xf, 0xf, true));
}

template <int CELL, int HU, int RG>
__global__ __launch_bounds__(P_THREADS) void rnn_fwd_persist_tag_kernel(PFwdArgs p) {
  constexpr int G = CELL == CELL_GRU ? 3 : 4;
  constexpr int RPG = 32 / RG;                         // rows of a group (32, 16, 8)
  constexpr int NF = RPG >= 16 ? RPG / 16 : 1;         // 16-row MFMA fragments
  constexpr int CF = HU / 4;                           // 16-column fragments: (4 units x 4 gates) each
  constexpr int NP = 2 * NF;                           // 16-byte pieces of h a lane multiplies
  constexpr int LDR = 16 * CF;                         // floats per row of a partial tile
  constexpr int TILE = RPG * LDR;                      // one wave's partial tile (512 floats for every (HU, RG))
  extern __shared__ __attribute__((aligned(16))) float redd[];       // [2][16 waves][RPG rows][LDR]
  __shared__ int s_abort;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, kq = lane >> 4;
  const PSeqF& q = p.q[blockIdx.y];
  const int H = p.H, R = p.R, T = p.T;
  const int nslice = H / HU;
  const int u0 = (blockIdx.x % nslice) * HU;
  const int row0 = (blockIdx.x / nslice) * RPG;
  const int NW = H / 32;                               // one 32-wide bf16 MFMA K block per wave
  const long RH = (long)R * H;
  const int GH = G * H;
  const bool stamp_on = p.stamps && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0;
  unsigned long long tprev_ = stamp_on ? __builtin_amdgcn_s_memtime() : 0ull;
  unsigned sacc_[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  const __amdgpu_buffer_rsrc_t rs_hs = make_rsrc(q.hs);
  if (tid == 0) s_abort = 0;
  if (row0 >= R) return;                               // a row group past the batch: nothing to do, nobody waits for it

  // W_hh fragments (B operand: lane = column l15 = (unit & 3) * 4 + gate of fragment c, k = wave*32 + 8*kq .. + 7),
  // split once, resident for all T steps
  bf16x8p wbh[CF], wbl[CF];
  const int kbase = wave * 32 + kq * 8;
#pragma unroll
  for (int c = 0; c < CF; ++c) {
#pragma unroll
    for (int e = 0; e < 8; ++e) { wbh[c][e] = (__bf16)0.f; wbl[c][e] = (__bf16)0.f; }
    if (wave < NW && (l15 & 3) < G) {
      const float* wr = q.W_hh + (long)((l15 & 3) * H + u0 + 4 * c + (l15 >> 2)) * H + kbase;
      const float4 v0 = *reinterpret_cast<const float4*>(wr), v1 = *reinterpret_cast<const float4*>(wr + 4);
      const f32x8p v = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
      wbh[c] = __builtin_convertvector(v, bf16x8p);
      wbl[c] = __builtin_convertvector(v - __builtin_convertvector(wbh[c], f32x8p), bf16x8p);
    }
  }

  // cell threads: (row, unit) = (tid / HU, tid % HU) of the block's RPG x HU = 128 pairs; state in registers
  const bool cellt = tid < 128;
  const int clr = tid / HU, cu = tid % HU;             // row within the group, unit within the slice
  const int crow = row0 + clr, cn = u0 + cu;
  const bool crv = cellt && crow < R;
  float hprev = 0.f, cprev = 0.f, bh[G], xa[G];
  int64_t len = 0;
#pragma unroll
  for (int g = 0; g < G; ++g) { bh[g] = 0.f; xa[g] = 0.f; }
  const int init_slot = q.reverse ? T : 0;
  if (crv) {
    hprev = q.hs[init_slot * RH + (long)crow * H + cn];
    if (CELL == CELL_LSTM) cprev = q.cs[init_slot * RH + (long)crow * H + cn];
#pragma unroll
    for (int g = 0; g < G; ++g) {
      bh[g] = q.b_hh[g * H + cn];
      if (q.xadd) xa[g] = q.xadd[(long)crow * GH + g * H + cn];
    }
    len = q.lengths ? q.lengths[crow] : (int64_t)T;
  }
  const int xrow = crv ? crow % q.xmod : 0;
  // the pieces of h_{t-1} this lane multiplies: piece 2f + j = row row0 + 16f + l15, k = kbase + 4j .. + 3
  unsigned mine = 0;
  if (wave < NW) {
#pragma unroll
    for (int f = 0; f < NF; ++f)
      if (16 * f + l15 < RPG && row0 + 16 * f + l15 < R) mine |= 3u << (2 * f);
  }
  const long poff = (long)(row0 + l15) * H + kbase;
  __syncthreads();

  // c and the gates of a step (saved for backward: nobody in this launch reads them) leave AFTER the next step's
  // poll loads have been issued: on the cell waves they would otherwise sit in the memory pipe in front of them
  f32x4v pend_c = {0.f, 0.f, 0.f, 0.f}, pend_g[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) pend_g[g] = f32x4v{0.f, 0.f, 0.f, 0.f};
  long pend_co = -1, pend_go = 0;
  auto flush_pending = [&]() {
    if (pend_co >= 0) {
      if (CELL == CELL_LSTM) *reinterpret_cast<f32x4v*>(q.cs + pend_co) = pend_c;
      if (q.gates) {
#pragma unroll
        for (int g = 0; g < 4; ++g) *reinterpret_cast<f32x4v*>(q.gates + pend_go + g * H) = pend_g[g];
      }
    }
  };

  for (int i = 0; i < T; ++i) {
    const int t = q.reverse ? T - 1 - i : i;
    const int in_slot = q.reverse ? t + 1 : t, out_slot = q.reverse ? t : t + 1;
    float* red = redd + (i & 1) * (16 * TILE);
    P_STAMP(0);

    u32x4v a[NP];
#pragma unroll
    for (int j = 0; j < NP; ++j) a[j] = u32x4v{0u, 0u, 0u, 0u};
    unsigned need = mine;
    const long base = in_slot * RH + poff;
    if (wave < NW) {
      // h_{t-1} cannot be there before the producers' cell phase and their stores' flight: polling earlier only
      // queues loads in front of this CU's own h stores (measured: the step got 25 % longer)
      if (i > 0)
        for (int k = (wave < 2 ? p.poll_delay_cell : p.poll_delay); k > 0; --k) __builtin_amdgcn_s_sleep(1);
#pragma unroll
      for (int j = 0; j < NP; ++j)
        if (need & (1u << j)) a[j] = ld16_sc1_poll(rs_hs, base + (long)(j >> 1) * 16 * H + 4 * (j & 1));
    }
    // behind the poll loads in the memory pipe: the previous step's c / gates, and this step's x-side
    // pre-activations (independent of the recurrence, consumed after the barrier)
    float x[G];
#pragma unroll
    for (int g = 0; g < G; ++g) x[g] = 0.f;
    if (cellt) {
      flush_pending();
      pend_co = -1;
      if (crv) {
        const float* xp = q.xproj + ((long)t * q.xmod + xrow) * GH + cn;
#pragma unroll
        for (int g = 0; g < G; ++g) x[g] = xp[g * H];
      }
    }
    if (wave < NW) {
      unsigned spins = 0;
      for (;;) {
#pragma unroll
        for (int j = 0; j < NP; ++j)
          if ((need & (1u << j)) && piece_landed(a[j])) need &= ~(1u << j);
        if (!__any(need != 0u)) break;
        __builtin_amdgcn_s_sleep(2);
        if (++spins > P_SPIN_LIMIT) {
          if (lane == 0) {
            __hip_atomic_store(p.status, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);   // host-mapped word
            s_abort = 1;
          }
          break;
        }
#pragma unroll
        for (int j = 0; j < NP; ++j)
          if (need & (1u << j)) a[j] = ld16_sc1_poll(rs_hs, base + (long)(j >> 1) * 16 * H + 4 * (j & 1));
      }
      P_STAMP(1);
#pragma unroll
      for (int f = 0; f < NF; ++f) {
        const f32x4v lo4 = __builtin_bit_cast(f32x4v, a[2 * f]), hi4 = __builtin_bit_cast(f32x4v, a[2 * f + 1]);
        const f32x8p v = {lo4[0], lo4[1], lo4[2], lo4[3], hi4[0], hi4[1], hi4[2], hi4[3]};
        const bf16x8p ah = __builtin_convertvector(v, bf16x8p);
        const bf16x8p al = __builtin_convertvector(v - __builtin_convertvector(ah, f32x8p), bf16x8p);
#pragma unroll
        for (int c = 0; c < CF; ++c) {
          f32x4v acc = {0.f, 0.f, 0.f, 0.f};
          acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, wbh[c], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, wbl[c], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, wbh[c], acc, 0, 0, 0);
          // C layout of 16x16x32: acc[r] = (row 16f + 4*kq + r, column l15)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (16 * f + 4 * kq + r < RPG) red[wave * TILE + (16 * f + 4 * kq + r) * LDR + 16 * c + l15] = acc[r];
        }
      }
    }
    P_STAMP(2);
    __syncthreads();
    if (s_abort) return;                                   // uniform: written before the barrier
    P_STAMP(3);

    if (cellt) {
      f32x4v s4 = {0.f, 0.f, 0.f, 0.f};
      const float* rp = red + clr * LDR + (cu >> 2) * 16 + (cu & 3) * 4;
      for (int wv = 0; wv < NW; ++wv) s4 += *reinterpret_cast<const f32x4v*>(rp + wv * TILE);
      float s[4] = {s4[0], s4[1], s4[2], s4[3]};
      P_STAMP(6);
      float hnew = hprev, cnew = cprev, g0, g1, g2, g3;
      const bool live = (int64_t)t < len;
#pragma unroll
      for (int g = 0; g < G; ++g) { s[g] += bh[g]; x[g] += xa[g]; }
      if (crv) {
        if constexpr (CELL == CELL_GRU) {
          const float rg = fast_sigmoid(x[0] + s[0]);
          const float zg = fast_sigmoid(x[1] + s[1]);
          const float nn = fast_tanh(x[2] + rg * s[2]);
          if (live) hnew = (1.f - zg) * nn + zg * hprev;
          g0 = rg; g1 = zg; g2 = nn; g3 = s[2];
        } else {
          const float ig = fast_sigmoid(x[0] + s[0]);
          const float fg = fast_sigmoid(x[1] + s[1]);
          const float gg = fast_tanh(x[2] + s[2]);
          const float og = fast_sigmoid(x[3] + s[3]);
          const float cc = fg * cprev + ig * gg;
          if (live) { hnew = og * fast_tanh(cc); cnew = cc; }
          g0 = ig; g1 = fg; g2 = gg; g3 = og;
        }
      } else {
        g0 = g1 = g2 = g3 = 0.f;
      }
      hprev = hnew; cprev = cnew;
      P_STAMP(7);
      // the hand-off payload first and alone: 4 units of a row in the lane of the first of them (quad-local
      // data-parallel moves, no LDS round trip), one 16-byte store
      f32x4v hv;
      hv[0] = hnew; hv[1] = quad_down<1>(hnew); hv[2] = quad_down<2>(hnew); hv[3] = quad_down<3>(hnew);
      if (crv && (cu & 3) == 0) st16_sc1(rs_hs, out_slot * RH + (long)crow * H + cn, hv);
      P_STAMP(4);
      pend_c[0] = cnew; pend_c[1] = quad_down<1>(cnew); pend_c[2] = quad_down<2>(cnew); pend_c[3] = quad_down<3>(cnew);
      const float sg[4] = {g0, g1, g2, g3};
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        pend_g[g][0] = sg[g]; pend_g[g][1] = quad_down<1>(sg[g]); pend_g[g][2] = quad_down<2>(sg[g]);
        pend_g[g][3] = quad_down<3>(sg[g]);
      }
      if (crv && (cu & 3) == 0) {
        pend_co = out_slot * RH + (long)crow * H + cn;
        pend_go = ((long)t * R + crow) * 4 * H + cn;
      }
    }
    P_STAMP(5);
  }
  if (cellt) flush_pending();
  if (stamp_on)
    for (int k = 0; k < 10; ++k) p.stamps[k] += sacc_[k];
}

// -------------------------------------------------------------------------------------- backward
// Step t of the backward recurrence, for the block that owns hidden units U (4 units, 4G gate rows):
//   1. dh_t[:, U] = carry + dhs_t[:, U] + sum over ALL blocks b of part_b[:, U]   (read the exchange)
//   2. cell backward for (row, unit in U)  ->  dG_t[:, gate rows of U]  (global, for the weight
//      gradients; and LDS, as the A operand of the next product)
//   3. part[32 x H] = dG_t[:, rows of U] (32 x 4G) . W_hh[rows of U, :] (4G x H): the block's K-slice
//      of dh_{t-1} for EVERY unit.  v_mfma_f32_16x16x4_f32, W slice in registers, no padding waste
//      (owning output columns instead would leave a 32 x 4 tile per block: 4x MFMA padding, and an
//      all-gather of the 32 x G*H gradient, 256 KB per block per step - measured 12 us per step).
//   4. publish part, destination-major: ex[dest block][this block][32 rows][4 units] (sc1 16-byte
//      stores), drain, flag.  The destination reads one contiguous nb x 512 B run.
// Two exchange buffers (step parity).  Deterministic: partials are summed in block order.
// HU = hidden units per block (4 or 8).  A block's traffic per step does not depend on HU (it publishes
// its 32 x H partial and gathers 32 x H of others' partials either way), so the exchange volume of a
// sequence is (H / HU) x 2 x 32 x H x 4 B per step: HU = 8 halves it (and the number of flags to poll) at
// twice the -- small -- MFMA and cell work per block.
// BF: the product on the bf16 matrix pipe with a three-term split of both operands ("bf16x6", as gemm_wide.hip:
// fp32-level accuracy).  The block's K is 4G..8G <= 32 = ONE v_mfma_f32_16x16x32_bf16 K block, so a (row fragment,
// column fragment) pair costs 6 x 16 cycles instead of 8 x 32 for v_mfma_f32_16x16x4_f32: the product phase was
// 1.7 us of the 6.1 us step.  dG is split ONCE, by the cell threads as they write it to LDS (all 16 waves read the
// same A fragments); W_hh's slice is split into register fragments before the loop.

template <int CELL, int HU, bool BF>
__global__ __launch_bounds__(P_THREADS) void rnn_bwd_persist_kernel(PBwdArgs p) {
  constexpr int G = CELL == CELL_GRU ? 3 : 4;
  constexpr int NC = HU * G;                         // live K of this block's slice (12 / 16 / 24 / 32)
  constexpr int KS = (NC + 3) / 4;                   // MFMA K steps
  constexpr int LDG = KS * 4 + 1;                    // dGl row stride: odd -> the fragment reads are conflict-free
  constexpr int GPR = HU / 4;                        // 16-byte granules per (row, destination) piece
  constexpr int PIECE = 32 * HU;                     // floats one block sends to one destination per step
  constexpr int NGRP = P_THREADS / (32 * GPR);       // source groups after the gather's first pass
  extern __shared__ __attribute__((aligned(16))) float dyn[];      // tile[32][H + 4]  (aliased: red[1024][4])
  constexpr int NKB = (NC + 31) / 32;                // BF: 32-wide MFMA K blocks (1; 2 at HU = 16)
  constexpr int DGS_LD = 32 * NKB + 8;               // bf16 per dG row in LDS (80 / 144 B): the 16 rows of a b128 fragment read cover all 64 banks
  __shared__ float dGl[BF ? 1 : 32][LDG];                          // dG of this block: [row][u*G + g]
  __shared__ __attribute__((aligned(16))) __bf16 dGs[BF ? 3 : 1][BF ? 32 * DGS_LD : 8];   // BF: hi | mid | lo, [row][k]
  __shared__ int s_abort;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, kq = lane >> 4;
  const PSeqB& q = p.q[blockIdx.y];
  const int H = p.H, R = p.R, T = p.T, nb = gridDim.x;
  const int GH = G * H, LDT = H + 4;
  const int u0 = blockIdx.x * HU;
  const long RH = (long)R * H;
  float* tile = dyn;
  float* red = dyn;
  unsigned* flg = p.flags + blockIdx.y * P_MAXB;
  const bool stamp_on = p.stamps && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0;
  unsigned long long tprev_ = stamp_on ? __builtin_amdgcn_s_memtime() : 0ull;
  unsigned sacc_[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  float* dgh_base = q.dgh ? q.dgh : q.dgx;
  // exchange: [parity][dest][src][32][HU] floats
  const long ex_par = (long)nb * nb * PIECE;
  const __amdgpu_buffer_rsrc_t rs_ex = make_rsrc(p.exch + (size_t)blockIdx.y * 2 * ex_par);
  if (tid == 0) s_abort = 0;
  if constexpr (BF) {
    for (int i = tid; i < 3 * 32 * DGS_LD; i += P_THREADS) (&dGs[0][0])[i] = (__bf16)0.f;
  } else {
    for (int i = tid; i < 32 * LDG; i += P_THREADS) (&dGl[0][0])[i] = 0.f;
  }

  // W_hh[gate rows of U, :] as B fragments: wave w owns output column fragments w, w + 16, ...
  // (H / 16 fragments); fragment nf, MFMA step s: lane (n = l15, kq) holds W_hh[row(4s + kq)][16*nf + l15]
  const int nfr = H / 16;
  float w[2][BF ? 1 : KS];
  bf16x8p wb[2][NKB][3];                                 // BF: lane (n = l15, kq) holds k = 32*kb + 8*kq .. + 7 (zero past NC)
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int nf = wave + 16 * j;
    if constexpr (BF) {
#pragma unroll
      for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int k = 32 * kb + 8 * kq + e;
          float v = 0.f;
          if (nf < nfr && k < NC) v = q.W_hh[(long)((k % G) * H + u0 + k / G) * H + 16 * nf + l15];
          __bf16 h_, m_, l_;
          split3(v, h_, m_, l_);
          wb[j][kb][0][e] = h_; wb[j][kb][1][e] = m_; wb[j][kb][2][e] = l_;
        }
    } else {
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        const int k = 4 * s + kq;
        w[j][s] = 0.f;
        if (nf < nfr && k < NC) w[j][s] = q.W_hh[(long)((k % G) * H + u0 + k / G) * H + 16 * nf + l15];
      }
    }
  }

  const bool cellt = tid < 32 * HU;
  const int crow = tid / HU, cu = tid % HU, cn = u0 + cu;
  const bool crv = cellt && crow < R;
  float dh_carry = 0.f, dc = 0.f;
  int64_t len = 0;
  if (crv) {
    if (q.dh_last) dh_carry = q.dh_last[(long)crow * H + cn];
    if (CELL == CELL_LSTM && q.dc_last) dc = q.dc_last[(long)crow * H + cn];
    len = q.lengths ? q.lengths[crow] : (int64_t)T;
  }
  // Rows whose gradient is exactly zero at step t (a padded step of a packed sequence; a decoder row
  // past its last target) produce an all-zero partial row: such rows are neither published nor
  // gathered.  bound(t) = 1 + the highest row that is still non-zero at step t; with the batch's rows
  // ordered by length that is the live-row count, and the exchange traffic follows it.
  const int64_t za = (lane < R) ? (q.zero_after ? q.zero_after[lane] : (q.lengths ? q.lengths[lane] : (int64_t)T)) : 0;
  auto row_bound = [&](int t) -> int {
    const unsigned long long m = __ballot(lane < 32 && (int64_t)t < za);
    return m ? 64 - __clzll(m) : 0;
  };
  int rb_prev = 0;                                      // bound of the partials published last
  __syncthreads();

  // 16-byte granule gq of a block's 32 x H floats: (block d = gq / (32*GPR), row = (gq / GPR) % 32,
  // part = gq % GPR); a thread's granules gq = tid + 1024 j share row and part (1024 % (32*GPR) == 0)
  const int grow = (tid / GPR) & 31, gpart = tid % GPR;
  const int ngran = 8 * H;                              // = nb * 32 * GPR

  // sum over all source blocks of their partial for (crow, cu); parity = buffer to read
  auto gather = [&](int parity, int bound) -> float {
    const long base = (long)parity * ex_par + (long)blockIdx.x * nb * PIECE;
    f32x4v sum = f32x4v{0.f, 0.f, 0.f, 0.f};
    f32x4v v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      v[j] = f32x4v{0.f, 0.f, 0.f, 0.f};
      const int gq = tid + P_THREADS * j;
      if (gq < ngran && grow < bound) v[j] = ld16_sc1(rs_ex, base + (long)gq * 4);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) sum += v[j];
    // red[group = tid / (32*GPR)][row][part][4]
    *reinterpret_cast<f32x4v*>(red + (size_t)tid * 4) = sum;
    __syncthreads();
    float tot = 0.f;
    if (cellt) {
#pragma unroll 8
      for (int g = 0; g < NGRP; ++g) tot += red[((size_t)(g * 32 + crow) * GPR + (cu >> 2)) * 4 + (cu & 3)];
    }
    __syncthreads();                                  // red is aliased with the product tile
    return tot;
  };

  for (int i = T - 1; i >= 0; --i) {
    const int t = q.reverse ? T - 1 - i : i;
    const int in_slot = q.reverse ? t + 1 : t, out_slot = q.reverse ? t : t + 1;
    // saved tensors of step t for this (row, unit): independent of the recurrence
    float gl[4] = {0.f, 0.f, 0.f, 0.f}, hin = 0.f, cin = 0.f, cout = 0.f, dhs_t = 0.f;
    if (crv) {
      const float* gp = q.gates + ((long)t * R + crow) * 4 * H + cn;
#pragma unroll
      for (int g = 0; g < 4; ++g) gl[g] = gp[g * H];
      hin = q.hs[in_slot * RH + (long)crow * H + cn];
      if (CELL == CELL_LSTM) {
        cin = q.cs[in_slot * RH + (long)crow * H + cn];
        cout = q.cs[out_slot * RH + (long)crow * H + cn];
      }
      if (q.dhs) dhs_t = q.dhs[(long)t * RH + (long)crow * H + cn];
    }
    float part = 0.f;
    P_STAMP(0);
    if (i < T - 1) {
      if (!wait_flags(flg, nb, (unsigned)(T - 1 - i), p.status, &s_abort)) return;
      P_STAMP(1);
      part = gather((T - 1 - i) & 1, rb_prev);
      P_STAMP(2);
    }
    if (cellt) {
      float ox[4] = {0.f, 0.f, 0.f, 0.f}, oh[4] = {0.f, 0.f, 0.f, 0.f};
      if (crv) {
        const bool live = (int64_t)t < len;
        const float dh = dh_carry + dhs_t + part;
        dh_carry = cell_bwd_point<CELL>(live, dh, dc, gl, 1, hin, cin, cout, ox, CELL == CELL_GRU ? oh : nullptr);
      }
      // A operand of the product: the h-side gradients
#pragma unroll
      for (int g = 0; g < G; ++g) {
        const float v = (CELL == CELL_GRU) ? oh[g] : ox[g];
        if constexpr (BF) {
          __bf16 h_, m_, l_;
          split3(v, h_, m_, l_);
          const int o = crow * DGS_LD + cu * G + g;
          dGs[0][o] = h_; dGs[1][o] = m_; dGs[2][o] = l_;
        } else {
          dGl[crow][cu * G + g] = v;
        }
      }
      // global copies for the weight gradients: 16-byte stores per (row, gate, 4 units)
#pragma unroll
      for (int g = 0; g < G; ++g) {
        f32x4v vx, vh;
        vx[0] = ox[g]; vh[0] = oh[g];
#pragma unroll
        for (int d = 1; d < 4; ++d) {
          vx[d] = __shfl_down(ox[g], d, 64);
          if (CELL == CELL_GRU) vh[d] = __shfl_down(oh[g], d, 64);
        }
        if (crv && (cu & 3) == 0) {
          const long o = ((long)t * R + crow) * GH + g * H + cn;
          *reinterpret_cast<f32x4v*>(q.dgx + o) = vx;
          if (CELL == CELL_GRU) *reinterpret_cast<f32x4v*>(dgh_base + o) = vh;
          else if (q.dgx != dgh_base) *reinterpret_cast<f32x4v*>(dgh_base + o) = vx;
        }
      }
    }
    __syncthreads();
    P_STAMP(3);
    // product: two 16-row fragments x this wave's column fragments, K = NC (KS steps).  Rows at or past
    // this step's bound have an all-zero dG: a 16-row fragment wholly past it is neither multiplied nor
    // written (its part of the tile is not published either).
    const int rb_cur = row_bound(t);
    if constexpr (BF) {
#pragma unroll
      for (int f = 0; f < 2; ++f) {
        if (rb_cur > 16 * f) {
          bf16x8p a[NKB][3];
#pragma unroll
          for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
            for (int tm = 0; tm < 3; ++tm)
              a[kb][tm] = *reinterpret_cast<const bf16x8p*>(&dGs[tm][(16 * f + l15) * DGS_LD + 32 * kb + 8 * kq]);
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const int nf = wave + 16 * j;
            if (nf < nfr) {
              // small terms first: mid.mid, hi.lo, lo.hi, hi.mid, mid.hi, hi.hi
              f32x4v acc = f32x4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll
              for (int kb = 0; kb < NKB; ++kb) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[kb][1], wb[j][kb][1], acc, 0, 0, 0);
#pragma unroll
              for (int kb = 0; kb < NKB; ++kb) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[kb][0], wb[j][kb][2], acc, 0, 0, 0);
#pragma unroll
              for (int kb = 0; kb < NKB; ++kb) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[kb][2], wb[j][kb][0], acc, 0, 0, 0);
#pragma unroll
              for (int kb = 0; kb < NKB; ++kb) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[kb][0], wb[j][kb][1], acc, 0, 0, 0);
#pragma unroll
              for (int kb = 0; kb < NKB; ++kb) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[kb][1], wb[j][kb][0], acc, 0, 0, 0);
#pragma unroll
              for (int kb = 0; kb < NKB; ++kb) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[kb][0], wb[j][kb][0], acc, 0, 0, 0);
#pragma unroll
              for (int r = 0; r < 4; ++r) tile[(16 * f + 4 * kq + r) * LDT + 16 * nf + l15] = acc[r];
            }
          }
        }
      }
    } else {
      float a[2][KS];
#pragma unroll
      for (int f = 0; f < 2; ++f)
#pragma unroll
        for (int s = 0; s < KS; ++s) a[f][s] = dGl[16 * f + l15][4 * s + kq];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int nf = wave + 16 * j;
        if (nf < nfr) {
          // C layout: acc[r] = (row 16f + 4*kq + r, column 16*nf + l15)
          if (rb_cur > 0) {
            f32x4v acc0 = f32x4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < KS; ++s) acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0][s], w[j][s], acc0, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; ++r) tile[(4 * kq + r) * LDT + 16 * nf + l15] = acc0[r];
          }
          if (rb_cur > 16) {
            f32x4v acc1 = f32x4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < KS; ++s) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[1][s], w[j][s], acc1, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; ++r) tile[(16 + 4 * kq + r) * LDT + 16 * nf + l15] = acc1[r];
          }
        }
      }
    }
    __syncthreads();
    // publish, destination-major: granule (dest d, row r, part) = tile[r][HU*d + 4*part ..] ->
    // ex[parity][d][me][r][4*part ..]
    {
      const long wbase = (long)((T - i) & 1) * ex_par + (long)blockIdx.x * PIECE;
      rb_prev = rb_cur;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int gq = tid + P_THREADS * j;
        if (gq < ngran && grow < rb_prev) {
          const int d = gq / (32 * GPR);
          const f32x4v v = *reinterpret_cast<const f32x4v*>(tile + grow * LDT + HU * d + 4 * gpart);
          st16_sc1(rs_ex, wbase + ((long)d * nb) * PIECE + (long)grow * HU + 4 * gpart, v);
        }
      }
    }
    P_STAMP(4);
    publish(flg, (unsigned)(T - i));
    P_STAMP(5);
  }
  // gradient wrt the initial state: carry + the last product
  float part = 0.f;
  if (T > 0) {
    if (!wait_flags(flg, nb, (unsigned)T, p.status, &s_abort)) return;
    part = gather(T & 1, rb_prev);
  }
  if (crv) {
    q.dh0[(long)crow * H + cn] = dh_carry + part;
    if (CELL == CELL_LSTM) q.dc0[(long)crow * H + cn] = dc;
  }
  if (stamp_on)
    for (int k = 0; k < 10; ++k) p.stamps[k] += sacc_[k];
}

// ------------------------------------------------------------------------ backward, in-band hand-off
// The backward step of rnn_bwd_persist_kernel<CELL, HU, true> (bf16x6 product, reduce-scatter of the partial dh)
// with the forward kernel's two changes:
//   * rows x units work split: a block owns HU units of ONE of RG row groups (RPG = 32 / RG rows; rows never
//     interact), so it publishes an RPG x H partial and gathers RPG x HU x (H / HU) floats per step: 64 KB each way
//     for (HU 8, RG 1), 32 KB for (8, 2), 16 KB for (16, 4), and exchanges with H / HU blocks only;
//   * in-band hand-off: the exchange buffer is armed with P_SENT; a gathering thread polls the 16-byte pieces it
//     sums (volatile sc1 loads) until they hold no sentinel word, and re-arms them (sc1 stores of the sentinel) as
//     soon as it has them.  A piece is rewritten by its producer two steps later at the earliest, behind a chain of
//     hand-offs that starts after this thread's next vmcnt(0) wait, which every wave makes before it publishes
//     again: the re-arm store has completed before that.  No flags, no drain, no publish barrier.
// The pieces a step publishes are known on both sides (row bound from zero_after / lengths), as before.
template <int CELL, int HU, int RG>
__global__ __launch_bounds__(P_THREADS) void rnn_bwd_persist_tag_kernel(PBwdArgs p) {
  constexpr int G = CELL == CELL_GRU ? 3 : 4;
  constexpr int NC = HU * G;                         // live K of this block's slice
  constexpr int NKB = (NC + 31) / 32;                // 32-wide MFMA K blocks
  constexpr int DGS_LD = 32 * NKB + 8;               // bf16 per dG row in LDS
  constexpr int RPG = 32 / RG;                       // rows of a group
  constexpr int NF = RPG >= 16 ? RPG / 16 : 1;       // 16-row MFMA fragments
  constexpr int MR = 16 * NF;                        // rows of the A operand in LDS (padding rows stay zero)
  constexpr int GPR = HU / 4;                        // 16-byte pieces per (row, destination)
  constexpr int PIECE = RPG * HU;                    // floats one block sends to one destination per step
  static_assert(RPG * GPR == 32, "a wave's 64 lanes cover two sources x 32 (row, piece) pairs");
  extern __shared__ __attribute__((aligned(16))) float dyn[];      // tile[RPG][H + 4] | red[16 waves][32][4]
  __shared__ __attribute__((aligned(16))) __bf16 dGs[3][MR * DGS_LD];           // hi | mid | lo, [row][k]
  __shared__ int s_abort;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, kq = lane >> 4;
  const PSeqB& q = p.q[blockIdx.y];
  const int H = p.H, R = p.R, T = p.T;
  const int nsl = H / HU;
  const int sl = blockIdx.x % nsl, rgi = blockIdx.x / nsl;
  const int GH = G * H, LDT = H + 4;
  const int u0 = sl * HU, row0 = rgi * RPG;
  const long RH = (long)R * H;
  float* tile = dyn;
  float* red = dyn + RPG * LDT;
  const bool stamp_on = p.stamps && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0;
  unsigned long long tprev_ = stamp_on ? __builtin_amdgcn_s_memtime() : 0ull;
  unsigned sacc_[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  float* dgh_base = q.dgh ? q.dgh : q.dgx;
  // exchange of this (sequence, row group): [parity][dest slice][src slice][RPG][HU] floats
  const long ex_par = (long)nsl * nsl * PIECE;
  const __amdgpu_buffer_rsrc_t rs_ex = make_rsrc(p.exch + ((size_t)blockIdx.y * RG + rgi) * 2 * ex_par);
  if (tid == 0) s_abort = 0;
  if (row0 >= R) return;                               // a row group past the batch
  for (int i = tid; i < 3 * MR * DGS_LD; i += P_THREADS) (&dGs[0][0])[i] = (__bf16)0.f;

  // W_hh[gate rows of the slice, :] as B fragments: wave w owns output column fragments w, w + 16
  const int nfr = H / 16;
  bf16x8p wb[2][NKB][3];                               // lane (n = l15, kq) holds k = 32*kb + 8*kq .. + 7 (zero past NC)
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int nf = wave + 16 * j;
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int k = 32 * kb + 8 * kq + e;
        float v = 0.f;
        if (nf < nfr && k < NC) v = q.W_hh[(long)((k % G) * H + u0 + k / G) * H + 16 * nf + l15];
        __bf16 h_, m_, l_;
        split3(v, h_, m_, l_);
        wb[j][kb][0][e] = h_; wb[j][kb][1][e] = m_; wb[j][kb][2][e] = l_;
      }
  }

  const bool cellt = tid < RPG * HU;                   // 128 threads
  const int clr = tid / HU, cu = tid % HU;
  const int crow = row0 + clr, cn = u0 + cu;
  const bool crv = cellt && crow < R;
  float dh_carry = 0.f, dc = 0.f;
  int64_t len = 0;
  if (crv) {
    if (q.dh_last) dh_carry = q.dh_last[(long)crow * H + cn];
    if (CELL == CELL_LSTM && q.dc_last) dc = q.dc_last[(long)crow * H + cn];
    len = q.lengths ? q.lengths[crow] : (int64_t)T;
  }
  // bound(t) = 1 + the highest row of the BATCH whose gradient is still non-zero at step t (see the flag kernel)
  const int64_t za = (lane < R) ? (q.zero_after ? q.zero_after[lane] : (q.lengths ? q.lengths[lane] : (int64_t)T)) : 0;
  auto row_bound = [&](int t) -> int {
    const unsigned long long m = __ballot(lane < 32 && (int64_t)t < za);
    return m ? 64 - __clzll(m) : 0;
  };
  int rb_prev = 0;                                      // bound of the partials published last
  __syncthreads();

  // 16-byte piece gq of the block's nsl x RPG x HU incoming (or outgoing) floats: (slice gq / 32, row (gq / GPR) % RPG,
  // part gq % GPR); a thread's pieces gq = tid + 1024 j share row and part
  const int grow = (tid / GPR) % RPG, gpart = tid % GPR;
  const int ngran = nsl * 32;
  constexpr int MAXJ = 4;                               // nsl <= 128

  // sum over all source slices of their partial for (clr, cu); parity = buffer to read.  Uniform false on timeout.
  auto gather = [&](int parity, int bound, float& tot) -> bool {
    const long base = (long)parity * ex_par + (long)sl * nsl * PIECE;
    u32x4v v[MAXJ];
    unsigned need = 0;
#pragma unroll
    for (int j = 0; j < MAXJ; ++j) {
      v[j] = u32x4v{0u, 0u, 0u, 0u};
      if (tid + P_THREADS * j < ngran && row0 + grow < bound) need |= 1u << j;
    }
    const unsigned mine = need;
    unsigned spins = 0;
    for (;;) {
#pragma unroll
      for (int j = 0; j < MAXJ; ++j)
        if (need & (1u << j)) v[j] = ld16_sc1_poll(rs_ex, base + (long)(tid + P_THREADS * j) * 4);
#pragma unroll
      for (int j = 0; j < MAXJ; ++j)
        if ((need & (1u << j)) && piece_landed(v[j])) need &= ~(1u << j);
      if (!__any(need != 0u)) break;
      __builtin_amdgcn_s_sleep(2);
      if (++spins > P_SPIN_LIMIT) {
        if (lane == 0) {
          __hip_atomic_store(p.status, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          s_abort = 1;
        }
        break;
      }
    }
    // re-arm what was taken (its producer writes it again two steps from now)
    const f32x4v sent = __builtin_bit_cast(f32x4v, u32x4v{P_SENT, P_SENT, P_SENT, P_SENT});
    f32x4v sum = f32x4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < MAXJ; ++j)
      if (mine & (1u << j)) {
        st16_sc1(rs_ex, base + (long)(tid + P_THREADS * j) * 4, sent);
        sum += __builtin_bit_cast(f32x4v, v[j]);
      }
    // lanes l and l + 32 hold the same (row, part) of two sources
#pragma unroll
    for (int e = 0; e < 4; ++e) sum[e] += __shfl_xor(sum[e], 32, 64);
    if (lane < 32) *reinterpret_cast<f32x4v*>(red + (size_t)(wave * 32 + lane) * 4) = sum;
    __syncthreads();
    if (s_abort) return false;
    tot = 0.f;
    if (cellt) {
#pragma unroll
      for (int w = 0; w < 16; ++w) tot += red[(size_t)(w * 32 + clr * GPR + (cu >> 2)) * 4 + (cu & 3)];
    }
    return true;
  };

  for (int i = T - 1; i >= 0; --i) {
    const int t = q.reverse ? T - 1 - i : i;
    const int in_slot = q.reverse ? t + 1 : t, out_slot = q.reverse ? t : t + 1;
    // saved tensors of step t for this (row, unit): independent of the recurrence
    float gl[4] = {0.f, 0.f, 0.f, 0.f}, hin = 0.f, cin = 0.f, cout = 0.f, dhs_t = 0.f;
    if (crv) {
      const float* gp = q.gates + ((long)t * R + crow) * 4 * H + cn;
#pragma unroll
      for (int g = 0; g < 4; ++g) gl[g] = gp[g * H];
      hin = q.hs[in_slot * RH + (long)crow * H + cn];
      if (CELL == CELL_LSTM) {
        cin = q.cs[in_slot * RH + (long)crow * H + cn];
        cout = q.cs[out_slot * RH + (long)crow * H + cn];
      }
      if (q.dhs) dhs_t = q.dhs[(long)t * RH + (long)crow * H + cn];
    }
    float part = 0.f;
    P_STAMP(0);
    if (i < T - 1) {
      for (int k = p.poll_delay; k > 0; --k) __builtin_amdgcn_s_sleep(1);
      if (!gather((T - 1 - i) & 1, rb_prev, part)) return;
      P_STAMP(2);
    }
    float ox[4] = {0.f, 0.f, 0.f, 0.f}, oh[4] = {0.f, 0.f, 0.f, 0.f};
    if (cellt) {
      if (crv) {
        const bool live = (int64_t)t < len;
        const float dh = dh_carry + dhs_t + part;
        dh_carry = cell_bwd_point<CELL>(live, dh, dc, gl, 1, hin, cin, cout, ox, CELL == CELL_GRU ? oh : nullptr);
      }
      // A operand of the product: the h-side gradients, split once
#pragma unroll
      for (int g = 0; g < G; ++g) {
        const float v = (CELL == CELL_GRU) ? oh[g] : ox[g];
        __bf16 h_, m_, l_;
        split3(v, h_, m_, l_);
        const int o = clr * DGS_LD + cu * G + g;
        dGs[0][o] = h_; dGs[1][o] = m_; dGs[2][o] = l_;
      }
    }
    __syncthreads();
    P_STAMP(3);
    // product: the group's row fragment(s) x this wave's column fragments, K = NC
    const int rb_cur = row_bound(t);
#pragma unroll
    for (int f = 0; f < NF; ++f) {
      if (rb_cur > row0 + 16 * f) {
        bf16x8p a[NKB][3];
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
          for (int tm = 0; tm < 3; ++tm)
            a[kb][tm] = *reinterpret_cast<const bf16x8p*>(&dGs[tm][(16 * f + l15) * DGS_LD + 32 * kb + 8 * kq]);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int nf = wave + 16 * j;
          if (nf < nfr) {
            // small terms first: mid.mid, hi.lo, lo.hi, hi.mid, mid.hi, hi.hi
            f32x4v acc = f32x4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[kb][1], wb[j][kb][1], acc, 0, 0, 0);
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[kb][0], wb[j][kb][2], acc, 0, 0, 0);
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[kb][2], wb[j][kb][0], acc, 0, 0, 0);
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[kb][0], wb[j][kb][1], acc, 0, 0, 0);
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[kb][1], wb[j][kb][0], acc, 0, 0, 0);
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[kb][0], wb[j][kb][0], acc, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; ++r)
              if (16 * f + 4 * kq + r < RPG) tile[(16 * f + 4 * kq + r) * LDT + 16 * nf + l15] = acc[r];
          }
        }
      }
    }
    // every re-arm store of this step's gather has completed before anything new is published (see the header)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // publish, destination-major: piece (dest d, row r, part) = tile[r][HU*d + 4*part ..] -> ex[parity][d][me][r][4*part ..]
    {
      const long wbase = (long)((T - i) & 1) * ex_par + (long)sl * PIECE;
      rb_prev = rb_cur;
#pragma unroll
      for (int j = 0; j < MAXJ; ++j) {
        const int gq = tid + P_THREADS * j;
        if (gq < ngran && row0 + grow < rb_prev) {
          const int d = gq / 32;
          const f32x4v v = *reinterpret_cast<const f32x4v*>(tile + grow * LDT + HU * d + 4 * gpart);
          st16_sc1(rs_ex, wbase + ((long)d * nsl) * PIECE + (long)grow * HU + 4 * gpart, v);
        }
      }
    }
    P_STAMP(4);
    // global copies of dG for the weight gradients (nobody in this launch reads them): behind the hand-off stores
    if (cellt) {
#pragma unroll
      for (int g = 0; g < G; ++g) {
        f32x4v vx, vh;
        vx[0] = ox[g]; vx[1] = quad_down<1>(ox[g]); vx[2] = quad_down<2>(ox[g]); vx[3] = quad_down<3>(ox[g]);
        if (CELL == CELL_GRU) { vh[0] = oh[g]; vh[1] = quad_down<1>(oh[g]); vh[2] = quad_down<2>(oh[g]); vh[3] = quad_down<3>(oh[g]); }
        if (crv && (cu & 3) == 0) {
          const long o = ((long)t * R + crow) * GH + g * H + cn;
          *reinterpret_cast<f32x4v*>(q.dgx + o) = vx;
          if (CELL == CELL_GRU) *reinterpret_cast<f32x4v*>(dgh_base + o) = vh;
          else if (q.dgx != dgh_base) *reinterpret_cast<f32x4v*>(dgh_base + o) = vx;
        }
      }
    }
    P_STAMP(5);
  }
  // gradient wrt the initial state: carry + the last product
  float part = 0.f;
  if (T > 0) {
    if (!gather(T & 1, rb_prev, part)) return;
  }
  if (crv) {
    q.dh0[(long)crow * H + cn] = dh_carry + part;
    if (CELL == CELL_LSTM) q.dc0[(long)crow * H + cn] = dc;
  }
  if (stamp_on)
    for (int k = 0; k < 10; ++k) p.stamps[k] += sacc_[k];
}

// ------------------------------------------------------------------------------------ host side
// library-owned sync words, one set per stream (lib_scratch): [SKINNY_MAX_BATCH][P_MAXB] epoch flags |
// 16 spare words | 16 stamp accumulators | int64 [SKINNY_MAX_BATCH][32] zero_after scratch
constexpr int P_FLAG_WORDS = SKINNY_MAX_BATCH * P_MAXB;
constexpr size_t P_ZA_OFFSET = (size_t)(P_FLAG_WORDS + 16) * 4 + 16 * 8;      // int64 [SKINNY_MAX_BATCH][32] after the stamps
static int g_stamps_on = -1;
static unsigned long long* g_stamps_last = nullptr;      // stamp words of the stream that launched last
int* persist_status_host();                              // api.hip
size_t persist_sync_bytes() { return P_ZA_OFFSET + (size_t)SKINNY_MAX_BATCH * 32 * 8; }
int* persist_status_copy_word(void* sync_buffer) { return reinterpret_cast<int*>(static_cast<unsigned*>(sync_buffer) + P_FLAG_WORDS); }
static int persist_sync(hipStream_t s, unsigned** out, int** status) {
  void* p = nullptr;
  CVAE_TRY(lib_scratch(LS_PERSIST_SYNC, s, persist_sync_bytes(), &p, 1));
  if (g_stamps_on < 0) {
    const char* e = getenv("CVAE_PERSIST_STAMPS");
    g_stamps_on = (e && e[0] == '1') ? 1 : 0;
  }
  *out = static_cast<unsigned*>(p);
  g_stamps_last = g_stamps_on == 1 ? reinterpret_cast<unsigned long long*>(*out + P_FLAG_WORDS + 16) : nullptr;
  return persist_status_word(status);
}
static unsigned long long* persist_stamps() { return g_stamps_last; }

static int g_persist_on = -1;        // -1: not set yet (env CVAE_PERSISTENT_RNN, default on); cvae_persistent_rnn() sets it
static int persist_enabled() {
  if (g_persist_on < 0) {
    const char* e = getenv("CVAE_PERSISTENT_RNN");
    g_persist_on = (e && e[0] == '0') ? 0 : 1;
  }
  return g_persist_on;
}
int persist_rnn_set(int on) {
  const int prev = persist_enabled();
  g_persist_on = on ? 1 : 0;
  return prev;
}
// CUs the whole-sequence kernels leave free (a data-parallel job keeps RCCL's kernels beside them)
static int g_persist_headroom = -1;
int persist_headroom_set(int cus) {
  if (g_persist_headroom < 0) { const char* e = getenv("CVAE_PERSIST_HEADROOM"); g_persist_headroom = e ? atoi(e) : 0; }
  const int prev = g_persist_headroom;
  if (cus >= 0) g_persist_headroom = cus;
  return prev;
}

static int cu_count() {
  static int cus = 0;
  if (!cus) cus = persistent_grid_blocks() / 2;
  return cus;
}

bool rnn_persist_supported(int cell, int T, int R, int H, int ns) {
  if (!persist_enabled() || T < 1 || R < 1 || R > 32 || ns < 1 || ns > SKINNY_MAX_BATCH) return false;
  if (H % 128 != 0 || H > 512) return false;
  return H / P_HU <= P_MAXB && (H / P_HU) * ns <= cu_count() - persist_headroom_set(-1);
}

// CVAE_PERSIST_TAG=0: the flag + payload hand-off (two round trips per step) instead of the in-band one
static bool persist_tag_enabled() {
  static int on = -1;
  if (on < 0) {
    const char* e = getenv("CVAE_PERSIST_TAG");
    on = (e && e[0] == '0') ? 0 : 1;
  }
  return on == 1;
}

int launch_rnn_forward_persist(const RnnFwd* a, int ns, hipStream_t s, ZeroBatch* zb) {
  unsigned* sync = nullptr;
  int* status = nullptr;
  CVAE_TRY(persist_sync(s, &sync, &status));
  const bool tag = persist_tag_enabled() && split_bf16_enabled() && a[0].H % 32 == 0 && a[0].H / 32 <= 16;
  ZeroBatch local(s);
  ZeroBatch* z = zb ? zb : &local;
  if (tag) {
    // every slot the kernel will write starts as the sentinel (the initial slot holds the initial state)
    for (int k = 0; k < ns; ++k) {
      const long RH = (long)a[k].R * a[k].H;
      z->add_fill(a[k].hs + (a[k].reverse ? 0 : RH), (long)a[k].T * RH, P_SENT);
    }
  } else {
    z->add(reinterpret_cast<float*>(sync), P_FLAG_WORDS);
  }
  CVAE_TRY(z->flush());
  PFwdArgs p = {};
  p.T = a[0].T; p.R = a[0].R; p.H = a[0].H;
  p.nw = 0;
  static int nw_max = 0;
  if (!nw_max) { const char* e = getenv("CVAE_PERSIST_NW"); nw_max = e ? atoi(e) : 16; if (nw_max < 2 || nw_max > 16) nw_max = 16; }
  for (int nw = nw_max; nw >= 2 && !p.nw; nw >>= 1)                    // K per wave: multiple of 16, <= 64
    if (p.H % (16 * nw) == 0 && p.H / nw <= 64) p.nw = nw;
  for (int nw = 16; nw >= 2 && !p.nw; nw >>= 1)
    if (p.H % (16 * nw) == 0 && p.H / nw <= 64) p.nw = nw;
  const bool bf = split_bf16_enabled() && p.H % 32 == 0 && p.H / 32 <= 16;
  if (bf) p.nw = p.H / 32;                               // one 32-wide bf16 MFMA K block per wave
  CVAE_REQUIRE(p.nw >= 1 && p.nw <= 16 && p.H % p.nw == 0 && (p.H / p.nw) % 16 == 0 && p.H / p.nw <= 64,
               "rnn_forward_persist: H=%d does not split over the block's waves", p.H);
  p.flags = sync; p.status = status; p.stamps = persist_stamps();
  for (int k = 0; k < ns; ++k) {
    const RnnFwd& r = a[k];
    CVAE_REQUIRE((reinterpret_cast<uintptr_t>(r.hs) & 15) == 0 && (!r.cs || (reinterpret_cast<uintptr_t>(r.cs) & 15) == 0) &&
                 (!r.gates || (reinterpret_cast<uintptr_t>(r.gates) & 15) == 0) &&
                 (reinterpret_cast<uintptr_t>(r.W_hh) & 15) == 0, "rnn_forward_persist: operands must be 16-byte aligned");
    p.q[k] = PSeqF{r.xproj, r.xadd, r.W_hh, r.b_hh, r.lengths, r.hs, r.cs, r.gates, r.xmod, r.reverse};
  }
  const dim3 grid(p.H / P_HU, ns), block(P_THREADS);
  ProfScope prof(PK_RNN_PERSIST, s, 2.0 * ns * p.T * p.R * (a[0].cell == CELL_GRU ? 3.0 : 4.0) * p.H * p.H);
  if (tag) {
    static int pd = -1, pdc = -1, split = -1;
    if (pd < 0) {
      const char* e = getenv("CVAE_PERSIST_POLL_DELAY");
      pd = e ? atoi(e) : 16;
      e = getenv("CVAE_PERSIST_POLL_DELAY_CELL");
      pdc = e ? atoi(e) : 4;
      e = getenv("CVAE_PERSIST_FWD_SPLIT");              // row groups per sequence: 1 (HU 4), 2 (HU 8), 4 (HU 16)
      split = e ? atoi(e) : 4;
      if (split != 1 && split != 2) split = 4;
    }
    p.poll_delay = pd; p.poll_delay_cell = pdc;
    int rg = split;
    while (rg > 1 && p.H % (4 * rg) != 0) rg >>= 1;
    constexpr size_t lds = 2 * 16 * 512 * sizeof(float);               // 64 KB: double-buffered partial tiles
    static bool tag_attr_set = false;
    const void* fns[6] = {reinterpret_cast<const void*>(&rnn_fwd_persist_tag_kernel<CELL_GRU, 4, 1>),
                          reinterpret_cast<const void*>(&rnn_fwd_persist_tag_kernel<CELL_LSTM, 4, 1>),
                          reinterpret_cast<const void*>(&rnn_fwd_persist_tag_kernel<CELL_GRU, 8, 2>),
                          reinterpret_cast<const void*>(&rnn_fwd_persist_tag_kernel<CELL_LSTM, 8, 2>),
                          reinterpret_cast<const void*>(&rnn_fwd_persist_tag_kernel<CELL_GRU, 16, 4>),
                          reinterpret_cast<const void*>(&rnn_fwd_persist_tag_kernel<CELL_LSTM, 16, 4>)};
    if (!tag_attr_set) {
      for (const void* f : fns)
        CVAE_TRY(check_hip(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds),
                           "persist fwd LDS attribute"));
      tag_attr_set = true;
    }
    const bool gru = a[0].cell == CELL_GRU;
    if (rg == 4) {
      if (gru) hipLaunchKernelGGL((rnn_fwd_persist_tag_kernel<CELL_GRU, 16, 4>), grid, block, lds, s, p);
      else hipLaunchKernelGGL((rnn_fwd_persist_tag_kernel<CELL_LSTM, 16, 4>), grid, block, lds, s, p);
    } else if (rg == 2) {
      if (gru) hipLaunchKernelGGL((rnn_fwd_persist_tag_kernel<CELL_GRU, 8, 2>), grid, block, lds, s, p);
      else hipLaunchKernelGGL((rnn_fwd_persist_tag_kernel<CELL_LSTM, 8, 2>), grid, block, lds, s, p);
    } else {
      if (gru) hipLaunchKernelGGL((rnn_fwd_persist_tag_kernel<CELL_GRU, 4, 1>), grid, block, lds, s, p);
      else hipLaunchKernelGGL((rnn_fwd_persist_tag_kernel<CELL_LSTM, 4, 1>), grid, block, lds, s, p);
    }
  } else if (bf) {
    if (a[0].cell == CELL_GRU) hipLaunchKernelGGL((rnn_fwd_persist_kernel<CELL_GRU, true>), grid, block, 0, s, p);
    else hipLaunchKernelGGL((rnn_fwd_persist_kernel<CELL_LSTM, true>), grid, block, 0, s, p);
  } else {
    if (a[0].cell == CELL_GRU) hipLaunchKernelGGL((rnn_fwd_persist_kernel<CELL_GRU, false>), grid, block, 0, s, p);
    else hipLaunchKernelGGL((rnn_fwd_persist_kernel<CELL_LSTM, false>), grid, block, 0, s, p);
  }
  CVAE_LAUNCH_CHECK("rnn_fwd_persist_kernel");
  return 0;
}

bool rnn_persist_bwd_supported(int cell, int T, int R, int H, int ns) {
  static int on = -1;
  if (on < 0) {
    const char* e = getenv("CVAE_PERSISTENT_RNN_BWD");
    on = (e && e[0] == '0') ? 0 : 1;
  }
  return on && rnn_persist_supported(cell, T, R, H, ns);
}

static int persist_exch(hipStream_t s, long floats, float** out) {
  void* p = nullptr;
  CVAE_TRY(lib_scratch(LS_PERSIST_EXCH, s, (size_t)floats * sizeof(float), &p));
  *out = static_cast<float*>(p);
  return 0;
}

// In-band backward (rnn_bwd_persist_tag_kernel): CVAE_PERSIST_TAG_BWD=0 selects the flag kernel;
// CVAE_PERSIST_BWD_SPLIT = row groups per sequence: 1 (HU 4), 2 (HU 8), 4 (HU 16)
static int launch_rnn_backward_persist_tag(const RnnBwd* a, int ns, hipStream_t s, int rg) {
  unsigned* sync = nullptr;
  int* status = nullptr;
  CVAE_TRY(persist_sync(s, &sync, &status));
  PBwdArgs p = {};
  p.T = a[0].T; p.R = a[0].R; p.H = a[0].H;
  p.flags = sync; p.status = status; p.stamps = persist_stamps();
  const int G = a[0].cell == CELL_GRU ? 3 : 4;
  const int hu = 4 * rg, nsl = p.H / hu;
  const long ex_floats = (long)ns * 2 * nsl * nsl * 32 * hu;
  CVAE_TRY(persist_exch(s, ex_floats, &p.exch));
  {
    ZeroBatch arm(s);                                     // every piece starts as "not there yet"
    arm.add_fill(p.exch, ex_floats, P_SENT);
    CVAE_TRY(arm.flush());
  }
  static int pd = -1;
  if (pd < 0) { const char* e = getenv("CVAE_PERSIST_POLL_DELAY_BWD"); pd = e ? atoi(e) : 8; }
  p.poll_delay = pd;
  for (int k = 0; k < ns; ++k) {
    const RnnBwd& r = a[k];
    float* dgh = r.dgh ? r.dgh : r.dgx;
    CVAE_REQUIRE((reinterpret_cast<uintptr_t>(r.dgx) & 15) == 0 && (reinterpret_cast<uintptr_t>(dgh) & 15) == 0,
                 "rnn_backward_persist: operands must be 16-byte aligned");
    const int64_t* za = r.zero_after;
    if (!za && !r.lengths && r.dhs && !r.dh_last && !r.dc_last && p.R <= 32) {
      int64_t* zbuf = reinterpret_cast<int64_t*>(reinterpret_cast<char*>(sync) + P_ZA_OFFSET) + (size_t)k * 32;
      CVAE_TRY(launch_last_nonzero_step(p.T, p.R, p.H, r.dhs, zbuf, s));
      za = zbuf;
    }
    p.q[k] = PSeqB{r.dhs, r.dh_last, r.dc_last, r.hs, r.cs, r.gates, r.W_hh, r.lengths, r.dgx, dgh, r.dh0, r.dc0, r.reverse, za};
  }
  const int rpg = 32 / rg;
  const size_t lds = ((size_t)rpg * (p.H + 4) + 16 * 32 * 4) * sizeof(float);
  const dim3 grid(nsl * rg, ns), block(P_THREADS);
  const void* fns[6] = {reinterpret_cast<const void*>(&rnn_bwd_persist_tag_kernel<CELL_GRU, 4, 1>),
                        reinterpret_cast<const void*>(&rnn_bwd_persist_tag_kernel<CELL_LSTM, 4, 1>),
                        reinterpret_cast<const void*>(&rnn_bwd_persist_tag_kernel<CELL_GRU, 8, 2>),
                        reinterpret_cast<const void*>(&rnn_bwd_persist_tag_kernel<CELL_LSTM, 8, 2>),
                        reinterpret_cast<const void*>(&rnn_bwd_persist_tag_kernel<CELL_GRU, 16, 4>),
                        reinterpret_cast<const void*>(&rnn_bwd_persist_tag_kernel<CELL_LSTM, 16, 4>)};
  static bool attr_set = false;
  if (!attr_set) {
    for (const void* f : fns)
      CVAE_TRY(check_hip(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024),
                         "persist bwd LDS attribute"));
    attr_set = true;
  }
  ProfScope prof(PK_RNN_PERSIST_BWD, s, 2.0 * ns * p.T * p.R * G * p.H * p.H);
  const bool gru = a[0].cell == CELL_GRU;
  if (rg == 4) {
    if (gru) hipLaunchKernelGGL((rnn_bwd_persist_tag_kernel<CELL_GRU, 16, 4>), grid, block, lds, s, p);
    else hipLaunchKernelGGL((rnn_bwd_persist_tag_kernel<CELL_LSTM, 16, 4>), grid, block, lds, s, p);
  } else if (rg == 2) {
    if (gru) hipLaunchKernelGGL((rnn_bwd_persist_tag_kernel<CELL_GRU, 8, 2>), grid, block, lds, s, p);
    else hipLaunchKernelGGL((rnn_bwd_persist_tag_kernel<CELL_LSTM, 8, 2>), grid, block, lds, s, p);
  } else {
    if (gru) hipLaunchKernelGGL((rnn_bwd_persist_tag_kernel<CELL_GRU, 4, 1>), grid, block, lds, s, p);
    else hipLaunchKernelGGL((rnn_bwd_persist_tag_kernel<CELL_LSTM, 4, 1>), grid, block, lds, s, p);
  }
  CVAE_LAUNCH_CHECK("rnn_bwd_persist_tag_kernel");
  return 0;
}

int launch_rnn_backward_persist(const RnnBwd* a, int ns, hipStream_t s) {
  {
    static int tag = -1, split = 2;
    if (tag < 0) {
      const char* e = getenv("CVAE_PERSIST_TAG_BWD");
      tag = (e && e[0] == '0') ? 0 : 1;
      e = getenv("CVAE_PERSIST_BWD_SPLIT");
      split = e ? atoi(e) : 2;
      if (split != 1 && split != 4) split = 2;
    }
    int rg = split;
    while (rg > 1 && a[0].H % (4 * rg) != 0) rg >>= 1;
    if (tag && persist_tag_enabled() && split_bf16_enabled() && a[0].H % 16 == 0 && a[0].H / 16 <= 32 &&
        a[0].H / (4 * rg) <= P_MAXB)
      return launch_rnn_backward_persist_tag(a, ns, s, rg);
  }
  unsigned* sync = nullptr;
  int* status = nullptr;
  CVAE_TRY(persist_sync(s, &sync, &status));
  CVAE_TRY(launch_zero(reinterpret_cast<float*>(sync), P_FLAG_WORDS, s));
  PBwdArgs p = {};
  p.T = a[0].T; p.R = a[0].R; p.H = a[0].H;
  p.flags = sync; p.status = status; p.stamps = persist_stamps();
  const int G = a[0].cell == CELL_GRU ? 3 : 4;
  static int hu_env = 0;                       // CVAE_PERSIST_BWD_HU=4: the 4-unit blocks (more, thinner blocks)
  if (!hu_env) { const char* e = getenv("CVAE_PERSIST_BWD_HU"); hu_env = e ? atoi(e) : 8; if (hu_env != 4 && hu_env != 16) hu_env = 8; }
  // 16-unit blocks (a quarter of the blocks and flags of the 4-unit form; CVAE_PERSIST_BWD_HU=16) only with the bf16
  // product (K = 64: two K blocks).  Measured slower than 8 (training step 7.26 vs 7.08 ms): opt-in, kept for larger H
  const int hu = (hu_env == 16 && p.H % 16 == 0 && split_bf16_enabled()) ? 16 : ((hu_env >= 8 && p.H % 8 == 0) ? 8 : 4);
  const int nb = p.H / hu;
  CVAE_REQUIRE(p.H % 16 == 0 && p.H / 16 <= 32 && nb <= P_MAXB, "rnn_backward_persist: H=%d", p.H);
  CVAE_TRY(persist_exch(s, (long)SKINNY_MAX_BATCH * 2 * nb * nb * 32 * hu, &p.exch));
  for (int k = 0; k < ns; ++k) {
    const RnnBwd& r = a[k];
    float* dgh = r.dgh ? r.dgh : r.dgx;
    CVAE_REQUIRE((reinterpret_cast<uintptr_t>(r.dgx) & 15) == 0 && (reinterpret_cast<uintptr_t>(dgh) & 15) == 0,
                 "rnn_backward_persist: operands must be 16-byte aligned");
    // unpacked sequences (decoders): rows past their last non-zero output gradient stay exactly zero
    const int64_t* za = r.zero_after;
    if (!za && !r.lengths && r.dhs && !r.dh_last && !r.dc_last && p.R <= 32) {
      int64_t* zbuf = reinterpret_cast<int64_t*>(reinterpret_cast<char*>(sync) + P_ZA_OFFSET) + (size_t)k * 32;
      CVAE_TRY(launch_last_nonzero_step(p.T, p.R, p.H, r.dhs, zbuf, s));
      za = zbuf;
    }
    p.q[k] = PSeqB{r.dhs, r.dh_last, r.dc_last, r.hs, r.cs, r.gates, r.W_hh, r.lengths, r.dgx, dgh, r.dh0, r.dc0, r.reverse, za};
  }
  long lds_floats = 32L * (p.H + 4);
  if (lds_floats < 4096) lds_floats = 4096;
  const size_t lds = (size_t)lds_floats * sizeof(float);
  const dim3 grid(nb, ns), block(P_THREADS);
  static bool attr_set = false;
  if (!attr_set) {
    const void* fns[10] = {reinterpret_cast<const void*>(&rnn_bwd_persist_kernel<CELL_GRU, 16, true>),
                          reinterpret_cast<const void*>(&rnn_bwd_persist_kernel<CELL_LSTM, 16, true>),
                          reinterpret_cast<const void*>(&rnn_bwd_persist_kernel<CELL_GRU, 4, false>),
                          reinterpret_cast<const void*>(&rnn_bwd_persist_kernel<CELL_LSTM, 4, false>),
                          reinterpret_cast<const void*>(&rnn_bwd_persist_kernel<CELL_GRU, 8, false>),
                          reinterpret_cast<const void*>(&rnn_bwd_persist_kernel<CELL_LSTM, 8, false>),
                          reinterpret_cast<const void*>(&rnn_bwd_persist_kernel<CELL_GRU, 4, true>),
                          reinterpret_cast<const void*>(&rnn_bwd_persist_kernel<CELL_LSTM, 4, true>),
                          reinterpret_cast<const void*>(&rnn_bwd_persist_kernel<CELL_GRU, 8, true>),
                          reinterpret_cast<const void*>(&rnn_bwd_persist_kernel<CELL_LSTM, 8, true>)};
    for (const void* f : fns)
      CVAE_TRY(check_hip(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024),
                         "persist bwd LDS attribute"));
    attr_set = true;
  }
  ProfScope prof(PK_RNN_PERSIST_BWD, s, 2.0 * ns * p.T * p.R * G * p.H * p.H);
  const bool gru = a[0].cell == CELL_GRU;
  if (hu == 16) {
    if (gru) hipLaunchKernelGGL((rnn_bwd_persist_kernel<CELL_GRU, 16, true>), grid, block, lds, s, p);
    else hipLaunchKernelGGL((rnn_bwd_persist_kernel<CELL_LSTM, 16, true>), grid, block, lds, s, p);
  } else if (split_bf16_enabled()) {
    if (hu == 8) {
      if (gru) hipLaunchKernelGGL((rnn_bwd_persist_kernel<CELL_GRU, 8, true>), grid, block, lds, s, p);
      else hipLaunchKernelGGL((rnn_bwd_persist_kernel<CELL_LSTM, 8, true>), grid, block, lds, s, p);
    } else {
      if (gru) hipLaunchKernelGGL((rnn_bwd_persist_kernel<CELL_GRU, 4, true>), grid, block, lds, s, p);
      else hipLaunchKernelGGL((rnn_bwd_persist_kernel<CELL_LSTM, 4, true>), grid, block, lds, s, p);
    }
  } else if (hu == 8) {
    if (gru) hipLaunchKernelGGL((rnn_bwd_persist_kernel<CELL_GRU, 8, false>), grid, block, lds, s, p);
    else hipLaunchKernelGGL((rnn_bwd_persist_kernel<CELL_LSTM, 8, false>), grid, block, lds, s, p);
  } else {
    if (gru) hipLaunchKernelGGL((rnn_bwd_persist_kernel<CELL_GRU, 4, false>), grid, block, lds, s, p);
    else hipLaunchKernelGGL((rnn_bwd_persist_kernel<CELL_LSTM, 4, false>), grid, block, lds, s, p);
  }
  CVAE_LAUNCH_CHECK("rnn_bwd_persist_kernel");
  return 0;
}

// Synchronises the device and reports (then clears) the timeout word of the persistent kernels.
int persist_status_and_clear() {
  int* host = persist_status_host();
  if (!host) return 0;                                   // no whole-sequence kernel has run yet
  if (hipDeviceSynchronize() != hipSuccess) return -1;
  const int st = __atomic_load_n(host, __ATOMIC_ACQUIRE);
  if (st) __atomic_store_n(host, 0, __ATOMIC_RELEASE);
  if (g_stamps_on == 1 && persist_stamps()) {
    unsigned long long acc[16];
    if (hipMemcpy(acc, persist_stamps(), sizeof(acc), hipMemcpyDeviceToHost) == hipSuccess) {
      fprintf(stderr, "[cvae persist stamps, cycles of block (0,0)] loop-top->wait %llu | wait %llu | load+mfma %llu | reduce %llu | "
              "cell+stores %llu | drain+flag %llu | [6] %llu | [7] %llu  (bwd: top | wait | gather | cell | product+publish stores | drain+flag)\n", acc[0], acc[1], acc[2], acc[3], acc[4], acc[5], acc[6], acc[7]);
      if (hipMemset(persist_stamps(), 0, sizeof(acc)) != hipSuccess) return -1;
    }
  }
  return st ? 1 : 0;
}

// The same word without synchronising or clearing: 1 as soon as a timed-out kernel's store has landed.
int persist_status_peek() {
  int* host = persist_status_host();
  return host ? (__atomic_load_n(host, __ATOMIC_ACQUIRE) ? 1 : 0) : 0;
}

}  // namespace cvae
