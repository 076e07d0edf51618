// Bandwidth-shaped kernels of the path: batch layout (a1/a2), embedding gather / scatter,
// dropout-mask application, reparameterisation, KL, coupling loss, IW log-sum-exp, the MI
// pairwise density, and the clip + Adam step.  All are one-pass, coalesced along the
// fastest dimension; reductions use 64-wide wave shuffles.
#include "common.hpp"
#include "misc.hpp"
#include <math.h>

namespace cvae {

#define GRID1D(n) dim3(cdiv((n), 256)), dim3(256)
#define IDX1D(n)                                                      \
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;       \
  if (idx >= (long)(n)) return;

// ------------------------------------------------------------------ a1 / a2
// sel == null: sentence b of the batch is CSR row b; otherwise CSR row sel[b] of a corpus that is
// resident on the device (8f.2: a batch then costs one copy of B indices, no token traffic).
__global__ void pack_batch_kernel(const int64_t* __restrict__ flat, const int64_t* __restrict__ off,
                                  const int64_t* __restrict__ sel, int B,
                                  int max_len, int width, int64_t pad, int64_t go, int64_t eos,
                                  int64_t* __restrict__ bare, int64_t* __restrict__ go_out,
                                  int64_t* __restrict__ eos_out, int64_t* __restrict__ len) {
  IDX1D((long)B * (width + 1));
  const int b = (int)(idx / (width + 1)), j = (int)(idx % (width + 1));
  const int64_t row = sel ? sel[b] : b;
  const int64_t beg = off[row];
  int64_t l = off[row + 1] - beg;
  if (l > max_len) l = max_len;
  if (j == 0) len[b] = l;
  // bare[j] = id_j | pad ; go[j] = go, id_{j-1} | pad ; eos[j] = id_j, eos at j == l | pad
  if (j < width) bare[(long)b * width + j] = j < l ? flat[beg + j] : pad;
  go_out[(long)b * (width + 1) + j] = j == 0 ? go : (j - 1 < l ? flat[beg + j - 1] : pad);
  eos_out[(long)b * (width + 1) + j] = j < l ? flat[beg + j] : (j == l ? eos : pad);
}

int launch_pack_batch(const int64_t* flat, const int64_t* off, const int64_t* sel, int B, int max_len, int width, int64_t pad,
                      int64_t go, int64_t eos, int64_t* bare, int64_t* go_out, int64_t* eos_out, int64_t* len,
                      hipStream_t s) {
  CVAE_REQUIRE(B >= 0 && max_len >= 0 && width >= 0 && width <= max_len, "pack_batch: bad shape");
  if (B == 0) return 0;
  CVAE_REQUIRE(flat && off && bare && go_out && eos_out && len, "pack_batch: null operand");
  hipLaunchKernelGGL(pack_batch_kernel, GRID1D((long)B * (width + 1)), 0, s, flat, off, sel, B, max_len, width, pad, go,
                     eos, bare, go_out, eos_out, len);
  CVAE_LAUNCH_CHECK("pack_batch_kernel");
  return 0;
}

// stable descending rank: pos(i) = #{j : len[j] > len[i]} + #{j < i : len[j] == len[i]}
__global__ void sort_resort_kernel(const int64_t* __restrict__ len, int B, int64_t* __restrict__ s_idx,
                                   int64_t* __restrict__ res_idx) {
  IDX1D(B);
  const int64_t li = len[idx];
  int pos = 0;
  for (int j = 0; j < B; ++j) {
    const int64_t lj = len[j];
    pos += (lj > li) || (lj == li && j < idx);
  }
  res_idx[idx] = pos;
  s_idx[pos] = idx;
}

int launch_sort_resort(const int64_t* len, int B, int64_t* s_idx, int64_t* res_idx, hipStream_t s) {
  if (B == 0) return 0;
  CVAE_REQUIRE(len && s_idx && res_idx, "sort_resort: null operand");
  hipLaunchKernelGGL(sort_resort_kernel, GRID1D(B), 0, s, len, B, s_idx, res_idx);
  CVAE_LAUNCH_CHECK("sort_resort_kernel");
  return 0;
}

// ------------------------------------------------------------------ embedding gather / scatter
// out[(slot0 + t), r, 0:E]   = emb[ids[b*T + t]] * mask
// out[(slot0 + t), r, E:2E]  = cond[r] * mask                      (when cond != null, W = 2E)
// prefix slot (slot0 == 1):  out[0, r, :] = [cond[r], cond[r]]
__global__ void embed_kernel(int T, int R, int E, int W, int slot0, const int64_t* __restrict__ ids, int Bids,
                             const float* __restrict__ emb, const float* __restrict__ cond,
                             const float* __restrict__ mask, const int64_t* __restrict__ mask_rows,
                             float* __restrict__ out) {
  IDX1D((long)(T + slot0) * R * W);
  const int e = (int)(idx % W);
  const int r = (int)((idx / W) % R);
  const int slot = (int)(idx / ((long)W * R));
  float v;
  if (slot < slot0) {
    v = cond[(long)r * E + (e < E ? e : e - E)];
  } else {
    const int t = slot - slot0;
    if (e < E) v = emb[ids[(long)(r % Bids) * T + t] * E + e];
    else v = cond[(long)r * E + (e - E)];
    if (mask) {
      const long mr = mask_rows ? mask_rows[r] : r;
      v *= mask[((long)t * R + mr) * W + e];
    }
  }
  out[idx] = v;
}

int launch_embed(int T, int R, int E, int with_cond, int prefix, const int64_t* ids, int Bids, const float* emb,
                 const float* cond, const float* mask, const int64_t* mask_rows, float* out, hipStream_t s) {
  const int W = with_cond ? 2 * E : E;
  CVAE_REQUIRE(!prefix || with_cond, "embed: a prefix slot needs cond");
  CVAE_REQUIRE(ids && emb && out && (!with_cond || cond) && Bids > 0, "embed: null operand");
  const long n = (long)(T + (prefix ? 1 : 0)) * R * W;
  if (n == 0) return 0;
  hipLaunchKernelGGL(embed_kernel, GRID1D(n), 0, s, T, R, E, W, prefix ? 1 : 0, ids, Bids, emb, cond, mask,
                     mask_rows, out);
  CVAE_LAUNCH_CHECK("embed_kernel");
  return 0;
}

// demb[ids[b,t], e] += dX[slot0 + t, b, e] * mask       (float atomics: rows repeat)
__global__ void embed_scatter_kernel(int T, int B, int E, int W, int slot0, const int64_t* __restrict__ ids,
                                     const float* __restrict__ dX, const float* __restrict__ mask,
                                     const int64_t* __restrict__ mask_rows, float* __restrict__ demb) {
  IDX1D((long)T * B * E);
  const int e = (int)(idx % E);
  const int b = (int)((idx / E) % B);
  const int t = (int)(idx / ((long)E * B));
  float v = dX[((long)(t + slot0) * B + b) * W + e];
  if (mask) {
    const long mr = mask_rows ? mask_rows[b] : b;
    v *= mask[((long)t * B + mr) * W + e];
  }
  if (v != 0.f) atomicAdd(demb + ids[(long)b * T + t] * E + e, v);   // padded steps / dropped elements: exact zeros
}

// Deterministic form: the same sums through 64-bit fixed point (2^-36 units; integer adds commute, so the order in
// which the atomics land does not matter), then demb += fix * 2^-36 in one pass.  Resolution 1.5e-11, range +-1.3e8
// per element.  An addend that is not finite or lies outside +-1e6 (a diverged step) does not go through fixed
// point: it is added to demb as a float, so a NaN / Inf / huge gradient stays visible in demb and in the clip norm
// exactly as in the default path and in the reference (only such a step loses bit-reproducibility).
constexpr double EMB_FIX_SCALE = 68719476736.0;   // 2^36
__global__ void embed_scatter_fix_kernel(int T, int B, int E, int W, int slot0, const int64_t* __restrict__ ids,
                                         const float* __restrict__ dX, const float* __restrict__ mask,
                                         const int64_t* __restrict__ mask_rows, unsigned long long* __restrict__ fix,
                                         float* __restrict__ demb) {
  IDX1D((long)T * B * E);
  const int e = (int)(idx % E);
  const int b = (int)((idx / E) % B);
  const int t = (int)(idx / ((long)E * B));
  float v = dX[((long)(t + slot0) * B + b) * W + e];
  if (mask) {
    const long mr = mask_rows ? mask_rows[b] : b;
    v *= mask[((long)t * B + mr) * W + e];
  }
  if (v != 0.f) {
    const long o = ids[(long)b * T + t] * E + e;
    if (!(fabsf(v) <= 1.0e6f)) {                      // NaN, Inf or out of the fixed-point range: propagate as it is
      atomicAdd(demb + o, v);
    } else {
      const long long q = __double2ll_rn((double)v * EMB_FIX_SCALE);
      atomicAdd(fix + o, (unsigned long long)q);      // two's complement: wraps like a signed add
    }
  }
}
__global__ void embed_fix_apply_kernel(long n, unsigned long long* __restrict__ fix, float* __restrict__ demb) {
  IDX1D(n);
  const long long q = (long long)fix[idx];
  if (q != 0) {
    demb[idx] += (float)((double)q * (1.0 / EMB_FIX_SCALE));
    fix[idx] = 0ull;                                 // the buffer is all zero again for the next call
  }
}

int launch_embed_scatter(int T, int B, int E, int W, int slot0, const int64_t* ids, const float* dX,
                         const float* mask, const int64_t* mask_rows, float* demb, hipStream_t s, long V) {
  if ((long)T * B * E == 0) return 0;
  if (deterministic_enabled() && V > 0) {
    void* buf = nullptr;
    bool fresh = false;
    CVAE_TRY(lib_scratch(LS_EMB_FIX, s, (size_t)V * E * sizeof(unsigned long long), &buf, 1, &fresh));
    unsigned long long* fix = static_cast<unsigned long long*>(buf);
    hipLaunchKernelGGL(embed_scatter_fix_kernel, GRID1D((long)T * B * E), 0, s, T, B, E, W, slot0, ids, dX, mask,
                       mask_rows, fix, demb);
    CVAE_LAUNCH_CHECK("embed_scatter_fix_kernel");
    hipLaunchKernelGGL(embed_fix_apply_kernel, GRID1D((long)V * E), 0, s, (long)V * E, fix, demb);
    CVAE_LAUNCH_CHECK("embed_fix_apply_kernel");
    return 0;
  }
  hipLaunchKernelGGL(embed_scatter_kernel, GRID1D((long)T * B * E), 0, s, T, B, E, W, slot0, ids, dX, mask,
                     mask_rows, demb);
  CVAE_LAUNCH_CHECK("embed_scatter_kernel");
  return 0;
}

// dinit[b,e] = dX[0,b,e] + dX[0,b,E+e] + sum_t dX[1+t,b,E+e] * mask[t,b,E+e]
__global__ __launch_bounds__(1024) void dec_dinit_kernel(int T, int B, int E, const float* __restrict__ dX,
                                                         const float* __restrict__ mask, float* __restrict__ dinit) {
  // block = (b, 64 columns); its 16 waves take the time steps round-robin, fixed summation order
  __shared__ float part[16][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b = blockIdx.y, e = blockIdx.x * 64 + lane;
  const long W = 2L * E;
  float acc = 0.f;
  if (e < E)
    for (int t = wave; t < T; t += 16) {
      float v = dX[((long)(t + 1) * B + b) * W + E + e];
      if (mask) v *= mask[((long)t * B + b) * W + E + e];
      acc += v;
    }
  part[wave][lane] = acc;
  __syncthreads();
  if (wave == 0 && e < E) {
    float tot = dX[(long)b * W + e] + dX[(long)b * W + E + e];
#pragma unroll
    for (int w = 0; w < 16; ++w) tot += part[w][lane];
    dinit[(long)b * E + e] = tot;
  }
}

int launch_dec_dinit(int T, int B, int E, const float* dX, const float* mask, float* dinit, hipStream_t s) {
  if ((long)B * E == 0) return 0;
  hipLaunchKernelGGL(dec_dinit_kernel, dim3(cdiv(E, 64), B), dim3(1024), 0, s, T, B, E, dX, mask, dinit);
  CVAE_LAUNCH_CHECK("dec_dinit_kernel");
  return 0;
}

// ------------------------------------------------------------------ elementwise helpers
__global__ void mul_kernel(long n, const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ o) {
  IDX1D(n);
  o[idx] = b ? a[idx] * b[idx] : a[idx];
}
int launch_mul(long n, const float* a, const float* b, float* o, hipStream_t s) {
  if (n == 0) return 0;
  hipLaunchKernelGGL(mul_kernel, GRID1D(n), 0, s, n, a, b, o);
  CVAE_LAUNCH_CHECK("mul_kernel");
  return 0;
}

__global__ void act_bwd_kernel(long n, int act, const float* __restrict__ y, const float* __restrict__ dy,
                               float* __restrict__ o) {
  IDX1D(n);
  const float yv = y[idx], g = dy[idx];
  float r = g;
  if (act == ACT_TANH) r = g * (1.f - yv * yv);
  else if (act == ACT_RELU) r = yv > 0.f ? g : 0.f;
  else if (act == ACT_EXP) r = g * yv;
  o[idx] = r;
}
int launch_act_bwd(long n, int act, const float* y, const float* dy, float* o, hipStream_t s) {
  if (n == 0) return 0;
  hipLaunchKernelGGL(act_bwd_kernel, GRID1D(n), 0, s, n, act, y, dy, o);
  CVAE_LAUNCH_CHECK("act_bwd_kernel");
  return 0;
}

__global__ void axpy_kernel(long n, const float* __restrict__ x, float a, float* __restrict__ y) {
  IDX1D(n);
  y[idx] += a * x[idx];
}
int launch_axpy(long n, const float* x, float a, float* y, hipStream_t s) {
  if (n == 0) return 0;
  hipLaunchKernelGGL(axpy_kernel, GRID1D(n), 0, s, n, x, a, y);
  CVAE_LAUNCH_CHECK("axpy_kernel");
  return 0;
}

// p[0..n) = 0 (a kernel launch costs the host ~4 us; hipMemsetAsync ~20 us)
__global__ __launch_bounds__(256) void zero_kernel(long n4, long n, float* __restrict__ p) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long stride = (long)gridDim.x * blockDim.x;
  float4* p4 = reinterpret_cast<float4*>(p);
  for (long k = i; k < n4; k += stride) p4[k] = make_float4(0.f, 0.f, 0.f, 0.f);
  for (long k = 4 * n4 + i; k < n; k += stride) p[k] = 0.f;
}
int launch_zero(float* p, long n, hipStream_t s) {
  if (n <= 0) return 0;
  if (!aligned16(p)) return check_hip(hipMemsetAsync(p, 0, (size_t)n * sizeof(float), s), "zero (unaligned)");
  const long n4 = n / 4;
  long blocks = (n4 + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(zero_kernel, dim3((unsigned)blocks), dim3(256), 0, s, n4, n, p);
  CVAE_LAUNCH_CHECK("zero_kernel");
  return 0;
}

// Several buffers zeroed by ONE launch (a launch is ~4.6 us of GPU timeline whatever its size):
// blockIdx.y = buffer.  Unaligned buffers take the scalar loop.
__global__ __launch_bounds__(256) void zero_multi_kernel(ZeroList z) {
  float* p = z.p[blockIdx.y];
  const long n = z.n[blockIdx.y];
  const float f = __uint_as_float(z.bits[blockIdx.y]);
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long stride = (long)gridDim.x * blockDim.x;
  if ((reinterpret_cast<uintptr_t>(p) & 15) == 0) {
    const long n4 = n / 4;
    float4* p4 = reinterpret_cast<float4*>(p);
    for (long k = i; k < n4; k += stride) p4[k] = make_float4(f, f, f, f);
    for (long k = 4 * n4 + i; k < n; k += stride) p[k] = f;
  } else {
    for (long k = i; k < n; k += stride) p[k] = f;
  }
}
void ZeroBatch::add(float* p, long n) { add_fill(p, n, 0u); }
void ZeroBatch::add_fill(float* p, long n, unsigned bits) {
  if (n <= 0 || !p) return;
  if (z.count == ZeroList::MAX) flush_status |= flush();
  z.p[z.count] = p; z.n[z.count] = n; z.bits[z.count] = bits; ++z.count;
}
int ZeroBatch::flush() {
  if (z.count == 0) return flush_status;
  long most = 0;
  for (int i = 0; i < z.count; ++i) most = z.n[i] > most ? z.n[i] : most;
  long blocks = (most / 4 + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(zero_multi_kernel, dim3((unsigned)blocks, z.count), dim3(256), 0, s, z);
  z.count = 0;
  CVAE_LAUNCH_CHECK("zero_multi_kernel");
  return flush_status;
}

// dst[0..n) = src[0..n)  (same host-cost argument as launch_zero vs hipMemcpyAsync)
__global__ __launch_bounds__(256) void copy_kernel(long n, const float* __restrict__ src, float* __restrict__ dst) {
  const long stride = (long)gridDim.x * blockDim.x;
  for (long k = (long)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += stride) dst[k] = src[k];
}
int launch_copy(float* dst, const float* src, long n, hipStream_t s) {
  if (n <= 0) return 0;
  long blocks = (n + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(copy_kernel, dim3((unsigned)blocks), dim3(256), 0, s, n, src, dst);
  CVAE_LAUNCH_CHECK("copy_kernel");
  return 0;
}

// out[r] = sum_t x[t*R + r]: a block owns 64 columns; its 16 waves take the time steps
// round-robin (fixed summation order: deterministic).
__global__ __launch_bounds__(1024) void sum_time_kernel(int T, int R, const float* __restrict__ x,
                                                        float* __restrict__ out) {
  __shared__ float part[16][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = blockIdx.x * 64 + lane;
  float acc = 0.f;
  if (r < R)
    for (int t = wave; t < T; t += 16) acc += x[(long)t * R + r];
  part[wave][lane] = acc;
  __syncthreads();
  if (wave == 0 && r < R) {
    float tot = 0.f;
#pragma unroll
    for (int w = 0; w < 16; ++w) tot += part[w][lane];
    out[r] = tot;
  }
}
int launch_sum_time(const float* x, int T, int R, float* out, hipStream_t s) {
  if (R == 0) return 0;
  hipLaunchKernelGGL(sum_time_kernel, dim3(cdiv(R, 64)), dim3(1024), 0, s, T, R, x, out);
  CVAE_LAUNCH_CHECK("sum_time_kernel");
  return 0;
}

// g_row[t*R + r] = gsent[r]
__global__ void bcast_time_kernel(int T, int R, const float* __restrict__ g, float* __restrict__ out) {
  IDX1D((long)T * R);
  out[idx] = g[idx % R];
}
int launch_bcast_time(const float* g, int T, int R, float* out, hipStream_t s) {
  if ((long)T * R == 0) return 0;
  hipLaunchKernelGGL(bcast_time_kernel, GRID1D((long)T * R), 0, s, T, R, g, out);
  CVAE_LAUNCH_CHECK("bcast_time_kernel");
  return 0;
}

// dst[r, c0 + c] = src[r, c]   /   dst[r, c] = src[r, c0 + c]
__global__ void copy_cols_kernel(int R, int C, const float* __restrict__ src, long lds, int soff,
                                 float* __restrict__ dst, long ldd, int doff) {
  IDX1D((long)R * C);
  const int c = (int)(idx % C), r = (int)(idx / C);
  dst[(long)r * ldd + doff + c] = src[(long)r * lds + soff + c];
}
int launch_copy_cols(int R, int C, const float* src, long lds, int soff, float* dst, long ldd, int doff,
                     hipStream_t s) {
  if ((long)R * C == 0) return 0;
  hipLaunchKernelGGL(copy_cols_kernel, GRID1D((long)R * C), 0, s, R, C, src, lds, soff, dst, ldd, doff);
  CVAE_LAUNCH_CHECK("copy_cols_kernel");
  return 0;
}

__global__ void targets_tm_kernel(int B, int T, int R, const int64_t* __restrict__ eos, int64_t* __restrict__ out) {
  IDX1D((long)T * R);
  const int r = (int)(idx % R), t = (int)(idx / R);
  out[idx] = eos[(long)(r % B) * T + t];
}
int launch_targets_tm(const int64_t* eos, int B, int T, int R, int64_t* out, hipStream_t s) {
  if ((long)T * R == 0) return 0;
  hipLaunchKernelGGL(targets_tm_kernel, GRID1D((long)T * R), 0, s, B, T, R, eos, out);
  CVAE_LAUNCH_CHECK("targets_tm_kernel");
  return 0;
}
__global__ void targets_sm_kernel(int B, int T, int K, const int64_t* __restrict__ eos, int64_t* __restrict__ out) {
  const long R = (long)B * K;
  IDX1D((long)T * R);
  const int r = (int)(idx % R), t = (int)(idx / R);
  out[idx] = eos[(long)(r / K) * T + t];
}
int launch_targets_sm(const int64_t* eos, int B, int T, int K, int64_t* out, hipStream_t s) {
  if ((long)T * B * K == 0) return 0;
  hipLaunchKernelGGL(targets_sm_kernel, GRID1D((long)T * B * K), 0, s, B, T, K, eos, out);
  CVAE_LAUNCH_CHECK("targets_sm_kernel");
  return 0;
}

// Greedy decoding (CV/Modules/decoder.py:176-201): index of the row maximum, entry `mask_id` excluded
// (the reference sets the <unk> logit to -inf first); ties go to the lowest index, as torch.argmax.
// One block per row.  Writes the id to out[b*out_ld + out_off] and to cur[b] (next step's input).
__global__ __launch_bounds__(256) void argmax_rows_kernel(int V, const float* __restrict__ logits, long ld,
                                                          int mask_id, int64_t* __restrict__ out, long out_ld,
                                                          long out_off, int64_t* __restrict__ cur) {
  __shared__ float smax[4];
  __shared__ int sidx[4];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* row = logits + (long)b * ld;
  float best = -INFINITY;
  int bi = 0x7fffffff;
  for (int v = tid; v < V; v += 256) {
    const float x = (v == mask_id) ? -INFINITY : row[v];
    if (x > best || (x == best && v < bi)) { best = x; bi = v; }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const float ob = __shfl_down(best, off, 64);
    const int oi = __shfl_down(bi, off, 64);
    if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
  }
  if (lane == 0) { smax[wave] = best; sidx[wave] = bi; }
  __syncthreads();
  if (tid == 0) {
#pragma unroll
    for (int w = 1; w < 4; ++w)
      if (smax[w] > best || (smax[w] == best && sidx[w] < bi)) { best = smax[w]; bi = sidx[w]; }
    out[(long)b * out_ld + out_off] = bi;
    cur[b] = bi;
  }
}
int launch_argmax_rows(int B, int V, const float* logits, long ld, int mask_id, int64_t* out, long out_ld, long out_off,
                       int64_t* cur, hipStream_t s) {
  if (B == 0) return 0;
  CVAE_REQUIRE(V > 0 && (V > 1 || mask_id != 0), "argmax_rows: nothing to choose from");
  hipLaunchKernelGGL(argmax_rows_kernel, dim3(B), dim3(256), 0, s, V, logits, ld, mask_id, out, out_ld, out_off, cur);
  CVAE_LAUNCH_CHECK("argmax_rows_kernel");
  return 0;
}

// Backward of a small linear layer y = act(x W^T + b) with M <= 64 rows (the Gaussian heads, toInit,
// PostRepr at training batch sizes: a5 / a6), in ONE launch instead of act_bwd + 2 GEMMs + zero + colsum:
//   dpre = dy * act'(y) (recomputed by every block from y, dy);
//   blocks [0, nA): dW[n-tile of 16][k-tile of 64] = dpre^T x, and db for the k-tile-0 blocks;
//   blocks [nA, nA + nB): dx[:, k-tile of 16] = dpre W.
// Plain fp32 FMAs: at most a few MFLOP, the cost is the launch.
__device__ __forceinline__ float act_grad(int act, float y, float g) {
  if (act == ACT_TANH) return g * (1.f - y * y);
  if (act == ACT_RELU) return y > 0.f ? g : 0.f;
  if (act == ACT_EXP) return g * y;
  return g;
}
__global__ __launch_bounds__(256) void linear_bwd_small_kernel(int M, int N, int K, int act,
                                                               const float* __restrict__ x,
                                                               const float* __restrict__ W,
                                                               const float* __restrict__ y,
                                                               const float* __restrict__ dy, float* __restrict__ dx,
                                                               float* __restrict__ dW, float* __restrict__ db, int nA,
                                                               int kt) {
  __shared__ float sd[64][17];
  __shared__ __attribute__((aligned(16))) float sb[64][68];      // x rows (dW blocks) / W rows (dx blocks: 16 used)
  const int tid = threadIdx.x;
  if ((int)blockIdx.x < nA) {
    const int n0 = ((int)blockIdx.x / kt) * 16, k0 = ((int)blockIdx.x % kt) * 64;
    for (int i = tid; i < 64 * 16; i += 256) {
      const int m = i >> 4, n = n0 + (i & 15);
      float v = 0.f;
      if (m < M && n < N) v = act_grad(act, act ? y[(long)m * N + n] : 0.f, dy[(long)m * N + n]);
      sd[m][i & 15] = v;
    }
    for (int i = tid; i < 64 * 64; i += 256) {
      const int m = i >> 6, k = k0 + (i & 63);
      sb[m][i & 63] = (m < M && k < K) ? x[(long)m * K + k] : 0.f;
    }
    __syncthreads();
    const int tn = tid >> 4, tk = (tid & 15) * 4;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int m = 0; m < M; ++m) {
      const float d = sd[m][tn];
      const float4 xv = *reinterpret_cast<const float4*>(&sb[m][tk]);
      acc[0] += d * xv.x; acc[1] += d * xv.y; acc[2] += d * xv.z; acc[3] += d * xv.w;
    }
    if (n0 + tn < N)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (k0 + tk + j < K) dW[(long)(n0 + tn) * K + k0 + tk + j] = acc[j];
    if (k0 == 0 && tid < 16 && n0 + tid < N) {
      float sum = 0.f;
      for (int m = 0; m < M; ++m) sum += sd[m][tid];
      db[n0 + tid] = sum;
    }
  } else {
    // dx: all M rows x 16 columns per block (fp32 FMAs issue at 4 cycles each and a block has one wave
    // per SIMD: narrow column tiles spread the N-long dot products over more CUs); N is walked in chunks
    // of 64 whose operands are fetched into registers one chunk ahead
    const int k0 = ((int)blockIdx.x - nA) * 16;
    const int g = tid >> 4, tk = tid & 15;                       // rows 4g .. 4g+3, column k0 + tk
    __shared__ __attribute__((aligned(16))) float sdd[64][68];   // dpre chunk [64 rows][64 n]
    float (*sdw)[68] = sb;                                       // W chunk [64 n][16 k]
    const float* yp = act ? y : dy;                              // act == 0: the value is not used
    float yv[16], gv[16], pd[16], pw[4];
    auto fetch = [&](int n0) {
      // unconditional loads from clamped addresses, selected afterwards: all loads in flight together
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const int i = tid + 256 * j;
        const int m = i >> 6, n = n0 + (i & 63);
        const long o = (long)min(m, M - 1) * N + min(n, N - 1);
        yv[j] = yp[o]; gv[j] = dy[o];
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int i = tid + 256 * j;
        const int nw = n0 + (i >> 4), k = k0 + (i & 15);
        pw[j] = W[(long)min(nw, N - 1) * K + min(k, K - 1)];
      }
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const int i = tid + 256 * j;
        const int m = i >> 6, n = n0 + (i & 63);
        pd[j] = (m < M && n < N) ? act_grad(act, yv[j], gv[j]) : 0.f;
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int i = tid + 256 * j;
        const int nw = n0 + (i >> 4), k = k0 + (i & 15);
        pw[j] = (nw < N && k < K) ? pw[j] : 0.f;
      }
    };
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    fetch(0);
    for (int n0 = 0; n0 < N; n0 += 64) {
      __syncthreads();
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const int i = tid + 256 * j;
        sdd[i >> 6][i & 63] = pd[j];
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int i = tid + 256 * j;
        sdw[i >> 4][i & 15] = pw[j];
      }
      __syncthreads();
      if (n0 + 64 < N) fetch(n0 + 64);
      if (4 * g < M) {
        // the chunk is zero-padded to 64: constant trip count, reads batched ahead of the FMAs
#pragma unroll 4
        for (int nn = 0; nn < 64; nn += 4) {
          float wv[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) wv[e] = sdw[nn + e][tk];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float4 d = *reinterpret_cast<const float4*>(&sdd[4 * g + r][nn]);
            acc[r] += d.x * wv[0] + d.y * wv[1] + d.z * wv[2] + d.w * wv[3];
          }
        }
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (4 * g + r < M && k0 + tk < K) dx[(long)(4 * g + r) * K + k0 + tk] = acc[r];
  }
}
bool linear_bwd_small_supported(int M, int N, int K) {
  static int on = -1;
  if (on < 0) { const char* e = getenv("CVAE_FUSED_SMALL_LINEAR"); on = (e && e[0] == '0') ? 0 : 1; }
  return on && M >= 1 && M <= 64 && N >= 1 && K >= 1 && (long)N * K <= 1024L * 1024;
}
int launch_linear_bwd_small(int M, int N, int K, int act, const float* x, const float* W, const float* y,
                            const float* dy, float* dx, float* dW, float* db, hipStream_t s) {
  CVAE_REQUIRE(linear_bwd_small_supported(M, N, K), "linear_bwd_small: M=%d N=%d K=%d not supported", M, N, K);
  CVAE_REQUIRE(act == ACT_NONE || y, "linear_bwd_small: act needs y");
  const int kt = (K + 63) / 64, nA = ((N + 15) / 16) * kt, nB = dx ? (K + 15) / 16 : 0;
  hipLaunchKernelGGL(linear_bwd_small_kernel, dim3(nA + nB), dim3(256), 0, s, M, N, K, act, x, W, y, dy, dx, dW, db,
                     nA, kt);
  CVAE_LAUNCH_CHECK("linear_bwd_small_kernel");
  return 0;
}

// ---- a11: loss assembly (CV/Experiment.py:753-757): batch means of up to six per-sentence terms and their
// weighted sum, in one launch (the torch expression is ~20 tiny kernels forward and ~10 backward).
constexpr int LOSS_MAX_TERMS = 6;      // seq, kl, seq_couple, couple, bow + one spare
struct LossTerms { const float* t[LOSS_MAX_TERMS]; float* d[LOSS_MAX_TERMS]; float coef[LOSS_MAX_TERMS]; };
__global__ __launch_bounds__(256) void loss_assemble_kernel(LossTerms a, int n, int B, float* __restrict__ out) {
  __shared__ float red[LOSS_MAX_TERMS][4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float tot = 0.f;
  for (int i = 0; i < n; ++i) {
    float s = 0.f;
    for (int b = tid; b < B; b += 256) s += a.t[i][b];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if (lane == 0) red[i][wave] = s;
  }
  __syncthreads();
  if (tid == 0) {
    for (int i = 0; i < n; ++i) {
      const float m = (red[i][0] + red[i][1] + red[i][2] + red[i][3]) / (float)B;
      out[1 + i] = m;
      tot += a.coef[i] * m;
    }
    out[0] = tot;
  }
}
__global__ void loss_assemble_bwd_kernel(LossTerms a, int n, int B, const float* __restrict__ gtot) {
  IDX1D((long)n * B);
  const int i = (int)(idx / B), b = (int)(idx % B);
  a.d[i][b] = gtot[0] * a.coef[i] / (float)B;
}
int launch_loss_assemble(int n, int B, const float* const* terms, const float* coef, float* out, hipStream_t s) {
  CVAE_REQUIRE(n >= 1 && n <= LOSS_MAX_TERMS && B > 0 && terms && coef && out, "loss_assemble: bad arguments (n=%d)", n);
  LossTerms a = {};
  for (int i = 0; i < n; ++i) {
    CVAE_REQUIRE(terms[i], "loss_assemble: null term %d", i);
    a.t[i] = terms[i]; a.coef[i] = coef[i];
  }
  hipLaunchKernelGGL(loss_assemble_kernel, dim3(1), dim3(256), 0, s, a, n, B, out);
  CVAE_LAUNCH_CHECK("loss_assemble_kernel");
  return 0;
}
int launch_loss_assemble_bwd(int n, int B, const float* gtot, const float* coef, float* const* dterms, hipStream_t s) {
  CVAE_REQUIRE(n >= 1 && n <= LOSS_MAX_TERMS && B > 0 && gtot && coef && dterms, "loss_assemble_backward: bad arguments (n=%d)", n);
  LossTerms a = {};
  for (int i = 0; i < n; ++i) {
    CVAE_REQUIRE(dterms[i], "loss_assemble_backward: null gradient %d", i);
    a.d[i] = dterms[i]; a.coef[i] = coef[i];
  }
  hipLaunchKernelGGL(loss_assemble_bwd_kernel, GRID1D((long)n * B), 0, s, a, n, B, gtot);
  CVAE_LAUNCH_CHECK("loss_assemble_bwd_kernel");
  return 0;
}

// ---- f3: beam branch of Decoder.forward (CV/Modules/decoder.py:114-168; vae_couple.py:122-172) -----------
// Per row: log_softmax over V with entry `mask_id` at -inf (decoder.py:138-142), then the K largest
// entries in descending order (torch.topk, decoder.py:143; ties go to the lower index here).  One block
// per row; K + 2 passes over the row (V floats from L2): max, sum, then K times "the best entry that
// comes strictly after the previous pick in (value desc, index asc) order" -- no exclusion list.
__device__ __forceinline__ void block_best(float& v, int& i, float* sv, int* si) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const float ov = __shfl_down(v, off, 64);
    const int oi = __shfl_down(i, off, 64);
    if (ov > v || (ov == v && oi < i)) { v = ov; i = oi; }
  }
  __syncthreads();                                   // sv / si of the previous round are consumed
  if (lane == 0) { sv[wave] = v; si[wave] = i; }
  __syncthreads();
  v = sv[0]; i = si[0];
#pragma unroll
  for (int w = 1; w < 4; ++w)
    if (sv[w] > v || (sv[w] == v && si[w] < i)) { v = sv[w]; i = si[w]; }
}
__global__ __launch_bounds__(256) void logsoftmax_topk_kernel(int V, int K, const float* __restrict__ logits, long ld,
                                                              int mask_id, float* __restrict__ lv,
                                                              int* __restrict__ li) {
  __shared__ float sv[4];
  __shared__ int si[4];
  __shared__ float ssum[4];
  const int r = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* row = logits + (long)r * ld;
  float best = -INFINITY;
  int bi = 0x7fffffff;
  for (int v = tid; v < V; v += 256) {
    const float x = (v == mask_id) ? -INFINITY : row[v];
    if (x > best || (x == best && v < bi)) { best = x; bi = v; }
  }
  block_best(best, bi, sv, si);
  const float m = best;
  float sum = 0.f;
  for (int v = tid; v < V; v += 256)
    if (v != mask_id) sum += expf(row[v] - m);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) sum += __shfl_down(sum, off, 64);
  if (lane == 0) ssum[wave] = sum;
  __syncthreads();
  const float lsum = logf(ssum[0] + ssum[1] + ssum[2] + ssum[3]);
  float pv = best;
  int pi = bi;
  for (int j = 0; j < K; ++j) {
    if (j > 0) {
      float cv = -INFINITY;
      int ci = 0x7fffffff;
      for (int v = tid; v < V; v += 256) {
        if (v == mask_id) continue;
        const float x = row[v];
        const bool after = x < pv || (x == pv && v > pi);
        if (after && (x > cv || (x == cv && v < ci))) { cv = x; ci = v; }
      }
      block_best(cv, ci, sv, si);
      pv = cv; pi = ci;
    }
    if (tid == 0) { lv[(long)r * K + j] = (pv - m) - lsum; li[(long)r * K + j] = pi; }
  }
}
int launch_logsoftmax_topk(int R, int V, int K, const float* logits, long ld, int mask_id, float* lv, int* li,
                           hipStream_t s) {
  if (R == 0) return 0;
  const int choices = V - ((mask_id >= 0 && mask_id < V) ? 1 : 0);
  CVAE_REQUIRE(K >= 1 && K <= choices, "logsoftmax_topk: K=%d of %d entries", K, choices);
  hipLaunchKernelGGL(logsoftmax_topk_kernel, dim3(R), dim3(256), 0, s, V, K, logits, ld, mask_id, lv, li);
  CVAE_LAUNCH_CHECK("logsoftmax_topk_kernel");
  return 0;
}

// One beam step's selection (decoder.py:148-166).  Rows are beam-major: row k*B + b is beam k of sentence b.
// Candidate c = k*K + l of sentence b extends source beam k (k < ns) by its l-th best token and costs
// nll_prev[k][b] - lv[k*B+b][l]; the reference sorts the ns*K candidates of a sentence with Python's
// stable sort on that cost and keeps the first K, i.e. K times the cheapest remaining candidate, lowest
// c on ties.  One thread per sentence (ns*K <= 64).  Writes the new cost, the source beam (for the state
// gather), the token (next input and history) and the back pointer of step t.
__global__ void beam_merge_kernel(int B, int K, int ns, int t, const float* __restrict__ lv,
                                  const int* __restrict__ li, const float* __restrict__ nll_prev,
                                  float* __restrict__ nll_new, int* __restrict__ src, int64_t* __restrict__ cur,
                                  int* __restrict__ bp, int* __restrict__ tok) {
  IDX1D(B);
  const int b = (int)idx;
  unsigned long long taken = 0ull;
  const int nc = ns * K;
  for (int j = 0; j < K; ++j) {
    float bv = INFINITY;
    int bc = -1;
    for (int c = 0; c < nc; ++c) {
      if ((taken >> c) & 1ull) continue;
      const int k = c / K, l = c - k * K;
      const float v = nll_prev[k * B + b] - lv[(long)(k * B + b) * K + l];
      if (bc < 0 || v < bv) { bv = v; bc = c; }
    }
    taken |= 1ull << bc;
    const int k = bc / K, l = bc - k * K;
    const int id = li[(long)(k * B + b) * K + l];
    const int o = j * B + b;
    nll_new[o] = bv; src[o] = k; cur[o] = id;
    bp[((long)t * K) * B + o] = k; tok[((long)t * K) * B + o] = id;
  }
}
int launch_beam_merge(int B, int K, int ns, int t, const float* lv, const int* li, const float* nll_prev,
                      float* nll_new, int* src, int64_t* cur, int* bp, int* tok, hipStream_t s) {
  CVAE_REQUIRE(B > 0 && K >= 1 && ns >= 1 && ns * K <= 64, "beam_merge: ns=%d K=%d", ns, K);
  hipLaunchKernelGGL(beam_merge_kernel, GRID1D((long)B), 0, s, B, K, ns, t, lv, li, nll_prev, nll_new, src, cur, bp,
                     tok);
  CVAE_LAUNCH_CHECK("beam_merge_kernel");
  return 0;
}

// state of the new beams: out[k*B + b] = in[src[k*B + b]*B + b]   (decoder.py:163)
__global__ void beam_gather_kernel(int B, int K, int H, const int* __restrict__ src, const float* __restrict__ in,
                                   float* __restrict__ out) {
  IDX1D((long)K * B * H);
  const int j = (int)(idx % H);
  const long o = idx / H;
  const int b = (int)(o % B);
  out[idx] = in[((long)src[o] * B + b) * H + j];
}
int launch_beam_gather(int B, int K, int H, const int* src, const float* in, float* out, hipStream_t s) {
  hipLaunchKernelGGL(beam_gather_kernel, GRID1D((long)K * B * H), 0, s, B, K, H, src, in, out);
  CVAE_LAUNCH_CHECK("beam_gather_kernel");
  return 0;
}

// beam[0].sent (decoder.py:168): follow the back pointers from beam 0 of the last step
__global__ void beam_backtrack_kernel(int B, int K, int T, const int* __restrict__ bp, const int* __restrict__ tok,
                                      const float* __restrict__ nll, int64_t* __restrict__ out,
                                      float* __restrict__ out_nll) {
  IDX1D(B);
  const int b = (int)idx;
  int k = 0;
  for (int t = T - 1; t >= 0; --t) {
    const long o = ((long)t * K + k) * B + b;
    out[(long)b * T + t] = tok[o];
    k = bp[o];
  }
  if (out_nll) out_nll[b] = nll[b];
}
int launch_beam_backtrack(int B, int K, int T, const int* bp, const int* tok, const float* nll, int64_t* out,
                          float* out_nll, hipStream_t s) {
  hipLaunchKernelGGL(beam_backtrack_kernel, GRID1D((long)B), 0, s, B, K, T, bp, tok, nll, out, out_nll);
  CVAE_LAUNCH_CHECK("beam_backtrack_kernel");
  return 0;
}

// Time-major list of the (t, b) pairs with t < za[b] + off, as row ids k = t*B + b in ascending order,
// and its length.  One block; T is walked in slabs of 256 steps with an LDS scan of the per-step counts.
// One block per job: the time-major list of the rows (t, b) with t < za[b] + off, their count, and optionally
// za_out[b] = za[b] + off (the same bound as an array for the backward recurrence).
__global__ __launch_bounds__(256) void live_rows_tm_kernel(LiveRowsBatch batch) {
  const LiveRowsJob& q = batch.j[blockIdx.x];
  const int T = q.T, B = batch.B, off = q.off;
  const int64_t* __restrict__ za = q.za;
  int* __restrict__ rows = q.rows;
  __shared__ int sc[256];
  __shared__ int base;
  const int tid = threadIdx.x;
  if (tid == 0) base = 0;
  if (q.za_out)
    for (int b = tid; b < B; b += 256) q.za_out[b] = za[b] + off;
  __syncthreads();
  for (int t0 = 0; t0 < T; t0 += 256) {
    const int t = t0 + tid;
    int n = 0;
    if (t < T)
      for (int b = 0; b < B; ++b) n += ((int64_t)t < za[b] + off) ? 1 : 0;
    sc[tid] = n;
    __syncthreads();
    for (int d = 1; d < 256; d <<= 1) {
      const int v = (tid >= d) ? sc[tid - d] : 0;
      __syncthreads();
      sc[tid] += v;
      __syncthreads();
    }
    int o = base + sc[tid] - n;
    if (t < T)
      for (int b = 0; b < B; ++b)
        if ((int64_t)t < za[b] + off) rows[o++] = t * B + b;
    __syncthreads();
    if (tid == 255) base += sc[255];
    __syncthreads();
  }
  if (tid == 0) *q.count = base;
}
int launch_live_rows_tm_n(const LiveRowsJob* jobs, int n, int B, hipStream_t s) {
  CVAE_REQUIRE(jobs && n >= 1 && n <= LiveRowsBatch::MAX && B > 0, "live_rows_tm: bad arguments");
  LiveRowsBatch batch = {};
  batch.B = B;
  for (int i = 0; i < n; ++i) {
    CVAE_REQUIRE(jobs[i].T >= 0 && jobs[i].za && jobs[i].rows && jobs[i].count, "live_rows_tm: bad job");
    batch.j[i] = jobs[i];
  }
  hipLaunchKernelGGL(live_rows_tm_kernel, dim3(n), dim3(256), 0, s, batch);
  CVAE_LAUNCH_CHECK("live_rows_tm_kernel");
  return 0;
}
int launch_live_rows_tm(int T, int B, const int64_t* za, int off, int* rows, int* count, hipStream_t s) {
  const LiveRowsJob j = {T, off, za, rows, count, nullptr};
  return launch_live_rows_tm_n(&j, 1, B, s);
}

__global__ void add_i64_kernel(const int64_t* __restrict__ x, int64_t c, int n, int64_t* __restrict__ y) {
  IDX1D(n);
  y[idx] = x[idx] + c;
}
int launch_add_i64(const int64_t* x, int64_t c, int n, int64_t* y, hipStream_t s) {
  if (n <= 0) return 0;
  hipLaunchKernelGGL(add_i64_kernel, GRID1D(n), 0, s, x, c, n, y);
  CVAE_LAUNCH_CHECK("add_i64_kernel");
  return 0;
}

// za[r] = max(za[r], 1 + the last step t with a non-zero entry in x[t, r, :]); za zeroed by the caller.
// grid (R, time chunks, tensors).
struct LastNzBatch { static constexpr int MAX = 4; const float* x[MAX]; unsigned long long* za[MAX]; };
__global__ __launch_bounds__(256) void last_nonzero_step_kernel(int T, int R, int H, int tchunk, LastNzBatch batch) {
  __shared__ int smax[4];
  const float* __restrict__ x = batch.x[blockIdx.z];
  const int r = blockIdx.x, tid = threadIdx.x;
  const int t0 = blockIdx.y * tchunk, t1 = min(T, t0 + tchunk);
  int best = 0;
  for (int t = t0; t < t1; ++t) {
    const float* row = x + ((long)t * R + r) * H;
    bool nz = false;
    for (int h = tid; h < H; h += 256) nz |= (row[h] != 0.f);
    if (nz) best = t + 1;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) best = max(best, __shfl_down(best, off, 64));
  if ((tid & 63) == 0) smax[tid >> 6] = best;
  __syncthreads();
  if (tid == 0) {
    const int m = max(max(smax[0], smax[1]), max(smax[2], smax[3]));
    if (m > 0) atomicMax(batch.za[blockIdx.z] + r, (unsigned long long)m);
  }
}
// n tensors [T, R, H] in one launch; every za[i] (int64 [R]) must already be zero
int launch_last_nonzero_step_n(int n, int T, int R, int H, const float* const* x, int64_t* const* za, hipStream_t s) {
  CVAE_REQUIRE(n >= 1 && n <= LastNzBatch::MAX && T > 0 && R > 0 && H > 0 && x && za, "last_nonzero_step: bad arguments");
  LastNzBatch batch = {};
  for (int i = 0; i < n; ++i) {
    CVAE_REQUIRE(x[i] && za[i], "last_nonzero_step: null operand");
    batch.x[i] = x[i]; batch.za[i] = reinterpret_cast<unsigned long long*>(za[i]);
  }
  const int tchunk = cdiv(T, 32);
  hipLaunchKernelGGL(last_nonzero_step_kernel, dim3(R, cdiv(T, tchunk), n), dim3(256), 0, s, T, R, H, tchunk, batch);
  CVAE_LAUNCH_CHECK("last_nonzero_step_kernel");
  return 0;
}
// za (int64 [R]) is zeroed here
int launch_last_nonzero_step(int T, int R, int H, const float* x, int64_t* za, hipStream_t s) {
  CVAE_REQUIRE(T > 0 && R > 0 && H > 0 && x && za, "last_nonzero_step: bad arguments");
  CVAE_TRY(launch_zero(reinterpret_cast<float*>(za), 2L * R, s));
  return launch_last_nonzero_step_n(1, T, R, H, &x, &za, s);
}

// logits[b, t, v] = tmp[t*B + b, v]
__global__ void logits_bt_kernel(int T, int B, long V, const float* __restrict__ tmp, float* __restrict__ out) {
  IDX1D((long)T * B * V);
  const long v = idx % V;
  const int t = (int)((idx / V) % T), b = (int)(idx / (V * T));
  out[idx] = tmp[((long)t * B + b) * V + v];
}
int launch_logits_bt(int T, int B, long V, const float* tmp, float* out, hipStream_t s) {
  if ((long)T * B * V == 0) return 0;
  hipLaunchKernelGGL(logits_bt_kernel, GRID1D((long)T * B * V), 0, s, T, B, V, tmp, out);
  CVAE_LAUNCH_CHECK("logits_bt_kernel");
  return 0;
}

// ------------------------------------------------------------------ latent-space kernels
__global__ void reparam_fwd_kernel(int n, int B, int Z, const float* __restrict__ mu, const float* __restrict__ sigma,
                                   const float* __restrict__ eps, float* __restrict__ z) {
  IDX1D((long)n * B * Z);
  const long bz = idx % ((long)B * Z);
  z[idx] = mu[bz] + sigma[bz] * eps[idx];
}
int launch_reparam_fwd(const float* mu, const float* sigma, const float* eps, int n, int B, int Z, float* z,
                       hipStream_t s) {
  if ((long)n * B * Z == 0) return 0;
  hipLaunchKernelGGL(reparam_fwd_kernel, GRID1D((long)n * B * Z), 0, s, n, B, Z, mu, sigma, eps, z);
  CVAE_LAUNCH_CHECK("reparam_fwd_kernel");
  return 0;
}

__global__ void reparam_bwd_kernel(long n, const float* __restrict__ dz, const float* __restrict__ eps,
                                   float* __restrict__ dmu, float* __restrict__ dsigma) {
  IDX1D(n);
  dmu[idx] += dz[idx];
  dsigma[idx] += dz[idx] * eps[idx];
}
int launch_reparam_bwd(const float* dz, const float* eps, int B, int Z, float* dmu, float* dsigma, hipStream_t s) {
  if ((long)B * Z == 0) return 0;
  hipLaunchKernelGGL(reparam_bwd_kernel, GRID1D((long)B * Z), 0, s, (long)B * Z, dz, eps, dmu, dsigma);
  CVAE_LAUNCH_CHECK("reparam_bwd_kernel");
  return 0;
}

// one wave per sentence; lanes stride over Z
__global__ __launch_bounds__(256) void kl_fwd_kernel(int B, int Z, const float* __restrict__ mu,
                                                     const float* __restrict__ sigma, const float* __restrict__ mu_p,
                                                     const float* __restrict__ sigma_p, float* __restrict__ kl) {
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (b >= B) return;
  float acc = 0.f;
  for (int j = lane; j < Z; j += 64) {
    const float m = mu[(long)b * Z + j], sg = sigma[(long)b * Z + j];
    if (mu_p) {
      const float mp = mu_p[(long)b * Z + j], sp = sigma_p[(long)b * Z + j];
      const float vr = (sg / sp) * (sg / sp), t1 = ((m - mp) / sp) * ((m - mp) / sp);
      acc += 0.5f * (vr + t1 - 1.f - logf(vr));
    } else {
      const float vr = sg * sg;
      acc += 0.5f * (vr + m * m - 1.f - logf(vr));
    }
  }
  acc = wave_sum(acc);
  if (lane == 0) kl[b] = acc;
}
int launch_kl_fwd(const float* mu, const float* sigma, const float* mu_p, const float* sigma_p, int B, int Z,
                  float* kl, hipStream_t s) {
  if (B == 0) return 0;
  CVAE_REQUIRE((mu_p == nullptr) == (sigma_p == nullptr), "kl: prior needs both mu_p and sigma_p");
  hipLaunchKernelGGL(kl_fwd_kernel, dim3(cdiv(B, 4)), dim3(256), 0, s, B, Z, mu, sigma, mu_p, sigma_p, kl);
  CVAE_LAUNCH_CHECK("kl_fwd_kernel");
  return 0;
}

__global__ void kl_bwd_kernel(int B, int Z, const float* __restrict__ mu, const float* __restrict__ sigma,
                              const float* __restrict__ mu_p, const float* __restrict__ sigma_p,
                              const float* __restrict__ gkl, float* __restrict__ dmu, float* __restrict__ dsigma,
                              float* __restrict__ dmu_p, float* __restrict__ dsigma_p) {
  IDX1D((long)B * Z);
  const float g = gkl[idx / Z];
  const float m = mu[idx], sg = sigma[idx];
  if (mu_p) {
    const float mp = mu_p[idx], sp = sigma_p[idx];
    const float sp2 = sp * sp, d = m - mp;
    dmu[idx] = g * d / sp2;
    dsigma[idx] = g * (sg / sp2 - 1.f / sg);
    if (dmu_p) dmu_p[idx] = -g * d / sp2;
    if (dsigma_p) dsigma_p[idx] = g * (-(sg * sg) / (sp2 * sp) - d * d / (sp2 * sp) + 1.f / sp);
  } else {
    dmu[idx] = g * m;
    dsigma[idx] = g * (sg - 1.f / sg);
  }
}
int launch_kl_bwd(const float* mu, const float* sigma, const float* mu_p, const float* sigma_p, const float* gkl,
                  int B, int Z, float* dmu, float* dsigma, float* dmu_p, float* dsigma_p, hipStream_t s) {
  if ((long)B * Z == 0) return 0;
  hipLaunchKernelGGL(kl_bwd_kernel, GRID1D((long)B * Z), 0, s, B, Z, mu, sigma, mu_p, sigma_p, gkl, dmu, dsigma,
                     dmu_p, dsigma_p);
  CVAE_LAUNCH_CHECK("kl_bwd_kernel");
  return 0;
}

// Free bits (surprising-fix*, CV/Experiment.py:1137-1142): each latent dimension's KL to N(0,1) is
// floored at `floor` before the sum; torch.clamp passes the gradient where kl_e >= floor.
__device__ __forceinline__ float kl_std_elem(float m, float sg) { return 0.5f * (sg * sg + m * m - 1.f) - logf(sg); }
__global__ __launch_bounds__(256) void kl_fb_fwd_kernel(int B, int Z, float floor, const float* __restrict__ mu,
                                                        const float* __restrict__ sigma, float* __restrict__ kl) {
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (b >= B) return;
  float acc = 0.f;
  for (int j = lane; j < Z; j += 64) acc += fmaxf(kl_std_elem(mu[(long)b * Z + j], sigma[(long)b * Z + j]), floor);
  acc = wave_sum(acc);
  if (lane == 0) kl[b] = acc;
}
__global__ void kl_fb_bwd_kernel(int B, int Z, float floor, const float* __restrict__ mu,
                                 const float* __restrict__ sigma, const float* __restrict__ gkl,
                                 float* __restrict__ dmu, float* __restrict__ dsigma) {
  IDX1D((long)B * Z);
  const float m = mu[idx], sg = sigma[idx];
  const float g = (kl_std_elem(m, sg) >= floor) ? gkl[idx / Z] : 0.f;
  dmu[idx] = g * m;
  dsigma[idx] = g * (sg - 1.f / sg);
}
int launch_kl_fb_fwd(const float* mu, const float* sigma, int B, int Z, float floor, float* kl, hipStream_t s) {
  if (B == 0) return 0;
  hipLaunchKernelGGL(kl_fb_fwd_kernel, dim3(cdiv(B, 4)), dim3(256), 0, s, B, Z, floor, mu, sigma, kl);
  CVAE_LAUNCH_CHECK("kl_fb_fwd_kernel");
  return 0;
}
int launch_kl_fb_bwd(const float* mu, const float* sigma, const float* gkl, int B, int Z, float floor, float* dmu,
                     float* dsigma, hipStream_t s) {
  if ((long)B * Z == 0) return 0;
  hipLaunchKernelGGL(kl_fb_bwd_kernel, GRID1D((long)B * Z), 0, s, B, Z, floor, mu, sigma, gkl, dmu, dsigma);
  CVAE_LAUNCH_CHECK("kl_fb_bwd_kernel");
  return 0;
}

__constant__ float kImqScales[7] = {.1f, .2f, .5f, 1.f, 2.f, 5.f, 10.f};

__global__ __launch_bounds__(256) void couple_fwd_kernel(int B, int E, int Z, int euclid,
                                                         const float* __restrict__ a, const float* __restrict__ c,
                                                         float* __restrict__ d, float* __restrict__ loss) {
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (b >= B) return;
  float acc = 0.f;
  for (int j = lane; j < E; j += 64) {
    const float t = a[(long)b * E + j] - c[(long)b * E + j];
    acc += t * t;
  }
  acc = wave_sum(acc);
  if (lane == 0) {
    d[b] = acc;
    if (euclid) {
      loss[b] = acc;
    } else {
      float imq = 0.f;
      for (int k = 0; k < 7; ++k) {
        const float C = 2.f * Z * kImqScales[k];
        imq += C / (C + acc);
      }
      loss[b] = -imq;
    }
  }
}
int launch_couple_fwd(const float* a, const float* c, int B, int E, int Z, int euclid, float* d, float* loss,
                      hipStream_t s) {
  if (B == 0) return 0;
  hipLaunchKernelGGL(couple_fwd_kernel, dim3(cdiv(B, 4)), dim3(256), 0, s, B, E, Z, euclid, a, c, d, loss);
  CVAE_LAUNCH_CHECK("couple_fwd_kernel");
  return 0;
}

__global__ void couple_bwd_kernel(int B, int E, int Z, int euclid, const float* __restrict__ a,
                                  const float* __restrict__ c, const float* __restrict__ d,
                                  const float* __restrict__ g, float* __restrict__ da) {
  IDX1D((long)B * E);
  const int b = (int)(idx / E);
  float dl_dd = 1.f;
  if (!euclid) {
    dl_dd = 0.f;
    const float dd = d[b];
    for (int k = 0; k < 7; ++k) {
      const float C = 2.f * Z * kImqScales[k];
      dl_dd += C / ((C + dd) * (C + dd));
    }
  }
  da[idx] = g[b] * dl_dd * 2.f * (a[idx] - c[idx]);
}
int launch_couple_bwd(const float* a, const float* c, const float* d, const float* g, int B, int E, int Z,
                      int euclid, float* da, hipStream_t s) {
  if ((long)B * E == 0) return 0;
  hipLaunchKernelGGL(couple_bwd_kernel, GRID1D((long)B * E), 0, s, B, E, Z, euclid, a, c, d, g, da);
  CVAE_LAUNCH_CHECK("couple_bwd_kernel");
  return 0;
}

// ------------------------------------------------------------------ IW estimator (a13/a14)
// One block per sentence b.  Pass 1: log-weights w_k = log p(z) - xent - log q(z|x) for all K
// samples (kept in LDS), running sums for kl / xent.  Pass 2: max-shifted log-sum-exp.
constexpr float kHalfLog2Pi = 0.91893853320467274178f;

__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// kl_k = log q(z_k|x) - log p(z_k) is a difference of two sums of Z terms of size O(1) each (O(50) in all) that
// leaves O(1): formed in fp32 it loses ~2e-6 absolute per sample, which is 5e-4 of a small KL.  Here every
// per-dimension term and the sums over dimensions and samples are formed in double from the fp32 inputs (the
// log sigma terms do not depend on the sample and are summed once per sentence), and rounded to fp32 once: the
// result is the exactly-rounded value of the reference's expression, 1e-4 of the reference's own fp32 evaluation.
__global__ __launch_bounds__(256) void iw_reduce_kernel(int K, int B, int Z, const float* __restrict__ z,
                                                        const float* __restrict__ xent, const float* __restrict__ mu,
                                                        const float* __restrict__ sigma,
                                                        const float* __restrict__ mu_p,
                                                        const float* __restrict__ sigma_p, float* __restrict__ nll,
                                                        float* __restrict__ kl, float* __restrict__ xent_mean) {
  extern __shared__ float w[];          // [K] log-weights, then block-reduction space (doubles, 8-byte aligned)
  double* red = reinterpret_cast<double*>(w + ((K + 1) & ~1));
  __shared__ double s_logs;             // sum_j (log sigma_p[j] - log sigma[j]): the sample-independent part of kl_k
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (wave == 0) {
    double a = 0.0;
    for (int j = lane; j < Z; j += 64) {
      a -= log((double)sigma[(long)b * Z + j]);
      if (mu_p) a += log((double)sigma_p[(long)b * Z + j]);
    }
    a = wave_sum_f64(a);
    if (lane == 0) s_logs = a;
  }
  __syncthreads();
  const double logs = s_logs;
  double kl_acc = 0.0, x_acc = 0.0;
  float wmax = -INFINITY;
  for (int k = tid; k < K; k += blockDim.x) {
    const float* zk = z + ((long)k * B + b) * Z;
    double d = logs;                    // log q - log p of this sample
    for (int j = 0; j < Z; ++j) {
      const double zv = zk[j], m = mu[(long)b * Z + j], sg = sigma[(long)b * Z + j];
      d -= ((zv - m) * (zv - m)) / (2.0 * sg * sg);
      if (mu_p) {
        const double mp = mu_p[(long)b * Z + j], sp = sigma_p[(long)b * Z + j];
        d += ((zv - mp) * (zv - mp)) / (2.0 * sp * sp);
      } else {
        d += zv * zv * 0.5;
      }
    }
    const float xe = xent[(long)k * B + b];
    const float wk = (float)(-d - (double)xe);        // log p(z) + log p(x|z) - log q(z|x)
    w[k] = wk;
    wmax = fmaxf(wmax, wk);
    kl_acc += d;
    x_acc += (double)xe;
  }
  kl_acc = wave_sum_f64(kl_acc); x_acc = wave_sum_f64(x_acc); wmax = wave_max(wmax);
  if (lane == 0) { red[wave] = kl_acc; red[4 + wave] = x_acc; red[8 + wave] = (double)wmax; }
  __syncthreads();
  const int nw = blockDim.x >> 6;
  float M = -INFINITY;
  double klt = 0.0, xt = 0.0;
  for (int i = 0; i < nw; ++i) { klt += red[i]; xt += red[4 + i]; M = fmaxf(M, (float)red[8 + i]); }
  __syncthreads();
  float sacc = 0.f;
  for (int k = tid; k < K; k += blockDim.x) sacc += expf(w[k] - M);
  sacc = wave_sum(sacc);
  if (lane == 0) red[wave] = (double)sacc;
  __syncthreads();
  if (tid == 0) {
    float st = 0.f;
    for (int i = 0; i < nw; ++i) st += (float)red[i];
    nll[b] = -(M + logf(st) - logf((float)K));
    kl[b] = (float)(klt / K);
    xent_mean[b] = (float)(xt / K);
  }
}
int launch_iw_reduce(const float* z, const float* xent, const float* mu, const float* sigma, const float* mu_p,
                     const float* sigma_p, int K, int B, int Z, float* nll, float* kl, float* xent_mean,
                     hipStream_t s) {
  CVAE_REQUIRE(K > 0 && B >= 0 && Z > 0, "iw_reduce: bad shape");
  CVAE_REQUIRE(K <= 12000, "iw_reduce: K=%d exceeds the LDS-resident limit (12000)", K);
  CVAE_REQUIRE((mu_p == nullptr) == (sigma_p == nullptr), "iw_reduce: prior needs both mu_p and sigma_p");
  if (B == 0) return 0;
  ProfScope prof(PK_IW_REDUCE, s, 4.0 * ((double)K * B * Z + (double)K * B + 2.0 * B * Z * (mu_p ? 2 : 1) + 3.0 * B));
  hipLaunchKernelGGL(iw_reduce_kernel, dim3(B), dim3(256), (((K + 1) & ~1) + 24) * sizeof(float), s, K, B, Z, z, xent, mu, sigma,
                     mu_p, sigma_p, nll, kl, xent_mean);
  CVAE_LAUNCH_CHECK("iw_reduce_kernel");
  return 0;
}

// ------------------------------------------------------------------ MI (a15)
// Block (k, i): sample z[k,i,:].  Threads loop over j: log q(z | x_j); online LSE over j != i.
// out_lqzx[k,i] = log q(z|x_i), out_lqz[k,i] = LSE_{j!=i} - log(B-1).
__global__ __launch_bounds__(256) void mi_pair_kernel(int K, int B, int Z, const float* __restrict__ z,
                                                      const float* __restrict__ mu, const float* __restrict__ sigma,
                                                      float* __restrict__ lqzx, float* __restrict__ lqz) {
  extern __shared__ float sh[];         // z row [Z] + reduction space [8]
  float* zrow = sh;
  float* red = sh + Z;
  const int i = blockIdx.x, k = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int j = tid; j < Z; j += blockDim.x) zrow[j] = z[((long)k * B + i) * Z + j];
  __syncthreads();
  float m = -INFINITY, sacc = 0.f;
  for (int j = tid; j < B; j += blockDim.x) {
    float lp = 0.f;
    for (int d = 0; d < Z; ++d) {
      const float mm = mu[(long)j * Z + d], sg = sigma[(long)j * Z + d], t = zrow[d] - mm;
      lp += -(t * t) / (2.f * sg * sg) - logf(sg) - kHalfLog2Pi;
    }
    if (j == i) { lqzx[(long)k * B + i] = lp; continue; }
    const float M = fmaxf(m, lp);
    sacc = sacc * ((m == M) ? 1.f : expf(m - M)) + expf(lp - M);
    m = M;
  }
  // block-wide (m, s) merge
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float m2 = __shfl_xor(m, o, 64), s2 = __shfl_xor(sacc, o, 64);
    const float M = fmaxf(m, m2);
    sacc = sacc * ((m == M) ? 1.f : expf(m - M)) + s2 * ((m2 == M) ? 1.f : expf(m2 - M));
    m = M;
  }
  if (lane == 0) { red[wave] = m; red[4 + wave] = sacc; }
  __syncthreads();
  if (tid == 0) {
    float Mx = -INFINITY, st = 0.f;
    const int nw = blockDim.x >> 6;
    for (int w2 = 0; w2 < nw; ++w2) Mx = fmaxf(Mx, red[w2]);
    for (int w2 = 0; w2 < nw; ++w2) st += red[4 + w2] * ((red[w2] == Mx) ? 1.f : expf(red[w2] - Mx));
    lqz[(long)k * B + i] = Mx + logf(st) - logf((float)(B - 1));
  }
}

// mi[i] = mean_k lqzx[k,i] - mean_k lqz[k,i]
__global__ void mi_mean_kernel(int K, int B, const float* __restrict__ lqzx, const float* __restrict__ lqz,
                               float* __restrict__ mi) {
  IDX1D(B);
  float a = 0.f, c = 0.f;
  for (int k = 0; k < K; ++k) { a += lqzx[(long)k * B + idx]; c += lqz[(long)k * B + idx]; }
  mi[idx] = a / K - c / K;
}

int launch_mi(const float* z, const float* mu, const float* sigma, int K, int B, int Z, float* lqzx, float* lqz,
              float* mi, hipStream_t s) {
  CVAE_REQUIRE(K > 0 && B > 1 && Z > 0 && K <= 65535, "mi: bad shape K=%d B=%d Z=%d", K, B, Z);
  hipLaunchKernelGGL(mi_pair_kernel, dim3(B, K), dim3(256), (Z + 8) * sizeof(float), s, K, B, Z, z, mu, sigma, lqzx,
                     lqz);
  CVAE_LAUNCH_CHECK("mi_pair_kernel");
  hipLaunchKernelGGL(mi_mean_kernel, GRID1D(B), 0, s, K, B, lqzx, lqz, mi);
  CVAE_LAUNCH_CHECK("mi_mean_kernel");
  return 0;
}

// ---- the same estimator for callers that cannot allocate (HIP graph capture, serving loops): the two
// [K, B] intermediates live in a file-scope device array instead of a caller workspace.  The array's name
// is only a host-side handle: its device address is looked up once per device (mi_static_init, called by
// cvae_init with that device current) and that pointer is what the kernels receive.
constexpr long MI_STATIC_MAX_KB = 131072;                 // K*B limit: 2 x 131072 floats = 1 MiB per GPU
constexpr int MI_STATIC_MAX_DEVICES = 64;
static __device__ __align__(256) float scratch[2 * MI_STATIC_MAX_KB];
static float* g_mi_scratch[MI_STATIC_MAX_DEVICES] = {};   // per device: address of that device's copy

int mi_static_init(int device) {
  CVAE_REQUIRE(device >= 0 && device < MI_STATIC_MAX_DEVICES, "mi_static_init: device %d out of range", device);
  if (g_mi_scratch[device]) return 0;
  float* ptr = nullptr;
  CVAE_TRY(check_hip(hipGetSymbolAddress(reinterpret_cast<void**>(&ptr), HIP_SYMBOL(scratch)),
                     "mi_static_init: hipGetSymbolAddress(scratch)"));
  CVAE_REQUIRE(ptr != nullptr, "mi_static_init: null device address for the static scratch array");
  g_mi_scratch[device] = ptr;
  return 0;
}

long mi_static_max_rows() { return MI_STATIC_MAX_KB; }

int launch_mi_static(const float* z, const float* mu, const float* sigma, int K, int B, int Z, float* mi,
                     hipStream_t s) {
  CVAE_REQUIRE(K > 0 && B > 1 && Z > 0, "mi_static: bad shape K=%d B=%d Z=%d", K, B, Z);
  CVAE_REQUIRE((long)K * B <= MI_STATIC_MAX_KB,
               "mi_static: K*B = %ld exceeds the static scratch limit of %ld (use cvae_mi with a workspace)",
               (long)K * B, MI_STATIC_MAX_KB);
  int device = -1;
  CVAE_TRY(check_hip(hipGetDevice(&device), "mi_static: hipGetDevice"));
  CVAE_REQUIRE(device >= 0 && device < MI_STATIC_MAX_DEVICES && g_mi_scratch[device] != nullptr,
               "mi_static: cvae_init(%d) has not been called for the current device", device);
  float* ptr = g_mi_scratch[device];
  // every element of [0, K*B) of both halves is written by mi_pair_kernel before mi_mean_kernel reads it:
  // nothing has to start at zero, whatever an earlier (larger) call left behind
  return launch_mi(z, mu, sigma, K, B, Z, ptr, ptr + (long)K * B, mi, s);
}

// ------------------------------------------------------------------ status word <-> device float
// A data-parallel job folds the recurrent kernels' time-out word into its gradient all-reduce: export writes it as a
// float into a spare arena slot before the collective, import raises the local word when the reduced slot is
// non-zero (some rank timed out), so that cvae_clip_adam skips the step on EVERY rank and the replicas stay identical.
__global__ void status_export_kernel(const int* __restrict__ status, float* __restrict__ dst) {
  *dst = (status && __hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)) ? 1.f : 0.f;
}
__global__ void status_import_kernel(const float* __restrict__ src, int* __restrict__ status) {
  if (*src != 0.f) __hip_atomic_store(status, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
int launch_status_export(float* dst, hipStream_t s) {
  CVAE_REQUIRE(dst, "persist_status_export: null destination");
  int* status = nullptr;
  CVAE_TRY(persist_status_word(&status));
  hipLaunchKernelGGL(status_export_kernel, dim3(1), dim3(1), 0, s, status, dst);
  CVAE_LAUNCH_CHECK("status_export_kernel");
  return 0;
}
int launch_status_import(const float* src, hipStream_t s) {
  CVAE_REQUIRE(src, "persist_status_import: null source");
  int* status = nullptr;
  CVAE_TRY(persist_status_word(&status));
  hipLaunchKernelGGL(status_import_kernel, dim3(1), dim3(1), 0, s, src, status);
  CVAE_LAUNCH_CHECK("status_import_kernel");
  return 0;
}

// ------------------------------------------------------------------ clip + Adam (a12)
// out[b] = block b's sum over all grads^2, out[SUMSQ_MAX_BLOCKS + b] = its sum over the alias segment.  No float
// atomics: the clip norm is the same bits on every run (and on every rank of a data-parallel job).
constexpr int SUMSQ_MAX_BLOCKS = 256;
__global__ __launch_bounds__(1024) void sumsq_kernel(long n, long a0, long a1, long skip0, long skip1,
                                                     const float* __restrict__ g, float* __restrict__ out,
                                                     const int* __restrict__ status_in, int* __restrict__ status_copy) {
  __shared__ float red[2][16];
  // one read of the recurrent kernels' timeout word (host-mapped) per optimizer step: clip_adam_kernel
  // reads the device copy and leaves the parameters alone when a kernel of this step timed out
  if (blockIdx.x == 0 && threadIdx.x == 0 && status_copy)
    *status_copy = status_in ? __hip_atomic_load(status_in, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : 0;
  float all = 0.f, al = 0.f;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float v = (i >= skip0 && i < skip1) ? 0.f : g[i];      // a group without a gradient this iteration
    all += v * v;
    if (i >= a0 && i < a1) al += v * v;
  }
  all = wave_sum(all); al = wave_sum(al);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) { red[0][wave] = all; red[1][wave] = al; }
  __syncthreads();
  if (threadIdx.x == 0) {           // one pair of partial sums per block; clip_adam_kernel adds them in a fixed order
    float s0 = 0.f, s1 = 0.f;
#pragma unroll
    for (int w = 0; w < 16; ++w) { s0 += red[0][w]; s1 += red[1][w]; }
    out[blockIdx.x] = s0;
    out[SUMSQ_MAX_BLOCKS + blockIdx.x] = s1;
  }
}

struct AdamSteps { float step_size[8]; float bc2_sqrt[8]; };   // per listing: lr/(1-b1^t), sqrt(1-b2^t)

// [l0, l1): the "late" parameter group (launch_clip_adam): its own step count (late.step_size[0]), or, with
// late_skip, no gradient this iteration: left exactly as it is, as torch's clip and Adam leave a parameter whose
// .grad is None.
__global__ __launch_bounds__(256) void clip_adam_kernel(long n, long a0, long a1, int mult, float max_norm,
                                                        float b1, float b2, float eps, AdamSteps other,
                                                        AdamSteps alias_steps, long l0, long l1, int late_skip,
                                                        AdamSteps late, const float* __restrict__ parts,
                                                        int nparts, float* __restrict__ sumsq,
                                                        float* __restrict__ p, float* __restrict__ g,
                                                        float* __restrict__ m, float* __restrict__ v,
                                                        float* __restrict__ norm_out,
                                                        const int* __restrict__ status_copy) {
  if (status_copy && *status_copy) return;            // invalid gradients (a recurrent kernel timed out): no step
  // the blocks' partial sums, added by every block in the same order (4 per lane, then a fixed butterfly)
  __shared__ float tot[2];
  if (threadIdx.x < 64) {
    float s0 = 0.f, s1 = 0.f;
#pragma unroll
    for (int j = 0; j < SUMSQ_MAX_BLOCKS / 64; ++j) {
      const int b = threadIdx.x + 64 * j;
      if (b < nparts) { s0 += parts[b]; s1 += parts[SUMSQ_MAX_BLOCKS + b]; }
    }
    s0 = wave_sum(s0); s1 = wave_sum(s1);
    if (threadIdx.x == 0) {
      tot[0] = s0; tot[1] = s1;
      if (blockIdx.x == 0) { sumsq[0] = s0; sumsq[1] = s1; }
    }
  }
  __syncthreads();
  // total norm as clip_grad_norm_ sees it: the aliased segment is listed `mult` times
  const float total = sqrtf(tot[0] + (float)(mult - 1) * tot[1]);
  float coef = max_norm / (total + 1e-6f);
  coef = coef > 1.f ? 1.f : coef;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const bool alias = (i >= a0 && i < a1);
    const bool is_late = (i >= l0 && i < l1);
    if (is_late && late_skip) continue;
    const int reps = alias ? mult : 1;
    float gv = g[i];
    for (int r = 0; r < reps; ++r) gv *= coef;        // scaled once per listing
    g[i] = gv;
    float pv = p[i], mv = m[i], vv = v[i];
    for (int r = 0; r < reps; ++r) {                  // one full Adam step per listing
      mv = mv + (gv - mv) * (1.f - b1);               // lerp_
      vv = vv * b2 + gv * gv * (1.f - b2);
      const float ss = alias ? alias_steps.step_size[r] : (is_late ? late.step_size[0] : other.step_size[0]);
      const float bs = alias ? alias_steps.bc2_sqrt[r] : (is_late ? late.bc2_sqrt[0] : other.bc2_sqrt[0]);
      const float denom = sqrtf(vv) / bs + eps;
      pv = pv - ss * (mv / denom);
    }
    p[i] = pv; m[i] = mv; v[i] = vv;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0 && norm_out) *norm_out = total;
}

int launch_clip_adam(float* p, float* g, float* m, float* v, long n, long alias_off, long alias_n, int mult,
                     float max_norm, float lr, float b1, float b2, float eps, long step_other, long step_alias,
                     long late_off, long late_n, long step_late, float* sumsq, float* norm_out, hipStream_t s) {
  // arguments first: nothing is acquired or launched for a call that is refused
  CVAE_REQUIRE(n >= 0 && alias_off >= 0 && alias_n >= 0 && alias_off + alias_n <= n && mult >= 1 && mult <= 8,
               "clip_adam: bad segment (alias_mult must be 1..8)");
  CVAE_REQUIRE(late_off >= 0 && late_n >= 0 && late_off + late_n <= n &&
               (late_n == 0 || alias_n == 0 || late_off >= alias_off + alias_n || late_off + late_n <= alias_off),
               "clip_adam: the late segment must lie inside the arena and outside the alias segment");
  if (n == 0) return 0;
  CVAE_REQUIRE(p && g && m && v && sumsq, "clip_adam: null operand");
  // timeout word of the whole-sequence recurrent kernels and a device-side copy of it (a spare word of this
  // stream's sync buffer)
  int* status_in = nullptr;
  CVAE_TRY(persist_status_word(&status_in));
  void* syncbuf = nullptr;
  CVAE_TRY(lib_scratch(LS_PERSIST_SYNC, s, persist_sync_bytes(), &syncbuf, 1));
  int* status_copy = persist_status_copy_word(syncbuf);
  void* partbuf = nullptr;
  CVAE_TRY(lib_scratch(LS_SUMSQ, s, 2 * SUMSQ_MAX_BLOCKS * sizeof(float), &partbuf));
  float* parts = static_cast<float*>(partbuf);
  int grid = cdiv(n, 256 * 8);
  if (grid > 2048) grid = 2048;
  if (grid < 1) grid = 1;
  int sgrid = cdiv(n, 1024 * 8);
  if (sgrid > SUMSQ_MAX_BLOCKS) sgrid = SUMSQ_MAX_BLOCKS;
  if (sgrid < 1) sgrid = 1;
  const bool late_skip = late_n > 0 && step_late < 0;
  hipLaunchKernelGGL(sumsq_kernel, dim3(sgrid), dim3(1024), 0, s, n, alias_off, alias_off + alias_n,
                     late_skip ? late_off : 0L, late_skip ? late_off + late_n : 0L, g, parts, status_in, status_copy);
  CVAE_LAUNCH_CHECK("sumsq_kernel");
  AdamSteps so = {}, sa = {}, sl = {};
  if (late_n > 0 && step_late >= 0) {
    sl.step_size[0] = (float)((double)lr / (1.0 - pow((double)b1, (double)(step_late + 1))));
    sl.bc2_sqrt[0] = (float)sqrt(1.0 - pow((double)b2, (double)(step_late + 1)));
  }
  so.step_size[0] = (float)((double)lr / (1.0 - pow((double)b1, (double)(step_other + 1))));
  so.bc2_sqrt[0] = (float)sqrt(1.0 - pow((double)b2, (double)(step_other + 1)));
  for (int r = 0; r < mult; ++r) {
    sa.step_size[r] = (float)((double)lr / (1.0 - pow((double)b1, (double)(step_alias + 1 + r))));
    sa.bc2_sqrt[r] = (float)sqrt(1.0 - pow((double)b2, (double)(step_alias + 1 + r)));
  }
  ProfScope prof(PK_CLIP_ADAM, s, 32.0 * n);              // algorithmic bytes: grads, params, m, v each read and written once
  hipLaunchKernelGGL(clip_adam_kernel, dim3(grid), dim3(256), 0, s, n, alias_off, alias_off + alias_n, mult,
                     max_norm, b1, b2, eps, so, sa, late_off, late_off + late_n, late_skip ? 1 : 0,
                     sl, parts, sgrid, sumsq, p, g, m, v, norm_out, status_copy);
  CVAE_LAUNCH_CHECK("clip_adam_kernel");
  return 0;
}

}  // namespace cvae
