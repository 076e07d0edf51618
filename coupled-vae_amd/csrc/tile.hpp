// LDS tile staging shared by the MFMA kernels (gemm.hip, projce.hip).
#pragma once
#include "common.hpp"

namespace cvae {

constexpr int BK = 32;


// Stage one (ROWS x BK) operand tile into LDS as Ts[k][row].
//  KC = true : operand stored [row][k]  (k contiguous)  -> transposing store, LD = ROWS+1
//  KC = false: operand stored [k][row]  (row contiguous) -> straight store,   LD = ROWS+4
template <int ROWS, bool KC>
struct TileLoader {
  static constexpr int LD = KC ? ROWS + 1 : ROWS + 4;
  static constexpr int NV = ROWS * BK / 4 / 256;   // float4 per thread
  float4 v[NV];

  __device__ __forceinline__ void load(const float* __restrict__ P, long ld, int row0, int nrows,
                                       int k0, int K, int vec, int tid) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
      if (KC) {
        const int kq = tid & 7, rr = (tid >> 3) + 32 * i;
        const int row = row0 + rr, k = k0 + 4 * kq;
        if (row < nrows) {
          const float* src = P + (long)row * ld + k;
          if (vec && k + 3 < K) {
            r = *reinterpret_cast<const float4*>(src);
          } else {
            if (k + 0 < K) r.x = src[0];
            if (k + 1 < K) r.y = src[1];
            if (k + 2 < K) r.z = src[2];
            if (k + 3 < K) r.w = src[3];
          }
        }
      } else {
        constexpr int C4 = ROWS / 4;              // float4 columns per k row
        const int c4 = tid % C4, kk = tid / C4 + (256 / C4) * i;
        const int row = row0 + 4 * c4, k = k0 + kk;
        if (k < K) {
          const float* src = P + (long)k * ld + row;
          if (vec && row + 3 < nrows) {
            r = *reinterpret_cast<const float4*>(src);
          } else {
            if (row + 0 < nrows) r.x = src[0];
            if (row + 1 < nrows) r.y = src[1];
            if (row + 2 < nrows) r.z = src[2];
            if (row + 3 < nrows) r.w = src[3];
          }
        }
      }
      v[i] = r;
    }
  }

  __device__ __forceinline__ void store(float* Ts, int tid) const {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      if (KC) {
        const int kq = tid & 7, rr = (tid >> 3) + 32 * i;
        float* d = Ts + (4 * kq) * LD + rr;
        d[0 * LD] = v[i].x; d[1 * LD] = v[i].y; d[2 * LD] = v[i].z; d[3 * LD] = v[i].w;
      } else {
        constexpr int C4 = ROWS / 4;
        const int c4 = tid % C4, kk = tid / C4 + (256 / C4) * i;
        *reinterpret_cast<float4*>(Ts + kk * LD + 4 * c4) = v[i];
      }
    }
  }
};

}  // namespace cvae
