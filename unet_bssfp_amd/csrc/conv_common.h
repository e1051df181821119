// Device-side argument block and shared epilogue of the implicit-GEMM convolution kernels.
#pragma once
#include "common.h"

struct ConvArgs {
  const char* x0; const char* x1;
  int c0, c1, ld0, ld1;
  int n, di, hi, wi;
  int do_, ho, wo;
  int ks, stride;
  int pd, ph, pw;
  const char* wp; int coutp;
  const float* bias;
  char* y; int ldy, cstore;
  int dy, hy, wy, os, od, oh, ow;
  float* stats;
  int tiles_d, tiles_h, tiles_w;   // halo kernel
  int nchunks;
  long long m_total;               // gather kernel: n*do*ho*wo
};

// Epilogue shared by both kernels.
//   acc[vt][ct][i]  : row = acc_row(i,h) of subtile vt, col = lane&31 of cout tile ct
//   yoff[vt]        : this lane's (row = lane&31) output ELEMENT offset of voxel `row`, or -1
// Stores z = acc + bias, and (optionally) per-WG channel sums of acc / acc^2 over valid rows.
template <typename T, int VT, int CT>
__device__ __forceinline__ void conv_epilogue(const ConvArgs& a, f32x16 (&acc)[VT][CT],
                                              const long long (&yoff)[VT], int co_base,
                                              int tile_index, float* red /* LDS, >= 4*CT*64 floats */) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, h = lane >> 5;
  float s1[CT], s2[CT];
#pragma unroll
  for (int ct = 0; ct < CT; ++ct) { s1[ct] = 0.f; s2[ct] = 0.f; }

#pragma unroll
  for (int vt = 0; vt < VT; ++vt) {
    const int lo = (int)(yoff[vt] & 0xffffffffll), hi = (int)(yoff[vt] >> 32);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int row = acc_row(i, h);
      const int olo = __shfl(lo, row, 64), ohi = __shfl(hi, row, 64);
      const long long off = ((long long)ohi << 32) | (unsigned int)olo;
      if (off >= 0) {
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
          const int co = co_base + ct * 32 + r;
          const float v = acc[vt][ct][i];
          s1[ct] += v;
          s2[ct] += v * v;
          if (co < a.cstore) {
            const float b = a.bias ? a.bias[co] : 0.f;
            Elem<T>::store(reinterpret_cast<T*>(a.y) + off + co, v + b);
          }
        }
      }
    }
  }
  if (a.stats) {
    // combine lane halves, then the 4 waves through LDS in a fixed order (deterministic)
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
      s1[ct] += __shfl_xor(s1[ct], 32, 64);
      s2[ct] += __shfl_xor(s2[ct], 32, 64);
    }
    __syncthreads();  // all waves are done reading the staging LDS
    if (h == 0) {
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) {
        red[((wave * CT + ct) * 2 + 0) * 32 + r] = s1[ct];
        red[((wave * CT + ct) * 2 + 1) * 32 + r] = s2[ct];
      }
    }
    __syncthreads();
    if (wave == 0 && h == 0) {
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) {
        float t1 = 0.f, t2 = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
          t1 += red[((w * CT + ct) * 2 + 0) * 32 + r];
          t2 += red[((w * CT + ct) * 2 + 1) * 32 + r];
        }
        const int co = co_base + ct * 32 + r;
        float* p = a.stats + ((long long)tile_index * 2) * a.coutp;
        p[co] = t1;
        p[a.coutp + co] = t2;
      }
    }
  }
}
