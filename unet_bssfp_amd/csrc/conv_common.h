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
  int ksplit;                      // halo kernel: contraction split over blockIdx.z (1 = off)
  float* kslab;                    // [ksplit][n*do*ho*wo][coutp] f32 partial sums
  int cls_cout;                    // gather kernel: transposed-conv classes folded into the cout index (0 = off)
  int nbias;                       // entries of bias
};

// Epilogue shared by both kernels.
//   acc[vt][ct][i]  : row = acc_row(i,h) of subtile vt, col = lane&31 of cout tile ct
//   yoff[vt]        : this lane's (row = lane&31) output ELEMENT offset of voxel `row`, or -1
// Stores z = acc + bias, and (optionally) per-WG channel sums of acc / acc^2 over valid rows.
template <typename T, int VT, int CT>
__device__ __forceinline__ void conv_epilogue(const ConvArgs& a, f32x16 (&acc)[VT][CT],
                                              const long long (&yoff)[VT], int co_base,
                                              int tile_index, float* red /* LDS, >= 4*CT*64 floats */,
                                              int co_store_base = -1, char* tpatch = nullptr /* LDS, 4 x 2 KB */) {
  if (co_store_base < 0) co_store_base = co_base;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, h = lane >> 5;
  float s1[CT], s2[CT];
#pragma unroll
  for (int ct = 0; ct < CT; ++ct) { s1[ct] = 0.f; s2[ct] = 0.f; }

  bool stored = false;
  if constexpr (sizeof(T) == 2) {
    // 16-bit outputs whose stored channel count is a multiple of 8: transpose each 32x32 tile through a
    // wave-private LDS patch and write 16 bytes per lane (as conv_epilogue_tile does) instead of 2-byte stores
    // -- the transposed-conv forward writes 8x its input volume and was store-issue bound (192 us for 268 MB).
    if ((a.cstore & 7) == 0 && tpatch != nullptr) {
      stored = true;
      char* wbuf = tpatch + wave * 2048;
#pragma unroll
      for (int vt = 0; vt < VT; ++vt) {
        const int lo = (int)(yoff[vt] & 0xffffffffll), hi = (int)(yoff[vt] >> 32);
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
          const int co = co_store_base + ct * 32 + r;
          const float bv = (a.bias && co < a.nbias) ? a.bias[co] : 0.f;
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            const int row = acc_row(i, h);
            const float v = acc[vt][ct][i];
            if (__shfl(hi, row, 64) >= 0) { s1[ct] += v; s2[ct] += v * v; }
            *reinterpret_cast<uint16_t*>(wbuf + row * 64 + r * 2) = f32_to_bf16_bits(v + bv);
          }
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          const int co0 = co_store_base + ct * 32 + (lane & 3) * 8;
#pragma unroll
          for (int pass = 0; pass < 2; ++pass) {
            const int vrow = pass * 16 + (lane >> 2);
            const int olo = __shfl(lo, vrow, 64), ohi = __shfl(hi, vrow, 64);
            const long long off = ((long long)ohi << 32) | (unsigned int)olo;
            const uint4 val = *reinterpret_cast<const uint4*>(wbuf + vrow * 64 + (lane & 3) * 16);
            if (off >= 0 && co0 + 8 <= a.cstore) *reinterpret_cast<uint4*>(reinterpret_cast<T*>(a.y) + off + co0) = val;
          }
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
      }
    }
  }
  if (!stored)
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
          const int co = co_store_base + ct * 32 + r;
          const float v = acc[vt][ct][i];
          s1[ct] += v;
          s2[ct] += v * v;
          if (co < a.cstore) {
            const float b = (a.bias && co < a.nbias) ? a.bias[co] : 0.f;
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

// Epilogue of the halo kernel: subtile row m = acc_row(i, h) sits at (m / TW, m % TW) of the subtile's
// RS x TW patch, so its output address is base + (m / TW) * hstride + (m % TW) * wstride with
// compile-time m / TW and (m % TW) - 4h: no shuffles, one 64-bit add per row.
template <int VT> struct TileOut {
  long long base[VT]; long long hstride, wstride;
  bool dvalid[VT]; int hleft[VT], wleft[VT];
};
struct TileOutDyn { long long hstride, wstride; };

template <typename T, int VT, int CT, int TW, int NW, typename TO>
__device__ __forceinline__ void conv_epilogue_tile(const ConvArgs& a, f32x16 (&acc)[VT][CT], const TO& to, int co_base,
                                                   int tile_index, float* red) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, h = lane >> 5;
  float s1[CT], s2[CT], bias[CT];
  T* yp[CT];
#pragma unroll
  for (int ct = 0; ct < CT; ++ct) {
    s1[ct] = 0.f; s2[ct] = 0.f;
    const int co = co_base + ct * 32 + r;
    bias[ct] = (a.bias && co < a.nbias) ? a.bias[co] : 0.f;
    yp[ct] = co < a.cstore ? reinterpret_cast<T*>(a.y) + co : nullptr;
  }
  if constexpr (sizeof(T) == 2) {
    // 16-bit outputs: a lane owns ONE channel of 16 voxels, so direct stores would be 2 bytes per lane
    // (32 store instructions per subtile, store-issue bound).  Transpose each 32x32 tile through a
    // wave-private LDS patch and write 16 bytes per lane: 2 store instructions per tile, whole 64-B rows.
    __syncthreads();                                        // the halo in LDS is dead for every wave
    char* wbuf = reinterpret_cast<char*>(red) + wave * 2048;
#pragma unroll
    for (int vt = 0; vt < VT; ++vt) {
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int m = acc_row(i, h);
          const float v = acc[vt][ct][i];
          const int m0 = (i & 3) + 8 * (i >> 2);
          const bool ok = to.dvalid[vt] && (m0 / TW) < to.hleft[vt] && (m0 % TW) + 4 * h < to.wleft[vt];
          if (ok) { s1[ct] += v; s2[ct] += v * v; }
          *reinterpret_cast<uint16_t*>(wbuf + m * 64 + r * 2) = f32_to_bf16_bits(v + bias[ct]);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this wave's LDS writes have landed (in-order LDS)
        const int co0 = co_base + ct * 32 + (lane & 3) * 8;
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
          const int v = pass * 16 + (lane >> 2);
          const int rh = v / TW, cw = v % TW;
          const uint4 val = *reinterpret_cast<const uint4*>(wbuf + v * 64 + (lane & 3) * 16);
          if (to.dvalid[vt] && rh < to.hleft[vt] && cw < to.wleft[vt] && co0 + 8 <= a.cstore)
            *reinterpret_cast<uint4*>(reinterpret_cast<T*>(a.y) + to.base[vt] + rh * to.hstride + cw * to.wstride + co0) = val;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // reads done before the patch is overwritten
      }
    }
  } else {
#pragma unroll
  for (int vt = 0; vt < VT; ++vt) {
    if (!to.dvalid[vt]) continue;
    // running row offset (one add per row, nothing loop-invariant to keep in registers)
    long long off = to.base[vt] + (long long)(4 * h) * to.wstride;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int m0 = (i & 3) + 8 * (i >> 2);          // row index without the lane-half term 4h (never carries)
      const int rh = m0 / TW, cw = m0 % TW;
      if (i > 0) {
        const int pm = ((i - 1) & 3) + 8 * ((i - 1) >> 2);
        const int drh = rh - pm / TW, dcw = cw - pm % TW;
        off += drh * to.hstride + dcw * to.wstride;
      }
      if (rh < to.hleft[vt] && cw + 4 * h < to.wleft[vt]) {
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
          const float v = acc[vt][ct][i];
          s1[ct] += v;
          s2[ct] += v * v;
          if (yp[ct]) Elem<T>::store(yp[ct] + off, v + bias[ct]);
        }
      }
    }
  }
  }
  if (a.stats) {
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
      s1[ct] += __shfl_xor(s1[ct], 32, 64);
      s2[ct] += __shfl_xor(s2[ct], 32, 64);
    }
    __syncthreads();
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
        for (int w = 0; w < NW; ++w) {
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
