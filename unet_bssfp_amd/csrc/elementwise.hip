// HBM-bound kernels of the path: layout pack/unpack, weight packing, channel statistics,
// fused norm + dropout + LeakyReLU (forward / backward), max-pool, L1 loss, multi-tensor AdamW.
// NDHWC rows are read and written as 16-byte vectors per lane (coalesced along channels).
#include <stdlib.h>
#include "common.h"

namespace {

constexpr int kRowsPerStatBlock = 2048;

// ------------------------------------------------------------------ space-to-depth addressing
// S(a)[n, jd, jh, jw, blk*C + c] = a[n, 2jd+bd-1, 2jh+bh-1, 2jw+bw-1, c], blk = bd*4 + bh*2 + bw, extents
// (D/2+1, H/2+1, W/2+1): a k4 s2 p1 convolution of `a` is a dense k2 s1 p0 convolution of S(a).
struct S2D { int d, h, w, cblk; };   // extents of the plain tensor (per sample); d == 0: off
// (cell row, block) of plain voxel `row`; `border`: bit set per axis (4 d, 2 h, 1 w) on which the voxel is the first or
// last one -- the same cell's block with that parity bit flipped would read a[-1] / a[D]: it must hold zeros, and the
// writer of the border voxel stores them (every slot of S(a) is then written by exactly one thread: the output
// tensor needs no prior zero-fill).
__device__ __forceinline__ void s2d_cell(const S2D& q, long long row /* n*D*H*W + ... */, long long& srow, int& blk, int& border) {
  const int w = (int)(row % q.w); long long t = row / q.w;
  const int h = (int)(t % q.h); t /= q.h;
  const int d = (int)(t % q.d); const long long n = t / q.d;
  srow = ((n * (q.d / 2 + 1) + ((d + 1) >> 1)) * (q.h / 2 + 1) + ((h + 1) >> 1)) * (q.w / 2 + 1) + ((w + 1) >> 1);
  blk = ((d + 1) & 1) * 4 + ((h + 1) & 1) * 2 + ((w + 1) & 1);
  border = ((d == 0 || d == q.d - 1) ? 4 : 0) | ((h == 0 || h == q.h - 1) ? 2 : 0) | ((w == 0 || w == q.w - 1) ? 1 : 0);
}
__device__ __forceinline__ long long s2d_offset(const S2D& q, long long row, int ld) {
  long long srow; int blk, border;
  s2d_cell(q, row, srow, blk, border);
  return srow * ld + (long long)blk * q.cblk;
}
// zeros into [e0, e0 + 16 B) of the border voxel's sibling blocks (all non-empty subsets of the border axes)
template <typename T>
__device__ __forceinline__ void s2d_zero_siblings(T* base, const S2D& q, long long srow, int blk, int border, int ld, int e0) {
  if (!border) return;
  Vec16<T> z;
#pragma unroll
  for (int j = 0; j < Vec16<T>::N; ++j) z.f[j] = 0.f;
  for (int sub = 1; sub < 8; ++sub)
    if ((sub & ~border) == 0) z.store(base + srow * ld + (long long)(blk ^ sub) * q.cblk + e0);
}

// ------------------------------------------------------------------ pack / unpack
// Optional second source: channels [c, c + c1) of the window come from src1 (torch.cat([x, y], 1) of the
// discriminator as ONE pass that writes whole rows; two single-source passes wrote 48 and 16 of every 64 bytes).
// Thread = (voxel, 16-byte piece of its channel window): the P pieces of a voxel row sit in P consecutive lanes, so a wave
// writes whole rows -- 64 / P voxels x P x 16 contiguous bytes -- instead of one 16-byte piece of 64 different rows per
// store instruction (the one-voxel-per-thread form ran the 251 MB -> 137 MB discriminator pack at 2.2 TB/s: 174 us);
// the loads of a piece are EPV channel-strided dwords, lanes of equal piece reading consecutive voxels (64-byte runs).
template <typename T>
__global__ __launch_bounds__(256) void pack_kernel(const float* __restrict__ src, T* __restrict__ dst, int c,
                                                    long long v, int ld, int coff, int zero_to, S2D q,
                                                    const float* __restrict__ src1, int c1, int pieces) {
  constexpr int EPV = Elem<T>::kPer16B;
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long vox = idx / pieces;
  const int piece = (int)(idx - vox * pieces);
  const int n = blockIdx.y;
  if (vox >= v) return;
  const float* s = src + (long long)n * c * v + vox;
  const float* s1 = src1 ? src1 + (long long)n * c1 * v + vox : nullptr;
  long long srow = 0; int blk = 0, border = 0;
  if (q.d) s2d_cell(q, (long long)n * v + vox, srow, blk, border);
  T* drow = q.d ? dst + srow * ld + (long long)blk * q.cblk : dst + ((long long)n * v + vox) * ld;
  const int e0 = coff + piece * EPV;
  Vec16<T> o;
#pragma unroll
  for (int j = 0; j < EPV; ++j) {
    const int ch = e0 + j - coff;
    o.f[j] = ch < c ? s[(long long)ch * v] : ((s1 && ch < c + c1) ? s1[(long long)(ch - c) * v] : 0.f);
  }
  o.store(drow + e0);
  if (q.d) s2d_zero_siblings<T>(dst, q, srow, blk, border, ld, e0);
}

// The same pack through an LDS tile, for channel windows of up to 64 elements (every activation of the path): a block owns
// 256 consecutive voxels; per source channel its threads read 256 consecutive floats (whole 128-byte lines: the
// piece-per-lane form above reads 64-byte runs of 4 channels per instruction and ran at 3 TB/s), the tile is turned in
// LDS (row stride odd in dwords: conflict-free 2- / 4-byte writes), and every voxel row leaves as contiguous 16-byte pieces.
template <typename T>
__global__ __launch_bounds__(256) void pack_tile_kernel(const float* __restrict__ src, T* __restrict__ dst, int c,
                                                         long long v, int ld, int coff, int zero_to, S2D q,
                                                         const float* __restrict__ src1, int c1) {
  constexpr int EPV = Elem<T>::kPer16B, ES = 16 / EPV;
  extern __shared__ __attribute__((aligned(16))) char tile[];
  const int wch = zero_to - coff;                         // window channels (multiple of EPV)
  const int rowb = wch * ES + 4;                          // LDS row bytes: +1 dword -> odd dword stride for 32 / 64-byte rows
  const long long v0 = (long long)blockIdx.x * 256;
  const int n = blockIdx.y, t = threadIdx.x;
  const long long vox = v0 + t;
  const bool in = vox < v;
  const float* s = src + (long long)n * c * v + vox;
  const float* s1 = src1 ? src1 + (long long)n * c1 * v + vox : nullptr;
  char* my = tile + t * rowb;
  for (int cb = 0; cb < wch; cb += 8) {                    // 8 loads in flight per thread (window channels are a multiple of 4 or 8)
    float f[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int ch = cb + j;
      f[j] = 0.f;
      if (in && ch < wch) f[j] = ch < c ? s[(long long)ch * v] : ((s1 && ch < c + c1) ? s1[(long long)(ch - c) * v] : 0.f);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (cb + j < wch) Elem<T>::store(reinterpret_cast<T*>(my + (cb + j) * ES), f[j]);
  }
  __syncthreads();
  const int pieces = wch / EPV;
  for (int idx = t; idx < 256 * pieces; idx += 256) {
    const int lv = idx / pieces, piece = idx - lv * pieces;
    const long long gv = v0 + lv;
    if (gv >= v) continue;
    const uint32_t* r = reinterpret_cast<const uint32_t*>(tile + lv * rowb + piece * 16);
    const uint4 val = make_uint4(r[0], r[1], r[2], r[3]);
    const int e0 = coff + piece * EPV;
    if (q.d) {
      long long srow; int blk, border;
      s2d_cell(q, (long long)n * v + gv, srow, blk, border);
      *reinterpret_cast<uint4*>(dst + srow * ld + (long long)blk * q.cblk + e0) = val;
      s2d_zero_siblings<T>(dst, q, srow, blk, border, ld, e0);
    } else {
      *reinterpret_cast<uint4*>(dst + ((long long)n * v + gv) * ld + e0) = val;
    }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void unpack_kernel(const T* __restrict__ src, float* __restrict__ dst, int c,
                                                      long long v, int ld, int coff, S2D q) {
  const long long vox = (long long)blockIdx.x * 256 + threadIdx.x;
  const int n = blockIdx.y;
  if (vox >= v) return;
  const T* srow = (q.d ? src + s2d_offset(q, (long long)n * v + vox, ld) : src + ((long long)n * v + vox) * ld) + coff;
  float* d = dst + (long long)n * c * v + vox;
  for (int ch = 0; ch < c; ++ch) d[(long long)ch * v] = Elem<T>::load(srow + ch);
}

// plain NDHWC activation -> its space-to-depth tensor S(a) (same element type): thread = (voxel, 16-byte piece)
template <typename T>
__global__ __launch_bounds__(256) void s2d_repack_kernel(const T* __restrict__ src, int lds_, T* __restrict__ dst, int ldd, long long v, int pieces, S2D q) {
  constexpr int EPV = Elem<T>::kPer16B;
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long vox = idx / pieces;
  const int piece = (int)(idx - vox * pieces);
  const int n = blockIdx.y;
  if (vox >= v) return;
  const long long row = (long long)n * v + vox;
  long long srow; int blk, border;
  s2d_cell(q, row, srow, blk, border);
  const uint4 val = *reinterpret_cast<const uint4*>(src + row * lds_ + piece * EPV);
  *reinterpret_cast<uint4*>(dst + srow * ldd + (long long)blk * q.cblk + piece * EPV) = val;
  s2d_zero_siblings<T>(dst, q, srow, blk, border, ldd, piece * EPV);
}

// Gradient seam between the PatchGAN and the generator (src/model.py:172, 268: the generator phase back-propagates through D into
// G): dz[v][ch] = g_ncdhw[ch][v] (the loss head's gradient of the NCDHW output, or absent) + unS(g_s)[v][ch] (the PatchGAN's
// gradient of S(y_hat), or absent), as the NDHWC activation gradient the final convolution's backward reads.  One pass instead of
// unpack (S -> NCDHW f32) + add + pack (NCDHW f32 -> NDHWC).  thread = (voxel, 16-byte piece of the output row)
template <typename T>
__global__ __launch_bounds__(256) void seam_grad_kernel(const float* __restrict__ gy, const T* __restrict__ gs, int ld_s, T* __restrict__ dz, int ld_dz,
                                                        int pieces, int c, long long v, S2D q) {
  constexpr int EPV = Elem<T>::kPer16B;
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long vox = idx / pieces;
  const int piece = (int)(idx - vox * pieces), n = blockIdx.y;
  if (vox >= v) return;
  Vec16<T> o;
#pragma unroll
  for (int j = 0; j < EPV; ++j) {
    const int ch = piece * EPV + j;
    o.f[j] = (gy && ch < c) ? gy[((long long)n * c + ch) * v + vox] : 0.f;
  }
  if (gs && piece * EPV < q.cblk) {
    Vec16<T> sv;
    sv.load(gs + s2d_offset(q, (long long)n * v + vox, ld_s) + piece * EPV);
#pragma unroll
    for (int j = 0; j < EPV; ++j) o.f[j] += sv.f[j];
  }
  o.store(dz + ((long long)n * v + vox) * ld_dz + piece * EPV);
}

// ------------------------------------------------------------------ weight pack
struct WpackArgs {
  const float* src; void* dst;
  int cout, cin, coutp, cinp, ks;
  long long s_co, s_ci, s_k0, s_k1, s_k2;
  int tb0, tb1, tb2, ts0, ts1, ts2;
  int s2d_mode, s2d_cp;   // 1: the GEMM cin index is (block, channel) of a space-to-depth tensor; 2: the cout index is
  const float* q_amax;    // fp8 packings: per-tensor max |w| (device), values are stored as w * 224 / amax
};
template <typename T>
__global__ __launch_bounds__(256) void wpack_kernel(const WpackArgs a) {
  const int ntaps = a.ks * a.ks * a.ks;
  const long long total = (long long)(a.cinp / 16) * ntaps * a.coutp * 16;
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int e = (int)(idx % 16);
  const int co = (int)((idx / 16) % a.coutp);
  const int tap = (int)((idx / (16ll * a.coutp)) % ntaps);
  const int chunk = (int)(idx / (16ll * a.coutp * ntaps));
  int ci = chunk * 16 + e;
  int cor = co, blk = 0;
  if (a.s2d_mode == 1) { blk = ci / a.s2d_cp; ci = ci % a.s2d_cp; }
  if (a.s2d_mode == 2) { blk = cor / a.s2d_cp; cor = cor % a.s2d_cp; }
  float v = 0.f;
  if (cor < a.cout && ci < a.cin && blk < 8) {
    const int td = tap / (a.ks * a.ks), th = (tap / a.ks) % a.ks, tw = tap % a.ks;
    v = a.src[cor * a.s_co + ci * a.s_ci + (a.tb0 + a.ts0 * td + (blk >> 2)) * a.s_k0 +
              (a.tb1 + a.ts1 * th + ((blk >> 1) & 1)) * a.s_k1 + (a.tb2 + a.ts2 * tw + (blk & 1)) * a.s_k2];
  }
  if constexpr (sizeof(T) == 1) v *= fp8_scale_of(a.q_amax);
  Elem<T>::store(reinterpret_cast<T*>(a.dst) + idx, v);
}

constexpr int kWpackChunk = 16;
struct WpackMulti { WpackArgs a[kWpackChunk]; };
template <typename T>
__global__ __launch_bounds__(256) void wpack_multi_kernel(const WpackMulti m) {
  const WpackArgs& a = m.a[blockIdx.y];
  const int ntaps = a.ks * a.ks * a.ks;
  const long long total = (long long)(a.cinp / 16) * ntaps * a.coutp * 16;
  const long long stride = (long long)gridDim.x * 256;
  for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += stride) {
    const int e = (int)(idx % 16);
    const int co = (int)((idx / 16) % a.coutp);
    const int tap = (int)((idx / (16ll * a.coutp)) % ntaps);
    const int chunk = (int)(idx / (16ll * a.coutp * ntaps));
    int ci = chunk * 16 + e;
    int cor = co, blk = 0;
    if (a.s2d_mode == 1) { blk = ci / a.s2d_cp; ci = ci % a.s2d_cp; }
    if (a.s2d_mode == 2) { blk = cor / a.s2d_cp; cor = cor % a.s2d_cp; }
    float v = 0.f;
    if (cor < a.cout && ci < a.cin && blk < 8) {
      const int td = tap / (a.ks * a.ks), th = (tap / a.ks) % a.ks, tw = tap % a.ks;
      v = a.src[cor * a.s_co + ci * a.s_ci + (a.tb0 + a.ts0 * td + (blk >> 2)) * a.s_k0 +
                (a.tb1 + a.ts1 * th + ((blk >> 1) & 1)) * a.s_k1 + (a.tb2 + a.ts2 * tw + (blk & 1)) * a.s_k2];
    }
    if constexpr (sizeof(T) == 1) v *= fp8_scale_of(a.q_amax);
    Elem<T>::store(reinterpret_cast<T*>(a.dst) + idx, v);
  }
}

// ------------------------------------------------------------------ channel statistics
// One block = up to kRowsPerStatBlock rows of one group.  Thread = (row-in-pass, 16-B piece).
// f(row values) is supplied by the functor; sums of two quantities per channel are produced.
template <typename T, typename F>
__device__ __forceinline__ void block_channel_sums(int c, long long row_begin, long long row_end, F f,
                                                   float* out0, float* out1) {
  constexpr int EPV = Elem<T>::kPer16B;
  __shared__ float red[256 * 16];
  const int lpr = c / EPV;                 // 16-B pieces per row (<= 128)
  const int rpp = 256 / lpr;               // rows per pass
  const int piece = threadIdx.x % lpr, rsub = threadIdx.x / lpr;
  float s0[EPV], s1[EPV];
#pragma unroll
  for (int j = 0; j < EPV; ++j) { s0[j] = 0.f; s1[j] = 0.f; }
  if (rsub < rpp)
    for (long long row = row_begin + rsub; row < row_end; row += rpp) f(row, piece * EPV, s0, s1);
#pragma unroll
  for (int j = 0; j < EPV; ++j) {
    red[threadIdx.x * 16 + j] = s0[j];
    red[threadIdx.x * 16 + 8 + j] = s1[j];
  }
  __syncthreads();
  for (int ch = threadIdx.x; ch < c; ch += 256) {
    const int p = ch / EPV, j = ch % EPV;
    float t0 = 0.f, t1 = 0.f;
    for (int q = 0; q < rpp; ++q) {
      t0 += red[(q * lpr + p) * 16 + j];
      t1 += red[(q * lpr + p) * 16 + 8 + j];
    }
    out0[ch] = t0;
    out1[ch] = t1;
  }
}

template <typename T>
__global__ __launch_bounds__(256) void channel_stats_kernel(const T* __restrict__ x, int ld, int c,
                                                             long long rows_per_group, float* __restrict__ part,
                                                             int blocks_per_group) {
  const int g = blockIdx.y, b = blockIdx.x;
  const long long rb = (rows_per_group + blocks_per_group - 1) / blocks_per_group;
  const long long r0 = (long long)b * rb;
  long long r1 = r0 + rb;
  if (r1 > rows_per_group) r1 = rows_per_group;
  const T* base = x + (long long)g * rows_per_group * ld;
  float* out = part + ((long long)g * blocks_per_group + b) * 2 * c;
  block_channel_sums<T>(c, r0, r1,
      [&](long long row, int ch0, float* s0, float* s1) {
        Vec16<T> v;
        v.load(base + row * ld + ch0);
#pragma unroll
        for (int j = 0; j < Vec16<T>::N; ++j) { s0[j] += v.f[j]; s1[j] += v.f[j] * v.f[j]; }
      },
      out, out + c);
}

// Sum of per-block/per-tile partials part[k][2][c] over k for 8 channels per workgroup:
// 1024 threads = 8 channels x 128 partial lanes, f64 accumulate, fixed-order shuffle + LDS combine (deterministic).
// The totals of channel ch0 + (tid & 7) are returned to the threads with tid < 8.
__device__ __forceinline__ void block_sum_parts(const float* __restrict__ part, int nparts, int c, int ch0,
                                                double& t0, double& t1) {
  __shared__ double red[16][8][2];
  const int cl = threadIdx.x & 7, pl = threadIdx.x >> 3;
  const int ch = ch0 + cl;
  double s0 = 0.0, s1 = 0.0;
  if (ch < c) {
    int k = pl;
    for (; k + 384 < nparts; k += 512) {
      const float a0 = part[(long long)k * 2 * c + ch], b0 = part[(long long)k * 2 * c + c + ch];
      const float a1 = part[(long long)(k + 128) * 2 * c + ch], b1 = part[(long long)(k + 128) * 2 * c + c + ch];
      const float a2 = part[(long long)(k + 256) * 2 * c + ch], b2 = part[(long long)(k + 256) * 2 * c + c + ch];
      const float a3 = part[(long long)(k + 384) * 2 * c + ch], b3 = part[(long long)(k + 384) * 2 * c + c + ch];
      s0 += ((double)a0 + (double)a1) + ((double)a2 + (double)a3);
      s1 += ((double)b0 + (double)b1) + ((double)b2 + (double)b3);
    }
    for (; k < nparts; k += 128) {
      s0 += (double)part[(long long)k * 2 * c + ch];
      s1 += (double)part[(long long)k * 2 * c + c + ch];
    }
  }
  // the 8 partial lanes of a wave (lane bits 3..5) by shuffles, then the 16 waves through LDS in a fixed order:
  // two barriers instead of a 7-level LDS tree (these kernels are pure latency: ~90 launches per step)
#pragma unroll
  for (int o = 8; o < 64; o <<= 1) {
    s0 += __shfl_xor(s0, o, 64);
    s1 += __shfl_xor(s1, o, 64);
  }
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) < 8) { red[wave][cl][0] = s0; red[wave][cl][1] = s1; }
  __syncthreads();
  double a0 = 0.0, a1 = 0.0;
#pragma unroll
  for (int w = 0; w < 16; ++w) { a0 += red[w][cl][0]; a1 += red[w][cl][1]; }
  t0 = a0;
  t1 = a1;
  __syncthreads();      // red is reused by the caller's next call
}

__global__ __launch_bounds__(1024) void norm_finalize_kernel(const float* __restrict__ part, int ppg, int c,
                                                             long long count, const float* __restrict__ shift,
                                                             int n_real, float eps, float* __restrict__ mean,
                                                             float* __restrict__ rstd, float* running_mean,
                                                             float* running_var, float momentum,
                                                             long long* batches_tracked, int groups_here) {
  // groups_here == 1: this workgroup's group is blockIdx.y.  groups_here > 1 (BatchNorm over several statistic groups in
  // one call -- the discriminator's fake and real batch stacked, src/model.py:185-186): the groups are walked IN ORDER by one
  // workgroup per channel block, so that the running statistics receive the momentum updates of two consecutive forward calls.
  const int ch0 = blockIdx.x * 8;
  if (batches_tracked && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) batches_tracked[0] += groups_here;   // BatchNorm's counter
  for (int gi = 0; gi < groups_here; ++gi) {
    const int g = groups_here > 1 ? gi : blockIdx.y;
    double s1, s2;
    block_sum_parts(part + (long long)g * ppg * 2 * c, ppg, c, ch0, s1, s2);
    const int ch = ch0 + (int)threadIdx.x;
    if (threadIdx.x >= 8 || ch >= c) continue;
    const double m = s1 / (double)count;
    double var = s2 / (double)count - m * m;
    if (var < 0.0) var = 0.0;
    const double mu = m + ((shift && ch < n_real) ? (double)shift[ch] : 0.0);
    mean[(long long)g * c + ch] = (float)mu;
    rstd[(long long)g * c + ch] = (float)(1.0 / sqrt(var + (double)eps));
    if (running_mean && ch < n_real) {
      const double unb = count > 1 ? var * (double)count / (double)(count - 1) : var;
      running_mean[ch] = (float)((1.0 - momentum) * running_mean[ch] + momentum * mu);
      running_var[ch] = (float)((1.0 - momentum) * running_var[ch] + momentum * unb);
    }
  }
}

// out[j] (+)= column sum of channel offset + j, j < n_out
__global__ __launch_bounds__(1024) void colsum_finalize_kernel(const float* __restrict__ part, int parts, int c, int offset,
                                                               float* __restrict__ out, int n_out, int accumulate) {
  const int ch0 = offset + blockIdx.x * 8;
  double s0, s1;
  block_sum_parts(part, parts, c, ch0, s0, s1);
  const int j = blockIdx.x * 8 + (int)threadIdx.x;
  if (threadIdx.x < 8 && j < n_out) out[j] = accumulate ? out[j] + (float)s0 : (float)s0;
}

// ------------------------------------------------------------------ norm + dropout + LeakyReLU
// effective seed: a per-call salt, optionally combined with a step counter kept in device memory (so
// that a launch captured in a hipGraph draws a new mask on every replay)
__device__ __forceinline__ unsigned long long eff_seed(unsigned long long salt, const unsigned long long* p) {
  return p ? p[0] * 0x9E3779B97F4A7C15ull + salt * 0xD1B54A32D192ED03ull + 1ull : salt;
}
// keep-mask of element `idx` (logical index, independent of ld): 16 bits per element, one 64-bit mix per
// aligned group of 4 elements (the compiler shares it across the 4 / 8 elements of a 16-byte piece)
__device__ __forceinline__ unsigned long long drop_group_bits(unsigned long long seed, unsigned long long group) {
  unsigned long long x = group * 0x9E3779B97F4A7C15ull + seed;
  x ^= x >> 32; x *= 0xD6E8FEB86659FD93ull; x ^= x >> 32;
  return x;
}
// keep-masks of the EPV (4 or 8) elements of one 16-byte piece starting at logical index e0 (a multiple of 4:
// channel counts are multiples of 16).  The mixes are computed explicitly once per group of 4 -- hipcc cannot
// prove the alignment of e0 and would otherwise mix once per element.
template <int EPV>
__device__ __forceinline__ unsigned drop_keep_mask(unsigned long long seed, unsigned long long e0, unsigned thr16) {
  unsigned m = 0;
#pragma unroll
  for (int g4 = 0; g4 < EPV / 4; ++g4) {
    const unsigned long long x = drop_group_bits(seed, (e0 >> 2) + g4);
#pragma unroll
    for (int k = 0; k < 4; ++k) m |= ((((unsigned)(x >> (16 * k))) & 0xffffu) >= thr16 ? 1u : 0u) << (4 * g4 + k);
  }
  return m;
}

// ---- e4m3 copies written by the PRODUCER of an fp8 convolution's operand (delayed per-tensor scaling) ----
// block maximum -> at most one atomic per workgroup, and none when the tensor-wide maximum is already there (m >= 0: bit
// order = value order).  Every thread of a 256-thread workgroup calls it.  One atomic per WAVE was measured to double
// the kernels: all waves finish together and ~6 ns per same-address atomic serialise behind each other.
__device__ __forceinline__ void amax_commit(float m, float* out) {
  __shared__ float red[4];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    const unsigned cur = __hip_atomic_load(reinterpret_cast<unsigned int*>(out), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (__float_as_uint(m) > cur) atomicMax(reinterpret_cast<unsigned int*>(out), __float_as_uint(m));
  }
}
// 8 results that are about to be stored as bf16 -> the 8 e4m3 bytes mi355_cast_fp8 would make of the stored tensor
// (rounded through bf16 first, then * 224 / amax); returns max |bf16 value| for the next step's scale
__device__ __forceinline__ float e4m3_piece(const float (&f)[8], float sc, uint8_t* dst) {
  float m = 0.f, r[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float b = __uint_as_float((uint32_t)f32_to_bf16_bits(f[j]) << 16);
    m = fmaxf(m, fabsf(b));
    r[j] = b * sc;
  }
  uint32_t w0 = cvt_pk_fp8(r[0], r[1], 0u, false), w1 = cvt_pk_fp8(r[4], r[5], 0u, false);
  w0 = cvt_pk_fp8(r[2], r[3], w0, true);
  w1 = cvt_pk_fp8(r[6], r[7], w1, true);
  *reinterpret_cast<uint2*>(dst) = make_uint2(w0, w1);
  return m;
}

struct NormActArgs {
  const char* z; int ldz; char* a; int lda;
  int c; long long rows_per_group; int groups;
  const float* mean; const float* rstd; const float* gamma; const float* beta;
  float slope; float drop_scale; unsigned thr16; unsigned long long seed; const unsigned long long* seed_ptr;
  const char* da; int ldda; char* dz; int lddz;
  float* part; int blocks_per_group; const float* sums; int batch_stats;
  int n_affine;   // entries of gamma / beta
  S2D s2d_a;      // forward: write `a` in space-to-depth layout
  S2D s2d_da;     // backward: read `da` from a space-to-depth tensor
  uint8_t* q8; int ld8; const float* q_use; float* q_next;   // e4m3 copy of a (fwd) / dz (bwd_apply), bf16 with c == 32 only
  // backward, optional: da is NOT materialised -- it is the data gradient of the 1x1x1 convolution that consumed a:
  // da[row][ch] = bf16(sum_k gz[row][k] * gw[k][ch]), k < gk <= 8 (bf16 only)
  const char* gz; int ldgz; const float* gw; int gw_ld; int gk;
  // forward, optional: the 1x1x1 convolution that consumes a, evaluated per row on the rounded bf16 values:
  // fy[row][k] = bf16(sum_ch a[row][ch] * bf16(gw[k][ch]) + fbias[k]), k < gk <= 8, channels gk .. fcp - 1 zero; skip_a: a itself is not stored
  char* fy; int ldfy; int fcp; const float* fbias; int skip_a;
  // backward, optional: da is NOT materialised -- a was consumed by MaxPool3d(2) (and, optionally, a skip connection whose gradient
  // is `da`): da[v][ch] = (pool_idx[o(v)][ch] == k(v) ? pool_dy[o(v)][ch] : 0) (+ da[v][ch]), rounded to T like the stored tensor
  const uint8_t* pool_idx; const char* pool_dy; int ldpdy; int pd, ph, pw;
  // forward, optional: MaxPool3d(2) of a in the pass that writes it: pool_y[o][ch] = max over the window, pool_widx = the window
  // positions (as mi355_maxpool2_fwd_idx), extents pd x ph x pw (even)
  char* pool_y; int ldpy; uint8_t* pool_widx;
};

// fp8 mode: the next convolution reads the e4m3 copy this launch writes, the bf16 tensor's next reader is the weight gradient a
// backward pass away -- stored non-temporally.  1x24x160^3, three interleaved rounds: fp8 18.48 against 18.65 ms per step (bf16: 18.53).
#ifndef MI355_DIAG_NO_NT
#define FP8_NT_A 1
#endif
// Vec16<T>::store with the non-temporal policy (a tensor whose next reader is far away)
__device__ __forceinline__ void st_nt_b128(void* p, const uint4 v) {
  typedef unsigned u4 __attribute__((ext_vector_type(4)));
  const u4 x = {v.x, v.y, v.z, v.w};
  __builtin_nontemporal_store(x, reinterpret_cast<u4*>(p));
}
__device__ __forceinline__ void store16_nt(const Vec16<float>& v, void* p) {
  st_nt_b128(p, make_uint4(__float_as_uint(v.f[0]), __float_as_uint(v.f[1]), __float_as_uint(v.f[2]), __float_as_uint(v.f[3])));
}
__device__ __forceinline__ void store16_nt(const Vec16<bf16_t>& v, void* p) {
  uint32_t w[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) w[i] = (uint32_t)f32_to_bf16_bits(v.f[2 * i]) | ((uint32_t)f32_to_bf16_bits(v.f[2 * i + 1]) << 16);
  st_nt_b128(p, make_uint4(w[0], w[1], w[2], w[3]));
}

template <typename T, bool DROP>
__global__ __launch_bounds__(256) void normact_fwd_kernel(const NormActArgs q) {
  constexpr int EPV = Elem<T>::kPer16B;
  const int g = blockIdx.y;
  const int lpr = q.c / EPV, rpp = 256 / lpr;
  const int piece = threadIdx.x % lpr, rsub = threadIdx.x / lpr;
  if (rsub >= rpp) return;
  const int ch0 = piece * EPV;
  const unsigned long long seed = DROP ? eff_seed(q.seed, q.seed_ptr) : 0ull;
  float sc[EPV], sh[EPV];
#pragma unroll
  for (int j = 0; j < EPV; ++j) {
    const int ch = ch0 + j;
    const float ga = q.gamma ? (ch < q.n_affine ? q.gamma[ch] : 0.f) : 1.f, be = (q.beta && ch < q.n_affine) ? q.beta[ch] : 0.f;
    if (q.mean) {
      const float rs = q.rstd[(long long)g * q.c + ch], mu = q.mean[(long long)g * q.c + ch];
      sc[j] = ga * rs;
      sh[j] = be - mu * ga * rs;
    } else { sc[j] = ga; sh[j] = be; }
  }
  const T* zb = reinterpret_cast<const T*>(q.z) + (long long)g * q.rows_per_group * q.ldz;
  T* ab = reinterpret_cast<T*>(q.a) + (long long)g * q.rows_per_group * q.lda;
  const long long stride = (long long)gridDim.x * rpp;
  const float sc8 = q.q8 ? fp8_scale_of(q.q_use) : 1.f;
  float m8 = 0.f;
  for (long long row = (long long)blockIdx.x * rpp + rsub; row < q.rows_per_group; row += stride) {
    Vec16<T> v;
#ifdef NORM_NT_FWD
    v.load_nt(zb + row * q.ldz + ch0);
#else
    v.load(zb + row * q.ldz + ch0);
#endif
    const unsigned long long e0 = ((unsigned long long)g * q.rows_per_group + row) * q.c + ch0;
    unsigned keep = 0;
    if constexpr (DROP) keep = drop_keep_mask<EPV>(seed, e0, q.thr16);
#pragma unroll
    for (int j = 0; j < EPV; ++j) {
      float t = v.f[j] * sc[j] + sh[j];
      if constexpr (DROP) t = (keep >> j) & 1u ? t * q.drop_scale : 0.f;
      v.f[j] = t > 0.f ? t : t * q.slope;
    }
    if (q.s2d_a.d) {
      long long srow; int blk, border;
      s2d_cell(q.s2d_a, (long long)g * q.rows_per_group + row, srow, blk, border);
      v.store(reinterpret_cast<T*>(q.a) + srow * q.lda + (long long)blk * q.s2d_a.cblk + ch0);
      s2d_zero_siblings<T>(reinterpret_cast<T*>(q.a), q.s2d_a, srow, blk, border, q.lda, ch0);
    }
#ifdef FP8_NT_A     // (fp8 mode: the next convolution reads the e4m3 copy; the bf16 tensor's next reader is the weight gradient, a backward pass away)
    else if (q.q8) store16_nt(v, ab + row * q.lda + ch0);
#endif
    else v.store(ab + row * q.lda + ch0);
    if constexpr (sizeof(T) == 2) {
      if (q.q8) m8 = fmaxf(m8, e4m3_piece(v.f, sc8, q.q8 + ((long long)g * q.rows_per_group + row) * q.ld8 + ch0));
    }
  }
  if constexpr (sizeof(T) == 2) {
    if (q.q8) amax_commit(m8, q.q_next);      // (c == 32: every lane of the wave is here)
  }
}

// Norm + act TOGETHER WITH the MaxPool3d(2) that consumes the result (an encoder level of the U-Net: the activation goes to the
// skip connection and to the pool, src/model.py:22-28 via MONAI's Down): thread = (pooled voxel, 16-byte piece) walks the eight
// voxels of its window -- reads z, writes a, keeps the running maximum of the ROUNDED values (what maxpool_fwd_kernel would
// read back) and its window position.  Saves the pool launch's read of a (134 MB at 128^3 x 32), in both generator forwards
// of a training step.  blockIdx.y = sample.
template <typename T, bool DROP>
__global__ __launch_bounds__(256) void normact_pool_fwd_kernel(const NormActArgs q) {
  constexpr int EPV = Elem<T>::kPer16B;
  const int lpr = q.c / EPV, vpp = 256 / lpr;                     // pooled voxels per pass of the workgroup
  const int piece = threadIdx.x % lpr, vsub = threadIdx.x / lpr;
  if (vsub >= vpp) return;
  const int ch0 = piece * EPV;
  const int n_ = blockIdx.y;
  const long long dhw = (long long)q.pd * q.ph * q.pw;
  const int g = (int)(((long long)n_ * dhw) / q.rows_per_group);
  const unsigned long long seed = DROP ? eff_seed(q.seed, q.seed_ptr) : 0ull;
  float sc[EPV], sh[EPV];
#pragma unroll
  for (int j = 0; j < EPV; ++j) {
    const int ch = ch0 + j;
    const float ga = q.gamma ? (ch < q.n_affine ? q.gamma[ch] : 0.f) : 1.f, be = (q.beta && ch < q.n_affine) ? q.beta[ch] : 0.f;
    if (q.mean) {
      const float rs = q.rstd[(long long)g * q.c + ch], mu = q.mean[(long long)g * q.c + ch];
      sc[j] = ga * rs;
      sh[j] = be - mu * ga * rs;
    } else { sc[j] = ga; sh[j] = be; }
  }
  const int od_ = q.pd / 2, oh_ = q.ph / 2, ow_ = q.pw / 2;
  const int pooled = od_ * oh_ * ow_;
  const T* zb = reinterpret_cast<const T*>(q.z);
  T* ab = reinterpret_cast<T*>(q.a);
  for (int o = blockIdx.x * vpp + vsub; o < pooled; o += gridDim.x * vpp) {
    const int ow = o % ow_, t = o / ow_, oh = t % oh_, od = t / oh_;
    Vec16<T> m;
    unsigned long long where = 0;
    unsigned nan_seen = 0;
#pragma unroll
    for (int kd = 0; kd < 2; ++kd)
#pragma unroll
      for (int kh = 0; kh < 2; ++kh)
#pragma unroll
        for (int kw = 0; kw < 2; ++kw) {
          const long long vox = (((long long)n_ * q.pd + 2 * od + kd) * q.ph + 2 * oh + kh) * q.pw + 2 * ow + kw;
          const unsigned long long kk = (unsigned long long)(kd * 4 + kh * 2 + kw);
          Vec16<T> v;
#ifdef NORM_NT_FWD
          v.load_nt(zb + vox * q.ldz + ch0);
#else
          v.load(zb + vox * q.ldz + ch0);
#endif
          unsigned keep = 0;
          if constexpr (DROP) keep = drop_keep_mask<EPV>(seed, (unsigned long long)vox * q.c + ch0, q.thr16);
#pragma unroll
          for (int j = 0; j < EPV; ++j) {
            float t2 = v.f[j] * sc[j] + sh[j];
            if constexpr (DROP) t2 = (keep >> j) & 1u ? t2 * q.drop_scale : 0.f;
            t2 = t2 > 0.f ? t2 : t2 * q.slope;
            if constexpr (sizeof(T) == 2) t2 = bf16_bits_to_f32(f32_to_bf16_bits(t2));      // the stored value
            v.f[j] = t2;
          }
          v.store(ab + vox * q.lda + ch0);
          if (kk == 0) {
            m = v;
#pragma unroll
            for (int j = 0; j < EPV; ++j) nan_seen |= (v.f[j] != v.f[j] ? 1u : 0u) << j;
          } else {
#pragma unroll
            for (int j = 0; j < EPV; ++j) {                       // (as maxpool_fwd_kernel)
              const bool isnan_ = v.f[j] != v.f[j];
              const bool take = v.f[j] > m.f[j] || isnan_;
              m.f[j] = take ? v.f[j] : m.f[j];
              const bool mark = ((nan_seen >> j) & 1u) ? false : take;
              where = mark ? ((where & ~(0xffull << (8 * j))) | (kk << (8 * j))) : where;
              nan_seen |= (isnan_ ? 1u : 0u) << j;
            }
          }
        }
    const long long orow = (long long)n_ * pooled + o;
    m.store(reinterpret_cast<T*>(q.pool_y) + orow * q.ldpy + ch0);
    if constexpr (EPV == 8) *reinterpret_cast<unsigned long long*>(q.pool_widx + orow * q.c + ch0) = where;
    else *reinterpret_cast<unsigned*>(q.pool_widx + orow * q.c + ch0) = (unsigned)where;
  }
}

// Norm + act of a 32-channel bf16 tensor TOGETHER WITH the 1x1x1 convolution that consumes it (NormActArgs::fy: the U-Net's last block
// and its final convolution), the convolution on the MATRIX pipe.  Forms tried at 128^3 (plain kernel 57 us + the convolution launch it
// replaces 57 us): thread = 8 channels with v_dot2c_f32_bf16, weights in LDS, 16 cross-lane adds per row: 98 us; lane = 16 channels of a
// row + two v_mfma_f32_32x32x16_bf16 per 32 rows (96 registers + 16 accumulators: 4 waves per SIMD): 65 - 73 us.  This one keeps the plain
// kernel's work per thread: a wave owns 16 consecutive rows, lane (n, p) = row n, channel piece p (8 channels, one 16-byte load), and the
// rounded bf16 words it stores ARE the B operand of ONE v_mfma_f32_16x16x32_bf16 (k = 8 p + e; A = the weights, rows = outputs): lane
// (n, p) ends up with outputs 4 p .. 4 p + 3 of row n -- outputs 8..15 are the zero padding of the 16-channel output row, so every lane
// stores 8 bytes and the row is complete.
typedef float f32x4v __attribute__((ext_vector_type(4)));
template <bool DROP>
__global__ __launch_bounds__(256) void normact_fwd_final32_kernel(const NormActArgs q) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n = lane & 15, p = lane >> 4;
  const int g = blockIdx.y, ch0 = 8 * p;
  const unsigned long long seed = DROP ? eff_seed(q.seed, q.seed_ptr) : 0ull;
  float sc[8], sh[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int ch = ch0 + j;
    const float ga = q.gamma ? (ch < q.n_affine ? q.gamma[ch] : 0.f) : 1.f, be = (q.beta && ch < q.n_affine) ? q.beta[ch] : 0.f;
    if (q.mean) {
      const float rs = q.rstd[(long long)g * 32 + ch], mu = q.mean[(long long)g * 32 + ch];
      sc[j] = ga * rs;
      sh[j] = be - mu * ga * rs;
    } else { sc[j] = ga; sh[j] = be; }
  }
  // A fragment: row m = n (an output channel, < gk), k = 8 p + e (input channel)
  uint4 wf;
  {
    unsigned w[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int ch = ch0 + 2 * i;
      const float lo = (n < q.gk && ch < q.gw_ld) ? q.gw[(long long)n * q.gw_ld + ch] : 0.f;
      const float hi = (n < q.gk && ch + 1 < q.gw_ld) ? q.gw[(long long)n * q.gw_ld + ch + 1] : 0.f;
      w[i] = (unsigned)f32_to_bf16_bits(lo) | ((unsigned)f32_to_bf16_bits(hi) << 16);
    }
    wf = make_uint4(w[0], w[1], w[2], w[3]);
  }
  float fb[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) fb[j] = (q.fbias && 4 * p + j < q.gk) ? q.fbias[4 * p + j] : 0.f;
  const bf16_t* zb = reinterpret_cast<const bf16_t*>(q.z) + (long long)g * q.rows_per_group * q.ldz;
  bf16_t* ab = reinterpret_cast<bf16_t*>(q.a) + (long long)g * q.rows_per_group * q.lda;
  bf16_t* yb = reinterpret_cast<bf16_t*>(q.fy) + (long long)g * q.rows_per_group * q.ldfy;
  const long long stride = (long long)gridDim.x * 64;
  // (the next block's piece is loaded before this block's is processed: load -> math -> MFMA -> store is one dependent chain per wave)
  const long long first = (long long)blockIdx.x * 64 + wave * 16;
  uint4 zn = make_uint4(0u, 0u, 0u, 0u);
  if (first < q.rows_per_group) {
    const long long rc = first + n < q.rows_per_group ? first + n : q.rows_per_group - 1;
    zn = *reinterpret_cast<const uint4*>(zb + rc * q.ldz + ch0);
  }
  for (long long row0 = first; row0 < q.rows_per_group; row0 += stride) {      // wave-uniform
    const long long row = row0 + n;
    const bool ok = row < q.rows_per_group;
    const long long rowc = ok ? row : q.rows_per_group - 1;
    const uint4 zc = zn;
    if (row0 + stride < q.rows_per_group) {
      const long long rc = row + stride < q.rows_per_group ? row + stride : q.rows_per_group - 1;
      zn = *reinterpret_cast<const uint4*>(zb + rc * q.ldz + ch0);
    }
    Vec16<bf16_t> v;
    v.from_bits(zc);
    unsigned keep = 0;
    if constexpr (DROP) keep = drop_keep_mask<8>(seed, ((unsigned long long)g * q.rows_per_group + rowc) * 32 + ch0, q.thr16);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float x = v.f[j] * sc[j] + sh[j];
      if constexpr (DROP) x = (keep >> j) & 1u ? x * q.drop_scale : 0.f;
      v.f[j] = x > 0.f ? x : x * q.slope;
    }
    unsigned w[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) w[i] = (unsigned)f32_to_bf16_bits(v.f[2 * i]) | ((unsigned)f32_to_bf16_bits(v.f[2 * i + 1]) << 16);
    const uint4 aw = make_uint4(w[0], w[1], w[2], w[3]);
    if (ok && !q.skip_a) *reinterpret_cast<uint4*>(ab + row * q.lda + ch0) = aw;
    f32x4v acc = {0.f, 0.f, 0.f, 0.f};
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf), __builtin_bit_cast(bf16x8, aw), acc, 0, 0, 0);
    if (ok && 4 * p < q.fcp) {
      // outputs 4 p .. 4 p + 3 of this lane's row (p >= 2: the padding channels, whose weight rows are zero)
      const uint2 yv = make_uint2((uint32_t)f32_to_bf16_bits(acc[0] + fb[0]) | ((uint32_t)f32_to_bf16_bits(acc[1] + fb[1]) << 16),
                                  (uint32_t)f32_to_bf16_bits(acc[2] + fb[2]) | ((uint32_t)f32_to_bf16_bits(acc[3] + fb[3]) << 16));
      *reinterpret_cast<uint2*>(yb + row * q.ldfy + 4 * p) = yv;
    }
  }
}

// per-thread channel constants of the backward kernels
template <int EPV> struct BwdConst { float mu[EPV], rs[EPV], ga[EPV], be[EPV]; };
template <int EPV>
__device__ __forceinline__ void load_bwd_const(const NormActArgs& q, int g, int ch0, BwdConst<EPV>& k) {
#pragma unroll
  for (int j = 0; j < EPV; ++j) {
    const int ch = ch0 + j;
    k.mu[j] = q.mean ? q.mean[(long long)g * q.c + ch] : 0.f;
    k.rs[j] = q.mean ? q.rstd[(long long)g * q.c + ch] : 1.f;
    k.ga[j] = q.gamma ? (ch < q.n_affine ? q.gamma[ch] : 0.f) : 1.f;
    k.be[j] = (q.beta && ch < q.n_affine) ? q.beta[ch] : 0.f;
  }
}
// implicit da (NormActArgs::gz): the 1x1x1 weights as bf16 PAIRS (k even | k odd), one uint4 per channel in LDS (512 B for 32
// channels: as registers -- 64 f32 or 32 packed -- they took the streaming kernels from 5 waves per SIMD to 3-4 and made them
// slower than reading the materialised gradient), and one row's EPV gradients by v_dot2c_f32_bf16 on the row's raw bf16 pairs.
// The weights are rounded to bf16 like the packed weights of the launch this replaces.
constexpr int kImplicitMaxC = 64;
__device__ __forceinline__ void fill_implicit_w(const NormActArgs& q, uint4* wtab) {
  const int ch = threadIdx.x;
  if (ch < q.c && ch < kImplicitMaxC) {
    unsigned w[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const bool okc = ch < q.gw_ld;
      const float lo = (okc && 2 * i < q.gk) ? q.gw[(long long)(2 * i) * q.gw_ld + ch] : 0.f;
      const float hi = (okc && 2 * i + 1 < q.gk) ? q.gw[(long long)(2 * i + 1) * q.gw_ld + ch] : 0.f;
      w[i] = (unsigned)f32_to_bf16_bits(lo) | ((unsigned)f32_to_bf16_bits(hi) << 16);
    }
    wtab[ch] = make_uint4(w[0], w[1], w[2], w[3]);
  }
  __syncthreads();
}
template <typename T, int EPV>
__device__ __forceinline__ void implicit_da(const NormActArgs& q, long long grow, const uint4* wtab, int ch0, Vec16<T>& dv) {
  if constexpr (sizeof(T) == 2) {
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    const uint4 gv = *reinterpret_cast<const uint4*>(reinterpret_cast<const T*>(q.gz) + grow * q.ldgz);   // channels 0..7 (gk <= 8)
    asm volatile("" : "+v"(ch0));      // opaque per row: otherwise the table reads are hoisted out of the row loop into 32 registers
#pragma unroll
    for (int j = 0; j < EPV; ++j) {
      const uint4 wv = wtab[ch0 + j];
      float t = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, gv.x), __builtin_bit_cast(bf16x2, wv.x), 0.f, false);
      t = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, gv.y), __builtin_bit_cast(bf16x2, wv.y), t, false);
      t = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, gv.z), __builtin_bit_cast(bf16x2, wv.z), t, false);
      t = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, gv.w), __builtin_bit_cast(bf16x2, wv.w), t, false);
      dv.f[j] = bf16_bits_to_f32(f32_to_bf16_bits(t));                   // what a stored bf16 gradient would hold
    }
  }
}
// implicit da (NormActArgs::pool_idx): the gradient MaxPool3d(2)'s backward would have written for global row `grow` (voxel
// (n, d, h, w) of the full-resolution grid): the pooled gradient where this voxel was its window's (first) maximum -- the window
// position recorded by maxpool_fwd_kernel -- plus the skip connection's gradient, rounded to T as maxpool_bwd_kernel stores it.
template <typename T, int EPV>
__device__ __forceinline__ void pooled_da(const NormActArgs& q, long long grow, int ch0, Vec16<T>& dv) {
  const unsigned v = (unsigned)grow, W = (unsigned)q.pw, H = (unsigned)q.ph, D = (unsigned)q.pd;
  const unsigned w_ = v % W, t1 = v / W, h_ = t1 % H, t2 = t1 / H, d_ = t2 % D, n_ = t2 / D;
  const long long o = (((long long)n_ * (D >> 1) + (d_ >> 1)) * (H >> 1) + (h_ >> 1)) * (W >> 1) + (w_ >> 1);
  const unsigned k = ((d_ & 1u) << 2) | ((h_ & 1u) << 1) | (w_ & 1u);
  Vec16<T> gy;
  gy.load(reinterpret_cast<const T*>(q.pool_dy) + o * q.ldpdy + ch0);
  unsigned long long ib;
  if constexpr (EPV == 8) ib = *reinterpret_cast<const unsigned long long*>(q.pool_idx + o * q.c + ch0);
  else ib = *reinterpret_cast<const unsigned*>(q.pool_idx + o * q.c + ch0);
  if (q.da) dv.load(reinterpret_cast<const T*>(q.da) + grow * q.ldda + ch0);
#pragma unroll
  for (int j = 0; j < EPV; ++j) {
    float t = ((unsigned)(ib >> (8 * j)) & 0xffu) == k ? gy.f[j] : 0.f;
    if (q.da) t += dv.f[j];
    if constexpr (sizeof(T) == 2) t = bf16_bits_to_f32(f32_to_bf16_bits(t));
    dv.f[j] = t;
  }
}
// g = da * dropout * lrelu'(pre);  xhat = (z - mean) * rstd  (xhat = z when there is no norm)
template <bool DROP>
__device__ __forceinline__ void bwd_elem(const NormActArgs& q, bool keep, float mu, float rs, float ga,
                                         float be, float zv, float dav, float& gout, float& xhat) {
  xhat = (zv - mu) * rs;
  float pre = xhat * ga + be;
  float gv = dav;
  if constexpr (DROP) {
    pre = keep ? pre : 0.f;
    gv = keep ? gv * q.drop_scale : 0.f;
  }
  gout = pre > 0.f ? gv : gv * q.slope;
}

// IMPL: 0 = da is a tensor, 1 = implicit 1x1x1 data gradient (gz), 2 = implicit max-pool backward (pool_idx)
template <typename T, bool DROP, int IMPL = 0>
__global__ __launch_bounds__(256) void normact_bwd_reduce_kernel(const NormActArgs q) {
  constexpr int EPV = Elem<T>::kPer16B;
  const int g = blockIdx.y, b = blockIdx.x;
  const long long rb = (q.rows_per_group + q.blocks_per_group - 1) / q.blocks_per_group;
  const long long r0 = (long long)b * rb;
  long long r1 = r0 + rb;
  if (r1 > q.rows_per_group) r1 = q.rows_per_group;
  const T* zb = reinterpret_cast<const T*>(q.z) + (long long)g * q.rows_per_group * q.ldz;
  const T* db = reinterpret_cast<const T*>(q.da) + (long long)g * q.rows_per_group * q.ldda;
  float* out = q.part + ((long long)g * q.blocks_per_group + b) * 2 * q.c;
  BwdConst<EPV> k;
  load_bwd_const<EPV>(q, g, (int)(threadIdx.x % (q.c / EPV)) * EPV, k);
  const unsigned long long seed = DROP ? eff_seed(q.seed, q.seed_ptr) : 0ull;
  __shared__ uint4 wtab[IMPL == 1 ? kImplicitMaxC : 1];
  if constexpr (IMPL == 1) fill_implicit_w(q, wtab);
  block_channel_sums<T>(q.c, r0, r1,
      [&](long long row, int ch0, float* s0, float* s1) {
        Vec16<T> zv, dv;
        zv.load(zb + row * q.ldz + ch0);
        if constexpr (IMPL == 1) implicit_da<T, EPV>(q, (long long)g * q.rows_per_group + row, wtab, ch0, dv);
        else if constexpr (IMPL == 2) pooled_da<T, EPV>(q, (long long)g * q.rows_per_group + row, ch0, dv);
        else if (q.s2d_da.d) dv.load(reinterpret_cast<const T*>(q.da) + s2d_offset(q.s2d_da, (long long)g * q.rows_per_group + row, q.ldda) + ch0);
        else dv.load(db + row * q.ldda + ch0);
        unsigned keep = 0;
        if constexpr (DROP) keep = drop_keep_mask<EPV>(seed, ((unsigned long long)g * q.rows_per_group + row) * q.c + ch0, q.thr16);
#pragma unroll
        for (int j = 0; j < EPV; ++j) {
          float gv, xh;
          bwd_elem<DROP>(q, (keep >> j) & 1u, k.mu[j], k.rs[j], k.ga[j], k.be[j], zv.f[j], dv.f[j], gv, xh);
          s0[j] += gv;
          s1[j] += gv * xh;
        }
      },
      out, out + q.c);
}

__global__ __launch_bounds__(1024) void normact_bwd_finalize_kernel(const float* __restrict__ part, int bpg,
                                                                    int groups, int c, float* __restrict__ sums,
                                                                    float* dgamma, float* dbeta, int n_affine, int accumulate) {
  const int ch0 = blockIdx.x * 8;
  const int ch = ch0 + (int)threadIdx.x;
  const bool owner = threadIdx.x < 8 && ch < c;
  double tg = 0.0, tb = 0.0;
  for (int g = 0; g < groups; ++g) {
    double s0, s1;
    block_sum_parts(part + (long long)g * bpg * 2 * c, bpg, c, ch0, s0, s1);
    if (owner) {
      sums[((long long)g * 2 + 0) * c + ch] = (float)s0;
      sums[((long long)g * 2 + 1) * c + ch] = (float)s1;
      tb += s0;
      tg += s1;
    }
  }
  if (owner && ch < n_affine) {
    if (dgamma) dgamma[ch] = accumulate ? dgamma[ch] + (float)tg : (float)tg;
    if (dbeta) dbeta[ch] = accumulate ? dbeta[ch] + (float)tb : (float)tb;
  }
}

template <typename T, bool DROP, int IMPL = 0>
__global__ __launch_bounds__(256, IMPL == 1 ? 5 : 1) void normact_bwd_apply_kernel(const NormActArgs q) {   // (IMPL 1: 97 registers without the bound: 4 waves)
  constexpr int EPV = Elem<T>::kPer16B;
  __shared__ uint4 wtab[IMPL == 1 ? kImplicitMaxC : 1];
  if constexpr (IMPL == 1) fill_implicit_w(q, wtab);     // (before the early return below: it ends with a barrier)
  const int g = blockIdx.y;
  const int lpr = q.c / EPV, rpp = 256 / lpr;
  const int piece = threadIdx.x % lpr, rsub = threadIdx.x / lpr;
  if (rsub >= rpp) return;
  const int ch0 = piece * EPV;
  const float inv = 1.f / (float)q.rows_per_group;
  BwdConst<EPV> k;
  load_bwd_const<EPV>(q, g, ch0, k);
  const unsigned long long seed = DROP ? eff_seed(q.seed, q.seed_ptr) : 0ull;
  float kk[EPV], m0[EPV], m1[EPV];
#pragma unroll
  for (int j = 0; j < EPV; ++j) {
    const int ch = ch0 + j;
    kk[j] = k.ga[j] * k.rs[j];
    const bool sub = q.mean && q.batch_stats;
    m0[j] = sub ? q.sums[((long long)g * 2 + 0) * q.c + ch] * inv : 0.f;
    m1[j] = sub ? q.sums[((long long)g * 2 + 1) * q.c + ch] * inv : 0.f;
  }
  const T* zb = reinterpret_cast<const T*>(q.z) + (long long)g * q.rows_per_group * q.ldz;
  const T* db = reinterpret_cast<const T*>(q.da) + (long long)g * q.rows_per_group * q.ldda;
  T* ob = reinterpret_cast<T*>(q.dz) + (long long)g * q.rows_per_group * q.lddz;
  const long long stride = (long long)gridDim.x * rpp;
  const float sc8 = q.q8 ? fp8_scale_of(q.q_use) : 1.f;
  float m8 = 0.f;
  for (long long row = (long long)blockIdx.x * rpp + rsub; row < q.rows_per_group; row += stride) {
    Vec16<T> zv, dv;
#ifdef NORM_NT_APPLY
    zv.load_nt(zb + row * q.ldz + ch0);
#else
    zv.load(zb + row * q.ldz + ch0);
#endif
    if constexpr (IMPL == 1) implicit_da<T, EPV>(q, (long long)g * q.rows_per_group + row, wtab, ch0, dv);
    else if constexpr (IMPL == 2) pooled_da<T, EPV>(q, (long long)g * q.rows_per_group + row, ch0, dv);
    else if (q.s2d_da.d) dv.load(reinterpret_cast<const T*>(q.da) + s2d_offset(q.s2d_da, (long long)g * q.rows_per_group + row, q.ldda) + ch0);
#ifdef NORM_NT_APPLY
    else dv.load_nt(db + row * q.ldda + ch0);
#else
    else dv.load(db + row * q.ldda + ch0);
#endif
    unsigned keep = 0;
    if constexpr (DROP) keep = drop_keep_mask<EPV>(seed, ((unsigned long long)g * q.rows_per_group + row) * q.c + ch0, q.thr16);
#pragma unroll
    for (int j = 0; j < EPV; ++j) {
      float gv, xh;
      bwd_elem<DROP>(q, (keep >> j) & 1u, k.mu[j], k.rs[j], k.ga[j], k.be[j], zv.f[j], dv.f[j], gv, xh);
      zv.f[j] = kk[j] * (gv - m0[j] - xh * m1[j]);
    }
    zv.store(ob + row * q.lddz + ch0);
    if constexpr (sizeof(T) == 2) {
      if (q.q8) m8 = fmaxf(m8, e4m3_piece(zv.f, sc8, q.q8 + ((long long)g * q.rows_per_group + row) * q.ld8 + ch0));
    }
  }
  if constexpr (sizeof(T) == 2) {
    if (q.q8) amax_commit(m8, q.q_next);
  }
}

// ------------------------------------------------------------------ norm + act of SMALL tensors, one kernel each way
// The low levels of the U-Net (16^3 x 256, 8^3 x 512 channels) and the last PatchGAN blocks (16^3 x 128 ... 4^3 x 512) hold
// 64 KB - 2 MB per tensor: statistics-from-the-conv-epilogue + norm_finalize + normact_fwd (forward) and reduce + finalize +
// apply (backward) are three launches of 4 - 13 us each on data that fits in L2 -- latency, not bandwidth.  Here ONE
// workgroup owns EPV channels (one 16-byte piece of every row) of ALL rows and groups: mean, then variance about that mean
// (two-pass: no sum-of-squares cancellation), then the apply pass, re-reading its L2-resident column; the groups are walked
// in order, so BatchNorm's running statistics and the affine gradients need no cross-workgroup reduction (deterministic).
struct NormSmallArgs {
  NormActArgs q;
  float eps, momentum;
  float* mean_out; float* rstd_out;                       // forward: [groups][c], saved for the backward pass
  float* running_mean; float* running_var; long long* batches_tracked; int n_real;
  float* dgamma; float* dbeta; int accumulate;            // backward: [n_affine]
};

constexpr int kSmallThreads = 512, kSmallWaves = kSmallThreads / 64;      // (1024 would cap the kernel at 128 registers: it spilled 500)
// per-thread f32 partial sums (a few rows each) -> block totals, combined in f64 in a fixed order (the three-launch path
// combines its per-block partials in f64 too: gradient sums with heavy cancellation keep their sign)
template <int EPV>
__device__ __forceinline__ void block_sum_vec(const float (&v)[EPV], double (&t)[EPV], double* red /* [kSmallWaves][EPV] */) {
#pragma unroll
  for (int j = 0; j < EPV; ++j) {
    double d = (double)v[j];
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) d += __shfl_xor(d, o, 64);
    t[j] = d;
  }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  __syncthreads();                                        // red may still be read from the previous reduction
  if (lane == 0) {
#pragma unroll
    for (int j = 0; j < EPV; ++j) red[wave * EPV + j] = t[j];
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < EPV; ++j) {
    double a = 0.0;
#pragma unroll
    for (int w = 0; w < kSmallWaves; ++w) a += red[w * EPV + j];
    t[j] = a;
  }
}

template <typename T, bool DROP>
__global__ __launch_bounds__(kSmallThreads) void normact_small_fwd_kernel(const NormSmallArgs p) {
  constexpr int EPV = Elem<T>::kPer16B;
  __shared__ double red[kSmallWaves * EPV];
  const NormActArgs& q = p.q;
  const int ch0 = blockIdx.x * EPV;
  const unsigned long long seed = DROP ? eff_seed(q.seed, q.seed_ptr) : 0ull;
  float ga[EPV], be[EPV];
#pragma unroll
  for (int j = 0; j < EPV; ++j) {
    const int ch = ch0 + j;
    ga[j] = q.gamma ? (ch < q.n_affine ? q.gamma[ch] : 0.f) : 1.f;
    be[j] = (q.beta && ch < q.n_affine) ? q.beta[ch] : 0.f;
  }
  if (p.batches_tracked && blockIdx.x == 0 && threadIdx.x == 0) p.batches_tracked[0] += q.groups;
  const double inv = 1.0 / (double)q.rows_per_group;
  for (int g = 0; g < q.groups; ++g) {
    const T* zb = reinterpret_cast<const T*>(q.z) + (long long)g * q.rows_per_group * q.ldz + ch0;
    float s[EPV], mu[EPV], rs[EPV];
    double t[EPV];
#pragma unroll
    for (int j = 0; j < EPV; ++j) s[j] = 0.f;
    // (512 threads x U rows in flight: the column is L2-resident and a pass is one latency -- with 256 threads walking it one
    //  dependent load per iteration the kernel took 24 us, longer than the three launches it replaces)
    constexpr int U = 8;
    for (long long row0 = threadIdx.x; row0 < q.rows_per_group; row0 += kSmallThreads * U) {
      uint4 raw[U];                                        // (packed while in flight: 4 registers per row, not EPV)
#pragma unroll
      for (int u = 0; u < U; ++u) if (row0 + u * kSmallThreads < q.rows_per_group) raw[u] = *reinterpret_cast<const uint4*>(zb + (row0 + u * kSmallThreads) * q.ldz);
#pragma unroll
      for (int u = 0; u < U; ++u)
        if (row0 + u * kSmallThreads < q.rows_per_group) {
          Vec16<T> v; v.from_bits(raw[u]);
#pragma unroll
          for (int j = 0; j < EPV; ++j) s[j] += v.f[j];
        }
    }
    block_sum_vec<EPV>(s, t, red);
#pragma unroll
    for (int j = 0; j < EPV; ++j) { mu[j] = (float)(t[j] * inv); s[j] = 0.f; }
    for (long long row0 = threadIdx.x; row0 < q.rows_per_group; row0 += kSmallThreads * U) {
      uint4 raw[U];
#pragma unroll
      for (int u = 0; u < U; ++u) if (row0 + u * kSmallThreads < q.rows_per_group) raw[u] = *reinterpret_cast<const uint4*>(zb + (row0 + u * kSmallThreads) * q.ldz);
#pragma unroll
      for (int u = 0; u < U; ++u)
        if (row0 + u * kSmallThreads < q.rows_per_group) {
          Vec16<T> v; v.from_bits(raw[u]);
#pragma unroll
          for (int j = 0; j < EPV; ++j) { const float d = v.f[j] - mu[j]; s[j] += d * d; }
        }
    }
    block_sum_vec<EPV>(s, t, red);
#pragma unroll
    for (int j = 0; j < EPV; ++j) rs[j] = (float)(1.0 / sqrt(t[j] * inv + (double)p.eps));
    if (threadIdx.x == 0) {
#pragma unroll
      for (int j = 0; j < EPV; ++j) {
        const int ch = ch0 + j;
        p.mean_out[(long long)g * q.c + ch] = mu[j];
        p.rstd_out[(long long)g * q.c + ch] = rs[j];
        if (p.running_mean && ch < p.n_real) {
          const double var = t[j] * inv;
          const double unb = q.rows_per_group > 1 ? var * (double)q.rows_per_group / (double)(q.rows_per_group - 1) : var;
          p.running_mean[ch] = (float)((1.0 - p.momentum) * p.running_mean[ch] + p.momentum * (double)mu[j]);
          p.running_var[ch] = (float)((1.0 - p.momentum) * p.running_var[ch] + p.momentum * unb);
        }
      }
    }
    float sc[EPV], sh[EPV];
#pragma unroll
    for (int j = 0; j < EPV; ++j) { sc[j] = ga[j] * rs[j]; sh[j] = be[j] - mu[j] * ga[j] * rs[j]; }
    T* ab = reinterpret_cast<T*>(q.a) + (long long)g * q.rows_per_group * q.lda + ch0;
    for (long long row0 = threadIdx.x; row0 < q.rows_per_group; row0 += kSmallThreads * U) {
      uint4 raw[U];
#pragma unroll
      for (int u = 0; u < U; ++u) if (row0 + u * kSmallThreads < q.rows_per_group) raw[u] = *reinterpret_cast<const uint4*>(zb + (row0 + u * kSmallThreads) * q.ldz);
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const long long row = row0 + u * kSmallThreads;
        if (row >= q.rows_per_group) continue;
        Vec16<T> v; v.from_bits(raw[u]);
        unsigned keep = 0;
        if constexpr (DROP) keep = drop_keep_mask<EPV>(seed, ((unsigned long long)g * q.rows_per_group + row) * q.c + ch0, q.thr16);
#pragma unroll
        for (int j = 0; j < EPV; ++j) {
          float t2 = v.f[j] * sc[j] + sh[j];
          if constexpr (DROP) t2 = (keep >> j) & 1u ? t2 * q.drop_scale : 0.f;
          v.f[j] = t2 > 0.f ? t2 : t2 * q.slope;
        }
        if (q.s2d_a.d) {
          long long srow; int blk, border;
          s2d_cell(q.s2d_a, (long long)g * q.rows_per_group + row, srow, blk, border);
          v.store(reinterpret_cast<T*>(q.a) + srow * q.lda + (long long)blk * q.s2d_a.cblk + ch0);
          s2d_zero_siblings<T>(reinterpret_cast<T*>(q.a), q.s2d_a, srow, blk, border, q.lda, ch0);
        } else v.store(ab + row * q.lda);
      }
    }
  }
}

template <typename T, bool DROP>
__global__ __launch_bounds__(kSmallThreads) void normact_small_bwd_kernel(const NormSmallArgs p) {
  constexpr int EPV = Elem<T>::kPer16B;
  __shared__ double red[kSmallWaves * EPV];
  const NormActArgs& q = p.q;
  const int ch0 = blockIdx.x * EPV;
  const unsigned long long seed = DROP ? eff_seed(q.seed, q.seed_ptr) : 0ull;
  const double inv = 1.0 / (double)q.rows_per_group;
  double tg[EPV], tb[EPV];
#pragma unroll
  for (int j = 0; j < EPV; ++j) { tg[j] = 0.0; tb[j] = 0.0; }
  for (int g = 0; g < q.groups; ++g) {
    BwdConst<EPV> k;
    load_bwd_const<EPV>(q, g, ch0, k);
    const T* zb = reinterpret_cast<const T*>(q.z) + (long long)g * q.rows_per_group * q.ldz + ch0;
    const T* db = reinterpret_cast<const T*>(q.da) + (long long)g * q.rows_per_group * q.ldda + ch0;
    auto load_da = [&](long long row) {
      if (q.s2d_da.d) return *reinterpret_cast<const uint4*>(reinterpret_cast<const T*>(q.da) + s2d_offset(q.s2d_da, (long long)g * q.rows_per_group + row, q.ldda) + ch0);
      return *reinterpret_cast<const uint4*>(db + row * q.ldda);
    };
    float s0[EPV], s1[EPV];
#pragma unroll
    for (int j = 0; j < EPV; ++j) { s0[j] = 0.f; s1[j] = 0.f; }
    constexpr int U = 2;                                   // (z and da rows in flight per thread: see the forward kernel)
    for (long long row0 = threadIdx.x; row0 < q.rows_per_group; row0 += kSmallThreads * U) {
      uint4 zr[U], dr[U];                                  // (packed while in flight)
#pragma unroll
      for (int u = 0; u < U; ++u)
        if (row0 + u * kSmallThreads < q.rows_per_group) { zr[u] = *reinterpret_cast<const uint4*>(zb + (row0 + u * kSmallThreads) * q.ldz); dr[u] = load_da(row0 + u * kSmallThreads); }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const long long row = row0 + u * kSmallThreads;
        if (row >= q.rows_per_group) continue;
        Vec16<T> zv, dv; zv.from_bits(zr[u]); dv.from_bits(dr[u]);
        unsigned keep = 0;
        if constexpr (DROP) keep = drop_keep_mask<EPV>(seed, ((unsigned long long)g * q.rows_per_group + row) * q.c + ch0, q.thr16);
#pragma unroll
        for (int j = 0; j < EPV; ++j) {
          float gv, xh;
          bwd_elem<DROP>(q, (keep >> j) & 1u, k.mu[j], k.rs[j], k.ga[j], k.be[j], zv.f[j], dv.f[j], gv, xh);
          s0[j] += gv;
          s1[j] += gv * xh;
        }
      }
    }
    double t0[EPV], t1[EPV];
    block_sum_vec<EPV>(s0, t0, red);
    block_sum_vec<EPV>(s1, t1, red);
    float kk[EPV], m0[EPV], m1[EPV];
    const bool sub = q.mean && q.batch_stats;
#pragma unroll
    for (int j = 0; j < EPV; ++j) {
      tb[j] += t0[j]; tg[j] += t1[j];
      kk[j] = k.ga[j] * k.rs[j];
      m0[j] = sub ? (float)(t0[j] * inv) : 0.f;
      m1[j] = sub ? (float)(t1[j] * inv) : 0.f;
    }
    T* ob = reinterpret_cast<T*>(q.dz) + (long long)g * q.rows_per_group * q.lddz + ch0;
    for (long long row0 = threadIdx.x; row0 < q.rows_per_group; row0 += kSmallThreads * U) {
      uint4 zr[U], dr[U];
#pragma unroll
      for (int u = 0; u < U; ++u)
        if (row0 + u * kSmallThreads < q.rows_per_group) { zr[u] = *reinterpret_cast<const uint4*>(zb + (row0 + u * kSmallThreads) * q.ldz); dr[u] = load_da(row0 + u * kSmallThreads); }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const long long row = row0 + u * kSmallThreads;
        if (row >= q.rows_per_group) continue;
        Vec16<T> zv, dv; zv.from_bits(zr[u]); dv.from_bits(dr[u]);
        unsigned keep = 0;
        if constexpr (DROP) keep = drop_keep_mask<EPV>(seed, ((unsigned long long)g * q.rows_per_group + row) * q.c + ch0, q.thr16);
#pragma unroll
        for (int j = 0; j < EPV; ++j) {
          float gv, xh;
          bwd_elem<DROP>(q, (keep >> j) & 1u, k.mu[j], k.rs[j], k.ga[j], k.be[j], zv.f[j], dv.f[j], gv, xh);
          zv.f[j] = kk[j] * (gv - m0[j] - xh * m1[j]);
        }
        zv.store(ob + row * q.lddz);
      }
    }
  }
  if (threadIdx.x == 0) {
#pragma unroll
    for (int j = 0; j < EPV; ++j) {
      const int ch = ch0 + j;
      if (ch < q.n_affine) {
        if (p.dgamma) p.dgamma[ch] = p.accumulate ? p.dgamma[ch] + (float)tg[j] : (float)tg[j];
        if (p.dbeta) p.dbeta[ch] = p.accumulate ? p.dbeta[ch] + (float)tb[j] : (float)tb[j];
      }
    }
  }
}

// Register-resident forms (round 4, third session).  The kernels above are chains of dependent L2 round trips: the forward
// kernel reads its column three times (mean, variance, apply), the backward kernel twice in four dependent iterations, and
// the statistic groups of a stacked PatchGAN pair are walked one after the other -- 14 - 31 us for 64 KB per workgroup.
// Every eligible tensor is at most 16 rows per thread (<= 512 K elements of >= 64 channels over 512 threads), so the rows
// of GC groups x S slots stay in registers as the packed 16-byte pieces: ONE load latency per launch for GC groups, the
// passes after it are arithmetic.  Row -> thread assignment, accumulation order and the f64 block sums are those of the
// kernels above (bit-identical results); the host picks S = 1 (<= 512 rows per group) or 8 (<= 4096) and GC = 2 for an even
// number of groups.
constexpr int kSmallResMaxGroups = 16;
// Block sums of NV per-thread f32 partials, combined in f64 in a fixed order, through an LDS transpose: block_sum_vec's 6-step
// f64 butterfly per value is 12 ds_bpermute + 8 ds_read_b64 per value and THREAD -- 320 LDS instructions per wave for 16
// values, and the LDS pipe of one CU serves all 8 waves: that, not the loads, was most of these kernels' time (two groups per
// chunk cost 8 us more than one).  Here every thread writes its partials ([value][thread]: conflict-free), TPV threads per
// value add 512 / TPV of them each (strided by TPV: conflict-free) and finish with a log2(TPV)-step butterfly on ONE value;
// ~4 LDS instructions per value and thread.  CH values at a time (part: CH x 512 floats of LDS).
template <int NV, int CH>
__device__ __forceinline__ void block_sum_lds(const float (&v)[NV], double (&t)[NV], float* part /* [CH][kSmallThreads] */,
                                              double* total /* [CH] */) {
  static_assert(NV % CH == 0, "chunk");
  constexpr int TPV = (kSmallThreads / CH) < 64 ? (kSmallThreads / CH) : 64, CNT = kSmallThreads / TPV;
  const int tid = threadIdx.x;
#pragma unroll
  for (int c0 = 0; c0 < NV; c0 += CH) {
    __syncthreads();                                       // part / total may still be read from the previous round
#pragma unroll
    for (int i = 0; i < CH; ++i) part[i * kSmallThreads + tid] = v[c0 + i];
    __syncthreads();
    if (tid < CH * TPV) {
      const int val = tid / TPV, sg = tid % TPV;
      double a = 0.0;
#pragma unroll
      for (int i = 0; i < CNT; ++i) a += (double)part[val * kSmallThreads + i * TPV + sg];
#pragma unroll
      for (int o = 1; o < TPV; o <<= 1) a += __shfl_xor(a, o, 64);
      if (sg == 0) total[val] = a;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < CH; ++i) t[c0 + i] = total[i];
  }
}
template <typename T, bool DROP, int S, int GC>
__global__ __launch_bounds__(kSmallThreads) void normact_small_res_fwd_kernel(const NormSmallArgs p) {
  constexpr int EPV = Elem<T>::kPer16B, NV = GC * EPV;
  constexpr int CH = NV < 16 ? NV : 16;
  __shared__ float part[CH * kSmallThreads];
  __shared__ double total[CH];
  const NormActArgs& q = p.q;
  const int ch0 = blockIdx.x * EPV;
  const unsigned long long seed = DROP ? eff_seed(q.seed, q.seed_ptr) : 0ull;
  float ga[EPV], be[EPV];
#pragma unroll
  for (int j = 0; j < EPV; ++j) {
    const int ch = ch0 + j;
    ga[j] = q.gamma ? (ch < q.n_affine ? q.gamma[ch] : 0.f) : 1.f;
    be[j] = (q.beta && ch < q.n_affine) ? q.beta[ch] : 0.f;
  }
  if (p.batches_tracked && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) p.batches_tracked[0] += q.groups;
  const double inv = 1.0 / (double)q.rows_per_group;
  const int rows = (int)q.rows_per_group;
  // blockIdx.y = a chunk of groups when they are independent (no running statistics: the host sets gridDim.y = groups / GC)
  for (int g0 = blockIdx.y * GC; g0 < q.groups; g0 += gridDim.y * GC) {
    uint4 raw[GC][S];
#pragma unroll
    for (int c = 0; c < GC; ++c) {
      const T* zb = reinterpret_cast<const T*>(q.z) + (long long)(g0 + c) * q.rows_per_group * q.ldz + ch0;
#pragma unroll
      for (int u = 0; u < S; ++u) {
        const int row = threadIdx.x + u * kSmallThreads;
        raw[c][u] = row < rows ? *reinterpret_cast<const uint4*>(zb + (long long)row * q.ldz) : make_uint4(0, 0, 0, 0);
      }
    }
    float s[NV], mu[NV], rs[NV];
    double t[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) s[i] = 0.f;
#pragma unroll
    for (int c = 0; c < GC; ++c)
#pragma unroll
      for (int u = 0; u < S; ++u)
        if ((int)threadIdx.x + u * kSmallThreads < rows) {
          Vec16<T> v; v.from_bits(raw[c][u]);
#pragma unroll
          for (int j = 0; j < EPV; ++j) s[c * EPV + j] += v.f[j];
        }
    block_sum_lds<NV, CH>(s, t, part, total);
#pragma unroll
    for (int i = 0; i < NV; ++i) { mu[i] = (float)(t[i] * inv); s[i] = 0.f; }
#pragma unroll
    for (int c = 0; c < GC; ++c)
#pragma unroll
      for (int u = 0; u < S; ++u)
        if ((int)threadIdx.x + u * kSmallThreads < rows) {
          Vec16<T> v; v.from_bits(raw[c][u]);
#pragma unroll
          for (int j = 0; j < EPV; ++j) { const float d = v.f[j] - mu[c * EPV + j]; s[c * EPV + j] += d * d; }
        }
    block_sum_lds<NV, CH>(s, t, part, total);
#pragma unroll
    for (int i = 0; i < NV; ++i) rs[i] = (float)(1.0 / sqrt(t[i] * inv + (double)p.eps));
    if (threadIdx.x == 0) {
#pragma unroll
      for (int c = 0; c < GC; ++c) {                       // (groups in order: the running statistics are a recurrence)
        const int g = g0 + c;
#pragma unroll
        for (int j = 0; j < EPV; ++j) {
          const int ch = ch0 + j;
          p.mean_out[(long long)g * q.c + ch] = mu[c * EPV + j];
          p.rstd_out[(long long)g * q.c + ch] = rs[c * EPV + j];
          if (p.running_mean && ch < p.n_real) {
            const double var = t[c * EPV + j] * inv;
            const double unb = q.rows_per_group > 1 ? var * (double)q.rows_per_group / (double)(q.rows_per_group - 1) : var;
            p.running_mean[ch] = (float)((1.0 - p.momentum) * p.running_mean[ch] + p.momentum * (double)mu[c * EPV + j]);
            p.running_var[ch] = (float)((1.0 - p.momentum) * p.running_var[ch] + p.momentum * unb);
          }
        }
      }
    }
#pragma unroll
    for (int c = 0; c < GC; ++c) {
      const int g = g0 + c;
      float sc[EPV], sh[EPV];
#pragma unroll
      for (int j = 0; j < EPV; ++j) { sc[j] = ga[j] * rs[c * EPV + j]; sh[j] = be[j] - mu[c * EPV + j] * ga[j] * rs[c * EPV + j]; }
      T* ab = reinterpret_cast<T*>(q.a) + (long long)g * q.rows_per_group * q.lda + ch0;
#pragma unroll
      for (int u = 0; u < S; ++u) {
        const long long row = threadIdx.x + u * kSmallThreads;
        if (row >= q.rows_per_group) continue;
        Vec16<T> v; v.from_bits(raw[c][u]);
        unsigned keep = 0;
        if constexpr (DROP) keep = drop_keep_mask<EPV>(seed, ((unsigned long long)g * q.rows_per_group + row) * q.c + ch0, q.thr16);
#pragma unroll
        for (int j = 0; j < EPV; ++j) {
          float t2 = v.f[j] * sc[j] + sh[j];
          if constexpr (DROP) t2 = (keep >> j) & 1u ? t2 * q.drop_scale : 0.f;
          v.f[j] = t2 > 0.f ? t2 : t2 * q.slope;
        }
        if (q.s2d_a.d) {
          long long srow; int blk, border;
          s2d_cell(q.s2d_a, (long long)g * q.rows_per_group + row, srow, blk, border);
          v.store(reinterpret_cast<T*>(q.a) + srow * q.lda + (long long)blk * q.s2d_a.cblk + ch0);
          s2d_zero_siblings<T>(reinterpret_cast<T*>(q.a), q.s2d_a, srow, blk, border, q.lda, ch0);
        } else v.store(ab + row * q.lda);
      }
    }
  }
}

template <typename T, bool DROP, int S, int GC>
__global__ __launch_bounds__(kSmallThreads) void normact_small_res_bwd_kernel(const NormSmallArgs p) {
  constexpr int EPV = Elem<T>::kPer16B, NV = 2 * GC * EPV;     // sum g | sum g * xhat, per group of the chunk
  constexpr int CH = NV < 16 ? NV : 16;
  __shared__ float part[CH * kSmallThreads];
  __shared__ double total[CH];
  const NormActArgs& q = p.q;
  const int ch0 = blockIdx.x * EPV;
  const unsigned long long seed = DROP ? eff_seed(q.seed, q.seed_ptr) : 0ull;
  const double inv = 1.0 / (double)q.rows_per_group;
  const int rows = (int)q.rows_per_group;
  constexpr int NST = S > 1 ? 3 : 0;                       // S = 8: the last 3 da rows wait in LDS between the passes (each thread its
  __shared__ uint4 stash[NST ? NST * kSmallThreads : 1];   //  own 16-byte slots, conflict-free; as registers next to z they spilled 13 - 70)
  __shared__ double tot[kSmallResMaxGroups * 2 * EPV];     // per group (sum g | sum g * xhat), written by thread 0: the sums over the
                                                           //  groups are formed at the end (as 32 registers of every thread they spilled)
  // blockIdx.y = a chunk of groups when the host gave scratch for the per-group sums (q.part: [groups][2][c] doubles); the affine
  // gradients are then summed over the groups, in the same order and in f64, by normact_small_affine_kernel
  double* const gpart = reinterpret_cast<double*>(q.part);
  for (int g0 = blockIdx.y * GC; g0 < q.groups; g0 += gridDim.y * GC) {
    BwdConst<EPV> k[GC];
    uint4 zr[GC][S], dr[GC][S];
#pragma unroll
    for (int c = 0; c < GC; ++c) {
      const int g = g0 + c;
      load_bwd_const<EPV>(q, g, ch0, k[c]);
      const T* zb = reinterpret_cast<const T*>(q.z) + (long long)g * q.rows_per_group * q.ldz + ch0;
      const T* db = reinterpret_cast<const T*>(q.da) + (long long)g * q.rows_per_group * q.ldda + ch0;
#pragma unroll
      for (int u = 0; u < S; ++u) {
        const int row = threadIdx.x + u * kSmallThreads;
        zr[c][u] = dr[c][u] = make_uint4(0, 0, 0, 0);
        if (row < rows) {
          zr[c][u] = *reinterpret_cast<const uint4*>(zb + (long long)row * q.ldz);
          dr[c][u] = q.s2d_da.d ? *reinterpret_cast<const uint4*>(reinterpret_cast<const T*>(q.da) + s2d_offset(q.s2d_da, (long long)g * q.rows_per_group + row, q.ldda) + ch0)
                                : *reinterpret_cast<const uint4*>(db + (long long)row * q.ldda);
        }
      }
    }
    float s[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) s[i] = 0.f;
#pragma unroll
    for (int c = 0; c < GC; ++c)
#pragma unroll
      for (int u = 0; u < S; ++u) {
        const long long row = threadIdx.x + u * kSmallThreads;
        if (row >= q.rows_per_group) continue;
        Vec16<T> zv, dv; zv.from_bits(zr[c][u]); dv.from_bits(dr[c][u]);
        unsigned keep = 0;
        if constexpr (DROP) keep = drop_keep_mask<EPV>(seed, ((unsigned long long)(g0 + c) * q.rows_per_group + row) * q.c + ch0, q.thr16);
#pragma unroll
        for (int j = 0; j < EPV; ++j) {
          float gv, xh;
          bwd_elem<DROP>(q, (keep >> j) & 1u, k[c].mu[j], k[c].rs[j], k[c].ga[j], k[c].be[j], zv.f[j], dv.f[j], gv, xh);
          s[(2 * c) * EPV + j] += gv;
          s[(2 * c + 1) * EPV + j] += gv * xh;
        }
        if (u >= S - NST) stash[(u - (S - NST)) * kSmallThreads + threadIdx.x] = dr[c][u];
      }
    double t[NV];
    block_sum_lds<NV, CH>(s, t, part, total);
    if (threadIdx.x == 0) {
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        if (gpart) gpart[((long long)g0 * 2 + i / EPV) * q.c + ch0 + i % EPV] = t[i];
        else tot[g0 * 2 * EPV + i] = t[i];
      }
    }
    const bool sub = q.mean && q.batch_stats;
#pragma unroll
    for (int c = 0; c < GC; ++c) {
      const int g = g0 + c;
      float kk[EPV], m0[EPV], m1[EPV];
#pragma unroll
      for (int j = 0; j < EPV; ++j) {
        const double t0 = t[(2 * c) * EPV + j], t1 = t[(2 * c + 1) * EPV + j];
        kk[j] = k[c].ga[j] * k[c].rs[j];
        m0[j] = sub ? (float)(t0 * inv) : 0.f;
        m1[j] = sub ? (float)(t1 * inv) : 0.f;
      }
      T* ob = reinterpret_cast<T*>(q.dz) + (long long)g * q.rows_per_group * q.lddz + ch0;
#pragma unroll
      for (int u = 0; u < S; ++u) {
        const long long row = threadIdx.x + u * kSmallThreads;
        if (row >= q.rows_per_group) continue;
        Vec16<T> zv, dv; zv.from_bits(zr[c][u]);
        if (u >= S - NST) dv.from_bits(stash[(u - (S - NST)) * kSmallThreads + threadIdx.x]); else dv.from_bits(dr[c][u]);
        unsigned keep = 0;
        if constexpr (DROP) keep = drop_keep_mask<EPV>(seed, ((unsigned long long)g * q.rows_per_group + row) * q.c + ch0, q.thr16);
#pragma unroll
        for (int j = 0; j < EPV; ++j) {
          float gv, xh;
          bwd_elem<DROP>(q, (keep >> j) & 1u, k[c].mu[j], k[c].rs[j], k[c].ga[j], k[c].be[j], zv.f[j], dv.f[j], gv, xh);
          zv.f[j] = kk[j] * (gv - m0[j] - xh * m1[j]);
        }
        zv.store(ob + row * q.lddz);
      }
    }
  }
  if (threadIdx.x == 0 && !gpart) {
#pragma unroll
    for (int j = 0; j < EPV; ++j) {
      const int ch = ch0 + j;
      if (ch < q.n_affine) {
        double tb = 0.0, tg = 0.0;
        for (int g = 0; g < q.groups; ++g) { tb += tot[(2 * g) * EPV + j]; tg += tot[(2 * g + 1) * EPV + j]; }
        if (p.dgamma) p.dgamma[ch] = p.accumulate ? p.dgamma[ch] + (float)tg : (float)tg;
        if (p.dbeta) p.dbeta[ch] = p.accumulate ? p.dbeta[ch] + (float)tb : (float)tb;
      }
    }
  }
}

// the affine gradients of the form above with blockIdx.y = chunk of groups: sum over the groups in order, in f64 (what thread 0 of
// the single-workgroup form does from its LDS totals: bit-identical)
__global__ __launch_bounds__(256) void normact_small_affine_kernel(const double* __restrict__ gpart, int groups, int c, int n_affine,
                                                                   float* dgamma, float* dbeta, int accumulate) {
  const int ch = blockIdx.x * 256 + threadIdx.x;
  if (ch >= n_affine) return;
  double tb = 0.0, tg = 0.0;
  for (int g = 0; g < groups; ++g) { tb += gpart[((long long)g * 2 + 0) * c + ch]; tg += gpart[((long long)g * 2 + 1) * c + ch]; }
  if (dgamma) dgamma[ch] = accumulate ? dgamma[ch] + (float)tg : (float)tg;
  if (dbeta) dbeta[ch] = accumulate ? dbeta[ch] + (float)tb : (float)tb;
}

// S = 1 / 8 row slots per thread and group (s_slots), GC = 1 / 2 groups per chunk (chunk; two groups of 8 slots spilled 100 - 900 registers)
#define MI355_SMALL_RES_LAUNCH(KERNEL, T, DROP, GC8)                                                                \
  do {                                                                                                              \
    if (s_slots == 1) {                                                                                             \
      if (chunk == 2) KERNEL<T, DROP, 1, 2><<<grid, dim3(kSmallThreads), 0, (hipStream_t)stream>>>(p);              \
      else KERNEL<T, DROP, 1, 1><<<grid, dim3(kSmallThreads), 0, (hipStream_t)stream>>>(p);                         \
    } else {                                                                                                        \
      if (chunk == 2) KERNEL<T, DROP, 8, GC8><<<grid, dim3(kSmallThreads), 0, (hipStream_t)stream>>>(p);            \
      else KERNEL<T, DROP, 8, 1><<<grid, dim3(kSmallThreads), 0, (hipStream_t)stream>>>(p);                         \
    }                                                                                                               \
  } while (0)

// ------------------------------------------------------------------ max-pool 2x2x2
template <typename T>
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const T* __restrict__ x, int ldx, T* __restrict__ y,
                                                           int ldy, int c, int d, int h, int w, long long total,
                                                           uint8_t* __restrict__ widx) {
  constexpr int EPV = Elem<T>::kPer16B;
  const int lpr = c / EPV;
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int piece = (int)(idx % lpr);
  long long o = idx / lpr;
  const int od_ = d / 2, oh_ = h / 2, ow_ = w / 2;
  const int ow = (int)(o % ow_); long long t = o / ow_;
  const int oh = (int)(t % oh_); t /= oh_;
  const int od = (int)(t % od_); const int n = (int)(t / od_);
  Vec16<T> m;
  bool firstv = true;
  // window position (kd, kh, kw as 3 bits) of the element the BACKWARD kernel routes the gradient to: the first one in scan
  // order that equals the maximum or is NaN (maxpool_bwd_kernel's `hit`), i.e. the first NaN if the window holds one, else the
  // first occurrence of the maximum
  unsigned long long where = 0;
  unsigned nan_seen = 0;
#pragma unroll
  for (int kd = 0; kd < 2; ++kd)
#pragma unroll
    for (int kh = 0; kh < 2; ++kh)
#pragma unroll
      for (int kw = 0; kw < 2; ++kw) {
        const long long vox = (((long long)n * d + 2 * od + kd) * h + 2 * oh + kh) * w + 2 * ow + kw;
        const unsigned long long kk = (unsigned long long)(kd * 4 + kh * 2 + kw);
        Vec16<T> v;
        v.load(x + vox * ldx + piece * EPV);
        if (firstv) {
          m = v; firstv = false;
#pragma unroll
          for (int j = 0; j < EPV; ++j) nan_seen |= (v.f[j] != v.f[j] ? 1u : 0u) << j;
        } else {
#pragma unroll
          for (int j = 0; j < EPV; ++j) {
            const bool isnan_ = v.f[j] != v.f[j];
            const bool take = v.f[j] > m.f[j] || isnan_;
            m.f[j] = take ? v.f[j] : m.f[j];
            const bool mark = ((nan_seen >> j) & 1u) ? false : take;          // after the first NaN the position stays
            where = mark ? ((where & ~(0xffull << (8 * j))) | (kk << (8 * j))) : where;
            nan_seen |= (isnan_ ? 1u : 0u) << j;
          }
        }
      }
  m.store(y + o * ldy + piece * EPV);
  if (widx) {
    if constexpr (EPV == 8) *reinterpret_cast<unsigned long long*>(widx + o * c + piece * EPV) = where;
    else *reinterpret_cast<unsigned*>(widx + o * c + piece * EPV) = (unsigned)where;
  }
}

// ODD: some extent is odd -- MaxPool3d(2) floors, the last plane / row / column belongs to no window: its gradient is zero
// (+ add); the thread of the last window along such an axis writes it.
template <typename T, bool ODD>
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const T* __restrict__ x, int ldx, const T* __restrict__ y,
                                                           int ldy, const T* __restrict__ dy, int lddy,
                                                           T* __restrict__ dx, int lddx, int c, int d, int h, int w,
                                                           long long total, const T* __restrict__ add, int ldadd) {
  constexpr int EPV = Elem<T>::kPer16B;
  const int lpr = c / EPV;
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int piece = (int)(idx % lpr);
  long long o = idx / lpr;
  const int od_ = d / 2, oh_ = h / 2, ow_ = w / 2;
  const int ow = (int)(o % ow_); long long t = o / ow_;
  const int oh = (int)(t % oh_); t /= oh_;
  const int od = (int)(t % od_); const int n = (int)(t / od_);
  Vec16<T> m, g;
  m.load(y + o * ldy + piece * EPV);
  g.load(dy + o * lddy + piece * EPV);
  bool taken[EPV];
#pragma unroll
  for (int j = 0; j < EPV; ++j) taken[j] = false;
  constexpr int KN = ODD ? 3 : 2;
  const bool xd = ODD && od == od_ - 1 && (d & 1), xh = ODD && oh == oh_ - 1 && (h & 1), xw = ODD && ow == ow_ - 1 && (w & 1);
#pragma unroll
  for (int kd = 0; kd < KN; ++kd)
#pragma unroll
    for (int kh = 0; kh < KN; ++kh)
#pragma unroll
      for (int kw = 0; kw < KN; ++kw) {
        const bool inwin = kd < 2 && kh < 2 && kw < 2;
        if (!inwin && ((kd == 2 && !xd) || (kh == 2 && !xh) || (kw == 2 && !xw))) continue;
        const long long vox = (((long long)n * d + 2 * od + kd) * h + 2 * oh + kh) * w + 2 * ow + kw;
        Vec16<T> v, outv;
        if (inwin) v.load(x + vox * ldx + piece * EPV);
#pragma unroll
        for (int j = 0; j < EPV; ++j) {
          const bool hit = inwin && !taken[j] && (v.f[j] == m.f[j] || v.f[j] != v.f[j]);
          outv.f[j] = hit ? g.f[j] : 0.f;
          taken[j] = taken[j] || hit;
        }
        if (add) {            // second gradient of the pooled tensor (its skip-connection use): summed here in f32
          Vec16<T> s2;
          s2.load(add + vox * ldadd + piece * EPV);
#pragma unroll
          for (int j = 0; j < EPV; ++j) outv.f[j] += s2.f[j];
        }
        outv.store(dx + vox * lddx + piece * EPV);
      }
}

// ------------------------------------------------------------------ L1 loss
constexpr int kL1PerBlock = 256 * 16;
__global__ __launch_bounds__(256) void l1_partial_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                         long long count, float* __restrict__ partials) {
  __shared__ float red[256];
  const long long base = (long long)blockIdx.x * kL1PerBlock;
  float s = 0.f;
#pragma unroll 4
  for (int i = 0; i < 16; ++i) {
    const long long idx = base + i * 256 + threadIdx.x;
    if (idx < count) s += fabsf(a[idx] - b[idx]);
  }
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) partials[blockIdx.x] = red[0];
}
__global__ __launch_bounds__(256) void l1_final_kernel(const float* __restrict__ partials, int n, long long count,
                                                       float* out) {
  __shared__ double red[256];
  double s = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) s += (double)partials[i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = (float)(red[0] / (double)count);
}
__global__ __launch_bounds__(256) void l1_bwd_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                     long long count, const float* __restrict__ gscale,
                                                     float* __restrict__ da) {
  const float k = gscale[0] / (float)count;
  const long long stride = (long long)gridDim.x * 256;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < count; i += stride) {
    const float dlt = a[i] - b[i];
    da[i] = dlt > 0.f ? k : (dlt < 0.f ? -k : 0.f);
  }
}

// ------------------------------------------------------------------ GAN loss heads (tiny logit maps: one workgroup)
// BCEWithLogits element (torch: (1 - t) x - log_sigmoid(x), log_sigmoid(x) = min(x, 0) - log1p(exp(-|x|)))
__device__ __forceinline__ float bce_logits_elem(float x, float t) {
  return (1.f - t) * x - (fminf(x, 0.f) - log1pf(expf(-fabsf(x))));
}
__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }
// block sum in f64 (256 threads), result valid in thread 0
__device__ __forceinline__ double block_sum_256(double s, double* red) {
  __syncthreads();
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  return red[0];
}
// _gen_step (src/model.py:126-137): out = {L1, recon = L1 / divisor * factor, adv = mean BCE(logits, 1), adv + recon}
__global__ __launch_bounds__(256) void gan_gen_loss_fwd_kernel(const float* __restrict__ logits, int n, const float* __restrict__ partials,
                                                               int np, long long count, float divisor, float factor, float* __restrict__ out) {
  __shared__ double red[256];
  double s = 0.0;
  for (int i = threadIdx.x; i < np; i += 256) s += (double)partials[i];
  const double l1sum = block_sum_256(s, red);
  s = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) s += (double)bce_logits_elem(logits[i], 1.f);
  const double bsum = block_sum_256(s, red);
  if (threadIdx.x == 0) {
    const float l1 = (float)(l1sum / (double)count), adv = (float)(bsum / (double)n);
    const float recon = l1 / divisor * factor;
    out[0] = l1; out[1] = recon; out[2] = adv; out[3] = adv + recon;
  }
}
// upstream[0] = d loss / d out[3]:  dlogits = upstream (sigmoid(x) - 1) / n,  l1_gscale[0] = upstream * factor / divisor
__global__ __launch_bounds__(256) void gan_gen_loss_bwd_kernel(const float* __restrict__ logits, int n, const float* __restrict__ upstream,
                                                               float divisor, float factor, float* __restrict__ dlogits,
                                                               float* __restrict__ l1_gscale) {
  const float g = upstream[0];
  for (int i = threadIdx.x; i < n; i += 256) dlogits[i] = (sigmoidf_(logits[i]) - 1.f) * g / (float)n;
  if (threadIdx.x == 0) l1_gscale[0] = g * factor / divisor;
}
// _discr_step (src/model.py:183-193): out[0] = (mean BCE(real, 1) + mean BCE(fake, 0)) / 2
__global__ __launch_bounds__(256) void gan_discr_loss_fwd_kernel(const float* __restrict__ fake, int n0, const float* __restrict__ real,
                                                                 int n1, float* __restrict__ out) {
  __shared__ double red[256];
  double s = 0.0;
  for (int i = threadIdx.x; i < n0; i += 256) s += (double)bce_logits_elem(fake[i], 0.f);
  const double s0 = block_sum_256(s, red);
  s = 0.0;
  for (int i = threadIdx.x; i < n1; i += 256) s += (double)bce_logits_elem(real[i], 1.f);
  const double s1 = block_sum_256(s, red);
  if (threadIdx.x == 0) out[0] = ((float)(s1 / (double)n1) + (float)(s0 / (double)n0)) / 2.f;
}
__global__ __launch_bounds__(256) void gan_discr_loss_bwd_kernel(const float* __restrict__ fake, int n0, const float* __restrict__ real,
                                                                 int n1, const float* __restrict__ upstream, float* __restrict__ dfake,
                                                                 float* __restrict__ dreal) {
  const float g = upstream[0] / 2.f;
  for (int i = threadIdx.x; i < n0; i += 256) dfake[i] = sigmoidf_(fake[i]) * g / (float)n0;
  for (int i = threadIdx.x; i < n1; i += 256) dreal[i] = (sigmoidf_(real[i]) - 1.f) * g / (float)n1;
}

// ------------------------------------------------------------------ AdamW (multi-tensor)
// Tensor pointers travel BY VALUE in the kernel arguments (chunks of kAdamChunk tensors), so the
// launch needs no device-side table and can be captured into a hipGraph; the step count is read
// from device memory when given (a captured launch then advances with its counter).
constexpr int kAdamChunk = 64;      // (2.5 KB of kernel arguments; 24 cost the generator 5 launches per update)
struct AdamChunk {
  float* p[kAdamChunk]; const float* g[kAdamChunk]; float* m[kAdamChunk]; float* v[kAdamChunk];
  long long n[kAdamChunk];
};
__global__ __launch_bounds__(256) void adamw_kernel(const AdamChunk c, float lr, double beta1, double beta2, float eps,
                                                    float wd, const long long* step_dev, long long step_host) {
  const int t = blockIdx.y;
  const long long nel = c.n[t];
  float* __restrict__ p = c.p[t];
  const float* __restrict__ g = c.g[t];
  float* __restrict__ m = c.m[t];
  float* __restrict__ v = c.v[t];
  const double step = (double)(step_dev ? step_dev[0] : step_host);
  const float bc1 = (float)(1.0 - pow(beta1, step));
  const float rsqrt_bc2 = (float)(1.0 / sqrt(1.0 - pow(beta2, step)));
  const float b1 = (float)beta1, b2 = (float)beta2;
  const float step_size = lr / bc1;
  const long long stride = (long long)gridDim.x * 256;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < nel; i += stride) {
#ifdef ADAM_NT      // (gradients and moments are touched once per step: keep them out of the caches the next kernels live off)
    const float gi = __builtin_nontemporal_load(g + i);
    float pi = p[i] * (1.f - lr * wd);
    const float mi = b1 * __builtin_nontemporal_load(m + i) + (1.f - b1) * gi;
    const float vi = b2 * __builtin_nontemporal_load(v + i) + (1.f - b2) * gi * gi;
    const float denom = sqrtf(vi) * rsqrt_bc2 + eps;
    pi -= step_size * (mi / denom);
    p[i] = pi; __builtin_nontemporal_store(mi, m + i); __builtin_nontemporal_store(vi, v + i);
#else
    const float gi = g[i];
    float pi = p[i] * (1.f - lr * wd);
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    const float denom = sqrtf(vi) * rsqrt_bc2 + eps;
    pi -= step_size * (mi / denom);
    p[i] = pi; m[i] = mi; v[i] = vi;
#endif
  }
}

// ------------------------------------------------------------------ MFMA layout probe
__global__ void mfma_selftest_kernel(float* out_f32, float* out_bf16) {
  const int lane = threadIdx.x & 63;
  const int r = lane & 31, h = lane >> 5;
  // A[i][k] = i + 1 (k = 0 only), B[k][j] = 100 * (j + 1) (k = 0 only)  =>  D[i][j] = (i+1)*100*(j+1)
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(h == 0 ? (float)(r + 1) : 0.f, h == 0 ? 100.f * (r + 1) : 0.f, acc, 0, 0, 0);
#pragma unroll
  for (int i = 0; i < 16; ++i) out_f32[acc_row(i, h) * 32 + r] = acc[i];
  // bf16: A[i][k] = (i+1) at k == 3, B[k][j] = (j+1) at k == 3 (k = 8h + e -> h = 0, e = 3), plus
  // A[i][k=12] = 1, B[12][j] = 0.5 (h = 1, e = 4)  =>  D[i][j] = (i+1)(j+1) + 0.5
  Frag<bf16_t> fa, fb;
  uint16_t ea[8] = {0, 0, 0, 0, 0, 0, 0, 0}, eb[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (h == 0) { ea[3] = f32_to_bf16_bits((float)(r + 1)); eb[3] = f32_to_bf16_bits((float)(r + 1)); }
  else { ea[4] = f32_to_bf16_bits(1.f); eb[4] = f32_to_bf16_bits(0.5f); }
  fa.v = make_uint4(ea[0] | (ea[1] << 16), ea[2] | (ea[3] << 16), ea[4] | (ea[5] << 16), ea[6] | (ea[7] << 16));
  fb.v = make_uint4(eb[0] | (eb[1] << 16), eb[2] | (eb[3] << 16), eb[4] | (eb[5] << 16), eb[6] | (eb[7] << 16));
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  mma16(fa, fb, acc);
#pragma unroll
  for (int i = 0; i < 16; ++i) out_bf16[acc_row(i, h) * 32 + r] = acc[i];
}

template <typename T> const T* cp(const void* p) { return reinterpret_cast<const T*>(p); }
template <typename T> T* mp(void* p) { return reinterpret_cast<T*>(p); }

int check_rows(int c, int ld, int dtype, const char* who) {
  const int epv = dtype == MI355_DT_F32 ? 4 : 8;
  MI355_REQUIRE(dtype == MI355_DT_F32 || dtype == MI355_DT_BF16, "%s: bad dtype", who);
  MI355_REQUIRE(c > 0 && c % 16 == 0 && c <= 1024, "%s: channels must be a multiple of 16 and <= 1024 (c=%d)", who, c);
  MI355_REQUIRE(ld >= c && ld % epv == 0, "%s: ld=%d must be >= c and keep rows 16-byte aligned", who, ld);
  MI355_REQUIRE(c / epv <= 256, "%s: too many channels", who);
  return MI355_OK;
}

}  // namespace

extern "C" {

static int pack_impl(const float* src, void* dst, int32_t n, int32_t c, int64_t v, int32_t ld, int32_t coff,
                     int32_t zero_to, int32_t dtype, S2D q, void* stream, const float* src1 = nullptr, int32_t c1 = 0) {
  MI355_REQUIRE(src && dst && n > 0 && c > 0 && v > 0 && c1 >= 0 && (c1 == 0 || src1), "pack: bad argument");
  const int epv = dtype == MI355_DT_F32 ? 4 : 8;
  MI355_REQUIRE(dtype == MI355_DT_F32 || dtype == MI355_DT_BF16, "pack: bad dtype");
  MI355_REQUIRE(coff % epv == 0 && zero_to <= ld && (zero_to - coff) % epv == 0 && zero_to - coff >= c + c1 && ld % epv == 0,
                "pack: channel window [%d,%d) of ld %d must be 16-byte aligned and hold c=%d", coff, zero_to, ld, c);
  const int pieces = (zero_to - coff) / epv;
  if (zero_to - coff <= 64 && v >= 4096) {                // LDS-tiled form: whole-line reads, whole-row writes
    const int es = dtype == MI355_DT_F32 ? 4 : 2;
    const size_t lds = (size_t)256 * ((zero_to - coff) * es + 4);
    dim3 grid((unsigned)((v + 255) / 256), n);
    if (dtype == MI355_DT_F32)
      hipLaunchKernelGGL(pack_tile_kernel<float>, grid, dim3(256), lds, (hipStream_t)stream, src, (float*)dst, c, (long long)v, ld, coff, zero_to, q, src1, c1);
    else
      hipLaunchKernelGGL(pack_tile_kernel<bf16_t>, grid, dim3(256), lds, (hipStream_t)stream, src, (bf16_t*)dst, c, (long long)v, ld, coff, zero_to, q, src1, c1);
    return mi355_check_launch("pack");
  }
  MI355_REQUIRE(((long long)v * pieces + 255) / 256 < (1ll << 31), "pack: too many voxels for one launch");
  dim3 grid((unsigned)(((long long)v * pieces + 255) / 256), n);
  if (dtype == MI355_DT_F32)
    hipLaunchKernelGGL(pack_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, src, (float*)dst, c, (long long)v, ld, coff, zero_to, q, src1, c1, pieces);
  else
    hipLaunchKernelGGL(pack_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)stream, src, (bf16_t*)dst, c, (long long)v, ld, coff, zero_to, q, src1, c1, pieces);
  return mi355_check_launch("pack");
}

static int unpack_impl(const void* src, float* dst, int32_t n, int32_t c, int64_t v, int32_t ld, int32_t coff,
                       int32_t dtype, S2D q, void* stream) {
  MI355_REQUIRE(src && dst && n > 0 && c > 0 && v > 0 && coff >= 0 && coff + c <= ld, "unpack: bad argument");
  MI355_REQUIRE(dtype == MI355_DT_F32 || dtype == MI355_DT_BF16, "unpack: bad dtype");
  dim3 grid((unsigned)((v + 255) / 256), n);
  if (dtype == MI355_DT_F32)
    hipLaunchKernelGGL(unpack_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, (const float*)src, dst, c, (long long)v, ld, coff, q);
  else
    hipLaunchKernelGGL(unpack_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)src, dst, c, (long long)v, ld, coff, q);
  return mi355_check_launch("unpack");
}

int mi355_pack_ncdhw(const float* src, void* dst, int32_t n, int32_t c, int64_t v, int32_t ld, int32_t coff,
                     int32_t zero_to, int32_t dtype, void* stream) {
  return pack_impl(src, dst, n, c, v, ld, coff, zero_to, dtype, S2D{0, 0, 0, 0}, stream);
}

int mi355_unpack_ncdhw(const void* src, float* dst, int32_t n, int32_t c, int64_t v, int32_t ld, int32_t coff,
                       int32_t dtype, void* stream) {
  return unpack_impl(src, dst, n, c, v, ld, coff, dtype, S2D{0, 0, 0, 0}, stream);
}

static int check_s2d(int d, int h, int w, int cblk, int ld, const char* who) {
  MI355_REQUIRE(d > 0 && h > 0 && w > 0 && d % 2 == 0 && h % 2 == 0 && w % 2 == 0, "%s: space-to-depth needs even extents", who);
  MI355_REQUIRE(cblk > 0 && cblk % 8 == 0 && ld >= 8 * cblk, "%s: space-to-depth row must hold 8 channel blocks (of a multiple of 8 channels)", who);
  return MI355_OK;
}

int mi355_pack_ncdhw_s2d(const float* src, void* dst, int32_t n, int32_t c, int32_t d, int32_t h, int32_t w,
                         int32_t cblk, int32_t ld, int32_t coff, int32_t zero_to, int32_t dtype, void* stream) {
  int rc = check_s2d(d, h, w, cblk, ld, "pack_s2d");
  if (rc) return rc;
  MI355_REQUIRE(zero_to <= cblk, "pack_s2d: channel window exceeds the block");
  return pack_impl(src, dst, n, c, (int64_t)d * h * w, ld, coff, zero_to, dtype, S2D{d, h, w, cblk}, stream);
}

int mi355_pack2_ncdhw(const float* src0, int32_t c0, const float* src1, int32_t c1, void* dst, int32_t n, int64_t v,
                      int32_t ld, int32_t coff, int32_t zero_to, int32_t dtype, void* stream) {
  MI355_REQUIRE(src1 && c1 > 0, "pack2: second source missing");
  return pack_impl(src0, dst, n, c0, v, ld, coff, zero_to, dtype, S2D{0, 0, 0, 0}, stream, src1, c1);
}

int mi355_pack2_ncdhw_s2d(const float* src0, int32_t c0, const float* src1, int32_t c1, void* dst, int32_t n, int32_t d,
                          int32_t h, int32_t w, int32_t cblk, int32_t ld, int32_t coff, int32_t zero_to, int32_t dtype,
                          void* stream) {
  int rc = check_s2d(d, h, w, cblk, ld, "pack2_s2d");
  if (rc) return rc;
  MI355_REQUIRE(src1 && c1 > 0 && zero_to <= cblk, "pack2_s2d: bad argument");
  return pack_impl(src0, dst, n, c0, (int64_t)d * h * w, ld, coff, zero_to, dtype, S2D{d, h, w, cblk}, stream, src1, c1);
}

int mi355_unpack_ncdhw_s2d(const void* src, float* dst, int32_t n, int32_t c, int32_t d, int32_t h, int32_t w,
                           int32_t cblk, int32_t ld, int32_t coff, int32_t dtype, void* stream) {
  int rc = check_s2d(d, h, w, cblk, ld, "unpack_s2d");
  if (rc) return rc;
  MI355_REQUIRE(coff + c <= cblk, "unpack_s2d: channel window exceeds the block");
  return unpack_impl(src, dst, n, c, (int64_t)d * h * w, ld, coff, dtype, S2D{d, h, w, cblk}, stream);
}

int mi355_s2d_repack(const void* src, int32_t ld_src, void* dst, int32_t ld_dst, int32_t n, int32_t d, int32_t h, int32_t w,
                     int32_t c, int32_t dtype, void* stream) {
  int rc = check_s2d(d, h, w, c, ld_dst, "s2d_repack");
  if (rc) return rc;
  MI355_REQUIRE(src && dst && n > 0 && ld_src >= c, "s2d_repack: bad argument");
  MI355_REQUIRE(dtype == MI355_DT_F32 || dtype == MI355_DT_BF16, "s2d_repack: bad dtype");
  const int epv = dtype == MI355_DT_F32 ? 4 : 8, pieces = c / epv;
  MI355_REQUIRE(ld_src % epv == 0 && ld_dst % epv == 0, "s2d_repack: rows must be 16-byte aligned");
  const long long v = (long long)d * h * w;
  const dim3 grid((unsigned)((v * pieces + 255) / 256), (unsigned)n);
  const S2D q{d, h, w, c};
  if (dtype == MI355_DT_F32) s2d_repack_kernel<float><<<grid, dim3(256), 0, (hipStream_t)stream>>>((const float*)src, ld_src, (float*)dst, ld_dst, v, pieces, q);
  else s2d_repack_kernel<bf16_t><<<grid, dim3(256), 0, (hipStream_t)stream>>>((const bf16_t*)src, ld_src, (bf16_t*)dst, ld_dst, v, pieces, q);
  return mi355_check_launch("s2d_repack");
}

int mi355_seam_grad(const float* g_ncdhw, const void* g_s2d, int32_t ld_s, int32_t cblk, void* dz, int32_t ld_dz, int32_t cpad,
                    int32_t n, int32_t c, int32_t d, int32_t h, int32_t w, int32_t dtype, void* stream) {
  MI355_REQUIRE(dz && (g_ncdhw || g_s2d) && n > 0 && c > 0, "seam_grad: bad argument");
  MI355_REQUIRE(dtype == MI355_DT_F32 || dtype == MI355_DT_BF16, "seam_grad: bad dtype");
  const int epv = dtype == MI355_DT_F32 ? 4 : 8;
  MI355_REQUIRE(cpad % epv == 0 && cpad >= c && ld_dz >= cpad && ld_dz % epv == 0, "seam_grad: bad output row");
  if (g_s2d) {
    int rc = check_s2d(d, h, w, cblk, ld_s, "seam_grad");
    if (rc) return rc;
    MI355_REQUIRE(cblk % epv == 0 && cblk <= cpad && ld_s % epv == 0, "seam_grad: bad space-to-depth block");
  }
  const long long v = (long long)d * h * w;
  const int pieces = cpad / epv;
  const dim3 grid((unsigned)((v * pieces + 255) / 256), (unsigned)n);
  const S2D q{d, h, w, g_s2d ? cblk : 0};
  if (dtype == MI355_DT_F32) seam_grad_kernel<float><<<grid, dim3(256), 0, (hipStream_t)stream>>>(g_ncdhw, (const float*)g_s2d, ld_s, (float*)dz, ld_dz, pieces, c, v, q);
  else seam_grad_kernel<bf16_t><<<grid, dim3(256), 0, (hipStream_t)stream>>>(g_ncdhw, (const bf16_t*)g_s2d, ld_s, (bf16_t*)dz, ld_dz, pieces, c, v, q);
  return mi355_check_launch("seam_grad");
}

int mi355_weight_pack(const mi355_wpack_desc* d, void* stream) {
  MI355_REQUIRE(d && d->src && d->dst, "weight_pack: null pointer");
  MI355_REQUIRE(d->coutp % 32 == 0 && d->cinp % 16 == 0 && d->cout <= d->coutp && d->cin <= d->cinp && d->ks >= 1 && d->ks <= 4,
                "weight_pack: bad extents");
  MI355_REQUIRE(d->s2d_mode >= 0 && d->s2d_mode <= 2 && (d->s2d_mode == 0 || (d->s2d_cp > 0 && d->s2d_cp % 8 == 0)),
                "weight_pack: bad space-to-depth mode");
  MI355_REQUIRE(d->dtype == MI355_DT_F32 || d->dtype == MI355_DT_BF16 || d->dtype == MI355_DT_FP8, "weight_pack: bad dtype");
  MI355_REQUIRE(d->dtype != MI355_DT_FP8 || d->q_amax, "weight_pack: fp8 packing needs q_amax");
  WpackArgs a;
  a.src = d->src; a.dst = d->dst; a.cout = d->cout; a.cin = d->cin; a.coutp = d->coutp; a.cinp = d->cinp; a.ks = d->ks;
  a.s_co = d->s_co; a.s_ci = d->s_ci; a.s_k0 = d->s_k[0]; a.s_k1 = d->s_k[1]; a.s_k2 = d->s_k[2];
  a.tb0 = d->tbase[0]; a.tb1 = d->tbase[1]; a.tb2 = d->tbase[2];
  a.ts0 = d->tstep[0]; a.ts1 = d->tstep[1]; a.ts2 = d->tstep[2];
  a.s2d_mode = d->s2d_mode; a.s2d_cp = d->s2d_cp;
  a.q_amax = d->q_amax;
  const long long total = (long long)d->cinp * d->ks * d->ks * d->ks * d->coutp;
  dim3 grid((unsigned)((total + 255) / 256));
  if (d->dtype == MI355_DT_F32) hipLaunchKernelGGL(wpack_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, a);
  else if (d->dtype == MI355_DT_FP8) hipLaunchKernelGGL(wpack_kernel<fp8_t>, grid, dim3(256), 0, (hipStream_t)stream, a);
  else hipLaunchKernelGGL(wpack_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)stream, a);
  return mi355_check_launch("weight_pack");
}

}  // extern "C"

namespace {
// Dense 3x3x3 weights W[R][C][27] (forward: R = cout, C = cin; data gradient: roles swapped, taps flipped): the
// kernels above gather 4-byte elements 27 floats (or C*27 floats) apart -- 16x read amplification on 34 M
// parameters after every optimiser step.  Here a block loads a 16 (R) x 16 (C) x 27 patch as 16 contiguous runs of
// 432 floats, keeps it in LDS and writes the packed [chunk][tap][co][16] order 512 contiguous bytes per tap.
template <typename T>
__global__ __launch_bounds__(256) void wpack_dense3_multi_kernel(const WpackMulti m) {
  const WpackArgs& a = m.a[blockIdx.y];
  constexpr int NT = 27, ROW = 16 * NT + 1;
  __shared__ float tile[16 * ROW];
  const int nchunk = a.cinp / 16, ncob = a.coutp / 16;
  if ((int)blockIdx.x >= nchunk * ncob) return;
  const int chunk = blockIdx.x / ncob, cob = blockIdx.x % ncob;
  const bool co_is_row = a.s_co > a.s_ci;                       // forward packing
  const int r0 = (co_is_row ? cob : chunk) * 16, c0 = (co_is_row ? chunk : cob) * 16;
  const int nr = co_is_row ? a.cout : a.cin, nc = co_is_row ? a.cin : a.cout;      // extents of W's two channel axes
  const long long rstride = co_is_row ? a.s_co : a.s_ci;
  const int cvalid = min(16, nc - c0);                          // may be <= 0: the patch is padding only
  for (int rl = 0; rl < 16; ++rl) {
    const float* src = a.src + (long long)(r0 + rl) * rstride + (long long)c0 * NT;
    const bool rok = r0 + rl < nr;
    for (int j = threadIdx.x; j < 16 * NT; j += 256) tile[rl * ROW + j] = (rok && j < cvalid * NT) ? src[j] : 0.f;
  }
  __syncthreads();
  // 16-byte stores: a thread packs EPT consecutive channels of one (tap, output channel) row of the chunk (2-byte stores, one
  // element per thread, left the launch at a third of the HBM rate: 120 us per generator update)
  constexpr int EPT = 16 / (int)sizeof(T), PPR = 16 / EPT;      // elements per 16-byte piece, pieces per 16-channel row
  const float qs = sizeof(T) == 1 ? fp8_scale_of(a.q_amax) : 1.f;
  T* dst0 = reinterpret_cast<T*>(a.dst) + ((long long)chunk * NT * a.coutp + (long long)cob * 16) * 16;
  for (int idx = threadIdx.x; idx < NT * 16 * PPR; idx += 256) {
    const int tap = idx / (16 * PPR), within = idx - tap * (16 * PPR);
    const int col = within / PPR, e0 = (within - col * PPR) * EPT;
    const int td = tap / 9, th = (tap / 3) % 3, tw = tap % 3;
    const int ts = (a.tb0 + a.ts0 * td) * 9 + (a.tb1 + a.ts1 * th) * 3 + (a.tb2 + a.ts2 * tw);
    alignas(16) T out[EPT];
#pragma unroll
    for (int j = 0; j < EPT; ++j) {
      const int e = e0 + j;
      const int rl = co_is_row ? col : e, cl = co_is_row ? e : col;
      Elem<T>::store(out + j, tile[rl * ROW + cl * NT + ts] * qs);
    }
    *reinterpret_cast<uint4*>(dst0 + (long long)tap * a.coutp * 16 + col * 16 + e0) = *reinterpret_cast<const uint4*>(out);
  }
}

// The PatchGAN k4 s2 weights W[co][c][4][4][4] packed for the space-to-depth formulation (8 dense taps j x 8 parity
// blocks blk; s2d_mode 1: the GEMM input-channel index is blk * cp + c (forward), 2: the GEMM output-channel index is
// (data gradient)).  For one (co, c) the 64 (j, blk) values are the contiguous 4x4x4 taps, so a block loads a
// [rows][cols][64] patch as contiguous runs and writes every (j, blk) slice as 128 contiguous packed elements
// (the gather kernel reads 4 bytes per 256-byte stride: 150 us per step for 11 M discriminator weights).
template <typename T>
__global__ __launch_bounds__(256) void wpack_s2d_multi_kernel(const WpackMulti m) {
  const WpackArgs& a = m.a[blockIdx.y];
  __shared__ float tile[128 * 65];
  const bool fwd = a.s2d_mode == 1;
  const int R = fwd ? 8 : 16, Ccols = fwd ? 16 : 8;                 // rows = W's first axis, cols = channels c
  const int nrb = (fwd ? a.coutp : a.cinp) / R, ncb = a.s2d_cp / Ccols;
  if ((int)blockIdx.x >= nrb * ncb) return;
  const int rb = blockIdx.x / ncb, cb = blockIdx.x % ncb;
  const int r0 = rb * R, c0 = cb * Ccols;
  const int nrow = fwd ? a.cout : a.cin, ncol = fwd ? a.cin : a.cout;   // real extents of W's two channel axes
  const long long rstride = fwd ? a.s_co : a.s_ci;                      // 64 * (channels of the second axis)
  for (int i = threadIdx.x; i < R * Ccols * 64; i += 256) {
    const int rl = i / (Ccols * 64), rest = i - rl * (Ccols * 64);      // rest = cl * 64 + tap: contiguous in W
    const int cl = rest >> 6;
    const bool ok = r0 + rl < nrow && c0 + cl < ncol;
    tile[(rl * Ccols + cl) * 65 + (rest & 63)] = ok ? a.src[(long long)(r0 + rl) * rstride + (long long)c0 * 64 + rest] : 0.f;
  }
  __syncthreads();
  // 64 (j, blk) slices of 128 packed elements each; 256 threads write two slices per pass
  const int half = threadIdx.x >> 7, q = threadIdx.x & 127;
  const int x8 = q >> 4, e = q & 15;                                   // fwd: (co_l, e = c_l); dgrad: (c_l, e = row_l)
  const int rl = fwd ? x8 : e, cl = fwd ? e : x8;
  for (int sidx = half; sidx < 64; sidx += 2) {
    const int j = sidx >> 3, blk = sidx & 7;
    const int jd = j >> 2, jh = (j >> 1) & 1, jw = j & 1;
    const int tap = (a.tb0 + a.ts0 * jd + (blk >> 2)) * 16 + (a.tb1 + a.ts1 * jh + ((blk >> 1) & 1)) * 4 + (a.tb2 + a.ts2 * jw + (blk & 1));
    const float v = tile[(rl * Ccols + cl) * 65 + tap];
    long long dst;
    if (fwd) {   // dest [chunk = (blk*cp + c0)/16][j][co][e]
      const int chunk = (blk * a.s2d_cp + c0) >> 4;
      dst = (((long long)chunk * 8 + j) * a.coutp + r0 + x8) * 16 + e;
    } else {     // dest [chunk = r0/16][j][co' = blk*cp + c0 + c_l][e]
      dst = (((long long)(r0 >> 4) * 8 + j) * a.coutp + (long long)blk * a.s2d_cp + c0 + x8) * 16 + e;
    }
    Elem<T>::store(reinterpret_cast<T*>(a.dst) + dst, v);
  }
}

static bool wpack_is_s2d_dense(const mi355_wpack_desc* d) {
  if (d->ks != 2 || (d->s2d_mode != 1 && d->s2d_mode != 2) || d->s2d_cp % 16 || d->s_k[2] != 1 || d->s_k[1] != 4 || d->s_k[0] != 16) return false;
  for (int k = 0; k < 3; ++k) {
    const int lo = d->tbase[k] < d->tbase[k] + d->tstep[k] ? d->tbase[k] : d->tbase[k] + d->tstep[k];
    const int hi = d->tbase[k] + d->tstep[k] + 1 > d->tbase[k] + 1 ? d->tbase[k] + d->tstep[k] + 1 : d->tbase[k] + 1;
    if (lo < 0 || hi > 3) return false;                              // taps tb + ts*j + b, j and b in {0, 1}, stay in 0..3
  }
  if (d->s2d_mode == 1) return d->s_ci == 64 && d->s_co == 64ll * d->cin && d->cinp == 8 * d->s2d_cp && d->coutp % 8 == 0 && d->cin <= d->s2d_cp;
  return d->s_co == 64 && d->s_ci == 64ll * d->cout && d->coutp == 8 * d->s2d_cp && d->cinp % 16 == 0 && d->cout <= d->s2d_cp;
}

static bool wpack_is_dense3(const mi355_wpack_desc* d) {
  if (d->ks != 3 || d->s2d_mode != 0 || d->s_k[2] != 1 || d->s_k[1] != 3 || d->s_k[0] != 9) return false;
  const long long lo = d->s_co < d->s_ci ? d->s_co : d->s_ci, hi = d->s_co < d->s_ci ? d->s_ci : d->s_co;
  const long long inner = d->s_co > d->s_ci ? d->cin : d->cout;              // channels along W's second axis
  for (int k = 0; k < 3; ++k) {
    const int first = d->tbase[k], last = d->tbase[k] + 2 * d->tstep[k];
    if (first < 0 || first > 2 || last < 0 || last > 2) return false;
  }
  return lo == 27 && hi == 27 * inner;
}
}  // namespace

extern "C" {

static int fill_wpack(const mi355_wpack_desc* d, WpackArgs* a) {
  MI355_REQUIRE(d && d->src && d->dst, "weight_pack: null pointer");
  MI355_REQUIRE(d->coutp % 32 == 0 && d->cinp % 16 == 0 && d->cout <= d->coutp && d->cin <= d->cinp && d->ks >= 1 && d->ks <= 4,
                "weight_pack: bad extents");
  MI355_REQUIRE(d->s2d_mode >= 0 && d->s2d_mode <= 2 && (d->s2d_mode == 0 || (d->s2d_cp > 0 && d->s2d_cp % 8 == 0)),
                "weight_pack: bad space-to-depth mode");
  a->src = d->src; a->dst = d->dst; a->cout = d->cout; a->cin = d->cin; a->coutp = d->coutp; a->cinp = d->cinp; a->ks = d->ks;
  a->s_co = d->s_co; a->s_ci = d->s_ci; a->s_k0 = d->s_k[0]; a->s_k1 = d->s_k[1]; a->s_k2 = d->s_k[2];
  a->tb0 = d->tbase[0]; a->tb1 = d->tbase[1]; a->tb2 = d->tbase[2];
  a->ts0 = d->tstep[0]; a->ts1 = d->tstep[1]; a->ts2 = d->tstep[2];
  a->s2d_mode = d->s2d_mode; a->s2d_cp = d->s2d_cp;
  a->q_amax = d->q_amax;
  return MI355_OK;
}

int mi355_weight_pack_multi(const mi355_wpack_desc* descs, int32_t n, void* stream) {
  MI355_REQUIRE(descs && n > 0, "weight_pack_multi: bad argument");
  const int dtype = descs[0].dtype;
  MI355_REQUIRE(dtype == MI355_DT_F32 || dtype == MI355_DT_BF16 || dtype == MI355_DT_FP8, "weight_pack_multi: bad dtype");
  // three passes over the list: dense 3x3x3 packings and the space-to-depth k4 packings go through their
  // LDS-transposing kernels, the rest through the gather
  for (int pass = 0; pass < 3; ++pass) {
    const bool want_dense = pass == 0, want_s2d = pass == 1;
    int i0 = 0;
    while (i0 < n) {
      WpackMulti m;
      int cnt = 0;
      long long mx = 0, patches = 0;
      const mi355_wpack_desc* firstd = nullptr;
      for (; i0 < n && cnt < kWpackChunk; ++i0) {
        const mi355_wpack_desc* d = &descs[i0];
        MI355_REQUIRE(d->dtype == dtype, "weight_pack_multi: mixed dtypes");
        const int klass = wpack_is_dense3(d) ? 0 : (wpack_is_s2d_dense(d) ? 1 : 2);
        if (klass != pass) continue;
        int rc = fill_wpack(d, &m.a[cnt]);
        if (rc) return rc;
        if (!firstd) firstd = d;
        const long long total = (long long)d->cinp * d->ks * d->ks * d->ks * d->coutp;
        const long long np = want_s2d ? (d->s2d_mode == 1 ? (long long)(d->coutp / 8) * (d->s2d_cp / 16) : (long long)(d->cinp / 16) * (d->s2d_cp / 8))
                                      : (long long)(d->cinp / 16) * (d->coutp / 16);
        if (total > mx) mx = total;
        if (np > patches) patches = np;
        ++cnt;
      }
      if (cnt == 0) break;
      for (int i = cnt; i < kWpackChunk; ++i) m.a[i] = m.a[0];       // unused slots: never indexed (grid.y = cnt)
      if (want_dense) {
        MI355_REQUIRE(patches < (1ll << 31), "weight_pack_multi: too many patches");
        dim3 grid((unsigned)patches, cnt);
        if (dtype == MI355_DT_F32) wpack_dense3_multi_kernel<float><<<grid, dim3(256), 0, (hipStream_t)stream>>>(m);
        else if (dtype == MI355_DT_FP8) wpack_dense3_multi_kernel<fp8_t><<<grid, dim3(256), 0, (hipStream_t)stream>>>(m);
        else wpack_dense3_multi_kernel<bf16_t><<<grid, dim3(256), 0, (hipStream_t)stream>>>(m);
      } else if (want_s2d) {
        MI355_REQUIRE(patches < (1ll << 31), "weight_pack_multi: too many patches");
        dim3 grid((unsigned)patches, cnt);
        MI355_REQUIRE(dtype != MI355_DT_FP8, "weight_pack_multi: no fp8 space-to-depth packing");
        if (dtype == MI355_DT_F32) wpack_s2d_multi_kernel<float><<<grid, dim3(256), 0, (hipStream_t)stream>>>(m);
        else wpack_s2d_multi_kernel<bf16_t><<<grid, dim3(256), 0, (hipStream_t)stream>>>(m);
      } else {
        long long nb = (mx + 256 * 8 - 1) / (256 * 8);
        if (nb > 512) nb = 512;
        dim3 grid((unsigned)nb, cnt);
        if (dtype == MI355_DT_F32) hipLaunchKernelGGL(wpack_multi_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, m);
        else if (dtype == MI355_DT_FP8) hipLaunchKernelGGL(wpack_multi_kernel<fp8_t>, grid, dim3(256), 0, (hipStream_t)stream, m);
        else hipLaunchKernelGGL(wpack_multi_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)stream, m);
      }
    }
  }
  return mi355_check_launch("weight_pack_multi");
}

int32_t mi355_channel_stats_blocks(int64_t rows_per_group) {
  // >= 128 rows per block, ~1024 blocks for the big tensors (2048 rows each at 128^3)
  long long b = rows_per_group / 128;
  if (b > 1024) b = (rows_per_group + kRowsPerStatBlock - 1) / kRowsPerStatBlock;
  if (b < 1024 && rows_per_group / 128 > 1024) b = 1024;
  // small tensors (16^3: 4 096 rows) got 32 workgroups -- 17 us for a 2 MB reduction in the replay trace: at least
  // min(256, rows / 16) of them
  const long long fine = rows_per_group / 16 < 256 ? rows_per_group / 16 : 256;
  if (b < fine) b = fine;
  if (b < 1) b = 1;
  if (b > 2048) b = 2048;
  return (int32_t)b;
}

int mi355_channel_stats(const void* x, int32_t ld, int32_t c, int64_t rows_per_group, int32_t groups, float* part,
                        int32_t blocks_per_group, int32_t dtype, void* stream) {
  MI355_REQUIRE(x && part && rows_per_group > 0 && groups > 0 && blocks_per_group > 0, "channel_stats: bad argument");
  int rc = check_rows(c, ld, dtype, "channel_stats");
  if (rc) return rc;
  dim3 grid(blocks_per_group, groups);
  if (dtype == MI355_DT_F32)
    hipLaunchKernelGGL(channel_stats_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, (const float*)x, ld, c, (long long)rows_per_group, part, blocks_per_group);
  else
    hipLaunchKernelGGL(channel_stats_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, ld, c, (long long)rows_per_group, part, blocks_per_group);
  return mi355_check_launch("channel_stats");
}

int mi355_norm_finalize(const float* part, int32_t parts_per_group, int32_t groups, int32_t c, int64_t count_per_group,
                        const float* shift, int32_t n_real, float eps, float* mean, float* rstd, float* running_mean,
                        float* running_var, float momentum, int64_t* batches_tracked, void* stream) {
  MI355_REQUIRE(part && mean && rstd && parts_per_group > 0 && groups > 0 && c > 0 && count_per_group > 0, "norm_finalize: bad argument");
  MI355_REQUIRE(!running_mean || running_var, "norm_finalize: running_mean without running_var");
  const bool serial = running_mean && groups > 1;         // running statistics: the groups' momentum updates in order
  hipLaunchKernelGGL(norm_finalize_kernel, dim3((c + 7) / 8, serial ? 1 : groups), dim3(1024), 0, (hipStream_t)stream, part,
                     parts_per_group, c, (long long)count_per_group, shift, n_real > 0 ? n_real : c, eps, mean, rstd, running_mean, running_var, momentum,
                     (long long*)batches_tracked, serial ? groups : 1);
  return mi355_check_launch("norm_finalize");
}

int mi355_colsum_finalize(const float* part, int32_t parts, int32_t c, float* out, void* stream) {
  return mi355_colsum_finalize_into(part, parts, c, out, c, 0, stream);
}

int mi355_colsum_finalize_into(const float* part, int32_t parts, int32_t c, float* out, int32_t n_out, int32_t accumulate,
                               void* stream) {
  return mi355_colsum_finalize_from(part, parts, c, 0, out, n_out, accumulate, stream);
}

int mi355_colsum_finalize_from(const float* part, int32_t parts, int32_t c, int32_t offset, float* out, int32_t n_out,
                               int32_t accumulate, void* stream) {
  MI355_REQUIRE(part && out && parts > 0 && c > 0 && n_out > 0 && offset >= 0 && offset + n_out <= c, "colsum_finalize: bad argument");
  hipLaunchKernelGGL(colsum_finalize_kernel, dim3((n_out + 7) / 8), dim3(1024), 0, (hipStream_t)stream, part, parts, c, offset, out, n_out,
                     accumulate);
  return mi355_check_launch("colsum_finalize");
}

static int fill_normact(const mi355_normact_desc* d, NormActArgs* q, const char* who) {
  MI355_REQUIRE(d && d->z, "%s: null pointer", who);
  int rc = check_rows(d->c, d->ldz, d->dtype, who);
  if (rc) return rc;
  MI355_REQUIRE(d->rows_per_group > 0 && d->groups > 0, "%s: empty", who);
  MI355_REQUIRE(!d->mean || d->rstd, "%s: mean without rstd", who);
  MI355_REQUIRE(d->drop_p >= 0.f && d->drop_p < 1.f, "%s: dropout p out of range", who);
  q->z = (const char*)d->z; q->ldz = d->ldz; q->a = (char*)d->a; q->lda = d->lda;
  q->c = d->c; q->rows_per_group = d->rows_per_group; q->groups = d->groups;
  q->mean = d->mean; q->rstd = d->rstd; q->gamma = d->gamma; q->beta = d->beta;
  q->slope = d->slope;
  q->thr16 = d->drop_p > 0.f ? (unsigned)(d->drop_p * 65536.f + 0.5f) : 0u;
  q->drop_scale = d->drop_p > 0.f ? 1.f / (1.f - d->drop_p) : 1.f;
  q->seed = d->seed;
  q->seed_ptr = (const unsigned long long*)d->seed_ptr;
  q->n_affine = d->n_affine > 0 ? d->n_affine : d->c;
  q->da = (const char*)d->da; q->ldda = d->ldda; q->dz = (char*)d->dz; q->lddz = d->lddz;
  q->part = d->part; q->blocks_per_group = d->blocks_per_group; q->sums = d->sums; q->batch_stats = d->batch_stats;
  q->s2d_a = S2D{0, 0, 0, 0};
  q->s2d_da = S2D{0, 0, 0, 0};
  MI355_REQUIRE(!d->q8 || (d->dtype == MI355_DT_BF16 && d->c == 32 && d->ld8 >= d->c && d->ld8 % 8 == 0 && d->q_use && d->q_next &&
                           !d->s2d_a && !d->s2d_da),
                "%s: the e4m3 copy is written for plain bf16 tensors of 32 channels (q8, q_use, q_next)", who);
  q->q8 = (uint8_t*)d->q8; q->ld8 = d->ld8; q->q_use = d->q_use; q->q_next = d->q_next;
  MI355_REQUIRE(!d->gz || (d->dtype == MI355_DT_BF16 && !d->da && !d->s2d_da && d->gw && d->gk > 0 && d->gk <= 8 && d->ldgz >= 8 &&
                           d->ldgz % 8 == 0 && d->gw_ld > 0 && d->c <= kImplicitMaxC),
                "%s: the implicit 1x1x1 data gradient needs bf16, gz rows of >= 8 channels, gk <= 8 and no da", who);
  q->gz = (const char*)d->gz; q->ldgz = d->ldgz; q->gw = d->gw; q->gw_ld = d->gw_ld; q->gk = d->gk;
  MI355_REQUIRE(!d->fy || (d->dtype == MI355_DT_BF16 && d->gw && d->gk > 0 && d->gk <= 8 && d->gw_ld > 0 && d->c == 32 &&
                           d->fcp >= 8 && d->fcp % 8 == 0 && d->fcp <= d->c && d->ldfy >= d->fcp && d->ldfy % 8 == 0 && !d->s2d_a &&
                           !d->q8 && d->ldz % 8 == 0 && (d->skip_a || d->lda % 8 == 0)),
                "%s: the fused 1x1x1 convolution needs bf16, 32 channels, gk <= 8 outputs in rows of fcp (8 .. 32) channels", who);
  MI355_REQUIRE(!d->skip_a || d->fy, "%s: skip_a without the fused convolution", who);
  q->fy = (char*)d->fy; q->ldfy = d->ldfy; q->fcp = d->fcp; q->fbias = d->fbias; q->skip_a = d->skip_a;
  MI355_REQUIRE(!d->pool_idx || (d->pool_dy && !d->gz && !d->s2d_da && !d->s2d_a && d->sd >= 2 && d->sh >= 2 && d->sw >= 2 &&
                                 !((d->sd | d->sh | d->sw) & 1) && d->ldpdy >= d->c && d->ldpdy % (d->dtype == MI355_DT_F32 ? 4 : 8) == 0 &&
                                 ((long long)d->rows_per_group * d->groups) % ((long long)d->sd * d->sh * d->sw) == 0 &&
                                 (long long)d->rows_per_group * d->groups < (1ll << 31) && (!d->da || d->ldda >= d->c)),
                "%s: the implicit max-pool gradient needs pool_dy, even extents sd/sh/sw that divide the row count, and plain layouts", who);
  q->pool_idx = (const uint8_t*)d->pool_idx; q->pool_dy = (const char*)d->pool_dy; q->ldpdy = d->ldpdy;
  q->pd = d->sd; q->ph = d->sh; q->pw = d->sw;
  MI355_REQUIRE(!d->pool_y || (d->pool_widx && !d->pool_idx && !d->s2d_a && !d->s2d_da && !d->q8 && !d->fy && d->a && d->sd >= 2 &&
                               d->sh >= 2 && d->sw >= 2 && !((d->sd | d->sh | d->sw) & 1) && d->ldpy >= d->c &&
                               d->ldpy % (d->dtype == MI355_DT_F32 ? 4 : 8) == 0 &&
                               d->rows_per_group % ((long long)d->sd * d->sh * d->sw) == 0),
                "%s: the fused max-pool needs pool_widx, even extents sd/sh/sw that divide the row count, and plain layouts", who);
  q->pool_y = (char*)d->pool_y; q->ldpy = d->ldpy; q->pool_widx = (uint8_t*)d->pool_widx;
  if (d->s2d_a || d->s2d_da) {
    MI355_REQUIRE((long long)d->sd * d->sh * d->sw * (d->groups == 1 ? 1 : 1) > 0 &&
                  ((long long)d->rows_per_group * d->groups) % ((long long)d->sd * d->sh * d->sw) == 0,
                  "%s: space-to-depth extents do not match the row count", who);
    if (d->s2d_a) { int rc2 = check_s2d(d->sd, d->sh, d->sw, d->c, d->lda, who); if (rc2) return rc2; q->s2d_a = S2D{d->sd, d->sh, d->sw, d->c}; }
    if (d->s2d_da) { int rc2 = check_s2d(d->sd, d->sh, d->sw, d->c, d->ldda, who); if (rc2) return rc2; q->s2d_da = S2D{d->sd, d->sh, d->sw, d->c}; }
  }
  return MI355_OK;
}

static unsigned stream_blocks(long long rows, int c, int dtype) {
  const int epv = dtype == MI355_DT_F32 ? 4 : 8;
  const int rpp = 256 / (c / epv);
  long long b = (rows + (long long)rpp * 8 - 1) / ((long long)rpp * 8);
  if (b < 1) b = 1;
  // 4 workgroups per CU, each streaming a long row range: measured against 256 ... 16384 in the full step
  // (interleaved A/B): 1024 and 768 best, 4096 +0.12 ms, 256 +0.75 ms
  if (b > 1024) b = 1024;
  return (unsigned)b;
}

int mi355_normact_fwd(const mi355_normact_desc* d, void* stream) {
  NormActArgs q;
  int rc = fill_normact(d, &q, "normact_fwd");
  if (rc) return rc;
  MI355_REQUIRE((d->a && d->lda >= d->c) || d->skip_a, "normact_fwd: bad output");
  dim3 grid(stream_blocks(d->rows_per_group, d->c, d->dtype), d->groups);
  if (q.pool_y) {
    const long long dhw = (long long)d->sd * d->sh * d->sw, samples = (long long)d->rows_per_group * d->groups / dhw;
    const int epv = d->dtype == MI355_DT_F32 ? 4 : 8, vpp = 256 / (d->c / epv);
    long long b = (dhw / 8 + vpp - 1) / vpp;                      // one window per thread and pass; up to ~2048 workgroups in all
    const long long cap = std::max(1ll, 2048 / samples);
    if (b > cap) b = cap;
    const dim3 gridp((unsigned)b, (unsigned)samples);
    if (d->dtype == MI355_DT_F32) {
      if (q.thr16) normact_pool_fwd_kernel<float, true><<<gridp, dim3(256), 0, (hipStream_t)stream>>>(q);
      else normact_pool_fwd_kernel<float, false><<<gridp, dim3(256), 0, (hipStream_t)stream>>>(q);
    } else {
      if (q.thr16) normact_pool_fwd_kernel<bf16_t, true><<<gridp, dim3(256), 0, (hipStream_t)stream>>>(q);
      else normact_pool_fwd_kernel<bf16_t, false><<<gridp, dim3(256), 0, (hipStream_t)stream>>>(q);
    }
    return mi355_check_launch("normact_pool_fwd");
  }
  if (d->dtype == MI355_DT_F32) {
    if (q.thr16) normact_fwd_kernel<float, true><<<grid, dim3(256), 0, (hipStream_t)stream>>>(q);
    else normact_fwd_kernel<float, false><<<grid, dim3(256), 0, (hipStream_t)stream>>>(q);
  } else if (q.fy) {
    // (the convolution on the matrix pipe; 64 rows per workgroup and pass)
    MI355_REQUIRE(d->fcp == 16, "normact_fwd: the fused convolution writes 16-channel rows");
    long long b = (d->rows_per_group + 64 * 8 - 1) / (64 * 8);
    const dim3 grid32((unsigned)(b < 1 ? 1 : (b > 2048 ? 2048 : b)), d->groups);
    if (q.thr16) normact_fwd_final32_kernel<true><<<grid32, dim3(256), 0, (hipStream_t)stream>>>(q);
    else normact_fwd_final32_kernel<false><<<grid32, dim3(256), 0, (hipStream_t)stream>>>(q);
  } else {
    if (q.thr16) normact_fwd_kernel<bf16_t, true><<<grid, dim3(256), 0, (hipStream_t)stream>>>(q);
    else normact_fwd_kernel<bf16_t, false><<<grid, dim3(256), 0, (hipStream_t)stream>>>(q);
  }
  return mi355_check_launch("normact_fwd");
}

int mi355_normact_bwd_reduce(const mi355_normact_desc* d, void* stream) {
  NormActArgs q;
  int rc = fill_normact(d, &q, "normact_bwd_reduce");
  if (rc) return rc;
  MI355_REQUIRE((d->gz || d->pool_idx || (d->da && d->ldda >= d->c)) && d->part && d->blocks_per_group > 0, "normact_bwd_reduce: bad argument");
  dim3 grid(d->blocks_per_group, d->groups);
  if (q.pool_idx) {
    if (d->dtype == MI355_DT_F32) {
      if (q.thr16) normact_bwd_reduce_kernel<float, true, 2><<<grid, dim3(256), 0, (hipStream_t)stream>>>(q);
      else normact_bwd_reduce_kernel<float, false, 2><<<grid, dim3(256), 0, (hipStream_t)stream>>>(q);
    } else {
      if (q.thr16) normact_bwd_reduce_kernel<bf16_t, true, 2><<<grid, dim3(256), 0, (hipStream_t)stream>>>(q);
      else normact_bwd_reduce_kernel<bf16_t, false, 2><<<grid, dim3(256), 0, (hipStream_t)stream>>>(q);
    }
  } else if (d->dtype == MI355_DT_F32) {
    if (q.thr16) normact_bwd_reduce_kernel<float, true><<<grid, dim3(256), 0, (hipStream_t)stream>>>(q);
    else normact_bwd_reduce_kernel<float, false><<<grid, dim3(256), 0, (hipStream_t)stream>>>(q);
  } else if (q.gz) {
    if (q.thr16) normact_bwd_reduce_kernel<bf16_t, true, 1><<<grid, dim3(256), 0, (hipStream_t)stream>>>(q);
    else normact_bwd_reduce_kernel<bf16_t, false, 1><<<grid, dim3(256), 0, (hipStream_t)stream>>>(q);
  } else {
    if (q.thr16) normact_bwd_reduce_kernel<bf16_t, true><<<grid, dim3(256), 0, (hipStream_t)stream>>>(q);
    else normact_bwd_reduce_kernel<bf16_t, false><<<grid, dim3(256), 0, (hipStream_t)stream>>>(q);
  }
  return mi355_check_launch("normact_bwd_reduce");
}

int mi355_normact_bwd_finalize(const float* part, int32_t blocks_per_group, int32_t groups, int32_t c, float* sums,
                               float* dgamma, float* dbeta, void* stream) {
  return mi355_normact_bwd_finalize_into(part, blocks_per_group, groups, c, sums, dgamma, dbeta, c, 0, stream);
}

int mi355_normact_bwd_finalize_into(const float* part, int32_t blocks_per_group, int32_t groups, int32_t c, float* sums,
                                    float* dgamma, float* dbeta, int32_t n_affine, int32_t accumulate, void* stream) {
  MI355_REQUIRE(part && sums && blocks_per_group > 0 && groups > 0 && c > 0 && n_affine > 0 && n_affine <= c,
                "normact_bwd_finalize: bad argument");
  hipLaunchKernelGGL(normact_bwd_finalize_kernel, dim3((c + 7) / 8), dim3(1024), 0, (hipStream_t)stream, part,
                     blocks_per_group, groups, c, sums, dgamma, dbeta, n_affine, accumulate);
  return mi355_check_launch("normact_bwd_finalize");
}

int mi355_normact_bwd_apply(const mi355_normact_desc* d, void* stream) {
  NormActArgs q;
  int rc = fill_normact(d, &q, "normact_bwd_apply");
  if (rc) return rc;
  MI355_REQUIRE((d->gz || d->pool_idx || (d->da && d->ldda >= d->c)) && d->dz && d->lddz >= d->c, "normact_bwd_apply: bad argument");
  MI355_REQUIRE(!(d->mean && d->batch_stats) || d->sums, "normact_bwd_apply: sums required");
  dim3 grid(stream_blocks(d->rows_per_group, d->c, d->dtype), d->groups);
  if (q.pool_idx) {
    if (d->dtype == MI355_DT_F32) {
      if (q.thr16) normact_bwd_apply_kernel<float, true, 2><<<grid, dim3(256), 0, (hipStream_t)stream>>>(q);
      else normact_bwd_apply_kernel<float, false, 2><<<grid, dim3(256), 0, (hipStream_t)stream>>>(q);
    } else {
      if (q.thr16) normact_bwd_apply_kernel<bf16_t, true, 2><<<grid, dim3(256), 0, (hipStream_t)stream>>>(q);
      else normact_bwd_apply_kernel<bf16_t, false, 2><<<grid, dim3(256), 0, (hipStream_t)stream>>>(q);
    }
  } else if (d->dtype == MI355_DT_F32) {
    if (q.thr16) normact_bwd_apply_kernel<float, true><<<grid, dim3(256), 0, (hipStream_t)stream>>>(q);
    else normact_bwd_apply_kernel<float, false><<<grid, dim3(256), 0, (hipStream_t)stream>>>(q);
  } else if (q.gz) {
    if (q.thr16) normact_bwd_apply_kernel<bf16_t, true, 1><<<grid, dim3(256), 0, (hipStream_t)stream>>>(q);
    else normact_bwd_apply_kernel<bf16_t, false, 1><<<grid, dim3(256), 0, (hipStream_t)stream>>>(q);
  } else {
    if (q.thr16) normact_bwd_apply_kernel<bf16_t, true><<<grid, dim3(256), 0, (hipStream_t)stream>>>(q);
    else normact_bwd_apply_kernel<bf16_t, false><<<grid, dim3(256), 0, (hipStream_t)stream>>>(q);
  }
  return mi355_check_launch("normact_bwd_apply");
}

int mi355_normact_small_fwd(const mi355_normact_small_desc* d, void* stream) {
  NormSmallArgs p;
  MI355_REQUIRE(d, "normact_small_fwd: null pointer");
  int rc = fill_normact(&d->base, &p.q, "normact_small_fwd");
  if (rc) return rc;
  MI355_REQUIRE(d->base.a && d->base.lda >= d->base.c && d->mean_out && d->rstd_out, "normact_small_fwd: bad output");
  MI355_REQUIRE(!d->running_mean || d->running_var, "normact_small_fwd: running_mean without running_var");
  p.eps = d->eps; p.momentum = d->momentum; p.mean_out = d->mean_out; p.rstd_out = d->rstd_out;
  p.running_mean = d->running_mean; p.running_var = d->running_var; p.batches_tracked = (long long*)d->batches_tracked;
  p.n_real = d->n_real > 0 ? d->n_real : d->base.c;
  p.dgamma = p.dbeta = nullptr; p.accumulate = 0;
  const int epv = d->base.dtype == MI355_DT_F32 ? 4 : 8;
  dim3 grid(d->base.c / epv);
  if (p.q.rows_per_group <= 8ll * kSmallThreads && p.q.groups <= kSmallResMaxGroups) {       // register-resident form: one load latency per chunk of groups
    const int s_slots = p.q.rows_per_group <= kSmallThreads ? 1 : 8, chunk = p.q.groups % 2 == 0 ? 2 : 1;
    // InstanceNorm (no running statistics, which are a recurrence over the groups): the chunks of groups are independent workgroups --
    // the reference's batch of 8 patches is 8 groups of 64 - 512 rows, walked 4 chunks deep by c / 8 = 16 - 32 workgroups otherwise
    if (!p.running_mean) grid.y = (unsigned)(p.q.groups / chunk);
    if (d->base.dtype == MI355_DT_F32) {
      if (p.q.thr16) MI355_SMALL_RES_LAUNCH(normact_small_res_fwd_kernel, float, true, 2);
      else MI355_SMALL_RES_LAUNCH(normact_small_res_fwd_kernel, float, false, 2);
    } else {
      if (p.q.thr16) MI355_SMALL_RES_LAUNCH(normact_small_res_fwd_kernel, bf16_t, true, 2);
      else MI355_SMALL_RES_LAUNCH(normact_small_res_fwd_kernel, bf16_t, false, 2);
    }
    return mi355_check_launch("normact_small_fwd");
  }
  if (d->base.dtype == MI355_DT_F32) {
    if (p.q.thr16) normact_small_fwd_kernel<float, true><<<grid, dim3(kSmallThreads), 0, (hipStream_t)stream>>>(p);
    else normact_small_fwd_kernel<float, false><<<grid, dim3(kSmallThreads), 0, (hipStream_t)stream>>>(p);
  } else {
    if (p.q.thr16) normact_small_fwd_kernel<bf16_t, true><<<grid, dim3(kSmallThreads), 0, (hipStream_t)stream>>>(p);
    else normact_small_fwd_kernel<bf16_t, false><<<grid, dim3(kSmallThreads), 0, (hipStream_t)stream>>>(p);
  }
  return mi355_check_launch("normact_small_fwd");
}

int mi355_normact_small_bwd(const mi355_normact_small_desc* d, void* stream) {
  NormSmallArgs p;
  MI355_REQUIRE(d, "normact_small_bwd: null pointer");
  int rc = fill_normact(&d->base, &p.q, "normact_small_bwd");
  if (rc) return rc;
  MI355_REQUIRE(d->base.da && d->base.dz && d->base.mean, "normact_small_bwd: null pointer");
  p.eps = d->eps; p.momentum = 0.f; p.mean_out = p.rstd_out = nullptr;
  p.running_mean = p.running_var = nullptr; p.batches_tracked = nullptr; p.n_real = 0;
  p.dgamma = d->dgamma; p.dbeta = d->dbeta; p.accumulate = d->accumulate;
  const int epv = d->base.dtype == MI355_DT_F32 ? 4 : 8;
  dim3 grid(d->base.c / epv);
  if (p.q.rows_per_group <= 8ll * kSmallThreads && p.q.groups <= kSmallResMaxGroups) {       // register-resident form: one load latency per chunk of groups
    const int s_slots = p.q.rows_per_group <= kSmallThreads ? 1 : 8, chunk = p.q.groups % 2 == 0 ? 2 : 1;
    // scratch for the per-group sums given (base.part: groups x 2 x c doubles): the chunks of groups are independent workgroups and the
    // affine gradients are summed over the groups by a second, tiny launch
    const bool split = d->base.part != nullptr && p.q.groups / chunk > 1;
    if (split) grid.y = (unsigned)(p.q.groups / chunk); else p.q.part = nullptr;
    if (d->base.dtype == MI355_DT_F32) {
      if (p.q.thr16) MI355_SMALL_RES_LAUNCH(normact_small_res_bwd_kernel, float, true, 1);
      else MI355_SMALL_RES_LAUNCH(normact_small_res_bwd_kernel, float, false, 1);
    } else {
      if (p.q.thr16) MI355_SMALL_RES_LAUNCH(normact_small_res_bwd_kernel, bf16_t, true, 1);
      else MI355_SMALL_RES_LAUNCH(normact_small_res_bwd_kernel, bf16_t, false, 1);
    }
    if (split && (p.dgamma || p.dbeta) && p.q.n_affine > 0)
      normact_small_affine_kernel<<<dim3((p.q.n_affine + 255) / 256), dim3(256), 0, (hipStream_t)stream>>>(
          reinterpret_cast<const double*>(d->base.part), p.q.groups, p.q.c, p.q.n_affine, p.dgamma, p.dbeta, p.accumulate);
    return mi355_check_launch("normact_small_bwd");
  }
  if (d->base.dtype == MI355_DT_F32) {
    if (p.q.thr16) normact_small_bwd_kernel<float, true><<<grid, dim3(kSmallThreads), 0, (hipStream_t)stream>>>(p);
    else normact_small_bwd_kernel<float, false><<<grid, dim3(kSmallThreads), 0, (hipStream_t)stream>>>(p);
  } else {
    if (p.q.thr16) normact_small_bwd_kernel<bf16_t, true><<<grid, dim3(kSmallThreads), 0, (hipStream_t)stream>>>(p);
    else normact_small_bwd_kernel<bf16_t, false><<<grid, dim3(kSmallThreads), 0, (hipStream_t)stream>>>(p);
  }
  return mi355_check_launch("normact_small_bwd");
}

int mi355_maxpool2_fwd(const void* x, int32_t ldx, void* y, int32_t ldy, int32_t n, int32_t c, int32_t d, int32_t h,
                       int32_t w, int32_t dtype, void* stream) {
  return mi355_maxpool2_fwd_idx(x, ldx, y, ldy, nullptr, n, c, d, h, w, dtype, stream);
}

int mi355_maxpool2_fwd_idx(const void* x, int32_t ldx, void* y, int32_t ldy, uint8_t* idx, int32_t n, int32_t c, int32_t d,
                           int32_t h, int32_t w, int32_t dtype, void* stream) {
  MI355_REQUIRE(x && y && n > 0, "maxpool_fwd: bad argument");
  MI355_REQUIRE(d >= 2 && h >= 2 && w >= 2, "maxpool: extents must be >= 2 (%d,%d,%d)", d, h, w);   // odd: floor, as MaxPool3d(2)
  int rc = check_rows(c, ldx, dtype, "maxpool_fwd");
  if (rc) return rc;
  rc = check_rows(c, ldy, dtype, "maxpool_fwd");
  if (rc) return rc;
  const int epv = dtype == MI355_DT_F32 ? 4 : 8;
  const long long total = (long long)n * (d / 2) * (h / 2) * (w / 2) * (c / epv);
  dim3 grid((unsigned)((total + 255) / 256));
  if (dtype == MI355_DT_F32)
    hipLaunchKernelGGL(maxpool_fwd_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, (const float*)x, ldx, (float*)y, ldy, c, d, h, w, total, idx);
  else
    hipLaunchKernelGGL(maxpool_fwd_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, ldx, (bf16_t*)y, ldy, c, d, h, w, total, idx);
  return mi355_check_launch("maxpool_fwd");
}

static int maxpool_bwd_impl(const void* x, int32_t ldx, const void* y, int32_t ldy, const void* dy, int32_t lddy, void* dx,
                            int32_t lddx, const void* add, int32_t ldadd, int32_t n, int32_t c, int32_t d, int32_t h,
                            int32_t w, int32_t dtype, void* stream) {
  MI355_REQUIRE(x && y && dy && dx && n > 0, "maxpool_bwd: bad argument");
  MI355_REQUIRE(d >= 2 && h >= 2 && w >= 2, "maxpool: extents must be >= 2 (%d,%d,%d)", d, h, w);
  const bool odd = (d | h | w) & 1;
  int rc = check_rows(c, ldx, dtype, "maxpool_bwd");
  if (rc) return rc;
  if ((rc = check_rows(c, ldy, dtype, "maxpool_bwd"))) return rc;
  if ((rc = check_rows(c, lddy, dtype, "maxpool_bwd"))) return rc;
  if ((rc = check_rows(c, lddx, dtype, "maxpool_bwd"))) return rc;
  if (add && (rc = check_rows(c, ldadd, dtype, "maxpool_bwd"))) return rc;
  const int epv = dtype == MI355_DT_F32 ? 4 : 8;
  const long long total = (long long)n * (d / 2) * (h / 2) * (w / 2) * (c / epv);
  dim3 grid((unsigned)((total + 255) / 256));
#define MP_BWD(T, O) hipLaunchKernelGGL((maxpool_bwd_kernel<T, O>), grid, dim3(256), 0, (hipStream_t)stream, (const T*)x, ldx, (const T*)y, ldy, (const T*)dy, lddy, (T*)dx, lddx, c, d, h, w, total, (const T*)add, ldadd)
  if (dtype == MI355_DT_F32) { if (odd) MP_BWD(float, true); else MP_BWD(float, false); }
  else { if (odd) MP_BWD(bf16_t, true); else MP_BWD(bf16_t, false); }
#undef MP_BWD
  return mi355_check_launch("maxpool_bwd");
}

int mi355_maxpool2_bwd(const void* x, int32_t ldx, const void* y, int32_t ldy, const void* dy, int32_t lddy, void* dx,
                       int32_t lddx, int32_t n, int32_t c, int32_t d, int32_t h, int32_t w, int32_t dtype, void* stream) {
  return maxpool_bwd_impl(x, ldx, y, ldy, dy, lddy, dx, lddx, nullptr, 0, n, c, d, h, w, dtype, stream);
}

int mi355_maxpool2_bwd_add(const void* x, int32_t ldx, const void* y, int32_t ldy, const void* dy, int32_t lddy, void* dx,
                           int32_t lddx, const void* add, int32_t ldadd, int32_t n, int32_t c, int32_t d, int32_t h,
                           int32_t w, int32_t dtype, void* stream) {
  MI355_REQUIRE(add, "maxpool_bwd_add: null pointer");
  return maxpool_bwd_impl(x, ldx, y, ldy, dy, lddy, dx, lddx, add, ldadd, n, c, d, h, w, dtype, stream);
}

int32_t mi355_l1_blocks(int64_t count) { return (int32_t)((count + kL1PerBlock - 1) / kL1PerBlock); }

int mi355_l1_fwd(const float* a, const float* b, int64_t count, float* partials, float* out, void* stream) {
  MI355_REQUIRE(a && b && partials && out && count > 0, "l1_fwd: bad argument");
  const int nb = mi355_l1_blocks(count);
  hipLaunchKernelGGL(l1_partial_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, a, b, (long long)count, partials);
  hipLaunchKernelGGL(l1_final_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, partials, nb, (long long)count, out);
  return mi355_check_launch("l1_fwd");
}

int mi355_l1_partials(const float* a, const float* b, int64_t count, float* partials, void* stream) {
  MI355_REQUIRE(a && b && partials && count > 0, "l1_partials: bad argument");
  hipLaunchKernelGGL(l1_partial_kernel, dim3(mi355_l1_blocks(count)), dim3(256), 0, (hipStream_t)stream, a, b, (long long)count, partials);
  return mi355_check_launch("l1_partials");
}

int mi355_gan_gen_loss_fwd(const float* logits, int32_t n, const float* l1_partials, int32_t n_partials, int64_t count,
                           float recon_divisor, float recon_factor, float* out4, void* stream) {
  MI355_REQUIRE(logits && l1_partials && out4 && n > 0 && n_partials > 0 && count > 0 && recon_divisor != 0.f, "gan_gen_loss_fwd: bad argument");
  hipLaunchKernelGGL(gan_gen_loss_fwd_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, logits, (int)n, l1_partials, (int)n_partials,
                     (long long)count, recon_divisor, recon_factor, out4);
  return mi355_check_launch("gan_gen_loss_fwd");
}

int mi355_gan_gen_loss_bwd(const float* logits, int32_t n, const float* upstream, float recon_divisor, float recon_factor,
                           float* dlogits, float* l1_gscale, void* stream) {
  MI355_REQUIRE(logits && upstream && dlogits && l1_gscale && n > 0 && recon_divisor != 0.f, "gan_gen_loss_bwd: bad argument");
  hipLaunchKernelGGL(gan_gen_loss_bwd_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, logits, (int)n, upstream, recon_divisor,
                     recon_factor, dlogits, l1_gscale);
  return mi355_check_launch("gan_gen_loss_bwd");
}

int mi355_gan_discr_loss_fwd(const float* logits_fake, int32_t n_fake, const float* logits_real, int32_t n_real, float* out1,
                             void* stream) {
  MI355_REQUIRE(logits_fake && logits_real && out1 && n_fake > 0 && n_real > 0, "gan_discr_loss_fwd: bad argument");
  hipLaunchKernelGGL(gan_discr_loss_fwd_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, logits_fake, (int)n_fake, logits_real,
                     (int)n_real, out1);
  return mi355_check_launch("gan_discr_loss_fwd");
}

int mi355_gan_discr_loss_bwd(const float* logits_fake, int32_t n_fake, const float* logits_real, int32_t n_real,
                             const float* upstream, float* dfake, float* dreal, void* stream) {
  MI355_REQUIRE(logits_fake && logits_real && upstream && dfake && dreal && n_fake > 0 && n_real > 0, "gan_discr_loss_bwd: bad argument");
  hipLaunchKernelGGL(gan_discr_loss_bwd_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, logits_fake, (int)n_fake, logits_real,
                     (int)n_real, upstream, dfake, dreal);
  return mi355_check_launch("gan_discr_loss_bwd");
}

int mi355_l1_bwd(const float* a, const float* b, int64_t count, const float* gscale, float* da, void* stream) {
  MI355_REQUIRE(a && b && gscale && da && count > 0, "l1_bwd: bad argument");
  long long nb = (count + 256 * 8 - 1) / (256 * 8);
  if (nb > 4096) nb = 4096;
  hipLaunchKernelGGL(l1_bwd_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, a, b, (long long)count, gscale, da);
  return mi355_check_launch("l1_bwd");
}

int mi355_adamw_multi(const void* const* ptrs, const int64_t* sizes, int32_t ntensors, float lr, float beta1,
                      float beta2, float eps, float weight_decay, const int64_t* step_dev, int64_t step,
                      void* stream) {
  MI355_REQUIRE(ptrs && sizes && ntensors > 0 && (step_dev || step >= 1), "adamw: bad argument");
  for (int base = 0; base < ntensors; base += kAdamChunk) {
    const int cnt = ntensors - base < kAdamChunk ? ntensors - base : kAdamChunk;
    AdamChunk c;
    long long mx = 0;
    for (int i = 0; i < kAdamChunk; ++i) {
      const int t = base + (i < cnt ? i : 0);
      c.p[i] = (float*)ptrs[4 * t]; c.g[i] = (const float*)ptrs[4 * t + 1];
      c.m[i] = (float*)ptrs[4 * t + 2]; c.v[i] = (float*)ptrs[4 * t + 3];
      c.n[i] = i < cnt ? sizes[t] : 0;
      MI355_REQUIRE(i >= cnt || (c.p[i] && c.g[i] && c.m[i] && c.v[i] && c.n[i] > 0), "adamw: null tensor %d", t);
      if (c.n[i] > mx) mx = c.n[i];
    }
    long long nb = (mx + 256 * 4 - 1) / (256 * 4);
    if (nb > 1024) nb = 1024;
    hipLaunchKernelGGL(adamw_kernel, dim3((unsigned)nb, cnt), dim3(256), 0, (hipStream_t)stream, c, lr, (double)beta1,
                       (double)beta2, eps, weight_decay, (const long long*)step_dev, (long long)step);
  }
  return mi355_check_launch("adamw");
}

int mi355_mfma_selftest(float* out_f32_1024, float* out_bf16_1024, void* stream) {
  MI355_REQUIRE(out_f32_1024 && out_bf16_1024, "selftest: null pointer");
  hipLaunchKernelGGL(mfma_selftest_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, out_f32_1024, out_bf16_1024);
  return mi355_check_launch("selftest");
}

}  // extern "C"

namespace {
// ------------------------------------------------------------------ fp8 operand preparation
__global__ __launch_bounds__(256) void amax_f32_kernel(const float* __restrict__ x, long long n, float* out) {
  float m = 0.f;
  const long long stride = (long long)gridDim.x * 256 * 4;
  for (long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += stride) {
    if (i + 4 <= n) {
      const float4 v = *reinterpret_cast<const float4*>(x + i);
      m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
    } else {
      for (long long j = i; j < n; ++j) m = fmaxf(m, fabsf(x[j]));
    }
  }
  amax_commit(m, out);
}
template <typename T>
__global__ __launch_bounds__(256) void amax_act_kernel(const T* __restrict__ x, int ld, int c, long long rows, float* out) {
  constexpr int EPV = Elem<T>::kPer16B;
  const int lpr = c / EPV;
  const long long total = rows * lpr, stride = (long long)gridDim.x * 256;
  float m = 0.f;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += stride) {
    const long long row = i / lpr;
    const int piece = (int)(i - row * lpr);
    Vec16<T> v;
    v.load(x + row * ld + piece * EPV);
#pragma unroll
    for (int j = 0; j < EPV; ++j) m = fmaxf(m, fabsf(v.f[j]));
  }
  amax_commit(m, out);
}
// 16 channels per thread: two 16-B loads of bf16 (four of f32), one 16-B store of e4m3
template <typename T>
__global__ __launch_bounds__(256) void cast_fp8_kernel(const T* __restrict__ x, int ld, int c, long long rows,
                                                        const float* __restrict__ amax, uint8_t* __restrict__ dst, int ld_dst,
                                                        float* __restrict__ next) {
  constexpr int EPV = Elem<T>::kPer16B, NV = 16 / EPV;
  const float sc = fp8_scale_of(amax);
  const int gpr = c / 16;
  const long long total = rows * gpr, stride = (long long)gridDim.x * 256;
  float m = 0.f;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += stride) {
    const long long row = i / gpr;
    const int g = (int)(i - row * gpr);
    float f[16];
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      Vec16<T> v;
      v.load(x + row * ld + g * 16 + k * EPV);
#pragma unroll
      for (int j = 0; j < EPV; ++j) {
        m = fmaxf(m, fabsf(v.f[j]));
        f[k * EPV + j] = v.f[j] * sc;
      }
    }
    uint32_t w[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      w[k] = cvt_pk_fp8(f[4 * k], f[4 * k + 1], 0u, false);
      w[k] = cvt_pk_fp8(f[4 * k + 2], f[4 * k + 3], w[k], true);
    }
    *reinterpret_cast<uint4*>(dst + row * ld_dst + g * 16) = make_uint4(w[0], w[1], w[2], w[3]);
  }
  if (next) amax_commit(m, next);
}
// delayed scaling: the amax gathered during a step becomes the scale of the next one
// sat (optional, int32 per slot): raised when the step that ends here SATURATED in that slot -- a value clamps at +-448 exactly
// when |value| * 224 / amax_in_use > 448, i.e. when the amax gathered during the step exceeds twice the amax in use (every
// kernel that casts an operand also gathers its maximum), so the roll sees it without any counter in the cast kernels
__global__ void fp8_scale_roll_kernel(float* __restrict__ table, int n, int* __restrict__ sat) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float use = table[2 * i], nx = table[2 * i + 1];
  if (sat && use > 0.f && nx > 2.f * use) sat[i] += 1;
  if (nx > 0.f) table[2 * i] = nx;
  table[2 * i + 1] = 0.f;
}
// D[i][j] = sum_k A[i][k] B[k][j], 32 x 32 x 64, A[i][k] = ((i + k) % 5) - 2, B[k][j] = ((2 k + j) % 7) - 3 (exact in e4m3)
__global__ void fp8_selftest_kernel(float* out) {
  typedef int i32x8 __attribute__((ext_vector_type(8)));
  const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
  uint32_t wa[8], wb[8];
#pragma unroll
  for (int w = 0; w < 8; ++w) {
    float fa[4], fb[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int k = 32 * h + 4 * w + b;
      fa[b] = (float)(((r + k) % 5) - 2);
      fb[b] = (float)(((2 * k + r) % 7) - 3);
    }
    wa[w] = cvt_pk_fp8(fa[2], fa[3], cvt_pk_fp8(fa[0], fa[1], 0u, false), true);
    wb[w] = cvt_pk_fp8(fb[2], fb[3], cvt_pk_fp8(fb[0], fb[1], 0u, false), true);
  }
  const i32x8 a = {(int)wa[0], (int)wa[1], (int)wa[2], (int)wa[3], (int)wa[4], (int)wa[5], (int)wa[6], (int)wa[7]};
  const i32x8 b = {(int)wb[0], (int)wb[1], (int)wb[2], (int)wb[3], (int)wb[4], (int)wb[5], (int)wb[6], (int)wb[7]};
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
#pragma unroll
  for (int i = 0; i < 16; ++i) out[acc_row(i, h) * 32 + r] = acc[i];
}
}  // namespace

extern "C" {

int mi355_amax_f32(const float* x, int64_t n, float* amax, void* stream) {
  MI355_REQUIRE(x && amax && n > 0, "amax_f32: bad argument");
  if (hipMemsetAsync(amax, 0, 4, (hipStream_t)stream) != hipSuccess) { mi355_set_error("amax: memset failed"); return MI355_ERR_HIP; }
  long long nb = (n + 1023) / 1024;
  if (nb > 1024) nb = 1024;
  hipLaunchKernelGGL(amax_f32_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, x, (long long)n, amax);
  return mi355_check_launch("amax_f32");
}

int mi355_amax_act(const void* x, int32_t ld, int32_t c, int64_t rows, int32_t dtype, float* amax, void* stream) {
  MI355_REQUIRE(x && amax && rows > 0, "amax_act: bad argument");
  int rc = check_rows(c, ld, dtype, "amax_act");
  if (rc) return rc;
  if (hipMemsetAsync(amax, 0, 4, (hipStream_t)stream) != hipSuccess) { mi355_set_error("amax: memset failed"); return MI355_ERR_HIP; }
  const int epv = dtype == MI355_DT_F32 ? 4 : 8;
  long long nb = (rows * (c / epv) + 256 * 8 - 1) / (256 * 8);
  if (nb > 2048) nb = 2048;
  if (nb < 1) nb = 1;
  if (dtype == MI355_DT_F32) hipLaunchKernelGGL(amax_act_kernel<float>, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, (const float*)x, ld, c, (long long)rows, amax);
  else hipLaunchKernelGGL(amax_act_kernel<bf16_t>, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, ld, c, (long long)rows, amax);
  return mi355_check_launch("amax_act");
}

int mi355_cast_fp8(const void* src, int32_t ld_src, int32_t c, int64_t rows, int32_t src_dtype, const float* amax,
                   void* dst, int32_t ld_dst, void* stream) {
  return mi355_cast_fp8_delayed(src, ld_src, c, rows, src_dtype, amax, nullptr, dst, ld_dst, stream);
}

int mi355_fp8_scale_roll(float* table, int32_t n, int32_t* sat, void* stream) {
  MI355_REQUIRE(table && n > 0, "fp8_scale_roll: bad argument");
  hipLaunchKernelGGL(fp8_scale_roll_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, table, (int)n, (int*)sat);
  return mi355_check_launch("fp8_scale_roll");
}

int mi355_cast_fp8_delayed(const void* src, int32_t ld_src, int32_t c, int64_t rows, int32_t src_dtype, const float* amax,
                           float* amax_next, void* dst, int32_t ld_dst, void* stream) {
  MI355_REQUIRE(src && dst && amax && rows > 0 && ld_dst >= c && ld_dst % 16 == 0, "cast_fp8: bad argument");
  int rc = check_rows(c, ld_src, src_dtype, "cast_fp8");
  if (rc) return rc;
  long long nb = (rows * (c / 16) + 256 * 4 - 1) / (256 * 4);
  if (nb > 2048) nb = 2048;
  if (nb < 1) nb = 1;
  if (src_dtype == MI355_DT_F32) hipLaunchKernelGGL(cast_fp8_kernel<float>, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, (const float*)src, ld_src, c, (long long)rows, amax, (uint8_t*)dst, ld_dst, amax_next);
  else hipLaunchKernelGGL(cast_fp8_kernel<bf16_t>, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)src, ld_src, c, (long long)rows, amax, (uint8_t*)dst, ld_dst, amax_next);
  return mi355_check_launch("cast_fp8");
}

int mi355_fp8_selftest(float* out_1024, void* stream) {
  MI355_REQUIRE(out_1024, "fp8_selftest: null pointer");
  hipLaunchKernelGGL(fp8_selftest_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, out_1024);
  return mi355_check_launch("fp8_selftest");
}

}  // extern "C"
