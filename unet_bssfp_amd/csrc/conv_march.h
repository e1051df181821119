// conv_march_kernel<F8>: 3x3x3 stride-1 convolution of the full-resolution layers with <= 32 input channels
// (U-Net conv_0.*, upcat_1.conv_1 and their data gradients: 32 -> 32 / 96 output channels at 128^3 / 160^3),
// bf16 operands (F8 = false) or OCP e4m3 operands on the block-scaled MFMA (F8 = true, BASELINE.json configs[4]).
//
// Why another structure (profiles/r01_pmc_conv_ru.txt): conv_ru_kernel stages the 6x6x34 halo of one 16-channel chunk
// per pass -- 2.39x the tile's own voxels, half of every 128-B line per pass -- and the L2 keeps neither the halo
// overlap nor the line until the second chunk pass: 4.0x the input tensor crossed the fabric, so the kernel ran at
// the memory system's pace (691 MB in 155 us), not the MFMA's.  Here a workgroup owns a 16 (h) x 32 (w) footprint and
// MARCHES along d:
//   * each input plane of the footprint (18 x 34 voxels, ALL 32 channels: whole voxel rows) is brought into LDS exactly
//     once per workgroup with `buffer_load_dwordx4 ... lds` (1 KB per instruction), while the previous plane is being
//     consumed (two plane buffers, one barrier per plane): halo re-reads 1.20x in (h, w) and (L + 2) / L along d for a
//     segment of L planes -- 1.35x instead of 4.0x;
//   * input-stationary: an input plane feeds the three output planes d-1, d, d+1 (kd = 2, 1, 0), whose accumulators
//     (3 planes x 4 rows x 32 voxels x 32 channels per wave = 192 registers) stay in registers; a plane is finished,
//     converted and stored when its kd = 2 contribution is in; the register sets rotate by unrolling the march by 3;
//   * the packed weights of the workgroup's 32 output channels (54 KB bf16 / 30 KB fp8) are loaded into LDS ONCE; a
//     weight fragment read from LDS feeds 4 rows;
//   * the weights are the MFMA's A operand (rows = output channels), the activations its B operand (columns = voxels),
//     and the weight rows are permuted so that a lane ends up with 16 CONTIGUOUS channels of one voxel: the epilogue is
//     two 16-byte stores per lane and row, no transposition;
//   * 4 waves = one per SIMD (launch bound 1: the 512-entry register file is this wave's), one workgroup per CU; the
//     segment length is chosen so that the grid is a whole number of 256-workgroup rounds (conv_api.hip);
//   * the per-channel statistics of the whole segment are accumulated in registers and written once.
//
// bf16: per (kd) block 2 chunks x 3 kw "groups" of {3 weight fragments (kh), 6 activation fragments (halo rows), 12
// v_mfma_f32_32x32x16_bf16}: an activation fragment feeds the 3 kh taps (row reuse), 0.75 ds_read_b128 per MFMA.
// fp8 : K = 64 per v_mfma_scale_f32_32x32x64_f8f6f4 = TWO taps x 32 channels (lane half h holds tap 2 pair + h), so the 9
// (kh, kw) taps of a kd block are 5 pairs (the 10th tap has zero weights): per pair one weight fragment (32 B per lane)
// and, per output row, one activation fragment whose lane halves read two different halo positions; 15 MFMAs of 64
// cycles per row and plane instead of 54 of 32.  Block scales are 1 (E8M0 127); the per-TENSOR scales of the operands
// (x8 = x * 224 / amax(x), likewise the weights) are divided out in the epilogue.
//
// LDS plane image: voxel-major; 16-B slot (q ^ swz(col)) of voxel (row, col) holds channel piece q, swz = (col >> 2) & 3
// for the 4 pieces of a 64-B bf16 voxel, (col >> 3) & 1 for the 2 pieces of a 32-B fp8 voxel: the DMA writes lane-linear,
// so the swizzle sits on the per-lane SOURCE address, and a fragment read (32 consecutive voxels of a row, one piece)
// touches 16 distinct slots mod 16 in every ds_read_b128 lane group.  The swizzle depends on the column only: the
// halo rows a wave reads are one address register plus immediate offsets.
#pragma once
#include "conv_common.h"

#ifndef MARCH_TICKET_FRONT
#define MARCH_TICKET_FRONT (NGR * 2 / 3)
#endif
constexpr int kMarchFH = 16, kMarchFW = 32, kMarchHR = kMarchFH + 2, kMarchHC = kMarchFW + 2;
constexpr int kMarchVox = kMarchHR * kMarchHC;                       // 612 halo voxels per plane

template <bool F8> struct MarchCfg {
  static constexpr int VB = F8 ? 32 : 64;                            // bytes per voxel (32 channels)
  static constexpr int PIECES = VB / 16;                             // 16-B pieces per voxel
  static constexpr int BLOCKS = F8 ? 20 : 40;                        // 1-KB DMA instructions per plane: a multiple of 4, so that
                                                                     // every wave issues the same number (no branch in a step)
  static constexpr int PLANE = BLOCKS * 1024;
  static constexpr int NWI = F8 ? 30 : 54;                           // 1-KB weight blocks
  static constexpr int WBYTES = NWI * 1024;
  static constexpr int LDS = 2 * PLANE + WBYTES + 4096 + (BLOCKS / 4) * 1024;   // + statistics scratch, bias; DMA offset table
};

struct MarchArgs { int seg_len, nseg, tiles_h, tiles_w; const float* amax_x; const float* amax_w; };

typedef int i32x8 __attribute__((ext_vector_type(8)));

template <bool F8>
__global__ __launch_bounds__(256, 1) void conv_march_kernel(const ConvArgs a, const MarchArgs m) {
  using T = bf16_t;                                                  // output element
  using Cfg = MarchCfg<F8>;
  constexpr int HC = kMarchHC, NI = Cfg::BLOCKS / 4, NW = (Cfg::NWI + 3) / 4, VB = Cfg::VB, XB = F8 ? 1 : 2;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef __attribute__((address_space(3))) void* lds_ptr;
  char* const wl = smem + 2 * Cfg::PLANE;
  char* const patch = wl + Cfg::WBYTES;

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  const int co_base = blockIdx.y * 32;
  int tile;
  {
    const int nwg = gridDim.x, q = nwg >> 3, rem = nwg & 7, xcd = blockIdx.x & 7, k = blockIdx.x >> 3;
    tile = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + k;
  }
  // tile -> (sample, d-segment, footprint): segments slowest, so that an XCD's contiguous tile range is a d-slab whose
  // footprints share their (h, w) halo columns in that XCD's L2
  const int per_seg = m.tiles_h * m.tiles_w, per_sample = per_seg * m.nseg;
  const int tn = tile / per_sample;
  int t = tile - tn * per_sample;
  const int seg = t / per_seg;
  t -= seg * per_seg;
  const int th_i = t / m.tiles_w, tw_i = t - th_i * m.tiles_w;
  const int d0 = seg * m.seg_len, d1 = min(a.do_, d0 + m.seg_len);     // output planes [d0, d1)
  const int h0 = th_i * kMarchFH, w0 = tw_i * kMarchFW;

  // ---- DMA set-up: instruction id = i * 4 + wave covers 64 / PIECES halo voxels, lane = (voxel, slot).  The per-lane
  //      source offsets live in LDS (one ds_read_b32 each per plane): as registers they were the ones hipcc spilled to
  //      scratch, and a scratch reload at the top of every step drags a vmcnt(0) in front of the DMA issue.
  int* const vtab = reinterpret_cast<int*>(patch + 4096) + tid;
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int id = i * 4 + wave, v = id * (64 / Cfg::PIECES) + lane / Cfg::PIECES;
    const int hy = v / HC, hx = v - hy * HC;
    const int swz = F8 ? ((hx >> 3) & 1) : ((hx >> 2) & 3);
    const int q = (lane % Cfg::PIECES) ^ swz;                        // the channel piece this slot holds
    const int gh = h0 - a.ph + hy, gw = w0 - a.pw + hx;
    const bool ok = v < kMarchVox && gh >= 0 && gh < a.hi && gw >= 0 && gw < a.wi;
    vtab[i * 256] = ok ? ((gh * a.wi + gw) * a.ld0 * XB + q * 16) : (int)0x80000000;
  }
  const long long nvox = (long long)a.n * a.di * a.hi * a.wi;
  // (the copies are issued as inline assembly -- common.h, dma_lds_b128: through the builtin hipcc waits for ALL of them
  //  before the next ds_read of any buffer, and the plane in flight never overlapped the MFMAs)
  const dma_rsrc_t rsx = dma_rsrc(a.x0, ((nvox - 1) * a.ld0 + a.c0) * XB), rsw = dma_rsrc(a.wp, 54ll * a.coutp * 16 * XB);
  const int plane_stride = a.hi * a.wi * a.ld0 * XB;                  // bytes per input plane
  auto load_plane = [&](int p) __attribute__((always_inline)) {       // p: input plane index (outside [0, D): zeros)
    const bool pin = p >= 0 && p < a.di;
    const int soff = pin ? (tn * a.di + p) * plane_stride : 0;
    const int kill = pin ? 0 : (int)0x80000000;                       // OR-ed in: out of range -> the DMA writes zeros (no branch)
    char* dst = smem + (p & 1) * Cfg::PLANE + wave * 1024;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      // (bf16: the select form, which hipcc turns into short branches, keeps the table reads next to their DMA -- hoisted
      //  together as in the OR form they cost 10 registers this variant does not have)
      const int voff = F8 ? (vtab[i * 256] | kill) : (pin ? vtab[i * 256] : (int)0x80000000);
      dma_lds_b128(rsx, dst + i * 4096, voff, soff);
    }
  };
  // ---- weights of this workgroup's 32 output channels, one 1-KB block = [lane half][row] x 16 B per fragment piece.  Row
  //      rho of a fragment holds output channel pi(rho) = 16 * ((rho >> 2) & 1) + 4 * (rho >> 3) + (rho & 3), which makes
  //      the 16 accumulator registers of lane half h' the contiguous channels 16 h' .. 16 h' + 15.  (The permutation costs
  //      nothing: it sits on the per-lane source address of the LDS-DMA.)
  const int wrow = 16 * ((r >> 2) & 1) + 4 * (r >> 3) + (r & 3);
#pragma unroll
  for (int i = 0; i < NW; ++i) {
    const int j = i * 4 + wave;
    if (j < Cfg::NWI) {
      int src;
      if constexpr (F8) {      // block j = (kd * 5 + pair) * 2 + piece: lane half h holds tap 2 pair + h of the kd plane (tap 9: zeros)
        const int piece = j & 1, pr = (j >> 1) % 5, kd = (j >> 1) / 5, tp = 2 * pr + h;
        src = tp < 9 ? ((piece * 27 + kd * 9 + tp) * a.coutp + co_base + wrow) * 16 : (int)0x80000000;
      } else {                 // block j = chunk * 27 + tap: lane half h holds channels 8 h .. 8 h + 7 of the chunk
        src = ((j * a.coutp + co_base + wrow) * 2 + h) * 16;
      }
      dma_lds_b128(rsw, wl + j * 1024, src, 0);
    }
  }
  load_plane(d0 - 1);

  // ---- per-lane LDS offsets of the activation fragments (halo row = 4 wave + immediate)
  //      bf16: [kw][chunk] -> voxels kw + r, piece 2 chunk + h;  fp8: [pair][piece] -> lane half h reads tap 2 pair + h
  int aoff[F8 ? 5 : 3][2];
  if constexpr (F8) {
#pragma unroll
    for (int pr = 0; pr < 5; ++pr) {
      const int tp = min(2 * pr + h, 8), kh = tp / 3, kw = tp - kh * 3;      // (pair 4, h = 1: any valid address, its weights are zero)
      const int col = kw + r, v = (4 * wave + kh) * HC + col, s = (col >> 3) & 1;
      aoff[pr][0] = (2 * v + s) * 16;
      aoff[pr][1] = (2 * v + (1 ^ s)) * 16;
    }
  } else {
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
      const int col = kw + r, v = 4 * wave * HC + col, s = (col >> 2) & 3;
#pragma unroll
      for (int c = 0; c < 2; ++c) aoff[kw][c] = (4 * v + ((2 * c + h) ^ s)) * 16;
    }
  }
  const char* const wlane = wl + lane * 16;

  // this lane: voxel w0 + r of a row, output channels co_base + 16 h .. + 15
  const int cch = co_base + 16 * h;
  // fp8: the operands are x * sx and w * sw with per-tensor s = 224 / amax; the accumulators are divided by sx * sw
  float deq = 1.f;
  if constexpr (F8) {
    const float ax = m.amax_x[0], aw = m.amax_w[0];
    deq = (ax > 0.f ? ax * (1.f / 224.f) : 1.f) * (aw > 0.f ? aw * (1.f / 224.f) : 1.f);
  }
  // The accumulators start at zero and hold z - bias (fp8: in units of deq); the fused statistics are taken of exactly
  // that (no after-the-fact bias correction, which cancels when |bias| >> sigma(z - bias): a loaded checkpoint), and the
  // bias is added where a finished row is converted for the store (16 values read from LDS, nothing resident).
  float s1[16], s2[16];
  float* const blds = reinterpret_cast<float*>(patch) + 512;          // [32] bias of this workgroup's channels
  if (tid < 32) blds[tid] = (a.bias && co_base + tid < a.nbias) ? a.bias[co_base + tid] : 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) { s1[i] = 0.f; s2[i] = 0.f; }
  const bool vox_ok = w0 + r < a.wo;
  const bool st0 = vox_ok && cch + 8 <= a.cstore, st1 = vox_ok && cch + 16 <= a.cstore;

  auto pack_row = [&](const f32x16& s, uint32_t (&w)[8], const bool stats) __attribute__((always_inline)) {
    const float4* bp = reinterpret_cast<const float4*>(blds + 16 * h);
#pragma unroll
    for (int i4 = 0; i4 < 4; ++i4) {
      const float4 bq = bp[i4];
      const float bb[4] = {bq.x, bq.y, bq.z, bq.w};
#pragma unroll
      for (int j = 0; j < 4; j += 2) {
        const int i = 4 * i4 + j;
        const float v0 = s[i], v1 = s[i + 1];                         // accumulator units (fp8: (z - bias) / deq); sums are rescaled once, at the end
#ifndef MARCH_DIAG_NO_STATS
        if (stats) { s1[i] += v0; s2[i] += v0 * v0; s1[i + 1] += v1; s2[i + 1] += v1 * v1; }
#endif
        const float z0 = F8 ? fmaf(v0, deq, bb[j]) : v0 + bb[j], z1 = F8 ? fmaf(v1, deq, bb[j + 1]) : v1 + bb[j + 1];
        w[i >> 1] = (uint32_t)f32_to_bf16_bits(z0) | ((uint32_t)f32_to_bf16_bits(z1) << 16);
      }
    }
  };
  // ---- epilogue of one finished output plane q (generic form): bf16, two 16-byte stores per lane and row
  auto store_plane = [&](f32x16 (&s)[4], int q) __attribute__((always_inline)) {
#pragma unroll
    for (int row = 0; row < 4; ++row) {
      const int gh = h0 + 4 * wave + row;
      if (gh >= a.ho) break;                                          // wave-uniform
      T* dst = reinterpret_cast<T*>(a.y) + ((((long long)tn * a.dy + q) * a.hy + gh) * a.wy + w0 + r) * a.ldy + cch;
      uint32_t w[8];
      pack_row(s[row], w, vox_ok);
      if (st0) *reinterpret_cast<uint4*>(dst) = make_uint4(w[0], w[1], w[2], w[3]);
      if (st1) *reinterpret_cast<uint4*>(dst + 8) = make_uint4(w[4], w[5], w[6], w[7]);
    }
  };
  auto init_row = [&](f32x16& s) __attribute__((always_inline)) {    // first contribution to an output row: C = 0
#pragma unroll
    for (int i = 0; i < 16; ++i) s[i] = 0.f;
  };

  // ---- fragment groups.  The NEXT group's reads are issued ahead of this group's MFMAs (one wave per SIMD: nothing else
  //      hides the LDS latency); hipcc's scheduler would sink every read next to its first use, so the order is pinned
  //      with sched_group_barrier.
  //      bf16 group gi (6 per kd block) = (chunk, kw): 3 weight + 6 activation fragments, 12 MFMAs
  //      fp8  group gi (5 per kd block) = tap pair   : 1 weight + 4 activation fragments (2 reads each), 4 MFMAs
  constexpr int NG = F8 ? 5 : 6, NRD = F8 ? 10 : 9, NMM = F8 ? 4 : 12;
  typedef int i32x4 __attribute__((ext_vector_type(4)));
  struct Group { uint4 b[3], x[6]; i32x8 w8, x8[4]; };                // (bf16: b / x; fp8: w8 / x8 -- the unused members vanish)
  auto ld32 = [&](const char* p0, const char* p1) __attribute__((always_inline)) {   // two 16-B pieces -> one 32-B MFMA operand
    const i32x4 lo = *reinterpret_cast<const i32x4*>(p0), hi = *reinterpret_cast<const i32x4*>(p1);
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  };
  auto load_group = [&](Group& g, const char* pl, const int kd, const int gi) __attribute__((always_inline)) {
    if constexpr (F8) {
      g.w8 = ld32(wlane + ((kd * 5 + gi) * 2) * 1024, wlane + ((kd * 5 + gi) * 2 + 1) * 1024);
#pragma unroll
      for (int row = 0; row < 4; ++row) g.x8[row] = ld32(pl + aoff[gi][0] + row * (HC * VB), pl + aoff[gi][1] + row * (HC * VB));
    } else {
      const int c = gi / 3, kw = gi - c * 3;
#pragma unroll
      for (int kh = 0; kh < 3; ++kh) g.b[kh] = *reinterpret_cast<const uint4*>(wlane + (c * 27 + kd * 9 + kh * 3 + kw) * 1024);
#pragma unroll
      for (int hy = 0; hy < 6; ++hy) g.x[hy] = *reinterpret_cast<const uint4*>(pl + aoff[kw][c] + hy * (HC * VB));
    }
  };
  auto mma_group = [&](const Group& g, f32x16 (&s)[4], const bool fresh) __attribute__((always_inline)) {
    if constexpr (F8) {
#pragma unroll
      for (int row = 0; row < 4; ++row) {
        if (fresh) init_row(s[row]);
        s[row] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(g.w8, g.x8[row], s[row], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
      }
    } else {
#pragma unroll
      for (int hy = 0; hy < 6; ++hy)
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
          const int row = hy - kh;
          if (row >= 0 && row < 4) {
            if (fresh && kh == 0) init_row(s[row]);
            s[row] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, g.b[kh]), __builtin_bit_cast(bf16x8, g.x[hy]),
                                                             s[row], 0, 0, 0);              // rows = output channels, columns = voxels
          }
        }
    }
  };
  auto pin_pipeline = [&](const int ngroups) __attribute__((always_inline)) {
#ifdef MARCH_F8_NOPIN
    if constexpr (F8) return;
#endif
    __builtin_amdgcn_sched_group_barrier(0x100, NRD, 0);
#pragma unroll
    for (int gi = 0; gi < ngroups; ++gi) {
#pragma unroll
      for (int k = 0; k < NMM; ++k) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        if (gi + 1 < ngroups) {
          if constexpr (F8) {
            if (k < 2) __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
            else __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
          } else if (k < NRD) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
      }
    }
  };
  auto block = [&](const char* pl, f32x16 (&s)[4], const int kd, const bool fresh) __attribute__((always_inline)) {
    Group g[2];
    load_group(g[0], pl, kd, 0);
#pragma unroll
    for (int gi = 0; gi < NG; ++gi) {
      if (gi + 1 < NG) load_group(g[(gi + 1) & 1], pl, kd, gi + 1);
      mma_group(g[gi & 1], s, fresh && gi == 0);
    }
    pin_pipeline(NG);
  };
  // ---- the steady-state step (all three output planes inside the segment, footprint inside the volume) as ONE basic
  //      block: the next plane's LDS-DMA, 3 x NG chained fragment groups (kd = 2, 1, 0) and the epilogue of the plane that
  //      kd = 2 completes (buffer stores: lanes that must not store get an out-of-range offset, no branch).
  const auto rsy = __builtin_amdgcn_make_buffer_rsrc((void*)a.y, 0, (int)((((long long)a.n * a.dy * a.hy * a.wy - 1) * a.ldy + a.cstore) * 2), 0x00020000);
  const int yrow = ((4 * wave) * a.wy + w0 + r) * a.ldy * 2 + cch * 2;        // byte offset of this lane's voxel in row 0 of a plane's footprint
  // The next plane's NI copies are issued ONE AT A TIME between the fragment groups (fenced: nothing is scheduled across a
  // copy): issued together at the top of the step they cost the wave their whole issue time with the matrix pipe idle; behind
  // a group, a copy's issue runs under the MFMA in flight.  The table entry of a copy is read one group ahead.  The finished
  // plane's four rows are converted and stored one per group (groups NG .. NG + 3).
  auto full_step = [&](int p, f32x16 (&s_m1)[4], f32x16 (&s_0)[4], f32x16 (&s_p1)[4]) __attribute__((always_inline)) {
    constexpr int NGR = 3 * NG;
    const char* pl = smem + (p & 1) * Cfg::PLANE;
    const bool pin = p + 1 >= 0 && p + 1 < a.di;
    const int soff = pin ? (tn * a.di + p + 1) * plane_stride : 0;
    char* const dst = smem + ((p + 1) & 1) * Cfg::PLANE + wave * 1024;
    const int ybase = ((tn * a.dy + (p - 1)) * a.hy + h0) * a.wy * a.ldy * 2;
    Group g[2];
    load_group(g[0], pl, 2, 0);
    __builtin_amdgcn_sched_group_barrier(0x100, NRD, 0);
    int e_next = vtab[0];
#pragma unroll
    for (int gi = 0; gi < NGR; ++gi) {
      // copy ticket i goes behind group (i * FRONT) / NI: spread over the FIRST part of the step only -- the step ends with a
      // wait for ALL copies (two plane buffers), so the last ticket needs a memory latency of MFMA work behind it
      constexpr int FRONT = MARCH_TICKET_FRONT;
      static_assert(FRONT >= NI && FRONT <= NGR, "one ticket per group at most");
      int ticket = -1;
#pragma unroll
      for (int i = 0; i < NI; ++i) if ((i * FRONT) / NI == gi) ticket = i;
      const int e_cur = e_next;
      if (ticket >= 0 && ticket + 1 < NI) e_next = vtab[(ticket + 1) * 256];
      if (gi + 1 < NGR) load_group(g[(gi + 1) & 1], pl, 2 - (gi + 1) / NG, (gi + 1) % NG);
      if (gi < NG) mma_group(g[gi & 1], s_m1, false);
      else if (gi < 2 * NG) mma_group(g[gi & 1], s_0, false);
      else mma_group(g[gi & 1], s_p1, gi == 2 * NG);
#ifdef MARCH_DIAG_NO_STORE
      if (gi == NG) asm volatile("" :: "v"(s_m1[0][0]), "v"(s_m1[1][5]), "v"(s_m1[2][9]), "v"(s_m1[3][15]));
#else
      if (gi >= NG && gi < NG + 4) {                                    // output plane p - 1 is complete: row gi - NG
        typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
        const int row = gi - NG;
        uint32_t w[8];
        pack_row(s_m1[row], w, true);
        // (the plane's base goes into the VECTOR offset, soffset = 0: with a register soffset hipcc places a VALU write
        //  of the store's data registers right behind a 16-byte store -- on gfx950 the store then read the NEW values
        //  in its last lanes; with a constant soffset the compiler's own hazard rule keeps the wait state)
        const int off = ybase + yrow + row * a.wy * a.ldy * 2;
        u32x4 lo = {w[0], w[1], w[2], w[3]}, hi = {w[4], w[5], w[6], w[7]};
        __builtin_amdgcn_raw_buffer_store_b128(lo, rsy, st0 ? off : (int)0x80000000, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b128(hi, rsy, st1 ? off + 16 : (int)0x80000000, 0, 0);
      }
#endif
      // this group's MFMAs with the next group's reads between them (see pin_pipeline)
#pragma unroll
      for (int k = 0; k < NMM; ++k) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        if (gi + 1 < NGR) {
          if constexpr (F8) {
            if (k < 2) __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
            else __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
          } else if (k < NRD) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
      }
#ifndef MARCH_DIAG_NO_DMA
      if (ticket >= 0) {
        __builtin_amdgcn_sched_barrier(0);
        dma_lds_b128(rsx, dst + ticket * 4096, (pin && e_cur >= 0) ? e_cur : (int)0x80000000, soff);
        __builtin_amdgcn_sched_barrier(0);
      }
#endif
    }
#ifndef MARCH_DIAG_NO_WAIT
    dma_wait_all();
#endif
#ifndef MARCH_DIAG_NO_BARRIER
    __syncthreads();
#endif
  };
  auto step = [&](int p, f32x16 (&s_m1)[4], f32x16 (&s_0)[4], f32x16 (&s_p1)[4]) __attribute__((always_inline)) {
    if (p < d0 - 1 || p > d1) return;                                 // wave-uniform: outside this segment's input planes
#ifndef MARCH_DIAG_NO_DMA
    if (p + 1 <= d1) load_plane(p + 1);                               // lands under this plane's MFMAs
#endif
    const char* pl = smem + (p & 1) * Cfg::PLANE;
    if (p - 1 >= d0) {                                                // kd = 2 completes output plane p - 1
      block(pl, s_m1, 2, false);
#ifndef MARCH_DIAG_NO_STORE
      store_plane(s_m1, p - 1);
#else
      asm volatile("" :: "v"(s_m1[0][0]), "v"(s_m1[1][5]), "v"(s_m1[2][9]), "v"(s_m1[3][15]));
#endif
    }
    if (p >= d0 && p < d1) block(pl, s_0, 1, false);
    if (p + 1 < d1) block(pl, s_p1, 0, true);
    dma_wait_all();               // plane p + 1 has landed (this wave's share) ...
    __syncthreads();              // ... everybody's; every wave is done with plane p's buffer
  };

  dma_wait_all();
  __syncthreads();                // weights and the first plane have landed
  // Two complete marches, never mixed (hipcc's register allocator cannot keep three 64-register accumulator sets in place
  // across a control-flow join of a straight-line step and a branchy one: it copied and spilled whole sets).
  //  * regular segment (footprint inside the volume, L = d1 - d0 >= 5 and L = 2 mod 3): output plane d0 + j lives in set
  //    j mod 3, so the two lead-in steps, the (L - 2) / 3 triples of straight-line full steps and the two lead-out steps
  //    all have compile-time register sets;
  //  * anything else (volume border, last short segment): the generic step with its wave-uniform branches.
  const bool interior = h0 + kMarchFH <= a.ho && w0 + kMarchFW <= a.wo && st1;
  const int slen = d1 - d0;
  if (interior && slen >= 5 && slen % 3 == 2) {
    f32x16 sa[4], sb[4], sc[4];
    int p = d0 - 1;
    load_plane(p + 1);
    block(smem + (p & 1) * Cfg::PLANE, sa, 0, true);                  // input plane d0 - 1 -> output d0
    dma_wait_all();
    __syncthreads();
    ++p;
    load_plane(p + 1);
    block(smem + (p & 1) * Cfg::PLANE, sa, 1, false);                 // input plane d0 -> outputs d0, d0 + 1
    block(smem + (p & 1) * Cfg::PLANE, sb, 0, true);
    dma_wait_all();
    __syncthreads();
    ++p;
    for (int t3 = 0; t3 < (slen - 2) / 3; ++t3) {
      full_step(p, sa, sb, sc);
      full_step(p + 1, sb, sc, sa);
      full_step(p + 2, sc, sa, sb);
      p += 3;
    }
    load_plane(p + 1);                                                // p = d1 - 1 (plane d1 may lie outside: zeros)
    block(smem + (p & 1) * Cfg::PLANE, sa, 2, false);
    store_plane(sa, p - 1);
    block(smem + (p & 1) * Cfg::PLANE, sb, 1, false);
    dma_wait_all();
    __syncthreads();
    ++p;
    block(smem + (p & 1) * Cfg::PLANE, sb, 2, false);                 // p = d1
    store_plane(sb, p - 1);
  } else {
    f32x16 acc[3][4];
    const int first = d0 - 1;
    const int pa = (first >= 0 ? first / 3 : -((-first + 2) / 3)) * 3;     // floor to a multiple of 3
    for (int pb = pa; pb <= d1; pb += 3) {
      step(pb, acc[2], acc[0], acc[1]);
      step(pb + 1, acc[0], acc[1], acc[2]);
      step(pb + 2, acc[1], acc[2], acc[0]);
    }
  }
  __syncthreads();

  if (a.stats) {
    // one row of partial statistics per workgroup: the 32 voxel lanes of each half by shuffles, then the 4 waves through
    // LDS, fixed order (deterministic)
    float* red = reinterpret_cast<float*>(patch);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      if constexpr (F8) { s1[i] *= deq; s2[i] *= deq * deq; }         // the sums are already those of (z - bias)
#pragma unroll
      for (int o = 1; o < 32; o <<= 1) {
        s1[i] += __shfl_xor(s1[i], o, 64);
        s2[i] += __shfl_xor(s2[i], o, 64);
      }
    }
    if (r == 0) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        red[(wave * 2 + 0) * 32 + 16 * h + i] = s1[i];
        red[(wave * 2 + 1) * 32 + 16 * h + i] = s2[i];
      }
    }
    __syncthreads();
    if (wave == 0 && h == 0) {
      float t1 = 0.f, t2 = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w) { t1 += red[(w * 2 + 0) * 32 + r]; t2 += red[(w * 2 + 1) * 32 + r]; }
      float* p = a.stats + ((long long)tile * 2) * a.coutp;
      p[co_base + r] = t1;
      p[a.coutp + co_base + r] = t2;
    }
  }
}
