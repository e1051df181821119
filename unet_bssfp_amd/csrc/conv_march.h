// conv_march_kernel: bf16 3x3x3 stride-1 convolution of the full-resolution layers with <= 32 input channels
// (U-Net conv_0.*, upcat_1.conv_1 and their data gradients: 32 -> 32 / 96 output channels at 128^3 / 160^3).
//
// Why another structure (profiles/r01_pmc_conv_ru.txt): conv_ru_kernel stages the 6x6x34 halo of one 16-channel chunk
// per pass -- 2.39x the tile's own voxels, half of every 128-B line per pass -- and the L2 keeps neither the halo
// overlap nor the line until the second chunk pass: 4.0x the input tensor crossed the fabric, so the kernel ran at
// the memory system's pace (691 MB in 155 us), not the MFMA's.  Here a workgroup owns a 16 (h) x 32 (w) footprint and
// MARCHES along d:
//   * each input plane of the footprint (18 x 34 voxels, ALL 32 channels: whole 64-B voxel rows) is brought into LDS
//     exactly once per workgroup with `buffer_load_dwordx4 ... lds` (1 KB = 16 voxels x 64 B per instruction), while
//     the previous plane is being consumed (two plane buffers, one barrier per plane): halo re-reads 1.20x in (h, w)
//     and (L + 2) / L along d for a segment of L planes -- 1.35x instead of 4.0x;
//   * input-stationary: an input plane feeds the three output planes d-1, d, d+1 (kd = 2, 1, 0), whose accumulators
//     (3 planes x 4 rows x 32 voxels x 32 channels per wave = 192 registers) stay in registers; a plane is finished,
//     converted and stored when its kd = 2 contribution is in; the register sets rotate by unrolling the march by 3;
//   * the packed weights of the workgroup's 32 output channels (2 chunks x 27 taps x 1 KB = 54 KB) are loaded into
//     LDS ONCE; a weight fragment read from LDS feeds 4 rows, an activation fragment 3 taps (kh): 0.75 ds_read_b128
//     per MFMA, no per-wave weight traffic through L1 at all;
//   * 4 waves = one per SIMD (launch bound 1: the 512-entry register file is this wave's), 140 KB of LDS: one
//     workgroup per CU, and the segment length is chosen so that the grid is a whole number of 256-workgroup rounds;
//   * the per-channel statistics of the whole segment are accumulated in registers and written once.
// LDS plane image: voxel-major, 64 B per voxel (4 octets of 8 channels), octet o of voxel v = (row, col) of the 18 x 34
// halo plane in 16-B slot 4 v + (o ^ ((col >> 2) & 3)): the DMA writes lane-linear, so the swizzle sits on the per-lane
// SOURCE address, and a fragment read (32 consecutive voxels of a row, one octet) touches 16 distinct slots mod 16 in
// every ds_read_b128 lane group.  The swizzle depends on the column only, so the 6 halo rows a wave reads are one
// address register plus immediate offsets.
#pragma once
#include "conv_common.h"
#ifndef MARCH_SGB_VMEM
#define MARCH_SGB_VMEM 0
#endif
#ifndef MARCH_SGB_VALU
#define MARCH_SGB_VALU 0
#endif
#ifndef MARCH_EPI_AT
#define MARCH_EPI_AT 5
#endif
#ifndef MARCH_SGB
#define MARCH_SGB 1
#endif

constexpr int kMarchFH = 16, kMarchFW = 32, kMarchHR = kMarchFH + 2, kMarchHC = kMarchFW + 2;
constexpr int kMarchVox = kMarchHR * kMarchHC;                       // 612 halo voxels per plane
constexpr int kMarchBlocks = 40;                                     // 1-KB DMA instructions per plane (16 voxels x 64 B; 39 needed,
                                                                     // 40 = 10 per wave: no branch inside a march step)
constexpr int kMarchPlane = kMarchBlocks * 1024;                     // 40 KB
constexpr int kMarchWeights = 2 * 27 * 1024;                         // 54 KB
constexpr int kMarchLds = 2 * kMarchPlane + kMarchWeights + 4096 + 10 * 1024;   // + statistics scratch, bias; DMA offset table

struct MarchArgs { int seg_len, nseg, tiles_h, tiles_w; };

__global__ __launch_bounds__(256, 1) void conv_march_kernel(const ConvArgs a, const MarchArgs m) {
  using T = bf16_t;
  constexpr int HC = kMarchHC, NI = (kMarchBlocks + 3) / 4, NW = (54 + 3) / 4;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef __attribute__((address_space(3))) void* lds_ptr;
  char* const wl = smem + 2 * kMarchPlane;
  char* const patch = wl + kMarchWeights;

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

  // ---- DMA set-up: instruction id = i * 4 + wave covers halo voxels id * 16 .. + 15, lane = (voxel, slot).  The ten
  //      per-lane source offsets live in LDS (one ds_read_b32 each per plane): as registers they were the ones hipcc
  //      spilled to scratch, and a scratch reload at the top of every step drags a vmcnt(0) in front of the DMA issue.
  int* const vtab = reinterpret_cast<int*>(patch + 4096) + tid;
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int id = i * 4 + wave, v = id * 16 + (lane >> 2);
    const int hy = v / HC, hx = v - hy * HC;
    const int o = (lane & 3) ^ ((hx >> 2) & 3);                      // the octet this slot holds
    const int gh = h0 - a.ph + hy, gw = w0 - a.pw + hx;
    const bool ok = id < kMarchBlocks && v < kMarchVox && gh >= 0 && gh < a.hi && gw >= 0 && gw < a.wi;
    vtab[i * 256] = ok ? ((gh * a.wi + gw) * a.ld0 + o * 8) * 2 : (int)0x80000000;
  }
  const long long nvox = (long long)a.n * a.di * a.hi * a.wi;
  const auto rsx = __builtin_amdgcn_make_buffer_rsrc((void*)a.x0, 0, (int)(((nvox - 1) * a.ld0 + a.c0) * 2), 0x00020000);
  const auto rsw = __builtin_amdgcn_make_buffer_rsrc((void*)a.wp, 0, 54 * a.coutp * 32, 0x00020000);
  const int plane_stride = a.hi * a.wi * a.ld0 * 2;                   // bytes per input plane
  auto load_plane = [&](int p) __attribute__((always_inline)) {                                      // p: input plane index (may be outside [0, D): zeros)
    const bool pin = p >= 0 && p < a.di;
    const int soff = pin ? (tn * a.di + p) * plane_stride : 0;
    char* dst = smem + (p & 1) * kMarchPlane + wave * 1024;
#pragma unroll
    for (int i = 0; i < NI; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsx, (lds_ptr)(dst + i * 4096), 16, pin ? vtab[i * 256] : (int)0x80000000, soff, 0, 0);
  };
  // ---- weights of this workgroup's 32 output channels: [chunk][tap] blocks of [lane half][row] x 16 B.  The weights are
  //      the MFMA's A operand (rows = output channels, columns = voxels), so that a lane ends up with 16 channels of ONE
  //      voxel; row rho of the fragment holds output channel pi(rho) = 16 * ((rho >> 2) & 1) + 4 * (rho >> 3) + (rho & 3),
  //      which makes the 16 accumulator registers of lane half h' the CONTIGUOUS channels 16 h' .. 16 h' + 15: the
  //      epilogue is two 16-byte stores per lane, no transposition.  (The permutation costs nothing: it sits on the
  //      per-lane source address of the LDS-DMA.)
  const int wrow = 16 * ((r >> 2) & 1) + 4 * (r >> 3) + (r & 3);
#pragma unroll
  for (int i = 0; i < NW; ++i) {
    const int j = i * 4 + wave;
    if (j < 54)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, (lds_ptr)(wl + j * 1024), 16, ((j * a.coutp + co_base + wrow) * 2 + h) * 16, 0, 0, 0);
  }
  load_plane(d0 - 1);

  // ---- per-lane LDS offsets of the activation fragments: halo row 4 wave + hy (immediate: hy * 34 * 64 B), voxels
  //      kw + r, octet 2 c + h
  int aoff[3][2];
#pragma unroll
  for (int kw = 0; kw < 3; ++kw) {
    const int col = kw + r, v = 4 * wave * HC + col, s = (col >> 2) & 3;
#pragma unroll
    for (int c = 0; c < 2; ++c) aoff[kw][c] = (4 * v + ((2 * c + h) ^ s)) * 16;
  }
  const char* const wlane = wl + lane * 16;

  // this lane: voxel w0 + r of a row, output channels co_base + 16 h .. + 15
  const int cch = co_base + 16 * h;
  // the bias is the accumulators' initial value (one LDS read of 16 floats where an output plane starts); the
  // statistics are taken of z = acc and turned into those of (z - bias) once, at the end
  float s1[16], s2[16];
  float* const blds = reinterpret_cast<float*>(patch) + 512;          // [32] bias of this workgroup's channels
  if (tid < 32) blds[tid] = (a.bias && co_base + tid < a.nbias) ? a.bias[co_base + tid] : 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) { s1[i] = 0.f; s2[i] = 0.f; }
  int nstat = 0;                                                      // voxels this lane has added to its sums
  const bool vox_ok = w0 + r < a.wo;
  const bool st0 = vox_ok && cch + 8 <= a.cstore, st1 = vox_ok && cch + 16 <= a.cstore;

  // ---- epilogue of one finished output plane q: + bias, bf16, two 16-byte stores per lane and row; statistics of
  //      (z - bias) over the valid voxels stay in registers (per lane: 16 channels of its voxel column)
  auto store_plane = [&](f32x16 (&s)[4], int q) __attribute__((always_inline)) {
#pragma unroll
    for (int row = 0; row < 4; ++row) {
      const int gh = h0 + 4 * wave + row;
      if (gh >= a.ho) break;                                          // wave-uniform
      nstat += vox_ok ? 1 : 0;
      T* dst = reinterpret_cast<T*>(a.y) + ((((long long)tn * a.dy + q) * a.hy + gh) * a.wy + w0 + r) * a.ldy + cch;
      uint32_t w[8];
#pragma unroll
      for (int i = 0; i < 16; i += 2) {
        const float v0 = s[row][i], v1 = s[row][i + 1];
#ifndef MARCH_DIAG_NO_STATS
        if (vox_ok) { s1[i] += v0; s2[i] += v0 * v0; s1[i + 1] += v1; s2[i + 1] += v1 * v1; }
#endif
        w[i >> 1] = (uint32_t)f32_to_bf16_bits(v0) | ((uint32_t)f32_to_bf16_bits(v1) << 16);
      }
#ifndef MARCH_DIAG_NO_GSTORE
      if (st0) *reinterpret_cast<uint4*>(dst) = make_uint4(w[0], w[1], w[2], w[3]);
      if (st1) *reinterpret_cast<uint4*>(dst + 8) = make_uint4(w[4], w[5], w[6], w[7]);
#else
      asm volatile("" :: "v"(w[0]), "v"(w[1]), "v"(w[2]), "v"(w[3]), "v"(w[4]), "v"(w[5]), "v"(w[6]), "v"(w[7]), "v"(dst));
#endif
    }
  };

  // ---- one (kd) block of a march step: 2 chunks x 3 kw groups of {3 weight fragments (kh), 6 activation fragments
  //      (halo rows), 12 MFMAs}; the next group's 9 fragment reads are issued ahead of this group's MFMAs (one wave per
  //      SIMD: nothing else hides the LDS latency)
  struct Group { Frag<T> b[3], x[6]; };
  auto load_group = [&](Group& g, const char* pl, const int kd, const int gi) __attribute__((always_inline)) {
    const int c = gi / 3, kw = gi - c * 3;
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) g.b[kh].load(wlane + (c * 27 + kd * 9 + kh * 3 + kw) * 1024);
#pragma unroll
    for (int hy = 0; hy < 6; ++hy) g.x[hy].load(pl + aoff[kw][c] + hy * (HC * 64));
  };
  auto mma_group = [&](const Group& g, f32x16 (&s)[4], const bool zero) __attribute__((always_inline)) {
#pragma unroll
    for (int hy = 0; hy < 6; ++hy)
#pragma unroll
      for (int kh = 0; kh < 3; ++kh) {
        const int row = hy - kh;
        if (row >= 0 && row < 4) {
          if (zero && kh == 0) {                                      // first contribution to this row: C = bias
            f32x16 z;
            const float4* bp = reinterpret_cast<const float4*>(blds + 16 * h);
#pragma unroll
            for (int i4 = 0; i4 < 4; ++i4) {
              const float4 q = bp[i4];
              z[4 * i4] = q.x; z[4 * i4 + 1] = q.y; z[4 * i4 + 2] = q.z; z[4 * i4 + 3] = q.w;
            }
            s[row] = z;
          }
          mma16(g.b[kh], g.x[hy], s[row]);                            // rows = output channels, columns = voxels
        }
      }
  };
  auto block = [&](const char* pl, f32x16 (&s)[4], const int kd, const bool fresh) __attribute__((always_inline)) {
    Group g0, g1;
    load_group(g0, pl, kd, 0);
#pragma unroll
    for (int gi = 0; gi < 6; gi += 2) {
      load_group(g1, pl, kd, gi + 1);
      mma_group(g0, s, fresh && gi == 0);
      if (gi + 2 < 6) load_group(g0, pl, kd, gi + 2);
      mma_group(g1, s, false);
    }
    // hipcc's scheduler sinks every read next to its first use (it minimises live registers), which exposes the LDS
    // latency ~100 times per plane.  Pin the software pipeline: group 0's 9 reads, then each group's 12 MFMAs with the
    // NEXT group's 9 reads interleaved one per MFMA.
    __builtin_amdgcn_sched_group_barrier(0x100, 9, 0);
#pragma unroll
    for (int gi = 0; gi < 5; ++gi) {
#pragma unroll
      for (int k = 0; k < 9; ++k) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
      __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
    }
    __builtin_amdgcn_sched_group_barrier(0x008, 12, 0);
  };
  // ---- the steady-state step (all three output planes inside the segment, footprint inside the volume) as ONE basic
  //      block: the next plane's 10 LDS-DMA instructions, 18 chained fragment groups (kd = 2, 1, 0) and the epilogue of the
  //      plane that kd = 2 completes, with the issue order pinned: per MFMA one fragment read of the NEXT group and up to
  //      two VALU instructions (epilogue arithmetic), one DMA / store per group.  One wave per SIMD: what is not placed
  //      between MFMAs is not overlapped with anything.
  const auto rsy = __builtin_amdgcn_make_buffer_rsrc((void*)a.y, 0, (int)((((long long)a.n * a.dy * a.hy * a.wy - 1) * a.ldy + a.cstore) * 2), 0x00020000);
  const int yrow = ((4 * wave) * a.wy + w0 + r) * a.ldy * 2 + cch * 2;        // byte offset of this lane's voxel in row 0 of a plane's footprint
  auto full_step = [&](int p, f32x16 (&s_m1)[4], f32x16 (&s_0)[4], f32x16 (&s_p1)[4]) __attribute__((always_inline)) {
    load_plane(p + 1);
    const char* pl = smem + (p & 1) * kMarchPlane;
    Group g[2];
    load_group(g[0], pl, 2, 0);
#pragma unroll
    for (int gi = 0; gi < 18; ++gi) {
      if (gi + 1 < 18) load_group(g[(gi + 1) & 1], pl, 2 - (gi + 1) / 6, (gi + 1) % 6);
      if (gi < 6) mma_group(g[gi & 1], s_m1, false);
      else if (gi < 12) mma_group(g[gi & 1], s_0, false);
      else mma_group(g[gi & 1], s_p1, gi == 12);
#ifdef MARCH_DIAG_NO_STORE
      if (gi == 5) asm volatile("" :: "v"(s_m1[0][0]), "v"(s_m1[1][5]), "v"(s_m1[2][9]), "v"(s_m1[3][15]));
#else
      if (gi == MARCH_EPI_AT) {                                       // output plane p - 1 is complete
        nstat += 4;
        const int ybase = ((tn * a.dy + (p - 1)) * a.hy + h0) * a.wy * a.ldy * 2;
#pragma unroll
        for (int row = 0; row < 4; ++row) {
          typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
          uint32_t w[8];
#pragma unroll
          for (int i = 0; i < 16; i += 2) {
            const float v0 = s_m1[row][i], v1 = s_m1[row][i + 1];
#ifndef MARCH_DIAG_NO_STATS
            s1[i] += v0; s2[i] += v0 * v0; s1[i + 1] += v1; s2[i + 1] += v1 * v1;
#endif
            w[i >> 1] = (uint32_t)f32_to_bf16_bits(v0) | ((uint32_t)f32_to_bf16_bits(v1) << 16);
          }
          const int off = yrow + row * a.wy * a.ldy * 2;
          u32x4 lo = {w[0], w[1], w[2], w[3]}, hi = {w[4], w[5], w[6], w[7]};
#ifndef MARCH_DIAG_NO_GSTORE
          __builtin_amdgcn_raw_buffer_store_b128(lo, rsy, st0 ? off : (int)0x80000000, ybase, 0);
          __builtin_amdgcn_raw_buffer_store_b128(hi, rsy, st1 ? off + 16 : (int)0x80000000, ybase, 0);
#else
          asm volatile("" :: "v"(lo), "v"(hi), "v"(off));
#endif
        }
      }
#endif
    }
#if MARCH_SGB
    __builtin_amdgcn_sched_group_barrier(0x100, 9, 0);
#endif
#pragma unroll
    for (int gi = 0; gi < (MARCH_SGB ? 18 : 0); ++gi) {
#pragma unroll
      for (int k = 0; k < 12; ++k) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        if (gi < 17 && k < 9) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
#if MARCH_SGB_VMEM == 1
        if (k == 10) __builtin_amdgcn_sched_group_barrier(0x030, 1, 0);       // one DMA / store per group
#elif MARCH_SGB_VMEM == 2
        if (gi == 0 && k < 10) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);   // the next plane's DMA: first thing, one per MFMA
        if (gi == 6 && k < 8) __builtin_amdgcn_sched_group_barrier(0x040, 1, 0);    // the finished plane's stores: right behind kd = 2
#elif MARCH_SGB_VMEM == 3
        if (gi == 6 && k < 8) __builtin_amdgcn_sched_group_barrier(0x040, 1, 0);
#endif
#if MARCH_SGB_VALU
        __builtin_amdgcn_sched_group_barrier(0x002, MARCH_SGB_VALU, 0);
#endif
      }
    }
    __syncthreads();
  };
  auto step = [&](int p, f32x16 (&s_m1)[4], f32x16 (&s_0)[4], f32x16 (&s_p1)[4]) __attribute__((always_inline)) {
    if (p < d0 - 1 || p > d1) return;                                 // wave-uniform: outside this segment's input planes
#ifndef MARCH_DIAG_NO_DMA
    if (p + 1 <= d1) load_plane(p + 1);                               // lands under this plane's MFMAs
#endif
    const char* pl = smem + (p & 1) * kMarchPlane;
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
    __syncthreads();              // vmcnt(0): plane p + 1 has landed; every wave is done with plane p's buffer
  };

  __syncthreads();                // weights and the first plane have landed
  // Two complete marches, never mixed (hipcc's register allocator cannot keep three 64-register accumulator sets in place
  // across a control-flow join of a straight-line step and a branchy one: it copied and spilled whole sets).
  //  * regular segment (footprint inside the volume, L = d1 - d0 >= 5 and L = 2 mod 3): output plane d0 + j lives in set
  //    j mod 3, so the two lead-in steps, the (L - 2) / 3 triples of straight-line full steps and the two lead-out steps
  //    all have compile-time register sets;
  //  * anything else (volume border, last short segment): the generic step with its wave-uniform branches.
  const bool interior = h0 + kMarchFH <= a.ho && w0 + kMarchFW <= a.wo && st1;
  const int slen = d1 - d0;
#ifndef MARCH_NO_FAST
  if (interior && slen >= 5 && slen % 3 == 2) {
    f32x16 sa[4], sb[4], sc[4];
    int p = d0 - 1;
    load_plane(p + 1);
    block(smem + (p & 1) * kMarchPlane, sa, 0, true);                 // input plane d0 - 1 -> output d0
    __syncthreads();
    ++p;
    load_plane(p + 1);
    block(smem + (p & 1) * kMarchPlane, sa, 1, false);                // input plane d0 -> outputs d0, d0 + 1
    block(smem + (p & 1) * kMarchPlane, sb, 0, true);
    __syncthreads();
    ++p;
    for (int t = 0; t < (slen - 2) / 3; ++t) {
      full_step(p, sa, sb, sc);
      full_step(p + 1, sb, sc, sa);
      full_step(p + 2, sc, sa, sb);
      p += 3;
    }
    load_plane(p + 1);                                                // p = d1 - 1 (plane d1 may lie outside: zeros)
    block(smem + (p & 1) * kMarchPlane, sa, 2, false);
    store_plane(sa, p - 1);
    block(smem + (p & 1) * kMarchPlane, sb, 1, false);
    __syncthreads();
    ++p;
    block(smem + (p & 1) * kMarchPlane, sb, 2, false);                // p = d1
    store_plane(sb, p - 1);
  } else
#endif
  {
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
    const float cnt = (float)nstat;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const float b = blds[16 * h + i];                               // sums of z -> sums of (z - bias)
      s2[i] = s2[i] - 2.f * b * s1[i] + cnt * b * b;
      s1[i] = s1[i] - cnt * b;
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
