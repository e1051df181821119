// conv_march2_kernel<ROWS>: dense 2x2x2 stride-1 convolution of wide tensors, bf16 operands -- the PatchGAN's strided layers
// as they run here: Conv3d(k = 4, s = 2, p = 1) of `a` (src/model.py:72-82, DownSampleConv at :42-65) is the dense k = 2
// convolution of the space-to-depth tensor S(a) (8 parity blocks per cell), and its data gradient is the
// same convolution with padding 1 on the flipped / transposed packing.  conv_halo_kernel<bf16_t, 2, ...> staged the 3 x 5 x 33
// halo of one 16-channel chunk per pass and fetched its weight fragments per wave from L2: 0.14 of the MFMA peak, 4.5x the
// HBM time of d1's operands.  This is the marching structure of conv_marchg.h cut to two taps per axis:
//   * a workgroup (4 waves, one per SIMD, one workgroup per CU) owns a (4 ROWS) x 32 footprint of output cells and marches
//     along d; an input plane enters LDS one 32-channel group at a time by LDS-DMA: unit (p, g) = (4 ROWS + 1) x 33 halo cells
//     x 64 B, the swizzled cell-major image of conv_march.h (every ds_read_b128 conflict-free), two unit buffers;
//   * a unit is TWO blocks: kd = 1 on the output plane the unit's input plane completes (p - 1 + pad), kd = 0 on the plane
//     it starts (p + pad); two accumulator sets (2 x ROWS x 16 registers) swap roles from plane to plane;
//   * the weights of block (g, kd) -- 4 (kh, kw) taps x 2 chunks = 8 fragment blocks, 8 KB -- stream through a ring of four
//     slots two blocks ahead; per fragment group (chunk, kw): 2 weight + ROWS + 1 activation fragments, 2 ROWS MFMAs;
//   * a unit is only 16 ROWS MFMAs per wave: shorter than a loaded memory latency, and in-order counted waits cannot keep a
//     copy in flight for longer than the next two blocks -- so the latency is hidden by OCCUPANCY instead: 8-row footprints
//     (ROWS = 2), two workgroups per CU.  Counted waits as in conv_marchg.h (the copies are inline assembly the compiler does not
//     track): all plane copies of the next unit are issued in the unit's first block, so that "everything but this block's
//     two weight copies" at the end of the second block covers them;
//   * ACCUMULATOR INIT FROM MEMORY (`addend`): a set that starts an output plane is loaded from an f32 tensor of the
//     output's geometry instead of being zeroed -- z = conv(x) + addend + bias, with the fused statistics taken of
//     conv(x) + addend.  The PatchGAN's first block is linear in its input cat([x, y]): the x-part of its convolution is
//     computed ONCE per training step (x is the same in the generator phase and in both calls of the discriminator
//     phase, src/model.py:172,184-186, and the discriminator's weights do not change in between) and enters the three
//     y-part launches this way; `y_f32` writes that x-part as f32 (no rounding between the two halves of the sum).
//   * D2S = true (depth-to-space): the TRANSPOSE of such a strided convolution -- the up-branch of MONAI's UpCat as one
//     ConvTranspose3d(k4, s2, p1) of the low-resolution tensor (upcat.hip).  Output class b = (bd, bh, bw) (voxel 2 j + b
//     of the plain output tensor) is a dense k2 convolution of the cells j - (1 - b) + e, e in {0, 1}: blockIdx.y = (class,
//     32-channel tile), the paddings are 1 - b per axis, weights column (class, co); output AND addend (bf16 or f32: the
//     skip part of the concatenated convolution, computed by another launch) use the plain tensor's addressing; the 26
//     border classes of the volume get their bias correction (`delta`) added to the accumulators before the statistics.
#pragma once
#include <type_traits>
#include "conv_marchg.h"

template <int ROWS> struct March2Cfg {
  static constexpr int FH = 4 * ROWS, FW = 32, HR = FH + 1, HC = FW + 1, VOX = HR * HC;
  static constexpr int BLOCKS = ((VOX * 4 + 63) / 64 + 3) / 4 * 4;   // 1-KB DMA instructions per plane unit, a multiple of 4
  static constexpr int NI = BLOCKS / 4;
  static constexpr int PLANE = BLOCKS * 1024;
  static constexpr int WBLK = 8, NWI = WBLK / 4, WUNIT = WBLK * 1024; // weight slot: 2 chunks x 4 taps
  static constexpr int WSLOTS = 4;
  // plane units: the one being read + one in flight.  ROWS = 2: 79 KB of LDS and at most 256 registers -- TWO workgroups per CU,
  // whose copy latencies and MFMA phases overlap each other (measured: d1 59 -> 48 us against 16-row footprints at one
  // workgroup per CU; a third unit buffer with copies two units ahead changed nothing there)
  static constexpr int ASLOTS = 2;
  static constexpr int MISC = 2048;                                   // statistics scratch [256 floats], bias [32 floats]
  static constexpr int LDS = ASLOTS * PLANE + WSLOTS * WUNIT + MISC + NI * 1024;
  static_assert(ROWS != 2 || 2 * LDS <= 160 * 1024, "ROWS = 2: two workgroups per CU");
};

struct March2Args { int seg_len, nseg, tiles_h, tiles_w; const void* addend; int ld_add; int y_f32; int add_n; int add_bf16; const float* delta; };

template <int ROWS, bool D2S>
__global__ __launch_bounds__(256, ROWS == 2 ? 2 : 1) void conv_march2_kernel(const ConvArgs a, const March2Args m) {
  using Cfg = March2Cfg<ROWS>;
  constexpr int HC = Cfg::HC, NI = Cfg::NI, NWI = Cfg::NWI, VB = 64, FH = Cfg::FH, HY = ROWS + 1, NG = 4;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const wl = smem + Cfg::ASLOTS * Cfg::PLANE;
  char* const patch = wl + Cfg::WSLOTS * Cfg::WUNIT;

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  const int co_base = blockIdx.y * 32;                                  // weight columns of this workgroup
  // D2S: blockIdx.y = class * (Co / 32) + channel tile; Co = a.coutp / 8 channels per class
  const int cpc = D2S ? a.coutp >> 3 : a.coutp, ctiles = cpc >> 5;
  const int blk = D2S ? (int)blockIdx.y / ctiles : 0, cot = D2S ? (int)blockIdx.y - blk * ctiles : (int)blockIdx.y;
  const int bd = blk >> 2, bh = (blk >> 1) & 1, bw = blk & 1;
  const int pd = D2S ? 1 - bd : a.pd, ph = D2S ? 1 - bh : a.ph, pw = D2S ? 1 - bw : a.pw;
  const int och = cot * 32;                                             // first stored / bias / statistics channel of this workgroup
  int tile;
  {
    const int nwg = gridDim.x, q = nwg >> 3, rem = nwg & 7, xcd = blockIdx.x & 7, k = blockIdx.x >> 3;
    tile = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + k;
  }
  const int per_seg = m.tiles_h * m.tiles_w, per_sample = per_seg * m.nseg;
  const int tn = tile / per_sample;
  int t = tile - tn * per_sample;
  const int seg = t / per_seg;
  t -= seg * per_seg;
  const int th_i = t / m.tiles_w, tw_i = t - th_i * m.tiles_w;
  const int d0 = seg * m.seg_len, d1 = min(a.do_, d0 + m.seg_len);     // output planes [d0, d1)
  const int h0 = th_i * FH, w0 = tw_i * Cfg::FW;
  const int ng = (a.c0 + a.c1) >> 5, ng0 = a.c0 >> 5;

  // ---- DMA source table (conv_marchg.h): entry = (cell index within its plane) * 4 + channel piece, or -1 (zeros)
  int* const vtab = reinterpret_cast<int*>(patch + Cfg::MISC) + tid;
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int id = i * 4 + wave, v = id * 16 + (lane >> 2);
    const int hy = v / HC, hx = v - hy * HC;
    const int q = (lane & 3) ^ ((hx >> 2) & 3);
    const int gh = h0 - ph + hy, gw = w0 - pw + hx;
    const bool ok = v < Cfg::VOX && gh >= 0 && gh < a.hi && gw >= 0 && gw < a.wi;
    vtab[i * 256] = ok ? (((gh * a.wi + gw) << 2) | q) : -1;
  }
  const long long nvox = (long long)a.n * a.di * a.hi * a.wi;
  const dma_rsrc_t rs0 = dma_rsrc(a.x0, ((nvox - 1) * a.ld0 + a.c0) * 2);
  const dma_rsrc_t rs1 = dma_rsrc(a.c1 ? a.x1 : a.x0, a.c1 ? ((nvox - 1) * a.ld1 + a.c1) * 2 : 0);
  const dma_rsrc_t rsw = dma_rsrc(a.wp, (long long)(ng * 2) * 8 * a.coutp * 32);

  struct ActSrc { dma_rsrc_t rs; int ldb, chb, soff; bool pin; };
  auto act_src = [&](int p, int g) __attribute__((always_inline)) {
    const bool first = g < ng0;
    ActSrc q;
    q.pin = p >= 0 && p < a.di && g < ng;
    q.ldb = (first ? a.ld0 : a.ld1) * 2;
    q.chb = (first ? g : g - ng0) * 64;
    q.soff = q.pin ? (tn * a.di + p) * (a.hi * a.wi * q.ldb) : 0;
    q.rs = first ? rs0 : rs1;
    return q;
  };
  auto act_ticket = [&](const ActSrc& q, int slot, int i, int e) __attribute__((always_inline)) {
    const int voff = (q.pin && e >= 0) ? (int)__umul24((unsigned)e >> 2, (unsigned)q.ldb) + (e & 3) * 16 + q.chb : (int)0x80000000;
    dma_lds_b128(q.rs, smem + slot * Cfg::PLANE + wave * 1024 + i * 4096, voff, q.soff);
  };
  // weights of block (g, kd): fragment block c * 4 + kh * 2 + kw = [lane half][row] x 16 B; row rho holds output channel pi(rho)
  const int wrow = 16 * ((r >> 2) & 1) + 4 * (r >> 3) + (r & 3);
  const int wsrc = ((co_base + wrow) * 2 + h) * 16;
  auto w_ticket = [&](bool runs, int g, int kd, int slot, int i) __attribute__((always_inline)) {
    const int j = i * 4 + wave, c = j >> 2, t4 = j & 3;                // wave-uniform
    const int soff = runs ? (((2 * g + c) * 8 + kd * 4 + t4) * a.coutp) * 32 : 0;
    dma_lds_b128(rsw, wl + slot * Cfg::WUNIT + wave * 1024 + i * 4096, runs ? wsrc : (int)0x80000000, soff);
  };
  // does block kd of input plane p run (its output plane lies inside the segment)?  kd = 1 -> plane p - 1 + pd, kd = 0 -> p + pd
  auto runs = [&](int p, int kd) __attribute__((always_inline)) {
    const int q = p + pd - kd;
    return q >= d0 && q < d1 && p <= d1 - pd;
  };

  int aoff[2][2];
#pragma unroll
  for (int kw = 0; kw < 2; ++kw) {
    const int col = kw + r, v = ROWS * wave * HC + col, s = (col >> 2) & 3;
#pragma unroll
    for (int c = 0; c < 2; ++c) aoff[kw][c] = (4 * v + ((2 * c + h) ^ s)) * 16;
  }
  const int wlane = lane * 16;

  const int cch = och + 16 * h;                                         // this lane: output channels cch .. cch + 15 of voxel / cell w0 + r
  float s1[16], s2[16];
  float* const blds = reinterpret_cast<float*>(patch) + 256;
  if (tid < 32) blds[tid] = (a.bias && och + tid < a.nbias) ? a.bias[och + tid] : 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) { s1[i] = 0.f; s2[i] = 0.f; }
  const bool vox_ok = w0 + r < a.wo;
  const int esz = m.y_f32 ? 4 : 2, asz = m.add_bf16 ? 2 : 4;
  const bool st0 = vox_ok && cch + 8 <= a.cstore, st1 = vox_ok && cch + 16 <= a.cstore;
  const auto rsy = __builtin_amdgcn_make_buffer_rsrc((void*)a.y, 0, (int)((((long long)a.n * a.dy * a.hy * a.wy - 1) * a.ldy + a.cstore) * esz), 0x00020000);
  // output voxel of cell (q, row, col): the cell itself, or (D2S) voxel 2 cell + b of the plain tensor
  constexpr int OS = D2S ? 2 : 1;
  const int yrow = ((OS * (ROWS * wave) + bh) * a.wy + OS * (w0 + r) + bw) * a.ldy * esz + cch * esz;
  // the addend: [n][dy][hy][wy][ld_add] (the output's geometry), f32 or bf16; this lane's 16 channels
  const int nadd = m.add_n > 0 ? m.add_n : a.n, tna = tn % nadd;       // (one addend under several samples: Discriminator.forward_pair)
  const auto rsa = __builtin_amdgcn_make_buffer_rsrc((void*)m.addend, 0,
                                                     m.addend ? (int)((((long long)nadd * a.dy * a.hy * a.wy - 1) * m.ld_add + cpc) * asz) : 0, 0x00020000);
  const int arow = ((OS * (ROWS * wave) + bh) * a.wy + OS * (w0 + r) + bw) * m.ld_add * asz + cch * asz;
  typedef float f32x4v __attribute__((ext_vector_type(4)));
  typedef unsigned int u32x4v __attribute__((ext_vector_type(4)));
  // (re)initialise row `row` of a set for the output plane q it starts next: the addend's values, or zeros
  auto init_row = [&](f32x16 (&s)[ROWS], int q, const int row) __attribute__((always_inline)) {
    const bool ok = m.addend != nullptr && q >= d0 && q < d1 && vox_ok && h0 + ROWS * wave + row < a.ho;
    const int off = ok ? ((tna * a.dy + OS * q + bd) * a.hy + OS * h0) * a.wy * m.ld_add * asz + arow + OS * row * a.wy * m.ld_add * asz
                       : (int)0x80000000;
    if (m.addend != nullptr) {
      if (m.add_bf16) {
#pragma unroll
        for (int i8 = 0; i8 < 2; ++i8) {
          const u32x4v v = __builtin_bit_cast(u32x4v, __builtin_amdgcn_raw_buffer_load_b128(rsa, off + i8 * 16, 0, 0));
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            s[row][8 * i8 + 2 * j] = __uint_as_float(v[j] << 16);
            s[row][8 * i8 + 2 * j + 1] = __uint_as_float(v[j] & 0xffff0000u);
          }
        }
      } else {
#pragma unroll
        for (int i4 = 0; i4 < 4; ++i4) {
          const f32x4v v = __builtin_bit_cast(f32x4v, __builtin_amdgcn_raw_buffer_load_b128(rsa, off + i4 * 16, 0, 0));
          s[row][4 * i4] = v.x; s[row][4 * i4 + 1] = v.y; s[row][4 * i4 + 2] = v.z; s[row][4 * i4 + 3] = v.w;
        }
      }
    } else {
#pragma unroll
      for (int i = 0; i < 16; ++i) s[row][i] = 0.f;
    }
  };

  // ---- epilogue of row `row` of the finished output plane q (q < 0: nothing finished); afterwards the row is initialised
  //      for the plane the set starts next (q_next)
  auto epilogue_row = [&](f32x16 (&s)[ROWS], int q, int q_next, const int row) __attribute__((always_inline)) {
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    const bool row_ok = q >= 0 && h0 + ROWS * wave + row < a.ho;        // wave-uniform
    const bool stat = vox_ok && row_ok;
    const float4* bp = reinterpret_cast<const float4*>(blds + 16 * h);
    const int ybase = ((tn * a.dy + OS * q + bd) * a.hy + OS * h0) * a.wy * a.ldy * esz;
    const int off = ybase + yrow + OS * row * a.wy * a.ldy * esz;
    if constexpr (D2S) {
      // voxels on the volume's border: the transposed convolution's bias reaches fewer taps there (upcat.hip): class-wise
      // correction of the accumulators, ahead of the statistics
      if (m.delta != nullptr && stat) {
        const int vd = 2 * q + bd, vh = 2 * (h0 + ROWS * wave + row) + bh, vw = 2 * (w0 + r) + bw;
        const int cls = (vd == 0 ? 0 : (vd == a.dy - 1 ? 2 : 1)) * 9 + (vh == 0 ? 0 : (vh == a.hy - 1 ? 2 : 1)) * 3 + (vw == 0 ? 0 : (vw == a.wy - 1 ? 2 : 1));
        if (cls != 13) {
          const float4* dp = reinterpret_cast<const float4*>(m.delta + cls * cpc + cch);
#pragma unroll
          for (int i4 = 0; i4 < 4; ++i4) {
            const float4 dv = dp[i4];
            s[row][4 * i4] += dv.x; s[row][4 * i4 + 1] += dv.y; s[row][4 * i4 + 2] += dv.z; s[row][4 * i4 + 3] += dv.w;
          }
        }
      }
    }
    if (m.y_f32) {
#pragma unroll
      for (int i4 = 0; i4 < 4; ++i4) {
        const float4 bq = bp[i4];
        const float bb[4] = {bq.x, bq.y, bq.z, bq.w};
        u32x4 w;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float v = s[row][4 * i4 + j];
          if (stat) { s1[4 * i4 + j] += v; s2[4 * i4 + j] += v * v; }
          w[j] = __float_as_uint(v + bb[j]);
        }
        const bool ok = row_ok && vox_ok && cch + 4 * i4 + 4 <= a.cstore;
        __builtin_amdgcn_raw_buffer_store_b128(w, rsy, ok ? off + i4 * 16 : (int)0x80000000, 0, 0);
      }
    } else {
      uint32_t w[8];
#pragma unroll
      for (int i4 = 0; i4 < 4; ++i4) {
        const float4 bq = bp[i4];
        const float bb[4] = {bq.x, bq.y, bq.z, bq.w};
#pragma unroll
        for (int j = 0; j < 4; j += 2) {
          const int i = 4 * i4 + j;
          const float v0 = s[row][i], v1 = s[row][i + 1];
          if (stat) { s1[i] += v0; s2[i] += v0 * v0; s1[i + 1] += v1; s2[i + 1] += v1 * v1; }
          w[i >> 1] = (uint32_t)f32_to_bf16_bits(v0 + bb[j]) | ((uint32_t)f32_to_bf16_bits(v1 + bb[j + 1]) << 16);
        }
      }
      u32x4 lo = {w[0], w[1], w[2], w[3]}, hi = {w[4], w[5], w[6], w[7]};
      __builtin_amdgcn_raw_buffer_store_b128(lo, rsy, (st0 && row_ok) ? off : (int)0x80000000, 0, 0);
      __builtin_amdgcn_raw_buffer_store_b128(hi, rsy, (st1 && row_ok) ? off + 16 : (int)0x80000000, 0, 0);
    }
    init_row(s, q_next, row);
  };

  // ---- one kd block = 4 fragment groups (chunk, kw): 2 weight + ROWS + 1 activation fragments, 2 ROWS MFMAs
  constexpr int NRD = 2 + HY, NMM = 2 * ROWS;
  struct Group { uint4 b[2], x[HY]; };
  auto load_group = [&](Group& g, const char* apl, const char* wpl, const int gi) __attribute__((always_inline)) {
    const int c = gi >> 1, kw = gi & 1;
#pragma unroll
    for (int kh = 0; kh < 2; ++kh) g.b[kh] = *reinterpret_cast<const uint4*>(wpl + wlane + (c * 4 + kh * 2 + kw) * 1024);
#pragma unroll
    for (int hy = 0; hy < HY; ++hy) g.x[hy] = *reinterpret_cast<const uint4*>(apl + aoff[kw][c] + hy * (HC * VB));
  };
  auto mma_group = [&](const Group& g, f32x16 (&s)[ROWS]) __attribute__((always_inline)) {
#pragma unroll
    for (int hy = 0; hy < HY; ++hy)
#pragma unroll
      for (int kh = 0; kh < 2; ++kh) {
        const int row = hy - kh;
        if (row >= 0 && row < ROWS)
          s[row] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, g.b[kh]), __builtin_bit_cast(bf16x8, g.x[hy]),
                                                           s[row], 0, 0, 0);                // rows = output channels, columns = cells
      }
  };
  auto block = [&](const bool run, const char* apl, const char* wpl, f32x16 (&s)[ROWS], auto within, auto between) __attribute__((always_inline)) {
    if (run) {
      Group g[2];
      load_group(g[0], apl, wpl, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, NRD, 0);
#pragma unroll
      for (int gi = 0; gi < NG; ++gi) {
        within(gi);
        if (gi + 1 < NG) load_group(g[(gi + 1) & 1], apl, wpl, gi + 1);
        mma_group(g[gi & 1], s);
#pragma unroll
        for (int k = 0; k < NMM; ++k) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          if (gi + 1 < NG) {
            const int nr = ((k + 1) * NRD) / NMM - (k * NRD) / NMM;
            if (nr == 1) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            else if (nr == 2) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
        between(gi);
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
#pragma unroll
      for (int gi = 0; gi < NG; ++gi) { within(gi); between(gi); }
    }
  };
  // copy tickets of a block: NWI weight copies, then na plane copies (the order matters for the counted waits); ticket t
  // goes behind fragment group t * NG / nt
  auto tickets = [&](const int gi, const int na, auto wt, auto at) __attribute__((always_inline)) {
    const int nt = NWI + na;
#pragma unroll
    for (int t = 0; t < NWI + NI; ++t) {
      if (t >= nt || (t * NG) / nt != gi) continue;
      if (t < NWI) wt(t);
      else at(t - NWI);
    }
  };
  static_assert(NWI == 2, "ticket order assumes two weight copies per wave and block");

  int u = 0, us = 0;                                                   // units done (blocks done 2 u); act slot of the current unit
  auto nothing = [&](const int) __attribute__((always_inline)) {};
  // unit (p, g): block kd = 1 on `done` (output plane p - 1 + pd), block kd = 0 on `fresh` (plane p + pd).  LAST (compile time):
  // the plane's last group -- `done` is complete after the kd = 1 block; its rows are converted / stored under the kd = 0 block.
  auto unit = [&](auto last_tag, int p, int g, f32x16 (&done)[ROWS], f32x16 (&fresh)[ROWS]) __attribute__((always_inline)) {
    constexpr bool LAST = decltype(last_tag)::value;
    const char* apl = smem + us * Cfg::PLANE;
    const int gn = LAST ? 0 : g + 1, pn = LAST ? p + 1 : p;             // the next unit
    const ActSrc nx = act_src(pn, pn <= d1 - pd ? gn : ng);           // (past the segment's last unit: zero-fills, same count)
    const int nslot = us ^ 1;
    int e[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) e[i] = vtab[i * 256];
    const bool do1 = runs(p, 1), do0 = runs(p, 0);
    const bool r1n = runs(pn, 1), r0n = runs(pn, 0);
    const int b = 2 * u;
    // kd = 1 (weight slot b & 3); requests: the next unit's kd = 1 weights -> slot (b + 2) & 3, the whole next plane unit
    block(do1, apl, wl + (b & 3) * Cfg::WUNIT, done, nothing, [&](const int gi) __attribute__((always_inline)) {
      tickets(gi, NI, [&](const int i) __attribute__((always_inline)) { w_ticket(r1n, gn, 1, (b + 2) & 3, i); },
              [&](const int i) __attribute__((always_inline)) { act_ticket(nx, nslot, i, e[i]); });
    });
    dma_wait_but<NWI + NI>();                  // in flight: this block's copies; landed: everything older (this unit's kd = 0 weights)
    mg_barrier();
    // kd = 0 (slot (b + 1) & 3); requests: the next unit's kd = 0 weights -> slot (b + 3) & 3
    const int qd = p - 1 + pd;
    block(do0, apl, wl + ((b + 1) & 3) * Cfg::WUNIT, fresh,
          [&](const int gi) __attribute__((always_inline)) { if constexpr (LAST) { if (gi < ROWS) epilogue_row(done, do1 ? qd : -1, qd + 2, gi); } },
          [&](const int gi) __attribute__((always_inline)) {
      tickets(gi, 0, [&](const int i) __attribute__((always_inline)) { w_ticket(r0n, gn, 0, (b + 3) & 3, i); }, nothing);
    });
    dma_wait_but<NWI>();                       // ... which stay in flight; everything older (the whole next plane unit) has landed
    mg_barrier();
    ++u;
    us ^= 1;
  };
  auto plane = [&](int p, f32x16 (&done)[ROWS], f32x16 (&fresh)[ROWS]) __attribute__((always_inline)) {
    for (int g = 0; g + 1 < ng; ++g) unit(std::false_type{}, p, g, done, fresh);
    unit(std::true_type{}, p, ng - 1, done, fresh);
  };

  // prologue: the first plane unit (input plane d0 - pd, group 0), the weights of its two blocks
  const int p0 = d0 - pd;
  {
    const ActSrc q0 = act_src(p0, 0);
    int e[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) e[i] = vtab[i * 256];                 // (written by this thread itself: no barrier needed)
#pragma unroll
    for (int i = 0; i < NI; ++i) act_ticket(q0, 0, i, e[i]);
#pragma unroll
    for (int i = 0; i < NWI; ++i) { w_ticket(runs(p0, 1), 0, 1, 0, i); w_ticket(runs(p0, 0), 0, 0, 1, i); }
  }
  {
    f32x16 acc[2][ROWS];
    // acc[1] starts output plane d0 (the first unit's `fresh` set); acc[0] is the `done` set of a plane that does not exist
    // and starts plane d0 + 1 after its (empty) epilogue
#pragma unroll
    for (int row = 0; row < ROWS; ++row) {
      init_row(acc[1], d0, row);
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[0][row][i] = 0.f;
    }
    dma_wait_all();
    __syncthreads();
    const int np = d1 - d0 + 1;                                         // input planes p0 .. p0 + np - 1
    int pb = p0;
    for (int t2 = 0; t2 < np / 2; ++t2) {
      plane(pb, acc[0], acc[1]);
      plane(pb + 1, acc[1], acc[0]);
      pb += 2;
    }
    if (np & 1) plane(pb, acc[0], acc[1]);
  }
  dma_wait_all();
  __syncthreads();

  if (a.stats) {
    float* red = reinterpret_cast<float*>(patch);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
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
      // one row per workgroup: [tile][2][coutp]; D2S: [tile * 8 + class][2][channels per class]
      float* p = a.stats + ((long long)(D2S ? tile * 8 + blk : tile) * 2) * cpc;
      p[och + r] = t1;
      p[cpc + och + r] = t2;
    }
  }
}
