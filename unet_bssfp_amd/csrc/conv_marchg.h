// conv_marchg_kernel<ROWS>: 3x3x3 stride-1 convolution (and data gradient) of the layers with MORE than 32 input channels,
// bf16 operands: upcat_1.conv_0 (96 -> 32 at 128^3: 30 % of the generator's FLOPs, src/model.py:22-28 via MONAI UpCat), the
// 64^3 level (64 -> 64, 128 -> 64, their data gradients) and the 32^3 level.  The marching structure of conv_march.h, with
// the two things that kept those layers off it removed:
//   * INPUT CHANNELS IN GROUPS OF 32.  A workgroup still owns a (4 ROWS) x 32 footprint and marches along d, but an input
//     plane enters LDS one 32-channel group at a time: unit (p, g) = halo plane p of group g = 18 x 34 voxels x 64 B (the same
//     swizzled voxel-major image as conv_march.h, so every fragment read stays conflict-free), two unit buffers, the next
//     unit in flight (LDS-DMA) while the current one feeds the MFMAs.  A group comes from the first or the second source
//     tensor: the U-Net's skip concatenations are never materialised.
//   * WEIGHTS STREAMED.  166 KB (96 -> 32) to 442 KB (256 -> 32 columns) of packed weights do not fit beside the planes, and
//     they do not have to: block (g, kd) -- 72 MFMAs per wave -- needs the 18 (kh, kw, 16-channel chunk) fragments of one kd plane
//     of one group, 18 KB.  Three such slots form a ring filled by LDS-DMA two blocks ahead; the weights are L2-resident
//     (every workgroup streams the same bytes), so this costs L2 -> LDS bandwidth only: (40 KB plane unit + 54 KB weights)
//     per 6 912 MFMA cycles = 14 B / clk / CU.
// Input-stationary as before: the three kd blocks of a unit feed the output planes p-1, p, p+1, whose accumulators
// (3 x ROWS x 16 registers) stay in the AGPRs over ALL groups; plane p-1 is complete after the last group's kd = 2 block and
// is converted, counted into the fused statistics and stored under the following blocks' MFMAs.
//
// One barrier per block (the ring's hand-over); waits are COUNTED: the LDS-DMA instructions are inline assembly the
// compiler does not track (common.h), so at the end of a block the kernel waits for every copy older than the ones that
// block itself issued -- s_waitcnt vmcnt(n), n = this block's DMA instructions -- which leaves the copies for the block
// after next in flight across the barrier.  The epilogue's output stores must NOT be added to n: measured on gfx950, stores
// retire out of order with respect to loads (a store is acknowledged by the L2 while older loads still wait for HBM), so
// "the n youngest operations" may consist of loads only once the stores have gone -- with n = copies + stores the kd = 0
// block read a weight slot whose copy had not landed (plausible-looking wrong sums, run-to-run differences at 128^3 only).
// Loads retire in order among themselves, which is all the count relies on; a pending store only makes a wait longer.
//
// ROWS = 4: 16 x 32 footprints (128^3, 64^3: enough footprints x d-segments to fill 256 CUs); ROWS = 2: 8 x 32 footprints for
// the 32^3 level (twice the workgroups, 1.17 instead of 0.75 LDS reads per MFMA).
#pragma once
#include <type_traits>
#include "conv_march.h"

template <int ROWS> struct MarchGCfg {
  static constexpr int FH = 4 * ROWS, FW = 32, HR = FH + 2, HC = FW + 2, VOX = HR * HC;
  static constexpr int BLOCKS = ((VOX * 4 + 63) / 64 + 3) / 4 * 4;   // 1-KB DMA instructions per plane unit, a multiple of 4:
  static constexpr int NI = BLOCKS / 4;                               // every wave issues the same number (counted waits)
  static constexpr int PLANE = BLOCKS * 1024;
  static constexpr int WBLK = 20, NWI = WBLK / 4, WUNIT = WBLK * 1024; // weight slot: 18 fragment blocks + 2 pad blocks
  static constexpr int MISC = 4096;                                   // statistics scratch, bias
  static constexpr int LDS = 2 * PLANE + 3 * WUNIT + MISC + NI * 1024; // ... + the DMA source table
};

template <int N> __device__ __forceinline__ void dma_wait_but() {     // all but the N youngest vector-memory operations are done
  static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit field");
#ifndef MG_DIAG_NO_WAIT
  asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory");
#endif
}
__device__ __forceinline__ void mg_barrier() {
#ifndef MG_DIAG_NO_BARRIER
  __syncthreads();
#endif
}

template <int ROWS>
__global__ __launch_bounds__(256, 1) void conv_marchg_kernel(const ConvArgs a, const MarchArgs m) {
  using T = bf16_t;
  using Cfg = MarchGCfg<ROWS>;
  constexpr int HC = Cfg::HC, NI = Cfg::NI, NWI = Cfg::NWI, VB = 64, FH = Cfg::FH, HY = ROWS + 2;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const wl = smem + 2 * Cfg::PLANE;
  char* const patch = wl + 3 * Cfg::WUNIT;

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  const int co_base = blockIdx.y * 32;
  int tile;
  {
    const int nwg = gridDim.x, q = nwg >> 3, rem = nwg & 7, xcd = blockIdx.x & 7, k = blockIdx.x >> 3;
    tile = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + k;
  }
  // tile -> (sample, d-segment, footprint), segments slowest: an XCD's contiguous tile range is a d-slab whose footprints
  // share their (h, w) halo columns in that XCD's L2
  const int per_seg = m.tiles_h * m.tiles_w, per_sample = per_seg * m.nseg;
  const int tn = tile / per_sample;
  int t = tile - tn * per_sample;
  const int seg = t / per_seg;
  t -= seg * per_seg;
  const int th_i = t / m.tiles_w, tw_i = t - th_i * m.tiles_w;
  const int d0 = seg * m.seg_len, d1 = min(a.do_, d0 + m.seg_len);     // output planes [d0, d1)
  const int h0 = th_i * FH, w0 = tw_i * Cfg::FW;
  const int ng = (a.c0 + a.c1) >> 5, ng0 = a.c0 >> 5;                  // 32-channel groups: all / in the first source

  // ---- DMA source table (LDS, one ds_read_b32 per copy): instruction id = i * 4 + wave covers 16 halo voxels x 4 pieces;
  //      entry = (voxel index within its plane) * 4 + channel piece, or -1 (padding voxel / beyond the halo: zeros)
  int* const vtab = reinterpret_cast<int*>(patch + Cfg::MISC) + tid;
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int id = i * 4 + wave, v = id * 16 + (lane >> 2);
    const int hy = v / HC, hx = v - hy * HC;
    const int q = (lane & 3) ^ ((hx >> 2) & 3);                        // the channel piece this slot holds (see conv_march.h)
    const int gh = h0 - a.ph + hy, gw = w0 - a.pw + hx;
    const bool ok = v < Cfg::VOX && gh >= 0 && gh < a.hi && gw >= 0 && gw < a.wi;
    vtab[i * 256] = ok ? (((gh * a.wi + gw) << 2) | q) : -1;
  }
  const long long nvox = (long long)a.n * a.di * a.hi * a.wi;
  const dma_rsrc_t rs0 = dma_rsrc(a.x0, ((nvox - 1) * a.ld0 + a.c0) * 2);
  const dma_rsrc_t rs1 = dma_rsrc(a.c1 ? a.x1 : a.x0, a.c1 ? ((nvox - 1) * a.ld1 + a.c1) * 2 : 0);
  const dma_rsrc_t rsw = dma_rsrc(a.wp, (long long)(ng * 2) * 27 * a.coutp * 32);
  // plane unit (p, g) -> act buffer `slot`; p outside [0, D) or g >= ng (past the last unit): zeros
  auto load_act = [&](int p, int g, int slot) __attribute__((always_inline)) {
    const bool first = g < ng0;
    const bool pin = p >= 0 && p < a.di && g < ng;
    const int ldb = (first ? a.ld0 : a.ld1) * 2;
    const int chb = (first ? g : g - ng0) * 64;
    const int soff = pin ? (tn * a.di + p) * (a.hi * a.wi * ldb) : 0;
    const dma_rsrc_t rs = first ? rs0 : rs1;
    char* dst = smem + slot * Cfg::PLANE + wave * 1024;
    // (all table entries first: every copy is an asm statement with a memory clobber, a table read between two of them
    //  would be waited for -- one LDS latency per copy -- before the next copy can issue)
    int e[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) e[i] = vtab[i * 256];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int voff = (pin && e[i] >= 0) ? (int)__umul24((unsigned)e[i] >> 2, (unsigned)ldb) + (e[i] & 3) * 16 + chb : (int)0x80000000;
#ifndef MG_DIAG_NO_DMA_ACT
      dma_lds_b128(rs, dst + i * 4096, voff, soff);
#else
      asm volatile("" :: "v"(voff), "s"(soff));
#endif
    }
  };
  // weights of block (g, kd) -> weight slot `slot`: fragment block c * 9 + kh * 3 + kw (c = 16-channel chunk of the group) =
  // [lane half][row] x 16 B; row rho holds output channel pi(rho) (conv_march.h: a lane ends up with 16 contiguous channels)
  const int wrow = 16 * ((r >> 2) & 1) + 4 * (r >> 3) + (r & 3);
  const int wsrc = ((co_base + wrow) * 2 + h) * 16;
  // (p: the input plane of the block; a block whose output plane lies outside the segment does not run and gets zeros)
  auto load_w = [&](int p, int g, int kd, int slot) __attribute__((always_inline)) {
    char* dst = wl + slot * Cfg::WUNIT + wave * 1024;
    const bool runs = p <= d1 && (kd == 2 ? p - 1 >= d0 : (kd == 1 ? (p >= d0 && p < d1) : p + 1 < d1));
#pragma unroll
    for (int i = 0; i < NWI; ++i) {
      const int j = i * 4 + wave, c = j >= 9 ? 1 : 0, t9 = j - 9 * c;  // wave-uniform
      const bool ok = j < 18 && runs;
      const int soff = ok ? (((2 * g + c) * 27 + kd * 9 + t9) * a.coutp) * 32 : 0;
#ifndef MG_DIAG_NO_DMA_W
      dma_lds_b128(rsw, dst + i * 4096, ok ? wsrc : (int)0x80000000, soff);
#else
      asm volatile("" :: "v"(ok ? wsrc : 0), "s"(soff));
#endif
    }
  };

  // The same copies ONE AT A TIME ("tickets"), for the steady state: issued all together at the start of a block, the 15-25
  // copies of a unit cost the wave their whole issue time with the matrix pipe idle (timing-only builds: 379 us with, 232 us
  // without any copy for 96 -> 32 at 128^3); spread between the fragment groups, a copy's issue runs under the MFMA in flight.
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
#ifndef MG_DIAG_NO_DMA_ACT
    dma_lds_b128(q.rs, smem + slot * Cfg::PLANE + wave * 1024 + i * 4096, voff, q.soff);
#else
    asm volatile("" :: "v"(voff), "s"(q.soff));
#endif
  };
  auto w_runs = [&](int p, int kd) __attribute__((always_inline)) {
    return p <= d1 && (kd == 2 ? p - 1 >= d0 : (kd == 1 ? (p >= d0 && p < d1) : p + 1 < d1));
  };
  auto w_ticket = [&](bool runs, int g, int kd, int slot, int i) __attribute__((always_inline)) {
    const int j = i * 4 + wave, c = j >= 9 ? 1 : 0, t9 = j - 9 * c;    // wave-uniform
    const bool ok = j < 18 && runs;
    const int soff = ok ? (((2 * g + c) * 27 + kd * 9 + t9) * a.coutp) * 32 : 0;
#ifndef MG_DIAG_NO_DMA_W
    dma_lds_b128(rsw, wl + slot * Cfg::WUNIT + wave * 1024 + i * 4096, ok ? wsrc : (int)0x80000000, soff);
#else
    asm volatile("" :: "v"(ok ? wsrc : 0), "s"(soff));
#endif
  };

  // ---- per-lane LDS offsets of the activation fragments: [kw][chunk] -> voxels kw + r of halo row ROWS * wave (+ immediate)
  int aoff[3][2];
#pragma unroll
  for (int kw = 0; kw < 3; ++kw) {
    const int col = kw + r, v = ROWS * wave * HC + col, s = (col >> 2) & 3;
#pragma unroll
    for (int c = 0; c < 2; ++c) aoff[kw][c] = (4 * v + ((2 * c + h) ^ s)) * 16;
  }
  const int wlane = lane * 16;

  // this lane: voxel w0 + r of a row, output channels co_base + 16 h .. + 15.  The accumulators hold z - bias; the fused
  // statistics are taken of exactly that, the bias is added where a finished row is converted for the store.
  const int cch = co_base + 16 * h;
  float s1[16], s2[16];
  float* const blds = reinterpret_cast<float*>(patch) + 512;
  if (tid < 32) blds[tid] = (a.bias && co_base + tid < a.nbias) ? a.bias[co_base + tid] : 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) { s1[i] = 0.f; s2[i] = 0.f; }
  const bool vox_ok = w0 + r < a.wo;
  const bool st0 = vox_ok && cch + 8 <= a.cstore, st1 = vox_ok && cch + 16 <= a.cstore;
  const auto rsy = __builtin_amdgcn_make_buffer_rsrc((void*)a.y, 0, (int)((((long long)a.n * a.dy * a.hy * a.wy - 1) * a.ldy + a.cstore) * 2), 0x00020000);
  const int yrow = ((ROWS * wave) * a.wy + w0 + r) * a.ldy * 2 + cch * 2;   // byte offset of this lane's voxel in row 0 of a plane's footprint

  // ---- epilogue of row `row` of the finished output plane q: statistics of (z - bias), + bias, bf16, two 16-byte buffer
  //      stores (rows / voxels / channels that must not be written get an out-of-range offset), and the row's registers are
  //      left zeroed for the set's next role.  q < 0: no finished plane (lead-in) -- nothing stored or counted.
  auto epilogue_row = [&](f32x16 (&s)[ROWS], int q, const int row) __attribute__((always_inline)) {
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    const bool row_ok = q >= 0 && h0 + ROWS * wave + row < a.ho;        // wave-uniform
    const bool stat = vox_ok && row_ok;
    const int ybase = ((tn * a.dy + q) * a.hy + h0) * a.wy * a.ldy * 2;
    uint32_t w[8];
    const float4* bp = reinterpret_cast<const float4*>(blds + 16 * h);
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
    // (plane base in the VECTOR offset, soffset = 0: see conv_march.h on the store-data hazard with a register soffset)
    const int off = ybase + yrow + row * a.wy * a.ldy * 2;
    u32x4 lo = {w[0], w[1], w[2], w[3]}, hi = {w[4], w[5], w[6], w[7]};
#ifndef MG_DIAG_NO_STORE
    __builtin_amdgcn_raw_buffer_store_b128(lo, rsy, (st0 && row_ok) ? off : (int)0x80000000, 0, 0);
    __builtin_amdgcn_raw_buffer_store_b128(hi, rsy, (st1 && row_ok) ? off + 16 : (int)0x80000000, 0, 0);
#else
    asm volatile("" :: "v"(lo), "v"(hi), "v"(off));
#endif
#pragma unroll
    for (int i = 0; i < 16; ++i) s[row][i] = 0.f;
  };
  auto zero_set = [&](f32x16 (&s)[ROWS]) __attribute__((always_inline)) {
#pragma unroll
    for (int row = 0; row < ROWS; ++row)
#pragma unroll
      for (int i = 0; i < 16; ++i) s[row][i] = 0.f;
  };

  // ---- one kd block = 6 fragment groups (chunk, kw): 3 weight + ROWS + 2 activation fragments, 3 * ROWS MFMAs; the next
  //      group's reads are issued ahead of this group's MFMAs, order pinned (one wave per SIMD: nothing else hides LDS latency)
  constexpr int NRD = 3 + HY, NMM = 3 * ROWS;
  struct Group { uint4 b[3], x[HY]; };
  auto load_group = [&](Group& g, const char* apl, const char* wpl, const int gi) __attribute__((always_inline)) {
    const int c = gi / 3, kw = gi - c * 3;
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) g.b[kh] = *reinterpret_cast<const uint4*>(wpl + wlane + (c * 9 + kh * 3 + kw) * 1024);
#pragma unroll
    for (int hy = 0; hy < HY; ++hy) g.x[hy] = *reinterpret_cast<const uint4*>(apl + aoff[kw][c] + hy * (HC * VB));
  };
  auto mma_group = [&](const Group& g, f32x16 (&s)[ROWS]) __attribute__((always_inline)) {
#pragma unroll
    for (int hy = 0; hy < HY; ++hy)
#pragma unroll
      for (int kh = 0; kh < 3; ++kh) {
        const int row = hy - kh;
        if (row >= 0 && row < ROWS)
          s[row] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, g.b[kh]), __builtin_bit_cast(bf16x8, g.x[hy]),
                                                           s[row], 0, 0, 0);                // rows = output channels, columns = voxels
      }
  };
  // between(gi): called after group gi's MFMAs, fenced (nothing is scheduled across): the block's copy tickets;
  // within(gi): called in front of group gi's MFMAs, unfenced: VALU work the scheduler may spread under the MFMAs (epilogue rows)
  auto block = [&](const bool run, const char* apl, const char* wpl, f32x16 (&s)[ROWS], auto within, auto between) __attribute__((always_inline)) {
    if (run) {
      Group g[2];
      load_group(g[0], apl, wpl, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, NRD, 0);
#pragma unroll
      for (int gi = 0; gi < 6; ++gi) {
        within(gi);
        if (gi + 1 < 6) load_group(g[(gi + 1) & 1], apl, wpl, gi + 1);
        mma_group(g[gi & 1], s);
#pragma unroll
        for (int k = 0; k < NMM; ++k) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          if (gi + 1 < 6) {
            const int nr = ((k + 1) * NRD) / NMM - (k * NRD) / NMM;      // the next group's NRD reads spread over this group's MFMAs
            if (nr == 1) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            else if (nr == 2) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
        between(gi);
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {                                   // a block whose output plane lies outside the segment: its copies and epilogue rows only
#pragma unroll
      for (int gi = 0; gi < 6; ++gi) { within(gi); between(gi); }
    }
  };

  // ---- unit (p, g): blocks kd = 2, 1, 0 on output planes p - 1 (s_m1), p (s_0), p + 1 (s_p1).  Weight slot of kd block k
  //      is 2 - kd (fixed: a unit is three blocks and the ring has three slots); at the start of a block the weights of the
  //      block after next are requested, at the start of a unit the next plane unit.
  //      Counted waits: at the end of a block everything issued BEFORE this block must have landed.
  int u = 0;                                                           // units done (act slot = u & 1)
  // LAST (compile time): this is the plane's last group -- its kd = 2 block completes output plane p - 1, whose epilogue runs
  // under the kd = 1 / kd = 0 blocks and leaves the register set zeroed for its next role (the plane after next's kd = 0).
  // The g loop is peeled instead of branching on g: with the epilogue inside a run-time diamond hipcc spilled 224 registers.
  constexpr int NA0 = (NI + 1) / 2, NA1 = NI - NA0;                    // plane-unit copies issued in the kd = 2 / kd = 1 block
  // copy tickets of a block: NW weight copies and NA plane copies alternate (W0 A0 W1 A1 ...), ticket t goes behind fragment
  // group t * 6 / NT -- at most two copies between two groups
  auto tickets = [&](const int gi, const int na, auto wt, auto at) __attribute__((always_inline)) {
    const int nt = NWI + na, pairs = na < NWI ? na : NWI;
#pragma unroll
    for (int t = 0; t < NWI + NI; ++t) {
      if (t >= nt || (t * 6) / nt != gi) continue;
      if (t < 2 * pairs) { if (t % 2 == 0) wt(t / 2); else at(t / 2); }
      else if (na < NWI) wt(pairs + t - 2 * pairs);
      else at(pairs + t - 2 * pairs);
    }
  };
  auto unit = [&](auto last_tag, int p, int g, f32x16 (&s_m1)[ROWS], f32x16 (&s_0)[ROWS], f32x16 (&s_p1)[ROWS]) __attribute__((always_inline)) {
    constexpr bool LAST = decltype(last_tag)::value;
    const char* apl = smem + (u & 1) * Cfg::PLANE;
    const int gn = LAST ? 0 : g + 1, pn = LAST ? p + 1 : p;             // the next unit
    // (past the segment's last unit the copies turn into zero-fills: same instruction count for the counted waits)
    const ActSrc nx = act_src(pn, pn <= d1 ? gn : ng);
    const int nslot = (u + 1) & 1;
    int e[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) e[i] = vtab[i * 256];
    const bool do2 = p - 1 >= d0 && p <= d1, do1 = p >= d0 && p < d1, do0 = p + 1 < d1;
    const bool r0 = w_runs(p, 0), r2n = w_runs(pn, 2), r1n = w_runs(pn, 1);
    auto nothing = [&](const int) __attribute__((always_inline)) {};
    // kd = 2 (weight slot 0); requests: this unit's kd = 0 weights -> slot 2, first half of the next plane unit
    block(do2, apl, wl + 0 * Cfg::WUNIT, s_m1, nothing, [&](const int gi) __attribute__((always_inline)) {
      tickets(gi, NA0, [&](const int i) __attribute__((always_inline)) { w_ticket(r0, g, 0, 2, i); },
              [&](const int i) __attribute__((always_inline)) { act_ticket(nx, nslot, i, e[i]); });
    });
    dma_wait_but<NWI + NA0>();
    mg_barrier();
    // kd = 1 (slot 1); requests: the next unit's kd = 2 weights -> slot 0, second half of the next plane unit.  In the plane's
    // last group output plane p - 1 is complete: its rows are converted and stored under this block's MFMAs.
    block(do1, apl, wl + 1 * Cfg::WUNIT, s_0,
          [&](const int gi) __attribute__((always_inline)) { if constexpr (LAST) { if (gi < ROWS) epilogue_row(s_m1, do2 ? p - 1 : -1, gi); } },
          [&](const int gi) __attribute__((always_inline)) {
      tickets(gi, NA1, [&](const int i) __attribute__((always_inline)) { w_ticket(r2n, gn, 2, 0, i); },
              [&](const int i) __attribute__((always_inline)) { act_ticket(nx, nslot, NA0 + i, e[NA0 + i]); });
    });
    dma_wait_but<NWI + NA1>();                 // (the epilogue's stores are NOT counted: see the note on counted waits above)
    mg_barrier();
    // kd = 0 (slot 2); requests: the next unit's kd = 1 weights -> slot 1
    block(do0, apl, wl + 2 * Cfg::WUNIT, s_p1, nothing, [&](const int gi) __attribute__((always_inline)) {
      tickets(gi, 0, [&](const int i) __attribute__((always_inline)) { w_ticket(r1n, gn, 1, 1, i); }, nothing);
    });
    dma_wait_but<NWI>();                       // ... which stay in flight; everything older (the whole next plane unit) has landed
    mg_barrier();
    ++u;
  };
  auto plane = [&](int p, f32x16 (&s_m1)[ROWS], f32x16 (&s_0)[ROWS], f32x16 (&s_p1)[ROWS]) __attribute__((always_inline)) {
    for (int g = 0; g + 1 < ng; ++g) unit(std::false_type{}, p, g, s_m1, s_0, s_p1);
    unit(std::true_type{}, p, ng - 1, s_m1, s_0, s_p1);
  };

  // prologue: first plane unit, the weights of its kd = 2 and kd = 1 blocks
  load_act(d0 - 1, 0, 0);
  load_w(d0 - 1, 0, 2, 0);
  load_w(d0 - 1, 0, 1, 1);
  dma_wait_all();
  __syncthreads();
  {
    f32x16 acc[3][ROWS];
    zero_set(acc[0]); zero_set(acc[1]); zero_set(acc[2]);
    // input planes d0 - 1 .. d1 in triples (the rotation of the three register sets returns to its start after three planes,
    // so the loop body needs no control flow between the planes -- with `if (p > d1) skip` joins inside the loop hipcc
    // copied and spilled whole 64-register sets); the one or two planes left over run behind the loop, nested, where no
    // accumulator is live across a join.
    const int np = d1 - d0 + 2;
    int pb = d0 - 1;
    for (int t3 = 0; t3 < np / 3; ++t3) {
      plane(pb, acc[2], acc[0], acc[1]);
      plane(pb + 1, acc[0], acc[1], acc[2]);
      plane(pb + 2, acc[1], acc[2], acc[0]);
      pb += 3;
    }
    if (np % 3 >= 1) {
      plane(pb, acc[2], acc[0], acc[1]);
      if (np % 3 == 2) plane(pb + 1, acc[0], acc[1], acc[2]);
    }
  }
  dma_wait_all();
  __syncthreads();

  if (a.stats) {
    // one row of partial statistics per workgroup: the 32 voxel lanes of each half by shuffles, then the 4 waves through
    // LDS, fixed order (deterministic)
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
      float* p = a.stats + ((long long)tile * 2) * a.coutp;
      p[co_base + r] = t1;
      p[a.coutp + co_base + r] = t2;
    }
  }
}
