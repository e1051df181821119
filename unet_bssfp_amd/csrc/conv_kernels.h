// Implicit-GEMM convolution kernels (NDHWC, MFMA 32x32, 64-lane waves, 4 waves per workgroup).
//
// GEMM view: D[voxel][cout] += A[voxel][k] * B[k][cout], k = (tap, cin).  A = activations
// (MFMA rows = 32 voxels of a subtile), B = packed weights (MFMA cols = 32 output channels), so
// each lane ends up with ONE output channel and 16 voxels: channel statistics are lane-local
// sums and every store instruction writes two full 128-B (f32) channel rows.
//
//  conv_halo_kernel    : KS^3 stride-1, KS = 3 (U-Net convs and their data gradients) or KS = 2
//      (PatchGAN k4s2 convs re-expressed as dense k2s1 convs on space-to-depth tensors, and the
//      data gradient of the k2s2 transposed convolution's forward).
//      The workgroup stages the (TD+KS-1)x(TH+KS-1)x(TW+KS-1) halo of a 16-channel chunk in LDS once and
//      re-reads it KS^3 times (LDS 256 B/clk/CU instead of L1 64 B/clk/CU); the next chunk's halo
//      is prefetched into registers while the MFMAs run.
//  conv_gather_kernel  : any cubic kernel / stride / padding (PatchGAN k4s2, transposed-conv
//      parity classes, 1x1x1): A fragments are gathered straight from global memory.
#pragma once
#include "conv_common.h"

template <typename T, int KS, int TD, int TH, int TW, int CT, int NW = 4>
__global__ __launch_bounds__(NW * 64, (sizeof(T) == 2 && CT == 2 && NW == 4) ? 3 : 1) void conv_halo_kernel(const ConvArgs a) {
  constexpr int ES = sizeof(T);
  constexpr int RS = 32 / TW;                 // rows (h) per 32-voxel subtile
  constexpr int SPD = TH / RS;                // subtiles per d-slice
  constexpr int NSUB = TD * SPD;
  constexpr int VT = NSUB / NW;
  constexpr int NTHR = NW * 64;
  static_assert(NSUB % NW == 0 && TW * RS == 32 && TH % RS == 0, "tile shape");
  constexpr int NTAP = KS * KS * KS;
  constexpr int HD = TD + KS - 1, HH = TH + KS - 1, HW = TW + KS - 1;
  constexpr int VS = 16 * ES + 16;            // LDS bytes per halo voxel (+16: bank spread)
  constexpr int PPV = ES;                     // 16-B pieces per voxel (16 ch * ES / 16)
  constexpr int NPIECE = HD * HH * HW * PPV;
  constexpr int NP = (NPIECE + NTHR - 1) / NTHR;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int co_base = blockIdx.y * (32 * CT);

  // XCD-aware tile order: workgroups are dealt round-robin over the 8 XCDs, so give each XCD a contiguous
  // range of tiles (neighbouring tiles share halo lines in that XCD's L2).  Bijective for any grid size.
  int tile;
  {
    const int nwg = gridDim.x, q = nwg >> 3, rem = nwg & 7, xcd = blockIdx.x & 7, k = blockIdx.x >> 3;
    tile = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + k;
  }
  // halo position of each of this thread's 16-B pieces, packed (hd << 16 | hh << 8 | hw): tile-invariant
  int hpos[NP];
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    const int p = tid + i * NTHR;
    const int vox = p / PPV;
    hpos[i] = p < NPIECE ? (((vox / (HW * HH)) << 16) | (((vox / HW) % HH) << 8) | (vox % HW)) : -1;
  }
  int gvox[NP];
  int tn, d0, h0, w0;                         // origin of the tile whose loads are being issued
  auto plan_tile = [&](int t) {
    const int tw_i = t % a.tiles_w; t /= a.tiles_w;
    const int th_i = t % a.tiles_h; t /= a.tiles_h;
    const int td_i = t % a.tiles_d;
    tn = t / a.tiles_d;
    d0 = td_i * TD; h0 = th_i * TH; w0 = tw_i * TW;
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int hp = hpos[i];
      const int gd = d0 + (hp >> 16) - a.pd, gh = h0 + ((hp >> 8) & 255) - a.ph, gw = w0 + (hp & 255) - a.pw;
      const bool ok = hp >= 0 && gd >= 0 && gd < a.di && gh >= 0 && gh < a.hi && gw >= 0 && gw < a.wi;
      gvox[i] = ok ? ((tn * a.di + gd) * a.hi + gh) * a.wi + gw : -1;
    }
  };
  uint4 stage[NP];
  auto load_chunk = [&](int c) {
    const int cb = c * 16;
    const bool first = cb < a.c0;
    const char* src = first ? a.x0 : a.x1;
    const long long ld = first ? a.ld0 : a.ld1;
    const int cbase = first ? cb : cb - a.c0;
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int part = (tid + i * NTHR) % PPV;
      if (gvox[i] >= 0)   // (a non-temporal load here was measured slower: halo overlap re-reads then miss L2)
        stage[i] = *reinterpret_cast<const uint4*>(src + ((long long)gvox[i] * ld + cbase) * ES + part * 16);
      else
        stage[i] = make_uint4(0, 0, 0, 0);
    }
  };
  auto store_chunk = [&]() {
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int p = tid + i * NTHR;
      if (p < NPIECE) *reinterpret_cast<uint4*>(smem + (p / PPV) * VS + (p % PPV) * 16) = stage[i];
    }
  };

  // ---- per-lane LDS base of each subtile's A fragment ----
  int lbase[VT];
#pragma unroll
  for (int vt = 0; vt < VT; ++vt) {
    const int s = wave * VT + vt;
    const int sd = s / SPD, sh = (s % SPD) * RS;
    lbase[vt] = ((sd * HH + sh + r / TW) * HW + (r % TW)) * VS + h * 16;
  }
  const char* wlane = a.wp + ((long long)co_base + r) * (16 * ES) + h * 16;
  const long long wtap = (long long)a.coutp * (16 * ES);   // bytes per (chunk, tap)

  f32x16 acc[VT][CT];
#pragma unroll
  for (int vt = 0; vt < VT; ++vt)
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[vt][ct][i] = 0.f;

  plan_tile(tile);
  // split-K: this workgroup contracts the chunk range [c_begin, c_end) only
  const int cps = (a.nchunks + a.ksplit - 1) / a.ksplit;
  const int c_begin = blockIdx.z * cps;
  const int c_end = min(a.nchunks, c_begin + cps);
  if (c_begin < c_end) load_chunk(c_begin);
  for (int c = c_begin; c < c_end; ++c) {
    __syncthreads();                 // previous chunk's LDS reads are done
    store_chunk();
    __syncthreads();
    if (c + 1 < c_end) load_chunk(c + 1);       // in flight under the MFMAs
    const char* wc = wlane + (long long)c * NTAP * wtap;
    // weight fragments are fetched PG taps at a time (all loads of a group in flight together);
    // the group size was measured irrelevant (3 / 9 / 27): the L2-resident weights are not the limiter
    constexpr int PG = (ES == 2) ? (KS == 3 ? 9 : NTAP) : KS;
#pragma unroll
    for (int g0 = 0; g0 < NTAP; g0 += PG) {
      Frag<T> b[PG][CT];
#pragma unroll
      for (int t = 0; t < PG; ++t)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) b[t][ct].load(wc + (g0 + t) * wtap + ct * (32 * 16 * ES));
#pragma unroll
      for (int t = 0; t < PG; ++t) {
        const int tap = g0 + t;
        const int kd = tap / (KS * KS), kh = (tap / KS) % KS, kw = tap % KS;
        const int toff = ((kd * HH + kh) * HW + kw) * VS;
#pragma unroll
        for (int vt = 0; vt < VT; ++vt) {
          Frag<T> af;
          af.load(smem + lbase[vt] + toff);
#pragma unroll
          for (int ct = 0; ct < CT; ++ct) mma16(af, b[t][ct], acc[vt][ct]);
        }
      }
    }
  }

  // ---- epilogue (rows of a subtile are an affine function of the accumulator index) ----
  TileOut<VT> to;
  to.hstride = (long long)a.os * a.wy * a.ldy;
  to.wstride = (long long)a.os * a.ldy;
#pragma unroll
  for (int vt = 0; vt < VT; ++vt) {
    const int s = wave * VT + vt;
    const int gd = d0 + s / SPD, gh = h0 + (s % SPD) * RS, gw = w0;
    to.base[vt] = ((((long long)tn * a.dy + (gd * a.os + a.od)) * a.hy + (gh * a.os + a.oh)) * a.wy + (gw * a.os + a.ow)) * a.ldy;
    to.dvalid[vt] = gd < a.do_;
    to.hleft[vt] = a.ho - gh;           // rows with index < hleft are inside
    to.wleft[vt] = a.wo - gw;
  }
  if (a.ksplit > 1) {
    // partial sums go to the f32 slab of this split; bias, statistics and the store in T happen in
    // conv_ksplit_reduce_kernel (fixed summation order => deterministic)
    ConvArgs s2 = a;
    s2.y = reinterpret_cast<char*>(a.kslab + (long long)blockIdx.z * a.m_total * a.coutp);
    s2.ldy = a.coutp; s2.cstore = a.coutp; s2.bias = nullptr; s2.stats = nullptr;
    to.hstride = (long long)a.wo * a.coutp;
    to.wstride = a.coutp;
#pragma unroll
    for (int vt = 0; vt < VT; ++vt) {
      const int s = wave * VT + vt;
      const int gd = d0 + s / SPD, gh = h0 + (s % SPD) * RS, gw = w0;
      to.base[vt] = ((((long long)tn * a.do_ + gd) * a.ho + gh) * a.wo + gw) * a.coutp;
    }
    conv_epilogue_tile<float, VT, CT, TW, NW>(s2, acc, to, co_base, tile, reinterpret_cast<float*>(smem));
    return;
  }
  conv_epilogue_tile<T, VT, CT, TW, NW>(a, acc, to, co_base, tile, reinterpret_cast<float*>(smem));
}

// ------------------------------------------------------------------------------------------
// conv_lowg_kernel: 3x3x3 stride-1 convolution (and data gradient) of the LOW levels in bf16 -- 16^3 / 8^3 of the U-Net
// (src/model.py:22-28: down_3, down_4, upcat_4, upcat_3): 4 096 / 512 output positions, 128 ... 512 channels either side.
// conv_halo_kernel ran these at 0.25 - 0.5 PFLOP/s whatever its tile or split: every WAVE fetched its own
// weight fragments from L2, 27 KB per 16-channel chunk in three dependent groups, each fragment used once -- L2 latency,
// not bandwidth, not the grid.  Here
//   * a workgroup owns 512 output positions (TD x TH x TW, four 32-voxel sub-tiles per wave) x 64 output channels:
//     a weight fragment feeds 4 MFMAs per wave, an activation fragment 2;
//   * weights go through LDS ONCE per workgroup: block (chunk, kd) = 9 taps x 64 channels x 32 B in a 48-byte-pitch slot
//     (conflict-free ds_read_b128: 16 lanes x 48 B cover the 16 bank groups), two slots; the copy for block b + 2 is in
//     flight in registers (loaded during block b, stored to the free slot at the start of block b + 1) -- one block =
//     72 MFMAs per wave = 2 304 cycles of cover for an L2 hit;
//   * the halo of the next 16-channel chunk is prefetched into registers as in conv_halo_kernel.
// Split-K over chunks (blockIdx.z) into the f32 slabs that conv_ksplit_reduce_kernel combines; one workgroup per CU
// (107 KB of LDS), so the plan aims at 256 workgroups.
// ------------------------------------------------------------------------------------------
template <int TD, int TH, int TW>
struct LowGCfg {
  static constexpr int HD = TD + 2, HH = TH + 2, HW = TW + 2, VS = 48;
  static constexpr int HALO = HD * HH * HW * VS;
  static constexpr int WROW = 48, WSLOT = 9 * 64 * WROW;
  static constexpr int LDS = 2 * HALO + 2 * WSLOT;            // two halo buffers, two weight slots: 158 976 B for 4x8x16
};

template <int TD, int TH, int TW>
__global__ __launch_bounds__(256, 1) void conv_lowg_kernel(const ConvArgs a) {
  using T = bf16_t;
  using Cfg = LowGCfg<TD, TH, TW>;
  constexpr int CT = 2, NW = 4;
  constexpr int RS = 32 / TW, SPD = TH / RS, NSUB = TD * SPD, VT = NSUB / NW;
  static_assert(NSUB % NW == 0 && TW * RS == 32 && TH % RS == 0 && VT == 4, "tile shape");
  constexpr int HH = Cfg::HH, HW = Cfg::HW, VS = Cfg::VS;
  constexpr int NPIECE = Cfg::HD * HH * HW * 2, NP = (NPIECE + 255) / 256;      // 16-B pieces of a chunk's halo
  constexpr int NWP = 9 * 64 * 2, NWI = (NWP + 255) / 256;                       // ... of a weight slot
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const wl = smem + 2 * Cfg::HALO;
  // piece -> (voxel or weight row, 16-byte half).  Round 3 took (p >> 1, p & 1): lanes 2 v and 2 v + 1 wrote the two halves of
  // row v, so 16 consecutive lanes covered 8 rows x 32 of every 48 bytes -- 13 of the 16 bank groups, three of them twice: the
  // ds_write_b128 took 16 cycles instead of 8 (tests/diag/lds_rw_bench.hip: SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE 0.50 for that
  // write pattern, 0 for the kernel's reads: THIS was the 42 % of profiles/r03_pmc_conv_lowg.txt).  Now 16 consecutive lanes
  // write the same half of 16 consecutive rows (row * 48: all 16 bank groups, the property of the fragment reads), the next
  // 16 lanes the other half; a trailing group of fewer than 32 pieces keeps the plain order.
#ifdef LOWG_DIAG_PLAIN_PIECES      // (timing A/B of round 3's order: tools/build_variants.sh plain="-DLOWG_DIAG_PLAIN_PIECES")
  auto piece_row = [](int p, int) __attribute__((always_inline)) { return p >> 1; };
  auto piece_half = [](int p, int) __attribute__((always_inline)) { return p & 1; };
#else
  auto piece_row = [](int p, int np) __attribute__((always_inline)) { return (p | 31) < np ? ((p & ~31) >> 1) + (p & 15) : p >> 1; };
  auto piece_half = [](int p, int np) __attribute__((always_inline)) { return (p | 31) < np ? (p >> 4) & 1 : p & 1; };
#endif

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int co_base = blockIdx.y * 64;
  int tile;
  {
    const int nwg = gridDim.x, q = nwg >> 3, rem = nwg & 7, xcd = blockIdx.x & 7, k = blockIdx.x >> 3;
    tile = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + k;
  }
  int tn, d0, h0, w0;
  {
    int t = tile;
    const int tw_i = t % a.tiles_w; t /= a.tiles_w;
    const int th_i = t % a.tiles_h; t /= a.tiles_h;
    const int td_i = t % a.tiles_d;
    tn = t / a.tiles_d;
    d0 = td_i * TD; h0 = th_i * TH; w0 = tw_i * TW;
  }
  // halo pieces of this thread: global voxel index (or -1) and LDS offset, chunk-invariant
  int gvox[NP];
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    const int p = min(tid + i * 256, NPIECE - 1), vox = piece_row(p, NPIECE);   // (threads beyond the last piece duplicate it)
    const int hd = vox / (HW * HH), hh = (vox / HW) % HH, hw = vox % HW;
    const int gd = d0 + hd - a.pd, gh = h0 + hh - a.ph, gw = w0 + hw - a.pw;
    const bool ok = gd >= 0 && gd < a.di && gh >= 0 && gh < a.hi && gw >= 0 && gw < a.wi;
    gvox[i] = ok ? ((tn * a.di + gd) * a.hi + gh) * a.wi + gw : -1;
  }
  const int cps = (a.nchunks + a.ksplit - 1) / a.ksplit;
  const int c_begin_of = blockIdx.z * cps;                 // split-K: this workgroup contracts chunks [c_begin_of, c_end)
  const int c_end = min(a.nchunks, c_begin_of + cps);
  uint4 stage[NP];
#define LOWG_LOAD_CHUNK(C)                                                                                       \
  {                                                                                                              \
    const int cb_ = (C) * 16;                                                                                    \
    const bool first_ = cb_ < a.c0;                                                                              \
    const char* src_ = first_ ? a.x0 : a.x1;                                                                     \
    const long long ld_ = first_ ? a.ld0 : a.ld1;                                                                \
    const int cbase_ = first_ ? cb_ : cb_ - a.c0;                                                                \
    _Pragma("unroll") for (int i = 0; i < NP; ++i) {                                                             \
      const int part_ = piece_half(min(tid + i * 256, NPIECE - 1), NPIECE);                                      \
      if (gvox[i] >= 0) stage[i] = *reinterpret_cast<const uint4*>(src_ + ((long long)gvox[i] * ld_ + cbase_) * 2 + part_ * 16); \
      else stage[i] = make_uint4(0, 0, 0, 0);                                                                    \
    }                                                                                                            \
  }
#define LOWG_STORE_CHUNK_PART(HB, K3) /* a third of the staged chunk's pieces */                                 \
  {                                                                                                              \
    constexpr int n3_ = (NP + 2) / 3, lo_ = (K3) * n3_, hi_ = lo_ + n3_ < NP ? lo_ + n3_ : NP;                   \
    _Pragma("unroll") for (int i = lo_; i < hi_; ++i) {                                                          \
      const int p_ = min(tid + i * 256, NPIECE - 1);   /* (clamped threads rewrite the last piece with its own value) */ \
      *reinterpret_cast<uint4*>(smem + (HB) * Cfg::HALO + piece_row(p_, NPIECE) * VS + piece_half(p_, NPIECE) * 16) = stage[i]; \
    }                                                                                                            \
  }
#define LOWG_STORE_CHUNK(HB)                                                                                     \
  {                                                                                                              \
    _Pragma("unroll") for (int i = 0; i < NP; ++i) {                                                             \
      const int p_ = tid + i * 256;                                                                              \
      if (p_ < NPIECE) *reinterpret_cast<uint4*>(smem + (HB) * Cfg::HALO + piece_row(p_, NPIECE) * VS + piece_half(p_, NPIECE) * 16) = stage[i]; \
    }                                                                                                            \
  }
  // weight slot of block (chunk c, kd): packed weights are [chunk][27 taps][coutp][16 channels].  TWO register sets: the
  // copy of block b + 1 is stored to the free slot at the start of block b and its set re-used for block b + 3, while
  // the other set holds block b + 2 -- every copy has two blocks (~4 600 cycles) of flight time, enough for a weight
  // slice that comes from HBM (one workgroup per CU: nothing else hides that latency).
  // (two plain arrays and macros: an array handed to a lambda by reference stayed in scratch memory -- a scratch store
  //  right behind every global load, i.e. the full load latency exposed per block: 3.6 us per block instead of 1.1)
  static_assert(NWI == 5, "five named registers per set below");
  uint4 wa0, wa1, wa2, wa3, wa4, wb0, wb1, wb2, wb3, wb4;      // (named scalars: arrays of these ended up in scratch memory)
  auto w_src = [&](int b, int i) __attribute__((always_inline)) {
    const int c = c_begin_of + b / 3, kd = b - (b / 3) * 3;
    const int p = min(tid + i * 256, NWP - 1);              // piece = (row = tap9 * 64 + cout, half)
    const int row = piece_row(p, NWP);
    return reinterpret_cast<const uint4*>(a.wp + ((((long long)c * 27 + kd * 9 + (row >> 6)) * a.coutp + co_base + (row & 63)) * 32 + piece_half(p, NWP) * 16));
  };
  auto w_dst = [&](int slot, int i) __attribute__((always_inline)) {
    const int p = min(tid + i * 256, NWP - 1);              // (the clamped threads rewrite piece NWP - 1 with its own value)
    return reinterpret_cast<uint4*>(wl + slot * Cfg::WSLOT + piece_row(p, NWP) * Cfg::WROW + piece_half(p, NWP) * 16);
  };
#define LOWG_LOAD_W(W, B) { W##0 = *w_src(B, 0); W##1 = *w_src(B, 1); W##2 = *w_src(B, 2); W##3 = *w_src(B, 3); W##4 = *w_src(B, 4); }
#define LOWG_STORE_W(W, SLOT) { *w_dst(SLOT, 0) = W##0; *w_dst(SLOT, 1) = W##1; *w_dst(SLOT, 2) = W##2; *w_dst(SLOT, 3) = W##3; *w_dst(SLOT, 4) = W##4; }

  int lbase[VT];
#pragma unroll
  for (int vt = 0; vt < VT; ++vt) {
    const int s = wave * VT + vt;
    const int sd = s / SPD, sh = (s % SPD) * RS;
    lbase[vt] = ((sd * HH + sh + r / TW) * HW + (r % TW)) * VS + h * 16;
  }
  const int wfrag = r * Cfg::WROW + h * 16;

  f32x16 acc[VT][CT];
#pragma unroll
  for (int vt = 0; vt < VT; ++vt)
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[vt][ct][i] = 0.f;

  const int nblk = (c_end - c_begin_of) * 3;
  if (nblk > 0) {
    LOWG_LOAD_CHUNK(c_begin_of)
    LOWG_LOAD_W(wa, 0)
    LOWG_LOAD_W(wb, 1)                                     // (nblk >= 3)
    LOWG_STORE_W(wa, 0)                                    // block 0 -> slot 0
    LOWG_STORE_CHUNK(0)                                    // chunk 0 -> halo buffer 0
    LOWG_LOAD_W(wa, 2)
    if (c_begin_of + 1 < c_end) LOWG_LOAD_CHUNK(c_begin_of + 1)
  }
  // Block b = 6 q + B6: weight slot P = B6 & 1, kd = B6 % 3, halo buffer HB = (B6 / 3) & 1 -- all compile-time.  ONE barrier
  // per block, at its start: it publishes what block b - 1 wrote (slot P; a third of the halo buffer) and frees what block
  // b - 1 read (slot 1 - P; after a chunk's last block its halo buffer).  During block b the wave writes block b + 1's
  // weights into slot 1 - P and a third of the NEXT chunk's halo into the other buffer -- unconditional LDS writes pinned
  // between the MFMAs (timing-only builds of the two-barrier form: copy phases and MFMAs did not overlap at all, 2.2 + 3.3 us
  // per chunk); the global loads that refill the registers follow the MFMA stream.  wb / wa (P = 0 / 1) holds block
  // b + 1, the other set block b + 2; `stage` holds the next chunk.  Writes of a block that does not exist land in free
  // buffers.
  auto block = [&](int b, auto b6) __attribute__((always_inline)) {
    constexpr int B6 = decltype(b6)::value, P = B6 & 1, kd = B6 % 3, HB = (B6 / 3) & 1;
    const int c = c_begin_of + b / 3;
    __syncthreads();
    const char* ws = wl + P * Cfg::WSLOT + wfrag;
    const char* hs = smem + HB * Cfg::HALO + kd * HH * HW * VS;
    // one wave per SIMD: nobody else hides the LDS latency, so the 6 fragment reads of tap t + 1 are issued between the 8
    // MFMAs of tap t (two register sets, the order pinned: left to itself hipcc read each fragment right before its use)
    uint4 f0a, f0b, f0c, f0d, f0e, f0f, f1a, f1b, f1c, f1d, f1e, f1f;
#define LOWG_READ(S, T9)                                                                                         \
  {                                                                                                              \
    const int t_ = ((T9) / 3 * HW + (T9) % 3) * VS;                                                              \
    S##a = *reinterpret_cast<const uint4*>(ws + ((T9) * 64) * Cfg::WROW);                                        \
    S##b = *reinterpret_cast<const uint4*>(ws + ((T9) * 64 + 32) * Cfg::WROW);                                   \
    S##c = *reinterpret_cast<const uint4*>(hs + lbase[0] + t_);                                                  \
    S##d = *reinterpret_cast<const uint4*>(hs + lbase[1] + t_);                                                  \
    S##e = *reinterpret_cast<const uint4*>(hs + lbase[2] + t_);                                                  \
    S##f = *reinterpret_cast<const uint4*>(hs + lbase[3] + t_);                                                  \
  }
#define LOWG_MMA(S)                                                                                              \
  {                                                                                                              \
    Frag<T> b0_, b1_, a_;                                                                                        \
    b0_.v = S##a; b1_.v = S##b;                                                                                  \
    a_.v = S##c; mma16(a_, b0_, acc[0][0]); mma16(a_, b1_, acc[0][1]);                                           \
    a_.v = S##d; mma16(a_, b0_, acc[1][0]); mma16(a_, b1_, acc[1][1]);                                           \
    a_.v = S##e; mma16(a_, b0_, acc[2][0]); mma16(a_, b1_, acc[2][1]);                                           \
    a_.v = S##f; mma16(a_, b0_, acc[3][0]); mma16(a_, b1_, acc[3][1]);                                           \
  }
#define LOWG_PIN()                                                                                               \
  {                                                                                                              \
    _Pragma("unroll") for (int q_ = 0; q_ < 6; ++q_) {                                                           \
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      /* one MFMA */                                     \
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);      /* one LDS read */                                 \
    }                                                                                                            \
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                           \
    __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);        /* one LDS write (next block's weights / next halo) */ \
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                           \
  }
  // one LDS write per tap, in program order BETWEEN the reads of tap t + 1 and those of tap t + 2 (hipcc keeps LDS writes
  // and reads in program order: it cannot know that they touch different buffers)
#define LOWG_WW(I) { if constexpr (P == 0) *w_dst(1, I) = wb##I; else *w_dst(0, I) = wa##I; }
#define LOWG_WS(J)                                                                                               \
  {                                                                                                              \
    constexpr int n3_ = (NP + 2) / 3, i_ = kd * n3_ + (J);                                                       \
    if constexpr ((J) < n3_ && i_ < NP) {                                                                        \
      const int p_ = min(tid + i_ * 256, NPIECE - 1);   /* (clamped threads rewrite the last piece with its own value) */ \
      *reinterpret_cast<uint4*>(smem + (1 - HB) * Cfg::HALO + piece_row(p_, NPIECE) * VS + piece_half(p_, NPIECE) * 16) = stage[i_]; \
    }                                                                                                            \
  }
#ifndef LOWG_DIAG_NO_MMA
    LOWG_READ(f0, 0)
    LOWG_READ(f1, 1) LOWG_WW(0) LOWG_MMA(f0) LOWG_PIN()
    LOWG_READ(f0, 2) LOWG_WW(1) LOWG_MMA(f1) LOWG_PIN()
    LOWG_READ(f1, 3) LOWG_WW(2) LOWG_MMA(f0) LOWG_PIN()
    LOWG_READ(f0, 4) LOWG_WW(3) LOWG_MMA(f1) LOWG_PIN()
    LOWG_READ(f1, 5) LOWG_WW(4) LOWG_MMA(f0) LOWG_PIN()
    LOWG_READ(f0, 6) LOWG_WS(0) LOWG_MMA(f1) LOWG_PIN()
    LOWG_READ(f1, 7) LOWG_WS(1) LOWG_MMA(f0) LOWG_PIN()
    LOWG_READ(f0, 8) LOWG_WS(2) LOWG_MMA(f1) LOWG_PIN()
    LOWG_MMA(f0)
#else
    LOWG_WW(0) LOWG_WW(1) LOWG_WW(2) LOWG_WW(3) LOWG_WW(4) LOWG_WS(0) LOWG_WS(1) LOWG_WS(2)
#endif
#undef LOWG_WW
#undef LOWG_WS
#undef LOWG_READ
#undef LOWG_MMA
#undef LOWG_PIN
    // refill the registers that were just stored (the copies have one to two blocks of flight time)
    if (b + 3 < nblk) {
      if constexpr (P == 0) LOWG_LOAD_W(wb, b + 3) else LOWG_LOAD_W(wa, b + 3)
    }
    if (kd == 2 && c + 2 < c_end) LOWG_LOAD_CHUNK(c + 2)
  };
  for (int b = 0; b < nblk; b += 6) {
    block(b, std::integral_constant<int, 0>{});
    block(b + 1, std::integral_constant<int, 1>{});
    block(b + 2, std::integral_constant<int, 2>{});
    if (b + 3 < nblk) {
      block(b + 3, std::integral_constant<int, 3>{});
      block(b + 4, std::integral_constant<int, 4>{});
      block(b + 5, std::integral_constant<int, 5>{});
    }
  }
#undef LOWG_STORE_CHUNK_PART
#undef LOWG_LOAD_W
#undef LOWG_LOAD_CHUNK
#undef LOWG_STORE_CHUNK
#undef LOWG_STORE_W

  // ---- epilogue: as conv_halo_kernel ----
#ifdef LOWG_DIAG_NO_STORE
  if (acc[0][0][0] != 12345.678f) return;      // (timing-only build)
#endif
  TileOut<VT> to;
  to.hstride = (long long)a.os * a.wy * a.ldy;
  to.wstride = (long long)a.os * a.ldy;
#pragma unroll
  for (int vt = 0; vt < VT; ++vt) {
    const int s = wave * VT + vt;
    const int gd = d0 + s / SPD, gh = h0 + (s % SPD) * RS, gw = w0;
    to.base[vt] = ((((long long)tn * a.dy + (gd * a.os + a.od)) * a.hy + (gh * a.os + a.oh)) * a.wy + (gw * a.os + a.ow)) * a.ldy;
    to.dvalid[vt] = gd < a.do_;
    to.hleft[vt] = a.ho - gh;
    to.wleft[vt] = a.wo - gw;
  }
  if (a.ksplit > 1) {
    ConvArgs s2 = a;
    s2.y = reinterpret_cast<char*>(a.kslab + (long long)blockIdx.z * a.m_total * a.coutp);
    s2.ldy = a.coutp; s2.cstore = a.coutp; s2.bias = nullptr; s2.stats = nullptr;
    to.hstride = (long long)a.wo * a.coutp;
    to.wstride = a.coutp;
#pragma unroll
    for (int vt = 0; vt < VT; ++vt) {
      const int s = wave * VT + vt;
      const int gd = d0 + s / SPD, gh = h0 + (s % SPD) * RS, gw = w0;
      to.base[vt] = ((((long long)tn * a.do_ + gd) * a.ho + gh) * a.wo + gw) * a.coutp;
    }
    __syncthreads();                 // every wave is done with the halo and the slots: the epilogue reuses the LDS
    conv_epilogue_tile<float, VT, CT, TW, NW>(s2, acc, to, co_base, tile, reinterpret_cast<float*>(smem));
    return;
  }
  __syncthreads();
  conv_epilogue_tile<T, VT, CT, TW, NW>(a, acc, to, co_base, tile, reinterpret_cast<float*>(smem));
}

// Combine the split-K slabs: z = sum_k slab[k] + bias, store in T at the (possibly scattered) output
// position, and emit the per-block channel statistics of (z - bias).  One block = rpb positions.
template <typename T>
__global__ __launch_bounds__(256) void conv_ksplit_reduce_kernel(const ConvArgs a, int rpb) {
  __shared__ float red[256 * 8];
  const int cblk = blockIdx.y * 1024;          // channel block of this workgroup (<= 1024 channels)
  const int cw = min(1024, a.coutp - cblk);
  const int lpr = cw / 4;                      // float4 pieces per position (<= 256)
  const int rpp = 256 / lpr;
  const int piece = threadIdx.x % lpr, rsub = threadIdx.x / lpr;
  const int ch0 = cblk + piece * 4;
  float s0[4] = {0.f, 0.f, 0.f, 0.f}, s1[4] = {0.f, 0.f, 0.f, 0.f};
  float bias[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) bias[j] = (a.bias && ch0 + j < a.nbias) ? a.bias[ch0 + j] : 0.f;
  const long long r0 = (long long)blockIdx.x * rpb;
  if (rsub < rpp)
    for (long long row = r0 + rsub; row < r0 + rpb && row < a.m_total; row += rpp) {
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      const float* sp = a.kslab + row * a.coutp + ch0;
      const long long ss = a.m_total * a.coutp;
      int k = 0;
      for (; k + 4 <= a.ksplit; k += 4) {          // 4 independent loads in flight, fixed summation order
        const float4 p0 = *reinterpret_cast<const float4*>(sp + (long long)k * ss);
        const float4 p1 = *reinterpret_cast<const float4*>(sp + (long long)(k + 1) * ss);
        const float4 p2 = *reinterpret_cast<const float4*>(sp + (long long)(k + 2) * ss);
        const float4 p3 = *reinterpret_cast<const float4*>(sp + (long long)(k + 3) * ss);
        v.x += (p0.x + p1.x) + (p2.x + p3.x); v.y += (p0.y + p1.y) + (p2.y + p3.y);
        v.z += (p0.z + p1.z) + (p2.z + p3.z); v.w += (p0.w + p1.w) + (p2.w + p3.w);
      }
      for (; k < a.ksplit; ++k) {
        const float4 p = *reinterpret_cast<const float4*>(sp + (long long)k * ss);
        v.x += p.x; v.y += p.y; v.z += p.z; v.w += p.w;
      }
      const float f[4] = {v.x, v.y, v.z, v.w};
      long long t = row;
      const int gw = (int)(t % a.wo); t /= a.wo;
      const int gh = (int)(t % a.ho); t /= a.ho;
      const int gd = (int)(t % a.do_); const long long n = t / a.do_;
      T* yp = reinterpret_cast<T*>(a.y) + (((n * a.dy + (gd * a.os + a.od)) * a.hy + (gh * a.os + a.oh)) * a.wy + (gw * a.os + a.ow)) * a.ldy;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        s0[j] += f[j];
        s1[j] += f[j] * f[j];
        if (ch0 + j < a.cstore) Elem<T>::store(yp + ch0 + j, f[j] + bias[j]);
      }
    }
  if (!a.stats) return;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    red[threadIdx.x * 8 + j] = s0[j];
    red[threadIdx.x * 8 + 4 + j] = s1[j];
  }
  __syncthreads();
  for (int cl = threadIdx.x; cl < cw; cl += 256) {
    const int p = cl / 4, j = cl % 4;
    float t0 = 0.f, t1 = 0.f;
    for (int qq = 0; qq < rpp; ++qq) {
      t0 += red[(qq * lpr + p) * 8 + j];
      t1 += red[(qq * lpr + p) * 8 + 4 + j];
    }
    float* out = a.stats + (long long)blockIdx.x * 2 * a.coutp;
    out[cblk + cl] = t0;
    out[a.coutp + cblk + cl] = t1;
  }
}

template <typename T, int KS, int TD, int TH, int TW, int CT = 2, int NW = 4>
constexpr int conv_halo_lds() {
  // staging halo; reused by the epilogue for the statistics reduction and (16-bit outputs) the
  // per-wave 32x32 transposition patches
  constexpr int halo = (TD + KS - 1) * (TH + KS - 1) * (TW + KS - 1) * (16 * (int)sizeof(T) + 16);
  constexpr int patch = NW * 2048 > NW * CT * 512 ? NW * 2048 : NW * CT * 512;   // transposition patches / stats
  return halo > patch ? halo : patch;
}

// ------------------------------------------------------------------------------------------
template <typename T, int VT, int CT>
__global__ __launch_bounds__(256) void conv_gather_kernel(const ConvArgs a) {
  constexpr int ES = sizeof(T);
  __shared__ float red[4 * CT * 64];
  __shared__ __attribute__((aligned(16))) char tpatch[ES == 2 ? 4 * 2048 : 16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int co_base = blockIdx.y * (32 * CT);

  // this lane's voxel (MFMA row r) in each subtile
  int vn[VT], vd[VT], vh[VT], vw[VT];
  bool vok[VT];
#pragma unroll
  for (int vt = 0; vt < VT; ++vt) {
    long long m = (long long)blockIdx.x * (128 * VT) + (wave * VT + vt) * 32 + r;
    vok[vt] = m < a.m_total;
    if (!vok[vt]) m = 0;
    vw[vt] = (int)(m % a.wo); m /= a.wo;
    vh[vt] = (int)(m % a.ho); m /= a.ho;
    vd[vt] = (int)(m % a.do_);
    vn[vt] = (int)(m / a.do_);
  }
  const char* wlane = a.wp + ((long long)co_base + r) * (16 * ES) + h * 16;
  const long long wtap = (long long)a.coutp * (16 * ES);
  const int ntaps = a.ks * a.ks * a.ks;

  f32x16 acc[VT][CT];
#pragma unroll
  for (int vt = 0; vt < VT; ++vt)
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[vt][ct][i] = 0.f;

  // split-K (few output positions, long contraction -- the transposed convolutions' data gradients at the low levels ran
  // 128 dependent (tap, chunk) steps in 32 workgroups): blockIdx.z takes a contiguous range of the (tap, chunk) pairs
  const int nit = ntaps * a.nchunks, ipz = (nit + a.ksplit - 1) / a.ksplit;
  const int it0 = blockIdx.z * ipz, it1 = min(nit, it0 + ipz);
  for (int tap = it0 / a.nchunks; tap < ntaps && tap * a.nchunks < it1; ++tap) {
    const int kd = tap / (a.ks * a.ks), kh = (tap / a.ks) % a.ks, kw = tap % a.ks;
    long long vox[VT];
#pragma unroll
    for (int vt = 0; vt < VT; ++vt) {
      const int id = vd[vt] * a.stride + kd - a.pd;
      const int ih = vh[vt] * a.stride + kh - a.ph;
      const int iw = vw[vt] * a.stride + kw - a.pw;
      const bool ok = vok[vt] && id >= 0 && id < a.di && ih >= 0 && ih < a.hi && iw >= 0 && iw < a.wi;
      vox[vt] = ok ? (((long long)vn[vt] * a.di + id) * a.hi + ih) * a.wi + iw : -1;
    }
    const int cb0 = max(0, it0 - tap * a.nchunks), cb1 = min(a.nchunks, it1 - tap * a.nchunks);
    for (int c = cb0; c < cb1; ++c) {
      const int cb = c * 16;
      const bool first = cb < a.c0;
      const char* src = first ? a.x0 : a.x1;
      const long long ld = first ? a.ld0 : a.ld1;
      const int cbase = first ? cb : cb - a.c0;
      const char* wc = wlane + ((long long)c * ntaps + tap) * wtap;
      Frag<T> b[CT];
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) b[ct].load(wc + ct * (32 * 16 * ES));
#pragma unroll
      for (int vt = 0; vt < VT; ++vt) {
        Frag<T> af;
        if (vox[vt] >= 0) af.load(src + (vox[vt] * ld + cbase) * ES + h * 16);
        else af.zero();
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) mma16(af, b[ct], acc[vt][ct]);
      }
    }
  }
  if (a.ksplit > 1) {
    // partial sums to the f32 slab of this split; bias, statistics and the store in T: conv_ksplit_reduce_kernel
    ConvArgs s2 = a;
    s2.y = reinterpret_cast<char*>(a.kslab + (long long)blockIdx.z * a.m_total * a.coutp);
    s2.ldy = a.coutp; s2.cstore = a.coutp; s2.bias = nullptr; s2.stats = nullptr;
    long long soff[VT];
#pragma unroll
    for (int vt = 0; vt < VT; ++vt) {
      const long long m = (long long)blockIdx.x * (128 * VT) + (wave * VT + vt) * 32 + r;
      soff[vt] = vok[vt] ? m * a.coutp : -1;
    }
    conv_epilogue<float, VT, CT>(s2, acc, soff, co_base, blockIdx.x, red, co_base, nullptr);
    return;
  }

  // transposed-conv classes folded into the cout index: this workgroup's 32*CT columns belong to ONE class
  int od = a.od, oh = a.oh, ow = a.ow, co_store = co_base;
  if (a.cls_cout) {
    const int cls = co_base / a.cls_cout;
    co_store = co_base - cls * a.cls_cout;
    od = cls >> 2; oh = (cls >> 1) & 1; ow = cls & 1;
  }
  long long yoff[VT];
#pragma unroll
  for (int vt = 0; vt < VT; ++vt) {
    yoff[vt] = vok[vt] ? ((((long long)vn[vt] * a.dy + (vd[vt] * a.os + od)) * a.hy +
                           (vh[vt] * a.os + oh)) * a.wy + (vw[vt] * a.os + ow)) * a.ldy
                       : -1;
  }
  conv_epilogue<T, VT, CT>(a, acc, yoff, co_base, blockIdx.x, red, co_store, ES == 2 ? tpatch : nullptr);
}

// ------------------------------------------------------------------------------------------
// conv_ru_kernel: bf16 3x3x3 stride-1 convolution for the wide (full-resolution) layers.
//   tile 4(d) x 4(h) x 32(w), 4 waves, wave = one d-slice of 4 rows ("row reuse": the A fragment of halo
//   row (d + kd, hy, kw) feeds every (row, kh) with row + kh = hy, and a weight fragment feeds the 4 rows:
//   54 LDS reads + 27 weight reads per 108 MFMAs per wave and chunk, instead of 108 + 54 -- with 32 output
//   channels per workgroup every fragment has ONE MFMA to feed, so the plain loop is bound by those reads).
//   The halo goes global -> LDS directly (`buffer_load_dwordx4 ... lds`, out-of-range = zero padding): no
//   staging registers, no ds_write pass; image = 2 planes (channels 0-7 / 8-15) of 16 B per voxel, a lane
//   half reads one plane with consecutive voxels 16 B apart (conflict-free without padding: measured 0
//   bank conflicts).  40 KB of LDS and <= 128 VGPRs: 4 workgroups per CU, whose load / MFMA / store phases
//   overlap each other.
// ------------------------------------------------------------------------------------------
constexpr int kRuVox = 6 * 6 * 34;                    // halo voxels
constexpr int kRuBlocks = (kRuVox + 63) / 64;         // 64-voxel wave-instructions per plane
constexpr int kRuPlane = kRuBlocks * 1024;
constexpr int kRuLds = 2 * kRuPlane;

template <int CT>
__global__ __launch_bounds__(256, CT == 1 ? 4 : 2) void conv_ru_kernel(const ConvArgs a) {
  using T = bf16_t;
  constexpr int TD = 4, TH = 4, TW = 32, NW = 4, VT = 4, HH = 6, HW = 34;
  constexpr int NI = (2 * kRuBlocks + NW - 1) / NW;   // LDS-DMA instructions per wave and chunk
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef __attribute__((address_space(3))) void* lds_ptr;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int co_base = blockIdx.y * (32 * CT);
  int tile;
  {
    const int nwg = gridDim.x, q = nwg >> 3, rem = nwg & 7, xcd = blockIdx.x & 7, k = blockIdx.x >> 3;
    tile = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + k;
  }
  // tile id -> tile coordinates.  The ~128 workgroups an XCD runs at a time (32 CUs x 4) should form a compact
  // 3-D block, so that the halo planes neighbouring tiles share are requested close together in time:
  // blocks of 8 (d) x 4 (h) x all (w) tiles, d-blocks fastest (-4 % on 96->32 at 128^3 against w-h-d order,
  // which runs one whole 4-plane d-layer at a time).
  int tw_i, th_i, td_i, tn;
  {
    constexpr int BD = 8, BH = 4;
    const int per = a.tiles_d * a.tiles_h * a.tiles_w;
    tn = tile / per;
    int t = tile - tn * per;
    if ((a.tiles_d % BD) == 0 && (a.tiles_h % BH) == 0) {
      const int bs = BD * BH * a.tiles_w, b = t / bs, in = t - b * bs, ndg = a.tiles_d / BD;
      tw_i = in % a.tiles_w;
      th_i = (b / ndg) * BH + (in / a.tiles_w) % BH;
      td_i = (b % ndg) * BD + in / (a.tiles_w * BH);
    } else {
      tw_i = t % a.tiles_w; t /= a.tiles_w;
      th_i = t % a.tiles_h;
      td_i = t / a.tiles_h;
    }
  }
  const int d0 = td_i * TD, h0 = th_i * TH, w0 = tw_i * TW;

  // this lane's source voxel for each of the wave's LDS-DMA instructions (instruction id = i * NW + wave:
  // plane = id & 1, 64-voxel block = id >> 1), -1 = zero padding / beyond the halo
  int gvox[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int id = i * NW + wave, v = (id >> 1) * 64 + lane;
    const int hd = v / (HH * HW), hh = (v / HW) % HH, hw = v % HW;
    const int gd = d0 + hd - a.pd, gh = h0 + hh - a.ph, gw = w0 + hw - a.pw;
    const bool ok = id < 2 * kRuBlocks && v < kRuVox && gd >= 0 && gd < a.di && gh >= 0 && gh < a.hi && gw >= 0 && gw < a.wi;
    gvox[i] = ok ? ((tn * a.di + gd) * a.hi + gh) * a.wi + gw : -1;
  }
  const long long nvox = (long long)a.n * a.di * a.hi * a.wi;
  const auto rs0 = __builtin_amdgcn_make_buffer_rsrc((void*)a.x0, 0, (int)(((nvox - 1) * a.ld0 + a.c0) * 2), 0x00020000);
  const auto rs1 = __builtin_amdgcn_make_buffer_rsrc((void*)(a.c1 ? a.x1 : a.x0), 0,
                                                     a.c1 ? (int)(((nvox - 1) * a.ld1 + a.c1) * 2) : 0, 0x00020000);

  const int lbase = (wave * HH * HW + r) * 16 + h * kRuPlane;
  const char* wlane = a.wp + ((long long)co_base + r) * 32 + h * 16;
  const long long wtap = (long long)a.coutp * 32;

  f32x16 acc[VT][CT];
#pragma unroll
  for (int vt = 0; vt < VT; ++vt)
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[vt][ct][i] = 0.f;

  for (int c = 0; c < a.nchunks; ++c) {
    const int cb = c * 16;
    const bool first = cb < a.c0;
    const int ld2 = (first ? a.ld0 : a.ld1) * 2;
    const int coff = (first ? cb : cb - a.c0) * 2 + (wave & 1) * 16;
    if (c) __syncthreads();                     // the previous chunk's LDS reads are done
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int id = i * NW + wave;
      if (id < 2 * kRuBlocks) {   // wave-uniform
        const int off = gvox[i] >= 0 ? gvox[i] * ld2 + coff : (int)0x80000000;   // out of range -> zeros land in LDS
        lds_ptr dst = (lds_ptr)(smem + (id & 1) * kRuPlane + (id >> 1) * 1024);
        if (first) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs0, dst, 16, off, 0, 0, 0);
        else __builtin_amdgcn_raw_ptr_buffer_load_lds(rs1, dst, 16, off, 0, 0, 0);
      }
    }
    __syncthreads();                            // hipcc drains vmcnt(0) in front of the barrier
    const char* wc = wlane + (long long)c * 27 * wtap;
    // 9 (kd, kw) steps, NOT unrolled (a fully unrolled body makes hipcc hoist every load and spill);
    // the next step's three weight fragments are fetched while this step's 12 MFMAs run
    Frag<T> bn[3][CT];
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) bn[kh][ct].load(wc + (kh * 3) * wtap + ct * 1024);
#pragma unroll 1
    for (int it = 0; it < 9; ++it) {
      const int kd = it / 3, kw = it - kd * 3;
      Frag<T> b[3][CT];
#pragma unroll
      for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) b[kh][ct] = bn[kh][ct];
      if (it < 8) {
        const int nd = (it + 1) / 3, nw = (it + 1) - nd * 3;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
          for (int ct = 0; ct < CT; ++ct) bn[kh][ct].load(wc + (nd * 9 + kh * 3 + nw) * wtap + ct * 1024);
      }
      const char* ap = smem + lbase + (kd * HH * HW + kw) * 16;
#pragma unroll
      for (int hy = 0; hy < TH + 2; ++hy) {
        Frag<T> af;
        af.load(ap + hy * HW * 16);
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
          const int row = hy - kh;
          if (row >= 0 && row < TH) {
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) mma16(af, b[kh][ct], acc[row][ct]);
          }
        }
      }
    }
  }

  TileOut<VT> to;
  to.hstride = (long long)a.os * a.wy * a.ldy;
  to.wstride = (long long)a.os * a.ldy;
#pragma unroll
  for (int vt = 0; vt < VT; ++vt) {
    const int gd = d0 + wave, gh = h0 + vt, gw = w0;
    to.base[vt] = ((((long long)tn * a.dy + (gd * a.os + a.od)) * a.hy + (gh * a.os + a.oh)) * a.wy + (gw * a.os + a.ow)) * a.ldy;
    to.dvalid[vt] = gd < a.do_;
    to.hleft[vt] = a.ho - gh;
    to.wleft[vt] = a.wo - gw;
  }
  conv_epilogue_tile<T, VT, CT, TW, NW>(a, acc, to, co_base, tile, reinterpret_cast<float*>(smem));
}

// ------------------------------------------------------------------------------------------
// deconv_fwd_kernel: ConvTranspose3d(k=2, s=2) forward in bf16 as ONE GEMM with 8*Cout columns (column block =
// output parity class), Cin <= 128.  The gather kernel ran one workgroup per (256 voxels, 64 columns): 8192
// workgroups of 16 MFMAs per wave whose index arithmetic, dependent fragment loads and epilogue set-up cost more
// than the 268 MB they write (95 us with neither A loads nor stores, 153 us in full for 64->64 at 64^3).  Here a
// workgroup owns 256 input voxels for ALL column blocks: the A fragments of its voxels (<= 8 chunks) are loaded
// once and stay in registers, the loop over the 8*Cout/64 column blocks fetches weights, runs 8 * nchunks MFMAs
// and writes one parity class of 64 channels (LDS-transposed 16-byte stores).
// ------------------------------------------------------------------------------------------
template <int MAXC>
__global__ __launch_bounds__(256) void deconv_fwd_kernel(const ConvArgs a) {
  using T = bf16_t;
  constexpr int VT = 2, CT = 2;
  __shared__ float red[4 * CT * 64];
  __shared__ __attribute__((aligned(16))) char tpatch[4 * 2048];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;

  int vn[VT], vd[VT], vh[VT], vw[VT];
  bool vok[VT];
  Frag<T> af[VT][MAXC];
#pragma unroll
  for (int vt = 0; vt < VT; ++vt) {
    long long m = (long long)blockIdx.x * (128 * VT) + (wave * VT + vt) * 32 + r;
    vok[vt] = m < a.m_total;
    const long long vox = vok[vt] ? m : 0;
    if (!vok[vt]) m = 0;
    vw[vt] = (int)(m % a.wo); m /= a.wo;
    vh[vt] = (int)(m % a.ho); m /= a.ho;
    vd[vt] = (int)(m % a.do_);
    vn[vt] = (int)(m / a.do_);
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
      if (c < a.nchunks && vok[vt]) af[vt][c].load(a.x0 + (vox * a.ld0 + c * 16) * 2 + h * 16);
      else af[vt][c].zero();
    }
  }
  const long long wtap = (long long)a.coutp * 32;       // bytes per chunk of the packed [chunk][1 tap][coutp][16] weights
  const int ncb = a.coutp / 64;
  for (int cb = 0; cb < ncb; ++cb) {
    const int co_base = cb * 64;
    const char* wlane = a.wp + ((long long)co_base + r) * 32 + h * 16;
    f32x16 acc[VT][CT];
#pragma unroll
    for (int vt = 0; vt < VT; ++vt)
#pragma unroll
      for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[vt][ct][i] = 0.f;
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
      if (c < a.nchunks) {
        Frag<T> b[CT];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) b[ct].load(wlane + c * wtap + ct * 1024);
#pragma unroll
        for (int vt = 0; vt < VT; ++vt)
#pragma unroll
          for (int ct = 0; ct < CT; ++ct) mma16(af[vt][c], b[ct], acc[vt][ct]);
      }
    }
    const int cls = co_base / a.cls_cout, co_store = co_base - cls * a.cls_cout;
    const int od = cls >> 2, oh = (cls >> 1) & 1, ow = cls & 1;
    long long yoff[VT];
#pragma unroll
    for (int vt = 0; vt < VT; ++vt)
      yoff[vt] = vok[vt] ? ((((long long)vn[vt] * a.dy + (vd[vt] * 2 + od)) * a.hy + (vh[vt] * 2 + oh)) * a.wy + (vw[vt] * 2 + ow)) * a.ldy : -1;
    ConvArgs q = a;
    q.stats = nullptr;
    conv_epilogue<T, VT, CT>(q, acc, yoff, co_base, blockIdx.x, red, co_store, tpatch);
  }
}

// ------------------------------------------------------------------------------------------
// pointwise_conv_kernel: bf16 1x1x1 stride-1 convolution with <= 32 input and <= 32 output channels at full
// resolution (generator head 24->24, final conv 32->6 and its data gradient): pure HBM streams (64 B in, 32-64 B
// out per voxel) that the gather kernel ran at half the achievable bandwidth -- one workgroup per 256 voxels, its
// index arithmetic, weight-fragment loads and dependent A loads exposed per tile.  Here a workgroup walks many
// tiles: the weight fragments stay in registers for the whole kernel and the next tile's A fragments are in
// flight while the current tile's MFMAs and stores run.
// ------------------------------------------------------------------------------------------
// wg_stats: ONE statistics row per workgroup (all its tiles summed in registers) instead of one per tile: at 128^3 the
// 8 192 per-tile rows made the following norm_finalize a 22 us launch (5 us for the rows of any other layer).  Only for
// one sample: a workgroup's tiles stride over the whole tensor, per-sample rows could not be kept apart.
template <int NCH>   // 16-channel chunks of the input (1 or 2)
__global__ __launch_bounds__(256) void pointwise_conv_kernel(const ConvArgs a, int ntiles, int wg_stats) {
  using T = bf16_t;
  constexpr int VT = 2, CT = 1;
  __shared__ float red[4 * CT * 64];
  __shared__ __attribute__((aligned(16))) char tpatch[4 * 2048];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, h = lane >> 5;
  Frag<T> b[NCH];
#pragma unroll
  for (int c = 0; c < NCH; ++c) b[c].load(a.wp + ((long long)c * a.coutp + r) * 32 + h * 16);
  auto load_tile = [&](int tile, Frag<T> (&af)[VT][NCH], long long (&m)[VT]) {
#pragma unroll
    for (int vt = 0; vt < VT; ++vt) {
      m[vt] = (long long)tile * (128 * VT) + (wave * VT + vt) * 32 + r;
      const bool ok = m[vt] < a.m_total;
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        if (ok) af[vt][c].load(a.x0 + (m[vt] * a.ld0 + c * 16) * 2 + h * 16);
        else af[vt][c].zero();
      }
      if (!ok) m[vt] = -1;
    }
  };
  Frag<T> cur[VT][NCH], nxt[VT][NCH];
  long long mc[VT], mn[VT];
  float ws1 = 0.f, ws2 = 0.f;                               // this lane's column (channel r), rows of its lane half
  ConvArgs an = a;
  if (wg_stats) an.stats = nullptr;
  int tile = blockIdx.x;
  if (tile < ntiles) load_tile(tile, cur, mc);
  for (; tile < ntiles; tile += gridDim.x) {
    const int tn = tile + gridDim.x;
    if (tn < ntiles) load_tile(tn, nxt, mn);              // in flight under this tile's MFMAs and stores
    f32x16 acc[VT][CT];
#pragma unroll
    for (int vt = 0; vt < VT; ++vt) {
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[vt][0][i] = 0.f;
#pragma unroll
      for (int c = 0; c < NCH; ++c) mma16(cur[vt][c], b[c], acc[vt][0]);
    }
    long long yoff[VT];
#pragma unroll
    for (int vt = 0; vt < VT; ++vt) yoff[vt] = mc[vt] >= 0 ? mc[vt] * a.ldy : -1;
    if (wg_stats && a.stats) {
#pragma unroll
      for (int vt = 0; vt < VT; ++vt) {
        const long long m0 = (long long)tile * (128 * VT) + (wave * VT + vt) * 32;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const float v = (m0 + acc_row(i, h) < a.m_total) ? acc[vt][0][i] : 0.f;
          ws1 += v;
          ws2 += v * v;
        }
      }
    }
    conv_epilogue<T, VT, CT>(an, acc, yoff, 0, tile, red, 0, tpatch);
#pragma unroll
    for (int vt = 0; vt < VT; ++vt) {
      mc[vt] = mn[vt];
#pragma unroll
      for (int c = 0; c < NCH; ++c) cur[vt][c] = nxt[vt][c];
    }
  }
  if (wg_stats && a.stats) {
    // lane halves, then the 4 waves in a fixed order (deterministic), as conv_epilogue does per tile
    ws1 += __shfl_xor(ws1, 32, 64);
    ws2 += __shfl_xor(ws2, 32, 64);
    __syncthreads();
    if (h == 0) { red[(wave * 2 + 0) * 32 + r] = ws1; red[(wave * 2 + 1) * 32 + r] = ws2; }
    __syncthreads();
    if (wave == 0 && h == 0) {
      float t1 = 0.f, t2 = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w) { t1 += red[(w * 2 + 0) * 32 + r]; t2 += red[(w * 2 + 1) * 32 + r]; }
      float* p = a.stats + ((long long)blockIdx.x * 2) * a.coutp;
      p[r] = t1;
      p[a.coutp + r] = t2;
    }
  }
}
