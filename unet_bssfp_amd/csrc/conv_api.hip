// Host dispatch of the implicit-GEMM convolution kernels (C ABI: mi355_conv_fwd).
#include <stdlib.h>
#include "conv_kernels.h"
#include "conv_march.h"
#include "conv_marchg.h"
#include "conv_march2.h"

namespace {

struct Plan {
  bool halo;
  int shape;      // halo tile: index into kTD/kTH/kTW/kVT
  int vt, ct;
  int tiles_d, tiles_h, tiles_w;
  long long tiles;
  int tiles_per_sample;
  int seg_len, nseg;          // marching kernel (shape 10): output planes per workgroup segment, segments per sample
  int ksplit, rpb;            // split-K factor (1 = off) and rows per reduce block
  long long stat_rows;        // rows of stats_part ( = tiles, or reduce blocks under split-K )
  bool pointwise, wg_stats;   // persistent 1x1x1 kernel; its statistics as one row per workgroup
  int stat_rows_per_sample;
};

int gcd_i(long long a, long long b) { while (b) { long long t = a % b; a = b; b = t; } return (int)a; }

// 0 wide (2x4x32, 2 subtiles/wave)  1 mid (2x8x16, 2)  2 small (4x8x8, 2)
// 3 wide4 (4x4x32, 4: bf16 thin-Cout layers)  4 mid1 (2x4x16, 1)  5 small1 (2x8x8, 1): more workgroups at the low levels
// 6 wide8 (4x4x32, 8 waves x 2 subtiles: 25 % less halo traffic than 2x4x32 at the same occupancy)
// 7, 8: retired experiments (row reuse inside conv_halo_kernel with register staging: 216 VGPRs or spills, slower)
// 9 ru (4x4x32, 4 waves, row-reuse loop, halo by LDS-DMA: conv_ru_kernel)
// 10 march (16x32 footprint marching along d, 32 input channels resident: conv_march_kernel; tile extents set in make_plan)
// 11 marchg (4 ROWS x 32 footprint marching along d, input channels in 32-channel groups, weights streamed through an LDS
//    ring: conv_marchg_kernel<ROWS>; vt = ROWS)
// 12 / 13 lowg (512-voxel tiles 4x8x16 / 8x8x8 x 64 output channels, weights through LDS once per workgroup: conv_lowg_kernel)
// 14 march2 (dense 2x2x2 on wide tensors -- the PatchGAN on space-to-depth operands: conv_march2_kernel<ROWS>; vt = ROWS)
const int kTD[14] = {2, 2, 4, 4, 2, 2, 4, 8, 4, 4, 0, 0, 4, 8}, kTH[14] = {4, 8, 8, 4, 4, 8, 4, 4, 4, 4, 0, 0, 8, 8},
          kTW[14] = {32, 16, 8, 32, 16, 8, 32, 32, 32, 32, 0, 0, 16, 8}, kVT[14] = {2, 2, 2, 4, 1, 1, 2, 4, 4, 4, 0, 0, 4, 4};

// Plan overrides for A/B runs exist only in the diagnostic build (-DMI355_DIAG, built by tools/build_diag.sh into
// tools/_build/, never shipped): MI355_CONV_SHAPE=<0|6|9|10>, MI355_CONV_CT=<1|2>, MI355_CONV_KSPLIT=<n>.
#ifdef MI355_DIAG
int env_int(const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; }
int forced_ct() { static const int v = env_int("MI355_CONV_CT", 0); return v; }
int forced_ksplit() { static const int v = env_int("MI355_CONV_KSPLIT", 0); return v; }
int forced_shape() { static const int v = env_int("MI355_CONV_SHAPE", -1); return v; }
// planner constants (sweeps: tools/sweep_plan.sh)
int tune_mg_minwg() { static const int v = env_int("MI355_MG_MINWG", 192); return v; }
int tune_mg_fix() { static const int v = env_int("MI355_MG_FIX", 900); return v; }
int tune_march_minwg() { static const int v = env_int("MI355_MARCH_MINWG", 128); return v; }
int tune_ks_target() { static const int v = env_int("MI355_KS_TARGET", 1024); return v; }
int tune_low_min() { static const int v = env_int("MI355_LOW_MIN", 512); return v; }
int tune_lowg() { static const int v = env_int("MI355_LOWG", 1); return v; }          // 0: the low levels stay on conv_halo_kernel
int tune_lowg_target() { static const int v = env_int("MI355_LOWG_TARGET", 256); return v; }
int tune_lowg_minch() { static const int v = env_int("MI355_LOWG_MINCH", 4); return v; }     // least 16-channel chunks
int tune_lowg_maxw() { static const int v = env_int("MI355_LOWG_MAXW", 20); return v; }      // widest row the low-level plans take
int tune_gather_split() { static const int v = env_int("MI355_GATHER_SPLIT", 1); return v; }
#else
constexpr int forced_ct() { return 0; }
constexpr int forced_ksplit() { return 0; }
constexpr int forced_shape() { return -1; }
constexpr int tune_mg_minwg() { return 192; }
constexpr int tune_mg_fix() { return 900; }
constexpr int tune_march_minwg() { return 128; }
constexpr int tune_ks_target() { return 1024; }
constexpr int tune_low_min() { return 512; }
constexpr int tune_lowg() { return 1; }
constexpr int tune_lowg_target() { return 256; }
constexpr int tune_lowg_minch() { return 4; }
// (20: the 20^3 level of a 160^3 volume -- BASELINE configs[4] -- is 2.56x its own work in conv_marchg_kernel's 16 x 32 footprints;
//  1x24x160^3 step, interleaved A/B of the diagnostic build: 16 -> 18.45 ms, 20 -> 17.90, 40 -> 18.03: profiles/r04c_ab_160_lowg_maxw.txt)
constexpr int tune_lowg_maxw() { return 20; }
constexpr int tune_gather_split() { return 1; }
#endif

int make_plan(const mi355_conv_desc* d, Plan* p) {
  MI355_REQUIRE(d && d->x0 && d->wp && d->y, "conv: null pointer");
  MI355_REQUIRE(d->dtype == MI355_DT_F32 || d->dtype == MI355_DT_BF16 || d->dtype == MI355_DT_FP8, "conv: bad dtype %d", d->dtype);
  MI355_REQUIRE(d->dtype != MI355_DT_FP8 || (d->q_amax_x && d->q_amax_w), "conv: fp8 operands need q_amax_x / q_amax_w");
  MI355_REQUIRE(d->c0 > 0 && d->c0 % 16 == 0 && d->c1 >= 0 && d->c1 % 16 == 0, "conv: channels must be multiples of 16 (c0=%d c1=%d)", d->c0, d->c1);
  MI355_REQUIRE(d->c1 == 0 || d->x1, "conv: c1 > 0 without x1");
  MI355_REQUIRE(d->ld0 >= d->c0 && (d->c1 == 0 || d->ld1 >= d->c1), "conv: ld < channels");
  {
    const int epv = d->dtype == MI355_DT_F32 ? 4 : (d->dtype == MI355_DT_FP8 ? 16 : 8);
    MI355_REQUIRE(d->ld0 % epv == 0 && (d->c1 == 0 || d->ld1 % epv == 0), "conv: ld must keep rows 16-byte aligned");
  }
  MI355_REQUIRE(d->coutp > 0 && d->coutp % 32 == 0, "conv: coutp %% 32 != 0");
  MI355_REQUIRE(d->cstore > 0 && d->cstore <= d->coutp && d->ldy >= d->cstore, "conv: bad cstore/ldy");
  MI355_REQUIRE(d->ks >= 1 && d->ks <= 4 && d->stride >= 1 && d->stride <= 2, "conv: unsupported ks=%d stride=%d", d->ks, d->stride);
  MI355_REQUIRE(d->n > 0 && d->di > 0 && d->hi > 0 && d->wi > 0 && d->do_ > 0 && d->ho > 0 && d->wo > 0, "conv: empty extent");
  MI355_REQUIRE(d->os >= 1, "conv: os < 1");
  MI355_REQUIRE(d->cls_cout == 0 || (d->ks == 1 && d->stride == 1 && d->os == 2 && d->cls_cout % 64 == 0 &&
                                     d->coutp == 8 * d->cls_cout && d->cstore <= d->cls_cout && !d->stats_part &&
                                     d->ooff[0] == 0 && d->ooff[1] == 0 && d->ooff[2] == 0),
                "conv: bad transposed-conv class folding (cls_cout=%d)", d->cls_cout);
  {
    const int hi_off = d->cls_cout ? 1 : 0;     // classes reach offset 1 in every dimension
    MI355_REQUIRE((d->do_ - 1) * d->os + d->ooff[0] + hi_off < d->dy && (d->ho - 1) * d->os + d->ooff[1] + hi_off < d->hy &&
                      (d->wo - 1) * d->os + d->ooff[2] + hi_off < d->wy && d->ooff[0] >= 0 && d->ooff[1] >= 0 && d->ooff[2] >= 0,
                  "conv: output grid exceeds the output tensor");
  }
  p->ct = (d->coutp % 64 == 0) ? 2 : 1;
  p->seg_len = p->nseg = 0;
  p->halo = ((d->ks == 3 || d->ks == 2) && d->stride == 1);
  if (p->halo) {
    if (forced_ct() == 1) p->ct = 1;
    // low levels in bf16: 512-voxel tiles x 64 output channels, weights through LDS once per workgroup (conv_lowg_kernel)
    const bool lowg_ok = tune_lowg() && d->dtype == MI355_DT_BF16 && d->ks == 3 && d->coutp % 64 == 0 &&
                         (d->c0 + d->c1) >= 16 * tune_lowg_minch() && (long long)d->n * d->do_ * d->ho * d->wo >= 256 &&
                         d->wo <= tune_lowg_maxw();
    p->shape = (d->wo > 16 && !lowg_ok) ? 0 : (d->wo > 8 ? 1 : 2);       // (0: the wide-level plans below)
    // dense k2 on wide bf16 tensors in whole 32-channel groups, plain output grid, padding 0 (forward on S(a)) or 1 (its data
    // gradient): the marching k2 kernel when its footprints x d-segments fill (most of) the chip and the rows are not mostly
    // tile padding.  Cost per workgroup and 32-channel group: (len + 1) input planes of fixed overhead + len output planes
    // of 2 x 4 x 2 ROWS MFMAs.
    bool march2 = false;
    if (d->d2s) {
      const long long nvi = (long long)d->n * d->di * d->hi * d->wi, nvo = (long long)d->n * d->dy * d->hy * d->wy;
      MI355_REQUIRE(d->dtype == MI355_DT_BF16 && d->ks == 2 && d->c0 % 32 == 0 && d->c1 == 0 && d->os == 1 && d->coutp % 256 == 0 &&
                        d->ooff[0] == 0 && d->ooff[1] == 0 && d->ooff[2] == 0 && d->pad[0] == 0 && d->pad[1] == 0 && d->pad[2] == 0 &&
                        d->do_ == d->di && d->ho == d->hi && d->wo == d->wi && d->dy == 2 * d->do_ && d->hy == 2 * d->ho && d->wy == 2 * d->wo &&
                        d->cstore <= d->coutp / 8 && (d->cstore & 7) == 0 && !d->y_f32 && d->add_n >= 0,
                    "conv: bad depth-to-space descriptor");
      MI355_REQUIRE(nvi * d->ld0 * 2 < (1ll << 31) && nvo * d->ldy * 2 < (1ll << 31) && (!d->addend || nvo * d->ld_add * (d->add_bf16 ? 2 : 4) < (1ll << 31)),
                    "conv: depth-to-space tensors exceed 32-bit byte offsets");
      // 8-row footprints, two workgroups per CU (512 per round)
      long long best = -1; int best_len = 0;
      const int best_rows = 2;
      {
        const int th = ceil_div(d->ho, 8), tw = ceil_div(d->wo, 32);
        const long long fp = (long long)d->n * th * tw * (d->coutp / 32);
        for (int ns = 1; ns <= d->do_ && ns <= 64; ++ns) {
          const int len = ceil_div(d->do_, ns), segs = ceil_div(d->do_, len);
          if (fp * segs < 256 && ns < d->do_ && ns < 64) continue;
          const long long rounds = (fp * segs + 511) / 512, cost = rounds * ((len + 1) * 500ll + (long long)len * 1024);
          if (best < 0 || cost < best) { best = cost; best_len = len; }
        }
      }
      MI355_REQUIRE(best >= 0, "conv: no depth-to-space plan");
      march2 = true;
      p->shape = 14;
      p->seg_len = best_len;
      p->nseg = ceil_div(d->do_, best_len);
      p->vt = best_rows;
    } else {
      const int pd = d->pad[0];
      const long long nvi = (long long)d->n * d->di * d->hi * d->wi, nvo = (long long)d->n * d->dy * d->hy * d->wy;
      const bool ok = d->dtype == MI355_DT_BF16 && d->ks == 2 && d->c0 % 32 == 0 && d->c1 % 32 == 0 && d->os == 1 &&
                      d->ooff[0] == 0 && d->ooff[1] == 0 && d->ooff[2] == 0 && (pd == 0 || pd == 1) && d->pad[1] == pd && d->pad[2] == pd &&
                      d->do_ == d->di - 1 + 2 * pd && d->ho == d->hi - 1 + 2 * pd && d->wo == d->wi - 1 + 2 * pd &&
                      d->dy == d->do_ && d->hy == d->ho && d->wy == d->wo && (d->cstore & 7) == 0 && d->wo >= 32 &&
                      nvi * d->ld0 * 2 < (1ll << 31) && nvi * (d->c1 ? d->ld1 : 0) * 2 < (1ll << 31) &&
                      nvo * d->ldy * (d->y_f32 ? 4 : 2) < (1ll << 31) && (!d->addend || nvo * d->ld_add * 4 < (1ll << 31)) && d->add_n >= 0 &&
                      forced_shape() != 0;
      if (ok) {
        // 8-row footprints, two workgroups per CU (512 per round); cost per workgroup and 32-channel group: (len + 1) input planes
        // of fixed overhead + len output planes of 32 MFMAs per wave
        long long best = -1; int best_len = 0;
        const int best_rows = 2;
        {
          const int th = ceil_div(d->ho, 8), tw = ceil_div(d->wo, 32);
          // (rows x columns the tiles cover against the ones that exist: S-layout gradients are 2^k + 1 wide)
          const bool tiles_ok = (long long)th * 8 * tw * 32 * 2 <= 3ll * d->ho * d->wo;
          const long long fp = (long long)d->n * th * tw * (d->coutp / 32);
          for (int ns = 1; tiles_ok && ns <= d->do_ && ns <= 64; ++ns) {
            const int len = ceil_div(d->do_, ns), segs = ceil_div(d->do_, len);
            if (fp * segs < 128) continue;
            const long long rounds = (fp * segs + 511) / 512, cost = rounds * ((len + 1) * 500ll + (long long)len * 1024);
            if (best < 0 || cost < best) { best = cost; best_len = len; }
          }
        }
        if (best >= 0) {
          march2 = true;
          p->shape = 14;
          p->seg_len = best_len;
          p->nseg = ceil_div(d->do_, best_len);
          p->vt = best_rows;
        }
      }
    }
    MI355_REQUIRE(march2 || !d->addend, "conv: addend needs the marching k2 plan (bf16, ks 2, 32-channel groups, wide rows)");
    MI355_REQUIRE(d->d2s || (!d->delta && !d->add_bf16), "conv: delta / add_bf16 belong to the depth-to-space mode");
    auto count = [&](int sh, int ct) {
      return (long long)ceil_div(d->do_, kTD[sh]) * ceil_div(d->ho, kTH[sh]) * ceil_div(d->wo, kTW[sh]) * d->n *
             (d->coutp / (32 * ct));
    };
    if (march2) {
    } else if (p->shape == 0) {
      if ((d->dtype == MI355_DT_BF16 || d->dtype == MI355_DT_FP8) && d->ks == 3) {
        // wide bf16 3x3x3 layers: the row-reuse + LDS-DMA kernel (shape 9) when its 4x4x32 tiles fill the chip,
        // else 8-wave 4x4x32 tiles (shape 6) / the plain 2x4x32 tile.
        const int f = forced_shape();
        const long long nv = (long long)d->n * d->di * d->hi * d->wi;
        const bool ru_ok = nv * d->ld0 * 2 < (1ll << 31) && nv * (d->c1 ? d->ld1 : 0) * 2 < (1ll << 31) &&
                           (long long)d->n * d->dy * d->hy * d->wy * d->ldy * 2 < (1ll << 31);                       // 32-bit byte offsets
        const bool big = count(6, p->ct) >= 1024;
        const long long c9 = count(9, p->ct);
        int pick = (p->ct == 1 ? c9 >= 1024 : c9 >= 512) ? 9 : (big ? 6 : 0);
        if (f == 0) pick = 0;
        else if (f == 6) pick = big ? 6 : 0;
        else if (f == 9) pick = 9;
        if (pick == 9 && !ru_ok) pick = big ? 6 : 0;
        // <= 32 input channels in one source, plain output grid: the marching kernel, when its footprints x d-segments
        // fill at least half the chip
        const bool march_ok = ru_ok && d->c1 == 0 && d->c0 == 32 && d->os == 1 && d->ooff[0] == 0 && d->ooff[1] == 0 && d->ooff[2] == 0 &&
                              d->pad[0] == 1 && d->pad[1] == 1 && d->pad[2] == 1 && d->do_ == d->di && d->ho == d->hi && d->wo == d->wi &&
                              d->dy == d->do_ && d->hy == d->ho && d->wy == d->wo && (d->cstore & 7) == 0;
        if (march_ok && (f < 0 || f == 10)) {
          const long long fp = (long long)d->n * ceil_div(d->ho, kMarchFH) * ceil_div(d->wo, kMarchFW) * (d->coutp / 32);
          // segment length L: the grid should be whole 256-workgroup rounds, each workgroup marches L + 2 input planes;
          // L = 2 (mod 3), L >= 5 takes the kernel's straight-line path (conv_march.h), so only such L are proposed
          // unless the volume is too shallow
          long long best = -1; int best_len = 0;
          for (int ns = 1; ns <= d->do_ && ns <= 64; ++ns) {
            int len = ceil_div(d->do_, ns);
            if (d->do_ >= 5) { if (len < 5) len = 5; len += (2 - len % 3 + 3) % 3; }
            const int segs = ceil_div(d->do_, len);
            const long long rounds = (fp * segs + 255) / 256, cost = rounds * (len + 2);
            if (best < 0 || cost < best) { best = cost; best_len = len; }
          }
          const int segs = ceil_div(d->do_, best_len);
          if (fp * segs >= tune_march_minwg() || f == 10 || d->dtype == MI355_DT_FP8) {
            pick = 10;
            p->seg_len = best_len;
            p->nseg = segs;
          }
        }
        // more than 32 input channels (whole 32-channel groups per source), plain output grid: the group-marching kernel.
        // Cost model per workgroup: (len + 2) input planes of fixed overhead (three block hand-overs per 32-channel group,
        // ~900 cycles) + len output planes of MFMA work (54 x 32 cycles per footprint row); ROWS = 4 unless only the
        // 8-row footprints can fill the chip.
        const bool marchg_ok = ru_ok && d->dtype == MI355_DT_BF16 && d->c0 % 32 == 0 && d->c1 % 32 == 0 && d->c0 + d->c1 > 32 &&
                               d->os == 1 && d->ooff[0] == 0 && d->ooff[1] == 0 && d->ooff[2] == 0 &&
                               d->pad[0] == 1 && d->pad[1] == 1 && d->pad[2] == 1 && d->do_ == d->di && d->ho == d->hi && d->wo == d->wi &&
                               d->dy == d->do_ && d->hy == d->ho && d->wy == d->wo && (d->cstore & 7) == 0;
        if (marchg_ok && (f < 0 || f == 11) && pick != 10) {
          long long best = -1; int best_len = 0, best_rows = 0;
          for (int rows = 4; rows >= 2; rows -= 2) {
            const long long fp = (long long)d->n * ceil_div(d->ho, 4 * rows) * ceil_div(d->wo, 32) * (d->coutp / 32);
            for (int ns = 1; ns <= d->do_ && ns <= 64; ++ns) {
              const int len = ceil_div(d->do_, ns), segs = ceil_div(d->do_, len);
              if (fp * segs < tune_mg_minwg() && f != 11) continue;     // must fill (most of) the chip
              const long long rounds = (fp * segs + 255) / 256, cost = rounds * ((len + 2) * (long long)tune_mg_fix() + (long long)len * 1728 * rows);
              if (best < 0 || cost < best) { best = cost; best_len = len; best_rows = rows; }
            }
            if (best >= 0) break;
          }
          if (best >= 0) {
            pick = 11;
            p->seg_len = best_len;
            p->nseg = ceil_div(d->do_, best_len);
            p->vt = best_rows;
          }
        }
        p->shape = pick;
      }
      // 32^3-level layers on the plain tile: one 64-channel tile per workgroup leaves <= 1 workgroup per CU on a
      // long K loop; 32-channel tiles double the workgroups (measured 128->64: 40 -> 30 us, 128->128: 48 -> 41 us)
      if (p->shape == 0 && p->ct == 2 && count(0, 2) < 512 && forced_ct() != 2) p->ct = 1;
      // ... and when even that leaves <= 2 workgroups per CU, half-width tiles (2x4x16, one subtile per wave) double
      // them again: -0.1 ms per step in the interleaved A/B (256->128 at 32^3: 98 -> 91 us, 128->64: 31 -> 26 us)
      if (p->shape == 0 && p->ct == 1 && d->ks == 3 && count(0, 1) <= 512) p->shape = 4;
    } else if (lowg_ok) {
      p->shape = d->wo > 8 ? 12 : 13;
      p->ct = 2;
    } else {
      // low levels: few tiles -> favour more, smaller workgroups (the K loop is long, the grid is not)
      if (count(p->shape, p->ct) < tune_low_min()) p->ct = 1;
      if (count(p->shape, p->ct) < tune_low_min()) p->shape += 3;
    }
    if (p->shape == 10) {
      p->vt = 4; p->ct = 1;
      p->tiles_d = p->nseg;
      p->tiles_h = ceil_div(d->ho, kMarchFH);
      p->tiles_w = ceil_div(d->wo, kMarchFW);
    } else if (p->shape == 11 || p->shape == 14) {
      p->ct = 1;                                     // (vt = ROWS was set with the plan)
      p->tiles_d = p->nseg;
      p->tiles_h = ceil_div(d->ho, 4 * p->vt);
      p->tiles_w = ceil_div(d->wo, 32);
    } else {
    p->vt = kVT[p->shape];
    p->tiles_d = ceil_div(d->do_, kTD[p->shape]);
    p->tiles_h = ceil_div(d->ho, kTH[p->shape]);
    p->tiles_w = ceil_div(d->wo, kTW[p->shape]);
    }
    p->tiles_per_sample = p->tiles_d * p->tiles_h * p->tiles_w;
    p->tiles = (long long)p->tiles_per_sample * d->n;
  } else {
    const long long per = (long long)d->do_ * d->ho * d->wo;
    const long long m = per * d->n;
    p->vt = (m >= 128 * 2 * 512) ? 2 : 1;
    const int wg = 128 * p->vt;
    p->tiles = (m + wg - 1) / wg;
    p->tiles_per_sample = (per % wg == 0) ? (int)(per / wg) : 0;
    p->tiles_d = p->tiles_h = p->tiles_w = 0;
    p->shape = 0;
  }
  MI355_REQUIRE(p->halo || (!d->addend && !d->y_f32 && !d->d2s), "conv: addend / y_f32 / d2s need the marching k2 plan");
  MI355_REQUIRE(!d->y_f32 || (p->halo && p->shape == 14), "conv: y_f32 is implemented by the marching k2 kernel");
  MI355_REQUIRE(d->dtype != MI355_DT_FP8 || (p->halo && p->shape == 10),
                "conv: the fp8 path covers 3x3x3 stride-1 layers with 32 input channels in one source and a plain output grid");
  MI355_REQUIRE(p->tiles < (1ll << 31), "conv: too many tiles");
  p->ksplit = 1; p->rpb = 0;
  p->stat_rows = p->tiles; p->stat_rows_per_sample = p->tiles_per_sample;
  if (d->d2s) { p->stat_rows *= 8; p->stat_rows_per_sample *= 8; }     // one statistics row per (tile, output class)
  p->pointwise = !p->halo && d->dtype == MI355_DT_BF16 && !d->cls_cout && d->ks == 1 && d->stride == 1 && d->os == 1 && p->vt == 2 &&
                 (d->c0 + d->c1) / 16 <= 2 && d->c1 == 0 && d->coutp == 32 && d->ooff[0] == 0 && d->ooff[1] == 0 && d->ooff[2] == 0 &&
                 d->dy == d->do_ && d->hy == d->ho && d->wy == d->wo && d->pad[0] == 0 && d->pad[1] == 0 && d->pad[2] == 0;
  p->wg_stats = p->pointwise && d->n == 1 && p->tiles > 2048;       // one statistics row per (persistent) workgroup
  if (p->wg_stats) { p->stat_rows = 2048; p->stat_rows_per_sample = 2048; }
  auto set_split = [&](long long ks) {
    p->ksplit = (int)ks;
    const long long per = (long long)d->do_ * d->ho * d->wo;
    // rows per reduce block: divides the per-sample position count (statistics groups) and leaves >= ~512 blocks
    const long long cblocks = (d->coutp + 1023) / 1024;
    int rpb = 64;
    while (rpb > 1 && (per % rpb != 0 || (per / rpb) * d->n * cblocks < 512)) rpb >>= 1;
    p->rpb = gcd_i(per, rpb);
    p->stat_rows_per_sample = (int)(per / p->rpb);
    p->stat_rows = (long long)p->stat_rows_per_sample * d->n;
  };
  if (!p->halo && !d->cls_cout && d->os == 1 && d->ooff[0] == 0 && d->ooff[1] == 0 && d->ooff[2] == 0 && tune_gather_split()) {
    // gather kernel with few output positions and a long contraction (the transposed convolutions' data gradients at the
    // 16^3 / 8^3 levels: 128 dependent (tap, chunk) steps in 32 workgroups, 64 us for 1 GFLOP): split the (tap, chunk) pairs
    const long long wgs = p->tiles * (d->coutp / (32 * p->ct));
    const long long nit = (long long)d->ks * d->ks * d->ks * ((d->c0 + d->c1) / 16);
    if (wgs <= 128 && nit >= 32) {
      long long ks = (256 + wgs - 1) / wgs;
      if (ks > nit / 8) ks = nit / 8;
      if (ks > 32) ks = 32;
      if (ks >= 2) set_split(ks);
    }
  }
  if (p->halo && p->shape != 10 && p->shape != 11 && p->shape != 14) {      // (the marching kernels walk the whole contraction themselves)
    // few output positions and a long contraction (8^3 / 16^3 U-Net levels, low PatchGAN levels): the
    // grid cannot fill 256 CUs and every workgroup streams its weights at one L2/HBM latency per tap
    // group => split the contraction over blockIdx.z and combine in a second kernel
    const long long wgs = p->tiles * (d->coutp / (32 * p->ct));
    const int nchunks = (d->c0 + d->c1) / 16;
    const int fk = forced_ksplit();
    const bool lowg = p->shape == 12 || p->shape == 13;       // one workgroup per CU: aim at 256 of them, down to one chunk each
    if ((fk == 0 && (lowg ? wgs < tune_lowg_target() : ((wgs < 256 && nchunks >= 8) || (wgs <= 512 && nchunks >= 16)))) || fk > 1) {   // (32^3 x 128 ch measured slower split)
      // low-level plan: the LARGEST split that still fits one round of tune_lowg_target() workgroups (rounded up, 60 workgroups -- the
      // 20^3 level of a 160^3 volume -- became 5 x 60 = 300: a second round for 44 of them; 24 at 10^3 became 264).  Same splits as
      // before wherever the count divides 256 (every low level of a 128^3 or 64^3 volume); 160^3: 17.99 -> 17.67 ms with the target
      // swept to the same effect (profiles/r04c_ab_lowg_target.txt)
      long long ks = lowg ? std::max(1ll, (long long)tune_lowg_target() / wgs) : (tune_ks_target() + wgs - 1) / wgs;
      if (fk > 1) ks = fk;
      if (ks > (lowg ? nchunks : nchunks / 2)) ks = lowg ? nchunks : nchunks / 2;
      if (ks > 32) ks = 32;
      if (ks >= 2) set_split(ks);
    }
  }
  return MI355_OK;
}

// A kernel that declares more than 64 KB of dynamic LDS needs its limit raised once per process; 0 on success.  A failure
// (another driver or LDS carve-out) is reported by name at the launch site instead of as a generic launch error later.
static int raise_lds(const void* fn, int bytes) {
  return (int)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
}

template <typename T>
int launch(const mi355_conv_desc* d, const Plan& p, hipStream_t st) {
  ConvArgs a;
  a.x0 = (const char*)d->x0; a.x1 = (const char*)d->x1;
  a.c0 = d->c0; a.c1 = d->c1; a.ld0 = d->ld0; a.ld1 = d->ld1;
  a.n = d->n; a.di = d->di; a.hi = d->hi; a.wi = d->wi;
  a.do_ = d->do_; a.ho = d->ho; a.wo = d->wo;
  a.ks = d->ks; a.stride = d->stride;
  a.pd = d->pad[0]; a.ph = d->pad[1]; a.pw = d->pad[2];
  a.wp = (const char*)d->wp; a.coutp = d->coutp; a.bias = d->bias;
  a.y = (char*)d->y; a.ldy = d->ldy; a.cstore = d->cstore;
  a.dy = d->dy; a.hy = d->hy; a.wy = d->wy; a.os = d->os;
  a.od = d->ooff[0]; a.oh = d->ooff[1]; a.ow = d->ooff[2];
  a.stats = d->stats_part;
  a.tiles_d = p.tiles_d; a.tiles_h = p.tiles_h; a.tiles_w = p.tiles_w;
  a.nchunks = (d->c0 + d->c1) / 16;
  a.m_total = (long long)d->n * d->do_ * d->ho * d->wo;
  a.ksplit = p.ksplit;
  a.cls_cout = d->cls_cout;
  a.nbias = d->nbias > 0 ? d->nbias : d->coutp;
  a.kslab = (float*)d->workspace;
  if (p.ksplit > 1) {
    const long long need = (long long)p.ksplit * a.m_total * d->coutp * 4;
    MI355_REQUIRE(d->workspace && d->workspace_bytes >= need, "conv: split-K workspace too small (%lld < %lld)",
                  (long long)d->workspace_bytes, need);
  }
  dim3 grid((unsigned)p.tiles, (unsigned)(d->coutp / (32 * p.ct)), (unsigned)p.ksplit);
  dim3 block(p.halo && p.shape == 6 ? 512 : 256);
#define HALO(KS, TD, TH, TW, CT)                                     \
  do {                                                               \
    constexpr int lds = conv_halo_lds<T, KS, TD, TH, TW, CT, 4>();   \
    conv_halo_kernel<T, KS, TD, TH, TW, CT><<<grid, block, lds, st>>>(a); \
  } while (0)
#define HALO_KS(KS)                                                  \
    switch (p.shape * 10 + p.ct) {                                   \
      case 1: HALO(KS, 2, 4, 32, 1); break;                          \
      case 2: HALO(KS, 2, 4, 32, 2); break;                          \
      case 11: HALO(KS, 2, 8, 16, 1); break;                         \
      case 12: HALO(KS, 2, 8, 16, 2); break;                         \
      case 21: HALO(KS, 4, 8, 8, 1); break;                          \
      case 22: HALO(KS, 4, 8, 8, 2); break;                          \
      case 41: HALO(KS, 2, 4, 16, 1); break;                         \
      case 42: HALO(KS, 2, 4, 16, 2); break;                         \
      case 51: HALO(KS, 2, 8, 8, 1); break;                          \
      case 52: HALO(KS, 2, 8, 8, 2); break;                          \
      default: mi355_set_error("conv: no kernel for plan %d/%d", p.shape, p.ct); return MI355_ERR_UNSUPPORTED; \
    }
  if (p.halo) {
    if (d->ks == 3) {
      if (p.shape == 3) { if constexpr (sizeof(T) == 2) { HALO(3, 4, 4, 32, 1); } }
      else if (p.shape == 6) {
        if constexpr (sizeof(T) == 2) {
          constexpr int lds = conv_halo_lds<T, 3, 4, 4, 32, 2, 8>();
          if (p.ct == 2) conv_halo_kernel<T, 3, 4, 4, 32, 2, 8><<<grid, block, lds, st>>>(a);
          else conv_halo_kernel<T, 3, 4, 4, 32, 1, 8><<<grid, block, lds, st>>>(a);
        }
      } else if (p.shape == 10) {
        if constexpr (sizeof(T) == 2) {
          static const int once = [] {
            return raise_lds((const void*)conv_march_kernel<false>, MarchCfg<false>::LDS) | raise_lds((const void*)conv_march_kernel<true>, MarchCfg<true>::LDS);
          }();
          if (once) { mi355_set_error("conv_march: cannot raise the dynamic LDS limit to %d bytes (hip error %d)", MarchCfg<false>::LDS, once); return MI355_ERR_HIP; }
          MarchArgs m{p.seg_len, p.nseg, p.tiles_h, p.tiles_w, d->q_amax_x, d->q_amax_w};
          if (d->dtype == MI355_DT_FP8) conv_march_kernel<true><<<grid, block, MarchCfg<true>::LDS, st>>>(a, m);
          else conv_march_kernel<false><<<grid, block, MarchCfg<false>::LDS, st>>>(a, m);
        }
      } else if (p.shape == 11) {
        if constexpr (sizeof(T) == 2) {
          static const int once = [] {
            return raise_lds((const void*)conv_marchg_kernel<4>, MarchGCfg<4>::LDS) | raise_lds((const void*)conv_marchg_kernel<2>, MarchGCfg<2>::LDS);
          }();
          if (once) { mi355_set_error("conv_marchg: cannot raise the dynamic LDS limit to %d bytes (hip error %d)", MarchGCfg<4>::LDS, once); return MI355_ERR_HIP; }
          MarchArgs m{p.seg_len, p.nseg, p.tiles_h, p.tiles_w, nullptr, nullptr};
          if (p.vt == 4) conv_marchg_kernel<4><<<grid, block, MarchGCfg<4>::LDS, st>>>(a, m);
          else conv_marchg_kernel<2><<<grid, block, MarchGCfg<2>::LDS, st>>>(a, m);
        }
      } else if (p.shape == 9) {
        if constexpr (sizeof(T) == 2) {
          if (p.ct == 2) conv_ru_kernel<2><<<grid, block, kRuLds, st>>>(a);
          else conv_ru_kernel<1><<<grid, block, kRuLds, st>>>(a);
        }
      } else if (p.shape == 12 || p.shape == 13) {
        if constexpr (sizeof(T) == 2) {
          static const int once = [] {
            return raise_lds((const void*)conv_lowg_kernel<4, 8, 16>, LowGCfg<4, 8, 16>::LDS) | raise_lds((const void*)conv_lowg_kernel<8, 8, 8>, LowGCfg<8, 8, 8>::LDS);
          }();
          if (once) { mi355_set_error("conv_lowg: cannot raise the dynamic LDS limit to %d bytes (hip error %d)", LowGCfg<4, 8, 16>::LDS, once); return MI355_ERR_HIP; }
          if (p.shape == 12) conv_lowg_kernel<4, 8, 16><<<grid, block, LowGCfg<4, 8, 16>::LDS, st>>>(a);
          else conv_lowg_kernel<8, 8, 8><<<grid, block, LowGCfg<8, 8, 8>::LDS, st>>>(a);
        }
      } else { HALO_KS(3) }
    } else if (p.shape == 14) {
      if constexpr (sizeof(T) == 2) {
        static const int once = [] {
          return raise_lds((const void*)conv_march2_kernel<2, false>, March2Cfg<2>::LDS) | raise_lds((const void*)conv_march2_kernel<2, true>, March2Cfg<2>::LDS);
        }();
        if (once) { mi355_set_error("conv_march2: cannot raise the dynamic LDS limit to %d bytes (hip error %d)", March2Cfg<2>::LDS, once); return MI355_ERR_HIP; }
        March2Args m{p.seg_len, p.nseg, p.tiles_h, p.tiles_w, d->addend, d->ld_add, d->y_f32, d->add_n, d->add_bf16, d->delta};
        if (d->d2s) conv_march2_kernel<2, true><<<grid, block, March2Cfg<2>::LDS, st>>>(a, m);
        else conv_march2_kernel<2, false><<<grid, block, March2Cfg<2>::LDS, st>>>(a, m);
      }
    } else { HALO_KS(2) }
  } else if (sizeof(T) == 2 && p.pointwise) {
    // full-resolution 1x1x1 convs with <= 32 channels either side: persistent streaming kernel
    const int ntiles = (int)p.tiles;
    const dim3 g1((unsigned)(ntiles < 2048 ? ntiles : 2048));
    if (a.nchunks == 1) pointwise_conv_kernel<1><<<g1, block, 0, st>>>(a, ntiles, p.wg_stats ? 1 : 0);
    else pointwise_conv_kernel<2><<<g1, block, 0, st>>>(a, ntiles, p.wg_stats ? 1 : 0);
  } else if (sizeof(T) == 2 && d->cls_cout && p.vt == 2 && a.nchunks <= 8 && d->c1 == 0 && (d->cstore & 7) == 0) {
    // transposed-conv forward, Cin <= 128: one workgroup per 256 voxels loops over all column blocks
    const dim3 g1((unsigned)p.tiles);
    if (a.nchunks <= 4) deconv_fwd_kernel<4><<<g1, block, 0, st>>>(a);
    else deconv_fwd_kernel<8><<<g1, block, 0, st>>>(a);
  } else {
    if (p.vt == 2) {
      if (p.ct == 2) conv_gather_kernel<T, 2, 2><<<grid, block, 0, st>>>(a);
      else conv_gather_kernel<T, 2, 1><<<grid, block, 0, st>>>(a);
    } else {
      if (p.ct == 2) conv_gather_kernel<T, 1, 2><<<grid, block, 0, st>>>(a);
      else conv_gather_kernel<T, 1, 1><<<grid, block, 0, st>>>(a);
    }
  }
#undef HALO
#undef HALO_KS
  if (p.ksplit > 1) {
    int rc = mi355_check_launch("conv_fwd");
    if (rc) return rc;
    conv_ksplit_reduce_kernel<T><<<dim3((unsigned)p.stat_rows, (unsigned)((d->coutp + 1023) / 1024)), dim3(256), 0, st>>>(a, p.rpb);
    return mi355_check_launch("conv_ksplit_reduce");
  }
  return mi355_check_launch("conv_fwd");
}

}  // namespace

extern "C" int mi355_conv_plan_id(const mi355_conv_desc* d) {
  Plan p;
  int rc = make_plan(d, &p);
  if (rc) return rc;
  return 10000 * d->ks + 1000 * (p.halo ? 1 : 0) + 100 * p.shape + 10 * p.vt + p.ct;
}

extern "C" int mi355_conv_num_tiles(const mi355_conv_desc* d, int32_t* tiles, int32_t* tiles_per_sample) {
  Plan p;
  int rc = make_plan(d, &p);
  if (rc) return rc;
  if (tiles) *tiles = (int32_t)p.stat_rows;
  if (tiles_per_sample) *tiles_per_sample = p.stat_rows_per_sample;
  return MI355_OK;
}

extern "C" int64_t mi355_conv_workspace_bytes(const mi355_conv_desc* d) {
  Plan p;
  if (make_plan(d, &p)) return -1;
  if (p.ksplit <= 1) return 0;
  return (int64_t)p.ksplit * d->n * d->do_ * d->ho * d->wo * d->coutp * 4;
}

extern "C" int mi355_conv_fwd(const mi355_conv_desc* d, void* stream) {
  Plan p;
  int rc = make_plan(d, &p);
  if (rc) return rc;
  hipStream_t st = (hipStream_t)stream;
  if (d->dtype == MI355_DT_F32) return launch<float>(d, p, st);
  return launch<bf16_t>(d, p, st);
}
