/*
 * mi355_unet.h -- C ABI of the MI355X-native (gfx950 / CDNA4) 3D U-Net + PatchGAN hot path.
 *
 * Drop-in boundary.  The reference (SomeUserName1/UNet-bSSFP) has no
 * FFI of its own: its hot path sits behind torch.nn.Module construction sites
 *   - torch.nn.Conv3d / BatchNorm3d / LeakyReLU   in DownSampleConv   (src/model.py:50-57)
 *   - mainets.nets.BasicUNet(...)                  in Generator        (src/model.py:22-28)
 *   - the Discriminator layer list                                    (src/model.py:72-83)
 * and torch.optim.AdamW (src/model.py:359-361).  Each entry point below names the reference
 * operator it stands in for.  The Python side (unet_bssfp_amd/) binds these with ctypes
 * (see INTEGRATION.md for the stub a reference maintainer would add).
 *
 * Conventions
 *  - Plain pointers and sizes only; no torch types.  All pointers are DEVICE pointers owned by
 *    the caller (PyTorch's caching allocator); the library never allocates, frees or keeps them.
 *  - Every function launches only on the `stream` passed in (a hipStream_t) and never syncs.
 *  - Return value: 0 on success, negative on error; mi355_last_error() gives the message
 *    (thread-local).  No exceptions, no abort.
 *  - Re-entrant: called from the main Python thread (forward) and the autograd thread (backward).
 *  - Activation layout inside the path: NDHWC, element (n,d,h,w,c) at
 *        ((((n*D + d)*H + h)*W + w) * ld + c),   ld >= C, C padded to a multiple of 16,
 *    pad channels hold zeros.  `ld` lets a tensor be a channel slice of a wider buffer
 *    (zero-copy skip-concat).  dtype: 0 = f32 (parity mode), 1 = bf16 (throughput mode;
 *    f32 accumulate, f32 statistics, f32 master weights).
 */
#ifndef MI355_UNET_H
#define MI355_UNET_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MI355_DT_F32 0
#define MI355_DT_BF16 1
#define MI355_DT_F64 2   /* mi355_dti_scalar_maps only */
#define MI355_DT_FP8 3   /* OCP e4m3: operands of the fp8 convolution path only (mi355_conv_fwd, mi355_weight_pack) */

#define MI355_OK 0
#define MI355_ERR_ARG (-1)
#define MI355_ERR_UNSUPPORTED (-2)
#define MI355_ERR_HIP (-3)

int mi355_version(void);
const char* mi355_last_error(void);

/* ------------------------------------------------------------------------------------------
 * Layout conversion at the module boundary.
 * Reference: the NCDHW tensors handed to Generator.forward / Discriminator.forward
 * (src/model.py:36-39, 85-92; torch.cat([x, y], 1) at :86 is realised by two packs into one
 * 32-channel buffer).
 * pack:   dst[n,d,h,w, coff + c] = src[n,c,d,h,w]  for c < C;  channels [coff+C, zero_to) = 0
 * unpack: dst[n,c,d,h,w] = src[n,d,h,w, coff + c]
 * src/dst NCDHW side is contiguous f32.  V = D*H*W.
 * ---------------------------------------------------------------------------------------- */
int mi355_pack_ncdhw(const float* src, void* dst, int32_t n, int32_t c, int64_t v,
                     int32_t ld, int32_t coff, int32_t zero_to, int32_t dtype, void* stream);
int mi355_unpack_ncdhw(const void* src, float* dst, int32_t n, int32_t c, int64_t v,
                       int32_t ld, int32_t coff, int32_t dtype, void* stream);
/* Space-to-depth variants (PatchGAN path).  For a plain tensor a (N,D,H,W,C), even extents,
 *   S(a)[n, jd, jh, jw, blk*cblk + c] = a[n, 2jd+bd-1, 2jh+bh-1, 2jw+bw-1, c],  blk = 4bd+2bh+bw,
 * extents (D/2+1, H/2+1, W/2+1), 8*cblk channels per row.  Blocks of the border cells that lie outside the volume
 * hold zeros: the thread that writes a border voxel also writes them (the written channel range of every such
 * block), so S needs no prior zero-fill once all its channels have been packed.  The reference's Conv3d(k=4, s=2, p=1) (src/model.py:44,50) on a equals a
 * dense k=2, s=1, p=0 convolution on S(a), so the PatchGAN runs on the stride-1 kernels.
 * pack:  writes channels [coff, coff+c) (zeros up to zero_to) of every block;  unpack: reads them. */
int mi355_pack_ncdhw_s2d(const float* src, void* dst, int32_t n, int32_t c, int32_t d, int32_t h, int32_t w,
                         int32_t cblk, int32_t ld, int32_t coff, int32_t zero_to, int32_t dtype, void* stream);
/* two sources in one pass: channels [coff, coff+c0) <- src0, [coff+c0, coff+c0+c1) <- src1, the rest of the
 * window up to zero_to <- 0: `torch.cat([x, y], 1)` of the discriminator (src/model.py:86) written as whole rows */
int mi355_pack2_ncdhw(const float* src0, int32_t c0, const float* src1, int32_t c1, void* dst, int32_t n, int64_t v,
                      int32_t ld, int32_t coff, int32_t zero_to, int32_t dtype, void* stream);
int mi355_pack2_ncdhw_s2d(const float* src0, int32_t c0, const float* src1, int32_t c1, void* dst, int32_t n, int32_t d,
                          int32_t h, int32_t w, int32_t cblk, int32_t ld, int32_t coff, int32_t zero_to, int32_t dtype,
                          void* stream);
int mi355_unpack_ncdhw_s2d(const void* src, float* dst, int32_t n, int32_t c, int32_t d, int32_t h, int32_t w,
                           int32_t cblk, int32_t ld, int32_t coff, int32_t dtype, void* stream);

/* ------------------------------------------------------------------------------------------
 * Weight packing.  Reference weights are torch layout: Conv3d (Cout,Cin,k,k,k) f32,
 * ConvTranspose3d (Cin,Cout,2,2,2) f32 (SURVEY.md 8(b) state_dict table).
 * dst[chunk][tap][co][16] (dtype) with ci = chunk*16 + e;  zero where co >= cout or ci >= cin.
 * Source element for GEMM indices (co, ci, tap=(td,th,tw) in a ks^3 kernel):
 *     src[co*s_co + ci*s_ci + kd*s_k[0] + kh*s_k[1] + kw*s_k[2]],  k* = tbase[*] + tstep[*]*t*
 * which expresses forward, flipped/transposed (dgrad) and the parity classes of the
 * stride-2 transposed convolutions without host-side tensor shuffles.
 * ---------------------------------------------------------------------------------------- */
typedef struct mi355_wpack_desc {
  const float* src;
  void* dst;
  int32_t cout, cin;          /* real GEMM extents            */
  int32_t coutp, cinp;        /* padded: coutp%32==0, cinp%16==0 */
  int32_t ks;                 /* dst kernel edge (taps = ks^3)   */
  int64_t s_co, s_ci;
  int64_t s_k[3];
  int32_t tbase[3], tstep[3];
  int32_t dtype;
  /* space-to-depth operands (mi355_pack_ncdhw_s2d): s2d_mode 1 = the GEMM cin index is
   * blk*s2d_cp + c (cin = real channels per block, cinp = 8*s2d_cp), 2 = the GEMM cout index is;
   * the block's parity bits (bd,bh,bw) are added to the source tap coordinates. 0 = off. */
  int32_t s2d_mode, s2d_cp;
  /* dtype MI355_DT_FP8: device f32[1] holding max |src| (mi355_amax_f32); packed value = e4m3(w * 224 / amax) */
  const float* q_amax;
} mi355_wpack_desc;
int mi355_weight_pack(const mi355_wpack_desc* d, void* stream);
/* n packings in ceil(n/16) launches (descriptors travel by value; all must share one dtype) */
int mi355_weight_pack_multi(const mi355_wpack_desc* descs, int32_t n, void* stream);

/* ------------------------------------------------------------------------------------------
 * Implicit-GEMM 3D convolution on MFMA (forward of Conv3d; data-gradient of Conv3d and
 * ConvTranspose3d; forward of ConvTranspose3d as 8 parity classes).
 * Reference ops: torch.nn.Conv3d in DownSampleConv (src/model.py:50-51), Discriminator.final
 * (:83), MONAI BasicUNet Convolution / final_conv / UpSample.deconv (call site :22-28).
 *
 *   z[n, p*os+ooff, co] = bias[co] + sum_{tap, ci} x[n, p*stride + tap - pad, ci] * w[tap][co][ci]
 *   for p in the conv grid (do_,ho,wo); x is the virtual concat [x0 | x1] (x1 may be NULL).
 * Out-of-range input positions read as zero.  Optional fused per-tile channel statistics of
 * (z - bias): stats_part[tile][2][coutp] = {sum, sum of squares} (deterministic, no atomics);
 * tiles are never shared between samples when ks==3 (InstanceNorm groups).
 * ---------------------------------------------------------------------------------------- */
typedef struct mi355_conv_desc {
  const void* x0; int32_t c0; int32_t ld0;
  const void* x1; int32_t c1; int32_t ld1;
  int32_t n, di, hi, wi;          /* input spatial extents            */
  int32_t do_, ho, wo;            /* conv grid extents                */
  int32_t ks, stride;
  int32_t pad[3];
  const void* wp;                 /* packed weights, see mi355_weight_pack */
  int32_t coutp;
  const float* bias;              /* f32 [nbias] or NULL; channels >= nbias get no bias */
  void* y; int32_t ldy; int32_t cstore;
  int32_t dy, hy, wy;             /* physical output extents          */
  int32_t os; int32_t ooff[3];
  float* stats_part;              /* NULL or [mi355_conv_num_tiles()][2][coutp] */
  int32_t dtype;
  /* scratch for the split-K path (few output positions, long contraction: the low U-Net / PatchGAN
   * levels): mi355_conv_workspace_bytes() bytes of f32, may be NULL when that returns 0 */
  void* workspace; int64_t workspace_bytes;
  /* ConvTranspose3d(k=2, s=2) forward as ONE 1x1x1 GEMM with 8*cls_cout output columns: column
   * blk*cls_cout + co is stored at output position 2p + (bd,bh,bw) (os must be 2), channel co.
   * 0 = off.  Weights: mi355_weight_pack with s2d_mode 2, s2d_cp = cls_cout. */
  int32_t cls_cout;
  int32_t nbias;                  /* length of bias (0: coutp, for callers that pad it) */
  /* dtype MI355_DT_FP8 (BASELINE.json configs[4]: 3x3x3 stride-1 layers with 32 input channels at full resolution):
   * x0 is e4m3 of x * 224 / amax_x (mi355_cast_fp8: one byte per channel, ld0 in bytes), wp is packed with
   * MI355_DT_FP8, y / statistics are bf16 / f32 as in the bf16 mode.  Both amax pointers: device f32[1]. */
  const float* q_amax_x; const float* q_amax_w;
  /* Accumulator start value (bf16 dense 2x2x2 stride-1 convolutions on the marching kernel only: PatchGAN blocks on
   * space-to-depth tensors, src/model.py:72-82): z = conv(x) + addend + bias, fused statistics of conv(x) + addend.
   * addend: f32 [n][do_][ho][wo][ld_add] (channels >= coutp), NULL = zeros.  y_f32 != 0: y is f32 (ldy in f32 elements)
   * instead of the operand type -- the x-part of the PatchGAN's first block, computed once per training step, travels
   * between two launches without a rounding.  Plans that cannot honour either field fail with MI355_ERR_UNSUPPORTED. */
  const float* addend; int32_t ld_add; int32_t y_f32;
  int32_t add_n;                  /* samples held by addend: sample i of the grid starts from addend sample i % add_n (0 = n) */
  /* d2s != 0 (depth-to-space; bf16, ks 2): the transpose of a Conv3d(k4, s2, p1) -- MONAI UpCat's up-branch as one
   * ConvTranspose3d(k4, s2, p1) of the low-resolution tensor (mi355_upcat_compose packs the weights: coutp = 8 * co columns
   * (class, channel)).  x0: [n][di][hi][wi][c0]; the grid (do_, ho, wo) must equal (di, hi, wi); y: the PLAIN tensor
   * [n][2 do_][2 ho][2 wo][ldy] (dy = 2 do_ ...), cstore <= co channels; output class b writes voxel 2 j + b.  bias [co];
   * addend (f32, or bf16 when add_bf16 != 0): [add_n or n][dy][hy][wy][ld_add]; delta: NULL or f32 [27][co], added to the
   * accumulators of the voxels on the volume's border (class 9 cd + 3 ch + cw: 0 first / 1 interior / 2 last voxel);
   * stats_part: [8 * tiles][2][co], rows (tile, class). */
  int32_t d2s; const float* delta; int32_t add_bf16;
} mi355_conv_desc;
int mi355_conv_fwd(const mi355_conv_desc* d, void* stream);
int64_t mi355_conv_workspace_bytes(const mi355_conv_desc* d);
/* which kernel instance mi355_conv_fwd() will launch for this descriptor (for profiling tools):
 * 10000*ks + 1000*halo + 100*tile_shape + 10*voxel_subtiles_per_wave + cout_subtiles_per_wave, <0 on error */
int mi355_conv_plan_id(const mi355_conv_desc* d);
/* number of spatial tiles (= rows of stats_part) and tiles per sample (0 if tiles span samples) */
int mi355_conv_num_tiles(const mi355_conv_desc* d, int32_t* tiles, int32_t* tiles_per_sample);

/* ------------------------------------------------------------------------------------------
 * The up-branch of MONAI's UpCat (BasicUNet(upsample="deconv"), call site src/model.py:22-28) without the up-sampled tensor:
 * ConvTranspose3d(cl -> cu, k2, s2) followed by the [:, ce:] part of Conv3d(ce + cu -> co, k3, p1) is ONE transposed
 * convolution of the low-resolution tensor with the 4x4x4 kernel k4[cl][co][4][4][4] (csrc/upcat.hip; even extents).
 *   mi355_upcat_compose: k4 from wd [cl][cu][2][2][2] and wc [co][ce + cu][3][3][3]; optionally its bf16 packing for
 *     mi355_conv_fwd's d2s mode (wp_d2s: [cl / 16][8][8 co][16]), the bias vector the consumer adds (biasp [co] =
 *     bc + the interior share of the transposed convolution's bias bd) and the corrections of the 27 border classes
 *     (delta [27][co], class = 9 cd + 3 ch + cw with 0 first voxel / 1 interior / 2 last voxel per axis).
 *   mi355_upcat_chain: dwd, dwc[:, ce:] (+ the bias-path term) and dbd from dk4 and the border sums of dz (esum [27][co],
 *     mi355_border_sums); dwc is the FULL [co][ce + cu][27] tensor, only the [:, ce:] part is written.
 *   mi355_border_sums: e[9 sd + 3 sh + sw][c] = sum of g over the region (per axis: 0 all / 1 first / 2 last voxel); the
 *     region (0, 0, 0) is NOT computed (0: the sum of a normalisation's input gradient over the volume is zero).
 *   mi355_s2d_repack: plain NDHWC activation -> its space-to-depth tensor (same element type), every slot written.
 * ---------------------------------------------------------------------------------------- */
int mi355_upcat_compose(const float* wd, const float* wc, const float* bd, const float* bc, int32_t cl, int32_t cu, int32_t ce,
                        int32_t co, float* k4, void* wp_d2s, float* biasp, float* delta, void* stream);
int mi355_upcat_chain(const float* dk4, const float* wd, const float* wc, const float* bd, const float* esum, int32_t cl, int32_t cu,
                      int32_t ce, int32_t co, float* dwd, float* dwc, float* dbd, int32_t accumulate, void* stream);
int64_t mi355_border_sums_workspace(int32_t n, int32_t d, int32_t c);
int mi355_border_sums(const void* g, int32_t ld, int32_t n, int32_t d, int32_t h, int32_t w, int32_t c, int32_t dtype, float* workspace,
                      float* e, void* stream);
int mi355_s2d_repack(const void* src, int32_t ld_src, void* dst, int32_t ld_dst, int32_t n, int32_t d, int32_t h, int32_t w,
                     int32_t c, int32_t dtype, void* stream);
/* Gradient seam PatchGAN -> generator (src/model.py:172, 268): dz[n][v][0..cpad) = g_ncdhw[n][ch][v] (f32 NCDHW gradient of the
 * generator's output from the loss head; NULL = absent) + the space-to-depth gradient g_s2d of S(output) (cblk channels per block;
 * NULL = absent), channels >= c zero: the NDHWC gradient the final convolution's backward reads, in one pass. */
int mi355_seam_grad(const float* g_ncdhw, const void* g_s2d, int32_t ld_s, int32_t cblk, void* dz, int32_t ld_dz, int32_t cpad,
                    int32_t n, int32_t c, int32_t d, int32_t h, int32_t w, int32_t dtype, void* stream);

/* ------------------------------------------------------------------------------------------
 * Weight gradient (autograd of Conv3d / ConvTranspose3d weights).
 *   dw[tap][ci][co] = sum_{n,p} x[n, p*stride + tap - pad, ci] * g[n, p, co]
 * computed as per-split partial slabs (deterministic) and reduced + scattered into the torch
 * layout by mi355_wgrad_reduce: dst[co*s_co + ci*s_ci + tap offsets] (+)= sum_splits slab.
 * workspace: mi355_conv_wgrad_workspace() bytes, f32.
 * ---------------------------------------------------------------------------------------- */
typedef struct mi355_wgrad_desc {
  const void* x0; int32_t c0; int32_t ld0;
  const void* x1; int32_t c1; int32_t ld1;
  int32_t n, di, hi, wi;
  const void* g; int32_t cg; int32_t ldg;   /* grad wrt conv output, cg channels (mult of 16) */
  int32_t do_, ho, wo;                      /* extents of g's grid */
  int32_t gd, gh, gw;                       /* physical extents of g */
  int32_t gs; int32_t goff[3];              /* g position = p*gs + goff (transposed-conv classes) */
  int32_t ks, stride;
  int32_t pad[3];
  float* workspace; int64_t workspace_bytes;
  /* destination (torch layout), real extents */
  float* dw; int32_t cout, cin;
  int64_t s_co, s_ci; int64_t s_k[3];
  int32_t tbase[3], tstep[3];
  int32_t accumulate;                       /* 0: overwrite, 1: add into dw */
  int32_t dtype;
  int32_t s2d_cp;                           /* >0: x is a space-to-depth tensor, ci = blk*s2d_cp + c (cin = real c) */
  int32_t g_cls_cout;                       /* >0: weight gradient of ConvTranspose3d(k2,s2) in one launch: GEMM column
                                               blk*g_cls_cout + co reads g at position 2p + (bd,bh,bw), channel co, and
                                               lands in tap (bd,bh,bw) of dw (ks must be 1, bf16 only) */
  int32_t xn;                               /* >0: x holds xn samples and sample i of the grid reads x sample i % xn (one input
                                               under several gradients: the PatchGAN's first block sees the same x in
                                               both calls of the discriminator phase, src/model.py:184-186); 0 = n */
} mi355_wgrad_desc;
int64_t mi355_conv_wgrad_workspace(const mi355_wgrad_desc* d);
int mi355_conv_wgrad(const mi355_wgrad_desc* d, void* stream);
/* which kernel family mi355_conv_wgrad takes for this descriptor (tests, profiles): 2 = marching bf16 kernel
 * (3x3x3, W >= 32), 3 = streaming bf16 kernel of the full-resolution 1x1x1 layers, 4 = marching bf16 kernel of the dense
 * 2x2x2 layers (PatchGAN on space-to-depth operands, UpCat's composite kernel; W >= 32), 1 = tile-form bf16 MFMA kernel,
 * 0 = exact-f32 kernel; -1 = invalid descriptor */
int mi355_conv_wgrad_plan_kind(const mi355_wgrad_desc* d);
/* The same weight gradient in two halves, for callers that own a whole backward pass (src/model.py:259-281: nothing reads a
 * weight's .grad before the optimiser step, or before the bucket's all-reduce under DDP, src/train.py:30-32):
 * mi355_conv_wgrad_partial launches only the kernel that writes the f32 slabs into d->workspace and fills `job`;
 * mi355_wgrad_reduce_multi sums the slabs of n such jobs into their dw in ONE launch per 16 jobs (same summation order per job as
 * mi355_conv_wgrad: bit-identical results).  The caller keeps every job's workspace alive and unmodified until the reduce has
 * run on the same stream, and two jobs of one call must not address the same dw element (accumulate = 1 jobs add to what
 * dw held BEFORE the call).  `job` is opaque. */
typedef struct mi355_wreduce_job { int64_t opaque[20]; } mi355_wreduce_job;
int mi355_conv_wgrad_partial(const mi355_wgrad_desc* d, mi355_wreduce_job* job, void* stream);
int mi355_wgrad_reduce_multi(const mi355_wreduce_job* jobs, int32_t n, void* stream);

/* ------------------------------------------------------------------------------------------
 * Channel statistics and fused normalisation + dropout + LeakyReLU.
 * Reference ops: torch.nn.BatchNorm3d + LeakyReLU(0.2) (src/model.py:53-57, 61-64) and MONAI ADN
 * "NDA" = InstanceNorm3d(affine) -> Dropout(p) -> LeakyReLU(0.1) (BasicUNet, call site :22-28).
 * Groups: InstanceNorm = one group per sample (groups = N); BatchNorm = one group (groups = 1).
 * ---------------------------------------------------------------------------------------- */
/* partial sums over voxel blocks: part[block][2][c] of x (f32 accumulate); rows = N*V.
 * blocks never span groups: blocks_per_group = ceil(rows_per_group / rows_per_block). */
int mi355_channel_stats(const void* x, int32_t ld, int32_t c, int64_t rows_per_group, int32_t groups,
                        float* part, int32_t blocks_per_group, int32_t dtype, void* stream);
int32_t mi355_channel_stats_blocks(int64_t rows_per_group);

/* mean/rstd per (group, channel) from partials (f64 combine).  shift[c] (may be NULL) is added to
 * the mean (conv bias when partials were taken before the bias add).  Optionally updates
 * BatchNorm running stats: rm = (1-mom)*rm + mom*mean, rv = (1-mom)*rv + mom*var*cnt/(cnt-1).
 * batches_tracked (optional, device int64[1]): BatchNorm's num_batches_tracked, advanced by one.
 */
int mi355_norm_finalize(const float* part, int32_t parts_per_group, int32_t groups, int32_t c,
                        int64_t count_per_group, const float* shift, int32_t n_real, float eps,
                        float* mean, float* rstd,
                        float* running_mean, float* running_var, float momentum, int64_t* batches_tracked, void* stream);

typedef struct mi355_normact_desc {
  const void* z; int32_t ldz;        /* conv output */
  void* a; int32_t lda;              /* activated output (fwd) / unused (bwd) */
  int32_t c; int64_t rows_per_group; int32_t groups;
  const float* mean; const float* rstd;   /* [groups][c]; NULL => no normalisation */
  const float* gamma; const float* beta;  /* [c]; NULL => 1 / 0 */
  float slope;                            /* LeakyReLU negative slope; 1.0 => identity */
  float drop_p; uint64_t seed;            /* element-wise dropout, p == 0 => off */
  int32_t dtype;
  /* backward only */
  const void* da; int32_t ldda;
  void* dz; int32_t lddz;
  float* part;                            /* [blocks][2][c] partial sums of g and g*xhat */
  int32_t blocks_per_group;
  const float* sums;                      /* [groups][2][c] reduced sums (bwd_apply) */
  int32_t batch_stats;                    /* 1: subtract mean terms (train), 0: eval-mode norm */
  /* space-to-depth coupling with the next k4 s2 convolution (see mi355_pack_ncdhw_s2d):
   * s2d_a: forward writes a as S(a) (lda = row stride of S, >= 8*c);  s2d_da: backward reads da
   * from a gradient in S layout.  sd/sh/sw = plain extents of z per sample. */
  int32_t s2d_a, s2d_da;
  int32_t sd, sh, sw;
  /* optional DEVICE pointer to a 64-bit step counter mixed into the dropout seed (`seed` is then a
   * per-call salt): a launch captured in a hipGraph draws a new mask on every replay */
  const uint64_t* seed_ptr;
  int32_t n_affine;               /* length of gamma/beta (0: c); channels beyond it use gamma 0, beta 0 */
  /* optional e4m3 copy for the fp8 convolution that consumes the result (BASELINE.json configs[4], delayed per-tensor
   * scaling): fwd writes q8 = e4m3(a * 224 / q_use[0]), bwd_apply q8 = e4m3(dz * 224 / q_use[0]) -- byte for byte what
   * mi355_cast_fp8 makes of the stored bf16 tensor -- one byte per channel, row stride ld8 bytes, and raises q_next[0] to
   * max |a| (|dz|), the scale of the NEXT step (mi355_fp8_scale_roll).  bf16, c == 32, plain layouts only.  NULL: off. */
  void* q8; int32_t ld8; const float* q_use; float* q_next;
  /* optional, backward (bwd_reduce / bwd_apply, bf16): `da` is NOT materialised -- it is the data gradient of the 1x1x1 convolution
   * that consumed a (the U-Net's final convolution, src/model.py:22-28 via MONAI BasicUNet.final_conv):
   * da[row][ch] = bf16(sum_{k < gk} gz[row][k] * bf16(gw[k * gw_ld + ch])); gz = gradient of that convolution's output, rows of ldgz >= 8
   * elements of `dtype` (gk <= 8 real channels first), gw = its f32 master weights [gk][gw_ld] (channels >= gw_ld: zero).
   * `da` must be NULL.  Saves the data-gradient launch (a 134-MB write at 128^3) and 100 MB of reads in each backward pass. */
  const void* gz; int32_t ldgz; const float* gw; int32_t gw_ld; int32_t gk;
  /* optional, forward (bf16, c = 32, plain layout, no q8): the same 1x1x1 convolution evaluated in the pass that writes a, on the
   * rounded bf16 values: fy[row][k] = bf16(sum_ch a[row][ch] * bf16(gw[k * gw_ld + ch]) + fbias[k]) for k < gk, zero for gk <= k < fcp
   * (rows of ldfy elements); skip_a = 1: a itself is not stored (a no-grad pass whose only consumer of a is that convolution).
   * Saves the convolution's launch and its 134-MB read of a at 128^3. */
  void* fy; int32_t ldfy; int32_t fcp; const float* fbias; int32_t skip_a;
  /* optional, backward (bwd_reduce / bwd_apply): `da` is NOT materialised by a max-pool backward launch -- a was consumed by
   * MaxPool3d(2) (MONAI BasicUNet's Down blocks, src/model.py:22-28) and, optionally, by a skip connection whose gradient is `da`
   * (NULL: none): da[v][ch] = (pool_idx[o][ch] == k ? pool_dy[o][ch] : 0) + da[v][ch], rounded to `dtype`, where v = (n, d, h, w) on the
   * sd x sh x sw grid (even extents), o = (n, d/2, h/2, w/2) and k = 4 (d&1) + 2 (h&1) + (w&1); pool_idx = the window positions
   * mi355_maxpool2_fwd_idx recorded ([rows/8][c] bytes), pool_dy = the pooled tensor's gradient (rows of ldpdy elements).
   * Bit-identical to mi355_maxpool2_bwd(_add) followed by the plain kernels; saves that launch's 134-MB write and 3 x 134 MB of
   * reads per backward pass at 128^3 x 32. */
  const void* pool_idx; const void* pool_dy; int32_t ldpdy;
  /* optional, forward: MaxPool3d(2) of a in the pass that writes it (the same Down blocks): pool_y[(n, d/2, h/2, w/2)][ch] (rows of
   * ldpy elements) and pool_widx (bytes, as mi355_maxpool2_fwd_idx) on the sd x sh x sw grid (even extents; a group is one sample
   * or the whole batch); plain layouts, no q8 / fy.  Bit-identical to mi355_normact_fwd + mi355_maxpool2_fwd_idx. */
  void* pool_y; int32_t ldpy; uint8_t* pool_widx;
} mi355_normact_desc;
int mi355_normact_fwd(const mi355_normact_desc* d, void* stream);
int mi355_normact_bwd_reduce(const mi355_normact_desc* d, void* stream);
/* sums[g][2][c] = sum over blocks of part; dgamma[c] = sum_g sums[g][1][c], dbeta = sum_g sums[g][0][c] */
int mi355_normact_bwd_finalize(const float* part, int32_t blocks_per_group, int32_t groups, int32_t c,
                               float* sums, float* dgamma, float* dbeta, void* stream);
/* the same, writing the affine gradients straight into caller-owned gradient storage (a parameter's .grad, possibly a
 * slice of a flat all-reduce bucket): only the first n_affine channels are written, `accumulate` adds to what is there
 * (second use of the layer in one backward pass -- Discriminator in _discr_step, src/model.py:185-186) */
int mi355_normact_bwd_finalize_into(const float* part, int32_t blocks_per_group, int32_t groups, int32_t c,
                                    float* sums, float* dgamma, float* dbeta, int32_t n_affine, int32_t accumulate,
                                    void* stream);
int mi355_normact_bwd_apply(const mi355_normact_desc* d, void* stream);

/* Norm + dropout + LeakyReLU of a SMALL tensor (the 16^3 / 8^3 U-Net levels, the last PatchGAN blocks: src/model.py:22-28,
 * 79-82) as ONE launch each way: a workgroup owns 16 bytes of channels of every row, computes the statistics itself
 * (mean, then variance about it) and applies them -- no statistics from the convolution, no finalize launch.
 * fwd: writes a, mean_out / rstd_out [groups][c] (kept for the backward pass) and, for BatchNorm in training mode, the running
 * statistics (momentum update with the unbiased variance, groups in order) and batches_tracked += groups.
 * bwd: base.mean / base.rstd = what fwd wrote; writes dz and the affine gradients (first n_affine channels; `accumulate`
 * adds to what is there).  base.sums / base.blocks_per_group are unused.  base.part is unused by fwd; bwd takes it as optional
 * scratch of groups x 2 x c DOUBLES (8-byte aligned): with it, chunks of statistic groups run as independent workgroups and a
 * second small launch sums the affine gradients over the groups (same order and precision: bit-identical to the form without). */
typedef struct mi355_normact_small_desc {
  mi355_normact_desc base;
  float eps, momentum;
  float* mean_out; float* rstd_out;
  float* running_mean; float* running_var; int64_t* batches_tracked; int32_t n_real;
  float* dgamma; float* dbeta; int32_t accumulate;
} mi355_normact_small_desc;
int mi355_normact_small_fwd(const mi355_normact_small_desc* d, void* stream);
int mi355_normact_small_bwd(const mi355_normact_small_desc* d, void* stream);

/* per-channel sum over all rows (bias gradient of a conv without normalisation):
 * out[c] = sum_rows x[row][c] from channel_stats partials (parts x [2][c]) */
int mi355_colsum_finalize(const float* part, int32_t parts, int32_t c, float* out, void* stream);
/* the same into caller-owned gradient storage: first n_out channels, optional accumulation */
int mi355_colsum_finalize_into(const float* part, int32_t parts, int32_t c, float* out, int32_t n_out, int32_t accumulate,
                               void* stream);
/* the same for channels offset .. offset + n_out of the partial rows -- e.g. the statistics a data-gradient convolution launch
 * emitted for ALL its output channels, of which the second source's are the bias gradient of the transposed convolution that
 * produced that source (MONAI UpCat, src/model.py:22-28): no separate pass over the gradient */
int mi355_colsum_finalize_from(const float* part, int32_t parts, int32_t c, int32_t offset, float* out, int32_t n_out,
                               int32_t accumulate, void* stream);

/* ------------------------------------------------------------------------------------------
 * MaxPool3d(kernel_size=2) -- MONAI BasicUNet `Down` (call site src/model.py:22-28).
 * Extents >= 2; an odd extent is floored (the last plane / row / column belongs to no window).  Backward routes
 * the gradient to the first maximum in (d,h,w) scan order (torch CPU semantics) and writes zeros elsewhere (dx
 * fully written, the planes outside every window included).
 * ---------------------------------------------------------------------------------------- */
int mi355_maxpool2_fwd(const void* x, int32_t ldx, void* y, int32_t ldy, int32_t n, int32_t c,
                       int32_t d, int32_t h, int32_t w, int32_t dtype, void* stream);
/* the same, also recording for every pooled element the window position (4 kd + 2 kh + kw, one byte, idx[(n,od,oh,ow)][c]) the
 * backward pass routes its gradient to -- what mi355_normact_desc::pool_idx reads instead of a max-pool backward launch */
int mi355_maxpool2_fwd_idx(const void* x, int32_t ldx, void* y, int32_t ldy, uint8_t* idx, int32_t n, int32_t c,
                           int32_t d, int32_t h, int32_t w, int32_t dtype, void* stream);
int mi355_maxpool2_bwd(const void* x, int32_t ldx, const void* y, int32_t ldy,
                       const void* dy, int32_t lddy, void* dx, int32_t lddx,
                       int32_t n, int32_t c, int32_t d, int32_t h, int32_t w,
                       int32_t dtype, void* stream);
/* same, plus `add` (a second gradient of the pooled tensor: the skip-connection use of a U-Net encoder
 * level, src/model.py:22-28 via BasicUNet's skip concat) summed into dx in the same pass */
int mi355_maxpool2_bwd_add(const void* x, int32_t ldx, const void* y, int32_t ldy, const void* dy, int32_t lddy,
                           void* dx, int32_t lddx, const void* add, int32_t ldadd,
                           int32_t n, int32_t c, int32_t d, int32_t h, int32_t w,
                           int32_t dtype, void* stream);

/* ------------------------------------------------------------------------------------------
 * Losses -- torch.nn.L1Loss (src/model.py:126,136): mean |a - b| over `count` f32 elements.
 * fwd writes the scalar to out[0]; partials workspace >= mi355_l1_blocks(count) floats.
 * bwd: da[i] = sign(a[i]-b[i]) * gscale[0] / count   (gscale is a DEVICE scalar: no host sync)
 * ---------------------------------------------------------------------------------------- */
int32_t mi355_l1_blocks(int64_t count);
int mi355_l1_fwd(const float* a, const float* b, int64_t count, float* partials, float* out, void* stream);
int mi355_l1_bwd(const float* a, const float* b, int64_t count, const float* gscale, float* da, void* stream);
/* the partial sums only (mi355_l1_blocks(count) floats): the GAN generator loss below finishes them */
int mi355_l1_partials(const float* a, const float* b, int64_t count, float* partials, void* stream);

/* ------------------------------------------------------------------------------------------
 * GAN loss heads on the PatchGAN logit maps (a few hundred values): ONE launch forward, ONE backward, instead of the
 * ~45 scalar-sized torch launches per step of BCEWithLogits + the loss arithmetic (a launch costs ~5 us in a graph replay).
 *  _gen_step (src/model.py:126-137):  out4 = { L1 = sum(l1_partials) / count, recon = L1 / recon_divisor * recon_factor,
 *      adv = mean BCEWithLogits(logits, 1), adv + recon };  bwd: dlogits = upstream[0] (sigmoid(x) - 1) / n and
 *      l1_gscale[0] = upstream[0] * recon_factor / recon_divisor (the device scalar mi355_l1_bwd takes).
 *  _discr_step (src/model.py:183-193): out1[0] = (mean BCEWithLogits(real, 1) + mean BCEWithLogits(fake, 0)) / 2;
 *      bwd: dfake = upstream[0] / 2 * sigmoid(x) / n_fake, dreal = upstream[0] / 2 * (sigmoid(x) - 1) / n_real.
 * BCEWithLogits(x, t) = (1 - t) x - log_sigmoid(x) (torch's form); sums in f64, everything else f32.
 * ---------------------------------------------------------------------------------------- */
int mi355_gan_gen_loss_fwd(const float* logits, int32_t n, const float* l1_partials, int32_t n_partials, int64_t count,
                           float recon_divisor, float recon_factor, float* out4, void* stream);
int mi355_gan_gen_loss_bwd(const float* logits, int32_t n, const float* upstream, float recon_divisor, float recon_factor,
                           float* dlogits, float* l1_gscale, void* stream);
int mi355_gan_discr_loss_fwd(const float* logits_fake, int32_t n_fake, const float* logits_real, int32_t n_real, float* out1,
                             void* stream);
int mi355_gan_discr_loss_bwd(const float* logits_fake, int32_t n_fake, const float* logits_real, int32_t n_real,
                             const float* upstream, float* dfake, float* dreal, void* stream);

/* ------------------------------------------------------------------------------------------
 * Fused multi-tensor AdamW -- torch.optim.AdamW(params, lr) with torch defaults
 * (src/model.py:164, 359-361): decoupled weight decay, bias-corrected moments.
 * ptrs: HOST array of 4*ntensors device pointers {param, grad, exp_avg, exp_avg_sq} (all f32),
 * sizes: HOST array of ntensors element counts (they travel by value in the kernel arguments, so
 * the launches can be captured into a hipGraph); the 1-based step number is read from DEVICE
 * memory (`step_dev`) when given, else taken from `step`.
 * ---------------------------------------------------------------------------------------- */
int mi355_adamw_multi(const void* const* ptrs, const int64_t* sizes, int32_t ntensors,
                      float lr, float beta1, float beta2, float eps, float weight_decay,
                      const int64_t* step_dev, int64_t step, void* stream);

/* ------------------------------------------------------------------------------------------
 * DTI scalar maps -- the voxel loop of `do_calc_scalar_maps`
 * (src/eval.py:73-135): symmetric tensor (Dxx,Dxy,Dxz,Dyy,Dyz,Dzz) -> eigh -> FA, MD, AD, RD,
 * azimuth, inclination (degrees) and the FA-weighted |principal eigenvector| RGB map.
 * Component c of voxel v is read at tensor[v*vox_stride + c*comp_stride] (elements of `dtype`,
 * MI355_DT_F32 or MI355_DT_F64): (X,Y,Z,6) NIfTI order = strides (1, 6); a generator output
 * [6][D][H][W] = strides (D*H*W, 1).  Each component is first mapped to x*scale + offset, the
 * min-max de-normalisation of `do_invert_dwi_tensor_norm` (src/eval.py:39-47, scale = |max-min|,
 * offset = min); pass (1, 0) for none.  Outputs are `dtype` too: six [nvox] maps and rgb [nvox][3].
 * Arithmetic is f64 like the reference.  The eigenvector sign (arbitrary in LAPACK) is fixed to
 * z >= 0, so azimuth/inclination agree with the reference up to the antipodal map
 * (az, inc) ~ (az +- 180, 180 - inc).  A zero tensor gives FA = NaN like numpy's 0/0.
 * ---------------------------------------------------------------------------------------- */
int mi355_dti_scalar_maps(const void* tensor, int32_t dtype, int64_t nvox, int64_t comp_stride,
                          int64_t vox_stride, double scale, double offset, void* fa, void* md, void* ad,
                          void* rd, void* azimuth, void* inclination, void* rgb, void* stream);

/* ------------------------------------------------------------------------------------------
 * Sliding-window inference -- the data movement of `predict_step` /
 * `test_step` (src/model.py:291-333) around Generator.forward: TorchIO's GridSampler patch
 * extraction and GridAggregator.add_batch / get_output_tensor (src/data_module.py:168-183).
 * f32; volume [C][D][H][W], patches [B][C][pd][ph][pw].  Locations are HOST arrays (they travel
 * by value in the kernel arguments, at most MI355_MAX_PATCHES per launch, longer lists are chunked).
 *   gather   : origins[b][3] = first voxel of patch b.
 *   aggregate: locs9[b] = {origin[3], keep_ini[3], keep_fin[3]} in volume coordinates (the kept
 *              region is the patch minus TorchIO's half-overlap crop); MI355_AGG_CROP assigns, the
 *              LAST patch of the list covering a voxel wins (the reference's assignment order);
 *              MI355_AGG_AVERAGE adds into vol and counts into count[D*H*W], both zero-initialised
 *              by the caller, and mi355_patch_average_finalize divides.  Deterministic (no atomics).
 * ---------------------------------------------------------------------------------------- */
#define MI355_MAX_PATCHES 48
#define MI355_AGG_CROP 0
#define MI355_AGG_AVERAGE 1
int mi355_patch_gather(const float* vol, int32_t c, int32_t d, int32_t h, int32_t w, const int32_t* origins,
                       int32_t npatches, int32_t pd, int32_t ph, int32_t pw, float* out, void* stream);
int mi355_patch_aggregate(const float* patches, const int32_t* locs9, int32_t npatches, int32_t pd, int32_t ph,
                          int32_t pw, int32_t mode, float* vol, float* count, int32_t c, int32_t d, int32_t h,
                          int32_t w, void* stream);
int mi355_patch_average_finalize(float* vol, const float* count, int32_t c, int64_t voxels, void* stream);

/* ------------------------------------------------------------------------------------------
 * Validation metrics -- MONAI's MAEMetric / PSNRMetric(1) /
 * SSIMMetric(3, data_range=1) as used by `compute_metrics` (src/model.py:158-160, 215-220).
 * f32 NCDHW inputs; one result per batch item.
 *   err_sums: out[item] = {sum |a-b|, sum (a-b)^2} over per_item elements (f64);
 *             partials >= items * mi355_err_blocks(per_item) * 2 doubles.
 *   ssim3d  : Gaussian window (`window` = HOST array of `win` <= 15 normalised 1-D weights), "valid"
 *             windows, out[item] = mean SSIM over channels and window positions (f64).
 * ---------------------------------------------------------------------------------------- */
int32_t mi355_err_blocks(int64_t per_item);
int mi355_err_sums(const float* a, const float* b, int64_t per_item, int32_t items, double* partials,
                   double* out, void* stream);
int64_t mi355_ssim3d_workspace_bytes(int32_t items, int32_t c, int32_t d, int32_t h, int32_t w, int32_t win);
int mi355_ssim3d(const float* x, const float* y, int32_t items, int32_t c, int32_t d, int32_t h, int32_t w,
                 int32_t win, const float* window, float c1, float c2, void* workspace,
                 int64_t workspace_bytes, double* out, void* stream);

/* ------------------------------------------------------------------------------------------
 * Intensity augmentations of the training transform (src/data_module.py:130-139), GPU side:
 * tio.RandomBiasField (x * exp(polynomial field), `coefficients` = HOST array of
 * (order+1)(order+2)(order+3)/6 values in TorchIO's x-y-z loop order, order <= 4), tio.RandomGamma
 * (sign(x) |x|^gamma) and tio.RandomNoise (x + N(mean, std^2), counter-based noise from `seed`).
 * f32, (C, D, H, W) contiguous; in-place allowed (out == x).  Random PARAMETERS are drawn by the caller.
 * ---------------------------------------------------------------------------------------- */
int mi355_aug_bias_field(const float* x, float* out, int32_t c, int32_t d, int32_t h, int32_t w,
                         const float* coefficients, int32_t order, void* stream);
int mi355_aug_gamma(const float* x, float* out, int64_t count, float gamma, void* stream);
int mi355_aug_noise(const float* x, float* out, int64_t count, float mean, float std, uint64_t seed, void* stream);

/* layout probe used by the tests: writes lane -> (row, col) maps of the MFMA accumulators */
int mi355_mfma_selftest(float* out_f32_1024, float* out_bf16_1024, void* stream);

/* ------------------------------------------------------------------------------------------
 * fp8 operand preparation (per-tensor scaling, OCP e4m3): amax = max |x| over a tensor (the call zeroes *amax first),
 * cast = e4m3(x * 224 / amax) (scale 1 if amax == 0), saturating at +-448.
 *   mi355_amax_f32 : contiguous f32 array (weights)
 *   mi355_amax_act : NDHWC activation rows (bf16 or f32), c channels of ld
 *   mi355_cast_fp8 : activation rows -> one byte per channel, row stride ld_dst bytes
 * mi355_fp8_selftest: D = A B on v_mfma_scale_f32_32x32x64_f8f6f4 with exact small-integer e4m3 operands.
 * ---------------------------------------------------------------------------------------- */
int mi355_amax_f32(const float* x, int64_t n, float* amax, void* stream);
int mi355_amax_act(const void* x, int32_t ld, int32_t c, int64_t rows, int32_t dtype, float* amax, void* stream);
int mi355_cast_fp8(const void* src, int32_t ld_src, int32_t c, int64_t rows, int32_t src_dtype, const float* amax,
                   void* dst, int32_t ld_dst, void* stream);
/* Delayed scaling (the producer of an operand cannot know its amax before it has written it): the cast uses amax[0], the
 * amax gathered during the PREVIOUS training step, and raises amax_next[0] (may be NULL) to max |src|; values beyond
 * 2 * amax saturate at +-448.  mi355_fp8_scale_roll: table[n][2] = (amax in use, amax being gathered); once per step
 * in-use = gathered where gathered > 0, gathered = 0.  The first step of a layer uses mi355_amax_act + mi355_cast_fp8.
 * sat (int32[n] or NULL): sat[i] += 1 when the step that ends here saturated in slot i (gathered > 2 * in-use: some value
 * clamped at +-448) -- an overflow record without counters in the cast kernels. */
int mi355_cast_fp8_delayed(const void* src, int32_t ld_src, int32_t c, int64_t rows, int32_t src_dtype, const float* amax,
                           float* amax_next, void* dst, int32_t ld_dst, void* stream);
int mi355_fp8_scale_roll(float* table, int32_t n, int32_t* sat, void* stream);
int mi355_fp8_selftest(float* out_1024, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MI355_UNET_H */
