"""Thin, autograd-free wrappers around the C ABI.  Tensors are NDHWC activations:
shape (N, D, H, W, C), stride(-1) == 1, row stride ``ld`` = stride(-2) (>= C, lets a tensor be a
channel slice of a wider buffer), dtype float32 (parity mode) or bfloat16 (throughput mode).
PyTorch is used for device memory and streams only."""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence, Tuple

import torch

from . import _lib
from ._lib import DT_BF16, DT_F32, DT_FP8

_DT = {torch.float32: DT_F32, torch.bfloat16: DT_BF16}


def round_up(v: int, m: int) -> int:
    return (v + m - 1) // m * m


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def require_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise _lib.Mi355Error("unet_bssfp_amd runs on the GPU only (no CPU fallback): got a CPU tensor")


def is_act(t: torch.Tensor) -> bool:
    """True if t is a valid NDHWC activation view."""
    if t.dim() != 5 or t.dtype not in _DT:
        return False
    n, d, h, w, c = t.shape
    st = t.stride()
    if c % 16 != 0 or st[4] != 1:
        return False
    ld = act_ld(t)
    epv = 4 if t.dtype == torch.float32 else 8
    if ld < c or ld % epv != 0 or t.data_ptr() % 16 != 0:
        return False
    exp = (d * h * w * ld, h * w * ld, w * ld, ld)
    dims = (n, d, h, w)
    return all(dims[i] == 1 or st[i] == exp[i] for i in range(4))


def act_ld(t: torch.Tensor) -> int:
    n, d, h, w, c = t.shape
    st = t.stride()
    if w > 1:
        return st[3]
    if h > 1:
        return st[2]
    if d > 1:
        return st[1]
    if n > 1:
        return st[0]
    return c


def as_act(t: torch.Tensor) -> torch.Tensor:
    return t if is_act(t) else t.contiguous()


def new_act(n, d, h, w, c, dtype, device) -> torch.Tensor:
    return torch.empty((n, d, h, w, c), dtype=dtype, device=device)


# ------------------------------------------------------------------------------ pack / unpack
def pack_ncdhw(src: torch.Tensor, dst: torch.Tensor, coff: int, zero_to: int):
    """dst[..., coff:coff+C] = src (NCDHW f32 contiguous); dst[..., coff+C:zero_to] = 0."""
    require_cuda(src, dst)
    assert src.dtype == torch.float32 and src.is_contiguous() and src.dim() == 5
    n, c = src.shape[:2]
    v = src.shape[2] * src.shape[3] * src.shape[4]
    lib = _lib.load()
    _lib.check(lib.mi355_pack_ncdhw(src.data_ptr(), dst.data_ptr(), n, c, v, act_ld(dst), coff, zero_to,
                                    _DT[dst.dtype], _stream()), "pack_ncdhw")


def unpack_ncdhw(src: torch.Tensor, c: int, coff: int = 0) -> torch.Tensor:
    require_cuda(src)
    n, d, h, w, _ = src.shape
    out = torch.empty((n, c, d, h, w), dtype=torch.float32, device=src.device)
    lib = _lib.load()
    _lib.check(lib.mi355_unpack_ncdhw(src.data_ptr(), out.data_ptr(), n, c, d * h * w, act_ld(src), coff,
                                      _DT[src.dtype], _stream()), "unpack_ncdhw")
    return out


def s2d_shape(n, d, h, w, cblk):
    """Shape of the space-to-depth tensor S(a) of a plain (n,d,h,w,cblk) activation."""
    return (n, d // 2 + 1, h // 2 + 1, w // 2 + 1, 8 * cblk)


def pack_ncdhw_s2d(src: torch.Tensor, dst: torch.Tensor, cblk: int, coff: int, zero_to: int):
    """Like pack_ncdhw, but dst is S(a): every plain voxel goes to its (cell, block); the out-of-volume blocks of the
    border cells get zeros in the written channel range (dst needs no prior zero-fill once all channels are packed)."""
    require_cuda(src, dst)
    assert src.dtype == torch.float32 and src.is_contiguous() and src.dim() == 5
    n, c, d, h, w = src.shape
    _lib.check(_lib.load().mi355_pack_ncdhw_s2d(src.data_ptr(), dst.data_ptr(), n, c, d, h, w, cblk, act_ld(dst), coff,
                                                zero_to, _DT[dst.dtype], _stream()), "pack_ncdhw_s2d")


def pack2(src0: torch.Tensor, src1: torch.Tensor, dst: torch.Tensor, coff: int, zero_to: int, s2d_cblk: int = 0):
    """torch.cat([src0, src1], 1) packed in ONE pass (whole rows): channels coff.. <- src0 then src1, zeros up to
    zero_to; s2d_cblk > 0: dst is the space-to-depth tensor with that many channels per block (fully written)."""
    require_cuda(src0, src1, dst)
    for t in (src0, src1):
        assert t.dtype == torch.float32 and t.is_contiguous() and t.dim() == 5
    assert src0.shape[0] == src1.shape[0] and src0.shape[2:] == src1.shape[2:]
    n, c0, d, h, w = src0.shape
    c1 = src1.shape[1]
    lib = _lib.load()
    if s2d_cblk:
        _lib.check(lib.mi355_pack2_ncdhw_s2d(src0.data_ptr(), c0, src1.data_ptr(), c1, dst.data_ptr(), n, d, h, w, s2d_cblk,
                                             act_ld(dst), coff, zero_to, _DT[dst.dtype], _stream()), "pack2_ncdhw_s2d")
    else:
        _lib.check(lib.mi355_pack2_ncdhw(src0.data_ptr(), c0, src1.data_ptr(), c1, dst.data_ptr(), n, d * h * w, act_ld(dst),
                                         coff, zero_to, _DT[dst.dtype], _stream()), "pack2_ncdhw")


def unpack_ncdhw_s2d(src: torch.Tensor, c: int, dims, cblk: int, coff: int = 0) -> torch.Tensor:
    require_cuda(src)
    n = src.shape[0]
    d, h, w = dims
    out = torch.empty((n, c, d, h, w), dtype=torch.float32, device=src.device)
    _lib.check(_lib.load().mi355_unpack_ncdhw_s2d(src.data_ptr(), out.data_ptr(), n, c, d, h, w, cblk, act_ld(src), coff,
                                                  _DT[src.dtype], _stream()), "unpack_ncdhw_s2d")
    return out


# ------------------------------------------------------------------------------ weights
FP8 = "fp8"       # dtype marker of the e4m3 packings / operands (stored as torch.uint8)


def weight_pack(src: torch.Tensor, cout: int, cin: int, ks: int, s_co: int, s_ci: int,
                s_k: Sequence[int], tbase: Sequence[int], tstep: Sequence[int], dtype,
                cinp: Optional[int] = None, coutp: Optional[int] = None, s2d_mode: int = 0,
                s2d_cp: int = 0, reuse: Optional[torch.Tensor] = None, q_amax: Optional[torch.Tensor] = None,
                src_offset: int = 0) -> Tuple[torch.Tensor, int, int]:
    """Returns (packed [cinp/16][ks^3][coutp][16], coutp, cinp).  `reuse`: re-pack in place into an
    earlier result (keeps the buffer address stable: required for hipGraph replays).
    dtype ops.FP8: e4m3 bytes of w * 224 / q_amax[0] (q_amax: device f32[1], see amax_f32)."""
    require_cuda(src)
    assert src.dtype == torch.float32 and src.is_contiguous()
    coutp = round_up(cout, 32) if coutp is None else coutp
    cinp = round_up(cin, 16) if cinp is None else cinp
    shape = (cinp // 16, ks ** 3, coutp, 16)
    fp8 = dtype == FP8
    if fp8:
        assert q_amax is not None
        dtype = torch.uint8
    if reuse is not None and tuple(reuse.shape) == shape and reuse.dtype == dtype and reuse.device == src.device:
        dst = reuse
    else:
        dst = torch.empty(shape, dtype=dtype, device=src.device)
    d = _lib.WpackDesc()
    d.src, d.dst = src.data_ptr() + 4 * src_offset, dst.data_ptr()      # (src_offset: elements; a channel slice of the weight)
    d._src_off = 4 * src_offset
    d.cout, d.cin, d.coutp, d.cinp, d.ks = cout, cin, coutp, cinp, ks
    d.s_co, d.s_ci = s_co, s_ci
    d.s_k = (C.c_int64 * 3)(*s_k)
    d.tbase = (C.c_int32 * 3)(*tbase)
    d.tstep = (C.c_int32 * 3)(*tstep)
    d.dtype = DT_FP8 if fp8 else _DT[dtype]
    d.q_amax = _ptr(q_amax)
    d.s2d_mode, d.s2d_cp = s2d_mode, s2d_cp
    _lib.check(_lib.load().mi355_weight_pack(C.byref(d), _stream()), "weight_pack")
    LAST_WPACK_DESC[0] = d
    return dst, coutp, cinp


LAST_WPACK_DESC = [None]      # descriptor of the latest weight_pack (picked up by the layer cache for batched re-packs)


def weight_pack_multi(descs):
    """Re-run a list of earlier packings (same sources / destinations) in a few launches."""
    if not descs:
        return
    arr = (_lib.WpackDesc * len(descs))(*descs)
    _lib.check(_lib.load().mi355_weight_pack_multi(arr, len(descs), _stream()), "weight_pack_multi")


# ------------------------------------------------------------------------------ convolution
def _conv_desc(x0, x1, wp, coutp, bias, ks, stride, pad, out, grid, os, ooff, stats, cls_cout=0, fp8=None, addend=None,
               d2s=False, delta=None):
    d = _lib.ConvDesc()
    n, di, hi, wi, c0 = x0.shape
    d.x0, d.c0, d.ld0 = x0.data_ptr(), c0, act_ld(x0)
    if x1 is not None:
        assert x1.shape[:4] == x0.shape[:4] and x1.dtype == x0.dtype
        d.x1, d.c1, d.ld1 = x1.data_ptr(), x1.shape[4], act_ld(x1)
    else:
        d.x1, d.c1, d.ld1 = None, 0, 0
    d.n, d.di, d.hi, d.wi = n, di, hi, wi
    d.do_, d.ho, d.wo = grid
    d.ks, d.stride = ks, stride
    d.pad = (C.c_int32 * 3)(*pad)
    d.wp, d.coutp = wp.data_ptr(), coutp
    d.bias = _ptr(bias)
    d.nbias = bias.numel() if bias is not None else 0
    d.y, d.ldy, d.cstore = out.data_ptr(), act_ld(out), out.shape[4]
    d.dy, d.hy, d.wy = out.shape[1:4]
    d.os = os
    d.ooff = (C.c_int32 * 3)(*ooff)
    d.stats_part = _ptr(stats)
    if fp8 is not None:                     # (amax_x, amax_w): x0 / wp are e4m3 bytes, the output stays bf16
        d.dtype = DT_FP8
        d.q_amax_x, d.q_amax_w = fp8[0].data_ptr(), fp8[1].data_ptr()
    else:
        d.dtype = _DT[x0.dtype]
    d.cls_cout = cls_cout
    # marching k2 kernel only (PatchGAN on space-to-depth tensors): accumulator start values / f32 output
    d.y_f32 = 1 if (out.dtype == torch.float32 and x0.dtype != torch.float32) else 0
    if d2s:                                 # depth-to-space (transposed k4 s2 p1 convolution): `out` is the plain 2x tensor
        d.d2s = 1
        assert tuple(out.shape[1:4]) == tuple(2 * e for e in grid) and coutp % 256 == 0
        d.delta = _ptr(delta)
    if addend is not None:
        assert addend.dim() == 5 and tuple(addend.shape[1:4]) == tuple(out.shape[1:4]) and addend.stride(4) == 1
        assert n % addend.shape[0] == 0 and addend.shape[4] >= (coutp // 8 if d2s else coutp)
        if d2s and addend.dtype == torch.bfloat16:
            d.add_bf16 = 1
        else:
            assert addend.dtype == torch.float32
        d.addend, d.ld_add = addend.data_ptr(), act_ld(addend)
        d.add_n = addend.shape[0] if addend.shape[0] != n else 0      # grid sample i starts from addend sample i % add_n
    return d


def conv_num_tiles(x0, x1, wp, coutp, ks, stride, pad, out, grid, os=1, ooff=(0, 0, 0), fp8=None, addend=None, d2s=False) -> Tuple[int, int]:
    d = _conv_desc(x0, x1, wp, coutp, None, ks, stride, pad, out, grid, os, ooff, None, 0, fp8, addend, d2s)
    tiles, tps = C.c_int32(0), C.c_int32(0)
    _lib.check(_lib.load().mi355_conv_num_tiles(C.byref(d), C.byref(tiles), C.byref(tps)), "conv_num_tiles")
    return tiles.value, tps.value


def conv_k2_marches(n: int, s_extents, c_in: int, coutp: int) -> bool:
    """Would a bf16 dense k2 (padding 0) convolution of an (n, *s_extents, c_in) space-to-depth tensor to coutp channels run
    on conv_march2_kernel -- the plan that honours `addend` / an f32 output?  (Asked before a layer is split.)"""
    d = _lib.ConvDesc()
    d.x0, d.c0, d.ld0, d.n = 1, c_in, c_in, n                # (no launch: the planner only checks for non-null pointers)
    d.di, d.hi, d.wi = s_extents
    d.do_, d.ho, d.wo = (e - 1 for e in s_extents)
    d.dy, d.hy, d.wy = d.do_, d.ho, d.wo
    d.ks, d.stride, d.os = 2, 1, 1
    d.wp, d.coutp, d.y, d.ldy, d.cstore = 1, coutp, 1, coutp, coutp
    d.dtype = DT_BF16
    pid = _lib.load().mi355_conv_plan_id(C.byref(d))
    return pid > 0 and (pid % 10000) // 100 == 24          # 10000 ks + 1000 halo + 100 shape + ...: halo plan, shape 14


# Optional launch probe (bench.py): called as probe(plan_id, desc, (cin, cout) real GEMM extents) and returns None or a callable
# that is invoked right after the launch (used to bracket one kernel family with HIP events).
CONV_PROBE = None


def conv_fwd(x0, x1, wp, coutp, bias, ks, stride, pad, out, grid, os=1, ooff=(0, 0, 0), stats=None, real=None,
             cls_cout=0, fp8=None, addend=None, d2s=False, delta=None):
    """fp8 = (amax_x, amax_w): x0 and wp hold e4m3 bytes (cast_fp8 / weight_pack(dtype=FP8)), `out` is bf16.
    addend (f32, the output's geometry): z = conv(x) + addend + bias; an f32 `out` on bf16 operands keeps the sums unrounded
    (both: the marching k2 kernel, i.e. the PatchGAN's first block split into its x- and y-part)."""
    require_cuda(x0, x1, wp, bias, out, stats, addend, delta)
    if fp8 is None:
        assert (out.dtype == x0.dtype or out.dtype == torch.float32) and wp.dtype == x0.dtype
    else:
        assert x0.dtype == torch.uint8 and wp.dtype == torch.uint8 and out.dtype == torch.bfloat16 and x1 is None
    assert bias is None or (bias.dtype == torch.float32 and bias.is_contiguous())
    d = _conv_desc(x0, x1, wp, coutp, bias, ks, stride, pad, out, grid, os, ooff, stats, cls_cout, fp8, addend, d2s, delta)
    lib = _lib.load()
    need = lib.mi355_conv_workspace_bytes(C.byref(d))
    if need < 0:
        _lib.check(-1, "conv_workspace_bytes")
    ws = None
    if need > 0:                                     # split-K partial sums (low levels)
        ws = torch.empty((need // 4,), dtype=torch.float32, device=x0.device)
        d.workspace, d.workspace_bytes = ws.data_ptr(), need
    after = CONV_PROBE(lib.mi355_conv_plan_id(C.byref(d)), d, real) if CONV_PROBE is not None else None
    _lib.check(lib.mi355_conv_fwd(C.byref(d), _stream()), "conv_fwd")
    if after is not None:
        after()


WGRAD_PROBE = None      # tests: called as probe(plan kind, desc) before a weight-gradient launch (2 = marching kernel)


def conv_wgrad(x0, x1, g, grid, gs, goff, ks, stride, pad, dw, cout, cin, s_co, s_ci, s_k, tbase, tstep,
               accumulate=False, s2d_cp=0, g_cls_cout=0, dw_offset=0, n=None, defer=None):
    """dw (torch layout, f32) (+)= sum_p x[p*stride+tap-pad] * g[p*gs+goff].
    dw_offset: element offset into dw (a channel slice of a layer's weight); n: samples of the grid when x0 holds fewer
    (x sample = grid sample % x0.shape[0]: one input under several gradients).
    defer: a list -- only the slab-producing kernel is launched and (job, workspace) is appended; ``wgrad_reduce_multi``
    sums the slabs of many layers in one launch later (dw is NOT valid before that)."""
    require_cuda(x0, x1, g, dw)
    assert dw.dtype == torch.float32 and g.dtype == x0.dtype
    d = _lib.WgradDesc()
    xn, di, hi, wi, c0 = x0.shape
    n = xn if n is None else n
    d.xn = xn if n != xn else 0
    d.x0, d.c0, d.ld0 = x0.data_ptr(), c0, act_ld(x0)
    if x1 is not None:
        d.x1, d.c1, d.ld1 = x1.data_ptr(), x1.shape[4], act_ld(x1)
    else:
        d.x1, d.c1, d.ld1 = None, 0, 0
    d.n, d.di, d.hi, d.wi = n, di, hi, wi
    d.g, d.cg, d.ldg = g.data_ptr(), g.shape[4], act_ld(g)
    d.do_, d.ho, d.wo = grid
    d.gd, d.gh, d.gw = g.shape[1:4]
    d.gs = gs
    d.goff = (C.c_int32 * 3)(*goff)
    d.ks, d.stride = ks, stride
    d.pad = (C.c_int32 * 3)(*pad)
    d.dw, d.cout, d.cin = dw.data_ptr() + 4 * dw_offset, cout, cin
    d.s_co, d.s_ci = s_co, s_ci
    d.s_k = (C.c_int64 * 3)(*s_k)
    d.tbase = (C.c_int32 * 3)(*tbase)
    d.tstep = (C.c_int32 * 3)(*tstep)
    d.accumulate = 1 if accumulate else 0
    d.dtype = _DT[x0.dtype]
    d.s2d_cp = s2d_cp
    d.g_cls_cout = g_cls_cout
    lib = _lib.load()
    need = lib.mi355_conv_wgrad_workspace(C.byref(d))
    if need < 0:
        _lib.check(-1, "conv_wgrad_workspace")
    if WGRAD_PROBE is not None:
        WGRAD_PROBE(lib.mi355_conv_wgrad_plan_kind(C.byref(d)), d)
    ws = torch.empty((need // 4,), dtype=torch.float32, device=x0.device)
    d.workspace, d.workspace_bytes = ws.data_ptr(), need
    if defer is not None:
        job = _lib.WreduceJob()
        _lib.check(lib.mi355_conv_wgrad_partial(C.byref(d), C.byref(job), _stream()), "conv_wgrad_partial")
        defer.append((job, ws))
        return
    _lib.check(lib.mi355_conv_wgrad(C.byref(d), _stream()), "conv_wgrad")


def wgrad_reduce_multi(jobs):
    """jobs: the (job, workspace) pairs ``conv_wgrad(..., defer=list)`` appended; one launch per 16 of them."""
    if not jobs:
        return
    arr = (_lib.WreduceJob * len(jobs))(*[j for j, _ in jobs])
    _lib.check(_lib.load().mi355_wgrad_reduce_multi(arr, len(jobs), _stream()), "wgrad_reduce_multi")


# ------------------------------------------------------------------------------ UpCat's up-branch as one transposed convolution
def upcat_compose(wd: torch.Tensor, wc: torch.Tensor, bd: Optional[torch.Tensor], bc: Optional[torch.Tensor], ce: int, out=None):
    """(k4 f32 [cl][co][4][4][4], its bf16 d2s packing [cl/16][8][8 co][16], biasp f32 [co], delta f32 [27][co]) from the
    transposed convolution's weight wd [cl][cu][2][2][2] / bias bd and the concatenated convolution's weight wc
    [co][ce + cu][3][3][3] / bias bc (csrc/upcat.hip).  out: an earlier result, overwritten in place (stable addresses)."""
    require_cuda(wd, wc, bd, bc)
    cl, cu = wd.shape[:2]
    co = wc.shape[0]
    assert wc.shape[1] == ce + cu and wd.is_contiguous() and wc.is_contiguous() and wd.dtype == wc.dtype == torch.float32
    dev = wd.device
    if out is None:
        out = (torch.empty((cl, co, 4, 4, 4), dtype=torch.float32, device=dev),
               torch.empty((cl // 16, 8, 8 * co, 16), dtype=torch.bfloat16, device=dev),
               torch.empty((co,), dtype=torch.float32, device=dev), torch.empty((27, co), dtype=torch.float32, device=dev))
    k4, wp, biasp, delta = out
    _lib.check(_lib.load().mi355_upcat_compose(wd.data_ptr(), wc.data_ptr(), _ptr(bd), _ptr(bc), cl, cu, ce, co, k4.data_ptr(),
                                               wp.data_ptr(), biasp.data_ptr() if bd is not None else None,
                                               delta.data_ptr() if bd is not None else None, _stream()), "upcat_compose")
    return out


def upcat_chain(dk4, wd, wc, bd, esum, ce: int, dwd, dwc, dbd, accumulate: bool):
    """dwd, dwc[:, ce:] and dbd (+)= the chain rule through k4 = compose(wd, wc) plus the bias-path terms from the border sums."""
    require_cuda(dk4, wd, wc, bd, esum, dwd, dwc, dbd)
    cl, cu = wd.shape[:2]
    co = wc.shape[0]
    assert dwc.is_contiguous() and dwc.shape == wc.shape and dwd.is_contiguous() and dk4.is_contiguous()
    _lib.check(_lib.load().mi355_upcat_chain(dk4.data_ptr(), wd.data_ptr(), wc.data_ptr(), _ptr(bd), esum.data_ptr(), cl, cu, ce, co,
                                             dwd.data_ptr(), dwc.data_ptr(), _ptr(dbd), 1 if accumulate else 0, _stream()), "upcat_chain")


def border_sums(g: torch.Tensor, c: Optional[int] = None) -> torch.Tensor:
    """e[27][c]: sums of the activation g over the 26 border regions of its volume (region (all, all, all): 0)."""
    require_cuda(g)
    n, d, h, w, cp = g.shape
    c = cp if c is None else c
    lib = _lib.load()
    ws = torch.empty((lib.mi355_border_sums_workspace(n, d, c) // 4,), dtype=torch.float32, device=g.device)
    e = torch.empty((27, c), dtype=torch.float32, device=g.device)
    _lib.check(lib.mi355_border_sums(g.data_ptr(), act_ld(g), n, d, h, w, c, _DT[g.dtype], ws.data_ptr(), e.data_ptr(), _stream()), "border_sums")
    return e


def seam_grad(gy: Optional[torch.Tensor], gs: Optional[torch.Tensor], out: torch.Tensor, c: int, dims):
    """out (NDHWC activation gradient) = gy (f32 NCDHW, or None) + un-space-to-depth(gs) (or None); channels >= c zero."""
    require_cuda(gy, gs, out)
    n, d, h, w, cp = out.shape
    assert (d, h, w) == tuple(dims) and (gy is None or (gy.dtype == torch.float32 and gy.is_contiguous() and tuple(gy.shape) == (n, c, d, h, w)))
    cblk = gs.shape[4] // 8 if gs is not None else 0
    assert gs is None or (gs.dtype == out.dtype and tuple(gs.shape[:4]) == tuple(s2d_shape(n, d, h, w, cblk)[:4]))
    _lib.check(_lib.load().mi355_seam_grad(_ptr(gy), _ptr(gs), act_ld(gs) if gs is not None else 0, cblk, out.data_ptr(), act_ld(out), cp,
                                           n, c, d, h, w, _DT[out.dtype], _stream()), "seam_grad")
    return out


def s2d_repack(x: torch.Tensor, c: Optional[int] = None) -> torch.Tensor:
    """plain activation (n, d, h, w, cp) -> the space-to-depth tensor of its first c channels (n, d/2+1, h/2+1, w/2+1, 8c), same dtype."""
    require_cuda(x)
    n, d, h, w, cp = x.shape
    c = cp if c is None else c
    out = torch.empty(s2d_shape(n, d, h, w, c), dtype=x.dtype, device=x.device)
    _lib.check(_lib.load().mi355_s2d_repack(x.data_ptr(), act_ld(x), out.data_ptr(), act_ld(out), n, d, h, w, c, _DT[x.dtype], _stream()),
               "s2d_repack")
    return out


# ------------------------------------------------------------------------------ fp8 operands
def amax_f32(x: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """max |x| of a contiguous f32 tensor as a device f32[1] (per-tensor scale of the e4m3 weight packings)."""
    require_cuda(x)
    assert x.dtype == torch.float32 and x.is_contiguous()
    out = torch.empty((1,), dtype=torch.float32, device=x.device) if out is None else out
    _lib.check(_lib.load().mi355_amax_f32(x.data_ptr(), x.numel(), out.data_ptr(), _stream()), "amax_f32")
    return out


def amax_act(x: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    require_cuda(x, out)
    n, d, h, w, c = x.shape
    out = torch.empty((1,), dtype=torch.float32, device=x.device) if out is None else out
    _lib.check(_lib.load().mi355_amax_act(x.data_ptr(), act_ld(x), c, n * d * h * w, _DT[x.dtype], out.data_ptr(), _stream()), "amax_act")
    return out


def cast_fp8(x: torch.Tensor, amax: torch.Tensor, amax_next: Optional[torch.Tensor] = None) -> torch.Tensor:
    """NDHWC activation -> e4m3 bytes of x * 224 / amax (uint8 tensor of the same shape, one byte per channel).
    amax_next (delayed scaling): raised to max |x| in the same pass -- `amax` is then the previous step's."""
    require_cuda(x, amax, amax_next)
    n, d, h, w, c = x.shape
    out = torch.empty((n, d, h, w, c), dtype=torch.uint8, device=x.device)
    _lib.check(_lib.load().mi355_cast_fp8_delayed(x.data_ptr(), act_ld(x), c, n * d * h * w, _DT[x.dtype], amax.data_ptr(),
                                                  _ptr(amax_next), out.data_ptr(), c, _stream()), "cast_fp8")
    return out


def fp8_scale_roll(table: torch.Tensor, n: int, sat: Optional[torch.Tensor] = None):
    """table f32 [rows][2] = (amax in use, amax being gathered): the first n rows start a new training step.
    sat (int32 [rows]): += 1 per slot whose ending step saturated (a value clamped at +-448: gathered amax > 2 x the one in use)."""
    require_cuda(table, sat)
    assert table.dtype == torch.float32 and table.is_contiguous() and table.shape[1] == 2 and 0 < n <= table.shape[0]
    assert sat is None or (sat.dtype == torch.int32 and sat.is_contiguous() and sat.numel() >= n)
    _lib.check(_lib.load().mi355_fp8_scale_roll(table.data_ptr(), n, _ptr(sat), _stream()), "fp8_scale_roll")


def conv_fp8_layer_ok(n: int, d: int, h: int, w: int, c_in: int, c_out: int) -> bool:
    """conv_fp8_supported for a layer known by its shapes only (bf16 NDHWC tensors with the channel counts stored):
    lets the PRODUCER of the operand decide whether to write the e4m3 copy."""
    if c_in != 32:
        return False
    key = (n, d, h, w, c_in, c_out)
    hit = _FP8_OK.get(key)
    if hit is None:
        d_ = _lib.ConvDesc()
        d_.x0, d_.c0, d_.ld0 = 256, c_in, c_in
        d_.x1, d_.c1, d_.ld1 = None, 0, 0
        d_.n, d_.di, d_.hi, d_.wi = n, d, h, w
        d_.do_, d_.ho, d_.wo = d, h, w
        d_.ks, d_.stride = 3, 1
        d_.pad = (C.c_int32 * 3)(1, 1, 1)
        d_.wp, d_.coutp = 256, round_up(c_out, 32)
        co = round_up(c_out, 16)
        d_.y, d_.ldy, d_.cstore = 256, co, co
        d_.dy, d_.hy, d_.wy = d, h, w
        d_.os = 1
        d_.ooff = (C.c_int32 * 3)(0, 0, 0)
        d_.dtype = DT_FP8
        d_.q_amax_x = d_.q_amax_w = 256
        hit = _FP8_OK[key] = _lib.load().mi355_conv_plan_id(C.byref(d_)) > 0
    return hit


_FP8_OK = {}


def conv_fp8_supported(x0: torch.Tensor, coutp: int, out: torch.Tensor, grid) -> bool:
    """Does mi355_conv_fwd take this 3x3x3 stride-1 pad-1 layer with e4m3 operands?  (32 input channels in one source,
    plain output grid: the marching kernel's layers.)"""
    n, di, hi, wi, c0 = x0.shape
    if c0 != 32 or not x0.is_cuda:
        return False
    d = _lib.ConvDesc()
    d.x0, d.c0, d.ld0 = x0.data_ptr(), c0, c0
    d.x1, d.c1, d.ld1 = None, 0, 0
    d.n, d.di, d.hi, d.wi = n, di, hi, wi
    d.do_, d.ho, d.wo = grid
    d.ks, d.stride = 3, 1
    d.pad = (C.c_int32 * 3)(1, 1, 1)
    d.wp, d.coutp = x0.data_ptr(), coutp
    d.y, d.ldy, d.cstore = out.data_ptr(), act_ld(out), out.shape[4]
    d.dy, d.hy, d.wy = out.shape[1:4]
    d.os = 1
    d.ooff = (C.c_int32 * 3)(0, 0, 0)
    d.dtype = DT_FP8
    d.q_amax_x = d.q_amax_w = x0.data_ptr()
    return _lib.load().mi355_conv_plan_id(C.byref(d)) > 0


def fp8_selftest(device) -> torch.Tensor:
    out = torch.zeros(1024, dtype=torch.float32, device=device)
    _lib.check(_lib.load().mi355_fp8_selftest(out.data_ptr(), _stream()), "fp8_selftest")
    return out.view(32, 32)


# ------------------------------------------------------------------------------ statistics / norm
def channel_stats(x: torch.Tensor, groups: int) -> Tuple[torch.Tensor, int]:
    """Partial {sum, sumsq} per channel: returns (part [groups*blocks][2][C], blocks_per_group)."""
    require_cuda(x)
    n, d, h, w, c = x.shape
    rows = n * d * h * w
    assert rows % groups == 0
    rpg = rows // groups
    lib = _lib.load()
    bpg = lib.mi355_channel_stats_blocks(rpg)
    part = torch.empty((groups * bpg, 2, c), dtype=torch.float32, device=x.device)
    _lib.check(lib.mi355_channel_stats(x.data_ptr(), act_ld(x), c, rpg, groups, part.data_ptr(), bpg,
                                       _DT[x.dtype], _stream()), "channel_stats")
    return part, bpg


def norm_finalize(part, parts_per_group, groups, c, count, shift, eps, running_mean=None, running_var=None,
                  momentum=0.0, n_real=0, batches_tracked=None):
    """n_real: entries of `shift` / the running buffers when they are shorter than the padded channel count c."""
    mean = torch.empty((groups, c), dtype=torch.float32, device=part.device)
    rstd = torch.empty_like(mean)
    _lib.check(_lib.load().mi355_norm_finalize(part.data_ptr(), parts_per_group, groups, c, count, _ptr(shift), n_real,
                                               eps, mean.data_ptr(), rstd.data_ptr(), _ptr(running_mean),
                                               _ptr(running_var), momentum, _ptr(batches_tracked), _stream()), "norm_finalize")
    return mean, rstd


def colsum(x: torch.Tensor) -> torch.Tensor:
    """Per-channel sum over all rows (f32 [C])."""
    part, bpg = channel_stats(x, 1)
    c = x.shape[4]
    out = torch.empty((c,), dtype=torch.float32, device=x.device)
    _lib.check(_lib.load().mi355_colsum_finalize(part.data_ptr(), bpg, c, out.data_ptr(), _stream()), "colsum")
    return out


def colsum_into(x: torch.Tensor, out: torch.Tensor, accumulate: bool):
    """Per-channel sum over all rows written into (or added to) the first out.numel() channels of `out` (f32)."""
    require_cuda(x, out)
    assert out.dtype == torch.float32 and out.is_contiguous() and out.numel() <= x.shape[4]
    part, bpg = channel_stats(x, 1)
    _lib.check(_lib.load().mi355_colsum_finalize_into(part.data_ptr(), bpg, x.shape[4], out.data_ptr(), out.numel(),
                                                      1 if accumulate else 0, _stream()), "colsum_into")


def colsum_from_parts(part: torch.Tensor, offset: int, out: torch.Tensor, accumulate: bool = False):
    """out[j] (+)= sum over the rows of part[:, 0, offset + j]: per-channel sums from the fused statistics a convolution
    launch emitted ([rows][2][C] partial {sum, sum of squares}), channels offset .. offset + out.numel()."""
    require_cuda(part, out)
    rows, two, c = part.shape
    assert two == 2 and part.is_contiguous() and part.dtype == out.dtype == torch.float32 and out.is_contiguous()
    assert 0 <= offset and offset + out.numel() <= c
    _lib.check(_lib.load().mi355_colsum_finalize_from(part.data_ptr(), rows, c, offset, out.data_ptr(), out.numel(),
                                                      1 if accumulate else 0, _stream()), "colsum_from_parts")


def _normact_desc(z, groups, mean, rstd, gamma, beta, slope, drop_p, seed, seed_t=None):
    d = _lib.NormActDesc()
    n, dd, h, w, c = z.shape
    rows = n * dd * h * w
    d.z, d.ldz, d.c = z.data_ptr(), act_ld(z), c
    d.rows_per_group, d.groups = rows // groups, groups
    # the kernels read mean[g * c + ch] for every group g: a (1, c) row with groups > 1 would be an out-of-bounds read
    for t in (mean, rstd):
        if t is not None and t.numel() != groups * c:
            raise ValueError(f"norm statistics hold {t.numel()} values for {groups} group(s) x {c} channels")
    d.mean, d.rstd, d.gamma, d.beta = _ptr(mean), _ptr(rstd), _ptr(gamma), _ptr(beta)
    d.n_affine = gamma.numel() if gamma is not None else 0
    d.slope, d.drop_p, d.seed = slope, drop_p, seed
    d.seed_ptr = _ptr(seed_t)
    d.dtype = _DT[z.dtype]
    return d


# Optional launch probe (bench.py): probe(kind "fwd" | "bwd", real channels, rows, bytes per element) -> None or a callable
# invoked right after the launches (HIP events around the fused norm + act kernels).
NORM_PROBE = None


def _norm_probe(kind, z, gamma):
    if NORM_PROBE is None:
        return None
    n, dd, h, w, c = z.shape
    return NORM_PROBE(kind, gamma.numel() if gamma is not None else c, n * dd * h * w, z.element_size())


def _set_q8(d, q8):
    """q8 = (e4m3 tensor to write [uint8, shape of the result], amax in use f32[1], amax being gathered f32[1])"""
    t8, use, nxt = q8
    require_cuda(t8, use, nxt)
    assert t8.dtype == torch.uint8 and t8.is_contiguous() and use.dtype == nxt.dtype == torch.float32
    d.q8, d.ld8, d.q_use, d.q_next = t8.data_ptr(), t8.shape[-1], use.data_ptr(), nxt.data_ptr()


def normact_fwd(z, groups, mean, rstd, gamma, beta, slope, drop_p=0.0, seed=0, out=None, s2d=False, seed_t=None, q8=None,
                final=None, skip_a=False, pool=False):
    """s2d=True: `out` is the space-to-depth tensor S(a) (s2d_shape) instead of a plain one; every slot of it is written.
    q8: also write the e4m3 copy of the result for the fp8 convolution that consumes it (see _set_q8).
    final=(w, bias, y): also evaluate the 1x1x1 convolution that consumes a -- y[..., k] = bf16(a @ bf16(w[k]) + bias[k]), w f32
    (k <= 8, cin[, 1, 1, 1]), y a bf16 NDHWC tensor on the same grid whose channels beyond k are zeroed; skip_a: a itself is not
    written (the returned tensor is uninitialised).
    pool=True: also MaxPool3d(2) of the result in the same pass -> (a, y, idx) as maxpool2_fwd(a, want_idx=True) (even extents,
    plain layout, no q8 / final)."""
    require_cuda(z, mean, rstd, gamma, beta, out)
    if out is None:
        out = torch.empty(z.shape, dtype=z.dtype, device=z.device)
    d = _normact_desc(z, groups, mean, rstd, gamma, beta, slope, drop_p, seed, seed_t)
    d.a, d.lda = out.data_ptr(), act_ld(out)
    keep_final = None
    if final is not None:
        fw, fb, fy = final
        require_cuda(fw, fb, fy)
        assert not s2d and z.dtype == fy.dtype == torch.bfloat16 and fw.dtype == torch.float32 and fw.is_contiguous()
        fw2 = fw.detach().reshape(fw.shape[0], -1)
        assert fw2.shape[0] <= 8 and fw2.shape[1] <= z.shape[4] and tuple(fy.shape[:4]) == tuple(z.shape[:4])
        assert fb is None or (fb.dtype == torch.float32 and fb.numel() == fw2.shape[0])
        d.gw, d.gw_ld, d.gk = fw2.data_ptr(), fw2.shape[1], fw2.shape[0]
        d.fy, d.ldfy, d.fcp, d.fbias = fy.data_ptr(), act_ld(fy), fy.shape[4], _ptr(fb.detach() if fb is not None else None)
        d.skip_a = 1 if skip_a else 0
        keep_final = (fw2, fb, fy)
    else:
        assert not skip_a
    if q8 is not None:
        _set_q8(d, q8)
    if s2d:
        d.s2d_a = 1
        d.sd, d.sh, d.sw = z.shape[1:4]
    py = pidx = None
    if pool:
        n, dd, h, w, c = z.shape
        assert not s2d and q8 is None and final is None and dd % 2 == 0 and h % 2 == 0 and w % 2 == 0
        py = torch.empty((n, dd // 2, h // 2, w // 2, c), dtype=z.dtype, device=z.device)
        pidx = torch.empty(py.shape, dtype=torch.uint8, device=z.device)
        d.pool_y, d.ldpy, d.pool_widx = py.data_ptr(), act_ld(py), pidx.data_ptr()
        d.sd, d.sh, d.sw = dd, h, w
    after = _norm_probe("fwd", z, gamma)
    _lib.check(_lib.load().mi355_normact_fwd(C.byref(d), _stream()), "normact_fwd")
    if after is not None:
        after()
    return (out, py, pidx) if pool else out


def normact_bwd(z, da, groups, mean, rstd, gamma, beta, slope, drop_p, seed, batch_stats, want_affine_grads,
                s2d=False, seed_t=None, affine_into=None, accumulate=False, q8=None, implicit=None, pool=None):
    """Returns (dz, dgamma, dbeta).  dgamma/dbeta are f32 [C] (None if there is no norm).
    s2d=True: `da` is a gradient in space-to-depth layout (the forward wrote S(a)).
    affine_into=(dgamma, dbeta): write (accumulate=True: add) the affine gradients into these caller-owned f32
    vectors of the real channel count instead of returning new ones (then None, None are returned for them).
    implicit=(gz, gw): `da` is not materialised (pass None): it is the data gradient of the 1x1x1 convolution that consumed a,
    da = bf16(gz[..., :k] @ gw) with gz the NDHWC gradient of that convolution's output and gw its f32 weights (k, cin[,1,1,1]).
    pool=(idx, dy): `da` is the gradient of a's skip-connection use only (or None): the activation was also consumed by MaxPool3d(2)
    (idx from maxpool2_fwd(want_idx=True), dy the pooled tensor's gradient) and the kernels form maxpool2_bwd(..., add=da) per row."""
    require_cuda(z, da)
    lib = _lib.load()
    n, dd, h, w, c = z.shape
    rows = n * dd * h * w
    d = _normact_desc(z, groups, mean, rstd, gamma, beta, slope, drop_p, seed, seed_t)
    keep_implicit = None
    if implicit is not None:
        gz, gw = implicit
        require_cuda(gz, gw)
        assert da is None and not s2d and z.dtype == torch.bfloat16 and gz.dtype == torch.bfloat16
        assert tuple(gz.shape[:4]) == tuple(z.shape[:4]) and gw.dtype == torch.float32 and gw.is_contiguous()
        gw2 = gw.detach().reshape(gw.shape[0], -1)
        assert gw2.shape[0] <= 8 and gw2.shape[0] <= gz.shape[4] and gw2.shape[1] <= c
        d.gz, d.ldgz, d.gw, d.gw_ld, d.gk = gz.data_ptr(), act_ld(gz), gw2.data_ptr(), gw2.shape[1], gw2.shape[0]
        keep_implicit = (gz, gw2)
    elif pool is not None:
        pidx, pdy = pool
        require_cuda(pidx, pdy)
        assert not s2d and pidx.dtype == torch.uint8 and pidx.is_contiguous() and pdy.dtype == z.dtype
        assert tuple(pidx.shape) == (n, dd // 2, h // 2, w // 2, c) == tuple(pdy.shape) and dd % 2 == 0 and h % 2 == 0 and w % 2 == 0
        d.pool_idx, d.pool_dy, d.ldpdy = pidx.data_ptr(), pdy.data_ptr(), act_ld(pdy)
        d.sd, d.sh, d.sw = dd, h, w
        if da is not None:
            assert tuple(da.shape) == tuple(z.shape) and da.dtype == z.dtype
            d.da, d.ldda = da.data_ptr(), act_ld(da)
    else:
        d.da, d.ldda = da.data_ptr(), act_ld(da)
    dz = torch.empty(z.shape, dtype=z.dtype, device=z.device)
    d.dz, d.lddz = dz.data_ptr(), act_ld(dz)
    d.batch_stats = 1 if batch_stats else 0
    if s2d:
        d.s2d_da = 1
        d.sd, d.sh, d.sw = z.shape[1:4]
    if q8 is not None:
        _set_q8(d, q8)                      # e4m3 copy of dz for the fp8 data-gradient convolution
    dgamma = dbeta = None
    keep = []
    after = _norm_probe("bwd", z, gamma)
    if mean is not None and (batch_stats or want_affine_grads):
        bpg = lib.mi355_channel_stats_blocks(rows // groups)
        part = torch.empty((groups * bpg, 2, c), dtype=torch.float32, device=z.device)
        d.part, d.blocks_per_group = part.data_ptr(), bpg
        _lib.check(lib.mi355_normact_bwd_reduce(C.byref(d), _stream()), "normact_bwd_reduce")
        sums = torch.empty((groups, 2, c), dtype=torch.float32, device=z.device)
        if affine_into is not None:
            g_into, b_into = affine_into
            require_cuda(g_into, b_into)
            assert g_into.dtype == b_into.dtype == torch.float32 and g_into.numel() == b_into.numel() <= c
            _lib.check(lib.mi355_normact_bwd_finalize_into(part.data_ptr(), bpg, groups, c, sums.data_ptr(),
                                                           g_into.data_ptr(), b_into.data_ptr(), g_into.numel(),
                                                           1 if accumulate else 0, _stream()), "normact_bwd_finalize")
        else:
            dgamma = torch.empty((c,), dtype=torch.float32, device=z.device)
            dbeta = torch.empty((c,), dtype=torch.float32, device=z.device)
            _lib.check(lib.mi355_normact_bwd_finalize(part.data_ptr(), bpg, groups, c, sums.data_ptr(),
                                                      dgamma.data_ptr(), dbeta.data_ptr(), _stream()),
                       "normact_bwd_finalize")
        d.sums = sums.data_ptr()
        keep += [part, sums]
    _lib.check(lib.mi355_normact_bwd_apply(C.byref(d), _stream()), "normact_bwd_apply")
    if after is not None:
        after()
    return dz, dgamma, dbeta


# tensors of up to 512 K elements (1 MB bf16: 8^3 x 512, 16^3 x 128 ...) with >= 64 channels take the one-launch kernels.  Measured in
# the full step (interleaved A/B, bench.py --small-norm-elements): 1 M (16^3 x 256 included) 12.42 ms, 600 K 12.21, 300 K 12.23,
# never 12.33 -- one workgroup per 16 bytes of channels is too few workgroups for a 2 MB tensor
SMALL_NORM_ELEMENTS = 1 << 19


SMALL_NORM_ELEMENTS_GROUPED = SMALL_NORM_ELEMENTS      # BatchNorm over several statistic groups walked in order by one workgroup


def norm_is_small(n: int, d: int, h: int, w: int, c: int, serial_groups: int = 1) -> bool:
    """Does a (n, d, h, w, c) activation take the fused small-tensor norm kernels (mi355_normact_small_fwd / _bwd)?"""
    limit = SMALL_NORM_ELEMENTS if serial_groups <= 1 else SMALL_NORM_ELEMENTS_GROUPED
    return c >= 64 and n * d * h * w * c <= limit


def normact_small_fwd(z, groups, gamma, beta, eps, slope, drop_p=0.0, seed=0, seed_t=None, running_mean=None,
                      running_var=None, momentum=0.0, batches_tracked=None, out=None, s2d=False):
    """Statistics + norm + dropout + LeakyReLU of a small tensor in ONE launch.  Returns (a, mean, rstd); BatchNorm running
    statistics (given: training mode) are updated in place."""
    require_cuda(z, gamma, beta, out, running_mean, running_var)
    n, dd, h, w, c = z.shape
    if out is None:
        out = torch.empty(z.shape, dtype=z.dtype, device=z.device)
    mean = torch.empty((groups, c), dtype=torch.float32, device=z.device)
    rstd = torch.empty_like(mean)
    d = _lib.NormSmallDesc()
    d.base = _normact_desc(z, groups, None, None, gamma, beta, slope, drop_p, seed, seed_t)
    d.base.a, d.base.lda = out.data_ptr(), act_ld(out)
    if s2d:
        d.base.s2d_a = 1
        d.base.sd, d.base.sh, d.base.sw = z.shape[1:4]
    d.eps, d.momentum = eps, momentum
    d.mean_out, d.rstd_out = mean.data_ptr(), rstd.data_ptr()
    d.running_mean, d.running_var, d.batches_tracked = _ptr(running_mean), _ptr(running_var), _ptr(batches_tracked)
    d.n_real = running_mean.numel() if running_mean is not None else 0
    after = _norm_probe("fwd", z, gamma)
    _lib.check(_lib.load().mi355_normact_small_fwd(C.byref(d), _stream()), "normact_small_fwd")
    if after is not None:
        after()
    return out, mean, rstd


def normact_small_bwd(z, da, groups, mean, rstd, gamma, beta, slope, drop_p, seed, batch_stats, s2d=False, seed_t=None,
                      affine_into=None, accumulate=False, want_affine=True, group_scratch=None):
    """Backward of the above in ONE launch: returns (dz, dgamma, dbeta) -- the affine gradients are None when written into
    ``affine_into`` (caller-owned f32 vectors of the real channel count) or not wanted.  ``group_scratch`` (None: for three or
    more statistic groups): hand the kernel scratch for per-group sums, see below."""
    require_cuda(z, da, mean, rstd)
    n, dd, h, w, c = z.shape
    d = _lib.NormSmallDesc()
    d.base = _normact_desc(z, groups, mean, rstd, gamma, beta, slope, drop_p, seed, seed_t)
    d.base.da, d.base.ldda = da.data_ptr(), act_ld(da)
    dz = torch.empty(z.shape, dtype=z.dtype, device=z.device)
    d.base.dz, d.base.lddz = dz.data_ptr(), act_ld(dz)
    d.base.batch_stats = 1 if batch_stats else 0
    if s2d:
        d.base.s2d_da = 1
        d.base.sd, d.base.sh, d.base.sw = z.shape[1:4]
    dgamma = dbeta = None
    if affine_into is not None:
        g_into, b_into = affine_into
        require_cuda(g_into, b_into)
        assert g_into.dtype == b_into.dtype == torch.float32 and g_into.numel() == b_into.numel() <= c
        d.base.n_affine = g_into.numel()
        d.dgamma, d.dbeta, d.accumulate = g_into.data_ptr(), b_into.data_ptr(), 1 if accumulate else 0
    elif want_affine:
        dgamma = torch.zeros((c,), dtype=torch.float32, device=z.device)
        dbeta = torch.zeros((c,), dtype=torch.float32, device=z.device)
        d.dgamma, d.dbeta, d.accumulate = dgamma.data_ptr(), dbeta.data_ptr(), 0
    # three or more statistic groups (the reference's batch of 8 patches): scratch for the per-group sums, so that the chunks of
    # groups run as independent workgroups and a second tiny launch sums the affine gradients over the groups (same order, f64)
    gpart = None
    if (groups >= 3) if group_scratch is None else group_scratch:
        gpart = torch.empty((groups, 2, c), dtype=torch.float64, device=z.device)
        d.base.part = gpart.data_ptr()
    after = _norm_probe("bwd", z, gamma)
    _lib.check(_lib.load().mi355_normact_small_bwd(C.byref(d), _stream()), "normact_small_bwd")
    if after is not None:
        after()
    return dz, dgamma, dbeta


# ------------------------------------------------------------------------------ pooling
def maxpool2_fwd(x: torch.Tensor, want_idx: bool = False):
    """y = MaxPool3d(2)(x); want_idx: -> (y, idx) with idx (uint8, y's shape) the window position 4 kd + 2 kh + kw the backward
    pass routes each pooled element's gradient to (normact_bwd's pool= reads it instead of a max-pool backward launch)."""
    require_cuda(x)
    n, d, h, w, c = x.shape
    y = torch.empty((n, d // 2, h // 2, w // 2, c), dtype=x.dtype, device=x.device)
    if not want_idx:
        _lib.check(_lib.load().mi355_maxpool2_fwd(x.data_ptr(), act_ld(x), y.data_ptr(), act_ld(y), n, c, d, h, w,
                                                  _DT[x.dtype], _stream()), "maxpool2_fwd")
        return y
    idx = torch.empty(y.shape, dtype=torch.uint8, device=x.device)
    _lib.check(_lib.load().mi355_maxpool2_fwd_idx(x.data_ptr(), act_ld(x), y.data_ptr(), act_ld(y), idx.data_ptr(), n, c, d, h, w,
                                                  _DT[x.dtype], _stream()), "maxpool2_fwd_idx")
    return y, idx


def maxpool2_bwd(x, y, dy, add=None) -> torch.Tensor:
    """dx of MaxPool3d(2); `add` (same shape as x, possibly a channel slice of a wider buffer) is summed into it"""
    require_cuda(x, y, dy)
    n, d, h, w, c = x.shape
    dx = torch.empty((n, d, h, w, c), dtype=x.dtype, device=x.device)
    if add is None:
        _lib.check(_lib.load().mi355_maxpool2_bwd(x.data_ptr(), act_ld(x), y.data_ptr(), act_ld(y), dy.data_ptr(),
                                                  act_ld(dy), dx.data_ptr(), act_ld(dx), n, c, d, h, w,
                                                  _DT[x.dtype], _stream()), "maxpool2_bwd")
    else:
        require_cuda(add)
        assert add.shape == x.shape and add.dtype == x.dtype
        _lib.check(_lib.load().mi355_maxpool2_bwd_add(x.data_ptr(), act_ld(x), y.data_ptr(), act_ld(y), dy.data_ptr(),
                                                      act_ld(dy), dx.data_ptr(), act_ld(dx), add.data_ptr(), act_ld(add),
                                                      n, c, d, h, w, _DT[x.dtype], _stream()), "maxpool2_bwd_add")
    return dx


# ------------------------------------------------------------------------------ loss / optimiser
def l1_fwd(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    require_cuda(a, b)
    assert a.dtype == b.dtype == torch.float32 and a.is_contiguous() and b.is_contiguous() and a.shape == b.shape
    lib = _lib.load()
    cnt = a.numel()
    partials = torch.empty((lib.mi355_l1_blocks(cnt),), dtype=torch.float32, device=a.device)
    out = torch.empty((), dtype=torch.float32, device=a.device)
    _lib.check(lib.mi355_l1_fwd(a.data_ptr(), b.data_ptr(), cnt, partials.data_ptr(), out.data_ptr(), _stream()),
               "l1_fwd")
    return out


def l1_bwd(a, b, gscale: torch.Tensor) -> torch.Tensor:
    require_cuda(a, b, gscale)
    da = torch.empty_like(a)
    gs = gscale.to(torch.float32).reshape(1).contiguous()
    _lib.check(_lib.load().mi355_l1_bwd(a.data_ptr(), b.data_ptr(), a.numel(), gs.data_ptr(), da.data_ptr(),
                                        _stream()), "l1_bwd")
    return da


def gan_gen_loss_fwd(logits: torch.Tensor, y_hat: torch.Tensor, y: torch.Tensor, divisor: float, factor: float) -> torch.Tensor:
    """-> f32[4] = (L1(y_hat, y), L1 / divisor * factor, mean BCEWithLogits(logits, 1), their sum): src/model.py:126-137"""
    require_cuda(logits, y_hat, y)
    assert logits.dtype == y_hat.dtype == y.dtype == torch.float32 and logits.is_contiguous() and y_hat.is_contiguous() and y.is_contiguous()
    assert y_hat.shape == y.shape
    lib = _lib.load()
    cnt = y_hat.numel()
    nb = lib.mi355_l1_blocks(cnt)
    partials = torch.empty((nb,), dtype=torch.float32, device=y.device)
    out = torch.empty((4,), dtype=torch.float32, device=y.device)
    _lib.check(lib.mi355_l1_partials(y_hat.data_ptr(), y.data_ptr(), cnt, partials.data_ptr(), _stream()), "l1_partials")
    _lib.check(lib.mi355_gan_gen_loss_fwd(logits.data_ptr(), logits.numel(), partials.data_ptr(), nb, cnt, divisor, factor,
                                          out.data_ptr(), _stream()), "gan_gen_loss_fwd")
    return out


def gan_gen_loss_bwd(logits: torch.Tensor, upstream: torch.Tensor, divisor: float, factor: float):
    """-> (dlogits, the device scalar l1_bwd scales with)"""
    require_cuda(logits, upstream)
    up = upstream.to(torch.float32).reshape(1).contiguous()
    dlogits = torch.empty_like(logits)
    gscale = torch.empty((1,), dtype=torch.float32, device=logits.device)
    _lib.check(_lib.load().mi355_gan_gen_loss_bwd(logits.data_ptr(), logits.numel(), up.data_ptr(), divisor, factor,
                                                  dlogits.data_ptr(), gscale.data_ptr(), _stream()), "gan_gen_loss_bwd")
    return dlogits, gscale


def gan_discr_loss_fwd(fake: torch.Tensor, real: torch.Tensor) -> torch.Tensor:
    """(mean BCEWithLogits(real, 1) + mean BCEWithLogits(fake, 0)) / 2 as f32[1]: src/model.py:183-193"""
    require_cuda(fake, real)
    assert fake.dtype == real.dtype == torch.float32 and fake.is_contiguous() and real.is_contiguous()
    out = torch.empty((1,), dtype=torch.float32, device=fake.device)
    _lib.check(_lib.load().mi355_gan_discr_loss_fwd(fake.data_ptr(), fake.numel(), real.data_ptr(), real.numel(), out.data_ptr(),
                                                    _stream()), "gan_discr_loss_fwd")
    return out


def gan_discr_loss_bwd(fake: torch.Tensor, real: torch.Tensor, upstream: torch.Tensor, dfake: torch.Tensor, dreal: torch.Tensor):
    require_cuda(fake, real, upstream, dfake, dreal)
    up = upstream.to(torch.float32).reshape(1).contiguous()
    _lib.check(_lib.load().mi355_gan_discr_loss_bwd(fake.data_ptr(), fake.numel(), real.data_ptr(), real.numel(), up.data_ptr(),
                                                    dfake.data_ptr(), dreal.data_ptr(), _stream()), "gan_discr_loss_bwd")


def mfma_selftest(device) -> Tuple[torch.Tensor, torch.Tensor]:
    a = torch.zeros(1024, dtype=torch.float32, device=device)
    b = torch.zeros(1024, dtype=torch.float32, device=device)
    _lib.check(_lib.load().mi355_mfma_selftest(a.data_ptr(), b.data_ptr(), _stream()), "mfma_selftest")
    return a.view(32, 32), b.view(32, 32)
