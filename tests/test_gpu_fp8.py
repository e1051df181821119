"""fp8 convolution path (BASELINE.json configs[4]: "fp8 MFMA implicit-GEMM conv path"): OCP e4m3 operands with per-tensor
scales on v_mfma_scale_f32_32x32x64_f8f6f4 for forward and data gradient of the 3x3x3 layers with 32 input channels at full
resolution, f32 accumulation / statistics / master weights, bf16 storage everywhere else.

Oracle: the reference's arithmetic is f32 (src/train.py:33); for the KERNEL the yard-stick is the same convolution on
operands rounded through e4m3 on the CPU (torch.float8_e4m3fn, same per-tensor scales): products of two e4m3 values are
exact in f32, so the only differences left are f32 summation order and the bf16 rounding of the output -- the same bounds as
the bf16 op tests.  For the NETWORK the bound is triangulated: fp8 may move the generator output by what its 3-bit mantissa
implies (measured and stated below), and 20 training steps must track the bf16 mode's losses."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import unet_ref as R

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def q8(t):
    """per-tensor scaled e4m3 round trip on the CPU: what the kernel's operand preparation does"""
    amax = float(t.abs().max())
    s = 224.0 / amax if amax > 0 else 1.0
    return (t * s).to(torch.float8_e4m3fn).float() / s


def test_fp8_mfma_selftest(hip):
    from unet_bssfp_amd import ops
    got = ops.fp8_selftest(DEV).cpu()
    i = torch.arange(32).view(32, 1, 1)
    j = torch.arange(32).view(1, 32, 1)
    k = torch.arange(64).view(1, 1, 64)
    ref = ((((i + k) % 5) - 2) * (((2 * k + j) % 7) - 3)).sum(2).float()
    assert torch.equal(got, ref)


def test_fp8_cast_matches_torch_e4m3(hip):
    from unet_bssfp_amd import ops
    g = torch.Generator().manual_seed(1)
    x = (torch.randn(1, 4, 6, 8, 32, generator=g) * 3).to(torch.bfloat16)
    xd = x.to(DEV)
    amax = ops.amax_act(xd)
    assert float(amax) == float(x.float().abs().max())
    got = ops.cast_fp8(xd, amax).cpu().view(torch.float8_e4m3fn).float()
    ref = (x.float() * (224.0 / float(amax))).to(torch.float8_e4m3fn).float()
    assert torch.equal(got, ref)


@pytest.mark.parametrize("cout,sp,n", [(32, (32, 64, 128), 1), (96, (20, 48, 96), 1), (32, (7, 30, 40), 2)])
def test_fp8_conv_fwd_dgrad_vs_e4m3_operands(hip, cout, sp, n):
    from tests.test_gpu_ops import close, close_f32_sum, from_act, to_act
    from unet_bssfp_amd import functional as Fn
    from unet_bssfp_amd.nn import Conv3d
    g = torch.Generator().manual_seed(3)
    torch.manual_seed(1)
    layer = Conv3d(32, cout, 3, 1, 1)
    layer.fp8 = True
    x = (torch.rand(n, 32, *sp, generator=g) - 0.3).to(torch.bfloat16).float()
    w_cpu = layer.weight.detach().clone()
    z_ref = F.conv3d(q8(x), q8(w_cpu), layer.bias.detach(), 1, 1)
    gz = (torch.rand(z_ref.shape, generator=g) - 0.5).to(torch.bfloat16).float()
    # the data gradient of a 32-channel layer is itself a 32-input-channel convolution (e4m3); with 96 output channels it
    # has 96 input channels, which the marching kernel does not take: it runs on bf16 operands
    dgrad8 = cout == 32
    dx_ref = torch.nn.grad.conv3d_input(x.shape, q8(w_cpu) if dgrad8 else w_cpu.to(torch.bfloat16).float(),
                                        q8(gz) if dgrad8 else gz, stride=1, padding=1)
    layer = layer.to(DEV)
    a = to_act(x, torch.bfloat16).requires_grad_(True)
    plans = []
    from unet_bssfp_amd import ops
    ops.CONV_PROBE = lambda pid, d, real: plans.append((pid, d.dtype))
    try:
        z, part = Fn.ConvFn.apply(a, None, layer.weight, layer.bias, layer.spec, True, False, 0, True)
        z.backward(to_act(gz, torch.bfloat16))
    finally:
        ops.CONV_PROBE = None
    assert plans[0][1] == 3 and len([p for p in plans if p[1] == 3]) == (2 if dgrad8 else 1), plans   # which calls ran in e4m3
    close(from_act(z, cout), z_ref, torch.bfloat16, "z")
    zc = z_ref - layer.bias.detach().cpu().view(1, -1, 1, 1, 1)
    s = part.sum(0).cpu()
    close_f32_sum(s[1, :cout], (zc * zc).sum((0, 2, 3, 4)), "sum (z-b)^2")
    e0 = (s[0, :cout] - zc.sum((0, 2, 3, 4))).abs()
    assert bool((e0 <= 2e-3 * ((zc.numel() / cout) * (zc * zc).sum((0, 2, 3, 4))).sqrt() + 1e-6).all()), e0.max()
    close(from_act(a.grad, 32), dx_ref, torch.bfloat16, "dx")
    # the weight gradient stays a bf16-operand / f32-output product: exact operands, f32 summation order only
    dw_ref = torch.nn.grad.conv3d_weight(x, w_cpu.shape, gz, stride=1, padding=1)
    close_f32_sum(layer.weight.grad.cpu(), dw_ref, "dw")


def test_fp8_generator_and_training_track_bf16(hip):
    """Network level.  (i) forward: per-voxel L1 distance of the fp8-mode generator output from the f32 oracle, against the
    bf16 mode's distance -- the unit round-off of e4m3 (2^-4) is 32x that of bf16 (2^-9), and only the four full-resolution
    3x3x3 layers with 32 input channels use it (their operands; outputs and every other layer stay bf16), so the bound is
    16x the bf16 distance (measured 7.3x: 3.4e-2 against 4.7e-3 mean absolute difference per voxel);
    (ii) 20 training steps at 64^3 in fp8 and in bf16 from the same initial weights: every loss within 5 % of the bf16
    run's, the L1 reconstruction loss decreasing in both."""
    import unet_bssfp_amd as M
    from unet_bssfp_amd.functional import DropoutState
    from unet_bssfp_amd.gan import bSSFPToDWITensorModel, synthetic_batch
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    torch.manual_seed(8)
    g = M.Generator("bssfp", dropout=0.0)
    ref = R.RefGenerator("bssfp", dropout=0.0).train()
    ref.load_state_dict(g.state_dict())
    x, _ = R.synthetic_batch(1, 64, seed=5)
    with torch.no_grad():
        y_ref = ref(x)
        g = g.to(DEV).train()
        y16 = M.set_compute_dtype(g, torch.bfloat16)(x.to(DEV)).cpu()
        y8 = M.set_compute_dtype(g, "fp8")(x.to(DEV)).cpu()
    d16, d8 = (y16 - y_ref).abs().mean().item(), (y8 - y_ref).abs().mean().item()
    assert not torch.equal(y8, y16)                                   # the e4m3 kernels did run
    assert d8 <= 16 * d16, (d8, d16)
    curves = {}
    batch = synthetic_batch(1, 64, seed=11, device=DEV)
    for mode in ("bf16", "fp8"):
        torch.manual_seed(2)
        DropoutState.reset()
        model = bSSFPToDWITensorModel("bssfp", gen=M.Generator("bssfp").to(DEV), discr=M.Discriminator("bssfp").to(DEV)).train()
        M.set_compute_dtype(model, mode)
        hist = []
        for i in range(20):
            model.training_step(batch, i)
            hist.append({k: float(v) for k, v in model.last_logs.items()})
        curves[mode] = hist
    for i in range(20):
        for k, v in curves["bf16"][i].items():
            assert np.isfinite(curves["fp8"][i][k])
            if "adversarial" in k or "discr" in k:
                continue                                              # GAN equilibrium terms wander in any arithmetic
            assert abs(curves["fp8"][i][k] - v) <= 0.05 * abs(v), (i, k, curves["fp8"][i][k], v)
    for mode in curves:
        assert curves[mode][-1]["train_gen_loss_recon_L1"] < curves[mode][0]["train_gen_loss_recon_L1"]
    try:
        with open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "fp8_curves.log"), "w") as fh:
            fh.write(f"forward L1 vs f32 oracle: bf16 {d16:.3e} fp8 {d8:.3e}\n")
            for i in range(20):
                fh.write(f"step {i}: " + " ".join(f"{k}={curves['bf16'][i][k]:.5f}/{curves['fp8'][i][k]:.5f}" for k in curves['bf16'][i]) + "\n")
    except OSError:
        pass


def test_fp8_gan_step_at_config5_size_160(hip):
    """BASELINE.json configs[4] itself: one full GAN training step on a 1 x 24 x 160^3 volume with
    ``set_compute_dtype(model, "fp8")`` (src/train.py:33 is the reference's precision knob).  Size-independent properties:
    finite losses and parameters, every used parameter moved by the two AdamW updates, a bit-identical rerun from the same
    state; step-0 losses within 5e-3 (discriminator loss 2e-2) of a bf16-mode run of the same step -- the bounds
    test_bf16_gan_step_at_config3_size_vs_f32_mode uses between bf16 and f32; and the e4m3 plan really was taken: the
    CONV_PROBE records operand type 3 (e4m3) on the full-resolution 3x3x3 layers with 32 input channels (forward of
    conv_0.conv_1 and upcat_1.conv_1 in both generator passes; the data gradients whose incoming gradient has 32 channels)."""
    import copy
    import unet_bssfp_amd as M
    from unet_bssfp_amd import ops
    from unet_bssfp_amd.ddp import used_parameters
    from unet_bssfp_amd.gan import bSSFPToDWITensorModel, synthetic_batch
    torch.manual_seed(0)
    gen, discr = M.Generator("bssfp", dropout=0.0), M.Discriminator("bssfp")
    state = copy.deepcopy((gen.state_dict(), discr.state_dict()))
    batch = synthetic_batch(1, 160, seed=77, device=DEV)
    logs, params, plans = {}, {}, []
    from unet_bssfp_amd.functional import Fp8Scales as _Scales
    sat0 = _Scales.saturated_steps(DEV)
    for mode in ("bf16", "fp8", "fp8_again"):
        g, d = M.Generator("bssfp", dropout=0.0), M.Discriminator("bssfp")
        g.load_state_dict(state[0])
        d.load_state_dict(state[1])
        model = bSSFPToDWITensorModel("bssfp", gen=g.to(DEV), discr=d.to(DEV)).train()
        M.set_compute_dtype(model, "bf16" if mode == "bf16" else "fp8")
        before = [p.detach().clone() for p in used_parameters(model.gen, "bssfp") + used_parameters(model.discr, "bssfp")]
        if mode == "fp8":
            ops.CONV_PROBE = lambda pid, dd, real: plans.append((pid, int(dd.dtype), int(dd.c0) + int(dd.c1), int(dd.cstore), int(dd.di))) and None
        try:
            model.training_step(batch, 0)
        finally:
            ops.CONV_PROBE = None
        torch.cuda.synchronize()
        logs[mode] = {k: float(v) for k, v in model.last_logs.items()}
        if mode == "fp8":
            after = used_parameters(model.gen, "bssfp") + used_parameters(model.discr, "bssfp")
            assert all(not torch.equal(a, b) for a, b in zip(before, after)), "a used parameter did not move"
        if mode != "bf16":
            # a second step: the delayed-scaling path (previous step's amax, e4m3 copies written by the norm kernels)
            model.training_step(batch, 1)
            torch.cuda.synchronize()
            logs[mode + "_step1"] = {k: float(v) for k, v in model.last_logs.items()}
            assert all(np.isfinite(v) for v in logs[mode + "_step1"].values())
        params[mode] = [p.detach().clone() for p in model.parameters()]
        del model, g, d
        torch.cuda.empty_cache()
    e4m3 = [p for p in plans if p[1] == 3]
    # every e4m3 launch is a 3x3x3 layer with 32 stored input channels on the marching plan (32041 = conv_march_kernel) ...
    assert e4m3 and all(p[2] == 32 and p[0] == 32041 for p in e4m3), e4m3
    # ... at full resolution: two generator forward passes x {conv_0.conv_0 (24 -> 32, stored as 32), conv_0.conv_1,
    # upcat_1.conv_1} and the data gradients of conv_0.conv_1, upcat_1.conv_0 (32 -> 96), upcat_1.conv_1, conv_0.conv_0's
    # consumer chain; plus down_1.conv_0 (32 -> 64 at 80^3) -- and NO 3x3x3 layer with one 32-channel source ran in bf16
    assert sum(1 for p in e4m3 if p[4] == 160) >= 8, e4m3
    k3_c32 = [p for p in plans if p[0] // 10000 == 3 and p[2] == 32 and p[0] in (32041, 31941, 31942)]
    # (except the skip part of upcat_1's fused up-branch, Fn.UpCatConvFn: its 32 -> 32 partial convolution -- f32 output, two
    #  generator passes -- and that part's data gradient stay on bf16 operands: three launches per step)
    assert len(k3_c32) - len(e4m3) == 3, (len(e4m3), len(k3_c32))
    for k, ref in logs["bf16"].items():
        got = logs["fp8"][k]
        assert np.isfinite(got)
        assert abs(got - ref) <= (5e-3 if "discr" not in k else 2e-2) * abs(ref), (k, got, ref)
    assert logs["fp8"] == logs["fp8_again"] and logs["fp8_step1"] == logs["fp8_again_step1"]
    # delayed scaling clamps what exceeds twice the previous step's amax: the overflow record must be empty here
    from unet_bssfp_amd.functional import Fp8Scales
    assert Fp8Scales.saturated_steps(DEV) == sat0, (Fp8Scales.saturated_steps(DEV), sat0)
    assert abs(logs["fp8_step1"]["train_gen_loss_recon_L1"] - logs["fp8"]["train_gen_loss_recon_L1"]) < 0.1 * logs["fp8"]["train_gen_loss_recon_L1"]
    assert all(torch.equal(a, b) for a, b in zip(params["fp8"], params["fp8_again"]))
    assert all(torch.isfinite(p).all() for p in params["fp8"])


def test_fp8_producer_side_copies_equal_the_cast_of_the_stored_tensor(hip):
    """Delayed scaling, kernel level: the e4m3 copy that normact_fwd / normact_bwd write next to their bf16 result must be
    byte for byte what mi355_cast_fp8 makes of that bf16 tensor with the same amax, and the amax they gather must be the
    one mi355_amax_act finds (bit-exact: max is order-independent); the one-pass cast gathers the same; the roll kernel
    moves gathered -> in use only where something was gathered."""
    from unet_bssfp_amd import ops
    g = torch.Generator().manual_seed(12)
    n, d, h, w, c = 2, 9, 20, 24, 32
    z = (torch.randn(n, d, h, w, c, generator=g) * 3).to(torch.bfloat16).to(DEV)
    da = torch.randn(n, d, h, w, c, generator=g).to(torch.bfloat16).to(DEV)
    gamma = (torch.rand(c, generator=g) + 0.5).to(DEV)
    beta = (torch.rand(c, generator=g) - 0.5).to(DEV)
    part, bpg = ops.channel_stats(z, n)
    mean, rstd = ops.norm_finalize(part, bpg, n, c, d * h * w, None, 1e-5)
    for p, seed in ((0.0, 0), (0.1, 77)):
        table = torch.tensor([[1.7, 0.0], [0.031, 0.0]], dtype=torch.float32, device=DEV)   # amax in use: neither is the true amax
        a8 = torch.empty((n, d, h, w, c), dtype=torch.uint8, device=DEV)
        a = ops.normact_fwd(z, n, mean, rstd, gamma, beta, 0.1, p, seed, q8=(a8, table[0, 0:1], table[0, 1:2]))
        assert torch.equal(a, ops.normact_fwd(z, n, mean, rstd, gamma, beta, 0.1, p, seed))            # the bf16 result is untouched
        assert torch.equal(a8, ops.cast_fp8(a, table[0, 0:1]))
        assert float(table[0, 1]) == float(ops.amax_act(a)) > 1.7                                     # (some values saturate)
        dz8 = torch.empty((n, d, h, w, c), dtype=torch.uint8, device=DEV)
        dz, dg, db = ops.normact_bwd(z, da, n, mean, rstd, gamma, beta, 0.1, p, seed, True, True, q8=(dz8, table[1, 0:1], table[1, 1:2]))
        dz_plain, dg2, db2 = ops.normact_bwd(z, da, n, mean, rstd, gamma, beta, 0.1, p, seed, True, True)
        assert torch.equal(dz, dz_plain) and torch.equal(dg, dg2) and torch.equal(db, db2)
        assert torch.equal(dz8, ops.cast_fp8(dz, table[1, 0:1]))
        assert float(table[1, 1]) == float(ops.amax_act(dz))
        nxt = torch.zeros(1, device=DEV)
        assert torch.equal(ops.cast_fp8(a, table[0, 0:1], nxt), a8) and float(nxt) == float(table[0, 1])
        gathered = table[:, 1].clone()
        table2 = torch.cat([table, torch.tensor([[0.5, 0.0]], device=DEV)])                        # a slot nobody touched
        ops.fp8_scale_roll(table2, 3)
        assert torch.equal(table2[:2, 0], gathered) and float(table2[2, 0]) == 0.5 and float(table2[:, 1].abs().max()) == 0.0
    with pytest.raises(RuntimeError, match="e4m3 copy"):                                              # 64 channels: not offered
        z64 = torch.zeros(1, 4, 4, 4, 64, dtype=torch.bfloat16, device=DEV)
        ops.normact_fwd(z64, 1, None, None, None, None, 0.1, q8=(torch.empty(1, 4, 4, 4, 64, dtype=torch.uint8, device=DEV),
                                                                table[0, 0:1], table[0, 1:2]))


def test_fp8_delayed_scaling_steps_producer_side_equals_cast_side_and_graph_replay(hip):
    """Delayed scaling, step level.  From the second training step on the e4m3 operands are scaled with the previous
    step's amax; the copies come from the producing norm kernels (forward activations, incoming gradients).  (i) Four
    training steps with the producer-side copies must equal, bit for bit, four steps in which every operand takes the
    one-pass cast instead (same bytes, same gathered amax) -- and the producer side really ran: fewer cast launches;
    (ii) the step captured in a hipGraph (scale roll, copies and gathering on the device) replays to the same parameters;
    (iii) every slot the step used is primed and holds a scale."""
    import unet_bssfp_amd as M
    from unet_bssfp_amd import functional as Fn, ops
    from unet_bssfp_amd.functional import DropoutState, Fp8Scales
    from unet_bssfp_amd.gan import GraphedTrainingStep, bSSFPToDWITensorModel, synthetic_batch

    def build():
        torch.manual_seed(6)
        DropoutState.reset()
        gen, discr = M.Generator("bssfp", dropout=0.05), M.Discriminator("bssfp")
        model = bSSFPToDWITensorModel("bssfp", gen=gen.to(DEV), discr=discr.to(DEV)).train()
        return M.set_compute_dtype(model, "fp8")

    batch = synthetic_batch(1, 64, seed=21, device=DEV)
    casts = {}
    real_cast = ops.cast_fp8
    results = {}
    try:
        for mode in ("producer", "cast"):
            Fp8Scales.producer_side = mode == "producer"
            count = [0]

            def counting(*a, **k):
                count[0] += 1
                return real_cast(*a, **k)
            ops.cast_fp8 = counting
            model = build()
            for i in range(4):
                model.training_step(batch, i)
            torch.cuda.synchronize()
            casts[mode] = count[0]
            results[mode] = [p.detach().clone() for p in model.parameters()]
            slots = [s for m in model.modules() if hasattr(m, "spec") for s in getattr(m.spec, "_fp8_slots", {}).values()]
            assert slots and all(s.primed or s.touched for s in slots)
            assert all(float(s.use) > 0 for s in slots)
    finally:
        ops.cast_fp8 = real_cast
        Fp8Scales.producer_side = True
    assert casts["producer"] < casts["cast"], casts
    for p, q in zip(results["producer"], results["cast"]):
        assert torch.equal(p, q)
    graphed = build()
    gs = GraphedTrainingStep(graphed, batch, warmup=2)
    gs()
    gs()
    torch.cuda.synchronize()
    for (name, q), p in zip(graphed.named_parameters(), results["producer"]):
        assert torch.equal(p, q), name


def test_fp8_delayed_scaling_records_saturation(hip):
    """delayed scaling saturates silently (values beyond twice the previous step's amax clamp at +-448).  Every cast
    kernel gathers the step's amax anyway, so the roll that ends a step knows: gathered > 2 x in-use <=> something clamped; it
    raises the slot's record (Fp8Scales.Slot.sat, summed by Fp8Scales.saturated_steps)."""
    from unet_bssfp_amd import ops
    from unet_bssfp_amd.functional import Fp8Scales
    dev = torch.device(DEV)
    before = Fp8Scales.saturated_steps(dev)
    slot = Fp8Scales.slot(dev)
    x = torch.randn(1, 8, 8, 32, 32, device=DEV).to(torch.bfloat16)
    # step 1: no history -> in-step amax (cannot saturate)
    ops.amax_act(x, out=slot.use)
    x8 = ops.cast_fp8(x, slot.use, slot.next)
    slot.touched = True
    Fp8Scales.advance(dev)
    assert int(slot.sat) == 0 and float(slot.use) > 0
    # step 2: the tensor grew 1.5x: inside the head-room of delayed scaling (e4m3 holds 448, the scale maps amax to 224)
    x8 = ops.cast_fp8(x * 1.5, slot.use, slot.next)
    slot.touched = True
    Fp8Scales.advance(dev)
    assert int(slot.sat) == 0
    # step 3: it grew 3x against the amax in use: values clamp, and the roll says so
    x8 = ops.cast_fp8(x * 4.5, slot.use, slot.next)
    assert int(x8.view(torch.uint8).max()) >= 0x7E                    # 448 = 0x7E in e4m3 (sign bit aside)
    slot.touched = True
    Fp8Scales.advance(dev)
    assert int(slot.sat) == 1
    assert Fp8Scales.saturated_steps(dev) == before + 1
