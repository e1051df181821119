"""Lightning-shaped checkpoint round trips (format part)."""
import numpy as np
import pytest
import torch

from oracle import unet_ref as R
from unet_bssfp_amd import checkpoint as ck, gan


def _cpu_model(seed=0):
    torch.manual_seed(seed)
    return gan.bSSFPToDWITensorModel("bssfp", batch_size=1, gen=R.RefGenerator("bssfp", dropout=0.0),
                                     discr=R.RefDiscriminator("bssfp"), optimizer_class=torch.optim.AdamW)


def _batch(s=32, seed=3):
    x, y = R.synthetic_batch(2, s, seed=seed)          # N=2: BatchNorm needs >1 value per channel at 32^3
    return {"bssfp": {"data": x}, "dwi-tensor_orig": {"data": y}}


def test_cpu_roundtrip_resumes_bit_identically(tmp_path):
    a = _cpu_model()
    a.training_step(_batch())
    ck.save_checkpoint(a, tmp_path / "m.ckpt", epoch=3, global_step=17)
    a.training_step(_batch(seed=4))
    b = _cpu_model(seed=9)                                   # different init: everything must come from the file
    info = ck.load_checkpoint(b, tmp_path / "m.ckpt")
    assert info["epoch"] == 3 and info["global_step"] == 17 and info["ignored_keys"] == []
    b.training_step(_batch(seed=4))
    for (ka, va), (kb, vb) in zip(a.state_dict().items(), b.state_dict().items()):
        assert ka == kb and torch.equal(va, vb), ka


def test_file_layout_is_lightnings_and_loads_without_unpickling_code(tmp_path):
    m = _cpu_model()
    m.training_step(_batch())
    ck.save_checkpoint(m, tmp_path / "m.ckpt")
    raw = torch.load(tmp_path / "m.ckpt", weights_only=True)          # nothing but tensors and plain containers
    assert {"state_dict", "optimizer_states", "hyper_parameters", "epoch", "global_step", "pytorch-lightning_version"} <= set(raw)
    assert raw["hyper_parameters"]["input_modality"] == "bssfp" and len(raw["optimizer_states"]) == 2
    keys = set(raw["state_dict"])
    assert "gen.blocks.unet.conv_0.conv_0.conv.weight" in keys and "gen.blocks.bssfp.bn.running_mean" in keys
    assert "discr.d1.bssfp.conv.weight" in keys and "discr.blocks.bssfp.conv.weight" in keys      # the double registration
    assert "discr.d5.bn.num_batches_tracked" in keys and "gen.blocks.unet.upcat_1.upsample.deconv.weight" in keys


def test_reference_style_checkpoint_with_foreign_keys(tmp_path):
    """a reference run also stores the Perceptual network under recon_criterion.*: listed, not loaded"""
    m = _cpu_model()
    d = ck.checkpoint_dict(m)
    d["state_dict"]["recon_criterion.perceptual.net.conv1.weight"] = torch.zeros(4, 1, 3, 3, 3)
    d["optimizer_states"] = []
    torch.save(d, tmp_path / "ref.ckpt")
    b = _cpu_model(seed=5)
    info = ck.load_checkpoint(b, tmp_path / "ref.ckpt")
    assert info["ignored_keys"] == ["recon_criterion.perceptual.net.conv1.weight"]
    assert torch.equal(b.gen.state_dict()["blocks.unet.final_conv.weight"], m.gen.state_dict()["blocks.unet.final_conv.weight"])
    del d["state_dict"]["gen.blocks.unet.final_conv.bias"]
    torch.save(d, tmp_path / "bad.ckpt")
    with pytest.raises(RuntimeError):
        ck.load_checkpoint(_cpu_model(), tmp_path / "bad.ckpt")
    assert ck.load_checkpoint(_cpu_model(), tmp_path / "bad.ckpt", strict=False)["epoch"] == 0


@pytest.mark.gpu
def test_gpu_resume_is_bit_identical_and_crosses_to_the_oracle(tmp_path):
    import unet_bssfp_amd as M
    torch.manual_seed(0)
    a = gan.bSSFPToDWITensorModel("bssfp", batch_size=1).cuda()
    b1 = {k: {"data": v["data"].cuda()} for k, v in _batch().items()}
    b2 = {k: {"data": v["data"].cuda()} for k, v in _batch(seed=4).items()}
    a.training_step(b1)
    ck.save_checkpoint(a, tmp_path / "g.ckpt", epoch=1, global_step=1)
    a.training_step(b2)                                       # dropout 0.05 is on: the mask counter must resume too
    b, info = ck.load_from_checkpoint(tmp_path / "g.ckpt", device="cuda")
    assert info["global_step"] == 1 and b.input_modality == "bssfp"
    b.training_step(b2)
    for (ka, va), (kb, vb) in zip(a.state_dict().items(), b.state_dict().items()):
        assert ka == kb and torch.equal(va, vb), ka
    for oa, ob in zip(a.optimizers(), b.optimizers()):
        oa.sync_step_counts(), ob.sync_step_counts()
        assert [int(s["step"]) for s in oa.state.values()] == [int(s["step"]) for s in ob.state.values()]
    # the same file drives the CPU oracle modules (state_dict keys are the reference's)
    c = _cpu_model(seed=7)
    ck.load_checkpoint(c, tmp_path / "g.ckpt", load_optimizers=False)
    d, _ = ck.load_from_checkpoint(tmp_path / "g.ckpt", device="cuda")
    x = b1["bssfp"]["data"]
    d.eval(), c.eval()
    with torch.no_grad():
        assert (d.gen(x).cpu() - c.gen(x.cpu())).abs().mean().item() <= 1e-4
