"""Lightning-shaped checkpoints for the step harness -- SURVEY.md 8(f) rank 4 (format part).

The reference trains under ``pl.Trainer`` with ``ModelCheckpoint`` (src/train.py:21-27) and resumes with
``bSSFPToDWITensorModel.load_from_checkpoint(ckpt_path)`` (src/train.py:56-57).  A Lightning ``.ckpt`` is a
``torch.save``'d dict; the parts that matter here:

    'state_dict'        module state, keys prefixed by the LightningModule attribute: ``gen.*`` / ``discr.*``
                        (key table: SURVEY.md 8(b)); a reference checkpoint also carries
                        ``recon_criterion.*`` (MedicalNet weights of the Perceptual term) -- ignored here
    'optimizer_states'  [gen AdamW state_dict, discr AdamW state_dict]   (configure_optimizers order, :359-361)
    'hyper_parameters'  what ``save_hyperparameters`` recorded (src/model.py:149): input_modality, lr,
                        batch_size, perceptual_factor, recon_factor
    'epoch', 'global_step', 'pytorch-lightning_version', 'lr_schedulers', 'callbacks', 'loops'

Files are read with ``torch.load(weights_only=True)`` only: nothing in the file is executed, and a
checkpoint whose pickle needs more than tensors and plain containers is refused by torch itself.
"""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import torch

LIGHTNING_VERSION = "2.2.1"          # requirements.txt:1 of the reference
HPARAM_KEYS = ("input_modality", "lr", "batch_size", "perceptual_factor", "recon_factor")
_OWN_PREFIXES = ("gen.", "discr.")


def _plain(v):
    """tensors to CPU, containers rebuilt from plain types: safe for weights_only loading"""
    if isinstance(v, torch.Tensor):
        return v.detach().cpu()
    if isinstance(v, dict):
        return {k: _plain(x) for k, x in v.items()}
    if isinstance(v, (list, tuple)):
        return [_plain(x) for x in v]
    return v


def checkpoint_dict(model, epoch: int = 0, global_step: int = 0) -> Dict:
    opts = model.optimizers()
    for o in opts:
        if hasattr(o, "sync_step_counts"):
            o.sync_step_counts()                     # device-side step counters -> state['step']
    extra = {}
    if next(model.gen.parameters()).is_cuda:
        from .functional import DropoutState
        dev = next(model.gen.parameters()).device
        extra["dropout_base"] = int(DropoutState.base(dev).item())
    return {
        "epoch": int(epoch), "global_step": int(global_step), "pytorch-lightning_version": LIGHTNING_VERSION,
        "state_dict": _plain({k: v for k, v in model.state_dict().items() if k.startswith(_OWN_PREFIXES)}),
        "optimizer_states": [_plain(o.state_dict()) for o in opts],
        "lr_schedulers": [], "callbacks": {}, "loops": {},
        "hyper_parameters": {k: getattr(model, k) for k in HPARAM_KEYS},
        "mi355": extra,
    }


def save_checkpoint(model, path: str, epoch: int = 0, global_step: int = 0) -> None:
    torch.save(checkpoint_dict(model, epoch, global_step), path)


def load_checkpoint(model, path: str, strict: bool = True, load_optimizers: bool = True) -> Dict:
    """Load a checkpoint written by ``save_checkpoint`` or by the reference's Lightning run into ``model``.
    Returns {'epoch', 'global_step', 'ignored_keys'}; keys outside gen./discr. (the reference's
    ``recon_criterion.*`` etc.) are listed, not loaded."""
    ckpt = torch.load(path, map_location="cpu", weights_only=True)
    if "state_dict" not in ckpt:
        raise KeyError(f"{path}: not a Lightning-shaped checkpoint (no 'state_dict')")
    sd = ckpt["state_dict"]
    own = {k: v for k, v in sd.items() if k.startswith(_OWN_PREFIXES)}
    ignored = sorted(k for k in sd if not k.startswith(_OWN_PREFIXES))
    target = {k for k in model.state_dict() if k.startswith(_OWN_PREFIXES)}
    missing, unexpected = sorted(target - set(own)), sorted(set(own) - target)
    if strict and (missing or unexpected):
        raise RuntimeError(f"checkpoint does not match the model: missing {missing[:5]}... unexpected {unexpected[:5]}...")
    model.load_state_dict(own, strict=False)
    if next(model.gen.parameters()).is_cuda:
        from .functional import repack_weights
        repack_weights(model.gen), repack_weights(model.discr)
    if load_optimizers and ckpt.get("optimizer_states"):
        states = ckpt["optimizer_states"]
        opts = model.optimizers()
        if len(states) != len(opts):
            raise RuntimeError(f"checkpoint has {len(states)} optimizer states, the model has {len(opts)} optimizers")
        for o, st in zip(opts, states):
            o.load_state_dict(st)
    base = (ckpt.get("mi355") or {}).get("dropout_base")
    if base is not None and next(model.gen.parameters()).is_cuda:
        from .functional import DropoutState
        DropoutState.base(next(model.gen.parameters()).device).fill_(int(base))
    return {"epoch": ckpt.get("epoch", 0), "global_step": ckpt.get("global_step", 0), "ignored_keys": ignored}


def load_from_checkpoint(path: str, device: Optional[str] = None, **overrides):
    """``bSSFPToDWITensorModel.load_from_checkpoint`` (src/train.py:57): rebuild the model from the stored
    hyper-parameters, then load weights and both optimizer states."""
    from .gan import bSSFPToDWITensorModel
    hp = dict(torch.load(path, map_location="cpu", weights_only=True).get("hyper_parameters") or {})
    hp = {k: v for k, v in hp.items() if k in HPARAM_KEYS}
    hp.update(overrides)
    if "input_modality" not in hp:
        raise KeyError(f"{path}: no 'input_modality' among the hyper-parameters; pass it explicitly")
    model = bSSFPToDWITensorModel(**hp)
    if device is not None:
        model = model.to(device)
    info = load_checkpoint(model, path)
    return model, info
