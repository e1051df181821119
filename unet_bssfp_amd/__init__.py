"""MI355X-native (gfx950 / CDNA4) 3D U-Net generator + PatchGAN discriminator training path.

Hand-written HIP kernels behind a C ABI (``include/mi355_unet.h``, ``libmi355_unet.so``), wrapped
as drop-in ``torch.nn.Module``s for the reference's construction sites (SomeUserName1/UNet-bSSFP,
src/model.py).  GPU only: the package raises if the library is missing -- there is no CPU path.
"""
from . import _lib  # noqa: F401
from .nn import (BasicUNet, Conv3d, ConvTranspose3d, Discriminator, DownSampleConv, Generator,  # noqa: F401
                 compute_dtype_from_name, set_compute_dtype, set_default_compute_dtype)
from .functional import l1_loss  # noqa: F401

__all__ = ["BasicUNet", "Conv3d", "ConvTranspose3d", "Discriminator", "DownSampleConv", "Generator",
           "compute_dtype_from_name", "set_compute_dtype", "set_default_compute_dtype", "l1_loss"]
