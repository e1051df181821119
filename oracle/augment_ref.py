"""CPU ORACLE -- TEST INFRASTRUCTURE ONLY.  Intensity augmentations, SURVEY.md 8(f) rank 4 (augmentation part).

The reference composes TorchIO transforms (/root/reference/src/data_module.py:130-139); TorchIO (0.19.5) is absent,
so the three that are built on the GPU are restated here from its published algorithm -- PARITY UNPINNED:

* RandomBiasField(coefficients=0.5, order=3): coefficients ~ U(-0.5, 0.5), one per monomial x^i y^j z^k with
  i + j + k <= order (x outer loop, then y, then z); coordinates np.arange(-n/2, n/2) + 0.5 per axis divided by
  their maximum; image *= exp(field).
* RandomGamma(log_gamma=(-0.3, 0.3)): gamma = exp(U(-0.3, 0.3)); sign(x) |x|^gamma.
* RandomNoise(mean=0, std=(0.01, 0.1) in the reference): x + N(mean, std^2) -- only its statistics can be tested.
"""
import numpy as np


def n_coefficients(order: int) -> int:
    return (order + 1) * (order + 2) * (order + 3) // 6


def bias_field(shape, coefficients, order=3):
    ranges = []
    for n in shape:
        r = np.arange(-n / 2, n / 2) + 0.5
        m = r.max()
        ranges.append(r / m if m > 0 else r)
    x, y, z = np.meshgrid(*ranges, indexing="ij")
    f = np.zeros(shape)
    i = 0
    for xo in range(order + 1):
        for yo in range(order + 1 - xo):
            for zo in range(order + 1 - (xo + yo)):
                f += coefficients[i] * x ** xo * y ** yo * z ** zo
                i += 1
    return np.exp(f).astype(np.float32)


def apply_bias_field(x, coefficients, order=3):
    return x * bias_field(x.shape[1:], coefficients, order)[None]


def apply_gamma(x, gamma):
    return np.sign(x) * np.abs(x) ** gamma
