"""Developer helper (never imported by the product): point the ctypes binding at another build of the same C ABI
before the library is first loaded, e.g. the diagnostic build made by tools/build_diag.sh
(tools/_build/libmi355_unet_diag.so: plan overrides through MI355_CONV_* environment variables).

    import tools.diaglib as D; D.use("tools/_build/libmi355_unet_diag.so")

bench.py --lib <path> goes through here and reports the non-default library in its JSON line."""
import os


def use(path: str) -> str:
    from unet_bssfp_amd import _lib
    path = os.path.abspath(path)
    if not os.path.exists(path):
        raise FileNotFoundError(path)
    if _lib._lib is not None:
        raise RuntimeError("the library is already loaded: call tools.diaglib.use() first")
    _lib.LIB_PATH = path
    return path
