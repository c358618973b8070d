"""attention-gan_amd: MI355X-native AttnGAN training hot path behind the reference's module API.

The directory name is not an identifier; import it with

    import importlib; agan = importlib.import_module("attention-gan_amd")

`agan.install_as_reference_namespace()` then registers `networks`, `losses`, `utilities` and `trainers` in
sys.modules so that the reference's own import lines (`from networks.generator import Generator`, ...) resolve here.
"""
import sys as _sys

from . import backend, losses, networks, trainers, utilities  # noqa: F401
from .backend import lib  # noqa: F401
from .backend.functional import get_precision, set_precision  # noqa: F401

__version__ = "0.1.0"


def install_as_reference_namespace() -> None:
    import importlib
    for pkg in ("networks", "losses", "utilities", "trainers"):
        mod = importlib.import_module(f"{__name__}.{pkg}")
        _sys.modules[pkg] = mod
        for sub in getattr(mod, "__all__", ()):
            _sys.modules[f"{pkg}.{sub}"] = importlib.import_module(f"{__name__}.{pkg}.{sub}")
