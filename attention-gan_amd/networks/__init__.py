__all__ = ["attention", "generator", "generator_submodules", "discriminators"]
from . import attention, discriminators, generator, generator_submodules  # noqa: F401,E402
