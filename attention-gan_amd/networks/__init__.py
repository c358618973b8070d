__all__ = ["attention", "generator", "generator_submodules", "discriminators", "cnn_encoder"]
from . import attention, cnn_encoder, discriminators, generator, generator_submodules  # noqa: F401,E402
