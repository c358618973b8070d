__all__ = ["attention", "generator", "generator_submodules", "discriminators", "cnn_encoder", "rnn_encoder"]
from . import attention, cnn_encoder, discriminators, generator, generator_submodules, rnn_encoder  # noqa: F401,E402
