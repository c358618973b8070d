__all__ = ["attention", "generator", "generator_submodules", "discriminators", "cnn_encoder", "rnn_encoder", "stage4"]
from . import attention, cnn_encoder, discriminators, generator, generator_submodules, rnn_encoder, stage4  # noqa: F401,E402
