__all__ = ["words_loss", "sentence_loss", "disc_loss", "gen_loss", "KL_loss"]
from . import KL_loss, disc_loss, gen_loss, sentence_loss, words_loss  # noqa: F401,E402
