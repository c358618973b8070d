__all__ = ["batches"]
from . import batches  # noqa: F401,E402
