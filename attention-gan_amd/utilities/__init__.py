__all__ = ["layers"]
from . import layers  # noqa: F401,E402
