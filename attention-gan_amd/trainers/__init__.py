__all__ = ["trainer"]
from . import trainer  # noqa: F401,E402
