"""`from MoeNet_lite2 import Net` of the reference (python/MoeNet_lite2.py:22) -> the engine-backed class."""
from .models import Net  # noqa: F401
