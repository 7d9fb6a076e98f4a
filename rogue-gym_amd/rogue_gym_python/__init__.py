"""Drop-in for the reference's `rogue_gym_python` extension package (python/setup.py:57)."""
from . import _rogue_gym  # noqa: F401
