"""Top-level package name of the reference's compiled Morton extension (`from MortonEncoding import _C`)."""
from . import _C  # noqa: F401
