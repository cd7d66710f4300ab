"""Importing this module loads libim2im_uq.so; every product module imports it first so a
missing HIP library is an ImportError at import time, never a silent CPU fallback."""
from .. import _lib  # noqa: F401
