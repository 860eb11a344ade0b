"""Drop-in for the reference's pybind11 module `main` (src/main.cpp:11-61): `main.RiiCpp` is the only name the
reference's Python layer uses (rii/rii.py:1,37).  `from rii_amd import main` (or the one-line shim of INTEGRATION.md §2)
puts the MI355X engine behind the unchanged `rii.Rii`."""
from .core import RiiGpu as RiiCpp

__all__ = ["RiiCpp"]
