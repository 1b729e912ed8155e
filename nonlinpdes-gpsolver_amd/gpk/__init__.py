"""gpk -- ctypes binding of libgpk.so (include/gpk.h): the MI355X Gram-assembly + Gauss-Newton library.

There is no CPU fallback here: importing works anywhere (so the host-side classes can be inspected), but creating a
`Context` without the compiled library or without a gfx950 device raises `GpkError`.
"""
from ._lib import GpkError, load_library, library_path, declared_symbols
from .device import Context, DeviceArray, GNProblem, LAYOUT, KERNEL, NUGGET, SYSTEM

__all__ = ['GpkError', 'load_library', 'library_path', 'declared_symbols', 'Context', 'DeviceArray', 'GNProblem',
           'LAYOUT', 'KERNEL', 'NUGGET', 'SYSTEM']
