"""dragonfly_amd -- MI355X-native GP-surrogate + acquisition engine for Dragonfly's hot path.

Hand-written HIP (gfx950) behind a ctypes C-ABI (include/dfhip.h); the Python classes mirror
the reference's GP / kernel / fitter / acquisition surfaces.
"""
__version__ = '0.1.0'
