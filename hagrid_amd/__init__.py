"""hagrid_amd -- MI355X (gfx950) implementation of Hagrid's irregular-grid build + traversal hot path.

Importing the package is cheap and needs no GPU; the compiled library (hagrid_amd/libhagrid_amd.so) is
loaded on first use of hagrid_amd.api and there is no CPU fallback.
"""
__version__ = "0.1.0"
