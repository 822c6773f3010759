"""uzliti_slam_amd — MI355X (gfx950) back end for uzliti_slam's edge estimation + pose-graph solve.

The product is libuzl_mi355x.so (hand-written HIP kernels behind the C ABI in include/uzl_mi355x.h);
this package holds its sources (csrc/), the ctypes binding and the host-side mirror of the reference's
plugin interfaces.  There is no CPU fallback: without the built library every compute call raises.
"""
__all__ = ["synth"]
