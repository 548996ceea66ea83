"""focalformer3d_amd - MI355X (gfx950) native Hard-Instance-Probing decoder path of FocalFormer3D.

Host side: Python mirror of the reference's mmdet3d_plugin head/decoder registry API.
Device side: hand-written HIP kernels behind the C ABI in include/ff3d.h (libff3d_hip.so).
"""
__version__ = '0.1.0'
