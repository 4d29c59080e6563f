"""audiolab_amd -- MI355X-native Process->Separate path for AudioLab.

Only what the hot path needs lives here: csrc/ (HIP kernels + C ABI), the ctypes door
(_lib), the MDX runner (mdx), the TFC-TDF network object (tdfnet) and the host-side mirror of
the reference's plugin interface (wrappers/, engine).  Importing the package does not load
the GPU library; the first Context() does, and raises if libalsep.so is missing.
"""
__version__ = "0.1.0"
