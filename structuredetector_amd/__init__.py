"""structuredetector_amd -- MI355X (gfx950) implementation of the SDNet hot path.

Host-side mirror of the reference's Python interface for the path named by BASELINE.json
(`Network`, `Loss`, `Encode`, `Decoder` and the tensor primitives of ``sdnet.utils``), backed by
hand-written HIP kernels in ``csrc/libsdnet_hip.so`` reached through a C ABI
(``include/sdnet_hip.h``).  There is no CPU or eager fallback: every op raises if the library
is missing or the tensors are not on the GPU.
"""
__version__ = "0.1.0"
