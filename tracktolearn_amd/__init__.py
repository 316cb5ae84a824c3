"""tracktolearn_amd -- MI355X-native tractography environment step.

A from-scratch implementation of TrackToLearn's vectorised environment
(reset / step / harvest / get_streamlines) as hand-written HIP kernels for
gfx950 behind a C ABI (include/ttl_hip.h, libttl_hip.so), driven by Python host
classes that keep the reference's class surface.
"""
__version__ = '0.1.0'
