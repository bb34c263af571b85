"""dagr_amd -- MI355X-native engine for DAGR's event-graph hot path.

Host-side mirror (Python, like the reference) of the reference's operator interface for the path
events -> spatio-temporal graph -> SplineConv stack + voxel pooling -> detection-head maps, over the
C ABI of ``dagr_amd/lib/libdagr_hip.so`` (include/dagr_hip.h).  Sub-packages follow the reference's
layout (``graph``, ``model.layers``, ``model.networks``, ``utils``).
"""
__version__ = "0.1.0"
