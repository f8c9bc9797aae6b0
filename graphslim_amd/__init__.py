"""graphslim_amd — MI355X-native GCond gradient-matching engine behind GraphSlim's reduce()/GraphData API.

Only the GCond hot path is implemented; arithmetic runs in libgslim_hip.so (HIP, gfx950).
"""
__version__ = "0.1.0"
