"""jorldy_amd -- MI355X (gfx950) native RL training hot path behind JORLDY's
core/agent + core/buffer API.  See DESIGN.md / INTEGRATION.md.

The compute path is libjorldy_hip.so (hand-written HIP kernels, C ABI in
include/jorldy_hip.h).  There is no CPU fallback: importing the sub-packages
that need the library raises if it has not been built.
"""
__version__ = "0.1.0"
