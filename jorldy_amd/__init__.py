"""jorldy_amd -- MI355X (gfx950) native RL training hot path behind JORLDY's
core/agent + core/buffer API.  See DESIGN.md / INTEGRATION.md.

The compute path is libjorldy_hip.so (hand-written HIP kernels, C ABI in
include/jorldy_hip.h).  There is no CPU fallback: importing the sub-packages
that need the library raises if it has not been built.
"""
__version__ = "0.1.0"

import os as _os

# Kernel arguments in VRAM (host writes them through the BAR) instead of host memory the GPU has to
# fetch over PCIe at every kernel start: ~0.3 ms per PPO iteration on MI355X.  Only effective when
# set before the HIP runtime initialises (i.e. import jorldy_amd before the first torch.cuda call).
_os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
