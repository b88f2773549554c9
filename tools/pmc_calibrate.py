#!/usr/bin/env python3
"""Streams a 1 GiB device buffer (past the 256 MiB Infinity Cache) with 4- / 8- / 16-byte loads per lane (jh_calib_stream), 5 launches each:
run under `rocprofv3 --pmc <COUNTER> --kernel-trace` (tools/pmc_calibrate.sh) to learn how many bytes one counter unit stands for per width."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from jorldy_amd import _lib as L

lib = L.load()
n = 1 << 30
src = torch.empty(n // 4, dtype=torch.float32, device="cuda").uniform_(0, 1)
out = torch.zeros(4, dtype=torch.float32, device="cuda")
for width in (4, 8, 16):
    for _ in range(5):
        L.check(lib.jh_calib_stream(L.ctx(0), L.ptr(src), n, width, L.ptr(out), L.stream_ptr()))
torch.cuda.synchronize()
print("streamed", n, "bytes x 5 per width")
