#!/usr/bin/env python3
"""Per-workgroup phase timeline of the LDS-DMA GEMM kernel (trace build ab/lib_trace.so copied over the library first): shader-clock
stamps of thread 0 at entry / operands described / every chunk's barrier / loop left / hand-off done / stores issued.
    cp ab/lib_trace.so jorldy_amd/csrc/libjorldy_hip.so; python tools/probes/tgemm_trace.py M N K [cfg] [a_kc b_kc epi]"""
import ctypes as C
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from jorldy_amd import ops, _lib as L

M, N, K = (int(x) for x in sys.argv[1:4])
cfg = sys.argv[4] if len(sys.argv) > 4 else ""
akc, bkc, epi = (int(x) for x in sys.argv[5:8]) if len(sys.argv) > 7 else (1, 1, 2)
lib = L.load()
lib.jh_tgemm_trace_buffer.restype = C.c_int
lib.jh_tgemm_trace_buffer.argtypes = [C.c_void_p]
A = torch.randn((M, K) if akc else (K, M), device="cuda")
B = torch.randn((N, K) if bkc else (K, N), device="cuda")
bias = torch.randn(N, device="cuda")
aux = torch.randn(M, N, device="cuda")
buf = torch.zeros(8192 * 32, dtype=torch.int64, device="cuda")
ops.tgemm_set_cfg(cfg)
for _ in range(3):
    ops.tgemm_dense(A, B, a_kcont=bool(akc), b_kcont=bool(bkc), epi=epi, bias=bias, aux=aux, M=M, N=N, K=K)
torch.cuda.synchronize()
L.check(lib.jh_tgemm_trace_buffer(C.c_void_p(buf.data_ptr())))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
ops.tgemm_dense(A, B, a_kcont=bool(akc), b_kcont=bool(bkc), epi=epi, bias=bias, aux=aux, M=M, N=N, K=K)
e1.record()
torch.cuda.synchronize()
L.check(lib.jh_tgemm_trace_buffer(None))
t = buf.cpu().numpy().reshape(-1, 32)
wgs = int((t[:, 0] != 0).sum())
t = t[:wgs].astype(np.float64)
base = t[:, 0].min()
print(f"M{M} N{N} K{K} cfg {cfg!r} layout {akc}{bkc} epi {epi}: {wgs} workgroups, event time {e0.elapsed_time(e1) * 1e3:.1f} us (clock ticks below; 100 ticks = 1 us if the counter is the 100 MHz one)")
rel = t - base
nchunk = int((t[0, 2:26] != 0).sum())
def col(i): return rel[:, i]
print("entry    : min %.0f  median %.0f  max %.0f" % (col(0).min(), np.median(col(0)), col(0).max()))
print("described: +%.0f (median since entry)" % np.median(t[:, 1] - t[:, 0]))
print("chunk 0 barrier: +%.0f since described" % np.median(t[:, 2] - t[:, 1]))
if nchunk > 1:
    d = np.diff(t[:, 2:2 + nchunk], axis=1)
    print("chunk-to-chunk: median %.0f  p10 %.0f  p90 %.0f   per chunk medians: %s" % (np.median(d), np.percentile(d, 10), np.percentile(d, 90), " ".join("%.0f" % x for x in np.median(d, axis=0))))
print("loop left: +%.0f since last chunk barrier" % np.median(t[:, 28] - t[:, 2 + nchunk - 1]))
print("hand-off : +%.0f" % np.median(t[:, 29] - t[:, 28]))
print("stores   : +%.0f" % np.median(t[:, 30] - t[:, 29]))
print("exit     : min %.0f  median %.0f  max %.0f" % (col(30).min(), np.median(col(30)), col(30).max()))
print("per-WG total median %.0f" % np.median(t[:, 30] - t[:, 0]))
