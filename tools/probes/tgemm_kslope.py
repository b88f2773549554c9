#!/usr/bin/env python3
"""Fixed cost vs per-chunk cost of one tile-engine launch without instrumenting it: time C = A B^T (+ bias, relu) at M x N for several K under the
library's event timers (R back-to-back launches per event pair) and fit T(K) = fixed + per_chunk * K / 32.
    python tools/probes/tgemm_kslope.py M N [cfg] [a_kc b_kc epi]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from jorldy_amd import ops

M, N = int(sys.argv[1]), int(sys.argv[2])
cfg = sys.argv[3] if len(sys.argv) > 3 else ""
akc, bkc, epi = (int(x) for x in sys.argv[4:7]) if len(sys.argv) > 6 else (1, 1, 2)
ops.tgemm_set_cfg(cfg)
res = []
for K in (256, 512, 1024, 2048, 4096):
    A = torch.randn((M, K) if akc else (K, M), device="cuda")
    B = torch.randn((N, K) if bkc else (K, N), device="cuda")
    bias = torch.randn(N, device="cuda")
    aux = torch.randn(M, N, device="cuda")
    f = lambda: ops.tgemm_dense(A, B, a_kcont=bool(akc), b_kcont=bool(bkc), epi=epi, bias=bias, aux=aux, M=M, N=N, K=K)
    for _ in range(3):
        f()
    ops.lib_profile(True, repeat=9)  # the report averages over executions: 9 back-to-back launches share one event pair's ~4 us
    for _ in range(10):
        f()
    torch.cuda.synchronize()
    p = ops.lib_profile_report()
    ops.lib_profile(False)
    k = [x for x in p if "tgemm" in x][0]
    res.append((K, p[k][1] / p[k][0] * 1e3))
ks = np.array([r[0] / 32 for r in res]); t = np.array([r[1] for r in res])
slope, icpt = np.polyfit(ks[1:], t[1:], 1)
print(f"M{M} N{N} cfg {cfg!r} layout {akc}{bkc} epi {epi}: " + "  ".join(f"K{k}: {v:.2f} us" for k, v in res) + f"   => fixed {icpt:.2f} us + {slope * 1e3:.0f} ns per 32-k chunk")
