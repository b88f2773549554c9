#!/usr/bin/env python3
"""Which ingredient breaks the tiled backward at HalfCheetah shapes (GPU run 3: d(loss)/d head.l.weight 1e-3 off at S = 17, A = 6,
B = 1024)?  jh_pponet_forward / _backward against the reference's module in float64 for shapes that vary ONE thing at a time."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch

from jorldy_amd import ops
from mirror.networks import Network

for cont, S, H, A, B in ((True, 17, 512, 6, 1024), (True, 17, 512, 3, 1024), (True, 11, 512, 6, 1024), (True, 16, 512, 6, 1024), (True, 20, 512, 3, 1024),
                         (True, 17, 512, 6, 512), (False, 17, 512, 12, 1024), (True, 17, 64, 6, 2048)):
    torch.manual_seed(0)
    ref = Network("continuous_policy_value" if cont else "discrete_policy_value", S, A, D_hidden=H).double()
    with torch.no_grad():
        for p in ref.parameters():
            p.add_(0.05 * torch.randn_like(p))
            p.copy_(p.float().double())
    net = ops.PPONet(S, H, A, cont, 4096, "cuda:0")
    net.params.copy_(torch.cat([p.detach().reshape(-1) for p in ref.parameters()]).float().cuda())
    g = torch.Generator().manual_seed(1)
    x = torch.randn(3000, S, generator=g)
    idx = torch.randperm(3000, generator=g)[:B]
    outs = net.forward(x.cuda(), idx=idx.cuda())
    r64 = ref.raw(x[idx].double())
    gs = [torch.randn(o.shape, generator=g) / B for o in r64]
    torch.autograd.backward(list(r64), [t.double() for t in gs])
    gd = [t.cuda() for t in gs]
    net.backward(x.cuda(), idx.cuda(), gd[0], gd[1] if cont else None, gd[2] if cont else gd[1])
    torch.cuda.synchronize()
    o, errs = 0, {}
    for k, p in ref.named_parameters():
        n = p.numel()
        ours = net.grads[o : o + n].double().cpu().view_as(p)
        errs[k] = float((ours - p.grad).abs().max() / p.grad.abs().max())
        o += n
    fe = [float((a.double().cpu() - b).abs().max() / b.abs().max()) for a, b in zip([t for t in outs if t is not None], r64)]
    print(f"cont={cont} S={S} H={H} A={A} B={B}: forward {['%.1e' % e for e in fe]} grads", {k: "%.1e" % v for k, v in errs.items()}, flush=True)
