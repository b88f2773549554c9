import os

import numpy as np

G = os.path.join(os.path.dirname(__file__), "golden")


def load(name):
    return np.load(os.path.join(G, name + ".npz"))


def cu(a, dtype=None):
    import torch

    t = torch.as_tensor(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.to(dtype)
    return t.cuda()


def f32(a):
    import torch

    return cu(np.asarray(a, dtype=np.float32))


def npy(t):
    return t.detach().cpu().numpy()
