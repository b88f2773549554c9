"""Optimizer factory with the reference's interface (core/optimizer/__init__.py:21-31):
`Optimizer(name, params=..., **kw)` -> the torch.optim class whose snake_case name is `name`."""
import inspect
import re

import torch.optim as _optim

_snake = lambda x: re.sub("([a-z])([A-Z])", r"\1_\2", x).lower()
optimizer_dict = {_snake(n): c for n, c in inspect.getmembers(_optim, inspect.isclass)}


def Optimizer(name, *args, **kwargs):
    if not isinstance(name, str):
        raise Exception("### name variable must be string! ###")
    key = name.lower()
    if key not in optimizer_dict:
        raise Exception(f"### can use only follows {list(optimizer_dict.keys())}")
    return optimizer_dict[key](*args, **kwargs)
