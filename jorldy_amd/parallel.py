"""Data-parallel learners: one process per GPU, one RCCL all-reduce of the flat fp32 gradient per
minibatch over xGMI.  The reference has a single learner and no collective at all
(distributed = Ray actors on host CPUs); DP parity is "same math as one learner on the concatenated
minibatch": grads are summed then divided by the world size BEFORE clip_grad_norm_, so every rank
computes the same global norm and takes the identical Adam step.

Gradient sizes here are ~1 MB (266 755 fp32 params for PPO CartPole): the all-reduce is latency-
bound, so a single flat bucket, in place, on the compute stream is the right shape.
"""
import torch


class FlatGradSync:
    def __init__(self, module, dist, group=None):
        self.dist, self.group = dist, group
        self.world = dist.get_world_size(group)
        self.params = [p for p in module.parameters() if p.requires_grad]
        n = sum(p.numel() for p in self.params)
        self.flat = torch.zeros(n, dtype=torch.float32, device=self.params[0].device)
        self.views, o = [], 0
        for p in self.params:
            self.views.append(self.flat[o : o + p.numel()].view_as(p))
            o += p.numel()

    def __call__(self):
        """Call between backward and clip/step: p.grad <- mean over ranks of p.grad."""
        torch._foreach_copy_(self.views, [p.grad for p in self.params])
        self.dist.all_reduce(self.flat, op=self.dist.ReduceOp.SUM, group=self.group)
        self.flat.div_(self.world)
        torch._foreach_copy_([p.grad for p in self.params], self.views)

    def reduce_flat(self, flat):
        """Native-backend form: the gradient already IS one flat bucket (jh_pponet_* writes it in
        state_dict order) -> one in-place all-reduce, no packing copies."""
        self.dist.all_reduce(flat, op=self.dist.ReduceOp.SUM, group=self.group)
        flat.div_(self.world)

    def broadcast_weights(self, src=0):
        for p in self.params:
            self.dist.broadcast(p.data, src=src, group=self.group)


def make_grad_sync(module, dist, group=None):
    sync = FlatGradSync(module, dist, group)
    sync.broadcast_weights(0)  # identical start (ncclBroadcast only at init/load)
    return sync
