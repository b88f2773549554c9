"""Data-parallel learners: one process per GPU, one RCCL all-reduce of the flat fp32 gradient per
minibatch over xGMI.  The reference has a single learner and no collective at all
(distributed = Ray actors on host CPUs); DP parity is "same math as one learner on the concatenated
minibatch": grads are summed then divided by the world size BEFORE clip_grad_norm_, so every rank
computes the same global norm and takes the identical Adam step.

Gradient sizes here are ~1 MB (266 755 fp32 params for PPO CartPole): the all-reduce is latency-
bound, so a single flat bucket, in place, on the compute stream is the right shape.
"""
import os

import torch


class Transport:
    """How the ranks' collectives travel.  Three forms, chosen once per process group:

      "rccl"   jh_comm_* of the C ABI (include/jorldy_hip.h): the library's own RCCL communicator, ncclAvg all-reduce in
               place on the CURRENT stream -- the default when the process group's backend is nccl (= RCCL);
               torch.distributed is only the side channel that ships the 128-byte unique id.  Capturable.
      "torch"  torch.distributed collectives on device tensors (backend nccl; JH_DP_COLLECTIVE=torch, or when the
               library's communicator cannot be created).  Capturable.
      "host"   backend gloo: device tensors are staged through host memory around the collective.  This is the
               two-ranks-on-one-GPU test transport (RCCL refuses two ranks on one device); never capturable.
      "peer"   JH_DP_COLLECTIVE=peer (round 6): the gradient bucket's mean and the <= 16-float exchanges through peer pointers
               (jh_peer_* of the C ABI: every rank reads its peers' arenas over xGMI, two one-hop exchanges inside the library's
               own launches, no collective library's launch and no ring); the process group -- any backend -- is the side
               channel for the 64-byte IPC handles and still carries broadcasts / the float64 all-gather.  One node only.
               Capturable.  Works across two processes on ONE GPU too (the functional test of tests/test_dp_two_ranks_gpu.py)."""

    def __init__(self, dist, group=None, device=None):
        self.dist, self.group = dist, group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        backend = str(dist.get_backend(group)).lower()
        self.kind = "host" if "gloo" in backend and "nccl" not in backend else "torch"
        self.comm = None
        if self.kind == "torch" and os.environ.get("JH_DP_COLLECTIVE", "rccl") != "torch" and device is not None and torch.device(device).type == "cuda":
            # a COLLECTIVE decision: every rank runs the same sequence of collectives whatever fails locally, and all ranks
            # end on the same transport (a per-rank try/except left ranks on different transports / off-by-one collectives)
            if self._create_comm(torch.device(device)):
                self.kind = "rccl"
        self.base_kind = self.kind  # what broadcasts / the float64 all-gather travel on when the bucket goes through peer pointers
        self.peer = None
        self._want_peer = os.environ.get("JH_DP_COLLECTIVE", "rccl") == "peer" and device is not None and torch.device(device).type == "cuda"
        self._device = torch.device(device) if device is not None else None
        self.capturable = self.kind in ("rccl", "torch")

    def ensure_peer(self, max_floats):
        """COLLECTIVE (every rank, same order): with JH_DP_COLLECTIVE=peer, create this rank's arena for buckets of up to max_floats floats,
        exchange the IPC handles over the process group and map the peers.  All ranks end on the peer transport or none does."""
        if not self._want_peer or (self.peer is not None and self._peer_floats >= max_floats):
            return self.kind == "peer"
        import ctypes as C

        from . import _lib as L

        lib = L.load()
        h, ok, err = C.c_void_p(), 1, None
        handle = torch.zeros(64, dtype=torch.uint8)
        try:
            L.check(lib.jh_peer_create(L.ctx(self._device.index), self.world, self.rank, int(max_floats), C.byref(h)))
            L.check(lib.jh_peer_handle(h, L.ptr(handle)))
        except Exception as e:  # noqa: BLE001
            ok, err = 0, e
        parts = [torch.zeros(65, dtype=torch.uint8) for _ in range(self.world)]
        mine = torch.cat([handle, torch.tensor([ok], dtype=torch.uint8)])
        if self.base_kind == "host":
            self.dist.all_gather(parts, mine, group=self.group)
        else:
            dev_parts = [q.to(self._device) for q in parts]
            self.dist.all_gather(dev_parts, mine.to(self._device), group=self.group)
            parts = [q.cpu() for q in dev_parts]
        all_ok = all(int(q[64]) == 1 for q in parts)
        if all_ok:
            try:
                handles = torch.cat([q[:64] for q in parts]).contiguous()
                L.check(lib.jh_peer_connect(h, L.ptr(handles)))
            except Exception as e:  # noqa: BLE001
                all_ok, err = False, e
        flag = torch.tensor([1 if all_ok else 0], dtype=torch.int32)
        if self.base_kind == "host":
            self.dist.all_reduce(flag, op=self.dist.ReduceOp.MIN, group=self.group)
        else:
            f = flag.to(self._device)
            self.dist.all_reduce(f, op=self.dist.ReduceOp.MIN, group=self.group)
            flag = f.cpu()
        if int(flag.item()) == 0:
            if h:
                lib.jh_peer_destroy(h)
            print(f"[jorldy_amd] rank {self.rank}: peer-pointer transport unavailable ({type(err).__name__ if err else 'another rank failed'}: {err}); ALL ranks stay on {self.kind!r}")
            self._want_peer = False
            return False
        if self.peer is not None:
            lib.jh_peer_destroy(self.peer)
        self.peer, self._peer_floats, self._lib, self._L = h, int(max_floats), lib, L
        self.kind, self.capturable = "peer", True
        return True

    def peer_status(self):
        """-> (bounded waits that gave up, completed all-reduces) of the peer transport."""
        import ctypes as C

        t, c = C.c_int32(), C.c_int64()
        self._L.check(self._lib.jh_peer_status(self.peer, C.byref(t), C.byref(c)))
        return t.value, c.value

    def _create_comm(self, device):
        """-> True when EVERY rank holds a working jh_comm communicator; otherwise nothing is left behind on any rank.
        Sequence on every rank, unconditionally: broadcast(128-byte id from rank 0; all zeros = rank 0 could not make one),
        [jh_comm_create], all_reduce(MIN) of the local success flag.  JH_COMM_INJECT=id|create makes the named step fail
        (tests of the fallback)."""
        import ctypes as C

        inject = os.environ.get("JH_COMM_INJECT", "")
        lib = L = None
        err = None
        try:
            from . import _lib as L

            lib = L.load()
        except Exception as e:
            err = e
        ident = torch.zeros(128, dtype=torch.uint8)
        if self.rank == 0 and lib is not None:
            try:
                if inject == "id":
                    raise RuntimeError("injected: jh_comm_unique_id")
                L.check(lib.jh_comm_unique_id(L.ptr(ident)))
            except Exception as e:
                err = e
                ident.zero_()
        ident = ident.to(device)  # the side channel: one 128-byte broadcast on the existing process group
        self.dist.broadcast(ident, src=self.dist.get_global_rank(self.group, 0) if self.group is not None else 0, group=self.group)
        ident = ident.cpu()
        h, ok = None, 0
        if lib is not None and bool(ident.any()):
            try:
                if inject == "create":
                    raise RuntimeError("injected: jh_comm_create")
                h = C.c_void_p()
                L.check(lib.jh_comm_create(L.ctx(device.index), self.world, self.rank, L.ptr(ident), C.byref(h)))
                ok = 1
            except Exception as e:
                err, h = e, None
        flag = torch.tensor([ok], dtype=torch.int32, device=device)
        self.dist.all_reduce(flag, op=self.dist.ReduceOp.MIN, group=self.group)
        if int(flag.item()) == 0:
            if h is not None:
                lib.jh_comm_destroy(h)
            why = f"{type(err).__name__}: {err}" if err is not None else "another rank failed"
            print(f"[jorldy_amd] rank {self.rank}: C-ABI RCCL communicator unavailable ({why}); ALL ranks use torch.distributed collectives")
            return False
        self.comm, self._lib, self._L = h, lib, L
        return True

    def __del__(self):
        try:
            if self.comm is not None:
                self._lib.jh_comm_destroy(self.comm)
                self.comm = None
            if getattr(self, "peer", None) is not None:
                self._lib.jh_peer_destroy(self.peer)
                self.peer = None
        except Exception:
            pass

    # ---- the three collectives the learners need ----------------------------------------------------
    def mean_(self, flat):
        """flat (fp32, contiguous) <- mean over ranks, in place."""
        if self.kind == "peer" and flat.is_cuda and flat.dtype == torch.float32 and flat.is_contiguous():
            n = int(flat.numel())
            if n <= 16:
                self._L.check(self._lib.jh_peer_allreduce_small_f32(self.peer, self._L.ptr(flat), n, 1, self._L.stream_ptr()))
                return flat
            if n <= self._peer_floats and flat.data_ptr() % 16 == 0:
                self._L.check(self._lib.jh_peer_allreduce_mean_f32(self.peer, self._L.ptr(flat), n, self._L.stream_ptr()))
                return flat
        kind = self.base_kind if self.kind == "peer" else self.kind
        if kind == "rccl":
            self._L.check(self._lib.jh_comm_allreduce_mean_f32(self.comm, self._L.ptr(flat), int(flat.numel()), self._L.stream_ptr()))
        elif kind == "torch" or not flat.is_cuda:
            self.dist.all_reduce(flat, op=self.dist.ReduceOp.SUM, group=self.group)
            flat.div_(self.world)
        else:
            h = flat.cpu()
            self.dist.all_reduce(h, op=self.dist.ReduceOp.SUM, group=self.group)
            flat.copy_(h.div_(self.world))
        return flat

    def broadcast_(self, t, src=0):
        kind = self.base_kind if self.kind == "peer" else self.kind
        if kind == "rccl" and t.is_contiguous():
            self._L.check(self._lib.jh_comm_broadcast(self.comm, self._L.ptr(t), int(t.numel() * t.element_size()), int(src), self._L.stream_ptr()))
            return t
        gsrc = self.dist.get_global_rank(self.group, src) if self.group is not None else src
        if kind != "host" or not t.is_cuda:
            self.dist.broadcast(t, src=gsrc, group=self.group)
        else:
            h = t.cpu()
            self.dist.broadcast(h, src=gsrc, group=self.group)
            t.copy_(h)
        return t

    def all_gather_f64_(self, out, loc):
        """out [world * n] <- the ranks' loc [n] (float64) in rank order."""
        kind = self.base_kind if self.kind == "peer" else self.kind
        if kind == "rccl":
            self._L.check(self._lib.jh_comm_allgather_f64(self.comm, self._L.ptr(loc), self._L.ptr(out), int(loc.numel()), self._L.stream_ptr()))
        elif kind == "torch" or not loc.is_cuda:
            self.dist.all_gather_into_tensor(out, loc, group=self.group)
        else:
            h = torch.empty(out.shape, dtype=out.dtype)
            self.dist.all_gather_into_tensor(h, loc.cpu(), group=self.group)
            out.copy_(h)
        return out


class FlatGradSync:
    """Mean gradient over ranks for a list of parameter tensors: grads packed into one flat fp32 bucket -> Transport.mean_ ->
    unpacked.  The agents' networks live in ONE flat bucket already and use BucketSync / reduce_flat; this packing form is kept
    for callers that hold separate tensors (and for the CPU collective tests)."""

    def __init__(self, module, dist, group=None, transport=None):
        self.dist, self.group = dist, group
        self.params = [p for p in module.parameters() if p.requires_grad]
        self.transport = transport or Transport(dist, group, self.params[0].device)
        self.world = self.transport.world
        n = sum(p.numel() for p in self.params)
        self.flat = torch.zeros(n, dtype=torch.float32, device=self.params[0].device)
        self.views, o = [], 0
        for p in self.params:
            self.views.append(self.flat[o : o + p.numel()].view_as(p))
            o += p.numel()

    @property
    def capturable(self):
        return self.transport.capturable

    def __call__(self):
        """Call between backward and clip/step: p.grad <- mean over ranks of p.grad."""
        torch._foreach_copy_(self.views, [p.grad for p in self.params])
        self.transport.mean_(self.flat)
        torch._foreach_copy_([p.grad for p in self.params], self.views)

    def reduce_flat(self, flat):
        """Native-backend form: the gradient already IS one flat bucket (jh_pponet_* writes it in
        state_dict order) -> one in-place all-reduce, no packing copies."""
        self.transport.mean_(flat)

    def broadcast_weights(self, src=0):
        for p in self.params:
            self.transport.broadcast_(p.data, src)


def make_grad_sync(module, dist, group=None, transport=None):
    sync = FlatGradSync(module, dist, group, transport)
    sync.broadcast_weights(0)  # identical start (ncclBroadcast only at init/load)
    return sync


class BucketSync:
    """Data-parallel hook for learners whose gradient already is ONE flat fp32 bucket in library-owned
    layout (ops.RainbowNet): `reduce_flat` between backward and the optimizer step = mean over ranks, in
    place, one RCCL all-reduce (Rainbow Atari: 12 MB -- bandwidth- rather than latency-bound, still a single
    bucket: the backward of a B=32 batch is ~200 us, there is nothing to overlap it with)."""

    def __init__(self, dist, group=None, transport=None, device=None):
        self.dist, self.group = dist, group
        self.transport = transport or Transport(dist, group, device)
        self.world = self.transport.world

    @property
    def capturable(self):
        return self.transport.capturable

    def reduce_flat(self, flat):
        self.transport.mean_(flat)

    def broadcast(self, *buckets, src=0):
        for b in buckets:
            self.transport.broadcast_(b, src)


def sharded_is_weights(p_sampled, root, count, usp, beta, dist=None, group=None):
    """Importance-sampling weights of per_buffer.py:88-94 for ONE logical PER buffer whose slots are sharded over the
    data-parallel ranks (every rank owns a shard with its own sum tree and samples its part of the global batch from
    it -- SURVEY.md §8e):

        P_i = (1 - usp) p_i / ROOT + usp / COUNT        ROOT = sum_g root_g      COUNT = sum_g count_g
        w_i = ((1 / COUNT) / P_i)^beta / max_j w_j      max over the GLOBAL batch (all ranks' samples)

    w is decreasing in p, so max_j w_j belongs to the smallest sampled priority of the global batch: one all-gather
    of three float64 per rank {root_g, count_g, min sampled p_g} is the only collective.  The result equals the
    weights of a single tree holding all shards for the same index lists.  Works on CPU (gloo) and device tensors;
    dist=None: a single shard (then identical to the local normalisation)."""
    p = p_sampled.reshape(-1).to(torch.float64)
    loc = torch.stack([torch.as_tensor(root, dtype=torch.float64, device=p.device).reshape(()),
                       torch.as_tensor(count, dtype=torch.float64, device=p.device).reshape(()), p.min()])
    if dist is not None and dist.get_world_size(group) > 1:
        parts = [torch.empty_like(loc) for _ in range(dist.get_world_size(group))]
        dist.all_gather(parts, loc, group=group)
        allv = torch.stack(parts)
    else:
        allv = loc.unsqueeze(0)
    root_t, count_t, min_p = allv[:, 0].sum(), allv[:, 1].sum(), allv[:, 2].min()
    uni = 1.0 / count_t
    prob = lambda q: (1.0 - usp) * (q / root_t) + usp * uni  # operation order of per_buffer.py:90-92
    w = (uni / prob(p)) ** beta
    w_max = (uni / prob(min_p)) ** beta
    return w / w_max


def attach_data_parallel(agent, dist, group=None):
    """One learner per GPU (north star: Ape-X's many-actor / one-learner re-expressed as DP learners):
    every rank keeps its own actors and its own replay shard / sum tree (no data-path collective), samples
    its own minibatch of `batch_size`, and the gradients are averaged before the optimizer step so all
    ranks hold identical weights.  PER: the IS weights are those of the single logical buffer (sharded_is_weights:
    one all-gather of {root, count, min sampled p} per learn()).  Works for PPO and the DQN / Rainbow / Ape-X family.
    Returns the hook (also stored as agent.grad_sync)."""
    net = getattr(agent, "_net", None)
    device = getattr(agent, "device", None)
    assert net is not None, "attach_data_parallel needs a jorldy_amd agent (its network lives in libjorldy_hip's flat buckets)"
    transport = Transport(dist, group, device)
    transport.ensure_peer(int(net.grads.numel()))  # JH_DP_COLLECTIVE=peer: arenas + IPC handles (a collective step; no-op otherwise)
    sync = BucketSync(dist, group, transport)  # ops.PPONet / ops.RainbowNet: the gradient already IS one flat bucket
    sync.broadcast(net.params, net.m, net.v, *([net.target] if hasattr(net, "target") else []))  # identical start (ncclBroadcast only at init/load)
    mem = getattr(agent, "memory", None)
    if mem is not None and hasattr(mem, "attach_shards"):  # PER: this rank's buffer is one shard of the logical buffer
        mem.attach_shards(dist, group, transport)
    agent.grad_sync = sync
    agent.graph_with_collective = bool(getattr(agent, "graph_with_collective", True)) and transport.capturable
    agent._graph = None
    if hasattr(agent, "_graphs"):
        agent._graphs = {}
    return sync


def pin_rank_to_cores(local_rank, local_world, min_cores=4):
    """North star: "actors pinned to host cores".  One process per GPU, each with its collector thread (busy-polling
    the acting exchange area) and its host envs: give every rank its own contiguous slice of the cores this process
    may run on, so the ranks' pollers never share a core.  Returns the core list (or None when affinity is not
    controllable here)."""
    import os

    try:
        avail = sorted(os.sched_getaffinity(0))
        per = len(avail) // max(1, local_world)
        if per < min_cores:  # a rank also runs the HIP runtime's helper threads: do not squeeze it onto one or two cores
            return None
        mine = avail[local_rank * per : (local_rank + 1) * per] or avail
        os.sched_setaffinity(0, mine)
        return mine
    except (AttributeError, OSError):
        return None


def _parse_cpulist(text):
    cpus = []
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        cpus += list(range(int(lo), int(hi or lo) + 1))
    return cpus


def pin_to_gpu_node(device_index=None, local_rank=0, ranks_on_node=1, min_cores=4):
    """Pin the CALLING thread (the one that will run the collector loop) to host cores on the GPU's own NUMA node /
    PCIe root.  Acting is two PCIe crossings per timestep: measured on a 2-socket MI355X host, 9.3 us per timestep from
    the local socket against 12.9 us from the far one (437 k vs 362 k env transitions/s in bench.py), and an unpinned
    process lands on either.  With several ranks per node each gets its own slice of the node's cores.
    Set JH_NO_PIN=1 to leave the affinity alone.  Returns the core list or None."""
    import ctypes as C
    import os

    from . import _lib as L

    if os.environ.get("JH_NO_PIN") == "1":
        return None
    try:
        buf = C.create_string_buffer(4096)
        L.check(L.load().jh_ctx_local_cpulist(L.ctx(device_index), buf, 4096))
        local = set(_parse_cpulist(buf.value.decode()))
        avail = sorted(local & set(os.sched_getaffinity(0)))
        if len(avail) < min_cores:
            return None
        per = len(avail) // max(1, ranks_on_node)
        mine = avail[local_rank % max(1, ranks_on_node) * per : (local_rank % max(1, ranks_on_node) + 1) * per] if per >= min_cores else avail
        os.sched_setaffinity(0, mine)
        return mine
    except Exception:
        return None


def _gpu_local_cpulist(index):
    """sysfs local_cpulist of HIP device `index`'s PCI function (no HIP context is created for it)."""
    import torch

    p = torch.cuda.get_device_properties(index)
    bdf = f"{p.pci_domain_id:04x}:{p.pci_bus_id:02x}:{p.pci_device_id:02x}.0"
    with open(f"/sys/bus/pci/devices/{bdf}/local_cpulist") as f:
        return f.read().strip()


def ranks_sharing_node(local_rank, local_world):
    """(slot, count) of this rank among the local ranks whose GPU hangs off the same NUMA node / PCIe root (equal
    sysfs local_cpulist), rank r using GPU r: the ranks of one node split its cores between them (pin_to_gpu_node).
    Nothing about the topology is assumed (round 1 hard-coded 4 GPUs per socket); when sysfs cannot be read every
    local rank is taken to share one node (smaller, still disjoint slices)."""
    try:
        lists = [_gpu_local_cpulist(i) for i in range(local_world)]
        same = [i for i, l in enumerate(lists) if l == lists[local_rank]]
        return same.index(local_rank), len(same)
    except Exception:
        return local_rank, max(1, local_world)
