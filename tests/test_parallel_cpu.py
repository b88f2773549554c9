"""world_size-2 gloo test of the data-parallel gradient path (jorldy_amd/parallel.py): the flat
all-reduce must give every rank the gradient of ONE learner on the concatenated minibatch."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _model():
    torch.manual_seed(0)
    return torch.nn.Sequential(torch.nn.Linear(4, 16), torch.nn.ReLU(), torch.nn.Linear(16, 3))


def _worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from jorldy_amd.parallel import make_grad_sync

    net = _model()
    if rank == 1:  # perturb: broadcast at init must restore rank 0's weights
        with torch.no_grad():
            for p in net.parameters():
                p.add_(1.0)
    sync = make_grad_sync(net, dist)
    rng = np.random.RandomState(5)
    x = torch.from_numpy(rng.randn(world * 8, 4).astype(np.float32))
    y = torch.from_numpy(rng.randn(world * 8, 3).astype(np.float32))
    xs, ys = x[rank * 8 : (rank + 1) * 8], y[rank * 8 : (rank + 1) * 8]
    opt = torch.optim.Adam(net.parameters(), lr=1e-2)
    for _ in range(3):
        opt.zero_grad(set_to_none=True)
        ((net(xs) - ys) ** 2).mean().backward()
        sync()
        torch.nn.utils.clip_grad_norm_(net.parameters(), 0.5)
        opt.step()
    torch.save({k: v.clone() for k, v in net.state_dict().items()}, os.path.join(out_dir, f"rank{rank}.pt"))
    dist.destroy_process_group()


def test_flat_grad_sync_equals_single_learner(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    sd = [torch.load(os.path.join(tmp_path, f"rank{r}.pt")) for r in range(world)]
    # single learner on the concatenated batch
    net = _model()
    rng = np.random.RandomState(5)
    x = torch.from_numpy(rng.randn(world * 8, 4).astype(np.float32))
    y = torch.from_numpy(rng.randn(world * 8, 3).astype(np.float32))
    opt = torch.optim.Adam(net.parameters(), lr=1e-2)
    for _ in range(3):
        opt.zero_grad(set_to_none=True)
        ((net(x) - y) ** 2).mean().backward()
        torch.nn.utils.clip_grad_norm_(net.parameters(), 0.5)
        opt.step()
    for k, v in net.state_dict().items():
        torch.testing.assert_close(sd[0][k], sd[1][k], rtol=0, atol=0)  # ranks stay bit-identical
        torch.testing.assert_close(sd[0][k], v, rtol=1e-5, atol=1e-6)


def _bucket_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from jorldy_amd.parallel import BucketSync

    sync = BucketSync(dist)
    params = torch.full((37,), float(rank + 1))
    moments = torch.full((37,), float(10 * (rank + 1)))
    sync.broadcast(params, moments)  # identical start: rank 0's buckets everywhere
    grads = torch.arange(37, dtype=torch.float32) * (rank + 1)
    sync.reduce_flat(grads)  # mean over ranks, in place
    torch.save({"params": params, "moments": moments, "grads": grads}, os.path.join(out_dir, f"b{rank}.pt"))
    dist.destroy_process_group()


def test_bucket_sync_mean_gradient_and_broadcast(tmp_path):
    """The flat-bucket form used by the native Rainbow learner (ops.RainbowNet buckets)."""
    world = 2
    mp.spawn(_bucket_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    out = [torch.load(os.path.join(tmp_path, f"b{r}.pt")) for r in range(world)]
    for r in range(world):
        assert torch.equal(out[r]["params"], torch.full((37,), 1.0)) and torch.equal(out[r]["moments"], torch.full((37,), 10.0))
        assert torch.equal(out[r]["grads"], torch.arange(37, dtype=torch.float32) * 1.5)


def test_pin_rank_to_cores_partitions_the_allowed_cores():
    from jorldy_amd.parallel import pin_rank_to_cores

    before = sorted(os.sched_getaffinity(0))
    try:
        world = min(4, len(before))
        seen = []
        for r in range(world):
            os.sched_setaffinity(0, before)
            mine = pin_rank_to_cores(r, world, min_cores=1)
            assert mine and sorted(os.sched_getaffinity(0)) == mine
            seen += mine
        assert len(seen) == len(set(seen)) and set(seen) <= set(before)  # disjoint slices of the allowed cores
    finally:
        os.sched_setaffinity(0, before)


# ---- PER under data parallelism: every rank owns a shard (own sum tree), samples its part of the
# global batch locally; the IS weights must be those of ONE tree holding all shards (per_buffer.py:88-94) and the
# averaged gradient bucket that of one learner on the concatenated batch.
def _shard(rank, n=40):
    from oracle.jorldy_oracle import PEROracle

    rng = np.random.RandomState(100 + rank)
    per = PEROracle(64, uniform_sample_prob=0.05)
    trs = [{"x": rng.randn(1, 5).astype(np.float32), "y": rng.randn(1, 1).astype(np.float32)} for _ in range(n + 3 * rank)]
    per.store(trs)
    for leaf in range(per.buffer_counter):
        per.update_priority(float(rng.rand() ** 2 + 0.01), leaf + per.first_leaf_index)
    return per


def _per_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from jorldy_amd.parallel import BucketSync, sharded_is_weights

    per = _shard(rank)
    np.random.seed(7 + rank)
    B, beta = 8, 0.6
    _, idx, _, _ = per.sample_indices(beta, B)  # local draw + descent on the shard's own tree
    p = torch.from_numpy(per.sum_tree[idx])
    w = sharded_is_weights(p, per.sum_tree[0], per.buffer_counter, per.uniform_sample_prob, beta, dist)
    # the native learners' data-parallel hook (ops.RainbowNet / jh_pponet buckets): one flat fp32 gradient bucket
    rows = [per.buffer[i - per.first_leaf_index] for i in idx]
    x = torch.from_numpy(np.concatenate([r["x"] for r in rows], 0))
    y = torch.from_numpy(np.concatenate([r["y"] for r in rows], 0))
    theta = torch.linspace(-1, 1, 5).reshape(5, 1).requires_grad_(True)
    loss = (w.float().unsqueeze(-1) * (x @ theta - y) ** 2).mean()
    (g,) = torch.autograd.grad(loss, theta)
    bucket = g.reshape(-1).clone()
    BucketSync(dist).reduce_flat(bucket)
    torch.save({"idx": idx, "w": w, "grad": bucket}, os.path.join(out_dir, f"per{rank}.pt"))
    dist.destroy_process_group()


def test_sharded_per_weights_and_gradient_equal_one_learner(tmp_path):
    from oracle.jorldy_oracle import PEROracle

    world, B, beta = 2, 8, 0.6
    mp.spawn(_per_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    out = [torch.load(os.path.join(tmp_path, f"per{r}.pt"), weights_only=False) for r in range(world)]
    # ONE learner: a single tree over the concatenation of the shards, fed the same per-rank index lists
    shards = [_shard(r) for r in range(world)]
    big = PEROracle(128, uniform_sample_prob=0.05)
    base, rows_all = [], []
    for s in shards:
        base.append(big.buffer_counter)
        big.store([s.buffer[i] for i in range(s.buffer_counter)])
        for leaf in range(s.buffer_counter):
            big.update_priority(float(s.sum_tree[leaf + s.first_leaf_index]), base[-1] + leaf + big.first_leaf_index)
    gidx = np.concatenate([out[r]["idx"] - shards[r].first_leaf_index + base[r] + big.first_leaf_index for r in range(world)])
    pr = big.sum_tree[gidx]
    uni = 1.0 / big.buffer_counter  # per_buffer.py:88-94 on the big tree
    w = (uni / ((1 - 0.05) * pr / big.sum_tree[0] + 0.05 * uni)) ** beta
    w /= w.max()
    np.testing.assert_allclose(np.concatenate([out[r]["w"].numpy() for r in range(world)]), w, rtol=1e-12)
    x = torch.from_numpy(np.concatenate([big.buffer[i - big.first_leaf_index]["x"] for i in gidx], 0))
    y = torch.from_numpy(np.concatenate([big.buffer[i - big.first_leaf_index]["y"] for i in gidx], 0))
    theta = torch.linspace(-1, 1, 5).reshape(5, 1).requires_grad_(True)
    loss = (torch.from_numpy(w).float().unsqueeze(-1) * (x @ theta - y) ** 2).mean()
    (g,) = torch.autograd.grad(loss, theta)
    for r in range(world):
        torch.testing.assert_close(out[r]["grad"], g.reshape(-1), rtol=1e-5, atol=1e-7)
    assert torch.equal(out[0]["grad"], out[1]["grad"])
