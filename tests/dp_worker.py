"""One rank of the two-ranks-on-one-GPU data-parallel tests (tests/test_dp_two_ranks_gpu.py).  Not a test module.

usage: python tests/dp_worker.py <mode: ppo|ppo_cnn|rainbow|apex|peer_unit> <rank> <world> <port> <out.npz> [gloo|nccl]
JH_DP_COLLECTIVE=peer in the environment: the gradient bucket and the critic sums travel through peer pointers (jh_peer_*, hipIpc handles
opened across the two processes on the one GPU) instead of the host-staged gloo all-reduce, captured inside the learn() graph.
nccl (= RCCL): rank r on GPU r -- the form the N-GPU bench runs; needs >= world GPUs (the driver's 8-GPU node).
RCCL refuses two ranks on one device, gloo does not: the process group is gloo, the gradient bucket is staged through
host memory around the all-reduce (jorldy_amd.parallel.Transport kind "host"); everything else -- the native agents'
DP branch (ppo_update(do_adam=0) -> reduce_flat -> adam_step; RainbowNet backward -> reduce_flat -> optim_step; the
sharded PER weights) -- is the code the N-GPU RCCL run executes."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import synth  # noqa: E402  (test infrastructure: recipes for weights / rollouts)

# config.ppo.cartpole (BASELINE configs[1]) at its own learning rate.  PPO's critic is max(mean(e1), mean(e2)) (ppo.py:147-154), a max of two MEANS
# over the whole minibatch; from the third update on |V - V_old| passes epsilon_clip for some rows and the two means differ.  Round 3 let every
# rank take the max of ITS means (equal to one learner only while the clamp is inactive: the test ran at lr 2.5e-6 and asserted that); round 4
# reduces {sum e1, sum e2} over the ranks before the backward (jh_pponet_ppo_update_dp_begin / _end), so DP == one learner with the clamp active.
PPO_CFG = dict(S=4, A=2, H=512, W=8, T=128, B=256, E=3, lr=2.5e-4, seed=20260925)
RB_CFG = dict(S=4, A=3, H=32, K=51, B=32, N=256, fill=200, n_step=3, lr=1e-3)


def ppo_agent(W, B, **kw):
    from jorldy_amd.core.agent import Agent

    c = PPO_CFG
    agent = Agent("ppo", state_size=c["S"], action_size=c["A"], hidden_size=c["H"], network="discrete_policy_value", optim_config={"name": "adam", "lr": c["lr"]},
                  batch_size=B, n_step=c["T"], n_epoch=c["E"], _lambda=0.95, epsilon_clip=0.1, vf_coef=1.0, ent_coef=0.01, clip_grad_norm=1.0, gamma=0.99,
                  run_step=100000, num_workers=W, device="cuda", backend="native", lr_decay=False, **kw)
    rec = synth.ppo_recipe({k: v.shape for k, v in agent.network.state_dict().items()}, c["seed"])
    agent.network.load_state_dict({k: torch.from_numpy(v) for k, v in rec.items()})
    agent.memory.first_store = False
    return agent


def ppo_rows(rank):
    c = PPO_CFG
    trs = synth.ppo_rollout(np.random.RandomState(50 + rank), c["W"] * c["T"], c["S"], c["A"], False, clamp_every=0)
    return {k: np.concatenate([t[k] for t in trs], 0) for k in ("state", "next_state", "reward", "done", "action")}


# PPO on the CNN head (round 6): a small image, minibatches of 8 rows per rank, a learning rate large enough for the value clamp to become active
PPO_CNN_CFG = dict(S=(4, 44, 52), A=4, H=64, W=2, T=16, B=8, E=3, lr=2e-3, seed=20260930)


def ppo_cnn_agent(W, B, **kw):
    from jorldy_amd.core.agent import Agent

    c = PPO_CNN_CFG
    agent = Agent("ppo", state_size=list(c["S"]), action_size=c["A"], hidden_size=c["H"], network="discrete_policy_value", head="cnn", optim_config={"name": "adam", "lr": c["lr"]},
                  batch_size=B, n_step=c["T"], n_epoch=c["E"], _lambda=0.95, epsilon_clip=0.1, vf_coef=1.0, ent_coef=0.01, clip_grad_norm=1.0, gamma=0.99,
                  run_step=100000, num_workers=W, device="cuda", lr_decay=False, **kw)
    rec = synth.ppo_recipe({k: v.shape for k, v in agent.network.state_dict().items()}, c["seed"])
    agent.network.load_state_dict({k: torch.from_numpy(v) for k, v in rec.items()})
    agent.memory.first_store = False
    return agent


def ppo_cnn_rows(rank):
    c = PPO_CNN_CFG
    trs = synth.ppo_image_rollout(np.random.RandomState(70 + rank), c["W"] * c["T"], c["S"], c["A"])
    return {k: np.concatenate([t[k] for t in trs], 0) for k in ("state", "next_state", "reward", "done", "action")}


def rainbow_agent(B, N, **kw):
    from jorldy_amd.core.agent import Agent

    c = RB_CFG
    agent = Agent("rainbow", state_size=c["S"], action_size=c["A"], hidden_size=c["H"], optim_config={"name": "adam", "lr": c["lr"]}, buffer_size=N, batch_size=B,
                  start_train_step=0, target_update_period=10000, run_step=100000, n_step=c["n_step"], num_support=c["K"], v_min=-1, v_max=10, alpha=0.5, beta=0.4,
                  learn_period=1, uniform_sample_prob=0.05, device="cuda", backend="native", **kw)
    shapes = {k: v.shape for k, v in agent.network.state_dict().items()}
    agent.network.load_state_dict({k: torch.from_numpy(v) for k, v in synth.recipe_state_dict(shapes, 77).items()})
    agent.target_network.load_state_dict({k: torch.from_numpy(v) for k, v in synth.recipe_state_dict(shapes, 78).items()})
    agent.memory.first_store = False
    return agent


def apex_agent(B, N, **kw):
    """Ape-X (core/agent/ape_x.py) at a small MLP width: dueling network, n-step double-Q, PER with actor-side priorities, centered RMSprop,
    gradient clipping -- the learner of config.ape_x.atari, one per rank, each with its own replay shard (north_star: "one learner per GPU
    with RCCL all-reduce of gradients")."""
    from jorldy_amd.core.agent import Agent

    c = RB_CFG
    agent = Agent("ape_x", state_size=c["S"], action_size=c["A"], hidden_size=c["H"], network="dueling", head="mlp",
                  optim_config={"name": "rmsprop", "eps": 1.5e-7, "lr": 1e-3, "centered": True}, gamma=0.99, buffer_size=N, batch_size=B, clip_grad_norm=40.0,
                  start_train_step=0, target_update_period=10000, run_step=100000, n_step=c["n_step"], alpha=0.6, beta=0.4, uniform_sample_prob=0.05,
                  num_workers=4, device="cuda", **kw)
    shapes = {k: v.shape for k, v in agent.network.state_dict().items()}
    agent.network.load_state_dict({k: torch.from_numpy(v) for k, v in synth.recipe_state_dict(shapes, 87).items()})
    agent.target_network.load_state_dict({k: torch.from_numpy(v) for k, v in synth.recipe_state_dict(shapes, 88).items()})
    agent.memory.first_store = False
    return agent


def rainbow_shard(rank):
    """This rank's replay shard: rows + priorities (slot order)."""
    c = RB_CFG
    rng = np.random.RandomState(900 + rank)
    n = c["fill"] + 17 * rank  # shards of different fill: COUNT and ROOT differ per rank
    cols = {"state": rng.randn(n, c["S"]).astype(np.float32), "next_state": rng.randn(n, c["S"]).astype(np.float32),
            "action": rng.randint(0, c["A"], size=(n, 1)), "reward": rng.choice([-1.0, 0.0, 1.0, 0.5], size=(n, c["n_step"], 1)),
            "done": rng.rand(n, c["n_step"], 1) < 0.1}
    prio = (rng.rand(n) ** 2 + 0.01).astype(np.float32).astype(np.float64) * (1.0 + rank)
    return cols, prio


def main():
    mode, rank, world, port, out = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
    backend = sys.argv[6] if len(sys.argv) > 6 else "gloo"
    import torch.distributed as dist

    from jorldy_amd.parallel import attach_data_parallel

    rccl = backend == "nccl"
    torch.cuda.set_device(rank if rccl else 0)
    kw = {"device_id": torch.device("cuda", rank)} if rccl else {}
    dist.init_process_group(backend, init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world, **kw)
    peer = os.environ.get("JH_DP_COLLECTIVE") == "peer"
    kinds = ("peer",) if peer else (("rccl", "torch") if rccl else ("host",))
    res = {}
    try:
        if mode == "peer_unit":
            # jh_peer_* by themselves: buckets of awkward lengths (not multiples of 4 / of the rank count), many calls in a row (flag and
            # buffer reuse), eager and replayed from a hipGraph; the small exchange in between
            from jorldy_amd.parallel import Transport

            tr = Transport(dist, None, torch.device("cuda", 0))
            assert tr.ensure_peer(300_000) and tr.kind == "peer" and tr.capturable
            g = torch.Generator().manual_seed(7)  # the SAME stream on both ranks: each knows the other's data
            means, smalls = [], []
            for n in (266_755, 17, 4096, 300_000, 33):
                for it in range(3):
                    both = torch.randn(world, n, generator=g)
                    mine = both[rank].cuda()
                    tr.mean_(mine)
                    want = both[0]
                    for r in range(1, world):
                        want = want + both[r]  # rank order, fp32
                    want = want * (1.0 / world)
                    means.append(float((mine.cpu() - want).abs().max()))
                    sm = torch.randn(world, 5, generator=g)
                    v = sm[rank].cuda()
                    tr._L.check(tr._lib.jh_peer_allreduce_small_f32(tr.peer, tr._L.ptr(v), 5, 0, tr._L.stream_ptr()))
                    smalls.append(float((v.cpu() - sm.sum(0)).abs().max()))
            # ... and from a captured graph: sequence numbers advance on the device
            buf = torch.zeros(70_001, device="cuda")
            gr = torch.cuda.CUDAGraph()
            s_ = torch.cuda.Stream()
            s_.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s_):
                tr.mean_(buf)  # warm
                torch.cuda.synchronize()
                with torch.cuda.graph(gr, stream=s_):
                    tr.mean_(buf)
            torch.cuda.synchronize()
            graph_err = []
            for it in range(6):
                both = torch.randn(world, 70_001, generator=g)
                buf.copy_(both[rank])
                gr.replay()
                torch.cuda.synchronize()
                want = (both[0] + both[1]) * 0.5 if world == 2 else both.sum(0) / world
                graph_err.append(float((buf.cpu() - want).abs().max()))
            # how long one all-reduce of the PPO bucket takes through this transport with both ranks on this ONE device (no xGMI in it: the software
            # floor of the two launches, their flag hand-offs and write-through copies); 200 back-to-back calls between two events
            bucket = torch.randn(266_755, device="cuda")
            for _ in range(20):
                tr.mean_(bucket)
            dist.barrier()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(200):
                tr.mean_(bucket)
            e1.record()
            torch.cuda.synchronize()
            us_per_allreduce = e0.elapsed_time(e1) * 1e3 / 200
            small = torch.randn(2, device="cuda")
            e0.record()
            for _ in range(200):
                tr.mean_(small)
            e1.record()
            torch.cuda.synchronize()
            us_per_small = e0.elapsed_time(e1) * 1e3 / 200
            timeouts, done = tr.peer_status()
            res = dict(means=np.asarray(means), smalls=np.asarray(smalls), graph_err=np.asarray(graph_err), timeouts=timeouts, done=done,
                       us_per_allreduce=us_per_allreduce, us_per_small=us_per_small)
            sync = type("S", (), {"transport": tr})()
        elif mode == "ppo":
            c = PPO_CFG
            agent = ppo_agent(c["W"], c["B"])
            if rank == 1:  # attach must overwrite rank 1's weights with rank 0's
                with torch.no_grad():
                    for p in agent.network.parameters():
                        p.add_(0.01)
            sync = attach_data_parallel(agent, dist)
            assert sync.transport.kind in kinds and (rccl or peer or not agent.graph_with_collective) and (not peer or agent.graph_with_collective)
            agent._predraw = None  # keep this learn()'s index lists in st["idx"] (no lists of a next learn() drawn ahead)
            np.random.seed(200 + rank)
            result = agent.process(ppo_rows(rank), c["T"])
            torch.cuda.synchronize()
            n_upd = c["E"] * (c["W"] * c["T"] // c["B"])
            perms = agent._static["idx"].cpu().numpy().reshape(c["E"], c["W"] * c["T"])  # the epochs' index lists this rank drew (np.random, seed 200 + rank)
            res = dict(params=agent._net.params.cpu().numpy(), perms=perms, stats=np.asarray(agent._static["stats_pin"].np[: n_upd + 1]).copy(),
                       grads=agent._net.grads.cpu().numpy(), n_upd=n_upd, graphed=int(agent._graph is not None),
                       peer_timeouts=(sync.transport.peer_status()[0] if peer else 0), **{f"result_{k}": v for k, v in result.items()})
        elif mode == "ppo_cnn":
            c = PPO_CNN_CFG
            agent = ppo_cnn_agent(c["W"], c["B"], use_graph=peer)
            if rank == 1:  # attach must overwrite rank 1's weights with rank 0's
                agent._net.params.add_(0.01)
            sync = attach_data_parallel(agent, dist)
            assert sync.transport.kind in kinds
            results = []
            for it in range(3 if peer else 1):  # peer: eager, capture + replay, replay -- the collectives inside the learn() graph
                if it:
                    agent.time_t = agent.learn_stamp = 0
                np.random.seed(200 + rank)
                results.append(agent.process(ppo_cnn_rows(rank), c["T"]))
            torch.cuda.synchronize()
            M = c["W"] * c["T"]
            n_upd = c["E"] * (M // c["B"])
            res = dict(params=agent._net.params.cpu().numpy(), perms=agent._static["idx"].cpu().numpy().reshape(c["E"], M), stats=agent._static["stats"].cpu().numpy(),
                       grads=agent._net.grads.cpu().numpy(), n_upd=n_upd, graphed=int(agent._graph is not None), peer_timeouts=(sync.transport.peer_status()[0] if peer else 0))
        else:
            c = RB_CFG
            agent = (apex_agent if mode == "apex" else rainbow_agent)(c["B"], c["N"], use_graph=peer)
            cols, prio = rainbow_shard(rank)
            agent.memory.store_soa(cols, priorities=prio)
            N = c["N"]
            sync = attach_data_parallel(agent, dist)
            assert sync.transport.kind in kinds and agent.memory._shards is not None
            np.random.seed(300 + rank)
            torch.manual_seed(40 + rank)  # the ranks draw DIFFERENT noise: the averaged gradient still gives identical weights
            losses = []
            for it in range(3):
                tree_before = agent.memory.sum_tree.copy()
                count_before = agent.memory.buffer_counter
                losses.append(agent.learn()["loss"])
            torch.cuda.synchronize()
            st = agent._static
            res = dict(params=agent._net.params.cpu().numpy(), target=agent._net.target.cpu().numpy(), idx=st["idx"].cpu().numpy(), w=st["w"].cpu().numpy(),
                       tree_before=tree_before, count_before=count_before, beta=agent.beta, usp=agent.uniform_sample_prob, losses=np.asarray(losses), N=N,
                       tree_after=agent.memory.sum_tree.copy(), peer_timeouts=(sync.transport.peer_status()[0] if peer else 0))
        dist.barrier()
    finally:
        dist.destroy_process_group()
    np.savez(out, **res)
    print(f"dp_worker {mode} rank {rank} ok (transport {sync.transport.kind})")


if __name__ == "__main__":
    main()
