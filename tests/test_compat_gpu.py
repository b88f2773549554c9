"""Which of the reference's configurations of the hot-path agents construct on libjorldy_hip, and that every other one fails at
CONSTRUCTION with the list of eligible configurations (the torch fallback is gone, so an unsupported configuration
must say so up front, not misbehave later).  The table is INTEGRATION.md's "Compatibility" section, one row per case here.
Reference configs: /root/reference/jorldy/config/{ppo,dqn,double,per,multistep,c51,rainbow,ape_x}/*.py."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ADAM = {"name": "adam", "lr": 1e-4}

# (label, agent, kwargs) -- constructs and acts
SUPPORTED = [
    ("config.ppo.cartpole", "ppo", dict(state_size=4, action_size=2, network="discrete_policy_value")),
    ("config.ppo.mountaincar", "ppo", dict(state_size=2, action_size=3, network="discrete_policy_value")),
    ("config.ppo.pendulum", "ppo", dict(state_size=3, action_size=1, network="continuous_policy_value")),
    ("config.ppo.mujoco hopper", "ppo", dict(state_size=11, action_size=3, network="continuous_policy_value")),
    ("config.ppo.mujoco half_cheetah / walker", "ppo", dict(state_size=17, action_size=6, network="continuous_policy_value")),
    ("config.ppo.mujoco ant", "ppo", dict(state_size=111, action_size=8, network="continuous_policy_value")),
    ("config.ppo.mujoco humanoid", "ppo", dict(state_size=376, action_size=17, network="continuous_policy_value")),
    ("config.ppo.pong_mlagent", "ppo", dict(state_size=8, action_size=3, network="discrete_policy_value")),
    ("config.ppo.hopper_mlagent", "ppo", dict(state_size=19, action_size=3, network="continuous_policy_value")),
    ("config.ppo.atari (head cnn, round 6)", "ppo", dict(state_size=(4, 84, 84), action_size=6, network="discrete_policy_value", head="cnn")),
    ("config.ppo.procgen (head cnn, round 6)", "ppo", dict(state_size=(3, 64, 64), action_size=15, network="discrete_policy_value", head="cnn")),
    ("config.dqn.cartpole", "dqn", dict(state_size=4, action_size=2)),
    ("config.dqn.atari", "dqn", dict(state_size=(4, 84, 84), action_size=6, head="cnn")),
    ("config.dqn.procgen", "dqn", dict(state_size=(3, 64, 64), action_size=15, head="cnn")),
    ("config.double.cartpole", "double", dict(state_size=4, action_size=2)),
    ("config.per.cartpole", "per", dict(state_size=4, action_size=2)),
    ("config.multistep.cartpole", "multistep", dict(state_size=4, action_size=2, n_step=4)),
    ("config.c51.cartpole", "c51", dict(state_size=4, action_size=2)),
    ("config.rainbow.cartpole", "rainbow", dict(state_size=4, action_size=2)),
    ("config.rainbow.atari", "rainbow", dict(state_size=(4, 84, 84), action_size=4, head="cnn")),
    ("config.ape_x.atari", "ape_x", dict(state_size=(4, 84, 84), action_size=6, head="cnn", network="dueling", optim_config={"name": "rmsprop", "lr": 6.25e-5, "eps": 1.5e-7, "centered": True})),
]

# (label, agent, kwargs, fragment of the message) -- raises ValueError when constructed
UNSUPPORTED = [
    ("ppo on the cnn head with a continuous policy", "ppo", dict(state_size=(4, 84, 84), action_size=3, network="continuous_policy_value", head="cnn"), "CNN head"),
    ("ppo on the cnn head with RMSprop", "ppo", dict(state_size=(4, 84, 84), action_size=3, network="discrete_policy_value", head="cnn", optim_config={"name": "rmsprop", "lr": 1e-3}), "CNN head"),
    ("config.ppo.drone_delivery_mlagent (head multi, list-valued state)", "ppo", dict(state_size=[(6, 64, 84), 95], action_size=3, network="continuous_policy_value", head="multi"), "head='mlp'"),
    ("ppo with an optimizer other than Adam", "ppo", dict(state_size=4, action_size=2, optim_config={"name": "rmsprop", "lr": 1e-3}), "optim_config name 'adam'"),
    ("ppo with Adam weight decay", "ppo", dict(state_size=4, action_size=2, optim_config={"name": "adam", "lr": 1e-3, "weight_decay": 1e-2}), "weight_decay"),
    ("ppo hidden_size not a multiple of 16", "ppo", dict(state_size=4, action_size=2, hidden_size=100), "hidden_size % 16"),
    ("ppo with more than 40 head outputs", "ppo", dict(state_size=30, action_size=21, network="continuous_policy_value"), "<= 40 head outputs"),
    ("backend='torch' (the mirror of rounds 1-3)", "ppo", dict(state_size=4, action_size=2, backend="torch"), "one backend"),
    ("dqn hidden_size not a multiple of 4", "dqn", dict(state_size=4, action_size=2, hidden_size=30), "hidden_size % 4"),
    ("dqn head multi", "dqn", dict(state_size=[(3, 64, 64), 10], action_size=3, head="multi"), "head 'mlp'"),
    ("dqn with SGD", "dqn", dict(state_size=4, action_size=2, optim_config={"name": "sgd", "lr": 1e-3}), "optim_config"),
    ("rainbow with an unknown noise type", "rainbow", dict(state_size=4, action_size=2, noise_type="gaussian_process"), "noise"),
    ("backend='torch' for the DQN family", "dqn", dict(state_size=4, action_size=2, backend="torch"), "one backend"),
]


@pytest.mark.parametrize("label,name,kw", SUPPORTED, ids=[c[0] for c in SUPPORTED])
def test_reference_config_constructs_and_acts(label, name, kw):
    from jorldy_amd.core.agent import Agent

    torch.manual_seed(0)
    np.random.seed(0)
    kw = dict(dict(hidden_size=64, optim_config=ADAM, buffer_size=64, batch_size=8, device="cuda"), **kw)
    agent = Agent(name, **kw)
    assert agent.backend == "native"
    S = kw["state_size"]
    state = np.random.randint(0, 256, size=(2,) + tuple(S), dtype=np.uint8) if isinstance(S, tuple) else np.random.randn(2, S).astype(np.float32)
    a = agent.act(state, True)["action"]
    assert a.shape[0] == 2 and np.all(np.isfinite(a))


@pytest.mark.parametrize("label,name,kw,fragment", UNSUPPORTED, ids=[c[0] for c in UNSUPPORTED])
def test_unsupported_configuration_raises_at_construction(label, name, kw, fragment):
    from jorldy_amd.core.agent import Agent

    kw = dict(dict(hidden_size=64, optim_config=ADAM, device="cuda"), **kw)
    with pytest.raises(ValueError) as e:
        Agent(name, **kw)
    assert fragment in str(e.value), str(e.value)
    assert "libjorldy_hip" in str(e.value)


def test_container_networks_refuse_a_forward():
    """core/network holds the reference's parameter containers (names, shapes, initialisation); computing with them is the library's job."""
    from jorldy_amd.core.network import Network

    net = Network("discrete_policy_value", 4, 2, D_hidden=32)
    with pytest.raises((RuntimeError, NotImplementedError)):
        net(torch.zeros(1, 4))
