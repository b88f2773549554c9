"""Agent factory with the reference's interface (core/agent/__init__.py:32-42): `Agent(name, **cfg)`.
Keys match the reference's auto-registered snake_case class names."""
from .base import BaseAgent
from .dqn import DQN, ApeX, Double, Multistep, PER
from .ppo import PPO
from .rainbow import C51, Rainbow

agent_dict = {"dqn": DQN, "double": Double, "multistep": Multistep, "per": PER, "ape_x": ApeX, "c51": C51, "rainbow": Rainbow, "ppo": PPO}


def Agent(name, *args, **kwargs):
    if not isinstance(name, str):
        raise Exception("### name variable must be string! ###")
    key = name.lower()
    if key not in agent_dict:
        raise Exception(f"### can use only follows {list(agent_dict.keys())}")
    return agent_dict[key](*args, **kwargs)
