"""Multi-agent to single-agent wrapper (reference: moog/env_wrappers/multi_agent.py:12-52): the caller controls one
key of a Composite action space, agent objects (`step(observation) -> action`) fill in the others.

Works over `Environment` (one env, as in the reference's demo) and over `BatchedEnvironment` (the agents then see
batched observations and must return batched actions)."""
from .abstract_wrapper import AbstractEnvironmentWrapper


class MultiAgentEnvironment(AbstractEnvironmentWrapper):
    def __init__(self, environment, agent_name, **other_agents):
        super(MultiAgentEnvironment, self).__init__(environment)
        self._agent_name = agent_name
        self._other_agents = other_agents

    def step(self, action):
        obs = self.observation()
        actions = {key: agent.step(obs) for key, agent in self._other_agents.items()}
        actions[self._agent_name] = action
        return self._environment.step(actions)
