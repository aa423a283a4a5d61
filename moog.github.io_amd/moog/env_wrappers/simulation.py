"""Mental-simulation wrapper (reference: moog/env_wrappers/simulation.py:16-91).

`sim_step(action)` pushes the current environment state on a stack and steps;
`sim_pop(index)` restores the state at that stack level and truncates the stack;
a real `step` first rewinds to the bottom of the stack.  The reference deep-copies
the Python object graph; here a stack level is a copy of the batched engine's two
state records (`BatchedEnvironment.snapshot`), so a tree search over thousands of
environments costs two device-to-device copies per level.

Works over `Environment` (reference semantics, scalar outputs) and over
`BatchedEnvironment` (all envs simulate in lock step; `sim_step` returns None, as
the reference does across an episode boundary, as soon as any env is about to reset).
"""


class SimulationEnvironment(object):
    def __init__(self, environment):
        self._environment = environment
        self._engine = getattr(environment, 'batched', environment)
        self.stack = []

    def __getattr__(self, attr):
        # AbstractEnvironmentWrapper (env_wrappers/abstract_wrapper.py): everything else
        # is the wrapped environment's
        return getattr(self._environment, attr)

    def reset(self):
        self.stack = []
        return self._environment.reset()

    def step(self, action):
        if self.stack:
            self.sim_pop(index=0)
        self.stack = []
        return self._environment.step(action)

    def sim_step(self, action):
        if bool(self._engine.reset_next_step.any().item()):
            return None   # should not simulate across episode boundaries
        self.stack.append(self._engine.snapshot())
        return self._environment.step(action)

    def sim_pop(self, index=-1):
        self._engine.restore(self.stack[index])
        self.stack = self.stack[:index]
