"""dm_env surface used by the package (reference: `import dm_env`,
moog/environment.py:8).  The real package is used when importable; otherwise
this minimal equivalent of StepType / TimeStep / specs is used."""
import collections
import enum

import numpy as np

try:  # pragma: no cover
    import dm_env as _real
    from dm_env import specs  # noqa: F401
    StepType = _real.StepType
    TimeStep = _real.TimeStep
except ImportError:
    class StepType(enum.IntEnum):
        FIRST = 0
        MID = 1
        LAST = 2

    class TimeStep(collections.namedtuple(
            'TimeStep', ['step_type', 'reward', 'discount', 'observation'])):
        __slots__ = ()

        def first(self):
            return self.step_type == StepType.FIRST

        def mid(self):
            return self.step_type == StepType.MID

        def last(self):
            return self.step_type == StepType.LAST

    class _Specs(object):
        class Array(object):
            def __init__(self, shape, dtype, name=None):
                self.shape, self.dtype, self.name = tuple(shape), np.dtype(dtype), name

        class BoundedArray(Array):
            def __init__(self, shape, dtype, minimum, maximum, name=None):
                super().__init__(shape, dtype, name)
                self.minimum, self.maximum = np.asarray(minimum), np.asarray(maximum)

        class DiscreteArray(BoundedArray):
            def __init__(self, num_values, dtype=np.int32, name=None):
                super().__init__((), dtype, 0, num_values - 1, name)
                self.num_values = num_values

    specs = _Specs
