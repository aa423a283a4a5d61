"""`dm_env.specs` stand-in (test infrastructure only, see __init__.py)."""
import numpy as np


class Array(object):
    def __init__(self, shape, dtype, name=None):
        self.shape = tuple(shape)
        self.dtype = np.dtype(dtype)
        self.name = name


class BoundedArray(Array):
    def __init__(self, shape, dtype, minimum, maximum, name=None):
        super().__init__(shape, dtype, name)
        self.minimum = np.asarray(minimum)
        self.maximum = np.asarray(maximum)


class DiscreteArray(BoundedArray):
    def __init__(self, num_values, dtype=np.int32, name=None):
        super().__init__((), dtype, 0, num_values - 1, name)
        self.num_values = num_values
