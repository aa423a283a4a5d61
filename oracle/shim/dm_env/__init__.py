"""Minimal stand-in for the `dm_env` package (TEST INFRASTRUCTURE ONLY).

The reference (`/root/reference/moog`) imports `dm_env` (environment.py:8) but the
package is not installed in this image.  This shim provides just the symbols the
reference touches so that it can be imported *in the build container* to
generate golden vectors (tests/golden/make_golden.py).  It is never imported by
the product package.
"""
import abc
import collections
import enum

from . import specs  # noqa: F401


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


class Environment(abc.ABC):
    pass


def restart(observation):
    return TimeStep(StepType.FIRST, None, None, observation)


def transition(reward, observation, discount=1.0):
    return TimeStep(StepType.MID, reward, discount, observation)


def termination(reward, observation):
    return TimeStep(StepType.LAST, reward, 0.0, observation)
