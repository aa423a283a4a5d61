"""Drop-in `moog` namespace backed by the MI355X batched step engine.

Same module / class / keyword names as the reference package (jazlab/moog.github.io,
`moog/__init__.py` and the sub-package `__init__`s cited in each module) so that
existing `get_config()` recipes load unchanged; the components here only record
their parameters, `environment.BatchedEnvironment` lowers them to a
`moog_program_t` (include/moog_engine.h) and all stepping, collision response,
game rules, rewards and rendering run in the HIP engine (csrc/).  There is no CPU
execution path: creating an environment without the compiled HIP library fails.
"""
__version__ = '0.1.0'
