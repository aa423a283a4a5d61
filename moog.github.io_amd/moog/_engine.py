"""ctypes binding of the HIP engine's C ABI (include/moog_engine.h).

There is deliberately no fallback: if `lib/libmoog_hip.so` is missing or fails
to load, `load_library()` raises -- the product path never runs on the CPU.
"""
import ctypes
import os

from . import _abi

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.normpath(os.path.join(_HERE, '..', 'lib', 'libmoog_hip.so'))

SYMBOLS = (
    'moog_abi_version', 'moog_source_digest', 'moog_last_error', 'moog_program_sizeof', 'moog_engine_create',
    'moog_engine_destroy', 'moog_engine_layout', 'moog_engine_load_state', 'moog_engine_reset',
    'moog_engine_step', 'moog_engine_physics_only', 'moog_engine_render',
    'moog_engine_set_timing', 'moog_engine_kernel_time', 'moog_engine_set_schedule',
    'moog_engine_set_debug', 'moog_engine_static_prefix', 'moog_engine_poll_faults',
    'moog_engine_layer_usage', 'moog_engine_set_action_dtype',
    'moog_engine_read_watch', 'moog_engine_set_reset_pool', 'moog_engine_get_reset_pool',
    'moog_engine_env_prefix', 'moog_engine_set_color_override',
    'moog_engine_kernel_variant', 'moog_engine_raster_path', 'moog_engine_read_draw_records', 'moog_program_step_kernel', 'moog_engine_step_kernel',
)

_LIB = None


class EngineError(RuntimeError):
    pass


def load_library(path=None):
    """Loads the HIP engine shared library and declares every prototype."""
    global _LIB
    if _LIB is not None and path is None:
        return _LIB
    path = path or os.environ.get('MOOG_HIP_LIB', LIB_PATH)
    # torch bundles its own HIP runtime (torch/lib/libamdhip64.so); it must be the
    # one already resident when the engine library resolves libamdhip64, so that
    # both share one runtime (streams, device pointers).
    import torch  # noqa: F401
    # (Kernels of different streams run beside each other only on different hardware queues, and the HIP runtime's default is
    #  4 for the whole process -- GPU_MAX_HW_QUEUES, read when the runtime initialises.  The asynchronous sub-batches want 2
    #  per sub-batch, the reset pool's fills use what is there (queues - 2, at most 8).  The library does not touch the
    #  variable: a process that wants more sets it before anything initialises HIP, as bench.py and the tools do.)
    if not os.path.exists(path):
        raise EngineError(
            'HIP engine library not found at %s -- build it with '
            '`python __graft_entry__.py` (hipcc --offload-arch=gfx950); there is no CPU fallback'
            % path)
    lib = ctypes.CDLL(path)
    vp, i32, i64, u64 = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64, ctypes.c_uint64
    lib.moog_abi_version.restype = ctypes.c_int
    lib.moog_source_digest.restype = ctypes.c_uint64
    lib.moog_last_error.restype = ctypes.c_char_p
    lib.moog_program_sizeof.restype = i64
    lib.moog_engine_create.argtypes = [ctypes.POINTER(_abi.Program), i32, i32, u64, i64,
                                       ctypes.POINTER(vp)]
    lib.moog_engine_destroy.argtypes = [vp]
    lib.moog_engine_layout.argtypes = [vp, ctypes.POINTER(_abi.Layout)]
    lib.moog_engine_load_state.argtypes = [vp, ctypes.POINTER(_abi.StateView)]
    lib.moog_engine_reset.argtypes = [vp, vp, ctypes.POINTER(_abi.Inject),
                                      ctypes.POINTER(_abi.StepOut), vp]
    lib.moog_engine_step.argtypes = [vp, vp, ctypes.POINTER(_abi.Inject),
                                     ctypes.POINTER(_abi.StepOut), vp]
    lib.moog_engine_physics_only.argtypes = [vp, ctypes.POINTER(_abi.Inject), vp]
    lib.moog_engine_render.argtypes = [vp, vp, vp]
    lib.moog_engine_set_schedule.argtypes = [vp, vp, vp]
    lib.moog_engine_set_timing.argtypes = [vp, i32]
    lib.moog_engine_set_reset_pool.argtypes = [vp, i32]
    lib.moog_engine_env_prefix.argtypes = [vp, ctypes.POINTER(i32)]
    lib.moog_engine_set_color_override.argtypes = [vp, vp]
    lib.moog_engine_kernel_variant.argtypes = [vp, ctypes.POINTER(i32), ctypes.POINTER(i32)]
    lib.moog_engine_raster_path.argtypes = [vp, ctypes.POINTER(i32)]
    lib.moog_engine_read_draw_records.argtypes = [vp, vp, i64, ctypes.POINTER(i64), ctypes.POINTER(i32)]
    lib.moog_program_step_kernel.argtypes = [ctypes.POINTER(_abi.Program), ctypes.POINTER(i32), ctypes.POINTER(i32), ctypes.POINTER(u64)]
    lib.moog_engine_step_kernel.argtypes = [vp, ctypes.POINTER(i32)]
    lib.moog_engine_get_reset_pool.argtypes = [vp, ctypes.POINTER(i32), ctypes.POINTER(i64)]
    lib.moog_engine_set_action_dtype.argtypes = [vp, i32]
    lib.moog_engine_layer_usage.argtypes = [vp, ctypes.POINTER(i32), ctypes.POINTER(i32)]
    lib.moog_engine_set_debug.argtypes = [vp, i32, i32]
    lib.moog_engine_static_prefix.argtypes = [vp, ctypes.POINTER(i32), vp, vp]
    lib.moog_engine_poll_faults.argtypes = [vp, i32, ctypes.POINTER(i32)]
    lib.moog_engine_kernel_time.argtypes = [vp, i32, ctypes.POINTER(ctypes.c_double),
                                            ctypes.POINTER(i64)]
    for name in SYMBOLS:
        fn = getattr(lib, name)
        if name not in ('moog_last_error', 'moog_program_sizeof', 'moog_abi_version', 'moog_source_digest'):
            fn.restype = ctypes.c_int
    if lib.moog_abi_version() != _abi.MOOG_ABI_VERSION:
        raise EngineError('ABI mismatch: library %d, header %d'
                          % (lib.moog_abi_version(), _abi.MOOG_ABI_VERSION))
    if lib.moog_program_sizeof() != ctypes.sizeof(_abi.Program):
        raise EngineError('moog_program_t size mismatch: library %d, python %d'
                          % (lib.moog_program_sizeof(), ctypes.sizeof(_abi.Program)))
    if path == os.environ.get('MOOG_HIP_LIB', LIB_PATH):
        _LIB = lib
    return lib


def check(lib, rc):
    if rc != 0:
        msg = lib.moog_last_error()
        raise EngineError('engine call failed (%d): %s' % (rc, msg.decode() if msg else '?'))
