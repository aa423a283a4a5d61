"""Environment (reference: moog/environment.py:28-158), batched on one MI355X.

`BatchedEnvironment(**config, num_envs=N)` keeps the reference's method names
(`reset`, `step`, `observation`, `observation_spec`, `action_spec`,
`state`-like accessors) with a leading env axis: step_type int32[N], reward
float64[N] (NaN where the reference returns None), discount float64[N],
observation {'image': uint8[N,H,W,3]} -- all torch tensors on the device.
Auto-reset follows environment.py:100-101 per env: the call after a LAST
timestep ignores that env's action and returns a FIRST timestep.

`Environment(**config)` is the single-env facade with numpy/scalar outputs.
"""
import collections
import ctypes

import numpy as np

from . import _abi
from . import _compiler
from . import _dm_env as dm_env
from . import _engine

_FAULT_EXC = (
    (_abi.MOOG_FAULT_SAMPLER_EXHAUSTED, RecursionError,
     'max_recursion_depth exceeded trying to initialize a non-overlapping sprite.'),
    (_abi.MOOG_FAULT_ODD_PORTALS, ValueError, 'There must be an even number of portals.'),
    (_abi.MOOG_FAULT_BAD_NORMAL, ValueError, 'collision_normal_norm is not close to 1.'),
    (_abi.MOOG_FAULT_INJECT_UNDERRUN, RuntimeError, 'injected uniform buffer exhausted.'),
    (_abi.MOOG_FAULT_DIST_EXHAUSTED, ValueError,
     'Maximum number of tried exceeded when trying to sample from a distribution.'),
    (_abi.MOOG_FAULT_PHASE_END, IndexError, 'tuple index out of range (PhaseSequence ran past its last phase).'),
    (_abi.MOOG_FAULT_LAYER_FULL, RuntimeError,
     'a rule appended to a layer whose slot capacity is used up (raise layer_capacity).'),
    (_abi.MOOG_FAULT_TETHER_ZIP, ValueError,
     'All layers fed into TetherAcrossLayers must have the same number of sprites.'),
    (_abi.MOOG_FAULT_OFF_GRID, ValueError, 'Object is not on the maze grid.'),
)


class BatchedEnvironment(object):
    """N independent MOOG environments stepped by one HIP engine handle."""

    def __init__(self, state_initializer, physics, task, action_space, observers, game_rules=(),
                 meta_state_initializer=None, num_envs=1, device=None, seed=0, env_index0=0,
                 layer_capacity=None, keep_sprite_factors=False, reset_pool='auto', specialize=False, _compiled=None,
                 _buffers=None):
        import torch
        self._torch = torch
        self._lib = _engine.load_library()  # raises when the HIP extension is missing
        # specialize: True compiles (once; needs hipcc, not a GPU; ~20 s) a step kernel for this very program -- same source, the
        # program a compile-time constant, bit-identical results, faster (moog/_spec.py).  Whatever the argument, an engine uses
        # such a kernel when one has been built for its program (moog_engine_step_kernel; MOOG_STEP_SPEC=0 turns that off).
        self._specialize = bool(specialize)
        # reset_pool: build every env's NEXT episode beside the step kernels and take it over when the episode ends
        # (moog_engine_set_reset_pool; same results, bit for bit).  'auto': on for configs whose state_initializer plays
        # physics forward (bounce_box_contact_prediction, red_green: a reset there costs as much as a hundred steps of the
        # whole batch) when the engine can (see self.reset_pool); True: on, or EngineError; False: off.
        self._reset_pool_arg = reset_pool
        if not torch.cuda.is_available():
            raise _engine.EngineError('no HIP device available: the MOOG engine has no CPU path')
        # (_compiled / _buffers: a SubBatchedEnvironment builds its parts over one lowered program and slices of one set of
        #  tensors)
        # layer_capacity: {layer: slots} for the layers rules append to (the reference's lists are unbounded,
        # create_sprites.py:34, change_layer.py:43; the engine's are slots), None for the default (initial count + 8), or
        # 'auto' / {'auto': True, layer: initial slots, ...}: start from those capacities and GROW -- whenever a step leaves
        # a layer's high-water mark (moog_engine_layer_usage) within a quarter of its capacity, the engine is re-created with
        # that layer doubled and every env's records are moved to the new layout before the next call (_grow_layers).
        # 'auto' also FITS, once: after `fit_after` calls of step() (default 128; {'auto': True, 'fit_after': k}; 0: never) the
        # layers are sized to what the batch has needed so far, high-water x 1.25 (fit_layer_capacity): a record holds every slot's
        # vertices, so roomy initial capacities cost envs per CU for the whole run otherwise (first_person_predators_prey at
        # {prey: 32, predators: 96}: one env per CU, 0.88 M env-steps/s; fitted: two, 1.6 M).  Growth on demand goes on afterwards.
        self._auto_capacity = False
        self._fit_after, self._n_step_calls = 0, 0
        if layer_capacity == 'auto':
            self._auto_capacity, layer_capacity, self._fit_after = True, None, 128
        elif isinstance(layer_capacity, dict) and layer_capacity.get('auto'):
            self._auto_capacity = True
            self._fit_after = int(layer_capacity.get('fit_after', 128))
            layer_capacity = {k: v for k, v in layer_capacity.items() if k not in ('auto', 'fit_after')}
        self._config_args = (state_initializer, physics, task, action_space, observers, game_rules, meta_state_initializer)
        self._keep_sprite_factors = keep_sprite_factors
        self._seed, self._env_index0 = int(seed), int(env_index0)
        self.compiled = _compiled if _compiled is not None else _compiler.compile_config(
            state_initializer, physics, task, action_space, observers, game_rules,
            meta_state_initializer, layer_capacity=layer_capacity,
            keep_sprite_factors=keep_sprite_factors)
        self.physics = physics
        self.task = task
        self.action_space = action_space
        self.observers = observers
        self.game_rules = game_rules
        self.num_envs = int(num_envs)
        # meta_state (environment.py:60-63,87): a host-side Python object per environment, touched only by
        # `ModifyMetaState` rules (arbitrary Python, never sprites): one object for a batch of one, a list of
        # num_envs objects otherwise.  Those rules run in a host loop over the envs, and every step then reads the
        # auto-reset flags back (one device-to-host copy per step): convenient, not fast.
        self._host_rules = [r for r in game_rules if getattr(r, 'host_side', False)]
        self._meta_state_initializer = meta_state_initializer
        self._meta_state = None
        self.device = torch.device('cuda', torch.cuda.current_device()) if device is None \
            else torch.device(device)
        P, L = self.compiled.program, self.compiled.layout
        self.layout = L
        n = self.num_envs
        if _buffers is not None:
            (self.state_f64, self.state_i32, self.reward, self.discount, self.step_type, self.image) = _buffers
        else:
            with torch.cuda.device(self.device):
                (self.state_f64, self.state_i32, self.reward, self.discount, self.step_type,
                 self.image) = self.allocate_buffers(torch, L, P, n, self.device)
        self._handle = ctypes.c_void_p()
        if self._specialize:
            from . import _spec
            _spec.build(P)
        dev_index = self.device.index if self.device.index is not None else torch.cuda.current_device()
        with torch.cuda.device(self.device):   # (the engine calls hipSetDevice: keep torch's current device)
            _engine.check(self._lib, self._lib.moog_engine_create(
                ctypes.byref(P), n, dev_index, int(seed), int(env_index0),
                ctypes.byref(self._handle)))
        view = _abi.StateView()
        view.f64 = ctypes.cast(self.state_f64.data_ptr(), ctypes.POINTER(ctypes.c_double))
        view.i32 = ctypes.cast(self.state_i32.data_ptr(), ctypes.POINTER(ctypes.c_int32))
        _engine.check(self._lib, self._lib.moog_engine_load_state(self._handle, ctypes.byref(view)))
        self._out = _abi.StepOut()
        self._out.reward = ctypes.cast(self.reward.data_ptr(), ctypes.POINTER(ctypes.c_double))
        self._out.discount = ctypes.cast(self.discount.data_ptr(), ctypes.POINTER(ctypes.c_double))
        self._out.step_type = ctypes.cast(self.step_type.data_ptr(), ctypes.POINTER(ctypes.c_int32))
        self._out.image = ctypes.cast(self.image.data_ptr(), ctypes.POINTER(ctypes.c_uint8))
        self._composite = hasattr(action_space, 'action_spaces')
        self._n_actions = max(1, int(P.n_actions))
        self._is_grid = (not self._composite) and P.action.kind == _abi.MOOG_ACTION_GRID
        self._dynamic_layers = any(P.layer_dynamic[i] for i in range(P.n_layers))
        # Device-side faults (the reference's exceptions): True = deferred -- the kernels OR fault bits
        # into a host-visible word that every call polls first, so a fault raises at the latest in the
        # call after the one that caused it, without a per-step synchronisation; 'sync' = also
        # synchronise and check after every call (the exception comes from the call itself);
        # False = never check.  Runs with injected uniforms / run-time sprite creation check at once.
        self.check_faults = True
        self._cost = self._perm = None
        self._action_f32 = False
        self._apply_reset_pool()
        self._setup_color_fn()

    def _setup_color_fn(self):
        """PILRenderer(color_to_rgb=<a callable>): the callable is evaluated here, on the host, once per distinct colour
        triple; the rasteriser takes the results per (env, sprite slot) from `self._rgb` (moog_engine_set_color_override).
        The engine's calls are then split -- step without frames, colours, frames -- and one small device-to-host read per
        call asks whether any colour changed: convenient, not fast (like the host-side ModifyMetaState rules)."""
        from .observers import pil_renderer
        ren = [o for o in self.observers.values() if isinstance(o, pil_renderer.PILRenderer)]
        self._color_fn = ren[0].color_to_rgb if ren and ren[0]._cmap == 'callable' else None
        if self._color_fn is None:
            return
        torch = self._torch
        S = self.layout.S
        self._rgb = torch.zeros((self.num_envs, S), dtype=torch.int32, device=self.device)
        self._rgb_seen = None          # colour bit patterns the entries of _rgb were computed from
        self._rgb_done = torch.zeros((self.num_envs, S), dtype=torch.bool, device=self.device)
        self._rgb_cache = {}
        _engine.check(self._lib, self._lib.moog_engine_set_color_override(self._handle, ctypes.c_void_p(self._rgb.data_ptr())))
        self._out_noimg = _abi.StepOut()
        self._out_noimg.reward, self._out_noimg.discount = self._out.reward, self._out.discount
        self._out_noimg.step_type = self._out.step_type

    @staticmethod
    def _call_color_fn(fn, triple):
        """color_to_rgb on one colour triple, the way pil_renderer.py:108-110 and Pillow's ink conversion treat the result:
        three integers, clipped to 0 .. 255.  Components that hold an integral value are passed as Python ints (a Sprite
        keeps the numbers it was given; the engine's records hold float64)."""
        import numbers
        args = tuple(int(v) if float(v).is_integer() else float(v) for v in triple)
        out = list(fn(args))
        if len(out) != 3:
            raise ValueError('color_to_rgb must return three components, got %r' % (out,))
        rgb = 0
        for k, v in enumerate(out):
            if not isinstance(v, (numbers.Integral, np.integer)):
                raise TypeError('color_to_rgb returned %r: Pillow takes integer colour components' % (v,))
            rgb |= min(255, max(0, int(v))) << (8 * k)
        return rgb

    def _refresh_colors(self):
        torch = self._torch
        L, S = self.layout, self.layout.S
        col = self.state_f64[:, L.o_color:L.o_color + 3 * S].view(torch.int64).view(self.num_envs, S, 3)
        alive = (self.state_i32[:, L.o_flags:L.o_flags + S] & _abi.MOOG_F_ALIVE) != 0
        if self._rgb_seen is None:
            need = alive
        else:
            need = alive & ((col != self._rgb_seen).any(-1) | ~self._rgb_done)
        idx = need.nonzero()            # (the one synchronising read of the call)
        if idx.shape[0]:
            bits = col[idx[:, 0], idx[:, 1]].cpu().numpy()
            vals = bits.view(np.float64)
            out = np.empty(len(bits), np.int32)
            for i in range(len(bits)):
                key = bits[i].tobytes()
                rgb = self._rgb_cache.get(key)
                if rgb is None:
                    rgb = self._rgb_cache[key] = self._call_color_fn(self._color_fn, vals[i])
                out[i] = rgb
            self._rgb[idx[:, 0], idx[:, 1]] = torch.as_tensor(out, device=self.device)
            self._rgb_done[idx[:, 0], idx[:, 1]] = True
        # (a slot without a live sprite forgets what it held: the sprite that comes to life there is evaluated again even if a
        #  host-side edit gave it the colour bits the snapshot already has)
        self._rgb_done &= alive
        self._rgb_seen = col.clone()

    def _render_with_colors(self):
        self._refresh_colors()
        with self._torch.cuda.device(self.device):
            _engine.check(self._lib, self._lib.moog_engine_render(
                self._handle, ctypes.c_void_p(self.image.data_ptr()), self._stream()))

    def _apply_reset_pool(self):
        want = self._reset_pool_arg
        P = self.compiled.program
        if want == 'auto':
            want = any(P.ops[o].cell_sel == _abi.MOOG_CELL_SIMULATE for o in range(P.n_ops))
            strict = False
        else:
            strict = bool(want)
        self.reset_pool_refusal = None
        if want:
            with self._torch.cuda.device(self.device):
                rc = self._lib.moog_engine_set_reset_pool(self._handle, 1)
            if rc != 0:
                if strict:
                    _engine.check(self._lib, rc)
                self.reset_pool_refusal = self._lib.moog_last_error().decode()

    @property
    def reset_pool(self):
        """The engine's reset pool: {'on', 'fills' (fill launches), 'adopted' (episodes opened from the pool), 'in_place'
        (episodes opened by a reset inside the step kernel), 'rejected' (pool records dropped by the input check), 'waited'
        (take-overs that waited for a fill)}; synchronises.  `reset_pool_refusal` holds the engine's reason when
        reset_pool='auto' wanted the pool and could not have it."""
        on, st = ctypes.c_int32(), (ctypes.c_int64 * 5)()
        _engine.check(self._lib, self._lib.moog_engine_get_reset_pool(self._handle, ctypes.byref(on), st))
        return dict(on=bool(on.value), fills=int(st[0]), adopted=int(st[1]), in_place=int(st[2]), rejected=int(st[3]),
                    waited=int(st[4]))

    @property
    def kernel_variant(self):
        """(step-kernel variant 0 / 1 / 2, late reset) -- include/moog_engine.h moog_engine_kernel_variant."""
        v, late = ctypes.c_int32(), ctypes.c_int32()
        _engine.check(self._lib, self._lib.moog_engine_kernel_variant(self._handle, ctypes.byref(v), ctypes.byref(late)))
        return int(v.value), bool(late.value)

    @property
    def env_prefix_slots(self):
        """Leading sprite slots the rasteriser keeps in a cached picture per env (moog_engine_env_prefix); 0: unused."""
        v = ctypes.c_int32()
        _engine.check(self._lib, self._lib.moog_engine_env_prefix(self._handle, ctypes.byref(v)))
        return int(v.value)

    @staticmethod
    def allocate_buffers(torch, L, P, n, device):
        """The state records and step outputs of n envs (the engine borrows their device pointers)."""
        return (torch.zeros((n, L.f64_per_env), dtype=torch.float64, device=device),
                torch.zeros((n, L.i32_per_env), dtype=torch.int32, device=device),
                torch.full((n,), float('nan'), dtype=torch.float64, device=device),
                torch.full((n,), float('nan'), dtype=torch.float64, device=device),
                torch.zeros((n,), dtype=torch.int32, device=device),
                torch.zeros((n, P.render.height, P.render.width, 3), dtype=torch.uint8, device=device))

    def enable_cost_schedule(self, enabled=True):
        """Launch the step kernel's workgroups in order of descending per-env cost of the
        previous step (longest-processing-time first): the envs with clustered contacts
        start first instead of landing in the under-filled tail of the launch.  The engine
        re-sorts the order after every step on a side stream.  A pure scheduling hint --
        results are identical."""
        torch = self._torch
        if enabled:
            self._cost = torch.zeros((self.num_envs,), dtype=torch.float32, device=self.device)
            self._perm = torch.arange(self.num_envs, dtype=torch.int32, device=self.device)
            with torch.cuda.device(self.device):
                _engine.check(self._lib, self._lib.moog_engine_set_schedule(
                    self._handle, ctypes.c_void_p(self._perm.data_ptr()),
                    ctypes.c_void_p(self._cost.data_ptr())))
        else:
            self._cost = self._perm = None
            _engine.check(self._lib, self._lib.moog_engine_set_schedule(self._handle, None, None))

    def step_kernel(self):
        """'specialised' when the engine steps with a kernel compiled for this program (moog/_spec.py), else 'generic'."""
        v = ctypes.c_int32()
        _engine.check(self._lib, self._lib.moog_engine_step_kernel(self._handle, ctypes.byref(v)))
        return 'specialised' if v.value else 'generic'

    def raster_path(self):
        """Which rasteriser draws this engine's frames: 'mask' (csrc/moog_raster_mask_core.h) or 'spans'
        (csrc/moog_raster_kernel.h) -- moog_engine_raster_path."""
        v = ctypes.c_int32()
        _engine.check(self._lib, self._lib.moog_engine_raster_path(self._handle, ctypes.byref(v)))
        return 'mask' if (v.value & 1) else 'spans'

    def draw_records(self):
        """The frames' draw records as the last launch left them (include/moog_engine.h moog_engine_read_draw_records;
        csrc/moog_draw_record.h): (uint8 array [num_envs, stride], written_by_the_step_kernel).  Synchronises; for tests."""
        import numpy as np
        stride, in_step = ctypes.c_int64(), ctypes.c_int32()
        _engine.check(self._lib, self._lib.moog_engine_read_draw_records(self._handle, None, 0, ctypes.byref(stride), ctypes.byref(in_step)))
        out = np.zeros((self.num_envs, stride.value), np.uint8)
        with self._torch.cuda.device(self.device):
            _engine.check(self._lib, self._lib.moog_engine_read_draw_records(
                self._handle, ctypes.c_void_p(out.ctypes.data), out.nbytes, ctypes.byref(stride), ctypes.byref(in_step)))
        return out, bool(in_step.value)

    def raster_compact_edges(self):
        """Whether the mask rasteriser keeps this program's edge records in their 4-byte form (csrc/moog_raster_mask_core.h
        RmEdgesCompact: picked when the 16-byte records would keep frames off a CU; MOOG_RASTER_COMPACT=0 / 1 forces)."""
        v = ctypes.c_int32()
        _engine.check(self._lib, self._lib.moog_engine_raster_path(self._handle, ctypes.byref(v)))
        return v.value == 3

    # -- plumbing ---------------------------------------------------------------------
    def _stream(self):
        return ctypes.c_void_p(self._torch.cuda.current_stream(self.device).cuda_stream)

    def _inject(self, uniforms):
        if uniforms is None:
            return None, None
        t = self._torch.as_tensor(uniforms, dtype=self._torch.float64, device=self.device).contiguous()
        assert t.dim() == 2 and t.shape[0] == self.num_envs
        inj = _abi.Inject()
        inj.uniforms = ctypes.cast(t.data_ptr(), ctypes.POINTER(ctypes.c_double))
        inj.per_env = t.shape[1]
        return inj, t

    def _timestep(self):
        return dm_env.TimeStep(self.step_type, self.reward, self.discount, self._observation())

    def _observation(self):
        from .observers import raw_state
        obs = {}
        for key, o in self.observers.items():   # the reference's dict order
            if isinstance(o, raw_state.RawState):
                obs[key] = raw_state.StateView(self)
            else:
                obs[key] = self.image
        return obs


    def _poll_faults(self):
        if not self.check_faults:
            return
        bits = ctypes.c_int32()
        _engine.check(self._lib, self._lib.moog_engine_poll_faults(self._handle, 0, ctypes.byref(bits)))
        if bits.value:
            self._torch.cuda.synchronize(self.device)
            self.raise_faults()

    def clear_faults(self):
        """Forgets the recorded faults (the per-env fault words and the engine's summary word)."""
        self._torch.cuda.synchronize(self.device)
        self.state_i32[:, self.layout.o_fault] = 0
        bits = ctypes.c_int32()
        _engine.check(self._lib, self._lib.moog_engine_poll_faults(self._handle, 1, ctypes.byref(bits)))

    def layer_usage(self):
        """Usage of the layers that rules append to (the reference's unbounded lists; here `layer_capacity` slots):
        {layer: dict(capacity=, high_water=, dropped=)} over all envs and calls since the environment was made --
        high_water is the most sprites an append ever needed room for (capacity + 1 once the layer overflowed), dropped
        the appends that found the layer full.  The sizing hint for `layer_capacity`.  Synchronises the device."""
        P = self.compiled.program
        hw = (ctypes.c_int32 * _abi.MOOG_MAX_LAYERS)()
        dr = (ctypes.c_int32 * _abi.MOOG_MAX_LAYERS)()
        self._torch.cuda.synchronize(self.device)
        _engine.check(self._lib, self._lib.moog_engine_layer_usage(self._handle, hw, dr))
        return {name: dict(capacity=int(P.layer_nslots[li]), high_water=int(hw[li]), dropped=int(dr[li]))
                for li, name in enumerate(self.compiled.layer_names) if P.layer_dynamic[li]}

    def _grow_if_close(self):
        """layer_capacity='auto': grows the layers whose high-water mark has come close to their capacity -- within a quarter of
        it: the layer is doubled; after fit_layer_capacity(headroom=h), within h / 2 of it: the layer grows by the factor 1 + h.
        Nothing has been dropped at that point (a dropped append raises, as ever), so the state moves over intact."""
        import math
        m = getattr(self, '_grow_margin', 0.25)
        factor = 2.0 if m >= 0.25 else 1.0 + 2.0 * m
        use = self.layer_usage()
        grow = {name: max(int(math.ceil(factor * u['capacity'])), u['high_water'] + 4) for name, u in use.items()
                if u['dropped'] == 0 and u['capacity'] - u['high_water'] <= max(2.0 if m < 0.25 else 0.0, m * u['capacity'])}
        if grow:
            self._grow_layers(grow)

    def fit_layer_capacity(self, headroom=0.25, min_room=4):
        """Sizes the layers that rules append to by what the batch has needed so far instead of by their capacity: every such
        layer gets `high-water mark x (1 + headroom)` slots (at least `min_room` more than the mark, never fewer than the last
        slot in use in any env), the engine is re-created and the records move over as in _grow_layers.  From then on the
        layers grow on demand (layer_capacity='auto').  A state record holds every slot's vertices, so its size -- and with it
        how many envs share a CU's LDS in the step kernel -- follows the capacities: first_person_predators_prey at
        {'prey': 32, 'predators': 96} is one env per CU, fitted to a 4096-env batch's needs it is two or three.
        Returns the new capacities ({} when nothing shrinks).  Results do not depend on capacities (a sprite keeps its layer
        and its place in the layer's list)."""
        import math
        torch = self._torch
        P, L = self.compiled.program, self.layout
        use = self.layer_usage()
        flags = self.state_i32[:, L.o_flags:L.o_flags + L.S] & 1
        caps = {}
        for li, name in enumerate(self.compiled.layer_names):
            if name not in use:
                continue
            s0, n_old = int(P.layer_slot0[li]), int(P.layer_nslots[li])
            alive = flags[:, s0:s0 + n_old].any(dim=0).nonzero()
            in_use = int(alive.max().item()) + 1 if alive.numel() else 0
            mark = max(use[name]['high_water'], in_use)
            # (never below the slots the initializer fills: a layer whose sprites have all vanished is re-made whole by the
            #  next reset, and compile_config refuses fewer slots than that)
            want = max(mark + int(min_room), int(math.ceil(mark * (1.0 + headroom))), 1, int(self.compiled.layer_n_init.get(name, 0)))
            if want < n_old:
                caps[name] = want
        if caps:
            self._grow_layers(caps)
            self._auto_capacity = True
            self._grow_margin = min(0.25, 0.5 * float(headroom))
        return caps

    def _grow_layers(self, new_caps):
        """Re-creates the engine with other capacities for the given layers ({layer: slots}; fewer slots only when the
        slots given up hold no sprite: fit_layer_capacity) and moves every env's state records to the new layout: a sprite keeps its layer and its position in the layer's list, everything that is not per
        slot (task / rule / action state, counters, the random streams' positions) is copied as is."""
        torch = self._torch
        old_c, old_L = self.compiled, self.layout
        old_P = old_c.program
        caps = {name: int(old_P.layer_nslots[li]) for li, name in enumerate(old_c.layer_names) if old_P.layer_dynamic[li]}
        caps.update({k: int(v) for k, v in new_caps.items()})
        new_c = _compiler.compile_config(*self._config_args, layer_capacity=caps, keep_sprite_factors=self._keep_sprite_factors)
        P, L = new_c.program, new_c.layout
        assert list(new_c.layer_names) == list(old_c.layer_names)
        # Other capacities = another program = another hash: the specialised step kernel of the old program does not fit.
        # specialize=True builds the new program's (~20 s of hipcc, once per capacity set; kept in the spec directory); otherwise the
        # engine falls back to the generic kernels -- said once, so that a slower run is not a mystery (step_kernel() tells).
        was_specialised = self.step_kernel() == 'specialised'
        if self._specialize:
            from . import _spec
            _spec.build(P)
        # index maps: for every word of a new record, the word of the old record it comes from (-1: zero)
        src_f = np.full(L.f64_per_env, -1, np.int64)
        src_q = np.full(L.i32_per_env, -1, np.int64)
        slot_map = []   # (old slot, new slot)
        for li in range(P.n_layers):
            o0, n_old, n0 = int(old_P.layer_slot0[li]), int(old_P.layer_nslots[li]), int(P.layer_slot0[li])
            slot_map += [(o0 + k, n0 + k) for k in range(min(n_old, int(P.layer_nslots[li])))]
        for name, width in (('o_pos', 2), ('o_vel', 2), ('o_angle', 1), ('o_angvel', 1), ('o_mass', 1), ('o_color', 3),
                            ('o_inertia', 2), ('o_maxr', 1), ('o_scale', 1), ('o_aspect', 1)):
            a, b = getattr(old_L, name), getattr(L, name)
            if a < 0 or b < 0:
                assert a < 0 and b < 0, name
                continue
            for so, sn in slot_map:
                src_f[b + width * sn:b + width * (sn + 1)] = np.arange(a + width * so, a + width * (so + 1))
        for so, sn in slot_map:   # vertices
            vo, vn, cap = int(old_P.slot_voff[so]), int(P.slot_voff[sn]), int(old_P.slot_vcap[so])
            assert int(P.slot_vcap[sn]) >= cap
            src_f[L.o_verts + 2 * vn:L.o_verts + 2 * (vn + cap)] = np.arange(old_L.o_verts + 2 * vo, old_L.o_verts + 2 * (vo + cap))
        n_act = 2 * max(1, int(P.n_actions))
        for name, count in (('o_action', n_act), ('o_task', int(P.n_tasks)), ('o_rule', int(P.n_rules)),
                            ('o_hdraw', int(P.n_hdraws)), ('o_rule2', int(P.n_rules))):
            a, b = getattr(old_L, name), getattr(L, name)
            if a >= 0 and b >= 0 and count > 0:
                src_f[b:b + count] = np.arange(a, a + count)
        for name in ('o_flags', 'o_nverts', 'o_opacity', 'o_shape', 'o_tele', 'o_valias', 'o_fmask'):
            a, b = getattr(old_L, name), getattr(L, name)
            if a < 0 or b < 0:
                assert a < 0 and b < 0, name
                continue
            for so, sn in slot_map:
                src_q[b + sn] = a + so
        for name, count in (('o_step_count', 1), ('o_reset_next', 1), ('o_fault', 1), ('o_rng', 4),
                            ('o_maze', _abi.MOOG_MAX_MAZE + _abi.MOOG_MAX_MAZE_POINTS)):
            a, b = getattr(old_L, name), getattr(L, name)
            if a >= 0 and b >= 0:
                src_q[b:b + count] = np.arange(a, a + count)
        torch.cuda.synchronize(self.device)
        with torch.cuda.device(self.device):
            def move(old, src, dtype):
                idx = torch.as_tensor(np.where(src < 0, 0, src), device=self.device)
                keep = torch.as_tensor(src >= 0, device=self.device)
                new = old.index_select(1, idx)
                return torch.where(keep.unsqueeze(0), new, torch.zeros((), dtype=dtype, device=self.device)).contiguous()
            new_f = move(self.state_f64, src_f, torch.float64)
            new_q = move(self.state_i32, src_q, torch.int32)
            # the new engine, over the new records (outputs stay where they are)
            had_schedule, f32 = self._perm is not None, self._action_f32
            self._lib.moog_engine_destroy(self._handle)
            self._handle = ctypes.c_void_p()
            self.compiled, self.layout = new_c, L
            self.state_f64, self.state_i32 = new_f, new_q
            dev_index = self.device.index if self.device.index is not None else torch.cuda.current_device()
            _engine.check(self._lib, self._lib.moog_engine_create(
                ctypes.byref(P), self.num_envs, dev_index, self._seed, self._env_index0, ctypes.byref(self._handle)))
            view = _abi.StateView()
            view.f64 = ctypes.cast(self.state_f64.data_ptr(), ctypes.POINTER(ctypes.c_double))
            view.i32 = ctypes.cast(self.state_i32.data_ptr(), ctypes.POINTER(ctypes.c_int32))
            _engine.check(self._lib, self._lib.moog_engine_load_state(self._handle, ctypes.byref(view)))
            if f32:
                _engine.check(self._lib, self._lib.moog_engine_set_action_dtype(self._handle, 1))
        if had_schedule:
            self.enable_cost_schedule(True)
        self._apply_reset_pool()
        self._setup_color_fn()
        self.capacity_growths = getattr(self, 'capacity_growths', []) + [dict(caps)]
        if was_specialised and self.step_kernel() != 'specialised' and not getattr(self, '_warned_generic', False):
            self._warned_generic = True
            import warnings
            warnings.warn('layer capacities changed to %s: the program-specialised step kernel was built for the old capacities, '
                          'the generic step kernel is in use from here on (BatchedEnvironment(specialize=True) or '
                          '`moog._spec.build(env.compiled.program)` builds one for the new program)' % (dict(caps),), RuntimeWarning)

    def raise_faults(self):
        """Re-raises device-side per-env faults with the reference's exception types.

        Faults are STICKY, like a broken environment in the reference: the per-env fault words (and the engine's summary
        word that _poll_faults reads) stay set, so every later reset() / step() / observation() raises again until
        clear_faults() -- after which the faulted envs should be reset."""
        faults = self.state_i32[:, self.layout.o_fault]
        if not bool((faults != 0).any().item()):
            return
        allbits = 0
        for v in faults.unique().tolist():
            allbits |= int(v)
        for bit, exc, msg in _FAULT_EXC:
            if allbits & bit:
                env = int((faults & bit).nonzero()[0].item())
                if bit == _abi.MOOG_FAULT_LAYER_FULL:   # which layer, and how much room it asked for
                    full = {k: v for k, v in self.layer_usage().items() if v['dropped'] > 0}
                    msg += ' Overflowing layers: %s -- pass layer_capacity={layer: slots} with more than high_water ' \
                           'slots (the reference\'s lists are unbounded; env.layer_usage() reports the demand).' % (full,)
                raise exc('%s (env %d)' % (msg, env))

    # -- dm_env surface (environment.py:82-131) ------------------------------------------
    def reset(self, env_mask=None, injected_uniforms=None):
        self._poll_faults()
        mask_t = None
        mask_ptr = None
        if env_mask is not None:
            mask_t = self._torch.as_tensor(env_mask, device=self.device).to(self._torch.uint8).contiguous()
            mask_ptr = ctypes.c_void_p(mask_t.data_ptr())
        inj, keep = self._inject(injected_uniforms)
        with self._torch.cuda.device(self.device):
            _engine.check(self._lib, self._lib.moog_engine_reset(
                self._handle, mask_ptr, ctypes.byref(inj) if inj else None,
                ctypes.byref(self._out if self._color_fn is None else self._out_noimg), self._stream()))
        if self._color_fn is not None:
            self._render_with_colors()
        if self.check_faults:
            self.raise_faults()   # (a reset is rare and its sampler is where most faults come from: check at once)
        del keep, mask_t
        self._host_reset()
        return self._timestep()

    def step(self, action, injected_uniforms=None):
        torch = self._torch
        self._poll_faults()
        if self._composite:
            a = self._pack_composite(action)
        elif self._is_grid:
            a = torch.as_tensor(action, device=self.device).to(torch.int32).contiguous()
            assert a.shape == (self.num_envs,)
        else:
            a = torch.as_tensor(action, device=self.device)
            # float32 actions -- the dtype of the reference's Joystick spec (joystick.py:42-43) -- stay float32: the
            # reference then multiplies by scaling_factor in float32, and so does the engine
            f32 = a.dtype == torch.float32
            if f32 != self._action_f32:
                _engine.check(self._lib, self._lib.moog_engine_set_action_dtype(self._handle, 1 if f32 else 0))
                self._action_f32 = f32
            a = (a if f32 else a.to(torch.float64)).contiguous()
            assert a.shape == (self.num_envs, 2)
        if self._host_rules or self._meta_state_initializer is not None:
            # environment.py:100-104: an auto-reset re-initialises the meta-state; either way
            # every rule steps once per call
            flags = self.reset_next_step.cpu().numpy()
            if self.num_envs == 1:
                if flags[0]:
                    self._host_reset()
                else:
                    for r in self._host_rules:
                        r.step(None, self._meta_state)
            else:
                if self._meta_state is None:   # stepped from a loaded state without a reset()
                    self._meta_state = [self._new_meta_state() for _ in range(self.num_envs)]
                for i in range(self.num_envs):
                    if flags[i]:
                        self._meta_state[i] = self._new_meta_state()
                    for r in self._host_rules:
                        r.step(None, self._meta_state[i])
        inj, keep = self._inject(injected_uniforms)
        with torch.cuda.device(self.device):
            _engine.check(self._lib, self._lib.moog_engine_step(
                self._handle, ctypes.c_void_p(a.data_ptr()), ctypes.byref(inj) if inj else None,
                ctypes.byref(self._out if self._color_fn is None else self._out_noimg), self._stream()))
        if self._color_fn is not None:
            self._render_with_colors()
        self._last_action = a
        # Programs whose rules append to layers can overflow a layer's capacity at any step:
        # surface that (one host sync per step; set check_faults = False to opt out).
        if self.check_faults == 'sync' or (self.check_faults and (injected_uniforms is not None or self._dynamic_layers)):
            self.raise_faults()
        del keep
        if self._auto_capacity and self._dynamic_layers:
            self._n_step_calls += 1
            if self._fit_after > 0 and self._n_step_calls == self._fit_after:
                self.fit_layer_capacity()   # (once; sets the margin _grow_if_close works with from here on)
            self._grow_if_close()
        return self._timestep()

    def _pack_composite(self, action):
        """Composite actions (composite.py:51-62): a dict {key: tensor [N, 2] or, for a Grid
        sub-space, [N] move indices}, or the packed f64 tensor [N, n_spaces, 2] itself."""
        torch = self._torch
        n, k = self.num_envs, self._n_actions
        if isinstance(action, dict):
            keys = self.action_space.action_keys
            if set(action) != set(keys):
                raise KeyError('composite action keys %s, expected %s' % (sorted(action), keys))
            a = torch.zeros((n, len(keys), 2), dtype=torch.float64, device=self.device)
            for i, key in enumerate(keys):
                v = torch.as_tensor(action[key], device=self.device).to(torch.float64)
                if v.dim() == 1 and v.shape[0] == n:      # Grid move indices
                    a[:, i, 0] = v
                else:
                    a[:, i, :] = v.reshape(n, 2)
        else:
            a = torch.as_tensor(action, device=self.device).to(torch.float64)
        a = a.reshape(n, -1)
        if self._n_actions == 1:   # a Composite of one space is that space's own buffer
            a = a[:, :2]
            if self.compiled.program.action.kind == _abi.MOOG_ACTION_GRID:
                return a[:, 0].to(torch.int32).contiguous()
            return a.contiguous()
        assert a.shape == (n, 2 * k), a.shape
        return a.contiguous()

    def _new_meta_state(self):
        return self._meta_state_initializer() if self._meta_state_initializer is not None else None

    def _host_reset(self):
        if self.num_envs == 1:
            self._meta_state = self._new_meta_state()
            for r in self._host_rules:
                r.step(None, self._meta_state)
            return
        self._meta_state = [self._new_meta_state() for _ in range(self.num_envs)]
        for m in self._meta_state:
            for r in self._host_rules:
                r.step(None, m)

    @property
    def meta_state(self):
        return self._meta_state

    # -- snapshot / restore (env_wrappers/simulation.py:64-91) ---------------------------------
    def snapshot(self):
        """Everything the reference's SimulationEnvironment deep-copies (state, action-space
        memory, task / rule counters, step_count, reset_next_step) lives in the two state
        records: a snapshot is a copy of them plus the host meta-state."""
        import copy
        return {'f64': self.state_f64.clone(), 'i32': self.state_i32.clone(),
                'meta_state': copy.deepcopy(self._meta_state)}

    def restore(self, snap):
        """Restores a snapshot.  The per-env random draw counters keep running (the
        reference does not rewind numpy's global RNG either), so repeated
        sim_step / sim_pop cycles sample different continuations."""
        o = self.layout.o_rng
        rng = self.state_i32[:, o:o + 4].clone()
        self.state_f64.copy_(snap['f64'])
        self.state_i32.copy_(snap['i32'])
        self.state_i32[:, o:o + 4] = rng
        self._meta_state = snap['meta_state']

    def physics_step(self, injected_uniforms=None):
        """`env.physics.step(env.state)` (tests/runtime_benchmark.py:101-107)."""
        inj, keep = self._inject(injected_uniforms)
        with self._torch.cuda.device(self.device):
            _engine.check(self._lib, self._lib.moog_engine_physics_only(
                self._handle, ctypes.byref(inj) if inj else None, self._stream()))
        del keep

    def observation(self):
        """Renders the current state (environment.py:128-131)."""
        self._poll_faults()
        if self._color_fn is not None:
            self._render_with_colors()
            return self._observation()
        with self._torch.cuda.device(self.device):
            _engine.check(self._lib, self._lib.moog_engine_render(
                self._handle, ctypes.c_void_p(self.image.data_ptr()), self._stream()))
        return self._observation()

    def observation_spec(self):
        from .observers import raw_state
        return {k: o.observation_spec() for k, o in self.observers.items()
                if not isinstance(o, raw_state.RawState)}

    def action_spec(self):
        return self.action_space.action_spec()

    def random_action(self):
        torch = self._torch
        if self._composite:
            out = {}
            for key, sub in self.action_space.action_spaces.items():
                kind = type(sub).__name__
                if kind == 'Grid':
                    out[key] = torch.randint(0, 5, (self.num_envs,), dtype=torch.int32, device=self.device)
                elif kind == 'SetPosition':
                    out[key] = torch.rand((self.num_envs, 2), dtype=torch.float64, device=self.device)
                else:
                    out[key] = torch.rand((self.num_envs, 2), dtype=torch.float64, device=self.device) * 2 - 1
            return out
        if self._is_grid:
            return torch.randint(0, 5, (self.num_envs,), dtype=torch.int32, device=self.device)
        return torch.rand((self.num_envs, 2), dtype=torch.float64, device=self.device) * 2 - 1

    @property
    def reset_next_step(self):
        return self.state_i32[:, self.layout.o_reset_next].bool()

    @property
    def step_count(self):
        return self.state_i32[:, self.layout.o_step_count]

    def static_prefix(self):
        """(number of leading sprite slots the rasteriser keeps as a cached picture, that
        picture uint8[H, W, 3] or None) -- include/moog_engine.h moog_engine_static_prefix."""
        n = ctypes.c_int32()
        _engine.check(self._lib, self._lib.moog_engine_static_prefix(self._handle, ctypes.byref(n), None, None))
        if n.value == 0:
            return 0, None
        img = self._torch.zeros_like(self.image[0])
        with self._torch.cuda.device(self.device):
            _engine.check(self._lib, self._lib.moog_engine_static_prefix(
                self._handle, ctypes.byref(n), ctypes.c_void_p(img.data_ptr()), self._stream()))
        return n.value, img

    def set_debug(self, step_debug=0, raster_stop=0):
        """Profiling aids of the kernels (include/moog_engine.h moog_engine_set_debug)."""
        _engine.check(self._lib, self._lib.moog_engine_set_debug(self._handle, int(step_debug), int(raster_stop)))

    # -- kernel timing -------------------------------------------------------------------
    def set_timing(self, enabled, kernels=None, every=1):
        """Brackets kernel launches with HIP events: all kernels, or only the MOOG_K_* ids in
        `kernels`; `every` = n brackets every n-th launch of a kernel (an event pair costs about
        5 microseconds of stream time, ~2 % of a step when every launch is bracketed)."""
        mask = 0
        if enabled:
            ids = range(_abi.MOOG_K_COUNT) if kernels is None else kernels
            for k in ids:
                mask |= 1 << int(k)
            mask |= (max(1, min(256, int(every))) - 1) << 8
        _engine.check(self._lib, self._lib.moog_engine_set_timing(self._handle, mask))

    def kernel_time(self, kernel_id):
        ms, cnt = ctypes.c_double(), ctypes.c_int64()
        _engine.check(self._lib, self._lib.moog_engine_kernel_time(
            self._handle, kernel_id, ctypes.byref(ms), ctypes.byref(cnt)))
        return ms.value, cnt.value

    # -- state views (sprite table as tensors) ----------------------------------------------
    def field(self, name):
        """Tensor view [N, S, ...] of one sprite field of the state record."""
        L, S = self.layout, self.layout.S
        f, q = self.state_f64, self.state_i32
        two = lambda o: f[:, o:o + 2 * S].view(-1, S, 2)
        one = lambda o: f[:, o:o + S]
        if name == 'position':
            return two(L.o_pos)
        if name == 'velocity':
            return two(L.o_vel)
        if name == 'angle':
            return one(L.o_angle)
        if name == 'angle_vel':
            return one(L.o_angvel)
        if name == 'mass':
            return one(L.o_mass)
        if name == 'color':
            return f[:, L.o_color:L.o_color + 3 * S].view(-1, S, 3)
        if name == 'inertia':
            return two(L.o_inertia)
        if name == 'max_radius':
            return one(L.o_maxr)
        if name == 'vertices':
            return f[:, L.o_verts:L.o_verts + 2 * L.TOTV].view(-1, L.TOTV, 2)
        if name == 'flags':
            return q[:, L.o_flags:L.o_flags + S]
        if name == 'alive':
            return (q[:, L.o_flags:L.o_flags + S] & _abi.MOOG_F_ALIVE).bool()
        if name == 'nverts':
            return q[:, L.o_nverts:L.o_nverts + S]
        if name == 'opacity':
            return q[:, L.o_opacity:L.o_opacity + S]
        raise KeyError(name)

    def sprites(self, env=0):
        """Host view of one env's state in the reference's shape (environment.py:143-146):
        OrderedDict layer name -> list of dicts with the Sprite.FACTOR_NAMES attributes
        (sprite.py:237-253), `vertices` and `slot`, live sprites in list order.  `scale` /
        `aspect_ratio` need `keep_sprite_factors=True` (else None)."""
        L, P, S = self.layout, self.compiled.program, self.layout.S
        f = self.state_f64[env].cpu().numpy()
        q = self.state_i32[env].cpu().numpy()
        out = collections.OrderedDict()
        for li, name in enumerate(self.compiled.layer_names):
            rows = []
            for s in range(P.layer_slot0[li], P.layer_slot0[li] + P.layer_nslots[li]):
                if not q[L.o_flags + s] & _abi.MOOG_F_ALIVE:
                    continue
                nv = int(q[L.o_nverts + s])
                o = L.o_verts + 2 * P.slot_voff[s]
                rows.append(dict(
                    x=float(f[L.o_pos + 2 * s]), y=float(f[L.o_pos + 2 * s + 1]),
                    shape=self.compiled.shape_names[int(q[L.o_shape + s])],
                    angle=float(f[L.o_angle + s]),
                    scale=float(f[L.o_scale + s]) if P.sprite_factors else None,
                    aspect_ratio=float(f[L.o_aspect + s]) if P.sprite_factors else None,
                    c0=float(f[L.o_color + 3 * s]), c1=float(f[L.o_color + 3 * s + 1]),
                    c2=float(f[L.o_color + 3 * s + 2]), opacity=int(q[L.o_opacity + s]),
                    x_vel=float(f[L.o_vel + 2 * s]), y_vel=float(f[L.o_vel + 2 * s + 1]),
                    angle_vel=float(f[L.o_angvel + s]), mass=float(f[L.o_mass + s]), metadata=None,
                    vertices=f[o:o + 2 * nv].reshape(nv, 2).copy(), slot=s))
            out[name] = rows
        return out

    def close(self):
        if self._handle:
            self._lib.moog_engine_destroy(self._handle)
            self._handle = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:  # pylint: disable=broad-except
            pass


class SubBatchedEnvironment(object):
    """N envs stepped as G independent sub-batches, each on its own HIP stream (an EnvPool-style surface):

        env = SubBatchedEnvironment(num_envs=4096, sub_batches=4, **config)
        env.reset()
        for g in range(env.sub_batches): env.step_async(g, actions_g)      # returns at once
        ts_g = env.recv(g)                                                 # TimeStep views of sub-batch g

    A `step()` of the whole batch lasts as long as its slowest env (contact-heavy envs take 2.5-3 x the mean,
    DESIGN 3.1), and the next call cannot start before it: the machine idles behind that barrier.  Sub-batches are
    chained only with themselves -- step -> frames -> next step on the sub-batch's stream -- so one sub-batch's slow
    tail runs beside another's head, and frames of one beside steps of another, by plain stream concurrency.

    Results do not depend on G: the parts are ordinary engine handles over contiguous slices of ONE set of tensors
    (`state_f64`, `image`, ... are whole-batch tensors here), and the random streams are keyed by the global env
    index, so env i computes exactly what it computes in a BatchedEnvironment of num_envs.

    Stream contract: `step_async(g, a)` makes sub-batch g's stream wait for the work already queued on the caller's
    current stream (the producer of `a`); `recv(g)` makes the caller's current stream wait for sub-batch g.  A consumer
    that wants the overlap must not funnel every sub-batch through one stream between recv and step_async: use
    `stream(g)` (e.g. `with torch.cuda.stream(env.stream(g))`) for the per-sub-batch policy work, or queue several
    steps ahead.  `step(actions)` / `reset()` are the synchronous whole-batch forms."""

    def __init__(self, state_initializer, physics, task, action_space, observers, game_rules=(),
                 meta_state_initializer=None, num_envs=1, sub_batches=2, device=None, seed=0, env_index0=0,
                 layer_capacity=None, keep_sprite_factors=False, reset_pool='auto'):
        import torch
        self._torch = torch
        if layer_capacity == 'auto' or (isinstance(layer_capacity, dict) and layer_capacity.get('auto')):
            # (growing a layer re-creates the engine over new records: the sub-batches are views of ONE set of tensors)
            raise NotImplementedError("SubBatchedEnvironment: layer_capacity='auto' is not available; pass {layer: slots} "
                                      "(BatchedEnvironment(layer_capacity='auto').layer_usage() reports the demand)")
        if meta_state_initializer is not None or any(getattr(r, 'host_side', False) for r in game_rules):
            raise NotImplementedError('SubBatchedEnvironment: host-side meta-state rules step the whole batch on the host')
        G = int(sub_batches)
        if G < 1 or num_envs % G != 0:
            raise ValueError('num_envs (%d) must be a multiple of sub_batches (%d)' % (num_envs, G))
        self.num_envs, self.sub_batches = int(num_envs), G
        self.device = torch.device('cuda', torch.cuda.current_device()) if device is None else torch.device(device)
        self.compiled = _compiler.compile_config(
            state_initializer, physics, task, action_space, observers, game_rules, None,
            layer_capacity=layer_capacity, keep_sprite_factors=keep_sprite_factors)
        P, L = self.compiled.program, self.compiled.layout
        self.layout = L
        with torch.cuda.device(self.device):
            bufs = BatchedEnvironment.allocate_buffers(torch, L, P, self.num_envs, self.device)
            self._streams = [torch.cuda.Stream(device=self.device) for _ in range(G)]
        (self.state_f64, self.state_i32, self.reward, self.discount, self.step_type, self.image) = bufs
        m = self.num_envs // G
        self.part_envs = m
        self.parts = []
        for g in range(G):
            views = tuple(b[g * m:(g + 1) * m] for b in bufs)
            self.parts.append(BatchedEnvironment(
                state_initializer, physics, task, action_space, observers, game_rules, None,
                num_envs=m, device=self.device, seed=seed, env_index0=int(env_index0) + g * m,
                reset_pool=reset_pool, _compiled=self.compiled, _buffers=views))
        self.physics, self.task, self.action_space = physics, task, action_space
        self.observers, self.game_rules = observers, game_rules
        self._is_grid = self.parts[0]._is_grid
        self._pending = [None] * G   # per sub-batch: the event of its last queued call

    def stream(self, g):
        return self._streams[g]

    def part_slice(self, g):
        return slice(g * self.part_envs, (g + 1) * self.part_envs)

    def enable_cost_schedule(self, enabled=True):
        """Cost-ordered launch inside every sub-batch (BatchedEnvironment.enable_cost_schedule)."""
        for g, p in enumerate(self.parts):
            with self._torch.cuda.stream(self._streams[g]):
                p.enable_cost_schedule(enabled)

    def _enqueue(self, g, fn):
        torch = self._torch
        s = self._streams[g]
        s.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(s):
            out = fn()
            ev = torch.cuda.Event()
            ev.record(s)
        self._pending[g] = ev
        return out

    def step_async(self, g, action):
        """Queues one step of sub-batch g (actions [num_envs / G, ...]) on its stream and returns."""
        if hasattr(action, 'record_stream') and action.is_cuda:
            action.record_stream(self._streams[g])
        self._enqueue(g, lambda: self.parts[g].step(action))

    def reset_async(self, g):
        self._enqueue(g, lambda: self.parts[g].reset())

    def recv(self, g):
        """TimeStep of sub-batch g's last queued call (views of the whole-batch tensors); the caller's current stream
        waits for it."""
        if self._pending[g] is not None:
            self._torch.cuda.current_stream(self.device).wait_event(self._pending[g])
        return self.parts[g]._timestep()

    def _join(self):
        cur = self._torch.cuda.current_stream(self.device)
        for ev in self._pending:
            if ev is not None:
                cur.wait_event(ev)

    def _timestep(self):
        return dm_env.TimeStep(self.step_type, self.reward, self.discount, {'image': self.image})

    def reset(self):
        for g in range(self.sub_batches):
            self.reset_async(g)
        self._join()
        return self._timestep()

    def step(self, action):
        """Whole-batch synchronous form: every sub-batch steps, the caller's stream waits for all of them."""
        m = self.part_envs
        for g in range(self.sub_batches):
            self.step_async(g, action[g * m:(g + 1) * m])
        self._join()
        return self._timestep()

    def random_action(self):
        torch = self._torch
        if self._is_grid:
            return torch.randint(0, 5, (self.num_envs,), dtype=torch.int32, device=self.device)
        return torch.rand((self.num_envs, 2), dtype=torch.float64, device=self.device) * 2 - 1

    def observation_spec(self):
        return self.parts[0].observation_spec()

    def action_spec(self):
        return self.parts[0].action_spec()

    def raise_faults(self):
        self._torch.cuda.synchronize(self.device)
        for p in self.parts:
            p.raise_faults()

    def set_timing(self, enabled, kernels=None, every=1):
        for p in self.parts:
            p.set_timing(enabled, kernels, every)

    def kernel_time(self, kernel_id):
        """(total ms, launches) summed over the sub-batches' engines."""
        ms, cnt = 0.0, 0
        for p in self.parts:
            a, b = p.kernel_time(kernel_id)
            ms += a
            cnt += b
        return ms, cnt

    def field(self, name):
        return BatchedEnvironment.field(self, name)

    def close(self):
        self._torch.cuda.synchronize(self.device)
        for p in self.parts:
            p.close()


class Environment(object):
    """Single-env facade with the reference's dm_env semantics (environment.py:28-158):
    numpy observation, Python-scalar reward, `None` reward/discount on FIRST."""

    def __init__(self, state_initializer, physics, task, action_space, observers, game_rules=(),
                 meta_state_initializer=None, **engine_kwargs):
        self._batched = BatchedEnvironment(
            state_initializer, physics, task, action_space, observers, game_rules,
            meta_state_initializer, num_envs=1, **engine_kwargs)
        self.physics = physics
        self.task = task
        self.action_space = action_space
        self.observers = observers
        self.game_rules = game_rules
        self.step_count = 0

    def _unbatch(self, ts):
        st = dm_env.StepType(int(ts.step_type[0].item()))
        obs = self._host_obs(ts.observation)
        self.step_count = int(self._batched.step_count[0].item())
        if st == dm_env.StepType.FIRST:
            return dm_env.TimeStep(st, None, None, obs)
        return dm_env.TimeStep(st, float(ts.reward[0].item()), float(ts.discount[0].item()), obs)

    def reset(self):
        return self._unbatch(self._batched.reset())

    def step(self, action):
        if isinstance(action, dict):   # Composite: {key: sub-action}
            keys = self._batched.action_space.action_keys
            packed = np.zeros((1, len(keys), 2))
            for i, key in enumerate(keys):
                v = np.asarray(action[key], dtype=np.float64).reshape(-1)
                packed[0, i, :len(v)] = v
            return self._unbatch(self._batched.step(packed))
        a = np.asarray(action)
        if self._batched._is_grid:
            a = a.reshape(1)
        else:
            a = a.reshape(1, 2).astype(np.float64)
        return self._unbatch(self._batched.step(a))

    @staticmethod
    def _host_obs(observation):
        """numpy frame; a RawState entry becomes the env's OrderedDict of sprite dicts
        (observers/raw_state.py:17-19 passes the state itself)."""
        return {k: (v.sprites(0) if hasattr(v, 'sprites') else v[0].cpu().numpy())
                for k, v in observation.items()}

    def observation(self):
        return self._host_obs(self._batched.observation())

    def observation_spec(self):
        return self._batched.observation_spec()

    def action_spec(self):
        return self._batched.action_spec()

    @property
    def reset_next_step(self):
        return bool(self._batched.reset_next_step[0].item())

    @property
    def meta_state(self):
        return self._batched.meta_state

    @property
    def batched(self):
        return self._batched

    def close(self):
        self._batched.close()
