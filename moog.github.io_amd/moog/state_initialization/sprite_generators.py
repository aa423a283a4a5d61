"""Sprite generators (reference: moog/state_initialization/sprite_generators.py:24-105).

`generate_sprites(...)` returns `_generate(disjoint=False, without_overlapping=[])`
exactly as in the reference; inside a traced state_initializer the call records
a generation op (rejection sampling runs on the device at every reset).
"""
from .. import _trace
from .. import sprite as sprite_lib


class _Generated(list):
    """What a fail_gracefully generator returned.  While tracing it always holds every placeholder; a config that
    asks `len(result) < n` / `not result` (red_green.py:152-155) is probed: in the run that explores op `short`, the list
    claims to be one sprite short -- if the initializer then starts over, the op is marked 'restart when short'."""

    def _short(self):
        t = _trace.active()
        if t is not None:
            t.len_observed.add(self.op_key)
            return t.short_op == self.op_key
        return False

    def __len__(self):
        return list.__len__(self) - (1 if self._short() else 0)

    def __bool__(self):
        return list.__len__(self) - (1 if self._short() else 0) > 0


def generate_sprites(factor_dist, num_sprites=1, max_recursion_depth=int(1e4),
                     fail_gracefully=False):
    def _generate(disjoint=False, without_overlapping=[]):
        t = _trace.active()
        if t is None:
            raise RuntimeError(
                'sprite generators only run inside an environment (the state_initializer is '
                'lowered to the device-side sampler; there is no host sampling path)')
        t.check_restart()   # the initializer called itself: `return state_initializer()` (red_green.py:155,203)
        n_calls = len(t.randint_calls)
        n = num_sprites() if callable(num_sprites) else num_sprites
        if len(t.randint_calls) > n_calls:
            lo, hi = t.randint_calls[-1]
            count_min, count_max = lo, hi - 1
        else:
            count_min = count_max = int(n)
        if count_max == 0:   # no sprite, no draw (sprite_generators.py:77-105 loops zero times): not an op at all
            return []
        # A component distribution of the config's own may call rng.uniform in sample() (red_green.py:31-48): those are
        # draws of every try of the sampler, interleaved with the factor samples; the first placeholder records them,
        # the others replay the same symbolic draws
        t.suspend = 'collect'
        t.collected, t.replay = [], None
        try:
            sprites = []
            for k in range(count_max):
                if k == 1:
                    t.replay = list(t.collected)
                t.replay_i = 0
                sprites.append(sprite_lib.Sprite(**factor_dist.sample()))
        finally:
            t.suspend = False
            t.replay = None
        op = _trace.GenOp(factor_dist, count_min, count_max, bool(disjoint),
                          list(without_overlapping), int(max_recursion_depth), sprites)
        op.fail_gracefully = bool(fail_gracefully)   # (:93-95) return the sprites made so far instead of raising
        if t.collected:
            proto = sprites[0]
            own = sorted((proto.factors[k].seq, ('factor', k)) for k in proto.sample_order)
            merged = sorted(own + [(seq, ('hdraw', idx)) for idx, seq in t.collected])
            op.draw_seq = [item for _, item in merged]
        t.add_op(op)
        if fail_gracefully:
            out = _Generated(sprites)
            out.op_key = len(t.ops) - 1
            op.gen_key = out.op_key
            return out
        return sprites

    return _generate


def chain_generators(*sprite_generators):
    """Concatenates the sprites of several generators (sprite_generators.py:110-128); each
    component stays its own generation op, run in order."""
    def _generate(*args, **kwargs):
        out = []
        for g in sprite_generators:
            out.extend(g(*args, **kwargs))
        return out
    return _generate


def sample_generator(sprite_generators, p=None):
    """sprite_generators.py:131-154: one of the generators, picked at random per call (`np.random.choice(generators,
    p=p)`), makes the sprites.  While tracing every alternative is run once; the device draws the index at each reset
    and runs only the ops of the picked alternative, all alternatives filling the same slots (so they must return the
    same number of sprites)."""
    sprite_generators = list(sprite_generators)

    def _generate(*args, **kwargs):
        t = _trace.active()
        if t is None:
            raise RuntimeError('sprite generators only run inside an environment')
        return t.choose(sprite_generators, p, args, kwargs)
    return _generate


def shuffle(sprite_generator):
    """sprite_generators.py:157-183: the generated sprites in a random order (`np.random.shuffle` of their indices).
    On the device the sprites are generated into their slots first and then swapped as numpy's shuffle swaps the list;
    the placeholders keep their positions, so the shuffled sprites must be the last of their layer (the next slot is
    the spare the swaps go through)."""
    def _generate(*args, **kwargs):
        t = _trace.active()
        if t is None:
            raise RuntimeError('sprite generators only run inside an environment')
        sprites = sprite_generator(*args, **kwargs)
        if len(sprites) > 1:
            t.add_op(_trace.ShuffleOp(list(sprites)))
        return sprites
    return _generate
