"""Episode logging wrapper (reference: moog/env_wrappers/logger.py:33-224) writing the same
on-disk format, so that logs of the batched engine can be read by the reference's tools
(moog_demos/restore_logged_data.py): a time-stamped directory with `attributes.txt`,
`description.txt` and one JSON file per episode (zero-padded episode number), each a list
of steps `[[time, t], [reward, r], [step_type, k], [action, a], [meta_state, m], state]`
where `state` lists every layer as `[name, [serialized sprites]]` and a serialized sprite
is the list of the attributes in attributes.txt, followed by its vertices when they are
logged (log_vertices = NEVER / WHEN_NECESSARY / ALWAYS; WHEN_NECESSARY = the first step a
sprite appears).

Works over the single-env `Environment` facade (one logged env, as in the reference).  The
environment must be built with `keep_sprite_factors=True` so that `scale` / `aspect_ratio`
are part of the state records.  `id` is assigned by this wrapper: unique per sprite within
the run and stable while the sprite lives (the reference's global construction counter,
sprite.py:32,325-327, also counts rejected samples and is not reproduced).
"""
from datetime import datetime
import json
import os
import time

import numpy as np

from .. import sprite as sprite_lib

_FILENAME_ZFILL = 5   # logger.py:30


class VertexLogging():
    NEVER = 'NEVER'
    ALWAYS = 'ALWAYS'
    WHEN_NECESSARY = 'WHEN_NECESSARY'


def _serialize(x):
    """logger.py:39-66"""
    if isinstance(x, np.ndarray):
        return x.tolist()
    if isinstance(x, (np.float32, np.float64)):
        return float(x)
    if isinstance(x, (np.int32, np.int64)):
        return int(x)
    if isinstance(x, list):
        return [_serialize(a) for a in x]
    if isinstance(x, tuple):
        return tuple([_serialize(a) for a in x])
    if isinstance(x, dict):
        return {k: _serialize(v) for k, v in x.items()}
    return x


_DESCRIPTION = (
    'Each numerical file in this directory is an episode of the task. Each such file contains '
    'a json-serialized list, each element of which represents an environment step in the '
    'episode. Each step is a list of four elements, [[`time`, time], [`reward`, reward], '
    '[`step_type`, step_type], [`action`, action], [`meta_state`, meta_state`], state].'
    '\n\n\n\ntime is a timestamp of the timestep.'
    '\n\n\n\nreward contains the value of the reward at that step.'
    '\n\n\n\nstep_type indicates the dm_env.StepType of that step, i.e. whether it was first, '
    'mid, or last.'
    '\n\n\n\naction contains the agent action for the step.'
    '\n\n\n\nmeta_state is the serialized meta_state of the environment.'
    '\n\n\n\nstate is a list, each element of which represents a layer in the environment '
    'state. The layer is represented as a list [k, [], [], [], ...], where k is the layer name '
    'and the subsequent elements are serialized sprites. Each serialized sprite is a list of '
    'attributes. See attributes.txt for the attributes contained.')


class LoggingEnvironment(object):
    def __init__(self, environment, log_dir='logs', log_vertices='WHEN_NECESSARY'):
        if not hasattr(VertexLogging, log_vertices):
            raise ValueError('log_vertices is {} but must be in VertexLogging values'.format(log_vertices))
        self._environment = environment
        self._engine = getattr(environment, 'batched', environment)
        if not self._engine.compiled.program.sprite_factors:
            raise ValueError('LoggingEnvironment needs Environment(..., keep_sprite_factors=True)')
        self._log_vertices = log_vertices
        now_str = datetime.now().strftime('%Y_%m_%d_%H_%M_%S')
        log_dir = os.path.join(log_dir if log_dir[0] == '/' else os.path.join(os.getcwd(), log_dir), now_str)
        os.makedirs(log_dir)
        self._log_dir = log_dir
        self._attributes = list(sprite_lib.FACTOR_NAMES) + ['id']
        with open(os.path.join(log_dir, 'attributes.txt'), 'w') as f:
            json.dump(self._attributes, f)
        description = _DESCRIPTION
        if log_vertices == VertexLogging.ALWAYS:
            description += (' Furthermore, a list of vertices is appended to the attribute list for '
                            'each serialized sprite.')
        elif log_vertices == VertexLogging.WHEN_NECESSARY:
            description += ('\n\n\n\nFurthermore, a list of vertices is appended to the attribute list '
                            'for a serialized for the first timestep in which that serialized sprite '
                            'appears, or when the sprite has changed shape.')
        with open(os.path.join(log_dir, 'description.txt'), 'w') as f:
            f.write(description)
        self._episode_count = 0
        self._episode_log = []
        self._ids = {}        # slot -> id of the sprite currently living there
        self._next_id = 0

    @property
    def log_dir(self):
        return self._log_dir

    def __getattr__(self, attr):
        return getattr(self._environment, attr)

    def _serialized_state(self, fresh_episode):
        """logger.py:175-196.  Sprite identity: a sprite keeps its slot while it lives, except
        in layers that rules append to, where it is re-identified by its order."""
        state = self._engine.sprites(0)
        if fresh_episode:
            self._ids = {}
        seen = {}
        out = []
        for name, rows in state.items():
            ser = []
            for r in rows:
                key = r['slot']
                new = key not in self._ids
                if new:
                    self._ids[key] = self._next_id
                    self._next_id += 1
                seen[key] = True
                attrs = [r[a] if a != 'id' else self._ids[key] for a in self._attributes]
                if self._log_vertices == VertexLogging.ALWAYS or (
                        self._log_vertices == VertexLogging.WHEN_NECESSARY and new):
                    attrs.append(r['vertices'].tolist())
                ser.append(attrs)
            out.append([name, ser])
        for key in list(self._ids):
            if key not in seen:
                del self._ids[key]
        return out

    def reset(self):
        return self._environment.reset()

    def step(self, action):
        """logger.py:198-224"""
        timestep = self._environment.step(action)
        self._episode_log.append([
            ['time', time.time()],
            ['reward', timestep.reward],
            ['step_type', timestep.step_type.value],
            ['action', _serialize(action)],
            ['meta_state', _serialize(self._environment.meta_state)],
            self._serialized_state(timestep.first())])
        if timestep.last():
            filename = os.path.join(self._log_dir, str(self._episode_count).zfill(_FILENAME_ZFILL))
            with open(filename, 'w') as f:
                json.dump(self._episode_log, f)
            self._episode_count += 1
            self._episode_log = []
        return timestep
