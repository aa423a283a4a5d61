"""pacman / maze configs with and without the rasteriser's per-env prefix: python tools/dbg/pacman_bench.py"""
import os, sys, subprocess
HERE = os.path.dirname(os.path.abspath(__file__))
if len(sys.argv) > 1 and sys.argv[1] == 'child':
    sys.path.insert(0, os.path.join(HERE, '..', '..', 'moog.github.io_amd'))
    sys.argv = sys.argv[:1]
    src = open(os.path.join(HERE, '..', 'bench_configs.py')).read().split("run('chase_avoid_torus', 4096)")[0]
    exec(compile(src, 'bench_configs.py', 'exec'))
    for name, n, st in (('pacman', 4096, 60), ('pacman', 1024, 60), ('maze_zoo', 4096, 60), ('functional_maze', 8192, 30), ('cleanup', 4096, 60)):
        run(name, n, steps=st, **({'image_size': (128, 128)} if name == 'functional_maze' else {}))
else:
    for v in ('0', '1'):
        print('== MOOG_RASTER_ENV_BG=%s' % v, flush=True)
        subprocess.call([sys.executable, os.path.abspath(__file__), 'child'], env=dict(os.environ, MOOG_RASTER_ENV_BG=v))
