"""Throughput of the configs that run the step-kernel variants carrying every component (m3 / m4): python tools/dbg/m_configs.py"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..', 'moog.github.io_amd'))
sys.argv = sys.argv[:1]
import importlib.util
spec = importlib.util.spec_from_file_location('bc', os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'bench_configs.py'))
src = open(spec.origin).read().split("run('chase_avoid_torus', 4096)")[0]
exec(compile(src, spec.origin, 'exec'))
for name, n, st in (('pacman', 4096, 30), ('predators_arena_l2', 4096, 30), ('multi_tracking_with_feature_l3', 4096, 30), ('match_to_sample_l3', 4096, 30),
                    ('parallelogram_catch', 4096, 30), ('bounce_box_contact_prediction', 1024, 20), ('red_green_l1', 1024, 20)):
    run(name, n, steps=st)
