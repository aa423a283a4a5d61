"""first_person_predators_prey / cleanup / rules_zoo_l1 (programs with layers that rules append to): python tools/dbg/fppp_bench.py"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..', 'moog.github.io_amd'))
sys.argv = sys.argv[:1]
src = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'bench_configs.py')).read().split("run('chase_avoid_torus', 4096)")[0]
exec(compile(src, 'bench_configs.py', 'exec'))
for name, n, st in (('first_person_predators_prey', 4096, 60), ('cleanup', 4096, 60), ('rules_zoo_l1', 4096, 40), ('pacman', 4096, 30)):
    run(name, n, steps=st)
