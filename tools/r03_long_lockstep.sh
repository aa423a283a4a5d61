cd $GRAFT_REPO_ROOT
run() { timeout 600 python tools/dbg/long_lockstep.py "$@" 2>&1 | grep -v amdgpu | tail -1; }
run colliding_predators_32 4096 230
run chase_avoid_torus 4096 220
run functional_maze 2048 150
run falling_balls_64 2048 130
run pacman 256 120
run first_person_predators_prey 512 100
run parallelogram_catch_l2 1024 120
run multi_tracking_with_feature_l3 1024 200
run match_to_sample_l3 1024 160
run predators_arena_l2 2048 150
run bounce_box_contact_prediction 256 80
run red_green_l1 256 60
run lookahead_zoo_l1 1024 100
run tracing_zoo 1024 100
run combo_zoo 1024 100
run maze_zoo_l2 1024 150
run rules_zoo_l1 2048 100
run lambda_zoo 2048 100
