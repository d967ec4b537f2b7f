import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)) + "/../..")
from tests.test_gpu_random import random_problem
which, seed = sys.argv[1], int(sys.argv[2])
p, opts, radius = random_problem(seed)
so = dict(opts, max_num_iterations=40, max_trust_region_radius=1e4, print_progress=1)
if which == "oracle":
    from tests import oracle_lib as O
    r, _ = O.solve(p, O.options(**so))
else:
    import mavmap_amd.api as M
    _, r = M.bundle_adjustment(p, so)
sys.stderr.write("%s %s %.12g %d %d\n" % (which, r["termination_name"], r["final_cost"], r["num_successful_steps"], r["num_unsuccessful_steps"]))
