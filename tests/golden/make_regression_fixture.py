"""Writes tests/golden/oracle_solve_regression.json from the oracle (run from the repo root)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mavmap_amd import synth  # noqa: E402
from tests import oracle_lib as O  # noqa: E402

scene = dict(num_images=6, num_points=200, track_len=4, models=[1, 2], seed=77)
options = dict(max_num_iterations=200, function_tolerance=1e-6, gradient_tolerance=1e-10)
O.set_threads(1)
p = synth.make_scene(**scene)
res, _ = O.solve(p, O.options(**options))
out = dict(scene=scene, options=options, num_successful_steps=res["num_successful_steps"],
           num_unsuccessful_steps=res["num_unsuccessful_steps"], final_cost=res["final_cost"],
           initial_cost=res["initial_cost"], poses=p.poses.tolist())
json.dump(out, open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "oracle_solve_regression.json"), "w"), indent=1)
print(res)
