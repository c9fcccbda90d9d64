#!/usr/bin/env python
"""The reference's only stored end-to-end numbers for the signature-kernel path
(examples/script_sequential_distribution.ipynb, cells 9 and 12: N=100, T=10, d=2, SignatureKernel(h=5, depth=4),
Adam lr=0.05 x 200 iterations) against this build, for several seeds and BOTH sign conventions of grad_k.

    python scripts/notebook_statistics.py [--seeds 5] > profiles/r02_notebook_statistics.json

The notebook run is unseeded on an unknown device, so this is a statistical comparison, not a parity fixture
(SURVEY.md §4, §8c)."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples"))

NOTEBOOK = {"variance_per_timestep": [0.0516, 0.5657, 0.6224, 0.5866, 1.2819, 0.7015, 0.6722, 0.5763, 0.4380, 0.1005],
            "mean_log_prob": -21.1506, "max_log_prob": -19.6702, "avg_path_length_cell12": 3.2981}

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, default=5)
    ap.add_argument("--steps", type=int, default=200)
    args = ap.parse_args()
    from sequential_distribution import run

    table = {"notebook": NOTEBOOK, "runs": []}
    for sign in (-1.0, 1.0):
        for seed in range(args.seeds):
            r = run(args.steps, seed, grad_k_sign=sign)
            r.update({"grad_k_sign": sign, "seed": seed})
            table["runs"].append(r)
    for sign in (-1.0, 1.0):
        rs = [r for r in table["runs"] if r["grad_k_sign"] == sign]
        n = len(rs)
        table[f"summary_sign_{int(sign):+d}"] = {
            "mean_log_prob": sum(r["mean_log_prob"] for r in rs) / n,
            "max_log_prob": sum(r["max_log_prob"] for r in rs) / n,
            "avg_path_length_cell12": sum(r["avg_path_length_cell12"] for r in rs) / n,
            "variance_per_timestep": [sum(r["variance_per_timestep"][t] for r in rs) / n for t in range(10)],
        }
    print(json.dumps(table, indent=1))
