"""Parses the reference's DATA fixture log/QPmodel.lp (an exported CPLEX LP file, written on a QP
failure: src/traj_optimizer.cpp:100-102) into tests/golden/qpmodel_lp.json.

Run in the build container only (needs /root/reference):  python tests/golden/make_qpmodel_golden.py
The JSON holds numbers only: quadratic / linear objective, rows, bounds, plus the tick-1 scene that
produced it (missions/empty/10agents/multi_random_10agents_1.json start positions, z = 0.7).
"""
import json
import re
import sys

REF = "/root/reference"


sys.path.insert(0, __import__("os").path.join(__import__("os").path.dirname(__import__("os").path.abspath(__file__)), ".."))
from lp_parse import parse_lp  # noqa: E402


def main():
    txt = open(f"{REF}/log/QPmodel.lp", encoding="latin-1").read()
    P = parse_lp(txt)
    lin, quad, rows, bounds = P["lin"], P["quad"], P["rows"], P["bounds"]
    mission = json.load(open(f"{REF}/missions/empty/10agents/multi_random_10agents_1.json"))
    starts = [[a["start"][0], a["start"][1], 0.7] for a in mission["agents"]]
    quad_t = mission["quadrotors"]["crazyflie"]
    out = {
        "source": "log/QPmodel.lp (reference data fixture) + missions/empty/10agents/multi_random_10agents_1.json",
        "lin": {str(k): v for k, v in lin.items()},
        "quad": quad,
        "rows": rows,
        "bounds": {str(k): v for k, v in bounds.items()},
        "scene": {"agent": 3, "starts_xy_z07": starts, "world": mission["world"][0]["dimension"],
                  "max_vel": quad_t["max_vel"], "max_acc": quad_t["max_acc"], "radius": quad_t["radius"],
                  "downwash": quad_t["downwash"], "nominal_velocity": quad_t["nominal_velocity"]},
    }
    json.dump(out, open("tests/golden/qpmodel_lp.json", "w"))
    print(len(rows), "rows,", len(quad), "quadratic terms,", len(bounds), "bounds")


if __name__ == "__main__":
    sys.exit(main())
