"""Generates tests/golden/multi_square16.json IN THE BUILD CONTAINER (needs /root/reference): the inputs of the one mission for
which the reference publishes an outcome (log/summary_LSC_16agents.csv: 16 agents, missions/multi_square16.json in
world/simple_forest.bt) -- quadrotor parameters, world box, starts and goals, in the reference's mission-file schema
(missions/readme.txt) so that the product's mission reader is exercised on it -- plus that published outcome.  Data only.

    python tests/golden/make_mission_golden.py
"""
import csv
import json
import os

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"

src = json.load(open(os.path.join(REF, "missions", "multi_square16.json")))
rows = list(csv.DictReader(open(os.path.join(REF, "log", "summary_LSC_16agents.csv"))))
out = {
    "quadrotors": {k: {f: v[f] for f in ("max_vel", "max_acc", "radius", "nominal_velocity", "downwash")} for k, v in src["quadrotors"].items()},
    "world": [{"dimension": src["world"][0]["dimension"]}],
    "agents": [{"type": a["type"], "cid": a["cid"], "start": a["start"], "goal": a["goal"]} for a in src["agents"]],
    "obstacles": [],
    "published_outcome": [{k: float(r[k]) for k in ("total_flight_time", "total_flight_distance", "is_collided", "safety_ratio_agent")} for r in rows],
    "published_outcome_source": "log/summary_LSC_16agents.csv (two runs on the authors' machines, multisim/max_noise 0.02, unseeded)",
}
json.dump(out, open(os.path.join(HERE, "multi_square16.json"), "w"), indent=1)
print(len(out["agents"]), "agents;", out["published_outcome"])
