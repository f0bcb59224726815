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


def var_index(name):
    k = "xyz".index(name[0])
    _, m, i = name.split("_")
    return k * 30 + int(m) * 6 + int(i)


def parse_expr(txt):
    """'- 25 x_0_0 + 25 x_0_1' -> {idx: coef}"""
    out = {}
    toks = txt.replace("+", " + ").replace("-", " - ").split()
    sign, coef = 1.0, None
    for t in toks:
        if t == "+":
            sign, coef = 1.0, None
        elif t == "-":
            sign, coef = -1.0, None
        elif re.match(r"^[xyz]_\d+_\d+$", t):
            out[var_index(t)] = out.get(var_index(t), 0.0) + sign * (coef if coef is not None else 1.0)
            coef = None
        else:
            coef = float(t)
    return out


def main():
    txt = open(f"{REF}/log/QPmodel.lp", encoding="latin-1").read()
    obj_txt = txt[txt.index("obj1:") + 5: txt.index("Subject To")]
    lin_txt, quad_txt = obj_txt.split("[", 1)
    quad_txt = quad_txt[: quad_txt.index("]")]
    lin = parse_expr(lin_txt)
    # quadratic section is "[ ... ] / 2"
    assert "/ 2" in obj_txt[obj_txt.index("]"):]
    quad = []
    for sign, coef, a, b in re.findall(r"([+-]?)\s*([\d.e+-]+)\s+([xyz]_\d+_\d+)\s*(?:\^2|\*\s*([xyz]_\d+_\d+))", quad_txt):
        v = float(coef) * (-1.0 if sign == "-" else 1.0)
        ia = var_index(a)
        ib = var_index(b) if b else ia
        quad.append([ia, ib, v])
    cons_txt = txt[txt.index("Subject To") + 10: txt.index("Bounds")]
    rows = []
    for name, body in re.findall(r"(c\d+):\s*(.*?)(?=\n c\d+:|\Z)", cons_txt, flags=re.S):
        body = " ".join(body.split())
        m = re.match(r"(.*?)(>=|<=|=)\s*([-\d.e+]+)$", body)
        expr, sense, rhs = m.group(1), m.group(2), float(m.group(3))
        e = parse_expr(expr)
        rows.append({"name": name, "idx": list(e.keys()), "val": list(e.values()), "sense": sense, "rhs": rhs})
    b_txt = txt[txt.index("Bounds") + 6: txt.index("End")]
    bounds = {}
    for line in b_txt.strip().splitlines():
        line = line.strip()
        m = re.match(r"([-\d.e+]+)\s*<=\s*([xyz]_\d+_\d+)\s*<=\s*([-\d.e+]+)", line)
        if m:
            bounds[var_index(m.group(2))] = [float(m.group(1)), float(m.group(3))]
            continue
        m = re.match(r"([xyz]_\d+_\d+)\s+Free", line)
        if m:
            bounds[var_index(m.group(1))] = [None, None]
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
