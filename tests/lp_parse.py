"""Reader of CPLEX LP files as the reference writes them (log/QPmodel.lp) and as lsc_dump_qp writes them: variables x_m_i / y_m_i /
z_m_i -> index k*30 + m*6 + i, objective "linear + [ quadratic ] / 2 + constant", rows c1.., bounds.  Numbers only."""
import re


def var_index(name):
    k = "xyz".index(name[0])
    _, m, i = name.split("_")
    return k * 30 + int(m) * 6 + int(i)


def parse_expr(txt):
    """'- 25 x_0_0 + 25 x_0_1' -> {idx: coef}"""
    out = {}
    toks = txt.replace("+", " + ").replace("-", " - ").split()
    sign, coef = 1.0, None
    pend = ""
    for t in toks:
        if t in ("+", "-") and pend.endswith("e"):      # exponent sign of a number split by the tokeniser: 1e - 05
            pend += t
            continue
        if pend:
            if re.match(r"^\d+$", t) and pend[-1] in "+-":
                coef = float(pend + t)
                pend = ""
                continue
            pend = ""
        if t == "+":
            sign, coef = 1.0, None
        elif t == "-":
            sign, coef = -1.0, None
        elif re.match(r"^[xyz]_\d+_\d+$", t):
            out[var_index(t)] = out.get(var_index(t), 0.0) + sign * (coef if coef is not None else 1.0)
            coef = None
        elif re.match(r"^[\d.]+e$", t):
            pend = t
        else:
            coef = float(t)
    return out


def parse_lp(txt):
    """-> dict(lin {idx: v}, quad [[i, j, v]], const, rows [dict(name, idx, val, sense, rhs)], bounds {idx: [lo, hi] | [None, None]})"""
    obj_txt = txt[txt.index("obj1:") + 5: txt.index("Subject To")]
    lin_txt, quad_txt = obj_txt.split("[", 1)
    tail = quad_txt[quad_txt.index("]"):]
    quad_txt = quad_txt[: quad_txt.index("]")]
    assert "/ 2" in tail
    m = re.search(r"/ 2\s*\+\s*([-\d.e+]+)", tail)
    const = float(m.group(1)) if m else 0.0
    lin = parse_expr(lin_txt)
    quad = []
    for sign, coef, a, b in re.findall(r"([+-]?)\s*([\d.]+(?:e[+-]?\d+)?)\s+([xyz]_\d+_\d+)\s*(?:\^2|\*\s*([xyz]_\d+_\d+))", quad_txt):
        v = float(coef) * (-1.0 if sign == "-" else 1.0)
        ia = var_index(a)
        ib = var_index(b) if b else ia
        quad.append([ia, ib, v])
    cons_txt = txt[txt.index("Subject To") + 10: txt.index("Bounds")]
    rows = []
    for name, body in re.findall(r"(c\d+):\s*(.*?)(?=\n c\d+:|\Z)", cons_txt, flags=re.S):
        body = " ".join(body.split())
        mm = re.match(r"(.*?)(>=|<=|=)\s*([-\d.e+]+)$", body)
        expr, sense, rhs = mm.group(1), mm.group(2), float(mm.group(3))
        e = parse_expr(expr)
        rows.append({"name": name, "idx": list(e.keys()), "val": list(e.values()), "sense": sense, "rhs": rhs})
    b_txt = txt[txt.index("Bounds") + 6: txt.index("End")]
    bounds = {}
    for line in b_txt.strip().splitlines():
        line = line.strip()
        mm = re.match(r"([-\d.e+]+)\s*<=\s*([xyz]_\d+_\d+)\s*<=\s*([-\d.e+]+)", line)
        if mm:
            bounds[var_index(mm.group(2))] = [float(mm.group(1)), float(mm.group(3))]
            continue
        mm = re.match(r"([xyz]_\d+_\d+)\s+Free", line)
        if mm:
            bounds[var_index(mm.group(1))] = [None, None]
    return {"lin": lin, "quad": quad, "const": const, "rows": rows, "bounds": bounds}
