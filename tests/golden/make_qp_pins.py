"""Generates tests/golden/qp_pins.json IN THE BUILD CONTAINER (needs /root/reference): the verdict of an independent
solver (HiGHS, SciPy's bundled copy) on the reference's own QP fixture log/QPmodel.lp, read verbatim by HiGHS's
LP-file reader.  Data only: solver name/version, model dimensions, model status.

    python tests/golden/make_qp_pins.py
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import highs_qp as H  # noqa: E402

REF_LP = "/root/reference/log/QPmodel.lp"

status, rows, cols, hnz = H.read_lp_file(REF_LP)
out = {"solver": "HiGHS " + H.version() + " (scipy.optimize._highspy)",
       "reference_lp": {"file": "log/QPmodel.lp", "rows": rows, "cols": cols, "hessian_nnz": hnz, "highs_status": status,
                        "note": "written by the reference when CPLEX fails (src/traj_optimizer.cpp:100-102)"}}
json.dump(out, open(os.path.join(HERE, "qp_pins.json"), "w"), indent=1)
print(out)
