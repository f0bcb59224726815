#!/usr/bin/env python3
"""Merges the four PMC summaries of tools/profile_round.sh (pmc_bench.json, pmc_forest.json, pmc_large.json, pmc_general.json) into
profiles/rNN_pmc_summary.json: {"kernels": {kernel@grid[#workload]: counters}, "notes": [...], "source": ...}.

    python profiles/merge_pmc.py gpurun_out/r05prof profiles/r05_pmc_summary.json "note" ..."""
import json
import os
import sys

src, out, notes = sys.argv[1], sys.argv[2], sys.argv[3:]
ker = {}
for f, tag in (("pmc_bench.json", ""), ("pmc_forest.json", "#forest256"), ("pmc_large.json", "#random1024"), ("pmc_general.json", "#general")):
    d = json.load(open(os.path.join(src, f)))
    for k, v in d.items():
        ker[k + tag] = v
json.dump({"kernels": ker, "notes": ["keys: kernel@grid size (threads) [#workload]; no tag = the bench command (tools/profile_round.sh: BENCH), #general = tools/general_profile.py, "
                                     "#forest256 = tools/config_runs.py forest256p, forest256, forest256x4p, forest256x4, #random1024 = 1024-agent random swarm",
                                     "template arguments are not part of a key; FETCH_SIZE / WRITE_SIZE in KB per launch, SQ_* summed over the chip per launch, each counter group collected in a pass of its own"] + notes,
           "source": "tools/profile_round.sh -> profiles/summarize_rocpd.py pmc -> profiles/merge_pmc.py"}, open(out, "w"), indent=1, sort_keys=True)
