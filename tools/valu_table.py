#!/usr/bin/env python3
"""VALU-side table of DESIGN.md from one SQ counter pass and the kernel-trace stats of the same command:
    valu_table.py profiles/<tag>_pmc_sq.csv profiles/<tag>_kernel_stats.csv
busy = SQ_ACTIVE_INST_VALU x 4 cycles / (duration x 1024 SIMDs x 2.4 GHz); stalled = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES."""
import csv, sys, collections
from rocprof_summary import short

as_json = "--json" in sys.argv
if as_json:
    sys.argv.remove("--json")

pm = collections.defaultdict(dict)
for r in csv.DictReader(open(sys.argv[1])):
    pm[short(r["kernel"])][r["counter"]] = pm[short(r["kernel"])].get(r["counter"], 0.0) + float(r["sum"])
    pm[short(r["kernel"])]["_n"] = max(pm[short(r["kernel"])].get("_n", 0), int(r["dispatches"]))
dur = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(sys.argv[2])):
    k = short(r["name"]); dur[k][0] += float(r["total_ns"]); dur[k][1] += int(r["calls"])
out = {}
if not as_json:
    print("| kernel | ms / launch | VALU instructions / launch | VALU-busy fraction of all SIMDs | waves stalled on issue |")
    print("|---|---|---|---|---|")
for k in sys.argv[3:] or sorted(pm):
    if k not in pm or k not in dur or not dur[k][1]:
        continue
    c, n = pm[k], pm[k]["_n"]
    ms = dur[k][0] / dur[k][1] / 1e6
    busy = c.get("SQ_ACTIVE_INST_VALU", 0) / n * 4 / (ms * 1e-3 * 1024 * 2.4e9)
    stall = c.get("SQ_WAIT_INST_ANY", 0) / max(c.get("SQ_WAVE_CYCLES", 1), 1)
    if as_json:
        out[k] = {"ms_per_launch": round(ms, 3), "valu_instructions_per_launch": int(c.get("SQ_INSTS_VALU", 0) / n), "valu_busy_frac": round(min(busy, 1.0), 3), "valu_busy_raw": round(busy, 3),
                  "waves_stalled_on_issue_frac": round(stall, 3)}
    else:
        print(f"| `k_{k}` | {ms:.3f} | {c.get('SQ_INSTS_VALU', 0) / n:.3g} | {busy:.2f} | {stall:.2f} |")
if as_json:
    import json
    out["_meta"] = {"source": [sys.argv[1], sys.argv[2]], "busy": "SQ_ACTIVE_INST_VALU x 4 cycles / (launch duration x 1024 SIMDs x 2.4 GHz): the share of all SIMD cycles with a VALU instruction executing (raw values slightly above 1 = the 2.4 GHz assumed for the engine clock is a little low; clamped)",
                    "ceiling": "39.3 T lane-ops/s for 4-cycle integer VALU instructions (256 CU x 4 SIMD x 64 lanes / 4 cycles x 2.4 GHz)"}
    print(json.dumps(out, indent=1, sort_keys=True))
