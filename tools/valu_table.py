#!/usr/bin/env python3
"""The VALU-issue bound of the instruction-bound kernels, from measurements only:

    valu_table.py --calib profiles/valu_calibration.json --pmc profiles/<tag>_pmc_sq.csv [--isa-dir DIR | --build-isa] [--json]

  * what an instruction costs   profiles/valu_calibration.json (tools/valu_calib.hip on the MI355X): SIMD cycles per wave64 instruction, per opcode,
                                with the SIMD saturated (8 independent chains, 8 waves per SIMD). Two classes and a few outliers: ~2.2 cycles for
                                add / sub / and / or / xor / right shifts / mov / f32 add, mul, fma -- ~4.1 cycles for everything else (multiplies of
                                any width, min / max, left shifts, three-operand integer forms, compares, selects, converts, DPP, packed 16-bit,
                                double precision add / mul / fma / min, the 32 x 32 + 64-bit multiply-add) -- ~8.1 for v_mad_u16.
  * how many were executed      SQ_INSTS_VALU per launch of the kernel (one `rocprofv3 --pmc` pass over bench.py; SQ_ACTIVE_INST_VALU is the same
                                number on gfx950: it counts instructions, not cycles -- the calibration pass shows it, profiles/r03h_pmc_calib.csv)
  * of which opcodes            the kernel's ISA (hipcc -S of the same source, same flags): STATIC opcode counts of the kernel's function. The dynamic
                                mix is not observable with counters; the static one stands in for it (the hot loops are most of these kernels' code).
  * over how many cycles        SQ_BUSY_CYCLES / 32 = the launch's duration in shader cycles (32 = the counter's instances: 8 XCDs x 4 shader engines;
                                checked against s_memtime spans in the calibration pass). No clock frequency is assumed anywhere.

  busy = SQ_INSTS_VALU x (static mean cycles per instruction) / (1024 SIMDs x SQ_BUSY_CYCLES / 32)

= the share of all SIMD issue cycles the kernel's VALU instructions occupy. 1.0 is the ceiling; what is left below it is latency the resident waves
did not cover (SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES is printed beside it)."""
import argparse
import collections
import csv
import json
import pathlib
import re
import subprocess
import sys
import tempfile

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent))
from rocprof_summary import short  # noqa: E402

ROOT = pathlib.Path(__file__).resolve().parent.parent
CSRC = ROOT / "basis_universal_amd" / "csrc"
SOURCES = ["etc1s_kernels.hip", "uastc_kernels.hip", "uastc_rdo_kernels.hip"]
DEFAULT_CYCLES = 4.1   # opcodes the calibration did not time are priced like the 4-cycle class (every multi-operand / non-trivial opcode measured sits there)


def base_opcode(tok):
    return re.sub(r"_(e32|e64|sdwa|dpp|e64_dpp)$", "", tok)


def load_calibration(path):
    d = json.loads(pathlib.Path(path).read_text())
    cost = {}
    for c in d["cells"]:
        if c["chains"] == 8 and c["waves_per_simd"] == 8 and "+" not in c["op"]:
            cost[c["op"]] = c["cycles_per_inst"]
    # opcodes that are the same hardware operation as a measured one
    alias = {"v_subrev_u32": "v_sub_u32", "v_sub_f32": "v_add_f32", "v_subrev_f32": "v_add_f32", "v_mac_f32": "v_fmac_f32", "v_max_u32": "v_min_u32", "v_min_i32": "v_min_i32",
             "v_max_f32": "v_max_f32", "v_not_b32": "v_xor_b32", "v_mov_b32_dpp": "v_mov_b32_dpp", "v_fma_f64": "v_fma_f64", "v_mul_f64": "v_mul_f64", "v_fmac_f64": "v_fma_f64", "v_max_f64": "v_min_f64", "v_mad_i64_i32": "v_mad_i64_i32", "v_cvt_f64_u32": "v_cvt_f64_i32", "v_mov_b64": "v_lshl_add_u64", "v_add_f64": "v_add_f64",
             "v_max3_u32": "v_min3_u32", "v_max3_i32": "v_min3_u32", "v_min3_i32": "v_min3_u32", "v_med3_u32": "v_med3_i32", "v_bfe_i32": "v_bfe_u32", "v_sub_co_u32": "v_add_co_u32",
             "v_addc_co_u32": "v_add_co_u32", "v_subb_co_u32": "v_add_co_u32", "v_subrev_co_u32": "v_add_co_u32", "v_lshlrev_b64": "v_lshlrev_b32", "v_lshrrev_b64": "v_lshlrev_b32",
             "v_mbcnt_hi_u32_b32": "v_mbcnt_lo_u32_b32", "v_cvt_f32_i32": "v_cvt_f32_u32", "v_cvt_i32_f32": "v_cvt_u32_f32"}
    for a, b in alias.items():
        if a not in cost and b in cost:
            cost[a] = cost[b]
    return cost


def opcode_cycles(op, cost):
    op = base_opcode(op)
    if op in cost:
        return cost[op], True
    if op.startswith("v_cmp") and "v_cmp_lt_u32" in cost:
        return cost["v_cmp_lt_u32"], True
    return DEFAULT_CYCLES, False


def isa_mix(isa_dir, cost):
    """kernel label (rocprof_summary.short of the demangled name is not available for .s symbols, so: mangled symbol) -> opcode histogram"""
    out, calls = {}, {}
    for s in sorted(pathlib.Path(isa_dir).glob("*.s")):
        cur, hist = None, None
        for line in s.read_text().splitlines():
            m = re.match(r"^(_Z\w+):\s*(;.*)?$", line)
            if m:
                cur, hist = m.group(1), collections.Counter()
                out[cur] = hist
                calls[cur] = set()
                continue
            if cur is None:
                continue
            if line.startswith(".Lfunc_end") or line.strip().startswith(".section"):
                cur = None
                continue
            t = line.strip().split()
            if t and t[0].startswith("v_"):
                hist[base_opcode(t[0])] += 1
            c = re.search(r"(_Z\w+)@rel32@lo", line)
            if c:
                calls[cur].add(c.group(1))
    # a kernel's mix includes the functions it calls (the UASTC kernels keep their big stages out of line: the kernel's own body is a few hundred instructions,
    # the work is in the callees)
    def closure(sym, seen):
        for c in calls.get(sym, ()):
            if c in out and c not in seen:
                seen.add(c)
                closure(c, seen)
        return seen
    merged = {}
    for sym, hist in out.items():
        h = collections.Counter(hist)
        for c in closure(sym, {sym}) - {sym}:
            h.update(out[c])
        merged[sym] = h
    return merged


def demangle(names):
    r = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True)
    return dict(zip(names, r.stdout.splitlines()))


def build_isa(dest):
    flags = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "--cuda-device-only", "-S"]
    for src in SOURCES:
        subprocess.check_call(["/opt/rocm/bin/hipcc", *flags, "-o", str(pathlib.Path(dest) / (src + ".s")), str(CSRC / src)], stderr=subprocess.DEVNULL)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--calib", required=True)
    ap.add_argument("--pmc", required=True)
    ap.add_argument("--isa-dir")
    ap.add_argument("--build-isa", action="store_true")
    ap.add_argument("--json", action="store_true")
    ap.add_argument("--commit", default="")
    ap.add_argument("kernels", nargs="*")
    a = ap.parse_args()
    cost = load_calibration(a.calib)
    tmp = None
    if a.build_isa:
        tmp = tempfile.TemporaryDirectory()
        build_isa(tmp.name)
        a.isa_dir = tmp.name
    mixes = isa_mix(a.isa_dir, cost) if a.isa_dir else {}
    by_label = {}
    dm = demangle(list(mixes))
    for sym, hist in mixes.items():
        if "k_" not in dm[sym] or not sum(hist.values()):
            continue
        lab = short(dm[sym])
        # several instantiations share a label (perceptual / linear, ...): keep the one with the most instructions per label + template args in the name
        key = (lab, dm[sym].split("(")[0])
        by_label.setdefault(lab, []).append((key[1], hist))

    pm = collections.defaultdict(lambda: collections.defaultdict(float))
    names = {}
    for r in csv.DictReader(open(a.pmc)):
        k = short(r["kernel"])
        pm[k][r["counter"]] += float(r["sum"])
        pm[k]["_n"] = max(pm[k]["_n"], int(r["dispatches"]))
        names.setdefault(k, r["kernel"].split("(")[0].replace("void ", ""))
    out = {}
    rows = []
    for k in a.kernels or sorted(pm):
        c = pm[k]
        n = c["_n"]
        if not n or not c.get("SQ_INSTS_VALU") or not c.get("SQ_BUSY_CYCLES"):
            continue
        # the instantiation that ran: match the profiled name (template arguments included) against the ISA functions of that label
        cands = by_label.get(k, [])
        hist = None
        for nm, h in cands:
            if nm.replace(" ", "") in names[k].replace(" ", "") or names[k].replace(" ", "") in nm.replace(" ", ""):
                hist = h
                break
        if hist is None and cands:
            hist = max(cands, key=lambda t: sum(t[1].values()))[1]
        if hist is None:
            continue
        total = sum(hist.values())
        cyc, timed = 0.0, 0
        cls = collections.Counter()
        unknown = collections.Counter()
        for op, cnt in hist.items():
            cy, known = opcode_cycles(op, cost)
            cyc += cy * cnt
            timed += cnt if known else 0
            cls["2-cycle" if cy < 3 else "4-cycle" if cy < 6 else "8-cycle"] += cnt
            if not known:
                unknown[op] += cnt
        mean = cyc / total
        insts = c["SQ_INSTS_VALU"] / n
        dur_cycles = c["SQ_BUSY_CYCLES"] / n / 32.0
        busy = insts * mean / (1024.0 * dur_cycles)
        stall = c.get("SQ_WAIT_INST_ANY", 0) / max(c.get("SQ_WAVE_CYCLES", 1), 1)
        rec = {"valu_instructions_per_launch": int(insts), "launch_shader_cycles": int(dur_cycles), "static_valu_instructions": total,
               "static_mix": {kk: round(v / total, 3) for kk, v in sorted(cls.items())}, "mean_cycles_per_instruction": round(mean, 3),
               "valu_busy_frac": round(busy, 3), "waves_stalled_on_issue_frac": round(stall, 3),
               "opcodes_priced_by_default": {"share": round(1 - timed / total, 3), "top": dict(unknown.most_common(6))},
               "top_opcodes": dict(hist.most_common(8))}
        out[k] = rec
        rows.append((k, rec))
    if a.json:
        out["_meta"] = {"source": {"calibration": a.calib, "pmc": a.pmc, "commit": a.commit},
                        "busy": "SQ_INSTS_VALU x static mean cycles per instruction / (1024 SIMDs x SQ_BUSY_CYCLES / 32); cycles per opcode measured by tools/valu_calib.hip "
                                "(profiles/valu_calibration.json), opcode mix = static counts of the kernel's ISA; no clock frequency assumed",
                        "ceiling": "1.0 = every SIMD issue cycle taken by a VALU instruction. Per opcode class (MI355X, measured): ~2.2 cycles per wave64 instruction for add/sub/logic/right "
                                   "shift/mov/f32 add-mul-fma, ~4.1 for every other integer, compare, select, convert, packed and double-precision opcode, ~8.1 for v_mad_u16"}
        print(json.dumps(out, indent=1, sort_keys=True))
    else:
        print("| kernel | VALU instructions / launch | static mix 2 / 4 / 8-cycle | mean cycles / instruction | launch, shader cycles | VALU-busy | waves stalled on issue |")
        print("|---|---|---|---|---|---|---|")
        for k, r in rows:
            m = r["static_mix"]
            print(f"| `k_{k}` | {r['valu_instructions_per_launch']:.3g} | {m.get('2-cycle', 0):.2f} / {m.get('4-cycle', 0):.2f} / {m.get('8-cycle', 0):.2f} | {r['mean_cycles_per_instruction']:.2f} | "
                  f"{r['launch_shader_cycles']:.3g} | **{r['valu_busy_frac']:.2f}** | {r['waves_stalled_on_issue_frac']:.2f} |")


if __name__ == "__main__":
    main()
