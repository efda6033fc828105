#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd database (bench_results.db) into a small CSV for profiles/.

  kernel stats : python tools/rocprof_summary.py stats <db> > profiles/<name>.csv
  PMC counters : python tools/rocprof_summary.py pmc <db> > profiles/<name>.csv   (per kernel: launches, mean/sum of each counter)
"""
import sqlite3
import sys


def cols(db, view):
    return [r[1] for r in db.execute(f"pragma table_info({view})")]


def stats(db):
    c = cols(db, "kernels")
    name = "name" if "name" in c else "kernel_name"
    rows = db.execute(f"select {name}, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels group by {name} order by 3 desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    print("name,calls,total_ns,avg_ns,pct,min_ns,max_ns")
    for n, k, t, a, lo, hi in rows:
        print(f"\"{n}\",{k},{t},{a:.1f},{100.0 * t / total:.2f},{lo},{hi}")


def pmc(db):
    c = cols(db, "counters_collection")
    sys.stderr.write("counters_collection columns: %s\n" % c)
    kname = "kernel_name" if "kernel_name" in c else "name"
    cname = "counter_name" if "counter_name" in c else "pmc_name"
    val = "value" if "value" in c else "counter_value"
    disp = "dispatch_id" if "dispatch_id" in c else "id"
    # one row per (dispatch, counter[, dimension instance]): sum the instances of a dispatch first
    q = (f"select {kname}, {cname}, count(*), avg(v), sum(v) from (select {kname}, {cname}, {disp}, sum({val}) as v "
         f"from counters_collection group by {kname}, {cname}, {disp}) group by {kname}, {cname} order by 5 desc")
    print("kernel,counter,dispatches,mean_per_dispatch,sum")
    for n, cn, k, a, s in db.execute(q):
        print(f"\"{n}\",{cn},{k},{a:.1f},{s:.1f}")


def short(name):
    """bench.py's kernel label for a mangled-demangled kernel name: bu::k_foo<...>(...) -> foo"""
    import re
    m = re.search(r"k_(\w+?)(?:<|\()", name)
    n = m.group(1) if m else name
    if n in ("tsvq_split", "tsvq_root"):  # the two instantiations are timed separately (selector vectors packed in a dword / 6-float endpoint vectors)
        n += "_packed16" if "packed16_rows" in name else "_float6"
    return {"refine_endpoint_clusterization": "refine_endpoint_clusterization", "fosc_resolve_and_stamp": "find_optimal_selector_clusters_stamp",
            "encode_etc1s_blocks_by_pixel": "encode_etc1s_blocks", "refine_sorted": "refine_endpoint_clusterization",
            "find_optimal_selector_clusters": "find_optimal_selector_clusters"}.get(n, n)


def traffic(fetch_db, write_db):
    """JSON {kernel: {fetch_bytes_per_launch, write_bytes_per_launch}}: mean per dispatch, FETCH_SIZE/WRITE_SIZE are KiB; FETCH doubled (gfx950)."""
    import json
    out = {}
    for db, key, mul in ((fetch_db, "fetch_bytes_per_launch", 2048.0), (write_db, "write_bytes_per_launch", 1024.0)):
        d = sqlite3.connect(db)
        q = ("select kernel_name, avg(v), count(*) from (select kernel_name, dispatch_id, sum(value) as v from counters_collection "
             "group by kernel_name, dispatch_id) group by kernel_name")
        acc = {}
        for n, a, k in d.execute(q):
            s_ = short(n)
            tot, cnt = acc.get(s_, (0.0, 0))
            acc[s_] = (tot + a * k, cnt + k)
        for s_, (tot, cnt) in acc.items():
            out.setdefault(s_, {"fetch_bytes_per_launch": 0, "write_bytes_per_launch": 0})[key] = int(tot / cnt * mul)
    print(json.dumps(out, indent=1, sort_keys=True))


def traffic_csv(fetch_csv, write_csv, commit):
    """The same from two `pmc` CSVs (what a GPU-box run brings home when the databases are too large to). The many-workgroup TSVQ path is
    one bench label ("tsvq_split_packed16_wide" = one batch of wide nodes = ~40 launches of the k_wide_* kernels + k_tsvq_cov_axis): its
    entry is the traffic of all of them per BATCH (batches = launches of k_wide_partition)."""
    import csv, json
    out, wide, batches = {}, {"fetch_bytes_per_launch": 0.0, "write_bytes_per_launch": 0.0}, 0
    for path, key, mul in ((fetch_csv, "fetch_bytes_per_launch", 2048.0), (write_csv, "write_bytes_per_launch", 1024.0)):
        acc = {}
        for r in csv.DictReader(open(path)):
            n, k, tot = r["kernel"], int(r["dispatches"]), float(r["sum"]) * mul
            if "k_wide_" in n or "k_tsvq_cov_axis" in n:
                if "k_wide_iota" in n or "<0>" in n:
                    continue  # the root record's pass belongs to tsvq_root_packed16
                wide[key] += tot
                if "k_wide_partition" in n:
                    batches = k
                continue
            if "rocprim" in n or n.startswith("__amd_rocclr") or n.startswith("_Z"):
                continue  # library kernels (sorts, scans, copies) and the k-means helpers of the fast mode: not part of the per-kernel table
            s_ = short(n)
            a, c = acc.get(s_, (0.0, 0))
            acc[s_] = (a + tot, c + k)
        for s_, (tot, cnt) in acc.items():
            out.setdefault(s_, {"fetch_bytes_per_launch": 0, "write_bytes_per_launch": 0})[key] = int(tot / cnt)
    if batches:
        out["tsvq_split_packed16_wide"] = {k: int(v / batches) for k, v in wide.items()}
    out["_meta"] = {"commit": commit, "source": [fetch_csv, write_csv], "units": "bytes per launch; FETCH_SIZE x 2048 (KiB, doubled per the gfx950 note in MI355X_MICROARCH.md), WRITE_SIZE x 1024"}
    print(json.dumps(out, indent=1, sort_keys=True))


if __name__ == "__main__":
    if sys.argv[1] == "traffic_csv":
        traffic_csv(sys.argv[2], sys.argv[3], sys.argv[4] if len(sys.argv) > 4 else "")
        sys.exit(0)
    if sys.argv[1] == "traffic":
        traffic(sys.argv[2], sys.argv[3])
        sys.exit(0)
    d = sqlite3.connect(sys.argv[2])
    {"stats": stats, "pmc": pmc}[sys.argv[1]](d)
