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


if __name__ == "__main__":
    d = sqlite3.connect(sys.argv[2])
    {"stats": stats, "pmc": pmc}[sys.argv[1]](d)
