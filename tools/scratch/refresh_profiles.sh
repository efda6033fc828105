#!/bin/bash
# After `gpurun -- bash tools/scratch/gpu_round.sh <tag> tests bench prof pmc [calib]`: copy the round's summaries from gpurun_out/ into profiles/ and regenerate the
# derived tables (pmc_traffic.json, valu_busy.json, <tag>_valu_table.md, <tag>_resource_usage.txt) at the current commit.   usage: tools/scratch/refresh_profiles.sh <tag>
set -e
cd "$(dirname "$0")/../.."
tag=$1
commit=$(git rev-parse --short HEAD)
for f in kernel_stats pmc_fetch_size pmc_write_size pmc_sq pmc_calib; do [ -s gpurun_out/${tag}_$f.csv ] && cp gpurun_out/${tag}_$f.csv profiles/; done
[ -s gpurun_out/bench_$tag.json ] && cp gpurun_out/bench_$tag.json profiles/
[ -s gpurun_out/valu_calibration_$tag.json ] && cp gpurun_out/valu_calibration_$tag.json profiles/valu_calibration.json
python tools/rocprof_summary.py traffic_csv profiles/${tag}_pmc_fetch_size.csv profiles/${tag}_pmc_write_size.csv $commit > profiles/pmc_traffic.json
python tools/valu_table.py --calib profiles/valu_calibration.json --pmc profiles/${tag}_pmc_sq.csv --build-isa --json --commit $commit > profiles/valu_busy.json
python tools/valu_table.py --calib profiles/valu_calibration.json --pmc profiles/${tag}_pmc_sq.csv --build-isa > profiles/${tag}_valu_table.md
tools/resource_usage.sh > profiles/${tag}_resource_usage.txt 2>/dev/null
ls -la profiles | grep "$tag\|pmc_traffic\|valu_busy\|valu_calibration"
