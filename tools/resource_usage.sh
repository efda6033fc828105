#!/bin/bash
# Register / scratch / LDS / occupancy of every kernel of libbasisu_hip.so, as hipcc reports them for gfx950 (-Rpass-analysis=kernel-resource-usage),
# one line per kernel:   tools/resource_usage.sh > profiles/<tag>_resource_usage.txt
cd "$(dirname "$0")/../basis_universal_amd/csrc"
echo "# hipcc -O3 --offload-arch=gfx950 -Rpass-analysis=kernel-resource-usage, $(git rev-parse --short HEAD 2>/dev/null)"
echo "# file | kernel | VGPRs | AGPRs | SGPRs | scratch B/lane | LDS B/workgroup | occupancy waves/SIMD"
for f in etc1s_kernels tsvq_kernels tsvq_wide_kernels tsvq_wide6_kernels unique_kernels bookkeeping_kernels kmeans_kernels mipmap_kernels uastc_kernels uastc_rdo_kernels; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off --cuda-device-only -c -Rpass-analysis=kernel-resource-usage -o /dev/null $f.hip 2>&1 |
  awk -v F=$f '
    /Function Name:/ { name=$0; sub(/.*Function Name: /,"",name); sub(/ \[-Rpass.*/,"",name) }
    /TotalSGPRs:/ { sg=$0; sub(/.*TotalSGPRs: /,"",sg); sub(/ .*/,"",sg) }
    / VGPRs:/ { vg=$0; sub(/.* VGPRs: /,"",vg); sub(/ .*/,"",vg) }
    /AGPRs:/ { ag=$0; sub(/.*AGPRs: /,"",ag); sub(/ .*/,"",ag) }
    /ScratchSize/ { sc=$0; sub(/.*: /,"",sc); sub(/ .*/,"",sc) }
    /Occupancy/ { oc=$0; sub(/.*: /,"",oc); sub(/ .*/,"",oc) }
    /LDS Size/ { ld=$0; sub(/.*: /,"",ld); sub(/ .*/,"",ld); print F " | " name " | " vg " | " ag " | " sg " | " sc " | " ld " | " oc }'
done | while IFS= read -r line; do
  sym=$(echo "$line" | cut -d'|' -f2 | tr -d ' ')
  dem=$(echo "$sym" | c++filt | sed 's/(anonymous namespace):://g; s/^void //; s/bu:://g' | sed 's/(.*//')
  case "$dem" in _Z*) dem=$(echo "$sym" | grep -o 'k_[a-z_0-9]*[a-z0-9]' | head -1) ;; esac   # c++filt does not know _Float16 parameter types
  case "$dem" in k_*|k6_*) ;; *) continue ;; esac   # this library's kernels only (hipCUB / rocPRIM instantiations are not ours to tune)
  echo "$line" | awk -F'|' -v D="$dem" '{print $1 "| " D " |" $3 "|" $4 "|" $5 "|" $6 "|" $7 "|" $8}'
done
