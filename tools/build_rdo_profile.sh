#!/bin/bash
# Builds tools/bin/libbasisu_hip_rdoprof.so: libbasisu_hip.so with the RDO strips kernel instrumented (-DRDO_PROFILE: clock64() deltas per phase of
# the serial step, strip 0, printed to stderr by bu_hip_k_uastc_rdo). Development aid for tools/rdo_step_profile.py; never loaded by the product.
set -e
cd "$(dirname "$0")/../basis_universal_amd/csrc"
mkdir -p ../../tools/bin/obj
F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fvisibility=hidden -Wall -Wno-unused-function -DRDO_PROFILE"
/opt/rocm/bin/hipcc $F -c uastc_rdo_kernels.hip -o ../../tools/bin/obj/uastc_rdo_kernels.o
/opt/rocm/bin/hipcc $F -c bu_hip_api.cpp -o ../../tools/bin/obj/bu_hip_api.o
OBJS=""
for o in etc1s_kernels tsvq_kernels tsvq_wide_kernels uastc_kernels unique_kernels bookkeeping_kernels kmeans_kernels mipmap_kernels; do OBJS="$OBJS ../lib/obj/$o.o"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/bin/libbasisu_hip_rdoprof.so $OBJS ../../tools/bin/obj/uastc_rdo_kernels.o ../../tools/bin/obj/bu_hip_api.o
echo built tools/bin/libbasisu_hip_rdoprof.so
