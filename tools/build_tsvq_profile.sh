#!/bin/bash
# Builds tools/bin/libbasisu_hip_tqprof.so: libbasisu_hip.so with the one-workgroup split kernel instrumented (-DTQ_PROFILE: clock64() deltas of thread 0 per
# phase of a split, summed over all workgroups; tsvq_profile_read). Development aid for tools/tsvq_split_profile.py; never loaded by the product.
set -e
cd "$(dirname "$0")/../basis_universal_amd/csrc"
mkdir -p ../../tools/bin/obj
F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fvisibility=hidden -Wall -Wno-unused-function -DTQ_PROFILE"
/opt/rocm/bin/hipcc $F -c tsvq_kernels.hip -o ../../tools/bin/obj/tsvq_kernels_prof.o
OBJS=""
for o in etc1s_kernels tsvq_wide_kernels tsvq_wide6_kernels uastc_kernels uastc_rdo_kernels unique_kernels bookkeeping_kernels kmeans_kernels mipmap_kernels bu_hip_api; do OBJS="$OBJS ../lib/obj/$o.o"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/bin/libbasisu_hip_tqprof.so $OBJS ../../tools/bin/obj/tsvq_kernels_prof.o
echo built tools/bin/libbasisu_hip_tqprof.so
