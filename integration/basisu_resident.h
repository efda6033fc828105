// integration/basisu_resident.h -- shared by basisu_resident_frontend.cpp and basisu_resident_backend.cpp: which resident (HIP)
// frontend stands behind a given reference basisu_frontend object. OURS; no reference code.
#pragma once
#include "basisu_hip_backend.h"   // this repository: include/ (pulls in basisu_hip_frontend.h and basisu_hip.h)

namespace basisu { class basisu_frontend; }

// The bu_frontend that basisu_frontend::compress() ran on (nullptr: none). Owned by the registry: destroyed when the same
// basisu_frontend object is initialised again, and at process exit.
bu_frontend* bu_resident_handle(const basisu::basisu_frontend* fe);

// Ends the resident state behind `fe` (device buffers back to the context's pool, registry entry erased). The reference's classes have no
// destructor hook, so the host calls this when it is done: basisu_resident_backend.cpp does at the end of basisu_backend::encode(), after
// which the basisu_frontend object still serves its getters from its own host members.
void bu_resident_release(const basisu::basisu_frontend* fe);
