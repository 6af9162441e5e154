// buildinfo.cpp -- identifies the sources a libdbaz_hip.so was built from (dotsboxesaz_amd/build.py passes
// -DDBAZ_SRC_HASH: sha256 over csrc/*.hip, csrc/*.h and include/dbaz.h, and the same over the network kernels alone).  bench.py compares it with the hash stored in
// profiles/*_pmc_*.json, so a counter file measured on another build of the kernels is never quoted as this build's traffic.
#include "../../include/dbaz.h"

#ifndef DBAZ_SRC_HASH
#define DBAZ_SRC_HASH "unknown"
#endif
#ifdef DBAZ_DEBUG
#define DBAZ_FLAVOUR " debug"
#else
#define DBAZ_FLAVOUR ""
#endif

extern "C" const char *dbaz_build_info(void) { return "src=" DBAZ_SRC_HASH DBAZ_FLAVOUR; }
