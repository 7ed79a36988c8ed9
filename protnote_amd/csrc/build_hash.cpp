// The source hash of this build (protnote_amd/build.py: csrc_hash() over csrc/*.hip, *.hpp, *.cpp and the API header),
// passed as -DPN_CSRC_HASH="...".  The marker prefix lets build.py read it from the file's bytes without loading it.
#include "../../include/protnote_hip.h"

#ifndef PN_CSRC_HASH
#error "build with protnote_amd/build.py (it passes -DPN_CSRC_HASH)"
#endif

static const char pn_hash_marker[] = "PN_CSRC_HASH=" PN_CSRC_HASH;

extern "C" const char* pn_build_hash(void) { return pn_hash_marker + 13; }
