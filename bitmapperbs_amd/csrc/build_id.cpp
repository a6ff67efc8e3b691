// bitmapperbs_amd/csrc/build_id.cpp -- sha256 (first 16 hex digits) of the library's sources, handed in by the Makefile: a run can show that
// the .so it loaded was built from the sources next to it.  A translation unit of its own so that a change to one kernel file does
// not rebuild the others.
#include "../../include/bmbs.h"
#ifndef BMBS_BUILD_ID
#define BMBS_BUILD_ID "unknown"
#endif
extern "C" const char* bmbs_build_id(void) { return BMBS_BUILD_ID; }
