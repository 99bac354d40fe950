// geoac_buildid.cpp - the identity of a build of libgeoac_hip.so: a hash of the sources, headers and flags it is compiled from (Makefile: SRCID).  An object of its
// own, recompiled exactly when that hash changes - hipcc's output is not bit-reproducible, so the file's own hash cannot say "the same build" (include/geoac_hip.h).
#ifndef GEOAC_SOURCE_ID
#define GEOAC_SOURCE_ID "unknown"
#endif
extern "C" const char* geoac_build_id(void){ return GEOAC_SOURCE_ID; }
