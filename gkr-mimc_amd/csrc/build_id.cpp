// build_id.cpp -- the SHA-256 of the sources and flags this library was built from (gkr-mimc_amd/build.py finds the marker in
// the file and compares it with the sources on disk; the loader refuses a library built from other sources).  A unit of its own:
// the hash changes with every edit, the other units are rebuilt only when what they include changed.
#ifndef GKRHIP_SOURCE_SHA
#define GKRHIP_SOURCE_SHA "unrecorded"
#endif
extern "C" __attribute__((visibility("default"))) const char* gkrhip_build_id(void) {
    static const char id[] = "GKRHIP_SOURCE_SHA=" GKRHIP_SOURCE_SHA;
    return id + sizeof("GKRHIP_SOURCE_SHA=") - 1;
}
