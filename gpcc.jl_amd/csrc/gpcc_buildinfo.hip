// gpcc_buildinfo.hip -- what the library was built from (gpcc.jl_amd/build.py passes the record; include/gpcc_hip.h: gpcc_build_info)
#ifndef GPCC_BUILD_INFO_STR
#define GPCC_BUILD_INFO_STR "src=unknown defines=[?]"   /* (a build that did not go through gpcc.jl_amd/build.py) */
#endif
extern "C" const char *gpcc_build_info(void) { return GPCC_BUILD_INFO_STR; }
