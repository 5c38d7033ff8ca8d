// Library-level entry points of libtkr_hip.so.
#include "tkr_common.h"
#include "../../include/tkr.h"

extern "C" int tkr_version(void) { return TKR_VERSION; }

// 1 when this library was built with `make LAB=1` (-DTKR_LAB): it then also holds the kernel forms that were measured and dropped
extern "C" int tkr_lab_build(void) {
#ifdef TKR_LAB
    return 1;
#else
    return 0;
#endif
}
