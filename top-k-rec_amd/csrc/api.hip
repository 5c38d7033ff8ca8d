// Library-level entry points of libtkr_hip.so.
#include "tkr_common.h"
#include "../../include/tkr.h"

extern "C" int tkr_version(void) { return TKR_VERSION; }
