#include "pwr.h"
extern "C" int pwr_abi_version(void) { return PWR_ABI_VERSION; }
