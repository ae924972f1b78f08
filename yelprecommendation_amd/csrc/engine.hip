// Library identity / load check for the gfx950 engine.
#include "common.h"

extern "C" int yr_engine_version(void) { return YR_ENGINE_VERSION; }

extern "C" const char* yr_engine_arch(void) { return "gfx950"; }
