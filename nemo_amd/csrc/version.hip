#include "mi355x_asr.h"
extern "C" const char* mi355x_asr_version(void) { return "mi355x_asr 0.1 (gfx950)"; }
