#include "mi355x_asr.h"
#include <stdint.h>
extern "C" const char* mi355x_asr_version(void) { return "mi355x_asr 0.2 (gfx950)"; }

// Device-side step word for the dropout keys (see DropCfg::step in common.h).  Process-wide, read by the launchers at launch
// time: a launch sequence captured into a hipGraph while the pointer is set keeps the pointer in its kernel arguments.
extern "C" const uint32_t* mi355x_step_counter_ptr = nullptr;
extern "C" int mi355x_set_step_counter(const void* dev_word) {
  mi355x_step_counter_ptr = (const uint32_t*)dev_word;
  return 0;
}

// measurement switch (see MI_LAUNCH in common.h): 1 = every launch site issues an empty kernel instead of its own
extern "C" int mi355x_null_launch_flag = 0;
extern "C" int mi355x_set_null_launch(int on) {
  const int old = mi355x_null_launch_flag;
  mi355x_null_launch_flag = on ? 1 : 0;
  return old;
}
