// Version / error-string entry points of libtmdiff_hip.so.
#include "common.h"

namespace tmdiff {
char* last_error_buf() {
  static thread_local char buf[512] = "";
  return buf;
}
}  // namespace tmdiff

extern "C" int tmdiff_version(void) { return TMDIFF_ABI_VERSION; }
extern "C" const char* tmdiff_last_error_string(void) { return tmdiff::last_error_buf(); }
