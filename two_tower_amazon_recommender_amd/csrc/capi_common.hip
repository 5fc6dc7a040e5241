// tt_abi_version / tt_last_error and the error plumbing shared by every entry point.
#include "common.h"

namespace tt {
static thread_local char g_err[512] = "";
char* err_buf() { return g_err; }
int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}
}  // namespace tt

extern "C" int tt_abi_version(void) { return TT_ABI_VERSION; }
extern "C" const char* tt_last_error(void) { return tt::err_buf(); }
