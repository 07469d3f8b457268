// Error plumbing + version for libast_hip.so.
#include <stdarg.h>
#include <stdio.h>
extern "C" {
static thread_local char g_err[512] = "";
int ast_version(void) { return 100; }
const char* ast_last_error(void) { return g_err; }
}
void ast_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
