// api.cpp — version + thread-local error string of the C-ABI (include/forge_hip.h).
#include <cstdarg>
#include <cstdio>

#include "../../include/forge_hip.h"

namespace forge {
static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace forge

extern "C" int forge_version(void) { return 210; /* 0.2.1: round-2 ABI (conv out3 / GRU residual, rotate slots, strided GRU backward, loss / camera / Winograd entry points) */ }
extern "C" const char* forge_last_error(void) { return forge::g_err; }
