// Host-side helpers shared by the C-ABI translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>

#include "leco_hip.h"

namespace leco {
// thread-local message returned by leco_last_error()
char* error_buffer();
int fail(int code, const char* fmt, ...);
// after a launch: translate hipGetLastError() into the ABI's return convention
int check_launch(const char* what);
static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }
}  // namespace leco
