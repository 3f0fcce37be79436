// Shared between the translation units of libnbody_mi355x.so; not installed.
#pragma once
#include <cstdarg>

#include "nbody_mi355x.h"

namespace nbody_internal {
// Text behind nbody_mi355x_last_error() (thread-local).
void set_error(const char* fmt, ...) __attribute__((format(printf, 1, 2)));
}  // namespace nbody_internal
