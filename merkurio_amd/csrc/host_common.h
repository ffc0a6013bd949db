// host_common.h -- shared by the host-side translation units of libmerkurio_hip.so
#pragma once
#include <stdint.h>

#include <string>

#include "../../include/merkurio_hip.h"

namespace mk {
extern thread_local std::string g_last_error;
// records the message for mk_last_error() and returns `code`
int fail(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));
}  // namespace mk
