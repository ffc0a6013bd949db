// host_common.h -- shared by the host-side translation units of libmerkurio_hip.so
#pragma once
#include <stdint.h>

#include <exception>
#include <new>

#include "../../include/merkurio_hip.h"

namespace mk {
// message of the calling thread's last failure (fixed buffer: reporting never allocates)
extern thread_local char g_last_error[512];
// records the message for mk_last_error() and returns `code`
int fail(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));
}  // namespace mk

// Nothing may throw across the C ABI (include/merkurio_hip.h): every extern "C" body that can
// allocate sits between these two macros.  std::bad_alloc -> MK_E_NOMEM, anything else ->
// MK_E_INVALID_ARG with the exception text.
#define MK_ABI_BEGIN try {
#define MK_ABI_END                                                                   \
    }                                                                                \
    catch (const std::bad_alloc &) {                                                 \
        return mk::fail(MK_E_NOMEM, "out of host memory");                           \
    }                                                                                \
    catch (const std::exception &e_) {                                               \
        return mk::fail(MK_E_INVALID_ARG, "unexpected exception: %s", e_.what());    \
    }                                                                                \
    catch (...) {                                                                    \
        return mk::fail(MK_E_INVALID_ARG, "unexpected exception");                   \
    }
