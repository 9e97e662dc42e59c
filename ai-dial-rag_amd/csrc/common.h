// Shared host-side helpers for libmiretr.so (gfx950 only; no dual paths).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/miretr.h"

namespace mir {

// thread-local error text behind mir_last_error()
void set_error(const char *fmt, ...) __attribute__((format(printf, 1, 2)));

#define MIR_HIP(call)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (call);                                                                    \
        if (e_ != hipSuccess) {                                                                    \
            ::mir::set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__,      \
                             __LINE__);                                                            \
            return MIR_ERR_HIP;                                                                    \
        }                                                                                          \
    } while (0)

#define MIR_REQUIRE(cond, ...)                                                                     \
    do {                                                                                           \
        if (!(cond)) {                                                                             \
            ::mir::set_error(__VA_ARGS__);                                                         \
            return MIR_ERR_INVALID;                                                                \
        }                                                                                          \
    } while (0)

// Picks `device` (or fails with MIR_ERR_NO_DEVICE) and reports its CU count.
int32_t use_device(int32_t device, int *num_cus);

}  // namespace mir
