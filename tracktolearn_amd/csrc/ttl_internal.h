// ttl_internal.h -- shared by the translation units of libttl_hip.so (not
// part of the ABI).
#ifndef TTL_INTERNAL_H
#define TTL_INTERNAL_H
#include <hip/hip_runtime.h>

#include "ttl_hip.h"

// records the message returned by ttl_last_error() (thread local) and
// returns `code`
int ttl_detail_fail(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));
#define fail ttl_detail_fail

#define HIP_TRY(expr)                                                        \
    do {                                                                     \
        hipError_t e_ = (expr);                                              \
        if (e_ != hipSuccess)                                                \
            return fail(TTL_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

constexpr int TTL_BLOCK = 256;   // threads per workgroup of every kernel
#endif
