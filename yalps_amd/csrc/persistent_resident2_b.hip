// persistent_resident2_b.hip -- resident2_kernel variants, part 2 (see persistent_tables.h)
#include <hip/hip_runtime.h>

#include <climits>
#include <cmath>
#include <cstdint>

#include "../../include/yalps_hip.h"
#include "persistent_tables.h"

#pragma clang fp contract(off)

namespace {
#include "common.cuh"

#include "resident_kernel.cuh"

#include "resident2_kernel.cuh"
#define RVARIANT(T, J, R) {T, J, R, reinterpret_cast<const void *>(&resident2_kernel<T, J, R>)}
} // namespace
PersistentTable yalps_resident2_table_b() { // (the shapes of yalps_resident_table_b)
    static const PersistentEntry kEntries[] = {
    RVARIANT(512, 2, 16),
    RVARIANT(512, 3, 4), RVARIANT(512, 3, 6), RVARIANT(512, 3, 9), RVARIANT(512, 3, 12),
    RVARIANT(512, 4, 4), RVARIANT(512, 4, 6), RVARIANT(512, 4, 8),
    RVARIANT(512, 5, 4), RVARIANT(512, 5, 6),
    RVARIANT(512, 6, 4),
};
    return {kEntries, (int)(sizeof kEntries / sizeof kEntries[0])};
}
