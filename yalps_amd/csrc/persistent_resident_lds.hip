// persistent_resident_lds.hip -- resident_kernel variants with rows parked in LDS (see persistent_tables.h)
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
static_assert(XROWS == YALPS_RESIDENT_LDS_MAX_ROWS, "host and kernel agree on the LDS row budget");
#define RVARIANT(T, J, R) {T, J, R, reinterpret_cast<const void *>(&resident_kernel<T, J, R, true>)}
} // namespace
PersistentTable yalps_resident_lds_table() { // (a function-local table: filled on first use, whatever the order of static initialisation)
    static const PersistentEntry kEntries[] = {
    // (one register row less than the plain variants where those sit at the 256-VGPR cap: the LDS code needs a few)
    RVARIANT(512, 1, 38), RVARIANT(512, 2, 16), RVARIANT(512, 3, 11), RVARIANT(512, 4, 7), RVARIANT(512, 5, 5), RVARIANT(512, 6, 3),
};
    return {kEntries, (int)(sizeof kEntries / sizeof kEntries[0])};
}
