// persistent_resident_tag.hip -- resident_kernel variants with tagged candidate rows (see persistent_tables.h)
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
#define RVARIANT(T, J, R) {T, J, R, reinterpret_cast<const void *>(&resident_kernel<T, J, R, false, true>)}
} // namespace
PersistentTable yalps_resident_tag_table() { // (a function-local table: filled on first use, whatever the order of static initialisation)
    static const PersistentEntry kEntries[] = {
        // Measured (whole solves, us per pivot): <256,1,4> 5.2 -> 4.5 (201^2 .. 513^2, 601x301); <256,1,9> 5.66 -> 5.63 and
        // <256,1,16> 7.35 -> 7.60: no gain with more rows per workgroup; with two or more units per lane the doubled
        // payload costs more than the round trip it saves (1025^2 5.65 -> 6.16, 2049^2 6.86 -> 7.93).  So: this one.
        RVARIANT(256, 1, 4),
    };
    return {kEntries, (int)(sizeof kEntries / sizeof kEntries[0])};
}
