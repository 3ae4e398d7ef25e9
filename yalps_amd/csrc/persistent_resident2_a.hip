// persistent_resident2_a.hip -- resident2_kernel variants, part 1 (see persistent_tables.h)
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
PersistentTable yalps_resident2_table_a() { // (the shapes of yalps_resident_table_a)
    // Only the shapes hipcc builds within the register budget: the scalar control flow of the second generation costs
    // more live registers; the others (taller / wider per workgroup) spill and stay on resident_kernel.
    static const PersistentEntry kEntries[] = {
    RVARIANT(256, 1, 4), RVARIANT(256, 1, 9),
    RVARIANT(256, 2, 4),
    RVARIANT(512, 2, 4), RVARIANT(512, 2, 6), RVARIANT(512, 2, 9),
};
    return {kEntries, (int)(sizeof kEntries / sizeof kEntries[0])};
}
