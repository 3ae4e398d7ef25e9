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
    RVARIANT(512, 3, 4), // (the other <512,3..6,*> shapes spill: see persistent_resident2_a.hip)
};
    return {kEntries, (int)(sizeof kEntries / sizeof kEntries[0])};
}
