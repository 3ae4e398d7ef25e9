// persistent_sweep.hip -- sweep_kernel variants (see persistent_tables.h)
#include <hip/hip_runtime.h>

#include <climits>
#include <cmath>
#include <cstdint>

#include "../../include/yalps_hip.h"
#include "persistent_tables.h"

#pragma clang fp contract(off)

namespace {
#include "common.cuh"

#include "resident_kernel.cuh" // (the sc1 load / store helpers)
#include "sweep_kernel.cuh"
// (R = 1: non-temporal row traffic, for tableaux beyond the Infinity Cache)
#define WVARIANT(T, J, C, NT) {T, J, NT, reinterpret_cast<const void *>(&sweep_kernel<T, J, C, NT != 0>)}
} // namespace
int yalps_sweep_sync_bytes() { return (int)sizeof(SweepSync); }
// 512 lanes x 8 / 16 units: a 1024-lane workgroup leaves each lane 128 registers, not enough for two 8-unit rows in flight
// beside the rest of the loop (it spilled); 512 lanes with 256 registers stream as fast (6.07 against 6.18 TB/s measured).
PersistentTable yalps_sweep_table() {
    static const PersistentEntry kSweep[] = {WVARIANT(512, 8, false, 0), WVARIANT(512, 16, false, 0), WVARIANT(512, 8, false, 1),
                                             WVARIANT(512, 16, false, 1)};
    return {kSweep, (int)(sizeof kSweep / sizeof kSweep[0])};
}
PersistentTable yalps_sweep_check_table() {
    // (<512,16,true> needs a few registers more than there are: checkCycles on rows wider than 8193 columns stays with the
    // DECIDE + APPLY launches)
    static const PersistentEntry kSweepCheck[] = {WVARIANT(512, 8, true, 0), WVARIANT(512, 8, true, 1)};
    return {kSweepCheck, (int)(sizeof kSweepCheck / sizeof kSweepCheck[0])};
}
