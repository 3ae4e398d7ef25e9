// persistent_dshard.hip -- dshard_kernel variants: row shards with delayed row updates (see persistent_tables.h)
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
#include "sweep_kernel.cuh" // (the buffer-descriptor row accessors)
#include "panel_flush.cuh"
#include "dshard_kernel.cuh"
// (R = 1: non-temporal row traffic, for shards beyond the Infinity Cache)
#define DSVARIANT(T, J, NT) {T, J, NT, reinterpret_cast<const void *>(&dshard_kernel<T, J, NT != 0>)}
} // namespace
PersistentTable yalps_dshard_table() {
    static const PersistentEntry kDshard[] = {DSVARIANT(512, 16, 0), DSVARIANT(512, 16, 1), DSVARIANT(512, 8, 0), DSVARIANT(512, 8, 1),
                                              DSVARIANT(512, 6, 0),  DSVARIANT(512, 6, 1),  DSVARIANT(512, 4, 0), DSVARIANT(512, 4, 1),
                                              DSVARIANT(512, 2, 0),  DSVARIANT(512, 2, 1),  DSVARIANT(512, 1, 0), DSVARIANT(512, 1, 1)};
    return {kDshard, (int)(sizeof kDshard / sizeof kDshard[0])};
}
const void *yalps_dshard_select_fn() { return reinterpret_cast<const void *>(&dshard_select_kernel); }
