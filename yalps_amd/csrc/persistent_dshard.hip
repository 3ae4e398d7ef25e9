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
// (R = NT | PANEL << 1: non-temporal row traffic for shards beyond the Infinity Cache; the sweep through LDS panels for shards with many
// rows per workgroup, straight from L2 for those with few)
#define DSVARIANT(T, J, NT, PANEL) {T, J, (NT) | ((PANEL) << 1), reinterpret_cast<const void *>(&dshard_kernel<T, J, NT != 0, PANEL != 0>)}
#define DSVARIANTS(T, J) DSVARIANT(T, J, 0, 0), DSVARIANT(T, J, 1, 0), DSVARIANT(T, J, 0, 1), DSVARIANT(T, J, 1, 1)
} // namespace
PersistentTable yalps_dshard_table() {
    static const PersistentEntry kDshard[] = {DSVARIANTS(512, 16), DSVARIANTS(512, 8), DSVARIANTS(512, 6), DSVARIANTS(512, 4), DSVARIANTS(512, 2), DSVARIANTS(512, 1)};
    return {kDshard, (int)(sizeof kDshard / sizeof kDshard[0])};
}
const void *yalps_dshard_select_fn(int lanes) {
    return lanes == 256 ? reinterpret_cast<const void *>(&dshard_select_kernel<256>) : reinterpret_cast<const void *>(&dshard_select_kernel<1024>);
}
