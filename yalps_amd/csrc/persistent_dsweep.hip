// persistent_dsweep.hip -- dshard_sweep_kernel (see persistent_tables.h)
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
#include "dsweep_kernel.cuh"
} // namespace
const void *yalps_dshard_sweep_fn(int nt) {
    return nt ? reinterpret_cast<const void *>(&dshard_sweep_kernel<true>) : reinterpret_cast<const void *>(&dshard_sweep_kernel<false>);
}
