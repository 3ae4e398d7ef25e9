// persistent_stream3.hip -- stream3_kernel variants (see persistent_tables.h)
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
#include "stream3_kernel.cuh"
// (R = NT | 2: non-temporal row traffic for tableaux beyond the Infinity Cache; 2 = the sweep through LDS panels -- this unit;
// persistent_stream3d.hip holds the forms that read the pending rows straight from L2, for tableaux with few rows per workgroup)
#define S3VARIANT(T, J, NT) {T, J, (NT) | 2, reinterpret_cast<const void *>(&stream3_kernel<T, J, NT != 0, false, true>)}
} // namespace
PersistentTable yalps_stream3_table() {
    static const PersistentEntry kStream3[] = {S3VARIANT(512, 16, 0), S3VARIANT(512, 16, 1), S3VARIANT(512, 8, 0), S3VARIANT(512, 8, 1),
                                               S3VARIANT(512, 6, 0), S3VARIANT(512, 6, 1), S3VARIANT(512, 4, 0), S3VARIANT(512, 4, 1),
                                               S3VARIANT(512, 2, 0), S3VARIANT(512, 2, 1), S3VARIANT(512, 1, 0), S3VARIANT(512, 1, 1)};
    return {kStream3, (int)(sizeof kStream3 / sizeof kStream3[0])};
}
#define S3CHECK(T, J, NT) {T, J, (NT) | 2, reinterpret_cast<const void *>(&stream3_kernel<T, J, NT != 0, true, true>)}
PersistentTable yalps_stream3_check_table() { // options.checkCycles
    static const PersistentEntry kStream3Check[] = {/* (512, 16): persistent_stream3d.hip's form only -- see yalps_hip.hip */ S3CHECK(512, 8, 0), S3CHECK(512, 8, 1),
                                                    S3CHECK(512, 6, 0), S3CHECK(512, 6, 1), S3CHECK(512, 4, 0), S3CHECK(512, 4, 1),
                                                    S3CHECK(512, 2, 0), S3CHECK(512, 2, 1), S3CHECK(512, 1, 0), S3CHECK(512, 1, 1)};
    return {kStream3Check, (int)(sizeof kStream3Check / sizeof kStream3Check[0])};
}
