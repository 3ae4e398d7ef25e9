// persistent_stream2.hip -- stream2_kernel variants (see persistent_tables.h)
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
#include "stream2_kernel.cuh"
// (R = 1: non-temporal row traffic, for tableaux beyond the Infinity Cache)
#define S2VARIANT(T, J, NT) {T, J, NT, reinterpret_cast<const void *>(&stream2_kernel<T, J, NT != 0>)}
} // namespace
PersistentTable yalps_stream2_table() {
    static const PersistentEntry kStream2[] = {
        // (one variant per row span T * J; 4096 units: 512 lanes x 8 -- three 8-unit row buffers need the 256 registers of a
        // 512-lane workgroup, <1024,4> has 128 and room for one)
        S2VARIANT(256, 1, 0), S2VARIANT(256, 2, 0), S2VARIANT(1024, 1, 0), S2VARIANT(1024, 2, 0), S2VARIANT(512, 8, 0),
        S2VARIANT(1024, 2, 1), S2VARIANT(512, 8, 1),
    };
    return {kStream2, (int)(sizeof kStream2 / sizeof kStream2[0])};
}
