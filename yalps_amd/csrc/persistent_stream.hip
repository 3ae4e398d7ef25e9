// persistent_stream.hip -- stream_kernel variants (see persistent_tables.h)
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
#include "stream_kernel.cuh"
#define SVARIANT(T, J, C) {T, J, 0, reinterpret_cast<const void *>(&stream_kernel<T, J, C>)}
} // namespace
PersistentTable yalps_stream_table() {
    static const PersistentEntry kStream[] = {
    SVARIANT(256, 1, false), SVARIANT(256, 2, false), SVARIANT(1024, 1, false), SVARIANT(1024, 2, false), SVARIANT(1024, 4, false),
    // (<1024,8> needs 77 VGPR + 118 SGPR spills at the 128-register cap and computed garbage on the GPU in round 1: rows
    // wider than 8193 columns take sweep_kernel / stream3_kernel.  -DYALPS_EXPERIMENT_STREAM8 builds it for the experiment of
    // DESIGN.md 4.8 -- tools/build_variant.sh stream8 -DYALPS_EXPERIMENT_STREAM8 stream --, never in the shipped library.)
#ifdef YALPS_EXPERIMENT_STREAM8
    SVARIANT(1024, 8, false),
#endif
};
    return {kStream, (int)(sizeof kStream / sizeof kStream[0])};
}
PersistentTable yalps_stream_check_table() {
    static const PersistentEntry kStreamCheck[] = { // checkCycles (<1024,4,true> would spill: those tableaux keep DECIDE + APPLY launches)
    SVARIANT(256, 1, true), SVARIANT(256, 2, true), SVARIANT(1024, 1, true), SVARIANT(1024, 2, true),
};
    return {kStreamCheck, (int)(sizeof kStreamCheck / sizeof kStreamCheck[0])};
}
