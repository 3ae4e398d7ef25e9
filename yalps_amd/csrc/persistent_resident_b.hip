// persistent_resident_b.hip -- resident_kernel variants, part 2 (see persistent_tables.h)
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
#define RVARIANT(T, J, R) {T, J, R, reinterpret_cast<const void *>(&resident_kernel<T, J, R>)}
} // namespace
PersistentTable yalps_resident_table_b() { // (a function-local table: filled on first use, whatever the order of static initialisation)
    static const PersistentEntry kEntries[] = {
    RVARIANT(512, 2, 16),
    RVARIANT(512, 3, 4), RVARIANT(512, 3, 6), RVARIANT(512, 3, 9), RVARIANT(512, 3, 12),
    RVARIANT(512, 4, 4), RVARIANT(512, 4, 6), RVARIANT(512, 4, 8), // (<512,4,9> spills one VGPR on top of 119 SGPRs)
    RVARIANT(512, 5, 4), RVARIANT(512, 5, 6),
    RVARIANT(512, 6, 4),
    // (<1024,1,9> fits its 128 VGPRs now but is no faster at 2049^2: 6.94 against 6.82 us/pivot)
    // (no variant may need AGPRs -- <256,1,32> (374 registers) left its last row slots unwritten on the
    // GPU -- or scratch: SGPR spills that end up in scratch computed garbage in stream_kernel<1024,8>;
    // tests/test_cabi_symbols.py checks the register counts of the built code object)
};
    return {kEntries, (int)(sizeof kEntries / sizeof kEntries[0])};
}
