// persistent_tables.h -- the instantiations of the persistent kernels (resident_kernel.cuh, stream_kernel.cuh) are
// compiled in translation units of their own (persistent_*.hip: the build compiles them side by side); the host
// side (yalps_hip.hip) sees them through these tables.  A kernel is passed as the address of its host stub: every
// translation unit includes common.cuh inside its own unnamed namespace, so `Desc` is formally a different (identical)
// type in each, and the host casts the address back to void (*)(Desc, int parity, int chunk) before the launch.
#pragma once

struct PersistentEntry {
    int T, J, R;    // lanes, 16-byte units per lane and row, rows per workgroup in registers (stream_kernel: 0)
    const void *fn; // __global__ void (Desc, int, int)
};
struct PersistentTable {
    const PersistentEntry *entries;
    int count;
};
// resident_kernel<T, J, R>: tableau in the register files
PersistentTable yalps_resident_table_a();
PersistentTable yalps_resident_table_b();
// resident2_kernel<T, J, R>: the same shapes with the second-generation pivot loop (resident2_kernel.cuh)
PersistentTable yalps_resident2_table_a();
PersistentTable yalps_resident2_table_b();
// resident_kernel<T, J, R, true>: up to XROWS more rows per workgroup parked in LDS
PersistentTable yalps_resident_lds_table();
constexpr int YALPS_RESIDENT_LDS_MAX_ROWS = 8;
// resident_kernel<T, J, R, false, true>: the candidate row travels as self-validating granules (narrow rows)
PersistentTable yalps_resident_tag_table();
// sweep_kernel<T, J, false> / <T, J, true>: persistent, in place, for tableaux that stream from HBM (sweep_kernel.cuh)
PersistentTable yalps_sweep_table();
PersistentTable yalps_sweep_check_table();
int yalps_sweep_sync_bytes(); // sizeof(SweepSync): its records at the head of the sweep part of the control block
// stream_kernel<T, J, false> / <T, J, true> (with hasCycle): persistent, in place
PersistentTable yalps_stream_table();
PersistentTable yalps_stream_check_table();
// stream2_kernel<T, J, NT>: the same with the row updates delayed by one pivot -- two pivots per sweep (stream2_kernel.cuh); R = NT
PersistentTable yalps_stream2_table();
// stream3_kernel<T, J, NT>: the same for rows of 8194 .. 16385 columns: objective replica in LDS, pending pivot rows in a global scratch
PersistentTable yalps_stream3_table();
PersistentTable yalps_stream3_check_table(); // ... with hasCycle (options.checkCycles)
PersistentTable yalps_stream3d_table();      // the same kernels with the pending rows read straight from L2 in the sweep (few rows per workgroup); R = NT | PANEL << 1
PersistentTable yalps_stream3d_check_table();
// dshard_kernel<T, J, NT>: one pivot of a row shard with delayed row updates -- __global__ void (Desc, int parity, int, int,
// const double *gather), launch-per-pivot like wide_kernel in MODE_SHARD; R = NT.  dshard_select_kernel: (Desc, int parity, double *send)
PersistentTable yalps_dshard_table();
const void *yalps_dshard_select_fn(int lanes); // lanes per workgroup: 1024, or 256 for shards of at most 256 workgroups
// dshard_sweep_kernel<NT> (dsweep_kernel.cuh): the shard's sweep as a launch of its own, (Desc, int parity), 512 lanes, depth x 8 KB of LDS
const void *yalps_dshard_sweep_fn(int nt);
// exchange_floor_kernel<T, J>: the bare hand-off of the resident kernels, for bench.py's measured on-chip floor
// (persistent_floor.hip): __global__ void (double *rows, unsigned long long *flags, int32_t *err, double *sink, int epochs, int variant)
const void *yalps_exchange_floor_fn(int lanes, int units);
