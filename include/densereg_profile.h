/* densereg_profile.h -- instrumentation of libdensereg_hip.so (exported by the product library next to densereg.h):
 * per-kernel HIP-event timing of the executors (bench.py's roofline leg) and one diagnostic counter. */
#ifndef DENSEREG_PROFILE_H_
#define DENSEREG_PROFILE_H_
#include "densereg.h"
#ifdef __cplusplus
extern "C" {
#endif

/* Per-kernel timing with HIP events on the caller's stream (bench.py roofline leg).  While enabled,
 * every op of the executors is bracketed by two events; dr_profile_read synchronises, aggregates by
 * kernel, returns one row per kernel that ran, and resets.  flops/bytes are the ALGORITHMIC counts
 * (SURVEY 8d), not measured traffic. */
typedef struct dr_kernel_stat {
    char name[64];
    int64_t launches;
    double total_ms;
    double flops;
    double bytes;
} dr_kernel_stat;
int dr_profile_enable(dr_handle* h, int on);
int dr_profile_read(dr_handle* h, dr_kernel_stat* out, int max_out, int* n_out);
/* Same records, one row per (kernel, conv layer); call BEFORE dr_profile_read (which resets). */
int dr_profile_detail(dr_handle* h, dr_kernel_stat* out, int max_out, int* n_out);

/* Training handles: how many look-back waits of the BatchReNorm apply kernels expired so far (train_kernels.h: the wait is
 * bounded so that a scheduling surprise can never hang the device; it must stay 0).  Synchronises the device. */
int dr_lookback_expired(dr_handle* h);

#ifdef __cplusplus
}
#endif
#endif
