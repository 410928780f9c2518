// gst_internal.hpp -- the few things the translation units of libgstfwd share besides the kernel launchers.
#pragma once
#include <hip/hip_runtime.h>
#include <string>

struct gst_plan;

namespace gst {

// Record the thread-local message gst_last_error() returns; returns `code` (use as `return set_error(...)`).
int set_error(int code, const std::string& msg);

// Bring `plan` onto its device if needed (GST_OK or a GST_E* code), then its stream / device ordinal.
int plan_ensure_device(gst_plan* plan);
hipStream_t plan_stream(const gst_plan* plan);
int plan_device(const gst_plan* plan);

// out[i] = sum_{r < n_slots} slots[r * stride + i] in ascending r (deterministic), i < n   (gst_kernels_normal.hip)
hipError_t launch_sum_slots(const double* slots, int n_slots, int64_t stride, int64_t n, double* out, hipStream_t s);

// clears *flag when any w[i], i < n, is not finite   (gst_kernels_normal.hip)
hipError_t launch_check_finite(const double* w, int64_t n, uint32_t* flag, hipStream_t s);

// Destination tracking (gst_track.cpp): device ranges that still hold the structural zeros of an analytic Jacobian.
void track_alloc(const void* p, size_t bytes);           // memory the library handed out or owns
void track_free(const void* p);                          // ... released (drops the claims inside it)
bool track_owned(const void* p, size_t bytes);           // [p, p + bytes) lies inside one tracked allocation
void track_touch(const void* p, size_t bytes);           // the library writes there (anything but a row scaling): claims dropped
uint32_t* track_claim_find(const void* base, size_t bytes, uint64_t sig);       // the claim's device word (1 = holds), or NULL
uint32_t* track_claim_set(const void* base, size_t bytes, uint64_t sig, int device, uint64_t owner);   // owner = the claiming plan's uid
uint32_t* track_claim_overlapping(const void* p, size_t bytes, bool* several, uint64_t* owner);  // the one claim a row scaling of this range meets

}  // namespace gst
