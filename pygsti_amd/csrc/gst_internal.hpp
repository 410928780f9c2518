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

}  // namespace gst
