// device.h — interface between the host engine (engine.cpp) and the HIP side (kernels.hip).
#pragma once
#include "../../include/hipsoxr.h"
#include "plan.h"

namespace hipsoxr {

// Build (once per precision) the device tables of a plan.  prec: 0 = f32 engine, 1 = f64 engine.
const char *device_bank_ensure(Plan *p, int prec);
void device_bank_release(Plan *p);

// Engine precision used for an element type: f32 for f32/i16, f64 for f64/i32.
inline int engine_prec(int elem) { return (elem == HIPSOXR_F32 || elem == HIPSOXR_I16) ? 0 : 1; }
inline size_t elem_size(int elem)
{
    return elem == HIPSOXR_F32 ? 4 : elem == HIPSOXR_F64 ? 8 : elem == HIPSOXR_I32 ? 4 : 2;
}

// Variable-rate launches: input position of local output i is the Q64.64 fixed-point quadratic
//   t(i) = T0 + i*S0 + D*i(i-1)/2     (128-bit two's-complement words, hi:lo)
struct VrPos {
    uint64_t t_hi, t_lo, s_hi, s_lo, d_hi, d_lo;
};

// Enqueue one job (validated by the caller) on `stream`.  vr != nullptr: positions come from *vr
// instead of the plan's rational ratio (interpolated-phase plans only).
const char *launch_job(Plan *p, const hipsoxr_job_t &job, void *stream, const VrPos *vr = nullptr);

int device_count();

// measurement helper: mode 0 = copy src -> dst, 1 = read src only (dst: >= 4 bytes)
const char *stream_kernel(void *dst, const void *src, size_t bytes, int mode, void *stream);

// frequency-domain engine (fft.hip): whole-signal float32 jobs
bool fft_job_eligible(const Plan &p, const hipsoxr_job_t &job);
const char *launch_fft(Plan *p, const hipsoxr_job_t &job, void *stream, bool *handled);
void fft_release(const Plan *p);

} // namespace hipsoxr
