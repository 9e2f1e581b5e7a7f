// plan.h — immutable conversion plan: rational ratio, polyphase bank, device tables.
//
// A plan is what libsoxr's soxr_create designs per handle (reference call sites
// src/soxr_ext.cpp:76-78, :230-232, :305-307), factored out so that any number of streams, batch
// jobs and ranks share one bank.
#pragma once
#include <cstdint>
#include <mutex>
#include <string>
#include <vector>

namespace hipsoxr {

struct QualitySpec {
    double bits;           // 0 (QQ cubic), 16, 20, 28
    double passband_end;   // fraction of the lower rate's Nyquist
    double stopband_begin; // ditto
};

// Device-resident tables for one engine precision (float or double).
struct DeviceBank {
    void *tap_major = nullptr;   // [T][Lpad]   (gather kernel: lanes read neighbouring phases)
    void *phase_major = nullptr; // [L][T] (k_wave_dot only; uploaded on first use)
    void *interp_tab = nullptr;  // interpolated-phase plans: [P][T] of Real4 (a0..a3)
    int64_t Lpad = 0;
    // tile kernel: per output-tile skewed, zero-padded half tables (see kernels.hip)
    void *tile_tab = nullptr;    // [n_rt][2][I_h][RT]
    int32_t RT = 0, n_rt = 0, I_h = 0;
    void *tile_tab_m = nullptr;  // f32 engine: tables in the k_tile_mfma geometry
    int32_t *tile_i0_m = nullptr;
    int32_t *tile_i0 = nullptr;  // [n_rt][2] device: first input offset (rel. to period start) of each half
    bool ready = false;
};

struct Plan {
    double in_rate = 0, out_rate = 0;
    unsigned long recipe = 0;
    QualitySpec q{};
    int64_t L = 1, M = 1; // out/in = L/M
    int32_t T = 8;        // taps per phase
    double att_db = 0, beta = 0;
    // phases == 0: exact rational bank [L][T].  phases == P > 0: interpolated-phase plan, bank holds
    // the cubic coefficient table [P][T][4] (see plan.cpp, "interpolated-phase plans").
    int32_t phases = 0;
    std::vector<double> bank; // float64
    double proto_scale = 0; // interpolated-phase plans: DC normalisation of the continuous-time prototype (plan_proto)
    // A bank installed from outside (hipsoxr_plan_set_bank / hipsoxr_plan_broadcast) that DIFFERS from the designed one:
    // engines that derive their tables from the analytic prototype instead of `bank` (the two-stage form) decline the plan.
    bool custom_bank = false;
    // ... and what the designed bank hashed to (FNV-1a over its bytes, taken when the first differing bank is installed), so that
    // installing the designed bank AGAIN puts the plan back on every engine (round 6: the flag used to be one-way).
    uint64_t designed_hash = 0;
    bool have_designed_hash = false;
    // device side (lazily built on first use, per precision: 0 = f32, 1 = f64)
    DeviceBank dev[2];
    std::mutex mu;
    int device = -1;
    struct TwoStage *two = nullptr; // arbitrary ratios, 1e-6-class float device jobs: FFT stage + short polyphase stage (twostage.hip)

    ~Plan();
};
// The prototype of an interpolated-phase plan as a function of continuous time tau (in INPUT samples), unit DC gain:
// output k at input position t weights input sample n with plan_proto(p, t - n).  (Windowed-sinc recipes only.)
double plan_proto(const Plan &p, double tau);
double bessel_i0(double x);
void twostage_release(Plan *p);

// Returns nullptr on success, else a static error string.
const char *quality_spec(unsigned long recipe, QualitySpec *q);
const char *reduce_ratio(double in_rate, double out_rate, int64_t *L, int64_t *M);
// force_interp: always build the interpolated-phase table (variable-rate streams need one whatever
// the ratio: their positions are not tied to L/M).
const char *plan_design(double in_rate, double out_rate, unsigned long recipe, Plan *p, bool force_interp = false);
uint64_t plan_out_len(const Plan &p, uint64_t n_in);

// Position of output k: first tap's absolute input index n0 and phase p.
inline void locate(const Plan &pl, int64_t k, int64_t *n0, int64_t *p)
{
    __int128 kM = (__int128)k * pl.M;
    *n0 = (int64_t)(kM / pl.L) - (pl.T / 2 - 1);
    *p = (int64_t)(kM % pl.L);
}

} // namespace hipsoxr
