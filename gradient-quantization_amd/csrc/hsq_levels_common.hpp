// The scalar level quantiser of probabilistic_scalar_compressor.py:12-27 as ONE device function, shared by the level
// kernel (hsq_levels.hip) and the fused levels + decode kernel (hsq_decode.hip) so that both give the same bits.
// Compile with -ffp-contract=off: sub, IEEE divide, exact * 2^n_bit, truncation are separate roundings like the reference's
// elementwise ops.
#pragma once
#include "gq_common.hpp"

namespace gq {

struct LevelQuant {
    float lb, ub, range, s, smax;
    bool flat;          // lb == ub: prob_scalar:15-16 -> all zeros
    int random_mode;    // GQ_RANDOM_OFF / GIVEN / DEVICE
    const float *r;     // GIVEN: the caller's draws, indexed like the projections
    uint64_t seed;      // DEVICE: counter-based generator

    __device__ __forceinline__ LevelQuant(float lb_, float ub_, int n_bit, int random_mode_, const float *r_, uint64_t seed_)
        : lb(lb_), ub(ub_), range(ub_ - lb_), s((float)(1 << n_bit)), smax((float)(1 << n_bit) - 1.0f),
          flat((lb_ - ub_) == 0.0f), random_mode(random_mode_), r(r_), seed(seed_) {}

    // level of projection uu = u[i]
    __device__ __forceinline__ int level(float uu, int64_t i) const {
        if (flat) return 0;
        const float q = (uu - lb) / range;
        const float x = fabsf(q) * s;
        const float c = fminf(fmaxf(x, 0.0f), smax);
        int l = (x != x) ? INT32_MIN : (int)c;   // clamp(NaN) stays NaN and NaN -> int32 is INT_MIN in the reference (x86)
        if (random_mode != GQ_RANDOM_OFF) {
            const float prob = x - (float)l;
            const float rr = (random_mode == GQ_RANDOM_GIVEN) ? r[i] : uniform01(seed, (uint64_t)i);
            l += (prob > rr) ? 1 : 0;
        }
        return l;
    }
};

}  // namespace gq
