// Host-side CDF quantisation: float32 PMF -> strictly increasing 16-bit integer CDF with
// frequency stealing.  Replaces compressai._CXX.pmf_to_quantized_cdf, which the reference reaches
// through BaseBottleneck.update() (sc2bench/models/layer.py:431-441) once per model.
// Integer result, bit-exact by construction: float32 scaling with round-half-away, integer
// renormalisation to 2^precision, then zero-frequency repair in index order.
#include <cmath>
#include <cstdint>
#include <vector>

#include "../../include/sc2_bottleneck.h"

void sc2_set_error(const char *fmt, ...);

extern "C" int sc2_pmf_to_quantized_cdf(const float *pmf, int n, int precision, uint32_t *cdf) {
    if (!pmf || !cdf || n <= 0 || precision <= 0 || precision > 16) {
        sc2_set_error("pmf_to_quantized_cdf: bad argument (n=%d precision=%d)", n, precision);
        return SC2_ERR_INVALID_ARG;
    }
    for (int i = 0; i < n; ++i) {
        if (pmf[i] < 0.0f || !std::isfinite(pmf[i])) {
            sc2_set_error("Invalid `pmf`, non-finite or negative element found: %g", (double)pmf[i]);
            return SC2_ERR_DOMAIN;
        }
    }
    const float scale = static_cast<float>(1 << precision);
    cdf[0] = 0;
    int total = 0;
    for (int i = 0; i < n; ++i) {
        const uint32_t f = static_cast<uint32_t>(std::round(pmf[i] * scale));
        cdf[i + 1] = f;
        total += static_cast<int>(f);
    }
    if (total == 0) {
        sc2_set_error("Invalid `pmf`: at least one element must have a non-zero probability.");
        return SC2_ERR_ZERO_PMF;
    }
    const uint64_t one = static_cast<uint64_t>(1) << precision;
    const uint32_t utotal = static_cast<uint32_t>(total);
    uint32_t run = 0;
    for (int i = 0; i <= n; ++i) {
        run += static_cast<uint32_t>((one * cdf[i]) / utotal);
        cdf[i] = run;
    }
    cdf[n] = static_cast<uint32_t>(one);
    for (int i = 0; i < n; ++i) {
        if (cdf[i] != cdf[i + 1]) continue;
        // symbol i has zero frequency: steal one count from the least frequent symbol that can spare it
        uint32_t best_freq = ~0u;
        int best = -1;
        for (int j = 0; j < n; ++j) {
            const uint32_t f = cdf[j + 1] - cdf[j];
            if (f > 1 && f < best_freq) {
                best_freq = f;
                best = j;
            }
        }
        if (best < 0) {
            sc2_set_error("pmf_to_quantized_cdf: no symbol can spare a count (n=%d too large for precision %d)", n,
                          precision);
            return SC2_ERR_INTERNAL;
        }
        if (best < i) {
            for (int j = best + 1; j <= i; ++j) --cdf[j];
        } else {
            for (int j = i + 1; j <= best; ++j) ++cdf[j];
        }
    }
    return SC2_OK;
}
