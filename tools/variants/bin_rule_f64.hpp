// bin_rule_f64.hpp — A/B BUILDS ONLY (-DFIVEEQ_BIN_RULE_F64, reported by fiveeq_build_flags()): the bin of an fp32 value by
// the fp64 formula rounds 2 and 3 used — convert, subtract, multiply, clamp, truncate, ~10 quarter-rate instructions per lane —
// in place of the product's one fp32 FMA.  Included by fiveeq_device.hpp in place of hist_bin / hist_bin2 for HistRule<float>.
__device__ __forceinline__ unsigned int hist_bin(const HistRule<float> r, const float v) {
    const double pos = ((double)v - r.lo64) * r.inv_w64;
    const unsigned int b = (unsigned int)(int)fmin(fmax(pos, 0.0), (double)r.top);
    return v == v ? b : (unsigned int)BIN_NAN;
}
__device__ __forceinline__ unsigned int hist_bin2(const HistRule<float> r, const float2v v) {
    return hist_bin(r, v.x) | (hist_bin(r, v.y) << 16);
}
