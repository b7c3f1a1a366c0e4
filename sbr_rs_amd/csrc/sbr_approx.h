/* sbr_approx.h — the approximation polynomials of the numerics contract.
 *
 * This is the ONLY product header the CPU oracle (oracle/) may include: it holds nothing but the
 * fixed-coefficient approximation of the transcendental the path needs, written with correctly
 * rounded IEEE-754 operations only (+, *, fma, min, max), so that gfx950 device code, the host and
 * the oracle produce the same bits.  Everything built on top of it — sigmoid, the LSTM cell, the
 * losses, the optimisers, the dot orders, the index generators — is stated twice, once in the
 * product (sbr_numerics.h) and once, independently, in the oracle's own numerics header.
 *
 * tanh(x) ~= P(x) / Q(x) on |x| <= 7.90531110763549805 (beyond that tanh rounds to +-1 in f32):
 *   P(x) = x (a1 + a3 x^2 + ... + a13 x^12),   Q(x) = b0 + b2 x^2 + b4 x^4 + b6 x^6
 * (odd 13 / even 6 rational minimax form widely used for single-precision tanh).  Measured against
 * float64 libm over 6.2e6 points: |P/Q - tanh| <= 2.9e-7 absolute and relative
 * (tests/test_oracle.py::test_activation_accuracy, tests/test_numerics_gpu.py).
 * Q(x) lies in [4.89e-3, 0.903], so products of up to four Q values stay far inside the f32 range
 * — the LSTM cell uses that to share one division between its four gates.
 *
 * The reference's activations are wyrm's "fast-math" approximations (Cargo.toml:29), themselves
 * not IEEE-exact; no reference test pins an activation value.
 */
#ifndef SBR_APPROX_H
#define SBR_APPROX_H

#if defined(__HIPCC__)
#define SBR_APPROX_HD __host__ __device__ __forceinline__
#else
#define SBR_APPROX_HD static inline
#endif

#define SBR_TANH_CLAMP 7.90531110763549805f

/* numerator and denominator of the rational tanh; the caller divides (or multiplies by a
 * reciprocal it obtained elsewhere) */
SBR_APPROX_HD void sbr_tanh_pq(float x, float* p, float* q) {
    /* comparison + select, not fmin / fmax: a NaN argument fails both comparisons and stays NaN, as the reference's
     * activations propagate it (fmax(NaN, -C) would be -C: diverged weights would give finite-looking states and scores) */
    x = x > SBR_TANH_CLAMP ? SBR_TANH_CLAMP : x;
    x = x < -SBR_TANH_CLAMP ? -SBR_TANH_CLAMP : x;
    const float x2 = x * x;
    float n = -2.76076847742355e-16f;
    n = __builtin_fmaf(n, x2, 2.00018790482477e-13f);
    n = __builtin_fmaf(n, x2, -8.60467152213735e-11f);
    n = __builtin_fmaf(n, x2, 5.12229709037114e-08f);
    n = __builtin_fmaf(n, x2, 1.48572235717979e-05f);
    n = __builtin_fmaf(n, x2, 6.37261928875436e-04f);
    n = __builtin_fmaf(n, x2, 4.89352455891786e-03f);
    *p = n * x;
    float dq = 1.19825839466702e-06f;
    dq = __builtin_fmaf(dq, x2, 1.18534705686654e-04f);
    dq = __builtin_fmaf(dq, x2, 2.26843463243900e-03f);
    dq = __builtin_fmaf(dq, x2, 4.89352518554385e-03f);
    *q = dq;
}

#endif /* SBR_APPROX_H */
