/* sbr_numerics.h — the scalar arithmetic contract of the MI355X sequence-recommender engine
 * (product side: gfx950 device code and the C++ host side of libsbr_hip.so).
 *
 * Everything that decides a *float bit* or an *index* on the hot path is defined here as plain
 * scalar code built from IEEE-754 correctly rounded operations only (+, *, /, sqrt, fma, min, max);
 * every fused multiply-add is written explicitly as sbr_fma() and the build uses -ffp-contract=off,
 * so host and device produce the same bits.  The CPU oracle does NOT include this header: it
 * restates the same formulas independently (the numerics header under oracle/) and shares only the
 * approximation polynomial of sbr_approx.h — tests/test_abi.py enforces that.
 *
 * Reference semantics being restated (sbr-rs, /root/reference):
 *   predict_single = bias + dot          src/models/lstm.rs:338-350, src/models/ewma.rs:353-365
 *   WARP negative search (<=5 tries)     src/models/sequence_model.rs:47-68
 *   hinge / BPR losses                   src/models/lstm.rs:313-320, src/models/ewma.rs:328-335
 *   EWMA recurrence                      src/models/ewma.rs:302-313
 *   LSTM cell (wyrm::nn::lstm, source absent; classic cell as recalled in SURVEY.md App. B)
 *   Adagrad / Adam (wyrm::optim, source absent)
 *   rand 0.5 index generators (source absent): XorShiftRng, gen::<[u8; 16]>,
 *   gen_range / shuffle, Uniform, Normal (ziggurat)
 */
#ifndef SBR_NUMERICS_H
#define SBR_NUMERICS_H

#include <math.h>
#include <stdint.h>

#include "sbr_approx.h"

#if defined(__HIPCC__)
#define SBR_HD __host__ __device__ __forceinline__
#else
#define SBR_HD static inline
#endif

/* ---- fixed constants of the contract ------------------------------------------------------ */
#define SBR_WARP_MAX_TRIES 5            /* sequence_model.rs:58 */
#define SBR_ADAGRAD_EPS 1e-10f          /* wyrm Adagrad eps (recalled) */
#ifndef SBR_DW_CHUNK_ROWS               /* (overridable for TIMING experiments only: another value is another contract) */
#define SBR_DW_CHUNK_ROWS 1024          /* split-K chunk (rows of the packed minibatch) for dense grads */
#endif
/* Per-row reduction of sparse gradient entries: a row's entries (sorted by packed row, kind) are cut into
 * chunks of SBR_SEG_CHUNK counted from the row's first entry; a chunk partial is the in-order sum of its
 * entries (the first one initialises), the row total the in-order sum of the chunk partials.  A row with
 * at most SBR_SEG_CHUNK entries is therefore a plain in-order sum; the hot rows of a skewed catalogue get
 * chunk-level parallelism with the same bits everywhere. */
#define SBR_SEG_CHUNK 256
#define SBR_F32_MIN (-3.40282347e+38f)  /* Rust std::f32::MIN, evaluation.rs:31 */

/* ---- primitive helpers ---------------------------------------------------------------------- */
SBR_HD float sbr_fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

/* ---- activations ------------------------------------------------------------------------------ */
/* tanh = P/Q (sbr_approx.h), one IEEE division; sigmoid(x) = 1/2 + 1/2 tanh(x/2) */
SBR_HD float sbr_tanhf(float x) {
    float p, q;
    sbr_tanh_pq(x, &p, &q);
    return p / q;
}
SBR_HD float sbr_sigmoidf(float x) { return sbr_fma(0.5f, sbr_tanhf(0.5f * x), 0.5f); }

/* ---- dot products --------------------------------------------------------------------------- */
/* Training-time score dot ("tree" order).  d = 4*L, L a power of two <= 64: lane l owns elements
 * 4l..4l+3 (one 16-byte load on the GPU), forms the partial  p_l = fma(x3,y3,fma(x2,y2,fma(x1,y1,
 * x0*y0))), then the L partials are combined by an xor-butterfly  p_l += p_{l^off}, off = L/2..1
 * (IEEE addition is commutative, so every lane ends with the same bits).  Prediction-time dots
 * (user_representation / predict / mrr_score) are k-ascending fma chains from 0 — the order an f32
 * MFMA accumulation produces. */
SBR_HD float sbr_dot4_partial(const float* x, const float* y) {
    float p = x[0] * y[0];
    p = sbr_fma(x[1], y[1], p);
    p = sbr_fma(x[2], y[2], p);
    p = sbr_fma(x[3], y[3], p);
    return p;
}

/* ---- counter-based negative sampling -------------------------------------------------------- */
/* The reference draws negatives from a sequential per-thread xorshift stream with a
 * data-dependent trip count (sequence_model.rs:58-65, :137).  That cannot be evaluated in
 * parallel, so the engine keys every draw by (fit seed, global epoch, position of the
 * subsequence in the epoch order, step t, try): draw = mix(key) mapped to [0, num_items) by a
 * 32x32->64 multiply-high.  No rejection of the positive or of seen items, and id 0 is drawable,
 * exactly as in the reference (sequence_model.rs:74). */
SBR_HD uint64_t sbr_mix64(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
SBR_HD uint64_t sbr_epoch_key(uint64_t fit_seed, uint64_t global_epoch) {
    return sbr_mix64(fit_seed ^ sbr_mix64(global_epoch * 0x9E3779B97F4A7C15ULL + 1ULL));
}
/* ctr = p * max_sequence_length + t  (p = position of the subsequence in this epoch's order) */
SBR_HD uint32_t sbr_neg_draw(uint64_t epoch_key, uint32_t ctr, uint32_t try_idx, uint32_t num_items) {
    uint64_t x = sbr_mix64(epoch_key + (((uint64_t)ctr << 3) | (uint64_t)try_idx) * 0x9E3779B97F4A7C15ULL);
    return (uint32_t)(((x >> 32) * (uint64_t)num_items) >> 32);
}

/* ---- losses (lstm.rs:316-319) ----------------------------------------------------------------- */
/* pos/neg are "dot + bias".  Returns the loss term and the coefficient g = dloss/dneg
 * (= -dloss/dpos).  Hinge/WARP: relu((1 + neg) - pos), association as in the reference graph. */
SBR_HD float sbr_loss_hinge(float pos, float neg, float* g) {
    float m = (1.0f + neg) - pos;
    *g = m > 0.0f ? 1.0f : 0.0f;
    return m > 0.0f ? m : 0.0f;
}
SBR_HD float sbr_loss_bpr(float pos, float neg, float* g) {
    float s = sbr_sigmoidf(neg - pos);
    *g = s * (1.0f - s);
    return s;
}
/* WARP acceptance test, association as in sequence_model.rs:62 */
SBR_HD int sbr_warp_violates(float pos, float neg) { return (1.0f - pos) + neg > 0.0f; }

/* ---- Adagrad element update ------------------------------------------------------------------- */
SBR_HD void sbr_adagrad(float* w, float* G, float g, float lr, float l2) {
    float g2 = sbr_fma(l2, *w, g);
    float acc = sbr_fma(g2, g2, *G);
    *G = acc;
    float step = lr / (SBR_ADAGRAD_EPS + __builtin_sqrtf(acc));
    *w = sbr_fma(-step, g2, *w);
}

/* ---- Adam element update (wyrm::optim::Adam as recalled: beta1 0.9, beta2 0.999, eps 1e-8, L2 folded
 * into the gradient, bias correction by the optimiser step count).  c1 = 1 - beta1^t and
 * c2 = 1 - beta2^t are computed on the host (double pow, cast to f32) and passed in.  Sparse
 * parameters are updated lazily: only the rows a step touches move. */
#define SBR_ADAM_B1 0.9f
#define SBR_ADAM_B2 0.999f
#define SBR_ADAM_EPS 1e-8f
SBR_HD void sbr_adam(float* w, float* m1, float* v2, float g, float lr, float l2, float c1, float c2) {
    float g2 = sbr_fma(l2, *w, g);
    float mm = sbr_fma(SBR_ADAM_B1, *m1, (1.0f - SBR_ADAM_B1) * g2);
    float vv = sbr_fma(SBR_ADAM_B2, *v2, (1.0f - SBR_ADAM_B2) * (g2 * g2));
    *m1 = mm;
    *v2 = vv;
    float mhat = mm / c1;
    float vhat = vv / c2;
    float step = lr / (__builtin_sqrtf(vhat) + SBR_ADAM_EPS);
    *w = sbr_fma(-step, mhat, *w);
}
static inline void sbr_adam_corrections(uint64_t t, float* c1, float* c2) {
    *c1 = (float)(1.0 - pow((double)SBR_ADAM_B1, (double)t));
    *c2 = (float)(1.0 - pow((double)SBR_ADAM_B2, (double)t));
}

/* ---- LSTM cell, element level ----------------------------------------------------------------- */
/* Forward for one hidden unit given the four pre-activations (i, f, g, o blocks of
 * z = [x_t ; h_{t-1}] W + b, each a k-ascending fma chain seeded with the bias):
 *   i = sig(zi), f = sig(zf), g = tanh(zg), o = sig(zo)   (coupled: i = 1 - f)
 *   c = f c_prev + i g,   h = o tanh(c)
 * The four gate activations are rational (P/Q) and share ONE division: with the denominators
 * qi, qf, qg, qo,  r = 1 / ((qi qf)(qg qo))  and  1/qi = (r (qg qo)) qf  etc. — on the GPU the
 * cell epilogue is VALU time that f32 MFMAs cannot hide, and an IEEE division is ~10 instructions.
 * Coupled cells have no input gate: qi = 1 exactly, and the same expressions apply. */
SBR_HD void sbr_lstm_cell_fwd(float zi, float zf, float zg, float zo, float c_prev, int coupled,
                              float* gi, float* gf, float* gg, float* go, float* c, float* h) {
    float pi = 0.0f, qi = 1.0f, pf, qf, pg, qg, po, qo;
    if (!coupled) sbr_tanh_pq(0.5f * zi, &pi, &qi);
    sbr_tanh_pq(0.5f * zf, &pf, &qf);
    sbr_tanh_pq(zg, &pg, &qg);
    sbr_tanh_pq(0.5f * zo, &po, &qo);
    const float q_if = qi * qf, q_go = qg * qo;
    const float r = 1.0f / (q_if * q_go);
    const float r_if = r * q_go, r_go = r * q_if;
    const float f = sbr_fma(0.5f, pf * (r_if * qi), 0.5f);
    const float i = coupled ? 1.0f - f : sbr_fma(0.5f, pi * (r_if * qf), 0.5f);
    const float g = pg * (r_go * qo);
    const float o = sbr_fma(0.5f, po * (r_go * qg), 0.5f);
    const float cc = sbr_fma(f, c_prev, i * g);
    *gi = i; *gf = f; *gg = g; *go = o; *c = cc;
    *h = o * sbr_tanhf(cc);
}
/* Backward for one hidden unit.  dh = total gradient w.r.t. h_t, dc_in = carry from step t+1
 * (0 at the last step).  Writes pre-activation grads and the carry for step t-1. */
SBR_HD void sbr_lstm_cell_bwd(float dh, float dc_in, float i, float f, float g, float o, float c,
                              float c_prev, int coupled, float* dzi, float* dzf, float* dzg,
                              float* dzo, float* dc_out) {
    float tc = sbr_tanhf(c);
    float d_o = dh * tc;
    float dc = sbr_fma(dh * o, 1.0f - tc * tc, dc_in);
    float di = dc * g;
    float dg = dc * i;
    float df = dc * c_prev;
    *dc_out = dc * f;
    if (coupled) {
        df = df - di; /* i = 1 - f */
        *dzi = 0.0f;
    } else {
        *dzi = di * (i * (1.0f - i));
    }
    *dzf = df * (f * (1.0f - f));
    *dzg = dg * (1.0f - g * g);
    *dzo = d_o * (o * (1.0f - o));
}

/* ---- host-side index generators: rand 0.5 as recalled ---------------------
 * The reference draws every index stream from rand 0.5 (Cargo.toml:19), whose source is not in this
 * image and which no reference test pins; the algorithms below are the crate's as recalled:
 *   XorShiftRng           Marsaglia xorshift128; seed = 16 bytes little endian, all-zero replaced;
 *                         next_u64 = low word first
 *   gen::<[u8; 16]>()     sixteen next_u32() calls, each truncated to its low byte
 *                         (XorShiftRng::from_seed(parameters.rng().gen()), sequence_model.rs:97)
 *   gen_range / shuffle   UniformInt::sample_single: zone = range << leading_zeros(range), draw
 *                         v = next_u64, accept when the low half of v*range is <= zone, result = high
 *                         half; shuffle = Fisher-Yates from the end (sequence_model.rs:84,109)
 *   Uniform::new(lo, hi)  zone = MAX - (MAX - range + 1) % range, same widening-multiply test
 *                         (user_based_split keys, data.rs:77-78)
 *   Normal                ziggurat, 256 layers (embedding_init lstm.rs:22-25; LSTM weights)
 * Host functions: never run on the device. */
#include "sbr_ziggurat_tables.h"
typedef struct { uint32_t x, y, z, w; } sbr_xorshift;
static inline void sbr_xs_seed(sbr_xorshift* r, const uint8_t seed[16]) {
    uint32_t s[4];
    for (int i = 0; i < 4; ++i)
        s[i] = (uint32_t)seed[4 * i] | ((uint32_t)seed[4 * i + 1] << 8) |
               ((uint32_t)seed[4 * i + 2] << 16) | ((uint32_t)seed[4 * i + 3] << 24);
    if ((s[0] | s[1] | s[2] | s[3]) == 0) {
        s[0] = 0x193a6754u; s[1] = 0xa8a7d469u; s[2] = 0x97830e05u; s[3] = 0x113ba7bbu;
    }
    r->x = s[0]; r->y = s[1]; r->z = s[2]; r->w = s[3];
}
static inline uint32_t sbr_xs_u32(sbr_xorshift* r) {
    uint32_t t = r->x ^ (r->x << 11);
    r->x = r->y; r->y = r->z; r->z = r->w;
    r->w = r->w ^ (r->w >> 19) ^ (t ^ (t >> 8));
    return r->w;
}
static inline uint64_t sbr_xs_u64(sbr_xorshift* r) {
    uint64_t lo = sbr_xs_u32(r);
    uint64_t hi = sbr_xs_u32(r);
    return lo | (hi << 32);
}
static inline void sbr_rand_gen_seed16(sbr_xorshift* r, uint8_t out[16]) {
    for (int i = 0; i < 16; ++i) out[i] = (uint8_t)sbr_xs_u32(r);
}
/* gen_range(0, n), n >= 1 (the crate asserts low < high; n = 0 returns 0 here without drawing instead of shifting by 64) */
static inline uint64_t sbr_rand_gen_range(sbr_xorshift* r, uint64_t n) {
    if (n == 0) return 0;
    const uint64_t zone = n << __builtin_clzll(n);
    for (;;) {
        const __uint128_t m = (__uint128_t)sbr_xs_u64(r) * (__uint128_t)n;
        if ((uint64_t)m <= zone) return (uint64_t)(m >> 64);
    }
}
/* Uniform::new(lo, hi).sample(), lo < hi */
static inline uint64_t sbr_rand_uniform(sbr_xorshift* r, uint64_t lo, uint64_t hi) {
    const uint64_t range = hi - lo; /* new_inclusive(lo, hi - 1): (hi - 1) - lo + 1 */
    const uint64_t zone = UINT64_MAX - (UINT64_MAX - range + 1) % range;
    for (;;) {
        const __uint128_t m = (__uint128_t)sbr_xs_u64(r) * (__uint128_t)range;
        if ((uint64_t)m <= zone) return lo + (uint64_t)(m >> 64);
    }
}
static inline double sbr_rand_bits_to_f64(uint64_t fraction52, int exponent) {
    union { uint64_t u; double f; } v;
    v.u = ((uint64_t)(1023 + exponent) << 52) | fraction52;
    return v.f;
}
static inline double sbr_rand_open01(sbr_xorshift* r) { /* Open01: (0, 1) */
    return sbr_rand_bits_to_f64(sbr_xs_u64(r) >> 12, 0) - (1.0 - 2.220446049250313e-16 / 2.0);
}
static inline double sbr_rand_standard_f64(sbr_xorshift* r) { /* Standard: [0, 1), 53 bits */
    return (double)(sbr_xs_u64(r) >> 11) * (1.0 / 9007199254740992.0);
}
/* StandardNormal */
static inline double sbr_rand_standard_normal(sbr_xorshift* r) {
    for (;;) {
        const uint64_t bits = sbr_xs_u64(r);
        const int i = (int)(bits & 0xff);
        const double u = sbr_rand_bits_to_f64(bits >> 12, 1) - 3.0; /* [2, 4) - 3 = [-1, 1) */
        const double x = u * SBR_ZIG_NORM_X[i];
        if (fabs(x) < SBR_ZIG_NORM_X[i + 1]) return x;
        if (i == 0) { /* tail */
            double tx = 1.0, ty = 0.0;
            while (-2.0 * ty < tx * tx) {
                const double a = sbr_rand_open01(r);
                const double b = sbr_rand_open01(r);
                tx = log(a) / SBR_ZIG_NORM_R;
                ty = log(b);
            }
            return u < 0.0 ? tx - SBR_ZIG_NORM_R : SBR_ZIG_NORM_R - tx;
        }
        if (SBR_ZIG_NORM_F[i + 1] + (SBR_ZIG_NORM_F[i] - SBR_ZIG_NORM_F[i + 1]) * sbr_rand_standard_f64(r) < exp(-x * x / 2.0))
            return x;
    }
}
/* Normal::new(mean, std).sample() as f32 */
static inline float sbr_rand_normal_f32(sbr_xorshift* r, double mean, double std_dev) {
    return (float)(mean + std_dev * sbr_rand_standard_normal(r));
}

#endif /* SBR_NUMERICS_H */
