/* sbr_numerics.h — the scalar arithmetic contract of the MI355X sequence-recommender engine.
 *
 * Everything that decides a *float bit* or an *index* on the hot path is defined here, once,
 * as plain scalar C that compiles unchanged for
 *   - the gfx950 device code (hipcc, -ffp-contract=off),
 *   - the C++ host side of libsbr_hip.so,
 *   - the C CPU oracle under oracle/ (gcc, -ffp-contract=off -mfma).
 * Only IEEE-754 correctly rounded operations are used (+, *, /, sqrt, fma), every fused
 * multiply-add is written explicitly as sbr_fma(), and no libm transcendental is called, so the
 * three builds produce bit-identical results.  The HIP kernels reproduce the *association
 * orders* documented next to each reduction below; that is what makes "bit-exact negatives,
 * ranks and parameters" a testable claim (tests/test_parity_gpu.py).
 *
 * Reference semantics being restated (sbr-rs, /root/reference):
 *   predict_single = bias + dot          src/models/lstm.rs:338-350, src/models/ewma.rs:353-365
 *   WARP negative search (<=5 tries)     src/models/sequence_model.rs:47-68
 *   hinge / BPR losses                   src/models/lstm.rs:313-320, src/models/ewma.rs:328-335
 *   EWMA recurrence                      src/models/ewma.rs:302-313
 *   LSTM cell (wyrm::nn::lstm, source absent; classic cell as recalled in SURVEY.md App. B)
 *   Adagrad (wyrm::optim::Adagrad, source absent)
 * The sigmoid/tanh/exp below are this engine's own polynomial kernels (the reference uses
 * wyrm's "fast-math" approximations, which are not available here and not IEEE-exact either).
 */
#ifndef SBR_NUMERICS_H
#define SBR_NUMERICS_H

#include <stdint.h>

#if defined(__HIPCC__)
#define SBR_HD __host__ __device__ __forceinline__
#else
#define SBR_HD static inline
#endif

/* ---- fixed constants of the contract ------------------------------------------------------ */
#define SBR_WARP_MAX_TRIES 5            /* sequence_model.rs:58 */
#define SBR_ADAGRAD_EPS 1e-10f          /* wyrm Adagrad eps (recalled) */
#define SBR_DW_CHUNK_ROWS 1024
/* Per-row reduction of sparse gradient entries: a row's entries (sorted by packed row, kind) are cut into
 * chunks of SBR_SEG_CHUNK counted from the row's first entry; a chunk partial is the in-order sum of its
 * entries (the first one initialises), the row total the in-order sum of the chunk partials.  A row with
 * at most SBR_SEG_CHUNK entries is therefore a plain in-order sum; the hot rows of a skewed catalogue get
 * chunk-level parallelism with the same bits everywhere. */
#define SBR_SEG_CHUNK 256          /* split-K chunk (rows of the packed minibatch) for dense grads */
#define SBR_F32_MIN (-3.40282347e+38f)  /* Rust std::f32::MIN, evaluation.rs:31 */

/* ---- primitive helpers ---------------------------------------------------------------------- */
SBR_HD float sbr_fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

SBR_HD float sbr_bits_to_float(uint32_t u) {
    union { uint32_t u; float f; } v;
    v.u = u;
    return v.f;
}
SBR_HD uint32_t sbr_float_to_bits(float f) {
    union { uint32_t u; float f; } v;
    v.f = f;
    return v.u;
}

/* exp(x), |rel err| ~ 1e-7 on [-87, 88]; input clamped to that range.  Cephes-style:
 * n = rne(x*log2 e); r = x - n*ln2 (two-term Cody-Waite); degree-5 minimax in r; scale by 2^n
 * through the exponent field.  The rne() is the 1.5*2^23 magic-number add (exact in fp32). */
SBR_HD float sbr_expf(float x) {
    if (x > 88.0f) x = 88.0f;
    if (x < -87.0f) x = -87.0f;
    float t = x * 1.44269504088896341f;
    float n = (t + 12582912.0f) - 12582912.0f;
    float r = sbr_fma(n, -0.693359375f, x);
    r = sbr_fma(n, 2.12194440e-4f, r);
    float p = 1.9875691500e-4f;
    p = sbr_fma(p, r, 1.3981999507e-3f);
    p = sbr_fma(p, r, 8.3334519073e-3f);
    p = sbr_fma(p, r, 4.1665795894e-2f);
    p = sbr_fma(p, r, 1.6666665459e-1f);
    p = sbr_fma(p, r, 5.0000001201e-1f);
    float y = sbr_fma(p, r * r, r);
    y = y + 1.0f;
    int32_t e = (int32_t)n + 127; /* in [1, 254] because of the clamp */
    return y * sbr_bits_to_float((uint32_t)e << 23);
}

SBR_HD float sbr_sigmoidf(float x) { return 1.0f / (1.0f + sbr_expf(-x)); }

SBR_HD float sbr_tanhf(float x) {
    float ax = x < 0.0f ? -x : x;
    if (ax < 0.625f) {
        float z = x * x;
        float p = -5.70498872745e-3f;
        p = sbr_fma(p, z, 2.06390887954e-2f);
        p = sbr_fma(p, z, -5.37397155531e-2f);
        p = sbr_fma(p, z, 1.33314422036e-1f);
        p = sbr_fma(p, z, -3.33332819422e-1f);
        return sbr_fma(p * z, x, x);
    }
    float e = sbr_expf(ax + ax);
    float r = 1.0f - 2.0f / (e + 1.0f);
    return x < 0.0f ? -r : r;
}

/* ---- dot products --------------------------------------------------------------------------- */
/* Training-time score dot ("tree" order).  d = 4*L, L a power of two <= 64: lane l owns elements
 * 4l..4l+3 (one 16-byte load on the GPU), forms the partial  p_l = fma(x3,y3,fma(x2,y2,fma(x1,y1,
 * x0*y0))), then the L partials are combined by an xor-butterfly  p_l += p_{l^off}, off = L/2..1
 * (IEEE addition is commutative, so every lane ends with the same bits).  This is the order the
 * wave-level reduction of the gather/WARP-score kernel produces. */
SBR_HD float sbr_dot4_partial(const float* x, const float* y) {
    float p = x[0] * y[0];
    p = sbr_fma(x[1], y[1], p);
    p = sbr_fma(x[2], y[2], p);
    p = sbr_fma(x[3], y[3], p);
    return p;
}
#if !defined(__HIP_DEVICE_COMPILE__)
static inline float sbr_dot_tree(const float* x, const float* y, int d) {
    float p[64], q[64];
    int L = d / 4;
    for (int l = 0; l < L; ++l) p[l] = sbr_dot4_partial(x + 4 * l, y + 4 * l);
    for (int off = L / 2; off >= 1; off /= 2) {
        for (int l = 0; l < L; ++l) q[l] = p[l] + p[l ^ off];
        for (int l = 0; l < L; ++l) p[l] = q[l];
    }
    return p[0];
}
/* Prediction-time dot ("chain" order): acc = 0; acc = fma(x_k, y_k, acc), k ascending — the
 * order an f32 MFMA accumulation over k produces (MI355X guide: v_mfma_f32_* is bit-for-bit a
 * k-ordered fmaf chain).  Used by user_representation/predict/mrr_score. */
static inline float sbr_dot_chain(const float* x, const float* y, int d) {
    float acc = 0.0f;
    for (int k = 0; k < d; ++k) acc = sbr_fma(x[k], y[k], acc);
    return acc;
}
#endif

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
#include <math.h>
#if 1 /* host-side helper */
static inline void sbr_adam_corrections(uint64_t t, float* c1, float* c2) {
    *c1 = (float)(1.0 - pow((double)SBR_ADAM_B1, (double)t));
    *c2 = (float)(1.0 - pow((double)SBR_ADAM_B2, (double)t));
}
#endif

/* ---- LSTM cell, element level ----------------------------------------------------------------- */
/* Forward for one hidden unit given the four pre-activations (i, f, g, o blocks of
 * z = [x_t ; h_{t-1}] W + b, each a k-ascending fma chain seeded with the bias). */
SBR_HD void sbr_lstm_cell_fwd(float zi, float zf, float zg, float zo, float c_prev, int coupled,
                              float* gi, float* gf, float* gg, float* go, float* c, float* h) {
    float f = sbr_sigmoidf(zf);
    float i = coupled ? 1.0f - f : sbr_sigmoidf(zi);
    float g = sbr_tanhf(zg);
    float o = sbr_sigmoidf(zo);
    float cc = sbr_fma(f, c_prev, i * g);
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

/* ---- host-side index generators (never run on the device) ----------------------------------- */
#if 1
/* Marsaglia xorshift128 — the algorithm of rand 0.5's XorShiftRng as recalled in SURVEY.md
 * App. C (seed = 16 bytes little endian, all-zero seed replaced).  The reference's exact
 * streams are unpinned (no rand source here), so this is the engine's own documented generator. */
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
/* uniform integer in [0, n), n >= 1: 64x64->128 multiply-high with rejection (unbiased) */
static inline uint64_t sbr_xs_below(sbr_xorshift* r, uint64_t n) {
    uint64_t thresh = (0 - n) % n;
    for (;;) {
        uint64_t v = sbr_xs_u64(r);
        __uint128_t m = (__uint128_t)v * (__uint128_t)n;
        if ((uint64_t)m >= thresh) return (uint64_t)(m >> 64);
    }
}
static inline double sbr_xs_unit(sbr_xorshift* r) { /* [0,1) with 53 bits */
    return (double)(sbr_xs_u64(r) >> 11) * (1.0 / 9007199254740992.0);
}
#endif

#endif /* SBR_NUMERICS_H */
