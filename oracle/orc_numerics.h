/* orc_numerics.h — the CPU oracle's OWN statement of the scalar arithmetic of the hot path
 * (test infrastructure, not the product).
 *
 * Written from the reference's text and SURVEY.md, independently of the product's
 * sbr_rs_amd/csrc/sbr_numerics.h: a wrong formula on either side shows up as a parity failure.
 * The one thing shared with the product is the fixed-coefficient approximation polynomial of the
 * transcendental (sbr_rs_amd/csrc/sbr_approx.h: numerator P and denominator Q of the rational
 * tanh) — transcendentals are not bit-portable between libm and a GPU, so both sides must evaluate
 * the same polynomial; its accuracy against float64 libm is tested on its own
 * (tests/test_oracle.py::test_activation_accuracy).  tests/test_abi.py forbids any other include
 * from sbr_rs_amd/ here.
 *
 * To be bit-comparable with the GPU the oracle commits to ONE association order per formula (the
 * order is part of the engine's documented contract), uses only correctly rounded
 * IEEE-754 operations, and is compiled with -ffp-contract=off so that a fused multiply-add happens
 * exactly where fmaf() is written.
 *
 * Parity status: wyrm ^0.9.1 and rand ^0.5 are un-vendored (Cargo.toml:19,29) and no Rust toolchain
 * exists here; the cell, the optimisers and the generators below are the published algorithms as
 * recalled in SURVEY.md App. B / C — "parity unpinned" below the level of the reference's own tests.
 */
#ifndef ORC_NUMERICS_H
#define ORC_NUMERICS_H

#include <math.h>
#include <stdint.h>

#include "../sbr_rs_amd/csrc/sbr_approx.h" /* sbr_tanh_pq: P(x), Q(x) of the rational tanh */
#include "orc_ziggurat_tables.h"

#define ORC_WARP_TRIES 5        /* `for _ in 0..5`, sequence_model.rs:58 */
#define ORC_DW_CHUNK_ROWS 1024  /* dense-gradient split: rows per chunk partial */
#define ORC_SEG_CHUNK 256       /* sparse-gradient split: entries per chunk partial */
#define ORC_EWMA_CHUNK_SEQS 256
#define ORC_F32_MIN (-3.40282347e+38f) /* std::f32::MIN, evaluation.rs:31 */

/* ---- activations: wyrm `.tanh()` / `.sigmoid()` nodes (lstm cell, ewma.rs:303, lstm.rs:317) ----- */
static inline float orc_tanh(float x) {
    float num, den;
    sbr_tanh_pq(x, &num, &den);
    return num / den;
}
/* sigmoid(x) = (1 + tanh(x/2)) / 2, evaluated as fma(1/2, tanh(x/2), 1/2) */
static inline float orc_sigmoid(float x) { return fmaf(0.5f, orc_tanh(x * 0.5f), 0.5f); }

/* ---- dot products ------------------------------------------------------------------------------ */
/* predict_single during training (lstm.rs:338-350: bias + simd_dot).  Contract order: the vector is
 * cut into quads; quad q gives  s_q = fma(x3,y3, fma(x2,y2, fma(x1,y1, x0*y0)));  the d/4 quad sums
 * are then combined pairwise, partner distance d/8, d/16, ..., 1 (s_q <- s_q + s_{q xor dist}). */
static inline float orc_dot_training(const float* x, const float* y, int d) {
    float s[64], n[64];
    const int quads = d / 4;
    for (int q = 0; q < quads; ++q) {
        const float* a = x + 4 * q;
        const float* b = y + 4 * q;
        float t = a[0] * b[0];
        t = fmaf(a[1], b[1], t);
        t = fmaf(a[2], b[2], t);
        t = fmaf(a[3], b[3], t);
        s[q] = t;
    }
    for (int dist = quads >> 1; dist > 0; dist >>= 1) {
        for (int q = 0; q < quads; ++q) n[q] = s[q] + s[q ^ dist];
        for (int q = 0; q < quads; ++q) s[q] = n[q];
    }
    return s[0];
}
/* predict at evaluation time (sequence_model.rs:213-232): plain left-to-right fma accumulation from 0 */
static inline float orc_dot_prediction(const float* x, const float* y, int d) {
    float acc = 0.0f;
    for (int k = 0; k < d; ++k) acc = fmaf(x[k], y[k], acc);
    return acc;
}

/* ---- negative draws ------------------------------------------------------------------------------
 * The reference samples `negative_item_range.sample(thread_rng)` from a sequential stream with a
 * data-dependent number of draws (sequence_model.rs:58-65,137) — not parallelisable.  The engine's
 * documented replacement: draw number `attempt` of step t of the subsequence at
 * epoch position p is a hash of (epoch key, p * max_len + t, attempt), the epoch key a hash of
 * (per-partition fit seed, global epoch); the hash is the SplitMix64 finaliser; the 32 high bits
 * are scaled to [0, num_items) by a multiply-high.  No rejection of the positive / seen items. */
static inline uint64_t orc_splitmix_finalise(uint64_t v) {
    v ^= v >> 30; v *= 0xBF58476D1CE4E5B9ULL;
    v ^= v >> 27; v *= 0x94D049BB133111EBULL;
    v ^= v >> 31;
    return v;
}
#define ORC_GOLDEN_GAMMA 0x9E3779B97F4A7C15ULL
static inline uint64_t orc_epoch_key_of(uint64_t fit_seed, uint64_t global_epoch) {
    return orc_splitmix_finalise(fit_seed ^ orc_splitmix_finalise(global_epoch * ORC_GOLDEN_GAMMA + 1));
}
static inline uint32_t orc_negative_draw(uint64_t epoch_key, uint32_t position_counter, uint32_t attempt, uint32_t num_items) {
    const uint64_t index = ((uint64_t)position_counter << 3) | attempt;
    const uint64_t h = orc_splitmix_finalise(epoch_key + index * ORC_GOLDEN_GAMMA);
    return (uint32_t)(((h >> 32) * (uint64_t)num_items) >> 32);
}

/* ---- losses: lstm.rs:313-320 / ewma.rs:328-335 ------------------------------------------------ */
/* Loss::Hinge | Loss::WARP => (1.0 + neg - pos).relu();  returns the value, *dneg = d loss / d neg */
static inline float orc_hinge(float pos, float neg, float* dneg) {
    const float margin = (1.0f + neg) - pos;
    if (margin > 0.0f) { *dneg = 1.0f; return margin; }
    *dneg = 0.0f;
    return 0.0f;
}
/* Loss::BPR => (neg - pos).sigmoid();  derivative s (1 - s) */
static inline float orc_bpr(float pos, float neg, float* dneg) {
    const float s = orc_sigmoid(neg - pos);
    *dneg = s * (1.0f - s);
    return s;
}
/* `if 1.0 - pos_prediction + neg_prediction > 0.0 { break; }`  sequence_model.rs:62 */
static inline int orc_warp_accepts(float pos, float neg) { return ((1.0f - pos) + neg) > 0.0f; }

/* ---- optimisers: wyrm::optim as recalled --------------------------------- */
/* Adagrad:  g' = g + l2 w;  G += g'^2;  w -= lr / (1e-10 + sqrt(G)) * g' */
static inline void orc_adagrad_step(float* w, float* G, float grad, float lr, float l2) {
    const float gp = fmaf(l2, *w, grad);
    const float Gn = fmaf(gp, gp, *G);
    const float rate = lr / (1e-10f + sqrtf(Gn));
    *G = Gn;
    *w = fmaf(-rate, gp, *w);
}
/* Adam: beta1 0.9, beta2 0.999, eps 1e-8, L2 folded into the gradient, bias correction by the number of
 * optimiser steps t: corr1 = 1 - 0.9^t, corr2 = 1 - 0.999^t (double pow, rounded to f32) */
static inline void orc_adam_bias_corrections(uint64_t t, float* corr1, float* corr2) {
    *corr1 = (float)(1.0 - pow((double)0.9f, (double)t));
    *corr2 = (float)(1.0 - pow((double)0.999f, (double)t));
}
static inline void orc_adam_step(float* w, float* m, float* v, float grad, float lr, float l2, float corr1, float corr2) {
    const float gp = fmaf(l2, *w, grad);
    const float mn = fmaf(0.9f, *m, (1.0f - 0.9f) * gp);
    const float vn = fmaf(0.999f, *v, (1.0f - 0.999f) * (gp * gp));
    *m = mn;
    *v = vn;
    const float rate = lr / (sqrtf(vn / corr2) + 1e-8f);
    *w = fmaf(-rate, mn / corr1, *w);
}

/* ---- LSTM cell: wyrm nn::lstm as recalled ----------------------------------
 *   forget f = sig(z_f), update gate i = sig(z_i) (coupled: i = 1 - f), update value g = tanh(z_g),
 *   output gate o = sig(z_o);  c_t = f c_{t-1} + i g;  h_t = o tanh(c_t).
 * Contract detail: the three (two) sigmoids and the tanh of the gates are rational functions
 * num/den whose four denominators are inverted with ONE division:
 *   R = 1 / ((den_i den_f)(den_g den_o)),   1/(den_i den_f) = R (den_g den_o),  1/den_i = that * den_f ...
 * (a coupled cell has no update gate: den_i = 1).  tanh(c_t) is a separate num/den division. */
typedef struct { float i, f, g, o, c, h; } orc_cell;
static inline orc_cell orc_lstm_cell(float z_i, float z_f, float z_g, float z_o, float c_prev, int coupled) {
    float num[4] = {0.0f, 0.0f, 0.0f, 0.0f}, den[4] = {1.0f, 1.0f, 1.0f, 1.0f};
    if (!coupled) sbr_tanh_pq(z_i * 0.5f, &num[0], &den[0]);
    sbr_tanh_pq(z_f * 0.5f, &num[1], &den[1]);
    sbr_tanh_pq(z_g, &num[2], &den[2]);
    sbr_tanh_pq(z_o * 0.5f, &num[3], &den[3]);
    const float den_if = den[0] * den[1];
    const float den_go = den[2] * den[3];
    const float R = 1.0f / (den_if * den_go);
    const float inv_if = R * den_go;
    const float inv_go = R * den_if;
    const float tanh_half_f = num[1] * (inv_if * den[0]);
    const float tanh_half_i = num[0] * (inv_if * den[1]);
    const float tanh_g = num[2] * (inv_go * den[3]);
    const float tanh_half_o = num[3] * (inv_go * den[2]);
    orc_cell r;
    r.f = fmaf(0.5f, tanh_half_f, 0.5f);
    r.i = coupled ? 1.0f - r.f : fmaf(0.5f, tanh_half_i, 0.5f);
    r.g = tanh_g;
    r.o = fmaf(0.5f, tanh_half_o, 0.5f);
    r.c = fmaf(r.f, c_prev, r.i * r.g);
    r.h = r.o * orc_tanh(r.c);
    return r;
}
/* Reverse mode through one cell.  Inputs: dL/dh_t (loss + recurrent), dL/dc_t carried from step
 * t+1, the stored gate values and cell states.  Outputs: dL/d(pre-activations) and the carry
 * dL/dc_{t-1}.  From h = o tanh(c):  do = dh tanh(c),  dc += dh o (1 - tanh(c)^2);
 * from c = f c_prev + i g:  df = dc c_prev, di = dc g, dg = dc i, dc_prev = dc f;  coupled: i = 1 - f
 * so df -= di;  sigmoid' = s (1 - s),  tanh' = 1 - t^2. */
typedef struct { float dz_i, dz_f, dz_g, dz_o, dc_prev; } orc_cell_grad;
static inline orc_cell_grad orc_lstm_cell_backward(float dh, float dc_carry, float i, float f, float g, float o,
                                                   float c, float c_prev, int coupled) {
    const float t = orc_tanh(c);
    const float grad_o = dh * t;
    const float grad_c = fmaf(dh * o, 1.0f - t * t, dc_carry);
    const float grad_i = grad_c * g;
    const float grad_g = grad_c * i;
    float grad_f = grad_c * c_prev;
    orc_cell_grad r;
    r.dc_prev = grad_c * f;
    if (coupled) {
        grad_f = grad_f - grad_i;
        r.dz_i = 0.0f;
    } else {
        r.dz_i = grad_i * (i * (1.0f - i));
    }
    r.dz_f = grad_f * (f * (1.0f - f));
    r.dz_g = grad_g * (1.0f - g * g);
    r.dz_o = grad_o * (o * (1.0f - o));
    return r;
}

/* ---- rand 0.5 as recalled: the reference's index and init streams -----------
 * XorShiftRng: Marsaglia xorshift128, state = four little-endian u32 of the 16 seed bytes (an
 * all-zero seed is replaced by fixed constants); next_u64 = two next_u32, low word first. */
typedef struct { uint32_t x, y, z, w; } orc_rng;
static inline void orc_rng_from_seed(orc_rng* r, const uint8_t seed[16]) {
    uint32_t word[4];
    for (int i = 0; i < 4; ++i) {
        word[i] = 0;
        for (int b = 3; b >= 0; --b) word[i] = (word[i] << 8) | seed[4 * i + b];
    }
    if (word[0] == 0 && word[1] == 0 && word[2] == 0 && word[3] == 0) {
        word[0] = 0x193a6754u; word[1] = 0xa8a7d469u; word[2] = 0x97830e05u; word[3] = 0x113ba7bbu;
    }
    r->x = word[0]; r->y = word[1]; r->z = word[2]; r->w = word[3];
}
static inline uint32_t orc_rng_u32(orc_rng* r) {
    const uint32_t x = r->x;
    const uint32_t t = x ^ (x << 11);
    r->x = r->y;
    r->y = r->z;
    r->z = r->w;
    const uint32_t w = r->w;
    r->w = w ^ (w >> 19) ^ (t ^ (t >> 8));
    return r->w;
}
static inline uint64_t orc_rng_u64(orc_rng* r) {
    const uint64_t first = orc_rng_u32(r);
    const uint64_t second = orc_rng_u32(r);
    return (second << 32) | first;
}
/* `parameters.rng().gen()` for a [u8; 16] seed (sequence_model.rs:97): sixteen u8 draws, each the low
 * byte of one next_u32 */
static inline void orc_rng_gen_seed(orc_rng* r, uint8_t out[16]) {
    for (int i = 0; i < 16; ++i) out[i] = (uint8_t)(orc_rng_u32(r) & 0xffu);
}
static inline void orc_wide_mul(uint64_t a, uint64_t b, uint64_t* hi, uint64_t* lo) {
    const __uint128_t p = (__uint128_t)a * b;
    *hi = (uint64_t)(p >> 64);
    *lo = (uint64_t)p;
}
/* Rng::gen_range(0, n) = UniformInt::<usize>::sample_single: zone = n << n.leading_zeros() */
static inline uint64_t orc_rng_gen_range(orc_rng* r, uint64_t n) {
    int lz = 0;
    while (!((n << lz) & 0x8000000000000000ULL)) ++lz;
    const uint64_t zone = n << lz;
    for (;;) {
        uint64_t hi, lo;
        orc_wide_mul(orc_rng_u64(r), n, &hi, &lo);
        if (lo <= zone) return hi;
    }
}
/* Uniform::new(low, high).sample(rng): ints_to_reject = (MAX - range + 1) % range */
static inline uint64_t orc_rng_uniform(orc_rng* r, uint64_t low, uint64_t high) {
    const uint64_t range = (high - 1) - low + 1;
    const uint64_t reject = (UINT64_MAX - range + 1) % range;
    const uint64_t zone = UINT64_MAX - reject;
    for (;;) {
        uint64_t hi, lo;
        orc_wide_mul(orc_rng_u64(r), range, &hi, &lo);
        if (lo <= zone) return low + hi;
    }
}
static inline double orc_f64_with_exponent(uint64_t fraction, int exponent) {
    union { uint64_t bits; double value; } u;
    u.bits = fraction | ((uint64_t)(exponent + 1023) << 52);
    return u.value;
}
/* StandardNormal (ziggurat).  One u64 gives the layer (low 8 bits) and u in [-1, 1) (top 52 bits as
 * the mantissa of a double in [2, 4), minus 3); rectangle test; layer 0 = tail by Marsaglia's
 * method with two Open01 draws per attempt; otherwise the wedge test with one [0,1) draw. */
static inline double orc_rng_standard_normal(orc_rng* r) {
    for (;;) {
        const uint64_t bits = orc_rng_u64(r);
        const unsigned layer = (unsigned)(bits & 0xffu);
        const double u = orc_f64_with_exponent(bits >> 12, 1) - 3.0;
        const double x = u * ORC_ZIG_NORM_X[layer];
        if (fabs(x) < ORC_ZIG_NORM_X[layer + 1]) return x;
        if (layer == 0) {
            double tx = 1.0, ty = 0.0;
            do {
                const double o1 = orc_f64_with_exponent(orc_rng_u64(r) >> 12, 0) - (1.0 - 0x1p-53);
                const double o2 = orc_f64_with_exponent(orc_rng_u64(r) >> 12, 0) - (1.0 - 0x1p-53);
                tx = log(o1) / ORC_ZIG_NORM_R;
                ty = log(o2);
            } while (-2.0 * ty < tx * tx);
            return u < 0.0 ? tx - ORC_ZIG_NORM_R : ORC_ZIG_NORM_R - tx;
        }
        const double unit = (double)(orc_rng_u64(r) >> 11) * 0x1p-53;
        const double f_hi = ORC_ZIG_NORM_F[layer], f_lo = ORC_ZIG_NORM_F[layer + 1];
        if (f_lo + (f_hi - f_lo) * unit < exp(-x * x / 2.0)) return x;
    }
}
/* Normal::new(mean, std_dev).sample(rng) as f32  (embedding_init, lstm.rs:22-25) */
static inline float orc_rng_normal_f32(orc_rng* r, double mean, double std_dev) {
    return (float)(mean + std_dev * orc_rng_standard_normal(r));
}

#endif /* ORC_NUMERICS_H */
