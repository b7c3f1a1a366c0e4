/* sbr_oracle.c — CPU ORACLE (test infrastructure, NOT the product).
 *
 * A plain-C restatement of the sbr-rs sequence-recommender hot path (sequential; one threaded mode for the timed CPU
 * baseline, orc_fit_threads), used only
 * by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg as the checker for the HIP
 * engine (libsbr_hip.so).  Nothing under sbr_rs_amd/ links, imports or executes this file.
 *
 * PARITY STATUS: the reference's arithmetic lives in the un-vendored crates wyrm ^0.9.1
 * (autodiff, LSTM cell, Adagrad, simd_dot; Cargo.toml:29, feature "fast-math"), rand ^0.5
 * (XorShiftRng, Uniform, Normal; Cargo.toml:19) and ndarray ^0.11; no Rust toolchain exists in
 * this image and Cargo.lock is git-ignored, so the reference cannot be built or run here.  The
 * oracle therefore restates the published algorithms (classic LSTM cell, Adagrad, xorshift128)
 * and is pinned against everything the reference's own tests hold for this path:
 *   - exact chunking known-answer test           src/data.rs:629-660   (tests/test_oracle.py)
 *   - FittingError::NoInteractions               src/models/lstm.rs:522-530
 *   - the five MovieLens-100K test-MRR lower bounds  lstm.rs:450-520, ewma.rs:463-507
 *   - split/CSR conservation property            src/data.rs:587-627
 * No reference test pins a float, an RNG output, a gradient or a rank, so bit-level parity with
 * the Rust path is UNPINNED ("parity unpinned" in DESIGN.md); parity of the HIP engine is
 * claimed against this oracle.
 *
 * Functions cite the reference lines they follow.  The scalar arithmetic (activations, dot orders,
 * losses, Adagrad/Adam, LSTM cell, negative-draw hash, rand 0.5 generators) is the oracle's own
 * statement in oracle/orc_numerics.h; from the product it shares nothing but the approximation
 * polynomial of sbr_rs_amd/csrc/sbr_approx.h (tests/test_abi.py enforces this).
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../include/sbr_hip.h"
#include "orc_numerics.h"

typedef struct orc_model {
    sbr_hparams hp;
    int d, ng; /* ng = gate blocks in W: 4 (normal) / 3 (coupled) / 0 (ewma); d = storage width (orc_storage_dim) */
    int dl;    /* embedding_dim as the caller sees it */
    float *E, *Eacc, *b, *bacc;
    float *W, *Wacc, *bW, *bWacc; /* W [2d][ng*d] */
    float *alpha, *alpha_acc;
    float *Em, *bm, *Wm, *bWm, *alpha_m; /* Adam first moments */
    float c1, c2;
    uint64_t opt_steps;
    orc_rng rng;
    uint64_t global_epoch;
    /* Reference-order mode (orc_model_set_reference_order; checker only, no engine counterpart): the two places where
     * the engine's contract departs from the reference's ORDER of work are replaced by the reference's own —
     *  (i)  negatives are drawn from the partition's sequential xorshift stream with rand 0.5's Uniform, one draw per try,
     *       exactly as sequence_model.rs:58-65 / :137 do (the contract keys every draw by a counter instead, so that
     *       draws can be evaluated in parallel);
     *  (ii) with num_devices > 1 every device's gradient is applied as its OWN optimiser step, one after the other in
     *       device order (wyrm's SynchronizedOptimizer as recalled, sequence_model.rs:163-166: the workers rendezvous,
     *       then each update goes in under exclusion — N Adagrad applications, G += g_q^2 each), where the contract adds
     *       the devices' gradients and applies one update.
     * batch_sequences must be 1 (the reference's schedule).  Used to measure whether the substitutions move test MRR
     * (tools/mrr_stream_sweep.py --reference-order, NOTES.md section 3). */
    int reference_order;
    float last_lagged_loss; /* orc_fit_end_lagged of the last orc_model_fit */
} orc_model;

typedef struct orc_local { /* one device's view of one minibatch */
    int R, B, Tm;
    int* off; /* [Tm+1] */
    uint32_t *in_idx, *out_idx, *ctr, *neg, *tries;
    float *coef, *loss;
    float *H, *C, *G, *dH, *dZ, *dX;
    float* dense;
    double loss_sum;
    uint64_t examples;
} orc_local;

typedef struct orc_plan {
    orc_model* m;
    int ndev;
    uint64_t nseq_total;
    uint64_t part_len;   /* subsequences per device partition */
    uint64_t* seq_start; /* [ndev][part_len] offsets into item_ids */
    uint32_t* seq_len;
    orc_rng* part_rng; /* [ndev] */
    uint64_t* fit_seed;     /* [ndev] */
    uint32_t* items;        /* copy of item_ids */
    uint64_t nnz;
    int Rmax;
    orc_local* loc; /* [ndev] */
    double loss_sum;
    uint64_t examples;
    double* loss_dev;       /* [ndev] per partition (sequence_model.rs:173-177 sums the partitions' ratios) */
    uint64_t* examples_dev;
    /* the figure the reference actually returns: see orc_lagged_loss_update */
    float* lagged_dev;
    float* lagged_node;     /* [ndev][max_sequence_length]: value left in the loss node of each length */
    uint64_t epochs_prepared;
    uint64_t epoch_key_epoch;
} orc_plan;

/* ------------------------------------------------------------------------------------------ */
/* The contract's dot orders (orc_numerics.h) are stated for widths 16 .. 256 in powers of two.  Any other
 * embedding_dim e <= 256 (lstm.rs:86-89 takes any usize) is DEFINED as the model of the next width up whose extra
 * embedding columns, weight rows / columns and alpha entries are zero: such a unit computes z = 0, c = 0, h = 0,
 * receives zero gradients and is left at zero by Adagrad / Adam, so that model is the e-wide model with +0 terms
 * in its sums (orc_model_padding_is_zero checks the invariant after training). */
static int orc_storage_dim(uint32_t e) {
    for (uint32_t p = 16; p <= 256; p *= 2) if (e >= 1 && e <= p) return (int)p;
    return 0;
}

/* ≙ Hyperparameters::build_params (lstm.rs:174-194, ewma.rs:167-198): E ~ N(0,(1/d)^2) drawn
 * row-major from the model RNG (embedding_init, lstm.rs:22-25: rand 0.5 Normal, f64 -> f32), biases 0,
 * alpha 0; then, from the same RNG,
 *   LSTM: wyrm nn::lstm::Parameters::new — recalled as four [(hidden+input) x hidden] matrices drawn
 *         one after the other in the order forget, update gate, update value, output gate, each
 *         xavier_normal (std 1/sqrt(rows) = 1/sqrt(2d)), rows = hidden part first (the cell stacks
 *         `hidden.stack(input)`); biases zero.  A coupled layer draws all four and ignores the update
 *         gate's.  Stored here as W[2d][ng*d], rows [x ; h], column blocks i,f,g,o (coupled: f,g,o).
 *   EWMA: the two unused d x d `dense_init` matrices fc1, fc2 (ewma.rs:179-188) — drawn and dropped so
 *         that the RNG the driver shuffles with is where the reference's is. */
int orc_model_create(const sbr_hparams* hp, orc_model** out) {
    if (!hp || !out) return SBR_ERR_INVALID_ARGUMENT;
    if (!orc_storage_dim(hp->embedding_dim) || hp->num_items == 0 || hp->max_sequence_length < 3 ||
        hp->num_devices == 0 || hp->num_devices > 16 || hp->batch_sequences == 0)
        return SBR_ERR_INVALID_ARGUMENT;
    if (hp->optimizer != SBR_OPT_ADAGRAD && hp->optimizer != SBR_OPT_ADAM) return SBR_ERR_INVALID_ARGUMENT;
    orc_model* m = (orc_model*)calloc(1, sizeof(orc_model));
    m->hp = *hp;
    int dl = m->dl = (int)hp->embedding_dim;
    int d = m->d = orc_storage_dim(hp->embedding_dim);
    m->ng = hp->model == SBR_MODEL_LSTM_NORMAL ? 4 : hp->model == SBR_MODEL_LSTM_COUPLED ? 3 : 0;
    size_t I = hp->num_items;
    m->E = (float*)calloc(I * d, sizeof(float));
    m->Eacc = (float*)calloc(I * d, sizeof(float));
    m->b = (float*)calloc(I, sizeof(float));
    m->bacc = (float*)calloc(I, sizeof(float));
    int adam = hp->optimizer == SBR_OPT_ADAM;
    m->c1 = m->c2 = 1.0f;
    if (adam) { m->Em = (float*)calloc(I * d, sizeof(float)); m->bm = (float*)calloc(I, sizeof(float)); }
    orc_rng_from_seed(&m->rng, hp->seed);
    double std_e = 1.0 / (double)dl; /* row-major, embedding_dim values per row */
    for (size_t r = 0; r < I; ++r)
        for (int c = 0; c < dl; ++c) m->E[r * d + c] = orc_rng_normal_f32(&m->rng, 0.0, std_e);
    if (m->ng) {
        size_t nw = (size_t)2 * d * m->ng * d;
        m->W = (float*)calloc(nw, sizeof(float));
        m->Wacc = (float*)calloc(nw, sizeof(float));
        m->bW = (float*)calloc((size_t)m->ng * d, sizeof(float));
        m->bWacc = (float*)calloc((size_t)m->ng * d, sizeof(float));
        if (adam) { m->Wm = (float*)calloc(nw, sizeof(float)); m->bWm = (float*)calloc((size_t)m->ng * d, sizeof(float)); }
        double std_w = 1.0 / sqrt((double)(2 * dl));
        int nz = m->ng * d;
        for (int gate = 0; gate < 4; ++gate) { /* wyrm order: forget, update gate, update value, output gate */
            int block = m->ng == 4 ? (gate == 0 ? 1 : gate == 1 ? 0 : gate) : (gate == 0 ? 0 : gate == 1 ? -1 : gate - 1);
            for (int row = 0; row < 2 * dl; ++row) {
                int k = row < dl ? d + row : row - dl; /* wyrm rows: hidden first; here rows are [x ; h] */
                for (int u = 0; u < dl; ++u) {
                    float v = orc_rng_normal_f32(&m->rng, 0.0, std_w);
                    if (block >= 0) m->W[(size_t)k * nz + block * d + u] = v;
                }
            }
        }
    } else {
        m->alpha = (float*)calloc(d, sizeof(float));
        m->alpha_acc = (float*)calloc(d, sizeof(float));
        if (adam) m->alpha_m = (float*)calloc(d, sizeof(float));
        double std_fc = sqrt(2.0 / (double)(dl + dl)); /* dense_init, ewma.rs:38-41 */
        for (int i = 0; i < 2 * dl * dl; ++i) (void)orc_rng_normal_f32(&m->rng, 0.0, std_fc);
    }
    *out = m;
    return SBR_OK;
}

void orc_model_destroy(orc_model* m) {
    if (!m) return;
    free(m->E); free(m->Eacc); free(m->b); free(m->bacc);
    free(m->W); free(m->Wacc); free(m->bW); free(m->bWacc);
    free(m->alpha); free(m->alpha_acc);
    free(m->Em); free(m->bm); free(m->Wm); free(m->bWm); free(m->alpha_m);
    free(m);
}

/* parameter arrays in the caller's shapes (embedding_dim dl): E [I][dl], W [2 dl][ng dl], bW [ng dl], alpha [dl];
 * kind: 0 flat, 1 rows of the item table, 2 LSTM weight matrix, 3 LSTM bias */
static float* orc_param_ptr(orc_model* m, int which, uint64_t* count, int* kind) {
    uint64_t I = m->hp.num_items, dl = (uint64_t)m->dl, ng = (uint64_t)m->ng;
    float* p = NULL; uint64_t n = 0; int k = 0;
    switch (which) {
        case SBR_PARAM_ITEM_EMBEDDING: p = m->E; n = I * dl; k = 1; break;
        case SBR_PARAM_ITEM_EMBEDDING_ACC: p = m->Eacc; n = I * dl; k = 1; break;
        case SBR_PARAM_ITEM_BIAS: p = m->b; n = I; break;
        case SBR_PARAM_ITEM_BIAS_ACC: p = m->bacc; n = I; break;
        case SBR_PARAM_LSTM_W: p = m->W; n = 2 * dl * ng * dl; k = 2; break;
        case SBR_PARAM_LSTM_W_ACC: p = m->Wacc; n = 2 * dl * ng * dl; k = 2; break;
        case SBR_PARAM_LSTM_B: p = m->bW; n = ng * dl; k = 3; break;
        case SBR_PARAM_LSTM_B_ACC: p = m->bWacc; n = ng * dl; k = 3; break;
        case SBR_PARAM_EWMA_ALPHA: p = m->alpha; n = ng ? 0 : dl; break;
        case SBR_PARAM_EWMA_ALPHA_ACC: p = m->alpha_acc; n = ng ? 0 : dl; break;
        case SBR_PARAM_ITEM_EMBEDDING_M: p = m->Em; n = m->Em ? I * dl : 0; k = 1; break;
        case SBR_PARAM_ITEM_BIAS_M: p = m->bm; n = m->bm ? I : 0; break;
        case SBR_PARAM_LSTM_W_M: p = m->Wm; n = m->Wm ? 2 * dl * ng * dl : 0; k = 2; break;
        case SBR_PARAM_LSTM_B_M: p = m->bWm; n = m->bWm ? ng * dl : 0; k = 3; break;
        case SBR_PARAM_EWMA_ALPHA_M: p = m->alpha_m; n = m->alpha_m ? dl : 0; break;
    }
    *count = n;
    if (kind) *kind = k;
    return p;
}
/* stored position of logical element i of a parameter array */
static size_t orc_param_stored_index(const orc_model* m, int kind, uint64_t count, uint64_t i) {
    uint64_t d = (uint64_t)m->d, dl = (uint64_t)m->dl, ng = (uint64_t)m->ng;
    switch (kind) {
        case 1: return (size_t)((i / dl) * d + i % dl);
        case 2: { uint64_t kl = i / (ng * dl), j = i % (ng * dl), k = kl < dl ? kl : d + (kl - dl); return (size_t)(k * ng * d + (j / dl) * d + j % dl); }
        case 3: return (size_t)((i / dl) * d + i % dl);
        default: (void)count; return (size_t)i;
    }
}
int orc_model_param_count(orc_model* m, int which, uint64_t* out) {
    orc_param_ptr(m, which, out, NULL);
    return SBR_OK;
}
int orc_model_get_param(orc_model* m, int which, float* out, uint64_t count) {
    uint64_t n; int kind;
    float* p = orc_param_ptr(m, which, &n, &kind);
    if (!p || n != count) return SBR_ERR_INVALID_ARGUMENT;
    for (uint64_t i = 0; i < n; ++i) out[i] = p[orc_param_stored_index(m, kind, n, i)];
    return SBR_OK;
}
/* selected rows of an item-table block: embeddings / their optimiser state [n][embedding_dim], biases [n] */
int orc_model_get_param_rows(orc_model* m, int which, const uint32_t* rows, uint64_t n, float* out) {
    uint64_t cnt; int kind;
    float* p = orc_param_ptr(m, which, &cnt, &kind);
    uint64_t I = m->hp.num_items, d = (uint64_t)m->d, dl = (uint64_t)m->dl;
    if (!p || !cnt || (kind != 1 && cnt != I)) return SBR_ERR_INVALID_ARGUMENT;
    for (uint64_t i = 0; i < n; ++i) {
        if (rows[i] >= I) return SBR_ERR_INVALID_ARGUMENT;
        if (kind == 1) memcpy(out + i * dl, p + (uint64_t)rows[i] * d, dl * 4);
        else out[i] = p[rows[i]];
    }
    return SBR_OK;
}
int orc_model_set_param(orc_model* m, int which, const float* in, uint64_t count) {
    uint64_t n; int kind;
    float* p = orc_param_ptr(m, which, &n, &kind);
    if (!p || n != count) return SBR_ERR_INVALID_ARGUMENT;
    for (uint64_t i = 0; i < n; ++i) p[orc_param_stored_index(m, kind, n, i)] = in[i]; /* the padding keeps its zeros */
    return SBR_OK;
}
/* 1 if every padded element of every parameter and optimiser-state array is still zero (the fixed point the
 * definition of a non-power-of-two embedding_dim rests on) */
int orc_model_padding_is_zero(orc_model* m) {
    uint64_t I = m->hp.num_items, d = (uint64_t)m->d, dl = (uint64_t)m->dl, ng = (uint64_t)m->ng;
    const float* tables[3] = {m->E, m->Eacc, m->Em};
    for (int a = 0; a < 3; ++a)
        if (tables[a])
            for (uint64_t r = 0; r < I; ++r)
                for (uint64_t c = dl; c < d; ++c) if (tables[a][r * d + c] != 0.0f) return 0;
    const float* mats[3] = {m->W, m->Wacc, m->Wm};
    for (int a = 0; a < 3; ++a)
        if (mats[a])
            for (uint64_t k = 0; k < 2 * d; ++k)
                for (uint64_t j = 0; j < ng * d; ++j) {
                    int pad = (k % d) >= dl || (j % d) >= dl;
                    if (pad && mats[a][k * ng * d + j] != 0.0f) return 0;
                }
    const float* vecs[6] = {m->bW, m->bWacc, m->bWm, m->alpha, m->alpha_acc, m->alpha_m};
    for (int a = 0; a < 6; ++a)
        if (vecs[a]) {
            uint64_t n = a < 3 ? ng * d : d;
            for (uint64_t j = 0; j < n; ++j) if ((j % d) >= dl && vecs[a][j] != 0.0f) return 0;
        }
    return 1;
}
uint64_t orc_model_get_epoch(orc_model* m) { return m->global_epoch; }
int orc_model_set_reference_order(orc_model* m, int on) {
    if (!m) return SBR_ERR_INVALID_ARGUMENT;
    m->reference_order = on ? 1 : 0;
    return SBR_OK;
}
uint64_t orc_model_get_opt_steps(orc_model* m) { return m->opt_steps; }
/* state of the model RNG as the 16 seed bytes that re-create it (x, y, z, w little-endian): what the reference's
 * Hyperparameters.rng holds after build_params (lstm.rs:174-194) */
void orc_model_get_rng(orc_model* m, uint8_t out[16]) {
    const uint32_t v[4] = { m->rng.x, m->rng.y, m->rng.z, m->rng.w };
    memcpy(out, v, 16);
}

/* optimiser element update (≙ wyrm optim::{Adagrad, Adam} as recalled) */
static void orc_opt(orc_model* m, float* w, float* acc, float* mom, float g) {
    if (m->hp.optimizer == SBR_OPT_ADAM) orc_adam_step(w, mom, acc, g, m->hp.learning_rate, m->hp.l2_penalty, m->c1, m->c2);
    else orc_adagrad_step(w, acc, g, m->hp.learning_rate, m->hp.l2_penalty);
}
static void orc_begin_optimizer_step(orc_model* m) {
    m->opt_steps += 1;
    if (m->hp.optimizer == SBR_OPT_ADAM) orc_adam_bias_corrections(m->opt_steps, &m->c1, &m->c2);
}
static void orc_dense_update(orc_model* m, const float* dg) {
    int d = m->d;
    float dummy = 0.0f;
    if (m->ng) {
        int nz = m->ng * d;
        for (size_t i = 0; i < (size_t)2 * d * nz; ++i) orc_opt(m, &m->W[i], &m->Wacc[i], m->Wm ? &m->Wm[i] : &dummy, dg[i]);
        for (int j = 0; j < nz; ++j) orc_opt(m, &m->bW[j], &m->bWacc[j], m->bWm ? &m->bWm[j] : &dummy, dg[(size_t)2 * d * nz + j]);
    } else {
        for (int k = 0; k < d; ++k) orc_opt(m, &m->alpha[k], &m->alpha_acc[k], m->alpha_m ? &m->alpha_m[k] : &dummy, dg[k]);
    }
}
static void orc_row_update(orc_model* m, uint64_t row, const float* g, int has_g, int has_b, float gb) {
    int d = m->d;
    float dummy = 0.0f;
    if (has_g) {
        float* wrow = m->E + (size_t)row * d; float* arow = m->Eacc + (size_t)row * d;
        for (int k = 0; k < d; ++k) orc_opt(m, &wrow[k], &arow[k], m->Em ? &m->Em[(size_t)row * d + k] : &dummy, g[k]);
    }
    if (has_b) orc_opt(m, &m->b[row], &m->bacc[row], m->bm ? &m->bm[row] : &dummy, gb);
}

/* ------------------------------------------------------------------------------------------ */
/* ≙ CompressedInteractionsUserChunkIterator::next (data.rs:406-431): the FIRST chunk is the
 * short one.  Writes chunk lengths; returns their number. */
int orc_chunk_lengths(uint64_t user_len, uint64_t chunk_size, uint64_t* out, int max_out) {
    int n = 0;
    uint64_t idx = 0;
    while (idx < user_len) {
        uint64_t mod = (user_len - idx) % chunk_size;
        uint64_t cs = mod == 0 ? chunk_size : mod;
        if (n < max_out) out[n] = cs;
        ++n;
        idx += cs;
    }
    return n;
}

/* rand 0.5 Rng::shuffle as recalled: `let mut i = len; while i >= 2 { i -= 1;
 * values.swap(i, self.gen_range(0, i + 1)); }` — applied to the (start, len) pairs */
static void orc_shuffle(uint64_t* start, uint32_t* len, uint64_t n, orc_rng* r) {
    uint64_t i = n;
    while (i >= 2) {
        i -= 1;
        uint64_t j = orc_rng_gen_range(r, i + 1);
        uint64_t ts = start[i]; start[i] = start[j]; start[j] = ts;
        uint32_t tl = len[i]; len[i] = len[j]; len[j] = tl;
    }
}

static void orc_local_alloc(orc_local* L, int Rmax, int Tmax, int d, int ng) {
    memset(L, 0, sizeof(*L));
    L->off = (int*)calloc(Tmax + 1, sizeof(int));
    L->in_idx = (uint32_t*)calloc(Rmax, 4); L->out_idx = (uint32_t*)calloc(Rmax, 4);
    L->ctr = (uint32_t*)calloc(Rmax, 4); L->neg = (uint32_t*)calloc(Rmax, 4);
    L->tries = (uint32_t*)calloc(Rmax, 4);
    L->coef = (float*)calloc(Rmax, 4); L->loss = (float*)calloc(Rmax, 4);
    size_t rd = (size_t)Rmax * d;
    L->H = (float*)calloc(rd, 4); L->dH = (float*)calloc(rd, 4); L->dX = (float*)calloc(rd, 4);
    if (ng) {
        L->C = (float*)calloc(rd, 4);
        L->G = (float*)calloc(rd * 4, 4);
        L->dZ = (float*)calloc(rd * ng, 4);
        L->dense = (float*)calloc((size_t)(2 * d + 1) * ng * d, 4);
    } else {
        L->dense = (float*)calloc(d, 4);
    }
}
static void orc_local_free(orc_local* L) {
    free(L->off); free(L->in_idx); free(L->out_idx); free(L->ctr); free(L->neg); free(L->tries);
    free(L->coef); free(L->loss); free(L->H); free(L->dH); free(L->dX); free(L->C); free(L->G);
    free(L->dZ); free(L->dense);
}

/* ≙ fit_sequence_model, set-up part (sequence_model.rs:74-98): subsequences = chunks with
 * len > 2 (:76-83); shuffle with the model RNG (:84); NoInteractions if empty (:86-88);
 * num_chunks = len / num_threads, zip drops the remainder (:91-98); one RNG per partition seeded
 * with 16 bytes from the model RNG (:97). */
int orc_fit_begin(orc_model* m, const uint64_t* user_ptr, const uint32_t* item_ids, uint64_t num_users,
                  orc_plan** out) {
    uint64_t T = m->hp.max_sequence_length;
    uint64_t nseq = 0;
    for (uint64_t u = 0; u < num_users; ++u) {
        uint64_t n = user_ptr[u + 1] - user_ptr[u], idx = 0;
        while (idx < n) {
            uint64_t mod = (n - idx) % T, cs = mod == 0 ? T : mod;
            if (cs > 2) ++nseq;
            idx += cs;
        }
    }
    if (nseq == 0) return SBR_ERR_NO_INTERACTIONS;
    if (m->reference_order && (m->hp.batch_sequences != 1 || (m->hp.num_devices > 1 && m->hp.parallelism != SBR_PAR_SYNCHRONOUS)))
        return SBR_ERR_INVALID_ARGUMENT; /* the reference's schedule; Hogwild has no order to restate */
    uint64_t* start = (uint64_t*)malloc(nseq * 8);
    uint32_t* len = (uint32_t*)malloc(nseq * 4);
    uint64_t k = 0;
    for (uint64_t u = 0; u < num_users; ++u) {
        uint64_t n = user_ptr[u + 1] - user_ptr[u], idx = 0;
        while (idx < n) {
            uint64_t mod = (n - idx) % T, cs = mod == 0 ? T : mod;
            if (cs > 2) { start[k] = user_ptr[u] + idx; len[k] = (uint32_t)cs; ++k; }
            idx += cs;
        }
    }
    orc_shuffle(start, len, nseq, &m->rng);
    int ndev = (int)m->hp.num_devices;
    uint64_t part = nseq / ndev;
    if (part == 0) { free(start); free(len); return SBR_ERR_INVALID_ARGUMENT; } /* reference panics (chunks_mut(0)) */
    orc_plan* p = (orc_plan*)calloc(1, sizeof(orc_plan));
    p->m = m; p->ndev = ndev; p->nseq_total = nseq; p->part_len = part;
    p->seq_start = start; p->seq_len = len;
    p->part_rng = (orc_rng*)calloc(ndev, sizeof(orc_rng));
    p->fit_seed = (uint64_t*)calloc(ndev, 8);
    for (int q = 0; q < ndev; ++q) {
        uint8_t seed[16];
        orc_rng_gen_seed(&m->rng, seed);
        orc_rng_from_seed(&p->part_rng[q], seed);
        /* contract: the key of the counter-based negative draws; the reference's thread_rng serves shuffles and draws only */
        if (!m->reference_order) p->fit_seed[q] = orc_rng_u64(&p->part_rng[q]);
    }
    p->nnz = user_ptr[num_users];
    p->items = (uint32_t*)malloc((p->nnz ? p->nnz : 1) * 4);
    memcpy(p->items, item_ids, p->nnz * 4);
    p->Rmax = (int)(m->hp.batch_sequences * (T - 1));
    p->loc = (orc_local*)calloc(ndev, sizeof(orc_local));
    for (int q = 0; q < ndev; ++q) orc_local_alloc(&p->loc[q], p->Rmax, (int)T, m->d, m->ng);
    p->lagged_node = (float*)calloc((size_t)ndev * T, sizeof(float));
    p->loss_dev = (double*)calloc(ndev, sizeof(double));
    p->examples_dev = (uint64_t*)calloc(ndev, sizeof(uint64_t));
    p->lagged_dev = (float*)calloc(ndev, sizeof(float));
    *out = p;
    return SBR_OK;
}

void orc_fit_plan_destroy(orc_plan* p) {
    if (!p) return;
    for (int q = 0; q < p->ndev; ++q) orc_local_free(&p->loc[q]);
    free(p->loc); free(p->seq_start); free(p->seq_len); free(p->part_rng); free(p->fit_seed); free(p->items);
    free(p->lagged_node); free(p->loss_dev); free(p->examples_dev); free(p->lagged_dev);
    free(p);
}

/* ≙ thread_rng.shuffle(partition) at the top of every epoch (sequence_model.rs:109) */
int orc_fit_epoch_prepare(orc_plan* p, uint64_t* out_num_minibatches) {
    for (int q = 0; q < p->ndev; ++q)
        orc_shuffle(p->seq_start + (uint64_t)q * p->part_len, p->seq_len + (uint64_t)q * p->part_len, p->part_len,
                    &p->part_rng[q]);
    p->epoch_key_epoch = p->m->global_epoch;
    p->m->global_epoch += 1;
    p->epochs_prepared += 1;
    uint64_t B = p->m->hp.batch_sequences;
    if (out_num_minibatches) *out_num_minibatches = (p->part_len + B - 1) / B;
    return SBR_OK;
}

/* Pack minibatch mb of device q: subsequences [mb*B, min((mb+1)*B, part_len)) of the partition,
 * ordered by length descending (stable), rows time-major: row(t, b) = off[t] + b.
 * ≙ the per-step index assignment loop (sequence_model.rs:115-142): in_t = item[t],
 * out_t = item[t+1] for t = 0..len-2. */
static void orc_pack(orc_plan* p, int q, uint64_t mb, orc_local* L) {
    uint64_t B = p->m->hp.batch_sequences, T = p->m->hp.max_sequence_length;
    uint64_t p0 = mb * B, p1 = p0 + B;
    if (p1 > p->part_len) p1 = p->part_len;
    int nb = (int)(p1 - p0);
    const uint64_t* st = p->seq_start + (uint64_t)q * p->part_len;
    const uint32_t* ln = p->seq_len + (uint64_t)q * p->part_len;
    /* stable counting sort by length descending */
    int* order = (int*)malloc(sizeof(int) * (nb ? nb : 1));
    int* cnt = (int*)calloc(T + 2, sizeof(int));
    for (int i = 0; i < nb; ++i) cnt[ln[p0 + i]]++;
    int* pos = (int*)calloc(T + 2, sizeof(int));
    int acc = 0;
    for (int l = (int)T; l >= 0; --l) { pos[l] = acc; acc += cnt[l]; }
    for (int i = 0; i < nb; ++i) order[pos[ln[p0 + i]]++] = i;
    int Tm = nb ? (int)ln[p0 + order[0]] - 1 : 0;
    L->B = nb; L->Tm = Tm;
    L->off[0] = 0;
    for (int t = 0; t < Tm; ++t) {
        int bt = 0;
        for (int b = 0; b < nb; ++b) if ((int)ln[p0 + order[b]] - 1 > t) bt = b + 1; else break;
        L->off[t + 1] = L->off[t] + bt;
    }
    L->R = L->off[Tm];
    for (int b = 0; b < nb; ++b) {
        uint64_t pp = p0 + order[b];
        int n = (int)ln[pp];
        for (int t = 0; t < n - 1; ++t) {
            int r = L->off[t] + b;
            L->in_idx[r] = p->items[st[pp] + t];
            L->out_idx[r] = p->items[st[pp] + t + 1];
            L->ctr[r] = (uint32_t)(pp * T + (uint64_t)t);
        }
    }
    free(order); free(cnt); free(pos);
}

/* SAMPLED parity at sizes the oracle cannot run whole (tests/test_parity_gpu.py::test_bench_regime_*): pack only the
 * sequences sel_b[0..nsel) — indices b of the FULL minibatch's packed order (length descending, stable), ascending — with the
 * position counters `ctr` they have in the full minibatch, so that forward, negative draws, loss and BPTT of each of them are
 * exactly what the full step computes for it (a sequence's rows depend on the parameters and on its own items only).
 * out_full_rows[compact row] = the packed row off_full[t] + b of the same (t, sequence) in the full minibatch. */
static int orc_pack_sample(orc_plan* p, int q, uint64_t mb, const uint32_t* sel_b, uint32_t nsel, orc_local* L, uint32_t* out_full_rows,
                           uint64_t* out_off_full /* [T] rows of the full minibatch before step t; may be NULL */) {
    uint64_t B = p->m->hp.batch_sequences, T = p->m->hp.max_sequence_length;
    uint64_t p0 = mb * B, p1 = p0 + B;
    if (p1 > p->part_len) p1 = p->part_len;
    int nb = (int)(p1 - p0);
    const uint64_t* st = p->seq_start + (uint64_t)q * p->part_len;
    const uint32_t* ln = p->seq_len + (uint64_t)q * p->part_len;
    for (uint32_t j = 0; j < nsel; ++j)
        if (sel_b[j] >= (uint32_t)nb || (j && sel_b[j] <= sel_b[j - 1])) return SBR_ERR_INVALID_ARGUMENT;
    int* order = (int*)malloc(sizeof(int) * (nb ? nb : 1));
    int* cnt = (int*)calloc(T + 2, sizeof(int));
    for (int i = 0; i < nb; ++i) cnt[ln[p0 + i]]++;
    int* pos = (int*)calloc(T + 2, sizeof(int));
    int acc = 0;
    for (int l = (int)T; l >= 0; --l) { pos[l] = acc; acc += cnt[l]; }
    for (int i = 0; i < nb; ++i) order[pos[ln[p0 + i]]++] = i;
    /* full minibatch: off_full[t] = rows before step t = sum over steps < t of the sequences alive there */
    uint64_t* off_full = (uint64_t*)calloc(T + 1, sizeof(uint64_t));
    for (uint64_t t = 0; t + 1 < T; ++t) {
        uint64_t alive = 0; /* sequences with len - 1 > t: lengths are sorted descending, count by the histogram */
        for (uint64_t l = t + 2; l <= T; ++l) alive += (uint64_t)cnt[l];
        off_full[t + 1] = off_full[t] + alive;
    }
    int Tm = nsel ? (int)ln[p0 + order[sel_b[0]]] - 1 : 0;
    L->B = (int)nsel; L->Tm = Tm;
    L->off[0] = 0;
    for (int t = 0; t < Tm; ++t) {
        int bt = 0;
        for (uint32_t j = 0; j < nsel; ++j) if ((int)ln[p0 + order[sel_b[j]]] - 1 > t) bt = (int)j + 1; else break;
        L->off[t + 1] = L->off[t] + bt;
    }
    L->R = L->off[Tm];
    for (uint32_t j = 0; j < nsel; ++j) {
        uint64_t pp = p0 + order[sel_b[j]];
        int n = (int)ln[pp];
        for (int t = 0; t < n - 1; ++t) {
            int r = L->off[t] + (int)j;
            L->in_idx[r] = p->items[st[pp] + t];
            L->out_idx[r] = p->items[st[pp] + t + 1];
            L->ctr[r] = (uint32_t)(pp * T + (uint64_t)t);
            out_full_rows[r] = (uint32_t)(off_full[t] + sel_b[j]);
        }
    }
    if (out_off_full) memcpy(out_off_full, off_full, T * sizeof(uint64_t));
    free(order); free(cnt); free(pos); free(off_full);
    return SBR_OK;
}

int orc_fit_minibatch_rows(orc_plan* p, int q, uint64_t mb, uint64_t* out_rows) {
    uint64_t B = p->m->hp.batch_sequences;
    uint64_t p0 = mb * B, p1 = p0 + B;
    if (p1 > p->part_len) p1 = p->part_len;
    const uint32_t* ln = p->seq_len + (uint64_t)q * p->part_len;
    uint64_t r = 0;
    for (uint64_t i = p0; i < p1; ++i) r += ln[i] - 1;
    *out_rows = r;
    return SBR_OK;
}

/* packed layout of the device's LAST local step: off[t] = packed rows before step t (Tm + 1 entries written; returns Tm) */
int orc_fit_last_offsets(orc_plan* p, int q, uint64_t* out_off, uint64_t cap, uint64_t* out_tm) {
    const orc_local* L = &p->loc[q];
    if ((uint64_t)L->Tm + 1 > cap) return SBR_ERR_INVALID_ARGUMENT;
    for (int t = 0; t <= L->Tm; ++t) out_off[t] = (uint64_t)L->off[t];
    *out_tm = (uint64_t)L->Tm;
    return SBR_OK;
}

/* ------------------------------------------------------------------------------------------ */
/* Recurrent forward over the packed minibatch.
 * LSTM (≙ wyrm nn::lstm Layer::forward, lstm.rs:293-298): z = bW + [x_t ; h_{t-1}] W as a
 * k-ascending fma chain per output column, h_{-1} = c_{-1} = 0.
 * EWMA (≙ ewma.rs:302-313): s_0 = E[in_0]; s_t = sig(alpha) * s_{t-1} + (1 - sig(alpha)) * E[in_t]. */
static void orc_forward(orc_model* m, orc_local* L) {
    int d = m->d, ng = m->ng, coupled = m->hp.model == SBR_MODEL_LSTM_COUPLED;
    if (ng) {
        int nz = ng * d;
        float* z = (float*)malloc(sizeof(float) * nz);
        for (int t = 0; t < L->Tm; ++t) {
            int bt = L->off[t + 1] - L->off[t];
            for (int b = 0; b < bt; ++b) {
                int r = L->off[t] + b;
                const float* x = m->E + (size_t)L->in_idx[r] * d;
                const float* hp = t ? L->H + (size_t)(L->off[t - 1] + b) * d : NULL;
                const float* cp = t ? L->C + (size_t)(L->off[t - 1] + b) * d : NULL;
                for (int j = 0; j < nz; ++j) z[j] = m->bW[j];
                for (int k = 0; k < d; ++k) {
                    float xv = x[k];
                    const float* w = m->W + (size_t)k * nz;
                    for (int j = 0; j < nz; ++j) z[j] = fmaf(xv, w[j], z[j]);
                }
                for (int k = 0; k < d; ++k) {
                    float hv = hp ? hp[k] : 0.0f;
                    const float* w = m->W + (size_t)(d + k) * nz;
                    for (int j = 0; j < nz; ++j) z[j] = fmaf(hv, w[j], z[j]);
                }
                float* g = L->G + (size_t)r * 4 * d;
                for (int u = 0; u < d; ++u) {
                    float zi = coupled ? 0.0f : z[u];
                    float zf = coupled ? z[u] : z[d + u];
                    float zg = coupled ? z[d + u] : z[2 * d + u];
                    float zo = coupled ? z[2 * d + u] : z[3 * d + u];
                    orc_cell cell = orc_lstm_cell(zi, zf, zg, zo, cp ? cp[u] : 0.0f, coupled);
                    g[u] = cell.i; g[d + u] = cell.f; g[2 * d + u] = cell.g; g[3 * d + u] = cell.o;
                    L->C[(size_t)r * d + u] = cell.c;
                    L->H[(size_t)r * d + u] = cell.h;
                }
            }
        }
        free(z);
    } else {
        for (int t = 0; t < L->Tm; ++t) {
            int bt = L->off[t + 1] - L->off[t];
            for (int b = 0; b < bt; ++b) {
                int r = L->off[t] + b;
                const float* x = m->E + (size_t)L->in_idx[r] * d;
                float* s = L->H + (size_t)r * d;
                if (t == 0) {
                    for (int k = 0; k < d; ++k) s[k] = x[k];
                } else {
                    const float* sp = L->H + (size_t)(L->off[t - 1] + b) * d;
                    for (int k = 0; k < d; ++k) {
                        float a = orc_sigmoid(m->alpha[k]);
                        float oma = 1.0f - a;
                        s[k] = fmaf(a, sp[k], oma * x[k]);
                    }
                }
            }
        }
    }
}

/* ≙ predict_single in training (lstm.rs:338-350): bias + dot, "tree" order */
static float orc_score_tree(const orc_model* m, const float* h, uint32_t item) {
    return m->b[item] + orc_dot_training(h, m->E + (size_t)item * m->d, m->d);
}

/* Negative sampling + loss + dloss/dh for every packed row.
 * ≙ sequence_model.rs:125-141 (negative choice), sample_warp_negative (:47-68), the loss nodes
 * (lstm.rs:300-320) and the dot-node backward into h. */
static void orc_score(orc_model* m, orc_local* L, uint64_t epoch_key, orc_rng* thread_rng) {
    int d = m->d;
    uint32_t I = m->hp.num_items;
    L->loss_sum = 0.0;
    for (int r = 0; r < L->R; ++r) {
        const float* h = L->H + (size_t)r * d;
        uint32_t pi = L->out_idx[r];
        float pos = orc_score_tree(m, h, pi);
        uint32_t nj = 0;
        float neg = 0.0f;
        uint32_t tries = 0;
        int max_tries = m->hp.loss == SBR_LOSS_WARP ? ORC_WARP_TRIES : 1;
        for (int k = 0; k < max_tries; ++k) {
            /* reference order (batch_sequences = 1: packed row r is step r of the one sequence): `uniform.sample(rng)`,
             * Uniform::new(0, num_items) over the worker's own stream, sequence_model.rs:59 / :137 */
            nj = m->reference_order ? (uint32_t)orc_rng_uniform(thread_rng, 0, I) : orc_negative_draw(epoch_key, L->ctr[r], (uint32_t)k, I);
            neg = orc_score_tree(m, h, nj);
            ++tries;
            if (orc_warp_accepts(pos, neg)) break;
        }
        float g, l;
        if (m->hp.loss == SBR_LOSS_BPR) l = orc_bpr(pos, neg, &g); else l = orc_hinge(pos, neg, &g);
        L->neg[r] = nj; L->tries[r] = tries; L->coef[r] = g; L->loss[r] = l;
        L->loss_sum += (double)l;
        const float* en = m->E + (size_t)nj * d;
        const float* ep = m->E + (size_t)pi * d;
        float* dh = L->dH + (size_t)r * d;
        for (int k = 0; k < d; ++k) dh[k] = g * en[k] - g * ep[k];
    }
    L->examples = (uint64_t)L->R;
}

/* The loss figure the reference returns.  sequence_model.rs:157 adds `loss.value().scalar_sum()` of the node
 * losses[loss_idx] BEFORE :160 runs `loss.forward()` on it, so what a worker accumulates for a sequence is whatever
 * earlier forward passes left in that node.  The nodes are the running sums of lstm.rs:322-328 / ewma.rs:337-343:
 * summed_losses[k] = summed_losses[k-1].clone() + loss_k shares the Rc node, so `forward()` on the node of a sequence
 * with s steps evaluates — and leaves its value in — every node 0 .. s-1, and does not touch the nodes above.  The
 * value read for a sequence with s steps is therefore the running sum L_{s-1} of the worker's most recent earlier
 * sequence with AT LEAST s steps (0 before the first one: the graph is built per fit call, :103; wyrm keeps a node's
 * value until its next forward, recalled).  Accumulated in f32 like the reference's `loss_value`; sequences of one
 * minibatch in minibatch order (with batch_sequences = 1 this is the reference's order).  The engine reports the true
 * sums (orc_fit_end); this figure exists so that the two can be told apart
 * (tests/test_oracle.py::test_lagged_loss_figure, ::test_lagged_loss_mixed_lengths). */
static void orc_lagged_loss_update(orc_plan* p, int q) {
    const orc_local* L = &p->loc[q];
    float* node = p->lagged_node + (size_t)q * p->m->hp.max_sequence_length;
    for (int b = 0; b < L->B; ++b) {
        int steps = 0;
        for (int t = 0; t < L->Tm && b < L->off[t + 1] - L->off[t]; ++t) ++steps;
        if (steps == 0) continue;
        p->lagged_dev[q] = p->lagged_dev[q] + node[steps - 1]; /* :157, before the forward pass of :160 */
        float sum = 0.0f;                                       /* L_t = L_{t-1} + l_t (lstm.rs:322-328) */
        for (int t = 0; t < steps; ++t) {
            sum = sum + L->loss[L->off[t] + b];
            node[t] = sum;
        }
    }
}

/* BPTT (≙ loss.backward(1.0), sequence_model.rs:161) + dense gradient of this device.
 * Dense reduction order: packed rows are cut into chunks of ORC_DW_CHUNK_ROWS; inside a chunk a
 * row-ascending fma chain from 0; chunk partials added in chunk order. */
static void orc_backward(orc_model* m, orc_local* L) {
    int d = m->d, ng = m->ng, coupled = m->hp.model == SBR_MODEL_LSTM_COUPLED;
    if (ng) {
        int nz = ng * d;
        float* dh_rec = (float*)calloc((size_t)L->B * d, 4);
        float* dc_rec = (float*)calloc((size_t)L->B * d, 4);
        float* dxh = (float*)malloc(sizeof(float) * 2 * d);
        /* W^T so the j-chain runs over contiguous memory */
        float* WT = (float*)malloc(sizeof(float) * (size_t)2 * d * nz);
        for (int k = 0; k < 2 * d; ++k) for (int j = 0; j < nz; ++j) WT[(size_t)j * 2 * d + k] = m->W[(size_t)k * nz + j];
        for (int t = L->Tm - 1; t >= 0; --t) {
            int bt = L->off[t + 1] - L->off[t];
            int bnext = t + 1 < L->Tm ? L->off[t + 2] - L->off[t + 1] : 0;
            for (int b = 0; b < bt; ++b) {
                int r = L->off[t] + b;
                int last = b >= bnext;
                const float* g = L->G + (size_t)r * 4 * d;
                const float* cp = t ? L->C + (size_t)(L->off[t - 1] + b) * d : NULL;
                float* dz = L->dZ + (size_t)r * nz;
                for (int u = 0; u < d; ++u) {
                    float dh = L->dH[(size_t)r * d + u] + (last ? 0.0f : dh_rec[(size_t)b * d + u]);
                    orc_cell_grad cg = orc_lstm_cell_backward(dh, last ? 0.0f : dc_rec[(size_t)b * d + u], g[u], g[d + u],
                                                              g[2 * d + u], g[3 * d + u], L->C[(size_t)r * d + u],
                                                              cp ? cp[u] : 0.0f, coupled);
                    dc_rec[(size_t)b * d + u] = cg.dc_prev;
                    if (coupled) { dz[u] = cg.dz_f; dz[d + u] = cg.dz_g; dz[2 * d + u] = cg.dz_o; }
                    else { dz[u] = cg.dz_i; dz[d + u] = cg.dz_f; dz[2 * d + u] = cg.dz_g; dz[3 * d + u] = cg.dz_o; }
                }
                /* dxh[k] = chain_j fma(dz[j], W[k][j], acc), acc0 = 0 */
                for (int k = 0; k < 2 * d; ++k) dxh[k] = 0.0f;
                for (int j = 0; j < nz; ++j) {
                    float dv = dz[j];
                    const float* w = WT + (size_t)j * 2 * d;
                    for (int k = 0; k < 2 * d; ++k) dxh[k] = fmaf(dv, w[k], dxh[k]);
                }
                for (int k = 0; k < d; ++k) L->dX[(size_t)r * d + k] = dxh[k];
                for (int k = 0; k < d; ++k) dh_rec[(size_t)b * d + k] = dxh[d + k];
            }
        }
        /* dense: dW[k][j] = sum_r xh[r][k] dz[r][j]; row 2d = bias grad = sum_r dz[r][j] */
        size_t nd = (size_t)(2 * d + 1) * nz;
        float* part = (float*)malloc(sizeof(float) * nd);
        for (size_t i = 0; i < nd; ++i) L->dense[i] = 0.0f;
        int nchunks = (L->R + ORC_DW_CHUNK_ROWS - 1) / ORC_DW_CHUNK_ROWS;
        /* row -> (t, b) lookup */
        int* row_t = (int*)malloc(sizeof(int) * (L->R ? L->R : 1));
        for (int t = 0; t < L->Tm; ++t) for (int r = L->off[t]; r < L->off[t + 1]; ++r) row_t[r] = t;
        for (int c = 0; c < nchunks; ++c) {
            int r0 = c * ORC_DW_CHUNK_ROWS, r1 = r0 + ORC_DW_CHUNK_ROWS;
            if (r1 > L->R) r1 = L->R;
            for (size_t i = 0; i < nd; ++i) part[i] = 0.0f;
            for (int r = r0; r < r1; ++r) {
                int t = row_t[r], b = r - L->off[t];
                const float* x = m->E + (size_t)L->in_idx[r] * d;
                const float* hp = t ? L->H + (size_t)(L->off[t - 1] + b) * d : NULL;
                const float* dz = L->dZ + (size_t)r * nz;
                for (int k = 0; k < d; ++k) {
                    float xv = x[k];
                    float* pr = part + (size_t)k * nz;
                    for (int j = 0; j < nz; ++j) pr[j] = fmaf(xv, dz[j], pr[j]);
                }
                for (int k = 0; k < d; ++k) {
                    float hv = hp ? hp[k] : 0.0f;
                    float* pr = part + (size_t)(d + k) * nz;
                    for (int j = 0; j < nz; ++j) pr[j] = fmaf(hv, dz[j], pr[j]);
                }
                float* pb = part + (size_t)2 * d * nz;
                for (int j = 0; j < nz; ++j) pb[j] = pb[j] + dz[j];
            }
            if (c == 0) for (size_t i = 0; i < nd; ++i) L->dense[i] = part[i];
            else for (size_t i = 0; i < nd; ++i) L->dense[i] = L->dense[i] + part[i];
        }
        free(row_t); free(part); free(WT); free(dxh); free(dh_rec); free(dc_rec);
    } else {
        /* EWMA: ds_t = dH_t + a*ds_{t+1}; dX_t = (1-a)*ds_t (t>0) / ds_0; per-sequence partial
         * da_b = chain over t descending of fma(ds_t, s_{t-1} - x_t, .); sequences reduced in
         * chunks of ORC_EWMA_CHUNK_SEQS (chain inside, chain across); times a(1-a). */
        float* carry = (float*)calloc((size_t)L->B * d, 4);
        float* dab = (float*)calloc((size_t)L->B * d, 4);
        float* av = (float*)malloc(sizeof(float) * d);
        for (int k = 0; k < d; ++k) av[k] = orc_sigmoid(m->alpha[k]);
        for (int t = L->Tm - 1; t >= 0; --t) {
            int bt = L->off[t + 1] - L->off[t];
            int bnext = t + 1 < L->Tm ? L->off[t + 2] - L->off[t + 1] : 0;
            for (int b = 0; b < bt; ++b) {
                int r = L->off[t] + b;
                int last = b >= bnext;
                const float* x = m->E + (size_t)L->in_idx[r] * d;
                const float* sp = t ? L->H + (size_t)(L->off[t - 1] + b) * d : NULL;
                for (int k = 0; k < d; ++k) {
                    float ds = L->dH[(size_t)r * d + k] + (last ? 0.0f : carry[(size_t)b * d + k]);
                    if (t > 0) {
                        float a = av[k], oma = 1.0f - a;
                        L->dX[(size_t)r * d + k] = oma * ds;
                        carry[(size_t)b * d + k] = a * ds;
                        dab[(size_t)b * d + k] = fmaf(ds, sp[k] - x[k], dab[(size_t)b * d + k]);
                    } else {
                        L->dX[(size_t)r * d + k] = ds;
                    }
                }
            }
        }
        int nchunks = (L->B + ORC_EWMA_CHUNK_SEQS - 1) / ORC_EWMA_CHUNK_SEQS;
        for (int k = 0; k < d; ++k) {
            float tot = 0.0f;
            for (int c = 0; c < nchunks; ++c) {
                int b0 = c * ORC_EWMA_CHUNK_SEQS, b1 = b0 + ORC_EWMA_CHUNK_SEQS;
                if (b1 > L->B) b1 = L->B;
                float pc = 0.0f;
                for (int b = b0; b < b1; ++b) pc = pc + dab[(size_t)b * d + k];
                tot = c == 0 ? pc : tot + pc;
            }
            float a = av[k];
            L->dense[k] = tot * (a * (1.0f - a));
        }
        free(carry); free(dab); free(av);
    }
}

int orc_fit_step_local(orc_plan* p, int q, uint64_t mb) {
    orc_local* L = &p->loc[q];
    orc_pack(p, q, mb, L);
    orc_forward(p->m, L);
    orc_score(p->m, L, orc_epoch_key_of(p->fit_seed[q], p->epoch_key_epoch), &p->part_rng[q]);
    orc_lagged_loss_update(p, q);
    orc_backward(p->m, L);
    return SBR_OK;
}

static uint64_t orc_ndense(const orc_model* m);
int orc_fit_step_local_sample(orc_plan* p, int q, uint64_t mb, const uint32_t* sel_b, uint32_t nsel, uint32_t* out_full_rows,
                              uint32_t* out_nrows, uint64_t* out_off_full) {
    if (!p || !sel_b || !out_full_rows || !out_nrows || nsel == 0) return SBR_ERR_INVALID_ARGUMENT;
    orc_local* L = &p->loc[q];
    int st = orc_pack_sample(p, q, mb, sel_b, nsel, L, out_full_rows, out_off_full);
    if (st != SBR_OK) return st;
    orc_forward(p->m, L);
    orc_score(p->m, L, orc_epoch_key_of(p->fit_seed[q], p->epoch_key_epoch), &p->part_rng[q]);
    orc_backward(p->m, L); /* L->dense is the sample's own dense gradient, not the minibatch's */
    *out_nrows = (uint32_t)L->R;
    return SBR_OK;
}

/* One item-table row's optimiser step from an explicit entry list, in the contract's order (orc_reduce_row + orc_row_update):
 * entry e contributes scale[e] * vecs[e][0..d) (input row: scale 1 and dX; target row: -coef and h; negative row: +coef and h)
 * and, if has_bias[e], scale[e] to the bias gradient; entries in (packed row, kind) order; chunks of ORC_SEG_CHUNK.  Updates
 * w / acc / *b / *bacc in place with the model's hyper-parameters (Adagrad).  For the sampled update check at bench size. */
int orc_row_step(orc_model* m, uint32_t n, const float* vecs, const float* scale, const uint8_t* has_bias, float* w, float* acc,
                 float* b, float* bacc) {
    if (!m || n == 0 || m->hp.optimizer != SBR_OPT_ADAGRAD) return SBR_ERR_INVALID_ARGUMENT;
    int d = m->d;
    float* g = (float*)malloc(sizeof(float) * d);
    float* part = (float*)malloc(sizeof(float) * d);
    float gb = 0.0f; int has_b = 0, first_chunk = 1;
    for (uint32_t c0 = 0; c0 < n; c0 += ORC_SEG_CHUNK) {
        uint32_t c1 = c0 + ORC_SEG_CHUNK < n ? c0 + ORC_SEG_CHUNK : n;
        float pb = 0.0f; int phb = 0, first = 1;
        for (uint32_t e = c0; e < c1; ++e) {
            const float* srcv = vecs + (size_t)e * d;
            float sc = scale[e];
            if (first) { for (int k = 0; k < d; ++k) part[k] = sc * srcv[k]; first = 0; }
            else for (int k = 0; k < d; ++k) part[k] = part[k] + sc * srcv[k];
            if (has_bias[e]) { pb = phb ? pb + sc : sc; phb = 1; }
        }
        if (first_chunk) { for (int k = 0; k < d; ++k) g[k] = part[k]; first_chunk = 0; }
        else for (int k = 0; k < d; ++k) g[k] = g[k] + part[k];
        if (phb) { gb = has_b ? gb + pb : pb; has_b = 1; }
    }
    float dummy = 0.0f;
    for (int k = 0; k < d; ++k) orc_opt(m, &w[k], &acc[k], &dummy, g[k]);
    if (has_b) orc_opt(m, b, bacc, &dummy, gb);
    free(g); free(part);
    return SBR_OK;
}

/* The two halves of orc_row_step, for a row touched by SEVERAL devices (the multi-device optimiser step: every device reduces its
 * own entries — orc_fit_scatter —, the owner adds the devices' sums in device order — orc_fit_owner_reduce: the first toucher
 * initialises —, every replica applies one update — orc_fit_apply_table): orc_row_reduce = one device's chunked in-order sum,
 * orc_row_apply = the optimiser update from an explicit gradient. */
int orc_row_reduce(uint32_t d, uint32_t n, const float* vecs, const float* scale, const uint8_t* has_bias, float* g, float* gb,
                   int32_t* has_b_out) {
    if (n == 0 || !g || !gb || !has_b_out) return SBR_ERR_INVALID_ARGUMENT;
    float* part = (float*)malloc(sizeof(float) * d);
    float gbv = 0.0f; int has_b = 0, first_chunk = 1;
    for (uint32_t c0 = 0; c0 < n; c0 += ORC_SEG_CHUNK) {
        uint32_t c1 = c0 + ORC_SEG_CHUNK < n ? c0 + ORC_SEG_CHUNK : n;
        float pb = 0.0f; int phb = 0, first = 1;
        for (uint32_t e = c0; e < c1; ++e) {
            const float* srcv = vecs + (size_t)e * d;
            float sc = scale[e];
            if (first) { for (uint32_t k = 0; k < d; ++k) part[k] = sc * srcv[k]; first = 0; }
            else for (uint32_t k = 0; k < d; ++k) part[k] = part[k] + sc * srcv[k];
            if (has_bias[e]) { pb = phb ? pb + sc : sc; phb = 1; }
        }
        if (first_chunk) { for (uint32_t k = 0; k < d; ++k) g[k] = part[k]; first_chunk = 0; }
        else for (uint32_t k = 0; k < d; ++k) g[k] = g[k] + part[k];
        if (phb) { gbv = has_b ? gbv + pb : pb; has_b = 1; }
    }
    free(part);
    *gb = gbv; *has_b_out = has_b;
    return SBR_OK;
}
int orc_row_apply(orc_model* m, const float* g, float gb, int32_t has_b, float* w, float* acc, float* b, float* bacc) {
    if (!m || m->hp.optimizer != SBR_OPT_ADAGRAD) return SBR_ERR_INVALID_ARGUMENT;
    float dummy = 0.0f;
    for (int k = 0; k < m->d; ++k) orc_opt(m, &w[k], &acc[k], &dummy, g[k]);
    if (has_b) orc_opt(m, b, bacc, &dummy, gb);
    return SBR_OK;
}

/* One element of the dense gradient from its two operand columns over ALL packed rows, in the contract's order (orc_backward):
 * rows in chunks of ORC_DW_CHUNK_ROWS, inside a chunk a row-ascending fma chain from 0 (a == NULL: the bias row, a plain add
 * chain), chunk partials added in chunk order. */
float orc_dense_chain(const float* a, const float* dz, uint64_t rows) {
    float total = 0.0f;
    for (uint64_t r0 = 0, c = 0; r0 < rows; r0 += ORC_DW_CHUNK_ROWS, ++c) {
        uint64_t r1 = r0 + ORC_DW_CHUNK_ROWS < rows ? r0 + ORC_DW_CHUNK_ROWS : rows;
        float part = 0.0f;
        if (a) for (uint64_t r = r0; r < r1; ++r) part = fmaf(a[r], dz[r], part);
        else for (uint64_t r = r0; r < r1; ++r) part = part + dz[r];
        total = c == 0 ? part : total + part;
    }
    return total;
}

/* The dense half of an optimiser step from an explicit gradient block (orc_begin_optimizer_step + orc_dense_update) */
int orc_model_apply_dense(orc_model* m, const float* dense, uint64_t count) {
    if (!m || !dense || count != orc_ndense(m)) return SBR_ERR_INVALID_ARGUMENT;
    orc_begin_optimizer_step(m);
    orc_dense_update(m, dense);
    return SBR_OK;
}

/* ---- exchange block: what one device contributes to an optimiser step --------------------- */
/* layout (all 4-byte words unless noted), Rmax = batch_sequences*(T-1):
 *   [0]      u32 R
 *   [1..3]   pad
 *   [4..5]   f64 loss_sum     [6..7] u64 examples
 *   in_idx[Rmax] out_idx[Rmax] neg[Rmax] coef[Rmax]  H[Rmax*d] dX[Rmax*d]  dense[ndense]  */
static uint64_t orc_ndense(const orc_model* m) {
    return m->ng ? (uint64_t)(2 * m->d + 1) * m->ng * m->d : (uint64_t)m->d;
}
uint64_t orc_fit_exchange_bytes(orc_plan* p) {
    uint64_t R = (uint64_t)p->Rmax, d = (uint64_t)p->m->d;
    uint64_t words = 8 + 4 * R + 2 * R * d + orc_ndense(p->m);
    return ((words * 4 + 15) / 16) * 16;
}
int orc_fit_export_local(orc_plan* p, int q, void* out) {
    orc_local* L = &p->loc[q];
    uint64_t R = (uint64_t)p->Rmax, d = (uint64_t)p->m->d;
    uint32_t* w = (uint32_t*)out;
    memset(out, 0, orc_fit_exchange_bytes(p));
    w[0] = (uint32_t)L->R;
    memcpy(w + 4, &L->loss_sum, 8);
    memcpy(w + 6, &L->examples, 8);
    uint32_t* q0 = w + 8;
    memcpy(q0, L->in_idx, (size_t)L->R * 4); q0 += R;
    memcpy(q0, L->out_idx, (size_t)L->R * 4); q0 += R;
    memcpy(q0, L->neg, (size_t)L->R * 4); q0 += R;
    memcpy(q0, L->coef, (size_t)L->R * 4); q0 += R;
    memcpy(q0, L->H, (size_t)L->R * d * 4); q0 += R * d;
    memcpy(q0, L->dX, (size_t)L->R * d * 4); q0 += R * d;
    memcpy(q0, L->dense, orc_ndense(p->m) * 4);
    return SBR_OK;
}

typedef struct { uint32_t row; uint32_t src; } orc_entry; /* src = device*3*Rmax + 3*r + kind */
static int orc_entry_cmp(const void* a, const void* b) {
    const orc_entry* x = (const orc_entry*)a; const orc_entry* y = (const orc_entry*)b;
    if (x->row != y->row) return x->row < y->row ? -1 : 1;
    if (x->src != y->src) return x->src < y->src ? -1 : 1;
    return 0;
}

/* One row's sparse gradient from its sorted entries ent[i..j): chunked in-order reduction, see
 * ORC_SEG_CHUNK in orc_numerics.h.  Every entry is (source vector, scale, bias flag): input row -> dX,
 * target row -> -g*h, negative row -> +g*h; target/negative entries also carry the bias gradient
 * (= scale).  `part` is scratch for one chunk partial [d]. */
typedef struct { const float* H; const float* dX; const float* coef; } orc_entry_src;
static void orc_reduce_row(const orc_entry* ent, uint64_t i, uint64_t j, int d, const orc_entry_src* s, uint32_t src_mod,
                           float* g, float* part, float* gb_out, int* has_b_out) {
    float gb = 0.0f; int has_b = 0, first_chunk = 1;
    for (uint64_t c0 = i; c0 < j; c0 += ORC_SEG_CHUNK) {
        uint64_t c1 = c0 + ORC_SEG_CHUNK < j ? c0 + ORC_SEG_CHUNK : j;
        float pb = 0.0f; int phb = 0, first = 1;
        for (uint64_t e = c0; e < c1; ++e) {
            uint32_t src = src_mod ? ent[e].src % src_mod : ent[e].src;
            uint32_t r = src / 3, kind = src % 3;
            const float* srcv = kind == 0 ? s->dX + (size_t)r * d : s->H + (size_t)r * d;
            float scale = kind == 0 ? 1.0f : kind == 1 ? -s->coef[r] : s->coef[r];
            if (first) { for (int k = 0; k < d; ++k) part[k] = scale * srcv[k]; first = 0; }
            else for (int k = 0; k < d; ++k) part[k] = part[k] + scale * srcv[k];
            if (kind != 0) { pb = phb ? pb + scale : scale; phb = 1; }
        }
        if (first_chunk) { for (int k = 0; k < d; ++k) g[k] = part[k]; first_chunk = 0; }
        else for (int k = 0; k < d; ++k) g[k] = g[k] + part[k];
        if (phb) { gb = has_b ? gb + pb : pb; has_b = 1; }
    }
    *gb_out = gb; *has_b_out = has_b;
}

/* Optimiser step from the gathered exchange blocks of all devices
 * (≙ optimizer.step / sync_optim.step, sequence_model.rs:163-169; wyrm Adagrad as recalled):
 *  dense: gradients of the devices added in device order, then Adagrad on every element;
 *  sparse: every (row, source) entry — input row with dX, target row with -g*h, negative row
 *  with +g*h — sorted by (row, device, packed row, kind); duplicates added in that order; one
 *  Adagrad update (with L2) per touched row, also for rows whose summed data-gradient is zero
 *  (SURVEY App. A-14); biases likewise for target/negative rows. */
/* one optimiser step from ONE worker's gradients: its dense gradient, then its sparse entries per row */
/* per-thread scratch of orc_apply_gradients, grown on demand and kept: the timed baseline (orc_fit_threads) takes one optimiser
 * step per ~30-row subsequence on every worker, and four malloc / free pairs per step are serialisation the reference does not have */
static __thread float* orc_tls_dense = NULL;
static __thread uint64_t orc_tls_dense_cap = 0;
static __thread orc_entry* orc_tls_ent = NULL;
static __thread uint64_t orc_tls_ent_cap = 0;
static __thread float* orc_tls_row = NULL; /* gsum | part */
static __thread int orc_tls_row_cap = 0;

static void orc_apply_gradients(orc_model* m, uint32_t R, const uint32_t* in_idx, const uint32_t* out_idx, const uint32_t* neg,
                                const orc_entry_src* es, const float* dense) {
    int d = m->d;
    uint64_t nd = orc_ndense(m);
    orc_begin_optimizer_step(m);
    if (nd > orc_tls_dense_cap) { free(orc_tls_dense); orc_tls_dense = (float*)malloc(nd * 4); orc_tls_dense_cap = nd; }
    float* dg = orc_tls_dense;
    memcpy(dg, dense, nd * 4);
    orc_dense_update(m, dg);
    uint64_t ne = 3ull * R;
    if (ne + 1 > orc_tls_ent_cap) { free(orc_tls_ent); orc_tls_ent_cap = 2 * (ne + 1); orc_tls_ent = (orc_entry*)malloc(sizeof(orc_entry) * orc_tls_ent_cap); }
    orc_entry* ent = orc_tls_ent;
    for (uint32_t r = 0; r < R; ++r) {
        ent[3 * r].row = in_idx[r]; ent[3 * r].src = 3u * r;
        ent[3 * r + 1].row = out_idx[r]; ent[3 * r + 1].src = 3u * r + 1;
        ent[3 * r + 2].row = neg[r]; ent[3 * r + 2].src = 3u * r + 2;
    }
    qsort(ent, ne, sizeof(orc_entry), orc_entry_cmp);
    if (d > orc_tls_row_cap) { free(orc_tls_row); orc_tls_row = (float*)malloc(sizeof(float) * 2 * d); orc_tls_row_cap = d; }
    float* gsum = orc_tls_row;
    float* part = orc_tls_row + d;
    uint64_t i = 0;
    while (i < ne) {
        uint32_t row = ent[i].row;
        uint64_t j = i;
        while (j < ne && ent[j].row == row) ++j;
        float gb; int has_b;
        orc_reduce_row(ent, i, j, d, es, 0, gsum, part, &gb, &has_b);
        orc_row_update(m, row, gsum, 1, has_b, gb);
        i = j;
    }
}

static int orc_apply_own_block(orc_plan* p, const void* block, int q) {
    /* one optimiser step from ONE device's exported block */
    orc_model* m = p->m;
    int d = m->d;
    uint64_t Rmax = (uint64_t)p->Rmax;
    const uint32_t* w = (const uint32_t*)block;
    const float* dense = (const float*)(w + 8 + 4 * Rmax + 2 * Rmax * (uint64_t)d);
    double ls; uint64_t ex;
    memcpy(&ls, w + 4, 8); memcpy(&ex, w + 6, 8);
    p->loss_sum += ls; p->examples += ex;
    p->loss_dev[q] += ls; p->examples_dev[q] += ex;
    const uint32_t* in_idx = w + 8; const uint32_t* out_idx = in_idx + Rmax; const uint32_t* neg = out_idx + Rmax;
    orc_entry_src es = { (const float*)(w + 8 + 4 * Rmax), (const float*)(w + 8 + 4 * Rmax) + Rmax * (uint64_t)d,
                         (const float*)(w + 8 + 3 * Rmax) };
    orc_apply_gradients(m, w[0], in_idx, out_idx, neg, &es, dense);
    return SBR_OK;
}

int orc_fit_step_apply(orc_plan* p, const void* all_blocks) {
    if (p->ndev != 1) return SBR_ERR_INVALID_ARGUMENT; /* multi-device: owner-reduce protocol below */
    return orc_apply_own_block(p, all_blocks, 0);
}

/* ---- multi-device optimiser step: owner-reduce protocol ---------------------------------------
 * (≙ the rendezvous of Parallelism::Synchronous, sequence_model.rs:163-166)
 * Rows of the item table are owned in contiguous slices of S = ceil(I / ndev) rows.  Per step:
 *   scatter      : device q reduces ITS OWN entries per row — sorted by (row, packed row, kind),
 *                  duplicates added in that order — into a dense send buffer of ndev chunks
 *                  (chunk p = the rows owned by device p): [G: S*d f32][gb: S f32][flags: S u32],
 *                  flags bit0 = embedding row touched, bit1 = bias touched.
 *   all-to-all   : chunk p of every device goes to device p.
 *   owner_reduce : the owner adds the devices' contributions in device order (first toucher
 *                  initialises, later ones add) -> one chunk of global sums for its rows.
 *   all-gather   : of the owners' chunks (= global gradient sums of the whole table) and of the
 *                  small dense blocks [8-word header | dense grads].
 *   apply_table  : every device applies the identical Adagrad update to every touched row and
 *                  to the dense parameters (device-order sum), so replicas stay bit-identical.
 * With one device this is exactly orc_fit_step_apply's flat order. */
static uint64_t orc_slice_rows(const orc_plan* p) { return ((uint64_t)p->m->hp.num_items + p->ndev - 1) / p->ndev; }
uint64_t orc_fit_chunk_bytes(orc_plan* p) { return orc_slice_rows(p) * ((uint64_t)p->m->d + 2) * 4; }
uint64_t orc_fit_dense_bytes(orc_plan* p) { return (8 + orc_ndense(p->m)) * 4; }

int orc_fit_export_dense(orc_plan* p, int q, void* out) {
    orc_local* L = &p->loc[q];
    uint32_t* w = (uint32_t*)out;
    memset(w, 0, 32);
    w[0] = (uint32_t)L->R;
    memcpy(w + 4, &L->loss_sum, 8);
    memcpy(w + 6, &L->examples, 8);
    memcpy(w + 8, L->dense, orc_ndense(p->m) * 4);
    return SBR_OK;
}

int orc_fit_scatter(orc_plan* p, int q, void* send) {
    orc_model* m = p->m;
    orc_local* L = &p->loc[q];
    int d = m->d;
    uint64_t S = orc_slice_rows(p), cw = S * ((uint64_t)d + 2);
    memset(send, 0, (size_t)p->ndev * cw * 4);
    uint64_t ne = 3ull * L->R;
    orc_entry* ent = (orc_entry*)malloc(sizeof(orc_entry) * (ne ? ne : 1));
    for (int r = 0; r < L->R; ++r) {
        ent[3 * r].row = L->in_idx[r]; ent[3 * r].src = 3u * r;
        ent[3 * r + 1].row = L->out_idx[r]; ent[3 * r + 1].src = 3u * r + 1;
        ent[3 * r + 2].row = L->neg[r]; ent[3 * r + 2].src = 3u * r + 2;
    }
    qsort(ent, ne, sizeof(orc_entry), orc_entry_cmp);
    orc_entry_src es = { L->H, L->dX, L->coef };
    float* part = (float*)malloc(sizeof(float) * d);
    uint64_t i = 0;
    while (i < ne) {
        uint32_t row = ent[i].row;
        uint64_t owner = row / S, lr = row % S;
        float* chunk = (float*)send + owner * cw;
        float* g = chunk + lr * d;
        float* gb = chunk + S * d + lr;
        uint32_t* fl = (uint32_t*)(chunk + S * d + S) + lr;
        uint64_t j = i;
        while (j < ne && ent[j].row == row) ++j;
        int has_b; float gbv;
        orc_reduce_row(ent, i, j, d, &es, 0, g, part, &gbv, &has_b);
        if (has_b) *gb = gbv;
        *fl = 1u | (has_b ? 2u : 0u);
        i = j;
    }
    free(ent); free(part);
    return SBR_OK;
}

int orc_fit_owner_reduce(orc_plan* p, const void* recv, void* own_chunk) {
    int d = p->m->d;
    uint64_t S = orc_slice_rows(p), cw = S * ((uint64_t)d + 2);
    float* out = (float*)own_chunk;
    memset(out, 0, cw * 4);
    uint32_t* ofl = (uint32_t*)(out + S * d + S);
    for (int q = 0; q < p->ndev; ++q) {
        const float* c = (const float*)recv + (size_t)q * cw;
        const uint32_t* fl = (const uint32_t*)(c + S * d + S);
        for (uint64_t i = 0; i < S; ++i) {
            if (fl[i] & 1u) {
                float* g = out + i * d;
                const float* s = c + i * d;
                if (ofl[i] & 1u) for (int k = 0; k < d; ++k) g[k] = g[k] + s[k];
                else for (int k = 0; k < d; ++k) g[k] = s[k];
            }
            if (fl[i] & 2u) {
                float* gb = out + S * d + i;
                if (ofl[i] & 2u) *gb = *gb + c[S * d + i]; else *gb = c[S * d + i];
            }
            ofl[i] |= fl[i];
        }
    }
    return SBR_OK;
}

int orc_fit_apply_table(orc_plan* p, const void* all_chunks, const void* dense_all) {
    orc_model* m = p->m;
    int d = m->d, ndev = p->ndev;
    uint64_t S = orc_slice_rows(p), cw = S * ((uint64_t)d + 2), nd = orc_ndense(m), I = m->hp.num_items;
    orc_begin_optimizer_step(m);
    float* dg = (float*)malloc(nd * 4);
    for (int q = 0; q < ndev; ++q) {
        const uint32_t* w = (const uint32_t*)dense_all + (size_t)q * (8 + nd);
        const float* dense = (const float*)(w + 8);
        if (q == 0) memcpy(dg, dense, nd * 4); else for (uint64_t i = 0; i < nd; ++i) dg[i] = dg[i] + dense[i];
        double ls; uint64_t ex;
        memcpy(&ls, w + 4, 8); memcpy(&ex, w + 6, 8);
        p->loss_sum += ls; p->examples += ex;
        p->loss_dev[q] += ls; p->examples_dev[q] += ex;
    }
    orc_dense_update(m, dg);
    free(dg);
    for (uint64_t row = 0; row < I; ++row) {
        const float* c = (const float*)all_chunks + (row / S) * cw;
        uint64_t lr_ = row % S;
        uint32_t fl = ((const uint32_t*)(c + S * d + S))[lr_];
        orc_row_update(m, row, c + lr_ * d, (fl & 1u) != 0, (fl & 2u) != 0, c[S * d + lr_]);
    }
    return SBR_OK;
}

/* The owner-APPLIED form of the same step (include/sbr_hip.h: sbr_fit_step_owner_update; the engine's Synchronous step since
 * round 6).  Rank `rank` adds the devices' contributions to ITS slice in device order and applies the one optimiser update of
 * every touched row of the slice to its own replica; the updated parameter slices (and, when a fit ends, the optimiser-state
 * slices) then travel instead of the gradient sums.  orc_fit_apply_table above does the same arithmetic for every slice on every
 * replica: same bits — which is what tests/test_distributed_cpu.py checks by running BOTH forms against the one-process emulation.
 * Checker-side halves of a multi-process run only (one orc_model per rank). */
int orc_fit_owner_update(orc_plan* p, int rank, const void* recv) {
    orc_model* m = p->m;
    int d = m->d;
    uint64_t S = orc_slice_rows(p), cw = S * ((uint64_t)d + 2), I = m->hp.num_items;
    uint64_t row0 = (uint64_t)rank * S;
    if (rank < 0 || rank >= p->ndev) return SBR_ERR_INVALID_ARGUMENT;
    orc_begin_optimizer_step(m);
    float* g = (float*)malloc(sizeof(float) * d);
    for (uint64_t i = 0; i < S && row0 + i < I; ++i) {
        uint32_t fl = 0;
        float gb = 0.0f;
        for (int q = 0; q < p->ndev; ++q) { /* device order; the first toucher initialises (orc_fit_owner_reduce) */
            const float* c = (const float*)recv + (size_t)q * cw;
            uint32_t f = ((const uint32_t*)(c + S * d + S))[i];
            if (f & 1u) {
                const float* s = c + i * d;
                if (fl & 1u) for (int k = 0; k < d; ++k) g[k] = g[k] + s[k];
                else for (int k = 0; k < d; ++k) g[k] = s[k];
            }
            if (f & 2u) gb = (fl & 2u) ? gb + c[S * d + i] : c[S * d + i];
            fl |= f;
        }
        orc_row_update(m, row0 + i, g, (fl & 1u) != 0, (fl & 2u) != 0, gb);
    }
    free(g);
    return SBR_OK;
}

/* the dense half of orc_fit_apply_table alone (the optimiser step was opened by orc_fit_owner_update) */
int orc_fit_apply_dense_blocks(orc_plan* p, const void* dense_all) {
    orc_model* m = p->m;
    int ndev = p->ndev;
    uint64_t nd = orc_ndense(m);
    float* dg = (float*)malloc(nd * 4);
    for (int q = 0; q < ndev; ++q) {
        const uint32_t* w = (const uint32_t*)dense_all + (size_t)q * (8 + nd);
        const float* dense = (const float*)(w + 8);
        if (q == 0) memcpy(dg, dense, nd * 4); else for (uint64_t i = 0; i < nd; ++i) dg[i] = dg[i] + dense[i];
        double ls; uint64_t ex;
        memcpy(&ls, w + 4, 8); memcpy(&ex, w + 6, 8);
        p->loss_sum += ls; p->examples += ex;
        p->loss_dev[q] += ls; p->examples_dev[q] += ex;
    }
    orc_dense_update(m, dg);
    free(dg);
    return SBR_OK;
}

/* slice `rank` (rows [rank * S, (rank + 1) * S), zero-padded past num_items) of an item-table block in stored layout: what a
 * rank contributes to / receives from the all-gather of the parameter (or optimiser-state) slices */
int orc_model_table_slice_bytes(orc_model* m, int which, uint64_t* out) {
    uint64_t n = m->hp.num_devices, S = ((uint64_t)m->hp.num_items + n - 1) / n;
    int row = which == SBR_PARAM_ITEM_EMBEDDING || which == SBR_PARAM_ITEM_EMBEDDING_ACC || which == SBR_PARAM_ITEM_EMBEDDING_M;
    *out = S * (row ? (uint64_t)m->d : 1) * 4;
    return SBR_OK;
}
static float* orc_table_block(orc_model* m, int which, uint64_t* width) {
    *width = 1;
    switch (which) {
        case SBR_PARAM_ITEM_EMBEDDING: *width = (uint64_t)m->d; return m->E;
        case SBR_PARAM_ITEM_EMBEDDING_ACC: *width = (uint64_t)m->d; return m->Eacc;
        case SBR_PARAM_ITEM_EMBEDDING_M: *width = (uint64_t)m->d; return m->Em;
        case SBR_PARAM_ITEM_BIAS: return m->b;
        case SBR_PARAM_ITEM_BIAS_ACC: return m->bacc;
        case SBR_PARAM_ITEM_BIAS_M: return m->bm;
    }
    return NULL;
}
int orc_model_get_table_slice(orc_model* m, int which, int rank, float* out) {
    uint64_t w, n = m->hp.num_devices, I = m->hp.num_items, S = (I + n - 1) / n;
    float* p = orc_table_block(m, which, &w);
    if (!p || rank < 0 || (uint64_t)rank >= n) return SBR_ERR_INVALID_ARGUMENT;
    uint64_t r0 = (uint64_t)rank * S, r1 = r0 + S < I ? r0 + S : I;
    memset(out, 0, S * w * 4);
    if (r1 > r0) memcpy(out, p + r0 * w, (r1 - r0) * w * 4);
    return SBR_OK;
}
int orc_model_set_table_slice(orc_model* m, int which, int rank, const float* in) {
    uint64_t w, n = m->hp.num_devices, I = m->hp.num_items, S = (I + n - 1) / n;
    float* p = orc_table_block(m, which, &w);
    if (!p || rank < 0 || (uint64_t)rank >= n) return SBR_ERR_INVALID_ARGUMENT;
    uint64_t r0 = (uint64_t)rank * S, r1 = r0 + S < I ? r0 + S : I;
    if (r1 > r0) memcpy(p + r0 * w, in, (r1 - r0) * w * 4);
    return SBR_OK;
}

/* One full optimiser step, all devices emulated in this process. */
int orc_fit_step(orc_plan* p, uint64_t mb) {
    if (p->ndev == 1) {
        uint64_t bytes = orc_fit_exchange_bytes(p);
        char* blk = (char*)malloc(bytes);
        orc_fit_step_local(p, 0, mb);
        orc_fit_export_local(p, 0, blk);
        int st = orc_fit_step_apply(p, blk);
        free(blk);
        return st;
    }
    int n = p->ndev;
    if (p->m->reference_order) {
        /* every worker's forward / backward against the parameters as they stand at the rendezvous, then the updates one
         * worker at a time (device order stands in for the reference's arrival order) */
        uint64_t bytes = orc_fit_exchange_bytes(p);
        char* blk = (char*)malloc((size_t)n * bytes);
        for (int q = 0; q < n; ++q) {
            orc_fit_step_local(p, q, mb);
            orc_fit_export_local(p, q, blk + (size_t)q * bytes);
        }
        int st = SBR_OK;
        for (int q = 0; q < n && st == SBR_OK; ++q) st = orc_apply_own_block(p, blk + (size_t)q * bytes, q);
        free(blk);
        return st;
    }
    uint64_t cb = orc_fit_chunk_bytes(p), db = orc_fit_dense_bytes(p);
    char* send = (char*)malloc((size_t)n * n * cb); /* [device][chunk] */
    char* recv = (char*)malloc((size_t)n * cb);
    char* all = (char*)malloc((size_t)n * cb);
    char* dense = (char*)malloc((size_t)n * db);
    for (int q = 0; q < n; ++q) {
        orc_fit_step_local(p, q, mb);
        orc_fit_scatter(p, q, send + (size_t)q * n * cb);
        orc_fit_export_dense(p, q, dense + (size_t)q * db);
    }
    for (int owner = 0; owner < n; ++owner) {
        for (int q = 0; q < n; ++q) memcpy(recv + (size_t)q * cb, send + ((size_t)q * n + owner) * cb, cb); /* all-to-all */
        orc_fit_owner_reduce(p, recv, all + (size_t)owner * cb);                                            /* + all-gather */
    }
    int st = orc_fit_apply_table(p, all, dense);
    free(send); free(recv); free(all); free(dense);
    return st;
}

/* Parallelism::Asynchronous (mod.rs:36-38) with more than one device.  The reference's Hogwild
 * workers read parameters their peers are still updating, so their gradients are stale by an
 * unpredictable amount.  The engine's deterministic analogue fixes the staleness at exactly one
 * step: every device computes minibatch k+1 on parameters that lack update k, which lets the
 * exchange of step k run underneath that computation.  One epoch, all devices emulated here;
 * with a single device there is nobody to be asynchronous with and the step is the synchronous one. */
int orc_fit_epoch_async(orc_plan* p, uint64_t nmb) {
    int n = p->ndev;
    if (n < 2 || nmb == 0) return SBR_ERR_INVALID_ARGUMENT;
    uint64_t cb = orc_fit_chunk_bytes(p), db = orc_fit_dense_bytes(p);
    char* send = (char*)malloc((size_t)n * n * cb);
    char* recv = (char*)malloc((size_t)n * cb);
    char* all = (char*)malloc((size_t)n * cb);
    char* dense = (char*)malloc((size_t)n * db);
    int st = SBR_OK;
    for (int q = 0; q < n; ++q) orc_fit_step_local(p, q, 0);
    for (uint64_t mb = 0; mb < nmb && st == SBR_OK; ++mb) {
        for (int q = 0; q < n; ++q) {
            orc_fit_scatter(p, q, send + (size_t)q * n * cb);
            orc_fit_export_dense(p, q, dense + (size_t)q * db);
        }
        if (mb + 1 < nmb)
            for (int q = 0; q < n; ++q) orc_fit_step_local(p, q, mb + 1); /* before update mb lands */
        for (int owner = 0; owner < n; ++owner) {
            for (int q = 0; q < n; ++q) memcpy(recv + (size_t)q * cb, send + ((size_t)q * n + owner) * cb, cb);
            orc_fit_owner_reduce(p, recv, all + (size_t)owner * cb);
        }
        st = orc_fit_apply_table(p, all, dense);
    }
    free(send); free(recv); free(all); free(dense);
    return st;
}

/* ---- The reference's PARALLEL SHAPE on the CPU: bench.py's cpu_baseline leg only ------------------------------------------
 * sequence_model.rs:90-102: the shuffled subsequences are cut into num_threads partitions (:91-98), one rayon worker per
 * partition (:99-102), ALL workers on ONE shared parameter set (Arc<HogwildParameter>, lstm.rs:175-181), one optimiser step per
 * batch_sequences subsequences (1 = the reference's schedule, :111-169); default num_threads = all cores (lstm.rs:68).
 *   synchronous = 0  Parallelism::Asynchronous (mod.rs:36-38): Hogwild — every worker writes its update into the shared
 *                    parameters the moment it has it, no locks (the data races are the algorithm, as in the reference);
 *   synchronous = 1  Parallelism::Synchronous (mod.rs:39-40; sequence_model.rs:163-166): wyrm's SynchronizedOptimizer as
 *                    recalled — the live workers rendezvous, each update goes in under exclusion, then all
 *                    are released.
 * Thread timing decides the order of the updates, so this mode is NOT deterministic and no parity test uses it: it is the
 * timed CPU baseline beside the GPU number (one shared ~1 GB table instead of round 3's private tables).  Runs the model's
 * num_epochs epochs, or until max_seconds of wall time have passed (checked between steps; <= 0: no limit).  `workers` may
 * exceed the 16 devices the parity paths are built for.  Negatives: the contract's counter-keyed draws (orc_score). */
#include <pthread.h>
#include <time.h>

typedef struct {
    orc_plan* p;
    int q, synchronous;
    uint64_t nmb;
    double t_start, max_seconds;
    pthread_barrier_t* bar;
    pthread_mutex_t* mu;
    volatile int* stop;
    uint64_t rows_done;
    double loss_done;
} orc_worker;

static double orc_now(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

static void* orc_worker_main(void* arg) {
    orc_worker* w = (orc_worker*)arg;
    orc_plan* p = w->p;
    orc_model* m = p->m;
    orc_local* L = &p->loc[w->q];
    for (uint64_t mb = 0; mb < w->nmb; ++mb) {
        if (!w->synchronous && w->max_seconds > 0 && orc_now() - w->t_start > w->max_seconds) break;
        orc_pack(p, w->q, mb, L);
        orc_forward(m, L);
        orc_score(m, L, orc_epoch_key_of(p->fit_seed[w->q], p->epoch_key_epoch), &p->part_rng[w->q]);
        orc_backward(m, L);
        orc_entry_src es = { L->H, L->dX, L->coef };
        if (w->synchronous) {
            pthread_barrier_wait(w->bar);                 /* sync_optim.step(): wait for every live worker ... */
            pthread_mutex_lock(w->mu);                    /* ... the updates go in one at a time ... */
            orc_apply_gradients(m, (uint32_t)L->R, L->in_idx, L->out_idx, L->neg, &es, L->dense);
            pthread_mutex_unlock(w->mu);
            if (w->q == 0 && w->max_seconds > 0 && orc_now() - w->t_start > w->max_seconds) *w->stop = 1;
            pthread_barrier_wait(w->bar);                 /* ... and everybody is released */
        } else {
            orc_apply_gradients(m, (uint32_t)L->R, L->in_idx, L->out_idx, L->neg, &es, L->dense); /* Hogwild: no lock */
        }
        w->rows_done += (uint64_t)L->R;
        w->loss_done += L->loss_sum;
        if (w->synchronous && *w->stop) break;
    }
    return NULL;
}

int orc_fit_threads(orc_model* m, const uint64_t* user_ptr, const uint32_t* item_ids, uint64_t num_users, uint32_t workers,
                    int synchronous, double max_seconds, uint64_t* out_interactions, double* out_seconds, float* out_loss) {
    if (!m || workers == 0 || workers > 4096) return SBR_ERR_INVALID_ARGUMENT;
    const uint32_t saved = m->hp.num_devices;
    m->hp.num_devices = workers; /* one partition per worker (sequence_model.rs:91-98) */
    orc_plan* p = NULL;
    int st = orc_fit_begin(m, user_ptr, item_ids, num_users, &p);
    m->hp.num_devices = saved;
    if (st != SBR_OK) return st;
    pthread_barrier_t bar;
    pthread_mutex_t mu;
    pthread_barrier_init(&bar, NULL, workers);
    pthread_mutex_init(&mu, NULL);
    orc_worker* ws = (orc_worker*)calloc(workers, sizeof(orc_worker));
    pthread_t* th = (pthread_t*)calloc(workers, sizeof(pthread_t));
    volatile int stop = 0;
    uint64_t rows = 0;
    double loss = 0.0;
    const double t0 = orc_now();
    for (uint32_t e = 0; e < m->hp.num_epochs && !stop; ++e) {
        uint64_t nmb = 0;
        orc_fit_epoch_prepare(p, &nmb);
        for (uint32_t q = 0; q < workers; ++q) {
            ws[q].p = p; ws[q].q = (int)q; ws[q].synchronous = synchronous; ws[q].nmb = nmb;
            ws[q].t_start = t0; ws[q].max_seconds = max_seconds; ws[q].bar = &bar; ws[q].mu = &mu; ws[q].stop = &stop;
            pthread_create(&th[q], NULL, orc_worker_main, &ws[q]);
        }
        for (uint32_t q = 0; q < workers; ++q) pthread_join(th[q], NULL);
        if (max_seconds > 0 && orc_now() - t0 > max_seconds) stop = 1;
    }
    const double dt = orc_now() - t0;
    float total = 0.0f; /* ≙ the fold at sequence_model.rs:173-177 (true loss sums) */
    for (uint32_t q = 0; q < workers; ++q) {
        rows += ws[q].rows_done;
        loss += ws[q].loss_done;
        total += (float)(ws[q].loss_done / (1.0 + (double)ws[q].rows_done));
    }
    if (out_interactions) *out_interactions = rows;
    if (out_seconds) *out_seconds = dt;
    if (out_loss) *out_loss = total;
    pthread_barrier_destroy(&bar);
    pthread_mutex_destroy(&mu);
    free(ws); free(th);
    orc_fit_plan_destroy(p);
    return SBR_OK;
}

int orc_fit_end(orc_plan* p, float* out_loss, uint64_t* out_examples) {
    /* ≙ the sum over the workers of loss_value / (1.0 + examples) (sequence_model.rs:173-177).  The
     * reference reads a stale node value (:157 before :160); reported here: the true
     * loss sums. */
    double total = 0.0;
    for (int q = 0; q < p->ndev; ++q) total += p->loss_dev[q] / (1.0 + (double)p->examples_dev[q]);
    if (out_loss) *out_loss = (float)total;
    if (out_examples) *out_examples = p->examples;
    return SBR_OK;
}

/* ≙ the value `fit` returns in the reference: sum over the workers of (stale node values) / (1 + examples) */
int orc_fit_end_lagged(orc_plan* p, float* out_loss) {
    if (!p || !out_loss) return SBR_ERR_INVALID_ARGUMENT;
    float total = 0.0f;
    for (int q = 0; q < p->ndev; ++q) total = total + p->lagged_dev[q] / (1.0f + (float)p->examples_dev[q]);
    *out_loss = total;
    return SBR_OK;
}

int orc_fit_debug_fetch(orc_plan* p, int q, int which, void* out, uint64_t bytes) {
    orc_local* L = &p->loc[q];
    uint64_t R = (uint64_t)L->R, d = (uint64_t)p->m->d;
    const void* src = NULL; uint64_t n = 0;
    switch (which) {
        case SBR_DBG_HIDDEN: src = L->H; n = R * d * 4; break;
        case SBR_DBG_NEGATIVES: src = L->neg; n = R * 4; break;
        case SBR_DBG_COEF: src = L->coef; n = R * 4; break;
        case SBR_DBG_LOSS: src = L->loss; n = R * 4; break;
        case SBR_DBG_DHIDDEN: src = L->dH; n = R * d * 4; break;
        case SBR_DBG_DINPUT: src = L->dX; n = R * d * 4; break;
        case SBR_DBG_DENSE_GRAD: src = L->dense; n = orc_ndense(p->m) * 4; break;
        case SBR_DBG_IN_IDX: src = L->in_idx; n = R * 4; break;
        case SBR_DBG_OUT_IDX: src = L->out_idx; n = R * 4; break;
        case SBR_DBG_TRIES: src = L->tries; n = R * 4; break;
        case SBR_DBG_DZ: src = L->dZ; n = R * (uint64_t)p->m->ng * d * 4; break;
        default: return SBR_ERR_INVALID_ARGUMENT;
    }
    if (bytes < n) return SBR_ERR_INVALID_ARGUMENT;
    memcpy(out, src, n);
    return SBR_OK;
}

/* ≙ fit_sequence_model (sequence_model.rs:70-178), whole call */
int orc_model_fit(orc_model* m, const uint64_t* user_ptr, const uint32_t* item_ids, uint64_t num_users,
                  float* out_loss) {
    orc_plan* p = NULL;
    int st = orc_fit_begin(m, user_ptr, item_ids, num_users, &p);
    if (st != SBR_OK) return st;
    for (uint32_t e = 0; e < m->hp.num_epochs; ++e) {
        uint64_t nmb = 0;
        orc_fit_epoch_prepare(p, &nmb);
        if (p->ndev > 1 && m->hp.parallelism == SBR_PAR_ASYNCHRONOUS) orc_fit_epoch_async(p, nmb);
        else for (uint64_t mb = 0; mb < nmb; ++mb) orc_fit_step(p, mb);
    }
    orc_fit_end(p, out_loss, NULL);
    orc_fit_end_lagged(p, &m->last_lagged_loss);
    orc_fit_plan_destroy(p);
    return SBR_OK;
}

/* what the reference's `fit` would have returned for the last orc_model_fit (sequence_model.rs:157, :173-177) */
int orc_model_last_fit_lagged_loss(const orc_model* m, float* out_loss) {
    if (!m || !out_loss) return SBR_ERR_INVALID_ARGUMENT;
    *out_loss = m->last_lagged_loss;
    return SBR_OK;
}

/* ------------------------------------------------------------------------------------------ */
/* ≙ user_representation (sequence_model.rs:182-211): last T items; empty history = one step
 * with the default index 0 (IndexInputNode::new(&[0;1]), lstm.rs:262-264). */
static int orc_user_representation_stored(orc_model* m, const uint32_t* item_ids, uint64_t n, float* out) {
    int d = m->d, ng = m->ng, coupled = m->hp.model == SBR_MODEL_LSTM_COUPLED;
    uint64_t T = m->hp.max_sequence_length;
    uint32_t zero = 0;
    if (n > T) { item_ids += n - T; n = T; }
    if (n == 0) { item_ids = &zero; n = 1; }
    for (uint64_t t = 0; t < n; ++t) if (item_ids[t] >= m->hp.num_items) return SBR_ERR_INVALID_ARGUMENT;
    if (ng) {
        int nz = ng * d;
        float* z = (float*)malloc(sizeof(float) * nz);
        float* h = (float*)calloc(d, 4); float* c = (float*)calloc(d, 4);
        float* hn = (float*)calloc(d, 4);
        for (uint64_t t = 0; t < n; ++t) {
            const float* x = m->E + (size_t)item_ids[t] * d;
            for (int j = 0; j < nz; ++j) z[j] = m->bW[j];
            for (int k = 0; k < d; ++k) { float xv = x[k]; const float* w = m->W + (size_t)k * nz; for (int j = 0; j < nz; ++j) z[j] = fmaf(xv, w[j], z[j]); }
            for (int k = 0; k < d; ++k) { float hv = h[k]; const float* w = m->W + (size_t)(d + k) * nz; for (int j = 0; j < nz; ++j) z[j] = fmaf(hv, w[j], z[j]); }
            for (int u = 0; u < d; ++u) {
                float zi = coupled ? 0.0f : z[u];
                float zf = coupled ? z[u] : z[d + u];
                float zg = coupled ? z[d + u] : z[2 * d + u];
                float zo = coupled ? z[2 * d + u] : z[3 * d + u];
                orc_cell cell = orc_lstm_cell(zi, zf, zg, zo, c[u], coupled);
                c[u] = cell.c; hn[u] = cell.h;
            }
            memcpy(h, hn, sizeof(float) * d);
        }
        memcpy(out, h, sizeof(float) * d);
        free(z); free(h); free(c); free(hn);
    } else {
        for (int k = 0; k < d; ++k) out[k] = m->E[(size_t)item_ids[0] * d + k];
        for (uint64_t t = 1; t < n; ++t) {
            const float* x = m->E + (size_t)item_ids[t] * d;
            for (int k = 0; k < d; ++k) {
                float a = orc_sigmoid(m->alpha[k]);
                float oma = 1.0f - a;
                out[k] = fmaf(a, out[k], oma * x[k]);
            }
        }
    }
    return SBR_OK;
}

int orc_user_representation(orc_model* m, const uint32_t* item_ids, uint64_t n, float* out) { /* out: embedding_dim floats */
    float rep[256];
    int st = orc_user_representation_stored(m, item_ids, n, rep);
    if (st == SBR_OK) memcpy(out, rep, sizeof(float) * (size_t)m->dl);
    return st;
}

/* ≙ predict (sequence_model.rs:213-232): bias + dot ("chain" order); non-finite fails the call */
static int orc_predict_stored(orc_model* m, const float* user, const uint32_t* item_ids, uint64_t n, float* out) {
    for (uint64_t i = 0; i < n; ++i) {
        if (item_ids[i] >= m->hp.num_items) return SBR_ERR_INVALID_ARGUMENT;
        float s = m->b[item_ids[i]] + orc_dot_prediction(user, m->E + (size_t)item_ids[i] * m->d, m->d);
        if (!isfinite(s)) return SBR_ERR_INVALID_PREDICTION;
        out[i] = s;
    }
    return SBR_OK;
}

int orc_predict(orc_model* m, const float* user, const uint32_t* item_ids, uint64_t n, float* out) { /* user: embedding_dim floats */
    float rep[256] = {0.0f};
    memcpy(rep, user, sizeof(float) * (size_t)m->dl);
    return orc_predict_stored(m, rep, item_ids, n, out);
}

/* ≙ mrr_score (evaluation.rs:12-48) */
int orc_mrr_score(orc_model* m, const uint64_t* user_ptr, const uint32_t* item_ids, uint64_t num_users,
                  float* out_mrr, uint32_t* out_ranks, uint64_t* out_num_ranked) {
    uint32_t I = m->hp.num_items;
    float* pred = (float*)malloc(sizeof(float) * I);
    float* rep = (float*)malloc(sizeof(float) * m->d);
    uint32_t* all = (uint32_t*)malloc(sizeof(uint32_t) * I);
    for (uint32_t i = 0; i < I; ++i) all[i] = i;
    float sum = 0.0f;
    uint64_t cnt = 0;
    int st = SBR_OK;
    for (uint64_t u = 0; u < num_users && st == SBR_OK; ++u) {
        uint64_t n = user_ptr[u + 1] - user_ptr[u];
        if (n < 2) continue; /* evaluation.rs:20 */
        const uint32_t* it = item_ids + user_ptr[u];
        uint32_t test_item = it[n - 1];
        st = orc_user_representation_stored(m, it, n - 1, rep);
        if (st != SBR_OK) break;
        st = orc_predict_stored(m, rep, all, I, pred);
        if (st != SBR_OK) break;
        for (uint64_t t = 0; t + 1 < n; ++t) pred[it[t]] = ORC_F32_MIN; /* :30-32, ALL history items */
        float ts = pred[test_item];
        uint32_t rank = 0;
        for (uint32_t i = 0; i < I; ++i) if (pred[i] >= ts) ++rank; /* :37-41 */
        if (out_ranks) out_ranks[cnt] = rank;
        sum += 1.0f / (float)rank; /* :43, :47 sequential f32 sum */
        ++cnt;
    }
    free(pred); free(rep); free(all);
    if (st != SBR_OK) return st;
    if (out_num_ranked) *out_num_ranked = cnt;
    if (out_mrr) *out_mrr = sum / (float)cnt;
    return SBR_OK;
}

/* ---- scalar primitives exported for unit tests --------------------------------------------- */
float orc_sigmoidf(float x) { return orc_sigmoid(x); }
float orc_tanhf(float x) { return orc_tanh(x); }
float orc_dot_tree(const float* x, const float* y, int d) { return orc_dot_training(x, y, d); }
float orc_dot_chain(const float* x, const float* y, int d) { return orc_dot_prediction(x, y, d); }
uint32_t orc_neg_draw(uint64_t epoch_key, uint32_t ctr, uint32_t try_idx, uint32_t num_items) {
    return orc_negative_draw(epoch_key, ctr, try_idx, num_items);
}
uint64_t orc_epoch_key(uint64_t fit_seed, uint64_t epoch) { return orc_epoch_key_of(fit_seed, epoch); }
void orc_adagrad(float* w, float* G, float g, float lr, float l2) { orc_adagrad_step(w, G, g, lr, l2); }
void orc_xorshift_stream(const uint8_t seed[16], uint32_t* out, int n) {
    orc_rng r;
    orc_rng_from_seed(&r, seed);
    for (int i = 0; i < n; ++i) out[i] = orc_rng_u32(&r);
}
/* out[M][N] = k-ascending fma chain seeded with c0 (NULL = 0): what an f32 MFMA accumulation
 * computes; used by tests/test_numerics_gpu.py to pin the "MFMA == fmaf chain" premise. */
void orc_fma_chain_gemm(const float* a, const float* b, const float* c0, int M, int K, int N, float* out) {
    for (int i = 0; i < M; ++i)
        for (int j = 0; j < N; ++j) {
            float acc = c0 ? c0[i * N + j] : 0.0f;
            for (int k = 0; k < K; ++k) acc = fmaf(a[i * K + k], b[k * N + j], acc);
            out[i * N + j] = acc;
        }
}
void orc_adam(float* w, float* m1, float* v2, float g, float lr, float l2, uint64_t t) {
    float c1, c2;
    orc_adam_bias_corrections(t, &c1, &c2);
    orc_adam_step(w, m1, v2, g, lr, l2, c1, c2);
}
/* rand 0.5 generators exported for unit tests (tests/test_rand05.py) */
void orc_rand_stream(const uint8_t seed[16], int what, uint64_t arg, uint64_t arg2, double* out, int n) {
    orc_rng r;
    orc_rng_from_seed(&r, seed);
    for (int i = 0; i < n; ++i) {
        if (what == 0) out[i] = (double)orc_rng_gen_range(&r, arg);
        else if (what == 1) out[i] = (double)orc_rng_uniform(&r, arg, arg2);
        else if (what == 2) out[i] = orc_rng_standard_normal(&r);
        else { uint8_t s[16]; orc_rng_gen_seed(&r, s); out[i] = (double)s[i % 16]; }
    }
}
uint64_t orc_rand_uniform_u64(const uint8_t seed[16], uint64_t lo, uint64_t hi, int skip) {
    orc_rng r;
    orc_rng_from_seed(&r, seed);
    uint64_t v = 0;
    for (int i = 0; i <= skip; ++i) v = orc_rng_uniform(&r, lo, hi);
    return v;
}
void orc_tanh_pq(float x, float* p, float* q) { sbr_tanh_pq(x, p, q); }
/* h of one normal cell with (zi, zf, zg, zo) = (x, x/2, -x, x/4 + 1), c_prev = 1/2: twin of the engine's selftest */
float orc_selftest_cell_h(float x) { return orc_lstm_cell(x, 0.5f * x, -x, fmaf(0.25f, x, 1.0f), 0.5f, 0).h; }
