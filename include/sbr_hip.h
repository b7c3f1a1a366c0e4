/* sbr_hip.h — C-ABI of libsbr_hip.so, the MI355X (gfx950) engine behind sbr's
 * sequence-recommender hot path.
 *
 * The reference crate (maciejkula/sbr-rs, mounted at /root/reference) has no FFI: its seam is the
 * private trait pair SequenceModelParameters / SequenceModel (src/models/sequence_model.rs:14-45)
 * under the public surface  Hyperparameters::build / fit, OnlineRankingModel, mrr_score and
 * data::CompressedInteractions.  Each entry point below cites the reference interface it
 * replaces; INTEGRATION.md shows the `extern "C"` block a maintainer of the Rust crate would add.
 *
 * Conventions: opaque handles; every call returns an sbr_status (no exceptions cross the ABI);
 * the caller owns all host buffers; the library owns device memory until *_destroy; item ids are
 * u32, CSR user pointers are u64 (reference: usize, src/lib.rs:77-81).  All float data is f32.
 * Calls on one model are not thread-safe except the const ones (user_representation / predict /
 * mrr_score), which serialise internally.
 */
#ifndef SBR_HIP_H
#define SBR_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- status codes: the two reference error enums (src/lib.rs:84-97) + ABI-level errors ------ */
typedef enum sbr_status {
    SBR_OK = 0,
    SBR_ERR_NO_INTERACTIONS = 1,    /* FittingError::NoInteractions      (lib.rs:93-97)  */
    SBR_ERR_INVALID_PREDICTION = 2, /* PredictionError::InvalidPredictionValue (lib.rs:84-89) */
    SBR_ERR_INVALID_ARGUMENT = 3,
    SBR_ERR_UNSUPPORTED = 4,
    SBR_ERR_NO_DEVICE = 5, /* no gfx950 device / HIP runtime failure: the engine never falls back to CPU */
    SBR_ERR_HIP = 6,
    SBR_ERR_OUT_OF_MEMORY = 7
} sbr_status;

/* ---- enums (src/models/mod.rs:15-41, src/models/lstm.rs:28-35) ------------------------------- */
typedef enum sbr_model_kind { SBR_MODEL_LSTM_NORMAL = 0, SBR_MODEL_LSTM_COUPLED = 1, SBR_MODEL_EWMA = 2 } sbr_model_kind;
typedef enum sbr_loss { SBR_LOSS_BPR = 0, SBR_LOSS_HINGE = 1, SBR_LOSS_WARP = 2 } sbr_loss;
typedef enum sbr_optimizer { SBR_OPT_ADAGRAD = 0, SBR_OPT_ADAM = 1 } sbr_optimizer;
/* Asynchronous with num_devices > 1 = staleness-one pipeline: minibatch k+1 is computed on parameters
 * that lack update k (the deterministic analogue of Hogwild); with one device both are
 * the same step.  sbr_group_fit honours it; a host driving the step halves itself orders them
 * [scatter k, dense k] -> {exchange k || step_local k+1} -> apply_table k. */
typedef enum sbr_parallelism { SBR_PAR_ASYNCHRONOUS = 0, SBR_PAR_SYNCHRONOUS = 1 } sbr_parallelism;

/* ---- hyper-parameters: lstm::Hyperparameters (lstm.rs:39-52) / ewma::Hyperparameters
 * (ewma.rs:45-57).  num_devices plays the role of num_threads (one partition of the shuffled
 * subsequences per device, sequence_model.rs:91-98); batch_sequences is the GPU minibatch:
 * that many subsequences are evaluated against one parameter snapshot and their gradients are
 * summed into one optimiser step (batch_sequences = 1 is the reference's per-sequence SGD). */
typedef struct sbr_hparams {
    uint32_t num_items;
    uint32_t max_sequence_length;
    uint32_t embedding_dim; /* 1 .. 256; widths other than 16 / 32 / 64 / 128 / 256 are stored zero-padded to the next one */
    float learning_rate;
    float l2_penalty;
    int32_t model;       /* sbr_model_kind */
    int32_t loss;        /* sbr_loss */
    int32_t optimizer;   /* sbr_optimizer */
    int32_t parallelism; /* sbr_parallelism */
    uint8_t seed[16];    /* XorShiftRng::from_seed (lstm.rs:129-132) */
    uint32_t num_epochs;
    uint32_t num_devices;      /* world size; this process drives exactly one device */
    uint32_t device_rank;      /* 0 <= rank < num_devices */
    uint32_t batch_sequences;  /* subsequences per optimiser step and device */
} sbr_hparams;

typedef struct sbr_model sbr_model;
typedef struct sbr_fit_plan sbr_fit_plan;

/* Parameter blocks for get/set (golden-vector tests, checkpoint/resume: the serde derives at
 * lstm.rs:204,386 / ewma.rs:208,401).  "*_ACC" are the Adagrad accumulators, which live next to
 * the value in wyrm's HogwildParameter and persist across fit calls. */
typedef enum sbr_param {
    SBR_PARAM_ITEM_EMBEDDING = 0,     /* [num_items][dim]                         */
    SBR_PARAM_ITEM_EMBEDDING_ACC = 1,
    SBR_PARAM_ITEM_BIAS = 2,          /* [num_items]                              */
    SBR_PARAM_ITEM_BIAS_ACC = 3,
    SBR_PARAM_LSTM_W = 4,             /* [2*dim][gates*dim], rows = [x ; h], column blocks i,f,g,o (coupled: f,g,o) */
    SBR_PARAM_LSTM_W_ACC = 5,
    SBR_PARAM_LSTM_B = 6,             /* [gates*dim]                              */
    SBR_PARAM_LSTM_B_ACC = 7,
    SBR_PARAM_EWMA_ALPHA = 8,         /* [dim]                                    */
    SBR_PARAM_EWMA_ALPHA_ACC = 9,
    /* Adam first moments (the "*_ACC" blocks hold the second moments under Adam); empty under Adagrad */
    SBR_PARAM_ITEM_EMBEDDING_M = 10,
    SBR_PARAM_ITEM_BIAS_M = 11,
    SBR_PARAM_LSTM_W_M = 12,
    SBR_PARAM_LSTM_B_M = 13,
    SBR_PARAM_EWMA_ALPHA_M = 14
} sbr_param;

/* Per-minibatch intermediates that tests fetch to compare against the oracle. */
typedef enum sbr_debug_buffer {
    SBR_DBG_HIDDEN = 0,     /* f32 [R][dim]   h_t (LSTM) / s_t (EWMA), packed time-major rows */
    SBR_DBG_NEGATIVES = 1,  /* u32 [R]        sampled negative item ids                         */
    SBR_DBG_COEF = 2,       /* f32 [R]        dloss/dneg                                        */
    SBR_DBG_LOSS = 3,       /* f32 [R]        loss terms                                        */
    SBR_DBG_DHIDDEN = 4,    /* f32 [R][dim]   dloss/dh from the scoring step                    */
    SBR_DBG_DINPUT = 5,     /* f32 [R][dim]   gradient w.r.t. the gathered input embedding      */
    SBR_DBG_DENSE_GRAD = 6, /* f32 dense grad block: LSTM [2*dim+1][gates*dim] (last row = bias) / EWMA [dim] */
    SBR_DBG_IN_IDX = 7,     /* u32 [R] */
    SBR_DBG_OUT_IDX = 8,    /* u32 [R] */
    SBR_DBG_TRIES = 9,      /* u32 [R] number of negatives scored (k of BASELINE.md §4) */
    SBR_DBG_DZ = 10         /* f32 [R][gates*dim] gradient w.r.t. the gate pre-activations (LSTM; column blocks as SBR_PARAM_LSTM_W) */
} sbr_debug_buffer;

/* ≙ Hyperparameters::build (lstm.rs:197-201, ewma.rs:201-205): allocates device parameters and
 * initialises them from hp->seed (embedding_init lstm.rs:22-25; biases, alpha zero). */
sbr_status sbr_model_create(const sbr_hparams* hp, sbr_model** out);
void sbr_model_destroy(sbr_model* m);

/* ≙ ImplicitLSTMModel::fit / ImplicitEWMAModel::fit (lstm.rs:395-397, ewma.rs:408-410) →
 * fit_sequence_model (sequence_model.rs:70-178).  CSR = CompressedInteractions
 * (data.rs:227-234): user_ptr[num_users+1], item_ids[user_ptr[num_users]], time-sorted per user.
 * Re-callable: every call trains num_epochs more epochs, optimiser state persists.
 * Returns SBR_ERR_NO_INTERACTIONS when no subsequence of length > 2 exists (:86-88). */
sbr_status sbr_model_fit(sbr_model* m, const uint64_t* user_ptr, const uint32_t* item_ids,
                         uint64_t num_users, float* out_loss);

/* The same fit, staged, so a caller (bench.py, the multi-GPU driver) can keep inputs resident in
 * HBM and time / interleave individual optimiser steps:
 *   begin          = chunking + shuffle + partition      (sequence_model.rs:76-98)
 *   epoch_prepare  = per-epoch reshuffle (:109) + packing + upload; returns #minibatches
 *   step           = one minibatch: forward, negative sampling, loss, BPTT, optimiser (:111-169)
 *   step_local/step_apply = the two halves of step (compute, optimiser); multi-device: see below
 *   end            = ≙ the fold at :173-177; returns loss_sum / (1 + examples)            */
sbr_status sbr_fit_begin(sbr_model* m, const uint64_t* user_ptr, const uint32_t* item_ids,
                         uint64_t num_users, sbr_fit_plan** out);
sbr_status sbr_fit_epoch_prepare(sbr_fit_plan* p, uint64_t* out_num_minibatches);
/* Optional: start shuffling/packing/uploading the NEXT epoch on a host thread while the device works
 * on the current one; the following sbr_fit_epoch_prepare consumes it.  Call only if another epoch
 * follows (the shuffle advances the partition RNGs and the epoch counter). */
sbr_status sbr_fit_epoch_prefetch(sbr_fit_plan* p);
sbr_status sbr_fit_step(sbr_fit_plan* p, uint64_t minibatch);
/* `count` consecutive optimiser steps from `first` (= that many sbr_fit_step calls; sbr_model_fit runs every epoch as ONE such
 * call).  At the reference's own schedule — one subsequence per optimiser step, sequence_model.rs:111-169 (batch_sequences = 1) — with
 * a single-negative loss (hinge, BPR) and Adagrad, a run of steps is ONE kernel launch — EWMA at embedding_dim <= 32 and
 * max_sequence_length <= 129; the LSTM (Normal) at embedding_dim 32 for steps of at most 48 rows (longer steps inside an epoch take
 * the separate launches, the run resumes behind them): one workgroup walks the steps with each step's working set in LDS (one gather
 * per step; nobody else touches the parameters at one sequence per step).  Same bits as the separate launches.  Single device. */
sbr_status sbr_fit_steps(sbr_fit_plan* p, uint64_t first, uint64_t count);
/* How one-sequence steps at embedding_dim <= 32 are launched: 0 = one launch per kernel family (eight per step), 1 = fused
 * launches (LSTM four per step: forward, scoring + header + key ordering, BPTT, gradient + updates; EWMA two), 2 (default) =
 * additionally runs of steps in one launch through sbr_fit_steps / sbr_model_fit where the shape allows.  No result bit depends
 * on it (tests run all three). */
sbr_status sbr_model_set_step_fusion(sbr_model* m, int32_t level);
/* REFERENCE ORDER of the negatives (one subsequence per step; every embedding_dim; steps whose h rows and 64-draw candidate window fit
 * one workgroup's LDS: max_sequence_length <= 256 at embedding_dim <= 64, <= 221 at <= 128, <= 81 at <= 256): the
 * negatives of a step are drawn from the worker's own sequential generator exactly as sequence_model.rs:58-65 / :137 draw them —
 * `Uniform::new(0, num_items).sample(thread_rng)` (rand 0.5 as recalled), one draw per try, the same generator that shuffles the
 * worker's partition every epoch (:109) — instead of the engine's counter-keyed draws (which exist so that draws can be evaluated in
 * parallel).  Everything else of a worker's step is unchanged.  Checked bit for bit against the oracle's reference-order mode; SBR_ERR_UNSUPPORTED outside
 * the shapes above.  Set before sbr_model_fit / sbr_fit_begin. */
sbr_status sbr_model_set_reference_order(sbr_model* m, int32_t on);
/* ... with several workers (Parallelism::Synchronous, replicated table): every worker's gradient is applied as its OWN optimiser
 * step, one after the other in worker order (sequence_model.rs:163-166; n Adagrad applications per step where the contract sums
 * the devices' gradients and applies one).  sbr_group_fit / sbr_group_step do this when the replicas are in reference order; hosts
 * that drive one process per GPU gather the devices' local blocks (sbr_fit_block_bytes each, block q at q * bytes; after
 * sbr_fit_step_local) and call sbr_fit_step_apply_blocks_in_order on every rank. */
sbr_status sbr_fit_block_bytes(const sbr_fit_plan* p, uint64_t* out_bytes);
sbr_status sbr_fit_step_apply_blocks_in_order(sbr_fit_plan* p, uint64_t minibatch, const void* device_blocks);
/* Profiling aid of the one-launch step runs: shader-clock ticks of the run's workgroup summed per phase since sbr_fit_begin —
 * [0] ids, key ordering, gather, [1] scan + scores, [2] backward scan, [3] reduction + updates, [4] closing barrier — and [5] the steps. */
sbr_status sbr_fit_debug_phase_clocks(sbr_fit_plan* p, uint64_t out[6]);
sbr_status sbr_fit_minibatch_rows(const sbr_fit_plan* p, uint64_t minibatch, uint64_t* out_rows);
sbr_status sbr_fit_end(sbr_fit_plan* p, float* out_loss, uint64_t* out_examples);
/* The number the reference's `fit` returns.  sequence_model.rs:157 adds `loss.value()` of the loss node BEFORE :160 runs its
 * forward pass; the loss nodes are shared running sums (lstm.rs:322-328), so a subsequence of s steps contributes the running sum
 * L_{s-1} of the worker's most recent earlier subsequence with AT LEAST s steps (0 if none since this fit began), accumulated in f32; `fit` returns the sum over the workers of accumulator / (1 + examples) (:173-177).
 * sbr_fit_end / sbr_model_fit report the true mean instead; this is the lagged figure for a caller that must return what the
 * crate returns.  sbr_fit_end_lagged: THIS device's term (single device: the whole figure; multi-device hosts add the terms in
 * device order in f32).  sbr_model_last_fit_lagged_loss: the whole figure of the last completed sbr_model_fit / sbr_group_fit. */
sbr_status sbr_fit_end_lagged(sbr_fit_plan* p, float* out_term);
sbr_status sbr_model_last_fit_lagged_loss(const sbr_model* m, float* out_loss);
/* Running totals of the plan, all devices: loss terms processed and negatives scored by the WARP
 * search (the k of BASELINE.md §4's bytes-per-interaction formula). */
sbr_status sbr_fit_counters(sbr_fit_plan* p, uint64_t* out_examples, uint64_t* out_negatives_scored);
/* The last step's sparse update on this device: gradient entries (3 per loss term: input, target and
 * negative row) and the distinct item-table rows they touch (each is read-modified-written once,
 * sequence_model.rs:163-169) — what the real HBM traffic of the update is priced from. */
sbr_status sbr_fit_sparse_stats(sbr_fit_plan* p, uint64_t* out_entries, uint64_t* out_unique_rows);
void sbr_fit_plan_destroy(sbr_fit_plan* p);

/* The two halves of a single-device step (sbr_fit_step = local + apply).  With one device sbr_fit_step_local COMMITS the
 * minibatch's loss terms and example count to the plan's totals (sbr_fit_end / sbr_fit_counters) — the header launch that
 * closes the scoring pass adds them — whether or not sbr_fit_step_apply follows; the exchange halves below do not count them a
 * second time. */
sbr_status sbr_fit_step_local(sbr_fit_plan* p, uint64_t minibatch);
sbr_status sbr_fit_step_apply(sbr_fit_plan* p, uint64_t minibatch);

/* Multi-device step (user-sharded data parallelism, one process per GPU; ≙ the rendezvous of
 * Parallelism::Synchronous, sequence_model.rs:163-166).  Table rows are owned in contiguous slices
 * of ceil(num_items / num_devices) rows.  Per step, after sbr_fit_step_local:
 *   scatter      : own entries reduced per row into num_devices dense chunks (send buffer, one chunk
 *                  per owner; chunk = [G: S*dim f32][gb: S f32][flags: S u32])
 *                                                                     -> host: all-to-all of the chunks
 *   dense        : the small dense block [8-word header | dense grads]; call it after the all-to-all
 *                  has been queued: it waits for the dense-gradient GEMM, which then overlaps the transfer
 *   owner_reduce : the devices' contributions for the owned slice, added in device order -> one chunk
 *                                                                     -> host: all-gather of the chunks
 *                                                                        and of the dense blocks
 *   apply_table  : every device applies the identical update (replicas stay bit-identical).
 * All pointers are device pointers supplied by the host (torch tensors in sbr_rs_amd/distributed.py). */
sbr_status sbr_fit_chunk_bytes(const sbr_fit_plan* p, uint64_t* out_bytes);
sbr_status sbr_fit_dense_bytes(const sbr_fit_plan* p, uint64_t* out_bytes);
sbr_status sbr_fit_step_scatter(sbr_fit_plan* p, uint64_t minibatch, void* device_send);
sbr_status sbr_fit_step_dense(sbr_fit_plan* p, void* device_dense_out);
sbr_status sbr_fit_step_owner_reduce(sbr_fit_plan* p, const void* device_recv, void* device_own_chunk);
/* The same kernel launched on `hip_stream` instead of the model's stream, without synchronising either: the
 * caller orders the two streams itself (events / wait_stream).  What the staleness-one pipeline
 * (Parallelism::Asynchronous) uses to keep the exchange of step k on a side stream underneath the computation
 * of step k+1. */
sbr_status sbr_fit_step_owner_reduce_on(sbr_fit_plan* p, const void* device_recv, void* device_own_chunk, void* hip_stream);
sbr_status sbr_fit_step_apply_table(sbr_fit_plan* p, const void* device_table, const void* device_dense_all);
/* sbr_fit_step_apply_table in two halves: the item-table rows (needs only the gathered chunks, so it can be enqueued
 * while the dense-gradient GEMM is still running on the engine's side stream), then the dense parameters (after
 * sbr_fit_step_dense has joined that GEMM and the dense blocks have been gathered).  Rows first — it opens the
 * optimiser step —, dense second, once each per step; same bits as the one-call form. */
sbr_status sbr_fit_step_apply_rows(sbr_fit_plan* p, const void* device_table);
sbr_status sbr_fit_step_apply_dense(sbr_fit_plan* p, const void* device_dense_all);

/* OWNER-APPLIED update — the Synchronous step of a replicated table since round 6 (≙ the ONE shared parameter and optimiser state
 * behind every worker, lstm.rs:259-260 / ewma.rs:267-269, under the synchronised step of sequence_model.rs:163-169).  Per step, after
 * sbr_fit_step_local:
 *   scatter (as above)                                                   -> host: all-to-all of the chunks
 *   owner_update : the devices' contributions to the owned slice added in device order AND the one optimiser update of every touched
 *                  row of that slice, in place in this replica (opens the optimiser step)
 *                                                                        -> host: all-gather, IN PLACE, of the updated PARAMETER slices:
 *                                                                           sbr_model_table_slice(ITEM_EMBEDDING) and (ITEM_BIAS)
 *   dense + all-gather of the dense blocks + apply_dense (as above)
 * Same sums in the same order and the same update arithmetic as owner_reduce + apply_rows — bit-identical results — and the same
 * bytes on the links, but no replica walks the whole table (sbr_fit_step_apply_rows visits every row's flag on every device) and a
 * row's optimiser state is maintained by its owner alone.  Consequence: while such a fit runs, a replica's copy of the item table's
 * optimiser state (SBR_PARAM_ITEM_*_ACC / _M) is current for its OWN rows only; sbr_model_get_param on those blocks and the
 * gradient-all-gather halves (apply_table / apply_rows) return SBR_ERR_INVALID_ARGUMENT until the host has all-gathered those blocks
 * too (same in-place slices) and called sbr_model_optimizer_state_gathered — the library's own drivers (sbr_group_fit_end,
 * sbr_model_fit_comm) and sbr_rs_amd/distributed.py do that when a fit ends.  The staleness-one pipeline (Asynchronous) keeps the
 * gradient all-gather: its update lands one step late on every replica.
 *   sbr_model_table_slice: device pointer of an item-table block of THIS replica (allocated for num_devices slices of
 *   ceil(num_items / num_devices) rows each, so that slices are equally long) and the bytes of one slice: rank r's slice is
 *   [base + r * slice_bytes, + slice_bytes).  which: SBR_PARAM_ITEM_EMBEDDING / _ACC / _M, SBR_PARAM_ITEM_BIAS / _ACC / _M; base = NULL
 *   for a block the model does not have (Adam moments under Adagrad). */
sbr_status sbr_fit_step_owner_update(sbr_fit_plan* p, const void* device_recv);
sbr_status sbr_model_table_slice(sbr_model* m, int32_t which, void** out_base, uint64_t* out_slice_bytes);
sbr_status sbr_model_optimizer_state_gathered(sbr_model* m);
sbr_status sbr_model_optimizer_state_is_partial(const sbr_model* m, int32_t* out);

/* The same multi-device fit driven from ONE process (≙ fit with num_threads(n) on one host,
 * sequence_model.rs:90-102): models[r] was created with num_devices = n, device_rank = r, the same
 * seed, on the device that was current at its creation (sbr_set_device).  The exchange runs as
 * peer copies between the devices' streams, ordered with events; results are bit-identical to the
 * one-process-per-GPU driver.  n = 1 is sbr_model_fit. */
sbr_status sbr_group_fit(sbr_model* const* models, uint32_t n, const uint64_t* user_ptr, const uint32_t* item_ids,
                         uint64_t num_users, float* out_loss);
sbr_status sbr_device_count(int32_t* out_count);

/* sbr_group_fit taken apart, for a host that wants the group's steps one at a time (a progress bar, early stopping, the bench's
 * `--driver group`, the parity tests): begin -> per epoch [epoch_prepare -> step(0) .. step(n_minibatches - 1)] -> fit_end.
 * sbr_group_fit IS this sequence over hp.num_epochs epochs.  (≙ the loop of sequence_model.rs:100-171 with num_threads(n).)
 *   sbr_group_epoch_prepare   every replica's sbr_fit_epoch_prepare (prefetch_next != 0: the next epoch is packed in the background)
 *   sbr_group_step            one optimiser step of the whole group (Synchronous / partitioned / the staleness-one pipeline);
 *                             returns when it is QUEUED on the devices' streams (a partitioned step too since round 6: its owners
 *                             wait for the devices' events on their streams and read the owner bounds on the device)
 *   sbr_group_step_local      parity access: only the local halves of `minibatch` (forward, scoring, BPTT on every replica); the
 *                             next sbr_group_step(minibatch) then runs the exchange and the update alone.  Not for Asynchronous.
 *   sbr_group_member_plan     replica r's plan, borrowed (sbr_fit_debug_fetch, sbr_fit_minibatch_rows, sbr_fit_counters)
 *   sbr_group_plan_set_host_threads   one host thread per device queues that device's launches (the reference runs one rayon
 *                             worker per partition, sequence_model.rs:100-102); default: from four devices on.  Same bits.
 *   sbr_group_plan_stats      host time spent inside sbr_group_step so far, the number of steps, the host threads in use
 *   sbr_group_fit_end         the loss of sbr_group_fit; destroys the plan.  sbr_group_plan_destroy: abandon a plan. */
typedef struct sbr_group_plan sbr_group_plan;
sbr_status sbr_group_fit_begin(sbr_model* const* models, uint32_t n, const uint64_t* user_ptr, const uint32_t* item_ids,
                               uint64_t num_users, sbr_group_plan** out);
sbr_status sbr_group_epoch_prepare(sbr_group_plan* g, uint64_t* out_num_minibatches, int32_t prefetch_next);
sbr_status sbr_group_step(sbr_group_plan* g, uint64_t minibatch);
sbr_status sbr_group_step_local(sbr_group_plan* g, uint64_t minibatch);
sbr_status sbr_group_member_plan(sbr_group_plan* g, uint32_t replica, sbr_fit_plan** out);
sbr_status sbr_group_synchronize(sbr_group_plan* g);
sbr_status sbr_group_plan_set_host_threads(sbr_group_plan* g, int32_t enable);
sbr_status sbr_group_plan_stats(const sbr_group_plan* g, double* out_host_enqueue_ms, uint64_t* out_steps, int32_t* out_host_threads);
/* The Synchronous step of a replicated group is the owner-applied update (sbr_fit_step_owner_update: parameter slices travel, in
 * place); gradient_all_gather != 0 selects the gradient all-gather + whole-table update of rounds 1-5 instead (A/B measurements and
 * the parity tests of those halves; the pipeline always runs it).  Same bits.  sbr_group_gather_optimizer_state: every replica's
 * copy of the item table's optimiser state made complete from the owners' slices — sbr_group_fit_end does it; a host that reads
 * SBR_PARAM_ITEM_*_ACC between steps calls it first. */
sbr_status sbr_group_plan_set_exchange(sbr_group_plan* g, int32_t gradient_all_gather);
sbr_status sbr_group_gather_optimizer_state(sbr_group_plan* g);
sbr_status sbr_group_fit_end(sbr_group_plan* g, float* out_loss);
void sbr_group_plan_destroy(sbr_group_plan* g);

/* Builds the n replicas of a single-process group in one call (replica r on HIP device r mod device
 * count; destroy each with sbr_model_destroy).  flags = 0: n full parameter replicas, exactly what n
 * sbr_model_create calls give.  SBR_GROUP_PARTITION_ITEM_TABLE (BASELINE configs[4]): the
 * item table (embeddings, biases and their optimiser state) exists ONCE — rows [r*S, (r+1)*S),
 * S = ceil(num_items / n), live on replica r's device and are mapped into every replica's address space
 * (HIP virtual memory management; remote rows are read over xGMI by the unchanged gather kernels).  In
 * sbr_group_fit each row is then updated by its owner only, from the devices' gradient lists merged in
 * device order: bitwise the same result as the replicated Synchronous exchange, with per-step traffic
 * proportional to the batch instead of to the table.  Prediction / mrr_score / get_param work on any
 * replica.  Parallelism::Asynchronous over a partitioned table runs the synchronous step (the owners update in
 * place after a rendezvous; there is no staleness-one pipeline to run) — in sbr_group_fit, under one process per GPU
 * and with the peer transport alike, so that a hyper-parameter draw never decides whether a configuration can run. */
#define SBR_GROUP_PARTITION_ITEM_TABLE 1u
sbr_status sbr_group_create(const sbr_hparams* hp, uint32_t n, uint32_t flags, sbr_model** out_models);
sbr_status sbr_model_is_partitioned(const sbr_model* m, int32_t* out);

/* The partitioned item table under ONE PROCESS PER GPU (the launcher model of bench.py --gpus N).  Every
 * process creates its model with the same hyper-parameters (num_devices = world, device_rank = its rank) on
 * its own device; the shared virtual range is planned identically everywhere and the parts (runs of pages)
 * homed on this rank are allocated.  The host then passes file descriptors between the processes
 * (SCM_RIGHTS over a Unix socket; sbr_rs_amd/partitioned.py): every part is exported by its home rank and
 * imported by all the others; sbr_partition_finalize then writes this rank's rows (the seeded initial
 * embeddings, zeroed optimiser state).  The fit is sequenced by the host:
 *   sbr_fit_lists_export / _import   once per plan: the ranks' gradient lists become readable by their peers
 *   per step: sbr_fit_step_local -> sbr_fit_step_reduce_own (returns this rank's owner bounds; stream
 *   drained) -> all-gather of the bounds and of the dense blocks -> sbr_fit_step_owner_apply (stream
 *   drained) -> barrier;  or, nothing drained, the *_queued forms below.
 * Same bits as sbr_group_fit over a partitioned group and as the replicated Synchronous exchange. */
/* Peer transport of the REPLICATED exchange (one process per GPU): every rank exports its send buffer and
 * its reduced own chunk once; the owner-reduce and table-update kernels then read the peers' buffers in
 * place through peer mappings (xGMI), so no bulk collective is involved.  Host sequencing per step:
 *   sbr_fit_step_local -> sbr_fit_step_scatter_shared -> barrier -> sbr_fit_step_owner_reduce_peers ->
 *   sbr_fit_step_dense + all-gather of the dense blocks (a barrier as well) -> sbr_fit_step_apply_table_peers
 *   -> barrier.  Every call returns with the stream drained.  Same bits as the collective transport. */
sbr_status sbr_fit_exchange_export(sbr_fit_plan* p, int32_t out_fds[2], uint64_t out_bytes[2]);
sbr_status sbr_fit_exchange_import(sbr_fit_plan* p, uint32_t peer_rank, const int32_t fds[2], const uint64_t bytes[2]);
sbr_status sbr_fit_step_scatter_shared(sbr_fit_plan* p, uint64_t minibatch);
sbr_status sbr_fit_step_owner_reduce_peers(sbr_fit_plan* p);
sbr_status sbr_fit_step_apply_table_peers(sbr_fit_plan* p, const void* device_dense_all);

sbr_status sbr_model_create_partitioned(const sbr_hparams* hp, sbr_model** out);
sbr_status sbr_partition_num_parts(const sbr_model* m, uint32_t* out);
sbr_status sbr_partition_part_info(const sbr_model* m, uint32_t part, uint32_t* out_home_rank, uint64_t* out_bytes);
sbr_status sbr_partition_export_part(sbr_model* m, uint32_t part, int32_t* out_fd);
sbr_status sbr_partition_import_part(sbr_model* m, uint32_t part, int32_t fd);
sbr_status sbr_partition_finalize(sbr_model* m);
sbr_status sbr_fit_lists_export(sbr_fit_plan* p, int32_t out_fds[4], uint64_t out_bytes[4]);
sbr_status sbr_fit_lists_import(sbr_fit_plan* p, uint32_t peer_rank, const int32_t fds[4], const uint64_t bytes[4]);
sbr_status sbr_fit_step_reduce_own(sbr_fit_plan* p, uint64_t minibatch, uint32_t* host_bounds, void* device_dense_out);
sbr_status sbr_fit_step_owner_apply(sbr_fit_plan* p, const uint32_t* all_bounds, const void* device_dense_all);
/* The same two halves WITHOUT the host in the loop (round 6), for hosts whose collectives run on the device (RCCL): nothing is
 * drained and nothing is read back.  reduce_own_queued leaves this rank's owner bounds (num_devices + 1 u32) on the device and
 * returns their address; the host all-gathers the ranks' bounds and dense blocks with DEVICE collectives on the model's stream (a
 * collective completes on a rank only after every rank's contribution: every rank has finished reading the table); the owner
 * builds its merge plan from the gathered bounds on the device (rank r's bounds at r * (num_devices + 1) words) and updates its
 * rows; a last small device collective keeps the next step's reads behind every owner's writes.  Same bits. */
sbr_status sbr_fit_step_reduce_own_queued(sbr_fit_plan* p, uint64_t minibatch, void** out_device_bounds, void* device_dense_out);
sbr_status sbr_fit_step_owner_apply_queued(sbr_fit_plan* p, const void* device_all_bounds, const void* device_dense_all);

/* The rendezvous of a step through RCCL INSIDE the library — for hosts that run one process per GPU and have no collective library
 * of their own (≙ the synchronised optimiser step of sequence_model.rs:92, 163-166 across processes; xGMI within a node).  librccl
 * is opened at run time; hosts that bring their own transport (torch.distributed, MPI: the sbr_fit_step_scatter / _owner_reduce /
 * _apply_* halves above) never load it.
 *   sbr_comm_unique_id    rank 0 makes the 128-byte id; the host hands it to the other ranks (a file, a socket, an environment
 *                         variable: any channel)
 *   sbr_comm_create       every rank, on its own device (sbr_set_device first), with the same id
 *   sbr_fit_step_exchange after sbr_fit_step_local: scatter -> all-to-all -> owner update (in place) -> all-gather of the updated
 *                         parameter slices into every replica's table, dense block -> all-gather -> dense update, all queued on
 *                         the model's stream; same bits as every other transport
 *   sbr_model_fit_comm    the whole fit of this rank (sbr_model_fit for num_devices = world across processes); the loss is the
 *                         all-rank figure, sbr_model_last_fit_lagged_loss holds THIS rank's term of the reference's figure
 * Replicated table, Parallelism::Synchronous order of work.  SBR_ERR_UNSUPPORTED: no librccl on this host. */
typedef struct sbr_comm sbr_comm;
sbr_status sbr_comm_unique_id(uint8_t out_id[128]);
sbr_status sbr_comm_create(const uint8_t id[128], uint32_t world, uint32_t rank, sbr_comm** out);
void sbr_comm_destroy(sbr_comm* c);
sbr_status sbr_fit_step_exchange(sbr_fit_plan* p, uint64_t minibatch, sbr_comm* c);
/* after the last sbr_fit_step_exchange of a fit driven step by step: the owners' optimiser-state slices all-gathered in place
 * (sbr_model_fit_comm does it itself) */
sbr_status sbr_comm_gather_optimizer_state(sbr_model* m, sbr_comm* c);
sbr_status sbr_model_fit_comm(sbr_model* m, sbr_comm* c, const uint64_t* user_ptr, const uint32_t* item_ids, uint64_t num_users,
                              float* out_loss);

/* Device pointer / stream plumbing for the host side (torch only supplies memory + streams). */
sbr_status sbr_model_set_stream(sbr_model* m, void* hip_stream);
sbr_status sbr_model_synchronize(sbr_model* m);

/* Debug / parity access to the last minibatch processed by sbr_fit_step*. */
sbr_status sbr_fit_debug_fetch(sbr_fit_plan* p, int32_t which, void* host_out, uint64_t bytes);

/* ≙ OnlineRankingModel::user_representation (lib.rs:105-108; impl sequence_model.rs:182-211):
 * keeps the last max_sequence_length items; empty history = step 0 with item 0. */
sbr_status sbr_user_representation(sbr_model* m, const uint32_t* item_ids, uint64_t n, float* out_dim);
/* ≙ OnlineRankingModel::predict (lib.rs:111-115; impl sequence_model.rs:213-232):
 * out[i] = bias[item_ids[i]] + <user, E[item_ids[i]]>; SBR_ERR_INVALID_PREDICTION if any is non-finite. */
sbr_status sbr_predict(sbr_model* m, const float* user_dim, const uint32_t* item_ids, uint64_t n, float* out);
/* ≙ evaluation::mrr_score (evaluation.rs:12-48).  out_ranks (optional) receives one u32 rank per
 * test user with >= 2 interactions, in user order; out_num_ranked their count. */
sbr_status sbr_mrr_score(sbr_model* m, const uint64_t* user_ptr, const uint32_t* item_ids,
                         uint64_t num_users, float* out_mrr, uint32_t* out_ranks, uint64_t* out_num_ranked);

/* ≙ the serde derives (lstm.rs:204,386; ewma.rs:208,401): element counts and raw access. */
sbr_status sbr_model_param_count(const sbr_model* m, int32_t which, uint64_t* out_count);
sbr_status sbr_model_get_param(sbr_model* m, int32_t which, float* host_out, uint64_t count);
sbr_status sbr_model_set_param(sbr_model* m, int32_t which, const float* host_in, uint64_t count);
/* Selected rows of an item-table block — SBR_PARAM_ITEM_EMBEDDING / _ACC / _M: host_out [n][embedding_dim]; SBR_PARAM_ITEM_BIAS /
 * _ACC / _M: host_out [n] — without moving the whole table (1e7 items x 256 is 10 GB per block). */
sbr_status sbr_model_get_param_rows(sbr_model* m, int32_t which, const uint32_t* rows, uint64_t n, float* host_out);
sbr_status sbr_model_get_epoch(const sbr_model* m, uint64_t* out_global_epoch);
/* Optimiser steps taken so far (Adam's bias-correction counter) and the epoch counter that keys the
 * negative draws: together with the parameter blocks this is the complete resumable state. */
sbr_status sbr_model_get_counters(const sbr_model* m, uint64_t* out_global_epoch, uint64_t* out_optimizer_steps);
/* State of the model RNG (the Hyperparameters' rng, which the reference serialises with the model,
 * lstm.rs:38-51): 16 bytes that re-create it through XorShiftRng::from_seed.  A restored model continues
 * the shuffle / partition-seed stream of the saved one. */
sbr_status sbr_model_get_rng(const sbr_model* m, uint8_t out_state[16]);
sbr_status sbr_model_set_rng(sbr_model* m, const uint8_t state[16]);
sbr_status sbr_model_set_counters(sbr_model* m, uint64_t global_epoch, uint64_t optimizer_steps);

/* Library / device identification ("gfx950", CU count, HBM bytes); device_name may be NULL. */
sbr_status sbr_device_info(char* device_name, uint64_t name_bytes, uint32_t* out_cus, uint64_t* out_hbm_bytes);
const char* sbr_status_string(sbr_status s);
uint32_t sbr_abi_version(void);

/* Scratch of a fit call (device and pinned-host blocks up to 64 MiB, at most 768 MiB of each kind per process) is kept for the
 * next call instead of going back to the driver: the reference's own bench re-fits one model in a loop
 * (benches/benchmark.rs:40-42) and ~50 allocations per call cost more than its kernels.  This returns everything that is idle
 * to the driver (a long-lived host process that is done fitting). */
void sbr_release_cached_memory(void);

/* Kernel timing hook for bench.py: wall time (ms, HIP events on the engine stream) and launch
 * count accumulated per kernel family since the last reset.  Families: see sbr_kernel_family. */
typedef enum sbr_kernel_family {
    SBR_K_RECURRENT_FWD = 0,
    SBR_K_SCORE = 1,       /* gather + negative sampling + loss: the HBM-roofline kernel */
    SBR_K_RECURRENT_BWD = 2,
    SBR_K_DENSE_GRAD = 3,
    SBR_K_DENSE_UPDATE = 4,
    SBR_K_SPARSE_UPDATE = 5,
    SBR_K_RANK = 6,
    SBR_K_SPARSE_SORT = 7, /* key ordering of the sparse update: radix passes + segment heads (sbr_sort.hip) */
    SBR_K_FAMILIES = 8
} sbr_kernel_family;
sbr_status sbr_model_timing_enable(sbr_model* m, int32_t enable);
/* Which families' launches are bracketed by events while timing is enabled (bit f = sbr_kernel_family f; default: all).  Every
 * bracketed launch costs two event records on its stream — a few microseconds each, which a 2.5 ms step of ~20 launches feels —
 * so a throughput measurement that needs one kernel's duration selects that family alone. */
sbr_status sbr_model_timing_select(sbr_model* m, uint32_t family_mask);
/* enable = 0 queues the side-stream work (key sort, dense-gradient GEMM) on the main stream, so that every
 * kernel family is timed running alone; results are identical.  Default: overlap on. */
sbr_status sbr_model_set_overlap(sbr_model* m, int32_t enable);
sbr_status sbr_model_timing_read(sbr_model* m, double* out_ms /*[SBR_K_FAMILIES]*/, uint64_t* out_launches /*[SBR_K_FAMILIES]*/);


/* Selects the HIP device this process drives (before sbr_model_create); default = current device. */
sbr_status sbr_set_device(int32_t ordinal);

/* Numerics-contract self-tests: run sbr_numerics.h primitives on the device so tests can compare
 * them bit for bit with the CPU oracle (no reference counterpart: wyrm's fast-math kernels are
 * replaced by the engine's own). */
sbr_status sbr_selftest_math(const float* x, uint64_t n, float* out_cell_h, float* out_sig, float* out_tanh);
sbr_status sbr_selftest_dot_tree(const float* x, const float* y, uint32_t d, uint64_t nrows, float* out);
sbr_status sbr_selftest_mfma(const float* a, const float* b, const float* c0, uint32_t k, float* out,
                             const float* a32, const float* b32, float* out32);
/* The sparse update's key ordering alone (sbr_sort.hip; ≙ the per-row visiting order of Optimizer::step over the sparse
 * gradients, sequence_model.rs:163-169): out_keys[n] = (rows[e] << 32 | e) in (row, e) order, row ids below 2^row_bits;
 * out_head_pos[*out_nheads + 1] = positions where a new row starts, then n. */
sbr_status sbr_selftest_sort(const uint32_t* rows, uint64_t n, uint32_t row_bits, uint64_t* out_keys, uint32_t* out_head_pos,
                             uint32_t* out_nheads);

#ifdef __cplusplus
}
#endif
#endif /* SBR_HIP_H */
