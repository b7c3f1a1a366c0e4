/* sbr_kernels.h — launch interface between the host engine (sbr_engine.hip) and the gfx950
 * kernels (sbr_kernels.hip).  Internal to libsbr_hip.so; the public ABI is include/sbr_hip.h. */
#ifndef SBR_KERNELS_H
#define SBR_KERNELS_H

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace sbr {

/* One device-local packed minibatch.  Rows are time-major: row(t, b) = off[t] + b, sequences b in
 * length-descending order, so the rows of step t are a prefix of the sequences of step t-1. */
struct MbView {
    int R;                   /* packed rows = sum over sequences of steps                     */
    int B;                   /* sequences                                                     */
    int Tm;                  /* max steps                                                     */
    const int* off;          /* device [Tm+1]                                                 */
    const int* steps;        /* device [B]   steps of sequence b                              */
    const int* prev_row;     /* device [R]   packed row of (t-1, b) or -1                      */
    const uint32_t* in_idx;  /* device [R]                                                    */
    const uint32_t* out_idx; /* device [R]   = in_idx of the same sequence's next step (both packers write item[t] / item[t + 1] of one
                              *               slice): the EWMA scans hand a step's target row on as the next step's input               */
    const uint32_t* ctr;     /* device [R]   p * max_sequence_length + t                       */
};

/* Views into one exchange block (layout: oracle/sbr_oracle.c "exchange block"). */
struct BlockView {
    uint32_t* header; /* [0]=R, [4..5]=f64 loss_sum, [6..7]=u64 examples */
    uint32_t* in_idx;
    uint32_t* out_idx;
    uint32_t* neg;
    float* coef;
    float* H;
    float* dX;
    float* dense;
};

struct ModelView {
    int d, ng;          /* ng = 4 / 3 / 0 */
    int coupled;
    uint32_t num_items;
    int loss;           /* sbr_loss */
    float lr, l2;
    float *E, *Eacc, *b, *bacc;
    float *W, *Wacc, *bW, *bWacc, *Wp, *WTp;
    float *alpha, *alpha_acc;
    /* Adam */
    int optimizer;      /* sbr_optimizer */
    float c1, c2;       /* 1 - beta^t of the current step */
    float *Em, *bm, *Wm, *bWm, *alpha_m;
};

/* Up to this many tiles the length-sorted tile list of the sequence-resident recurrent kernels is folded (alternate groups of
 * 256 reversed): with about as many tiles as resident slots (512) every CU then gets long + short.  ms per step on one box,
 * folded / not: 375 tiles 2.29 / 2.55, 512 tiles 2.65 / 2.97 (BPTT alone 0.72 / 0.91 — it had no fold before), 625 tiles 3.16 /
 * 3.33, 750 and 1 024 tiles equal; with several rounds of tiles the dispatcher's own order — longest first, next tile to the
 * first free slot — is the better schedule: 1 563 tiles 13.65 / 13.53. */
#define SBR_FOLD_MAX_TILES_DEFAULT 1024
struct WorkView { /* per-plan scratch, sized for Rmax rows / Bmax sequences */
    float *C, *G, *dH, *dZ;
    float* X;               /* [Rmax][d] copy of the gathered input rows (forward) for the dense-gradient GEMM */
    float *dHrec, *dCrec;   /* [Bmax][d] */
    float* dab;             /* EWMA per-sequence dalpha partials [Bmax][d] */
    float* partials;        /* dense-gradient chunk partials */
    float* loss;            /* [Rmax] */
    uint32_t* tries;        /* [Rmax] */
    double* part_loss;      /* per-workgroup partials of the score kernel [2048] */
    unsigned int* part_tries;
    float* zeros;           /* 256 zeros (h_{-1} of the dense-gradient GEMM, 64-bit address path) */
    int fold_max_tiles;     /* the recurrent kernels fold their length-sorted tile list up to this many tiles */
    int wide_addresses;     /* 1: the dense-gradient GEMM takes its 64-bit per-lane address path even where the buffer path would do (tests) */
    int stream_activations; /* set per launch by launch_recurrent_forward: 1 = gates / cell states / input copy are stored nt (the step's h rows fit
                             * the Infinity Cache and the score kernel, next on the stream, finds them there: SBR_STREAM_MAX_H_BYTES) */
};

/* one chunk pointer per device: slices of one gathered buffer (collective transport), or the peers' own
 * buffers read in place through peer mappings (peer transport) */
struct ChunkPtrs {
    const void* p[16];
};

/* Which segments of the ordered keys take the chunked path (seg_chunk_kernel: one lane group per <= SBR_SEG_CHUNK-entry chunk, the
 * chunks spread evenly over the chip; seg_finish_kernel adds a segment's chunk partials in order) instead of being reduced by the
 * lane group that meets them in seg_short_kernel: every segment of more than SBR_SEG_ROUTE entries.  The CONTRACT's chunk stays
 * SBR_SEG_CHUNK = 256 (a segment of 33..256 entries is ONE chunk: the same in-order sum either way, same bits); the routing
 * threshold only decides who walks it.  Under a skewed catalogue the grid-stride short-segment pass otherwise waits for the few lane
 * groups that meet several 100-entry segments one after the other (Zipf(1) items at 8 192 sequences per step: 0.81 ms against 0.28
 * for uniform items). */
#ifndef SBR_SEG_ROUTE
#define SBR_SEG_ROUTE 32
#endif
/* scratch of the long-segment path of the sparse reduction */
struct SegScratch {
    uint32_t* counters;   /* [0] long segments, [1] chunk units */
    uint32_t* long_start; /* first / one-past-last key position of every long segment */
    uint32_t* long_end;
    uint32_t* unit_base;  /* exclusive prefix of the long segments' chunk counts */
    float* P;             /* chunk partials [units][D] */
    float* Pb;
    uint32_t* Pf;         /* 1 = the chunk has a bias contribution */
    uint32_t cap;         /* capacity of the long-segment arrays */
    /* segment heads of the sorted keys (positions whose row differs from the previous key's), ascending, plus
     * the sentinel head_pos[*nheads] = number of keys; produced right after the sort, on its stream */
    uint32_t* head_pos;
    uint32_t* nheads;
    uint32_t prelisted;   /* 1: the long segments were listed and their chunk units counted right after the ordering (launch_seg_prelist):
                           * the short-segment pass skips them, their chunks are reduced beside it on another stream */
};
/* single device, large step: the hot rows' part of the sparse update off the update's critical path.  launch_seg_prelist (the
 * ordering's stream, underneath BPTT) lists the segments of more than SBR_SEG_CHUNK entries and counts their chunk units;
 * launch_seg_apply then runs the short segments only, and launch_seg_hot_apply reduces and applies the listed ones on another stream
 * (disjoint table rows). */
void launch_seg_prelist(const SegScratch& sc, hipStream_t s);

/* where the owner's row range lies in every device's sorted keys — DEVICE-resident, written by merge_plan_kernel from the devices'
 * owner bounds (round 6: the host no longer reads the bounds back, so a partitioned step is queued without a host rendezvous) */
struct MergePlan {
    uint32_t lo[16];   /* first position of the owner's row range in device r's sorted keys */
    uint32_t base[17]; /* exclusive prefix sum of the range lengths; base[16] = their total */
};
/* the devices' gradient lists as the owner of a row range sees them (device pointers, peer-readable) */
struct PeerLists {
    const uint64_t* keys[16];
    const float* G[16];
    const float* gb[16];
    const uint32_t* fl[16];
    const MergePlan* plan; /* on the owner's device */
};
/* device r's owner bounds [ndev + 1] (owner_bounds_kernel), peer-readable or gathered */
struct PeerBounds {
    const uint32_t* b[16];
};

/* recurrent forward over all steps of the minibatch (LSTM d <= 128: one sequence-resident launch; d = 256: one launch per step) */
void launch_recurrent_forward(const ModelView& m, const MbView& mb, float* H, const WorkView& w, int tm_host,
                              const int* off_host, hipStream_t s);
/* sbr_wave.hip: the wave-per-sequence form of the recurrent pass for small minibatches at d <= 32 (false: not taken, the
 * caller launches the tile kernels; same outputs bit for bit) */
bool launch_wave_forward(const ModelView& m, const MbView& mb, float* H, const WorkView& w, int tm_host, hipStream_t s);
bool launch_wave_backward(const ModelView& m, const MbView& mb, const BlockView& blk, const WorkView& w, int tm_host, int b_host,
                          hipStream_t s);
/* dense-gradient chunk partials of a small step at d <= 32: one wave per 32 x 32 output block (w.partials as the tile kernels') */
bool launch_wave_dense_gradient(const ModelView& m, const MbView& mb, const BlockView& blk, const WorkView& w, int rows_host,
                                hipStream_t s);
/* gather + negative sampling + loss + dloss/dh; also copies in/out idx into the block */
/* A step of ONE subsequence (the reference's own schedule, sequence_model.rs:111-169) at d <= 32: what follows the score pass —
 * block header + loss accumulators, the lagged loss figure, the (row, entry) ordering of the step's 3 R keys and its segment
 * heads — runs at the end of the score launch itself (one workgroup) instead of in three more launches of ~5 us each. */
#define SBR_SMALL_TAIL_MAX_ROWS 255
struct SmallTail {
    uint32_t* header;
    double* loss_acc;            /* may be null (several devices: the header travels) */
    unsigned long long* ex_acc;
    float* lag_state;            /* [accumulator | (loss node k, its staged next value) per step k] (sbr_report.hip) */
    uint64_t* keys_sorted;
    uint32_t* head_pos;
    uint32_t* nheads;
};
bool small_tail_shape_ok(const ModelView& m, int sequences_host, int rows_host, bool wide = false); /* wide: every kernel width (the reference-order step) */
bool reference_order_shape_ok(int d, int max_rows); /* sbr_steps.hip: the sequential-stream scorer's LDS */
/* Reference order (one sequence per step, one device): negatives from the worker's sequential xorshift stream (rng_state: 4 words on
 * the device, advanced by exactly the draws consumed), then the SmallTail.  Returns false where the one-sequence form does not exist. */
bool launch_score_reference_order(const ModelView& m, const MbView& mb, const BlockView& blk, const WorkView& w, uint32_t* rng_state,
                                  int rows_host, hipStream_t s, const SmallTail& tail);
/* The optimiser half of a small single-device LSTM step at d <= 32 (one dense-gradient chunk, single-launch sparse update) in ONE
 * launch: dense gradient (per-element row chains) + dense update + sparse update, instead of three. */
bool small_back_shape_ok(const ModelView& m, int rows_host);
/* A RUN of consecutive one-sequence optimiser steps (the reference's own schedule, sequence_model.rs:111-169) at d <= 32 in ONE
 * launch: one workgroup walks the steps with each step's working set in LDS — one gather of the step's 3 n rows and of the touched
 * rows' optimiser state, then scan, scores, backward scan, dalpha, key ordering, per-row reduction and the Adagrad updates out of
 * LDS (ewma_steps_kernel, sbr_steps.hip).  EWMA with a single-negative loss, Adagrad, at most SBR_EWMA_STEPS_MAX_ROWS rows per
 * step.  desc[i] = the i-th step of the epoch's packed arrays (one sequence each). */
struct StepDesc { uint32_t rows, row_base, off_base, seq_base; };
struct EpochView {
    const int* off;
    const int* steps;
    const int* prev_row;
    const uint32_t *in_idx, *out_idx, *ctr;
    const StepDesc* desc;
};
#define SBR_EPOCH_STEPS_MAX_LDS (150 * 1024)  /* dynamic LDS of the run's workgroup */
/* The same for the LSTM (Normal, d = 32, single-negative loss, Adagrad: the reference's Criterion shape): lstm_steps_kernel walks a
 * run of steps of at most lstm_steps_max_rows() rows each; lag_rows = max_sequence_length - 1 (the loss nodes of sbr_report.hip). */
bool lstm_steps_shape_ok(const ModelView& m, int lag_rows_host);
int lstm_steps_max_rows();
void launch_lstm_steps(const ModelView& m, const EpochView& ev, const BlockView& blk, const WorkView& w, uint64_t epoch_key,
                       const SmallTail& tail, int step_begin, int step_end, int lag_rows_host, int run_max_rows_host,
                       unsigned long long* phase_clocks, hipStream_t s);
bool epoch_steps_shape_ok(const ModelView& m, int max_rows_host);
void launch_epoch_steps(const ModelView& m, const EpochView& ev, const BlockView& blk, const WorkView& w, uint64_t epoch_key,
                        const SmallTail& tail, int step_begin, int step_end, int max_rows_host,
                        unsigned long long* phase_clocks /* [6] or null */, hipStream_t s);
void launch_seg_hot_apply(const ModelView& m, const BlockView& blk, const uint64_t* keys_sorted, const SegScratch& sc, hipStream_t s);
void launch_small_back(const ModelView& m, const MbView& mb, const BlockView& blk, const WorkView& w, uint32_t rows_host,
                       const uint64_t* keys_sorted, const SegScratch& sc, hipStream_t s);
void launch_score(const ModelView& m, const MbView& mb, const BlockView& blk, const WorkView& w, uint64_t epoch_key,
                  int rows_host, hipStream_t s, const SmallTail* tail = nullptr);
/* EWMA with a single-negative loss (hinge / BPR): forward scan, scoring and backward scan of a sequence in ONE pass (replaces
 * launch_recurrent_forward + launch_score + launch_recurrent_backward's scan; same outputs bit for bit; the dalpha reduction stays
 * launch_dense_gradient's) */
void launch_ewma_sequences(const ModelView& m, const MbView& mb, const BlockView& blk, const WorkView& w, uint64_t epoch_key,
                           int rows_host, hipStream_t s, const SmallTail* tail = nullptr);
/* debug only: dloss/dh of every packed row (the training path never materialises it) */
void launch_materialize_dh(const ModelView& m, const BlockView& blk, int rows_host, float* dH, hipStream_t s);
/* header of the exchange block (rows, loss sum, examples); loss_acc / ex_acc non-null (single device): the plan's accumulators
 * take the header in the same launch; lag_state non-null (small step, <= SBR_HEADER_LAG_MAX_B sequences): the lagged loss figure
 * (below) in the same launch too */
#define SBR_HEADER_LAG_MAX_B 1024
void launch_block_header(const ModelView& m, const BlockView& blk, const WorkView& w, const MbView& mb, int rows_host, double* loss_acc,
                         unsigned long long* ex_acc, float* lag_state, hipStream_t s);
void launch_block_header_parts(uint32_t* header, int rows_host, const double* part_loss, const unsigned int* part_tries, int nparts,
                               double* loss_acc, unsigned long long* ex_acc, const MbView& mb, const float* loss, float* lag_state,
                               hipStream_t s);
/* sbr_report.hip — the loss figure the reference's `fit` returns (sequence_model.rs:157 reads the loss node BEFORE :160 runs its
 * forward pass: a subsequence of s steps contributes the running sum L_{s-1} of the worker's most recent earlier subsequence with at
 * least s steps).  seq_loss: the sequences' running sums, t ascending (px[b] = what sequence b + 1 reads; the nodes' next values);
 * lagged_chain: the strictly sequential f32 accumulation over the minibatch's sequences on one wave.
 * lag_state = [accumulator | (node k, its staged next value) per step k]. */
void launch_seq_loss(const MbView& mb, const float* loss, float* px, float* lag_state, int b_host, hipStream_t s);
void launch_lagged_chain(const MbView& mb, const float* px, int b_host, float* lag_state, hipStream_t s);
/* BPTT (dX, dZ) and, separately, the dense gradient into blk.dense (may run on a second stream:
 * it reads only dZ, X, H) */
void launch_recurrent_backward(const ModelView& m, const MbView& mb, const BlockView& blk, const WorkView& w,
                               int tm_host, int rows_host, int b_host, const int* off_host, hipStream_t s);
/* returns the number of chunk partials left unreduced in w.partials (defer_reduce and more than one chunk), else 0 with blk.dense complete */
int launch_dense_gradient(const ModelView& m, const MbView& mb, const BlockView& blk, const WorkView& w, int rows_host,
                          int b_host, hipStream_t s, bool defer_reduce = false);
void launch_dense_reduce(const ModelView& m, const WorkView& w, int nchunks, const BlockView& blk, hipStream_t s);
void launch_dense_reduce_apply(const ModelView& m, const WorkView& w, int nchunks, const BlockView& blk, hipStream_t s);
/* dense: sum over device blocks in device order + Adagrad (+ repack of the LSTM weights) */
void launch_dense_apply(const ModelView& m, const uint8_t* all_blocks, uint64_t block_bytes, uint64_t dense_off,
                        int ndev, hipStream_t s);
void launch_repack_lstm(const ModelView& m, hipStream_t s);
/* sparse: keys of the device's own entries in (row, entry) order + the list of segment heads (sbr_sort.hip: a
 * stable LSD radix sort over the row bits, key_bits = 32 + bits of a row id; needs only indices and negatives) ->
 * per-row reduction in the contract's chunked order -> one of three consumers: optimiser update (single device),
 * the owners' dense send chunks (replicated multi-device), the position-addressed list + owner bounds
 * (partitioned table).  `keys` is a scratch array of the same size as `keys_sorted`. */
size_t sparse_sort_temp_bytes(size_t max_entries, int key_bits);
/* early_mb != null (single-negative losses): keys from the minibatch's index arrays and the negative-draw hash, i.e.
 * without waiting for the score kernel */
void launch_own_sort(const BlockView& blk, uint32_t rows_host, uint64_t* keys, uint64_t* keys_sorted, void* sort_temp,
                     size_t sort_temp_bytes, int key_bits, const SegScratch& sc, hipStream_t s, const MbView* early_mb = nullptr,
                     uint64_t epoch_key = 0, uint32_t num_items = 0);
void launch_seg_apply(const ModelView& m, const BlockView& blk, uint32_t rows_host, const uint64_t* keys_sorted,
                      const SegScratch& sc, hipStream_t s);
void launch_seg_scatter(const ModelView& m, const BlockView& blk, uint32_t rows_host, int ndev, uint64_t slice_rows,
                        void* send, const uint64_t* keys_sorted, const SegScratch& sc, hipStream_t s);
void launch_seg_list(const ModelView& m, const BlockView& blk, uint32_t rows_host, int ndev, uint64_t slice_rows,
                     const uint64_t* keys_sorted, float* G, float* gbl, uint32_t* fl, uint32_t* bounds, const SegScratch& sc,
                     hipStream_t s);
void launch_owner_reduce(const ModelView& m, const ChunkPtrs& recv, int ndev, uint64_t slice_rows, void* own, hipStream_t s);
void launch_table_apply(const ModelView& m, const ChunkPtrs& table, uint64_t slice_rows, hipStream_t s);
/* owner-applied update: device-order sum of the devices' contributions to the owner's rows [row0, row0 + nrows) and ONE optimiser
 * update of each touched row, in place (owner_reduce + table_apply for the owner's slice in one pass; the parameter slices travel) */
void launch_owner_update(const ModelView& m, const ChunkPtrs& recv, int ndev, uint64_t slice_rows, uint64_t row0, uint64_t nrows, hipStream_t s);
/* partitioned item table: the owner merges the peers' lists (read through peer mappings) in device order
 * and updates its rows */
void launch_owner_list_apply(const ModelView& m, const PeerLists& pl, const PeerBounds& pb, int ndev, int owner, MergePlan* plan,
                             uint32_t capacity, uint64_t* mkeys, uint64_t* mkeys_sorted, void* sort_temp, size_t sort_temp_bytes, hipStream_t s);
/* accumulate loss/examples headers of all blocks into the plan accumulators */
/* owner side of the partitioned table: (row, device, position) keys of the peers' list heads in that order
 * (generated in (device, position) order, so again a stable sort on the row bits) */
void launch_merge_sort(const PeerLists& pl, int ndev, uint32_t total, uint64_t* mkeys, uint64_t* mkeys_sorted, void* sort_temp,
                       size_t sort_temp_bytes, hipStream_t s);
void launch_accumulate_loss(const uint8_t* all_blocks, uint64_t block_bytes, int ndev, double* loss_acc,
                            unsigned long long* ex_acc, hipStream_t s);
/* prediction side */
void launch_predict(const ModelView& m, const float* user, const uint32_t* items, uint64_t n, float* out, hipStream_t s);
void launch_rank(const ModelView& m, const float* reps, const int* rep_row, uint32_t num_users, const uint32_t* test_item,
                 const uint32_t* test_in_hist, const uint64_t* hist_ptr, const uint32_t* hist_items, float* ts_scratch,
                 uint32_t* ranks, uint32_t* nonfinite_flag, hipStream_t s);
/* device self-tests of the numerics contract (tests/test_numerics_gpu.py) */
void launch_selftest_math(const float* x, float* out_cell_h, float* out_sig, float* out_tanh, uint64_t n, hipStream_t s);
void launch_selftest_dot_tree(const float* x, const float* y, int d, uint64_t nrows, float* out, hipStream_t s);
void launch_selftest_mfma_chain(const float* a, const float* b, const float* c0, int k, float* out, hipStream_t s);
void launch_selftest_mfma32_chain(const float* a, const float* b, int k, float* out, hipStream_t s);
/* the key ordering alone: keys (rows[e] << 32 | e) of n entries in (row, e) order + segment heads */
void launch_selftest_sort(const uint32_t* rows, uint32_t n, int row_bits, uint64_t* tmp, uint64_t* out, void* temp, uint32_t* head_pos,
                          uint32_t* nheads, hipStream_t s);

}  // namespace sbr
#endif
