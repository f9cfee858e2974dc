/*
 * mf_hip.h -- C ABI of libmf_hip.so, the gfx950 (MI355X / CDNA4) implementation of
 * the two-tower matrix-factorization hot path.
 *
 * The reference (yxtay/matrix-factorization-torch, package xfmr_rec) is pure Python
 * and has no FFI layer; its boundary for this path is a torch.nn.Module call
 * (SURVEY.md 8b).  Each entry point below names the reference interface it
 * replaces (file:line under /root/reference).  INTEGRATION.md shows the ctypes
 * binding a maintainer of the reference would add.
 *
 * Conventions
 *   * every pointer is a DEVICE pointer (hipMalloc / torch.cuda storage) unless
 *     its name ends in _host; all float data is fp32, all ids are int64;
 *   * matrices are dense row-major, rows 16-byte aligned; the embedding width d
 *     must be one of 32, 64, 128, 256 (callers zero-pad other widths; zero columns
 *     change no value, see mf_numerics.h);
 *   * `stream` is a hipStream_t (NULL = default stream); no entry point allocates,
 *     frees or synchronises -- scratch comes from the caller (`ws`, sized by the
 *     matching *_ws_bytes query), so every call can be captured in a hipGraph;
 *   * return value 0 = ok, otherwise a negative MF_E* code; mf_last_error() gives
 *     the text (thread-local).
 */
#ifndef MF_HIP_H
#define MF_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* mf_stream_t; /* hipStream_t */

enum {
    MF_OK = 0,
    MF_EINVAL = -1,   /* bad argument (shape, width, NULL pointer) */
    MF_ENOSPC = -2,   /* workspace too small */
    MF_ELAUNCH = -3,  /* HIP launch error */
    MF_ENOTSUP = -4   /* valid request outside the implemented range */
};

/* Loss classes, in the order of get_loss_fns (xfmr_rec/lightning.py:267-287). */
enum mf_loss_kind {
    MF_ALIGNMENT = 0,             /* AlignmentLoss                  losses.py:249-259 */
    MF_CONTRASTIVE = 1,           /* ContrastiveLoss                losses.py:262-274 */
    MF_ALIGNMENT_CONTRASTIVE = 2, /* AlignmentContrastiveLoss       losses.py:277-291 */
    MF_INFONCE = 3,               /* InfomationNoiseContrastive...  losses.py:294-306 */
    MF_MINE = 4,                  /* MutualInformationNeuralEst...  losses.py:309-321 */
    MF_PAIRWISE_HINGE = 5,        /* PairwiseHingeLoss              losses.py:357-359 */
    MF_PAIRWISE_LOGISTIC = 6,     /* PairwiseLogisticLoss           losses.py:352-354 */
    MF_NUM_KINDS = 7
};

const char* mf_last_error(void);
int mf_version(void);

/* Optional HIP-event timing of the dominant kernels ("loss_fwd_dense",
 * "loss_bwd_du", "loss_bwd_dv", "mining_select", "topk_select", "topk_small", "gather_rows",
 * "update_rows"), recorded on the launch stream.  mf_timing_enable(k): 0 = off, k >= 1 = time every
 * k-th launch of each name (an event pair costs a few us of stream time).  mf_timing_get blocks until
 * the recorded spans finished and returns their count (total_ms = summed duration). */
void mf_timing_enable(int every);
void mf_timing_reset(void);
int64_t mf_timing_get(const char* name, double* total_ms);

/* ------------------------------------------------------------------ towers ---
 * Replaces the tower forward `MatrixFactorizationLitModule.forward`
 * (xfmr_rec/lightning.py:60-74: text -> BERT -> mean-pool -> L2-normalise) by an
 * embedding-table row gather (north-star; no reference implementation):
 *   out[r, :] = table[idx[r], :]            (normalize == 0)
 *   out[r, :] = table[idx[r], :] / max(||.||, 1e-12)   (normalize != 0, mirrors
 *               sentence_transformers Normalize, xfmr_rec/models.py:59)
 * out_inv_norm (nullable) receives 1 / max(||row||, 1e-12). */
int mf_gather_rows(const float* table, int64_t n_rows, int d, const int64_t* idx, int64_t n,
                   int normalize, float* out, float* out_inv_norm, mf_stream_t stream);

/* Hash / bloom embedding tower (BASELINE config 5: catalogs too large for one row per id; our spec,
 * mf_numerics.h mf_hash_bucket): id -> num_hashes (1..4) bucket rows of a [num_buckets, d] table,
 *   out[r, :] = sum_j table[bucket_j(idx[r]), :]   (j ascending; then / max(||.||, 1e-12) if normalize)
 * mf_hash_buckets writes the bucket rows themselves, out_buckets[r * num_hashes + j] -- the row ids
 * the sparse update of the table is called with.  mf_normalize_backward turns the gradient w.r.t.
 * the normalised output into the gradient w.r.t. the summed rows (the same for each of the id's
 * rows): graw = (g - u (u . g)) * inv_norm, u = the normalised output row. */
int mf_gather_hashed(const float* table, int64_t num_buckets, int d, const int64_t* idx, int64_t n,
                     int num_hashes, uint64_t seed, int normalize, float* out, float* out_inv_norm,
                     mf_stream_t stream);
int mf_hash_buckets(const int64_t* idx, int64_t n, int num_hashes, uint64_t seed, int64_t num_buckets,
                    int64_t* out_buckets, mf_stream_t stream);
int mf_normalize_backward(const float* out_unit, const float* inv_norm, const float* grad, int64_t n, int d,
                          float* grad_raw, mf_stream_t stream);

/* out[r] = chain-ordered ||x_r||^2 (mf_numerics.h); helper of the loss path. */
int mf_row_sqnorm(const float* x, int64_t n, int d, float* out, mf_stream_t stream);

/* Raw score tile engine exposed for tests: out[i][j] = chain dot(u_i, v_j) computed
 * by the fp32 MFMA path (must equal oracle/chain.c bit for bit). */
int mf_scores(const float* u, int64_t B, const float* v, int64_t N, int d, float* out,
              mf_stream_t stream);

/* --------------------------------------------------------------- batch ids ---
 * Stable sort of n int64 keys (0 <= key < 2^39, n < 2^24): perm[p] = original
 * position of the p-th smallest key, sorted_keys[p] = its key (nullable).
 * Used for duplicate-row handling and for the positive-mask range search that
 * replaces the B x N x P comparison of negative_masks (xfmr_rec/losses.py:108). */
size_t mf_sort_ws_bytes(int64_t n);
int mf_sort_keys(const int64_t* keys, int64_t n, int32_t* perm, int64_t* sorted_keys, void* ws,
                 size_t ws_bytes, mf_stream_t stream);
/* Stable grouping by a SMALL key (0 <= key < nkeys <= 64, n <= 32,768; the owner rank of a routed id, distributed.py): perm
 * and sorted_keys (nullable) as mf_sort_keys, bounds[nkeys + 1] = where each key's group starts.  One launch, one workgroup
 * (a counting sort in LDS); MF_ENOTSUP beyond the limits. */
int mf_group_keys(const int64_t* keys, int64_t n, int nkeys, int32_t* perm, int64_t* sorted_keys, int64_t* bounds,
                  mf_stream_t stream);

/* ------------------------------------------------------------------ losses ---
 * Replaces `EmbeddingLoss.forward` for all seven classes
 * (xfmr_rec/losses.py:39-52; call site xfmr_rec/lightning.py:137-146):
 *   u[B,d] user_embed, v[N,d] item_embed (N >= B; row j < B is the positive of
 *   user j), target[B] (the rating: fp32, or the reference's int64 with MF_LOSS_TARGET_I64),
 *   item_idx[N], pos_idx[B,P] (0-padded, nullable with P = 0), logq (nullable; our logQ
 *   correction L_ij -= logq_j: one value per column, logq[N], when logq_rows = 0, else a
 *   table of logq_rows values looked up by the batch ids, logq_j = logq[item_idx[j]],
 *   ids outside the table counting as 0 -- which needs item_idx != NULL).
 * flags: MF_LOSS_TARGET_I64 (target is int64), MF_LOSS_MASKS_READY (mf_loss_masks already ran on this workspace for
 *   these ids: same as item_idx = NULL, but the ids stay available to a logQ table lookup), MF_LOSS_ROWC (kind_mask has ONE bit set and that
 *   loss will be differentiated: the forward's tail also prepares the backward's per-row
 *   coefficients, and mf_loss_bwd is then called with MF_LOSS_ROWC too and skips that launch).
 * kind_mask selects which losses to evaluate in the one pass (bit k = kind k);
 * out_losses[7] receives them (the entries of the other kinds are set to 0).  The workspace keeps
 * the per-row statistics / mined negatives for mf_loss_bwd and must stay intact
 * between the two calls.  out_mask_bits (nullable, B x ceil(N/32) uint32, bit c of
 * word [i][w] = column 32w+c) receives the post-mining negative mask
 * (negative_masks + semi_hard_mining, losses.py:92-162) for tests.
 * num_negatives follows losses.py:137-141: <= 0 or >= N disables mining.
 * Mining supports num_negatives <= 64 (MF_ENOTSUP beyond).
 *
 * The hit masks (negative_masks, losses.py:92-110) depend on the ids only.  mf_loss_masks builds
 * them into the workspace ahead of time -- typically on a second stream, beside the tower gathers --
 * and the forward is then called with item_idx = NULL on the SAME workspace, after the caller has
 * ordered the two streams (event); with item_idx != NULL mf_loss_fwd builds them itself. */
size_t mf_loss_ws_bytes(int64_t B, int64_t N, int d, int P, int num_negatives);
int mf_loss_masks(int64_t B, int64_t N, int d, int P, int num_negatives, const int64_t* item_idx,
                  const int64_t* pos_idx, void* ws, size_t ws_bytes, mf_stream_t stream);
/* CSR form of the positives (the batch producer's own lists, mf_sample_batch's pos_off / pos_items): the positives of batch
 * row i are pos_items[pos_off[u] .. pos_off[u + 1]), u = user_ids[i] (a u outside [0, num_users) has none).  No [B, P]
 * tensor exists on this path: the reference pads every batch to its longest list (xfmr_rec/data/lightning.py:274-280,
 * data/load.py:38-55), which for MovieLens-25M means > 10^4 columns.  Same masks, bit for bit, as the padded form of the
 * same lists (order and duplicates inside a list do not matter; lists need not be sorted). */
int mf_loss_masks_csr(int64_t B, int64_t N, int d, int num_negatives, const int64_t* item_idx, const int64_t* user_ids,
                      const int64_t* pos_off, const int64_t* pos_items, int64_t num_users, void* ws, size_t ws_bytes,
                      mf_stream_t stream);
enum { MF_LOSS_TARGET_I64 = 1, MF_LOSS_ROWC = 2, MF_LOSS_MASKS_READY = 4 };
int mf_loss_fwd(int64_t B, int64_t N, int d, int P, int num_negatives, float sigma, float margin,
                int kind_mask, const float* u, const float* v, const void* target,
                const int64_t* item_idx, const int64_t* pos_idx, const float* logq, int64_t logq_rows,
                int flags, void* ws, size_t ws_bytes, float* out_losses, uint32_t* out_mask_bits,
                mf_stream_t stream);

/* mf_loss_fwd with CSR positives (see mf_loss_masks_csr); everything else as above.  mf_loss_ws_bytes(B, N, d, 0, k) sizes ws. */
int mf_loss_fwd_csr(int64_t B, int64_t N, int d, int num_negatives, float sigma, float margin, int kind_mask, const float* u,
                    const float* v, const void* target, const int64_t* item_idx, const int64_t* user_ids, const int64_t* pos_off,
                    const int64_t* pos_items, int64_t num_users, const float* logq, int64_t logq_rows, int flags, void* ws,
                    size_t ws_bytes, float* out_losses, uint32_t* out_mask_bits, mf_stream_t stream);

/* Backward of one loss of the preceding mf_loss_fwd (same shapes/hyper-parameters,
 * same ws): du[B,d] = grad_out * dloss/du, dv[N,d] = grad_out * dloss/dv, with
 * grad_out a device scalar.  Masks and mining are constants of the backward
 * (@torch.no_grad in the reference, losses.py:92,134).  The targets and logQ values are the
 * workspace's copies of the forward's.  flags: MF_LOSS_ROWC as above. */
int mf_loss_bwd(int64_t B, int64_t N, int d, int P, int num_negatives, float sigma, float margin,
                int kind, const float* u, const float* v, int flags, void* ws, size_t ws_bytes,
                const float* grad_out, float* du, float* dv, mf_stream_t stream);

/* The candidate search of the mined losses (0 < num_negatives < N) has two implementations with IDENTICAL results: the fp32
 * streaming selection, and a split-bf16 prefilter on the bf16 matrix cores with exact fp32 rescoring of the few columns that
 * pass (csrc/mf_mine_bf.h; it can serve d in {64, 128}, B >= 256, N >= 2048, num_negatives <= 32).  mode 1 (default): the
 * prefilter where it is the faster one (B >= 4096); mode 2: wherever it can serve (what the parity tests use
 * to compare the two on small shapes); mode 0: the fp32 search everywhere.  MF_MINE_BF=0|1|2 in the environment sets the initial
 * mode.  Process-wide; not a per-stream setting. */
void mf_set_mining_prefilter(int mode);

/* API parity with the public helper methods of EmbeddingLoss, on caller-provided tensors (not the
 * hot path): negative_masks (losses.py:92-110) -> out_mask[B,N] bytes, 1 = valid negative;
 * hard_mining (losses.py:112-132, semi_hard = 0: keep the k highest logits among the valid
 * negatives) and semi_hard_mining (losses.py:134-162, semi_hard = 1) on a MATERIALISED logits[B,N]
 * matrix, mask[B,N] bytes updated in place; k <= 0 or k >= N leaves it unchanged.  Any k. */
size_t mf_negative_masks_ws_bytes(int64_t B, int64_t N, int P);
int mf_negative_masks(int64_t B, int64_t N, int P, const int64_t* item_idx, const int64_t* pos_idx, void* ws,
                      size_t ws_bytes, uint8_t* out_mask, mf_stream_t stream);
int mf_mine_logits(const float* logits, int64_t B, int64_t N, int k, int semi_hard, uint8_t* mask,
                   mf_stream_t stream);

/* --------------------------------------------------------------- optimiser ---
 * Replaces `configure_optimizers` (xfmr_rec/lightning.py:238-239, dense AdamW) by
 * sparse row updates of an embedding table (north-star; our spec):
 * duplicate ids are summed first (batch order), then each touched row is updated
 * once.  `normalized` != 0 means grad is w.r.t. the L2-normalised row and is first
 * mapped through the normalisation Jacobian of the raw row.
 *   sgd : row -= lr * (g + wd * row)
 *   adam: lazy row-wise AdamW (moments of touched rows only, global step for the
 *         bias correction, decoupled weight decay).  The step (1-based) comes by value, or --
 *         step_dev != NULL -- from device memory: a launch captured in a hipGraph freezes its
 *         by-value arguments, a device counter bumped inside the graph keeps counting.  Either
 *         way the bias corrections are evaluated on the device: eager and replayed steps agree
 *         bit for bit. */
size_t mf_update_ws_bytes(int64_t n, int d);
int mf_update_sgd(float* table, int64_t n_rows, int d, const int64_t* idx, int64_t n,
                  const float* grad, int normalized, float lr, float weight_decay, void* ws,
                  size_t ws_bytes, mf_stream_t stream);
int mf_update_adam(float* table, float* exp_avg, float* exp_avg_sq, int64_t n_rows, int d,
                   const int64_t* idx, int64_t n, const float* grad, int normalized, int64_t step,
                   const int64_t* step_dev, float lr, float beta1, float beta2, float eps,
                   float weight_decay, void* ws, size_t ws_bytes, mf_stream_t stream);

/* Both tables of a training step -- the user rows and the item rows touched by one batch (xfmr_rec/lightning.py:189-192: one
 * optimizer.step() per step covers every parameter) -- in ONE launch: the two sparse updates are independent, so their
 * workgroups run side by side.  Same arithmetic, same results as two mf_update_sgd / mf_update_adam calls (adam = 0 / 1), the same
 * hyper-parameters for both tables, one workspace each (mf_update_ws_bytes).  MF_ENOTSUP for lists longer than 65,536 ids or empty
 * ones: callers then update the tables one by one. */
int mf_update_pair(int adam, int d, float* table_a, float* exp_avg_a, float* exp_avg_sq_a, int64_t n_rows_a, const int64_t* idx_a,
                   int64_t n_a, const float* grad_a, int normalized_a, void* ws_a, size_t ws_a_bytes, float* table_b,
                   float* exp_avg_b, float* exp_avg_sq_b, int64_t n_rows_b, const int64_t* idx_b, int64_t n_b, const float* grad_b,
                   int normalized_b, void* ws_b, size_t ws_b_bytes, int64_t step, const int64_t* step_dev, float lr, float beta1,
                   float beta2, float eps, float weight_decay, mf_stream_t stream);

/* -------------------------------------------------- the default step in one launch ---
 * The reference trains with BATCH_SIZE = 32 pairs (xfmr_rec/params.py:18), PairwiseHingeLoss and 4 mined negatives
 * (xfmr_rec/lightning.py:38-39): a step of ~0.5 MFLOP that the multi-kernel path spends in launch latency.  mf_step_small
 * runs the WHOLE training step -- mf_gather_rows of both towers, mf_loss_fwd (all kinds of kind_mask into out_losses[7]),
 * mf_loss_bwd of `kind` with upstream gradient 1, mf_update_sgd / mf_update_adam of both tables -- in ONE workgroup and one
 * launch, for B <= 128 pairs, N <= 256 columns, tables of < 2^36 rows and a mined loss (0 < num_negatives <= 64, < N).  Results (losses, both
 * tables, Adam moments) are bit-identical to that sequence of calls.  Positives: padded pos_idx[B, P], or -- pos_off != NULL --
 * CSR lists indexed by user_ids (see mf_loss_fwd_csr).  adam = 0: SGD (lr, weight_decay); else lazy row-wise AdamW with the
 * global step by value or from step_dev.  MF_ENOTSUP for shapes / losses outside that range (callers fall back). */
size_t mf_step_small_ws_bytes(int d);
int mf_step_small(float* user_table, float* user_m, float* user_v, int64_t num_users, float* item_table, float* item_m,
                  float* item_v, int64_t num_items, int d, int normalize, const int64_t* user_ids, const int64_t* item_ids,
                  const void* target, int target_i64, const int64_t* pos_idx, int P, const int64_t* pos_off,
                  const int64_t* pos_items, int64_t pos_users, int64_t B, int64_t N, int kind, int kind_mask, int num_negatives,
                  float sigma, float margin, const float* logq, int64_t logq_rows, int adam, int64_t step, const int64_t* step_dev,
                  float lr, float beta1, float beta2, float eps, float weight_decay, void* ws, size_t ws_bytes, float* out_losses,
                  mf_stream_t stream);

/* --------------------------------------------------------------- retrieval ---
 * Replaces `ItemProcessor.search` (xfmr_rec/data/lightning.py:237-259; LanceDB
 * cosine ANN with prefilter) by EXACT brute-force top-k over the indexed item
 * matrix: score = chain dot(q, item) (= 1 - cosine distance for unit-norm rows),
 * exclusion prefilter by per-query id lists (CSR: excl_off[Q+1], excl_idx[],
 * GLOBAL item row indices in any order; both nullable), best first, ties by lowest
 * item index.  `idx_base` is the global index of items[0] (row-sharded catalogs).
 * out_scores[Q,k] fp32, out_idx[Q,k] int64 global indices (-1 / -inf padding when
 * fewer than k candidates).  k <= 64. */
size_t mf_topk_ws_bytes(int64_t Q, int64_t N, int d, int k);
/* Host-only geometry query: the number of catalog chunks (one workgroup column each) mf_topk would use for this shape, and
 * the rows per chunk -- a chunk is staged through one 32-bit buffer descriptor, so rows_per_chunk * d * 4 <= ~4 GiB always
 * holds.  0 = unsupported shape. */
int mf_topk_chunks(int64_t Q, int64_t N, int d, int k, int64_t* rows_per_chunk);
int mf_topk(const float* q, int64_t Q, const float* items, int64_t N, int d, int k,
            const int64_t* excl_off, const int64_t* excl_idx, int64_t idx_base, void* ws,
            size_t ws_bytes, float* out_scores, int64_t* out_idx, mf_stream_t stream);

/* Merge G partial results (e.g. the all-gathered per-shard top-k of a row-sharded
 * catalog) part_*[G,Q,k] into the global top-k with the same order. */
int mf_topk_merge(const float* part_scores, const int64_t* part_idx, int G, int64_t Q, int k,
                  float* out_scores, int64_t* out_idx, mf_stream_t stream);

/* Sharded retrieval in ONE exchange: mf_topk_pack turns a rank's partial result (n = Q * k scores and LOCAL rows of its
 * shard; -1 = none) into one int64 per entry -- score bits << 32 | global row (local * stride + offset, < 2^32), -1 for
 * none -- and mf_topk_merge_packed merges G such blocks [G][Q][k] (as they arrive from one all-to-all) into the global
 * top-k, exactly like mf_topk_merge on (scores, global rows). */
int mf_topk_pack(const float* scores, const int64_t* rows, int64_t n, int64_t stride, int64_t offset, int64_t* out_packed,
                 mf_stream_t stream);
int mf_topk_merge_packed(const int64_t* packed, int G, int64_t Q, int k, float* out_scores, int64_t* out_idx,
                         mf_stream_t stream);

/* Small-batch form of mf_topk: Q <= 32 queries (the reference's search takes ONE query per call,
 * xfmr_rec/data/lightning.py:237-259; recommend, xfmr_rec/lightning.py:76-95).  A matrix-vector scan is
 * bandwidth-bound, so this path streams a BLOCKED copy of the catalog -- [block of 64 rows][16-byte chunk][row],
 * built once per index by mf_topk_blocked_build into mf_topk_blocked_bytes(N, d) bytes -- with one row per lane
 * and the canonical fmaf chain per (query, row): results are bit-identical to mf_topk's (scores, order, rows,
 * exclusion semantics, tail of -inf / -1).  Two launches, no memset, no scatter. */
size_t mf_topk_blocked_bytes(int64_t N, int d);
int mf_topk_blocked_build(const float* items, int64_t N, int d, float* out_blocked, mf_stream_t stream);
size_t mf_topk_small_ws_bytes(int64_t Q, int64_t N, int d, int k);
int mf_topk_small(const float* q, int64_t Q, const float* blocked, int64_t N, int d, int k,
                  const int64_t* excl_off, const int64_t* excl_idx, int64_t idx_base, void* ws,
                  size_t ws_bytes, float* out_scores, int64_t* out_idx, mf_stream_t stream);

/* Many-query form of mf_topk (Q >= ~64, d in {64, 128, 256}): the catalog is scanned TWICE with bf16 operands
 * (v_mfma_f32_32x32x16_bf16: 16x the fp32 MFMA rate, half the bytes) -- once for per-query bounds, once for the rows
 * above them -- and only those few dozen rows per query are scored with the canonical fp32 fmaf chain.  A rigorous
 * bound on the bf16 error (csrc/mf_topk_bf3.hip) makes the candidate set a superset of the true top k, so scores,
 * order, rows, exclusion semantics and the -inf / -1 tail are IDENTICAL to mf_topk's (finite inputs).  `index`:
 * mf_topk_bf3_index_bytes(N, d) bytes filled once per catalog by mf_topk_bf3_build (bf16 rows + the largest row
 * norm); `items` are the same fp32 rows mf_topk takes. */
size_t mf_topk_bf3_index_bytes(int64_t N, int d);
int mf_topk_bf3_build(const float* items, int64_t N, int d, void* index, size_t index_bytes, mf_stream_t stream);
size_t mf_topk_bf3_ws_bytes(int64_t Q, int64_t N, int d, int k);
int mf_topk_bf3(const float* q, int64_t Q, const float* items, const void* index, int64_t N, int d, int k,
                const int64_t* excl_off, const int64_t* excl_idx, int64_t idx_base, void* ws, size_t ws_bytes,
                float* out_scores, int64_t* out_idx, mf_stream_t stream);

/* Retrieval metrics @k on device, straight from the top-k output.  Replaces the
 * per-example torchmetrics updates of `update_metrics` / `get_metrics` (xfmr_rec/lightning.py:149-187,
 * :289-306: RetrievalNormalizedDCG / Recall / Precision / MAP / HitRate / MRR, top_k = 20).
 * topk_idx[Q,k]: retrieved item ids, best first (-1 = none); the targets of query q are
 * tgt_idx / tgt_rel[tgt_off[q] .. tgt_off[q+1]) (item id, rating); an unretrieved target ranks below
 * every retrieved item (the reference gives it -U(0,1): the same whenever retrieved scores are
 * positive).  out[q][6] = {ndcg, recall, precision, map, hit_rate, mrr} of query q with torchmetrics'
 * definitions: linear gain / log2 discount, binary relevance = rating > 0 for the other five,
 * precision divided by k, a query without positive target scores 0 everywhere. */
int mf_retrieval_metrics(const int64_t* topk_idx, int64_t Q, int k, const int64_t* tgt_off,
                         const int64_t* tgt_idx, const float* tgt_rel, float* out, mf_stream_t stream);

/* The batch producer on the device (replaces the host datapipe of
 * xfmr_rec/data/lightning.py:311-363 + data/load.py:38-141 for id-only towers).  The interaction list
 * pair_*[n_pairs] and the users' positive lists (CSR pos_off[num_users + 1], pos_items) stay in HBM;
 * examples start .. start + B - 1 of the reshuffled-every-epoch stream are written as one batch:
 * out_user[B], out_item[2 B] (the pairs' items, then B uniform negatives in 1 .. num_items - 1),
 * out_target[B], out_pos[B, P] (the user's positives, truncated to P, 0-padded on the right).
 * Counter-based: a batch depends on (seed, start) only (mf_data.hip; spec oracle/data.py). */
int mf_sample_batch(const int64_t* pair_user, const int64_t* pair_item, const float* pair_target, int64_t n_pairs,
                    const int64_t* pos_off, const int64_t* pos_items, int64_t num_items, uint64_t seed,
                    int64_t start, int64_t B, int P, int64_t* out_user, int64_t* out_item, float* out_target,
                    int64_t* out_pos, mf_stream_t stream);

/* ----------------------------------------------------------- sharded path ---
 * The reference has no collective; these serve the row-sharded design of distributed.py.
 * mf_init_rows fills a shard on its own device: local row l = global row row_start + l * row_stride of a
 * virtual [rows, d] table of std-scaled normal variates that are a pure function of (seed, global row,
 * column) -- no host copy of the whole table, values independent of the number of ranks.
 * mf_comm_*: RCCL (opened lazily) called on the CALLER'S stream: an exchange is a node between two kernels,
 * not a cross-stream join.  Bootstrap: rank 0 gets 128 bytes from mf_comm_unique_id, every rank receives
 * them by any side channel and calls mf_comm_create (a collective).  mf_comm_all_to_all_rows is a direct
 * all-to-all of row blocks (grouped send / recv: point-to-point xGMI links, no ring) with the per-peer row
 * counts on the host; mf_comm_all_gather the fixed-size gather of the retrieval queries. */
int mf_init_rows(float* table, int64_t n_local, int d, int64_t row_start, int64_t row_stride, uint64_t seed,
                 float std, mf_stream_t stream);
int mf_comm_unique_id(void* out128);
int mf_comm_create(int world, int rank, const void* id128, void** out_comm);
int mf_comm_destroy(void* comm);
int mf_comm_world(void* comm);
/* where the RCCL entry points came from: "shared: ..." (the librccl torch had already mapped: RTLD_NOLOAD),
 * "own: dlopen" (a process without one), "not loaded" */
const char* mf_comm_source(void);
int mf_comm_all_to_all_rows(void* comm, const void* send, const int64_t* send_rows_host, void* recv,
                            const int64_t* recv_rows_host, int64_t row_bytes, mf_stream_t stream);
int mf_comm_all_gather(void* comm, const void* send, void* recv, int64_t bytes, mf_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* MF_HIP_H */
