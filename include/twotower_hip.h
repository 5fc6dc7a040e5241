/*
 * twotower_hip.h — C ABI of the MI355X (gfx950) two-tower retrieval training hot path.
 *
 * This is the drop-in boundary.  Plain pointers and sizes only; no
 * torch / HIP C++ types.  All pointers are DEVICE pointers unless a comment says HOST.
 * `stream` is a hipStream_t passed as void* (NULL = the default stream).  Every entry
 * point enqueues on `stream` and returns without synchronising; it never allocates.
 * Return value: TT_OK (0) or a TT_ERR_* code; tt_last_error() gives the message of the
 * calling thread's last failure.  No C++ exception crosses this boundary.
 *
 * What each group replaces in the reference (citations into /root/reference):
 *   - The reference declares the training path but ships no code for it:
 *       src/models/__init__.py:1, src/training/__init__.py:1 (docstring stubs);
 *       entry point `train-model = "src.training.train:main"`   pyproject.toml:67;
 *       hyper-parameter schema                                  configs/data_config.yaml:54-71.
 *     The arithmetic would have come from tensorflow / tensorflow-recommenders
 *     (pyproject.toml:22,24).  The ops below are what those packages' CPU kernels
 *     (tf.gather, Dense matmul, tfrs.tasks.Retrieval, Keras SGD/Adagrad) would run.
 *   - Inputs are the int64 ids produced by
 *       scripts/data_processing/prepare_training_data.py:113-123,209-210 (user_idx,item_idx)
 *       src/data/preprocessor.py:478-491 (user_id_encoded, item_id_encoded, category_encoded).
 */
#ifndef TWOTOWER_HIP_H
#define TWOTOWER_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TT_ABI_VERSION 9

enum {
  TT_OK = 0,
  TT_ERR_INVALID_ARG = 1,   /* null pointer, bad size/alignment, unsupported dim            */
  TT_ERR_LAUNCH = 2,        /* hipLaunch / hip runtime error                                */
  TT_ERR_UNSUPPORTED = 3,   /* valid request this build does not implement                  */
  TT_ERR_WORKSPACE = 4      /* caller-provided workspace too small                          */
};

enum { TT_OPT_SGD = 0, TT_OPT_ADAGRAD = 1 };
enum { TT_IDS_UNIFORM = 0, TT_IDS_POWERLAW = 1 };

typedef void* tt_stream_t;

int tt_abi_version(void);
const char* tt_last_error(void);
/* sizeof of the structs below as this library was built (0 tt_train_step, 1 tt_dense_fwd_args, 2 tt_dense_bwd_args,
 * 3 tt_sparse_table_ids, 4 tt_dense_seg, 5 tt_id_buckets, 6 tt_dense_lookup; else -1): a binding checks its mirrors with it. */
int64_t tt_abi_struct_bytes(int32_t which);

/* ---------------------------------------------------------------------------------------
 * Built-in kernel timing (SURVEY.md §5 "Tracing / profiling": the reference has none).
 * tt_profile_enable("score_bwd,gather", 4096) makes every launch of the kernels carrying one
 * of those tags record a hipEvent pair on ITS OWN stream (capacity = launches kept per tag);
 * tags: fill, gather, sparse_plan, sparse_apply, dense_fwd, dense_bwd (dx+dw in one launch), dense_bwd_dx, dense_bwd_dw,
 * dense_update, optimizer (sparse + dense in one launch), score_fwd, score_bwd, score_fused, score_rank, score_aux, route, scatter_rows, encode_ids.
 * An empty string (or NULL) disables it.
 * tt_profile_read synchronises on the recorded events, writes up to `cap` durations in
 * milliseconds (launch order) to the HOST array `ms`, stores the number of durations written in
 * *count (HOST) and clears the tag.  Disabled = one predictable branch per launch.           */
int tt_profile_enable(const char* tags_csv, int32_t capacity_per_tag);
int tt_profile_read(const char* tag, float* ms, int32_t cap, int32_t* count);
/* Bracket only every stride-th launch of a tag (default 1 = every launch).  A hipEventRecord is a barrier packet
 * that costs the stream 4-7 us: a whole-step throughput measurement that also wants live kernel durations
 * samples them (bench.py: every 4th step).                                                                  */
int tt_profile_set_stride(int32_t stride);

/* ---------------------------------------------------------------------------------------
 * Synthetic inputs (SURVEY.md §8d "Synthetic inputs"; no reference counterpart).
 * Counter-based splitmix64; bit-identical to oracle/synth.py.
 *   value(i) = fl32(fl32(u(start+i) * scale) + lo),  u in [0,1) with 24 bits.          */
int tt_fill_uniform_f32(float* dst, int64_t n, uint64_t seed, uint64_t tensor_id,
                        int64_t start, float lo, float scale, tt_stream_t stream);
/* Rows row_start, row_start+row_stride, ... of a [*, dim] tensor whose flat element (r, d) uses counter
 * r*dim + d: the shard of a row-sharded table (owner = row % world) without materialising the whole. */
int tt_fill_uniform_rows_f32(float* dst, int64_t n_rows, int32_t dim, int64_t row_start, int64_t row_stride,
                             uint64_t seed, uint64_t tensor_id, float lo, float scale, tt_stream_t stream);
/* ids in [0,num_rows): variant TT_IDS_UNIFORM or TT_IDS_POWERLAW (floor(N*u^4)).        */
int tt_fill_ids_i64(int64_t* dst, int64_t n, uint64_t seed, uint64_t tensor_id,
                    int64_t start, int64_t num_rows, int32_t variant, tt_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * a6/a7 — id encoding (scripts/data_processing/prepare_training_data.py:113-123,209-210 sorted(unique)+enumerate+map;
 * src/data/preprocessor.py:478-491 LabelEncoder().fit_transform): codes[i] = rank of string i among the sorted
 * DISTINCT strings, int64.  `rows` is an [n, width] uint8 matrix of UTF-8 bytes, zero-padded on the right, width a
 * multiple of 8, no NUL inside a string (Python str order = code-point order = UTF-8 byte order; a shorter string
 * that is a prefix sorts first, like the zero padding).  *n_unique (device int32, may be NULL) receives the vocabulary size. */
int64_t tt_encode_ids_workspace_bytes(int64_t n);
int tt_encode_ids_u8(const uint8_t* rows, int64_t n, int32_t width, void* workspace, int64_t workspace_bytes,
                     int64_t* codes, int32_t* n_unique, tt_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * a1 — embedding lookup (Keras Embedding / tf.gather; configs/data_config.yaml:55).
 *   out[b, :] = table[ids[b], :]          table [num_rows, dim] f32 row-major, dim % 4 == 0
 * Ids outside [0,num_rows) produce a zero row and set *oob_flag (device int32, may be
 * NULL) to 1 — the caller turns that into the error TF's CPU gather raises.  The id -1 is a
 * PADDING slot (fixed-capacity all-to-all buffers of the row-sharded path): zero row, no flag.
 * The `2` form gathers the user and the item table in ONE launch.                        */
int tt_embedding_gather_f32(const float* table, int64_t num_rows, int32_t dim,
                            const int64_t* ids, int64_t n_ids, float* out,
                            int32_t* oob_flag, tt_stream_t stream);
int tt_embedding_gather2_f32(const float* table_a, int64_t rows_a, const int64_t* ids_a, float* out_a,
                             const float* table_b, int64_t rows_b, const int64_t* ids_b, float* out_b,
                             int32_t dim, int64_t n_ids, int32_t* oob_flag, tt_stream_t stream);
/* A further feature summed into a tower input (BASELINE configs[4]: the hashed category feature added to the
 * item tower's input): out[p,:] += table[ids[p],:] (one f32 add per element; id -1 / out of range adds
 * nothing, out of range sets the flag).  Its gradient rows are the tower-input gradient rows themselves.  */
int tt_embedding_gather_add_f32(const float* table, int64_t num_rows, int32_t dim,
                                const int64_t* ids, int64_t n_ids, float* out,
                                int32_t* oob_flag, tt_stream_t stream);
/* Hash feature ids: out[i] = FNV-1a-64(bytes of row i up to its first NUL) mod n_buckets, rows as for
 * tt_encode_ids_u8 (zero-padded [n, width] u8).  The reference names no hash (SURVEY.md Appendix A "not
 * specified anywhere"); this one is restated in oracle/hashing.py.                                          */
int tt_hash_bucket_u8(const uint8_t* rows_u8, int64_t n, int32_t width, int64_t n_buckets,
                      int64_t* out, tt_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * a5 — sparse optimizer on embedding rows (Keras SGD / Adagrad on IndexedSlices,
 * duplicates summed before the update; configs/data_config.yaml:63 learning_rate).
 *
 * tt_sparse_plan: stable sort of (id, position) by id.  Outputs
 *   sorted_ids [n_ids] int64 ascending, order [n_ids] int32 (positions, ascending inside equal ids).
 *   Every id outside [0, num_rows) (the padding id -1 included) is written to sorted_ids as the sentinel
 *   2^bits - 1 >= num_rows (bits = bit length of num_rows): such ids sort last, can never cut the run of a
 *   valid id, and the apply kernels skip them.
 * Only the ids are needed, so the plan can run before / beside the forward pass.  Up to
 * tt_sparse_plan_max_lds_ids() (16384) ids per table the sort is ONE launch of a hand-written LDS sort (n/128
 * workgroups per table, each sorting the ids of its own row range; tt_sparse_plan_batched sorts up to 4 tables —
 * user, item, hashed category — in that one launch) and needs no workspace; up to 16x that, 16384-id chunks are
 * radix-sorted by one launch into the workspace and a second launch merges them by rank; only longer lists fall back
 * to rocPRIM's device radix sort.
 *
 * tt_sparse_{sgd,adagrad}_f32: for every distinct id u (rows >= num_rows are skipped):
 *   g  = sum of grads[p, :] over the positions p of u in ascending p.  Order of the f32 adds: the run of u in
 *        the sorted list is cut at global multiples of 64 sorted slots; each piece is summed sequentially, then
 *        the pieces are added in order (a run inside one 64-slot block is a plain sequential sum).  The pieces
 *        of a run that crosses a block boundary are summed by different lane groups; the last one to arrive
 *        (a ticket in apply_ws) adds them in index order and applies the update — one launch in every case.
 *   apply_ws: tt_sparse_apply_workspace_bytes(n_ids, dim) bytes per table, 256-byte aligned (piece sums);
 *        ZERO it once after allocation — the kernels leave it zeroed.
 *   SGD:      w[u] = w[u] - fl(lr*g)
 *   Adagrad:  acc[u] += g*g ; w[u] -= fl(lr*g) / sqrt(acc[u] + eps)      (Keras 2.15)
 * In place.  The `2` forms update the user and the item table in one launch.             */
typedef struct tt_sparse_plan_args {
  const int64_t* ids;        /* [n_ids] */
  int64_t n_ids;
  int64_t num_rows;
  void* workspace;           /* tt_sparse_plan_workspace_bytes(n_ids) bytes; may be NULL when n_ids <= max_lds_ids */
  int64_t workspace_bytes;
  int64_t* sorted_ids;       /* [n_ids] out */
  int32_t* order;            /* [n_ids] out */
} tt_sparse_plan_args;
int32_t tt_sparse_plan_max_lds_ids(void);
int64_t tt_sparse_plan_workspace_bytes(int64_t n_ids);
int64_t tt_sparse_apply_workspace_bytes(int64_t n_ids, int32_t dim);
int tt_sparse_plan(const int64_t* ids, int64_t n_ids, int64_t num_rows,
                   void* workspace, int64_t workspace_bytes,
                   int64_t* sorted_ids, int32_t* order, tt_stream_t stream);
int tt_sparse_plan_batched(const tt_sparse_plan_args* tables, int32_t n_tables, tt_stream_t stream);
int tt_sparse_sgd_f32(float* table, int64_t num_rows, int32_t dim,
                      const float* grads, const int64_t* sorted_ids, const int32_t* order,
                      int64_t n_ids, float lr, void* apply_ws, tt_stream_t stream);
int tt_sparse_adagrad_f32(float* table, float* accum, int64_t num_rows, int32_t dim,
                          const float* grads, const int64_t* sorted_ids, const int32_t* order,
                          int64_t n_ids, float lr, float eps, void* apply_ws, tt_stream_t stream);
int tt_sparse_update2_f32(int32_t opt,
                          float* table_a, float* accum_a, int64_t rows_a, const float* grads_a,
                          const int64_t* sorted_ids_a, const int32_t* order_a,
                          float* table_b, float* accum_b, int64_t rows_b, const float* grads_b,
                          const int64_t* sorted_ids_b, const int32_t* order_b,
                          int32_t dim, int64_t n_ids, float lr, float eps,
                          void* apply_ws_a, void* apply_ws_b, tt_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * Row-sharded tables (multi-GPU): requester-side routing for the all-to-all exchange.
 * Row `id` lives on rank id % world at local row id / world.
 *   tt_route_by_owner_i64: stable partition of ids[n_ids] by owner into send_ids [world*cap] (local row ids,
 *     bucket o at [o*cap, (o+1)*cap), ascending position inside a bucket, padding -1) and
 *     pos_flat[p] = the slot of position p (or -1).  flags (device int32[2], may be NULL): [0] |= 1 when an id
 *     is outside [0,num_rows) (routed nowhere), [1] |= 1 when a bucket overflows `cap`.  world <= 16.
 *   tt_scatter_rows_f32: dst[idx[p], :] = src[p, :] for 0 <= idx[p] < dst_rows (per-position gradient rows
 *     into the send buffer; duplicates are summed later, on the owner, by the sparse optimizer).            */
int tt_route_by_owner_i64(const int64_t* ids, int64_t n_ids, int32_t world, int64_t num_rows, int32_t cap,
                          int64_t* send_ids, int64_t* pos_flat, int32_t* flags, tt_stream_t stream);
/* Several tables in one launch and ONE exchange: each rank keeps its shards of all tables in one combined
 * allocation (table t at rows [local_offset_t, local_offset_t + ceil(num_rows_t/world))), so the owner runs
 * one gather, one sort plan and one sparse update over every table's ids.  send_ids is
 * [world][n_tables][cap]: bucket (o, t) at ((o*n_tables + t)*cap ...), value id/world + local_offset_t,
 * padding -1; tables[t].pos_flat[p] = flat slot of position p in that buffer (or -1).  flags as above.   */
#define TT_ROUTE_MAX_TABLES 4
typedef struct {
  const int64_t* ids;       /* [n_ids] global row ids of this table            */
  int64_t num_rows;         /* global rows of this table                       */
  int64_t local_offset;     /* first row of this table in the owner's shard    */
  int64_t* pos_flat;        /* [n_ids] out                                     */
} tt_route_table;
int tt_route_tables_by_owner_i64(const tt_route_table* tables, int32_t n_tables, int64_t n_ids, int32_t world,
                                 int32_t cap, int64_t* send_ids, int32_t* flags, tt_stream_t stream);
int tt_scatter_rows_f32(const float* src, const int64_t* idx, int64_t n, int32_t dim,
                        float* dst, int64_t dst_rows, tt_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * a2 — MLP tower layers (Keras Dense; configs/data_config.yaml:56-57 *_tower_dims).
 * All matrices f32 row-major.  x [m,k], w [k,n] (Keras kernel layout), b [n], y [m,n].
 * k % 4 == 0 and n % 4 == 0; m arbitrary.  f32-input MFMA (exact f32 products).
 *
 *   fwd:  y = x@w + b ; relu != 0 applies max(.,0)
 *   bwd:  dz is dLoss/d(pre-activation) of this layer (the caller's upstream gradient,
 *         already masked by this layer's own ReLU — see dx_relu_src);
 *         dx = dz@w^T, and if dx_relu_src != NULL (the [m,k] output of the previous
 *         ReLU layer, i.e. x itself) dx is multiplied by (dx_relu_src > 0) so that it
 *         is directly the previous layer's dz;  dx may be NULL to skip it; dw_slabs AND db_slabs may
 *         both be NULL to skip the weight gradients (a caller that wants every dx before any dw, so the
 *         embedding gradients can travel while the dw GEMMs run, calls the layer twice).
 *         dw_slabs [n_slabs, k, n] and db_slabs [n_slabs, n] receive split-K partial
 *         sums of x^T@dz and colsum(dz); the dense update sums them in slab order.
 *         n_slabs = tt_dense_bwd_num_slabs(m).                                            */
int tt_dense_fwd_f32(const float* x, const float* w, const float* b, float* y,
                     int64_t m, int32_t k, int32_t n, int32_t relu, tt_stream_t stream);
/* With inverted dropout on the output (configs/data_config.yaml:58 dropout_rate): element (r, c) is dropped iff
 * the top 24 bits of the counter-based hash of (seed, tensor_id, counter_offset + r*n + c) are < round(rate*2^24);
 * kept elements are multiplied by 1/(1-rate).  Reproducible: oracle/synth.py::dropout_keep restates it.  The
 * backward pass needs no mask tensor: (y > 0) marks the kept, active units; pass dx_scale = 1/(1-rate) below. */
int tt_dense_fwd_dropout_f32(const float* x, const float* w, const float* b, float* y,
                             int64_t m, int32_t k, int32_t n, int32_t relu,
                             float drop_rate, uint64_t seed, uint64_t tensor_id, uint64_t counter_offset,
                             tt_stream_t stream);
int32_t tt_dense_bwd_num_slabs(int64_t m);
int tt_dense_bwd_f32(const float* x, const float* w, const float* dz,
                     float* dx, const float* dx_relu_src,
                     float* dw_slabs, float* db_slabs,
                     int64_t m, int32_t k, int32_t n, tt_stream_t stream);
/* Same, with dx additionally multiplied by dx_scale where dx_relu_src > 0 (dropout on the previous layer). */
int tt_dense_bwd_scaled_f32(const float* x, const float* w, const float* dz,
                            float* dx, const float* dx_relu_src, float dx_scale,
                            float* dw_slabs, float* db_slabs,
                            int64_t m, int32_t k, int32_t n, tt_stream_t stream);

/* Batched forms: the same layer of the user AND the item tower (identical shapes) in one launch each —
 * fwd: 1 launch, bwd: 1 launch (the dx tiles and the dw+db tiles of the layer side by side; 2 launches when
 * only dx or only dw is asked for) — instead of twice as many half-size launches.
 * `probs` is a HOST array of n_probs (1 or 2) entries.                                                     */
/* Embedding lookup FUSED into the layer (a1 + a2, the tower's first Dense): with a non-NULL `ids` the layer's input
 * x is never materialised — row r of x is row ids[r] of `table` [table_rows, k] (+ row ids2[r] of `table2`, the
 * hashed-category feature summed into the item tower's input; one f32 add per element), read straight into the
 * GEMM's LDS tiles by the forward pass (x @ w) and by the backward pass's dW = x^T @ dz.  Ids outside
 * [0, table_rows) give a zero row; -1 is a silent padding id, any other sets *oob_flag.  All-zero = no lookup.
 * Needs m <= 32768 (the ids of one dW split are staged in LDS).                                                  */
/* Row-range id lists (ABI v9; optional - all-zero = none).  tt_optimizer_step_ids_f32 gives every sorting workgroup one row
 * range [g*width, (g+1)*width) of a table, and each of them used to read ALL the batch's ids to find its own ~64.  The forward
 * lookup reads every id anyway: with `buckets` it also appends (id - g*width, batch position) to range g's list - one
 * returning atomic add on counts[g], one 8-byte store - and the optimizer launch of the SAME step, given the same descriptor,
 * reads only its own list (and falls back to the scan for a range whose list overflowed `cap`: counts[g] keeps counting).
 *   counts [groups][64] uint32, word 0 of each 256-byte line used (a counter per line: device-scope atomics on one line
 *          serialise): entries appended; ZERO before the first step - the optimizer launch resets every counter it reads
 *   pairs  [groups][cap] uint64: bits 0..31 local key, 32..47 batch position, 48..63 generation (entries of another
 *          generation - a forward pass whose optimizer step never ran - are ignored)
 * groups / width / cap must be the ones tt_optimizer_ids_geometry reports for the step (else the optimizer ignores the lists).
 * Lists exist for n_ids <= 16384, dim <= 128; filled by tt_tower_fwd2_batched_f32 and by tt_dense_fwd_batched_f32 (layer 0 with a
 * lookup).  One descriptor per table; tt_id_buckets_workspace_bytes() bytes hold one.                                          */
typedef struct tt_id_buckets {
  uint32_t* counts; uint64_t* pairs;
  int32_t groups; uint32_t width; int32_t cap; uint32_t gen;
} tt_id_buckets;
typedef struct tt_dense_lookup {
  const float* table;  const int64_t* ids;  int64_t table_rows;
  const float* table2; const int64_t* ids2; int64_t table2_rows;     /* optional */
  int32_t* oob_flag;                                                 /* optional */
  tt_id_buckets buckets;                                             /* optional: row-range lists of `ids` (forward pass only) */
} tt_dense_lookup;
/* ReLU sign bits (optional, n % 32 == 0): the forward pass can write, beside y, one bit per element — word
 * [row][col / 32] of a [m, n/32] uint32 array, bit col % 32 = (y[row][col] > 0) — and the backward pass of the NEXT layer
 * takes them as its dx mask (`dx_relu_bits`, [m, k/32]) instead of re-reading the whole activation through
 * `dx_relu_src`: 1/32 of the mask bytes (cfg3: 0.5 MB instead of 16.8 MB per step) and no 4-byte strided loads in
 * front of the dx tiles.  Same mask, same results bit for bit.                                                       */
typedef struct tt_dense_fwd_args {
  const float* x; const float* w; const float* b; float* y;
  uint64_t dropout_tensor_id;       /* counter stream of this problem's dropout mask (ignored at rate 0) */
  tt_dense_lookup lookup;           /* x may be NULL when lookup.ids is given */
  uint32_t* relu_bits;              /* optional [m, n/32]: sign bits of y (after ReLU / dropout), see above */
} tt_dense_fwd_args;
typedef struct tt_dense_bwd_args {
  const float* x; const float* w; const float* dz; float* dx; const float* dx_relu_src;
  float* dw_slabs; float* db_slabs;
  tt_dense_lookup lookup;           /* x may be NULL when lookup.ids is given (dW reads the table rows) */
  const uint32_t* dx_relu_bits;     /* optional [m, k/32]: the dx mask as sign bits; takes precedence over dx_relu_src */
} tt_dense_bwd_args;
int tt_dense_fwd_batched_f32(const tt_dense_fwd_args* probs, int32_t n_probs, int64_t m, int32_t k, int32_t n,
                             int32_t relu, float drop_rate, uint64_t seed, uint64_t counter_offset,
                             tt_stream_t stream);
int tt_dense_bwd_batched_f32(const tt_dense_bwd_args* probs, int32_t n_probs, float dx_scale,
                             int64_t m, int32_t k, int32_t n, tt_stream_t stream);

/* The forward pass of a TWO-layer tower in ONE launch (csrc/tower.hip): h = relu(x @ w0 + b0) [dropout], y = h @ w1 + b1 for
 * the user and the item tower together.  A workgroup owns 32 batch rows for both layers; the hidden tile stays in LDS (it
 * is still written to layer0[i].y - and its sign bits to layer0[i].relu_bits - because the backward pass reads them).
 * layer0[i] / layer1[i] are the per-layer descriptions tt_dense_fwd_batched_f32 takes (layer1[i].x must be layer0[i].y or
 * NULL; only layer 0 may carry a lookup); dropout applies to the hidden layer, with layer0[i].dropout_tensor_id and
 * counter_offset as there.  Bit-identical to the two tt_dense_fwd_batched_f32 calls.  Shapes: k0 % 32 == 0, k0 <= 512,
 * h and n1 in {128, 256} (tt_tower_fwd2_supported; otherwise TT_ERR_UNSUPPORTED - call the layers one by one).         */
int32_t tt_tower_fwd2_supported(int64_t m, int32_t k0, int32_t h, int32_t n1);
int tt_tower_fwd2_batched_f32(const tt_dense_fwd_args* layer0, const tt_dense_fwd_args* layer1, int32_t n_probs, int64_t m,
                              int32_t k0, int32_t h, int32_t n1, float drop_rate, uint64_t seed, uint64_t counter_offset,
                              tt_stream_t stream);

/* Dense parameter update over up to TT_MAX_DENSE_SEGS segments in one launch.
 *   g = sum_s grad_slabs[s*slab_stride + i] (s ascending) + 2*l2*w[i]
 *   SGD: w -= fl(lr*g);  Adagrad: acc += g*g; w -= fl(lr*g)/sqrt(acc+eps)
 * If grad_out != NULL the summed gradient (WITHOUT the l2 term) is written there and,
 * when apply == 0, nothing else happens (the multi-GPU path all-reduces grad_out and
 * calls again with n_slabs = 1).  `segs` is a HOST array.                                 */
#define TT_MAX_DENSE_SEGS 16
typedef struct tt_dense_seg {
  float* param;              /* [count]                                   */
  float* accum;              /* [count] Adagrad accumulator or NULL (SGD) */
  const float* grad_slabs;   /* [n_slabs][slab_stride]                    */
  float* grad_out;           /* [count] or NULL                           */
  int64_t count;
  int64_t slab_stride;
  int32_t n_slabs;
  float l2;                  /* l2_regularization (configs/data_config.yaml:59); 0 for biases */
} tt_dense_seg;
int tt_dense_update_f32(const tt_dense_seg* segs, int32_t n_segs, int32_t opt, int32_t apply,
                        float lr, float eps, tt_stream_t stream);
/* tt_dense_bwd_batched_f32 with a dense parameter update RIDING in the same launch (ABI v9): `segs` (apply = 1 semantics of
 * tt_dense_update_f32) must belong to ANOTHER layer - in a backward pass, the layer above, whose gradient slabs the previous
 * backward launch completed; a segment whose slabs this launch writes or whose weights it reads is refused.  The update's
 * blocks are the first workgroups of the dx+dw launch: no launch of their own, and their slab traffic (cfg3: 9 of the step's
 * 18 MB) leaves the optimizer launch, whose HBM burst is the embedding rows'.  Same arithmetic, bit for bit.  With dx or dw
 * alone (two launches anyway) the update runs as its own launch behind them.                                            */
int tt_dense_bwd_batched_update_f32(const tt_dense_bwd_args* probs, int32_t n_probs, float dx_scale, int64_t m, int32_t k, int32_t n,
                                    const tt_dense_seg* segs, int32_t n_segs, int32_t opt, float lr, float eps, tt_stream_t stream);

/* Two layers' backward passes in ONE launch (r04): `upper` = layer l (k1 -> n), `lower` = layer l-1 (k0 -> k1), upper[i].dx
 * must BE lower[i].dz.  The same tiles as two tt_dense_bwd_batched_f32 calls, results identical bit for bit; the lower layer's
 * tiles wait, inside the launch, for the 64-row blocks of dz they read (agent-scope release / acquire on a counter per row
 * block in `workspace`: tt_tower_bwd2_workspace_bytes(m) bytes, 256-byte aligned, ZEROED ONCE - the launch leaves it zeroed).
 * What it saves is the boundary between the two launches (the platform's ~4 us + one launch's tail + the other's ramp).  The
 * wait is bounded: if it ever ran out, int32 word [4 * (m / 64)] of the workspace is set and the results of that step are wrong.
 * Both layers need dx and dw_slabs; only the lower layer may carry the fused lookup.  tt_tower_bwd2_supported: m % 64 == 0
 * and the batch splits of the dW GEMMs (tt_dense_bwd_num_slabs) whole numbers of 64-row blocks.                         */
int32_t tt_tower_bwd2_supported(int64_t m, int32_t k0, int32_t k1, int32_t n);
int64_t tt_tower_bwd2_workspace_bytes(int64_t m);
int tt_tower_bwd2_batched_f32(const tt_dense_bwd_args* upper, const tt_dense_bwd_args* lower, int32_t n_probs, float dx_scale_upper,
                              float dx_scale_lower, int64_t m, int32_t k0, int32_t k1, int32_t n, void* workspace, tt_stream_t stream);

/* The whole optimizer of a train step in ONE launch: the fused sparse update of up to 3 embedding tables (user, item,
 * hashed category; same dim and n_ids, each with its own sort plan and apply workspace) AND the dense update of every
 * tower segment (apply = 1 semantics of tt_dense_update_f32).  Same arithmetic, bit for bit, as tt_sparse_update2_f32 /
 * tt_sparse_{sgd,adagrad}_f32 followed by tt_dense_update_f32; the two halves are independent and memory-bound, so one
 * launch overlaps them and saves a launch boundary.  `tables` and `segs` are HOST arrays.                          */
typedef struct tt_sparse_table {
  float* table; float* accum;            /* [rows, dim]; accum NULL for SGD        */
  int64_t rows;
  const float* grads;                    /* [n_ids, dim] per-position gradient rows */
  const int64_t* sorted_ids; const int32_t* order;   /* tt_sparse_plan outputs      */
  void* apply_ws;                        /* tt_sparse_apply_workspace_bytes(n_ids, dim), zeroed once */
} tt_sparse_table;
int tt_optimizer_step_f32(int32_t opt, const tt_sparse_table* tables, int32_t n_tables, int32_t dim, int64_t n_ids,
                          const tt_dense_seg* segs, int32_t n_segs, float lr, float eps, tt_stream_t stream);

/* The same step from the RAW ids (n_ids <= tt_optimizer_ids_max_ids() = 65536, else TT_ERR_UNSUPPORTED): no
 * tt_sparse_plan launch and no sorted ids in HBM — the sorting workgroups of each table (one per row range) apply the
 * update to their own rows inside the one optimizer launch.  Same results, bit for bit, as tt_sparse_plan_batched +
 * tt_optimizer_step_f32 (same piece boundaries at global multiples of 64 sorted slots).  Lists of more than
 * tt_sparse_plan_max_lds_ids() (16384) ids take the long-list kernel (r04: ids scanned in chunks, the LDS list keeps
 * 16384 slots; a row range that holds more ids than that - a degenerate batch - is sorted in a global scratch inside
 * apply_ws, which tt_sparse_apply_workspace_bytes sizes for it).                                                   */
int32_t tt_optimizer_ids_max_ids(void);
typedef struct tt_sparse_table_ids {
  float* table; float* accum;            /* [rows, dim]; accum NULL for SGD        */
  int64_t rows;
  const float* grads;                    /* [n_ids, dim] per-position gradient rows */
  const int64_t* ids;                    /* [n_ids] the batch's ids, unsorted       */
  void* apply_ws;                        /* tt_sparse_apply_workspace_bytes(n_ids, dim) */
  tt_id_buckets buckets;                 /* optional (ABI v9): the row-range lists this step's forward lookup filled */
} tt_sparse_table_ids;
/* The row ranges tt_optimizer_step_ids_f32 will use for these tables (HOST outputs, [n_tables] each): table t is cut into
 * groups[t] ranges of width[t] rows; *cap = entries per list (0: this shape takes no lists - dim > 128 or n_ids > 16384).   */
int tt_optimizer_ids_geometry(const int64_t* table_rows, int32_t n_tables, int32_t dim, int64_t n_ids,
                              const tt_dense_seg* segs, int32_t n_segs, int32_t* groups, uint32_t* width, int32_t* cap);
int64_t tt_id_buckets_workspace_bytes(void);      /* one table's counts + pairs, 256-byte aligned inside */
/* A trainer's skew probe (r04): out_max[t] (DEVICE, [n_tables]) = the largest number of this batch's ids that fall into ONE
 * of table t's row ranges (the ranges of tt_optimizer_ids_geometry).  The one-launch optimizer gives a range to one
 * workgroup: ids uniform over the rows put n_ids / groups in each, a vocabulary in order of frequency puts thousands into
 * the first (cfg3, ids ~ rows * u^4: launch 12 -> 169 us).  Run every few dozen steps, copy the words to pinned host memory
 * without waiting, and take tt_sparse_plan_batched + tt_optimizer_step_f32 while a range holds more than ~384 ids.
 * ids: HOST array of n_tables device pointers.  One launch, one workgroup per table.                                    */
int tt_id_range_load(const int64_t* const* ids, const int64_t* table_rows, int32_t n_tables, int32_t dim, int64_t n_ids,
                     const tt_dense_seg* segs, int32_t n_segs, int32_t* out_max, tt_stream_t stream);
int tt_optimizer_step_ids_f32(int32_t opt, const tt_sparse_table_ids* tables, int32_t n_tables, int32_t dim, int64_t n_ids,
                              const tt_dense_seg* segs, int32_t n_segs, float lr, float eps, tt_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * The WHOLE train step as one call (ABI v8): what `train-model` (pyproject.toml:67, src/training/train.py - declared,
 * never written) would run per batch.  A HOST struct of pointers describes the step once - every buffer is caller-owned
 * and fixed from step to step; per step the caller only rewrites the id pointers, the dropout row counter and the optional
 * per-pair inputs - and tt_train_step_f32 enqueues, on `stream`, exactly the launches the separate entry points would:
 *   for each layer l:            tt_dense_fwd_batched_f32(fwd[l], 2, ...)       (layer 0 reads the embedding rows itself;
 *                                two-layer towers of supported shapes: ONE tt_tower_fwd2_batched_f32 launch instead)
 *   scorer + loss + dq, dc:      tt_retrieval_fwd_bwd_f32 / _bf16x3_f32         (q, c = fwd[n_layers-1][*].y; dq, dc = bwd[n_layers-1][*].dz)
 *   for each layer l, last first: tt_dense_bwd_batched_f32(bwd[l], 2, ...)
 *   optimizer:                   tt_optimizer_step_ids_f32(tables, segs)       (sort + duplicate sums + sparse update + dense update)
 * Eight launches for two 2-layer towers of a fused shape (nine otherwise); no other work, no allocation, no synchronisation.
 * It exists for the HOST side: a caller that reaches the library through an FFI (ctypes: ~7 us per call) pays that once per
 * step instead of once per launch - cfg1 (B 256) is 7 launches of a few us each.  Results are identical to the separate
 * calls, bit for bit.
 * Both towers must have the same layer shapes (the batched launches); batch <= 65536 (tt_optimizer_step_ids_f32) and
 * <= 32768 with a fused lookup.                                                                                       */
#define TT_MAX_TOWER_LAYERS 8
typedef struct tt_train_step {
  int64_t batch;                                   /* pairs per step = rows of every activation                    */
  int32_t n_layers;                                /* Dense layers per tower                                       */
  int32_t dims[TT_MAX_TOWER_LAYERS + 1];           /* embedding_dim, then the width of every layer                 */
  tt_dense_fwd_args fwd[TT_MAX_TOWER_LAYERS][2];   /* [layer][0 = user tower, 1 = item tower]; hidden layers: ReLU */
  tt_dense_bwd_args bwd[TT_MAX_TOWER_LAYERS][2];
  float dropout_rate;                              /* configs/data_config.yaml:58; 0 = none                        */
  uint64_t dropout_seed;
  uint64_t dropout_row0;                           /* first global batch row of this step: layer l's counter offset = row0 * dims[l+1] */
  int32_t scorer_precision;                        /* 0 = exact f32 products, 1 = bf16x3                           */
  float inv_temperature;                           /* 1 / retrieval.temperature (configs/data_config.yaml:70)      */
  const float* sample_weight;                      /* [batch] or NULL                                              */
  const float* cand_prob;                          /* [batch] or NULL (candidate_sampling_probability)             */
  const int64_t* cand_ids;                         /* [batch] or NULL (remove_accidental_hits)                     */
  void* retrieval_ws; int64_t retrieval_ws_bytes;  /* tt_retrieval_workspace_bytes(batch, batch, dims[n_layers])   */
  float* lse; float* per_row; float* loss;         /* [batch], [batch], [1]                                        */
  int32_t opt;                                     /* TT_OPT_SGD / TT_OPT_ADAGRAD                                  */
  int32_t n_tables;                                /* 2, or 3 with the hashed category table                       */
  tt_sparse_table_ids tables[3];
  int32_t n_segs;
  tt_dense_seg segs[TT_MAX_DENSE_SEGS];
  float lr, eps;
  void* id_bucket_ws;                              /* optional (ABI v9): n_tables * tt_id_buckets_workspace_bytes() bytes, 256-byte   */
  int64_t id_bucket_ws_bytes;                      /* aligned, ZEROED once: the forward lookup hands the optimizer launch its row-range id lists */
} tt_train_step;
int tt_train_step_f32(const tt_train_step* step, tt_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * a3 + a4 — batched dot-product scorer fused with the in-batch sampled-softmax loss
 * (tfrs.tasks.Retrieval.call: matmul(q, c^T) / temperature, optional sampling-probability
 * correction and accidental-hit removal, CategoricalCrossentropy(from_logits=True,
 * reduction=SUM); configs/data_config.yaml:69-70).  Softmax probabilities never reach HBM (see tt_retrieval_fwd_bwd_f32 for the
 * raw dot products the exact-f32 training entry keeps between its passes).
 *
 *   s_ij   = <q_i, c_j> * inv_temperature  - log(clip(cand_prob_j, 1e-6, 1))
 *            (+ -inf where cand_ids_j == cand_ids_{i+diag_offset} and j != i+diag_offset)
 *   lse_i  = log sum_j exp(s_ij)
 *   row_i  = w_i * (lse_i - s_{i,i+diag_offset}) ;  loss = sum_i row_i
 *   dq     = grad_scale * sum_j  w_i/T (softmax_ij - [j == i+diag_offset]) c_j ;  dc likewise.
 *
 * q [nq,dim], c [nc,dim] f32 row-major; dim in {32,64,128,256}; nq + diag_offset <= nc.
 * sample_weight [nq], cand_prob [nc], cand_ids [nc] (int64) may each be NULL.
 * hard_thr [nq] (may be NULL): per-query thresholds from tt_retrieval_hard_negative_thresholds_f32 — only the
 * positive and the negatives scoring at or above the threshold take part (tfrs num_hard_negatives).
 * Outputs: lse [nq], per_row [nq], loss [1]; dq [nq,dim], dc [nc,dim].
 * Workspace: tt_retrieval_workspace_bytes(nq, nc, dim) bytes, 256-byte aligned.  For the fused training entries this
 * INCLUDES the raw dot products of pass 1, ceil(nq/32) * ceil(nc/32) blocks of 4 KB = 4*nq*nc bytes (268 MB at 8192 x
 * 8192, 4.3 GB at 32768 x 32768; ABI v5 and later - a workspace sized by an older library is refused with
 * TT_ERR_WORKSPACE).  Both precisions keep them, at every dim (ABI v8; bf16x3 at dim 256 recomputed through v7).    */
int64_t tt_retrieval_workspace_bytes(int64_t nq, int64_t nc, int32_t dim);
/* Host query: the number of slices a scorer pass cuts its streamed side into for this shape (pass 0: stationary q - loss + dq,
 * forward, rank; 1: stationary c - dc with the products recomputed; 2: dc from the stored products).  r04: chosen by a model of
 * the launch (rounds of resident workgroups) instead of "double until 512 workgroups" - batch 8200 ran 0.879 ms against 0.558
 * for 8192 because 8 of 520 workgroups ran a second round alone; every power-of-two shape keeps its r03 value.               */
int32_t tt_retrieval_num_splits(int64_t nq, int64_t nc, int32_t dim, int32_t pass);
/* the part of it the forward-only (tt_retrieval_fwd_f32) and separate-backward (tt_retrieval_bwd_f32) entries need: no
 * [nq][nc] logit buffer (only the fused training entries keep the raw dot products between their two passes) */
int64_t tt_retrieval_fwd_workspace_bytes(int64_t nq, int64_t nc, int32_t dim);
/* What tt_retrieval_rank_f32 alone needs (no gradient slabs: those make the full workspace as large as the candidate
 * corpus once nc >= 65536).                                                                                   */
int64_t tt_retrieval_rank_workspace_bytes(int64_t nq, int64_t nc, int32_t dim);
int tt_retrieval_fwd_f32(const float* q, const float* c, int64_t nq, int64_t nc, int32_t dim,
                         int64_t diag_offset, float inv_temperature,
                         const float* sample_weight, const float* cand_prob, const int64_t* cand_ids,
                         const float* hard_thr,
                         void* workspace, int64_t workspace_bytes,
                         float* lse, float* per_row, float* loss, tt_stream_t stream);
int tt_retrieval_bwd_f32(const float* q, const float* c, int64_t nq, int64_t nc, int32_t dim,
                         int64_t diag_offset, float inv_temperature,
                         const float* sample_weight, const float* cand_prob, const int64_t* cand_ids,
                         const float* hard_thr, const float* lse, float grad_scale,
                         void* workspace, int64_t workspace_bytes,
                         float* dq, float* dc, tt_stream_t stream);

/* tfrs.tasks.Retrieval(num_hard_negatives=k) / tfrs.layers.loss.HardNegativeMining: thr[i] separates the k
 * highest-scoring negatives of query i (after temperature, sampling-probability correction and accidental-hit
 * removal) from the rest (midpoint between the k-th and the next lower logit; ties at the k-th value are all kept;
 * fewer than k negatives: everything is kept).  scratch: nq*nc floats (a logit row block, materialised).             */
int tt_retrieval_hard_negative_thresholds_f32(const float* q, const float* c, int64_t nq, int64_t nc, int32_t dim,
                                              int64_t diag_offset, float inv_temperature,
                                              const float* cand_prob, const int64_t* cand_ids,
                                              int32_t num_hard_negatives,
                                              void* workspace, int64_t workspace_bytes,
                                              float* scratch, int64_t scratch_bytes,
                                              float* thr, tt_stream_t stream);

/* Fused training form: loss AND both gradients in two passes over the logits instead of three
 * (pass 1: online softmax with the candidate-weighted sum -> lse, per_row, loss, dq;  pass 2: dc).
 * Same semantics and outputs as tt_retrieval_fwd_f32 followed by tt_retrieval_bwd_f32.
 * Both entries keep the raw dot products [nq][nc] (f32) in the workspace between the passes — pass 2 reads them
 * back instead of recomputing them (half its matrix-pipe work; 6*nq*nc*dim executed FLOPs in total instead of 8) —
 * which is why tt_retrieval_workspace_bytes includes 4*nq*nc bytes; tt_retrieval_fwd_workspace_bytes is enough for the
 * forward-only and separate-backward entries.                                                                        */
int tt_retrieval_fwd_bwd_f32(const float* q, const float* c, int64_t nq, int64_t nc, int32_t dim,
                             int64_t diag_offset, float inv_temperature,
                             const float* sample_weight, const float* cand_prob, const int64_t* cand_ids,
                             const float* hard_thr, float grad_scale, void* workspace, int64_t workspace_bytes,
                             float* lse, float* per_row, float* loss, float* dq, float* dc,
                             tt_stream_t stream);

/* The same fused training form with every matrix product on the bf16 matrix cores through a three-way split of the
 * f32 operands (x = hi + mid + lo, each bf16, residuals exact): "bf16x3", an f32-EMULATED precision — 6 bf16 products
 * per logit term (error ~2^-24 relative), 3 per gradient term (~2^-16), f32 accumulation, f32 softmax; inputs and
 * outputs stay f32 and the results meet the same 1e-4 bars as the exact-f32 form (tests/test_gpu_parity.py).  2.7x
 * fewer matrix-pipe cycles than v_mfma_f32_32x32x2_f32.  dim must be 128 or 256 (else TT_ERR_UNSUPPORTED: use the f32 form). */
int tt_retrieval_fwd_bwd_bf16x3_f32(const float* q, const float* c, int64_t nq, int64_t nc, int32_t dim,
                                    int64_t diag_offset, float inv_temperature,
                                    const float* sample_weight, const float* cand_prob, const int64_t* cand_ids,
                                    const float* hard_thr, float grad_scale, void* workspace, int64_t workspace_bytes,
                                    float* lse, float* per_row, float* loss, float* dq, float* dc,
                                    tt_stream_t stream);
/* The forward-only (validation) pass and the metric (rank) pass in the same f32-emulated precision: both are pure GEMM1, so
 * the 6-product bf16 split (2^-24 relative on every logit) is the whole kernel.  Arguments as tt_retrieval_fwd_f32 /
 * tt_retrieval_rank_f32; dim in {128, 256}.                                                                          */
int tt_retrieval_fwd_bf16x3_f32(const float* q, const float* c, int64_t nq, int64_t nc, int32_t dim,
                                int64_t diag_offset, float inv_temperature, const float* sample_weight,
                                const float* cand_prob, const int64_t* cand_ids, const float* hard_thr,
                                void* workspace, int64_t workspace_bytes, float* lse, float* per_row, float* loss,
                                tt_stream_t stream);
int tt_retrieval_rank_bf16x3_f32(const float* q, const float* c, int64_t nq, int64_t nc, int32_t dim,
                                 float inv_temperature, const float* cand_prob, const int64_t* pos_index,
                                 void* workspace, int64_t workspace_bytes, int32_t* rank, tt_stream_t stream);

/* Retrieval metrics (configs/data_config.yaml:71 top_k_eval; tfrs.metrics.FactorizedTopK's role):
 * rank[i] = number of candidates j != pos_index[i] with s_ij > s_{i,pos_index[i]} over ALL nc candidates
 * (nc may be the whole item corpus; nq <= or > nc both allowed).  Recall@K = mean(rank < K),
 * NDCG@K = mean([rank < K] / log2(rank + 2)).  Workspace: tt_retrieval_rank_workspace_bytes(nq, nc, dim).     */
int tt_retrieval_rank_f32(const float* q, const float* c, int64_t nq, int64_t nc, int32_t dim,
                          float inv_temperature, const float* cand_prob, const int64_t* pos_index,
                          void* workspace, int64_t workspace_bytes, int32_t* rank, tt_stream_t stream);
/* The same metric pass on the IN-BATCH candidates - tfrs.tasks.Retrieval(batch_metrics=[top-k categorical accuracy]):
 * rank[i] = number of candidates j != i + diag_offset whose logit (after temperature, -log clip(cand_prob) and, with
 * cand_ids, accidental-hit removal: the scores the loss sees) is strictly above the positive's; top-k accuracy =
 * mean(rank < k).  Workspace: tt_retrieval_rank_workspace_bytes(nq, nc, dim).                                        */
int tt_retrieval_batch_rank_f32(const float* q, const float* c, int64_t nq, int64_t nc, int32_t dim, int64_t diag_offset,
                                float inv_temperature, const float* cand_prob, const int64_t* cand_ids,
                                void* workspace, int64_t workspace_bytes, int32_t* rank, tt_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* TWOTOWER_HIP_H */
