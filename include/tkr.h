/* tkr.h -- C ABI of libtkr_hip.so: the MI355X (gfx950) hot path of top-k-rec.
 *
 * The reference (domainxz/top-k-rec) has no FFI: its boundary is the Python class API
 * (single/rec.py, single/bpr.py, single/vbpr.py) and the evaluate.py CLI, and internally the
 * two seams where Python hands work to a numerics runtime.  Each entry point below replaces
 * one such seam; the citation names the reference call site (file:line under the reference
 * root) whose work it does.
 *
 * Conventions
 *   - every function returns int: 0 = ok, > 0 = hipError_t, < 0 = TKR_E* below; nothing throws
 *   - all buffers are caller-owned DEVICE pointers (e.g. torch tensor.data_ptr()) with explicit
 *     sizes; the library allocates no persistent memory
 *   - `stream` is a hipStream_t passed as void*; all calls are asynchronous and stream-ordered;
 *     scalar results are written to device memory
 *   - ids are int32, parameters fp32
 */
#ifndef TKR_H
#define TKR_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TKR_VERSION 100 /* 0.1.0 */
#define TKR_E_INVAL (-1)
#define TKR_E_UNSUPPORTED (-2)

int tkr_version(void);

/* ---- K1: (u,i,j) draw + batch plan ---------------------------------------------------------
 * Replaces BPR._uniform_user_sampling (single/bpr.py:155-165) and the gradient de-duplication
 * bookkeeping TF does inside sess.run (single/bpr.py:100,141).
 *   tr_users[n_tr]           users with >= 1 positive (single/bpr.py:65)
 *   row_ptr[n_users+1], pos_cols[nnz]   tr_data as CSR, file order, duplicates kept (bpr.py:167-171)
 *   cols_sorted[nnz]         per-row ascending copy (membership test of bpr.py:163)
 *   seed, first_triplet      counter-based stream position (triplet g = first_triplet + b*B + t);
 *   ctl                      optional device int64[1]: batch index added to the stream position at
 *                            run time (lets one captured hipGraph walk through an epoch)
 *   out_u/out_i/out_j        [n_batches*B]
 *   task                     [n_batches][3B][4]  (row | kind<<31, occ_start, occ_count, 0); -1 = unused
 *   occ                      [n_batches][3B][2]  user occurrence: (i, j); item occurrence: (u, other|role<<31)
 * batch_size <= 8192.  Output is bit-exact against oracle/plan_np.py. */
int tkr_sample_plan(const int32_t* tr_users, int32_t n_tr, const int32_t* row_ptr, const int32_t* pos_cols,
                    const int32_t* cols_sorted, int32_t n_items, uint64_t seed, uint64_t first_triplet,
                    const int64_t* ctl, int32_t n_batches, int32_t batch_size, int32_t* out_u, int32_t* out_i,
                    int32_t* out_j, int32_t* task, int32_t* occ, void* stream);

/* ---- K2: BPR mini-batch step ---------------------------------------------------------------
 * Replaces sess.run([solver, obj]) (single/bpr.py:141) on the graph of single/bpr.py:81-100.
 * Double-buffered tables: U, msU are [2][n_users][k]; V, msV [2][n_items][k]; b, msb [2][n_items];
 * ustamp[n_users], istamp[n_items] = (serial<<1 | buffer holding the current row), 0 initially. */
typedef struct {
    float* U;
    float* msU;
    int32_t* ustamp;
    float* V;
    float* msV;
    float* b;
    float* msb;
    int32_t* istamp;
    int32_t n_users, n_items, k;
    int32_t mode;            /* 0 = 'l2' (single/bpr.py:92-95), 1 = L1 variant (:96-99) */
    float lu, li, lj, lb;    /* lambda_u, lambda_i, lambda_j, lambda_b (single/bpr.py:20) */
    float lr;                /* RMSPropOptimizer(lr) (single/bpr.py:100) */
    float rho, eps;          /* TF defaults 0.9, 1e-10 */
} tkr_bpr_state;

/* one batch; serial in [1, 2^30) strictly increasing over the life of the tables;
 * loss_out (nullable) device float, the batch objective is ADDED to it */
int tkr_bpr_step(const tkr_bpr_state* st, const int32_t* task, const int32_t* occ, int32_t batch_size,
                 int32_t serial, float* loss_out, void* stream);
/* n_batches consecutive batches of a plan (the inner loop of single/bpr.py:139-147);
 * loss_out (nullable) is float[n_batches], pre-zeroed by the caller */
int tkr_bpr_run(const tkr_bpr_state* st, const int32_t* task, const int32_t* occ, int32_t batch_size,
                int32_t n_batches, int32_t first_serial, float* loss_out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* TKR_H */
