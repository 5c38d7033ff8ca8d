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
 *     sizes; the library keeps no device memory of its own, with one stated exception: K4's per-shape
 *     work-item table (a few KB, cached per device for the 8 most recent shapes)
 *   - NOT thread-safe: the reference is a single Python thread and so is the caller this
 *     library is written for.  Process-global state without locks: K4's work-item table cache and its
 *     arithmetic mode (tkr_topk_set_math), per-device attributes cached by K1 and K2f on first use.  One host
 *     thread per process calls in; one process per GPU.
 *   - `stream` is a hipStream_t passed as void*; all calls are asynchronous and stream-ordered;
 *     scalar results are written to device memory
 *   - ids are int32, parameters fp32
 *   - device memory means ORDINARY (coarse-grained) device allocations -- hipMalloc, a torch CUDA tensor.  loss_out, the K3
 *     workspace and every table must not be fine-grained / host-coherent memory (hipHostMalloc, hipMallocManaged,
 *     hipExtMallocWithFlags(fine-grained)): the loss sums of K2f / K3 use the hardware's global_atomic_add_f32, which such memory
 *     silently drops on gfx90a and later, and the write-through granule protocol of K2f / K2o assumes device-local lines
 */
#ifndef TKR_H
#define TKR_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TKR_VERSION 118 /* 0.1.18: K4 second form of bound-and-refine (csrc/topk_refine.hip; tkr_topk_workspace_bytes_for grows by the pieces' packed lists), any k (bpr_wide_kernel, score_topk_wide_kernel), tkr_lab_build; tkr_topk_set_finish is gone, tkr_topk_set_math(0) and tkr_vbpr_set_pairs(1 | 2) need the lab library. 0.1.17: tkr_vbpr_set_pairs (tkr_vbpr_workspace_floats + 64). 0.1.16: tkr_topk_set_finish (larger tkr_topk_workspace_bytes), tkr_bpr_own_plan_run plans inside the step's launch. 0.1.15: tkr_bpr_own_owners_shared; tkr_bpr_run takes `rec` non-const. 0.1.14: per-task loss sums instead of atomics on loss_out (larger tkr_vbpr_workspace_floats; K2 writes word 15 of its records). 0.1.13: tkr_bpr_own_plan_run, K4 to k = 768. 0.1.12: tkr_bpr_own_run_between. 0.1.11: K2o (tkr_sample_plan_owned, tkr_bpr_own_run: item rows owned by one workgroup each, resident in its LDS); prec[5] = last batch of the call that updated the row. 0.1.10: tkr_topk_workspace_bytes_for (K4 stages pre-converted fp16 tiles). 0.1.9: tkr_vbpr_colplan + tkr_vbpr_run_cols (VBPR in three launches per batch). 0.1.8: tkr_sync_flow_* (exchange of the granule tables). 0.1.7: K4 bound-and-refine arithmetic (tkr_topk_set_math(2), the default; larger tkr_topk_workspace_bytes); K2f leaves its ticket words zero. 0.1.6: K2f persistent dataflow step, tkr_plan_rollback, batches above 8192 */
#define TKR_OK 0
#define TKR_E_INVAL (-1)
#define TKR_E_UNSUPPORTED (-2)
#define TKR_E_IO (-3)          /* host text I/O: file cannot be opened / written */
#define TKR_E_PARSE (-4)       /* host text I/O: malformed line (where the reference raises) */
#define TKR_E_NOMEM (-5)       /* host text I/O: allocation failed */

int tkr_version(void);
/* 1: built with `make LAB=1` -- the library also holds the kernel forms that were measured slower and are nobody's default (K2o scalar
 * exchange / scout / 16 waves / loader ring, K4 bf16x3, the VBPR pair-sum placements 1 and 2); 0: asking for one returns TKR_E_UNSUPPORTED */
int tkr_lab_build(void);

/* ---- K1: (u,i,j) draw + batch plan ---------------------------------------------------------
 * Replaces BPR._uniform_user_sampling (single/bpr.py:155-165) and the gradient de-duplication
 * bookkeeping TF does inside sess.run (single/bpr.py:100,141).
 *   tr_users[n_tr]           users with >= 1 positive (single/bpr.py:65)
 *   row_ptr[n_users+1], pos_cols[nnz]   tr_data as CSR, file order, duplicates kept (bpr.py:167-171)
 *   cols_sorted[nnz]         per-row ascending copy (membership test of bpr.py:163)
 *   seed, first_triplet      counter-based stream position (triplet g = first_triplet + b*B + t);
 *   ctl                      optional device int64[1]: batch index added to the stream position at
 *                            run time (lets one captured hipGraph walk through an epoch)
 *   ucnt[n_users], icnt[n_items]   running number of updates of every row (in/out); parity = which
 *                            of the two table buffers currently holds the row
 *   touch_u[n_users*16], touch_i[n_items*16]   scratch bitmaps, all-zero on entry and on exit
 *   out_u/out_i/out_j        [n_batches*B]
 *   task                     [n_batches][3B][4]  (row | kind<<31, occ_start, occ_count, parity); -1 = unused
 *   occ                      [n_batches][3B][2]  user occurrence: (i, j); item occurrence: (u, other|role<<31);
 *                            bit 30 of every id = parity of that row
 *   rec                      [n_batches][tkr_plan_max_blocks(B)*tkr_plan_team(B)][16]  per-wave launch records
 *   hdr                      [n_batches][4]  (workgroups used, light workgroups, heavy tasks, tasks)
 *   occt                     [n_batches][3B] triplet index t of every sorted occurrence (used by K3)
 *   tpar                     (nullable) [n_batches*B] per triplet: parity of u | parity of i << 1 | parity of j << 2 at its
 *                            batch -- lets K3's sparse view score a triplet where it projects it (no per-occurrence launch)
 *   prec, pocc               (both NULL, or both set = DATAFLOW form for K2f; rec / hdr / tpar are then not written and may
 *                            be NULL)  pocc [n_batches][3B][4]: per sorted occurrence (a, version of a, b | role<<31,
 *                            version of b); prec [n_batches][3B][32]: one 128-byte record per task slot:
 *                            [0] row | kind<<31 (-1 = unused slot) [1] version of the row [2] occurrences [3] index of its
 *                            first occurrence in pocc counted from batch 0 [4] batch [5] last batch < [4] of this call
 *                            that updated the row (-1: none) [8+4q..] pocc of occurrence q < 4 [24+q] the index of that
 *                            occurrence's triplet in its batch (= occt)
 *   workspace                (required for batch_size > 8192, nullable below) tkr_plan_workspace_bytes(batch_size, n_batches) bytes
 *                            of device scratch.  With it, batches of 1025 .. 65,536 are planned by a counting sort over row ranges
 *                            (csrc/planner_mid.hip: six launches; range sums, above 16,384 also 12 bytes per triplet of run lists) and larger ones grid-wide (device radix
 *                            sort of batch|row|occurrence keys, scans) instead of one workgroup per batch -- single/bpr.py:103-113
 *                            accepts any batch_size.  The touch maps are scratch INSIDE a call (zero before and after)
 * n_batches <= 512, ids < 2^30, batch_size <= 2^20 (the dataflow form: batch_size <= 8192).  Output is bit-exact against
 * oracle/plan_np.py for every batch size. */
int tkr_plan_team(int32_t batch_size);        /* waves per step workgroup / heavy-row team: 4 (B <= 1024), 8 (<= 16384) or 16 */
int tkr_plan_max_blocks(int32_t batch_size);  /* workgroups a batch can need */
int tkr_sample_plan(const int32_t* tr_users, int32_t n_tr, const int32_t* row_ptr, const int32_t* pos_cols,
                    const int32_t* cols_sorted, int32_t n_users, int32_t n_items, uint64_t seed,
                    uint64_t first_triplet, const int64_t* ctl, int32_t n_batches, int32_t batch_size,
                    int32_t* ucnt, int32_t* icnt, uint32_t* touch_u, uint32_t* touch_i, int32_t* out_u,
                    int32_t* out_i, int32_t* out_j, int32_t* task, int32_t* occ, int32_t* rec, int32_t* hdr,
                    int32_t* occt, int32_t* tpar, int32_t* prec, int32_t* pocc, void* workspace, int64_t workspace_bytes,
                    void* stream);
/* device scratch tkr_sample_plan needs for batch_size > 8192 (sort keys, ranks, hipcub storage); 0 for smaller batches */
int64_t tkr_plan_workspace_bytes(int32_t batch_size, int32_t n_batches);

/* Take batches [first_batch, first_batch + n_batches) of a plan out of the update counters again.  tkr_sample_plan
 * advances ucnt / icnt for every batch it PLANS; a caller that drops the rest of a plan (BPR.train stopping inside a
 * chunk, a parameter read-back between chunks) calls this so that the counters equal the updates that really RAN.
 * `task` is the array tkr_sample_plan wrote. */
int tkr_plan_rollback(const int32_t* task, int32_t batch_size, int32_t first_batch, int32_t n_batches, int32_t* ucnt,
                      int32_t* icnt, void* stream);

/* ---- K2: BPR mini-batch step ---------------------------------------------------------------
 * Replaces sess.run([solver, obj]) (single/bpr.py:141) on the graph of single/bpr.py:81-100.
 * Double-buffered tables: U, msU are [2][n_users][k]; V, msV [2][n_items][k]; b, msb [2][n_items];
 * the buffer holding row r is (ucnt[r] & 1) resp. (icnt[r] & 1) -- see K1.
 * opt = 1 selects the legacy plain-SGD update P -= lr * g of old/methods/bpr.py:57-61: the
 * objective and gradients are the same as mode 0 (old/methods/bpr.py:43-51), the ms* tables are neither read
 * nor written and may be NULL. */
typedef struct {
    float* U;
    float* msU;
    float* V;
    float* msV;
    float* b;
    float* msb;
    int32_t n_users, n_items, k;
    int32_t mode;            /* 0 = 'l2' (single/bpr.py:92-95), 1 = L1 variant (:96-99) */
    float lu, li, lj, lb;    /* lambda_u, lambda_i, lambda_j, lambda_b (single/bpr.py:20) */
    float lr;                /* RMSPropOptimizer(lr) (single/bpr.py:100) */
    float rho, eps;          /* TF defaults 0.9, 1e-10 */
    int32_t opt;             /* 0 = sparse RMSProp (single/bpr.py:100), 1 = plain SGD (old/methods/bpr.py:57-61) */
} tkr_bpr_state;

/* n_batches consecutive batches of a plan (the inner loop of single/bpr.py:139-147), one launch
 * each, in plan order; loss_out (nullable) is float[n_batches], pre-zeroed by the caller: the
 * batch objective (single/bpr.py:93-99) is added to loss_out[b].  How (all entries with a loss_out): thousands of waves summing
 * into one word with atomics cost 10 ns apiece, one after the other (4-5x the step at batch 8192).  Every user task therefore
 * leaves its sum in a place of its own -- here word 15 of its launch record in `rec` (K1 leaves it zero; that one word of `rec` is
 * WRITTEN by the step, which is why `rec` is not const; a plan run again with a loss_out overwrites it), under K2o a {sum, epoch} slot of `xch`, under K2f its workgroup's LDS, under K3 one of
 * 64 slots behind the workspace -- and one small launch per CALL adds them up into loss_out (tkr_bpr_own_run in its default
 * form ASSIGNS loss_out[b]: no fill is needed in front of it). */
/* any k: up to 512 (256 above batch_size 1024) a wave holds its row in registers; wider rows take the generic form
 * (csrc/bpr_step.hip bpr_wide_kernel: two passes over k per occurrence, the gradient sum through memory) */
int tkr_bpr_run(const tkr_bpr_state* st, int32_t* rec, const int32_t* occ, const int32_t* hdr,
                int32_t batch_size, int32_t n_batches, float* loss_out, void* stream);

/* ---- K2f: the same step as ONE persistent launch per chunk (dataflow form; csrc/bpr_flow.hip) ------------------
 * Replaces the same call site as K2 (single/bpr.py:139-147, the loop around sess.run) for small batches, where one launch
 * per batch is 96 % idle.  Batches are ordered by the data instead of by kernel boundaries: every table element is an
 * 8-byte granule {fp32 value, uint32 version tag}, double-buffered (version v of a row lives in buffer v & 1):
 *   U, msU   [2][n_users][kp] granules, kp = tkr_flow_row_granules(k) (k rounded up to 128; padding: value 0 / slot 1)
 *   V, msV   [2][n_items][kp]
 *   tailU/V  [2][n][4] granules per row: {item bias, its RMSProp slot, expect[0], expect[1] (uint32 bits)}; users carry
 *            zeros for the first two
 *   rdU/V    [n][2] uint32: acknowledged partner reads of the row, by parity of the version read
 * A freshly assigned table holds version 0 in buffer 0 (all tags 0, expects 0, rd 0) and tags 0xffffffff in buffer 1.
 * item_bufs = 4 (round 4): the ITEM tables have FOUR buffers -- V, msV [4][n_items][kp], tailV [4][n_items][8] granules per row
 * {bias, its slot, expect[0..3], 0, 0}, rdV [n_items][4], version v in buffer v & 3 -- so the task that writes version v+1 waits
 * for the readers of v-3 instead of v-1: with the rows of a chain handed over in LDS (K2o) the acknowledge round trip of TWO
 * buffers (publish -> partner reads -> acknowledge -> next publish, once per two batches) is what paces batch 256.
 * The plan is tkr_sample_plan's dataflow form (prec / pocc non-NULL there): `prec` points at the record of the first
 * task of the first batch to run, `pocc` and `loss_out` (nullable, pre-zeroed) at batch 0 of that plan call.
 * ctl: tkr_flow_ctl_words() uint32 of caller-owned device memory, zeroed once by the caller; every launch leaves its
 * ticket words at zero again (the last workgroup out does it), so calls on one stream follow each other without a
 * memset between them.  One ctl serves one launch at a time.  ctl[status word] != 0 after a launch means a bounded spin
 * ran out (results invalid; zero ctl again before anything else runs on it).
 * waves_per_cu: 0 = default (one 4-wave workgroup per CU below batch 320, two from there on: csrc/bpr_flow.hip).
 * k <= 256, 3 * batch_size * n_batches < 2^29. */
typedef struct {
    void* U;
    void* msU;
    void* tailU;
    uint32_t* rdU;
    void* V;
    void* msV;
    void* tailV;
    uint32_t* rdV;
    int32_t n_users, n_items, k;
    int32_t mode;            /* as tkr_bpr_state */
    float lu, li, lj, lb;
    float lr, rho, eps;
    int32_t opt;             /* 0 = sparse RMSProp, 1 = plain SGD (ms tables unused, may be NULL) */
    int32_t item_bufs;       /* buffers per ITEM row: 0 or 2 = two (as the user rows), 4 = four (see below) */
} tkr_flow_state;
int32_t tkr_flow_row_granules(int32_t k);
int32_t tkr_flow_ctl_words(void);
#define TKR_FLOW_CTL_STATUS 1026 /* index of the status word inside ctl */
#define TKR_FLOW_CTL_SPINS 1027  /* diagnostics: spin passes taken since the caller last zeroed it */
int tkr_bpr_flow_run(const tkr_flow_state* st, const int32_t* prec, const int32_t* pocc, int32_t batch_size,
                     int32_t n_batches, uint32_t* ctl, float* loss_out, int32_t waves_per_cu, void* stream);

/* ---- K2o: the persistent step with OWNED item rows (csrc/bpr_own.hip) ---------------------------------------------
 * Replaces the same call site as K2 / K2f (single/bpr.py:139-147) on the same granule tables (tkr_flow_state), and leaves them
 * in the same state as K2f does: the two kernels, the exchange and get / set may be mixed freely between launches.  Item row r is
 * served by workgroup r % n_owner, which keeps the row, its slot, bias and acknowledge totals in LDS from the row's first update
 * in a launch on: the task of batch t+1 finds the row of batch t there instead of polling memory for it (what bounded K2f at
 * batch 256).  Partner rows are read from the tables as in K2f; `owner_waves` bit 15 selects the form in which the two item tasks
 * of a triplet exchange the scalars <u, v> + b through an 8-byte slot of `xch` instead of reading each other's rows (measured
 * slower: csrc/bpr_own.hip has both protocols and the numbers).  User tasks are handed out by tickets as in K2f.
 *   tkr_bpr_own_owners(n_items, k)   n_owner for the current device (its CU count), or 0 when ceil(n_items / CUs) rows of
 *                                    8*kp + 16 bytes do not fit one CU's 160 KB of LDS (use K2f then)
 *   tkr_sample_plan_owned            tkr_sample_plan's dataflow form (prec, pocc; batch_size <= 8192) with the records of every
 *                                    batch's ITEM tasks in (row % n_owner, row) order -- same slots -- and
 *                                    ohdr[owner * ohdr_stride + batch] = first slot | tasks << 16 (ohdr_stride >= n_batches).
 *                                    (K2f runs such a plan too: the order of tasks inside a batch means nothing to it.)
 *   tkr_bpr_own_run                  batches [first_batch, first_batch + n_batches) of such a plan in ONE launch of n_owner
 *                                    workgroups (all resident: n_owner <= CUs).  prec / pocc / occt / ohdr / loss_out point at
 *                                    batch 0 of the plan call; owner_waves: waves per workgroup that serve the owner queue, 0 =
 *                                    default (workgroups of 12 waves at k <= 128: 10 on the owner queue, 2 on user tickets; 8
 *                                    waves above: 7 + 1; bits 13 / 14: 8 / 16 waves at k <= 128); ctl as for tkr_bpr_flow_run.
 *                                    xch: 16 * batch_size bytes per batch of the plan (the scalar slots {value, epoch} of every
 *                                    (triplet, role); only the scalar form writes them), caller-owned, zeroed once; epoch: a
 *                                    number > 0 that no earlier launch on this xch used (a counter per plan buffer does).
 * k <= 256, n_batches <= 512. */
int32_t tkr_bpr_own_owners(int32_t n_items, int32_t k);
/* the same when `share` processes split one device's CUs between them (ranks packed on one GPU: each runs CUs / share owners, so
 * that the workgroups of all of them are resident together); 0 when ceil(n_items / (CUs / share)) rows do not fit an owner's LDS */
int32_t tkr_bpr_own_owners_shared(int32_t n_items, int32_t k, int32_t share);
int tkr_sample_plan_owned(const int32_t* tr_users, int32_t n_tr, const int32_t* row_ptr, const int32_t* pos_cols,
                          const int32_t* cols_sorted, int32_t n_users, int32_t n_items, uint64_t seed, uint64_t first_triplet,
                          int32_t n_batches, int32_t batch_size, int32_t* ucnt, int32_t* icnt, uint32_t* touch_u,
                          uint32_t* touch_i, int32_t* out_u, int32_t* out_i, int32_t* out_j, int32_t* task, int32_t* occ,
                          int32_t* occt, int32_t* prec, int32_t* pocc, int32_t n_owner, int32_t* ohdr, int32_t ohdr_stride,
                          void* stream);
int tkr_bpr_own_run(const tkr_flow_state* st, const int32_t* prec, const int32_t* pocc, const int32_t* occt, const int32_t* ohdr,
                    int32_t ohdr_stride, int32_t n_owner, int32_t batch_size, int32_t first_batch, int32_t n_batches, uint32_t* ctl, float* loss_out,
                    int32_t owner_waves, void* xch, uint32_t epoch, void* stream);
/* One call for a SHORT training call -- K1 of a chunk and the persistent step on it, back to back: tkr_sample_plan_owned with the
 * arguments of `plan` (a struct: a caller fills it once per plan buffer and changes first_triplet / n_batches per call), then
 * tkr_bpr_own_run on batches [first_batch, first_batch + n_batches) of that plan, with the caller's two events (hipEvent_t or NULL)
 * recorded around the step launch.  No trip back through the host language between them: a 20-batch call is ~90 us of device
 * work, and an interpreter between the launches leaves the device waiting for the host.
 * From 0.1.16 the whole plan is what most short calls run, and then K1 runs INSIDE the step's launch (ONE launch: workgroups
 * 0 .. n_batches-1 plan a batch each in front of the step, csrc/bpr_own.hip PLAN; the last workgroup out adds the losses up): when
 * first_batch == 0, n_batches == plan->n_batches <= min(64, n_owner), batch_size <= 256, k <= 128, the default step form, n_users and
 * n_items below 2^25; otherwise tkr_sample_plan_owned's launches + the step's, as before.  Same words in every plan array and the same
 * tables either way.  owner_waves bit 12: never the one-launch form; bit 9: the losses always by their own launch; bit 8: the loader /
 * consumer form of the step (long launches only; measured slower). */
typedef struct tkr_plan_call {
    const int32_t *tr_users, *row_ptr, *pos_cols, *cols_sorted;
    int32_t *ucnt, *icnt;
    uint32_t *touch_u, *touch_i;
    int32_t *out_u, *out_i, *out_j, *task, *occ, *occt, *prec, *pocc, *ohdr;
    uint64_t seed, first_triplet;
    int32_t n_tr, n_users, n_items, n_batches, batch_size, n_owner, ohdr_stride, reserved;
} tkr_plan_call;
int tkr_bpr_own_plan_run(const tkr_plan_call* plan, const tkr_flow_state* st, int32_t first_batch, int32_t n_batches, uint32_t* ctl,
                         float* loss_out, int32_t owner_waves, void* xch, uint32_t epoch, void* ev_before, void* ev_after, void* stream);
/* tkr_bpr_own_run between two events of the caller (hipEvent_t or NULL), recorded on `stream` right around the launch */
int tkr_bpr_own_run_between(void* ev_before, void* ev_after, const tkr_flow_state* st, const int32_t* prec, const int32_t* pocc,
                            const int32_t* occt, const int32_t* ohdr, int32_t ohdr_stride, int32_t n_owner, int32_t batch_size,
                            int32_t first_batch, int32_t n_batches, uint32_t* ctl, float* loss_out, int32_t owner_waves, void* xch,
                            uint32_t epoch, void* stream);

/* ---- K3: VBPR mini-batch step -------------------------------------------------------------
 * Replaces sess.run([solver, obj]) of single/vbpr.py:114 on the graph of single/vbpr.py:50-73 and the
 * per-batch host gather of two dense [B, d] feature slices (vbpr.py:114).  kh = k // 2.
 *   U, msU      [2][n_users][2*kh]  rows = [ure | uce]  (vbpr.py:37-40), double-buffered like K2
 *   I, msI      [2][n_items][kh]    ire (vbpr.py:41);   irb, msirb [2][n_items] (vbpr.py:43)
 *   cem, mscem  [d][kh], icb, msicb [d]: dense variables, updated in place every batch (vbpr.py:45-48,73)
 *   feat        [n_items][d] dense fp32, resident (REC.load_content_data, rec.py:23-33)
 * Optional sparse view of feat (f_ptr != NULL; content features are tf-idf-like, ~0.5 % dense at d = 20,000): the
 * projection gathers cem rows per nonzero instead of contracting over d, and the dense-variable update walks the
 * columns of feat (CSC) against the per-item sums of the batch.  Same arithmetic up to the order of the fp32 sums;
 * every element of cem / icb still gets TF's dense ApplyRMSProp every batch (vbpr.py:65,67,73).
 *   f_ptr [n_items+1], f_col / f_val [nnz]   CSR over items, ascending columns inside a row
 *   c_ptr [d+1], c_item / c_val [nnz]        CSC over feature columns, ascending items inside a column
 *   item_tag [n_items] int64                 scratch, zeroed ONCE by the caller and then owned by the library (per-batch
 *                                            membership bitmaps, slots and a batch counter live in it) */
typedef struct {
    float* U;
    float* msU;
    float* I;
    float* msI;
    float* irb;
    float* msirb;
    float* cem;
    float* mscem;
    float* icb;
    float* msicb;
    const float* feat;
    int32_t n_users, n_items, kh, d;
    int32_t mode;                /* 0 = 'l2' (vbpr.py:63-67), 1 = L1 variant (:68-72) */
    float lu, li, lj, lb, le;    /* lambda_u, lambda_i, lambda_j, lambda_b, lambda_e (vbpr.py:18) */
    float lr, rho, eps;
    const int32_t* f_ptr;
    const int32_t* f_col;
    const float* f_val;
    const int32_t* c_ptr;
    const int32_t* c_item;
    const float* c_val;
    int64_t* item_tag;
} tkr_vbpr_state;

/* floats of scratch tkr_vbpr_run needs (split-K partials, s_t, P_t, W_t) */
int64_t tkr_vbpr_workspace_floats(int32_t batch_size, int32_t kh, int32_t d);
/* tkr_vbpr_run_cols: where the [B, B] pair sums S_t, T_t of a batch (vbpr.py:61) are formed -- 0: a launch of their own between the
 * projection and the update (three launches per batch); 1: every task of the update works out the sums it needs (batch <= 256);
 * 2: the first blocks of the update launch form them for everybody (two launches per batch).  Same sums, same order of summation,
 * bit for bit (1: within fp32 rounding of the others).  Initial value from TKR_VBPR_PAIRS. */
int tkr_vbpr_set_pairs(int32_t mode);
/* n_batches consecutive batches planned by tkr_sample_plan (tri_i / tri_j = its out_i / out_j);
 * kh <= 128 (any kh: tkr_vbpr_run_cols), batch_size <= 65536 (batches above 8192 are planned grid-wide, see tkr_sample_plan); loss_out as in tkr_bpr_run */
int tkr_vbpr_run(const tkr_vbpr_state* st, const int32_t* tri_i, const int32_t* tri_j, const int32_t* rec,
                 const int32_t* occ, const int32_t* hdr, const int32_t* occt, const int32_t* tri_u /*nullable*/,
                 const int32_t* tpar /*nullable*/, int32_t batch_size,
                 int32_t n_batches, float* workspace, float* loss_out, void* stream);

/* Column-plan form of the same step (batch_size <= 1024, CSR view of feat; kh % 4 == 0 and kh <= 128: rows and cem rows in registers,
 * any other kh: the generic kernels of csrc/vbpr_wide.hip on the same plan): three launches per batch
 * (project + score / pair sums / row tasks and column tasks side by side) instead of four to six, csrc/vbpr_cols.hip.
 * Which (triplet, feature column) pairs a batch touches depends on the sampled triplets and on the structure of feat only,
 * so it is prepared beside K1, off the step's critical path:
 *   tkr_vbpr_colplan   for each of n_batches batches (tri_i / tri_j = tkr_sample_plan's out_i / out_j), from the CSR
 *                      f_ptr / f_col / f_val of feat, with tcap = 2 * row_cap and row_cap >= the longest row of feat:
 *                      tcnt [n_batches][B], tent [n_batches][B][tcap][2]   per triplet the (column, value bits) of f_i's nonzeros
 *                                          (+value) followed by f_j's (-value): the gather list of the projection
 *                      cent [n_batches][B * tcap][2]   the same entries as (t, +-value bits) grouped by feature column, every
 *                                          column's run in (t, side) order
 *                      colh [n_batches][d][8]   per column (entries, first entry in cent, then its first three entries inline)
 *                      One workgroup per batch keeps the d column counters in LDS: tkr_vbpr_colplan_lds_bytes(batch_size, d)
 *                      <= 160 KB, else TKR_E_UNSUPPORTED (the caller then stays with tkr_vbpr_run).
 *   tkr_vbpr_run_cols  the batches themselves; arguments as tkr_vbpr_run (tri_u and tpar required; st->feat and the CSC
 *                      members of st are not read) plus the column plan.  cols_per_block: feature columns per 256-thread
 *                      workgroup of the dense update, 0 = as many as fit (sparse features); 1-2 when runs are long (a narrow
 *                      dense feat: every column meets every triplet).
 * Same arithmetic as tkr_vbpr_run up to the order of fp32 sums; bitwise reproducible run to run. */
int64_t tkr_vbpr_colplan_lds_bytes(int32_t batch_size, int32_t d);
int tkr_vbpr_colplan(const int32_t* f_ptr, const int32_t* f_col, const float* f_val, int32_t d, const int32_t* tri_i,
                     const int32_t* tri_j, int32_t batch_size, int32_t n_batches, int32_t row_cap, int32_t* colh, int32_t* cent,
                     int32_t* tcnt, int32_t* tent, void* stream);
int tkr_vbpr_run_cols(const tkr_vbpr_state* st, const int32_t* tri_i, const int32_t* tri_j, const int32_t* rec, const int32_t* occ,
                      const int32_t* hdr, const int32_t* occt, const int32_t* tri_u, const int32_t* tpar, const int32_t* colh,
                      const int32_t* cent, const int32_t* tcnt, const int32_t* tent, int32_t row_cap, int32_t cols_per_block,
                      int32_t batch_size, int32_t n_batches, float* workspace, float* loss_out, void* stream);

/* ---- K4: full-catalogue score -> rated mask -> top-K -------------------------------------------
 * Replaces evaluate.py:78-81 (np.dot + bias + np.argsort over every row) and the rated-item filter
 * of the rank walk (evaluate.py:96-105).
 *   U [*, k] user factors; user_idx[n_rows] (nullable) picks the rows to rank (default 0..n_rows-1)
 *   Vt [n_cols, k], bias[n_cols] (nullable): the scenario's test-item rows, already gathered
 *     (evaluate.py:75-77; bias gathered per test id -- the reference's broadcast at :79-80 is only
 *     defined when the scenario's id list equals vid)
 *   mask [ceil(n_cols/32)][mask_pitch] (nullable): bit (c & 31) of word (c >> 5, row) = column c is
 *     train-rated by row's user (evaluate.py:98); built by tkr_build_rated_mask into a zeroed buffer
 *   out_ids [n_rows, K]: the K best unrated columns, descending score, ties -> higher column first;
 *     -1 where fewer than K unrated columns exist.  out_scores (nullable) alike, -inf padded.
 *   workspace (nullable, device, workspace_bytes): scratch for per-item-range partial lists; with
 *     tkr_topk_workspace_bytes(n_rows, K) bytes the launch splits the catalogue so that the grid fills
 *     the 256 CUs in whole rounds (results are identical with or without it)
 * Arithmetic of the scores (tkr_topk_set_math, initial value from TKR_TOPK_MATH=refine|bf16x3|fp32):
 *   2 "refine" (default; k <= 128, needs the workspace, n_cols < 2^27): bound-and-refine.  ONE fp16 product per element
 *     (factors scaled by powers of two, v_mfma_f32_32x32x16_f16) scores the catalogue within a rigorous margin
 *     (2^-10 * 1.05 * |u| * max|v_i| + roundings, see csrc/topk.hip); the candidate lists keep everything within twice the
 *     margin of the K-th best approximate score, and the survivors are rescored with the arithmetic of mode 1, which
 *     decides the order: ids and score bits are those of mode 1, at 2.1x its speed.  A list that cannot hold its margin
 *     (massive near-ties) sends its user block through the mode-1 kernel.  Without a workspace the call runs as mode 1.
 *   0 "bf16x3" (k <= 128; lab library only, `make LAB=1`: TKR_E_UNSUPPORTED otherwise): each fp32 factor is split exactly into three bf16 parts and a product is the six
 *     leading partial products (each exact in fp32, the dropped ones < 2^-23 |ab|) accumulated in fp32 by
 *     v_mfma_f32_32x32x16_bf16 -- an fp32 dot product with yet another summation order: same error against fp64 as
 *     np.dot / the fp32 kernel, identical results whenever the partial sums are representable; finite inputs only;
 *   1 "fp32": v_mfma_f32_32x32x2_f32, used for every k and always for k > 128: per score the fma chain
 *     acc <- fma(v[j], u[j], acc); acc <- fma(v[KH+j], u[KH+j], acc), j = 0 .. KH-1, KH = ceil(k/2), then fl(acc + bias)
 *     (oracle/ref_np.py mfma_chain_scores restates it; bit-exact on generic inputs).
 *   On gfx950 the fp32 MFMA runs at the vector-ALU rate and overlaps nothing; mode 0 is 1.4-1.6x faster than mode 1,
 *   mode 2 1.5-2.1x.
 * K <= 32 per launch (larger K: rank 32, add the found columns to the mask with tkr_build_rated_mask, rank
 * again -- top-k-rec_amd/tkr_hip.py score_topk does this); any k (above 768: score_topk_wide_kernel, one wave per row, a lane per item,
 * the same fma chain; 256 < k <= 768: mode 1 with the k dimension in slabs of 256 --
 * the chain of a score runs slab after slab, halves [256 s, 256 s + 128) and [256 s + 128, 256 s + 256) in place of the two
 * halves of k; 63 % of the fp32-MFMA peak at k = 512, register spills above 512). */
int64_t tkr_topk_workspace_bytes(int32_t n_rows, int32_t K);
/* the same + room for the item factors of mode 2 pre-converted to scaled fp16 (k <= 128), laid out as the LDS image of every
 * 32-item tile: with a workspace of at least this size a K4 call converts V once (a ~5 us pre-pass) and every workgroup stages
 * its tiles with one direct-to-LDS load per thread instead of converting them again (results are identical either way) */
int64_t tkr_topk_workspace_bytes_for(int32_t n_rows, int32_t n_cols, int32_t k, int32_t K);
int tkr_topk_set_math(int32_t mode);
int tkr_build_rated_mask(const int64_t* rated_ptr, const int32_t* rated_cols, int32_t n_rows, int32_t n_cols,
                         uint32_t* mask, int32_t mask_pitch, void* stream);
int tkr_score_topk(const float* U, const int32_t* user_idx, int32_t n_rows, const float* Vt, const float* bias,
                   int32_t n_cols, int32_t k, const uint32_t* mask, int32_t mask_pitch, int32_t K,
                   int32_t* out_ids, float* out_scores, void* workspace, int64_t workspace_bytes, void* stream);

/* ---- K5: hit counting of the rank walk (evaluate.py:99-103) ------------------------------------
 * first_bucket[p / step] += 1 (uint64 atomics) for every kept position p < interval*step of row r
 * whose column is in like_cols[like_ptr[r] .. like_ptr[r+1]) (ascending); hits[q] of the reference
 * is the running sum over buckets <= q. */
int tkr_count_hits(const int32_t* ids, int32_t n_rows, int32_t K, const int64_t* like_ptr, const int32_t* like_cols,
                   int32_t step, int32_t interval, uint64_t* first_bucket, void* stream);

/* ---- K6 / K7: the rank walk of utils.evaluate (utils.py:101-127) -----------------
 * utils.evaluate buckets a hit by the item's RAW rank t -- train-rated columns included (utils.py:113-117:
 * j = t // step) -- and sums reciprocal ranks 1/(t+1).  With ids = K4's first K unrated columns of each row,
 *   raw_rank[r][p] = p + #{rated columns of row r ranked before ids[r][p]}   (-1 where ids is -1)
 * under the canonical order (descending score, ties -> higher column first); kept and rated columns are scored
 * by the same fp32 dot routine inside the kernel.  rated_ptr/rated_cols: the CSR given to tkr_build_rated_mask.
 * tkr_count_hits_rr: hit_first[r][j] += 1 and rr_first[r][j] += 1/(t+1) (fp64) for every kept liked column with
 * j = t / step < interval, sequentially per row (reproducible); the caller zeroes both arrays, sums over rows and
 * accumulates buckets <= k (utils.py:115-117).  K <= 256, k <= 256. */
int tkr_raw_ranks(const float* U, const int32_t* user_idx, int32_t n_rows, const float* Vt, const float* bias, int32_t k,
                  const int64_t* rated_ptr, const int32_t* rated_cols, const int32_t* ids, int32_t K, int32_t* raw_rank,
                  void* stream);
int tkr_count_hits_rr(const int32_t* ids, const int32_t* raw_rank, int32_t n_rows, int32_t K, const int64_t* like_ptr,
                      const int32_t* like_cols, int32_t step, int32_t interval, int32_t* hit_first, double* rr_first,
                      void* stream);

/* ---- multi-GPU: pack / unpack of the replicated item-side tables around the per-epoch all-reduce ---------
 * (new design, the reference is single-process: SURVEY.md §8e).  Users are sharded over the GPUs, every rank updates its
 * own copy of the item tables and once per epoch  P <- P0 + sum_g (P_g - P0),  ms <- mean_g ms_g.  For a table
 * P, ms [2][n][w] whose current row r lives in buffer (cnt[r] & 1) (cnt NULL: a single-buffered [n][w] table):
 *   tkr_sync_snapshot  start[r] = current P[r]                                   (at the start of the epoch)
 *   tkr_sync_pack      flat_delta[r] = current P[r] - start[r];  flat_ms[r] = current ms[r] * inv_world
 *   (the caller all-reduces flat_delta | flat_ms with SUM: RCCL over xGMI through torch.distributed)
 *   tkr_sync_unpack    P[0][r] = start[r] + flat_delta[r];  ms[0][r] = flat_ms[r];  the caller then zeroes cnt */
int tkr_sync_snapshot(const float* P, const int32_t* cnt, float* start, int64_t n, int32_t w, void* stream);
/* the same exchange for the granule tables of tkr_flow_state (item side: V, msV, tailV, rdV; n = n_items): start / flat_delta /
 * flat_ms hold the n*k elements of V followed by the n item biases.  tkr_sync_flow_unpack leaves the tables as a fresh assignment
 * does (version 0 in buffer 0, none in the others, expect = rd = 0) and zeroes the update counters itself.  item_bufs = 2 or 4:
 * the buffers per item row of the tables (tkr_flow_state.item_bufs). */
int tkr_sync_flow_snapshot(const void* V, const void* tailV, const int32_t* icnt, float* start, int32_t n, int32_t k, int32_t item_bufs,
                           void* stream);
int tkr_sync_flow_pack(const void* V, const void* msV, const void* tailV, const int32_t* icnt, const float* start, float* flat_delta,
                       float* flat_ms, int32_t n, int32_t k, float inv_world, int32_t item_bufs, void* stream);
int tkr_sync_flow_unpack(void* V, void* msV, void* tailV, uint32_t* rdV, int32_t* icnt, float* start /* in: epoch start; out: the new
                         values = the next epoch's start (no snapshot needed if nothing else writes the tables in between) */,
                         const float* flat_delta, const float* flat_ms, int32_t n, int32_t k, int32_t item_bufs, void* stream);
int tkr_sync_pack(const float* P, const float* ms, const int32_t* cnt, const float* start, float* flat_delta, float* flat_ms,
                  int64_t n, int32_t w, float inv_world, void* stream);
int tkr_sync_unpack(float* P, float* ms, const float* start, const float* flat_delta, const float* flat_ms, int64_t n, int32_t w,
                    void* stream);

/* ---- profiling aid: dst[r] = src[r] + 1 for the n listed rows of a [*, k] table, with the step
 * kernels' access pattern; used by scripts/pmc_calibrate.py to calibrate rocprofv3 byte counters */
int tkr_calib_rowcopy(const float* src, float* dst, const int32_t* rows, int32_t n, int32_t k, void* stream);

/* ---- host-side text I/O of the reference's data formats (no GPU work) ------------
 * The reference parses its inputs with per-element Python loops (utils.py:58-70 get_data_from_file,
 * evaluate.py:30-45 get_history and :84-93 the test file, utils.py:28-44 / evaluate.py:19-28 the '%f ' matrices,
 * utils.py:47-55 the writer); these do the same in one pass and return flat arrays.
 *
 * id map: token -> index, built from the n '\n'-separated tokens of `blob` (no trailing newline) and their
 *   indices (the host passes the reference's own dict, duplicate-line quirk of utils.py:10-16 included).
 * ratings ("uid,iid:like,iid:like,..." per line): one record per line -- line_user[l] = index of the uid or -1,
 *   entries [line_ptr[l], line_ptr[l+1]) in field order with item[e] = index of the iid or -1 and like[e] = the
 *   integer after the first ':' (a field without ':' or with a non-integer like -> TKR_E_PARSE, where the
 *   reference raises IndexError / ValueError).  Which entries count (known user, known item, like == 1, last line
 *   of a user wins, ...) is the caller's rule, as it differs between utils.py:58-70, evaluate.py:30-45 and :84-93.
 * matrix: every line parsed as ' '-separated decimal numbers (strtod, narrowed to fp32: the two roundings of
 *   np.float32(str)); ragged rows -> TKR_E_PARSE.  tkr_matrix_write emits "%f " per element and '\n' per row,
 *   byte-identical to utils.py:47-55.
 * Handles are created by *_create / *_parse / *_read and released by *_destroy. */
int tkr_idmap_create(const char* blob, int64_t blob_len, const int32_t* index, int64_t n, void** out_map);
int tkr_idmap_destroy(void* map);
int tkr_ratings_parse(const char* path, const void* user_map, const void* item_map, void** out_ratings);
int tkr_ratings_sizes(const void* ratings, int64_t* n_lines, int64_t* n_entries);
int tkr_ratings_copy(const void* ratings, int32_t* line_user, int64_t* line_ptr /*[n_lines+1]*/, int32_t* item,
                     int32_t* like);
int tkr_ratings_destroy(void* ratings);
int tkr_matrix_read(const char* path, void** out_matrix);
int tkr_matrix_sizes(const void* matrix, int64_t* rows, int64_t* cols);
int tkr_matrix_copy(const void* matrix, float* dst /*[rows*cols]*/);
int tkr_matrix_destroy(void* matrix);
int tkr_matrix_write(const char* path, const float* data, int64_t rows, int64_t cols);

#ifdef __cplusplus
}
#endif
#endif /* TKR_H */
