/*
 * dgll_hip.h -- C ABI of libdgll_hip.so, the MI355X (gfx950) sparse GNN aggregation engine that sits
 * behind the dgll.nn conv-layer API.
 *
 * This is the drop-in boundary of the hot path.  The reference's own native
 * boundary has the same shape -- two extern "C" launchers taking borrowed device pointers and plain
 * ints (/root/reference/dgll/FusedKernel/gcn_fused_kernel.cu:190-195 and :238-244, bound from Python by
 * gcn_extension.cpp:5-18,46-55).  Each entry point below cites the reference call site it serves.
 *
 * Conventions (all entry points):
 *   - every pointer is a DEVICE pointer owned by the caller and borrowed for the duration of the launch
 *     (as gcn_extension.cpp:46-55 borrows tensor.data_ptr()); nothing is allocated or freed on the data
 *     path; scratch memory is passed in by the caller (`workspace`);
 *   - `stream` is a hipStream_t passed as void*; launches are ASYNCHRONOUS on it (the reference launcher
 *     synchronises the whole device, gcn_fused_kernel.cu:229 -- deliberately not reproduced);
 *   - return value: 0 on success, negative DGLL_ERR_* otherwise; dgll_hip_last_error() gives the text
 *     (the reference calls exit(1), gcn_fused_kernel.cu:224-227 -- deliberately not reproduced);
 *   - CSR: int64 rowptr[n_rows+1], int32 col[nnz], optional fp32 val[nnz] (NULL = all ones);
 *     dense matrices are row-major with an explicit leading dimension in ELEMENTS;
 *   - dtypes: DGLL_F32 or DGLL_BF16 storage, fp32 accumulation always;
 *   - re-entrant and thread-safe: no global mutable state besides the thread-local error string.
 */
#ifndef DGLL_HIP_H
#define DGLL_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DGLL_HIP_ABI_VERSION 1

enum { DGLL_OK = 0, DGLL_ERR_INVALID = -1, DGLL_ERR_HIP = -2, DGLL_ERR_UNSUPPORTED = -3, DGLL_ERR_WORKSPACE = -4 };
enum { DGLL_F32 = 0, DGLL_BF16 = 1 };
enum { DGLL_REDUCE_SUM = 0, DGLL_REDUCE_MEAN = 1 };
/* epilogue bit-mask applied to a finished output row: y = act(scale*acc + bias) */
enum { DGLL_EPI_NONE = 0, DGLL_EPI_BIAS = 1, DGLL_EPI_RELU = 2 };

typedef struct dgll_csr_plan dgll_csr_plan; /* opaque: load-balancing schedule of one CSR structure */

/* ---- library -------------------------------------------------------------------------------------- */
int dgll_hip_abi_version(void);
const char* dgll_hip_last_error(void);
/* Fills name (<= name_len bytes), compute-unit count and total global memory of `device`. */
int dgll_hip_device_info(int device, char* name, int name_len, int* compute_units, int64_t* global_mem_bytes);

/* Diagnostics only: kernel tuning knobs used by tools/spmm_tune.py (0 unroll depth, 1 rows per wavefront, 2 flags,
 * 3 default long-row threshold).  Not part of the data path; not thread-safe. */
int dgll_hip_debug_tune(int key, int value);

/* ---- CSR schedule ----------------------------------------------------------------------------------
 * Built once per adjacency structure (the reference builds its adjacency once per graph,
 * nn/utils/utils.py:171,179).  Rows longer than `long_row_threshold` nonzeros (<= 0 selects the default,
 * 128) are split into chunks that are reduced in a fixed order, so results are bit-reproducible and
 * power-law rows do not serialise the launch.  Synchronises `stream` (it reads a count back).
 * long_row_threshold < 0 is the caller's guarantee that NO row exceeds the default threshold (a sampled block whose
 * fan-out is at most 128): the plan is then created on the host alone -- no scan, no allocation, no synchronisation --
 * which matters for blocks that are rebuilt every mini-batch.                                            */
int dgll_hip_csr_plan_create(void* stream, const int64_t* rowptr, int64_t n_rows, int64_t nnz,
                             int long_row_threshold, dgll_csr_plan** out_plan);
void dgll_hip_csr_plan_destroy(dgll_csr_plan* plan);
/* Scratch bytes dgll_hip_spmm_csr / dgll_hip_gat_* need for `feat` output columns with this plan. */
size_t dgll_hip_csr_plan_workspace_bytes(const dgll_csr_plan* plan, int feat);
int64_t dgll_hip_csr_plan_num_long_rows(const dgll_csr_plan* plan);
int64_t dgll_hip_csr_plan_num_chunks(const dgll_csr_plan* plan);

/* ---- a1 / a3 / a7-forward: Y[n_rows, feat] = epilogue( reduce_j A[i,j] * X[j, :] ) -------------------
 * Serves F.spmm(adj, support) (dgll/nn/Convolution/gcnconv.py:31, gcn.py:39), torch.sparse.mm
 * (Evaluation/PPI/gcn_model.py:76), the K-axis mean/sum of NeighborAggregator (sageconv.py:33-36, a
 * constant-degree CSR) and SpecialSpmmFunction.forward (gatconv.py:66-69).  With the transposed CSR it is
 * also every grad_X = A^T.g (autograd of the above; gatconv.py:80).
 * `plan` may be NULL (every row is then handled by one wavefront).  `val` may be NULL (unweighted).
 * `bias` is fp32[feat] and only read when epilogue has DGLL_EPI_BIAS.                                   */
int dgll_hip_spmm_csr(void* stream, const dgll_csr_plan* plan,
                      const int64_t* rowptr, const int32_t* col, const float* val,
                      const void* X, int64_t ldx, int x_dtype,
                      void* Y, int64_t ldy, int y_dtype,
                      int64_t n_rows, int64_t n_cols, int feat,
                      int reduce, int epilogue, const float* bias,
                      void* workspace, size_t workspace_bytes);

/* Same launch with two extras used by the partitioned (multi-GPU) path, where one destination row's neighbours are
 * split over an owned-columns CSR and a halo-columns CSR: `accumulate` != 0 adds the row already in Y before the
 * epilogue (Y = act(scale * (A.X + Y) + bias)); `row_scale` (fp32[n_rows], may be NULL) replaces the reduce's own
 * 1/nnz(row) so both halves share the full degree.  `accumulate` == 2 is the increment form, Y += gate(scale * A.X)
 * with rows that have no edge left untouched (no bias / ReLU): the halo half and the reduction of returned gradient
 * pieces touch a fraction of the rows, and re-writing all of them cost more than the gathers.                  */
int dgll_hip_spmm_csr_ex(void* stream, const dgll_csr_plan* plan,
                         const int64_t* rowptr, const int32_t* col, const float* val,
                         const void* X, int64_t ldx, int x_dtype, void* Y, int64_t ldy, int y_dtype,
                         int64_t n_rows, int64_t n_cols, int feat, int reduce, int epilogue, const float* bias,
                         void* workspace, size_t workspace_bytes, const float* row_scale, int accumulate);

/* dgll_hip_spmm_csr_ex plus a fused ReLU backward for the layer BELOW: where gate[i, c] <= 0 (gate: [n_rows, ldg] of Y's
 * dtype, the forward activations relu produced; may be NULL) the output element is written as 0.  Used when Y is the
 * gradient w.r.t. those activations (sageconv.py:81-82 applies the activation last): the separate masking pass over
 * [N, F] disappears.                                                                                              */
int dgll_hip_spmm_csr_gated(void* stream, const dgll_csr_plan* plan,
                            const int64_t* rowptr, const int32_t* col, const float* val,
                            const void* X, int64_t ldx, int x_dtype, void* Y, int64_t ldy, int y_dtype,
                            int64_t n_rows, int64_t n_cols, int feat, int reduce, int epilogue, const float* bias,
                            void* workspace, size_t workspace_bytes, const float* row_scale, int accumulate,
                            const void* gate, int64_t ldg);

/* ---- a7 backward: edge_out[k] = <G[row(k), :], B[col[k], :]> ------------------------------------------
 * The sampled dense-dense product SpecialSpmmFunction.backward computes through a dense N x N matmul
 * (gatconv.py:76-78).  G and B share `dtype`; both must be 16-byte aligned with leading dimensions padded to
 * a multiple of 16 bytes; feat <= 64 vectors of 16 bytes (256 fp32 / 512 bf16 columns).                      */
int dgll_hip_sddmm_csr(void* stream, const int64_t* rowptr, const int32_t* col,
                       const void* G, int64_t ldg, const void* B, int64_t ldb, int dtype,
                       float* edge_out, int64_t n_rows, int feat);

/* ---- a6 / a8 / a9: fused multi-head edge-softmax + aggregation ------------------------------------------
 * All `heads` attention heads of one layer in one launch (the reference loops heads in Python,
 * gatconv.py:168,196).  H is [n_cols, heads*fo] (the concatenated per-head X.W of gatconv.py:117), S and T are
 * fp32 [n, heads] with S[i,k] = a_k[:fo].h_i^k, T[j,k] = a_k[fo:].h_j^k (gatconv.py:122-125 without
 * materialising edge_h).  mode 0 = sparseGatConv: w_ij = exp(-leakyrelu_alpha(S_i+T_j)) (gatconv.py:125);
 * mode 1 = gatConv restricted to the adjacency's nonzeros: softmax_j(+leakyrelu) with the row maximum
 * subtracted (gatconv.py:34-36,53-54).  out_i = act(sum_j w_ij*scale_ij*h_j / sum_j w_ij), act = ELU when
 * apply_elu (gatconv.py:143-145); `edge_scale` (fp32 [nnz, heads] or NULL) carries attention-dropout
 * multipliers, applied after the row sum exactly as gatconv.py:129-135 orders them.  rowsum (and rowmax in
 * mode 1) are fp32 [n_rows, heads] outputs kept for the backward pass.  `fo` must make fo*sizeof(dtype)/16 a
 * power of two (the host pads each head with zero columns).                                                */
int dgll_hip_gat_fwd(void* stream, const dgll_csr_plan* plan, const int64_t* rowptr, const int32_t* col,
                     const void* H, int64_t ldh, const float* S, const float* T, const float* edge_scale,
                     void* out, int64_t ldo, int dtype, float* rowsum, float* rowmax,
                     int64_t n_rows, int heads, int fo, float alpha, int apply_elu, int mode,
                     void* workspace, size_t workspace_bytes);
/* Scratch for the long-row partials of dgll_hip_gat_fwd / _bwd with this plan (0 without long rows). `plan` (and
 * `t_plan`, the plan of A^T, in the backward) may be NULL: every row is then handled by one wavefront.          */
size_t dgll_hip_gat_workspace_bytes(const dgll_csr_plan* plan, int heads, int fo);

/* Backward of dgll_hip_gat_fwd: two gather passes (rows of A, then rows of A^T given by t_rowptr/t_col with
 * t_perm[k] = A's edge slot of A^T's k-th edge), nothing stored per edge.  Scratch: dn_scratch
 * [n_rows, ldn] in `dtype`, dd_scratch fp32 [n_rows, heads].  Outputs: grad_H [n_cols, ldgh] in `dtype`,
 * grad_S fp32 [n_rows, heads], grad_T fp32 [n_cols, heads].                                                 */
int dgll_hip_gat_bwd(void* stream, const dgll_csr_plan* plan, const dgll_csr_plan* t_plan,
                     const int64_t* rowptr, const int32_t* col,
                     const int64_t* t_rowptr, const int32_t* t_col, const int64_t* t_perm,
                     const void* H, int64_t ldh, const float* S, const float* T, const float* edge_scale,
                     const void* out, int64_t ldo, const void* grad_out, int64_t ldg, int dtype,
                     const float* rowsum, const float* rowmax,
                     void* dn_scratch, int64_t ldn, float* dd_scratch,
                     void* grad_H, int64_t ldgh, float* grad_S, float* grad_T,
                     int64_t n_rows, int64_t n_cols, int heads, int fo, float alpha, int apply_elu, int mode,
                     void* workspace, size_t workspace_bytes);   /* >= max of the two plans' workspace bytes */

/* The same three launches as separate entry points for the partitioned (multi-GPU) path, where a destination row's
 * neighbours are split over an owned-columns CSR and a halo-columns CSR (mode 0 only):
 *   dgll_hip_gat_fwd_ex   raw != 0: leave the row un-normalised (numerator in `out`, denominator in rowsum);
 *                         accumulate != 0: add the numerator / denominator already there, then (unless raw) normalise + ELU;
 *   dgll_hip_gat_bwd_rows pass 1 over one half; accumulate = 0: first (or only) launch, writes DN / DD / grad_S, dd_i = -DN_i . hp_i with
 *                         hp_i recovered from the stored output row; 1: a further launch over another column half (grad_S +=, DN / DD
 *                         left alone); 3: DECLARED the only launch over these rows (sparseGatConv form): dd_i is formed from the
 *                         pass's own dot products, sum_j w_ij (DN_i . h_j) / den_i -- exact in the stored operands (what
 *                         dgll_hip_gat_bwd and the *_strided entry points do);
 *   dgll_hip_gat_bwd_cols pass 2 over one transposed structure: rows = source nodes (Hrow, T_row), columns = destination
 *                         rows (dn, S_col, dd_col, rowmax_col).                                                     */
/* dgll_hip_gat_bwd_rows_split: the rows pass over SEVERAL column halves of A with the exact dd_i (ds_i is bilinear in the sums
 * (sum c.dot, sum c, sum w.dot), so no launch can finalise its own share): phase 4 = first launch (writes DN, parks the sums in
 * partial3), 5 = a middle one, 6 = the last (adds partial3, writes dd and grad_S).  partial3: fp32 [n_rows, 3 * heads], caller-owned,
 * the same buffer for every phase; mode 0 (sparseGatConv form) only.                                                           */
int dgll_hip_gat_bwd_rows_split(void* stream, const dgll_csr_plan* plan, const int64_t* rowptr, const int32_t* col,
                                const void* H, int64_t ldh, const float* S, const float* T, const void* out, int64_t ldo,
                                const void* grad_out, int64_t ldg, int dtype, const float* rowsum, void* dn, int64_t ldn, float* dd,
                                float* grad_S, int64_t n_rows, int heads, int fo, float alpha, int apply_elu, int phase,
                                float* partial3, void* workspace, size_t workspace_bytes);
int dgll_hip_gat_fwd_ex(void* stream, const dgll_csr_plan* plan, const int64_t* rowptr, const int32_t* col,
                        const void* H, int64_t ldh, const float* S, const float* T, const float* edge_scale,
                        void* out, int64_t ldo, int dtype, float* rowsum, int64_t n_rows, int heads, int fo, float alpha,
                        int apply_elu, void* workspace, size_t workspace_bytes, int raw, int accumulate);
int dgll_hip_gat_bwd_rows(void* stream, const dgll_csr_plan* plan, const int64_t* rowptr, const int32_t* col,
                          const void* H, int64_t ldh, const float* S, const float* T, const float* edge_scale,
                          const void* out, int64_t ldo, const void* grad_out, int64_t ldg, int dtype,
                          const float* rowsum, const float* rowmax, void* dn, int64_t ldn, float* dd, float* grad_S,
                          int64_t n_rows, int heads, int fo, float alpha, int apply_elu, int mode, int accumulate,
                          void* workspace, size_t workspace_bytes);
int dgll_hip_gat_bwd_cols(void* stream, const dgll_csr_plan* t_plan, const int64_t* t_rowptr, const int32_t* t_col,
                          const int64_t* t_perm, const void* dn, int64_t ldn, const void* Hrow, int64_t ldh,
                          const float* T_row, const float* S_col, const float* dd_col, const float* rowmax_col,
                          const float* edge_scale, void* grad_H, int64_t ldgh, float* grad_T, int dtype,
                          int64_t n_rows_t, int heads, int fo, float alpha, int mode, void* workspace, size_t workspace_bytes);

/* sparseGatConv's form (mode 0, no attention dropout) with STRIDED score arrays -- the layout the gather passes are
 * fastest with.  These passes are bound by cache-line fills per edge: four for a 512-byte feature row plus one for every
 * separate per-node array gathered per edge.  So:
 *   T           is read at T[j * t_stride + head]: a compact [n_cols, heads] array (t_stride = heads), or a slot in the
 *               PADDING of the H rows themselves (a 47-class output row is 94 of 128 bytes: T = (float*)((char*)H + 96),
 *               t_stride = ldh * elem_size / 4) -- the score then shares the row's cache line and costs nothing;
 *   sd_scratch  fp32 scratch, sd_stride floats per destination row (>= 2 * heads): the rows pass leaves
 *               {s_i[0:heads], dd_i[0:heads]} side by side there and the transposed pass gathers both with ONE line fill
 *               -- a separate [n_rows, 2 * heads] buffer or a slot in the padding of the dn_scratch rows;
 *   T_rows      the compact [n_cols, heads] T for the row side of the transposed pass.
 * `fo` (per-head width) may be any multiple of the 16-byte vector (8 bf16 / 4 fp32 elements): a head that does not fill its
 * power-of-two lane group leaves lanes idle, no columns are padded.  Gradients w.r.t. H, S, T as dgll_hip_gat_bwd.      */
int dgll_hip_gat_fwd_strided(void* stream, const dgll_csr_plan* plan, const int64_t* rowptr, const int32_t* col,
                             const void* H, int64_t ldh, const float* S, const float* T, int t_stride, void* out,
                             int64_t ldo, int dtype, float* rowsum, int64_t n_rows, int heads, int fo, float alpha,
                             int apply_elu, void* workspace, size_t workspace_bytes);
/* The forward pass with the gathered-side scores FORMED from the gathered rows: t_j = sum_f H[j, head fo + f] . attn2[head fo + f], as
 * sparseGatConv itself builds its logit from the gathered rows (gatconv.py:122-125) -- no score row is fetched per edge (one cache
 * line fewer per edge than dgll_hip_gat_fwd_strided when the rows fill their lines).  attn2: fp32 [heads * fo], a2 of every head laid
 * out like a row of H (bf16 storage: rounded to bf16 for the packed dot product, as the score product of the caller rounds it).  S as
 * before ([n_rows, heads] fp32).  raw / accumulate as dgll_hip_gat_fwd_ex.  sparseGatConv's form only (exp(-leakyrelu), no dropout).
 * n_cols = rows of H: the gathers use 32-bit byte offsets, so H may hold at most 2^24 rows and 4 GB (DGLL_ERR_UNSUPPORTED past
 * that: take dgll_hip_gat_fwd_strided).                                                                                         */
int dgll_hip_gat_fwd_rowscore(void* stream, const dgll_csr_plan* plan, const int64_t* rowptr, const int32_t* col,
                              const void* H, int64_t ldh, const float* S, const float* attn2, void* out, int64_t ldo, int dtype,
                              float* rowsum, int64_t n_rows, int64_t n_cols, int heads, int fo, float alpha, int apply_elu,
                              void* workspace, size_t workspace_bytes, int raw, int accumulate);
/* The rows pass of the backward in the same form (dgll_hip_gat_bwd_rows_strided with attn2 instead of T / t_stride; n_cols = rows of
 * H, bounded as above).                                                                                                          */
int dgll_hip_gat_bwd_rows_rowscore(void* stream, const dgll_csr_plan* plan, const int64_t* rowptr, const int32_t* col,
                                   const void* H, int64_t ldh, const float* S, const float* attn2,
                                   const void* out, int64_t ldo, const void* grad_out, int64_t ldg, int dtype,
                                   const float* rowsum, void* dn_scratch, int64_t ldn, float* sd_scratch, int sd_stride,
                                   float* grad_S, int64_t n_rows, int64_t n_cols, int heads, int fo, float alpha, int apply_elu,
                                   void* workspace, size_t workspace_bytes);
/* the two passes of dgll_hip_gat_bwd_strided one by one (rows of A: DN, {s, dd}, grad_S; then rows of A^T: grad_H, grad_T) */
int dgll_hip_gat_bwd_rows_strided(void* stream, const dgll_csr_plan* plan, const int64_t* rowptr, const int32_t* col,
                                  const void* H, int64_t ldh, const float* S, const float* T, int t_stride,
                                  const void* out, int64_t ldo, const void* grad_out, int64_t ldg, int dtype,
                                  const float* rowsum, void* dn_scratch, int64_t ldn, float* sd_scratch, int sd_stride,
                                  float* grad_S, int64_t n_rows, int heads, int fo, float alpha, int apply_elu,
                                  void* workspace, size_t workspace_bytes);
/* attn1 / attn2 / grad_S_rows (all three or none): when the scores are S = H.a1, T = H.a2 per head (gatconv.py:122-125), their
 * own contribution to grad_H is added in the epilogue -- grad_H[j, f] += grad_S[j, head(f)] * attn1[f] + grad_T[j, head(f)] *
 * attn2[f], attn* fp32 [heads * fo] laid out like a row of H, grad_S_rows fp32 [n_cols, heads] from the rows pass -- instead of
 * in a separate [n, 2 heads] x [2 heads, heads * fo] product and an add over [n, heads * fo].                              */
int dgll_hip_gat_bwd_cols_strided(void* stream, const dgll_csr_plan* t_plan, const int64_t* t_rowptr,
                                  const int32_t* t_col, const void* dn_scratch, int64_t ldn, const void* H,
                                  int64_t ldh, const float* T_rows, const float* sd_scratch, int sd_stride,
                                  void* grad_H, int64_t ldgh, float* grad_T, int dtype, int64_t n_cols, int heads,
                                  int fo, float alpha, void* workspace, size_t workspace_bytes,
                                  const float* attn1, const float* attn2, const float* grad_S_rows);
int dgll_hip_gat_bwd_strided(void* stream, const dgll_csr_plan* plan, const dgll_csr_plan* t_plan,
                             const int64_t* rowptr, const int32_t* col, const int64_t* t_rowptr, const int32_t* t_col,
                             const void* H, int64_t ldh, const float* S, const float* T, int t_stride,
                             const float* T_rows, const void* out, int64_t ldo, const void* grad_out, int64_t ldg,
                             int dtype, const float* rowsum, void* dn_scratch, int64_t ldn, float* sd_scratch,
                             int sd_stride, void* grad_H, int64_t ldgh, float* grad_S, float* grad_T, int64_t n_rows,
                             int64_t n_cols, int heads, int fo, float alpha, int apply_elu, void* workspace,
                             size_t workspace_bytes);

/* ---- a10 / a4 as ONE launch: aggregate -> transform (bf16 storage, fp32 accumulation) ------------------------------------
 *   out[i, :] = act( A1[i, :K1] . W1 + reduce_{j in row i} X[j, :feat] . W2 + bias )
 * The MI355X form of the reference's only native kernel, relu(A.X.W) in one launch
 * (dgll/FusedKernel/gcn_fused_kernel.cu:5-74; A1 == NULL gives exactly that shape), and of sageConv's
 * act(src.W_s + mean(nbr).W_n) (sageconv.py:33-41,70-83): a workgroup gathers and reduces a tile of 32 destination rows
 * into LDS and feeds it to the MFMAs directly -- the aggregated matrix is never read back from HBM.
 *   Wt1 / Wt2     the TRANSPOSED weights, bf16, zero-padded to [w_rows >= 32 * ceil(N / 32), 64 * ceil(K / 64)] (ldw*: row
 *                 stride in elements).  Wt2 == NULL: the aggregate is ADDED to the output instead (feat == N) -- the
 *                 narrowing layer, whose neighbour product is aggregated after its transform;
 *   agg_out       optional [n_rows, ldagg] bf16: the aggregated rows (training keeps them for the weight gradient
 *                 agg^T . g).  REQUIRED when the plan has rows longer than its threshold: those are aggregated first by the
 *                 SpMM's chunk path into agg_out and loaded from there (workspace as for dgll_hip_spmm_csr);
 *   feat, N <= 256; X / A1 / out / agg_out rows 16-byte aligned; val NULL = unit edge values.                        */
int dgll_hip_sage_fused_forward(void* stream, const dgll_csr_plan* plan, const int64_t* rowptr, const int32_t* col,
                                const float* val, const void* X, int64_t ldx, int feat, int reduce,
                                const void* A1, int64_t lda1, int K1, const void* Wt1, int64_t ldw1,
                                const void* Wt2, int64_t ldw2, int w_rows, const float* bias, int relu, void* out,
                                int64_t ldo, int N, void* agg_out, int64_t ldagg, int64_t n_rows, int64_t n_cols,
                                void* workspace, size_t workspace_bytes);

/* ---- a3 (max): Y[i,f] = max_k X[col[k], f], arg[i,f] = the source row holding it (-1 / 0.0 for empty rows) --
 * NeighborAggregator's "max" (sageconv.py:37-38).  Y and arg share the leading dimension ldy; X/Y 16-byte
 * aligned with padded leading dimensions.                                                                   */
int dgll_hip_segment_max(void* stream, const int64_t* rowptr, const int32_t* col, const void* X, int64_t ldx,
                         void* Y, int32_t* arg, int64_t ldy, int dtype, int64_t n_rows, int feat);

/* Backward of the max: grad[j,f] = sum_{i lists j} (arg[i,f] == j) ? G[i,f] : 0  -- the gradient reaches the arg-max row
 * only (torch.max semantics behind sageconv.py:37-38).  (t_rowptr, t_col) is the TRANSPOSED structure (row j lists the
 * destination rows i in ascending order; a repeated pair counts once); one wavefront per output row, fixed order, no
 * atomics.  G / grad 16-byte aligned with leading dimensions padded like segment_max's; arg as segment_max wrote it. */
int dgll_hip_segment_max_bwd(void* stream, const int64_t* t_rowptr, const int32_t* t_col, const void* G, int64_t ldg,
                             const int32_t* arg, int64_t ldarg, void* grad, int64_t ldgrad, int dtype, int64_t n_src,
                             int feat);

/* ---- f2: feature-row gather through the hot-node cache -------------------------------------------------------
 * out[i,:] = cache[slot[idx[i]],:] if slot[idx[i]] >= 0 else host[idx[i],:]   -- GraphCacheServer.fetch_data,
 * dgll/FeatureCache/storage.py:151-198 (mask split, GPU gather of cached rows, CPU gather + copy of the rest, merge)
 * in one launch; `host` may be PINNED HOST memory (read over PCIe by the kernel) or a device matrix.  slot == NULL:
 * plain gather from `host` (dgll/data/dgraph.py:105 `features[nodes]`).  *miss_count (optional, device) is
 * incremented by the number of rows served from `host` (storage.py:213-220).                                 */
int dgll_hip_gather_rows(void* stream, const void* cache, int64_t ldc, const void* host, int64_t ldh,
                         const int64_t* idx, const int64_t* slot, void* out, int64_t ldo, int64_t n, int feat,
                         int dtype, unsigned long long* miss_count);

/* Same gather when the cache server holds a PARTITION of the graph: idx / slot are keyed by the partition-local id and
 * host_map[local id] is the row of `host` (the full-graph feature store) -- storage.py:27 `nid_map`, :104-107.  The
 * hit / miss split, both gathers and the miss count stay one launch; host_map == NULL is dgll_hip_gather_rows.      */
int dgll_hip_gather_rows_mapped(void* stream, const void* cache, int64_t ldc, const void* host, int64_t ldh,
                                const int64_t* idx, const int64_t* slot, const int64_t* host_map, void* out,
                                int64_t ldo, int64_t n, int feat, int dtype, unsigned long long* miss_count);

/* The reduction over a sampled block read STRAIGHT from the feature store:
 *   out[i,:] = reduce_{k in [rowptr[i], rowptr[i+1])} row(idx[k]),  row(v) = cache[slot[v],:] if slot[v] >= 0 else host[host_map[v] or v,:]
 * -- the K-axis mean of sageconv.py:33-36 over the outermost hop's neighbour features fused with GraphCacheServer.fetch_data
 * (storage.py:151-198): the fan-out x batch gathered rows are never materialised.  reduce: DGLL_REDUCE_SUM / _MEAN (empty rows: 0).
 * rowptr: int64 [n_rows + 1] over idx.  Rows must be 4-byte granular; 16-byte lanes are used when every pitch is a whole number of
 * 16-byte vectors (the padding columns are then summed and written as well).  *miss_count as in dgll_hip_gather_rows.            */
int dgll_hip_aggregate_rows_mapped(void* stream, const void* cache, int64_t ldc, const void* host, int64_t ldh,
                                   const int64_t* idx, const int64_t* slot, const int64_t* host_map, const int64_t* rowptr,
                                   void* out, int64_t ldo, int64_t n_rows, int feat, int dtype, int reduce,
                                   unsigned long long* miss_count);

/* The outermost hop of a natively sampled batch leaves the host as neighbour POSITIONS inside each seed's adjacency list
 * (dgll_host_sample_batch_seeded with defer_last; reference loop: base_sampler.py:45-58 keeps `neighbors[j]`, the lookup
 * `g.get_neighbors` is dgll/data/dgraph.py:49-62).  One launch turns them into node ids on the device:
 *     out_ids[k] = indices[ indptr[ seeds[r] ] + positions[k] ]      for k in [rowptr[r], rowptr[r + 1])
 * indptr / indices: the graph's CSR arrays in device memory (int64); seeds [n_rows] int64; rowptr [n_rows + 1] int64 (the hop's
 * row pointers: rowptr[r + 1] - rowptr[r] kept neighbours of seed r); positions: int16 / int32 / int64 entries (pos_bytes = 2, 4,
 * 8), rowptr[n_rows] of them.  Replaces five torch launches (gather of the starts, repeat_interleave, widening, add, gather). */
int dgll_hip_translate_positions(void* stream, const int64_t* indptr, const int64_t* indices, const int64_t* seeds,
                                 const int64_t* rowptr, int64_t n_rows, const void* positions, int pos_bytes, int64_t* out_ids);

/* Backward of the K-axis reduction over a SAMPLED block (sageconv.py:33-36 on a block whose source rows each belong to exactly one
 * destination row: col == arange(nnz), base_sampler.py:30-43 keeps duplicates): source row k of destination row r receives
 *     out[k, :] = scale_r * g[r, :]        scale_r = 1 / deg(r) for the mean (mean != 0), 1 for the sum,
 * k in [rowptr[r], rowptr[r + 1]); rows rowptr[n_rows] .. n_out_rows-1 of `out` (the unused tail of a block on static shapes) are
 * zeroed.  One launch instead of the degree / reciprocal / scale / searchsorted / gather chain of tensor ops (nine launches per
 * block and batch).  dtype: DGLL_F32 or DGLL_BF16 for both matrices (fp32 arithmetic); leading dimensions in elements.
 * accumulate != 0: out[k, :] += scale_r * g[r, :] instead (rows behind rowptr[n_rows] untouched): the rows already hold another
 * contribution to the same gradient (the layer's self path), written by an earlier launch.                                      */
int dgll_hip_expand_rows(void* stream, const int64_t* rowptr, int64_t n_rows, const void* g, int64_t ldg, void* out, int64_t ldo,
                         int64_t n_out_rows, int feat, int dtype, int mean, int accumulate);

/* The LOADING STAGE of one sampled mini-batch as ONE call (buffer_queues.py:22-46's `sample_generator` body: stage the batch on
 * the side stream; storage.py:151-198's fetch per hop; graphage.py:52-53's labels): everything the stage enqueues for a batch --
 *   1. the upload of the batch's staging buffer (seeds | source ids per hop | row pointers per hop: dgll_host_sample_batch_seeded's
 *      arrays at their upper-bound offsets) and of the outermost hop's neighbour positions, from pinned host memory,
 *   2. positions -> node ids of the outermost hop (dgll_hip_translate_positions),
 *   3. one cache gather per hop 0 .. n_hops-1 straight into that hop's rows of the consumer's input (dgll_hip_gather_rows_mapped),
 *   4. the outermost hop's reduction straight out of the cache (dgll_hip_aggregate_rows_mapped),
 *   5. the row pointers of hops 0 .. n_hops-2 padded to the consumer's static block shapes (entries past the batch's rows = its
 *      edge count: empty rows), and the seeds' labels (entries past the batch = label_fill) --
 * is issued on `stream` from native code: through Python the same work is ~15 launches and 1.0-1.8 ms of interpreter time per
 * batch on the loading thread, which bounds a pipeline whose GPU side takes 1.6 ms.  All pointers in device memory unless named
 * *_host (pinned).  Offsets are in int64 entries of the staging buffer.  rows[h] = rows of hop h (rows[h + 1] = edges of hop h);
 * n_outer = edges of the outermost hop.  miss_count: optional device counter (hit / miss accounting, storage.py:213-220).      */
typedef struct dgll_batch_load {
    const void* staged_host; int64_t staged_entries; int64_t* staged_dev;
    const void* pos_host; int64_t n_outer; int pos_bytes; void* pos_dev;
    const int64_t* indptr; const int64_t* indices;
    int n_hops; int64_t rows[8]; int64_t seeds_off; int64_t src_off[8]; int64_t ptr_off[8];
    const void* cache; int64_t ldc; const void* host; int64_t ldh; const int64_t* slot; const int64_t* host_map;
    int feat; int dtype; unsigned long long* miss_count;
    void* feat_out[8]; int64_t ld_feat;
    void* reduced_out; int64_t ld_reduced; int reduce;
    int64_t* ids_out;
    int64_t* rowptr_out[8]; int64_t rowptr_cap[8];
    const int64_t* labels; int64_t* labels_out; int64_t labels_cap; int64_t label_fill;
    /* Optional (stage_map NULL, or no slot map: the uncached rows of the outermost hop are read zero-copy by its reduction): fetch them
     * into HBM first -- every DISTINCT uncached node of the hop once, by a small grid that the link, not the CUs, bounds -- so that the
     * reduction reads HBM only and does not hold the chip while it waits for PCIe.  stage_map: int64[n_nodes] scratch of ONE loading
     * stream, zero-initialised once, never cleared (entries carry stage_serial, which the caller advances per call, never 0);
     * stage_rows: [stage_cap, ld_stage] elements of the store's dtype; stage_list: int64[stage_cap]; stage_count: one device word.
     * Nodes past stage_cap stay zero-copy reads.  stage_blocks: workgroups of the fetch (0 = about 192 KB of reads in flight: 20 at 1204-byte rows).                                   */
    int64_t* stage_map; void* stage_rows; int64_t ld_stage; int64_t stage_cap; int64_t* stage_list; unsigned int* stage_count;
    unsigned int stage_serial; int stage_blocks;
    int upload_blocks;      /* workgroups of the two uploads (0 = 16, DGLL_LOADER_UPLOAD_BLOCKS): few beside staged misses, more (128) when
                             * nothing else uses the link                                                                          */
} dgll_batch_load;
int dgll_hip_load_sampled_batch(void* stream, const dgll_batch_load* batch);

/* ---- f1 (host code): one hop of the reference's neighbour sampler, bit-exact with CPython 3.10's random.sample -------
 * For every seed in order: all neighbours if deg <= fanout (or fanout < 0), else random.sample(neighbors, fanout)
 * (/root/reference/dgll/sampling/base_sampler.py:45-58), drawn from the MT19937 state passed in (`random.getstate()`:
 * 624 words + index) and updated in place, so the ids and the generator stream are identical to the Python loop.
 * indptr/indices: CSR copy of DGraph.edges (HOST pointers, like every argument of this call).  `setsize` is
 * CPython's pool/set switch-over (21, or 21 + 4**ceil(log(3*fanout, 4)) when fanout > 5), computed by the caller.  */
int dgll_host_sample_neighbors(uint32_t* mt_state, int* mt_index, const int64_t* indptr, const int64_t* indices,
                               const int64_t* seeds, int64_t n_seeds, int64_t fanout, int64_t setsize,
                               int64_t* out_src, int64_t* out_dst, int64_t* out_counts, int64_t capacity, int64_t* n_out);
/* out_dst == NULL above stops after the (sequential) draw phase: out_src then holds POSITIONS inside each seed's adjacency
 * list.  This call finishes the job -- positions -> neighbour ids, out_dst = the seed of every edge -- and touches no
 * generator state, so a pipeline can run it on another thread while the next batch is being drawn.              */
int dgll_host_translate_neighbors(const int64_t* indptr, const int64_t* indices, const int64_t* seeds, int64_t n_seeds,
                                  const int64_t* counts, int64_t* src_inout, int64_t* out_dst);
/* random.seed(int) of CPython (init_by_array over `key` = the 32-bit little-endian words of abs(seed), [0] for 0) -> the 624-word
 * state and index (624) the calls above take: a sampler stream that lives OUTSIDE the interpreter's global generator.          */
int dgll_host_mt_seed(const uint32_t* key, int64_t key_len, uint32_t* mt_state, int* mt_index);
/* A whole mini-batch under its own seed: what the reference's loop (dgllsampler.py:10-21 over base_sampler.py:45-58) draws when
 * random.seed(seed) is called right before the batch.  Individually seeded batches are independent, so several host threads may
 * each draw whole batches concurrently, every one bit-identical to the reference loop under its seed (the reference's loop is
 * unseeded, base_sampler.py:56; its DataLoader-with-workers shape, MQGCN.py:114-128, gives every worker its own stream too).
 * Hops in SAMPLING order (reversed(fanouts)): hop h draws around the sources of hop h-1 (hop 0 around `seeds`), duplicates kept;
 * fanouts[h] < 0 = all neighbours; setsizes[h] as in dgll_host_sample_neighbors.  out_src/out_dst/out_counts: n_hops caller-owned
 * arrays of capacity[h] edges (counts: one per hop seed).  defer_last: the last hop keeps POSITIONS in out_src[h], out_dst[h] is
 * not written.  max_threads bounds the helper threads of the id translation inside this call.                                  */
int dgll_host_sample_batch_seeded(const uint32_t* key, int64_t key_len, const int64_t* indptr, const int64_t* indices,
                                  const int64_t* seeds, int64_t n_seeds, const int64_t* fanouts, const int64_t* setsizes, int n_hops,
                                  int64_t* const* out_src, int64_t* const* out_dst, int64_t* const* out_counts,
                                  const int64_t* capacity, int64_t* n_out, int defer_last, int max_threads);

/* ---- f1 / f2 (host code): a pool of native sampler threads behind one in-order hand-over -- the first of the reference's three
 * queues (README.md:27-29; buffer_queues.py:22-46's sample_generator) without an interpreter in the producers.  `n_threads` workers
 * draw the batches of one epoch -- batch i = train_nodes[i * batch_size, (i + 1) * batch_size), under
 * random.seed((base_seed << 40) | (epoch << 20) | i): exactly dgll_host_sample_batch_seeded's ids (outermost hop left as positions) --
 * each into slot i % n_slots of caller-owned (pinned) buffers:
 *   staged_bufs[slot]  int64[staged_entries]: seeds at off_seeds, the source ids of hop h < n_hops - 1 at off_src[h], the row pointers
 *                      of hop h (rows + 1 prefix sums of the kept-neighbour counts) at off_ptr[h] -- ONE upload for the loading stage;
 *   pos_bufs[slot]     the outermost hop's neighbour POSITIONS as pos_bytes-byte (2 / 4 / 8) unsigned integers.
 * fanouts / setsizes in SAMPLING order (the reference's reversed(fanouts)), setsizes as for dgll_host_sample_neighbors.
 * n_slots >= n_threads + 1; base_seed < 2**24, epoch and the batch count < 2**20 (the seed is one 64-bit word).
 * _next: the next batch IN ORDER, blocking: 0 = out[0] batch, out[1] slot, out[2] n_hops, out[3 ..) rows per hop, then edges per hop;
 *        1 = epoch over; < 0 = a worker failed.  ONE consumer thread.   _release: the slot's uploads are complete, it may be rewritten.
 * _destroy: stops and joins the workers.  Threading: the pool's own; the adjacency arrays are read-only and shared.               */
typedef struct dgll_sampler_pool dgll_sampler_pool;
int dgll_host_sampler_pool_create(dgll_sampler_pool** out, const int64_t* indptr, const int64_t* indices, const int64_t* train_nodes,
                                  int64_t n_train, int64_t batch_size, const int64_t* fanouts, const int64_t* setsizes, int n_hops,
                                  uint64_t base_seed, uint64_t epoch, int n_threads, int n_slots, int64_t* const* staged_bufs,
                                  int64_t staged_entries, int64_t off_seeds, const int64_t* off_src, const int64_t* off_ptr,
                                  void* const* pos_bufs, int pos_bytes);
int dgll_host_sampler_pool_next(dgll_sampler_pool* pool, int64_t* out, double* sample_ms);
int dgll_host_sampler_pool_release(dgll_sampler_pool* pool, int slot);
int dgll_host_sampler_pool_destroy(dgll_sampler_pool* pool);

/* ---- the optimizer step of the training loops as ONE launch (torch.optim.Adam's arithmetic; MQGCN.py:141-144, train_gcn.py:26) --
 * Adam over a flat fp32 parameter buffer: param/grad/exp_avg/exp_avg_sq [n]; `step` = 1, 2, ... (bias corrections formed in double
 * on the host); grad_scale multiplies the gradient first (1 / world_size of the RaCoM average, MQGCN.py:64).  Segments (optional,
 * at most 24): element range [seg_begin, seg_end) is a [rows, seg_cols] weight matrix whose updated values are ALSO written, as
 * bf16, into seg_packed ([rows, seg_ld], W itself) and seg_packed_t ([cols, seg_ldt], W transposed) -- the zero-padded forms
 * dgll_hip_transform_bf16 takes (what dgll_hip_pack_weight_bf16 produces per product otherwise); either pointer may be NULL.   */
int dgll_hip_adam_flat(void* stream, float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr,
                       float beta1, float beta2, float eps, float weight_decay, int64_t step, float grad_scale, int n_segments,
                       const int64_t* seg_begin, const int64_t* seg_end, const int* seg_cols, void* const* seg_packed,
                       const int64_t* seg_ld, void* const* seg_packed_t, const int64_t* seg_ldt);

/* ---- dense transform, exact fp32: C[M,N] = act(A[M,K].B[K,N] + bias) --------------------------------------
 * F.mm / F.matmul of gcnconv.py:30, sageconv.py:41,72, gatconv.py:31,117 for callers that only have the C ABI
 * (fmaf accumulation in k order: bit-stable).  relu != 0 fuses max(.,0); bias may be NULL.                   */
int dgll_hip_gemm_f32(void* stream, const float* A, int64_t lda, const float* B, int64_t ldb, float* C, int64_t ldc,
                      int64_t M, int N, int K, const float* bias, int relu);

/* ---- fp32 dense products on the matrix cores (the 1e-4 parity path of the layers) -------------------------------------------
 * dgll_hip_mm_f32:  C[M, N] = act(A[M, K] . Wt[N, K]^T + addend + bias), everything fp32, v_mfma_f32_32x32x2_f32 (fp32 inputs, fp32
 * accumulation: exact fp32 FMA arithmetic).  Wt is the weight TRANSPOSED ([N, K] row-major; for an input gradient g . W^T pass
 * Wt := W).  N <= 256 per call (split columns on the host); any alignment (16-byte aligned rows take float4 loads).
 * dgll_hip_mm2_f32:  C = gate(act(A1 . W1t^T + A2 . W2t^T + addend + bias)): sageConv's self + neighbour term (sageconv.py:72-75), or
 * a layer's two input-gradient products, in ONE accumulation; A2 may be NULL (then W2t / K2 are ignored); gate (optional, [M, ldgate]):
 * outputs are zeroed where gate <= 0 -- the ReLU mask of the layer below applied by the epilogue.  N <= 256 per call.
 * dgll_hip_grad_weight_f32:  dW[K, N] = X[M, K]^T . G[M, N] on the same instruction: the long reduction is split over `slabs` row
 * slabs (rows summed in order inside a slab) whose partials (workspace: dgll_hip_grad_weight_f32_workspace bytes) are summed in
 * slab order -- deterministic, no atomics.                                                                                  */
int dgll_hip_mm_f32(void* stream, const float* A, int64_t lda, const float* Wt, int64_t ldw, float* C, int64_t ldc,
                    int64_t M, int N, int K, const float* bias, int relu, const float* addend, int64_t ldadd);
int dgll_hip_mm2_f32(void* stream, const float* A1, int64_t lda1, const float* W1t, int64_t ldw1, int K1, const float* A2,
                     int64_t lda2, const float* W2t, int64_t ldw2, int K2, float* C, int64_t ldc, int64_t M, int N,
                     const float* bias, int relu, const float* addend, int64_t ldadd, const float* gate, int64_t ldgate);
int64_t dgll_hip_grad_weight_f32_workspace(int K, int N, int slabs);
int dgll_hip_grad_weight_f32(void* stream, const float* X, int64_t ldx, const float* G, int64_t ldg, float* dW,
                             int64_t lddw, int64_t M, int K, int N, void* workspace, int64_t workspace_bytes, int slabs);

/* Wt operand of the transforms from a parameter: dst[r, c] = bf16(src[r*stride_row + c*stride_col]) for r < n, c < k, zero
 * elsewhere, dst bf16 [rows >= n, ld >= k].  src fp32 or bf16, any element strides (pass a [K, N] parameter as its transposed view:
 * stride_row = 1, stride_col = N) -- the cast, the zero padding and the layout in one launch.                                    */
int dgll_hip_pack_weight_bf16(void* stream, const void* src, int src_dtype, int64_t stride_row, int64_t stride_col, int n, int k,
                              void* dst, int64_t ld, int rows);
/* ---- bf16 MFMA transform: out[M,N] = act( A1[M,K1].Wt1[N,K1]^T (+ A2[M,K2].Wt2[N,K2]^T) + bias ) -------------
 * The dense W-transform next to the aggregation on matrix cores (v_mfma_f32_32x32x16_bf16, fp32 accumulation):
 * sageConv's act(src.W_s + agg.W_n) in ONE pass (sageconv.py:71-82), gcnConv / GAT x.W (gcnconv.py:30,
 * gatconv.py:31,117), and their input gradients g.W^T (pass Wt := W; `relu_mask`, same shape as A1, fuses the
 * ReLU backward: A1 is zeroed where the mask is <= 0).  A*: bf16 row-major, 16-byte aligned, lda a multiple of 8.
 * Wt*: the weight TRANSPOSED, bf16, zero padded to [64 rows if N <= 64, 128 if N <= 128, else 256] x
 * [ld >= 64*ceil(K/64) columns] (the kernel stages 2, 4 or 8 column tiles of 32 rows of Wt).  N <= 256.
 * out: bf16 or fp32 [M, ldo].  A2/Wt2/relu_mask/bias may be NULL.
 * relu: bit 0 = fuse max(., 0); bit 1 (bf16 output, ldo a multiple of 8, 16-byte aligned) = the row padding [N, ldo)
 * belongs to the output and is written with zeros: rows are stored as whole 16-byte vectors / whole lines (a 47-column
 * row on a 128-byte pitch: 256 -> 47 runs at the speed of 256 -> 64 instead of 25 % slower).                         */
int dgll_hip_transform_bf16(void* stream, const void* A1, int64_t lda1, int K1, const void* Wt1, int64_t ldw1,
                            const void* A2, int64_t lda2, int K2, const void* Wt2, int64_t ldw2, int wt_rows,
                            const void* relu_mask, int64_t ldm, void* out, int64_t ldo, int out_dtype,
                            int64_t M, int N, int relu, const float* bias);   /* wt_rows: rows allocated in Wt1/Wt2 */
/* The same with an OUTPUT gate: out[i, n] is written as 0 where out_gate[i, n] <= 0 (bf16 [M, ldgate], may be NULL) --
 * the input gradient g.Ws^T + gz.Wn^T of a layer whose input came out of a ReLU, masked in the epilogue -- and an
 * optional per-row factor (fp32 [M], may be NULL): out = act(row_scale[i] * (A.W) + bias), e.g. the 1/deg of a mean
 * aggregation folded into the product that is aggregated next, so that SpMM runs unweighted.                       */
int dgll_hip_transform_bf16_gated(void* stream, const void* A1, int64_t lda1, int K1, const void* Wt1, int64_t ldw1,
                                  const void* A2, int64_t lda2, int K2, const void* Wt2, int64_t ldw2, int wt_rows,
                                  const void* relu_mask, int64_t ldm, void* out, int64_t ldo, int out_dtype,
                                  int64_t M, int N, int relu, const float* bias, const void* out_gate, int64_t ldgate,
                                  const float* row_scale);

/* The same transform with a bf16 [M, ldadd] matrix added before the activation:
 * out = act(row_scale * (A1.Wt1^T + A2.Wt2^T) + bias + addend) -- the self term of a transform-first SAGE layer, whose
 * neighbour term arrives already aggregated (sageconv.py:72-75 computes src.W + aggregated as two ops and an add). */
int dgll_hip_transform_bf16_add(void* stream, const void* A1, int64_t lda1, int K1, const void* Wt1, int64_t ldw1,
                                const void* A2, int64_t lda2, int K2, const void* Wt2, int64_t ldw2, int wt_rows,
                                const void* relu_mask, int64_t ldm, void* out, int64_t ldo, int out_dtype, int64_t M,
                                int N, int relu, const float* bias, const void* out_gate, int64_t ldgate,
                                const float* row_scale, const void* addend, int64_t ldadd);

/* The transform with the ReLU gate carried as ONE BIT PER ELEMENT instead of a bf16 matrix (bf16 output; no input mask, row
 * scale or addend).  The reference keeps the activation itself for autograd's ReLU backward (models/sage.py applies F.relu between
 * sageConv layers; sageconv.py:72-82 the in-layer activation): 512 bytes of a 256-column row are read back only to learn 256 signs.
 *   bits_out (may be NULL): uint32 [M, ld_bits_out]; word w of row i, bit b <- out[i, 32 w + b] > 0, from the values the epilogue
 *     stores (stores only: the producing transform is no slower); words up to 4 * ceil(N / 128) are written, zero past N;
 *   gate_bits (may be NULL): such a matrix, read INSTEAD of out_gate: out[i, n] is written as 0 where its bit is clear.  The
 *     resident-weights kernel fetches a block's words one reduction phase ahead of its epilogue; shapes that run the 4-wave
 *     kernel read out_gate (pass both; gate_bits alone is an error there).  Both describe the same gate: out_gate[i, n] > 0.
 * ld_*: words per row, multiples of 4, >= 4 * ceil(N / 128); 16-byte aligned bases.  Other arguments as dgll_hip_transform_bf16. */
int dgll_hip_transform_bf16_bits(void* stream, const void* A1, int64_t lda1, int K1, const void* Wt1, int64_t ldw1,
                                 const void* A2, int64_t lda2, int K2, const void* Wt2, int64_t ldw2, int wt_rows,
                                 void* out, int64_t ldo, int64_t M, int N, int relu, const float* bias,
                                 const void* out_gate, int64_t ldgate, const uint32_t* gate_bits, int64_t ld_gate_bits,
                                 uint32_t* bits_out, int64_t ld_bits_out);

/* Two products of ONE activation matrix: out1 = A.Wt1^T and out2 = A.Wt2^T, A read once -- the two input gradients
 * g.Ws^T and g.Wn^T of a SAGE layer (what autograd derives for sageconv.py:72-75's `src @ W` pair).  bf16 A [M, lda],
 * Wt1 / Wt2 [wt_rows >= 256, ldw] zero-padded as dgll_hip_transform_bf16 wants them, bf16 out1 / out2 [M, ldo >= N];
 * K, N <= 256.                                                                                                      */
int dgll_hip_transform_bf16_dual(void* stream, const void* A, int64_t lda, int K, const void* Wt1, const void* Wt2,
                                 int64_t ldw, int wt_rows, void* out1, int64_t ldo1, void* out2, int64_t ldo2, int64_t M,
                                 int N);

/* ---- weight gradients of the dense transform: dW1 = X1^T . G and (optionally) dW2 = X2^T . G ---------------------
 * What autograd derives for `self.W(h)` / `neigh @ self.W` (sageconv.py:41,72-75; gcnconv.py:30; the reference leaves it
 * to two library GEMMs that each read G).  bf16 X1 [M, ldx1 >= K1], X2 [M, ldx2 >= K2] (NULL / K2 = 0: one product),
 * G [M, ldg >= N]; fp32 dW1 [K1, lddw1 >= N], dW2 [K2, lddw2 >= N]; K1, K2, N <= 256; rows 16-byte aligned (bases, and leading
 * dimensions multiples of 8).  Split over `n_slabs` row slabs (1..4096) whose partials are summed in slab order (deterministic);
 * `workspace`: dgll_hip_grad_weight_workspace(K1, K2, n_slabs) bytes of device memory.  M = 0 (inputs may be NULL) writes zeros.      */
int64_t dgll_hip_grad_weight_workspace(int K1, int K2, int n_slabs);
int dgll_hip_grad_weight_bf16(void* stream, const void* X1, int64_t ldx1, int K1, const void* X2, int64_t ldx2, int K2,
                              const void* G, int64_t ldg, int N, int64_t M, void* workspace, int64_t workspace_bytes,
                              int n_slabs, float* dW1, int64_t lddw1, float* dW2, int64_t lddw2);
/* The same launch with both results stored transposed: dW1 [N, lddw1 >= K1] = G^T . X1, dW2 [N, lddw2 >= K2] = G^T . X2.  Called with
 * X1 = g, X2 = A^T g, G = h it yields dWs = h^T . g and dWn = h^T . (A^T g) of the narrowing SAGE layer (sageconv.py:72-82 with the
 * narrow product aggregated) with the wide operand h read once for both.                                                          */
int dgll_hip_grad_weight_bf16_tr(void* stream, const void* X1, int64_t ldx1, int K1, const void* X2, int64_t ldx2, int K2,
                                 const void* G, int64_t ldg, int N, int64_t M, void* workspace, int64_t workspace_bytes,
                                 int n_slabs, float* dW1, int64_t lddw1, float* dW2, int64_t lddw2);

/* ---- the loss at the end of the path: softmax cross-entropy with class-index targets ---------------------------
 * nn.CrossEntropyLoss on the last layer's output (Evaluation/PPI/train_gcn.py:27,45), one pass per direction:
 * row_loss[i] = logsumexp(z_i) - z_i[label_i]  and/or  grad[i, c] = *grad_scale * (softmax(z_i)[c] - [c == label_i]).
 * logits/grad: fp32 or bf16 [n_rows, ld]; labels int64, a label outside [0, n_classes) (e.g. -100) is ignored (loss 0,
 * zero gradient); grad_scale: DEVICE pointer to one float (NULL = 1), so upstream gradients need no host sync.
 * row_loss or grad may be NULL.  n_classes <= 1024.                                                              */
int dgll_hip_softmax_xent(void* stream, const void* logits, int64_t ldz, int dtype, const int64_t* labels,
                          float* row_loss, void* grad, int64_t ldg, const float* grad_scale, int64_t n_rows,
                          int n_classes);
/* The same loss with probability / multi-hot targets (fp32 [n_rows, ldt]) -- what nn.CrossEntropyLoss computes for the
 * float label matrix of the PPI loop (train_gcn.py:27,45): row_loss[i] = -sum_c t_ic log softmax(z_i)[c],
 * grad[i, c] = *grad_scale * (softmax(z_i)[c] * sum_c t_ic - t_ic).                                             */
int dgll_hip_softmax_xent_soft(void* stream, const void* logits, int64_t ldz, int dtype, const float* targets,
                               int64_t ldt, float* row_loss, void* grad, int64_t ldg, const float* grad_scale,
                               int64_t n_rows, int n_classes);
/* Both target kinds behind one entry (exactly one of labels / targets non-NULL) + flags.  bit 0: the logits are ReLU outputs and
 * the gradient returned is d loss / d PRE-activation: zero wherever the logit is <= 0 (the last sageConv's ReLU backward,
 * sageconv.py:83, folded into the loss's gradient pass instead of a separate pass over [n_rows, n_classes]).  bit 1: the padding
 * [n_classes, ldg) of the gradient rows belongs to the output and may be written with zeros (whole 16-byte stores).            */
int dgll_hip_softmax_xent_ex(void* stream, const void* logits, int64_t ldz, int dtype, const int64_t* labels, const float* targets,
                             int64_t ldt, float* row_loss, void* grad, int64_t ldg, const float* grad_scale, int64_t n_rows,
                             int n_classes, int flags);
/* The loss of a (mini-)batch out of its per-row losses in ONE launch: out4 = {sum of row_loss, number of labels in [0, n_classes)
 * (labels NULL: n_rows), their quotient -- nn.CrossEntropyLoss(reduction='mean'), train_gcn.py:27 / buffer_queues.py:106 --, 1 / count (the
 * scale of the gradient pass)}.  One workgroup, fixed summation order; n_rows <= 2^20.                                          */
int dgll_hip_xent_reduce(void* stream, const float* row_loss, const int64_t* labels, int64_t n_rows, int n_classes, float* out4);


/* ---- a10: H = relu(A_csr . (X[:, :actual_F] . W[:actual_F, :])) --------------------------------------------
 * launch_gcn_fused_kernel is the reference's own symbol with its exact signature
 * (/root/reference/dgll/FusedKernel/gcn_fused_kernel.cu:190-195, bound at gcn_extension.cpp:5-10,46-55): int32 CSR,
 * fp32, default stream, synchronous, `num_neighbors` = diff(row_ptr); a libgcn replacement can be relinked against
 * this library unchanged.  dgll_hip_gcn_fused_forward is the same computation with this library's conventions
 * (explicit stream, caller-owned workspace of dgll_hip_gcn_fused_workspace_bytes(), error code).               */
void launch_gcn_fused_kernel(const int* row_ptr, const int* col_idx, const float* values, const float* X, const float* W,
                             float* H, const int* num_neighbors, int N, int F_padded, int actual_F, int H_dim,
                             int total_nnz);
/* The backward twin with the reference's exact signature (gcn_fused_kernel.cu:238-244).  Computes the gradient of the
 * forward above -- G = grad_output * (A.X.W > 0), grad_W[:actual_F] = (A.X)^T.G, grad_X[:, :actual_F] = A^T.(G.W^T) --
 * not the reference kernel's known-wrong arithmetic.  Default stream, synchronous.        */
void launch_gcn_fused_kernel_backward_optimized(const int* row_ptr, const int* col_idx, const float* values,
                                                const float* X, const float* W, const float* grad_output,
                                                float* grad_W, float* grad_X, const int* num_neighbors,
                                                int N, int F_padded, int actual_F, int H_dim, int total_nnz);
int dgll_hip_gcn_fused_forward(void* stream, const int32_t* row_ptr, const int32_t* col_idx, const float* values,
                               const float* X, const float* W, float* H, int N, int F_padded, int actual_F, int H_dim,
                               int total_nnz, void* workspace, size_t workspace_bytes);
size_t dgll_hip_gcn_fused_workspace_bytes(int N, int actual_F, int H_dim);

#ifdef __cplusplus
}
#endif
#endif /* DGLL_HIP_H */
