/*
 * blgnn.h -- C ABI of libblgnn.so: the MI355X (gfx950) bi-level message-passing
 * hot path.  This is the drop-in boundary.
 *
 * The reference (codeKgu/BiLevel-Graph-Neural-Network) is pure Python and has no
 * FFI of its own; what it binds for this path are the calls it makes into
 * torch-geometric 1.1.2 / torch-scatter 1.1.2 / torch-sparse 0.2.4 and stock
 * torch.  Each entry point below names the reference call site(s) it replaces
 * (path:line under /root/reference).  INTEGRATION.md shows the ctypes stub a
 * maintainer of the reference would add.
 *
 * Conventions
 *   - plain pointers and sizes only; every pointer is a DEVICE pointer unless
 *     the name ends in _host.  The caller owns every buffer (outputs are
 *     pre-allocated); the library never allocates, frees or synchronises.
 *   - feature matrices are row-major fp32 with an explicit leading dimension
 *     (ld*, in floats) so that a result can be written straight into a column
 *     slice of a wider buffer (multi-scale concat, GMN f_node input).
 *   - graph structure is CSR with int32 indices: rowptr[n+1], col[nnz]; row =
 *     aggregation target (edge_index[1] of PyG's source_to_target flow), col =
 *     the sources; nnz = rowptr[n] is passed by the caller (it selects the launch
 *     shape: per-row-group kernels for molecular graphs, one workgroup per row for
 *     high-degree interaction graphs).  Graph/segment membership is seg[G+1].
 *   - `stream` is a hipStream_t passed as void*; NULL = the null stream.  All
 *     functions are asynchronous with respect to the host and re-entrant per
 *     stream (no global mutable state except the thread-local error string).
 *   - return value: 0 = BLGNN_OK, negative = error; blgnn_last_error() returns
 *     a human-readable description for the calling thread.
 */
#ifndef BLGNN_H
#define BLGNN_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BLGNN_OK             0
#define BLGNN_ERR_ARG       -1   /* null pointer, bad size, bad alignment, bad enum      */
#define BLGNN_ERR_DEVICE    -2   /* no HIP device / not a gfx950 device                  */
#define BLGNN_ERR_LAUNCH    -3   /* hipLaunchKernel / hipMemsetAsync reported an error   */
#define BLGNN_ERR_UNSUPPORTED -4 /* shape outside what the kernels are written for       */

/* activation ids: model/layers_util.py:60-76 (create_act) */
#define BLGNN_ACT_IDENTITY 0
#define BLGNN_ACT_RELU     1
#define BLGNN_ACT_PRELU    2   /* per-channel slope vector                              */
#define BLGNN_ACT_SIGMOID  3
#define BLGNN_ACT_TANH     4

/* segment reduce kinds: model/layers_aggregation.py:16-19 */
#define BLGNN_REDUCE_SUM  0    /* torch_scatter.scatter_add                              */
#define BLGNN_REDUCE_MEAN 1    /* torch_scatter.scatter_mean (count clamped to >= 1)     */

/* GAT softmax grouping side */
#define BLGNN_GAT_GROUP_SRC 0  /* PyG 1.1.x: softmax(alpha, edge_index[0], N)            */
#define BLGNN_GAT_GROUP_DST 1  /* textbook / PyG >= 1.2                                  */

/* ---- library ---------------------------------------------------------- */
int         blgnn_version(void);
const char *blgnn_last_error(void);
/* number of visible gfx950 devices; 0 on a box without one (never negative) */
int         blgnn_device_count(void);

/* HIP stream restricted to the compute units whose bits are set in mask_words_host (num_words x 32
 * bits; on MI355X bit i = CU i / 8 of XCD i % 8), on the calling thread's current device.  The
 * reference runs everything on one stream (src/train.py:75-108); the engine runs the molecular sweep
 * and the interaction-level pass of consecutive forwards side by side on disjoint CUs.  The caller
 * owns the stream and releases it with blgnn_stream_destroy after synchronising it. */
int blgnn_stream_create_cu_mask(const uint32_t *mask_words_host, int32_t num_words, void **stream_out);
int blgnn_stream_destroy(void *stream);

/* ---- graph layout ------------------------------------------------------ */

/* COO -> CSR.  Replaces the per-forward index plumbing of PyG's
 * MessagePassing.propagate (call sites model/layers.py:51-54) and the
 * per-forward rebuild in model/layers_load_interaction_graph.py:11-20.
 *   key[e] = aggregation target of edge e, nbr[e] = its source (int64, the
 *   reference's edge_index rows); n = number of nodes.
 *   rowptr[n+1], col[nnz]: outputs.  Neighbours of a row come out in
 *   ascending order (deterministic summation order) for any input order.
 *   work: int32[n+1] scratch.  */
int blgnn_csr_from_coo(const int64_t *key, const int64_t *nbr, int64_t nnz,
                       int32_t n, int32_t *rowptr, int32_t *col,
                       int32_t *work, void *stream);

/* sorted graph-id-per-node vector (src/merged_graph.py:56-57,82 `batch`) ->
 * segment offsets seg[G+1]. */
int blgnn_segments_from_batch(const int64_t *batch, int64_t n, int32_t G,
                              int32_t *seg, void *stream);

/* ---- sparse aggregation (gather -> wavefront segment-reduce) ----------- */

/* y[i] = self_coef * x[i] + sum_{j in row i, (j != i if skip_self)} w_e * x[j]
 * GINConv aggregation: model/layers.py:26-31,54 (self_coef = 1+eps, skip_self=1,
 * w = NULL).  Also torch_scatter.scatter_add over gathered rows
 * (model/layers_gmn.py:42) with self_coef = 0.  x and y must not alias. */
int blgnn_spmm_sum(const int32_t *rowptr, const int32_t *col, int64_t nnz, const float *w,
                   const float *x, int64_t ldx, float *y, int64_t ldy,
                   int32_t n, int32_t d, float self_coef, int32_t skip_self,
                   void *stream);

/* The same aggregation with the result in the sp format of blgnn_linear_sp (below) and no fp32 copy: the GIN layer
 * as aggregation + two pre-split GEMMs (model/layers.py:9-63 at the widths the single-launch f16x3 layer does not
 * cover: d = 512).  d % 256 == 0, d <= 2048, 16-byte aligned x rows. */
int blgnn_spmm_sum_sp(const int32_t *rowptr, const int32_t *col, int64_t nnz, const float *x, int64_t ldx,
                      int32_t n, int32_t d, float self_coef, int32_t skip_self, void *y_sp, int64_t ld_ysp,
                      uint8_t *y_spe, int64_t ld_yspe, void *stream);

/* One whole GIN NodeEmbedding layer in ONE launch: model/layers.py:26-31,51-57.
 *   y = affine( act( act( agg @ w1 + b1 ) @ w2 + b2 ) ),
 *   agg_i = (1 + eps) x_i + sum_{j in row i, j != i} x_j
 * x [n,K] -> y [n,N]; w1 [K,N], w2 [N,N] row-major ([in,out]: pass nn.Linear weights
 * transposed); `act` is applied after BOTH linears (the reference reuses one act module,
 * layers.py:26,29,55); affine = eval BatchNorm (bn_scale/bn_shift nullable).
 * The aggregated tile and the hidden activation live in LDS only.  Shapes outside
 * N in {64,128,256}, K % 4 == 0, K <= 256 return BLGNN_ERR_UNSUPPORTED (use
 * blgnn_spmm_sum + 2 x blgnn_linear, which compute the same thing). */
int blgnn_gin_layer(const int32_t *rowptr, const int32_t *col, int64_t nnz,
                    const float *x, int64_t ldx, int32_t n, int32_t K,
                    const float *w1, const float *b1, const float *w2,
                    const float *b2, int32_t N, float eps, int32_t act,
                    const float *act_param, const float *bn_scale,
                    const float *bn_shift, float *y, int64_t ldy, void *stream);

/* Same layer with the two dense transforms on the bf16 matrix cores at fp32 accuracy
 * ("bf16x6": operands split into three bf16 pieces, six MFMAs per product term, fp32
 * accumulation; error vs fp64 at or below a sequential fp32 fma chain -- DESIGN.md section 4).
 * w1_packed / w2_packed come from blgnn_pack_weight_x6 (call once per weight version):
 * [3 pieces][Kp/8][N][8] bf16 with Kp = K rounded up to 32, blgnn_packed_weight_x6_elems(K, N)
 * bf16 elements.  N in {128,256}, K % 4 == 0, K <= 256, else BLGNN_ERR_UNSUPPORTED. */
int64_t blgnn_packed_weight_x6_elems(int32_t K, int32_t N);
int blgnn_pack_weight_x6(const float *w, int64_t ldw, int32_t K, int32_t N, void *out,
                         void *stream);
int blgnn_gin_layer_x6(const int32_t *rowptr, const int32_t *col, int64_t nnz,
                       const float *x, int64_t ldx, int32_t n, int32_t K,
                       const void *w1_packed, const float *b1, const void *w2_packed,
                       const float *b2, int32_t N, float eps, int32_t act,
                       const float *act_param, const float *bn_scale,
                       const float *bn_shift, float *y, int64_t ldy, void *stream);

/* Same layer on the fp16 matrix cores at fp32 accuracy ("f16x3": operands split in TWO fp16
 * pieces, three MFMAs per product; activation rows and weight matrices range-normalised by
 * exact powers of two).  Half the matrix-core work and two thirds of the operand bytes of the
 * bf16x6 variant; 64-row tiles.  w*_packed = blgnn_pack_weight_h3 images
 * (blgnn_packed_weight_h3_elems fp16 elements each, 16-byte aligned: 16-byte header holding
 * 1/scale, then [2 pieces][Kp/8][N][8]).  N in {128, 256}, K % 4 == 0, K <= N, else
 * BLGNN_ERR_UNSUPPORTED (use blgnn_gin_layer_x6).
 * Fused node -> graph readout (model/layers_aggregation.py:16-19,32-41; north_star "with the node
 * readout fused in"): with pooled != NULL every tile also adds its rows' column sums into
 *   pooled[row_graph[i], 0:N] += y[i, :] * (pool_kind == MEAN ? 1 / max(|graph|, 1) : 1)
 * (row_graph = the sorted batch vector as int32[n], seg = graph row ranges int32[G+1], pooled
 * ZERO-INITIALISED by the caller, typically a column slice of the multi-scale readout buffer).
 * One hardware float atomic per (graph, column, tile): a graph covered by <= 2 tiles sums in an
 * order-independent way; larger ones may differ in the last bit between runs.  pooled == NULL: off.
 * y_row_amax (optional, [n]): receives max |y[i, :]| of every row -- the row magnitudes a following
 * blgnn_linear_h3 on y normalises by (the GMN message GEMM), without a pass over y.
 * y_sp / y_spe (optional, N == 256 only; blgnn_gin_layer_h3): y additionally in the sp format of
 * blgnn_linear_sp (below), ld_ysp in fp16 elements, one inverse block scale per row at y_spe[i * ld_yspe]:
 * the GMN propagator's GEMMs read the layer output from there with no conversion pass. */
int64_t blgnn_packed_weight_h3_elems(int32_t K, int32_t N);
int blgnn_pack_weight_h3(const float *w, int64_t ldw, int32_t K, int32_t N, void *out,
                         void *stream);
int blgnn_gin_layer_h3(const int32_t *rowptr, const int32_t *col, int64_t nnz,
                       const float *x, int64_t ldx, int32_t n, int32_t K,
                       const void *w1_packed, const float *b1, const void *w2_packed,
                       const float *b2, int32_t N, float eps, int32_t act,
                       const float *act_param, const float *bn_scale,
                       const float *bn_shift, float *y, int64_t ldy, const int32_t *row_graph,
                       const int32_t *seg, float *pooled, int64_t ldp, int32_t pool_kind,
                       float *y_row_amax, void *y_sp, int64_t ld_ysp, uint8_t *y_spe, int64_t ld_yspe, void *stream);

/* blgnn_gin_layer_h3 for the FIRST lower layer with categorical input: instead of the one-hot rows
 * x[n, F_in] (utils/data/representation_node_feat.py:86-97: one block per attribute, one 1 per block)
 * the kernel reads 8 bytes per atom -- codes[n][8], byte a = the value index of attribute a -- and
 * expands them on the fly (6 bytes instead of 52 floats per atom on DrugBank).  code_offsets_host is a
 * HOST array of num_attr + 1 ints: first column of every attribute's block, last = F_in (multiple of
 * 4, <= 64).  Result identical to blgnn_gin_layer_h3 on the expanded rows. */
int blgnn_gin_layer_h3_codes(const int32_t *rowptr, const int32_t *col, int64_t nnz,
                             const void *codes, const int32_t *code_offsets_host,
                             int32_t num_attr, int32_t n, const void *w1_packed,
                             const float *b1, const void *w2_packed, const float *b2,
                             int32_t N, float eps, int32_t act, const float *act_param,
                             const float *bn_scale, const float *bn_shift, float *y,
                             int64_t ldy, const int32_t *row_graph, const int32_t *seg,
                             float *pooled, int64_t ldp, int32_t pool_kind, float *y_row_amax,
                             void *stream);

/* The whole lower level in ONE launch: num_layers (<= 8) stacked GIN NodeEmbedding layers
 * (model/layers.py:26-31,51-57) + the node -> graph readout of every layer that is pooled
 * (model/layers_aggregation.py:27-42, multi-scale concat = one column slice per layer).
 * Works on tiles made of WHOLE molecules (<= 64 rows each): tile_row[T+1] / tile_mol[T+1]
 * give the first row / first molecule of every tile, seg[G+1] the molecule row offsets; a
 * molecule's neighbours are then inside its tile at every depth and the layers run out of
 * LDS back to back.  Layer l: weights w1 [K_l,N], w2 [N,N] (K_0 = K0, K_l = N), biases,
 * act id (applied after both linears), eps, eval-BN affine (scale/shift nullable), PReLU
 * slopes (nullable), pool_col = column offset of its pooled rows in `pooled` or -1.
 * All *_host arguments are HOST arrays indexed by layer (device pointers / values).
 * pooled[row_map ? row_map[g] : g][pool_col + 0..N) = mean/sum over molecule g;
 * y_last (nullable) receives the last layer's activations [rows, N].
 * N in {64,128,256}, K0 % 4 == 0, K0 <= 256, else BLGNN_ERR_UNSUPPORTED. */
int blgnn_gin_stack(const int32_t *rowptr, const int32_t *col, int64_t nnz,
                    const float *x0, int64_t ldx0, int32_t K0, int32_t N,
                    int32_t num_tiles, const int32_t *tile_row,
                    const int32_t *tile_mol, const int32_t *seg,
                    int32_t num_layers, const float *const *w1_host,
                    const float *const *b1_host, const float *const *w2_host,
                    const float *const *b2_host, const float *const *slope_host,
                    const float *const *scale_host, const float *const *shift_host,
                    const int32_t *act_host, const float *eps_host,
                    const int32_t *pool_col_host, int32_t reduce_kind, float *pooled,
                    int64_t ldp, const int32_t *row_map, float *y_last, int64_t ldy,
                    void *stream);

/* blgnn_gin_stack with both dense transforms of every layer on the fp16 matrix cores at fp32 accuracy
 * (f16x3, see blgnn_gin_layer_h3): w1_packed_host / w2_packed_host are HOST arrays of num_layers device
 * pointers to blgnn_pack_weight_h3 images.  N == 128, K0 <= 128; 64-row whole-molecule tiles, two
 * workgroups per CU; the per-layer readout (model/layers_aggregation.py:32-41) is part of the launch. */
int blgnn_gin_stack_h3(const int32_t *rowptr, const int32_t *col, int64_t nnz, const float *x0,
                       int64_t ldx0, int32_t K0, int32_t N, int32_t num_tiles,
                       const int32_t *tile_row, const int32_t *tile_mol, const int32_t *seg,
                       int32_t num_layers, const void *const *w1_packed_host,
                       const float *const *b1_host, const void *const *w2_packed_host,
                       const float *const *b2_host, const float *const *slope_host,
                       const float *const *scale_host, const float *const *shift_host,
                       const int32_t *act_host, const float *eps_host,
                       const int32_t *pool_col_host, int32_t reduce_kind, float *pooled,
                       int64_t ldp, const int32_t *row_map, float *y_last, int64_t ldy,
                       void *stream);

/* GCNConv: model/layers.py:23,52 (formula documented at
 * model/layers_meta.py:128-148).  Step 1: dis[i] = (1 + sum_{j!=i} w_ij)^-1/2
 * (inf -> 0).  */
int blgnn_gcn_deg_inv_sqrt(const int32_t *rowptr, const int32_t *col,
                           const float *w, int32_t n, float *dis, void *stream);
/* Step 2: y[i] = epi( dis_i^2 h_i + sum_{j!=i} dis_i w_ij dis_j h_j + bias [+ y[i]] ),
 * epi = activation then optional per-channel affine (eval BatchNorm:
 * model/layers.py:55-57).  bias/act_param/bn_scale/bn_shift may be NULL.
 * accumulate != 0 adds the previous contents of y before the activation: the sum over edge types
 * of NodeModelAggrByEdge('gcn') (model/layers_meta.py:68,74-79) without a temporary per type. */
int blgnn_gcn_aggregate(const int32_t *rowptr, const int32_t *col, int64_t nnz,
                        const float *w, const float *dis, const float *h,
                        int64_t ldh, float *y, int64_t ldy, int32_t n, int32_t d,
                        const float *bias, int32_t act, const float *act_param,
                        const float *bn_scale, const float *bn_shift,
                        int32_t accumulate, void *stream);

/* GATConv heads=1: model/layers.py:33,54; model/layers_meta.py:70,78.
 * Step 1: s_dst[i] = h_i . att[0:d], s_src[i] = h_i . att[d:2d]. */
int blgnn_gat_scores(const float *h, int64_t ldh, const float *att, int32_t n,
                     int32_t d, float *s_dst, float *s_src, void *stream);
/* Steps 0+1 in one launch for the interaction level (a few thousand rows, N <= 128):
 * h = x[M,K] @ w[K,N] (exact-f32 MFMA, split-K over 16 wavefronts) and both attention halves
 * of every row.  N > 128 returns BLGNN_ERR_UNSUPPORTED (use blgnn_linear + blgnn_gat_scores). */
int blgnn_gat_transform(const float *x, int64_t ldx, const float *w, int64_t ldw,
                        const float *att, int32_t M, int32_t N, int32_t K, float *h,
                        int64_t ldh, float *s_dst, float *s_src, void *stream);
/* Step 2: per softmax group g: mx[g] = max_e a_e, den[g] = sum_e exp(a_e - mx[g]),
 * a_e = leaky_relu(s_dst[target] + s_src[source], slope) over the edges of the
 * group plus the appended self loop; existing self loops are dropped.
 * (rowptr,col) must be the CSR keyed by the grouping side; for the symmetric
 * graphs of this path (model/layers_util.py:133) it is the same CSR either way.
 * stats_packed (optional, [n][4] floats, 16-byte aligned): additionally (s_src[g], mx[g], den[g], 0) per node in one
 * 16-byte word -- what blgnn_gat_aggregate gathers per edge under SRC grouping (one sector per edge instead of three). */
int blgnn_gat_softmax_stats(const int32_t *rowptr, const int32_t *col, int64_t nnz,
                            const float *s_dst, const float *s_src, int32_t n,
                            int32_t group, float slope, float *mx, float *den,
                            float *stats_packed, void *stream);
/* Step 3: y[i] (+)= epi( sum_{j in N(i) u {i}} alpha_ij h_j + bias ),
 * alpha_ij = exp(a_ij - mx[g]) / (den[g] + 1e-16), g = j (SRC) or i (DST).
 * accumulate != 0 adds into y (NodeModelAggrByEdge: model/layers_meta.py:74-79);
 * the epilogue (act, then affine) is applied to the value being written.
 * rows / num_rows (optional): compute only the target rows rows[0 .. num_rows) (int64 device array, duplicates
 * allowed -- computed again, same values --, negative entries skipped: the padding of a de-duplicated list) and leave
 * every other row of y untouched -- the LAST interaction-level layer feeds only the rows the pair scorer reads
 * (model/layers_link_pred.py:56-61), and a rank of a target-row-sharded job computes its own rows of the others.
 * Honoured by the high-degree kernels (one workgroup per row; one wavefront per row from 16 384 rows on, from 8 192
 * listed rows of a graph that large); low-degree graphs compute every row.  stats_packed (optional): the packed statistics blgnn_gat_softmax_stats wrote (same values). */
int blgnn_gat_aggregate(const int32_t *rowptr, const int32_t *col, int64_t nnz,
                        const float *h, int64_t ldh, const float *s_dst,
                        const float *s_src, const float *mx, const float *den,
                        int32_t group, float slope, float *y, int64_t ldy,
                        int32_t n, int32_t d, const float *bias, int32_t act,
                        const float *act_param, const float *bn_scale,
                        const float *bn_shift, int32_t accumulate, const int64_t *rows, int32_t num_rows,
                        const float *stats_packed, void *stream);

/* GMN message: model/layers_gmn.py:38-42 restructured:
 * y[i] = sum_{e: key_e = i} relu(p[i] + q[nbr_e])  (self loops kept).
 * row_amax (optional, [n], initialised by the caller): row_amax[i] = max(row_amax[i], max_c y[i, c]) by atomic
 * maximum -- one of the three producers of the f_node input's row magnitudes (blgnn_linear_h3), so that no
 * pass over the [n, 4D] buffer is needed.
 * y_sp / y_spe (optional, d % 256 == 0): the result in the sp format of blgnn_linear_sp (f_node's first Linear
 * reads it); y may then be NULL. */
int blgnn_edge_relu_sum(const int32_t *rowptr, const int32_t *col, int64_t nnz,
                        const float *p, int64_t ldp, const float *q, int64_t ldq,
                        float *y, int64_t ldy, int32_t n, int32_t d, float *row_amax,
                        void *y_sp, int64_t ld_ysp, uint8_t *y_spe, int64_t ld_yspe, void *stream);
/* row degree as float (for the deg_i * b2 term of the restructured message) */
int blgnn_row_degree(const int32_t *rowptr, int32_t n, float *deg, void *stream);

/* ---- readout ------------------------------------------------------------ */

/* out[row_map ? row_map[g] : g][0:d] = reduce_{i in seg[g]..seg[g+1]} x[i]
 * torch_scatter.scatter_mean / scatter_add over the sorted `batch` vector:
 * model/layers_aggregation.py:16-19,32-41; with row_map it also performs the
 * per-gid init_x row writes of model/layers_aggregation.py:68-74. */
int blgnn_segment_reduce(const int32_t *seg, int32_t G, const float *x,
                         int64_t ldx, int32_t d, int32_t kind, float *out,
                         int64_t ldo, const int32_t *row_map, void *stream);

/* Multi-scale readout in ONE launch: model/layers_aggregation.py:32-39
 * (torch.cat([agg(x, batch) for x in acts], dim=1)).  xs_host / ldxs_host / ds_host are HOST
 * arrays of num_inputs (<= 8) device pointers, leading dimensions and widths; input i is
 * reduced into columns [sum_{k<i} d_k, +d_i) of out. */
int blgnn_segment_reduce_multi(const int32_t *seg, int32_t G, int32_t num_inputs,
                               const float *const *xs_host, const int64_t *ldxs_host,
                               const int32_t *ds_host, int32_t kind, float *out,
                               int64_t ldo, const int32_t *row_map, void *stream);

/* ---- dense feature transform (fp32 MFMA) -------------------------------- */

/* Y = epi(X[M,K] @ W[K,N] + bias) (+ addend),  epi = act then per-channel affine.
 * Replaces torch.mm / nn.Linear inside GINConv's nn (model/layers.py:27-30),
 * GATConv/GCNConv's x @ weight, MLP (model/layers_util.py:44-57) and the
 * post-conv act + eval BatchNorm (model/layers.py:55-57).
 * W is [K,N] row-major (PyG layout); pass nn.Linear weights transposed.
 * rowvec [M] and rowvec_w [N] (both or neither) add the rank-1 term
 * rowvec[m] * rowvec_w[n] before the epilogue (the deg_i * b2 term of the
 * restructured GMN message). */
int blgnn_linear(const float *x, int64_t ldx, const float *w, int64_t ldw,
                 float *y, int64_t ldy, int32_t M, int32_t N, int32_t K,
                 const float *bias, const float *rowvec, const float *rowvec_w,
                 int32_t act, const float *act_param, const float *bn_scale,
                 const float *bn_shift, void *stream);

/* blgnn_linear with the product on the bf16 matrix cores at fp32 accuracy (bf16x6: operands split
 * in three bf16 pieces, six MFMAs per product, see blgnn_gin_layer_x6).  Same call sites as
 * blgnn_linear; w_packed = blgnn_pack_weight_x6(W[K,N]).  Needs K % 4 == 0, ldx % 4 == 0 and
 * 16-byte aligned x / w_packed, otherwise BLGNN_ERR_UNSUPPORTED (use blgnn_linear). */
int blgnn_linear_x6(const float *x, int64_t ldx, const void *w_packed, float *y,
                    int64_t ldy, int32_t M, int32_t N, int32_t K, const float *bias,
                    const float *rowvec, const float *rowvec_w, int32_t act,
                    const float *act_param, const float *bn_scale,
                    const float *bn_shift, void *stream);

/* W[K,N] fp32 -> the packed image blgnn_linear_h3 reads: a 16-byte header (inverse of the power-of-two
 * scale that moves max |w| into [2^14, 2^15)) followed by ceil(N / 128) x ceil(K / 64) tile images of
 * 32 KB, each laid out exactly as the kernel keeps a (column block, K step) in LDS, so a tile comes in
 * by 32 linear LDS-DMA pieces.  `out` holds blgnn_packed_weight_h3g_elems(K, N) fp16 elements. */
int64_t blgnn_packed_weight_h3g_elems(int32_t K, int32_t N);
int blgnn_pack_weight_h3g(const float *w, int64_t ldw, int32_t K, int32_t N, void *out, void *stream);

/* out[m] = max_k |x[m, k]|: the row magnitudes blgnn_linear_h3 normalises its X operand by.  A producer
 * that already knows them (it wrote the rows) may supply its own array instead. */
int blgnn_row_absmax(const float *x, int64_t ldx, int32_t M, int32_t K, float *out, void *stream);

/* blgnn_linear with the product on the fp16 matrix cores at fp32 accuracy (f16x3: operands range-
 * normalised by exact powers of two -- X per row from row_amax[M], W as a whole when packed -- and
 * split in two fp16 pieces, three MFMAs per product, see blgnn_gin_layer_h3).  Same call sites as
 * blgnn_linear (the GMN propagator's f_message / f_node MLPs, model/layers_gmn.py:19-26,39-47, are
 * the large ones); w_packed = blgnn_pack_weight_h3g(W[K,N]); row_amax[m] >= max_k |x[m, k]| (any
 * value within a factor 2 above the true maximum gives the same result).  y_row_amax (optional, [M],
 * ZERO-INITIALISED by the caller) receives max_n |y[m, n]| through atomic maxima from the column
 * tiles -- the row_amax of a following blgnn_linear_h3 on y, for free.  Needs K % 4 == 0,
 * ldx % 4 == 0 and 16-byte aligned x / w_packed, otherwise BLGNN_ERR_UNSUPPORTED (use blgnn_linear). */
int blgnn_linear_h3(const float *x, int64_t ldx, const float *row_amax, const void *w_packed,
                    float *y, int64_t ldy, float *y_row_amax, int32_t M, int32_t N, int32_t K,
                    const float *bias,
                    const float *rowvec, const float *rowvec_w, int32_t act,
                    const float *act_param, const float *bn_scale, const float *bn_shift,
                    void *stream);

/* ---- f16x3 GEMM on pre-split activations ("sp" format) -------------------- */

/* The GEMMs of the GMN propagator over all atoms (model/layers_gmn.py:19-26,39-47: f_message's first
 * Linear, f_node's two Linears; model/layers_util.py:44-57) with the fp32 -> two-fp16-piece conversion of
 * the activations done ONCE by the kernel that produces them instead of inside every GEMM that reads them.
 *
 * sp format of an activation matrix X[M, K], K % 256 == 0 (csrc/sp_common.hpp):
 *   xs: fp16, row m at xs + m * ldxs (ldxs >= 2 K elements, rows 16-byte aligned); per 32-column chunk c of a
 *       row 64 values: the 32 high pieces h1 of columns 32c..32c+31, then their 32 low pieces h2;
 *   xe: uint8 [M][K / 256] (row stride ldxe bytes): the block's power-of-two scale as the biased fp32 exponent e
 *       of its inverse, x[m, k] = (h1 + h2) * 2^(e - 127); 2^(127 - e) moves the block's largest magnitude
 *       into [2^14, 2^15).  A GEMM fetches a row's (<= 8) exponents with one 8-byte load: the X operand of
 *       blgnn_linear_sp needs ldxe % 8 == 0 and an 8-byte aligned xe (allocate 8 bytes per row).
 * A column range that starts and ends on multiples of 256 is again an sp matrix (xs + 2 * c0, xe + c0 / 256):
 * producers write their slices of the f_node input [x | S | u] independently.
 *
 * blgnn_split_rows_sp / blgnn_join_rows_sp convert fp32 rows to / from the format (one streaming pass).
 * blgnn_pack_weight_sp: W[K, N] fp32 (N % 256 == 0) -> the tile-ordered two-piece image the GEMM streams.
 * blgnn_linear_sp: epi(X @ W + bias + rowvec (x) rowvec_w) as three fp16 MFMA products per fp32 product
 * (fp32-equivalent, see blgnn_linear_h3), 256 x 256 tiles, both operands by LDS-DMA.  The result goes to
 * y (fp32 rows, ldy % 4 == 0), to ys / ye (sp format) or to both; pass NULL for the one not wanted.
 * Needs N % 256 == 0 and K % 256 == 0 (K <= 2048); other shapes: BLGNN_ERR_UNSUPPORTED (use blgnn_linear_h3).
 * addend (optional, fp32 rows of >= N columns, < 2 GiB): epi(X @ W + bias + rowvec (x) rowvec_w + addend[r(m), :]) with
 * r(m) = addend_rows[m] (int32 [M], rows of addend: num_addend_rows) or m -- the K-split of a GEMM whose leading
 * input columns repeat across rows: f_node's first Linear on [x | S | u] in the first propagator of a pair batch,
 * where [x | S] @ W[:3D] is computed once per distinct molecule (blgnn_expand_rows) and u @ W[3D:] for every atom.
 * pooled (optional, fp32 [G, >= N], zero-initialised by the caller): the node -> graph readout of the RESULT
 * (model/layers_aggregation.py:16-19,32-41) in the epilogue, as in blgnn_gin_layer_h3 -- pooled[row_graph[m], :] +=
 * result[m, :] (BLGNN_REDUCE_SUM) or / rows of that graph (BLGNN_REDUCE_MEAN); row_graph int32 [M] sorted, seg [G + 1].
 * With pooled, y and ys may both be NULL (the rows themselves are then never written: the last propagator's hidden
 * layer, whose second Linear commutes with the mean).  relu / identity only, no addend; fp32 atomics (low bits vary). */
int64_t blgnn_packed_weight_sp_elems(int32_t K, int32_t N);
int blgnn_pack_weight_sp(const float *w, int64_t ldw, int32_t K, int32_t N, void *out, void *stream);
int blgnn_split_rows_sp(const float *x, int64_t ldx, int32_t M, int32_t K, void *xs, int64_t ldxs,
                        uint8_t *xe, int64_t ldxe, void *stream);
int blgnn_join_rows_sp(const void *xs, int64_t ldxs, const uint8_t *xe, int64_t ldxe, int32_t M, int32_t K,
                       float *x, int64_t ldx, void *stream);
int blgnn_linear_sp(const void *xs, int64_t ldxs, const uint8_t *xe, int64_t ldxe, const void *w_packed,
                    float *y, int64_t ldy, void *ys, int64_t ldys, uint8_t *ye, int64_t ldye,
                    int32_t M, int32_t N, int32_t K, const float *bias, const float *rowvec,
                    const float *rowvec_w, int32_t act, const float *act_param,
                    const float *bn_scale, const float *bn_shift, const float *addend, int64_t ld_addend,
                    const int32_t *addend_rows, int32_t num_addend_rows, const int32_t *row_graph,
                    const int32_t *seg, float *pooled, int64_t ldp, int32_t pool_kind, void *stream);

/* ---- GMN cross-graph attention ------------------------------------------ */

/* model/layers_gmn.py:55-84.  Graphs (2k, 2k+1) of seg[] form pair k.
 * u[i] = x[i] - sum_j softmax_j(cos(x_i, x_j)) x_j, j over the partner graph.
 * seg has 2*num_pairs+1 entries.
 * row_amax (optional, [rows], initialised by the caller): merged with max_c |u[i, c]| by atomic maximum
 * (see blgnn_edge_relu_sum).
 * max_nodes: the largest graph of the batch when the caller knows it (the reference's ind_list is a host list),
 * 0 = unknown.  Pairs are handled by three kernels by size class (<= 64, <= 128, larger); with the hint the classes
 * nobody is in are not launched.  classes: bit mask of the size classes to run (1: both graphs <= 64 nodes, 2: a graph
 * of 65..128, 4: larger), 0 = all -- they are independent, so a caller may issue them on different streams.
 * pair_order (device int32, optional) / class_counts_host (host int32[4]): the pair indices grouped by the size of their
 * larger graph (<= 32, <= 64, <= 128, larger; pairs of two empty graphs left out), as the host can derive from ind_list:
 * the one-wavefront-per-pair kernels of the first two classes then run exactly one wavefront per listed pair instead
 * of scanning all pairs. */
int blgnn_gmn_match(const int32_t *seg, int32_t num_pairs, const float *x,
                    int64_t ldx, float *u, int64_t ldu, int32_t d, float *row_amax,
                    int32_t max_nodes, int32_t classes, const int32_t *pair_order,
                    const int32_t *class_counts_host, void *stream);

/* The same attention for the pairs whose two graphs both have <= 64 nodes, on x in the sp format (the planes the
 * producing layer wrote; d = 256: one scale block per row) with the result written straight into the sp format that
 * f_node's first Linear reads (model/layers_gmn.py:45-50: in_f_node = cat([x, m, u])) -- no fp32 u, no separate split
 * pass.  One wavefront per pair, both directions from the same staged rows, three fp16 MFMA products per fp32 product
 * (csrc/gmn_match_sp.hip).  pair_order / class_counts_host as in blgnn_gmn_match (required): classes 0 and 1 are
 * processed, the pairs of classes 2 and 3 are the caller's to run through blgnn_gmn_match(classes = 6) +
 * blgnn_split_rows_sp_pairs.  row_map (optional, device int32 [merged atoms]): x plane row of every merged atom (the
 * shared prefix of a pair batch keeps x on the distinct molecules: blgnn_expand_rows' map); x_rows = rows of the x
 * planes.  n_le48: how many of the class-1 pairs, listed first, have both graphs <= 48 nodes (0 = unknown): they run with
 * 48-row chunk images, four instead of three wavefronts per CU.  The result row's scale is a bound (2 * max(own row's
 * scale, largest partner scale)), not its maximum. */
int blgnn_gmn_match_sp(const int32_t *seg, int32_t num_pairs, const void *xs, int64_t ldxs, const uint8_t *xe,
                       int64_t ldxe, const int32_t *row_map, int64_t x_rows, void *us, int64_t ldus,
                       uint8_t *ue, int64_t ldue, int32_t d, const int32_t *pair_order,
                       const int32_t *class_counts_host, int32_t n_le48, void *stream);

/* blgnn_split_rows_sp for the rows of the listed pairs only (pair k = rows seg[2k] .. seg[2k + 2]): the pairs the
 * fp32 attention kernels took when the others went through blgnn_gmn_match_sp. */
int blgnn_split_rows_sp_pairs(const float *x, int64_t ldx, const int32_t *seg, const int32_t *pair_list,
                              int32_t n_list, int32_t K, void *xs, int64_t ldxs, uint8_t *xe, int64_t ldxe,
                              void *stream);

/* GMNPropagator(f_node='GRU'): the gate arithmetic of torch.nn.GRUCell (model/layers_gmn.py:27-30,48-50) on the two
 * gate tables gi = in W_ih^T + b_ih, gh = h W_hh^T + b_hh ([n, 3D], gate order r | z | n; two blgnn_linear calls):
 *   out = (1 - z) * tanh(gi_n + r * gh_n) + z * h,  r = sigmoid(gi_r + gh_r),  z = sigmoid(gi_z + gh_z). */
int blgnn_gru_gates(const float *gi, int64_t ldgi, const float *gh, int64_t ldgh, const float *h,
                    int64_t ldh, int32_t n, int32_t D, float *out, int64_t ldo, void *stream);

/* Backward of blgnn_gru_gates (training GMNPropagator(f_node='GRU'), model/layers_gmn.py:27-30,48-50 under
 * src/train.py:177-182): from the forward's gate tables gi, gh, the state h and g = d loss / d h' (all [n, .]) ->
 * dgi, dgh [n, 3D] (gate order r | z | n) and the state's direct share dh = g z [n, D]; the gates are recomputed.
 * The four products around it (d in = dgi W_ih, dW_ih = dgi^T in, dh += dgh W_hh, dW_hh = dgh^T h) and the bias sums
 * are blgnn_linear / blgnn_grad_weight / blgnn_col_reduce. */
int blgnn_gru_gates_backward(const float *gi, int64_t ldgi, const float *gh, int64_t ldgh, const float *h,
                             int64_t ldh, const float *g, int64_t ldg, int32_t n, int32_t D, float *dgi,
                             int64_t lddgi, float *dgh, int64_t lddgh, float *dh, int64_t lddh, void *stream);

/* out = a * b elementwise: the gate of GMNAggregatorPairs (model/layers_aggregation.py:88-93: sigmoid(gate) * weight,
 * the sigmoid rides on the producing blgnn_linear's epilogue). */
int blgnn_hadamard(const float *a, int64_t lda, const float *b, int64_t ldb, int32_t n, int32_t d,
                   float *out, int64_t ldo, void *stream);

/* ---- pair scorer front end ----------------------------------------------- */

/* out[b] = [ z[ids1[b]] / max(|z[ids1[b]]|, eps) , z[ids2[b]] / max(...) ]
 * model/layers_link_pred.py:44,56-61 (normalize + row gather + cat). */
int blgnn_pair_gather_normalize(const float *z, int64_t ldz, const int64_t *ids1,
                                const int64_t *ids2, int32_t B, int32_t d,
                                float eps, float *out, int64_t ldo, void *stream);

/* The whole mlp_concat pair scorer in one launch: model/layers_link_pred.py:43-67.
 *   out[b] = final_act( MLP( [normalize(z[ids1[b]]), normalize(z[ids2[b]])] ) )
 * MLP = num_layers (<= 4) Linear layers, hidden layers followed by hidden_act
 * (model/layers_util.py:44-57, bn=False as LinkPred builds it).  w_host / b_host are HOST
 * arrays of device pointers (w[l] is [dims[l], dims[l+1]] row-major, i.e. nn.Linear weights
 * transposed; b[l] nullable); dims_host = {2*d, h1, ..., out}.  Widths above 2048 return
 * BLGNN_ERR_UNSUPPORTED (use blgnn_pair_gather_normalize + blgnn_linear). */
int blgnn_link_pred_head(const float *z, int64_t ldz, const int64_t *ids1,
                         const int64_t *ids2, int32_t B, int32_t d, float eps,
                         int32_t num_layers, const float *const *w_host,
                         const float *const *b_host, const int32_t *dims_host,
                         int32_t hidden_act, int32_t final_act, float *out,
                         int64_t ldo, void *stream);

/* y[i] = x[i] / max(|x[i]|_2, eps): model/layers.py:60-61. */
int blgnn_l2_normalize_rows(const float *x, int64_t ldx, float *y, int64_t ldy,
                            int32_t n, int32_t d, float eps, void *stream);

/* dst[row_map[g]] = src[g] (model/layers_aggregation.py:72-74 when the readout
 * is not a plain segment reduce). */
int blgnn_scatter_rows(const float *src, int64_t lds, const int32_t *row_map,
                       float *dst, int64_t ldd, int32_t G, int32_t d, void *stream);

/* dst row i = src row row_map[i] for i < num_rows; rows are opaque bytes (row_bytes each; leading dimensions in
 * BYTES).  The replication step of a pair batch: src/batch.py:105-144 with batch_unique_graphs = False
 * (src/config.py:258) appends a molecule's tensors once per pair it occurs in, so every layer in front of the first
 * cross-graph attention computes identical rows ~(2 * pairs / distinct drugs) times; here those layers run on the
 * distinct molecules and their results (fp32 rows, sp piece planes, sp scale bytes) are replicated by this call.
 * 16-byte units when row_bytes, both leading dimensions and both pointers are multiples of 16, bytes otherwise.
 * row_map values must lie in [0, rows of src): they are not checked on the device. */
int blgnn_expand_rows(const void *src, int64_t ld_src, const int32_t *row_map, int64_t num_rows,
                      void *dst, int64_t ld_dst, int64_t row_bytes, void *stream);

/* ---- backward pass / train mode (src/train.py:177-182) -------------------- */
/* The large products of the backward pass are the forward entry points on other operands:
 *   dX = dY @ W        -> blgnn_linear(dY, w_kn = nn.Linear.weight ([out, in] IS [K'=out, N'=in]))
 *   dW = dY^T @ X      -> blgnn_grad_weight(dY, X)   (row-chunked reduction, below)
 *   GIN aggregation    -> blgnn_spmm_sum on the same (symmetric) CSR
 * The entry points below are the reductions and pointwise pieces around them. */

#define BLGNN_COL_SUM   0   /* out[c] = scale * sum_m a[m, c]                       */
#define BLGNN_COL_DOT   1   /* out[c] = scale * sum_m a[m, c] * b[m, c]             */
#define BLGNN_COL_SQDEV 2   /* out[c] = scale * sum_m (a[m, c] - mean[c])^2         */
#define BLGNN_COL_DOT_NORM 3 /* out[c] = scale * sum_m a[m, c] * (b[m, c] - mean[c]) * rstd[c] */
#define BLGNN_COL_DOT_NEG 4 /* out[c] = scale * sum_m a[m, c] * min(b[m, c], 0): the slope gradient of nn.PReLU(out_dim)
                             * (model/layers_util.py:63-64; model/layers.py:24,26,34), a = dy, b = pre-activation */

/* Column reduction over all M rows: bias gradients (sum of dY), train-mode BatchNorm1d statistics
 * (model/layers.py:39-40: mean = COL_SUM / M, biased var = COL_SQDEV / M, two-pass like torch) and its
 * gradients (dbeta = COL_SUM(dy), dgamma = COL_DOT_NORM(dy, x)).  Deterministic: per-512-row partial sums
 * in scratch (blgnn_col_reduce_scratch_elems(M, N) floats) then a fixed-order final pass. */
int64_t blgnn_col_reduce_scratch_elems(int32_t M, int32_t N);
int blgnn_col_reduce(const float *a, int64_t lda, const float *b, int64_t ldb, const float *mean,
                     const float *rstd, int32_t M, int32_t N, int32_t mode, float scale,
                     float *scratch, float *out, void *stream);

/* Weight gradient of a Linear over all rows: dw[o, i] = sum_m g[m, o] * x[m, i] (dw contiguous [O, I];
 * nn.Linear's [out, in] layout for g = dY, x = X; GATConv's [in, out] with the operands swapped).
 * Rows are reduced in chunks with exact-f32 MFMA, partials in scratch
 * (blgnn_grad_weight_scratch_elems(M, O, I) floats), final sum in a fixed order: deterministic. */
int64_t blgnn_grad_weight_scratch_elems(int32_t M, int32_t O, int32_t I);
int blgnn_grad_weight(const float *g, int64_t ldg, const float *x, int64_t ldx, int32_t M, int32_t O,
                      int32_t I, float *scratch, float *dw, void *stream);

/* The [N]-vector step between the two statistics passes and the normalisation of train-mode BatchNorm1d
 * (model/layers.py:39-40,56-57 under src/train.py:177-182), one launch: rstd = 1 / sqrt(var + eps) and, when the
 * running buffers are given, torch's momentum update running = (1 - m) running + m {mean, var * unbias}
 * (unbias = M / (M - 1)). */
int blgnn_bn_finalize(const float *mean, const float *var, int32_t N, float eps, float momentum, float unbias,
                      float *rstd, float *running_mean, float *running_var, void *stream);

/* Train-mode BatchNorm1d forward: y = act((x - mean) * rstd * gamma + beta) (gamma / beta nullable;
 * act = the activation that follows the BatchNorm in MLP: model/layers_util.py:51-54). */
int blgnn_bn_apply(const float *x, int64_t ldx, const float *mean, const float *rstd,
                   const float *gamma, const float *beta, int32_t M, int32_t N, int32_t act,
                   float *y, int64_t ldy, void *stream);

/* Train-mode BatchNorm1d backward given dgamma = sum dy * xhat and dbeta = sum dy over the `count` rows the
 * statistics were taken over (count = M on one GPU; the global row count when the sums were all-reduced
 * across data-parallel ranks):
 *   dx = gamma * rstd * (dy - dbeta / count - xhat * dgamma / count),  xhat = (x - mean) * rstd. */
int blgnn_bn_backward(const float *dy, int64_t lddy, const float *x, int64_t ldx, const float *mean,
                      const float *rstd, const float *gamma, const float *dgamma,
                      const float *dbeta, int32_t M, int32_t N, int64_t count, float *dx,
                      int64_t lddx, void *stream);

/* dx = dy * act'(.) evaluated from the activation's OUTPUT y (relu / sigmoid / tanh / identity); for PReLU the
 * derivative (v > 0 ? 1 : slope[c]) takes the sign of what is passed as y -- the pre-activation when slopes are
 * trained (any sign of the slope), the output where slope > 0 is known; dx may alias dy. */
int blgnn_act_backward(const float *dy, int64_t lddy, const float *y, int64_t ldy, int32_t act,
                       const float *slope, int32_t M, int32_t N, float *dx, int64_t lddx,
                       void *stream);

/* y = act(x), PReLU with one slope per column: the activation of a NodeEmbedding layer as its own step on the
 * training path, where the backward needs the pre-activation (model/layers.py:55 `x = self.act(x)` with
 * create_act('prelu', out_dim), model/layers_util.py:63-64); y may alias x. */
int blgnn_act_forward(const float *x, int64_t ldx, int32_t act, const float *slope, int32_t M, int32_t N,
                      float *y, int64_t ldy, void *stream);

/* y[N, M] = x[M, N]^T. */
int blgnn_transpose(const float *x, int64_t ldx, int32_t M, int32_t N, float *y, int64_t ldy,
                    void *stream);

/* Gradient of blgnn_segment_reduce (model/layers_aggregation.py:16-19): every row of segment g
 * receives dp[row_map ? row_map[g] : g] (/ max(count, 1) for MEAN); accumulate != 0 adds into dx
 * (multi-scale readout: each layer's activation also carries the next layer's gradient). */
int blgnn_segment_broadcast(const int32_t *seg, int32_t G, const float *dp, int64_t lddp,
                            const int32_t *row_map, int32_t d, int32_t kind, float *dx,
                            int64_t lddx, int32_t accumulate, void *stream);

/* Gradient of blgnn_l2_normalize_rows: dz = (dzn - zn (zn . dzn)) / max(|z|, eps). */
int blgnn_l2_normalize_backward(const float *z, int64_t ldz, const float *dzn, int64_t lddzn,
                                int32_t M, int32_t d, float eps, float *dz, int64_t lddz,
                                void *stream);

/* Gradient of the pair scorer's row gather (model/layers_link_pred.py:56-61):
 * dz[ids[b], :] += dcat[b, col0 : col0 + d]; dz zero-initialised by the caller (float atomics). */
int blgnn_pair_scatter_add(const float *dcat, int64_t ldc, int32_t col0, const int64_t *ids,
                           int32_t B, int32_t d, float *dz, int64_t lddz, void *stream);

/* Backward of the graph half of GATConv (model/layers.py:33,54), given dz = the gradient of
 * z_i = sum_{j in N(i) u {i}} alpha_ij h_j and the forward's scores / softmax statistics.
 * blgnn_gat_backward_scores: ds_dst[i], ds_src[j] = gradients of the two attention halves
 *   (d alpha_ij = dz_i . h_j, softmax + leaky-relu backward per group); ds_* ZERO-INITIALISED by the
 *   caller (the side opposite to the softmax group is accumulated with scalar float atomics).
 * blgnn_gat_backward_features: dh_j = sum_i alpha_ij dz_i + ds_dst[j] att[0:d] + ds_src[j] att[d:2d].
 * The caller finishes with blgnn_linear: datt = [ds_dst^T h, ds_src^T h], dW = x^T dh, dx = dh W^T.
 * (rowptr, col) = the symmetric CSR the forward used. */
int blgnn_gat_backward_scores(const int32_t *rowptr, const int32_t *col, int64_t nnz,
                              const float *h, int64_t ldh, const float *dz, int64_t lddz,
                              const float *s_dst, const float *s_src, const float *mx,
                              const float *den, int32_t n, int32_t d, int32_t group, float slope,
                              float *ds_dst, float *ds_src, void *stream);
int blgnn_gat_backward_features(const int32_t *rowptr, const int32_t *col, int64_t nnz,
                                const float *dz, int64_t lddz, const float *s_dst,
                                const float *s_src, const float *mx, const float *den,
                                const float *ds_dst, const float *ds_src, const float *att,
                                int32_t n, int32_t d, int32_t group, float slope, float *dh,
                                int64_t lddh, void *stream);

/* Backward of blgnn_edge_relu_sum on a SYMMETRIC CSR (model/layers_gmn.py:38-42):
 *   dp[i] = sum_{j in N(i)} [p_i + q_j > 0] ds[i],   dq[i] = sum_{j in N(i)} [p_j + q_i > 0] ds[j]. */
int blgnn_edge_relu_sum_backward(const int32_t *rowptr, const int32_t *col, int64_t nnz,
                                 const float *p, int64_t ldp, const float *q, int64_t ldq,
                                 const float *ds, int64_t ldds, int32_t n, int32_t d, float *dp,
                                 int64_t lddp, float *dq, int64_t lddq, void *stream);

/* Backward of blgnn_gmn_match (model/layers_gmn.py:55-84): dx from du, per pair of graphs, any size.
 * cell_off: device int64[num_pairs + 1], first cell of every pair's n1 x n2 matrices (prefix sum of
 * n1 * n2); cell_scratch: 4 * total_cells floats; row_scratch: 3 * total_rows floats (total_rows =
 * seg[2 * num_pairs]).  dx must not alias x / du. */
int blgnn_gmn_match_backward(const int32_t *seg, int32_t num_pairs, const float *x, int64_t ldx,
                             const float *du, int64_t lddu, const int64_t *cell_off,
                             int64_t total_cells, float *cell_scratch, float *row_scratch,
                             float *dx, int64_t lddx, int32_t d, int64_t total_rows, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* BLGNN_H */
