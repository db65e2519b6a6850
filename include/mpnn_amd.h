/*
 * mpnn_amd -- C ABI of the MI355X (gfx950) message-passing hot path.
 *
 * The reference (hochshi/mpnn) has no FFI: its hot path is three families of Python
 * nn.Module operators lowered to ATen ops.  Each entry point below replaces the ATen
 * call sequence of one reference operator; the citation after "replaces:" is the
 * reference file:line (relative to the reference root) whose arithmetic it performs.
 *
 * Conventions (every function):
 *   - all data pointers are DEVICE pointers owned by the caller; fp32 data, int32 indices;
 *   - `stream` is a hipStream_t passed as void*; work is enqueued, never synchronised;
 *   - no allocation, no host<->device copy;
 *   - process-wide state is exactly: the last-error string (one per calling thread), the environment switches
 *     (alternate kernels for A/B runs, README "Switches"; read once by mpnn_init() or by the first call that needs
 *     one), and one cached hipFuncSetAttribute result per kernel family that needs more than 64 KB of LDS (set on
 *     that family's first launch; a failure is returned as MPNN_ELAUNCH by every launch of it).  All three are
 *     initialised exactly once under C++11 static-initialisation guarantees, so concurrent first calls are safe;
 *   - re-entrant; returns 0 on success, a negative MPNN_E* code on a rejected call;
 *   - row-major, densely packed arrays; "V" = atoms (rows of node arrays), "E" = directed
 *     edges sorted by destination atom (CSR: row_ptr[V+1], col_idx[E] = source atom).
 *
 * Python binds these with ctypes (mpnn_amd/_lib.py); see INTEGRATION.md for the stub a
 * maintainer of the reference would add.
 */
#ifndef MPNN_AMD_H
#define MPNN_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MPNN_OK 0
#define MPNN_EINVAL (-1)   /* bad argument (null pointer, negative size, unsupported width) */
#define MPNN_ELAUNCH (-2)  /* the HIP runtime rejected a launch (see mpnn_last_error_string) */
#define MPNN_EWORKSPACE (-3) /* workspace too small */

#define MPNN_MAX_FEATURES 512   /* largest nf / mf any kernel accepts */

/* ABI version: major*10000 + minor*100 + patch. */
int mpnn_version(void);
/* Message for the last non-zero return on the calling thread ("" if none). */
const char* mpnn_last_error_string(void);
/* Optional explicit initialisation: reads the environment switches now (otherwise the first call that consults one
 * does).  Idempotent, thread-safe, touches no device. */
int mpnn_init(void);

/* ------------------------------------------------------------------ graph build ---- */
/*
 * Dense padded batch -> CSR by destination, in exactly the order adj.nonzero() yields
 * (lexicographic (b,i,j)); replaces: the implicit all-pairs iteration of
 * mpnn_functions/message/edge_network.py:34-38 and the adj multiply of
 * mpnn_functions/message_aggregators/adjacent_message_agg.py:18.
 *   adj  [rows, cols]      (rows = B*N, cols = N), may be NULL
 *   bfm  [rows, cols, ef]  may be NULL (ef ignored then)
 * A pair (row, j) is an edge when adj != 0 or any of its ef bond features != 0.
 * Step 1 writes row_ptr[rows+1] (row_ptr[rows] = E, device side); the caller reads E,
 * allocates, and calls step 2.  Node ids are flat padded ids: source = (row/cols)*cols + j.
 */
size_t mpnn_csr_workspace_bytes(int64_t rows);
int mpnn_csr_count(const float* adj, const float* bfm, int64_t rows, int cols, int ef,
                   int32_t* row_ptr, void* workspace, size_t workspace_bytes, void* stream);
int mpnn_csr_fill(const float* adj, const float* bfm, int64_t rows, int cols, int ef,
                  const int32_t* row_ptr, int32_t* col_idx,
                  float* edge_weight /* [E] adj value (1 if adj NULL), may be NULL */,
                  float* edge_feat   /* [E, ef] gathered bfm rows, may be NULL */,
                  void* stream);

/* ------------------------------------------------------------------ message -------- */
/*
 * Typed edge message: msg[e, :] = A[type[e]] (mf x nf) . h[src[e], :]
 * replaces: mpnn_functions/message/edge_network.py:40,52 (per-pair form),
 *           :50 (the bmm), mpnn_functions/message/ggnn_msg_pass.py:19-31.
 * K may be anything from 1 to E: few types (discrete bond features) run as dense contractions on the matrix
 * cores; above 4096 types (continuous bond features, a matrix per bond) each matrix is streamed once and applied
 * to its one or two edges.
 * Edges are visited in type-sorted order so one tile multiplies by one matrix:
 *   order[E]     edge ids stably sorted by type
 *   type_ptr[K+1] start of each type's run inside `order`
 *   gate         optional [E, nf]: msg = A . (gate[e] * h[src[e]])  (AttEdgeNetwork,
 *                mpnn_functions/message/att_edge_network.py:26-31), may be NULL
 */
int mpnn_edge_message_f32(const float* h, const float* A, const int32_t* src,
                          const int32_t* order, const int32_t* type_ptr, const float* gate,
                          float* msg, int64_t V, int64_t E, int K, int nf, int mf, void* stream);
/* Backward of the above for d(msg):
 *   dx[e, :] = A[type[e]]^T . dmsg[e, :]   (E x nf; multiplied by gate / scattered by caller kernels)
 *   dA[k]   += sum_{e in type k} dmsg[e] (x) x[e]     (K x mf x nf, must be zeroed by the caller)
 */
int mpnn_edge_message_bwd_f32(const float* h, const float* A, const int32_t* src,
                              const int32_t* order, const int32_t* type_ptr, const float* gate,
                              const float* dmsg, float* dx, float* dA,
                              int64_t V, int64_t E, int K, int nf, int mf, void* stream);

/*
 * Backward of message FOLLOWED BY adjacency-weighted aggregation, weight-gradient part, without a
 * materialised dmsg: dA[k] += sum_{e in type k} (w[e] * dagg[dst[e]]) (x) (gate[e] * h[src[e]]).
 * replaces: the autograd of edge_network.py:40,52 composed with adjacent_message_agg.py:18.
 * nf = mf in {64, 128, 256} only (returns MPNN_EINVAL otherwise: callers fall back to segsum_bwd + edge_message_bwd).
 */
int mpnn_edge_message_agg_bwd_da_f32(const float* dagg, const float* h, const int32_t* src,
                                     const int32_t* dst, const float* w /* may be NULL */,
                                     const int32_t* order, const int32_t* type_ptr, const float* gate,
                                     float* dA, int64_t V, int64_t E, int K, int nf, int mf, void* stream);

/*
 * The same with AttEdgeNetwork's gate evaluated in flight, no (E, nf) gate tensor:
 *   gate[e, c] = exp2(log2(e) * (z_atom[dst e, c] + q[type e, c]) - stats[dst e, type e].x) * stats[dst e, type e].y,
 * stats_by_atom [V, K, 2] = (log2(e) * max_c(z + q), 1 / sum_c exp(z + q - max)) as mpnn_message_aggregate_wide_gated_bwd_f32
 * writes them.  replaces: the autograd of att_edge_network.py:26-31 composed with adjacent_message_agg.py:18 for the
 * matrices.  nf = mf = 128 only.
 */
int mpnn_edge_message_agg_bwd_da_att_f32(const float* dagg, const float* h, const int32_t* src, const int32_t* dst,
                                         const int32_t* order, const int32_t* type_ptr, const float* z_atom,
                                         const float* q, const float* stats_by_atom, float* dA,
                                         int64_t V, int64_t E, int K, int nf, int mf, void* stream);

/*
 * Feature gate of AttEdgeNetwork: gate[e, :] = softmax_f( z_atom[dst[e], :] + q[type[e], :] )
 * replaces: mpnn_functions/message/att_edge_network.py:18-21 (cat[h_i, e_ij], Linear(nf+ef -> nf), Softmax(dim=-1))
 * after the caller split the Linear into its atom part z_atom[i] = W_h h_i + b  [V,F] and its bond part
 * q[k] = W_e e_k  [K,F] (one row per distinct bond-feature row).  dst[E] = destination atom of each edge.
 * F a multiple of 4, <= 256.
 * Backward, from d(gate): dz = gate * (dgate - <gate, dgate>);
 *   dz_atom[i, :] = sum_{e in row i} dz[e, :]  (written),  dq[k, :] += sum_{e of type k} dz[e, :]  (caller zeroes dq).
 */
int mpnn_att_gate_f32(const float* z_atom, const float* q, const int32_t* dst, const int32_t* edge_type,
                      float* gate, int64_t V, int64_t E, int K, int F, void* stream);
int mpnn_att_gate_bwd_f32(const float* gate, const float* dgate, const int32_t* row_ptr, const int32_t* edge_type,
                          float* dz_atom, float* dq, int64_t V, int64_t E, int K, int F, void* stream);

/*
 * Backward of gated message FOLLOWED BY adjacency-weighted aggregation, gate-gradient part, without a materialised
 * dmsg or dx:  dgate[e, :] = (A[type e]^T . (w[e] * dagg[dst[e], :])) * h[src[e], :].
 * replaces: the autograd of att_edge_network.py:26-31 composed with adjacent_message_agg.py:18 for d(gate).
 * nf = mf in {64, 128}, K <= 64 only (MPNN_EINVAL otherwise: callers use segsum_bwd + edge_message_bwd + a product).
 */
int mpnn_edge_message_agg_bwd_dgate_f32(const float* dagg, const float* A, const float* h, const int32_t* src,
                                        const int32_t* dst, const float* w /* may be NULL */,
                                        const int32_t* order, const int32_t* type_ptr, float* dgate,
                                        int64_t V, int64_t E, int K, int nf, int mf, void* stream);

/*
 * Typed edge message FUSED with the neighbour sum, no (E, mf) message tensor in HBM:
 *   out[i, :] = sum_{e in row i} A[type[e]] . h[src[e], :]
 * replaces: mpnn_functions/message/edge_network.py:50-51 (edge_embed.bmm(...): message and sum as one product),
 *           i.e. edge_network.py:40,52 composed with message_aggregators/adjacent_message_agg.py:18;
 *           also ggnn_msg_pass.py:30-31 (A = the bond-type table).
 * The edges arrive as a TILE PLAN (built once per batch from graph_ptr / row_ptr / src / type; layout in
 * mpnn_amd/graph.py::TilePlan): molecule-aligned tiles of at most mpnn_message_aggregate_tile_atoms() atoms whose h rows
 * are staged in LDS once; the tile's atoms sorted by per-type in-degree and dealt in blocks of 16 (one per wave);
 * a row-tile = the rank-th incoming type-k edge of each atom of one block.
 *   tile_rec[T][16]   first atom, atoms, first row-tile of block 0..7, end of block 7, zeros
 *   tile_atom[T][128] atom id at every sorted position of the tile (block = position / 16), -1 = none
 *   slots[16R]        (source atom - first atom) | valid << 14 | bond type << 16; an empty slot names source row 128
 *                     (zeros); row-tiles in (tile, block, type, rank) order
 * nf = mf = 64, K <= mpnn_message_aggregate_max_types(), at most mpnn_message_aggregate_max_row_tiles() row-tiles per
 * block, unit edge weights only (callers run mpnn_edge_message_f32 + mpnn_segsum_f32 otherwise).  Sum order inside
 * a row: types ascending, edge order within a type (deterministic).
 * mpnn_plan_tiles_host is a HOST helper (host pointers, no device work): greedy packing of whole molecules into
 * tiles; writes tile_ptr (capacity G+2) and returns the tile count, or -1 if a molecule exceeds max_atoms.
 */
int64_t mpnn_plan_tiles_host(const int32_t* graph_ptr_host, int64_t G, int max_atoms, int32_t* tile_ptr_host);
int mpnn_message_aggregate_tile_atoms(void);
int mpnn_message_aggregate_max_types(void);
int mpnn_message_aggregate_max_row_tiles(void);
int mpnn_message_aggregate_f32(const float* h, const float* A, const int32_t* tile_rec, const int32_t* tile_atom,
                               const int32_t* slots, float* out,
                               int64_t V, int64_t num_tiles, int K, int nf, int mf, void* stream);

/*
 * The same fused product at nf = mf in {128, 256} for molecules of up to
 * mpnn_message_aggregate_wide_tile_atoms() = 256 atoms: typed aggregate-then-contract,
 *   out[i, :] = sum_k A[k] . S_k[i],   S_k[i] = sum_{e in row i, type(e) = k} h[src[e], :]
 * replaces: mpnn_functions/message/edge_network.py:50-51 (edge_embed.bmm(...)), i.e. :40,52 composed with
 *           mpnn_functions/message_aggregators/adjacent_message_agg.py:18, where neither the K matrices nor a tile's
 *           h rows fit in LDS.  The h rows are read once and the out rows written once; no (E, mf) tensor exists.
 * Work list (built once per batch, mpnn_amd/graph.py::WidePlan): molecule-aligned tiles of <= 256 atoms, their atoms
 * sorted by bond-type pattern into blocks of 32 (one wave each);
 *   tile_rec[T,4]     first atom, atoms, first slot row, slot rows (<= mpnn_message_aggregate_wide_max_rows())
 *   tile_atom[T,256]  atom id of (block, row), -1 = none
 *   blk_off[T,8K+1]   first slot row of every (block, type), relative to the tile's first; then the tile's count
 *   slots[32 R]       16-bit words, row (block, type, rank): word m = source row of the rank-th type-k neighbour of the
 *                     block's atom m, relative to the tile's first atom; 256 = none
 * `workspace` (mpnn_message_aggregate_wide_workspace_bytes(K, nf) bytes) receives the K matrices as fp16 pieces in the
 * kernel's chunk order on every call.  K <= mpnn_message_aggregate_wide_max_types(), unit edge weights, no gate
 * (callers run mpnn_edge_message_f32 + mpnn_segsum_f32 otherwise).  Sum order inside an output row: types ascending,
 * within a type the edge order -- deterministic.
 */
int mpnn_message_aggregate_wide_tile_atoms(void);
int mpnn_message_aggregate_wide_max_types(void);
int mpnn_message_aggregate_wide_max_rows(void);
size_t mpnn_message_aggregate_wide_workspace_bytes(int K, int nf);
int mpnn_message_aggregate_wide_f32(const float* h, const float* A, const int32_t* tile_rec, const int32_t* tile_atom,
                                    const int32_t* blk_off, const int16_t* slots, float* out, void* workspace,
                                    size_t workspace_bytes, int64_t V, int64_t num_tiles, int K, int nf, int mf,
                                    void* stream);
/*
 * The same with the feature gate of AttEdgeNetwork applied inside the kernel (replaces: att_edge_network.py:18-31
 * composed with adjacent_message_agg.py:18): out[i] = sum_e A[type e] . (gate[e] * h[src e]) with
 * gate[e, :] = softmax_f(z_atom[dst e, :] + q[type e, :]).  The gate depends on the destination atom and the bond type
 * only, so the kernel forms it per (atom, type) from two softmax statistics (a small pre-pass into the workspace) and
 * multiplies the typed neighbour sums by it; no (E, nf) gate tensor exists.  z_atom [V,nf], q [K,nf].
 * nf = mf = 128, K <= 4; workspace: mpnn_message_aggregate_wide_gated_workspace_bytes(K, nf, num_tiles).
 */
size_t mpnn_message_aggregate_wide_gated_workspace_bytes(int K, int nf, int64_t num_tiles);
int mpnn_message_aggregate_wide_gated_f32(const float* h, const float* A, const float* z_atom, const float* q,
                                          const int32_t* tile_rec, const int32_t* tile_atom, const int32_t* blk_off,
                                          const int16_t* slots, float* out, void* workspace, size_t workspace_bytes,
                                          int64_t V, int64_t num_tiles, int K, int nf, int mf, void* stream);

/*
 * Backward of the gated form for the gate logits, per (atom, type) on the forward's plan -- no (E, nf) gate or
 * gate-gradient tensor (replaces: the autograd of att_edge_network.py:18-31 composed with adjacent_message_agg.py:18 for
 * z_atom and q).  With X_ik = gate_ik * S_ik (S_ik = the typed neighbour sum), T_ik = A_k^T dagg_i, u = X * T,
 * D_ik = sum_f u_ikf:   dz_atom[i] = sum_k (u_ik - gate_ik D_ik),   dq[k] = sum_i (u_ik - gate_ik D_ik).
 *   fwd_workspace   the workspace mpnn_message_aggregate_wide_gated_f32 was given for the same (z_atom, q, plan): its
 *                   softmax statistics are read back
 *   dz_atom [V,nf]  written (every row)
 *   dq_part         [mpnn_message_aggregate_wide_gated_bwd_parts()][K][nf], ZERO-FILLED by the caller: partial sums of
 *                   sum_i gate_ik D_ik;  dq[k] = colsum_over_mf(A[k] * dA[k]) - sum_over_parts(dq_part[:, k])  with dA the
 *                   weight gradient (sum_i u_ik = the column sums of A_k * dA_k)
 *   stats_by_atom   [V,K,2] written: the statistics in atom order, for mpnn_edge_message_agg_bwd_da_att_f32
 * nf = mf = 128, K <= 4; workspace: mpnn_message_aggregate_wide_gated_bwd_workspace_bytes(K, nf).
 */
size_t mpnn_message_aggregate_wide_gated_bwd_workspace_bytes(int K, int nf);
int mpnn_message_aggregate_wide_gated_bwd_parts(void);
int mpnn_message_aggregate_wide_gated_bwd_f32(const float* h, const float* A, const float* z_atom, const float* q,
                                              const float* dagg, const void* fwd_workspace, size_t fwd_workspace_bytes,
                                              const int32_t* tile_rec, const int32_t* tile_atom, const int32_t* blk_off,
                                              const int16_t* slots, float* dz_atom, float* dq_part, float* stats_by_atom,
                                              void* workspace, size_t workspace_bytes, int64_t V, int64_t num_tiles,
                                              int K, int nf, int mf, void* stream);

/*
 * BiLiniearEdgeNetwork message on the dense padded batch: out[b,i,j,k] = sum_{a,c} afm[b,j,a] T[b,i,j][a,k,c] afm[b,i,c]
 * with T = the pair's nf^3 bond features viewed (nf, nf, nf).
 * replaces: mpnn_functions/message/bilinear_edge_network.py:25-37.  afm [B,N,nf], bfm [B,N,N,nf^3], out [B,N,N,nf]; nf <= 8.
 */
int mpnn_bilinear_message_f32(const float* afm, const float* bfm, float* out, int64_t B, int N, int nf, void* stream);

/* ------------------------------------------------------------------ edge tower ----- */
/*
 * The run of n_layers aliased Linear(L, L, bias=False) + ReLU blocks of the bond-feature tower
 * replaces: mpnn_functions/message/edge_network.py:20 (`[Sequential(Linear, act)] * 50`), evaluated
 * on the R distinct bond-feature rows of a batch.
 *   x [R, L], W [L, L] (out, in)  ->  acts [(n_layers+1), R, L]: acts[0] = x, acts[l+1] = relu(acts[l] W^T)
 * Backward of the chain for dout = d/d acts[n_layers]:
 *   dys [n_layers, R, L] (gradient at each layer's pre-activation), dx [R, L];
 *   the weight gradient is dW = sum_l dys[l]^T acts[l], one GEMM left to the caller.
 * L <= 256.
 */
int mpnn_tower_chain_f32(const float* x, const float* W, float* acts, int R, int L, int n_layers, void* stream);
int mpnn_tower_chain_bwd_f32(const float* dout, const float* W, const float* acts, float* dys, float* dx,
                             int R, int L, int n_layers, void* stream);

/* ------------------------------------------------------------------ aggregator ----- */
/*
 * out[i, :] = sum_{e in row i} w[e] * msg[e, :]      (w == NULL: plain sum)
 * replaces: mpnn_functions/message_aggregators/adjacent_message_agg.py:18 and, with the
 * caller's weights, weighted_adjacent_message_agg.py:20 / attention_message_agg.py:24.
 * Summation order inside a row is edge order (deterministic).
 */
int mpnn_segsum_f32(const float* msg, const int32_t* row_ptr, const float* w, float* out,
                    int64_t V, int F, void* stream);
/* dmsg[e, :] = w[e] * dout[row(e), :] */
int mpnn_segsum_bwd_f32(const float* dout, const int32_t* row_ptr, const float* w, float* dmsg,
                        int64_t V, int F, void* stream);
/*
 * Gathered variant: out[i, :] = sum_{e in row i} w[e] * x[idx[e], :]  -- used for the
 * transposed scatter of the backward pass (idx = edge ids sorted by source) and for
 * per-molecule sums (graph_ptr as row_ptr, idx NULL).
 */
int mpnn_segsum_gather_f32(const float* x, const int32_t* row_ptr, const int32_t* idx,
                           const float* w, float* out, int64_t V, int F, void* stream);

/* ------------------------------------------------------------------ update --------- */
/*
 * Masked GRU cell, gate order r,z,n, weights stored (in, 3H) as the reference does:
 *   gi = m W_ih + b_ih ; gh = h W_hh + b_hh
 *   r = sigmoid(gi_r+gh_r)*mask ; z = sigmoid(gi_z+gh_z)*mask ; n = tanh(gi_n + r*gh_n)*mask
 *   out = ((1-z)*n + z*h) * mask
 * replaces: mpnn_functions/update/gru_update.py:26-35 and :66-68.
 *   m [V,H], h [V,H], mask [V], W_ih [H,3H], W_hh [H,3H], b_ih [3H], b_hh [3H], out [V,H]
 *   saved: optional [V,4H] = (r, z, n, gh_n) kept for the backward pass, may be NULL
 *   workspace: optional, mpnn_gru_fwd_workspace_bytes(V, H) bytes (0 at most widths).  At H = 128 / 256 it receives the
 *   two weight matrices as fp16 pieces in the kernel's chunk order, written once per call and copied global -> LDS by
 *   the GRU kernel; with NULL the kernel splits its weight chunks itself (same results, more vector work).
 */
size_t mpnn_gru_fwd_workspace_bytes(int64_t V, int H);
int mpnn_gru_update_f32(const float* m, const float* h, const float* mask,
                        const float* W_ih, const float* W_hh, const float* b_ih, const float* b_hh,
                        float* out, float* saved, void* workspace, size_t workspace_bytes, int64_t V, int H,
                        void* stream);
/*
 * The same update with the masked batch norm that the lipo / attention models put in front of its `h` input fused in
 * (replaces: models/mask_batch_norm.py:5-38 composed with gru_update.py:26-35 as models/att_model.py:58 and
 * lipo_basic_model.py:85 call them).  `h_raw` is the previous update's RAW output y; the state the
 * reference normalises in passes of its own is  hn = (y * h_scale[col] + h_shift[col]) * mask  (h_scale = gamma /
 * std, h_shift = beta - mean * gamma / std of the norm in question) and is formed inside the kernel.  The caller folds
 * the same affine map into the hidden weights: W_hh_folded[k, :] = h_scale[k] * W_hh[k, :], b_hh_folded = b_hh +
 * h_shift W_hh.  out_moments [2H] doubles, ACCUMULATED (caller zeroes): column sums of `out` and of out^2 over all
 * atoms -- the moments the norm that follows this update needs, so it runs no reduction pass.  With saved != NULL the
 * kernel also writes h_norm [V,H] (= hn, the `h` operand of mpnn_gru_update_bwd_f32).  mpnn_gru_update_norm_supported(H):
 * 2 = the wide split-precision kernels (H = 128 / 256; workspace of mpnn_gru_fwd_workspace_bytes required), 1 = the generic
 * fp32 kernel (every other H <= 256, and every H under MPNN_GRU_MATH=fp32: the kernel the plain update runs on at those
 * widths, 32 / 64 / 128 excepted, which have faster un-normed forms; no workspace), 0 = none.
 */
int mpnn_gru_update_norm_supported(int H);
int mpnn_gru_update_norm_f32(const float* m, const float* h_raw, const float* mask, const float* W_ih,
                             const float* W_hh_folded, const float* b_ih, const float* b_hh_folded,
                             const float* h_scale, const float* h_shift, float* out, float* saved,
                             float* h_norm, double* out_moments, void* workspace, size_t workspace_bytes,
                             int64_t V, int H, void* stream);
/*
 * Backward: given dout [V,H] and `saved`, writes dm [V,H], dh [V,H] and ACCUMULATES into
 * dW_ih, dW_hh [H,3H], db_ih, db_hh [3H] (caller zeroes them).  `workspace`: mpnn_gru_bwd_workspace_bytes(V, H)
 * bytes.  At H = 64 the backward is one kernel that keeps the gate gradients in LDS and needs NO workspace (the
 * function returns a token 16 bytes); at H = 128 / 256 it holds the compact gate gradients
 * (d a_r | d a_z | d a_n | r * d a_n), 4H values per atom, that the dm/dh and dW kernels read -- as two fp16 pieces
 * per value behind one power-of-two scale per ATOM (16 H bytes per atom, V rounded up to 32, plus one float per atom),
 * followed by the dm/dh kernel's pre-split weight images; at other widths (and with MPNN_GRU_MATH=fp32) the
 * pre-activation gradients [V,6H] = (dgi | dgh).  Precision: an atom's dm / dh rows keep the float32 bar whatever the
 * gradient magnitudes of its neighbours in memory are (tests/test_backward_gpu.py::test_gru_backward_range_guards,
 * profiles rows_1e6_in_tile / rows_1e8_in_tile).
 */
size_t mpnn_gru_bwd_workspace_bytes(int64_t V, int H);
int mpnn_gru_update_bwd_f32(const float* dout, const float* m, const float* h, const float* mask,
                            const float* W_ih, const float* W_hh, const float* saved,
                            float* dm, float* dh, float* dW_ih, float* dW_hh,
                            float* db_ih, float* db_hh, void* workspace, size_t workspace_bytes,
                            int64_t V, int H, void* stream);
/*
 * Backward of mpnn_gru_update_norm_f32 with BOTH masked-norm backward passes fused in (models/mask_batch_norm.py:5-38
 * differentiated).  Same outputs and accumulation rules as mpnn_gru_update_bwd_f32 on h = h_norm, plus:
 *   out_norm_k [3H] = k1 | k2 | k4, or NULL.  Non-NULL: `dout` is the gradient of norm(out) (out = this update's raw
 *     output y), and the gate-gradient kernel forms  dy = dout * k1 + y * k2 + k4  on rows with mask 1 from the saved
 *     gates (y is not read).  For a norm with scale s (sqrt(var + eps), or sqrt(var) + eps), weight g, bias b, batch
 *     mean, count n and the column sums S_b = sum dout, S_y = sum dout * out (against the norm's RAW input, so that a
 *     weight entry of 0 keeps its gradient and nothing is divided by g):
 *       S_g = S_y - mean S_b;  dvar = -g S_g / (2 s^2 root)  (root = s, or sqrt(var) with eps outside);
 *       k1 = g / s;  k2 = 2 dvar / n;  k4 = -g S_b / (s n) - mean k2;  dweight = S_g / s;  dbias = S_b.
 *   in_norm_sums [2H] doubles, ACCUMULATED (caller zeroes), or NULL.  Non-NULL: h_norm = norm(h_raw), h_raw [V,H] = the
 *     `h_raw` mpnn_gru_update_norm_f32 was called with; the dm | dh kernel adds the column sums of dh_norm and of
 *     dh_norm * h_raw -- S_b and S_y of THAT norm -- so the update before this one can be called with their out_norm_k
 *     and no norm-backward pass runs at all.  h_raw may be NULL when in_norm_sums is.
 * Widths as mpnn_gru_update_norm_f32; workspace: mpnn_gru_norm_bwd_workspace_bytes(V, H).
 */
size_t mpnn_gru_norm_bwd_workspace_bytes(int64_t V, int H);
int mpnn_gru_update_norm_bwd_f32(const float* dout, const float* m, const float* h_norm, const float* mask,
                                 const float* W_ih, const float* W_hh, const float* saved, const float* out_norm_k,
                                 float* dm, float* dh_norm, float* dW_ih, float* dW_hh, float* db_ih, float* db_hh,
                                 double* in_norm_sums, const float* h_raw, void* workspace, size_t workspace_bytes,
                                 int64_t V, int H, void* stream);

/* ------------------------------------------------------------------ masked batch norm */
/*
 * Batch normalisation over the real atoms of a batch; replaces: models/mask_batch_norm.py:5-15 (MaskBatchNorm,
 * flags = MPNN_BN_EPS_INSIDE) and :18-38 (MaskBatchNorm1d, flags = MPNN_BN_MASKED_MEAN; eval mode adds
 * MPNN_BN_USE_STATS and reads `mean` / `var` as the running statistics).
 *   x, y [V,F]; mask [V] or NULL; weight, bias [F] or both NULL; mean, var [F] (batch statistics, written
 *   unless USE_STATS; var is the biased masked variance); count_out: device scalar sum(mask), may be NULL.
 * Backward (training mode only): dx [V,F], dweight / dbias [F] (may be NULL); `count` = device scalar.
 */
#define MPNN_BN_MASKED_MEAN 1
#define MPNN_BN_EPS_INSIDE 2
#define MPNN_BN_USE_STATS 4
size_t mpnn_masked_bn_workspace_bytes(int F);
int mpnn_masked_bn_fwd_f32(const float* x, const float* mask, const float* weight, const float* bias,
                           float* y, float* mean, float* var, float* count_out, int64_t V, int F, float eps,
                           int flags, void* workspace, size_t workspace_bytes, void* stream);
int mpnn_masked_bn_bwd_f32(const float* dout, const float* x, const float* mask, const float* weight,
                           const float* mean, const float* var, float* dx, float* dweight, float* dbias,
                           int64_t V, int F, float eps, int flags, const float* count, void* workspace,
                           size_t workspace_bytes, void* stream);

/*
 * Per-column constants of the norm fused into the update (row f2): tiny kernels, one launch each.
 * mpnn_norm_fold_f32: moments [2F] doubles (sum y | sum y^2 from mpnn_gru_update_norm_f32), count [1] = sum(mask) ->
 *   mean, var [F] (biased variance, as mask_batch_norm.py:14,31), h_scale, h_shift [F] of the norm selected by
 *   (weight, bias, eps, flags: the MPNN masked-norm flags of mpnn_masked_bn_fwd_f32), and W_hh_folded [F,3F],
 *   b_hh_folded [3F] as mpnn_gru_update_norm_f32 takes them.  F <= 256.
 * mpnn_norm_bwd_consts_f32: sums [2F] doubles (in_norm_sums of mpnn_gru_update_norm_bwd_f32) + the norm's statistics
 *   -> out_norm_k [3F] for the update in front of that norm; dweight / dbias [F] (affine norms) are ACCUMULATED.
 * mpnn_norm_bwd_sums_f32: the same two sums for a norm whose output left the chain (no dm | dh kernel behind it):
 *   sums [2F] doubles += (sum dout * mask | sum dout * mask * y_raw), one read of dout and of the norm's raw INPUT.
 *   F = 4 * 2^k <= 1024.
 */
int mpnn_norm_bwd_sums_f32(const float* dout, const float* y_raw, const float* mask, double* sums, int64_t V, int F,
                           void* stream);
int mpnn_norm_fold_f32(const double* moments, const float* count, const float* weight, const float* bias,
                       const float* W_hh, const float* b_hh, float* mean, float* var, float* h_scale, float* h_shift,
                       float* W_hh_folded, float* b_hh_folded, int F, float eps, int flags, void* stream);
int mpnn_norm_bwd_consts_f32(const double* sums, const float* mean, const float* var, const float* count,
                             const float* weight, float* out_norm_k, float* dweight, float* dbias, int F, float eps,
                             int flags, void* stream);

/* ------------------------------------------------------------------ per-batch index work (round 4) */
/*
 * The CSR-derived arrays of a molecule batch and the work lists of the fused message + sum kernels, built tile by tile (one
 * workgroup per tile of whole molecules) instead of by generic sorts; replaces per NEW batch what the reference's collate
 * does on the host (pre_process/data_loader.py:50-70) plus this package's own index work (mpnn_amd/graph.py holds the same
 * builders in torch ops: the CPU path, and the fallback when a batch is not separate molecules).  All device pointers.
 *   mpnn_plan_index_tiles: tile_ptr [T+1] = molecule-aligned tiles of at most mpnn_plan_index_tile_atoms() atoms
 *     (mpnn_plan_tiles_host).  Writes edge_dst [E], t_row_ptr [V+1] / t_eid [E] (edges grouped by source atom, ascending
 *     edge id inside a group) and hist [T,K] (edges of type k per tile; skipped when K > mpnn_plan_index_max_types()).
 *     flags [1] (caller zeroes): |= 1 an edge leaves its tile, |= 2 a tile with too many edges -- results invalid then.
 *   mpnn_plan_type_order: offsets [K,T] = exclusive scan of hist in (type, tile) order (int64) -> order [E] (edge ids
 *     stably sorted by type), type_ptr [K+1].
 *   mpnn_tile_plan_count / _fill: graph.py::TilePlan (tiles of <= 128 atoms): count writes need [T,8,K] (row-tiles per
 *     relabelled block and type, int64), tile_atom [T,128], atom_slot [V] (block << 4 | row), flags |= 4 for an edge that
 *     leaves its tile; the caller scans need (rt_start [T*8*K+1], int64, exclusive + total) and fill writes
 *     slots / slot_eid [16 * total] and rt_ptr [T*8+1].  K <= 8.
 *   mpnn_wide_plan_count / _fill: graph.py::WidePlan (tiles of <= 256 atoms, blocks of 32): need [T,8,K], tile_atom
 *     [T,256], atom_slot [V] (position in the tile); fill writes slots (int16) / slot_eid [32 * total], tile_rec [T,4],
 *     blk_off [T,8K+1].
 */
int mpnn_plan_index_tile_atoms(void);
int mpnn_plan_index_max_types(void);
int mpnn_plan_index_tiles(const int32_t* row_ptr, const int32_t* col_idx, const int32_t* edge_type,
                          const int32_t* tile_ptr, int64_t V, int64_t E, int64_t T, int K, int32_t* edge_dst,
                          int32_t* t_row_ptr, int32_t* t_eid, int32_t* hist, int32_t* flags, void* stream);
int mpnn_plan_type_order(const int32_t* row_ptr, const int32_t* edge_type, const int32_t* tile_ptr,
                         const int64_t* offsets, int64_t E, int64_t T, int K, int32_t* order, int32_t* type_ptr,
                         void* stream);
int mpnn_tile_plan_count(const int32_t* row_ptr, const int32_t* col_idx, const int32_t* edge_type,
                         const int32_t* tile_ptr, int64_t T, int K, int64_t* need, int32_t* tile_atom,
                         int32_t* atom_slot, int32_t* flags, void* stream);
int mpnn_tile_plan_fill(const int32_t* row_ptr, const int32_t* col_idx, const int32_t* edge_type,
                        const int32_t* tile_ptr, int64_t T, int K, const int64_t* rt_start, const int32_t* atom_slot,
                        int32_t* slots, int32_t* slot_eid, int32_t* rt_ptr, void* stream);
int mpnn_wide_plan_count(const int32_t* row_ptr, const int32_t* col_idx, const int32_t* edge_type,
                         const int32_t* tile_ptr, int64_t T, int K, int64_t* need, int32_t* tile_atom,
                         int32_t* atom_slot, int32_t* flags, void* stream);
int mpnn_wide_plan_fill(const int32_t* row_ptr, const int32_t* col_idx, const int32_t* edge_type,
                        const int32_t* tile_ptr, int64_t T, int K, const int64_t* start, const int32_t* atom_slot,
                        int16_t* slots, int32_t* slot_eid, int32_t* tile_rec, int32_t* blk_off, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MPNN_AMD_H */
