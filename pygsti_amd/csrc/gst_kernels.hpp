// gst_kernels.hpp -- launch interface between the C ABI (gst_abi.cpp) and the HIP kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace gst {

// Sentinel of the overlap protocol: the base-state cache and the base probabilities are filled with this bit pattern
// (both 32-bit halves equal, so one hipMemsetD32Async does it) before the fused launch; a consumer that reads it knows
// the producing chain has not got there yet.  A quiet NaN with a payload no arithmetic produces.
constexpr uint32_t OVL_SENTINEL32 = 0x7FF8DEADu;
constexpr unsigned long long OVL_SENTINEL64 = 0x7FF8DEAD7FF8DEADull;


// Per-lane perturbation ("special") tables, one entry per lane slot (n_waves * 64 entries).
// Lane slot q of wave w = w*64 + q.  kind -1 = no perturbation for that special.
struct LaneTables {
    const int32_t* col;     // output column (dest index) of the lane, -1 = idle lane
    const int32_t* kind[2]; // GST_KIND_* of special s
    const int32_t* obj[2];
    const int32_t* elem[2];
};

enum EmitMode : int32_t {
    EMIT_PROBS = 0,   // out[dest] = p                                   (lane 0 of the wave writes)
    EMIT_FD = 1,      // out[dest*ld + col] = (p - pbase[dest]) / eps    (+ optional raw[dest*ldraw + col] = p)
    EMIT_HESS = 2,    // out[(dest*ld + row)*ld2 + col] = ((p - prow[dest*ldrow + rowidx]) / eps - dcol[dest*lddcol + colidx]) / eps
};

struct WalkArgs {
    // plan (device)
    const uint32_t* prog;
    const int64_t* task_off;
    const int32_t* eff_ptr;
    const int32_t* eff_label;
    const int32_t* eff_dest;
    // model (device)
    const double* gates;     // [nG][D][D] row-major (special-row set-up)
    const double* gates_t;   // [nG][D][D] transposed: gates_t[g][j][i] = gates[g][i][j] (column sweeps)
    const double* rhos;
    const double* effects;
    int32_t n_gates, n_effects;
    const double* base_cache;   // [n_state_ids][D] states of the base pass (read by S>0 passes)
    double* base_cache_w;       // same buffer, written by the S=0 pass at every NODE marker (may be NULL)
    // lanes
    LaneTables lanes;
    int32_t multi_start, start0;   // walk_base_kernel only: > 0 = walk from `multi_start` start vectors (rhos[]), lane group q
                                   // from rhos[start0 + q], states stored at base_cache_w[id][start0 + q][D]
    const uint32_t* block_order;   // optional: blockIdx.x -> task * n_pwaves + pw, expensive pairs first (NULL: identity)
    // persistent launch (launch_walk_persistent): one workgroup per CU; its wavefronts pop (task, wavefront) pairs from
    // per-SIMD queues the host packed to equal estimated work, then take from the other queues
    const int32_t* bin_ptr;        // [n_bins + 1]
    const uint32_t* bin_items;     // pair ids (task * n_pwaves + pw), longest first inside a queue
    uint32_t* bin_head;            // [n_bins] next unpopped position (zeroed before the launch)
    int32_t n_bins;
    int32_t lds_wave_doubles;      // LDS doubles (save slots) per wavefront of the workgroup
    // hand-over of a walk between two SIMDs (persistent launch): queue items carry a part in their top two bits --
    // 1: walk the pair up to its task's split position, store the 64 lane states, raise the flag; 2: wait for the
    // flag, pick the states up, walk the rest (gst::task_split_candidates)
    const int32_t* ho_pc;          // [n_split] where split pair h is cut: word index relative to its task's start
    const int32_t* ho_index;       // [n_tasks * n_pwaves] hand-over buffer of a split pair, or -1
    double* ho_state;              // [n_split][ho_blocks][D][64]: block 0 = the lane states, 1 + s = save slot s
    int32_t* ho_id;                // [n_split] >= 0: the state is the clean base state of that id (nothing stored)
    uint32_t* ho_flag;             // [n_split] zeroed before the launch
    unsigned long long* trace;     // development aid (GST_FD_TRACE): [0] = record count, then (pair, t0, t1, hw_id) records
    // TP POVM complement in the Hessian pass (walk_kernel's COMP): effect comp_index = identity - sum(others)
    int32_t comp_index, n_others;
    const int32_t* comp_others;    // [n_others], the reference's summation order
    const double* comp_identity;   // [D]
    // Whole-model mode of the S = 0 probability pass (gst_fill_dprobs_models: finite differences over ANY
    // parameterisation): n_models > 0 model sets, each laid out [gates_t | rhos | effects] with model_stride doubles
    // between consecutive sets; wavefront (task, m) walks task with set m and writes out[m * out_model_stride + dest].
    int32_t n_models;
    int64_t model_stride, out_model_stride;
    int32_t mm_tasks;        // number of tasks (walk_base_kernel derives (task, model) from blockIdx.x)
    // Fused base lane (launch-bound plans, S = 1): lane 63 of every wavefront carries NO perturbation, i.e. the base
    // model; the quotient uses ITS probability instead of pbase[], nothing is taken from the base-state cache, and the
    // separate base pass is not launched at all.  Lane 63 of parameter wavefront 0 writes probs_out (may be NULL).
    int32_t fused;
    double* probs_out;
    int32_t n_pwaves;        // wavefronts along the parameter dimension (grid.x)
    int32_t rows_S;          // walk_rows_kernel only: number of perturbations per wavefront (0, 1, 2); its
                             // `lanes` tables then hold ONE entry per wavefront instead of one per lane
    // output
    int32_t mode;
    double* out;
    int64_t ld, ld2;
    double eps;
    const double* pbase;     // EMIT_FD: base probabilities [nE]
    double* raw;             // EMIT_FD: optional raw perturbed probabilities [nE][ldraw]
    int64_t ldraw;
    const double* prow;      // EMIT_HESS: probs at theta+eps e_i      [nE][ldrow]
    int64_t ldrow;
    const double* dcol;      // EMIT_HESS: FD dprobs at theta over block 2 [nE][lddcol]
    int64_t lddcol;
    const int32_t* wave_row;     // EMIT_HESS: per wave: dest row (block-1 position) and its index into prow
    const int32_t* wave_rowidx;
    const int32_t* lane_colidx;  // EMIT_HESS: per lane slot: index into dcol's columns
    // (new fields go HERE, at the end: the persistent FD kernel re-reads its arguments from the kernarg segment with wide
    //  scalar loads, and shifting them by 8 bytes -- one int32 inserted mid-struct -- cost 5 % on a 1/8 atom)
    int32_t chain_share;     // walk_base_kernel: this many chain passes run side by side on the device (0 / 1: this one alone)
    // ---- overlap of the base pass with the persistent FD launch (walk_kernel<..., OVL>; gst_chain.hpp) ----------------
    // ovl_n_tasks > 0: the chains (S = 0 walks) of all tasks run INSIDE this launch: task t on workgroup t % gridDim.x, by
    // its wavefront t / gridDim.x (the first, i.e. oldest, wavefronts), before that wavefront starts popping pairs.  They publish states
    // (base_cache) and probabilities (pbase_w == pbase) with write-through stores; both buffers were pre-filled with
    // OVL_SENTINEL64, and a finite-difference walk that reads the sentinel waits for the value (bounded, see abort_flag).
    int32_t ovl_n_tasks;
    int32_t ovl_chain_doubles;   // LDS doubles of one chain's region (behind the wavefronts' save slots)
    double* pbase_w;
    // Bounded waits: a wavefront whose wait (hand-over flag, sentinel) outlasts ~0.1 s -- its producer sits in a workgroup
    // that is not resident, e.g. because another process holds the CUs -- raises *abort_flag and leaves; every wavefront
    // checks the flag when it pops a pair.  The host has ALREADY enqueued stand-by launches of the same work in the safe
    // form (separate base pass, dispatcher-placed FD walk, no hand-over) behind this one, with guard == abort_flag: they
    // leave at once unless the flag is raised.  Nothing hangs and nothing is left half-written.
    uint32_t* abort_flag;
    const uint32_t* guard;
    // hand-overs at positions where save slots are live: bit s of ho_live[h] = slot s travels with split pair h (its tag
    // in ho_tag[h * 4 + s], its data in block 1 + s of the pair's ho_state region of ho_blocks blocks of D * 64 doubles)
    const uint32_t* ho_live;
    int32_t* ho_tag;
    int32_t ho_blocks;
    int32_t ovl_test_skip;       // tests: walk no chain (every consumer's wait then runs out and the stand-by launches take over)
    // Whole-vector preparation perturbations (Lindblad-parameterised preparations, gst_set_lindblad): a GST_KIND_RHO lane's
    // perturbed preparation is rho_models[elem * rho_model_stride ...] (D doubles) instead of rhos[obj] with element `elem`
    // stepped.  NULL: the one-element form.
    const double* rho_models;
    int64_t rho_model_stride;
};

// Launch the walk over all tasks x parameter wavefronts.  S = number of specials per lane (0,1,2);
// n_slots = save slots the programs use (LDS: n_slots * D * 512 bytes per wavefront).
hipError_t launch_walk(int D, int S, const WalkArgs& a, int64_t n_tasks, int n_slots, hipStream_t stream, int split = 1,
                       bool complement = false);

// Analytic Jacobian (gst_kernels_analytic.hip): one wavefront per circuit walks the state-id graph backwards.
struct AnaArgs {
    int64_t n_circuits;
    const int32_t* circ_leaf;     // [n_circuits] state id of the circuit's final state
    const int32_t* node_parent;   // [n_state_ids] (-1: a rho state)
    const int32_t* node_sym;      // [n_state_ids] gate index (rho index for rho states)
    const int32_t* node_run;      // [n_state_ids] number of upward steps whose parent is id-1 (0: rho state or a jump)
    const int32_t* eff_ptr;
    const int32_t* eff_label;
    const int32_t* eff_dest;
    const double* gates_t;        // [nG][D][D] transposed
    const double* effects;
    const double* base_cache;     // forward states of the base pass
    int32_t n_gates, n_rhos, n_effects;
    const int32_t* gate_col0;     // [nG]: >=0 first of D*D consecutive dest columns; -1 use colmap_gate; -2 none requested
    const int32_t* colmap_gate;   // [nG*D*D] dest column or -1
    const int32_t* colmap_rho;    // [nR*D]
    const int32_t* colmap_eff;    // [nEl*D]
    double* out;
    int64_t ld;
    // MFMA path (D = 16): backward states of the reversed plan, pair tables, dynamic work counter
    const double* rev_cache;      // [n_rev_states][n_effects][D]
    const int32_t* rev_leaf;      // [n_circuits] state id of B_0 in the reversed plan
    const int32_t* pair_f;        // state id of F_{k-1}
    const int32_t* pair_r;        // state id of B_k
    const int64_t* pos_ptr;       // [n_circuits * n_gates + 1]
    const int32_t* circ_rho;      // [n_circuits]
    const int32_t* circ_order;    // [n_circuits] circuits sorted by their reversed string (= by rev_leaf)
    const int32_t* circ_partner;  // D = 16: [n_items] second circuit of a work item (same last applications), or -1; NULL: none
    const int32_t* pair_common;   // [n_items * n_gates] applications of each gate in the common tail of a two-circuit item
    const uint32_t* range_begin;  // [9] the suffix-ordered list cut into 8 ranges of equal work (one per XCD)
    // Hessian rows reuse the D = 16 MFMA contraction with derivative states in place of one of the two caches:
    uint32_t fwd_stride, rev_stride;   // bytes between consecutive states of base_cache / rev_cache (0: the plain layouts)
    int32_t rho_zero, eff_zero;        // write zeros to the rho / effect columns instead of B_0 / F_n
    int32_t accumulate;                // add to the output instead of storing
    uint32_t* work_counter;       // [8] one per range, zeroed before the launch
    int32_t group_fetch;          // D = 16: the 4 wavefronts of a workgroup take 4 consecutive items together (see the kernel)
    // D = 16, two-circuit items as one stream of 4-slot blocks (NULL: gate by gate): slot = forward id of the first circuit,
    // of the second (-1: that circuit has no application here) and the backward id; blk_ptr[item * nG + g] = first block of gate g
    const int32_t *blk_f1, *blk_f2, *blk_r, *blk_ptr;
    int32_t wide;                 // a state cache (or a derivative-state cache of a Hessian row) is 4 GB or larger: 64-bit lane offsets
    int32_t zeros_resident;       // D = 16 stream form: the blocks of gates an item never applies already hold zeros in `out`: not stored
    const uint32_t* zeros_ok;     // ... unless this device word (NULL: none) reads 0: a row scaling met a non-finite factor since
};
// The D = 16 contraction over TILES of circuits that share their forward chains row-wise and their backward chains
// column-wise (gst_kernels_tiles.hip; the host finds them: build_tiles in gst_fill_analytic.cpp)
constexpr int TILE_ROWS = 8, TILE_COLS = 4;
struct TileArgs {
    AnaArgs a;                    // caches, column maps, destination, effect CSR, circ_leaf / rev_leaf / circ_rho, zeros_*
    int32_t n_tiles;
    const int32_t* tile_order;    // [n_tiles] launch order (heaviest first)
    const int32_t* tile_cid;      // [n_tiles][TILE_ROWS * TILE_COLS]: circuit of (row, column) at row * TILE_COLS + column, or -1
    const int32_t* tile_blk;      // [n_tiles][n_gates + 1]: the segment blocks of gate g are tile_blk[.. + g] .. tile_blk[.. + g + 1]
    const int32_t* tsf;           // [n_blocks][4 slots][TILE_ROWS]: forward state id per row (-1: dead slot / absent row)
    const int32_t* tsr;           // [n_blocks][4 slots][TILE_COLS]: backward state id per column (-1: dead)
    const int32_t* rem_ptr;       // [n_tiles][n_gates][TILE_ROWS * TILE_COLS + 1]: remnant blocks of (gate, circuit slot)
    const int32_t* rem_f;         // [n_rem_blocks][4]: forward id (-1: dead slot)
    const int32_t* rem_r;         // [n_rem_blocks][4]: backward id (always valid)
    uint32_t* counter;            // one word, zeroed before the launch
    int32_t debug;                // development: 1 = no stores, 2 = no remnants, 4 = no segment stream (GST_TEST_FORCE tile_dbg=)
};
hipError_t launch_analytic_tiles(const TileArgs& t, int n_cus, hipStream_t stream);
hipError_t launch_analytic(int D, const AnaArgs& a, hipStream_t stream);
hipError_t launch_analytic_mfma(const AnaArgs& a, hipStream_t stream);   // D = 16
int analytic_stream_chunks();        // chunks of 4 slots per block of the D = 16 stream (the host pads every gate's slots to whole blocks)
hipError_t launch_analytic_mfma64(const AnaArgs& a, hipStream_t stream); // D = 64 (uses work_counter[0] only)
hipError_t launch_analytic_small(const AnaArgs& a, hipStream_t stream);  // D = 4: same contraction, VALU (no counter)

hipError_t launch_scale_rows(double* J, int64_t n_rows, int64_t n_cols, int64_t ld, const double* w, hipStream_t s);
hipError_t launch_symmetrize(double* C, int64_t n, hipStream_t s);
// Normal equations (gst_kernels_normal.hip): split-K MFMA fp64 J^T J and streaming J^T f
int jtj_num_slabs(int64_t n_rows, int n_cols, int n_cus = 256);
int jtf_num_slabs(int64_t n_rows);
// pmask (may be NULL): per 16-row panel, bit t = the panel holds a non-zero in the 128 columns of tile t
// (launch_jtj_panel_masks; n_cols <= 32 * 128); panels whose two tiles are not both marked are skipped
// w (may be NULL): row weights applied while a panel is staged -- (diag(w) J)^T (diag(w) J) with J left untouched
hipError_t launch_jtj(const double* J, int64_t n_rows, int n_cols, int64_t ld, double* part, int n_slabs, double* C, hipStream_t s,
                      const uint32_t* pmask = nullptr, const double* w = nullptr);
// one streaming pass over J: the panel masks (of diag(w) J when w != NULL), and with write_back the row scaling
// J <- diag(w) J as well
hipError_t launch_jtj_panel_masks(double* J, int64_t n_rows, int n_cols, int64_t ld, const double* w, uint32_t* pmask, hipStream_t s,
                                  bool write_back = true);
// the masks of diag(w) J and (diag(w) J)^T f -> y in one read-only pass (part: [n_ranges][n_cols] scratch)
int jtj_mask_jtf_ranges(int64_t n_rows, int n_cols);
hipError_t launch_jtj_mask_jtf(const double* J, int64_t n_rows, int n_cols, int64_t ld, const double* w, const double* f, uint32_t* pmask,
                               double* part, int n_ranges, double* y, hipStream_t s);
int jtj_mask_tiles(int n_cols);
int64_t jtj_mask_panels(int64_t n_rows);
int64_t jtj_mask_words(int64_t n_rows);      // allocation of a mask buffer: the panel masks + pair counts + ranked pairs
hipError_t launch_jtf(const double* J, const double* f, int64_t n_rows, int n_cols, int64_t ld, double* part, int n_slabs,
                      double* y, hipStream_t s, const double* w = nullptr);

// C[:, colmap[j]] += A[:, a_col0 : a_col0 + K] . B[:, j]  (B row-major [K][n]; colmap[j] < 0: skip) -- MFMA fp64
hipError_t launch_chain_rule_gemm(const double* A, int64_t ldA, int64_t a_col0, int K, const double* B, int n, const int32_t* colmap,
                                  double* C, int64_t ldC, int64_t n_rows, hipStream_t s, bool overwrite = false,
                                  const uint64_t* rowmask = nullptr, int mask_bit = 0);      // rowmask: [ceil(n_rows / 128) * 4] words, see the kernel

// Derivative states for the analytic Hessian (gst_kernels_analytic.hip): lane group q of a wavefront carries
//   dS^{theta_q}: dS_0 = start (a unit vector or 0);  dS_k = G_k dS_{k-1} + [g_k = inj_gate[q]] e_{inj_dst[q]} * S_{k-1}[inj_src[q]]
// over the walk programs of a plan (forward plan + gates_t for dF, reversed plan + row-major gates for dB), S being
// the cached base states of that plan.  D = 4, 16: four thetas per wavefront (lane groups); D = 64: one per wavefront.
struct DWalkArgs {
    const uint32_t* prog;
    const int64_t* task_off;
    const double* tile;          // [nG][D][D], out_i += tile[g][j][i] * v_j
    int32_t n_gates;
    const double* base;          // base states: element idx of state id at base[id * bstride + idx * bmul + boff]
    int64_t bstride, bmul, boff;
    const int32_t* inj_gate;     // [n_theta] gate whose element theta is, or -1
    const int32_t* inj_dst;      // component that receives the injection
    const int32_t* inj_src;      // component of the base state that is injected
    const int32_t* start_obj;    // RHO argument for which the walk starts from e_{start_idx}; -1: any; -2: never (start 0)
    const int32_t* start_idx;
    int32_t n_theta;
    double* out;                 // dS of state id, theta slot q, component i at out[(id * 4 + q) * ostride + i * omul + ooff]
    int64_t ostride, omul, ooff;
};
hipError_t launch_dwalk(int D, const DWalkArgs& a, int64_t n_tasks, int n_slots, hipStream_t stream);   // D = 16, 64

// Objective Hessian block from device-resident hprobs / dprobs blocks and the objective's dterms / hterms.
hipError_t launch_objective_coeffs(int kind, const double* probs, const double* counts, const double* totals, int64_t n, double min_p,
                                   double radius, double* dterms, double* hterms, hipStream_t s);
// Chain rule of a Hessian block for linear parameterisations: out[e][dest1[i]][dest2[j]] =
// sum over (a, w1) in column i of W1 and (b, w2) in column j of W2 of w1 * w2 * H[e][a][b]   (W1, W2 in CSC form,
// rows = positions in the element-Hessian block H [nE][m1][m2])
hipError_t launch_hessian_chain_rule(const double* H, int64_t nE, int m1, int m2, const int32_t* ptr1, const int32_t* row1,
                                     const double* w1, const int32_t* dest1, int n1, const int32_t* ptr2, const int32_t* row2,
                                     const double* w2, const int32_t* dest2, int n2, double* out, int64_t ld1, int64_t ld2,
                                     hipStream_t s);
int hessian_block_slabs(int64_t nE, int n1, int n2);
hipError_t launch_hessian_block(const double* H, const double* d1, const double* d2, const double* dco, const double* hco, int64_t nE,
                                int n1, int n2, double* part, int n_slabs, double* out, hipStream_t s);

// Element-wise objective maps; `part` receives n_blocks partial sums of the terms.
hipError_t launch_objective_rows(int kind, double* probs, const double* counts, const double* totals, int64_t n, double min_p,
                                 double radius, double clip_lo, double clip_hi, double* lsvec, double* rowscale,
                                 double* terms_out, double* part, int n_blocks, hipStream_t s);

// D = 64 derivative passes run NWB = rows_group(D, n_slots) models per workgroup (walk_rows_shared_kernel); a block_order
// for launch_walk_rows then indexes (task, group of NWB parameter wavefronts) pairs.  1 when not applicable.
int rows_group(int D, int n_slots);
// D <= 16: whether the chain kernel (walk_base_kernel: gates, effects, emit ring and program window in LDS) can run this
// model -- it is the only kernel with multi-start (backward, one lane group per effect) walks
bool chain_kernel_fits(int D, int n_gates, int n_effects, int n_slots);
size_t chain_lds_doubles(int D, int n_gates, int n_effects, int n_slots);   // LDS of one single-wavefront chain (tables + private region)

// Row-per-lane variant (gst_kernels_rows.hip): one model per wavefront; D = 4, 16 or 64; S = a.rows_S.
// FD columns of effect parameters evaluated on the cached final states (TP POVMs: see effect_fd_kernel)
struct EffectFDArgs {
    int64_t n_circuits;
    int32_t n_cols, D, comp_index;
    const int32_t* circ_leaf;
    const int32_t* eff_ptr;
    const int32_t* eff_label;
    const int32_t* eff_dest;
    const double* effects;
    const double* base_cache;
    const double* pbase;
    const int32_t* col_obj;           // [n_cols] effect the column's parameter belongs to
    const int32_t* col_elem;          // [n_cols] its component
    const int32_t* col_dest;          // [n_cols] output column
    const int32_t* col_touches_comp;  // [n_cols] the effect is one of the complement's "other" effects
    const double* col_own;            // [n_cols] perturbed component of the effect itself
    const double* col_comp;           // [n_cols] the same component of the recomputed complement
    double* out;
    int64_t ld;
    double* raw;
    int64_t ldraw;
    double eps;
};
hipError_t launch_effect_fd(const EffectFDArgs& a, hipStream_t stream);

// Dense members from Lindblad parameters (gst_kernels_lindblad.hip): member m = (static factor) composed with
// exp(L_m), L_m = sum_k Re(c_k) term_re[term_off[m] + k] + Im(c_k) term_im[...], c = the member's coefficient blocks
// evaluated at theta[param0[m] .. + n_params[m]).  Block types 0 'ham', 1 'other_diagonal', 2 'other'; modes 0 'elements',
// 1 'cholesky' (lindbladcoefficients.py).  set_param == NULL: workgroup m builds member m into `sets` (the base model,
// one set); otherwise workgroup s builds set s = the base model with parameter set_param[s] stepped by eps.
// A set is [gates_t (n_gates D D) | rhos (n_rhos D) | effects (n_effects D)], set_stride doubles apart.
constexpr int LB_MAX_BLOCKS = 4;
constexpr int lb_max_coeffs(int D) { return LB_MAX_BLOCKS * (D - 1) * (D - 1); }   // basis size of n qubits: 4^n - 1 = D - 1
struct LbArgs {
    int32_t n_members, n_gates, n_rhos, n_effects;
    const int32_t *kind, *obj, *n_eff, *n_params, *n_blocks, *blk_type, *blk_mode, *blk_n;   // blk_*: [n_members][LB_MAX_BLOCKS]
    const int64_t *param0, *term_off, *static_off;
    const double *theta, *term_re, *term_im, *statics;
    const double* base_set;        // the base model in set layout (set_param != NULL)
    const int64_t* set_param;      // [n_sets] global parameter stepped in each set, or NULL (base build)
    double* sets;
    int64_t set_stride;
    double* gates_rowmajor;        // base build only (may be NULL): gates[n_gates][D][D] row-major as well
    double eps;
    double* deriv_out;             // launch_lindblad_derivs: the members' derivative matrices, member m at deriv_off[m]
    const int64_t* deriv_off;
    int32_t member_only;           // set_param != NULL: write ONLY the changed member, at the start of its set (gate: transposed
                                   // [D][D]; state [D]; POVM [n_eff][D]) -- what walk_pert_kernel reads; no copy of the base model
};
hipError_t launch_lindblad_build(int D, const LbArgs& a, int64_t n_sets, hipStream_t stream);
// d(dense member)/d(parameter set_param[b]) for workgroup b: column (set_param[b] - param0[m]) of member m's row-major
// [n_elem][n_params[m]] matrix at deriv_out + deriv_off[m] (a POVM: n_eff matrices [D][n_params] one after another)
hipError_t launch_lindblad_derivs(int D, const LbArgs& a, int64_t n_params_total, hipStream_t stream);
// D = 64 (gst_kernels_lindblad.hip): base members through a per-member workspace (scaled generator, Taylor terms, squares)
// that the derivative kernel of the same parameter vector re-uses
size_t lindblad64_workspace_doubles(int n_members);
hipError_t launch_lindblad64_build(const LbArgs& a, double* ws, int max_member_params, hipStream_t stream);
hipError_t launch_lindblad64_derivs(const LbArgs& a, const double* ws, int64_t n_params_total, hipStream_t stream);

// Layer operations of implicit models (gst_kernels_composite.hip): layer g = Emb(f_{n-1}) ... Emb(f_0) over its factors
// gate_fptr[g] .. gate_fptr[g + 1]; factor f embeds leaf factor_leaf[f] (leaf_dim x leaf_dim, row-major values at
// leaf_values + leaf_off, the model parameter of each element in leaf_param, -1 = none) on the qubits factor_targets[f][0..2]
// (-1 = unused; qubit 0 = most significant base-4 digit of the state index).
struct CompositeArgs {
    int32_t D, nq, n_gates, n_rhos, n_effects, n_leaves, max_leaf_dim;
    const int32_t *leaf_dim;
    const int64_t *leaf_off, *leaf_param;
    const double* leaf_values;
    // general leaves (any member the host can densify): leaf_np[l] > 0 parameters, listed ascending at leaf_plist +
    // leaf_plist_off[l]; their derivative matrices [dl*dl][np] at leaf_deriv + leaf_deriv_off[l]; their values after each
    // parameter's finite-difference step [np][dl*dl] at leaf_fd + leaf_fd_off[l] (NULL: not supplied).  leaf_np NULL: none.
    const int32_t* leaf_np;
    const int64_t *leaf_plist_off, *leaf_plist, *leaf_deriv_off, *leaf_fd_off;
    const double *leaf_deriv, *leaf_fd;
    const int32_t *gate_fptr, *factor_leaf, *factor_targets;
    const double *rhos, *effects;            // the base SPAM vectors
    const int32_t *pkind, *pobj, *pelem;     // the plan's parameter map (SPAM columns of the model sets), or NULL
    const int64_t* set_param;                // [n_sets]: the parameter stepped in each set, or NULL (base build)
    const double* base_set;                  // the base model in set layout (layers a step does not move are copied), or NULL
    double eps;
    double* sets;                            // [n_sets][set_stride]: [gates_t | rhos | effects]
    int64_t set_stride;
    double* gates_rowmajor;                  // base build only (may be NULL)
    // launch_composite_derivs: work item = one column of one layer's derivative matrix
    const int32_t *item_gate, *item_col;
    const int64_t *item_param;
    const int32_t* gate_ncols;               // [n_gates]
    const int64_t* gate_doff;                // [n_gates]: offset of layer g's [D*D][ncols] matrix in deriv_out
    double* deriv_out;
};
hipError_t launch_composite_build(const CompositeArgs& a, int64_t n_sets, hipStream_t stream);
hipError_t launch_composite_derivs(const CompositeArgs& a, int64_t n_items, hipStream_t stream);

// Finite-difference walks for columns that change a whole object (gst_kernels_pert.hip): a work item is a task's dirty
// program for the member's class walked for the 64/D columns col0 .. col0 + ncols of ONE member (kind / obj / n_eff of
// parameter wavefront pw); column c's changed member sits at pert + c * pert_stride (layout as LbArgs::member_only), its
// Jacobian column is col_dest[c].
struct PertArgs {
    const uint32_t* prog;                          // the DIRTY programs (gst::build_dirty_programs), concatenated
    const int64_t* prog_off;                       // [n_tasks * n_classes + 1]
    const uint32_t* item_prog;                     // [n_items] work item -> program index (task * n_classes + class), expensive first
    const int32_t* item_pw;                        // [n_items] work item -> parameter wavefront
    const int32_t *eff_ptr, *eff_label, *eff_dest;
    const double *gates_t, *rhos, *effects;       // the base model
    int32_t n_gates, n_effects;
    const double* base_cache;                      // states of the base pass
    const double* pbase;                           // base probabilities
    const double* pert;
    int64_t pert_stride;
    const int32_t *wave_kind, *wave_obj, *wave_neff, *wave_col0, *wave_ncols;   // [n_pwaves]
    const int32_t* col_dest;
    int32_t n_pwaves;
    double* out;
    int64_t ld;
    double eps;
};
bool pert_kernel_fits(int D, int n_gates, int n_effects, int n_slots);
hipError_t launch_walk_pert(int D, const PertArgs& a, int64_t n_items, int n_slots, hipStream_t stream);
// out[e * ld + (dense0 >= 0 ? dense0 + c : col_dest[c])] = 0 for e < nE, c < n_cols
hipError_t launch_zero_columns(double* out, int64_t ld, int64_t nE, const int32_t* col_dest, int32_t n_cols, int32_t dense0, hipStream_t stream);
// the columns col0 .. col0 + ncols of ONE POVM member (effects obj .. obj + n_eff), from the circuits' final base states
hipError_t launch_effect_columns(int D, const PertArgs& a, const int32_t* circ_leaf, int64_t n_circuits, int32_t obj, int32_t n_eff,
                                 int32_t col0, int32_t ncols, hipStream_t stream);

// Persistent form of the D <= 16 FD walk: n_wg workgroups of 16 wavefronts (see WalkArgs::bin_ptr)
// a.ovl_n_tasks > 0 selects the overlap form (D = 16): LDS = wavefronts' save slots + chains_per_wg chain regions
hipError_t launch_walk_persistent(int D, const WalkArgs& a, int n_wg, int n_slots, hipStream_t stream);
int persistent_waves(int D);                 // wavefronts per workgroup of the persistent launch
hipError_t launch_walk_rows(int D, const WalkArgs& a, int64_t n_tasks, int n_slots, hipStream_t stream);
// the S = 0 walk of a D = 64 plan on the matrix cores (gst_kernels_chain64.hip): modes without an ordering contract only
hipError_t launch_chain64(const WalkArgs& a, int64_t n_tasks, int n_slots, hipStream_t stream);
bool chain64_fits(int nv, int n_slots);
// the multi-start walk with every gate resident in registers (<= 10 gates), two tasks interleaved per workgroup: persistent, a.bin_head = zeroed counter,
// a.block_order = task order (longest first) or NULL
bool chain64_resident_fits(int nv, int n_slots, int n_gates);
hipError_t launch_chain64_resident(const WalkArgs& a, int64_t n_tasks, int n_slots, int n_cus, hipStream_t stream);
// out[e * ld + dest[m]] = (raw[m * raw_stride + e] - pbase[e]) / eps for m < n_models (dest == NULL: column m0 + m):
// the finite-difference quotient of gst_fill_dprobs_models, transposed through LDS tiles (correctly rounded division)
hipError_t launch_fd_from_models(const double* raw, int64_t raw_stride, const double* pbase, int64_t nE, int32_t n_models,
                                 const int32_t* dest, int32_t m0, double eps, double* out, int64_t ld, hipStream_t stream);

// H[(e * ld1 + row) * ld2 + dest2[j]] = (Ji[e * n2 + j] - J0[e * n2 + j]) / eps: one row of an FD-of-FD Hessian block composed
// from two FD Jacobians, as MapForwardSimulator._mapfill_hprobs_atom composes it (mapforwardsim.py:432-436)
hipError_t launch_hess_compose(const double* Ji, const double* J0, int64_t nE, int32_t n2, double eps, double* H, int64_t ld1, int64_t ld2,
                               int64_t row, const int32_t* dest2, hipStream_t stream);


// ---- log-depth evaluation of the state tries (gst_levels.hpp, gst_kernels_levels.hip) ---------------------------------
constexpr int32_t LV_KIND_ROWS_ = 0, LV_KIND_MAT_ = 1, LV_BMAT_IDENT_ = -1;       // (= gst_levels.hpp's LV_KIND_* / LV_BMAT_IDENT)
struct LevelArgs {
    const int32_t* words;      // level programs of all tasks (LevelProgram::words)
    const int64_t* task_off;   // [n_tasks + 1]
    const int32_t* ids;        // pool of source / destination state ids
    const double* bmats;       // [n_gates][16][16]: the gates in ROW form (forward: transposed table; backward: row-major table)
    const double* starts;      // start vectors [start][vector][16] (forward: the state preparations; backward: the effects)
    double* cache;             // [n_states][16][nv]
    double* mats;              // [n_tasks][max_mats][16][16] scratch (germ products and their squarings)
    int32_t nv, max_mats;
    const int64_t* task_ids_off;   // [n_tasks + 1] (LevelProgram::task_ids_off)
    int32_t stage_lds, lds_ints;   // 1: a task's words + ids (at most lds_ints of them) are staged in LDS
};
hipError_t launch_level_pass(const LevelArgs& a, int64_t n_tasks, hipStream_t stream);
hipError_t launch_probs_from_cache(const double* cache, const int32_t* circ_leaf, const int32_t* eff_ptr, const int32_t* eff_label,
                                   const int32_t* eff_dest, const double* effects, int64_t n_circuits, int D, double* out, hipStream_t stream);

}  // namespace gst
