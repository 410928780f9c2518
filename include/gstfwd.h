/*
 * gstfwd.h -- C ABI of the MI355X-native dense-matrix forward simulator (libgstfwd.so).
 *
 * This is the drop-in boundary for ONE hot path of pyGSTi: circuit-outcome probabilities and their
 * Jacobian / Hessian blocks for `densitymx` models,
 *     ForwardSimulator.bulk_fill_probs / bulk_fill_dprobs / bulk_fill_hprobs
 *     (reference: pygsti/forwardsims/forwardsim.py:584,628,701), reached per layout atom through
 *     MapForwardSimulator._bulk_fill_{probs,dprobs,hprobs}_atom (pygsti/forwardsims/mapforwardsim.py:372-391).
 * Today that seam is a Cython cimport of C++ classes (mapforwardsim_calc_densitymx.pyx:15-17); the
 * functions below are what a ctypes/cffi binding of the same seam would bind.  Plain C, plain
 * pointers and sizes, no framework types.  Every function returns 0 on success or a negative
 * GST_E* code; gst_last_error() gives the message (thread-local).  Nothing here aborts the process
 * and nothing here computes on the CPU: if no gfx950 device/HIP runtime is usable, the fill calls
 * fail with GST_ENODEVICE.
 *
 * Ownership: the caller owns every host array for the duration of the call only; the library owns
 * the plan handle and all device memory.  Calls on one plan are not re-entrant; different plans may
 * be used from different threads.  Host-output calls are blocking; `_dev` calls enqueue on the
 * plan's stream and return (use gst_sync()).
 */
#ifndef GSTFWD_H
#define GSTFWD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* The library is built with -fvisibility=hidden: these entry points are its whole dynamic symbol table
 * (tests/test_abi_and_rules.py compares `nm -D` with this header, both ways). */
#define GST_API __attribute__((visibility("default")))

#define GST_OK 0
#define GST_EINVAL (-1)     /* bad argument / inconsistent plan description */
#define GST_ENODEVICE (-2)  /* no usable HIP device (the library never falls back to the CPU) */
#define GST_EHIP (-3)       /* a HIP runtime call failed */
#define GST_ENOMEM (-4)     /* host or device allocation failed (Python shell raises MemoryError) */
#define GST_ESTATE (-5)     /* call order violated (e.g. fill before gst_set_model) */
#define GST_EUNSUPPORTED (-6)

/* parameter kinds of gst_set_param_map (one parameter <-> one dense element: the reference's
 * `full` parameterisation, modelmembers/operations/fullarbitraryop.py:139-164) */
#define GST_KIND_NONE (-1)  /* parameter of an object this plan never applies: derivative exactly 0 */
#define GST_KIND_GATE 0
#define GST_KIND_RHO 1
#define GST_KIND_EFFECT 2

/* derivative modes of gst_fill_dprobs */
#define GST_DERIV_FD 0        /* forward finite differences, bit-for-bit the reference's Map path
                                 (mapforwardsim_calc_densitymx.pyx:290-383) */
#define GST_DERIV_ANALYTIC 1  /* exact derivative (what MatrixForwardSimulator computes,
                                 matrixforwardsim.py:1059-1140): forward states x backward effect vectors;
                                 `eps` is ignored.  D = 4 and 16. */

typedef struct gst_plan gst_plan;

/*
 * Plan description in the reference's own per-atom format (what _MapCOPALayoutAtom holds,
 * pygsti/layouts/maplayout.py:101-134, after the int conversion of convert_maplayout,
 * mapforwardsim_calc_densitymx.pyx:55-77):
 *   row k of the prefix table: expanded circuit t_dest[k] = cached state t_start[k] (or, when
 *   t_start[k] == -1, state preparation t_rho[k]) followed by gate_idx[row_ptr[k]..row_ptr[k+1]);
 *   the resulting state is kept in cache slot t_cache[k] (-1: not cached).
 *   expanded circuit i yields elements eff_dest[eff_ptr[i]..eff_ptr[i+1]) (indices into the atom's
 *   slice of the 'e' array) using effect vectors eff_label[...] (elbl_indices_by_expcircuit /
 *   elindices_by_expcircuit).
 * The library re-derives every full circuit from the table, builds its own prefix trie and walk
 * programs from them (the table's caching choices do not constrain the device schedule) -- results
 * are identical because each state is still the same left-to-right product.
 */
typedef struct {
    int32_t D;           /* state-vector length: 4 / 16 / 64 (one to three qubits) run natively; any other 2 .. 64 (a qutrit's 9 in the
                            Gell-Mann basis, a qubit with a leakage level ...) runs zero-padded at the next of those -- every array of
                            this interface keeps the CALLER's D (gates [n][D][D], element indices i * D + j ...), results are those of the
                            un-padded sums up to the sign of an exact zero (the padded terms are + 0.0) */
    int32_t n_gates;     /* number of distinct layer operators (atom.op_labels) */
    int32_t n_rhos;      /* atom.rho_labels */
    int32_t n_effects;   /* atom.full_effect_labels */
    int32_t n_rows;      /* table rows == expanded circuits */
    int32_t cache_size;  /* atom.cache_size (only used for validation) */
    int64_t n_elements;  /* atom.num_elements */
    const int32_t *t_dest, *t_start, *t_cache, *t_rho;   /* [n_rows] */
    const int64_t *row_ptr;                              /* [n_rows+1] */
    const int32_t *gate_idx;                             /* [row_ptr[n_rows]] */
    const int64_t *eff_ptr;                              /* [n_rows+1] */
    const int32_t *eff_label, *eff_dest;                 /* [eff_ptr[n_rows]] */
} gst_table_desc;

/* The same information as raw circuits (no prefix table): circuit i = rho circ_rho[i] followed by
 * circ_gates[circ_ptr[i]..circ_ptr[i+1]); effect CSR as above, indexed by circuit. */
typedef struct {
    int32_t D, n_gates, n_rhos, n_effects;
    int32_t n_circuits;
    int64_t n_elements;
    const int32_t *circ_rho;     /* [n_circuits] */
    const int64_t *circ_ptr;     /* [n_circuits+1] */
    const int32_t *circ_gates;
    const int64_t *eff_ptr;      /* [n_circuits+1] */
    const int32_t *eff_label, *eff_dest;
} gst_circuits_desc;

typedef struct {
    int32_t device;          /* HIP device ordinal; -1 = current device */
    int32_t target_tasks;    /* 0 = default; how many independent walk programs to aim for */
    int32_t max_slots;       /* 0 = default; save slots (LDS-resident states) a walk program may use */
    int32_t fd_split;        /* 0 = auto; 1, 2 or 4 wavefronts share one (task, 64 columns) pair of the FD Jacobian
                                (rows of every mat-vec split over them; same results, finer scheduling granularity) */
    int32_t timing;          /* HIP events behind gst_stats.last_kernel_ms / last_total_ms: 0 = auto (recorded unless the plan
                                is launch-bound: <= 65,536 states, where six event records were a third of a fill),
                                1 = always, 2 = never */
    int32_t reserved[3];
} gst_options;

typedef struct {
    int64_t n_circuits, n_elements;
    int64_t sum_depth;         /* gate applications without any prefix sharing */
    int64_t trie_nodes;        /* distinct prefixes (states) over all circuits */
    int64_t applies_per_pass;  /* gate applications one pass actually executes (incl. task-prefix recompute) */
    int64_t n_tasks;
    int64_t prog_words;
    int32_t max_slots;         /* deepest save-slot index any walk program uses, plus one */
    int32_t max_depth;
    double last_kernel_ms;     /* device time of the dominant kernel of the last fill (HIP events) */
    double last_total_ms;      /* device time of the whole last fill */
    int64_t last_launches;
    int32_t last_fd_form;      /* last finite-difference fill: 0 = one workgroup per (task, 64 columns) pair; 1 = persistent
                                  per-SIMD queues; 2 = persistent with the base pass walked inside the same launch */
    int32_t last_fd_aborted;   /* 1: a bounded wait of that persistent launch ran out (producer workgroup not resident, e.g. a
                                  shared device) and the stand-by launches behind it produced the result instead */
    int32_t last_levels;       /* 1: the last fill took its forward states from the log-depth level pass (GST_OPT_FAST_CHAINS /
                                  GST_OPT_FAST_PROBS), 0: from the sequential walk */
    int32_t last_zeros_resident;   /* 1: the last exact Jacobian fill did not re-store the destination's structural zeros */
    int32_t last_tiles;            /* tiles the last exact Jacobian fill contracted on the tile kernel (0: item kernel only) */
    int32_t lm_graph_replays;      /* gst_lm_step_dev: iterations served by replaying the captured HIP graph so far (-1: a capture was refused) */
    int64_t last_tiled_circuits;   /* tiles: the circuits they held */
} gst_stats;

GST_API int gst_plan_create_from_table(const gst_table_desc *desc, const gst_options *opt, gst_plan **out);

/* Run-time options of a plan.
 *   GST_OPT_ANALYTIC_KEEP_ZEROS (value 0 / 1 / 2): the structural zeros of GST_DERIV_ANALYTIC Jacobians (D = 16).
 *       Entry (element of circuit c, parameter of gate g) is an exact zero when c never applies g -- 30 % of a GST Jacobian.
 *       A fill whose destination, leading dimension and column request equal an earlier exact fill's need not store those
 *       zeros again (a third of the contraction's stores) if nothing else was written there in between.
 *       2 (default): only for destinations the library can vouch for -- memory the caller allocated as TRACKED
 *          (gst_device_malloc_tracked: an explicit statement per allocation) and the plan's own staging buffer behind
 *          host destinations; plain gst_device_malloc memory and foreign pointers are NOT eligible.  Every entry point that writes device memory (fills, gst_memcpy_h2d,
 *          gst_copy_block_dev, the objective maps, the gst_comm_* collectives) invalidates what it overwrites;
 *          gst_fill_jtj_dev's in-place row scaling keeps zeros zero and is checked for non-finite factors on the stream.
 *          A caller that writes tracked memory with ITS OWN kernels announces it with gst_device_touch; a row scaling
 *          issued through ANOTHER plan than the one that filled the Jacobian ends the claim (different streams).
 *          (gst_stats.last_zeros_resident reports the decision of the last fill.)
 *       1: additionally for ANY destination pointer: the caller promises that nothing but row scalings was written into
 *          the buffer since the previous exact fill of this plan -- what an optimizer does that reuses one device Jacobian
 *          every iteration.  The first fill into a destination (and any fill after the destination, the columns or the
 *          option changed) writes everything.
 *       0: every fill stores every entry. */
#define GST_OPT_ANALYTIC_KEEP_ZEROS 1
/*   GST_OPT_FAST_CHAINS (value 0 / 1 / 2; D = 16 and 64): how the modes WITHOUT an ordering contract -- GST_DERIV_ANALYTIC, and
 *       gst_fill_probs* under GST_OPT_FAST_PROBS -- obtain the states of the circuit tries.  The sequential walk applies a
 *       GST family's ~1,150 gates one after another (0.5 ms of latency on any hardware; it IS the finite-difference mode's
 *       bit-parity contract and stays there).  The level pass finds the periodic (germ-power) paths of the tries, forms the
 *       germ's matrix and its squarings on the matrix cores and evaluates every path in ~2 m + 15 dependent stages (the
 *       reference's Matrix simulator multiplies sub-products over an eval tree for the same reason, matrixforwardsim.py:
 *       675-727, evaltree.py:31-189).  Results differ from the walk's by re-association only (<= 1e-13 observed at depth
 *       1,030; the mode's bars are 1e-10 for probabilities, 1e-8 for derivatives).  1 (default): where the stages are few
 *       against the chains; 0: never; 2: whenever a level program exists (tests).  D = 64 has no level programs: any
 *       non-zero value sends these modes' walks to the matrix cores instead (one workgroup per task, the state a
 *       [start vectors][64] row block, gst_kernels_chain64.hip), again equal up to re-association; 0 keeps the row kernel.
 *   GST_OPT_FAST_PROBS (value 0 / 1; default 0): gst_fill_probs / gst_fill_probs_dev through the level pass (D = 16) or the
 *       matrix-core walk (D = 64): probabilities
 *       within 1e-10 of the reference's, NOT bit-identical -- for callers (line searches of an optimizer) that do not
 *       difference them.  Finite-difference fills never use it. */
#define GST_OPT_FAST_CHAINS 2
#define GST_OPT_FAST_PROBS 3
/*   GST_OPT_ANALYTIC_TILES (value 0 / 1; default 0; set before the plan's first exact fill): the D = 16 exact contraction
 *       over TILES of the design's product structure -- circuits prep_i . W . meas_m read the same forward states whatever m and
 *       the same backward states whatever i; a workgroup stages them once in LDS for 8 x 4 circuits (gst_kernels_tiles.hip,
 *       gst_get_tile_stats) -- instead of one or two circuits per wavefront.  Same values to re-association (1e-12).  OFF by
 *       default: measured on the 2Q L <= 1024 design it removes the L1 miss stalls of the item kernel (its block stream runs at
 *       0.77 of the fp64 matrix pipes) but serialises what the item kernel's twelve wavefronts per CU overlap -- 2.08 + 0.70 ms
 *       (tiles + the circuits no tile holds) against 2.47 ms. */
#define GST_OPT_ANALYTIC_TILES 4
GST_API int gst_set_option(gst_plan *plan, int32_t option, int64_t value);
GST_API int gst_plan_create_from_circuits(const gst_circuits_desc *desc, const gst_options *opt, gst_plan **out);
GST_API int gst_plan_destroy(gst_plan *plan);

/* Dense model arrays (what the reference's reps borrow: OpRepDenseSuperop.base, StateRepDense.data,
 * EffectRepConjugatedState.state_rep.data; evotypes/densitymx/opreps.pyx:75-90), C-contiguous f64:
 * gates[n_gates][D][D], rhos[n_rhos][D], effects[n_effects][D].  Re-call after every
 * model.from_vector(); costs one small H2D copy. */
GST_API int gst_set_model(gst_plan *plan, const double *gates, const double *rhos, const double *effects);

/* Parameter p perturbs element `elem[p]` (flat, row-major) of object `obj[p]` of kind `kind[p]`. */
GST_API int gst_set_param_map(gst_plan *plan, int32_t n_params, const int32_t *kind, const int32_t *obj,
                      const int32_t *elem);

/* "full TP" POVMs in the finite-difference modes.  The last effect of a TPPOVM is not parameterised: it is
 * identity - sum(other effects), recomputed whenever one of the others changes
 * (pygsti/modelmembers/povms/complementeffect.py:72-78, called from TPPOVM.from_vector), so the FD step of an
 * effect parameter also moves the complement's outcome.  Declares effect `comp_index` to be such a complement of
 * effects others[0..n_others) (in the reference's summation order) with the given identity vector [D].  While set,
 * GST_DERIV_FD columns of GST_KIND_EFFECT parameters are evaluated on the cached final states with the complement
 * recomputed the way the reference does it (bit-identical), and so are gst_fill_hprobs' FD-of-FD blocks (D = 4, 16:
 * the complement is re-derived after each of the two steps, as MapForwardSimulator._bulk_fill_hprobs_atom's
 * from_vector / set_parameter_value do); no parameter may map to the complement itself; the GST_DERIV_ANALYTIC element
 * map returns GST_EUNSUPPORTED (use gst_set_derivs for exact derivatives of such models).  TPState / FullTPOp need nothing beyond gst_set_param_map: their parameters are
 * plain dense elements.  comp_index < 0 clears.  One complement per plan (one POVM per atom alphabet). */
GST_API int gst_set_complement_effect(gst_plan *plan, int32_t comp_index, const double *identity, int32_t n_others,
                              const int32_t *others);

/* General parameterisations (TP, CPTP, ... -- anything whose members answer deriv_wrt_params) in GST_DERIV_ANALYTIC:
 * what MatrixForwardSimulator assembles from `_doperation` = member.deriv_wrt_params()
 * (pygsti/forwardsims/matrixforwardsim.py:126-190; modelmembers/.../deriv_wrt_params).  Object o = (kind[o], obj[o])
 * depends on n_cols[o] model parameters, listed in param_idx (concatenated over objects, = the member's gpindices), and
 * deriv holds d(dense element)/d(parameter), row-major [n_elem][n_cols[o]] per object (n_elem = D*D for gates, D for
 * rhos and effects), concatenated.  While set, gst_fill_dprobs(_dev)(GST_DERIV_ANALYTIC) returns
 *     d p / d parameter = sum over objects of (d p / d element) . deriv
 * for parameters 0 .. n_params-1 (the (kind,obj,elem) map of gst_set_param_map is not used in that mode; a parameter
 * may appear in several objects, e.g. a TP POVM's complement effect).  Re-call after gst_set_model whenever the
 * derivatives depend on the parameter values (non-linear parameterisations).  n_objs = 0 clears. */
GST_API int gst_set_derivs(gst_plan *plan, int32_t n_params, int32_t n_objs, const int32_t *kind, const int32_t *obj,
                   const int32_t *n_cols, const int64_t *param_idx, const double *deriv);

/* Lindblad-parameterised models (CPTPLND, GLND, H+S ...: what `target_model('CPTPLND')` and StandardGST's default modes
 * build) with the members built ON THE DEVICE from the parameter vector -- SURVEY 8(f) row f4.  In the reference every
 * member is a static factor composed with an exponentiated Lindblad error generator,
 *     gate  ComposedOp([static U, ExpErrorgenOp(L)])        dense = expm(L) . U        (modelmembers/operations/composedop.py,
 *     prep  ComposedState(static rho0, ExpErrorgenOp(L))     dense = expm(L) . rho0      experrorgenop.py:49-213,
 *     POVM  ComposedPOVM(ExpErrorgenOp(L), base POVM)        effect_i = expm(L)^T e_i    states/composedstate.py, povms/composedpovm.py)
 * and L is linear in complex coefficients c, L = Re(sum_k c_k S_k) = sum_k Re(c_k) term_re[k] + Im(c_k) term_im[k]
 * (lindbladerrorgen.py:658-742; S_k = the blocks' Lindblad term superoperators, lindbladcoefficients.py:664-702, 1073-1198 --
 * the caller passes term_re = Re S_k, term_im = -Im S_k, row-major [D][D] each, in the basis the model's dense members use),
 * with c the concatenated block_data of the member's coefficient blocks, functions of its real parameters
 * (lindbladcoefficients.py:164-470):
 *     block_type 0 'ham', 1 'other_diagonal': n coefficients from n parameters v:  mode 0 'elements' c = v, 1 'cholesky' c = v^2
 *     block_type 2 'other': n*n coefficients (row-major) from n*n parameters read as a matrix p:
 *         mode 1 'cholesky' c = C C^dag, C lower triangular, C_ii = p_ii, C_ij = p_ij + i p_ji (i > j)
 *         mode 0 'elements' c Hermitian, c_ii = p_ii, c_ij = p_ij + i p_ji (i > j)
 * A member's parameters are model parameters param0 .. param0 + (sum of its blocks' parameter counts), blocks in order.
 * Members that share a basis share their terms: give them the same term_offset.  EVERY object of the plan must belong
 * to a member (kind GST_KIND_EFFECT = a POVM: the n_eff consecutive effects from index obj, static_part = their base effect
 * vectors [n_eff][D]; gates: static_part = U [D][D]; preparations: rho0 [D]).  D = 4, 16.
 *   gst_set_lindblad         describes the members (copied); n_members = 0 clears.
 *   gst_set_lindblad_params  takes the model's parameter vector theta[n_params] -- call it where gst_set_model would be
 *       called, after every model.from_vector(): the device assembles L, exponentiates (scaled Taylor series, fp64) and
 *       composes every member; the result is the plan's model for ALL fills (gst_get_model reads it back).
 * While set, GST_DERIV_FD columns of gst_fill_dprobs(_dev) are those of mapfill_dprobs_atom for such a model
 * (mapforwardsim_calc_densitymx.pyx:349-381: set_parameter_value(i, theta_i + eps), re-propagate, (p2 - p) / eps): the
 * device builds the dense model after every step itself -- no host densification and no PCIe traffic per column.
 * THAT ROUTE IS OFFERED ONLY WHERE IT MEETS THE 1e-8 PARITY BAR: plans whose deepest circuit has at most
 * GST_LINDBLAD_FD_MAX_DEPTH gate applications.  The device exponentiates by a scaled Taylor series, the reference by
 * scipy's Pade approximant; the perturbed members differ in their last bits (~1e-16), and a finite-difference quotient
 * amplifies that by (occurrences of the member in the circuit) / eps -- measured against the Map simulator: <= 6.9e-9 to
 * depth 16, 5e-8 at depth 41-80, 1.2e-7 at depth 81-160 (profiles/r04_cptplnd_depth_profile_*.json).  No implementation can
 * do better short of reproducing the LAPACK solve inside scipy's expm bit for bit.  On a deeper plan the FD request fails
 * with GST_EUNSUPPORTED (round 6; until then the header carried the envelope as a warning) and the caller takes
 *   - GST_DERIV_ANALYTIC (below): exact, 1e-11 at depth 1,030, faster than the FD walk -- what the adapter's
 *     derivative_mode="auto" has always selected for these models; or
 *   - gst_fill_dprobs_models on a plain plan (gst_set_lindblad cleared, gst_set_model with the host's dense members): the
 *     walk over the REFERENCE's own perturbed dense members, 4e-9 flat in depth -- what the adapter's "fd" does at depth.
 * GST_DERIV_ANALYTIC columns are exact: the device computes every member's d(dense)/d(parameter) itself -- the Frechet
 * derivative of the exponential in the direction dL/dtheta_p, composed with the static factor: what the reference's
 * ExpErrorgenOp.deriv_wrt_params() (experrorgenop.py:213-260) hands to MatrixForwardSimulator._doperation -- and applies the
 * chain rule to the element Jacobian (<= 1e-8 against the Matrix simulator).  (An explicit gst_set_derivs still takes
 * precedence in that mode; exact Hessian blocks of such models need it, with gst_set_second_derivs.) */
#define GST_LINDBLAD_FD_MAX_DEPTH 16
typedef struct {
    int32_t kind, obj, n_eff, n_blocks;
    int32_t block_type[4], block_mode[4], block_n[4];
    int64_t param0;
    int64_t term_offset;         /* first of this member's terms in term_re / term_im */
    const double *static_part;
} gst_lindblad_member;
GST_API int gst_set_lindblad(gst_plan *plan, int32_t n_params, int32_t n_members, const gst_lindblad_member *members,
                     int64_t n_terms, const double *term_re, const double *term_im);
GST_API int gst_set_lindblad_params(gst_plan *plan, const double *theta);
/* The plan's current dense model (row-major, the layout of gst_set_model); any pointer may be NULL. */
GST_API int gst_get_model(gst_plan *plan, double *gates, double *rhos, double *effects);

/* Implicit models (pyGSTi's LocalNoiseModel / CloudNoiseModel, e.g. create_crosstalk_free_model): a circuit layer is not a
 * stored dense superoperator but a ComposedOp of EmbeddedOps -- small one- and two-qubit operations embedded into the
 * register (pygsti/modelmembers/operations/embeddedop.py, composedop.py; reps: evotypes/densitymx/opcreps.cpp:93-158
 * `OpCRep_Embedded::acton`, :242-276 `OpCRep_Composed::acton`), and one small operation usually stands behind several
 * layers (`independent_gates=False`), so its parameters are SHARED between the plan's gates.  The reference's Map path
 * steps such a parameter with set_parameter_value (models/model.py:1198-1310: every member that holds it moves) and walks the
 * circuits through the reps factor by factor.  gst_set_composite describes that structure once; afterwards the DEVICE builds
 * the dense layers, the complete perturbed model of every finite-difference column, and the layers' exact derivative
 * matrices -- no host to_dense() / deriv_wrt_params() per model update or per column.
 *   leaves:   n_leaves dense operations of dimension leaf_dim[l] (4 / 16 / 64 = one / two / three qubits); their elements
 *             (row-major) concatenated; leaf_param[e] = the model parameter that IS element e (FullArbitraryOp, the rows of
 *             a FullTPOp ...), -1 for elements that are no parameter (static leaves, a TP operation's first row).
 *   factors:  factor f = leaf factor_leaf[f] embedded on the qubits factor_targets[3 f .. 3 f + 2] (as many as the leaf has
 *             qubits, in the order of the leaf's own tensor factors; the rest -1).  Qubit 0 is the most significant base-4
 *             digit of the state index (the register's Pauli-product basis is the Kronecker product in qubit order).
 *   layers:   gate g of the plan = Emb(f_{n-1}) ... Emb(f_1) Emb(f_0) over factors gate_factor_ptr[g] .. gate_factor_ptr[g+1]
 *             (f_0 acts first, as ComposedOp applies its factorops in order); an empty list is the identity.
 * Preparations and effects stay dense vectors; parameters that are elements of those are declared as usual with
 * gst_set_param_map (GST_KIND_NONE for every parameter that belongs to a leaf).
 *   gst_set_composite          copies the description (NULL or n_leaves = 0 clears).  D = 4, 16, 64.
 *   gst_set_composite_values   takes the leaves' current elements (same concatenation) and the dense preparations / effects
 *       -- call it where gst_set_model would be called, after every model.from_vector(); the device builds every layer and the
 *       result is the plan's model for ALL fills (gst_get_model reads it back).
 * While set, GST_DERIV_FD columns of gst_fill_dprobs(_dev) are mapfill_dprobs_atom's for such a model
 * (mapforwardsim_calc_densitymx.pyx:349-381): per column the device moves every leaf element that is the parameter by eps,
 * rebuilds the layers that contain the leaf, and walks the circuits with that complete dense model (the whole-model walk of
 * gst_fill_dprobs_models, fed from device memory).  Accuracy as gst_fill_dprobs_models: the reference propagates factor by
 * factor, this path through the dense product -- probabilities to ~1e-15, quotients to ~1e-8, not bit for bit.
 * GST_DERIV_ANALYTIC columns are exact: the device forms every layer's d(dense)/d(parameter) by the product rule over its
 * factors (what ComposedOp / EmbeddedOp.deriv_wrt_params() hand to MatrixForwardSimulator._doperation) and applies the chain
 * rule to the element Jacobian (<= 1e-8 against the Matrix simulator).  Exact HESSIAN blocks of such models are not built
 * here (a layer is bilinear in two leaves): use gst_set_derivs + gst_set_second_derivs for those.  An explicit
 * gst_set_derivs takes precedence in the analytic mode.
 * GENERAL leaves: a leaf whose elements are not parameters themselves (an exponentiated Lindblad generator behind a CPTPLND
 * one- or two-qubit gate, a ComposedOp of such, ...) is declared with leaf_n_params[l] > 0 and its model parameters (ascending)
 * in leaf_param_list; its leaf_param entries are -1.  Such a leaf is at most 16 x 16 (64 x 64), so the host keeps
 * differentiating IT with the reference's own code -- what never happens on the host any more is the embedding and the
 * products at the register's dimension:
 *   gst_set_composite_general  after every gst_set_composite_values: leaf_derivs = the general leaves' deriv_wrt_params()
 *       (row-major [leaf_dim^2][n_params], concatenated in leaf order; may be NULL when only FD fills follow) and
 *       leaf_fd_values = their dense elements after each of their parameters' finite-difference steps
 *       ([n_params][leaf_dim^2] per leaf, concatenated; to_dense() after set_parameter_value(q, theta_q + fd_eps), restored
 *       afterwards; may be NULL when only exact fills follow).  GST_DERIV_FD fills must then use eps == fd_eps. */
typedef struct gst_composite_desc {
    int32_t n_leaves;
    const int32_t *leaf_dim;          /* [n_leaves] */
    const int64_t *leaf_param;        /* [sum leaf_dim^2] */
    const int32_t *gate_factor_ptr;   /* [n_gates + 1] */
    const int32_t *factor_leaf;       /* [n_factors] */
    const int32_t *factor_targets;    /* [n_factors][3] */
    const int32_t *leaf_n_params;     /* [n_leaves]: > 0 for general leaves; NULL: none */
    const int64_t *leaf_param_list;   /* the general leaves' parameters, concatenated */
} gst_composite_desc;
GST_API int gst_set_composite(gst_plan *plan, int32_t n_params, const gst_composite_desc *desc);
GST_API int gst_set_composite_values(gst_plan *plan, const double *leaf_values, const double *rhos, const double *effects);
GST_API int gst_set_composite_general(gst_plan *plan, const double *leaf_derivs, const double *leaf_fd_values, double fd_eps);
/* The dense models the device builds for the finite-difference steps of parameters param_idx (theta_p + eps each):
 * gates[n][n_gates][D][D], rhos[n][n_rhos][D], effects[n][n_effects][D] (host; tests compare them with the reference's). */
GST_API int gst_get_lindblad_model_sets(gst_plan *plan, const int64_t *param_idx, int64_t n_param, double eps, double *gates,
                                double *rhos, double *effects);

/* probs: out[n_elements] (host). */
GST_API int gst_fill_probs(gst_plan *plan, double *out);

/* dprobs: out[k*ld + dest_idx[c]] = d p_k / d theta_{param_idx[c]} for c < n_param (host, row-major,
 * leading dimension ld so that a column window of an 'ep' array can be filled in place).
 * dest_idx == NULL means dest c.  probs_out (may be NULL) receives the probabilities, as
 * bulk_fill_dprobs(pr_array_to_fill=...) does.  mode/eps: GST_DERIV_FD with the simulator's
 * derivative_eps (1e-7 in the reference, mapforwardsim.py:166-172). */
GST_API int gst_fill_dprobs(gst_plan *plan, double *out, int64_t ld, const int64_t *param_idx,
                    const int64_t *dest_idx, int64_t n_param, int mode, double eps, double *probs_out);

/* Finite-difference Jacobian columns for ANY parameterisation (CPTPLND, composed / embedded / exponentiated members,
 * parameters shared between members ...): what mapfill_dprobs_atom computes by stepping the model itself,
 *     for each parameter i:  model.set_parameter_value(i, orig + eps);  probs2 = probabilities;  (probs2 - probs) / eps
 * (pygsti/forwardsims/mapforwardsim_calc_densitymx.pyx:349-381, models/model.py:1198-1310).  The caller steps ITS model
 * on the host and hands over the n_models perturbed models as complete dense sets -- gates[m][n_gates][D][D],
 * rhos[m][n_rhos][D], effects[m][n_effects][D], m < n_models, C-contiguous f64 (what to_dense('minimal') returns after
 * each step) -- so nothing is assumed about which elements a parameter moves.  Column dest_idx[m] (m when dest_idx is
 * NULL) of `out` receives (p(model set m) - p(base)) / eps, the base being the model of gst_set_model; probs_out (may be
 * NULL) the base probabilities.  gst_set_param_map / gst_set_derivs are not needed.  The device runs every (walk
 * program, model set) pair as an independent probability walk -- no state is shared with the base pass, which is the
 * price of generality: members that are one-parameter-per-element (full, full TP) should use gst_fill_dprobs.
 * Accuracy: the reference propagates non-dense members (ComposedOp / ExpErrorgenOp reps, opcreps.cpp:242-524) factor by
 * factor, this path through their dense product, so probabilities agree to rounding (~1e-16) and the quotients to
 * ~1e-16 / eps, not bit for bit. */
GST_API int gst_fill_dprobs_models(gst_plan *plan, int64_t n_models, const double *gates, const double *rhos,
                           const double *effects, double *out, int64_t ld, const int64_t *dest_idx, double eps,
                           double *probs_out);
/* device-resident output (d_out, d_probs_out on the plan's device); the model sets are still host arrays */
GST_API int gst_fill_dprobs_models_dev(gst_plan *plan, int64_t n_models, const double *gates, const double *rhos,
                               const double *effects, double *d_out, int64_t ld, const int64_t *dest_idx, double eps,
                               double *d_probs_out);

/* Second derivatives of the dense elements with respect to the parameters, for the objects of the last gst_set_derivs
 * call, in the same order: what MatrixForwardSimulator._hoperation = member.hessian_wrt_params() feeds into the exact
 * Hessian (pygsti/forwardsims/matrixforwardsim.py:192-224, 1190-1287) for members that are not linear in their
 * parameters (CPTPLND / exponentiated error generators, ...).  nonzero[o] != 0 marks the objects that have such a
 * tensor; hess holds them concatenated, each row-major [n_elem][n_cols[o]][n_cols[o]].  While set,
 * gst_fill_hprobs_analytic / gst_objective_hessian_block(analytic) add  sum_a (d p / d elem_a) d^2 elem_a / d p1 d p2.
 * Cleared by gst_set_derivs (call it first); n_objs = 0 clears. */
GST_API int gst_set_second_derivs(gst_plan *plan, int32_t n_objs, const int32_t *nonzero, const double *hess);

/* hprobs block: out[(k*ld1 + dest1[a])*ld2 + dest2[b]] = d2 p_k / d theta_{idx1[a]} d theta_{idx2[b]}
 * by finite differences of finite differences with step eps, bit-for-bit
 * MapForwardSimulator._mapfill_hprobs_atom (mapforwardsim.py:394-438). */
GST_API int gst_fill_hprobs(gst_plan *plan, double *out, int64_t ld1, int64_t ld2,
                    const int64_t *idx1, const int64_t *dest1, int64_t n1,
                    const int64_t *idx2, const int64_t *dest2, int64_t n2, double eps);

/* The same block with EXACT second derivatives -- what MatrixForwardSimulator._bulk_fill_hprobs_atom returns
 * (pygsti/forwardsims/matrixforwardsim.py:1190-1287, 1289-1381) -- for the `full` parameterisation:
 * derivative forward / backward states of every row parameter over the prefix / suffix tries, contracted with the
 * cached backward / forward states (on the MFMA cores at D = 16 and 64).  Same argument meaning as gst_fill_hprobs (no step size).
 * While gst_set_derivs is set the block is that of the general parameterisation, for members whose dense elements are
 * LINEAR in their parameters (TP: has_nonzero_hessian() is False for every member): the element Hessian of the elements
 * the requested parameters touch, contracted with the derivative columns on both sides.  Members with second
 * derivatives (CPTPLND, ...) additionally need gst_set_second_derivs (the J_elem . hessian_wrt_params term). */
GST_API int gst_fill_hprobs_analytic(gst_plan *plan, double *out, int64_t ld1, int64_t ld2, const int64_t *idx1,
                             const int64_t *dest1, int64_t n1, const int64_t *idx2, const int64_t *dest2, int64_t n2);

/* Device-resident variants: `d_out` / `d_probs_out` are device pointers on the plan's device
 * (e.g. a buffer the caller shares with RCCL).  Asynchronous on the plan's stream. */
GST_API int gst_fill_probs_dev(gst_plan *plan, double *d_out);
GST_API int gst_fill_dprobs_dev(gst_plan *plan, double *d_out, int64_t ld, const int64_t *param_idx,
                        const int64_t *dest_idx, int64_t n_param, int mode, double eps,
                        double *d_probs_out);
GST_API int gst_sync(gst_plan *plan);

/* Normal equations of the least-squares fit on the device ("next" row f1 of SURVEY 8(f); the reference does this
 * on the host after the Jacobian has been scaled by the objective's dterms: layout.fill_jtj / fill_jtf,
 * pygsti/layouts/distlayout.py:1220-1359, copalayout.py:549-598, consumed by optimize/simplerlm.py:677-678).
 * d_J is a device-resident row-major [n_rows][ld] Jacobian block (n_cols <= ld columns used), e.g. what
 * gst_fill_dprobs_dev left in HBM -- so the 7 GB Jacobian never crosses PCIe, only n_cols^2 + n_cols numbers do.
 *   d_row_scale (may be NULL): per-row factor w_k, J_s = diag(w) J (the objective's dlsvec scaling).  The rows of d_J are
 *       multiplied IN PLACE by one streaming pass before the product (the reference scales its Jacobian in place, too:
 *       objectivefns.py:4633-4665), so d_J holds J_s afterwards: call gst_fill_jtj_dev with the scale ONCE per Jacobian,
 *       then gst_fill_jtf_dev (which has no scale argument and contracts whatever d_J holds); a second scaled call
 *       would scale the rows twice.  Pass NULL to contract d_J as it is.
 *   gst_fill_jtj_dev: d_jtj[n_cols][n_cols] = J_s^T J_s   (full symmetric matrix; split-K MFMA fp64 kernel)
 *   gst_fill_jtf_dev: d_jtf[n_cols] = d_J^T f,  f = d_f[n_rows]   (= J_s^T f after a scaled gst_fill_jtj_dev)
 * With several ranks each computes the partial sums of its own rows; gst_comm_allreduce_sum (RCCL) adds the
 * n_cols^2 doubles on the device. */
GST_API int gst_fill_jtj_dev(gst_plan *plan, double *d_J, int64_t n_rows, int64_t n_cols, int64_t ld,
                     const double *d_row_scale, double *d_jtj);
GST_API int gst_fill_jtf_dev(gst_plan *plan, const double *d_J, int64_t n_rows, int64_t n_cols, int64_t ld,
                     const double *d_f, double *d_jtf);
/* The same normal equations WITHOUT touching d_J (round 5): d_jtj = (diag(w) J)^T (diag(w) J) and d_jtf =
 * (diag(w) J)^T f with w = d_row_scale (NULL: all ones) applied while the rows are staged -- every weighted element is
 * rounded exactly as the in-place scaling of gst_fill_jtj_dev stores it, so d_jtj comes out with the same bits (d_jtf too,
 * except on the block-sparse path of large matrices -- >= 16,384 rows, > 384 columns, both outputs requested -- where the
 * pass that marks the live panels also carries J_s^T f: one read of d_J for both, another fixed summation order), but the
 * 2 x n_rows x n_cols x 8 bytes of the scaling pass' read-modify-write are not moved and d_J stays the plain Jacobian
 * (so it may be contracted again, copied out, or -- for an exact Jacobian in tracked memory -- keep its resident zeros
 * whatever the weights are).  Either output may be NULL (d_jtf needs d_f).  Replaces the same reference lines as
 * gst_fill_jtj_dev / gst_fill_jtf_dev (objectivefns.py:4633-4665 scaling, distlayout.py:1220-1359 products). */
GST_API int gst_fill_normal_eqs_dev(gst_plan *plan, const double *d_J, int64_t n_rows, int64_t n_cols, int64_t ld,
                            const double *d_row_scale, const double *d_f, double *d_jtj, double *d_jtf);
GST_API int gst_memcpy_h2d(gst_plan *plan, void *d_dst, const void *src, int64_t nbytes);
/* A rows x cols block of doubles between two device arrays with their own leading dimensions (in doubles), enqueued on
 * the plan's stream.  No counterpart in the reference, whose arrays live on the host: it is what re-assembles whole
 * Jacobian rows from the column blocks the parameter-processors of an atom-processor hold (the reference broadcasts
 * transposed column blocks between host arrays for the same purpose, layouts/distlayout.py:1306-1346). */
GST_API int gst_copy_block_dev(gst_plan *plan, double *d_dst, int64_t dst_ld, const double *d_src, int64_t src_ld,
                       int64_t n_rows, int64_t n_cols);

/* Element-wise objective maps on device-resident probabilities (row f1): what the reference evaluates with numpy
 * between bulk_fill_dprobs and fill_jtj -- RawChi2Function / RawPoissonPicDeltaLogLFunction .terms/.lsvec/.dterms
 * (pygsti/objectivefns/objectivefns.py:1814-1885, 2040-2084, 2944-3160, 3185-3195) combined as
 * TimeIndependentMDCObjectiveFunction.lsvec / .dlsvec do (:4573-4593, 4633-4665):
 *   d_lsvec[k]    = sqrt(terms_k) (chi^2: (p-f) sqrt(N/max(p, min_prob_clip)), signed)
 *   d_rowscale[k] = (|lsvec_k| < 1e-100 ? 0 : 0.5 / lsvec_k) * dterms_k      (the factor of row k of dprobs in dlsvec)
 *   d_terms[k]    = terms_k (may be NULL);  *sum_terms = sum_k terms_k (may be NULL; blocks until done)
 * d_probs is clipped in place to [prob_clip_lo, prob_clip_hi] first when prob_clip_lo < prob_clip_hi (_clip_probs
 * :4766-4774).  d_counts / d_totals are the data set's counts and total counts per element (device, f64).
 * Feed d_rowscale to gst_fill_jtj_dev and d_lsvec to gst_fill_jtf_dev. */
#define GST_OBJ_CHI2 0
#define GST_OBJ_POISSON_DLOGL 1
typedef struct gst_objective_desc {
    int32_t kind;            /* GST_OBJ_* */
    int32_t hessian_mode;    /* gst_objective_hessian_block only: GST_DERIV_FD (0) = FD-of-FD hprobs and FD dprobs, the Map
                                path's semantics; GST_DERIV_ANALYTIC (1) = exact hprobs and dprobs, the Matrix path's */
    double min_prob_clip;    /* chi^2: min_prob_clip_for_weighting; dlogl: min_prob_clip ('minp' regularisation) */
    double radius;           /* dlogl: zero-frequency radius ("harsh" regularisation) */
    double prob_clip_lo, prob_clip_hi;
} gst_objective_desc;
GST_API int gst_objective_rows_dev(gst_plan *plan, const gst_objective_desc *desc, double *d_probs, const double *d_counts,
                           const double *d_totals, int64_t n, double *d_lsvec, double *d_rowscale, double *d_terms,
                           double *sum_terms);

/* One Levenberg-Marquardt evaluation in one BLOCKING call (round 6): the model of the last gst_set_model is uploaded, then
 *   gst_fill_dprobs_dev(all n_params parameters, GST_DERIV_FD, eps) -> d_J [nE][ld], d_probs
 *   gst_objective_rows_dev                                          -> d_lsvec, d_rowscale, *sum_terms
 *   gst_fill_normal_eqs_dev                                         -> d_jtj [n_params][n_params], d_jtf [n_params]
 * run back to back -- the same kernels, the same bits.  For LAUNCH-BOUND plans (at most 65,536 states: the 1Q designs of
 * BASELINE configs[1], where such an iteration is seven kernel launches, two memsets and two copies around microseconds of
 * work) the sequence is captured into a HIP graph at the second call with the same arguments and replayed afterwards: one
 * graph launch per iteration.  The graph reads the model from a page-locked buffer of the library's that every call
 * overwrites; another request, other pointers or another parameter map drop it; a capture the runtime refuses is not
 * retried.  Element-mapped models (gst_set_param_map, full TP without a complement effect) only.  Replaces, per iteration,
 * simplerlm.py:663-678 + objectivefns.py:4573-4665 + distlayout.py:1220-1359 on device data. */
GST_API int gst_lm_step_dev(gst_plan *plan, const gst_objective_desc *desc, int64_t n_params, double eps, const double *d_counts,
                    const double *d_totals, double *d_J, int64_t ld, double *d_probs, double *d_lsvec, double *d_rowscale,
                    double *d_jtj, double *d_jtf, double *sum_terms);

/* One (n1 x n2) block of the objective's Hessian without moving the hprobs block off the device: what
 * TimeIndependentMDCObjectiveFunction._construct_hessian does per rectangle (objectivefns.py:1640-1690) with
 * _iter_atom_hprobs_by_rectangle (forwardsims/distforwardsim.py:304-340) and _hessian_from_block (:4914-4968):
 *   out[i][j] = sum over this plan's elements e of
 *                 hterms_e * dprobs[e][idx1[i]] * dprobs[e][idx2[j]]  +  dterms_e * hprobs[e][idx1[i]][idx2[j]]
 * with the FD-of-FD hprobs and FD dprobs of gst_fill_hprobs / gst_fill_dprobs (step eps for both, as the reference's
 * Map path), dterms / hterms the raw objective's first and second derivatives in the probabilities (chi^2 or Poisson
 * dlogl, on the clipped probabilities when the interval is given).  d_counts / d_totals: device, per element.
 * out: host, row-major [n1][n2]; atoms / ranks add their blocks.  The (nE x n1 x n2) hprobs block lives in device
 * memory only for the duration of the call -- size the rectangles to fit.
 * desc->hessian_mode = GST_DERIV_ANALYTIC uses the exact dprobs / hprobs instead; with gst_set_derivs set (linear
 * general parameterisations, see gst_fill_hprobs_analytic) that is the only mode.  A complement effect
 * (gst_set_complement_effect) is honoured by the FD mode. */
GST_API int gst_objective_hessian_block(gst_plan *plan, const gst_objective_desc *desc, const double *d_counts,
                                const double *d_totals, const int64_t *idx1, int64_t n1, const int64_t *idx2,
                                int64_t n2, double eps, double *out);

/* ---- Multi-GPU exchange: one process per GPU, row blocks and normal equations travel between DEVICE buffers -------------
 * The path shards by layout atoms (disjoint circuit groups with contiguous element slices, pygsti/layouts/
 * distlayout.py:326-332, 404-415): every rank fills the rows of its own atoms and no collective is needed inside a fill.
 * What the reference then does with MPI on host arrays -- Gatherv / Allgatherv of row blocks
 * (pygsti/baseobjs/resourceallocation.py:316-348 `gather_base`, used by layout.gather_local_array / allgather_local_array,
 * layouts/distlayout.py:1010-1156, copalayout.py:479-518) and Allreduce(SUM) of J^T J / J^T f
 * (resourceallocation.py:441-508 `allreduce_sum`, used by fill_jtj / fill_jtf, distlayout.py:1220-1359) -- these entry
 * points do from device pointers, over xGMI:
 *   GST_TRANSPORT_RCCL: RCCL (librccl, bound at run time with dlopen: single-GPU users never load it).  Row blocks move
 *       as ONE grouped ncclSend/ncclRecv exchange -- every pair of GPUs has its own xGMI link, so a fan-in / all-to-all of
 *       point-to-point transfers uses all links at once where a ring is bound by one; sums use ncclAllReduce.
 *       Stream-ordered: operations are enqueued behind the fills on the plan's stream (gst_sync / gst_comm_sync wait).
 *   GST_TRANSPORT_IPC: intra-node peer writes -- every rank maps the destination buffers of the others through HIP IPC
 *       handles published in a POSIX shared-memory mailbox and copies its blocks straight into them (SDMA engines, no
 *       compute units); sums gather every rank's copy and add them in rank order (bit-reproducible).  Works when ranks
 *       share a GPU (which RCCL refuses), and is the fallback when RCCL cannot initialise.  Blocking: returns when the
 *       data has landed on every receiver.
 * Rendezvous: rank 0 calls gst_comm_get_unique_id, the caller distributes the GST_COMM_ID_BYTES bytes by any means
 * (MPI, a torch.distributed store, a file), every rank calls gst_comm_create (collective).  `plan` (may be NULL) selects
 * the stream: the plan's stream when given (must live on the comm's device), else the comm's own stream. */
typedef struct gst_comm gst_comm;
#define GST_COMM_ID_BYTES 128
#define GST_TRANSPORT_RCCL 0
#define GST_TRANSPORT_IPC 1
GST_API int gst_comm_get_unique_id(int transport, void *id_out);
GST_API int gst_comm_create(int transport, int device, int rank, int size, const void *id, gst_comm **out);
GST_API int gst_comm_destroy(gst_comm *comm);

/* Row blocks of a row-major [n_rows][row_doubles] f64 array: block b = rows [blk_row0[b], blk_row0[b] + blk_rows[b]) of the
 * ASSEMBLED array, owned (filled) by rank blk_owner[b] (a rank may own several: atoms r, r + size, ...).  Every rank passes
 * the same block list.
 *   gst_comm_allgather_rows: d_full is the full-size array on EVERY rank with the rank's own blocks already in place
 *       (the _dev fills write there directly); on return (stream order) every rank holds all blocks.
 *   gst_comm_gather_rows: only `root` holds the full-size array (d_full, own blocks in place); the other ranks pass
 *       d_local = their own blocks packed one after another in block order (d_full is ignored there).  Gatherv. */
GST_API int gst_comm_allgather_rows(gst_comm *comm, gst_plan *plan, double *d_full, int64_t row_doubles, int32_t n_blocks,
                            const int32_t *blk_owner, const int64_t *blk_row0, const int64_t *blk_rows);
GST_API int gst_comm_gather_rows(gst_comm *comm, gst_plan *plan, const double *d_local, double *d_full, int64_t row_doubles,
                         int32_t n_blocks, const int32_t *blk_owner, const int64_t *blk_row0, const int64_t *blk_rows,
                         int32_t root);
/* The fan-in without a copy (round 6): collective; every rank receives in *d_mapped a device pointer, valid in ITS process,
 * onto `root`'s buffer d_buf (the root passes its own pointer and gets it back; the others pass anything).  A rank then fills
 * its row block straight into the assembled array -- gst_fill_dprobs_dev(plan, (double *)mapped + row0 * ld, ...) -- and the
 * kernel's stores cross xGMI while it runs: what `north_star` calls the gather of the Jacobian blocks to rank 0 costs no pass
 * of its own (the reference gathers host arrays after the fill, resourceallocation.py:329-348).  The root must not free d_buf
 * before every rank is done with the mapping (gst_comm_barrier); mappings are closed by gst_comm_destroy.  Transport: the
 * root's HIP IPC handle travels through the shared-memory mailbox (GST_TRANSPORT_IPC) or through the RCCL communicator itself
 * (80 bytes, ncclSend / ncclRecv); ranks must be on one node. */
GST_API int gst_comm_map_root_buffer(gst_comm *comm, int32_t root, void *d_buf, void **d_mapped);
/* General block exchange between device arrays (an Alltoallv): block b = blk_count[b] doubles read at d_src + blk_src_off[b]
 * on rank blk_src_rank[b] and written at d_dst + blk_dst_off[b] on rank blk_dst_rank[b] (offsets in doubles; a block whose two
 * ranks coincide is a local copy).  Every rank passes the same list and its OWN d_src / d_dst.  This is the exchange step of
 * the normal equations when the Jacobian's columns are distributed over the parameter-processors of an atom-processor
 * (layouts/distlayout.py:1306-1346: the reference broadcasts each column slice's transpose between host arrays): every
 * rank sends its columns of the row range another rank will contract and receives that range's other columns.  Grouped
 * ncclSend / ncclRecv under RCCL (each pair over its own xGMI link), peer copies under the IPC transport. */
GST_API int gst_comm_exchange_blocks(gst_comm *comm, gst_plan *plan, const double *d_src, double *d_dst, int32_t n_blocks,
                             const int32_t *blk_src_rank, const int32_t *blk_dst_rank, const int64_t *blk_src_off,
                             const int64_t *blk_dst_off, const int64_t *blk_count);
/* d_buf[0..n) <- sum over ranks of their d_buf, on every rank (in place). */
GST_API int gst_comm_allreduce_sum(gst_comm *comm, gst_plan *plan, double *d_buf, int64_t n);
GST_API int gst_comm_barrier(gst_comm *comm);                 /* all ranks have reached this call; outstanding exchanges done */
GST_API int gst_comm_sync(gst_comm *comm);                    /* the comm's own stream (plan == NULL operations) */
typedef struct {
    int32_t transport, rank, size, device;
    int32_t rccl_version;        /* ncclGetVersion code of the library bound at run time (0 for the IPC transport) */
    int32_t ipc_opens;           /* IPC transport: peer allocations this rank has mapped so far (hipIpcOpenMemHandle calls);
                                    destinations that alternate are re-published under their old id and do not re-open */
    int32_t reserved[2];
} gst_comm_info;
GST_API int gst_comm_get_info(const gst_comm *comm, gst_comm_info *out);

/* Plain device-buffer helpers on the plan's device, so that callers without any GPU framework can
 * keep results resident (bench.py, tests).  Buffers from any other allocator work equally. */
GST_API int gst_device_malloc(gst_plan *plan, int64_t nbytes, void **d_ptr);
/* The same allocation, declared TRACKED: the caller states that every write to this memory goes through this library
 * (fills, gst_memcpy_h2d, gst_copy_block_dev, the objective maps, gst_comm_*; each reports what it overwrites) or is
 * announced with gst_device_touch.  Only such memory -- and the plans' private staging buffers -- is eligible for the
 * default of GST_OPT_ANALYTIC_KEEP_ZEROS (2): a repeated exact fill skips the structural zeros it left there itself.
 * Memory from gst_device_malloc or from any other allocator is never trusted without the explicit promise (value 1).
 * Released with gst_device_free. */
GST_API int gst_device_malloc_tracked(gst_plan *plan, int64_t nbytes, void **d_ptr);
GST_API int gst_device_free(gst_plan *plan, void *d_ptr);
/* The caller wrote [d_ptr, d_ptr + nbytes) of gst_device_malloc_tracked memory by means other than this library (its own
 * kernel, a peer copy): whatever the library remembered about the contents is forgotten. */
GST_API int gst_device_touch(gst_plan *plan, void *d_ptr, int64_t nbytes);
GST_API int gst_memcpy_d2h(gst_plan *plan, void *dst, const void *d_src, int64_t nbytes);
/* The same copy enqueued on the plan's stream without waiting for it (gst_sync completes it): lets one process drain
 * the plans of several GPUs side by side -- with a page-locked destination (gst_host_register) the copies of different
 * devices overlap; a pageable one makes the call synchronous (the library copies through its own page-locked staging
 * buffer: the device never touches pageable caller memory), which is still correct. */
GST_API int gst_memcpy_d2h_async(gst_plan *plan, void *dst, const void *d_src, int64_t nbytes);

/* Page-lock a caller-owned host array (hipHostRegister, portable across devices) so that the host-output fills
 * (gst_fill_probs / gst_fill_dprobs / gst_fill_hprobs*) copy into it at full PCIe rate instead of through the runtime's
 * pageable staging -- the counterpart of the reference allocating its 'ep' arrays once per objective
 * (layout.allocate_local_array, pygsti/layouts/copalayout.py:284-361) and reusing them every iteration.  Unregister before
 * the memory is freed.  Both fail with GST_ENODEVICE when no device exists (the array then simply stays pageable).
 * Register memory that owns its pages (an mmap of its own, whole aligned pages; numpy gives such memory to arrays of
 * >= 32 MB): a range inside the malloc heap shares its first and last page with unrelated allocations, and registering /
 * unregistering such ranges over and over is what a rare GPU memory fault on a host-heap address was traced to.
 * A host destination that is NOT registered is filled through the library's own page-locked staging buffer.
 * The region is also mapped into the device's address space: a finite-difference gst_fill_dprobs whose destination lies
 * inside it (requests of >= 64 columns) has its kernel write the Jacobian straight into the host array (512-byte row segments over PCIe while the
 * walk is still computing; no HBM staging of the result), honouring (ld, dest_idx) as always. */
GST_API int gst_host_register(void *ptr, int64_t nbytes);
GST_API int gst_host_unregister(void *ptr);

/* Introspection (tests, bench, DESIGN.md numbers). */
GST_API int gst_get_stats(const gst_plan *plan, gst_stats *out);
/* Copies up to `cap` program words of the concatenated walk programs; returns the total count in
 * *n_words.  task_off (may be NULL) receives n_tasks+1 offsets when cap_tasks suffices. */
GST_API int gst_get_program(const gst_plan *plan, uint32_t *words, int64_t cap, int64_t *n_words,
                    int64_t *task_off, int64_t cap_tasks);
/* The "dirty programs" of finite differences over whole-object perturbations (gst_set_lindblad): for task t and object
 * class c (gate c for c < n_gates, preparation c - n_gates behind them; *n_classes = n_gates + n_rhos) the part of the
 * task's walk that a perturbation of that object changes -- program (t, c) = words[prog_off[t * n_classes + c] ...
 * prog_off[t * n_classes + c + 1]), empty when no outcome of the task sees the object.  GST_OP_CACHE id starts from state
 * `id` of the base pass; the rest are the ordinary opcodes.  Copies up to `cap` words, the total in *n_words; prog_off
 * (may be NULL) receives n_tasks * n_classes + 1 offsets when cap_progs suffices. */
/* The level program of the forward plan (which = 0; which = 2: its probability-only form, the circuits' final states and
 * their sources) or of the plan of the reversed circuits (which = 1; nv = n_effects vectors per state), built on the host for this call (no device needed) -- what GST_OPT_FAST_CHAINS executes; format in
 * csrc/gst_levels.hpp.  Two-call pattern: sizes come back in *n_words / *n_ids; arrays are filled when their capacities
 * suffice.  node_parent / node_sym (may be NULL): the state graph the ids refer to (for which = 1 the reversed plan's).
 * info[13] = usable, worthwhile, nv, scratch matrices per task, most stages of a task, stages, tiles, chains, nodes on
 * chains, sum over tasks of the deepest node, states, tasks, states produced. */
GST_API int gst_get_level_program(const gst_plan *plan, int32_t which, int32_t *words, int64_t cap_words, int64_t *n_words, int32_t *ids,
                          int64_t cap_ids, int64_t *n_ids, int64_t *task_off, int64_t cap_tasks, int32_t *node_parent,
                          int32_t *node_sym, int64_t cap_nodes, int64_t *info);
GST_API int gst_get_dirty_programs(const gst_plan *plan, uint32_t *words, int64_t cap, int64_t *n_words, int64_t *prog_off,
                           int64_t cap_progs, int32_t *n_classes);

/* Host-side utility of layout construction (no device, no plan): the circuits in prefix order.  Circuit c's key is
 * (circ_head[c], circ_syms[circ_ptr[c]] ... circ_syms[circ_ptr[c+1] - 1]) -- the state preparation, then the gate symbols,
 * as integers whose order the caller chooses -- compared element by element, a proper prefix first (Python's tuple order).
 * order_out[k] = the k-th circuit in that order (stable), lcp_out[k] = the number of leading key elements it shares with its
 * predecessor (0 for k = 0).  What `PrefixTable` / the distributed layouts' circuit partition compute with Python tuple
 * compares (layouts/prefixtable.py:26-101, 154-288: 16 s at 2Q L<=1024): the atoms of a multi-GPU layout are contiguous runs
 * of this order, cut where lcp is small.  circ_head may be NULL (one preparation). */
GST_API int gst_sort_circuits(int64_t n_circuits, const int64_t *circ_ptr, const int32_t *circ_syms, const int32_t *circ_head,
                      int64_t *order_out, int64_t *lcp_out);
/* first_out[c * n_syms + g] = position of the first occurrence of symbol g in circuit c, -1 if it never occurs (host-only):
 * what decides which parameter wavefronts of a finite-difference Jacobian re-propagate a state (those of the gates on its
 * path) -- the measure the atoms of a multi-GPU layout are balanced on. */
GST_API int gst_circuit_first_use(int64_t n_circuits, const int64_t *circ_ptr, const int32_t *circ_syms, int32_t n_syms,
                          int64_t *first_out);

/* The per-SIMD queues the persistent finite-difference launch of a small atom would use for these columns (host-side
 * only, no device needed): estimated work of each of n_queues queues after longest-first packing and hand-overs
 * (handover: 0 none, 1 balance, 2 cut every walk).  Needs gst_set_param_map.  D <= 16. */
GST_API int gst_get_fd_queues(gst_plan *plan, const int64_t *param_idx, int64_t n_param, int32_t n_queues, int32_t handover,
                      int64_t *load_out, int32_t *n_pairs, int32_t *n_handovers);

/* EXACT arithmetic of one GST_DERIV_FD fill of these columns by the lane-per-model walk (D <= 16; host-side only, no device
 * needed; needs gst_set_param_map) -- what bench.py's roofline line divides by the kernel time.  The kernel skips every
 * state no lane of a wavefront has perturbed (bit-identical to the base pass by construction); this walks the plan's
 * programs with that same rule.  out[8] =
 *   [0] wavefront-applications executed      [1] column-applications executed (live lanes only)
 *   [2] wavefront-dots executed (EMIT outcomes with real arithmetic)      [3] column-dots executed
 *   [4] wavefront-applications of the full schedule (wavefronts x applications per pass)   [5] column-applications of it
 *   [6] wavefronts the columns are packed into   [7] tasks
 * Executed flops of the fill = 2 D^2 * 64 * out[0] + 2 D * 64 * out[2] (every lane of a wavefront issues); the reference
 * schedule's flops = n_param * (2 D^2 A + 2 D nE). */
GST_API int gst_get_fd_work(gst_plan *plan, const int64_t *param_idx, int64_t n_param, int64_t *out);
/* Host-only (no device needed): the TILES the D = 16 exact contraction forms for this plan (gst_kernels_tiles.hip: circuits
 * prep_i . W . meas_m that share their forward states row-wise and their backward states column-wise over W, 8 x 4 per
 * workgroup), checked against the per-circuit application tables the item kernel reads.  out[8] =
 *   [0] tiles   [1] circuits in tiles   [2] segment slots (x rows x columns = applications served from LDS)
 *   [3] remnant applications (gathered per circuit)   [4] INCONSISTENCIES found by the check (0)   [5] longest segment
 *   [6] circuits of the plan   [7] circuits that appear in exactly one tile (= [1]) */
GST_API int gst_get_tile_stats(gst_plan *plan, int64_t *out);

/* The state-id graph behind the NODE markers: parent state id (-1 for a state preparation) and gate / rho index of
 * every state, and the id of each expanded circuit's final state (what the analytic mode walks backwards). */
GST_API int gst_get_state_graph(const gst_plan *plan, int32_t *node_parent, int32_t *node_sym, int64_t cap_nodes,
                        int32_t *circ_leaf, int64_t cap_circuits, int64_t *n_nodes);

GST_API int gst_device_count(int32_t *n);
GST_API const char *gst_last_error(void);
GST_API const char *gst_version(void);

/* Walk-program encoding (one 32-bit word per instruction, opcode in the top 4 bits) -- public so
 * that tests can interpret programs independently of the device code. */
#define GST_OP_END 0u    /* end of task */
#define GST_OP_RHO 1u    /* v <- rho[arg] */
#define GST_OP_APPLY 2u  /* v <- G[arg] v */
#define GST_OP_SAVE 3u   /* slot[arg] <- v */
#define GST_OP_LOAD 4u   /* v <- slot[arg] */
#define GST_OP_EMIT 5u   /* elements of expanded circuit arg: p = E . v */
#define GST_OP_NODE 6u   /* marker after every RHO/APPLY: the state just produced has global id arg (the base
                            pass stores it in the base-state cache; derivative passes use it to skip work
                            that is bit-identical to the base pass) */
#define GST_OP_CACHE 7u  /* derived programs only (finite differences over whole-object perturbations): v <- state arg of the
                            base pass's cache */
#define GST_OP(word) ((word) >> 28)
#define GST_ARG(word) ((word) & 0x0FFFFFFFu)

#ifdef __cplusplus
}
#endif
#endif /* GSTFWD_H */
