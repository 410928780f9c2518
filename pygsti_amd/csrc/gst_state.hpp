// gst_state.hpp -- what the translation units behind the C ABI share: the plan object (host plan + every device buffer
// and cached request of a `gst_plan`), the error / guard helpers, and the prototypes of the internal drivers.
//   gst_abi.cpp           plan life cycle, model / parameter-map upload, the fill entry points and their dispatch, memory
//                         utilities, introspection
//   gst_fill_fd.cpp       probabilities, the level passes' host side, finite-difference Jacobians (lane packing, queues,
//                         launch forms), whole-model finite differences
//   gst_fill_analytic.cpp reversed plan, exact Jacobians (two-cache contraction), general parameterisations (chain rule)
//   gst_hessian.cpp       Hessian blocks: FD of FD, composed, exact, general; objective-Hessian rectangles
//   gst_lindblad_abi.cpp  Lindblad members on the device: description, parameters, FD and exact Jacobians
//   gst_normal_abi.cpp    objective maps, J^T J, J^T f
// Private to pygsti_amd/csrc: nothing here is part of include/gstfwd.h.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <queue>
#include <set>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/gstfwd.h"
#include "gst_internal.hpp"
#include "gst_kernels.hpp"
#include "gst_plan.hpp"
#include "gst_levels.hpp"

namespace gst_impl {

inline int fail(int code, const std::string& msg) { return gst::set_error(code, msg); }

#define HIP_TRY(expr)                                                                           \
    do {                                                                                        \
        hipError_t e_ = (expr);                                                                 \
        if (e_ != hipSuccess) {                                                                 \
            const int code_ = (e_ == hipErrorOutOfMemory) ? GST_ENOMEM                          \
                              : (e_ == hipErrorNoDevice || e_ == hipErrorInvalidDevice) ? GST_ENODEVICE : GST_EHIP; \
            return fail(code_, std::string(#expr) + ": " + hipGetErrorString(e_));              \
        }                                                                                       \
    } while (0)

// No exception may cross the C ABI (it would reach std::terminate): every extern "C" entry runs inside this guard.
template <typename F>
int guarded(F&& body)
{
    try {
        return body();
    } catch (const std::bad_alloc&) {
        return fail(GST_ENOMEM, "out of host memory");
    } catch (const std::length_error& e) {
        return fail(GST_EINVAL, std::string("invalid size: ") + e.what());
    } catch (const std::exception& e) {
        return fail(GST_EINVAL, std::string("internal error: ") + e.what());
    } catch (...) {
        return fail(GST_EINVAL, "internal error (unknown exception)");
    }
}

// Timing events (HIP events around the dominant kernel / the whole call) cost a few microseconds of device time each;
// on launch-bound plans (1Q) they were a third of a fill, so they are recorded only when the plan asks for them.
#define TIME_REC(p, ev)                                                      \
    do {                                                                     \
        if ((p)->timing) HIP_TRY(hipEventRecord((p)->ev, (p)->stream));      \
    } while (0)

// GST_TEST_FORCE poison=1 (tests): a (re)grown device buffer starts from 0xFF bytes -- NaNs, index -1, counters at their
// maximum -- instead of zeros, so that any path that READS a word nothing wrote fails loudly instead of quietly (the
// diagnosis tool for "works because fresh memory happens to be zero")
extern int g_poison_fill;
template <typename T>
struct DevBuf {
    T* p = nullptr;
    size_t n = 0;
    unsigned flags = 0;                 // hipExtMallocWithFlags flags (0: plain hipMalloc)
    hipError_t ensure(size_t count)
    {
        if (count <= n && p) return hipSuccess;
        if (p) (void)hipFree(p);
        p = nullptr; n = 0;
        hipError_t e = flags ? hipExtMallocWithFlags((void**)&p, std::max<size_t>(count, 1) * sizeof(T), flags)
                             : hipMalloc((void**)&p, std::max<size_t>(count, 1) * sizeof(T));
        if (e != hipSuccess) { p = nullptr; return e; }
        n = std::max<size_t>(count, 1);
        // A (re)grown buffer starts from zeros, not from whatever an earlier allocation of this process left there: a table
        // entry or padding word that some path does not write is then the same harmless value in every run (once per
        // growth, at memory speed).
        e = hipMemset(p, g_poison_fill, n * sizeof(T));
        if (e == hipSuccess) e = hipStreamSynchronize(nullptr);      // (done before any of the plan's own, non-blocking streams touches it)
        return e;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; n = 0; }
};

}  // namespace gst_impl

using gst_impl::DevBuf;

struct gst_plan {
    gst::HostPlan hp;
    int device = -1;
    bool dev_ready = false;
    hipStream_t stream = nullptr;
    hipStream_t stream2 = nullptr;      // the backward chain pass of the analytic mode runs beside the forward one
    hipEvent_t ev0 = nullptr, ev1 = nullptr, evk0 = nullptr, evk1 = nullptr, ev_fork = nullptr, ev_join = nullptr;

    // device copies of the plan
    DevBuf<uint32_t> d_prog;
    DevBuf<int64_t> d_task_off;
    DevBuf<int32_t> d_eff_ptr, d_eff_label, d_eff_dest;
    // model
    std::vector<double> h_gates, h_gates_t, h_rhos, h_effects;
    bool have_model = false;
    // the model arrays [gates | gates transposed | rhos | effects] live in ONE device buffer filled by ONE copy from a
    // pinned staging buffer per gst_set_model (a 1Q fill is launch-bound: four pageable copies were a third of it)
    struct DevView { double* p = nullptr; };
    DevView d_gates, d_gates_t, d_rhos, d_effects;
    DevBuf<double> d_model;
    double* h_model_pinned[2] = {nullptr, nullptr};      // two staging buffers, used alternately
    hipEvent_t ev_upload[2] = {nullptr, nullptr};        // "the copy out of staging buffer i has been done"
    size_t h_model_pinned_n = 0;
    // page-locked staging for copies between device memory and PAGEABLE caller memory (gst_abi.cpp: d2h_bytes & co.)
    char* h_stage = nullptr;
    size_t h_stage_bytes = 0;
    // page-locked upload ring (h2d_async): the library's own tables and small caller arrays are copied INTO it at call time and
    // leave it asynchronously on `stream`; a wrap waits for the stream
    char* h_up = nullptr;
    size_t h_up_bytes = 0, h_up_at = 0;
    int upload_turn = 0;
    bool model_dirty = true;
    // parameter map
    std::vector<int32_t> pkind, pobj, pelem;
    bool have_pmap = false;
    // Caller's state dimension when it is none of 4 / 16 / 64 (a qutrit's 9, a leakage model ...): the plan runs at hp.D = the
    // next of those, every array that crosses the ABI is zero-padded / un-padded here (0: no padding).  Padded rows and
    // columns of gates and padded components of states are exact zeros, so every sum the kernels form gains only + 0.0 terms.
    int D_user = 0;
    int user_D() const { return D_user ? D_user : hp.D; }
    // work buffers
    DevBuf<double> d_jtj_part, d_jtf_part;   // split-K partial sums of the normal equations
    DevBuf<double> d_base_cache;   // [n_state_ids][D] states of the last base pass
    DevBuf<double> d_pbase, d_out, d_raw, d_dcol, d_probs_tmp;
    DevBuf<int32_t> d_lane[7];   // col, kind0, obj0, elem0, kind1, obj1, elem1
    // analytic mode, MFMA path (D = 16): plan of the reversed circuits, backward-state cache, pair tables
    gst::HostPlan rev;
    bool rev_ready = false;
    DevBuf<uint32_t> d_rprog;
    DevBuf<int64_t> d_rtask_off, d_pos_ptr;
    DevBuf<int32_t> d_reff_ptr, d_rev_leaf, d_pair_f, d_pair_r, d_circ_rho, d_circ_order, d_circ_partner, d_pair_common;
    DevBuf<int32_t> d_blk_f1, d_blk_f2, d_blk_r, d_blk_ptr;   // two-circuit items as one stream of 4-application blocks (ensure_reverse)
    static constexpr bool ana_stream = true;       // two-circuit items as one block stream (the gate-by-gate form remains for > 63 gates)
    static constexpr bool ana_pairs = true;        // two-circuit work items in the D = 16 contraction
    static constexpr bool ana_germ_order = true;   // germ-major order of the work items
    int ana_keep_zeros = 2;             // gst_set_option(GST_OPT_ANALYTIC_KEEP_ZEROS): 0 never, 1 any destination (the caller's
                                        // promise), 2 destinations the library tracks (gst_track.cpp) -- the default
    uint64_t uid = 0;                   // process-unique plan number, request_serial: bumped when the analytic column tables are
    uint64_t request_serial = 0;        // rebuilt -- (uid, serial, ld) is the signature of a Jacobian's zero pattern
    bool last_zeros_resident = false;
    DevBuf<uint64_t> d_rowmask;         // ensure_rowmask: rows x gates that can be non-zero in the element Jacobian
    bool rowmask_built = false, rowmask_ok = false;
    bool jelem_call = false;            // run_element_jacobian is filling its scratch (general parameterisations)
    const void* ana_zero_out = nullptr; // destination of the last stream-form analytic Jacobian, its leading dimension
    int64_t ana_zero_ld = 0;
    bool ana_zero_valid = false;
    // tiles of the D = 16 contraction (gst_kernels_tiles.hip; build_tiles in gst_fill_analytic.cpp) and the item tables of the
    // circuits no tile holds
    bool ana_tiles = false;             // GST_OPT_ANALYTIC_TILES (GST_TEST_FORCE tiles=1): the product tiles of the design on the tile kernel
    int32_t n_tiles = 0;
    bool ana_lpt = true;                // items dispatched longest power-of-two bucket first (GST_TEST_FORCE lpt=0: locality order only)
    int32_t tile_dbg = 0;               // GST_TEST_FORCE tile_dbg= (development: see TileArgs::debug)
    int64_t n_lo_circuits = 0;
    int64_t tile_stats[3] = {0, 0, 0};  // tiled circuits, segment slots, remnant slots
    DevBuf<int32_t> d_tile_order, d_tile_cid, d_tile_blk, d_tsf, d_tsr, d_rem_ptr, d_rem_f, d_rem_r;
    DevBuf<uint32_t> d_tile_counter;
    DevBuf<uint32_t> d_c64_order, d_c64_counter;   // resident-gate backward walk (D = 64): tasks longest first, pop counter
    bool chain_resident = false;        // GST_TEST_FORCE chain_resident=1: backward walk with register-resident gates (measured slower overall: DESIGN 4.5)
    DevBuf<int32_t> d_lo_order, d_lo_partner, d_lo_common, d_lo_blk_f1, d_lo_blk_f2, d_lo_blk_r, d_lo_blk_ptr;
    DevBuf<uint32_t> d_lo_range_begin, d_lo_counter;
    bool last_tiles = false;            // the last exact fill ran the tile kernel
    static constexpr bool ana_group_fetch = true;  // the four wavefronts of a workgroup take four consecutive items together
    DevBuf<double> d_rev_cache;
    DevBuf<uint32_t> d_work_counter, d_range_begin;
    bool want_cache_path = false;       // set by the Hessian driver around its set-up Jacobian call
    gst::AnaArgs last_ana;              // arguments of the last MFMA contraction (column maps, tables, caches)
    bool last_ana_valid = false;
    DevBuf<double> d_hscratch, d_dF, d_dB;
    DevBuf<int32_t> d_theta;            // 2 x 5 x 4 ints: derivative-walk parameter tables
    // general parameterisations (gst_set_derivs)
    bool derivs_set = false;
    // TP POVM complement (gst_set_complement_effect)
    int32_t comp_index = -1;
    std::vector<int32_t> comp_others;
    std::vector<double> comp_identity;
    std::vector<int32_t> ecol_tab;            // [4][n_ecols]: effect, component, output column, touches-complement
    std::vector<double> ecol_val;             // [2][n_ecols]: perturbed own component, recomputed complement component
    DevBuf<int32_t> d_ecol_tab;
    DevBuf<double> d_ecol_val;
    bool leaf_uploaded = false;
    int32_t dv_n_params = 0;
    std::vector<int32_t> dv_kind, dv_obj, dv_ncols;
    std::vector<int64_t> dv_param_idx, dv_off_cols, dv_off_deriv;
    std::vector<double> dv_deriv_h;     // host copy (the Hessian chain rule reads its sparsity)
    std::vector<int64_t> dv2_off;       // per object: offset of its second-derivative tensor in d_dv2 (-1: linear member)
    DevBuf<double> d_dv2;
    bool dv2_set = false;
    DevBuf<double> d_helem, d_hw;       // element-Hessian block, CSC weights
    DevBuf<int32_t> d_hcsc;             // CSC pointers / rows / destinations of both blocks
    DevBuf<double> d_dv_deriv, d_jelem;
    DevBuf<double> d_obj_dt, d_obj_ht, d_obj_pc, d_obj_tmp, d_hess_part, d_hess_out;   // objective Hessian blocks
    DevBuf<int32_t> d_dv_colmap;
    static constexpr bool ana_mfma = true;         // D = 16 / 64 analytic mode on the MFMA path (the one-kernel VALU form serves D = 4 and gate sets beyond LDS)
    // log-depth chain passes (gst_levels.hpp): level programs of the forward and of the reversed plan
    struct Levels {
        gst::LevelProgram prog;
        bool built = false, usable = false, uploaded = false;
        std::string why;                // why the plan has no level program (diagnostics)
        DevBuf<int32_t> d_words, d_ids;
        DevBuf<int64_t> d_task_off, d_ids_off;
        DevBuf<double> d_mats;
    } lv_fwd, lv_rev, lv_probs;      // lv_probs: the forward plan's probability-only program (only the circuits' final states and their sources)
    int fast_chains = 1;                // GST_OPT_FAST_CHAINS: 0 never, 1 where the stages are few against the chains (default), 2 always (tests)
    bool fast_probs = false;            // GST_OPT_FAST_PROBS: gst_fill_probs* through the level pass (<= 1e-10, not bit-exact)
    bool last_levels = false;           // the last fill took its states from the level pass
    int fd_split = 0;                   // gst_options.fd_split: 0 auto, 1 / 2 / 4 wavefronts per (task, 64 columns) pair
    int n_cus = 256;
    DevBuf<double> d_mm_models, d_mm_raw;   // gst_fill_dprobs_models: perturbed model sets, their probability vectors
    // gst_set_lindblad: members = static factor x exp(Lindblad error generator), built on the device
    struct Lindblad {
        bool set = false, have_theta = false, uploaded = false;
        int32_t n_params = 0, n_members = 0;
        std::vector<int32_t> kind, obj, n_eff, n_par, n_blocks, blk_type, blk_mode, blk_n;
        std::vector<int64_t> param0, term_off, static_off;
        std::vector<double> statics, term_re, term_im, theta;
    } lb;
    // gst_set_composite: layer operations of implicit models = products of embedded leaves, built on the device
    struct Composite {
        bool set = false, have_values = false, uploaded = false;
        int32_t n_params = 0, n_leaves = 0, max_leaf_dim = 0;
        std::vector<int32_t> leaf_dim, gate_fptr, factor_leaf, factor_targets;
        std::vector<int64_t> leaf_off, leaf_param;
        std::vector<std::vector<int64_t>> gate_params;      // per layer: the distinct parameters of its factors' leaves, ascending
        // general leaves (leaf_np[l] > 0): parameter lists, offsets of their derivative matrices / finite-difference values
        std::vector<int32_t> leaf_np;
        std::vector<int64_t> leaf_plist_off, leaf_plist, leaf_deriv_off, leaf_fd_off;
        int64_t n_deriv_doubles = 0, n_fd_doubles = 0;
        bool any_general = false, have_general_derivs = false, have_general_fd = false;
        double general_fd_eps = 0.0;
    } cmp;
    DevBuf<double> d_cmp_gderiv, d_cmp_gfd;
    DevBuf<int32_t> d_cmp_i32, d_cmp_pmap, d_cmp_items32;   // leaf_dim | gate_fptr | factor_leaf | factor_targets; the parameter map; deriv items
    DevBuf<int64_t> d_cmp_i64, d_cmp_setparam;              // leaf_off | leaf_param; the stepped parameter of each set / deriv item tables
    DevBuf<double> d_cmp_values, d_cmp_spam, d_cmp_base, d_cmp_gates_rm;
    DevBuf<int32_t> d_lb_i32;               // kind | obj | n_eff | n_par | n_blocks | blk_type | blk_mode | blk_n
    DevBuf<int64_t> d_lb_i64, d_lb_setparam;   // param0 | term_off | static_off; the stepped parameter of each set
    DevBuf<double> d_lb_statics, d_lb_term_re, d_lb_term_im, d_lb_theta, d_lb_base, d_lb_gates_rm, d_lb_pert;
    DevBuf<double> d_lb_ws64;               // D = 64: per-member workspace of the exponential (gst_kernels_lindblad.hip)
    DevBuf<int32_t> d_lb_waves;             // walk_pert_kernel's wave tables: kind | obj | n_eff | col0 | ncols | col_dest
    int32_t lb_n_pwaves = 0, lb_n_zero = 0;
    int64_t lb_n_sets = 0;                  // perturbed member sets of the cached request (columns of parameters that belong to a member)
    int64_t lb_n_items = 0;
    std::vector<int32_t> lb_povm_cols;      // per POVM member with requested columns: obj, n_eff, col0, ncols
    gst::DirtyPrograms dirty;               // gst::build_dirty_programs, once per plan
    bool dirty_ready = false;
    DevBuf<uint32_t> d_dirty_words;
    DevBuf<int64_t> d_dirty_off;
    DevBuf<int32_t> d_lb_item_pw;
    DevBuf<int32_t> d_lbr_lane[4];          // preparation columns on the lane-per-model kernel: col | kind | obj | elem (= model set)
    DevBuf<uint32_t> d_lbr_order;           // ... and their launch order (longest tasks first)
    int32_t lbr_n_waves = 0;
    static constexpr bool lb_rho_lanes = true;               // preparation columns of a Lindblad FD Jacobian on the lane-per-model kernel
    static constexpr bool lb_share = true;                   // Lindblad FD walks share the base pass's states (independent walks remain the fall-back for plans the shared kernel does not fit)
    DevBuf<int32_t> d_mm_dest;
    DevBuf<double> d_obj_part;          // per-block partial sums of the objective terms
    DevBuf<uint32_t> d_block_order;     // FD launch order of the cached request (expensive (task, wavefront) pairs first)
    bool have_block_order = false;
    DevBuf<int32_t> d_bin_ptr;          // persistent FD launch: per-SIMD queues of pairs
    DevBuf<uint32_t> d_bin_items, d_bin_head;
    DevBuf<uint64_t> d_trace;           // GST_FD_TRACE records
    int32_t n_bins = 0;
    bool have_bins = false;
    bool fd_persist = true;             // GST_TEST_FORCE persist=0: one workgroup per pair, placed by the dispatcher
    bool fd_persist_always = false;     // persist=2: per-SIMD queues whatever the number of pairs
    bool fd_fused = true;               // fused=0: launch-bound plans keep the separate base pass
    bool host_direct = true;            // host_direct=0: page-locked destinations are filled by a copy, not by the kernel
    int64_t host_direct_min_cols = 32;  // (narrower column windows would cross PCIe in segments of less than 256 bytes; 64 until round 3 --
                                        //  the 1Q model's 60 columns were just below it: blocking fill 103 -> 81 us with the kernel's direct stores)
    int fd_handover = 1;                // (GST_TEST_FORCE handover=) 0 never cut a walk, 1 cut to balance the per-SIMD queues, 2 cut every walk
    bool fd_overlap = true;             // overlap=0: the persistent FD launch keeps the separate base pass in front of it
    double test_cache_limit = 0;        // (GST_TEST_FORCE cache_limit=, bytes; tests) stands in for the 4 GB of 32-bit cache offsets
    static constexpr bool jtj_sparse = true;             // J^T J skips all-zero panels
    DevBuf<uint32_t> d_jtj_pmask;
    static constexpr bool fd_overlap_diag = false;       // (measurement form of round 3, retired)
    static constexpr bool fd_standby = true;             // stand-by launches behind the persistent one
    bool last_overlap = false;          // the last FD fill ran its base pass inside the persistent launch
    int last_fd_form = 0;               // gst_stats.last_fd_form
    bool test_skip_chains = false;      // skip_chains=1 (tests): the overlap launch walks no chain, so every wait runs out
    bool split_ready = false;
    std::vector<int32_t> cand_ptr, cand_pc;      // gst::task_split_candidates: where a walk may be handed over
    std::vector<float> cand_frac;
    std::vector<uint32_t> cand_live;
    int32_t n_split = 0;
    DevBuf<int32_t> d_task_split_pc, d_ho_index, d_ho_id, d_ho_live, d_ho_tag;
    DevBuf<uint32_t> d_ho_flag;
    DevBuf<double> d_ho_state;
    bool cached_fused = false;          // the cached lane tables were packed for the fused form
    std::vector<int32_t> task_cost;     // gst::task_gate_costs, computed at the first FD request
    DevBuf<int32_t> d_wave_row, d_wave_rowidx, d_lane_colidx;
    DevBuf<double> d_hrow;              // composed FD-of-FD Hessians: the stepped model's Jacobian over block 2
    DevBuf<int32_t> d_hdest;            // ... and the destination columns of block 2
    bool hess_composed = false;         // hess_composed=1: every FD-of-FD block through the composed route (tests)
    DevBuf<int32_t> d_node_parent, d_node_sym, d_node_run, d_circ_leaf, d_gate_col0, d_cm_gate, d_cm_rho, d_cm_eff;
    bool graph_uploaded = false;
    // the lane tables / column maps on the device describe this request (skip re-packing when it repeats)
    std::vector<int64_t> cached_pidx, cached_didx;
    int cached_kind = 0;        // 0 none, 1 FD lane tables, 2 analytic column maps
    bool cached_has_didx = false;
    int32_t cached_n_waves = 0;
    std::vector<int64_t> cached_none_cols;

    bool request_cached(int kind, const int64_t* pidx, const int64_t* didx, int64_t n) const
    {
        if (cached_kind != kind || (int64_t)cached_pidx.size() != n || cached_has_didx != (didx != nullptr)) return false;
        if (n && std::memcmp(cached_pidx.data(), pidx, sizeof(int64_t) * n)) return false;
        if (didx && n && std::memcmp(cached_didx.data(), didx, sizeof(int64_t) * n)) return false;
        return true;
    }
    void remember_request(int kind, const int64_t* pidx, const int64_t* didx, int64_t n)
    {
        cached_kind = kind; cached_pidx.assign(pidx, pidx + n); cached_has_didx = didx != nullptr;
        if (didx) cached_didx.assign(didx, didx + n); else cached_didx.clear();
    }

    // gst_lm_step_dev: the captured LM-iteration sequence of a launch-bound plan (gst_normal_abi.cpp)
    struct LmGraph {
        hipGraphExec_t exec = nullptr;
        uint64_t sig = 0, warm_sig = 0, request_serial = 0;
        double eps = 0;
        gst_objective_desc desc{};
        double* h_model = nullptr;      // page-locked: the graph's model source
        size_t h_model_n = 0;
        double* h_part = nullptr;       // page-locked: the objective's partial sums
        int n_blocks = 0;
        bool failed = false;
        int64_t replays = 0;
    } lm_graph;
    bool lm_graph_enabled = true;       // GST_TEST_FORCE lm_graph=0
    uint64_t fd_request_serial = 0;     // bumped whenever the FD request tables are rebuilt

    double last_kernel_ms = 0, last_total_ms = 0;
    int64_t last_launches = 0;
    bool timing = true;         // gst_options.timing: record the HIP events behind gst_stats.last_*_ms

    ~gst_plan()
    {
        if (!dev_ready) return;
        (void)hipSetDevice(device);
        if (d_out.p) gst::track_touch(d_out.p, d_out.n * 8);
        if (d_jelem.p) gst::track_touch(d_jelem.p, d_jelem.n * 8);
        d_cmp_i32.release(); d_cmp_pmap.release(); d_cmp_items32.release(); d_cmp_i64.release(); d_cmp_setparam.release();
        d_cmp_values.release(); d_cmp_spam.release(); d_cmp_base.release(); d_cmp_gates_rm.release(); d_cmp_gderiv.release(); d_cmp_gfd.release();
        d_lb_i32.release(); d_lb_i64.release(); d_lb_setparam.release(); d_lb_statics.release(); d_lb_term_re.release();
        d_lb_term_im.release(); d_lb_theta.release(); d_lb_base.release(); d_lb_gates_rm.release(); d_lb_pert.release(); d_lb_ws64.release(); d_lb_waves.release(); d_dirty_words.release(); d_dirty_off.release(); d_lb_item_pw.release(); d_jtj_pmask.release(); for (auto& b : d_lbr_lane) b.release(); d_lbr_order.release();
        d_prog.release(); d_block_order.release(); d_obj_part.release(); d_bin_ptr.release(); d_bin_items.release();
        d_bin_head.release(); d_trace.release(); d_ecol_tab.release(); d_ecol_val.release(); d_rprog.release();
        d_rtask_off.release(); d_pos_ptr.release(); d_reff_ptr.release(); d_rev_leaf.release(); d_pair_f.release();
        d_pair_r.release(); d_circ_rho.release(); d_circ_order.release(); d_circ_partner.release();
        d_pair_common.release(); d_rev_cache.release(); d_work_counter.release(); d_range_begin.release();
        d_blk_f1.release(); d_blk_f2.release(); d_blk_r.release(); d_blk_ptr.release();
        d_tile_order.release(); d_tile_cid.release(); d_tile_blk.release(); d_tsf.release(); d_tsr.release(); d_rem_ptr.release(); d_rem_f.release(); d_rem_r.release(); d_tile_counter.release();
        d_lo_order.release(); d_lo_partner.release(); d_lo_common.release(); d_lo_blk_f1.release(); d_lo_blk_f2.release(); d_lo_blk_r.release(); d_lo_blk_ptr.release(); d_lo_range_begin.release(); d_lo_counter.release();
        d_dv_deriv.release(); d_dv2.release(); d_helem.release(); d_hw.release(); d_hcsc.release();
        d_rowmask.release(); d_jelem.release(); d_dv_colmap.release(); d_hscratch.release(); d_dF.release(); d_dB.release();
        d_theta.release(); d_obj_dt.release(); d_obj_ht.release(); d_obj_pc.release(); d_obj_tmp.release();
        d_hess_part.release(); d_hess_out.release(); d_task_off.release(); d_eff_ptr.release();
        d_eff_label.release(); d_eff_dest.release();
        d_model.release();
        if (h_stage) (void)hipHostFree(h_stage);
        h_stage = nullptr; h_stage_bytes = 0;
        if (lm_graph.exec) (void)hipGraphExecDestroy(lm_graph.exec);
        if (lm_graph.h_model) (void)hipHostFree(lm_graph.h_model);
        if (lm_graph.h_part) (void)hipHostFree(lm_graph.h_part);
        if (h_up) (void)hipHostFree(h_up);
        h_up = nullptr; h_up_bytes = 0;
        for (int i = 0; i < 2; i++) {
            if (h_model_pinned[i]) (void)hipHostFree(h_model_pinned[i]);
            if (ev_upload[i]) (void)hipEventDestroy(ev_upload[i]);
        }
        d_mm_models.release(); d_mm_raw.release(); d_mm_dest.release();
        d_task_split_pc.release(); d_ho_index.release(); d_ho_id.release(); d_ho_live.release(); d_ho_tag.release(); d_ho_flag.release(); d_ho_state.release();
        d_pbase.release(); d_base_cache.release(); d_jtj_part.release(); d_jtf_part.release(); d_out.release(); d_raw.release(); d_dcol.release(); d_probs_tmp.release(); d_hrow.release(); d_hdest.release();
        for (auto& b : d_lane) b.release();
        d_wave_row.release(); d_wave_rowidx.release(); d_lane_colidx.release();
        d_node_parent.release(); d_node_sym.release(); d_node_run.release(); d_circ_leaf.release(); d_gate_col0.release();
        d_cm_gate.release(); d_cm_rho.release(); d_cm_eff.release();
        if (ev0) (void)hipEventDestroy(ev0);
        if (ev1) (void)hipEventDestroy(ev1);
        if (evk0) (void)hipEventDestroy(evk0);
        if (evk1) (void)hipEventDestroy(evk1);
        if (ev_fork) (void)hipEventDestroy(ev_fork);
        if (ev_join) (void)hipEventDestroy(ev_join);
        if (stream2) (void)hipStreamDestroy(stream2);
        if (stream) (void)hipStreamDestroy(stream);
    }
};

namespace gst_impl {

struct LaneLayout {
    std::vector<int32_t> col, kind[2], obj[2], elem[2];
    int32_t n_waves = 0;
};


int finish_create(gst_plan* p, const gst_options* opt, gst_plan** out);
int ensure_device(gst_plan* p);
int upload_model(gst_plan* p);
void base_args(gst_plan* p, gst::WalkArgs& a);
int run_probs(gst_plan* p, double* d_dst, bool fill_cache, int chain_share = 1, const uint32_t* guard = nullptr, bool reassoc = false);
void build_levels_host(gst_plan* p, bool rev, bool probs_only = false);
int ensure_levels(gst_plan* p, bool rev, bool probs_only = false);
void level_args(const gst_plan::Levels& L, gst::LevelArgs& a);
bool levels_wanted(const gst_plan* p, const gst_plan::Levels& L);
int run_levels_forward(gst_plan* p, double* d_dst, bool probs_only = false);
int run_probs_any(gst_plan* p, double* d_dst);
void pack_lanes(const gst_plan* p, const int64_t* param_idx, const int64_t* dest_idx, int64_t n, LaneLayout& L, bool keep_lane63_idle = false);
void pack_waves(const gst_plan* p, const int64_t* param_idx, const int64_t* dest_idx, int64_t n, LaneLayout& L);
int upload_i32(gst_plan* p, DevBuf<int32_t>& b, const std::vector<int32_t>& v);
size_t jac_extent(int64_t n_rows, int64_t ld, const int64_t* dest_idx, int64_t n_param);
int64_t nE_total(const gst_plan* p);
int stage_out(gst_plan* p, size_t count, bool keeps_claims = false);
int check_params(const gst_plan* p, const int64_t* idx, int64_t n);
void fd_items(gst_plan* p, const LaneLayout& L, bool rows, std::vector<std::pair<int32_t, uint32_t>>& items, int32_t& n_units);
int run_dprobs_fd(gst_plan* p, double* d_out, int64_t ld, const int64_t* param_idx, const int64_t* dest_idx, int64_t n_param, double eps, double* d_probs_out, double* d_raw, int64_t ldraw);
int ensure_reverse(gst_plan* p);
int run_dprobs_analytic(gst_plan* p, double* d_out, int64_t ld, const int64_t* param_idx, const int64_t* dest_idx, int64_t n_param, double* d_probs_out);
int copy_out_dprobs(gst_plan* p, double* out, int64_t ld, const int64_t* dest_idx, int64_t n_param, double* probs_out);
int run_models_chunk(gst_plan* p, int64_t nm, int64_t m0, const double* d_base, double* d_out, int64_t ld, const int32_t* d_dest, double eps);
int run_dprobs_models(gst_plan* p, int64_t n_models, const double* gates, const double* rhos, const double* effects, double* d_out, int64_t ld, const int64_t* dest_idx, double eps, double* d_probs_out);
size_t lb_set_stride(const gst_plan* p);
int lb_upload(gst_plan* p);
void lb_args(gst_plan* p, gst::LbArgs& a);
int run_dprobs_lindblad_shared(gst_plan* p, double* d_out, int64_t ld, const int64_t* param_idx, const int64_t* dest_idx, int64_t n_param, double eps, double* d_probs_out);
int run_dprobs_lindblad(gst_plan* p, double* d_out, int64_t ld, const int64_t* param_idx, const int64_t* dest_idx, int64_t n_param, double eps, double* d_probs_out);
int begin_call(gst_plan* p);
int end_call(gst_plan* p, bool sync);
int run_element_jacobian(gst_plan* p, double* d_probs_out);
int run_dprobs_general(gst_plan* p, double* d_out, int64_t ld, const int64_t* param_idx, const int64_t* dest_idx, int64_t n_param, double* d_probs_out);
int run_dprobs_composite(gst_plan* p, double* d_out, int64_t ld, const int64_t* param_idx, const int64_t* dest_idx, int64_t n_param, double eps, double* d_probs_out);
int run_dprobs_composite_analytic(gst_plan* p, double* d_out, int64_t ld, const int64_t* param_idx, const int64_t* dest_idx, int64_t n_param, double* d_probs_out);
int run_dprobs_lindblad_analytic(gst_plan* p, double* d_out, int64_t ld, const int64_t* param_idx, const int64_t* dest_idx, int64_t n_param, double* d_probs_out);

// device address of a host pointer inside a region registered with gst_host_register, or NULL (gst_abi.cpp)
void* mapped_device_pointer(const void* ptr, size_t bytes);
// EVERY copy between device memory and host memory that is not the library's own page-locked allocation goes through these
// (gst_abi.cpp; tests/test_abi_and_rules.py greps the sources for any other hipMemcpy* with a host side).  A destination /
// source inside a region the caller page-locked with gst_host_register is copied directly (asynchronous, on the plan's
// stream); anything else goes through the plan's own page-locked buffers and a host memcpy.  The device never touches pageable
// memory -- the caller's or the library's own heap vectors: the runtime's lock-on-the-fly of arbitrary heap ranges is what a
// rare "Memory access fault ... Reason: Unknown" on a host-heap address was traced to.
//   d2h_bytes / d2h_rows / h2d_bytes  complete on return (staging buffer + memcpy)
//   h2d_async                         the source is consumed on return (copied into the upload ring), the device copy is
//                                     asynchronous on p->stream; sources above half the ring take h2d_bytes
int d2h_bytes(gst_plan* p, void* dst, const void* d_src, size_t bytes);
int h2d_bytes(gst_plan* p, void* d_dst, const void* src, size_t bytes);
int h2d_async(gst_plan* p, void* d_dst, const void* src, size_t bytes);
int d2h_rows(gst_plan* p, double* dst, int64_t dst_ld, const double* d_src, int64_t src_ld, int64_t n_rows, int64_t n_cols);
#define H2D_TRY(p, dst, src, bytes)                                              \
    do {                                                                         \
        int rc_h2d_ = gst_impl::h2d_async((p), (dst), (src), (bytes));           \
        if (rc_h2d_) return rc_h2d_;                                             \
    } while (0)

}  // namespace gst_impl
