// gst_abi.cpp -- the C ABI of include/gstfwd.h on top of the plan compiler and the HIP kernels.
//
// Host-side counterpart of what mapfill_probs_atom / mapfill_dprobs_atom do around the hot loop in
// the reference (mapforwardsim_calc_densitymx.pyx:149-190, 290-383): marshal the plan once (the
// reference re-converts it on EVERY call, :170-181), upload the small model arrays per call, launch.
// There is no CPU compute path in this file: without a usable HIP device every fill fails loudly.
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

namespace {

thread_local std::string g_err;

int fail(int code, const std::string& msg)
{
    g_err = msg;
    return code;
}

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
int g_poison_fill = 0;

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

}  // namespace

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
    int upload_turn = 0;
    bool model_dirty = true;
    // parameter map
    std::vector<int32_t> pkind, pobj, pelem;
    bool have_pmap = false;
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
    const void* ana_zero_out = nullptr; // destination of the last stream-form analytic Jacobian, its leading dimension
    int64_t ana_zero_ld = 0;
    bool ana_zero_valid = false;
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
    DevBuf<int32_t> d_lb_i32;               // kind | obj | n_eff | n_par | n_blocks | blk_type | blk_mode | blk_n
    DevBuf<int64_t> d_lb_i64, d_lb_setparam;   // param0 | term_off | static_off; the stepped parameter of each set
    DevBuf<double> d_lb_statics, d_lb_term_re, d_lb_term_im, d_lb_theta, d_lb_base, d_lb_gates_rm, d_lb_pert;
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

    double last_kernel_ms = 0, last_total_ms = 0;
    int64_t last_launches = 0;
    bool timing = true;         // gst_options.timing: record the HIP events behind gst_stats.last_*_ms

    ~gst_plan()
    {
        if (!dev_ready) return;
        (void)hipSetDevice(device);
        if (d_out.p) gst::track_touch(d_out.p, d_out.n * 8);
        d_lb_i32.release(); d_lb_i64.release(); d_lb_setparam.release(); d_lb_statics.release(); d_lb_term_re.release();
        d_lb_term_im.release(); d_lb_theta.release(); d_lb_base.release(); d_lb_gates_rm.release(); d_lb_pert.release(); d_lb_waves.release(); d_dirty_words.release(); d_dirty_off.release(); d_lb_item_pw.release(); d_jtj_pmask.release(); for (auto& b : d_lbr_lane) b.release(); d_lbr_order.release();
        d_prog.release(); d_block_order.release(); d_obj_part.release(); d_bin_ptr.release(); d_bin_items.release();
        d_bin_head.release(); d_trace.release(); d_ecol_tab.release(); d_ecol_val.release(); d_rprog.release();
        d_rtask_off.release(); d_pos_ptr.release(); d_reff_ptr.release(); d_rev_leaf.release(); d_pair_f.release();
        d_pair_r.release(); d_circ_rho.release(); d_circ_order.release(); d_circ_partner.release();
        d_pair_common.release(); d_rev_cache.release(); d_work_counter.release(); d_range_begin.release();
        d_blk_f1.release(); d_blk_f2.release(); d_blk_r.release(); d_blk_ptr.release();
        d_dv_deriv.release(); d_dv2.release(); d_helem.release(); d_hw.release(); d_hcsc.release();
        d_jelem.release(); d_dv_colmap.release(); d_hscratch.release(); d_dF.release(); d_dB.release();
        d_theta.release(); d_obj_dt.release(); d_obj_ht.release(); d_obj_pc.release(); d_obj_tmp.release();
        d_hess_part.release(); d_hess_out.release(); d_task_off.release(); d_eff_ptr.release();
        d_eff_label.release(); d_eff_dest.release();
        d_model.release();
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

namespace {

int finish_create(gst_plan* p, const gst_options* opt, gst_plan** out)
{
    static std::atomic<uint64_t> next_uid{1};
    p->uid = next_uid.fetch_add(1);
    // default slot budget: what fits in LDS at 4 wavefronts per SIMD (16 per CU): one 8 KB slot at D=16
    int32_t max_slots = opt ? opt->max_slots : 0;
    if (p->hp.D != 4 && p->hp.D != 16 && p->hp.D != 64) {
        const int D = p->hp.D;
        delete p;
        return fail(GST_EUNSUPPORTED, "state dimension " + std::to_string(D) + " not supported (4, 16 or 64)");
    }
    // D = 64: the register-blocked derivative kernel keeps save slots in registers and tracks 2 (plans that ask for
    // more run on the LDS-slot kernels).  D <= 16: the lane-per-model kernel keeps slots in LDS at 8*D*64 bytes each
    // and tracks at most 4.
    if (max_slots <= 0) max_slots = (p->hp.D == 64) ? 2 : (p->hp.D == 16) ? 1 : 4;
    max_slots = std::min(max_slots, p->hp.D == 64 ? 32 : 4);
    std::string err = gst::compile_plan(p->hp, opt ? opt->target_tasks : 0, max_slots);
    if (!err.empty()) { delete p; return fail(GST_EINVAL, err); }
    p->device = opt ? opt->device : -1;
    p->fd_split = opt ? opt->fd_split : 0;
    {   // timing events: auto = only where they are noise (plans that are not launch-bound)
        const int t = opt ? opt->timing : 0;
        p->timing = t == 1 || (t != 2 && p->hp.n_state_ids > 65536);
        if (const char* e = std::getenv("GST_TIMING")) p->timing = std::atoi(e) != 0;
    }
    // GST_TEST_FORCE="key=value,key=value,...": the ONE test hook of the library.  On the small fixtures that carry reference
    // vectors it selects the launch forms that big plans take on their own (and the fall-backs they take under pressure),
    // so that every production form is compared bit for bit with the reference; nothing here is a tuning knob.
    //   persist=0|1|2     FD walk: 0 dispatcher-placed workgroups, 2 per-SIMD queues whatever the number of pairs
    //   fused=0           launch-bound plans keep the separate base pass (instead of the fused base lane)
    //   overlap=0|1       base pass outside / inside the persistent FD launch
    //   handover=0|1|2    never cut a walk / cut to balance the queues (default) / cut every walk that can be cut
    //   skip_chains=1     the overlap launch walks no chain: every bounded wait runs out, the stand-by launches take over
    //   host_direct=0|2   page-locked destinations filled by a copy / by the kernel's own stores at any column count
    //   hess_composed=1   every FD-of-FD Hessian block through the composed route
    //   cache_limit=BYTES stands in for the 4 GB of 32-bit cache offsets (the WIDE contraction kernels)
    //   poison=1          grown device buffers start from 0xFF bytes instead of zeros (reads of unwritten words show)
    if (const char* spec = std::getenv("GST_TEST_FORCE")) {
        std::string str(spec);
        size_t pos = 0;
        while (pos < str.size()) {
            size_t end = str.find(',', pos);
            if (end == std::string::npos) end = str.size();
            const std::string item = str.substr(pos, end - pos);
            pos = end + 1;
            const size_t eq = item.find('=');
            if (eq == std::string::npos) continue;
            const std::string key = item.substr(0, eq);
            const double val = std::atof(item.c_str() + eq + 1);
            const int iv = (int)val;
            if (key == "persist") { p->fd_persist = iv != 0; p->fd_persist_always = iv == 2; }
            else if (key == "fused") p->fd_fused = iv != 0;
            else if (key == "overlap") p->fd_overlap = iv != 0;
            else if (key == "handover") p->fd_handover = iv;
            else if (key == "skip_chains") p->test_skip_chains = iv != 0;
            else if (key == "host_direct") { p->host_direct = iv != 0; if (iv == 2) p->host_direct_min_cols = 1; }
            else if (key == "hess_composed") p->hess_composed = iv != 0;
            else if (key == "cache_limit") p->test_cache_limit = val;
            else if (key == "poison") g_poison_fill = iv ? 0xFF : 0;
            else { delete p; return fail(GST_EINVAL, "GST_TEST_FORCE: unknown key '" + key + "'"); }
        }
    }
    if (p->fd_split != 0 && p->fd_split != 1 && p->fd_split != 2 && p->fd_split != 4) p->fd_split = 0;
    *out = p;
    return GST_OK;
}

// Bring the plan onto the device (lazily, at the first call that needs it).
int ensure_device(gst_plan* p)
{
    if (p->dev_ready) { HIP_TRY(hipSetDevice(p->device)); return GST_OK; }
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return fail(GST_ENODEVICE, std::string("no HIP device available (") + hipGetErrorString(e) +
                                       "); libgstfwd has no CPU fallback");
    if (p->device < 0) { int cur = 0; HIP_TRY(hipGetDevice(&cur)); p->device = cur; }
    if (p->device >= n) return fail(GST_ENODEVICE, "device ordinal out of range");
    HIP_TRY(hipSetDevice(p->device));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, p->device));
    p->n_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(GST_ENODEVICE, std::string("device is ") + prop.gcnArchName + ", this library is built for gfx950 only");
    HIP_TRY(hipStreamCreateWithFlags(&p->stream, hipStreamNonBlocking));
    HIP_TRY(hipEventCreate(&p->ev0)); HIP_TRY(hipEventCreate(&p->ev1));
    HIP_TRY(hipEventCreate(&p->evk0)); HIP_TRY(hipEventCreate(&p->evk1));
    HIP_TRY(hipStreamCreateWithFlags(&p->stream2, hipStreamNonBlocking));
    HIP_TRY(hipEventCreateWithFlags(&p->ev_fork, hipEventDisableTiming)); HIP_TRY(hipEventCreateWithFlags(&p->ev_join, hipEventDisableTiming));
    const gst::HostPlan& h = p->hp;
    HIP_TRY(p->d_prog.ensure(h.prog.size()));
    HIP_TRY(p->d_task_off.ensure(h.task_off.size()));
    HIP_TRY(p->d_eff_ptr.ensure(h.eff_ptr.size()));
    HIP_TRY(p->d_eff_label.ensure(h.eff_label.size()));
    HIP_TRY(p->d_eff_dest.ensure(h.eff_dest.size()));
    HIP_TRY(hipMemcpy(p->d_prog.p, h.prog.data(), h.prog.size() * 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(p->d_task_off.p, h.task_off.data(), h.task_off.size() * 8, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(p->d_eff_ptr.p, h.eff_ptr.data(), h.eff_ptr.size() * 4, hipMemcpyHostToDevice));
    if (!h.eff_label.empty()) {
        HIP_TRY(hipMemcpy(p->d_eff_label.p, h.eff_label.data(), h.eff_label.size() * 4, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(p->d_eff_dest.p, h.eff_dest.data(), h.eff_dest.size() * 4, hipMemcpyHostToDevice));
    }
    HIP_TRY(p->d_pbase.ensure(h.n_elements));
    p->dev_ready = true;
    return GST_OK;
}

int upload_model(gst_plan* p)
{
    const size_t ng = p->h_gates.size(), nr = p->h_rhos.size(), ne = p->h_effects.size();
    const size_t total = 2 * ng + nr + ne;
    if (!p->model_dirty && p->d_model.p) return GST_OK;           // same arrays as the last call: already resident
    HIP_TRY(p->d_model.ensure(std::max<size_t>(total, 1)));
    if (p->h_model_pinned_n < total || !p->ev_upload[0]) {
        HIP_TRY(hipStreamSynchronize(p->stream));
        for (int i = 0; i < 2; i++) {
            if (p->h_model_pinned[i]) (void)hipHostFree(p->h_model_pinned[i]);
            p->h_model_pinned[i] = nullptr;
            HIP_TRY(hipHostMalloc((void**)&p->h_model_pinned[i], std::max<size_t>(total, 1) * 8, hipHostMallocDefault));
            if (!p->ev_upload[i]) HIP_TRY(hipEventCreateWithFlags(&p->ev_upload[i], hipEventDisableTiming));
        }
        p->h_model_pinned_n = total;
    }
    p->d_gates.p = p->d_model.p; p->d_gates_t.p = p->d_model.p + ng; p->d_rhos.p = p->d_model.p + 2 * ng; p->d_effects.p = p->d_model.p + 2 * ng + nr;
    const int turn = p->upload_turn;
    p->upload_turn ^= 1;
    HIP_TRY(hipEventSynchronize(p->ev_upload[turn]));          // (the copy that last used this staging buffer: long done)
    double* h = p->h_model_pinned[turn];
    if (ng) { std::memcpy(h, p->h_gates.data(), ng * 8); std::memcpy(h + ng, p->h_gates_t.data(), ng * 8); }
    std::memcpy(h + 2 * ng, p->h_rhos.data(), nr * 8);
    std::memcpy(h + 2 * ng + nr, p->h_effects.data(), ne * 8);
    if (total) HIP_TRY(hipMemcpyAsync(p->d_model.p, h, total * 8, hipMemcpyHostToDevice, p->stream));
    HIP_TRY(hipEventRecord(p->ev_upload[turn], p->stream));
    p->model_dirty = false;
    return GST_OK;
}

void base_args(gst_plan* p, gst::WalkArgs& a)
{
    std::memset(&a, 0, sizeof(a));
    a.prog = p->d_prog.p; a.task_off = p->d_task_off.p;
    a.eff_ptr = p->d_eff_ptr.p; a.eff_label = p->d_eff_label.p; a.eff_dest = p->d_eff_dest.p;
    a.gates = p->d_gates.p; a.gates_t = p->d_gates_t.p; a.rhos = p->d_rhos.p; a.effects = p->d_effects.p;
    a.n_gates = p->hp.n_gates; a.n_effects = p->hp.n_effects;
    a.n_pwaves = 1;
}

// Base probabilities into d_dst (device), S = 0 walk: one wavefront per task.  With `fill_cache` the
// pass also stores every state it produces (the derivative passes start from them).
// `reassoc`: the caller has no ordering contract (exact derivatives, GST_OPT_FAST_PROBS) -- a D = 64 plan then walks on
// the matrix cores (gst_kernels_chain64.hip), 5x faster per dependent step, results equal up to re-association.
int run_probs(gst_plan* p, double* d_dst, bool fill_cache, int chain_share = 1, const uint32_t* guard = nullptr, bool reassoc = false)
{
    gst::WalkArgs a;
    base_args(p, a);
    a.chain_share = chain_share;
    a.guard = guard;
    a.mode = gst::EMIT_PROBS;
    a.out = d_dst;
    if (fill_cache) {
        HIP_TRY(p->d_base_cache.ensure((size_t)p->hp.n_state_ids * p->hp.D));
        a.base_cache_w = p->d_base_cache.p;
    }
    a.rows_S = 0;
    if (reassoc && p->hp.D == 64 && p->fast_chains && !guard) {
        HIP_TRY(gst::launch_chain64(a, p->hp.n_tasks(), p->hp.max_slots, p->stream));
        p->last_levels = true;
    } else HIP_TRY(gst::launch_walk_rows(p->hp.D, a, p->hp.n_tasks(), p->hp.max_slots, p->stream));
    p->last_launches++;
    return GST_OK;
}

// ---- log-depth chain passes (gst_levels.hpp / gst_kernels_levels.hip) -------------------------------------------------------
int upload_i32(DevBuf<int32_t>& b, const std::vector<int32_t>& v, hipStream_t s);

// The level program of the forward plan (`rev` false) or of the reversed plan, built once on the host (the reversed
// plan's while its state graph still exists: ensure_reverse).
void build_levels_host(gst_plan* p, bool rev, bool probs_only = false)
{
    gst_plan::Levels& L = rev ? p->lv_rev : (probs_only ? p->lv_probs : p->lv_fwd);
    if (L.built) return;
    L.built = true;
    const gst::HostPlan& h = rev ? p->rev : p->hp;
    L.why = gst::build_level_program(h, rev ? p->hp.n_effects : 1, L.prog, probs_only ? &p->hp.circ_leaf : nullptr);
    L.usable = L.why.empty();
}

int ensure_levels(gst_plan* p, bool rev, bool probs_only = false)
{
    gst_plan::Levels& L = rev ? p->lv_rev : (probs_only ? p->lv_probs : p->lv_fwd);
    const gst::HostPlan& h = rev ? p->rev : p->hp;
    build_levels_host(p, rev, probs_only);
    if (!L.usable || L.uploaded) return GST_OK;
    int rc;
    if ((rc = upload_i32(L.d_words, L.prog.words, p->stream))) return rc;
    if ((rc = upload_i32(L.d_ids, L.prog.ids, p->stream))) return rc;
    HIP_TRY(L.d_task_off.ensure(L.prog.task_off.size()));
    HIP_TRY(hipMemcpyAsync(L.d_task_off.p, L.prog.task_off.data(), L.prog.task_off.size() * 8, hipMemcpyHostToDevice, p->stream));
    HIP_TRY(L.d_ids_off.ensure(L.prog.task_ids_off.size()));
    HIP_TRY(hipMemcpyAsync(L.d_ids_off.p, L.prog.task_ids_off.data(), L.prog.task_ids_off.size() * 8, hipMemcpyHostToDevice, p->stream));
    HIP_TRY(L.d_mats.ensure(std::max<size_t>((size_t)h.n_tasks() * (size_t)std::max(L.prog.max_mats, 1) * 256, 1)));
    HIP_TRY(hipStreamSynchronize(p->stream));
    L.uploaded = true;
    return GST_OK;
}

void level_args(const gst_plan::Levels& L, gst::LevelArgs& a)
{
    std::memset(&a, 0, sizeof(a));
    a.words = L.d_words.p; a.task_off = L.d_task_off.p; a.ids = L.d_ids.p; a.task_ids_off = L.d_ids_off.p; a.mats = L.d_mats.p;
    a.nv = L.prog.nv; a.max_mats = std::max(L.prog.max_mats, 1);
    a.stage_lds = L.prog.max_task_ints * 4 <= 60 * 1024 ? 1 : 0;       // (a plan with huger tasks reads its program from memory)
    a.lds_ints = (int32_t)std::min<int64_t>(L.prog.max_task_ints, 15 * 1024);
}

bool levels_wanted(const gst_plan* p, const gst_plan::Levels& L)
{
    return L.usable && (p->fast_chains == 2 || (p->fast_chains == 1 && L.prog.worthwhile));
}

// Every state of the forward trie into the base-state cache by the level pass, then (d_dst != NULL) the probabilities from
// the circuits' final states.  The caller has checked levels_wanted(p, p->lv_fwd).
int run_levels_forward(gst_plan* p, double* d_dst, bool probs_only = false)
{
    const gst::HostPlan& h = p->hp;
    int rc;
    HIP_TRY(p->d_base_cache.ensure((size_t)h.n_state_ids * h.D));
    gst::LevelArgs a;
    level_args(probs_only ? p->lv_probs : p->lv_fwd, a);
    a.bmats = p->d_gates_t.p; a.starts = p->d_rhos.p; a.cache = p->d_base_cache.p;
    HIP_TRY(gst::launch_level_pass(a, h.n_tasks(), p->stream));
    p->last_launches++;
    if (d_dst) {
        if (!p->leaf_uploaded) {
            if ((rc = upload_i32(p->d_circ_leaf, h.circ_leaf, p->stream))) return rc;
            HIP_TRY(hipStreamSynchronize(p->stream));
            p->leaf_uploaded = true;
        }
        HIP_TRY(gst::launch_probs_from_cache(p->d_base_cache.p, p->d_circ_leaf.p, p->d_eff_ptr.p, p->d_eff_label.p, p->d_eff_dest.p,
                                             p->d_effects.p, h.n_circuits, h.D, d_dst, p->stream));
        p->last_launches++;
    }
    p->last_levels = true;
    return GST_OK;
}

// gst_fill_probs*: the sequential walk (bit-identical to the reference), or with GST_OPT_FAST_PROBS the level pass
int run_probs_any(gst_plan* p, double* d_dst)
{
    p->last_levels = false;
    if (p->fast_probs && p->hp.D == 16 && p->fast_chains) {
        int rc = ensure_levels(p, false, true);
        if (rc) return rc;
        if (levels_wanted(p, p->lv_probs)) return run_levels_forward(p, d_dst, true);
    }
    return run_probs(p, d_dst, false, 1, nullptr, p->fast_probs);
}

struct LaneLayout {
    std::vector<int32_t> col, kind[2], obj[2], elem[2];
    int32_t n_waves = 0;
};

// Pack the requested parameter columns into wavefronts of 64 lanes.  Lanes of one wavefront should
// perturb the same gate (the special-row work is paid per gate per wavefront): SPAM parameters are
// packed together, each gate's parameters start on a wavefront boundary when the gate has >= 32
// requested parameters.
void pack_lanes(const gst_plan* p, const int64_t* param_idx, const int64_t* dest_idx, int64_t n, LaneLayout& L,
                bool keep_lane63_idle = false)
{
    struct Item { int32_t kind, obj, elem, col; };
    std::vector<Item> spam, none;
    std::vector<std::vector<Item>> per_gate(p->hp.n_gates);
    for (int64_t c = 0; c < n; c++) {
        const int64_t pi = param_idx[c];
        Item it{p->pkind[pi], p->pobj[pi], p->pelem[pi], (int32_t)(dest_idx ? dest_idx[c] : c)};
        if (it.kind == GST_KIND_GATE) per_gate[it.obj].push_back(it);
        else if (it.kind == GST_KIND_NONE) none.push_back(it);
        else spam.push_back(it);
    }
    auto idle = [&]() { L.col.push_back(-1); L.kind[0].push_back(GST_KIND_NONE); L.obj[0].push_back(0); L.elem[0].push_back(0); };
    auto push = [&](const Item& it) {
        if (keep_lane63_idle && L.col.size() % 64 == 63) idle();       // (fused base lane: see WalkArgs::fused)
        L.col.push_back(it.col); L.kind[0].push_back(it.kind); L.obj[0].push_back(it.obj); L.elem[0].push_back(it.elem);
    };
    auto pad = [&]() {
        while (L.col.size() % 64) { L.col.push_back(-1); L.kind[0].push_back(GST_KIND_NONE); L.obj[0].push_back(0); L.elem[0].push_back(0); }
    };
    for (auto& it : spam) push(it);
    for (auto& it : none) push(it);
    for (auto& g : per_gate) {
        if (g.size() >= 32) pad();
        for (auto& it : g) push(it);
    }
    if (keep_lane63_idle && L.col.empty()) idle();
    pad();
    L.n_waves = (int32_t)(L.col.size() / 64);
}

// Row-per-lane kernel: one wavefront per requested column.
void pack_waves(const gst_plan* p, const int64_t* param_idx, const int64_t* dest_idx, int64_t n, LaneLayout& L)
{
    for (int64_t c = 0; c < n; c++) {
        const int64_t pi = param_idx[c];
        L.col.push_back((int32_t)(dest_idx ? dest_idx[c] : c));
        L.kind[0].push_back(p->pkind[pi]); L.obj[0].push_back(p->pobj[pi]); L.elem[0].push_back(p->pelem[pi]);
    }
    L.n_waves = (int32_t)n;
}

int upload_i32(DevBuf<int32_t>& b, const std::vector<int32_t>& v, hipStream_t s)
{
    HIP_TRY(b.ensure(v.size()));
    if (!v.empty()) HIP_TRY(hipMemcpyAsync(b.p, v.data(), v.size() * 4, hipMemcpyHostToDevice, s));
    return GST_OK;
}

// bytes from a Jacobian destination's first entry to one past its last: rows of `ld` doubles, columns dest_idx (or 0 .. n - 1)
size_t jac_extent(int64_t n_rows, int64_t ld, const int64_t* dest_idx, int64_t n_param)
{
    if (n_rows <= 0 || n_param <= 0) return 0;
    int64_t max_col = n_param - 1;
    if (dest_idx) { max_col = 0; for (int64_t c = 0; c < n_param; c++) max_col = std::max(max_col, dest_idx[c]); }
    return (size_t)((n_rows - 1) * ld + max_col + 1) * 8;
}
int64_t nE_total(const gst_plan* p) { return p->hp.n_elements; }

// The plan's staging buffer for host destinations.  Every use but an exact Jacobian overwrites the zeros a previous one
// may have left there (gst_track.cpp); so does growing it.
int stage_out(gst_plan* p, size_t count, bool keeps_claims = false)
{
    if (p->d_out.p && (!keeps_claims || count > p->d_out.n)) gst::track_touch(p->d_out.p, p->d_out.n * 8);
    HIP_TRY(p->d_out.ensure(count));
    return GST_OK;
}

int check_params(const gst_plan* p, const int64_t* idx, int64_t n)
{
    if (n < 0) return fail(GST_EINVAL, "negative parameter count");
    if (n > 0 && !idx) return fail(GST_EINVAL, "param_idx is NULL");
    for (int64_t c = 0; c < n; c++)
        if (idx[c] < 0 || idx[c] >= (int64_t)p->pkind.size()) return fail(GST_EINVAL, "parameter index out of range");
    return GST_OK;
}

// Estimated cost of every (task, wavefront) pair of an FD request, longest first: a wavefront's work in a task is set
// by the objects its lanes perturb (a gate the task never applies costs almost nothing, a gate of the germ costs the
// whole chain; gst::task_gate_costs).  D = 64: the unit is a group of rows_group() consecutive wavefronts of one task.
void fd_items(gst_plan* p, const LaneLayout& L, bool rows, std::vector<std::pair<int32_t, uint32_t>>& items, int32_t& n_units)
{
    const int nG = p->hp.n_gates;
    const int64_t nT = p->hp.n_tasks();
    const int stride = nG + 2;
    std::vector<uint64_t> wave_gates(L.n_waves, 0);
    std::vector<uint8_t> wave_rho(L.n_waves, 0);
    for (size_t q = 0; q < L.col.size(); q++) {
        if (L.col[q] < 0) continue;
        const size_t w = rows ? q : q / 64;
        if (L.kind[0][q] == GST_KIND_GATE) wave_gates[w] |= 1ull << L.obj[0][q];
        else if (L.kind[0][q] == GST_KIND_RHO) wave_rho[w] = 1;
    }
    const int32_t grp = rows ? gst::rows_group(p->hp.D, p->hp.max_slots) : 1;
    n_units = (L.n_waves + grp - 1) / grp;
    items.clear();
    items.reserve((size_t)nT * n_units);
    for (int64_t t = 0; t < nT; t++) {
        const int32_t* c = p->task_cost.data() + (size_t)t * stride;
        for (int32_t u = 0; u < n_units; u++) {
            int32_t best = c[nG + 1] / 4;
            for (int32_t w = u * grp; w < std::min<int32_t>((u + 1) * grp, L.n_waves); w++) {
                if (wave_rho[w]) best = std::max(best, c[nG + 1] / 4 + c[nG]);
                else
                    for (uint64_t m = wave_gates[w]; m; m &= m - 1) best = std::max(best, c[nG + 1] / 4 + c[__builtin_ctzll(m)]);
            }
            items.emplace_back(-best, (uint32_t)(t * n_units + u));
        }
    }
    std::stable_sort(items.begin(), items.end(), [](const auto& x, const auto& y) { return x.first < y.first; });
}

// FD Jacobian columns into device memory.  d_raw (optional) receives the perturbed probabilities.
int run_dprobs_fd(gst_plan* p, double* d_out, int64_t ld, const int64_t* param_idx, const int64_t* dest_idx,
                  int64_t n_param, double eps, double* d_probs_out, double* d_raw, int64_t ldraw)
{
    // base probabilities (pyx:349)
    double* d_base = d_probs_out ? d_probs_out : p->d_pbase.p;
    const bool rows = (p->hp.D == 64);
    // Launch-bound plans (1Q fits: a fill is two ~30 us chains behind each other): ONE launch -- lane 63 of every
    // wavefront walks the unperturbed model, so neither the base pass nor its state cache is needed.  Not for
    // complement effects (their columns are evaluated on the cached final states) and not for the Hessian driver's
    // passes (d_raw), which reuse the base pass's by-products.
    const bool fused = !rows && n_param > 0 && p->hp.n_state_ids <= 65536 && p->comp_index < 0 && !d_raw && p->fd_fused;
    int rc = GST_OK;
    p->last_overlap = false;
    p->last_fd_form = 0;
    if (n_param == 0) return run_probs(p, d_base, false);
    // (the base pass itself is enqueued below, once the launch form is known: the persistent launch of a small atom runs
    //  it inside its own kernel)
    if (p->cached_fused != fused) p->cached_kind = 0;
    if (!p->request_cached(1, param_idx, dest_idx, n_param)) {
        // (an optimizer asks for the same columns every iteration: pack and upload the lane tables once)
        p->cached_kind = 0;            // nothing below may leave a half-updated request looking cached when it fails
        LaneLayout L;
        // a declared complement effect: effect-parameter columns leave the walk (effect_fd_kernel below does them on
        // the cached final states, where the recomputed complement can be substituted)
        std::vector<int64_t> w_idx, w_dest;
        p->ecol_tab.clear();
        if (p->comp_index >= 0) {
            std::vector<int32_t> eo, ee, ed, et;
            for (int64_t c = 0; c < n_param; c++) {
                const int64_t pi = param_idx[c];
                const int64_t dst = dest_idx ? dest_idx[c] : c;
                if (p->pkind[pi] != GST_KIND_EFFECT) { w_idx.push_back(pi); w_dest.push_back(dst); continue; }
                const int32_t e = p->pobj[pi];
                if (e == p->comp_index) return fail(GST_EINVAL, "a parameter maps to the complement effect");
                eo.push_back(e); ee.push_back(p->pelem[pi]); ed.push_back((int32_t)dst);
                et.push_back(std::find(p->comp_others.begin(), p->comp_others.end(), e) != p->comp_others.end() ? 1 : 0);
            }
            p->ecol_tab.insert(p->ecol_tab.end(), eo.begin(), eo.end());
            p->ecol_tab.insert(p->ecol_tab.end(), ee.begin(), ee.end());
            p->ecol_tab.insert(p->ecol_tab.end(), ed.begin(), ed.end());
            p->ecol_tab.insert(p->ecol_tab.end(), et.begin(), et.end());
            if ((rc = upload_i32(p->d_ecol_tab, p->ecol_tab, p->stream))) return rc;
        }
        const bool filtered = p->comp_index >= 0;
        const int64_t* l_idx = filtered ? w_idx.data() : param_idx;
        const int64_t* l_dest = filtered ? w_dest.data() : dest_idx;
        const int64_t l_n = filtered ? (int64_t)w_idx.size() : n_param;
        if (rows) pack_waves(p, l_idx, l_dest, l_n, L);     // one perturbed model per wavefront
        else pack_lanes(p, l_idx, l_dest, l_n, L, fused);
        p->cached_kind = 0;
        p->cached_fused = fused;
        if ((rc = upload_i32(p->d_lane[0], L.col, p->stream))) return rc;
        if ((rc = upload_i32(p->d_lane[1], L.kind[0], p->stream))) return rc;
        if ((rc = upload_i32(p->d_lane[2], L.obj[0], p->stream))) return rc;
        if ((rc = upload_i32(p->d_lane[3], L.elem[0], p->stream))) return rc;
        // Launch order: longest (task, wavefront) pairs first.  A wavefront's work in a task is set by the objects
        // its lanes perturb (a gate the task never applies costs almost nothing, a gate of the germ costs the whole
        // chain), so with only a few pairs per SIMD -- a 1/8 atom of the 2Q design has 4.4 -- the order decides how
        // long the last SIMD runs.
        p->have_block_order = false;
        const int64_t nT = p->hp.n_tasks();
        if (p->task_cost.empty()) gst::task_gate_costs(p->hp, p->task_cost);
        if (!p->task_cost.empty() && nT * L.n_waves < 0x7fffffffLL && nT * L.n_waves > 1) {
            std::vector<std::pair<int32_t, uint32_t>> items;       // (-cost, task * n_units + unit), longest first
            int32_t n_units = 0;
            fd_items(p, L, rows, items, n_units);
            std::vector<uint32_t> order(items.size());
            for (size_t i = 0; i < items.size(); i++) order[i] = items[i].second;
            HIP_TRY(p->d_block_order.ensure(order.size()));
            HIP_TRY(hipMemcpyAsync(p->d_block_order.p, order.data(), order.size() * 4, hipMemcpyHostToDevice, p->stream));
            HIP_TRY(hipStreamSynchronize(p->stream));
            p->have_block_order = true;
            if (const char* tp = std::getenv("GST_FD_TRACE")) {          // development aid: the estimates next to the trace
                if (FILE* f = std::fopen((std::string(tp) + ".cost").c_str(), "wb")) {
                    for (auto& it : items) { int64_t r[2] = {(int64_t)it.second, (int64_t)-it.first}; std::fwrite(r, 8, 2, f); }
                    std::fclose(f);
                }
            }
            // Persistent launch (D <= 16): pack the pairs into one queue per SIMD with equal estimated work -- longest
            // first, each into the least loaded queue -- instead of leaving the placement to the dispatcher.
            // Measured on MI355X (2Q design, kernel ms, interleaved repeats, queues vs dispatcher): 1/8 atom 4.54 vs
            // 4.63, 1/4 atom 7.17 vs 7.37, 1/2 atom 14.3 vs 13.6 -- with many pairs per SIMD the dispatcher's dynamic
            // placement is as good or better, so the queues are used below 10 pairs per SIMD -- and not below one pair
            // per SIMD, where a fill is launch-bound and the extra memset and pops cost 10 us (1Q L<=128: 100 vs 110 us).
            p->have_bins = false;
            if (!rows && p->fd_persist && p->hp.max_slots <= 4 && items.size() < (1u << 30) && (p->fd_persist_always || (items.size() <= (size_t)40 * p->n_cus && items.size() >= (size_t)4 * p->n_cus)) &&
                (size_t)16 * std::max(p->hp.max_slots, 1) * p->hp.D * 64 * 8 <= 160 * 1024) {
                const int n_bins = 4 * p->n_cus;
                if (p->fd_handover != 0 && !p->split_ready) {
                    // cuts only where no save slot is live and the interpreter's outer loop stands (an EMIT, the word after a
                    // LOAD), the one nearest a walk's middle.  (Cuts at any position with live slots travelling along balance the
                    // ESTIMATED loads to 2 % and make the kernel slower -- 4.45 against 4.05 ms on a 1/8 atom: a second part
                    // popped before its first part is done occupies a wavefront.  Measured in round 3, removed in round 4.)
                    gst::task_split_candidates(p->hp, p->cand_ptr, p->cand_pc, p->cand_frac, 1 << 20, nullptr);
                    p->cand_live.clear();
                    p->split_ready = true;
                }
                gst::FdQueues Q;
                gst::pack_fd_queues(items, n_units, nT, n_bins, p->fd_handover, p->cand_ptr, p->cand_pc, p->cand_frac, p->cand_live, Q);
                const int32_t n_split = Q.n_split;
                const std::vector<int32_t>& bptr = Q.bin_ptr;
                const std::vector<uint32_t>& bitems = Q.bin_items;
                const std::vector<int32_t>& ho_index = Q.ho_index;
                const std::vector<int32_t>& ho_pc = Q.ho_pc;
                std::vector<int32_t> ho_live(Q.ho_live.begin(), Q.ho_live.end());
                if ((rc = upload_i32(p->d_bin_ptr, bptr, p->stream))) return rc;
                HIP_TRY(p->d_bin_items.ensure(bitems.size()));
                HIP_TRY(hipMemcpyAsync(p->d_bin_items.p, bitems.data(), bitems.size() * 4, hipMemcpyHostToDevice, p->stream));
                HIP_TRY(p->d_bin_head.ensure((size_t)n_bins + 1));            // (+ the abort flag)
                p->n_split = n_split;
                if (n_split > 0) {
                    if ((rc = upload_i32(p->d_ho_index, ho_index, p->stream))) return rc;
                    if ((rc = upload_i32(p->d_task_split_pc, ho_pc, p->stream))) return rc;
                    if ((rc = upload_i32(p->d_ho_live, ho_live, p->stream))) return rc;
                    HIP_TRY(p->d_ho_state.ensure((size_t)n_split * (size_t)(1 + std::max(p->hp.max_slots, 0)) * p->hp.D * 64));
                    HIP_TRY(p->d_ho_tag.ensure((size_t)n_split * 4));
                    HIP_TRY(p->d_ho_id.ensure((size_t)n_split));
                    HIP_TRY(p->d_ho_flag.ensure((size_t)n_split));
                }
                HIP_TRY(hipStreamSynchronize(p->stream));
                p->n_bins = n_bins;
                p->have_bins = true;
            }
        }
        HIP_TRY(hipStreamSynchronize(p->stream));       // the host vectors go out of scope
        p->remember_request(1, param_idx, dest_idx, n_param);
        p->cached_n_waves = L.n_waves;
    }
    // ---- the base pass ----------------------------------------------------------------------------------------------
    // Persistent launches (small atoms, D = 16) walk the base chains INSIDE the FD kernel: ~0.45 ms of pure latency that
    // nothing overlapped (a 1/8 atom of the 2Q design: 4.4 ms per step).  Needs the chain kernel's tables in LDS next to
    // the walks' save slots and at most two chains per workgroup.
    const int split_req = (p->fd_split == 2 || p->fd_split == 4) && p->hp.D == 16 ? p->fd_split : 1;
    const bool persist = !rows && split_req == 1 && p->have_bins && p->have_block_order && p->cached_n_waves > 0;
    bool overlap = false;
    int32_t ovl_chain_doubles = 0;
    if (persist && !fused && p->fd_overlap && p->hp.D == 16 && p->comp_index < 0 && !d_raw &&
        gst::chain_kernel_fits(p->hp.D, p->hp.n_gates, p->hp.n_effects, p->hp.max_slots)) {
        const int waves = gst::persistent_waves(p->hp.D);
        const int64_t chains = (p->hp.n_tasks() + p->n_cus - 1) / p->n_cus;
        ovl_chain_doubles = (int32_t)((gst::chain_lds_doubles(p->hp.D, p->hp.n_gates, p->hp.n_effects, p->hp.max_slots) + 1) & ~(size_t)1);
        const size_t lds = ((size_t)waves * std::max(p->hp.max_slots, 1) * p->hp.D * 64 + (size_t)chains * ovl_chain_doubles) * 8;
        overlap = chains <= 2 && lds <= 160 * 1024;
    }
    if (!fused && (!overlap || p->fd_overlap_diag) && (rc = run_probs(p, d_base, true))) return rc;
    gst::WalkArgs a;
    base_args(p, a);
    a.mode = gst::EMIT_FD;
    a.out = d_out; a.ld = ld; a.eps = eps; a.pbase = d_base;
    a.raw = d_raw; a.ldraw = ldraw;
    a.base_cache = p->d_base_cache.p;
    a.fused = fused ? 1 : 0; a.probs_out = fused ? d_base : nullptr;
    a.lanes.col = p->d_lane[0].p; a.lanes.kind[0] = p->d_lane[1].p; a.lanes.obj[0] = p->d_lane[2].p; a.lanes.elem[0] = p->d_lane[3].p;
    a.n_pwaves = p->cached_n_waves;
    a.block_order = p->have_block_order ? p->d_block_order.p : nullptr;
    const char* trace_path = std::getenv("GST_FD_TRACE");          // development aid: per-pair timestamps (tools/trace_stats.py)
    const size_t n_trace = (size_t)p->hp.n_tasks() * ((size_t)std::max(p->cached_n_waves, 1) + 1) + (size_t)std::max(p->n_split, 0);
    if (trace_path && !rows) {
        HIP_TRY(p->d_trace.ensure(1 + 4 * n_trace));
        HIP_TRY(hipMemsetAsync(p->d_trace.p, 0, 8, p->stream));
        a.trace = (unsigned long long*)p->d_trace.p;
    }
    TIME_REC(p, evk0);
    if (p->comp_index >= 0 && !p->ecol_tab.empty()) {
        const int D = p->hp.D;
        const int32_t nc = (int32_t)(p->ecol_tab.size() / 4);
        // this call's perturbed values: theta + eps for the effect itself; identity - sum(others), the others summed
        // from 0 in the declared order (Python's sum()), for the complement (complementeffect.py:72-78)
        p->ecol_val.assign((size_t)2 * nc, 0.0);
        for (int32_t k = 0; k < nc; k++) {
            const int32_t e = p->ecol_tab[k], i = p->ecol_tab[nc + k];
            const double own = p->h_effects[(size_t)e * D + i] + eps;
            double sum = 0.0;
            for (int32_t o : p->comp_others) sum = sum + (o == e ? own : p->h_effects[(size_t)o * D + i]);
            p->ecol_val[k] = own;
            p->ecol_val[nc + k] = p->comp_identity[i] - sum;
        }
        HIP_TRY(p->d_ecol_val.ensure(p->ecol_val.size()));
        HIP_TRY(hipMemcpyAsync(p->d_ecol_val.p, p->ecol_val.data(), p->ecol_val.size() * 8, hipMemcpyHostToDevice, p->stream));
        if (!p->leaf_uploaded) {
            if ((rc = upload_i32(p->d_circ_leaf, p->hp.circ_leaf, p->stream))) return rc;
            p->leaf_uploaded = true;
        }
        gst::EffectFDArgs ea;
        std::memset(&ea, 0, sizeof(ea));
        ea.n_circuits = p->hp.n_circuits; ea.n_cols = nc; ea.D = D; ea.comp_index = p->comp_index;
        ea.circ_leaf = p->d_circ_leaf.p; ea.eff_ptr = p->d_eff_ptr.p; ea.eff_label = p->d_eff_label.p; ea.eff_dest = p->d_eff_dest.p;
        ea.effects = p->d_effects.p; ea.base_cache = p->d_base_cache.p; ea.pbase = d_base;
        ea.col_obj = p->d_ecol_tab.p; ea.col_elem = p->d_ecol_tab.p + nc; ea.col_dest = p->d_ecol_tab.p + 2 * nc;
        ea.col_touches_comp = p->d_ecol_tab.p + 3 * nc;
        ea.col_own = p->d_ecol_val.p; ea.col_comp = p->d_ecol_val.p + nc;
        ea.out = d_out; ea.ld = ld; ea.raw = d_raw; ea.ldraw = ldraw; ea.eps = eps;
        HIP_TRY(gst::launch_effect_fd(ea, p->stream));
        p->last_launches++;
    }
    if (a.n_pwaves == 0) {
        // (every requested column was an effect parameter)
    } else if (rows) {
        a.rows_S = 1;
        HIP_TRY(gst::launch_walk_rows(p->hp.D, a, p->hp.n_tasks(), p->hp.max_slots, p->stream));
    } else {
        // gst_options.fd_split > 1 splits every (task, 64 columns) pair's rows over 2 or 4 wavefronts (walk_kernel's
        // NW).  Bit-identical, but measured on MI355X it costs 1.35-1.45x the SIMD time per pair (one barrier per gate
        // application) and that cancels the balance it buys on a 1/8 atom (4.63 vs 4.70 ms), so "auto" is 1.
        const int split = split_req;
        if (persist) {
            // persistent launch: one workgroup per CU, pairs popped from the per-SIMD queues
            gst::WalkArgs sb = a;                  // (the stand-by launches' arguments: the plain dispatcher-placed form)
            a.bin_ptr = p->d_bin_ptr.p; a.bin_items = p->d_bin_items.p; a.bin_head = p->d_bin_head.p; a.n_bins = p->n_bins;
            if (p->n_split > 0) {
                a.ho_pc = p->d_task_split_pc.p; a.ho_index = p->d_ho_index.p; a.ho_state = p->d_ho_state.p;
                a.ho_id = p->d_ho_id.p; a.ho_flag = p->d_ho_flag.p;
                a.ho_live = (const uint32_t*)p->d_ho_live.p; a.ho_tag = p->d_ho_tag.p; a.ho_blocks = 1 + std::max(p->hp.max_slots, 0);
                HIP_TRY(hipMemsetAsync(p->d_ho_flag.p, 0, (size_t)p->n_split * 4, p->stream));
            }
            a.lds_wave_doubles = std::max(p->hp.max_slots, 1) * p->hp.D * 64;
            // queue heads, and behind them the abort flag of the bounded waits (hand-over, overlap)
            HIP_TRY(hipMemsetAsync(p->d_bin_head.p, 0, ((size_t)p->n_bins + 1) * 4, p->stream));
            uint32_t* const d_abort = p->d_bin_head.p + p->n_bins;
            const bool can_wait = p->n_split > 0 || overlap;
            a.abort_flag = can_wait ? d_abort : nullptr;
            if (overlap) {
                HIP_TRY(p->d_base_cache.ensure((size_t)p->hp.n_state_ids * p->hp.D));
                a.base_cache = p->d_base_cache.p;
                // consumers tell "not produced yet" from a value by this bit pattern (gst_chain.hpp)
                if (!p->fd_overlap_diag) {
                    HIP_TRY(hipMemsetD32Async((hipDeviceptr_t)d_base, (int)gst::OVL_SENTINEL32, (size_t)p->hp.n_elements * 2, p->stream));
                    HIP_TRY(hipMemsetD32Async((hipDeviceptr_t)p->d_base_cache.p, (int)gst::OVL_SENTINEL32, (size_t)p->hp.n_state_ids * p->hp.D * 2, p->stream));
                }
                a.ovl_n_tasks = (int32_t)p->hp.n_tasks();
                a.ovl_chain_doubles = ovl_chain_doubles;
                a.pbase_w = d_base;
                a.ovl_test_skip = (p->test_skip_chains || p->fd_overlap_diag) ? 1 : 0;
                p->last_overlap = true;
            }
            HIP_TRY(gst::launch_walk_persistent(p->hp.D, a, p->n_cus, p->hp.max_slots, p->stream));
            p->last_fd_form = overlap ? 2 : 1;
            if (can_wait && p->fd_standby) {
                // Stand-by launches: the same work in the form that waits for nothing -- separate base pass, one workgroup
                // per pair, no hand-over -- guarded by the abort flag: every workgroup leaves at once unless a bounded
                // wait of the launch above ran out (its producer not resident: a shared device).  Costs two empty
                // launches per fill; buys "never hangs, never returns a half-written Jacobian" without a host round trip.
                if (overlap) {
                    if ((rc = run_probs(p, d_base, true, 1, d_abort))) return rc;
                    p->last_launches--;            // (counted below, once, like every FD fill)
                }
                sb.base_cache = p->d_base_cache.p;
                sb.guard = d_abort;
                HIP_TRY(gst::launch_walk(p->hp.D, 1, sb, p->hp.n_tasks(), p->hp.max_slots, p->stream, 1));
            }
        } else
            HIP_TRY(gst::launch_walk(p->hp.D, 1, a, p->hp.n_tasks(), p->hp.max_slots, p->stream, split));
    }
    TIME_REC(p, evk1);
    p->last_launches++;
    if (a.trace) {
        std::vector<uint64_t> h(1 + 4 * n_trace);
        HIP_TRY(hipMemcpyAsync(h.data(), p->d_trace.p, h.size() * 8, hipMemcpyDeviceToHost, p->stream));
        HIP_TRY(hipStreamSynchronize(p->stream));
        if (FILE* f = std::fopen(trace_path, "wb")) { std::fwrite(h.data(), 8, h.size(), f); std::fclose(f); }
    }
    return GST_OK;
}

// Analytic mode, D = 16: reversed plan + pair tables, built and uploaded once per plan.
int ensure_reverse(gst_plan* p)
{
    if (p->rev_ready) return GST_OK;
    const gst::HostPlan& h = p->hp;
    const int32_t rev_tasks = 0;
    std::string err = gst::build_reverse_plan(h, p->rev, rev_tasks, h.D == 16 ? 1 : (h.D == 64 ? 8 : 4));
    if (!err.empty()) return fail(GST_EINVAL, "reversed plan: " + err);
    if (p->rev.max_slots > (h.D == 64 ? 32 : 4)) return fail(GST_EUNSUPPORTED, "reversed plan needs too many save slots");
    std::vector<int32_t> pf, pr;
    std::vector<int64_t> pos_ptr;
    gst::build_pair_tables(h, p->rev, pf, pr, pos_ptr);
    HIP_TRY(p->d_rprog.ensure(p->rev.prog.size() + 64));
    HIP_TRY(hipMemsetAsync(p->d_rprog.p, 0, (p->rev.prog.size() + 64) * 4, p->stream));
    HIP_TRY(hipMemcpyAsync(p->d_rprog.p, p->rev.prog.data(), p->rev.prog.size() * 4, hipMemcpyHostToDevice, p->stream));
    HIP_TRY(p->d_rtask_off.ensure(p->rev.task_off.size()));
    HIP_TRY(hipMemcpyAsync(p->d_rtask_off.p, p->rev.task_off.data(), p->rev.task_off.size() * 8, hipMemcpyHostToDevice, p->stream));
    HIP_TRY(p->d_pos_ptr.ensure(pos_ptr.size()));
    HIP_TRY(hipMemcpyAsync(p->d_pos_ptr.p, pos_ptr.data(), pos_ptr.size() * 8, hipMemcpyHostToDevice, p->stream));
    int rc;
    std::vector<int32_t> zeros((size_t)h.n_circuits + 1, 0);
    if ((rc = upload_i32(p->d_reff_ptr, zeros, p->stream))) return rc;
    if ((rc = upload_i32(p->d_rev_leaf, p->rev.circ_leaf, p->stream))) return rc;
    if ((rc = upload_i32(p->d_pair_f, pf, p->stream))) return rc;
    if ((rc = upload_i32(p->d_pair_r, pr, p->stream))) return rc;
    if ((rc = upload_i32(p->d_circ_rho, h.circ_rho, p->stream))) return rc;
    std::vector<int32_t> order((size_t)h.n_circuits);
    for (int64_t c = 0; c < h.n_circuits; c++) order[(size_t)c] = (int32_t)c;
    std::stable_sort(order.begin(), order.end(), [&](int32_t x, int32_t y) { return p->rev.circ_leaf[x] < p->rev.circ_leaf[y]; });
    const int nG = h.n_gates;
    // applications of every gate that two circuits have in common at their ends (equal backward-state ids)
    auto common_tail = [&](int32_t c, int32_t c2, int32_t* per_gate) {
        int64_t common = 0;
        for (int g = 0; g < nG; g++) {
            const int64_t a0 = pos_ptr[(size_t)c * nG + g], a1 = pos_ptr[(size_t)c * nG + g + 1];
            const int64_t b0 = pos_ptr[(size_t)c2 * nG + g], b1 = pos_ptr[(size_t)c2 * nG + g + 1];
            int64_t n = 0;
            while (n < a1 - a0 && n < b1 - b0 && pr[(size_t)(a1 - 1 - n)] == pr[(size_t)(b1 - 1 - n)]) n++;
            if (per_gate) per_gate[g] = (int32_t)n;
            common += n;
        }
        return common;
    };
    auto similar = [&](int32_t c, int32_t c2, int64_t common) {
        const int64_t longer = std::max(h.circ_ptr[c + 1] - h.circ_ptr[c], h.circ_ptr[c2 + 1] - h.circ_ptr[c2]);
        return common >= 8 && 2 * common >= longer;
    };
    if (h.D == 16 && p->ana_germ_order && h.n_circuits > 1) {
        // Locality of the FORWARD states.  Pure suffix order keeps the backward chains of neighbours together but walks
        // through every prefix family (preparation fiducial x germ) for each measurement fiducial and germ power, so the
        // forward chains -- 128 bytes per application of every item -- never stay in an XCD's 4 MB L2.  Runs of
        // neighbours that end alike (one germ power and measurement fiducial behind all the preparation fiducials)
        // are kept whole, and the runs are ordered by the forward-trie family their first member belongs to (the root
        // of its state's parent chain = the task of the forward plan): all the runs of one germ become consecutive, their
        // 16 forward chains (2 MB) stay in L2 while the germ's backward chains stream through once.
        std::vector<int32_t> root((size_t)h.n_state_ids, -2);
        auto root_of = [&](int32_t id) {
            int32_t r = id;
            while (root[(size_t)r] == -2 && h.node_parent[(size_t)r] >= 0) r = h.node_parent[(size_t)r];
            const int32_t top = root[(size_t)r] == -2 ? r : root[(size_t)r];
            for (int32_t q = id; q != r; q = h.node_parent[(size_t)q]) root[(size_t)q] = top;
            root[(size_t)r] = top;
            return top;
        };
        std::vector<int32_t> run_of((size_t)h.n_circuits, 0), run_key;
        int32_t run = 0;
        run_key.push_back(root_of(h.circ_leaf[(size_t)order[0]]));
        for (int64_t k = 1; k < h.n_circuits; k++) {
            const int32_t c = order[(size_t)k - 1], c2 = order[(size_t)k];
            if (!similar(c, c2, common_tail(c, c2, nullptr))) { run++; run_key.push_back(0x7fffffff); }
            run_of[(size_t)k] = run;
            run_key[(size_t)run] = std::min(run_key[(size_t)run], root_of(h.circ_leaf[(size_t)c2]));
        }
        std::vector<int32_t> posn((size_t)h.n_circuits);
        for (int64_t k = 0; k < h.n_circuits; k++) posn[(size_t)k] = (int32_t)k;
        std::stable_sort(posn.begin(), posn.end(), [&](int32_t x, int32_t y) { return run_key[(size_t)run_of[(size_t)x]] < run_key[(size_t)run_of[(size_t)y]]; });
        std::vector<int32_t> reordered((size_t)h.n_circuits);
        for (int64_t k = 0; k < h.n_circuits; k++) reordered[(size_t)k] = order[(size_t)posn[(size_t)k]];
        order.swap(reordered);
    }
    // Work items of the D = 16 contraction: a circuit, or TWO neighbours of the suffix order whose last applications
    // coincide (same germ power and measurement fiducial behind different preparation fiducials): over the common
    // tail their backward states are the same vectors and the kernel gathers them once for both.
    std::vector<int32_t> item_first, item_partner, item_common;
    const bool pairing = h.D == 16 && h.n_effects == 4 && p->ana_pairs;
    auto plain4 = [&](int32_t c) {
        if (h.eff_ptr[c + 1] - h.eff_ptr[c] != 4) return false;
        for (int x = 0; x < 4; x++) if (h.eff_label[(size_t)h.eff_ptr[c] + x] != x) return false;
        return true;
    };
    for (int64_t k = 0; k < h.n_circuits; k++) {
        const int32_t c = order[(size_t)k];
        bool paired = false;
        if (pairing && k + 1 < h.n_circuits) {
            const int32_t c2 = order[(size_t)k + 1];
            if (plain4(c) && plain4(c2)) {
                std::vector<int32_t> cg((size_t)nG, 0);
                const int64_t common = common_tail(c, c2, cg.data());
                if (similar(c, c2, common)) {
                    item_first.push_back(c); item_partner.push_back(c2);
                    item_common.insert(item_common.end(), cg.begin(), cg.end());
                    paired = true;
                    k++;
                }
            }
        }
        if (!paired) {
            item_first.push_back(c); item_partner.push_back(-1);
            item_common.insert(item_common.end(), (size_t)nG, 0);
        }
    }
    const int64_t n_items = (int64_t)item_first.size();
    if (h.D == 16) {
        if ((rc = upload_i32(p->d_circ_order, item_first, p->stream))) return rc;
        if ((rc = upload_i32(p->d_circ_partner, item_partner, p->stream))) return rc;
        if ((rc = upload_i32(p->d_pair_common, item_common, p->stream))) return rc;
    } else {
        if ((rc = upload_i32(p->d_circ_order, order, p->stream))) return rc;
    }
    // 8 contiguous ranges of the item list with equal numbers of gate applications (+ a constant per circuit)
    std::vector<uint32_t> range_begin(9, 0);
    {
        auto work = [&](int64_t k) {
            double w = (double)(h.circ_ptr[item_first[(size_t)k] + 1] - h.circ_ptr[item_first[(size_t)k]]) + 24.0;
            if (item_partner[(size_t)k] >= 0) w += (double)(h.circ_ptr[item_partner[(size_t)k] + 1] - h.circ_ptr[item_partner[(size_t)k]]) + 24.0;
            return w;
        };
        double total = 0;
        for (int64_t k = 0; k < n_items; k++) total += work(k);
        double acc = 0;
        int r = 1;
        for (int64_t k = 0; k < n_items && r < 8; k++) {
            acc += work(k);
            while (r < 8 && acc >= total * r / 8.0) range_begin[r++] = (uint32_t)(k + 1);
        }
        for (; r < 8; r++) range_begin[r] = (uint32_t)n_items;
        range_begin[8] = (uint32_t)n_items;
    }
    if (h.D != 16) {            // (the other contraction kernels index the plain permutation; their ranges are unused)
        for (int r = 0; r <= 8; r++) range_begin[r] = (uint32_t)(h.n_circuits * r / 8);
    }
    HIP_TRY(p->d_range_begin.ensure(9));
    HIP_TRY(hipMemcpyAsync(p->d_range_begin.p, range_begin.data(), 9 * 4, hipMemcpyHostToDevice, p->stream));
    HIP_TRY(hipStreamSynchronize(p->stream));
    HIP_TRY(p->d_work_counter.ensure(8));
    HIP_TRY(hipStreamSynchronize(p->stream));
    if (h.D == 16 && pairing && p->ana_stream && nG <= 63) {
        // Two-circuit items as ONE stream of blocks of 4 "slots": per gate the common tail (slot = one application of
        // both circuits: their two forward ids and the shared backward id), then what each circuit has before the tail
        // (the other circuit's forward id = -1: its operand is zeroed), padded to a multiple of 4 with dead slots.  The
        // contraction's gather pipeline then runs through a whole item without draining at every gate and segment.
        const size_t blk_slots = 4 * (size_t)gst::analytic_stream_chunks();
        std::vector<int32_t> bf1, bf2, br, bptr((size_t)n_items * (size_t)nG + 1, 0);
        bf1.reserve(pf.size()); bf2.reserve(pf.size()); br.reserve(pf.size());
        for (int64_t k = 0; k < n_items; k++) {
            const int32_t c = item_first[(size_t)k], c2 = item_partner[(size_t)k];
            for (int g = 0; g < nG; g++) {
                bptr[(size_t)k * nG + g] = (int32_t)(bf1.size() / blk_slots);
                if (c2 < 0) continue;
                const int64_t p0 = pos_ptr[(size_t)c * nG + g], p1 = pos_ptr[(size_t)c * nG + g + 1];
                const int64_t q0 = pos_ptr[(size_t)c2 * nG + g], q1 = pos_ptr[(size_t)c2 * nG + g + 1];
                const int64_t cg = item_common[(size_t)k * nG + g];
                for (int64_t t = 0; t < cg; t++) { bf1.push_back(pf[(size_t)(p1 - cg + t)]); bf2.push_back(pf[(size_t)(q1 - cg + t)]); br.push_back(pr[(size_t)(p1 - cg + t)]); }
                for (int64_t j = p0; j < p1 - cg; j++) { bf1.push_back(pf[(size_t)j]); bf2.push_back(-1); br.push_back(pr[(size_t)j]); }
                for (int64_t j = q0; j < q1 - cg; j++) { bf1.push_back(-1); bf2.push_back(pf[(size_t)j]); br.push_back(pr[(size_t)j]); }
                while (bf1.size() % blk_slots) { bf1.push_back(-1); bf2.push_back(-1); br.push_back(br.empty() ? 0 : br.back()); }
            }
            if (bf1.size() / 4 > 0x1ffffff0u) return fail(GST_EUNSUPPORTED, "analytic block stream too long");
        }
        bptr[(size_t)n_items * nG] = (int32_t)(bf1.size() / blk_slots);
        if (bf1.empty()) { bf1.assign(blk_slots, -1); bf2.assign(blk_slots, -1); br.assign(blk_slots, 0); }
        if ((rc = upload_i32(p->d_blk_f1, bf1, p->stream))) return rc;
        if ((rc = upload_i32(p->d_blk_f2, bf2, p->stream))) return rc;
        if ((rc = upload_i32(p->d_blk_r, br, p->stream))) return rc;
        if ((rc = upload_i32(p->d_blk_ptr, bptr, p->stream))) return rc;
        HIP_TRY(hipStreamSynchronize(p->stream));
    }
    if (h.D == 16) build_levels_host(p, true);      // (needs the reversed plan's state graph, dropped below; whatever GST_OPT_FAST_CHAINS says NOW)
    // (the host copies of the reversed programs are not needed any more)
    p->rev.prog.clear(); p->rev.prog.shrink_to_fit();
    p->rev.node_parent.clear(); p->rev.node_parent.shrink_to_fit();
    p->rev.node_sym.clear(); p->rev.node_sym.shrink_to_fit();
    p->rev_ready = true;
    return GST_OK;
}

// Exact Jacobian columns (GST_DERIV_ANALYTIC) into device memory.
int run_dprobs_analytic(gst_plan* p, double* d_out, int64_t ld, const int64_t* param_idx, const int64_t* dest_idx,
                        int64_t n_param, double* d_probs_out)
{
    const gst::HostPlan& h = p->hp;
    if (p->comp_index >= 0 && !p->derivs_set)
        return fail(GST_EUNSUPPORTED, "a complement effect is declared: exact derivatives of TP POVMs need gst_set_derivs");
    if (h.D != 4 && h.D != 16 && h.D != 64) return fail(GST_EUNSUPPORTED, "the analytic mode supports D = 4, 16 and 64");
    if (h.D == 64 && !p->ana_mfma) return fail(GST_EUNSUPPORTED, "D = 64 analytic derivatives exist on the MFMA path only");
    double* d_base = d_probs_out ? d_probs_out : p->d_pbase.p;
    // (the backward chain pass needs only the model arrays already on their way: it forks here, onto the second stream --
    //  only where the two-cache contraction will run: a plain 1Q Jacobian is launch-bound and takes the single kernel)
    const bool will_fork = p->ana_mfma && (h.D != 4 || p->want_cache_path);
    if (will_fork) HIP_TRY(hipEventRecord(p->ev_fork, p->stream));
    // Forward states: the sequential walk (bit-identical probabilities), or -- this mode has no ordering contract -- the
    // log-depth level pass where the plan's germ-power paths make it pay (GST_OPT_FAST_CHAINS; probabilities <= 1e-10)
    int rc;
    bool lv_f = false;
    p->last_levels = false;
    if (h.D == 16 && will_fork && n_param > 0 && p->fast_chains) {
        if ((rc = ensure_levels(p, false))) return rc;
        lv_f = levels_wanted(p, p->lv_fwd);
    }
    rc = lv_f ? run_levels_forward(p, d_base)
              : run_probs(p, d_base, n_param > 0, will_fork && n_param > 0 ? 2 : 1, nullptr, will_fork && n_param > 0);   // probabilities + every forward state
    if (rc) return rc;
    if (n_param == 0) return GST_OK;
    const bool request_was_cached = p->request_cached(2, param_idx, dest_idx, n_param);
    if (!request_was_cached) {
        p->request_serial++;
        const int D = h.D, DD = D * D;
        std::vector<int32_t> cm_gate((size_t)std::max(h.n_gates, 1) * DD, -1), cm_rho((size_t)h.n_rhos * D, -1),
            cm_eff((size_t)h.n_effects * D, -1), col0(std::max(h.n_gates, 1), -2);
        std::vector<int64_t> none_cols;
        for (int64_t c = 0; c < n_param; c++) {
            const int64_t pi = param_idx[c];
            const int32_t col = (int32_t)(dest_idx ? dest_idx[c] : c);
            switch (p->pkind[pi]) {
            case GST_KIND_GATE: cm_gate[(size_t)p->pobj[pi] * DD + p->pelem[pi]] = col; break;
            case GST_KIND_RHO: cm_rho[(size_t)p->pobj[pi] * D + p->pelem[pi]] = col; break;
            case GST_KIND_EFFECT: cm_eff[(size_t)p->pobj[pi] * D + p->pelem[pi]] = col; break;
            default: none_cols.push_back(col);
            }
        }
        for (int g = 0; g < h.n_gates; g++) {
            const int32_t* m = cm_gate.data() + (size_t)g * DD;
            bool any = false, contiguous = m[0] >= 0;
            for (int k = 0; k < DD; k++) { any = any || m[k] >= 0; contiguous = contiguous && m[k] == m[0] + k; }
            col0[g] = contiguous ? m[0] : (any ? -1 : -2);
        }
        if (!p->graph_uploaded) {
            if ((rc = upload_i32(p->d_node_parent, h.node_parent, p->stream))) return rc;
            if ((rc = upload_i32(p->d_node_sym, h.node_sym, p->stream))) return rc;
            {   // run[id] = 1 + run[id-1] while parent(id) == id-1 is a gate state reached by a consecutive id
                std::vector<int32_t> run(h.n_state_ids, 0);
                for (int64_t i = 1; i < h.n_state_ids; i++)
                    if (h.node_parent[i] == i - 1) run[i] = 1 + ((h.node_parent[i - 1] >= 0 && h.node_parent[i - 1] == i - 2) ? run[i - 1] : 0);
                if ((rc = upload_i32(p->d_node_run, run, p->stream))) return rc;
                HIP_TRY(hipStreamSynchronize(p->stream));
            }
            if ((rc = upload_i32(p->d_circ_leaf, h.circ_leaf, p->stream))) return rc;
            p->graph_uploaded = true;
        }
        p->cached_kind = 0;
        if ((rc = upload_i32(p->d_gate_col0, col0, p->stream))) return rc;
        if ((rc = upload_i32(p->d_cm_gate, cm_gate, p->stream))) return rc;
        if ((rc = upload_i32(p->d_cm_rho, cm_rho, p->stream))) return rc;
        if ((rc = upload_i32(p->d_cm_eff, cm_eff, p->stream))) return rc;
        HIP_TRY(hipStreamSynchronize(p->stream));          // host vectors above go out of scope
        p->remember_request(2, param_idx, dest_idx, n_param);
        p->cached_none_cols = none_cols;
    }
    const int D = h.D;
    const std::vector<int64_t>& none_cols = p->cached_none_cols;
    for (int64_t col : none_cols)                      // parameters of objects this atom never applies: exact zeros
        HIP_TRY(hipMemset2DAsync(d_out + col, (size_t)ld * 8, 0, 8, (size_t)h.n_elements, p->stream));
    gst::AnaArgs a;
    std::memset(&a, 0, sizeof(a));
    a.n_circuits = h.n_circuits;
    a.circ_leaf = p->d_circ_leaf.p; a.node_parent = p->d_node_parent.p; a.node_sym = p->d_node_sym.p; a.node_run = p->d_node_run.p;
    a.eff_ptr = p->d_eff_ptr.p; a.eff_label = p->d_eff_label.p; a.eff_dest = p->d_eff_dest.p;
    a.gates_t = p->d_gates_t.p; a.effects = p->d_effects.p; a.base_cache = p->d_base_cache.p;
    a.n_gates = h.n_gates; a.n_rhos = h.n_rhos; a.n_effects = h.n_effects;
    a.gate_col0 = p->d_gate_col0.p; a.colmap_gate = p->d_cm_gate.p; a.colmap_rho = p->d_cm_rho.p; a.colmap_eff = p->d_cm_eff.p;
    a.out = d_out; a.ld = ld;
    // The MFMA kernels address both state caches with a uniform 64-bit base + 32-bit per-lane byte offsets; a cache of
    // 4 GB or more selects their WIDE instantiation (64-bit lane offsets: two more address registers per gather in
    // flight), nothing is refused.  (GST_TEST_FORCE cache_limit=: tests lower the 4 GB so that a small plan takes that form.)
    const double cache_limit = p->test_cache_limit > 0 ? p->test_cache_limit : 4.0e9;
    const bool caches_small = (double)h.n_state_ids * D * 8 < cache_limit;
    // the two-cache contraction: MFMA at D = 16 / 64; at D = 4 (VALU) only when a Hessian needs its tables -- a plain 1Q
    // Jacobian is launch-bound and the single backward-walking kernel below is one launch instead of three
    // D <= 16: the backward pass needs the chain kernel, whose tables (all gates, effects, emit ring) live in LDS; a gate
    // set too large for it takes the single-kernel path below (Jacobians) or is refused (Hessians need the caches)
    const bool chain_ok = D == 64 || gst::chain_kernel_fits(D, h.n_gates, h.n_effects, 4);
    if (!chain_ok && p->want_cache_path)
        return fail(GST_EUNSUPPORTED, "exact Hessians at D <= 16 need the gate set in LDS (at most " +
                                          std::to_string(128 * 1024 / (D * D * 8)) + " gates at this D)");
    bool rev_small = true;
    if (will_fork && chain_ok) {
        if ((rc = ensure_reverse(p))) return rc;
        rev_small = (double)p->rev.n_state_ids * h.n_effects * D * 8 < cache_limit;
    }
    if (will_fork && chain_ok) {
        a.wide = (caches_small && rev_small) ? 0 : 1;
        // backward states: the chain kernel over the reversed plan, transposed gates (= the row-major array), one lane
        // group per effect (64/D effects per pass)
        gst::WalkArgs w;
        std::memset(&w, 0, sizeof(w));
        w.prog = p->d_rprog.p; w.task_off = p->d_rtask_off.p;
        w.eff_ptr = p->d_reff_ptr.p; w.eff_label = p->d_reff_ptr.p; w.eff_dest = p->d_reff_ptr.p;
        w.gates = p->d_gates_t.p; w.gates_t = p->d_gates.p;          // (G^T)^T = G: the roles of the two layouts swap
        w.rhos = p->d_effects.p; w.effects = p->d_effects.p;
        w.n_gates = h.n_gates; w.n_effects = 0;
        w.n_pwaves = 1; w.rows_S = 0; w.mode = gst::EMIT_PROBS; w.out = p->d_pbase.p;
        HIP_TRY(p->d_rev_cache.ensure((size_t)p->rev.n_state_ids * h.n_effects * D));
        w.base_cache_w = p->d_rev_cache.p;
        w.multi_start = h.n_effects;
        w.chain_share = 2;                      // (the forward pass runs beside this one)
        TIME_REC(p, evk0);
        // Both chain passes are latency-bound (one wavefront per task, a fraction of the SIMDs): the backward one runs
        // on the second stream beside the forward pass launched above, and the contraction waits for both.
        HIP_TRY(hipStreamWaitEvent(p->stream2, p->ev_fork, 0));
        bool lv_r = false;
        if (D == 16 && p->fast_chains) {
            if ((rc = ensure_levels(p, true))) return rc;
            lv_r = levels_wanted(p, p->lv_rev);
        }
        if (lv_r) {                                            // backward states by the level pass: one launch, all effects
            gst::LevelArgs ra;
            level_args(p->lv_rev, ra);
            ra.bmats = p->d_gates.p; ra.starts = p->d_effects.p; ra.cache = p->d_rev_cache.p;
            HIP_TRY(gst::launch_level_pass(ra, p->rev.n_tasks(), p->stream2));
            p->last_launches++;
        } else if (D == 64 && p->fast_chains) {                // all effects of a task as one row block on the matrix cores
            for (int e0 = 0; e0 < h.n_effects; e0 += 16) {
                w.start0 = e0;
                HIP_TRY(gst::launch_chain64(w, p->rev.n_tasks(), p->rev.max_slots, p->stream2));
                p->last_launches++;
            }
        } else if (D == 64) {                                  // one wavefront per (task, effect), a single launch
            w.start0 = 0; w.n_pwaves = h.n_effects;
            HIP_TRY(gst::launch_walk_rows(D, w, p->rev.n_tasks(), p->rev.max_slots, p->stream2));
            p->last_launches++;
        } else {
            for (int e0 = 0; e0 < h.n_effects; e0 += 64 / D) {  // four effects (lane groups) per pass of the chain kernel
                w.start0 = e0;
                HIP_TRY(gst::launch_walk_rows(D, w, p->rev.n_tasks(), p->rev.max_slots, p->stream2));
                p->last_launches++;
            }
        }
        HIP_TRY(hipEventRecord(p->ev_join, p->stream2));
        HIP_TRY(hipStreamWaitEvent(p->stream, p->ev_join, 0));
        a.rev_cache = p->d_rev_cache.p; a.rev_leaf = p->d_rev_leaf.p; a.pair_f = p->d_pair_f.p; a.pair_r = p->d_pair_r.p;
        a.circ_partner = D == 16 ? p->d_circ_partner.p : nullptr; a.pair_common = D == 16 ? p->d_pair_common.p : nullptr;
        a.pos_ptr = p->d_pos_ptr.p; a.circ_rho = p->d_circ_rho.p; a.circ_order = p->d_circ_order.p; a.work_counter = p->d_work_counter.p; a.range_begin = p->d_range_begin.p;
        a.group_fetch = p->ana_group_fetch ? 1 : 0;
        if (D == 16 && p->d_blk_ptr.p) { a.blk_f1 = p->d_blk_f1.p; a.blk_f2 = p->d_blk_f2.p; a.blk_r = p->d_blk_r.p; a.blk_ptr = p->d_blk_ptr.p; }
        // GST_OPT_ANALYTIC_KEEP_ZEROS: the blocks of gates an item never applies are exact zeros; when THIS destination got
        // THIS request last time (and the caller promised, by setting the option, to write nothing but row scalings into it
        // in between) they are zero already and are not stored again -- a third of the D = 16 contraction's stores
        // Without the option (value 2, the default) the same holds for destinations the library can vouch for: memory from
        // gst_device_malloc and the plan's own staging buffer, whose every other writer reports to gst_track.cpp.
        const bool same_dest = p->ana_zero_out == (const void*)d_out && p->ana_zero_ld == ld && p->ana_zero_valid && request_was_cached;
        const bool zero_form = D == 16 && !p->derivs_set && !p->want_cache_path;
        const size_t extent = jac_extent(nE_total(p), ld, dest_idx, n_param);
        const uint64_t sig = (p->uid * 0x9E3779B97F4A7C15ull) ^ (p->request_serial * 0xC2B2AE3D27D4EB4Full) ^ (uint64_t)ld;
        bool claim = false;
        a.zeros_resident = 0; a.zeros_ok = nullptr;
        if (zero_form && p->ana_keep_zeros == 1) a.zeros_resident = same_dest ? 1 : 0;
        else if (zero_form && p->ana_keep_zeros == 2 && (d_out == p->d_out.p || gst::track_owned(d_out, extent))) {
            claim = true;
            if (const uint32_t* w = gst::track_claim_find(d_out, extent, sig)) { a.zeros_resident = 1; a.zeros_ok = w; }
        }
        if (!claim) gst::track_touch(d_out, extent);
        p->last_zeros_resident = a.zeros_resident != 0;
        p->ana_zero_out = d_out; p->ana_zero_ld = ld; p->ana_zero_valid = (D == 16);
        HIP_TRY(hipMemsetAsync(p->d_work_counter.p, 0, 8 * sizeof(uint32_t), p->stream));
        if (D == 64) HIP_TRY(gst::launch_analytic_mfma64(a, p->stream));
        else if (D == 16) HIP_TRY(gst::launch_analytic_mfma(a, p->stream));
        else HIP_TRY(gst::launch_analytic_small(a, p->stream));
        TIME_REC(p, evk1);
        p->last_launches++;
        if (claim) {         // what this fill leaves behind; the word reads 1 again whatever a row scaling did to it before
            if (uint32_t* w = gst::track_claim_set(d_out, extent, sig, p->device)) HIP_TRY(hipMemsetD32Async((hipDeviceptr_t)w, 1, 1, p->stream));
        }
        p->last_ana = a; p->last_ana_valid = true;     // (the Hessian rows re-launch the contraction with other caches)
        p->last_ana.zeros_resident = 0; p->last_ana.zeros_ok = nullptr;
        return GST_OK;
    }
    if (D == 64) return fail(GST_EUNSUPPORTED, "D = 64 analytic derivatives exist on the MFMA path only");
    gst::track_touch(d_out, jac_extent(h.n_elements, ld, dest_idx, n_param));
    p->last_zeros_resident = false;
    TIME_REC(p, evk0);
    HIP_TRY(gst::launch_analytic(D, a, p->stream));
    TIME_REC(p, evk1);
    p->last_launches++;
    return GST_OK;
}

int end_call(gst_plan* p, bool sync);

// D2H of the dense staging Jacobian p->d_out [nE][n_param] into the caller's (ld, dest_idx) window, then end_call.  A
// contiguous destination window -- the `dest_param_slice` of the reference's seam (mapforwardsim.py:379-383) -- is one
// strided 2-D copy; only a scattered dest_idx needs host staging.
int copy_out_dprobs(gst_plan* p, double* out, int64_t ld, const int64_t* dest_idx, int64_t n_param, double* probs_out)
{
    const int64_t nE = p->hp.n_elements;
    bool window = true;
    for (int64_t c = 1; dest_idx && c < n_param; c++) window = window && dest_idx[c] == dest_idx[0] + c;
    const int64_t d0 = (dest_idx && n_param > 0) ? dest_idx[0] : 0;
    std::vector<double> stage;
    if (n_param > 0) {
        if (window) {
            HIP_TRY(hipMemcpy2DAsync(out + d0, (size_t)ld * 8, p->d_out.p, (size_t)n_param * 8, (size_t)n_param * 8, (size_t)nE,
                                     hipMemcpyDeviceToHost, p->stream));
        } else {
            stage.resize((size_t)nE * n_param);
            HIP_TRY(hipMemcpyAsync(stage.data(), p->d_out.p, (size_t)nE * n_param * 8, hipMemcpyDeviceToHost, p->stream));
        }
    }
    if (probs_out) HIP_TRY(hipMemcpyAsync(probs_out, p->d_pbase.p, nE * 8, hipMemcpyDeviceToHost, p->stream));
    int rc;
    if ((rc = end_call(p, true))) return rc;
    if (n_param > 0 && !window)
        for (int64_t k = 0; k < nE; k++)
            for (int64_t c = 0; c < n_param; c++) out[k * ld + dest_idx[c]] = stage[(size_t)k * n_param + c];
    return GST_OK;
}

// Finite differences over ANY parameterisation (gst_fill_dprobs_models): column m = (p(model set m) - p(base)) / eps, the
// base being gst_set_model's model.  Every model set is a complete dense model (what set_parameter_value + to_dense give
// on the host), so nothing is assumed about which elements a parameter moves; the price is that no state is shared
// with the base pass -- each (task, model set) pair is a full probability walk (chain kernel at D <= 16, row-per-lane
// kernel at D = 64).  Model sets are processed in chunks that bound the scratch (probability vectors) to 2 GB.
// `nm` model sets resident in d_mm_models ([gates_t | rhos | effects] each): one independent probability walk per (walk
// program, set) into d_mm_raw, then columns m0 .. m0 + nm (or d_dest) of d_out = (p_set - p_base) / eps.
int run_models_chunk(gst_plan* p, int64_t nm, int64_t m0, const double* d_base, double* d_out, int64_t ld, const int32_t* d_dest, double eps)
{
    const gst::HostPlan& h = p->hp;
    const int D = h.D;
    const int64_t nE = h.n_elements, nT = h.n_tasks();
    const size_t ng = (size_t)h.n_gates * D * D, nr = (size_t)h.n_rhos * D, ne = (size_t)h.n_effects * D;
    gst::WalkArgs w;
    base_args(p, w);
    w.gates = nullptr;
    w.gates_t = p->d_mm_models.p; w.rhos = p->d_mm_models.p + ng; w.effects = p->d_mm_models.p + ng + nr;
    w.n_models = (int32_t)nm; w.model_stride = (int64_t)(ng + nr + ne); w.out_model_stride = nE; w.mm_tasks = (int32_t)nT;
    w.n_pwaves = (int32_t)nm;
    w.mode = gst::EMIT_PROBS; w.rows_S = 0; w.out = p->d_mm_raw.p;
    HIP_TRY(gst::launch_walk_rows(D, w, nT, h.max_slots, p->stream));
    HIP_TRY(gst::launch_fd_from_models(p->d_mm_raw.p, nE, d_base, nE, (int32_t)nm, d_dest, (int32_t)m0, eps, d_out, ld, p->stream));
    p->last_launches += 2;
    return GST_OK;
}

int run_dprobs_models(gst_plan* p, int64_t n_models, const double* gates, const double* rhos, const double* effects,
                      double* d_out, int64_t ld, const int64_t* dest_idx, double eps, double* d_probs_out)
{
    const gst::HostPlan& h = p->hp;
    const int D = h.D;
    const int64_t nE = h.n_elements;
    const size_t ng = (size_t)h.n_gates * D * D, nr = (size_t)h.n_rhos * D, ne = (size_t)h.n_effects * D;
    const size_t stride = ng + nr + ne;
    double* d_base = d_probs_out ? d_probs_out : p->d_pbase.p;
    int rc = run_probs(p, d_base, false);
    if (rc) return rc;
    if (n_models == 0) return GST_OK;
    if (n_models > 0x7fffffffLL) return fail(GST_EINVAL, "too many model sets");
    const int64_t nT = h.n_tasks();
    int64_t chunk = std::max<int64_t>(1, (int64_t)(2.0e9 / (8.0 * (double)std::max<int64_t>(nE, 1))));
    chunk = std::min<int64_t>(chunk, std::max<int64_t>(1, 0x7fffffffLL / std::max<int64_t>(nT, 1)));
    chunk = std::min<int64_t>(chunk, n_models);
    HIP_TRY(p->d_mm_models.ensure((size_t)chunk * stride));
    HIP_TRY(p->d_mm_raw.ensure((size_t)chunk * (size_t)std::max<int64_t>(nE, 1)));
    std::vector<int32_t> dest32;
    if (dest_idx) {
        dest32.resize((size_t)n_models);
        for (int64_t m = 0; m < n_models; m++) {
            if (dest_idx[m] < 0 || dest_idx[m] >= ld) return fail(GST_EINVAL, "destination column out of range");
            dest32[(size_t)m] = (int32_t)dest_idx[m];
        }
        if ((rc = upload_i32(p->d_mm_dest, dest32, p->stream))) return rc;
    } else if (n_models > ld) return fail(GST_EINVAL, "more model sets than columns");
    std::vector<double> stage((size_t)chunk * stride);
    TIME_REC(p, evk0);
    for (int64_t m0 = 0; m0 < n_models; m0 += chunk) {
        const int64_t nm = std::min<int64_t>(chunk, n_models - m0);
        for (int64_t m = 0; m < nm; m++) {
            double* dst = stage.data() + (size_t)m * stride;
            const double* G = gates + (size_t)(m0 + m) * ng;
            for (int g = 0; g < h.n_gates; g++)
                for (int i = 0; i < D; i++)
                    for (int j = 0; j < D; j++) dst[((size_t)g * D + j) * D + i] = G[((size_t)g * D + i) * D + j];
            std::memcpy(dst + ng, rhos + (size_t)(m0 + m) * nr, nr * 8);
            std::memcpy(dst + ng + nr, effects + (size_t)(m0 + m) * ne, ne * 8);
        }
        HIP_TRY(hipMemcpyAsync(p->d_mm_models.p, stage.data(), (size_t)nm * stride * 8, hipMemcpyHostToDevice, p->stream));
        if ((rc = run_models_chunk(p, nm, m0, d_base, d_out, ld, dest_idx ? p->d_mm_dest.p + m0 : nullptr, eps))) return rc;
        HIP_TRY(hipStreamSynchronize(p->stream));          // the staging vector is refilled by the next chunk
    }
    TIME_REC(p, evk1);
    return GST_OK;
}

// ---- Lindblad-parameterised members (gst_set_lindblad) -------------------------------------------------------------------
size_t lb_set_stride(const gst_plan* p)
{
    const int D = p->hp.D;
    return (size_t)p->hp.n_gates * D * D + (size_t)p->hp.n_rhos * D + (size_t)p->hp.n_effects * D;
}

int lb_upload(gst_plan* p)
{
    gst_plan::Lindblad& L = p->lb;
    if (L.uploaded) return GST_OK;
    std::vector<int32_t> i32;
    for (const auto* v : {&L.kind, &L.obj, &L.n_eff, &L.n_par, &L.n_blocks, &L.blk_type, &L.blk_mode, &L.blk_n}) i32.insert(i32.end(), v->begin(), v->end());
    std::vector<int64_t> i64;
    for (const auto* v : {&L.param0, &L.term_off, &L.static_off}) i64.insert(i64.end(), v->begin(), v->end());
    int rc;
    if ((rc = upload_i32(p->d_lb_i32, i32, p->stream))) return rc;
    HIP_TRY(p->d_lb_i64.ensure(i64.size()));
    HIP_TRY(hipMemcpyAsync(p->d_lb_i64.p, i64.data(), i64.size() * 8, hipMemcpyHostToDevice, p->stream));
    HIP_TRY(p->d_lb_statics.ensure(L.statics.size()));
    HIP_TRY(hipMemcpyAsync(p->d_lb_statics.p, L.statics.data(), L.statics.size() * 8, hipMemcpyHostToDevice, p->stream));
    HIP_TRY(p->d_lb_term_re.ensure(L.term_re.size()));
    HIP_TRY(hipMemcpyAsync(p->d_lb_term_re.p, L.term_re.data(), L.term_re.size() * 8, hipMemcpyHostToDevice, p->stream));
    HIP_TRY(p->d_lb_term_im.ensure(L.term_im.size()));
    HIP_TRY(hipMemcpyAsync(p->d_lb_term_im.p, L.term_im.data(), L.term_im.size() * 8, hipMemcpyHostToDevice, p->stream));
    HIP_TRY(p->d_lb_theta.ensure((size_t)std::max(L.n_params, 1)));
    HIP_TRY(p->d_lb_base.ensure(lb_set_stride(p)));
    HIP_TRY(p->d_lb_gates_rm.ensure(std::max<size_t>((size_t)p->hp.n_gates * p->hp.D * p->hp.D, 1)));
    HIP_TRY(hipStreamSynchronize(p->stream));
    L.uploaded = true;
    return GST_OK;
}

void lb_args(gst_plan* p, gst::LbArgs& a)
{
    const gst_plan::Lindblad& L = p->lb;
    std::memset(&a, 0, sizeof(a));
    const size_t nm = (size_t)L.n_members;
    a.n_members = L.n_members; a.n_gates = p->hp.n_gates; a.n_rhos = p->hp.n_rhos; a.n_effects = p->hp.n_effects;
    const int32_t* i = p->d_lb_i32.p;
    a.kind = i; a.obj = i + nm; a.n_eff = i + 2 * nm; a.n_params = i + 3 * nm; a.n_blocks = i + 4 * nm;
    a.blk_type = i + 5 * nm; a.blk_mode = a.blk_type + nm * gst::LB_MAX_BLOCKS; a.blk_n = a.blk_mode + nm * gst::LB_MAX_BLOCKS;
    const int64_t* l = p->d_lb_i64.p;
    a.param0 = l; a.term_off = l + nm; a.static_off = l + 2 * nm;
    a.theta = p->d_lb_theta.p; a.term_re = p->d_lb_term_re.p; a.term_im = p->d_lb_term_im.p; a.statics = p->d_lb_statics.p;
    a.base_set = p->d_lb_base.p;
    a.set_stride = (int64_t)lb_set_stride(p);
}

// FD Jacobian columns of a Lindblad-parameterised model WITH state sharing (gst_kernels_pert.hip): the device builds the
// one changed member of every column; the base pass fills the state cache; every (task, member) pair whose dirty program
// is not empty is walked for 64/D columns per wavefront; POVM columns come from the circuits' final base states; every
// entry no walk reaches is an exact zero, written by one streaming pass.
int run_dprobs_lindblad_shared(gst_plan* p, double* d_out, int64_t ld, const int64_t* param_idx, const int64_t* dest_idx,
                               int64_t n_param, double eps, double* d_probs_out)
{
    const gst::HostPlan& h = p->hp;
    const int D = h.D, G = 64 / D;
    const gst_plan::Lindblad& L = p->lb;
    double* d_base = d_probs_out ? d_probs_out : p->d_pbase.p;
    int rc = run_probs(p, d_base, n_param > 0);
    if (rc || n_param == 0) return rc;
    const int64_t nT = h.n_tasks();
    if (!p->dirty_ready) {
        gst::build_dirty_programs(h, p->dirty);
        HIP_TRY(p->d_dirty_words.ensure(p->dirty.words.size() + 64));
        HIP_TRY(hipMemsetAsync(p->d_dirty_words.p, 0, (p->dirty.words.size() + 64) * 4, p->stream));
        HIP_TRY(hipMemcpyAsync(p->d_dirty_words.p, p->dirty.words.data(), p->dirty.words.size() * 4, hipMemcpyHostToDevice, p->stream));
        HIP_TRY(p->d_dirty_off.ensure(p->dirty.off.size()));
        HIP_TRY(hipMemcpyAsync(p->d_dirty_off.p, p->dirty.off.data(), p->dirty.off.size() * 8, hipMemcpyHostToDevice, p->stream));
        HIP_TRY(hipStreamSynchronize(p->stream));
        p->dirty_ready = true;
    }
    if (!p->leaf_uploaded) {
        if ((rc = upload_i32(p->d_circ_leaf, h.circ_leaf, p->stream))) return rc;
        p->leaf_uploaded = true;
    }
    if (!p->request_cached(5, param_idx, dest_idx, n_param)) {
        p->cached_kind = 0;
        // columns grouped by member, G per wavefront
        std::vector<std::vector<int64_t>> by_member((size_t)L.n_members);
        std::vector<int64_t> no_member;
        for (int64_t c = 0; c < n_param; c++) {
            const int64_t gp = param_idx[c];
            int m = -1;
            for (int mm = 0; mm < L.n_members; mm++)
                if (gp >= L.param0[(size_t)mm] && gp < L.param0[(size_t)mm] + L.n_par[(size_t)mm]) m = mm;
            const int64_t dst = dest_idx ? dest_idx[c] : c;
            if (dst < 0 || dst >= ld) return fail(GST_EINVAL, "destination column out of range");
            // a parameter of no member of THIS plan (an object the atom never applies, cf. GST_KIND_NONE): the reference's
            // step changes no probability of the atom -- the column is an exact zero, written by the zero-fill pass
            if (m < 0) no_member.push_back(c);
            else by_member[(size_t)m].push_back(c);
        }
        std::vector<int64_t> set_param;
        std::vector<int32_t> wk, wo, wn, w0, wc, cdest, zero_dest;
        std::vector<int32_t> rl_col, rl_kind, rl_obj, rl_elem;       // preparation columns: lanes of the lane-per-model FD kernel
        p->lb_povm_cols.clear();
        for (int m = 0; m < L.n_members; m++) {
            const auto& cols = by_member[(size_t)m];
            if (cols.empty()) continue;
            if (L.kind[(size_t)m] == GST_KIND_RHO && p->lb_rho_lanes) {
                // A perturbed preparation changes every state of every circuit but no gate: these columns are exactly what
                // walk_kernel's lanes are for -- 64 columns per wavefront, every gate coefficient a scalar operand -- once a
                // lane may start from a whole perturbed vector (WalkArgs::rho_models) instead of one stepped element.
                for (size_t k = 0; k < cols.size(); k++) {
                    const int64_t c = cols[k];
                    rl_col.push_back((int32_t)(dest_idx ? dest_idx[c] : c)); rl_kind.push_back(GST_KIND_RHO);
                    rl_obj.push_back(L.obj[(size_t)m]); rl_elem.push_back((int32_t)set_param.size());
                    set_param.push_back(param_idx[c]);
                    cdest.push_back((int32_t)(dest_idx ? dest_idx[c] : c));
                }
                while (rl_col.size() % 64) { rl_col.push_back(-1); rl_kind.push_back(GST_KIND_NONE); rl_obj.push_back(0); rl_elem.push_back(0); }
                continue;
            }
            const bool povm = L.kind[(size_t)m] == GST_KIND_EFFECT;
            if (povm) {
                p->lb_povm_cols.push_back(L.obj[(size_t)m]); p->lb_povm_cols.push_back(L.n_eff[(size_t)m]);
                p->lb_povm_cols.push_back((int32_t)set_param.size()); p->lb_povm_cols.push_back((int32_t)cols.size());
            }
            for (size_t k = 0; k < cols.size(); k += (size_t)G) {
                const size_t n = std::min<size_t>((size_t)G, cols.size() - k);
                if (!povm) {
                    wk.push_back(L.kind[(size_t)m]); wo.push_back(L.obj[(size_t)m]); wn.push_back(L.n_eff[(size_t)m]);
                    w0.push_back((int32_t)set_param.size()); wc.push_back((int32_t)n);
                }
                for (size_t q = 0; q < n; q++) {
                    const int64_t c = cols[k + q];
                    set_param.push_back(param_idx[c]);
                    const int32_t dst = (int32_t)(dest_idx ? dest_idx[c] : c);
                    cdest.push_back(dst);
                    if (!povm) zero_dest.push_back(dst);
                }
            }
        }
        p->lb_n_sets = (int64_t)set_param.size();          // (member-less columns come last: no perturbed set is built for them)
        for (int64_t c : no_member) {
            const int32_t dst = (int32_t)(dest_idx ? dest_idx[c] : c);
            cdest.push_back(dst); zero_dest.push_back(dst);
        }
        std::sort(zero_dest.begin(), zero_dest.end());
        const int32_t n_pw = (int32_t)wk.size();
        // wave tables: kind | obj | n_eff | col0 | ncols (n_pw each), then col_dest (n_param), then the zero-fill list
        std::vector<int32_t> tab;
        for (const auto* v : {&wk, &wo, &wn, &w0, &wc, &cdest, &zero_dest}) tab.insert(tab.end(), v->begin(), v->end());
        if ((rc = upload_i32(p->d_lb_waves, tab, p->stream))) return rc;
        p->lb_n_zero = (int32_t)zero_dest.size();
        HIP_TRY(p->d_lb_setparam.ensure(set_param.size()));
        HIP_TRY(hipMemcpyAsync(p->d_lb_setparam.p, set_param.data(), set_param.size() * 8, hipMemcpyHostToDevice, p->stream));
        HIP_TRY(p->d_lb_pert.ensure((size_t)n_param * D * D));
        // work items: (dirty program of (task, the wavefront's class), wavefront), expensive first; empty programs -- the task
        // never shows the member to an outcome -- are no items at all
        const int nC = p->dirty.n_classes;
        std::vector<std::pair<int64_t, std::pair<uint32_t, int32_t>>> items;
        items.reserve((size_t)nT * std::max(n_pw, 1));
        for (int64_t t = 0; t < nT; t++)
            for (int32_t w = 0; w < n_pw; w++) {
                const int cls = wk[(size_t)w] == GST_KIND_GATE ? wo[(size_t)w] : h.n_gates + wo[(size_t)w];
                const size_t pi = (size_t)t * nC + cls;
                if (p->dirty.off[pi + 1] == p->dirty.off[pi]) continue;
                items.push_back({-((int64_t)p->dirty.applies[pi] + p->dirty.emits[pi] / 2 + 8), {(uint32_t)pi, w}});
            }
        std::stable_sort(items.begin(), items.end(), [](const auto& x, const auto& y) { return x.first < y.first; });
        std::vector<uint32_t> iprog(items.size());
        std::vector<int32_t> ipw(items.size());
        for (size_t i = 0; i < items.size(); i++) { iprog[i] = items[i].second.first; ipw[i] = items[i].second.second; }
        HIP_TRY(p->d_block_order.ensure(iprog.size() + 1));
        if (!iprog.empty()) HIP_TRY(hipMemcpyAsync(p->d_block_order.p, iprog.data(), iprog.size() * 4, hipMemcpyHostToDevice, p->stream));
        if ((rc = upload_i32(p->d_lb_item_pw, ipw, p->stream))) return rc;
        p->lbr_n_waves = (int32_t)(rl_col.size() / 64);
        if (p->lbr_n_waves > 0) {
            if ((rc = upload_i32(p->d_lbr_lane[0], rl_col, p->stream)) || (rc = upload_i32(p->d_lbr_lane[1], rl_kind, p->stream)) ||
                (rc = upload_i32(p->d_lbr_lane[2], rl_obj, p->stream)) || (rc = upload_i32(p->d_lbr_lane[3], rl_elem, p->stream))) return rc;
            // (task, wavefront) pairs, longest programs first: every pair walks its whole task
            std::vector<int64_t> order((size_t)nT);
            for (int64_t t = 0; t < nT; t++) order[(size_t)t] = t;
            std::stable_sort(order.begin(), order.end(), [&](int64_t x, int64_t y) { return h.task_off[(size_t)x + 1] - h.task_off[(size_t)x] > h.task_off[(size_t)y + 1] - h.task_off[(size_t)y]; });
            std::vector<uint32_t> bo;
            bo.reserve((size_t)nT * p->lbr_n_waves);
            for (int64_t t : order) for (int32_t w = 0; w < p->lbr_n_waves; w++) bo.push_back((uint32_t)(t * p->lbr_n_waves + w));
            HIP_TRY(p->d_lbr_order.ensure(bo.size() + 1));
            HIP_TRY(hipMemcpyAsync(p->d_lbr_order.p, bo.data(), bo.size() * 4, hipMemcpyHostToDevice, p->stream));
        }
        HIP_TRY(hipStreamSynchronize(p->stream));          // the host vectors go out of scope
        p->lb_n_pwaves = n_pw;
        p->lb_n_items = (int64_t)items.size();
        p->remember_request(5, param_idx, dest_idx, n_param);
    }
    // the changed member of every column
    gst::LbArgs b;
    lb_args(p, b);
    b.set_param = p->d_lb_setparam.p; b.sets = p->d_lb_pert.p; b.set_stride = (int64_t)D * D; b.member_only = 1; b.eps = eps;
    HIP_TRY(gst::launch_lindblad_build(D, b, p->lb_n_sets, p->stream));
    gst::PertArgs a;
    std::memset(&a, 0, sizeof(a));
    a.prog = p->d_dirty_words.p; a.prog_off = p->d_dirty_off.p;
    a.item_prog = p->d_block_order.p; a.item_pw = p->d_lb_item_pw.p;
    a.eff_ptr = p->d_eff_ptr.p; a.eff_label = p->d_eff_label.p; a.eff_dest = p->d_eff_dest.p;
    a.gates_t = p->d_gates_t.p; a.rhos = p->d_rhos.p; a.effects = p->d_effects.p;
    a.n_gates = h.n_gates; a.n_effects = h.n_effects;
    a.base_cache = p->d_base_cache.p; a.pbase = d_base;
    a.pert = p->d_lb_pert.p; a.pert_stride = (int64_t)D * D;
    const int32_t npw = p->lb_n_pwaves;
    const int32_t* t = p->d_lb_waves.p;
    a.wave_kind = t; a.wave_obj = t + npw; a.wave_neff = t + 2 * npw; a.wave_col0 = t + 3 * npw; a.wave_ncols = t + 4 * npw;
    a.col_dest = t + 5 * npw;
    a.n_pwaves = npw;
    a.out = d_out; a.ld = ld; a.eps = eps;
    TIME_REC(p, evk0);
    HIP_TRY(gst::launch_zero_columns(d_out, ld, h.n_elements, a.col_dest + n_param, p->lb_n_zero, -1, p->stream));
    HIP_TRY(gst::launch_walk_pert(D, a, p->lb_n_items, h.max_slots, p->stream));
    if (p->lbr_n_waves > 0) {          // preparation columns: the lane-per-model FD kernel, each lane starting from its own vector
        gst::WalkArgs w;
        base_args(p, w);
        w.mode = gst::EMIT_FD;
        w.out = d_out; w.ld = ld; w.eps = eps; w.pbase = d_base; w.base_cache = p->d_base_cache.p;
        w.lanes.col = p->d_lbr_lane[0].p; w.lanes.kind[0] = p->d_lbr_lane[1].p; w.lanes.obj[0] = p->d_lbr_lane[2].p; w.lanes.elem[0] = p->d_lbr_lane[3].p;
        w.n_pwaves = p->lbr_n_waves;
        w.block_order = p->d_lbr_order.p;
        w.rho_models = p->d_lb_pert.p; w.rho_model_stride = (int64_t)D * D;
        HIP_TRY(gst::launch_walk(D, 1, w, nT, h.max_slots, p->stream, 1));
        p->last_launches++;
    }
    for (size_t k = 0; k + 3 < p->lb_povm_cols.size(); k += 4)
        HIP_TRY(gst::launch_effect_columns(D, a, p->d_circ_leaf.p, h.n_circuits, p->lb_povm_cols[k], p->lb_povm_cols[k + 1],
                                           p->lb_povm_cols[k + 2], p->lb_povm_cols[k + 3], p->stream));
    TIME_REC(p, evk1);
    p->last_launches += 3 + (int64_t)p->lb_povm_cols.size() / 4;
    return GST_OK;
}

// FD Jacobian columns of a Lindblad-parameterised model: the device builds the dense model after every parameter step
// (one workgroup per column) and walks every (program, model set) pair; base probabilities from the base model, which
// gst_set_lindblad_params built.
int run_dprobs_lindblad(gst_plan* p, double* d_out, int64_t ld, const int64_t* param_idx, const int64_t* dest_idx, int64_t n_param,
                        double eps, double* d_probs_out)
{
    const gst::HostPlan& h = p->hp;
    const int64_t nE = h.n_elements, nT = h.n_tasks();
    if (!p->lb.have_theta) return fail(GST_ESTATE, "gst_set_lindblad_params has not been called");
    if (!(eps != 0.0)) return fail(GST_EINVAL, "eps must be non-zero");
    for (int64_t c = 0; c < n_param; c++)
        if (param_idx[c] < 0 || param_idx[c] >= p->lb.n_params) return fail(GST_EINVAL, "parameter index out of range");
    if (p->lb_share && gst::pert_kernel_fits(h.D, h.n_gates, h.n_effects, h.max_slots) && h.n_gates <= 64)
        return run_dprobs_lindblad_shared(p, d_out, ld, param_idx, dest_idx, n_param, eps, d_probs_out);
    p->cached_kind = 0;
    double* d_base = d_probs_out ? d_probs_out : p->d_pbase.p;
    int rc = run_probs(p, d_base, false);
    if (rc || n_param == 0) return rc;
    const size_t stride = lb_set_stride(p);
    int64_t chunk = std::max<int64_t>(1, (int64_t)(2.0e9 / (8.0 * (double)std::max<int64_t>(nE, 1))));
    chunk = std::min<int64_t>(chunk, std::max<int64_t>(1, 0x7fffffffLL / std::max<int64_t>(nT, 1)));
    chunk = std::min<int64_t>(chunk, n_param);
    HIP_TRY(p->d_mm_models.ensure((size_t)chunk * stride));
    HIP_TRY(p->d_mm_raw.ensure((size_t)chunk * (size_t)std::max<int64_t>(nE, 1)));
    HIP_TRY(p->d_lb_setparam.ensure((size_t)n_param));
    HIP_TRY(hipMemcpyAsync(p->d_lb_setparam.p, param_idx, (size_t)n_param * 8, hipMemcpyHostToDevice, p->stream));
    std::vector<int32_t> dest32;
    if (dest_idx) {
        dest32.resize((size_t)n_param);
        for (int64_t m = 0; m < n_param; m++) {
            if (dest_idx[m] < 0 || dest_idx[m] >= ld) return fail(GST_EINVAL, "destination column out of range");
            dest32[(size_t)m] = (int32_t)dest_idx[m];
        }
        if ((rc = upload_i32(p->d_mm_dest, dest32, p->stream))) return rc;
    } else if (n_param > ld) return fail(GST_EINVAL, "more columns than the leading dimension");
    HIP_TRY(hipStreamSynchronize(p->stream));              // (param_idx / dest32 are the caller's / a local)
    TIME_REC(p, evk0);
    for (int64_t m0 = 0; m0 < n_param; m0 += chunk) {
        const int64_t nm = std::min<int64_t>(chunk, n_param - m0);
        gst::LbArgs a;
        lb_args(p, a);
        a.set_param = p->d_lb_setparam.p + m0;
        a.sets = p->d_mm_models.p;
        a.eps = eps;
        HIP_TRY(gst::launch_lindblad_build(h.D, a, nm, p->stream));
        p->last_launches++;
        if ((rc = run_models_chunk(p, nm, m0, d_base, d_out, ld, dest_idx ? p->d_mm_dest.p + m0 : nullptr, eps))) return rc;
    }
    TIME_REC(p, evk1);
    return GST_OK;
}

int begin_call(gst_plan* p)
{
    if (!p) return fail(GST_EINVAL, "plan is NULL");
    int rc = ensure_device(p);
    if (rc) return rc;
    if (!p->have_model) return fail(GST_ESTATE, "gst_set_model has not been called");
    p->last_launches = 0;
    p->last_kernel_ms = 0;
    TIME_REC(p, ev0);
    TIME_REC(p, evk0);
    TIME_REC(p, evk1);
    return upload_model(p);
}

int end_call(gst_plan* p, bool sync)
{
    TIME_REC(p, ev1);
    if (sync) {
        HIP_TRY(hipStreamSynchronize(p->stream));
        float ms = 0;
        if (p->timing && hipEventElapsedTime(&ms, p->ev0, p->ev1) == hipSuccess) p->last_total_ms = ms;
        if (p->timing && hipEventElapsedTime(&ms, p->evk0, p->evk1) == hipSuccess) p->last_kernel_ms = ms;
    }
    return GST_OK;
}

}  // namespace

namespace {
// Host regions the caller page-locked through gst_host_register (mapped into the device's address space): fills whose
// destination lies inside one write their results straight into it.
std::mutex g_reg_mutex;
std::vector<std::pair<char*, size_t>> g_registered;

// Device address of a host pointer inside a registered region covering [ptr, ptr + bytes), or nullptr.
void* mapped_device_pointer(const void* ptr, size_t bytes)
{
    std::lock_guard<std::mutex> lock(g_reg_mutex);
    for (const auto& r : g_registered) {
        if ((const char*)ptr >= r.first && (const char*)ptr + bytes <= r.first + r.second) {
            void* d = nullptr;
            if (hipHostGetDevicePointer(&d, const_cast<void*>(ptr), 0) == hipSuccess) return d;
            (void)hipGetLastError();
            return nullptr;
        }
    }
    return nullptr;
}
}  // namespace

namespace gst {
int set_error(int code, const std::string& msg) { return fail(code, msg); }
int plan_ensure_device(gst_plan* plan) { return plan ? ensure_device(plan) : fail(GST_EINVAL, "plan is NULL"); }
hipStream_t plan_stream(const gst_plan* plan) { return plan->stream; }
int plan_device(const gst_plan* plan) { return plan->device; }
}  // namespace gst

extern "C" {

const char* gst_last_error(void) { return g_err.c_str(); }
const char* gst_version(void) { return "gstfwd 0.3 (gfx950)"; }

int gst_device_count(int32_t* n)
{
    return guarded([&]() -> int {
    if (!n) return fail(GST_EINVAL, "n is NULL");
    int c = 0;
    hipError_t e = hipGetDeviceCount(&c);
    *n = (e == hipSuccess) ? c : 0;
    return GST_OK;
    });
}

int gst_plan_create_from_table(const gst_table_desc* d, const gst_options* opt, gst_plan** out)
{
    return guarded([&]() -> int {
    if (!d || !out) return fail(GST_EINVAL, "NULL argument");
    *out = nullptr;
    gst_plan* p = new (std::nothrow) gst_plan();
    if (!p) return fail(GST_ENOMEM, "out of host memory");
    try {
        gst::HostPlan& h = p->hp;
        h.D = d->D; h.n_gates = d->n_gates; h.n_rhos = d->n_rhos; h.n_effects = d->n_effects;
        h.n_elements = d->n_elements;
        if (d->n_elements < 0 || d->n_elements > 0x7fffffffLL) { delete p; return fail(GST_EINVAL, "n_elements out of range"); }
        if (d->n_rows < 0 || !d->t_dest || !d->t_start || !d->t_cache || !d->t_rho || !d->row_ptr || !d->eff_ptr) {
            delete p; return fail(GST_EINVAL, "table arrays missing");
        }
        std::string err = gst::expand_table(h, d->n_rows, d->cache_size, d->t_dest, d->t_start, d->t_cache,
                                            d->t_rho, d->row_ptr, d->gate_idx);
        if (!err.empty()) { delete p; return fail(GST_EINVAL, err); }
        if (d->eff_ptr[d->n_rows] > 0x7fffffffLL || d->eff_ptr[d->n_rows] < 0) { delete p; return fail(GST_EINVAL, "bad eff_ptr"); }
        if (d->eff_ptr[d->n_rows] > 0 && (!d->eff_label || !d->eff_dest)) { delete p; return fail(GST_EINVAL, "effect arrays missing"); }
        h.eff_ptr.resize(d->n_rows + 1);
        for (int32_t i = 0; i <= d->n_rows; i++) h.eff_ptr[i] = (int32_t)d->eff_ptr[i];
        h.eff_label.assign(d->eff_label, d->eff_label + d->eff_ptr[d->n_rows]);
        h.eff_dest.assign(d->eff_dest, d->eff_dest + d->eff_ptr[d->n_rows]);
        return finish_create(p, opt, out);
    } catch (const std::bad_alloc&) {
        delete p; return fail(GST_ENOMEM, "out of host memory while compiling the plan");
    }
    });
}

int gst_plan_create_from_circuits(const gst_circuits_desc* d, const gst_options* opt, gst_plan** out)
{
    return guarded([&]() -> int {
    if (!d || !out) return fail(GST_EINVAL, "NULL argument");
    *out = nullptr;
    gst_plan* p = new (std::nothrow) gst_plan();
    if (!p) return fail(GST_ENOMEM, "out of host memory");
    try {
        gst::HostPlan& h = p->hp;
        h.D = d->D; h.n_gates = d->n_gates; h.n_rhos = d->n_rhos; h.n_effects = d->n_effects;
        h.n_elements = d->n_elements; h.n_circuits = d->n_circuits;
        if (d->n_elements < 0 || d->n_elements > 0x7fffffffLL) { delete p; return fail(GST_EINVAL, "n_elements out of range"); }
        if (d->n_circuits < 0 || !d->circ_rho || !d->circ_ptr || !d->eff_ptr) { delete p; return fail(GST_EINVAL, "circuit arrays missing"); }
        const int64_t nC = d->n_circuits;
        h.circ_rho.assign(d->circ_rho, d->circ_rho + nC);
        h.circ_ptr.assign(d->circ_ptr, d->circ_ptr + nC + 1);
        if (h.circ_ptr[nC] < 0) { delete p; return fail(GST_EINVAL, "bad circ_ptr"); }
        if (h.circ_ptr[nC] > 0) h.circ_gates.assign(d->circ_gates, d->circ_gates + h.circ_ptr[nC]);
        if (d->eff_ptr[nC] > 0x7fffffffLL || d->eff_ptr[nC] < 0) { delete p; return fail(GST_EINVAL, "bad eff_ptr"); }
        if (d->eff_ptr[nC] > 0 && (!d->eff_label || !d->eff_dest)) { delete p; return fail(GST_EINVAL, "effect arrays missing"); }
        h.eff_ptr.resize(nC + 1);
        for (int64_t i = 0; i <= nC; i++) h.eff_ptr[i] = (int32_t)d->eff_ptr[i];
        h.eff_label.assign(d->eff_label, d->eff_label + d->eff_ptr[nC]);
        h.eff_dest.assign(d->eff_dest, d->eff_dest + d->eff_ptr[nC]);
        return finish_create(p, opt, out);
    } catch (const std::bad_alloc&) {
        delete p; return fail(GST_ENOMEM, "out of host memory while compiling the plan");
    }
    });
}

int gst_plan_destroy(gst_plan* plan)
{
    return guarded([&]() -> int {
    delete plan;
    return GST_OK;
    });
}

int gst_set_model(gst_plan* p, const double* gates, const double* rhos, const double* effects)
{
    return guarded([&]() -> int {
    if (!p || !rhos || !effects || (p->hp.n_gates > 0 && !gates)) return fail(GST_EINVAL, "NULL argument");
    const int D = p->hp.D;
    const size_t ng = (size_t)p->hp.n_gates * D * D;
    p->h_gates.assign(gates, gates + ng);
    p->h_gates_t.resize(ng);
    for (int g = 0; g < p->hp.n_gates; g++)
        for (int i = 0; i < D; i++)
            for (int j = 0; j < D; j++)
                p->h_gates_t[((size_t)g * D + j) * D + i] = gates[((size_t)g * D + i) * D + j];
    p->h_rhos.assign(rhos, rhos + (size_t)p->hp.n_rhos * D);
    p->h_effects.assign(effects, effects + (size_t)p->hp.n_effects * D);
    p->have_model = true;
    p->model_dirty = true;
    return GST_OK;
    });
}

int gst_set_param_map(gst_plan* p, int32_t n_params, const int32_t* kind, const int32_t* obj, const int32_t* elem)
{
    return guarded([&]() -> int {
    if (!p || n_params < 0 || (n_params > 0 && (!kind || !obj || !elem))) return fail(GST_EINVAL, "bad argument");
    const int D = p->hp.D;
    for (int32_t i = 0; i < n_params; i++) {
        const int k = kind[i];
        if (k == GST_KIND_NONE) continue;
        const int nobj = k == GST_KIND_GATE ? p->hp.n_gates : k == GST_KIND_RHO ? p->hp.n_rhos : k == GST_KIND_EFFECT ? p->hp.n_effects : -1;
        const int nel = k == GST_KIND_GATE ? D * D : D;
        if (nobj < 0 || obj[i] < 0 || obj[i] >= nobj || elem[i] < 0 || elem[i] >= nel)
            return fail(GST_EINVAL, "parameter map entry " + std::to_string(i) + " out of range");
    }
    p->pkind.assign(kind, kind + n_params);
    p->pobj.assign(obj, obj + n_params);
    p->pelem.assign(elem, elem + n_params);
    p->have_pmap = true;
    p->cached_kind = 0;      // device lane tables / column maps describe the old map
    return GST_OK;
    });
}

int gst_set_complement_effect(gst_plan* p, int32_t comp_index, const double* identity, int32_t n_others, const int32_t* others)
{
    return guarded([&]() -> int {
    if (!p) return fail(GST_EINVAL, "plan is NULL");
    p->cached_kind = 0;
    if (comp_index < 0) { p->comp_index = -1; p->comp_others.clear(); p->comp_identity.clear(); return GST_OK; }
    if (comp_index >= p->hp.n_effects || !identity || n_others < 0 || (n_others > 0 && !others))
        return fail(GST_EINVAL, "bad complement description");
    for (int32_t k = 0; k < n_others; k++) {
        if (others[k] < 0 || others[k] >= p->hp.n_effects || others[k] == comp_index)
            return fail(GST_EINVAL, "complement: other effect " + std::to_string(k) + " out of range");
        for (int32_t m = 0; m < k; m++)
            if (others[m] == others[k]) return fail(GST_EINVAL, "complement: an effect is listed twice");
    }
    p->comp_index = comp_index;
    p->comp_others.assign(others, others + n_others);
    p->comp_identity.assign(identity, identity + p->hp.D);
    return GST_OK;
    });
}

int gst_set_derivs(gst_plan* p, int32_t n_params, int32_t n_objs, const int32_t* kind, const int32_t* obj,
                   const int32_t* n_cols, const int64_t* param_idx, const double* deriv)
{
    return guarded([&]() -> int {
    if (!p || n_params < 0 || n_objs < 0) return fail(GST_EINVAL, "bad argument");
    p->cached_kind = 0;
    if (n_objs == 0) { p->derivs_set = false; p->dv2_set = false; p->dv2_off.clear(); return GST_OK; }
    if (!kind || !obj || !n_cols || !param_idx || !deriv) return fail(GST_EINVAL, "bad argument");
    const int D = p->hp.D;
    std::vector<int64_t> off_c((size_t)n_objs + 1, 0), off_d((size_t)n_objs + 1, 0);
    for (int32_t o = 0; o < n_objs; o++) {
        const int k = kind[o];
        const int nobj = k == GST_KIND_GATE ? p->hp.n_gates : k == GST_KIND_RHO ? p->hp.n_rhos : k == GST_KIND_EFFECT ? p->hp.n_effects : -1;
        if (nobj < 0 || obj[o] < 0 || obj[o] >= nobj || n_cols[o] < 0) return fail(GST_EINVAL, "derivative object " + std::to_string(o) + " out of range");
        off_c[o + 1] = off_c[o] + n_cols[o];
        off_d[o + 1] = off_d[o] + (int64_t)(k == GST_KIND_GATE ? D * D : D) * n_cols[o];
    }
    for (int64_t c = 0; c < off_c[n_objs]; c++)
        if (param_idx[c] < 0 || param_idx[c] >= n_params) return fail(GST_EINVAL, "derivative parameter index out of range");
    int rc = ensure_device(p);
    if (rc) return rc;
    p->dv_kind.assign(kind, kind + n_objs); p->dv_obj.assign(obj, obj + n_objs); p->dv_ncols.assign(n_cols, n_cols + n_objs);
    p->dv_param_idx.assign(param_idx, param_idx + off_c[n_objs]);
    p->dv_off_cols = off_c; p->dv_off_deriv = off_d;
    p->dv_n_params = n_params;
    p->dv_deriv_h.assign(deriv, deriv + off_d[n_objs]);
    p->dv2_set = false; p->dv2_off.clear();
    HIP_TRY(p->d_dv_deriv.ensure((size_t)std::max<int64_t>(off_d[n_objs], 1)));
    if (off_d[n_objs] > 0) {
        HIP_TRY(hipMemcpyAsync(p->d_dv_deriv.p, deriv, (size_t)off_d[n_objs] * 8, hipMemcpyHostToDevice, p->stream));
        HIP_TRY(hipStreamSynchronize(p->stream));
    }
    p->derivs_set = true;
    return GST_OK;
    });
}

int gst_set_second_derivs(gst_plan* p, int32_t n_objs, const int32_t* nonzero, const double* hess)
{
    return guarded([&]() -> int {
    if (!p || n_objs < 0) return fail(GST_EINVAL, "bad argument");
    if (n_objs == 0) { p->dv2_set = false; p->dv2_off.clear(); return GST_OK; }
    if (!p->derivs_set || (size_t)n_objs != p->dv_kind.size()) return fail(GST_ESTATE, "gst_set_second_derivs follows gst_set_derivs, object for object");
    if (!nonzero) return fail(GST_EINVAL, "bad argument");
    const int D = p->hp.D;
    std::vector<int64_t> off((size_t)n_objs, -1);
    int64_t total = 0;
    for (int32_t o = 0; o < n_objs; o++) {
        if (!nonzero[o]) continue;
        const int64_t K = p->dv_kind[o] == GST_KIND_GATE ? D * D : D;
        off[(size_t)o] = total;
        total += K * p->dv_ncols[o] * (int64_t)p->dv_ncols[o];
    }
    if (total > 0 && !hess) return fail(GST_EINVAL, "hess is NULL");
    int rc = ensure_device(p);
    if (rc) return rc;
    HIP_TRY(p->d_dv2.ensure((size_t)std::max<int64_t>(total, 1)));
    if (total > 0) {
        HIP_TRY(hipMemcpyAsync(p->d_dv2.p, hess, (size_t)total * 8, hipMemcpyHostToDevice, p->stream));
        HIP_TRY(hipStreamSynchronize(p->stream));
    }
    p->dv2_off = off;
    p->dv2_set = total > 0;
    return GST_OK;
    });
}

// The element Jacobian [nE][n_el] ([rhos | effects | gates] of the `full` layout) of the current model into d_jelem,
// through the ordinary analytic path with the identity element map.
int run_element_jacobian(gst_plan* p, double* d_probs_out)
{
    const gst::HostPlan& h = p->hp;
    const int D = h.D;
    const int64_t nE = h.n_elements;
    const int64_t n_el = (int64_t)h.n_rhos * D + (int64_t)h.n_effects * D + (int64_t)h.n_gates * D * D;
    std::vector<int32_t> ek((size_t)n_el), eo((size_t)n_el), ee((size_t)n_el);
    {
        int64_t q = 0;
        for (int r = 0; r < h.n_rhos; r++) for (int j = 0; j < D; j++, q++) { ek[q] = GST_KIND_RHO; eo[q] = r; ee[q] = j; }
        for (int e = 0; e < h.n_effects; e++) for (int j = 0; j < D; j++, q++) { ek[q] = GST_KIND_EFFECT; eo[q] = e; ee[q] = j; }
        for (int g = 0; g < h.n_gates; g++) for (int j = 0; j < D * D; j++, q++) { ek[q] = GST_KIND_GATE; eo[q] = g; ee[q] = j; }
    }
    std::vector<int64_t> all((size_t)n_el);
    for (int64_t q = 0; q < n_el; q++) all[(size_t)q] = q;
    HIP_TRY(p->d_jelem.ensure((size_t)std::max<int64_t>(nE * n_el, 1)));
    p->pkind.swap(ek); p->pobj.swap(eo); p->pelem.swap(ee);
    p->cached_kind = 0;
    int rc = run_dprobs_analytic(p, p->d_jelem.p, n_el, all.data(), nullptr, n_el, d_probs_out);
    p->pkind.swap(ek); p->pobj.swap(eo); p->pelem.swap(ee);
    p->cached_kind = 0;
    return rc;
}

// GST_DERIV_ANALYTIC with gst_set_derivs: element Jacobian (the `full` layout [rhos | effects | gates]) into scratch,
// then one MFMA chain-rule product per object, accumulated into the requested parameter columns.
int run_dprobs_general(gst_plan* p, double* d_out, int64_t ld, const int64_t* param_idx, const int64_t* dest_idx,
                       int64_t n_param, double* d_probs_out)
{
    const gst::HostPlan& h = p->hp;
    const int D = h.D;
    const int64_t nE = h.n_elements;
    const int64_t n_el = (int64_t)h.n_rhos * D + (int64_t)h.n_effects * D + (int64_t)h.n_gates * D * D;
    for (int64_t c = 0; c < n_param; c++)
        if (param_idx[c] < 0 || param_idx[c] >= p->dv_n_params) return fail(GST_EINVAL, "parameter index out of range");
    std::vector<int32_t> dest_of((size_t)p->dv_n_params, -1);
    for (int64_t c = 0; c < n_param; c++) {
        if (dest_of[(size_t)param_idx[c]] >= 0) return fail(GST_EINVAL, "a parameter is requested twice (not supported with gst_set_derivs)");
        dest_of[(size_t)param_idx[c]] = (int32_t)(dest_idx ? dest_idx[c] : c);
    }
    int rc = run_element_jacobian(p, d_probs_out);
    if (rc) return rc;
    if (n_param == 0) return GST_OK;
    // Object by object: the first object that maps onto a destination column STORES its product there, later ones (shared
    // parameters: the effects of a POVM) add to it.  Only when some requested column is reached by no object at all are
    // the columns zeroed first (8.4 GB of memset and as much read-modify-write traffic for the 2Q CPTPLND Jacobian otherwise).
    std::vector<uint8_t> first_writer(p->dv_kind.size(), 0);
    bool all_covered = true;
    {
        std::vector<uint8_t> seen((size_t)p->dv_n_params, 0);
        for (size_t o = 0; o < p->dv_kind.size(); o++) {
            bool any_seen = false, any_new = false;
            for (int64_t c = p->dv_off_cols[o]; c < p->dv_off_cols[o + 1]; c++) {
                const int64_t gp = p->dv_param_idx[(size_t)c];
                if (dest_of[(size_t)gp] < 0) continue;
                if (seen[(size_t)gp]) any_seen = true; else any_new = true;
            }
            first_writer[o] = (any_new && !any_seen) ? 1 : 0;
            // an object that is neither purely first nor purely later (it shares SOME columns) must add: those columns need zeros
            if (any_new && any_seen) all_covered = false;
            for (int64_t c = p->dv_off_cols[o]; c < p->dv_off_cols[o + 1]; c++) seen[(size_t)p->dv_param_idx[(size_t)c]] = 1;
        }
        for (int64_t c = 0; c < n_param; c++) all_covered = all_covered && seen[(size_t)param_idx[c]];
        if (!all_covered) std::fill(first_writer.begin(), first_writer.end(), (uint8_t)0);
    }
    bool window = true;
    for (int64_t c = 1; dest_idx && c < n_param; c++) window = window && dest_idx[c] == dest_idx[0] + c;
    if (all_covered) {
        // (every requested column gets its first value by a store)
    } else if (window) {
        const int64_t d0 = dest_idx ? dest_idx[0] : 0;
        HIP_TRY(hipMemset2DAsync(d_out + d0, (size_t)ld * 8, 0, (size_t)n_param * 8, (size_t)nE, p->stream));
    } else {
        for (int64_t c = 0; c < n_param; c++) HIP_TRY(hipMemset2DAsync(d_out + dest_idx[c], (size_t)ld * 8, 0, 8, (size_t)nE, p->stream));
    }
    const int64_t base_rho = 0, base_eff = (int64_t)h.n_rhos * D, base_gate = base_eff + (int64_t)h.n_effects * D;
    HIP_TRY(p->d_dv_colmap.ensure((size_t)std::max<int64_t>(p->dv_off_cols.back(), 1)));
    std::vector<int32_t> colmap((size_t)p->dv_off_cols.back());
    for (size_t c = 0; c < colmap.size(); c++) colmap[c] = dest_of[(size_t)p->dv_param_idx[c]];
    if (!colmap.empty()) HIP_TRY(hipMemcpyAsync(p->d_dv_colmap.p, colmap.data(), colmap.size() * 4, hipMemcpyHostToDevice, p->stream));
    HIP_TRY(hipStreamSynchronize(p->stream));          // `colmap` goes out of scope
    for (size_t o = 0; o < p->dv_kind.size(); o++) {
        const int k = p->dv_kind[o];
        const int K = k == GST_KIND_GATE ? D * D : D;
        const int64_t a0 = (k == GST_KIND_GATE ? base_gate : k == GST_KIND_RHO ? base_rho : base_eff) + (int64_t)p->dv_obj[o] * K;
        HIP_TRY(gst::launch_chain_rule_gemm(p->d_jelem.p, n_el, a0, K, p->d_dv_deriv.p + p->dv_off_deriv[o], p->dv_ncols[o],
                                            p->d_dv_colmap.p + p->dv_off_cols[o], d_out, ld, nE, p->stream, first_writer[o] != 0));
        p->last_launches++;
    }
    return GST_OK;
}

// GST_DERIV_ANALYTIC for a Lindblad-parameterised model: the members' d(dense)/d(parameter) matrices are computed ON THE
// DEVICE (lindblad_deriv_kernel: Frechet derivative of the exponential, composed with the static factor) into the
// buffers gst_set_derivs would have filled from the host's deriv_wrt_params(), then the ordinary chain rule runs.
int run_dprobs_lindblad_analytic(gst_plan* p, double* d_out, int64_t ld, const int64_t* param_idx, const int64_t* dest_idx,
                                 int64_t n_param, double* d_probs_out)
{
    const gst::HostPlan& h = p->hp;
    const int D = h.D;
    gst_plan::Lindblad& L = p->lb;
    if (!L.have_theta) return fail(GST_ESTATE, "gst_set_lindblad_params has not been called");
    // objects in gst_set_derivs' terms: a gate / preparation member is one object, a POVM member one per effect
    std::vector<int32_t> kind, obj, ncols;
    std::vector<int64_t> pidx, member_off((size_t)L.n_members, 0), set_param;
    int64_t doff = 0;
    for (int m = 0; m < L.n_members; m++) {
        const int np = L.n_par[(size_t)m];
        member_off[(size_t)m] = doff;
        const int reps = L.kind[(size_t)m] == GST_KIND_EFFECT ? L.n_eff[(size_t)m] : 1;
        for (int e = 0; e < reps; e++) {
            kind.push_back(L.kind[(size_t)m]); obj.push_back(L.obj[(size_t)m] + e); ncols.push_back(np);
            for (int q = 0; q < np; q++) pidx.push_back(L.param0[(size_t)m] + q);
            doff += (int64_t)(L.kind[(size_t)m] == GST_KIND_GATE ? D * D : D) * np;
        }
        for (int q = 0; q < np; q++) set_param.push_back(L.param0[(size_t)m] + q);
    }
    const int32_t n_objs = (int32_t)kind.size();
    p->dv_kind = kind; p->dv_obj = obj; p->dv_ncols = ncols; p->dv_param_idx = pidx;
    p->dv_off_cols.assign((size_t)n_objs + 1, 0); p->dv_off_deriv.assign((size_t)n_objs + 1, 0);
    for (int32_t o = 0; o < n_objs; o++) {
        p->dv_off_cols[(size_t)o + 1] = p->dv_off_cols[(size_t)o] + ncols[(size_t)o];
        p->dv_off_deriv[(size_t)o + 1] = p->dv_off_deriv[(size_t)o] + (int64_t)(kind[(size_t)o] == GST_KIND_GATE ? D * D : D) * ncols[(size_t)o];
    }
    p->dv_n_params = L.n_params;
    p->dv_deriv_h.clear();                 // (no host copy: exact Hessian blocks of Lindblad models still come through gst_set_derivs)
    p->dv2_set = false; p->dv2_off.clear();
    HIP_TRY(p->d_dv_deriv.ensure((size_t)std::max<int64_t>(doff, 1)));
    HIP_TRY(p->d_lb_setparam.ensure(set_param.size() + member_off.size()));
    HIP_TRY(hipMemcpyAsync(p->d_lb_setparam.p, set_param.data(), set_param.size() * 8, hipMemcpyHostToDevice, p->stream));
    HIP_TRY(hipMemcpyAsync(p->d_lb_setparam.p + set_param.size(), member_off.data(), member_off.size() * 8, hipMemcpyHostToDevice, p->stream));
    gst::LbArgs a;
    lb_args(p, a);
    a.set_param = p->d_lb_setparam.p; a.deriv_out = p->d_dv_deriv.p; a.deriv_off = p->d_lb_setparam.p + set_param.size(); a.eps = 0.0;
    HIP_TRY(gst::launch_lindblad_derivs(D, a, (int64_t)set_param.size(), p->stream));
    HIP_TRY(hipStreamSynchronize(p->stream));          // (the host vectors above go out of scope)
    p->last_launches++;
    p->cached_kind = 0;
    return run_dprobs_general(p, d_out, ld, param_idx, dest_idx, n_param, d_probs_out);
}

int gst_fill_probs_dev(gst_plan* p, double* d_out)
{
    return guarded([&]() -> int {
    int rc = begin_call(p);
    if (rc) return rc;
    if (!d_out) return fail(GST_EINVAL, "d_out is NULL");
    gst::track_touch(d_out, (size_t)p->hp.n_elements * 8);
    TIME_REC(p, evk0);
    if ((rc = run_probs_any(p, d_out))) return rc;
    TIME_REC(p, evk1);
    return end_call(p, false);
    });
}

int gst_fill_probs(gst_plan* p, double* out)
{
    return guarded([&]() -> int {
    int rc = begin_call(p);
    if (rc) return rc;
    if (!out) return fail(GST_EINVAL, "out is NULL");
    TIME_REC(p, evk0);
    if ((rc = run_probs_any(p, p->d_pbase.p))) return rc;
    TIME_REC(p, evk1);
    HIP_TRY(hipMemcpyAsync(out, p->d_pbase.p, p->hp.n_elements * 8, hipMemcpyDeviceToHost, p->stream));
    return end_call(p, true);
    });
}

int gst_fill_dprobs_dev(gst_plan* p, double* d_out, int64_t ld, const int64_t* param_idx, const int64_t* dest_idx,
                        int64_t n_param, int mode, double eps, double* d_probs_out)
{
    return guarded([&]() -> int {
    int rc = begin_call(p);
    if (rc) return rc;
    if (mode != GST_DERIV_FD && mode != GST_DERIV_ANALYTIC) return fail(GST_EINVAL, "unknown derivative mode");
    if (!d_out && n_param > 0) return fail(GST_EINVAL, "d_out is NULL");
    // what this call overwrites no longer holds an earlier exact Jacobian's zeros (the plain exact fill keeps its own books)
    if (d_probs_out) gst::track_touch(d_probs_out, (size_t)p->hp.n_elements * 8);
    if (n_param > 0 && (mode == GST_DERIV_FD || p->lb.set || p->derivs_set)) gst::track_touch(d_out, jac_extent(p->hp.n_elements, ld, dest_idx, n_param));
    if (p->lb.set && !p->derivs_set) {
        if (n_param < 0 || (n_param > 0 && !param_idx)) return fail(GST_EINVAL, "bad parameter list");
        if (mode == GST_DERIV_FD) rc = run_dprobs_lindblad(p, d_out, ld, param_idx, dest_idx, n_param, eps, d_probs_out);
        else rc = run_dprobs_lindblad_analytic(p, d_out, ld, param_idx, dest_idx, n_param, d_probs_out);
        if (rc) return rc;
        return end_call(p, false);
    }
    if (p->lb.set && mode == GST_DERIV_FD) {
        if (n_param < 0 || (n_param > 0 && !param_idx)) return fail(GST_EINVAL, "bad parameter list");
        if ((rc = run_dprobs_lindblad(p, d_out, ld, param_idx, dest_idx, n_param, eps, d_probs_out))) return rc;
        return end_call(p, false);
    }
    if (p->derivs_set) {
        if (mode != GST_DERIV_ANALYTIC) return fail(GST_EUNSUPPORTED, "general parameterisations (gst_set_derivs) exist in GST_DERIV_ANALYTIC only");
        if (n_param < 0 || (n_param > 0 && !param_idx)) return fail(GST_EINVAL, "bad parameter list");
        if ((rc = run_dprobs_general(p, d_out, ld, param_idx, dest_idx, n_param, d_probs_out))) return rc;
        return end_call(p, false);
    }
    if (!p->have_pmap) return fail(GST_ESTATE, "gst_set_param_map has not been called");
    if ((rc = check_params(p, param_idx, n_param))) return rc;
    if (mode == GST_DERIV_ANALYTIC) rc = run_dprobs_analytic(p, d_out, ld, param_idx, dest_idx, n_param, d_probs_out);
    else rc = run_dprobs_fd(p, d_out, ld, param_idx, dest_idx, n_param, eps, d_probs_out, nullptr, 0);
    if (rc) return rc;
    return end_call(p, false);
    });
}

int gst_fill_dprobs(gst_plan* p, double* out, int64_t ld, const int64_t* param_idx, const int64_t* dest_idx,
                    int64_t n_param, int mode, double eps, double* probs_out)
{
    return guarded([&]() -> int {
    int rc = begin_call(p);
    if (rc) return rc;
    if (mode != GST_DERIV_FD && mode != GST_DERIV_ANALYTIC) return fail(GST_EINVAL, "unknown derivative mode");
    if (!out && n_param > 0) return fail(GST_EINVAL, "out is NULL");
    if (p->lb.set && (mode == GST_DERIV_FD || !p->derivs_set)) {
        if (n_param < 0 || (n_param > 0 && !param_idx)) return fail(GST_EINVAL, "bad parameter list");
        if ((rc = stage_out(p, (size_t)p->hp.n_elements * std::max<int64_t>(n_param, 1)))) return rc;
        if (mode == GST_DERIV_FD) rc = run_dprobs_lindblad(p, p->d_out.p, n_param, param_idx, nullptr, n_param, eps, nullptr);
        else rc = run_dprobs_lindblad_analytic(p, p->d_out.p, n_param, param_idx, nullptr, n_param, nullptr);
        if (rc) return rc;
        return copy_out_dprobs(p, out, ld, dest_idx, n_param, probs_out);
    }
    if (p->derivs_set && mode != GST_DERIV_ANALYTIC)
        return fail(GST_EUNSUPPORTED, "general parameterisations (gst_set_derivs) exist in GST_DERIV_ANALYTIC only");
    if (!p->derivs_set && !p->have_pmap) return fail(GST_ESTATE, "gst_set_param_map has not been called");
    if (!p->derivs_set && (rc = check_params(p, param_idx, n_param))) return rc;
    if (n_param < 0 || (n_param > 0 && !param_idx)) return fail(GST_EINVAL, "bad parameter list");
    const int64_t nE = p->hp.n_elements;
    // A page-locked destination (gst_host_register): the FD kernel writes the Jacobian straight into it -- 512-byte row
    // segments over PCIe while the walk is still computing -- instead of filling 7 GB of HBM first and copying afterwards
    // (kernel and transfer overlap completely; the (ld, dest_idx) window is honoured by the kernel itself).
    // (the analytic contraction writes every requested entry exactly once as well -- streaming stores -- and takes the same route)
    if (!p->derivs_set && (mode == GST_DERIV_FD || mode == GST_DERIV_ANALYTIC) && n_param >= p->host_direct_min_cols && nE > 0 && p->hp.D <= 16 &&
        p->comp_index < 0 && p->host_direct && !(mode == GST_DERIV_ANALYTIC && p->ana_keep_zeros == 1)) {
        int64_t max_col = 0;
        bool plain = true;          // (analytic: columns of parameters the atom never uses are zero-filled by a 2-D memset -- staged route)
        for (int64_t c = 0; c < n_param; c++) {
            max_col = std::max<int64_t>(max_col, dest_idx ? dest_idx[c] : c);
            if (mode == GST_DERIV_ANALYTIC && p->pkind[(size_t)param_idx[c]] == GST_KIND_NONE) plain = false;
        }
        void* d_host = plain ? mapped_device_pointer(out, (size_t)((nE - 1) * ld + max_col + 1) * 8) : nullptr;
        if (d_host) {
            if (mode == GST_DERIV_ANALYTIC) rc = run_dprobs_analytic(p, (double*)d_host, ld, param_idx, dest_idx, n_param, nullptr);
            else rc = run_dprobs_fd(p, (double*)d_host, ld, param_idx, dest_idx, n_param, eps, nullptr, nullptr, 0);
            if (rc) return rc;
            if (probs_out) HIP_TRY(hipMemcpyAsync(probs_out, p->d_pbase.p, nE * 8, hipMemcpyDeviceToHost, p->stream));
            return end_call(p, true);
        }
    }
    // device staging is dense [nE][n_param]; scattered into the caller's (ld, dest_idx) window on the host
    if ((rc = stage_out(p, (size_t)nE * std::max<int64_t>(n_param, 1), !p->derivs_set && mode == GST_DERIV_ANALYTIC))) return rc;
    if (p->derivs_set) rc = run_dprobs_general(p, p->d_out.p, n_param, param_idx, nullptr, n_param, nullptr);
    else if (mode == GST_DERIV_ANALYTIC) rc = run_dprobs_analytic(p, p->d_out.p, n_param, param_idx, nullptr, n_param, nullptr);
    else rc = run_dprobs_fd(p, p->d_out.p, n_param, param_idx, nullptr, n_param, eps, nullptr, nullptr, 0);
    if (rc) return rc;
    return copy_out_dprobs(p, out, ld, dest_idx, n_param, probs_out);
    });
}

int gst_fill_dprobs_models_dev(gst_plan* p, int64_t n_models, const double* gates, const double* rhos, const double* effects,
                               double* d_out, int64_t ld, const int64_t* dest_idx, double eps, double* d_probs_out)
{
    return guarded([&]() -> int {
        int rc = begin_call(p);
        if (rc) return rc;
        if (n_models < 0 || (n_models > 0 && (!rhos || !effects || (p->hp.n_gates > 0 && !gates) || !d_out)))
            return fail(GST_EINVAL, "bad argument");
        if (!(eps != 0.0)) return fail(GST_EINVAL, "eps must be non-zero");
        if (d_probs_out) gst::track_touch(d_probs_out, (size_t)p->hp.n_elements * 8);
        gst::track_touch(d_out, jac_extent(p->hp.n_elements, ld, dest_idx, n_models));
        if ((rc = run_dprobs_models(p, n_models, gates, rhos, effects, d_out, ld, dest_idx, eps, d_probs_out))) return rc;
        return end_call(p, false);
    });
}

int gst_fill_dprobs_models(gst_plan* p, int64_t n_models, const double* gates, const double* rhos, const double* effects,
                           double* out, int64_t ld, const int64_t* dest_idx, double eps, double* probs_out)
{
    return guarded([&]() -> int {
        int rc = begin_call(p);
        if (rc) return rc;
        if (n_models < 0 || (n_models > 0 && (!rhos || !effects || (p->hp.n_gates > 0 && !gates) || !out)))
            return fail(GST_EINVAL, "bad argument");
        if (!(eps != 0.0)) return fail(GST_EINVAL, "eps must be non-zero");
        const int64_t nE = p->hp.n_elements;
        if ((rc = stage_out(p, (size_t)nE * std::max<int64_t>(n_models, 1)))) return rc;
        if ((rc = run_dprobs_models(p, n_models, gates, rhos, effects, p->d_out.p, n_models, nullptr, eps, nullptr))) return rc;
        return copy_out_dprobs(p, out, ld, dest_idx, n_models, probs_out);
    });
}

// The host copy of the model with parameter `pi` stepped by eps, as `model.from_vector(vec)` leaves a model of
// one-parameter-per-element members (mapforwardsim.py:425-428); a TP POVM's complement is re-derived from the stepped
// effect in the reference's summation order (complementeffect.py:72-78).
static void step_host_model(gst_plan* p, int64_t pi, double eps)
{
    const int D = p->hp.D;
    const int32_t k = p->pkind[pi], o = p->pobj[pi], el = p->pelem[pi];
    if (k == GST_KIND_GATE) {
        const size_t at = (size_t)o * D * D + el;
        const double v = p->h_gates[at] + eps;
        p->h_gates[at] = v;
        p->h_gates_t[((size_t)o * D + el % D) * D + el / D] = v;
    } else if (k == GST_KIND_RHO) {
        p->h_rhos[(size_t)o * D + el] = p->h_rhos[(size_t)o * D + el] + eps;
    } else if (k == GST_KIND_EFFECT) {
        p->h_effects[(size_t)o * D + el] = p->h_effects[(size_t)o * D + el] + eps;
        if (p->comp_index >= 0 && std::find(p->comp_others.begin(), p->comp_others.end(), o) != p->comp_others.end()) {
            double sum = 0.0;
            for (int32_t q : p->comp_others) sum = sum + p->h_effects[(size_t)q * D + el];
            p->h_effects[(size_t)p->comp_index * D + el] = p->comp_identity[el] - sum;
        }
    }
    p->model_dirty = true;
}

// FD-of-FD Hessian block COMPOSED from FD Jacobians, literally as MapForwardSimulator._mapfill_hprobs_atom does it
// (mapforwardsim.py:420-436): dprobs over block 2 at theta; for every row parameter i the model is stepped to
// theta + eps e_i, dprobs2 = its FD Jacobian over block 2 (own base pass, `(orig + eps) + eps` where i is in block 2),
// row i = (dprobs2 - dprobs) / eps.  Every Jacobian is bit-identical to the reference's on its model, hence so is the
// block.  The route for plans the fused two-perturbation kernels do not cover (D = 64 with a complement effect): n1 + 2
// Jacobian passes instead of one fused launch.  Leaves behind what run_hprobs_dev leaves behind.
static int run_hprobs_composed(gst_plan* p, double* d_H, int64_t ld1, int64_t ld2, const int64_t* idx1, const int64_t* dest1,
                               int64_t n1, const int64_t* idx2, const int64_t* dest2, int64_t n2, double eps)
{
    int rc;
    const int64_t nE = p->hp.n_elements;
    HIP_TRY(p->d_dcol.ensure((size_t)nE * std::max<int64_t>(n2, 1)));
    HIP_TRY(p->d_hrow.ensure((size_t)nE * std::max<int64_t>(n2, 1)));
    HIP_TRY(p->d_probs_tmp.ensure((size_t)nE * std::max<int64_t>(n1, 1)));
    if ((rc = run_dprobs_fd(p, p->d_dcol.p, n2, idx2, nullptr, n2, eps, nullptr, nullptr, 0))) return rc;
    const int32_t* d_dest2 = nullptr;
    if (dest2) {
        std::vector<int32_t> d2(dest2, dest2 + n2);
        if ((rc = upload_i32(p->d_hdest, d2, p->stream))) return rc;
        d_dest2 = p->d_hdest.p;
    }
    const std::vector<double> g0 = p->h_gates, gt0 = p->h_gates_t, r0 = p->h_rhos, e0 = p->h_effects;
    auto restore = [&]() { p->h_gates = g0; p->h_gates_t = gt0; p->h_rhos = r0; p->h_effects = e0; p->model_dirty = true; };
    for (int64_t a = 0; a < n1; a++) {
        step_host_model(p, idx1[a], eps);
        if ((rc = upload_model(p)) || (rc = run_dprobs_fd(p, p->d_hrow.p, n2, idx2, nullptr, n2, eps, nullptr, nullptr, 0))) { restore(); return rc; }
        hipError_t he = gst::launch_hess_compose(p->d_hrow.p, p->d_dcol.p, nE, (int32_t)n2, eps, d_H, ld1, ld2, dest1 ? dest1[a] : a, d_dest2, p->stream);
        restore();
        if (he != hipSuccess) return fail(GST_EHIP, std::string("hess_compose: ") + hipGetErrorString(he));
    }
    if ((rc = upload_model(p))) return rc;
    // (the by-products the objective-Hessian rectangle reads: probabilities at theta, dprobs over block 1)
    return run_dprobs_fd(p, p->d_probs_tmp.p, n1, idx1, nullptr, n1, eps, nullptr, nullptr, 0);
}

// FD-of-FD Hessian block into the device buffer d_H [nE][ld1][ld2] (mapforwardsim.py:394-438).  Leaves behind, on the
// device: probabilities (d_pbase), FD dprobs over block 2 (d_dcol, [nE][n2]) and over block 1 (d_probs_tmp, [nE][n1]).
static int run_hprobs_dev(gst_plan* p, double* d_H, int64_t ld1, int64_t ld2, const int64_t* idx1, const int64_t* dest1,
                          int64_t n1, const int64_t* idx2, const int64_t* dest2, int64_t n2, double eps)
{
    int rc;
    if (p->comp_index >= 0) {
        for (int64_t c = 0; c < n1 + n2; c++) {
            const int64_t pi = c < n1 ? idx1[c] : idx2[c - n1];
            if (p->pkind[pi] == GST_KIND_EFFECT && p->pobj[pi] == p->comp_index) return fail(GST_EINVAL, "a parameter maps to the complement effect");
        }
    }
    // the fused two-perturbation kernels have no D = 64 form that re-derives a complement effect: that block is composed
    if (p->hess_composed || (p->comp_index >= 0 && p->hp.D == 64))
        return run_hprobs_composed(p, d_H, ld1, ld2, idx1, dest1, n1, idx2, dest2, n2, eps);
    const int64_t nE = p->hp.n_elements;
    // (1) dprobs over block 2 at theta (mapforwardsim.py:420-421), FD step = eps
    HIP_TRY(p->d_dcol.ensure((size_t)nE * n2));
    if ((rc = run_dprobs_fd(p, p->d_dcol.p, n2, idx2, nullptr, n2, eps, nullptr, nullptr, 0))) return rc;
    // (2) probabilities at theta + eps e_i for every i of block 1 (the `probs` of the inner FD, pyx:349)
    HIP_TRY(p->d_raw.ensure((size_t)nE * n1));
    HIP_TRY(p->d_probs_tmp.ensure((size_t)nE * n1));
    if ((rc = run_dprobs_fd(p, p->d_probs_tmp.p, n1, idx1, nullptr, n1, eps, nullptr, p->d_raw.p, n1))) return rc;
    // (3) all (i, j) pairs: wavefront = (row i, 64 columns j)
    const bool rows = (p->hp.D == 64);
    LaneLayout L2;
    if (rows) pack_waves(p, idx2, nullptr, n2, L2);
    else pack_lanes(p, idx2, nullptr, n2, L2);   // col = position in block 2
    const int32_t w2 = L2.n_waves;
    LaneLayout L;
    std::vector<int32_t> wave_row, wave_rowidx, lane_colidx;
    for (int64_t a = 0; rows && a < n1; a++) {       // one wavefront per (i, j) pair
        const int64_t pi = idx1[a];
        for (int32_t w = 0; w < w2; w++) {
            wave_row.push_back((int32_t)(dest1 ? dest1[a] : a));
            wave_rowidx.push_back((int32_t)a);
            const int32_t c = L2.col[w];
            L.col.push_back((int32_t)(dest2 ? dest2[c] : c));
            lane_colidx.push_back(c);
            L.kind[0].push_back(p->pkind[pi]); L.obj[0].push_back(p->pobj[pi]); L.elem[0].push_back(p->pelem[pi]);
            L.kind[1].push_back(L2.kind[0][w]); L.obj[1].push_back(L2.obj[0][w]); L.elem[1].push_back(L2.elem[0][w]);
        }
    }
    for (int64_t a = 0; !rows && a < n1; a++) {
        const int64_t pi = idx1[a];
        for (int32_t w = 0; w < w2; w++) {
            wave_row.push_back((int32_t)(dest1 ? dest1[a] : a));
            wave_rowidx.push_back((int32_t)a);
            for (int q = 0; q < 64; q++) {
                const size_t s = (size_t)w * 64 + q;
                const int32_t c = L2.col[s];
                L.col.push_back(c < 0 ? -1 : (int32_t)(dest2 ? dest2[c] : c));
                lane_colidx.push_back(c < 0 ? 0 : c);
                L.kind[0].push_back(p->pkind[pi]); L.obj[0].push_back(p->pobj[pi]); L.elem[0].push_back(p->pelem[pi]);
                L.kind[1].push_back(L2.kind[0][s]); L.obj[1].push_back(L2.obj[0][s]); L.elem[1].push_back(L2.elem[0][s]);
            }
        }
    }
    L.n_waves = rows ? (int32_t)L.col.size() : (int32_t)(L.col.size() / 64);
    p->cached_kind = 0;      // the shared lane tables are about to hold the (i, j) pairs
    if ((rc = upload_i32(p->d_lane[0], L.col, p->stream))) return rc;
    for (int s = 0; s < 2; s++) {
        if ((rc = upload_i32(p->d_lane[1 + 3 * s], L.kind[s], p->stream))) return rc;
        if ((rc = upload_i32(p->d_lane[2 + 3 * s], L.obj[s], p->stream))) return rc;
        if ((rc = upload_i32(p->d_lane[3 + 3 * s], L.elem[s], p->stream))) return rc;
    }
    if ((rc = upload_i32(p->d_wave_row, wave_row, p->stream))) return rc;
    if ((rc = upload_i32(p->d_wave_rowidx, wave_rowidx, p->stream))) return rc;
    if ((rc = upload_i32(p->d_lane_colidx, lane_colidx, p->stream))) return rc;
    HIP_TRY(hipStreamSynchronize(p->stream));
    gst::WalkArgs a;
    base_args(p, a);
    a.mode = gst::EMIT_HESS;
    a.out = d_H; a.ld = ld1; a.ld2 = ld2; a.eps = eps;
    a.prow = p->d_raw.p; a.ldrow = n1; a.dcol = p->d_dcol.p; a.lddcol = n2;
    a.pbase = p->d_pbase.p; a.base_cache = p->d_base_cache.p;
    a.lanes.col = p->d_lane[0].p;
    for (int s = 0; s < 2; s++) {
        a.lanes.kind[s] = p->d_lane[1 + 3 * s].p; a.lanes.obj[s] = p->d_lane[2 + 3 * s].p; a.lanes.elem[s] = p->d_lane[3 + 3 * s].p;
    }
    a.wave_row = p->d_wave_row.p; a.wave_rowidx = p->d_wave_rowidx.p; a.lane_colidx = p->d_lane_colidx.p;
    a.n_pwaves = L.n_waves;
    TIME_REC(p, evk0);
    if (rows) {
        a.rows_S = 2;
        HIP_TRY(gst::launch_walk_rows(p->hp.D, a, p->hp.n_tasks(), p->hp.max_slots, p->stream));
    } else {
        const bool comp = p->comp_index >= 0;
        if (comp) {       // the complement description rides in the (otherwise unused here) effect-column tables
            if ((rc = upload_i32(p->d_ecol_tab, p->comp_others, p->stream))) return rc;
            HIP_TRY(p->d_ecol_val.ensure(p->comp_identity.size()));
            HIP_TRY(hipMemcpyAsync(p->d_ecol_val.p, p->comp_identity.data(), p->comp_identity.size() * 8, hipMemcpyHostToDevice, p->stream));
            a.comp_index = p->comp_index; a.n_others = (int32_t)p->comp_others.size();
            a.comp_others = p->d_ecol_tab.p; a.comp_identity = p->d_ecol_val.p;
            p->cached_kind = 0;
        }
        HIP_TRY(gst::launch_walk(p->hp.D, 2, a, p->hp.n_tasks(), p->hp.max_slots, p->stream, 1, comp));
    }
    TIME_REC(p, evk1);
    p->last_launches++;
    return GST_OK;
}

// Exact Hessian block (what MatrixForwardSimulator returns) into the device buffer d_H [nE][ld1][ld2], D = 16 / 64:
//   H[e, t1, t2] = sum_{k: g_k = g2} B_k[a2] dF^{t1}_{k-1}[b2] + dB^{t1}_k[a2] F_{k-1}[b2]     (+ the SPAM columns)
// with the derivative states dF^{t1} (forward plan) and dB^{t1} (reversed plan, per effect) of four rows t1 at a time
// (dwalk_kernel) and the Jacobian's MFMA contraction run twice per row with one of the two caches swapped.
static int run_hprobs_analytic(gst_plan* p, double* d_H, int64_t ld1, int64_t ld2, const int64_t* idx1, const int64_t* dest1,
                               int64_t n1, const int64_t* idx2, const int64_t* dest2, int64_t n2)
{
    const gst::HostPlan& h = p->hp;
    if (!p->ana_mfma) return fail(GST_EUNSUPPORTED, "analytic Hessians need the two-cache contraction path");
    const int D = h.D, nEf = h.n_effects;
    const int64_t nE = h.n_elements;
    int rc;
    // set-up through the Jacobian path: column maps of block 2 (in the caller's column numbering), F and B caches
    HIP_TRY(p->d_hscratch.ensure((size_t)nE * ld2));
    p->last_ana_valid = false;
    p->want_cache_path = true;
    rc = run_dprobs_analytic(p, p->d_hscratch.p, ld2, idx2, dest2, n2, nullptr);
    p->want_cache_path = false;
    if (rc) return rc;
    if (!p->last_ana_valid) return fail(GST_EUNSUPPORTED, "analytic Hessians need the two-cache contraction path");
    gst::AnaArgs base = p->last_ana;
    const std::vector<int64_t> none_cols = p->cached_none_cols;
    {   // derivative-state caches (four row parameters per state) of 4 GB or more: the contraction's wide form
        const double cache_limit = p->test_cache_limit > 0 ? p->test_cache_limit : 4.0e9;
        if ((double)p->rev.n_state_ids * 4 * D * nEf * 8 >= cache_limit || (double)h.n_state_ids * 4 * D * 8 >= cache_limit) base.wide = 1;
    }
    HIP_TRY(p->d_dF.ensure((size_t)h.n_state_ids * 4 * D));
    HIP_TRY(p->d_dB.ensure((size_t)p->rev.n_state_ids * 4 * D * nEf));
    HIP_TRY(p->d_theta.ensure(5 * 4 * (size_t)(1 + nEf)));
    for (int64_t i0 = 0; i0 < n1; i0 += 4) {
        const int nt = (int)std::min<int64_t>(4, n1 - i0);
        // parameter tables: block 0 = forward walk, block 1 + x = backward walk from effect x
        std::vector<int32_t> th((size_t)5 * 4 * (1 + nEf), 0);
        for (int v = 0; v < 1 + nEf; v++) {
            int32_t* t = th.data() + (size_t)v * 20;          // inj_gate[4] inj_dst[4] inj_src[4] start_obj[4] start_idx[4]
            for (int q = 0; q < 4; q++) {
                t[q] = -1; t[12 + q] = -2;
                if (q >= nt) continue;
                const int64_t pi = idx1[i0 + q];
                const int k = p->pkind[pi], o = p->pobj[pi], el = p->pelem[pi];
                if (k == GST_KIND_GATE) {
                    t[q] = o;
                    t[4 + q] = (v == 0) ? el / D : el % D;       // forward: row a1 receives F[b1]; backward: row b1 receives B[a1]
                    t[8 + q] = (v == 0) ? el % D : el / D;
                } else if (k == GST_KIND_RHO && v == 0) { t[12 + q] = o; t[16 + q] = el; }
                else if (k == GST_KIND_EFFECT && v > 0 && o == v - 1) { t[12 + q] = -1; t[16 + q] = el; }
            }
        }
        HIP_TRY(hipMemcpyAsync(p->d_theta.p, th.data(), th.size() * 4, hipMemcpyHostToDevice, p->stream));
        HIP_TRY(hipStreamSynchronize(p->stream));
        gst::DWalkArgs w;
        std::memset(&w, 0, sizeof(w));
        w.n_gates = h.n_gates; w.n_theta = nt;
        auto tables = [&](int v) {
            const int32_t* t = p->d_theta.p + (size_t)v * 20;
            w.inj_gate = t; w.inj_dst = t + 4; w.inj_src = t + 8; w.start_obj = t + 12; w.start_idx = t + 16;
        };
        // dF over the forward plan
        tables(0);
        w.prog = p->d_prog.p; w.task_off = p->d_task_off.p; w.tile = p->d_gates_t.p;
        w.base = p->d_base_cache.p; w.bstride = D; w.bmul = 1; w.boff = 0;
        w.out = p->d_dF.p; w.ostride = D; w.omul = 1; w.ooff = 0;
        HIP_TRY(gst::launch_dwalk(D, w, h.n_tasks(), h.max_slots, p->stream));
        // dB over the reversed plan, one pass per effect
        w.prog = p->d_rprog.p; w.task_off = p->d_rtask_off.p; w.tile = p->d_gates.p;
        for (int x = 0; x < nEf; x++) {
            tables(1 + x);
            // backward-state layouts: D <= 16 [state][component][effect], D = 64 [state][effect][component]
            w.base = p->d_rev_cache.p; w.bstride = (int64_t)D * nEf;
            w.out = p->d_dB.p; w.ostride = (int64_t)D * nEf;
            if (D <= 16) { w.bmul = nEf; w.boff = x; w.omul = nEf; w.ooff = x; }
            else { w.bmul = 1; w.boff = (int64_t)x * D; w.omul = 1; w.ooff = (int64_t)x * D; }
            HIP_TRY(gst::launch_dwalk(D, w, p->rev.n_tasks(), p->rev.max_slots, p->stream));
        }
        p->last_launches += 1 + nEf;
        for (int q = 0; q < nt; q++) {
            const int64_t row = dest1 ? dest1[i0 + q] : i0 + q;
            gst::AnaArgs a = base;
            a.out = d_H + row * ld2; a.ld = ld1 * ld2;
            // theta_1 earlier than theta_2: derivative forward states against the backward states
            a.base_cache = p->d_dF.p + (size_t)q * D; a.fwd_stride = 4 * D * 8;
            a.rev_cache = p->d_rev_cache.p; a.rev_stride = 0;
            a.rho_zero = 1; a.eff_zero = 0; a.accumulate = 0;
            HIP_TRY(hipMemsetAsync(p->d_work_counter.p, 0, 8 * sizeof(uint32_t), p->stream));
            if (D == 64) HIP_TRY(gst::launch_analytic_mfma64(a, p->stream));
            else if (D == 16) HIP_TRY(gst::launch_analytic_mfma(a, p->stream));
            else HIP_TRY(gst::launch_analytic_small(a, p->stream));
            // theta_1 later: forward states against the derivative backward states, added
            a.base_cache = p->d_base_cache.p; a.fwd_stride = 0;
            a.rev_cache = p->d_dB.p + (size_t)q * D * nEf; a.rev_stride = (uint32_t)(4 * D * nEf * 8);
            a.rho_zero = 0; a.eff_zero = 1; a.accumulate = 1;
            HIP_TRY(hipMemsetAsync(p->d_work_counter.p, 0, 8 * sizeof(uint32_t), p->stream));
            if (D == 64) HIP_TRY(gst::launch_analytic_mfma64(a, p->stream));
            else if (D == 16) HIP_TRY(gst::launch_analytic_mfma(a, p->stream));
            else HIP_TRY(gst::launch_analytic_small(a, p->stream));
            p->last_launches += 2;
            for (int64_t col : none_cols)
                HIP_TRY(hipMemset2DAsync(d_H + row * ld2 + col, (size_t)ld1 * ld2 * 8, 0, 8, (size_t)nE, p->stream));
        }
    }
    return GST_OK;
}

// Exact Hessian block with gst_set_derivs, for parameterisations whose dense elements are LINEAR in the parameters
// (TP, ...): H_param[p1][p2] = sum_{a, b} (d elem_a / d p1) H_elem[a][b] (d elem_b / d p2), what
// MatrixForwardSimulator._hprobs_from_rho_e assembles when the members' hessian_wrt_params vanish
// (matrixforwardsim.py:1190-1287).  The element block is computed for the elements the requested parameters touch
// (identity element map, as run_dprobs_general does) and contracted with the sparse derivative columns on the device.
static int run_hprobs_general(gst_plan* p, double* d_H, int64_t ld1, int64_t ld2, const int64_t* idx1, const int64_t* dest1,
                              int64_t n1, const int64_t* idx2, const int64_t* dest2, int64_t n2)
{
    const gst::HostPlan& h = p->hp;
    const int D = h.D;
    const int64_t nE = h.n_elements;
    const int64_t n_el = (int64_t)h.n_rhos * D + (int64_t)h.n_effects * D + (int64_t)h.n_gates * D * D;
    const int64_t base_rho = 0, base_eff = (int64_t)h.n_rhos * D, base_gate = base_eff + (int64_t)h.n_effects * D;
    struct Csc { std::vector<int32_t> ptr, row, dest; std::vector<double> w; std::vector<int64_t> elems; };
    auto build = [&](const int64_t* idx, const int64_t* dest, int64_t n, Csc& c) -> int {
        std::vector<int32_t> pos((size_t)p->dv_n_params, -1);
        for (int64_t k = 0; k < n; k++) {
            if (idx[k] < 0 || idx[k] >= p->dv_n_params) return fail(GST_EINVAL, "parameter index out of range");
            if (pos[(size_t)idx[k]] >= 0) return fail(GST_EINVAL, "a parameter is requested twice (not supported with gst_set_derivs)");
            pos[(size_t)idx[k]] = (int32_t)k;
        }
        std::vector<std::vector<std::pair<int64_t, double>>> cols((size_t)n);      // (global element, weight) per requested parameter
        for (size_t o = 0; o < p->dv_kind.size(); o++) {
            const int k = p->dv_kind[o];
            const int K = k == GST_KIND_GATE ? D * D : D;
            const int64_t a0 = (k == GST_KIND_GATE ? base_gate : k == GST_KIND_RHO ? base_rho : base_eff) + (int64_t)p->dv_obj[o] * K;
            const int nc = p->dv_ncols[o];
            const double* dm = p->dv_deriv_h.data() + p->dv_off_deriv[o];
            for (int c2 = 0; c2 < nc; c2++) {
                const int32_t at = pos[(size_t)p->dv_param_idx[(size_t)p->dv_off_cols[o] + c2]];
                if (at < 0) continue;
                for (int r = 0; r < K; r++)
                    if (dm[(size_t)r * nc + c2] != 0.0) cols[(size_t)at].emplace_back(a0 + r, dm[(size_t)r * nc + c2]);
            }
        }
        std::vector<int64_t> el;
        for (auto& v : cols) for (auto& e : v) el.push_back(e.first);
        std::sort(el.begin(), el.end());
        el.erase(std::unique(el.begin(), el.end()), el.end());
        c.elems = el;
        c.ptr.assign((size_t)n + 1, 0);
        for (int64_t k = 0; k < n; k++) {
            for (auto& e : cols[(size_t)k]) {
                c.row.push_back((int32_t)(std::lower_bound(el.begin(), el.end(), e.first) - el.begin()));
                c.w.push_back(e.second);
            }
            c.ptr[(size_t)k + 1] = (int32_t)c.row.size();
            c.dest.push_back((int32_t)(dest ? dest[k] : k));
        }
        return GST_OK;
    };
    Csc c1, c2;
    int rc;
    if ((rc = build(idx1, dest1, n1, c1)) || (rc = build(idx2, dest2, n2, c2))) return rc;
    const int64_t m1 = (int64_t)c1.elems.size(), m2 = (int64_t)c2.elems.size();
    // (m1 == 0 or m2 == 0: nothing the atom applies depends on one of the blocks; the contraction then writes exact zeros)
    if ((double)nE * (double)std::max<int64_t>(m1, 1) * (double)std::max<int64_t>(m2, 1) * 8.0 > 64.0e9)
        return fail(GST_ENOMEM, "element-Hessian block too large: request smaller parameter blocks");
    HIP_TRY(p->d_helem.ensure((size_t)std::max<int64_t>(nE * m1 * m2, 1)));
    if (m1 > 0 && m2 > 0) {
        // element Hessian through the `full` path with the identity element map
        std::vector<int32_t> ek((size_t)n_el), eo((size_t)n_el), ee((size_t)n_el);
        int64_t q = 0;
        for (int r = 0; r < h.n_rhos; r++) for (int j = 0; j < D; j++, q++) { ek[q] = GST_KIND_RHO; eo[q] = r; ee[q] = j; }
        for (int e = 0; e < h.n_effects; e++) for (int j = 0; j < D; j++, q++) { ek[q] = GST_KIND_EFFECT; eo[q] = e; ee[q] = j; }
        for (int g = 0; g < h.n_gates; g++) for (int j = 0; j < D * D; j++, q++) { ek[q] = GST_KIND_GATE; eo[q] = g; ee[q] = j; }
        p->pkind.swap(ek); p->pobj.swap(eo); p->pelem.swap(ee);
        p->cached_kind = 0;
        const bool ds = p->derivs_set;
        const int32_t ci = p->comp_index;          // (a complement declared for the FD modes plays no role here: the
        p->derivs_set = false;                     //  derivative columns already carry its -1 entries)
        p->comp_index = -1;
        rc = run_hprobs_analytic(p, p->d_helem.p, m1, m2, c1.elems.data(), nullptr, m1, c2.elems.data(), nullptr, m2);
        p->derivs_set = ds;
        p->comp_index = ci;
        p->pkind.swap(ek); p->pobj.swap(eo); p->pelem.swap(ee);
        p->cached_kind = 0;
        if (rc) return rc;
    }
    // CSC tables: [ptr1 | row1 | dest1 | ptr2 | row2 | dest2], weights [w1 | w2]
    std::vector<int32_t> tab;
    const size_t o_p1 = 0, o_r1 = o_p1 + c1.ptr.size(), o_d1 = o_r1 + c1.row.size(), o_p2 = o_d1 + c1.dest.size(),
                 o_r2 = o_p2 + c2.ptr.size(), o_d2 = o_r2 + c2.row.size();
    tab.insert(tab.end(), c1.ptr.begin(), c1.ptr.end()); tab.insert(tab.end(), c1.row.begin(), c1.row.end());
    tab.insert(tab.end(), c1.dest.begin(), c1.dest.end());
    tab.insert(tab.end(), c2.ptr.begin(), c2.ptr.end()); tab.insert(tab.end(), c2.row.begin(), c2.row.end());
    tab.insert(tab.end(), c2.dest.begin(), c2.dest.end());
    std::vector<double> w(c1.w);
    w.insert(w.end(), c2.w.begin(), c2.w.end());
    if (w.empty()) w.push_back(0.0);
    if ((rc = upload_i32(p->d_hcsc, tab, p->stream))) return rc;
    HIP_TRY(p->d_hw.ensure(w.size()));
    HIP_TRY(hipMemcpyAsync(p->d_hw.p, w.data(), w.size() * 8, hipMemcpyHostToDevice, p->stream));
    HIP_TRY(gst::launch_hessian_chain_rule(p->d_helem.p, nE, (int)m1, (int)m2, p->d_hcsc.p + o_p1, p->d_hcsc.p + o_r1, p->d_hw.p,
                                           p->d_hcsc.p + o_d1, (int)n1, p->d_hcsc.p + o_p2, p->d_hcsc.p + o_r2,
                                           p->d_hw.p + c1.w.size(), p->d_hcsc.p + o_d2, (int)n2, d_H, ld1, ld2, p->stream));
    p->last_launches++;
    HIP_TRY(hipStreamSynchronize(p->stream));          // the host tables go out of scope
    if (p->dv2_set) {
        // members that are not linear in their parameters: + sum_a (d p / d elem_a) d^2 elem_a / d p1 d p2, one MFMA
        // product per such object: [nE x K] (element Jacobian) . [K x (n_o x n_o)] scattered into the block's entries
        if (ld1 * ld2 > 0x7fffffffLL) return fail(GST_EINVAL, "Hessian block too wide");
        if ((rc = run_element_jacobian(p, nullptr))) return rc;
        std::vector<int32_t> pos1((size_t)p->dv_n_params, -1), pos2((size_t)p->dv_n_params, -1);
        for (int64_t k = 0; k < n1; k++) pos1[(size_t)idx1[k]] = (int32_t)(dest1 ? dest1[k] : k);
        for (int64_t k = 0; k < n2; k++) pos2[(size_t)idx2[k]] = (int32_t)(dest2 ? dest2[k] : k);
        for (size_t o = 0; o < p->dv_kind.size(); o++) {
            if (p->dv2_off[o] < 0) continue;
            const int k = p->dv_kind[o];
            const int K = k == GST_KIND_GATE ? D * D : D;
            const int nc = p->dv_ncols[o];
            const int64_t a0 = (k == GST_KIND_GATE ? base_gate : k == GST_KIND_RHO ? base_rho : base_eff) + (int64_t)p->dv_obj[o] * K;
            std::vector<int32_t> cmap((size_t)nc * nc, -1);
            bool any = false;
            for (int ca = 0; ca < nc; ca++) {
                const int32_t i = pos1[(size_t)p->dv_param_idx[(size_t)p->dv_off_cols[o] + ca]];
                if (i < 0) continue;
                for (int cb = 0; cb < nc; cb++) {
                    const int32_t j = pos2[(size_t)p->dv_param_idx[(size_t)p->dv_off_cols[o] + cb]];
                    if (j >= 0) { cmap[(size_t)ca * nc + cb] = (int32_t)((int64_t)i * ld2 + j); any = true; }
                }
            }
            if (!any) continue;
            if ((rc = upload_i32(p->d_dv_colmap, cmap, p->stream))) return rc;
            HIP_TRY(gst::launch_chain_rule_gemm(p->d_jelem.p, n_el, a0, K, p->d_dv2.p + p->dv2_off[o], nc * nc, p->d_dv_colmap.p,
                                                d_H, ld1 * ld2, nE, p->stream));
            p->last_launches++;
            HIP_TRY(hipStreamSynchronize(p->stream));      // `cmap` goes out of scope; the next object reuses the buffer
        }
    }
    return GST_OK;
}

int gst_fill_hprobs_analytic(gst_plan* p, double* out, int64_t ld1, int64_t ld2, const int64_t* idx1, const int64_t* dest1,
                             int64_t n1, const int64_t* idx2, const int64_t* dest2, int64_t n2)
{
    return guarded([&]() -> int {
    int rc = begin_call(p);
    if (rc) return rc;
    if (!p->derivs_set && !p->have_pmap) return fail(GST_ESTATE, "gst_set_param_map has not been called");
    if (!out && n1 > 0 && n2 > 0) return fail(GST_EINVAL, "out is NULL");
    if (n1 < 0 || n2 < 0 || (n1 > 0 && !idx1) || (n2 > 0 && !idx2)) return fail(GST_EINVAL, "bad parameter list");
    if (!p->derivs_set && ((rc = check_params(p, idx1, n1)) || (rc = check_params(p, idx2, n2)))) return rc;
    const int64_t nE = p->hp.n_elements;
    if (n1 == 0 || n2 == 0) return end_call(p, true);
    if ((rc = stage_out(p, (size_t)nE * ld1 * ld2))) return rc;
    const bool dense = (!dest1 && !dest2 && ld1 == n1 && ld2 == n2);
    if (!dense) HIP_TRY(hipMemcpyAsync(p->d_out.p, out, (size_t)nE * ld1 * ld2 * 8, hipMemcpyHostToDevice, p->stream));
    if (p->derivs_set) rc = run_hprobs_general(p, p->d_out.p, ld1, ld2, idx1, dest1, n1, idx2, dest2, n2);
    else rc = run_hprobs_analytic(p, p->d_out.p, ld1, ld2, idx1, dest1, n1, idx2, dest2, n2);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(out, p->d_out.p, (size_t)nE * ld1 * ld2 * 8, hipMemcpyDeviceToHost, p->stream));
    return end_call(p, true);
    });
}

int gst_fill_hprobs(gst_plan* p, double* out, int64_t ld1, int64_t ld2, const int64_t* idx1, const int64_t* dest1,
                    int64_t n1, const int64_t* idx2, const int64_t* dest2, int64_t n2, double eps)
{
    return guarded([&]() -> int {
    int rc = begin_call(p);
    if (rc) return rc;
    if (p->derivs_set) return fail(GST_EUNSUPPORTED, "Hessians need the one-parameter-per-element map (gst_set_derivs is set)");
    if (!p->have_pmap) return fail(GST_ESTATE, "gst_set_param_map has not been called");
    if (!out && n1 > 0 && n2 > 0) return fail(GST_EINVAL, "out is NULL");
    if ((rc = check_params(p, idx1, n1)) || (rc = check_params(p, idx2, n2))) return rc;
    const int64_t nE = p->hp.n_elements;
    if (n1 == 0 || n2 == 0) return end_call(p, true);
    // dense device output [nE][n1'][n2'] in the caller's leading dimensions
    if ((rc = stage_out(p, (size_t)nE * ld1 * ld2))) return rc;
    // rows/columns of the caller's block that this call does not own must survive: start from the caller's data
    const bool dense = (!dest1 && !dest2 && ld1 == n1 && ld2 == n2);
    if (!dense) HIP_TRY(hipMemcpyAsync(p->d_out.p, out, (size_t)nE * ld1 * ld2 * 8, hipMemcpyHostToDevice, p->stream));
    if ((rc = run_hprobs_dev(p, p->d_out.p, ld1, ld2, idx1, dest1, n1, idx2, dest2, n2, eps))) return rc;
    HIP_TRY(hipMemcpyAsync(out, p->d_out.p, (size_t)nE * ld1 * ld2 * 8, hipMemcpyDeviceToHost, p->stream));
    return end_call(p, true);
    });
}

int gst_objective_hessian_block(gst_plan* p, const gst_objective_desc* d, const double* d_counts, const double* d_totals,
                                const int64_t* idx1, int64_t n1, const int64_t* idx2, int64_t n2, double eps, double* out)
{
    return guarded([&]() -> int {
    int rc = begin_call(p);
    if (rc) return rc;
    if (!d || !d_counts || !d_totals) return fail(GST_EINVAL, "bad argument");
    if (d->kind != GST_OBJ_CHI2 && d->kind != GST_OBJ_POISSON_DLOGL) return fail(GST_EINVAL, "unknown objective kind");
    if (!(d->min_prob_clip > 0.0) || (d->kind == GST_OBJ_POISSON_DLOGL && !(d->radius > 0.0)))
        return fail(GST_EINVAL, "min_prob_clip and radius must be positive");
    if (p->derivs_set && d->hessian_mode != GST_DERIV_ANALYTIC)
        return fail(GST_EUNSUPPORTED, "general parameterisations (gst_set_derivs) exist in GST_DERIV_ANALYTIC only");
    if (!p->derivs_set && !p->have_pmap) return fail(GST_ESTATE, "gst_set_param_map has not been called");
    if (!out && n1 > 0 && n2 > 0) return fail(GST_EINVAL, "out is NULL");
    if (n1 < 0 || n2 < 0 || (n1 > 0 && !idx1) || (n2 > 0 && !idx2)) return fail(GST_EINVAL, "bad parameter list");
    if (!p->derivs_set && ((rc = check_params(p, idx1, n1)) || (rc = check_params(p, idx2, n2)))) return rc;
    if (n1 > 0x7fffffff || n2 > 0x7fffffff) return fail(GST_EINVAL, "block too large");
    const int64_t nE = p->hp.n_elements;
    if (n1 == 0 || n2 == 0) return end_call(p, true);
    if ((rc = stage_out(p, (size_t)nE * n1 * n2))) return rc;
    const double* d_d1 = nullptr;
    const double* d_d2 = nullptr;
    if (d->hessian_mode == GST_DERIV_ANALYTIC && p->derivs_set) {
        // linear general parameterisation (TP): chain-ruled Jacobians of both blocks and the chain-ruled Hessian block
        HIP_TRY(p->d_probs_tmp.ensure((size_t)nE * n1));
        HIP_TRY(p->d_dcol.ensure((size_t)nE * n2));
        if ((rc = run_hprobs_general(p, p->d_out.p, n1, n2, idx1, nullptr, n1, idx2, nullptr, n2))) return rc;
        if ((rc = run_dprobs_general(p, p->d_probs_tmp.p, n1, idx1, nullptr, n1, nullptr))) return rc;
        if ((rc = run_dprobs_general(p, p->d_dcol.p, n2, idx2, nullptr, n2, nullptr))) return rc;
        d_d1 = p->d_probs_tmp.p; d_d2 = p->d_dcol.p;
    } else if (d->hessian_mode == GST_DERIV_ANALYTIC) {
        HIP_TRY(p->d_probs_tmp.ensure((size_t)nE * n1));
        if ((rc = run_dprobs_analytic(p, p->d_probs_tmp.p, n1, idx1, nullptr, n1, nullptr))) return rc;
        if ((rc = run_hprobs_analytic(p, p->d_out.p, n1, n2, idx1, nullptr, n1, idx2, nullptr, n2))) return rc;
        d_d1 = p->d_probs_tmp.p; d_d2 = p->d_hscratch.p;     // (the Hessian driver leaves the block-2 Jacobian in its scratch)
    } else if (d->hessian_mode == GST_DERIV_FD) {
        if ((rc = run_hprobs_dev(p, p->d_out.p, n1, n2, idx1, nullptr, n1, idx2, nullptr, n2, eps))) return rc;
        d_d1 = p->d_probs_tmp.p; d_d2 = p->d_dcol.p;
    } else return fail(GST_EINVAL, "unknown hessian_mode");
    // objective coefficients on the (optionally clipped) probabilities, then the contraction over elements
    HIP_TRY(p->d_obj_dt.ensure((size_t)nE)); HIP_TRY(p->d_obj_ht.ensure((size_t)nE));
    double* d_probs = p->d_pbase.p;
    if (d->prob_clip_lo < d->prob_clip_hi) {
        // _clip_probs (objectivefns.py:4766-4774) through the element-wise objective kernel, on a copy
        HIP_TRY(p->d_obj_pc.ensure((size_t)nE)); HIP_TRY(p->d_obj_tmp.ensure((size_t)2 * nE));
        HIP_TRY(hipMemcpyAsync(p->d_obj_pc.p, p->d_pbase.p, (size_t)nE * 8, hipMemcpyDeviceToDevice, p->stream));
        const int nb = (int)std::max<int64_t>(1, std::min<int64_t>(1024, (nE + 255) / 256));
        HIP_TRY(p->d_obj_part.ensure((size_t)nb));
        HIP_TRY(gst::launch_objective_rows(d->kind, p->d_obj_pc.p, d_counts, d_totals, nE, d->min_prob_clip, d->radius, d->prob_clip_lo,
                                           d->prob_clip_hi, p->d_obj_tmp.p, p->d_obj_tmp.p + nE, nullptr, p->d_obj_part.p, nb, p->stream));
        d_probs = p->d_obj_pc.p;
    }
    HIP_TRY(gst::launch_objective_coeffs(d->kind, d_probs, d_counts, d_totals, nE, d->min_prob_clip, d->radius, p->d_obj_dt.p,
                                         p->d_obj_ht.p, p->stream));
    const int n_slabs = gst::hessian_block_slabs(nE, (int)n1, (int)n2);
    HIP_TRY(p->d_hess_part.ensure((size_t)n_slabs * n1 * n2));
    HIP_TRY(p->d_hess_out.ensure((size_t)n1 * n2));
    HIP_TRY(gst::launch_hessian_block(p->d_out.p, d_d1, d_d2, p->d_obj_dt.p, p->d_obj_ht.p, nE, (int)n1, (int)n2,
                                      p->d_hess_part.p, n_slabs, p->d_hess_out.p, p->stream));
    p->last_launches += 2;
    HIP_TRY(hipMemcpyAsync(out, p->d_hess_out.p, (size_t)n1 * n2 * 8, hipMemcpyDeviceToHost, p->stream));
    return end_call(p, true);
    });
}

int gst_fill_jtj_dev(gst_plan* p, double* d_J, int64_t n_rows, int64_t n_cols, int64_t ld, const double* d_row_scale,
                     double* d_jtj)
{
    return guarded([&]() -> int {
    if (!p || !d_J || !d_jtj || n_rows < 0 || n_cols < 0 || ld < n_cols || n_cols > 0x7fffffff) return fail(GST_EINVAL, "bad argument");
    int rc = ensure_device(p);
    if (rc) return rc;
    gst::track_touch(d_jtj, (size_t)n_cols * n_cols * 8);
    if (d_row_scale && n_rows > 0 && n_cols > 0) {
        // the in-place row scaling keeps an exact Jacobian's zeros zero -- unless a factor is not finite (0 * inf): the
        // claim's device word is cleared on the stream then, and the next exact fill stores everything (gst_track.cpp)
        bool several = false;
        const size_t bytes = (size_t)((n_rows - 1) * ld + n_cols) * 8;
        uint32_t* w = gst::track_claim_overlapping(d_J, bytes, &several);
        if (several) gst::track_touch(d_J, bytes);
        else if (w) HIP_TRY(gst::launch_check_finite(d_row_scale, n_rows, w, p->stream));
    }
    TIME_REC(p, ev0);
    // Block sparsity (a row is exactly zero in the columns of gates its circuit never applies): one streaming pass marks,
    // per 16-row panel, the 128-column tiles that hold anything -- fused with the row scaling when there is one -- and
    // the product skips every (panel, tile pair) with an empty side.  Worth the pass from ~16 K rows x 4 tiles on.
    const bool sparse = p->jtj_sparse && n_rows >= 16384 && gst::jtj_mask_tiles((int)n_cols) >= 4 && gst::jtj_mask_tiles((int)n_cols) <= 32;
    const uint32_t* d_pmask = nullptr;
    if (sparse) {
        HIP_TRY(p->d_jtj_pmask.ensure((size_t)gst::jtj_mask_panels(n_rows)));
        HIP_TRY(gst::launch_jtj_panel_masks(d_J, n_rows, (int)n_cols, ld, d_row_scale, p->d_jtj_pmask.p, p->stream));
        d_pmask = p->d_jtj_pmask.p;
    } else if (d_row_scale && n_rows > 0 && n_cols > 0) HIP_TRY(gst::launch_scale_rows(d_J, n_rows, n_cols, ld, d_row_scale, p->stream));
    if (n_cols > 0) {
        const int n_slabs = gst::jtj_num_slabs(n_rows, (int)n_cols);
        HIP_TRY(p->d_jtj_part.ensure((size_t)n_slabs * n_cols * n_cols));
        TIME_REC(p, evk0);
        HIP_TRY(gst::launch_jtj(d_J, n_rows, (int)n_cols, ld, p->d_jtj_part.p, n_slabs, d_jtj, p->stream, d_pmask));
        TIME_REC(p, evk1);
    }
    TIME_REC(p, ev1);
    return GST_OK;
    });
}

int gst_fill_jtf_dev(gst_plan* p, const double* d_J, int64_t n_rows, int64_t n_cols, int64_t ld, const double* d_f,
                     double* d_jtf)
{
    return guarded([&]() -> int {
    if (!p || !d_J || !d_f || !d_jtf || n_rows < 0 || n_cols < 0 || ld < n_cols || n_cols > 0x7fffffff) return fail(GST_EINVAL, "bad argument");
    int rc = ensure_device(p);
    if (rc) return rc;
    if (n_cols == 0) return GST_OK;
    const int n_slabs = (int)std::max<int64_t>(1, std::min<int64_t>(1024, (n_rows + 255) / 256));
    gst::track_touch(d_jtf, (size_t)n_cols * 8);
    HIP_TRY(p->d_jtf_part.ensure((size_t)n_slabs * n_cols));
    HIP_TRY(gst::launch_jtf(d_J, d_f, n_rows, (int)n_cols, ld, p->d_jtf_part.p, n_slabs, d_jtf, p->stream));
    return GST_OK;
    });
}

int gst_objective_rows_dev(gst_plan* p, const gst_objective_desc* d, double* d_probs, const double* d_counts,
                           const double* d_totals, int64_t n, double* d_lsvec, double* d_rowscale, double* d_terms,
                           double* sum_terms)
{
    return guarded([&]() -> int {
    if (!p || !d || !d_probs || !d_counts || !d_totals || !d_lsvec || !d_rowscale || n < 0) return fail(GST_EINVAL, "bad argument");
    if (d->kind != GST_OBJ_CHI2 && d->kind != GST_OBJ_POISSON_DLOGL) return fail(GST_EINVAL, "unknown objective kind");
    if (!(d->min_prob_clip > 0.0) || (d->kind == GST_OBJ_POISSON_DLOGL && !(d->radius > 0.0)))
        return fail(GST_EINVAL, "min_prob_clip and radius must be positive");
    int rc = ensure_device(p);
    if (rc) return rc;
    if (sum_terms) *sum_terms = 0.0;
    if (n == 0) return GST_OK;
    for (double* w : {d_probs, d_lsvec, d_rowscale, d_terms}) gst::track_touch(w, (size_t)n * 8);
    const int n_blocks = (int)std::max<int64_t>(1, std::min<int64_t>(1024, (n + 255) / 256));
    HIP_TRY(p->d_obj_part.ensure((size_t)n_blocks));
    HIP_TRY(gst::launch_objective_rows(d->kind, d_probs, d_counts, d_totals, n, d->min_prob_clip, d->radius, d->prob_clip_lo,
                                       d->prob_clip_hi, d_lsvec, d_rowscale, d_terms, p->d_obj_part.p, n_blocks, p->stream));
    if (sum_terms) {
        std::vector<double> part((size_t)n_blocks);
        HIP_TRY(hipMemcpyAsync(part.data(), p->d_obj_part.p, part.size() * 8, hipMemcpyDeviceToHost, p->stream));
        HIP_TRY(hipStreamSynchronize(p->stream));
        double s = 0.0;
        for (double x : part) s += x;
        *sum_terms = s;
    }
    return GST_OK;
    });
}

int gst_memcpy_h2d(gst_plan* p, void* d_dst, const void* src, int64_t nbytes)
{
    return guarded([&]() -> int {
    if (!p || !d_dst || !src || nbytes < 0) return fail(GST_EINVAL, "bad argument");
    int rc = ensure_device(p);
    if (rc) return rc;
    gst::track_touch(d_dst, (size_t)nbytes);
    HIP_TRY(hipMemcpyAsync(d_dst, src, (size_t)nbytes, hipMemcpyHostToDevice, p->stream));
    HIP_TRY(hipStreamSynchronize(p->stream));
    return GST_OK;
    });
}

int gst_copy_block_dev(gst_plan* p, double* d_dst, int64_t dst_ld, const double* d_src, int64_t src_ld, int64_t n_rows, int64_t n_cols)
{
    return guarded([&]() -> int {
    if (!p || n_rows < 0 || n_cols < 0 || dst_ld < n_cols || src_ld < n_cols) return fail(GST_EINVAL, "bad argument");
    if (n_rows == 0 || n_cols == 0) return GST_OK;
    if (!d_dst || !d_src) return fail(GST_EINVAL, "NULL pointer");
    int rc = ensure_device(p);
    if (rc) return rc;
    gst::track_touch(d_dst, (size_t)((n_rows - 1) * dst_ld + n_cols) * 8);
    HIP_TRY(hipMemcpy2DAsync(d_dst, (size_t)dst_ld * 8, d_src, (size_t)src_ld * 8, (size_t)n_cols * 8, (size_t)n_rows, hipMemcpyDeviceToDevice, p->stream));
    return GST_OK;
    });
}

int gst_device_malloc(gst_plan* p, int64_t nbytes, void** d_ptr)
{
    return guarded([&]() -> int {
    if (!p || !d_ptr || nbytes < 0) return fail(GST_EINVAL, "bad argument");
    int rc = ensure_device(p);
    if (rc) return rc;
    HIP_TRY(hipMalloc(d_ptr, (size_t)std::max<int64_t>(nbytes, 1)));
    gst::track_alloc(*d_ptr, (size_t)std::max<int64_t>(nbytes, 1));
    return GST_OK;
    });
}

int gst_device_free(gst_plan* p, void* d_ptr)
{
    return guarded([&]() -> int {
    if (!p) return fail(GST_EINVAL, "plan is NULL");
    int rc = ensure_device(p);
    if (rc) return rc;
    gst::track_free(d_ptr);
    HIP_TRY(hipFree(d_ptr));
    return GST_OK;
    });
}

int gst_device_touch(gst_plan* p, void* d_ptr, int64_t nbytes)
{
    return guarded([&]() -> int {
    if (!p || nbytes < 0 || (nbytes > 0 && !d_ptr)) return fail(GST_EINVAL, "bad argument");
    gst::track_touch(d_ptr, (size_t)nbytes);
    return GST_OK;
    });
}

int gst_memcpy_d2h(gst_plan* p, void* dst, const void* d_src, int64_t nbytes)
{
    return guarded([&]() -> int {
    if (!p || !dst || !d_src || nbytes < 0) return fail(GST_EINVAL, "bad argument");
    int rc = ensure_device(p);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(p->stream));
    HIP_TRY(hipMemcpy(dst, d_src, (size_t)nbytes, hipMemcpyDeviceToHost));
    return GST_OK;
    });
}

int gst_memcpy_d2h_async(gst_plan* p, void* dst, const void* d_src, int64_t nbytes)
{
    return guarded([&]() -> int {
    if (!p || !dst || !d_src || nbytes < 0) return fail(GST_EINVAL, "bad argument");
    int rc = ensure_device(p);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(dst, d_src, (size_t)nbytes, hipMemcpyDeviceToHost, p->stream));
    return GST_OK;
    });
}

int gst_host_register(void* ptr, int64_t nbytes)
{
    return guarded([&]() -> int {
        if (!ptr || nbytes <= 0) return fail(GST_EINVAL, "bad argument");
        int n = 0;
        hipError_t e = hipGetDeviceCount(&n);
        if (e != hipSuccess || n <= 0) return fail(GST_ENODEVICE, std::string("no HIP device available (") + hipGetErrorString(e) + ")");
        HIP_TRY(hipHostRegister(ptr, (size_t)nbytes, hipHostRegisterPortable | hipHostRegisterMapped));
        {
            std::lock_guard<std::mutex> lock(g_reg_mutex);
            g_registered.emplace_back((char*)ptr, (size_t)nbytes);
        }
        return GST_OK;
    });
}

int gst_host_unregister(void* ptr)
{
    return guarded([&]() -> int {
        if (!ptr) return fail(GST_EINVAL, "bad argument");
        int n = 0;
        hipError_t e = hipGetDeviceCount(&n);
        if (e != hipSuccess || n <= 0) return fail(GST_ENODEVICE, std::string("no HIP device available (") + hipGetErrorString(e) + ")");
        {
            std::lock_guard<std::mutex> lock(g_reg_mutex);
            for (size_t k = 0; k < g_registered.size(); k++)
                if (g_registered[k].first == (char*)ptr) { g_registered.erase(g_registered.begin() + (long)k); break; }
        }
        HIP_TRY(hipHostUnregister(ptr));
        return GST_OK;
    });
}

int gst_sync(gst_plan* p)
{
    return guarded([&]() -> int {
    if (!p) return fail(GST_EINVAL, "plan is NULL");
    if (!p->dev_ready) return GST_OK;
    HIP_TRY(hipSetDevice(p->device));
    HIP_TRY(hipStreamSynchronize(p->stream));
    float ms = 0;
    if (p->timing && hipEventElapsedTime(&ms, p->ev0, p->ev1) == hipSuccess) p->last_total_ms = ms;
    if (p->timing && hipEventElapsedTime(&ms, p->evk0, p->evk1) == hipSuccess) p->last_kernel_ms = ms;
    return GST_OK;
    });
}

int gst_get_stats(const gst_plan* p, gst_stats* s)
{
    return guarded([&]() -> int {
    if (!p || !s) return fail(GST_EINVAL, "NULL argument");
    const gst::HostPlan& h = p->hp;
    s->n_circuits = h.n_circuits; s->n_elements = h.n_elements; s->sum_depth = h.sum_depth;
    s->trie_nodes = h.trie_nodes; s->applies_per_pass = h.applies_per_pass; s->n_tasks = h.n_tasks();
    s->prog_words = (int64_t)h.prog.size(); s->max_slots = h.max_slots; s->max_depth = h.max_depth;
    s->last_kernel_ms = p->last_kernel_ms; s->last_total_ms = p->last_total_ms; s->last_launches = p->last_launches;
    s->last_fd_form = p->last_fd_form; s->last_fd_aborted = 0;
    s->last_levels = p->last_levels ? 1 : 0; s->last_zeros_resident = p->last_zeros_resident ? 1 : 0;
    if (p->last_fd_form >= 1 && p->d_bin_head.p && p->n_bins > 0 && p->dev_ready) {
        uint32_t flag = 0;                  // the abort flag sits behind the queue heads
        HIP_TRY(hipSetDevice(p->device));
        HIP_TRY(hipMemcpyAsync(&flag, p->d_bin_head.p + p->n_bins, 4, hipMemcpyDeviceToHost, p->stream));
        HIP_TRY(hipStreamSynchronize(p->stream));
        s->last_fd_aborted = flag != 0 ? 1 : 0;
    }
    return GST_OK;
    });
}

int gst_get_state_graph(const gst_plan* p, int32_t* node_parent, int32_t* node_sym, int64_t cap_nodes,
                        int32_t* circ_leaf, int64_t cap_circuits, int64_t* n_nodes)
{
    return guarded([&]() -> int {
    if (!p || !n_nodes) return fail(GST_EINVAL, "NULL argument");
    const gst::HostPlan& h = p->hp;
    *n_nodes = h.n_state_ids;
    if (node_parent && node_sym && cap_nodes >= h.n_state_ids) {
        std::memcpy(node_parent, h.node_parent.data(), sizeof(int32_t) * h.n_state_ids);
        std::memcpy(node_sym, h.node_sym.data(), sizeof(int32_t) * h.n_state_ids);
    }
    if (circ_leaf && cap_circuits >= h.n_circuits) std::memcpy(circ_leaf, h.circ_leaf.data(), sizeof(int32_t) * h.n_circuits);
    return GST_OK;
    });
}

int gst_get_program(const gst_plan* p, uint32_t* words, int64_t cap, int64_t* n_words, int64_t* task_off, int64_t cap_tasks)
{
    return guarded([&]() -> int {
    if (!p || !n_words) return fail(GST_EINVAL, "NULL argument");
    const gst::HostPlan& h = p->hp;
    *n_words = (int64_t)h.prog.size();
    if (words && cap > 0) std::memcpy(words, h.prog.data(), sizeof(uint32_t) * std::min<int64_t>(cap, *n_words));
    if (task_off && cap_tasks >= (int64_t)h.task_off.size())
        std::memcpy(task_off, h.task_off.data(), sizeof(int64_t) * h.task_off.size());
    return GST_OK;
    });
}

int gst_get_level_program(const gst_plan* p, int32_t which, int32_t* words, int64_t cap_words, int64_t* n_words, int32_t* ids,
                          int64_t cap_ids, int64_t* n_ids, int64_t* task_off, int64_t cap_tasks, int32_t* node_parent, int32_t* node_sym,
                          int64_t cap_nodes, int64_t* info)
{
    return guarded([&]() -> int {
    if (!p || !n_words || !n_ids || !info) return fail(GST_EINVAL, "NULL argument");
    if (which < 0 || which > 2) return fail(GST_EINVAL, "which: 0 = forward plan, 1 = reversed plan, 2 = forward plan, probability-only");
    // (a const plan, possibly without a device: built for this call only, exactly as ensure_levels / ensure_reverse build them)
    gst::HostPlan R;
    const gst::HostPlan* h = &p->hp;
    if (which == 1) {
        std::string err = gst::build_reverse_plan(p->hp, R, 0, p->hp.D == 16 ? 1 : (p->hp.D == 64 ? 8 : 4));
        if (!err.empty()) return fail(GST_EINVAL, "reversed plan: " + err);
        h = &R;
    }
    gst::LevelProgram L;
    const std::string why = gst::build_level_program(*h, which == 1 ? p->hp.n_effects : 1, L, which == 2 ? &p->hp.circ_leaf : nullptr);
    info[0] = why.empty() ? 1 : 0; info[1] = L.worthwhile ? 1 : 0; info[2] = L.nv; info[3] = L.max_mats; info[4] = L.max_stages;
    info[5] = L.n_stages; info[6] = L.n_tiles; info[7] = L.n_chains; info[8] = L.chain_nodes; info[9] = L.sum_task_depth; info[12] = L.n_produced;
    info[10] = h->n_state_ids; info[11] = h->n_tasks();
    *n_words = (int64_t)L.words.size(); *n_ids = (int64_t)L.ids.size();
    if (!why.empty()) return GST_OK;
    if (words && cap_words >= *n_words) std::memcpy(words, L.words.data(), sizeof(int32_t) * L.words.size());
    if (ids && cap_ids >= *n_ids) std::memcpy(ids, L.ids.data(), sizeof(int32_t) * L.ids.size());
    if (task_off && cap_tasks >= (int64_t)L.task_off.size()) std::memcpy(task_off, L.task_off.data(), sizeof(int64_t) * L.task_off.size());
    if (node_parent && node_sym && cap_nodes >= h->n_state_ids) {
        std::memcpy(node_parent, h->node_parent.data(), sizeof(int32_t) * (size_t)h->n_state_ids);
        std::memcpy(node_sym, h->node_sym.data(), sizeof(int32_t) * (size_t)h->n_state_ids);
    }
    return GST_OK;
    });
}

int gst_get_dirty_programs(const gst_plan* p, uint32_t* words, int64_t cap, int64_t* n_words, int64_t* prog_off, int64_t cap_progs,
                           int32_t* n_classes)
{
    return guarded([&]() -> int {
    if (!p || !n_words || !n_classes) return fail(GST_EINVAL, "NULL argument");
    if (p->hp.n_gates > 64) return fail(GST_EUNSUPPORTED, "dirty programs exist for at most 64 gates");
    gst::DirtyPrograms local;
    const gst::DirtyPrograms* d = &p->dirty;
    if (!p->dirty_ready) { gst::build_dirty_programs(p->hp, local); d = &local; }     // (a const plan: built for this call only)
    *n_words = (int64_t)d->words.size();
    *n_classes = d->n_classes;
    if (words && cap > 0) std::memcpy(words, d->words.data(), sizeof(uint32_t) * std::min<int64_t>(cap, *n_words));
    if (prog_off && cap_progs >= (int64_t)d->off.size()) std::memcpy(prog_off, d->off.data(), sizeof(int64_t) * d->off.size());
    return GST_OK;
    });
}

int gst_set_option(gst_plan* p, int32_t option, int64_t value)
{
    return guarded([&]() -> int {
    if (!p) return fail(GST_EINVAL, "plan is NULL");
    switch (option) {
    case GST_OPT_ANALYTIC_KEEP_ZEROS:
        if (value < 0 || value > 2) return fail(GST_EINVAL, "GST_OPT_ANALYTIC_KEEP_ZEROS takes 0, 1 or 2");
        p->ana_keep_zeros = (int)value;
        p->ana_zero_valid = false;            // a promise starts now: the next fill writes every zero
        return GST_OK;
    case GST_OPT_FAST_CHAINS:
        if (value < 0 || value > 2) return fail(GST_EINVAL, "GST_OPT_FAST_CHAINS takes 0, 1 or 2");
        p->fast_chains = (int)value;
        return GST_OK;
    case GST_OPT_FAST_PROBS:
        p->fast_probs = value != 0;
        return GST_OK;
    default:
        return fail(GST_EINVAL, "unknown option " + std::to_string(option));
    }
    });
}

int gst_set_lindblad(gst_plan* p, int32_t n_params, int32_t n_members, const gst_lindblad_member* members, int64_t n_terms,
                     const double* term_re, const double* term_im)
{
    return guarded([&]() -> int {
    if (!p) return fail(GST_EINVAL, "plan is NULL");
    gst_plan::Lindblad& L = p->lb;
    // (the request tables of the last fill -- wave tables, set_param order, POVM columns, dirty items -- are derived from
    //  the members' kinds, objects and parameter ranges: a new description, or none, invalidates them)
    if (n_members == 0) { L = gst_plan::Lindblad(); p->cached_kind = 0; return GST_OK; }
    const int D = p->hp.D;
    if (D != 4 && D != 16) return fail(GST_EUNSUPPORTED, "Lindblad members are built on the device for D = 4 and 16");
    if (n_members < 0 || n_params < 0 || !members || n_terms <= 0 || !term_re || !term_im) return fail(GST_EINVAL, "bad argument");
    gst_plan::Lindblad N;
    N.n_params = n_params; N.n_members = n_members;
    const int nb_max = D - 1;                      // Pauli basis of n qubits without the identity: 4^n - 1 = D - 1
    std::vector<uint8_t> gate_seen((size_t)p->hp.n_gates, 0), rho_seen((size_t)p->hp.n_rhos, 0), eff_seen((size_t)p->hp.n_effects, 0);
    for (int32_t m = 0; m < n_members; m++) {
        const gst_lindblad_member& M = members[m];
        const std::string who = "member " + std::to_string(m) + ": ";
        if (!M.static_part) return fail(GST_EINVAL, who + "static_part is NULL");
        if (M.n_blocks < 1 || M.n_blocks > gst::LB_MAX_BLOCKS) return fail(GST_EINVAL, who + "1.." + std::to_string(gst::LB_MAX_BLOCKS) + " coefficient blocks");
        int64_t np = 0, nc = 0;
        for (int b = 0; b < M.n_blocks; b++) {
            const int bt = M.block_type[b], md = M.block_mode[b], n = M.block_n[b];
            if (bt < 0 || bt > 2 || md < 0 || md > 1 || n < 1 || n > nb_max) return fail(GST_EINVAL, who + "bad coefficient block");
            np += bt == 2 ? (int64_t)n * n : n; nc += bt == 2 ? (int64_t)n * n : n;
        }
        if (np > gst::lb_max_coeffs(D)) return fail(GST_EUNSUPPORTED, who + "too many parameters for one member");
        if (M.param0 < 0 || M.param0 + np > n_params) return fail(GST_EINVAL, who + "parameter range outside the model's");
        if (M.term_offset < 0 || M.term_offset + nc > n_terms) return fail(GST_EINVAL, who + "term range outside the term table");
        size_t n_static = 0;
        if (M.kind == GST_KIND_GATE) {
            if (M.obj < 0 || M.obj >= p->hp.n_gates || gate_seen[(size_t)M.obj]++) return fail(GST_EINVAL, who + "bad or repeated gate index");
            n_static = (size_t)D * D;
        } else if (M.kind == GST_KIND_RHO) {
            if (M.obj < 0 || M.obj >= p->hp.n_rhos || rho_seen[(size_t)M.obj]++) return fail(GST_EINVAL, who + "bad or repeated state index");
            n_static = (size_t)D;
        } else if (M.kind == GST_KIND_EFFECT) {
            if (M.n_eff < 1 || M.obj < 0 || M.obj + M.n_eff > p->hp.n_effects) return fail(GST_EINVAL, who + "bad effect range");
            // (a perturbed member occupies one D*D slot of the per-column member sets: n_eff * D doubles must fit)
            if (M.n_eff > D) return fail(GST_EUNSUPPORTED, who + "a POVM member with more than D effects (over-complete POVM) is not built on the device");
            for (int e = M.obj; e < M.obj + M.n_eff; e++) if (eff_seen[(size_t)e]++) return fail(GST_EINVAL, who + "effect listed twice");
            n_static = (size_t)M.n_eff * D;
        } else return fail(GST_EINVAL, who + "unknown kind");
        N.kind.push_back(M.kind); N.obj.push_back(M.obj); N.n_eff.push_back(M.kind == GST_KIND_EFFECT ? M.n_eff : 1);
        N.n_par.push_back((int32_t)np); N.n_blocks.push_back(M.n_blocks);
        N.param0.push_back(M.param0); N.term_off.push_back(M.term_offset); N.static_off.push_back((int64_t)N.statics.size());
        N.statics.insert(N.statics.end(), M.static_part, M.static_part + n_static);
    }
    // a parameter belongs to exactly ONE member: the build / derivative kernels and the column tables step the member whose
    // range holds the column's parameter, so members that share an error generator (same gpindices for two gates) would
    // have only one of them stepped, where the reference's set_parameter_value moves both -- refuse, the caller then
    // takes the host-stepped dense-model route (gst_fill_dprobs_models)
    {
        std::vector<std::pair<int64_t, int64_t>> rng;
        for (int32_t m = 0; m < n_members; m++) if (N.n_par[(size_t)m] > 0) rng.emplace_back(N.param0[(size_t)m], N.param0[(size_t)m] + N.n_par[(size_t)m]);
        std::sort(rng.begin(), rng.end());
        for (size_t k = 1; k < rng.size(); k++)
            if (rng[k].first < rng[k - 1].second)
                return fail(GST_EUNSUPPORTED, "two Lindblad members share parameters (overlapping parameter ranges): not built on the device");
    }
    // every object of the plan must belong to a member: the device builds the WHOLE model
    for (uint8_t v : gate_seen) if (!v) return fail(GST_EINVAL, "a gate of the plan belongs to no Lindblad member");
    for (uint8_t v : rho_seen) if (!v) return fail(GST_EINVAL, "a state preparation of the plan belongs to no Lindblad member");
    for (uint8_t v : eff_seen) if (!v) return fail(GST_EINVAL, "an effect of the plan belongs to no Lindblad member");
    N.blk_type.assign((size_t)n_members * gst::LB_MAX_BLOCKS, 0); N.blk_mode = N.blk_type; N.blk_n = N.blk_type;
    for (int32_t m = 0; m < n_members; m++)
        for (int b = 0; b < members[m].n_blocks; b++) {
            N.blk_type[(size_t)m * gst::LB_MAX_BLOCKS + b] = members[m].block_type[b];
            N.blk_mode[(size_t)m * gst::LB_MAX_BLOCKS + b] = members[m].block_mode[b];
            N.blk_n[(size_t)m * gst::LB_MAX_BLOCKS + b] = members[m].block_n[b];
        }
    N.term_re.assign(term_re, term_re + (size_t)n_terms * D * D);
    N.term_im.assign(term_im, term_im + (size_t)n_terms * D * D);
    N.set = true;
    L = std::move(N);
    p->cached_kind = 0;
    return GST_OK;
    });
}

int gst_set_lindblad_params(gst_plan* p, const double* theta)
{
    return guarded([&]() -> int {
    if (!p || !theta) return fail(GST_EINVAL, "NULL argument");
    if (!p->lb.set) return fail(GST_ESTATE, "gst_set_lindblad has not been called");
    int rc = ensure_device(p);
    if (rc) return rc;
    if ((rc = lb_upload(p))) return rc;
    gst_plan::Lindblad& L = p->lb;
    L.theta.assign(theta, theta + L.n_params);
    HIP_TRY(hipMemcpyAsync(p->d_lb_theta.p, L.theta.data(), (size_t)L.n_params * 8, hipMemcpyHostToDevice, p->stream));
    gst::LbArgs a;
    lb_args(p, a);
    a.set_param = nullptr; a.sets = p->d_lb_base.p; a.gates_rowmajor = p->d_lb_gates_rm.p; a.eps = 0.0;
    HIP_TRY(gst::launch_lindblad_build(p->hp.D, a, L.n_members, p->stream));
    // the base model also becomes the plan's model (what gst_set_model would have been given): 13 KB back over PCIe
    const int D = p->hp.D;
    const size_t ng = (size_t)p->hp.n_gates * D * D, nr = (size_t)p->hp.n_rhos * D, ne = (size_t)p->hp.n_effects * D;
    std::vector<double> set(ng + nr + ne);
    p->h_gates.resize(ng);
    HIP_TRY(hipMemcpyAsync(set.data(), p->d_lb_base.p, set.size() * 8, hipMemcpyDeviceToHost, p->stream));
    if (ng) HIP_TRY(hipMemcpyAsync(p->h_gates.data(), p->d_lb_gates_rm.p, ng * 8, hipMemcpyDeviceToHost, p->stream));
    HIP_TRY(hipStreamSynchronize(p->stream));
    p->h_gates_t.assign(set.begin(), set.begin() + (long)ng);
    p->h_rhos.assign(set.begin() + (long)ng, set.begin() + (long)(ng + nr));
    p->h_effects.assign(set.begin() + (long)(ng + nr), set.end());
    p->have_model = true;
    p->model_dirty = true;
    L.have_theta = true;
    return GST_OK;
    });
}

int gst_get_model(gst_plan* p, double* gates, double* rhos, double* effects)
{
    return guarded([&]() -> int {
    if (!p) return fail(GST_EINVAL, "plan is NULL");
    if (!p->have_model) return fail(GST_ESTATE, "no model has been set");
    if (gates) std::memcpy(gates, p->h_gates.data(), p->h_gates.size() * 8);
    if (rhos) std::memcpy(rhos, p->h_rhos.data(), p->h_rhos.size() * 8);
    if (effects) std::memcpy(effects, p->h_effects.data(), p->h_effects.size() * 8);
    return GST_OK;
    });
}

int gst_get_lindblad_model_sets(gst_plan* p, const int64_t* param_idx, int64_t n_param, double eps, double* gates, double* rhos, double* effects)
{
    return guarded([&]() -> int {
    if (!p || n_param < 0 || (n_param > 0 && !param_idx)) return fail(GST_EINVAL, "bad argument");
    if (!p->lb.set || !p->lb.have_theta) return fail(GST_ESTATE, "gst_set_lindblad / gst_set_lindblad_params have not been called");
    int rc = ensure_device(p);
    if (rc) return rc;
    for (int64_t c = 0; c < n_param; c++)
        if (param_idx[c] < 0 || param_idx[c] >= p->lb.n_params) return fail(GST_EINVAL, "parameter index out of range");
    if (n_param == 0) return GST_OK;
    const int D = p->hp.D;
    const size_t ng = (size_t)p->hp.n_gates * D * D, nr = (size_t)p->hp.n_rhos * D, ne = (size_t)p->hp.n_effects * D, stride = ng + nr + ne;
    HIP_TRY(p->d_mm_models.ensure((size_t)n_param * stride));
    HIP_TRY(p->d_lb_setparam.ensure((size_t)n_param));
    HIP_TRY(hipMemcpyAsync(p->d_lb_setparam.p, param_idx, (size_t)n_param * 8, hipMemcpyHostToDevice, p->stream));
    gst::LbArgs a;
    lb_args(p, a);
    a.set_param = p->d_lb_setparam.p; a.sets = p->d_mm_models.p; a.eps = eps;
    HIP_TRY(gst::launch_lindblad_build(D, a, n_param, p->stream));
    std::vector<double> h((size_t)n_param * stride);
    HIP_TRY(hipMemcpyAsync(h.data(), p->d_mm_models.p, h.size() * 8, hipMemcpyDeviceToHost, p->stream));
    HIP_TRY(hipStreamSynchronize(p->stream));
    for (int64_t m = 0; m < n_param; m++) {
        const double* s = h.data() + (size_t)m * stride;
        if (gates)
            for (int g = 0; g < p->hp.n_gates; g++)
                for (int i = 0; i < D; i++)
                    for (int j = 0; j < D; j++) gates[(size_t)m * ng + ((size_t)g * D + i) * D + j] = s[((size_t)g * D + j) * D + i];
        if (rhos) std::memcpy(rhos + (size_t)m * nr, s + ng, nr * 8);
        if (effects) std::memcpy(effects + (size_t)m * ne, s + ng + nr, ne * 8);
    }
    return GST_OK;
    });
}

int gst_sort_circuits(int64_t n_circuits, const int64_t* circ_ptr, const int32_t* circ_syms, const int32_t* circ_head,
                      int64_t* order_out, int64_t* lcp_out)
{
    return guarded([&]() -> int {
    if (n_circuits < 0 || (n_circuits > 0 && (!circ_ptr || !order_out || !lcp_out))) return fail(GST_EINVAL, "bad argument");
    for (int64_t c = 0; c < n_circuits; c++)
        if (circ_ptr[c + 1] < circ_ptr[c]) return fail(GST_EINVAL, "circ_ptr must be non-decreasing");
    if (n_circuits > 0 && circ_ptr[n_circuits] > circ_ptr[0] && !circ_syms) return fail(GST_EINVAL, "circ_syms is NULL");
    // key of circuit c: (head[c], syms[ptr[c]] ... syms[ptr[c+1]-1]) compared element by element, a proper prefix first
    auto common = [&](int64_t x, int64_t y) -> int64_t {          // equal leading key elements
        if (circ_head && circ_head[x] != circ_head[y]) return 0;
        const int32_t* a = circ_syms + circ_ptr[x];
        const int32_t* b = circ_syms + circ_ptr[y];
        const int64_t la = circ_ptr[x + 1] - circ_ptr[x], lb = circ_ptr[y + 1] - circ_ptr[y], m = std::min(la, lb);
        int64_t j = 0;
        while (j < m && a[j] == b[j]) j++;
        return 1 + j;
    };
    auto less = [&](int64_t x, int64_t y) -> bool {
        if (circ_head && circ_head[x] != circ_head[y]) return circ_head[x] < circ_head[y];
        const int32_t* a = circ_syms + circ_ptr[x];
        const int32_t* b = circ_syms + circ_ptr[y];
        const int64_t la = circ_ptr[x + 1] - circ_ptr[x], lb = circ_ptr[y + 1] - circ_ptr[y], m = std::min(la, lb);
        for (int64_t j = 0; j < m; j++)
            if (a[j] != b[j]) return a[j] < b[j];
        return la < lb;
    };
    for (int64_t c = 0; c < n_circuits; c++) order_out[c] = c;
    std::stable_sort(order_out, order_out + n_circuits, less);
    for (int64_t k = 0; k < n_circuits; k++) lcp_out[k] = k ? common(order_out[k - 1], order_out[k]) : 0;
    return GST_OK;
    });
}

int gst_circuit_first_use(int64_t n_circuits, const int64_t* circ_ptr, const int32_t* circ_syms, int32_t n_syms, int64_t* first_out)
{
    return guarded([&]() -> int {
    if (n_circuits < 0 || n_syms < 0 || (n_circuits > 0 && (!circ_ptr || (n_syms > 0 && !first_out)))) return fail(GST_EINVAL, "bad argument");
    for (int64_t c = 0; c < n_circuits; c++) {
        int64_t* f = first_out + c * n_syms;
        for (int32_t g = 0; g < n_syms; g++) f[g] = -1;
        int32_t found = 0;
        for (int64_t k = circ_ptr[c]; k < circ_ptr[c + 1] && found < n_syms; k++) {
            const int32_t g = circ_syms[k];
            if (g < 0 || g >= n_syms) return fail(GST_EINVAL, "symbol out of range in circuit " + std::to_string(c));
            if (f[g] < 0) { f[g] = k - circ_ptr[c]; found++; }
        }
    }
    return GST_OK;
    });
}

int gst_get_fd_queues(gst_plan* p, const int64_t* param_idx, int64_t n_param, int32_t n_queues, int32_t handover,
                      int64_t* load_out, int32_t* n_pairs, int32_t* n_handovers)
{
    return guarded([&]() -> int {
    if (!p || !load_out || n_queues <= 0) return fail(GST_EINVAL, "bad argument");
    if (p->hp.D == 64) return fail(GST_EUNSUPPORTED, "per-SIMD queues exist for D <= 16");
    int rc = check_params(p, param_idx, n_param);
    if (rc) return rc;
    LaneLayout L;
    pack_lanes(p, param_idx, nullptr, n_param, L, false);
    if (p->task_cost.empty()) gst::task_gate_costs(p->hp, p->task_cost);
    if (p->task_cost.empty()) return fail(GST_EUNSUPPORTED, "no work table (more than 64 gates)");
    std::vector<std::pair<int32_t, uint32_t>> items;
    int32_t n_units = 0;
    fd_items(p, L, false, items, n_units);
    std::vector<int32_t> cptr, cpc;
    std::vector<float> cfr;
    std::vector<uint32_t> clive;
    if (handover != 0) gst::task_split_candidates(p->hp, cptr, cpc, cfr, 1 << 20, nullptr);
    gst::FdQueues Q;
    gst::pack_fd_queues(items, n_units, p->hp.n_tasks(), n_queues, handover, cptr, cpc, cfr, clive, Q);
    for (int32_t b = 0; b < n_queues; b++) load_out[b] = Q.load[(size_t)b];
    if (n_pairs) *n_pairs = (int32_t)items.size();
    if (n_handovers) *n_handovers = Q.n_split;
    return GST_OK;
    });
}

}  // extern "C"
