// gst_abi.cpp -- the C ABI of include/gstfwd.h on top of the plan compiler and the HIP kernels.
//
// Host-side counterpart of what mapfill_probs_atom / mapfill_dprobs_atom do around the hot loop in
// the reference (mapforwardsim_calc_densitymx.pyx:149-190, 290-383): marshal the plan once (the
// reference re-converts it on EVERY call, :170-181), upload the small model arrays per call, launch.
// The drivers live in gst_fill_*.cpp / gst_hessian.cpp / gst_lindblad_abi.cpp / gst_normal_abi.cpp (gst_state.hpp).
// There is no CPU compute path in this library: without a usable HIP device every fill fails loudly.
#include "gst_state.hpp"

#include <csignal>
#include <cstring>
#include <execinfo.h>
#include <unistd.h>

using namespace gst_impl;

int gst_impl::g_poison_fill = 0;

namespace {
thread_local std::string g_err;

// Diagnostics (GST_ABORT_BACKTRACE, set by tests/conftest.py): a process that dies through abort() -- glibc's heap checks, the
// HIP runtime's fatal paths -- prints the native call stack first; Python's faulthandler then adds its own.  glibc itself
// writes its message to the terminal, not to stderr, unless LIBC_FATAL_STDERR_ is set (conftest.py sets that too).
struct sigaction g_prev_abrt;
int g_abort_fd = 2;                 // GST_ABORT_BACKTRACE = 1: stderr; = "n:pid", n > 2: that descriptor (a test runner that captures fd 2
                                    // hands over a duplicate of the real one: what is written to a captured fd dies with the process)
void on_abort(int sig)
{
    static const char head[] = "gstfwd: SIGABRT -- native stack:\n";
    (void)!write(g_abort_fd, head, sizeof(head) - 1);
    void* bt[64];
    const int n = backtrace(bt, 64);
    backtrace_symbols_fd(bt, n, g_abort_fd);
    sigaction(SIGABRT, &g_prev_abrt, nullptr);          // hand over to whoever was there before (faulthandler / default)
    raise(sig);
}
struct AbortBacktrace {
    AbortBacktrace()
    {
        const char* e = std::getenv("GST_ABORT_BACKTRACE");
        if (!e || std::atoi(e) == 0) return;
        // "fd:pid": the descriptor is meaningful in the process that duplicated it only (a spawned child inherits the variable,
        // not the descriptor -- or worse, has the number in use for something else)
        const char* colon = std::strchr(e, ':');
        if (std::atoi(e) > 2 && colon && std::atol(colon + 1) == (long)getpid()) g_abort_fd = std::atoi(e);
        struct sigaction sa;
        std::memset(&sa, 0, sizeof(sa));
        sa.sa_handler = on_abort;
        sigemptyset(&sa.sa_mask);
        sigaction(SIGABRT, &sa, &g_prev_abrt);
    }
} g_abort_backtrace;
}  // namespace

namespace gst {
int set_error(int code, const std::string& msg)
{
    g_err = msg;
    return code;
}
int plan_ensure_device(gst_plan* plan) { return plan ? ensure_device(plan) : fail(GST_EINVAL, "plan is NULL"); }
hipStream_t plan_stream(const gst_plan* plan) { return plan->stream; }
int plan_device(const gst_plan* plan) { return plan->device; }
}  // namespace gst

namespace {
// Host regions the caller page-locked through gst_host_register (mapped into the device's address space): fills whose
// destination lies inside one write their results straight into it.
std::mutex g_reg_mutex;
std::vector<std::pair<char*, size_t>> g_registered;
}  // namespace

namespace gst_impl {

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


// ---- copies between device memory and the caller's host memory -----------------------------------------------------------
namespace {
constexpr size_t STAGE_BYTES = (size_t)8 << 20;
int ensure_stage(gst_plan* p, size_t at_least)
{
    const size_t want = std::max(STAGE_BYTES, at_least);
    if (p->h_stage && p->h_stage_bytes >= want) return GST_OK;
    if (p->h_stage) { HIP_TRY(hipStreamSynchronize(p->stream)); (void)hipHostFree(p->h_stage); p->h_stage = nullptr; p->h_stage_bytes = 0; }
    HIP_TRY(hipHostMalloc((void**)&p->h_stage, want, hipHostMallocDefault));
    p->h_stage_bytes = want;
    return GST_OK;
}
}  // namespace

int d2h_bytes(gst_plan* p, void* dst, const void* d_src, size_t bytes)
{
    if (bytes == 0) return GST_OK;
    if (mapped_device_pointer(dst, bytes)) {          // page-locked by the caller: straight DMA, asynchronous
        HIP_TRY(hipMemcpyAsync(dst, d_src, bytes, hipMemcpyDeviceToHost, p->stream));
        return GST_OK;
    }
    int rc = ensure_stage(p, 0);
    if (rc) return rc;
    for (size_t off = 0; off < bytes; off += p->h_stage_bytes) {
        const size_t n = std::min(p->h_stage_bytes, bytes - off);
        HIP_TRY(hipMemcpyAsync(p->h_stage, (const char*)d_src + off, n, hipMemcpyDeviceToHost, p->stream));
        HIP_TRY(hipStreamSynchronize(p->stream));
        std::memcpy((char*)dst + off, p->h_stage, n);
    }
    return GST_OK;
}

int h2d_bytes(gst_plan* p, void* d_dst, const void* src, size_t bytes)
{
    if (bytes == 0) return GST_OK;
    if (mapped_device_pointer(src, bytes)) {
        HIP_TRY(hipMemcpyAsync(d_dst, src, bytes, hipMemcpyHostToDevice, p->stream));
        return GST_OK;
    }
    int rc = ensure_stage(p, 0);
    if (rc) return rc;
    for (size_t off = 0; off < bytes; off += p->h_stage_bytes) {
        const size_t n = std::min(p->h_stage_bytes, bytes - off);
        HIP_TRY(hipStreamSynchronize(p->stream));                 // (the previous chunk has left the staging buffer)
        std::memcpy(p->h_stage, (const char*)src + off, n);
        HIP_TRY(hipMemcpyAsync((char*)d_dst + off, p->h_stage, n, hipMemcpyHostToDevice, p->stream));
    }
    HIP_TRY(hipStreamSynchronize(p->stream));
    return GST_OK;
}

int h2d_async(gst_plan* p, void* d_dst, const void* src, size_t bytes)
{
    constexpr size_t RING = (size_t)4 << 20;
    if (bytes == 0) return GST_OK;
    if (bytes > RING / 2 || mapped_device_pointer(src, bytes)) return h2d_bytes(p, d_dst, src, bytes);
    if (!p->h_up) {
        HIP_TRY(hipHostMalloc((void**)&p->h_up, RING, hipHostMallocDefault));
        p->h_up_bytes = RING; p->h_up_at = 0;
    }
    size_t at = (p->h_up_at + 63) & ~(size_t)63;
    if (at + bytes > p->h_up_bytes) {                 // wrap: every copy out of the ring so far was issued on p->stream
        HIP_TRY(hipStreamSynchronize(p->stream));
        at = 0;
    }
    std::memcpy(p->h_up + at, src, bytes);
    HIP_TRY(hipMemcpyAsync(d_dst, p->h_up + at, bytes, hipMemcpyHostToDevice, p->stream));
    p->h_up_at = at + bytes;
    return GST_OK;
}

// rows of n_cols doubles: device [n_rows][src_ld] -> host [n_rows][dst_ld]
int d2h_rows(gst_plan* p, double* dst, int64_t dst_ld, const double* d_src, int64_t src_ld, int64_t n_rows, int64_t n_cols)
{
    if (n_rows <= 0 || n_cols <= 0) return GST_OK;
    if (mapped_device_pointer(dst, (size_t)((n_rows - 1) * dst_ld + n_cols) * 8)) {
        HIP_TRY(hipMemcpy2DAsync(dst, (size_t)dst_ld * 8, d_src, (size_t)src_ld * 8, (size_t)n_cols * 8, (size_t)n_rows,
                                 hipMemcpyDeviceToHost, p->stream));
        return GST_OK;
    }
    if (dst_ld == n_cols && src_ld == n_cols) return d2h_bytes(p, dst, d_src, (size_t)n_rows * n_cols * 8);
    int rc = ensure_stage(p, (size_t)n_cols * 8);
    if (rc) return rc;
    const int64_t chunk = std::max<int64_t>(1, (int64_t)(p->h_stage_bytes / ((size_t)n_cols * 8)));
    for (int64_t r0 = 0; r0 < n_rows; r0 += chunk) {
        const int64_t nr = std::min(chunk, n_rows - r0);
        HIP_TRY(hipMemcpy2DAsync(p->h_stage, (size_t)n_cols * 8, d_src + r0 * src_ld, (size_t)src_ld * 8, (size_t)n_cols * 8, (size_t)nr,
                                 hipMemcpyDeviceToHost, p->stream));
        HIP_TRY(hipStreamSynchronize(p->stream));
        const double* st = (const double*)p->h_stage;
        for (int64_t r = 0; r < nr; r++) std::memcpy(dst + (r0 + r) * dst_ld, st + r * n_cols, (size_t)n_cols * 8);
    }
    return GST_OK;
}

int finish_create(gst_plan* p, const gst_options* opt, gst_plan** out)
{
    static std::atomic<uint64_t> next_uid{1};
    p->uid = next_uid.fetch_add(1);
    // default slot budget: what fits in LDS at 4 wavefronts per SIMD (16 per CU): one 8 KB slot at D=16
    int32_t max_slots = opt ? opt->max_slots : 0;
    if (p->hp.D != 4 && p->hp.D != 16 && p->hp.D != 64) {
        // Any other dimension up to 64 (statecreps.cpp:20-58 is dimension-generic: a qutrit in the Gell-Mann basis has 9, a
        // qubit with a leakage level as well) runs zero-padded at the next supported one.
        const int D = p->hp.D;
        if (D < 2 || D > 64) {
            delete p;
            return fail(GST_EUNSUPPORTED, "state dimension " + std::to_string(D) + " not supported (2 .. 64)");
        }
        p->D_user = D;
        p->hp.D = D <= 4 ? 4 : D <= 16 ? 16 : 64;
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
            else if (key == "tiles") p->ana_tiles = iv != 0;
            else if (key == "tile_dbg") p->tile_dbg = iv;
            else if (key == "lpt") p->ana_lpt = iv != 0;
            else if (key == "chain_resident") p->chain_resident = iv != 0;
            else if (key == "lm_graph") p->lm_graph_enabled = iv != 0;
            else if (key == "comm_self") {}          // (read by gst_comm_create: gst_comm.cpp)
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
    H2D_TRY(p, p->d_prog.p, h.prog.data(), h.prog.size() * 4);
    H2D_TRY(p, p->d_task_off.p, h.task_off.data(), h.task_off.size() * 8);
    H2D_TRY(p, p->d_eff_ptr.p, h.eff_ptr.data(), h.eff_ptr.size() * 4);
    if (!h.eff_label.empty()) {
        H2D_TRY(p, p->d_eff_label.p, h.eff_label.data(), h.eff_label.size() * 4);
        H2D_TRY(p, p->d_eff_dest.p, h.eff_dest.data(), h.eff_dest.size() * 4);
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

int upload_i32(gst_plan* p, DevBuf<int32_t>& b, const std::vector<int32_t>& v)
{
    HIP_TRY(b.ensure(v.size()));
    if (!v.empty()) H2D_TRY(p, b.p, v.data(), v.size() * 4);
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
int stage_out(gst_plan* p, size_t count, bool keeps_claims)
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

// D2H of the dense staging Jacobian p->d_out [nE][n_param] into the caller's (ld, dest_idx) window, then end_call.  A
// contiguous destination window -- the `dest_param_slice` of the reference's seam (mapforwardsim.py:379-383) -- is one
// strided 2-D copy; only a scattered dest_idx needs host staging.
int copy_out_dprobs(gst_plan* p, double* out, int64_t ld, const int64_t* dest_idx, int64_t n_param, double* probs_out)
{
    const int64_t nE = p->hp.n_elements;
    bool window = true;
    for (int64_t c = 1; dest_idx && c < n_param; c++) window = window && dest_idx[c] == dest_idx[0] + c;
    const int64_t d0 = (dest_idx && n_param > 0) ? dest_idx[0] : 0;
    int rc;
    if (n_param > 0 && nE > 0) {
        if (window) {
            if ((rc = d2h_rows(p, out + d0, ld, p->d_out.p, n_param, nE, n_param))) return rc;
        } else {
            // scattered columns: dense rows through the staging buffer, scattered on the host
            if ((rc = ensure_stage(p, (size_t)n_param * 8))) return rc;
            const int64_t chunk = std::max<int64_t>(1, (int64_t)(p->h_stage_bytes / ((size_t)n_param * 8)));
            for (int64_t r0 = 0; r0 < nE; r0 += chunk) {
                const int64_t nr = std::min(chunk, nE - r0);
                HIP_TRY(hipMemcpyAsync(p->h_stage, p->d_out.p + r0 * n_param, (size_t)nr * n_param * 8, hipMemcpyDeviceToHost, p->stream));
                HIP_TRY(hipStreamSynchronize(p->stream));
                const double* st = (const double*)p->h_stage;
                for (int64_t k = 0; k < nr; k++)
                    for (int64_t c = 0; c < n_param; c++) out[(r0 + k) * ld + dest_idx[c]] = st[k * n_param + c];
            }
        }
    }
    if (probs_out && (rc = d2h_bytes(p, probs_out, p->d_pbase.p, (size_t)nE * 8))) return rc;
    return end_call(p, true);
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

}  // namespace gst_impl

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
    const int D = p->hp.D, Du = p->user_D();
    const size_t ng = (size_t)p->hp.n_gates * D * D;
    p->h_gates.assign(ng, 0.0);
    p->h_gates_t.assign(ng, 0.0);
    for (int g = 0; g < p->hp.n_gates; g++)
        for (int i = 0; i < Du; i++)
            for (int j = 0; j < Du; j++) {
                const double x = gates[((size_t)g * Du + i) * Du + j];
                p->h_gates[((size_t)g * D + i) * D + j] = x;
                p->h_gates_t[((size_t)g * D + j) * D + i] = x;
            }
    p->h_rhos.assign((size_t)p->hp.n_rhos * D, 0.0);
    p->h_effects.assign((size_t)p->hp.n_effects * D, 0.0);
    for (int r = 0; r < p->hp.n_rhos; r++) std::memcpy(&p->h_rhos[(size_t)r * D], rhos + (size_t)r * Du, (size_t)Du * 8);
    for (int e = 0; e < p->hp.n_effects; e++) std::memcpy(&p->h_effects[(size_t)e * D], effects + (size_t)e * Du, (size_t)Du * 8);
    p->have_model = true;
    p->model_dirty = true;
    return GST_OK;
    });
}

int gst_set_param_map(gst_plan* p, int32_t n_params, const int32_t* kind, const int32_t* obj, const int32_t* elem)
{
    return guarded([&]() -> int {
    if (!p || n_params < 0 || (n_params > 0 && (!kind || !obj || !elem))) return fail(GST_EINVAL, "bad argument");
    const int D = p->hp.D, Du = p->user_D();
    for (int32_t i = 0; i < n_params; i++) {
        const int k = kind[i];
        if (k == GST_KIND_NONE) continue;
        const int nobj = k == GST_KIND_GATE ? p->hp.n_gates : k == GST_KIND_RHO ? p->hp.n_rhos : k == GST_KIND_EFFECT ? p->hp.n_effects : -1;
        const int nel = k == GST_KIND_GATE ? Du * Du : Du;
        if (nobj < 0 || obj[i] < 0 || obj[i] >= nobj || elem[i] < 0 || elem[i] >= nel)
            return fail(GST_EINVAL, "parameter map entry " + std::to_string(i) + " out of range");
    }
    p->pkind.assign(kind, kind + n_params);
    p->pobj.assign(obj, obj + n_params);
    p->pelem.assign(elem, elem + n_params);
    if (Du != D)                                   // a gate element (i, j) of the caller's Du x Du matrix sits at i * D + j here
        for (int32_t i = 0; i < n_params; i++)
            if (kind[i] == GST_KIND_GATE) p->pelem[(size_t)i] = (elem[i] / Du) * D + elem[i] % Du;
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
    p->comp_identity.assign((size_t)p->hp.D, 0.0);
    std::memcpy(p->comp_identity.data(), identity, (size_t)p->user_D() * 8);
    return GST_OK;
    });
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
    if ((rc = d2h_bytes(p, out, p->d_pbase.p, (size_t)p->hp.n_elements * 8))) return rc;
    return end_call(p, true);
    });
}

// FD over device-built Lindblad members exists only where it meets the 1e-8 bar (include/gstfwd.h, gst_set_lindblad)
static int lindblad_fd_gate(const gst_plan* p, int mode)
{
    if (mode != GST_DERIV_FD || !p->lb.set || p->hp.max_depth <= GST_LINDBLAD_FD_MAX_DEPTH) return GST_OK;
    return fail(GST_EUNSUPPORTED, "finite differences over device-built Lindblad members are offered for circuits of depth <= " +
                std::to_string(GST_LINDBLAD_FD_MAX_DEPTH) + " only (this plan: " + std::to_string(p->hp.max_depth) +
                "): beyond, the quotient amplifies last-bit differences of the exponential past 1e-8 -- use GST_DERIV_ANALYTIC, or "
                "gst_fill_dprobs_models over the host's own perturbed members");
}

int gst_fill_dprobs_dev(gst_plan* p, double* d_out, int64_t ld, const int64_t* param_idx, const int64_t* dest_idx,
                        int64_t n_param, int mode, double eps, double* d_probs_out)
{
    return guarded([&]() -> int {
    int rc = begin_call(p);
    if (rc) return rc;
    if (mode != GST_DERIV_FD && mode != GST_DERIV_ANALYTIC) return fail(GST_EINVAL, "unknown derivative mode");
    if (!d_out && n_param > 0) return fail(GST_EINVAL, "d_out is NULL");
    if ((rc = lindblad_fd_gate(p, mode))) return rc;
    // what this call overwrites no longer holds an earlier exact Jacobian's zeros (the plain exact fill keeps its own books)
    if (d_probs_out) gst::track_touch(d_probs_out, (size_t)p->hp.n_elements * 8);
    if (n_param > 0 && (mode == GST_DERIV_FD || p->lb.set || p->cmp.set || p->derivs_set)) gst::track_touch(d_out, jac_extent(p->hp.n_elements, ld, dest_idx, n_param));
    if (p->cmp.set && (mode == GST_DERIV_FD || !p->derivs_set)) {        // implicit models: device-built layers (gst_set_composite)
        if (n_param < 0 || (n_param > 0 && !param_idx)) return fail(GST_EINVAL, "bad parameter list");
        if (mode == GST_DERIV_FD) rc = run_dprobs_composite(p, d_out, ld, param_idx, dest_idx, n_param, eps, d_probs_out);
        else rc = run_dprobs_composite_analytic(p, d_out, ld, param_idx, dest_idx, n_param, d_probs_out);
        if (rc) return rc;
        return end_call(p, false);
    }
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
    if ((rc = lindblad_fd_gate(p, mode))) return rc;
    if (p->cmp.set && (mode == GST_DERIV_FD || !p->derivs_set)) {
        if (n_param < 0 || (n_param > 0 && !param_idx)) return fail(GST_EINVAL, "bad parameter list");
        if ((rc = stage_out(p, (size_t)p->hp.n_elements * std::max<int64_t>(n_param, 1)))) return rc;
        if (mode == GST_DERIV_FD) rc = run_dprobs_composite(p, p->d_out.p, n_param, param_idx, nullptr, n_param, eps, nullptr);
        else rc = run_dprobs_composite_analytic(p, p->d_out.p, n_param, param_idx, nullptr, n_param, nullptr);
        if (rc) return rc;
        return copy_out_dprobs(p, out, ld, dest_idx, n_param, probs_out);
    }
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
            if (probs_out && (rc = d2h_bytes(p, probs_out, p->d_pbase.p, (size_t)nE * 8))) return rc;
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

int gst_memcpy_h2d(gst_plan* p, void* d_dst, const void* src, int64_t nbytes)
{
    return guarded([&]() -> int {
    if (!p || !d_dst || !src || nbytes < 0) return fail(GST_EINVAL, "bad argument");
    int rc = ensure_device(p);
    if (rc) return rc;
    gst::track_touch(d_dst, (size_t)nbytes);
    if ((rc = h2d_bytes(p, d_dst, src, (size_t)nbytes))) return rc;
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
    return GST_OK;
    });
}

int gst_device_malloc_tracked(gst_plan* p, int64_t nbytes, void** d_ptr)
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
    if ((rc = d2h_bytes(p, dst, d_src, (size_t)nbytes))) return rc;
    HIP_TRY(hipStreamSynchronize(p->stream));
    return GST_OK;
    });
}

int gst_memcpy_d2h_async(gst_plan* p, void* dst, const void* d_src, int64_t nbytes)
{
    return guarded([&]() -> int {
    if (!p || !dst || !d_src || nbytes < 0) return fail(GST_EINVAL, "bad argument");
    int rc = ensure_device(p);
    if (rc) return rc;
    // (asynchronous for a page-locked destination -- gst_host_register -- and complete on return for a pageable one)
    return d2h_bytes(p, dst, d_src, (size_t)nbytes);
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
    s->last_tiles = p->last_tiles ? p->n_tiles : 0; s->lm_graph_replays = p->lm_graph.failed ? -1 : (int32_t)std::min<int64_t>(p->lm_graph.replays, 0x7fffffff); s->last_tiled_circuits = p->last_tiles ? p->tile_stats[0] : 0;
    if (p->last_fd_form >= 1 && p->d_bin_head.p && p->n_bins > 0 && p->dev_ready) {
        uint32_t flag = 0;                  // the abort flag sits behind the queue heads
        HIP_TRY(hipSetDevice(p->device));
        int rc = d2h_bytes(const_cast<gst_plan*>(p), &flag, p->d_bin_head.p + p->n_bins, 4);      // (through the staging buffer, as every copy)
        if (rc) return rc;
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
    case GST_OPT_ANALYTIC_TILES:
        if (p->rev_ready && (value != 0) != p->ana_tiles) return fail(GST_ESTATE, "GST_OPT_ANALYTIC_TILES must be set before the plan's first exact fill");
        p->ana_tiles = value != 0;
        return GST_OK;
    default:
        return fail(GST_EINVAL, "unknown option " + std::to_string(option));
    }
    });
}

int gst_get_model(gst_plan* p, double* gates, double* rhos, double* effects)
{
    return guarded([&]() -> int {
    if (!p) return fail(GST_EINVAL, "plan is NULL");
    if (!p->have_model) return fail(GST_ESTATE, "no model has been set");
    const int D = p->hp.D, Du = p->user_D();
    if (Du == D) {
        if (gates) std::memcpy(gates, p->h_gates.data(), p->h_gates.size() * 8);
        if (rhos) std::memcpy(rhos, p->h_rhos.data(), p->h_rhos.size() * 8);
        if (effects) std::memcpy(effects, p->h_effects.data(), p->h_effects.size() * 8);
        return GST_OK;
    }
    if (gates)
        for (int g = 0; g < p->hp.n_gates; g++)
            for (int i = 0; i < Du; i++) std::memcpy(gates + ((size_t)g * Du + i) * Du, &p->h_gates[((size_t)g * D + i) * D], (size_t)Du * 8);
    if (rhos) for (int r = 0; r < p->hp.n_rhos; r++) std::memcpy(rhos + (size_t)r * Du, &p->h_rhos[(size_t)r * D], (size_t)Du * 8);
    if (effects) for (int e = 0; e < p->hp.n_effects; e++) std::memcpy(effects + (size_t)e * Du, &p->h_effects[(size_t)e * D], (size_t)Du * 8);
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

int gst_get_fd_work(gst_plan* p, const int64_t* param_idx, int64_t n_param, int64_t* out)
{
    return guarded([&]() -> int {
    if (!p || !out || n_param < 0 || (n_param > 0 && !param_idx)) return fail(GST_EINVAL, "bad argument");
    if (p->hp.D == 64) return fail(GST_EUNSUPPORTED, "the lane-per-model walk exists for D <= 16");
    if (p->hp.n_gates > 64) return fail(GST_EUNSUPPORTED, "more than 64 gates");
    int rc = check_params(p, param_idx, n_param);
    if (rc) return rc;
    LaneLayout L;
    pack_lanes(p, param_idx, nullptr, n_param, L, false);
    std::vector<uint64_t> wg((size_t)L.n_waves, 0);
    std::vector<uint8_t> wr((size_t)L.n_waves, 0), we((size_t)L.n_waves, 0);
    std::vector<int32_t> wc((size_t)L.n_waves, 0);
    for (size_t q = 0; q < L.col.size(); q++) {
        if (L.col[q] < 0) continue;
        const size_t w = q / 64;
        wc[w]++;
        if (L.kind[0][q] == GST_KIND_GATE) wg[w] |= 1ull << L.obj[0][q];
        else if (L.kind[0][q] == GST_KIND_RHO) wr[w] = 1;
        else if (L.kind[0][q] == GST_KIND_EFFECT) we[w] = 1;
    }
    int64_t work[6];
    gst::fd_executed_work(p->hp, wg, wr, we, wc, work);
    for (int i = 0; i < 6; i++) out[i] = work[i];
    out[6] = L.n_waves;
    out[7] = p->hp.n_tasks();
    return GST_OK;
    });
}

}  // extern "C"
