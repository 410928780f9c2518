"""ctypes binding of libgstfwd.so (include/gstfwd.h).  No PyTorch, no numpy-side compute.

The library is the product's only compute path: if it is missing this module raises ImportError-like
RuntimeError at first use, and if no gfx950 device is usable every fill raises `GstDeviceError`.
There is deliberately NO CPU fallback (the CPU checker lives in oracle/ and is test-only).
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GST_LIBGSTFWD") or os.path.join(_HERE, "libgstfwd.so")     # (override: A/B builds in development)

GST_OK = 0
GST_EINVAL, GST_ENODEVICE, GST_EHIP, GST_ENOMEM, GST_ESTATE, GST_EUNSUPPORTED = -1, -2, -3, -4, -5, -6
KIND_NONE, KIND_GATE, KIND_RHO, KIND_EFFECT = -1, 0, 1, 2
DERIV_FD, DERIV_ANALYTIC = 0, 1
OPT_ANALYTIC_KEEP_ZEROS = 1
OPT_FAST_CHAINS = 2       # 0 never / 1 where it pays (default) / 2 always: log-depth level pass for the modes without an ordering contract
OPT_FAST_PROBS = 3        # fill_probs through the level pass (<= 1e-10, not bit-identical)
OPT_ANALYTIC_TILES = 4    # the D = 16 exact contraction over product tiles (off by default; before the first exact fill)
OBJ_CHI2, OBJ_POISSON_DLOGL = 0, 1
TRANSPORT_RCCL, TRANSPORT_IPC = 0, 1
COMM_ID_BYTES = 128

OP_END, OP_RHO, OP_APPLY, OP_SAVE, OP_LOAD, OP_EMIT, OP_NODE = 0, 1, 2, 3, 4, 5, 6


class GstError(RuntimeError):
    pass


class GstUnsupported(GstError):
    """GST_EUNSUPPORTED: the request is outside what the library offers (dimension, mode, depth ...)."""


LINDBLAD_FD_MAX_DEPTH = 16      # GST_LINDBLAD_FD_MAX_DEPTH: FD over device-built Lindblad members meets 1e-8 up to this depth only


class GstDeviceError(GstError):
    """No usable MI355X / HIP runtime.  The library never computes on the CPU instead."""


class TableDesc(C.Structure):
    _fields_ = [("D", C.c_int32), ("n_gates", C.c_int32), ("n_rhos", C.c_int32), ("n_effects", C.c_int32),
                ("n_rows", C.c_int32), ("cache_size", C.c_int32), ("n_elements", C.c_int64),
                ("t_dest", C.c_void_p), ("t_start", C.c_void_p), ("t_cache", C.c_void_p), ("t_rho", C.c_void_p),
                ("row_ptr", C.c_void_p), ("gate_idx", C.c_void_p),
                ("eff_ptr", C.c_void_p), ("eff_label", C.c_void_p), ("eff_dest", C.c_void_p)]


class CircuitsDesc(C.Structure):
    _fields_ = [("D", C.c_int32), ("n_gates", C.c_int32), ("n_rhos", C.c_int32), ("n_effects", C.c_int32),
                ("n_circuits", C.c_int32), ("n_elements", C.c_int64),
                ("circ_rho", C.c_void_p), ("circ_ptr", C.c_void_p), ("circ_gates", C.c_void_p),
                ("eff_ptr", C.c_void_p), ("eff_label", C.c_void_p), ("eff_dest", C.c_void_p)]


class Options(C.Structure):
    _fields_ = [("device", C.c_int32), ("target_tasks", C.c_int32), ("max_slots", C.c_int32),
                ("fd_split", C.c_int32), ("timing", C.c_int32), ("reserved", C.c_int32 * 3)]


class ObjectiveDesc(C.Structure):
    _fields_ = [("kind", C.c_int32), ("hessian_mode", C.c_int32), ("min_prob_clip", C.c_double), ("radius", C.c_double),
                ("prob_clip_lo", C.c_double), ("prob_clip_hi", C.c_double)]


class LindbladMemberDesc(C.Structure):
    _fields_ = [("kind", C.c_int32), ("obj", C.c_int32), ("n_eff", C.c_int32), ("n_blocks", C.c_int32),
                ("block_type", C.c_int32 * 4), ("block_mode", C.c_int32 * 4), ("block_n", C.c_int32 * 4),
                ("param0", C.c_int64), ("term_offset", C.c_int64), ("static_part", C.c_void_p)]


class CompositeDesc(C.Structure):
    _fields_ = [("n_leaves", C.c_int32), ("leaf_dim", C.c_void_p), ("leaf_param", C.c_void_p), ("gate_factor_ptr", C.c_void_p),
                ("factor_leaf", C.c_void_p), ("factor_targets", C.c_void_p), ("leaf_n_params", C.c_void_p), ("leaf_param_list", C.c_void_p)]


class CommInfo(C.Structure):
    _fields_ = [("transport", C.c_int32), ("rank", C.c_int32), ("size", C.c_int32), ("device", C.c_int32),
                ("rccl_version", C.c_int32), ("ipc_opens", C.c_int32), ("reserved", C.c_int32 * 2)]


class Stats(C.Structure):
    _fields_ = [("n_circuits", C.c_int64), ("n_elements", C.c_int64), ("sum_depth", C.c_int64),
                ("trie_nodes", C.c_int64), ("applies_per_pass", C.c_int64), ("n_tasks", C.c_int64),
                ("prog_words", C.c_int64), ("max_slots", C.c_int32), ("max_depth", C.c_int32),
                ("last_kernel_ms", C.c_double), ("last_total_ms", C.c_double), ("last_launches", C.c_int64),
                ("last_fd_form", C.c_int32), ("last_fd_aborted", C.c_int32), ("last_levels", C.c_int32), ("last_zeros_resident", C.c_int32),
                ("last_tiles", C.c_int32), ("lm_graph_replays", C.c_int32), ("last_tiled_circuits", C.c_int64)]


# every symbol include/gstfwd.h declares (tests check the library exports all of them)
EXPORTS = ["gst_plan_create_from_table", "gst_plan_create_from_circuits", "gst_plan_destroy", "gst_set_model",
           "gst_set_param_map", "gst_set_complement_effect", "gst_set_derivs", "gst_set_second_derivs", "gst_fill_probs", "gst_fill_dprobs", "gst_fill_hprobs", "gst_fill_hprobs_analytic", "gst_fill_probs_dev",
           "gst_fill_dprobs_dev", "gst_fill_jtj_dev", "gst_fill_jtf_dev", "gst_fill_normal_eqs_dev", "gst_objective_rows_dev", "gst_lm_step_dev", "gst_objective_hessian_block", "gst_memcpy_h2d", "gst_copy_block_dev", "gst_sync", "gst_device_malloc", "gst_device_malloc_tracked", "gst_device_free", "gst_device_touch", "gst_memcpy_d2h", "gst_memcpy_d2h_async", "gst_get_stats", "gst_get_program", "gst_get_level_program", "gst_get_dirty_programs", "gst_get_state_graph", "gst_get_fd_queues", "gst_get_fd_work", "gst_get_tile_stats", "gst_sort_circuits", "gst_circuit_first_use", "gst_device_count",
           "gst_last_error", "gst_version", "gst_host_register", "gst_host_unregister",
           "gst_fill_dprobs_models", "gst_fill_dprobs_models_dev",
           "gst_set_option", "gst_set_lindblad", "gst_set_lindblad_params", "gst_set_composite", "gst_set_composite_values", "gst_set_composite_general", "gst_get_model", "gst_get_lindblad_model_sets",
           "gst_comm_get_unique_id", "gst_comm_create", "gst_comm_destroy", "gst_comm_allgather_rows",
           "gst_comm_gather_rows", "gst_comm_map_root_buffer", "gst_comm_exchange_blocks", "gst_comm_allreduce_sum", "gst_comm_barrier", "gst_comm_sync", "gst_comm_get_info"]

_lib = None


def lib():
    """Load libgstfwd.so (once).  Fails loudly when the HIP extension has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise GstError("libgstfwd.so not found at %s: build it with `make -C pygsti_amd/csrc` "
                           "(or __graft_entry__.build()); there is no CPU fallback" % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        L.gst_last_error.restype = C.c_char_p
        L.gst_version.restype = C.c_char_p
        vp, i64, i32, dbl = C.c_void_p, C.c_int64, C.c_int32, C.c_double
        L.gst_plan_create_from_table.argtypes = [C.POINTER(TableDesc), C.POINTER(Options), C.POINTER(vp)]
        L.gst_plan_create_from_circuits.argtypes = [C.POINTER(CircuitsDesc), C.POINTER(Options), C.POINTER(vp)]
        L.gst_plan_destroy.argtypes = [vp]
        L.gst_set_model.argtypes = [vp, vp, vp, vp]
        L.gst_set_param_map.argtypes = [vp, i32, vp, vp, vp]
        L.gst_fill_probs.argtypes = [vp, vp]
        L.gst_fill_dprobs.argtypes = [vp, vp, i64, vp, vp, i64, C.c_int, dbl, vp]
        L.gst_fill_hprobs.argtypes = [vp, vp, i64, i64, vp, vp, i64, vp, vp, i64, dbl]
        L.gst_fill_hprobs_analytic.argtypes = [vp, vp, i64, i64, vp, vp, i64, vp, vp, i64]
        L.gst_fill_probs_dev.argtypes = [vp, vp]
        L.gst_fill_dprobs_dev.argtypes = [vp, vp, i64, vp, vp, i64, C.c_int, dbl, vp]
        L.gst_fill_jtj_dev.argtypes = [vp, vp, i64, i64, i64, vp, vp]
        L.gst_fill_jtf_dev.argtypes = [vp, vp, i64, i64, i64, vp, vp]
        L.gst_fill_normal_eqs_dev.argtypes = [vp, vp, i64, i64, i64, vp, vp, vp, vp]
        L.gst_set_derivs.argtypes = [vp, i32, i32, vp, vp, vp, vp, vp]
        L.gst_set_complement_effect.argtypes = [vp, i32, vp, i32, vp]
        L.gst_set_second_derivs.argtypes = [vp, i32, vp, vp]
        L.gst_objective_rows_dev.argtypes = [vp, C.POINTER(ObjectiveDesc), vp, vp, vp, i64, vp, vp, vp, C.POINTER(C.c_double)]
        L.gst_objective_hessian_block.argtypes = [vp, C.POINTER(ObjectiveDesc), vp, vp, vp, i64, vp, i64, C.c_double, vp]
        L.gst_memcpy_h2d.argtypes = [vp, vp, vp, i64]
        L.gst_copy_block_dev.argtypes = [vp, vp, i64, vp, i64, i64, i64]
        L.gst_sync.argtypes = [vp]
        L.gst_set_composite.argtypes = [vp, i32, C.POINTER(CompositeDesc)]
        L.gst_set_composite_values.argtypes = [vp, vp, vp, vp]
        L.gst_set_composite_general.argtypes = [vp, vp, vp, dbl]
        L.gst_device_malloc.argtypes = [vp, i64, C.POINTER(vp)]
        L.gst_device_malloc_tracked.argtypes = [vp, i64, C.POINTER(vp)]
        L.gst_device_free.argtypes = [vp, vp]
        L.gst_device_touch.argtypes = [vp, vp, C.c_int64]
        L.gst_memcpy_d2h.argtypes = [vp, vp, vp, i64]
        L.gst_memcpy_d2h_async.argtypes = [vp, vp, vp, i64]
        L.gst_get_stats.argtypes = [vp, C.POINTER(Stats)]
        L.gst_get_program.argtypes = [vp, vp, i64, C.POINTER(i64), vp, i64]
        L.gst_get_dirty_programs.argtypes = [vp, vp, i64, C.POINTER(i64), vp, i64, C.POINTER(i32)]
        L.gst_get_fd_queues.argtypes = [vp, vp, i64, i32, i32, vp, C.POINTER(i32), C.POINTER(i32)]
        L.gst_get_fd_work.argtypes = [vp, vp, i64, vp]
        L.gst_get_tile_stats.argtypes = [vp, vp]
        L.gst_lm_step_dev.argtypes = [vp, vp, i64, C.c_double, vp, vp, vp, i64, vp, vp, vp, vp, vp, C.POINTER(C.c_double)]
        L.gst_sort_circuits.argtypes = [i64, vp, vp, vp, vp, vp]
        L.gst_circuit_first_use.argtypes = [i64, vp, vp, i32, vp]
        L.gst_get_state_graph.argtypes = [vp, vp, vp, i64, vp, i64, C.POINTER(i64)]
        L.gst_device_count.argtypes = [C.POINTER(i32)]
        L.gst_fill_dprobs_models.argtypes = [vp, i64, vp, vp, vp, vp, i64, vp, dbl, vp]
        L.gst_fill_dprobs_models_dev.argtypes = [vp, i64, vp, vp, vp, vp, i64, vp, dbl, vp]
        L.gst_set_option.argtypes = [vp, i32, i64]
        L.gst_set_lindblad.argtypes = [vp, i32, i32, C.POINTER(LindbladMemberDesc), i64, vp, vp]
        L.gst_set_lindblad_params.argtypes = [vp, vp]
        L.gst_get_model.argtypes = [vp, vp, vp, vp]
        L.gst_get_lindblad_model_sets.argtypes = [vp, vp, i64, dbl, vp, vp, vp]
        L.gst_host_register.argtypes = [vp, i64]
        L.gst_host_unregister.argtypes = [vp]
        L.gst_comm_get_unique_id.argtypes = [C.c_int, vp]
        L.gst_comm_create.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, vp, C.POINTER(vp)]
        L.gst_comm_destroy.argtypes = [vp]
        L.gst_comm_allgather_rows.argtypes = [vp, vp, vp, i64, i32, vp, vp, vp]
        L.gst_comm_gather_rows.argtypes = [vp, vp, vp, vp, i64, i32, vp, vp, vp, i32]
        L.gst_comm_map_root_buffer.argtypes = [vp, i32, vp, C.POINTER(C.c_void_p)]
        L.gst_comm_exchange_blocks.argtypes = [vp, vp, vp, vp, i32, vp, vp, vp, vp, vp]
        L.gst_comm_allreduce_sum.argtypes = [vp, vp, vp, i64]
        L.gst_comm_barrier.argtypes = [vp]
        L.gst_comm_sync.argtypes = [vp]
        L.gst_comm_get_info.argtypes = [vp, C.POINTER(CommInfo)]
        _lib = L
    return _lib


def check(rc):
    if rc == GST_OK:
        return
    msg = lib().gst_last_error().decode("utf-8", "replace")
    if rc == GST_ENOMEM:
        raise MemoryError(msg)          # the reference raises MemoryError (mapforwardsim.py:374-387)
    if rc == GST_ENODEVICE:
        raise GstDeviceError(msg)
    if rc == GST_EINVAL:
        raise ValueError(msg)
    if rc == GST_EUNSUPPORTED:
        raise GstUnsupported("libgstfwd error %d: %s" % (rc, msg))
    raise GstError("libgstfwd error %d: %s" % (rc, msg))


def device_count():
    n = C.c_int32(0)
    check(lib().gst_device_count(C.byref(n)))
    return n.value


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def pin_host_array(arr):
    """Page-lock a C-contiguous numpy array for full-rate PCIe copies (gst_host_register); the lock is released when
    the array is garbage-collected or through `unpin_host_array`.  Returns True when pinned, False when no device
    exists (the array then stays pageable -- fills still work where a device is).  Meant for arrays with pages of their own
    (numpy serves >= 64 MB by an mmap per array; the library's own callers page-lock nothing smaller): a range inside the
    brk heap shares its first and last page with other allocations."""
    import weakref
    if not arr.flags.c_contiguous or arr.nbytes == 0:
        return False
    addr = arr.ctypes.data
    try:
        check(lib().gst_host_register(C.c_void_p(addr), arr.nbytes))
    except (GstDeviceError, GstError):
        return False
    _pinned[addr] = weakref.finalize(arr, _unregister, addr)
    return True


def unpin_host_array(arr):
    fin = _pinned.pop(arr.ctypes.data, None)
    if fin is not None:
        fin()


def _unregister(addr):
    try:
        lib().gst_host_unregister(C.c_void_p(addr))
    except Exception:
        pass


_pinned = {}


def _i32(x):
    return np.ascontiguousarray(x, dtype=np.int32)


def _i64(x):
    return np.ascontiguousarray(x, dtype=np.int64)


def _f64(x):
    return np.ascontiguousarray(x, dtype=np.float64)


class Plan:
    """One compiled evaluation plan (== one layout atom) living on one GPU."""

    def __init__(self, handle, D, n_gates, n_rhos, n_effects, n_elements):
        self._h = handle
        self.D, self.n_gates, self.n_rhos, self.n_effects = D, n_gates, n_rhos, n_effects
        self.n_elements = int(n_elements)
        self.n_params = 0

    # -- construction --------------------------------------------------------------------------
    @classmethod
    def from_table(cls, D, n_gates, n_rhos, n_effects, n_elements, cache_size, t_dest, t_start, t_cache, t_rho,
                   row_ptr, gate_idx, eff_ptr, eff_label, eff_dest, device=-1, target_tasks=0, max_slots=0, fd_split=0,
                   timing=0):
        a = dict(t_dest=_i32(t_dest), t_start=_i32(t_start), t_cache=_i32(t_cache), t_rho=_i32(t_rho),
                 row_ptr=_i64(row_ptr), gate_idx=_i32(gate_idx) if len(gate_idx) else np.zeros(1, np.int32),
                 eff_ptr=_i64(eff_ptr), eff_label=_i32(eff_label), eff_dest=_i32(eff_dest))
        d = TableDesc(int(D), int(n_gates), int(n_rhos), int(n_effects), len(a["t_dest"]), int(cache_size),
                      int(n_elements), _ptr(a["t_dest"]), _ptr(a["t_start"]), _ptr(a["t_cache"]), _ptr(a["t_rho"]),
                      _ptr(a["row_ptr"]), _ptr(a["gate_idx"]), _ptr(a["eff_ptr"]), _ptr(a["eff_label"]),
                      _ptr(a["eff_dest"]))
        opt = Options(int(device), int(target_tasks), int(max_slots), int(fd_split), int(timing))
        h = C.c_void_p()
        check(lib().gst_plan_create_from_table(C.byref(d), C.byref(opt), C.byref(h)))
        return cls(h, int(D), int(n_gates), int(n_rhos), int(n_effects), n_elements)

    @classmethod
    def from_circuits(cls, D, n_gates, n_rhos, n_effects, n_elements, circ_rho, circ_ptr, circ_gates,
                      eff_ptr, eff_label, eff_dest, device=-1, target_tasks=0, max_slots=0, fd_split=0, timing=0):
        a = dict(circ_rho=_i32(circ_rho), circ_ptr=_i64(circ_ptr),
                 circ_gates=_i32(circ_gates) if len(circ_gates) else np.zeros(1, np.int32),
                 eff_ptr=_i64(eff_ptr), eff_label=_i32(eff_label), eff_dest=_i32(eff_dest))
        d = CircuitsDesc(int(D), int(n_gates), int(n_rhos), int(n_effects), len(a["circ_rho"]), int(n_elements),
                         _ptr(a["circ_rho"]), _ptr(a["circ_ptr"]), _ptr(a["circ_gates"]),
                         _ptr(a["eff_ptr"]), _ptr(a["eff_label"]), _ptr(a["eff_dest"]))
        opt = Options(int(device), int(target_tasks), int(max_slots), int(fd_split), int(timing))
        h = C.c_void_p()
        check(lib().gst_plan_create_from_circuits(C.byref(d), C.byref(opt), C.byref(h)))
        return cls(h, int(D), int(n_gates), int(n_rhos), int(n_effects), n_elements)

    def close(self):
        if self._h is not None and self._h.value:
            for ptr, _ in getattr(self, "_workspaces", {}).values():
                try:
                    lib().gst_device_free(self._h, C.c_void_p(int(ptr)))
                except Exception:
                    pass
            self._workspaces = {}
            lib().gst_plan_destroy(self._h)
        self._h = None

    def workspace(self, name, nbytes):
        """A named device buffer that lives as long as the plan (grown on demand): what an optimizer's iterations reuse
        instead of allocating the Jacobian every call."""
        ws = self.__dict__.setdefault("_workspaces", {})
        have = ws.get(name)
        if have is not None and have[1] >= nbytes:
            return have[0]
        if have is not None:
            self.device_free(have[0])
            del ws[name]
        ptr = self.device_malloc(max(int(nbytes), 8), tracked=True)      # (only this wrapper writes its workspaces)
        ws[name] = (ptr, int(nbytes))
        return ptr

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- model -------------------------------------------------------------------------------------
    def set_model(self, gates, rhos, effects):
        g = _f64(gates).reshape(self.n_gates, self.D, self.D) if self.n_gates else np.zeros((1,), np.float64)
        r = _f64(rhos).reshape(self.n_rhos, self.D)
        e = _f64(effects).reshape(self.n_effects, self.D)
        check(lib().gst_set_model(self._h, _ptr(g), _ptr(r), _ptr(e)))

    def set_param_map(self, kind, obj, elem):
        k, o, e = _i32(kind), _i32(obj), _i32(elem)
        assert len(k) == len(o) == len(e)
        check(lib().gst_set_param_map(self._h, len(k), _ptr(k), _ptr(o), _ptr(e)))
        self.n_params = len(k)

    def set_complement_effect(self, comp_index, identity=None, others=()):
        """Declare effect `comp_index` as a TP POVM's complement, identity - sum(effects[others]) in that order
        (gst_set_complement_effect); comp_index < 0 clears.  Only the FD Jacobian honours it."""
        if comp_index is None or comp_index < 0:
            check(lib().gst_set_complement_effect(self._h, -1, None, 0, None))
            return
        ident = _f64(np.asarray(identity, np.float64).ravel())
        assert ident.size == self.D, "identity must have D components"
        oth = _i32(others)
        check(lib().gst_set_complement_effect(self._h, int(comp_index), _ptr(ident), len(oth), _ptr(oth)))

    def set_derivs(self, n_params, objs):
        """General parameterisations for the analytic mode: objs = [(kind, obj, param_idx[n], deriv[n_elem, n]), ...]
        (gst_set_derivs); an empty list clears."""
        objs = list(objs)
        if not objs:
            check(lib().gst_set_derivs(self._h, int(n_params), 0, None, None, None, None, None))
            return
        kind = _i32([o[0] for o in objs]); obj = _i32([o[1] for o in objs])
        ncols = _i32([len(o[2]) for o in objs])
        pidx = _i64(np.concatenate([np.asarray(o[2], np.int64).ravel() for o in objs]))
        ne = [self.D * self.D if o[0] == KIND_GATE else self.D for o in objs]
        for o, n in zip(objs, ne):
            assert np.asarray(o[3]).shape == (n, len(o[2])), "deriv must be [n_elem, n_params_of_object]"
        der = _f64(np.concatenate([np.ascontiguousarray(o[3], np.float64).ravel() for o in objs]))
        check(lib().gst_set_derivs(self._h, int(n_params), len(objs), _ptr(kind), _ptr(obj), _ptr(ncols), _ptr(pidx), _ptr(der)))
        self.n_params = int(n_params)

    def set_option(self, option, value):
        """gst_set_option: OPT_ANALYTIC_KEEP_ZEROS, ..."""
        check(lib().gst_set_option(self._h, int(option), int(value)))

    def set_lindblad(self, model):
        """Lindblad-parameterised members built on the device (gst_set_lindblad): `model` is a
        pygsti_amd.lindblad.LindbladModel (or anything with .members / .num_params in its shape); None clears."""
        if model is None:
            check(lib().gst_set_lindblad(self._h, 0, 0, None, 0, None, None))
            return
        members = list(model.members)
        # members over the same basis with the same block structure share their terms
        tables, offsets, keep = {}, [], []
        n_terms = 0
        for m in members:
            # (members built from Pauli matrices share a table per block structure; given term arrays are shared when they
            #  are the same array contents)
            tk = getattr(m, "_term_key", None)
            key = (m.n_qubits, tuple(m.blocks), tk if tk is not None else hash(m.term_re.tobytes()) ^ hash(m.term_im.tobytes()))
            if key not in tables:
                tables[key] = n_terms
                n_terms += m.n_coeffs
                keep.append(m)
            offsets.append(tables[key])
        term_re = _f64(np.concatenate([m.term_re.reshape(m.n_coeffs, -1) for m in keep]))
        term_im = _f64(np.concatenate([m.term_im.reshape(m.n_coeffs, -1) for m in keep]))
        arr = (LindbladMemberDesc * len(members))()
        statics = []
        for k, m in enumerate(members):
            st = _f64(m.static)
            statics.append(st)
            d = arr[k]
            d.kind = {0: KIND_GATE, 1: KIND_RHO, 2: KIND_EFFECT}[m.kind]; d.obj = m.obj; d.n_eff = m.n_eff; d.n_blocks = len(m.blocks)
            for b, (t, mode, n) in enumerate(m.blocks):
                d.block_type[b] = t; d.block_mode[b] = mode; d.block_n[b] = n
            d.param0 = m.param0; d.term_offset = offsets[k]; d.static_part = st.ctypes.data
        check(lib().gst_set_lindblad(self._h, int(model.num_params), len(members), arr, int(n_terms), _ptr(term_re), _ptr(term_im)))
        self.n_params = int(model.num_params)

    def set_composite(self, model):
        """gst_set_composite: the layer structure of an implicit model (a composite.CompositeModel; None clears)."""
        if model is None:
            check(lib().gst_set_composite(self._h, 0, None))
            self._composite = None
            return
        assert len(model.gate_factors) == self.n_gates and model.D == self.D
        leaf_dim, leaf_param, fptr, fl, ft = model.arrays()
        n_gen, gen_list = model.general_arrays()
        keep = [_i32(leaf_dim), _i64(leaf_param), _i32(fptr), _i32(fl) if len(fl) else np.zeros(1, np.int32),
                _i32(ft) if len(ft) else np.full((1, 3), -1, np.int32), _i32(n_gen), _i64(gen_list) if len(gen_list) else np.zeros(1, np.int64)]
        d = CompositeDesc(len(leaf_dim), *[a.ctypes.data for a in keep])
        check(lib().gst_set_composite(self._h, int(model.num_params), C.byref(d)))
        self._composite = model
        self.n_params = int(model.num_params)

    def set_composite_values(self, leaf_values, rhos, effects):
        """gst_set_composite_values: the leaves' current elements + the dense SPAM vectors; the device builds every layer."""
        v = _f64(leaf_values)
        r = _f64(rhos).reshape(self.n_rhos, self.D); e = _f64(effects).reshape(self.n_effects, self.D)
        check(lib().gst_set_composite_values(self._h, _ptr(v), _ptr(r), _ptr(e)))

    def set_composite_general(self, leaf_derivs=None, leaf_fd_values=None, fd_eps=1e-7):
        """gst_set_composite_general: the general leaves' deriv_wrt_params ([d*d][np] each, concatenated) and / or their dense
        elements after every parameter's finite-difference step ([np][d*d] each, concatenated), for the current parameters."""
        dv = None if leaf_derivs is None else _f64(leaf_derivs)
        fd = None if leaf_fd_values is None else _f64(leaf_fd_values)
        check(lib().gst_set_composite_general(self._h, _ptr(dv), _ptr(fd), float(fd_eps)))

    def set_lindblad_params(self, theta):
        """The model's parameter vector: the device builds every dense member from it (gst_set_lindblad_params)."""
        th = _f64(theta)
        assert len(th) == self.n_params
        check(lib().gst_set_lindblad_params(self._h, _ptr(th)))

    def get_model(self):
        """(gates, rhos, effects) the plan currently holds (gst_get_model)."""
        D = self.D
        g = np.empty((self.n_gates, D, D)); r = np.empty((self.n_rhos, D)); e = np.empty((self.n_effects, D))
        check(lib().gst_get_model(self._h, _ptr(g), _ptr(r), _ptr(e)))
        return g, r, e

    def lindblad_model_sets(self, param_idx, eps):
        """Dense models the device builds for the FD steps of `param_idx` (gst_get_lindblad_model_sets)."""
        pidx = _i64(param_idx); n = len(pidx); D = self.D
        g = np.empty((n, self.n_gates, D, D)); r = np.empty((n, self.n_rhos, D)); e = np.empty((n, self.n_effects, D))
        check(lib().gst_get_lindblad_model_sets(self._h, _ptr(pidx), n, float(eps), _ptr(g), _ptr(r), _ptr(e)))
        return g, r, e

    def set_second_derivs(self, hessians):
        """hessian_wrt_params of the objects given to set_derivs, in the same order: a list whose entries are None
        (member linear in its parameters) or arrays [n_elem, n, n] (gst_set_second_derivs); an empty list clears."""
        hs = list(hessians)
        if not hs:
            check(lib().gst_set_second_derivs(self._h, 0, None, None))
            return
        nz = _i32([0 if h is None else 1 for h in hs])
        parts = [np.ascontiguousarray(h, np.float64).ravel() for h in hs if h is not None]
        flat = _f64(np.concatenate(parts)) if parts else np.zeros(1)
        check(lib().gst_set_second_derivs(self._h, len(hs), _ptr(nz), _ptr(flat)))

    # -- fills (host arrays) ---------------------------------------------------------------------------
    def fill_probs(self, out=None):
        if out is None:
            out = np.empty(self.n_elements, np.float64)
        assert out.dtype == np.float64 and out.flags.c_contiguous and out.size == self.n_elements
        check(lib().gst_fill_probs(self._h, _ptr(out)))
        return out

    def fill_dprobs(self, out=None, param_idx=None, dest_idx=None, eps=1e-7, probs_out=None, mode=DERIV_FD):
        if param_idx is None:
            param_idx = np.arange(self.n_params)
        pi = _i64(param_idx)
        di = None if dest_idx is None else _i64(dest_idx)
        if out is None:
            ncol = len(pi) if di is None else (int(di.max()) + 1 if len(di) else 0)
            out = np.empty((self.n_elements, ncol), np.float64)
        assert out.dtype == np.float64 and out.ndim == 2 and out.shape[0] == self.n_elements
        if out.shape[1] == 0 or out.shape[0] == 0:
            ld = max(out.shape[1], 1)
        else:
            assert out.strides[1] == 8 and out.strides[0] % 8 == 0, "rows must be contiguous in the parameter dimension"
            ld = out.strides[0] // 8 if out.shape[0] > 1 else max(out.shape[1], 1)
        if probs_out is not None:
            assert probs_out.dtype == np.float64 and probs_out.flags.c_contiguous and probs_out.size == self.n_elements
        check(lib().gst_fill_dprobs(self._h, _ptr(out), ld, _ptr(pi), _ptr(di), len(pi), int(mode), float(eps),
                                    _ptr(probs_out)))
        return out

    def fill_dprobs_models(self, gates, rhos, effects, out=None, dest_idx=None, eps=1e-7, probs_out=None):
        """FD columns from explicitly perturbed dense model sets (gst_fill_dprobs_models): gates [n, nG, D, D],
        rhos [n, nR, D], effects [n, nEl, D]; column dest_idx[m] = (p(set m) - p(base model)) / eps."""
        r = _f64(rhos).reshape(-1, self.n_rhos, self.D)
        n = r.shape[0]
        g = _f64(gates).reshape(n, self.n_gates, self.D, self.D) if self.n_gates else np.zeros((1,), np.float64)
        e = _f64(effects).reshape(n, self.n_effects, self.D)
        di = None if dest_idx is None else _i64(dest_idx)
        if out is None:
            out = np.empty((self.n_elements, n if di is None else (int(di.max()) + 1 if len(di) else 0)), np.float64)
        assert out.dtype == np.float64 and out.ndim == 2 and out.shape[0] == self.n_elements
        if out.shape[1] == 0 or out.shape[0] == 0:
            ld = max(out.shape[1], 1)
        else:
            assert out.strides[1] == 8 and out.strides[0] % 8 == 0, "rows must be contiguous in the parameter dimension"
            ld = out.strides[0] // 8 if out.shape[0] > 1 else max(out.shape[1], 1)
        if probs_out is not None:
            assert probs_out.dtype == np.float64 and probs_out.flags.c_contiguous and probs_out.size == self.n_elements
        check(lib().gst_fill_dprobs_models(self._h, n, _ptr(g), _ptr(r), _ptr(e), _ptr(out), ld, _ptr(di), float(eps),
                                           _ptr(probs_out)))
        return out

    def fill_hprobs_models(self, gates, rhos, effects, eps=1e-5):
        """FD-of-FD Hessian block from two-level perturbed dense model sets, composed exactly as
        MapForwardSimulator._mapfill_hprobs_atom does (mapforwardsim.py:420-436): gates [n1+1, n2+1, nG, D, D] (rhos,
        effects alike); [r, 0] is the model at theta (r = 0) or theta + eps e_{i_r} (r >= 1), [r, 1+c] that model after the
        FD step of column parameter c.  Row r's Jacobian block comes from gst_fill_dprobs_models with [r, 0] as the base
        model; H[:, r-1, :] = (dprobs_r - dprobs_0) / eps on the host, as the reference's numpy line does."""
        r = _f64(rhos)
        n1p, n2p = r.shape[0], r.shape[1]
        g = _f64(gates).reshape(n1p, n2p, self.n_gates, self.D, self.D)
        r = r.reshape(n1p, n2p, self.n_rhos, self.D)
        e = _f64(effects).reshape(n1p, n2p, self.n_effects, self.D)
        H = np.empty((self.n_elements, n1p - 1, n2p - 1), np.float64)
        d0 = None
        for k in range(n1p):
            self.set_model(g[k, 0], r[k, 0], e[k, 0])
            dk = self.fill_dprobs_models(g[k, 1:], r[k, 1:], e[k, 1:], eps=eps)
            if k == 0:
                d0 = dk
            else:
                H[:, k - 1, :] = (dk - d0) / eps
        return H

    def fill_hprobs(self, out=None, idx1=None, idx2=None, dest1=None, dest2=None, eps=1e-5, mode=DERIV_FD):
        i1 = _i64(np.arange(self.n_params) if idx1 is None else idx1)
        i2 = _i64(np.arange(self.n_params) if idx2 is None else idx2)
        d1 = None if dest1 is None else _i64(dest1)
        d2 = None if dest2 is None else _i64(dest2)
        if out is None:
            out = np.empty((self.n_elements, len(i1), len(i2)), np.float64)
        assert out.dtype == np.float64 and out.ndim == 3 and out.flags.c_contiguous
        if mode == DERIV_ANALYTIC:
            check(lib().gst_fill_hprobs_analytic(self._h, _ptr(out), out.shape[1], out.shape[2], _ptr(i1), _ptr(d1), len(i1),
                                                 _ptr(i2), _ptr(d2), len(i2)))
        else:
            check(lib().gst_fill_hprobs(self._h, _ptr(out), out.shape[1], out.shape[2], _ptr(i1), _ptr(d1), len(i1),
                                        _ptr(i2), _ptr(d2), len(i2), float(eps)))
        return out

    # -- fills (device pointers) -------------------------------------------------------------------------
    def fill_probs_dev(self, d_out_ptr):
        check(lib().gst_fill_probs_dev(self._h, C.c_void_p(int(d_out_ptr))))

    def fill_dprobs_dev(self, d_out_ptr, ld, param_idx, dest_idx=None, eps=1e-7, d_probs_ptr=None, mode=DERIV_FD):
        pi = _i64(param_idx)
        di = None if dest_idx is None else _i64(dest_idx)
        check(lib().gst_fill_dprobs_dev(self._h, C.c_void_p(int(d_out_ptr)), int(ld), _ptr(pi), _ptr(di), len(pi),
                                        int(mode), float(eps),
                                        None if d_probs_ptr is None else C.c_void_p(int(d_probs_ptr))))

    def sync(self):
        check(lib().gst_sync(self._h))

    def fill_jtj_dev(self, d_J, n_rows, n_cols, ld, d_jtj, d_row_scale=None):
        check(lib().gst_fill_jtj_dev(self._h, C.c_void_p(int(d_J)), int(n_rows), int(n_cols), int(ld),
                                     None if d_row_scale is None else C.c_void_p(int(d_row_scale)), C.c_void_p(int(d_jtj))))

    def fill_jtf_dev(self, d_J, n_rows, n_cols, ld, d_f, d_jtf):
        check(lib().gst_fill_jtf_dev(self._h, C.c_void_p(int(d_J)), int(n_rows), int(n_cols), int(ld),
                                     C.c_void_p(int(d_f)), C.c_void_p(int(d_jtf))))

    def fill_normal_eqs_dev(self, d_J, n_rows, n_cols, ld, d_row_scale=None, d_f=None, d_jtj=None, d_jtf=None):
        """(diag(w) J)^T (diag(w) J) -> d_jtj and/or (diag(w) J)^T f -> d_jtf with the weights applied on the fly: d_J is
        only read (gst_fill_normal_eqs_dev).  d_jtj has the bits of fill_jtj_dev(..., d_row_scale); d_jtf has the bits of fill_jtf_dev
        EXCEPT on the block-sparse path (>= 16,384 rows, > 384 columns, both outputs requested), where the pass that marks
        the live panels also carries J_s^T f: deterministic, but another fixed summation order (include/gstfwd.h)."""
        vp = lambda x: None if x is None else C.c_void_p(int(x))
        check(lib().gst_fill_normal_eqs_dev(self._h, C.c_void_p(int(d_J)), int(n_rows), int(n_cols), int(ld), vp(d_row_scale),
                                            vp(d_f), vp(d_jtj), vp(d_jtf)))

    def objective_rows_dev(self, kind, d_probs, d_counts, d_totals, n, d_lsvec, d_rowscale, d_terms=None,
                           min_prob_clip=1e-4, radius=1e-4, prob_clip_interval=None, want_sum=True):
        """Element-wise objective maps on device probabilities (gst_objective_rows_dev).  kind: 'chi2' | 'logl'.
        Returns sum(terms) when want_sum (blocks), else None."""
        k = {"chi2": OBJ_CHI2, "logl": OBJ_POISSON_DLOGL}[kind] if isinstance(kind, str) else int(kind)
        lo, hi = (0.0, 0.0) if prob_clip_interval is None else (float(prob_clip_interval[0]), float(prob_clip_interval[1]))
        d = ObjectiveDesc(k, 0, float(min_prob_clip), float(radius), lo, hi)
        s = C.c_double(0.0)
        check(lib().gst_objective_rows_dev(self._h, C.byref(d), C.c_void_p(int(d_probs)), C.c_void_p(int(d_counts)),
                                           C.c_void_p(int(d_totals)), int(n), C.c_void_p(int(d_lsvec)),
                                           C.c_void_p(int(d_rowscale)),
                                           None if d_terms is None else C.c_void_p(int(d_terms)),
                                           C.byref(s) if want_sum else None))
        return s.value if want_sum else None

    def lsq_step(self, n_params, counts, total_counts, objective="logl", eps=1e-7, mode=DERIV_FD, min_prob_clip=1e-4,
                 radius=1e-4, prob_clip_interval=None, lsvec_out=None, probs_out=None):
        """Everything one Levenberg-Marquardt iteration needs from this plan's rows without the Jacobian leaving HBM:
        probabilities and Jacobian (gst_fill_dprobs_dev, FD or analytic; the model / parameter map / derivative state of
        the plan must be set), the objective's element-wise maps (lsvec and the dlsvec row scale of
        TimeIndependentMDCObjectiveFunction, objectivefns.py:4573-4665), then J_s^T J_s and J_s^T lsvec
        (optimize/simplerlm.py:677-678).  counts / total_counts: host arrays in the plan's element order.
        Returns (sum(terms), jtj [nP, nP], jtf [nP]); buffers are plan-lifetime workspaces (an optimizer calls this every
        iteration)."""
        nE, nP = self.n_elements, int(n_params)
        sizes = (nE * nP * 8, nE * 8, nE * 8, nE * 8, nE * 8, nE * 8, nP * nP * 8, nP * 8)
        d_J, d_pr, d_c, d_N, d_ls, d_w, d_jtj, d_jtf = [self.workspace("lsq%d" % k, nb) for k, nb in enumerate(sizes)]
        self.memcpy_h2d(d_c, np.ascontiguousarray(counts, np.float64)); self.memcpy_h2d(d_N, np.ascontiguousarray(total_counts, np.float64))
        self.fill_dprobs_dev(d_J, nP, np.arange(nP, dtype=np.int64), None, eps, d_pr, mode)
        total = self.objective_rows_dev(objective, d_pr, d_c, d_N, nE, d_ls, d_w, None, min_prob_clip, radius, prob_clip_interval)
        self.fill_normal_eqs_dev(d_J, nE, nP, nP, d_w, d_ls, d_jtj, d_jtf)     # weights applied on the fly: J is only read
        jtj = np.empty((nP, nP)); self.memcpy_d2h(jtj, d_jtj)
        jtf = np.empty(nP); self.memcpy_d2h(jtf, d_jtf)
        if lsvec_out is not None:
            self.memcpy_d2h(lsvec_out, d_ls)
        if probs_out is not None:
            self.memcpy_d2h(probs_out, d_pr)
        return total, jtj, jtf

    def lm_step_dev(self, n_params, d_counts, d_totals, d_J, ld, d_probs, d_lsvec, d_rowscale, d_jtj, d_jtf, objective="logl",
                    eps=1e-7, min_prob_clip=1e-4, radius=1e-4, prob_clip_interval=None):
        """gst_lm_step_dev: model upload -> FD Jacobian -> objective rows -> J_s^T J_s, J_s^T lsvec in one blocking call on device
        buffers the caller keeps (replayed as ONE HIP graph launch on launch-bound plans from the third call on).  Returns
        sum(terms)."""
        k = {"chi2": OBJ_CHI2, "logl": OBJ_POISSON_DLOGL}[objective] if isinstance(objective, str) else int(objective)
        lo, hi = (0.0, 0.0) if prob_clip_interval is None else (float(prob_clip_interval[0]), float(prob_clip_interval[1]))
        d = ObjectiveDesc(k, 0, float(min_prob_clip), float(radius), lo, hi)
        vp = lambda x: C.c_void_p(int(x))
        tot = C.c_double(0.0)
        check(lib().gst_lm_step_dev(self._h, C.byref(d), int(n_params), float(eps), vp(d_counts), vp(d_totals), vp(d_J), int(ld), vp(d_probs),
                                    vp(d_lsvec), vp(d_rowscale), vp(d_jtj), vp(d_jtf), C.byref(tot)))
        return tot.value

    def objective_hessian_block(self, kind, d_counts, d_totals, idx1, idx2, eps=1e-5, min_prob_clip=1e-4, radius=1e-4,
                                prob_clip_interval=None, mode=DERIV_FD):
        """(len(idx1), len(idx2)) block of the objective's Hessian, contracted over this plan's elements on the device
        (gst_objective_hessian_block).  kind: 'chi2' | 'logl'."""
        k = {"chi2": OBJ_CHI2, "logl": OBJ_POISSON_DLOGL}[kind] if isinstance(kind, str) else int(kind)
        lo, hi = (0.0, 0.0) if prob_clip_interval is None else (float(prob_clip_interval[0]), float(prob_clip_interval[1]))
        d = ObjectiveDesc(k, int(mode), float(min_prob_clip), float(radius), lo, hi)
        i1, i2 = _i64(idx1), _i64(idx2)
        out = np.zeros((len(i1), len(i2)), np.float64)
        check(lib().gst_objective_hessian_block(self._h, C.byref(d), C.c_void_p(int(d_counts)), C.c_void_p(int(d_totals)),
                                                _ptr(i1), len(i1), _ptr(i2), len(i2), float(eps), _ptr(out)))
        return out

    def memcpy_h2d(self, d_ptr, arr, offset_bytes=0):
        arr = np.ascontiguousarray(arr)
        check(lib().gst_memcpy_h2d(self._h, C.c_void_p(int(d_ptr) + int(offset_bytes)), _ptr(arr), arr.nbytes))

    def copy_block_dev(self, d_dst, dst_ld, d_src, src_ld, n_rows, n_cols):
        """rows x cols doubles between device arrays with leading dimensions dst_ld / src_ld (doubles); asynchronous"""
        check(lib().gst_copy_block_dev(self._h, C.c_void_p(int(d_dst)), int(dst_ld), C.c_void_p(int(d_src)), int(src_ld), int(n_rows), int(n_cols)))

    def device_malloc(self, nbytes, tracked=False):
        """tracked=True: gst_device_malloc_tracked -- memory written only through this library (or announced with
        `device_touch`), eligible for the resident structural zeros of exact Jacobians (OPT_ANALYTIC_KEEP_ZEROS = 2)."""
        p = C.c_void_p()
        fn = lib().gst_device_malloc_tracked if tracked else lib().gst_device_malloc
        check(fn(self._h, int(nbytes), C.byref(p)))
        return p.value

    def device_free(self, d_ptr):
        check(lib().gst_device_free(self._h, C.c_void_p(int(d_ptr))))

    def device_touch(self, d_ptr, nbytes):
        """the caller wrote device_malloc memory by its own means: what the library remembered about it is forgotten"""
        check(lib().gst_device_touch(self._h, C.c_void_p(int(d_ptr)), int(nbytes)))

    def memcpy_d2h(self, out, d_ptr, offset_bytes=0):
        assert out.flags.c_contiguous
        check(lib().gst_memcpy_d2h(self._h, _ptr(out), C.c_void_p(int(d_ptr) + int(offset_bytes)), out.nbytes))
        return out

    def memcpy_d2h_async(self, out, d_ptr, offset_bytes=0):
        """enqueue the copy behind the plan's work and return; `sync()` completes it (keep `out` alive until then)"""
        assert out.flags.c_contiguous
        check(lib().gst_memcpy_d2h_async(self._h, _ptr(out), C.c_void_p(int(d_ptr) + int(offset_bytes)), out.nbytes))
        return out

    # -- introspection -----------------------------------------------------------------------------------
    def stats(self):
        s = Stats()
        check(lib().gst_get_stats(self._h, C.byref(s)))
        return {f[0]: getattr(s, f[0]) for f in Stats._fields_}

    def fd_queues(self, param_idx, n_queues=1024, handover=1):
        """Estimated work per queue of the persistent FD launch for these columns (host-side; gst_get_fd_queues)."""
        pidx = _i64(param_idx)
        load = np.zeros(int(n_queues), np.int64)
        n_pairs, n_ho = C.c_int32(0), C.c_int32(0)
        check(lib().gst_get_fd_queues(self._h, _ptr(pidx), len(pidx), int(n_queues), int(handover), _ptr(load),
                                      C.byref(n_pairs), C.byref(n_ho)))
        return load, n_pairs.value, n_ho.value

    def fd_work(self, param_idx):
        """gst_get_fd_work: the exact arithmetic of one FD fill of these columns under the kernel's clean/dirty rule."""
        pidx = _i64(param_idx)
        out = np.zeros(8, np.int64)
        check(lib().gst_get_fd_work(self._h, _ptr(pidx), len(pidx), _ptr(out)))
        keys = ("wave_applies_executed", "col_applies_executed", "wave_dots_executed", "col_dots_executed",
                "wave_applies_schedule", "col_applies_schedule", "n_waves", "n_tasks")
        return {k: int(v) for k, v in zip(keys, out)}

    def tile_stats(self):
        """gst_get_tile_stats (host only): the tiles of the D = 16 exact contraction, checked against the application tables."""
        out = (C.c_int64 * 8)()
        check(lib().gst_get_tile_stats(self._h, out))
        keys = ("n_tiles", "tiled_circuits", "segment_slots", "remnant_applications", "inconsistencies", "longest_segment", "n_circuits", "circuits_in_one_tile")
        return dict(zip(keys, [int(v) for v in out]))

    def state_graph(self):
        n = C.c_int64(0)
        check(lib().gst_get_state_graph(self._h, None, None, 0, None, 0, C.byref(n)))
        nc = self.stats()["n_circuits"]
        par = np.empty(max(n.value, 1), np.int32); sym = np.empty(max(n.value, 1), np.int32)
        leaf = np.empty(max(nc, 1), np.int32)
        check(lib().gst_get_state_graph(self._h, _ptr(par), _ptr(sym), n.value, _ptr(leaf), nc, C.byref(n)))
        return par[:n.value], sym[:n.value], leaf[:nc]

    def program(self):
        n = C.c_int64(0)
        check(lib().gst_get_program(self._h, None, 0, C.byref(n), None, 0))
        words = np.empty(max(n.value, 1), np.uint32)
        nt = self.stats()["n_tasks"]
        off = np.empty(nt + 1, np.int64)
        check(lib().gst_get_program(self._h, _ptr(words), n.value, C.byref(n), _ptr(off), nt + 1))
        return words[:n.value], off


def _level_program(self, which=0):
    """gst_get_level_program: the log-depth level program of the forward plan (which = 0) or of the reversed plan (1), built
    on the host.  Returns dict(usable, worthwhile, nv, max_mats, max_stages, n_stages, n_tiles, n_chains, chain_nodes,
    sum_task_depth, n_states, n_tasks, words, ids, task_off, node_parent, node_sym)."""
    nw, ni = C.c_int64(0), C.c_int64(0)
    info = np.zeros(13, np.int64)
    check(lib().gst_get_level_program(self._h, int(which), None, 0, C.byref(nw), None, 0, C.byref(ni), None, 0, None, None, 0, _ptr(info)))
    keys = ("usable", "worthwhile", "nv", "max_mats", "max_stages", "n_stages", "n_tiles", "n_chains", "chain_nodes", "sum_task_depth",
            "n_states", "n_tasks", "n_produced")
    out = {k: int(v) for k, v in zip(keys, info)}
    if not out["usable"]:
        return out
    words = np.empty(max(nw.value, 1), np.int32); ids = np.empty(max(ni.value, 1), np.int32)
    off = np.empty(out["n_tasks"] + 1, np.int64)
    par = np.empty(max(out["n_states"], 1), np.int32); sym = np.empty(max(out["n_states"], 1), np.int32)
    check(lib().gst_get_level_program(self._h, int(which), _ptr(words), len(words), C.byref(nw), _ptr(ids), len(ids), C.byref(ni),
                                      _ptr(off), len(off), _ptr(par), _ptr(sym), len(par), _ptr(info)))
    out.update(words=words[:nw.value], ids=ids[:ni.value], task_off=off, node_parent=par[:out["n_states"]], node_sym=sym[:out["n_states"]])
    return out


Plan.level_program = _level_program


def _dirty_programs(self):
    """(words, prog_off, n_classes) of gst_get_dirty_programs: program (task t, class c) = words[off[t * n_classes + c] : off[... + 1]]"""
    n = C.c_int64(0); nc = C.c_int32(0)
    check(lib().gst_get_dirty_programs(self._h, None, 0, C.byref(n), None, 0, C.byref(nc)))
    words = np.empty(max(n.value, 1), np.uint32)
    n_prog = self.stats()["n_tasks"] * nc.value
    off = np.empty(n_prog + 1, np.int64)
    check(lib().gst_get_dirty_programs(self._h, _ptr(words), n.value, C.byref(n), _ptr(off), n_prog + 1, C.byref(nc)))
    return words[:n.value], off, nc.value


Plan.dirty_programs = _dirty_programs


def sort_circuits(circ_ptr, circ_syms, circ_head=None):
    """(order, lcp) of gst_sort_circuits: the circuits in prefix order of (head, symbols...) and each one's common prefix
    length with its predecessor.  Host-only."""
    ptr = _i64(circ_ptr); syms = _i32(circ_syms)
    n = len(ptr) - 1
    head = None if circ_head is None else _i32(circ_head)
    order = np.empty(n, np.int64); lcp = np.empty(n, np.int64)
    check(lib().gst_sort_circuits(n, _ptr(ptr), _ptr(syms), _ptr(head), _ptr(order), _ptr(lcp)))
    return order, lcp


def circuit_first_use(circ_ptr, circ_syms, n_syms):
    """[n_circuits, n_syms] position of each symbol's first occurrence in each circuit, -1 where it never occurs (host-only)"""
    ptr = _i64(circ_ptr); syms = _i32(circ_syms)
    n = len(ptr) - 1
    out = np.empty((n, max(int(n_syms), 0)), np.int64)
    check(lib().gst_circuit_first_use(n, _ptr(ptr), _ptr(syms), int(n_syms), _ptr(out)))
    return out


class Comm:
    """The device-side exchange between one-process-per-GPU ranks (gst_comm_* of include/gstfwd.h): RCCL over xGMI
    (`TRANSPORT_RCCL`), or intra-node peer writes through HIP IPC (`TRANSPORT_IPC`, also for ranks sharing a GPU).
    Row blocks are (owner rank, first row, number of rows) triples of the assembled row-major array."""

    @staticmethod
    def unique_id(transport=TRANSPORT_RCCL):
        """Rank 0: the rendezvous token (bytes) every rank must pass to `Comm(...)`."""
        buf = (C.c_char * COMM_ID_BYTES)()
        check(lib().gst_comm_get_unique_id(int(transport), buf))
        return bytes(buf)

    def __init__(self, rank, size, uid, device=-1, transport=TRANSPORT_RCCL):
        assert len(uid) == COMM_ID_BYTES
        self._h = C.c_void_p()
        buf = (C.c_char * COMM_ID_BYTES).from_buffer_copy(uid)
        check(lib().gst_comm_create(int(transport), int(device), int(rank), int(size), buf, C.byref(self._h)))
        self.rank, self.size, self.transport = int(rank), int(size), int(transport)

    def close(self):
        if self._h is not None and self._h.value:
            lib().gst_comm_destroy(self._h)
        self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @staticmethod
    def _blocks(blocks):
        owner = _i32([b[0] for b in blocks]); row0 = _i64([b[1] for b in blocks]); rows = _i64([b[2] for b in blocks])
        return owner, row0, rows

    def allgather_rows(self, d_full, row_doubles, blocks, plan=None):
        owner, row0, rows = self._blocks(blocks)
        check(lib().gst_comm_allgather_rows(self._h, None if plan is None else plan._h, C.c_void_p(int(d_full)),
                                            int(row_doubles), len(owner), _ptr(owner), _ptr(row0), _ptr(rows)))

    def gather_rows(self, d_local, d_full, row_doubles, blocks, root=0, plan=None):
        owner, row0, rows = self._blocks(blocks)
        check(lib().gst_comm_gather_rows(self._h, None if plan is None else plan._h,
                                         None if d_local is None else C.c_void_p(int(d_local)),
                                         None if d_full is None else C.c_void_p(int(d_full)), int(row_doubles),
                                         len(owner), _ptr(owner), _ptr(row0), _ptr(rows), int(root)))

    def map_root_buffer(self, d_buf, root=0):
        """Collective (gst_comm_map_root_buffer): a device pointer, valid in THIS process, onto rank `root`'s buffer `d_buf`
        (the root gets its own pointer back): fills write their row blocks straight into the assembled array over xGMI."""
        out = C.c_void_p()
        check(lib().gst_comm_map_root_buffer(self._h, int(root), C.c_void_p(int(d_buf or 0)), C.byref(out)))
        return int(out.value or 0)

    def exchange_blocks(self, d_src, d_dst, blocks, plan=None):
        """blocks = [(src rank, dst rank, src offset, dst offset, count)] in doubles: the same list on every rank, each with
        its own d_src / d_dst (gst_comm_exchange_blocks)"""
        sr = _i32([b[0] for b in blocks]); dr = _i32([b[1] for b in blocks])
        so = _i64([b[2] for b in blocks]); do = _i64([b[3] for b in blocks]); cn = _i64([b[4] for b in blocks])
        check(lib().gst_comm_exchange_blocks(self._h, None if plan is None else plan._h,
                                             None if d_src is None else C.c_void_p(int(d_src)),
                                             None if d_dst is None else C.c_void_p(int(d_dst)),
                                             len(sr), _ptr(sr), _ptr(dr), _ptr(so), _ptr(do), _ptr(cn)))

    def allreduce_sum(self, d_buf, n, plan=None):
        check(lib().gst_comm_allreduce_sum(self._h, None if plan is None else plan._h, C.c_void_p(int(d_buf)), int(n)))

    def barrier(self):
        check(lib().gst_comm_barrier(self._h))

    def sync(self):
        check(lib().gst_comm_sync(self._h))

    def info(self):
        i = CommInfo()
        check(lib().gst_comm_get_info(self._h, C.byref(i)))
        return {"transport": "rccl" if i.transport == TRANSPORT_RCCL else "ipc", "rank": i.rank, "size": i.size,
                "device": i.device, "rccl_version": i.rccl_version, "ipc_opens": i.ipc_opens}
