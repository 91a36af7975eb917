"""iq-tree_amd -- Python plumbing over the MI355X likelihood engine.

The product is two shared libraries built in-tree by `make` (see __graft_entry__.build):
  lib/libiqhip.so   HIP kernels + the C ABI of include/iqhip.h (the drop-in boundary)
  lib/libiqhost.so  C++ host mirror of the PhyloTree slice that drives the kernels
This package only loads them with ctypes; there is no Python or CPU implementation of the
likelihood path here, and loading fails loudly when the libraries are missing.

The directory name has a hyphen, so it is imported through `__graft_entry__.load_package()`
(or tests/conftest.py) under the module name `iqtree_amd`.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# IQHIP_LIB_DIR: alternative build directory (A/B timing of two builds on one GPU box)
LIB_DIR = os.environ.get("IQHIP_LIB_DIR") or os.path.join(_HERE, "lib")
REPO_ROOT = os.path.dirname(_HERE)

# every symbol include/iqhip.h declares (kept in sync by tests/test_abi.py)
IQHIP_SYMBOLS = [
    "iqhip_last_error", "iqhip_abi_version", "iqhip_device_count", "iqhip_create", "iqhip_destroy",
    "iqhip_set_stream", "iqhip_reserve", "iqhip_release", "iqhip_rekey", "iqhip_set_alignment",
    "iqhip_set_ptn_freq", "iqhip_set_ptn_invar", "iqhip_set_ascertainment", "iqhip_set_model", "iqhip_update_partials",
    "iqhip_branch_lnl", "iqhip_traverse_lnl", "iqhip_compute_theta", "iqhip_derv",
    "iqhip_lnl_from_theta", "iqhip_newton_branch", "iqhip_optimize_branch", "iqhip_bind_result_buffer", "iqhip_result_device_ptr",
    "iqhip_result_capacity", "iqhip_traverse_lnl_async", "iqhip_derv_async", "iqhip_result_read",
    "iqhip_synchronize", "iqhip_fetch_scale_num", "iqhip_fetch_pattern_lh", "iqhip_fetch_partial",
    "iqhip_fetch_theta", "iqhip_upload_partial", "iqhip_timing_enable", "iqhip_timing_read",
    "iqhip_fetch_pattern_lh_scaled", "iqhip_set_boot_samples", "iqhip_rell", "iqhip_rell_async",
    "iqhip_set_mixture_model", "iqhip_pattern_lh_cat", "iqhip_optimize_branch_batch",
    "iqhip_create_sharded", "iqhip_num_shards", "iqhip_shard_range", "iqhip_comm_unique_id", "iqhip_comm_init_rank",
    "iqhip_comm_size", "iqhip_update_partials_async", "iqhip_lnl_from_theta_async",
    "iqhip_newton_host_init", "iqhip_newton_host_update", "iqhip_newton_host_result",
    "iqhip_debug_create_planner", "iqhip_debug_plan", "iqhip_timing_plan_bytes", "iqhip_timing_collective_read", "iqhip_optimize_sweep", "iqhip_debug_cherry_tables",
]


class NodeOp(C.Structure):
    """struct iqhip_node_op (include/iqhip.h)."""
    _fields_ = [("dst_key", C.c_uint64), ("left_key", C.c_uint64), ("right_key", C.c_uint64),
                ("left_leaf", C.c_int32), ("right_leaf", C.c_int32),
                ("left_len", C.c_double), ("right_len", C.c_double), ("flags", C.c_uint32), ("_pad", C.c_uint32)]


class BranchEnd(C.Structure):
    """struct iqhip_branch_end (include/iqhip.h)."""
    _fields_ = [("key", C.c_uint64), ("leaf", C.c_int32), ("_pad", C.c_int32)]


def leaf_end(leaf):
    return BranchEnd(0, int(leaf), 0)


def key_end(key):
    return BranchEnd(int(key), -1, 0)


ALLREDUCE_HOOK = C.CFUNCTYPE(None, C.c_void_p, C.c_int, C.c_void_p)

_libs = {}


def _load(name):
    if name in _libs:
        return _libs[name]
    path = os.path.join(LIB_DIR, name)
    if not os.path.exists(path):
        raise ImportError(
            f"{path} is missing: the HIP extension has not been built "
            "(run `python -c 'import __graft_entry__ as g; g.build()'` or `make`). "
            "There is no CPU fallback for the likelihood path.")
    lib = C.CDLL(path, mode=C.RTLD_GLOBAL)
    _libs[name] = lib
    return lib


def libiqhip():
    lib = _load("libiqhip.so")
    if getattr(lib, "_iq_typed", False):
        return lib
    dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int16)
    vp = C.c_void_p
    lib.iqhip_last_error.restype = C.c_char_p
    lib.iqhip_create.argtypes = [C.POINTER(vp), C.c_int, C.c_int, C.c_int, C.c_int64, C.c_int]
    lib.iqhip_destroy.argtypes = [vp]
    lib.iqhip_destroy.restype = None
    lib.iqhip_set_stream.argtypes = [vp, vp]
    lib.iqhip_reserve.argtypes = [vp, C.c_int]
    lib.iqhip_release.argtypes = [vp, C.c_uint64]
    lib.iqhip_rekey.argtypes = [vp, C.c_uint64, C.c_uint64]
    lib.iqhip_set_alignment.argtypes = [vp, C.POINTER(C.c_uint8), dp, dp]
    lib.iqhip_set_ptn_freq.argtypes = [vp, dp]
    lib.iqhip_set_ptn_invar.argtypes = [vp, dp]
    lib.iqhip_set_ascertainment.argtypes = [vp, C.c_int64, C.c_double]
    lib.iqhip_set_model.argtypes = [vp, dp, dp, dp, dp, dp, C.c_int, dp]
    lib.iqhip_update_partials.argtypes = [vp, C.POINTER(NodeOp), C.c_int, dp]
    lib.iqhip_branch_lnl.argtypes = [vp, BranchEnd, BranchEnd, C.c_double, dp]
    lib.iqhip_traverse_lnl.argtypes = [vp, C.POINTER(NodeOp), C.c_int, BranchEnd, BranchEnd,
                                       C.c_double, dp, dp]
    lib.iqhip_compute_theta.argtypes = [vp, BranchEnd, BranchEnd]
    lib.iqhip_derv.argtypes = [vp, C.c_double, dp, dp]
    lib.iqhip_lnl_from_theta.argtypes = [vp, C.c_double, dp]
    lib.iqhip_optimize_branch.argtypes = [vp, C.POINTER(NodeOp), C.c_int, BranchEnd, BranchEnd, C.c_double,
                                          C.c_double, C.c_double, C.c_double, C.c_int, dp, dp, dp,
                                          C.POINTER(C.c_int)]
    lib.iqhip_newton_branch.argtypes = [vp, C.c_double, C.c_double, C.c_double, C.c_double, C.c_int, dp, dp,
                                        C.POINTER(C.c_int)]
    lib.iqhip_bind_result_buffer.argtypes = [vp, vp, C.c_int]
    lib.iqhip_result_device_ptr.argtypes = [vp]
    lib.iqhip_result_device_ptr.restype = vp
    lib.iqhip_result_capacity.argtypes = [vp]
    lib.iqhip_traverse_lnl_async.argtypes = [vp, C.POINTER(NodeOp), C.c_int, BranchEnd, BranchEnd,
                                             C.c_double]
    lib.iqhip_derv_async.argtypes = [vp, C.c_double]
    lib.iqhip_result_read.argtypes = [vp, dp, C.c_int]
    lib.iqhip_synchronize.argtypes = [vp]
    lib.iqhip_fetch_scale_num.argtypes = [vp, C.c_uint64, ip]
    lib.iqhip_fetch_pattern_lh.argtypes = [vp, dp]
    lib.iqhip_fetch_partial.argtypes = [vp, C.c_uint64, dp]
    lib.iqhip_fetch_theta.argtypes = [vp, dp]
    lib.iqhip_upload_partial.argtypes = [vp, C.c_uint64, dp, ip]
    lib.iqhip_fetch_pattern_lh_scaled.argtypes = [vp, BranchEnd, BranchEnd, dp]
    lib.iqhip_set_boot_samples.argtypes = [vp, C.POINTER(C.c_float), C.c_int]
    lib.iqhip_rell.argtypes = [vp, BranchEnd, BranchEnd, dp]
    lib.iqhip_rell_async.argtypes = [vp, BranchEnd, BranchEnd]
    lib.iqhip_create_sharded.argtypes = [C.POINTER(vp), C.POINTER(C.c_int), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int64,
                                         C.c_int]
    lib.iqhip_num_shards.argtypes = [vp]
    lib.iqhip_shard_range.argtypes = [vp, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int)]
    lib.iqhip_comm_unique_id.argtypes = [vp]
    lib.iqhip_comm_init_rank.argtypes = [vp, C.c_int, C.c_int, vp]
    lib.iqhip_comm_size.argtypes = [vp]
    lib.iqhip_update_partials_async.argtypes = [vp, C.POINTER(NodeOp), C.c_int]
    lib.iqhip_lnl_from_theta_async.argtypes = [vp, C.c_double]
    lib.iqhip_newton_host_init.argtypes = [vp, C.c_double, C.c_double, C.c_double, C.c_double, C.c_int, dp]
    lib.iqhip_newton_host_update.argtypes = [vp, C.c_double, C.c_double, dp, C.POINTER(C.c_int)]
    lib.iqhip_newton_host_result.argtypes = [vp, dp, dp, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    lib.iqhip_timing_enable.argtypes = [vp, C.c_int]
    lib.iqhip_timing_read.argtypes = [vp, dp, C.POINTER(C.c_int64), C.c_int]
    lib.iqhip_debug_create_planner.argtypes = [C.POINTER(vp), C.c_int, C.c_int, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int]
    lib.iqhip_timing_plan_bytes.argtypes = [vp, dp, dp]
    lib.iqhip_timing_collective_read.argtypes = [vp, dp, C.POINTER(C.c_int64), C.c_int]
    lib.iqhip_debug_cherry_tables.argtypes = [vp, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    lib.iqhip_debug_plan.argtypes = [vp, C.POINTER(NodeOp), C.c_int]
    lib._iq_typed = True
    return lib


def libiqhost():
    libiqhip()  # dependency, loaded RTLD_GLOBAL first
    lib = _load("libiqhost.so")
    if getattr(lib, "_iq_typed", False):
        return lib
    dp, vp = C.POINTER(C.c_double), C.c_void_p
    lib.iqhost_last_error.restype = C.c_char_p
    lib.iqhost_create.argtypes = [C.POINTER(vp), C.c_char_p, C.POINTER(C.c_char_p), C.c_int]
    lib.iqhost_destroy.argtypes = [vp]
    lib.iqhost_destroy.restype = None
    lib.iqhost_set_alignment.argtypes = [vp, C.c_int, C.c_int, C.c_int64, C.POINTER(C.c_uint8), dp, dp]
    lib.iqhost_set_ascertainment.argtypes = [vp, C.c_int64, C.c_double]
    lib.iqhost_set_ptn_freq.argtypes = [vp, dp]
    lib.iqhost_set_ptn_invar.argtypes = [vp, dp]
    lib.iqhost_set_model.argtypes = [vp, C.c_int, dp, dp, dp, dp, dp]
    lib.iqhost_set_mixture_model.argtypes = [vp, C.c_int, C.c_int, C.POINTER(C.c_int), dp, dp, dp, dp, dp]
    lib.iqhost_set_mem_mode.argtypes = [vp, C.c_int]
    lib.iqhost_set_kernel.argtypes = [vp, C.c_int]
    lib.iqhost_attach_engine.argtypes = [vp, C.c_int]
    lib.iqhost_attach_engine_sharded.argtypes = [vp, C.POINTER(C.c_int), C.c_int, C.c_int]
    lib.iqhost_attach_comm.argtypes = [vp, C.c_int, C.c_int, C.c_char_p]
    lib.iqhost_set_dry_run.argtypes = [vp, C.c_int]
    lib.iqhost_set_heavy_first.argtypes = [vp, C.c_int]
    lib.iqhost_set_device_newton.argtypes = [vp, C.c_int]
    lib.iqhost_set_device_sweep.argtypes = [vp, C.c_int]
    lib.iqhost_num_derv_calls.argtypes = [vp]
    lib.iqhost_num_derv_calls.restype = C.c_long
    lib.iqhost_engine.argtypes = [vp]
    lib.iqhost_engine.restype = vp
    lib.iqhost_set_allreduce_hook.argtypes = [vp, ALLREDUCE_HOOK, vp]
    for f in ("iqhost_num_nodes", "iqhost_num_leaves", "iqhost_root", "iqhost_state_unknown"):
        getattr(lib, f).argtypes = [vp]
    lib.iqhost_tip_partial_lh.argtypes = [vp, dp]
    lib.iqhost_neighbors.argtypes = [vp, C.c_int, C.POINTER(C.c_int), dp, C.c_int]
    lib.iqhost_set_branch_length.argtypes = [vp, C.c_int, C.c_int, C.c_double, C.c_int]
    lib.iqhost_neighbor_info.argtypes = [vp, C.c_int, C.c_int, C.POINTER(C.c_int),
                                         C.POINTER(C.c_uint64), dp, dp]
    lib.iqhost_initialize_all_partial_lh.argtypes = [vp]
    lib.iqhost_clear_all_partial_lh.argtypes = [vp]
    lib.iqhost_compute_likelihood.argtypes = [vp, dp, dp]
    lib.iqhost_clear_and_compute_likelihood.argtypes = [vp, dp]
    lib.iqhost_current_branch.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    lib.iqhost_compute_partial.argtypes = [vp, C.c_int, C.c_int]
    lib.iqhost_compute_branch.argtypes = [vp, C.c_int, C.c_int, dp]
    lib.iqhost_compute_derv.argtypes = [vp, C.c_int, C.c_int, dp, dp]
    lib.iqhost_reset_theta.argtypes = [vp]
    lib.iqhost_compute_from_buffer.argtypes = [vp, dp]
    lib.iqhost_optimize_one_branch.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, dp]
    lib.iqhost_optimize_all_branches.argtypes = [vp, C.c_int, C.c_double, C.c_int, dp]
    lib.iqhost_set_branch_bounds.argtypes = [vp, C.c_double, C.c_double]
    lib.iqhost_nni_for_branch.argtypes = [vp, C.c_int, C.c_int, C.c_int, dp]
    lib.iqhost_evaluate_nnis_batch.argtypes = [vp, C.POINTER(C.c_int), dp, C.c_int, C.POINTER(C.c_int)]
    lib.iqhost_compute_all_partial_lh.argtypes = [vp]
    lib.iqhost_evaluate_nnis5_batch.argtypes = [vp, C.POINTER(C.c_int), dp, C.c_int, C.POINTER(C.c_int)]
    lib.iqhost_tree_string.argtypes = [vp, C.c_char_p, C.c_int]
    lib.iqhost_fetch_scale_num.argtypes = [vp, C.c_int, C.c_int, C.POINTER(C.c_int16)]
    lib.iqhost_fetch_partial.argtypes = [vp, C.c_int, C.c_int, dp]
    lib.iqhost_fetch_pattern_lh.argtypes = [vp, dp]
    lib.iqhost_compute_pattern_likelihood.argtypes = [vp, dp]
    lib.iqhost_compute_pattern_lh_cat.argtypes = [vp, dp]
    lib.iqhost_set_boot_samples.argtypes = [vp, C.POINTER(C.c_float), C.c_int]
    lib.iqhost_compute_rell.argtypes = [vp, dp, C.c_int]
    lib.iqhost_last_plan.argtypes = [vp, C.POINTER(C.c_int), dp, C.POINTER(C.c_uint64), C.c_int]
    lib.iqhost_num_partial_lh_computations.argtypes = [vp]
    lib.iqhost_num_partial_lh_computations.restype = C.c_long
    lib.iqhost_num_submissions.argtypes = [vp]
    lib.iqhost_num_submissions.restype = C.c_long
    # model / alignment producers (iq-tree_amd/host/iqmodel_c.cpp)
    ip, u8p = C.POINTER(C.c_int), C.POINTER(C.c_uint8)
    lib.iqmodel_last_error.restype = C.c_char_p
    lib.iqmodel_decompose.argtypes = [dp, dp, C.c_int, C.c_int, dp, dp, dp]
    lib.iqmodel_gamma_rates.argtypes = [C.c_double, C.c_int, C.c_int, C.c_double, dp]
    for f, n in (("iqmodel_ln_gamma", 1), ("iqmodel_incomplete_gamma", 2), ("iqmodel_point_normal", 1),
                 ("iqmodel_point_chi2", 2)):
        getattr(lib, f).argtypes = [C.c_double] * n
        getattr(lib, f).restype = C.c_double
    lib.iqmodel_genetic_code.argtypes = [C.c_int]
    lib.iqmodel_genetic_code.restype = C.c_char_p
    lib.iqaln_read.argtypes = [C.POINTER(vp), C.c_char_p, C.c_char_p, C.c_char_p]
    lib.iqaln_destroy.argtypes = [vp]
    lib.iqaln_destroy.restype = None
    for f in ("iqaln_nseq", "iqaln_nsite", "iqaln_npattern", "iqaln_nstates", "iqaln_seq_type", "iqaln_state_unknown"):
        getattr(lib, f).argtypes = [vp]
    lib.iqaln_frac_const_sites.argtypes = [vp]
    lib.iqaln_frac_const_sites.restype = C.c_double
    lib.iqaln_seq_name.argtypes = [vp, C.c_int]
    lib.iqaln_seq_name.restype = C.c_char_p
    lib.iqaln_append_unobserved.argtypes = [vp, ip]
    lib.iqaln_get.argtypes = [vp, u8p, dp, ip, ip]
    lib.iqaln_ptn_invar.argtypes = [vp, C.c_double, dp, dp]
    lib.iqaln_state_freq.argtypes = [vp, dp]
    lib.iqaln_codon_freq.argtypes = [vp, C.c_int, dp, dp]
    lib.iqaln_write_sitelh.argtypes = [vp, C.c_char_p, dp]
    lib.iqmodel_build.argtypes = [vp, C.c_char_p, ip, dp, ip, dp, dp, dp, dp, dp, dp]
    lib._iq_typed = True
    return lib


def _dptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


class HostError(RuntimeError):
    pass


LK_EIGEN, LK_EIGEN_SSE, LK_EIGEN_HIP = 0, 1, 2
REDUCE_RCCL, REDUCE_HOST = 0, 1


class NewtonStateMachine:
    """Optimization::minimizeNewton as the engine's step-at-a-time state machine (iqhip_newton_host_*)."""

    def __init__(self, xguess, x1, x2, xacc, max_steps):
        self.lib = libiqhip()
        self.buf = C.create_string_buffer(128)
        x = C.c_double()
        if self.lib.iqhip_newton_host_init(self.buf, xguess, x1, x2, xacc, max_steps, C.byref(x)) != 0:
            raise HostError(self.lib.iqhip_last_error().decode())
        self.x, self.done = x.value, False

    def update(self, df_sum, ddf_sum):
        x, d = C.c_double(), C.c_int()
        self.lib.iqhip_newton_host_update(self.buf, df_sum, ddf_sum, C.byref(x), C.byref(d))
        self.x, self.done = x.value, bool(d.value)
        return self.x

    def result(self):
        optx, d2l, n, st = C.c_double(), C.c_double(), C.c_int(), C.c_int()
        if self.lib.iqhip_newton_host_result(self.buf, C.byref(optx), C.byref(d2l), C.byref(n), C.byref(st)) != 0:
            raise HostError(self.lib.iqhip_last_error().decode())
        return optx.value, d2l.value, n.value, st.value


def comm_unique_id():
    """ncclGetUniqueId through the engine library (rank 0; broadcast the 128 bytes to the other ranks)."""
    lib = libiqhip()
    buf = C.create_string_buffer(128)
    if lib.iqhip_comm_unique_id(buf) != 0:
        raise HostError(lib.iqhip_last_error().decode())
    return buf.raw
LM_PER_NODE, LM_ALL_BRANCH = 0, 1
SEQ_DNA, SEQ_PROTEIN, SEQ_CODON, SEQ_OTHER = 0, 1, 2, 3


class PhyloTree:
    """Thin OO view of iqhost::PhyloTree (iq-tree_amd/host/phylo_host.h)."""

    def __init__(self, newick, names=None):
        self.lib = libiqhost()
        self.h = C.c_void_p()
        if names:
            arr = (C.c_char_p * len(names))(*[n.encode() for n in names])
            rc = self.lib.iqhost_create(C.byref(self.h), newick.encode(), arr, len(names))
        else:
            rc = self.lib.iqhost_create(C.byref(self.h), newick.encode(), None, 0)
        self._chk(rc)
        self.nptn = 0
        self.block = 0

    def _chk(self, rc):
        if rc != 0:
            raise HostError(self.lib.iqhost_last_error().decode())

    def close(self):
        if self.h:
            self.lib.iqhost_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- inputs
    def set_alignment(self, nstates, seq_type, states, ptn_freq, ptn_invar=None):
        states = np.ascontiguousarray(states, dtype=np.uint8)
        self.nptn = states.shape[1]
        self.nstates = nstates
        f = np.ascontiguousarray(ptn_freq, dtype=np.float64)
        iv = np.zeros(self.nptn) if ptn_invar is None else np.ascontiguousarray(ptn_invar, dtype=np.float64)
        self._chk(self.lib.iqhost_set_alignment(self.h, nstates, seq_type, self.nptn,
                                                states.ctypes.data_as(C.POINTER(C.c_uint8)),
                                                _dptr(f), _dptr(iv)))

    def set_ptn_freq(self, ptn_freq):
        f = np.ascontiguousarray(ptn_freq, dtype=np.float64)
        assert f.size == self.nptn
        self._chk(self.lib.iqhost_set_ptn_freq(self.h, _dptr(f)))

    def set_ptn_invar(self, ptn_invar):
        f = np.ascontiguousarray(ptn_invar, dtype=np.float64)
        assert f.size == self.nptn
        self._chk(self.lib.iqhost_set_ptn_invar(self.h, _dptr(f)))

    def set_ascertainment(self, n_unobserved, nsites):
        """+ASC: the last n_unobserved patterns of set_alignment are the unobserved constant patterns."""
        self._chk(self.lib.iqhost_set_ascertainment(self.h, int(n_unobserved), float(nsites)))

    def set_model(self, model):
        """model: object with eval, evec, inv_evec, rates, props (see synth.Model); a mixture model also
        has nclass > 1 and cat_class (component -> eigen-system), its arrays concatenated per class."""
        a = [np.ascontiguousarray(x, dtype=np.float64) for x in
             (model.eval, model.evec, model.inv_evec, model.rates, model.props)]
        self.ncat = len(a[3])
        self.block = self.nstates * self.ncat
        nclass = int(getattr(model, "nclass", 1))
        if nclass > 1:
            cls = np.ascontiguousarray(model.cat_class, dtype=np.int32)
            self._chk(self.lib.iqhost_set_mixture_model(self.h, nclass, self.ncat, cls.ctypes.data_as(C.POINTER(C.c_int)),
                                                        *[_dptr(x) for x in a]))
        else:
            self._chk(self.lib.iqhost_set_model(self.h, self.ncat, *[_dptr(x) for x in a]))

    def set_mem_mode(self, lm):
        self._chk(self.lib.iqhost_set_mem_mode(self.h, lm))

    def set_likelihood_kernel(self, lk):
        self._chk(self.lib.iqhost_set_kernel(self.h, lk))

    def attach_engine(self, device=0):
        self._chk(self.lib.iqhost_attach_engine(self.h, device))

    def attach_engine_sharded(self, devices, reduce_mode=REDUCE_RCCL):
        """one engine handle over several GPUs of this process (iqhip_create_sharded)"""
        arr = (C.c_int * len(devices))(*[int(d) for d in devices])
        self._chk(self.lib.iqhost_attach_engine_sharded(self.h, arr, len(devices), int(reduce_mode)))

    def attach_comm(self, nranks, rank, unique_id):
        """one process per GPU: join this tree's engine to the other ranks' (iqhip_comm_init_rank);
        unique_id = the 128 bytes rank 0 got from comm_unique_id(), distributed by the caller"""
        assert len(unique_id) == 128
        self._chk(self.lib.iqhost_attach_comm(self.h, int(nranks), int(rank), bytes(unique_id)))

    def set_device_newton(self, on=True):
        self._chk(self.lib.iqhost_set_device_newton(self.h, int(on)))

    def set_device_sweep(self, on=True):
        """optimize_all_branches: every sweep as ONE engine submission (iqhip_optimize_sweep) instead of one per branch"""
        self._chk(self.lib.iqhost_set_device_sweep(self.h, int(on)))

    @property
    def num_derv_calls(self):
        return self.lib.iqhost_num_derv_calls(self.h)

    def set_heavy_first(self, on=True):
        self._chk(self.lib.iqhost_set_heavy_first(self.h, int(on)))

    def set_dry_run(self, on=True):
        self._chk(self.lib.iqhost_set_dry_run(self.h, int(on)))

    @property
    def engine(self):
        return self.lib.iqhost_engine(self.h)

    def set_allreduce_hook(self, fn):
        """fn(device_ptr:int, ndoubles:int) all-reduces the device result vector in place."""
        if fn is None:
            self._hook = None
            self._chk(self.lib.iqhost_set_allreduce_hook(self.h, ALLREDUCE_HOOK(0), None))
            return
        self._hook = ALLREDUCE_HOOK(lambda ptr, n, ctx: fn(ptr, n))
        self._chk(self.lib.iqhost_set_allreduce_hook(self.h, self._hook, None))

    # ---- structure
    @property
    def num_nodes(self):
        return self.lib.iqhost_num_nodes(self.h)

    @property
    def num_leaves(self):
        return self.lib.iqhost_num_leaves(self.h)

    @property
    def root(self):
        return self.lib.iqhost_root(self.h)

    @property
    def state_unknown(self):
        return self.lib.iqhost_state_unknown(self.h)

    def tip_partial_lh(self):
        out = np.zeros((self.state_unknown + 1) * self.nstates)
        self.lib.iqhost_tip_partial_lh(self.h, _dptr(out))
        return out.reshape(self.state_unknown + 1, self.nstates)

    def neighbors(self, node):
        ids = (C.c_int * 8)()
        lens = (C.c_double * 8)()
        d = self.lib.iqhost_neighbors(self.h, node, ids, lens, 8)
        return [(ids[i], lens[i]) for i in range(d)]

    def set_branch_length(self, a, b, length, clear_reverse=True):
        self._chk(self.lib.iqhost_set_branch_length(self.h, a, b, length, int(clear_reverse)))

    def neighbor_info(self, frm, to):
        comp, key = C.c_int(), C.c_uint64()
        sf, ln = C.c_double(), C.c_double()
        self._chk(self.lib.iqhost_neighbor_info(self.h, frm, to, C.byref(comp), C.byref(key),
                                                C.byref(sf), C.byref(ln)))
        return dict(computed=comp.value, key=key.value, lh_scale_factor=sf.value, length=ln.value)

    def tree_string(self):
        buf = C.create_string_buffer(1 << 20)
        self.lib.iqhost_tree_string(self.h, buf, len(buf))
        return buf.value.decode()

    # ---- the reference's call sequence
    def initialize_all_partial_lh(self):
        self._chk(self.lib.iqhost_initialize_all_partial_lh(self.h))

    def clear_all_partial_lh(self):
        self._chk(self.lib.iqhost_clear_all_partial_lh(self.h))

    def compute_likelihood(self, want_pattern_lh=False):
        lnl = C.c_double()
        if want_pattern_lh:
            plh = np.zeros(self.nptn)
            self._chk(self.lib.iqhost_compute_likelihood(self.h, C.byref(lnl), _dptr(plh)))
            return lnl.value, plh
        self._chk(self.lib.iqhost_compute_likelihood(self.h, C.byref(lnl), None))
        return lnl.value

    def clear_and_compute_likelihood(self):
        """clearAllPartialLH(); computeLikelihood() -- the model optimisers' target function"""
        lnl = C.c_double()
        self._chk(self.lib.iqhost_clear_and_compute_likelihood(self.h, C.byref(lnl)))
        return lnl.value

    def current_branch(self):
        a, b = C.c_int(), C.c_int()
        if self.lib.iqhost_current_branch(self.h, C.byref(a), C.byref(b)):
            return None
        return a.value, b.value

    def compute_partial_likelihood(self, dad, node):
        self._chk(self.lib.iqhost_compute_partial(self.h, dad, node))

    def compute_likelihood_branch(self, dad, node):
        lnl = C.c_double()
        self._chk(self.lib.iqhost_compute_branch(self.h, dad, node, C.byref(lnl)))
        return lnl.value

    def compute_likelihood_derv(self, dad, node):
        df, ddf = C.c_double(), C.c_double()
        self._chk(self.lib.iqhost_compute_derv(self.h, dad, node, C.byref(df), C.byref(ddf)))
        return df.value, ddf.value

    def reset_theta(self):
        self._chk(self.lib.iqhost_reset_theta(self.h))

    def compute_likelihood_from_buffer(self):
        lnl = C.c_double()
        self._chk(self.lib.iqhost_compute_from_buffer(self.h, C.byref(lnl)))
        return lnl.value

    def optimize_one_branch(self, a, b, clear_lh=True, max_nr_step=100):
        ln = C.c_double()
        self._chk(self.lib.iqhost_optimize_one_branch(self.h, a, b, int(clear_lh), max_nr_step, C.byref(ln)))
        return ln.value

    def optimize_all_branches(self, iterations=100, tolerance=0.001, max_nr_step=100):
        lnl = C.c_double()
        self._chk(self.lib.iqhost_optimize_all_branches(self.h, iterations, tolerance, max_nr_step, C.byref(lnl)))
        return lnl.value

    def nni_for_branch(self, a, b, nni5=False):
        """getBestNNIForBran: [(newloglh, swapped subtree at a, swapped subtree at b, [newLen...]), x2]"""
        out = np.zeros(16)
        self._chk(self.lib.iqhost_nni_for_branch(self.h, a, b, int(nni5), _dptr(out)))
        return [(out[8 * c], int(out[8 * c + 1]), int(out[8 * c + 2]), list(out[8 * c + 3:8 * c + 8])) for c in range(2)]

    def evaluate_nnis_batch(self):
        """all nni1 candidates (2 per internal branch) in one submission ->
        list of dict(node1, node2, node1_nei, node2_nei, new_len, newloglh)"""
        cap = 4 * self.num_nodes
        ids = np.zeros(4 * cap, dtype=np.int32)
        vals = np.zeros(2 * cap)
        n = C.c_int()
        self._chk(self.lib.iqhost_evaluate_nnis_batch(self.h, ids.ctypes.data_as(C.POINTER(C.c_int)), _dptr(vals), cap,
                                                      C.byref(n)))
        return [dict(node1=int(ids[4 * k]), node2=int(ids[4 * k + 1]), node1_nei=int(ids[4 * k + 2]),
                     node2_nei=int(ids[4 * k + 3]), new_len=float(vals[2 * k]), newloglh=float(vals[2 * k + 1]))
                for k in range(n.value)]

    def evaluate_nnis5_batch(self):
        """all nni5 candidates in ten submissions -> list of dict(node1, node2, node1_nei, node2_nei, new_lens[5], newloglh)"""
        cap = 4 * self.num_nodes
        ids = np.zeros(4 * cap, dtype=np.int32)
        vals = np.zeros(6 * cap)
        n = C.c_int()
        self._chk(self.lib.iqhost_evaluate_nnis5_batch(self.h, ids.ctypes.data_as(C.POINTER(C.c_int)), _dptr(vals), cap,
                                                       C.byref(n)))
        return [dict(node1=int(ids[4 * k]), node2=int(ids[4 * k + 1]), node1_nei=int(ids[4 * k + 2]),
                     node2_nei=int(ids[4 * k + 3]), new_lens=[float(x) for x in vals[6 * k:6 * k + 5]],
                     newloglh=float(vals[6 * k + 5])) for k in range(n.value)]

    def compute_all_partial_lh(self):
        self._chk(self.lib.iqhost_compute_all_partial_lh(self.h))

    def set_branch_bounds(self, lo, hi):
        self._chk(self.lib.iqhost_set_branch_bounds(self.h, lo, hi))

    # ---- host views
    def fetch_scale_num(self, frm, to):
        out = np.zeros(self.nptn, dtype=np.int16)
        self._chk(self.lib.iqhost_fetch_scale_num(self.h, frm, to, out.ctypes.data_as(C.POINTER(C.c_int16))))
        return out

    def fetch_partial(self, frm, to):
        out = np.zeros(self.nptn * self.block)
        self._chk(self.lib.iqhost_fetch_partial(self.h, frm, to, _dptr(out)))
        return out.reshape(self.nptn, self.block)

    def fetch_pattern_lh(self):
        out = np.zeros(self.nptn)
        self._chk(self.lib.iqhost_fetch_pattern_lh(self.h, _dptr(out)))
        return out

    def compute_pattern_likelihood(self):
        """PhyloTree::computePatternLikelihood: per-pattern lnL with the scaling events put back."""
        out = np.zeros(self.nptn)
        self._chk(self.lib.iqhost_compute_pattern_likelihood(self.h, _dptr(out)))
        return out

    def compute_pattern_lh_cat(self):
        """_pattern_lh_cat[nptn, ncat] of the current branch (unscaled category likelihoods)."""
        out = np.zeros((self.nptn, self.ncat))
        self._chk(self.lib.iqhost_compute_pattern_lh_cat(self.h, _dptr(out)))
        return out

    def set_boot_samples(self, samples):
        """UFBoot boot_samples: float32 [nsamples, nptn] pattern weights, uploaded once."""
        s = np.ascontiguousarray(samples, dtype=np.float32)
        assert s.ndim == 2 and s.shape[1] == self.nptn
        self._nboot = s.shape[0]
        self._chk(self.lib.iqhost_set_boot_samples(self.h, s.ctypes.data_as(C.POINTER(C.c_float)), s.shape[0]))

    def compute_rell(self):
        out = np.zeros(self._nboot)
        self._chk(self.lib.iqhost_compute_rell(self.h, _dptr(out), out.size))
        return out

    def last_plan(self):
        cap = 4 * self.num_nodes + 8
        ints = (C.c_int * (7 * cap))()
        lens = (C.c_double * (2 * cap))()
        keys = (C.c_uint64 * (3 * cap))()
        n = self.lib.iqhost_last_plan(self.h, ints, lens, keys, cap)
        plan = []
        for k in range(n):
            plan.append(dict(dst=(ints[7 * k], ints[7 * k + 1]), left=ints[7 * k + 2], right=ints[7 * k + 3],
                             left_leaf=ints[7 * k + 4], right_leaf=ints[7 * k + 5], flags=ints[7 * k + 6],
                             left_len=lens[2 * k], right_len=lens[2 * k + 1],
                             dst_key=keys[3 * k], left_key=keys[3 * k + 1], right_key=keys[3 * k + 2]))
        return plan

    @property
    def num_partial_lh_computations(self):
        return self.lib.iqhost_num_partial_lh_computations(self.h)

    @property
    def num_submissions(self):
        return self.lib.iqhost_num_submissions(self.h)


# ---------------------------------------------------------------------------------------------
# model / alignment producers: C++ in iq-tree_amd/host/{model,alignment}_host.cpp
# ---------------------------------------------------------------------------------------------

class AttrDict(dict):
    """dict whose keys also read as attributes (so it can stand where a synth.Model is expected)."""
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)


def _mchk(lib, rc):
    if rc != 0:
        raise HostError(lib.iqmodel_last_error().decode())


def decompose_rate_matrix(rate_matrix, state_freq, ignore_state_freq=False):
    """ModelGTR::decomposeRateMatrix -> dict(eval, evec, inv_evec) (row-major, reference layout)."""
    lib = libiqhost()
    r = np.ascontiguousarray(rate_matrix, dtype=np.float64)
    f = np.ascontiguousarray(state_freq, dtype=np.float64)
    n = f.size
    ev, U, Ui = np.zeros(n), np.zeros((n, n)), np.zeros((n, n))
    _mchk(lib, lib.iqmodel_decompose(_dptr(r), _dptr(f), n, int(ignore_state_freq), _dptr(ev), _dptr(U), _dptr(Ui)))
    return AttrDict(eval=ev, evec=U, inv_evec=Ui)


def gamma_rates(shape, ncat, median=False, p_invar=0.0):
    """RateGamma::computeRates."""
    lib = libiqhost()
    out = np.zeros(ncat)
    _mchk(lib, lib.iqmodel_gamma_rates(float(shape), int(ncat), int(median), float(p_invar), _dptr(out)))
    return out


class Alignment:
    """iqhost::Alignment: PHYLIP/FASTA reader + site->pattern compression."""

    def __init__(self, filename=None, content=None, seq_type=""):
        self.lib = libiqhost()
        self.h = C.c_void_p()
        rc = self.lib.iqaln_read(C.byref(self.h), filename.encode() if filename else None,
                                 content.encode() if content is not None else None, seq_type.encode())
        _mchk(self.lib, rc)
        self.n_unobserved = 0

    def __del__(self):
        if getattr(self, "h", None):
            self.lib.iqaln_destroy(self.h)
            self.h = None

    nseq = property(lambda s: s.lib.iqaln_nseq(s.h))
    nsite = property(lambda s: s.lib.iqaln_nsite(s.h))
    npattern = property(lambda s: s.lib.iqaln_npattern(s.h))
    nstates = property(lambda s: s.lib.iqaln_nstates(s.h))
    seq_type = property(lambda s: s.lib.iqaln_seq_type(s.h))
    state_unknown = property(lambda s: s.lib.iqaln_state_unknown(s.h))
    frac_const_sites = property(lambda s: s.lib.iqaln_frac_const_sites(s.h))

    @property
    def seq_names(self):
        return [self.lib.iqaln_seq_name(self.h, i).decode() for i in range(self.nseq)]

    def append_unobserved_const_patterns(self):
        n = C.c_int()
        _mchk(self.lib, self.lib.iqaln_append_unobserved(self.h, C.byref(n)))
        self.n_unobserved = n.value
        return n.value

    def arrays(self):
        """-> states[nseq, nptn] uint8, ptn_freq[nptn], site_pattern[nsite], const_char[nptn] (-1 = not constant)"""
        ns, npt, nsite = self.nseq, self.npattern, self.nsite
        st = np.zeros((ns, npt), dtype=np.uint8)
        fr = np.zeros(npt)
        sp = np.zeros(nsite, dtype=np.int32)
        cc = np.zeros(npt, dtype=np.int32)
        _mchk(self.lib, self.lib.iqaln_get(self.h, st.ctypes.data_as(C.POINTER(C.c_uint8)), _dptr(fr),
                                           sp.ctypes.data_as(C.POINTER(C.c_int)), cc.ctypes.data_as(C.POINTER(C.c_int))))
        return st, fr, sp, cc

    def ptn_invar(self, p_invar, state_freq):
        out = np.zeros(self.npattern)
        f = np.ascontiguousarray(state_freq, dtype=np.float64)
        _mchk(self.lib, self.lib.iqaln_ptn_invar(self.h, float(p_invar), _dptr(f), _dptr(out)))
        return out

    def state_freq(self):
        out = np.zeros(self.nstates)
        _mchk(self.lib, self.lib.iqaln_state_freq(self.h, _dptr(out)))
        return out

    def codon_freq(self, f3x4=False):
        out, nt = np.zeros(self.nstates), np.zeros(12)
        _mchk(self.lib, self.lib.iqaln_codon_freq(self.h, int(f3x4), _dptr(out), _dptr(nt)))
        return out, nt

    def write_sitelh(self, filename, pattern_lh):
        p = np.ascontiguousarray(pattern_lh, dtype=np.float64)
        _mchk(self.lib, self.lib.iqaln_write_sitelh(self.h, filename.encode(), _dptr(p)))

    def build_model(self, model_string):
        """-m string -> the dict PhyloTree.set_model() takes (+ state_freq, p_invar, asc)."""
        n = self.nstates
        ncat, asc, pinv = C.c_int(), C.c_int(), C.c_double()
        ev, U, Ui, fr = np.zeros(n), np.zeros((n, n)), np.zeros((n, n)), np.zeros(n)
        rates, props = np.zeros(64), np.zeros(64)
        _mchk(self.lib, self.lib.iqmodel_build(self.h, model_string.encode(), C.byref(ncat), C.byref(pinv), C.byref(asc),
                                               _dptr(ev), _dptr(U), _dptr(Ui), _dptr(fr), _dptr(rates), _dptr(props)))
        k = ncat.value
        return AttrDict(nstates=n, ncat=k, eval=ev, evec=U, inv_evec=Ui, state_freq=fr, rates=rates[:k].copy(),
                    props=props[:k].copy(), p_invar=pinv.value, asc=bool(asc.value))
