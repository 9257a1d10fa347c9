"""ctypes binding of libnfmc_hip.so (include/nfmc_hip.h).  No torch types cross the boundary:
tensors are passed as raw device pointers (`tensor.data_ptr()`), work is enqueued on the
caller's current HIP stream.

The library is REQUIRED: importing a sampler on a machine where it cannot be loaded raises
(there is no CPU or eager-PyTorch fallback for the hot path).
"""
import ctypes as C
import os

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('NFMC_LIB', os.path.join(HERE, 'libnfmc_hip.so'))

OK, EINVAL, ESHAPE, EALIGN, EUNSUPPORTED, ESCRATCH = 0, -1, -2, -3, -4, -5
POT_QUADRATIC, POT_FUNNEL = 0, 1
TAG_NOISE, TAG_ACCEPT, TAG_LATENT, TAG_JUMP = 0, 1, 2, 3
CNT_ACCEPTED, CNT_ATTEMPTED, CNT_NONFINITE, CNT_WORDS = 0, 1, 2, 4
MAX_STEPS_PER_CALL = 512
IMH_PARALLEL_MAX_STEPS = 65536

c_fp = C.c_void_p  # all device pointers travel as void*
NFMC_ABI_VERSION = 4   # include/nfmc_hip.h


class NfmcPotential(C.Structure):
    _fields_ = [('kind', C.c_int32), ('reserved', C.c_int32), ('a', c_fp), ('b', c_fp),
                ('a_scalar', C.c_float), ('b_scalar', C.c_float)]


class NfmcRng(C.Structure):
    _fields_ = [('seed', C.c_uint64), ('chain_offset', C.c_uint64), ('step0', C.c_uint32), ('rounds', C.c_uint32),
                ('replay_normals', c_fp), ('replay_uniforms', c_fp)]


class NfmcStats(C.Structure):
    _fields_ = [('sum_x', c_fp), ('sum_x2', c_fp), ('counters', c_fp), ('scratch', c_fp),
                ('scratch_bytes', C.c_int64), ('defer', C.c_int32), ('tail_slot', C.c_int32)]


class NfmcRealNVP(C.Structure):
    _fields_ = [('d', C.c_int32), ('n_coupling', C.c_int32), ('n_hidden', C.c_int32), ('n_hidden_layers', C.c_int32),
                ('min_scale', C.c_float), ('n_bins', C.c_int32),
                ('ea0_log_scale', c_fp), ('ea0_shift', c_fp), ('ea1_log_scale', c_fp), ('ea1_shift', c_fp),
                ('weights', c_fp), ('layer_stride', C.c_int64), ('spline_bound', C.c_float), ('reserved', C.c_int32),
                ('scratch', c_fp), ('scratch_bytes', C.c_int64)]


class NfmcSampleStore(C.Structure):
    _fields_ = [('base', c_fp), ('stride', C.c_int32), ('countdown', C.c_int32), ('ring_rows', C.c_int32), ('row', C.c_int32)]


class NfmcJumpTail(C.Structure):
    _fields_ = [('flow', NfmcRealNVP), ('adjusted', C.c_int32), ('reserved', C.c_int32), ('counters', c_fp),
                ('replay_latent', c_fp), ('replay_uniform', c_fp), ('mask_out', c_fp), ('log_ratio_out', c_fp)]


class NfmcTune(C.Structure):
    _fields_ = [('state', c_fp), ('inv_mass_diag', c_fp), ('tune_step_size', C.c_int32), ('tune_inv_mass_diag', C.c_int32),
                ('every', C.c_int32), ('reserved', C.c_int32)]


TUNE_STEP_SIZE, TUNE_LOG_SMOOTH, TUNE_ERROR_SUM, TUNE_ITERATION, TUNE_ANCHOR, TUNE_LOG_RAW = 0, 1, 2, 3, 4, 5
TUNE_TARGET, TUNE_KAPPA, TUNE_GAMMA, TUNE_IMD_ADJUSTMENT, TUNE_TICKET, TUNE_WORDS = 6, 7, 8, 9, 10, 12


class NfmcMalaArgs(C.Structure):
    _fields_ = [('x', c_fp), ('n', C.c_int64), ('d', C.c_int32), ('n_steps', C.c_int32),
                ('step_size', C.c_float), ('adjust', C.c_int32), ('inv_mass_diag', c_fp),
                ('pot', NfmcPotential), ('rng', NfmcRng), ('stats', NfmcStats),
                ('samples', NfmcSampleStore), ('masks_out', c_fp), ('log_ratio_out', c_fp), ('jump', C.POINTER(NfmcJumpTail)),
                ('tune', NfmcTune)]


class NfmcHmcArgs(C.Structure):
    _fields_ = [('x', c_fp), ('n', C.c_int64), ('d', C.c_int32), ('n_steps', C.c_int32),
                ('step_size', C.c_float), ('n_leapfrog', C.c_int32), ('adjust', C.c_int32), ('reserved', C.c_int32),
                ('inv_mass_diag', c_fp), ('pot', NfmcPotential), ('rng', NfmcRng), ('stats', NfmcStats),
                ('samples', NfmcSampleStore), ('masks_out', c_fp), ('log_ratio_out', c_fp), ('jump', C.POINTER(NfmcJumpTail)),
                ('tune', NfmcTune)]


class NfmcFlowMhArgs(C.Structure):
    _fields_ = [('x', c_fp), ('logq', c_fp), ('n', C.c_int64), ('n_steps', C.c_int32), ('logq_cached', C.c_int32),
                ('adjusted', C.c_int32), ('reserved', C.c_int32), ('flow', NfmcRealNVP), ('pot', NfmcPotential),
                ('rng', NfmcRng), ('stats', NfmcStats), ('samples', NfmcSampleStore), ('masks_out', c_fp),
                ('log_ratio_out', c_fp)]


class NfmcNeutraHmcArgs(C.Structure):
    _fields_ = [('z', c_fp), ('n', C.c_int64), ('n_steps', C.c_int32), ('n_leapfrog', C.c_int32),
                ('step_size', C.c_float), ('adjust', C.c_int32), ('inv_mass_diag', c_fp),
                ('flow', NfmcRealNVP), ('pot', NfmcPotential), ('rng', NfmcRng), ('stats', NfmcStats),
                ('samples', NfmcSampleStore), ('masks_out', c_fp), ('log_ratio_out', c_fp), ('scratch', c_fp),
                ('scratch_bytes', C.c_int64)]


class NfmcSelectArgs(C.Structure):
    _fields_ = [('x', c_fp), ('x_prime', c_fp), ('n', C.c_int64), ('d', C.c_int32), ('n_carry', C.c_int32),
                ('log_ratio', c_fp), ('uniforms', c_fp), ('carry', c_fp * 2), ('carry_prime', c_fp * 2),
                ('rng', NfmcRng), ('rng_tag', C.c_int32), ('reserved', C.c_int32), ('stats', NfmcStats),
                ('mask_out', c_fp)]


class NfmcAdamW(C.Structure):
    _fields_ = [('lr', C.c_float), ('beta1', C.c_float), ('beta2', C.c_float), ('eps', C.c_float),
                ('weight_decay', C.c_float), ('step', C.c_int32)]


class NfmcFlowFit(C.Structure):
    _fields_ = [('flow', NfmcRealNVP), ('params', c_fp), ('adam_m', c_fp), ('adam_v', c_fp), ('n_params', C.c_int64),
                ('ea_off', C.c_int64), ('partial', c_fp), ('partial_floats', C.c_int64), ('status', c_fp),
                ('x_val', c_fp), ('n_val', C.c_int64), ('params_prev', c_fp), ('best', c_fp), ('run_state', c_fp),
                ('scratch', c_fp), ('scratch_bytes', C.c_int64)]


class NfmcBlobPiece(C.Structure):
    _fields_ = [('param', c_fp), ('vec_off', C.c_int64), ('rows', C.c_int32), ('cols', C.c_int32),
                ('vec_row_stride', C.c_int32), ('vec_col_stride', C.c_int32)]


class NfmcFitControl(C.Structure):
    _fields_ = [('n_epochs', C.c_int32), ('early_stopping', C.c_int32), ('early_stopping_threshold', C.c_int32),
                ('keep_best_weights', C.c_int32), ('skip_nonfinite', C.c_int32), ('reserved', C.c_int32)]


# indices into NfmcFlowFit.run_state (include/nfmc_hip.h: NFMC_FIT_*)
FIT_BEST_LOSS, FIT_SINCE_BEST, FIT_APPLIED, FIT_STOPPED, FIT_DIVERGED, FIT_LAST_LOSS, FIT_LAST_VAL, FIT_BOOKED = range(8)
FIT_STATE_FLOATS = 8


class NfmcLimits(C.Structure):
    _fields_ = [('abi_version', C.c_int32), ('max_d_sampler', C.c_int32), ('max_d_flow', C.c_int32),
                ('max_hidden_valu', C.c_int32), ('max_hidden', C.c_int32), ('max_steps_per_call', C.c_int32)]


# every symbol include/nfmc_hip.h declares: (name, restype, argtypes)
SYMBOLS = [
    ('nfmc_stats_scratch_bytes', C.c_int64, [C.c_int32]),
    ('nfmc_tune_state_doubles', C.c_int64, [C.c_int32]),
    ('nfmc_mala_steps_f32', C.c_int, [C.POINTER(NfmcMalaArgs), c_fp]),
    ('nfmc_hmc_steps_f32', C.c_int, [C.POINTER(NfmcHmcArgs), c_fp]),
    ('nfmc_realnvp_padded_hidden', C.c_int32, [C.c_int32]),
    ('nfmc_realnvp_layer_floats', C.c_int64, [C.c_int32, C.c_int32, C.c_int32]),
    ('nfmc_coupling_layer_floats', C.c_int64, [C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    ('nfmc_flow_scratch_bytes', C.c_int64, [C.POINTER(NfmcRealNVP), C.c_int64, C.c_int32]),
    ('nfmc_realnvp_forward_f32', C.c_int, [C.POINTER(NfmcRealNVP), c_fp, C.c_int64, c_fp, c_fp, c_fp, c_fp]),
    ('nfmc_realnvp_inverse_f32', C.c_int, [C.POINTER(NfmcRealNVP), c_fp, C.c_int64, c_fp, c_fp, c_fp,
                                           C.POINTER(NfmcRng), c_fp]),
    ('nfmc_flow_mh_steps_f32', C.c_int, [C.POINTER(NfmcFlowMhArgs), c_fp]),
    ('nfmc_flow_mh_supported_f32', C.c_int, [C.POINTER(NfmcFlowMhArgs)]),
    ('nfmc_imh_parallel_supported_f32', C.c_int, [C.POINTER(NfmcFlowMhArgs)]),
    ('nfmc_imh_parallel_work_bytes', C.c_int64, [C.c_int64, C.c_int32, C.c_int32]),
    ('nfmc_imh_parallel_f32', C.c_int, [C.POINTER(NfmcFlowMhArgs), c_fp, C.c_int64, c_fp]),
    ('nfmc_neutra_scratch_bytes', C.c_int64, [C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    ('nfmc_neutra_hmc_steps_f32', C.c_int, [C.POINTER(NfmcNeutraHmcArgs), c_fp]),
    ('nfmc_neutra_potential_grad_f32', C.c_int, [C.POINTER(NfmcRealNVP), C.POINTER(NfmcPotential), c_fp, C.c_int64,
                                                 c_fp, c_fp, c_fp]),
    ('nfmc_mh_accept_select_f32', C.c_int, [C.POINTER(NfmcSelectArgs), c_fp]),
    ('nfmc_langevin_propose_f32', C.c_int, [c_fp, c_fp, c_fp, C.c_float, C.c_int64, C.c_int32, C.POINTER(NfmcRng),
                                            c_fp, c_fp]),
    ('nfmc_langevin_log_ratio_f32', C.c_int, [c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, C.c_float, C.c_int64,
                                              C.c_int32, c_fp, c_fp]),
    ('nfmc_moments_update_f32', C.c_int, [c_fp, C.c_int64, C.c_int32, C.POINTER(NfmcStats), c_fp]),
    ('nfmc_stats_fold_f32', C.c_int, [C.POINTER(NfmcStats), C.c_int32, C.c_uint64, c_fp, C.c_uint64, c_fp]),
    ('nfmc_philox_normals_f32', C.c_int, [C.POINTER(NfmcRng), C.c_int32, C.c_int64, C.c_int32, c_fp, c_fp]),
    ('nfmc_philox_uniforms_f32', C.c_int, [C.POINTER(NfmcRng), C.c_int32, C.c_int64, c_fp, c_fp]),
    ('nfmc_flow_fit_supported_f32', C.c_int, [C.POINTER(NfmcRealNVP)]),
    ('nfmc_flow_fit_partial_floats', C.c_int64, [C.c_int64, C.c_int64]),
    ('nfmc_flow_fit_workspace', C.c_int64, [C.POINTER(NfmcRealNVP), C.c_int64, C.c_int64, C.c_int64, C.POINTER(C.c_int64)]),
    ('nfmc_flow_fit_step_f32', C.c_int, [C.POINTER(NfmcFlowFit), c_fp, C.c_int64, C.POINTER(NfmcAdamW), c_fp]),
    ('nfmc_flow_variational_fit_step_f32', C.c_int, [C.POINTER(NfmcFlowFit), C.POINTER(NfmcPotential), c_fp, C.c_int64,
                                                    C.POINTER(NfmcAdamW), c_fp]),
    ('nfmc_flow_fit_epochs_f32', C.c_int, [C.POINTER(NfmcFlowFit), C.POINTER(NfmcPotential), c_fp, C.c_int64, C.c_int64,
                                          C.POINTER(NfmcAdamW), C.POINTER(NfmcFitControl), C.c_int32, C.c_int32, c_fp]),
    ('nfmc_flow_blob_copy_f32', C.c_int, [c_fp, C.POINTER(NfmcBlobPiece), C.c_int32, C.c_int32, c_fp]),
    ('nfmc_rows_sample_f32', C.c_int, [c_fp, C.c_int64, C.c_int32, C.c_uint64, C.c_int64, c_fp, C.c_int64, c_fp, c_fp]),
    ('nfmc_rows_sample_index', C.c_int64, [C.c_int64, C.c_uint64, C.c_int64]),
    ('nfmc_limits', C.c_int, [C.POINTER(NfmcLimits)]),
    ('nfmc_error_string', C.c_char_p, [C.c_int]),
    ('nfmc_build_digest', C.c_char_p, []),
]

_lib = None


def lib():
    """Load the library (once).  Raises if it is missing: the hot path has no fallback."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                'nfmc_amd: %s is missing. Build it with `python -m nfmc_amd.build` (needs hipcc); the MI355X '
                'path has no CPU/eager fallback.' % LIB_PATH)
        handle = C.CDLL(LIB_PATH)
        # the structs below are ABI version NFMC_ABI_VERSION: a stale or side-built library (NFMC_LIB, tools/ab_*.sh
        # variants) under them would read pointers at the wrong offsets -- refuse it before binding anything else
        probe = getattr(handle, 'nfmc_limits', None)
        if probe is None:
            raise RuntimeError('nfmc_amd: %s does not export nfmc_limits (not an nfmc_hip library?)' % LIB_PATH)
        probe.restype, probe.argtypes = C.c_int, [C.POINTER(NfmcLimits)]
        lim = NfmcLimits()
        rc = probe(C.byref(lim))
        if rc != 0 or lim.abi_version != NFMC_ABI_VERSION:
            raise RuntimeError('nfmc_amd: %s has ABI version %d (status %d); this package binds version %d. Rebuild it '
                               'with `python -m nfmc_amd.build --force`.' % (LIB_PATH, lim.abi_version, rc, NFMC_ABI_VERSION))
        for name, restype, argtypes in SYMBOLS:
            fn = getattr(handle, name)
            fn.restype = restype
            fn.argtypes = argtypes
        _lib = handle
    return _lib


class NfmcArgumentError(ValueError):
    """A negative status of the library (`code`): ValueError is the reference's only error channel."""

    def __init__(self, message, code):
        super().__init__(message)
        self.code = code

    @property
    def no_kernel(self):
        """The request is valid but no kernel covers it (shape / width beyond a fused kernel): callers that have a
        composed path take it."""
        return self.code in (EUNSUPPORTED, ESHAPE)


def check(rc, what):
    """Map a return code to the reference's error channels: ValueError for argument errors
    (the reference's only error type, langevin.py:111), RuntimeError for HIP errors."""
    if rc == 0:
        return
    msg = lib().nfmc_error_string(rc).decode()
    if rc < 0:
        raise NfmcArgumentError('%s: %s (code %d)' % (what, msg, rc), rc)
    raise RuntimeError('%s: HIP error %d: %s' % (what, rc, msg))


def require_gpu():
    if not torch.cuda.is_available():
        raise RuntimeError('nfmc_amd needs a ROCm GPU (MI355X / gfx950); torch.cuda.is_available() is False. '
                           'There is no CPU fallback for the hot path.')
    return torch.device('cuda', torch.cuda.current_device())


def ptr(t, dtype=torch.float32):
    """Device pointer of a contiguous CUDA tensor of `dtype` (None -> NULL)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise ValueError('expected a GPU tensor')
    if t.dtype != dtype:
        raise ValueError('expected dtype %s, got %s' % (dtype, t.dtype))
    if not t.is_contiguous():
        raise ValueError('expected a contiguous tensor')
    return C.c_void_p(t.data_ptr())


_raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', None)


def stream():
    """torch's current stream on the current device as a hipStream_t (the raw binding: the public
    `torch.cuda.current_stream()` builds a Stream object, ~10 us per call, once per kernel launch)."""
    if _raw_stream is not None:
        return C.c_void_p(_raw_stream(torch.cuda.current_device()))
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def build_digest():
    """Source digest the loaded library was built from (nfmc_build_digest)."""
    return lib().nfmc_build_digest().decode()


def limits():
    out = NfmcLimits()
    check(lib().nfmc_limits(C.byref(out)), 'nfmc_limits')
    return out


class DeviceStats:
    """Device-resident accumulators behind NfmcStats (fp64 sums, u64 counters, scratch slabs).

    Deferred mode (`struct(defer=True, attempted=...)`): the sampling kernels only add their per-workgroup
    partials to the zeroed scratch and ONE fold (nfmc_stats_fold_f32) runs when the totals are read, instead
    of one fold kernel per launch.  `sum_x`, `sum_x2`, `counters` and `jump_counters` fold lazily, and a
    non-deferred `struct()` folds first, so the two modes can be mixed freely within a run.
    """

    def __init__(self, d, device):
        self.d = d
        self.device = device
        # one backing allocation (8-byte words), so the totals come back in ONE device-to-host copy (host_totals):
        # each separate .cpu() is ~30 us of latency at the end of every sample() call
        self._buf = torch.zeros(2 * d + 2 * CNT_WORDS, dtype=torch.float64, device=device)
        self._sum_x = self._buf[:d]
        self._sum_x2 = self._buf[d:2 * d]
        self._counters = self._buf[2 * d:2 * d + CNT_WORDS].view(torch.int64)
        self._jump_counters = self._buf[2 * d + CNT_WORDS:].view(torch.int64)
        nbytes = int(lib().nfmc_stats_scratch_bytes(d))
        self.scratch = torch.zeros(nbytes // 8, dtype=torch.float64, device=device)
        self._pending = False
        self._attempted = 0
        self._jump_attempted = 0

    def _raw(self, defer=0, tail_slot=0, counters=None):
        return NfmcStats(ptr(self._sum_x, torch.float64), ptr(self._sum_x2, torch.float64),
                         ptr(self._counters if counters is None else counters, torch.int64),
                         ptr(self.scratch, torch.float64), self.scratch.numel() * 8, defer, tail_slot)

    def struct(self, defer=False, attempted=0, jump=False, jump_attempted=0):
        """NfmcStats for one launch.  defer=True: `attempted` = chain-transitions this launch attempts (the fold
        books them; `jump_attempted` those of a jump fused behind it); jump=True: the launch is a jump (accept
        counts go to `jump_counters`).  NFMC_STATS_DEFER=0 turns deferral off (A/B measurements)."""
        if not defer or os.environ.get('NFMC_STATS_DEFER', '1') == '0':
            self.fold()
            return self._raw(counters=self._jump_counters if jump else None)
        self._pending = True
        if jump:
            self._jump_attempted += int(attempted)
        else:
            self._attempted += int(attempted)
            self._jump_attempted += int(jump_attempted)
        return self._raw(1, 2 if jump else 0)

    def fold(self):
        if not self._pending:
            return
        st = self._raw()
        check(lib().nfmc_stats_fold_f32(C.byref(st), self.d, self._attempted, ptr(self._jump_counters, torch.int64),
                                        self._jump_attempted, stream()), 'nfmc_stats_fold_f32')
        self._pending = False
        self._attempted = 0
        self._jump_attempted = 0

    @property
    def sum_x(self):
        self.fold()
        return self._sum_x

    @property
    def sum_x2(self):
        self.fold()
        return self._sum_x2

    @property
    def counters(self):
        self.fold()
        return self._counters

    @property
    def jump_counters(self):
        self.fold()
        return self._jump_counters

    def host_totals(self):
        """(sum_x, sum_x2, counters, jump_counters) as CPU tensors from one copy (synchronises the stream)."""
        self.fold()
        h = self._buf.cpu()
        d = self.d
        return (h[:d], h[d:2 * d], h[2 * d:2 * d + CNT_WORDS].view(torch.int64),
                h[2 * d + CNT_WORDS:].view(torch.int64))

    def zero_(self):
        self.fold()
        self._buf.zero_()


def dense_store(t):
    """NfmcSampleStore that keeps every transition of one call in the rows of `t` (k, n, d), in order (None: nothing)."""
    if t is None:
        return NfmcSampleStore(None, 1, 0, 1, 0)
    return NfmcSampleStore(ptr(t), 1, 0, int(t.shape[0]), 0)


def store_struct(samples, k):
    """NfmcSampleStore for a call that offers k transitions: `samples` is None (keep nothing), a
    containers.DeviceSampleStore (thinning / bounded window decided here, before the launch), or a dense (k, n, d)
    device tensor (every transition kept, in order)."""
    if samples is None:
        return dense_store(None)
    if hasattr(samples, 'struct'):
        return samples.struct(k)
    if int(samples.shape[0]) != int(k):
        raise ValueError('dense sample view has %d rows for %d transitions' % (samples.shape[0], k))
    return dense_store(samples)


def null_stats():
    return NfmcStats(None, None, None, None, 0, 0, 0)


def make_rng(seed, chain_offset, step0, replay_normals=None, replay_uniforms=None, rounds=0):
    """rounds: 0 / 10 = Philox4x32-10 (default stream), 7 = the opt-in Philox4x32-7 stream."""
    return NfmcRng(int(seed) & 0xFFFFFFFFFFFFFFFF, int(chain_offset), int(step0) & 0xFFFFFFFF, int(rounds),
                   ptr(replay_normals), ptr(replay_uniforms))
