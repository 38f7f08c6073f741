"""ctypes binding of ``libsbc_hip.so`` (C ABI declared in ``include/sbc_hip.h``).

There is no fallback: if the shared library is missing or a call fails, an exception is raised -- the
product path never routes around the HIP kernels.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('SBC_LIB_PATH') or os.path.join(_HERE, 'libsbc_hip.so')   # env override: A/B builds (tools/)
ABI_VERSION = 14

EXPORTS = ('sbc_abi_version', 'sbc_set_persistent_cus', 'sbc_last_error', 'sbc_device_count', 'sbc_op_launch', 'sbc_plan_create',
           'sbc_plan_run', 'sbc_plan_set_persistent_cus', 'sbc_plan_destroy', 'sbc_plan_profile', 'sbc_plan_profile_read',
           'sbc_pack_conv_weight', 'sbc_pack_conv_weight_winograd',
           'sbc_pack_conv_weight_split', 'sbc_pack_conv_weight_winograd_split',
           'sbc_pack_conv_weight_f16', 'sbc_pack_conv_weight_winograd_f16',
           'sbc_pack_conv_weight_f16x2', 'sbc_pack_conv_weight_winograd_f16x2', 'sbc_pack_conv_weight_pooled_f16x2', 'sbc_range_flag',
           'sbc_f16x2_calibration_input', 'sbc_f16x2_calibrate', 'sbc_debug_philox4x32', 'sbc_debug_complex_normal',
           'sbc_score_create', 'sbc_score_buffers', 'sbc_score_ops', 'sbc_score_level_source', 'sbc_score_forward',
           'sbc_score_destroy', 'sbc_wgrad_scratch_floats')


class SbcError(RuntimeError):
    pass


class sbc_op(C.Structure):
    _fields_ = [('kind', C.c_int32), ('flags', C.c_int32),
                ('B', C.c_int32), ('H', C.c_int32), ('W', C.c_int32),
                ('cin', C.c_int32), ('cout', C.c_int32), ('ksize', C.c_int32), ('dil', C.c_int32),
                ('up_h', C.c_int32), ('up_w', C.c_int32), ('tag', C.c_int32),
                ('in_', C.c_void_p), ('out', C.c_void_p), ('weight', C.c_void_p), ('bias', C.c_void_p),
                ('stats', C.c_void_p), ('res1', C.c_void_p), ('res2', C.c_void_p), ('up', C.c_void_p),
                ('ext', C.c_void_p), ('weight_wino', C.c_void_p), ('weight_split', C.c_void_p),
                ('weight_wino_split', C.c_void_p),
                # training operators (ABI 7)
                ('grad', C.c_void_p), ('aux', C.c_void_p), ('wgrad', C.c_void_p), ('bgrad', C.c_void_p),
                ('weight2_split', C.c_void_p),
                ('calib', C.c_void_p),                # ABI 11: NULL (set by sbc_f16x2_calibrate on its own copies)
                ('bias2', C.c_void_p), ('norm2', C.c_void_p),   # ABI 12: RES_BLOCK (second convolution's bias, second norm's alpha|gamma|beta)
                ('weight2_wino_split', C.c_void_p),             # ABI 13: calibration only (every form of a layer gets the same scale)
                # ABI 14: launch lanes of a plan (0 = the run stream), event ids recorded behind / waited for in front of the record
                ('lane', C.c_int32), ('signal', C.c_int32), ('wait', C.c_int32 * 2)]


class sbc_endconv(C.Structure):
    _fields_ = [('sigmas', C.c_void_p), ('labels', C.c_void_p), ('sigma_of_step', C.c_void_p),
                ('step', C.c_void_p)]


class sbc_langevin(C.Structure):
    _fields_ = [('X', C.c_void_p), ('score', C.c_void_p), ('P', C.c_void_p), ('p_index', C.c_void_p),
                ('Y', C.c_void_p), ('Htrue', C.c_void_p), ('h_index', C.c_void_p), ('sched', C.c_void_p),
                ('group', C.c_void_p), ('noise', C.c_void_p), ('nmse', C.c_void_p), ('step', C.c_void_p),
                ('traj_id', C.c_void_p), ('meas_scale', C.c_void_p), ('seed', C.c_uint64),
                ('n_steps', C.c_int32), ('Nt', C.c_int32), ('Nr', C.c_int32), ('Np', C.c_int32)]


class sbc_dsm(C.Structure):
    _fields_ = [('sigmas', C.c_void_p), ('labels', C.c_void_p), ('noise', C.c_void_p), ('sample_id', C.c_void_p),
                ('seed', C.c_uint64), ('offset', C.c_int32), ('anneal_power', C.c_float), ('step', C.c_void_p),
                ('grad_scale', C.c_float)]


class sbc_adam(C.Structure):
    _fields_ = [('n', C.c_int64), ('lr', C.c_double), ('beta1', C.c_double), ('beta2', C.c_double), ('eps', C.c_double),
                ('ema_mu', C.c_double), ('step', C.c_void_p)]


class sbc_chain(C.Structure):
    _fields_ = [('n_blocks', C.c_int32), ('type', C.c_int32 * 6), ('dil', C.c_int32 * 6), ('w1', C.c_void_p * 6), ('w2', C.c_void_p * 6),
                ('w1_wino', C.c_void_p * 6), ('w2_wino', C.c_void_p * 6), ('w3', C.c_void_p * 6), ('bias1', C.c_void_p * 6),
                ('bias2', C.c_void_p * 6), ('bias3', C.c_void_p * 6), ('norm1', C.c_void_p * 6), ('norm2', C.c_void_p * 6)]


class sbc_tensor_ref(C.Structure):
    _fields_ = [('name', C.c_char_p), ('data', C.c_void_p), ('numel', C.c_int64)]


class sbc_score_desc(C.Structure):
    _fields_ = [('ngf', C.c_int32), ('channels', C.c_int32), ('nt', C.c_int32), ('nr', C.c_int32), ('batch', C.c_int32),
                ('conv_mode', C.c_int32), ('sigmas', C.c_void_p), ('num_classes', C.c_int32), ('flags', C.c_int32)]


_lib = None


def lib():
    """Load the library once; raise ``SbcError`` with a build hint if it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise SbcError('%s not found: build it with `make -C %s` (or `python -c "import __graft_entry__ as g; '
                       'g.build()"`); there is no CPU fallback' % (LIB_PATH, os.path.join(_HERE, 'csrc')))
    h = C.CDLL(LIB_PATH)
    h.sbc_abi_version.restype = C.c_int
    h.sbc_last_error.restype = C.c_char_p
    h.sbc_device_count.restype = C.c_int
    h.sbc_op_launch.argtypes = [C.POINTER(sbc_op), C.c_void_p]
    h.sbc_plan_create.argtypes = [C.POINTER(sbc_op), C.c_int32, C.POINTER(C.c_void_p)]
    h.sbc_plan_run.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32]
    h.sbc_plan_set_persistent_cus.argtypes = [C.c_void_p, C.c_int32]
    h.sbc_plan_destroy.argtypes = [C.c_void_p]
    h.sbc_plan_destroy.restype = None
    h.sbc_plan_profile.argtypes = [C.c_void_p, C.c_int32]
    h.sbc_plan_profile_read.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int64)]
    h.sbc_pack_conv_weight.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]
    h.sbc_pack_conv_weight_winograd.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]
    h.sbc_pack_conv_weight_split.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]
    h.sbc_pack_conv_weight_winograd_split.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]
    h.sbc_pack_conv_weight_f16.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]
    h.sbc_pack_conv_weight_winograd_f16.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]
    h.sbc_pack_conv_weight_f16x2.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]
    h.sbc_pack_conv_weight_winograd_f16x2.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]
    h.sbc_pack_conv_weight_pooled_f16x2.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]
    h.sbc_range_flag.argtypes = [C.POINTER(C.c_int32), C.c_int32]
    h.sbc_f16x2_calibration_input.argtypes = [C.c_void_p, C.c_int64]
    h.sbc_f16x2_calibrate.argtypes = [C.POINTER(sbc_op), C.c_int32, C.c_void_p]
    h.sbc_debug_philox4x32.argtypes = [C.c_void_p, C.c_int32, C.c_void_p]
    h.sbc_debug_complex_normal.argtypes = [C.c_uint64, C.c_int64, C.c_int32, C.c_int32, C.c_void_p]
    h.sbc_score_create.argtypes = [C.POINTER(sbc_score_desc), C.POINTER(sbc_tensor_ref), C.c_int32, C.POINTER(C.c_void_p)]
    h.sbc_score_buffers.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]
    h.sbc_score_ops.argtypes = [C.c_void_p, C.POINTER(C.POINTER(sbc_op)), C.POINTER(C.c_int32)]
    h.sbc_score_level_source.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    h.sbc_score_forward.argtypes = [C.c_void_p, C.c_void_p]
    h.sbc_score_destroy.argtypes = [C.c_void_p]
    h.sbc_score_destroy.restype = None
    h.sbc_wgrad_scratch_floats.argtypes = [C.c_int32] * 6
    h.sbc_wgrad_scratch_floats.restype = C.c_int64
    if h.sbc_abi_version() != ABI_VERSION:
        raise SbcError('libsbc_hip.so ABI %d != expected %d' % (h.sbc_abi_version(), ABI_VERSION))
    _lib = h
    return h


def check(rc):
    if rc != 0:
        raise SbcError('libsbc_hip: %s (status %d)' % (lib().sbc_last_error().decode(), rc))


RANGE_OVERFLOW, RANGE_UNDERFLOW, RANGE_ELU = 1, 2, 4


def range_flag(reset=True, device=None):
    """``sbc_range_flag`` of ``device`` (default: the current one): the bit set the f16x2 convolutions collected since the last
    reset -- ``RANGE_OVERFLOW``: a staged activation left the fp16 range; ``RANGE_UNDERFLOW``: a region of an input was so far
    below its layer's calibrated scale that the two-term split there is no longer fp32-class.  Waits for every stream of the
    device."""
    v = C.c_int32()
    if device is not None:
        import torch
        with torch.cuda.device(device):
            check(lib().sbc_range_flag(C.byref(v), 1 if reset else 0))
    else:
        check(lib().sbc_range_flag(C.byref(v), 1 if reset else 0))
    return v.value


def describe_range(bits):
    what = []
    if bits & RANGE_OVERFLOW:
        what.append('an activation left the fp16 range (|x| * act_scale >= 16000)')
    if bits & RANGE_UNDERFLOW:
        what.append('a region of an input stayed below 2^-6 of its layer\'s scale (denormal low terms)')
    if bits & RANGE_ELU:
        what.append('a fused RCU launch met inputs below 2^-4, where its exp(x) - 1 form of ELU loses relative accuracy')
    return '; '.join(what)


def check_range(what='run', device=None):
    """Raise if the f16x2 kernels flagged this run: its numbers are not fp32-class (``driver.run_trajectories`` re-runs such a
    batch in ``bf16x3`` instead of raising)."""
    bits = range_flag(True, device)
    if bits:
        raise SbcError('%s: %s in conv_mode f16x2; results are not fp32-class -- use conv_mode bf16x3 for this input'
                       % (what, describe_range(bits)))


def calibration_input(n):
    """The library's fixed calibration pattern (``sbc_f16x2_calibration_input``) as a float32 numpy array of ``n`` values."""
    import numpy as np
    x = np.empty(int(n), np.float32)
    check(lib().sbc_f16x2_calibration_input(x.ctypes.data_as(C.c_void_p), int(n)))
    return x


def calibrate_f16x2(ops, stream):
    """``sbc_f16x2_calibrate`` over a list of bound ``sbc_op`` records (synchronises ``stream``)."""
    arr = (sbc_op * len(ops))(*ops)
    check(lib().sbc_f16x2_calibrate(arr, len(ops), C.c_void_p(stream)))


class Plan:
    """Owner of an ``sbc_plan*``.  ``keepalive`` holds whatever owns the device buffers the ops point to."""

    def __init__(self, ops, keepalive=None):
        arr = (sbc_op * len(ops))(*ops)
        handle = C.c_void_p()
        check(lib().sbc_plan_create(arr, len(ops), C.byref(handle)))
        self._h = handle
        self._keep = keepalive
        self.n_ops = len(ops)

    def run(self, stream, n_iters=1, use_graph=False):
        # use_graph: False / True (a flat graph: lane records in list order on the run stream) / 2 (lanes as parallel branches of the graph)
        check(lib().sbc_plan_run(self._h, C.c_void_p(stream), int(n_iters), 2 if use_graph == 2 else 1 if use_graph else 0))

    def set_persistent_cus(self, n):
        """Grid width (CUs) of this plan's persistent kernels; 0 = the process default (``sbc_plan_set_persistent_cus``)."""
        check(lib().sbc_plan_set_persistent_cus(self._h, int(n)))

    def profile(self, tag):
        check(lib().sbc_plan_profile(self._h, int(tag)))

    def profile_read(self):
        ms, n = C.c_double(), C.c_int64()
        check(lib().sbc_plan_profile_read(self._h, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def close(self):
        if self._h:
            lib().sbc_plan_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
