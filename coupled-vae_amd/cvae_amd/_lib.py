"""ctypes binding of libcvae_hip.so (include/cvae_hip.h).

The product path has no CPU fallback: if the HIP library is missing this module raises at
import, and every call checks the library's return code.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(os.path.dirname(_HERE), 'lib', 'libcvae_hip.so')
if os.environ.get('CVAE_HIP_LIB'):        # diagnostics: an experimental build of the same library (scripts/build_variant.sh)
    LIB_PATH = os.environ['CVAE_HIP_LIB']

if not os.path.exists(LIB_PATH):
    raise ImportError(
        'libcvae_hip.so not found at %s: build it with `make -C coupled-vae_amd/csrc` '
        '(or `python -c "import __graft_entry__ as g; g.build()"`). There is no CPU fallback.' % LIB_PATH)

lib = ctypes.CDLL(LIB_PATH)

P, I, L, F = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_float

# name -> argtypes, exactly as declared in include/cvae_hip.h (restype int unless listed)
SIGNATURES = {
    'cvae_version': [],
    'cvae_release_stream': [P],
    'cvae_persistent_rnn': [I],
    'cvae_rnn_large_wreg': [I],
    'cvae_persist_status_export': [P, P],
    'cvae_persist_status_import': [P, P],
    'cvae_persist_headroom': [I],
    'cvae_shutdown': [],
    'cvae_init': [I],
    'cvae_pack_batch': [P, P, I, I, I, L, L, L, P, P, P, P, P],
    'cvae_pack_batch_indexed': [P, P, P, I, I, I, L, L, L, P, P, P, P, P],
    'cvae_sort_resort': [P, I, P, P, P],
    'cvae_gemm': [I, I, I, I, I, P, L, P, L, P, L, P, I, I, P],
    'cvae_linear_forward': [P, P, P, I, I, I, I, P, P],
    'cvae_linear_backward': [P, P, P, P, I, I, I, I, P, P, P, P, P],
    'cvae_encoder_forward': [I, I, I, I, I, L, P, P, P, P, P, P, P, P, P, P, P, P, P],
    'cvae_encoder_backward_scratch': [I, I, I, I, I],
    'cvae_encoder_backward': [I, I, I, I, I, L, P, P, P, P, P, P, P, P, P, P, P, P, P, P],
    'cvae_encoder_forward_n': [I, I, I, I, I, I, L] + [P] * 13,
    'cvae_encoder_backward_n': [I, I, I, I, I, I, L] + [P] * 14,
    'cvae_decoder_forward': [I, I, I, I, I, I, I, L, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P],
    'cvae_decoder_backward_scratch': [I, I, I, I, I],
    'cvae_decoder_backward': [I, I, I, I, I, L, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P],
    'cvae_decoder_forward_sorted': [I, I, I, I, I, I, L] + [P] * 13,
    'cvae_targets_sentence_major': [P, I, I, I, P, P],
    'cvae_decoder_forward_noprefix': [I, I, I, I, I, L] + [P] * 14,
    'cvae_decoder_backward_noprefix': [I, I, I, I, I, L] + [P] * 16,
    'cvae_decoder_greedy_scratch': [I, I, I, I, I, L],
    'cvae_decoder_greedy': [I, I, I, I, I, L, P, P, P, P, P, P, I, P, P, P],
    'cvae_decoder_beam_scratch': [I, I, I, I, I, I, L],
    'cvae_loss_assemble': [I, I, P, P, P, P],
    'cvae_loss_assemble_backward': [I, I, P, P, P, P],
    'cvae_decoder_beam': [I, I, I, I, I, I, L, P, P, P, P, P, P, I, P, P, P, P],
    'cvae_decoder_forward_n': [I, I, I, I, I, I, L] + [P] * 14,
    'cvae_decoder_backward_n': [I, I, I, I, I, I, L] + [P] * 16,
    'cvae_persist_status': [],
    'cvae_persist_status_peek': [],
    'cvae_persist_status_set': [I],
    'cvae_projce_workspace': [L, L, L],
    'cvae_split_bf16': [I],
    'cvae_stream_rnn_min_bytes': [L],
    'cvae_projce_forward': [P, P, P, P, L, I, I, I, I, L, P, P, P, P, P],
    'cvae_projce_backward_workspace': [L, L, L],
    'cvae_projce_backward': [P, P, P, P, L, P, P, I, I, I, I, L, P, P, P, P, P],
    'cvae_projce_logits_workspace': [L, L, L],
    'cvae_projce_forward_train': [P, P, P, P, L, I, I, I, I, L, P, P, P, P, P, P],
    'cvae_projce_backward_saved': [P, P, P, P, L, P, P, I, I, I, I, L, P, P, P, P, P, P],
    'cvae_projce_forward_train_n': [I, P, P, P, P, L, I, I, I, L, P, P, P, P, P, P],
    'cvae_projce_backward_saved_n_workspace': [L],
    'cvae_projce_backward_saved_n': [I, P, P, P, P, L, P, P, I, I, I, L, P, P, P, P, P, P],
    'cvae_targets_time_major': [P, I, I, I, P, P],
    'cvae_logits': [P, P, P, I, I, I, L, P, P, P],
    'cvae_reparam_forward': [P, P, P, I, I, I, P, P],
    'cvae_reparam_backward': [P, P, I, I, P, P, P],
    'cvae_kl_forward': [P, P, P, P, I, I, P, P],
    'cvae_kl_backward': [P, P, P, P, P, I, I, P, P, P, P, P],
    'cvae_kl_fb_forward': [P, P, I, I, F, P, P],
    'cvae_kl_fb_backward': [P, P, P, I, I, F, P, P, P],
    'cvae_couple_forward': [P, P, I, I, I, I, P, P, P],
    'cvae_couple_backward': [P, P, P, P, I, I, I, I, P, P],
    'cvae_iw_reduce': [P, P, P, P, P, P, I, I, I, P, P, P, P],
    'cvae_mi': [P, P, P, I, I, I, P, P, P],
    'cvae_mi_static': [P, P, P, I, I, I, P, P],
    'cvae_mi_static_max_rows': [],
    'cvae_clip_adam': [P, P, P, P, L, L, L, I, F, F, F, F, F, L, L, P, P, P],
    'cvae_clip_adam_late': [P, P, P, P, L, L, L, I, F, F, F, F, F, L, L, L, L, L, P, P, P],
    'cvae_deterministic': [I],
    'cvae_profile_enable': [I],
    'cvae_profile_read': [I, P, P, P],
    'cvae_sum_time': [P, I, I, P, P],
    'cvae_axpy': [P, F, L, P, P],
}
_RESTYPE_LONG = {'cvae_encoder_backward_scratch', 'cvae_decoder_backward_scratch', 'cvae_projce_workspace',
                 'cvae_decoder_greedy_scratch', 'cvae_projce_logits_workspace', 'cvae_projce_backward_saved_n_workspace', 'cvae_decoder_beam_scratch',
                 'cvae_projce_backward_workspace', 'cvae_mi_static_max_rows', 'cvae_stream_rnn_min_bytes'}

for _name, _args in SIGNATURES.items():
    _fn = getattr(lib, _name)          # AttributeError here = header and library out of sync
    _fn.argtypes = _args
    _fn.restype = L if _name in _RESTYPE_LONG else I
lib.cvae_last_error.argtypes = []
lib.cvae_last_error.restype = ctypes.c_char_p
if lib.cvae_version() < 0 and os.environ.get('CVAE_ALLOW_TIMING_BUILD') != '1':
    raise ImportError('%s is a timing-only diagnostic build (an *_EXP_* switch removed work from a kernel: its '
                      'results are wrong); set CVAE_ALLOW_TIMING_BUILD=1 to load it for timing' % LIB_PATH)


class CvaeHipError(RuntimeError):
    pass


def ptr(t):
    """Device pointer of a contiguous tensor (None -> NULL)."""
    if t is None:
        return None
    assert t.is_contiguous(), 'libcvae_hip needs contiguous tensors'
    return t.data_ptr()


_raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', None)


def stream():
    """The current torch stream's hipStream_t (the raw getter: torch.cuda.current_stream() builds a Stream object
    for ~9 us of host time per call, ~0.25 ms of a 2 ms step)."""
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


def call(name, *args):
    rc = getattr(lib, name)(*args)
    if rc != 0:
        raise CvaeHipError('%s failed (%d): %s' % (name, rc, lib.cvae_last_error().decode()))


def query(name, *args):
    return int(getattr(lib, name)(*args))


def ptr_array(tensors):
    """const float* const* argument: a host array of device pointers (kept alive by the caller)."""
    arr = (ctypes.c_void_p * len(tensors))(*[ptr(t) for t in tensors])
    return arr


_initialised = set()


def init(device_index=None):
    if device_index is None:
        device_index = torch.cuda.current_device()
    if device_index not in _initialised:
        call('cvae_init', device_index)
        _initialised.add(device_index)
