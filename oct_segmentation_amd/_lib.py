"""ctypes binding of ``liboctseg_hip.so`` (C ABI declared in ``include/octseg.h``).

The product path has no CPU fallback: if the library is missing or a call
fails, a ``RuntimeError`` is raised with ``octseg_last_error()``.
"""
import ctypes as C
import os

# HIP maps streams onto GPU_MAX_HW_QUEUES hardware queues (default 4) in creation order.  The engine overlaps the weight gradients (side
# stream) and a forward lane with the caller's stream; once RCCL has created its own streams (init_process_group("nccl")) the side
# stream can land on the SAME hardware queue as the main one and the overlap is silently gone: measured on one MI355X, U-Net++/resnet101
# B=16, 74.2 -> 80.8 ms per step with a one-rank RCCL group and nothing else changed, 74.6 with 8 (or 2) queues.  The setting is NOT
# free for hipGraph replay, though: with 8 queues the replayed serving ensemble runs at 15.7 ms per frame instead of 7.0 and a replayed
# training step at 98.5 ms instead of 79.2 (profiles/r3_hw_queues_ab.txt), and with 2 graph instantiation fails.  So it is applied to
# the processes that open a process group -- the ranks of a torchrun launch (WORLD_SIZE > 1); single-GPU processes keep HIP's
# default.  Must be in the environment before the HIP runtime initialises (first torch.cuda call); an explicit setting of the user wins.
def set_hw_queues_for_collectives():
    os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')


if int(os.environ.get('WORLD_SIZE', '1') or 1) > 1:
    set_hw_queues_for_collectives()

# torch bundles its own libamdhip64; it must be the HIP runtime this process binds (streams and device
# pointers are torch's), so torch is loaded before liboctseg_hip.so resolves its libamdhip64 dependency.
import torch  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'liboctseg_hip.so')

F32, BF16, F16 = 0, 1, 2
P_CONV, P_CONVT, P_STEM, P_VEC = 0, 1, 2, 3
OPT_KINDS = {'SGD': 0, 'Adam': 1, 'RMSprop': 2, 'RAdam': 3}


class NetDesc(C.Structure):
    _fields_ = [('arch', C.c_char_p), ('encoder', C.c_char_p), ('classes', C.c_int), ('batch', C.c_int),
                ('height', C.c_int), ('width', C.c_int), ('dtype', C.c_int)]


class ParamInfo(C.Structure):
    _fields_ = [('name', C.c_char * 128), ('kind', C.c_int), ('R', C.c_int), ('S', C.c_int), ('O', C.c_int),
                ('I', C.c_int), ('KP', C.c_int), ('offset', C.c_size_t), ('numel', C.c_size_t)]


class BNInfo(C.Structure):
    _fields_ = [('name', C.c_char * 128), ('C', C.c_int), ('mean_offset', C.c_size_t), ('var_offset', C.c_size_t)]


# every exported symbol with (restype, argtypes); tests check the library exports all of them
_P = C.c_void_p
SYMBOLS = {
    'octseg_version': (C.c_int, []),
    'octseg_last_error': (C.c_char_p, []),
    'octseg_plan_create': (C.c_int, [C.POINTER(NetDesc), C.POINTER(_P)]),
    'octseg_plan_destroy': (C.c_int, [_P]),
    'octseg_plan_workspace_bytes': (C.c_size_t, [_P]),
    'octseg_plan_param_numel': (C.c_size_t, [_P]),
    'octseg_plan_buffer_numel': (C.c_size_t, [_P]),
    'octseg_plan_num_params': (C.c_int, [_P]),
    'octseg_plan_param_info': (C.c_int, [_P, C.c_int, C.POINTER(ParamInfo)]),
    'octseg_plan_num_bn': (C.c_int, [_P]),
    'octseg_plan_bn_info': (C.c_int, [_P, C.c_int, C.POINTER(BNInfo)]),
    'octseg_plan_fwd_macs': (C.c_double, [_P]),
    'octseg_plan_exec_macs': (C.c_int, [_P, C.POINTER(C.c_double)]),
    'octseg_plan_find_tensor': (C.c_int, [_P, C.c_char_p, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t), C.POINTER(C.c_int)]),
    'octseg_profile_start': (C.c_int, []),
    'octseg_profile_stop': (C.c_int, [C.POINTER(C.c_double)]),
    'octseg_plan_params_changed': (C.c_int, [_P]),
    'octseg_plan_set_graph': (C.c_int, [_P, C.c_int]),
    'octseg_augment': (C.c_int, [_P, _P, _P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P]),
    'octseg_mask_assemble': (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P, _P]),
    'octseg_net_forward': (C.c_int, [_P, _P, _P, _P, _P, _P, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float),
                                     C.c_int, _P]),
    'octseg_dice_forward': (C.c_int, [_P, _P, _P, _P, _P, _P, _P]),
    'octseg_net_backward': (C.c_int, [_P, _P, _P, _P, _P, _P, C.c_float, _P]),
    'octseg_optim_step': (C.c_int, [C.c_int, _P, _P, _P, _P, C.c_size_t, C.c_float, C.c_float, C.c_int, C.c_float, _P]),
    'octseg_debug_set_stamp': (C.c_int, [_P]),
    'octseg_debug_set_serial': (C.c_int, [C.c_int]),
    'octseg_conv2d_scratch_bytes': (C.c_size_t, [C.c_int] * 8),
    'octseg_conv2d_forward': (C.c_int, [C.c_int, _P, _P, _P, _P] + [C.c_int] * 10 + [_P, _P]),
    'octseg_conv2d_backward_data': (C.c_int, [C.c_int, _P, _P, _P] + [C.c_int] * 10 + [_P, _P]),
    'octseg_conv2d_backward_weight': (C.c_int, [C.c_int, _P, _P, _P] + [C.c_int] * 10 + [_P]),
}

SYMBOLS['octseg_set_deterministic'] = (C.c_int, [C.c_int])
SYMBOLS['octseg_plan_set_dropout'] = (C.c_int, [_P, _P])
SYMBOLS['octseg_net_train_step'] = (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float),
                                              C.c_float, _P])
SYMBOLS['octseg_plan_set_train_graph'] = (C.c_int, [_P, C.c_int])
SYMBOLS['octseg_plan_set_loss'] = (C.c_int, [_P, C.c_int])
SYMBOLS['octseg_plan_set_drop_connect'] = (C.c_int, [_P, _P])
SYMBOLS['octseg_plan_num_drop_connect'] = (C.c_int, [_P])
SYMBOLS['octseg_plan_drop_connect_rate'] = (C.c_float, [_P, C.c_int])
LOSS_KINDS = {'dice': 0, 'bce': 1, 'dice+bce': 2}
SLICE_CB = C.CFUNCTYPE(None, C.c_void_p, C.c_int, C.c_size_t, C.c_size_t)
SYMBOLS['octseg_net_backward_sliced'] = (C.c_int, [_P, _P, _P, _P, _P, _P, C.c_float, _P, C.c_int, _P, SLICE_CB, _P])

_lib = None


def lib():
    """Load the shared library (once).  Raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f'{LIB_PATH} not found: build it with `python -c "import __graft_entry__ as g; g.build()"` '
                f'(or `make -C oct_segmentation_amd/csrc`). There is no CPU fallback.')
        h = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(h, name)  # AttributeError if a declared symbol is not exported
            fn.restype = res
            fn.argtypes = args
        _lib = h
    return _lib


def check(rc):
    if rc != 0:
        msg = lib().octseg_last_error().decode()
        if 'Expected more than 1 value per channel' in msg:   # torch raises ValueError for this one
            raise ValueError(msg)
        raise RuntimeError(f'octseg error {rc}: {msg}')


def ptr(t):
    """Device pointer of a torch tensor (or None)."""
    return None if t is None else C.c_void_p(t.data_ptr())


def stream_ptr():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)
