"""ctypes binding of libsc2amd.so (the C-ABI declared in include/sc2_bottleneck.h).

PyTorch is plumbing here: it owns device memory and streams; every compute call below goes through
the C-ABI with raw pointers and the current HIP stream.  There is NO fallback: if the shared library
is missing, or a tensor is not on a HIP device, the call raises.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('SC2_LIB') or os.path.join(_HERE, 'libsc2amd.so')   # SC2_LIB: A/B builds (tools/)
_lib = None

AOP_NONE, AOP_ABS, AOP_SQUARE = 0, 1, 2
EPI_NONE, EPI_GDN, EPI_IGDN, EPI_BIAS, EPI_BIAS_RELU, EPI_BIAS_ADD_RELU, EPI_FUSED_GDN, EPI_FUSED_IGDN, \
    EPI_BIAS_LEAKY_RELU, EPI_GDN2, EPI_IGDN2 = range(11)
FUSABLE_GDN_CHANNELS = (32, 48, 64, 96)   # conv + GDN1 in one launch: one tile must hold every output channel
OUT_BF16_NHWC, OUT_F32_NCHW, OUT_F32_NHWC, OUT_I32_NCHW_SYM = 0, 1, 2, 3
EB_NOISE, EB_DEQUANTIZE = 0, 1
EB_PARAM_STRIDE = 64

# symbols the header declares; tests check each is exported
ABI_SYMBOLS = [
    'sc2_abi_version', 'sc2_last_error', 'sc2_device_count', 'sc2_policy_default', 'sc2_policy_set', 'sc2_policy_get',
    'sc2_nchw_f32_to_nhwc_bf16', 'sc2_nhwc_bf16_to_nchw_f32', 'sc2_avgpool_nhwc', 'sc2_maxpool_nhwc', 'sc2_bn_ws_floats', 'sc2_bn_train_fwd', 'sc2_bn_train_bwd', 'sc2_fc_fwd',
    'sc2_conv_weight_rows', 'sc2_conv_weight_pitch', 'sc2_conv_fused_gdn_supported', 'sc2_conv_patch_supported',
    'sc2_conv2d_fwd', 'sc2_gdn1_bwd_gemm', 'sc2_colsum_bf16', 'sc2_nchw_f32_to_nhwc_f32', 'sc2_conv_f32_chunk_channels', 'sc2_conv2d_f32_fwd',
    'sc2_conv2x2_gdn512_supported', 'sc2_conv2x2_gdn512_fwd', 'sc2_conv1x1_stream_supported', 'sc2_conv1x1_stream_mask_supported', 'sc2_conv1x1_stream_fwd', 'sc2_conv1x1_pair_supported', 'sc2_conv1x1_pair_fwd',
    'sc2_conv0_gdn96_supported', 'sc2_conv0_gdn96_fwd', 'sc2_conv0_gdn96_nchw_fwd', 'sc2_conv2_gdn48_supported', 'sc2_conv2_gdn48_fwd', 'sc2_conv2x2_c48_supported', 'sc2_conv2x2_c48_fwd', 'sc2_conv1x1_kres_supported', 'sc2_conv1x1_kres_fwd', 'sc2_conv1x1_win_supported', 'sc2_conv1x1_win_fwd', 'sc2_conv3x3_win_supported', 'sc2_conv3x3_win_fwd', 'sc2_conv3x3s2_win_supported', 'sc2_conv3x3s2_win_fwd', 'sc2_conv2x2_win_supported', 'sc2_conv2x2_win_fwd', 'sc2_conv2x2_win_tail_supported', 'sc2_conv2x2_win_tail_fwd', 'sc2_conv2d_wgrad', 'sc2_gdn_bwd_pre', 'sc2_gdn_bwd_post', 'sc2_gdn1_rows_supported', 'sc2_gdn1_rows_fwd', 'sc2_gdn1_rows_bwd',
    'sc2_eb_forward', 'sc2_eb_backward', 'sc2_eb_bits_partial_len', 'sc2_eb_symbols', 'sc2_eb_dequantize',
    'sc2_gc_forward', 'sc2_gc_backward', 'sc2_gc_symbols_indexes', 'sc2_gc_dequantize',
    'sc2_pmf_to_quantized_cdf',
    'sc2_rans_max_bytes', 'sc2_rans_workspace_bytes', 'sc2_rans_encode_batch', 'sc2_rans_decode_batch', 'sc2_rans_decode_dequantize_batch', 'sc2_rans_decode_dequantize_batch_ev',
    'sc2_mse_partial_len', 'sc2_mse_sum_bf16', 'sc2_mse_grad_bf16', 'sc2_relu_bwd_bf16', 'sc2_relu_bwd_mse_bf16',
    'sc2_rans_host_tables_create', 'sc2_rans_host_tables_destroy', 'sc2_rans_host_rcp_div', 'sc2_rans_code_host', 'sc2_clock_probe', 'sc2_rans_encode_host', 'sc2_rans_decode_host',
]


class ConvDesc(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in (
        'N', 'H', 'W', 'Cin', 'Cout', 'KH', 'KW', 'stride_h', 'stride_w', 'pad_h', 'pad_w', 'OH', 'OW',
        'a_op', 'epilogue', 'out_format', 'Kpad', 'Cout_pad', 'out_H', 'out_W', 'out_stride_h', 'out_stride_w',
        'out_off_h', 'out_off_w', 'k_order', 'dil_h', 'dil_w')]


class Sc2Error(RuntimeError):
    pass


# --------------------------------------------------------------------------------------------- #
# dispatch policy: no environment variable steers a kernel choice.  `Policy` mirrors `sc2_policy` of the
# C-ABI (the library's own choices), `HostPolicy` holds the choices made on this side of the boundary (which fused /
# persistent kernel a layer is sent to).  Defaults = the measured choices; `configure(name=value, ...)` changes either kind;
# tools/env_policy.py maps the SC2_* variables of the A/B scripts onto it -- the package itself never reads them.
# --------------------------------------------------------------------------------------------- #
POLICY_FIELDS = ('struct_bytes', 'conv_patch3', 'conv_s2', 'conv_persist', 'conv_half', 'conv_big4', 'conv_no_big', 'conv_force_big',
                 'conv_no_epx', 'conv_debug', 'conv_chunk', 'w2_run', 'win_half', 'win_dbg', 'win_stamps', 'p1_half',
                 'p1_nbuf', 'pair_alt', 'f32_persist0', 'dec_stagger', 'wgrad_wgs', 'rans_lds_pad_kb', 'rans_pad_waves', 'rans_ragged2',
                 'rans_ragged2_waves', 'rans_lut8', 'rans_dq_lds', 'wgrad_ct')


class Policy(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in POLICY_FIELDS] + [('reserved', ctypes.c_int32 * 6)]


class HostPolicy(object):
    """Kernel choices made in Python (per layer shape); every attribute is an A/B switch with its measured default."""
    conv_patch = True          # enc.conv2 + GDN48 on the LDS-patch tile kernel where the persistent kernel does not apply
    k_order_tap = False        # True: tap-major K everywhere (default: slab-major where several taps re-read pixels)
    b_tile_major = True        # weights packed [k-slab][row][32]
    conv0_fused = True         # conv0 + GDN96 as one persistent launch
    conv2_fused = True         # conv2 + GDN48 as one persistent launch
    conv_kres = 1              # weights-in-registers 1x1 kernel: 0 off, 1 the K = 1024 layers, 2 also K = 2048
    conv_win = True            # window-plane 3x3 kernel
    conv_win_s2 = True         # ... its stride-2 form
    conv2x2_win = True         # window-plane 2x2 decoder kernel
    w2_tail = True             # decoder's last conv takes layer2.0's conv1 + downsample along
    conv_stream = True         # persistent streaming 1x1 kernel
    conv1x1_pair = True        # conv3 + next block's conv1 in one launch
    conv_c48 = True            # streaming last encoder conv
    conv1x1_win = '1'          # window-plane 1x1 kernel: '0' none, '1' the layers it measured faster on, 'all'
    conv_dilation = True       # dilated layers through the descriptor's dilation (False: phase grids)
    fc_kernel = True           # dedicated classifier kernel
    dense_head = True          # DeepLab / FCN heads and the FPN on the library's kernels in bf16 eval
    rans_fused_dq = True       # decode + dequantise in one coder launch
    gdn_rows = True            # 96- / 256- / 512-channel GDN1 in training: forward and the whole backward on the resident-row kernel (gdn512_rows.hip)
    gdn_bwd_fused = True       # GDN1 backward: element-wise halves in the epilogues of its two GEMMs (sc2_gdn1_bwd_gemm)
    dgrad_win_halves = True    # data gradient of dec.conv2 as two 256-channel launches of the window-plane 2x2 kernel
    train_fused_conv2 = True   # training forward: encoder[2] + GDN1(48) as the fused inference launch that also emits the conv output
    train_fused_dec0 = True    # ... and decoder[0] + IGDN1(512) likewise (conv_gdn512.hip)
    train_fused_conv0 = True   # ... and encoder[0] + GDN1(96) (conv0_gdn96.hip, pixel-pair input)
    bn_train_hip = True        # BatchNorm2d (training mode) + ReLU + residual add of trainable Bottleneck blocks on sc2_bn_train_* (stage 2)
    conv_train_hip = True      # ... and their convolutions on autograd._ConvFn (forward, data and weight gradient on the library's kernels)
    pack_gather = True         # pack_conv_weight as one cast + one gather through a cached index map (False: the chain of layout ops)
    maxpool_hip = True         # nn.MaxPool2d behind a frozen stem (teacher, input-compression classifier) on sc2_maxpool_nhwc (False: torch's)
    relu_mask_fused = True     # the ReLU gradient behind a frozen block's conv2 / conv3 data gradient inside that launch's epilogue (window-plane kernels)
    mse_fused = True           # a feature-matching MSE term on a frozen stack's output: its gradient inside the stack's first ReLU-gradient pass
    teacher_stream = True      # distillation step: the frozen teacher's forward on a stream of its own beside the student's (+ 2 %)
    host_coder_max_streams = 64   # batches of up to this many streams go to the HOST range coder (bs-1 evaluation)
    head_ds_side_stream = False  # the head's downsample layers on a side stream beside conv1 -> conv2 of their block: measured SLOWER (head 2.60 -> 2.67 ms, bench - 1 %: profiles/r06e_ab_ds_side.txt)
    pipeline_host_steps = True   # StagePipeline(host_steps=None): the first batches of a run (up to 4, by the host's core count) are coded by the host thread pool while the device coder's first group is under way, their back stages gated behind the opening burst of front stages: + 1.5 - 2.5 % at K = 20 and K = 100 (profiles/r06o_host_steps_ab.txt); False: device coder only
    eval_graphs = True         # the updated eval forward of SplittableResNet at small batch replays HIP graphs of its device halves (graphs.py)
    eval_graph_max_batch = 1   # ... for batches up to this size (the reference evaluates at batch size 1)


host_policy = HostPolicy()


def get_policy():
    p = Policy()
    lib().sc2_policy_get(ctypes.byref(p))
    return p


def configure(**kw):
    """Sets dispatch-policy fields by name: fields of `sc2_policy` (the library) and attributes of `host_policy` (this side).
    -> the library policy in force.  Unknown names raise."""
    p = get_policy()
    touched = False
    for k, v in kw.items():
        if k in POLICY_FIELDS and k != 'struct_bytes':
            setattr(p, k, int(v))
            touched = True
        elif hasattr(HostPolicy, k):
            setattr(host_policy, k, v)
        else:
            raise Sc2Error('configure: unknown policy field {!r}'.format(k))
    if touched:
        _check(lib().sc2_policy_set(ctypes.byref(p)), 'policy_set')
    return p


def library_fingerprint():
    """{'lib_sha256': hash of the libsc2amd.so this process loads, 'csrc_sha256': hash of the kernel sources beside it (csrc/*.hip,
    *.h, *.cpp and include/sc2_bottleneck.h, names and contents in sorted order)}: what a committed measurement of the kernels
    (profiles/traffic.json) records, so that a reader -- bench.py -- can tell whether it still describes the library that runs.
    The source hash survives a rebuild (hipcc's output is not guaranteed to be byte-reproducible), the library hash a checkout
    without sources."""
    import hashlib
    out = {'lib_sha256': None, 'csrc_sha256': None}
    path = os.environ.get('SC2_LIB') or LIB_PATH
    if os.path.exists(path):
        h = hashlib.sha256()
        with open(path, 'rb') as f:
            for blk in iter(lambda: f.read(1 << 20), b''):
                h.update(blk)
        out['lib_sha256'] = h.hexdigest()
    here = os.path.dirname(os.path.abspath(__file__))
    csrc = os.path.join(here, 'csrc')
    files = sorted(os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith(('.hip', '.h', '.cpp'))) if os.path.isdir(csrc) else []
    header = os.path.join(here, '..', 'include', 'sc2_bottleneck.h')
    if files and os.path.exists(header):
        h = hashlib.sha256()
        for f in files + [header]:
            h.update(os.path.basename(f).encode())
            h.update(open(f, 'rb').read())
        out['csrc_sha256'] = h.hexdigest()
    return out


def lib():
    """Loads libsc2amd.so; raises if it has not been built (python sc2-benchmark_amd/build.py)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise Sc2Error('libsc2amd.so not found at {}: the HIP extension is not built '
                       '(run `python -c "import __graft_entry__ as g; g.build()"`). '
                       'There is no CPU fallback.'.format(LIB_PATH))
    L = ctypes.CDLL(LIB_PATH)
    vp, i32, i64, f32 = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_float
    L.sc2_abi_version.restype = i32
    L.sc2_last_error.restype = ctypes.c_char_p
    L.sc2_device_count.restype = i32
    L.sc2_policy_default.argtypes = [vp]
    L.sc2_policy_default.restype = None
    L.sc2_policy_set.argtypes = [vp]
    L.sc2_policy_get.argtypes = [vp]
    L.sc2_policy_get.restype = None
    L.sc2_nchw_f32_to_nhwc_bf16.argtypes = [vp, vp, i32, i32, i32, i32, i32, vp]
    L.sc2_nhwc_bf16_to_nchw_f32.argtypes = [vp, vp, i32, i32, i32, i32, vp]
    L.sc2_avgpool_nhwc.argtypes = [vp, vp, vp, i32, i32, i32, vp]
    L.sc2_maxpool_nhwc.argtypes = [vp, vp] + [i32] * 10 + [vp]
    L.sc2_bn_ws_floats.argtypes = [ctypes.c_longlong, i32]
    L.sc2_bn_ws_floats.restype = ctypes.c_longlong
    L.sc2_bn_train_fwd.argtypes = [vp, vp, vp, vp, vp, vp, ctypes.c_float, ctypes.c_float, i32, vp, vp, vp, vp, ctypes.c_longlong, i32, vp]
    L.sc2_bn_train_bwd.argtypes = [vp] * 11 + [ctypes.c_longlong, i32, vp]
    L.sc2_fc_fwd.argtypes = [vp, vp, vp, vp, i32, i32, i32, vp]
    L.sc2_conv_weight_rows.argtypes = [i32]
    L.sc2_conv_weight_pitch.argtypes = [i32]
    L.sc2_conv_fused_gdn_supported.argtypes = [ctypes.POINTER(ConvDesc)]
    L.sc2_conv_patch_supported.argtypes = [ctypes.POINTER(ConvDesc)]
    L.sc2_conv2d_fwd.argtypes = [ctypes.POINTER(ConvDesc), vp, vp, vp, vp, vp, vp]
    L.sc2_gdn1_bwd_gemm.argtypes = [ctypes.POINTER(ConvDesc), vp, vp, vp, vp, vp, vp, vp, vp]
    L.sc2_colsum_bf16.argtypes = [vp, ctypes.c_longlong, i32, vp, vp]
    L.sc2_nchw_f32_to_nhwc_f32.argtypes = [vp, vp, i32, i32, i32, i32, i32, vp]
    L.sc2_conv_f32_chunk_channels.argtypes = [i32]
    L.sc2_conv2d_f32_fwd.argtypes = [ctypes.POINTER(ConvDesc), vp, vp, vp, vp, vp, vp]
    L.sc2_conv2x2_gdn512_supported.argtypes = [i32] * 6
    L.sc2_conv2x2_gdn512_fwd.argtypes = [vp, vp, i32, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp]
    L.sc2_conv0_gdn96_supported.argtypes = [i32, i32, i32]
    L.sc2_conv0_gdn96_fwd.argtypes = [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp]
    L.sc2_conv0_gdn96_nchw_fwd.argtypes = [vp, vp, vp, vp, vp, i32, i32, i32, i32, vp]
    L.sc2_conv1x1_kres_supported.argtypes = [i32, i32, i32]
    L.sc2_conv1x1_kres_fwd.argtypes = [vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp]
    L.sc2_conv3x3_win_supported.argtypes = [i32, i32, i32, i32]
    L.sc2_conv3x3_win_fwd.argtypes = [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp]
    L.sc2_conv3x3s2_win_supported.argtypes = [i32, i32, i32, i32]
    L.sc2_conv1x1_win_supported.argtypes = [i32, i32, i32]
    L.sc2_conv2x2_c48_supported.argtypes = [i32, i32, i32, i32]
    L.sc2_conv2x2_c48_fwd.argtypes = [vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp]
    L.sc2_conv1x1_win_fwd.argtypes = [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp]
    L.sc2_conv3x3s2_win_fwd.argtypes = [vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp]
    L.sc2_conv2x2_win_supported.argtypes = [i32, i32, i32, i32, i32]
    L.sc2_conv2x2_win_fwd.argtypes = [vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp]
    L.sc2_conv2x2_win_tail_supported.argtypes = [i32, i32, i32]
    L.sc2_conv2x2_win_tail_fwd.argtypes = [vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp]
    L.sc2_conv2_gdn48_supported.argtypes = [i32, i32, i32]
    L.sc2_conv2_gdn48_fwd.argtypes = [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp]
    L.sc2_conv1x1_pair_supported.argtypes = [i32, i32, i32]
    L.sc2_conv1x1_pair_fwd.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, ctypes.c_longlong, i32, i32, i32, vp]
    L.sc2_conv1x1_stream_supported.argtypes = [i32, i32, i32]
    L.sc2_conv1x1_stream_fwd.argtypes = [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp]
    L.sc2_conv1x1_stream_mask_supported.argtypes = [i32, i32, i32]
    L.sc2_conv2d_wgrad.argtypes = [ctypes.POINTER(ConvDesc), vp, vp, vp, vp]
    L.sc2_gdn1_rows_supported.argtypes = [i32]
    L.sc2_gdn1_rows_fwd.argtypes = [vp, vp, vp, vp, ctypes.c_longlong, i32, i32, vp]
    L.sc2_gdn1_rows_bwd.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, ctypes.c_longlong, i32, i32, vp]
    L.sc2_gdn_bwd_pre.argtypes = [vp, vp, vp, ctypes.c_longlong, i32, i32, vp, vp, vp, vp]
    L.sc2_gdn_bwd_post.argtypes = [vp, vp, vp, ctypes.c_longlong, vp, vp]
    L.sc2_eb_forward.argtypes = [vp, vp, vp, i32, i32, i32, i32, f32, vp, vp, vp, vp, i32, vp]
    L.sc2_eb_backward.argtypes = [vp, vp, vp, i32, i32, i32, i32, f32, vp, vp, vp, vp, i32, vp]
    L.sc2_eb_bits_partial_len.argtypes = [i32, i32, i32]
    L.sc2_gc_forward.argtypes = [vp, vp, i64, vp, i64, vp, i64, i64, i32, f32, f32, vp, vp, vp]
    L.sc2_gc_backward.argtypes = [vp, vp, i64, vp, i64, vp, i64, i64, f32, f32, vp, vp, vp, vp, vp, vp]
    L.sc2_gc_symbols_indexes.argtypes = [vp, vp, i64, vp, i64, i64, i64, vp, i32, f32, vp, vp, vp]
    L.sc2_gc_dequantize.argtypes = [vp, vp, i64, i64, i32, i32, vp, vp, vp]
    L.sc2_eb_symbols.argtypes = [vp, vp, i32, i32, i32, vp, vp]
    L.sc2_eb_dequantize.argtypes = [vp, vp, i32, i32, i32, vp, vp, vp]
    L.sc2_pmf_to_quantized_cdf.argtypes = [ctypes.POINTER(ctypes.c_float), i32, i32,
                                           ctypes.POINTER(ctypes.c_uint32)]
    L.sc2_rans_max_bytes.argtypes = [i64]
    L.sc2_rans_max_bytes.restype = i64
    L.sc2_rans_workspace_bytes.argtypes = [i32, i64, i32, i32]
    L.sc2_rans_workspace_bytes.restype = i64
    L.sc2_rans_encode_batch.argtypes = [vp, vp, i64, i32, i64, vp, i32, i32, vp, vp, vp, i64, vp, vp, vp, vp, i64, vp]
    L.sc2_rans_decode_batch.argtypes = [vp, i64, vp, vp, vp, i64, i32, i64, vp, i32, i32, vp, vp, vp, vp, vp, i64,
                                        vp]
    L.sc2_rans_decode_dequantize_batch.argtypes = [vp, i64, vp, vp, i64, i32, i64, vp, i32, i32, vp, vp, vp, vp, vp, vp, vp, i64, vp]
    L.sc2_rans_decode_dequantize_batch_ev.argtypes = [vp, i64, vp, vp, i64, i32, i64, vp, i32, i32, vp, vp, vp, vp, vp, vp, vp, i64, vp, vp, vp]
    L.sc2_mse_partial_len.argtypes = [ctypes.c_longlong]
    L.sc2_mse_sum_bf16.argtypes = [vp, vp, ctypes.c_longlong, vp, vp]
    L.sc2_mse_grad_bf16.argtypes = [vp, vp, ctypes.c_longlong, vp, vp, vp]
    L.sc2_relu_bwd_bf16.argtypes = [vp, vp, vp, ctypes.c_longlong, vp, vp]
    L.sc2_relu_bwd_mse_bf16.argtypes = [vp, vp, vp, vp, ctypes.c_longlong, i32, vp, vp]
    L.sc2_rans_host_tables_create.argtypes = [vp, i32, i32, vp, vp, ctypes.POINTER(vp)]
    L.sc2_rans_host_tables_destroy.argtypes = [vp]
    L.sc2_rans_host_tables_destroy.restype = None
    L.sc2_rans_host_rcp_div.argtypes = [ctypes.c_uint64, ctypes.c_uint32]
    L.sc2_rans_host_rcp_div.restype = ctypes.c_uint64
    L.sc2_clock_probe.argtypes = [vp, i32, i32, ctypes.c_uint32, vp]
    L.sc2_rans_code_host.argtypes = [vp, vp, vp, i64, i32, i64, vp, i64, vp, vp, vp, vp, i32]
    L.sc2_rans_encode_host.argtypes = [vp, vp, vp, i64, i32, i64, vp, i64, vp, vp, vp, i32]
    L.sc2_rans_decode_host.argtypes = [vp, vp, i64, vp, vp, vp, i64, i32, i64, vp, vp, i32]
    for name in ABI_SYMBOLS:
        getattr(L, name)  # raises AttributeError if the library lacks a declared symbol
    _lib = L
    return L


def last_error():
    return lib().sc2_last_error().decode('utf-8', 'replace')


def _check(rc, what):
    if rc != 0:
        msg = last_error()
        if rc in (-1, -3, -4):
            raise ValueError('{}: {}'.format(what, msg))
        raise Sc2Error('{} failed (code {}): {}'.format(what, rc, msg))


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


class KernelTimer(object):
    """Optional HIP-event timing of individual launches, on the stream they are launched on.

    ``with hip.KernelTimer() as t: ...`` records an event pair around every launch made through this module
    whose tag passes ``select``; ``t.summary()`` (after a synchronize) returns {tag: (count, mean_ms)}.
    Used by bench.py to measure the dominant kernel live inside the timed region.
    """
    active = None

    def __init__(self, select=None):
        self.select = select
        self.records = []

    def __enter__(self):
        KernelTimer.active = self
        return self

    def __exit__(self, *exc):
        KernelTimer.active = None
        return False

    def summary(self):
        out = {}
        for rec in self.records:
            out.setdefault(rec[0], []).append(rec[1].elapsed_time(rec[2]))
        return {k: (len(v), sum(v) / len(v)) for k, v in out.items()}

    def total_ms(self, tag):
        """sum of the durations of every launch tagged `tag` (launches that cover different amounts of work, e.g. the coder's
        dequantise pass over 1, 2, 4 or 8 steps' streams, have no meaningful mean)."""
        return sum(rec[1].elapsed_time(rec[2]) for rec in self.records if rec[0] == tag)


class _timed(object):
    def __init__(self, tag):
        t = KernelTimer.active
        self.t = t if (t is not None and (t.select is None or t.select(tag))) else None
        self.tag = tag

    def __enter__(self):
        if self.t is not None:
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e0.record()
        return self

    def __exit__(self, *exc):
        if self.t is not None:
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record()
            self.t.records.append((self.tag, self.e0, e1))
        return False


def _dev(t, name):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise Sc2Error('{} must be a tensor on a HIP device (got {}); there is no CPU fallback in this package'
                       .format(name, 'device=' + str(t.device) if isinstance(t, torch.Tensor) else type(t)))
    return t


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


# --------------------------------------------------------------------------------------------- #
# layout
# --------------------------------------------------------------------------------------------- #
def nchw_f32_to_nhwc_bf16(x, cpad=None, tag=None):
    """x: f32 [N,C,H,W] contiguous -> bf16 tensor of shape [N,H,W,Cpad] (NHWC memory)."""
    _dev(x, 'x')
    assert x.dtype == torch.float32 and x.dim() == 4
    x = x.contiguous()
    N, C, H, W = x.shape
    cpad = C if cpad is None else cpad
    y = torch.empty((N, H, W, cpad), dtype=torch.bfloat16, device=x.device)
    with _timed(tag or 'layout.nchw_to_nhwc'):
        _check(lib().sc2_nchw_f32_to_nhwc_bf16(_ptr(x), _ptr(y), N, C, H, W, cpad, _stream()), 'nchw_f32_to_nhwc_bf16')
    return y


def nhwc_bf16_to_nchw_f32(x):
    """x: bf16 [N,H,W,C] -> f32 [N,C,H,W]."""
    _dev(x, 'x')
    assert x.dtype == torch.bfloat16 and x.dim() == 4 and x.is_contiguous()
    N, H, W, C = x.shape
    y = torch.empty((N, C, H, W), dtype=torch.float32, device=x.device)
    _check(lib().sc2_nhwc_bf16_to_nchw_f32(_ptr(x), _ptr(y), N, C, H, W, _stream()), 'nhwc_bf16_to_nchw_f32')
    return y


# --------------------------------------------------------------------------------------------- #
# conv
# --------------------------------------------------------------------------------------------- #
def weight_rows(cout):
    return lib().sc2_conv_weight_rows(int(cout))


def weight_pitch(k):
    return lib().sc2_conv_weight_pitch(int(k))


K_TAP_MAJOR, K_SLAB_MAJOR, K_B_TILE_MAJOR, K_B_FRAG_MAJOR = 0, 1, 2, 4   # the last two: flags OR-ed into the k order


def _pack_layout(t, k_order, fill):
    """The packed-weight LAYOUT on any tensor t [Cout, Cin, KH, KW] (any dtype): -> [Cout_pad, Kpad], padding = `fill`.
    K_TAP_MAJOR: k = (kh*KW+kw)*Cin+ci.  K_SLAB_MAJOR (Cin % 32 == 0): k = ((ci//32)*KH*KW + kh*KW+kw)*32 + ci%32."""
    cout, cin, kh, kw = t.shape
    k = cin * kh * kw
    rows, pitch = weight_rows(cout), weight_pitch(k)
    packed = torch.full((rows, pitch), fill, dtype=t.dtype, device=t.device)
    tile_major = bool(k_order & K_B_TILE_MAJOR)
    frag_major = bool(k_order & K_B_FRAG_MAJOR)
    k_order = k_order & 1
    if k_order == K_SLAB_MAJOR:
        assert cin % 32 == 0
        flat = t.reshape(cout, cin // 32, 32, kh * kw).permute(0, 1, 3, 2).reshape(cout, k)
    else:
        flat = t.permute(0, 2, 3, 1).reshape(cout, k)
    packed[:cout, :k] = flat
    if frag_major:   # [k-step][16-row tile][lane = fq*16 + frow][8]: one MFMA operand fragment = 1 KB contiguous
        rows, kpad = packed.shape
        assert rows % 16 == 0 and not tile_major
        packed = packed.reshape(rows // 16, 16, kpad // 32, 4, 8).permute(2, 0, 3, 1, 4).contiguous().reshape(rows, kpad)
    if tile_major:   # [k-slab][row][32]: same bytes, the B tile of a k-slab contiguous; kept 2-D for the shape checks
        rows, kpad = packed.shape
        packed = packed.reshape(rows, kpad // 32, 32).permute(1, 0, 2).contiguous().reshape(rows, kpad)
    return packed


def _pack_conv_weight_reference(w, k_order=K_TAP_MAJOR):
    """pack_conv_weight as the chain of layout ops it was written as (tests compare the gather form with it)."""
    return _pack_layout(w.detach().to(torch.bfloat16), k_order, 0.0)


_PACK_INDEX = {}      # (Cout, Cin, KH, KW, k_order, sub-filter key, device) -> (int32 gather map into the flat weight + one zero, shape)


def _pack_index(shape, k_order, device, sub=None):
    """Gather map of the packed layout: entry -> element of the flat [Cout, Cin, KH, KW] parameter it holds, or n (an appended zero)
    for padding.  `sub` = (rh, sh, rw, sw): the map of the DATA GRADIENT's sub-filter of one stride-parity class -- the weight seen as
    [Cin, Cout, KH, KW], taps kh = rh + sh t, kw = rw + sw u, both flipped (hip.conv2d_dgrad) -- packed straight from the parameter."""
    key = (tuple(shape), k_order, sub, str(device))
    hit = _PACK_INDEX.get(key)
    if hit is None:
        n = 1
        for v in shape:
            n *= int(v)
        idx = torch.arange(n, dtype=torch.int64, device=device).reshape(tuple(shape))
        if sub is not None:
            rh, sh, rw, sw = sub
            idx = idx.permute(1, 0, 2, 3)[:, :, rh::sh, rw::sw].flip(2, 3).contiguous()
        lay = _pack_layout(idx, k_order, n)
        hit = (lay.reshape(-1).to(torch.int32), tuple(lay.shape), bool((lay == n).any().item()))
        if len(_PACK_INDEX) > 512:
            _PACK_INDEX.clear()
        _PACK_INDEX[key] = hit
    return hit


def pack_conv_weight(w, k_order=K_TAP_MAJOR, _sub=None):
    """w: [Cout, Cin, KH, KW] (any float dtype, device) -> bf16 [Cout_pad, Kpad].

    K_TAP_MAJOR: k = (kh*KW+kw)*Cin+ci.  K_SLAB_MAJOR (Cin % 32 == 0): k = ((ci//32)*KH*KW + kh*KW+kw)*32 + ci%32.
    One cast and ONE gather through a cached index map (round 5: the chain of layout ops this replaces was five small launches per
    call, and a training step packs every trainable conv -- and every sub-filter of its data gradient -- anew)."""
    if not host_policy.pack_gather:      # A/B: the chain of layout ops
        if _sub is not None:
            rh, sh, rw, sw = _sub
            w = w.detach().permute(1, 0, 2, 3)[:, :, rh::sh, rw::sw].flip(2, 3).contiguous()
        return _pack_conv_weight_reference(w, k_order)
    idx, shape, padded = _pack_index(w.shape, k_order, w.device, _sub)
    flat = w.detach().reshape(-1).to(torch.bfloat16)
    if padded:
        flat = torch.cat([flat, flat.new_zeros(1)])
    return torch.index_select(flat, 0, idx).reshape(shape)


def conv_patch_supported(x_shape, cout, kh, kw, stride, pad, out_format=OUT_BF16_NHWC, epilogue=EPI_NONE):
    """True if this conv runs on the LDS-resident-patch kernel (weights then packed K_SLAB_MAJOR | K_B_FRAG_MAJOR)."""
    if not host_policy.conv_patch:      # A/B switch (tools/)
        return False
    sh, sw = (stride, stride) if isinstance(stride, int) else stride
    ph, pw = (pad, pad) if isinstance(pad, int) else pad
    N, H, W, Cin = x_shape
    OH = (H + 2 * ph - kh) // sh + 1
    OW = (W + 2 * pw - kw) // sw + 1
    d = ConvDesc(N, H, W, Cin, cout, kh, kw, sh, sw, ph, pw, OH, OW, AOP_NONE, epilogue, out_format,
                 weight_pitch(kh * kw * Cin), weight_rows(cout), 0, 0, 0, 0, 0, 0, K_SLAB_MAJOR | K_B_FRAG_MAJOR)
    return bool(lib().sc2_conv_patch_supported(ctypes.byref(d)))


def preferred_k_order(cin, kh, kw):
    """Slab-major pays when several taps re-read overlapping pixels and the channel count allows it."""
    if host_policy.k_order_tap:      # A/B switch (tools/)
        return K_TAP_MAJOR | (K_B_TILE_MAJOR if host_policy.b_tile_major else 0)
    # measured (same box, tools/layer_times.py): +1.5 % on the 2x2 decoder convs, -4 % on the 25-tap stride-2 conv
    base = K_SLAB_MAJOR if (cin % 32 == 0 and 1 < kh * kw <= 9) else K_TAP_MAJOR
    return base | (K_B_TILE_MAJOR if host_policy.b_tile_major else 0)


def pack_conv0_weight_pairs(w):
    """First encoder conv (Cin=3, k5 s2 p2) on the pixel-pair view of the input.

    The input is stored as bf16 NHWC with channels padded 3->4, viewed as [N, H, W/2, 8]
    (8 = 2 pixels x 4 channels).  A 5-tap stride-2 row filter becomes a 3-tap stride-1 filter over
    pixel pairs: pair tap t covers original taps kw = 2t, 2t+1 (kw = 5 and channel 3 are zero).
    Returns bf16 [Cout_pad, Kpad] with k = (kh*3 + t)*8 + (dw*4 + c).
    """
    cout, cin, kh, kw = w.shape
    assert cin <= 4 and kw == 5, 'pixel-pair packing is for the 3-channel 5x5 stride-2 first conv'
    wp = torch.zeros((cout, kh, 3, 2, 4), dtype=torch.float32, device=w.device)
    wd = w.detach().float()
    for t in range(3):
        for dw in range(2):
            k = 2 * t + dw
            if k < kw:
                wp[:, :, t, dw, :cin] = wd[:, :, :, k].permute(0, 2, 1)
    k_total = kh * 3 * 8
    rows, pitch = weight_rows(cout), weight_pitch(k_total)
    packed = torch.zeros((rows, pitch), dtype=torch.bfloat16, device=w.device)
    packed[:cout, :k_total] = wp.reshape(cout, k_total).to(torch.bfloat16)
    return packed


EPI_GDN1_BWD_PRE, EPI_IGDN1_BWD_PRE, EPI_GDN1_BWD_POST = 11, 12, 13


def gdn1_bwd_gemm(x_nhwc, w_packed, epilogue, ep_x, ep_x2, beta=None, tag=None):
    """One of the two GEMMs of the GDN1 backward with the element-wise half of that backward in its epilogue
    (sc2_gdn1_bwd_gemm): -> (y, y2 or None)."""
    for t, name in ((x_nhwc, 'x'), (w_packed, 'w'), (ep_x, 'ep_x'), (ep_x2, 'ep_x2')):
        _dev(t, name)
        assert t.dtype == torch.bfloat16 and t.is_contiguous(), name
    C = x_nhwc.shape[-1]
    M = x_nhwc.numel() // C
    assert ep_x.numel() == M * C and ep_x2.numel() == M * C
    post = epilogue == EPI_GDN1_BWD_POST
    if not post:
        _dev(beta, 'beta')
        assert beta.dtype == torch.float32 and beta.is_contiguous() and beta.numel() == C
    d = ConvDesc(M, 1, 1, C, C, 1, 1, 1, 1, 0, 0, 1, 1, AOP_NONE if post else AOP_ABS, epilogue, OUT_BF16_NHWC, w_packed.shape[1],
                 w_packed.shape[0], 0, 0, 0, 0, 0, 0, K_TAP_MAJOR, 1, 1)
    y = torch.empty_like(x_nhwc)
    y2 = None if post else torch.empty_like(x_nhwc)
    with _timed(tag or 'gdn.bwd.gemm'):
        _check(lib().sc2_gdn1_bwd_gemm(ctypes.byref(d), _ptr(x_nhwc), _ptr(w_packed), _ptr(y), _ptr(y2), _ptr(ep_x), _ptr(ep_x2),
                                       _ptr(beta), _stream()), 'gdn1_bwd_gemm')
    return y, y2


def gdn1_rows_supported(x_nhwc, C):
    """True if GDN1 over C channels of this bf16 [..., C] tensor runs on the resident-row kernel (C = 96, 256 or 512, < 2 GB)."""
    return bool(host_policy.gdn_rows) and x_nhwc.dtype == torch.bfloat16 and x_nhwc.is_cuda and x_nhwc.is_contiguous() and \
        x_nhwc.shape[-1] == C and x_nhwc.numel() * 2 < 0x7FF00000 and bool(lib().sc2_gdn1_rows_supported(C))


def gdn1_rows_fwd(x_nhwc, gamma_frag, beta, inverse, tag=None):
    """y = GDN1(x) / inverse GDN1(x) for a bf16 [..., 512] tensor; gamma_frag = pack_weight_fragments(effective gamma)."""
    for t, name in ((x_nhwc, 'x'), (gamma_frag, 'gamma_frag'), (beta, 'beta')):
        _dev(t, name)
    C = x_nhwc.shape[-1]
    assert x_nhwc.dtype == torch.bfloat16 and x_nhwc.is_contiguous()
    assert gamma_frag.dtype == torch.bfloat16 and gamma_frag.is_contiguous() and tuple(gamma_frag.shape) == (C // 16, C // 32, 64, 8)
    assert beta.dtype == torch.float32 and beta.is_contiguous() and beta.numel() == C
    y = torch.empty_like(x_nhwc)
    with _timed(tag or 'gdn.rows.fwd'):
        _check(lib().sc2_gdn1_rows_fwd(_ptr(x_nhwc), _ptr(gamma_frag), _ptr(beta), _ptr(y), x_nhwc.numel() // C, C, 1 if inverse else 0,
                                       _stream()), 'gdn1_rows_fwd')
    return y


def gdn1_rows_bwd(x_nhwc, gy_nhwc, gamma_frag, gamma_t_frag, beta, inverse, tag=None, want_d_beta=False):
    """-> (d_norm, dx[, d_beta]), d_norm / dx bf16 like x: the two GEMMs of the GDN1 backward and its element-wise halves in one launch;
    want_d_beta: also the column sums of d_norm (f32 [C]; inside the launch for C = 96 / 256, a column-sum launch behind it for 512)."""
    for t, name in ((x_nhwc, 'x'), (gy_nhwc, 'gy'), (gamma_frag, 'gamma_frag'), (gamma_t_frag, 'gamma_t_frag'), (beta, 'beta')):
        _dev(t, name)
    C = x_nhwc.shape[-1]
    assert x_nhwc.dtype == torch.bfloat16 and x_nhwc.is_contiguous() and gy_nhwc.dtype == torch.bfloat16 and gy_nhwc.is_contiguous()
    assert gy_nhwc.numel() == x_nhwc.numel()
    for g in (gamma_frag, gamma_t_frag):
        assert g.dtype == torch.bfloat16 and g.is_contiguous() and tuple(g.shape) == (C // 16, C // 32, 64, 8)
    assert beta.dtype == torch.float32 and beta.is_contiguous() and beta.numel() == C
    d_norm, dx = torch.empty_like(x_nhwc), torch.empty_like(x_nhwc)
    d_beta = torch.empty((C,), dtype=torch.float32, device=x_nhwc.device) if want_d_beta else None
    with _timed(tag or 'gdn.rows.bwd'):
        _check(lib().sc2_gdn1_rows_bwd(_ptr(x_nhwc), _ptr(gy_nhwc), _ptr(gamma_frag), _ptr(gamma_t_frag), _ptr(beta), _ptr(d_norm),
                                       _ptr(dx), _ptr(d_beta), x_nhwc.numel() // C, C, 1 if inverse else 0, _stream()), 'gdn1_rows_bwd')
    return (d_norm, dx, d_beta) if want_d_beta else (d_norm, dx)


def colsum_bf16(x, C):
    """bf16 [..., C] -> f32 [C] column sums."""
    _dev(x, 'x')
    assert x.dtype == torch.bfloat16 and x.is_contiguous() and x.shape[-1] == C
    out = torch.empty((C,), dtype=torch.float32, device=x.device)
    _check(lib().sc2_colsum_bf16(_ptr(x), x.numel() // C, C, _ptr(out), _stream()), 'colsum_bf16')
    return out


def gdn1_backward(gy_nhwc, x_nhwc, beta, gamma, inverse):
    """Backward of GDN1 / inverse GDN1 on the HIP kernels: returns (dx bf16 NHWC, d_beta f32 [C], d_gamma f32 [C,C]).

    beta [C] / gamma [C,C] are the effective (reparametrised) f32 tensors.  Round 5: for C = 96 and C a multiple of 128 the two
    element-wise halves ride in the epilogues of the two GEMMs (norm GEMM -> d_norm + direct term; gamma^T GEMM -> dx): nine
    passes over the [pixels, C] tensors instead of fifteen (`host_policy.gdn_bwd_fused = False`: the five-launch form, A/B)."""
    C = beta.numel()
    M = x_nhwc.numel() // C
    gy_nhwc = gy_nhwc.contiguous()
    beta = beta.detach().float().contiguous()
    if gdn1_rows_supported(x_nhwc, C):
        # C = 256 / 512: both GEMMs and both element-wise halves in ONE launch with the pixel tile's whole channel row in LDS (four passes
        # over the [pixels, C] tensors: x, g in; d_norm, dx out)
        g = gamma.detach()
        d_norm, dx, d_beta = gdn1_rows_bwd(x_nhwc, gy_nhwc, pack_weight_fragments(g), pack_weight_fragments(g.t()), beta, inverse,
                                           want_d_beta=True)
        d_gamma = conv2d_wgrad(x_nhwc, d_norm, 1, 1, 1, 0, x_abs=True).reshape(C, C)
        return dx, d_beta, d_gamma
    if host_policy.gdn_bwd_fused and (weight_rows(C) % 128 == 0 or weight_rows(C) == 96) and M < (1 << 31):
        g = gamma.detach()
        d_norm, dxd = gdn1_bwd_gemm(x_nhwc, pack_conv_weight(g.reshape(C, C, 1, 1)), EPI_IGDN1_BWD_PRE if inverse else EPI_GDN1_BWD_PRE,
                                    gy_nhwc, x_nhwc, beta, tag='gdn.bwd.pre')
        dx, _ = gdn1_bwd_gemm(d_norm, pack_conv_weight(g.t().reshape(C, C, 1, 1)), EPI_GDN1_BWD_POST, dxd, x_nhwc, tag='gdn.bwd.post')
        d_beta = colsum_bf16(d_norm, C)
        d_gamma = conv2d_wgrad(x_nhwc, d_norm, 1, 1, 1, 0, x_abs=True).reshape(C, C)
        return dx, d_beta, d_gamma
    norm = conv2d_fwd(x_nhwc, pack_conv_weight(gamma.detach().reshape(C, C, 1, 1)), C, 1, 1, 1, 0, a_op=AOP_ABS,
                      epilogue=EPI_BIAS, ep_beta=beta, tag='gdn.bwd.norm')
    d_norm = torch.empty_like(x_nhwc)
    dxd = torch.empty_like(x_nhwc)
    d_beta = torch.empty((C,), dtype=torch.float32, device=x_nhwc.device)
    _check(lib().sc2_gdn_bwd_pre(_ptr(gy_nhwc), _ptr(x_nhwc), _ptr(norm), M, C, 1 if inverse else 0, _ptr(d_norm),
                                 _ptr(dxd), _ptr(d_beta), _stream()), 'gdn_bwd_pre')
    t = conv2d_fwd(d_norm, pack_conv_weight(gamma.detach().t().reshape(C, C, 1, 1)), C, 1, 1, 1, 0, tag='gdn.bwd.gT')
    dx = torch.empty_like(x_nhwc)
    _check(lib().sc2_gdn_bwd_post(_ptr(dxd), _ptr(x_nhwc), _ptr(t), x_nhwc.numel(), _ptr(dx), _stream()),
           'gdn_bwd_post')
    d_gamma = conv2d_wgrad(x_nhwc, d_norm, 1, 1, 1, 0, x_abs=True).reshape(C, C)
    return dx, d_beta, d_gamma


def conv2d_wgrad(x_nhwc, gy_nhwc, kh, kw, stride, pad, x_abs=False):
    """Weight gradient: x bf16 [N,H,W,Cin], gy bf16 [N,OH,OW,Cout] -> f32 [Cout, Cin, KH, KW]."""
    _dev(x_nhwc, 'x')
    _dev(gy_nhwc, 'gy')
    assert x_nhwc.dtype == torch.bfloat16 and gy_nhwc.dtype == torch.bfloat16
    assert x_nhwc.is_contiguous() and gy_nhwc.is_contiguous()
    sh, sw = (stride, stride) if isinstance(stride, int) else stride
    ph, pw = (pad, pad) if isinstance(pad, int) else pad
    N, H, W, Cin = x_nhwc.shape
    OH = (H + 2 * ph - kh) // sh + 1
    OW = (W + 2 * pw - kw) // sw + 1
    cout = gy_nhwc.shape[3]
    assert tuple(gy_nhwc.shape) == (N, OH, OW, cout)
    d = ConvDesc(N, H, W, Cin, cout, kh, kw, sh, sw, ph, pw, OH, OW, AOP_ABS if x_abs else AOP_NONE, 0, 0, 0, 0, 0, 0,
                 0, 0, 0, 0, 0)
    dw = torch.empty((cout, kh * kw * Cin), dtype=torch.float32, device=x_nhwc.device)
    with _timed('wgrad'):
        _check(lib().sc2_conv2d_wgrad(ctypes.byref(d), _ptr(x_nhwc), _ptr(gy_nhwc), _ptr(dw), _stream()),
               'conv2d_wgrad')
    return dw.view(cout, kh, kw, Cin).permute(0, 3, 1, 2)


def conv2d_dgrad(gy_nhwc, weight, stride, pad, in_hw, out_dtype=torch.bfloat16, cache=None):
    """Data gradient of y = conv2d(x, weight, stride, pad) on the forward implicit-GEMM kernel.

    gy_nhwc: bf16 [N,OH,OW,Cout]; weight: [Cout,Cin,KH,KW]; returns NHWC [N,H,W,Cin] (H, W = in_hw).
    A transposed convolution is, per stride-parity class (ih % s, iw % s), a stride-1 correlation of gy with the
    sub-filter of taps kh = r + s*t (r = (ih + p) % s), flipped; each class is one launch that scatters its rows
    to every s-th pixel of the gradient.  Stride 1 is the single-class case (full flip, pad k-1-p).
    `cache`: a dict owned by the caller of a FROZEN weight -- the packed sub-filters are then built once, not per step.
    """
    cout, cin, KH, KW = weight.shape
    sh, sw = (stride, stride) if isinstance(stride, int) else stride
    ph, pw = (pad, pad) if isinstance(pad, int) else pad
    H, W = in_hw
    N = gy_nhwc.shape[0]
    fmt = OUT_BF16_NHWC if out_dtype == torch.bfloat16 else OUT_F32_NHWC
    gx = torch.empty((N, H, W, cin), dtype=out_dtype, device=gy_nhwc.device)
    covered_h = covered_w = 0
    wt = weight.detach().permute(1, 0, 2, 3)   # [Cin, Cout, KH, KW]
    for ch in range(sh):
        rh = (ch + ph) % sh
        khs = list(range(rh, KH, sh))
        rows = (H - ch + sh - 1) // sh if H > ch else 0
        for cw in range(sw):
            rw = (cw + pw) % sw
            kws = list(range(rw, KW, sw))
            cols = (W - cw + sw - 1) // sw if W > cw else 0
            if rows == 0 or cols == 0:
                continue
            if not khs or not kws:      # no tap reaches this parity class: its gradient is zero
                gx[:, ch::sh, cw::sw] = 0
                continue
            qh, qw = (ch + ph - rh) // sh, (cw + pw - rw) // sw
            pad_h, pad_w = len(khs) - 1 - qh, len(kws) - 1 - qw
            if pad_h < 0 or pad_w < 0:
                raise Sc2Error('conv2d_dgrad: unsupported geometry k={} s={} p={}'.format((KH, KW), (sh, sw), (ph, pw)))
            win = (sh == 1 and sw == 1 and out_dtype == torch.bfloat16 and pad_h == pad_w and gy_nhwc.is_contiguous() and
                   conv2x2_win_supported(tuple(gy_nhwc.shape), cin, len(khs), len(kws), 1, pad_h))
            # ... and a 512-channel gradient (dec.conv2: 512 -> 256, k2, p0) as two 256-channel halves of that kernel (pad 1)
            win2 = (not win and cin == 512 and pad_h == 1 and sh == 1 and sw == 1 and out_dtype == torch.bfloat16 and pad_w == 1 and
                    gy_nhwc.is_contiguous() and host_policy.dgrad_win_halves and gx.numel() * 2 < 0x7FF00000 and
                    conv2x2_win_supported(tuple(gy_nhwc.shape), 256, len(khs), len(kws), 1, 1))
            # (the packed sub-filters belong to THIS weight tensor at this version, stride and padding: a cache dict shared
            #  between layers, or kept across an optimizer step, must miss -- ADVICE r4)
            key = (ch, cw, win, win2, weight.data_ptr(), weight._version, tuple(weight.shape), (sh, sw), (ph, pw))
            packed = cache.get(key) if cache is not None else None
            if packed is None:
                # (strided SLICES, not index lists: an index list is a host tensor copied to the device per call)
                if win or win2:
                    sub = wt[:, :, rh::sh, rw::sw].flip(2, 3).contiguous()      # taps in correlation order
                    packed = (pack_conv2x2_win(sub[:256]), pack_conv2x2_win(sub[256:])) if win2 else pack_conv2x2_win(sub)
                else:   # (one gather from the parameter itself: sub-filter, flip and packed layout folded into the index map)
                    packed = pack_conv_weight(weight, K_TAP_MAJOR, _sub=(rh, sh, rw, sw))
                if cache is not None:
                    cache[key] = packed
            if sh == 1 and sw == 1:
                if win2:     # (1.07 ms on the 256-row tile kernel at bs 256)
                    conv2x2_win_fwd(gy_nhwc, packed[0], 1, tag='dgrad', out=gx, channel0=0)
                    conv2x2_win_fwd(gy_nhwc, packed[1], 1, tag='dgrad', out=gx, channel0=256)
                    continue
                if win:
                    # the decoder's last conv (256 -> 256, k2, p1): its data gradient is the k2 p0 conv of the first window-plane
                    # geometry (1.16 ms on the tile kernel at bs 256, 0.37 here)
                    gx = conv2x2_win_fwd(gy_nhwc, packed, pad_h, tag='dgrad')
                    continue
                conv2d_fwd(gy_nhwc, packed, cin, len(khs), len(kws), 1, (pad_h, pad_w), out_format=fmt, out=gx,
                           tag='dgrad')
            else:
                conv2d_fwd(gy_nhwc, packed, cin, len(khs), len(kws), 1, (pad_h, pad_w), out_format=fmt, tag='dgrad',
                           scatter=(rows, cols, gx, sh, sw, ch, cw))
    return gx


def conv_fused_gdn_supported(x_shape, cout, kh, kw, stride, pad, out_format=OUT_BF16_NHWC):
    """Non-zero if conv + GDN1 runs as ONE launch (EPI_FUSED_GDN / EPI_FUSED_IGDN) for this geometry (x_shape =
    [N,H,W,Cin]): 1 = ep_x takes GDN1.effective()'s packed-row gamma, 2 = GDN1.effective_fragments()'s."""
    sh, sw = (stride, stride) if isinstance(stride, int) else stride
    ph, pw = (pad, pad) if isinstance(pad, int) else pad
    N, H, W, Cin = x_shape
    OH = (H + 2 * ph - kh) // sh + 1
    OW = (W + 2 * pw - kw) // sw + 1
    d = ConvDesc(N, H, W, Cin, cout, kh, kw, sh, sw, ph, pw, OH, OW, AOP_NONE, EPI_FUSED_GDN, out_format,
                 weight_pitch(kh * kw * Cin), weight_rows(cout), 0, 0, 0, 0, 0, 0, K_TAP_MAJOR)
    return int(lib().sc2_conv_fused_gdn_supported(ctypes.byref(d)))   # 0 no, 1 packed-row gamma, 2 fragment-major gamma


def conv2d_fwd(x_nhwc, w_packed, cout, kh, kw, stride, pad, a_op=AOP_NONE, epilogue=EPI_NONE,
               out_format=OUT_BF16_NHWC, ep_x=None, ep_beta=None, out=None, tag=None, scatter=None,
               k_order=K_TAP_MAJOR, dilation=1):
    """x_nhwc: bf16 [N,H,W,Cin]; returns the output tensor.

    out_format OUT_BF16_NHWC -> bf16 [N,OH,OW,Cout]; OUT_F32_NCHW -> f32 [N,Cout,OH,OW];
    OUT_F32_NHWC -> f32 [N,OH,OW,Cout].  stride / pad / dilation: int or (h, w) (dilation > 1: Cout > 96, plain epilogues).
    """
    _dev(x_nhwc, 'x')
    _dev(w_packed, 'w_packed')
    assert x_nhwc.dtype == torch.bfloat16 and x_nhwc.dim() == 4 and x_nhwc.is_contiguous()
    assert w_packed.dtype == torch.bfloat16 and w_packed.is_contiguous()
    sh, sw = (stride, stride) if isinstance(stride, int) else stride
    ph, pw = (pad, pad) if isinstance(pad, int) else pad
    dh, dw = (dilation, dilation) if isinstance(dilation, int) else dilation
    N, H, W, Cin = x_nhwc.shape
    OH = (H + 2 * ph - dh * (kh - 1) - 1) // sh + 1
    OW = (W + 2 * pw - dw * (kw - 1) - 1) // sw + 1
    sc = (0, 0, 0, 0, 0, 0)
    if scatter is not None:   # (OH, OW, out tensor [N,out_H,out_W,cout], stride_h, stride_w, off_h, off_w)
        OH, OW, out, s_h, s_w, o_h, o_w = scatter
        assert out.dtype == (torch.bfloat16 if out_format == OUT_BF16_NHWC else torch.float32) and out.is_contiguous()
        assert out.shape[0] == N and out.shape[3] == cout
        sc = (out.shape[1], out.shape[2], s_h, s_w, o_h, o_w)
    d = ConvDesc(N, H, W, Cin, cout, kh, kw, sh, sw, ph, pw, OH, OW, a_op, epilogue, out_format,
                 w_packed.shape[1], w_packed.shape[0], *(sc + (k_order, dh, dw)))
    if out is None:
        if out_format == OUT_BF16_NHWC:
            out = torch.empty((N, OH, OW, cout), dtype=torch.bfloat16, device=x_nhwc.device)
        elif out_format == OUT_F32_NCHW:
            out = torch.empty((N, cout, OH, OW), dtype=torch.float32, device=x_nhwc.device)
        elif out_format == OUT_I32_NCHW_SYM:
            out = torch.empty((N, cout, OH, OW), dtype=torch.int32, device=x_nhwc.device)
        else:
            out = torch.empty((N, OH, OW, cout), dtype=torch.float32, device=x_nhwc.device)
    if ep_x is not None:
        _dev(ep_x, 'ep_x')
        assert ep_x.dtype == torch.bfloat16 and ep_x.is_contiguous()
        if epilogue in (EPI_FUSED_GDN, EPI_FUSED_IGDN):   # ep_x carries the gamma matrix (packed rows or fragments)
            assert tuple(ep_x.shape) in ((weight_rows(cout), weight_pitch(cout)), (cout // 16, cout // 32, 64, 8))
        else:
            assert ep_x.numel() == N * OH * OW * cout
    if ep_beta is not None:
        _dev(ep_beta, 'ep_beta')
        assert ep_beta.dtype == torch.float32 and ep_beta.is_contiguous() and ep_beta.numel() == cout
    with _timed(tag or 'conv{}x{}_{}to{}'.format(kh, kw, Cin, cout)):
        _check(lib().sc2_conv2d_fwd(ctypes.byref(d), _ptr(x_nhwc), _ptr(w_packed), _ptr(out), _ptr(ep_x),
                                    _ptr(ep_beta), _stream()), 'conv2d_fwd')
    return out


def conv2x2_gdn512_supported(cin, cout, kh, kw, stride, pad):
    """True if Conv2d(cin -> 512, k2, s1, p1) followed by a 512-channel GDN1 runs as the single fused launch."""
    sh, sw = (stride, stride) if isinstance(stride, int) else stride
    ph, pw = (pad, pad) if isinstance(pad, int) else pad
    return sh == sw and ph == pw and bool(lib().sc2_conv2x2_gdn512_supported(cin, cout, kh, kw, sh, ph))


def conv0_gdn96_supported(x_pairs_shape, cout):
    """True if the pixel-pair first conv + GDN1(96) runs as the single persistent launch (any width: 112-pixel output
    segments)."""
    if not host_policy.conv0_fused:      # A/B switch (tools/)
        return False
    return len(x_pairs_shape) == 4 and bool(lib().sc2_conv0_gdn96_supported(x_pairs_shape[3], cout, x_pairs_shape[2]))


def conv0_gdn96_fwd(x_pairs, w_frag, gamma_frag, beta, inverse=False, tag=None, want_t=False):
    """x_pairs bf16 [N,H,W/2,8] -> bf16 NHWC [N,(H-1)//2+1,W/2,96]; w_frag / gamma_frag: pack_weight_fragments of the
    pair-packed weights [96,128] and of the effective gamma [96,96]."""
    for t, name in ((x_pairs, 'x_pairs'), (w_frag, 'w_frag'), (gamma_frag, 'gamma_frag'), (beta, 'beta')):
        _dev(t, name)
    assert x_pairs.dtype == torch.bfloat16 and x_pairs.dim() == 4 and x_pairs.is_contiguous() and x_pairs.shape[3] == 8
    N, H, WP, _ = x_pairs.shape
    assert w_frag.dtype == torch.bfloat16 and w_frag.is_contiguous() and tuple(w_frag.shape) == (6, 4, 64, 8)
    assert gamma_frag.dtype == torch.bfloat16 and gamma_frag.is_contiguous() and tuple(gamma_frag.shape) == (6, 3, 64, 8)
    assert beta.dtype == torch.float32 and beta.is_contiguous() and beta.numel() == 96
    out = torch.empty((N, (H - 1) // 2 + 1, WP, 96), dtype=torch.bfloat16, device=x_pairs.device)
    t = torch.empty_like(out) if want_t else None        # (training: the conv output in front of the GDN)
    with _timed(tag or 'conv0_gdn96'):
        _check(lib().sc2_conv0_gdn96_fwd(_ptr(x_pairs), _ptr(w_frag), _ptr(gamma_frag), _ptr(beta), _ptr(out), _ptr(t), N, H, WP,
                                         1 if inverse else 0, _stream()), 'conv0_gdn96_fwd')
    return (out, t) if want_t else out


def conv0_gdn96_nchw_fwd(x_nchw, w_frag, gamma_frag, beta, inverse=False, tag=None):
    """The same launch on the f32 NCHW image batch [N,3,H,W] itself (W even): the colour planes are read in place and rounded
    to bf16 as they are staged -- no layout pass, bit-identical to nchw_f32_to_nhwc_bf16(x, 4) + conv0_gdn96_fwd."""
    for t, name in ((x_nchw, 'x_nchw'), (w_frag, 'w_frag'), (gamma_frag, 'gamma_frag'), (beta, 'beta')):
        _dev(t, name)
    assert x_nchw.dtype == torch.float32 and x_nchw.dim() == 4 and x_nchw.is_contiguous() and x_nchw.shape[1] == 3
    N, _, H, W = x_nchw.shape
    assert W % 2 == 0
    assert w_frag.dtype == torch.bfloat16 and w_frag.is_contiguous() and tuple(w_frag.shape) == (6, 4, 64, 8)
    assert gamma_frag.dtype == torch.bfloat16 and gamma_frag.is_contiguous() and tuple(gamma_frag.shape) == (6, 3, 64, 8)
    assert beta.dtype == torch.float32 and beta.is_contiguous() and beta.numel() == 96
    out = torch.empty((N, (H - 1) // 2 + 1, W // 2, 96), dtype=torch.bfloat16, device=x_nchw.device)
    with _timed(tag or 'conv0_gdn96'):
        _check(lib().sc2_conv0_gdn96_nchw_fwd(_ptr(x_nchw), _ptr(w_frag), _ptr(gamma_frag), _ptr(beta), _ptr(out), N, H, W,
                                              1 if inverse else 0, _stream()), 'conv0_gdn96_nchw_fwd')
    return out


def conv2_gdn48_supported(x_shape, cout, kh, kw, stride, pad):
    """True if this conv + GDN1(48) runs as the persistent weights-in-registers launch (96 -> 48, k5 s2 p2, W = 112)."""
    if not host_policy.conv2_fused:      # A/B switch (tools/)
        return False
    sh, sw = (stride, stride) if isinstance(stride, int) else stride
    ph, pw = (pad, pad) if isinstance(pad, int) else pad
    return (len(x_shape) == 4 and (kh, kw, sh, sw, ph, pw) == (5, 5, 2, 2, 2, 2) and
            bool(lib().sc2_conv2_gdn48_supported(x_shape[3], cout, x_shape[2])))


def conv2_gdn48_fwd(x_nhwc, w_frag, gamma_frag, beta, inverse=False, tag=None, want_t=False):
    """x bf16 [N,H,W,96] -> bf16 NHWC [N,(H-1)//2+1,(W-1)//2+1,48] (W = 112: the static geometry of the 224 x 224 operating point;
    any other width: 56-column output segments); w_frag: pack_conv_weight(w, K_SLAB_MAJOR | K_B_FRAG_MAJOR);
    gamma_frag: pack_weight_fragments of the effective gamma zero-padded to [48, 64]."""
    for t, name in ((x_nhwc, 'x'), (w_frag, 'w_frag'), (gamma_frag, 'gamma_frag'), (beta, 'beta')):
        _dev(t, name)
    assert x_nhwc.dtype == torch.bfloat16 and x_nhwc.dim() == 4 and x_nhwc.is_contiguous() and x_nhwc.shape[3] == 96
    N, H, W, _ = x_nhwc.shape
    assert w_frag.dtype == torch.bfloat16 and w_frag.is_contiguous() and tuple(w_frag.shape) == (48, weight_pitch(2400))
    assert gamma_frag.dtype == torch.bfloat16 and gamma_frag.is_contiguous() and tuple(gamma_frag.shape) == (3, 2, 64, 8)
    assert beta.dtype == torch.float32 and beta.is_contiguous() and beta.numel() == 48
    out = torch.empty((N, (H - 1) // 2 + 1, (W - 1) // 2 + 1, 48), dtype=torch.bfloat16, device=x_nhwc.device)
    t = torch.empty_like(out) if want_t else None        # (training: the conv output in front of the GDN)
    with _timed(tag or 'conv2_gdn48'):
        _check(lib().sc2_conv2_gdn48_fwd(_ptr(x_nhwc), _ptr(w_frag), _ptr(gamma_frag), _ptr(beta), _ptr(out), _ptr(t), N, H, W,
                                         1 if inverse else 0, _stream()), 'conv2_gdn48_fwd')
    return (out, t) if want_t else out


def avgpool_nhwc(x_nhwc, want_f32=True, want_bf16=False):
    """bf16 NHWC [N,H,W,C] -> (mean over H,W as f32 [N,C] or None, the same rounded to bf16 [N,C] or None)."""
    _dev(x_nhwc, 'x')
    assert x_nhwc.dtype == torch.bfloat16 and x_nhwc.dim() == 4 and x_nhwc.is_contiguous()
    N, H, W, C = x_nhwc.shape
    f32 = torch.empty((N, C), dtype=torch.float32, device=x_nhwc.device) if want_f32 else None
    b16 = torch.empty((N, C), dtype=torch.bfloat16, device=x_nhwc.device) if want_bf16 else None
    with _timed('avgpool'):
        _check(lib().sc2_avgpool_nhwc(_ptr(x_nhwc), _ptr(f32), _ptr(b16), N, H * W, C, _stream()), 'avgpool_nhwc')
    return f32, b16


def maxpool_nhwc(x_nhwc, kernel, stride, pad, tag=None):
    """nn.MaxPool2d (floor mode) on a bf16 NHWC map [N,H,W,C], C % 8 == 0 -> [N,OH,OW,C]; bit-identical to torch's."""
    _dev(x_nhwc, 'x')
    assert x_nhwc.dtype == torch.bfloat16 and x_nhwc.dim() == 4 and x_nhwc.is_contiguous()
    N, H, W, C = x_nhwc.shape
    (kh, kw), (sh, sw), (ph, pw) = [(v, v) if isinstance(v, int) else tuple(v) for v in (kernel, stride, pad)]
    out = torch.empty((N, (H + 2 * ph - kh) // sh + 1, (W + 2 * pw - kw) // sw + 1, C), dtype=torch.bfloat16, device=x_nhwc.device)
    with _timed(tag or 'maxpool'):
        _check(lib().sc2_maxpool_nhwc(_ptr(x_nhwc), _ptr(out), N, H, W, C, kh, kw, sh, sw, ph, pw, _stream()), 'maxpool_nhwc')
    return out


def bn_train_fwd(x_nhwc, gamma, beta, running_mean, running_var, momentum, eps, relu, residual=None, tag=None):
    """BatchNorm2d with batch statistics on a bf16 NHWC map (+ residual, ReLU): -> (y, save_mean, save_rstd).  gamma / beta /
    running_* : f32 [C] (running_* may be None); running statistics are updated in place as nn.BatchNorm2d does."""
    _dev(x_nhwc, 'x')
    assert x_nhwc.dtype == torch.bfloat16 and x_nhwc.dim() == 4 and x_nhwc.is_contiguous()
    C = x_nhwc.shape[3]
    M = x_nhwc.numel() // C
    if M <= 1:
        raise ValueError('Expected more than 1 value per channel when training, got input size {}'.format(tuple(x_nhwc.shape)))
    for t in (gamma, beta, running_mean, running_var):
        assert t is None or (t.dtype == torch.float32 and t.is_contiguous() and t.numel() == C and t.device == x_nhwc.device)
    assert residual is None or (residual.dtype == torch.bfloat16 and residual.shape == x_nhwc.shape and residual.is_contiguous())
    y = torch.empty_like(x_nhwc)
    stats = torch.empty((2, C), dtype=torch.float32, device=x_nhwc.device)      # save_mean, save_rstd
    ws = torch.empty((int(lib().sc2_bn_ws_floats(M, C)),), dtype=torch.float32, device=x_nhwc.device)
    with _timed(tag or 'bn.fwd'):
        _check(lib().sc2_bn_train_fwd(_ptr(x_nhwc), _ptr(residual), _ptr(gamma), _ptr(beta), _ptr(running_mean), _ptr(running_var),
                                      float(momentum), float(eps), 1 if relu else 0, _ptr(y), stats[0].data_ptr(), stats[1].data_ptr(),
                                      _ptr(ws), M, C, _stream()), 'bn_train_fwd')
    return y, stats[0], stats[1]


def bn_train_bwd(dy, x_nhwc, y_or_none, gamma, save_mean, save_rstd, want_dz=False, tag=None):
    """-> (dx bf16, dz bf16 or None, dgamma f32 [C], dbeta f32 [C]); y_or_none: the forward's output if it applied the ReLU."""
    _dev(dy, 'dy')
    assert dy.dtype == torch.bfloat16 and dy.is_contiguous() and dy.shape == x_nhwc.shape and x_nhwc.is_contiguous()
    assert y_or_none is None or (y_or_none.is_contiguous() and y_or_none.shape == x_nhwc.shape)
    C = x_nhwc.shape[3]
    M = x_nhwc.numel() // C
    dx = torch.empty_like(x_nhwc)
    dz = torch.empty_like(x_nhwc) if want_dz else None
    out = torch.empty((2, C), dtype=torch.float32, device=dy.device)           # dgamma, dbeta
    ws = torch.empty((int(lib().sc2_bn_ws_floats(M, C)),), dtype=torch.float32, device=dy.device)
    with _timed(tag or 'bn.bwd'):
        _check(lib().sc2_bn_train_bwd(_ptr(dy), _ptr(x_nhwc), _ptr(y_or_none), _ptr(gamma), _ptr(save_mean), _ptr(save_rstd), _ptr(dx),
                                      _ptr(dz), out[0].data_ptr(), out[1].data_ptr(), _ptr(ws), M, C, _stream()), 'bn_train_bwd')
    return dx, dz, out[0], out[1]


def fc_fwd(a, w_frag, bias, tag=None):
    """out f32 [M, Npad] = a [M, K] (bf16) @ W^T + bias; w_frag = pack_weight_fragments(W [Npad, K]), Npad % 16 == 0."""
    for t, name in ((a, 'a'), (w_frag, 'w_frag'), (bias, 'bias')):
        _dev(t, name)
    assert a.dtype == torch.bfloat16 and a.dim() == 2 and a.is_contiguous()
    M, K = a.shape
    npad = w_frag.shape[0] * 16
    assert w_frag.dtype == torch.bfloat16 and w_frag.is_contiguous() and tuple(w_frag.shape) == (npad // 16, K // 32, 64, 8)
    assert bias.dtype == torch.float32 and bias.is_contiguous() and bias.numel() == npad
    out = torch.empty((M, npad), dtype=torch.float32, device=a.device)
    with _timed(tag or 'fc'):
        _check(lib().sc2_fc_fwd(_ptr(a), _ptr(w_frag), _ptr(bias), _ptr(out), M, K, npad, _stream()), 'fc_fwd')
    return out


def pack_weight_fragments(w2d):
    """[N, K] matrix (row = output channel) -> bf16 MFMA-fragment blocks [N/16][K/32][64][8]: entry (jt, ks, lane =
    fq*16 + frow, e) = w2d[jt*16 + frow, ks*32 + fq*8 + e], one operand fragment = 1 KB contiguous."""
    _dev(w2d, 'w2d')
    n, k = w2d.shape
    assert n % 16 == 0 and k % 32 == 0
    g = w2d.detach().to(torch.bfloat16).reshape(n // 16, 16, k // 32, 4, 8)      # jt, frow, ks, fq, e
    return g.permute(0, 2, 3, 1, 4).contiguous().reshape(n // 16, k // 32, 64, 8)


def conv1x1_kres_supported(cin, cout, kh, kw, stride, pad):
    """True if this 1x1 conv runs on the weights-in-registers kernel (K = 1024 or 2048, stride 1 or 2)."""
    sh, sw = (stride, stride) if isinstance(stride, int) else stride
    ph, pw = (pad, pad) if isinstance(pad, int) else pad
    mode = str(host_policy.conv_kres)          # A/B switch (tools/): 0 off, 2 = also the K = 2048 layers
    if mode == '0':
        return False
    if cin == 2048 and mode != '2':   # measured: layer4 conv1 (2048 -> 512, 12 544 pixels) 0.065 ms vs 0.063 ms on the tile kernel
        return False
    return (kh, kw, ph, pw) == (1, 1, 0, 0) and sh == sw and bool(lib().sc2_conv1x1_kres_supported(cin, cout, sh))


def conv1x1_kres_fwd(x_nhwc, w_frag, bias, stride=1, relu=False, tag=None):
    """x bf16 [N,H,W,Cin] -> bf16 [N,OH,OW,Cout]; w_frag = pack_weight_fragments(w[Cout, Cin]); Cin 1024 or 2048."""
    for t, name in ((x_nhwc, 'x'), (w_frag, 'w_frag'), (bias, 'bias')):
        _dev(t, name)
    assert x_nhwc.dtype == torch.bfloat16 and x_nhwc.dim() == 4 and x_nhwc.is_contiguous()
    N, H, W, Cin = x_nhwc.shape
    cout = w_frag.shape[0] * 16
    assert w_frag.dtype == torch.bfloat16 and w_frag.is_contiguous() and tuple(w_frag.shape) == (cout // 16, Cin // 32, 64, 8)
    assert bias.dtype == torch.float32 and bias.is_contiguous() and bias.numel() == cout
    stride = int(stride)
    out = torch.empty((N, (H - 1) // stride + 1, (W - 1) // stride + 1, cout), dtype=torch.bfloat16, device=x_nhwc.device)
    with _timed(tag or 'conv1x1_kres'):
        _check(lib().sc2_conv1x1_kres_fwd(_ptr(x_nhwc), _ptr(w_frag), _ptr(bias), _ptr(out), N, H, W, Cin, cout, stride,
                                          1 if relu else 0, _stream()), 'conv1x1_kres_fwd')
    return out


def conv3x3_win_supported(h, w, cin, cout, kh, kw, stride, pad, dilation=(1, 1)):
    """True if this conv runs on the window-plane 3x3 kernel: stride 1, pad 1 on 28 / 14 / 7 pixel maps, or stride 2, pad 1 on
    56 / 28 / 14 pixel maps (h, w = the INPUT map)."""
    sh, sw = (stride, stride) if isinstance(stride, int) else stride
    ph, pw = (pad, pad) if isinstance(pad, int) else pad
    if not host_policy.conv_win:      # A/B switch (tools/)
        return False
    if (kh, kw, ph, pw) != (3, 3, 1, 1) or tuple(dilation) != (1, 1):
        return False
    if (sh, sw) == (2, 2):
        return host_policy.conv_win_s2 and bool(lib().sc2_conv3x3s2_win_supported(h, w, cin, cout))
    return (sh, sw) == (1, 1) and \
        bool(lib().sc2_conv3x3_win_supported(h, w, cin, cout))   # (the caller checks the 2 GB operand bound: head._Conv)


def pack_conv_win(w):
    """[Cout, Cin, KH, KW] -> bf16 [Cin/32 * KH*KW][Cout/16][64][8], the weight stream of the window-plane kernels
    (sc2_conv3x3_win_fwd / sc2_conv2x2_win_fwd; layout: include/sc2_bottleneck.h): k-step = slab * KH*KW + tap, one MFMA
    operand fragment = 1 KB contiguous, output rows permuted so that a lane ends up with eight consecutive channels."""
    _dev(w, 'w')
    cout, cin, kh, kw = w.shape
    assert cout % 32 == 0 and cin % 32 == 0
    # cout = 32 g + 8 a + 4 j + b   (frow = 4 a + b);   cin = 32 cb + 8 fq + e
    g = w.detach().to(torch.bfloat16).reshape(cout // 32, 4, 2, 4, cin // 32, 4, 8, kh * kw)   # g, a, j, b, cb, fq, e, tap
    g = g.permute(4, 7, 0, 2, 5, 1, 3, 6)                                                   # cb, tap, g, j, fq, a, b, e
    return g.contiguous().reshape(cin // 32 * kh * kw, cout // 16, 64, 8)


def pack_conv3x3_win(w):
    """[Cout, Cin, 3, 3] -> bf16 [Cin/32 * 9][Cout/16][64][8] for sc2_conv3x3_win_fwd."""
    assert tuple(w.shape[2:]) == (3, 3)
    return pack_conv_win(w)


def conv2x2_win_supported(x_shape, cout, kh, kw, stride, pad):
    """True if this conv runs on the window-plane decoder kernel (k2, s1, Cout 256, width 56 / pad 0 or 55 / pad 1)."""
    sh, sw = (stride, stride) if isinstance(stride, int) else stride
    ph, pw = (pad, pad) if isinstance(pad, int) else pad
    if not host_policy.conv2x2_win:      # A/B switch (tools/)
        return False
    N, H, W, Cin = x_shape
    if max(N * H * W * Cin, N * (H + 2 * ph - 1) * (W + 2 * pw - 1) * cout) * 2 >= 0x7FF00000:   # 32-bit buffer offsets
        return False
    return (kh, kw, sh, sw) == (2, 2, 1, 1) and ph == pw and bool(lib().sc2_conv2x2_win_supported(H, W, Cin, cout, ph))


def pack_conv2x2_win(w, gamma=None):
    """Weight stream of sc2_conv2x2_win_fwd: the conv's k-steps, then (fused GDN1) the effective gamma [256, 256] as a 1x1
    layer."""
    parts = [pack_conv_win(w)]
    if gamma is not None:
        assert tuple(gamma.shape) == (w.shape[0], w.shape[0])
        parts.append(pack_conv_win(gamma.reshape(gamma.shape[0], gamma.shape[1], 1, 1)))
    return torch.cat(parts).contiguous()


def conv2x2_win_tail_supported(x_shape):
    """True if the last decoder conv can take the head's first two 1x1 layers with it (55 x 55 input, sc2_conv2x2_win_tail_fwd)."""
    if not host_policy.conv2x2_win or not host_policy.w2_tail:      # A/B switches (tools/)
        return False
    N, H, W, Cin = x_shape
    if N * 56 * 56 * 256 * 2 >= 0x7FF00000 or N * H * W * Cin * 2 >= 0x7FF00000:
        return False
    return bool(lib().sc2_conv2x2_win_tail_supported(H, W, Cin))


def pack_conv2x2_win_tail(w, w1, wds):
    """Weight stream of sc2_conv2x2_win_tail_fwd: conv [256, Cin, 2, 2], then W1 [128, 256] (zero rows 128..255), then the two
    256-row halves of Wds [512, 256] as 1x1 layers."""
    assert tuple(w1.shape) == (128, 256) and tuple(wds.shape) == (512, 256) and w.shape[0] == 256
    w1p = torch.cat([w1, torch.zeros_like(w1)]).reshape(256, 256, 1, 1)
    parts = [pack_conv_win(w), pack_conv_win(w1p), pack_conv_win(wds[:256].reshape(256, 256, 1, 1)),
             pack_conv_win(wds[256:].reshape(256, 256, 1, 1))]
    return torch.cat(parts).contiguous()


def conv2x2_win_tail_fwd(x_nhwc, w_stream, bias1, bias_ds, want_y=False, tag=None):
    """-> (o1 [N,56,56,128], ods [N,28,28,512], y [N,56,56,256] or None); see include/sc2_bottleneck.h."""
    for t, name in ((x_nhwc, 'x'), (w_stream, 'w_stream'), (bias1, 'bias1'), (bias_ds, 'bias_ds')):
        _dev(t, name)
    assert x_nhwc.dtype == torch.bfloat16 and x_nhwc.dim() == 4 and x_nhwc.is_contiguous()
    N, H, W, Cin = x_nhwc.shape
    assert w_stream.dtype == torch.bfloat16 and w_stream.is_contiguous() and tuple(w_stream.shape) == (Cin // 32 * 4 + 24, 16, 64, 8)
    assert bias1.dtype == torch.float32 and bias1.numel() == 128 and bias_ds.dtype == torch.float32 and bias_ds.numel() == 512
    dev = x_nhwc.device
    o1 = torch.empty((N, 56, 56, 128), dtype=torch.bfloat16, device=dev)
    ods = torch.empty((N, 28, 28, 512), dtype=torch.bfloat16, device=dev)
    y = torch.empty((N, 56, 56, 256), dtype=torch.bfloat16, device=dev) if want_y else None
    with _timed(tag or 'conv2x2_win_tail'):
        _check(lib().sc2_conv2x2_win_tail_fwd(_ptr(x_nhwc), _ptr(w_stream), _ptr(bias1), _ptr(bias_ds), _ptr(y) if want_y else None,
                                              _ptr(o1), _ptr(ods), N, H, W, Cin, _stream()), 'conv2x2_win_tail_fwd')
    return o1, ods, y


def conv2x2_win_fwd(x_nhwc, w_frag, pad, beta=None, inverse=True, tag=None, out=None, channel0=0):
    """y = conv2x2(x, stride 1, pad) [-> GDN1 / inverse GDN1 when beta is given]; bf16 NHWC in / out, Cout 256.
    `out` [N, OH, OW, 512] with `channel0` 0 or 256: this launch's 256 channels go into that half of a 512-channel tensor (plain
    pad-1 convs only: the data gradient of a 512 -> 256 k2 layer is two such launches)."""
    for t, name in ((x_nhwc, 'x'), (w_frag, 'w_frag')):
        _dev(t, name)
    assert x_nhwc.dtype == torch.bfloat16 and x_nhwc.dim() == 4 and x_nhwc.is_contiguous()
    N, H, W, Cin = x_nhwc.shape
    fused = beta is not None
    assert w_frag.dtype == torch.bfloat16 and w_frag.is_contiguous() and \
        tuple(w_frag.shape) == (Cin // 32 * 4 + (8 if fused else 0), 16, 64, 8)
    if fused:
        _dev(beta, 'beta')
        assert beta.dtype == torch.float32 and beta.is_contiguous() and beta.numel() == 256
    pad = int(pad)
    if out is None:
        assert channel0 == 0
        out = torch.empty((N, H + 2 * pad - 1, W + 2 * pad - 1, 256), dtype=torch.bfloat16, device=x_nhwc.device)
    else:
        _dev(out, 'out')
        assert out.dtype == torch.bfloat16 and out.is_contiguous() and \
            tuple(out.shape) == (N, H + 2 * pad - 1, W + 2 * pad - 1, 512) and channel0 in (0, 256)
    with _timed(tag or 'conv2x2_win'):
        _check(lib().sc2_conv2x2_win_fwd(_ptr(x_nhwc), _ptr(w_frag), _ptr(beta) if fused else None, _ptr(out), N, H, W, Cin, pad,
                                         1 if fused else 0, 1 if inverse else 0, int(out.shape[3]), int(channel0), _stream()),
               'conv2x2_win_fwd')
    return out


def conv3x3_win_fwd(x_nhwc, w_frag, bias, relu=False, tag=None, stride=1, mask=None):
    """y = act(conv3x3(x, stride 1 or 2, pad 1) + bias); bf16 NHWC in / out; w_frag = pack_conv3x3_win(w).
    mask (stride 1, bf16 like y, no relu): y = mask > 0 ? conv + bias : 0 -- the gradient through a ReLU whose output is `mask`."""
    for t, name in ((x_nhwc, 'x'), (w_frag, 'w_frag'), (bias, 'bias')):
        _dev(t, name)
    assert x_nhwc.dtype == torch.bfloat16 and x_nhwc.dim() == 4 and x_nhwc.is_contiguous()
    N, H, W, Cin = x_nhwc.shape
    cout = w_frag.shape[1] * 16
    assert w_frag.dtype == torch.bfloat16 and w_frag.is_contiguous() and tuple(w_frag.shape) == (Cin // 32 * 9, cout // 16, 64, 8)
    assert bias.dtype == torch.float32 and bias.is_contiguous() and bias.numel() == cout
    stride = int(stride)
    if stride == 2:
        out = torch.empty((N, H // 2, W // 2, cout), dtype=torch.bfloat16, device=x_nhwc.device)
        with _timed(tag or 'conv3x3s2_win'):
            _check(lib().sc2_conv3x3s2_win_fwd(_ptr(x_nhwc), _ptr(w_frag), _ptr(bias), _ptr(out), N, H, W, Cin, cout,
                                               1 if relu else 0, _stream()), 'conv3x3s2_win_fwd')
        return out
    assert stride == 1
    out = torch.empty((N, H, W, cout), dtype=torch.bfloat16, device=x_nhwc.device)
    if mask is not None:
        _dev(mask, 'mask')
        assert not relu and mask.dtype == torch.bfloat16 and mask.is_contiguous() and tuple(mask.shape) == tuple(out.shape)
    with _timed(tag or 'conv3x3_win'):
        _check(lib().sc2_conv3x3_win_fwd(_ptr(x_nhwc), _ptr(w_frag), _ptr(bias), _ptr(mask), _ptr(out), N, H, W, Cin, cout,
                                         1 if relu else 0, _stream()), 'conv3x3_win_fwd')
    return out


def conv1x1_stream_supported(cin, cout, kh, kw, stride, pad):
    sh, sw = (stride, stride) if isinstance(stride, int) else stride
    ph, pw = (pad, pad) if isinstance(pad, int) else pad
    if not host_policy.conv_stream:      # A/B switch (tools/)
        return False
    return kh == 1 and kw == 1 and sh == sw and ph == 0 and pw == 0 and \
        bool(lib().sc2_conv1x1_stream_supported(cin, cout, sh))


def conv1x1_stream_mask_supported(cin, cout, stride):
    return host_policy.conv_stream and bool(lib().sc2_conv1x1_stream_mask_supported(cin, cout, int(stride)))


def conv1x1_stream_fwd(x_nhwc, w_frag, bias, stride=1, residual=None, relu=False, tag=None, mask=None):
    """y = act(conv1x1(x) + bias [+ residual]) on the persistent streaming kernel; bf16 NHWC in / out."""
    for t, name in ((x_nhwc, 'x'), (w_frag, 'w_frag'), (bias, 'bias')):
        _dev(t, name)
    assert x_nhwc.dtype == torch.bfloat16 and x_nhwc.dim() == 4 and x_nhwc.is_contiguous()
    N, H, W, Cin = x_nhwc.shape
    cout = w_frag.shape[0] * 16
    assert w_frag.dtype == torch.bfloat16 and w_frag.is_contiguous() and tuple(w_frag.shape) == (cout // 16, Cin // 32, 64, 8)
    assert bias.dtype == torch.float32 and bias.is_contiguous() and bias.numel() == cout
    OH, OW = (H - 1) // stride + 1, (W - 1) // stride + 1
    out = torch.empty((N, OH, OW, cout), dtype=torch.bfloat16, device=x_nhwc.device)
    if residual is not None:
        _dev(residual, 'residual')
        assert residual.dtype == torch.bfloat16 and residual.is_contiguous() and tuple(residual.shape) == tuple(out.shape)
    if mask is not None:
        _dev(mask, 'mask')
        assert not relu and mask.dtype == torch.bfloat16 and mask.is_contiguous() and tuple(mask.shape) == tuple(out.shape)
    with _timed(tag or 'conv1x1_stream'):
        _check(lib().sc2_conv1x1_stream_fwd(_ptr(x_nhwc), _ptr(w_frag), _ptr(bias), _ptr(residual), _ptr(mask), _ptr(out), N, H, W,
                                            Cin, cout, int(stride), 1 if relu else 0, _stream()), 'conv1x1_stream_fwd')
    return out


def conv1x1_pair_supported(k1, c, n2):
    """True if conv3 (k1 -> c, + residual + ReLU) of one Bottleneck block and conv1 (c -> n2, + ReLU) of the next run as one
    launch (sc2_conv1x1_pair_fwd); SC2_CONV1X1_PAIR=0: A/B switch."""
    return host_policy.conv1x1_pair and bool(lib().sc2_conv1x1_pair_supported(k1, c, n2))


def conv1x1_pair_fwd(o_nhwc, w3_frag, b3, identity, w1_frag, b1, tag=None):
    """h = relu(conv1x1(o; W3) + b3 + identity); u = relu(conv1x1(h; W1) + b1): -> (h [N,H,W,C], u [N,H,W,N2]), bf16 NHWC."""
    for t, name in ((o_nhwc, 'o'), (w3_frag, 'w3_frag'), (b3, 'b3'), (identity, 'identity'), (w1_frag, 'w1_frag'), (b1, 'b1')):
        _dev(t, name)
        assert t.is_contiguous(), name
    N, H, W, K1 = o_nhwc.shape
    C, N2 = w3_frag.shape[0] * 16, w1_frag.shape[0] * 16
    assert o_nhwc.dtype == torch.bfloat16 and identity.dtype == torch.bfloat16 and tuple(identity.shape) == (N, H, W, C)
    assert tuple(w3_frag.shape) == (C // 16, K1 // 32, 64, 8) and tuple(w1_frag.shape) == (N2 // 16, C // 32, 64, 8)
    assert b3.dtype == torch.float32 and b3.numel() == C and b1.dtype == torch.float32 and b1.numel() == N2
    h = torch.empty((N, H, W, C), dtype=torch.bfloat16, device=o_nhwc.device)
    u = torch.empty((N, H, W, N2), dtype=torch.bfloat16, device=o_nhwc.device)
    with _timed(tag or 'conv1x1_pair'):
        _check(lib().sc2_conv1x1_pair_fwd(_ptr(o_nhwc), _ptr(w3_frag), _ptr(b3), _ptr(identity), _ptr(h), _ptr(w1_frag), _ptr(b1),
                                          _ptr(u), N * H * W, K1, C, N2, _stream()), 'conv1x1_pair_fwd')
    return h, u


def conv2x2_c48_supported(x_shape, cout, kh, kw, stride, pad):
    """True if this conv runs on the streaming kernel of the FP encoder's last layer (Cin 48, k2, s1, p0, Cout <= 32)."""
    sh, sw = (stride, stride) if isinstance(stride, int) else stride
    ph, pw = (pad, pad) if isinstance(pad, int) else pad
    if not host_policy.conv_c48:      # A/B switch (tools/)
        return False
    N, H, W, Cin = x_shape
    if N * H * W * Cin * 2 >= 0x7FF00000:
        return False
    return (kh, kw, sh, sw, ph, pw) == (2, 2, 1, 1, 0, 0) and bool(lib().sc2_conv2x2_c48_supported(H, W, Cin, cout))


def pack_conv2x2_c48(w):
    """[Cout <= 32, 48, 2, 2] -> bf16 fragment blocks [2][6][64][8] of W[co][kh*96 + kw*48 + ci] (rows >= Cout zero)."""
    _dev(w, 'w')
    cout = w.shape[0]
    assert tuple(w.shape[1:]) == (48, 2, 2) and cout <= 32
    w2d = torch.zeros((32, 192), dtype=torch.float32, device=w.device)
    w2d[:cout] = w.detach().float().permute(0, 2, 3, 1).reshape(cout, 192)
    return pack_weight_fragments(w2d)


def conv2x2_c48_fwd(x_nhwc, w_frag, cout, medians=None, tag=None, out=None):
    """x bf16 [N,H,W,48] -> f32 latent [N,cout,H-1,W-1], or (medians given) the int32 symbols round(y - median).
    `out`: a contiguous tensor of that dtype and element count to write into (e.g. a row block of a coder-group buffer)."""
    for t, name in ((x_nhwc, 'x'), (w_frag, 'w_frag')):
        _dev(t, name)
    assert x_nhwc.dtype == torch.bfloat16 and x_nhwc.dim() == 4 and x_nhwc.is_contiguous()
    N, H, W, Cin = x_nhwc.shape
    assert w_frag.dtype == torch.bfloat16 and w_frag.is_contiguous() and tuple(w_frag.shape) == (2, 6, 64, 8)
    sym = medians is not None
    if sym:
        _dev(medians, 'medians')
        assert medians.dtype == torch.float32 and medians.is_contiguous() and medians.numel() == cout
    dt = torch.int32 if sym else torch.float32
    if out is None:
        out = torch.empty((N, cout, H - 1, W - 1), dtype=dt, device=x_nhwc.device)
    else:
        _dev(out, 'out')
        assert out.dtype == dt and out.is_contiguous() and out.numel() == N * cout * (H - 1) * (W - 1)
        out = out.view(N, cout, H - 1, W - 1)
    with _timed(tag or 'conv2x2_c48'):
        _check(lib().sc2_conv2x2_c48_fwd(_ptr(x_nhwc), _ptr(w_frag), _ptr(medians) if sym else None, _ptr(out), N, H, W, Cin, cout,
                                         1 if sym else 0, _stream()), 'conv2x2_c48_fwd')
    return out


def conv1x1_win_supported(cin, cout, kh, kw, stride, pad):
    """True if this 1x1 conv can run on the window-plane 1x1 kernel (Cin, Cout multiples of 128, stride 1 or 2)."""
    sh, sw = (stride, stride) if isinstance(stride, int) else stride
    ph, pw = (pad, pad) if isinstance(pad, int) else pad
    return (kh, kw, ph, pw) == (1, 1, 0, 0) and sh == sw and bool(lib().sc2_conv1x1_win_supported(cin, cout, sh))


def conv1x1_win_fwd(x_nhwc, w_frag, bias, stride=1, residual=None, relu=False, tag=None, mask=None):
    """y = act(conv1x1(x) + bias [+ residual]) on the window-plane 1x1 kernel; bf16 NHWC in / out;
    w_frag = pack_conv_win(w.reshape(Cout, Cin, 1, 1)).  mask (bf16 like y, no relu): y = mask > 0 ? value : 0."""
    for t, name in ((x_nhwc, 'x'), (w_frag, 'w_frag'), (bias, 'bias')):
        _dev(t, name)
    assert x_nhwc.dtype == torch.bfloat16 and x_nhwc.dim() == 4 and x_nhwc.is_contiguous()
    N, H, W, Cin = x_nhwc.shape
    cout = w_frag.shape[1] * 16
    assert w_frag.dtype == torch.bfloat16 and w_frag.is_contiguous() and tuple(w_frag.shape) == (Cin // 32, cout // 16, 64, 8)
    assert bias.dtype == torch.float32 and bias.is_contiguous() and bias.numel() == cout
    stride = int(stride)
    out = torch.empty((N, (H - 1) // stride + 1, (W - 1) // stride + 1, cout), dtype=torch.bfloat16, device=x_nhwc.device)
    if residual is not None:
        _dev(residual, 'residual')
        assert residual.dtype == torch.bfloat16 and residual.is_contiguous() and tuple(residual.shape) == tuple(out.shape)
    if mask is not None:
        _dev(mask, 'mask')
        assert not relu and mask.dtype == torch.bfloat16 and mask.is_contiguous() and tuple(mask.shape) == tuple(out.shape)
    with _timed(tag or 'conv1x1_win'):
        _check(lib().sc2_conv1x1_win_fwd(_ptr(x_nhwc), _ptr(w_frag), _ptr(bias), _ptr(residual), _ptr(mask), _ptr(out), N, H, W, Cin,
                                         cout, stride, 1 if relu else 0, _stream()), 'conv1x1_win_fwd')
    return out


def pack_gamma_fragments(gamma):
    """Effective gamma [C, C] (row = output channel) -> bf16 fragment-major [C/16][C/32][64][8]: entry
    (jt, ks, lane = fq*16 + frow, e) = gamma[jt*16 + frow, ks*32 + fq*8 + e], so that one MFMA operand fragment
    (16 channels x 32 k) is 1 KB contiguous.  Layout of the `gamma_frag` argument of sc2_conv2x2_gdn512_fwd."""
    _dev(gamma, 'gamma')
    C = gamma.shape[0]
    assert gamma.dim() == 2 and gamma.shape[1] == C and C % 32 == 0
    g = gamma.detach().to(torch.bfloat16).reshape(C // 16, 16, C // 32, 4, 8)      # jt, frow, ks, fq, e
    return g.permute(0, 2, 3, 1, 4).contiguous().reshape(C // 16, C // 32, 64, 8)


def conv2x2_gdn512_fwd(x_nhwc, w_packed, gamma_packed, beta, inverse, tag=None, want_t=False):
    """y = GDN1_512(conv2x2(x)) (s1, p1) in one launch; x bf16 [N,H,W,Cin] -> bf16 [N,H+1,W+1,512].
    gamma_packed: pack_gamma_fragments(effective gamma)."""
    for t, name in ((x_nhwc, 'x'), (w_packed, 'w_packed'), (gamma_packed, 'gamma_packed'), (beta, 'beta')):
        _dev(t, name)
    assert x_nhwc.dtype == torch.bfloat16 and x_nhwc.dim() == 4 and x_nhwc.is_contiguous()
    N, H, W, Cin = x_nhwc.shape
    assert w_packed.dtype == torch.bfloat16 and w_packed.is_contiguous() and w_packed.shape[0] == 512
    assert gamma_packed.dtype == torch.bfloat16 and gamma_packed.is_contiguous() and tuple(gamma_packed.shape) == (32, 16, 64, 8)
    assert beta.dtype == torch.float32 and beta.is_contiguous() and beta.numel() == 512
    out = torch.empty((N, H + 1, W + 1, 512), dtype=torch.bfloat16, device=x_nhwc.device)
    t = torch.empty_like(out) if want_t else None        # (training: the conv output in front of the GDN)
    with _timed(tag or 'conv2x2_gdn512'):
        _check(lib().sc2_conv2x2_gdn512_fwd(_ptr(x_nhwc), _ptr(w_packed), w_packed.shape[1], _ptr(gamma_packed), _ptr(beta),
                                            _ptr(out), _ptr(t), N, H, W, Cin, 1 if inverse else 0, _stream()), 'conv2x2_gdn512_fwd')
    return (out, t) if want_t else out


# --------------------------------------------------------------------------------------------- #
# entropy bottleneck
# --------------------------------------------------------------------------------------------- #
def eb_forward(y, params, mode, noise=None, lik_bound=1e-9, want_y_hat=True, want_nhwc=False, want_lik=True,
               want_bits=False):
    """y: f32 [N,C,*spatial] contiguous (NCHW). Returns (y_hat, y_hat_nhwc_bf16, lik, bits_partial)."""
    _dev(y, 'y')
    _dev(params, 'params')
    assert y.dtype == torch.float32 and y.is_contiguous() and params.dtype == torch.float32
    N, C = y.shape[0], y.shape[1]
    HW = y.numel() // (N * C)
    assert params.shape == (C, EB_PARAM_STRIDE) and params.is_contiguous()
    if noise is not None:
        _dev(noise, 'noise')
        assert noise.shape == y.shape and noise.dtype == torch.float32 and noise.is_contiguous()
    y_hat = torch.empty_like(y) if want_y_hat else None
    nhwc = torch.empty((N,) + tuple(y.shape[2:]) + (C,), dtype=torch.bfloat16, device=y.device) if want_nhwc else None
    lik = torch.empty_like(y) if want_lik else None
    nb = lib().sc2_eb_bits_partial_len(N, C, HW) if want_bits else 0
    bits = torch.empty((nb,), dtype=torch.float32, device=y.device) if want_bits else None
    _check(lib().sc2_eb_forward(_ptr(y), _ptr(noise), _ptr(params), N, C, HW, int(mode), float(lik_bound),
                                _ptr(y_hat), _ptr(nhwc), _ptr(lik), _ptr(bits), nb, _stream()), 'eb_forward')
    return y_hat, nhwc, lik, bits


def eb_backward(y, params, mode, noise, g_yhat, g_lik, lik_bound=1e-9):
    """Returns (g_y f32 like y, g_params f32 [C, 64]) -- the per-workgroup partial rows are summed here."""
    _dev(y, 'y')
    N, C = y.shape[0], y.shape[1]
    HW = y.numel() // (N * C)
    for t in (noise, g_yhat, g_lik):
        if t is not None:
            _dev(t, 'grad/noise')
            assert t.dtype == torch.float32 and t.is_contiguous() and t.numel() == y.numel()
    g_y = torch.empty_like(y)
    n_partial = lib().sc2_eb_bits_partial_len(N, C, HW)
    part = torch.empty((n_partial, EB_PARAM_STRIDE), dtype=torch.float32, device=y.device)
    _check(lib().sc2_eb_backward(_ptr(y), _ptr(noise), _ptr(params), N, C, HW, int(mode), float(lik_bound),
                                 _ptr(g_yhat), _ptr(g_lik), _ptr(g_y), _ptr(part), n_partial, _stream()),
           'eb_backward')
    g_params = part.view(N, C, n_partial // (N * C), EB_PARAM_STRIDE).sum(dim=(0, 2))
    return g_y, g_params


def eb_symbols(y, medians):
    _dev(y, 'y')
    _dev(medians, 'medians')
    assert y.dtype == torch.float32 and y.is_contiguous() and medians.dtype == torch.float32
    N, C = y.shape[0], y.shape[1]
    HW = y.numel() // (N * C)
    assert medians.numel() == C and medians.is_contiguous()
    sym = torch.empty(y.shape, dtype=torch.int32, device=y.device)
    _check(lib().sc2_eb_symbols(_ptr(y), _ptr(medians), N, C, HW, _ptr(sym), _stream()), 'eb_symbols')
    return sym


def eb_dequantize(symbols, medians, want_f32=True, want_nhwc=False):
    _dev(symbols, 'symbols')
    _dev(medians, 'medians')
    assert symbols.dtype == torch.int32 and symbols.is_contiguous()
    N, C = symbols.shape[0], symbols.shape[1]
    HW = symbols.numel() // (N * C)
    f32 = torch.empty(symbols.shape, dtype=torch.float32, device=symbols.device) if want_f32 else None
    nhwc = torch.empty((N,) + tuple(symbols.shape[2:]) + (C,), dtype=torch.bfloat16,
                       device=symbols.device) if want_nhwc else None
    _check(lib().sc2_eb_dequantize(_ptr(symbols), _ptr(medians), N, C, HW, _ptr(f32), _ptr(nhwc), _stream()),
           'eb_dequantize')
    return f32, nhwc


# --------------------------------------------------------------------------------------------- #
# GaussianConditional (hyperprior bottlenecks)
# --------------------------------------------------------------------------------------------- #
def _img_slice(t, ref, name):
    """f32 tensor with ref's shape that is dense inside each batch item (possibly a channel slice of a wider
    tensor, e.g. gaussian_params.chunk(2, 1)) -> (tensor, per-image element stride)."""
    _dev(t, name)
    assert t.dtype == torch.float32 and tuple(t.shape) == tuple(ref.shape), name
    if t.dim() < 2 or not t[0].is_contiguous():
        t = t.contiguous()
    return t, (t.stride(0) if t.shape[0] > 1 else t[0].numel())


def gc_forward(y, scales, means=None, noise=None, mode=EB_DEQUANTIZE, scale_bound=0.11, lik_bound=1e-9,
               want_y_hat=True, want_lik=True):
    """GaussianConditional.forward: y f32 [N,C,*spatial] contiguous -> (y_hat, likelihoods)."""
    _dev(y, 'y')
    assert y.dtype == torch.float32 and y.is_contiguous()
    N, chw = y.shape[0], y[0].numel()
    s_stride = m_stride = 0
    if scales is not None:
        scales, s_stride = _img_slice(scales, y, 'scales')
    if means is not None:
        means, m_stride = _img_slice(means, y, 'means')
    if noise is not None:
        _dev(noise, 'noise')
        assert noise.shape == y.shape and noise.dtype == torch.float32 and noise.is_contiguous()
    y_hat = torch.empty_like(y) if want_y_hat else None
    lik = torch.empty_like(y) if want_lik else None
    _check(lib().sc2_gc_forward(_ptr(y), _ptr(scales), s_stride, _ptr(means), m_stride, _ptr(noise), N, chw, int(mode),
                                float(scale_bound), float(lik_bound), _ptr(y_hat), _ptr(lik), _stream()), 'gc_forward')
    return y_hat, lik


def gc_backward(y, scales, means, noise, g_yhat, g_lik, scale_bound=0.11, lik_bound=1e-9):
    """Backward of gc_forward in noise mode -> (g_y, g_scales, g_means or None), dense f32 like y."""
    _dev(y, 'y')
    assert y.dtype == torch.float32 and y.is_contiguous()
    N, chw = y.shape[0], y[0].numel()
    scales, s_stride = _img_slice(scales, y, 'scales')
    m_stride = 0
    if means is not None:
        means, m_stride = _img_slice(means, y, 'means')
    for t in (noise, g_yhat, g_lik):
        if t is not None:
            _dev(t, 'grad/noise')
            assert t.dtype == torch.float32 and t.is_contiguous() and t.shape == y.shape
    g_y, g_s = torch.empty_like(y), torch.empty_like(y)
    g_m = torch.empty_like(y) if means is not None else None
    _check(lib().sc2_gc_backward(_ptr(y), _ptr(scales), s_stride, _ptr(means), m_stride, _ptr(noise), N, chw,
                                 float(scale_bound), float(lik_bound), _ptr(g_yhat), _ptr(g_lik), _ptr(g_y), _ptr(g_s),
                                 _ptr(g_m), _stream()), 'gc_backward')
    return g_y, g_s, g_m


def gc_symbols_indexes(y, scales, means, scale_table, scale_bound=0.11, want_symbols=True, want_indexes=True):
    """-> (symbols int32 like y or None, indexes int32 like scales or None)."""
    ref = y if y is not None else scales
    _dev(ref, 'y')
    N, chw = ref.shape[0], ref[0].numel()
    s_stride = m_stride = 0
    if y is not None:
        assert y.dtype == torch.float32 and y.is_contiguous()
    if scales is not None:
        scales, s_stride = _img_slice(scales, ref, 'scales')
    if means is not None:
        means, m_stride = _img_slice(means, ref, 'means')
    n_table = 0
    if want_indexes:
        _dev(scale_table, 'scale_table')
        assert scale_table.dtype == torch.float32 and scale_table.is_contiguous()
        n_table = scale_table.numel()
    sym = torch.empty(ref.shape, dtype=torch.int32, device=ref.device) if want_symbols else None
    idx = torch.empty(ref.shape, dtype=torch.int32, device=ref.device) if want_indexes else None
    _check(lib().sc2_gc_symbols_indexes(_ptr(y), _ptr(scales), s_stride, _ptr(means), m_stride, N, chw,
                                        _ptr(scale_table) if want_indexes else None, n_table, float(scale_bound),
                                        _ptr(sym), _ptr(idx), _stream()), 'gc_symbols_indexes')
    return sym, idx


def gc_dequantize(symbols, means=None, want_f32=True, want_nhwc=False):
    """symbols int32 [N,C,*spatial] -> (y_hat f32 NCHW or None, y_hat bf16 NHWC or None); y_hat = symbols + means."""
    _dev(symbols, 'symbols')
    assert symbols.dtype == torch.int32 and symbols.is_contiguous() and symbols.dim() >= 3
    N, C = symbols.shape[0], symbols.shape[1]
    HW = symbols[0, 0].numel()
    m_stride = 0
    if means is not None:
        _dev(means, 'means')
        assert means.dtype == torch.float32 and tuple(means.shape) == tuple(symbols.shape)
        if not means[0].is_contiguous():
            means = means.contiguous()
        m_stride = means.stride(0) if N > 1 else means[0].numel()
    y_hat = torch.empty(symbols.shape, dtype=torch.float32, device=symbols.device) if want_f32 else None
    nhwc = torch.empty((N,) + tuple(symbols.shape[2:]) + (C,), dtype=torch.bfloat16, device=symbols.device) \
        if want_nhwc else None
    _check(lib().sc2_gc_dequantize(_ptr(symbols), _ptr(means), m_stride, N, C, HW, _ptr(y_hat), _ptr(nhwc), _stream()),
           'gc_dequantize')
    return y_hat, nhwc


# --------------------------------------------------------------------------------------------- #
# CDF + rANS
# --------------------------------------------------------------------------------------------- #
# --------------------------------------------------------------------------------------------- #
# element-wise pieces of the distillation step: csrc/loss.hip
# --------------------------------------------------------------------------------------------- #
def _same_dense_bf16(*ts):
    """True if the tensors are bf16 device tensors of one shape AND one dense memory layout (element i of each is the same
    logical element), so that an element-wise kernel may walk their storage linearly."""
    a = ts[0]
    if not all(isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.bfloat16 and t.shape == a.shape and
               t.stride() == a.stride() for t in ts):
        return False
    return (a.is_contiguous() or (a.dim() == 4 and a.is_contiguous(memory_format=torch.channels_last))) and a.numel() % 8 == 0


def mse_sum(x, y):
    """sum((x - y)^2) as an f32 device scalar; x, y: bf16, same shape and dense layout (_same_dense_bf16)."""
    assert _same_dense_bf16(x, y)
    n = x.numel()
    partial = torch.empty((int(lib().sc2_mse_partial_len(n)),), dtype=torch.float32, device=x.device)
    with _timed('mse_sum'):
        _check(lib().sc2_mse_sum_bf16(_ptr(x), _ptr(y), n, _ptr(partial), _stream()), 'mse_sum')
    return partial.double().sum().float()


def mse_grad(x, y, scale):
    """bf16(2 * scale * (x - y)) laid out like x; scale: f32 device scalar tensor."""
    assert _same_dense_bf16(x, y) and scale.is_cuda and scale.dtype == torch.float32 and scale.numel() == 1
    gx = torch.empty_like(x)
    with _timed('mse_grad'):
        _check(lib().sc2_mse_grad_bf16(_ptr(x), _ptr(y), x.numel(), _ptr(scale), _ptr(gx), _stream()), 'mse_grad')
    return gx


def relu_bwd(g, out, add=None):
    """(g [+ add]) * (out > 0), bf16, laid out like g (all operands one shape and dense layout)."""
    ops = (g, out) if add is None else (g, out, add)
    assert _same_dense_bf16(*ops)
    gi = torch.empty_like(g)
    with _timed('relu_bwd'):
        _check(lib().sc2_relu_bwd_bf16(_ptr(g), _ptr(out), _ptr(add), g.numel(), _ptr(gi), _stream()), 'relu_bwd')
    return gi


def relu_bwd_mse(g, out, t, scale, relu=True):
    """([g] + 2 * scale * (out - t)) [* (out > 0)], bf16 like out; g may be None; scale: f32 device scalar tensor."""
    ops = (out, t) if g is None else (g, out, t)
    assert _same_dense_bf16(*ops) and scale.is_cuda and scale.dtype == torch.float32 and scale.numel() == 1
    gi = torch.empty_like(out)
    with _timed('relu_bwd'):
        _check(lib().sc2_relu_bwd_mse_bf16(_ptr(g), _ptr(out), _ptr(t), _ptr(scale), out.numel(), 1 if relu else 0, _ptr(gi), _stream()),
               'relu_bwd_mse')
    return gi


# --------------------------------------------------------------------------------------------- #
# reference-precision (f32 operand) convolution / GDN1: csrc/conv_f32.hip
# --------------------------------------------------------------------------------------------- #
def nchw_f32_to_nhwc_f32(x, cpad=None):
    """x: f32 [N,C,H,W] -> f32 [N,H,W,Cpad] (channels >= C zero)."""
    _dev(x, 'x')
    assert x.dtype == torch.float32 and x.dim() == 4
    x = x.contiguous()
    N, C, H, W = x.shape
    cpad = (C + 3) // 4 * 4 if cpad is None else cpad
    y = torch.empty((N, H, W, cpad), dtype=torch.float32, device=x.device)
    _check(lib().sc2_nchw_f32_to_nhwc_f32(_ptr(x), _ptr(y), N, C, H, W, cpad, _stream()), 'nchw_f32_to_nhwc_f32')
    return y


def conv_f32_fused_gdn_supported(cout):
    """conv + GDN1 in one f32 launch needs every channel of a pixel in one wave (Cout <= 96) AND a gamma stream as long as the
    chunk is wide: ceil(Cout / 16) * 16 == chunk width (96 / 48 / 32 / 20 ... yes; 64 or 16 no: two launches)."""
    cout = int(cout)
    return cout <= 96 and (cout + 15) // 16 * 16 == int(lib().sc2_conv_f32_chunk_channels(cout))


def pack_conv_f32(weight, cin_pad=None):
    """Conv weight [Cout, Cin, KH, KW] (any float dtype) -> the f32 fragment-major stream of sc2_conv2d_f32_fwd:
    [chunks][steps][NT][64 lanes][4], entry (ch, s, nt, lane = q*16 + r, j) = W[ch*cc + nt*16 + r][16 s + 4 q + j] with
    k = (kh*KW + kw)*cin_pad + ci."""
    w = weight.detach().float()
    Cout, Cin, KH, KW = w.shape
    cin_pad = (Cin + 3) // 4 * 4 if cin_pad is None else cin_pad
    cc = int(lib().sc2_conv_f32_chunk_channels(Cout))
    chunks, NT = (Cout + cc - 1) // cc, cc // 16
    K = KH * KW * cin_pad
    steps = (K + 15) // 16
    m = torch.zeros((chunks * cc, KH, KW, cin_pad), dtype=torch.float32, device=w.device)
    m[:Cout, :, :, :Cin] = w.permute(0, 2, 3, 1)
    m = torch.nn.functional.pad(m.reshape(chunks * cc, K), (0, steps * 16 - K))
    m = m.reshape(chunks, NT, 16, steps, 4, 4)            # [ch, nt, r, s, q, j]
    return m.permute(0, 3, 1, 4, 2, 5).contiguous()       # [ch, s, nt, q, r, j]


def conv2d_f32_fwd(x_nhwc, w_frag, cout, kh, kw, stride, padding, a_op=AOP_NONE, epilogue=EPI_NONE, out_format=None,
                   ep_x=None, ep_beta=None, out=None, tag=None, cin_real=0, x_is_nchw_rgb=False):
    """x_nhwc: f32 [N,H,W,Cin] (Cin % 4 == 0) -> per out_format: OUT_F32_NHWC [N,OH,OW,Cout] (default), OUT_F32_NCHW
    [N,Cout,OH,OW], OUT_I32_NCHW_SYM int32 [N,Cout,OH,OW] (ep_beta = medians).  cin_real: the module's channel count when x_nhwc
    carries zero padding channels (3 of 4: the kernel skips the padding channel's products).  x_is_nchw_rgb: x_nhwc is the f32 NCHW
    image [N,3,H,W] itself (w_frag packed for cin_pad 4 as always): the kernel reads the three planes in place."""
    _dev(x_nhwc, 'x')
    assert x_nhwc.dtype == torch.float32 and x_nhwc.dim() == 4 and x_nhwc.is_contiguous()
    assert w_frag.dtype == torch.float32 and w_frag.is_contiguous()
    out_format = OUT_F32_NHWC if out_format is None else out_format
    if x_is_nchw_rgb:
        N, c3, H, W = x_nhwc.shape
        assert c3 == 3
        Cin, cin_real = 4, 3
    else:
        N, H, W, Cin = x_nhwc.shape
    sh = stride[0] if isinstance(stride, (tuple, list)) else stride
    ph = padding[0] if isinstance(padding, (tuple, list)) else padding
    OH, OW = (H + 2 * ph - kh) // sh + 1, (W + 2 * ph - kw) // sh + 1
    d = ConvDesc(N=N, H=H, W=W, Cin=Cin, Cout=cout, KH=kh, KW=kw, stride_h=sh, stride_w=sh, pad_h=ph, pad_w=ph, OH=OH, OW=OW,
                 a_op=a_op, epilogue=epilogue, out_format=out_format, Kpad=int(cin_real or 0), Cout_pad=0, out_H=0, out_W=0, out_stride_h=0,
                 out_stride_w=0, out_off_h=0, out_off_w=0, k_order=1 if x_is_nchw_rgb else 0)
    if out_format == OUT_F32_NHWC:
        y = torch.empty((N, OH, OW, cout), dtype=torch.float32, device=x_nhwc.device)
    elif out_format == OUT_F32_NCHW:
        y = torch.empty((N, cout, OH, OW), dtype=torch.float32, device=x_nhwc.device)
    else:
        assert out_format == OUT_I32_NCHW_SYM
        if out is not None:
            assert out.dtype == torch.int32 and out.is_contiguous() and out.numel() == N * cout * OH * OW
            y = out.view(N, cout, OH, OW)
        else:
            y = torch.empty((N, cout, OH, OW), dtype=torch.int32, device=x_nhwc.device)
    for t in (ep_x, ep_beta):
        if t is not None:
            _dev(t, 'epilogue operand')
            assert t.dtype == torch.float32 and t.is_contiguous()
    with _timed(tag or 'conv_f32'):
        _check(lib().sc2_conv2d_f32_fwd(ctypes.byref(d), _ptr(x_nhwc), _ptr(w_frag), _ptr(y), _ptr(ep_x), _ptr(ep_beta),
                                        _stream()), 'conv2d_f32_fwd')
    return y


def pmf_to_quantized_cdf(pmf, precision=16):
    """Host function. pmf: sequence / 1-D CPU tensor of floats -> torch.IntTensor (len+1)."""
    if isinstance(pmf, torch.Tensor):
        pmf = pmf.detach().cpu().tolist()
    n = len(pmf)
    arr = (ctypes.c_float * n)(*pmf)
    out = (ctypes.c_uint32 * (n + 1))()
    _check(lib().sc2_pmf_to_quantized_cdf(arr, n, int(precision), out), 'pmf_to_quantized_cdf')
    return torch.IntTensor(list(out))


def rans_max_bytes(n_sym):
    return int(lib().sc2_rans_max_bytes(int(n_sym)))


# ---- host range coder (csrc/rans_host.cpp): a few streams are coded faster by CPU cores than by GPU lanes ----------------
def host_coder_max_streams():
    """Stream count up to which compress() / decompress() use the library's HOST coder (0 = always the device coder)."""
    # measured (profiles/r03c_*): the device coder takes ~120 ns per symbol of ONE stream however few streams there are (one
    # lane each), a host thread ~15-25 ns; with up to 32 threads the host wins far beyond a handful of streams (16 streams
    # of 393 k symbols, the 513 x 513 batch: 85 ms on the device coder)
    return int(host_policy.host_coder_max_streams)


class HostRansTables(object):
    """Prepared CDF tables of the host coder (opaque handle of sc2_rans_host_tables_create), built from int32 tensors /
    arrays `cdfs` [n_cdfs, stride], `cdf_sizes` [n_cdfs], `offsets` [n_cdfs] wherever they live."""

    def __init__(self, cdfs, cdf_sizes, offsets):
        import numpy as np
        to_np = lambda t: np.ascontiguousarray((t.detach().cpu().numpy() if isinstance(t, torch.Tensor) else np.asarray(t)), dtype=np.int32)  # noqa: E731
        c, z, o = to_np(cdfs), to_np(cdf_sizes).reshape(-1), to_np(offsets).reshape(-1)
        assert c.ndim == 2 and z.shape[0] == c.shape[0] == o.shape[0]
        self.n_cdfs = int(c.shape[0])
        self._h = ctypes.c_void_p(0)
        _check(lib().sc2_rans_host_tables_create(c.ctypes.data, c.shape[0], c.shape[1], z.ctypes.data, o.ctypes.data,
                                                 ctypes.byref(self._h)), 'rans_host_tables_create')

    def __del__(self):
        h, self._h = getattr(self, '_h', None), None
        if h and _lib is not None:
            _lib.sc2_rans_host_tables_destroy(h)


def host_cores():
    """CPUs this process may reasonably load: its affinity mask, and -- under a launcher that starts several ranks on the node
    (LOCAL_WORLD_SIZE) -- no more than its share of the machine (eight ranks each spawning a coder thread per CPU of the node would
    oversubscribe it eightfold)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        ranks = max(1, int(os.environ.get('LOCAL_WORLD_SIZE', '1')))
    except ValueError:
        ranks = 1
    return max(1, min(n, (os.cpu_count() or n) // ranks))


def _host_threads(n_streams, cap=32):
    return max(1, min(int(n_streams), host_cores(), cap))


def rans_code_host(tables, symbols, index_div, dec_out, out_stride=None, threads=None, scratch=None):
    """HOST encode + decode of every row of `symbols` (int32 numpy [n_streams, n_sym], implicit indexes) in one library call;
    the decoded symbols land in `dec_out` (int32 numpy of the same shape, e.g. a view of a pinned tensor).
    -> (buf uint8 [n_streams, stride], offset, nbytes, status) numpy; status = encode | decode bits.
    `scratch`: a dict that keeps the 37 MB of byte rows between calls -- a fresh allocation is first touched by the coder's threads,
    one page fault per 4 KB under the process's memory-map lock: 3 - 4 ms of a 5 ms call on a 128-core host."""
    import numpy as np
    n_streams, n_sym = symbols.shape
    assert symbols.dtype == np.int32 and symbols.flags['C_CONTIGUOUS'] and dec_out.dtype == np.int32 and dec_out.flags['C_CONTIGUOUS']
    assert dec_out.shape == symbols.shape
    if out_stride is None:
        out_stride = 2 * n_sym + 64
    out_stride = (int(out_stride) + 3) // 4 * 4
    key = ('rows', n_streams, out_stride)
    if scratch is not None and key in scratch:
        buf = scratch[key]
    else:
        buf = np.empty((n_streams, out_stride // 4), dtype=np.uint32)
        if scratch is not None:
            buf.fill(0)         # (touch the pages once, here)
            scratch[key] = buf
    off = np.empty((n_streams,), dtype=np.int32)
    nb = np.empty((n_streams,), dtype=np.int32)
    st = np.empty((n_streams,), dtype=np.int32)
    n_thr = int(threads) if threads else _host_threads(n_streams, cap=64)
    _check(lib().sc2_rans_code_host(tables._h, symbols.ctypes.data, None, int(index_div), n_streams, n_sym, buf.ctypes.data, out_stride,
                                    off.ctypes.data, nb.ctypes.data, dec_out.ctypes.data, st.ctypes.data, n_thr), 'rans_code_host')
    return buf.view(np.uint8).reshape(n_streams, out_stride), off, nb, st


def clock_probe(n_workgroups=16, n_samples=64, period_us=20.0, stream=None):
    """DIAGNOSTIC (csrc/diag.hip): launches the clock probe on `stream` (default: the current one) and returns the int64 tensor
    [n_workgroups, n_samples, 3] = (s_memtime, s_memrealtime, XCC id) it fills; read it after a synchronize."""
    out = torch.zeros((n_workgroups, n_samples, 3), dtype=torch.int64, device='cuda')
    s = ctypes.c_void_p(stream.cuda_stream) if stream is not None else _stream()
    _check(lib().sc2_clock_probe(_ptr(out), n_workgroups, n_samples, int(round(period_us * 100.0)), s), 'clock_probe')
    return out


def rans_encode_host(tables, symbols, indexes=None, index_div=0, out_stride=None):
    """symbols: int32 numpy [n_streams, n_sym] (HOST) -> (list[bytes], status int32 numpy [n_streams])."""
    import numpy as np
    symbols = np.ascontiguousarray(symbols, dtype=np.int32)
    n_streams, n_sym = symbols.shape
    if indexes is not None:
        indexes = np.ascontiguousarray(indexes, dtype=np.int32)
        assert indexes.shape == symbols.shape
    if out_stride is None:
        out_stride = 2 * n_sym + 64
    out_stride = (int(out_stride) + 3) // 4 * 4
    buf = np.empty((n_streams, out_stride // 4), dtype=np.uint32)
    off = np.empty((n_streams,), dtype=np.int32)
    nb = np.empty((n_streams,), dtype=np.int32)
    st = np.empty((n_streams,), dtype=np.int32)
    _check(lib().sc2_rans_encode_host(tables._h, symbols.ctypes.data, indexes.ctypes.data if indexes is not None else None,
                                      int(index_div), n_streams, n_sym, buf.ctypes.data, out_stride, off.ctypes.data,
                                      nb.ctypes.data, st.ctypes.data, _host_threads(n_streams)), 'rans_encode_host')
    raw = buf.view(np.uint8).reshape(n_streams, out_stride)
    return [raw[i, int(off[i]):int(off[i]) + int(nb[i])].tobytes() for i in range(n_streams)], st


def rans_decode_host(tables, strings, n_sym, indexes=None, index_div=0):
    """strings: list[bytes] -> (int32 numpy [n_streams, n_sym] (HOST), status int32 numpy [n_streams])."""
    import numpy as np
    n_streams = len(strings)
    stride = (max([len(q) for q in strings] + [8]) + 3) // 4 * 4
    buf = np.zeros((n_streams, stride // 4), dtype=np.uint32)
    raw = buf.view(np.uint8).reshape(n_streams, stride)
    nb = np.empty((n_streams,), dtype=np.int32)
    for i, q in enumerate(strings):
        raw[i, :len(q)] = np.frombuffer(q, dtype=np.uint8)
        nb[i] = len(q)
    off = np.zeros((n_streams,), dtype=np.int32)
    if indexes is not None:
        indexes = np.ascontiguousarray(indexes, dtype=np.int32)
        assert indexes.shape == (n_streams, n_sym)
    sym = np.empty((n_streams, int(n_sym)), dtype=np.int32)
    st = np.empty((n_streams,), dtype=np.int32)
    _check(lib().sc2_rans_decode_host(tables._h, buf.ctypes.data, stride, off.ctypes.data, nb.ctypes.data,
                                      indexes.ctypes.data if indexes is not None else None, int(index_div), n_streams,
                                      int(n_sym), sym.ctypes.data, st.ctypes.data, _host_threads(n_streams)), 'rans_decode_host')
    return sym, st


def rans_encode_batch(symbols, cdfs, cdf_sizes, offsets, indexes=None, index_div=0, out_stride=None):
    """symbols: i32 [n_streams, n_sym] (device).  Returns (buf u8 [n_streams, stride], offset, nbytes, status)."""
    _dev(symbols, 'symbols')
    for name, t in (('cdfs', cdfs), ('cdf_sizes', cdf_sizes), ('offsets', offsets)):
        _dev(t, name)
        assert t.dtype == torch.int32 and t.is_contiguous(), name
    assert symbols.dtype == torch.int32 and symbols.dim() == 2 and symbols.is_contiguous()
    n_streams, n_sym = symbols.shape
    if indexes is not None:
        _dev(indexes, 'indexes')
        assert indexes.shape == symbols.shape and indexes.dtype == torch.int32 and indexes.is_contiguous()
    if out_stride is None:
        out_stride = 2 * n_sym + 64  # 16 bits / symbol + flush; overflow is reported via status
    out_stride = (int(out_stride) + 3) // 4 * 4
    dev = symbols.device
    buf = torch.empty((n_streams, out_stride), dtype=torch.uint8, device=dev)
    off = torch.empty((n_streams,), dtype=torch.int32, device=dev)
    nb = torch.empty((n_streams,), dtype=torch.int32, device=dev)
    st = torch.empty((n_streams,), dtype=torch.int32, device=dev)
    ws_bytes = int(lib().sc2_rans_workspace_bytes(n_streams, n_sym, cdfs.shape[0], cdfs.shape[1]))
    ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=dev)
    with _timed('rans_encode' if indexes is None else 'rans_encode.indexed'):   # (.indexed: per-symbol CDF rows, the hyperprior's y stream)
        _check(lib().sc2_rans_encode_batch(_ptr(symbols), _ptr(indexes), int(index_div), n_streams, n_sym, _ptr(cdfs),
                                       cdfs.shape[0], cdfs.shape[1], _ptr(cdf_sizes), _ptr(offsets), _ptr(buf),
                                         out_stride, _ptr(off), _ptr(nb), _ptr(st), _ptr(ws), ws_bytes, _stream()),
               'rans_encode_batch')
    return buf, off, nb, st


def rans_decode_dequantize_batch(buf, off, nb, n_sym, cdfs, cdf_sizes, offsets, index_div, medians, want_symbols=False):
    """Decode (implicit indexes: row = position // index_div) with the dequantisation fused into the last pass:
    -> (y_hat bf16 NHWC [n_streams, index_div, C] flat pixels, status, symbols or None)."""
    for name, t in (('buf', buf), ('off', off), ('nb', nb), ('cdfs', cdfs), ('cdf_sizes', cdf_sizes), ('offsets', offsets),
                    ('medians', medians)):
        _dev(t, name)
        assert t.is_contiguous(), name
    assert buf.dtype == torch.uint8 and buf.dim() == 2 and medians.dtype == torch.float32
    n_streams, stride = buf.shape
    C = cdfs.shape[0]
    assert medians.numel() == C and n_sym == C * int(index_div)
    dev = buf.device
    y_hat = torch.empty((n_streams, int(index_div), C), dtype=torch.bfloat16, device=dev)
    sym = torch.empty((n_streams, n_sym), dtype=torch.int32, device=dev) if want_symbols else None
    st = torch.empty((n_streams,), dtype=torch.int32, device=dev)
    ws_bytes = int(lib().sc2_rans_workspace_bytes(n_streams, n_sym, cdfs.shape[0], cdfs.shape[1]))
    ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=dev)
    # the last pass of the call (dequantise + NHWC transposition) is EntropyModel.dequantize of the reference's decode: an active
    # KernelTimer that selects 'dec.dequantize' gets its own event pair around it (recorded by the library on this stream)
    t = KernelTimer.active
    ev = None
    if t is not None and (t.select is None or t.select('dec.dequantize')):
        ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        for e in ev:
            e.record()         # creates the hipEvent_t; the library records it again where it belongs
        t.records.append(('dec.dequantize', ev[0], ev[1], n_streams))
    with _timed('rans_decode'):
        _check(lib().sc2_rans_decode_dequantize_batch_ev(_ptr(buf), stride, _ptr(off), _ptr(nb), int(index_div), n_streams, int(n_sym),
                                                         _ptr(cdfs), cdfs.shape[0], cdfs.shape[1], _ptr(cdf_sizes), _ptr(offsets),
                                                         _ptr(medians), _ptr(sym) if want_symbols else None, _ptr(y_hat), _ptr(st),
                                                         _ptr(ws), ws_bytes, _stream(),
                                                         ctypes.c_void_p(ev[0].cuda_event) if ev else None,
                                                         ctypes.c_void_p(ev[1].cuda_event) if ev else None),
               'rans_decode_dequantize_batch')
    return y_hat, st, sym


def rans_decode_dequantize_supported(n_cdfs, cdf_stride):
    """True if sc2_rans_decode_dequantize_batch takes these tables (channel count % 8 == 0, <= 64, rows within the LUT decoder)."""
    return n_cdfs % 8 == 0 and n_cdfs <= 64 and cdf_stride <= 4096 and host_policy.rans_fused_dq


def rans_decode_batch(buf, off, nb, n_sym, cdfs, cdf_sizes, offsets, indexes=None, index_div=0):
    """buf: u8 [n_streams, stride] (device), stream i at buf[i, off[i]:off[i]+nb[i]].  Returns (symbols, status)."""
    for name, t in (('buf', buf), ('off', off), ('nb', nb), ('cdfs', cdfs), ('cdf_sizes', cdf_sizes),
                    ('offsets', offsets)):
        _dev(t, name)
        assert t.is_contiguous(), name
    assert buf.dtype == torch.uint8 and buf.dim() == 2
    n_streams, stride = buf.shape
    dev = buf.device
    sym = torch.empty((n_streams, n_sym), dtype=torch.int32, device=dev)
    st = torch.empty((n_streams,), dtype=torch.int32, device=dev)
    if indexes is not None:
        _dev(indexes, 'indexes')
        assert indexes.shape == sym.shape and indexes.dtype == torch.int32 and indexes.is_contiguous()
    ws_bytes = int(lib().sc2_rans_workspace_bytes(n_streams, n_sym, cdfs.shape[0], cdfs.shape[1]))
    ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=dev)
    with _timed('rans_decode' if indexes is None else 'rans_decode.indexed'):
        _check(lib().sc2_rans_decode_batch(_ptr(buf), stride, _ptr(off), _ptr(nb), _ptr(indexes), int(index_div),
                                       n_streams, int(n_sym), _ptr(cdfs), cdfs.shape[0], cdfs.shape[1],
                                       _ptr(cdf_sizes), _ptr(offsets), _ptr(sym), _ptr(st), _ptr(ws), ws_bytes,
                                           _stream()), 'rans_decode_batch')
    return sym, st
