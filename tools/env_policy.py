"""The ONLY place that maps SC2_* environment variables onto the dispatch policy (sc2_policy of the C-ABI + hip.host_policy).
The package reads none of them: A/B shell scripts of this directory set the variables as before and the Python
entry points they drive call `apply()` explicitly (`bench.py --policy-env`, tools/k_times.py, tools/layer_times.py, ...).

    from tools import env_policy; env_policy.apply()            # -> {field: value} of what was changed
    python bench.py --policy rans_lut8=0,conv0_fused=0          # the same without the environment
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

# variable -> (policy field, converter)
_INT = int
_ON = lambda v: v != '0'                       # noqa: E731  ("1" default switches: anything but '0' is on)
ENV = {
    # library (sc2_policy)
    'SC2_CONV_PATCH3': ('conv_patch3', _INT), 'SC2_CONV_S2': ('conv_s2', _INT), 'SC2_CONV_PERSIST': ('conv_persist', _INT),
    'SC2_CONV_HALF': ('conv_half', _INT), 'SC2_CONV_BIG4': ('conv_big4', _INT), 'SC2_CONV_NO_BIG': ('conv_no_big', lambda v: 1),
    'SC2_CONV_FORCE_BIG': ('conv_force_big', lambda v: 1), 'SC2_CONV_NO_EPX': ('conv_no_epx', lambda v: 1),
    'SC2_CONV_DEBUG': ('conv_debug', _INT), 'SC2_CONV_CHUNK': ('conv_chunk', _INT),
    'SC2_W2_RUN': ('w2_run', _INT), 'SC2_WIN_HALF': ('win_half', _INT), 'SC2_WIN_DBG': ('win_dbg', _INT),
    'SC2_WIN_STAMPS': ('win_stamps', _INT), 'SC2_P1_HALF': ('p1_half', _INT), 'SC2_P1_NBUF': ('p1_nbuf', _INT),
    'SC2_PAIR_ALT': ('pair_alt', _INT), 'SC2_F32_PERSIST0': ('f32_persist0', lambda v: 0 if v.startswith('0') else 1),
    'SC2_DEC_STAGGER': ('dec_stagger', _INT), 'SC2_WGRAD_WGS': ('wgrad_wgs', _INT), 'SC2_RANS_LDS_PAD': ('rans_lds_pad_kb', _INT),
    'SC2_RANS_PAD_WAVES': ('rans_pad_waves', _INT), 'SC2_RANS_RAGGED2': ('rans_ragged2', _INT),
    'SC2_RANS_RAGGED2_WAVES': ('rans_ragged2_waves', _INT), 'SC2_RANS_LUT8': ('rans_lut8', _INT),
    # host side (hip.host_policy)
    'SC2_CONV_PATCH': ('conv_patch', _ON), 'SC2_K_ORDER': ('k_order_tap', lambda v: v == 'tap'), 'SC2_B_TILE': ('b_tile_major', _ON),
    'SC2_CONV0_FUSED': ('conv0_fused', _ON), 'SC2_CONV2_FUSED': ('conv2_fused', _ON), 'SC2_CONV_KRES': ('conv_kres', _INT),
    'SC2_CONV_WIN': ('conv_win', _ON), 'SC2_CONV_WIN_S2': ('conv_win_s2', _ON), 'SC2_CONV2X2_WIN': ('conv2x2_win', _ON),
    'SC2_W2_TAIL': ('w2_tail', _ON), 'SC2_CONV_STREAM': ('conv_stream', _ON), 'SC2_CONV1X1_PAIR': ('conv1x1_pair', _ON),
    'SC2_CONV_C48': ('conv_c48', _ON), 'SC2_CONV1X1_WIN': ('conv1x1_win', str), 'SC2_CONV_DILATION': ('conv_dilation', _ON),
    'SC2_FC_KERNEL': ('fc_kernel', _ON), 'SC2_DENSE_HEAD': ('dense_head', _ON), 'SC2_RANS_FUSED_DQ': ('rans_fused_dq', _ON),
    'SC2_HOST_CODER_MAX_STREAMS': ('host_coder_max_streams', _INT),
}


def apply(environ=None):
    """Applies every SC2_* dispatch variable found in the environment; -> {field: value}."""
    import sc2bench_amd as S
    environ = os.environ if environ is None else environ
    changed = {}
    for name, (field, conv) in ENV.items():
        if name in environ:
            changed[field] = conv(environ[name])
    if changed:
        S.hip.configure(**changed)
    return changed


def parse(spec):
    """'a=1,b=0,conv1x1_win=all' -> {field: value} (ints where they parse, else strings; true / false accepted)."""
    out = {}
    for item in filter(None, (q.strip() for q in (spec or '').split(','))):
        k, _, v = item.partition('=')
        lv = v.lower()
        out[k.strip()] = True if lv == 'true' else False if lv == 'false' else int(v) if v.lstrip('-').isdigit() else v
    return out


if __name__ == '__main__':
    print(apply())
