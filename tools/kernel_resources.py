"""Per-kernel resource usage read from the BUILT libsc2amd.so (no recompilation): walks the clang offload bundles of the
.hip_fatbin section, opens every gfx950 code object, and decodes the AMDGPU metadata note (msgpack) -> for each kernel
its VGPR / SGPR counts, LDS and `.private_segment_fixed_size` (scratch bytes per lane).

    python tools/kernel_resources.py [path/to/lib.so] [--scratch-only]

Why: a conv_igemm_impl.h edit once sent the accumulators of the 256-wide decoder kernels to scratch (528 B / lane,
dec.conv2 0.92 -> 1.12 ms) without any test failing.  tests/test_abi.py asserts that the hot-path kernels use none.
"""
import struct
import sys

MAGIC = b'__CLANG_OFFLOAD_BUNDLE__'


def code_objects(blob):
    pos = 0
    while True:
        i = blob.find(MAGIC, pos)
        if i < 0:
            return
        n = struct.unpack_from('<Q', blob, i + len(MAGIC))[0]
        p = i + len(MAGIC) + 8
        for _ in range(n):
            off, size, tlen = struct.unpack_from('<QQQ', blob, p)
            triple = blob[p + 24:p + 24 + tlen].decode()
            p += 24 + tlen
            if 'gfx950' in triple and size:
                yield blob[i + off:i + off + size]
        pos = i + len(MAGIC)


def notes(elf):
    """AMDGPU metadata dict of one code object (ELF64 little endian)."""
    import msgpack
    shoff, = struct.unpack_from('<Q', elf, 0x28)
    shentsize, shnum = struct.unpack_from('<HH', elf, 0x3A)
    for k in range(shnum):
        sh = shoff + k * shentsize
        sh_type, = struct.unpack_from('<I', elf, sh + 4)
        if sh_type != 7:     # SHT_NOTE
            continue
        off, size = struct.unpack_from('<QQ', elf, sh + 0x18)
        p, end = off, off + size
        while p + 12 <= end:
            namesz, descsz, ntype = struct.unpack_from('<III', elf, p)
            p += 12
            name = elf[p:p + namesz].rstrip(b'\0')
            p += (namesz + 3) & ~3
            desc = elf[p:p + descsz]
            p += (descsz + 3) & ~3
            if name == b'AMDGPU' and ntype == 32:
                return msgpack.unpackb(desc, raw=False, strict_map_key=False)
    return None


def kernels(path):
    blob = open(path, 'rb').read()
    out = []
    for co in code_objects(blob):
        md = notes(co)
        for k in (md or {}).get('amdhsa.kernels', []):
            out.append({'name': k['.name'], 'vgpr': k.get('.vgpr_count'), 'sgpr': k.get('.sgpr_count'),
                        'agpr': k.get('.agpr_count', 0), 'lds': k.get('.group_segment_fixed_size'),
                        'scratch': k.get('.private_segment_fixed_size'), 'spill_vgpr': k.get('.vgpr_spill_count', 0)})
    return out


def demangle_hint(name):
    for key in ('conv_igemm8_kernel', 'conv_igemm4_kernel', 'conv_igemm_kernel', 'conv5s2_patch_kernel'):
        if key in name:
            return key + name.split(key)[1][:70]
    return name[:90]


if __name__ == '__main__':
    import os
    args = [a for a in sys.argv[1:] if not a.startswith('--')]
    path = args[0] if args else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                             'sc2-benchmark_amd', 'libsc2amd.so')
    for k in sorted(kernels(path), key=lambda k: (-k['scratch'], k['name'])):
        if '--scratch-only' in sys.argv and not k['scratch']:
            continue
        print('{:>5} B scratch  {:>3} vgpr {:>3} sgpr {:>6} lds  {}'.format(k['scratch'], k['vgpr'], k['sgpr'], k['lds'],
                                                                           demangle_hint(k['name'])))
