#!/usr/bin/env python3
"""Register / scratch / LDS figures of every kernel in a built libpphip.so, read from the gfx950 code objects inside its
.hip_fatbin section (clang offload bundles -> AMDGPU ELF -> NT_AMDGPU_METADATA msgpack note).  No GPU needed.

    python tools/kernel_resources.py [path/to/libpphip.so] [--json]

Used by tests/test_kernel_resources.py as a regression gate on scratch use (the round-1 abort: a planner took the last byte
of HBM and the runtime could not allocate the scratch its kernels need at first dispatch)."""
import json
import struct
import sys

import msgpack

MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def _section(data, name):
    """(offset, size) of an ELF64 section by name."""
    shoff, = struct.unpack_from("<Q", data, 0x28)
    shentsize, shnum, shstrndx = struct.unpack_from("<HHH", data, 0x3A)
    def hdr(i):
        return struct.unpack_from("<IIQQQQIIQQ", data, shoff + i * shentsize)
    stroff = hdr(shstrndx)[4]
    out = []
    for i in range(shnum):
        h = hdr(i)
        nm = data[stroff + h[0]:data.index(b"\0", stroff + h[0])].decode()
        if nm == name:
            out.append((h[4], h[5], h[1]))
    return out


def code_objects(so_path, arch="gfx950"):
    data = open(so_path, "rb").read()
    secs = _section(data, ".hip_fatbin")
    if not secs:
        raise RuntimeError("no .hip_fatbin section in %s" % so_path)
    off, size, _ = secs[0]
    fat = data[off:off + size]
    pos = 0
    while True:
        b = fat.find(MAGIC, pos)
        if b < 0:
            break
        n, = struct.unpack_from("<Q", fat, b + len(MAGIC))
        p = b + len(MAGIC) + 8
        for _ in range(n):
            eoff, esize, tlen = struct.unpack_from("<QQQ", fat, p)
            triple = fat[p + 24:p + 24 + tlen].decode()
            p += 24 + tlen
            if arch in triple and esize:
                yield fat[b + eoff:b + eoff + esize]
        pos = b + len(MAGIC)


def kernels_of(elf):
    out = []
    for off, size, typ in _section(elf, ".note"):
        p = off
        while p < off + size:
            namesz, descsz, ntype = struct.unpack_from("<III", elf, p)
            p += 12
            name = elf[p:p + namesz].rstrip(b"\0")
            p += (namesz + 3) & ~3
            desc = elf[p:p + descsz]
            p += (descsz + 3) & ~3
            if name == b"AMDGPU" and ntype == 32:  # NT_AMDGPU_METADATA
                md = msgpack.unpackb(desc, raw=False, strict_map_key=False)
                for k in md.get("amdhsa.kernels", []):
                    out.append(dict(name=k[".name"], vgpr=k.get(".vgpr_count"), agpr=k.get(".agpr_count", 0), sgpr=k.get(".sgpr_count"),
                                    vgpr_spill=k.get(".vgpr_spill_count", 0), sgpr_spill=k.get(".sgpr_spill_count", 0),
                                    scratch_bytes_per_lane=k.get(".private_segment_fixed_size", 0), lds_bytes=k.get(".group_segment_fixed_size", 0),
                                    max_flat_workgroup_size=k.get(".max_flat_workgroup_size")))
    return out


def demangle_short(name):
    import re
    m = re.search(r"\d+(k_[a-z0-9_]+)", name)
    s = m.group(1) if m else name
    if "ILb1E" in name:
        s += "<true>"
    elif "ILb0E" in name:
        s += "<false>"
    return s


def resources(so_path):
    res = []
    for elf in code_objects(so_path):
        for k in kernels_of(elf):
            k["kernel"] = demangle_short(k["name"])
            res.append(k)
    return res


if __name__ == "__main__":
    import os
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    so = args[0] if args else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "pathplanning_amd", "lib", "libpphip.so")
    res = resources(so)
    if "--json" in sys.argv:
        print(json.dumps(res, indent=1))
    else:
        print("%-34s %5s %5s %10s %10s %12s %8s" % ("kernel", "vgpr", "sgpr", "vgpr_spill", "sgpr_spill", "scratch B/ln", "LDS B"))
        for k in sorted(res, key=lambda k: k["kernel"]):
            print("%-34s %5s %5s %10s %10s %12s %8s" % (k["kernel"], k["vgpr"], k["sgpr"], k["vgpr_spill"], k["sgpr_spill"], k["scratch_bytes_per_lane"], k["lds_bytes"]))
