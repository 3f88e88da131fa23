#!/usr/bin/env python
"""Registers / spills / LDS of every kernel in a hipcc object or shared library (reads the gfx950 code object's notes).

    python tools/kernel_resources.py umnn_amd/csrc/cc_forward_bf16.o [name filter]
"""
import re
import struct
import subprocess
import sys
import tempfile

READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def code_objects(blob):
    out, pos = [], 0
    while True:
        pos = blob.find(MAGIC, pos)
        if pos < 0:
            return out
        n = struct.unpack_from("<Q", blob, pos + 24)[0]
        p = pos + 32
        for _ in range(n):
            off, size, tlen = struct.unpack_from("<QQQ", blob, p)
            triple = blob[p + 24:p + 24 + tlen].decode()
            p += 24 + tlen
            if "gfx" in triple and size:
                out.append(blob[pos + off:pos + off + size])
        pos += 24


def main():
    blob = open(sys.argv[1], "rb").read()
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    for co in code_objects(blob):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(co)
            f.flush()
            txt = subprocess.run([READELF, "--notes", f.name], capture_output=True, text=True).stdout
        for blk in re.split(r"\n\s+- \.agpr_count:", txt)[1:]:
            blk = ".agpr_count:" + blk
            g = lambda k: (re.search(r"\.%s:\s+(\S+)" % k, blk) or [None, "?"])[1]      # noqa: E731
            name = subprocess.run(["c++filt", g("name")], capture_output=True, text=True).stdout.strip()
            if flt in name:
                print(f"vgpr {g('vgpr_count'):>4} agpr {g('agpr_count'):>4} sgpr {g('sgpr_count'):>4} spill {g('vgpr_spill_count'):>3} "
                      f"scratch {g('private_segment_fixed_size'):>5} lds {g('group_segment_fixed_size'):>6}  {name[:110]}")


if __name__ == "__main__":
    main()
