#!/usr/bin/env python
"""Disassemble the gfx950 code object(s) inside a hipcc object / shared library:  python tools/disasm.py file.o > out.s"""
import subprocess
import sys
import tempfile

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from tools.kernel_resources import code_objects  # noqa: E402

for co in code_objects(open(sys.argv[1], "rb").read()):
    with tempfile.NamedTemporaryFile(suffix=".co") as f:
        f.write(co)
        f.flush()
        sys.stdout.write(subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-objdump", "-d", f.name], capture_output=True, text=True).stdout)
