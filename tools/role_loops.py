#!/usr/bin/env python
"""Instruction mix of every barrier-paced loop of a kernel (the role loops of the workgroup-pipeline backward).
    python tools/disasm.py umnn_amd/csrc/cc_backward_bf16.o > /tmp/b.s; python tools/role_loops.py /tmp/b.s cc_bwd_ws16_kernelILi13ELb0E
Loops = backward branches whose body holds at least one s_barrier and one MFMA; per loop: VALU (non-matrix), MFMA by shape, LDS
reads / writes, SALU, s_waitcnt, global memory, s_nop."""
import re
import sys


def main():
    path, flt = sys.argv[1], sys.argv[2]
    lines = open(path).read().splitlines()
    start = next(i for i, l in enumerate(lines) if re.match(r"^[0-9a-f]+ <.*%s" % re.escape(flt), l))
    end = next((i for i in range(start + 1, len(lines)) if re.match(r"^[0-9a-f]+ <", lines[i])), len(lines))
    ins = []
    for l in lines[start + 1:end]:
        m = re.match(r"^\s+(\S+)\s*(.*?)//\s*([0-9A-F]+):", l)
        if m:
            ins.append((int(m.group(3), 16), m.group(1), m.group(2)))
    addr_idx = {a: i for i, (a, _, _) in enumerate(ins)}
    loops = []
    for i, (a, op, rest) in enumerate(ins):
        if op.startswith("s_cbranch") or op == "s_branch":
            off = int(rest.split()[0])
            if off >= 32768:
                off -= 65536
            tgt = a + 4 + 4 * off
            if tgt <= a and tgt in addr_idx:
                loops.append((addr_idx[tgt], i))
    # one loop per barrier: the smallest one that contains it
    best = {}
    for lo, hi in loops:
        for k in range(lo, hi + 1):
            if ins[k][1] == "s_barrier" and (k not in best or hi - lo < best[k][1] - best[k][0]):
                best[k] = (lo, hi)
    for lo, hi in sorted(set(best.values())):
        body = ins[lo:hi + 1]
        ops = [o for _, o, _ in body]
        if "s_barrier" not in ops or not any(o.startswith("v_mfma") for o in ops):
            continue
        c = {}
        def add(k):
            c[k] = c.get(k, 0) + 1
        for o in ops:
            if o.startswith("v_mfma"):
                add("mfma32" if "32x32" in o else ("mfma16_f32" if "x4_f32" in o or "x4f32" in o else "mfma16"))
            elif o.startswith("v_accvgpr"):
                add("accvgpr")
            elif o.startswith("v_"):
                add("valu")
                if "exp" in o or "rcp" in o or "log" in o or "sqrt" in o: add("transc")
                if o.startswith("v_mov"): add("v_mov")
                if "cvt_pk" in o: add("cvt_pk")
                if "fma_mix" in o: add("fma_mix")
                if "permlane" in o or "dpp" in o or "readlane" in o or "readfirstlane" in o: add("xlane")
                if "cndmask" in o: add("cndmask")
                if "cmp" in o: add("v_cmp")
            elif o.startswith("ds_read") or o.startswith("ds_load"):
                add("lds_rd")
            elif o.startswith("ds_write") or o.startswith("ds_store"):
                add("lds_wr")
            elif o.startswith("ds_"):
                add("lds_other")
            elif o.startswith("s_waitcnt"):
                add("waitcnt")
            elif o.startswith("s_nop"):
                add("s_nop")
            elif o.startswith("s_barrier"):
                add("barrier")
            elif o.startswith("s_"):
                add("salu")
            elif o.startswith("global_") or o.startswith("buffer_") or o.startswith("flat_") or o.startswith("scratch_"):
                add("vmem")
            else:
                add("other")
        print(f"loop @{ins[lo][0]:x}..{ins[hi][0]:x}  n={len(body)}  " + "  ".join(f"{k}={v}" for k, v in sorted(c.items())))


main()
