#!/usr/bin/env python3
"""Run-length-compressed instruction trace of one kernel from `hipcc -S` output (scheduling inspection aid).
usage: isa_trace.py file.s <substring of kernel symbol> [first_line last_line]"""
import sys
path, pat = sys.argv[1], sys.argv[2]
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and pat in l and ": ; @" in l)
end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
body = lines[start:end]
lo = int(sys.argv[3]) if len(sys.argv) > 3 else 0
hi = int(sys.argv[4]) if len(sys.argv) > 4 else len(body)
prev, cnt = None, 0
def flush():
    if prev is not None:
        print(f"{cnt:3d}x {prev}")
for i, l in enumerate(body[lo:hi], lo):
    t = l.strip()
    if not t or t.startswith(";") or t.startswith("."):
        if t.startswith(".LBB"):
            flush(); prev, cnt = None, 0; print(f"[{i}] {t}")
        continue
    op = t.split()[0]
    key = op
    if op in ("s_waitcnt", "s_cbranch_scc1", "s_cbranch_scc0", "s_cbranch_vccz", "s_cbranch_vccnz", "s_cbranch_execz", "s_cbranch_execnz", "s_branch", "s_barrier"):
        key = t.split(";")[0].strip()
    elif op.startswith("v_") and not op.startswith("v_mfma"):
        key = "valu"
    elif op.startswith("s_"):
        key = "salu"
    if key == prev:
        cnt += 1
    else:
        flush(); prev, cnt = key, 1
flush()
