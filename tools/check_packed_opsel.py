"""scan the device code of libim2im_uq.so for the packed-fp32 instruction forms that are not reliable beside other processes on this
MI355X pool (profiles/r06_multiprocess_determinism.txt, tools/hwprobe/pkfma_probe.hip): v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 whose
op_sel feeds the HIGH register of the SECOND or THIRD source to the low lane -- op_sel:[x,1] / op_sel:[x,1,y] / op_sel:[x,y,1].  The
compiler picks these forms on its own (SLP-paired scalar code), so a source change or a compiler update can bring them back: the CPU
test tests/test_abi.py::test_no_unreliable_packed_fp32_forms runs this on the built library.
usage: python tools/check_packed_opsel.py [library.so]      exit status 1 and one line per kernel when a form is found"""
import collections
import os
import re
import shutil
import subprocess
import sys
import tempfile

LLVM_BIN = "/opt/rocm/lib/llvm/bin"
PK = re.compile(r"\b(v_pk_(?:fma|mul|add)_f32)\b.*?\bop_sel:\[([01])(?:,([01]))?(?:,([01]))?\]")
LABEL = re.compile(r"^[0-9a-f]+ <([^>]+)>:")


def objdump() -> str:
    cand = os.path.join(LLVM_BIN, "llvm-objdump")
    return cand if os.path.exists(cand) else (shutil.which("llvm-objdump") or "")


def scan(lib: str):
    """-> (number of code objects, {kernel symbol: Counter of 'instruction op_sel:[...]'}) ; raises RuntimeError without llvm-objdump"""
    tool = objdump()
    if not tool:
        raise RuntimeError("llvm-objdump not found")
    found = collections.defaultdict(collections.Counter)
    with tempfile.TemporaryDirectory() as tmp:
        copy = os.path.join(tmp, "lib.so")
        shutil.copy(lib, copy)
        subprocess.run([tool, "--offloading", copy], check=True, capture_output=True)       # writes lib.so.N.<target> next to the copy
        objs = sorted(f for f in os.listdir(tmp) if "amdgcn" in f)
        for f in objs:
            p = subprocess.Popen([tool, "-d", "--no-show-raw-insn", os.path.join(tmp, f)], stdout=subprocess.PIPE, text=True)
            sym = "?"
            for line in p.stdout:
                if "v_pk_" not in line:
                    m = LABEL.match(line)
                    if m:
                        sym = m.group(1)
                    continue
                m = PK.search(line)
                if m and ("1" in (m.group(3) or "", m.group(4) or "")):
                    sel = ",".join(g for g in m.groups()[1:] if g is not None)
                    found[sym][f"{m.group(1)} op_sel:[{sel}]"] += 1
            p.wait()
    return len(objs), found


def main():
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "im2im_uq_amd", "lib", "libim2im_uq.so")
    n, found = scan(lib)
    for sym, c in sorted(found.items()):
        print(f"{sym[:150]}: " + ", ".join(f"{k} x{v}" for k, v in c.items()))
    print(f"[check_packed_opsel] {n} code objects, {len(found)} kernels with a second/third-source op_sel on a packed fp32 instruction")
    sys.exit(1 if found else 0)


if __name__ == "__main__":
    main()
