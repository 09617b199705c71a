"""Scan the device code of a built libdm_hip.so for packed fp32 instructions whose LOW lane takes the HIGH register of
the second source (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 with op_sel:[x,1,...]).

That form was measured on MI355X to lose the low lane's product in lanes 48-63 while waves of another kernel that
issue matrix-core instructions and direct-to-LDS buffer loads shared the compute unit (the vendor's 16-bit GEMMs, this
library's 3x3 convolutions). tools/experiments/pk_forms_repro.hip is a stand-alone reproducer that walks through the
operand-select forms: op_sel:[0,1,0] (fma) and op_sel:[0,1] (mul, add) fail, every other form - op_sel:[1,0,0],
[0,0,1], [1,1,0], all op_sel_hi forms, v_pk_mov_b32 - does not; this check flags any op_sel with the second bit set.
The compiler chooses the form for the odd elements of a 16-byte vector that multiply a pair of rows (an fp32 inner loop
fed by 16-byte LDS reads); kernels that must be exact write such loops with explicit v_fmac_f32 (fma_acc in
csrc/small_ops.hip). This check keeps the form from coming back unnoticed after a compiler or source change.

  python tools/check_packed_forms.py [path/to/libdm_hip.so]      exit status 1 and a list of kernels if any is found
"""
import os
import re
import subprocess
import sys
import tempfile

LLVM = os.environ.get("ROCM_LLVM_BIN", "/opt/rocm/lib/llvm/bin")
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
TARGET = "hipv4-amdgcn-amd-amdhsa--gfx950"
FORM = re.compile(r"\b(v_pk_(?:fma|mul|add)_f32)\b.*\bop_sel:\[[01],1[,\]]")


def device_disassembly(lib_path):
    """Yield the disassembly text of every gfx950 code object embedded in the library."""
    with tempfile.TemporaryDirectory() as tmp:
        fat = os.path.join(tmp, "fat.bin")
        subprocess.check_call([os.path.join(LLVM, "llvm-objcopy"), f"--dump-section=.hip_fatbin={fat}", lib_path,
                               os.path.join(tmp, "copy.so")])
        blob = open(fat, "rb").read()
        starts = [m.start() for m in re.finditer(re.escape(MAGIC), blob)]
        for i, begin in enumerate(starts):
            piece = os.path.join(tmp, f"bundle{i}.bin")
            with open(piece, "wb") as f:
                f.write(blob[begin:starts[i + 1] if i + 1 < len(starts) else len(blob)])
            code = os.path.join(tmp, f"code{i}.co")
            subprocess.check_call([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o",
                                   f"--input={piece}", f"--targets={TARGET}", f"--output={code}"],
                                  stderr=subprocess.DEVNULL)
            yield subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", code], check=True, capture_output=True,
                                 text=True).stdout


def find_forms(lib_path):
    """[(kernel symbol, instruction text)] for every occurrence of the form; also the number of code objects read."""
    found, objects = [], 0
    for text in device_disassembly(lib_path):
        objects += 1
        symbol = "?"
        for line in text.splitlines():
            m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
            if m:
                symbol = m.group(1)
            elif FORM.search(line):
                found.append((symbol, " ".join(line.split("//")[0].split())))
    return found, objects


if __name__ == "__main__":
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(here, "diffusion-models_amd", "dm_hip", "libdm_hip.so")
    hits, n = find_forms(lib)
    print(f"{lib}: {n} code objects, {len(hits)} packed fp32 instructions with the low lane on the second source's high register")
    for sym, ins in hits[:40]:
        print(f"  {sym}: {ins}")
    sys.exit(1 if hits else 0)
