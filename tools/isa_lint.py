"""ISA lint of the BUILT library: no packed-fp32 arithmetic whose op_sel routes the high dword of src1 into the low result.

Why (profiles/r6_corunner_defect.txt, profiles/r6_pk_opsel_probe.txt, tools/proto/pk_opsel_probe.hip): on gfx950 a
``v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 ... op_sel:[x,1(,x)]`` reads that operand as 0.0 in lanes 48-63 now and then while a wave
of ANOTHER kernel on the same SIMD executes one of the 128-bit-operand matrix instructions (v_mfma_f32_32x32x16_f16 / _bf16,
v_mfma_f32_16x16x32_f16) -- i.e. next to this package's own split-f16 GEMMs.  Alone, or next to the 64-bit-operand MFMAs, it never
does; op_sel on src0 / src2, op_sel_hi and v_pk_mov_b32 were clean over 2e10 lane-executions each.  hipcc emits the form for any
``vec * other_vec[odd index]``; kernels that did are compiled with COOCC_SCALAR_FP32 (common.h: target("no-packed-fp32-ops")).

The check disassembles what is actually shipped: every code object of ``co_occ_amd/libcoocc_hip.so``'s .hip_fatbin section.

    python tools/isa_lint.py [path/to/lib.so]        exit status 1 if a kernel holds the form; prints the offenders
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = os.environ.get("COOCC_LLVM_BIN", "/opt/rocm/lib/llvm/bin")
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
TARGET = "hipv4-amdgcn-amd-amdhsa--gfx950"
BAD = re.compile(r"\b(v_pk_(?:fma|mul|add|min|max)\w*_f32)\b.*\bop_sel:\[([01]),([01])")
LABEL = re.compile(r"^[0-9a-f]+ <([^>]+)>:")


def code_objects(lib, tmp):
    """The gfx950 code objects of every translation unit linked into ``lib`` (one clang offload bundle each)."""
    fat = os.path.join(tmp, "fat.bin")
    subprocess.run([os.path.join(LLVM, "llvm-objcopy"), "--dump-section", ".hip_fatbin=" + fat, lib, os.path.join(tmp, "copy.so")], check=True)
    blob = open(fat, "rb").read()
    starts = [m.start() for m in re.finditer(re.escape(MAGIC), blob)]
    outs = []
    for i, s in enumerate(starts):
        e = starts[i + 1] if i + 1 < len(starts) else len(blob)
        b = os.path.join(tmp, "bundle%d.bin" % i)
        open(b, "wb").write(blob[s:e])
        o = os.path.join(tmp, "co%d.o" % i)
        r = subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", "--targets=" + TARGET, "--input=" + b, "--output=" + o],
                           capture_output=True, text=True)
        if r.returncode == 0 and os.path.exists(o) and os.path.getsize(o) > 0:
            outs.append(o)
    return outs


def scan(lib):
    """-> (kernels seen, [(kernel, instruction text)]) for the forbidden form."""
    bad, kernels = [], set()
    with tempfile.TemporaryDirectory() as tmp:
        for o in code_objects(lib, tmp):
            dis = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--mcpu=gfx950", o], capture_output=True, text=True, check=True).stdout
            cur = None
            for line in dis.splitlines():
                m = LABEL.match(line)
                if m:
                    cur = m.group(1)
                    kernels.add(cur)
                    continue
                m = BAD.search(line)
                if m and m.group(3) == "1":
                    bad.append((cur, line.split("//")[0].strip()))
    return kernels, bad


def main():
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "co_occ_amd", "libcoocc_hip.so")
    kernels, bad = scan(lib)
    by = {}
    for k, ins in bad:
        by.setdefault(k, []).append(ins)
    for k in sorted(by):
        print("%-90s %3d  e.g. %s" % (k[:90], len(by[k]), by[k][0]))
    print("%d functions disassembled, %d with packed-fp32 op_sel[src1] = 1 (%d instructions)" % (len(kernels), len(by), len(bad)))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
