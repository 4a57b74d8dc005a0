#!/usr/bin/env python
"""Scan the gfx950 code objects of the built library for the one hazard the inline-assembly DPP instructions
(csrc/sc_row16.h: v_fmac_f64_dpp / v_mov_b64_dpp row_newbcast) can hit without the compiler noticing:

    a VGPR written by a VALU instruction must not be read as the DPP (src0) operand by one of the next two
    instructions ("VALU writes VGPR -> DPP reads that VGPR: 2 wait states"; an s_nop N counts N + 1 states).

(The in-place elimination updates of sc_row16.h -- v_fmac_f64_dpp X, X(dpp), -m followed by further DPP reads of X -- are
issued in groups ordered so that the rule holds for them as for everything else: cfnma_inplace.)

    python tools/check_dpp_hazards.py [library.so]      exit status 1 and a listing if a violation is found
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
BUNDLER = "/opt/rocm/lib/llvm/bin/clang-offload-bundler"


def _regs(tok):
    out = []
    for m in re.finditer(r'\bv\[(\d+):(\d+)\]|\bv(\d+)\b', tok):
        if m.group(1) is not None:
            out += list(range(int(m.group(1)), int(m.group(2)) + 1))
        else:
            out.append(int(m.group(3)))
    return out


def disassemble(lib):
    """text of every gfx950 code object bundled in `lib`"""
    with tempfile.TemporaryDirectory() as tmp:
        co = os.path.join(tmp, "dev.co")
        # the fat binary sits in section .hip_fatbin of the host library
        fat = os.path.join(tmp, "fat.bin")
        subprocess.check_call(["/opt/rocm/lib/llvm/bin/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", lib, fat])
        # one offload bundle per translation unit, concatenated: cut at the bundle magic
        blob = open(fat, "rb").read()
        magic = b"__CLANG_OFFLOAD_BUNDLE__"
        starts = [m.start() for m in re.finditer(re.escape(magic), blob)] + [len(blob)]
        text = ""
        for i in range(len(starts) - 1):
            piece = os.path.join(tmp, f"bundle{i}.bin")
            with open(piece, "wb") as fh:
                fh.write(blob[starts[i]:starts[i + 1]])
            listing = subprocess.check_output([BUNDLER, "--list", "--type=o", f"--input={piece}"], text=True).split()
            for target in (t for t in listing if "gfx950" in t):
                subprocess.check_call([BUNDLER, "--unbundle", "--type=o", f"--input={piece}", f"--targets={target}",
                                       f"--output={co}"])
                text += subprocess.check_output([OBJDUMP, "-d", "--no-show-raw-insn", co], text=True)
        return text


def scan(text):
    """[(function, address line, offending writer line)]"""
    bad, func = [], "?"
    window = []          # (states ago it issued, set of VGPRs it wrote as a VALU, text, in-place DPP update?)
    n_dpp = n_inplace = 0
    for line in text.split("\n"):
        m = re.match(r'^[0-9a-f]+ <(.+)>:$', line)
        if m:
            func, window = m.group(1), []
            continue
        s = line.strip()
        if not s or s.startswith(("/", ".")) or ":" in s.split()[0]:
            continue
        s = s.split("//")[0].strip()
        op = s.split()[0]
        ops = [t.strip() for t in s[len(op):].split(",")]
        if op == "s_nop":
            k = int(ops[0], 0) + 1
            window = [(age + k, w, t, ip) for age, w, t, ip in window if age + k <= 2]
            continue
        if "_dpp" in op or "row_newbcast" in s or "row_ror" in s or "row_shr" in s or "quad_perm" in s:
            n_dpp += 1
            src0 = _regs(ops[1]) if len(ops) > 1 else []
            for age, wrote, t, inplace in window:
                if age <= 2 and wrote.intersection(src0):
                    bad.append((func, s, t))
        wrote, inplace = set(), False
        if op.startswith("v_") and not op.startswith(("v_cmp", "v_readlane", "v_readfirstlane")):
            wrote = set(_regs(ops[0]))
            inplace = op == "v_fmac_f64_dpp" and len(ops) > 1 and set(_regs(ops[1])) == wrote
        window = [(age + 1, w, t, ip) for age, w, t, ip in window if age + 1 <= 2] + [(1, wrote, s, inplace)]
    return bad, n_dpp, n_inplace


def main(argv):
    lib = argv[1] if len(argv) > 1 else os.path.join(ROOT, "semiclassical_amd", "libsemiclassical_hip.so")
    bad, n_dpp, n_inplace = scan(disassemble(lib))
    print(f"{lib}: {n_dpp} DPP instructions scanned, {len(bad)} hazard violations")
    for func, reader, writer in bad[:40]:
        print(f"  {func}\n      {writer}\n   -> {reader}")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv))
