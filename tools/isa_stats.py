#!/usr/bin/env python3
"""Per-kernel ISA statistics from a hipcc -save-temps .s file: loads, waits, branches (dev tool)."""
import re
import sys

s = open(sys.argv[1]).read()
pat = sys.argv[2] if len(sys.argv) > 2 else ""
R_LD = re.compile(r"global_load|buffer_load")
R_W = re.compile(r"s_waitcnt[^\n]*vmcnt")
R_W0 = re.compile(r"vmcnt\(0\)")
R_BR = re.compile(r"s_cbranch")
R_MF = re.compile(r"v_mfma")
for f in re.split(r"\n\s*\.globl\s+", s)[1:]:
    name = f.split("\n", 1)[0].strip()
    if pat and not re.search(pat, name):
        continue
    body = f[: f.find(".end_amdhsa_kernel")] if ".end_amdhsa_kernel" in f else f
    print("%-70s lines %6d loads %4d vmcnt-waits %4d vmcnt(0) %3d branches %3d mfma %4d" % (
        name[:70], body.count("\n"), len(R_LD.findall(body)), len(R_W.findall(body)), len(R_W0.findall(body)),
        len(R_BR.findall(body)), len(R_MF.findall(body))))
