#!/usr/bin/env python3
"""Register / scratch / LDS use of every kernel in a built library, read from the code object's metadata notes (no GPU needed).
usage: kernel_resources.py [lib.so] [name filter]
Columns: private segment (scratch) bytes, VGPRs, AGPRs, SGPRs, spilled VGPRs, LDS bytes, demangled-ish name."""
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"


def kernel_resources(so):
    """-> list of dicts(name, priv, vgpr, agpr, sgpr, spill, lds)"""
    with tempfile.TemporaryDirectory() as td:
        fat, co = os.path.join(td, "fat.bin"), os.path.join(td, "k.co")
        subprocess.check_call([os.path.join(LLVM, "llvm-objcopy"), "--dump-section", ".hip_fatbin=" + fat, so, os.path.join(td, "x.so")])
        subprocess.check_call([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", "--input=" + fat,
                               "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co])
        notes = subprocess.check_output([os.path.join(LLVM, "llvm-readelf"), "--notes", co]).decode()
    out = []
    for blk in notes.split("- .agpr_count:")[1:]:
        def f(key):
            m = re.search(r"\.%s:\s+(\S+)" % key, blk)
            return m.group(1) if m else "0"
        out.append(dict(name=f("name"), priv=int(f("private_segment_fixed_size")), vgpr=int(f("vgpr_count")), agpr=int(blk.split()[0]),
                        sgpr=int(f("sgpr_count")), spill=int(f("vgpr_spill_count")), lds=int(f("group_segment_fixed_size"))))
    return out


if __name__ == "__main__":
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    so = sys.argv[1] if len(sys.argv) > 1 else os.path.join(root, "bipymc_amd", "libbipymc_hip.so")
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    print("%6s %5s %5s %5s %6s %6s  name" % ("priv", "vgpr", "agpr", "sgpr", "spill", "lds"))
    for k in sorted(kernel_resources(so), key=lambda k: k["name"]):
        if flt in k["name"]:
            print("%6d %5d %5d %5d %6d %6d  %s" % (k["priv"], k["vgpr"], k["agpr"], k["sgpr"], k["spill"], k["lds"], k["name"]))
