#!/usr/bin/env python3
"""Per-kernel resource usage of the gfx950 code objects in an object file / the library:
VGPRs, SGPRs, spills, scratch, static LDS (the AMDGPU metadata notes).

    python scripts/kernel_meta.py [pink_amd/csrc/libpinkhip.so | pink_amd/csrc/build/packed_30_32_0.o ...]
"""
import glob
import os
import re
import shutil
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kernels(path):
    out = []
    with tempfile.TemporaryDirectory() as tmp:
        local = os.path.join(tmp, os.path.basename(path))
        shutil.copy(path, local)
        subprocess.run([f"{LLVM}/llvm-objdump", "--offloading", local], check=True, capture_output=True)
        for co in sorted(glob.glob(local + ".*gfx950*")):
            notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], check=True, capture_output=True, text=True).stdout
            for block in notes.split("  - .")[1:]:
                f = dict(re.findall(r"\.?(\w+):\s+(\S+)", "." + block))
                if "vgpr_count" in f:
                    out.append(f)
    return out


def main():
    paths = sys.argv[1:] or [os.path.join(ROOT, "pink_amd", "csrc", "libpinkhip.so")]
    print(f"{'kernel':70s} {'vgpr':>5s} {'agpr':>5s} {'sgpr':>5s} {'vspill':>6s} {'sspill':>6s} {'scratch':>7s} {'lds':>6s}")
    for p in paths:
        for k in kernels(p):
            name = subprocess.run(["c++filt", k.get("name", "?")], capture_output=True, text=True).stdout.strip()
            name = re.sub(r"^void pinkhip::|\(pinkhip::\w+\)$", "", name)
            print(f"{name[:70]:70s} {k.get('vgpr_count', '?'):>5s} {k.get('agpr_count', '0'):>5s} {k.get('sgpr_count', '?'):>5s} "
                  f"{k.get('vgpr_spill_count', '0'):>6s} {k.get('sgpr_spill_count', '0'):>6s} "
                  f"{k.get('private_segment_fixed_size', '0'):>7s} {k.get('group_segment_fixed_size', '0'):>6s}")


if __name__ == "__main__":
    main()
