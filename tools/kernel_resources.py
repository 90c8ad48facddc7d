#!/usr/bin/env python3
"""Per-kernel register / scratch / LDS figures of a gfx950 object file or bundle (reads the AMDGPU metadata note with
llvm-readelf): tools/kernel_resources.py iron_amd/csrc/build/trace.o [other.o]  -- two files are printed side by side."""
import re
import subprocess
import sys

READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"
BUNDLER = "/opt/rocm/lib/llvm/bin/clang-offload-bundler"


def device_elf(path):
    """The gfx950 code object inside a host object's .hip_fatbin section (an offload bundle)."""
    import os
    import tempfile
    fat = tempfile.mktemp(suffix=".fatbin")
    r = subprocess.run(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", path, fat], capture_output=True, text=True)
    if r.returncode != 0 or not os.path.exists(fat) or os.path.getsize(fat) == 0:
        return path
    out = tempfile.mktemp(suffix=".co")
    r = subprocess.run([BUNDLER, "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--input=" + fat, "--output=" + out],
                       capture_output=True, text=True)
    if r.returncode != 0 or not os.path.exists(out) or os.path.getsize(out) == 0:
        sys.stderr.write(r.stderr)
        return path
    return out


def resources(path):
    txt = subprocess.run([READELF, "--notes", device_elf(path)], capture_output=True, text=True).stdout
    res = {}
    for blk in txt.split("- .agpr_count:")[1:]:
        blk = ".agpr_count:" + blk
        name = re.search(r"\.name:\s+(\S+)", blk)
        if not name:
            continue
        g = lambda k: int(re.search(r"\.%s:\s+(\d+)" % k, blk).group(1)) if re.search(r"\.%s:\s+(\d+)" % k, blk) else -1
        res[name.group(1)] = (g("vgpr_count"), g("agpr_count"), g("sgpr_count"), g("private_segment_fixed_size"), g("vgpr_spill_count"), g("sgpr_spill_count"))
    return res


def demangle(n):
    r = subprocess.run(["c++filt", n], capture_output=True, text=True).stdout.strip()
    return re.sub(r"\(.*", "", r)[:70]


if __name__ == "__main__":
    rs = [resources(p) for p in sys.argv[1:]]
    names = sorted(set().union(*[set(r) for r in rs]))
    print("%-72s %s" % ("kernel", "   |   ".join("vgpr agpr sgpr scratch vspill sspill" for _ in rs)))
    for n in names:
        print("%-72s %s" % (demangle(n), "   |   ".join(("%4d %4d %4d %7d %6d %6d" % r[n]) if n in r else "-" for r in rs)))
