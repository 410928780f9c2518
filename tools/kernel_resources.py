#!/usr/bin/env python3
"""Print per-kernel resource usage of libgstfwd.so's gfx950 code objects (VGPRs, SGPRs, spills, LDS, scratch):
what decides occupancy.  Usage: tools/kernel_resources.py [regex]"""
import glob, os, re, shutil, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
objdump, readelf = "/opt/rocm/lib/llvm/bin/llvm-objdump", "/opt/rocm/lib/llvm/bin/llvm-readelf"
pat = re.compile(sys.argv[1] if len(sys.argv) > 1 else ".")
filt = shutil.which("c++filt") or "/opt/rocm/lib/llvm/bin/llvm-cxxfilt"
with tempfile.TemporaryDirectory() as td:
    so = os.path.join(td, "libgstfwd.so")
    shutil.copy(os.path.join(ROOT, "pygsti_amd", "libgstfwd.so"), so)
    subprocess.check_call([objdump, "--offloading", so], stdout=subprocess.DEVNULL)
    rows = []
    for co in sorted(glob.glob(so + ".*gfx950*")):
        notes = subprocess.check_output([readelf, "--notes", co]).decode()
        for blk in re.split(r"\n\s+- \.agpr_count", notes)[1:]:
            g = lambda k: (re.search(r"\." + k + r":\s+(\S+)", blk) or [None, "?"])[1]
            name = g("name")
            dem = subprocess.check_output([filt, name]).decode().strip()
            dem = re.sub(r"^void ", "", dem).replace("(anonymous namespace)::", ""); dem = re.sub(r"\(.*$", "", dem)
            if pat.search(dem):
                rows.append((dem, g("vgpr_count"), g("sgpr_count"), g("vgpr_spill_count"), g("sgpr_spill_count"),
                             g("group_segment_fixed_size"), g("private_segment_fixed_size"), g("max_flat_workgroup_size")))
    print("%-70s %5s %5s %6s %6s %7s %7s %5s" % ("kernel", "vgpr", "sgpr", "vspill", "sspill", "lds", "scratch", "wg"))
    for r in rows:
        print("%-70s %5s %5s %6s %6s %7s %7s %5s" % r)
