#!/usr/bin/env python3
"""Register / spill table of the kernels of one object file: tools/dev/kregs.py lib/obj/x.o [name filter]"""
import re, subprocess, sys, tempfile, os
LL = "/opt/rocm/lib/llvm/bin/"
obj, flt = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
d = tempfile.mkdtemp()
subprocess.check_call([LL + "llvm-objcopy", "--dump-section", ".hip_fatbin=%s/fat.bin" % d, obj])
subprocess.check_call([LL + "clang-offload-bundler", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--input=%s/fat.bin" % d,
                       "--output=%s/k.co" % d, "--unbundle"])
t = subprocess.check_output([LL + "llvm-readelf", "--notes", d + "/k.co"], text=True)
dem = lambda n: subprocess.check_output(["c++filt", n], text=True).strip()
for b in t.split("  - .agpr_count:")[1:]:
    nm = dem(re.search(r"\.name:\s+(\S+)", b).group(1))
    if flt not in nm:
        continue
    g = lambda k: re.search(r"\.%s:\s+(\S+)" % k, b).group(1)
    print("%-80s agpr %3s vgpr %3s sgpr %3s scratch %4s spill %3s" % (nm[:80], b.split()[0], g("vgpr_count"), g("sgpr_count"),
                                                                      g("private_segment_fixed_size"), g("vgpr_spill_count")))
