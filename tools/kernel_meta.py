"""Kernel metadata (VGPRs, spills, scratch, LDS) of a built translation unit without recompiling: unbundles the gfx950 code object from
typesense_amd/_obj/<tag>/<tu>.o and reads its notes. Usage: python tools/kernel_meta.py [obj] [name substring ...]"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
obj = sys.argv[1] if len(sys.argv) > 1 and sys.argv[1].endswith(".o") else os.path.join(ROOT, "typesense_amd", "_obj", "default", "tsgpu.hip.o")
pats = [a for a in sys.argv[1:] if not a.endswith(".o")] or ["kw_find2_kernel", "kw_score_kernel"]
LLVM = "/opt/rocm/lib/llvm/bin/"
with tempfile.TemporaryDirectory() as d:
    fat, co = os.path.join(d, "fat.bin"), os.path.join(d, "dev.co")
    subprocess.check_call([LLVM + "llvm-objcopy", "--dump-section", ".hip_fatbin=" + fat, obj])
    subprocess.check_call([LLVM + "clang-offload-bundler", "--type=o", "--unbundle", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--input=" + fat, "--output=" + co])
    notes = subprocess.run([LLVM + "llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
for blk in re.split(r"\n\s+- \.agpr_count", notes)[1:]:
    name = re.search(r"\.name:\s+(\S+)", blk)
    if not name:
        continue
    dem = subprocess.run(["c++filt", name.group(1)], capture_output=True, text=True).stdout.strip()
    if not any(p in dem for p in pats):
        continue
    f = lambda k: (re.search(r"\." + k + r":\s+(\d+)", blk) or [None, "?"])[1]
    print("%-95s vgpr %s sgpr %s vspill %s sspill %s scratch %s lds %s" % (dem[:95], f("vgpr_count"), f("sgpr_count"), f("vgpr_spill_count"), f("sgpr_spill_count"),
                                                                        f("private_segment_fixed_size"), f("group_segment_fixed_size")))
