"""Register / spill / LDS table of every kernel in the built objects (facialmmt_amd/build*/*.o): python tools/kernel_regs.py [objdir]"""
import glob, os, re, subprocess, sys
LLVM = "/opt/rocm/lib/llvm/bin/"
objdir = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "facialmmt_amd", "build")
for o in sorted(glob.glob(os.path.join(objdir, "*.o"))):
    subprocess.run([LLVM + "llvm-objcopy", "--dump-section", ".hip_fatbin=/tmp/_fat.bin", o], check=True)
    out = subprocess.run([LLVM + "clang-offload-bundler", "--list", "--type=o", "--input=/tmp/_fat.bin"], capture_output=True, text=True).stdout
    tgt = [t for t in out.split() if "gfx950" in t]
    if not tgt:
        continue
    subprocess.run([LLVM + "clang-offload-bundler", "--unbundle", "--type=o", "--input=/tmp/_fat.bin", "--targets=" + tgt[0], "--output=/tmp/_dev.co"], check=True)
    notes = subprocess.run([LLVM + "llvm-readelf", "--notes", "/tmp/_dev.co"], capture_output=True, text=True).stdout
    kern = notes.split("amdhsa.kernels:")[1].split("amdhsa.target:")[0]
    for b in re.split(r"\n  - ", kern)[1:]:
        def g(k):
            m = re.search(re.escape(k) + r":\s+(\d+)", b)
            return m.group(1) if m else "-"
        sym = re.search(r"\.name:\s+(\S+)", b).group(1)
        name = subprocess.run(["c++filt", sym], capture_output=True, text=True).stdout.strip().replace("(anonymous namespace)::", "")
        print(f"{os.path.basename(o):18s} vgpr {g('.vgpr_count'):>4} agpr {g('.agpr_count'):>4} vspill {g('.vgpr_spill_count'):>3} sspill {g('.sgpr_spill_count'):>3} lds {g('.group_segment_fixed_size'):>6}  {name[:120]}")
