"""Kernel resource table from the code objects (llvm-readelf --notes): VGPRs (arch + acc), LDS, workgroup size.
rocprofv3's vgpr column reports half of the allocated registers for wave64 kernels; this is what decides co-residency."""
import re, subprocess, sys, glob, os
here = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "vognet-pytorch_amd", "csrc")
pat = sys.argv[1] if len(sys.argv) > 1 else "."
cos = [a for a in sys.argv[2:]]
if not cos:      # device-only code objects of every kernel translation unit, built on demand under /tmp (round 6: build.py no longer makes them)
    for tu in ("gemm", "attention", "elementwise", "lstm", "txtail", "visenc", "pair", "loss", "assemble", "backward"):
        src, co = os.path.join(here, tu + ".hip"), f"/tmp/kmeta_{tu}.co"
        hdrs = glob.glob(os.path.join(here, "*.h"))
        if not os.path.exists(co) or any(os.path.getmtime(f) > os.path.getmtime(co) for f in [src] + hdrs):
            subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "--cuda-device-only",
                            "--no-gpu-bundle-output", src, "-o", co], capture_output=True)
        cos.append(co)
for co in cos:
    txt = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
    for blk in txt.split("  - .agpr_count:")[1:]:
        g = lambda k: re.search(r"\.%s:\s+(\S+)" % k, blk)
        name = g("name").group(1)
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        dem = re.sub(r"\(.*\)$", "", dem).replace("vog::", "").replace("void ", "")
        if not re.search(pat, dem):
            continue
        agpr = int(blk.split()[0])
        print(f"{int(g('vgpr_count').group(1)):4d} vgpr {agpr:3d} agpr  lds_static {int(g('group_segment_fixed_size').group(1)):6d}  wg {int(g('max_flat_workgroup_size').group(1)):4d}  "
              f"scratch {int(g('private_segment_fixed_size').group(1)):4d}  {dem[:120]}")
