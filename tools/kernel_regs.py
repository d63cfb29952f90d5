"""VGPR / SGPR / spill / LDS figures of every kernel in a hipcc -save-temps .s file (or of a source: compiled to /tmp)."""
import re, subprocess, sys, os
src = sys.argv[1]
if src.endswith(".hip"):
    import tempfile
    d = tempfile.mkdtemp(prefix="kr_")
    out = os.path.join(d, "k")
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-munsafe-fp-atomics", "-fPIC",
                    "-Wno-unused-function", "-c", src, "-o", out + ".o", "-save-temps=obj"] + sys.argv[2:], check=True)
    src = os.path.join(d, os.path.splitext(os.path.basename(src))[0] + "-hip-amdgcn-amd-amdhsa-gfx950.s")
txt = open(src).read()
for blk in txt.split("  - .agpr_count:")[1:]:
    g = lambda k: (re.search(r"\." + k + r":\s*(\S+)", blk) or [None, "?"])[1]
    name = subprocess.run(["c++filt", g("name")], capture_output=True, text=True).stdout.strip().split("(")[0]
    print(f"{name:90s} vgpr {g('vgpr_count'):>4s} spill {g('vgpr_spill_count'):>3s} sgpr {g('sgpr_count'):>4s} "
          f"sspill {g('sgpr_spill_count'):>3s} lds {g('group_segment_fixed_size'):>6s} scratch {g('private_segment_fixed_size')}")
