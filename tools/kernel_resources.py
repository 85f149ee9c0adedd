"""tools/kernel_resources.py -- VGPR / SGPR / LDS / scratch of every kernel in libfpx's code objects (llvm-readelf --notes on
the gfx950 code object bundled in build/*.o): the table occupancy claims are checked against.  Writes profiles/<name>."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = "/opt/rocm/lib/llvm/bin"
out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", "r03_kernel_resources.txt")
rows = []
for obj in sorted(os.listdir(os.path.join(ROOT, "acoustid-index_amd", "build"))):
    if not obj.endswith(".o"):
        continue
    path = os.path.join(ROOT, "acoustid-index_amd", "build", obj)
    with tempfile.TemporaryDirectory() as d:
        co, fat = os.path.join(d, "dev.co"), os.path.join(d, "fat.bin")
        r = subprocess.run([f"{BIN}/llvm-objcopy", f"--dump-section=.hip_fatbin={fat}", path], capture_output=True)
        if r.returncode != 0 or not os.path.exists(fat):
            continue
        r = subprocess.run([f"{BIN}/clang-offload-bundler", "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                            f"--input={fat}", f"--output={co}"], capture_output=True)
        if r.returncode != 0 or not os.path.exists(co) or os.path.getsize(co) == 0:
            continue
        notes = subprocess.run([f"{BIN}/llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
    for m in re.finditer(r"- \.agpr_count:.*?(?=\n\s+- \.agpr_count:|\namdhsa\.target|\Z)", notes, flags=re.S):
        blk = m.group(0)
        def f(k):
            mm = re.search(rf"\.{k}:\s+(\S+)", blk)
            return mm.group(1) if mm else "?"
        name = f("name")
        try:
            name = subprocess.run([f"{BIN}/llvm-cxxfilt", name], capture_output=True, text=True).stdout.strip().split("(")[0]
        except OSError:
            pass
        if "rocprim" in name or "CPRIM" in name:
            name = "rocprim::" + name.split("::")[-1][:40]
        rows.append((obj, name[-70:], f("vgpr_count"), f("agpr_count"), f("sgpr_count"), f("vgpr_spill_count"), f("sgpr_spill_count"),
                     f("group_segment_fixed_size"), f("private_segment_fixed_size"), f("max_flat_workgroup_size")))
with open(out, "w") as fh:
    fh.write("# llvm-readelf --notes of the gfx950 code objects in acoustid-index_amd/build/*.o (tools/kernel_resources.py)\n")
    fh.write(f"{'object':14s} {'kernel':70s} {'vgpr':>5s} {'agpr':>5s} {'sgpr':>5s} {'vspill':>6s} {'sspill':>6s} {'lds_static':>10s} {'scratch':>8s} {'wg_max':>6s}\n")
    for r in rows:
        fh.write(f"{r[0]:14s} {r[1]:70s} {r[2]:>5s} {r[3]:>5s} {r[4]:>5s} {r[5]:>6s} {r[6]:>6s} {r[7]:>10s} {r[8]:>8s} {r[9]:>6s}\n")
print(open(out).read())
