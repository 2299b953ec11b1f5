"""VGPR / SGPR / LDS / scratch of the kernels in the built library (the code object's metadata notes).  usage: kernel_resources.py [substring ...]"""
import os, re, subprocess, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from settlers_of_catan_rl_amd import _lib
with tempfile.NamedTemporaryFile(suffix=".co") as f:
    f.write(_lib.device_code_object(sys.argv[1] if len(sys.argv) > 1 and sys.argv[1].endswith(".so") else None)); f.flush()
    t = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", f.name], stdout=subprocess.PIPE).stdout.decode()
want = [a for a in sys.argv[1:] if not a.endswith(".so")]
for blk in t.split("  - .agpr_count:")[1:]:
    g = lambda k: (re.search(r"\." + k + r":\s+(\S+)", blk) or [None, "?"])[1]
    name = subprocess.run(["c++filt", g("name")], stdout=subprocess.PIPE).stdout.decode().strip()
    name = re.sub(r"\(.*", "", name)
    if want and not any(w in name for w in want):
        continue
    print(f"{name:70s} vgpr {g('vgpr_count'):>4s} sgpr {g('sgpr_count'):>4s} lds {g('group_segment_fixed_size'):>7s} scratch {g('private_segment_fixed_size'):>5s} wg {g('max_flat_workgroup_size'):>5s}")
