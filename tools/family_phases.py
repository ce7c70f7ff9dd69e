import sys
sys.path.insert(0, "/root/repo")
import numpy as np
from breakfast_amd import _lib, synth
fam, n, d = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
uf = list(dict.fromkeys(synth.generate_family(fam, n) if fam != "default" else synth.generate_profiles(n)))
ip, ix, _ = _lib.build_csr(uf, " ")
ctx = _lib.Context(0)
if len(sys.argv) > 4:
    ctx.set_candidate_path(sys.argv[4])
ctx.upload_csr(ip, ix)
d_out = ctx.alloc(4 * len(uf))
ctx.cluster(d, d_out); ctx.sync()
ctx.set_profiling(True)
for _ in range(8):
    ctx.cluster(d, d_out)
st = ctx.sync()
print({k: (round(v, 4) if isinstance(v, float) else v) for k, v in st.items()})
