import re, sys, gzip, numpy as np, ctypes, os
sys.path.insert(0, ".")
md = open("INTEGRATION.md").read()
blocks = re.findall(r"```python\n(.*?)```", md, re.S)
src = blocks[0].replace('ctypes.CDLL("libkrisp_hip.so")', 'ctypes.CDLL(os.path.abspath("krisp_amd/libkrisp_hip.so"))')
ns = {"os": os}
exec(src, ns)
exec(blocks[1], ns)
from krisp_amd import fasta
d = "tests/golden/c1"
files = [f"{d}/ingroup0.fasta.gz", f"{d}/ingroup1.fasta.gz", f"{d}/outgroup0.fasta.gz", f"{d}/outgroup1.fasta.gz", f"{d}/outgroup2.fasta.gz"]
texts = [fasta.to_bases(fasta.read_records(f)).tobytes() for f in files]
recs = ns["diagnostic_records"](texts, [1, 1, 0, 0, 0], 25, 1, 2, False)
print("records", len(recs), sorted(set(int(k) >> 8 for k in recs["key"]))[:2])
ctx = ns["lib"].kr_create(0, 0)
hits = ns["diagnostic_hits"](ctx, texts, [1, 1, 0, 0, 0], 30, 40, 30, False)
print("wide hits", len(hits), sorted(set(hits["cand"].tolist())))
ns["lib"].kr_destroy(ctx)
