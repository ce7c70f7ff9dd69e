set -e
cd $GRAFT_REPO_ROOT
python - <<'PY'
import sys
sys.path.insert(0,'.')
from breakfast_amd.synth import generate_profiles
rows = generate_profiles(1000000)
with open('/tmp/b.tsv','w') as f:
    f.write("accession\tdna_profile\n")
    for i,r in enumerate(rows): f.write(f"seq{i:07d}\t{r}\n")
PY
python -m breakfast_amd --input-file /tmp/b.tsv --outdir /tmp/o0 --output-cache /tmp/c.bfkc > /dev/null
for k in 1 2; do
echo "--- in + out, run $k"
BFK_FRONT_TIMING=1 python -m breakfast_amd --input-file /tmp/b.tsv --outdir /tmp/o$k --input-cache /tmp/c.bfkc --output-cache /tmp/d.bfkc 2>&1 >/dev/null | grep -E "pipeline|prepare:|open:|waited" 
done
