set -e
mkdir -p gpurun_out/pair && cd gpurun_out/pair
python - <<'PY'
import sys
sys.path.insert(0,'../..'); sys.path.insert(0,'../../tests')
from golden_cases import ecoli_like_pair
from slamem_amd import synth
ref,qry=ecoli_like_pair()
synth.write_fasta_reference("ref.fa", ref, "ecoli_like_ref")
synth.write_fasta_reference("qry.fa", qry, "ecoli_like_strain")
PY
../../slamem_amd/host/slaMEM-hip -b -l 20 -o out.txt ref.fa qry.fa > log.txt 2>&1
SLAMEM_SEED=0 ../../slamem_amd/host/slaMEM-hip -b -l 20 -o out_noseed.txt ref.fa qry.fa > log2.txt 2>&1
wc -c out.txt out_noseed.txt
rm -f ref.fa qry.fa
