import sys, os, time, subprocess
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(sys.path[0], "tests"))
import numpy as np
from slamem_amd import synth
rng = np.random.default_rng(1)
n = 4_641_652
ref = synth.make_reference(n, seed=11)
alpha = np.frombuffer(b"ACGT", dtype=np.uint8)
q = ref.copy()
m = rng.random(n) < 0.015
q[m] = rng.choice(alpha, size=int(m.sum()))
comp = np.zeros(256, dtype=np.uint8)
for x, y in zip(b"ACGT", b"TGCA"): comp[x] = y
for a in (500_000, 2_000_000, 3_500_000):   # three inversions
    q[a:a+40_000] = comp[q[a:a+40_000][::-1]]
q = np.delete(q, np.r_[1_000_000:1_040_000, 3_000_000:3_044_046])  # two deletions
d = "/tmp/ec"; os.makedirs(d, exist_ok=True)
synth.write_fasta_reference(d + "/ref.fa", ref, "ecoli_like_ref")
synth.write_fasta_reference(d + "/qry.fa", q, "ecoli_like_strain")
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
t0 = time.time()
r = subprocess.run([root + "/slamem_amd/host/slaMEM-hip", "-b", "-l", "20", "-o", d + "/out.txt", d + "/ref.fa", d + "/qry.fa"], stdout=subprocess.PIPE)
print("CLI wall", round(time.time() - t0, 3), "s rc", r.returncode)
print(r.stdout.decode()[-700:])
# oracle check
from oracle import pyoracle as po
import hostlib
t0 = time.time()
idx = po.OracleIndex(ref.tobytes())
off = np.array([0, len(q)], dtype=np.uint64)
mems, bc = idx.match_batch(q, off, 20, True)
print("oracle build+match", round(time.time() - t0, 2), "s", len(mems), bc)
refl = hostlib.Loaded(d + "/ref.fa", 1)
tri = np.stack([mems["ref_pos"], mems["query_pos"], mems["length"]], axis=1).astype(np.uint32)
exp = hostlib.format_block(b"ecoli_like_strain", 0, tri[: int(bc[0])], refl) + hostlib.format_block(b"ecoli_like_strain", 1, tri[int(bc[0]):], refl)
got = open(d + "/out.txt", "rb").read()
print("byte-identical to oracle:", got == exp, len(got))
