"""SLAMEM_STREAM_TRACE=1 python tools/trace_leg.py [sizes in thousands of reads ...]: the stage timeline of the host-to-host leg
(stderr), optionally with an explicit batch schedule."""
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from slamem_amd import engine
dev = torch.device("cuda:0")
n, M, L = 100_000_000, 10_000_000, 150
ref = engine.synth_reference(n, 42, dev)
idx = engine.Index.build(ref, dev)
reads = engine.synth_reads(ref, 0, M, L, 0.02, 42, 50)
sched = [int(a) * 1000 for a in sys.argv[1:]] or None
r = engine.host_to_host_leg(idx, reads, M, L, 20, True, steps=1, batch_reads=1_000_000, slots=6, schedule=sched)
print(r["host_to_host_ms"], r["host_to_host"]["kernel_ms_sum"], file=sys.stderr)
# (host_to_host_leg closes its stream and pinned buffers; the index and the device tensors go here, in order, with the device
#  idle -- not at interpreter teardown: a profiler that traces copies otherwise waits 30 s for completions it cannot see)
idx.close()
del reads, ref
torch.cuda.synchronize(dev)
