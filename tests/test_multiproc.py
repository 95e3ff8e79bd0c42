"""N > 1 path on CPU: two gloo ranks shard the stream set (no data-path collective), run the host planner
for their shard, and reduce timing with MAX exactly as bench.py does."""
import os
import socket
import subprocess
import sys
import textwrap

import pytest

from audiomod_amd.sharding import shard_range

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_range_partitions():
    for total in (0, 1, 7, 128, 1024, 1000):
        for world in (1, 2, 3, 8):
            spans = [shard_range(total, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            for a, b in zip(spans, spans[1:]):
                assert a[1] == b[0]
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_range(8, 2, 2)


WORKER = textwrap.dedent("""
    import os, sys, json
    sys.path.insert(0, %r)
    import torch, torch.distributed as dist
    from audiomod_amd import engine as E
    from audiomod_amd.sharding import shard_range, max_over_ranks, sum_over_ranks
    dist.init_process_group(backend="gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    lo, hi = shard_range(1024, world, rank)            # cfg5: 1024 streams over the ranks
    # every rank plans its own shard (same config -> same plan; no data exchange needed)
    avail, shift, phase, info = E.plan_simulate([480] * 100, channels=2, semitones=4.0)
    mine = (hi - lo) * int(avail.sum())
    dist.barrier()
    total = sum_over_ranks(mine, dist)
    slowest = max_over_ranks(0.1 * (rank + 1), dist)
    if rank == 0:
        print(json.dumps({"total": total, "slowest": slowest, "per_stream": int(avail.sum()), "world": world}))
    dist.barrier()
    dist.destroy_process_group()
""")


def test_two_rank_gloo_sharding(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % ROOT)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
           "127.0.0.1", "--master-port", str(port), str(script)]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    import json
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    r = json.loads(line)
    assert r["world"] == 2
    assert r["total"] == 1024 * r["per_stream"]
    assert abs(r["slowest"] - 0.2) < 1e-12
