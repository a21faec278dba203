"""`python bench.py --gpus N` without a launcher must start the N ranks itself: rehearsed here on CPU with the
gloo backend (--dry-run: launch, rendezvous, contiguous sharding, chunking, the SAM gather in rank order; placeholder records)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(cmd, env=None):
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    p = subprocess.run(cmd, cwd=ROOT, env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    lines = [ln for ln in p.stdout.decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout.decode()          # ONE JSON line, from rank 0
    return json.loads(lines[0])


def test_bench_starts_its_own_ranks():
    d = _run([sys.executable, "bench.py", "--gpus", "2", "--dry-run", "--total-reads", "5000", "--reads", "1500"])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["dry_run"] is True
    assert d["launched_by"] == "bench.py"
    assert d["config"]["reads_per_rank"] == [2500, 2500] and d["config"]["chunks_per_rank"] == [2, 2]
    g = d["gather"]
    assert g["identical_to_unsharded"] is True and len(g["per_rank_bytes"]) == 2 and g["bytes"] == sum(g["per_rank_bytes"])


def test_bench_under_torchrun_uses_the_launcher_ranks():
    d = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "3", "--master-addr", "127.0.0.1",
              "--master-port", str(29600 + os.getpid() % 300), "bench.py", "--gpus", "3", "--dry-run", "--total-reads", "1000"])
    assert d["n_gpus"] == 3 and d["launched_by"] == "external launcher"
    assert d["config"]["reads_per_rank"] == [333, 333, 334] and d["gather"]["identical_to_unsharded"] is True


def test_chunk_bounds():
    sys.path.insert(0, ROOT)
    import bench
    assert bench.chunk_bounds(1250000, 1000000) == [0, 625000, 1250000]
    assert bench.chunk_bounds(1000000, 1000000) == [0, 1000000]
    assert bench.chunk_bounds(0, 1000000) == [0, 0]
    b = bench.chunk_bounds(5000000, 1000000)
    assert len(b) == 6 and all(b[i + 1] - b[i] == 1000000 for i in range(5))


def test_chunk_count_is_a_multiple_of_the_contexts():
    """a rank's share of a sharded read set is cut into equal chunks of at most --reads reads, a multiple of --inflight of them when more than one is needed"""
    import importlib.util, os
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    for n, cap, k, want in ((1000000, 1000000, 2, 1), (1250000, 1000000, 2, 2), (2500000, 1000000, 2, 4), (2500000, 1000000, 1, 3), (5000000, 1000000, 2, 6), (10000000, 1000000, 3, 12), (0, 1000000, 2, 1)):
        cb = b.chunk_bounds(n, cap, k)
        sizes = [cb[i + 1] - cb[i] for i in range(len(cb) - 1)]
        assert len(sizes) == want and sum(sizes) == n and (not sizes or max(sizes) <= cap) and max(sizes) - min(sizes) <= 1
