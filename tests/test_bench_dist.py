"""bench.py's multi-process plumbing on CPU: two gloo ranks, request sharding with
no data-path collective, barrier + max-over-ranks timing, whole-job aggregation."""
import json
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent("""
    import json, os, sys, time
    sys.path.insert(0, %r)
    import bench
    env = bench.DistEnv(2, backend="gloo")
    ids = bench.shard_requests(4, env.rank, env.world)
    work = {"n": 0}
    def step():
        time.sleep(0.002 * (1 + env.rank))     # rank 1 is the slow one
        work["n"] += len(ids)
    elapsed = bench.timed_steps(env, lambda: None, step, warmup=2, steps=5)
    total = env.reduce(float(work["n"]), "SUM")
    # the path the tensor-parallel bench uses to hand the communicator id from rank 0 to the others
    ident = bytes(range(128)) if env.rank == 0 else None
    got = env.broadcast_bytes(ident, 128)
    same = env.reduce(1.0 if got == bytes(range(128)) else 0.0, "SUM")
    if env.rank == 0:
        print(json.dumps({"ids0": ids, "elapsed": elapsed, "total": total, "world": env.world, "id_ok": same}))
    env.close()
""") % ROOT


def test_two_rank_gloo_sharding_and_timing(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29611", OMP_NUM_THREADS="1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", "29611", str(script)],
                         env=env, capture_output=True, text=True, timeout=240)
    assert out.returncode == 0, out.stderr[-3000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    r = json.loads(line)
    assert r["world"] == 2
    assert r["ids0"] == [0, 2, 4, 6]              # request r -> replica r mod world
    assert r["total"] == 2 * 4 * 7                # every rank ran warmup + steps on its own 4 requests
    assert r["elapsed"] >= 5 * 0.004 * 0.9        # the MAX over ranks: the slow rank's time
    assert r["id_ok"] == 2.0                      # every rank holds rank 0's 128-byte id


def test_shards_partition_the_request_stream():
    import bench
    for world in (1, 2, 4, 8):
        seen = []
        for rank in range(world):
            ids = bench.shard_requests(64, rank, world)
            assert len(ids) == 64
            seen += ids
        assert sorted(seen) == list(range(64 * world))
