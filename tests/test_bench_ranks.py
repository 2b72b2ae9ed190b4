"""The multi-rank path of bench.py rehearsed on CPU ranks (gloo, world size 2): self-launch from a plain
`python bench.py --gpus 2`, sharding, barriers, the MAX reduction, the double-buffered scatter -> hop -> gather loop and
the JSON fields the driver reads.  DN_BENCH_REHEARSAL=1 swaps the hop for `out = 2 * frames` (no kernels here: this
container has no GPU); the real thing runs in tests/test_gpu_parity.py::test_bench_two_ranks_on_one_gpu and on the driver's node."""
import json
import os
import subprocess
import sys

from conftest import REPO


def _run(args, env_extra, timeout=300):
    env = dict(os.environ, **env_extra)
    env.pop("WORLD_SIZE", None), env.pop("RANK", None), env.pop("LOCAL_RANK", None)
    return subprocess.run([sys.executable, os.path.join(REPO, "bench.py")] + args, capture_output=True, text=True, env=env, timeout=timeout, cwd=REPO)


def test_bench_self_launches_two_gloo_ranks_and_reports_both_ingress_variants():
    r = _run(["--gpus", "2", "--steps", "4", "--warmup", "1", "--batch", "6"], {"DN_BENCH_REHEARSAL": "1", "DN_DIST_BACKEND": "gloo"})
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1                                   # rank 0 prints ONE JSON line
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["ranks_seen"] == 2 and d["scaling"] == "weak" and d["ingress"] == "local"
    assert d["steps"] == 4 and d["warmup"] == 1 and d["config"]["frames_per_step"] == 12
    assert d["data"].startswith("rehearsal")                 # never mistaken for a measurement
    iv = d["ingress_variant"]
    assert iv["ingress"] == "scatter_gather" and iv["backend"] == "gloo" and iv["root_output_finite"] is True
    assert iv["bytes_per_step_each_way"] == 12 * 1024 * 4 and iv["value"] > 0


def test_bench_ingress_variant_runs_the_headlines_hop_groups_on_two_gloo_ranks():
    """The scatter -> launch -> gather loop in units of four hops per launch (the headline's schedule since round 4), rank plumbing only."""
    r = _run(["--gpus", "2", "--steps", "8", "--warmup", "1", "--batch", "5"], {"DN_BENCH_REHEARSAL": "1", "DN_DIST_BACKEND": "gloo", "DN_REHEARSAL_GROUP": "4"})
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    iv = d["ingress_variant"]
    assert d["ingress_ok"] is True and iv["hops_per_launch"] == 4 and iv["root_output_finite"] is True and iv["value"] > 0


def test_bench_refuses_a_rank_count_that_does_not_match():
    env = {"DN_BENCH_REHEARSAL": "1", "DN_DIST_BACKEND": "gloo", "WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"}
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, env=dict(os.environ, **env), timeout=120, cwd=REPO)
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr


def test_bench_needs_a_gpu_outside_rehearsal():
    r = _run(["--steps", "1", "--warmup", "0"], {"DN_BENCH_REHEARSAL": "0"})
    import torch
    if not torch.cuda.is_available():
        assert r.returncode != 0 and "needs a GPU" in r.stderr
