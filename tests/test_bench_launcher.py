"""bench.py's own launcher: `python bench.py --gpus N` without a launcher must start N ranks itself --
fresh child processes, spawned before the parent has touched the GPU (the parent never imports torch) -- relay rank 0's
JSON line and exit non-zero when a rank fails.  The CPU tests use --dry-run (rendezvous + barrier + relay, no GPU work);
the GPU test runs the real step with two ranks sharing the one GPU of the test box (gloo rehearsal)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra=None, timeout=600):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True,
                          timeout=timeout)


@pytest.mark.parametrize("n", [2, 3])
def test_direct_invocation_spawns_n_ranks(n):
    r = _run(["--gpus", str(n), "--dry-run", "--steps", "4", "--warmup", "1"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, "exactly one JSON line (rank 0's)"
    j = json.loads(lines[0])
    assert j["n_gpus"] == n and j["steps"] == 4 and j["warmup"] == 1 and j["dry_run"] is True
    assert "gloo" in j["config"]["collective"]          # the line says what ran, not what the product path would use


def test_eight_ranks_dry_run_with_the_collective_preflight():
    """The node run's shape without a GPU: eight ranks, the preflight's control flow (stage, vote, record) in the line."""
    r = _run(["--gpus", "8", "--dry-run", "--steps", "2", "--warmup", "1"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    j = json.loads(lines[0])
    assert j["n_gpus"] == 8 and j["metric"] == "grad+prox updates/sec" and j["unit"] == "updates/s"
    assert j["collective_preflight"]["torch"].startswith("ok") and j["collective_preflight"]["rccl"].startswith("ok")
    assert len(lines[0]) < 6000, "the line must fit the tail of a log"


@pytest.mark.parametrize("who", ["1", "all"])
def test_a_failed_native_rccl_preflight_still_ends_in_a_line(who):
    """One rank (or every rank) fails the native-RCCL stage: the vote sends EVERY rank to torch.distributed, the run completes and
    the line carries the record of what failed."""
    r = _run(["--gpus", "3", "--dry-run", "--steps", "2", "--warmup", "1"], {"CIAO_BENCH_FORCE_RCCL_FAIL": who})
    assert r.returncode == 0, r.stderr[-2000:]
    j = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert j["n_gpus"] == 3
    rec = j["collective_preflight"]
    assert rec["torch"].startswith("ok")
    assert ("failed" in rec["rccl"]) if who == "all" else ("failed on another rank" in rec["rccl"]), rec   # rank 0's own view
    assert "fell back to torch.distributed" in j["config"]["collective"]


def test_a_hung_preflight_stage_ends_the_rank_with_the_record():
    """The watchdog of a stage: a collective that never returns costs the run, not the node -- the rank prints the record and exits 3."""
    code = ("import sys, time; sys.path.insert(0, %r); import bench; pf = bench.Preflight(0, 0.5); "
            "pf.stage('torch', lambda: None); pf.stage('rccl', lambda: time.sleep(30)); print('not reached')" % ROOT)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=60)
    assert r.returncode == 3 and "not reached" not in r.stdout
    assert "collective_preflight" in r.stderr and '"torch": "ok"' in r.stderr and "timeout after" in r.stderr


def test_rank_under_an_external_launcher_does_not_respawn():
    """With WORLD_SIZE set (torch.distributed.run) the process IS a rank: world 1 here, so it just prints its line."""
    r = _run(["--gpus", "1", "--dry-run"], {"WORLD_SIZE": "1", "RANK": "0"})
    assert r.returncode == 0 and json.loads(r.stdout.strip().splitlines()[-1])["n_gpus"] == 1


def test_a_failed_rank_fails_the_run():
    import torch
    if torch.cuda.is_available():
        pytest.skip("needs the GPU-less container: there the real (non-dry) ranks die at context creation")
    r = _run(["--gpus", "2", "--steps", "1", "--rows-per-gpu", "100", "--no-cpu", "--no-extras"], {"CIAO_BENCH_BACKEND": "gloo"})
    assert r.returncode != 0
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")], "no result line may be relayed from a failed run"
    assert "failed with exit code" in r.stderr


def test_parent_never_imports_torch():
    """The parent must not be able to touch the GPU: the spawning path runs with torch made unimportable."""
    code = ("import sys; sys.modules['torch'] = None; sys.argv = ['bench.py', '--gpus', '2', '--dry-run']; "
            f"sys.path.insert(0, {ROOT!r}); import bench; rc = bench.main(); assert 'torch' not in "
            "[m for m in sys.modules if sys.modules[m] is not None]; sys.exit(rc)")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])["n_gpus"] == 2


@pytest.mark.gpu
def test_two_ranks_on_the_one_gpu_real_step():
    """The real sweep, two ranks sharing GPU 0 (RCCL takes one rank per device, so the rehearsal collective is gloo): the
    line must say n_gpus = 2, count both shards' rows, and name the collective that actually ran."""
    r = _run(["--gpus", "2", "--steps", "3", "--warmup", "1", "--rows-per-gpu", "200000", "--no-cpu", "--no-extras", "--no-chains"],
             {"CIAO_BENCH_BACKEND": "gloo"})
    assert r.returncode == 0, r.stderr[-3000:]
    j = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert j["n_gpus"] == 2 and j["config"]["N_total"] == 400000 and j["config"]["rows_per_gpu"] == 200000
    assert "gloo" in j["config"]["collective"] and "rccl" not in j["config"]["collective"].lower().replace("(= rccl)", "")
    assert j["rccl_ranks"] is None and j["allreduce_us_per_step"] is not None
    assert j["allreduce_us_per_step_peer_mailboxes"] is not None, "the peer exchange is timed beside whatever collective ran"
    assert j["value"] > 0 and j["roofline"]["kernel_launches"] == 3
    assert j["config"]["launcher"] == "bench.py spawned the ranks"
    # first-contact hardening: the preflight record of the paths that were used, and rank 0's cpu_baseline also with several ranks
    assert j["collective_preflight"]["torch"].startswith("ok") and j["collective_preflight"]["peer_probe"].startswith("ok"), j["collective_preflight"]


@pytest.mark.gpu
def test_two_ranks_rank0_reports_cpu_baseline_and_the_update_figures():
    """With --gpus N > 1 rank 0 still times the CPU baseline and the chain / Finito figures (on its own shard, on a context without a
    collective) after the timed region, while the other rank waits at the final barrier."""
    r = _run(["--gpus", "2", "--steps", "3", "--warmup", "1", "--rows-per-gpu", "200000", "--no-extras", "--cpu-rows", "20000",
              "--cpu-seconds", "0.5", "--cpu-chain-seconds", "0.5", "--cpu-chain-updates", "2000", "--chain-rows", "100000"],
             {"CIAO_BENCH_BACKEND": "gloo"})
    assert r.returncode == 0, r.stderr[-3000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    j = json.loads(line)
    assert j["n_gpus"] == 2 and j["cpu_baseline"]["value"] > 0 and j["cpu_baseline"]["cores"] == 1
    for key in ("svrg_updates_per_sec", "saga_updates_per_sec", "finito_samples_per_sec"):
        assert j[key]["value"] > 0 and 0 < j[key]["roofline"]["frac"] < 1 and j[key]["cpu_baseline"]["value"] > 0, (key, j[key])
    assert j["saga_updates_per_sec"]["f64"]["value"] > 0
    assert j["finito_samples_per_sec"]["share_512_rows"]["value"] > 0
    assert any(k.startswith("svrg_epochs_per_sec_N") for k in j)
    assert len(line) < 6000, len(line)


@pytest.mark.gpu
def test_two_ranks_on_the_one_gpu_real_step_through_the_peer_mailboxes():
    """CIAO_BENCH_COLLECTIVE=peer: the sweep's reductions travel through the peer mailboxes (csrc/peer_kernels.h), fused into
    finalize / epilogue; the line names that collective and reports its per-step cost."""
    r = _run(["--gpus", "2", "--steps", "3", "--warmup", "1", "--rows-per-gpu", "200000", "--no-cpu", "--no-extras", "--no-chains"],
             {"CIAO_BENCH_BACKEND": "gloo", "CIAO_BENCH_COLLECTIVE": "peer"})
    assert r.returncode == 0, r.stderr[-3000:]
    j = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert j["n_gpus"] == 2 and j["config"]["N_total"] == 400000
    assert "peer mailboxes" in j["config"]["collective"] and j["rccl_ranks"] is None
    assert j["allreduce_us_per_step"] is not None and j["allreduce_us_per_step"] == j["allreduce_us_per_step_peer_mailboxes"]
    assert j["value"] > 0 and j["roofline"]["kernel_launches"] == 3


def _torchrun(n, args, env_extra=None, timeout=600, port=29631):
    """The driver's own command for N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env.update(env_extra or {})
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", str(n)] + args
    return subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout)


def test_the_drivers_torchrun_command_line_dry():
    """Under torch.distributed.run every process is one rank (no respawn); rank 0 prints the one line."""
    r = _torchrun(2, ["--dry-run", "--steps", "3", "--warmup", "1"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["steps"] == 3 and j["config"]["launcher"] == "external launcher (WORLD_SIZE set)"


@pytest.mark.gpu
def test_the_drivers_torchrun_command_line_real_step():
    """The same command with the real sweep, two ranks sharing the one GPU of the test box (gloo rehearsal of the collective)."""
    r = _torchrun(2, ["--steps", "3", "--warmup", "1", "--rows-per-gpu", "200000", "--no-cpu", "--no-extras", "--no-chains"],
                  {"CIAO_BENCH_BACKEND": "gloo"}, port=29633)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["config"]["N_total"] == 400000 and j["value"] > 0
    assert j["config"]["launcher"] == "external launcher (WORLD_SIZE set)"
