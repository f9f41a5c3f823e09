"""The N > 1 code paths of bench.py on a ONE-GPU box (multi-GPU unexercised above world 1): KZG_BENCH_REHEARSAL=1 lets
`python bench.py --gpus 2` start its two ranks on GPU 0 (small table budget each, gloo for the harness's own collectives).
What this checks: the launcher with real workers, per-rank batches, the gathered proof vector against every rank's own
proofs, the strong-scaling legs (configs 4 and 5 split two ways, gathered bytes == the one-GPU output), the one-line JSON
record, and that a library communicator that cannot be built (RCCL refuses two ranks on one GPU) is REPORTED, not hidden.
It measures nothing: the ranks share the GPU."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _rehearse(extra):
    env = dict(os.environ, KZG_BENCH_REHEARSAL="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "ETH_KZG_AMD_TABLE_GB"):  # (the suite's "max" budget is for ONE context per GPU: the rehearsal sets its own)
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--blobs-per-gpu", "128"] + extra,
                       env=env, capture_output=True, text=True, timeout=400)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout[:2000]  # stdout carries exactly the record
    return json.loads(lines[0]), r.stderr


@pytest.mark.timeout(900)
def test_two_ranks_on_one_gpu_run_the_sharded_paths():
    d, _ = _rehearse(["--exchange", "torch"])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["config"]["ranks"] == 2
    assert d["config"]["launcher"] == "bench.py" and d["config"]["rehearsal"]
    assert d["config"]["gathered_proofs_checked"] is True
    strong = d["configs_strong"]
    for key in ("config4_compute_512_blobs", "config5_recover_256_blobs_half_erased"):
        assert strong[key]["gathered_equals_one_gpu_output"] is True and strong[key]["blobs_per_rank"] * 2 >= strong[key]["total_blobs"]
    assert "cpu_baseline" not in d and "configs" not in d  # rank-0-at-N=1 legs only
    # beside the process-per-GPU form, rank 0 alone drives ONE context over the device list (here 0,0)
    leg = d["single_process_device_list"]
    assert "error" not in leg, leg
    assert leg["devices"] == [0, 0] and leg["blobs"] == 256 and leg["blobs_per_s"] > 0


@pytest.mark.timeout(900)
def test_a_library_communicator_that_cannot_be_built_is_reported():
    d, err = _rehearse([])  # default --exchange library: RCCL refuses two ranks on one GPU
    assert d["n_gpus"] == 2 and d["config"]["gathered_proofs_checked"] is True
    assert d["config"]["library_communicator_error"] and "FAILED" in d["config"]["exchange"]
    assert "library's RCCL communicator is unavailable" in err
