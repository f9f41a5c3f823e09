"""bench.py --gpus N: the launcher starts one rank per GPU itself when no torchrun-style environment is present
(the driver's command shape is `python bench.py --gpus N ...`).  CPU only: --launcher-selftest stops
before any GPU work; the ranks meet on a gloo rendezvous and rank 0 prints one JSON line, relayed by the parent."""
import importlib
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _run(extra, env=None, timeout=240):
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra, env=e, capture_output=True, text=True, timeout=timeout)


@pytest.mark.timeout(300)
def test_gpus_2_spawns_two_ranks_and_relays_rank0():
    r = _run(["--gpus", "2", "--launcher-selftest"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout  # ONE JSON line, from rank 0
    out = json.loads(lines[0])
    assert out["launcher_selftest"] and out["n_gpus"] == 2 and out["gpus_arg"] == 2
    ranks = sorted(out["ranks"], key=lambda x: x["rank"])
    assert [x["rank"] for x in ranks] == [0, 1] and [x["local_rank"] for x in ranks] == [0, 1]
    assert all(x["world"] == 2 and x["launcher"] == "bench.py" for x in ranks)
    assert ranks[0]["pid"] != ranks[1]["pid"] and ranks[0]["master"] == ranks[1]["master"]
    assert ranks[0]["master"].startswith("127.0.0.1:")


@pytest.mark.timeout(300)
def test_a_failing_rank_fails_the_launcher():
    r = _run(["--gpus", "2", "--launcher-selftest"], env={"KZG_BENCH_SELFTEST_FAIL_RANK": "1"})
    assert r.returncode == 7
    assert "rank 1 exited with status 7" in r.stderr


@pytest.mark.timeout(300)
def test_a_rank_stuck_in_the_communicator_init_is_timed_out_on_every_rank():
    """ncclCommInitRank is a collective and has never run at world > 1 here.  If one rank never returns from it,
    the watchdog of sharding.attach_library_comm (the code bench.py runs) gives every rank the same verdict after the timeout -- the
    library communicator is unusable, go on over torch's -- with the reason; nobody waits for the sleeper, nothing is re-executed,
    and the launcher still relays exactly one record and exits 0."""
    r = _run(["--gpus", "2", "--launcher-selftest"], env={"KZG_BENCH_SELFTEST_HANG_RANK": "1", "KZG_COMM_INIT_TIMEOUT_S": "2"})
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    wd = sorted(json.loads(lines[0])["watchdog"], key=lambda x: x["rank"])
    assert [w["attached"] for w in wd] == [False, False]
    assert [w["stuck"] for w in wd] == [False, True] and all("did not return within 2 s" in w["error"] for w in wd)
    assert all(w["waited_s"] < 30 for w in wd)  # (bounded by the timeout, not by the sleeper's hour)
    assert wd[0]["destroyed"] and not wd[1]["destroyed"]  # a communicator another thread is still building is left alone


@pytest.mark.timeout(300)
def test_external_launcher_environment_is_respected_and_checked():
    # under torchrun the environment is already there: no second level of children
    r = _run(["--gpus", "1", "--launcher-selftest"], env={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode == 0 and json.loads(r.stdout.strip().splitlines()[-1])["ranks"][0]["launcher"] == "external"
    # --gpus must agree with the launcher's world size
    r = _run(["--gpus", "4", "--launcher-selftest"], env={"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE=2" in r.stderr


def test_present_masks_of_a_2d_array_are_per_row():
    """rows of a 2-D numpy array are fresh views whose id() repeats; every blob must get ITS mask."""
    kzg = importlib.import_module("rust-eth-kzg_amd")
    m = kzg.present_masks(3, np.array([[0, 2, 4], [1, 3, 5], [6, 7, 64]]))
    assert [int(x) for x in m] == [0b10101, 0, 0b101010, 0, 0b11000000, 1]
    idx = list(range(0, 128, 2))
    m = kzg.present_masks(4, [idx] * 4)  # the shared-list fast path
    assert all(int(m[2 * b]) == 0x5555555555555555 and int(m[2 * b + 1]) == 0x5555555555555555 for b in range(4))
    rows = [[0, 1], (2, 3), np.array([127])]
    m = kzg.present_masks(3, rows)
    assert [int(x) for x in m] == [3, 0, 12, 0, 0, 1 << 63]
    with pytest.raises(kzg.KzgError):
        kzg.present_masks(1, [[128]])
