"""The W > 1 exchange code of the product, executed: m6a_comm_init -> m6a_gather / m6a_gather_reads with 2, 3 and
8 ranks SHARING the one GPU of a test box, through a test-only stand-in for librccl (tests/stub_rccl: the nccl* entry points over
/dev/shm files, bound with M6A_RCCL_LIB).  What this proves: gather_group's receive offsets and counts for ragged, empty and odd
shards, every destination, device and host pointers, the group closed when a Send fails, the CLI's M6A_EXCHANGE=rccl leg.  What it
cannot prove: RCCL's own transport over xGMI -- that needs a node."""
import os
import subprocess
import sys
import tempfile

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)


@pytest.fixture(scope="module")
def stub_lib(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("stub") / "libstub_rccl.so")
    subprocess.check_call([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "-O1", "-std=c++17", "-fPIC", "-shared", "-pthread",
                           os.path.join(REPO, "tests", "stub_rccl", "stub_rccl.cpp"), "-o", out])
    return out


def run_world(stub_lib, world, scenario, extra_env=None, timeout=600):
    xdir = tempfile.mkdtemp(prefix="m6a_stub_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    env = dict(os.environ, M6A_RCCL_LIB=stub_lib, HSA_ENABLE_IPC_MODE_LEGACY="0", STUB_RCCL_LOG=os.path.join(xdir, "log"))
    env.pop("STUB_RCCL_FAIL_SEND", None)
    procs = []
    for r in range(world):
        e = dict(env, **{k: v for k, v in (extra_env or {}).get(r, {}).items()})
        procs.append(subprocess.Popen([sys.executable, os.path.join(REPO, "tests", "stub_rccl", "worker.py"), str(r), str(world), xdir, scenario],
                                      env=e, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = []
    try:
        for p in procs:
            outs.append(p.communicate(timeout=timeout)[0])
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    log = open(os.path.join(xdir, "log")).read().splitlines() if os.path.exists(os.path.join(xdir, "log")) else []
    ok = [os.path.exists(os.path.join(xdir, "ok%d" % r)) for r in range(world)]
    import shutil
    shutil.rmtree(xdir, ignore_errors=True)
    assert all(p.returncode == 0 for p in procs) and all(ok), "\n".join("--- rank %d rc=%s\n%s" % (i, p.returncode, o[-1500:]) for i, (p, o) in enumerate(zip(procs, outs)))
    return log


@pytest.mark.parametrize("world", [2, 8])
def test_native_gather_world_n_via_stub_transport(stub_lib, world):
    """The job cut by m6a_shard_plan over `world` ranks, every rank's m6a_infer on its shard with its job offset, ONE m6a_gather +
    one m6a_gather_reads to rank 0 through the library's own communicator: gathered == the unsharded job, bit for bit -- device
    pointers and host pointers.  Every rank executed real sends, rank 0 real receives from every rank (the stub's log)."""
    log = run_world(stub_lib, world, "planned")
    sends = [l for l in log if l.startswith("send")]
    recvs = [l for l in log if l.startswith("recv")]
    assert {int(l.split("rank=")[1].split()[0]) for l in sends} == set(range(world))
    assert {int(l.split("peer=")[1].split()[0]) for l in recvs} == set(range(world)) and all("rank=0 " in l for l in recvs)
    assert len(sends) == len(recvs) == 6 * world                       # (site + mod + reads) x (device, host) per rank


@pytest.mark.parametrize("world", [2, 3, 8])
def test_native_gather_ragged_empty_and_odd_shards_every_destination(stub_lib, world):
    """Hand-made cuts with empty, one-site and odd-sized shards, destinations first / middle / last, device and host pointers:
    values encode their global index, so a wrong receive offset ((cuts[r] - cuts[0]) * esz) or count cannot pass."""
    log = run_world(stub_lib, world, "ragged")
    assert not any("bytes=0 " in l for l in log)                       # empty shards post no send and no receive


def test_native_gather_a_failing_send_leaves_no_open_group(stub_lib):
    """The destination's own first ncclSend fails: m6a_gather reports the RCCL error, the thread's group is closed
    (stub_rccl_group_depth() == 0), and a new communicator on the same context gathers correctly."""
    run_world(stub_lib, 3, "failsend", extra_env={0: {"STUB_RCCL_FAIL_SEND": "1"}})


def test_cli_gpus_with_the_rccl_exchange_leg_via_stub_transport(stub_lib, tmp_path):
    """`m6anet_amd inference --gpus 3` with M6A_EXCHANGE=rccl -- the launcher's id hand-off through the exchange directory,
    m6a_comm_init on three ranks, ncclCommCount checked against the world, m6a_gather to rank 0, rank 0's check of the gathered
    results -- on one GPU through the stand-in transport: the CSV bytes are those of the one-GPU run."""
    data = os.path.join(REPO, "tests", "golden", "ref_tests_data")
    outs = {}
    for gpus, env in ((1, {}), (3, {"M6A_EXCHANGE": "rccl", "M6A_SHARE_GPU": "1", "M6A_RCCL_LIB": stub_lib, "M6A_RCCL_STANDIN": "1"})):
        out = str(tmp_path / ("o%d" % gpus))
        e = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", **env)
        r = subprocess.run([sys.executable, "-m", "m6anet_amd", "inference", "--input_dir", data, "--out_dir", out, "--num_iterations", "50",
                            "--n_processes", "1", "--gpus", str(gpus)], cwd=REPO, env=e, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs[gpus] = [open(os.path.join(out, f), "rb").read() for f in ("data.site_proba.csv", "data.indiv_proba.csv")]
    assert outs[1] == outs[3] and len(outs[1][0]) > 1000
    # a REAL librccl is still refused for ranks that share a GPU (it would fail inside RCCL): only a named stand-in may
    r = subprocess.run([sys.executable, "-m", "m6anet_amd", "inference", "--input_dir", data, "--out_dir", str(tmp_path / "x"), "--num_iterations", "5",
                        "--gpus", "2"], cwd=REPO, env={k: v for k, v in dict(os.environ, M6A_EXCHANGE="rccl", M6A_SHARE_GPU="1").items() if k not in ("M6A_RCCL_LIB", "M6A_RCCL_STANDIN")},
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "one GPU per rank" in r.stderr
